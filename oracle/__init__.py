"""ORACLE -- test infrastructure only (see pyg_ref.py header).  Parity status: unpinned for
operator outputs (PyG absent, reference tests hold no model outputs); data-side functions pinned
by the reference's own test vectors under tests/golden/."""
