"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU, exports every function that
include/hydra_mp.h declares, the ctypes struct mirrors have the C layout, and compute entry points refuse to run
without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from hydra_gnn_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hydra_mp.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmp_[a-z0-9_]+)\s*\(", src)))


def test_library_exists_and_loads_without_gpu():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    assert lib.hmp_abi_version() == header_abi_version() == _lib.ABI_VERSION


def header_abi_version():
    return int(re.search(r"^#define\s+HMP_ABI_VERSION\s+(\d+)", open(HEADER).read(), re.M).group(1))


def test_build_entry_point_runs_and_agrees_with_the_header():
    """`__graft_entry__.build()` carried a hard-coded ABI number and raised at HEAD.  The driver's build check is this
    call: make (a no-op when the library is current), dlopen, bind every symbol, compare the ABI number with the header."""
    import importlib
    import sys

    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    assert ge._header_abi_version() == header_abi_version()
    text = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert not re.search(r"hmp_abi_version\(\)\s*==\s*\d", text), "no literal ABI number in the entry point"
    ge.build()


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/hydra_mp.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in hydra_gnn_amd/_lib.py"
    for n in _lib.SIGNATURES:
        assert n in names, f"{n} is bound but not declared in the header"


def test_struct_mirrors_match_c_layout():
    lib = _lib.load()
    for i, st in enumerate(_lib._STRUCTS):
        assert lib.hmp_sizeof(i) == C.sizeof(st), st.__name__


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box WITHOUT a GPU")
def test_no_cpu_fallback():
    lib = _lib.load()
    assert lib.hmp_device_count() == 0
    with pytest.raises(_lib.HydraMPError):
        _lib.require_device()
    spec = _lib.NetSpec()
    h = C.c_void_p()
    assert lib.hmp_net_create(C.byref(spec), C.byref(h)) != 0
    assert b"gfx950" in lib.hmp_last_error()
    from hydra_gnn_amd import workloads
    from hydra_gnn_amd.models import HeterogeneousNetwork

    net = HeterogeneousNetwork({"objects": 306, "rooms": 6}, output_dim=26, conv_block="GraphSAGE", hidden_dim=8, num_layers=2)
    with pytest.raises(_lib.HydraMPError):
        net(workloads.mp3d_like_batch(1, seed=1))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "hydra-gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"
