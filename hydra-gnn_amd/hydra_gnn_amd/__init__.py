"""hydra_gnn_amd -- MI355X-native heterogeneous message-passing engine behind Hydra-GNN's model API.

Host-side mirror of the reference's ``hydra_gnn.models`` interface (``HeterogeneousNetwork``,
``HeterogeneousNeuralTreeNetwork``, ``HomogeneousNetwork``) over a C-ABI HIP library
(``libhydra_mp.so``, declared in ``include/hydra_mp.h``).  There is no CPU fallback: every compute
entry point raises if the HIP library is missing or no gfx950 device is visible.
"""
__version__ = "0.1.0"
