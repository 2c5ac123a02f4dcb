"""Mirror of the reference's ``hydra_gnn.models`` package (``src/hydra_gnn/models/__init__.py``)."""
from .heterogeneous_network import HeterogeneousNetwork
from .heterogeneous_neural_tree_network import HeterogeneousNeuralTreeNetwork, LeafPool
from .homogeneous_network import HomogeneousNetwork
from .homogeneous_neural_tree_network import HomogeneousNeuralTreeNetwork

__all__ = ["HeterogeneousNetwork", "HeterogeneousNeuralTreeNetwork", "HomogeneousNetwork", "HomogeneousNeuralTreeNetwork", "LeafPool"]
