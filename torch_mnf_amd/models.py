"""Import-path mirror of ``torch_mnf.models`` (models/__init__.py): ``MLP`` (the conditioner container of the coupling
layers) and the two containers of MNF layers.  (The plain ``LeNet`` of the reference has no flow in it and is not here.)"""
from .flows import MLP
from .layers import MNFFeedForward, MNFLeNet

__all__ = ["MLP", "MNFFeedForward", "MNFLeNet"]
