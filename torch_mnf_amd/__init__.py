"""torch_mnf_amd -- the coupling-flow hot path of janosh/torch-mnf on MI355X (gfx950).

Only the path is here: the Flow modules (``flows``), the MNF caller that feeds them
(``layers.MNFLinear``), the one cross-rank reduction (``dist``) and the HIP library they call
(``csrc/`` -> ``libmnf_hip.so``, C ABI in ``include/mnf_hip.h``).  Importing the package does
not touch the GPU; the library is loaded on first use and its absence is an error.

(The directory is ``torch_mnf_amd``: ``torch-mnf_amd`` is not an importable name.)
"""
from . import _dispatch, _lib
from ._lib import MnfHipError, deterministic, last_kernel
from .layers import MNFConv2d, MNFFeedForward, MNFLeNet, MNFLinear
from .train import FlatParameters, FusedAdam, GraphedStep
from .flows import (
    MLP,
    ActNormFlow,
    AffineConstantFlow,
    AffineHalfFlow,
    FusedAffineStack,
    FusedSplineBlock,
    Glow,
    IAF,
    MADE,
    MAF,
    MaskedLinear,
    NormalizingFlow,
    NormalizingFlowModel,
    NSF_AR,
    NSF_CL,
    RNVP,
    StandardNormal,
    rqs,
)

__all__ = [
    "last_kernel",
    "deterministic",
    "MLP", "MADE", "MaskedLinear", "MAF", "IAF", "ActNormFlow", "AffineConstantFlow", "AffineHalfFlow", "Glow", "NormalizingFlow",
    "NormalizingFlowModel", "NSF_AR", "NSF_CL", "RNVP", "StandardNormal", "FusedSplineBlock", "FusedAffineStack", "rqs", "MNFLinear", "MNFConv2d", "MNFLeNet", "MNFFeedForward", "FlatParameters", "FusedAdam", "GraphedStep", "MnfHipError", "library_path",
]


def library_path() -> str:
    return _lib.LIB_PATH
