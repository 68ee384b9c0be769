"""MI355X-native BBFMM matvec for ferreus_rbf_rs's RBF solve hot path.

`FmmTree` mirrors ferreus_rbf_utils::FmmTree / py_ferreus_bbfmm.FmmTree; all passes run
as HIP kernels behind the C ABI in include/ferreus_bbfmm_hip.h.
"""
from .fmm_tree import (FmmError, FmmKernelType, FmmParams, FmmTree, KernelDoesNotSupportGradients,
                       KernelParams, KernelType, M2LCompressionType, PointOutsideTree,
                       SpheroidalOrder, mfma_f64_selftest, fp64_valu_selftest)

from . import solvers  # noqa: E402  (FGMRES / Schwarz drivers, iterative_solvers.rs)

__all__ = ["solvers", "FmmTree", "FmmParams", "KernelParams", "KernelType", "FmmKernelType",
           "SpheroidalOrder", "M2LCompressionType", "FmmError", "PointOutsideTree",
           "KernelDoesNotSupportGradients", "mfma_f64_selftest", "fp64_valu_selftest"]
