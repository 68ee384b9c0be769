"""ctypes binding of libferreus_bbfmm_hip.so (the C ABI in include/ferreus_bbfmm_hip.h).

The product path fails loudly when the HIP library is missing: there is no Python or
CPU fallback for any compute entry point.
"""
from __future__ import annotations

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# FERREUS_BBFMM_HIP_LIB: another build of the same sources (the host-sanitizer libraries of build.py --sanitize)
LIB_PATH = os.environ.get("FERREUS_BBFMM_HIP_LIB") or os.path.join(HERE, "libferreus_bbfmm_hip.so")

# bbfmm_status
OK, POINT_OUTSIDE_TREE, KERNEL_NO_GRADIENTS, BAD_ARGUMENT, DEVICE_ERROR, UNSUPPORTED = range(6)
FLAG_HOST_ONLY = 1
FLAG_M2L_SHARED_BASIS = 2  # extension beyond the reference: M2L stages in one basis per level
FLAG_DIRECT_SMALL_W_LEAVES = 4  # extension beyond the reference: small W-list leaves as near field
FLAG_DETERMINISTIC = 8  # fixed summation order everywhere (no f64 atomics): bitwise reproducible results
N_PHASES = 11
PHASE_NAMES = ["gather", "P2M", "M2M", "M2L_stage1", "M2L_stage2", "P2L", "L2L", "P2P", "M2P",
               "L2P", "scatter"]

c_i32, c_i64, c_f64, c_u32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_uint32
c_p = ctypes.c_void_p


class Params(ctypes.Structure):
    """bbfmm_params <-> FmmParams (ferreus_bbfmm/src/bbfmm.rs:77-104)."""
    _fields_ = [("max_points_per_cell", c_i64), ("compression_type", c_i32),
                ("epsilon", c_f64), ("eval_chunk_size", c_i64)]


class TreeStats(ctypes.Structure):
    _fields_ = [("d", c_i32), ("order", c_i32), ("n_nodes", c_i32), ("depth", c_i32),
                ("n_points", c_i64), ("n_cells", c_i64), ("n_leaves", c_i64),
                ("n_u", c_i64), ("n_v", c_i64), ("n_w", c_i64), ("n_x", c_i64),
                ("p2p_pairs", c_i64), ("p2p_tile_bytes_k1", c_i64), ("m2l_flops_k1", c_f64),
                ("center", c_f64 * 3), ("radius", c_f64), ("wx_pairs", c_i64), ("wx_tile_bytes_k1", c_i64),
                ("m2l_basis_rank", c_i32), ("m2l_basis_len", c_i32),
                ("m2l_batches", c_i32), ("m2l_rhs_per_pass", c_i32), ("m2l_slots_bytes_per_rhs", c_i64),
                ("m2l_intermediate_bytes", c_i64)]


# every symbol include/ferreus_bbfmm_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bbfmm_params_defaults": (None, [c_i32, c_p]),
    "bbfmm_create": (ctypes.c_int, [c_p, c_i64, c_i32, c_i64, c_i32, c_i32, c_f64, c_f64, c_i32,
                                    c_i32, c_p, c_p, c_u32, c_p]),
    "bbfmm_create_on_devices": (ctypes.c_int, [c_p, c_i64, c_i32, c_i64, c_i32, c_i32, c_f64, c_f64, c_i32,
                                               c_i32, c_p, c_p, c_u32, c_p, c_i32, c_p]),
    "bbfmm_device_count": (c_i32, [c_p]),
    "bbfmm_part_device": (c_i32, [c_p, c_i32]),
    "bbfmm_group_bounds": (ctypes.c_int, [c_p, c_p]),
    "bbfmm_get_part_phase_ms": (ctypes.c_int, [c_p, c_i32, c_p, c_p]),
    "bbfmm_destroy": (None, [c_p]),
    "bbfmm_last_error": (ctypes.c_char_p, [c_p]),
    "bbfmm_set_weights": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64]),
    "bbfmm_set_local_coefficients": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64]),
    "bbfmm_evaluate": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p, c_i64, c_i64, c_p, c_i64,
                                      c_p]),
    "bbfmm_evaluate_with_gradients": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p, c_i64,
                                                     c_i64, c_p, c_i64, c_p, c_i64, c_p]),
    "bbfmm_evaluate_leaves": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p, c_i64, c_i64, c_p,
                                             c_i64, c_p]),
    "bbfmm_evaluate_leaves_with_gradients": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p,
                                                            c_i64, c_i64, c_p, c_i64, c_p, c_i64,
                                                            c_p]),
    "bbfmm_source_points": (ctypes.c_int, [c_p, c_p, c_i64]),
    "bbfmm_prepare_target_subset": (ctypes.c_int, [c_p, c_p, c_i64]),
    "bbfmm_fast_matrix_vector_product": (ctypes.c_int, [c_p, c_p, c_i64, c_i64, c_p, c_i64, c_p,
                                                        c_i64, c_f64, c_p]),
    "bbfmm_matvec_device": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_i64, c_i32]),
    "bbfmm_target_subset_create": (ctypes.c_int, [c_p, c_p, c_i64, c_p]),
    "bbfmm_matvec_subset_device": (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i32]),
    "bbfmm_stream": (c_p, [c_p]),
    "bbfmm_set_partition": (ctypes.c_int, [c_p, c_i32, c_i32]),
    "bbfmm_partition_row_count": (c_i64, [c_p]),
    "bbfmm_partition_rows": (ctypes.c_int, [c_p, c_p]),
    "bbfmm_partition_coarse_count": (c_i64, [c_p]),
    "bbfmm_matvec_partition_upward": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_p, c_p]),
    "bbfmm_matvec_partition_finish": (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_i32, c_p]),
    "bbfmm_partition_world": (c_i32, [c_p]),
    "bbfmm_partition_rank": (c_i32, [c_p]),
    "bbfmm_partition_bounds": (ctypes.c_int, [c_p, c_i32, c_p]),
    "bbfmm_matvec_partition_finish_sorted": (ctypes.c_int, [c_p, c_p, c_p, c_i64, c_p]),
    "bbfmm_partition_scatter": (ctypes.c_int, [c_p, c_p, c_i32, c_i32, c_i64, c_i32, c_p, c_i64]),
    "bbfmm_debug_partition_upward_counts": (ctypes.c_int, [c_p, c_p, c_p, c_p]),
    "bbfmm_debug_host_copy_rates": (ctypes.c_int, [c_i64, c_p]),
    "bbfmm_get_tree_stats": (ctypes.c_int, [c_p, c_p]),
    "bbfmm_tree_built_on_device": (ctypes.c_int, [c_p]),
    "bbfmm_last_evaluate_at_sources": (ctypes.c_int, [c_p]),
    "bbfmm_debug_targets_are_sources": (ctypes.c_int, [c_p, c_p, c_i64, c_i64]),
    "bbfmm_debug_rows_of_sources": (ctypes.c_int, [c_p, c_p, c_i64, c_i64, c_p]),
    "bbfmm_get_cells": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_get_leaf_sources": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_get_list": (ctypes.c_int, [c_p, ctypes.c_char, c_p, c_p, c_p]),
    "bbfmm_get_m2l_ranks": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_get_m2l_operator": (ctypes.c_int, [c_p, c_i32, c_i32, c_p]),
    "bbfmm_get_m2l_factors": (ctypes.c_int, [c_p, c_i32, c_i32, c_p, c_p]),
    "bbfmm_get_permutation_tables": (ctypes.c_int, [c_p, c_p, c_p, c_p, c_p, c_p]),
    "bbfmm_points_to_leaves": (ctypes.c_int, [c_p, c_p, c_i64, c_i64, c_p, c_p]),
    "bbfmm_set_profiling": (ctypes.c_int, [c_p, c_i32]),
    "bbfmm_get_phase_ms": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_reset_phase_ms": (ctypes.c_int, [c_p]),
    "bbfmm_mfma_f64_selftest": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_fp64_valu_selftest": (ctypes.c_int, [c_p, c_p]),
    "bbfmm_debug_dense_m2m": (ctypes.c_int, [c_p, c_i32, c_p]),
    "bbfmm_debug_apply_m2l_tables_host": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_debug_m2l_variants": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_debug_get_coefficients": (ctypes.c_int, [c_p, ctypes.c_char, c_i32, c_p]),
    "bbfmm_debug_morton_encode": (ctypes.c_uint64, [c_i32, c_p, ctypes.c_uint64]),
    "bbfmm_debug_morton_decode": (None, [c_i32, ctypes.c_uint64, c_p, c_p]),
    "bbfmm_debug_morton_neighbours": (c_i32, [c_i32, ctypes.c_uint64, c_p]),
    "bbfmm_debug_direction_vectors": (c_i32, [c_i32, c_p]),
    "bbfmm_debug_reference_vectors": (ctypes.c_int, [c_p, c_p, c_p]),
    "bbfmm_givens_rotation": (None, [c_f64, c_f64, c_p, c_p, c_p]),
    "bbfmm_fgmres": (ctypes.c_int, [c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_f64,
                                    c_p, c_p, c_p, c_p, c_p]),
    "bbfmm_schwarz_ddm_solver": (ctypes.c_int, [c_i64, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_f64,
                                                c_p, c_p, c_p, c_p, c_p]),
    "bbfmm_rbf_system_apply": (ctypes.c_int, [c_p, c_p, c_p, c_i64]),
    "bbfmm_ddm_params_defaults": (None, [c_p]),
    "bbfmm_ddm_params_for_points": (None, [c_i64, c_p]),
    "bbfmm_ddm_build": (ctypes.c_int, [c_p, c_i64, c_i32, c_i64, c_p, c_p]),
    "bbfmm_ddm_destroy": (None, [c_p]),
    "bbfmm_ddm_num_levels": (c_i32, [c_p]),
    "bbfmm_ddm_level_size": (c_i64, [c_p, c_i32]),
    "bbfmm_ddm_level_points": (ctypes.c_int, [c_p, c_i32, c_p]),
    "bbfmm_ddm_num_domains": (c_i64, [c_p, c_i32]),
    "bbfmm_ddm_domain_size": (c_i64, [c_p, c_i32, c_i64]),
    "bbfmm_ddm_domain": (ctypes.c_int, [c_p, c_i32, c_i64, c_p, c_p, c_p]),
    "bbfmm_schwarz_create": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p, c_p, c_p]),
    "bbfmm_schwarz_create_sharded": (ctypes.c_int, [c_p, c_p, c_i64, c_i32, c_i64, c_p, c_p, c_i32, c_i32, c_p, c_i64, c_p, c_p, c_p]),
    "bbfmm_schwarz_factor_bytes": (c_i64, [c_p]),
    "bbfmm_schwarz_domains_owned": (c_i64, [c_p, c_i32, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "bbfmm_schwarz_destroy": (None, [c_p]),
    "bbfmm_schwarz_basis_size": (c_i64, [c_p]),
    "bbfmm_schwarz_num_levels": (c_i32, [c_p]),
    "bbfmm_schwarz_monomial_matrix": (c_p, [c_p]),
    "bbfmm_debug_evaluate_monomials": (ctypes.c_int, [c_p, c_i64, c_i32, c_i64, c_i32, c_p, c_p, c_p]),
    "bbfmm_schwarz_apply": (ctypes.c_int, [c_p, c_p, c_p, c_i64]),
    "bbfmm_schwarz_level_size": (c_i64, [c_p, c_i32]),
    "bbfmm_schwarz_level_points": (ctypes.c_int, [c_p, c_i32, c_p]),
    "bbfmm_schwarz_debug_level_solve": (ctypes.c_int, [c_p, c_i32, c_p, c_p, c_i64, c_i32]),
}


class Interpolant(ctypes.Structure):
    """bbfmm_interpolant"""
    _fields_ = [("kernel_type", c_i32), ("polynomial_degree", c_i32), ("nugget", c_f64), ("base_range", c_f64),
                ("total_sill", c_f64), ("flags", c_u32)]


FLAG_GLOBAL_SCALING = 1


class DdmParams(ctypes.Structure):
    """bbfmm_ddm_params <-> DDMParams (ferreus_rbf/src/config.rs:42-69)."""
    _fields_ = [("leaf_threshold", c_i64), ("overlap_quota", c_f64), ("coarse_ratio", c_f64),
                ("coarse_threshold", c_i64)]

# bbfmm_apply_fn, bbfmm_iteration_fn
APPLY_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_p, ctypes.POINTER(c_f64), ctypes.POINTER(c_f64), c_i64)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, c_p, c_i64)     # bbfmm_allreduce_fn
ITERATION_FN = ctypes.CFUNCTYPE(None, c_p, c_i64, c_f64, c_f64)
ACCURACY_ABSOLUTE, ACCURACY_RELATIVE = 0, 1


class RbfSystem(ctypes.Structure):
    """bbfmm_rbf_system"""
    _fields_ = [("tree", c_p), ("basis_size", c_i64), ("monomial_matrix", c_p), ("ld_monomial", c_i64),
                ("nugget", c_f64)]

_lib = None


class LibraryMissing(RuntimeError):
    pass


def load():
    """Load the shared library (built by ferreus_rbf_rs_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m ferreus_rbf_rs_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process do not
    # both see the GPU.  Import torch first (when present) so this library binds to the copy
    # torch already loaded (same SONAME) and tensors / RCCL share one runtime with the kernels.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
