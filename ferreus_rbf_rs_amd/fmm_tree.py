"""Python host-side mirror of the reference's evaluator API over the C ABI.

Mirrors the only existing foreign-language binding of the seam, `py_ferreus_bbfmm`
(py_ferreus_bbfmm/src/python_bindings.rs:66-390): same class names, constructor
arguments, method names, array conventions and error behaviour, so the parity
tests read like the reference's examples.  All compute goes through
libferreus_bbfmm_hip.so (HIP kernels); nothing here computes potentials.
"""
from __future__ import annotations

import ctypes
import enum
from typing import Optional

import numpy as np

from . import _lib as L


class KernelType(enum.IntEnum):
    """ferreus_rbf_utils::KernelType (ferreus_rbf_utils/src/utils.rs:558-571)."""
    LinearRbf = 0
    ThinPlateSplineRbf = 1
    CubicRbf = 2
    Spheroidal3Rbf = 3
    Spheroidal5Rbf = 4
    Spheroidal7Rbf = 5
    Spheroidal9Rbf = 6
    Laplacian = 7
    OneOverR2 = 8
    OneOverR4 = 9
    # extension kernels of this repository (not in the reference)
    GaussianExt = 100
    MultiquadricExt = 101


class FmmKernelType(enum.Enum):
    """py_ferreus_bbfmm FmmKernelType (python_bindings.rs:66-76) + the two extensions."""
    LinearRbf = "LinearRbf"
    ThinPlateSplineRbf = "ThinPlateSplineRbf"
    CubicRbf = "CubicRbf"
    SpheroidalRbf = "SpheroidalRbf"
    Laplacian = "Laplacian"
    OneOverR2 = "OneOverR2"
    OneOverR4 = "OneOverR4"
    GaussianExt = "GaussianExt"
    MultiquadricExt = "MultiquadricExt"


class SpheroidalOrder(enum.Enum):
    """python_bindings.rs:78-85"""
    Three = 3
    Five = 5
    Seven = 7
    Nine = 9


class M2LCompressionType(enum.IntEnum):
    """ferreus_bbfmm::M2LCompressionType (bbfmm.rs:62-73; python name None_)."""
    None_ = 0
    SVD = 1
    ACA = 2


class FmmParams:
    """FmmParams (bbfmm.rs:77-104; python_bindings.rs:106-131)."""

    def __init__(self, max_points_per_cell: int, compression_type: M2LCompressionType,
                 epsilon: float, eval_chunk_size: int):
        self.max_points_per_cell = int(max_points_per_cell)
        self.compression_type = M2LCompressionType(compression_type)
        self.epsilon = float(epsilon)
        self.eval_chunk_size = int(eval_chunk_size)

    @staticmethod
    def new_defaults(interpolation_order: int) -> "FmmParams":
        p = L.Params()
        L.load().bbfmm_params_defaults(interpolation_order, ctypes.byref(p))
        return FmmParams(p.max_points_per_cell, M2LCompressionType(p.compression_type), p.epsilon,
                         p.eval_chunk_size)

    def _c(self) -> L.Params:
        return L.Params(self.max_points_per_cell, int(self.compression_type), self.epsilon,
                        self.eval_chunk_size)


class KernelParams:
    """KernelParams (kernel_helpers.rs:17-79; python_bindings.rs:133-188)."""

    def __init__(self, kernel_type, *, spheroidal_order: Optional[SpheroidalOrder] = None,
                 base_range: Optional[float] = None, total_sill: Optional[float] = None):
        if isinstance(kernel_type, KernelType):
            kt = kernel_type
        else:
            kernel_type = FmmKernelType(kernel_type)
            if kernel_type is FmmKernelType.SpheroidalRbf:
                order = spheroidal_order if spheroidal_order is not None else SpheroidalOrder.Three
                kt = {3: KernelType.Spheroidal3Rbf, 5: KernelType.Spheroidal5Rbf,
                      7: KernelType.Spheroidal7Rbf, 9: KernelType.Spheroidal9Rbf}[
                    SpheroidalOrder(order).value]
            else:
                kt = KernelType[kernel_type.name]
        self.kernel_type = kt
        self.base_range = 1.0 if base_range is None else float(base_range)     # builder defaults,
        self.total_sill = 1.0 if total_sill is None else float(total_sill)     # kernel_helpers.rs:25-31
        # KernelParamsBuilder::build asserts (kernel_helpers.rs:69-70)
        assert self.base_range > 0.0
        assert self.total_sill <= self.base_range


class FmmError(ValueError):
    """FmmError (bbfmm.rs:20-45); the PyO3 binding raises ValueError with these messages."""


class PointOutsideTree(FmmError):
    def __init__(self, point_index: int, leaf: bool = False):
        what = "FMM leaf evaluation failed" if leaf else "FMM evaluation failed"
        super().__init__(f"{what}: target point at row {point_index} lies outside the tree extents")
        self.point_index = point_index


class KernelDoesNotSupportGradients(FmmError):
    def __init__(self):
        super().__init__("FMM evaluation failed: gradient evaluation requested but kernel does "
                         "not support gradients")


def _as_f64_2d(a, name):
    """numpy_to_matref (python_bindings.rs:20-36): 1-D or 2-D float64."""
    arr = np.asarray(a)
    if arr.dtype != np.float64 or arr.ndim not in (1, 2):
        raise TypeError(f"Expected a 1D/2D float64 array for {name}")
    if arr.ndim == 1:
        arr = arr[:, None]
    return np.asfortranarray(arr)


class FmmTree:
    """ferreus_rbf_utils::FmmTree (utils.rs:383-494) / py FmmTree (python_bindings.rs:191-390)."""

    def __init__(self, source_points, interpolation_order: int, kernel_params: KernelParams,
                 adaptive_tree: bool, sparse: bool, *, extents=None,
                 params: Optional[FmmParams] = None, host_only: bool = False,
                 m2l_shared_basis: bool = False, direct_small_w_leaves: bool = False, deterministic: bool = False,
                 devices=None):
        """m2l_shared_basis: BBFMM_FLAG_M2L_SHARED_BASIS, an extension beyond the reference (off by default):
        the M2L stages run in one orthonormal basis per level, cut at params.epsilon.
        direct_small_w_leaves: BBFMM_FLAG_DIRECT_SMALL_W_LEAVES, likewise an extension: W-list leaves with no
        more points than nodes are summed directly instead of through M2P / P2L.
        deterministic: BBFMM_FLAG_DETERMINISTIC -- fixed summation order everywhere (no f64 atomics), bitwise
        reproducible results from run to run like the reference's.
        devices: HIP device ids of a handle that spans several devices of this process (bbfmm_create_on_devices): the
        matvec entry points run partitioned over them, nothing else changes for the caller.  An id may repeat (logical
        parts on one device).  None: the current device -- or the list in FERREUS_BBFMM_DEVICES, which is how the
        unchanged Rust caller is switched."""
        lib = L.load()
        pts = _as_f64_2d(source_points, "source_points")
        n, d = pts.shape
        ext = None
        if extents is not None:
            ext = np.ascontiguousarray(np.asarray(extents, dtype=np.float64).reshape(-1))
            d = len(ext) // 2                       # bbfmm.rs:291
        cpar = params._c() if params is not None else None
        h = ctypes.c_void_p()
        flags = ((L.FLAG_HOST_ONLY if host_only else 0) |
                 (L.FLAG_M2L_SHARED_BASIS if m2l_shared_basis else 0) |
                 (L.FLAG_DIRECT_SMALL_W_LEAVES if direct_small_w_leaves else 0) |
                 (L.FLAG_DETERMINISTIC if deterministic else 0))
        args = (pts.ctypes.data, n, d, n, int(interpolation_order),
                int(kernel_params.kernel_type), kernel_params.base_range,
                kernel_params.total_sill, int(bool(adaptive_tree)), int(bool(sparse)),
                ext.ctypes.data if ext is not None else None,
                ctypes.byref(cpar) if cpar is not None else None, flags)
        if devices is None:
            rc = lib.bbfmm_create(*args, ctypes.byref(h))
        else:
            dev = np.ascontiguousarray(np.asarray(devices, dtype=np.int32).reshape(-1))
            rc = lib.bbfmm_create_on_devices(*args, dev.ctypes.data, len(dev), ctypes.byref(h))
        self._h = h
        self._lib = lib
        if rc != L.OK:
            msg = lib.bbfmm_last_error(h).decode() if h else "bbfmm_create failed"
            if h:
                lib.bbfmm_destroy(h)
                self._h = None
            if rc == L.DEVICE_ERROR:
                raise RuntimeError(msg)
            raise ValueError(msg)              # the reference panics
        self.interpolation_order = int(interpolation_order)
        self.kernel_params = kernel_params
        self.adaptive_tree = bool(adaptive_tree)
        self.sparse = bool(sparse)
        self.n_points = n
        self.dim = d
        self.deterministic = bool(deterministic)
        self._nrhs = 0
        self._compressed = params is None or params.compression_type != M2LCompressionType.None_

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.bbfmm_destroy(h)
            self._h = None

    # -- errors
    def _raise(self, rc, bad=None, leaf=False):
        if rc == L.OK:
            return
        msg = self._lib.bbfmm_last_error(self._h).decode()
        if rc == L.POINT_OUTSIDE_TREE:
            raise PointOutsideTree(int(bad.value), leaf)
        if rc == L.KERNEL_NO_GRADIENTS:
            raise KernelDoesNotSupportGradients()
        if rc == L.DEVICE_ERROR:
            raise RuntimeError(msg)
        raise ValueError(msg)

    # -- API
    def set_weights(self, weights) -> None:
        w = _as_f64_2d(weights, "weights")
        self._raise(self._lib.bbfmm_set_weights(self._h, w.ctypes.data, w.shape[0], w.shape[1],
                                                w.shape[0]))
        self._nrhs = w.shape[1]

    def set_local_coefficients(self, weights) -> None:
        w = _as_f64_2d(weights, "weights")
        self._raise(self._lib.bbfmm_set_local_coefficients(self._h, w.ctypes.data, w.shape[0],
                                                           w.shape[1], w.shape[0]))
        self._nrhs = w.shape[1]           # (the column count may also have been fixed by a matvec on the C side)

    def _targets(self, target_points):
        """m x d column-major targets; the C ABI takes a pointer and reads d columns, so the column count is checked here
        (the reference's faer views panic on a missing column; found by the ASan run of scripts/sanitize_host.sh)."""
        x = _as_f64_2d(target_points, "target_points")
        if x.shape[1] != self.dim:
            raise ValueError(f"target_points must have {self.dim} columns, got {x.shape[1]}")
        return x

    def _eval(self, fn, weights, target_points, grads, leaf):
        x = self._targets(target_points)
        if weights is None and leaf:
            # leaves-only calls may reuse the weights resident on the device (header: bbfmm_evaluate_leaves)
            wp, rows, k = None, self.n_points, self._nrhs
        else:
            w = _as_f64_2d(weights, "weights")
            wp, rows, k = w.ctypes.data, w.shape[0], w.shape[1]
        m = x.shape[0]
        out = np.zeros((m, k), order="F")
        bad = ctypes.c_int64(-1)
        if grads:
            g = np.zeros((m, k * self.dim), order="F")
            rc = fn(self._h, wp, rows, k, rows, x.ctypes.data, m, max(m, 1),
                    out.ctypes.data, max(m, 1), g.ctypes.data, max(m, 1), ctypes.byref(bad))
            self._raise(rc, bad, leaf)
            return out, g
        rc = fn(self._h, wp, rows, k, rows, x.ctypes.data, m, max(m, 1),
                out.ctypes.data, max(m, 1), ctypes.byref(bad))
        self._raise(rc, bad, leaf)
        return out

    def evaluate(self, weights, target_points):
        return self._eval(self._lib.bbfmm_evaluate, weights, target_points, False, False)

    def evaluate_with_gradients(self, weights, target_points):
        return self._eval(self._lib.bbfmm_evaluate_with_gradients, weights, target_points, True, False)

    def evaluate_leaves(self, weights, target_points):
        return self._eval(self._lib.bbfmm_evaluate_leaves, weights, target_points, False, True)

    def evaluate_leaves_with_gradients(self, weights, target_points):
        return self._eval(self._lib.bbfmm_evaluate_leaves_with_gradients, weights, target_points,
                          True, True)

    def source_points(self):
        out = np.zeros((self.n_points, self.dim), order="F")
        self._raise(self._lib.bbfmm_source_points(self._h, out.ctypes.data, self.n_points))
        return out

    # -- the FGMRES matvec (ferreus_rbf/src/rbf.rs:1338-1379)
    def fast_matrix_vector_product(self, weights, basis_size=0, target_indices=None,
                                   polynomial_matrix=None, nugget=0.0):
        w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64).reshape(-1))
        res = np.zeros_like(w)
        idx = None
        if target_indices is not None:
            idx = np.ascontiguousarray(np.asarray(target_indices, dtype=np.int64))
        poly = None
        if polynomial_matrix is not None:
            poly = np.asfortranarray(np.asarray(polynomial_matrix, dtype=np.float64))
        rc = self._lib.bbfmm_fast_matrix_vector_product(
            self._h, w.ctypes.data, len(w), int(basis_size),
            idx.ctypes.data if idx is not None else None, len(idx) if idx is not None else 0,
            poly.ctypes.data if poly is not None else None,
            poly.shape[0] if poly is not None else 0, float(nugget), res.ctypes.data)
        self._raise(rc)
        self._nrhs = 1
        return res

    def prepare_target_subset(self, target_indices) -> None:
        """bbfmm_prepare_target_subset: build and cache the plan of a later partial product ahead of time."""
        idx = np.ascontiguousarray(np.asarray(target_indices, dtype=np.int64))
        self._raise(self._lib.bbfmm_prepare_target_subset(self._h, idx.ctypes.data, len(idx)))

    # -- device-resident matvec (what bench.py times); d_w / d_out are device pointers
    def matvec_device(self, d_w_ptr: int, ldw: int, k: int, d_out_ptr: int, ldo: int, sync=True):
        self._raise(self._lib.bbfmm_matvec_device(self._h, d_w_ptr, ldw, k, d_out_ptr, ldo,
                                                  int(sync)))
        self._nrhs = int(k)

    def stream(self) -> int:
        return int(self._lib.bbfmm_stream(self._h) or 0)

    # -- one handle over several devices (bbfmm_create_on_devices / FERREUS_BBFMM_DEVICES)
    def device_count(self) -> int:
        """Parts of the handle (1: an ordinary handle on one device)."""
        return int(self._lib.bbfmm_device_count(self._h))

    def part_device(self, part: int = 0) -> int:
        """HIP device of a part (part 0: the handle's own device, where device-resident vectors live)."""
        return int(self._lib.bbfmm_part_device(self._h, int(part)))

    def device(self) -> int:
        return self.part_device(0)

    def group_bounds(self) -> np.ndarray:
        """parts + 1 offsets into the tree's sorted points: part g owns [bounds[g], bounds[g + 1])."""
        b = np.zeros(self.device_count() + 1, dtype=np.int64)
        self._raise(self._lib.bbfmm_group_bounds(self._h, b.ctypes.data))
        return b

    def part_phase_ms(self, part: int):
        ms = np.zeros(L.N_PHASES)
        cnt = np.zeros(L.N_PHASES, dtype=np.int64)
        self._raise(self._lib.bbfmm_get_part_phase_ms(self._h, int(part), ms.ctypes.data, cnt.ctypes.data))
        return dict(zip(L.PHASE_NAMES, ms.tolist())), dict(zip(L.PHASE_NAMES, cnt.tolist()))

    def set_partition(self, rank: int, world: int):
        self._raise(self._lib.bbfmm_set_partition(self._h, rank, world))

    def partition_rows(self) -> np.ndarray:
        n = self._lib.bbfmm_partition_row_count(self._h)
        rows = np.zeros(n, dtype=np.int64)
        self._raise(self._lib.bbfmm_partition_rows(self._h, rows.ctypes.data))
        return rows

    def partition_coarse_count(self) -> int:
        """Doubles per right-hand side of the coarse multipoles a partition exchanges (0: none)."""
        return int(self._lib.bbfmm_partition_coarse_count(self._h))

    def matvec_partition_upward(self, d_w: int, ldw: int, k: int, d_coarse: int, comm_stream: int = 0):
        """First half of the partitioned matvec (device pointers): this rank's share of the upward pass; packs
        k x partition_coarse_count() partial coarse multipoles into d_coarse for the all-reduce, then queues the near
        field.  comm_stream: the HIP stream (pointer) the all-reduce will be issued on (0: ordered by the caller)."""
        self._raise(self._lib.bbfmm_matvec_partition_upward(self._h, d_w, ldw, k, d_coarse or None, comm_stream or None))

    def matvec_partition_finish(self, d_coarse: int, d_out: int, ldo: int, sync: bool = True, comm_stream: int = 0):
        """Second half: d_coarse summed over the ranks; downward + leaf pass of the owned targets."""
        self._raise(self._lib.bbfmm_matvec_partition_finish(self._h, d_coarse or None, d_out, ldo, int(sync), comm_stream or None))

    def partition_world(self) -> int:
        """Parts of the handle's partition (1: none set)."""
        return int(self._lib.bbfmm_partition_world(self._h))

    def partition_rank(self) -> int:
        """The part of its partition this handle owns."""
        return int(self._lib.bbfmm_partition_rank(self._h))

    def partition_bounds(self) -> np.ndarray:
        """world + 1 offsets into the tree's sorted points: part r owns bounds[r] .. bounds[r + 1) (the same on every
        rank of a `set_partition(rank, world)`)."""
        world = self.partition_world()
        b = np.zeros(world + 1, dtype=np.int64)
        self._raise(self._lib.bbfmm_partition_bounds(self._h, world, b.ctypes.data))
        return b

    def matvec_partition_finish_sorted(self, d_coarse: int, d_seg: int, ld: int, comm_stream: int = 0):
        """Second half with the owned potentials left in sorted order: k rows of d_seg (stride ld), the block a rank
        sends to the all-gather (asynchronous on the handle's stream)."""
        self._raise(self._lib.bbfmm_matvec_partition_finish_sorted(self._h, d_coarse or None, d_seg, ld, comm_stream or None))

    def partition_scatter(self, d_all: int, first_part: int, n_parts: int, m_max: int, k: int, d_out: int, ldo: int):
        """Gathered blocks d_all[n_parts][k][m_max] of parts first_part .. first_part + n_parts -> their rows of d_out."""
        self._raise(self._lib.bbfmm_partition_scatter(self._h, d_all, first_part, n_parts, m_max, k, d_out, ldo))

    def debug_partition_upward_counts(self):
        """(counts, reads, info): the rank's upward plan walked with point counts (see the header)."""
        c = self.stats().n_cells
        counts = np.zeros(c, dtype=np.int64)
        reads = np.zeros(c, dtype=np.uint8)
        info = np.zeros(4, dtype=np.int64)
        self._raise(self._lib.bbfmm_debug_partition_upward_counts(self._h, counts.ctypes.data, reads.ctypes.data,
                                                                 info.ctypes.data))
        return counts, reads, info

    def set_profiling(self, on: bool):
        self._lib.bbfmm_set_profiling(self._h, int(on))

    def phase_ms(self, reset=False, counts=False):
        """Accumulated per-phase device milliseconds (and interval counts) since the last reset."""
        ms = (ctypes.c_double * L.N_PHASES)()
        cnt = (ctypes.c_int64 * L.N_PHASES)()
        self._lib.bbfmm_get_phase_ms(self._h, ms, cnt)
        if reset:
            self._lib.bbfmm_reset_phase_ms(self._h)
        d = dict(zip(L.PHASE_NAMES, list(ms)))
        if counts:
            return d, dict(zip(L.PHASE_NAMES, list(cnt)))
        return d

    # -- introspection
    def stats(self) -> L.TreeStats:
        s = L.TreeStats()
        self._raise(self._lib.bbfmm_get_tree_stats(self._h, ctypes.byref(s)))
        return s

    def last_evaluate_at_sources(self) -> bool:
        """True when the last evaluate() found its targets to be the source points (bit for bit, row for row) and
        ran the resident-target path of the matvec (include/ferreus_bbfmm_hip.h, bbfmm_last_evaluate_at_sources)."""
        return int(self._lib.bbfmm_last_evaluate_at_sources(self._h)) == 1

    def debug_rows_of_sources(self, target_points):
        """Source rows of targets that are rows of the sources (bit for bit), or None when one of them is no source point."""
        x = self._targets(target_points)
        rows = np.zeros(x.shape[0], dtype=np.int64)
        ok = self._lib.bbfmm_debug_rows_of_sources(self._h, x.ctypes.data, x.shape[0], max(x.shape[0], 1), rows.ctypes.data)
        return rows if ok else None

    def last_evaluate_path(self) -> int:
        """0: the general path; 1: targets = the sources (resident target set; partitioned over the parts of a device group);
        2: targets = rows of the sources (cached plan of the index set); 3: arbitrary targets sharded over the parts of a
        device group -- bbfmm_last_evaluate_at_sources"""
        return int(self._lib.bbfmm_last_evaluate_at_sources(self._h))

    def debug_targets_are_sources(self, target_points) -> bool:
        """The host-side comparison bbfmm_evaluate runs on m == N targets (bit for bit, row for row)."""
        x = self._targets(target_points)
        return bool(self._lib.bbfmm_debug_targets_are_sources(self._h, x.ctypes.data, x.shape[0], max(x.shape[0], 1)))

    def tree_built_on_device(self) -> bool:
        return bool(self._lib.bbfmm_tree_built_on_device(self._h))

    def cells(self):
        s = self.stats()
        keys = np.zeros(s.n_cells, dtype=np.uint64)
        leaf = np.zeros(s.n_cells, dtype=np.uint8)
        self._raise(self._lib.bbfmm_get_cells(self._h, keys.ctypes.data, leaf.ctypes.data))
        return keys, leaf

    def leaf_sources(self):
        s = self.stats()
        ptr = np.zeros(s.n_cells + 1, dtype=np.int64)
        idx = np.zeros(max(s.n_points, 1), dtype=np.int64)
        self._raise(self._lib.bbfmm_get_leaf_sources(self._h, ptr.ctypes.data, idx.ctypes.data))
        return ptr, idx[:ptr[-1]]

    def interaction_list(self, which: str):
        s = self.stats()
        n = ctypes.c_int64(0)
        self._raise(self._lib.bbfmm_get_list(self._h, which.encode()[0:1], None, None, ctypes.byref(n)))
        ptr = np.zeros(s.n_cells + 1, dtype=np.int64)
        idx = np.zeros(max(n.value, 1), dtype=np.int32)
        self._raise(self._lib.bbfmm_get_list(self._h, which.encode()[0:1], ptr.ctypes.data,
                                             idx.ctypes.data, ctypes.byref(n)))
        return ptr, idx[:n.value]

    def m2l_ranks(self):
        s = self.stats()
        nref = ctypes.c_int32(0)
        self._raise(self._lib.bbfmm_get_m2l_ranks(self._h, None, ctypes.byref(nref)))
        ranks = np.zeros((s.depth + 1, nref.value), dtype=np.int32)
        self._raise(self._lib.bbfmm_get_m2l_ranks(self._h, ranks.ctypes.data, ctypes.byref(nref)))
        return ranks

    def m2l_operator(self, level: int, ref: int):
        n = self.stats().n_nodes
        out = np.zeros((n, n), order="F")
        self._raise(self._lib.bbfmm_get_m2l_operator(self._h, level, ref, out.ctypes.data))
        return out

    def m2l_factors(self, level: int, ref: int):
        """(U [n x r], Vt [r x n] or None) of one reference operator."""
        n = self.stats().n_nodes
        r = int(self.m2l_ranks()[level, ref])
        u = np.zeros((n, r), order="F")
        compressed = r < n or self._compressed
        vt = np.zeros((r, n), order="F") if compressed else None
        self._raise(self._lib.bbfmm_get_m2l_factors(self._h, level, ref, u.ctypes.data,
                                                    vt.ctypes.data if vt is not None else None))
        return u, vt

    def permutation_tables(self):
        n = self.stats().n_nodes
        nperm = ctypes.c_int32(0)
        self._raise(self._lib.bbfmm_get_permutation_tables(self._h, ctypes.byref(nperm), None, None,
                                                           None, None))
        nvec = 7 ** self.dim
        perm = np.zeros((nperm.value, n), dtype=np.int32)
        inv = np.zeros((nperm.value, n), dtype=np.int32)
        pl = np.zeros(nvec, dtype=np.int32)
        rl = np.zeros(nvec, dtype=np.int32)
        self._raise(self._lib.bbfmm_get_permutation_tables(
            self._h, ctypes.byref(nperm), perm.ctypes.data, inv.ctypes.data, pl.ctypes.data,
            rl.ctypes.data))
        return perm, inv, pl, rl

    def points_to_leaves(self, x):
        x = self._targets(x)
        m = x.shape[0]
        cells = np.zeros(max(m, 1), dtype=np.int32)
        bad = ctypes.c_int64(-1)
        rc = self._lib.bbfmm_points_to_leaves(self._h, x.ctypes.data, m, max(m, 1),
                                              cells.ctypes.data, ctypes.byref(bad))
        self._raise(rc, bad)
        return cells[:m]

    def debug_dense_m2m(self, child_index: int):
        n = self.stats().n_nodes
        out = np.zeros((n, n), order="F")
        self._raise(self._lib.bbfmm_debug_dense_m2m(self._h, child_index, out.ctypes.data))
        return out

    def debug_m2l_variants(self):
        """(number of stage-1 boundary variants, source cells that use one)"""
        nv, nc = ctypes.c_int64(), ctypes.c_int64()
        self._raise(self._lib.bbfmm_debug_m2l_variants(self._h, ctypes.byref(nv), ctypes.byref(nc)))
        return nv.value, nc.value

    def debug_get_coefficients(self, which: str, k: int) -> np.ndarray:
        s = self.stats()
        out = np.zeros((k, s.n_cells, s.n_nodes))
        self._raise(self._lib.bbfmm_debug_get_coefficients(self._h, which.encode()[0:1], k,
                                                           out.ctypes.data))
        return out

    def debug_apply_m2l_tables_host(self, M: np.ndarray) -> np.ndarray:
        M = np.ascontiguousarray(M, dtype=np.float64)
        Lc = np.zeros_like(M)
        self._raise(self._lib.bbfmm_debug_apply_m2l_tables_host(self._h, M.ctypes.data,
                                                                Lc.ctypes.data))
        return Lc


def fp64_valu_selftest():
    """(chip-wide v_fma_f64 TFLOP/s, shader clock in MHz during the run) on the current device."""
    tf, mhz = ctypes.c_double(0), ctypes.c_double(0)
    if L.load().bbfmm_fp64_valu_selftest(ctypes.byref(tf), ctypes.byref(mhz)) != L.OK:
        raise RuntimeError("the FP64 VALU microbenchmark needs a HIP device")
    return tf.value, mhz.value


def mfma_f64_selftest():
    """(measured FP64 MFMA TFLOP/s, lane-layout mismatches) on the current device."""
    tf = ctypes.c_double(0)
    errs = ctypes.c_int32(-1)
    info = (ctypes.c_double * 6)()
    rc = L.load().bbfmm_mfma_f64_selftest(ctypes.byref(tf), ctypes.byref(errs), info)
    if rc != L.OK:
        raise RuntimeError("MFMA self-test needs a HIP device")
    mfma_f64_selftest.info = dict(zip(["cycles_per_mfma_lone_wave", "clock_mhz_lone_wave",
                                       "cycles_per_mfma_per_simd_busy", "clock_mhz_busy",
                                       "tflops_1wave_per_simd", "tflops_2waves_per_simd"], list(info)))
    return tf.value, errs.value
