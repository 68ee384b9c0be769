"""Domain decomposition hierarchy of ferreus_rbf's Schwarz preconditioner (host part):
`DDMTree` mirrors ferreus_rbf::preconditioning::domain_decomposition::DDMTree -- levels from finest to
coarsest, each with its leaf domains (overlapping point indices, internal mask, extents).  Built by
libferreus_bbfmm_hip.so (`bbfmm_ddm_build`).  `SchwarzPreconditioner` = decomposition + local factorisations
on the device + the level sweep of schwarz.rs as an FGMRES callback (`bbfmm_schwarz_*`)."""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np

from . import _lib as L


class DDMParams:
    """config.rs:42-69"""

    def __init__(self, leaf_threshold: int = 1024, overlap_quota: float = 0.5, coarse_ratio: float = 0.125,
                 coarse_threshold: int = 4096):
        self.leaf_threshold = int(leaf_threshold)
        self.overlap_quota = float(overlap_quota)
        self.coarse_ratio = float(coarse_ratio)
        self.coarse_threshold = int(coarse_threshold)

    def _c(self) -> L.DdmParams:
        return L.DdmParams(self.leaf_threshold, self.overlap_quota, self.coarse_ratio, self.coarse_threshold)

    @staticmethod
    def for_points(n: int) -> "DDMParams":
        """Extension (bbfmm_ddm_params_for_points): the defaults with coarse_threshold raised so that at most
        three fine levels are built over n points."""
        c = L.DdmParams()
        L.load().bbfmm_ddm_params_for_points(int(n), ctypes.byref(c))
        return DDMParams(c.leaf_threshold, c.overlap_quota, c.coarse_ratio, c.coarse_threshold)


class Domain:
    """domain.rs:86-117 (bookkeeping part)"""

    def __init__(self, overlapping_point_indices, internal_points_mask, extents):
        self.overlapping_point_indices = overlapping_point_indices
        self.internal_points_mask = internal_points_mask
        self.extents = extents


class Level:
    """domain_decomposition.rs:39-57"""

    def __init__(self, point_indices, leaf_domains: List[Domain]):
        self.point_indices = point_indices
        self.leaf_domains = leaf_domains


class DDMTree:
    """DDMTree::new (domain_decomposition.rs:67-347), global trend None."""

    def __init__(self, points, ddm_params: Optional[DDMParams] = None):
        pts = np.asfortranarray(np.atleast_2d(np.asarray(points, dtype=np.float64)))
        n, d = pts.shape
        lib = L.load()
        h = ctypes.c_void_p()
        prm = (ddm_params or DDMParams())._c()
        rc = lib.bbfmm_ddm_build(pts.ctypes.data, n, d, n, ctypes.byref(prm), ctypes.byref(h))
        if rc != L.OK:
            raise ValueError(f"bbfmm_ddm_build failed with status {rc}")
        try:
            self.levels: List[Level] = []
            for lv in range(lib.bbfmm_ddm_num_levels(h)):
                pi = np.zeros(lib.bbfmm_ddm_level_size(h, lv), dtype=np.int64)
                lib.bbfmm_ddm_level_points(h, lv, pi.ctypes.data)
                doms = []
                for dm in range(lib.bbfmm_ddm_num_domains(h, lv)):
                    k = lib.bbfmm_ddm_domain_size(h, lv, dm)
                    idx = np.zeros(k, dtype=np.int64)
                    mask = np.zeros(k, dtype=np.uint8)
                    ext = np.zeros(2 * d)
                    lib.bbfmm_ddm_domain(h, lv, dm, idx.ctypes.data, mask.ctypes.data, ext.ctypes.data)
                    doms.append(Domain(idx, mask.astype(bool), ext))
                self.levels.append(Level(pi, doms))
        finally:
            lib.bbfmm_ddm_destroy(h)


class InterpolantSettings:
    """InterpolantSettings (interpolant_config.rs:118-190) as the solver reads it.  kernel_type: a
    KernelType of the RBF family (Linear, ThinPlateSpline, Cubic, Spheroidal3/5/7/9); drift: None ->
    the kernel's minimum (get_min_drift), else -1 none / 0 constant / 1 linear / 2 quadratic."""

    _MIN = {0: 0, 1: 1, 2: 1}

    def __init__(self, kernel_type, dimensions: int, drift: Optional[int] = None, nugget: float = 0.0,
                 base_range: float = 1.0, total_sill: float = 1.0):
        self.kernel_type = int(kernel_type)
        if not 0 <= self.kernel_type <= 6:
            raise ValueError("the solver supports the RBF kernels only (Linear, ThinPlateSpline, Cubic, Spheroidal)")
        mn = self._MIN.get(self.kernel_type, -1)
        self.polynomial_degree = mn if drift is None else int(drift)
        if self.polynomial_degree < mn:
            raise ValueError(f"Min degree for kernel: {mn}")              # interpolant_config.rs:173
        k = self.polynomial_degree + 1
        self.basis_size = 0 if k <= 0 else {1: k, 2: k * (k + 1) // 2, 3: k * (k + 1) * (k + 2) // 6}[dimensions]
        self.nugget, self.base_range, self.total_sill = float(nugget), float(base_range), float(total_sill)


class SchwarzPreconditioner:
    """schwarz_preconditioner (preconditioning/schwarz.rs:32-155) with the local solves batched on
    the device.  `tree`: an FmmTree over the same points / kernel (serves matvec_partial).  Use it
    as the `m` of solvers.fgmres, or call it on a residual of N + basis_size values."""

    def __init__(self, tree, points, settings: InterpolantSettings, ddm_params: Optional[DDMParams] = None,
                 global_scaling: bool = False, shard_group=None):
        """global_scaling: scale the coarse domain's monomials by the extents of all points (the basis of the
        system's monomial matrix) instead of by its own extents as the reference does (domain.rs:171-172);
        see csrc/schwarz.cpp.
        shard_group: a torch.distributed process group (True: the default group): the factors are sharded over its ranks
        (bbfmm_schwarz_create_sharded: every rank factorises and solves a contiguous share of each fine level's domains,
        a level's corrections are summed over the ranks -- RCCL all-reduce on the exchange tensor, staged through the host
        when the group's backend is gloo); every rank must create it and call it in step."""
        pts = np.asfortranarray(np.atleast_2d(np.asarray(points, dtype=np.float64)))
        n, d = pts.shape
        lib = L.load()
        st = L.Interpolant(settings.kernel_type, settings.polynomial_degree, settings.nugget, settings.base_range,
                           settings.total_sill, L.FLAG_GLOBAL_SCALING if global_scaling else 0)
        prm = (ddm_params or DDMParams())._c()
        h = ctypes.c_void_p()
        self._xchg = self._allreduce_cb = None
        self.rank, self.world = 0, 1
        if shard_group is not None:
            import torch
            import torch.distributed as dist
            group = None if shard_group is True else shard_group
            self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
            # The ranks run the replicated parts of the sweep (partial products, coarse solve) and the solver around it in
            # lock step: a default tree accumulates with f64 atomics, ranks would differ in their last bits, an FGMRES run
            # near its tolerance could end on one rank while the others wait in the next all-reduce.
            if self.world > 1 and not getattr(tree, "deterministic", False):
                raise ValueError("a sharded preconditioner needs a tree created with deterministic=True (BBFMM_FLAG_DETERMINISTIC): "
                                 "the ranks must take the same branches")
            dev = torch.device("cuda", tree.device())              # the handle's own device (bbfmm_part_device), not torch's current one
            self._xchg = torch.zeros(n, dtype=torch.float64, device=dev)
            staged = dist.get_backend(group) == "gloo"
            host = torch.zeros(n, dtype=torch.float64).pin_memory() if staged else None
            xchg = self._xchg

            def _allreduce(_user, count):      # bbfmm_allreduce_fn: sum xchg[:count] over the ranks, result visible on return
                try:
                    if staged:
                        host[:count].copy_(xchg[:count])
                        dist.all_reduce(host[:count], group=group)
                        xchg[:count].copy_(host[:count])
                    else:
                        dist.all_reduce(xchg[:count], group=group)
                    torch.cuda.synchronize(dev)
                    return 0
                except BaseException:  # noqa: BLE001 -- must not unwind through C
                    return 1

            self._allreduce_cb = L.ALLREDUCE_FN(_allreduce)
        if self.world > 1:
            torch.cuda.synchronize(dev)
            rc = lib.bbfmm_schwarz_create_sharded(tree._h, pts.ctypes.data, n, d, n, ctypes.byref(st), ctypes.byref(prm),
                                                  self.rank, self.world, self._xchg.data_ptr(), n,
                                                  ctypes.cast(self._allreduce_cb, ctypes.c_void_p), None, ctypes.byref(h))
        else:
            rc = lib.bbfmm_schwarz_create(tree._h, pts.ctypes.data, n, d, n, ctypes.byref(st), ctypes.byref(prm),
                                          ctypes.byref(h))
        if self.world > 1:
            # every rank raises, or none does: a rank that failed alone (a local system that is not positive definite, out of
            # memory) would leave its peers waiting in the first level's all-reduce
            ok = torch.tensor([1 if rc == L.OK else 0], dtype=torch.int64, device="cpu" if staged else dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                if rc == L.OK and h:
                    lib.bbfmm_schwarz_destroy(h)
                raise ValueError(f"bbfmm_schwarz_create_sharded failed on a rank of the group (this rank: status {rc})"
                                 + (" (a local system is not positive definite)" if rc == L.UNSUPPORTED else ""))
        if rc != L.OK:
            raise ValueError(f"bbfmm_schwarz_create failed with status {rc}"
                             + (" (a local system is not positive definite)" if rc == L.UNSUPPORTED else ""))
        self._h, self._lib, self._tree = h, lib, tree
        self.n = n
        self.basis_size = int(lib.bbfmm_schwarz_basis_size(h))
        self.num_levels = int(lib.bbfmm_schwarz_num_levels(h))
        mp = lib.bbfmm_schwarz_monomial_matrix(h)
        self.monomial_matrix = None
        if mp:
            buf = (ctypes.c_double * (n * self.basis_size)).from_address(mp)
            self.monomial_matrix = np.frombuffer(buf, dtype=np.float64).reshape((n, self.basis_size), order="F").copy(order="F")
        from .solvers import _Operator
        # (the operator must not keep `self` alive: a cycle would hold the factors -- 95 GB of HBM at 10M points -- until the
        # cyclic collector happens to run; whoever passes the preconditioner to a solver holds it for the call anyway)
        self._op = _Operator(ctypes.cast(lib.bbfmm_schwarz_apply, ctypes.c_void_p), h, ())
        self._op.errors = []

    def __call__(self, residual):
        r = np.ascontiguousarray(residual, dtype=np.float64).reshape(-1)
        z = np.zeros_like(r)
        rc = self._lib.bbfmm_schwarz_apply(self._h, r.ctypes.data, z.ctypes.data, r.size)
        if rc != L.OK:
            raise RuntimeError(f"bbfmm_schwarz_apply failed with status {rc}")
        return z

    def factor_bytes(self) -> int:
        """device bytes of the packed factors this handle (this rank) holds"""
        return int(self._lib.bbfmm_schwarz_factor_bytes(self._h))

    def domains_owned(self, level: int):
        """(domains of the level factorised here, the first of them, the level's total)"""
        first, total = ctypes.c_int64(0), ctypes.c_int64(0)
        own = self._lib.bbfmm_schwarz_domains_owned(self._h, level, ctypes.byref(first), ctypes.byref(total))
        return int(own), int(first.value), int(total.value)

    def level_points(self, level: int) -> np.ndarray:
        """Level::point_indices of one level (finest first)"""
        out = np.zeros(self._lib.bbfmm_schwarz_level_size(self._h, level), dtype=np.int64)
        self._lib.bbfmm_schwarz_level_points(self._h, level, out.ctypes.data)
        return out

    def debug_level_solve(self, level: int, residual, add_poly: bool = True) -> np.ndarray:
        """solve_fine_level / solve_coarse_level (schwarz.rs:84-155) of one level"""
        r = np.ascontiguousarray(residual, dtype=np.float64).reshape(-1)
        z = np.zeros_like(r)
        rc = self._lib.bbfmm_schwarz_debug_level_solve(self._h, level, r.ctypes.data, z.ctypes.data, r.size, int(add_poly))
        if rc != L.OK:
            raise RuntimeError(f"bbfmm_schwarz_debug_level_solve failed with status {rc}")
        return z

    def close(self) -> None:
        """Release the factors and every device buffer now (also done when the last reference goes)."""
        h = getattr(self, "_h", None)
        if h:
            self._lib.bbfmm_schwarz_destroy(h)
            self._h = None
            self._op.user_ptr = None

    def __del__(self):
        self.close()
