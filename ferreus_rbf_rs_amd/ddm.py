"""Domain decomposition hierarchy of ferreus_rbf's Schwarz preconditioner (host part):
`DDMTree` mirrors ferreus_rbf::preconditioning::domain_decomposition::DDMTree -- levels from finest to
coarsest, each with its leaf domains (overlapping point indices, internal mask, extents).  Built by
libferreus_bbfmm_hip.so (`bbfmm_ddm_build`); the local factorisations and the apply are not part of
this package yet."""
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
