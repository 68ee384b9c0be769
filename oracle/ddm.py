"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy/scipy) of the reference's Schwarz / domain
decomposition preconditioner: ferreus_rbf/src/preconditioning/
{domain_decomposition,schwarz}.rs, domain.rs, polynomials.rs and the helpers they call.  Only tests/
may import this; the product never does.

Parity unpinned: the reference cannot be built or imported here.  What the reference's own tests
assert for these files (structural invariants of the hierarchy, domain_decomposition.rs:378-596;
domain solve == naive solve, domain.rs:732-763) is asserted for this restatement in
tests/test_oracle_ddm.py, and the preconditioned FGMRES is checked against dense solves.

Not restated: the global trend transform (global_trend.rs; `None` in every call here) and the
rectangular-full-packed storage of the Cholesky factors (linalg.rs:37-470: a storage format, the
factor is the same).  Third-party pieces restated by their published meaning: faer's
`col_piv_qr` (column-pivoted Householder QR, LAPACK dgeqp3 rule: largest remaining column norm,
here scipy.linalg.qr(pivoting=True)), `full_piv_lu` / `partial_piv_lu` solves (numpy.linalg.solve),
rstar's `locate_in_envelope_intersecting` (all stored boxes that intersect the closed query box;
the iteration order, which only breaks exact distance ties, is taken as ascending index).
"""
from __future__ import annotations

import math
from collections import deque

import numpy as np
import scipy.linalg as sla

from . import bbfmm_oracle as O


class DDMParams:
    """config.rs:42-69"""

    def __init__(self, leaf_threshold=1024, overlap_quota=0.5, coarse_ratio=0.125, coarse_threshold=4096):
        self.leaf_threshold = leaf_threshold
        self.overlap_quota = overlap_quota
        self.coarse_ratio = coarse_ratio
        self.coarse_threshold = coarse_threshold


class InterpolantSettings:
    """interpolant_config.rs:118-190 reduced to what the preconditioner reads.  drift: -1 none,
    0 constant, 1 linear, 2 quadratic (polynomial_degree); basis_size by set_basis_size."""

    MIN_DEGREE = {0: 0, 1: 1, 2: 1}          # Linear, ThinPlateSpline, Cubic; spheroidal: -1

    def __init__(self, kernel_id, dimensions, drift=None, nugget=0.0, base_range=1.0, total_sill=1.0):
        self.kernel_id = kernel_id
        self.nugget = nugget
        self.base_range = base_range
        self.total_sill = total_sill
        min_degree = self.MIN_DEGREE.get(kernel_id, -1)
        degree = min_degree if drift is None else drift                     # get_min_drift, :38-45
        if degree < min_degree:
            raise ValueError(f"Min degree for kernel: {min_degree}")        # :173
        k = degree + 1
        if degree < 0:
            self.basis_size = 0
        else:
            self.basis_size = {1: k, 2: k * (k + 1) // 2, 3: k * (k + 1) * (k + 2) // 6}[dimensions]
        self.polynomial_degree = degree


def pointarray_extents(p):
    """ferreus_rbf_utils/src/utils.rs:196-228: [mins..., maxs...]"""
    return np.concatenate([p.min(axis=0), p.max(axis=0)])


def cheb_cube_scaling_factors(p):
    """common.rs:299-322"""
    e = pointarray_extents(p)
    d = p.shape[1]
    tr = (e[d:] + e[:d]) / 2.0
    sc = (e[d:] - e[:d]) / 2.0
    sc[sc == 0.0] = 1.0
    return tr, sc


def evaluate_monomials(points, degree, basis_size, translation, scale):
    """polynomials.rs:30-74: 1 | x_i | x_i x_j (i <= j), on (x - translation) / scale"""
    s = (points - translation) / scale
    n, d = s.shape
    m = np.zeros((n, basis_size))
    m[:, 0] = 1.0
    if degree >= 1:
        m[:, 1:1 + d] = s
    if degree == 2:
        k = 1 + d
        for i in range(d):
            for j in range(i, d):
                m[:, k] = s[:, i] * s[:, j]
                k += 1
    return m


def a_matrix(points, st):
    """get_a_matrix_symmetric_solver (ferreus_rbf_utils/src/utils.rs:316-349): kernel matrix + nugget I"""
    n = points.shape[0]
    a = np.array(O.kernel_matrix(st.kernel_id, st.base_range, st.total_sill, points, points))
    a[np.diag_indices(n)] += st.nugget
    return a


def farthest_point_sampling(points, num_wanted, seed_index):
    """common.rs:246-288"""
    n = points.shape[0]
    selected = [int(seed_index)]
    is_sel = np.zeros(n, bool)
    is_sel[seed_index] = True
    min_d = np.full(n, np.inf)
    for _ in range(1, num_wanted):
        last = selected[-1]
        dist = np.sqrt(((points - points[last]) ** 2).sum(axis=1))
        upd = (~is_sel) & (dist < min_d)
        min_d[upd] = dist[upd]
        cand = np.where(is_sel, -1.0, min_d)          # first index with the largest min distance
        far = int(np.argmax(cand)) if (cand > -1.0).any() else 0
        selected.append(far)
        is_sel[far] = True
    return selected


class Domain:
    """domain.rs:86-475"""

    def __init__(self, indices):
        self.overlapping_point_indices = [int(i) for i in indices]
        self.internal_points_mask = []
        self.extents = None
        self.solve_for_poly = False
        self.chol = None
        self.q_top = None
        self.a_special_rows = None
        self.special_monomials = None
        self.n_special = 0

    def internal_indices(self):
        return [g for g, m in zip(self.overlapping_point_indices, self.internal_points_mask) if m]

    def factorise(self, points, st, solve_for_poly):
        idx = np.asarray(self.overlapping_point_indices)
        dom = points[idx]
        if st.basis_size != 0:
            tr, sc = cheb_cube_scaling_factors(dom)
            mono = evaluate_monomials(dom, st.polynomial_degree, st.basis_size, tr, sc)
            _, rc, piv = sla.qr(mono, mode="economic", pivoting=True)                     # :186-189
            diag = np.abs(np.diag(rc))
            rank = int((diag > 1e-10 * diag[0]).sum())                                   # :192-201
            cols = sorted(int(c) for c in piv[:rank])
            full = mono[:, cols]
            _, _, pivr = sla.qr(full.T, mode="economic", pivoting=True)                   # :217-221
            special = sorted(int(c) for c in pivr[:rank])
            sset = set(special)
            non_special = [i for i in range(len(idx)) if i not in sset]
            order = special + non_special                                                # :250-279
            mask = list(self.internal_points_mask) + [False] * (len(idx) - len(self.internal_points_mask))
            self.overlapping_point_indices = [int(idx[i]) for i in order]
            self.internal_points_mask = [bool(mask[i]) for i in order]
            self.n_special = rank
            sp_m, ns_m = full[special], full[non_special]
            a = a_matrix(points[np.asarray(self.overlapping_point_indices)], st)
            lag = np.linalg.solve(sp_m, np.eye(rank))                                    # get_lagrange_coefficients
            q = -(ns_m @ lag).T                                                          # :303-307, rank x m
            a11, a12, a21, a22 = a[:rank, :rank], a[:rank, rank:], a[rank:, :rank], a[rank:, rank:]
            lhs = q.T @ (a11 @ q) + q.T @ a12 + a21 @ q + a22                            # :312-346
            self.q_top = q
            if solve_for_poly:
                self.solve_for_poly = True
                self.a_special_rows = a[:rank].copy()
                self.special_monomials = sp_m.copy()
        else:
            lhs = a_matrix(dom, st)
        sym = 0.5 * (lhs + lhs.T)
        try:
            self.chol = sla.cho_factor(sym, lower=True)                                  # LltRfp, linalg.rs
            self.indefinite = None
        except np.linalg.LinAlgError:
            # DomainSolver::new (domain.rs:60-68): a failed Cholesky switches the domain to the Bunch-Kaufman
            # LBL^T solver (linalg.rs:514-616), restated by its meaning: a stable symmetric indefinite solve
            self.chol = None
            self.indefinite = sym

    def solve(self, values):
        """domain.rs:393-475; values: global (n_total x k).  Returns (point coefficients in the
        domain's point order, polynomial coefficients or None)."""
        v = values.reshape(values.shape[0], -1)
        d = v[np.asarray(self.overlapping_point_indices)]
        ns = self.n_special
        if self.q_top is not None:
            rhs = self.q_top.T @ d[:ns] + d[ns:]
        else:
            rhs = d
        gamma = sla.cho_solve(self.chol, rhs) if self.chol is not None else np.linalg.solve(self.indefinite, rhs)
        if self.q_top is not None:
            coef = np.vstack([self.q_top @ gamma, gamma])
        else:
            coef = gamma
        poly = None
        if self.solve_for_poly:
            r = d[:ns] - self.a_special_rows @ coef
            poly = np.linalg.solve(self.special_monomials, r)
        return coef, poly


class Level:
    def __init__(self, point_indices):
        self.point_indices = list(point_indices)
        self.leaf_domains = []


def build_ddm_tree(points, st, params=None):
    """DDMTree::new (domain_decomposition.rs:67-347), global trend None.  Returns the list of levels,
    finest first, the single coarse domain last."""
    params = params or DDMParams()
    n, dims = points.shape
    levels = []
    active = list(range(n))
    while len(active) > params.coarse_threshold:
        root = Domain(active)
        root.internal_points_mask = [True] * len(active)
        root.extents = pointarray_extents(points[np.asarray(active)])
        queue = deque([root])
        level = Level(active)
        coarse_pts = []
        while queue:
            cur = queue.popleft()
            cidx = np.asarray(cur.overlapping_point_indices)
            cp = points[cidx]
            ext = pointarray_extents(cp)
            lengths = ext[dims:] - ext[:dims]
            axis = 0                                                    # argmax: first strictly greater than 0
            best = 0.0
            for a in range(dims):
                if lengths[a] > best:
                    best, axis = lengths[a], a
            order = np.argsort(cp[:, axis], kind="stable")              # argsort (stable sort_by)
            sorted_idx = cidx[order]
            mid = len(cidx) // 2
            left = sorted(int(i) for i in sorted_idx[:mid])
            right = sorted(int(i) for i in sorted_idx[mid:])
            mid_coord = points[int(sorted_idx[mid]), axis]
            ld, rd = Domain(left), Domain(right)
            ld.extents = cur.extents.copy()
            ld.extents[axis + dims] = mid_coord
            rd.extents = cur.extents.copy()
            rd.extents[axis] = mid_coord
            npts = len(cidx)
            if npts + npts * params.overlap_quota >= 2.0 * params.leaf_threshold:   # :150-153
                queue.extend([ld, rd])
            else:
                for dom in (ld, rd):
                    dom.internal_points_mask = [True] * len(dom.overlapping_point_indices)
                level.leaf_domains.extend([ld, rd])
        leaves = level.leaf_domains
        num_coarse = int(math.ceil(math.ceil(len(active) * params.coarse_ratio) / len(leaves)))   # :165-168
        boxes = np.array([dom.extents for dom in leaves])
        for i, dom in enumerate(leaves):
            internal = dom.internal_indices()
            ip = points[np.asarray(internal)]
            sample = min(len(internal), num_coarse)
            # get_centroid (domain_decomposition.rs:350-359) folds each column left to right; numpy's sum is pairwise on a
            # contiguous axis (1-D point sets), which moves the centroid by an ulp and with it the argmin among tied points
            center = np.add.accumulate(ip, axis=0)[-1] / ip.shape[0]
            dist = np.sqrt(((ip - center) ** 2).sum(axis=1))
            center_index = int(np.argmin(dist))
            sel = farthest_point_sampling(ip, sample, center_index)
            coarse_pts.extend(sorted(internal[j] for j in sel))
            # neighbours: leaf boxes that intersect this one (closed boxes), self excluded (rtree.rs:76-88)
            lo, hi = boxes[:, :dims], boxes[:, dims:]
            hit = np.all((lo <= dom.extents[dims:]) & (hi >= dom.extents[:dims]), axis=1)
            neigh = [j for j in np.nonzero(hit)[0] if j != i]
            num_overlap = int(math.ceil(len(dom.overlapping_point_indices) * 2 * params.overlap_quota))
            nidx = []
            for j in neigh:
                nidx.extend(leaves[j].internal_indices())
            if nidx:
                npnts = points[np.asarray(nidx)]
                clipped = np.maximum(np.minimum(npnts, dom.extents[dims:]), dom.extents[:dims])
                bd = np.sqrt(((npnts - clipped) ** 2).sum(axis=1))
                take = np.argsort(bd, kind="stable")[:min(num_overlap, len(nidx))]
                dom.overlapping_point_indices.extend(int(nidx[t]) for t in take)
            dom.internal_points_mask.extend([False] * num_overlap)                     # :300-306 (as written)
        for dom in leaves:
            dom.factorise(points, st, False)
        levels.append(level)
        active = sorted(coarse_pts)
    coarse = Level(active)
    cd = Domain(active)
    cd.internal_points_mask = [True] * len(active)
    cd.factorise(points, st, st.basis_size != 0)
    coarse.leaf_domains.append(cd)
    levels.append(coarse)
    return levels


def orthonormal_poly(points, st, translation, scale):
    """rbf.rs:476-495: thin Q of the global monomial matrix"""
    mono = evaluate_monomials(points, st.polynomial_degree, st.basis_size, translation, scale)
    q, _ = np.linalg.qr(mono)
    return mono, q


def schwarz_preconditioner(rg, levels, matvec_partial, st, ortho):
    """schwarz.rs:32-155.  rg: residual (N + basis_size); matvec_partial(weights, indices) returns the
    system product on the rows `indices` and zeros elsewhere (rbf.rs:119-133)."""
    rg = np.asarray(rg, dtype=np.float64).reshape(-1)
    sl = np.zeros_like(rg)
    coarse_idx = len(levels) - 1
    coarse_points = levels[coarse_idx].point_indices

    def solve_fine(res, i):
        s1 = np.zeros_like(res)
        for dom in levels[i].leaf_domains:
            coef, _ = dom.solve(res[:, None])
            for local, (g, m) in enumerate(zip(dom.overlapping_point_indices, dom.internal_points_mask)):
                if m:
                    s1[g] = coef[local, 0]
        if st.basis_size != 0:
            npnt = res.shape[0] - st.basis_size
            s1[:npnt] -= ortho @ (ortho.T @ s1[:npnt])
        return s1

    def solve_coarse(res, add_poly):
        sc = np.zeros_like(res)
        dom = levels[coarse_idx].leaf_domains[0]
        coef, poly = dom.solve(res[:, None])
        sc[np.asarray(dom.overlapping_point_indices)] = coef[:, 0]
        if dom.solve_for_poly and add_poly:
            sc[res.shape[0] - poly.shape[0]:] = poly[:, 0]
        return sc

    if coarse_idx > 0:
        for i in range(coarse_idx):
            sl = sl + solve_fine(rg - matvec_partial(sl, levels[i].point_indices), i)
            sl = sl + solve_coarse(rg - matvec_partial(sl, coarse_points), i == coarse_idx - 1)
    else:
        sl = sl + solve_coarse(rg - matvec_partial(sl, coarse_points), True)
    return sl
