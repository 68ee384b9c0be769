"""CPU restatement of the ferreus_bbfmm matvec (tree, lists, operators, passes).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED by
upstream tests; pinned against the dense direct sum and the fixtures in
``tests/golden``.

The host-side structure (Morton keys, hash sets/maps of cells, interaction
lists, Chebyshev tables, ACA) is restated in Python/numpy following the
reference line by line; the per-matvec arithmetic runs in ``oracle/passes.c``
(gcc + OpenMP, loaded with ctypes).  Where the reference iterates a Rust
``HashSet`` (random order) this restatement iterates in sorted key order, so
cell column numbering is deterministic (finding 5).

All citations are relative to ``/root/reference``.
"""
from __future__ import annotations

import ctypes
import itertools
import math
import os
import subprocess
from collections import deque
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "liboracle_passes.so")

# ----------------------------------------------------------------------------
# kernel ids: ferreus_rbf_utils/src/utils.rs:558-571 (order of the registry)
KERNEL_IDS = {
    "LinearRbf": 0,
    "ThinPlateSplineRbf": 1,
    "CubicRbf": 2,
    "Spheroidal3Rbf": 3,
    "Spheroidal5Rbf": 4,
    "Spheroidal7Rbf": 5,
    "Spheroidal9Rbf": 6,
    "Laplacian": 7,
    "OneOverR2": 8,
    "OneOverR4": 9,
    # extension kernels of this repo, NOT in the reference
    "GaussianExt": 100,
    "MultiquadricExt": 101,
}

COMPRESSION_NONE, COMPRESSION_SVD, COMPRESSION_ACA = 0, 1, 2


def build_passes(force: bool = False) -> str:
    """Compile oracle/passes.c -> oracle/_build/liboracle_passes.so (gcc, OpenMP)."""
    src = os.path.join(_HERE, "passes.c")
    if os.environ.get("ORACLE_PASSES_SANITIZE") == "1":
        # the checker checked: AddressSanitizer + UBSan build (scripts/sanitize_host.sh preloads gcc's libasan.so)
        so = os.path.join(_BUILD, "liboracle_passes_asan.so")
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(_BUILD, exist_ok=True)
            subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                                   "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared",
                                   "-std=c11", "-o", so, src, "-lm"])
        return so
    if (not force) and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(_BUILD, exist_ok=True)
    # -ffp-contract=off: the reference (rustc) never fuses a*b+c; keep plain IEEE arithmetic.
    # -fno-math-errno: sqrt() is the instruction, not a call that may set errno (values unchanged; lets loops vectorise).
    cmd = ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fno-math-errno", "-fopenmp", "-fPIC", "-shared",
           "-std=c11", "-o", _SO, src, "-lm"]
    try:
        subprocess.check_call(cmd)
    except subprocess.CalledProcessError:
        # -march=native can fail on exotic hosts; retry generic.
        cmd.remove("-march=native")
        subprocess.check_call(cmd)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build_passes())
        _lib.oracle_kernel_phi_r2.restype = ctypes.c_double
        _lib.oracle_kernel_phi_r2.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_double]
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
        _lib.oracle_set_num_threads.restype = None
    return _lib


def _p(a, dtype=None):
    if a is None:
        return ctypes.c_void_p(0)
    if dtype is not None:
        assert a.dtype == dtype, (a.dtype, dtype)
    assert a.flags["C_CONTIGUOUS"] or a.flags["F_CONTIGUOUS"]
    return ctypes.c_void_p(a.ctypes.data)


I64 = ctypes.c_int64
I32 = ctypes.c_int
F64 = ctypes.c_double

# ----------------------------------------------------------------------------
# Morton primitives -- ferreus_bbfmm/src/morton.rs, morton_constants.rs
MAXIMUM_LEVEL = 16          # morton_constants.rs:12
LEVEL_DISPLACEMENT = 15     # morton_constants.rs:15
LEVEL_MASK = 0x7FFF         # morton_constants.rs:18

# morton_constants.rs:32-74 (order matters only for list construction order)
DIRECTIONS = {
    1: [(-1,), (1,)],
    2: [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)],
    3: [(-1, -1, -1), (-1, -1, 0), (-1, -1, 1), (-1, 0, -1), (-1, 1, -1), (-1, 0, 0),
        (-1, 0, 1), (-1, 1, 0), (-1, 1, 1), (0, -1, -1), (1, -1, -1), (0, -1, 0),
        (0, -1, 1), (1, -1, 0), (1, -1, 1), (0, 0, -1), (0, 1, -1), (1, 0, -1),
        (1, 1, -1), (0, 0, 1), (0, 1, 0), (0, 1, 1), (1, 0, 0), (1, 0, 1), (1, 1, 0),
        (1, 1, 1)],
}


def get_side_length(radius: float, level: int) -> float:
    """morton.rs:29-32"""
    return 2.0 * radius / float(1 << level)


def _spread(v: int, d: int) -> int:
    """Bit-interleave the low 16 bits of v with stride d.  Equals the byte LUTs
    MORTON_ENCODE_*_LOOKUP (morton_constants.rs:77-287; checked entry by entry)."""
    r = 0
    for i in range(16):
        r |= ((v >> i) & 1) << (d * i)
    return r


def encode_morton_point(anchor, level: int, d: int) -> int:
    """morton.rs:58-119.  Only the low 16 bits of each coordinate take part
    (two byte lookups per axis)."""
    code = 0
    for a in range(d):
        code |= _spread(int(anchor[a]) & 0xFFFF, d) << a
    return (code << LEVEL_DISPLACEMENT) | level


def get_level(key: int) -> int:
    return key & LEVEL_MASK


def decode_key(key: int, d: int):
    """morton.rs:127-167 -> (anchor tuple, level)."""
    level = key & LEVEL_MASK
    k = key >> LEVEL_DISPLACEMENT
    anchor = [0] * d
    if d == 1:
        anchor[0] = k & 0xFFFF
    else:
        nbits = 21 if d == 3 else 28      # 7 loops x 3 (or 4) bits per axis
        for a in range(d):
            v = 0
            for i in range(nbits):
                v |= ((k >> (d * i + a)) & 1) << i
            anchor[a] = v
    return tuple(anchor), level


def get_parent(key: int, d: int):
    """morton.rs:170-190"""
    level = key & LEVEL_MASK
    if level == 0:
        return None
    return (((key >> LEVEL_DISPLACEMENT) >> d) << LEVEL_DISPLACEMENT) | (level - 1)


def get_ancestors(key: int, d: int):
    """morton.rs:193-210 (includes key itself)."""
    out = [key]
    cur = key
    while True:
        par = get_parent(cur, d)
        if par is None:
            break
        out.append(par)
        cur = par
    return out


def get_children(key: int, d: int):
    """morton.rs:266-297"""
    level = key & LEVEL_MASK
    root = (key >> LEVEL_DISPLACEMENT) << d
    return [((root | s) << LEVEL_DISPLACEMENT) | (level + 1) for s in range(1 << d)]


def get_child_index(key: int, d: int) -> int:
    """morton.rs:300-305"""
    return (key >> LEVEL_DISPLACEMENT) & ((1 << d) - 1)


def get_neighbours(key: int, d: int):
    """morton.rs:214-263"""
    anchor, level = decode_key(key, d)
    nmax = 1 << level
    out = []
    for direction in DIRECTIONS[d]:
        na = [anchor[a] + direction[a] for a in range(d)]
        if all(0 <= v < nmax for v in na):
            out.append(encode_morton_point(na, level, d))
    return out


_CL_CACHE: dict = {}


def get_center_length(key: int, center, radius: float, d: int):
    """morton.rs:328-346 (memoised: pure function of its arguments)."""
    ck = (key, tuple(center), radius, d)
    hit = _CL_CACHE.get(ck)
    if hit is not None:
        return hit
    if len(_CL_CACHE) > 4_000_000:
        _CL_CACHE.clear()
    res = _get_center_length(key, center, radius, d)
    _CL_CACHE[ck] = res
    return res


def _get_center_length(key: int, center, radius: float, d: int):
    anchor, level = decode_key(key, d)
    side = get_side_length(radius, level)
    c = tuple((float(anchor[a]) + 0.5) * side + (center[a] - radius) for a in range(d))
    return c, side


def are_adjacent(a: int, b: int, center, radius: float, d: int) -> bool:
    """morton.rs:308-325"""
    ca, la = get_center_length(a, center, radius, d)
    cb, lb = get_center_length(b, center, radius, d)
    length = 0.5 * (la + lb)
    return all(abs(vb - va) <= 1e-6 + length for va, vb in zip(ca, cb))


def calculate_tree_center_and_radius(extents):
    """morton.rs:349-373; extents = [mins..., maxs...]"""
    d = len(extents) // 2
    lo = [math.floor(v) for v in extents[:d]]
    hi = [math.ceil(v) for v in extents[d:]]
    center = [(l + u) / 2.0 for l, u in zip(lo, hi)]
    radius = -math.inf
    for l, u in zip(lo, hi):
        radius = max(radius, (u - l) / 2.0 + 1e-3)
    return center, radius


def get_pointarray_extents(points: np.ndarray):
    """ferreus_bbfmm/src/utils.rs:13-46"""
    return list(points.min(axis=0)) + list(points.max(axis=0))


def points_to_anchor_keys(points: np.ndarray, level: int, center, radius: float) -> np.ndarray:
    """Vectorised point_to_anchor + encode_morton_point (morton.rs:35-51, 58-119).
    Rust `as u64` saturates: negative / NaN -> 0, huge -> u64::MAX (then masked
    to 16 bits by the byte lookups)."""
    n, d = points.shape
    side = get_side_length(radius, level)
    keys = np.zeros(n, dtype=np.uint64)
    for a in range(d):
        disp = center[a] - radius
        f = np.floor((points[:, a] - disp) / side)
        f = np.where(np.isnan(f), 0.0, f)
        f = np.clip(f, 0.0, 1.8446744073709552e19)
        big = f >= 1.8446744073709552e19
        v = np.where(big, 0.0, f).astype(np.uint64)
        v = np.where(big, np.uint64(0xFFFFFFFFFFFFFFFF), v) & np.uint64(0xFFFF)
        s = np.zeros(n, dtype=np.uint64)
        for i in range(16):
            s |= ((v >> np.uint64(i)) & np.uint64(1)) << np.uint64(d * i)
        keys |= s << np.uint64(a)
    return (keys << np.uint64(LEVEL_DISPLACEMENT)) | np.uint64(level)


# ----------------------------------------------------------------------------
# linear tree -- ferreus_bbfmm/src/linear_tree.rs
@dataclass
class TreeLists:
    """bbfmm.rs:111-147 (hash containers -> dict/set; deterministic numbering)."""
    tree: set
    leaves: set
    children: dict
    u_lists: dict
    v_lists: dict
    x_lists: dict | None
    w_lists: dict | None
    level_cells_map: dict
    key_to_index_map: dict
    leaf_source_indices: dict
    depth: int = 0


def build_tree(points: np.ndarray, center, radius: float, max_points_per_cell: int,
               store_empty_leaves: bool, d: int, adaptive_tree: bool) -> TreeLists:
    """linear_tree.rs:20-175"""
    n_points = points.shape[0]
    optimal_depth = int(math.ceil(math.log2(float(n_points)) / d)) if n_points > 0 else 0

    all_nodes = {0}
    leaf_nodes = set()
    children = {}
    level_cells_map = {0: [0]}
    cells_point_indices = {0: np.arange(n_points, dtype=np.int64)}
    leaf_source_indices = {}
    active_cells = deque([0])
    current_level = 0

    while active_cells:
        next_level_cells = set()
        child_level = current_level + 1
        any_child_exceeds = False

        while active_cells:
            cell = active_cells.popleft()
            cell_children = set()
            cell_points = cells_point_indices.get(cell)
            if cell_points is not None and len(cell_points):
                keys = points_to_anchor_keys(points[cell_points], child_level, center, radius)
                order = np.argsort(keys, kind="stable")
                ks = keys[order]
                uniq, starts = np.unique(ks, return_index=True)
                bounds = list(starts) + [len(ks)]
                for ui, key in enumerate(uniq):
                    key = int(key)
                    cell_children.add(key)
                    idx = cell_points[order[bounds[ui]:bounds[ui + 1]]]   # ascending order kept
                    if key in cells_point_indices:
                        cells_point_indices[key] = np.concatenate([cells_point_indices[key], idx])
                    else:
                        cells_point_indices[key] = idx

            if store_empty_leaves:
                active_children = get_children(cell, d)
            else:
                active_children = sorted(cell_children)

            for child in active_children:
                all_nodes.add(child)
                children.setdefault(child, [])
                level_cells_map.setdefault(child_level, []).append(child)
                child_points = cells_point_indices.get(child)
                if child_points is not None:
                    if adaptive_tree:
                        if len(child_points) > max_points_per_cell and child_level < MAXIMUM_LEVEL:
                            next_level_cells.add(child)
                        else:
                            leaf_nodes.add(child)
                            leaf_source_indices.setdefault(child, [])
                            leaf_source_indices[child] = list(leaf_source_indices[child]) + \
                                [int(i) for i in child_points]
                    elif len(child_points) > max_points_per_cell:
                        any_child_exceeds = True
                elif adaptive_tree and store_empty_leaves:
                    leaf_nodes.add(child)

            children[cell] = list(active_children)
            if not adaptive_tree:
                next_level_cells.update(active_children)

        should_subdivide = adaptive_tree or (any_child_exceeds and child_level < MAXIMUM_LEVEL
                                             and child_level < optimal_depth)
        if should_subdivide and next_level_cells:
            active_cells.extend(sorted(next_level_cells))
            current_level += 1
        elif not adaptive_tree:
            for leaf in next_level_cells:
                idx = cells_point_indices.get(leaf)
                if idx is not None:
                    leaf_source_indices.setdefault(leaf, [int(i) for i in idx])
            leaf_nodes.update(next_level_cells)

    depth = current_level + 1

    # deterministic column numbering: by level, then key (reference: HashSet order)
    ordered = sorted(all_nodes, key=lambda k: (k & LEVEL_MASK, k))
    key_to_index = {k: i for i, k in enumerate(ordered)}
    for lvl in level_cells_map:
        level_cells_map[lvl] = sorted(level_cells_map[lvl])

    if adaptive_tree:
        u, v, x, w = get_interaction_lists_adaptive(all_nodes, leaf_nodes, center, radius, d)
    else:
        u, v = get_interaction_lists_regular(all_nodes, leaf_nodes, cells_point_indices, children,
                                             center, radius, d)
        x, w = None, None

    return TreeLists(tree=all_nodes, leaves=leaf_nodes, children=children, u_lists=u, v_lists=v,
                     x_lists=x, w_lists=w, level_cells_map=level_cells_map,
                     key_to_index_map=key_to_index, leaf_source_indices=leaf_source_indices,
                     depth=depth)


def get_interaction_lists_adaptive(tree: set, leaves: set, center, radius: float, d: int):
    """linear_tree.rs:177-395"""
    u_lists, v_lists, w_lists, x_lists = {}, {}, {}, {}
    adj_cache = {}

    def adj(a, b):
        return are_adjacent(a, b, center, radius, d)

    for key in sorted(tree):
        cell_u, cell_v, cell_w = set(), set(), set()
        parent = get_parent(key, d)
        if parent is not None:
            for col in get_neighbours(parent, d):                       # 278-293
                for pcc in get_children(col, d):
                    if pcc in tree and not adj(key, pcc):
                        cell_v.add(pcc)
            if key in leaves:                                             # 295-365
                colleagues = get_neighbours(key, d)
                colleagues_children = [c for col in colleagues for c in get_children(col, d)]
                queue = deque(colleagues)
                visited = set()
                while queue:
                    cur = queue.popleft()
                    if cur in visited:
                        continue
                    visited.add(cur)
                    if adj(key, cur):
                        if cur in leaves:
                            cell_u.add(cur)
                        else:
                            par = get_parent(cur, d)
                            if par is not None:
                                queue.append(par)
                queue = deque(c for c in colleagues_children if c in tree)
                while queue:
                    cur = queue.popleft()
                    if adj(key, cur):
                        if cur in leaves:
                            cell_u.add(cur)
                        else:
                            queue.extend(c for c in get_children(cur, d) if c in tree)
                    else:
                        cell_w.add(cur)
                cell_u.add(key)
        if cell_u:
            u_lists[key] = cell_u
        if cell_v:
            v_lists[key] = cell_v
        if cell_w:
            w_lists[key] = cell_w
    for cell, wl in w_lists.items():                                     # 388-392
        for w in wl:
            x_lists.setdefault(w, set()).add(cell)
    del adj_cache
    return u_lists, v_lists, x_lists, w_lists


def get_interaction_lists_regular(tree, leaves, cells_points_indices, children, center, radius, d):
    """linear_tree.rs:397-485"""
    u_lists, v_lists = {}, {}
    for cell in sorted(tree):
        u, v = set(), set()
        parent = get_parent(cell, d)
        if parent is not None:
            if cell in leaves:
                for sib in children.get(parent, []):
                    if sib in cells_points_indices:
                        u.add(sib)
            for pc in get_neighbours(parent, d):
                if pc not in tree:
                    continue
                for colleague in children.get(pc, []):
                    if colleague in cells_points_indices:
                        if are_adjacent(cell, colleague, center, radius, d):
                            if cell in leaves:
                                u.add(colleague)
                        else:
                            v.add(colleague)
        if cell in leaves:
            u_lists[cell] = u
        v_lists[cell] = v
    return u_lists, v_lists


class PointOutsideTree(Exception):
    """FmmError::PointOutsideTree{point_index} -- bbfmm.rs:20-27"""

    def __init__(self, point_index):
        super().__init__(
            f"FMM evaluation failed: target point at row {point_index} lies outside the tree extents")
        self.point_index = point_index


class KernelDoesNotSupportGradients(Exception):
    """FmmError::KernelDoesNotSupportGradients -- bbfmm.rs:25-26"""


def points_to_keys(points: np.ndarray, leaves: set, depth: int, center, radius: float, d: int):
    """linear_tree.rs:487-520: deepest-level key, walk up until a leaf."""
    keys = points_to_anchor_keys(points, depth, center, radius)
    uniq, inverse = np.unique(keys, return_inverse=True)        # the walk is a function of the deepest key: once per key
    res = np.zeros(len(uniq), dtype=np.uint64)
    bad = []
    for j, k in enumerate(uniq.tolist()):
        cur = k
        while cur not in leaves:
            cur = get_parent(cur, d)
            if cur is None:
                break
        if cur is None:
            bad.append(j)
        else:
            res[j] = cur
    if bad:                                                     # smallest failing row: results are scanned in order (514-517)
        raise PointOutsideTree(int(np.nonzero(np.isin(inverse, np.asarray(bad)))[0][0]))
    return res[inverse]


# ----------------------------------------------------------------------------
# Chebyshev operators -- ferreus_bbfmm/src/chebyshev.rs, aca.rs
def generate_chebyshev_nodes(p: int) -> np.ndarray:
    """chebyshev.rs:32-40 (ascending)"""
    return np.array([math.cos(math.pi * (i + 0.5) / p) for i in reversed(range(p))])


def evaluate_chebyshev_polynomials(p: int, x: np.ndarray, with_derivatives=False):
    """chebyshev.rs:47-110 -> (T [len(x) x p], dT or None)"""
    x = np.asarray(x, dtype=np.float64)
    T = np.ones((len(x), p))
    dT = np.zeros((len(x), p)) if with_derivatives else None
    if p > 1:
        T[:, 1] = x
        if dT is not None:
            dT[:, 1] = 1.0
    for j in range(2, p):
        T[:, j] = 2.0 * x * T[:, j - 1] - T[:, j - 2]
        if dT is not None:
            dT[:, j] = 2.0 * T[:, j - 1] + 2.0 * x * dT[:, j - 1] - dT[:, j - 2]
    return T, dT


def calculate_sn(T: np.ndarray, polynomial_nodes: np.ndarray, p: int) -> np.ndarray:
    """chebyshev.rs:114-127"""
    return ((T @ polynomial_nodes.T) * 2.0 - 1.0) / p


def cartesian_product(values, ncols: int) -> np.ndarray:
    """ferreus_bbfmm/src/utils.rs:123-134 (axis 0 slowest)."""
    values = np.asarray(values)
    base = len(values)
    rows = base ** ncols
    out = np.empty((rows, ncols), dtype=values.dtype)
    i = np.arange(rows)
    for j in range(ncols):
        out[:, j] = values[(i // base ** (ncols - j - 1)) % base]
    return out


def argsort_stable(data):
    """ferreus_bbfmm/src/utils.rs:138-146 (sort_by is stable)."""
    return sorted(range(len(data)), key=lambda i: data[i])


def get_m2m_transfer_matrices(p: int, nodes: np.ndarray, polynomial_nodes: np.ndarray, d: int):
    """chebyshev.rs:146-241 -> list of 2^d matrices [parent_node, child_node]."""
    child_nodes = np.concatenate([(nodes - 1.0) * 0.5, (nodes + 1.0) * 0.5])     # 157-168
    T, _ = evaluate_chebyshev_polynomials(p, child_nodes)
    sn = calculate_sn(T, polynomial_nodes, p)
    halves = (sn[:p], sn[p:])
    mats = []
    for i in range(1 << d):
        acc = None
        for j in range(d):                                                       # 183-192: bit j <-> axis j
            m = halves[1] if (i >> j) & 1 else halves[0]
            acc = m.copy() if acc is None else np.kron(acc, m)
        mats.append(np.ascontiguousarray(acc.T))
    return mats, halves


def get_m2l_vectors(d: int):
    """chebyshev.rs:267-297"""
    all_vecs = cartesian_product(np.arange(-3, 4, dtype=np.int32), d)
    base = cartesian_product(np.arange(0, 4, dtype=np.int32), d)
    ref = []
    for row in base:
        if row[0] >= 2 and all(row[i] <= row[i - 1] for i in range(1, d)):
            ref.append(row.copy())
    return all_vecs, np.array(ref, dtype=np.int32).reshape(-1, d)


def _map_multi_index_to_k(alpha, p):
    """chebyshev.rs:300-315"""
    m = 0
    for a in alpha:
        m = m * p + (a - 1)
    return m


def get_permutation_lookups(d: int, p: int, all_vecs: np.ndarray, ref_vecs: np.ndarray):
    """chebyshev.rs:486-585"""
    axis_order_perms = [list(pm) for pm in itertools.permutations(range(d))]
    axis_sign_perms = cartesian_product(np.array([-1, 1], dtype=np.int32), d)
    multi = cartesian_product(np.arange(1, p + 1, dtype=np.int64), d)
    size = multi.shape[0]

    def diag_perm(sort_idx):                      # 339-342, 346-375
        out = [0] * size
        for j in range(size):
            alpha = multi[j]
            ap = [int(alpha[s]) for s in sort_idx]
            out[_map_multi_index_to_k(ap, p)] = j
        return out

    def axial_perm(signs):                        # 318-336
        out = [0] * size
        for j in range(size):
            alpha = multi[j]
            ap = [p - (int(a) - 1) if signs[i] < 0 else int(a) for i, a in enumerate(alpha)]
            out[_map_multi_index_to_k(ap, p)] = j
        return out

    diag = [diag_perm(pm) for pm in axis_order_perms]
    axial = [axial_perm(row) for row in axis_sign_perms]
    combos = [(a, b) for a in range(len(axial)) for b in range(len(diag))]
    combined = [[axial[a][i] for i in diag[b]] for (a, b) in combos]             # 544-555
    inverse = [argsort_stable(c) for c in combined]                              # 557-560

    sign_rows = [tuple(int(v) for v in row) for row in axis_sign_perms]
    axial_cases = [sign_rows.index(tuple(-1 if v < 0 else 1 for v in vec)) for vec in all_vecs]
    diag_cases = [axis_order_perms.index(argsort_stable([-abs(int(v)) for v in vec]))
                  for vec in all_vecs]
    perm_lookups = [combos.index((a, b)) for a, b in zip(axial_cases, diag_cases)]

    sorted_refs = [tuple(sorted(int(v) for v in row)) for row in ref_vecs]        # 448-483
    ref_lookups = []
    for vec in all_vecs:
        t = tuple(sorted(abs(int(v)) for v in vec))
        ref_lookups.append(sorted_refs.index(t) if t in sorted_refs else 0)
    return (np.array(combined, dtype=np.int32), np.array(inverse, dtype=np.int32),
            np.array(perm_lookups, dtype=np.int32), np.array(ref_lookups, dtype=np.int32))


def kernel_block(kernel_id: int, base_range: float, total_sill: float,
                 tgt: np.ndarray, src: np.ndarray) -> np.ndarray:
    """ferreus_bbfmm/src/utils.rs:64-88 -- a[i, j] = K(tgt_i, src_j)."""
    tgt = np.ascontiguousarray(tgt, dtype=np.float64)
    src = np.ascontiguousarray(src, dtype=np.float64)
    m, d = tgt.shape
    n = src.shape[0]
    a = np.empty((m, n), dtype=np.float64, order="F")
    lib().oracle_kernel_block(I32(kernel_id), F64(base_range), F64(total_sill), I32(d),
                              I64(m), _p(tgt), I64(n), _p(src), _p(a))
    return a


def _argmax_masked(data, mask):
    """aca.rs:146-161 (first strict maximum of |data|*mask; 0 if all zero)."""
    return int(np.argmax(np.abs(data) * mask))


def aca_partial_pivoting(num_rows, num_cols, gen, epsilon):
    """aca.rs:23-136.  gen(r0, r1, c0, c1) -> dense sub-block."""
    unused_rows = np.ones(num_rows)
    unused_cols = np.ones(num_cols)
    max_it = min(num_rows, num_cols)
    tol = epsilon ** 2
    u = np.zeros((num_rows, max_it))
    v = np.zeros((num_cols, max_it))
    residual_norm = 0.0
    i = 0
    sum_k = 0.0
    k = 0
    for _ in range(max_it):
        v_row = gen(i, i + 1, 0, num_cols)[0, :].copy()
        unused_rows[i] = 0
        if k > 0:
            v_row -= u[i, :k] @ v[:, :k].T
        j = _argmax_masked(v_row, unused_cols)
        if v_row[j] == 0.0 and k > 0:
            # Residual row is exactly zero: the cross approximation is exact.  The
            # reference would divide by zero here (aca.rs:76) and fill U/V with NaN;
            # this restatement (and the product) stop instead.  Documented deviation.
            break
        v_row *= 1.0 / v_row[j]
        u_col = gen(0, num_rows, j, j + 1)[:, 0].copy()
        unused_cols[j] = 0
        if k > 0:
            u_col -= (v[j, :k] @ u[:, :k].T)
        i = _argmax_masked(u_col, unused_rows)
        if k > 0:
            sum_k = float((u[:, :k].T @ u_col) @ (v[:, :k].T @ v_row))
        norm_u_v_2 = float(u_col @ u_col) * float(v_row @ v_row)
        residual_norm += norm_u_v_2 + 2.0 * sum_k
        u[:, k] = u_col
        v[:, k] = v_row
        k += 1
        if norm_u_v_2 <= tol * residual_norm:
            break
    return u[:, :k].copy(), v[:, :k].copy()


def calculate_singular_values_cutoff(sigma, epsilon):
    """aca.rs:210-247"""
    sq = np.asarray(sigma, dtype=np.float64) ** 2
    cum = np.cumsum(sq[::-1])[::-1]
    eps_qr = cum[0] * epsilon * epsilon
    below = np.nonzero(cum < eps_qr)[0]
    return int(below[0]) if len(below) else len(cum)


def recompress_aca(u_aca, v_aca, epsilon):
    """aca.rs:173-200 (QR and SVD are faer's in the reference; standard LAPACK here)."""
    qu, ru = np.linalg.qr(u_aca)
    qv, rv = np.linalg.qr(v_aca)
    ur, sr, vrt = np.linalg.svd(ru @ rv.T)
    r = calculate_singular_values_cutoff(sr, epsilon)
    u = qu @ (ur[:, :r] * sr[:r])
    vt = vrt[:r] @ qv.T
    return u, vt


@dataclass
class Operators:
    """bbfmm.rs:154-184"""
    p: int
    d: int
    n: int
    nodes: np.ndarray
    nodes_nd: np.ndarray
    polynomial_nodes: np.ndarray
    m2m: list
    m2m_halves: tuple
    all_vecs: np.ndarray
    ref_vecs: np.ndarray
    perm: np.ndarray
    invperm: np.ndarray
    perm_lookup: np.ndarray
    ref_lookup: np.ndarray
    u: dict = field(default_factory=dict)     # level -> list of U (n x r)
    vt: dict = field(default_factory=dict)    # level -> list of Vt (r x n)


def precompute_approximation_operators(p, d, radius, depth, kernel_id, base_range, total_sill,
                                       compression, epsilon) -> Operators:
    """chebyshev.rs:650-814"""
    try:  # small LAPACK calls: keep BLAS single-threaded (oversubscription is 30x slower)
        from threadpoolctl import threadpool_limits
        with threadpool_limits(1):
            return _precompute_approximation_operators(p, d, radius, depth, kernel_id, base_range,
                                                       total_sill, compression, epsilon)
    except ImportError:
        return _precompute_approximation_operators(p, d, radius, depth, kernel_id, base_range,
                                                   total_sill, compression, epsilon)


def _precompute_approximation_operators(p, d, radius, depth, kernel_id, base_range, total_sill,
                                        compression, epsilon) -> Operators:
    n = p ** d
    nodes = generate_chebyshev_nodes(p)
    nodes_nd = cartesian_product(nodes, d)
    polyn, _ = evaluate_chebyshev_polynomials(p, nodes)
    m2m, halves = get_m2m_transfer_matrices(p, nodes, polyn, d)
    all_vecs, ref_vecs = get_m2l_vectors(d)
    perm, invperm, perm_lookup, ref_lookup = get_permutation_lookups(d, p, all_vecs, ref_vecs)
    ops = Operators(p=p, d=d, n=n, nodes=nodes, nodes_nd=nodes_nd, polynomial_nodes=polyn,
                    m2m=m2m, m2m_halves=halves, all_vecs=all_vecs, ref_vecs=ref_vecs, perm=perm,
                    invperm=invperm, perm_lookup=perm_lookup, ref_lookup=ref_lookup)
    for level in range(2, depth + 1):
        length = radius / float(2 ** (level - 1))                                 # 702
        target_points = nodes_nd * (0.5 * length)                                 # 588-600
        ul, vl = [], []
        for i in range(ref_vecs.shape[0]):
            idx = cartesian_product(np.arange(p), d)
            source_points = (ref_vecs[i][None, :].astype(np.float64) + nodes[idx] * 0.5) * length  # 604-627
            if compression == COMPRESSION_ACA:
                def gen(r0, r1, c0, c1, sp=source_points, tp=target_points):
                    return kernel_block(kernel_id, base_range, total_sill, sp[r0:r1], tp[c0:c1])
                u, v = aca_partial_pivoting(n, n, gen, epsilon)
                tu, tvt = recompress_aca(u, v, epsilon)
            elif compression == COMPRESSION_SVD:
                a = kernel_block(kernel_id, base_range, total_sill, source_points, target_points)
                ur, sr, vrt = np.linalg.svd(a)
                r = calculate_singular_values_cutoff(sr, epsilon)
                tu = ur[:, :r].copy()
                tvt = sr[:r, None] * vrt[:r]
            else:
                tu = kernel_block(kernel_id, base_range, total_sill, source_points, target_points)
                tvt = None
            ul.append(np.asfortranarray(tu))
            vl.append(None if tvt is None else np.asfortranarray(tvt))
        ops.u[level] = ul
        ops.vt[level] = vl
    return ops


# ----------------------------------------------------------------------------
# FmmTree -- ferreus_bbfmm/src/bbfmm.rs
@dataclass
class FmmParams:
    """bbfmm.rs:77-104"""
    max_points_per_cell: int = 256
    compression_type: int = COMPRESSION_ACA
    epsilon: float = 1e-7
    eval_chunk_size: int = 1024

    @staticmethod
    def new_defaults(order: int) -> "FmmParams":
        return FmmParams(256, COMPRESSION_ACA, 10.0 ** (-order), 1024)


def _csr(C, key_to_index, lists, with_keys=False):
    ptr = np.zeros(C + 1, dtype=np.int64)
    items = []
    if lists:
        by_idx = {key_to_index[k]: sorted(v) for k, v in lists.items()}
    else:
        by_idx = {}
    for c in range(C):
        lst = by_idx.get(c, ())
        ptr[c + 1] = ptr[c] + len(lst)
        items.extend(lst)
    idx = np.array([key_to_index[k] for k in items], dtype=np.int64)
    if with_keys:
        return ptr, idx, items
    return ptr, idx


class FmmTree:
    """Restatement of ferreus_bbfmm::FmmTree<K> for the closed kernel set
    (ferreus_rbf_utils::FmmTree, utils.rs:383-494)."""

    def __init__(self, source_points, interpolation_order, kernel_id, adaptive_tree=True,
                 sparse=True, extents=None, params: FmmParams | None = None,
                 base_range=1.0, total_sill=1.0):
        pts = np.array(source_points, dtype=np.float64)
        if pts.ndim == 1:
            pts = pts[:, None]
        self.source_points = np.ascontiguousarray(pts)
        self.order = int(interpolation_order)
        self.kernel_id = int(kernel_id)
        self.base_range = float(base_range)
        self.total_sill = float(total_sill)
        self.adaptive_tree = bool(adaptive_tree)
        self.sparse_tree = bool(sparse)
        tree_extents = list(extents) if extents is not None else get_pointarray_extents(pts)  # bbfmm.rs:281-284
        self.params = params if params is not None else FmmParams.new_defaults(self.order)
        self.d = len(tree_extents) // 2
        if self.d not in (1, 2, 3):
            raise ValueError(f"Unsupported number of dimensions: {self.d}")               # bbfmm.rs:293-298
        self.center, self.radius = calculate_tree_center_and_radius(tree_extents)
        self.nrhs = 1
        self._parent_of = None
        # cpu_baseline mode (bench.py): M2L through register-blocked GEMMs and the near field through gathered,
        # vectorised loops (oracle/passes.c: oracle_m2l_gemm, leaf-pass flag 8) -- the arithmetic the reference's
        # faer calls and monomorphised loops do; off for every test that uses the oracle as the checker
        self.gemm_shaped = False
        self.tl = build_tree(self.source_points, self.center, self.radius,
                             self.params.max_points_per_cell, not self.sparse_tree, self.d,
                             self.adaptive_tree)
        self.depth = self.tl.depth
        self.ops = precompute_approximation_operators(
            self.order, self.d, self.radius, self.depth, self.kernel_id, self.base_range,
            self.total_sill, self.params.compression_type, self.params.epsilon)
        self._flatten()
        self.M = None
        self.L = None

    # -- flat arrays for passes.c
    def _flatten(self):
        tl, d = self.tl, self.d
        k2i = tl.key_to_index_map
        self.cell_keys = sorted(tl.tree, key=lambda k: (k & LEVEL_MASK, k))
        C = self.C = len(self.cell_keys)
        self.cell_level = np.array([k & LEVEL_MASK for k in self.cell_keys], dtype=np.int32)
        cl = [get_center_length(k, self.center, self.radius, d) for k in self.cell_keys]
        self.centers = np.array([c for c, _ in cl], dtype=np.float64).reshape(C, d)
        self.lengths = np.array([l for _, l in cl], dtype=np.float64)
        self.octant = np.array([get_child_index(k, d) for k in self.cell_keys], dtype=np.int32)
        self.is_leaf = np.array([k in tl.leaves for k in self.cell_keys], dtype=np.uint8)
        self.child_ptr, self.child_idx = _csr(C, k2i, tl.children)
        self.src_ptr = np.zeros(C + 1, dtype=np.int64)
        src = []
        for c, k in enumerate(self.cell_keys):
            lst = tl.leaf_source_indices.get(k, ())
            self.src_ptr[c + 1] = self.src_ptr[c] + len(lst)
            src.extend(lst)
        self.src_idx = np.array(src, dtype=np.int64)
        self.u_ptr, self.u_idx = _csr(C, k2i, tl.u_lists)
        self.v_ptr, self.v_idx, v_keys = _csr(C, k2i, tl.v_lists, with_keys=True)
        self.w_ptr, self.w_idx = _csr(C, k2i, tl.w_lists or {})
        self.x_ptr, self.x_idx = _csr(C, k2i, tl.x_lists or {})
        # transfer index of every V pair: bbfmm.rs:872-888, 989-998
        tidx = np.zeros(len(self.v_idx), dtype=np.int32)
        for c in range(C):
            cc, ln = self.centers[c], self.lengths[c]
            for q in range(self.v_ptr[c], self.v_ptr[c + 1]):
                vc = self.centers[self.v_idx[q]]
                t = 0
                for a in range(d):
                    val = (cc[a] - vc[a]) / ln
                    r = int(math.floor(abs(val) + 0.5)) * (1 if val >= 0 else -1)   # f64::round
                    t = t * 7 + (r + 3)
                tidx[q] = t
        self.v_tidx = tidx
        self.level_cells = {lvl: np.array([k2i[k] for k in keys], dtype=np.int64)
                            for lvl, keys in tl.level_cells_map.items()}
        self.leaf_cells = np.nonzero(self.is_leaf)[0].astype(np.int64)
        self.m2m_flat = np.ascontiguousarray(np.stack(self.ops.m2m))
        self.polyn = np.ascontiguousarray(self.ops.polynomial_nodes)
        self.nodes_nd = np.ascontiguousarray(self.ops.nodes_nd)
        # operator buffers per level
        self.opbuf = {}
        for level in range(2, self.depth + 1):
            parts, u_off, vt_off, rank = [], [], [], []
            off = 0
            for ref in range(self.ops.ref_vecs.shape[0]):
                U = self.ops.u[level][ref]
                Vt = self.ops.vt[level][ref]
                u_off.append(off)
                parts.append(U.ravel(order="F"))
                off += U.size
                rank.append(U.shape[1])
                if Vt is not None:
                    vt_off.append(off)
                    parts.append(Vt.ravel(order="F"))
                    off += Vt.size
                else:
                    vt_off.append(0)
            self.opbuf[level] = (np.concatenate(parts), np.array(u_off, dtype=np.int64),
                                 np.array(vt_off, dtype=np.int64), np.array(rank, dtype=np.int32))

    def set_m2l_operators(self, factors):
        """Replace the M2L operators: factors[level][ref] = (U, Vt or None).  Used by the parity
        tests to run the oracle's passes on the product's host-computed operators, so that only
        summation order differs."""
        for level, lst in factors.items():
            self.ops.u[level] = [np.asfortranarray(u) for u, _ in lst]
            self.ops.vt[level] = [None if vt is None else np.asfortranarray(vt) for _, vt in lst]
        self._flatten()

    # -- helpers
    def _ancestor_flags(self, leaf_cell_indices):
        """cells_with_sources / cells_with_targets: ancestors of the given leaves
        (bbfmm.rs:395-398, 475-478)."""
        flags = np.zeros(self.C, dtype=np.uint8)
        if self._parent_of is None:                             # parent index of every cell (-1: none), from the children lists
            par = np.full(self.C, -1, dtype=np.int64)
            counts = np.diff(self.child_ptr)
            par[self.child_idx] = np.repeat(np.arange(self.C, dtype=np.int64), counts)
            self._parent_of = par
        cur = np.unique(np.asarray(leaf_cell_indices, dtype=np.int64))
        while len(cur):                                         # get_ancestors includes the cell itself (morton.rs:193-210)
            cur = cur[flags[cur] == 0]
            flags[cur] = 1
            cur = np.unique(self._parent_of[cur])
            cur = cur[cur >= 0]
        return flags

    def _zeroed(self, name, shape):
        """A zero-filled coefficient array (reset_*_coefficients, bbfmm.rs:619-632).  In the cpu_baseline mode the array of
        the previous pass is kept and cleared by all threads (oracle_zero) instead of taking fresh pages from the kernel."""
        old = getattr(self, name, None)
        if self.gemm_shaped and isinstance(old, np.ndarray) and old.shape == tuple(shape) and old.flags["C_CONTIGUOUS"]:
            lib().oracle_zero(_p(old), I64(old.size))
            return old
        return np.zeros(shape)

    def _w(self, weights):
        w = np.asarray(weights, dtype=np.float64)
        if w.ndim == 1:
            w = w[:, None]
        return np.asfortranarray(w)

    # -- public API (bbfmm.rs:383-616)
    def set_weights(self, weights):
        """bbfmm.rs:383-401 + upward_pass 666-688"""
        w = self._w(weights)
        self.nrhs = K = w.shape[1]
        n, C = self.ops.n, self.C
        self.M = self._zeroed("M", (K, C, n))
        L = lib()
        leafs_with_sources = [c for c in self.leaf_cells if self.src_ptr[c + 1] > self.src_ptr[c]]
        with_src = self._ancestor_flags(leafs_with_sources)
        lw = np.array(leafs_with_sources, dtype=np.int64)
        L.oracle_p2m(I32(self.order), I32(self.d), I64(C), I32(K), _p(lw), I64(len(lw)),
                     _p(self.centers), _p(self.lengths), _p(self.src_ptr), _p(self.src_idx),
                     _p(self.source_points), _p(w), I64(w.shape[0]), _p(self.polyn), _p(self.M))
        for level in range(self.depth - 1, 0, -1):
            cells = self.level_cells.get(level)
            if cells is None:
                continue
            parents = np.ascontiguousarray(cells[with_src[cells] == 1])
            L.oracle_m2m(I32(n), I64(C), I32(K), _p(parents), I64(len(parents)),
                         _p(self.child_ptr), _p(self.child_idx), _p(self.octant),
                         _p(self.m2m_flat), _p(self.M))

    def _assign_targets(self, target_points):
        """points_to_keys + get_points_to_leaves_map (linear_tree.rs:487-534)."""
        tp = np.array(target_points, dtype=np.float64)
        if tp.ndim == 1:
            tp = tp[:, None]
        tp = np.ascontiguousarray(tp)
        keys = points_to_keys(tp, self.tl.leaves, self.depth, self.center, self.radius, self.d)
        k2i = self.tl.key_to_index_map
        uniq, inverse = np.unique(keys, return_inverse=True)
        cell_of = np.array([k2i[int(k)] for k in uniq.tolist()], dtype=np.int64)[inverse]
        order = np.argsort(cell_of, kind="stable")             # rows ascending inside a leaf
        counts = np.bincount(cell_of, minlength=self.C)
        tgt_ptr = np.zeros(self.C + 1, dtype=np.int64)
        np.cumsum(counts, out=tgt_ptr[1:])
        return tp, tgt_ptr, order.astype(np.int64), cell_of

    def _downward(self, w, active):
        """downward_pass, bbfmm.rs:778-857"""
        K, n, C = self.nrhs, self.ops.n, self.C
        self.L = self._zeroed("L", (K, C, n))
        L = lib()
        compressed = 0 if self.params.compression_type == COMPRESSION_NONE else 1
        for level in range(1, self.depth + 1):
            cells = self.level_cells.get(level)
            if cells is None:
                continue
            cells = np.ascontiguousarray(cells[active[cells] == 1])
            if level >= 2 and len(cells):
                buf, u_off, vt_off, rank = self.opbuf[level]
                m2l = L.oracle_m2l_gemm if self.gemm_shaped else L.oracle_m2l
                m2l(I32(n), I64(C), I32(K), _p(cells), I64(len(cells)), _p(self.v_ptr),
                             _p(self.v_idx), _p(self.v_tidx), I32(len(rank)), _p(u_off),
                             _p(vt_off), _p(rank), _p(buf), I32(compressed), _p(self.ops.perm),
                             _p(self.ops.invperm), _p(self.ops.perm_lookup),
                             _p(self.ops.ref_lookup), _p(self.M), _p(self.L))
            if self.adaptive_tree and len(cells):
                L.oracle_p2l(I32(self.kernel_id), F64(self.base_range), F64(self.total_sill),
                             I32(n), I32(self.d), I64(C), I32(K), _p(cells), I64(len(cells)),
                             _p(self.centers), _p(self.lengths), _p(self.nodes_nd),
                             _p(self.x_ptr), _p(self.x_idx), _p(self.src_ptr), _p(self.src_idx),
                             _p(self.source_points), _p(w), I64(w.shape[0]), _p(self.L))
        for level in range(1, self.depth + 1):
            cells = self.level_cells.get(level)
            if cells is None:
                continue
            parents = np.ascontiguousarray(cells[active[cells] == 1])
            L.oracle_l2l(I32(n), I64(C), I32(K), _p(parents), I64(len(parents)),
                         _p(self.child_ptr), _p(self.child_idx), _p(self.octant), _p(active),
                         _p(self.m2m_flat), _p(self.L))

    def _leaf_pass(self, w, tp, tgt_ptr, tgt_idx, with_grads, flags=7):
        """leaf_pass, bbfmm.rs:1089-1159"""
        K, d = self.nrhs, self.d
        m = tp.shape[0]
        out = np.zeros((m, K), order="F")
        grad = np.zeros((m, K * d), order="F") if with_grads else None
        leaves = np.ascontiguousarray(self.leaf_cells)
        lib().oracle_leaf_pass(
            I32(self.kernel_id), F64(self.base_range), F64(self.total_sill), I32(self.order),
            I32(d), I64(self.C), I32(K), _p(leaves), I64(len(leaves)), _p(self.centers),
            _p(self.lengths), _p(self.nodes_nd), _p(self.polyn), _p(self.u_ptr), _p(self.u_idx),
            _p(self.w_ptr) if self.adaptive_tree else ctypes.c_void_p(0), _p(self.w_idx),
            _p(self.src_ptr), _p(self.src_idx), _p(tgt_ptr), _p(tgt_idx), _p(self.source_points),
            _p(tp), _p(w), I64(w.shape[0]), _p(self.M), _p(self.L), _p(out), I64(m),
            _p(grad), I64(m), I32(flags | (8 if self.gemm_shaped else 0)))
        return out, grad

    def _check_grads(self):
        if not lib().oracle_kernel_has_gradient(I32(self.kernel_id)):
            raise KernelDoesNotSupportGradients()

    def _eval(self, weights, target_points, with_grads, flags=7):
        """_eval, bbfmm.rs:444-507"""
        w = self._w(weights)
        tp, tgt_ptr, tgt_idx, cell_of = self._assign_targets(target_points)
        active = self._ancestor_flags(np.unique(cell_of))
        self._downward(w, active)
        if with_grads:
            self._check_grads()
        return self._leaf_pass(w, tp, tgt_ptr, tgt_idx, with_grads, flags)

    def evaluate(self, weights, target_points):
        return self._eval(weights, target_points, False)[0]

    def evaluate_with_gradients(self, weights, target_points):
        return self._eval(weights, target_points, True)

    def set_local_coefficients(self, weights):
        """bbfmm.rs:518-524"""
        self._downward(self._w(weights), np.ones(self.C, dtype=np.uint8))

    def _eval_leaves(self, weights, target_points, with_grads):
        """_eval_leaves, bbfmm.rs:570-616"""
        w = self._w(weights)
        tp, tgt_ptr, tgt_idx, _ = self._assign_targets(target_points)
        if with_grads:
            self._check_grads()
        return self._leaf_pass(w, tp, tgt_ptr, tgt_idx, with_grads)

    def evaluate_leaves(self, weights, target_points):
        return self._eval_leaves(weights, target_points, False)[0]

    def evaluate_leaves_with_gradients(self, weights, target_points):
        return self._eval_leaves(weights, target_points, True)

    # -- structure dump used by the parity tests (sets, not column numbers)
    def structure(self):
        tl = self.tl
        def as_sorted(dct):
            return {int(k): sorted(int(v) for v in vs) for k, vs in (dct or {}).items()}
        return {
            "depth": int(self.depth),
            "center": [float(c) for c in self.center],
            "radius": float(self.radius),
            "tree": sorted(int(k) for k in tl.tree),
            "leaves": sorted(int(k) for k in tl.leaves),
            "leaf_source_indices": {int(k): [int(i) for i in v]
                                    for k, v in tl.leaf_source_indices.items()},
            "u": as_sorted(tl.u_lists), "v": as_sorted(tl.v_lists),
            "w": as_sorted(tl.w_lists), "x": as_sorted(tl.x_lists),
        }


# ----------------------------------------------------------------------------
def dense_sum(kernel_id, base_range, total_sill, targets, sources, weights, with_grads=False):
    """Ground truth: y = K(X_t, X_s) W  (ferreus_rbf_utils/src/utils.rs:288-312)."""
    t = np.ascontiguousarray(np.atleast_2d(np.asarray(targets, dtype=np.float64)))
    s = np.ascontiguousarray(np.atleast_2d(np.asarray(sources, dtype=np.float64)))
    w = np.asarray(weights, dtype=np.float64)
    if w.ndim == 1:
        w = w[:, None]
    w = np.asfortranarray(w)
    m, d = t.shape
    K = w.shape[1]
    if K > 16:
        raise ValueError("dense_sum handles at most 16 right-hand sides (use kernel_matrix)")
    out = np.zeros((m, K), order="F")
    grad = np.zeros((m, K * d), order="F") if with_grads else None
    lib().oracle_dense_sum(I32(kernel_id), F64(base_range), F64(total_sill), I32(d), I64(m),
                           _p(t), I64(s.shape[0]), _p(s), I32(K), _p(w), I64(w.shape[0]),
                           _p(out), I64(m), _p(grad), I64(m))
    return (out, grad) if with_grads else out


def kernel_matrix(kernel_id, base_range, total_sill, targets, sources):
    """A[i, j] = phi(t_i, s_j)  (get_a_matrix, ferreus_rbf_utils/src/utils.rs:288-312)"""
    t = np.ascontiguousarray(np.atleast_2d(np.asarray(targets, dtype=np.float64)))
    s = np.ascontiguousarray(np.atleast_2d(np.asarray(sources, dtype=np.float64)))
    out = np.zeros((t.shape[0], s.shape[0]), order="F")
    lib().oracle_kernel_matrix(I32(kernel_id), F64(base_range), F64(total_sill), I32(t.shape[1]), I64(t.shape[0]),
                               _p(t), I64(s.shape[0]), _p(s), _p(out), I64(t.shape[0]))
    return out


def kernel_phi(kernel_id, r, base_range=1.0, total_sill=1.0):
    """kernel_phi, ferreus_rbf_utils/src/utils.rs:537-551"""
    return float(lib().oracle_kernel_phi_r2(I32(kernel_id), F64(base_range), F64(total_sill),
                                            F64(r * r)))


def select_mat_rows(existing_mat, row_indices):
    """select_mat_rows, ferreus_rbf_utils/src/utils.rs:44-56: an owned matrix made of the wanted rows, in the order given
    (out[i, j] = existing_mat[row_indices[i], j], the reference's Mat::from_fn)."""
    return np.take(np.asarray(existing_mat), np.asarray(row_indices, dtype=np.int64), axis=0).copy()


def fast_matrix_vector_product(tree: FmmTree, weights, basis_size=0, target_indices=None,
                               polynomial_matrix=None, nugget=0.0):
    """ferreus_rbf/src/rbf.rs:1338-1379"""
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    result = np.zeros(len(w))
    wlen = len(w) - basis_size
    idx = np.arange(wlen) if target_indices is None else np.asarray(target_indices, dtype=np.int64)
    tree.set_weights(w[:, None])
    targets = tree.source_points if target_indices is None else select_mat_rows(tree.source_points, idx)   # rbf.rs:1359-1360
    vals = tree.evaluate(w[:, None], targets)[:, 0]
    result[idx] = vals + w[idx] * nugget
    if polynomial_matrix is not None:
        result[idx] += polynomial_matrix[idx] @ w[wlen:wlen + basis_size]
    return result
