"""The integer tables the reference ships for this path (tests/golden/reference_tables.json, extracted by
tests/golden/make_reference_tables.py from morton_constants.rs:12-346 and chebyshev.rs:245-266) against the
oracle's and the product's bit arithmetic -- bit-exact, no GPU.

`lut_encode` / `lut_decode` below restate morton.rs:58-167 literally ON the reference's lookup tables; the
oracle (oracle/bbfmm_oracle.py) and the product (csrc/morton.hpp through the bbfmm_debug_morton_* hooks)
compute the same keys without tables."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from ferreus_rbf_rs_amd import _lib as L
from oracle import bbfmm_oracle as O

GOLD = os.path.join(ROOT, "tests", "golden", "reference_tables.json")
T = json.load(open(GOLD))
C = T["morton_constants"]


def lut_encode(anchor, level, d):
    """morton.rs:58-119 on the reference's tables"""
    bm, bd = C["BYTE_MASK"], C["BYTE_DISPLACEMENT"]
    code = 0
    if d == 1:
        (x,) = anchor
        code |= C["MORTON_ENCODE_1D_LOOKUP"][(x >> bd) & bm]
        code <<= 8
        code |= C["MORTON_ENCODE_1D_LOOKUP"][x & bm]
    elif d == 2:
        x, y = anchor
        code |= C["MORTON_ENCODE_2D_Y_LOOKUP"][(y >> bd) & bm] | C["MORTON_ENCODE_2D_X_LOOKUP"][(x >> bd) & bm]
        code <<= 16
        code |= C["MORTON_ENCODE_2D_Y_LOOKUP"][y & bm] | C["MORTON_ENCODE_2D_X_LOOKUP"][x & bm]
    else:
        x, y, z = anchor
        code |= (C["MORTON_ENCODE_3D_Z_LOOKUP"][(z >> bd) & bm] | C["MORTON_ENCODE_3D_Y_LOOKUP"][(y >> bd) & bm]
                 | C["MORTON_ENCODE_3D_X_LOOKUP"][(x >> bd) & bm])
        code <<= 24
        code |= (C["MORTON_ENCODE_3D_Z_LOOKUP"][z & bm] | C["MORTON_ENCODE_3D_Y_LOOKUP"][y & bm]
                 | C["MORTON_ENCODE_3D_X_LOOKUP"][x & bm])
    return ((code << C["LEVEL_DISPLACEMENT"]) | level) & 0xFFFFFFFFFFFFFFFF


def lut_decode(key, d):
    """morton.rs:127-167 on the reference's tables"""
    level = key & C["LEVEL_MASK"]
    k = key >> C["LEVEL_DISPLACEMENT"]
    a = [0] * d
    if d == 1:
        a[0] |= C["MORTON_DECODE_1D_LOOKUP"][(k >> 8) & C["BYTE_MASK"]] << 8
        a[0] |= C["MORTON_DECODE_1D_LOOKUP"][k & C["BYTE_MASK"]]
    elif d == 2:
        for i in range(7):
            a[0] |= C["MORTON_DECODE_2D_X_LOOKUP"][(k >> (i * 8)) & C["EIGHT_BIT_MASK"]] << (4 * i)
            a[1] |= C["MORTON_DECODE_2D_Y_LOOKUP"][(k >> (i * 8)) & C["EIGHT_BIT_MASK"]] << (4 * i)
    else:
        for i in range(7):
            a[0] |= C["MORTON_DECODE_3D_X_LOOKUP"][(k >> (i * 9)) & C["NINE_BIT_MASK"]] << (3 * i)
            a[1] |= C["MORTON_DECODE_3D_Y_LOOKUP"][(k >> (i * 9)) & C["NINE_BIT_MASK"]] << (3 * i)
            a[2] |= C["MORTON_DECODE_3D_Z_LOOKUP"][(k >> (i * 9)) & C["NINE_BIT_MASK"]] << (3 * i)
    return tuple(a), level


def _anchors(d, rng):
    edge = [0, 1, 2, 127, 128, 255, 256, 257, 32767, 32768, 65534, 65535]
    out = [tuple(rng.choice(edge, d)) for _ in range(300)]
    out += [tuple(int(v) for v in rng.integers(0, 1 << 16, d)) for _ in range(3000)]
    out += [tuple([v] * d) for v in edge]
    return [tuple(int(v) for v in a) for a in out]


def test_fixture_is_what_the_reference_holds():
    """Regenerating from /root/reference reproduces the committed file (skipped where the reference is absent)."""
    if not os.path.exists("/root/reference/ferreus_bbfmm/src/morton_constants.rs"):
        pytest.skip("reference checkout not present on this machine")
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_reference_tables as M
    consts = M.parse_constants("/root/reference/ferreus_bbfmm/src/morton_constants.rs")
    v2, v3 = M.parse_reference_vectors("/root/reference/ferreus_bbfmm/src/chebyshev.rs")
    assert consts == C and v2 == T["m2l_reference_vectors"]["2"] and v3 == T["m2l_reference_vectors"]["3"]


def test_scalar_constants():
    assert (C["MAXIMUM_LEVEL"], C["LEVEL_DISPLACEMENT"], C["LEVEL_MASK"]) == (O.MAXIMUM_LEVEL, O.LEVEL_DISPLACEMENT, O.LEVEL_MASK)
    assert (C["MAXIMUM_LEVEL"], C["LEVEL_DISPLACEMENT"], C["LEVEL_MASK"]) == (16, 15, 0x7FFF)     # csrc/morton.hpp:16-18


@pytest.mark.parametrize("d", [1, 2, 3])
def test_every_lookup_table_entry_is_the_bit_spread(d):
    """morton_constants.rs:77-346 entry by entry: encode tables spread the 8 bits of a byte with stride d (axis a
    shifted by a); decode tables gather every d-th bit of a 8/8/9-bit group."""
    names = {1: [""], 2: ["X", "Y"], 3: ["X", "Y", "Z"]}[d]
    for a, ax in enumerate(names):
        enc = C[f"MORTON_ENCODE_{d}D_{ax}_LOOKUP".replace("__", "_")] if d > 1 else C["MORTON_ENCODE_1D_LOOKUP"]
        dec = C[f"MORTON_DECODE_{d}D_{ax}_LOOKUP".replace("__", "_")] if d > 1 else C["MORTON_DECODE_1D_LOOKUP"]
        assert enc == [O._spread(v, d) << a for v in range(256)]
        width = 9 if d == 3 else 8
        assert len(dec) == 1 << width
        want = []
        for g in range(1 << width):
            v = 0
            for i in range((width + d - 1 - a) // d):
                v |= ((g >> (d * i + a)) & 1) << i
            want.append(v)
        assert dec == want


@pytest.mark.parametrize("d", [1, 2, 3])
def test_encode_decode_bit_exact_oracle_and_product(d):
    lib = L.load()
    rng = np.random.default_rng(d)
    for anchor in _anchors(d, rng):
        for level in (0, 1, 7, 16):
            key = lut_encode(anchor, level, d)
            assert O.encode_morton_point(anchor, level, d) == key
            arr = (ctypes.c_uint64 * 3)(*anchor)
            assert lib.bbfmm_debug_morton_encode(d, arr, level) == key
            want = lut_decode(key, d)
            assert want == (anchor, level)                                   # the tables invert each other
            assert O.decode_key(key, d) == want
            out = (ctypes.c_uint64 * 3)()
            lv = ctypes.c_uint64()
            lib.bbfmm_debug_morton_decode(d, key, out, ctypes.byref(lv))
            assert (tuple(out[:d]), lv.value) == want
    # keys that do not come from encode (bits above 16 per axis set): decode reads 21 / 28 / 16 bits per axis
    for _ in range(500):
        key = int(rng.integers(0, 1 << 63)) | (int(rng.integers(0, 2)) << 63)
        want = lut_decode(key, d)
        assert O.decode_key(key, d) == want
        out = (ctypes.c_uint64 * 3)()
        lv = ctypes.c_uint64()
        lib.bbfmm_debug_morton_decode(d, key, out, ctypes.byref(lv))
        assert (tuple(out[:d]), lv.value) == want


@pytest.mark.parametrize("d", [1, 2, 3])
def test_direction_vectors_and_neighbour_order(d):
    """morton_constants.rs:32-74: the order fixes the order in which colleagues are visited (linear_tree.rs)."""
    lib = L.load()
    ref = C[f"DIRECTION_VECTORS_{d}D"]
    ref = [[v] for v in ref] if d == 1 else ref
    assert [list(v) for v in O.DIRECTIONS[d]] == ref
    buf = (ctypes.c_int32 * (26 * 3))()
    n = lib.bbfmm_debug_direction_vectors(d, buf)
    assert n == len(ref) and [list(buf[i * d:(i + 1) * d]) for i in range(n)] == ref
    rng = np.random.default_rng(10 + d)
    for level in (1, 2, 5, 16):
        for _ in range(200):
            anchor = tuple(int(v) for v in rng.integers(0, 1 << level, d))
            if rng.random() < 0.3:                                            # on the boundary of the root box
                anchor = tuple(rng.choice([0, (1 << level) - 1]) if rng.random() < 0.7 else a for a in anchor)
            anchor = tuple(int(v) for v in anchor)
            key = lut_encode(anchor, level, d)
            want = [lut_encode(tuple(a + dv for a, dv in zip(anchor, vec)), level, d) for vec in ref
                    if all(0 <= a + dv < (1 << level) for a, dv in zip(anchor, vec))]        # morton.rs:214-263
            assert O.get_neighbours(key, d) == want
            out = (ctypes.c_uint64 * 26)()
            cnt = lib.bbfmm_debug_morton_neighbours(d, key, out)
            assert list(out[:cnt]) == want


@pytest.mark.parametrize("d", [2, 3])
def test_m2l_reference_vectors(d):
    """chebyshev.rs:245-266 lists the reference vectors; the code generates them in lexicographic order
    (chebyshev.rs:272-294), so they are compared as sets."""
    import ferreus_rbf_rs_amd as F
    want = sorted(tuple(v) for v in T["m2l_reference_vectors"][str(d)])
    _, ref = O.get_m2l_vectors(d)
    assert sorted(tuple(int(x) for x in v) for v in ref) == want
    t = F.FmmTree(np.random.default_rng(0).random((200, d)), 3, F.KernelParams(F.FmmKernelType.LinearRbf), True, True,
                  host_only=True)
    n_ref = ctypes.c_int32()
    L.load().bbfmm_debug_reference_vectors(t._h, None, ctypes.byref(n_ref))
    buf = (ctypes.c_int32 * (n_ref.value * d))()
    assert L.load().bbfmm_debug_reference_vectors(t._h, buf, None) == 0
    got = [tuple(buf[i * d:(i + 1) * d]) for i in range(n_ref.value)]
    assert sorted(got) == want
    assert got == [tuple(int(x) for x in v) for v in ref]                    # same (code) order as the oracle
