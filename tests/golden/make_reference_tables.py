#!/usr/bin/env python3
"""Extracts the integer tables the reference ships for this path into tests/golden/reference_tables.json.

Run in the build container (needs /root/reference; the GPU box has only the JSON):
    python tests/golden/make_reference_tables.py

Sources (numbers only -- table VALUES are data, no source text is kept):
  * ferreus_bbfmm/src/morton_constants.rs:12-29   scalar constants (level bits, masks)
  * ferreus_bbfmm/src/morton_constants.rs:32-74   neighbour direction vectors, 1-D / 2-D / 3-D, in order
  * ferreus_bbfmm/src/morton_constants.rs:77-346  the Morton encode / decode byte lookup tables
  * ferreus_bbfmm/src/chebyshev.rs:245-266        the 7 (2-D) and 16 (3-D) M2L reference vectors listed in
                                                  the doc comment of get_m2l_vectors

These are the only bit-exact vectors the reference holds for the BBFMM path; the oracle
(oracle/bbfmm_oracle.py) and the product (csrc/morton.hpp) replace the lookup tables by bit arithmetic and
are tested against them in tests/test_reference_tables.py.
"""
import json
import os
import re
import sys

REF = os.environ.get("FERREUS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _ints(text):
    return [int(t, 0) for t in re.findall(r"-?(?:0x[0-9A-Fa-f]+|\d+)", text)]


def parse_constants(path):
    src = open(path).read()
    out = {}
    # pub const NAME: TYPE = VALUE;   (VALUE a scalar or a possibly nested array)
    for m in re.finditer(r"pub const (\w+):\s*([^=]+?)=\s*(.*?);", src, re.S):
        name, ty, val = m.group(1), m.group(2).strip(), m.group(3)
        nums = _ints(val)
        if "[" not in ty:
            out[name] = nums[0]
            continue
        dims = [int(x) for x in re.findall(r";\s*(\d+)\]", ty)]          # innermost first
        if len(dims) == 1:
            assert len(nums) == dims[0], (name, len(nums), dims)
            out[name] = nums
        else:
            inner, outer = dims[0], dims[1]
            assert len(nums) == inner * outer, (name, len(nums), dims)
            out[name] = [nums[i * inner:(i + 1) * inner] for i in range(outer)]
    return out


def parse_reference_vectors(path):
    lines = open(path).read().splitlines()[244:266]                      # chebyshev.rs:245-266
    v2, v3 = [], []
    for ln in lines:
        for grp in re.findall(r"\[([^\]]*)\]", ln):
            nums = _ints(grp)
            if len(nums) == 2:
                v2.append(nums)
            elif len(nums) == 3:
                v3.append(nums)
    return v2, v3


def main():
    consts = parse_constants(os.path.join(REF, "ferreus_bbfmm/src/morton_constants.rs"))
    v2, v3 = parse_reference_vectors(os.path.join(REF, "ferreus_bbfmm/src/chebyshev.rs"))
    assert len(v2) == 7 and len(v3) == 16, (v2, v3)
    out = {
        "source": {"morton_constants": "ferreus_bbfmm/src/morton_constants.rs:12-346",
                   "reference_vectors": "ferreus_bbfmm/src/chebyshev.rs:245-266 (doc comment of get_m2l_vectors)"},
        "morton_constants": consts,
        "m2l_reference_vectors": {"2": v2, "3": v3},
    }
    path = os.path.join(HERE, "reference_tables.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"), sort_keys=True)
        f.write("\n")
    print(path, {k: (len(v) if isinstance(v, list) else v) for k, v in consts.items()})
    return 0


if __name__ == "__main__":
    sys.exit(main())
