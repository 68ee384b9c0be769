#!/usr/bin/env python3
"""Extracts the known answers of the reference's monomial-basis tests into tests/golden/reference_monomials.json.

Run in the build container (needs /root/reference; the GPU box has only the JSON):
    python tests/golden/make_reference_monomials.py

Source (numbers only -- the matrices' VALUES are data, no source text is kept):
  * ferreus_rbf/src/polynomials.rs:163-242   nine `#[test]` cases of evaluate_monomials (polynomials.rs:30-74): points,
    degree 0 / 1 / 2 in 1-D / 2-D / 3-D, the expected monomial matrix; translation 0 and scale 1 (run_case, 144-161),
    tolerance atol 1e-12 + rtol 1e-10.

evaluate_monomials builds the polynomial part P of the system the FGMRES matvec applies (`y_i += P[i,:] lambda`,
rbf.rs:1366-1376, built at rbf.rs:476-491) and every domain's polynomial block (domain.rs:171-212): these nine matrices
are the only f64 known answers the reference holds on that path.  tests/test_reference_monomials.py checks the oracle
(oracle/ddm.py) and the product (bbfmm_debug_evaluate_monomials -> csrc/ddm_monomials.hpp, the one definition the
solver's and the domains' matrices use) against them."""
import json
import os
import re

REF = os.environ.get("FERREUS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def _matrix(text):
    rows = re.findall(r"\[([^\[\]]*)\]", text)
    return [[float(x) for x in re.findall(r"-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?", r)] for r in rows if r.strip()]


def main():
    src = open(os.path.join(REF, "ferreus_rbf", "src", "polynomials.rs")).read()
    tests = src[src.index("#[cfg(test)]"):]
    cases = []
    for m in re.finditer(r"fn (monomials_\w+)\(\)\s*\{(.*?)run_case\(points,\s*(\d+),\s*expected\);", tests, re.S):
        name, body, degree = m.group(1), m.group(2), int(m.group(3))
        pm = re.search(r"let points = mat!\[(.*?)\];", body, re.S)
        em = re.search(r"let expected = mat!\[(.*?)\];", body, re.S)
        pts, exp = _matrix(pm.group(1)), _matrix(em.group(1))
        assert len(pts) == len(exp) and all(len(r) == len(exp[0]) for r in exp), name
        cases.append({"name": name, "degree": degree, "points": pts, "expected": exp})
    assert len(cases) == 9, len(cases)
    out = {"source": "ferreus_rbf/src/polynomials.rs:163-242 (values of the `points` and `expected` matrices of the nine "
                     "monomials_* tests; translation 0, scale 1; tolerance atol 1e-12 + rtol 1e-10, polynomials.rs:137-142)",
           "cases": cases}
    with open(os.path.join(HERE, "reference_monomials.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(len(cases), "cases:", ", ".join(c["name"] for c in cases))


if __name__ == "__main__":
    main()
