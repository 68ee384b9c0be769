"""The C-ABI library loads and exports every symbol include/ferreus_bbfmm_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from ferreus_rbf_rs_amd import _lib as L

HEADER = os.path.join(ROOT, "include", "ferreus_bbfmm_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbfmm_[a-z0-9_]+)\s*\(", text)) - {"bbfmm_handle", "bbfmm_params",
                                                                        "bbfmm_tree_stats", "bbfmm_status"})


def test_header_symbols_are_exported():
    lib = ctypes.CDLL(L.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"


def test_python_binding_covers_header():
    assert sorted(L.SIGNATURES) == declared_functions()
    L.load()


def test_params_defaults_match_reference():
    # FmmParams::new_defaults, ferreus_bbfmm/src/bbfmm.rs:96-103
    lib = L.load()
    for order in (5, 7, 9, 11):
        p = L.Params()
        lib.bbfmm_params_defaults(order, ctypes.byref(p))
        assert p.max_points_per_cell == 256
        assert p.compression_type == 2          # ACA
        assert p.eval_chunk_size == 1024
        assert p.epsilon == pytest.approx(10.0 ** (-order), rel=1e-14)


def test_compute_entry_points_fail_loudly_without_device_state():
    """A host-only handle must refuse every compute call (no CPU fallback exists)."""
    import ferreus_rbf_rs_amd as F
    pts = np.random.default_rng(0).random((500, 3))
    t = F.FmmTree(pts, 5, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
    w = np.ones((500, 1))
    with pytest.raises(RuntimeError):
        t.set_weights(w)
    with pytest.raises(RuntimeError):
        t.evaluate(w, pts)
    with pytest.raises(RuntimeError):
        t.fast_matrix_vector_product(np.ones(500))


def test_bad_arguments_are_reported_not_thrown():
    import ferreus_rbf_rs_amd as F
    kp = F.KernelParams(F.FmmKernelType.LinearRbf)
    with pytest.raises(ValueError):          # d = 4: "Unsupported number of dimensions" (bbfmm.rs:293-298)
        F.FmmTree(np.zeros((10, 4)), 5, kp, True, True, host_only=True)
    with pytest.raises(ValueError):
        F.FmmTree(np.random.rand(10, 3), 1, kp, True, True, host_only=True)
    with pytest.raises(TypeError):           # numpy_to_matref accepts float64 only
        F.FmmTree(np.zeros((10, 3), dtype=np.float32), 5, kp, True, True, host_only=True)


# ---- the Rust shim (integration/ferreus_rbf_utils_hip) binds the same symbols with the same signatures
RUST_SHIM = os.path.join(ROOT, "integration", "ferreus_rbf_utils_hip", "src", "lib.rs")
C_TO_RUST = {
    "const double *": "*const f64", "double *": "*mut f64", "double": "f64", "int64_t": "i64", "int32_t": "i32",
    "uint32_t": "u32", "int": "c_int", "const int64_t *": "*const i64", "int64_t *": "*mut i64",
    "bbfmm_handle *": "*mut BbfmmHandle", "const bbfmm_handle *": "*const BbfmmHandle",
    "bbfmm_handle **": "*mut *mut BbfmmHandle", "const bbfmm_params *": "*const BbfmmParams",
    "const char *": "*const c_char", "void": "",
}


def _header_prototypes():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\n([A-Za-z_][\w \*]*?)\b(bbfmm_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        types = []
        for a in [x.strip() for x in args.split(",") if x.strip() and x.strip() != "void"]:
            mm = re.match(r"(.*?)(\w+)$", a)                    # split off the parameter name
            t = mm.group(1).strip()
            types.append(re.sub(r"\s*\*", " *", t).replace("* *", "**").strip())
        protos[name] = (ret, types)
    return protos


def test_rust_shim_extern_block_matches_the_header():
    src = open(RUST_SHIM).read()
    block = re.search(r'unsafe extern "C" \{(.*?)\n\}', src, re.S).group(1)
    protos = _header_prototypes()
    fns = re.findall(r"fn (bbfmm_\w+)\s*\((.*?)\)\s*(?:->\s*([\w\*: ]+))?;", block, re.S)
    assert len(fns) >= 11
    for name, args, ret in fns:
        assert name in protos, f"{name} bound by the shim but not declared in the header"
        c_ret, c_types = protos[name]
        r_types = [" ".join(a.split(":", 1)[1].split()) for a in " ".join(args.split()).split(",") if ":" in a]
        assert [C_TO_RUST[t] for t in c_types] == r_types, (name, c_types, r_types)
        assert C_TO_RUST[c_ret] == (ret or "").strip(), (name, c_ret, ret)
    bound = {f[0] for f in fns}
    # every method of the reference's FmmTree (utils.rs:392-493) has its entry point bound and a method defined
    for sym in ("bbfmm_create", "bbfmm_set_weights", "bbfmm_set_local_coefficients", "bbfmm_evaluate",
                "bbfmm_evaluate_with_gradients", "bbfmm_evaluate_leaves", "bbfmm_evaluate_leaves_with_gradients",
                "bbfmm_source_points", "bbfmm_destroy", "bbfmm_last_error"):
        assert sym in bound
    for method in ("new", "set_weights", "set_local_coefficients", "evaluate", "evaluate_with_gradients",
                   "evaluate_leaves", "evaluate_leaves_with_gradients", "source_points"):
        assert re.search(r"pub fn %s\s*[(<]" % method, src), method
    # the struct layout of bbfmm_params
    assert re.search(r"struct BbfmmParams \{\s*max_points_per_cell: i64,\s*compression_type: i32,\s*epsilon: f64,\s*"
                     r"eval_chunk_size: i64,\s*\}", src)
