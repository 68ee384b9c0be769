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
