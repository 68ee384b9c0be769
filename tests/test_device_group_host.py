"""Host-side behaviour of the one-handle-several-devices entry points (bbfmm_create_on_devices, FERREUS_BBFMM_DEVICES)
that needs no GPU: what is refused and with which message, what a host-only handle reports, and the bookkeeping the group's
parts rest on -- G subtree partitions of ONE tree whose ranges tile the sorted points and whose upward plans, walked with point
counts, add up to the whole upward pass (the reference passes being split: ferreus_bbfmm/src/bbfmm.rs:383-401, 666-772)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd import _lib as L
from test_partition_upward import _true_counts

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pts(n=4000, seed=5):
    return np.random.default_rng(seed).random((n, 3))


def test_a_device_list_on_a_host_only_handle_is_refused_with_a_message():
    with pytest.raises(ValueError, match="BBFMM_FLAG_HOST_ONLY"):
        F.FmmTree(_pts(), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True, devices=[0, 0])


def test_an_empty_device_list_is_a_bad_argument():
    lib = L.load()
    pts = np.asfortranarray(_pts())
    h = ctypes.c_void_p()
    dev = np.zeros(0, dtype=np.int32)
    rc = lib.bbfmm_create_on_devices(pts.ctypes.data, len(pts), 3, len(pts), 4, 0, 1.0, 1.0, 1, 1, None, None, 0,
                                     dev.ctypes.data, 0, ctypes.byref(h))
    assert rc == L.BAD_ARGUMENT and not h


def test_host_only_handles_ignore_the_environment_switch_and_report_one_part():
    code = r"""
import numpy as np, ferreus_rbf_rs_amd as F
t = F.FmmTree(np.random.default_rng(0).random((3000, 3)), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True)
assert t.device_count() == 1 and t.part_device(0) == -1 and t.part_device(1) == -1
print("OK")
"""
    env = dict(os.environ, FERREUS_BBFMM_DEVICES="0,0,0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0 and b"OK" in out.stdout, out.stderr.decode()[-2000:]


@pytest.mark.parametrize("text,needle", [("0,,1", "malformed"), ("zero", "malformed"), ("0,-1", "malformed"), ("0;1", "malformed")])
def test_a_malformed_environment_list_fails_loudly(text, needle):
    """bbfmm_create reads FERREUS_BBFMM_DEVICES itself (the reference's constructor has no argument for it): a list it cannot
    read is an error with a message, never a silent one-device handle."""
    code = r"""
import numpy as np, ferreus_rbf_rs_amd as F
try:
    F.FmmTree(np.random.default_rng(0).random((3000, 3)), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True)
except (ValueError, RuntimeError) as e:
    print("ERR", e)
"""
    env = dict(os.environ, FERREUS_BBFMM_DEVICES=text, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    assert b"ERR" in out.stdout and needle.encode() in out.stdout and b"FERREUS_BBFMM_DEVICES" in out.stdout


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_the_parts_of_a_group_tile_the_sorted_points_and_their_upward_plans_add_up(parts):
    """What DeviceGroup::init relies on, on host-only handles: every part cuts the same bounds, the owned rows are a disjoint
    cover, and the partial coarse multipoles (walked with point counts) summed over the parts are the whole upward pass."""
    rng = np.random.default_rng(9)
    pts = np.vstack([rng.random((30000, 3)), np.clip(rng.normal(size=(6000, 3)) * 0.05 + 0.5, 0, 0.999)])
    n = len(pts)
    trees = [F.FmmTree(pts, 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, host_only=True) for _ in range(parts)]
    bounds, rows, sums = None, [], None
    for g, t in enumerate(trees):
        t.set_partition(g, parts)
        b = t.partition_bounds()
        assert bounds is None or np.array_equal(b, bounds)
        bounds = b
        rows.append(t.partition_rows())
        assert len(rows[-1]) == b[g + 1] - b[g]
        counts, reads, info = t.debug_partition_upward_counts()
        cc = int(info[1])
        part_sum = np.where(counts[:cc] < 0, 0, counts[:cc])
        sums = part_sum if sums is None else sums + part_sum
        assert t.partition_coarse_count() == (cc * 64 if cc else 0)      # n_pad = 64 at order 4
    assert bounds[0] == 0 and bounds[-1] == n and np.all(np.diff(bounds) >= 0)
    assert np.array_equal(np.sort(np.concatenate(rows)), np.arange(n))
    truth, level = _true_counts(trees[0], 3)
    sel = (level[:len(sums)] >= 1)
    assert np.array_equal(sums[sel], truth[:len(sums)][sel])               # what the slot exchange delivers to every device
