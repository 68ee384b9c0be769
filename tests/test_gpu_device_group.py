"""One handle, several devices, one process (bbfmm_create_on_devices / FERREUS_BBFMM_DEVICES): the drop-in's multi-GPU
path.  The reference keeps ONE FmmTree behind a Mutex and calls `set_weights(w)` + `evaluate(w, sources)`
(ferreus_rbf/src/rbf.rs:85-133, 1357-1364; ferreus_rbf_utils/src/utils.rs:392-449), so the partition lives behind those
unchanged methods.  A one-GPU box rehearses it with G logical parts on device 0 (peer copies become device copies; the
upward plans, the slot exchange, the restricted downward passes, the per-part copies back and the host's inverse
permutation are the N-device code).  Bars: the group = the one-part handle to 1e-12, = the oracle at 1e-11."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ferreus_rbf_rs_amd as F
from conftest import clustered_points, inject_product_operators, relerr
from oracle import bbfmm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-11
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def case():
    rng = np.random.default_rng(606)
    pts = np.vstack([rng.random((110000, 3)), clustered_points(rng, 12000, 3)])   # mixed levels: W / X lists live
    kp = F.KernelParams(F.FmmKernelType.LinearRbf)
    one = F.FmmTree(pts, 7, kp, True, True)
    ref = O.FmmTree(pts, 7, 0, True, True)
    inject_product_operators(one, ref)
    assert one.stats().n_w > 0 and one.device_count() == 1
    return rng, pts, kp, one, ref


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_unchanged_caller_on_a_group_equals_one_part_and_the_oracle(case, parts):
    rng, pts, kp, one, ref = case
    n = len(pts)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0] * parts)
    assert g.device_count() == parts and [g.part_device(i) for i in range(parts)] == [0] * parts
    b = g.group_bounds()
    assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) > 0)
    w = np.asfortranarray(rng.standard_normal((n + 4, 1)))       # N + basis_size rows, as the solver's vectors (rbf.rs:1344)
    g.set_weights(w)
    y = g.evaluate(w, pts.copy())                                # select_mat_rows(source_points, all rows): a fresh copy
    assert g.last_evaluate_at_sources() == 1
    one.set_weights(w)
    y1 = one.evaluate(w, pts)
    assert relerr(y, y1) < 1e-12
    ref.set_weights(w[:n])
    assert relerr(y, ref.evaluate(w[:n], pts)) < TOL
    # a second product on the same handle (other weights), then the patched entry point with nugget and polynomial tail
    w2 = np.asfortranarray(rng.standard_normal((n + 4, 1)))
    g.set_weights(w2)
    one.set_weights(w2)
    assert relerr(g.evaluate(w2, pts), one.evaluate(w2, pts)) < 1e-12
    poly = np.asfortranarray(np.hstack([np.ones((n, 1)), pts]))
    ym = g.fast_matrix_vector_product(w[:, 0].copy(), basis_size=4, polynomial_matrix=poly, nugget=0.25)
    ym1 = one.fast_matrix_vector_product(w[:, 0].copy(), basis_size=4, polynomial_matrix=poly, nugget=0.25)
    assert relerr(ym, ym1) < 1e-12 and np.all(ym[n:] == 0.0)
    assert relerr(ym[:n], y1[:, 0] + 0.25 * w[:n, 0] + poly @ w[n:, 0]) < 1e-12


def test_two_rhs_and_a_second_evaluate_behind_one_set_weights(case):
    rng, pts, kp, one, ref = case
    n = len(pts)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0])
    w = np.asfortranarray(rng.standard_normal((n, 2)))
    g.set_weights(w)
    y = g.evaluate(w, pts)
    assert g.last_evaluate_at_sources() == 1 and y.shape == (n, 2)
    ref.set_weights(w)
    assert relerr(y, ref.evaluate(w, pts)) < TOL
    y2 = g.evaluate(w, pts)                                       # the upward pass is run again from the staged weights
    assert g.last_evaluate_at_sources() == 1 and relerr(y2, y) < 1e-12
    one.set_weights(w)
    assert relerr(y, one.evaluate(w, pts)) < 1e-12


def test_matvec_partial_row_sets_go_to_the_parts_that_own_them_and_what_remains_to_the_first_device(case):
    rng, pts, kp, one, ref = case
    n = len(pts)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0])
    w = np.asfortranarray(rng.standard_normal((n, 1)))
    # matvec_partial (rbf.rs:119-133), patched caller: the rows of the set dealt to the parts that own them, nugget and
    # polynomial tail on the host; an unsorted set with a repeated row; the empty set
    idx = np.sort(rng.choice(n, n // 7, replace=False))
    poly = np.asfortranarray(np.hstack([np.ones((n, 1)), pts]))
    w4 = np.concatenate([w[:, 0], rng.standard_normal(4)])
    yp = g.fast_matrix_vector_product(w4, basis_size=4, target_indices=idx, polynomial_matrix=poly, nugget=0.5)
    yp1 = one.fast_matrix_vector_product(w4, basis_size=4, target_indices=idx, polynomial_matrix=poly, nugget=0.5)
    assert g.last_evaluate_path() == 2 and relerr(yp, yp1) < 1e-12 and np.count_nonzero(yp) <= len(idx)
    ref.set_weights(w)
    assert relerr(yp[idx], ref.evaluate(w, pts[idx])[:, 0] + 0.5 * w[idx, 0] + poly[idx] @ w4[n:]) < TOL
    assert relerr(g.fast_matrix_vector_product(w4, basis_size=4, target_indices=idx, polynomial_matrix=poly, nugget=0.5), yp) < 1e-12
    idx2 = rng.permutation(idx)[: n // 30]
    idx2[5] = idx2[9]
    yq = g.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx2)
    assert relerr(yq, one.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx2)) < 1e-12
    assert not g.fast_matrix_vector_product(w[:, 0].copy(), target_indices=np.zeros(0, dtype=np.int64)).any()
    with pytest.raises(ValueError, match="out of range"):
        g.fast_matrix_vector_product(w[:, 0].copy(), target_indices=np.array([0, n]))
    # the unchanged caller of it: set_weights + evaluate at rows of the sources -- recognised, partitioned from the second
    # sighting of the set on (the first one is served by the first device, as on one device)
    g.set_weights(w)
    assert relerr(g.evaluate(w, pts[idx]), yp1[idx, None] - 0.5 * w[idx] - (poly[idx] @ w4[n:])[:, None]) < 1e-11
    assert g.last_evaluate_path() == 2                           # (a set the patched call has shown before)
    idx3 = np.sort(rng.choice(n, n // 9, replace=False))          # a set never seen
    g.set_weights(w)
    ys = g.evaluate(w, pts[idx3])
    assert g.last_evaluate_path() == 0
    g.set_weights(w)
    ys2 = g.evaluate(w, pts[idx3])
    assert g.last_evaluate_path() == 2
    one.set_weights(w)
    y_one = one.evaluate(w, pts[idx3])
    assert relerr(ys, y_one) < 1e-12 and relerr(ys2, y_one) < 1e-12
    ys3 = g.evaluate(w, pts[idx3])                               # behind the same set_weights: the upward pass is run again
    assert g.last_evaluate_path() == 2 and relerr(ys3, y_one) < 1e-12
    # the group's own path still works afterwards
    y = g.evaluate(w, pts)
    assert g.last_evaluate_at_sources() == 1
    y1 = one.evaluate(w, pts)
    assert relerr(y, y1) < 1e-12
    # arbitrary targets, gradients, and the reference's mixture (other weights in evaluate than in set_weights)
    x = rng.random((5000, 3))
    g.set_weights(w)
    assert relerr(g.evaluate(w, x), one.evaluate(w, x)) < 1e-12
    v, gr = g.evaluate_with_gradients(w, x)
    v1, gr1 = one.evaluate_with_gradients(w, x)
    assert relerr(v, v1) < 1e-12 and relerr(gr, gr1) < 1e-10
    w2 = np.asfortranarray(rng.standard_normal((n, 1)))
    g.set_weights(w)
    one.set_weights(w)
    ymix = g.evaluate(w2, pts)                                    # old multipoles, new near field (bbfmm.rs:444-507)
    assert g.last_evaluate_at_sources() == 0
    assert relerr(ymix, one.evaluate(w2, pts)) < 1e-12
    # Leaves mode on a group handle
    g.set_weights(w)
    g.set_local_coefficients(w)
    one.set_weights(w)
    one.set_local_coefficients(w)
    assert relerr(g.evaluate_leaves(w, x), one.evaluate_leaves(w, x)) < 1e-12
    # and a partition of the caller's own is refused: the group owns it
    with pytest.raises(ValueError, match="device group"):
        g.set_partition(0, 2)


def test_device_resident_vectors_on_a_group(case):
    import torch
    rng, pts, kp, one, ref = case
    n = len(pts)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0, 0, 0])
    for k in (1, 3):
        w = torch.from_numpy(np.ascontiguousarray(rng.standard_normal((k, n)))).cuda()
        out = torch.zeros((k, n), dtype=torch.float64, device="cuda")
        out1 = torch.zeros_like(out)
        torch.cuda.synchronize()
        g.matvec_device(w.data_ptr(), n, k, out.data_ptr(), n, sync=True)
        one.matvec_device(w.data_ptr(), n, k, out1.data_ptr(), n, sync=True)
        assert relerr(out.cpu().numpy(), out1.cpu().numpy()) < 1e-12
    # A call the first device serves alone right behind a device-resident product, with no set_weights in between (what a
    # plain handle allows: bbfmm_matvec_device leaves the multipoles of its weights behind).  The partitioned upward pass left
    # the first device with the partial sums of its own subtree: it completes them from its copy of the device weights.
    w1 = torch.from_numpy(np.ascontiguousarray(rng.standard_normal((1, n)))).cuda()
    o1, o2 = torch.zeros((1, n), dtype=torch.float64, device="cuda"), torch.zeros((1, n), dtype=torch.float64, device="cuda")
    g.matvec_device(w1.data_ptr(), n, 1, o1.data_ptr(), n, sync=True)
    one.matvec_device(w1.data_ptr(), n, 1, o2.data_ptr(), n, sync=True)
    x = pts[rng.choice(n, 300, replace=False)] * 0.999 + 0.0005 * rng.random((300, pts.shape[1]))
    w1h = np.asfortranarray(w1.cpu().numpy().T)
    assert g.last_evaluate_path() == 1
    y_g, y_1 = g.evaluate(w1h, x), one.evaluate(w1h, x)
    assert g.last_evaluate_path() == 0 and relerr(y_g, y_1) < 1e-12
    # and the product after that is whole again
    g.matvec_device(w1.data_ptr(), n, 1, o1.data_ptr(), n, sync=True)
    assert relerr(o1.cpu().numpy(), o2.cpu().numpy()) < 1e-12
    # a host-buffer product right behind the device-resident one (the staged weights were replaced)
    wh = np.asfortranarray(rng.standard_normal((n, 1)))
    g.set_weights(wh)
    one.set_weights(wh)
    assert relerr(g.evaluate(wh, pts), one.evaluate(wh, pts)) < 1e-12


def test_the_environment_switch_reaches_the_unchanged_constructor():
    """FERREUS_BBFMM_DEVICES is read by bbfmm_create itself: what the Rust shim (and this binding without `devices`) calls."""
    code = r"""
import numpy as np, ferreus_rbf_rs_amd as F
rng = np.random.default_rng(1)
pts = rng.random((30000, 3)); w = rng.standard_normal((30000, 1))
t = F.FmmTree(pts, 5, F.KernelParams(F.FmmKernelType.ThinPlateSplineRbf), True, True)
assert t.device_count() == 3, t.device_count()
t.set_weights(w); y = t.evaluate(w, pts)
assert t.last_evaluate_at_sources() == 1
import os; os.environ.pop("FERREUS_BBFMM_DEVICES")
u = F.FmmTree(pts, 5, F.KernelParams(F.FmmKernelType.ThinPlateSplineRbf), True, True)
assert u.device_count() == 1
u.set_weights(w); y1 = u.evaluate(w, pts)
print("REL", float(np.abs(y - y1).max() / np.abs(y1).max()))
"""
    env = dict(os.environ, FERREUS_BBFMM_DEVICES="0,0,0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    rel = float(out.stdout.decode().split("REL")[1])
    assert rel < 1e-12


def test_deterministic_group_is_bitwise_reproducible(case):
    rng, pts, kp, one, ref = case
    n = len(pts)
    w = np.asfortranarray(rng.standard_normal((n, 1)))
    ys = []
    for _ in range(2):
        g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0], deterministic=True)
        g.set_weights(w)
        ys.append(g.evaluate(w, pts))
        del g
    assert np.array_equal(ys[0], ys[1])
    ref.set_weights(w)
    assert relerr(ys[0], ref.evaluate(w, pts)) < TOL


@pytest.mark.parametrize("d,adaptive,kind", [(2, True, F.FmmKernelType.ThinPlateSplineRbf), (3, False, F.FmmKernelType.CubicRbf),
                                             (1, True, F.FmmKernelType.LinearRbf)])
def test_other_dimensions_and_regular_trees_on_a_group(d, adaptive, kind):
    """2-D and 1-D clouds and a regular (non-adaptive) tree through a four-part group: no W / X lists in the regular tree, a
    shallow exchange prefix in 1-D -- the group equals the one-part handle whatever the tree looks like."""
    rng = np.random.default_rng(40 + d)
    n = 60000 if d > 1 else 20000
    pts = rng.random((n, d))
    kp = F.KernelParams(kind)
    one = F.FmmTree(pts, 6, kp, adaptive, True)
    g = F.FmmTree(pts, 6, kp, adaptive, True, devices=[0, 0, 0, 0])
    w = np.asfortranarray(rng.standard_normal((n, 2)))
    g.set_weights(w)
    one.set_weights(w)
    y, y1 = g.evaluate(w, pts), one.evaluate(w, pts)
    assert g.last_evaluate_at_sources() == 1 and relerr(y, y1) < 1e-12
    ym, ym1 = g.fast_matrix_vector_product(w[:, 0].copy()), one.fast_matrix_vector_product(w[:, 0].copy())
    assert relerr(ym, ym1) < 1e-12


def test_a_device_that_does_not_exist_is_refused_with_a_message():
    import torch
    nd = torch.cuda.device_count()
    with pytest.raises(ValueError, match="does not exist"):
        F.FmmTree(np.random.default_rng(0).random((2000, 3)), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, devices=[0, nd])
    t = F.FmmTree(np.random.default_rng(0).random((2000, 3)), 4, F.KernelParams(F.FmmKernelType.LinearRbf), True, True, devices=[0])
    assert t.device_count() == 1 and t.part_device(0) == 0          # one entry: a plain handle on that device


def test_the_solver_drivers_run_on_a_group_handle(case):
    """bbfmm_fgmres with bbfmm_rbf_system_apply (= bbfmm_fast_matrix_vector_product per iteration, rbf.rs:105-117) on a group
    handle: the same iterates as on one part (the matvecs agree to 1e-12, so do a few FGMRES iterations)."""
    from ferreus_rbf_rs_amd import solvers as S
    rng, pts, kp, one, ref = case
    n = len(pts)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0])
    rhs = rng.standard_normal(n)
    hist = {}
    for name, t in (("group", g), ("one", one)):
        op = S.RbfSystemOperator(t, 0, None, 0.0)
        x, h = S.fgmres(op, rhs, None, None, 1, 4, S.FittingAccuracy(1e-30))
        hist[name] = (x, [r for _, r in h])
    assert len(hist["group"][1]) == len(hist["one"][1]) == 4
    assert np.allclose(hist["group"][1], hist["one"][1], rtol=1e-9)
    assert relerr(hist["group"][0], hist["one"][0]) < 1e-8


def test_arbitrary_targets_are_sharded_over_the_parts(case, monkeypatch):
    """Evaluator use (many targets that are no sources; rbf.rs:677-690, 836-838): with the weights of set_weights every part
    completes its own multipoles and evaluates a contiguous share of the target rows -- values, gradients and Leaves mode
    equal the one-part handle's; a target outside the tree is reported with the smallest offending row of the WHOLE call."""
    rng, pts, kp, one, ref = case
    n = len(pts)
    monkeypatch.setenv("BBFMM_GROUP_SHARD_MIN", "2000")             # (read when the group is created; default 16384 rows per part)
    g = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0])
    w = np.asfortranarray(rng.standard_normal((n, 2)))
    x = rng.random((9001, 3))
    g.set_weights(w)
    one.set_weights(w)
    y, y1 = g.evaluate(w, x), one.evaluate(w, x)
    assert g.last_evaluate_path() == 3 and relerr(y, y1) < 1e-12
    ref.set_weights(w)
    assert relerr(y, ref.evaluate(w, x)) < TOL
    v, gr = g.evaluate_with_gradients(w, x)
    v1, gr1 = one.evaluate_with_gradients(w, x)
    assert g.last_evaluate_path() == 3 and relerr(v, v1) < 1e-12 and relerr(gr, gr1) < 1e-10
    # few targets: not worth sharding
    g.evaluate(w, x[:500])
    assert g.last_evaluate_path() == 0
    # the at-sources product still runs partitioned afterwards, and another sharded call after that
    assert relerr(g.evaluate(w, pts), one.evaluate(w, pts)) < 1e-12 and g.last_evaluate_path() == 1
    assert relerr(g.evaluate(w, x), y1) < 1e-12 and g.last_evaluate_path() == 3
    # Leaves mode over the group; weights resident (w = None) and passed again
    g.set_local_coefficients(w)
    one.set_local_coefficients(w)
    z1 = one.evaluate_leaves(w, x)
    assert relerr(g.evaluate_leaves(w, x), z1) < 1e-12 and g.last_evaluate_path() == 3
    assert relerr(g.evaluate_leaves(None, x), z1) < 1e-12
    zg, gg = g.evaluate_leaves_with_gradients(w, x)
    z1g, g1g = one.evaluate_leaves_with_gradients(w, x)
    assert relerr(zg, z1g) < 1e-12 and relerr(gg, g1g) < 1e-10
    # Leaves mode: an evaluate at the sources must not destroy the stored expansions (it is not partitioned then)
    assert relerr(g.evaluate(w, pts), one.evaluate(w, pts)) < 1e-12
    assert relerr(g.evaluate_leaves(w, x), z1) < 1e-12
    # a target outside the tree, in the last part's share and one in the second part's: the smaller row is reported
    xb = x.copy()
    xb[8000] = [5.0, 5.0, 5.0]
    xb[4000] = [7.0, 0.5, 0.5]
    g.set_weights(w)
    with pytest.raises(F.PointOutsideTree) as e:
        g.evaluate(w, xb)
    assert e.value.point_index == 4000
    # other weights than set_weights': the reference's mixture, on the first device
    w2 = np.asfortranarray(rng.standard_normal((n, 2)))
    g.set_weights(w)
    one.set_weights(w)
    assert relerr(g.evaluate(w2, x), one.evaluate(w2, x)) < 1e-12 and g.last_evaluate_path() == 0


@pytest.mark.parametrize("jobs", ["wave_per_small_leaf", "size_rule"])
@pytest.mark.parametrize("n", [2, 300, 5000])
def test_tiny_clouds_on_a_group(n, jobs, monkeypatch):
    """Fewer leaves than parts, trees of depth 0 or 1 (nothing to exchange), parts that own no row at all."""
    if jobs == "size_rule":   # (the library's own choice for trees this small: every leaf a workgroup job)
        monkeypatch.delenv("BBFMM_P2P_SYM_WAVE_MIN", raising=False)
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3))
    kp = F.KernelParams(F.FmmKernelType.LinearRbf)
    one = F.FmmTree(pts, 4, kp, True, True)
    g = F.FmmTree(pts, 4, kp, True, True, devices=[0, 0, 0, 0])
    w = np.asfortranarray(rng.standard_normal((n, 1)))
    g.set_weights(w)
    one.set_weights(w)
    y, y1 = g.evaluate(w, pts), one.evaluate(w, pts)
    assert relerr(y, y1) < 1e-12
    assert relerr(g.fast_matrix_vector_product(w[:, 0].copy()), one.fast_matrix_vector_product(w[:, 0].copy())) < 1e-12
    idx = np.arange(0, n, 2)
    assert relerr(g.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx),
                  one.fast_matrix_vector_product(w[:, 0].copy(), target_indices=idx)) < 1e-12


def test_extension_flags_on_a_group(case):
    """The labelled extensions (shared-basis M2L, small W leaves summed directly) keep working behind a group handle."""
    rng, pts, kp, one, ref = case
    n = len(pts)
    w = np.asfortranarray(rng.standard_normal((n, 1)))
    for kw in ({"m2l_shared_basis": True}, {"direct_small_w_leaves": True}):
        a = F.FmmTree(pts, 7, kp, True, True, **kw)
        b = F.FmmTree(pts, 7, kp, True, True, devices=[0, 0, 0], **kw)
        a.set_weights(w)
        b.set_weights(w)
        assert relerr(b.evaluate(w, pts), a.evaluate(w, pts)) < 1e-12, kw


@pytest.mark.timeout(900)
def test_random_call_sequences_on_a_group_equal_the_plain_handle():
    """tests/checks/group_sequence_fuzz.py in small: random sequences of the evaluator's calls (set_weights, evaluate at the
    sources / few / many targets / rows of the sources, other weights than set_weights', gradients, Leaves mode, the matvec
    entry points on host and device vectors) on a group of 2-5 logical parts and on a plain handle -- the same values at 1e-11
    or the same refusal (and the same offending row) after every call.  What a group keeps between calls must never show."""
    env = {k: v for k, v in os.environ.items() if not k.startswith("BBFMM_")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "group_sequence_fuzz.py"), "10", "5", "14"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    lines = p.stdout.decode().strip().splitlines()
    assert p.returncode == 0, "\n".join(l for l in lines if '"ok": false' in l)[:3000] + p.stderr.decode()[-1500:]
    assert '"failures": 0' in lines[-1]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("parts", [1, 3])
def test_handles_give_their_device_memory_back(parts):
    """scripts/group_lifecycle_check.py in small: handles (plain, and a group of logical parts) created, used through every kind
    of call and destroyed in a loop -- the device's free memory and the host's resident set do not drift.  (Found in round 6: a
    2-D copy to pageable host memory out of a per-call buffer kept that buffer from ever returning to the device on ROCm 7.2,
    4 MB per handle that had evaluated 80k targets; fmm_tree.cpp columns_to_host.)"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("BBFMM_")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "group_lifecycle_check.py"), "16", "60000", str(parts)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert p.returncode == 0, p.stdout.decode()[-1500:] + p.stderr.decode()[-1500:]
    assert '"ok": true' in p.stdout.decode()
