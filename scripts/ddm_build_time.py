"""Setup time of the Schwarz preconditioner (decomposition + factorisation); BBFMM_VERBOSE=1 prints the stages.
args: points coarse_threshold [kernel_id]"""
import numpy as np, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
n = int(sys.argv[1]); ct = int(sys.argv[2]); kid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pts = np.random.default_rng(42).random((n, 3))
tree = F.FmmTree(pts, 7, F.KernelParams(F.KernelType(kid)), True, True)
t0 = time.time()
try:
    pre = SchwarzPreconditioner(tree, pts, InterpolantSettings(kid, 3), DDMParams(coarse_threshold=ct))
except Exception as e:                      # debug variants of the factorisation produce garbage on purpose
    print("create failed:", e)
print("total", round(time.time() - t0, 2))
