import numpy as np, time, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ferreus_rbf_rs_amd as F
n=10_000_000
pts=np.random.default_rng(42).random((n,3))
t0=time.time(); t=F.FmmTree(pts,7,F.KernelParams(F.FmmKernelType.LinearRbf),True,True); print("total", time.time()-t0)
