import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ferreus_rbf_rs_amd as F
n=1_000_000
rng=np.random.default_rng(5)
pts=rng.random((n,3))*2-1
w=rng.random((n,1))
tree=F.FmmTree(pts,7,F.KernelParams(F.FmmKernelType.LinearRbf),True,False,extents=[-2,-2,-2,2,2,2])
tree.set_weights(w); tree.set_local_coefficients(w)
m=2_000_000
x=rng.random((m,3))*2-1
tree.evaluate_leaves(w,x)
tree.set_profiling(True); tree.phase_ms(reset=True)
t0=time.perf_counter(); tree.evaluate_leaves(w,x); dt=time.perf_counter()-t0
print("total ms", dt*1e3, {k:round(v,2) for k,v in tree.phase_ms().items() if v>0.05})
xf=np.asfortranarray(x)
t0=time.perf_counter(); tree.evaluate_leaves(w,xf); print("fortran-ordered input ms", (time.perf_counter()-t0)*1e3)
t0=time.perf_counter(); leaves=tree.points_to_leaves(xf); print("points_to_leaves ms", (time.perf_counter()-t0)*1e3)
