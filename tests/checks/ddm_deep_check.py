import numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tests"))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
from oracle import ddm as D
from test_gpu_schwarz import _dense_partial
rng=np.random.default_rng(5)
for (n,prm) in [(6000,(40,0.5,0.25,60)),(8000,(60,0.5,0.125,30))]:
    pts=rng.random((n,3)); kid=1
    st=InterpolantSettings(kid,3); ost=D.InterpolantSettings(kid,3)
    tree=F.FmmTree(pts,9,F.KernelParams(F.KernelType(kid)),True,True)
    pre=SchwarzPreconditioner(tree,pts,st,DDMParams(*prm))
    levels=D.build_ddm_tree(pts,ost,D.DDMParams(*prm))
    A,P,partial=_dense_partial(pts,ost)
    tr,sc=D.cheb_cube_scaling_factors(pts); mono,ortho=D.orthonormal_poly(pts,ost,tr,sc)
    r=rng.standard_normal(n+4); r[n:]=0
    z=pre(r); zo=D.schwarz_preconditioner(r,levels,partial,ost,ortho)
    print(n, prm, 'levels', pre.num_levels, [len(l.point_indices) for l in levels], 'rel diff', np.abs(z-zo).max()/np.abs(zo).max())
