import numpy as np, sys, os
root=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root,"tests"))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
from oracle import ddm as D
from test_gpu_schwarz import _dense_partial
rng=np.random.default_rng(5)
for kid in (1,0):
    n=5856; prm=(1024,0.5,0.125,700)
    pts=rng.random((n,3))
    st=InterpolantSettings(kid,3); ost=D.InterpolantSettings(kid,3)
    tree=F.FmmTree(pts,9,F.KernelParams(F.KernelType(kid)),True,True)
    pre=SchwarzPreconditioner(tree,pts,st,DDMParams(*prm))
    levels=D.build_ddm_tree(pts,ost,D.DDMParams(*prm))
    A,P,partial=_dense_partial(pts,ost)
    tr,sc=D.cheb_cube_scaling_factors(pts); mono,ortho=D.orthonormal_poly(pts,ost,tr,sc)
    r=rng.standard_normal(n+ost.basis_size); r[n:]=0
    z=pre(r); zo=D.schwarz_preconditioner(r,levels,partial,ost,ortho)
    print(kid,'levels',[len(l.point_indices) for l in levels],'domain sizes',sorted({len(d.overlapping_point_indices) for d in levels[0].leaf_domains}),'rel diff %.2e'%(np.abs(z-zo).max()/np.abs(zo).max()))
    for lv in range(len(levels)):
        z1=pre.debug_level_solve(lv,r,True)
        # oracle single level
        if lv<len(levels)-1:
            s1=np.zeros_like(r)
            for dom in levels[lv].leaf_domains:
                coef,_=dom.solve(r[:,None])
                for local,(g,m) in enumerate(zip(dom.overlapping_point_indices,dom.internal_points_mask)):
                    if m: s1[g]=coef[local,0]
            s1[:n]-=ortho@(ortho.T@s1[:n]) if ost.basis_size else 0
        else:
            dom=levels[lv].leaf_domains[0]; coef,poly=dom.solve(r[:,None]); s1=np.zeros_like(r); s1[np.asarray(dom.overlapping_point_indices)]=coef[:,0]
            if poly is not None: s1[n:]=poly[:,0]
        print('   level',lv,'single-level rel diff %.2e'%(np.abs(z1-s1).max()/max(np.abs(s1).max(),1e-300)))
