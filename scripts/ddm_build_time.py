import numpy as np, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ferreus_rbf_rs_amd as F
from ferreus_rbf_rs_amd.ddm import DDMParams, InterpolantSettings, SchwarzPreconditioner
n=int(sys.argv[1]); ct=int(sys.argv[2])
pts=np.random.default_rng(42).random((n,3))
tree=F.FmmTree(pts,7,F.KernelParams(F.KernelType(0)),True,True)
t0=time.time(); pre=SchwarzPreconditioner(tree,pts,InterpolantSettings(0,3),DDMParams(coarse_threshold=ct)); print("total", round(time.time()-t0,2))
