"""round 3: the residual kernel K1 of config 2 alone (compute_jacobian = False: no K2 before it, no write drain)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, mrhyde_amd

class A: pass
a = A(); a.order = 2; a.ncell = int(os.environ.get("NCELL", "64")); a.mesh = "affine"; a.path = "auto"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
w = bench.setup_thermal(a, torch, mrhyde_amd, 0, 1, dev)
blk, u, res, vals = w["_blk"], w["_u"], w["_res"], w["_vals"]
for jac in (False, True):
    for _ in range(5):
        blk.assemble_jacres(u, res, vals, compute_jacobian=jac, overwrite=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        blk.assemble_jacres(u, res, vals, compute_jacobian=jac, overwrite=True)
    e1.record(); torch.cuda.synchronize()
    print("compute_jacobian=%s: %.1f us per assembly  (MHA_K1=%s MHA_K1_DBG=%s)" % (jac, e0.elapsed_time(e1) * 1e3 / 50, os.environ.get("MHA_K1"), os.environ.get("MHA_K1_DBG")))
