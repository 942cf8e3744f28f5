import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import mrhyde_amd
nc, order = 16, 2
m = mrhyde_amd.mesh_structured(3, order, (nc,)*3)
blk = mrhyde_amd.Block(3, order, quadrature=4, workset_size=100)
blk.set_mesh(m["nodes"], m["lids"], m["offsets"], m["ndof"], m["boundary"])
blk.set_graph()
rowptr, colind = blk.get_graph()
blk.set_function("thermal source", ("sinprod", 12*np.pi**2, [2*np.pi]*3))
blk.set_function("thermal diffusion", 1.0)
nd = m["ndof"]; nnz = len(colind)
u = torch.rand(nd, dtype=torch.float64, device="cuda")*2-1
def asm(path, fill):
    res = torch.full((nd,), 3.0, dtype=torch.float64, device="cuda")
    vals = torch.full((nnz,), fill, dtype=torch.float64, device="cuda")
    blk.assemble_jacres(u, res, vals, path=path, overwrite=True)
    torch.cuda.synchronize()
    return res.cpu().numpy(), vals.cpu().numpy()
rf, vf = asm(mrhyde_amd.PATH_ROW_OWNER, -2.0)
rf2, vf2 = asm(mrhyde_amd.PATH_ROW_OWNER, 7.0)
rg, vg = asm(mrhyde_amd.PATH_ROW_GATHER, -2.0)
print("kind", blk.info("row_owner_kind"), "patterns", blk.info("block_patterns"))
d = np.abs(vf - vg); print("max diff", d.max(), "ref max", np.abs(vg).max())
d2 = np.abs(vf - vf2); print("fast vs fast(other fill) max", d2.max(), "count", (d2 > 0).sum())
bad = np.nonzero(d > 1e-13 * np.abs(vg).max())[0]
print("bad entries", len(bad))
rows = np.searchsorted(rowptr, bad, side='right') - 1
lens = rowptr[rows+1] - rowptr[rows]
pos = bad - rowptr[rows]
for k in range(min(40, len(bad))):
    print(rows[k], lens[k], pos[k], vf[bad[k]], vg[bad[k]], vf2[bad[k]])
import collections
print(collections.Counter(zip(lens.tolist(), pos.tolist())).most_common(30))
