"""Debug: per-view scores / score gradients of the chain vs the bf16 emulation (GPU box).
usage: python tools/debug_chain.py <case>   cases: full32 | c32g2 | evalns"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_chain as T
from oracle import pooling_oracle as O
from oracle.chain_emulation import emulated_chain
from deepviewagg_amd import ops, fused_chain

DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "full32"
cfg = {"full32": (T.full32, 2048, 64, 4, True, True, True), "evalns": (T.ragged_long, 2000, 64, 4, False, True, False),
       "c32g2": (T.ragged_long, 1500, 32, 2, True, True, True), "c32g2e": (T.ragged_long, 1500, 32, 2, False, True, True),
       "c32g4": (T.ragged_long, 1500, 32, 4, True, True, True), "c64g2": (T.ragged_long, 1500, 64, 2, True, True, True)}[which]
sizes_fn, N, C, G, train, gating, scaling = cfg
case = T.make_case(13, N, C, sizes_fn)
gen = case["gen"]
V, csr = case["V"], case["csr"]
R = 777
rows = (torch.randn(R, C, generator=gen)).bfloat16()
row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
ref, m = T.build(case, G, train, gating=gating, scaling=scaling)
x_map = case["x_map"]

captured = {}
rows_d = rows.to(DEV).requires_grad_()
gf = ops.GatheredFeatures(rows_d, row_idx.to(DEV), None, True, None)
fused_chain.FORCE = True
out = fused_chain.chain_pool(m, gf, x_map.to(DEV), csr.to(DEV))
saved = out.grad_fn.saved_tensors
scores_dev = saved[17].cpu()
dev_invstd = {1: saved[12][1].cpu(), 2: saved[13][1].cpu(), 6: saved[15][1].cpu()}
orig_empty = torch.empty
def spy_empty(*a, **k):
    t = orig_empty(*a, **k)
    if len(a) == 1 and isinstance(a[0], tuple) and a[0] == (V, 4) and k.get("dtype") == torch.float32 and "dc" not in captured:
        captured["dc"] = t
    return t
torch.empty = spy_empty
try:
    dev_params = [p for n, p in m.named_parameters() if not n.startswith("E_mod")]
    g_dev = torch.autograd.grad((out.float() * case["w"].to(DEV)).sum(), [rows_d] + dev_params, allow_unused=True)
finally:
    torch.empty = orig_empty
dc = captured["dc"].cpu()[:, :G]

rows_ref = rows.float().requires_grad_()
out_ref, compat = emulated_chain(ref, rows_ref[row_idx.long()], x_map, csr, dev_invstd=dev_invstd, dev_scores=(scores_dev[:, :G] if os.environ.get("DEV_SCORES") else None), return_scores=True)
chain_params = [p for n, p in ref.named_parameters() if not n.startswith("E_mod")]
g_ref = torch.autograd.grad((out_ref * case["w"]).sum(), [rows_ref, compat] + chain_params, allow_unused=True)
dc_ref = g_ref[1]
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
print("out rel", rel(out.float().cpu(), out_ref.detach()))
print("scores rel", rel(scores_dev[:, :G], compat.detach()), " max abs", float((scores_dev[:, :G] - compat.detach()).abs().max()))
print("dc rel L2", rel(dc, dc_ref))
d = dc - dc_ref
pv = d.abs().max(1).values
vp = O.dense_index(csr)
sizes = csr[1:] - csr[:-1]
top = torch.topk(pv, 10)
for v, e in zip(top.indices.tolist(), top.values.tolist()):
    p = int(vp[v]); n = int(sizes[p])
    print(f"view {v} point {p} n={n} pos={v - int(csr[p])} err={e:.3e} dc={[round(x, 5) for x in dc[v].tolist()]} ref={[round(x, 5) for x in dc_ref[v].tolist()]}")
err_pt = torch.zeros(N).index_add_(0, vp, pv)
bad = err_pt > 1e-2 * float(dc_ref.abs().max())
print("points with large dc error:", int(bad.sum()), "of", N, " sizes hist:", torch.bincount(sizes[bad].clamp(max=101))[:102].nonzero().view(-1).tolist() if bad.any() else None)
# error restricted to points of <= 32 views
small = (sizes[vp] <= 32)
print("dc rel L2 on points <= 32 views:", rel(dc[small], dc_ref[small]), "  > 32 views:", rel(dc[~small], dc_ref[~small]) if (~small).any() else None)
names = ["rows"] + [n for n, _ in ref.named_parameters() if not n.startswith("E_mod")]
for n, a, b in zip(names, g_dev, [g_ref[0]] + list(g_ref[2:])):
    if b is not None and a is not None:
        print(f"  {n:44s} {rel(a.float().cpu(), b):.4f}")
