"""Debug: per-view score gradients of the chain backward vs the bf16 emulation (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_chain as T
from oracle import pooling_oracle as O
from deepviewagg_amd import ops, fused_chain, fused_chain_bwd

DEV = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "full32"
cfg = {"full32": (T.full32, 2048, 64, 4, True, True, True), "evalns": (T.ragged_long, 2000, 64, 4, False, True, False)}[which]
sizes_fn, N, C, G, train, gating, scaling = cfg
case = T.make_case(13, N, C, sizes_fn)
gen = case["gen"]
V, csr = case["V"], case["csr"]
R = 777
rows = (torch.randn(R, C, generator=gen)).bfloat16()
row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32)
ref, m = T.build(case, G, train, gating=gating, scaling=scaling)

# emulation with access to compat
import torch.nn.functional as F
E = ref.E_map
idx = O.dense_index(csr)
_bf = T._bf
def bn_act(blk, z): return F.leaky_relu(blk[1](z), 0.2)
x_map = case["x_map"]
a1 = _bf(bn_act(E.mlp_elt_1[0], x_map @ _bf(E.mlp_elt_1[0][0].weight).t()))
a2f = bn_act(E.mlp_elt_1[1], a1 @ _bf(E.mlp_elt_1[1][0].weight).t())
x_set = O.segment_csr(a2f, csr, 'max')
set_num = torch.sqrt(1 / (csr[1:] - csr[:-1] + 1e-3))
x_set = torch.cat((x_set, set_num.view(-1, 1).float()), dim=1)
s = E.mlp_set(x_set)
Wc = E.mlp_elt_2[0][0].weight
u = s @ Wc[:, 32:].t()
a5 = _bf(bn_act(E.mlp_elt_2[0], _bf(a2f) @ _bf(Wc[:, :32]).t() + u[idx]))
a6 = _bf(bn_act(E.mlp_elt_2[1], a5 @ _bf(E.mlp_elt_2[1][0].weight).t()))
compat = a6 @ _bf(ref.E_score.weight).t() + ref.E_score.bias
compat.retain_grad()
rows_ref = rows.float().requires_grad_()
out_ref, att, gate = O.attention_tail(rows_ref[row_idx.long()], compat, csr, ref.G, ref.num_groups, ref.out_mod, ref.group_scaling)
(out_ref * case["w"]).sum().backward()
dc_ref = compat.grad

# device: hook the backward to capture dc
captured = {}
lib_check = fused_chain_bwd.check
import deepviewagg_amd._lib as L
rows_d = rows.to(DEV).requires_grad_()
gf = ops.GatheredFeatures(rows_d, row_idx.to(DEV), None, True, None)
fused_chain.FORCE = True
orig_empty = torch.empty
def spy_empty(*a, **k):
    t = orig_empty(*a, **k)
    if len(a) == 1 and isinstance(a[0], tuple) and a[0] == (V, 4):
        captured["dc"] = t
    return t
out = fused_chain.chain_pool(m, gf, x_map.to(DEV), csr.to(DEV))
torch.empty = spy_empty
try:
    (out.float() * case["w"].to(DEV)).sum().backward()
finally:
    torch.empty = orig_empty
dc = captured["dc"].cpu()
print("compat fwd check: out rel", float((out.float().cpu() - out_ref).norm() / out_ref.norm()))
d = (dc - dc_ref)
print("dc rel L2", float(d.norm() / dc_ref.norm()))
pv = d.abs().max(1).values
top = torch.topk(pv, 12)
vp = O.dense_index(csr)
for v, e in zip(top.indices.tolist(), top.values.tolist()):
    p = int(vp[v]); n = int(csr[p + 1] - csr[p])
    print(f"view {v} point {p} n={n} pos_in_point={v - int(csr[p])} err={e:.4e} dc={dc[v].tolist()} ref={dc_ref[v].tolist()}")
# per-point aggregate
err_pt = torch.zeros(N).index_add_(0, vp, pv)
print("points with large error:", int((err_pt > 1e-3 * float(dc_ref.abs().max())).sum()), "of", N)
sizes = csr[1:] - csr[:-1]
bad = err_pt > 1e-2 * float(dc_ref.abs().max())
print("sizes of bad points (hist):", torch.bincount(sizes[bad].clamp(max=40))[:41].tolist() if bad.any() else None)
