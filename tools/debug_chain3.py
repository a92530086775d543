"""Per-parameter error table of the fp32-class chain against an fp64 evaluation of the oracle module."""
import copy
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import pooling_oracle as O
from deepviewagg_amd import fused_chain_f32, fused_deepset
from deepviewagg_amd.modules.multimodal import pooling as P

DEV = "cuda:0"
gen = torch.Generator().manual_seed(15)
N, G, use_num = int(sys.argv[1]) if len(sys.argv) > 1 else 6000, 4, True
train = (sys.argv[2] != "eval") if len(sys.argv) > 2 else True
sizes = torch.randint(0, 10, (N,), generator=gen)
sizes[N // 3:N // 3 + 20] = 150
csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
V = int(csr[-1])
ref = O.DeepSetFeat(8, 32, use_num=use_num)
lin = torch.nn.Linear(32, G)
with torch.no_grad():
    for p in list(ref.parameters()) + list(lin.parameters()):
        p.copy_(torch.randn(p.shape, generator=gen) * 0.4)
ref.train(train)
dev = P.DeepSetFeat(8, 32, use_num=use_num)
dev.load_state_dict(ref.state_dict(), strict=True)
e_map, e_lin = dev.to(DEV).train(train), copy.deepcopy(lin).to(DEV)
x_map = torch.rand(V, 8, generator=gen)
w = torch.randn(V, G, generator=gen)
ref64, lin64 = copy.deepcopy(ref).double(), copy.deepcopy(lin).double()
s64 = lin64(ref64(x_map.double(), csr))
g64 = torch.autograd.grad((s64 * w.double()).sum(), list(ref64.parameters()) + list(lin64.parameters()))
e_map2, e_lin2 = copy.deepcopy(e_map), copy.deepcopy(e_lin)


def run(fn, em, el):
    s = fn(em, el, x_map.to(DEV), csr.to(DEV))
    gr = torch.autograd.grad((s * w.to(DEV)).sum(), list(em.parameters()) + list(el.parameters()))
    return s, gr


s3, g3 = run(fused_chain_f32.chain_scores, e_map, e_lin)
s1, g1 = run(fused_deepset.deepset_linear, e_map2, e_lin2)
print("V", V, "scores: chain3 max err %.3e  stored %.3e  (max |s| %.3f)" % (
    float((s3.double().cpu() - s64).abs().max()), float((s1.double().cpu() - s64).abs().max()), float(s64.abs().max())))
names = [n for n, _ in ref.named_parameters()] + ["Ws", "bs"]
for n, a, a1, b in zip(names, g3, g1, g64):
    scale = float(b.abs().max()) + 1e-12
    print("%-28s chain3 %.3e   stored %.3e   scale %.3e" % (
        n, float((a.double().cpu() - b).abs().max()) / scale, float((a1.double().cpu() - b).abs().max()) / scale, scale))
