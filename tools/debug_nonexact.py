#!/usr/bin/env python
"""Debug: E_mod parameter gradients of the lazy non-exact route, the materialised route and the fp32 oracle."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepviewagg_amd.modules.multimodal import pooling as P
from deepviewagg_amd import ops
from oracle import pooling_oracle as O
DEV = torch.device("cuda:0")
gen = torch.Generator().manual_seed(11)
B, C, H, W, N = 4, 64, 16, 24, 1500
k = torch.randint(0, 7, (N,), generator=gen)
csr = torch.cat([torch.zeros(1, dtype=torch.long), k.cumsum(0)])
V = int(csr[-1])
atoms = torch.randint(1, 10, (V,), generator=gen)
atom_ptr = torch.cat([torch.zeros(1, dtype=torch.long), atoms.cumsum(0)])
Pn = int(atom_ptr[-1])
images = torch.randint(0, B, (V,), generator=gen)
pixels = torch.stack([torch.randint(0, W, (Pn,), generator=gen), torch.randint(0, H, (Pn,), generator=gen)], 1)
x0 = torch.randn(B, C, H, W, generator=gen)
x_map = torch.rand(V, 8, generator=gen)
wout = torch.randn(N, C, generator=gen)
torch.manual_seed(5)
view_pool = P.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True).to(DEV).train()
sd = {k_: v.clone() for k_, v in view_pool.state_dict().items()}
atomic = P.BimodalCSRPool(mode='max')

def run(lazy_nonexact):
    ops.LAZY_NONEXACT = lazy_nonexact
    view_pool.load_state_dict(sd)
    for p in view_pool.parameters():
        p.grad = None
    x = x0.to(DEV).to(memory_format=torch.channels_last).requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        xb = x.to(torch.bfloat16)
        lazy = ops.lazy_gather_nearest_mapping(xb, images.to(DEV), atom_ptr.to(DEV), pixels.to(DEV).to(torch.int16), 1.0, exact=False)
        pooled = atomic(None, lazy, None, atom_ptr.to(DEV))
        out = view_pool(None, pooled, x_map.to(DEV), csr.to(DEV))
    (out.float() * wout.to(DEV)).sum().backward()
    return out.detach().float().cpu(), x.grad.float().cpu(), {n: p.grad.float().cpu().clone() for n, p in view_pool.named_parameters()}

o_l, gx_l, g_l = run(True)
o_m, gx_m, g_m = run(False)
# oracle fp32
om = O.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True).train()
om.load_state_dict({k_: v.cpu() for k_, v in sd.items()})
xr = x0.clone().requires_grad_()
img_at = images.repeat_interleave(atoms)
xm = O.gather_nearest(xr, img_at, pixels)
xm = O.segment_csr(xm, atom_ptr, 'max')
oo = om(None, xm, x_map, csr)
(oo * wout).sum().backward()
g_o = {n: p.grad.clone() for n, p in om.named_parameters()}
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-20))
print("out lazy/mat vs oracle", rel(o_l, oo.detach()), rel(o_m, oo.detach()))
print("gx  lazy/mat vs oracle", rel(gx_l, xr.grad), rel(gx_m, xr.grad))
for n in g_o:
    print(f"{n:45s} lazy {rel(g_l[n], g_o[n]):.3f} mat {rel(g_m[n], g_o[n]):.3f}  norms {float(g_l[n].norm()):.3g} {float(g_m[n].norm()):.3g} {float(g_o[n].norm()):.3g}")
