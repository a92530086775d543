import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from types import SimpleNamespace
from conftest import load_golden, t
import test_gpu_data as T
from deepviewagg_amd.core.multimodal.image import ImageData
from deepviewagg_amd.modules.multimodal import UnimodalBranch, BimodalCSRPool, GroupBimodalCSRPool, BimodalFusion
from deepviewagg_amd.modules.multimodal.modules import MultimodalBlockDown, multimodal_input
from deepviewagg_amd.modules.SparseConv3d import ResNetDown, ResNetUp
DEV = "cuda:0"
g = load_golden("branch_nearest")
n_set = int(g["n_settings"])
torch.manual_seed(0)
Conv = T.Conv if hasattr(T, "Conv") else None

def fresh_inputs():
    xs = [t(g[f"s{i}_x_img"], DEV).requires_grad_() for i in range(n_set)]
    sds = [T.make_image_data(g, f"s{i}_", xs[i], g[f"s{i}_ref_size"], DEV) for i in range(n_set)]
    x_3d = t(g["x_3d"]); n = x_3d.shape[0]
    side = int(np.ceil(n ** (1 / 3))) + 1
    lin = torch.randperm(side ** 3, generator=torch.Generator().manual_seed(7))[:n]
    coords = torch.stack([lin % side, (lin // side) % side, lin // (side * side)], 1).int()
    class _Batch(SimpleNamespace):
        def to(self, device): return self
    data = _Batch(x=x_3d.requires_grad_(), coords=coords, batch=torch.zeros(n, dtype=torch.long), pos=None,
                  modalities={"image": ImageData(sds)})
    return xs, data

def branch(c3d):
    pool = GroupBimodalCSRPool(in_map=8, in_mod=8, num_groups=4, use_num=True)
    return UnimodalBranch(T.Conv(6, 8), BimodalCSRPool(mode="max"), pool, BimodalFusion(mode="concatenation"))
nc = int(g["x_3d"].shape[1])
enc = MultimodalBlockDown(ResNetDown(down_conv_nn=[nc, 16], N=1), ResNetDown(down_conv_nn=[24, 32], stride=1, kernel_size=3, N=1), image=branch(16)).to(DEV).eval()
dec = ResNetUp(up_conv_nn=[32, nc, 12], N=1).to(DEV).eval()

CAP = {}
def hook(name):
    def f(mod, inp, out):
        from deepviewagg_amd import ops
        o = out
        if isinstance(o, ops.GatheredFeatures):
            o = o.materialize()
        if not isinstance(o, torch.Tensor):
            o = getattr(o, "F", None)
        if isinstance(o, torch.Tensor):
            CAP.setdefault(name, []).append(o.detach().clone())
        ins = []
        for a in inp:
            if isinstance(a, ops.GatheredFeatures):
                ins.append(a.materialize().detach().clone())
            elif isinstance(a, torch.Tensor):
                ins.append(a.detach().clone())
        CAP.setdefault(name + "_in", []).append(ins)
    return f
for nm in ("conv", "atomic_pool", "view_pool", "fusion"):
    getattr(enc.image, nm).register_forward_hook(hook(nm))

def stages():
    xs, data = fresh_inputs()
    mm = multimodal_input(data, DEV)
    skip = mm["x_3d"]
    res = []
    with torch.enable_grad():
        mm = MultimodalBlockDown.forward_3d_block_down(mm, enc.block_1); res.append(mm["x_3d"].F.clone())
        mm = enc.image(mm, "image"); res.append(mm["x_3d"].F.clone())
        mm = MultimodalBlockDown.forward_3d_block_down(mm, enc.block_2); res.append(mm["x_3d"].F.clone())
        y = dec(mm["x_3d"], skip); res.append(y.F.clone())
    return res
def full():
    xs, data = fresh_inputs()
    mm = multimodal_input(data, DEV)
    skip = mm["x_3d"]
    out = enc(mm)
    return dec(out["x_3d"], skip).F.detach().clone()
# like the test: one training-mode step with gradients first
enc.train(), dec.train()
xs, data = fresh_inputs(); mm = multimodal_input(data, DEV); skip = mm["x_3d"]
y = dec(enc(mm)["x_3d"], skip)
torch.autograd.grad(y.F.square().mean(), xs + [data.x] + list(enc.parameters()) + list(dec.parameters()), allow_unused=True)
enc.eval(), dec.eval()
CAP.clear()
ref = stages()
CAPREF = None
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    CAP.clear()
    cur = stages()
    capcur = {k: v for k, v in CAP.items()}
    if it == 0:
        CAPREF = capcur
    y2 = None
    if it > 0 and not all(torch.equal(a, b) for a, b in zip(ref, cur)) and bad < 3:
        for k in capcur:
            if k.endswith("_in"):
                for j, (la, lb) in enumerate(zip(CAPREF[k], capcur[k])):
                    for q, (a, b) in enumerate(zip(la, lb)):
                        if a.shape == b.shape and not torch.equal(a, b):
                            print("   input differs:", k, j, q, float((a.float() - b.float()).abs().max()))
            else:
                for j, (a, b) in enumerate(zip(CAPREF[k], capcur[k])):
                    if not torch.equal(a, b):
                        print("   output differs:", k, j, float((a.float() - b.float()).abs().max()))
    eq = [torch.equal(a, b) for a, b in zip(ref, cur)]
    if not all(eq):
        bad += 1
        d = [float((a - b).abs().max()) for a, b in zip(ref, cur)]
        print("iter", it, "equal per stage [block1, branch, block2, decoder]:", eq, d)
print("mismatching runs:", bad)
