#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python source.

TEST INFRASTRUCTURE — runnable only in the build container (needs /root/reference, which does not
exist on the GPU box).  Nothing in the product, the -m gpu tests, smoke() or bench.py runs this; they
read the committed .npz files.

How the reference is made importable: the seven third-party packages its
hot path imports and the image lacks (numba, torch_scatter, torch_geometric, pykeops, hydra,
omegaconf, torchsparse) are replaced by the tiny stand-ins in oracle/shims/ (ours, written for this
repo; numba.njit = identity so the @njit bodies run as plain NumPy).  PYTORCH_JIT=0 keeps
@torch.jit.script from scripting the stubs.

Float-width contract of the mapping build (DESIGN.md "float-width contract"): under numba any
float32_array (op) python_scalar promotes to float64, under NumPy 2 it does not.  To get numba-style
promotion where it decides pixel coordinates, the reference is run with np.pi replaced by
np.float64(np.pi) in its module namespace and with r_min / r_max / voxel / k_swell / d_swell passed
as np.float64.  Candidate order (KDTree order in the reference) is an explicit input: identity.

Usage:  PYTORCH_JIT=0 python oracle/gen_golden.py
"""
import os
import sys
import types

os.environ.setdefault("PYTORCH_JIT", "0")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "shims"), "/root/reference"]
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from torch_points3d.modules.multimodal import pooling as ref_pooling  # noqa: E402
from torch_points3d.modules.multimodal import modules as ref_modules  # noqa: E402
from torch_points3d.modules.multimodal import fusion as ref_fusion  # noqa: E402
from torch_points3d.core.multimodal import visibility as ref_vis  # noqa: E402
from torch_points3d.core.multimodal import image as ref_image  # noqa: E402
from torch_points3d.core.multimodal.csr import CSRData, CSRBatch  # noqa: E402
from torch_points3d.utils import multimodal as ref_mm  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def randomize(module, gen):
    """Non-trivial parameters AND BatchNorm buffers so that every term of the maths is exercised."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.5 + (1.0 if n.endswith('batch_norm.weight') else 0.0))
        for n, b in module.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            elif n.endswith('running_var'):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)


def state(module, prefix='sd/'):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


def random_csr(n_groups, max_size, gen, p_empty=0.2):
    sizes = torch.randint(1, max_size + 1, (n_groups,), generator=gen)
    sizes[torch.rand(n_groups, generator=gen) < p_empty] = 0
    return torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])


# ------------------------------------------------------------------------------------------------
def gen_softmax():
    print("softmax known answers (pooling.py:913-921)")
    src = torch.arange(15).float().view(-1, 1).repeat_interleave(2, dim=1)
    csr = torch.LongTensor([0, 5, 10, 15])
    save("softmax_known", src=src, csr=csr,
         out=ref_pooling.segment_softmax_csr(src, csr),
         out_scaled=ref_pooling.segment_softmax_csr(src, csr, scaling=True))
    gen = torch.Generator().manual_seed(1)
    csr = random_csr(40, 9, gen)
    for G in (1, 4):
        src = torch.randn(int(csr[-1]), G, generator=gen, requires_grad=True)
        w = torch.randn(int(csr[-1]), G, generator=gen)
        res = {}
        for sc in (False, True):
            out = ref_pooling.segment_softmax_csr(src, csr, scaling=sc)
            (g,) = torch.autograd.grad((out * w).sum(), src)
            res[f'out_{int(sc)}'] = out
            res[f'grad_{int(sc)}'] = g
        save(f"softmax_random_G{G}", src=src, csr=csr, w=w, **res)


def gen_segment():
    print("segment_csr / gather_csr (torch_scatter semantics as restated by the shim)")
    gen = torch.Generator().manual_seed(2)
    csr = random_csr(50, 7, gen)
    M = int(csr[-1])
    src = torch.randn(M, 5, generator=gen)
    src[3] = src[2]  # a tie inside a group
    res = dict(src=src, csr=csr)
    for red in ('sum', 'mean', 'max', 'min'):
        s = src.clone().requires_grad_()
        out = ref_pooling.segment_csr(s, csr, reduce=red)
        w = torch.randn(out.shape, generator=gen)
        (g,) = torch.autograd.grad((out * w).sum(), s)
        res[f'out_{red}'], res[f'w_{red}'], res[f'grad_{red}'] = out, w, g
    pts = torch.randn(50, 5, generator=gen)
    res['gather_src'] = pts
    res['gather_out'] = ref_pooling.gather_csr(pts, csr)
    save("segment_csr", **res)


def pool_case(name, cls, kwargs, N, max_views, C, F_map, F_main, gen, train):
    csr = random_csr(N, max_views, gen)
    V = int(csr[-1])
    module = cls(**kwargs)
    randomize(module, gen)
    module.train(train)
    sd = state(module)
    x_mod = torch.randn(V, C, generator=gen, requires_grad=True)
    x_map = torch.rand(V, F_map, generator=gen, requires_grad=True)
    x_main = torch.randn(N, F_main, generator=gen, requires_grad=True) if F_main else None
    module.save_last = True
    out = module(x_main, x_mod, x_map, csr)
    w = torch.randn(out.shape, generator=gen)
    params = [p for p in module.parameters()]
    ins = [x_mod, x_map] + ([x_main] if x_main is not None else [])
    grads = torch.autograd.grad((out * w).sum(), ins + params, allow_unused=True)
    res = dict(csr=csr, x_mod=x_mod, x_map=x_map, w=w, out=out, train=np.array(int(train)),
               grad_x_mod=grads[0], grad_x_map=grads[1],
               last_C=module._last_C, last_A=module._last_A)
    if module.G is not None:
        res['last_G'] = module._last_G
    if x_main is not None:
        res['x_main'], res['grad_x_main'] = x_main, grads[2]
    for (n, p), g in zip(module.named_parameters(), grads[len(ins):]):
        res['gp/' + n] = g if g is not None else torch.zeros_like(p)
    # BatchNorm running stats after the forward (train mode updates them)
    res.update(state(module, prefix='sd_after/'))
    res.update(sd)
    res['kwargs'] = np.array(repr(kwargs))
    save(name, **res)


def gen_pools():
    print("GroupBimodalCSRPool / QKVBimodalCSRPool forward + backward")
    gen = torch.Generator().manual_seed(3)
    G = ref_pooling.GroupBimodalCSRPool
    Q = ref_pooling.QKVBimodalCSRPool
    # default S3DIS/KITTI/ScanNet settings (sparseconv3d.yaml:6660-6666), small C
    base = dict(in_map=8, in_mod=16, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True)
    pool_case("pool_group_default_train", G, base, 30, 6, 16, 8, 0, gen, True)
    pool_case("pool_group_default_eval", G, base, 30, 6, 16, 8, 0, gen, False)
    # docstring example shape (pooling.py:185-204): C=7 not divisible by G=2
    pool_case("pool_group_docstring", G, dict(in_map=3, in_mod=7, num_groups=2), 12, 5, 7, 3, 0, gen, True)
    pool_case("pool_group_usemod_nogate", G,
              dict(in_map=8, in_mod=12, out_mod=20, num_groups=5, use_mod=True, gating=False,
                   group_scaling=False, map_encoder='MinMaxDiffSetFeat', use_num=True),
              25, 4, 12, 8, 0, gen, True)
    pool_case("pool_group_mlpset_g1", G,
              dict(in_map=8, in_mod=8, num_groups=1, map_encoder='MLPSetFeat'), 20, 5, 8, 8, 0, gen, True)
    pool_case("pool_group_minmaxpool", G,
              dict(in_map=8, in_mod=16, num_groups=16, map_encoder='DeepSetFeat', pool='min_max',
                   fusion='both', use_num=False), 20, 5, 16, 8, 0, gen, True)
    pool_case("pool_qkv_default", Q,
              dict(in_main=10, in_map=8, in_mod=16, num_groups=4, nc_qk=8, use_num=True), 30, 6, 16, 8, 10,
              gen, True)
    pool_case("pool_qkv_modqk", Q,
              dict(in_main=6, in_map=8, in_mod=8, num_groups=2, nc_qk=4, use_mod_q=True, use_mod_k=True,
                   group_scaling=True, dim_scaling=False), 20, 4, 8, 8, 6, gen, False)

    print("BimodalCSRPool / HeuristicBimodalCSRPool")
    csr = random_csr(40, 6, gen)
    V = int(csr[-1])
    x_mod = torch.randn(V, 6, generator=gen)
    x_map = torch.rand(V, 8, generator=gen)
    res = dict(csr=csr, x_mod=x_mod, x_map=x_map)
    for mode in ('max', 'mean', 'min', 'sum'):
        res[f'pool_{mode}'] = ref_pooling.BimodalCSRPool(mode=mode)(None, x_mod, x_map, csr)
    for mode in ('max', 'min'):
        for feat in (0, 'occlusion'):
            res[f'heur_{mode}_{feat}'] = ref_pooling.HeuristicBimodalCSRPool(mode=mode, feat=feat)(
                None, x_mod, x_map, csr)
    a, b = torch.randn(7, 4, generator=gen), torch.randn(7, 4, generator=gen)
    for mode in ref_fusion.BimodalFusion.MODES:
        res[f'fusion_{mode}'] = ref_fusion.BimodalFusion(mode=mode)(a, b)
    res['fusion_a'], res['fusion_b'] = a, b
    save("pool_simple", **res)


def gen_pools_headline():
    """The headline instantiation of the view pooling (C = 64, G = 4, DeepSetFeat, use_num) on a small ragged set that
    contains 32-view points, unseen points and points with more than 32 views: the reference's own forward + backward
    for the bf16 recompute chain (tests/test_gpu_chain.py).  Inputs lie on the bf16 grid so that the device holds
    exactly the same values."""
    print("GroupBimodalCSRPool, headline shape (C = 64, G = 4)")
    gen = torch.Generator().manual_seed(11)
    kwargs = dict(in_map=8, in_mod=64, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True)
    for name, train in (("pool_group_c64_train", True), ("pool_group_c64_eval", False)):
        sizes = torch.randint(1, 9, (64,), generator=gen)
        sizes[torch.rand(64, generator=gen) < 0.2] = 0
        sizes[:8] = 32
        sizes[8], sizes[9] = 40, 70
        sizes = sizes[torch.randperm(64, generator=gen)]
        csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
        V = int(csr[-1])
        module = ref_pooling.GroupBimodalCSRPool(**kwargs)
        randomize(module, gen)
        module.train(train)
        sd = state(module)
        x_mod = torch.randn(V, 64, generator=gen).bfloat16().float().requires_grad_()
        x_map = torch.rand(V, 8, generator=gen)
        module.save_last = True
        out = module(None, x_mod, x_map, csr)
        w = torch.randn(out.shape, generator=gen)
        params = [p for p in module.parameters()]
        grads = torch.autograd.grad((out * w).sum(), [x_mod] + params, allow_unused=True)
        res = dict(csr=csr, x_mod=x_mod, x_map=x_map, w=w, out=out, train=np.array(int(train)), grad_x_mod=grads[0])
        for (n, p), g in zip(module.named_parameters(), grads[1:]):
            res['gp/' + n] = g if g is not None else torch.zeros_like(p)
        res.update(state(module, prefix='sd_after/'))
        res.update(sd)
        res['kwargs'] = np.array(repr(kwargs))
        save(name, **res)


def gen_pools_bilinear():
    """The fused bilinear path's shape (KITTI-360 level 0: C_in = 128 -> C_o = 32, G = 4; sparseconv3d.yaml:7281-7290)
    run by the reference's own `sparse_interpolation` (core/multimodal/image.py:105-170, called as in
    `get_mapped_features`, image.py:1278-1283) + `GroupBimodalCSRPool` (modules/multimodal/pooling.py:263-315): forward
    + backward in train and eval mode on a ragged set with 32-view points, points with 40 / 70 views (> one tile),
    unseen points and views on the image border (replicated padding).  Inputs lie on the bf16 grid so that the device
    holds exactly the same values.  Held by tests/test_gpu_bilinear.py::test_fused_bilinear_against_reference_fixture
    and tests/test_oracle_golden.py."""
    print("sparse_interpolation + GroupBimodalCSRPool, fused bilinear shape (128 -> 32, G = 4)")
    gen = torch.Generator().manual_seed(23)
    B, C_in, H, W, C_o, UP = 3, 128, 8, 12, 32, 8
    kwargs = dict(in_map=8, in_mod=C_in, out_mod=C_o, num_groups=4, use_mod=False, map_encoder='DeepSetFeat',
                  use_num=True)
    msize = (W * UP, H * UP)
    for name, train in (("pool_group_bilinear_train", True), ("pool_group_bilinear_eval", False)):
        n = 96
        sizes = torch.randint(1, 9, (n,), generator=gen)
        sizes[torch.rand(n, generator=gen) < 0.2] = 0
        sizes[:8] = 32
        sizes[8], sizes[9] = 40, 70
        sizes = sizes[torch.randperm(n, generator=gen)]
        csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
        V = int(csr[-1])
        images = torch.randint(0, B, (V,), generator=gen)
        pixels = torch.stack([torch.randint(0, msize[0], (V,), generator=gen),
                              torch.randint(0, msize[1], (V,), generator=gen)], 1).short()
        pixels[:8, 0] = torch.tensor([0, 0, msize[0] - 1, msize[0] - 1, 1, msize[0] - 2, 0, 5])
        pixels[:8, 1] = torch.tensor([0, msize[1] - 1, 0, msize[1] - 1, 1, msize[1] - 2, 7, 0])
        module = ref_pooling.GroupBimodalCSRPool(**kwargs)
        randomize(module, gen)
        with torch.no_grad():                 # E_mod's Linears at fan-in scale: activations stay O(1) over 128 inputs
            for nme, p in module.named_parameters():
                if nme.startswith('E_mod') and p.dim() == 2:
                    p.mul_(2.0 / p.shape[1] ** 0.5)
        x = torch.randn(B, C_in, H, W, generator=gen).bfloat16().float().requires_grad_()
        x_map = torch.rand(V, 8, generator=gen)
        # image.py:1278-1283 (get_mapped_features, interpolate=True)
        resolution = torch.Tensor([msize])
        coords = (pixels / (resolution - 1))[:, [1, 0]]
        if not train:
            # an eval-mode model has running statistics that describe its data: twenty train-mode forwards of the
            # reference module on this batch (momentum 0.1) bring the randomised buffers to within 12 % of them
            module.train(True)
            with torch.no_grad():
                for _ in range(20):
                    module(None, ref_image.sparse_interpolation(x, coords, images), x_map, csr)
        module.train(train)
        sd = state(module)
        x_mod = ref_image.sparse_interpolation(x, coords, images)
        module.save_last = True
        out = module(None, x_mod, x_map, csr)
        w = torch.randn(out.shape, generator=gen)
        params = [p for p in module.parameters()]
        grads = torch.autograd.grad((out * w).sum(), [x] + params, allow_unused=True)
        res = dict(csr=csr, images=images, pixels=pixels, mapping_size=np.array(msize), x=x, x_map=x_map, w=w,
                   out=out, x_interp_head=x_mod[:64], train=np.array(int(train)), grad_x=grads[0], last_A=module._last_A)
        for (nme, p), g in zip(module.named_parameters(), grads[1:]):
            res['gp/' + nme] = g if g is not None else torch.zeros_like(p)
        res.update(state(module, prefix='sd_after/'))
        res.update(sd)
        res['kwargs'] = np.array(repr(kwargs))
        save(name, **res)


# ------------------------------------------------------------------------------------------------
def gen_gather():
    print("get_mapped_features: nearest (after downscale) and bilinear (sparse_interpolation)")
    gen = torch.Generator().manual_seed(4)
    B, C, H, W = 4, 16, 8, 16          # feature maps
    ref_w, ref_h = 128, 64             # mapping resolution (ref_size), ratio 8
    N, n_img = 60, B
    # dense triples (point, image, pixel): every point seen by 0..3 images, 1 pixel per view (exact)
    pts, imgs = [], []
    for p in range(N):
        k = int(torch.randint(0, 4, (1,), generator=gen))
        sel = torch.randperm(n_img, generator=gen)[:k]
        pts += [p] * k
        imgs += sel.tolist()
    pts, imgs = torch.LongTensor(pts), torch.LongTensor(imgs)
    pix = torch.stack([torch.randint(0, ref_w, (len(pts),), generator=gen),
                       torch.randint(0, ref_h, (len(pts),), generator=gen)], dim=1).short()
    feats = torch.rand(len(pts), 8, generator=gen)
    mapping = ref_image.ImageMapping.from_dense(pts, imgs, pix, feats, num_points=N)
    sd = ref_image.SameSettingImageData(
        path=np.array([f'img_{i}' for i in range(B)]), pos=torch.zeros(B, 3), opk=torch.zeros(B, 3),
        ref_size=(ref_w, ref_h), proj_upscale=1, mappings=mapping)
    x = torch.randn(B, C, H, W, generator=gen, requires_grad=True)
    sd.x = x   # setter: downscale <- 8
    res = dict(x=x, point_ids=pts, image_ids=imgs, pixels_dense=pix, map_features_dense=feats,
               pointers=mapping.pointers, images=mapping.images, atom_pointers=mapping.values[1].pointers,
               pixels=mapping.pixels, map_features=mapping.features, downscale=np.array(sd.downscale),
               mapping_size=np.array(sd.mapping_size))
    for interp in (False, True):
        out = sd.get_mapped_features(interpolate=interp)
        w = torch.randn(out.shape, generator=gen)
        (g,) = torch.autograd.grad((out * w).sum(), x)
        tag = 'bilinear' if interp else 'nearest'
        res[f'out_{tag}'], res[f'w_{tag}'], res[f'grad_x_{tag}'] = out, w, g
    save("gather", **res)

    # multi-pixel views (non-exact mapping): atomic max pool absorbs duplicates after downscale
    pts2 = pts.repeat_interleave(3)
    imgs2 = imgs.repeat_interleave(3)
    pix2 = (pix.long().repeat_interleave(3, dim=0) + torch.randint(0, 12, (len(pts2), 2), generator=gen))
    pix2[:, 0].clamp_(max=ref_w - 1)
    pix2[:, 1].clamp_(max=ref_h - 1)
    feats2 = torch.rand(len(pts2), 8, generator=gen)
    uniq = ref_mm.lexargunique(pts2, imgs2, pix2[:, 0], pix2[:, 1])
    pts2, imgs2, pix2, feats2 = pts2[uniq], imgs2[uniq], pix2[uniq].short(), feats2[uniq]
    mapping2 = ref_image.ImageMapping.from_dense(pts2, imgs2, pix2, feats2, num_points=N)
    sd2 = ref_image.SameSettingImageData(
        path=np.array([f'img_{i}' for i in range(B)]), pos=torch.zeros(B, 3), opk=torch.zeros(B, 3),
        ref_size=(ref_w, ref_h), proj_upscale=1, mappings=mapping2)
    sd2.x = x
    out = sd2.get_mapped_features(interpolate=False)
    pooled = ref_pooling.BimodalCSRPool(mode='max')(None, out, None, sd2.mappings.atomic_csr_indexing)
    save("gather_multipixel", x=x, point_ids=pts2, image_ids=imgs2, pixels_dense=pix2,
         map_features_dense=feats2, pointers=mapping2.pointers, images=mapping2.images,
         atom_pointers=mapping2.values[1].pointers, pixels=mapping2.pixels,
         map_features=mapping2.features, downscale=np.array(sd2.downscale), out_nearest=out,
         out_atomic_max=pooled)
    return sd, x


class _RefConv(torch.nn.Module):
    """Tiny 2D encoder with the reference's conv call contract forward(x, reset) (modules.py:472-476)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = torch.nn.Conv2d(c_in, c_out, 3, stride=2, padding=1)

    def forward(self, x, reset=True):
        return torch.relu(self.conv(x))


def gen_branch():
    print("UnimodalBranch forward + backward (modules.py:303-440), ImageData with two settings")
    gen = torch.Generator().manual_seed(5)
    N, C3d, C_in, C = 40, 5, 6, 8
    settings = [dict(B=3, ref=(32, 16), hw=(16, 32)), dict(B=2, ref=(64, 32), hw=(32, 64))]

    def dense(B, ref):
        pts, imgs = [], []
        for p in range(N):
            k = int(torch.randint(0, B + 1, (1,), generator=gen))
            pts += [p] * k
            imgs += torch.randperm(B, generator=gen)[:k].tolist()
        pts, imgs = torch.LongTensor(pts), torch.LongTensor(imgs)
        pix = torch.stack([torch.randint(0, ref[0], (len(pts),), generator=gen),
                           torch.randint(0, ref[1], (len(pts),), generator=gen)], dim=1).short()
        return pts, imgs, pix, torch.rand(len(pts), 8, generator=gen)

    data = [dense(s['B'], s['ref']) for s in settings]
    x_imgs = [torch.randn(s['B'], C_in, *s['hw'], generator=gen) for s in settings]
    for interp in (False, True):
        sds, xs = [], []
        for s, (pts, imgs, pix, feats), x0 in zip(settings, data, x_imgs):
            mapping = ref_image.ImageMapping.from_dense(pts, imgs, pix, feats, num_points=N)
            sd = ref_image.SameSettingImageData(
                path=np.array([f'img_{i}' for i in range(s['B'])]), pos=torch.zeros(s['B'], 3),
                opk=torch.zeros(s['B'], 3), ref_size=s['ref'], proj_upscale=1, mappings=mapping)
            x = x0.clone().requires_grad_()
            sd.x = x
            sds.append(sd)
            xs.append(x)
        conv = _RefConv(C_in, C)
        view_pool = ref_pooling.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True)
        randomize(conv, gen)
        randomize(view_pool, gen)
        branch = ref_modules.UnimodalBranch(
            conv, ref_pooling.BimodalCSRPool(mode='max'), view_pool,
            ref_fusion.BimodalFusion(mode='concatenation'), interpolate=interp)
        branch.train()
        sd_conv, sd_pool = state(conv, 'sd_conv/'), state(view_pool, 'sd_pool/')
        x_3d = torch.randn(N, C3d, generator=gen, requires_grad=True)
        mm = {'x_3d': x_3d, 'x_seen': None, 'modalities': {'image': ref_image.ImageData(sds)}}
        out = branch(mm, 'image')
        y = out['x_3d']
        w = torch.randn(y.shape, generator=gen)
        grads = torch.autograd.grad((y * w).sum(), xs + [x_3d])
        res = dict(x_3d=x_3d, out=y, x_seen=out['x_seen'], w=w, grad_x_3d=grads[-1],
                   n_settings=np.array(len(settings)))
        for i, (s, (pts, imgs, pix, feats), sd) in enumerate(zip(settings, data, sds)):
            m = sd.mappings
            res.update({f's{i}_x_img': xs[i], f's{i}_grad_x_img': grads[i], f's{i}_point_ids': pts,
                        f's{i}_image_ids': imgs, f's{i}_pixels_dense': pix,
                        f's{i}_map_features_dense': feats, f's{i}_pointers': m.pointers,
                        f's{i}_images': m.images, f's{i}_atom_pointers': m.values[1].pointers,
                        f's{i}_pixels': m.pixels, f's{i}_map_features': m.features,
                        f's{i}_ref_size': np.array(s['ref'])})
        save(f"branch_{'bilinear' if interp else 'nearest'}", **res, **sd_conv, **sd_pool)


def gen_branch_fused():
    """The two-setting UnimodalBranch of gen_branch at the width the product's fused bilinear path starts at (out_mod = 32,
    interpolate=True): ImageData with two settings of different map sizes, view_cat_sorting into point order
    (core/multimodal/image.py:1549-1588, modules/multimodal/modules.py:514-525).  Inputs and the 2D encoder's weights lie on
    the bf16 grid (the product runs this under autocast).  Own generator: gen_branch's files do not change."""
    print("UnimodalBranch, two settings, interpolate=True, out_mod = 32 (the fused bilinear path's multi-setting case)")
    gen = torch.Generator().manual_seed(55)
    N, C3d, C_in, C = 600, 5, 6, 32
    settings = [dict(B=3, ref=(32, 16), hw=(16, 32)), dict(B=2, ref=(64, 32), hw=(32, 64))]

    def dense(B, ref):
        pts, imgs = [], []
        for p in range(N):
            k = int(torch.randint(0, B + 1, (1,), generator=gen))
            pts += [p] * k
            imgs += torch.randperm(B, generator=gen)[:k].tolist()
        pts, imgs = torch.LongTensor(pts), torch.LongTensor(imgs)
        pix = torch.stack([torch.randint(0, ref[0], (len(pts),), generator=gen),
                           torch.randint(0, ref[1], (len(pts),), generator=gen)], dim=1).short()
        return pts, imgs, pix, torch.rand(len(pts), 8, generator=gen)

    data = [dense(s['B'], s['ref']) for s in settings]
    sds, xs = [], []
    for s, (pts, imgs, pix, feats) in zip(settings, data):
        mapping = ref_image.ImageMapping.from_dense(pts, imgs, pix, feats, num_points=N)
        sd = ref_image.SameSettingImageData(
            path=np.array([f'img_{i}' for i in range(s['B'])]), pos=torch.zeros(s['B'], 3),
            opk=torch.zeros(s['B'], 3), ref_size=s['ref'], proj_upscale=1, mappings=mapping)
        x = torch.randn(s['B'], C_in, *s['hw'], generator=gen).bfloat16().float().requires_grad_()
        sd.x = x
        sds.append(sd)
        xs.append(x)
    conv = _RefConv(C_in, C)
    view_pool = ref_pooling.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True)
    randomize(conv, gen)
    randomize(view_pool, gen)
    with torch.no_grad():
        for p_ in conv.parameters():
            p_.copy_(p_.bfloat16().float())
    branch = ref_modules.UnimodalBranch(
        conv, ref_pooling.BimodalCSRPool(mode='max'), view_pool,
        ref_fusion.BimodalFusion(mode='concatenation'), interpolate=True)
    branch.train()
    sd_conv, sd_pool = state(conv, 'sd_conv/'), state(view_pool, 'sd_pool/')
    x_3d = torch.randn(N, C3d, generator=gen, requires_grad=True)
    mm = {'x_3d': x_3d, 'x_seen': None, 'modalities': {'image': ref_image.ImageData(sds)}}
    out = branch(mm, 'image')
    y = out['x_3d']
    w = torch.randn(y.shape, generator=gen)
    params = list(view_pool.parameters())
    grads = torch.autograd.grad((y * w).sum(), xs + [x_3d] + params)
    res = dict(x_3d=x_3d, out=y, x_seen=out['x_seen'], w=w, grad_x_3d=grads[len(xs)], n_settings=np.array(len(settings)))
    for (n, _), g in zip(view_pool.named_parameters(), grads[len(xs) + 1:]):
        res['gp/' + n] = g
    for i, (s, (pts, imgs, pix, feats), sd) in enumerate(zip(settings, data, sds)):
        m = sd.mappings
        res.update({f's{i}_x_img': xs[i], f's{i}_grad_x_img': grads[i], f's{i}_point_ids': pts,
                    f's{i}_image_ids': imgs, f's{i}_pixels_dense': pix,
                    f's{i}_map_features_dense': feats, f's{i}_pointers': m.pointers,
                    f's{i}_ref_size': np.array(s['ref'])})
    save("branch_bilinear_c32", **res, **sd_conv, **sd_pool)


# ------------------------------------------------------------------------------------------------
def room_cloud(n, gen, size=(4.0, 4.0, 2.5)):
    """Points on the six faces of a room box + small noise (notebook cell 4 of
    notebooks/synthetic_multimodal_dataset.ipynb), float32."""
    face = torch.randint(0, 6, (n,), generator=gen)
    uvw = torch.rand(n, 3, generator=gen)
    axis = face // 2
    uvw[torch.arange(n), axis] = (face % 2).float()
    xyz = uvw * torch.tensor(size)
    xyz = xyz + (torch.randn(n, 3, generator=gen) * 1e-3).clamp(-0.05, 0.05)
    return xyz.float()


def patch_numba_promotion():
    """np.pi -> np.float64(np.pi) inside the reference module (float-width contract)."""
    proxy = types.ModuleType('np_proxy')
    proxy.__dict__.update(np.__dict__)
    proxy.pi = np.float64(np.pi)
    ref_vis.np = proxy


def run_visibility(name, camera, xyz, img_xyz, gen, exact, img_size, extra=None, mask=None, crop=(0, 0),
                   r_min=0.2, r_max=10.0, voxel=0.05, k_swell=1.0, d_swell=1000):
    extra = extra or {}
    n = xyz.shape[0]
    lin, pla, sca = (torch.rand(n, generator=gen) for _ in range(3))
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1)
    model = ref_vis.SplattingVisibility(
        img_size=img_size, crop_top=crop[0], crop_bottom=crop[1], r_max=np.float64(r_max),
        r_min=np.float64(r_min), camera=camera, voxel=np.float64(voxel), k_swell=np.float64(k_swell),
        d_swell=d_swell, exact=exact)
    kw = dict(img_opk=None, img_intrinsic_pinhole=None, img_intrinsic_fisheye=None, img_extrinsic=None,
              img_mask=mask)
    kw.update(extra)
    # intermediates (camera_projection_cpu output, visibility.py:478-538)
    idx_1, dist, x_proj, y_proj = model._camera_projection(xyz, img_xyz, **kw)
    out = model(xyz, img_xyz, linearity=lin, planarity=pla, scattering=sca, normals=nrm, **kw)
    arrays = dict(xyz=xyz, img_xyz=img_xyz, img_size=np.array(img_size), crop=np.array(crop),
                  r_min=np.array(r_min), r_max=np.array(r_max), voxel=np.array(voxel),
                  k_swell=np.array(k_swell), d_swell=np.array(d_swell), exact=np.array(int(exact)),
                  camera=np.array(camera), linearity=lin, planarity=pla, scattering=sca, normals=nrm,
                  proj_idx=idx_1, proj_dist=dist, proj_x=x_proj, proj_y=y_proj,
                  idx=out['idx'], x=out['x'], y=out['y'], depth=out['depth'], features=out['features'])
    for k, v in kw.items():
        if v is not None:
            arrays[k] = v
    save(name, **arrays)
    return out


def gen_visibility():
    print("SplattingVisibility (camera_projection_cpu + visibility_from_splatting_cpu + features)")
    patch_numba_promotion()
    gen = torch.Generator().manual_seed(6)
    xyz = room_cloud(4000, gen)
    cam = torch.tensor([2.0, 1.7, 1.2])
    size = (512, 256)
    for exact in (True, False):
        tag = 'exact' if exact else 'dense'
        run_visibility(f"vis_equirect_{tag}", 's3dis_equirectangular', xyz, cam, gen, exact, size,
                       extra=dict(img_opk=torch.zeros(3)))
    # rotated pose + crop + mask
    mask = torch.ones(size, dtype=torch.bool)
    mask[100:140, 60:90] = False
    run_visibility("vis_equirect_rot_crop_mask", 's3dis_equirectangular', xyz, cam, gen, True, size,
                   extra=dict(img_opk=torch.tensor([0.3, -0.2, 1.1])), mask=mask, crop=(16, 24))
    # bigger splats (k_swell) and a coarse voxel: many depth conflicts
    run_visibility("vis_equirect_bigsplat", 's3dis_equirectangular', xyz, cam, gen, True, (256, 128),
                   extra=dict(img_opk=torch.zeros(3)), voxel=0.15, k_swell=1.5, d_swell=1e6)

    # pinhole, ScanNet convention: extrinsic = camera-to-world pose, inverted at :232
    def look_at(eye, yaw):
        c, s = np.cos(yaw), np.sin(yaw)
        # camera looks along +z_cam = (c, s, 0) world, x_cam = (s, -c, 0), y_cam = (0, 0, -1)
        R = np.array([[s, 0, c], [-c, 0, s], [0, -1, 0]], dtype=np.float32)
        E = np.eye(4, dtype=np.float32)
        E[:3, :3] = R
        E[:3, 3] = eye
        return torch.from_numpy(E)
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = 288.9, 289.4, 159.5, 119.5
    cam_to_world = look_at(np.array([2.0, 1.7, 1.2], dtype=np.float32), 0.4)
    run_visibility("vis_pinhole_scannet", 'scannet', xyz, cam_to_world[:3, 3].clone(), gen, True, (320, 240),
                   extra=dict(img_extrinsic=cam_to_world, img_intrinsic_pinhole=K), r_min=0.1, r_max=8.0,
                   voxel=0.03)
    K2 = torch.eye(4)
    K2[0, 0], K2[1, 1], K2[0, 2], K2[1, 2] = 552.55, 552.55, 682.05, 238.77
    run_visibility("vis_pinhole_kitti", 'kitti360_perspective', xyz, cam_to_world[:3, 3].clone(), gen, True,
                   (1408, 376), extra=dict(img_extrinsic=cam_to_world, img_intrinsic_pinhole=K2),
                   r_min=0.1, r_max=20.0, voxel=0.05, k_swell=1.5, d_swell=1e6)
    fish = torch.tensor([2.2134, 0.016798, 1.6548, 1336.3, 1335.8, 716.94, 705.76])
    run_visibility("vis_fisheye_kitti", 'kitti360_fisheye', xyz, cam_to_world[:3, 3].clone(), gen, True,
                   (1400, 1400), extra=dict(img_extrinsic=cam_to_world, img_intrinsic_fisheye=fish),
                   r_min=0.1, r_max=20.0, voxel=0.05, k_swell=1.5, d_swell=1e6)
    # empty result (camera far away): visibility.py:1721-1729
    run_visibility("vis_equirect_empty", 's3dis_equirectangular', xyz, torch.tensor([100.0, 100.0, 100.0]),
                   gen, True, size, extra=dict(img_opk=torch.zeros(3)))


def gen_visibility_models():
    """BiasuttiVisibility / DepthBasedVisibility of the reference (visibility.py:1356-1496, :1779-1803) on the room cloud:
    the reference's own classes, KeOps through oracle/shims/pykeops, the S3DIS depth PNG written here with PIL."""
    print("BiasuttiVisibility, DepthBasedVisibility (reference classes)")
    import tempfile
    from PIL import Image
    patch_numba_promotion()
    gen = torch.Generator().manual_seed(16)
    xyz = room_cloud(3000, gen)
    cam = torch.tensor([2.0, 1.7, 1.2])
    size = (256, 128)
    n = xyz.shape[0]
    lin, pla, sca = (torch.rand(n, generator=gen) for _ in range(3))
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1)
    base = dict(img_size=size, crop_top=0, crop_bottom=0, r_max=np.float64(10.0), r_min=np.float64(0.2),
                camera='s3dis_equirectangular')
    kw = dict(img_opk=torch.tensor([0.1, -0.05, 0.7]), img_intrinsic_pinhole=None, img_intrinsic_fisheye=None,
              img_extrinsic=None, img_mask=None)
    common = dict(xyz=xyz, img_xyz=cam, img_size=np.array(size), r_min=np.array(0.2), r_max=np.array(10.0),
                  linearity=lin, planarity=pla, scattering=sca, normals=nrm, img_opk=kw['img_opk'])
    for name, k, margin, thr in (("vis_biasutti", 20, None, None), ("vis_biasutti_wrap", 12, 20, 0.5)):
        model = ref_vis.BiasuttiVisibility(k=k, margin=margin, threshold=thr, **base)
        idx_1, dist, x_proj, y_proj = model._camera_projection(xyz, cam, **kw)
        nbr = ref_vis.k_nn_image_system(x_proj, y_proj, k=k, x_margin=margin, x_width=size[0])
        out = model(xyz, cam, linearity=lin, planarity=pla, scattering=sca, normals=nrm, **kw)
        save(name, k=np.array(k), margin=np.array(-1 if margin is None else margin),
             threshold=np.array(-1.0 if thr is None else thr), proj_idx=idx_1, proj_dist=dist, proj_x=x_proj,
             proj_y=y_proj, neighbors=nbr, idx=out['idx'], x=out['x'], y=out['y'], depth=out['depth'],
             features=out['features'], **common)
    # depth map: the true distance of the closest projected point per pixel + noise, a block of missing pixels
    model = ref_vis.DepthBasedVisibility(depth_threshold=0.05, **base)
    idx_1, dist, x_proj, y_proj = model._camera_projection(xyz, cam, **kw)
    dm = np.full(size, 65535, dtype=np.uint16)
    order = np.argsort(-dist.numpy(), kind='stable')            # the closest point of a pixel is written last
    dm[x_proj.long().numpy()[order], y_proj.long().numpy()[order]] = np.round(dist.numpy()[order] * 512).astype(np.uint16)
    dm[40:60, 30:50] = 65535
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "depth.png")
        Image.fromarray(dm.T).save(path)                          # read_s3dis_depth_map transposes back
        out = model(xyz, cam, linearity=lin, planarity=pla, scattering=sca, normals=nrm, depth_map_path=path, **kw)
        depth_m = ref_vis.read_s3dis_depth_map(path, img_size=size, empty=-1)
    save("vis_depth_map", depth_threshold=np.array(0.05), depth_png_u16=dm, depth_map=depth_m, proj_idx=idx_1,
         proj_dist=dist, proj_x=x_proj, proj_y=y_proj, idx=out['idx'], x=out['x'], y=out['y'], depth=out['depth'],
         features=out['features'], **common)


def gen_lex_and_csr():
    print("lexargsort / lexargunique / lexunique and CSR ops")
    gen = torch.Generator().manual_seed(7)
    a = torch.randint(0, 30, (500,), generator=gen)
    b = torch.randint(0, 5, (500,), generator=gen)
    c = torch.randint(0, 40, (500,), generator=gen).short()
    comp = ref_mm.CompositeTensor(a, b, c)
    res = dict(a=a, b=b, c=c, composite=comp.data, argunique=ref_mm.lexargunique(a, b, c))
    ua, ub, uc = ref_mm.lexunique(a, b, c)
    res.update(unique_a=ua, unique_b=ub, unique_c=uc)
    sa, sb, sc = ref_mm.lexsort(a, b, c)
    res.update(sort_a=sa, sort_b=sb, sort_c=sc)
    # argsort: only the sorted keys are order-independent of the (unstable) algorithm
    order = ref_mm.lexargsort(a, b, c)
    res['argsort_keys'] = comp.data[order]
    # CSR: pointers from sorted indices, index-select, insert_empty_groups
    idx = torch.sort(torch.randint(0, 20, (60,), generator=gen))[0]
    vals = torch.randn(60, generator=gen)
    csr = CSRData(idx, vals, dense=True)
    res.update(csr_idx=idx, csr_vals=vals, csr_pointers=csr.pointers)
    sel = torch.LongTensor([3, 0, 0, 7, 5])
    sub = csr[sel]
    res.update(sel=sel, sel_pointers=sub.pointers, sel_vals=sub.values[0])
    groups = torch.unique(idx)
    csr2 = CSRData(idx, vals, dense=True).insert_empty_groups(groups, num_groups=25)
    res.update(groups=groups, ins_pointers=csr2.pointers)
    # batching round-trip (csr.py:347-456)
    items = []
    for i in range(3):
        ii = torch.sort(torch.randint(0, 6, (15,), generator=gen))[0]
        items.append(CSRData(ii, torch.randint(0, 4 + i, (15,), generator=gen), torch.randn(15, generator=gen),
                             dense=True, is_index_value=[True, False]))
        res[f'b{i}_pointers'], res[f'b{i}_v0'], res[f'b{i}_v1'] = items[-1].pointers, items[-1].values[0], \
            items[-1].values[1]
    batch = CSRBatch.from_csr_list(items)
    res.update(batch_pointers=batch.pointers, batch_v0=batch.values[0], batch_v1=batch.values[1],
               batch_sizes=batch.__sizes__)
    save("lex_csr", **res)


def gen_mapping():
    """from_dense / select_points / MapImages post-processing on top of SplattingVisibility."""
    print("ImageMapping.from_dense / select_points(pick, merge) / MapImages assembly")
    patch_numba_promotion()
    gen = torch.Generator().manual_seed(8)
    N = 3000
    xyz = room_cloud(N, gen)
    lin, pla, sca = (torch.rand(N, generator=gen) for _ in range(3))
    nrm = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=1)
    ref_size, proj_upscale = (256, 128), 2
    proj_size = (ref_size[0] * proj_upscale, ref_size[1] * proj_upscale)
    cams = torch.tensor([[1.0, 1.0, 1.2], [3.0, 1.2, 1.0], [2.0, 3.0, 1.5], [50.0, 50.0, 50.0]])
    model = ref_vis.SplattingVisibility(
        img_size=proj_size, r_max=np.float64(10.0), r_min=np.float64(0.2), voxel=np.float64(0.05),
        k_swell=np.float64(1.0), d_swell=1000, exact=True)
    # MapImages._process restated (core/data_transform/multimodal/image.py:238-353, :372-417); the sphere
    # sampling keeps every point here (r_max covers the room), candidate order = identity
    image_ids, point_ids, features, pixels = [], [], [], []
    for i_img in range(cams.shape[0]):
        out = model(xyz, cams[i_img], img_opk=torch.zeros(3), linearity=lin, planarity=pla,
                    scattering=sca, normals=nrm)
        if out['idx'].shape[0] == 0:
            continue
        pid = out['idx']
        px = out['x'].long() // proj_upscale
        py = out['y'].long() // proj_upscale
        keep = torch.where((px >= 0) & (py >= 0) & (px < ref_size[0]) & (py < ref_size[1]))
        px, py, pid, ft = px[keep], py[keep], pid[keep], out['features'].float()[keep]
        u = ref_mm.lexargunique(pid, px, py)
        px, py, pid, ft = px[u], py[u], pid[u], ft[u]
        image_ids.append(i_img)
        point_ids.append(pid)
        features.append(ft)
        pixels.append(torch.stack((px, py), dim=1).short())
    image_ids = torch.LongTensor(image_ids)
    seen = ref_mm.lexunique(image_ids)
    image_ids = torch.bucketize(image_ids, seen)
    image_ids = image_ids.repeat_interleave(torch.LongTensor([x.shape[0] for x in point_ids]))
    point_ids, pixels, features = torch.cat(point_ids), torch.cat(pixels), torch.cat(features)
    mapping = ref_image.ImageMapping.from_dense(point_ids, image_ids, pixels, features, num_points=N)
    res = dict(xyz=xyz, cams=cams, linearity=lin, planarity=pla, scattering=sca, normals=nrm,
               ref_size=np.array(ref_size), proj_upscale=np.array(proj_upscale), seen_images=seen,
               dense_point_ids=point_ids, dense_image_ids=image_ids, dense_pixels=pixels,
               dense_features=features, pointers=mapping.pointers, images=mapping.images,
               atom_pointers=mapping.values[1].pointers, pixels=mapping.pixels, features=mapping.features)
    # select_points 'pick' (image.py:2167-2209) and 'merge' (:2211-2273)
    pick = torch.randperm(N, generator=gen)[:500]
    mp = mapping.select_points(pick, mode='pick')
    res.update(pick_idx=pick, pick_pointers=mp.pointers, pick_images=mp.images,
               pick_atom_pointers=mp.values[1].pointers, pick_pixels=mp.pixels, pick_features=mp.features)
    merge = torch.randint(0, 700, (N,), generator=gen)
    merge[:700] = torch.arange(700)  # every output voxel exists
    mm = mapping.select_points(merge, mode='merge')
    res.update(merge_idx=merge, merge_pointers=mm.pointers, merge_images=mm.images,
               merge_atom_pointers=mm.values[1].pointers, merge_pixels=mm.pixels, merge_features=mm.features)
    save("mapping_build", **res)


def gen_mapping_cylinder():
    """MapImages with cylinder=True (the KITTI-360 configuration, core/data_transform/multimodal/image.py:242-243):
    per image the candidates are the points inside the vertical cylinder of radius r_max around the camera
    (CylinderSampling: KD-tree radius query in the xy plane, core/data_transform/transforms.py:353-403), in the
    order of the query result -- made explicit here by sorting it --, then the same per-image loop as
    gen_mapping() with the kitti360_perspective camera."""
    print("MapImages, cylinder candidate sampling (KITTI-360 perspective camera)")
    from sklearn.neighbors import KDTree
    patch_numba_promotion()
    gen = torch.Generator().manual_seed(12)
    N = 6000
    xyz = room_cloud(N, gen, size=(30.0, 30.0, 4.0))
    lin, pla, sca = (torch.rand(N, generator=gen) for _ in range(3))
    nrm = torch.nn.functional.normalize(torch.randn(N, 3, generator=gen), dim=1)
    ref_size, proj_upscale, r_max = (352, 94), 2, 9.0
    proj_size = (ref_size[0] * proj_upscale, ref_size[1] * proj_upscale)

    def look_at(eye, yaw):
        c, sn = np.cos(yaw), np.sin(yaw)
        R = np.array([[sn, 0, c], [-c, 0, sn], [0, -1, 0]], dtype=np.float32)
        E = np.eye(4, dtype=np.float32)
        E[:3, :3] = R
        E[:3, 3] = eye
        return torch.from_numpy(E)
    poses = [look_at(np.array([6.0, 7.0, 1.6], dtype=np.float32), 0.3),
             look_at(np.array([20.0, 12.0, 1.5], dtype=np.float32), 2.2),
             look_at(np.array([14.0, 25.0, 1.7], dtype=np.float32), -1.4)]
    fx, fy, mx, my = 276.3, 276.3, 341.0, 119.4          # KITTI-360 intrinsics scaled to the 704 x 188 projection
    K = torch.eye(4)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = fx, fy, mx, my
    model = ref_vis.SplattingVisibility(
        img_size=proj_size, r_max=np.float64(r_max), r_min=np.float64(0.3), camera='kitti360_perspective',
        voxel=np.float64(0.08), k_swell=np.float64(1.2), d_swell=1000, exact=True)
    tree = KDTree(np.asarray(xyz[:, :-1]), leaf_size=50)
    image_ids, point_ids, features, pixels, n_cand = [], [], [], [], []
    for i_img, E in enumerate(poses):
        ind = torch.LongTensor(np.sort(tree.query_radius(np.asarray(E[:2, 3]).reshape(1, 2), r=r_max)[0]))
        n_cand.append(len(ind))
        out = model(xyz[ind], E[:3, 3].clone(), img_extrinsic=E, img_intrinsic_pinhole=K, linearity=lin[ind],
                    planarity=pla[ind], scattering=sca[ind], normals=nrm[ind])
        if out['idx'].shape[0] == 0:
            continue
        pid = ind[out['idx']]
        px = out['x'].long() // proj_upscale
        py = out['y'].long() // proj_upscale
        keep = torch.where((px >= 0) & (py >= 0) & (px < ref_size[0]) & (py < ref_size[1]))
        px, py, pid, ft = px[keep], py[keep], pid[keep], out['features'].float()[keep]
        u = ref_mm.lexargunique(pid, px, py)
        image_ids.append(i_img)
        point_ids.append(pid[u])
        features.append(ft[u])
        pixels.append(torch.stack((px[u], py[u]), dim=1).short())
    assert min(n_cand) < N and len(image_ids) == len(poses)
    image_ids = torch.arange(len(image_ids)).repeat_interleave(torch.LongTensor([x.shape[0] for x in point_ids]))
    point_ids, pixels, features = torch.cat(point_ids), torch.cat(pixels), torch.cat(features)
    mapping = ref_image.ImageMapping.from_dense(point_ids, image_ids, pixels, features, num_points=N)
    save("mapping_build_cylinder", xyz=xyz, linearity=lin, planarity=pla, scattering=sca, normals=nrm,
         extrinsic=torch.stack(poses), fx=np.array(fx), fy=np.array(fy), mx=np.array(mx), my=np.array(my),
         ref_size=np.array(ref_size), proj_upscale=np.array(proj_upscale), r_max=np.array(r_max),
         n_candidates=np.array(n_cand), pointers=mapping.pointers, images=mapping.images,
         atom_pointers=mapping.values[1].pointers, pixels=mapping.pixels, features=mapping.features)


def import_ref_transforms():
    """core/data_transform/multimodal/image.py of the reference, loaded as a single file: its package __init__
    pulls torch_geometric.transforms, torch_cluster, torch_points_kernels, datasets ... (absent).  The names the
    file imports from there (samplers, FAISS finder, torchvision) are placeholders: none of the transforms
    pinned below calls them."""
    import importlib.util
    name = "torch_points3d.core.data_transform.multimodal.image"
    if name in sys.modules:
        return sys.modules[name]

    def placeholder(mod_name, names, is_pkg=False):
        m = types.ModuleType(mod_name)
        if is_pkg:
            m.__path__ = []
        for n in names:
            setattr(m, n, type(n, (), {}))
        sys.modules[mod_name] = m
        return m
    placeholder("torch_points3d.core.data_transform",
                ["SphereSampling", "CylinderSampling", "GridSampling3D", "SaveOriginalPosId"], is_pkg=True)
    placeholder("torch_points3d.core.data_transform.multimodal", [], is_pkg=True)
    placeholder("torch_points3d.core.spatial_ops", [], is_pkg=True)
    placeholder("torch_points3d.core.spatial_ops.neighbour_finder", ["FAISSGPUKNNNeighbourFinder"])
    tv = placeholder("torchvision", [], is_pkg=True)
    tv.transforms = placeholder("torchvision.transforms", ["ColorJitter", "GaussianBlur", "Normalize"])
    spec = importlib.util.spec_from_file_location(
        name, "/root/reference/torch_points3d/core/data_transform/multimodal/image.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _mapping_arrays(prefix, m):
    return {prefix + "pointers": m.pointers, prefix + "images": m.images,
            prefix + "atom_pointers": m.values[1].pointers, prefix + "pixels": m.pixels,
            prefix + "features": m.features}


def gen_transforms():
    """ImageMapping rescale / crop / select_views (core/multimodal/image.py:1901-2027, 2095-2165, 2279-2342) and the
    online transforms SelectMappingFromPointId, PickImagesFromMappingArea, CenterRoll, CropImageGroups,
    PickImagesFromMemoryCredit (core/data_transform/multimodal/image.py:615-1141), run from the reference source."""
    print("ImageMapping rescale / crop / select_views + online mapping transforms")
    from torch_geometric.data import Data
    T = import_ref_transforms()
    gen = torch.Generator().manual_seed(21)
    B, N = 6, 150
    ref_w, ref_h = 128, 64
    # non-exact mapping: each view owns 1-4 pixels clustered in a window of its image (so that rollings and
    # croppings have something to do); points seen by 0..4 images
    pts, imgs, pix = [], [], []
    centre = torch.stack([torch.randint(0, ref_w, (B,), generator=gen), torch.randint(10, ref_h - 10, (B,), generator=gen)], 1)
    span = torch.tensor([[15, 10], [45, 20], [100, 50], [20, 15], [125, 60], [8, 8]])
    for p in range(N):
        k = int(torch.randint(0, 5, (1,), generator=gen))
        for i in torch.randperm(B, generator=gen)[:k].tolist():
            na = int(torch.randint(1, 5, (1,), generator=gen))
            dx = torch.randint(0, int(span[i, 0]), (na,), generator=gen) - int(span[i, 0]) // 2
            dy = torch.randint(0, int(span[i, 1]), (na,), generator=gen) - int(span[i, 1]) // 2
            x = (int(centre[i, 0]) + dx) % ref_w               # spherical images wrap along the width
            y = (int(centre[i, 1]) + dy).clamp(0, ref_h - 1)
            pts += [p] * na
            imgs += [i] * na
            pix += torch.stack([x, y], 1).tolist()
    pts, imgs, pix = torch.LongTensor(pts), torch.LongTensor(imgs), torch.LongTensor(pix)
    u = ref_mm.lexargunique(pts, imgs, pix[:, 0], pix[:, 1])
    pts, imgs, pix = pts[u], imgs[u], pix[u].short()
    feats = torch.rand(len(pts), 4, generator=gen)
    x_img = torch.randint(0, 255, (B, 1, ref_h, ref_w), generator=gen, dtype=torch.uint8)
    pos = torch.rand(B, 3, generator=gen) * 5

    def fresh():
        mapping = ref_image.ImageMapping.from_dense(pts, imgs, pix, feats, num_points=N)
        sd = ref_image.SameSettingImageData(
            path=np.array([f'img_{i}' for i in range(B)]), pos=pos.clone(), opk=torch.zeros(B, 3),
            ref_size=(ref_w, ref_h), proj_upscale=1, mappings=mapping, x=x_img.clone())
        return sd
    res = dict(point_ids=pts, image_ids=imgs, pixels_dense=pix, map_features_dense=feats, x=x_img, pos=pos,
               ref_size=np.array((ref_w, ref_h)), num_points=np.array(N))
    sd = fresh()
    m = sd.mappings
    res.update(_mapping_arrays("in_", m))
    # ---- ImageMapping methods
    for r in (2, 4, 8):
        res.update(_mapping_arrays(f"down{r}_", m.downscale_images(r)))
    res.update(_mapping_arrays("up2_", m.upscale_images(2)))
    res.update(_mapping_arrays("up3nc_", m.upscale_images(3, center=False)))
    crop_size = (48, 32)
    crop_off = torch.stack([torch.randint(0, ref_w - crop_size[0], (B,), generator=gen),
                            torch.randint(0, ref_h - crop_size[1], (B,), generator=gen)], 1)
    res.update(crop_size=np.array(crop_size), crop_offsets=crop_off)
    res.update(_mapping_arrays("crop_", m.crop(crop_size, crop_off)))
    view_mask = (torch.rand(m.num_items, generator=gen) < 0.6) & (m.images != 1) & (m.images != 4)   # two images vanish
    out = m.select_views(view_mask)
    mv, seen = out
    assert seen is not None
    res.update(view_mask=view_mask, sv_seen_images=seen)
    res.update(_mapping_arrays("sv_", mv))
    bb = m.bounding_boxes
    res.update(bbox=torch.stack([b.long() for b in bb], 0))
    # ---- SelectMappingFromPointId
    keep = torch.randperm(N, generator=gen)[:40]
    data = Data(pos=torch.rand(40, 3, generator=gen), mapping_index=keep.clone())
    data.num_nodes = 40
    d2, sd2 = T.SelectMappingFromPointId()(data, fresh())
    res.update(sel_keep=keep, sel_mapping_index=d2.mapping_index, sel_pos=sd2.pos, sel_num_views=np.array(sd2.num_views))
    res.update(_mapping_arrays("sel_", sd2.mappings))
    # ---- PickImagesFromMappingArea (pixel count and bounding box variants)
    data = Data(pos=torch.rand(N, 3, generator=gen), mapping_index=torch.arange(N))
    for tag, kw in (("area", dict(area_ratio=0.003, n_max=4)), ("bbox", dict(area_ratio=0.05, n_max=None, use_bbox=True))):
        _, sda = T.PickImagesFromMappingArea(**kw)(data, fresh())
        res.update({f"pick_{tag}_pos": sda.pos, f"pick_{tag}_x": sda.x})
        res.update(_mapping_arrays(f"pick_{tag}_", sda.mappings))
    # ---- CenterRoll
    _, sdr = T.CenterRoll(angular_res=16)(data, fresh())
    res.update(roll_rollings=sdr.rollings, roll_x=sdr.x)
    res.update(_mapping_arrays("roll_", sdr.mappings))
    # ---- CropImageGroups (after the roll, like the S3DIS pipeline), then PickImagesFromMemoryCredit on the groups
    _, idata = T.CropImageGroups(padding=2, min_size=16)(data, sdr)
    res["crop_groups"] = np.array(len(idata))
    for gi, g_sd in enumerate(idata):
        res.update({f"cg{gi}_crop_size": np.array(g_sd.crop_size), f"cg{gi}_crop_offsets": g_sd.crop_offsets,
                    f"cg{gi}_pos": g_sd.pos, f"cg{gi}_x": g_sd.x, f"cg{gi}_rollings": g_sd.rollings})
        res.update(_mapping_arrays(f"cg{gi}_", g_sd.mappings))
    data.num_nodes = N
    np.random.seed(5)
    credit = int(sum(g.img_size[0] * g.img_size[1] for g in idata) * 1.6)
    _, picked = T.PickImagesFromMemoryCredit(credit=credit, k_coverage=2)(data, idata)
    res.update(credit=np.array(credit), credit_seed=np.array(5), credit_groups=np.array(len(picked)))
    for gi, g_sd in enumerate(picked):
        res.update({f"mc{gi}_pos": g_sd.pos, f"mc{gi}_crop_size": np.array(g_sd.crop_size)})
        res.update(_mapping_arrays(f"mc{gi}_", g_sd.mappings))
    # ---- transforms on the raw images that touch the mappings / the channel layout (:1163-1232), two images
    sdf = fresh()[torch.LongTensor([0, 3])]
    torch.manual_seed(3)
    _, sdf = T.RandomHorizontalFlip(p=1.0)(data, sdf)
    res.update(flip_x=sdf.x)
    res.update(_mapping_arrays("flip_", sdf.mappings))
    _, sdf = T.ToFloatImage()(data, sdf)
    _, sdf = T.AddPixelHeightFeature()(data, sdf)
    _, sdf = T.AddPixelWidthFeature()(data, sdf)
    res.update(feat_x=sdf.x[:, :, ::4, ::4])       # sub-sampled: the added channels are linear ramps
    save("transforms", **res)


def gen_neighborhood():
    """NeighborhoodBasedMappingFeatures._process (core/data_transform/multimodal/image.py:482-612) run from the
    reference source: K-NN through the KeOps branch (argKmin of the shim: brute force, ties to the lower index),
    density and occlusion exactly as the reference writes them.  Two clouds: a noisy surface (no tied distances) and
    a cloud with duplicated points (zero distances)."""
    print("NeighborhoodBasedMappingFeatures")
    from torch_geometric.data import Data
    T = import_ref_transforms()
    gen = torch.Generator().manual_seed(33)
    res = {}
    for tag, n in (("surf", 1200), ("dup", 400)):
        xyz = torch.rand(n, 3, generator=gen) * torch.tensor([4.0, 4.0, 2.5])
        if tag == "surf":
            face = torch.randint(0, 3, (n,), generator=gen)
            xyz[torch.arange(n), face] = 0.0
            xyz += torch.randn(n, 3, generator=gen) * 1e-3
        else:
            xyz[n // 2:] = xyz[:n - n // 2]            # every point twice
        B = 5
        pts, imgs = [], []
        for p in range(n):
            k = int(torch.randint(0, 4, (1,), generator=gen))
            for i in torch.randperm(B, generator=gen)[:k].tolist():
                pts.append(p)
                imgs.append(i)
        pts, imgs = torch.LongTensor(pts), torch.LongTensor(imgs)
        pix = torch.randint(0, 32, (len(pts), 2), generator=gen).short()
        feats = torch.rand(len(pts), 3, generator=gen)
        mapping = ref_image.ImageMapping.from_dense(pts, imgs, pix, feats, num_points=n)
        sd = ref_image.SameSettingImageData(
            path=np.array([f'img_{i}' for i in range(B)]), pos=torch.rand(B, 3, generator=gen), opk=torch.zeros(B, 3),
            ref_size=(32, 32), proj_upscale=1, mappings=mapping)
        data = Data(pos=xyz.clone())
        data.num_nodes = n
        k_list = [20, 5]
        tr = T.NeighborhoodBasedMappingFeatures(k=k_list, voxel=0.05, density=True, occlusion=True, use_faiss=False)
        assert tr.k_list == [5, 20]
        _, sd_out = tr(data, sd)
        f = sd_out.mappings.features
        assert f.shape[1] == 3 + 4
        res.update({f"{tag}_xyz": xyz, f"{tag}_point_ids": pts, f"{tag}_image_ids": imgs, f"{tag}_pixels": pix,
                    f"{tag}_features_in": feats, f"{tag}_features_out": f, f"{tag}_pointers": sd_out.mappings.pointers,
                    f"{tag}_images": sd_out.mappings.images})
    save("neighborhood", k_list=np.array([5, 20]), voxel=np.array(0.05), **res)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    # bit-reproducible files: one thread and deterministic CPU algorithms -- the multi-threaded
    # index_add / scatter-add of the reference's backward sums in a run-dependent order (1 ulp in grad_x)
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)
    torch.manual_seed(0)
    np.random.seed(0)
    only = set(sys.argv[1:])
    jobs = dict(softmax=gen_softmax, segment=gen_segment, pools=gen_pools, pools_headline=gen_pools_headline,
                pools_bilinear=gen_pools_bilinear,
                gather=gen_gather,
                branch=gen_branch, branch_fused=gen_branch_fused, visibility=gen_visibility, lex=gen_lex_and_csr, mapping=gen_mapping,
                transforms=gen_transforms, cylinder=gen_mapping_cylinder,
                neighborhood=gen_neighborhood, visibility_models=gen_visibility_models)
    for name, fn in jobs.items():
        if not only or name in only:
            fn()
