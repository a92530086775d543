"""One small invocation of the hot path on a HIP device, checked against the CPU oracle
(called by ``__graft_entry__.smoke()``; the oracle import lives here because smoke() is one of the
three places allowed to use it)."""
import os
import sys

import torch


def run(device="cuda:0"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import pooling_oracle as O
    from .modules.multimodal import pooling as P
    from . import ops

    gen = torch.Generator().manual_seed(0)
    N, C, B, H, W = 2000, 64, 4, 16, 32
    sizes = torch.randint(0, 5, (N,), generator=gen)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    images = torch.randint(0, B, (V,), generator=gen)
    pixels = torch.stack([torch.randint(0, W, (V,), generator=gen),
                          torch.randint(0, H, (V,), generator=gen)], 1).short()
    x = torch.randn(B, C, H, W, generator=gen)
    x_map = torch.rand(V, 8, generator=gen)
    kwargs = dict(in_map=8, in_mod=C, num_groups=4, use_mod=False, map_encoder='DeepSetFeat', use_num=True)
    ref = O.GroupBimodalCSRPool(**kwargs)
    mod = P.GroupBimodalCSRPool(**kwargs)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(device)

    # oracle
    xr = x.clone().requires_grad_()
    out_ref = ref(None, O.gather_nearest(xr, images, pixels), x_map, csr)
    (g_ref,) = torch.autograd.grad(out_ref.square().sum(), xr)
    # HIP path
    xd = x.to(device).requires_grad_()
    packed = ops.pack_gather_index(images.to(device), torch.arange(V + 1, device=device), pixels.to(device))
    out = mod(None, ops.gather_nearest(xd, packed), x_map.to(device), csr.to(device))
    (g_dev,) = torch.autograd.grad(out.square().sum(), xd)
    torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu(), out_ref, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(g_dev.cpu(), g_ref, rtol=1e-2, atol=1e-3)
    print(f"smoke ok: N={N} V={V} C={C} out={tuple(out.shape)} max|out-ref|="
          f"{(out.cpu() - out_ref).abs().max().item():.2e}")
