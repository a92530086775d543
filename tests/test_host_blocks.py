"""Host logic of MultimodalBlockDown that needs no device (reference modules/multimodal/modules.py:21-236)."""
import pytest
import torch


def test_block_down_ctor_contract_and_identity_passthrough():
    from deepviewagg_amd.modules.multimodal.modules import MultimodalBlockDown, IdentityBranch
    from deepviewagg_amd.core.common_modules.base_modules import Identity
    b = MultimodalBlockDown(None, None, image=IdentityBranch())
    assert isinstance(b.block_1, Identity) and isinstance(b.block_2, Identity) and b.modalities == ['image']
    assert b.sampler == [None, None]
    d = dict(x_3d=torch.zeros(3, 2), x_seen=None, modalities={})
    assert b(d) is d                       # Identity blocks leave the dictionary untouched
    with pytest.raises(AssertionError):
        MultimodalBlockDown(None, None, lidar=IdentityBranch())
    with pytest.raises(AssertionError):
        MultimodalBlockDown(None, None, image=torch.nn.Linear(2, 2))


def test_block_down_rejects_unknown_3d_formats():
    from deepviewagg_amd.modules.multimodal.modules import MultimodalBlockDown

    class Block(torch.nn.Module):
        def forward(self, x):
            return x

    with pytest.raises(NotImplementedError):
        MultimodalBlockDown.forward_3d_block_down(dict(x_3d=object(), x_seen=None, modalities={}), Block())


def test_voxel_oracle_floor_and_lookup():
    import numpy as np
    from oracle import voxel_oracle as VO
    c = np.array([[0, 1, 2, 0], [3, -1, 5, 1], [-4, -3, 7, 1], [8, 8, 8, 0]], dtype=np.int32)
    assert VO.floor_coords(c, 2).tolist() == [[0, 0, 2, 0], [2, -2, 4, 1], [-4, -4, 6, 1], [8, 8, 8, 0]]
    out = np.array([[2, -2, 4, 1], [0, 0, 2, 0], [-4, -4, 6, 1]], dtype=np.int32)
    assert VO.voxel_parent_index(c, out, 2).tolist() == [1, 0, 2, -1]


def test_modality_dropout_drops_the_whole_tensor():
    """dropout.py:5-15: train = all-or-nothing Bernoulli(1 - p) per call, eval = scale by 1 / (1 - p)."""
    import pytest
    import torch
    from deepviewagg_amd.modules.multimodal.dropout import ModalityDropout
    torch.manual_seed(0)
    x = torch.randn(50, 7)
    drop = ModalityDropout(p=0.3, inplace=True).train()
    kept = 0
    for _ in range(400):
        y = drop(x)
        assert torch.equal(y, x) or float(y.abs().sum()) == 0.0
        kept += int(torch.equal(y, x))
    assert 0.6 < kept / 400 < 0.8 and torch.equal(x, x.clone())      # input untouched
    assert torch.allclose(drop.eval()(x), x / 0.7)
    with pytest.raises(AssertionError):
        ModalityDropout(p=1.5)


def test_multimodal_input_builds_the_block_dictionary():
    """applications/sparseconv3d.py:145-165 (`_set_input`): voxel tensor with (x, y, z, batch) int32 coordinates,
    x_seen None, the batch's modalities moved to the device; a bare voxel tensor for a 3D-only model."""
    from types import SimpleNamespace
    import torch
    from deepviewagg_amd.modules.multimodal.modules import multimodal_input, _is_voxel_tensor

    class Batch(SimpleNamespace):
        def to(self, device):
            self.moved_to = device
            return self
    data = Batch(x=torch.randn(5, 3), coords=torch.arange(15).view(5, 3), batch=torch.tensor([0, 0, 1, 1, 1]),
                 modalities={"image": object()})
    mm = multimodal_input(data, "cpu")
    assert set(mm) == {"x_3d", "x_seen", "modalities"} and mm["x_seen"] is None
    assert mm["modalities"] is data.modalities and data.moved_to == "cpu"
    x = mm["x_3d"]
    assert _is_voxel_tensor(x) and x.s == 1 and x.C.dtype == torch.int32 and x.C.shape == (5, 4)
    assert torch.equal(x.C[:, 3], data.batch.int()) and torch.equal(x.C[:, :3], data.coords.int())
    assert torch.equal(x.F, data.x) and x.coord_maps[1] is x.C
    assert _is_voxel_tensor(multimodal_input(data, "cpu", is_multimodal=False))


def test_zero_pool_pieces_do_not_share_a_version_counter():
    """ops._zero_piece (the allocator behind ops.zeros_small) on a CPU pool: pieces share the pool's storage but each has
    its own autograd version counter, so an in-place update of one (AccumulateGrad on a stolen gradient) does not
    invalidate another that a fused node saved for backward.  Reproduces the failure mode with two
    autograd Functions saving / returning pieces."""
    import torch
    from deepviewagg_amd import ops
    dev = torch.device("cpu")
    key = ("test-pool", 0)
    ops._ZERO_POOLS.pop(key, None)

    def piece(n):
        return ops._zero_piece(key, (n * 4 + 255) & ~255, n * 4, (n,), torch.float32, dev)

    class Node(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            saved = piece(4)            # e.g. BatchNorm moments written by a kernel
            ctx.save_for_backward(saved)
            return x * w.sum()

        @staticmethod
        def backward(ctx, g):
            (saved,) = ctx.saved_tensors        # raises if the shared counter moved
            gw = piece(3)                       # parameter gradient handed out of the pool
            gw += g.sum()
            return g, gw

    x = torch.ones(5, requires_grad=True)
    w = torch.nn.Parameter(torch.ones(3))
    y = Node.apply(Node.apply(x, w), w).sum()
    y.backward(retain_graph=True)
    y.backward()                                # .grad defined: in-place accumulation into a stolen pool piece
    assert torch.allclose(w.grad, torch.full((3,), 20.0))
    a, b = piece(8), piece(8)
    assert a._base is None and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    v = b._version
    a.add_(1)
    assert b._version == v and float(b.sum()) == 0.0
    ops._ZERO_POOLS.pop(key, None)
