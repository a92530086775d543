"""Multi-process CPU test of the N > 1 path (gloo, world_size 2): tile sharding + the flat gradient
bucket all-reduce used by bench.py --gpus N.  Sum-of-gradients equivalence: the averaged sharded
gradient equals the single-process gradient of the mean loss over both tiles."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepviewagg_amd.parallel import GradientBucket, tile_partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))


def _data():
    g = torch.Generator().manual_seed(1)
    xyz = torch.rand(400, 3, generator=g) * torch.tensor([8.0, 3.0, 2.0])
    feats = torch.randn(400, 6, generator=g)
    return xyz, feats


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xyz, feats = _data()
    tiles = tile_partition(xyz, world)
    model = _model()
    bucket = GradientBucket(model.parameters())
    loss = model(feats[tiles[rank]]).square().mean()
    loss.backward()
    bucket.reduce(average=True)
    if rank == 0:
        torch.save([p.grad.clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_partition_properties():
    xyz, _ = _data()
    tiles = tile_partition(xyz, 4)
    allidx = torch.cat(tiles)
    assert allidx.shape[0] == 400 and torch.equal(allidx.sort().values, torch.arange(400))
    assert max(len(t) for t in tiles) - min(len(t) for t in tiles) <= 1
    # slabs along x (the longest axis): tile i lies left of tile i+1
    for a, b in zip(tiles[:-1], tiles[1:]):
        assert xyz[a, 0].max() <= xyz[b, 0].min()
    assert all(torch.equal(t, t.sort().values) for t in tiles)


def test_gradient_bucket_allreduce_world2(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    xyz, feats = _data()
    tiles = tile_partition(xyz, world)
    model = _model()
    loss = sum(model(feats[t]).square().mean() for t in tiles) / world
    loss.backward()
    for g, p in zip(got, model.parameters()):
        torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-6)
