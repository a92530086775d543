"""Multi-process CPU test of the N > 1 path (gloo, world_size 2): tile sharding + the flat gradient
bucket all-reduce used by bench.py --gpus N.  Sum-of-gradients equivalence: the averaged sharded
gradient equals the single-process gradient of the mean loss over both tiles."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepviewagg_amd.parallel import GradientBucket, shard_mapping, tile_partition


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
    # a bucket of ONE contiguous fp32 gradient is reduced where it lies (no flatten / unflatten copies)
    big = torch.nn.Parameter(torch.zeros(1000))
    big.grad = torch.full((1000,), float(rank + 1))
    held = big.grad
    single = GradientBucket([big], bucket_bytes=1024)                     # four chunks
    single.start(average=True)
    single.finish()
    assert big.grad is held and single._inplace
    torch.testing.assert_close(big.grad, torch.full((1000,), 1.5))         # mean of 1 and 2
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


# ---- the real view-pooling maths per tile (oracle on the CPU): a scene split into two tiles, each rank pools
#      ITS points (the views of a point travel with the point), gradients summed by the bucket
def _pool_scene():
    g = torch.Generator().manual_seed(3)
    N, C = 600, 16
    sizes = torch.randint(0, 6, (N,), generator=g)
    csr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)])
    V = int(csr[-1])
    xyz = torch.rand(N, 3, generator=g) * torch.tensor([9.0, 4.0, 2.0])
    return dict(csr=csr, xyz=xyz, x_mod=torch.randn(V, C, generator=g), x_map=torch.rand(V, 8, generator=g),
                w=torch.randn(N, C, generator=g), C=C)


def _pool_module(C):
    from oracle import pooling_oracle as O
    torch.manual_seed(5)
    # eval-mode BatchNorm: per-tile batch statistics are a property of the reference's per-process BatchNorm,
    # the sum-of-gradients identity below holds for the tile-independent maths
    return O.GroupBimodalCSRPool(in_map=8, in_mod=C, num_groups=4, use_num=True).eval()


def _pool_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = _pool_scene()
    tiles = tile_partition(sc["xyz"], world)
    csr_t, views = shard_mapping(sc["csr"], tiles[rank])
    model = _pool_module(sc["C"])
    bucket = GradientBucket(model.parameters(), bucket_bytes=4096)      # several buckets
    pooled = model(None, sc["x_mod"][views], sc["x_map"][views], csr_t)
    (pooled * sc["w"][tiles[rank]]).sum().backward()
    bucket.start(average=False)
    bucket.finish()
    if rank == 0:
        torch.save([None if p.grad is None else p.grad.clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_mapping_keeps_views_with_points():
    sc = _pool_scene()
    tiles = tile_partition(sc["xyz"], 3)
    seen = torch.zeros(int(sc["csr"][-1]), dtype=torch.long)
    for t in tiles:
        csr_t, views = shard_mapping(sc["csr"], t)
        assert int(csr_t[-1]) == views.shape[0]
        seen[views] += 1
        sizes = sc["csr"][1:] - sc["csr"][:-1]
        assert torch.equal(csr_t[1:] - csr_t[:-1], sizes[t])
        # view k of tile point i is view k of the original point
        i = int((sizes[t] > 1).nonzero()[0])
        assert int(views[csr_t[i] + 1]) == int(sc["csr"][t[i]]) + 1
    assert bool((seen == 1).all())


def test_view_pooling_tiles_sum_to_scene_gradient_world2(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "pool_grads.pt")
    mp.spawn(_pool_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    sc = _pool_scene()
    model = _pool_module(sc["C"])
    pooled = model(None, sc["x_mod"], sc["x_map"], sc["csr"])
    (pooled * sc["w"]).sum().backward()
    for g, p in zip(got, model.parameters()):
        if p.grad is None:
            assert g is None or float(g.abs().max()) == 0
            continue
        torch.testing.assert_close(g, p.grad, rtol=1e-4, atol=1e-5)
