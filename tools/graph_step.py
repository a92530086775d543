"""Eager vs HIP-graph replay of the bench step: python tools/graph_step.py [--log2-points 20]
Captures one forward + backward of bench.step() in a torch.cuda.CUDAGraph (= hipGraph) after warm-up and times
replays against eager steps; checks that the replayed gradients equal the eager ones."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-points", type=int, default=20)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    N, views, C, H, W = 1 << args.log2_points, 32, 64, 64, 128
    scene = bench.make_scene(N, views, 32, C, H, W, torch.bfloat16, dev, seed=1234)
    mods = bench.build_modules(C, dev)

    def one():
        return bench.step(scene, None, mods, torch.bfloat16, lazy=True)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / args.steps * 1e3
    g_eager = scene["x"].grad.clone()
    # capture on a side stream (torch requirement), static inputs = the scene tensors
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            one()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    scene["x"].grad = None
    with torch.cuda.graph(graph):
        one()
    torch.cuda.synchronize()
    g_cap = scene["x"].grad
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / args.steps * 1e3
    err = float((g_cap.float() - g_eager.float()).abs().max() / (g_eager.float().abs().max() + 1e-30))
    print(f"eager {eager:.3f} ms/step   graph replay {replay:.3f} ms/step   max rel grad difference {err:.2e}")


if __name__ == "__main__":
    main()
