"""cProfile of the host side of bench.step() on a tiny scene (launch-bound): where the per-step CPU time goes."""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
N, views, C = 1 << 12, 32, 64
wl = sys.argv[1] if len(sys.argv) > 1 else "S1"
scene = bench.make_scene(N, views, 32, C, 64, 128, torch.bfloat16, dev, seed=1, workload=wl)
mods = bench.build_modules(C, dev)
for _ in range(10):
    bench.step(scene, None, mods, torch.bfloat16)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    bench.step(scene, None, mods, torch.bfloat16)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
