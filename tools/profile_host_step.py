import cProfile, pstats, sys, os, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device("cuda", 0)
N, views, C, H, W = 1 << int(os.environ.get("LOG2N", "20")), 32, 64, 64, 128
scene = bench.make_scene(N, views, 32, C, H, W, torch.bfloat16, dev, seed=1)
mods = bench.build_modules(C, dev)
for _ in range(2):
    bench.step(scene, None, mods, torch.bfloat16)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    bench.step(scene, None, mods, torch.bfloat16)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
