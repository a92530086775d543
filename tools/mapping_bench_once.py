import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
r = bench.mapping_build_bench(torch.device('cuda:0'))
print(json.dumps(r, indent=1))
