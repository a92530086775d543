import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
torch.empty(1 << 30, dtype=torch.uint8, device=dev).zero_()
print("variant", os.environ.get("DVA_COPY_VARIANT", "0"), round(bench.copy_ceiling(dev), 1), "GB/s")
