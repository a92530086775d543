"""Where does the split plan start to pay?  plan + rows gradient, permutation form against split form, over the view count
(R = 2^18 rows or R = views / 32, C = 64 bf16, G = 4).  Prints one JSON line."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepviewagg_amd import ops

DEV = "cuda:0"


def bench(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    out = []
    gen = torch.Generator().manual_seed(1)
    for V in (1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
        for R in (1 << 18, max(V // 32, 1024)):
            N, C, G = V // 8, 64, 4
            row_idx = torch.randint(0, R, (V,), generator=gen, dtype=torch.int32).to(DEV)
            rec = torch.zeros(V, 4, dtype=torch.int32)
            rec[:, 0] = torch.randint(0, N, (V,), generator=gen, dtype=torch.int32)
            rec[:, 1:3] = torch.randn(V, 4, generator=gen).to(torch.bfloat16).view(torch.int32)
            rec[:, 3] = row_idx.cpu()
            rec = rec.to(DEV)
            gout = torch.randn(N, C, generator=gen).to(torch.bfloat16).to(DEV)
            st = torch.cuda.current_stream().cuda_stream

            def run(split):
                ops.SPLIT_PLAN, ops.SPLIT_PLAN_MIN_VIEWS = split, 0
                plan, counts = ops.row_plan(row_idx, R)
                return ops.rows_grad_rec16(gout, plan, rec.clone(), R, C, G, torch.bfloat16, st)
            t_clone = bench(lambda: rec.clone())
            out.append({"views": V, "rows": R, "permutation_ms": bench(lambda: run(False)) - t_clone,
                        "split_ms": bench(lambda: run(True)) - t_clone})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
