import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_qkv_chain as T
from test_gpu_chain import rel, make_case, ragged, ragged_long, full32
for sizes_fn, N, C, G, nc, train in [(ragged_long, 1500, 32, 1, 32, False), (ragged_long, 1500, 32, 1, 32, True), (ragged, 3000, 64, 1, 32, False), (ragged_long, 1500, 32, 2, 16, False)]:
    case = make_case(17, N, C, sizes_fn)
    x_main = torch.randn(N, 6, generator=case["gen"])
    ref, m = T.build(case, G, nc, train)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    out_ref, g_ref = T.oracle(case, ref, x_main, False)
    ref.load_state_dict(sd)
    out_amp, g_amp = T.oracle(case, ref, x_main, True)
    out, g, _ = T.run_dev(case, m, x_main, True)
    m.load_state_dict(sd)
    out_b, g_b, _ = T.run_dev(case, m, x_main, False)
    names = ["x", "x_main"] + [n for n, _ in ref.named_parameters()]
    for n, a, b, c, d in zip(names, g, g_ref, g_amp, g_b):
        if n.startswith("G.") or n.startswith("K.") or n.startswith("Q."):
            print(sizes_fn.__name__, N, C, G, train, n, "chain", round(rel(a, b), 4), "autocast", round(rel(c, b), 4), "stored-act path", round(rel(d, b), 4),
                  "values", a.flatten()[:2].tolist(), b.flatten()[:2].tolist(), d.flatten()[:2].tolist())
