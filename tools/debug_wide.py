#!/usr/bin/env python
"""Round-4 debugging aid: per-tensor gradient errors of the fused bilinear path against the fp32 oracle and the oracle
under autocast for a list of (sizes, N, C_in, C_out, G, train) cases."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_bilinear as T  # noqa: E402
from test_gpu_chain import rel, ragged, ragged_long, full32  # noqa: E402

CASES = [(full32, 256, 64, 128, 1, True), (full32, 512, 64, 128, 1, True), (ragged, 900, 64, 128, 1, True),
         (full32, 256, 64, 128, 4, True), (full32, 256, 64, 64, 1, True), (full32, 256, 64, 128, 2, True),
         (full32, 256, 64, 128, 1, False)]
for sizes_fn, N, C_in, C_out, G, train in CASES:
    case = T.make_case(21, N, C_in, sizes_fn)
    w = torch.randn(N, C_out, generator=case["gen"])
    ref, m = T.build(case, C_out, G, train)
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    out_ref, g_ref = T.oracle(case, ref, w, autocast=False)
    ref.load_state_dict(sd)
    out_amp, g_amp = T.oracle(case, ref, w, autocast=True)
    out, g, used = T.run_dev(case, m, w, fused=True)
    m.load_state_dict(sd)
    out_b, g_b, _ = T.run_dev(case, m, w, fused=False)
    names = ["x"] + [n for n, _ in ref.named_parameters()]
    keys = ("x", "E_map.mlp_elt_1.0.0.weight", "E_map.mlp_elt_2.1.0.weight", "E_mod.1.0.weight", "E_score.weight",
            "E_score.bias", "G.weight")
    rep = {n: (round(rel(a, b), 4), round(rel(c, b), 4), round(rel(d, b), 4))
           for n, a, b, c, d in zip(names, g, g_ref, g_amp, g_b) if n in keys and b is not None}
    print(sizes_fn.__name__, N, C_in, C_out, G, train, used["fn"], "out", round(rel(out, out_ref), 4),
          "(ours, autocast oracle, materialised path):", rep, flush=True)
