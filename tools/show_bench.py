"""Pretty-print the bench JSON line (kernel table)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), " value", round(d["value"] / 1e6, 2), "M points/s")
tot = 0
for k, v in d["kernels"].items():
    per = v["launches_per_step"] if "launches_per_step" in v else v["launches"] / d["steps"]
    tot += v["avg_ms"] * per
    print(f"{k:32s} {v['avg_ms']:.3f} ms  {v['GBps']:7.0f} GB/s  x{per:.0f}/step")
print("sum of timed kernels per step:", round(tot, 3), "ms")
for k in ("roofline", "roofline_view_gather_attention"):
    if k in d:
        print(k, {a: d[k][a] for a in ("kernel", "frac", "avg_launch_ms") if a in d[k]})
