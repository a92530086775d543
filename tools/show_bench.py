"""Pretty-print a bench record (kernel table): the full record (`bench_detail.json` / `<name>.detail.json`) when it is
there, else the compact stdout line."""
import json, os, sys
path = sys.argv[1]
d = json.loads(open(path).read().strip().splitlines()[-1])
detail = path[:-5] + ".detail.json" if path.endswith(".json") else path + ".detail"
if "kernels" not in d and os.path.exists(detail):
    d = json.load(open(detail))
print("ms/step", round(d["ms_per_step"], 3), " value", round(d["value"] / 1e6, 2), "M points/s")
tot = 0
if "kernels" in d:
    for k, v in d["kernels"].items():
        per = v["launches_per_step"] if "launches_per_step" in v else v["launches"] / d["steps"]
        tot += v["avg_ms"] * per
        print(f"{k:32s} {v['avg_ms']:.3f} ms  {v['GBps']:7.0f} GB/s  x{per:.0f}/step")
    print("sum of timed kernels per step:", round(tot, 3), "ms")
else:
    for k, v in d.get("kernels_ms_per_step", {}).items():
        print(f"{k:32s} {v:.3f} ms/step")
for k in ("roofline", "roofline_view_gather_attention"):
    if d.get(k):
        print(k, {a: d[k][a] for a in ("kernel", "frac", "avg_launch_ms") if a in d[k]})
