#!/bin/bash
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04z
mkdir -p $OUT
cd $ROOT
cat > /tmp/mb.py <<PY
import json, torch, sys
sys.path.insert(0, "$ROOT")
import bench
r = bench.mapping_build_bench(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("images_per_s", "ms_per_image", "indices_bit_exact_vs_oracle")}))
PY
for nb in 2048 4096 8192 65536; do
  (cd /tmp && export TMPDIR=/tmp && DVA_RASTER_BLOCKS=$nb timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_$nb -o mapping --output-format csv -- python /tmp/mb.py > $OUT/prof_$nb.log 2>&1)
  grep images_per_s $OUT/prof_$nb.log | cut -c1-100
  python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_$nb/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "tile_raster" in r["Name"]:
        print("blocks $nb tile_raster avg us", float(r["AverageNs"]) / 1e3, "calls", r["Calls"])
PY
done
