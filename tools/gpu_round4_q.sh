#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04q
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python $ROOT/tools/level_once.py 512 256 3 1 > $OUT/l.out 2> $OUT/l.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/l512_kernel_stats.csv \;
rm -rf $OUT/prof
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/l512_kernel_stats.csv")))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>3s} avg {float(r['AverageNs']) / 1e6:8.3f} ms")
PY
