#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04t
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_qkv_chain.py tests/test_gpu_chain.py tests/test_gpu_pool_modules.py -m gpu -q --tb=short 2>&1 | tail -5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- python $ROOT/tools/workload_once.py qkv 5 > $OUT/q.out 2> $OUT/q.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/qkv_kernel_stats.csv \;
rm -rf $OUT/prof
python - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/qkv_kernel_stats.csv")))
steps = 7
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print("sum per step ms", round(tot, 2))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f"{r['Name'][:90]:90s} x{int(r['Calls'])/steps:5.1f} {float(r['TotalDurationNs'])/steps/1e6:7.3f} ms/step")
PY
