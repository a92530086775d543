#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04k
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pool_modules.py tests/test_gpu_data.py -m gpu -q --tb=short 2>&1 | tail -5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mapping-build --no-secondary > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
r = json.load(open("$OUT/bench.json")); k = r["kernels"]
print("ms/step", round(r["ms_per_step"], 3), {n: round(v["avg_ms"], 3) for n, v in k.items() if "concat" in n})
PY
timeout 300 python tools/workload_once.py s3dis 30 > $OUT/s3dis.json 2>/dev/null; cut -c1-330 $OUT/s3dis.json
