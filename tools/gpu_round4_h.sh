#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for M in 0 1; do
  DVA_ROWS_GRAD_PLANREC=$M timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mapping-build --no-secondary > $OUT/bench_planrec$M.json 2> $OUT/bench_planrec$M.err
  python - <<PY
import json
r = json.load(open("$OUT/bench_planrec$M.json"))
k = r.get("kernels", {})
print("planrec $M: ms/step", round(r["ms_per_step"], 3), {n: round(v.get("avg_launch_ms", 0), 3) for n, v in k.items() if n in ("view_gather_rows_grad", "chain_attn_bwd", "plan_inverse", "row_plan")})
PY
done
DVA_ROWS_GRAD_PLANREC=1 timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -x -k "chain" 2>&1 | tail -4
DVA_ROWS_GRAD_PLANREC=1 timeout 300 python tools/workload_once.py F-L 10 > $OUT/fl_planrec1.json 2>/dev/null
timeout 300 python tools/workload_once.py F-L 10 > $OUT/fl_planrec0.json 2>/dev/null
python - <<PY
import json
for m in (0, 1):
    r = json.load(open("$OUT/fl_planrec%d.json" % m))
    print("F-L planrec", m, round(r["ms_per_step"], 2), {k: round(v, 3) for k, v in r["top_kernels_ms"].items()})
PY
