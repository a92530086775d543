#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04v
mkdir -p $OUT
cd $ROOT
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
python tools/show_bench.py $OUT/bench_default.json | head -2
python -c "
import json; r=json.load(open('$OUT/bench_default.json')); print({k:(round(v.get('ms_per_step',v.get('ms_all_levels',v.get('ms_per_step_eager',0))),2)) for k,v in r['workloads'].items()}); print(r['cpu_baseline']['value'], r['cpu_baseline']['sample'][:80])"
