#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 300 python tools/level_once.py 256 128 5 2 > $OUT/l256.json 2> $OUT/l256.err
python -c "
import json; r=json.load(open('$OUT/l256.json')); print(round(r['ms_per_step'],2), r['kernels_ms'], round(r['timed_sum_ms'],2))"
timeout 900 python -m pytest tests/test_gpu_bilinear.py -m gpu -q --tb=short -x -k "128" 2>&1 | tail -3
