#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04p
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bilinear.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -k "bilinear" 2>&1 | tail -30 > $OUT/pytest_bilinear.log
tail -3 $OUT/pytest_bilinear.log
for L in "512 256" "256 128"; do
  timeout 600 python tools/level_once.py $L 3 2 > $OUT/l.json 2> $OUT/l.err
  python -c "
import json; r=json.load(open('$OUT/l.json')); print(r['level'], round(r['ms_per_step'],2), {k:v for k,v in r['kernels_ms'].items() if v>1.4}, r['sanity']['out_abs_mean'], r['sanity']['grad_x_abs_mean'])"
done
