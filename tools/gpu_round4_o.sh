#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04o
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bilinear.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -k "bilinear" 2>&1 | tail -30 > $OUT/pytest_bilinear.log
tail -3 $OUT/pytest_bilinear.log
for L in "128 32" "64 64" "256 128" "512 256"; do
  for M in 1 0; do
  DVA_ANCHOR_ORDER_STATS=$M timeout 600 python tools/level_once.py $L 3 2 > $OUT/l.json 2> $OUT/l.err
  python -c "
import json; r=json.load(open('$OUT/l.json')); k=r['kernels_ms']; print('anchor-order $M', r['level'], round(r['ms_per_step'],2), {n:k[n] for n in ('emod_stats1','row_plan','bilinear_anchor_sum') if n in k})"
  done
done
