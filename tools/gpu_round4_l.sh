#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04l
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bilinear.py -m gpu -q --tb=short 2>&1 | tail -40 > $OUT/pytest_bilinear.log
tail -3 $OUT/pytest_bilinear.log
timeout 600 python tools/level_once.py 512 256 3 2 > $OUT/l512.json 2> $OUT/l512.err
python -c "
import json; r=json.load(open('$OUT/l512.json')); print(round(r['ms_per_step'],2), r['kernels_ms'], round(r['timed_sum_ms'],2), r['sanity'])"
tail -3 $OUT/l512.err
