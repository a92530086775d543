#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python tools/debug_wide.py > $OUT/debug_wide.log 2>&1
tail -12 $OUT/debug_wide.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=long -k "just_below" 2>&1 | tail -40 > $OUT/pytest_limit.log
timeout 600 python tools/workload_once.py s3dis 30 > $OUT/s3dis.json 2> $OUT/s3dis.err; cat $OUT/s3dis.json | cut -c1-900
