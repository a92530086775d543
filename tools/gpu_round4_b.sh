#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bilinear.py tests/test_gpu_fullsize.py -m gpu -q --tb=short \
    -k "reference_fixture or anchor_scatter or fused_bilinear_full or just_below" 2>&1 | tail -40 > $OUT/pytest_new.log
tail -3 $OUT/pytest_new.log
