#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bilinear.py -m gpu -q --tb=short 2>&1 | tail -60 > $OUT/pytest_bilinear.log
tail -3 $OUT/pytest_bilinear.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -k "bilinear" 2>&1 | tail -30 > $OUT/pytest_fullsize.log
tail -3 $OUT/pytest_fullsize.log
