#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04r
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_qkv_chain.py -m gpu -q --tb=short 2>&1 | tail -40 > $OUT/pytest_qkv.log
tail -25 $OUT/pytest_qkv.log | cut -c1-400
timeout 600 python tools/workload_once.py qkv 10 > $OUT/qkv.json 2> $OUT/qkv.err; cut -c1-1300 $OUT/qkv.json; tail -3 $OUT/qkv.err
