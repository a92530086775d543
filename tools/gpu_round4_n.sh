#!/bin/bash
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pool_modules.py -m gpu -x -q -k "segment_max or non_exact or golden" > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
bash tools/gpu_round4_n3.sh
