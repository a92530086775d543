#!/bin/bash
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pool_modules.py tests/test_gpu_chain.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
bash tools/gpu_round4_n3.sh | grep -E "ms_per_step|rows_grad|speedup"
