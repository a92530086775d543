#!/bin/bash
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04x2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_qkv_chain.py tests/test_gpu_pool_modules.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k qkv > $OUT/pytest_full.log 2>&1
tail -3 $OUT/pytest_full.log
timeout 300 python tools/workload_once.py qkv 10 > $OUT/qkv.json 2> $OUT/qkv.err
python -c "
import json; r=json.load(open('$OUT/qkv.json')); print('qkv', r['ms_per_step'], r.get('top_kernels_ms'))"
