#!/bin/bash
# staged several-points walk of attn_fwd: parity tests of the ragged cases + S2 timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04w
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_pool_modules.py tests/test_gpu_qkv_chain.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python tools/workload_once.py S2 20 > $OUT/s2.json 2> $OUT/s2.err
python - <<PY
import json
r=json.load(open('$OUT/s2.json'))
print('S2 ms/step', r['ms_per_step'])
k=r.get('kernels',{})
for n,v in sorted(k.items(), key=lambda kv:-kv[1]['avg_ms'])[:12]:
    print(f"  {n:28s} {v['avg_ms']:.3f}")
print({kk:vv for kk,vv in r.items() if 'roofline' in kk})
PY
