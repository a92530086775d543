#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bilinear.py -m gpu -q --tb=short -x 2>&1 | tail -5 > $OUT/pytest_bilinear.log
tail -2 $OUT/pytest_bilinear.log
for OCC in 1 2; do
  DVA_EMOD_ABWD128_OCC=$OCC timeout 300 python tools/level_once.py 256 128 5 2 > $OUT/l256_occ$OCC.json 2> $OUT/l256_occ$OCC.err
  python -c "
import json; r=json.load(open('$OUT/l256_occ$OCC.json')); print('occ $OCC', round(r['ms_per_step'],2), r['kernels_ms'], round(r['timed_sum_ms'],2))"
done
timeout 300 python tools/level_once.py 128 64 5 2 > $OUT/l128_64.json 2> $OUT/l128_64.err
python -c "
import json; r=json.load(open('$OUT/l128_64.json')); print(round(r['ms_per_step'],2), r['kernels_ms'], round(r['timed_sum_ms'],2))"
timeout 600 python tools/workload_once.py pyramid_eval 5 > $OUT/pyramid_eval.json 2> $OUT/pyramid_eval.err
python -c "
import json; r=json.load(open('$OUT/pyramid_eval.json')); print({k:(round(v['ms'],2)) for k,v in r['levels'].items()}, r['ms_all_levels'])"
