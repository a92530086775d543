#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04z
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_transforms.py tests/test_gpu_transforms_golden.py tests/test_gpu_data.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
python - > $OUT/mapping.json 2> $OUT/mapping.err <<PY
import json, torch, bench
from deepviewagg_amd import ops
r = bench.mapping_build_bench(torch.device("cuda:0"))
print(json.dumps(r))
PY
python -c "
import json; r=json.load(open('$OUT/mapping.json')); print({k:v for k,v in r.items() if k in ('images_per_s','ms_per_image','indices_bit_exact_vs_oracle','single_image_calls')}, r['roofline']['frac'])" || tail -5 $OUT/mapping.err
