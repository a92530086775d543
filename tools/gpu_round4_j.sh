#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_chain.py -m gpu -q --tb=short 2>&1 | tail -25 > $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 300 python tools/workload_once.py qkv 10 > $OUT/qkv.json 2> $OUT/qkv.err; cut -c1-1200 $OUT/qkv.json
python - <<PY
import sys, torch, json
sys.path.insert(0, "$ROOT")
import bench
dev = torch.device("cuda:0")
r = bench.mapping_build_bench(dev)
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)})[:900])
print(json.dumps(bench.neighborhood_bench(dev))[:700])
PY
