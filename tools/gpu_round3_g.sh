#!/bin/bash
TAG=${1:-r03g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_bilinear.py tests/test_gpu_ops.py tests/test_gpu_data.py -q --tb=short 2>&1 | tail -30 > $OUT/tests.log
tail -4 $OUT/tests.log
timeout 600 python -m pytest tests/test_gpu_bilinear.py -q --tb=short -s 2>&1 | grep "rel err" | cut -c1-600 > $OUT/bil_report.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mapping-build --no-secondary"
for cfg in "--interpolate" "--interpolate --materialize" "--interpolate --channels 128 --out-channels 32" "--interpolate --channels 128 --out-channels 32 --materialize"; do
  name=$(echo $cfg | tr -d ' -')
  timeout 600 $B $cfg > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "== $cfg"; python tools/show_bench.py $OUT/bench_$name.json | head -16
done
