#!/bin/bash
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rm -f gpurun_out/emu_report_r3.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -60 > $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cp gpurun_out/emu_report_r3.txt $OUT/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mapping-build --no-secondary > $OUT/bench.json 2> $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | head -24
