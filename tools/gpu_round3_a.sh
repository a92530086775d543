#!/bin/bash
# round-3 call A: full GPU suite + bench line + stage-5 occupancy A/B + kernel stats
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rm -f gpurun_out/emu_report_r3.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short -x --deselect tests/test_gpu_chain.py 2>&1 | tail -15 > $OUT/pytest_gpu_rest.log
timeout 900 python -m pytest tests/test_gpu_chain.py -q --tb=short 2>&1 | tail -80 > $OUT/pytest_gpu_chain.log
tail -3 $OUT/pytest_gpu_rest.log; tail -3 $OUT/pytest_gpu_chain.log
cp gpurun_out/emu_report_r3.txt $OUT/ 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | head -30
DVA_STAGE5_OCC=3 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mapping-build --no-secondary > $OUT/bench_occ3.json 2> $OUT/bench_occ3.err
python tools/show_bench.py $OUT/bench_occ3.json | head -16
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- $BENCH > $OUT/bench_prof.json 2> $OUT/prof.err)
rm -f $OUT/prof/*kernel_trace.csv
ls $OUT $OUT/prof/* | head -30
