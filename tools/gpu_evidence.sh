#!/bin/bash
# Full evidence pass on the GPU box: GPU test suite, default bench line, rocprofv3 kernel stats, PMC traffic,
# SQ counters, torchrun (1 rank, RCCL) run of bench.py.  Usage: tools/gpu_evidence.sh <tag>   (writes gpurun_out/<tag>/)
exec < /dev/null
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -6 > $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python tools/show_bench.py $OUT/bench.json | head -3
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-mapping-build > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
python tools/show_bench.py $OUT/bench_torchrun1.json | head -1
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- $BENCH > $OUT/bench_prof.json 2> $OUT/prof.err)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc --output-format csv -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$C.err)
done
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_sq.err)
python profiles/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python tools/pmc_sq.py $OUT/pmc_sq/sq_counter_collection.csv > $OUT/sq_counters.txt 2>&1
# keep the merged output small: raw traces are large
rm -f $OUT/prof/*kernel_trace.csv $OUT/pmc_*/pmc_kernel_trace.csv $OUT/pmc_sq/sq_kernel_trace.csv
ls $OUT
