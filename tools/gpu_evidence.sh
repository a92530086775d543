#!/bin/bash
# Full evidence pass on the GPU box: GPU test suite, smoke, the driver's bench line, the 1-rank RCCL run, the launcher on a
# 1-GPU box, rocprofv3 kernel stats, PMC traffic (stamped with the kernel-source hash), SQ counters.
# Usage: tools/gpu_evidence.sh <tag>   (writes gpurun_out/<tag>/; copy what is to be judged into profiles/<tag>_*)
exec < /dev/null
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -8 > $OUT/pytest_gpu.log
tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
# profiler passes first: the PMC table of THIS build is in place (profiles/pmc_traffic_latest.json of the box's copy)
# when the bench lines are taken, so that their roofline objects carry `traffic`
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary --no-pmc"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- $BENCH > $OUT/bench_prof.json 2> $OUT/prof.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc --output-format csv -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$C.err)
  find $OUT/pmc_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$C/pmc_counter_collection.csv \; 2>/dev/null
done
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_sq.err)
python profiles/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python tools/pmc_sq.py $(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1) > $OUT/sq_counters.txt 2>&1
cp $OUT/pmc_traffic.json $ROOT/profiles/pmc_traffic_latest.json
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench.detail.json > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python tools/show_bench.py $OUT/bench.json | head -3
HEAD="--steps 20 --warmup 5 --no-cpu-baseline --no-mapping-build --no-secondary --no-pmc"
timeout 600 python bench.py --gpus 1 $HEAD --detail-file $OUT/bench_plain.detail.json > $OUT/bench_plain.json 2> /dev/null
for V in "" "--no-standin"; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 1 $HEAD $V --detail-file $OUT/bench_torchrun1$(echo $V | tr -d ' ' | tr - _).detail.json > $OUT/bench_torchrun1$(echo $V | tr -d ' ' | tr - _).json 2> $OUT/bench_torchrun1.err
done
python - <<PY
import json
a = json.load(open("$OUT/bench_plain.json"))["ms_per_step"]
b = json.load(open("$OUT/bench_torchrun1.json"))
c = json.load(open("$OUT/bench_torchrun1__no_standin.json"))["ms_per_step"]
print(f"plain {a:.3f} ms | 1-rank RCCL, pooling bucket only {c:.3f} ms | + 112 MB stand-in bucket {b['ms_per_step']:.3f} ms "
      f"(allreduce {b.get('allreduce_ms')}, exposed {b.get('exposed_ms')})")
PY
# the launcher on a box with ONE device: two ranks start, the second has no device, the run fails loudly
timeout 300 python bench.py --gpus 2 --steps 1 --warmup 0 > $OUT/bench_gpus2.out 2> $OUT/bench_gpus2.err; echo "bench.py --gpus 2 on this box: rc=$?" | tee $OUT/bench_gpus2.rc
grep -E "launching 2 ranks|has no HIP device" $OUT/bench_gpus2.err | head -3 | tee -a $OUT/bench_gpus2.rc
# keep the merged output small: raw traces are large
rm -rf $OUT/prof $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_sq
grep -E "rows_grad|attn_fwd|_stamp|calibration" $OUT/pmc_traffic.txt | head -8
ls $OUT
