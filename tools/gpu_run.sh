#!/bin/bash
# ONE parametrised GPU-box script (replaces the per-experiment gpu_round*.sh of rounds 3-4).
# Usage (through gpurun):  tools/gpu_run.sh <tag> <job> [<job> ...]     -> writes gpurun_out/<tag>/
# Jobs (run in the order given; every job is bounded by its own `timeout`):
#   tests[:path]        pytest -m gpu (-x) of tests/ or of one path            -> pytest_gpu.log
#   smoke               __graft_entry__.smoke()
#   bench[:args]        python bench.py <args> (default: the driver's line)     -> bench.json
#   headline[:ENV=V,..] bench.py --no-secondary --no-cpu-baseline --no-mapping-build --steps 20 --warmup 5 under the
#                       given environment switches (A/B of kernel variants)    -> headline_<envs>.json
#   torchrun1[:args]    the 1-rank RCCL path, same steps as `headline`          -> torchrun1*.json
#   prof[:args]         rocprofv3 --kernel-trace --stats of the headline        -> kernel_stats.csv
#   workload:<name>[:steps[:ENV=V,..]]   tools/workload_once.py <name>          -> wl_<name>*.json
#   profwl:<name>[:steps]    rocprofv3 kernel stats of tools/workload_once.py   -> <name>_kernel_stats.csv
#   sq[:args]           rocprofv3 --pmc SQ_* of one headline step                -> sq_counters.txt
#   trace:<name>[:steps] ordered kernel list (start order, durations, gaps) of the last step -> <name>_last_step.txt
#   py:<script>[:args]  python <script> <args> (tools/*.py experiments)         -> py_<script>.log
exec < /dev/null
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
HEAD="--no-secondary --no-cpu-baseline --no-mapping-build --no-pmc --steps 20 --warmup 5"
envs() { echo "$1" | tr ',' ' '; }
for JOB in "$@"; do
  KIND=${JOB%%:*}; REST=""; [ "$JOB" != "$KIND" ] && REST=${JOB#*:}
  echo "== $JOB"
  case $KIND in
    tests)
      timeout 1500 python -m pytest ${REST:-tests} -m gpu -q --tb=short 2>&1 | tail -40 > $OUT/pytest_gpu.log
      grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -12 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    bench)
      ( time timeout 1200 python bench.py $REST --detail-file $OUT/bench.detail.json > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
      python tools/show_bench.py $OUT/bench.json | head -4 ;;
    headline)
      NAME=headline_$(echo "${REST:-base}" | tr -c 'A-Za-z0-9_=\n' '_')
      env $(envs "$REST") timeout 600 python bench.py $HEAD --detail-file $OUT/$NAME.detail.json > $OUT/$NAME.json 2> $OUT/$NAME.err
      python tools/show_bench.py $OUT/$NAME.json | head -2 ;;
    torchrun1)
      NAME=torchrun1_$(echo "${REST:-base}" | tr -c 'A-Za-z0-9_=\n' '_')
      timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
          --master-port 29517 bench.py --gpus 1 $HEAD $REST --detail-file $OUT/$NAME.detail.json > $OUT/$NAME.json 2> $OUT/$NAME.err
      python tools/show_bench.py $OUT/$NAME.json | head -1 ;;
    prof)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p --output-format csv -- \
          python $ROOT/bench.py --no-secondary --no-cpu-baseline --no-mapping-build --no-pmc $REST --detail-file $OUT/bench_prof.detail.json > $OUT/bench_prof.json 2> $OUT/prof.err)
      find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; rm -rf $OUT/prof
      head -14 $OUT/kernel_stats.csv | cut -c1-150 ;;
    workload)
      IFS=: read -r NAME STEPS ENVS <<< "$REST"
      FN=wl_${NAME}_$(echo "${ENVS:-base}" | tr -c 'A-Za-z0-9_=\n' '_')
      env $(envs "$ENVS") timeout 900 python tools/workload_once.py $NAME ${STEPS:-10} > $OUT/$FN.json 2> $OUT/$FN.err
      head -c 600 $OUT/$FN.json; echo ;;
    profwl)
      IFS=: read -r NAME STEPS <<< "$REST"
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$NAME -o p --output-format csv -- \
          python $ROOT/tools/workload_once.py $NAME ${STEPS:-3} > $OUT/${NAME}.out 2> $OUT/${NAME}.err)
      find $OUT/prof_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/${NAME}_kernel_stats.csv \; ; rm -rf $OUT/prof_$NAME
      head -12 $OUT/${NAME}_kernel_stats.csv | cut -c1-150 ;;
    sq)
      # SQ counters of the headline (one step): VALU / LDS / wait shares per kernel -> sq_counters.txt
      (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
          --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- python $ROOT/bench.py --no-secondary --no-cpu-baseline \
          --no-mapping-build --no-pmc --steps 1 --warmup 1 $REST > /dev/null 2> $OUT/pmc_sq.err)
      python tools/pmc_sq.py $(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1) > $OUT/sq_counters.txt 2>&1
      rm -rf $OUT/pmc_sq
      head -40 $OUT/sq_counters.txt | cut -c1-170 ;;
    trace)
      # ordered kernel list of the LAST step of a workload: trace:<name>:<steps>
      IFS=: read -r NAME STEPS ENVS <<< "$REST"
      FN=${NAME}$(echo "${ENVS:+_$ENVS}" | tr -c 'A-Za-z0-9_=\n' '_')
      (cd /tmp && env $(envs "$ENVS") timeout 900 rocprofv3 --kernel-trace -d $OUT/trace_$FN -o t --output-format csv -- \
          python $ROOT/tools/workload_once.py $NAME ${STEPS:-3} > $OUT/${FN}_trace.out 2> $OUT/${FN}_trace.err)
      python tools/last_step_trace.py $(find $OUT/trace_$FN -name "*kernel_trace.csv" | head -1) ${STEPS:-3} > $OUT/${FN}_last_step.txt 2>&1
      rm -rf $OUT/trace_$FN
      python tools/last_step_trace.py --sum $OUT/${FN}_last_step.txt | head -${TRACE_TOP:-12} ;;
    py)
      IFS=: read -r SCRIPT ARGS <<< "$REST"
      timeout 900 python $SCRIPT $ARGS > $OUT/py_$(basename $SCRIPT .py).log 2>&1
      tail -25 $OUT/py_$(basename $SCRIPT .py).log ;;
    *) echo "unknown job $JOB" ;;
  esac
done
ls $OUT | head -60
