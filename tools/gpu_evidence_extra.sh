#!/bin/bash
# Round 4: rocprofv3 kernel stats + PMC traffic of the SECONDARY workloads (F-L, S2, fp32, the
# bilinear workloads, the 256 -> 128 level, the reference-sized batch, the batched mapping build).
# Usage: tools/gpu_evidence_extra.sh <tag>   (writes gpurun_out/<tag>/)
exec < /dev/null
TAG=${1:-r04x}
ONLY=${2:-all}          # second argument: space-separated subset of the workload names below
want() { [ "$ONLY" = "all" ] || [[ " $ONLY " == *" $1 "* ]]; }
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
prof() {   # prof <name> <pmc: 0|1> <command...>
  local NAME=$1 PMC=$2; shift 2
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$NAME -o p --output-format csv -- "$@" > $OUT/${NAME}.out 2> $OUT/${NAME}.err)
  find $OUT/prof_$NAME -name "*kernel_stats.csv" -exec cp {} $OUT/${NAME}_kernel_stats.csv \;
  rm -rf $OUT/prof_$NAME
  if [ "$PMC" = "1" ]; then
    mkdir -p $OUT/pmc_$NAME
    for C in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$NAME/pmc_$C -o pmc --output-format csv -- "$@" > /dev/null 2> $OUT/${NAME}_pmc_$C.err)
      find $OUT/pmc_$NAME/pmc_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$NAME/pmc_$C/pmc_counter_collection.csv \; 2>/dev/null
    done
    python profiles/summarize_pmc.py $OUT/pmc_$NAME $OUT/${NAME}_pmc_traffic.json > $OUT/${NAME}_pmc_traffic.txt 2>&1
    rm -rf $OUT/pmc_$NAME
  fi
  head -c 300 $OUT/${NAME}.out; echo
}
want FL && prof FL 1 python $ROOT/tools/workload_once.py F-L 3
want S2 && prof S2 1 python $ROOT/tools/workload_once.py S2 3
want f32 && prof f32 1 python $ROOT/tools/workload_once.py f32 3
want qkv && prof qkv 1 python $ROOT/tools/workload_once.py qkv 3
want nonexact && prof nonexact 0 python $ROOT/tools/workload_once.py nonexact 3
want bilinear_128_32 && prof bilinear_128_32 1 python $ROOT/tools/level_once.py 128 32 3 1
want bilinear_64_64 && prof bilinear_64_64 1 python $ROOT/tools/level_once.py 64 64 3 1
want bilinear_256_128 && prof bilinear_256_128 1 python $ROOT/tools/level_once.py 256 128 3 1
want bilinear_512_256 && prof bilinear_512_256 1 python $ROOT/tools/level_once.py 512 256 3 1
want s3dis && prof s3dis 0 python $ROOT/tools/workload_once.py s3dis_eager 40
want mapping && prof mapping 0 python $ROOT/tools/mapping_bench_once.py
ls $OUT | head -40
