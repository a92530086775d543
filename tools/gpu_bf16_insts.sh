#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bf16inst
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary --steps 1 --warmup 1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_CVT SQ_WAVES --kernel-trace -d $OUT/pmc_in -o sq --output-format csv -- $BENCH > /dev/null 2> $OUT/pmc_in.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc_in/sq_counter_collection.csv > $OUT/inst_counters.txt 2>&1
rm -f $OUT/pmc_in/sq_kernel_trace.csv
grep "^kernel\|chain::" $OUT/inst_counters.txt | cut -c1-190
