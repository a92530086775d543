#!/bin/bash
# SQ counters of the fp32 step (MFMA utilisation of the fp32 chain)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/f32sq
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --dtype f32 --no-cpu-baseline --no-mapping-build --no-secondary --steps 1 --warmup 1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- $BENCH > /dev/null 2> $OUT/pmc_sq.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc_sq/sq_counter_collection.csv > $OUT/sq_counters.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace -d $OUT/pmc_in -o sq --output-format csv -- $BENCH > /dev/null 2> $OUT/pmc_in.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc_in/sq_counter_collection.csv > $OUT/inst_counters.txt 2>&1
rm -f $OUT/pmc_*/sq_kernel_trace.csv
head -40 $OUT/sq_counters.txt; head -40 $OUT/inst_counters.txt
