#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/bf16sq2
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary --steps 1 --warmup 1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU --kernel-trace -d $OUT/pmc -o sq --output-format csv -- $BENCH > /dev/null 2> $OUT/pmc.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc/sq_counter_collection.csv > $OUT/counters.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/pmc2 -o sq --output-format csv -- $BENCH > /dev/null 2> $OUT/pmc2.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc2/sq_counter_collection.csv > $OUT/counters2.txt 2>&1
rm -f $OUT/pmc*/sq_kernel_trace.csv
grep "^kernel\|chain3::\|att_\|rows_grad" $OUT/counters.txt | cut -c1-200
grep "^kernel\|chain3::\|att_\|rows_grad" $OUT/counters2.txt | cut -c1-200
