#!/bin/bash
# SQ counters of the 256 -> 128 level (wide kernels)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $ROOT/tools/level_once.py 256 128 1 1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_sq.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc_sq/sq_counter_collection.csv > $OUT/sq.txt 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU --kernel-trace -d $OUT/pmc_sq2 -o sq --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_sq2.err)
python $ROOT/tools/pmc_sq.py $OUT/pmc_sq2/sq_counter_collection.csv > $OUT/sq2.txt 2>&1
rm -rf $OUT/pmc_sq $OUT/pmc_sq2
grep -E "kernel|emod" $OUT/sq.txt | cut -c1-180
grep -E "kernel|emod" $OUT/sq2.txt | cut -c1-180
