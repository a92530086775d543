#!/bin/bash
# SQ counters of the mapping build kernels (two PMC passes, kernel-trace only)
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04z
mkdir -p $OUT
cd $ROOT
cat > /tmp/mb.py <<PY
import json, torch, sys
sys.path.insert(0, "$ROOT")
import bench
r = bench.mapping_build_bench(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("images_per_s", "ms_per_image", "indices_bit_exact_vs_oracle")}))
PY
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_sq -o sq --output-format csv -- python /tmp/mb.py > /dev/null 2> $OUT/pmc_sq.err)
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES --kernel-trace -d $OUT/pmc_in -o sq --output-format csv -- python /tmp/mb.py > /dev/null 2> $OUT/pmc_in.err)
ls $OUT/pmc_sq $OUT/pmc_in
