#!/bin/bash
# Re-take ONLY the PMC traffic passes of the headline (FETCH_SIZE / WRITE_SIZE, one step each) and stamp them with the hash
# of the kernel sources: what `bench.py`'s roofline.traffic needs after a change under deepviewagg_amd/csrc or include/.
# Usage: tools/gpu_stamp_refresh.sh <tag>  -> gpurun_out/<tag>/pmc_traffic.json (copy to profiles/pmc_traffic_latest.json)
exec < /dev/null
TAG=${1:-stamp}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-mapping-build --no-secondary"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_$C -o pmc --output-format csv -- $BENCH --steps 1 --warmup 1 > /dev/null 2> $OUT/pmc_$C.err)
  find $OUT/pmc_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$C/pmc_counter_collection.csv \; 2>/dev/null
done
python profiles/summarize_pmc.py $OUT $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
grep -E "bucket_rows|attn_fwd|layer_bwd_kernel<5|_stamp|calibration" $OUT/pmc_traffic.txt | head -8
