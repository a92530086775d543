#!/bin/bash
# round 4, call A: the new fused-bilinear verification tests + the rows-gradient channel-slab A/B on F-L (C = 512)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bilinear.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -x \
    -k "reference_fixture or anchor_scatter or fused_bilinear_full or just_below" 2>&1 | tail -25 > $OUT/pytest_new.log
tail -3 $OUT/pytest_new.log
for SLAB in 0 64 128 256; do
  DVA_ROWS_GRAD_SLAB=$SLAB timeout 300 python tools/workload_once.py F-L 10 > $OUT/fl_slab_$SLAB.json 2> $OUT/fl_slab_$SLAB.err
  python - <<PY
import json
r = json.load(open("$OUT/fl_slab_$SLAB.json"))
print("slab $SLAB: ms/step", round(r["ms_per_step"], 2), {k: round(v, 3) for k, v in r["top_kernels_ms"].items()}, r["sanity"])
PY
done
