#!/bin/bash
# mapping build: parity tests, bench entry, rocprofv3 kernel stats (csv); every command bounded, no stdin reads
exec < /dev/null
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04z
mkdir -p $OUT
cd $ROOT
timeout 400 python -m pytest tests/test_gpu_mapping.py tests/test_gpu_transforms_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1
tail -2 $OUT/pytest.log
cat > /tmp/mb.py <<PY
import json, torch, sys
sys.path.insert(0, "$ROOT")
import bench
r = bench.mapping_build_bench(torch.device("cuda:0"))
print(json.dumps({k: r[k] for k in ("images_per_s", "ms_per_image", "indices_bit_exact_vs_oracle")}))
PY
timeout 120 python /tmp/mb.py 2> $OUT/mb.err | tail -1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o mapping --output-format csv -- python /tmp/mb.py > $OUT/prof.log 2>&1)
f=$(ls $OUT/prof2/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" $OUT/mapping_kernel_stats.csv; fi
