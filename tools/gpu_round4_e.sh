#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04e
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bilinear.py -m gpu -q --tb=short 2>&1 | tail -30 > $OUT/pytest_bilinear.log
tail -3 $OUT/pytest_bilinear.log
timeout 900 python tools/workload_once.py pyramid_train 5 > $OUT/pyramid_train.json 2> $OUT/pyramid_train.err
python - <<PY
import json
r = json.load(open("$OUT/pyramid_train.json"))
for k, v in r["levels"].items():
    print(k, round(v["ms_per_step"], 2), v["fused_path"], {a: round(b, 2) for a, b in v["top_kernels_ms"].items()},
          "mat", v.get("materialised_ms_per_step"), v["sanity"]["out_abs_mean"], v["sanity"]["grad_x_abs_mean"],
          (v.get("sanity_materialised") or {}).get("out_abs_mean"), (v.get("sanity_materialised") or {}).get("grad_x_abs_mean"))
print("total", r["ms_all_levels"])
PY
timeout 600 python tools/workload_once.py pyramid_eval 5 > $OUT/pyramid_eval.json 2> $OUT/pyramid_eval.err
python -c "
import json; r=json.load(open('$OUT/pyramid_eval.json')); print({k:(round(v['ms'],2), v['out_abs_mean']) for k,v in r['levels'].items()}, r['ms_all_levels'])"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_s3dis -o s3dis --output-format csv -- python $ROOT/tools/workload_once.py s3dis_eager 40 > $OUT/s3dis_eager.json 2> $OUT/s3dis_prof.err)
cat $OUT/s3dis_eager.json | cut -c1-400
ls $OUT/prof_s3dis/* | head
find $OUT/prof_s3dis -name "*kernel_stats.csv" -exec cp {} $OUT/s3dis_kernel_stats.csv \;
rm -rf $OUT/prof_s3dis
