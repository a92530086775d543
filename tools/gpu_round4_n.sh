#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd $ROOT
( time python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mapping-build > $OUT/b1.json 2>/dev/null ) 2>&1 | grep real
python -c "
import json; r=json.load(open('$OUT/b1.json')); w=r['workloads']; print('no-mapping-build: F-L', round(w['F-L']['ms_per_step'],2), 'S2', round(w['S2']['ms_per_step'],2), 'f32', round(w['f32']['ms_per_step'],2))"
( time python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b2.json 2>/dev/null ) 2>&1 | grep real
python -c "
import json; r=json.load(open('$OUT/b2.json')); w=r['workloads']; print('with mapping build: F-L', round(w['F-L']['ms_per_step'],2), 'S2', round(w['S2']['ms_per_step'],2), 'f32', round(w['f32']['ms_per_step'],2))"
