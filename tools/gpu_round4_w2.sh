#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04w
mkdir -p $OUT
cd $ROOT
for occ in 3 4; do
  DVA_ATTN_FWD_OCC=$occ python tools/workload_once.py S2 20 > $OUT/s2_occ$occ.json 2> $OUT/s2.err
  python -c "
import json; r=json.load(open('$OUT/s2_occ$occ.json')); print('occ $occ', r['ms_per_step'], r['chain_attn_fwd'])"
done
