#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04s
timeout 600 python tools/fwd_shape_sweep.py > gpurun_out/r04s/fwd_shape_sweep.json 2> gpurun_out/r04s/err.log || tail -20 gpurun_out/r04s/err.log
python -c "
import json; d=json.load(open('gpurun_out/r04s/fwd_shape_sweep.json'))
for r in d['rows']: print(r['shape'][:40].ljust(42), r.get('chain_attn_fwd_eval'), r.get('chain_attn_fwd_train'), r.get('chain_attn_bwd_train'))"
