#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04n
timeout 400 python tools/workload_once.py nonexact 5 > gpurun_out/r04n/nonexact.json 2> gpurun_out/r04n/nonexact.err || tail -20 gpurun_out/r04n/nonexact.err
python -c "
import json; r=json.load(open('gpurun_out/r04n/nonexact.json')); print(json.dumps({k:r[k] for k in ('lazy','materialised','speedup_vs_materialised')}, indent=1)[:2500])"
