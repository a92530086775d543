#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04n
timeout 300 python tools/debug_nonexact.py 2>&1 | tail -40
