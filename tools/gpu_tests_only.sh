#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tests
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -15 > $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
