#!/bin/bash
# what the driver runs at round end: the GPU test suite, smoke(), the default bench line (wall-clock)
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/final
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 600 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err ) 2>&1 | grep real
python tools/show_bench.py gpurun_out/final/bench_default.json | head -2
