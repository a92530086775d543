#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k qkv 2>&1 | tail -5
