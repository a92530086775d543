#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -k "bilinear or qkv" 2>&1 | tail -25 | cut -c1-300
