#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python tools/debug_qkv.py 2>&1 | grep -v amdgpu | cut -c1-330
