#!/bin/bash
# the A/B switches of round 4 must leave the suite green: scores-in QKV form, materialised non-exact route
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
DVA_QKV_ONE_KERNEL=0 timeout 600 python -m pytest tests/test_gpu_qkv_chain.py tests/test_gpu_fullsize.py -m gpu -q -k "qkv and not full_size" 2>&1 | tail -2
DVA_LAZY_NONEXACT=0 timeout 600 python -m pytest tests/test_gpu_pool_modules.py tests/test_gpu_data.py -m gpu -q -k "not non_exact" 2>&1 | tail -2
