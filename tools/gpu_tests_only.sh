#!/bin/bash
exec < /dev/null
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest ${1:-tests} -m gpu -x -q 2>&1 | tail -8
