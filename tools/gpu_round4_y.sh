#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04y
mkdir -p $OUT
cd $ROOT
python tools/strong_bn_diff.py 20 8 > $OUT/strong_bn_diff.json 2> $OUT/strong.err || tail -20 $OUT/strong.err
cat $OUT/strong_bn_diff.json | head -60
