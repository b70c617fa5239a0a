#!/bin/bash
# round 4: shapes beside the bench one with the final kernels -- 50 000 samples x 100 kbp (three column chunks per row), config 4, config 5
TAG=${1:-r04s}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/check_large_n.py > $OUT/check_large_n.json 2> $OUT/check_large_n.err; tail -c 1500 $OUT/check_large_n.json; tail -3 $OUT/check_large_n.err
timeout 900 python scripts/bench_config5.py > $OUT/bench_config5.json 2> $OUT/c5.err; tail -c 600 $OUT/bench_config5.json; tail -2 $OUT/c5.err
timeout 900 python scripts/bench_config4.py > $OUT/bench_config4.json 2> $OUT/c4.err; tail -c 600 $OUT/bench_config4.json; tail -2 $OUT/c4.err
