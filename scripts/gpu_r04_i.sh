#!/bin/bash
# round 4: does cutting the list walk into cache-sized site segments pay?
TAG=${1:-r04i}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python scripts/probe_nn_segments.py > $OUT/segments.log 2>&1; cat $OUT/segments.log
