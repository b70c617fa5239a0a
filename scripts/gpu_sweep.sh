#!/bin/bash
# usage: bash scripts/gpu_sweep.sh <tag> <samples> <sites> <variant ids...>
TAG=$1; NS=$2; NL=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do
  echo -n "variant $v: " | tee -a $OUT/sweep.txt
  TRACS_TILE_VARIANT=$v timeout 300 python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $NS $NL 3 2>&1 | tail -1 | tee -a $OUT/sweep.txt
done
