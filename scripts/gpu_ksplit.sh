#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
for k in auto 1 2 3 4 8 16 64; do
  echo -n "ksplit $k: " | tee -a $OUT/ksplit.txt
  if [ $k = auto ]; then unset TRACS_KSPLIT; else export TRACS_KSPLIT=$k; fi
  timeout 300 python3 $GRAFT_REPO_ROOT/scripts/prof_target.py $2 $3 3 2>&1 | tail -1 | tee -a $OUT/ksplit.txt
done
