#!/bin/bash
# usage: bash scripts/gpu_prof_cmd.sh <tag> <python script and args...>  -- rocprofv3 kernel trace + stats of one command;
# only the stats summary comes back (gpurun_out/<tag>_kernel_stats.csv)
TAG=$1; shift
OUT=/tmp/prof_$TAG
mkdir -p $OUT $GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 "$@" > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats.csv; head -28 "$f" | cut -c1-230; else echo "no stats file"; ls -R $OUT | head; fi
