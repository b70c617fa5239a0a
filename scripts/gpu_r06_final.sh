#!/bin/bash
# the last micro-change of the round (site_lists_kernel's register budget): the list / class / full-size tests again, the driver's line, its kernel trace, config 2
TAG=${1:-r06e}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lists.py tests/test_gpu_site_classes.py tests/test_gpu_golden.py tests/test_gpu_nw_gram.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json; cut -c1-300 $OUT/bench_c3.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$TAG/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/trace.log 2>&1
cp $(find /tmp/$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_c3_kernel_stats.csv
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --samples 1000 --sites 1000000 --steps 20 --warmup 5 --no-extras > $OUT/bench_c2.log 2>&1; tail -1 $OUT/bench_c2.log > $OUT/bench_c2.json; cut -c1-250 $OUT/bench_c2.json
