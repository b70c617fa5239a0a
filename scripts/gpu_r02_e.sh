#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_dirichlet_priors.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
TRACS_BENCH_PARTIAL=0.005 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | tail -1 > $OUT/bench_c3_partial.json; cut -c1-300 $OUT/bench_c3_partial.json; grep -o '"kernel_ms": [0-9.]*' $OUT/bench_c3_partial.json
cd /tmp
TRACS_BENCH_PARTIAL=0.005 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_partial -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/trace_partial.log 2>&1
rm -f $OUT/trace_partial/trace_kernel_trace.csv
head -6 $OUT/trace_partial/trace_kernel_stats.csv | cut -c1-160
cd $GRAFT_REPO_ROOT
timeout 900 python scripts/bench_config5.py --samples 100000 > $OUT/config5.log 2>&1; tail -1 $OUT/config5.log | tee $OUT/bench_config5.json
timeout 900 python scripts/bench_distance_cli.py > $OUT/bench_distance_cli.log 2>&1; tail -3 $OUT/bench_distance_cli.log
