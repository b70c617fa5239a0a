#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "beyond_one_lds_row or config2" 2>&1 | tail -3
timeout 1500 python bench.py --steps 3 --warmup 1 > $OUT/bench_c3.log 2>&1; tail -1 $OUT/bench_c3.log > $OUT/bench_c3.json
python - <<PY
import json
d=json.load(open('$OUT/bench_c3.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['config']['transcluster_ms_per_step'], d['roofline']['kernel_ms'], d['roofline_general']['kernel_ms'], d['roofline']['traffic'])
PY
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
rm -f $OUT/trace/trace_kernel_trace.csv
grep -E "pairsnp_mfma|general_fixup|tc_" $OUT/trace/trace_kernel_stats.csv | cut -c1-70,150-330
