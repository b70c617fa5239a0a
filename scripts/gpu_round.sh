#!/bin/bash
# usage: bash scripts/gpu_round.sh <tag>   -- parity tests, bench, rocprof; outputs under gpurun_out/<tag>/
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > $OUT/rocminfo.txt
nproc > $OUT/host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/host.txt; free -g | head -2 >> $OUT/host.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
# config 2 (1000 x 1 Mbp) quick line, then the headline config
timeout 600 python bench.py --samples 1000 --sites 1000000 --steps 5 --warmup 2 --cpu-seconds 5 2>&1 | tail -3 | tee $OUT/bench_c2.json
timeout 1500 python bench.py --steps 3 --warmup 1 2>&1 | tail -3 | tee $OUT/bench_c3.json
