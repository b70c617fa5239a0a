#!/bin/bash
# the whole GPU suite + smoke, log to gpurun_out/<tag>/
TAG=${1:-r05suite}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
timeout 3300 python -m pytest tests -q -m gpu -x > gpurun_out/$TAG/pytest_gpu_full_suite.log 2>&1; tail -5 gpurun_out/$TAG/pytest_gpu_full_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
PARTIAL=0.005 timeout 600 python scripts/probe_single_pass.py 2>&1 | grep -E "stages|kernels" | tail -2
