set -x
cd $GRAFT_REPO_ROOT
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -20
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -30
