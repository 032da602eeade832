cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_gpu_train.py -q -k batch_256 2>&1 | grep -E "^E|assert" | head)
python tools/gpu_debug_g7.py S256 2>&1 | grep -v amdgpu | sort -k6 -g | tail -3
