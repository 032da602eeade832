export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_indel.py -q -x 2>&1 | grep -E "^E|passed|failed|Error" | head -12) > gpurun_out/r4b/tests.txt
(MURAL_DEBUG_LINEAR_TILE=1 python tools/time_train.py; python tools/time_train.py; python tools/time_train.py; python tools/time_train_graphed.py 2>&1 | tail -1) > gpurun_out/r4b/time.txt 2>&1
cat gpurun_out/r4b/tests.txt; grep -v amdgpu.ids gpurun_out/r4b/time.txt
