cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -3) > gpurun_out/r4b/tests.txt
(MURAL_TRAIN_AUTOGRAD_PARAMS=1 python tools/time_train.py; python tools/time_train.py; python tools/time_train.py; python tools/time_train_graphed.py 2>&1 | tail -1; python tools/host_time_train_calls.py | tail -2) > gpurun_out/r4b/time.txt 2>&1
cat gpurun_out/r4b/tests.txt; grep -v amdgpu.ids gpurun_out/r4b/time.txt
