cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
(timeout 1200 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3) > gpurun_out/r4b/tests.txt
(MURAL_TRAIN_NO_FOLD=1 python tools/time_train.py; python tools/time_train.py; python tools/time_train.py) > gpurun_out/r4b/time.txt 2>&1
cat gpurun_out/r4b/tests.txt; grep -v amdgpu.ids gpurun_out/r4b/time.txt
