cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
(timeout 1200 python -m pytest tests/test_gpu_train.py -q -k reproducible 2>&1 | tail -30) > gpurun_out/r4b/tests.txt
(MURAL_TRAIN_CONV_CL=1 timeout 1200 python -m pytest tests/test_gpu_train.py -q -k reproducible 2>&1 | tail -30) >> gpurun_out/r4b/tests.txt
cat gpurun_out/r4b/tests.txt
