cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
(WHICH=cw timeout 600 python tools/gpu_debug_conv32_cl.py | tail -3) > gpurun_out/r4b/tests.txt 2>&1
(timeout 1200 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -3) >> gpurun_out/r4b/tests.txt
(WHICH=cw python tools/time_conv32_cl.py; WHICH=cw WFRAG=0 python tools/time_conv32_cl.py) > gpurun_out/r4b/time.txt 2>&1
(python tools/time_train.py; python tools/time_train.py) >> gpurun_out/r4b/time.txt 2>&1
cat gpurun_out/r4b/tests.txt; grep -v amdgpu.ids gpurun_out/r4b/time.txt
