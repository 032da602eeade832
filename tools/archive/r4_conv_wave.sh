export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
(WHICH=cw timeout 600 python tools/gpu_debug_conv32_cl.py) > gpurun_out/r4a/parity.txt 2>&1
echo "rc=$?" >> gpurun_out/r4a/parity.txt
(WHICH=cw python tools/time_conv32_cl.py) > gpurun_out/r4a/time.txt 2>&1
for d in 15 240; do WHICH=cw MURAL_DEBUG_CW=$d python tools/time_conv32_cl.py 2>&1 | grep "L= 134"; done >> gpurun_out/r4a/time.txt
(MURAL_TRAIN_CONV_CL=1 python tools/time_train.py; python tools/time_train.py; python tools/time_train.py) >> gpurun_out/r4a/time.txt 2>&1
tail -3 gpurun_out/r4a/parity.txt; grep -v amdgpu.ids gpurun_out/r4a/time.txt
