cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
(WHICH=cw timeout 600 python tools/gpu_debug_conv32_cl.py) > gpurun_out/r4a/parity.txt 2>&1
echo "rc=$?" >> gpurun_out/r4a/parity.txt
export TICKETS=0
(WHICH=cw python tools/time_conv32_cl.py) > gpurun_out/r4a/time.txt 2>&1
for d in 1 2 4 8 7 15 16 32 64 128 112 240; do WHICH=cw MURAL_DEBUG_CW=$d python tools/time_conv32_cl.py 2>&1 | grep "L= 134"; done >> gpurun_out/r4a/time.txt
tail -4 gpurun_out/r4a/parity.txt; cat gpurun_out/r4a/time.txt
