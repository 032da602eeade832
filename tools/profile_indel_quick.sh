#!/bin/bash
# GPU box: kernel trace of the INDEL forward (tools/bench_indel.py) -> gpurun_out/indel_quick.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/indel_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_indel.py > $OUT/trace.log 2>&1
python3 $REPO/tools/kernel_times.py $OUT/trace 40 > $REPO/gpurun_out/indel_quick.txt 2>&1
