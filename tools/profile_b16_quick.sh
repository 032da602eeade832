#!/bin/bash
# GPU box: kernel trace of the 16-site dense call loop (tools/bench_b16.py) -> gpurun_out/b16_quick.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/b16_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_b16.py > $OUT/trace.log 2>&1
python3 $REPO/tools/kernel_times.py $OUT/trace 12 > $REPO/gpurun_out/b16_quick.txt 2>&1
tail -2 $OUT/trace.log >> $REPO/gpurun_out/b16_quick.txt
