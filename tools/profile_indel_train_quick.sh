#!/bin/bash
# GPU box: kernel trace of the INDEL training step (tools/bench_indel_train.py) -> gpurun_out/indel_train_quick.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/indel_train_quick
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_indel_train.py 128 10 > $OUT/trace.log 2>&1
python3 $REPO/tools/kernel_times.py $OUT/trace 45 > $REPO/gpurun_out/indel_train_quick.txt 2>&1
tail -2 $OUT/trace.log >> $REPO/gpurun_out/indel_train_quick.txt
