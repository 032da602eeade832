cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4c
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $OUT/trace 24 | head -40
