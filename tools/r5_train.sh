# round 5: training-step checks on the GPU box: parity tests of the step, the bench's training leg, one kernel-trace timeline
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5a}
rm -rf $OUT; mkdir -p $OUT
(timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -15) > $OUT/tests.txt
cat $OUT/tests.txt
timeout 600 python tools/archive/r4_train_only.py 300 > $OUT/train.json 2> $OUT/train.err; tail -2 $OUT/train.err; cat $OUT/train.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/archive/r4_train_only.py 20 > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $OUT/trace 30 > $OUT/kernel_times.txt 2>&1; head -45 $OUT/kernel_times.txt
python3 $GRAFT_REPO_ROOT/tools/archive/r4_timeline.py $OUT/trace cw_wfrag 6 > $OUT/timeline.txt 2>&1; tail -3 $OUT/timeline.txt
find $OUT/trace -name "*.csv" -size +20M -delete
