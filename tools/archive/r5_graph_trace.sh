# round 5: GPU-side timeline of ONE replay of the graphed training step (no host enqueue delays): which stream is the critical path
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5g}
rm -rf $OUT; mkdir -p $OUT
python tools/time_train_graphed.py 200 2>&1 | tail -1 > $OUT/graphed.txt; cat $OUT/graphed.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/time_train_graphed.py 12 > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/archive/r4_timeline.py $OUT/trace cw_wfrag 4 > $OUT/timeline.txt 2>&1; tail -3 $OUT/timeline.txt
find $OUT/trace -name "*.csv" -size +20M -delete
