export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
cd /tmp && export TMPDIR=/tmp
for v in 0 2 4 6 14; do
MURAL_DEBUG_MLP=$v rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/mlp_prof$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 100000 --no-cpu-baseline --no-train > /dev/null 2>&1
echo "dbg $v: $(python3 $GRAFT_REPO_ROOT/tools/train_kernel_stats.py $GRAFT_REPO_ROOT/gpurun_out/mlp_prof$v/run_results.db 4 6 | grep -i mlp)"
done
