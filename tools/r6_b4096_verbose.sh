export MURAL_HIP_FLAVOR=debug      # development switches are honoured by the debug flavour of the library only
cd $GRAFT_REPO_ROOT
for v in 0 1; do
echo "== MURAL_CW_FULL_GRID=$v"
MURAL_CW_FULL_GRID=$v MURAL_TEST_VERBOSE=1 timeout 600 python -m pytest tests/test_gpu_train.py -q -x -s -k "benchmark_batch_4096" 2>&1 | grep -E "hip-vs-exact|mean distance|passed|failed" | awk '{ if ($0 ~ /hip-vs-exact/) { if ($5+0 > 3e-4) print } else print }'
done
