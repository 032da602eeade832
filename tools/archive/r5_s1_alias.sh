cd $GRAFT_REPO_ROOT
for a in 0 1; do
  if [ $a = 1 ]; then export MURAL_DEBUG_S1_ALIAS=1; else unset MURAL_DEBUG_S1_ALIAS; fi
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/s1p$a
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s1p$a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --batch 100000 --no-cpu-baseline --no-train > /tmp/s1p$a.log 2>&1
  echo alias $a; python3 $GRAFT_REPO_ROOT/tools/kernel_times.py /tmp/s1p$a 8 | head -9
done
