#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel traces + PMC passes behind the round-5 numbers.  Sections (arguments; default all):
#   train   : tools/bench_train_sym.py (13 steps at batch 4096, the bench's own symbol-route composition) -- kernel stats, FETCH_SIZE / WRITE_SIZE, two SQ passes
#   predict : bench.py (per-window headline)                -- kernel stats, two SQ passes, FETCH_SIZE / WRITE_SIZE
#   indel   : tools/bench_indel.py 2048 packed              -- kernel stats, FETCH_SIZE / WRITE_SIZE, two SQ passes
#   indel_train : tools/bench_indel_train.py                -- kernel stats, FETCH_SIZE / WRITE_SIZE, two SQ passes
#   reuse   : tools/bench_reuse.py (WHICH=reuse)            -- kernel stats
# Counters are collected in their own runs with --kernel-trace only.  Writes gpurun_out/prof_r05/; tools/archive/profile_r05_facts.py condenses
# it into the summaries copied to profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SECTIONS=${@:-train predict indel indel_train reuse}
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
SQB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
prof() {   # name, program + args
  local W=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/trace -- python3 "$@" > $OUT/${W}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$W/pmcF -- python3 "$@" > $OUT/${W}_pmcF.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$W/pmcW -- python3 "$@" > $OUT/${W}_pmcW.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $OUT/$W/pmcA -- python3 "$@" > $OUT/${W}_pmcA.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $OUT/$W/pmcB -- python3 "$@" > $OUT/${W}_pmcB.log 2>&1
}
for S in $SECTIONS; do
  case $S in
    train) prof train $REPO/tools/bench_train_sym.py ;;
    predict) prof predict $REPO/bench.py --steps 5 --warmup 1 --batch 100000 --no-cpu-baseline --no-train ;;
    indel) prof indel $REPO/tools/bench_indel.py 2048 packed ;;
    indel_train)
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/indel_train/pmcF -- python3 $REPO/tools/bench_indel_train.py > $OUT/indel_train_pmcF.log 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/indel_train/pmcW -- python3 $REPO/tools/bench_indel_train.py > $OUT/indel_train_pmcW.log 2>&1
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/indel_train/trace -- python3 $REPO/tools/bench_indel_train.py > $OUT/indel_train_trace.log 2>&1
      rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d $OUT/indel_train/pmcA -- python3 $REPO/tools/bench_indel_train.py > $OUT/indel_train_pmcA.log 2>&1
      rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d $OUT/indel_train/pmcB -- python3 $REPO/tools/bench_indel_train.py > $OUT/indel_train_pmcB.log 2>&1 ;;
    reuse)
      export WHICH=reuse STEPS=3
      rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reuse/trace -- python3 $REPO/tools/bench_reuse.py > $OUT/reuse_trace.log 2>&1 ;;
  esac
done
python3 $REPO/tools/archive/profile_r05_facts.py $OUT
ls $OUT
