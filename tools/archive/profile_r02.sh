#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel traces + PMC passes behind the round-2 numbers of bench.py.
#   predict : bench.py (per-window headline) -- kernel stats, SQ / LDS counters, FETCH_SIZE / WRITE_SIZE (separate passes)
#   reuse   : tools/bench_reuse.py (WHICH=reuse) -- kernel stats
#   train   : tools/bench_train.py (13 steps at batch 4096) -- kernel stats, FETCH_SIZE / WRITE_SIZE
#   indel   : tools/bench_indel.py -- kernel stats, FETCH_SIZE / WRITE_SIZE
# Writes gpurun_out/prof_r02/{*.txt,*.json}; the summaries are copied into profiles/ by hand.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 5 --warmup 1 --batch 100000 --no-cpu-baseline --no-train"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/predict/trace -- python3 $REPO/bench.py $BARGS > $OUT/predict_trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/predict/pmcA -- python3 $REPO/bench.py $BARGS > $OUT/predict_pmcA.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/predict/pmcB -- python3 $REPO/bench.py $BARGS > $OUT/predict_pmcB.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/predict/pmcF -- python3 $REPO/bench.py $BARGS > $OUT/predict_pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/predict/pmcW -- python3 $REPO/bench.py $BARGS > $OUT/predict_pmcW.log 2>&1
python3 $REPO/tools/summarize_prof.py $OUT/predict > $OUT/r02_predict_rocprof_summary.txt 2>&1
export WHICH=reuse STEPS=3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reuse/trace -- python3 $REPO/tools/bench_reuse.py > $OUT/reuse_trace.log 2>&1
python3 $REPO/tools/kernel_times.py $OUT/reuse/trace 16 > $OUT/r02_reuse_rocprof_summary.txt 2>&1
for W in train indel; do
  PROG=$REPO/tools/bench_$W.py
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$W/trace -- python3 $PROG > $OUT/${W}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$W/pmcF -- python3 $PROG > $OUT/${W}_pmcF.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$W/pmcW -- python3 $PROG > $OUT/${W}_pmcW.log 2>&1
done
python3 $REPO/tools/archive/profile_r02_facts.py $OUT
ls $OUT
