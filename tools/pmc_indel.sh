#!/bin/bash
# GPU box: SQ counters of the INDEL forward kernels (tools/bench_indel.py) -> gpurun_out/indel_pmc.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/indel_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $REPO/tools/bench_indel.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/b -- python3 $REPO/tools/bench_indel.py > $OUT/b.log 2>&1
python3 - <<PY > $REPO/gpurun_out/indel_pmc.txt
import csv, glob, collections
for tag in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("mural::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in acc:
        if "convblock_kernel<8" in k or "conv1d_kernel<16, 7" in k or "convblock_mfma_kernel<16" in k:
            print(k, {c: "%.3g" % (v / cnt[(k, c)]) for c, v in acc[k].items()})
PY
