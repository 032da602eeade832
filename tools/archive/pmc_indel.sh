#!/bin/bash
# PMC pass over the INDEL forward (tools/bench_indel.py 2048 packed): vector / scalar / LDS instruction counts and busy cycles per kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_indel
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $REPO/tools/bench_indel.py 2048 packed > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $REPO/tools/bench_indel.py 2048 packed > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_indel"
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for f in glob.glob(root + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("mural::(anonymous namespace)::", "").replace("mural::", "").replace("void ", "").split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",): cnt[k] += 1
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if gui <= 0: continue
    simd_cyc = gui * 1024
    rows.append((gui, k, cnt[k], c))
rows.sort(reverse=True)
print("%-42s %5s %9s %7s %7s %7s %7s %7s %7s" % ("kernel", "n", "cycles/n", "VALUbusy", "valu/c", "salu/c", "lds/c", "smem/c", "mfma"))
for gui, k, n, c in rows[:16]:
    simd_cyc = gui * 1024
    print("%-42s %5d %9.0f %7.3f %7.3f %7.3f %7.3f %7.3f %7.3f" % (k, n, gui / max(n, 1), 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cyc,
          4 * c.get("SQ_INSTS_VALU", 0) / simd_cyc, 4 * c.get("SQ_INSTS_SALU", 0) / simd_cyc, 4 * c.get("SQ_INSTS_LDS", 0) / simd_cyc,
          4 * c.get("SQ_INSTS_SMEM", 0) / simd_cyc, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simd_cyc))
PY
