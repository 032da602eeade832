cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_train -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_train"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-44:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:8]:
    m = v["SQ_INSTS_MFMA"]
    print("%-46s n=%4d valu/mfma %.2f lds/mfma %.2f salu/mfma %.2f mfma_busy %.0f%%" % (k, cnt[k], (v["SQ_INSTS_VALU"] - m) / max(m, 1), v["SQ_INSTS_LDS"] / max(m, 1), v["SQ_INSTS_SALU"] / max(m, 1), 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["GRBM_GUI_ACTIVE"] / 8 * 1024, 1)))
PY
