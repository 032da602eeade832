#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + HBM byte counters of the SNV training step (tools/bench_train.py, 13 steps).
# usage: tools/profile_train.sh <tag>
set -u
TAG=${1:-t1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/train_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- python3 $REPO/tools/bench_train.py > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmcF -- python3 $REPO/tools/bench_train.py > $OUT/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmcW -- python3 $REPO/tools/bench_train.py > $OUT/pmcW.log 2>&1
grep "train step" $OUT/trace.log
python3 $REPO/tools/train_kernel_stats.py $OUT/trace/run_results.db 13 45 > $OUT/summary.txt
python3 - $OUT >> $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
tot = {}
per = defaultdict(lambda: defaultdict(float))
for tag, name in (("pmcF", "FETCH_SIZE"), ("pmcW", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(root, tag, "**", "*counter_collection.csv"), recursive=True):
        s = 0.0
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                s += float(r["Counter_Value"])
                per[r["Kernel_Name"].split("(")[0][-40:]][name] += float(r["Counter_Value"])
        tot[name] = s
print("== HBM counters over the whole run (13 steps, KiB) ==", tot)
if len(tot) == 2:
    f, w = tot["FETCH_SIZE"] * 1024 / 13, tot["WRITE_SIZE"] * 1024 / 13
    print("per step: FETCH_SIZE %.3f GB (raw; streaming reads are under-counted up to 2x on gfx950), WRITE_SIZE %.3f GB" % (f / 1e9, w / 1e9))
    print("per step lower bound %.3f GB, with the 2x read correction %.3f GB" % ((f + w) / 1e9, (2 * f + w) / 1e9))
for k, v in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))[:12]:
    print("  %-42s fetch %.1f MB/step  write %.1f MB/step" % (k, v["FETCH_SIZE"] * 1024 / 13 / 1e6, v["WRITE_SIZE"] * 1024 / 13 / 1e6))
PY
cat $OUT/summary.txt
