"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary for profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:8]:
        print("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:60], r.get("Calls"), r.get("TotalDurationNs"),
                                                              r.get("AverageNs"), r.get("Percentage")))
for f in find("trace/**/*kernel_trace.csv"):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        if "snv_towers" in r.get("Kernel_Name", ""):
            print("  dispatch: vgpr=%s accum_vgpr=%s sgpr=%s lds=%s scratch=%s wg=%s grid=%s" % (
                r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size")))
            break
print("== PMC (per dispatch of snv_towers_fused, averaged) ==")
for tag in ("pmcA", "pmcB", "pmcF", "pmcW"):
    for f in find(tag + "/**/*counter_collection.csv"):
        acc, cnt = defaultdict(float), defaultdict(int)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "snv_towers" not in r.get("Kernel_Name", ""):
                    continue
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print("  %-28s %.6g  (n=%d)" % (k, acc[k] / max(cnt[k], 1), cnt[k]))
