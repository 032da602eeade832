"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) median / min / max duration and share of the total."""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("mural::(anonymous namespace)::", "").replace("mural::", "")
    key = (n.split("(")[0][:34], r.get("Grid_Size_X") or r.get("Grid_Size"))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    v.sort()
    print("%-36s grid=%-8s n=%4d med=%7.1f us min=%7.1f max=%7.1f  share=%4.1f%%" % (k[0], k[1], len(v), v[len(v) // 2], v[0], v[-1],
                                                                                100 * sum(v) / tot))
