"""Per-kernel-class time of the last full step in a rocprofv3 kernel trace (argument: the trace directory)."""
import collections, csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "relayout_multi"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
s = rows[idx[-2]:idx[-1]]
detail = len(sys.argv) > 3
c = collections.defaultdict(lambda: [0, 0.0])
for r in s:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = re.sub(r"\(anonymous namespace\)::|mural::|void ", "", r["Kernel_Name"])
    if detail:
        n = re.sub(r"\(.*", "", n)
        print("%6.1f %-46s grid %s/%s/%s" % (d, n[:46], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"]))
    n = re.sub(r"<.*", "", n)
    n = re.sub(r"\(.*", "", n)
    c[n][0] += 1
    c[n][1] += d
tot = sum(v[1] for v in c.values())
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-40s n=%3d %7.1f us %4.1f%%" % (k[:40], v[0], v[1], 100 * v[1] / tot))
print(len(s), "launches", "%.1f us" % tot)
