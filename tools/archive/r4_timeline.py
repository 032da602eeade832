"""Timeline of one steady-state step in a rocprofv3 kernel trace: start offset, duration, queue of every launch, the union of busy
time and the idle gaps.  python tools/archive/r4_timeline.py <trace dir> [marker] [steps back]"""
import csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "relayout_multi"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
s = rows[idx[-back - 1]:idx[-back]]
t0 = int(s[0]["Start_Timestamp"])
queues = {}
busy_end = t0
idle = 0.0
overlap_area = 0.0
for r in s:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
    n = re.sub(r"\(anonymous namespace\)::|mural::|void ", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)
    gap = (a - busy_end) / 1e3
    if gap > 0:
        idle += gap
    print("%8.1f %7.1f q%d %s%-44s grid %d x %s" % ((a - t0) / 1e3, (b - a) / 1e3, q, "  " * q, n[:44],
                                                 int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]) +
          (("   <- %.1f us idle before" % gap) if gap > 1.0 else ""))
    busy_end = max(busy_end, b)
    overlap_area += (b - a) / 1e3
span = (int(rows[idx[-back]]["Start_Timestamp"]) - t0) / 1e3
print("step span %.1f us, kernel time %.1f us, idle %.1f us, %d launches" % (span, overlap_area, idle, len(s)))
