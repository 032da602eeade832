"""Launch list of the INDEL training step from a rocprofv3 kernel trace of tools/time_indel_train_graphed.py: kernels of the LAST
graph replay (the trace's last N launches, N = launches between the two last occurrences of the step's first kernel), by name."""
import collections, csv, glob, os, re, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|mural::|void ", "", r["Kernel_Name"]))
# the replayed step is periodic: find the period from the tail
names = [name(r) for r in rows]
tail = names[-4000:]
period = next(p for p in range(100, 1500) if tail[-p:] == tail[-2 * p:-p])
s = rows[-period:]
c = collections.defaultdict(lambda: [0, 0.0])
for r in s:
    c[name(r)[:60]][0] += 1
    c[name(r)[:60]][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("launches per replayed step: %d, kernel time %.1f us, span %.1f us" % (period, sum(v[1] for v in c.values()),
      (int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e3))
for k, v in sorted(c.items(), key=lambda kv: -kv[1][0]):
    print("%4d %8.1f us  %s" % (v[0], v[1], k))
