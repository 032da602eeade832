"""Per-kernel summary of a rocprofv3 rocpd database of the training step (tools/bench_train.py runs 13 steps)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 13.0
c = db.cursor()
q = "select name, grid_x, count(*), sum(end-start)/1e3, min(end-start)/1e3 from kernels group by name, grid_x order by 4 desc"
rows = list(c.execute(q))
tot = sum(r[3] for r in rows)
print("kernel time: %.1f us total, %.1f us per step (%g steps), %d launches per step" % (tot, tot / steps, steps, sum(r[2] for r in rows) / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = r[0].replace("mural::(anonymous namespace)::", "").replace("mural::", "").split("(")[0][:50]
    print("%-52s g=%-8s n=%4d tot=%9.1f avg=%8.1f min=%8.1f %5.1f%%" % (n, r[1], r[2], r[3], r[3] / r[2], r[4], 100 * r[3] / tot))
