"""Condense the rocprofv3 output of tools/archive/profile_r02.sh into the summaries and JSON facts that bench.py reads (profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def trace_rows(sub):
    f = glob.glob(os.path.join(root, sub, "trace", "*", "*kernel_trace.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def counter_sum(sub, tag, name, match=None):
    tot, per = 0.0, defaultdict(float)
    for f in glob.glob(os.path.join(root, sub, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            if match and match not in k:
                continue
            tot += float(r["Counter_Value"])
            per[k.replace("mural::(anonymous namespace)::", "").replace("mural::", "").split("(")[0][:44]] += float(r["Counter_Value"])
    return tot, per


def kernel_table(rows, units, unit_name, top=40):
    d = defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"].replace("mural::(anonymous namespace)::", "").replace("mural::", "").split("(")[0][:50]
        d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    lines = ["kernel time: %.1f us total, %.1f us per %s (%g), %.1f launches per %s" % (tot, tot / units, unit_name, units,
                                                                                       sum(len(v) for v in d.values()) / units, unit_name)]
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
        lines.append("%-52s n=%5d tot=%10.1f avg=%8.1f min=%8.1f %5.1f%%" % (k, len(v), sum(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
    return lines, tot, sum(len(v) for v in d.values())


# ---- training step: tools/bench_train.py runs 13 steps (3 warm-up + 10 timed), all of them profiled
STEPS = 13.0
rows = trace_rows("train")
if rows:
    lines, tot, n = kernel_table(rows, STEPS, "step")
    # launches of a step in steady state: the average over all 13 steps also counts the one-off launches of the first step (the
    # optimizer's ~300 state fills, lazy initialisations); between two consecutive head_fwd_kernel launches there is exactly one step
    # (+ the 8 launches of the bench loop's own input generation: arange, strand bits, k-mer / one-hot encoders)
    srt = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
    heads = [i for i, r in enumerate(srt) if "head_fwd_kernel" in r["Kernel_Name"]]
    gaps = sorted(b - a for a, b in zip(heads[3:], heads[4:]))
    steady = gaps[len(gaps) // 2] if gaps else None
    lines.insert(1, "steady state (median over the last %d steps): %s launches per step, input generation of the bench loop included"
                 % (len(gaps), steady))
    f, perf = counter_sum("train", "pmcF", "FETCH_SIZE")
    w, perw = counter_sum("train", "pmcW", "WRITE_SIZE")
    fb, wb = f * 1024 / STEPS, w * 1024 / STEPS
    lines.append("== HBM counters per step: FETCH_SIZE %.3f GB raw (x2 for 16-byte-per-lane streaming reads on gfx950 = %.3f GB), WRITE_SIZE %.3f GB"
                 % (fb / 1e9, 2 * fb / 1e9, wb / 1e9))
    for k in sorted(set(perf) | set(perw), key=lambda k: -(2 * perf.get(k, 0) + perw.get(k, 0)))[:14]:
        lines.append("  %-46s fetch(x2) %8.1f MB/step  write %8.1f MB/step" % (k, 2 * perf.get(k, 0) * 1024 / STEPS / 1e6, perw.get(k, 0) * 1024 / STEPS / 1e6))
    open(os.path.join(root, "r02_train_step_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"hbm_bytes_per_step": 2 * fb + wb, "fetch_bytes_per_step_raw": fb, "write_bytes_per_step": wb,
               "launches_per_step": steady if steady else n / STEPS, "launches_per_step_all_13_steps": n / STEPS,
               "kernel_us_per_step": tot / STEPS,
               "source": "rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_train.py (13 steps, batch "
                         "4096, incl. input encode, loss, clip, Adam); FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane streaming "
                         "reads (MI355X_MICROARCH.md, HBM section); summary: profiles/r02_train_step_rocprof_summary.txt"},
              open(os.path.join(root, "r02_train_step.json"), "w"), indent=1)

# ---- INDEL forward: tools/bench_indel.py runs 7 forwards of 2048 positions
rows = trace_rows("indel")
if rows:
    POS = 7.0 * 2048
    lines, tot, n = kernel_table(rows, 7.0, "forward of 2048 positions")
    f, perf = counter_sum("indel", "pmcF", "FETCH_SIZE")
    w, perw = counter_sum("indel", "pmcW", "WRITE_SIZE")
    fb, wb = f * 1024 / POS, w * 1024 / POS
    lines.append("== HBM counters per position: FETCH_SIZE %.1f KB raw (x2 = %.1f KB), WRITE_SIZE %.1f KB" % (fb / 1e3, 2 * fb / 1e3, wb / 1e3))
    for k in sorted(set(perf) | set(perw), key=lambda k: -(2 * perf.get(k, 0) + perw.get(k, 0)))[:12]:
        lines.append("  %-46s fetch(x2) %8.1f KB/pos  write %8.1f KB/pos" % (k, 2 * perf.get(k, 0) * 1024 / POS / 1e3, perw.get(k, 0) * 1024 / POS / 1e3))
    open(os.path.join(root, "r02_indel_forward_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"hbm_bytes_per_position": 2 * fb + wb, "kernel_us_per_2048_positions": tot / 7.0,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_indel.py; FETCH_SIZE doubled per the gfx950 "
                         "correction; summary: profiles/r02_indel_forward_rocprof_summary.txt"},
              open(os.path.join(root, "r02_indel_forward.json"), "w"), indent=1)

# ---- predict: HBM traffic of the tower kernel per site
f, _ = counter_sum("predict", "pmcF", "FETCH_SIZE", "snv_towers_fused")
w, _ = counter_sum("predict", "pmcW", "WRITE_SIZE", "snv_towers_fused")
n_launch = sum(1 for r in trace_rows("predict") if "snv_towers_fused" in r["Kernel_Name"])
if f and n_launch:
    sites = 6 * 100000.0          # bench.py --steps 5 --warmup 1 --batch 100000
    json.dump({"kernel": "snv_towers_fused", "fetch_size_kib_total": f, "write_size_kib_total": w, "sites": sites, "launches": n_launch,
               "hbm_bytes_per_site": (2 * f + w) * 1024 / sites,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1 --batch 100000`; FETCH_SIZE "
                       "doubled per the gfx950 correction for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section)"},
              open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
print("facts written to", root)
