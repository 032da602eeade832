"""Condense the rocprofv3 output of tools/archive/profile_r03.sh into the summaries and JSON facts that bench.py reads (profiles/)."""
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
    open(os.path.join(root, "r03_train_step_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"hbm_bytes_per_step": 2 * fb + wb, "fetch_bytes_per_step_raw": fb, "write_bytes_per_step": wb,
               "launches_per_step": steady if steady else n / STEPS, "launches_per_step_all_13_steps": n / STEPS,
               "kernel_us_per_step": tot / STEPS,
               "source": "rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/bench_train.py (13 steps, batch "
                         "4096, incl. input encode, loss, clip, Adam); FETCH_SIZE doubled per the gfx950 correction for 16-byte-per-lane streaming "
                         "reads (MI355X_MICROARCH.md, HBM section); summary: profiles/r03_train_step_rocprof_summary.txt"},
              open(os.path.join(root, "r03_train_step.json"), "w"), indent=1)

# ---- INDEL forward: tools/bench_indel.py 2048 packed runs 7 forwards of 2048 positions through the packed entry
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
    open(os.path.join(root, "r03_indel_forward_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"hbm_bytes_per_position": 2 * fb + wb, "kernel_us_per_2048_positions": tot / 7.0,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/bench_indel.py; FETCH_SIZE doubled per the gfx950 "
                         "correction; summary: profiles/r03_indel_forward_rocprof_summary.txt"},
              open(os.path.join(root, "r03_indel_forward.json"), "w"), indent=1)

# ---- predict: HBM traffic of the tower kernel per site
f, _ = counter_sum("predict", "pmcF", "FETCH_SIZE", "snv_tower")
w, _ = counter_sum("predict", "pmcW", "WRITE_SIZE", "snv_tower")
n_launch = sum(1 for r in trace_rows("predict") if "snv_tower" in r["Kernel_Name"])
if f and n_launch:
    sites = 6 * 100000.0          # bench.py --steps 5 --warmup 1 --batch 100000
    json.dump({"kernel": "snv_tower_wave", "fetch_size_kib_total": f, "write_size_kib_total": w, "sites": sites, "launches": n_launch,
               "hbm_bytes_per_site": (2 * f + w) * 1024 / sites,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 5 --warmup 1 --batch 100000`; FETCH_SIZE "
                       "doubled per the gfx950 correction for 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM section)"},
              open(os.path.join(root, "hbm_traffic.json"), "w"), indent=1)
# ---- predict: held clock and MFMA-pipe busy of the tower kernel (per template instance and over all launches)
def per_kernel_counter(tag, name):
    d = defaultdict(list)
    for f in glob.glob(os.path.join(root, "predict", tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "snv_tower" in r["Kernel_Name"]:
                d[r["Kernel_Name"].split("(")[0].replace("void mural::", "")].append(float(r["Counter_Value"]))
    return d


dur = defaultdict(list)      # durations of the SAME pass that counted GRBM_GUI_ACTIVE (pmcB): clock = cycles / time of one run
for f in glob.glob(os.path.join(root, "predict", "pmcB", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "snv_tower" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0].replace("void mural::", "")].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
gui, busy = per_kernel_counter("pmcB", "GRBM_GUI_ACTIVE"), per_kernel_counter("pmcA", "SQ_VALU_MFMA_BUSY_CYCLES")
valu, mfma = per_kernel_counter("pmcA", "SQ_INSTS_VALU"), per_kernel_counter("pmcA", "SQ_INSTS_MFMA")
if dur and gui and busy:
    out, lines = {}, ["== held clock and MFMA-pipe busy of the tower kernel (GRBM_GUI_ACTIVE / 8 XCDs = cycles of the launch; busy = "
                      "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x those cycles); durations from the kernel trace of the GRBM pass) =="]
    tot_busy = tot_cyc = tot_ns = tot_valu = tot_mfma = 0.0
    for k in sorted(dur):
        if k not in gui or k not in busy:
            continue
        n = len(dur[k])
        ns, cyc, b = sum(dur[k]) / n, sum(gui[k]) / len(gui[k]) / 8.0, sum(busy[k]) / len(busy[k])
        v, m = sum(valu[k]) / len(valu[k]), sum(mfma[k]) / len(mfma[k])
        out[k] = {"launches": n, "avg_us": ns / 1e3, "held_clock_ghz": cyc / ns, "mfma_pipe_busy": b / (1024.0 * cyc),
                  "valu_insts_per_mfma_incl_mfma": v / m, "valu_insts_per_mfma_excl_mfma": v / m - 1.0}
        lines.append("  %-32s n=%3d avg %8.1f us  clock %.3f GHz  MFMA busy %.3f  SQ_INSTS_VALU / SQ_INSTS_MFMA %.2f (%.2f without the MFMAs themselves)"
                     % (k, n, ns / 1e3, cyc / ns, b / (1024.0 * cyc), v / m, v / m - 1.0))
        tot_busy += b * n; tot_cyc += cyc * n; tot_ns += ns * n; tot_valu += v * n; tot_mfma += m * n
    out["all_launches"] = {"held_clock_ghz": tot_cyc / tot_ns, "mfma_pipe_busy": tot_busy / (1024.0 * tot_cyc),
                           "valu_insts_per_mfma_incl_mfma": tot_valu / tot_mfma, "valu_insts_per_mfma_excl_mfma": tot_valu / tot_mfma - 1.0,
                           "source": "rocprofv3 --pmc passes of tools/archive/profile_r03.sh over `bench.py --steps 5 --warmup 1 --batch 100000`; "
                                     "summary: profiles/r03_predict_rocprof_summary.txt"}
    lines.append("  all launches: clock %.3f GHz, MFMA busy %.3f, SQ_INSTS_VALU / SQ_INSTS_MFMA %.2f (%.2f without the MFMAs themselves)"
                 % (tot_cyc / tot_ns, tot_busy / (1024.0 * tot_cyc), tot_valu / tot_mfma, tot_valu / tot_mfma - 1.0))
    json.dump(out, open(os.path.join(root, "r03_predict_pmc.json"), "w"), indent=1)
    with open(os.path.join(root, "r03_predict_rocprof_summary.txt"), "a") as fh:
        fh.write("\n".join(lines) + "\n")
print("facts written to", root)
