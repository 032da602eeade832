"""Condense the rocprofv3 output of tools/archive/profile_r04.sh (gpurun_out/prof_r04) into per-section summaries: kernel table from the
trace, HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (FETCH doubled per the gfx950 correction for 16-byte-per-lane streaming
reads, MI355X_MICROARCH.md), and per-kernel SQ facts from the two SQ passes (MFMA pipe busy, vector / LDS instructions per MFMA,
wait fractions, LDS bank-conflict rate).  Writes r04_<section>_rocprof_summary.txt and r04_<section>.json next to the raw output."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
UNITS = {"train": (13.0, "step"), "indel": (7.0, "forward of 2048 positions"), "predict": (6.0, "step of 100k sites"),
         "indel_train": (13.0, "step"), "reuse": (4.0, "run")}


def short(k):
    return k.replace("mural::(anonymous namespace)::", "").replace("mural::", "").replace("void ", "").split("(")[0][:52]


def trace_rows(sub):
    f = glob.glob(os.path.join(root, sub, "trace", "*", "*kernel_trace.csv"))
    return list(csv.DictReader(open(f[0]))) if f else []


def counters(sub, tag):
    per = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for f in glob.glob(os.path.join(root, sub, tag, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                n[k] += 1
    return per, n


for sub, (units, uname) in UNITS.items():
    rows = trace_rows(sub)
    if not rows:
        continue
    d = defaultdict(list)
    for r in rows:
        d[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    nl = sum(len(v) for v in d.values())
    lines = ["kernel time: %.1f us total, %.1f us per %s (%g), %.1f launches per %s" % (tot, tot / units, uname, units, nl / units, uname)]
    facts = {"kernel_us_per_unit": tot / units, "launches_per_unit_all": nl / units, "unit": uname}
    if sub == "train":
        srt = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
        heads = [i for i, r in enumerate(srt) if "head_fwd_kernel" in r["Kernel_Name"]]
        gaps = sorted(b - a for a, b in zip(heads[3:], heads[4:]))
        steady = gaps[len(gaps) // 2] if gaps else None
        lines.append("steady state (median over the last %d steps): %s launches per step, input generation of the bench loop included" % (len(gaps), steady))
        facts["launches_per_step"] = steady
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:36]:
        lines.append("%-54s n=%5d tot=%10.1f avg=%8.1f min=%8.1f %5.1f%%" % (k, len(v), sum(v), sum(v) / len(v), min(v), 100 * sum(v) / tot))
    pf, _ = counters(sub, "pmcF")
    pw, _ = counters(sub, "pmcW")
    if pf or pw:
        f = sum(v["FETCH_SIZE"] for v in pf.values()) * 1024 / units
        w = sum(v["WRITE_SIZE"] for v in pw.values()) * 1024 / units
        lines.append("== HBM counters per %s: FETCH_SIZE %.3f GB raw (x2 for 16-byte-per-lane streaming reads on gfx950 = %.3f GB), WRITE_SIZE %.3f GB; total %.3f GB"
                     % (uname, f / 1e9, 2 * f / 1e9, w / 1e9, (2 * f + w) / 1e9))
        facts.update({"hbm_bytes_per_unit": 2 * f + w, "fetch_bytes_raw_per_unit": f, "write_bytes_per_unit": w})
        for k in sorted(set(pf) | set(pw), key=lambda k: -(2 * pf[k]["FETCH_SIZE"] + pw[k]["WRITE_SIZE"]))[:14]:
            lines.append("  %-52s fetch(x2) %8.1f MB  write %8.1f MB per %s" % (k, 2 * pf[k]["FETCH_SIZE"] * 1024 / units / 1e6, pw[k]["WRITE_SIZE"] * 1024 / units / 1e6, uname))
    pa, na = counters(sub, "pmcA")
    pb, nb = counters(sub, "pmcB")
    if pa or pb:
        lines.append("== SQ counters per kernel (separate --pmc passes; busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); waits and"
                     " active fractions are of SQ_WAVE_CYCLES; conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)")
        sq = {}
        for k in sorted(set(pa) | set(pb), key=lambda k: -sum(d.get(k, [0])))[:14]:
            a, b = pa.get(k, {}), pb.get(k, {})
            m = a.get("SQ_INSTS_MFMA", 0.0)
            wc = a.get("SQ_WAVE_CYCLES", 0.0)
            gui = b.get("GRBM_GUI_ACTIVE", 0.0)
            rec = {
                "mfma_per_launch": m / max(na.get(k, 1), 1),
                "valu_per_mfma": (a.get("SQ_INSTS_VALU", 0.0) - m) / m if m else None,
                "mfma_busy": None,
                "wait_any": a.get("SQ_WAIT_ANY", 0.0) / wc if wc else None,
                "wait_inst_any": a.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
                "active_inst_any": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
                "lds_per_mfma": b.get("SQ_INSTS_LDS", 0.0) * (na.get(k, 1) / max(nb.get(k, 1), 1)) / m if m else None,
                "lds_conflict": b.get("SQ_LDS_BANK_CONFLICT", 0.0) / b["SQ_LDS_IDX_ACTIVE"] if b.get("SQ_LDS_IDX_ACTIVE") else None,
                "insts_valu_per_launch": a.get("SQ_INSTS_VALU", 0.0) / max(na.get(k, 1), 1),
                "insts_lds_per_launch": b.get("SQ_INSTS_LDS", 0.0) / max(nb.get(k, 1), 1),
                "insts_salu_per_launch": b.get("SQ_INSTS_SALU", 0.0) / max(nb.get(k, 1), 1),
                "insts_vmem_per_launch": b.get("SQ_INSTS_VMEM", 0.0) / max(nb.get(k, 1), 1),
                "active_inst_valu_frac_of_gui": None,
            }
            # MFMA busy from pass A needs GRBM_GUI_ACTIVE of pass B: scale by launch counts
            if gui and a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                rec["mfma_busy"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(na.get(k, 1), 1) / (1024.0 * gui / max(nb.get(k, 1), 1) / 8.0)
            if gui and b.get("SQ_ACTIVE_INST_VALU"):
                rec["active_inst_valu_frac_of_gui"] = b["SQ_ACTIVE_INST_VALU"] / (1024.0 * gui / 8.0) / 4.0
            sq[k] = rec
            fmt = lambda v, p="%.2f": "-" if v is None else p % v
            lines.append("  %-46s mfma/launch %9.0f  valu/mfma %s  lds/mfma %s  mfma_busy %s  wait_any %s  wait_inst %s  lds_conflict %s"
                         % (k[:46], rec["mfma_per_launch"], fmt(rec["valu_per_mfma"]), fmt(rec["lds_per_mfma"]), fmt(rec["mfma_busy"]),
                            fmt(rec["wait_any"]), fmt(rec["wait_inst_any"]), fmt(rec["lds_conflict"])))
        facts["sq"] = sq
    facts["source"] = ("rocprofv3 --kernel-trace --stats, and separate --pmc passes (FETCH_SIZE; WRITE_SIZE; two SQ sets) over the command of "
                       "tools/archive/profile_r04.sh section '%s'; summary: profiles/r04_%s_rocprof_summary.txt" % (sub, sub))
    open(os.path.join(root, "r04_%s_rocprof_summary.txt" % sub), "w").write("\n".join(lines) + "\n")
    json.dump(facts, open(os.path.join(root, "r04_%s.json" % sub), "w"), indent=1)
    print("\n".join(lines[:12]))
