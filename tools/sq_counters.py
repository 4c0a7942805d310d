"""Fold rocprofv3 PMC passes (one directory per pass; SQ has 8 slots per pass, MI355X_MICROARCH.md §rocprofv3 PMC slots)
into one per-kernel table: average counter value per dispatch, plus derived ratios for the attention claims
(MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES-equivalent), wait fractions of SQ_WAVE_CYCLES).

    python tools/sq_counters.py <out_csv> <pass_dir> [<pass_dir> ...] [--match attn]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    out_csv, dirs = args[0], args[1:]
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if match and match not in k:
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
                if "Start_Timestamp" in r and r.get("End_Timestamp"):
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    counters = sorted({c for k in tot for c in tot[k]})
    rows = []
    for k in sorted(tot):
        avg = {c: tot[k][c] / max(cnt[k][c], 1) for c in tot[k]}
        row = dict(kernel=k[:120], dispatches=max(cnt[k].values()), avg_ns_under_profiler=int(sum(dur[k]) / max(len(dur[k]), 1)))
        row.update({c: round(avg.get(c, float("nan")), 1) for c in counters})
        wc = avg.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if c in avg:
                    row["frac_" + c[3:].lower() + "_of_wave_cycles"] = round(avg[c] / wc, 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg and avg["SQ_BUSY_CYCLES"]:
            row["mfma_busy_over_sq_busy"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / avg["SQ_BUSY_CYCLES"], 4)
        rows.append(row)
    keys = []
    for r in rows:
        for c in r:
            if c not in keys:
                keys.append(c)
    with open(out_csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=keys)
        w.writeheader()
        for r in rows:
            w.writerow(r)
    for r in rows:
        print(r)


if __name__ == "__main__":
    main()
