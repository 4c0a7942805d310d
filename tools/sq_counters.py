"""Fold rocprofv3 PMC passes (one directory per pass; SQ has 8 slots per pass, MI355X_MICROARCH.md §rocprofv3 PMC slots)
into one per-kernel table: average counter value per dispatch, plus derived ratios for the attention claims
(MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES-equivalent), wait fractions of SQ_WAVE_CYCLES).

    python tools/sq_counters.py <out_csv> <pass_dir> [<pass_dir> ...] [--match attn] [--useful 0.714]

Normalised matrix-pipe numbers (the old `mfma_busy_over_sq_busy` was a ratio of two unnormalised sums):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)   -- SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed
      over the chip's SIMDs (32 per 32x32x16 bf16 MFMA, MI355X_MICROARCH.md); kernel cycles = GRBM_GUI_ACTIVE of the dispatch
      (summed over the 8 XCDs by rocprofv3, hence / 8; `implied_clock_GHz` = cycles / duration is printed as a sanity check)
      when that counter was collected in a pass, else duration_ns x 2.1 GHz (flagged in `cycles_source`);
  useful_frac = mfma_busy_frac x --useful (the share of the issued MFMA work that is algorithmic: d = 40 attention pads
      Q.K^T 40 -> 48 and P.V 40 -> 64, 640 useful of 896 issued cycles = 0.714): the fraction of the matrix pipe's capacity AT
      THE CLOCK THE KERNEL RAN AT that did algorithmic work; x implied_clock / 2.4 GHz gives the fraction of the 2.5 PFLOP/s
      spec peak that the timing shows.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    useful = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    if "--useful" in sys.argv:
        useful = float(sys.argv[sys.argv.index("--useful") + 1])
        args = [a for a in args if a != sys.argv[sys.argv.index("--useful") + 1]]
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
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            n_simd = 1024                                    # 256 CUs x 4 SIMDs
            if avg.get("GRBM_GUI_ACTIVE"):
                # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (taken as one chip-wide count it would imply a
                # 13.6 GHz clock on a 1.93 ms kernel); per-XCD average = chip cycles of the dispatch
                cycles, row["cycles_source"] = avg["GRBM_GUI_ACTIVE"] / 8.0, "GRBM_GUI_ACTIVE / 8 XCDs"
                row["implied_clock_GHz"] = round(cycles / max(row["avg_ns_under_profiler"], 1), 3)
            else:
                cycles, row["cycles_source"] = row["avg_ns_under_profiler"] * 2.1, "duration_ns x 2.1 GHz (assumed clock)"
            row["mfma_busy_frac"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (n_simd * cycles), 4)
            if useful is not None:
                row["useful_frac"] = round(row["mfma_busy_frac"] * useful, 4)
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
