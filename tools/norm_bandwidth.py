"""GB/s of the memory-bound kernels (GroupNorm stats / apply, ScaleU stats / apply, row statistics, statistics finalize,
LayerNorm, conv_in) from (a) the per-kernel HBM bytes of the two PMC passes (tools/pmc_summary.py output) and (b) the
rocprofv3 --kernel-trace --stats average durations of eager forwards at the same batch.

    python tools/norm_bandwidth.py <pmc_traffic_bNN.json> <forward_kernel_stats.csv> <n_forwards_in_stats_run> <out_json>

Achieved = measured HBM bytes (FETCH x2 correction + WRITE) / average duration; `frac_of_8TBps` against the 8 TB/s spec
(6.29 TB/s is what a float4 copy reaches, MI355X_MICROARCH.md).  The ALGORITHMIC figures for GroupNorm as a whole
(2 B in + 2 B out per element over stats + apply time) are in the `groupnorm_algorithmic` row."""
import csv
import json
import sys

pmc, stats_csv, n_fwd, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
k = json.load(open(pmc))["kernels"]
st = {r["Name"]: r for r in csv.DictReader(open(stats_csv))}
rows = []
want = ("gn_stats", "gn_apply", "scaleu_stats", "scaleu_apply", "row_stats", "stats_finalize", "ln_kernel", "conv_in")
for name, v in k.items():
    if not any(w in name for w in want):
        continue
    s = st.get(name)
    if s is None:
        continue
    avg_us = float(s["AverageNs"]) / 1e3
    short = name.replace("void ", "").replace("(anonymous namespace)::", "")
    short = short[:short.index("(")] if "(" in short else short
    rows.append(dict(kernel=short[:60],
                     launches_per_forward=round(int(s["Calls"]) / n_fwd, 1), avg_us=round(avg_us, 1),
                     fetch_MB=round(v["fetch_bytes_corrected_x2_avg"] / 1e6, 1), write_MB=round(v["write_bytes_avg"] / 1e6, 1),
                     hbm_MB_per_launch=round(v["hbm_bytes_per_launch"] / 1e6, 1),
                     achieved_GBps=int(v["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9),
                     frac_of_8TBps=round(v["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 8e12, 3)))
gs = [r for r in rows if "gn_stats" in r["kernel"]]
ga = [r for r in rows if "gn_apply" in r["kernel"]]
if gs and ga:
    # algorithmic bytes of GroupNorm = what apply writes, twice (2 B read + 2 B written per element), per GroupNorm; time =
    # every apply launch + every statistics launch of the forward (round 5: fewer statistics than apply launches -- the conv
    # epilogue leaves the partials of 32 of the 61 GroupNorms; that epilogue's cost sits in the conv family, not here)
    alg = 2 * ga[0]["write_MB"]
    n_apply, n_stats = ga[0]["launches_per_forward"], gs[0]["launches_per_forward"]
    t_fwd_us = ga[0]["avg_us"] * n_apply + gs[0]["avg_us"] * n_stats
    meas = ga[0]["hbm_MB_per_launch"] * n_apply + gs[0]["hbm_MB_per_launch"] * n_stats
    rows.append(dict(kernel="groupnorm_algorithmic (stats + apply)", launches_per_forward=n_apply, stats_launches_per_forward=n_stats,
                     avg_us=round(t_fwd_us / n_apply, 1), algorithmic_MB=round(alg, 1), measured_MB=round(meas / n_apply, 1),
                     achieved_GBps_on_algorithmic_bytes=int(alg * n_apply * 1e6 / (t_fwd_us * 1e-6) / 1e9),
                     frac_of_8TBps=round(alg * n_apply * 1e6 / (t_fwd_us * 1e-6) / 8e12, 3),
                     ms_per_forward=round(t_fwd_us / 1e3, 2)))
json.dump(dict(note=__doc__.strip().splitlines()[0], pmc=pmc, stats=stats_csv, rows=rows), open(out, "w"), indent=1)
for r in rows:
    print(r)
