"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- collected in SEPARATE runs, MI355X_MICROARCH.md §HBM) over
tools/profile_forward.py into (a) a per-kernel table and (b) the per-op-family table bench.py reads for
`roofline.traffic` (profiles/pmc_traffic.json, keyed by forward batch).

    python tools/pmc_summary.py <fetch_dir> <write_dir> <batch> <out_json> [<family_table_json>]

FETCH_SIZE is reported in KB and counts 16-B/lane streams at 1/2 on gfx950 -> doubled here (the correction the guide
prescribes; cross-checked on ln_kernel, whose read and write volumes are equal).  WRITE_SIZE is in KB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return tot, n


def _targs(name):
    """Template arguments of a demangled kernel name, e.g. 'gemm_kernel_big<0, 256, 320, 64, 2, true, false>(...)'."""
    if "<" not in name:
        return []
    inner = name[name.index("<") + 1:name.rindex(">")] if ">" in name else ""
    return [a.strip() for a in inner.split(",")]


def family(name):
    if "attn" in name:
        return "attention"
    if "splitk_reduce" in name:
        return "conv3x3"                         # only the 8x8-level convs / ff-out GEMMs split K; booked with the convs
    if any(k in name for k in ("mlp320", "qkv320w", "qkv640w", "geglu640w")):    # the fused GEGLU feed-forward and the row-resident projection kernels
        return "gemm"
    if "gemm_kernel_big" in name:                # <DT, BM, BN, BKT, NSTG, CONV, SPLIT>
        a = _targs(name)
        return "conv3x3" if len(a) > 5 and a[5] in ("true", "1") else "gemm"
    if "gemm_kernel_pp" in name:                 # <DT, BN, CONV, DL>
        a = _targs(name)
        return "conv3x3" if len(a) > 2 and a[2] in ("true", "1") else "gemm"
    if "gemm_kernel" in name:                    # <DT, BM, BN, WM, WN, CONV>
        a = _targs(name)
        return "conv3x3" if a and a[-1] in ("true", "1") else "gemm"
    if "row_stats" in name:
        return "gemm"                            # the LayerNorm statistics pass belongs to the GEMM that consumes it
    if "gn_stats" in name or "gn_apply" in name:
        return "groupnorm"
    if "scaleu" in name:
        return "scaleu_concat"
    return None


def main():
    fetch_dir, write_dir, batch, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    ft, fn = load(fetch_dir, "FETCH_SIZE")
    wt, wn = load(write_dir, "WRITE_SIZE")
    kernels = {}
    fam = defaultdict(lambda: dict(bytes=0.0, launches=0))
    for k in sorted(set(ft) | set(wt)):
        n = max(fn.get(k, 0), wn.get(k, 0))
        fetch_b = 2.0 * 1024.0 * ft.get(k, 0.0)
        write_b = 1024.0 * wt.get(k, 0.0)
        kernels[k] = dict(launches=n, fetch_size_kb_raw_avg=round(ft.get(k, 0.0) / max(fn.get(k, 1), 1), 1),
                          fetch_bytes_corrected_x2_avg=int(fetch_b / max(n, 1)), write_bytes_avg=int(write_b / max(n, 1)),
                          hbm_bytes_per_launch=int((fetch_b + write_b) / max(n, 1)))
        f = family(k)
        if f:
            fam[f]["bytes"] += fetch_b + write_b
            fam[f]["launches"] += n
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over eager full-model UNet forwards at batch "
            f"{batch} (tools/profile_forward.py). FETCH_SIZE doubled (gfx950 1/2 correction). Counter unit KB.")
    json.dump(dict(note=note, batch=batch, kernels=kernels), open(out_json, "w"), indent=1)
    if len(sys.argv) > 5:
        path = sys.argv[5]
        tab = json.load(open(path)) if os.path.exists(path) else {}
        tab[str(batch)] = {f: dict(hbm_bytes_per_launch=int(v["bytes"] / max(v["launches"], 1)), launches_in_pass=v["launches"],
                                   source=os.path.basename(out_json)) for f, v in fam.items()}
        json.dump(tab, open(path, "w"), indent=1)
    for f, v in fam.items():
        print(f, v["launches"], "launches", round(v["bytes"] / max(v["launches"], 1) / 1e6, 1), "MB/launch")


if __name__ == "__main__":
    main()
