"""Where does the fp16 leg lose against bf16?  Joins two `rocprofv3 --kernel-trace --stats` kernel_stats.csv files of the SAME
command (tools/profile_forward.py 128 3, PROFILE_DTYPE=bf16 / fp16, one box, back to back) by kernel FAMILY -- the fp16
instantiation of a template is matched to the bf16 one by replacing its dtype template argument -- and prints total time per
family and the difference.     python tools/dtype_kernel_diff.py bf16_stats.csv fp16_stats.csv [forwards=3] > table.txt"""
import csv
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)                       # drop the parameter list
    n = re.sub(r"<([01]),", "<DT,", n)              # leading dtype template argument (IDF_BF16 = 0, IDF_F16 = 1)
    n = re.sub(r"<([01])>", "<DT>", n)
    return n


def load(path):
    tot = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        f = family(r["Name"])
        tot[f][0] += float(r["TotalDurationNs"]) / 1e6
        tot[f][1] += int(r["Calls"])
    return tot


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    fw = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    keys = sorted(set(a) | set(b), key=lambda k: -(a.get(k, [0, 0])[0] + b.get(k, [0, 0])[0]))
    ta = sum(v[0] for v in a.values()) / fw
    tb = sum(v[0] for v in b.values()) / fw
    print(f"{'kernel family':<78} {'bf16 ms':>9} {'fp16 ms':>9} {'diff ms':>8} {'diff %':>7}   (per forward; {fw} forwards each)")
    for k in keys:
        x, y = a.get(k, [0.0, 0])[0] / fw, b.get(k, [0.0, 0])[0] / fw
        if max(x, y) < 0.05:
            continue
        print(f"{k[:78]:<78} {x:9.3f} {y:9.3f} {y - x:8.3f} {100 * (y - x) / max(x, 1e-9):7.1f}")
    print(f"{'TOTAL kernel time':<78} {ta:9.3f} {tb:9.3f} {tb - ta:8.3f} {100 * (tb - ta) / ta:7.1f}")


if __name__ == "__main__":
    main()
