"""Where is the spill code?  Prints, per kernel of a `hipcc -S --cuda-device-only` listing, the run-length sequence of
MFMA (M), scratch store / load (S / L), barrier (B), global store / load (G / g), LDS-DMA (D) instructions in program order:
a K loop shows as `B1 ... M40`; S / L between the Ms of that run mean scratch traffic in the hot loop.
Usage: python tools/isa_events.py listing.s [name-filter]"""
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r"\n\t\.type\t(_Z\S+),@function\n", txt)
for name, body in zip(parts[1::2], parts[2::2]):
    if flt not in name:
        continue
    body = body.split("\n\t.section")[0] if "s_endpgm" not in body else body[:body.rindex("s_endpgm")]
    ev = []
    for l in body.split("\n"):
        l = l.strip()
        if l.startswith("v_mfma"): ev.append("M")
        elif l.startswith("scratch_store"): ev.append("S")
        elif l.startswith("scratch_load"): ev.append("L")
        elif l.startswith("s_barrier"): ev.append("B")
        elif l.startswith("global_store") or l.startswith("buffer_store"): ev.append("G")
        elif l.startswith("global_load_lds"): ev.append("D")
        elif l.startswith("global_load") or l.startswith("buffer_load"): ev.append("g")
    out, last, cnt = [], None, 0
    for e in ev:
        if e == last:
            cnt += 1
        else:
            if last:
                out.append(f"{last}{cnt}")
            last, cnt = e, 1
    if last:
        out.append(f"{last}{cnt}")
    print(re.sub(r"_ZN\d+_GLOBAL__N_1\d+", "", name)[:80])
    print("   ", " ".join(out))
