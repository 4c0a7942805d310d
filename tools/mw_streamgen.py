"""The scheduler shared by the generators of the row kernels' instruction streams (gen_qkvw_stream.py, gen_qkv640w_stream.py,
gen_gegluw_stream.py -> instancediffusion_amd/csrc/*_stream.inc; gen_mlpw_stream.py, the first of the family, keeps its own).

A stream is ONE wave's program for one pipeline step: every statement becomes an `asm volatile` primitive of csrc/mw_prims.h, so
the order written here is the order issued.  The rules:
  * the step's MFMAs are the backbone; "gap g" is the room behind MFMA g (a 32 x 32 x 16 MFMA occupies the matrix pipe for 32
    cycles: ~4-6 other instructions issue under it for free)
  * LDS operations -- reads AND writes -- retire in order through one counter: `Stream` numbers them, and a consumer waits with
    the exact s_waitcnt lgkmcnt(N) that lets every younger operation stay in flight
  * a W fragment read is issued LA gaps ahead of its MFMA, a constant read of the epilogue min(LA, 2) gaps ahead of the gap of
    its first use, never more than 13 outstanding (the counter has 4 bits)
  * the LDS-DMA pieces of the next W chunk go in front of the first MFMA (PRE_DMA of them) and one per gap behind it; the
    epilogue of the previous work item is spread over the gaps behind the pieces, at most MAXV statements per gap, the rest
    trails the last MFMA
An epilogue is a list of (kind, code, needs, defines): kind 'r' a hoistable LDS read, 'l' an LDS operation that stays in place
(a write, or the read-back of a staging image), 's' any other statement; `needs` names the reads it consumes.
"""
import os
import sys


class Stream:
    """statements in issue order; LDS operations (reads AND writes) are counted: a wait for read r is lgkmcnt(issued - seq(r) - 1)"""

    def __init__(self):
        self.lines, self.issued, self.done, self.seq = [], 0, 0, {}

    def lds(self, code, name=None):
        self.lines.append("  " + code)
        if name:
            self.seq[name] = self.issued
        self.issued += 1

    def wait(self, name):
        s = self.seq[name]
        if s < self.done:
            return
        n = self.issued - s - 1
        assert 0 <= n <= 15, (name, n)
        self.lines.append(f"  mw_wait_lgkm<{n}>();")
        self.done = s + 1

    def stmt(self, code, needs=()):
        for r in needs:
            self.wait(r)
        self.lines.append("  " + code)


ARGS = "f32x16 (&accC)[2], f32x16 (&accN)[2], const {ctx}& c"     # accC: the item in its epilogue, accN: the one in its MFMAs


def schedule(decl, mfs, pieces, items, la, pre_dma, maxv, top=None, xload=None):
    """One stream as the text of a __device__ function.
    decl: the function's declaration; mfs: [(MFMA statement, read name, read statement)]; pieces: the LDS-DMA statements;
    items: the epilogue (module docstring); top: the statement that opens the step (wait + barrier) or None; xload: the next
    tile's row fetch (last step of a tile) or None."""
    st = Stream()
    st.lines.append(decl + " {")
    if top:
        st.lines.append("  " + top)
    ngap = len(mfs)
    pre, rest = pieces[:pre_dma], pieces[pre_dma:]
    # epilogue statements over the gaps behind the DMA pieces
    vgaps = list(range(min(len(rest), ngap), ngap))
    per_gap = {g: [] for g in range(ngap + 1)}
    n_under = min(len(items), maxv * len(vgaps))
    for k in range(n_under):
        per_gap[vgaps[k * len(vgaps) // n_under]].append(items[k])
    for k in range(n_under, len(items)):
        per_gap[ngap].append(items[k])
    hoist = []                              # (need gap, order, code, name, gaps ahead)
    for g, (_, rn, rc) in enumerate(mfs):
        hoist.append((g, 1, rc, rn, la))
    for g in range(ngap + 1):
        for kind, code, needs, defs in per_gap[g]:
            if kind == "r":
                hoist.append((g, 0, code, defs, min(la, 2)))
    hoist.sort(key=lambda h: (h[0], h[1]))
    hp = [0]

    def issue_upto(gap):
        while hp[0] < len(hoist):
            need, _, code, rn, ahead = hoist[hp[0]]
            if need - ahead > gap or st.issued - st.done >= 13:
                break
            st.lds(code, rn)
            hp[0] += 1

    def force(rn):                          # a consumer is about to wait for rn: it must have been issued
        while rn not in st.seq:
            _, _, code, r2, _ = hoist[hp[0]]
            st.lds(code, r2)
            hp[0] += 1

    def emit(kind, ecode, needs, defs):
        if kind == "r":
            return
        for r in needs:
            force(r)
        if kind == "l":
            for r in needs:
                st.wait(r)
            st.lds(ecode, defs)
        else:
            st.stmt(ecode, needs)

    if xload:
        st.lines.append("  " + xload)
    issue_upto(0)
    for s in pre:
        st.lines.append("  " + s)
    for g in range(ngap):
        code, rn, _ = mfs[g]
        issue_upto(g)
        force(rn)
        st.stmt(code, [rn])
        issue_upto(g + 1)
        if g < len(rest):
            st.lines.append("  " + rest[g])
        for it in per_gap[g]:
            emit(*it)
    for s in rest[ngap:]:
        st.lines.append("  " + s)
    for it in per_gap[ngap]:
        emit(*it)
    assert hp[0] == len(hoist), (decl, hp[0], len(hoist))
    st.lines.append("}")
    return "\n".join(st.lines)


def header(script, la, pre_dma, maxv):
    return [f"// GENERATED by tools/{script} -- do not edit; see that script for the schedule rules.",
            f"// LA = {la} gaps of LDS-read lookahead, {pre_dma} LDS-DMA pieces in front of the first MFMA, <= {maxv} epilogue statements per gap.", ""]


def finish(parts, out):
    """write the .inc; `-o FILE` writes a variant elsewhere, `--check` exits 1 when the checked-in file is stale"""
    txt = "\n\n".join(parts) + "\n"
    if "-o" in sys.argv:
        open(sys.argv[sys.argv.index("-o") + 1], "w").write(txt)
        return
    if "--check" in sys.argv:
        cur = open(out).read() if os.path.exists(out) else ""
        sys.exit(0 if cur == txt else 1)
    open(out, "w").write(txt)
    print("wrote", out, len(txt.split("\n")), "lines")
