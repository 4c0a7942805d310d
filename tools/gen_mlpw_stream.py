"""Generator of instancediffusion_amd/csrc/mlpw_stream.inc: the straight-line instruction streams of mlp320w_kernel
(mlp_fused.hip, the one-wave-per-SIMD form of the fused GEGLU feed-forward).

A wave of that kernel issues everything of its chunk pipeline from ONE in-order instruction stream, so WHERE an instruction sits
between the MFMAs decides whether the matrix pipe waits for it.  Every statement of a stream is `asm volatile` (source order =
issue order: the compiler's scheduler clusters pure VALU / LDS operations in front of the MFMAs otherwise, profiles/NOTES_r06.md),
which also means the compiler inserts no `s_waitcnt lgkmcnt` for the stream's LDS reads: this script places the reads LA gaps
ahead of their consumers, keeps them in FIFO order and emits the counted waits.

One iteration j of a tile (40 chunks of 32 intermediate columns), "gap" = the issue slot behind one MFMA:
    top      s_waitcnt vmcnt(0) + s_barrier      (the LDS-DMA pieces of iteration j - 1 have landed, everybody is done with it)
    G2 part  second product of chunk j - 1 (20 MFMAs) -- the LDS-DMA pieces of iteration j ride in its first gaps
    G1 part  first product of chunk j + 1 (40 MFMAs: two 32 x 32 fragments, interleaved)
    VALU     LayerNorm fold + bias + GEGLU of chunk j (144 instructions), spread over the gaps that carry no DMA piece
Variants: 11 (steady state), 01 (j = 0: no second product yet), 10 (j = 39: no next first product), drain (j = 40: second
product only), pro (first product of chunk 0 only).

    python tools/gen_mlpw_stream.py            # rewrites the .inc (checked in; CPU test test_mlpw_stream_is_current)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "mlpw_stream.inc")

LA = int(os.environ.get("MW_LA", 3))            # gaps of lookahead of an LDS read in front of its consumer
PRE_DMA = int(os.environ.get("MW_PRE_DMA", 3))   # LDS-DMA pieces between the top barrier's first reads and the first MFMA (they cover the read latency)
MAXV = int(os.environ.get("MW_MAXV", 4))         # at most this many VALU statements per gap; the rest trails behind the last MFMA
# timing experiments only (wrong results): parts of the stream left out
NO_VALU = os.environ.get("MW_NO_VALU") == "1"
NO_DMA = os.environ.get("MW_NO_DMA") == "1"
NO_MFMA = os.environ.get("MW_NO_MFMA") == "1"
STAGGER = os.environ.get("MW_STAGGER", "0") == "1"   # the LDS-DMA piece of gap g is issued by wave g % 4 only (one wave at a time on the CU's address path)
TRACE = os.environ.get("MW_TRACE") == "1"        # s_memtime marks at the segment borders of mw_body_11 (a -DIDF_MLPW_TRACE build reads them)


class Read:
    def __init__(self, name, expr, need):
        self.name, self.expr, self.need = name, expr, need
        self.seq = None


def valu_half(h):
    """fold + bias + GEGLU of fragment h of the chunk being activated, stage by stage (4 independent pairs per stage).
    Returns [(statement, [const reads it needs])]; a statement that is not an instruction has an empty tag 'glue'."""
    A = f"accC[{h}]"
    out = []
    cn = {k: [f"c{h}_{k}{q}" for q in range(2)] for k in ("cv", "cg", "dv", "dg")}

    def cpair(k, p):
        return f"mw_half<{p & 1}>({cn[k][p >> 1]})", cn[k][p >> 1]

    for p in range(4):
        e, r = cpair("cv", p)
        out.append((f"const f32x2 tv{h}{p} = mw_pk_fma(c.nmu2, {e}, mw_pair<{2 * p}>({A}));", [r]))
    for p in range(4):
        e, r = cpair("cg", p)
        out.append((f"const f32x2 tg{h}{p} = mw_pk_fma(c.nmu2, {e}, mw_pair<{8 + 2 * p}>({A}));", [r]))
    for p in range(4):
        e, r = cpair("dv", p)
        out.append((f"const f32x2 va{h}{p} = mw_pk_fma(c.rstd2, tv{h}{p}, {e});", [r]))
    for p in range(4):
        e, r = cpair("dg", p)
        out.append((f"const f32x2 ga{h}{p} = mw_pk_fma(c.rstd2, tg{h}{p}, {e});", [r]))
    for p in range(4):
        out.append((f"const float xa{h}{p} = mw_med3(ga{h}{p}.x, c.lo8, c.hi8);", []))
        out.append((f"const float xb{h}{p} = mw_med3(ga{h}{p}.y, c.lo8, c.hi8);", []))
    for p in range(4):
        out.append((f"const f32x2 xc{h}{p} = {{xa{h}{p}, xb{h}{p}}}; const f32x2 x2{h}{p} = mw_pk_mul(xc{h}{p}, xc{h}{p});", []))
    for p in range(4):
        out.append((f"const f32x2 qa{h}{p} = mw_pk_fma(x2{h}{p}, c.k1, c.k2);", []))
    for p in range(4):
        out.append((f"const f32x2 qb{h}{p} = mw_pk_fma(qa{h}{p}, x2{h}{p}, c.k3);", []))
    for p in range(4):
        out.append((f"const f32x2 tt{h}{p} = mw_pk_mul(xc{h}{p}, qb{h}{p});", []))
    for p in range(4):
        out.append((f"const float ea{h}{p} = mw_exp2(tt{h}{p}.x);", []))
        out.append((f"const float eb{h}{p} = mw_exp2(tt{h}{p}.y);", []))
    for p in range(4):
        out.append((f"const f32x2 ee{h}{p} = {{ea{h}{p}, eb{h}{p}}}; const f32x2 ss{h}{p} = mw_pk_add(ee{h}{p}, c.one2);", []))
    for p in range(4):
        out.append((f"const float ra{h}{p} = mw_rcp(ss{h}{p}.x);", []))
        out.append((f"const float rb{h}{p} = mw_rcp(ss{h}{p}.y);", []))
    for p in range(4):
        out.append((f"const f32x2 rr{h}{p} = {{ra{h}{p}, rb{h}{p}}}; const f32x2 gl{h}{p} = mw_pk_mul(ga{h}{p}, rr{h}{p});", []))
    for p in range(4):
        out.append((f"const f32x2 oo{h}{p} = mw_pk_mul(va{h}{p}, gl{h}{p});", []))
    for p in range(4):
        out.append((f"hC[{h}][{p}] = mw_cvt_pk<DT>(oo{h}{p}.x, oo{h}{p}.y);", []))
    return out


def const_reads(h, need):
    """the 8 fold-constant reads of fragment h: c[64] | d[64] fp32 per chunk; the lane's rows are 8 q + 4 hi + e (values)
    and 16 + 8 q + 4 hi + e (gates) of fragment h (cda already holds + 16 hi bytes)"""
    idx = {"cv": 0, "cg": 16, "dv": 64, "dg": 80}
    rs = []
    for k in ("cv", "cg", "dv", "dg"):
        for q in range(2):
            off = (32 * h + idx[k] + 8 * q) * 4
            rs.append(Read(f"c{h}_{k}{q}", f"mw_lds128f<{off}>(c.cda)", need))
    return rs


DMA_IMM = os.environ.get("MW_DMA_IMM", "1") == "1"    # pieces as one asm statement with immediate offsets (mw_dma)


def dma_pieces():
    ps = []
    if DMA_IMM:
        for kt in range(5):
            for u in range(2):
                ps.append(f"mw_dma<{kt * 8192 + u * 4096}, {kt * 128}>(c.w1dst, c.w1_vj, c.w1b[{u}]);")
        for t in range(5):
            ps.append(f"mw_dma<{t * 4096}, 0>(c.w2dst, c.w2_vj, c.w2b[{t}]);")
        ps.append("if (c.wave == 3) mw_dma<0, 0>(c.cddst, c.cd_vj, c.cdb);")
        return ps
    for kt in range(5):
        for u in range(2):
            ps.append(f"mlp_dma16(c.w1src + {u} * c.w1_ustride + {kt * 128}, c.w1_voff, c.w1dst + {kt * 8192 + u * 4096});")
    for t in range(5):
        ps.append(f"mlp_dma16(c.w2src + {t} * c.w2_tstride, c.w2_voff, c.w2dst + {t * 4096});")
    ps.append("if (c.wave == 3) mlp_dma16(c.cdsrc, c.cd_voff, c.cddst);")
    return ps


def g2_mfmas():
    """second product: accumulators 0..4 take k-step 0 first, 5..9 k-step 1 first (the order of the 8-wave kernel, whose
    wn = 1 waves own the upper 160 output columns and add their own fragment first: bit-identical sums)"""
    order = [(a, 0) for a in range(5)] + [(a, 1) for a in range(5, 10)] + [(a, 1) for a in range(5)] + [(a, 0) for a in range(5, 10)]
    ms = []
    for i, (a, kk) in enumerate(order):
        rd = Read(f"w2_{a}_{kk}", f"mw_lds128<{a * 2048}>(c.w2a[{kk}])", None)
        ms.append((f"mw_mf2<DT, {a}>({rd.name}, hP[{kk}]);", rd))
    return ms


def g1_mfmas():
    ms = []
    for i in range(40):
        ks, f = i >> 1, i & 1
        rd = Read(f"w1_{ks}_{f}", f"mw_lds128<{(ks >> 2) * 8192 + f * 4096}>(c.w1a[{ks & 3}])", None)
        first = "true" if ks == 0 else "false"
        ms.append((f"mw_mf1<DT, {ks}, {first}>(accN[{f}], {rd.name});", rd))
    return ms


def build(name, g2, g1, valu, dma, top=True):
    mf = (g2_mfmas() if g2 else []) + (g1_mfmas() if g1 else [])
    ngap = len(mf)
    pieces = dma_pieces() if (dma and not NO_DMA) else []
    pre = pieces[:PRE_DMA]
    rest = pieces[PRE_DMA:]
    dma_gaps = len(rest)                       # one per gap from gap 0
    if STAGGER:
        dma_gaps = 0                           # every gap may carry a piece (of one wave): the VALU statements go everywhere
    vl = (valu_half(0) + valu_half(1)) if valu else []
    if NO_VALU and valu:
        vl = [(f"hC[{h}] = u32x4{{0u, 0u, 0u, 0u}};", []) for h in range(2)]
        valu = False
    # VALU statements over the gaps without a DMA piece; what does not fit under MFMAs trails behind the last one
    vgaps = list(range(min(dma_gaps, ngap), ngap))
    per_gap = {g: [] for g in range(ngap + 1)}
    if vl:
        if vgaps:
            n_under = min(len(vl), MAXV * len(vgaps))
            for k in range(n_under):
                per_gap[vgaps[k * len(vgaps) // n_under]].append(vl[k])
            for k in range(n_under, len(vl)):
                per_gap[ngap].append(vl[k])
        else:
            per_gap[ngap] = list(vl)
    # need positions of the reads: an MFMA's fragment at its gap; a constant read at the gap of the first VALU that uses it
    reads = []
    for g, (_, rd) in enumerate(mf):
        rd.need = g
        reads.append(rd)
    cr = {}
    if valu:
        first_use = {}
        for g in range(ngap + 1):
            for st, needs in per_gap[g]:
                for r in needs:
                    first_use.setdefault(r, g)
        for h in range(2):
            for rd in const_reads(h, None):
                rd.need = first_use[rd.name]
                cr[rd.name] = rd
                reads.append(rd)
    reads.sort(key=lambda r: (r.need, 0 if r.name.startswith("c") else 1))
    for i, r in enumerate(reads):
        r.seq = i
    byname = {r.name: r for r in reads}

    lines = []
    state = dict(issued=0, done=0)

    def issue_upto(gap):
        # FIFO; a constant read goes out at most 2 gaps ahead (8 of them at once: the lgkmcnt field counts to 15)
        while state["issued"] < len(reads):
            r = reads[state["issued"]]
            ahead = min(LA, 2) if r.name.startswith("c") else LA
            if r.need - ahead > gap - LA or state["issued"] - state["done"] >= 14:
                break
            ty = "f32x4" if r.name.startswith("c") else "u32x4"
            lines.append(f"  const {ty} {r.name} = {r.expr};")
            state["issued"] += 1

    def wait_for(rname):
        s = byname[rname].seq
        if s < state["done"]:
            return
        assert s < state["issued"], (name, rname)
        n = state["issued"] - s - 1
        assert 0 <= n <= 15, (name, rname, n)
        lines.append(f"  mw_wait_lgkm<{n}>();")
        state["done"] = s + 1

    args = "f32x16 (&accC)[2], f32x16 (&accN)[2], const u32x4 (&hP)[2], u32x4 (&hC)[2], const MwCtx& c"
    lines.append(f"template <int DT> __device__ __forceinline__ void {name}({args}) {{")
    if TRACE and name == "mw_body_11":
        lines.append("  MW_TR_DECL")
    tr = TRACE and name == "mw_body_11"
    marks = {len(rest): 3, 20: 4, 40: 5}           # mark i in front of the MFMA of gap g
    if tr:
        lines.append("  MW_TR_MARK(0)")
    if top:
        lines.append("  MW_TOP")
    if tr:
        lines.append("  MW_TR_MARK(1)")
    issue_upto(LA)
    for s in pre:
        lines.append("  " + s)
    if tr:
        lines.append("  MW_TR_MARK(2)")
    for g in range(ngap):
        st, rd = mf[g]
        if tr and g in marks:
            lines.append(f"  MW_TR_MARK({marks[g]})")
        wait_for(rd.name)
        if not NO_MFMA:
            lines.append("  " + st)
        issue_upto(g + 1 + LA)
        if STAGGER:
            if (g >> 2) < len(rest):
                pc = rest[g >> 2]
                if not pc.startswith("if (c.wave == 3)"):
                    lines.append(f"  if (c.wave == {g & 3}) {{ {pc} }}")
                elif (g & 3) == 3:
                    lines.append("  " + pc)               # the constants' piece belongs to wave 3 alone
        elif g < len(rest):
            lines.append("  " + rest[g])
        for vs, needs in per_gap[g]:
            for r in needs:
                wait_for(r)
            lines.append("  " + vs)
    for s in (rest[(ngap + 3) >> 2:] if STAGGER else rest[ngap:]):
        lines.append("  " + s)
    issue_upto(10 ** 9)
    for vs, needs in per_gap[ngap]:
        for r in needs:
            wait_for(r)
        lines.append("  " + vs)
    assert state["issued"] == len(reads)
    if tr:
        lines.append("  MW_TR_MARK(6)")
        lines.append("  MW_TR_END")
    lines.append("}")
    return "\n".join(lines)


def main():
    parts = ["// GENERATED by tools/gen_mlpw_stream.py -- do not edit; see that script for the schedule rules.",
             f"// LA = {LA} gaps of LDS-read lookahead, {PRE_DMA} LDS-DMA pieces in front of the first MFMA.", ""]
    parts.append(build("mw_body_11", True, True, True, True))
    parts.append(build("mw_body_01", False, True, True, True))
    parts.append(build("mw_body_10", True, False, True, True))
    parts.append(build("mw_drain", True, False, False, False))
    parts.append(build("mw_pro", False, True, False, False, top=False))
    txt = "\n\n".join(parts) + "\n"
    if "-o" in sys.argv:
        open(sys.argv[sys.argv.index("-o") + 1], "w").write(txt)
        return
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        sys.exit(0 if cur == txt else 1)
    open(OUT, "w").write(txt)
    print("wrote", OUT, len(txt.split("\n")), "lines")


if __name__ == "__main__":
    main()
