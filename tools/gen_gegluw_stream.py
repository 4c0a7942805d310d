"""Generator of instancediffusion_amd/csrc/gegluw_stream.inc: the straight-line instruction streams of geglu640w_kernel
(geglu_fused.hip: the GEGLU projection of a C = 640 transformer block -- value * gelu(gate) of LN(x) . W1^T + b1 -- with the
activation rows resident in registers).

Same rules as tools/gen_mlpw_stream.py / gen_qkvw_stream.py.  One pipeline step i of a 128-row tile (160 chunks of 32 packed W1
rows = 16 output columns each):
    top      s_waitcnt vmcnt(VMC) + s_barrier: the ten LDS-DMA pieces of step i - 1 (W chunk i + 1) landed; a store group issued
             behind them (VMC = 4 after a store step) may still be in flight
    MFMA     first product of chunk i + 1: ONE 32 x 32 fragment, K = 640 = 40 k-steps, split over two accumulators (even / odd
             k-steps) so that consecutive MFMAs are independent; the ten pieces of chunk i + 2 ride in front of / in its first gaps
    epilogue of chunk i: sum of the two accumulators, LayerNorm fold + bias, GEGLU (the fused MLP's x sigmoid(p(x)) form), 16-bit,
             8 B per lane into the wave's staging image [32 rows][128 B]; every fourth chunk (ST) the image is read back and
             stored as whole 128-B lines (4 stores)
Variants: gw_pro (MFMA of chunk 0 only), gw_step<ST> (steady state), gw_last (epilogue of chunk 159 only; it also fetches the next
tile's rows BEFORE its stores).

    python tools/gen_gegluw_stream.py            # rewrites the .inc (checked in; CPU test in tests/test_capi.py)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "gegluw_stream.inc")
LA = int(os.environ.get("GW_LA", 3))
PRE_DMA = int(os.environ.get("GW_PRE_DMA", 3))
MAXV = int(os.environ.get("GW_MAXV", 6))
NO_EPI = os.environ.get("GW_NO_EPI") == "1"       # timing experiments (wrong results)
NO_DMA = os.environ.get("GW_NO_DMA") == "1"


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


def mf_items():
    out = []
    for ks in range(40):
        name = f"w_{ks}"
        first = "true" if ks < 2 else "false"
        out.append((f"mw_mf1<DT, {ks}, {first}>(accN[{ks & 1}], {name});", name,
                    f"const u32x4 {name} = mw_lds128<{(ks >> 2) * 4096}>(c.w1a[{ks & 3}]);"))
    return out


def epilogue(store):
    """[(kind, code, needs, defines)]: 'r' hoistable constant read, 'l' in-place LDS operation, 's' plain statement.
    acc[4 q + e] of the summed fragment = packed W1 row 8 q + 4 hi + e: q = 0, 1 values, q = 2, 3 gates of output columns
    {4 hi + e, 8 + 4 hi + e} of the chunk's 16 (mlp_fused.hip)."""
    it = []
    for r in range(16):
        it.append(("s", f"const float a{r} = mw_add(accC[0][{r}], accC[1][{r}]);", [], None))
    cn = {}
    for k, base in (("cv", 0), ("cg", 16), ("dv", 5120), ("dg", 5120 + 16)):
        for q in range(2):
            cn[k, q] = f"{k}{q}"
    # stage-ordered over the 8 outputs (q, e)
    outs = [(q, e) for q in range(2) for e in range(4)]
    for q in range(2):
        it.append(("r", f"const f32x4 cv{q} = mw_lds128f<{(8 * q) * 4}>(c.cda);", [], f"cv{q}"))
    for q, e in outs:
        it.append(("s", f"const float tv{q}{e} = mw_fma(c.nmu, cv{q}[{e}], a{4 * q + e});", [f"cv{q}"], None))
    for q in range(2):
        it.append(("r", f"const f32x4 cg{q} = mw_lds128f<{(16 + 8 * q) * 4}>(c.cda);", [], f"cg{q}"))
    for q, e in outs:
        it.append(("s", f"const float tg{q}{e} = mw_fma(c.nmu, cg{q}[{e}], a{8 + 4 * q + e});", [f"cg{q}"], None))
    for q in range(2):
        it.append(("r", f"const f32x4 dv{q} = mw_lds128f<{(5120 + 8 * q) * 4}>(c.cda);", [], f"dv{q}"))
    for q, e in outs:
        it.append(("s", f"const float va{q}{e} = mw_fma(c.rstd, tv{q}{e}, dv{q}[{e}]);", [f"dv{q}"], None))
    for q in range(2):
        it.append(("r", f"const f32x4 dg{q} = mw_lds128f<{(5120 + 16 + 8 * q) * 4}>(c.cda);", [], f"dg{q}"))
    for q, e in outs:
        it.append(("s", f"const float ga{q}{e} = mw_fma(c.rstd, tg{q}{e}, dg{q}[{e}]);", [f"dg{q}"], None))
    for q, e in outs:
        it.append(("s", f"const float xc{q}{e} = mw_med3(ga{q}{e}, c.lo8, c.hi8);", [], None))
    for q, e in outs:
        it.append(("s", f"const float x2{q}{e} = mw_mul(xc{q}{e}, xc{q}{e});", [], None))
    for q, e in outs:
        it.append(("s", f"const float qa{q}{e} = mw_fma(x2{q}{e}, c.k1, c.k2);", [], None))
    for q, e in outs:
        it.append(("s", f"const float qb{q}{e} = mw_fma(qa{q}{e}, x2{q}{e}, c.k3);", [], None))
    for q, e in outs:
        it.append(("s", f"const float tt{q}{e} = mw_mul(xc{q}{e}, qb{q}{e});", [], None))
    for q, e in outs:
        it.append(("s", f"const float ee{q}{e} = mw_exp2(tt{q}{e});", [], None))
    for q, e in outs:
        it.append(("s", f"const float ss{q}{e} = mw_add(ee{q}{e}, c.one);", [], None))
    for q, e in outs:
        it.append(("s", f"const float rr{q}{e} = mw_rcp(ss{q}{e});", [], None))
    for q, e in outs:
        it.append(("s", f"const float gl{q}{e} = mw_mul(ga{q}{e}, rr{q}{e});", [], None))
    for q, e in outs:
        it.append(("s", f"const float oo{q}{e} = mw_mul(va{q}{e}, gl{q}{e});", [], None))
    for q in range(2):
        for h in range(2):
            it.append(("s", f"const unsigned p{q}{h} = mw_cvt_pk<DT>(oo{q}{2 * h}, oo{q}{2 * h + 1});", [], None))
    for q in range(2):
        it.append(("l", f"mw_lds_write64<0>(c.qwj[{q}], p{q}0, p{q}1);", [], None))
    if store:
        for i in range(4):
            it.append(("l", f"const u32x4 o{i} = mw_lds128<0>(c.qr[{i}]);", [], f"o{i}"))
        for i in range(4):
            it.append(("s", f"mw_store128(c.qst[{i}], o{i}, c.obase);", [f"o{i}"], None))
    return it


def dma_pieces():
    return [f"mw_dma<{kt * 4096}, {kt * 128}>(c.w1dst, c.w1_vj, c.wb);" for kt in range(10)]


def build(name, epi, mf, store, top=True, xload=False):
    st = Stream()
    args = "f32x16 (&accC)[2], f32x16 (&accN)[2], const GwCtx& c"
    st.lines.append(f"template <int DT, int VMC> __device__ __forceinline__ void {name}({args}) {{")
    if top:
        st.lines.append("  mw_wait_vm_barrier<VMC>();")
    mfs = mf_items() if mf else []
    ngap = len(mfs)
    pieces = dma_pieces() if (mf and not NO_DMA) else []
    pre, rest = pieces[:PRE_DMA], pieces[PRE_DMA:]
    items = epilogue(store) if (epi and not NO_EPI) else []
    vgaps = list(range(min(len(rest), ngap), ngap))
    per_gap = {g: [] for g in range(ngap + 1)}
    n_under = min(len(items), MAXV * len(vgaps))
    for k in range(n_under):
        per_gap[vgaps[k * len(vgaps) // n_under]].append(items[k])
    for k in range(n_under, len(items)):
        per_gap[ngap].append(items[k])
    hoist = []
    for g, (_, rn, rc) in enumerate(mfs):
        hoist.append((g, 1, rc, rn, LA))
    for g in range(ngap + 1):
        for kind, code, needs, defs in per_gap[g]:
            if kind == "r":
                hoist.append((g, 0, code, defs, min(LA, 2)))
    hoist.sort(key=lambda h: (h[0], h[1]))
    hp = [0]

    def issue_upto(gap):
        while hp[0] < len(hoist):
            need, _, code, rn, ahead = hoist[hp[0]]
            if need - ahead > gap or st.issued - st.done >= 13:
                break
            st.lds(code, rn)
            hp[0] += 1

    def force(rn):
        while rn not in st.seq:
            need, _, code, r2, ahead = hoist[hp[0]]
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
        st.lines.append("  if (c.has_next) { mw_static_for<40>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(c.xnext); }); "
                        "asm volatile(\"global_load_dwordx2 a[160:161], %0, off\" ::\"v\"(c.snext) : \"memory\"); }")
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
    assert hp[0] == len(hoist), (name, hp[0], len(hoist))
    st.lines.append("}")
    return "\n".join(st.lines)


def main():
    parts = ["// GENERATED by tools/gen_gegluw_stream.py -- do not edit; see that script for the schedule rules.",
             f"// LA = {LA} gaps of LDS-read lookahead, {PRE_DMA} LDS-DMA pieces in front of the first MFMA, <= {MAXV} epilogue statements per gap.", ""]
    parts.append(build("gw_pro", False, True, False, top=False))
    parts.append(build("gw_step", True, True, False))
    parts.append(build("gw_step_st", True, True, True))
    parts.append(build("gw_last", True, False, True, xload=True))
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
