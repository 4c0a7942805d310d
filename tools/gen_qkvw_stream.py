"""Generator of instancediffusion_amd/csrc/qkvw_stream.inc: the straight-line instruction streams of qkv320w_kernel
(qkv_fused.hip: the fused q | k | v projection of a C = 320 transformer block with the activation rows resident in registers).

Same rules as tools/gen_mlpw_stream.py (every statement `asm volatile`, LDS reads LA gaps ahead of their consumers in FIFO order
with counted lgkmcnt waits -- LDS WRITES count too, they retire in the same queue).  One pipeline step i of a 128-row tile:
    top      s_waitcnt vmcnt(VMC) + s_barrier: the LDS-DMA pieces of step i - 1 (W chunk i + 1) landed; the VMC stores that step
             issued behind them may still be in flight
    MFMA     first product of chunk i + 1 (64 W rows: two 32 x 32 fragments, two independent chains, 40 MFMAs); the 10 LDS-DMA
             pieces of chunk i + 2 ride in front of / in its first gaps
    epilogue of chunk i: LayerNorm fold + bias, 16-bit, through the wave's LDS staging slot, 4 stores
Chunks 0..9 are q | k columns (a lane owns a token, stores token-major rows of 128 B), chunks 10..14 V columns, computed with the
MFMA operands swapped (a lane owns a channel and 32 tokens: V^T rows of 64 B).  Variants: pro (MFMA of chunk 0 only), qq, qv, vv,
v_ (epilogue of chunk 14 only; it also fetches the next tile's rows BEFORE its stores).

    python tools/gen_qkvw_stream.py            # rewrites the .inc (checked in; CPU test test_qkv320w_stream_is_current...)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "qkvw_stream.inc")
LA = int(os.environ.get("QW_LA", 3))
PRE_DMA = int(os.environ.get("QW_PRE_DMA", 3))
MAXV = int(os.environ.get("QW_MAXV", 6))          # epilogue statements per MFMA gap at most; the rest trails
NO_EPI = os.environ.get("QW_NO_EPI") == "1"       # timing experiments (wrong results)
NO_DMA = os.environ.get("QW_NO_DMA") == "1"


class Stream:
    """statements in issue order; LDS operations are counted so that a wait for read r is lgkmcnt(issued - seq(r) - 1)"""

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


def mf_items(kind):
    """[(statement, read name, read expr)] of the 40 MFMAs of a chunk"""
    fn = "mw_mf1" if kind == "q" else "mw_mf1t"
    out = []
    for i in range(40):
        ks, f = i >> 1, i & 1
        name = f"w_{ks}_{f}"
        first = "true" if ks == 0 else "false"
        out.append((f"{fn}<DT, {ks}, {first}>(accN[{f}], {name});", name,
                    f"const u32x4 {name} = mw_lds128<{(ks >> 2) * 8192 + f * 4096}>(c.w1a[{ks & 3}]);"))
    return out


def epi_q():
    """epilogue of a q | k chunk: [(kind, code, needs, defines)]; kind 'r' hoistable constant read, 'l' in-place LDS operation,
    's' plain statement"""
    it = []
    for f in range(2):
        for q in range(4):
            it.append(("r", f"const f32x4 cq{f}{q} = mw_lds128f<{(32 * f + 8 * q) * 4}>(c.cdq);", [], f"cq{f}{q}"))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float t{f}{q}{e} = mw_fma(c.nmu, cq{f}{q}[{e}], accC[{f}][{4 * q + e}]);", [f"cq{f}{q}"], None))
        for q in range(4):
            it.append(("r", f"const f32x4 dq{f}{q} = mw_lds128f<{3840 + (32 * f + 8 * q) * 4}>(c.cdq);", [], f"dq{f}{q}"))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float v{f}{q}{e} = mw_fma(c.rstd, t{f}{q}{e}, dq{f}{q}[{e}]);", [f"dq{f}{q}"], None))
        for q in range(4):
            for h in range(2):
                it.append(("s", f"const unsigned p{f}{q}{h} = mw_cvt_pk<DT>(v{f}{q}{2 * h}, v{f}{q}{2 * h + 1});", [], None))
        for q in range(4):
            it.append(("l", f"mw_lds_write64<0>(c.qw[{4 * f + q}], p{f}{q}0, p{f}{q}1);", [], None))
    for i in range(4):
        it.append(("l", f"const u32x4 o{i} = mw_lds128<0>(c.qr[{i}]);", [], f"o{i}"))
    for i in range(4):
        it.append(("s", f"mw_store128(c.qst[{i}], o{i}, c.obase);", [f"o{i}"], None))
    return it


def epi_v():
    it = []
    for f in range(2):
        it.append(("r", f"const float cn{f} = mw_lds32f<{128 * f}>(c.cdv);", [], f"cn{f}"))
        it.append(("r", f"const float dn{f} = mw_lds32f<{3840 + 128 * f}>(c.cdv);", [], f"dn{f}"))
        for q in range(4):
            for h in range(2):
                it.append(("r", f"const f32x4 s{f}{q}{h} = mw_lds128f<{64 * q + 16 * h}>(c.stt);", [], f"s{f}{q}{h}"))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float t{f}{q}{e} = mw_fma(s{f}{q}{e >> 1}[{2 * (e & 1)}], cn{f}, accC[{f}][{4 * q + e}]);",
                           [f"s{f}{q}{e >> 1}", f"cn{f}"], None))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float v{f}{q}{e} = mw_fma(s{f}{q}{e >> 1}[{2 * (e & 1) + 1}], t{f}{q}{e}, dn{f});", [f"dn{f}"], None))
        for q in range(4):
            for h in range(2):
                it.append(("s", f"unsigned p{f}{q}{h} = mw_cvt_pk<DT>(v{f}{q}{2 * h}, v{f}{q}{2 * h + 1});", [], None))
        # token groups G_q = tokens 8 q + 4 hi .. + 3: G0 <-> G2 and G1 <-> G3 across the half-waves leave a lane with 16
        # consecutive tokens 16 hi .. + 15: {p0h, p2h, p1h, p3h} in token order
        for h in range(2):
            it.append(("s", f"mw_swap32(p{f}0{h}, p{f}2{h});", [], None))
        for h in range(2):
            it.append(("s", f"mw_swap32(p{f}1{h}, p{f}3{h});", [], None))
        it.append(("l", f"mw_lds_write128<{2048 * f}>(c.vw[0], u32x4{{p{f}00, p{f}01, p{f}20, p{f}21}});", [], None))
        it.append(("l", f"mw_lds_write128<{2048 * f}>(c.vw[1], u32x4{{p{f}10, p{f}11, p{f}30, p{f}31}});", [], None))
    for f in range(2):
        for i in range(2):
            it.append(("l", f"const u32x4 o{f}{i} = mw_lds128<{2048 * f}>(c.vr[{i}]);", [], f"o{f}{i}"))
    for f in range(2):
        for i in range(2):
            it.append(("s", f"mw_store128(c.vst[{i}], o{f}{i}, c.vtb[{f}]);", [f"o{f}{i}"], None))
    return it


def dma_pieces():
    return [f"mw_dma<{kt * 8192 + u * 4096}, {kt * 128}>(c.w1dst, c.w1_vj, c.w1b[{u}]);" for kt in range(5) for u in range(2)]


def build(name, epi, mf, top=True, xload=False):
    st = Stream()
    args = "f32x16 (&accC)[2], f32x16 (&accN)[2], const QwCtx& c"
    st.lines.append(f"template <int DT, int VMC> __device__ __forceinline__ void {name}({args}) {{")
    if top:
        st.lines.append("  mw_wait_vm_barrier<VMC>();")
    mfs = mf_items(mf) if mf else []
    ngap = len(mfs)
    pieces = dma_pieces() if (mf and not NO_DMA) else []
    pre, rest = pieces[:PRE_DMA], pieces[PRE_DMA:]
    items = (epi_q() if epi == "q" else epi_v()) if (epi and not NO_EPI) else []
    # epilogue statements over the gaps behind the DMA pieces
    vgaps = list(range(min(len(rest), ngap), ngap))
    per_gap = {g: [] for g in range(ngap + 1)}
    n_under = min(len(items), MAXV * len(vgaps))
    for k in range(n_under):
        per_gap[vgaps[k * len(vgaps) // n_under]].append(items[k])
    for k in range(n_under, len(items)):
        per_gap[ngap].append(items[k])
    # hoistable reads: a fragment read LA gaps ahead of its MFMA; a constant read 2 gaps ahead of the gap of its first use
    hoist = []                              # (need gap, order, code, name)
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

    def force(rn):                          # a consumer is about to wait for rn: it must have been issued
        while rn not in st.seq:
            need, _, code, r2, ahead = hoist[hp[0]]
            st.lds(code, r2)
            hp[0] += 1

    if xload:
        st.lines.append("  if (c.has_next) { mw_static_for<20>([&](auto kc) { mw_load_x<decltype(kc)::value>(c.xnext); }); "
                        "asm volatile(\"global_load_dwordx2 a[240:241], %0, off\" ::\"v\"(c.snext) : \"memory\"); }")
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
        for kind, ecode, needs, defs in per_gap[g]:
            if kind == "r":
                continue
            for r in needs:
                force(r)
            if kind == "l":
                for r in needs:
                    st.wait(r)
                st.lds(ecode, defs)
            else:
                st.stmt(ecode, needs)
    for s in rest[ngap:]:
        st.lines.append("  " + s)
    for kind, ecode, needs, defs in per_gap[ngap]:
        if kind == "r":
            continue
        for r in needs:
            force(r)
        if kind == "l":
            for r in needs:
                st.wait(r)
            st.lds(ecode, defs)
        else:
            st.stmt(ecode, needs)
    assert hp[0] == len(hoist), (name, hp[0], len(hoist))
    st.lines.append("}")
    return "\n".join(st.lines)


def main():
    parts = ["// GENERATED by tools/gen_qkvw_stream.py -- do not edit; see that script for the schedule rules.",
             f"// LA = {LA} gaps of LDS-read lookahead, {PRE_DMA} LDS-DMA pieces in front of the first MFMA, <= {MAXV} epilogue statements per gap.", ""]
    parts.append(build("qw_pro", None, "q", top=False))
    parts.append(build("qw_qq", "q", "q"))
    parts.append(build("qw_qv", "q", "v"))
    parts.append(build("qw_vv", "v", "v"))
    parts.append(build("qw_v_", "v", None, xload=True))
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
