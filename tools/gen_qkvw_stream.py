"""Generator of instancediffusion_amd/csrc/qkvw_stream.inc: the straight-line instruction streams of qkv320w_kernel
(qkv_fused.hip: the fused q | k | v projection of a C = 320 transformer block with the activation rows resident in registers).

The scheduler and its rules: tools/mw_streamgen.py.  A wave owns TWO groups of 32 rows of a 256-row tile; a work item = (W chunk, row group), 30 per tile, chunk-major, so that a
chunk stays in LDS for both row groups (half the LDS-DMA pieces and half the barriers per flop of a one-group tile).  One
pipeline step s:
    top      (steps whose MFMAs start a new chunk: RGM == 0) s_waitcnt vmcnt(VMC) + s_barrier: the chunk's LDS-DMA pieces landed;
             the VMC stores issued behind them may still be in flight
    MFMA     first product of item s + 1 = row group RGM of its chunk (64 W rows: two 32 x 32 fragments, two independent
             chains, 40 MFMAs); five of the NEXT chunk's ten LDS-DMA pieces of the wave ride in front of / in its first gaps
    epilogue of item s: LayerNorm fold + bias, 16-bit, through the wave's LDS staging slot, 4 stores
Chunks 0..9 are q | k columns (a lane owns a token, stores token-major rows of 128 B), chunks 10..14 V columns, computed with the
MFMA operands swapped (a lane owns a channel and 32 tokens; the wave's two row groups together hold 64 tokens of a channel: V^T
rows of 128 B, stored by the second item of a chunk).  Variants: pro (MFMA of item 0 only), qq, qv, vv1 / vv0 (epilogue of the first
/ second item of a V chunk), v_ (epilogue of the last item only; it also fetches the next tile's rows BEFORE its stores).

    python tools/gen_qkvw_stream.py            # rewrites the .inc (checked in; CPU test test_qkv320w_stream_is_current...)
"""
import os

from mw_streamgen import ARGS, finish, header, schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "qkvw_stream.inc")
LA = int(os.environ.get("QW_LA", 3))
PRE_DMA = int(os.environ.get("QW_PRE_DMA", 2))
MAXV = int(os.environ.get("QW_MAXV", 6))          # epilogue statements per MFMA gap at most; the rest trails
NO_EPI = os.environ.get("QW_NO_EPI") == "1"       # timing experiments (wrong results)
NO_DMA = os.environ.get("QW_NO_DMA") == "1"
NO_STORE = os.environ.get("QW_NO_STORE", "")      # "q" / "v" / "qv": the q | k / V^T stores left out (timing experiments)


def mf_items(kind):
    """[(statement, read name, read expr)] of the 40 MFMAs of an item; the x fragments of row group RGM are a[80 RGM + 4 ks ..]"""
    fn = "mw_mf1" if kind == "q" else "mw_mf1t"
    out = []
    for i in range(40):
        ks, f = i >> 1, i & 1
        name = f"w_{ks}_{f}"
        first = "true" if ks == 0 else "false"
        out.append((f"{fn}<DT, 20 * RGM + {ks}, {first}>(accN[{f}], {name});", name,
                    f"const u32x4 {name} = mw_lds128<{(ks >> 2) * 8192 + f * 4096}>(c.w1a[{ks & 3}]);"))
    # k-step 20: the LayerNorm mean term -mu c[n] as four 16-bit products (c and -mu split hi + lo; fragment 40 + RGM of the
    # AGPR block holds the lane's token side, the c table in LDS the W-row side): acc leaves the MFMA as x . w - mu c
    for f in range(2):
        name = f"wx_{f}"
        out.append((f"{fn}<DT, 40 + RGM, false>(accN[{f}], {name});", name, f"const u32x4 {name} = mw_lds128<{512 * f}>(c.cxa);"))
    return out


def epi_q():
    """epilogue of a q | k item: [(kind, code, needs, defines)]; kind 'r' hoistable constant read, 'l' in-place LDS operation,
    's' plain statement.  acc = x . w - mu c (the mean term rode the MFMAs): v = rstd acc + d"""
    it = []
    for f in range(2):
        for q in range(4):
            it.append(("r", f"const f32x4 dq{f}{q} = mw_lds128f<{(32 * f + 8 * q) * 4}>(c.cdq);", [], f"dq{f}{q}"))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float v{f}{q}{e} = mw_fma(c.rstd, accC[{f}][{4 * q + e}], dq{f}{q}[{e}]);", [f"dq{f}{q}"], None))
        for q in range(4):
            for h in range(2):
                it.append(("s", f"const unsigned p{f}{q}{h} = mw_cvt_pk<DT>(v{f}{q}{2 * h}, v{f}{q}{2 * h + 1});", [], None))
        for q in range(4):
            it.append(("l", f"mw_lds_write64<0>(c.qw[{4 * f + q}], p{f}{q}0, p{f}{q}1);", [], None))
    for i in range(4):
        it.append(("l", f"const u32x4 o{i} = mw_lds128<0>(c.qr[{i}]);", [], f"o{i}"))
    for i in range(4):
        if "q" not in NO_STORE:
            it.append(("s", f"mw_store128(c.qst[{i}], o{i}, c.obase);", [f"o{i}"], None))
        else:
            it.append(("s", f"asm volatile(\"\" :: \"v\"(o{i}));", [f"o{i}"], None))
    return it


def epi_v(second):
    """epilogue of a V item (a lane owns channel n, its registers the tokens 8 q + 4 hi + e): v = rstd[token] acc + d[n].
    The two row groups of a wave hold 64 consecutive tokens of a channel: the first item of a chunk (row group 0) only writes its
    half of the staging image [32 channels][128 B], the second one writes the other half, reads the image back and stores whole
    128-B lines of V^T (8 channel rows per instruction)."""
    it = []
    rge = 1 if second else 0
    for f in range(2):
        it.append(("r", f"const float dn{f} = mw_lds32f<{128 * f}>(c.cdv);", [], f"dn{f}"))
        for q in range(4):
            it.append(("r", f"const f32x4 s{f}{q} = mw_lds128f<{32 * q}>(c.stt);", [], f"s{f}{q}"))
        for q in range(4):
            for e in range(4):
                it.append(("s", f"const float v{f}{q}{e} = mw_fma(s{f}{q}[{e}], accC[{f}][{4 * q + e}], dn{f});", [f"s{f}{q}", f"dn{f}"], None))
        for q in range(4):
            for h in range(2):
                it.append(("s", f"unsigned p{f}{q}{h} = mw_cvt_pk<DT>(v{f}{q}{2 * h}, v{f}{q}{2 * h + 1});", [], None))
        # token groups G_q = tokens 8 q + 4 hi .. + 3: G0 <-> G2 and G1 <-> G3 across the half-waves leave a lane with 16
        # consecutive tokens 16 hi .. + 15: {p0h, p2h, p1h, p3h} in token order
        for h in range(2):
            it.append(("s", f"mw_swap32(p{f}0{h}, p{f}2{h});", [], None))
        for h in range(2):
            it.append(("s", f"mw_swap32(p{f}1{h}, p{f}3{h});", [], None))
        it.append(("l", f"mw_lds_write128<{4096 * f}>(c.vw[{2 * rge}], u32x4{{p{f}00, p{f}01, p{f}20, p{f}21}});", [], None))
        it.append(("l", f"mw_lds_write128<{4096 * f}>(c.vw[{2 * rge + 1}], u32x4{{p{f}10, p{f}11, p{f}30, p{f}31}});", [], None))
    if second:
        for f in range(2):
            for i in range(4):
                it.append(("l", f"const u32x4 o{f}{i} = mw_lds128<{4096 * f}>(c.qr[{i}]);", [], f"o{f}{i}"))
            for i in range(4):
                if "v" not in NO_STORE:
                    it.append(("s", f"mw_store128(c.vstw[{i}], o{f}{i}, c.vtb[{f}]);", [f"o{f}{i}"], None))
                else:
                    it.append(("s", f"asm volatile(\"\" :: \"v\"(o{f}{i}));", [f"o{f}{i}"], None))
    return it


def dma_pieces():
    """the wave's ten pieces (kt, u) of the next chunk: rows 8 (wave + 4 u) .. + 7 of K-tile kt -- all of them in the step that
    opens a chunk (RGM == 0), right behind its barrier: loads and stores retire through ONE in-order counter, so a piece is only
    known to have landed when every OLDER store has completed; issued early, the pieces have two steps' worth of stores behind
    them and only the stores of three steps back in front (with five pieces per step the top wait sat behind the stores of
    the step before: 100 us of a 395-us launch, NOTES_r06.md)"""
    return [f"if constexpr (RGM == 0) mw_dma<{kt * 8192 + u * 4096}, {kt * 128}>(c.w1dst, c.w1_vj, c.w1b[{u}]);" for kt in range(5) for u in range(2)]


def build(name, epi, mf, top=True, xload=False):
    """epi: None / "q" / "v1" / "v2" (first / second item of a V chunk); mf: None / "q" / "v" """
    decl = f"template <int DT, int VMC, int RGM> __device__ __forceinline__ void {name}({ARGS.format(ctx='QwCtx')})"
    xl = ("if (c.has_next) { mw_static_for<20>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(c.xnext); "
          "mw_load_x2<20 + decltype(kc)::value, decltype(kc)::value>(c.xnext2); }); "
          "asm volatile(\"global_load_dwordx2 a[168:169], %0, off\\n\\tglobal_load_dwordx2 a[170:171], %0, off offset:256\" ::\"v\"(c.snext) : \"memory\"); }")
    items = (epi_q() if epi == "q" else epi_v(epi == "v2")) if (epi and not NO_EPI) else []
    return schedule(decl, mf_items(mf) if mf else [], dma_pieces() if (mf and not NO_DMA) else [], items, LA, PRE_DMA, MAXV,
                    top="if constexpr (RGM == 0) mw_wait_vm_barrier<VMC>();" if top else None, xload=xl if xload else None)


def main():
    parts = header("gen_qkvw_stream.py", LA, PRE_DMA, MAXV)
    parts.append(build("qw_pro", None, "q", top=False))
    parts.append(build("qw_qq", "q", "q"))
    parts.append(build("qw_qv", "q", "v"))
    parts.append(build("qw_vv1", "v1", "v"))        # epilogue of a chunk's first V item (row group 0): RGM = 1 only
    parts.append(build("qw_vv0", "v2", "v"))        # epilogue of its second item: RGM = 0 only
    parts.append(build("qw_v_", "v2", None, xload=True))
    finish(parts, OUT)


if __name__ == "__main__":
    main()
