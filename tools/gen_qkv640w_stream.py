"""Generator of instancediffusion_amd/csrc/qkv640w_stream.inc: the straight-line instruction streams of qkv640w_kernel
(qkv640_fused.hip: the fused q | k | v projection of a C = 640 transformer block with the activation rows resident in registers;
the skeleton of geglu640w_kernel -- tools/gen_gegluw_stream.py -- with the epilogues of qkv320w_kernel).

The scheduler and its rules: tools/mw_streamgen.py.  One pipeline step i of a 128-row tile (60 chunks of 32 W rows: 40 of q | k columns, 20 of V):
    top      s_waitcnt vmcnt(VMC) + s_barrier: the ten LDS-DMA pieces of step i - 1 (W chunk i + 1) landed; the VMC stores issued
             behind them may still be in flight
    MFMA     first product of chunk i + 1: ONE 32 x 32 fragment, K = 640 = 40 k-steps on two accumulators (even / odd k-steps); V
             chunks with the MFMA operands swapped (a lane owns a channel, its registers the wave's 32 tokens)
    epilogue of chunk i: sum of the two accumulators, LayerNorm fold + bias, 16-bit, through the wave's staging image: q | k chunks
             8 B per lane into [32 tokens][128 B] (two chunks), stored as whole lines every second chunk (4 stores); V chunks
             [32 channels][64 B], 2 stores per chunk
Variants: qm_pro, qm_qq / qm_qq_st (q | k epilogue without / with the store group), qm_qv_st (last q | k chunk, first V MFMAs),
qm_vv, qm_last (epilogue of the last V chunk; it also fetches the next tile's rows BEFORE its stores).

    python tools/gen_qkv640w_stream.py            # rewrites the .inc (checked in; CPU test in tests/test_capi.py)
"""
import os

from mw_streamgen import ARGS, finish, header, schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "qkv640w_stream.inc")
LA = int(os.environ.get("QM_LA", 3))
PRE_DMA = int(os.environ.get("QM_PRE_DMA", 3))
MAXV = int(os.environ.get("QM_MAXV", 6))
NO_EPI = os.environ.get("QM_NO_EPI") == "1"       # timing experiments (wrong results)
NO_DMA = os.environ.get("QM_NO_DMA") == "1"


def mf_items(kind):
    fn = "mw_mf1" if kind == "q" else "mw_mf1t"
    out = []
    for ks in range(40):
        name = f"w_{ks}"
        first = "true" if ks < 2 else "false"
        out.append((f"{fn}<DT, {ks}, {first}>(accN[{ks & 1}], {name});", name,
                    f"const u32x4 {name} = mw_lds128<{(ks >> 2) * 4096}>(c.w1a[{ks & 3}]);"))
    return out


def epi_q(store):
    """q | k chunk (a lane owns a token, its registers the 32 columns 8 q + 4 hi + e of the chunk): v = rstd (acc - mu c) + d"""
    it = []
    for r in range(16):
        it.append(("s", f"const float a{r} = mw_add(accC[0][{r}], accC[1][{r}]);", [], None))
    for q in range(4):
        it.append(("r", f"const f32x4 cq{q} = mw_lds128f<{(8 * q) * 4}>(c.cdq);", [], f"cq{q}"))
    for q in range(4):
        for e in range(4):
            it.append(("s", f"const float t{q}{e} = mw_fma(c.nmu, cq{q}[{e}], a{4 * q + e});", [f"cq{q}"], None))
    for q in range(4):
        it.append(("r", f"const f32x4 dq{q} = mw_lds128f<{(1920 + 8 * q) * 4}>(c.cdq);", [], f"dq{q}"))
    for q in range(4):
        for e in range(4):
            it.append(("s", f"const float v{q}{e} = mw_fma(c.rstd, t{q}{e}, dq{q}[{e}]);", [f"dq{q}"], None))
    for q in range(4):
        for h in range(2):
            it.append(("s", f"const unsigned p{q}{h} = mw_cvt_pk<DT>(v{q}{2 * h}, v{q}{2 * h + 1});", [], None))
    for q in range(4):
        it.append(("l", f"mw_lds_write64<0>(c.qwj[{q}], p{q}0, p{q}1);", [], None))
    if store:
        for i in range(4):
            it.append(("l", f"const u32x4 o{i} = mw_lds128<0>(c.qr[{i}]);", [], f"o{i}"))
        for i in range(4):
            it.append(("s", f"mw_store128(c.qst[{i}], o{i}, c.obase);", [f"o{i}"], None))
    return it


def epi_v():
    """V chunk (a lane owns channel n, its registers the tokens 8 q + 4 hi + e): v = rstd[t] (acc - mu[t] c[n]) + d[n]"""
    it = []
    for r in range(16):
        it.append(("s", f"const float a{r} = mw_add(accC[0][{r}], accC[1][{r}]);", [], None))
    it.append(("r", "const float cn = mw_lds32f<0>(c.cdv);", [], "cn"))
    it.append(("r", f"const float dn = mw_lds32f<{1920 * 4}>(c.cdv);", [], "dn"))
    for q in range(4):
        for h in range(2):
            it.append(("r", f"const f32x4 s{q}{h} = mw_lds128f<{64 * q + 16 * h}>(c.stt);", [], f"s{q}{h}"))
    for q in range(4):
        for e in range(4):
            it.append(("s", f"const float t{q}{e} = mw_fma(s{q}{e >> 1}[{2 * (e & 1)}], cn, a{4 * q + e});", [f"s{q}{e >> 1}", "cn"], None))
    for q in range(4):
        for e in range(4):
            it.append(("s", f"const float v{q}{e} = mw_fma(s{q}{e >> 1}[{2 * (e & 1) + 1}], t{q}{e}, dn);", ["dn"], None))
    for q in range(4):
        for h in range(2):
            it.append(("s", f"unsigned p{q}{h} = mw_cvt_pk<DT>(v{q}{2 * h}, v{q}{2 * h + 1});", [], None))
    for h in range(2):
        it.append(("s", f"mw_swap32(p0{h}, p2{h});", [], None))
    for h in range(2):
        it.append(("s", f"mw_swap32(p1{h}, p3{h});", [], None))
    it.append(("l", "mw_lds_write128<0>(c.vw[0], u32x4{p00, p01, p20, p21});", [], None))
    it.append(("l", "mw_lds_write128<0>(c.vw[1], u32x4{p10, p11, p30, p31});", [], None))
    for i in range(2):
        it.append(("l", f"const u32x4 o{i} = mw_lds128<0>(c.vr[{i}]);", [], f"o{i}"))
    for i in range(2):
        it.append(("s", f"mw_store128(c.vst[{i}], o{i}, c.vtb);", [f"o{i}"], None))
    return it


def dma_pieces():
    return [f"mw_dma<{kt * 4096}, {kt * 128}>(c.w1dst, c.w1_vj, c.wb);" for kt in range(10)]


def build(name, epi, mf, store, top=True, xload=False):
    """epi: None / "q" / "v"; mf: None / "q" / "v" """
    decl = f"template <int DT, int VMC> __device__ __forceinline__ void {name}({ARGS.format(ctx='QmCtx')})"
    xl = ("if (c.has_next) { mw_static_for<40>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(c.xnext); }); "
          "asm volatile(\"global_load_dwordx2 a[160:161], %0, off\" ::\"v\"(c.snext) : \"memory\"); }")
    items = (epi_q(store) if epi == "q" else epi_v()) if (epi and not NO_EPI) else []
    return schedule(decl, mf_items(mf) if mf else [], dma_pieces() if (mf and not NO_DMA) else [], items, LA, PRE_DMA, MAXV,
                    top="mw_wait_vm_barrier<VMC>();" if top else None, xload=xl if xload else None)


def main():
    parts = header("gen_qkv640w_stream.py", LA, PRE_DMA, MAXV)
    parts.append(build("qm_pro", None, "q", False, top=False))
    parts.append(build("qm_qq", "q", "q", False))
    parts.append(build("qm_qq_st", "q", "q", True))
    parts.append(build("qm_qv_st", "q", "v", True))
    parts.append(build("qm_vv", "v", "v", True))
    parts.append(build("qm_last", "v", None, True, xload=True))
    finish(parts, OUT)


if __name__ == "__main__":
    main()
