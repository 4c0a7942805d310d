"""Generator of instancediffusion_amd/csrc/gegluw_stream.inc: the straight-line instruction streams of geglu640w_kernel
(geglu_fused.hip: the GEGLU projection of a C = 640 transformer block -- value * gelu(gate) of LN(x) . W1^T + b1 -- with the
activation rows resident in registers).

The scheduler and its rules: tools/mw_streamgen.py.  One pipeline step i of a 128-row tile (160 chunks of 32 packed W1
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

from mw_streamgen import ARGS, finish, header, schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "instancediffusion_amd", "csrc", "gegluw_stream.inc")
LA = int(os.environ.get("GW_LA", 3))
PRE_DMA = int(os.environ.get("GW_PRE_DMA", 3))
MAXV = int(os.environ.get("GW_MAXV", 6))
NO_EPI = os.environ.get("GW_NO_EPI") == "1"       # timing experiments (wrong results)
NO_DMA = os.environ.get("GW_NO_DMA") == "1"


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
    decl = f"template <int DT, int VMC> __device__ __forceinline__ void {name}({ARGS.format(ctx='GwCtx')})"
    xl = ("if (c.has_next) { mw_static_for<40>([&](auto kc) { mw_load_x2<decltype(kc)::value, decltype(kc)::value>(c.xnext); }); "
          "asm volatile(\"global_load_dwordx2 a[160:161], %0, off\" ::\"v\"(c.snext) : \"memory\"); }")
    return schedule(decl, mf_items() if mf else [], dma_pieces() if (mf and not NO_DMA) else [],
                    epilogue(store) if (epi and not NO_EPI) else [], LA, PRE_DMA, MAXV,
                    top="mw_wait_vm_barrier<VMC>();" if top else None, xload=xl if xload else None)


def main():
    parts = header("gen_gegluw_stream.py", LA, PRE_DMA, MAXV)
    parts.append(build("gw_pro", False, True, False, top=False))
    parts.append(build("gw_step", True, True, False))
    parts.append(build("gw_step_st", True, True, True))
    parts.append(build("gw_last", True, False, True, xload=True))
    finish(parts, OUT)


if __name__ == "__main__":
    main()
