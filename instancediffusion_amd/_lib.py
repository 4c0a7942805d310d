"""ctypes binding of the C ABI declared in ``include/idf.h`` (``libidf_gfx950.so``).

Fails LOUDLY if the shared library is missing: there is no CPU / eager fallback in the product path.
Build with ``instancediffusion_amd/csrc/build.sh`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDF_LIB_PATH", os.path.join(_HERE, "libidf_gfx950.so"))   # override: A/B builds only

IDF_BF16, IDF_F16 = 0, 1
IDF_STAT_GEMM_BIG_LAUNCHES, IDF_STAT_ATTN2_LAUNCHES, IDF_STAT_GEMM_RING_LAUNCHES, IDF_STAT_ATTN8_LAUNCHES, IDF_STAT_GN_EPI_LAUNCHES = 0, 1, 2, 3, 4   # idf_get_stat
IDF_STAT_QKV_ROW_LAUNCHES, IDF_STAT_GEGLU_ROW_LAUNCHES = 6, 7
IDF_TUNE_GEMM_BIG, IDF_TUNE_ATTN2, IDF_TUNE_GEMM_RING, IDF_TUNE_BIG_MIN_EFF, IDF_TUNE_ATTN8, IDF_TUNE_MLP, IDF_TUNE_QKV_ROW, IDF_TUNE_GEGLU_ROW = 0, 1, 2, 3, 4, 5, 6, 7      # idf_set_tuning
EPI_LN_ROW, EPI_LN_COL, EPI_GEGLU_P32 = 512, 1024, 2048
EPI_BIAS, EPI_ROWBIAS, EPI_RES, EPI_GATE, EPI_SILU, EPI_GELU, EPI_GEGLU, EPI_OUT_F32, EPI_OUT_NCHW = \
    1, 2, 4, 8, 16, 32, 64, 128, 256

vp, ll, ci, cf = C.c_void_p, C.c_longlong, C.c_int, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("W", vp), ("out", vp), ("bias", vp), ("rowbias", vp), ("res", vp), ("gate", vp),
                ("M", ci), ("N", ci), ("K", ci),
                ("lda", ci), ("ldw", ci), ("ldo", ci), ("ldr", ci), ("ld_rowbias", ci),
                ("rows_per_batch", ci),
                ("batch", ci), ("strideA", ll), ("strideW", ll), ("strideO", ll), ("strideR", ll),
                ("epi", ci), ("dtype", ci), ("ws", vp), ("ws_bytes", ll),
                ("ln_stats", vp), ("stride_ln_stats", ll), ("ln_c", vp), ("ln_d", vp),
                ("out_stats", vp), ("out_stats_eps", cf),
                ("ln_eps", cf), ("ln_stats_out", vp),
                ("vt_out", vp), ("ld_vt", ci), ("vt_col0", ci)]


class ConvArgs(C.Structure):
    _fields_ = [("x", vp), ("W", vp), ("out", vp), ("bias", vp), ("rowbias", vp), ("res", vp),
                ("B", ci), ("Hin", ci), ("Win", ci), ("Cin", ci), ("Cout", ci),
                ("stride", ci), ("upsample", ci),
                ("ldx", ci), ("ldo", ci), ("ldr", ci), ("ld_rowbias", ci),
                ("n_valid", ci), ("epi", ci), ("dtype", ci), ("ws", vp), ("ws_bytes", ll), ("gn_partial", vp)]


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("ldq", ci), ("strideQ", ll), ("nq", ci),
                ("k0", vp), ("ldk0", ci), ("strideK0", ll), ("vt0", vp), ("ldv0", ci), ("strideV0", ll), ("n0", ci),
                ("k1", vp), ("ldk1", ci), ("strideK1", ll), ("vt1", vp), ("ldv1", ci), ("strideV1", ll), ("n1", ci),
                ("out", vp), ("ldo", ci), ("strideO", ll),
                ("B", ci), ("H", ci), ("d", ci), ("scale", cf), ("dtype", ci),
                ("qbits", vp), ("strideQb", ll), ("kbits0", vp), ("strideKb0", ll), ("kbits1", vp), ("strideKb1", ll)]


class MlpArgs(C.Structure):
    _fields_ = [("x", vp), ("ldx", ci), ("ln_stats", vp), ("w1", vp), ("ldw1", ci), ("cd", vp), ("w2p", vp), ("ldw2", ci),
                ("b2", vp), ("gate", vp), ("out", vp), ("ldo", ci), ("M", ci), ("C", ci), ("dtype", ci)]


# every symbol include/idf.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "idf_abi_version": (ci, []),
    "idf_build_info": (C.c_char_p, []),
    "idf_set_tuning": (ci, [ci, ci]),
    "idf_get_stat": (ll, [ci]),
    "idf_gemm": (ci, [C.POINTER(GemmArgs), vp]),
    "idf_conv3x3": (ci, [C.POINTER(ConvArgs), vp]),
    "idf_mlp_geglu": (ci, [C.POINTER(MlpArgs), vp]),
    "idf_conv_in": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "idf_attention": (ci, [C.POINTER(AttnArgs), vp]),
    "idf_groupnorm_ws_floats": (ll, [ci, ci]),
    "idf_groupnorm": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, cf, ci, ci, vp]),
    "idf_groupnorm_stats": (ci, [vp, vp, ci, ci, ci, ci, ci, vp]),
    "idf_groupnorm_apply": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, ci, ci, vp]),
    "idf_layernorm": (ci, [vp, ci, vp, ci, vp, vp, ci, ci, cf, ci, vp]),
    "idf_row_stats": (ci, [vp, ci, vp, ci, ci, cf, ci, vp]),
    "idf_layernorm_patch2": (ci, [vp, vp, ci, vp, vp, ci, ci, ci, ci, cf, ci, vp]),
    "idf_seg_in_conv": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "idf_dwconv7x7": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "idf_scaleu_concat": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "idf_timestep_embedding": (ci, [vp, vp, ci, ci, ci, vp]),
    "idf_unifusion_embed": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "idf_cfg_combine": (ci, [vp, vp, cf, vp, ll, vp]),
    "idf_plms_update": (ci, [vp, vp, vp, vp, vp, vp, ci, cf, cf, cf, vp, ll, vp]),
    "idf_mis_merge": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "idf_cast_f32_to_16": (ci, [vp, vp, ll, ci, vp]),
    "idf_softmax_rows": (ci, [vp, vp, ll, ci, ll, ll, cf, ci, vp]),
    "idf_pointwise_nchw": (ci, [vp, vp, vp, vp, ci, ci, ci, ll, cf, vp]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
            f"(run instancediffusion_amd/csrc/build.sh or __graft_entry__.build()). "
            f"There is no CPU fallback for the InstanceDiffusion sampling path.")
    # PyTorch-ROCm ships its own libamdhip64.so (same SONAME as /opt/rocm's).  It must be in the process BEFORE this
    # library is dlopen'ed so that our DT_NEEDED entry resolves to that one runtime; loaded the other way round the
    # process holds two HIP runtimes and launches on torch's streams fail with hipErrorNoDevice (seen on the GPU box
    # when build() ran before the first `import torch`).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.idf_abi_version() != 5:
        raise RuntimeError("libidf_gfx950.so ABI version mismatch")
    _lib = lib
    return lib


class IdfError(RuntimeError):
    pass


def check(status: int, what: str):
    if status != 0:
        kind = {-1: "invalid argument", -2: "misaligned pointer/leading dimension", -3: "unsupported"}.get(
            status, f"hipError_t {status}")
        raise IdfError(f"{what} failed: {kind}")
