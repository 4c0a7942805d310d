"""VAE decoder on a real MI355X: the two decoder-specific kernels and the decoder shapes of the shared
kernels vs fp32 PyTorch references, then ``AutoencoderKL.decode`` end to end against goldens of the unmodified
reference and against the live CPU oracle.

Stated tolerance: bf16 storage / fp32 accumulate -> image rel-RMS <= 3e-2 (the same bar as one UNet forward; the
reference decodes in fp32, inference.py:95 is outside its autocast block), fp16 <= 5e-3; kernels 2^-7 of the output max.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_TOL = 2.0 ** -7


@pytest.fixture(scope="module")
def ops():
    from instancediffusion_amd.ops import HipOps
    return HipOps(torch.bfloat16)


@pytest.fixture(scope="module")
def ref():
    from tests.emul_ops import EmulOps
    return EmulOps(torch.float32)


def gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def relmax(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max() / b.float().abs().max().clamp_min(1e-20))


@pytest.mark.parametrize("rows,n,scale", [(8, 64, 1.0), (300, 256, 512 ** -0.5), (2 * 4096, 4096, 512 ** -0.5),
                                          (5, 9216, 0.05)])
def test_softmax_rows(ops, rows, n, scale):
    s = gen((rows, n), 11, 6.0)
    s[0, 3] = 80.0                                                    # one dominant score: exp must not overflow
    want = torch.softmax(s * scale, -1)
    out = ops.softmax_rows(s.cuda(), ops.empty((rows, n)), scale)
    torch.cuda.synchronize()
    assert relmax(out, want) < BF16_TOL
    assert float((out.float().sum(-1) - 1).abs().max()) < 2e-2        # bf16 probabilities still sum to ~1


def test_softmax_rows_f16_and_batched_shape():
    from instancediffusion_amd.ops import HipOps
    o16 = HipOps(torch.float16)
    s = gen((2, 64, 128), 12, 3.0)
    out = o16.softmax_rows(s.cuda(), o16.empty((2, 64, 128)), 0.3)
    torch.cuda.synchronize()
    assert relmax(out, torch.softmax(s * 0.3, -1)) < 2.0 ** -10


def test_pointwise_nchw(ops, ref):
    x, w, b = gen((3, 4, 24, 40), 1), gen((4, 4), 2), gen((4,), 3)
    want = ref.pointwise_nchw(x, w, b, torch.empty(3, 4, 24, 40), 1.0 / 0.18215)
    out = ops.pointwise_nchw(x.cuda(), w.cuda(), b.cuda(), ops.empty((3, 4, 24, 40), torch.float32), 1.0 / 0.18215)
    torch.cuda.synchronize()
    assert relmax(out, want) < 1e-5
    w2 = gen((6, 3), 4)
    out = ops.pointwise_nchw(x[:, :3].contiguous().cuda(), w2.cuda(), None, ops.empty((3, 6, 24, 40), torch.float32))
    assert relmax(out, ref.pointwise_nchw(x[:, :3].contiguous(), w2, None, torch.empty(3, 6, 24, 40))) < 1e-5


def test_conv_in_512_channels_uses_large_lds(ops, ref):
    """The decoder's first conv 4 -> 512 stages 72 KB of fp32 weights in LDS (> the 64 KB default limit)."""
    x, w, b = gen((2, 4, 16, 16), 1), gen((512, 4, 3, 3), 2, 1 / 6.0), gen((512,), 3, 0.1)
    want = ref.conv_in(x, w, b, torch.empty(2, 16, 16, 512))
    out = ops.conv_in(x.cuda(), w.cuda(), b.cuda(), ops.empty((2, 16, 16, 512)))
    torch.cuda.synchronize()
    assert relmax(out, want) < BF16_TOL


@pytest.mark.parametrize("B,H,W,Cin,Cout,up", [(1, 32, 32, 128, 128, 0), (1, 16, 16, 512, 512, 1), (2, 24, 24, 512, 256, 0),
                                               (1, 40, 40, 256, 128, 0), (1, 64, 64, 256, 256, 1)])
def test_conv3x3_decoder_shapes(ops, ref, B, H, W, Cin, Cout, up):
    from instancediffusion_amd.engine import pack_conv3x3
    x = gen((B, H, W, Cin), 1).to(torch.bfloat16)
    w4 = gen((Cout, Cin, 3, 3), 2, (9 * Cin) ** -0.5)
    w = pack_conv3x3(w4).to(torch.bfloat16)
    b = gen((Cout,), 3, 0.1)
    res = gen((B, H << up, W << up, Cout), 4).to(torch.bfloat16)
    want = ref.conv3x3(x.float(), w.float(), torch.empty(B, H << up, W << up, Cout), bias=b, res=res.float(), upsample=up)
    out = ops.conv3x3(x.cuda(), w.cuda(), ops.empty((B, H << up, W << up, Cout)), bias=b.cuda(), res=res.cuda(), upsample=up)
    torch.cuda.synchronize()
    assert relmax(out, want) < BF16_TOL


def test_conv3x3_rgb_out_nchw(ops, ref):
    """conv_out 128 -> 3 (weights zero-padded to 64 rows) storing fp32 NCHW."""
    from instancediffusion_amd.engine import pack_conv3x3
    x = gen((2, 48, 48, 128), 1).to(torch.bfloat16)
    w4 = torch.zeros(64, 128, 3, 3)
    w4[:3] = gen((3, 128, 3, 3), 2, (9 * 128) ** -0.5)
    b = torch.zeros(64)
    b[:3] = gen((3,), 3, 0.1)
    w = pack_conv3x3(w4).to(torch.bfloat16)
    want = ref.conv3x3(x.float(), w.float(), torch.empty(2, 3, 48, 48), bias=b, n_valid=3)
    out = ops.conv3x3(x.cuda(), w.cuda(), ops.empty((2, 3, 48, 48), torch.float32), bias=b.cuda(), n_valid=3)
    torch.cuda.synchronize()
    assert relmax(out, want) < BF16_TOL


def test_attention_gemm_chain_shapes(ops):
    """scores (fp32 out, strided q/k views of one [M, 2C] buffer) and P.V^T (+bias) as the decoder issues them."""
    B, N, C = 2, 256, 512
    qk = gen((B, N, 2 * C), 1).to(torch.bfloat16)
    s = ops.gemm(qk.cuda()[:, :, :C], qk.cuda()[:, :, C:], ops.empty((B, N, N), torch.float32))
    want = torch.matmul(qk[:, :, :C].float(), qk[:, :, C:].float().transpose(1, 2))
    torch.cuda.synchronize()
    assert relmax(s, want) < 1e-5
    p = torch.softmax(want * C ** -0.5, -1).to(torch.bfloat16)
    vt = gen((B, C, N), 2).to(torch.bfloat16)
    bv = gen((C,), 3)
    o = ops.gemm(p.cuda(), vt.cuda(), ops.empty((B, N, C)), bias=bv.cuda())
    torch.cuda.synchronize()
    assert relmax(o, torch.matmul(p.float(), vt.float().transpose(1, 2)) + bv) < BF16_TOL


@pytest.mark.parametrize("B,HW,C", [(1, 4096, 128), (2, 1024, 512), (1, 65536, 128)])
def test_groupnorm_decoder_shapes(ops, ref, B, HW, C):
    x = (gen((B, HW, C), 5) * 2 + 0.5).to(torch.bfloat16)
    gm, bt = 1 + 0.1 * gen((C,), 6), 0.1 * gen((C,), 7)
    want = ref.groupnorm(x.float(), torch.empty(B, HW, C), gm, bt, 1e-6, True)
    out = ops.groupnorm(x.cuda(), ops.empty((B, HW, C)), gm.cuda(), bt.cuda(), 1e-6, True)
    torch.cuda.synchronize()
    assert relmax(out, want) < 2 * BF16_TOL


# ---- end to end ------------------------------------------------------------------------------------------------
def _decode(tag, dtype):
    from tests import cases
    gold = cases.load_golden(tag)
    meta = gold["meta"]
    ae = cases.build_vae(cases.vae_cfg_for(meta["variant"]), meta["salt"])
    ae.compute_dtype = dtype
    z = cases.vae_latent(meta).cuda()
    img = ae.decode(z)
    img2 = ae.decode(z)
    assert torch.equal(img, img2), "hipGraph replay must be bitwise identical to the eager warm-up"
    return img.float().cpu(), gold["img"]


@pytest.mark.parametrize("tag", ["vae_tiny", "vae_full_16"])
def test_vae_decode_matches_reference_golden(tag):
    from tests import cases
    img, want = _decode(tag, torch.bfloat16)
    err = cases.rel_rms(img, want)
    mx = float((img - want).abs().max() / want.pow(2).mean().sqrt())
    print(f"[parity] VAE decode {tag} bf16: rel-rms {err:.3e}  max-abs/rms {mx:.3e}")
    assert torch.isfinite(img).all() and err < 3e-2 and mx < 0.25


def test_vae_decode_fp16():
    from tests import cases
    img, want = _decode("vae_full_16", torch.float16)
    err = cases.rel_rms(img, want)
    print(f"[parity] VAE decode vae_full_16 fp16: rel-rms {err:.3e}")
    assert torch.isfinite(img).all() and err < 5e-3


def test_vae_decode_32x32_latent_vs_live_oracle():
    """Full SD-1.5 KL-f8 decoder on a 32x32 latent (256x256 image, 1024-token mid attention) against the CPU oracle."""
    from oracle import ref_cpu
    from tests import cases
    cfg = cases.vae_cfg_for("full")
    ae = cases.build_vae(cfg)
    z = gen((2, 4, 32, 32), 21, 0.18215 * 4)
    with torch.no_grad():
        want = ref_cpu.vae_decode({k: v.detach() for k, v in ae.state_dict().items()}, cfg, z[:1])
    img = ae.decode(z.cuda()).float().cpu()
    err = cases.rel_rms(img[:1], want)
    print(f"[parity] VAE decode 32x32 latent bf16 vs oracle: rel-rms {err:.3e}")
    assert img.shape == (2, 3, 256, 256) and err < 3e-2


def test_vae_decode_full_size_properties():
    """64x64 latents -> 512x512 images (the reference's inference.py:95 call): finite, batch entries independent
    (decoding [z0, z1, z0] gives bitwise-equal images 0 and 2), chunking over max_decode_batch is transparent."""
    from tests import cases
    ae = cases.build_vae(cases.vae_cfg_for("full"))
    z = gen((3, 4, 64, 64), 22, 0.18215 * 4)
    z[2] = z[0]
    img = ae.decode(z.cuda())
    assert img.shape == (3, 3, 512, 512) and torch.isfinite(img).all()
    same = cases.rel_rms(img[2].float().cpu(), img[0].float().cpu())
    print(f"[property] batch entries 0 and 2 (same latent): bitwise equal = {torch.equal(img[0], img[2])}, rel-rms {same:.1e}")
    assert same < 1e-3 and not torch.equal(img[0], img[1])
    ae.max_decode_batch = 2
    img2 = ae.decode(z.cuda())
    # a different batch size takes different tile schedules only in fp32 summation order -> 16-bit-level agreement
    assert cases.rel_rms(img2.float().cpu(), img.float().cpu()) < 1e-2


def test_vae_decode_full_size_vs_reference_digest():
    """64x64 latent -> 512x512 image against the digest of the unmodified reference's decode (8x8 average pooling of
    the image, its std and its first 32 pixels): parity AT the full size, not only properties."""
    from tests import cases
    gold = cases.load_golden("vae_full_64")
    meta = gold["meta"]
    ae = cases.build_vae(cases.vae_cfg_for("full"), meta["salt"])
    img = ae.decode(cases.vae_latent(meta).cuda()).float().cpu()
    err = cases.rel_rms(torch.nn.functional.avg_pool2d(img, 8), gold["img_pool8"])
    std_err = abs(float(img.std()) - gold["img_fp"]["std"]) / gold["img_fp"]["std"]
    head_err = float((img.flatten()[:32] - gold["img_fp"]["head"]).abs().max()) / gold["img_fp"]["absmax"]
    print(f"[parity] VAE decode 64x64 latent bf16 vs reference digest: pooled rel-rms {err:.3e}, std {std_err:.2e}, "
          f"first pixels {head_err:.2e} of max")
    assert torch.isfinite(img).all() and err < 3e-2 and std_err < 2e-2 and head_err < 5e-2
