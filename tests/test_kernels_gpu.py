"""Kernel-level parity on the MI355X: each hand-written HIP kernel against a plain fp32 restatement of the
same op (tolerances written next to each check). Calls go through the C-ABI (include/mmiss_debug.h)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16(t):
    import torch

    return t.to(torch.bfloat16)


@pytest.fixture(scope="module")
def env():
    import torch
    import mmiss_amd  # noqa: F401
    from mmiss_amd import _lib

    lib = _lib.load()
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return torch, _lib, lib


def _gemm(env, epi, A, W, out, bias=None, aux=None, p0=0, p1=0, bm=0):
    torch, _lib, lib = env
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.mmiss_dbg_gemm(0, None, epi, bm, A.data_ptr(), W.data_ptr(), out.data_ptr(),
                                  bias.data_ptr() if bias is not None else None,
                                  aux.data_ptr() if aux is not None else None, M, N, K, p0, p1))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (384, 768, 768), (1280, 2304, 768), (256, 768, 3072)])
def test_gemm_f32_epilogue(env, M, N, K):
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm(env, _lib.EPI_F32, A, W, out)
    ref = A.float() @ W.float().T
    # bf16 products are exact in fp32; only the accumulation order differs: 1e-4 relative to the row scale
    err = (out - ref).abs().max().item()
    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("bm,M", [(160, 320), (192, 384), (160, 1600), (192, 768)])
def test_gemm_tile_heights(env, bm, M):
    """The 160- and 192-row tiles (picked per shape to fill the CUs evenly) against the same fp32 reference."""
    torch, _lib, lib = env
    N, K = 256, 192
    g = torch.Generator(device="cuda").manual_seed(bm + M)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm(env, _lib.EPI_F32, A, W, out, bm=bm)
    assert (out - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    _gemm(env, _lib.EPI_BIAS_RESID_F32, A, W, x, bias=bias, bm=bm)
    assert torch.allclose(x, x0 + ref + bias, rtol=1e-5, atol=2e-4)
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_QGELU_BF16, A, W, ob, bias=bias, bm=bm)
    r2 = ref + bias
    assert torch.allclose(ob.float(), r2 * torch.sigmoid(1.702 * r2), rtol=2 ** -7, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 512, 128), (512, 256, 192), (768, 768, 768), (1280, 2304, 768), (256, 3072, 3072)])
def test_gemm_256_tile_pipeline(env, M, N, K):
    """The 256x256 phase-pipelined kernel: K-tile counts 1, 2, 3, 12, 48 cover prologue, tail and steady state
    of its counted-vmcnt schedule. Repeated to catch a schedule race (RAW/WAR on the LDS slots)."""
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T
    tol = 2e-4 * max(1.0, ref.abs().max().item())
    for rep in range(5):
        out = torch.full((M, N), float("nan"), device="cuda")
        _gemm(env, _lib.EPI_F32, A, W, out, bm=256)
        assert (out - ref).abs().max().item() <= tol, rep
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_QGELU_BF16, A, W, ob, bias=bias, bm=256)
    r2 = ref + bias
    assert torch.allclose(ob.float(), r2 * torch.sigmoid(1.702 * r2), rtol=2 ** -7, atol=2e-3)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    _gemm(env, _lib.EPI_BIAS_RESID_F32, A, W, x, bias=bias, bm=256)
    assert torch.allclose(x, x0 + ref + bias, rtol=1e-5, atol=2e-4)
    # bit-identical to the 128-row kernel: same per-element k order
    o128 = torch.zeros(M, N, device="cuda")
    o256 = torch.zeros(M, N, device="cuda")
    _lib.set_option("gemm_skinny", 0)  # M = 256 would otherwise take the weight-streaming kernel ...
    _lib.set_option("gemm_splitk", 0)  # ... or, with few tiles and a long K, the split-K path
    try:
        _gemm(env, _lib.EPI_F32, A, W, o128, bm=128)
    finally:
        _lib.set_option("gemm_skinny", 1)
        _lib.set_option("gemm_splitk", 1)
    _gemm(env, _lib.EPI_F32, A, W, o256, bm=256)
    assert torch.equal(o128, o256)


@pytest.mark.parametrize("M,N,K", [(384, 768, 3072), (1280, 768, 3072), (256, 256, 1536), (640, 1024, 4096)])
def test_gemm_split_k(env, M, N, K):
    """Few tiles and a long K: the K loop is cut into slices run by different workgroups (partials in scratch, summed
    in a fixed order by a second kernel that applies the epilogue). Same restatement, same tolerances; repeatable bits."""
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T
    _lib.set_option("gemm_skinny", 0)  # keep the tiled kernel in charge of these row counts
    try:
        out = torch.full((M, N), float("nan"), device="cuda")
        _gemm(env, _lib.EPI_F32, A, W, out, bm=128)
        assert (out - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
        again = torch.zeros(M, N, device="cuda")
        _gemm(env, _lib.EPI_F32, A, W, again, bm=128)
        assert torch.equal(out, again)
        _lib.set_option("gemm_splitk", 0)
        whole = torch.zeros(M, N, device="cuda")
        _gemm(env, _lib.EPI_F32, A, W, whole, bm=128)
        _lib.set_option("gemm_splitk", 1)
        assert not torch.equal(out, whole)  # the split path really ran (different summation order) ...
        assert (out - whole).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())  # ... and agrees
        x0 = torch.randn(M, N, device="cuda", generator=g)
        x = x0.clone()
        _gemm(env, _lib.EPI_BIAS_RESID_F32, A, W, x, bias=bias, bm=128)
        assert torch.allclose(x, x0 + ref + bias, rtol=1e-5, atol=3e-4)
        ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        _gemm(env, _lib.EPI_BIAS_QGELU_BF16, A, W, ob, bias=bias, bm=128)
        r2 = ref + bias
        assert torch.allclose(ob.float(), r2 * torch.sigmoid(1.702 * r2), rtol=2 ** -7, atol=2e-3)
    finally:
        _lib.set_option("gemm_skinny", 1)
        _lib.set_option("gemm_splitk", 1)


def test_gemm_asymmetric_layout(env):
    """A = identity-like, asymmetric W: catches a transposed or permuted C write (guide §3)."""
    torch, _lib, lib = env
    M = N = 128
    K = 128
    A = torch.zeros(M, K, device="cuda")
    A[torch.arange(M), torch.arange(M) % K] = 1.0
    W = (torch.arange(N * K, device="cuda", dtype=torch.float32).reshape(N, K) % 251) / 16.0
    out = torch.zeros(M, N, device="cuda")
    _gemm(env, _lib.EPI_F32, _bf16(A), _bf16(W), out)
    ref = _bf16(A).float() @ _bf16(W).float().T
    assert torch.equal(out, ref)


def test_gemm_bias_bf16_and_qgelu(env):
    torch, _lib, lib = env
    M, N, K = 256, 512, 256
    g = torch.Generator(device="cuda").manual_seed(7)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    ref = A.float() @ W.float().T + bias
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_BF16, A, W, out, bias=bias)
    # output rounded to bf16: half an ulp = 2^-9 relative
    assert torch.allclose(out.float(), ref, rtol=2 ** -8, atol=1e-3)
    out2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_QGELU_BF16, A, W, out2, bias=bias)
    ref2 = ref * torch.sigmoid(1.702 * ref)
    assert torch.allclose(out2.float(), ref2, rtol=2 ** -7, atol=2e-3)


def test_gemm_residual_and_patch(env):
    torch, _lib, lib = env
    M, N, K = 256, 256, 192
    g = torch.Generator(device="cuda").manual_seed(9)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    _gemm(env, _lib.EPI_BIAS_RESID_F32, A, W, x, bias=bias)
    ref = x0 + A.float() @ W.float().T + bias
    assert torch.allclose(x, ref, rtol=1e-5, atol=2e-4)
    # patch epilogue: G = 4 patches per image, T = 5 tokens; rows land at img*T + 1 + patch, + pos[1 + patch]
    G, T = 4, 5
    imgs = M // G
    pos = torch.randn(T, N, device="cuda", generator=g)
    out = torch.zeros(imgs * T, N, device="cuda")
    _gemm(env, _lib.EPI_PATCH_F32, A, W, out, aux=pos, p0=G, p1=T)
    acc = (A.float() @ W.float().T).reshape(imgs, G, N) + pos[1:][None]
    assert torch.allclose(out.reshape(imgs, T, N)[:, 1:], acc, rtol=1e-5, atol=2e-4)
    assert torch.equal(out.reshape(imgs, T, N)[:, 0], torch.zeros(imgs, N, device="cuda"))


@pytest.mark.parametrize("M", [1, 5, 16, 17, 50, 77, 129, 200, 256])
def test_gemm_skinny_all_epilogues(env, M):
    """M <= 256 rows take the weight-streaming kernel (gemm_skinny.h): the reference's one-request-at-a-time regime
    (50 / <=77 token rows), the pruned last layer and the projection head. Same fp32 restatement, same tolerances."""
    torch, _lib, lib = env
    N, K = 384, 768
    _lib.set_option("gemm_skinny_max_m", 256)  # the dispatch stops at 128 rows; the kernel itself goes to 256
    g = torch.Generator(device="cuda").manual_seed(100 + M)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    acc = A.float() @ W.float().T
    out = torch.full((M, N), float("nan"), device="cuda")
    _gemm(env, _lib.EPI_F32, A, W, out)
    assert (out - acc).abs().max().item() <= 2e-4 * max(1.0, acc.abs().max().item())
    ob = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_BF16, A, W, ob, bias=bias)
    assert torch.allclose(ob.float(), acc + bias, rtol=2 ** -8, atol=1e-3)
    og = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    _gemm(env, _lib.EPI_BIAS_QGELU_BF16, A, W, og, bias=bias)
    ref = acc + bias
    assert torch.allclose(og.float(), ref * torch.sigmoid(1.702 * ref), rtol=2 ** -7, atol=2e-3)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    _gemm(env, _lib.EPI_BIAS_RESID_F32, A, W, x, bias=bias)
    assert torch.allclose(x, x0 + acc + bias, rtol=1e-5, atol=2e-4)
    # the tiled kernel on the same operands (option gemm_skinny = 0), rows padded to its tile height
    Mp = (M + 127) // 128 * 128
    Ap = torch.zeros(Mp, K, device="cuda", dtype=torch.bfloat16)
    Ap[:M] = A
    outp = torch.zeros(Mp, N, device="cuda")
    _lib.set_option("gemm_skinny", 0)
    try:
        _gemm(env, _lib.EPI_F32, Ap, W, outp)
    finally:
        _lib.set_option("gemm_skinny", 1)
        _lib.set_option("gemm_skinny_max_m", 0)
    assert (outp[:M] - out).abs().max().item() <= 2e-4 * max(1.0, acc.abs().max().item())


def test_gemm_skinny_patch_epilogue_and_determinism(env):
    torch, _lib, lib = env
    G, T, imgs, N, K = 49, 50, 3, 256, 3072
    M = G * imgs  # 147 rows
    _lib.set_option("gemm_skinny_max_m", 256)
    g = torch.Generator(device="cuda").manual_seed(3)
    A = _bf16(torch.randn(M, K, device="cuda", generator=g))
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    pos = torch.randn(T, N, device="cuda", generator=g)
    out = torch.zeros(imgs * T, N, device="cuda")
    _gemm(env, _lib.EPI_PATCH_F32, A, W, out, aux=pos, p0=G, p1=T)
    acc = (A.float() @ W.float().T).reshape(imgs, G, N) + pos[1:][None]
    assert torch.allclose(out.reshape(imgs, T, N)[:, 1:], acc, rtol=1e-5, atol=3e-4)
    assert torch.equal(out.reshape(imgs, T, N)[:, 0], torch.zeros(imgs, N, device="cuda"))
    again = torch.zeros(imgs * T, N, device="cuda")
    _gemm(env, _lib.EPI_PATCH_F32, A, W, again, aux=pos, p0=G, p1=T)
    _lib.set_option("gemm_skinny_max_m", 0)
    assert torch.equal(out, again)  # the K-quarters are summed in a fixed order


@pytest.mark.parametrize("d", [128, 512, 768, 1024])
def test_layernorm(env, d):
    torch, _lib, lib = env
    M = 333
    g = torch.Generator(device="cuda").manual_seed(d)
    x = torch.randn(M, d, device="cuda", generator=g) * 3 + 0.5
    gam = torch.randn(d, device="cuda", generator=g)
    bet = torch.randn(d, device="cuda", generator=g)
    ref = torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5)
    out = torch.zeros(M, d, device="cuda")
    _lib.check(lib.mmiss_dbg_layernorm(0, None, x.data_ptr(), gam.data_ptr(), bet.data_ptr(), out.data_ptr(), 0, M, d, 1e-5))
    torch.cuda.synchronize()
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    outb = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mmiss_dbg_layernorm(0, None, x.data_ptr(), gam.data_ptr(), bet.data_ptr(), outb.data_ptr(), 1, M, d, 1e-5))
    torch.cuda.synchronize()
    assert torch.allclose(outb.float(), ref, rtol=2 ** -8, atol=1e-3)


def _attn_ref(torch, qkv, B, T, H, causal, with_bound=False):
    """fp32 softmax attention of torch. with_bound: also the per-element bound the kernels are held to (round 6; was a flat
    2e-2): 2.5e-3 + 2^-8 |out| (the output's bf16 rounding, half an ulp) + 4 x 2^-9 sqrt(sum_i p_i^2 v_i^2) (P is rounded to
    bf16, 2^-9 relative per probability, independent roundings: four standard deviations of their sum). On the 0.1-0.3 outputs
    of a 257-key softmax over N(0,1) values that is <= 4e-3 (measured error 2.1e-3)."""
    d = H * 64
    x = qkv.float().reshape(B, T, 3, H, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.triu(torch.full((T, T), float("-inf"), device=qkv.device), diagonal=1)
    p = torch.softmax(s, dim=-1)
    out = (p @ v).transpose(1, 2).reshape(B * T, d)
    if not with_bound:
        return out
    spread = ((p * p) @ (v * v)).sqrt().transpose(1, 2).reshape(B * T, d)
    return out, 2.5e-3 + out.abs() * 2.0 ** -8 + spread * (4 * 2.0 ** -9)


def _assert_attention_close(torch, ctx, ref, bound):
    assert torch.isfinite(ctx.float()).all()
    err = (ctx.float() - ref).abs()
    assert (err <= bound).all(), (err.max().item(), (err - bound).max().item())


@pytest.mark.parametrize("B,T,H,causal", [(3, 50, 12, 0), (2, 77, 8, 1), (2, 16, 2, 1), (1, 5, 2, 0), (2, 257, 4, 0), (1, 248, 3, 1), (5, 33, 2, 1),
                                            (2, 150, 2, 1), (1, 288, 2, 0), (40, 129, 16, 0),
                                            (64, 257, 16, 0), (128, 130, 12, 1), (43, 257, 12, 0), (16, 257, 16, 0)])   # (257 keys from 256 pairs on: attention_stream_kernel)
def test_attention(env, B, T, H, causal):
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    qkv = _bf16(torch.randn(B * T, 3 * H * 64, device="cuda", generator=g))
    ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, causal))
    torch.cuda.synchronize()
    ref, bound = _attn_ref(torch, qkv, B, T, H, bool(causal), with_bound=True)
    _assert_attention_close(torch, ctx, ref, bound)
    # repeatable bits (round 4: inline-asm maxima read MFMA results inside the hazard window until an s_nop was put in front of
    # them — the softmax offsets, and with them the bf16 roundings of P, then differed from launch to launch)
    for _ in range(3):
        ctx2 = torch.zeros_like(ctx)
        _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx2.data_ptr(), B, T, H, causal))
        torch.cuda.synchronize()
        assert torch.equal(ctx, ctx2)


def test_attention_peaked_softmax(env):
    """One key dominating every query (large logits): the max-subtraction path must not overflow."""
    torch, _lib, lib = env
    B, T, H = 1, 50, 1
    qkv = torch.zeros(B * T, 3 * 64, device="cuda")
    qkv[:, 0:64] = 4.0          # q
    qkv[7, 64:128] = 6.0        # key 7 has a huge dot with every q
    qkv[:, 128:192] = torch.arange(T, device="cuda", dtype=torch.float32)[:, None] / 8.0  # v
    qkv = _bf16(qkv)
    ctx = torch.zeros(B * T, 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize()
    ref, bound = _attn_ref(torch, qkv, B, T, H, False, with_bound=True)
    _assert_attention_close(torch, ctx, ref, bound)


@pytest.mark.parametrize("causal", [0, 1])
def test_attention_long_form_rescale_branch(env, causal):
    """The long-sequence kernel (T > 128) is a ONE-pass online softmax whose offset is raised — and the output tile and the
    denominator rescaled — only when a score exceeds it by more than 2^8: a rare, data-dependent branch (guide rule 26:
    it needs an input that FORCES it and a full independent reference). Keys 40, 133 and 200 carry growing spikes against
    every query (logits ~ +12, +40, +90 over the rest: each raises the maximum past the threshold in a later key tile),
    a fourth query-specific spike sits at key 250 for queries 100..119 only, and key 7 is a mild +3 that must NOT rescale.
    Against the fp32 softmax of torch on the whole tensor."""
    torch, _lib, lib = env
    B, T, H = 2, 257, 3
    g = torch.Generator(device="cuda").manual_seed(77)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda", generator=g) * 0.5
    x = qkv.view(B, T, 3, H, 64)
    x[:, :, 0, :, 0] = 4.0                       # every query: component 0 = 4
    for key, val in ((7, 6.0), (40, 24.0), (133, 80.0), (200, 180.0)):
        x[:, key, 1, :, 0] = val                 # logit += 4 * val / 8
    x[:, 100:120, 0, :, 1] = 8.0                 # queries 100..119: a second direction ...
    x[:, 250, 1, :, 1] = 120.0                   # ... that only key 250 answers (+120 for them, 0 for the others' component 1 ~ N(0, .5))
    qkv = _bf16(qkv)
    ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, causal))
    torch.cuda.synchronize()
    ref, bound = _attn_ref(torch, qkv, B, T, H, bool(causal), with_bound=True)
    _assert_attention_close(torch, ctx, ref, bound)


@pytest.mark.parametrize("B,H", [(2, 3), (64, 16)])   # attention_long_kernel / attention_stream_kernel (>= 256 pairs)
def test_attention_257_keys_last_query_every_key_counts(env, B, H):
    """The 257th query of ViT-L/14 takes its own code path in both long kernels (the odd last key tile; in the streaming kernel
    it is split by keys over nine waves and recombined), so it is compared ALONE, on data where losing any single key shows:
    V = +-8 in every component, mild scores, so a dropped key moves every output by ~8/257 = 0.03 — six times the bound —
    which the test checks of its own reference first (the bound must be able to fail)."""
    torch, _lib, lib = env
    T = 257
    g = torch.Generator(device="cuda").manual_seed(257 + B)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda", generator=g) * 0.3
    x = qkv.view(B, T, 3, H, 64)
    x[:, :, 2] = torch.where(torch.rand(B, T, H, 64, device="cuda", generator=g) < 0.5, -8.0, 8.0)
    qkv = _bf16(qkv)
    ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
    torch.cuda.synchronize()
    ref, bound = _attn_ref(torch, qkv, B, T, H, False, with_bound=True)
    last = torch.arange(B, device="cuda") * T + (T - 1)
    assert bound[last].max().item() < 2e-2                                   # (2.5e-3 + 2^-8 |out| <= 8e-3 + 4e-3 here; one lost key: 3.1e-2)
    _assert_attention_close(torch, ctx[last], ref[last], bound[last])
    _assert_attention_close(torch, ctx, ref, bound)
    # the power of the test: the same reference with ONE key removed (the last, the first, one in the middle) is outside the bound
    xf = qkv.float().reshape(B, T, 3, H, 64)
    q = xf[:, T - 1:, 0].transpose(1, 2)
    for drop in (T - 1, 0, 130):
        keep = [j for j in range(T) if j != drop]
        k, v = xf[:, keep, 1].transpose(1, 2), xf[:, keep, 2].transpose(1, 2)
        p = torch.softmax((q @ k.transpose(-1, -2)) * 0.125, dim=-1)
        wrong = (p @ v).transpose(1, 2).reshape(B, H * 64)
        assert ((wrong - ref[last]).abs() > bound[last]).float().mean().item() > 0.9, drop


@pytest.mark.parametrize("S,P", [(224, 32), (64, 32), (224, 14)])
def test_im2col(env, S, P):
    torch, _lib, lib = env
    B = 3
    G = S // P
    Kreal = 3 * P * P
    Kp = (Kreal + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(S + P)
    px = torch.randn(B, 3, S, S, device="cuda", generator=g)
    out = torch.full((B * G * G, Kp), 7.0, device="cuda").to(torch.bfloat16)
    _lib.check(lib.mmiss_dbg_im2col(0, None, px.data_ptr(), out.data_ptr(), B, S, P, Kp))
    torch.cuda.synchronize()
    ref = px.reshape(B, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(B * G * G, Kreal).to(torch.bfloat16)
    assert torch.equal(out[:, :Kreal], ref)
    assert torch.equal(out[:, Kreal:].float(), torch.zeros(B * G * G, Kp - Kreal, device="cuda"))


def _p256(env, epi, A, W, out, bias, aux=None, stats=None, m_valid=None, eps=1e-5):
    torch, _lib, lib = env
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.mmiss_dbg_gemm_p256(0, None, epi, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                       aux.data_ptr() if aux is not None else None,
                                       stats.data_ptr() if stats is not None else None, eps, M, N, K,
                                       M if m_valid is None else m_valid, 0, None))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,mv", [(512, 512, 256, 512), (2560, 2304, 768, 2560), (768, 1536, 512, 700),
                                      (12800, 2304, 768, 12800), (12800, 3072, 768, 12750), (19712, 2048, 512, 19712),
                                      (1792, 9984, 512, 1792), (33024, 1024, 768, 32896)])
def test_gemm_p256_persistent(env, M, N, K, mv):
    """The persistent 256 x 256 kernel (gemm_bf16_p256.h): one tile per workgroup (T < 256 tiles), 1-2 and 2-3 tiles per
    workgroup as ONE K stream (450 / 600 / 616 tiles: counted waits across the epilogue stores, LDS-DMA'd bias and
    LayerNorm statistics, ragged per-XCD ranges), 273 tiles over 7 row blocks (too few row blocks for the per-XCD ranges:
    the global order), 516 tiles of a 129-row-block ViT-L/14 batch, pad rows. Plain epilogues bit-identical to the one-tile-per-workgroup
    256 x 256 kernel (same k order, same formula); folded-LayerNorm epilogues against an fp32 restatement. Repeated to
    catch a schedule race."""
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    x = torch.randn(M, K, device="cuda", generator=g) * (1 + torch.rand(M, 1, device="cuda", generator=g)) + \
        0.5 * torch.randn(M, 1, device="cuda", generator=g)
    A = _bf16(x)
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    acc = A.float() @ W.float().T
    for epi in (_lib.EPI_BIAS_BF16, _lib.EPI_BIAS_QGELU_BF16):
        want = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        _gemm(env, epi, A, W, want, bias=bias, bm=256)
        r = acc + bias
        if epi == _lib.EPI_BIAS_QGELU_BF16:
            r = r * torch.sigmoid(1.702 * r)
        assert torch.allclose(want.float(), r, rtol=2 ** -7, atol=2e-3)
        for rep in range(3):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            _p256(env, epi, A, W, out, bias, m_valid=mv)
            assert torch.equal(out[:mv].view(torch.int16), want[:mv].view(torch.int16)), (epi, rep)
            if mv < M:   # pad rows: untouched except the dump row M - 1
                assert torch.isnan(out[mv:M - 1].float()).all() and torch.isfinite(out[M - 1].float()).all()
    # folded LayerNorm: A = raw rows, W = gamma-folded weights, c[n] = sum_k W[n,k], statistics per 64 columns
    a32 = A.float()
    parts = a32.view(M, K // 64, 64)
    stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()     # [M][K/64][2]
    cvec = W.float().sum(1).contiguous()
    mean = a32.mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt((a32 * a32).mean(1, keepdim=True) - mean * mean + 1e-5)
    ref = rstd * (acc - mean * cvec) + bias
    for epi, r in ((7, ref), (8, ref * torch.sigmoid(1.702 * ref))):
        if K not in (512, 768):
            break            # (the raw statistics of a tile must fit beside the staging buffers: hidden 512 / 768 only)
        for rep in range(3):
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            _p256(env, epi, A, W, out, bias, aux=cvec, stats=stats, m_valid=mv)
            assert torch.allclose(out[:mv].float(), r[:mv], rtol=2 ** -7, atol=4e-3), (epi, rep, (out[:mv].float() - r[:mv]).abs().max())


def _resid16(env, variant, A, W, out, bias, stats=None, m_valid=None):
    torch, _lib, lib = env
    M, K = A.shape
    N = W.shape[0]
    _lib.check(lib.mmiss_dbg_gemm_resid16(0, None, variant, A.data_ptr(), W.data_ptr(), out.data_ptr(), bias.data_ptr(),
                                          stats.data_ptr() if stats is not None else None, M, N, K,
                                          M if m_valid is None else m_valid, 0, None))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,mv", [(12800, 768, 768, 12800), (12800, 768, 3072, 12800), (2560, 768, 768, 2500),
                                      (19840, 512, 2048, 19712), (160, 256, 256, 160), (6400, 768, 3072, 6400),
                                      (1280, 1024, 4096, 1280), (33120, 1024, 1024, 33024)])
def test_gemm_p160_resid16(env, M, N, K, mv):
    """The 160 x 256 tile on the staggered loop (gemm_bf16_p160.h: out-projection / FC2 on the bf16 residual stream): uneven m
    halves (48 + 32 rows per wave), filler LDS-DMA pieces, the residual rows prefetched before the K loop, pad rows.
    Bit-identical to the 128-column kernel of the same epilogue — same k order inside the 16 x 16 x 32 MFMA chain, same
    bf16 rounding point, same statistics arithmetic — and within bf16 rounding of an fp32 restatement. Repeated with
    changing inputs to catch a schedule race."""
    torch, _lib, lib = env
    g = torch.Generator(device="cuda").manual_seed(7 * M + 3 * N + K)
    W = _bf16(torch.randn(N, K, device="cuda", generator=g) * K ** -0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    for rep in range(3):
        A = _bf16(torch.randn(M, K, device="cuda", generator=g))
        x0 = _bf16(torch.randn(M, N, device="cuda", generator=g))
        ref = x0.float() + (A.float() @ W.float().T + bias)
        want = x0.clone()
        st_want = torch.zeros(M, N // 64, 2, device="cuda")
        _resid16(env, 160, A, W, want, bias, stats=st_want)
        out = x0.clone()
        st = torch.full((M, N // 64, 2), float("nan"), device="cuda")
        _resid16(env, 0, A, W, out, bias, stats=st, m_valid=mv)
        assert torch.allclose(out[:mv].float(), ref[:mv], rtol=2 ** -7, atol=4e-3), (rep, (out[:mv].float() - ref[:mv]).abs().max())
        assert torch.equal(out[:mv].view(torch.int16), want[:mv].view(torch.int16)), rep
        assert torch.equal(st[:mv].view(torch.int32), st_want[:mv].view(torch.int32)), rep
        if mv < M:   # pad rows: untouched except the dump row M - 1; their statistics are not written
            assert torch.equal(out[mv:M - 1].view(torch.int16), x0[mv:M - 1].view(torch.int16))
            assert torch.isnan(st[mv:]).all()


def test_streaming_attention_equals_the_split_form_on_the_full_tiles(env):
    """attention_stream_kernel (round 5: 257 keys, >= 256 (item, head) pairs; persistent workgroups, the next pair's K/V by LDS-DMA,
    the 257th query merged from nine partial softmaxes) against attention_long_kernel on the same input: the 256 queries of the
    16 full tiles run the same instructions in the same order — equal bytes —, the last query within the bf16 tolerance of the
    fp32 softmax; with and without the rescale branch forced (spiked keys), and a pair count that is not a multiple of the grid."""
    torch, _lib, lib = env
    B, T, H = 35, 257, 16
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda", generator=g) * 0.7
    x = qkv.view(B, T, 3, H, 64)
    x[:, :, 0, :, 0] = 3.0
    for key, val in ((40, 20.0), (133, 70.0), (256, 150.0)):   # growing spikes, the last one on the lone key of the 17th tile
        x[: B // 2, key, 1, :, 0] = val
    qkv = _bf16(qkv)
    outs = []
    try:
        for stream in (0, 1):
            _lib.set_option("attention_stream", stream)
            ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
            _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
            torch.cuda.synchronize()
            outs.append(ctx)
    finally:
        _lib.set_option("attention_stream", 1)
    rows = torch.arange(B * T, device="cuda") % T
    full = rows < 256
    assert torch.equal(outs[0][full], outs[1][full])
    ref, bound = _attn_ref(torch, qkv, B, T, H, False, with_bound=True)
    for o in outs:
        _assert_attention_close(torch, o, ref, bound)
        _assert_attention_close(torch, o[~full], ref[~full], bound[~full])    # the 257th query alone
