"""attention_stream_kernel (round 5) against attention_long_kernel on the ViT-L/14 shape: equality of the 16 full tiles,
the last query within tolerance, bf16 and MXFP8 outputs, and isolated timings (us per launch)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
lib = _lib.load()
T = 257


def run(B, H, stream, mx):
    _lib.set_option("attention_stream", stream)
    g = torch.Generator(device="cuda").manual_seed(B * 100 + H)
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
    if mx:
        c8 = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.uint8)
        cs = torch.zeros(B * T, (H * 64 // 512 + (1 if (H * 64) % 512 else 0)) * 16, device="cuda", dtype=torch.uint8)
        f = lambda: _lib.check(lib.mmiss_dbg_attention_mx(0, None, qkv.data_ptr(), c8.data_ptr(), cs.data_ptr(), B, T, H))
        outs = (c8, cs)
    else:
        ctx = torch.zeros(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
        f = lambda: _lib.check(lib.mmiss_dbg_attention(0, None, qkv.data_ptr(), ctx.data_ptr(), B, T, H, 0))
        outs = (ctx,)
    f()
    torch.cuda.synchronize()
    res = [o.clone() for o in outs]
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    return res, best, qkv


_lib.set_option("attention_stream_min_pairs", 1)
for B, H in ((128, 16), (64, 16), (33, 16), (43, 12), (24, 16), (16, 16), (8, 16), (4, 16), (1, 16)):
    for mx in (0, 1):
        old, t_old, qkv = run(B, H, 0, mx)
        new, t_new, _ = run(B, H, 1, mx)
        if mx:
            rows = torch.arange(B * T, device="cuda") % T
            full = rows < 256
            eq_full = torch.equal(old[0][full], new[0][full]) and torch.equal(old[1][full], new[1][full])
            d = (old[0][~full].int() - new[0][~full].int()).abs()
            print(f"B={B} H={H} mx: full tiles equal {eq_full}; last query: fp8 codes differing {int((d > 0).sum())} of {d.numel()}, "
                  f"scales equal {torch.equal(old[1][~full], new[1][~full])}; old {t_old:.1f} us, stream {t_new:.1f} us", flush=True)
        else:
            rows = torch.arange(B * T, device="cuda") % T
            full = rows < 256
            eq_full = torch.equal(old[0][full], new[0][full])
            x = qkv.float().reshape(B, T, 3, H, 64)
            q, k, v = x[:, 256:, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
            ref = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1) @ v).transpose(1, 2).reshape(B, H * 64)
            e_old = (old[0][~full].float() - ref).abs().max().item()
            e_new = (new[0][~full].float() - ref).abs().max().item()
            print(f"B={B} H={H} bf16: full tiles equal {eq_full}; last query |err| vs fp32 softmax: old {e_old:.2e}, stream {e_new:.2e}; "
                  f"old {t_old:.1f} us, stream {t_new:.1f} us", flush=True)
