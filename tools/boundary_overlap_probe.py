#!/usr/bin/env python3
"""What could overlap ACROSS the kernel boundaries of one ViT-B/32 layer at batch 256? Upper bound per
boundary, without building the in-kernel hand-off: kernel X and the kernel Y that follows it in a layer, (a) back to back on
one stream, (b) X on one stream and Y on a second one with NO dependency at all — Y's workgroups enter as X's leave the CUs
(the persistent GEMMs fill a CU's LDS: nothing of Y is resident before a workgroup of X exits), which is the most any
band-ready / row-block-ready counter inside one batch could achieve (it can only delay Y's workgroups further). Results of
(b) are meaningless numbers; only the time counts. 12 repetitions per pair, best of 5 rounds, interleaved in one process."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmiss_amd  # noqa
from mmiss_amd import _lib

lib = _lib.load()
B, T, H, d, mlp = 256, 50, 12, 768, 3072
M = 12800
dev = "cuda"
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x16, h, qkv, ctx, u = bf(M, d), bf(M, d), bf(M, 3 * d), bf(M, d), bf(M, mlp)
wqkv, wo, w1, w2 = bf(3 * d, d), bf(d, d), bf(mlp, d), bf(d, mlp)
bq, bo, b1, b2 = (torch.randn(n, device=dev) * 0.01 for n in (3 * d, d, mlp, d))
stats = torch.zeros(M, d // 64, 2, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())

def k_qkv(st): _lib.check(lib.mmiss_dbg_gemm_p256(0, st, 1, P(h), P(wqkv), P(qkv), P(bq), None, None, C.c_float(1e-5), M, 3 * d, d, M, 0, None))
def k_att(st): _lib.check(lib.mmiss_dbg_attention(0, st, P(qkv), P(ctx), B, T, H, 0))
def k_out(st): _lib.check(lib.mmiss_dbg_gemm_resid16(0, st, 0, P(ctx), P(wo), P(x16), P(bo), P(stats), M, d, d, M, 0, None))
def k_fc1(st): _lib.check(lib.mmiss_dbg_gemm_p256(0, st, 2, P(h), P(w1), P(u), P(b1), None, None, C.c_float(1e-5), M, mlp, d, M, 0, None))
def k_fc2(st): _lib.check(lib.mmiss_dbg_gemm_resid16(0, st, 0, P(u), P(w2), P(x16), P(b2), P(stats), M, d, mlp, M, 0, None))
K = {"QKV": k_qkv, "attention": k_att, "out-proj": k_out, "FC1": k_fc1, "FC2": k_fc2}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
p1, p2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
REP = 12

def timed(fn):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / REP * 1e6

def alone(x):
    return timed(lambda: [K[x](p1) for _ in range(REP)])

def serial(x, y):
    def f():
        for _ in range(REP):
            K[x](p1); K[y](p1)
    return timed(f)

def overlapped(x, y):
    # pair i: X on s1, Y on s2; the next pair's X waits for this pair's Y and X (as a layer would), Y never waits for its own X
    evx = [torch.cuda.Event() for _ in range(REP)]
    evy = [torch.cuda.Event() for _ in range(REP)]
    def f():
        for i in range(REP):
            if i: s1.wait_event(evy[i - 1])
            K[x](p1); evx[i].record(s1)
            if i: s2.wait_event(evx[i - 1])
            K[y](p2); evy[i].record(s2)
    return timed(f)

print("one ViT-B/32 layer at batch 256 (12 800 rows): us per pair", flush=True)
tot_s = tot_o = 0.0
for x, y in (("QKV", "attention"), ("attention", "out-proj"), ("out-proj", "FC1"), ("FC1", "FC2"), ("FC2", "QKV")):
    a, b_, s, o = alone(x), alone(y), serial(x, y), overlapped(x, y)
    tot_s += s; tot_o += o
    print(f"{x:10s} -> {y:10s} alone {a:6.1f} + {b_:6.1f}   one stream {s:6.1f}   two streams, no dependency {o:6.1f}   upper bound of the gain {s - o:5.1f} us ({100 * (s - o) / s:4.1f} %)", flush=True)
print(f"sum over the five boundaries (every kernel counted twice): one stream {tot_s:.1f}, two streams {tot_o:.1f}: {100 * (tot_s - tot_o) / tot_s:.1f} % of a layer at most", flush=True)
