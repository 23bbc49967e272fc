#!/usr/bin/env python3
"""Probe of the block-scaled fp8 MFMA's operand / scale layout through mmiss_dbg_gemm8 (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mmiss_amd  # noqa
from mmiss_amd import _lib
from oracle import fp8_oracle as fo
lib = _lib.load()
M, N, K = 128, 128, 512
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()

def run(A8, e, W8):
    out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    ws = dev(np.ones(N, np.float32)); b = dev(np.zeros(N, np.float32))
    osc = torch.zeros((M, 16), dtype=torch.uint8, device="cuda")
    a, sc, w = dev(A8), dev(fo.permute_scales(e)), dev(W8)   # keep the tensors alive across the call
    _lib.check(lib.mmiss_dbg_gemm8(0, None, 2, 128, a.data_ptr(), sc.data_ptr(), w.data_ptr(),
                                   ws.data_ptr(), b.data_ptr(), out.data_ptr(), osc.data_ptr(), M, N, K))
    torch.cuda.synchronize()
    return out.cpu().numpy()

rng = np.random.default_rng(0)
tab = fo.e4m3_table()
ok = np.nonzero(~np.isnan(tab) & (np.abs(tab) <= 8))[0].astype(np.uint8)
A8 = rng.choice(ok, size=(M, K)); W8 = rng.choice(ok, size=(N, K))
e127 = np.full((M, K // 32), 127, np.uint8)
got = run(A8, e127, W8)
ref = fo.e4m3_decode(A8).astype(np.float64) @ fo.e4m3_decode(W8).astype(np.float64).T
print("unit scales: max abs diff", np.abs(got - ref).max(), "ref max", np.abs(ref).max())
ONE = np.uint8(0x38)
A1 = np.full((M, K), ONE); W1 = np.full((N, K), ONE)
print("all ones:", np.unique(run(A1, e127, W1)))
for b in range(16):
    e = e127.copy(); e[3, b] = 130   # x8 on block b of row 3
    g = run(A1, e, W1)
    rows = np.nonzero((g != K).any(axis=1))[0]
    print("block", b, "-> rows changed", rows[:8], "values", np.unique(g[rows]) if rows.size else None)
# one-hot k: A row 5 has a single 1.0 at k, W = k index pattern? use W8[n, k] = 1 only for n == k % 128
for k in (0, 1, 15, 16, 31, 32, 63, 64, 100, 127, 128, 200):
    A = np.zeros((M, K), np.uint8); A[5, k] = ONE
    W = np.zeros((N, K), np.uint8)
    for kk in range(K): W[kk % 128, kk] = ONE
    g = run(A, e127, W)
    nz = np.argwhere(g != 0)
    print("one-hot k", k, "-> nonzero at", nz[:4].tolist(), g[g != 0][:4])

print("---- random scales, shapes")
def run2(epi, bm, M, N, K, seed=0):
    rng = np.random.default_rng(seed)
    ok = np.nonzero(~np.isnan(tab) & (np.abs(tab) <= 32))[0].astype(np.uint8)
    A8 = rng.choice(ok, size=(M, K)); W8 = rng.choice(ok, size=(N, K))
    e = rng.integers(121, 133, size=(M, K // 32)).astype(np.uint8)
    A = fo.mx_dequantize(A8, e); W = fo.e4m3_decode(W8).astype(np.float64)
    ref = A @ W.T
    out = torch.zeros((M, N), dtype=torch.float32 if epi == 2 else torch.bfloat16, device="cuda")
    ws = dev(np.ones(N, np.float32)); b = dev(np.zeros(N, np.float32))
    osc = torch.zeros((M, fo.scale_row_bytes(N)), dtype=torch.uint8, device="cuda")
    a, sc, w = dev(A8), dev(fo.permute_scales(e)), dev(W8)
    _lib.check(lib.mmiss_dbg_gemm8(0, None, epi, bm, a.data_ptr(), sc.data_ptr(), w.data_ptr(), ws.data_ptr(), b.data_ptr(),
                                   out.data_ptr(), osc.data_ptr(), M, N, K))
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    rel = np.abs(got - ref) / (np.abs(ref) + 1.0)
    bad = rel > 0.02
    print(f"epi {epi} bm {bm} M {M} N {N} K {K}: bad {bad.mean():.3f}; bad rows {np.unique(np.nonzero(bad)[0])[:12]}, bad cols {np.unique(np.nonzero(bad)[1])[:12]}")
    return got, ref, e
for cfg in [(2, 128, 128, 128, 512), (2, 128, 128, 128, 128), (2, 128, 256, 256, 512), (0, 128, 128, 128, 512), (0, 128, 256, 256, 512)]:
    got, ref, e = run2(*cfg)
got, ref, e = run2(2, 128, 128, 128, 128)
print("ratio got/ref row0..3 col0..3\n", (got[:4, :4] / ref[:4, :4]).round(4))
print("e[0:4]", e[:4])

print("---- subnormals")
for code in (0x01, 0x04, 0x07, 0x08, 0x81, 0x87):
    A = np.zeros((M, K), np.uint8); A[5, 7] = code
    W = np.zeros((N, K), np.uint8); W[9, 7] = ONE
    g = run(A, e127, W)
    print(hex(code), "value", tab[code], "-> got", g[5, 9])
    A = np.zeros((M, K), np.uint8); A[5, 7] = ONE
    W = np.zeros((N, K), np.uint8); W[9, 7] = code
    g = run(A, e127, W)
    print(hex(code), "as weight -> got", g[5, 9])
# accumulation precision: 1 big + many small
A = np.full((M, K), 0x08, np.uint8)   # 2^-6 each
A[5, 0] = 0x7E                         # 448
W = np.full((N, K), ONE)
g = run(A, e127, W)
print("448 + 511 * 2^-6 =", 448 + 511 * 2.0 ** -6, "got", g[5, 0], "row without big:", g[6, 0], "expected", 512 * 2.0 ** -6)
ok2 = np.nonzero(~np.isnan(tab) & (np.abs(tab) <= 32) & (np.abs(tab) >= 2.0 ** -6) | (tab == 0))[0].astype(np.uint8)
rng = np.random.default_rng(5)
A8 = rng.choice(ok2, size=(M, K)); W8 = rng.choice(ok2, size=(N, K))
e = rng.integers(121, 133, size=(M, K // 32)).astype(np.uint8)
ref = fo.mx_dequantize(A8, e) @ fo.e4m3_decode(W8).astype(np.float64).T
absacc = np.abs(fo.mx_dequantize(A8, e)) @ np.abs(fo.e4m3_decode(W8).astype(np.float64)).T
out = torch.zeros((M, N), dtype=torch.float32, device="cuda")
ws = dev(np.ones(N, np.float32)); b = dev(np.zeros(N, np.float32)); osc = torch.zeros((M, 16), dtype=torch.uint8, device="cuda")
a, sc, w = dev(A8), dev(fo.permute_scales(e)), dev(W8)
_lib.check(lib.mmiss_dbg_gemm8(0, None, 2, 128, a.data_ptr(), sc.data_ptr(), w.data_ptr(), ws.data_ptr(), b.data_ptr(), out.data_ptr(), osc.data_ptr(), M, N, K))
torch.cuda.synchronize()
got = out.cpu().numpy()
print("no subnormals: max |got-ref| / absacc =", (np.abs(got - ref) / absacc).max(), " (2^-24 =", 2.0 ** -24, ")")
