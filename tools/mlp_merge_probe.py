#!/usr/bin/env python3
"""price FC1 -> FC2 of one ViT-B/32 layer (12 800 rows, bf16) as ONE persistent tile stream before building it.

Hardware part (timing only; results are not looked at):
  (a) today's pair: FC1 on gemm256p_kernel (600 tiles of 256 x 256 x 768, LayerNorm-folded QuickGELU epilogue) followed by FC2 on
      gemm160p_kernel (240 tiles of 160 x 256 x 3072, residual epilogue), back to back on one stream — HIP events around 30 pairs;
  (b) each alone;
  (c) the UPPER BOUND of a merged list, the data dependency ignored: one launch of the persistent kernel over 1200 tiles of
      256 x 256 x 768 = FC1's 600 tiles + 600 tiles standing for FC2's 150 x (four K chunks of 768): the same 14 400 K-tile units
      in one banded list on 256 CUs, 450 epilogues too many (subtracted below at the measured epilogue cost);
  (d) FC2 as 150 tiles of 256 x 256 x 3072 on the persistent kernel (one short round): the length of the tile a merged list
      would have to schedule.
Model part: list scheduling of the real merged list on 256 CUs WITH the dependency (an FC2 tile (bm, .) may start once the 12
FC1 tiles of row block bm are complete), K-tile and epilogue costs taken from the measurements above.

usage: python tools/mlp_merge_probe.py   (on the GPU box)"""
import ctypes as C
import heapq
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import mmiss_amd  # noqa: F401,E402
from mmiss_amd import _lib  # noqa: E402

lib = _lib.load()
M, D, MLP = 12800, 768, 3072
g = torch.Generator(device="cuda").manual_seed(0)


def operands(N, K):
    A = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    return A, W, bias


def fold_extras(A, W, K):
    cvec = W.float().sum(1).contiguous()
    parts = A.float().view(M, K // 64, 64)
    stats = torch.stack([parts.sum(-1), (parts * parts).sum(-1)], dim=-1).contiguous()
    return cvec, stats


A1, W1, b1 = operands(MLP, D)
c1, s1 = fold_extras(A1, W1, D)
u = torch.zeros(M, MLP, device="cuda", dtype=torch.bfloat16)
A2, W2, b2 = operands(D, MLP)
x = torch.zeros(M, D, device="cuda", dtype=torch.bfloat16)
st2 = torch.zeros(M, D // 64, 2, device="cuda")
Wm = torch.cat([W1, W1]).contiguous()                      # 6144 x 768: the stand-in list's weights
bm_, cm = torch.cat([b1, b1]).contiguous(), torch.cat([c1, c1]).contiguous()
um = torch.zeros(M, 2 * MLP, device="cuda", dtype=torch.bfloat16)


def fc1(iters=0, ms=None):
    _lib.check(lib.mmiss_dbg_gemm_p256(0, None, 8, A1.data_ptr(), W1.data_ptr(), u.data_ptr(), b1.data_ptr(), c1.data_ptr(),
                                       s1.data_ptr(), 1e-5, M, MLP, D, M, iters, ms))


def fc2(iters=0, ms=None):
    _lib.check(lib.mmiss_dbg_gemm_resid16(0, None, 0, A2.data_ptr(), W2.data_ptr(), x.data_ptr(), b2.data_ptr(), st2.data_ptr(),
                                          M, D, MLP, M, iters, ms))


def merged(iters=0, ms=None):
    _lib.check(lib.mmiss_dbg_gemm_p256(0, None, 8, A1.data_ptr(), Wm.data_ptr(), um.data_ptr(), bm_.data_ptr(), cm.data_ptr(),
                                       s1.data_ptr(), 1e-5, M, 2 * MLP, D, M, iters, ms))


def fc2_256(iters=0, ms=None):    # FC2's shape on 256 x 256 tiles (plain bias epilogue: the residual form exists on the 160-row tile only)
    _lib.check(lib.mmiss_dbg_gemm_p256(0, None, 1, A2.data_ptr(), W2.data_ptr(), x.data_ptr(), b2.data_ptr(), None, None, 1e-5,
                                       M, D, MLP, M, iters, ms))


def alone(fn, iters=30):
    ms = C.c_float(0)
    fn(iters, C.byref(ms))
    return ms.value * 1e3


def pair(reps=30):
    for _ in range(3):
        fc1(); fc2()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fc1(); fc2()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {}
for rep in range(3):
    res.setdefault("FC1 alone (600 tiles, K 768)", []).append(alone(fc1))
    res.setdefault("FC2 alone (p160: 240 tiles, K 3072)", []).append(alone(fc2))
    res.setdefault("FC1 + FC2 back to back", []).append(pair())
    res.setdefault("merged stand-in (1200 tiles, K 768, dependency ignored)", []).append(alone(merged))
    res.setdefault("FC2 shape as 150 tiles of 256 x 256 x 3072", []).append(alone(fc2_256))
for k, v in res.items():
    print("%-62s %s us" % (k, " ".join("%.1f" % t for t in v)), flush=True)

t_fc1, t_pair, t_m, t_fc2_256 = (min(res[k]) for k in ("FC1 alone (600 tiles, K 768)", "FC1 + FC2 back to back",
                                                        "merged stand-in (1200 tiles, K 768, dependency ignored)",
                                                        "FC2 shape as 150 tiles of 256 x 256 x 3072"))
# costs per K-tile and per epilogue from the two persistent launches of known structure: FC1 = 2 rounds + a round of half tiles
# (2.64 tile-times for 2.34 of work), a tile = 12 kt + epi; the one-round K = 3072 launch = 48 kt + epi + fill
kt = (t_fc2_256 - t_fc1 / 2.64) / (48 - 12)
epi = t_fc1 / 2.64 - 12 * kt
print("derived: K-tile %.2f us, epilogue + tile switch %.2f us (FC1 tile %.1f us)" % (kt, epi, 12 * kt + epi))
ub = t_m - 450 * epi / 256
print("upper bound of the merged list (dependency ignored, 450 surplus epilogues removed): %.1f us against %.1f us for the pair "
      "= %.1f %% of the pair" % (ub, t_pair, 100 * (t_pair - ub) / t_pair))


def simulate(order_fc2_first=True, fc2_tile=None, per_rb=3):
    """256 CUs, dynamic list scheduling: FC1 tiles in row-block order; an FC2 tile becomes available when its row block's 12 FC1
    tiles are complete; a free CU takes an available FC2 tile first (the long job), else the next FC1 tile."""
    t1 = 12 * kt + epi
    t2 = fc2_tile if fc2_tile else 48 * kt + epi * 1.5       # (the residual epilogue is the heavier one)
    nrb = M // 256
    free = [(0.0, c) for c in range(256)]
    heapq.heapify(free)
    done_at = {}                  # FC1 tile -> finish time
    rb_left = [12] * nrb
    fc1_next, fc2_ready, fc2_left = 0, [], per_rb * nrb
    events = []                   # (time, row block) FC1 completions not yet folded into fc2_ready
    end = 0.0
    while fc1_next < 12 * nrb or fc2_left > 0:
        t, c = heapq.heappop(free)
        while events and events[0][0] <= t:
            _, rb = heapq.heappop(events)
            rb_left[rb] -= 1
            if rb_left[rb] == 0:
                fc2_ready += [rb] * per_rb
        if fc2_ready and order_fc2_first:
            fc2_ready.pop()
            fc2_left -= 1
            heapq.heappush(free, (t + t2, c))
            end = max(end, t + t2)
        elif fc1_next < 12 * nrb:
            rb = fc1_next // 12
            fc1_next += 1
            heapq.heappush(events, (t + t1, rb))
            heapq.heappush(free, (t + t1, c))
            end = max(end, t + t1)
        elif fc2_ready:
            fc2_ready.pop()
            fc2_left -= 1
            heapq.heappush(free, (t + t2, c))
            end = max(end, t + t2)
        else:                      # nothing available yet: wait for the next FC1 completion
            heapq.heappush(free, (events[0][0], c))
    return end


sim = simulate()
print("list scheduling WITH the dependency (FC2 tile = 48 K-tiles, available once its row block's 12 FC1 tiles are done), a free CU "
      "takes an available FC2 tile first: %.1f us = %+.1f %% against the pair" % (sim, 100 * (t_pair - sim) / t_pair))
sim1 = simulate(order_fc2_first=False)
print("  ... takes FC1 tiles while there are any (FC2 only fills the CUs FC1's last round leaves idle): %.1f us = %+.1f %%"
      % (sim1, 100 * (t_pair - sim1) / t_pair))
sim_half = simulate(fc2_tile=(48 * kt + epi * 1.5) * 0.6, per_rb=6)
print("  ... FC2 in 128-row half tiles (300 jobs of 0.6 tile time), FC2 first: %.1f us = %+.1f %%" % (sim_half, 100 * (t_pair - sim_half) / t_pair))
print("(why the bound is not reachable: the LAST row block's FC2 tiles cannot start before all of FC1 is done — 28 K-tile units on the "
      "whole chip at best — and run 48 units on one CU each: >= 76 units against 56.25 for the ideal packing and ~84 today)")
