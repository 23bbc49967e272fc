"""oracle/fp8_oracle.py (the checker of the fp8 kernels) pinned against torch.float8_e4m3fn — an independent
implementation of the OCP e4m3 format — and against the MX definition (a block's values times 2^(scale-127))."""
import numpy as np
import pytest

from oracle import fp8_oracle as fo


def test_e4m3_table_and_rounding_match_torch():
    torch = pytest.importorskip("torch")
    b = np.arange(256, dtype=np.uint8)
    t = torch.from_numpy(b).view(torch.float8_e4m3fn).float().numpy()
    tab = fo.e4m3_table()
    assert np.array_equal(np.isnan(t), np.isnan(tab)) and np.array_equal(t[~np.isnan(t)], tab[~np.isnan(tab)])
    assert tab[0x7E] == 448.0 and tab[0x01] == 2.0 ** -9
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * 100).astype(np.float32).clip(-448, 448)
    x[:125] = (tab[1:126] + tab[2:127]) / 2          # every tie
    x[125:1125] = rng.choice(tab[~np.isnan(tab)], 1000)  # exact values
    x[1125:1200] = rng.standard_normal(75).astype(np.float32) * 1e-3  # subnormal range
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    np.testing.assert_array_equal(fo.e4m3_encode(x), ref)


def test_mx_quantize_roundtrip_and_scale_layout():
    rng = np.random.default_rng(1)
    y = (rng.standard_normal((7, 1024)) * np.exp(rng.standard_normal((7, 1)) * 3)).astype(np.float32)
    y[2, 100] = 3000.0                                   # an outlier only costs its own block
    for block in (32, 64):
        q, e = fo.mx_quantize(y, block)
        assert e.shape == (7, 32)
        back = fo.mx_dequantize(q, e)
        g = np.abs(y).reshape(7, -1, block).max(axis=2).repeat(block, axis=1)
        assert (np.abs(back - y) <= g * 2.0 ** -4 + 1e-30).all()     # half an ulp of a 3-bit mantissa, relative to the block max
        assert (np.abs(fo.e4m3_decode(q)) <= 448).all() and not np.isnan(fo.e4m3_decode(q)).any()
        # the scale is the smallest power of two that fits the block maximum into 448
        sc = np.exp2(e.astype(np.float64) - 127)
        gmax = np.abs(y).reshape(7, -1, 32).max(axis=2) if block == 32 else np.abs(y).reshape(7, -1, 64).max(axis=2).repeat(2, axis=1)
        assert (gmax <= 448 * sc).all() and (gmax > 448 * sc / 2).all()
    p = fo.permute_scales(e)
    assert p.shape == (7, 32) and fo.scale_row_bytes(768) == 32 and fo.scale_row_bytes(4096) == 128
    np.testing.assert_array_equal(fo.unpermute_scales(p, 1024), e)
    # one dword = the scales of one lane group (k-blocks 4*kt + g) for four consecutive 128-wide K-tiles
    for kt in range(8):
        for g_ in range(4):
            assert fo.scale_offset(4 * kt + g_) == (kt // 4) * 16 + g_ * 4 + kt % 4


def test_weight_quantisation_restatement():
    rng = np.random.default_rng(2)
    w = rng.standard_normal((5, 256)).astype(np.float32) * 0.03
    w[3] = 0.0
    q, sc = fo.quantize_weights(w)
    back = fo.e4m3_decode(q) * sc[:, None]
    assert np.abs(back - w).max() <= np.abs(w).max() * 2.0 ** -4
    assert sc[3] == 1.0 and (q[3] == 0).all()
    assert (np.abs(fo.e4m3_decode(q)).max(axis=1)[[0, 1, 2, 4]] == 448).all()   # the row maximum maps onto 448
