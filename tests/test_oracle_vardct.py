"""Cross-checks of the (upstream-unpinned) oracle functions against independent f64 formulas and
invariants: the oracle is a restatement of jxl-oxide's generic code, these tests make sure the
restatement computes the JPEG XL transforms it claims to."""
import ctypes as C
import math

import numpy as np
import pytest
import scipy.ndimage as ndi

from jxl_oxide_amd import abi
from jxl_oxide_amd.synth import VardctWorkload, _load_up_weights


def _idct_mat(n):
    m = np.zeros((n, n))
    for k in range(n):
        for i in range(n):
            m[k, i] = math.cos(i * (2 * k + 1) / n * math.pi / 2) * (math.sqrt(2) if i else 1)
    return m


DCT_TYPES = [0, 4, 5, 6, 7, 8, 9, 10, 11, 18, 19, 20, 21, 22, 23, 24, 25, 26]


@pytest.mark.parametrize("t", DCT_TYPES)
def test_dct_types_match_f64_idct(oracle, t):
    bw, bh = abi.DCT_SELECT_SIZE[t]
    W, H = bw * 8, bh * 8
    rng = np.random.default_rng(t)
    c = rng.uniform(-1, 1, size=(H, W)).astype(np.float32)
    got = oracle.transform_block(c, t)
    exp = _idct_mat(H) @ c.astype(np.float64) @ _idct_mat(W).T
    assert np.allclose(got, exp, atol=2e-5 * math.sqrt(W * H))


@pytest.mark.parametrize("t", range(27))
def test_dc_only_block_is_flat_and_transform_is_linear(oracle, t):
    bw, bh = abi.DCT_SELECT_SIZE[t]
    W, H = bw * 8, bh * 8
    c = np.zeros((H, W), dtype=np.float32)
    c[0, 0] = 0.75
    got = oracle.transform_block(c, t)
    assert np.allclose(got, 0.75, atol=1e-5), "DC coefficient must spread to a constant block"
    rng = np.random.default_rng(100 + t)
    a = rng.uniform(-1, 1, size=(H, W)).astype(np.float32)
    b = rng.uniform(-1, 1, size=(H, W)).astype(np.float32)
    lhs = oracle.transform_block((a + b).astype(np.float32), t)
    rhs = oracle.transform_block(a, t) + oracle.transform_block(b, t)
    assert np.allclose(lhs, rhs, rtol=1e-5, atol=1e-5 * (W * H) ** 0.5 * 4)


def test_dct4x8_and_8x4_are_transposes(oracle):
    rng = np.random.default_rng(9)
    c = rng.uniform(-1, 1, size=(8, 8)).astype(np.float32)
    a = oracle.transform_block(c, abi.TRANSFORM_NAMES.index("Dct4x8"))
    b = oracle.transform_block(c, abi.TRANSFORM_NAMES.index("Dct8x4"))
    assert np.array_equal(a, b.T)


def test_afv_variants_are_flips(oracle):
    rng = np.random.default_rng(10)
    c = rng.uniform(-1, 1, size=(8, 8)).astype(np.float32)
    base = oracle.transform_block(c, abi.TRANSFORM_NAMES.index("Afv0"))
    # AFV1 flips the 4x4 corner horizontally, AFV2 vertically (transform.rs:199-218); the corner
    # block of Afv0 appears mirrored at the flipped position
    a1 = oracle.transform_block(c, abi.TRANSFORM_NAMES.index("Afv1"))
    a2 = oracle.transform_block(c, abi.TRANSFORM_NAMES.index("Afv2"))
    assert np.array_equal(a1[0:4, 4:8], base[0:4, 0:4][:, ::-1])
    assert np.array_equal(a2[4:8, 0:4], base[0:4, 0:4][::-1, :])


def _planes(oracle, wl, stages):
    return oracle.vardct_render(wl.desc(), stages, wl.width, wl.height)[0]


def test_gabor_matches_f64_convolution(oracle):
    wl = VardctWorkload(75, 53, seed=1, epf_iters=0, gabor=True)
    before = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM)
    after = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_GABOR)
    w0, w1 = 0.115169525, 0.061248592
    k = np.array([[w1, w0, w1], [w0, 1.0, w0], [w1, w0, w1]]) / (1 + 4 * w0 + 4 * w1)
    for c in range(3):
        exp = ndi.correlate(before[c].astype(np.float64), k, mode="nearest")
        assert np.allclose(after[c], exp, atol=2e-6 * max(1.0, np.abs(exp).max()))


def _epf_f64(img, sigma, step, fp):
    """Straight f64 restatement of the JPEG XL edge-preserving filter definition."""
    H, W = img.shape[1:]
    pad = 3
    p = np.pad(img.astype(np.float64), ((0, 0), (pad, pad), (pad, pad)), mode="symmetric")
    k1 = [(0, -1), (0, 1), (-1, 0), (1, 0)]
    k2 = [(0, -2), (-1, -1), (0, -1), (1, -1), (-2, 0), (-1, 0), (1, 0), (2, 0), (-1, 1), (0, 1), (1, 1), (0, 2)]
    plus = [(0, -1), (0, 0), (0, 1), (-1, 0), (1, 0)]
    kernel = k2 if step == 0 else k1
    dist = [(0, 0)] if step == 2 else plus
    scale = [40.0, 5.0, 3.5]
    smul = {0: 0.9, 1: 1.0, 2: 6.5}[step]
    yy, xx = np.mgrid[0:H, 0:W]
    border = (((yy + 1) & 6) == 0) | ((xx & 7) == 0) | ((xx & 7) == 7)
    sm = np.where(border, smul * (2.0 / 3.0), smul)
    sig = np.repeat(np.repeat(sigma, 8, axis=0), 8, axis=1)[:H, :W].astype(np.float64)

    def sh(c, dx, dy):
        return p[c, pad + dy:pad + dy + H, pad + dx:pad + dx + W]
    sum_w = np.ones((H, W))
    acc = [img[c].astype(np.float64).copy() for c in range(3)]
    for (kx, ky) in kernel:
        d = np.zeros((H, W))
        for c in range(3):
            a = np.zeros((H, W))
            for (ix, iy) in dist:
                a += np.abs(sh(c, kx + ix, ky + iy) - sh(c, ix, iy))
            d += scale[c] * a
        w = np.maximum(0.0, 1.0 + d * (6.6 * (math.sqrt(0.5) - 1.0) / np.maximum(sig, 1e-9)) * sm)
        sum_w += w
        for c in range(3):
            acc[c] += w * sh(c, kx, ky)
    out = np.stack([acc[c] / sum_w for c in range(3)])
    return np.where(sig[None] < 0.3, img, out)


@pytest.mark.parametrize("iters", [1, 2, 3])
def test_epf_matches_f64_definition(oracle, iters):
    wl = VardctWorkload(61, 44, seed=2 + iters, epf_iters=iters, gabor=False)
    x = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM).astype(np.float64)
    got = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_EPF)
    steps = {1: [1], 2: [1, 2], 3: [0, 1, 2]}[iters]
    for s in steps:
        x = _epf_f64(x, wl.sigma, s, None)
    assert np.allclose(got, x, atol=3e-5)


def test_epf_small_sigma_is_identity(oracle):
    wl = VardctWorkload(40, 40, seed=5, epf_iters=3, gabor=False)
    wl.sigma[:] = 0.1
    a = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM)
    b = _planes(oracle, wl, abi.STAGE_LF | abi.STAGE_TRANSFORM | abi.STAGE_EPF)
    assert np.array_equal(a, b)


def test_lf_stage_formulas(oracle):
    wl = VardctWorkload(200, 120, seed=8, skip_lf_smoothing=True)
    d = wl.desc()
    _, lf = oracle.vardct_render(d, abi.STAGE_LF, wl.width, wl.height, want_lf=True, w8=wl.w8, h8=wl.h8)
    # V1 + V2 in f64 (extra_precision 0 for the single LF group at (0,0))
    sc = lambda m: m * 512.0 / (wl.global_scale * 16.0)
    y = wl.lfq[0].astype(np.float64) * sc(0.25)
    x = wl.lfq[1].astype(np.float64) * sc(1 / 32.0) + (126 - 128) / 84.0 * y
    b = wl.lfq[2].astype(np.float64) * sc(0.5) + (1.0 + (131 - 128) / 84.0) * y
    assert np.allclose(lf[1], y, rtol=1e-6) and np.allclose(lf[0], x, atol=1e-6) and np.allclose(lf[2], b, rtol=1e-5, atol=1e-6)
    # V3: interior changes, border does not; smoothing a constant image is the identity
    wl2 = VardctWorkload(200, 120, seed=8)
    _, lf2 = oracle.vardct_render(wl2.desc(), abi.STAGE_LF, wl2.width, wl2.height, want_lf=True, w8=wl2.w8, h8=wl2.h8)
    assert np.array_equal(lf2[:, 0, :], lf[:, 0, :]) and np.array_equal(lf2[:, :, -1], lf[:, :, -1])
    assert not np.array_equal(lf2[:, 1:-1, 1:-1], lf[:, 1:-1, 1:-1])


@pytest.mark.parametrize("k", [2, 4, 8])
def test_upsampling_matches_independent_numpy(oracle, k):
    up2, up4, up8 = _load_up_weights()
    weights = {2: up2, 4: up4, 8: up8}[k]
    rng = np.random.default_rng(k)
    img = rng.uniform(0, 1, size=(9, 11)).astype(np.float32)
    out = np.zeros((9 * k, 11 * k), dtype=np.float32)
    f32p = oracle.f32p
    oracle.lib().orc_upsample_inner(img.ctypes.data_as(f32p), 11, 11, 9, out.ctypes.data_as(f32p), 11 * k, k,
                                    weights.ctypes.data_as(f32p))
    # independent: build the full (k x k) x (5 x 5) kernel set from the symmetric weight list
    n = k // 2
    full = np.zeros((5 * n, 5 * n))
    it = iter(weights.astype(np.float64))
    for i in range(5 * n):
        for j in range(i, 5 * n):
            full[i, j] = full[j, i] = next(it)
    p = np.pad(img.astype(np.float64), 2, mode="symmetric")
    exp = np.zeros((9 * k, 11 * k))
    for y in range(9 * k):
        for x in range(11 * k):
            ym, xm = y % k, x % k
            my, mx = min(ym, k - 1 - ym), min(xm, k - 1 - xm)
            ker = full[5 * my:5 * my + 5, 5 * mx:5 * mx + 5]
            if ym >= n:
                ker = ker[::-1, :]
            if xm >= n:
                ker = ker[:, ::-1]
            win = p[y // k:y // k + 5, x // k:x // k + 5]
            exp[y, x] = min(max((ker * win).sum(), win.min()), win.max())
    assert np.allclose(out, exp, atol=1e-5)
    const = np.full((6, 7), 0.625, dtype=np.float32)
    out2 = np.zeros((6 * k, 7 * k), dtype=np.float32)
    oracle.lib().orc_upsample_inner(const.ctypes.data_as(f32p), 7, 7, 6, out2.ctypes.data_as(f32p), 7 * k, k,
                                    weights.ctypes.data_as(f32p))
    assert np.allclose(out2, 0.625, atol=1e-6)


def test_lf_frame_equals_own_lf_fed_back(oracle):
    """vardct/mod.rs:175-179: with an LF frame the LF image is taken as it is.  Feeding the oracle's own V1-V3 output
    back as the LF frame must reproduce the ordinary render bit for bit (V4-V8 and the filters read nothing else of
    the LF path), and a different LF frame must change it."""
    from jxl_oxide_amd import abi
    from jxl_oxide_amd.synth import VardctWorkload
    wl = VardctWorkload(300, 264, seed=4)
    ref, lf = oracle.vardct_render(wl.desc(), abi.STAGE_ALL, 300, 264, want_lf=True, w8=wl.w8, h8=wl.h8)
    wl2 = VardctWorkload(300, 264, seed=4, lf_frame=True)
    assert np.array_equal(wl2.coeff, wl.coeff) and np.array_equal(wl2.kind, wl.kind)
    other, _ = oracle.vardct_render(wl2.desc(), abi.STAGE_ALL, 300, 264)
    assert not np.array_equal(other, ref)
    wl2.lf_frame = np.ascontiguousarray(np.pad(lf, ((0, 0), (0, 0), (0, wl2.lf_stride - wl.w8))))
    got, lf2 = oracle.vardct_render(wl2.desc(), abi.STAGE_ALL, 300, 264, want_lf=True, w8=wl.w8, h8=wl.h8)
    assert np.array_equal(lf2.view(np.uint32), lf.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_truncated_groups_keep_their_lf(oracle):
    """allow_partial: a group whose stream ended before its first varblock renders as its LF alone (zero HF):
    untouched groups are bit-identical to the full render at least 3 varblock rows away from the truncated one
    (EPF / Gabor reach), the truncated group differs."""
    from jxl_oxide_amd import abi
    from jxl_oxide_amd.synth import VardctWorkload
    wl = VardctWorkload(768, 256, seed=9, zero_fraction=0.4)
    S = abi.STAGE_LF | abi.STAGE_TRANSFORM
    full, _ = oracle.vardct_render(wl.desc(), S, 768, 256)
    part, _ = oracle.vardct_render(wl.desc(partial={1: 0}), S, 768, 256)
    assert np.array_equal(full[:, :, :256], part[:, :, :256]) and np.array_equal(full[:, :, 512:], part[:, :, 512:])
    assert not np.array_equal(full[:, :, 256:512], part[:, :, 256:512])


def test_scale_f_table_is_its_closed_form():
    """SCALE_F (vardct/dct_common.rs:77-114; pinned textually by tests/test_reference_tables.py) is
    cos(c pi / 512) cos(c pi / 256) cos(c pi / 128): the low-frequency resampling scale of the specification —
    so the table the oracle and the kernels carry is the right table, not merely the same table."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_tables.json")))["tables"]
    key = [k for k in gold if "SCALE_F" in k.upper()]
    assert key, list(gold)[:10]
    tab = np.array([float(v) for v in gold[key[0]]], dtype=np.float64)
    c = np.arange(tab.size, dtype=np.float64)
    closed = np.cos(c * math.pi / 512) * np.cos(c * math.pi / 256) * np.cos(c * math.pi / 128)
    assert np.abs(tab - closed).max() < 1e-7


def test_whole_transform_stage_matches_an_independent_f64_pipeline(oracle):
    """V4 - V8 of a whole frame (square DCT8 / DCT16 / DCT32 varblocks) against numpy f64 written from the formulas,
    not from the oracle: HF dequantisation (vardct/mod.rs:527-537), chroma-from-luma on the coefficients per 64x64
    tile (:589-600), the lowest frequencies from the LF image (forward DCT of the bw x bh LF samples divided by the
    scale_f products, transform_common.rs:40-66), orthonormal-cosine IDCT as matrices."""
    w, h = 264, 200
    wl = VardctWorkload(w, h, seed=12, types=[0, 4, 5], zero_fraction=0.5)
    S = abi.STAGE_LF | abi.STAGE_TRANSFORM
    got, lf = oracle.vardct_render(wl.desc(), S, w, h, want_lf=True, w8=wl.w8, h8=wl.h8)
    d = wl.desc()
    gs, cf = float(d.global_scale), float(d.colour_factor)
    qb, qbn = [float(x) for x in d.quant_bias], float(d.quant_bias_numerator)
    qm = [0.8 ** (int(d.x_qm_scale) - 2), 1.0, 0.8 ** (int(d.b_qm_scale) - 2)]
    hr, wr = wl.coeff.shape[1:]
    deq = np.zeros((3, hr, wr))
    ys, xs = np.nonzero(wl.kind <= 26)
    for cy, cx in zip(ys, xs):
        t = int(wl.kind[cy, cx])
        bw, bh = abi.DCT_SELECT_SIZE[t]
        W, H = 8 * bw, 8 * bh
        for c in range(3):
            q = wl.coeff[c, cy * 8:cy * 8 + H, cx * 8:cx * 8 + W].astype(np.float64)
            v = np.where(np.abs(q) <= 1.0, q * qb[c], q - qbn / np.where(q == 0, 1.0, q))
            v = v * wl.mats[t][c].astype(np.float64).reshape(H, W) * (65536.0 / (gs * float(wl.hf_mul[cy, cx])) * qm[c])
            deq[c, cy * 8:cy * 8 + H, cx * 8:cx * 8 + W] = v
    # chroma from luma on coefficients: the factor of the 64x64 tile the coefficient SAMPLE lies in
    ty, tx = np.arange(hr) // 64, np.arange(wr) // 64
    kx = float(d.base_correlation_x) + wl.xfy.astype(np.float64)[np.minimum(ty, wl.xfy.shape[0] - 1)][:, np.minimum(tx, wl.xfy.shape[1] - 1)] / cf
    kb = float(d.base_correlation_b) + wl.bfy.astype(np.float64)[np.minimum(ty, wl.bfy.shape[0] - 1)][:, np.minimum(tx, wl.bfy.shape[1] - 1)] / cf
    deq[0] += kx * deq[1]
    deq[2] += kb * deq[1]
    scale_f = lambda c: math.cos(c * math.pi / 512) * math.cos(c * math.pi / 256) * math.cos(c * math.pi / 128)
    exp = np.zeros((3, hr, wr))
    for cy, cx in zip(ys, xs):
        t = int(wl.kind[cy, cx])
        bw, bh = abi.DCT_SELECT_SIZE[t]
        W, H = 8 * bw, 8 * bh
        MH, MW = _idct_mat(H), _idct_mat(W)
        for c in range(3):
            blk = deq[c, cy * 8:cy * 8 + H, cx * 8:cx * 8 + W].copy()
            lfb = lf[c, cy:cy + bh, cx:cx + bw].astype(np.float64)
            if bw * bh == 1:
                llf = lfb
            else:
                mh, mw = _idct_mat(bh), _idct_mat(bw)
                llf = (mh.T @ lfb @ mw) / (bh * bw)          # forward DCT = inverse of x = M c
                sy = np.array([scale_f(y << (5 - int(math.log2(bh)))) for y in range(bh)])
                sx = np.array([scale_f(x << (5 - int(math.log2(bw)))) for x in range(bw)])
                llf = llf / np.outer(sy, sx)
            blk[:bh, :bw] = llf
            exp[c, cy * 8:cy * 8 + H, cx * 8:cx * 8 + W] = MH @ blk @ MW.T
    err = np.abs(got.astype(np.float64) - exp[:, :h, :w]).max()
    assert err < 5e-5 * max(1.0, np.abs(exp).max()), err


@pytest.mark.parametrize("t", [0, 1, 2, 3, 12, 13, 14, 15, 16, 17])
def test_small_transforms_have_orthogonal_full_rank_bases(oracle, t):
    """Every 8x8 transform of the specification but Hornuss (DCT8, DCT2, DCT4, DCT4x8, DCT8x4, AFV0-3) maps its 64
    coefficients onto 64 mutually ORTHOGONAL pixel patterns (cosine, Haar-like and AFV bases are orthogonal sets;
    Hornuss codes differences from a block average: full rank, not orthogonal — checked for rank only):
    the 64 x 64 matrix of the oracle's transform, built from unit coefficient blocks, must have a diagonal Gram
    matrix with no vanishing entry.  An indexing slip in a special transform (a swapped quadrant, a wrong
    interleave) breaks this even where linearity and the DC response still hold."""
    T = np.zeros((64, 64))
    for i in range(64):
        c = np.zeros((8, 8), dtype=np.float32)
        c[i // 8, i % 8] = 1.0
        T[:, i] = oracle.transform_block(c, t).astype(np.float64).reshape(-1)
    G = T.T @ T
    if t == 1:
        assert np.linalg.cond(T) < 100.0
        return
    off = G - np.diag(np.diag(G))
    assert np.abs(off).max() < 2e-5 * np.abs(np.diag(G)).max(), np.abs(off).max()
    assert np.diag(G).min() > 1e-3
