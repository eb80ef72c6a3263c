"""Default dequantisation matrices — host-side data model that *feeds* the hot path.

Restates jxl-vardct/src/dequant.rs:76-148 (default parameters) and :159-401 (`into_matrix`),
:631-651 (transposed copies) in numpy f32.  In a real integration the Rust host keeps computing
these (`DequantMatrixSet`); the device only consumes them through `JxlGpuVardctDesc.dequant`, so
the parity tests treat them as inputs (`powf` may differ in the last ulp between
libms, which is why they are not outputs of the parity contract).
"""
import numpy as np

from .abi import DCT_SELECT_SIZE, NUM_TRANSFORMS

F = np.float32

SEQ_A = [-1.025, -0.78, -0.65012, -0.19041574, -0.20819396, -0.421064, -0.32733846]
SEQ_B = [-0.30419582, -0.36330363, -0.3566038, -0.34430745, -0.33699593, -0.30180866, -0.27321684]
SEQ_C = [-1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5]
DCT4X8_PARAMS = [
    [2198.0505, -0.96269625, -0.7619425, -0.65511405],
    [764.36554, -0.926302, -0.967523, -0.2784529],
    [527.10754, -1.4594386, -1.4500821, -1.5843723],
]
DCT4_PARAMS = [
    [2200.0, 0.0, 0.0, 0.0],
    [392.0, 0.0, 0.0, 0.0],
    [112.0, -0.25, -0.25, -0.5],
]


def _common_seq(a, b, c):
    return [[a] + SEQ_A, [b] + SEQ_B, [c] + SEQ_C]


# index = dequant_matrix_param_index (dct_select.rs:78-100); (width, height) = dequant_matrix_size
_DCT_PARAMS = {
    0: ((8, 8), [[3150.0, 0.0, -0.4, -0.4, -0.4, -2.0],
                 [560.0, 0.0, -0.3, -0.3, -0.3, -0.3],
                 [512.0, -2.0, -1.0, 0.0, -1.0, -2.0]]),
    4: ((16, 16), [[8996.873, -1.3000778, -0.4942453, -0.43909377, -0.6350102, -0.9017726, -1.6162099],
                   [3191.4836, -0.67424583, -0.80745816, -0.4492584, -0.3586544, -0.3132239, -0.37615025],
                   [1157.504, -2.0531423, -1.4, -0.5068713, -0.4270873, -1.4856834, -4.920914]]),
    5: ((32, 32), [[15718.408, -1.025, -0.98, -0.9012, -0.4, -0.48819396, -0.421064, -0.27],
                   [7305.7637, -0.8041958, -0.76330364, -0.5566038, -0.49785304, -0.43699592, -0.40180868, -0.27321684],
                   [3803.5317, -3.0607336, -2.041327, -2.023565, -0.54953897, -0.4, -0.4, -0.3]]),
    6: ((16, 8), [[7240.7734, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5],
                  [1448.1547, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2],
                  [506.85413, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6]]),
    7: ((32, 8), [[16283.249, -1.7812846, -1.6309059, -1.0382179, -0.85, -0.7, -0.9, -1.2360638],
                  [5089.1577, -0.3200494, -0.3536285, -0.3034, -0.61, -0.5, -0.5, -0.6],
                  [3397.7761, -0.32132736, -0.3450762, -0.7034, -0.9, -1.0, -1.0, -1.1754606]]),
    8: ((32, 16), [[13844.971, -0.971138, -0.658, -0.42026, -0.22712, -0.2206, -0.226, -0.6],
                   [4798.964, -0.6112531, -0.8377079, -0.7901486, -0.26927274, -0.38272768, -0.22924222, -0.20719099],
                   [1807.2369, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5]]),
    11: ((64, 64), _common_seq(23966.166, 8380.191, 4493.024)),
    12: ((64, 32), _common_seq(15358.898, 5597.3604, 2919.9617)),
    13: ((128, 128), _common_seq(47932.332, 16760.383, 8986.048)),
    14: ((128, 64), _common_seq(30717.797, 11194.721, 5839.9233)),
    15: ((256, 256), _common_seq(95864.664, 33520.766, 17972.096)),
    16: ((256, 128), _common_seq(61435.594, 24209.441, 12979.847)),
}

# TransformType -> param index (dct_select.rs:78-100)
PARAM_INDEX = [0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16]


def _mult(x):
    x = F(x)
    return F(1.0) + x if x > 0 else F(1.0) / (F(1.0) - x)


def _interpolate(pos, maxv, bands):
    # dequant.rs:162-179
    n = len(bands)
    if n == 1:
        return np.full(np.shape(pos), bands[0], dtype=F)
    pos = np.asarray(pos, dtype=F)
    scaled_pos = pos * F(n - 1) / F(maxv)
    scaled_index = scaled_pos.astype(np.int64)
    frac = scaled_pos - scaled_index.astype(F)
    bands = np.asarray(bands, dtype=F)
    a = bands[scaled_index]
    b = bands[scaled_index + 1]
    return (a * np.power(b / a, frac, dtype=F)).astype(F)


def _dct_quant_weights(params, width, height):
    # dequant.rs:185-214
    bands = [F(params[0])]
    for v in params[1:]:
        bands.append(F(bands[-1] * _mult(v)))
    x = np.arange(width, dtype=F)[None, :] / F(width - 1)
    y = np.arange(height, dtype=F)[:, None] / F(height - 1)
    dist = np.sqrt(x * x + y * y, dtype=F)
    return _interpolate(dist, F(np.sqrt(F(2.0))) + F(1e-6), bands)  # (height, width)


def _weights_for_param(idx):
    """Returns [3] arrays of shape (height, width) in dequant_matrix_size orientation (weights,
    before the reciprocal)."""
    if idx in _DCT_PARAMS:
        (w, h), params = _DCT_PARAMS[idx]
        return [_dct_quant_weights(p, w, h) for p in params]
    if idx == 1:  # Hornuss, dequant.rs:227-234
        out = []
        for p in [[280.0, 3160.0, 3160.0], [60.0, 864.0, 864.0], [18.0, 200.0, 200.0]]:
            m = np.full(64, p[0], dtype=F)
            m[0] = 1.0
            m[1] = p[1]
            m[8] = p[1]
            m[9] = p[2]
            out.append(m.reshape(8, 8))
        return out
    if idx == 2:  # Dct2, dequant.rs:235-257
        out = []
        for p in [[3840.0, 2560.0, 1280.0, 640.0, 480.0, 300.0],
                  [960.0, 640.0, 320.0, 180.0, 140.0, 120.0],
                  [640.0, 320.0, 128.0, 64.0, 32.0, 16.0]]:
            m = np.zeros(64, dtype=F)
            m[0] = 1.0
            for i, val in enumerate(p):
                dim = 1 << (i // 2)
                if i % 2 == 0:
                    for y in range(dim):
                        for x in range(dim, dim * 2):
                            m[y * 8 + x] = val
                            m[x * 8 + y] = val
                else:
                    for y in range(dim, dim * 2):
                        for x in range(dim, dim * 2):
                            m[y * 8 + x] = val
            out.append(m.reshape(8, 8))
        return out
    if idx == 3:  # Dct4, dequant.rs:258-278
        out = []
        for dp in DCT4_PARAMS:
            mat = _dct_quant_weights(dp, 4, 4).reshape(-1)
            m = np.zeros(64, dtype=F)
            for y in range(4):
                for x in range(4):
                    v = mat[y * 4 + x]
                    m[y * 16 + x * 2] = v
                    m[y * 16 + x * 2 + 1] = v
                    m[(y * 2 + 1) * 8 + x * 2] = v
                    m[(y * 2 + 1) * 8 + x * 2 + 1] = v
            # params = [1.0, 1.0]: the three divisions are by 1.0
            out.append(m.reshape(8, 8))
        return out
    if idx == 9:  # Dct4x8, dequant.rs:279-294
        out = []
        for dp in DCT4X8_PARAMS:
            mat = _dct_quant_weights(dp, 8, 4)  # (4, 8)
            m = np.repeat(mat, 2, axis=0).astype(F)  # each row twice
            out.append(m)
        return out
    if idx == 10:  # Afv, dequant.rs:295-366
        FREQS = [0.0, 0.0, 0.8517779, 5.3777843, 0.0, 0.0, 4.734748, 5.4492455, 1.659827, 4.0,
                 7.275749, 10.423227, 2.6629324, 7.6306577, 8.962389, 12.971662]
        lo, hi = F(FREQS[2]), F(FREQS[15])
        afv_params = [
            [3072.0, 3072.0, 256.0, 256.0, 256.0, 414.0, 0.0, 0.0, 0.0],
            [1024.0, 1024.0, 50.0, 50.0, 50.0, 58.0, 0.0, 0.0, 0.0],
            [384.0, 384.0, 12.0, 12.0, 12.0, 22.0, -0.25, -0.25, -0.25],
        ]
        out = []
        for params, dp, dp4 in zip(afv_params, DCT4X8_PARAMS, DCT4_PARAMS):
            w48 = _dct_quant_weights(dp, 8, 4)
            w44 = _dct_quant_weights(dp4, 4, 4)
            bands = [F(params[5])]
            for p in params[6:]:
                bands.append(F(bands[-1] * _mult(p)))
            m = np.zeros(64, dtype=F)
            for y in range(4):
                for x in range(4):
                    if (x, y) == (0, 0):
                        v = F(1.0)
                    elif (x, y) == (0, 1):
                        v = F(params[2])
                    elif (x, y) == (1, 0):
                        v = F(params[3])
                    elif (x, y) == (1, 1):
                        v = F(params[4])
                    else:
                        v = _interpolate(F(FREQS[y * 4 + x]) - lo, hi - lo + F(1e-6), bands)
                    m[16 * y + 2 * x] = v
            for y in range(4):
                for x in range(8):
                    m[16 * y + 8 + x] = F(params[0]) if (y == 0 and x == 0) else w48[y, x]
                for x in range(4):
                    m[16 * y + 2 * x + 1] = F(params[1]) if (y == 0 and x == 0) else w44[y, x]
            out.append(m.reshape(8, 8))
        return out
    raise ValueError(idx)


_cache = None


def default_dequant_matrices():
    """Returns mats[t][c]: contiguous f32 array, raster (8*bh rows x 8*bw cols) *as applied*
    (jxl-render/src/vardct/mod.rs:516-520): transposed when need_transpose()."""
    global _cache
    if _cache is not None:
        return _cache
    by_param = {}
    for idx in sorted(set(PARAM_INDEX)):
        ws = _weights_for_param(idx)
        mats = []
        for w in ws:
            m = (F(1.0) / w.astype(F)).astype(F)
            assert np.all(m > 0) and np.all(m < 1e8)
            mats.append(m)
        by_param[idx] = mats
    out = []
    special = {1, 2, 3, 12, 13, 14, 15, 16, 17}  # Hornuss, Dct2, Dct4, Dct4x8, Dct8x4, Afv*
    for t in range(NUM_TRANSFORMS):
        bw, bh = DCT_SELECT_SIZE[t]
        need_transpose = (t not in special) and (bh >= bw)  # dct_select.rs:135-151
        mats = []
        for c in range(3):
            m = by_param[PARAM_INDEX[t]][c]  # (height, width) with width >= height
            if need_transpose:
                # dequant.rs:635-650: out[idx] = matrix[(idx % h) * w + idx / h], row length h
                m = np.ascontiguousarray(m.T)
            m = np.ascontiguousarray(m, dtype=F)
            assert m.shape == (bh * 8, bw * 8), (t, m.shape)
            mats.append(m)
        out.append(mats)
    _cache = out
    return out
