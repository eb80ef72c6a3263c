"""Synthetic boundary-state generator: valid *post-entropy-decode* state of a JPEG XL frame.

There is no JPEG XL encoder (and no fixture) in this environment, so workloads are generated at
the device boundary exactly as SURVEY.md §8(d) specifies: coefficient planes as `write_hf_coeff`
leaves them, varblock tilings as `HfMetadata::parse` would produce (jxl-vardct/src/
hf_metadata.rs:55-229: raster placement, never across a 256-px group, Occupied cells, hf_mul>0,
sigma from sharpness), LF-quant planes, CfL maps and the default dequant matrices.

`VardctWorkload.desc()` builds the `JxlGpuVardctDesc` that both libjxlgpu.so and the oracle take.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import DCT_SELECT_SIZE
from .dequant import default_dequant_matrices

SEED_BASE = 0x4A584C00

# jxl-image/src/lib.rs:533- (data tables)
D_UP2 = np.array([
    -0.01716200, -0.03452303, -0.04022174, -0.02921014, -0.00624645,
    0.14111091, 0.28896755, 0.00278718, -0.01610267, 0.56661550,
    0.03777607, -0.01986694, -0.03144731, -0.01185068, -0.00213539], dtype=np.float32)


def _load_up_weights():
    import os
    import re
    # parsed once from the committed copy of the table (tests/golden/upsampling_weights.npz)
    here = os.path.dirname(os.path.abspath(__file__))
    p = os.path.join(here, "upsampling_weights.npz")
    z = np.load(p)
    return z["up2"], z["up4"], z["up8"]


def make_extra_channel(width, height, seed=0, i16=True, bit_depth=8, float_sample=False, exp_bits=0, upsampling_log2=0):
    """An extra channel as the Modular decode leaves it (integer samples of its own bit depth) and the descriptor of
    jxlgpu_frame_render_extra.  `width` x `height` is the channel's own size.  Returns (ExtraChannel, keep-alive list)."""
    rng = np.random.default_rng(SEED_BASE + 977 * (seed + 1))
    dt = np.int16 if i16 else np.int32
    if float_sample:
        # valid bit patterns of the (1, exp_bits, mantissa) float: finite, exponent field not all-ones and not zero
        mant_bits = bit_depth - exp_bits - 1
        e = rng.integers(1, (1 << exp_bits) - 1, size=(height, width))
        m = rng.integers(0, 1 << mant_bits, size=(height, width))
        sgn = rng.integers(0, 2, size=(height, width))
        pat = (sgn << (bit_depth - 1)) | (e << mant_bits) | m
        data = pat.astype(np.uint32).view(np.int32).astype(np.int64)
        if i16:
            data = pat.astype(np.uint16).view(np.int16)
        data = np.ascontiguousarray(data.astype(dt))
    else:
        # a soft matte plus samples slightly outside [0, max] (the Modular decode does not clamp)
        yy, xx = np.mgrid[0:height, 0:width]
        v = (np.sin(xx / 7.0) * np.cos(yy / 5.0) * 0.5 + 0.5) * ((1 << bit_depth) - 1)
        v = v + rng.integers(-3, 4, size=(height, width))
        lim = 32767 if i16 else (1 << 31) - 1
        data = np.ascontiguousarray(np.clip(np.rint(v), -lim - 1, lim).astype(dt))
    ec = abi.ExtraChannel()
    ec.data = data.ctypes.data
    ec.width, ec.height = width, height
    ec.sample_type = abi.SAMPLE_I16 if i16 else abi.SAMPLE_I32
    ec.bit_depth, ec.float_sample, ec.exp_bits = bit_depth, 1 if float_sample else 0, exp_bits
    ec.upsampling_log2 = upsampling_log2
    up = _load_up_weights()
    ec.weights.up2_weight = up[0].ctypes.data_as(abi.f32p)
    ec.weights.up4_weight = up[1].ctypes.data_as(abi.f32p)
    ec.weights.up8_weight = up[2].ctypes.data_as(abi.f32p)
    return ec, [data, up]


# jxl-image/src/color.rs:613-627 OpsinInverseMatrix defaults
OPSIN_INV = np.array([
    11.031566901960783, -9.866943921568629, -0.16462299647058826,
    -3.254147380392157, 4.418770392156863, -0.16462299647058826,
    -3.6588512862745097, 2.7129230470588235, 1.9459282392156863], dtype=np.float32)
OPSIN_BIAS = np.float32(-0.0037930732552754493)
QUANT_BIAS = np.array([1.0 - 0.05465007330715401, 1.0 - 0.07005449891748593,
                       1.0 - 0.049935103337343655], dtype=np.float32)
QUANT_BIAS_NUMERATOR = np.float32(0.145)

# cfg-2 type mix
_MIX = [
    (0.60, [abi.TRANSFORM_NAMES.index("Dct8")]),
    (0.20, [abi.TRANSFORM_NAMES.index(n) for n in ("Dct16", "Dct8x16", "Dct16x8")]),
    (0.08, [abi.TRANSFORM_NAMES.index(n) for n in ("Dct32", "Dct32x8", "Dct8x32", "Dct32x16", "Dct16x32")]),
    (0.10, [abi.TRANSFORM_NAMES.index(n) for n in ("Dct4x8", "Dct8x4", "Dct4", "Afv0", "Afv1", "Afv2", "Afv3", "Hornuss", "Dct2")]),
    (0.02, [abi.TRANSFORM_NAMES.index("Dct64")]),
]


def sec_half_large():
    """sec_half(n) for n = 64,128,256 as jxl-render/src/vardct/dct_common.rs:56-66 computes them
    (f32).  Generated once on the host and handed to both sides so the tables are identical."""
    out = []
    for n in (64, 128, 256):
        k = np.arange(n // 2, dtype=np.float32)
        theta = (np.float32(2) * k + np.float32(1)) / np.float32(2 * n) * np.float32(np.pi)
        out.append(((np.float32(1.0) / np.cos(theta, dtype=np.float32)) / np.float32(2.0)).astype(np.float32))
    return out


def draw_tiling(rng, w8, h8, mix=_MIX, types=None, group_cells=32):
    """Raster varblock placement, hf_metadata.rs:109-205.  Returns (kind u8[h8,w8], hf_mul i32)."""
    kind = np.full((h8, w8), abi.BLOCK_UNINIT, dtype=np.uint8)
    hf_mul = np.zeros((h8, w8), dtype=np.int32)
    if types is not None:
        cand = np.asarray(types)
        draws = cand[rng.integers(0, len(cand), size=(h8, w8))]
    else:
        probs = np.array([m[0] for m in mix])
        cls = rng.choice(len(mix), size=(h8, w8), p=probs / probs.sum())
        sub = rng.integers(0, 1 << 16, size=(h8, w8))
        draws = np.zeros((h8, w8), dtype=np.int64)
        for i, (_, lst) in enumerate(mix):
            lst = np.asarray(lst)
            sel = cls == i
            draws[sel] = lst[sub[sel] % len(lst)]
    muls = rng.integers(1, 17, size=(h8, w8)).astype(np.int32)
    occ = np.zeros((h8, w8), dtype=bool)
    for y in range(h8):
        x = 0
        row_occ = occ[y]
        while x < w8:
            if row_occ[x]:
                x += 1
                continue
            t = int(draws[y, x])
            bw, bh = DCT_SELECT_SIZE[t]
            if (bw > 1 or bh > 1) and (
                (x % group_cells) + bw > group_cells or (y % group_cells) + bh > group_cells
                or x + bw > w8 or y + bh > h8 or occ[y:y + bh, x:x + bw].any()):
                t = 0  # falls back to Dct8 when the drawn type does not fit
                bw = bh = 1
            occ[y:y + bh, x:x + bw] = True
            kind[y:y + bh, x:x + bw] = abi.BLOCK_OCCUPIED
            kind[y, x] = t
            hf_mul[y, x] = muls[y, x]
            x += bw
    return kind, hf_mul


SRGB_LUMINANCES = [0.2126, 0.7152, 0.0722]
# linear sRGB -> Display P3 (D65), the merged `primaries_to_xyz * xyz_to_primaries` Matrix op
SRGB_TO_P3 = [0.8224621, 0.1775380, 0.0000000,
              0.0331941, 0.9668058, 0.0000000,
              0.0170827, 0.0723974, 0.9105199]


def make_noise_params(seed):
    """LfGlobal.noise as a photon-noise encode would write it: 8 strengths in 1/1024 steps
    (jxl-frame/src/data/noise.rs:9-15), frame counters for rng_seed0."""
    rng = np.random.default_rng(SEED_BASE + 7919 * (seed + 1))
    p = abi.NoiseParams()
    p.enabled = 1
    p.lut[:] = [float(v) / 1024.0 for v in rng.integers(20, 400, 8)]
    p.visible_frames = int(rng.integers(0, 5))
    p.invisible_frames = int(rng.integers(0, 3))
    return p


def configure_color(cp, mode):
    """Op lists `ColorTransform::new` builds for a few XYB -> target encodings besides the sRGB
    and PQ ones (jxl-color/src/convert.rs:208-549).  `cp` already holds XybToMixedLms + Matrix."""
    if mode == "tone_map_srgb":
        # HDR image (intensity_target > 255) shown on an SDR sRGB target, perceptual intent:
        # ToneMapRec2408{target 255} -> GamutMap{0.3} -> sRGB  (convert.rs:478-500)
        assert cp.intensity_target > 255.0
        cp.tone_map = 1
        cp.tm_luminances[:] = SRGB_LUMINANCES
        cp.tm_min_nits = 0.0
        cp.tm_target_display_luminance = 255.0
        cp.tm_gamut_map = 1
        cp.tm_gamut_saturation_factor = 0.3
        cp.transfer_function = abi.TF_SRGB
    elif mode == "tone_map_min_nits":
        # same, relative intent (no GamutMap after the tone map), non-zero black level, linear out
        assert cp.intensity_target > 255.0
        cp.tone_map = 1
        cp.tm_luminances[:] = SRGB_LUMINANCES
        cp.tm_min_nits = 0.05
        cp.tm_target_display_luminance = 255.0
        cp.tm_gamut_map = 0
        cp.transfer_function = abi.TF_LINEAR
    elif mode == "bt709":
        cp.transfer_function = abi.TF_BT709
    elif mode == "clip_p3_dci":
        # relative intent to Display P3 primaries with the DCI gamma: Clip -> Matrix -> Gamma(1/2.6)
        cp.gamut_map = abi.GAMUT_CLIP
        cp.has_matrix2 = 1
        cp.matrix2[:] = SRGB_TO_P3
        cp.transfer_function = abi.TF_GAMMA
        cp.gamma = float(np.float32(1.0) / np.float32(2.6))
    elif mode == "gamma22":
        cp.transfer_function = abi.TF_GAMMA
        cp.gamma = float(np.float32(1e7) / np.float32(22000000))  # Gamma{g: 22000000, inverted: false}
    elif mode == "hlg":
        # XYB image shown on an HLG target: TransferFunction{Hlg} = inverse OOTF with the image's intensity target, then
        # linear_to_hlg (convert.rs:1021-1032; HLG is an HDR encoding, so no tone map: :478)
        cp.transfer_function = abi.TF_HLG
        cp.hlg_luminances[:] = SRGB_LUMINANCES
        cp.hlg_ootf_intensity_target = cp.intensity_target
    elif mode == "pq_to_hlg":
        # PQ image, HLG target, perceptual intent (`from_pq`, convert.rs:501-536): ToneMapRec2408{target 1000} ->
        # HlgInverseOotf{1000} -> GamutMap{0.1} -> linear_to_hlg (the transfer function's own OOTF skipped: 300)
        assert not 999.0 <= cp.intensity_target <= 1001.0
        cp.tone_map = 1
        cp.tm_luminances[:] = SRGB_LUMINANCES
        cp.tm_min_nits = 0.0
        cp.tm_target_display_luminance = 1000.0
        cp.hlg_luminances[:] = SRGB_LUMINANCES
        cp.hlg_ootf_intensity_target = 1000.0
        cp.tm_gamut_map = 1
        cp.tm_gamut_saturation_factor = 0.1
        cp.transfer_function = abi.TF_HLG
    elif mode == "pq_to_hlg_1000":
        # the same for a 1000-nit image: no tone map, no inverse OOTF, only GamutMap{0.1} and linear_to_hlg
        assert 999.0 <= cp.intensity_target <= 1001.0
        cp.tm_luminances[:] = SRGB_LUMINANCES
        cp.tm_gamut_map = 1
        cp.tm_gamut_saturation_factor = 0.1
        cp.transfer_function = abi.TF_HLG
    else:
        raise ValueError(mode)


class VardctWorkload:
    """Holds numpy arrays (kept alive) + builds the C descriptor."""

    def __init__(self, width, height, seed=0, epf_iters=2, gabor=True, tf=abi.TF_SRGB,
                 intensity_target=255.0, upsampling=1, types=None, lf_i16=True,
                 zero_fraction=0.85, skip_lf_smoothing=False, hdr_pq=False, group_dim=256,
                 color_mode=None, noise=False, lf_frame=False, nz_fraction=None):
        """nz_fraction (None: the historical generator, ~4.7 % of the coefficients end up non-zero): the fraction of
        ALL coefficients of the frame that are non-zero after quantisation — the largest-magnitude draws survive, so
        the non-zeros sit at the low frequencies like a real stream's.  SURVEY §8(d) specifies 0.15 for the headline."""
        rng = np.random.default_rng(SEED_BASE + seed)
        self.width, self.height = width, height
        self.group_dim = group_dim
        w8, h8 = -(-width // 8), -(-height // 8)
        self.w8, self.h8 = w8, h8
        wr, hr = w8 * 8, h8 * 8
        self.wr, self.hr = wr, hr
        self.global_scale = int(rng.integers(3000, 6001))
        self.quant_lf = 16
        self.mats = default_dequant_matrices()
        self.sec = sec_half_large()

        kind, hf_mul = draw_tiling(rng, w8, h8, types=types)
        self.kind, self.hf_mul = kind, hf_mul

        # ---- HF coefficients: Laplacian, scale ~ 1/(1+freq), ~85 % zeros, LLF corner zero
        coeff = np.zeros((3, hr, wr), dtype=np.int32)
        ys, xs = np.nonzero(kind <= 26)
        tt = kind[ys, xs]
        for t in np.unique(tt):
            bw, bh = DCT_SELECT_SIZE[int(t)]
            W, H = bw * 8, bh * 8
            sel = tt == t
            n = int(sel.sum())
            fy = np.arange(H, dtype=np.float32)[:, None] / bh
            fx = np.arange(W, dtype=np.float32)[None, :] / bw
            # target dequantised amplitude per channel (X, Y, B), decaying ~ 1/(1+freq); the
            # quantised value is that amplitude divided by the dequant step of its position
            amp = np.array([0.002, 0.02, 0.015], dtype=np.float32)[:, None, None] / (1.0 + np.sqrt(fx * fx + fy * fy))
            step = np.stack([self.mats[int(t)][c] for c in range(3)]) * np.float32(65536.0 / self.global_scale)
            scale = amp / step
            blk_mul = hf_mul[ys[sel], xs[sel]].astype(np.float32)[:, None, None, None]
            vals = rng.laplace(0.0, 1.0, size=(n, 3, H, W)).astype(np.float32) * scale * blk_mul
            if nz_fraction is None:
                keep = rng.random(size=(n, 3, H, W)) >= zero_fraction
                q = np.where(keep, np.rint(vals), 0).astype(np.int32)
            else:
                # gain such that exactly the top `nz_fraction` of the magnitudes (LLF corner excluded) round away from zero
                vals[:, :, :bh, :bw] = 0
                mag = np.abs(vals).ravel()
                k = min(mag.size - 1, max(0, int(round(nz_fraction * mag.size))))
                tau = np.partition(mag, mag.size - 1 - k)[mag.size - 1 - k] if k > 0 else np.inf
                gain = np.float32(0.5) / np.float32(tau) if np.isfinite(tau) and tau > 0 else np.float32(0)
                q = np.clip(np.rint(vals * gain), -32767, 32767).astype(np.int32)
            q[:, :, :bh, :bw] = 0  # LLF positions are not coded in HF (hf_coeff.rs)
            for i, (cy, cx) in enumerate(zip(ys[sel], xs[sel])):
                coeff[:, cy * 8:cy * 8 + H, cx * 8:cx * 8 + W] = q[i]
        self.coeff = coeff

        # ---- LF quant: blurred random image, channel order of the reference: Y, X, B
        def smooth(amp, offset):
            g = rng.normal(size=(h8 // 8 + 3, w8 // 8 + 3))
            g = np.kron(g, np.ones((8, 8)))[:h8 + 8, :w8 + 8]
            k = np.ones(9) / 9.0
            g = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, g)
            g = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, g)
            g = g[4:4 + h8, 4:4 + w8]
            # ~0.6 quantisation steps of noise: small enough that adaptive LF smoothing engages
            return np.rint(offset + amp * g + 0.6 * rng.normal(size=(h8, w8)))
        lf_dtype = np.int16 if lf_i16 else np.int32
        self.lf_sample_type = abi.SAMPLE_I16 if lf_i16 else abi.SAMPLE_I32
        self.lfq = [smooth(180.0, 220.0).astype(lf_dtype),   # Y
                    smooth(60.0, 0.0).astype(lf_dtype),      # X
                    smooth(90.0, 120.0).astype(lf_dtype)]    # B

        # ---- LF frame (frame_header.flags.use_lf_frame): the LF image arrives as f32 XYB samples of an
        # earlier frame's render — drawn after everything else so that the other planes do not depend on the switch
        self.lf_frame = None
        if lf_frame:
            r2 = np.random.default_rng(SEED_BASE + seed + 7717)
            self.lf_stride = w8 + 5   # rows of the LF frame's buffer are wider than the image
            base = np.stack([0.01 * r2.normal(size=(h8, self.lf_stride)), 0.3 + 0.2 * r2.random(size=(h8, self.lf_stride)),
                             0.3 + 0.2 * r2.random(size=(h8, self.lf_stride))])
            self.lf_frame = np.ascontiguousarray(base.astype(np.float32))

        w64, h64 = -(-width // 64), -(-height // 64)
        self.xfy = rng.integers(-16, 17, size=(h64, w64)).astype(np.int32)
        self.bfy = rng.integers(-16, 17, size=(h64, w64)).astype(np.int32)

        # ---- EPF sigma, hf_metadata.rs:110-115,160-161,200-210
        self.epf_iters = epf_iters
        sharp_lut = (np.arange(8, dtype=np.float32) / np.float32(7.0)).astype(np.float32)
        sharp_lut[7] = 1.0
        sharpness = rng.integers(0, 8, size=(h8, w8))
        quant_mul = np.float32(0.46) * np.float32(65536.0) / np.float32(self.global_scale)
        owner_mul = np.zeros((h8, w8), dtype=np.float32)
        for (cy, cx) in zip(ys, xs):
            bw, bh = DCT_SELECT_SIZE[int(kind[cy, cx])]
            owner_mul[cy:cy + bh, cx:cx + bw] = hf_mul[cy, cx]
        self.sigma = ((quant_mul / owner_mul).astype(np.float32) * sharp_lut[sharpness]).astype(np.float32)

        self.filter = abi.FilterParams()
        self.filter.gab_enabled = 1 if gabor else 0
        for c in range(3):
            self.filter.gab_weights[c][0] = 0.115169525
            self.filter.gab_weights[c][1] = 0.061248592
        self.filter.epf_iters = epf_iters
        self.filter.epf_channel_scale[:] = [40.0, 5.0, 3.5]
        self.filter.epf_pass0_sigma_scale = 0.9
        self.filter.epf_pass2_sigma_scale = 6.5
        self.filter.epf_border_sad_mul = 2.0 / 3.0
        self.filter.epf_sigma_for_modular = 1.0

        self.color = abi.ColorParams()
        self.color.enabled = 1
        self.color.opsin_bias[:] = [OPSIN_BIAS] * 3
        self.color.intensity_target = intensity_target
        self.color.matrix[:] = list(OPSIN_INV)
        self.color.transfer_function = tf
        if hdr_pq:
            # Appendix C of SURVEY.md: XYB -> linear sRGB -> GamutMap -> Rec.2020 -> PQ
            self.color.gamut_map = 1
            self.color.gamut_luminances[:] = [0.2126, 0.7152, 0.0722]
            self.color.gamut_saturation_factor = 0.3
            self.color.has_matrix2 = 1
            self.color.matrix2[:] = [0.6274039, 0.3292830, 0.0433131,
                                     0.0690973, 0.9195404, 0.0113623,
                                     0.0163914, 0.0880133, 0.8955953]
            self.color.transfer_function = abi.TF_PQ
        if color_mode is not None:
            configure_color(self.color, color_mode)

        self.noise = make_noise_params(seed) if noise else abi.NoiseParams()
        self.up_factor = upsampling
        self.up_w = None
        if upsampling > 1:
            self.up_w = _load_up_weights()

        self.x_qm_scale = 3
        self.b_qm_scale = 2
        self.skip_lf_smoothing = skip_lf_smoothing
        self._keep = []

    # ---- descriptor
    def desc(self, coeff_transport="dense_i32", sparse_split=False, partial=None, pass_shifts=None):
        """`coeff_transport`: "dense_i32" (the reference's framebuffer), "dense_i16", "sparse_i32",
        "sparse_i16" (SURVEY §8f rank 2).  `sparse_split` splits every value over two list entries
        at the same position, as two passes of a progressive frame would (hf_coeff.rs:234 `+=`)."""
        d = abi.VardctDesc()
        d.abi = abi.ABI_VERSION
        d.width, d.height = self.width, self.height
        d.group_dim = self.group_dim
        d.lf_sample_type = self.lf_sample_type
        keep_c = []
        if self.lf_frame is not None:
            for c in range(3):
                d.lf_frame[c] = self.lf_frame[c].ctypes.data_as(abi.f32p)
            d.lf_frame_stride = self.lf_stride
        coeff_planes = self.coeff
        if partial is not None and not pass_shifts:
            # a truncated stream (allow_partial): `partial` = {group index: varblocks decoded before the stream ended}.
            # Dense transports hold what the decoder wrote: zeros for the varblocks it never reached.
            d.allow_partial = 1
            coeff_planes = self.truncated_coeff(partial)
            keep_c.append(coeff_planes)
        v16 = coeff_transport.endswith("i16")
        d.coeff_sample_type = abi.SAMPLE_I16 if v16 else abi.SAMPLE_I32
        d.coeff_stride = self.wr
        if coeff_transport.startswith("dense"):
            d.coeff_format = abi.COEFF_DENSE
            for c in range(3):
                plane = coeff_planes[c]
                if v16:
                    assert np.abs(plane).max() < 32768
                    plane = np.ascontiguousarray(plane.astype(np.int16))
                    keep_c.append(plane)
                d.coeff[c] = plane.ctypes.data
        elif coeff_transport == "grouped":
            d.coeff_format = abi.COEFF_GROUPED
            if pass_shifts:
                # a progressive frame: pass p carries the part of every coefficient above bit pass_shifts[p] that the
                # earlier passes have not sent (`unpack_signed(ucoeff) << coeff_shift`, hf_coeff.rs:235); the parts sum
                # to the coefficient
                # `partial` (a truncated progressive stream): {(pass, group): varblocks of that pass group decoded before its
                # section ended}; what the decoder leaves is the sum of the truncated parts (progressive_truncated_coeff)
                assert pass_shifts[-1] == 0
                if partial is not None:
                    d.allow_partial = 1
                rest = self.coeff.astype(np.int64)
                per_pass = []
                for pi, sft in enumerate(pass_shifts):
                    part = (rest >> sft) << sft
                    rest = rest - part
                    ppart = None if partial is None else {g: k for (p, g), k in partial.items() if p == pi}
                    per_pass.append(self.grouped_lists(ppart or None, coeff=part.astype(np.int32)))
                n = len(per_pass[0][0])
                hf_groups = (abi.HfGroup * (n * len(per_pass)))()
                for pi, (hf, arrays) in enumerate(per_pass):
                    for g in range(n):
                        hf_groups[pi * n + g] = hf[g]
                keep_c += [hf_groups, per_pass]
                d.num_passes = len(per_pass)
                d.num_hf_groups = n
            else:
                hf_groups, arrays = self.grouped_lists(partial)
                keep_c += [hf_groups, arrays]
                d.num_hf_groups = len(hf_groups)
            d.hf_groups = C.cast(hf_groups, C.POINTER(abi.HfGroup))
        else:
            d.coeff_format = abi.COEFF_SPARSE
            for c in range(3):
                flat = coeff_planes[c].reshape(-1)
                pos = np.flatnonzero(flat).astype(np.uint32)
                val = flat[pos]
                if sparse_split:  # second "pass" refines the same positions; order shuffled
                    first = val // 2
                    pos = np.concatenate([pos, pos])
                    val = np.concatenate([first, val - first])
                    perm = np.random.default_rng(c).permutation(pos.size)
                    pos, val = pos[perm], val[perm]
                val = np.ascontiguousarray(val.astype(np.int16 if v16 else np.int32))
                pos = np.ascontiguousarray(pos)
                keep_c += [pos, val]
                d.coeff[c] = val.ctypes.data if val.size else None
                d.sparse_pos[c] = pos.ctypes.data_as(C.POINTER(C.c_uint32)) if pos.size else None
                d.sparse_count[c] = pos.size
        lf_dim = self.group_dim * 8
        gx_n, gy_n = -(-self.width // lf_dim), -(-self.height // lf_dim)
        groups = (abi.LfGroup * (gx_n * gy_n))()
        cells = self.group_dim
        tiles = lf_dim // 64
        keep = []
        for gy in range(gy_n):
            for gx in range(gx_n):
                g = groups[gy * gx_n + gx]
                wpx = min(lf_dim, self.width - gx * lf_dim)
                hpx = min(lf_dim, self.height - gy * lf_dim)
                g.width_px, g.height_px = wpx, hpx
                bw, bh = -(-wpx // 8), -(-hpx // 8)
                cw, ch = -(-wpx // 64), -(-hpx // 64)
                sl = (slice(gy * cells, gy * cells + bh), slice(gx * cells, gx * cells + bw))
                tl = (slice(gy * tiles, gy * tiles + ch), slice(gx * tiles, gx * tiles + cw))
                arrs = dict(
                    lfq=[np.ascontiguousarray(p[sl]) for p in self.lfq],
                    kind=np.ascontiguousarray(self.kind[sl]),
                    mul=np.ascontiguousarray(self.hf_mul[sl]),
                    sigma=np.ascontiguousarray(self.sigma[sl]),
                    xfy=np.ascontiguousarray(self.xfy[tl]),
                    bfy=np.ascontiguousarray(self.bfy[tl]),
                )
                keep.append(arrs)
                for c in range(3):
                    g.lf_quant[c] = arrs["lfq"][c].ctypes.data
                g.extra_precision = (gx + gy + getattr(self, 'lf_group_row0', 0)) % 2  # exercise per-LF-group precision
                g.has_hf_meta = 1
                g.block_kind = arrs["kind"].ctypes.data_as(abi.u8p)
                g.hf_mul = arrs["mul"].ctypes.data_as(abi.i32p)
                g.epf_sigma = arrs["sigma"].ctypes.data_as(abi.f32p)
                g.x_from_y = arrs["xfy"].ctypes.data_as(abi.i32p)
                g.b_from_y = arrs["bfy"].ctypes.data_as(abi.i32p)
        d.num_lf_groups = gx_n * gy_n
        d.lf_groups = C.cast(groups, C.POINTER(abi.LfGroup))
        d.global_scale, d.quant_lf = self.global_scale, self.quant_lf
        d.m_lf[:] = [1.0 / 32.0, 1.0 / 4.0, 1.0 / 2.0]
        d.colour_factor = 84
        d.base_correlation_x, d.base_correlation_b = 0.0, 1.0
        d.x_factor_lf, d.b_factor_lf = 126, 131
        d.x_qm_scale, d.b_qm_scale = self.x_qm_scale, self.b_qm_scale
        d.quant_bias[:] = list(QUANT_BIAS)
        d.quant_bias_numerator = QUANT_BIAS_NUMERATOR
        d.skip_adaptive_lf_smoothing = 1 if self.skip_lf_smoothing else 0
        for t in range(abi.NUM_TRANSFORMS):
            for c in range(3):
                d.dequant[t][c] = self.mats[t][c].ctypes.data_as(abi.f32p)
        for i in range(3):
            d.sec_half_large[i] = self.sec[i].ctypes.data_as(abi.f32p)
        d.filter = self.filter
        d.color = self.color
        d.noise = self.noise
        d.upsampling.factor = self.up_factor
        if self.up_w is not None:
            d.upsampling.up2_weight = self.up_w[0].ctypes.data_as(abi.f32p)
            d.upsampling.up4_weight = self.up_w[1].ctypes.data_as(abi.f32p)
            d.upsampling.up8_weight = self.up_w[2].ctypes.data_as(abi.f32p)
        self._keep = [groups, keep, keep_c]
        return d

    def _decode_order(self):
        gc = self.group_dim // 8
        groups_x = -(-self.width // self.group_dim)
        ys, xs = np.nonzero(self.kind <= 26)
        gid = (ys // gc) * groups_x + xs // gc
        order = np.lexsort((xs, ys, gid))          # group, then raster inside the group
        return ys[order], xs[order], gid[order]

    def progressive_truncated_coeff(self, pass_shifts, partial):
        """The dense planes a decoder leaves for a multi-pass frame whose pass group (p, g) ended after partial[(p, g)]
        varblocks: the sum over the passes of each pass's part, truncated per (pass, group) (hf_coeff.rs:207-244)."""
        rest = self.coeff.astype(np.int64)
        total = np.zeros_like(rest)
        for pi, sft in enumerate(pass_shifts):
            part = (rest >> sft) << sft
            rest = rest - part
            total += self.truncated_coeff({g: k for (p, g), k in partial.items() if p == pi}, coeff=part)
        return total.astype(np.int32)

    def truncated_coeff(self, partial, coeff=None):
        """The dense planes a decoder leaves when group g's stream ends after partial[g] varblocks."""
        ys, xs, gid = self._decode_order()
        out = (self.coeff if coeff is None else coeff).copy()
        for g, keep in partial.items():
            idx = np.flatnonzero(gid == g)[keep:]
            for i in idx:
                bw, bh = DCT_SELECT_SIZE[int(self.kind[ys[i], xs[i]])]
                out[:, ys[i] * 8:(ys[i] + bh) * 8, xs[i] * 8:(xs[i] + bw) * 8] = 0
        return out

    def grouped_lists(self, partial=None, coeff=None):
        """JXLGPU_COEFF_GROUPED: per 256x256 pass group, the `non_zeros` counts and the
        (dx, dy, coeff) triples in the order `write_hf_coeff` decodes them (hf_coeff.rs:97-254):
        Data cells of the group in raster order, channels Y, X, B.  The order of the triples inside
        one (varblock, channel) — the coefficient order of the pass in a real stream — is shuffled."""
        gc = self.group_dim // 8
        h8, w8 = self.kind.shape
        groups_x, groups_y = -(-self.width // self.group_dim), -(-self.height // self.group_dim)
        ys, xs = np.nonzero(self.kind <= 26)
        gid = (ys // gc) * groups_x + xs // gc
        order = np.lexsort((xs, ys, gid))          # group, then raster inside the group
        ys, xs, gid = ys[order], xs[order], gid[order]
        nvb = ys.size
        owner = np.full((h8, w8), -1, dtype=np.int64)  # cell -> index of its varblock in decode order
        for k in np.unique(self.kind[ys, xs]):
            bw, bh = DCT_SELECT_SIZE[int(k)]
            sel = np.flatnonzero(self.kind[ys, xs] == k)
            for dy in range(bh):
                for dx in range(bw):
                    owner[ys[sel] + dy, xs[sel] + dx] = sel
        counts = np.zeros((nvb, 3), dtype=np.int64)
        ent_vb, ent_slot, ent_word = [], [], []
        for slot, c in enumerate((1, 0, 2)):       # decoded Y, X, B (hf_coeff.rs:138-140)
            src = self.coeff if coeff is None else coeff
            py, px = np.nonzero(src[c])
            vb = owner[py // 8, px // 8]
            assert (vb >= 0).all()
            val = src[c][py, px]
            assert np.abs(val).max(initial=0) < 32768
            dx, dy = px - xs[vb] * 8, py - ys[vb] * 8
            counts[:, slot] = np.bincount(vb, minlength=nvb)
            ent_vb.append(vb)
            ent_slot.append(np.full(vb.size, slot))
            ent_word.append((dx | (dy << 8) | ((val.astype(np.int64) & 0xFFFF) << 16)).astype(np.uint32))
        ent_vb, ent_slot, ent_word = map(np.concatenate, (ent_vb, ent_slot, ent_word))
        shuffle = np.random.default_rng(SEED_BASE ^ 0x6E7A).random(ent_vb.size)
        o = np.lexsort((shuffle, ent_slot, ent_vb))
        words = np.ascontiguousarray(ent_word[o])
        counts16 = np.ascontiguousarray(counts.astype(np.uint16).reshape(-1))
        n_groups = groups_x * groups_y
        vb_first = np.searchsorted(gid, np.arange(n_groups + 1))      # varblock range of every group
        nz_first = np.concatenate([[0], np.cumsum(counts.sum(axis=1))])[vb_first]
        hf = (abi.HfGroup * n_groups)()
        u16p, u32p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint32)
        nz_cum = np.concatenate([[0], np.cumsum(counts.sum(axis=1))])
        for g in range(n_groups):
            hf[g].num_varblocks = int(vb_first[g + 1] - vb_first[g])
            hf[g].num_nz = int(nz_first[g + 1] - nz_first[g])
            if partial and g in partial:   # the group's stream ended after this many varblocks
                hf[g].num_varblocks = min(hf[g].num_varblocks, int(partial[g]))
                hf[g].num_nz = int(nz_cum[vb_first[g] + hf[g].num_varblocks] - nz_first[g])
            hf[g].nz_count = C.cast(counts16.ctypes.data + 6 * int(vb_first[g]), u16p)
            hf[g].nz = C.cast(words.ctypes.data + 4 * int(nz_first[g]), u32p)
        return hf, (counts16, words)

    def out_size(self, stages):
        f = self.up_factor if (stages & abi.STAGE_UPSAMPLE) else 1
        return self.width * f, self.height * f


JPEG_MODES = {
    # frame_header.jpeg_upsampling per framebuffer channel (Cb, Y, Cr); ChannelShift::
    # from_jpeg_upsampling (jxl-modular/src/param.rs:105-122): 0 = "as subsampled as the frame
    # gets", 1 = full resolution, 2 = vertical subsampling only, 3 = horizontal only
    "444": [0, 0, 0],
    "420": [0, 1, 0],
    "422": [0, 2, 0],   # chroma halved horizontally
    "440": [0, 3, 0],   # chroma halved vertically
    "mixed": [0, 1, 2], # Cb 2x2, Y full, Cr vertical only: three geometries
}


def jpeg_shifts(ju):
    hscale = any(v in (1, 2) for v in ju)
    vscale = any(v in (1, 3) for v in ju)
    out = []
    for v in ju:
        h, w = {0: (hscale, vscale), 1: (False, False), 2: (False, vscale), 3: (hscale, False)}[v]
        out.append((int(h), int(w)))
    return out, hscale, vscale


def _ssize(n, has, sub):
    if not has:
        return n
    s = -(-n // 2)
    return s if sub else s * 2


class JpegWorkload:
    """A JPEG-transcode style VarDCT frame at the device boundary: DCT8 everywhere, YCbCr planes
    (Cb, Y, Cr) with optional chroma subsampling, no chroma-from-luma use, LF smoothing skipped
    (SURVEY §8f rank 4).  Coefficient magnitudes follow the same scaling rule as VardctWorkload."""

    def __init__(self, width, height, mode="420", seed=0, epf_iters=0, gabor=False, lf_i16=True):
        rng = np.random.default_rng(SEED_BASE + 0x4A00 + seed)
        self.width, self.height, self.mode = width, height, mode
        self.ju = JPEG_MODES[mode]
        self.shifts, self.has_h, self.has_v = jpeg_shifts(self.ju)
        self.group_dim = 256
        W8, H8 = -(-width // 8), -(-height // 8)
        self.W8r = -(-W8 // 2) * 2 if self.has_h else W8
        self.H8r = -(-H8 // 2) * 2 if self.has_v else H8
        self.cw = [_ssize(W8, self.has_h, h) for (h, v) in self.shifts]
        self.ch = [_ssize(H8, self.has_v, v) for (h, v) in self.shifts]
        self.global_scale = int(rng.integers(3000, 6001))
        self.quant_lf = 16
        self.mats = default_dequant_matrices()
        self.sec = sec_half_large()
        self.lf_sample_type = abi.SAMPLE_I16 if lf_i16 else abi.SAMPLE_I32
        lf_dt = np.int16 if lf_i16 else np.int32
        self.hf_mul = rng.integers(2, 9, size=(self.H8r, self.W8r)).astype(np.int32)
        self.sigma = rng.uniform(0.2, 2.5, size=(self.H8r, self.W8r)).astype(np.float32)
        self.kind = np.zeros((self.H8r, self.W8r), dtype=np.uint8)  # JXLGPU_DCT8
        m_lf = [1.0 / 32.0, 1.0 / 4.0, 1.0 / 2.0]
        self.lfq, self.coeff = [], []
        for c in range(3):
            cw, ch = self.cw[c], self.ch[c]
            yy, xx = np.mgrid[0:ch, 0:cw]
            step = m_lf[c] * 512.0 / (self.global_scale * self.quant_lf)
            target = (0.2 if c == 1 else 0.08) * np.sin(xx / 5.0 + c) * np.cos(yy / 7.0)  # centred YCbCr
            self.lfq.append(np.rint(target / step + rng.normal(0, 0.6, size=(ch, cw))).astype(lf_dt))
            # HF: Laplacian quantised coefficients, ~85 % zeros, DC position empty (hf_coeff.rs)
            pos = np.add.outer(np.arange(8), np.arange(8))
            amp = 6.0 / (1.0 + pos)
            vals = rng.laplace(0.0, 1.0, size=(ch, cw, 8, 8)) * amp
            keep = rng.random(size=(ch, cw, 8, 8)) >= 0.85
            q = np.where(keep, np.rint(vals), 0).astype(np.int32)
            q[:, :, 0, 0] = 0
            self.coeff.append(np.ascontiguousarray(q.transpose(0, 2, 1, 3).reshape(ch * 8, cw * 8)))
        self.filter = abi.FilterParams()
        self.filter.gab_enabled = 1 if gabor else 0
        for c in range(3):
            self.filter.gab_weights[c][0] = 0.115169525
            self.filter.gab_weights[c][1] = 0.061248592
        self.filter.epf_iters = epf_iters
        self.filter.epf_channel_scale[:] = [40.0, 5.0, 3.5]
        self.filter.epf_pass0_sigma_scale = 0.9
        self.filter.epf_pass2_sigma_scale = 6.5
        self.filter.epf_border_sad_mul = 2.0 / 3.0
        self.filter.epf_sigma_for_modular = 1.0
        self.color = abi.ColorParams()
        self.color.ycbcr = 1
        self._keep = []

    def desc(self):
        d = abi.VardctDesc()
        d.abi = abi.ABI_VERSION
        d.width, d.height, d.group_dim = self.width, self.height, self.group_dim
        d.lf_sample_type = self.lf_sample_type
        d.jpeg_upsampling[:] = self.ju
        d.coeff_format, d.coeff_sample_type = abi.COEFF_DENSE, abi.SAMPLE_I32
        d.coeff_stride = self.W8r * 8
        for c in range(3):
            assert self.coeff[c].shape[1] == (d.coeff_stride >> self.shifts[c][0])
            d.coeff[c] = self.coeff[c].ctypes.data
        lf_dim = self.group_dim * 8
        gx_n, gy_n = -(-self.width // lf_dim), -(-self.height // lf_dim)
        groups = (abi.LfGroup * (gx_n * gy_n))()
        keep = []
        SRC = [1, 0, 2]  # lf_quant channel k holds framebuffer channel SRC^-1: [0]=Y, [1]=X/Cb, [2]=B/Cr
        for gy in range(gy_n):
            for gx in range(gx_n):
                g = groups[gy * gx_n + gx]
                wpx = min(lf_dim, self.width - gx * lf_dim)
                hpx = min(lf_dim, self.height - gy * lf_dim)
                g.width_px, g.height_px = wpx, hpx
                gbw, gbh = -(-wpx // 8), -(-hpx // 8)
                bw = -(-gbw // 2) * 2 if self.has_h else gbw
                bh = -(-gbh // 2) * 2 if self.has_v else gbh
                cwt, cht = -(-wpx // 64), -(-hpx // 64)
                cells = self.group_dim
                sl = (slice(gy * cells, gy * cells + bh), slice(gx * cells, gx * cells + bw))
                arrs = dict(kind=np.ascontiguousarray(self.kind[sl]), mul=np.ascontiguousarray(self.hf_mul[sl]),
                            sigma=np.ascontiguousarray(self.sigma[sl]),
                            xfy=np.zeros((cht, cwt), np.int32), bfy=np.zeros((cht, cwt), np.int32), lfq=[None] * 3)
                assert arrs["kind"].shape == (bh, bw)
                for k in range(3):
                    c = SRC[k]
                    h, v = self.shifts[c]
                    lw, lh = _ssize(gbw, self.has_h, h), _ssize(gbh, self.has_v, v)
                    ox, oy = (gx * cells) >> h, (gy * cells) >> v
                    a = np.ascontiguousarray(self.lfq[c][oy:oy + lh, ox:ox + lw])
                    assert a.shape == (lh, lw)
                    arrs["lfq"][k] = a
                    g.lf_quant[k] = a.ctypes.data
                keep.append(arrs)
                g.extra_precision = (gx + gy) % 2
                g.has_hf_meta = 1
                g.block_kind = arrs["kind"].ctypes.data_as(abi.u8p)
                g.hf_mul = arrs["mul"].ctypes.data_as(abi.i32p)
                g.epf_sigma = arrs["sigma"].ctypes.data_as(abi.f32p)
                g.x_from_y = arrs["xfy"].ctypes.data_as(abi.i32p)
                g.b_from_y = arrs["bfy"].ctypes.data_as(abi.i32p)
        d.num_lf_groups = gx_n * gy_n
        d.lf_groups = C.cast(groups, C.POINTER(abi.LfGroup))
        d.global_scale, d.quant_lf = self.global_scale, self.quant_lf
        d.m_lf[:] = [1.0 / 32.0, 1.0 / 4.0, 1.0 / 2.0]
        d.colour_factor = 84
        d.base_correlation_x, d.base_correlation_b = 0.0, 1.0
        d.x_factor_lf, d.b_factor_lf = 128, 128
        d.x_qm_scale, d.b_qm_scale = 2, 2
        d.quant_bias[:] = list(QUANT_BIAS)
        d.quant_bias_numerator = QUANT_BIAS_NUMERATOR
        d.skip_adaptive_lf_smoothing = 1
        for t in range(abi.NUM_TRANSFORMS):
            for c in range(3):
                d.dequant[t][c] = self.mats[t][c].ctypes.data_as(abi.f32p)
        for i in range(3):
            d.sec_half_large[i] = self.sec[i].ctypes.data_as(abi.f32p)
        d.filter = self.filter
        d.color = self.color
        d.upsampling.factor = 1
        self._keep = [groups, keep]
        return d
