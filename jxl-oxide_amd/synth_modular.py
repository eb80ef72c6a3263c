"""Synthetic Modular boundary state: channel buffers as they look *after* the per-group entropy
decode and *before* `modular_image.prepare_subimage().finish(pool)` (jxl-render/src/modular.rs:134),
i.e. full-resolution buffers holding the squeezed pyramid as nested sub-rectangles
(jxl-modular/src/transform.rs:343-437).

The forward transforms here (forward Squeeze / RCT / Gradient residuals / palette indexing) are
written independently of oracle/modular.c, in numpy, from the definitions of the inverse — so
`inverse(forward(x)) == x` pins both the oracle and the HIP kernels bit-exactly.
"""
import ctypes as C

import numpy as np

from . import abi
from .synth import OPSIN_BIAS, OPSIN_INV, SEED_BASE


def _trunc_div(x, d):
    return np.sign(x) * (np.abs(x) // d)


def tendency(a, b, c):
    """jxl-modular/src/transform/squeeze.rs:1104-1137, int64 (no wrapping at test magnitudes)."""
    a = a.astype(np.int64); b = b.astype(np.int64); c = c.astype(np.int64)
    out = np.zeros_like(a)
    m1 = (a >= b) & (b >= c)
    x = _trunc_div(4 * a - 3 * c - b + 6, 12)
    x = np.where(x - (x & 1) > 2 * (a - b), 2 * (a - b) + 1, x)
    x = np.where(x + (x & 1) > 2 * (b - c), 2 * (b - c), x)
    out = np.where(m1, x, out)
    m2 = (a <= b) & (b <= c) & ~m1
    x = _trunc_div(4 * a - 3 * c - b - 6, 12)
    x = np.where(x + (x & 1) < 2 * (a - b), 2 * (a - b) - 1, x)
    x = np.where(x - (x & 1) < 2 * (b - c), 2 * (b - c), x)
    out = np.where(m2, x, out)
    return out


def forward_squeeze_rows(v):
    """v: (..., n) int64 along the last axis -> (avg (..., ceil(n/2)), residual (..., n//2)).
    Derived from the inverse (squeeze.rs:59-88): first = avg + trunc(diff/2), second = first - diff,
    diff = residual + tendency(left, avg, next_avg), left = previous `second`."""
    n = v.shape[-1]
    pairs = n // 2
    A = v[..., 0:2 * pairs:2]
    B = v[..., 1:2 * pairs:2]
    avg = (A + B + (A > B)) >> 1
    if n & 1:
        avg = np.concatenate([avg, v[..., -1:]], axis=-1)
    if pairs == 0:
        return avg, np.zeros(v.shape[:-1] + (0,), dtype=np.int64)
    next_avg = np.concatenate([avg[..., 1:], avg[..., -1:]], axis=-1)[..., :pairs]
    if avg.shape[-1] == pairs:  # even n: the last pair has no next avg -> itself
        next_avg[..., -1] = avg[..., pairs - 1]
    left = np.concatenate([avg[..., :1], B[..., :-1]], axis=-1)
    res = (A - B) - tendency(left, avg[..., :pairs], next_avg)
    return avg, res


class _Grid:
    """A transformed channel: a sub-rectangle of buffer `buf` (>= 0) or meta table ~buf, with the shifts
    Squeeze has accumulated (-1: an unshiftable meta channel) and the size of the untransformed channel."""
    def __init__(self, buf, x0, y0, w, h, hshift=0, vshift=0, orig_w=None, orig_h=None):
        self.buf, self.x0, self.y0, self.w, self.h = buf, x0, y0, w, h
        self.hshift, self.vshift = hshift, vshift
        self.orig_w = w if orig_w is None else orig_w
        self.orig_h = h if orig_h is None else orig_h


def default_squeeze_params(grids):
    """Squeeze::set_default_params (jxl-modular/src/transform.rs:285-341), no meta channels."""
    sp = []
    w, h = grids[0].w, grids[0].h
    if len(grids) >= 3 and grids[1].w == w and grids[1].h == h:
        sp.append((1, 0, 1, 2))
        sp.append((0, 0, 1, 2))
    num_c = len(grids)
    if h >= w and h > 8:
        sp.append((0, 1, 0, num_c)); h = (h + 1) // 2
    while w > 8 or h > 8:
        if w > 8:
            sp.append((1, 1, 0, num_c)); w = (w + 1) // 2
        if h > 8:
            sp.append((0, 1, 0, num_c)); h = (h + 1) // 2
    return sp  # (horizontal, in_place, begin_c, num_c)


def forward_squeeze(bufs, grids, steps, quant=None):
    """Applies the steps in order on the int64 working buffers, carving sub-rectangles exactly as
    transform_channel_info does; `quant(level, residual)` optionally quantises residuals (lossy)."""
    for level, (horizontal, in_place, begin, num_c) in enumerate(steps):
        end = begin + num_c
        residuals = []
        for g in grids[begin:end]:
            view = bufs[g.buf][g.y0:g.y0 + g.h, g.x0:g.x0 + g.w]
            if horizontal:
                avg, res = forward_squeeze_rows(view.copy())
                aw = avg.shape[1]
                if quant is not None:
                    res = quant(level, res)
                view[:, :aw] = avg
                view[:, aw:] = res
                g.hshift += 1
                r = _Grid(g.buf, g.x0 + aw, g.y0, g.w - aw, g.h, g.hshift, g.vshift, g.orig_w, g.orig_h)
                g.w = aw
            else:
                avg, res = forward_squeeze_rows(view.T.copy())
                ah = avg.shape[1]
                if quant is not None:
                    res = quant(level, res)
                view[:ah, :] = avg.T
                view[ah:, :] = res.T
                g.vshift += 1
                r = _Grid(g.buf, g.x0, g.y0 + ah, g.w, g.h - ah, g.hshift, g.vshift, g.orig_w, g.orig_h)
                g.h = ah
            residuals.append(r)
        at = end if in_place else len(grids)
        grids[at:at] = residuals
    return grids


def forward_rct(a, b, c, rct_type):
    """Inverse of inverse_row_*_base + inverse_permute (jxl-modular/src/transform/rct.rs:154-256):
    given output planes (d, e, f) after un-permutation, produce the coded (a, b, c)."""
    permutation, ty = rct_type // 7, rct_type % 7
    planes = [a, b, c]
    # inverse_permute maps coded rows (d,e,f) -> output; undo it first
    inv = {0: (0, 1, 2), 1: (1, 2, 0), 2: (2, 0, 1), 3: (0, 2, 1), 4: (1, 0, 2), 5: (2, 1, 0)}
    # output[k] = coded[src[k]]: derive from the swap sequences in inverse_permute
    src = {0: [0, 1, 2], 1: [2, 0, 1], 2: [1, 2, 0], 3: [0, 2, 1], 4: [1, 0, 2], 5: [2, 1, 0]}[permutation]
    coded = [None, None, None]
    for k in range(3):
        coded[src[k]] = planes[k]
    d, e, f = coded
    if ty == 6:
        # d = f + b ; f = tmp - (b>>1) ; e = c + tmp ; tmp = a - (c>>1)
        bb = d - f
        tmp = f + (bb >> 1)
        cc = e - tmp
        aa = tmp + (cc >> 1)
        return aa, bb, cc
    aa = d
    cc = f - aa if (ty & 1) else f
    if (ty >> 1) == 1:
        bb = e - aa
    elif (ty >> 1) == 2:
        bb = e - ((aa + f) >> 1)
    else:
        bb = e
    return aa, bb, cc


def channel_tiles(grids, nb_meta, group_dim):
    """For every transformed channel (in order) the list of its (x0, y0, w, h) decode units, as the
    format carves them (18181-1 clause on GlobalModular / ModularLfGroup / ModularGroup): the leading
    meta channels and the leading channels no larger than one group are coded whole in the global
    section; once a channel is larger than group_dim in either direction, it and every later channel
    is coded per group — in group_dim-sized groups scaled by the channel's shifts when min(hshift,
    vshift) < 3, in 8 x group_dim LF groups otherwise — with as many groups as the UNTRANSFORMED
    channel has.  Written from that rule, not from oracle/ or csrc/."""
    out = []
    grouped = False
    for i, g in enumerate(grids):
        if not grouped and (i < nb_meta or (g.w <= group_dim and g.h <= group_dim)):
            out.append([(0, 0, g.w, g.h)] if g.w and g.h else [])
            continue
        grouped = True
        assert g.hshift >= 0 and g.vshift >= 0
        dim = group_dim if min(g.hshift, g.vshift) < 3 else group_dim * 8
        tw, th = dim >> g.hshift, dim >> g.vshift
        assert tw > 0 and th > 0
        ncols, nrows = -(-g.orig_w // dim), -(-g.orig_h // dim)
        tiles = []
        for gy in range(nrows):
            for gx in range(ncols):
                x0, y0 = gx * tw, gy * th
                if x0 < g.w and y0 < g.h:
                    tiles.append((x0, y0, min(tw, g.w - x0), min(th, g.h - y0)))
        out.append(tiles)
    return out


_wp_lib = None


def _wp_forward_lib():
    """synth_wp.c (the encoder-side weighted predictor, a C transcription of weighted_residuals) built
    on demand next to this file."""
    global _wp_lib
    if _wp_lib is None:
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        src, so = os.path.join(here, "synth_wp.c"), os.path.join(here, "_synth_wp.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-fwrapv", "-shared", "-fPIC", src, "-o", so])
        _wp_lib = C.CDLL(so)
        _wp_lib.synth_wp_residuals_tile.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                   C.POINTER(C.c_int32)]
    return _wp_lib


def tile_residuals(t, predictor, offset=0, wp=None, fast_wp=True):
    """Residuals of ONE decode unit `t` (2-D int64) for a single-leaf tree with `predictor`."""
    if predictor == 6:
        wp = DEFAULT_WP if wp is None else wp
        if not fast_wp:
            return weighted_residuals(t, max(t.shape), wp) - offset
        t = np.ascontiguousarray(t, dtype=np.int64)
        out = np.zeros_like(t)
        h, w = t.shape
        _wp_forward_lib().synth_wp_residuals_tile(t.ctypes.data, w, w, h, out.ctypes.data, w, (C.c_int32 * 11)(*wp))
        return out - offset
    return predictor_residuals(t, max(t.shape), predictor, offset)


def residuals_in_place(bufs, metas, grids, nb_meta, group_dim, predictor, offset=0, wp=None):
    """Replaces the samples of every transformed channel by its residuals, decode unit by decode unit."""
    for g, tiles in zip(grids, channel_tiles(grids, nb_meta, group_dim)):
        arr = bufs[g.buf] if g.buf >= 0 else metas[~g.buf]
        view = arr[g.y0:g.y0 + g.h, g.x0:g.x0 + g.w]
        src = view.copy()
        for (x0, y0, w, h) in tiles:
            view[y0:y0 + h, x0:x0 + w] = tile_residuals(src[y0:y0 + h, x0:x0 + w].astype(np.int64), predictor, offset, wp)


def channel_units(grids, nb_meta, group_dim):
    """The decode units of every transformed channel in the order JxlGpuModularDesc::unit_leaves lists them: like
    channel_tiles, but a grouped channel contributes all ncols x nrows subgrids of its fixed-count grid in raster order —
    None for the ones that fall outside the (smaller) transformed channel — and a channel with a zero dimension nothing."""
    out = []
    grouped = False
    for i, g in enumerate(grids):
        if not (g.w and g.h):
            out.append([])
            continue
        if not grouped and (i < nb_meta or (g.w <= group_dim and g.h <= group_dim)):
            out.append([(0, 0, g.w, g.h)])
            continue
        grouped = True
        dim = group_dim if min(g.hshift, g.vshift) < 3 else group_dim * 8
        tw, th = dim >> g.hshift, dim >> g.vshift
        ncols, nrows = -(-g.orig_w // dim), -(-g.orig_h // dim)
        units = []
        for gy in range(nrows):
            for gx in range(ncols):
                x0, y0 = gx * tw, gy * th
                units.append((x0, y0, min(tw, g.w - x0), min(th, g.h - y0)) if x0 < g.w and y0 < g.h else None)
        out.append(units)
    return out


def residuals_in_place_leaves(bufs, metas, grids, nb_meta, group_dim, rng, wp=None, predictors=None, axis_leaves=None):
    """Like residuals_in_place, with a leaf of its own for every decode unit — what a tree that splits on the static
    properties channel / stream index gives (make_flat_tree, ma.rs:38-41): predictor drawn per unit, multiplier 1 (the chain
    stays lossless), a small offset.  Returns the leaves [(predictor, multiplier, offset)] in unit_leaves order.
    `axis_leaves` (a list, appended to): about half of the units get a tree that splits on y (property 2) or on x (property 3)
    instead — runs of rows / columns with a leaf each, as thresholds on one property give; their unit entry is
    (LEAF_BY_ROW | LEAF_BY_COLUMN, first index into axis_leaves, 0) and axis_leaves receives one leaf per row / column.  The
    residual of a sample is that of ITS leaf's predictor computed from the finished image: the predictor state (and the
    self-correcting predictor's, whose residuals come from the independent forward of synth_wp.c run over the whole unit) does
    not depend on which leaf coded the other samples."""
    predictors = list(range(14)) if predictors is None else predictors
    leaves = []
    for g, units in zip(grids, channel_units(grids, nb_meta, group_dim)):
        arr = bufs[g.buf] if g.buf >= 0 else metas[~g.buf]
        view = arr[g.y0:g.y0 + g.h, g.x0:g.x0 + g.w]
        src = view.copy()
        for u in units:
            pred = int(predictors[int(rng.integers(0, len(predictors)))])
            off = int(rng.integers(-3, 4))
            mapped = axis_leaves is not None and u is not None and rng.random() < 0.5
            if not mapped:
                leaves.append((pred, 1, off))
                if u is None:
                    continue
                x0, y0, w, h = u
                view[y0:y0 + h, x0:x0 + w] = tile_residuals(src[y0:y0 + h, x0:x0 + w].astype(np.int64), pred, off, wp)
                continue
            x0, y0, w, h = u
            by_row = bool(rng.integers(0, 2))
            n = h if by_row else w
            # 1..5 runs along the axis (thresholds of a small tree), a leaf each
            cuts = sorted(set(int(c) for c in rng.integers(1, max(n, 2), size=int(rng.integers(0, 5))) if c < n))
            bounds = [0] + cuts + [n]
            per = []
            for a, b in zip(bounds[:-1], bounds[1:]):
                lp = int(predictors[int(rng.integers(0, len(predictors)))])
                lo = int(rng.integers(-3, 4))
                per += [(lp, 1, lo)] * (b - a)
            leaves.append((abi.LEAF_BY_ROW if by_row else abi.LEAF_BY_COLUMN, len(axis_leaves), 0))
            axis_leaves += per
            tile = src[y0:y0 + h, x0:x0 + w].astype(np.int64)
            res_of = {p_: tile_residuals(tile, p_, 0, wp) for p_ in sorted({p_ for p_, _, _ in per})}
            out = np.empty_like(tile)
            for i, (lp, _, lo) in enumerate(per):
                if by_row:
                    out[i, :] = res_of[lp][i, :] - lo
                else:
                    out[:, i] = res_of[lp][:, i] - lo
            view[y0:y0 + h, x0:x0 + w] = out
    return leaves


def gradient_residuals(img, group_dim):
    """Residuals such that decode_simple_grad's arithmetic (image.rs:821-872) rebuilds `img`,
    independently per group_dim x group_dim tile."""
    H, W = img.shape
    out = np.zeros_like(img)
    for y0 in range(0, H, group_dim):
        for x0 in range(0, W, group_dim):
            t = img[y0:y0 + group_dim, x0:x0 + group_dim].astype(np.int64)
            h, w = t.shape
            pred = np.zeros_like(t)
            pred[0, 1:] = t[0, :-1]
            pred[1:, 0] = t[:-1, 0]
            n = t[:-1, 1:]; wv = t[1:, :-1]; nw = t[:-1, :-1]
            pred[1:, 1:] = np.clip(n + wv - nw, np.minimum(n, wv), np.maximum(n, wv))
            out[y0:y0 + h, x0:x0 + w] = t - pred
    return out


def _tdiv(a, b):
    """Integer division truncating toward zero (Rust `/` on i64)."""
    q = np.abs(a) // b
    return np.where(a < 0, -q, q)


def predictor_residuals(img, group_dim, predictor, offset=0):
    """Residuals of a single-leaf tree with one of the stateless predictors (all but 6) so that
    `residual + offset + predict(neighbours)` rebuilds `img`, independently per tile.  The
    neighbour rules are written from the format's definition (JPEG XL 18181-1 H.3: W = left, or
    N at x = 0, or 0 at the origin; N = top or W; NW = top-left or W; NE = top-right or N;
    NN = two up or N; NEE = two right of N or NE; WW = two left or W), vectorised over the
    finished image — not from the decoder's running state."""
    H, W = img.shape
    out = np.zeros_like(img)
    for y0 in range(0, H, group_dim):
        for x0 in range(0, W, group_dim):
            t = img[y0:y0 + group_dim, x0:x0 + group_dim].astype(np.int64)
            h, w = t.shape
            wv = np.zeros_like(t)
            wv[:, 1:] = t[:, :-1]
            wv[1:, 0] = t[:-1, 0]
            n = wv.copy()
            n[1:, :] = t[:-1, :]
            nw = wv.copy()
            nw[1:, 1:] = t[:-1, :-1]
            ne = n.copy()
            ne[1:, :-1] = t[:-1, 1:]
            nn = n.copy()
            nn[2:, :] = t[:-2, :]
            nee = ne.copy()
            nee[1:, :-2] = t[:-1, 2:]
            ww = wv.copy()
            ww[:, 2:] = t[:, :-2]
            if predictor == 0: pred = np.zeros_like(t)
            elif predictor == 1: pred = wv
            elif predictor == 2: pred = n
            elif predictor == 3: pred = _tdiv(wv + n, 2)
            elif predictor == 4: pred = np.where(np.abs(n - nw) < np.abs(wv - nw), wv, n)
            elif predictor == 5: pred = np.clip(n + wv - nw, np.minimum(n, wv), np.maximum(n, wv))
            elif predictor == 7: pred = ne
            elif predictor == 8: pred = nw
            elif predictor == 9: pred = ww
            elif predictor == 10: pred = _tdiv(wv + nw, 2)
            elif predictor == 11: pred = _tdiv(n + nw, 2)
            elif predictor == 12: pred = _tdiv(n + ne, 2)
            elif predictor == 13: pred = _tdiv(6 * n - 2 * nn + 7 * wv + ww + nee + 3 * ne + 8, 16)
            else: raise ValueError(predictor)
            out[y0:y0 + h, x0:x0 + w] = t - pred - offset
    return out


DELTA_PALETTE = [
    (0, 0, 0), (4, 4, 4), (11, 0, 0), (0, 0, -13), (0, -12, 0), (-10, -10, -10),
    (-18, -18, -18), (-27, -27, -27), (-18, -18, 0), (0, 0, -32), (-32, 0, 0), (-37, -37, -37),
    (0, -32, -32), (24, 24, 45), (50, 50, 50), (-45, -24, -24), (-24, -45, -45), (0, -24, -24),
    (-34, -34, 0), (-24, 0, -24), (-45, -45, -24), (64, 64, 64), (-32, 0, -32), (0, -32, 0),
    (-32, 0, 32), (-24, -45, -24), (45, 24, 45), (24, -24, -45), (-45, -24, 24), (80, 80, 80),
    (64, 0, 0), (0, 0, -64), (0, -64, -64), (-24, -24, 45), (96, 96, 96), (64, 64, 0),
    (45, -24, -24), (34, -34, 0), (112, 112, 112), (24, -45, -45), (45, 45, -24), (0, -32, 32),
    (24, -24, 45), (0, 96, 96), (45, -24, 24), (24, -45, -24), (-24, -45, 24), (0, -64, 0),
    (96, 0, 0), (128, 128, 128), (64, 0, 64), (144, 144, 144), (96, 96, 0), (-36, -36, 36),
    (45, -24, -45), (45, -45, -24), (0, 0, -96), (0, 128, 128), (0, 96, 0), (45, 24, -45),
    (-128, 0, 0), (24, -45, 24), (-45, 24, -45), (64, 0, -64), (64, -64, -64), (96, 0, 96),
    (45, -45, 24), (24, 45, -45), (64, 64, -64), (128, 128, 0), (0, 0, -128), (-24, 45, -45),
]  # the format's fixed delta palette (18181-1 table H.x), data


def palette_delta_reference(idx, pal, ncol, ndelta, d_pred, bit_depth, wrap_bits):
    """Slow, direct evaluation of the palette inverse with delta entries for the stateless
    predictors W(1), N(2), Gradient(5) and Zero(0): colour lookup per the format, then a raster
    scan per channel where delta pixels add the prediction from already finished neighbours."""
    H, W = idx.shape
    out = np.zeros((3, H, W), dtype=np.int64)
    maxv = (1 << bit_depth) - 1
    for y in range(H):
        for x in range(W):
            i = int(idx[y, x])
            for c in range(3):
                if 0 <= i < ncol:
                    v = int(pal[c, i])
                elif i >= ncol:
                    j = i - ncol
                    if j < 64:
                        v = ((j >> (2 * c)) % 4) * maxv // 4 + (1 << max(bit_depth - 3, 0))
                    else:
                        j -= 64
                        v = ((j // 5 ** c) % 5) * maxv // 4
                else:
                    j = (-(i + 1)) % 143
                    v = DELTA_PALETTE[(j + 1) >> 1][c]
                    if j & 1 == 0:
                        v = -v
                    if bit_depth > 8:
                        v <<= min(bit_depth, 24) - 8
                out[c, y, x] = v
    half = 1 << (wrap_bits - 1)
    for c in range(3):
        rec = out[c].copy()  # recorded (untruncated) values
        for y in range(H):
            for x in range(W):
                if idx[y, x] < ndelta:
                    wv = rec[y, x - 1] if x > 0 else (rec[y - 1, x] if y > 0 else 0)
                    n = rec[y - 1, x] if y > 0 else wv
                    nw = rec[y - 1, x - 1] if (x > 0 and y > 0) else wv
                    if d_pred == 0: p = 0
                    elif d_pred == 1: p = wv
                    elif d_pred == 2: p = n
                    elif d_pred == 5: p = min(max(n + wv - nw, min(n, wv)), max(n, wv))
                    else: raise ValueError(d_pred)
                    rec[y, x] = ((rec[y, x] + p + (1 << 31)) % (1 << 32)) - (1 << 31)
                    out[c, y, x] = ((rec[y, x] + half) % (2 * half)) - half
    return out


DEFAULT_WP = [16, 10, 7, 7, 7, 0, 0, 13, 12, 12, 12]  # WpHeader defaults (predictor.rs:8-21)


def weighted_residuals(img, group_dim, wp=DEFAULT_WP):
    """Residuals for the self-correcting (weighted) predictor: a straight sequential Python
    transcription of JPEG XL 18181-1 H.5 (error-weighted blend of four sub-predictors), run on the
    true samples.  Slow; for small tiles only."""
    H, W = img.shape
    out = np.zeros_like(img)
    p1, p2, p3a, p3b, p3c, p3d, p3e = wp[:7]
    wmax = wp[7:11]
    div = [0] + [(1 << 24) // i for i in range(1, 65)]
    for y0 in range(0, H, group_dim):
        for x0 in range(0, W, group_dim):
            t = [[int(v) for v in row] for row in img[y0:y0 + group_dim, x0:x0 + group_dim]]
            h, w = len(t), len(t[0])
            terr = [[0] * w for _ in range(h)]                 # true_err(x, y)
            serr = [[[0] * 4 for _ in range(w)] for _ in range(h)]  # sub-predictor errors
            for y in range(h):
                for x in range(w):
                    def S(xx, yy):
                        return t[yy][xx]
                    Wv = S(x - 1, y) if x > 0 else (S(x, y - 1) if y > 0 else 0)
                    N = S(x, y - 1) if y > 0 else Wv
                    NW = S(x - 1, y - 1) if (x > 0 and y > 0) else Wv
                    NE = S(x + 1, y - 1) if (x + 1 < w and y > 0) else N
                    NN = S(x, y - 2) if y > 1 else N
                    # error neighbours: zero outside the tile, except that N/NE fall back like samples
                    def TE(xx, yy):
                        return terr[yy][xx] if (0 <= xx < w and 0 <= yy < h) else 0
                    te_w = TE(x - 1, y) if x > 0 else 0
                    te_n = TE(x, y - 1) if y > 0 else 0
                    te_nw = TE(x - 1, y - 1) if (x > 0 and y > 0) else te_n
                    te_ne = TE(x + 1, y - 1) if (x + 1 < w and y > 0) else te_n
                    def SE(xx, yy, i):
                        return serr[yy][xx][i] if (0 <= xx < w and 0 <= yy < h) else 0
                    n3, nw3, ne3, w3, nn3 = N << 3, NW << 3, NE << 3, Wv << 3, NN << 3
                    sub = [w3 + ne3 - n3,
                           n3 - (((te_w + te_n + te_ne) * p1) >> 5),
                           w3 - (((te_w + te_n + te_nw) * p2) >> 5),
                           n3 - ((te_nw * p3a + te_n * p3b + te_ne * p3c + (nn3 - n3) * p3d + (nw3 - w3) * p3e) >> 5)]
                    weight = []
                    for i in range(4):
                        # err_sum = N + W + NW + WW + NE sub-errors (W and WW counted via the running sums)
                        e_n = SE(x, y - 1, i) if y > 0 else 0
                        e_w = SE(x - 1, y, i) if x > 0 else 0
                        e_ww = SE(x - 2, y, i) if x > 1 else 0
                        e_nw = SE(x - 1, y - 1, i) if (x > 0 and y > 0) else e_n   # NW falls back onto N
                        e_ne = SE(x + 1, y - 1, i) if (x + 1 < w and y > 0) else e_n
                        es = (e_n + e_w + e_ww + e_nw + e_ne) & 0xFFFFFFFF
                        if x + 1 == w and x > 0:   # last column: NE folds onto N, which already carries W
                            es = (es + e_w) & 0xFFFFFFFF
                        shift = max(((es + 1) >> 5).bit_length() - 1, 0)
                        weight.append(4 + ((wmax[i] * div[(es >> shift) + 1]) >> shift))
                    lw = (sum(weight) >> 4).bit_length() - 1
                    weight = [v >> lw for v in weight]
                    sw = sum(weight)
                    s_ = (sw >> 1) - 1 + sum(a * b for a, b in zip(sub, weight))
                    pred = (s_ * div[sw]) >> 24
                    if ((te_n ^ te_w) | (te_n ^ te_nw)) <= 0:
                        pred = min(max(pred, min(n3, w3, ne3)), max(n3, w3, ne3))
                    v = t[y][x]
                    out[y0 + y, x0 + x] = v - ((pred + 3) >> 3)
                    terr[y][x] = pred - (v << 3)
                    serr[y][x] = [(abs(sp - (v << 3)) + 3) >> 3 for sp in sub]
    return out


class ModularWorkload:
    """kind: 'lossless_rgb8' (cfg 1: Gradient residuals + RCT), 'squeeze' (cfg 3: RCT/XYB ints +
    default Squeeze, optional lossy quantisation), 'palette', 'raw' (random data through explicit
    transforms, exercises wrapping)."""

    def __init__(self, width, height, kind="squeeze", seed=0, i16=True, lossy=True, rct_type=None,
                 xyb=True, epf_iters=0, gabor=False, bit_depth=8, predictor=5, pred_offset=0, residual=None,
                 group_dim=256, leaves=None, squeeze_plan=None):
        """`residual` (kinds 'squeeze', 'palette'): a Predictor id — the buffers then hold the RESIDUALS of
        that predictor (single-leaf MA tree) for every transformed channel, computed decode unit by decode
        unit (channel_tiles) from the transformed samples; None: they hold the samples themselves.
        `squeeze_plan` (kind 'squeeze'): one entry per Squeeze transform, applied in order — None = default parameters
        (set_default_params on the channel list as it stands then), or explicit steps [(horizontal, in_place, begin_c,
        num_c)]: steps may squeeze the residual channels of earlier steps again."""
        rng = np.random.default_rng(SEED_BASE + 0x100 + seed)
        self.width, self.height, self.kind = width, height, kind
        self.group_dim = group_dim
        self.dtype = np.int16 if i16 else np.int32
        self.sample_type = abi.SAMPLE_I16 if i16 else abi.SAMPLE_I32
        self.bit_depth = bit_depth
        self.xyb = xyb and kind == "squeeze"
        H, W = height, width
        yy, xx = np.mgrid[0:H, 0:W]
        base = [np.rint(110 + 90 * np.sin(xx / (17.0 + 5 * c)) * np.cos(yy / (23.0 - 3 * c)) +
                        6 * rng.normal(size=(H, W))).astype(np.int64) for c in range(3)]
        self.transforms = []
        self.meta = []
        self.residual_predictor = 0xFFFFFFFF
        self.residual_multiplier, self.residual_offset = 1, 0
        self.expected = None  # exact integer result when the chain is lossless
        # `leaves` = "mixed" (kinds 'predictor', 'squeeze', 'palette'): every decode unit gets a leaf of its own
        # (JxlGpuModularDesc::unit_leaves); a list of predictor ids restricts the draw
        self.unit_leaves = None
        # `leaves` = "axis": as "mixed", and about half of the units carry a tree that splits on y or on x (JxlGpuModularDesc::axis_leaves)
        self.axis_leaves = [] if leaves == "axis" else None
        leaf_rng = np.random.default_rng(SEED_BASE + 0x777 + seed)
        leaf_preds = None if leaves in (None, "mixed", "axis") else list(leaves)

        if kind == "predictor":
            # single-leaf tree with an arbitrary predictor on plain RGB8 (no transforms)
            rgb = [np.clip(p, 0, 255) for p in base]
            self.expected = [p.astype(self.dtype) for p in rgb]
            if leaves is not None:
                chans = [p.copy() for p in rgb]
                self.unit_leaves = residuals_in_place_leaves(chans, [], [_Grid(i, 0, 0, W, H) for i in range(3)], 0, group_dim,
                                                             leaf_rng, predictors=leaf_preds, axis_leaves=self.axis_leaves)
            elif predictor == 6:
                chans = [weighted_residuals(p, group_dim) for p in rgb]
            else:
                chans = [predictor_residuals(p, group_dim, predictor, pred_offset) for p in rgb]
            if leaves is None:
                self.residual_predictor = predictor
                self.residual_offset = pred_offset
            self.buffers = [p.astype(self.dtype) for p in chans]

        elif kind == "lossless_rgb8":
            rgb = [np.clip(p, 0, 255) for p in base]
            self.expected = [p.astype(self.dtype) for p in rgb]
            t = 6 if rct_type is None else rct_type
            a, b, c = forward_rct(rgb[0], rgb[1], rgb[2], t)
            self.transforms.append(("rct", 0, t))
            chans = [gradient_residuals(p, group_dim) for p in (a, b, c)]
            self.residual_predictor = 5
            self.buffers = [p.astype(self.dtype) for p in chans]
        elif kind == "squeeze":
            planes = base
            if self.xyb:
                # XYB-ish integers in Modular channel order Y, X, B (B stored as B - Y)
                yv = np.clip(base[0], 0, 255) * 4
                xv = np.rint((base[1] - 110) / 6.0).astype(np.int64)
                bv = np.clip(base[2], 0, 255) * 2 - yv
                planes = [yv, xv, bv]
            if rct_type is not None:
                planes = list(forward_rct(planes[0], planes[1], planes[2], rct_type))
                self.transforms.append(("rct", 0, rct_type))
            if not lossy:
                # expected = what the inverse chain must reproduce (before the RCT is undone the
                # planes are `planes`; after, the originals)
                orig = base if not self.xyb else [yv, xv, bv]
                self.expected = [p.astype(self.dtype) for p in orig]
            bufs = [p.copy() for p in planes]
            grids = [_Grid(i, 0, 0, W, H) for i in range(3)]
            for plan in ([None] if squeeze_plan is None else squeeze_plan):
                steps = default_squeeze_params(grids) if plan is None else [tuple(st) for st in plan]
                nsteps = len(steps)

                def quant(level, res):
                    if not lossy:
                        return res
                    q = 1 << max(0, (nsteps - level) // 5)  # coarser for the finest levels
                    return _trunc_div(res, q) * 1  # quantised residuals (dequantised form is what is coded)
                forward_squeeze(bufs, grids, steps, quant)
                self.transforms.append(("squeeze", None if plan is None else steps))
            if leaves is not None:
                self.unit_leaves = residuals_in_place_leaves(bufs, [], grids, 0, group_dim, leaf_rng, predictors=leaf_preds, axis_leaves=self.axis_leaves)
            elif residual is not None:
                residuals_in_place(bufs, [], grids, 0, group_dim, residual, pred_offset)
                self.residual_predictor, self.residual_offset = residual, pred_offset
            self.buffers = [b.astype(self.dtype) for b in bufs]
        elif kind == "palette":
            ncol = 37
            pal = rng.integers(0, 256, size=(3, ncol)).astype(np.int64)
            idx = rng.integers(0, ncol, size=(H, W)).astype(np.int64)
            self.expected = [pal[c][idx].astype(self.dtype) for c in range(3)]
            self.transforms.append(("palette", 0, 3, ncol))
            if residual is not None or leaves is not None:
                # transformed channel list: [palette table (meta, unshiftable), index channel]
                grids = [_Grid(~0, 0, 0, ncol, 3, -1, -1), _Grid(0, 0, 0, W, H)]
                bufs, metas = [idx], [pal]
                if leaves is not None:
                    self.unit_leaves = residuals_in_place_leaves(bufs, metas, grids, 1, group_dim, leaf_rng, predictors=leaf_preds, axis_leaves=self.axis_leaves)
                else:
                    residuals_in_place(bufs, metas, grids, 1, group_dim, residual, pred_offset)
                    self.residual_predictor, self.residual_offset = residual, pred_offset
            self.meta.append(pal.astype(self.dtype))
            self.buffers = [idx.astype(self.dtype), np.zeros((H, W), self.dtype), np.zeros((H, W), self.dtype)]
        elif kind == "palette_delta":
            # lossy-palette style index plane: regular entries, implicit colours (index >= nb_colours),
            # delta entries (negative indices and the first nb_deltas palette rows) that add a prediction
            ncol, ndelta = 29, 4
            pal = rng.integers(0, 256, size=(3, ncol)).astype(np.int64)
            pal[:, :ndelta] = rng.integers(-9, 10, size=(3, ndelta))
            idx = rng.integers(ndelta, ncol, size=(H, W)).astype(np.int64)
            r = rng.random(size=(H, W))
            idx = np.where(r < 0.15, rng.integers(ncol, ncol + 64 + 125, size=(H, W)), idx)       # implicit
            idx = np.where((r >= 0.15) & (r < 0.30), -rng.integers(1, 300, size=(H, W)), idx)      # DELTA_PALETTE
            idx = np.where((r >= 0.30) & (r < 0.38), rng.integers(0, ndelta, size=(H, W)), idx)    # palette deltas
            self.palette_delta = (ncol, ndelta, predictor)
            self.transforms.append(("palette", 0, 3, ncol, ndelta, predictor))
            self.meta.append(pal.astype(self.dtype))
            self.buffers = [idx.astype(self.dtype), np.zeros((H, W), self.dtype), np.zeros((H, W), self.dtype)]
            self.expected = None
            self.index_plane, self.palette = idx, pal
        elif kind == "predictor_random":
            # single-leaf tree: random small residuals (any residual plane decodes to SOME image; the
            # oracle defines which).  No Python-side encode, so usable at 8K.
            self.residual_predictor = predictor
            self.residual_offset = pred_offset
            self.buffers = [rng.integers(-3, 4, size=(H, W)).astype(self.dtype) for _ in range(3)]
        elif kind == "gray":
            # grayscale image (encoded_color_channels == 1): one channel, default Squeeze (no chroma pre-steps), lossless
            g = np.clip(base[0], 0, (1 << bit_depth) - 1)
            self.expected = [g.astype(self.dtype)]
            bufs = [g.copy()]
            grids = [_Grid(0, 0, 0, W, H)]
            forward_squeeze(bufs, grids, default_squeeze_params(grids), lambda level, res: res)
            self.transforms.append(("squeeze", None))
            if leaves is not None:
                self.unit_leaves = residuals_in_place_leaves(bufs, [], grids, 0, group_dim, leaf_rng, predictors=leaf_preds, axis_leaves=self.axis_leaves)
            elif residual is not None:
                residuals_in_place(bufs, [], grids, 0, group_dim, residual, pred_offset)
                self.residual_predictor, self.residual_offset = residual, pred_offset
            self.buffers = [b.astype(self.dtype) for b in bufs]
        elif kind in ("ycbcr420", "ycbcr422", "ycbcr440", "ycbcr444"):
            # frame_header.do_ycbcr on a Modular frame with jpeg_upsampling: channels Cb, Y, Cr, chroma subsampled;
            # plain 8-bit samples (centred chroma), no transforms
            sh = {"ycbcr420": (1, 1), "ycbcr422": (1, 0), "ycbcr440": (0, 1), "ycbcr444": (0, 0)}[kind]
            cw, ch = (W + 1) // 2 if sh[0] else W, (H + 1) // 2 if sh[1] else H
            yv = np.clip(base[0], 0, 255) - 128
            cb = (np.clip(base[1], 0, 255) - 128)[:ch, :cw]
            cr = (np.clip(base[2], 0, 255) - 128)[:ch, :cw]
            self.buffers = [np.ascontiguousarray(cb.astype(self.dtype)), yv.astype(self.dtype), np.ascontiguousarray(cr.astype(self.dtype))]
            self.ycbcr = True
        elif kind == "raw":
            # arbitrary data straight into the inverse chain: wrapping arithmetic included
            info = np.iinfo(self.dtype)
            self.buffers = [rng.integers(info.min // 2, info.max // 2, size=(H, W)).astype(self.dtype) for _ in range(3)]
            self.transforms.append(("rct", 0, 6 if rct_type is None else rct_type))
            self.transforms.append(("squeeze", None))
        else:
            raise ValueError(kind)

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
        self.color.intensity_target = 255.0
        self.color.matrix[:] = list(OPSIN_INV)
        self.color.transfer_function = abi.TF_SRGB
        self._keep = None

    def shapes(self):
        return [b.shape for b in self.buffers]

    def desc(self):
        d = abi.ModularDesc()
        d.abi = abi.ABI_VERSION
        d.sample_type = self.sample_type
        d.bit_depth = self.bit_depth
        n = len(self.buffers)
        chans = (abi.ModularChannel * n)()
        for i, b in enumerate(self.buffers):
            chans[i].data = b.ctypes.data
            chans[i].width, chans[i].height = b.shape[1], b.shape[0]
        d.num_channels = n
        d.num_color_channels = 1 if self.kind == "gray" else 3
        d.channels = C.cast(chans, C.POINTER(abi.ModularChannel))
        metas = (abi.ModularChannel * max(1, len(self.meta)))()
        for i, m in enumerate(self.meta):
            metas[i].data = m.ctypes.data
            metas[i].width, metas[i].height = m.shape[1], m.shape[0]
        d.num_meta_channels = len(self.meta)
        d.meta_channels = C.cast(metas, C.POINTER(abi.ModularChannel))
        trs = (abi.Transform * len(self.transforms))()
        sq_keep = []
        for i, t in enumerate(self.transforms):
            if t[0] == "rct":
                trs[i].kind = abi.TR_RCT
                trs[i].begin_c, trs[i].rct_type = t[1], t[2]
            elif t[0] == "squeeze":
                trs[i].kind = abi.TR_SQUEEZE
                trs[i].num_sq = 0
                if t[1]:
                    sq = (abi.SqueezeStep * len(t[1]))()
                    for j, (hor, inp, beg, num) in enumerate(t[1]):
                        sq[j].horizontal, sq[j].in_place, sq[j].begin_c, sq[j].num_c = hor, inp, beg, num
                    trs[i].num_sq = len(t[1])
                    trs[i].sq = C.cast(sq, C.POINTER(abi.SqueezeStep))
                    sq_keep.append(sq)
            else:
                trs[i].kind = abi.TR_PALETTE
                trs[i].begin_c, trs[i].num_c, trs[i].nb_colours = t[1], t[2], t[3]
                if len(t) > 4:
                    trs[i].nb_deltas, trs[i].d_pred = t[4], t[5]
                    trs[i].wp_params[:] = DEFAULT_WP
        d.num_transforms = len(self.transforms)
        d.transforms = C.cast(trs, C.POINTER(abi.Transform))
        d.residual_predictor = self.residual_predictor
        d.residual_multiplier = self.residual_multiplier
        d.residual_offset = self.residual_offset
        d.wp_params[:] = DEFAULT_WP
        d.group_dim = self.group_dim
        d.xyb_encoded = 1 if self.xyb else 0
        d.m_lf_unscaled[:] = [(1.0 / 32.0) / 128.0, (1.0 / 4.0) / 128.0, (1.0 / 2.0) / 128.0]
        d.filter = self.filter
        d.upsampling.factor = 1
        d.color = self.color
        if getattr(self, "ycbcr", False):
            d.color.ycbcr = 1
        d.noise = getattr(self, "noise", abi.NoiseParams())
        self._keep = [chans, metas, trs, sq_keep]
        if self.unit_leaves is not None:
            lv = (abi.MaLeaf * max(1, len(self.unit_leaves)))()
            for i, (pred, mul, off) in enumerate(self.unit_leaves):
                lv[i].predictor, lv[i].multiplier, lv[i].offset = pred, mul, off
            d.unit_leaves = C.cast(lv, C.POINTER(abi.MaLeaf))
            d.num_unit_leaves = len(self.unit_leaves)
            self._keep.append(lv)
            if self.axis_leaves:
                al = (abi.MaLeaf * len(self.axis_leaves))()
                for i, (pred, mul, off) in enumerate(self.axis_leaves):
                    al[i].predictor, al[i].multiplier, al[i].offset = pred, mul, off
                d.axis_leaves = C.cast(al, C.POINTER(abi.MaLeaf))
                d.num_axis_leaves = len(self.axis_leaves)
                self._keep.append(al)
        return d
