"""A CPU model of the SCHEDULE of the lane-packed predictor kernels (csrc/modular.hip, predict_lanes_kernel /
predict_lanes_narrow_kernel; at the end of the file predict_lanes_wp4_kernel, whose rings are indexed by the step): which lane produces which sample at which step, which LDS ring slot it lands in, and which
slots the rows below read — checked for every read against what the ring holds at that moment.  It pins the three sizing rules
the host code applies (run_inverse, "M4"):

  * lanes per subgrid P = min(64, max(1, pow2ceil(gw) / 4)); columns per round DP = max(4 P, pow2ceil(gw)); row r + 1 trails
    row r by D = DP / P columns (4, 8 for subgrids of up to 512 columns, 16 up to 1024: group_dim 1024);
  * the sample ring of a lane has 16 columns for D <= 8 and 64 for D = 16 (row r - 2 is 2 D columns ahead; reads of a step come
    before its writes, so 2 D == ring size is still fine);
  * the self-correcting predictor's error rows are shared by all rows of a subgrid and overwritten in place, like the
    reference's (predictor.rs:394-441): row r must read column x + 2 after row r - 1 wrote it and before row r itself does.

The arithmetic is not modelled (the GPU parity tests do that); only data movement is: a sample is its (row, column) tag."""
import pytest


def pow2ceil(v):
    p = 1
    while p < v:
        p <<= 1
    return p


def plan(gw):
    """lanes_of / dp_of of run_inverse."""
    p = min(64, max(1, pow2ceil(gw) // 4))
    dp = max(4 * p, pow2ceil(gw))
    return p, dp


def ring_columns(gw):
    p, dp = plan(gw)
    return 16 if dp // p <= 8 else 64


def simulate(gw, gh, ring):
    """Runs the wave schedule of one subgrid; raises AssertionError at the first read that does not find the sample the
    reference's PredictorState would hold (N / NE / NEE from row r - 1, NN from row r - 2, the error of (r - 1, x + 2))."""
    P, DP = plan(gw)
    D = DP // P
    log2dp = DP.bit_length() - 1
    s_out = [[None] * ring for _ in range(P)]      # per lane: tag of the sample in each ring slot
    s_in = [[None] * 16 for _ in range(P)]         # residuals parked ahead (stream position & 15)
    err = [None] * max(gw, 1)                      # the shared error row: tag of the row that wrote the column last
    produced = set()
    steps = gw + D * (gh - 1)

    def where(k, q):
        if q < 0:
            return None
        r, x = k + (q >> log2dp) * P, q & (DP - 1)
        return (r, x) if r < gh and x < gw else None

    for s0 in range(-16, steps, 8):
        for j in range(8):
            step = s0 + j
            reads, writes, err_reads, err_writes = [], [], [], []
            for k in range(P):
                u = step - D * k
                # residual pipeline: park position u + 8 (requested 8 steps ago), request u + 16
                ahead = where(k, u + 8)
                if ahead is not None:
                    s_in[k][(u + 8) & 15] = ahead
                here = where(k, u)
                if here is None:
                    continue
                r, x = here
                assert s_in[k][u & 15] == (r, x), f"residual of {(r, x)} not parked (gw {gw})"
                rnd = u >> log2dp
                k1, w1 = (k - 1, 0) if k >= 1 else (k - 1 + P, 1)
                k2, w2 = ((k - 2, 0) if k >= 2 else (k - 2 + P, 1)) if P > 1 else (0, 2)
                if P == 1:
                    k1, w1 = 0, 1
                elif P == 2 and k < 2:
                    k2, w2 = k, 1     # q2 = k - 2 + P = k: the lane's own previous round
                pb1 = ((rnd - w1) << log2dp) & (ring - 1)
                pb2 = ((rnd - w2) << log2dp) & (ring - 1)
                if r >= 1:
                    if x == 0:
                        reads.append((k1, pb1 & (ring - 1), (r - 1, 0), 'n0'))
                    if x + 1 < gw:
                        reads.append((k1, (pb1 + x + 1) & (ring - 1), (r - 1, x + 1), 'ne'))   # NE, and N of the next column
                    if x + 2 < gw:
                        reads.append((k1, (pb1 + x + 2) & (ring - 1), (r - 1, x + 2), 'nee'))   # NEE
                    err_reads.append((min(x + 2, gw - 1), r - 1))
                    if x == 0:
                        err_reads.append((0, r - 1))
                        err_reads.append((min(1, gw - 1), r - 1))
                if r >= 2:
                    reads.append((k2, (pb2 + x) & (ring - 1), (r - 2, x), 'nn'))               # NN
                writes.append((k, u & (ring - 1), (r, x)))
                err_writes.append((x, r))
            # a wave executes a step in lock-step: every read of the step before its writes (the statement order of the kernels)
            for lane, slot, want, _ in reads:
                assert s_out[lane][slot] == want, f"gw {gw} gh {gh} ring {ring}: slot holds {s_out[lane][slot]}, wanted {want} at step {step}"
            for col, want_row in err_reads:
                assert err[col] == want_row, f"gw {gw} gh {gh}: error column {col} holds row {err[col]}, wanted {want_row} at step {step}"
            for lane, slot, tag in writes:
                s_out[lane][slot] = tag
                produced.add(tag)
            for col, row in err_writes:
                err[col] = row
            # the one read behind the write: N of the next column (Properties::record) — the producer is D - 1 >= 3 columns further
            for lane, slot, want, kind in reads:
                if kind == 'ne':
                    assert s_out[lane][slot] == want, f"gw {gw} gh {gh} ring {ring}: N of the next column overwritten at step {step}"
    assert len(produced) == gw * gh, f"gw {gw} gh {gh}: {len(produced)} of {gw * gh} samples produced in {steps} steps"


_WIDTHS = [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 31, 33, 64, 65, 100, 128, 255, 256, 257, 300, 511, 512, 513, 525, 777, 1023, 1024]


@pytest.mark.parametrize("gw", _WIDTHS)
def test_ring_and_error_rows_hold_what_the_rows_below_read(gw):
    P, DP = plan(gw)
    assert DP >= gw and DP % P == 0 and DP // P >= 4 and (DP // P) % 4 == 0
    for gh in sorted({1, 2, 3, P - 1, P, P + 1, 2 * P + 1, 3 * P} - {0, -1}):
        if gw * gh > 200_000:      # keep the CPU suite short: wide subgrids with up to P + 1 rows
            continue
        simulate(gw, gh, ring_columns(gw))


def test_the_model_sees_a_ring_that_is_too_small():
    """D = 16 (a subgrid wider than 512 columns) with the 16-column ring: row r - 2 is 32 columns ahead and has overwritten
    what row r reads — the bug a group_dim 1024 frame would hit without the 64-column ring."""
    with pytest.raises(AssertionError):
        simulate(600, 5, 16)
    simulate(600, 5, 64)
    # 2 D == ring size is the limit that still works (reads of a step precede its writes): D = 8 with 16 columns
    simulate(512, 4, 16)
    with pytest.raises(AssertionError):
        simulate(512, 4, 8)


def test_error_row_width_rule():
    """The dynamic LDS of a launch holds 5 error rows of `lane_err_w` columns; a subgrid of P lanes owns lane_err_w * P / 64 of
    them (ecol).  lane_err_w must be the MAXIMUM the launch's waves need — 256 by default, 512 with a wave of DP > 256, 1024 with
    DP > 512 — whatever order the waves are planned in (a wave of DP 512 planned after one of DP 1024 must not shrink it)."""
    def lane_err_w(widths):
        w = 256
        for gw in widths:
            _, dp = plan(gw)
            if dp > 256:
                w = max(w, 512)
            if dp > 512:
                w = 1024
        return w
    for widths in ([1024, 512, 26], [512, 1024], [26, 525, 300], [300, 200], [64]):
        w = lane_err_w(widths)
        for gw in widths:
            p, _ = plan(gw)
            assert w * p // 64 >= gw, (widths, gw, w)
        assert w == lane_err_w(list(reversed(widths)))


# ---------------------------------------------------------------- predict_lanes_wp4_kernel (round 6): D = 4, rings indexed by the STEP
def simulate_wp4(gw, gh, sample_ring=16, err_ring=4):
    """The D = 4 form of the self-correcting predictor step.  A lane writes its sample into slot s & 15 of its own ring and its
    error record into slot s & 3, whatever its stream position; what the row below needs was written at a step that does not
    depend on the lane: NE (column x + 1) at s + 1 - D, NN at s - 2 D, the errors of column x + 2 at s + 2 - D — also across the
    wrap from lane 0 to lane P - 1 of the round before.  Everything is requested one step before it is used (into
    registers); a row start (only at steps that are multiples of 4) reads columns 0, 1, 2 of the row above in place.  Row 0
    reads a ring of zeros (not modelled: it has nothing to check).  Statement order of a step, as in the kernel: park
    residuals, row-start reads, requests for step s + 1, [arithmetic], writes."""
    P, DP = plan(gw)
    D = DP // P
    assert D == 4, "the kernel serves launches in which every wave has D = 4"
    log2dp = DP.bit_length() - 1
    s_out = [[None] * sample_ring for _ in range(P)]
    s_err = [[None] * err_ring for _ in range(P)]
    s_in = [[None] * 16 for _ in range(P)]
    nx = [dict(res=None, pne=None, pnn=None, err=None) for _ in range(P)]     # registers: requested at the previous step
    use_prev_err = [False] * P                                                  # pse / pte: the previous lane's ring (True) or the zeros
    produced = set()
    steps = gw + D * (gh - 1)

    def where(k, q):
        if q < 0:
            return None
        r, x = k + (q >> log2dp) * P, q & (DP - 1)
        return (r, x) if r < gh and x < gw else None

    for s0 in range(-16, steps, 16):
        for j in range(16):
            s = s0 + j
            cur = [dict(n) for n in nx]
            # 1. residuals: four at a time at every fourth step (positions u + 8 .. u + 11 of every lane)
            if j % 4 == 0:
                for k in range(P):
                    u = s - D * k
                    for i in range(4):
                        s_in[k][(j + 8 + i) & 15] = where(k, u + 8 + i)
            # 2. row starts: columns 0, 1, 2 of the row above, in place
            for k in range(P):
                u = s - D * k
                here = where(k, u)
                x = u & (DP - 1)
                if j % 4 == 0 and x == 0:
                    r = k + (u >> log2dp) * P if u >= 0 else None
                    use_prev_err[k] = bool(r)      # row 0 (and positions before the first row): the zeros
                    if here is not None and r >= 1:
                        kp = (k - 1) % P
                        assert s_out[kp][(j + 12) % sample_ring] == (r - 1, 0), f"wp4 gw {gw} gh {gh}: N of column 0 at step {s}"
                        assert s_err[kp][(j + 0) % err_ring] == (r - 1, 0), f"wp4 gw {gw} gh {gh}: errors of column 0 at step {s}"
                        if gw > 1:
                            assert s_err[kp][(j + 1) % err_ring] == (r - 1, 1), f"wp4 gw {gw} gh {gh}: errors of column 1 at step {s}"
                        if gw > 2:
                            assert s_err[kp][(j + 2) % err_ring] == (r - 1, 2), f"wp4 gw {gw} gh {gh}: errors of column 2 at step {s}"
                            cur[k]["err"] = s_err[kp][(j + 2) % err_ring]
            # 3. requests for step s + 1
            ahead = 1
            for k in range(P):
                kp, kpp = (k - 1) % P, (k - 2) % P
                nx[k] = dict(res=s_in[k][(j + ahead) & 15],
                             pne=s_out[kp][(j + ahead + 1 - D) % sample_ring],
                             pnn=s_out[kpp][(j + ahead - 2 * D) % sample_ring],
                             err=s_err[kp][(j + ahead + 2 - D) % err_ring] if use_prev_err[k] else None)
            # 4. what the arithmetic of this step consumes
            tags = []
            for k in range(P):
                u = s - D * k
                here = where(k, u)
                tags.append(here)
                if here is None:
                    continue
                r, x = here
                assert cur[k]["res"] == (r, x), f"wp4 gw {gw} gh {gh}: residual of {(r, x)} at step {s}: {cur[k]['res']}"
                if r >= 1 and x + 1 < gw:
                    assert cur[k]["pne"] == (r - 1, x + 1), f"wp4 gw {gw} gh {gh}: NE of {(r, x)}: {cur[k]['pne']}"
                if r >= 2:
                    assert cur[k]["pnn"] == (r - 2, x), f"wp4 gw {gw} gh {gh}: NN of {(r, x)}: {cur[k]['pnn']}"
                if r >= 1 and x + 2 < gw:
                    assert cur[k]["err"] == (r - 1, x + 2), f"wp4 gw {gw} gh {gh}: errors two columns ahead of {(r, x)}: {cur[k]['err']}"
                produced.add(here)
            # 5. writes (unconditional: an off-grid lane writes garbage)
            for k in range(P):
                s_out[k][j % sample_ring] = tags[k]
                s_err[k][j % err_ring] = tags[k]
    assert len(produced) == gw * gh, f"wp4 gw {gw} gh {gh}: {len(produced)} of {gw * gh} samples produced"


@pytest.mark.parametrize("gw", [w for w in _WIDTHS if w <= 256])
def test_step_indexed_rings_hold_what_the_rows_below_read(gw):
    P, DP = plan(gw)
    assert DP == 4 * P
    for gh in sorted({1, 2, 3, 4, P - 1, P, P + 1, 2 * P + 1, 3 * P, 5 * P + 2} - {0, -1}):
        if gw * gh > 120_000:
            continue
        simulate_wp4(gw, gh)


def test_the_step_indexed_model_sees_its_own_limits():
    """Four error slots are exactly enough (a row start reads the slot that this step's write is about to take: reads come
    first); two or three are not.  The sample ring needs eight columns (NN: written 8 steps earlier, requested 7 steps
    later); the kernel has 16, the unroll length of its step loop; four lose NN."""
    simulate_wp4(100, 40)
    simulate_wp4(100, 40, sample_ring=8)
    for kw in (dict(err_ring=2), dict(err_ring=3), dict(sample_ring=4)):
        with pytest.raises(AssertionError):
            simulate_wp4(100, 40, **kw)


def simulate_wp4_stores(gw, gh):
    """The four-sample path's stores: a lane keeps the four groups of its current 16-column block (packs[g], g = group of the
    16 x unrolled loop) and stores them when its column is 15 (mod 16); groups past the subgrid's width go to the sink; a wave
    with a width that is not a multiple of 16 runs 12 steps longer (its last block ends after its last sample).  Every sample
    must reach memory exactly once, at its own (row, column)."""
    P, DP = plan(gw)
    assert DP == 4 * P and DP >= 16 and gw % 4 == 0
    log2dp = DP.bit_length() - 1
    all16 = gw % 16 == 0
    steps = gw + 4 * (gh - 1) + (0 if all16 else 12)
    packs = [[[None] * 4 for _ in range(4)] for _ in range(P)]
    row_ok, row = [False] * P, [None] * P
    stored = {}
    for s0 in range(-16, steps, 16):
        for j in range(16):
            s = s0 + j
            for k in range(P):
                u = s - 4 * k
                x = u & (DP - 1)
                if j % 4 == 0 and x == 0:
                    r = k + (u >> log2dp) * P if u >= 0 else None
                    row_ok[k] = r is not None and r < gh
                    row[k] = r
                on = row_ok[k] and x < gw
                packs[k][(j >> 2) & 3][j & 3] = (row[k], x) if on else None
                if j % 4 == 3:
                    g, x0 = (j >> 2) & 3, x - 15
                    if row_ok[k] and (x & 12) == 12 and x0 < gw:
                        for q in range(4):
                            if x0 + 4 * q < gw:
                                for i, tag in enumerate(packs[k][(g + 1 + q) & 3]):
                                    want = (row[k], x0 + 4 * q + i)
                                    assert tag == want, f"stores gw {gw} gh {gh}: {tag} stored at {want}"
                                    assert want not in stored, f"stores gw {gw} gh {gh}: {want} stored twice"
                                    stored[want] = s
    assert len(stored) == gw * gh, f"stores gw {gw} gh {gh}: {len(stored)} of {gw * gh} samples stored"


@pytest.mark.parametrize("gw", [12, 16, 20, 32, 44, 64, 72, 120, 128, 200, 256])
def test_sixteen_sample_blocks_reach_memory_once(gw):
    P, _ = plan(gw)
    for gh in sorted({1, 2, 3, P - 1, P, P + 1, 2 * P + 1, 3 * P + 2} - {0}):
        if gw * gh <= 120_000:
            simulate_wp4_stores(gw, gh)
