"""A CPU model of the SCHEDULE of the lane-packed predictor kernels (csrc/modular.hip, predict_lanes_kernel /
predict_lanes_narrow_kernel): which lane produces which sample at which step, which LDS ring slot it lands in, and which
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
