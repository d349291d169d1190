"""CPU suite: integer identities the HIP kernels rely on, checked exhaustively over the ranges the plan can produce
(numpy restatements of the device expressions; no GPU, no library call)."""
import numpy as np


def test_tile_offset_to_row_by_multiply_high_is_exact():
    """pf::strip_body turns a candidate's tile byte offset k back into (row, x) with row = umulhi(k, ceil(2^32 / tpitch))
    (FusedLevel::tp_recip).  tpitch is a multiple of 16; a tile never holds more than 160 KiB: exact for every k below
    2^18 and every pitch up to 4096."""
    k = np.arange(1 << 18, dtype=np.uint64)
    for tp in range(48, 4097, 16):
        recip = ((1 << 32) + tp - 1) // tp
        assert recip < (1 << 32)
        row = (k * np.uint64(recip)) >> np.uint64(32)
        assert np.array_equal(row, k // np.uint64(tp)), tp


def test_staging_vector_index_to_row_by_multiply_high_is_exact():
    """The staging loops map a 16-byte vector index i < 2^16 to its tile row with umulhi(i, ceil(2^32 / vpr)),
    vpr = tpitch / 16 (FusedLevel::vpr_recip)."""
    i = np.arange(1 << 16, dtype=np.uint64)
    for vpr in range(3, 257):
        recip = ((1 << 32) + vpr - 1) // vpr
        assert np.array_equal((i * np.uint64(recip)) >> np.uint64(32), i // np.uint64(vpr)), vpr


def test_packed_sad_halves_decide_like_the_scalar_form():
    """The prefilter packs the four SADs of a 4-pixel group as {left : up} and {right : down} (v_sad_hi_u8), takes one
    packed 16-bit max and the minimum of the two halves: the same value as min(max(up, down), max(left, right)).  Every
    SAD of four bytes is at most 1020, so the halves never carry into each other."""
    rng = np.random.default_rng(7)
    s = rng.integers(0, 1021, (200000, 4), dtype=np.uint32)          # up, down, left, right
    s[:16] = np.array([[0, 0, 0, 0], [1020, 1020, 1020, 1020], [1020, 0, 0, 1020], [0, 1020, 1020, 0]] * 4, dtype=np.uint32)
    up, down, left, right = s.T
    a, b = (left << 16) + up, (right << 16) + down
    mm = (np.maximum(a >> 16, b >> 16) << 16) | np.maximum(a & 0xffff, b & 0xffff)     # v_pk_max_u16
    lo = np.minimum(mm & 0xffff, mm >> 16)                                             # v_min_u32_sdwa WORD_0, WORD_1
    assert np.array_equal(lo, np.minimum(np.maximum(up, down), np.maximum(left, right)))


def test_nms_byte_compare_trick():
    """Phase D decides "me >= e" for four neighbours at once as bit 7 of v_lerp_u8(me, ~e, 1) = (me + (255 - e) + 1) >> 1
    per byte, and "l >= me" as bit 7 of v_lerp_u8(l, ~me, 1): exhaustive over all byte pairs."""
    me, e = np.meshgrid(np.arange(256, dtype=np.uint32), np.arange(256, dtype=np.uint32), indexing="ij")
    lerp = (me + (255 - e) + 1) >> 1
    assert np.array_equal((lerp >> 7) & 1, (me >= e).astype(np.uint32))


def _lerp(a, b, r):
    """v_lerp_u8 on one byte lane: (a + b + (r & 1)) >> 1."""
    return (a + b + (r & 1)) >> 1


def test_fast_thresholds_as_byte_lerps_are_exact():
    """pdev::fast9_mm: bright  p > c + t  <=>  bit 7 of v_lerp_u8(p, max(255 - t - c, 0), 0)  (no special case for
    c + t >= 255: the addend is 0 there and no p reaches 256);  NOT dark  p >= max(c - t, 0) (Fast.h:63-64)  <=>
    bit 7 of v_lerp_u8(p, min(255 + t - c, 255), 1).  Exhaustive over p, c and every threshold byte."""
    p, c = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    for t in range(256):
        bright = _lerp(p, np.maximum(255 - t - c, 0), 0) >> 7
        assert np.array_equal(bright, (p > c + t).astype(np.int64)), t
        notdark = _lerp(p, np.minimum(255 + t - c, 255), 1) >> 7
        assert np.array_equal(notdark, (p >= np.maximum(c - t, 0)).astype(np.int64)), t
        assert np.array_equal(1 - notdark, (p < c - t).astype(np.int64)), t


def test_byte_domain_pretest_is_a_necessary_condition_one_grey_level_wide():
    """strip_body's pretest works on h = v_lerp_u8(N, ~C, 1) = 128 + floor((N - C) / 2): a compass point counts as bright
    iff h >= min(128 + (t + 1) // 2, 255) and as dark iff NOT h >= 129 - (t + 2) // 2.  Every point the exact test calls
    bright (N > C + t) / dark (N < C - t) must pass (the pretest only FILTERS candidates for the segment test), and a
    point that passes is at most one grey level short of the exact condition.  Exhaustive over N, C, t."""
    n, c = np.meshgrid(np.arange(256, dtype=np.int64), np.arange(256, dtype=np.int64), indexing="ij")
    h = _lerp(n, 255 - c, 1)
    assert np.array_equal(h, 128 + np.floor_divide(n - c, 2))
    for t in range(256):
        kb = min(128 + (t + 1) // 2, 255)
        kd1 = 129 - (t + 2) // 2
        assert 1 <= kd1 <= 128
        bright = (_lerp(h, 255 - kb, 1) >> 7).astype(bool)           # [h >= kb]
        dark = ~(_lerp(h, 255 - kd1, 1) >> 7).astype(bool)           # NOT [h >= kd1]
        assert np.array_equal(bright, h >= kb) and np.array_equal(dark, h < kd1)
        assert not np.any((n > c + t) & ~bright), t                  # necessary
        assert not np.any((n < c - t) & ~dark), t
        # at most one level wider (t = 255: the constant is clamped to a byte, N - C >= 254 passes where no point can be
        # brighter than C + 255 — FAST rejects it)
        assert not np.any(bright & (n - c < t - (t == 255))), t
        assert not np.any(dark & (c - n < t)), t
        if t % 2 == 1:                                               # odd t: the bright side is exact
            assert np.array_equal(bright, n > c + t) or kb == 255
        else:                                                        # even t: the dark side is exact
            assert np.array_equal(dark, n < c - t)


def test_harris_row_pair_byte_selectors():
    """pdev::harris_score_mm joins the column-4/5 bytes of two rows with v_perm_b32(hi, lo, sel) (selector byte k picks
    byte k of {hi : lo}, lo = bytes 0..3): 0x05040100 -> {lo.b0, lo.b1, hi.b0, hi.b1}, 0x07060302 -> the b2 / b3 pairs,
    0x06050201 -> the b1 / b2 pairs — the three operands of the dy chain d_c, d_c+2, d_c+1 for c = 4, 5 of both rows."""
    def perm(hi, lo, sel):
        src = [(lo >> (8 * k)) & 0xff for k in range(4)] + [(hi >> (8 * k)) & 0xff for k in range(4)]
        return sum(src[(sel >> (8 * k)) & 0xff] << (8 * k) for k in range(4))
    lo, hi = 0x44332211, 0x88776655
    assert perm(hi, lo, 0x05040100) == 0x66552211
    assert perm(hi, lo, 0x07060302) == 0x88774433
    assert perm(hi, lo, 0x06050201) == 0x77663322
