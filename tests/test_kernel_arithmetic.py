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
