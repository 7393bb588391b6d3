"""CPU model of the LDS images of the MFMA kernels (no GPU needed): for every fragment read it
enumerates the lanes of each hardware service group and checks that
  * DMA-write (lane-linear destination, permuted SOURCE) followed by the fragment read returns the
    matrix element the MFMA operand needs (the permutation pair is an involution), and
  * every ds_read_b128 service group touches 16 distinct 16-byte slots of the 256-byte bank row
    (MI355X_MICROARCH.md, LDS table: b128 is served in 4 groups of 16 lanes, bank = (addr/4) % 64).
The formulas are the ones in gemm_hls_amd/csrc/mm_mfma_{f32,f64,f16}.hip."""
import itertools

import pytest

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def slots16(addrs):
    """16-byte slot index inside the 256-byte bank row for each byte address."""
    return [(a // 16) % 16 for a in addrs]


def assert_conflict_free_b128(addr_of_lane):
    for grp in B128_GROUPS:
        s = slots16([addr_of_lane(l) for l in grp])
        assert len(set(s)) == 16, (grp, s)


# ---------------------------------------------------------------- fp32 A / B images ------------
@pytest.mark.parametrize("BK", [16, 32])
def test_f32_a_image_roundtrip_and_banks(BK):
    CPR = BK // 4
    SH = 2 if CPR == 4 else 1
    BM = 128
    # DMA: slot s (16 B) <- A[row = s // CPR][logical chunk = (s % CPR) ^ swz(row)]
    lds = {}
    for s in range(BM * CPR):
        row, pc = divmod(s, CPR)
        lc = pc ^ ((row >> SH) & (CPR - 1))
        lds[s * 16] = (row, lc)  # what lives at this byte address
    for wm_rows, mi, kg in itertools.product([0, 64], [0, 1], range(BK // 8)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            a_swz = hi ^ ((lo >> SH) & (CPR - 1))
            return (wm_rows + lo) * BK * 4 + mi * 32 * BK * 4 + ((2 * kg) ^ a_swz) * 16
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm_rows + mi * 32 + (l & 31)
            assert lc == 2 * kg + (l >> 5)           # k = 8*kg + 4*(lane>>5) + 0..3
        assert_conflict_free_b128(addr)


def test_f32_b_image_banks():
    BN = 256
    for kg, p, wn in itertools.product(range(4), range(4), range(2)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            return (4 * hi) * BN * 4 + (wn * 128 + 4 * lo) * 4 + (kg * 8 + p) * BN * 4
        assert_conflict_free_b128(addr)
        for l in range(64):  # element addressed = B[k][col] row-major, untouched by the DMA
            k, col = divmod(addr(l) // 4, BN)
            assert k == kg * 8 + p + 4 * (l >> 5) and col == wn * 128 + 4 * (l & 31)


def test_f32_accumulation_covers_every_k_once():
    # inside an 8-deep group MFMA p multiplies k = p (lanes < 32) and k = p + 4 (lanes >= 32)
    ks = sorted(p + 4 * hi for p in range(4) for hi in range(2))
    assert ks == list(range(8))


# ---------------------------------------------------------------- fp64 -------------------------
def test_f64_a_image_roundtrip_and_banks():
    CPR, BK, BM = 8, 16, 256
    lds = {}
    for s in range(BM * CPR):
        row, pc = divmod(s, CPR)
        lds[s * 16] = (row, pc ^ ((row >> 1) & 7))
    for wm, mi, kg in itertools.product(range(4), range(4), range(2)):
        def addr(l):
            lo, g4 = l & 15, l >> 4
            a_swz = (lo >> 1) & 7
            return (wm * 64 + lo) * BK * 8 + mi * 16 * BK * 8 + (((4 * kg) + g4) ^ a_swz) * 16
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 64 + mi * 16 + (l & 15)
            assert lc == 4 * kg + (l >> 4)           # doubles 2*lc, 2*lc+1 -> k = 8kg + 2*g4 + p
        assert_conflict_free_b128(addr)


def test_f64_b_image_banks():
    BN = 128
    for kg, p, pr, wn in itertools.product(range(2), range(2), range(2), range(2)):
        def addr(l):
            lo, g4 = l & 15, l >> 4
            return (2 * g4) * BN * 8 + (wn * 64 + 2 * lo) * 8 + (kg * 8 + p) * BN * 8 + pr * 32 * 8
        assert_conflict_free_b128(addr)
        for l in range(64):
            k, col = divmod(addr(l) // 8, BN)
            assert k == kg * 8 + 2 * (l >> 4) + p and col == wn * 64 + pr * 32 + 2 * (l & 15)
    assert sorted(2 * g + p for p in range(2) for g in range(4)) == list(range(8))


# ---------------------------------------------------------------- fp16 B (transpose reads) ------
def test_f16_b_image_transpose_read_gather_and_banks():
    BN, BROW, BCH = 256, 512, 32
    # DMA: slot -> (k row, physical chunk); holds logical chunk pc ^ ((k&3)<<2)
    lds = {}
    for kr in range(64):
        for pc in range(BCH):
            lds[kr * BROW + pc * 16] = (kr, pc ^ ((kr & 3) << 2))

    def element_at(byte_addr):  # (k, column) of the half stored at this LDS byte address
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 8 + (byte_addr - base) // 2

    for ks, h, ni, wn in itertools.product(range(4), range(2), range(4), range(2)):
        def addr(l):
            x, gq, hi = l & 15, (l >> 4) & 1, l >> 5
            r = x >> 2
            lane_base = (8 * hi + r) * BROW + (wn * 16 + 2 * gq + ((x & 3) >> 1)) * 16 + (x & 1) * 8
            return lane_base + (ni ^ r) * 64 + ks * 16 * BROW + h * 4 * BROW
        # hardware transpose inside each 16-lane group: out[i][j] = in[4j + (i>>2)][i&3]
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(4):
                src_lane = grp + 4 * j + (i >> 2)
                k, col = element_at(addr(src_lane) + 2 * (i & 3))
                assert k == ks * 16 + 8 * (l >> 5) + 4 * h + j
                assert col == wn * 128 + ni * 32 + (l & 31)
        # b64 reads are served per 32-lane half, bank = (addr/4) % 64: 32 lanes x 8 B must be disjoint
        for half in (range(0, 32), range(32, 64)):
            banks = set()
            for l in half:
                for d in (0, 4):
                    banks.add(((addr(l) + d) // 4) % 64)
            assert len(banks) == 64


def test_i8_b_image_transpose_read_gather_and_banks():
    BROW, BCH = 256, 16
    lds = {}
    for kr in range(128):
        for pc in range(BCH):
            lds[kr * BROW + pc * 16] = (kr, pc ^ ((kr & 7) << 1))

    def element_at(byte_addr):
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 16 + (byte_addr - base)

    for ks, h, ni, wn in itertools.product(range(4), range(2), range(4), range(2)):
        def addr(l):
            y, gq, hi = l & 15, (l >> 4) & 1, l >> 5
            r, q = y >> 1, y & 1
            return (16 * hi + r) * BROW + 8 * q + ((((wn * 4 + ni) ^ r) * 2) + gq) * 16 + ks * 32 * BROW + h * 8 * BROW
        # hardware: out[i][j] = in[2j + (i>>3)][i&7] inside each 16-lane group
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(8):
                src_lane = grp + 2 * j + (i >> 3)
                k, col = element_at(addr(src_lane) + (i & 7))
                assert k == ks * 32 + 16 * (l >> 5) + 8 * h + j      # MFMA operand byte 8h+j <-> k = 16*(l>>5) + 8h + j
                assert col == wn * 128 + ni * 32 + (l & 31)
        for half in (range(0, 32), range(32, 64)):
            banks = set()
            for l in half:
                for d in (0, 4):
                    banks.add(((addr(l) + d) // 4) % 64)
            assert len(banks) == 64


def test_xcd_remap_is_a_bijection():
    def remap(bid, nwg):
        q, r = divmod(nwg, 8)
        xcd, slot = bid % 8, bid // 8
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        return base + slot
    for nwg in [1, 7, 8, 9, 63, 64, 65, 1000, 8192, 8191]:
        assert sorted(remap(b, nwg) for b in range(nwg)) == list(range(nwg))


# ================================================================ round 2: ping-pong kernels =======
def b64_half_banks_disjoint(addr_of_lane, lanes):
    """ds_read_b64 / ds_read_b64_tr_*: served per 32-lane half, bank = (addr/4) % 64; identical addresses
    broadcast, distinct addresses must not share a bank."""
    seen = {}
    for l in lanes:
        a = addr_of_lane(l)
        for d in (0, 4):
            bank = ((a + d) // 4) % 64
            assert seen.setdefault(bank, a) == a, (l, a, seen[bank])


def test_pingpong_a_image_64_byte_rows_roundtrip_and_banks():
    """mfma_f16_pp_kernel / mfma_i8_pp_kernel / mfma_f32_pp_kernel / valu_tile_dma_kernel: A slab
    [rows][64 B], DMA piece = 16 rows, source chunk = pc ^ ((row>>2)&3); fragment chunk 2*ks + hi."""
    lds = {}
    for piece in range(16):
        for lane in range(64):
            row, pc = piece * 16 + lane // 4, lane % 4
            lds[piece * 1024 + lane * 16] = (row, pc ^ ((row >> 2) & 3))
    for wm, mi, ks in itertools.product(range(2), range(4), range(2)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            ca = hi ^ ((lo >> 2) & 3)
            return (wm * 128 + lo) * 64 + ((ca, ca ^ 2)[ks]) * 16 + mi * 32 * 64
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 128 + mi * 32 + (l & 31) and lc == 2 * ks + (l >> 5)
        assert_conflict_free_b128(addr)


def test_pingpong_a_image_full_line_rows_roundtrip_and_banks():
    """mfma_f16_pp2_kernel / mfma_i8_pp2_kernel: A double slab [256 rows][128 B], DMA piece = 8 rows,
    source chunk = pc ^ ((row>>1)&7); fragment chunk (4*h + 2*ks + hi) for slab parity h."""
    lds = {}
    for piece in range(32):
        for lane in range(64):
            row, pc = piece * 8 + lane // 8, lane % 8
            lds[piece * 1024 + lane * 16] = (row, pc ^ ((row >> 1) & 7))
    for wm, mi, h, ks in itertools.product(range(2), range(4), range(2), range(2)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            ca = hi ^ ((lo >> 1) & 7)
            return (wm * 128 + lo) * 128 + ((2 * (2 * h + ks)) ^ ca) * 16 + mi * 32 * 128
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 128 + mi * 32 + (l & 31) and lc == 4 * h + 2 * ks + (l >> 5)
        assert_conflict_free_b128(addr)


def test_pingpong_f16_b_image_for_64_column_waves():
    """B slab [32 k][256 cols] halves, DMA piece = 2 k-rows, source chunk = pb ^ ((k&3)<<2); 8 waves
    as 2 x 4 -> a wave covers 64 columns: logical chunk = wn*8 + ni*4 + 2*gq + ((x&3)>>1)."""
    BROW = 512
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 2 + lane // 32, lane % 32
            lds[piece * 1024 + lane * 16] = (kr, pb ^ ((kr & 3) << 2))

    def element_at(byte_addr):
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 8 + (byte_addr - base) // 2

    for ks, h, ni, wn in itertools.product(range(2), range(2), range(2), range(4)):
        def addr(l):
            x, gq, hi = l & 15, (l >> 4) & 1, l >> 5
            r = x >> 2
            logical = wn * 8 + ni * 4 + 2 * gq + ((x & 3) >> 1)
            return (8 * hi + r) * BROW + (logical ^ (r << 2)) * 16 + (x & 1) * 8 + ks * 16 * BROW + h * 4 * BROW
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(4):   # hardware transpose: out[i][j] = in[4j + (i>>2)][i&3]
                k, col = element_at(addr(grp + 4 * j + (i >> 2)) + 2 * (i & 3))
                assert k == ks * 16 + 8 * (l >> 5) + 4 * h + j and col == wn * 64 + ni * 32 + (l & 31)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


def test_pingpong_i8_b_image_for_64_column_waves():
    BROW = 256
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 4 + lane // 16, lane % 16
            lds[piece * 1024 + lane * 16] = (kr, pb ^ ((kr & 7) << 1))

    def element_at(byte_addr):
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 16 + (byte_addr - base)

    for ks, h, ni, wn in itertools.product(range(2), range(2), range(2), range(4)):
        def addr(l):
            y, gq, hi = l & 15, (l >> 4) & 1, l >> 5
            r, q = y >> 1, y & 1
            return (16 * hi + r) * BROW + 8 * q + ((((wn * 2 + ni) ^ r) * 2) + gq) * 16 + ks * 32 * BROW + h * 8 * BROW
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(8):   # out[i][j] = in[2j + (i>>3)][i&7]
                k, col = element_at(addr(grp + 2 * j + (i >> 3)) + (i & 7))
                assert k == ks * 32 + 16 * (l >> 5) + 8 * h + j and col == wn * 64 + ni * 32 + (l & 31)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


def test_valu_tile_dma_a_pairs_roundtrip_and_banks():
    """valu_tile_dma_kernel: A slab [128 rows][16 k] floats; a thread reads the (k, k+1) pair of each of
    its 8 rows with ds_read_b64; the four ty values of a wave must land on different bank groups."""
    lds = {}
    for piece in range(8):
        for lane in range(64):
            row, pc = piece * 16 + lane // 4, lane % 4
            lds[piece * 1024 + lane * 16] = (row, pc ^ ((row >> 2) & 3))
    for wave, kk, i in itertools.product(range(4), range(0, 16, 2), range(8)):
        def addr(l):
            ty = (wave * 64 + l) // 16
            r = ty * 4 + i if i < 4 else 64 + ty * 4 + (i - 4)
            return r * 64 + (((kk >> 2) ^ (ty & 3)) * 16) + (kk & 3) * 4
        for l in range(64):
            ty = (wave * 64 + l) // 16
            a = addr(l)
            row, lc = lds[(a // 16) * 16]
            assert row == (ty * 4 + i if i < 4 else 64 + ty * 4 + (i - 4))
            assert lc * 4 + (a % 16) // 4 == kk                        # first element of the pair is k = kk
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


# ---------------------------------------------------------------- 16 x 16 matrix instructions (round 3) ------------
def test_pingpong_16x16_a_operand_on_full_line_rows_roundtrip_and_banks():
    """mfma_f16_pp2s_kernel / mfma_i8_pp2s_kernel: the A double slab of the full-line kernels, unchanged
    ([256 rows][128 B], source chunk = pc ^ ((row>>1)&7)), read for the 16x16x32 (16x16x64) operand: lane l takes row
    l&15 of a 16-row block and the 16-byte chunk 4*H + (l>>4) (H = slab parity inside the double slab)."""
    lds = {}
    for piece in range(32):
        for lane in range(64):
            row, pc = piece * 8 + lane // 8, lane % 8
            lds[piece * 1024 + lane * 16] = (row, pc ^ ((row >> 1) & 7))
    for wm, rb, H in itertools.product(range(2), range(8), range(2)):
        def addr(l):
            l15, g = l & 15, l >> 4
            return (wm * 128 + l15) * 128 + ((4 * H + g) ^ (l15 >> 1)) * 16 + rb * 16 * 128
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 128 + rb * 16 + (l & 15) and lc == 4 * H + (l >> 4)   # k = 32*H + 8*(l>>4) .. +7 (halves)
        assert_conflict_free_b128(addr)


def test_pingpong_16x16x32_f16_b_image_roundtrip_and_banks():
    """B slab [32 k][256 cols] halves for the 16x16x32 operand: the two 16-lane groups of a half-wave differ in k by 8
    (not in column by 16), so the source chunk is pb ^ ((k&3)<<2) ^ (((k>>3)&1)<<1); lane group g = l>>4 gathers
    k = 8g .. 8g+7 of 16 columns with two transpose reads."""
    BROW = 512
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 2 + lane // 32, lane % 32
            lds[piece * 1024 + lane * 16] = (kr, pb ^ ((kr & 3) << 2) ^ (((kr >> 3) & 1) << 1))

    def element_at(byte_addr):
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 8 + (byte_addr - base) // 2

    for h2, nb, wn in itertools.product(range(2), range(4), range(4)):
        def addr(l):
            l15, g = l & 15, l >> 4
            r, piece = l15 >> 2, l15 & 3
            logical = wn * 8 + nb * 2 + (piece >> 1)
            return (8 * g + r) * BROW + (logical ^ (r << 2) ^ ((g & 1) << 1)) * 16 + (piece & 1) * 8 + h2 * 4 * BROW
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(4):   # hardware transpose: out[i][j] = in[4j + (i>>2)][i&3]
                k, col = element_at(addr(grp + 4 * j + (i >> 2)) + 2 * (i & 3))
                assert k == 8 * (l >> 4) + 4 * h2 + j and col == wn * 64 + nb * 16 + (l & 15)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


def test_pingpong_16x16x64_i8_b_image_roundtrip_and_banks():
    """B slab [64 k][256 cols] bytes for the 16x16x64 operand: source chunk = pb ^ (((k&7)<<1) | ((k>>4)&1)); lane group
    g gathers k = 16g .. 16g+15 of 16 columns with two 8-bit transpose reads."""
    BROW = 256
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 4 + lane // 16, lane % 16
            lds[piece * 1024 + lane * 16] = (kr, pb ^ (((kr & 7) << 1) | ((kr >> 4) & 1)))

    def element_at(byte_addr):
        base = (byte_addr // 16) * 16
        kr, lc = lds[base]
        return kr, lc * 16 + (byte_addr - base)

    for h2, nb, wn in itertools.product(range(2), range(4), range(4)):
        def addr(l):
            l15, g = l & 15, l >> 4
            r, q = l15 >> 1, l15 & 1
            return (16 * g + r) * BROW + 8 * q + ((wn * 4 + nb) ^ ((r << 1) | (g & 1))) * 16 + h2 * 8 * BROW
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(8):   # out[i][j] = in[2j + (i>>3)][i&7]
                k, col = element_at(addr(grp + 2 * j + (i >> 3)) + (i & 7))
                assert k == 16 * (l >> 4) + 8 * h2 + j and col == wn * 64 + nb * 16 + (l & 15)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


def test_pingpong_16x16_a_operand_on_64_byte_rows_roundtrip_and_banks():
    """pingpong_k32 / pingpong_k64 (round 3): A slab [256 rows][64 B], DMA piece = 16 rows, source chunk =
    pc ^ (-(row>>2))&3; the 16x16 operand read takes row l&15 and chunk l>>4.  (The (row>>2)&3 swizzle of the 32x32
    kernels would put rows 0-3 / chunk 0 and rows 4-7 / chunk 1 of one service group into the same slots.)"""
    lds = {}
    for piece in range(16):
        for lane in range(64):
            row, pc = piece * 16 + lane // 4, lane % 4
            lds[piece * 1024 + lane * 16] = (row, pc ^ ((0 - (row >> 2)) & 3))
    for wm, rb in itertools.product(range(2), range(8)):
        def addr(l):
            l15, g = l & 15, l >> 4
            return (wm * 128 + l15) * 64 + (g ^ ((0 - (l15 >> 2)) & 3)) * 16 + rb * 16 * 64
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 128 + rb * 16 + (l & 15) and lc == (l >> 4)
        assert_conflict_free_b128(addr)


def test_pingpong_16x16_kxn_a_gather_f16_and_i8():
    """K x N A on the 16x16 ping-pong kernels: the A slab is staged [k][256 rows] with B's swizzle and the operand
    (row l&15 of block rb, k = 8g.. / 16g..) is gathered by the same pair of transpose reads as B's."""
    # f16: [32 k][256 rows] halves
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 2 + lane // 32, lane % 32
            lds[piece * 1024 + lane * 16] = (kr, pb ^ ((kr & 3) << 2) ^ (((kr >> 3) & 1) << 1))
    for h2, rb, wm in itertools.product(range(2), range(8), range(2)):
        def addr(l):
            l15, g = l & 15, l >> 4
            r, piece = l15 >> 2, l15 & 3
            xk = (r << 2) ^ ((g & 1) << 1)
            return (8 * g + r) * 512 + ((wm * 16 + rb * 2 + (piece >> 1)) ^ xk) * 16 + (piece & 1) * 8 + h2 * 4 * 512
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(4):
                a = addr(grp + 4 * j + (i >> 2)) + 2 * (i & 3)
                kr, lc = lds[(a // 16) * 16]
                assert kr == 8 * (l >> 4) + 4 * h2 + j and lc * 8 + (a % 16) // 2 == wm * 128 + rb * 16 + (l & 15)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))
    # i8: [64 k][256 rows] bytes
    lds = {}
    for piece in range(16):
        for lane in range(64):
            kr, pb = piece * 4 + lane // 16, lane % 16
            lds[piece * 1024 + lane * 16] = (kr, pb ^ (((kr & 7) << 1) | ((kr >> 4) & 1)))
    for h2, rb, wm in itertools.product(range(2), range(8), range(2)):
        def addr(l):
            l15, g = l & 15, l >> 4
            r, q = l15 >> 1, l15 & 1
            return (16 * g + r) * 256 + 8 * q + ((wm * 8 + rb) ^ ((r << 1) | (g & 1))) * 16 + h2 * 8 * 256
        for l in range(64):
            i, grp = l & 15, l & ~15
            for j in range(8):
                a = addr(grp + 2 * j + (i >> 3)) + (i & 7)
                kr, lc = lds[(a // 16) * 16]
                assert kr == 16 * (l >> 4) + 8 * h2 + j and lc * 16 + a % 16 == wm * 128 + rb * 16 + (l & 15)
        b64_half_banks_disjoint(addr, range(0, 32))
        b64_half_banks_disjoint(addr, range(32, 64))


# ---------------------------------------------------------------- fp32 64 x 64 geometry (round 3) ------------
# gemm_hls_amd/csrc/mm_mfma_f32_small.inc: mfma_f32_small_kernel.  A slab [64 rows][8 chunks of 4 k], chunk ^ (row >> 1) & 7;
# B slab [32 k][16 chunks of 4 columns], chunk ^ 8 for k & 4, read one float per lane (ds_read_b32 / ds_read2st64_b32).
def test_f32_small_a_image_roundtrip_and_banks():
    lds = {}
    for s in range(64 * 8):                       # DMA: 16-B slot s <- A[row = s >> 3][logical chunk (s & 7) ^ swz(row)]
        row, pc = s >> 3, s & 7
        lds[s * 16] = (row, pc ^ ((row >> 1) & 7))
    assert len({v for v in lds.values()}) == 64 * 8          # a permutation of the slab
    for wm, kg in itertools.product(range(2), range(4)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            row = wm * 32 + lo
            return row * 128 + (((2 * kg) ^ (hi ^ ((row >> 1) & 7))) * 16)
        for l in range(64):
            row, lc = lds[addr(l)]
            assert row == wm * 32 + (l & 31) and lc == 2 * kg + (l >> 5)    # k = 8 kg + 4 (lane >> 5) + 0..3
        assert_conflict_free_b128(addr)


def test_f32_small_b_image_roundtrip_and_banks():
    lds = {}
    for s in range(32 * 16):                      # DMA: slot s <- B[k = s >> 4][column chunk (s & 15) ^ (8 if k & 4)]
        k, pc = s >> 4, s & 15
        lds[s * 16] = (k, pc ^ ((k & 4) << 1))
    assert len({v for v in lds.values()}) == 32 * 16
    for wn, kg, p in itertools.product(range(2), range(4), range(4)):
        def addr(l):
            lo, hi = l & 31, l >> 5
            return (4 * hi) * 256 + ((((wn * 32 + lo) >> 2) ^ (8 * hi)) * 16) + (lo & 3) * 4 + (kg * 8 + p) * 256
        banks = set()
        for l in range(64):
            a = addr(l)
            k, chunk = lds[a - a % 16]
            assert k == kg * 8 + p + 4 * (l >> 5)                            # MFMA p: k = p (lanes < 32), p + 4 (lanes >= 32)
            assert chunk * 4 + (a % 16) // 4 == wn * 32 + (l & 31)           # column lane & 31 of the wavefront's 32
            banks.add((a // 4) % 64)
        assert len(banks) == 64                                              # one dword per bank: the 64 lanes in one pass
