"""Host-side model of the LDS index algebra of the attention backward kernels (univtg_amd/csrc/attn.hip, round 3).

The kernels derive most fragment offsets from a few per-lane bases (`TileRT::step / rows8 / rows4`, the [key][query] dS image of the
fused kernel, the XCD-aware block decode).  These identities were checked by hand once; this file keeps them checked: it restates the
formulas (it does not parse the kernel), emulates `ds_read_b64_tr_b16` as tools/probe_tr.hip measured it (within a 16-lane group lane i
supplies row i >> 2, columns 4 (i & 3) .. + 3 of a [4][16] block and receives column i), and asserts that every fragment element is the
element the MFMA operand layout needs.  No GPU, no library call."""
import itertools

HD = 128


def off(r, col):                     # TileRT<128, true>::off
    return r * HD + ((((col >> 3) ^ (((r & 3) << 2) | ((r >> 2) & 3)))) << 3) + (col & 7)


def test_swizzled_tile_is_a_bijection_and_steps_are_xors():
    rows = 256
    assert sorted(off(r, c) for r in range(rows) for c in range(HD)) == list(range(rows * HD))
    for r in range(64):
        for g in (0, 1):
            for ks in range(8):      # k-steps of a row fragment: TileRT::step(base, 16 ks)
                assert off(r, 8 * g) ^ (16 * ks) == off(r, 8 * g + 16 * ks)
        for qd, i3, blk in itertools.product((0, 1), range(4), range(4)):      # head-dim blocks of a transposed fragment
            c = 16 * qd + 4 * i3
            assert off(r, c) ^ (32 * blk) == off(r, c + 32 * blk)


def test_row_steps():
    for base in range(0, 256, 16):
        for rr in range(8):          # rows8: + 8 rows from a row with r % 16 < 8
            for c in range(0, HD, 4):
                assert (off(base + rr, c) ^ 16) + 8 * HD == off(base + rr + 8, c)
        for hi, rr in itertools.product((0, 8), range(4)):      # rows4: + 4 rows from a row with r % 8 < 4
            for c in range(0, HD, 4):
                assert (off(base + hi + rr, c) ^ 8) + 4 * HD == off(base + hi + rr + 4, c)
    for wave, l31, g in itertools.product(range(8), range(32), (0, 1)):      # K row of a lane's key = its Q row + 32 wave rows
        assert off(wave * 32 + l31, 8 * g) == off(l31, 8 * g) + wave * 32 * HD


def _tr_read(img, addrs):
    """ds_read_b64_tr_b16: per 16-lane group a [4][16] block; lane i receives column i, rows 0..3."""
    out = {}
    for grp in range(4):
        lanes = list(range(16 * grp, 16 * grp + 16))
        block = {}
        for i, l in enumerate(lanes):
            for e in range(4):
                block[(i >> 2, 4 * (i & 3) + e)] = img[addrs[l] + e]
        for i, l in enumerate(lanes):
            out[l] = [block[(rw, i)] for rw in range(4)]
    return out


def test_fused_kernel_ds_image_round_trip():
    """Every lane (= key) stores the two packed halves of its dK operand; the dQ pass must get B[k = key][n = query] fragments back."""
    img = {}
    for wave, lane in itertools.product(range(8), range(64)):
        g, l31 = lane >> 5, lane & 31
        key = wave * 32 + l31
        dsw = key * 32 + (((key >> 1) & 7) << 2)
        for hf in (0, 1):
            qs = [((8 * hf + e) & 3) + 8 * ((8 * hf + e) >> 2) + 4 * g for e in range(8)]      # accumulator register -> query
            a0, a1 = dsw ^ ((4 * hf + g) << 2), dsw ^ ((4 * hf + 2 + g) << 2)
            for e in range(4):
                assert a0 + e not in img and a1 + e not in img
                img[a0 + e] = (key, qs[e])
                img[a1 + e] = (key, qs[4 + e])
    assert sorted(img) == list(range(256 * 32))
    for khalf, kk in itertools.product((0, 1), range(8)):
        a0, a1 = {}, {}
        for lane in range(64):
            g, i16, qd = lane >> 5, lane & 15, (lane >> 4) & 1
            dsr0 = (khalf * 128 + 8 * g + (i16 >> 2)) * 32 + (((4 * qd + (i16 & 3)) ^ ((4 * g + (i16 >> 3)) & 7)) << 2)
            a0[lane] = dsr0 + 16 * kk * 32
            a1[lane] = (dsr0 ^ 8) + 4 * 32 + 16 * kk * 32
        o0, o1 = _tr_read(img, a0), _tr_read(img, a1)
        for lane in range(64):
            g, l31 = lane >> 5, lane & 31
            for j, (k, q) in enumerate(o0[lane] + o1[lane]):
                assert (k, q) == (khalf * 128 + 16 * kk + 8 * g + j, l31)


def _banks_ok(addrs_el, lanes):      # one 32-lane group of a 64-bit read: all (address / 4) % 64 dwords distinct
    seen = set()
    for l in lanes:
        b = (addrs_el[l] * 2) // 4
        for e in (0, 1):
            if (b + e) % 64 in seen:
                return False
            seen.add((b + e) % 64)
    return True


def test_transposing_reads_spread_over_all_banks():
    lanes = range(64)
    q = {l: off(4 * (l >> 5) + ((l & 15) >> 2), 16 * ((l >> 4) & 1) + 4 * (l & 3)) for l in lanes}
    assert _banks_ok(q, range(32)) and _banks_ok(q, range(32, 64))
    padded = {l: (4 * (l >> 5) + ((l & 15) >> 2)) * 136 + 16 * ((l >> 4) & 1) + 4 * (l & 3) for l in lanes}      # HD + 8 rows
    assert not _banks_ok(padded, range(32))          # (what the swizzle is for: four rows on overlapping banks)


def test_xcd_aware_block_decode_is_a_permutation_that_keeps_heads_together():
    def decode(bid, total, nblk, H):
        q, r, xcd, idx = total // 8, total % 8, bid % 8, bid // 8
        v = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
        return v % nblk, (v // nblk) % H, v // nblk // H
    for nblk, H, B in ((10, 8, 32), (3, 2, 1), (5, 8, 64), (1, 8, 256), (7, 3, 5)):
        total = nblk * H * B
        seen = {}
        for bid in range(total):
            blk, h, b = decode(bid, total, nblk, H)
            assert (blk, h, b) not in seen and blk < nblk and h < H and b < B
            seen[(blk, h, b)] = bid % 8
        assert len(seen) == total
        if total >= 8 * nblk * 2:                    # enough work: at most two XCDs ever share a head (range boundaries)
            for h, b in itertools.product(range(H), range(B)):
                assert len({seen[(blk, h, b)] for blk in range(nblk)}) <= 2
