"""Lane-level model of the LDS images of gemm_bf16.hip (no GPU): the [k][row] bf16 image with its 16-byte-chunk swizzle, the
per-lane addresses of the transposing reads (ds_read_b64_tr_b16) and of the padded k-fast image (ds_read_b128).

Checked here, for both tile widths: every lane's fragment holds exactly the elements v_mfma_f32_32x32x16_bf16 expects
(lane l: row l & 31, k = 8 (l >> 5) + j of the 16-deep step), and the reads of a 32-lane half are bank-conflict free.
The transposing read is modelled as the CDNA4 ISA describes it: within a group of 16 lanes, lane 4q+p supplies the address
of row q, elements 4p .. 4p+3 of a 4 x 16 block; lane i receives column i of the four rows."""
import numpy as np
import pytest

KT = 64                       # k per tile (gemm_bf16.hip)
KF_STRIDE = KT * 2 + 16       # bytes per row of the k-fast image


def ks_sw(T, row):            # chunk swizzle of a [k][T] image (ks_sw in gemm_bf16.hip)
    return ((row & 3) << 2) if T == 128 else (((row >> 1) & 1) << 2)


def ks_image(X, T):
    """Bytes-as-elements model of the stash: element (k, c) of the tile -> element slot in LDS (2-byte units)."""
    RB = T * 2
    img = np.full(KT * RB // 2, -1, np.int64)
    for k in range(KT):
        for ch in range(T // 8):
            off = k * RB + 16 * (ch ^ ks_sw(T, k))
            img[off // 2: off // 2 + 8] = X[k, 8 * ch: 8 * ch + 8]
    return img


def tr_read(img, addr_of_lane):
    """ds_read_b64_tr_b16: addr_of_lane[64] byte addresses -> (64, 4) elements."""
    out = np.zeros((64, 4), np.int64)
    for grp in range(4):
        base = 16 * grp
        for i in range(16):
            pprime, e = i >> 2, i & 3
            for j in range(4):                       # row j of the block is addressed by lanes 4j .. 4j+3
                a = addr_of_lane[base + 4 * j + pprime]
                assert a % 8 == 0
                out[base + i, j] = img[a // 2 + e]
    return out


@pytest.mark.parametrize("T", [64, 128])
def test_transposing_reads_deliver_the_mfma_fragments_without_bank_conflicts(T):
    RB = T * 2
    rng = np.random.default_rng(T)
    X = rng.permutation(KT * T).reshape(KT, T)       # distinct values: any misplaced element shows
    img = ks_image(X, T)
    assert (img >= 0).all()                          # the swizzle is a bijection on the tile
    WR = T // 2
    for wr in (0, WR):                               # the wave's offset along the tile
        for i in range(WR // 32):                    # its 32-wide MFMA tiles
            for u in range(KT // 16):                # k-steps of the k-tile
                frag = np.zeros((64, 8), np.int64)
                for e in range(2):
                    addr = np.zeros(64, np.int64)
                    for lane in range(64):
                        h, g, q4, p4 = lane >> 5, lane >> 4, (lane & 15) >> 2, lane & 3
                        ch = (wr + 32 * i) // 8 + 2 * (g & 1) + (p4 >> 1)
                        xb = (8 * h + q4) * RB + 16 * (ch ^ ks_sw(T, q4)) + 8 * (p4 & 1)       # lane base (kernel: xb[i])
                        addr[lane] = xb + (16 * u + 4 * e) * RB
                    # bank check: 64 banks of 4 bytes, counted per 32-lane half, 8 bytes per lane
                    for half in range(2):
                        slots = (addr[32 * half: 32 * half + 32] // 8) % 32
                        assert len(set(slots.tolist())) == 32, (T, wr, i, u, e, half)
                    frag[:, 4 * e: 4 * e + 4] = tr_read(img, addr)
                for lane in range(64):
                    h = lane >> 5
                    col = wr + 32 * i + (lane & 31)
                    want = [X[16 * u + 8 * h + j, col] for j in range(8)]
                    assert frag[lane].tolist() == want, (T, wr, i, u, lane)


def test_k_fast_image_rows_cover_all_banks():
    """[row][k] image, rows of 64 k (128 B) padded to 144 B: the ds_read_b128 of 16 consecutive rows (one k half) touches
    64 distinct banks; lane l reads row l & 31, bytes 32 u + 16 (l >> 5)."""
    for u in range(4):
        for h in range(2):
            for r0 in (0, 16):
                banks = set()
                for r in range(r0, r0 + 16):
                    a = r * KF_STRIDE + 32 * u + 16 * h
                    assert a % 16 == 0
                    banks.update(((a // 4) + d) % 64 for d in range(4))
                assert len(banks) == 64, (u, h, r0)


def test_epilogue_transpose_buffer_is_read_back_in_row_order():
    """dX epilogue: lane (column c = l & 31, half h) parks registers 4q .. 4q+3 of MFMA tile i -- rows 8q + 4h .. + 3 -- at
    [c][32 i + 8q + 4h]; lane l then reads 16 bytes (8 bf16) of row l // NCH, piece l % NCH: together the 64 lanes of a pass
    cover whole rows in order."""
    for WR in (32, 64):
        row_elems = WR + 8                           # EPI_ROW = WR * 2 + 16 bytes
        buf = np.full(32 * row_elems, -1, np.int64)
        T = np.arange(WR * 32).reshape(WR, 32)       # T[r][c] of the wave's slab (r along the MFMA rows)
        for lane in range(64):
            c, h = lane & 31, lane >> 5
            for i in range(WR // 32):
                for q in range(4):
                    for k in range(4):
                        r = 32 * i + 8 * q + 4 * h + k
                        buf[c * row_elems + r] = T[r, c]
        NCH = WR * 2 // 16
        rows_pp = 64 // NCH
        seen = np.zeros((32, WR), bool)
        for ps in range(32 // rows_pp):
            for lane in range(64):
                row, piece = lane // NCH + rows_pp * ps, lane % NCH
                vals = buf[row * row_elems + 8 * piece: row * row_elems + 8 * piece + 8]
                assert vals.tolist() == [T[8 * piece + j, row] for j in range(8)]
                seen[row, 8 * piece: 8 * piece + 8] = True
        assert seen.all()


def test_pair_swap_dump_writes_row_major_bf16_rows():
    """Activation dump of the fused training forward (mlp_fused.hip dump_pair): a 32 x 32 accumulator tile has lane l = column
    (sample) l & 31, half h = l >> 5, register i = row (feature) (i & 3) + 8 (i >> 2) + 4 h.  The packed B-operand fragment u
    holds registers 8u .. 8u+7 pairwise (dword q = registers 8u + 2q, 8u + 2q + 1).  v_permlane32_swap(x, y) exchanges x of
    lanes 32..63 with y of lanes 0..31; after swapping dwords (0, 2) and (1, 3) of fragment qp a lane holds eight consecutive
    features and stores them at element offset sample * ld + 32 rt + 8 (2 qp + h)."""
    rt, ld = 3, 256
    tile = np.arange(32 * 32).reshape(32, 32)            # tile[feature_row][sample]
    acc = np.zeros((64, 16), np.int64)
    for lane in range(64):
        for i in range(16):
            acc[lane, i] = tile[(i & 3) + 8 * (i >> 2) + 4 * (lane >> 5), lane & 31]
    mem = np.full((32, ld), -1, np.int64)
    for qp in range(2):
        w = [[(acc[lane, 8 * qp + 2 * q], acc[lane, 8 * qp + 2 * q + 1]) for q in range(4)] for lane in range(64)]

        def swap(x, y):      # x, y: per-lane dword lists -> (x', y')
            x2, y2 = list(x), list(y)
            for lane in range(32):
                x2[lane + 32], y2[lane] = y[lane], x[lane + 32]
            return x2, y2
        r0x, r0y = swap([w[l][0] for l in range(64)], [w[l][2] for l in range(64)])
        r1x, r1y = swap([w[l][1] for l in range(64)], [w[l][3] for l in range(64)])
        for lane in range(64):
            h, sample = lane >> 5, lane & 31
            vals = list(r0x[lane]) + list(r1x[lane]) + list(r0y[lane]) + list(r1y[lane])
            off = 32 * rt + 8 * (2 * qp + h)
            mem[sample, off: off + 8] = vals
    for sample in range(32):
        assert mem[sample, 32 * rt: 32 * rt + 32].tolist() == tile[:, sample].tolist()


def test_chain64_image_serves_the_k_fast_read_and_the_in_place_epilogue():
    """bwd64_chain.hip keeps ONE image of a 128 x 64 bf16 tile ([m][64], 128-byte rows, chunk-swizzled like the [k][64] image
    above): besides the transposing reads (k = m, checked above for T = 64) it is read along a row as the k-fast operand of the
    dX product and rewritten in place by the dX accumulators.  Lane-level model of both:
    * k-fast read: lane l (row m = block + (l & 31), half h = l >> 5) of k-step u reads the 16 bytes at chunk (2u + h) ^ sw(m)
      and must get columns 16u + 8h .. + 7 of its row -- the B operand of v_mfma_f32_32x32x16_bf16;
    * epilogue: accumulator registers 4q .. 4q+3 of lane l hold T[i = wr + 8q + 4h + (0..3)][m = block + (l & 31)]; they go
      to the 8 bytes at row m, chunk ((wr >> 3) + q) ^ sw(m), byte 8h -- every (m, i) of the tile written exactly once, at
      the slot the NEXT layer's reads (both kinds) expect."""
    RT, W, RB = 128, 64, 128

    def sw(row):
        return ((row >> 1) & 1) << 2

    rng = np.random.default_rng(7)
    Z = rng.permutation(RT * W).reshape(RT, W)
    img = np.full(RT * RB // 2, -1, np.int64)
    for m in range(RT):                                      # the stash of bwd64_chain.hip (stash_tile)
        for ch in range(W // 8):
            off = m * RB + 16 * (ch ^ sw(m))
            img[off // 2: off // 2 + 8] = Z[m, 8 * ch: 8 * ch + 8]
    assert (img >= 0).all()
    # k-fast fragments of the dX product
    for wc2 in (0, 64):
        for j in range(2):
            for u in range(W // 16):
                for lane in range(64):
                    h, row = lane >> 5, wc2 + 32 * j + (lane & 31)
                    off = row * RB + 16 * ((2 * u + h) ^ sw(row))
                    assert (img[off // 2: off // 2 + 8] == Z[row, 16 * u + 8 * h: 16 * u + 8 * h + 8]).all()
    # in-place epilogue: where each accumulator element lands, as (row, column) of the image
    new = np.full(RT * RB // 2, -1, np.int64)
    T = rng.permutation(W * RT).reshape(W, RT)               # T[i][m], the dX product of the tile
    for wave in range(4):
        wr, wc2 = (wave >> 1) * 32, (wave & 1) * 64
        for j in range(2):
            for lane in range(64):
                h, row = lane >> 5, wc2 + 32 * j + (lane & 31)
                for q in range(4):
                    off = row * RB + 16 * (((wr >> 3) + q) ^ sw(row)) + 8 * h
                    for e in range(4):                       # register 4q + e <-> T row (i) wr + e + 8q + 4h  (C/D map of the 32x32 MFMA)
                        assert new[off // 2 + e] == -1
                        new[off // 2 + e] = T[wr + 8 * q + 4 * h + e, row]
    assert (new >= 0).all()
    for m in range(RT):                                      # read back as the stash layout: element (m, i) at its swizzled slot
        for ch in range(W // 8):
            off = m * RB + 16 * (ch ^ sw(m))
            assert (new[off // 2: off // 2 + 8] == T[8 * ch: 8 * ch + 8, m]).all()
