"""Lane-level model of the LDS images of bwd256_fused.hip (no GPU): ONE image per operand serves a row read and a transposed read.

A 64-sample tile of dZ ([m][W] bf16, rows of 2 W bytes) and of the layer's input X ([m][128] bf16, the workgroup's half of the
input columns) arrive by LDS-DMA: the LDS side of a DMA instruction is linear in the lane index (16 bytes per lane, 1 KB per
instruction), so the chunk swizzle -- chunk c of row r at chunk c ^ f(r), f(r) = (r & 3) << 2 | (r >> 2) & 3 -- is applied to the
GLOBAL address each lane fetches.  Checked here with the kernel's own address arithmetic, for W = 256 and 128:
  * the DMA lane map fills the image with every element exactly once, each lane fetching one whole in-row chunk;
  * dX^T = W^T dZ^T: the B operand is a ds_read_b128 of the dZ image along k = o; lane l gets row m = l & 31 (+ 32 mb), k =
    16 u + 8 (l >> 5) .. + 7, and the sixteen lanes of a quarter wave touch sixteen different 16-byte bank groups;
  * dW = dZ^T X: both operands are ds_read_b64_tr_b16 reads across k = m of the same images; lane l gets row o (or i) = base +
    (l & 31), k = 16 u + 8 (l >> 5) .. + 7, and a 32-lane half touches 32 different 8-byte slots of the 256-byte bank period;
  * the ReLU mask words (8-byte reads of the X image) are the four input columns a lane's accumulator registers 4j .. 4j+3 hold;
  * the dX wave's staging buffer ([64 m][32 i] bf16, chunk j of row m at j ^ (m >> 2) & 3) returns every element to the row-major
    16-byte pieces the flush stores.
The transposing read is modelled as in test_gemm3_layout.py (CDNA4 ISA: within 16 lanes, lane 4q+p addresses row q, elements
4p .. 4p+3 of a 4 x 16 block; lane i receives column i of the four rows)."""
import numpy as np
import pytest

from test_gemm3_layout import tr_read

MT, HI = 64, 128
XROW = HI * 2


def fsw(row):
    return ((row & 3) << 2) | ((row >> 2) & 3)


def dma_image(T, cols, row_bytes, n_waves=8):
    """The image a tile's DMA instructions leave: T[m][c] (distinct values) -> element slots (2-byte units).  Instruction n of the
    image covers LDS bytes [1024 n, 1024 n + 1024); lane l writes bytes 16 l .. + 15 of it and fetches row ROWS n + l / CH, logical
    chunk (l % CH) ^ f(row) (bwd256_kernel: zdma / xdma)."""
    ch_per_row = row_bytes // 16
    rows_per_instr = 1024 // row_bytes
    n_instr = MT * row_bytes // 1024
    img = np.full(MT * row_bytes // 2, -1, np.int64)
    for n in range(n_instr):
        for lane in range(64):
            row = rows_per_instr * n + lane // ch_per_row
            logical = (lane % ch_per_row) ^ fsw(row)
            assert 0 <= logical < ch_per_row and 8 * logical + 8 <= cols
            dst = 1024 * n + 16 * lane
            assert dst == row * row_bytes + 16 * (logical ^ fsw(row))          # = the address every reader computes
            img[dst // 2: dst // 2 + 8] = T[row, 8 * logical: 8 * logical + 8]
    assert n_instr % n_waves == 0                                               # whole instructions per wave
    return img


def banks_distinct(addrs, bytes_per_lane, period=256):
    slots = (np.asarray(addrs) // bytes_per_lane) % (period // bytes_per_lane)
    return len(set(slots.tolist())) == len(addrs)


@pytest.mark.parametrize("W", [256, 128])
def test_dma_images_and_every_fragment_read_of_the_fused_backward_layer(W):
    ZROW = W * 2
    rng = np.random.default_rng(W)
    Z = rng.permutation(MT * W).reshape(MT, W)                  # dZ tile: distinct values, any misplaced element shows
    X = 10 ** 6 + rng.permutation(MT * HI).reshape(MT, HI)      # the half of the input columns this workgroup owns
    zi = dma_image(Z, W, ZROW)
    xi = dma_image(X, HI, XROW)
    assert (zi >= 0).all() and (xi >= 0).all()                   # every slot written exactly once (values are distinct)
    assert sorted(zi.tolist()) == sorted(Z.reshape(-1).tolist()) and sorted(xi.tolist()) == sorted(X.reshape(-1).tolist())

    # ---- dX waves: ds_read_b128 of the dZ image, k = o fast -------------------------------------------------------------
    for mb in range(2):
        for u in range(W // 16):
            addr = np.zeros(64, np.int64)
            for lane in range(64):
                l31, h = lane & 31, lane >> 5
                fz = fsw(l31)                                    # (kernel: f of rows l31 and 32 + l31 alike)
                assert fz == fsw(32 * mb + l31)
                addr[lane] = l31 * ZROW + 32 * mb * ZROW + 16 * ((2 * u + h) ^ fz)
                got = zi[addr[lane] // 2: addr[lane] // 2 + 8].tolist()
                assert got == Z[32 * mb + l31, 16 * u + 8 * h: 16 * u + 8 * h + 8].tolist(), (mb, u, lane)
            for q in range(4):                                   # a b128 read is served a quarter wave at a time
                assert banks_distinct(addr[16 * q: 16 * q + 16], 16), (mb, u, q)

    # ---- dW waves: transposing reads of both images, k = m ---------------------------------------------------------------
    def tr_addrs(row_bytes, chunk0, u, e):
        a = np.zeros(64, np.int64)
        for lane in range(64):
            h, g, q4, p4 = lane >> 5, lane >> 4, (lane & 15) >> 2, lane & 3
            row = 8 * h + q4 + 4 * e                             # (kernel: ztr / xtr bases, + 16 u rows as an immediate)
            sub = 2 * (g & 1) + (p4 >> 1)
            a[lane] = row * row_bytes + 16 * ((chunk0 + sub) ^ fsw(row)) + 8 * (p4 & 1) + 16 * u * row_bytes
            assert fsw(row) == fsw(row + 16 * u)                 # which is why the k-step is an immediate offset
        return a

    for u in range(MT // 16):
        for ob in range(W // 32):                                # A operand: dZ^T rows o = 32 ob + (l & 31)
            frag = np.zeros((64, 8), np.int64)
            for e in range(2):
                base = tr_addrs(ZROW, 4 * (ob & 3), u, e) + 256 * (ob >> 2)      # (kernel: o-blocks ob and ob + 4 are 256 bytes apart)
                for half in range(2):
                    assert banks_distinct(base[32 * half: 32 * half + 32], 8), (ob, u, e, half)
                frag[:, 4 * e: 4 * e + 4] = tr_read(zi, base)
            for lane in range(64):
                h = lane >> 5
                assert frag[lane].tolist() == [Z[16 * u + 8 * h + j, 32 * ob + (lane & 31)] for j in range(8)], (ob, u, lane)
        for ib in range(4):                                      # B operand: X^T rows i = 32 ib + (l & 31)
            frag = np.zeros((64, 8), np.int64)
            for e in range(2):
                base = tr_addrs(XROW, 4 * ib, u, e)
                for half in range(2):
                    assert banks_distinct(base[32 * half: 32 * half + 32], 8), (ib, u, e, half)
                frag[:, 4 * e: 4 * e + 4] = tr_read(xi, base)
            for lane in range(64):
                h = lane >> 5
                assert frag[lane].tolist() == [X[16 * u + 8 * h + j, 32 * ib + (lane & 31)] for j in range(8)], (ib, u, lane)

    # ---- the ReLU mask words of a dX^T accumulator: lane l, registers 4j .. 4j+3 = input columns 32 ib + 8 j + 4 (l >> 5) .. + 3
    for ib in range(4):
        for mb in range(2):
            for j in range(4):
                for lane in range(64):
                    l31, h = lane & 31, lane >> 5
                    a = l31 * XROW + 8 * h + 32 * mb * XROW + 16 * ((4 * ib + j) ^ fsw(l31))
                    i0 = 32 * ib + 8 * j + 4 * h
                    assert xi[a // 2: a // 2 + 4].tolist() == X[32 * mb + l31, i0: i0 + 4].tolist()


def test_staging_buffer_of_a_dx_wave_returns_row_major_pieces():
    """[64 m][32 i] bf16 per dX wave: the accumulator of sample-block mb leaves as 8-byte writes (lane l: row 32 mb + (l & 31),
    columns 8 j + 4 (l >> 5) .. + 3 at chunk j ^ (l31 >> 2) & 3), the flush reads 16-byte pieces (lane l: row l >> 2 + 16 e, logical
    chunk l & 3) and stores them at input columns 8 (l & 3) .. + 7 of that row."""
    buf = np.full(64 * 32, -1, np.int64)
    T = np.arange(64 * 32).reshape(64, 32)                       # T[m][i]: what the accumulators hold
    for mb in range(2):
        for j in range(4):
            addrs = []
            for lane in range(64):
                l31, h = lane & 31, lane >> 5
                a = 32 * mb * 64 + l31 * 64 + 8 * h + 16 * (j ^ ((l31 >> 2) & 3))
                addrs.append(a)
                buf[a // 2: a // 2 + 4] = T[32 * mb + l31, 8 * j + 4 * h: 8 * j + 4 * h + 4]
            assert len(set(addrs)) == 64
    assert (buf >= 0).all()
    for e in range(4):
        for lane in range(64):
            row, fch = (lane >> 2) + 16 * e, lane & 3
            a = row * 64 + 16 * (fch ^ ((row >> 2) & 3))
            assert buf[a // 2: a // 2 + 8].tolist() == T[row, 8 * fch: 8 * fch + 8].tolist()
