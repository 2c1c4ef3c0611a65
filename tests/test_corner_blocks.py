"""The block geometry of the blocked 4-clique gather (csrc/gm_tables.hip ensure_core_tri, csrc/gm_cgather.hip cgatherb_kernel), restated on the
CPU: blocks of consecutive core rows whose stored words (row p keeps the words from (p + 1) >> 5 on) fit the LDS budget; gathering a row of a
vertex's matrix from the block images gives the same bits as gathering it from the row-major bitmap.  CPU only."""
import numpy as np
import pytest

KW, KR = 16384, 1024  # gm_mine.h kCgbWords, kCgbMaxRows


def geometry(h, delta=0):
    """delta = core_base & 31: column bits are counted from the base rounded down to a multiple of 32 (gm_mine.h cgb_first_word / cgb_words)"""
    words = (h + delta + 31) // 32
    bid, rowbase, blk = np.zeros(h, np.int64), np.zeros(h, np.int64), []
    total, p0, used = 0, 0, 0
    for p in range(h + 1):
        span = words - ((p + delta + 1) >> 5) if p < h else 0
        if p == h or used + span > KW or p - p0 >= KR:
            if p > p0:
                blk.append((p0, p - p0, total, used))
                total += (used + 3) & ~3
            p0, used = p, 0
            if p == h:
                break
        bid[p] = len(blk)
        rowbase[p] = used - ((p + delta + 1) >> 5)
        used += span
    return words, bid, rowbase, blk, total


@pytest.mark.parametrize("h,delta", [(64, 0), (100, 7), (777, 31), (4096, 0), (32768, 0), (32768, 19)])
def test_blocks_partition_the_rows_and_fit_the_budget(h, delta):
    words, bid, rowbase, blk, total = geometry(h, delta)
    assert sum(b[1] for b in blk) == h and blk[0][0] == 0
    for k, (p0, n, off, used) in enumerate(blk):
        assert used <= KW and n <= KR and off % 4 == 0
        assert (bid[p0:p0 + n] == k).all()
        spans = [words - ((p + delta + 1) >> 5) for p in range(p0, p0 + n)]
        assert used == sum(spans)
        # a row's words [first, words) land at image offsets [rowbase + first, rowbase + words): inside the image, rows one after the other
        lo = rowbase[p0:p0 + n] + np.array([(p + delta + 1) >> 5 for p in range(p0, p0 + n)])
        assert lo[0] == 0 and (np.diff(lo) == spans[:-1]).all() and lo[-1] + spans[-1] == used


@pytest.mark.parametrize("delta", [0, 13])
def test_rows_gathered_from_the_images_equal_the_bitmap_rows(delta):
    rng = np.random.default_rng(5)
    h = 1500
    words, bid, rowbase, blk, total = geometry(h, delta)
    dense = np.triu(rng.random((h, h)) < 0.1, 1)  # strictly upper triangular adjacency of the core
    tri = np.zeros(total, np.uint32)
    r, c = np.nonzero(dense)
    q = c + delta  # bit position of a column: counted from the base rounded down to a multiple of 32
    off = np.array([blk[bid[p]][2] + rowbase[p] for p in r])
    np.bitwise_or.at(tri, off + (q >> 5), (np.uint32(1) << (q & 31).astype(np.uint32)))
    for _ in range(20):  # a "vertex": a sorted set of core positions; row i of its matrix = the bits of core row s_i at the columns s_j, j > i
        s = np.sort(rng.choice(h, size=int(rng.integers(2, 300)), replace=False))
        for i in rng.choice(len(s), size=min(len(s), 8), replace=False):
            want = dense[s[i], s[i + 1:]]
            off = blk[bid[s[i]]][2] + rowbase[s[i]]
            cols = s[i + 1:] + delta
            got = (tri[off + (cols >> 5)] >> (cols & 31).astype(np.uint32)) & 1
            assert (got.astype(bool) == want).all()
