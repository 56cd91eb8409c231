"""Host-side frame copies of the NumPy boundary (photonbend_amd/utils/hostcopy.py): CPU-only."""

import numpy as np

from photonbend_amd.utils.hostcopy import par_copy, row_chunks


def test_par_copy_matches_plain_copy_for_small_and_large_frames():
    rng = np.random.default_rng(0)
    for shape in ((3, 5, 3), (40, 17, 3), (700, 1024, 3), (2048, 1024, 3)):
        src = rng.integers(0, 256, shape, dtype=np.uint8)
        dst = np.zeros_like(src)
        par_copy(dst, src)
        assert np.array_equal(dst, src)
        # non-contiguous views (row blocks of a larger buffer) copy as well
        big = np.zeros((shape[0] + 7, shape[1], 3), np.uint8)
        par_copy(big[3 : 3 + shape[0]], src, parts=5)
        assert np.array_equal(big[3 : 3 + shape[0]], src) and not big[:3].any() and not big[3 + shape[0] :].any()


def test_row_chunks_tile_the_rows_exactly():
    for rows, nbytes in ((1, 10), (7, 1 << 20), (4096, 100 << 20), (4096, 50 << 20), (3888, 90 << 20), (5, 1 << 30)):
        chunks = row_chunks(rows, nbytes)
        assert chunks[0][0] == 0 and chunks[-1][1] == rows
        assert all(a < b for a, b in chunks)
        assert all(chunks[i][1] == chunks[i + 1][0] for i in range(len(chunks) - 1))
        assert 1 <= len(chunks) <= 8
    assert len(row_chunks(4096, 100 << 20)) > 1 and len(row_chunks(64, 1 << 20)) == 1
