"""CPU tests of the column-sorted pass layout (scs-python_amd/csrc/spmv_cs.hpp) that the large-matrix SpMV kernels
K1/K2 read: the host builder + a host walk of the layout in the kernel's order (scs_hip_cs_layout_host_spmv, no GPU)
must reproduce the oracle's sequential per-row sums bit for bit — every chunk geometry, ragged and empty rows,
padding passes, patterns the bit fields cannot hold."""
import numpy as np
import pytest
from scipy import sparse

import problem_gen as pg


@pytest.fixture(scope="module")
def hip():
    from scs import _scs_hip
    return _scs_hip


@pytest.fixture(scope="module")
def oracle():
    from oracle import scs_oracle
    return scs_oracle


@pytest.mark.parametrize("shape,per_col,rpt", [
    ((3000, 1700), 6, 0), ((3000, 1700), 6, 1), ((70000, 20000), 5, 0), ((70000, 20000), 5, 2),
    ((40000, 90000), 3, 4), ((20000, 30000), 9, 8), ((33000, 9000), 4, 16), ((100, 5), 2, 0),
])
def test_layout_walk_matches_oracle_bits(hip, oracle, shape, per_col, rpt):
    rng = np.random.default_rng(41)
    A = pg.random_sparse(*shape, per_col, rng)
    x, y = rng.standard_normal(shape[1]), rng.standard_normal(shape[0])
    got = hip.cs_layout_host_spmv(A, x, rpt=rpt)
    assert got is not None
    np.testing.assert_array_equal(got, oracle.spmv(A, x))
    got_t = hip.cs_layout_host_spmv(A, y, transpose=True, rpt=rpt)
    assert got_t is not None
    np.testing.assert_array_equal(got_t, oracle.spmv(A, y, trans=True))


@pytest.mark.parametrize("split", [2, 4])
@pytest.mark.parametrize("shape,per_col", [((70000, 20000), 5), ((3000, 1700), 6), ((20000, 50000), 12)])
def test_layout_split_workgroups_per_chunk(hip, oracle, shape, per_col, split):
    """split = 2, 4 (what scs_init uses for the large matrices): every part of a chunk's column-sorted stream is summed
    on its own and the partial sums are added in part order — not the oracle's sequential order any more, but within
    1 ulp-ish of it"""
    rng = np.random.default_rng(43)
    A = pg.random_sparse(*shape, per_col, rng)
    x, y = rng.standard_normal(shape[1]), rng.standard_normal(shape[0])
    for trans, vec in ((False, x), (True, y)):
        got = hip.cs_layout_host_spmv(A, vec, transpose=trans, split=split)
        ref = oracle.spmv(A, vec, trans=trans)
        assert got is not None
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        assert (got != ref).any() or A.nnz < 100000  # it really is a different summation order


def test_layout_empty_rows_and_columns(hip, oracle):
    rng = np.random.default_rng(3)
    A = pg.random_sparse(5000, 4000, 2, rng).tolil()
    A[100:900, :] = 0.0      # empty rows
    A[:, 50:700] = 0.0       # empty columns
    A = sparse.csc_matrix(A)
    A.eliminate_zeros()
    A.sort_indices()
    x, y = rng.standard_normal(4000), rng.standard_normal(5000)
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(A, x), oracle.spmv(A, x))
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(A, y, transpose=True), oracle.spmv(A, y, trans=True))
    Z = sparse.csc_matrix((300, 200))
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(Z, np.ones(200)), np.zeros(300))


def _with_dense_block(rng, shape, per_col, k):
    A = pg.random_sparse(*shape, per_col, rng)
    ii, jj = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
    B = (A + sparse.csc_matrix((rng.standard_normal(k * k), (ii.ravel(), jj.ravel())), shape=A.shape)).tocsc()
    B.sort_indices()
    return B


def test_rows_too_long_for_the_count_fields_are_peeled(hip, oracle, monkeypatch):
    """8 and 16 rows per lane: 6-bit counts per (row, pass).  A dense 70 x 70 block puts 70 nonzeros of a row into one
    pass: those rows are peeled off the layout and summed from the plain CSR (sequential order: still the oracle's
    bits), the rest of the matrix keeps the column-sorted passes.  SCS_HIP_CS_PEEL=0 restores round 1's behaviour —
    the whole matrix is refused."""
    rng = np.random.default_rng(5)
    B = _with_dense_block(rng, (40000, 30000), 4, 70)
    x, y = rng.standard_normal(30000), rng.standard_normal(40000)
    for rpt in (16, 8, 4, 0):
        for split in (1, 2):
            got = hip.cs_layout_host_spmv(B, x, rpt=rpt, split=split)
            assert got is not None
            if split == 1:
                np.testing.assert_array_equal(got, oracle.spmv(B, x))
            else:
                np.testing.assert_allclose(got, oracle.spmv(B, x), rtol=0, atol=1e-13 * np.abs(got).max())
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(B, y, transpose=True, rpt=16), oracle.spmv(B, y, trans=True))
    if not hip.labs_build():
        return  # (refusing such patterns as round 1 did is a switch of the labs build)
    monkeypatch.setenv("SCS_HIP_CS_PEEL", "0")
    assert hip.cs_layout_host_spmv(B, x, rpt=16) is None
    assert hip.cs_layout_host_spmv(B, x, rpt=8) is None
    assert hip.cs_layout_host_spmv(B, x, rpt=4) is not None      # 12-bit counts hold it


def test_power_law_and_banded_patterns(hip, oracle):
    """skewed row lengths (a few rows with thousands of nonzeros, Zipf-like) and a banded matrix: the layout is kept
    (long rows peeled), bit for bit the oracle's sums for rows up to 2048 nonzeros"""
    rng = np.random.default_rng(8)
    m, n = 60000, 50000
    lens = np.minimum((rng.pareto(1.2, m) * 3 + 1).astype(np.int64), 6000)
    rows = np.repeat(np.arange(m), lens)
    cols = rng.integers(0, n, size=rows.size)
    A = sparse.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    assert np.diff(A.tocsr().indptr).max() > 2048
    x = rng.standard_normal(n)
    got, ref = hip.cs_layout_host_spmv(A, x), oracle.spmv(A, x)
    assert got is not None
    short = np.diff(A.tocsr().indptr) <= 2048
    np.testing.assert_array_equal(got[short], ref[short])
    np.testing.assert_allclose(got[~short], ref[~short], rtol=1e-12, atol=1e-12)
    # banded: 21 diagonals
    offs = np.arange(-10, 11)
    Bd = sparse.diags([rng.standard_normal(40000 - abs(o)) for o in offs], offs, shape=(40000, 40000), format="csc")
    Bd.sort_indices()
    xb = rng.standard_normal(40000)
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(Bd, xb), oracle.spmv(Bd, xb))
    np.testing.assert_array_equal(hip.cs_layout_host_spmv(Bd, xb, transpose=True), oracle.spmv(Bd, xb, trans=True))


def test_layout_rejects_very_wide_sparse_chunks(hip):
    rng = np.random.default_rng(5)
    # very wide and very sparse: every pass is cut at 2^19 columns => mostly padding => rejected
    W = pg.random_sparse(20000, 3000000, 1, rng)
    assert hip.cs_layout_host_spmv(W, np.ones(W.shape[1]), rpt=0) is None


def test_long_rows_as_pieces_in_the_passes(hip, oracle):
    """Round 3: a row too long for the count fields keeps an empty slot and its k-th nonzero goes to piece k mod np; the pieces are
    extra row slots of every chunk, their sums are added per row by one wavefront (64 lanes striding, shuffle tree).  Walked on the
    host exactly as the kernels do it: rows that stay whole keep the oracle's bits, cut rows agree to the rounding of the tree;
    every piece length that builds gives the same answer to that accuracy; nothing to cut -> None."""
    rng = np.random.default_rng(11)
    m, n = 60000, 50000
    lens = np.minimum((rng.pareto(1.2, m) * 3 + 1).astype(np.int64), 9000)
    rows = np.repeat(np.arange(m), lens)
    cols = rng.integers(0, n, size=rows.size)
    A = sparse.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, n))
    A.sum_duplicates()
    A.sort_indices()
    rl = np.diff(A.tocsr().indptr)
    assert rl.max() > 4000
    x, y = rng.standard_normal(n), rng.standard_normal(m)
    ref = oracle.spmv(A, x)
    whole = rl <= 4095  # 60000 rows -> 256 rows per chunk, one row per lane: 13-bit counts, rows up to 2048 ... 4095 may stay whole
    for piece_len in (6, 24, 100):
        got = hip.cs_layout_host_spmv_pieces(A, x, piece_len=piece_len)
        assert got is not None
        short = rl <= 2048
        np.testing.assert_array_equal(got[short], ref[short])
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
        assert (got[~whole] != 0).all()
    # the transposed product has no long rows (columns are uniform): nothing to cut
    assert np.diff(A.indptr).max() < 64
    assert hip.cs_layout_host_spmv_pieces(A, y, transpose=True, piece_len=24) is None
    # a dense row and a dense column at 8 rows per lane (6-bit counts): both orientations are cut
    nb = 2400000
    rb = np.concatenate([rng.integers(0, 300000, nb), np.full(40000, 17), np.arange(300000)])
    cb = np.concatenate([rng.integers(0, 40000, nb), np.arange(40000), np.full(300000, 3)])
    B = sparse.csc_matrix((rng.standard_normal(rb.size), (rb, cb)), shape=(300000, 40000))
    B.sum_duplicates()
    B.sort_indices()
    xb, yb = rng.standard_normal(40000), rng.standard_normal(300000)
    for tr, v in ((False, xb), (True, yb)):
        got, refb = hip.cs_layout_host_spmv_pieces(B, v, transpose=tr, piece_len=24), oracle.spmv(B, v, trans=tr)
        assert got is not None
        np.testing.assert_allclose(got, refb, rtol=0, atol=1e-13 * np.abs(refb).max())
        lens_b = np.diff(B.indptr) if tr else np.diff(B.tocsr().indptr)
        np.testing.assert_array_equal(got[lens_b <= 63], refb[lens_b <= 63])


@pytest.mark.parametrize("shape,row_len,piece_len", [((300, 6000), 5000, 24), ((1, 40000), 40000, 6), ((5000, 3000), 2500, 100),
                                                     ((70000, 20000), 3000, 7)])
def test_pieces_when_every_row_is_long(hip, oracle, shape, row_len, piece_len):
    """edge geometries of the virtual-row layout: EVERY row is cut (the real-row slots stay empty), a single row, more pieces than
    rows (the chunk geometry grows: more rows per lane), a tall matrix whose first 40 rows are long — against the oracle"""
    rng = np.random.default_rng(shape[0] + row_len)
    m, n = shape
    nlong = m if m <= 5000 else 40
    rows = np.repeat(np.arange(nlong), row_len)
    cols = np.concatenate([rng.choice(n, row_len, replace=False) for _ in range(nlong)])
    if nlong < m:  # the other rows: 5 nonzeros each
        rows = np.concatenate([rows, np.repeat(np.arange(nlong, m), 5)])
        cols = np.concatenate([cols, rng.integers(0, n, 5 * (m - nlong))])
    A = sparse.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=shape)
    A.sum_duplicates()
    A.sort_indices()
    x = rng.standard_normal(n)
    got, ref = hip.cs_layout_host_spmv_pieces(A, x, piece_len=piece_len), oracle.spmv(A, x)
    assert got is not None
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    rl = np.diff(A.tocsr().indptr)
    np.testing.assert_array_equal(got[rl <= 63], ref[rl <= 63])
