"""-m gpu: the planes GEMM path (csrc/gemm_pl.hip) through the C ABI.

Operands are fp16 high + low planes with one power-of-two exponent PER ROW (mi_planes_t).  The point of
the per-row exponent is that every example keeps fp32-level accuracy relative to ITS OWN magnitude — the
round-1 matrix-wide scale lost it for rows far below the matrix abs-max (VERDICT r1, weak point 2: dY rows
of well-fit examples) — so the error here is measured PER OUTPUT ROW, relative to that row's own norm, on
operands whose rows span 2^30.  Bars: row-relative error < 1e-5 against fp64 (north_star's bar); exact
equality on small-integer data (layout / index errors cannot hide behind a tolerance); the planes a kernel
writes equal, bit for bit, mi_split_rows of its fp32 result."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.util import dev, dropout_mask

pytestmark = pytest.mark.gpu


def _st():
    from mi355x_rec import _lib
    return _lib.cur_stream()


def _chk(rc, what="call"):
    from mi355x_rec import _lib
    _lib.check(rc, what)


class PB:
    """planes buffer on the device + its mi_planes_t"""

    def __init__(self, lib, rows, K, pad=0):
        from mi355x_rec import _lib
        self.rows, self.K = rows, K
        self.nblk = (K + 15) // 16
        self.rows_alloc = max(rows, 1) + pad                    # pad: a block stride larger than 64 * rows
        assert int(lib.mi_planes_bytes(self.rows_alloc, K)) == self.nblk * self.rows_alloc * 64
        self.data = torch.zeros(self.nblk, self.rows_alloc, 32, dtype=torch.int16, device="cuda")
        self.exp = torch.zeros(max(rows, 1), dtype=torch.int32, device="cuda")
        self.s = _lib.Planes(self.data.data_ptr(), self.exp.data_ptr(), 64 * self.rows_alloc)

    def bits(self):
        """[rows][nblk][32] int16: per row, its 64-byte pieces in k order (-0.0 folded into +0.0)"""
        a = self.data.cpu().numpy()[:, :self.rows, :].transpose(1, 0, 2).reshape(self.rows, -1)
        return np.where(a == -32768, 0, a)

    @property
    def ref(self):
        return C.byref(self.s)

    def merged(self, lib):
        out = torch.empty(self.rows, self.K, device="cuda")
        _chk(lib.mi_merge_rows(self.ref, self.rows, self.K, out.data_ptr(), self.K, _st()))
        return out.cpu().numpy()


def split(lib, X, transpose=False, pad=0):
    x = dev(np.ascontiguousarray(X))
    rows, K = (X.shape[1], X.shape[0]) if transpose else X.shape
    pb = PB(lib, rows, K, pad)
    _chk(lib.mi_split_rows(x.data_ptr(), X.shape[1], rows, K, 1 if transpose else 0, pb.ref, None, _st()))
    return pb


def host_planes(X):
    """numpy restatement of the format: per-row exponent, fp16 hi / lo (RNE), blocks of 16 k"""
    rows, K = X.shape
    K16 = (K + 15) // 16 * 16
    mx = np.abs(X).max(1) if K else np.zeros(rows)
    e = (mx.astype(np.float32).view(np.uint32) >> 23) & 0xff
    s = np.clip(141 - e.astype(np.int64), -100, 100).astype(np.int32)
    u = np.zeros((rows, K16), np.float32)
    u[:, :K] = X.astype(np.float32) * np.exp2(s.astype(np.float32))[:, None]
    hi = u.astype(np.float16)
    lo = (u - hi.astype(np.float32)).astype(np.float16)
    out = np.zeros((rows, K16 // 16, 2, 16), np.float16)
    out[:, :, 0, :] = hi.reshape(rows, -1, 16)
    out[:, :, 1, :] = lo.reshape(rows, -1, 16)
    bits = out.reshape(rows, 2 * K16).view(np.int16)
    return np.where(bits == -32768, 0, bits), s


def row_rel_err(got, ref):
    """max over rows of max|got - ref| / rms(ref row)  (rows that are exactly zero must match exactly)"""
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    rms = np.sqrt(np.mean(ref * ref, 1))
    err = np.abs(got - ref).max(1)
    z = rms == 0
    assert np.all(err[z] == 0)
    return float((err[~z] / rms[~z]).max()) if (~z).any() else 0.0


def rows_spread(rng, M, K, lo_exp=-30):
    """rows whose magnitudes span 2^lo_exp .. 1"""
    X = rng.standard_normal((M, K)).astype(np.float32)
    X *= np.exp2(rng.integers(lo_exp, 1, M)).astype(np.float32)[:, None]
    return X


@pytest.mark.parametrize("rows,K", [(1, 16), (5, 104), (300, 1664), (64, 48), (1000, 128), (130, 512), (37, 32)])
def test_split_rows_format_and_round_trip(lib, rows, K):
    rng = np.random.default_rng(rows + K)
    X = rows_spread(rng, rows, K)
    X[0, :] *= 0 if rows > 3 else 1                       # an all-zero row
    pb = split(lib, X, pad=1)
    ref_bits, ref_exp = host_planes(X)
    assert np.array_equal(pb.exp.cpu().numpy()[:rows], ref_exp)
    got_bits = pb.bits()
    assert float(pb.data[:, rows:, :].abs().max()) == 0 if pb.rows_alloc > rows else True     # nothing written past the rows
    bad = np.argwhere(got_bits != ref_bits)
    assert len(bad) == 0, (len(bad), bad[:8].tolist(), [(hex(int(got_bits[i, j]) & 0xffff), hex(int(ref_bits[i, j]) & 0xffff), float(X[i, (j // 32) * 16 + j % 16])) for i, j in bad[:8]])
    back = pb.merged(lib)
    assert row_rel_err(back, X) < 2 ** -20
    # transposed source: rows of the output are columns of the input (the forward pass's weights)
    pt = split(lib, np.ascontiguousarray(X.T), transpose=True)
    assert np.array_equal(pt.bits(), ref_bits)
    assert np.array_equal(pt.exp.cpu().numpy()[:rows], ref_exp)


SHAPES = [  # M, N, K
    (256, 512, 1664),     # config-3 layer 1: one column tile of 512 (planes out)
    (300, 256, 512),      # layer 2, ragged M
    (128, 128, 256),      # layer 3
    (70, 64, 32),         # N below the narrowest tile, two k-tiles
    (257, 1664, 512),     # layer-1 data gradient shape: 7 column tiles (fp32 result only), ragged M and N
    (64, 16, 16),         # the smallest legal shape: one k-tile
    (192, 384, 48),       # N between tile widths, K = 3 k-tiles
]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dense_fwd_planes_exact_on_integers(lib, M, N, K):
    """small integers: every product and partial sum is exact in fp16 / fp32 -> the result must be EXACT"""
    rng = np.random.default_rng(M + N + K)
    X = rng.integers(-8, 9, (M, K)).astype(np.float32)
    W = rng.integers(-8, 9, (K, N)).astype(np.float32)      # asymmetric, no structure
    b = rng.integers(-3, 4, N).astype(np.float32)
    xp, wt = split(lib, X), split(lib, W, transpose=True)
    Y = torch.full((M, N + 4), -7.0, device="cuda")
    yp = PB(lib, M, N) if N <= 512 else None
    _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dev(b).data_ptr(), Y.data_ptr(), N + 4, yp.ref if yp else None, M, N, K, 0, 1.0,
                                 0, None, None, 0, _st()))
    ref = X.astype(np.float64) @ W.astype(np.float64) + b
    got = Y.cpu().numpy()
    assert np.array_equal(got[:, :N], ref.astype(np.float32))
    assert np.all(got[:, N:] == -7.0)                        # nothing written outside [M, N]
    if yp is not None:
        hb, he = host_planes(ref.astype(np.float32))
        assert np.array_equal(yp.exp.cpu().numpy(), he)
        assert np.array_equal(yp.bits(), hb)


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_dense_fwd_planes_row_relative_error(lib, M, N, K):
    rng = np.random.default_rng(M * 3 + N + K)
    X = rows_spread(rng, M, K)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    W[:, : N // 4] *= np.float32(2.0 ** -12)                 # output columns of very different weight scale
    b = np.zeros(N, np.float32)
    xp, wt = split(lib, X), split(lib, W, transpose=True)
    Y = torch.empty(M, N, device="cuda")
    yp = PB(lib, M, N) if N <= 512 else None
    amax = torch.zeros(64, device="cuda")
    _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dev(b).data_ptr(), Y.data_ptr(), N, yp.ref if yp else None, M, N, K, 0, 1.0, 0,
                                 amax.data_ptr(), None, 0, _st()))
    ref = X.astype(np.float64) @ W.astype(np.float64)
    got = Y.cpu().numpy()
    assert row_rel_err(got, ref) < 1e-5
    assert float(amax.max()) == float(np.abs(got).max())
    if yp is not None:
        assert row_rel_err(yp.merged(lib), ref) < 1e-5
        hb, he = host_planes(got)                           # the planes written == split of the fp32 result
        assert np.array_equal(yp.exp.cpu().numpy(), he)
        assert np.array_equal(yp.bits(), hb)


def test_dense_fwd_planes_bias_relu_dropout(lib):
    M, N, K = 200, 256, 64
    rng = np.random.default_rng(3)
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((K, N)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    xp, wt = split(lib, X), split(lib, W, transpose=True)
    Y0 = torch.empty(M, N, device="cuda"); Y1 = torch.empty(M, N, device="cuda")
    yp = PB(lib, M, N)
    seed, keep = 0x1234567, 0.9
    _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dev(b).data_ptr(), Y0.data_ptr(), N, None, M, N, K, 1, 1.0, seed, None, None, 0, _st()))
    mb = torch.full((M, N // 32 + 1), -1, dtype=torch.int32, device="cuda")          # (one word of slack: mask_ld > N / 32)
    _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dev(b).data_ptr(), Y1.data_ptr(), N, yp.ref, M, N, K, 1, keep, seed, None,
                                 mb.data_ptr(), mb.shape[1], _st()))
    ref = np.maximum(X.astype(np.float64) @ W.astype(np.float64) + b, 0)
    y0 = Y0.cpu().numpy()
    assert np.max(np.abs(y0 - ref)) / np.sqrt(np.mean(ref * ref)) < 1e-5
    assert (y0 == 0).mean() > 0.3                                               # relu did something
    mask = dropout_mask(seed, M, N, keep)
    assert np.array_equal(Y1.cpu().numpy(), (y0 / np.float32(keep)) * mask)      # tf.nn.dropout: div(x, keep) * mask
    hi = yp.data.cpu().numpy()[:, :M, :16].transpose(1, 0, 2).reshape(M, N).view(np.float16)
    assert np.array_equal(hi > 0, Y1.cpu().numpy() > 0)                          # "hi > 0" is the backward's mask
    # ... and so is the one-bit form: bit n & 31 of word n >> 5 of a row
    words = mb.cpu().numpy().view(np.uint32)
    bits = ((words[:, :N // 32, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(M, N).astype(bool)
    assert np.array_equal(bits, Y1.cpu().numpy() > 0) and np.all(words[:, N // 32] == 0xffffffff)      # (the slack word is not touched)


@pytest.mark.parametrize("M,N,K", [(256, 128, 256), (300, 256, 512), (257, 512, 1664), (130, 64, 48)])
def test_dense_bwd_data_planes(lib, M, N, K):
    """dX[M][K] = (dY[M][N] W[K][N]^T) .* (Xact > 0) / keep with dY rows spanning 2^30 (the rows of well-fit
    examples): per-row error against fp64."""
    rng = np.random.default_rng(M + N + K)
    dY = rows_spread(rng, M, N)
    W = (rng.standard_normal((K, N)) / np.sqrt(N)).astype(np.float32)
    Xact = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32) * np.float32(3.0)
    keep = 0.9
    dyp, wp, xap = split(lib, dY), split(lib, W), split(lib, Xact)
    dX = torch.empty(M, K, device="cuda")
    dxp = PB(lib, M, K) if K <= 512 else None
    _chk(lib.mi_dense_bwd_data_planes(dyp.ref, wp.ref, xap.ref, dX.data_ptr(), K, dxp.ref if dxp else None, M, N, K, keep, None,
                                      None, 0, _st()))
    ref = (dY.astype(np.float64) @ W.astype(np.float64).T) * (Xact > 0) / np.float64(np.float32(keep))
    got = dX.cpu().numpy()
    assert row_rel_err(got, ref) < 1e-5
    assert np.array_equal(got == 0, ref == 0) or np.mean((got == 0) != (ref == 0)) < 1e-4
    if dxp is not None:
        hb, he = host_planes(got)
        assert np.array_equal(dxp.exp.cpu().numpy(), he)
        assert np.array_equal(dxp.bits(), hb)
    # the one-bit mask instead of the activation's planes: the same decisions, the same bits (fp32 and planes)
    pos = Xact > 0
    Kw = (K + 31) // 32
    padded = np.zeros((M, Kw * 32), bool); padded[:, :K] = pos
    words = (padded.reshape(M, Kw, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(2).astype(np.uint32)
    dX2 = torch.empty(M, K, device="cuda")
    dxp2 = PB(lib, M, K) if K <= 512 else None
    mbt = dev(words.view(np.int32))
    _chk(lib.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None, dX2.data_ptr(), K, dxp2.ref if dxp2 else None, M, N, K, keep, None,
                                      mbt.data_ptr(), Kw, _st()))
    assert np.array_equal(dX2.cpu().numpy().view(np.uint32), got.view(np.uint32))
    if dxp is not None:
        assert np.array_equal(dxp2.bits(), dxp.bits()) and np.array_equal(dxp2.exp.cpu().numpy(), dxp.exp.cpu().numpy())
    # without a mask (the layer-1 data gradient: the concat has no activation)
    _chk(lib.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None, dX.data_ptr(), K, None, M, N, K, 1.0, None, None, 0, _st()))
    assert row_rel_err(dX.cpu().numpy(), dY.astype(np.float64) @ W.astype(np.float64).T) < 1e-5


# (the 256-row tiles — and with them the fixed-option kernels for 128 and 256 columns — are taken when they fill the chip:
# 65536 rows; the 512-column data gradient has 128-row tiles at any size)
@pytest.mark.parametrize("M,N,K", [(65536, 256, 512), (65536, 128, 256), (512, 256, 512)])
def test_training_variant_of_the_forward_equals_the_general_kernel_bitwise(lib, M, N, K):
    """gemm_pl_k<..., HOT>: the call of a steady-state training step (whole tiles, no fp32 copy, planes + mask bits + abs-max
    out, dropout on) runs a kernel whose launch options are compile-time constants.  Same planes, exponents, mask bits and
    abs-max as the general kernel — which the same call with an fp32 copy asked for runs."""
    rng = np.random.default_rng(M + N + K)
    X = rows_spread(rng, M, K)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = (rng.standard_normal(N) * 1e-3).astype(np.float32)
    xp, wt = split(lib, X), split(lib, W, transpose=True)
    out = []
    for fp32_copy in (True, False):
        Y = torch.empty(M, N, device="cuda") if fp32_copy else None
        yp = PB(lib, M, N)
        mb = torch.zeros(M, N // 32, dtype=torch.int32, device="cuda")
        am = torch.zeros(64, device="cuda")
        _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dev(b).data_ptr(), Y.data_ptr() if fp32_copy else None, N, yp.ref, M, N, K, 1,
                                     0.9, 0xabcdef, am.data_ptr(), mb.data_ptr(), N // 32, _st()))
        out.append((yp.bits(), yp.exp.cpu().numpy(), mb.cpu().numpy(), float(am.max().item()), Y))
    (b0, e0, m0, a0, Y), (b1, e1, m1, a1, _) = out
    assert np.array_equal(b0, b1) and np.array_equal(e0, e1) and np.array_equal(m0, m1) and a0 == a1
    hb, he = host_planes(Y.cpu().numpy())
    assert np.array_equal(b1, hb) and np.array_equal(e1, he) and a1 == float(Y.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(768, 256, 512), (65536, 128, 256), (65536, 64, 128), (512, 128, 256)])
def test_training_variant_of_the_data_gradient_equals_the_general_kernel_bitwise(lib, M, N, K):
    """... and the data gradient's (mask as bits, planes + abs-max out, no fp32 copy): dX[M][K] from dY[M][N]"""
    rng = np.random.default_rng(M + N + K + 1)
    dY = rows_spread(rng, M, N)
    W = (rng.standard_normal((K, N)) / np.sqrt(N)).astype(np.float32)
    words = rng.integers(0, 2 ** 32, (M, K // 32), dtype=np.uint64).astype(np.uint32)
    dyp, wp, mbt = split(lib, dY), split(lib, W), dev(words.view(np.int32))
    out = []
    for fp32_copy in (True, False):
        dX = torch.empty(M, K, device="cuda") if fp32_copy else None
        dxp = PB(lib, M, K)
        am = torch.zeros(64, device="cuda")
        _chk(lib.mi_dense_bwd_data_planes(dyp.ref, wp.ref, None, dX.data_ptr() if fp32_copy else None, K, dxp.ref, M, N, K, 0.9,
                                          am.data_ptr(), mbt.data_ptr(), K // 32, _st()))
        out.append((dxp.bits(), dxp.exp.cpu().numpy(), float(am.max().item()), dX))
    (b0, e0, a0, dX), (b1, e1, a1, _) = out
    assert np.array_equal(b0, b1) and np.array_equal(e0, e1) and a0 == a1
    got = dX.cpu().numpy()
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(M, K).astype(bool)
    r = slice(0, 2048)                                          # (fp64 on the first rows)
    ref = (dY[r].astype(np.float64) @ W.astype(np.float64).T) * bits[r] / np.float64(np.float32(0.9))
    assert row_rel_err(got[r], ref) < 1e-5 and np.all(got[~bits] == 0)
    hb, he = host_planes(got)
    assert np.array_equal(b1, hb) and np.array_equal(e1, he)


def test_planes_entries_refuse_bad_shapes(lib):
    from mi355x_rec._lib import MiError
    X = np.ones((64, 24), np.float32)
    xp = split(lib, X)
    wt = split(lib, np.ones((24, 16), np.float32), transpose=True)
    Y = torch.empty(64, 16, device="cuda")
    with pytest.raises(MiError, match="multiples of 16"):
        _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, None, Y.data_ptr(), 16, None, 64, 16, 24, 0, 1.0, 0, None, None, 0, _st()))
    xp2 = split(lib, np.ones((64, 32), np.float32))
    w2 = split(lib, np.ones((32, 1024), np.float32), transpose=True)
    yp = PB(lib, 64, 1024)
    with pytest.raises(MiError, match="N <= 512"):
        _chk(lib.mi_dense_fwd_planes(xp2.ref, w2.ref, None, None, 0, yp.ref, 64, 1024, 32, 0, 1.0, 0, None, None, 0, _st()))


@pytest.mark.parametrize("E,F,B,nd,tail", [(64, 26, 300, 0, 0), (128, 40, 65, 0, 0), (32, 5, 129, 0, 0), (48, 3, 17, 0, 0),
                                           (64, 26, 300, 13, 32), (64, 26, 129, 13, 128), (32, 5, 65, 16, 16), (48, 3, 17, 1, 16),
                                           (128, 7, 33, 40, 64)])
def test_embed_fm_planes_fwd(lib, E, F, B, nd, tail):
    """the gather that writes the concat as planes: same bits as mi_split_rows of the materialised concat;
    sumv / fm equal to the fp32 gather kernel's.  nd / tail: the canned estimators' raw numeric columns — nd values and
    tail - nd zero columns after the embedding columns, under the example's one exponent (an example whose largest value
    is a numeric column takes its exponent from it)."""
    rng = np.random.default_rng(E + F + B)
    vocab = rng.integers(2, 50, F)
    off = np.concatenate([[0], np.cumsum(vocab)]).astype(np.int64)
    table = rng.standard_normal((int(off[-1]), E)).astype(np.float32)
    table[:3] *= np.float32(2.0 ** -20)
    ids = np.stack([rng.integers(0, v, B) for v in vocab], 1).astype(np.int32)
    x = None
    if nd:
        x = rng.standard_normal((B, nd)).astype(np.float32)
        x[::3] *= np.float32(40.0)                       # every third example: a numeric column is the largest value
        x[1::7] *= np.float32(2.0 ** -12)
    t, fo, di = dev(table), dev(off[:-1].copy()), dev(ids)
    dx = dev(x) if nd else None
    cp = PB(lib, B, F * E + tail, pad=1)
    sumv = torch.empty(B, E, device="cuda"); fm = torch.empty(B, device="cuda"); amax = torch.zeros(64, device="cuda")
    _chk(lib.mi_embed_fm_planes_fwd(t.data_ptr(), fo.data_ptr(), di.data_ptr(), B, F, E, sumv.data_ptr(), fm.data_ptr(), cp.ref,
                                    amax.data_ptr(), dx.data_ptr() if nd else None, nd, tail, 0, _st()))
    concat = torch.empty(B, F * E, device="cuda"); sumv2 = torch.empty(B, E, device="cuda"); fm2 = torch.empty(B, device="cuda")
    _chk(lib.mi_embed_fm_linear_fwd(t.data_ptr(), None, fo.data_ptr(), di.data_ptr(), B, F, E, concat.data_ptr(), F * E,
                                    sumv2.data_ptr(), fm2.data_ptr(), None, None, 1, 0, _st()))
    rows = ids.astype(np.int64) + off[:-1][None, :]
    assert np.array_equal(concat.cpu().numpy(), table[rows].reshape(B, F * E))
    full = concat.cpu().numpy()
    if tail:
        full = np.concatenate([full, x, np.zeros((B, tail - nd), np.float32)], 1)
    hb, he = host_planes(full)
    assert np.array_equal(cp.exp.cpu().numpy(), he)
    assert np.array_equal(cp.bits(), hb)
    assert torch.equal(sumv, sumv2) and torch.equal(fm, fm2)
    assert float(amax.max()) == float(np.abs(full).max())
    if tail:      # refused: a tail that is no whole number of k-blocks, more numeric columns than the tail holds, a tail wider than 4 E
        assert lib.mi_embed_fm_planes_fwd(t.data_ptr(), fo.data_ptr(), di.data_ptr(), B, F, E, None, None, cp.ref, None, dx.data_ptr(), nd, tail + 8, 0, _st()) != 0
        assert lib.mi_embed_fm_planes_fwd(t.data_ptr(), fo.data_ptr(), di.data_ptr(), B, F, E, None, None, cp.ref, None, dx.data_ptr(), tail + 1, tail, 0, _st()) != 0
        assert lib.mi_embed_fm_planes_fwd(t.data_ptr(), fo.data_ptr(), di.data_ptr(), B, F, E, None, None, cp.ref, None, dx.data_ptr(), nd, 4 * E + 16, 0, _st()) != 0


def test_split_weights_one_launch(lib):
    """all weight planes of a step in one launch: both orientations, one exponent from the block's abs-max"""
    from mi355x_rec import _lib
    rng = np.random.default_rng(5)
    shapes = [(1664, 512), (512, 256), (256, 128), (48, 16)]
    offs, o = [], 0
    for k, n in shapes:
        offs.append(o); o += (k * n + 15) // 16 * 16
    dense = (rng.standard_normal(o) * 0.1).astype(np.float32)
    d = dev(dense)
    amax = torch.zeros(64, device="cuda")
    _chk(lib.mi_absmax(d.data_ptr(), o, amax.data_ptr(), _st()))
    jobs = (_lib.WeightJob * len(shapes))()
    bufs = []
    for q, ((k, n), off) in enumerate(zip(shapes, offs)):
        pw, pt = PB(lib, k, n), PB(lib, n, k, pad=3)
        jobs[q].offset, jobs[q].K, jobs[q].N, jobs[q].w, jobs[q].wt = off, k, n, pw.s, pt.s
        bufs.append((pw, pt))
    _chk(lib.mi_split_weights(d.data_ptr(), jobs, len(shapes), amax.data_ptr(), _st()))
    e = (np.float32(np.abs(dense).max()).view(np.uint32) >> 23) & 0xff
    sx = int(np.clip(141 - int(e), -100, 100))
    for (k, n), off, (pw, pt) in zip(shapes, offs, bufs):
        W = dense[off:off + k * n].reshape(k, n)
        assert np.all(pw.exp.cpu().numpy() == sx) and np.all(pt.exp.cpu().numpy() == sx)
        assert np.max(np.abs(pw.merged(lib) - W)) <= 2.0 ** -21 * np.abs(dense).max()
        assert np.max(np.abs(pt.merged(lib) - W.T)) <= 2.0 ** -21 * np.abs(dense).max()
        # same bits as the format's restatement with the common exponent
        u = W * np.float32(2.0) ** sx
        hi = u.astype(np.float16)
        assert np.array_equal(pw.data.cpu().numpy()[:, :k, :16].transpose(1, 0, 2).reshape(k, -1)[:, :n].view(np.float16), hi)


# (N = 128 / 256 / 512 with K a multiple of 128: the LDS-DMA kernel, wgrad_pl.hip; every other whole-tile shape — N = 384,
# 640 here — the register-staged one, gemm_wgrad_pl.inc.  The shipped library has no switch between them.)
@pytest.mark.parametrize("M,N,K,bias", [(4096, 128, 256, True), (8192, 512, 1664, True), (1024, 256, 512, False), (32, 128, 128, True),
                                        (2080, 256, 384, True), (4096, 384, 128, True), (2080, 640, 256, True), (32, 384, 128, False)])
def test_dense_bwd_weight_planes_against_fp64(lib, M, N, K, bias):
    """dW = X^T dY and db = colsum(dY) from planes whose rows span 2^-20 .. 1 (examples with tiny gradients next
    to large ones): error relative to the rms of the exact result at fp32 level, like the fp32-operand entry;
    twice the same bits (fixed-order split-K)."""
    from mi355x_rec import _lib as L
    rng = np.random.default_rng(M + N + K)
    X = np.maximum(rows_spread(rng, M, K, -6), 0).astype(np.float32)
    dY = (rows_spread(rng, M, N, -20) * 1e-4).astype(np.float32)
    xp, dyp = split(lib, X), split(lib, dY)
    ax = torch.zeros(L.AMAX_SLOTS, device="cuda"); ady = torch.zeros(L.AMAX_SLOTS, device="cuda")
    _chk(lib.mi_absmax(dev(X).data_ptr(), X.size, ax.data_ptr(), _st()))
    _chk(lib.mi_absmax(dev(dY).data_ptr(), dY.size, ady.data_ptr(), _st()))
    ga = L.GemmAmax(ax.data_ptr(), ady.data_ptr(), None)
    ws = torch.empty(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K) + 256, dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        dW = torch.full((K, N), float("nan"), device="cuda"); db = torch.full((N,), float("nan"), device="cuda")
        _chk(lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), db.data_ptr() if bias else None, M, N, K,
                                            ws.data_ptr(), ws.numel(), C.byref(ga), _st()))
        outs.append((dW.cpu().numpy(), db.cpu().numpy()))
    refW = X.astype(np.float64).T @ dY.astype(np.float64)
    assert np.max(np.abs(outs[0][0] - refW)) / np.sqrt(np.mean(refW * refW)) < 1e-5
    if bias:
        refb = dY.astype(np.float64).sum(0)
        assert np.max(np.abs(outs[0][1] - refb)) / np.sqrt(np.mean(refb * refb)) < 1e-5
        assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0])


def test_dense_bwd_weight_planes_refuses_ragged_shapes(lib):
    from mi355x_rec import _lib as L
    xp, dyp = PB(lib, 64, 128), PB(lib, 64, 128)
    a = torch.zeros(L.AMAX_SLOTS, device="cuda")
    ga = L.GemmAmax(a.data_ptr(), a.data_ptr(), None)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    dW = torch.empty(128, 128, device="cuda")
    for M, N, K in ((48, 128, 128), (64, 96, 128), (64, 128, 64)):
        assert lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), None, M, N, K, ws.data_ptr(), ws.numel(),
                                              C.byref(ga), _st()) != 0


@pytest.mark.parametrize("N", [128, 384])              # the LDS-DMA kernel / the register-staged kernel
def test_weight_gradient_keeps_fp16_subnormal_operands(lib, N):
    """An example 2^-30 below the matrices' abs-max reaches the matrix pipe as fp16 SUBNORMAL values (the per-example
    factor 2^d brings its rows to matrix-wide scales).  The MFMA must not flush them: with one such example
    carrying all of the signal, dW is its outer product (exact here: powers of two)."""
    from mi355x_rec import _lib as L
    M, K = 32, 128
    X = np.zeros((M, K), np.float32); dY = np.zeros((M, N), np.float32)
    X[0, :] = 1.0                       # example 0: full-size activations, a gradient 2^-30 below the batch maximum
    dY[0, :] = 2.0 ** -30
    X[1, 0] = 1.0                       # example 1 sets the abs-max of dY and contributes to dW[0, :] only
    dY[1, :] = 1.0
    xp, dyp = split(lib, X), split(lib, dY)
    ax = torch.zeros(L.AMAX_SLOTS, device="cuda"); ady = torch.zeros(L.AMAX_SLOTS, device="cuda")
    _chk(lib.mi_absmax(dev(X).data_ptr(), X.size, ax.data_ptr(), _st()))
    _chk(lib.mi_absmax(dev(dY).data_ptr(), dY.size, ady.data_ptr(), _st()))
    ga = L.GemmAmax(ax.data_ptr(), ady.data_ptr(), None)
    ws = torch.empty(lib.mi_dense_bwd_weight_planes_workspace_bytes(M, N, K) + 256, dtype=torch.uint8, device="cuda")
    dW = torch.empty(K, N, device="cuda")
    _chk(lib.mi_dense_bwd_weight_planes(xp.ref, dyp.ref, dW.data_ptr(), None, M, N, K, ws.data_ptr(), ws.numel(), C.byref(ga), _st()))
    ref = X.astype(np.float64).T @ dY.astype(np.float64)
    got = dW.cpu().numpy()
    assert np.array_equal(got[1:], ref[1:].astype(np.float32)), (got[1, :4], ref[1, :4])     # rows fed by the tiny example alone
    assert np.allclose(got[0], ref[0], rtol=1e-6)


@pytest.mark.parametrize("M,K,mask", [(300, 128, True), (65, 16, False), (1000, 512, True), (33, 1040, True)])
def test_logits_layer_data_gradient_as_planes_matches_gemv_then_split_bitwise(lib, M, K, mask):
    """mi_dense_bwd_data_vec_planes == mi_dense_bwd_data (N = 1, ReLU) followed by mi_split_rows: same fp32 bits, same planes"""
    rng = np.random.default_rng(M + K)
    dY = (rng.standard_normal(M) * np.exp2(rng.integers(-20, 1, M))).astype(np.float32)
    W = rng.standard_normal(K).astype(np.float32)
    Xact = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    dy, w, xa = dev(dY), dev(W), dev(Xact)
    keep = 0.9
    ref = torch.empty(M, K, device="cuda")
    _chk(lib.mi_dense_bwd_data(dy.data_ptr(), 1, w.data_ptr(), xa.data_ptr() if mask else None, K, ref.data_ptr(), K, M, 1, K, keep, 1,
                               None, _st()))
    refp = PB(lib, M, K)
    _chk(lib.mi_split_rows(ref.data_ptr(), K, M, K, 0, refp.ref, None, _st()))
    got = torch.full((M, K), float("nan"), device="cuda")
    gotp = PB(lib, M, K)
    from mi355x_rec import _lib as L
    am = torch.zeros(L.AMAX_SLOTS, device="cuda")
    _chk(lib.mi_dense_bwd_data_vec_planes(dy.data_ptr(), 1, w.data_ptr(), xa.data_ptr() if mask else None, K, keep, got.data_ptr(), K,
                                          gotp.ref, M, K, am.data_ptr(), None, 0, _st()))
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
    assert np.array_equal(gotp.bits(), refp.bits()) and np.array_equal(gotp.exp.cpu().numpy(), refp.exp.cpu().numpy())
    assert float(am.max()) == float(ref.abs().max())
    # planes only
    gotp2 = PB(lib, M, K)
    _chk(lib.mi_dense_bwd_data_vec_planes(dy.data_ptr(), 1, w.data_ptr(), xa.data_ptr() if mask else None, K, keep, None, 0,
                                          gotp2.ref, M, K, None, None, 0, _st()))
    assert np.array_equal(gotp2.bits(), refp.bits())
    if mask:          # the one-bit mask instead of the fp32 activation: same bits
        Kw = (K + 31) // 32
        padded = np.zeros((M, Kw * 32), bool); padded[:, :K] = Xact > 0
        words = (padded.reshape(M, Kw, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(2).astype(np.uint32)
        mbt = dev(words.view(np.int32))
        gotp3 = PB(lib, M, K)
        got3 = torch.full((M, K), float("nan"), device="cuda")
        _chk(lib.mi_dense_bwd_data_vec_planes(dy.data_ptr(), 1, w.data_ptr(), None, K, keep, got3.data_ptr(), K, gotp3.ref, M, K, None,
                                              mbt.data_ptr(), Kw, _st()))
        assert np.array_equal(got3.cpu().numpy().view(np.uint32), ref.cpu().numpy().view(np.uint32))
        assert np.array_equal(gotp3.bits(), refp.bits())


@pytest.mark.parametrize("M,K,keep,parts", [(300, 128, 0.9, (True, True)), (4099, 64, 1.0, (True, False)), (1000, 256, 0.75, (False, True)),
                                            (65, 128, 0.9, (False, False))])
def test_logits_head_fused_equals_the_unfused_sequence(lib, M, K, keep, parts):
    """mi_logits_head_fused (round 4: the logits layer's forward, the head, and the layer's backward in one pass over the last
    hidden layer's output) against the five launches it replaces — mi_dense_fwd (N = 1), mi_sigmoid_ce_head, mi_dense_bwd_weight
    (N = 1), mi_dense_bwd_data_vec_planes: the logits layer's dot product to 2e-6 of sum |x w|; from there on everything per
    example (logits, d_logit, the planes and the fp32 copy of the data gradient, its abs-max) bit for bit, the sums over examples
    (loss, d_logit_sum = db, dW) to 1e-6 of their size (another association), twice the same bits."""
    from mi355x_rec import _lib as L
    has_lin, has_fm = parts
    rng = np.random.default_rng(M + K)
    X = (np.maximum(rng.standard_normal((M, K)), 0) * (rng.random((M, K)) < keep) / keep).astype(np.float32)     # a relu + dropout layer's output
    w = (rng.standard_normal(K) / np.sqrt(K)).astype(np.float32)
    b = np.float32(0.03)
    lin = rng.standard_normal(M).astype(np.float32) * 0.1; fm = rng.standard_normal(M).astype(np.float32) * 0.1
    y = (rng.random(M) < 0.3).astype(np.uint8)
    scale = np.float32(1.0 / M)
    Kw = (K + 31) // 32
    padded = np.zeros((M, Kw * 32), bool); padded[:, :K] = X > 0
    words = (padded.reshape(M, Kw, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(2).astype(np.uint32)
    dX_, dw, db_, dlin, dfm, dy_, dbits = dev(X), dev(w), dev(np.array([b], np.float32)), dev(lin), dev(fm), dev(y), dev(words.view(np.int32))
    lb = dev(np.array([0.07], np.float32))
    pl = dlin.data_ptr() if has_lin else None
    pf = dfm.data_ptr() if has_fm else None
    # ---- fused, twice
    outs = []
    for _ in range(2):
        dnn1 = torch.full((M,), float("nan"), device="cuda"); logits1 = torch.empty(M, device="cuda"); loss1 = torch.empty(1, device="cuda")
        dl1 = torch.empty(M, device="cuda"); ds1 = torch.empty(1, device="cuda"); dW1 = torch.empty(K, device="cuda"); dB1 = torch.empty(1, device="cuda")
        p1 = PB(lib, M, K); g1 = torch.full((M, K), float("nan"), device="cuda"); am1 = torch.zeros(L.AMAX_SLOTS, device="cuda")
        tws = torch.empty(int(lib.mi_logits_head_fused_workspace_bytes(M, K)) + 256, dtype=torch.uint8, device="cuda")
        _chk(lib.mi_logits_head_fused(dX_.data_ptr(), K, dw.data_ptr(), db_.data_ptr(), pl, lb.data_ptr(), pf, dy_.data_ptr(), M, K, float(scale),
                                      dbits.data_ptr(), Kw, keep, dnn1.data_ptr(), logits1.data_ptr(), loss1.data_ptr(), dl1.data_ptr(), ds1.data_ptr(),
                                      dW1.data_ptr(), dB1.data_ptr(), p1.ref, g1.data_ptr(), K, am1.data_ptr(), tws.data_ptr(), tws.numel(), _st()))
        torch.cuda.synchronize()
        outs.append((dnn1, logits1, loss1, dl1, ds1, dW1, dB1, p1.bits(), p1.exp.cpu().numpy(), g1, am1))
    # ---- the unfused sequence (the head and the backward on the fused kernel's dnn)
    dnn0 = torch.empty(M, device="cuda"); logits0 = torch.empty(M, device="cuda"); loss0 = torch.empty(1, device="cuda")
    dl0 = torch.empty(M, device="cuda"); ds0 = torch.empty(1, device="cuda")
    _chk(lib.mi_dense_fwd(dX_.data_ptr(), K, dw.data_ptr(), db_.data_ptr(), dnn0.data_ptr(), 1, M, 1, K, 0, 1.0, 0, None, _st()))
    # the dot product: gemv_fwd_k's association, but hipcc contracts the two kernels' multiply-adds differently — to 2e-6 of
    # sum |x w|; everything after it is compared on the fused kernel's own dnn
    dnn1 = outs[0][0]
    assert float(((dnn1 - dnn0).abs() / ((dX_.abs() * dw.abs()[None, :]).sum(1) + 1e-6)).max()) < 2e-6
    dnn0 = dnn1.clone()
    hws = torch.empty(int(lib.mi_head_workspace_bytes(M)) + 256, dtype=torch.uint8, device="cuda")
    _chk(lib.mi_sigmoid_ce_head(pl, lb.data_ptr(), pf, dnn0.data_ptr(), dy_.data_ptr(), M, float(scale), logits0.data_ptr(), loss0.data_ptr(),
                                dl0.data_ptr(), ds0.data_ptr(), hws.data_ptr(), hws.numel(), _st()))
    dW0 = torch.empty(K, device="cuda"); dB0 = torch.empty(1, device="cuda")
    wws = torch.empty(int(lib.mi_dense_bwd_weight_workspace_bytes(M, 1, K)) + 256, dtype=torch.uint8, device="cuda")
    _chk(lib.mi_dense_bwd_weight(dX_.data_ptr(), K, dl0.data_ptr(), 1, dW0.data_ptr(), dB0.data_ptr(), M, 1, K, wws.data_ptr(), wws.numel(), None, _st()))
    p0 = PB(lib, M, K); g0 = torch.empty(M, K, device="cuda"); am0 = torch.zeros(L.AMAX_SLOTS, device="cuda")
    _chk(lib.mi_dense_bwd_data_vec_planes(dl0.data_ptr(), 1, dw.data_ptr(), None, K, keep, g0.data_ptr(), K, p0.ref, M, K, am0.data_ptr(),
                                          dbits.data_ptr(), Kw, _st()))
    dnn1, logits1, loss1, dl1, ds1, dW1, dB1, bits1, exp1, g1, am1 = outs[0]
    bit = lambda a, b_: np.array_equal(a.cpu().numpy().view(np.uint32), b_.cpu().numpy().view(np.uint32))
    assert bit(logits1, logits0) and bit(dl1, dl0) and bit(g1, g0)
    assert np.array_equal(bits1, p0.bits()) and np.array_equal(exp1, p0.exp.cpu().numpy()) and float(am1.max()) == float(am0.max())
    close = lambda a, b_, ref: float((a - b_).abs().max()) <= 1e-6 * float(ref.abs().max()) + 1e-12
    assert close(loss1, loss0, loss0) and close(ds1, ds0, dl0.abs().sum()[None]) and close(dB1, dB0, dl0.abs().sum()[None])
    assert close(dW1, dW0, (dX_.abs() * dl0.abs()[:, None]).sum(0))
    assert bit(ds1, dB1)
    for a, b_ in zip(outs[0], outs[1]):          # reproducible
        assert np.array_equal(a, b_) if isinstance(a, np.ndarray) else torch.equal(a, b_)


@pytest.mark.parametrize("M,K,keep,parts", [(4096, 256, 0.9, (True, True)), (300, 64, 1.0, (True, False)), (1029, 512, 0.75, (False, True)),
                                            (128, 128, 0.9, (False, False))])
def test_hidden_logits_head_fused_equals_the_layer_forward_then_the_fused_tail(lib, M, K, keep, parts):
    """mi_hidden_logits_head_fused (round 4: the last hidden layer's GEMM with the logits layer, the head and the logits layer's
    backward in its epilogue — the layer's output never reaches memory) against mi_dense_fwd_planes + the unfused tail on
    that output: the logits layer's dot product to 2e-6 of sum |h w|; from there on, on the kernel's own dnn, everything per
    example bit for bit (logits, d_logit, the planes of the data gradient, their exponents and abs-max), the sums over examples
    (loss, d_logit_sum = db, dW) to 1e-6 of their size; twice the same bits."""
    from mi355x_rec import _lib as L
    N = 128
    has_lin, has_fm = parts
    rng = np.random.default_rng(M + K + 7)
    X = rows_spread(rng, M, K, lo_exp=-8)
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    w = (rng.standard_normal(N) / np.sqrt(N)).astype(np.float32)
    b = np.float32(0.03)
    lin = rng.standard_normal(M).astype(np.float32) * 0.1; fm = rng.standard_normal(M).astype(np.float32) * 0.1
    y = (rng.random(M) < 0.3).astype(np.uint8)
    scale = np.float32(1.0 / M)
    seed = 0x5eed1234
    xp, wt = split(lib, X), split(lib, W, transpose=True)
    dbias, dw, db_, dlin, dfm, dy_ = dev(bias), dev(w), dev(np.array([b], np.float32)), dev(lin), dev(fm), dev(y)
    lb = dev(np.array([0.07], np.float32))
    pl = dlin.data_ptr() if has_lin else None
    pf = dfm.data_ptr() if has_fm else None
    # ---- the layer alone: its output and mask bits
    Y = torch.empty(M, N, device="cuda")
    mb = torch.zeros(M, N // 32, dtype=torch.int32, device="cuda")
    _chk(lib.mi_dense_fwd_planes(xp.ref, wt.ref, dbias.data_ptr(), Y.data_ptr(), N, None, M, N, K, 1, keep, seed, None, mb.data_ptr(), N // 32, _st()))
    # ---- fused, twice
    outs = []
    for _ in range(2):
        dnn1 = torch.full((M,), float("nan"), device="cuda"); logits1 = torch.empty(M, device="cuda"); loss1 = torch.empty(1, device="cuda")
        dl1 = torch.empty(M, device="cuda"); ds1 = torch.empty(1, device="cuda"); dW1 = torch.empty(N, device="cuda"); dB1 = torch.empty(1, device="cuda")
        p1 = PB(lib, M, N); am1 = torch.zeros(L.AMAX_SLOTS, device="cuda")
        tws = torch.empty(int(lib.mi_hidden_logits_head_fused_workspace_bytes(M, N)) + 256, dtype=torch.uint8, device="cuda")
        _chk(lib.mi_hidden_logits_head_fused(xp.ref, wt.ref, dbias.data_ptr(), M, N, K, 1, keep, seed, dw.data_ptr(), db_.data_ptr(), pl,
                                             lb.data_ptr(), pf, dy_.data_ptr(), float(scale), dnn1.data_ptr(), logits1.data_ptr(),
                                             loss1.data_ptr(), dl1.data_ptr(), ds1.data_ptr(), dW1.data_ptr(), dB1.data_ptr(), p1.ref,
                                             am1.data_ptr(), tws.data_ptr(), tws.numel(), _st()))
        torch.cuda.synchronize()
        outs.append((dnn1, logits1, loss1, dl1, ds1, dW1, dB1, p1.bits(), p1.exp.cpu().numpy(), am1))
    dnn1, logits1, loss1, dl1, ds1, dW1, dB1, bits1, exp1, am1 = outs[0]
    # the logits layer's dot product on the layer's output (fp64)
    ref_dnn = (Y.double() * dw.double()[None, :]).sum(1) + float(b)
    assert float(((dnn1.double() - ref_dnn).abs() / ((Y.abs() * dw.abs()[None, :]).sum(1).double() + 1e-6)).max()) < 2e-6
    # ---- the unfused head and backward on the kernel's own dnn
    logits0 = torch.empty(M, device="cuda"); loss0 = torch.empty(1, device="cuda"); dl0 = torch.empty(M, device="cuda"); ds0 = torch.empty(1, device="cuda")
    hws = torch.empty(int(lib.mi_head_workspace_bytes(M)) + 256, dtype=torch.uint8, device="cuda")
    _chk(lib.mi_sigmoid_ce_head(pl, lb.data_ptr(), pf, dnn1.data_ptr(), dy_.data_ptr(), M, float(scale), logits0.data_ptr(), loss0.data_ptr(),
                                dl0.data_ptr(), ds0.data_ptr(), hws.data_ptr(), hws.numel(), _st()))
    dW0 = torch.empty(N, device="cuda"); dB0 = torch.empty(1, device="cuda")
    wws = torch.empty(int(lib.mi_dense_bwd_weight_workspace_bytes(M, 1, N)) + 256, dtype=torch.uint8, device="cuda")
    _chk(lib.mi_dense_bwd_weight(Y.data_ptr(), N, dl0.data_ptr(), 1, dW0.data_ptr(), dB0.data_ptr(), M, 1, N, wws.data_ptr(), wws.numel(), None, _st()))
    p0 = PB(lib, M, N); am0 = torch.zeros(L.AMAX_SLOTS, device="cuda")
    _chk(lib.mi_dense_bwd_data_vec_planes(dl0.data_ptr(), 1, dw.data_ptr(), None, N, keep, None, N, p0.ref, M, N, am0.data_ptr(),
                                          mb.data_ptr(), N // 32, _st()))
    bit = lambda a, b_: np.array_equal(a.cpu().numpy().view(np.uint32), b_.cpu().numpy().view(np.uint32))
    assert bit(logits1, logits0) and bit(dl1, dl0)
    assert np.array_equal(bits1, p0.bits()) and np.array_equal(exp1, p0.exp.cpu().numpy()) and float(am1.max()) == float(am0.max())
    close = lambda a, b_, ref: float((a - b_).abs().max()) <= 1e-6 * float(ref.abs().max()) + 1e-12
    assert close(loss1, loss0, loss0) and close(ds1, ds0, dl0.abs().sum()[None]) and close(dB1, dB0, dl0.abs().sum()[None])
    assert close(dW1, dW0, (Y.abs() * dl0.abs()[:, None]).sum(0))
    assert bit(ds1, dB1)
    for a, b_ in zip(outs[0], outs[1]):          # reproducible
        assert np.array_equal(a, b_) if isinstance(a, np.ndarray) else torch.equal(a, b_)
