"""CPU model of the int8 tier's COSINE bound chain (yams_amd/csrc/scan_i8_kernel.hip, file header).

A numpy restatement, with the same roundings in the same places as the device code (fp32 where the kernels use fp32):

    shadow_build_i8_kernel : unit rows x~, int8 xi with one scale s_b per block of 64 rows, e_b >= the largest
                             measured residue |x~ - s_b xi| of the block
    prep_i8_kernel         : unit query q~, int8 qi with scale t_q, c_q >= |t_q qi|, f_q >= |q~ - t_q qi| + slop
    filter score           : u = fmaf(float(I), s_b * t_q, fmaf(e_b, c_q, f_q)),  I = xi . qi exactly (int32)
    integer threshold      : A = (tau - f)/t shrunk by 2^-19, B = (c/t)(1 + 2^-19),
                             -T = trunc(clamp(fmaf(-A, 1/s_b, fmaf(B, e_b/s_b, 2)))),  survives iff I - T >= 0

The completeness proof of the re-score rests on two properties, checked here exhaustively on small shards:

    (P1)  u >= float(cos(x, q))            the filter score is an upper bound of the exact similarity AS THE REFERENCE
                                            ROUNDS IT (sqlite_vec_backend.cpp:4271-4276: fp64, then one cast)
    (P2)  u >= tau  =>  I >= T              the integer test never dismisses a row whose bound reaches the threshold

for ordinary rows and for hostile ones (one-hot, heavy-tailed, nearly parallel to the query, tiny norms), queries of
any length.  The model also reports the slack u - cos the tier pays for being an int8 filter.  Test infrastructure only."""
import numpy as np
import pytest

from test_l2_int8_model import build_shadow, fma, neg_threshold

f32 = np.float32
U24 = 5.9604645e-8


def prep_queries_cosine(q):
    """prep_queries_kernel (unit query, fp64 norm then one division per element in fp64 -> fp32) + prep_i8_kernel (raw = 0)."""
    nq, d = q.shape
    qn = np.sqrt((q.astype(np.float64) ** 2).sum(1))
    qp = (q.astype(np.float64) / qn[:, None]).astype(f32)                # qprep
    am = np.abs(qp).max(1)
    t = (am / f32(127.0)).astype(f32); it = (f32(127.0) / am).astype(f32)
    qi = np.clip(np.rint((qp * it[:, None]).astype(f32)), -127, 127).astype(f32)
    qq = (t[:, None] * qi).astype(f32)
    dd = fma(-t[:, None], qi, qp)
    # fp32 running sums of squares (fmaf chains split over 256 threads + a tree: any order; the `up` factor covers it)
    csum = (qq.astype(np.float64) ** 2).sum(1).astype(f32)
    fsum = (dd.astype(np.float64) ** 2).sum(1).astype(f32)
    up = f32(1.0 + (d + 16.0) * U24)
    c = (np.sqrt(csum).astype(f32) * up).astype(f32)
    f = (np.sqrt(fsum).astype(f32) * up + f32((d + 32.0) * U24 + 1e-6) * f32(1.0)).astype(f32)
    return qi.astype(np.int64), t, c, f, qn


def run_model(x, q, tau_rank=16, stride=8):
    n, d = x.shape
    xi, s, e = build_shadow(x)
    qi, t, c, f, qn = prep_queries_cosine(q)
    I = xi.astype(np.int64) @ qi.T                                        # exact integer dot products [n, nq]
    blk = np.arange(n) // 64
    S = (s[blk][:, None] * t[None, :]).astype(f32)                        # m.x * qm.x
    K = fma(e[blk][:, None], c[None, :], f[None, :])                      # fmaf(m.y, qm.y, qm.z)
    u = fma(I.astype(f32), S, K)
    # the reference's similarity: fp64 dot / (sqrt(nsq) * |q|), one cast (:4271-4276)
    xd = x.astype(np.float64)
    cos = (xd @ q.astype(np.float64).T) / (np.sqrt((xd ** 2).sum(1))[:, None] * qn[None, :])
    sim = cos.astype(f32)
    assert (u >= sim).all(), f"P1 violated: min(u - sim) = {(u.astype(np.float64) - sim).min()}"
    # tau: the tau_rank-th largest score of every stride-th row (what select_tau makes of the sample pass)
    tau = np.sort(u[::stride], axis=0)[-tau_rank].astype(f32)
    A = ((tau - f) / t).astype(f32)
    A = (A - np.abs(A) * f32(1.9073486e-6)).astype(f32)
    B = ((c / t).astype(f32) * f32(1.0 + 1.9073486e-6)).astype(f32)
    is_ = (f32(1.0) / s).astype(f32)
    g = (e * is_).astype(f32)
    nt = neg_threshold(A[None, :], is_[blk][:, None], B[None, :], g[blk][:, None])
    passes = I + nt >= 0
    wanted = u >= tau[None, :]
    missed = wanted & ~passes
    assert not missed.any(), f"P2 violated: {missed.sum()} rows whose bound reaches tau fail the integer test"
    slack = (u.astype(np.float64) - cos)
    return int(passes.sum()), int(wanted.sum()), float(slack.mean()), float(slack.max())


def corpus(rng, n, d, kind):
    if kind == "uniform":
        return rng.uniform(-1, 1, (n, d)).astype(f32)
    if kind == "gauss_scaled":                      # any row length: the shadow is of the unit rows
        return (rng.normal(0, 1, (n, d)) * rng.uniform(1e-3, 1e3, (n, 1))).astype(f32)
    if kind == "heavy_tailed":                      # a few huge components per row: the quantisation's worst case
        x = rng.normal(0, 1, (n, d)).astype(f32)
        idx = rng.integers(0, d, (n, 3))
        x[np.arange(n)[:, None], idx] *= f32(40.0)
        return x
    if kind == "one_hot":                           # residue 0 on most components, everything in one
        x = (rng.normal(0, 1e-3, (n, d))).astype(f32)
        x[np.arange(n), rng.integers(0, d, n)] = f32(1.0)
        return x
    if kind == "clustered":                         # rows nearly parallel to each other and to the queries
        centres = rng.normal(0, 1, (4, d)).astype(f32)
        return (centres[rng.integers(0, 4, n)] + 0.01 * rng.normal(0, 1, (n, d))).astype(f32)
    raise ValueError(kind)


@pytest.mark.parametrize("d", [256, 768])
@pytest.mark.parametrize("kind", ["uniform", "gauss_scaled", "heavy_tailed", "one_hot", "clustered"])
def test_filter_score_bounds_the_similarity_and_the_integer_test_keeps_every_bound_above_tau(kind, d):
    rng = np.random.default_rng(23)
    n, nq = 2048 + 37, 16
    x = corpus(rng, n, d, kind)
    if kind == "clustered":
        q = (x[rng.integers(0, n, nq)] + 0.005 * rng.normal(0, 1, (nq, d))).astype(f32)
    else:
        q = corpus(rng, nq, d, kind)
        q[0] *= f32(1e4); q[1] *= f32(1e-4)       # queries of other lengths than the rows
        q[2] = x[5]                               # a query equal to a row: cos = 1 exactly
    passed, wanted, slack_mean, slack_max = run_model(x, q)
    # tightness: the integer test may keep more than reach tau, not orders of magnitude more
    assert passed <= 4 * wanted + 64 * nq, (kind, d, passed, wanted)
    if kind == "uniform":
        assert slack_mean < 0.03, slack_mean      # ~0.01-0.02 above the true value


def test_a_row_identical_to_the_query_is_never_lost():
    """cos = 1: u must reach 1 however the residues fall (the reference's self-match test,
    vector_smoke_catch2_test.cpp:263-302, on the int8 tier's arithmetic)."""
    rng = np.random.default_rng(3)
    x = corpus(rng, 4096, 384, "uniform")
    q = x[[0, 100, 4095]].copy()
    q[1] *= f32(7.0)
    passed, wanted, _, _ = run_model(x, q, tau_rank=4, stride=16)
    assert wanted >= 3
