"""CPU model of the int8 tier's L2 threshold (yams_amd/csrc/scan_i8_kernel.hip, "L2 on the int8 tier").

A numpy restatement of the chain  quantised shadow -> score bound G -> per-query line under h(n) ->
integer threshold T(block, query) + a_r m_q, with the same roundings in the same places as the device code
(fp32 where the kernels use fp32, fp64 where i8_l2_thresholds_kernel does).  The property the filter's
completeness proof rests on is checked exhaustively on small shards:

    every (row, query) whose score bound reaches tau passes the integer test,   G(u, |x|^2) >= tau  =>  I >= T + a_r m_q

and G itself bounds g = q.x - |x|^2 / 2 from above.  The model also reports how much more than the true
survivors the integer test lets through, on norm distributions from "all rows unit length" to "norms within
a factor of two" — the quantity that decides whether the tier is worth taking.  Test infrastructure only."""
import numpy as np
import pytest

f32 = np.float32
EPS24 = f32(5.9604645e-8)


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def build_shadow(x):
    """shadow_build_i8_kernel: unit rows, one scale per block of 64 rows, measured residue bound."""
    n, d = x.shape
    xu = (x.astype(np.float64) / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)).astype(f32)
    nb = (n + 63) // 64
    xi = np.zeros((n, d), np.int32); s = np.zeros(nb, f32); e = np.zeros(nb, f32)
    for b in range(nb):
        blk = xu[64 * b:64 * b + 64]
        umax = np.abs(blk).max()
        sc = f32(umax / f32(127.0)); isc = f32(f32(127.0) / umax)
        q = np.clip(np.rint(blk * isc), -127, 127).astype(f32)
        res = (blk.astype(np.float64) - np.float64(sc) * q).astype(f32)
        emax = np.sqrt((res.astype(np.float64) ** 2).sum(1).max())
        xi[64 * b:64 * b + 64] = q.astype(np.int32)
        s[b] = sc
        e[b] = f32(emax * (1.0 + (d + 16.0) * 5.96e-8) + (d + 64.0) * 5.96e-8)
    return xi, s, e


def prep_queries(q):
    """prep_i8_kernel with raw (L2) queries."""
    nq, d = q.shape
    am = np.abs(q).max(1)
    t = (am / f32(127.0)).astype(f32); it = (f32(127.0) / am).astype(f32)
    qi = np.clip(np.rint(q * it[:, None]), -127, 127).astype(f32)
    qq = t[:, None] * qi
    dd = fma(-t[:, None], qi, q)
    up = f32(1.0 + (d + 16.0) * 5.9604645e-8)
    csum = (qq.astype(np.float64) ** 2).sum(1)
    c = (np.sqrt(csum).astype(f32) * up).astype(f32)
    f = (np.sqrt((dd.astype(np.float64) ** 2).sum(1)).astype(f32) * up
         + f32((d + 32.0) * 5.9604645e-8 + 1e-6) * c).astype(f32)
    return qi.astype(np.int32), t, c, f


def l2_bound(u, nsq, eps):
    """i8_l2_bound"""
    t1 = (np.sqrt(nsq).astype(f32) * u).astype(f32)
    return fma(eps, fma(f32(0.5), nsq, np.abs(t1)), fma(f32(-0.5), nsq, t1))


def spread(n, nmin):
    return np.maximum(f32(0), ((n - nmin).astype(f32) - (n * f32(2.3841858e-7)).astype(f32)).astype(f32))


def neg_threshold(A, is_, B, g):
    t = fma(-A, is_, fma(B, g, f32(2.0)))
    t = np.clip(t, f32(-1.0737418e9), f32(1.0737418e9))
    return np.trunc(t).astype(np.int64)


def run_model(x, q, tau_rank=16, stride=8):
    n, d = x.shape
    nq = q.shape[0]
    eps = f32((d + 64.0) * 5.9604645e-8)
    xi, s, e = build_shadow(x)
    qi, t, c, f = prep_queries(q)
    nsq = (x.astype(np.float64) ** 2).sum(1).astype(f32)          # rows_nsq of the bf16 shadow build
    I = xi.astype(np.int64) @ qi.astype(np.int64).T               # [n, nq] exact integer dot products
    blk = np.arange(n) // 64
    S = (s[blk][:, None] * t[None, :]).astype(f32)
    K = fma(e[blk][:, None], c[None, :], f[None, :])
    u = fma(I.astype(f32), S, K)
    G = l2_bound(u, nsq[:, None], eps)
    g_true = x.astype(np.float64) @ q.astype(np.float64).T - 0.5 * (x.astype(np.float64) ** 2).sum(1)[:, None]
    assert (G.astype(np.float64) >= g_true).all(), "G must bound g from above"

    # tau: the tau_rank-th largest score of every stride-th row (what select_tau makes of the sample pass)
    tau = np.sort(G[::stride], axis=0)[-tau_rank].astype(f32)

    # (1) norm statistics
    nrm = np.sqrt(nsq).astype(f32)
    nb = (n + 63) // 64
    nmin = np.array([nrm[64 * b:64 * b + 64].min() for b in range(nb)], f32)
    d_over_s = (spread(nrm, nmin[blk]) / s[blk]).astype(f32)
    nsq_lo, nsq_hi, dmax = nsq.min(), nsq.max(), d_over_s.max()
    W = f32(f32(dmax * f32(1.0 / 255.0)) * f32(1.0 + 9.5367432e-7))
    # (2) per-query halves (fp64 like the kernel)
    n_lo, n_hi = np.sqrt(np.float64(nsq_lo)), np.sqrt(np.float64(nsq_hi))
    u_max = 3.0 * c.astype(np.float64) + f
    tp = tau.astype(np.float64) - 1.5 * np.float64(eps) * (n_hi * u_max + 0.5 * n_hi * n_hi)
    beta = np.maximum(0.5 - tp / (n_lo * n_hi), 0.0)
    slope = 0.5 - beta
    alpha = np.minimum(tp / n_lo + slope * n_lo, tp / n_hi + slope * n_hi)
    with np.errstate(invalid="ignore", divide="ignore"):
        ns = np.sqrt(tp / slope)
        inside = (tp > 0) & (slope > 0) & (ns > n_lo) & (ns < n_hi)
        alpha = np.where(inside, np.minimum(alpha, 2.0 * np.sqrt(np.abs(tp * slope))), alpha)
    alpha = alpha - (np.abs(alpha) * 1e-12 + 1e-300)
    e_max = np.float64(e.max()) * (1.0 + 1e-6)
    A = ((alpha - f - c.astype(np.float64) * e_max) / t).astype(f32)
    A = (A - np.abs(A) * f32(1.9073486e-6)).astype(f32)
    B = (beta / t).astype(f32)
    m = np.clip(np.floor(beta / t * np.float64(W) * (1.0 - 1e-6)), 0, 2097152).astype(np.int64)
    # (3) thresholds meta and row biases
    a = np.zeros(n, np.int64)
    if W > 0:
        a = np.clip(np.floor((d_over_s / W).astype(f32) * f32(1.0 - 9.5367432e-7)), 0, 255).astype(np.int64)
    eb2 = (-nmin * f32(1.0 - 4.0531158e-6)).astype(f32)
    # filter kernel: accumulators start at -T - a_r m_q
    is_ = (f32(1.0) / s).astype(f32)
    g2 = (eb2 * is_).astype(f32)
    nt = neg_threshold(A[None, :], is_[blk][:, None], B[None, :], g2[blk][:, None])      # [n, nq] (per block, really)
    acc = I + nt - a[:, None] * m[None, :]
    passes = acc >= 0
    wanted = G >= tau[None, :]
    missed = wanted & ~passes
    assert not missed.any(), f"{missed.sum()} survivors dismissed by the integer test"
    # the gather kernel's way back: I from the log entry, then G and the per-row test
    I_back = acc - nt + a[:, None] * m[None, :]
    assert (I_back == I).all()
    return int(passes.sum()), int(wanted.sum()), float(n_hi / n_lo)


def corpus(rng, n, d, kind):
    x = rng.uniform(-1, 1, (n, d)).astype(f32)
    if kind == "unit":
        x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(f32)
    elif kind == "spread":                     # norms within a factor of two
        x = (x * rng.uniform(0.85, 1.35, (n, 1))).astype(f32)
    elif kind == "clustered":                  # close neighbours: tau' > 0, the convex branch
        centres = rng.normal(0, 1, (8, d)).astype(f32)
        x = (centres[rng.integers(0, 8, n)] + 0.05 * rng.normal(0, 1, (n, d))).astype(f32)
    return x


@pytest.mark.parametrize("kind", ["uniform", "unit", "spread", "clustered"])
def test_integer_threshold_never_dismisses_a_row_whose_bound_reaches_tau(kind):
    rng = np.random.default_rng(11)
    n, d, nq = 4096 + 37, 256, 24
    x = corpus(rng, n, d, kind)
    if kind == "clustered":
        q = (x[rng.integers(0, n, nq)] + 0.02 * rng.normal(0, 1, (nq, d))).astype(f32)
    else:
        q = corpus(rng, nq, d, kind)
        q[0] *= f32(3.0); q[1] *= f32(0.2)        # queries of other lengths than the rows
    passed, wanted, ratio = run_model(x, q)
    assert ratio <= 2.0
    # tightness: the integer test may let more through than reach tau, not orders of magnitude more
    assert passed <= 6 * wanted + 64 * nq, (kind, passed, wanted)


def test_model_on_a_ragged_tail_block_and_a_zero_query():
    rng = np.random.default_rng(5)
    x = corpus(rng, 4096 + 1, 256, "uniform")
    q = corpus(rng, 4, 256, "uniform")
    q[2] = 0                                       # valid under L2 (vec0 accepts it)
    with np.errstate(invalid="ignore", divide="ignore"):
        q[2, 0] = f32(1e-20)                       # (the model's prep needs a non-zero maximum; the kernel special-cases 0)
    run_model(x, q)


def test_fp32_accumulation_never_undercuts_the_proofs_lower_bound(oracle):
    """The completeness proof of an L2 search under fp32 accumulation (YAMS_SCAN_FLAG_L2_ACC_F32*, scan_kernels.hip
    rescore_select_kernel) rests on: the distance such arithmetic GIVES a row is at least
    sqrt(d2 * (1 - (dim + 8) u 1.01) - 1e-30) * (1 - 1.2e-7), d2 the exact squared distance.  Checked here on the CPU
    against the oracle's fp32 definitions (1, 8, 16 lanes) over dimensions 3 .. 8192 and magnitudes from subnormal
    squares to 1e15, with an exact (integer-scaled) reference for d2."""
    rng = np.random.default_rng(31)
    u = 2.0 ** -24
    worst = 0.0
    for dim in (3, 37, 64, 100, 384, 768, 1024, 4096, 8192):
        for scale in (1e-21, 1e-6, 1.0, 3.0e4, 1e15):
            rows = (rng.standard_normal((24, dim)) * scale).astype(np.float32)
            rows[0] = np.abs(rows[0])                       # all differences of one sign: the worst case for a running sum
            q = (rng.standard_normal(dim) * scale).astype(np.float32)
            q[::2] = -np.abs(q[::2]) if dim > 4 else q[::2]
            d2 = np.sum((rows.astype(np.float64) - q.astype(np.float64)) ** 2, axis=1)   # (fp64 of fp32 inputs: relative error 1e-16 * dim)
            slack = (dim + 8) * u * 1.01
            lower = np.sqrt(np.maximum(d2 * (1.0 - slack) - 1e-30, 0.0)) * (1.0 - 1.2e-7)
            for lanes in (1, 8, 16, -1, -8, -16):       # (negative: the fused multiply-add forms — one rounding fewer per term)
                got = oracle.l2_f32acc_many(rows, q, lanes).astype(np.float64)
                ok = (got >= lower) | ~np.isfinite(got)     # (inf: the row is skipped, which is never below the bound)
                assert ok.all(), (dim, scale, lanes, got[~ok][:3], lower[~ok][:3])
                fin = np.isfinite(got) & (d2 > 1e-20)     # (below: the absolute 1e-30 term carries the bound, squares are subnormal)
                if fin.any():
                    worst = max(worst, float(np.max((np.sqrt(d2[fin]) - got[fin]) / np.sqrt(d2[fin]) / (0.5 * slack))))
    assert worst < 1.0, worst   # the observed undercut stays inside the allowance (it uses a few per cent of it)
