"""The guard band of the matrix-pipe tile screen (mv-lm-icp_amd/csrc/nn_mfma.hip), checked numerically on the CPU.

The kernel skips a target point only when V = |b|^2 - 2 a.b - T, evaluated by one v_mfma_f32_32x32x16_f16 from f16 pieces, is >= 0; the
claim (tau_pieces) is that every point with true distance^2 D <= best comes out strictly negative WHATEVER the instruction's internal fp32
accumulation does, as long as its error stays below kAcc * 2^-24 * sum |terms|.  This test restates the operand construction of
build_mfma (targets, host, exact) and scan_block / tau_pieces (queries and thresholds, device, fp32) in numpy with the same roundings
(np.float16 / np.float32 are IEEE round-to-nearest-even like v_cvt_f16_f32 and the fp32 VALU), evaluates the sum of the exact products in
fp64, ADDS the worst accumulation error the model allows, and checks the sign for every (query, point) pair that must not be missed —
over blocks of every scale, queries inside and far outside them, thresholds from tiny to the whole search radius, BND margins.
It also checks that the pieces reproduce what they stand for (|b|^2, -T) to the stated residuals.  No GPU, no library call: it pins
the derivation, the GPU parity tests pin the implementation."""
import numpy as np
import pytest

F16, F32 = np.float16, np.float32
KACC = 34.0


def f16(x):
    return np.asarray(x, dtype=np.float64).astype(F16)


def split_targets(p, c, scale):
    """build_mfma: beta = (p - c) * scale in fp64; bh = rn16(beta), bl = rn16(beta - bh); n = |bh + bl|^2 split in three f16 pieces."""
    beta = (p - c) * scale
    bh = f16(beta)
    bl = f16(beta - bh.astype(np.float64))
    bt = bh.astype(np.float64) + bl.astype(np.float64)
    nn = (bt * bt).sum(axis=1)
    n1 = f16(nn); n2 = f16(nn - n1.astype(np.float64)); n3 = f16(nn - n1.astype(np.float64) - n2.astype(np.float64))
    en = np.abs(nn - n1.astype(np.float64) - n2.astype(np.float64) - n3.astype(np.float64)).max() * 1.000001 + 1e-30
    db = np.sqrt(((beta - bt) ** 2).sum(axis=1)).max()
    return beta, bh, bl, (n1, n2, n3), F32(en), db


def block_frame(p):
    """build_mfma: origin = centre of the block's box, power-of-two scale putting the points into [-127, 127]^3."""
    lo, hi = p.min(axis=0), p.max(axis=0)
    c = 0.5 * (lo + hi)
    ext = max((hi - c).max(), (c - lo).max())
    s = 0
    if ext > 0:
        s = int(np.floor(np.log2(127.0 / ext)))
        while np.ldexp(ext, s) > 127.0:
            s -= 1
    return c, np.ldexp(1.0, s)


def query_pieces(q, c, scale, rbest, mu, cloud_max, db_host):
    """scan_block + tau_pieces, in fp32 where the device computes in fp32.  Returns the B fragment's values and the bookkeeping."""
    scale_f = F32(scale)
    a = ((q - c) * scale).astype(F32)                      # (float)((L.qx - cx) * scale)
    amax = np.abs(a).max(axis=1)
    ah = a.astype(F16)
    al = (a - ah.astype(F32)).astype(F16)
    t = ah.astype(F32) + al.astype(F32)
    a2 = (t[:, 2] * t[:, 2] + (t[:, 1] * t[:, 1] + t[:, 0] * t[:, 0])).astype(F32)
    db = F32((db_host + (cloud_max + np.abs(c).sum()) * scale * 4.5e-16) * 1.000001 + 1e-30)
    dab = (F32(1.7320508) * (amax * F32(4.8e-7) + F32(1.2e-7)) + F32(cloud_max) * scale_f * F32(1.8e-15) + db).astype(F32)
    mu_s = F32(mu) * scale_f * F32(1.000001)
    rb = ((rbest.astype(F32) * scale_f + mu_s + dab) * F32(1.000001)).astype(F32)
    U = (rb * rb * F32(1.000001)).astype(F32)
    s = (U + a2).astype(F32)
    return a, amax, ah, al, a2, dab, U, s, scale_f


def tau_of(U, a2, s, amax, en):
    E = (F32(2.0 * KACC * 49160.0 / 16777216.0 + 1e-5) + F32(2.0) * en + amax * F32(3.7e-4 + 2.0 * KACC * 770.0 / 16777216.0) +
         s * F32(2.0 * KACC * 1.002 / 16777216.0 + 1.0e-6)).astype(F32)
    tau = (-((U - a2).astype(F32) + E)).astype(F32)
    tau = (tau - np.abs(tau) * F32(2.4e-7)).astype(F32)
    # a threshold too large for the pieces (-T beyond 2^27: t2 = f16(tau - 4096 t1) would overflow from |t1| = 2^15 on, where f16 spacing
    # is 32): the lane admits everything instead (tau_pieces' overflow guard)
    tau = np.where(tau > F32(-134217728.0), tau, F32(-60000.0 * 4096.0)).astype(F32)
    t1 = (tau * F32(1.0 / 4096.0)).astype(F16)
    r1 = (tau.astype(np.float64) - 4096.0 * t1.astype(np.float64)).astype(F32)     # fmaf(-4096, t1, tau): exact
    assert np.array_equal(r1.astype(np.float64), tau.astype(np.float64) - 4096.0 * t1.astype(np.float64))
    t2 = r1.astype(F16)
    r2 = (r1 - t2.astype(F32)).astype(F32)
    t3 = (r2 - np.abs(r2) * F32(1.0e-3) - F32(6.0e-8)).astype(F16)
    for piece in (t1, t2, t3):   # a non-finite piece makes V NaN / -inf-free +inf: a silent miss
        assert np.isfinite(piece.astype(np.float64)).all()
    return E, tau, t1, t2, t3


@pytest.mark.parametrize("seed", range(12))
def test_no_true_candidate_survives_the_screen_unflagged(seed):
    rng = np.random.default_rng(seed)
    extent = float(10.0 ** rng.uniform(-4, 0))                      # block sizes from 0.1 mm to 1 m
    centre = rng.uniform(-1, 1, 3) * float(10.0 ** rng.uniform(-2, 1))
    shape = np.array([1.0, rng.uniform(0.05, 1.0), rng.uniform(0.001, 1.0)])
    p = centre + rng.uniform(-1, 1, (2048, 3)) * extent * shape     # one block of 64 tiles
    if seed % 3 == 0:
        p[1024:] = p[:1024]                                          # duplicated points
    c, scale = block_frame(p)
    beta, bh, bl, (n1, n2, n3), en, db_host = split_targets(p, c, scale)
    assert np.abs(beta).max() <= 127.0 and np.abs(bh.astype(np.float64)).max() <= 128.0
    cloud_max = np.abs(p).max()
    # queries: near the points, around the block, and far outside it (up to the operand range of 4096 scaled units)
    nq = 512
    pick = rng.integers(0, len(p), nq)
    spread = np.where(rng.random(nq) < 0.6, 10.0 ** rng.uniform(-3, 0, nq), 10.0 ** rng.uniform(0, 1.4, nq))[:, None]
    q = p[pick] + rng.normal(0, 1, (nq, 3)) * extent * spread
    D = ((q[:, None, :] - p[None, :, :]) ** 2).sum(axis=2)          # true squared distances (metres^2), fp64
    dmin = np.sqrt(D.min(axis=1))
    # running best: from exactly the nearest point's distance up to far looser (the seed of an early round), sqrt rounded up as sqrt_up()
    best = (dmin * np.where(rng.random(nq) < 0.3, 1.0, 10.0 ** rng.uniform(0, 2, nq))) ** 2
    rbest = (np.sqrt(best).astype(F32) * F32(1.000001)).astype(F32)
    rbest = np.maximum(rbest, np.nextafter(np.sqrt(best).astype(F32), F32(np.inf)))   # an upper bound, like the device keeps
    for mu in (0.0, 0.02 * extent):
        a, amax, ah, al, a2, dab, U, s, scale_f = query_pieces(q, c, scale, rbest, mu, cloud_max, db_host)
        inr = amax <= 4096.0
        E, tau, t1, t2, t3 = tau_of(U, a2, s, amax, en)
        # the sum the instruction forms, from the exact products of its f16 operands (fp64 holds them exactly)
        AH, AL = ah.astype(np.float64), al.astype(np.float64)
        BH, BL = bh.astype(np.float64), bl.astype(np.float64)
        N = n1.astype(np.float64) + n2.astype(np.float64) + n3.astype(np.float64)
        T = 4096.0 * t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64)
        dot = AH @ BH.T + AH @ BL.T + AL @ BH.T                      # the three piece products per axis
        V = N[None, :] - 2.0 * dot + T[:, None]
        terms = (np.abs(n1.astype(np.float64)) + np.abs(n2.astype(np.float64)) + np.abs(n3.astype(np.float64)))[None, :] \
            + 2.0 * (np.abs(AH) @ np.abs(BH).T + np.abs(AH) @ np.abs(BL).T + np.abs(AL) @ np.abs(BH).T) \
            + (4096.0 * np.abs(t1.astype(np.float64)) + np.abs(t2.astype(np.float64)) + np.abs(t3.astype(np.float64)))[:, None]
        e_acc = KACC * 2.0 ** -24 * terms
        # the model's own bound on sum |terms| (tau_pieces' comment) must hold, or e_acc would be under-budgeted
        bound = 49160.0 + 770.0 * amax.astype(np.float64)[:, None] + 1.002 * (U.astype(np.float64) + a2.astype(np.float64))[:, None]
        clamped = tau <= F32(-60000.0 * 4096.0)     # lanes that admit everything: V = S - 2.4576e8, no guard band needed (|e_acc| < 1e3)
        chk = inr & ~clamped
        assert (terms[chk] <= bound[chk] * (1 + 1e-9)).all(), (seed, float((terms[chk] / bound[chk]).max()))
        # every pair that must not be missed: true distance <= sqrt(best) + mu
        must = np.sqrt(D) <= (np.sqrt(best) + mu)[:, None]
        must &= inr[:, None]
        assert must.any()
        worst = (V + e_acc)[must]
        assert (worst < 0.0).all(), (seed, mu, float(worst.max()))
        # ... with the margin the derivation promises (V_computed <= -E / 2)
        assert (worst <= -0.49 * E.astype(np.float64)[:, None].repeat(V.shape[1], 1)[must]).all(), (seed, mu)
        # the pieces stand for what they should: |b~|^2 to en, -T at most 1e-3 relative + 1e-7 below tau (rounded DOWN, never up)
        assert (T <= tau.astype(np.float64) + 1e-12).all()
        assert (tau.astype(np.float64) - T <= np.abs(tau.astype(np.float64)) * 2.0 ** -20 + 2e-7).all()


def test_out_of_range_and_unbounded_lanes_admit_everything():
    """A lane farther than 4096 scaled units from the block's origin, or without a finite threshold, gets zeroed direction operands and
    -T = -60000 * 4096: V = |b|^2 - 2.4e8 < 0 for every point (|b|^2 <= 49152) — and a finished lane +60000 * 4096, never a hit."""
    n_max = 3 * 128.0 ** 2
    assert n_max - 60000.0 * 4096.0 < 0
    assert 0.0 - 2.0 * 3 * 256.0 * 4096.0 * 1.001 + 60000.0 * 4096.0 > 0   # the most negative -2 a.b a finished in-range lane can reach


def test_threshold_pieces_stay_finite_up_to_the_admit_everything_guard():
    """for |tau| in [2^27, 2.4e8) the old guard (t1 only) let r1 = tau - 4096 t1 reach +-65536 and t2 round to +-inf (about
    4e-4 of the band).  With the guard at 2^27 every tau the split ever sees gives finite pieces, and the split still stands for tau
    (rounded down).  The band is reached when a lane's threshold radius is ~90-120 block half-extents (an unseeded first round with the
    cutoff near the object size on a dense cloud)."""
    rng = np.random.default_rng(0)
    # tau values up to and across the guard, dense around the former failure band and its lower edge
    mags = np.concatenate([10.0 ** rng.uniform(0, 8.4, 400000), rng.uniform(2.0 ** 26, 2.45e8, 400000), 2.0 ** 27 + rng.uniform(-70000, 70000, 200000)])
    tau = (-mags).astype(F32)
    zero = np.zeros_like(tau)
    # drive tau_of with U - a2 + E == -tau: U = -tau, a2 = 0, s = 0, amax = 0, en = 0 leaves only the constant part of E
    E, tau_c, t1, t2, t3 = tau_of((-tau).astype(F32), zero, zero, zero, F32(0.0))
    clamped = tau_c <= F32(-60000.0 * 4096.0)
    T = 4096.0 * t1.astype(np.float64) + t2.astype(np.float64) + t3.astype(np.float64)
    assert clamped.any() and (~clamped).any()
    assert (T[~clamped] <= tau_c.astype(np.float64)[~clamped] + 1e-12).all()
    assert (tau_c.astype(np.float64)[~clamped] - T[~clamped] <= np.abs(tau_c.astype(np.float64)[~clamped]) * 2.0 ** -20 + 2e-7).all()
    # the old guard really did overflow in the band (the case this test exists for)
    band = (tau <= F32(-134217728.0)) & (tau > F32(-2.4e8))
    tb = tau[band]
    t1o = (tb * F32(1.0 / 4096.0)).astype(F16)
    r1o = (tb.astype(np.float64) - 4096.0 * t1o.astype(np.float64)).astype(F32)
    with np.errstate(over="ignore"):
        assert not np.isfinite(r1o.astype(F16).astype(np.float64)).all()
