"""The walk reads a child box's 16-bit plane indices as floats without a conversion (theia_amd/csrc/dev_walk.hpp: QRay,
qPlane): one v_perm_b32 puts the index under the exponent byte of 2^23, which IS the float 2^23 + q, and the ray's offset takes
the 2^23 back out, b' = b - 2^23 a. b' is rounded at the size of 2^23 a, so the planes a ray sees lie up to half a quantisation
step off -- inside the one whole step the builder moves every plane outwards (bvh_build.cpp: qlo / qhi; the tree's own box spans
q = 2 ... 65 533, so no plane loses that step to a clamp). This restates both sides in numpy float32 (fma emulated through
float64) and checks, for random trees, boxes and rays -- axis-parallel ones and rays starting far outside included -- that
  * the bit pattern 0x4B000000 | q is the float 2^23 + q,
  * a decoded plane lies within half a step (plus the rounding every form has) of where its index says, and
  * the box test never culls a box the exact ray reaches within [0, tmax] (conservative)."""
import numpy as np

F = np.float32
BASE = F(8388608.0)
SLACK = F(1.000001)
CLAMP = F(1.0e24)


def fma(a, b, c):  # float32 fma: the product of two float32 is exact in float64
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def test_the_bit_pattern_is_the_float():
    q = np.arange(0, 65536, dtype=np.uint32)
    assert np.array_equal((np.uint32(0x4B000000) | q).view(F), (BASE + q.astype(F)).astype(F))
    assert np.array_equal((np.uint32(0x4B000000) | q).view(F).astype(np.float64), 8388608.0 + q)


def _frame(lo, hi):
    """bvh_build.cpp: emitTree -- step and origin of a tree's frame, per axis"""
    step = ((hi - lo) / F(65531.0)).astype(F)
    step = np.maximum(step, (F(1e-7) * np.maximum(np.abs(lo), np.abs(hi)) + F(1e-30)).astype(F))
    origin = (lo - F(2.0) * step).astype(F)
    for _ in range(64):
        bad = fma(np.full_like(step, 2.0), step, origin) > lo
        if not bad.any():
            break
        origin = np.where(bad, np.nextafter(origin, F(-np.inf)), origin)
    for _ in range(64):
        bad = fma(np.full_like(step, 65533.0), step, origin) < hi
        if not bad.any():
            break
        step = np.where(bad, np.nextafter(step, F(np.inf)), step)
    return step, origin


def _qlo(v, step, origin):
    q = np.clip(np.floor((v - origin) / step), 0, 65534).astype(np.int64)
    for _ in range(4):
        over = (q > 0) & (fma(q.astype(F), step, origin) > v)
        q = np.where(over, q - 1, q)
    return np.maximum(q - 1, 0)


def _qhi(v, step, origin):
    q = np.clip(np.ceil((v - origin) / step), 0, 65534).astype(np.int64)
    for _ in range(4):
        under = (q < 65534) & (fma(q.astype(F), step, origin) < v)
        q = np.where(under, q + 1, q)
    return np.minimum(q + 1, 65535)


def _cases(rng, n):
    centre = rng.uniform(-50.0, 50.0, (n, 3))
    extent = np.exp(rng.uniform(np.log(1e-2), np.log(1e3), (n, 1))) * rng.uniform(0.2, 1.0, (n, 3))
    lo, hi = (centre - extent).astype(F), (centre + extent).astype(F)
    # child box inside the tree's box (thin and degenerate ones included)
    u = np.sort(rng.uniform(0.0, 1.0, (n, 3, 2)), axis=2)
    flat = rng.random((n, 3)) < 0.1
    u[..., 1] = np.where(flat, u[..., 0], u[..., 1])
    snap = rng.random((n, 3)) < 0.1  # boxes that touch the tree's own faces
    u[..., 0] = np.where(snap, 0.0, u[..., 0])
    clo = (lo + (hi - lo) * u[..., 0].astype(F)).astype(F)
    chi = np.maximum((lo + (hi - lo) * u[..., 1].astype(F)).astype(F), clo)
    chi = np.minimum(chi, hi)
    # rays: most aim at the child box, from inside the tree or from far away; some with exact zeros in the direction
    target = clo + (chi - clo) * rng.uniform(-0.2, 1.2, (n, 3)).astype(F)
    far = rng.random((n, 1)) < 0.3
    org = np.where(far, centre + rng.normal(size=(n, 3)) * extent * 100.0, centre + rng.uniform(-1.2, 1.2, (n, 3)) * extent).astype(F)
    d = (target - org).astype(np.float64)
    d += rng.normal(size=(n, 3)) * 1e-3 * np.linalg.norm(d, axis=1, keepdims=True) * (rng.random((n, 1)) < 0.5)
    zero = rng.random((n, 3)) < 0.08
    d = np.where(zero, 0.0, d)
    d[np.all(d == 0.0, axis=1)] = (1.0, 0.0, 0.0)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    tmax = np.where(rng.random(n) < 0.5, 1e4, np.linalg.norm(target - org, axis=1) * rng.uniform(0.5, 1.5, n)).astype(F)
    return lo, hi, clo, chi, org, d, tmax


def test_decoded_planes_and_box_test_are_conservative():
    rng = np.random.default_rng(20261004)
    n = 400_000
    lo, hi, clo, chi, org, d, tmax = _cases(rng, n)
    # bvh_build.cpp: every node box is padded by 2e-6 of its largest coordinate before it is quantised (the tree's own box too)
    pad = (F(2e-6) * np.maximum(np.abs(lo), np.abs(hi)) + F(1e-30)).astype(F)
    lo, hi = (lo - pad).astype(F), (hi + pad).astype(F)
    cpad = (F(2e-6) * np.maximum(np.abs(clo), np.abs(chi)) + F(1e-30)).astype(F)
    plo_box, phi_box = np.maximum((clo - cpad).astype(F), lo), np.minimum((chi + cpad).astype(F), hi)
    step, origin = _frame(lo, hi)
    ql, qh = _qlo(plo_box, step, origin), _qhi(phi_box, step, origin)
    assert (ql >= 0).all() and (qh <= 65535).all() and (ql < qh).all()
    # the planes the indices stand for lie outside the padded box, by one step wherever a step is more than a few ulps of the
    # coordinates (next to the tree's own faces as well: no clamp at q = 0 / 65 535 takes the step away)
    plo, phi = fma(ql.astype(F), step, origin).astype(np.float64), fma(qh.astype(F), step, origin).astype(np.float64)
    assert (plo <= plo_box).all() and (phi >= phi_box).all()
    coarse = step >= 16.0 * np.spacing(np.maximum(np.abs(lo), np.abs(hi)))
    assert coarse.mean() > 0.5
    assert (plo <= plo_box.astype(np.float64) - 0.9 * step)[coarse].all() and (phi >= phi_box.astype(np.float64) + 0.9 * step)[coarse].all()

    # dev_walk.hpp: slabInv (v_rcp_f32: 1 ulp, emulated by a random last bit) and qRay
    with np.errstate(divide="ignore"):
        inv = (F(1.0) / d).astype(F)
    wob = rng.integers(-1, 2, inv.shape)
    inv = np.where(np.isfinite(inv), np.where(wob > 0, np.nextafter(inv, F(np.inf)), np.where(wob < 0, np.nextafter(inv, F(-np.inf)), inv)), inv)
    inv = np.minimum(np.maximum(inv, -CLAMP), CLAMP)
    a = (step * inv).astype(F)
    b = fma(origin, inv, -(org * inv).astype(F))
    bm = fma(np.full_like(a, -BASE), a, b)
    assert np.isfinite(a).all() and np.isfinite(bm).all()
    neg = a < 0
    qn, qf = np.where(neg, qh, ql), np.where(neg, ql, qh)
    tn_ax = fma((BASE + qn.astype(F)).astype(F), a, bm)
    tf_ax = fma((BASE + qf.astype(F)).astype(F), a, bm)

    # a decoded plane against the plane its index stands for, in units of the step along the ray (|a|): half a step from b',
    # the rest is what the unshifted form q a + b carries as well (relative to the distance)
    exact_n = qn * a.astype(np.float64) + b.astype(np.float64)
    err = np.abs(tn_ax.astype(np.float64) - exact_n) / np.abs(a.astype(np.float64))
    allow = 0.5 + 2.0 ** -23 * (np.abs(exact_n) / np.abs(a.astype(np.float64)) + 8388608.0 * 2.0 ** -1) + 1e-3
    assert (err <= allow).all(), float((err - allow).max())

    tn = np.maximum(np.maximum(tn_ax[:, 0], tn_ax[:, 1]), np.maximum(tn_ax[:, 2], F(0.0)))
    tf = np.minimum(np.minimum(tf_ax[:, 0], tf_ax[:, 1]), np.minimum(tf_ax[:, 2], tmax))
    dev_hit = tn <= (tf * SLACK).astype(F)

    # exact: does the ray reach the (unpadded) child box within [0, tmax]?
    o64, d64 = org.astype(np.float64), d.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (clo - o64) / d64, (chi - o64) / d64
    inside = (o64 >= clo) & (o64 <= chi)
    near_ax = np.where(d64 == 0.0, np.where(inside, -np.inf, np.inf), np.minimum(t0, t1))  # a ray parallel to a slab is in it for good or never
    far_ax = np.where(d64 == 0.0, np.where(inside, np.inf, -np.inf), np.maximum(t0, t1))
    near = np.maximum(near_ax.max(axis=1), 0.0)
    farr = np.minimum(far_ax.min(axis=1), tmax.astype(np.float64))
    true_hit = near <= farr
    assert true_hit.sum() > n // 5 and (~true_hit).sum() > n // 20
    missed = true_hit & ~dev_hit
    assert not missed.any(), f"{missed.sum()} boxes culled that the ray reaches"
    # ... and not by becoming useless: most of the boxes the ray misses are culled
    assert (dev_hit & ~true_hit).sum() < 0.05 * (~true_hit).sum()
