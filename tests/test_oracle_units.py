"""
Closed-form pins of the oracle's restated shader functions, mirroring the
reference's shader unit tests (inputs regenerated with
np.random.default_rng(0xC0FFEE), tests/conftest.py:73-76 of the reference;
tolerances are the reference's).
"""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
import scenarios as S
from theia_amd import material as M

L = orc.lib()
FP = C.POINTER(C.c_float)


def fp(a):
    return a.ctypes.data_as(FP)


@pytest.fixture
def rng():
    return np.random.default_rng(0xC0FFEE)


def test_sign_bit():
    """reference tests/test_math.py:8-38"""
    for x, e in [(1.5, 1.0), (-2.0, -1.0), (0.0, 1.0), (-0.0, -1.0), (np.inf, 1.0), (-np.inf, -1.0)]:
        assert L.orc_sign_bit(x) == e


def test_lookup_against_interp(rng, oracle_backend):
    """reference tests/test_lookup.py:41-65 (1e-6)"""
    data = rng.random(257).astype(np.float32)
    from theia_amd.lookup import uploadTables

    blob, (adr,) = uploadTables([data])
    u = rng.random(4000).astype(np.float32)
    got = np.array([L.orc_lookup(adr, float(x), -1.0) for x in u])
    want = np.interp(u, np.linspace(0, 1, 257), data)
    assert np.abs(got - want).max() < 1e-6
    assert L.orc_lookup(adr, -3.0, 0.0) == data[0] and L.orc_lookup(adr, 7.0, 0.0) == data[-1]  # clamped
    assert L.orc_lookup(0, 0.3, 42.0) == 42.0  # null table -> default
    for k in (0, 17, 256):  # exactly on grid points: hi == lo
        assert L.orc_lookup(adr, k / 256.0, 0.0) == pytest.approx(data[k], rel=1e-6)


def test_scatter_dir(rng):
    """reference tests/test_scatter.py:15-86: unit length and cos(theta) preserved (5e-7 -> fp32 1e-6)"""
    n = 8192
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ct = rng.uniform(-1, 1, n).astype(np.float32)
    phi = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    out = np.zeros((n, 3), np.float32)
    for i in range(n):
        L.orc_scatter_dir(fp(d[i]), float(ct[i]), float(phi[i]), fp(out[i]))
    assert np.abs(np.linalg.norm(out, axis=1) - 1.0).max() < 1e-6
    assert np.abs((out * d).sum(1) - ct).max() < 2e-6
    # degenerate poles of the local frame
    for z in (1.0, -1.0):
        v = np.array([0, 0, z], np.float32)
        o = np.zeros(3, np.float32)
        L.orc_scatter_dir(fp(v), 0.5, 1.0, fp(o))
        assert np.isfinite(o).all() and abs(np.dot(o, v) - 0.5) < 1e-6


def test_fresnel(rng):
    """reference tests/test_scatter.py:89-173 (1e-4) against the textbook Fresnel equations"""
    n = 4096
    cos_i = rng.uniform(-1, 1, n)
    n_i = rng.uniform(1.0, 1.6, n)
    n_t = rng.uniform(1.0, 1.6, n)
    out = np.zeros(2, np.float32)
    for ci, a, b in zip(cos_i, n_i, n_t):
        L.orc_fresnel(float(ci), float(a), float(b), fp(out))
        c = abs(ci)
        st = np.sqrt(max(1 - c * c, 0)) * a / b
        ctt = np.sqrt(max(1 - st * st, 0.0))
        rs = (a * c - b * ctt) / (a * c + b * ctt)
        rp = (b * c - a * ctt) / (b * c + a * ctt)
        assert abs(out[0] - rs) < 1e-4 and abs(out[1] - rp) < 1e-4
    # total internal reflection: |r| = 1
    L.orc_fresnel(0.2, 1.5, 1.0, fp(out))
    assert np.allclose(np.abs(out), 1.0, atol=1e-6)
    # 45 deg on BK7 from air: R = 0.0516 +- 0.005 (reference tests/test_trace.py:1196-1277)
    L.orc_fresnel(float(np.cos(np.pi / 4)), 1.0, 1.5168, fp(out))
    assert abs(0.5 * (out[0] ** 2 + out[1] ** 2) - 0.0516) < 0.005


def test_scatter_prob_and_sampling_follow_hg(rng, oracle_backend):
    """reference tests/test_scatter.py:176-331 (5e-5 on probabilities)"""
    g = 0.7
    model = S.HomogeneousModel(n=1.33, mu_a=0.0, mu_s=0.1, g=g)
    store = M.MaterialStore([], media=[model.createMedium()])
    med = store.media["homogenous"]
    n = 4096
    a = rng.normal(size=(n, 3)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3)).astype(np.float32)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    got = np.array([L.orc_scatter_prob(med, fp(a[i]), fp(b[i])) for i in range(n)])
    want = np.exp(model.log_phase_function((a * b).sum(1).astype(np.float64)))
    # the device evaluates a 1024-sample linear LUT of log p: discretisation error < 5e-4 relative
    assert np.abs(got / want - 1).max() < 5e-4
    assert np.abs(got - want).mean() < 5e-5
    # sampled directions: reported pdf == scatterProb of the sampled direction; <cos> == g
    u0, u1 = rng.random(n).astype(np.float32), rng.random(n).astype(np.float32)
    out = np.zeros(3, np.float32)
    p = C.c_float()
    cos = []
    for i in range(n):
        L.orc_scatter_sample(med, fp(a[i]), float(u0[i]), float(u1[i]), fp(out), C.byref(p))
        q = L.orc_scatter_prob(med, fp(a[i]), fp(out))
        assert abs(p.value - q) <= 2e-3 * max(q, 1e-3)
        cos.append(float(np.dot(out, a[i])))
    assert abs(np.mean(cos) - g) < 0.02
    # vacuum / missing tables -> isotropic
    assert L.orc_scatter_prob(0, fp(a[0]), fp(b[0])) == pytest.approx(1 / (4 * np.pi))


def test_sphere_target_guide(rng):
    """reference tests/test_target_guide.py:38-72: pdf = 1/(2 pi (1-cos_max)) inside the cone, 0 outside"""
    g = orc.GuideParams()
    g.position[:] = (1.0, -2.0, 0.5)
    g.radius = 2.0
    out2, out5 = np.zeros(2, np.float32), np.zeros(5, np.float32)
    c = np.array(g.position[:], np.float32)
    for _ in range(2000):
        obs = (c + rng.normal(size=3) * 6).astype(np.float32)
        d = float(np.linalg.norm(obs - c))
        L.orc_sample_guide(C.byref(g), fp(obs), float(rng.random()), float(rng.random()), fp(out5))
        if d <= 2.0:
            assert out5[4] == 0.0
            continue
        cos_max = np.sqrt(1 - (2.0 / d) ** 2)
        pdf = 1 / (2 * np.pi * (1 - cos_max))
        assert out5[4] == pytest.approx(pdf, rel=2e-3)
        assert np.linalg.norm(out5[:3]) == pytest.approx(1.0, abs=1e-5)
        assert np.dot(out5[:3], (c - obs) / d) >= cos_max - 1e-5  # inside the cone
        assert out5[3] == pytest.approx(d + 2.0, rel=1e-6)
        L.orc_eval_guide(C.byref(g), fp(obs), fp(out5[:3].copy()), fp(out2))
        assert out2[1] == pytest.approx(pdf, rel=2e-3)  # evaluating the sampled direction
        away = ((obs - c) / d).astype(np.float32)
        L.orc_eval_guide(C.byref(g), fp(obs), fp(away), fp(out2))
        assert out2[1] == 0.0


def test_polarization_helpers(rng):
    """reference tests/test_polarization.py:14-164: frame rotation keeps S0/S3 and the
    degree of polarization; new reference is perpendicular to both directions"""
    for _ in range(500):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        nw = rng.normal(size=3); nw /= np.linalg.norm(nw)
        ref = np.cross(d, rng.normal(size=3)); ref /= np.linalg.norm(ref)
        d, nw, ref = (x.astype(np.float32) for x in (d, nw, ref))
        out_ref, cs = np.zeros(3, np.float32), np.zeros(2, np.float32)
        L.orc_rotate_polref(fp(d), fp(ref), fp(nw), fp(out_ref), fp(cs))
        assert abs(np.dot(out_ref, d)) < 1e-5 and abs(np.dot(out_ref, nw)) < 1e-5
        assert np.linalg.norm(out_ref) == pytest.approx(1.0, abs=1e-5)
        assert cs[0] ** 2 + cs[1] ** 2 == pytest.approx(1.0, abs=1e-4)
        phi = np.arctan2(np.dot(np.cross(ref, out_ref), d), np.dot(ref, out_ref))
        assert cs[0] == pytest.approx(np.cos(2 * phi), abs=1e-4) and cs[1] == pytest.approx(np.sin(2 * phi), abs=1e-4)
    # dir == new: identity branch
    d = np.array([0, 0, 1], np.float32)
    ref = np.array([1, 0, 0], np.float32)
    out_ref, cs = np.zeros(3, np.float32), np.zeros(2, np.float32)
    L.orc_rotate_polref(fp(d), fp(ref), fp(d), fp(out_ref), fp(cs))
    assert np.array_equal(out_ref, ref) and tuple(cs) == (1.0, 0.0)
    # polarizer: column-major matrix of polarization.glsl:109-120
    s_in = np.array([1.0, 0.3, -0.2, 0.1], np.float32)
    s_out = np.zeros(4, np.float32)
    p, s = 0.8, 0.3
    L.orc_polarizer(p, s, fp(s_in), fp(s_out))
    att = p * p + s * s
    m12, m33 = (p * p - s * s) / att, 2 * p * s / att
    want = np.array([[1, m12, 0, 0], [m12, 1, 0, 0], [0, 0, m33, 0], [0, 0, 0, m33]]) @ s_in
    assert np.allclose(s_out, want, atol=1e-6)


def test_phase_matrix_contraction(rng, oracle_backend):
    """(phase * rotate) * stokes == explicit 4x4 products built from the tables
    (reference tests/test_polarization.py:168-255)"""
    from theia_amd.testing import WaterTestModel

    model = WaterTestModel()
    store = M.MaterialStore([], media=[model.createMedium()])
    med = store.media["water"]
    for _ in range(300):
        ct = float(rng.uniform(-1, 1))
        ang = float(rng.uniform(0, np.pi))
        c, s = np.cos(2 * ang), np.sin(2 * ang)
        v = np.array([1.0, *rng.uniform(-0.5, 0.5, 3)], np.float32)
        out = np.zeros(4, np.float32)
        L.orc_phase_matrix_apply(med, ct, float(c), float(s), fp(v), fp(out))
        m12, m22, m33 = model.phase_m12(ct), model.phase_m22(ct), model.phase_m33(ct)
        P = np.array([[1, m12, 0, 0], [m12, m22, 0, 0], [0, 0, m33, 0.0], [0, 0, -0.0, m33]])
        R = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]])
        assert np.allclose(out, P @ R @ v, atol=2e-3)
    # vacuum -> identity phase matrix
    out = np.zeros(4, np.float32)
    v = np.array([1, 0.2, 0.3, 0.4], np.float32)
    L.orc_phase_matrix_apply(0, 0.3, 1.0, 0.0, fp(v), fp(out))
    assert np.allclose(out, v)
