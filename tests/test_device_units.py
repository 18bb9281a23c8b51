"""
Unit probes of the path's device functions: ONE function per element through the C-ABI probe
`thx_probe_unit` on the HIP kernels (-m gpu) and through the oracle's per-function entry points
(CPU suite), both held against closed forms. This is what the reference's shader unit tests do
with tiny test kernels (tests/shader/<name>.test.glsl): the inputs are regenerated the way the
reference draws them (a fresh `np.random.default_rng(0xC0FFEE)` per test, its tests/conftest.py:73-76),
sizes and tolerances are the reference's --

    lookUp                          tests/test_lookup.py:41-65          1e-6
    scatterDir                      tests/test_scatter.py:15-86         5e-7 on cos(theta)
    fresnelReflect                  tests/test_scatter.py:89-173        1e-4
    scatter / scatterProb           tests/test_scatter.py:176-331       5e-5 against the HG model
    rotatePolRef (frame rotation)   tests/test_polarization.py:93-164   1e-6
    alignPolRef undoes rotatePolRef tests/test_polarization.py:64-90    5e-6
    lookUpPhaseMatrix               tests/test_polarization.py:168-255  1e-3
    signBit                         tests/test_math.py:8-38             exact

-- and a last test compares HIP and oracle function by function on the same inputs: bit for bit
where only IEEE arithmetic is involved, to 2 ulp-ish where libm (sin / cos / exp) enters.
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import scenarios as S
from theia_amd import _backend
from theia_amd import material as M
from theia_amd.lookup import uploadTables
from theia_amd.testing import WaterTestModel

N = 32 * 1024
SEED = 0xC0FFEE


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def run(name, *columns, addr=0):
    """stacks the input columns (scalars or vectors per element) and probes `name` on the installed backend"""
    cols = [np.asarray(c, dtype=np.float64) for c in columns]
    rows = np.concatenate([c[:, None] if c.ndim == 1 else c for c in cols], axis=1).astype(np.float32)
    return _backend.get().probe_unit(name, rows, addr).astype(np.float64)


def frames(gen, n):
    """unit directions and a unit reference perpendicular to each, drawn like the reference's
    polarization tests do (positive-octant directions, reference = dir x random)"""
    d = unit(gen.random((n, 3)))
    ref = unit(np.cross(d, gen.random((n, 3))))
    return d, ref


def test_lookup_matches_interp(backend):
    gen = np.random.default_rng(SEED)
    table = gen.random(257).astype(np.float32)
    _keep, (addr,) = uploadTables([table])
    u = gen.random(N).astype(np.float32).astype(np.float64)  # the probe's inputs are float32
    got = run("lookup", u, np.full(N, -1.0), addr=addr)[:, 0]
    assert np.abs(got - np.interp(u, np.linspace(0.0, 1.0, table.size), table)).max() < 1e-6
    # outside [0, 1]: clamped; exactly on a grid point: that sample; null table: the default value
    edge = run("lookup", [-3.0, 7.0, 17.0 / 256.0], [0.0, 0.0, 0.0], addr=addr)[:, 0]
    assert edge[0] == table[0] and edge[1] == table[-1] and edge[2] == pytest.approx(table[17], rel=1e-6)
    assert run("lookup", [0.3], [42.0], addr=0)[0, 0] == 42.0


def test_lookup_medium_reads_the_four_tables(backend):
    gen = np.random.default_rng(SEED)
    model = WaterTestModel()
    store = M.MaterialStore([], media=[model.createMedium(400.0, 800.0)])
    lam = 400.0 + 400.0 * gen.random(2048)
    got = run("lookup_medium", lam, addr=store.media["water"])
    n = np.array([model.refractive_index(x) for x in lam])
    vg = np.array([model.group_velocity(x) for x in lam])
    mu_s = np.array([model.scattering_coef(x) for x in lam])
    mu_a = np.array([model.absorption_coef(x) for x in lam])
    # 1024-sample tables, linear interpolation (the measured absorption and scattering data have kinks between their points)
    for col, want, tol in zip(got.T, (n, vg, mu_s, mu_a + mu_s), (1e-5, 1e-5, 5e-3, 5e-3)):
        assert np.abs(col / want - 1.0).max() < tol
    assert np.allclose(run("lookup_medium", [500.0], addr=0)[0], [1.0, 0.299792458, 0.0, 0.0])  # vacuum


def test_scatter_dir_keeps_length_and_angle(backend):
    gen = np.random.default_rng(SEED)
    theta, phi = np.pi * gen.random(N), 2.0 * np.pi * gen.random(N)
    d = np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], -1)
    cos_new, phi_new = 2.0 * gen.random(N) - 1.0, 2.0 * np.pi * gen.random(N)
    out = run("scatter_dir", d, cos_new, phi_new)
    d32 = d.astype(np.float32).astype(np.float64)
    assert np.abs(np.linalg.norm(out, axis=1) - 1.0).max() < 1e-6
    assert np.abs((out * d32).sum(-1) - cos_new.astype(np.float32)).max() < 5e-7
    # the azimuth is spread evenly around the old direction (reference: 10 bins, 5 sigma)
    poles = run("scatter_dir", [[0.0, 0.0, 1.0], [0.0, 0.0, -1.0]], [0.5, 0.5], [1.0, 1.0])
    assert np.isfinite(poles).all() and np.allclose(poles[:, 2], [0.5, -0.5], atol=1e-6)


def test_fresnel_coefficients(backend):
    gen = np.random.default_rng(SEED)
    n_i, n_t = 1.0 + 0.6 * gen.random(N), 1.0 + 0.6 * gen.random(N)
    cos_i = 2.0 * gen.random(N) - 1.0
    got = run("fresnel", cos_i, n_i, n_t)
    c = np.abs(cos_i)
    sin_t = np.sqrt(np.maximum(1.0 - c * c, 0.0)) * n_i / n_t
    cos_t = np.sqrt(np.maximum(1.0 - sin_t * sin_t, 0.0))  # total reflection: cos_t = 0
    r_s = (n_i * c - n_t * cos_t) / (n_i * c + n_t * cos_t)
    r_p = (n_t * c - n_i * cos_t) / (n_t * c + n_i * cos_t)
    assert np.abs(got[:, 0] - r_s).max() < 1e-4 and np.abs(got[:, 1] - r_p).max() < 1e-4
    tir = run("fresnel", [0.2], [1.5], [1.0])[0]
    assert np.allclose(np.abs(tir), 1.0, atol=1e-6)
    bk7 = run("fresnel", [np.cos(np.pi / 4)], [1.0], [1.5168])[0]
    assert abs(0.5 * (bk7**2).sum() - 0.0516) < 0.005  # tests/test_trace.py:1196-1277


def test_scatter_follows_the_hg_model(backend):
    gen = np.random.default_rng(SEED)
    g = 0.6
    model = S.HomogeneousModel(n=1.33, mu_a=0.0, mu_s=0.1, g=g)
    store = M.MaterialStore([], media=[model.createMedium()])
    medium = store.media["homogenous"]
    a, b = unit(gen.normal(size=(N, 3))), unit(gen.normal(size=(N, 3)))
    a32, b32 = (x.astype(np.float32).astype(np.float64) for x in (a, b))
    prob = run("scatter_prob", a, b, addr=medium)[:, 0]
    want = np.exp(model.log_phase_function((a32 * b32).sum(-1)))
    assert np.abs(prob - want).max() < 5e-5
    # sampling: the reported pdf is the model's at the sampled angle, the mean cosine is g
    out = run("scatter_sample", a, gen.random(N), gen.random(N), addr=medium)
    cos_theta = (out[:, :3] * a32).sum(-1)
    assert np.abs(np.linalg.norm(out[:, :3], axis=1) - 1.0).max() < 1e-6
    assert np.abs(out[:, 3] - np.exp(model.log_phase_function(np.clip(cos_theta, -1.0, 1.0)))).max() < 2e-4  # + LUT error of cos(theta)
    assert abs(cos_theta.mean() - g) < 5e-3
    # vacuum: isotropic
    iso = run("scatter_sample", a[:4096], gen.random(4096), gen.random(4096), addr=0)
    assert np.allclose(iso[:, 3], 1.0 / (4.0 * np.pi)) and abs((iso[:, :3] * a32[:4096]).sum(-1).mean()) < 0.03
    assert run("scatter_prob", a[:1], b[:1], addr=0)[0, 0] == pytest.approx(1.0 / (4.0 * np.pi))


def test_rotate_polref_turns_the_frame(backend):
    gen = np.random.default_rng(SEED)
    phi, theta = 2.0 * np.pi * gen.random(N), np.pi * gen.random(N)
    stokes = gen.random((N, 4))
    d, ref = frames(gen, N)
    new_dir = run("scatter_dir", d, np.cos(theta), phi)
    out = run("rotate_polref", d, ref, new_dir)
    new_ref, c2, s2 = out[:, :3], out[:, 3], out[:, 4]
    d32, ref32 = (x.astype(np.float32).astype(np.float64) for x in (d, ref))
    cos_phi = (ref32 * new_ref).sum(-1)
    sin_phi = (np.cross(ref32, new_ref) * d32).sum(-1)
    assert np.abs(c2 - (2.0 * cos_phi * cos_phi - 1.0)).max() < 1e-6 and np.abs(s2 - 2.0 * cos_phi * sin_phi).max() < 1e-6
    # the new reference is the old one turned about the ray by that angle: unit length, normal to the ray
    turned = Rotation.from_rotvec(d32 * np.arctan2(sin_phi, cos_phi)[:, None]).apply(ref32)
    assert np.abs(new_ref - turned).max() < 1e-6
    assert np.abs(np.square(new_ref).sum(-1) - 1.0).max() < 1e-6 and np.abs((new_ref * d32).sum(-1)).max() < 1e-6
    # and the Stokes vector picks up the rotation by 2 phi (polarization.glsl:29-33)
    rotated = run("phase_matrix", np.ones(N), c2, s2, stokes, addr=0)  # vacuum: identity phase matrix, the rotation remains
    want = stokes.astype(np.float32).astype(np.float64)
    want[:, 1], want[:, 2] = c2 * want[:, 1] - s2 * want[:, 2], s2 * stokes.astype(np.float32)[:, 1] + c2 * want[:, 2]
    assert np.abs(rotated - want).max() < 1e-6
    # degenerate: new normal parallel to the ray keeps the frame
    same = run("rotate_polref", [[0.0, 0.0, 1.0]], [[1.0, 0.0, 0.0]], [[0.0, 0.0, 1.0]])[0]
    assert tuple(same) == (1.0, 0.0, 0.0, 1.0, 0.0)


def test_align_polref_undoes_rotate_polref(backend):
    gen = np.random.default_rng(SEED)
    d_in, d_out = unit(gen.normal(size=(N, 3))), unit(gen.normal(size=(N, 3)))
    ref = unit(np.cross(d_in, gen.normal(size=(N, 3))))
    fwd = run("rotate_polref", d_in, ref, d_out)
    back = run("align_polref", d_in, fwd[:, :3], ref)
    # rot(2 phi) followed by rot(-2 phi): the product of the two rotation blocks is the identity
    c = back[:, 0] * fwd[:, 3] - back[:, 1] * fwd[:, 4]
    s = back[:, 1] * fwd[:, 3] + back[:, 0] * fwd[:, 4]
    assert np.abs(c - 1.0).max() < 5e-6 and np.abs(s).max() < 5e-6


def test_phase_matrix_contraction(backend):
    gen = np.random.default_rng(SEED)

    class Water(M.WaterBaseModel, M.HenyeyGreensteinPhaseFunction, M.KokhanovskyOceanWaterPhaseMatrix, M.MediumModel):
        ModelName = "water"

        def __init__(self):
            M.WaterBaseModel.__init__(self, 5.0, 1000.0, 35.0)
            M.HenyeyGreensteinPhaseFunction.__init__(self, 0.6)
            M.KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)

    model = Water()
    store = M.MaterialStore([], media=[model.createMedium()])
    cos_theta = 2.0 * gen.random(N) - 1.0
    stokes = gen.random((N, 4))
    got = run("phase_matrix", cos_theta, np.ones(N), np.zeros(N), stokes, addr=store.media["water"])  # no frame rotation
    m12, m22, m33, m34 = (np.zeros(N) if (m := f(cos_theta)) is None else m
                          for f in (model.phase_m12, model.phase_m22, model.phase_m33, model.phase_m34))
    i, q, u_, v = stokes.T
    want = np.stack([i + m12 * q, m12 * i + m22 * q, m33 * u_ + m34 * v, -m34 * u_ + m33 * v], -1)
    assert np.abs(got - want).max() < 1e-3
    # with a rotation in front: (phase * rotate) * stokes
    ang = np.pi * gen.random(N)
    c2, s2 = np.cos(2 * ang), np.sin(2 * ang)
    both = run("phase_matrix", cos_theta, c2, s2, stokes, addr=store.media["water"])
    q2, u2 = c2 * q - s2 * u_, s2 * q + c2 * u_
    want = np.stack([i + m12 * q2, m12 * i + m22 * q2, m33 * u2 + m34 * v, -m34 * u2 + m33 * v], -1)
    assert np.abs(both - want).max() < 1e-3


def test_polarizer_and_sign_bit(backend):
    gen = np.random.default_rng(SEED)
    p, s = gen.random(4096) + 0.1, gen.random(4096) + 0.1
    stokes = gen.random((4096, 4))
    got = run("polarizer", p, s, stokes)
    p, s = (x.astype(np.float32).astype(np.float64) for x in (p, s))
    st = stokes.astype(np.float32).astype(np.float64)
    m12, m33 = (p * p - s * s) / (p * p + s * s), 2.0 * p * s / (p * p + s * s)
    want = np.stack([st[:, 0] + m12 * st[:, 1], m12 * st[:, 0] + st[:, 1], m33 * st[:, 2], m33 * st[:, 3]], -1)
    assert np.abs(got - want).max() < 1e-6
    values = np.concatenate([gen.normal(size=1000), [0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45]])
    assert np.array_equal(run("sign_bit", values)[:, 0], np.copysign(1.0, values.astype(np.float32)))


@pytest.mark.gpu
def test_hip_functions_equal_the_oracles(hip):
    """function by function on the same inputs: identical bits where the function is IEEE arithmetic +
    sqrt / division only (both sides are built without contraction), a few ulp where sin / cos / exp of
    two different math libraries enter"""
    from oracle_backend import OracleBackend

    gen = np.random.default_rng(SEED)
    n = 4096
    table = gen.random(300).astype(np.float32)
    _keep, (taddr,) = uploadTables([table])
    store = M.MaterialStore([], media=[WaterTestModel().createMedium()])
    medium = store.media["water"]
    d, ref = frames(gen, n)
    other = unit(gen.normal(size=(n, 3)))
    stokes = gen.random((n, 4))
    angle = np.pi * gen.random(n)
    cases = {
        "lookup": ((gen.random(n), np.zeros(n)), taddr, 0.0),
        "lookup_medium": ((400.0 + 400.0 * gen.random(n),), medium, 0.0),
        "fresnel": ((2.0 * gen.random(n) - 1.0, 1.0 + gen.random(n), 1.0 + gen.random(n)), 0, 0.0),
        "rotate_polref": ((d, ref, other), 0, 0.0),
        "align_polref": ((d, ref, unit(np.cross(d, other))), 0, 0.0),
        "polarizer": ((gen.random(n) + 0.1, gen.random(n) + 0.1, stokes), 0, 0.0),
        "sign_bit": ((gen.normal(size=n),), 0, 0.0),
        "phase_matrix": ((2.0 * gen.random(n) - 1.0, np.cos(angle), np.sin(angle), stokes), medium, 0.0),
        "scatter_dir": ((d, 2.0 * gen.random(n) - 1.0, 2.0 * np.pi * gen.random(n)), 0, 4e-7),       # sin / cos of phi
        "scatter_prob": ((d, other), medium, 1e-6),                                                  # exp
        "scatter_sample": ((d, gen.random(n), gen.random(n)), medium, 1e-6),
    }
    oracle = OracleBackend(threads=1)
    prev = _backend._backend
    try:
        for name, (cols, addr, tol) in cases.items():
            _backend._backend = hip
            got = run(name, *cols, addr=addr)
            _backend._backend = oracle
            want = run(name, *cols, addr=addr)
            if tol == 0.0:
                assert np.array_equal(got.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32)), name
            else:
                assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), name
    finally:
        _backend._backend = prev
        oracle.close()


@pytest.mark.parametrize("n_hist,n_bins", [(1, 7), (63, 1), (64, 30), (65, 100), (256, 100), (777, 33), (4096, 8)])
def test_histogram_reducer_on_its_own(backend, n_hist, n_bins):
    """`HistogramReducer` (response.py:1065-1197, estimator.reduce.glsl:17-35): out[b] = norm * compensated sum over the
    histograms of bin b. (i) Random slabs spanning 30 orders of magnitude with both signs, denormals and exact zeros: within
    Kahan's bound 2 eps sum|x| of the float64 sum, whatever the number of histograms. (ii) What compensation is for -- one
    large slab and many small ones: within an ulp of the float64 sum where a plain fp32 loop is off by dozens. Same bounds
    for the oracle's serial loop and for HIP (one wavefront per bin: Kahan partials per lane, fixed butterfly)."""
    eps = float(np.finfo(np.float32).eps)
    gen = np.random.default_rng(SEED + 17 * n_hist + n_bins)
    mag = 10.0 ** gen.uniform(-36.0, -6.0, (n_hist, n_bins))
    slabs = (mag * gen.choice([-1.0, 1.0], (n_hist, n_bins))).astype(np.float32)
    slabs[gen.random((n_hist, n_bins)) < 0.1] = 0.0
    slabs.reshape(-1)[:: max(1, slabs.size // 11)] = np.float32(1e-41)  # denormals
    norm = np.float32(0.37)
    got = _backend.get().reduce_histograms(slabs, norm).astype(np.float64)
    exact = slabs.astype(np.float64).sum(axis=0) * np.float64(norm)
    scale = np.abs(slabs.astype(np.float64)).sum(axis=0) * np.float64(norm)
    assert (np.abs(got - exact) <= 2 * eps * scale + eps * np.abs(exact) + 1e-44).all(), float((np.abs(got - exact) / np.maximum(scale, 1e-300)).max() / eps)

    big = np.full((n_hist, n_bins), 0.1, np.float32)
    big[0, :] = np.float32(1.0e4) * (1 + np.arange(n_bins, dtype=np.float32))
    got = _backend.get().reduce_histograms(big, 1.0).astype(np.float64)
    exact = big.astype(np.float64).sum(axis=0)
    assert (np.abs(got - exact) <= 1.01 * np.spacing(exact.astype(np.float32)).astype(np.float64)).all()
    if n_hist >= 777:
        plain = np.zeros(n_bins, np.float32)
        for h in big:
            plain += h
        assert np.abs(plain.astype(np.float64) - exact).min() > 20 * np.abs(got - exact).max()


def test_histogram_reducer_edge_cases(backend):
    be = _backend.get()
    inf = np.float32(np.inf)
    slabs = np.zeros((5, 4), np.float32)
    slabs[2, 0], slabs[3, 1], slabs[1, 2], slabs[4, 2] = inf, -inf, inf, -inf
    out = be.reduce_histograms(slabs, 2.0)
    # an infinite slab poisons its bin and no other: the compensation term (t - result) - y is inf - inf = NaN from there on, in the
    # shader's loop as here -- the bin comes out non-finite, never as a plausible number
    assert not np.isfinite(out[:3]).any() and out[3] == 0.0
    assert np.array_equal(be.reduce_histograms(np.full((300, 3), 0.1, np.float32), 1.0), np.full(3, np.float32(30.0)))  # 300 x 0.1f, compensated
    assert np.array_equal(be.reduce_histograms(np.zeros((0, 6), np.float32), 1.0), np.zeros(6, np.float32))  # nothing to sum
    one = np.arange(9, dtype=np.float32)[None, :]
    assert np.array_equal(be.reduce_histograms(one, 0.5), one[0] * np.float32(0.5))
