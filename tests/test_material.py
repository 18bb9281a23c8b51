"""Host medium models + packing, pinned against tables produced by the
reference's own numpy models (tests/golden/media.npz, tools/gen_golden.py) and the
reference's test data (tests/data/*.csv of the reference, stored as arrays)."""
import os

import numpy as np
import pytest

import scenarios as S
import theia_amd.units as u
from theia_amd import material as M
from theia_amd.testing import WaterTestModel

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "media.npz"))
PROPS = ["refractive_index", "group_velocity", "absorption_coef", "scattering_coef", "log_phase_function",
         "phase_sampling", "phase_m12", "phase_m22", "phase_m33", "phase_m34"]


class WaterModel(M.WaterBaseModel, M.HenyeyGreensteinPhaseFunction, M.MediumModel):
    def __init__(self):
        M.WaterBaseModel.__init__(self, temperature=4.0, pressure=10_000, salinity=35.0)
        M.HenyeyGreensteinPhaseFunction.__init__(self, 0.9)


CASES = {
    "ex01": lambda: S.HomogeneousModel(n=1.33, mu_a=0.05 / u.m, mu_s=0.02 / u.m, g=0.2).createMedium(),
    "cross_a0.05_s0.005_g0.5": lambda: S.EnergyTestModel(0.05, 0.005, 0.5).createMedium(),
    "cross_a0_s0.01_g0": lambda: S.EnergyTestModel(0.0, 0.01, 0.0).createMedium(),
    "truth_a0.05_s0.01_g0.9": lambda: S.EnergyTestModel(0.05, 0.01, 0.9).createMedium(),
    "ex02_water": lambda: WaterModel().createMedium(400.0 * u.nm, 800.0 * u.nm, name="water"),
    "ex02_glass": lambda: M.BK7Model().createMedium(400.0 * u.nm, 800.0 * u.nm, name="glass"),
    "water_test": lambda: WaterTestModel().createMedium(),
    "bk7": lambda: M.BK7Model().createMedium(),
}


@pytest.mark.parametrize("key", sorted(CASES))
def test_models_reproduce_reference_tables(key):
    med = CASES[key]()
    lam = G[f"{key}/lambda_range"]
    assert (med.lambda_min, med.lambda_max) == (lam[0], lam[1])
    for p in PROPS:
        mine = getattr(med, p)
        name = f"{key}/{p}"
        if name not in G.files:
            assert mine is None, p
            continue
        ref = G[name]
        assert mine is not None and np.shape(mine) == ref.shape, p
        with np.errstate(invalid="ignore"):
            ok = np.isclose(np.asarray(mine, float), ref, rtol=1e-12, atol=1e-15) | (np.isinf(ref) & (np.asarray(mine) == ref))
        assert ok.all(), (p, np.nanmax(np.abs(np.asarray(mine, float) - ref)))


def test_henyey_greenstein_reference_csv():
    """reference tests/test_material.py:88-111 (tests/data/log_phase_hg.csv, 1e-6)"""
    data = G["testdata/log_phase_hg"]
    for col, g in enumerate([0.3, 0.0, -0.5, 0.99], start=1):
        hg = M.HenyeyGreensteinPhaseFunction(g)
        assert np.abs(data[:, col] - hg.log_phase_function(data[:, 0])).max() < 1e-6
        # normalisation and sampling consistency
        from scipy.integrate import quad

        total = quad(lambda x: np.exp(hg.log_phase_function(x)), -1.0, 1.0)[0] * 2 * np.pi
        assert abs(total - 1.0) < 1e-5
    eta = np.random.default_rng(0xC0FFEE).random(200000)
    hg = M.HenyeyGreensteinPhaseFunction(0.3)
    assert abs(hg.phase_sampling(eta).mean() - 0.3) < 5e-3  # <cos theta> = g


def test_fournier_forand_reference_csv():
    data = G["testdata/log_phase_ff"]
    model = M.FournierForandPhaseFunction(1.175, 4.065)
    assert np.all(np.abs(data[:, 1] - model.log_phase_function(data[:, 0])) < 1e-6)


def test_bk7_and_water_against_reference_measurements():
    """reference tests/test_material.py:11-63"""
    n_exp = G["testdata/bk7_refractive_index"]
    assert np.abs(n_exp[:, 1] - M.BK7Model().refractive_index(n_exp[:, 0] * u.nm)).max() < 5e-5
    w = G["testdata/water_n_10C_35S"]
    assert np.abs(w[:, 1] - M.WaterBaseModel(10.0, 0.0, 35.0).refractive_index(w[:, 0] * u.nm)).max() < 0.005
    smith = G["data/water_smith81"]
    model = M.WaterBaseModel(10.0, 0.0, 35.0)
    assert np.abs(smith[:, 1] - model.absorption_coef(smith[:, 0])).max() < 1e-6
    assert np.abs(smith[:, 2] - model.scattering_coef(smith[:, 0])).max() < 1e-6
    lam = np.linspace(300.0, 800.0, 200)
    n = model.refractive_index(lam)
    vg = 1.0 / (n - lam * np.gradient(n, 500 / (len(lam) - 1))) * u.c
    assert np.abs((vg - model.group_velocity(lam)) / vg).max() < 5e-3


def test_material_flags_match_reference():
    for s, v in zip(G["flags/strings"], G["flags/values"]):
        assert int(M.parseMaterialFlags(str(s))) == int(v), s
    assert int(M.parseMaterialFlags("TR")) == 0 and int(M.parseMaterialFlags("T")) == 0x18 and int(M.parseMaterialFlags("BD")) == 0x7B
    with pytest.raises(ValueError):
        M.parseMaterialFlags("X")


def test_struct_sizes_and_constants():
    from ctypes import sizeof

    assert [sizeof(M.Medium.GLSL), sizeof(M.Material.GLSL)] == list(G["sizes/medium_material"])
    assert M.speed_of_light == G["units/c"][1] == 0.299792458


def test_pack_layout_and_device_view(oracle_backend):
    """the packed blob, read back through the oracle's lookUp/lookUpMedium, gives the tables"""
    import oracle as orc

    water = WaterTestModel().createMedium()
    glass = M.BK7Model().createMedium(name="glass")
    mats = [M.Material("glass_water", glass, water, flags=("TR", "T")), M.Material("air_glass", None, "glass", flags="T"),
            M.Material("det", None, water, flags="BD")]
    store = M.MaterialStore(mats)
    assert set(store.media) == {"water", "glass"} and set(store.material) == {"glass_water", "air_glass", "det"}
    assert all(a % 8 == 0 for a in list(store.media.values()) + list(store.material.values()))
    raw = oracle_backend.download(store._blob.address, store.nbytes)
    base = store._blob.address
    # Material struct
    mat = M.Material.GLSL.from_buffer_copy(raw[store.material["glass_water"] - base:][:24])
    assert (mat.inside, mat.outside) == (store.media["glass"], store.media["water"])
    assert (mat.flagsInwards, mat.flagsOutwards) == (0, 0x18)
    mat = M.Material.GLSL.from_buffer_copy(raw[store.material["air_glass"] - base:][:24])
    assert (mat.inside, mat.outside, mat.flagsInwards) == (0, store.media["glass"], 0x18)
    # Medium constants through the oracle == linear interpolation of the tables (test_lookup.py:41-65, 1e-6 rel)
    out = np.zeros(4, np.float32)
    for lam in (200.0, 333.3, 512.25, 799.0, 800.0):
        orc.lib().orc_lookup_medium(store.media["water"], lam, out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)))
        x = np.linspace(200.0, 800.0, 1024)
        n = np.interp(lam, x, np.float32(water.refractive_index))
        vg = np.interp(lam, x, np.float32(water.group_velocity))
        mus = np.interp(lam, x, np.float32(water.scattering_coef))
        mua = np.interp(lam, x, np.float32(water.absorption_coef))
        assert np.allclose(out, [n, vg, mus, mua + mus], rtol=2e-6)
    # vacuum for the null medium
    orc.lib().orc_lookup_medium(0, 500.0, out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_float)))
    assert np.allclose(out, [1.0, 0.299792458, 0.0, 0.0])


def test_store_update_rules(oracle_backend):
    med = S.HomogeneousModel(n=1.33, mu_a=0.05, mu_s=0.02, g=0.2).createMedium()
    frozen = M.MaterialStore([], media=[med])
    with pytest.raises(RuntimeError):
        frozen.updateMedium(med)
    store = M.MaterialStore([], media=[med], freeze=False)
    med2 = S.HomogeneousModel(n=1.4, mu_a=0.01, mu_s=0.02, g=0.2).createMedium()
    store.updateMedium(med2)
    store.flush()
    with pytest.raises(ValueError):
        store.updateMedium(M.Medium("other", 200.0, 800.0))
    with pytest.raises(ValueError):
        M.MaterialStore([M.Material("m", "missing", None)])


@pytest.mark.parametrize("where", ["zip", "directory"])
def test_materials_on_disk(tmp_path, where):
    """saveMaterials / loadMaterials (the reference's tests/test_material.py:360-442, restated): media with tables of different
    lengths, materials that hold or name their media, an extra medium; to a zip file and to a directory"""
    import json

    import theia_amd.material as M

    gen = np.random.default_rng(0xC0FFEE)
    f32 = lambda n, scale=1.0: (gen.random(n, dtype=np.float32) * scale)
    med1 = M.Medium("med1", 400.0, 800.0, refractive_index=f32(256, 2.0), group_velocity=f32(128), absorption_coef=f32(64, 5.0),
                    scattering_coef=f32(97, 5.0), log_phase_function=f32(156), phase_sampling=f32(648, 3.14), phase_m12=f32(354),
                    phase_m22=f32(416), phase_m33=f32(422), phase_m34=f32(235))
    med2 = M.Medium("med2", 500.0, 700.0, refractive_index=f32(199, 4.5), group_velocity=f32(256, 0.8), absorption_coef=f32(145, 25.0),
                    scattering_coef=f32(263, 25.0), log_phase_function=f32(105), phase_sampling=f32(156))
    extra = M.Medium("extra", 450.0, 650.0, refractive_index=f32(199), group_velocity=f32(156, 0.8))
    mat1 = M.Material("mat1", "med1", med2, flags=("R", "Tfb"))
    mat2 = M.Material("mat2", med1, None, flags=("B", "RbTf"))
    path = tmp_path / ("materials.zip" if where == "zip" else "materials")
    if where == "directory":
        path.mkdir()
    M.saveMaterials(path, [mat1, mat2], media=[extra])
    mat, med = M.loadMaterials(path)
    assert med.keys() == {"med1", "med2", "extra"} and mat.keys() == {"mat1", "mat2"}
    for got, want in ((med["med1"], med1), (med["med2"], med2), (med["extra"], extra)):
        assert (got.lambda_min, got.lambda_max) == (want.lambda_min, want.lambda_max)
        assert got.tables().keys() == want.tables().keys()
        for key, table in want.tables().items():
            assert np.array_equal(got.tables()[key], table), key
    assert mat["mat1"].inside is med["med1"] and mat["mat1"].outside is med["med2"]
    assert mat["mat2"].inside is med["med1"] and mat["mat2"].outside is None
    for name, want in (("mat1", mat1), ("mat2", mat2)):
        assert mat[name].flagsInward == want.flagsInward and mat[name].flagsOutward == want.flagsOutward
    # what a broken tree says
    with pytest.raises(ValueError, match="unknown medium"):
        M.saveMaterials(tmp_path / "other.zip", [M.Material("m", "nowhere", None)])
    if where == "directory":
        listing = json.loads((path / "material.json").read_text())
        (path / "material.json").write_text(json.dumps(listing + [dict(listing[0])]))
        with pytest.raises(ValueError, match="Duplicate"):
            M.loadMaterials(path)
        (path / "material.json").write_text(json.dumps([{"name": "m", "inside": "gone", "outside": None, "flagsInward": 0, "flagsOutward": 0}]))
        with pytest.raises(ValueError, match="unknown medium"):
            M.loadMaterials(path)
        (path / "material.json").write_text(json.dumps([{"name": "m", "inside": None, "outside": None, "flagsInward": -1, "flagsOutward": 0}]))
        with pytest.raises(ValueError, match="negative"):
            M.loadMaterials(path)
        (path / "material.json").write_text("{")
        with pytest.raises(ValueError, match="Invalid material description"):
            M.loadMaterials(path)
        (path / "material.json").unlink()
        with pytest.raises(ValueError, match="Missing file"):
            M.loadMaterials(path)
    with pytest.raises(ValueError):
        M.loadMaterials(tmp_path / "nothing-here")


def test_sample_table_2d():
    """lookup.sampleTable2D: scattered samples of a plane come back as the plane on the regular grid (linear and cubic), rows = y"""
    from theia_amd.lookup import sampleTable2D

    gen = np.random.default_rng(3)
    pts = gen.random((4000, 2)) * [2.0, 3.0]
    data = np.column_stack([pts, 1.0 + 2.0 * pts[:, 0] - pts[:, 1]])
    x, y = np.linspace(0.2, 1.8, 16), np.linspace(0.3, 2.7, 8)
    want = 1.0 + 2.0 * x[None, :] - y[:, None]
    for mode in ("linear", "cubic"):
        table = sampleTable2D(data, 16, 8, boundaries=((0.2, 1.8), (0.3, 2.7)), mode=mode)
        assert np.asarray(table.data).shape == (8, 16) and np.allclose(table.data, want, atol=1e-5)
    assert np.asarray(sampleTable2D(data, 4, 5).data).shape == (5, 4)
    with pytest.raises(RuntimeError):
        sampleTable2D(data, 4, 4, mode="nearest")
    with pytest.raises(RuntimeError):
        sampleTable2D(data, 4, 4, boundaries=(0.0, 1.0, 2.0))
