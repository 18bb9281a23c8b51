#!/usr/bin/env python3
"""
Generates tests/golden/*.npz from the reference (run in the build container only).

What it does
------------
* imports the *numpy-only* medium models of the reference (`theia.material`,
  `theia.units`) with `hephaistos`/`jsonschema` stubbed in `sys.modules`
  (SURVEY.md 8(c)); nothing Vulkan-related is touched. Bytecode writing is
  disabled so the read-only reference tree stays untouched.
* evaluates the models into the exact look-up tables `MediumModel.createMedium`
  would upload, for every medium used by the BASELINE configs and tests;
* records `parseMaterialFlags` results, ctypes struct sizes and the reference's
  own test data tables (tests/data/*.csv, src/theia/data/*.csv) as arrays;
* converts the reference's mesh assets into plain arrays (merged vertices +
  faces) with this repo's own loader.

The outputs are data (inputs / expected outputs). No reference source travels.
Usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np  # noqa: E402


def _stub_modules():
    hp = types.ModuleType("hephaistos")

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Anything()

        def __call__(self, *a, **k):
            return _Anything()

    hp.__getattr__ = lambda name: _Anything  # type: ignore
    sys.modules["hephaistos"] = hp
    js = types.ModuleType("jsonschema")
    js.validate = lambda *a, **k: None
    sys.modules["jsonschema"] = js


def _import_reference():
    _stub_modules()
    # import the two modules as a synthetic package so that theia/__init__.py
    # (which pulls in the Vulkan stack) is never executed
    import importlib.util

    pkg = types.ModuleType("theia")
    pkg.__path__ = [os.path.join(REF, "src", "theia")]
    sys.modules["theia"] = pkg
    mods = {}
    for name in ("units", "lookup", "material"):
        spec = importlib.util.spec_from_file_location(
            f"theia.{name}", os.path.join(REF, "src", "theia", f"{name}.py")
        )
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"theia.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
        mods[name] = mod
    return mods


PROPS = [
    "refractive_index",
    "group_velocity",
    "absorption_coef",
    "scattering_coef",
    "log_phase_function",
    "phase_sampling",
    "phase_m12",
    "phase_m22",
    "phase_m33",
    "phase_m34",
]


def dump_medium(out, key, medium):
    out[f"{key}/lambda_range"] = np.array([medium.lambda_min, medium.lambda_max], dtype=np.float64)
    for p in PROPS:
        v = getattr(medium, p)
        if v is not None:
            out[f"{key}/{p}"] = np.asarray(v, dtype=np.float64)


def main():
    os.makedirs(OUT, exist_ok=True)
    mods = _import_reference()
    mat, u = mods["material"], mods["units"]

    out = {}

    # the models lazily read their data tables through importlib.resources,
    # which the synthetic package cannot serve -> preload the class attributes
    dd = os.path.join(REF, "src", "theia", "data")
    mat.WaterBaseModel.DataTable = np.loadtxt(os.path.join(dd, "water_smith81.csv"), delimiter=",", skiprows=1)
    mat.BK7Model.TransmissionTable = np.loadtxt(os.path.join(dd, "bk7_transmission.csv"), delimiter=",", skiprows=2)

    class HomModel(mat.DispersionFreeMedium, mat.HenyeyGreensteinPhaseFunction, mat.MediumModel):
        def __init__(self, *, n, mu_a, mu_s, g):
            mat.DispersionFreeMedium.__init__(self, n=n, ng=n, mu_a=mu_a, mu_s=mu_s)
            mat.HenyeyGreensteinPhaseFunction.__init__(self, g)

        ModelName = "homogenous"

    class EnergyModel(
        mat.DispersionFreeMedium,
        mat.HenyeyGreensteinPhaseFunction,
        mat.KokhanovskyOceanWaterPhaseMatrix,
        mat.MediumModel,
    ):
        def __init__(self, a, s, g, *, n=1.33, ng=1.33):
            mat.DispersionFreeMedium.__init__(self, n=n, ng=ng, mu_a=a, mu_s=s)
            mat.HenyeyGreensteinPhaseFunction.__init__(self, g)
            mat.KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)

        ModelName = "homogenous"

    class WaterModel(mat.WaterBaseModel, mat.HenyeyGreensteinPhaseFunction, mat.MediumModel):
        def __init__(self):
            mat.WaterBaseModel.__init__(self, temperature=4.0, pressure=10_000, salinity=35.0)
            mat.HenyeyGreensteinPhaseFunction.__init__(self, 0.9)

    class WaterTestModel(
        mat.WaterBaseModel,
        mat.HenyeyGreensteinPhaseFunction,
        mat.KokhanovskyOceanWaterPhaseMatrix,
        mat.MediumModel,
    ):
        # parameters of theia/testing.py:641-656
        def __init__(self):
            mat.WaterBaseModel.__init__(self, 5.0, 1000.0, 35.0)
            mat.HenyeyGreensteinPhaseFunction.__init__(self, 0.6)
            mat.KokhanovskyOceanWaterPhaseMatrix.__init__(self, p90=0.66, theta0=0.25, alpha=4.0, xi=25.6)

        ModelName = "water"

    # example 01 (examples/01_volume_tracing.ipynb cell 5)
    dump_medium(out, "ex01", HomModel(n=1.33, mu_a=0.05 / u.m, mu_s=0.02 / u.m, g=0.2).createMedium())
    # tests/test_trace_energy.py:198-239 Crosscheck rows
    dump_medium(out, "cross_a0.05_s0.005_g0.5", EnergyModel(0.05, 0.005, 0.5).createMedium())
    dump_medium(out, "cross_a0_s0.01_g0", EnergyModel(0.0, 0.01, 0.0).createMedium())
    # tests/test_trace_energy.py:48-56 GroundTruth rows
    dump_medium(out, "truth_a0.05_s0.01_g0.9", EnergyModel(0.05, 0.01, 0.9).createMedium())
    dump_medium(out, "truth_a0_s0.005_g0", EnergyModel(0.0, 0.005, 0.0).createMedium())
    # example 02 (cell 10)
    lamMin, lamMax = 400.0 * u.nm, 800.0 * u.nm
    dump_medium(out, "ex02_water", WaterModel().createMedium(lamMin, lamMax, name="water"))
    dump_medium(out, "ex02_glass", mat.BK7Model().createMedium(lamMin, lamMax, name="glass"))
    # WaterTestModel, default range and 400-800
    dump_medium(out, "water_test", WaterTestModel().createMedium())
    dump_medium(out, "water_test_400_800", WaterTestModel().createMedium(lamMin, lamMax))
    dump_medium(out, "bk7", mat.BK7Model().createMedium())

    # flags + struct sizes
    flag_strings = ["", "B", "D", "DB", "BD", "T", "R", "TR", "RT", "V", "TRV", "Tf", "Tb", "Rf", "Rb", "L", "BDTR", "TbRf"]
    out["flags/strings"] = np.array(flag_strings)
    out["flags/values"] = np.array([int(mat.parseMaterialFlags(s)) for s in flag_strings], dtype=np.int64)
    from ctypes import sizeof

    out["sizes/medium_material"] = np.array([sizeof(mat.Medium.GLSL), sizeof(mat.Material.GLSL)], dtype=np.int64)
    out["units/c"] = np.array([1.0 * u.c, mat.speed_of_light])

    # the reference's data tables (published measurement data)
    dd = os.path.join(REF, "src", "theia", "data")
    out["data/water_smith81"] = np.loadtxt(os.path.join(dd, "water_smith81.csv"), delimiter=",", skiprows=1)
    out["data/bk7_transmission"] = np.loadtxt(os.path.join(dd, "bk7_transmission.csv"), delimiter=",", skiprows=2)
    td = os.path.join(REF, "tests", "data")
    out["testdata/log_phase_hg"] = np.loadtxt(os.path.join(td, "log_phase_hg.csv"), delimiter=",", skiprows=1)
    out["testdata/log_phase_ff"] = np.loadtxt(os.path.join(td, "log_phase_ff.csv"), delimiter=",", skiprows=1)
    out["testdata/bk7_refractive_index"] = np.loadtxt(os.path.join(td, "bk7_refractive_index.csv"), delimiter=",", skiprows=2)
    out["testdata/water_n_10C_35S"] = np.loadtxt(os.path.join(td, "water_n_10C_35S.csv"), delimiter=",", skiprows=3)

    np.savez_compressed(os.path.join(OUT, "media.npz"), **out)
    print("media.npz:", len(out), "arrays")

    # meshes through this repo's own loader
    sys.path.insert(0, REPO)
    from theia_amd.meshio import load_mesh_arrays

    meshes = {}
    for name, fn in [
        ("sphere", "sphere.stl"),
        ("torus", "torus.stl"),
        ("suzanne", "suzanne.stl"),
        ("cone", "cone.stl"),
        ("cube", "cube.ply"),
    ]:
        v, f = load_mesh_arrays(os.path.join(REF, "assets", fn))
        meshes[f"{name}/vertices"] = v.astype(np.float32)
        meshes[f"{name}/faces"] = f.astype(np.uint32)
        print(name, v.shape, f.shape)
    np.savez_compressed(os.path.join(REPO, "theia_amd", "data", "meshes.npz"), **meshes)


if __name__ == "__main__":
    main()
