"""Host-side mirror of the reference's module API: signatures, formulas, stage
parameters, pipeline protocol, error behaviour. The host-only
checks run on the oracle stand-in; the flows of the three example notebooks (and the event
counters) run on the oracle and, with -m gpu, on the HIP library."""
import numpy as np
import pytest

import scenarios as S
import theia_amd.units as u
from theia_amd.hp import pipeline as pl
from theia_amd.light import ConstWavelengthSource, SphericalLightSource, UniformWavelengthSource
from theia_amd.random import PhiloxRNG
from theia_amd.response import EmptyResponse, HistogramHitResponse, HitResponse, UniformValueResponse, ValueResponse
from theia_amd.scene import RectBBox, Transform
from theia_amd.target import SphereTarget, SphereTargetGuide
from theia_amd.trace import EventStatisticCallback, SceneForwardTracer, VolumeForwardTracer


def test_alias_packages_resolve():
    import hephaistos as hp
    import hephaistos.pipeline as hpl
    import theia
    import theia.units as tu
    from hephaistos.glsl import vec3
    from theia.material import MaterialStore
    from theia.trace import SceneForwardTracer as T1, VolumeForwardTracer as T2

    assert T1 is SceneForwardTracer and T2 is VolumeForwardTracer and tu.c is u.c
    assert hpl.Pipeline is pl.Pipeline and hp.isRaytracingEnabled()
    assert tuple(vec3(1, 2, 3)) == (1.0, 2.0, 3.0)


def test_units():
    assert (1.0, 2.0) * u.km == (1000.0, 2000.0) and 1.0 * u.c == 0.299792458
    assert 0.05 / u.m == 0.05 and 5.0 * u.per_km == 0.005 and 50.0 * u.cm == 0.5
    assert np.allclose(np.array([1.0, 2.0]) * u.deg, np.deg2rad([1.0, 2.0]))
    assert u.convert(1500.0, u.um) == 1.5


def test_stage_parameter_protocol():
    light = SphericalLightSource(position=(1.0, 2.0, 3.0), timeRange=(0.0, 10.0), budget=5.0)
    assert light.getParams() == {"position": (1.0, 2.0, 3.0), "timeRange": (0.0, 10.0), "budget": 5.0}
    light.position = (4.0, 5.0, 6.0)  # attribute access to public fields
    assert light.getParam("position") == (4.0, 5.0, 6.0)
    light.setParams(budget=7.0, unknown=1)  # unknown names are ignored
    assert light.budget == 7.0
    with pytest.raises(AttributeError):
        light._contribFwd  # private fields are not attributes
    light.update(1)  # _finishParams fills the private fields, config 1 only
    assert light._getParamsBlock("LightParams", 1)._contribFwd == 7.0
    assert light._getParamsBlock("LightParams", 0)._contribFwd == 0.0
    wl = UniformWavelengthSource(lambdaRange=(300.0, 700.0), normalize=False)
    wl.update(0)
    assert wl._getParamsBlock("WavelengthParams", 0)._contrib == 400.0 and wl.nRNGSamples == 1
    tgt = SphereTarget(position=(-1.0, 0.0, 0.0), radius=0.5)
    tgt.update(0)
    b = tgt._getParamsBlock("TargetParams", 0)
    assert b._hemisphereProb == pytest.approx(1 / (2 * np.pi * 0.25)) and tuple(b._invPos) == (2.0, 0.0, 0.0) and b._invRad == 2.0
    rng = PhiloxRNG(key=(1 << 40) + 5, offset=(1 << 100) + 3)
    assert rng.key == (1 << 40) + 5 and rng.offset == (1 << 100) + 3


def test_tracer_formulas_match_reference(oracle_backend):
    # VolumeForwardTracer, trace.py:623-634: nScattering 10, MIS on, direct on
    sc = S.config1_volume(1000)
    tr = sc.tracer
    assert tr.maxHitsPerThread == 21 and tr.nRNGSamples == 3 + 0 + 7 * 9 and tr.maxHits == 21 * 1000
    assert tr.normalization == 1e-3 and tr.capacity == 1000 and tr.blockSize == 128 and not tr.polarized
    assert [n for n, _ in tr.collectStages()] == ["rng", "photons", "lightSource", "target", "tracer", "callback", "response"]
    assert tr.collectStages(useOriginalNames=True)[4][0] == "Volume Forward Tracer"
    sc = S.config1_volume(1000, disableTargetSampling=True, disableDirectLighting=True)
    assert sc.tracer.maxHitsPerThread == 10 and sc.tracer.nRNGSamples == 3 + 3 * 10
    # SceneForwardTracer, trace.py:1176-1187
    sc = S.config2_sphere(2048)
    tr = sc.tracer
    assert tr.maxHitsPerThread == 9 * 2 + 1 and tr.nRNGSamples == 3 + 1 + (4 + 2) * 10
    assert [n for n, _ in tr.collectStages()] == ["rng", "photons", "lightSource", "guide", "tracer", "callback", "response"]
    sc = S.config2_sphere(2048, guide=False)
    assert sc.tracer.maxHitsPerThread == 10 and sc.tracer.nRNGSamples == 3 + 1 + 4 * 10
    assert "guide" not in [n for n, _ in sc.tracer.collectStages()]
    # batch size rules, trace.py:399-408
    tr = sc.tracer
    tr.batchSize = 100
    assert tr.getParam("_groupCountX") == 1 and tr.normalization == 0.01
    with pytest.raises(ValueError):
        tr.batchSize = 4096
    with pytest.raises(ValueError):
        tr.batchSize = -1


def test_unsupported_components_raise_clearly(oracle_backend):
    sc = S.config1_volume(100)

    class MyValue(ValueResponse):
        pass

    with pytest.raises(NotImplementedError, match="UniformValueResponse"):
        HistogramHitResponse(MyValue())
    class MyResponse(HitResponse):
        pass

    with pytest.raises(NotImplementedError, match="HistogramHitResponse"):
        VolumeForwardTracer(10, sc.light, sc.target, sc.wavelength, MyResponse(), sc.rng, medium=0)
    VolumeForwardTracer(10, sc.light, sc.target, sc.wavelength, EmptyResponse(), sc.rng, medium=0)  # response.empty.glsl
    with pytest.raises(NotImplementedError, match="SphereTarget"):
        VolumeForwardTracer(10, sc.light, SphereTargetGuide(), sc.wavelength, sc.response, sc.rng, medium=0)
    with pytest.raises(NotImplementedError):
        HistogramHitResponse(UniformValueResponse(), nBins=10000)


def test_pipeline_and_scheduler_example01_flow(backend):
    """examples/01_volume_tracing.ipynb cells 24-38 on BASELINE config 1 (plumbing, CPU)"""
    sc = S.config1_volume(20_000)
    pipeline = pl.Pipeline(sc.tracer.collectStages())
    assert "lightSource__budget" in pipeline.getParams() and "tracer__scatterCoefficient" in pipeline.getParams()
    hists, cfgs = [], []

    def process(config, batch, args):
        hists.append(sc.response.result(config).copy())
        cfgs.append((config, batch))

    scheduler = pl.PipelineScheduler(pipeline, processFn=process)
    scheduler.schedule([{}] * 3)
    scheduler.schedule([{"lightSource__budget": 2.0e5}])
    scheduler.wait()
    scheduler.destroy()
    assert cfgs == [(0, 0), (1, 1), (0, 2), (1, 3)] and scheduler.tasksFinished == 4
    assert sc.rng.offset == 4 * sc.tracer.nRNGSamples  # advanced once per batch
    tot = [h.sum() for h in hists]
    assert all(t > 0 for t in tot) and not np.array_equal(hists[0], hists[1])  # different random numbers
    assert abs(tot[1] / tot[0] - 1) < 0.2 and abs(tot[3] / tot[0] - 2) < 0.4  # budget doubled
    with pytest.raises(ValueError):
        pipeline.setParams(nosuchstage__x=1)


def test_scheduler_surfaces_errors(oracle_backend):
    sc = S.config1_volume(100)
    pipeline = pl.Pipeline(sc.tracer.collectStages())

    def boom(config, batch, args):
        raise RuntimeError("boom")

    scheduler = pl.PipelineScheduler(pipeline, processFn=boom)
    scheduler.schedule([{}])
    with pytest.raises(RuntimeError, match="boom"):
        scheduler.wait()
    scheduler.destroy()


def test_statistics_callback_counts(backend):
    """event counters add up (reference tests/test_trace.py:947-1045)"""
    n = 5000
    sc = S.config2_sphere(n, stats=True)
    S.run_batches(sc, 1)
    cb = sc.callback
    assert cb.created == n and cb.error == 0 and cb.mismatch == 0
    assert cb.lost + cb.decayed + cb.absorbed + cb.maxIter == n  # every photon ends exactly once
    assert cb.traversal["steps"] == cb.scattered + cb.hit + cb.detected + cb.volume + cb.lost + cb.decayed + cb.absorbed + cb.missed + cb.error
    assert "created: 5,000" in str(cb)
    cb.reset()
    assert cb.created == 0


def test_transform_and_bbox():
    t = Transform.TRS(scale=2.0, rotate=(0.0, 0.0, 1.0, 90.0), translate=(1.0, 0.0, 0.0))
    p = t.apply(np.array([[1.0, 0.0, 0.0]]))
    assert np.allclose(p, [[1.0, 2.0, 0.0]])
    assert np.allclose(t.inverse().apply(p), [[1.0, 0.0, 0.0]])
    assert np.allclose((Transform.Translation(1, 2, 3) @ Transform.Scale(2)).numpy(), [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3]])
    with pytest.raises(ValueError):
        Transform(np.zeros((4, 4)))
    box = RectBBox((-1.0, -2.0, -3.0), (1.0, 2.0, 3.0))
    assert box.diagonal == pytest.approx(np.sqrt(4 + 16 + 36))
    moved = box.transform(Transform.Translation(10, 0, 0))
    assert moved.lowerCorner == (9.0, -2.0, -3.0) and moved.upperCorner == (11.0, 2.0, 3.0)
    v = Transform.View(position=(0, 0, 5), direction=(0, 0, -1), up=(0, 1, 0)).numpy()
    assert np.allclose(v[:, 2], [0, 0, -1]) and np.allclose(v[:, 3], [0, 0, 5])


def test_mesh_io_roundtrip(tmp_path):
    from theia_amd.meshio import icosphere, load_mesh_arrays, save_stl

    v, f = icosphere(3)
    assert v.shape == (642, 6) and f.shape == (1280, 3)
    assert np.allclose(np.linalg.norm(v[:, :3], axis=1), 1.0, atol=1e-6)
    assert ((v[:, :3] * v[:, 3:]).sum(1) > 0.99).all()  # vertex normals point outwards
    path = tmp_path / "ico.stl"
    save_stl(path, v, f)
    v2, f2 = load_mesh_arrays(path)
    assert v2.shape == v.shape and f2.shape == f.shape  # vertices merged again
    ref = S.bundled_mesh("sphere")
    assert ref.vertices.shape == (642, 6) and ref.indices.shape == (1280, 3)
    r = np.linalg.norm(ref.vertices[:, :3], axis=1)
    assert r.min() > 0.99 and r.max() < 1.001
    (tmp_path / "t.ply").write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                                    "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 2\n")
    pv, pf = load_mesh_arrays(tmp_path / "t.ply")
    assert pv.shape == (3, 6) and pf.tolist() == [[0, 1, 2]] and np.allclose(pv[:, 3:], [[0, 0, 1]] * 3)
    (tmp_path / "x.obj").write_text("v 0 0 0\n")
    with pytest.raises(ValueError):
        load_mesh_arrays(tmp_path / "x.obj")


def test_scene_validation(oracle_backend):
    from theia_amd.material import Material, MaterialStore
    from theia_amd.scene import MeshStore, Scene

    store = MaterialStore([Material("m", None, None, flags="B")])
    meshes = MeshStore({"sphere": S.bundled_mesh("sphere")})
    with pytest.raises(ValueError, match="empty"):
        Scene([], store.material)
    inst = meshes.createInstance("sphere", "nope")
    with pytest.raises(ValueError, match="Unknown material"):
        Scene([inst], store.material)
    inst = meshes.createInstance("sphere", "m", Transform.Translation(1, 2, 3), scale=2.0, detectorId=7)
    assert inst.instance.customIndex == 7 and np.allclose(inst.transform.numpy(), [[2, 0, 0, 1], [0, 2, 0, 2], [0, 0, 2, 3]])
    assert np.allclose(inst.bbox.lowerCorner, [-1, 0, 1], atol=1e-2)
    sc = Scene([inst], store, medium=0)
    assert sc.medium == 0 and sc.bbox.diagonal == pytest.approx(np.sqrt(3) * 2000.0)


def test_example02_scene_tracing_flow(backend):
    """examples/02_scene_tracing.ipynb (cells 3-10, reduced batch): media by name, vacuum source medium,
    nested glass spheres around the light, detector sphere with a target guide, scheduler"""
    import hephaistos.pipeline as hpl
    import theia
    from theia.light import SphericalLightSource, UniformWavelengthSource
    from theia.material import BK7Model, HenyeyGreensteinPhaseFunction, Material, MaterialStore, MediumModel, WaterBaseModel
    from theia.random import PhiloxRNG
    from theia.response import HistogramHitResponse, UniformValueResponse
    from theia.scene import MeshStore, Scene, Transform
    from theia.target import SphereTargetGuide
    from theia.trace import SceneForwardTracer

    meshStore = MeshStore({"sphere": S.bundled_mesh("sphere")})  # notebook: ../assets/*.stl

    class WaterModel(WaterBaseModel, HenyeyGreensteinPhaseFunction, MediumModel):
        def __init__(self):
            WaterBaseModel.__init__(self, temperature=4.0, pressure=10_000, salinity=35.0)
            HenyeyGreensteinPhaseFunction.__init__(self, 0.9)

    lamMin, lamMax = 400.0 * u.nm, 800.0 * u.nm
    water = WaterModel().createMedium(lamMin, lamMax, name="water")
    glass = BK7Model().createMedium(lamMin, lamMax, name="glass")
    glass_water = Material("glass_water", glass, water, flags=("TR", "T"))
    air_glass = Material("air_glass", None, "glass", flags="T")
    det_water = Material("det_water", None, water, flags="BD")
    matStore = MaterialStore([glass_water, air_glass, det_water])

    light_pos = (3.0, 0.0, 0.0) * u.m
    det_pos = (0.0, 3.0, 0.0) * u.m
    outer_sphere = meshStore.createInstance("sphere", "glass_water", Transform.TRS(scale=80.0 * u.cm, translate=light_pos))
    inner_sphere = meshStore.createInstance("sphere", "air_glass", Transform.TRS(scale=75.0 * u.cm, translate=light_pos))
    det_sphere = meshStore.createInstance("sphere", "det_water", Transform.TRS(scale=60.0 * u.cm, translate=det_pos), detectorId=1)
    scene = Scene([outer_sphere, inner_sphere, det_sphere], materials=matStore.material, medium=matStore.media["water"])

    batch_size, n_bins, bin_size = 16 * 1024, 100, 5.0 * u.ns
    wavelength = UniformWavelengthSource(lambdaRange=(lamMin, lamMax))
    light = SphericalLightSource(position=light_pos, timeRange=(0.0, 10.0) * u.ns, budget=1.0e5)
    guide = SphereTargetGuide(position=det_pos, radius=60.0 * u.cm)
    response = HistogramHitResponse(UniformValueResponse(), nBins=n_bins, binSize=bin_size)
    rng = PhiloxRNG(key=42)
    stats = EventStatisticCallback()
    tracer = SceneForwardTracer(batch_size, light, wavelength, response, rng, scene, maxPathLength=10, sourceMedium=0,
                                scatterCoefficient=0.05 / u.m, targetId=1, targetGuide=guide, callback=stats)
    rng.autoAdvance = tracer.nRNGSamples

    hists = []
    pipeline = hpl.Pipeline(tracer.collectStages())
    scheduler = hpl.PipelineScheduler(pipeline, processFn=lambda c, b, a: hists.append(response.result(c).copy()))
    scheduler.schedule([{}] * 4)
    scheduler.wait()
    scheduler.destroy()
    final_hist = np.mean(hists, 0)
    assert len(hists) == 4 and final_hist.sum() > 0
    # a few rays are totally reflected inside the glass shell at a surface that only transmits ("T"):
    # GLSL refract() returns 0 there, the ray turns NaN and is counted as an error, never raised
    assert stats.created == 4 * batch_size and stats.mismatch == 0 and stats.error < 1e-3 * stats.created
    # light leaves the air sphere through two glass surfaces, then ~3.2 m of water: nothing before ~12 ns
    first = np.nonzero(final_hist)[0][0] * bin_size
    assert 10.0 <= first <= 25.0


def test_example03_shared_stages_between_pipelines(backend):
    """examples/03_multiple_lightsources.ipynb: two tracers share target, wavelength source, response
    and RNG; a scheduler feeds two named pipelines. The shared response must return the result of the
    pipeline that just ran."""
    import hephaistos.pipeline as hpl
    import theia

    medium = theia.testing.WaterTestModel().createMedium()
    store = theia.material.MaterialStore([], media=[medium])
    rng = theia.random.PhiloxRNG(key=0xABBA)
    response = theia.response.HistogramHitResponse(theia.response.UniformValueResponse(), nBins=100, binSize=5.0 * u.ns)
    target = theia.target.SphereTarget(position=(-1.0, 0.0, 0.0) * u.m, radius=50.0 * u.cm)
    wavelength = theia.light.ConstWavelengthSource(500.0 * u.nm)
    sharedStages = [target, wavelength, response, rng]
    sphereLight = theia.light.SphericalLightSource()
    pencilLight = theia.light.PencilLightSource()
    batchSize = 16 * 1024
    tracerParams = {"nScattering": 10, "medium": store.media["water"]}
    sphereTracer = theia.trace.VolumeForwardTracer(batchSize, sphereLight, *sharedStages, **tracerParams)
    pencilTracer = theia.trace.VolumeForwardTracer(batchSize, pencilLight, *sharedStages, **tracerParams)
    rng.autoAdvance = max(sphereTracer.nRNGSamples, pencilTracer.nRNGSamples)
    spherePipeline = hpl.Pipeline(sphereTracer.collectStages())
    pencilPipeline = hpl.Pipeline(pencilTracer.collectStages())

    hists = []
    scheduler = hpl.PipelineScheduler({"sphere": spherePipeline, "pencil": pencilPipeline},
                                      processFn=lambda c, b, a: hists.append(response.result(c).copy()))
    tasks = [
        ("sphere", {"lightSource__budget": 1e5, "lightSource__timeRange": (0.0, 10.0) * u.ns,
                    "lightSource__position": (1.0, 0.0, 0.0) * u.m}),
        ("sphere", {}), ("sphere", {}),
        ("pencil", {"lightSource__position": (0.0, 1.0, 0.0) * u.m, "lightSource__direction": (-0.8, -0.36, -0.24),
                    "lightSource__budget": 5e6, "lightSource__timeRange": (8.0, 8.5) * u.ns}),
        ("pencil", {}), ("pencil", {}),
    ]
    scheduler.schedule(tasks)
    scheduler.wait()
    scheduler.destroy()
    sphereSignal = np.mean(hists[:3], 0)
    pencilSignal = np.mean(hists[3:], 0)
    assert sphereSignal.sum() > 0 and pencilSignal.sum() > 0
    # the pencil beam starts at 8 ns, 1.35 m from the target: nothing in the first bin (0..5 ns);
    # the spherical source (t in 0..10 ns, 1.5 m away) already contributes to bin 1 (5..10 ns)
    assert pencilSignal[0] == 0 and pencilSignal[1] == 0
    assert sphereSignal[1] > 0
    # different sources, different curves: the shared response returned each pipeline's own result
    assert pencilSignal.sum() > sphereSignal.sum() and not np.allclose(pencilSignal, sphereSignal, rtol=0.1)


def test_addresses_keep_their_memory_alive(backend):
    """theia hands device addresses around as integers; a tracer built inside a function from the addresses of a material store that
    dies with the function must not read freed memory (a GPU memory fault on HIP): the addresses hold on to the allocation"""
    import gc
    import weakref

    import theia.light
    import theia.material
    import theia.random
    import theia.response
    import theia.target
    import theia.trace
    from theia.testing import WaterTestModel

    def build():
        store = theia.material.MaterialStore([], media=[WaterTestModel().createMedium()])
        address = store.media["water"]
        assert isinstance(address, int) and address + 0 == int(address) and type(address + 0) is int
        rng = theia.random.PhiloxRNG(key=5)
        response = theia.response.HistogramHitResponse(theia.response.UniformValueResponse(), nBins=20, binSize=10.0 * u.ns)
        tracer = theia.trace.VolumeForwardTracer(4096, theia.light.SphericalLightSource(budget=1.0e3), theia.target.SphereTarget(position=(3.0, 0.0, 0.0), radius=1.0),
                                                 theia.light.ConstWavelengthSource(), response, rng, medium=address, nScattering=5)
        return tracer, response, weakref.ref(store._blob)

    tracer, response, blob = build()
    gc.collect()
    assert blob() is not None  # the store is gone, its memory is not
    pl.runPipeline(tracer.collectStages())
    assert response.result(0).sum() > 0
    del tracer
    gc.collect()
    assert blob() is None


def test_buffers_tensors_and_copy_commands(backend):
    """the hephaistos memory surface user code stages its own data with (theia_amd/hp/memory.py): typed buffers and tensors, update /
    retrieve / clear with offsets, execute and beginSequence, addresses that fit buffer_reference parameters"""
    import ctypes

    import hephaistos as hp

    src = hp.FloatBuffer(np.arange(12, dtype=np.float32).reshape(3, 4))
    assert src.size_bytes == 48 and src.numpy().shape == (3, 4)
    t = hp.FloatTensor(12)
    out = hp.FloatBuffer(12)
    hp.execute(hp.updateTensor(src, t), hp.retrieveTensor(t, out))
    assert np.array_equal(out.numpy(), np.arange(12))
    # a range in the middle, then cleared words
    part = hp.FloatBuffer([100.0, 101.0])
    seq = hp.beginSequence().And(hp.updateTensor(part, t, tensorOffset=16)).Then(hp.clearTensor(t, offset=40, size=8, data=0x3F800000)).Then(hp.retrieveTensor(t, out))
    assert np.array_equal(out.numpy(), np.arange(12))  # (nothing ran yet)
    seq.Submit().wait()
    assert np.array_equal(out.numpy(), [0, 1, 2, 3, 100, 101, 6, 7, 8, 9, 1, 1])
    with pytest.raises(ValueError):
        hp.updateTensor(part, t, tensorOffset=44, size=8)
    with pytest.raises(ValueError):
        hp.clearTensor(t, offset=2, size=4)
    # tensors made from data; addresses are integers that keep the allocation alive
    u = hp.UnsignedIntTensor(np.array([7, 8, 9], dtype=np.uint32))
    back = hp.UnsignedIntBuffer(3)
    hp.execute([hp.retrieveTensor(u, back)], hp.flushMemory())
    assert list(back.numpy()) == [7, 8, 9] and isinstance(u.address, int) and int(u) == u.address and u.address.owner is u._blob

    class Item(ctypes.Structure):
        _fields_ = [("a", ctypes.c_float), ("b", ctypes.c_uint32)]

    items = hp.ArrayBuffer(Item, 4)
    items.numpy()["b"] = [1, 2, 3, 4]
    dev = hp.ArrayTensor(Item, 4)
    again = hp.ArrayBuffer(Item, 4)
    hp.execute(hp.updateTensor(items, dev), hp.retrieveTensor(dev, again))
    assert list(again.numpy()["b"]) == [1, 2, 3, 4]
    one = hp.StructureBuffer(Item)
    one.a, one.b = 2.5, 9
    assert one.numpy().view(np.float32)[0] == 2.5 and one.b == 9
    with pytest.raises(NotImplementedError):
        hp.Tensor(16, mapped=True)
    with pytest.raises(NotImplementedError):
        hp.Program(b"")


def test_util_helpers():
    """theia.util: integers in and out of uvec fields, the structure-of-arrays view, the macro text; the shader loaders say what they are"""
    import ctypes

    import theia.util as tu

    big = 0x0123456789ABCDEF_FEDCBA9876543210
    assert tu.uvec4ToInt(tu.intToUvec4(big)) == big and tu.intToUvec4(big).w == 0x01234567
    assert tu.unpackUint64(tu.packUint64(0xDEADBEEF_00C0FFEE)) == 0xDEADBEEF_00C0FFEE and tu.packUint64(5).y == 0
    assert tu.createPreamble(N=4, POLARIZATION=True, OFF=False, NAME="x") == "#define N 4\n#define POLARIZATION\n#define NAME x\n"

    class Item(ctypes.Structure):
        _fields_ = [("value", ctypes.c_float), ("id", ctypes.c_uint32)]

    raw = np.concatenate([np.arange(5, dtype=np.float32).view(np.uint32), np.arange(10, 15, dtype=np.uint32)])
    view = tu.viewSoA(raw.ctypes.data, Item, 5)
    assert list(view["value"]) == [0, 1, 2, 3, 4] and list(view["id"]) == [10, 11, 12, 13, 14]
    with pytest.raises(NotImplementedError):
        tu.compileShader("tracer.scene.forward.glsl")


def test_what_is_not_there_says_so():
    """names of the reference this backend does not provide fail with a reason, not with an AttributeError deep inside user code"""
    import theia.camera
    import theia.trace

    with pytest.raises(NotImplementedError, match="backward"):
        theia.camera.PencilCamera
    with pytest.raises(AttributeError):
        theia.camera.nonsense
    with pytest.raises(NotImplementedError, match="forward tracing path"):
        theia.trace.BidirectionalPathTracer
