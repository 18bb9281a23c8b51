"""scenes of tools/sdwa_check.py / tools/determinism.py: production-variant launches whose curves must not depend on the launch"""
import test_reference_energy as E
from theia_amd import workloads as S


def shells(batch, polarized=False):
    """the reference's two-shell energy scene (tests/test_trace_energy.py:479-568 of the reference): paths of up to 50 steps through two
    nested refracting spheres -- the scene on which the round-2 nondeterminism showed"""
    inner = E.medium_model(0.0, 0.01, 0.9, n=1.8).createMedium(name="inner")
    outer = E.medium_model(0.0, 0.04, 0.9, n=1.2).createMedium(name="outer")
    store = E.TM.MaterialStore([E.TM.Material("det", outer, None, flags="DB"), E.TM.Material("inner", inner, outer, flags="TR")])
    meshes = E.TS.MeshStore({"sphere": E.S.bundled_mesh("sphere")})
    inst = [meshes.createInstance("sphere", name, E.TS.Transform.TRS(scale=r, translate=E.OFF_CENTRE)) for name, r in (("inner", 65.0 * E.u.m), ("det", 100.0 * E.u.m))]
    scene = E.TS.Scene(inst, store.material)
    rng = E.PhiloxRNG(key=0xC0FFEE)
    response = E.TR.HistogramHitResponse(E.TR.UniformValueResponse(), nBins=400, binSize=100.0 * E.u.ns)
    tracer = E.TT.SceneForwardTracer(batch, E.point_source((2.0, 5.0, 0.0) * E.u.m, 1e9), E.TL.ConstWavelengthSource(E.LAMBDA), response, rng, scene,
                                     maxPathLength=50, scatterCoefficient=0.03, sourceMedium=store.media["inner"], maxTime=float("inf"),
                                     polarized=polarized)
    return S.Scenario(name="shells", tracer=tracer, response=response, rng=rng, scene=scene, callback=tracer.callback, store=store, meshes=meshes)


PRODUCTION_CASES = [
    ("shells", shells, dict(batch=2_000_000)),
    ("shells pol", shells, dict(batch=1_000_000, polarized=True)),
    ("config2", S.config2_sphere, dict(batch=2_000_000)),
    ("config3", S.config3_cube_torus, dict(batch=1_000_000, polarized=True)),
    ("config5", S.config5_suzanne_grid, dict(batch=500_000, polarized=True)),
]
