#!/usr/bin/env python3
"""What a Cherenkov-family light source costs a tracer: BASELINE config 2 at 10^7 photons with its own point source and with each of
the four sources the sampler kernel draws first (api.cpp: thx_tracer_run) -- ms per batch by HIP events around sampler + tracing
kernel, median of 5 after 3 warm-up batches.   tools/source_cost.py [photons]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import theia_amd.light as TL
import theia_amd.units as u
from theia_amd import workloads as S

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
track = TL.ParticleTrack(4)
track.setVertices([[-6.0, -20.0, 2.5, 0.0], [-6.0, 0.0, 2.5, 66.7], [-6.5, 20.0, 2.5, 133.4]])
lights = {
    "spherical (config 2)": None,
    "cherenkov": TL.CherenkovLightSource(trackStart=(-6.0, -20.0, 2.5), trackEnd=(-6.0, 20.0, 2.5), startTime=0.0, endTime=133.4, usePhotonCount=True),
    "cherenkov track": TL.CherenkovTrackLightSource(track, usePhotonCount=True),
    "muon track": TL.MuonTrackLightSource((-6.0, -20.0, 2.5), 0.0, (-6.0, 20.0, 2.5), 133.4, 2.0 * u.TeV),
    "cascade": TL.ParticleCascadeLightSource(startPosition=(-6.0, -3.0, 2.5), direction=(0.0, 1.0, 0.0), effectiveLength=5000.0, a_angular=0.54,
                                             b_angular=3.3, a_long=6.4, b_long=0.57),
}
for name, light in lights.items():
    sc = S.config2_sphere(batch=B, light=light, stats=False)
    ms = []
    for k in range(8):
        S.run_batches(sc, 1)
        ms.append(sc.tracer.program.last_kernel_ms(0))
    print(f"{name:22s} B={B}: {np.median(ms[3:]):7.3f} ms per batch (sampler + tracing kernel)", flush=True)
