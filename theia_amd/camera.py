"""
`theia.camera` belongs to the backward / bidirectional tracers (a camera samples the rays those tracers start from), which this backend
does not provide (the forward tracing path). The module exists so that `import theia.camera` works
and every name in it says why it is not there.
"""
from __future__ import annotations

__all__: list[str] = []

_NAMES = ("Camera", "CameraRayItem", "PolarizedCameraRayItem", "CameraRaySampler", "HostCamera", "PencilCamera", "FlatCamera", "ConeCamera",
          "SphereCamera", "PointCamera", "MeshCamera")


def __getattr__(name: str):
    if name in _NAMES:
        raise NotImplementedError(f"theia.camera.{name}: cameras feed the backward and bidirectional tracers, which the HIP backend does not "
                                  "provide (forward tracers only: VolumeForwardTracer, SceneForwardTracer and the photon tracers)")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
