"""
theia_amd -- MI355X-native transient photon tracing behind theia's module API.

The forward tracers (`theia_amd.trace.SceneForwardTracer`,
`VolumeForwardTracer`) run as hand-written HIP kernels in ``libtheia_hip.so``
(sources in ``theia_amd/csrc``, C ABI in ``include/theia_hip.h``). The modules
here mirror the reference's `theia.*` modules for the components on that path;
``import theia`` / ``import hephaistos`` resolve to them through the alias
packages at the repository root.
"""
import os as _os

# Batches in flight side by side (PipelineScheduler -> thx_tracer_set_concurrency) need hardware queues: the HIP runtime reads
# GPU_MAX_HW_QUEUES (default 4: two kernels at once on MI355X) when it initialises, so the variable has to be in place before the
# first HIP call of the process -- libtheia_hip.so sets it when it is loaded, this line covers processes in which another
# library (torch) touches HIP first. A value the user has set is kept.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import units  # noqa: F401

__version__ = "0.1.0"

_SUBMODULES = ("units", "lookup", "material", "random", "light", "target", "response", "scene", "trace", "testing", "meshio", "distributed")


def __getattr__(name: str):
    if name in _SUBMODULES:
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
