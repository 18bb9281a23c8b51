"""
theia_amd -- MI355X-native transient photon tracing behind theia's module API.

The forward tracers (`theia_amd.trace.SceneForwardTracer`,
`VolumeForwardTracer`) run as hand-written HIP kernels in ``libtheia_hip.so``
(sources in ``theia_amd/csrc``, C ABI in ``include/theia_hip.h``). The modules
here mirror the reference's `theia.*` modules for the components on that path;
``import theia`` / ``import hephaistos`` resolve to them through the alias
packages at the repository root.
"""
from . import units  # noqa: F401

__version__ = "0.1.0"

_SUBMODULES = ("units", "lookup", "material", "random", "light", "target", "response", "scene", "trace", "testing", "meshio", "distributed")


def __getattr__(name: str):
    if name in _SUBMODULES:
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
