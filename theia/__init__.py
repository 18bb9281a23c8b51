"""
`theia` alias package: lets scripts and notebooks written against
tkerscher/theia (`import theia`, `import theia.units as u`,
`from theia.trace import SceneForwardTracer`, ...) run on the MI355X backend
`theia_amd` unchanged. Only the forward-tracing path is provided; see
DESIGN.md for the supported component set.
"""
import importlib as _importlib
import sys as _sys

import hephaistos as _hp  # noqa: F401  (mirrors the reference: device comes up with the package)

_NAMES = ("units", "lookup", "material", "random", "light", "target", "response", "scene", "trace", "testing", "task", "cascades", "util", "camera")
for _n in _NAMES:
    _m = _importlib.import_module(f"theia_amd.{_n}")
    _sys.modules[f"{__name__}.{_n}"] = _m
    globals()[_n] = _m
del _n, _m
