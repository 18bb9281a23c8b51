"""
`hephaistos` alias package: the part of the hephaistos API that theia's forward
tracers, examples and tests use, served by theia_amd's HIP backend
(`theia_amd.hp`). No Vulkan involved.
"""
import sys as _sys

from theia_amd import hp as _hp
from theia_amd.hp import *  # noqa: F401,F403
from theia_amd.hp import glsl, pipeline, queue  # noqa: F401

_sys.modules[__name__ + ".pipeline"] = pipeline
_sys.modules[__name__ + ".glsl"] = glsl
_sys.modules[__name__ + ".queue"] = queue
