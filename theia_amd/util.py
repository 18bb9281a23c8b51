"""
The helpers of `theia.util` that make sense without a GLSL compiler: integers in and out of `uvec2` / `uvec4` parameter fields, a
structure-of-arrays view onto host memory, and `createPreamble` (kept for code that builds the macro text of a component). The shader
loaders (`loadShader`, `compileShader`, `ShaderLoader`, `getCompiler`) have nothing to load here -- components are compiled kernel
variants of libtheia_hip.so -- and say so.
"""
from __future__ import annotations

from ctypes import Structure
from typing import Any

import numpy as np

from .hp.glsl import uvec2, uvec4

__all__ = ["viewSoA", "createPreamble", "uvec4ToInt", "intToUvec4", "packUint64", "unpackUint64", "getCompiler", "getShaderPath", "loadShader",
           "compileShader", "ShaderLoader"]

_WORD = 0xFFFFFFFF


def viewSoA(address: int, item: type[Structure], count: int) -> np.ndarray:
    """structured numpy view of `count` items stored column by column at a HOST address (one array of `count` entries per field)"""

    class Columns(Structure):
        _fields_ = [(name, ctype * count) for name, ctype in item._fields_]

    return np.ctypeslib.as_array(Columns.from_address(address))


def createPreamble(**macros: Any) -> str:
    """`#define NAME value` per entry; a bool defines the bare name when true and nothing when false"""
    lines = []
    for name, value in macros.items():
        if isinstance(value, bool):
            if value:
                lines.append(f"#define {name}\n")
        else:
            lines.append(f"#define {name} {value}\n")
    return "".join(lines)


def uvec4ToInt(value: uvec4) -> int:
    return sum(int(word) << (32 * k) for k, word in enumerate((value.x, value.y, value.z, value.w)))


def intToUvec4(value: int) -> uvec4:
    return uvec4(*((int(value) >> (32 * k)) & _WORD for k in range(4)))


def packUint64(value: int) -> uvec2:
    return uvec2(int(value) & _WORD, (int(value) >> 32) & _WORD)


def unpackUint64(value: uvec2) -> int:
    return int(value.x) | (int(value.y) << 32)


def _no_shaders(*_a, **_k):
    raise NotImplementedError("there is no GLSL to load or compile on the HIP backend: the components are compiled kernel variants of "
                              "libtheia_hip.so (DESIGN.md section 1)")


getCompiler = getShaderPath = loadShader = compileShader = _no_shaders


class ShaderLoader:
    def __init__(self, *_a, **_k) -> None:
        _no_shaders()
