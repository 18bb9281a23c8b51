"""
ctypes mirrors of the GLSL vector/matrix types used in parameter blocks
(`hephaistos.glsl` in the reference stack: vec2/vec3/vec4, uvec2/uvec4, mat3,
mat4, mat4x3, buffer_reference). Scalar block layout: tightly packed floats.
"""
from __future__ import annotations

from ctypes import Structure, c_float, c_int32, c_uint32

__all__ = ["vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4",
           "mat2", "mat3", "mat4", "mat4x3", "mat3x4", "buffer_reference"]


def _vector(name: str, ctype, comps: str):
    class _V(Structure):
        _fields_ = [(c, ctype) for c in comps]

        def __init__(self, *args, **kwargs):
            if len(args) == 1 and hasattr(args[0], "__iter__"):
                args = tuple(args[0])
            super().__init__(*args, **kwargs)

        def __iter__(self):
            return (getattr(self, c) for c in comps)

        def __len__(self):
            return len(comps)

        def __getitem__(self, i):
            return tuple(self)[i]

        def __setitem__(self, i, value):
            setattr(self, comps[i], value)

        def __repr__(self):
            return f"{name}({', '.join(repr(v) for v in self)})"

    _V.__name__ = _V.__qualname__ = name
    return _V


vec2 = _vector("vec2", c_float, "xy")
vec3 = _vector("vec3", c_float, "xyz")
vec4 = _vector("vec4", c_float, "xyzw")
ivec2 = _vector("ivec2", c_int32, "xy")
ivec3 = _vector("ivec3", c_int32, "xyz")
ivec4 = _vector("ivec4", c_int32, "xyzw")
uvec2 = _vector("uvec2", c_uint32, "xy")
uvec3 = _vector("uvec3", c_uint32, "xyz")
uvec4 = _vector("uvec4", c_uint32, "xyzw")


def _matrix(name: str, cols: int, rows: int):
    class _M(Structure):
        # column major: `cols` columns of `rows` floats
        _fields_ = [("data", c_float * (cols * rows))]
        shape = (cols, rows)

        def __repr__(self):
            return f"{name}({list(self.data)})"

    _M.__name__ = _M.__qualname__ = name
    return _M


mat2 = _matrix("mat2", 2, 2)
mat3 = _matrix("mat3", 3, 3)
mat4 = _matrix("mat4", 4, 4)
mat4x3 = _matrix("mat4x3", 4, 3)
mat3x4 = _matrix("mat3x4", 3, 4)


class buffer_reference(Structure):
    """64-bit device address stored as two 32-bit words (uvec2), so that it packs
    at 4-byte alignment in scalar-layout blocks."""

    _fields_ = [("lo", c_uint32), ("hi", c_uint32)]

    @property
    def value(self) -> int:
        return self.lo | (self.hi << 32)

    @value.setter
    def value(self, address: int) -> None:
        self.lo = address & 0xFFFFFFFF
        self.hi = (address >> 32) & 0xFFFFFFFF


def _layout(**columns) -> list:
    """ctypes `_fields_` from keyword arguments (their order is the memory order)"""
    return list(columns.items())
