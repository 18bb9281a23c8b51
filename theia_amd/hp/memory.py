"""
`hephaistos` buffers, tensors and the commands that move data between them, as far as user code around theia's forward tracers
needs them ("hephaistos surface actually touched": `hp.Buffer` / `hp.Tensor` and their typed forms, `updateTensor`,
`retrieveTensor`, `clearTensor`, `execute`, `beginSequence`, `flushMemory`). A `Buffer` is host memory with a numpy view, a `Tensor`
is device memory (`thx_mem_alloc`) known by its address -- the integer theia's parameter blocks carry. Commands are zero-argument
callables, as everywhere in this pipeline layer (`theia_amd/hp/pipeline.py`); they run when executed, synchronously.

Not here: `hp.Program` / `hp.Compiler` (GLSL compiled at run time -- the components of this backend are compiled kernel variants),
images, and tensors mapped into host memory (`Tensor(..., mapped=True)` raises: write through a `Buffer` and `updateTensor`).
"""
from __future__ import annotations

import ctypes
from typing import Callable

import numpy as np

from .. import _backend
from ..lookup import DeviceAddress, DeviceBlob

__all__ = ["Buffer", "Tensor", "ByteBuffer", "ByteTensor", "FloatBuffer", "FloatTensor", "IntBuffer", "IntTensor", "UnsignedIntBuffer",
           "UnsignedIntTensor", "ArrayBuffer", "ArrayTensor", "StructureBuffer", "StructureTensor", "updateTensor", "retrieveTensor", "clearTensor", "execute",
           "beginSequence", "flushMemory", "Program", "Compiler"]

Command = Callable[[], None]


class Buffer:
    """`size` bytes of host memory"""

    def __init__(self, size: int) -> None:
        self._bytes = np.zeros(int(size), dtype=np.uint8)

    size = property(lambda self: self._bytes.nbytes, doc="size in bytes")
    size_bytes = size
    address = property(lambda self: self._bytes.ctypes.data, doc="address of the memory in this process")

    def numpy(self) -> np.ndarray:
        return self._bytes


class _Typed(Buffer):
    dtype = np.uint8

    def __init__(self, count_or_data) -> None:
        if isinstance(count_or_data, (int, np.integer)):
            Buffer.__init__(self, int(count_or_data) * np.dtype(self.dtype).itemsize)
        else:
            data = np.ascontiguousarray(count_or_data, dtype=self.dtype)
            Buffer.__init__(self, data.nbytes)
            self._bytes[:] = data.reshape(-1).view(np.uint8)
            self._shape = data.shape

    def numpy(self) -> np.ndarray:
        flat = self._bytes.view(self.dtype)
        return flat.reshape(getattr(self, "_shape", flat.shape))


class ByteBuffer(_Typed):
    dtype = np.uint8


class FloatBuffer(_Typed):
    dtype = np.float32


class IntBuffer(_Typed):
    dtype = np.int32


class UnsignedIntBuffer(_Typed):
    dtype = np.uint32


class ArrayBuffer(Buffer):
    """`count` items of a ctypes structure; `.numpy()` is a structured array over them"""

    def __init__(self, item: type[ctypes.Structure], count: int) -> None:
        Buffer.__init__(self, ctypes.sizeof(item) * int(count))
        self.item, self.count = item, int(count)

    def numpy(self) -> np.ndarray:
        return np.ctypeslib.as_array((self.item * self.count).from_buffer(self._bytes))


class StructureBuffer(Buffer):
    """one ctypes structure in host memory; its fields are attributes"""

    def __init__(self, item: type[ctypes.Structure]) -> None:
        Buffer.__init__(self, ctypes.sizeof(item))
        object.__setattr__(self, "_value", item.from_buffer(self._bytes))

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_value"), name)

    def __setattr__(self, name, value) -> None:
        if name.startswith("_"):
            object.__setattr__(self, name, value)
        else:
            setattr(self._value, name, value)


class Tensor:
    """`size` bytes of device memory, or a copy of the given bytes / array / `Buffer`"""

    def __init__(self, size_or_data, mapped: bool = False) -> None:
        if mapped:
            raise NotImplementedError("tensors mapped into host memory are not available on the HIP backend: fill a Buffer and updateTensor()")
        data = None
        if isinstance(size_or_data, (int, np.integer)):
            size = int(size_or_data)
        else:
            data = size_or_data.numpy() if isinstance(size_or_data, Buffer) else np.ascontiguousarray(size_or_data)
            data = data.reshape(-1).view(np.uint8)
            size = data.nbytes
        self._blob = DeviceBlob(max(size, 4))
        self._size = size
        if data is not None:
            self._blob.upload(data)

    size = property(lambda self: self._size, doc="size in bytes")
    size_bytes = size
    isMapped = property(lambda self: False)
    memory = property(lambda self: 0, doc="host address of a mapped tensor (never mapped here)")
    address = property(lambda self: DeviceAddress(self._blob.address, self._blob), doc="device address: what a buffer_reference parameter takes")

    def __int__(self) -> int:
        return self._blob.address


def _typed_tensor(name: str, dtype):
    def __init__(self, count_or_data, mapped: bool = False) -> None:
        if isinstance(count_or_data, (int, np.integer)):
            Tensor.__init__(self, int(count_or_data) * np.dtype(dtype).itemsize, mapped)
        else:
            Tensor.__init__(self, np.ascontiguousarray(count_or_data.numpy() if isinstance(count_or_data, Buffer) else count_or_data, dtype=dtype), mapped)

    return type(name, (Tensor,), {"__init__": __init__, "dtype": dtype, "__doc__": f"device memory for {np.dtype(dtype).name} items (a count, or the data)"})


ByteTensor = _typed_tensor("ByteTensor", np.uint8)
FloatTensor = _typed_tensor("FloatTensor", np.float32)
IntTensor = _typed_tensor("IntTensor", np.int32)
UnsignedIntTensor = _typed_tensor("UnsignedIntTensor", np.uint32)


class ArrayTensor(Tensor):
    def __init__(self, item: type[ctypes.Structure], count: int, mapped: bool = False) -> None:
        Tensor.__init__(self, ctypes.sizeof(item) * int(count), mapped)
        self.item, self.count = item, int(count)


class StructureTensor(Tensor):
    def __init__(self, item: type[ctypes.Structure], mapped: bool = False) -> None:
        Tensor.__init__(self, ctypes.sizeof(item), mapped)
        self.item = item


def _span(have_src: int, have_dst: int, src_offset: int, dst_offset: int, size) -> int:
    n = min(have_src - src_offset, have_dst - dst_offset) if size is None else int(size)
    if n < 0 or src_offset < 0 or dst_offset < 0 or src_offset + n > have_src or dst_offset + n > have_dst:
        raise ValueError("copy range lies outside the buffer or the tensor")
    return n


def updateTensor(src: Buffer, dst: Tensor, *, bufferOffset: int = 0, tensorOffset: int = 0, size: int | None = None, **_ignored) -> Command:
    """command: host buffer -> device tensor"""
    n = _span(src.size, dst.size, bufferOffset, tensorOffset, size)
    return lambda: _backend.get().upload(int(dst) + tensorOffset, src._bytes[bufferOffset:bufferOffset + n]) if n else None


def retrieveTensor(src: Tensor, dst: Buffer, *, bufferOffset: int = 0, tensorOffset: int = 0, size: int | None = None, **_ignored) -> Command:
    """command: device tensor -> host buffer"""
    n = _span(src.size, dst.size, tensorOffset, bufferOffset, size)

    def run() -> None:
        if n:
            dst._bytes[bufferOffset:bufferOffset + n] = np.frombuffer(_backend.get().download(int(src) + tensorOffset, n), dtype=np.uint8)

    return run


def clearTensor(tensor: Tensor, *, offset: int = 0, size: int | None = None, data: int = 0, **_ignored) -> Command:
    """command: fills (a range of) a tensor with a repeated 32-bit word"""
    n = _span(tensor.size, tensor.size, offset, offset, size)
    if n % 4 or offset % 4:
        raise ValueError("clearTensor works on whole 32-bit words")
    return lambda: _backend.get().upload(int(tensor) + offset, np.full(n // 4, data & 0xFFFFFFFF, dtype=np.uint32)) if n else None


def execute(*commands) -> None:
    """runs commands (or lists of commands) one after the other and returns when they are done"""
    for c in commands:
        if c is None:
            continue
        if callable(c):
            c()
        else:
            execute(*c)


class _Submission:
    def wait(self, *_a) -> None:
        pass

    forgettable = property(lambda self: False)


class _Sequence:
    """`beginSequence().And(cmd).Then(cmd).Submit().wait()`: the commands run in the order they were added, at `Submit()`"""

    def __init__(self) -> None:
        self._commands: list = []

    def And(self, command):
        self._commands.append(command)
        return self

    Then = And

    def AndList(self, commands):
        self._commands.extend(commands)
        return self

    ThenList = AndList

    def WaitFor(self, *_a):
        return self

    def Submit(self) -> _Submission:
        execute(*self._commands)
        self._commands = []
        return _Submission()


def beginSequence(*_a) -> _Sequence:
    return _Sequence()


def flushMemory() -> Command:
    """command ordering shader writes before transfers: nothing to do, every command here completes before it returns"""
    return lambda: None


class Program:
    def __init__(self, *_a, **_k) -> None:
        raise NotImplementedError("hp.Program compiles GLSL at run time; this backend's components are compiled kernel variants of libtheia_hip.so "
                                  "(DESIGN.md section 1: closed component set)")


Compiler = Program
