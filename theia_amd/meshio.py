"""
Mesh file readers (binary/ASCII STL, binary/ASCII PLY) producing the arrays the
scene builder consumes: vertices ``(N, 6)`` float32 = position + vertex normal,
faces ``(M, 3)`` uint32.

Replaces the reference's `trimesh.load_mesh` use (src/theia/scene.py:434-449,
`_createMeshFromTrimesh`: vertices = concat(position, vertex_normals)). The
vertex normal is only used to orient the geometric normal
(shader/scene.intersect.glsl:50-58), so an area-weighted average is sufficient.
"""
from __future__ import annotations

import os
import struct

import numpy as np

__all__ = ["load_mesh_arrays", "compute_vertex_normals", "save_stl", "icosphere"]


def compute_vertex_normals(positions: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Area weighted vertex normals (sum of un-normalized face normals)."""
    p = positions.astype(np.float64)
    e1 = p[faces[:, 1]] - p[faces[:, 0]]
    e2 = p[faces[:, 2]] - p[faces[:, 0]]
    fn = np.cross(e1, e2)
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    length = np.sqrt(np.square(vn).sum(-1, keepdims=True))
    length[length == 0.0] = 1.0
    return (vn / length).astype(np.float32)


def _merge_vertices(tri: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """tri: (M,3,3) float32 corner positions -> unique vertices + faces"""
    flat = np.ascontiguousarray(tri.reshape(-1, 3))
    # bit-exact comparison; -0.0 and +0.0 are folded
    flat = flat + np.float32(0.0)
    uniq, inverse = np.unique(flat, axis=0, return_inverse=True)
    faces = inverse.reshape(-1, 3).astype(np.uint32)
    # drop degenerate faces (two identical corners)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return uniq.astype(np.float32), faces[ok]


def _load_stl(data: bytes) -> tuple[np.ndarray, np.ndarray]:
    n = struct.unpack_from("<I", data, 80)[0] if len(data) >= 84 else 0
    if len(data) == 84 + 50 * n and n > 0:
        rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
        tri = rec["v"].copy()
    else:  # ASCII
        verts = []
        for line in data.decode("ascii", errors="ignore").splitlines():
            parts = line.split()
            if len(parts) == 4 and parts[0] == "vertex":
                verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
        tri = np.asarray(verts, dtype=np.float32).reshape(-1, 3, 3)
    pos, faces = _merge_vertices(tri)
    nrm = compute_vertex_normals(pos, faces)
    return np.concatenate([pos, nrm], axis=-1), faces


_PLY_TYPES = {
    "char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
    "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
    "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8",
}


def _load_ply(data: bytes) -> tuple[np.ndarray, np.ndarray]:
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii").splitlines()
    fmt = "ascii"
    elements = []  # (name, count, [(kind, ...)])
    for line in header:
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "format":
            fmt = parts[1]
        elif parts[0] == "element":
            elements.append((parts[1], int(parts[2]), []))
        elif parts[0] == "property":
            if parts[1] == "list":
                elements[-1][2].append(("list", parts[2], parts[3], parts[4]))
            else:
                elements[-1][2].append(("scalar", parts[1], parts[2]))
    vertices = None
    faces = []
    if fmt == "ascii":
        tokens = data[end:].decode("ascii").split()
        pos = 0
        for name, count, props in elements:
            rows = []
            for _ in range(count):
                row = {}
                for pr in props:
                    if pr[0] == "scalar":
                        row[pr[2]] = float(tokens[pos]); pos += 1
                    else:
                        k = int(tokens[pos]); pos += 1
                        row[pr[3]] = [int(t) for t in tokens[pos:pos + k]]; pos += k
                rows.append(row)
            if name == "vertex":
                vertices = rows
            elif name == "face":
                faces = [r.get("vertex_indices", r.get("vertex_index")) for r in rows]
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        off = end
        for name, count, props in elements:
            if all(p[0] == "scalar" for p in props):
                dt = np.dtype([(p[2], bo + _PLY_TYPES[p[1]]) for p in props])
                arr = np.frombuffer(data, dtype=dt, count=count, offset=off)
                off += dt.itemsize * count
                if name == "vertex":
                    vertices = arr
            else:
                rows = []
                for _ in range(count):
                    row = {}
                    for pr in props:
                        if pr[0] == "scalar":
                            dt = np.dtype(bo + _PLY_TYPES[pr[1]])
                            row[pr[2]] = np.frombuffer(data, dt, 1, off)[0]; off += dt.itemsize
                        else:
                            cdt = np.dtype(bo + _PLY_TYPES[pr[1]])
                            k = int(np.frombuffer(data, cdt, 1, off)[0]); off += cdt.itemsize
                            idt = np.dtype(bo + _PLY_TYPES[pr[2]])
                            row[pr[3]] = np.frombuffer(data, idt, k, off).tolist(); off += idt.itemsize * k
                    rows.append(row)
                if name == "face":
                    faces = [r.get("vertex_indices", r.get("vertex_index")) for r in rows]
    if vertices is None:
        raise ValueError("PLY file has no vertex element")
    if isinstance(vertices, list):
        get = lambda k: np.array([v.get(k, np.nan) for v in vertices], dtype=np.float32)
        names = set(vertices[0].keys())
    else:
        get = lambda k: vertices[k].astype(np.float32) if k in vertices.dtype.names else None
        names = set(vertices.dtype.names)
    pos = np.stack([get("x"), get("y"), get("z")], -1)
    tris = []
    for f in faces:  # fan-triangulate polygons
        for k in range(1, len(f) - 1):
            tris.append([f[0], f[k], f[k + 1]])
    tri = np.asarray(tris, dtype=np.uint32).reshape(-1, 3)
    if {"nx", "ny", "nz"} <= names:
        nrm = np.stack([get("nx"), get("ny"), get("nz")], -1)
    else:
        nrm = compute_vertex_normals(pos, tri)
    return np.concatenate([pos, nrm], axis=-1).astype(np.float32), tri


def load_mesh_arrays(path: str | os.PathLike) -> tuple[np.ndarray, np.ndarray]:
    """Returns (vertices (N,6) float32, faces (M,3) uint32)."""
    with open(path, "rb") as fh:
        data = fh.read()
    ext = os.path.splitext(str(path))[1].lower()
    if ext == ".stl":
        v, f = _load_stl(data)
    elif ext == ".ply":
        v, f = _load_ply(data)
    else:
        raise ValueError(f"Unsupported mesh format '{ext}' (supported: .stl, .ply)")
    return np.ascontiguousarray(v, dtype=np.float32), np.ascontiguousarray(f, dtype=np.uint32)


def save_stl(path: str | os.PathLike, positions: np.ndarray, faces: np.ndarray) -> None:
    """Writes a binary STL (used by tests and to materialise bundled meshes)."""
    p = positions[:, :3].astype(np.float32)
    tri = p[faces]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ln = np.sqrt(np.square(n).sum(-1, keepdims=True))
    ln[ln == 0] = 1.0
    rec = np.zeros(len(faces), dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["n"] = n / ln
    rec["v"] = tri
    with open(path, "wb") as fh:
        fh.write(b"theia_amd binary stl".ljust(80, b"\0"))
        fh.write(struct.pack("<I", len(faces)))
        fh.write(rec.tobytes())


def icosphere(subdivisions: int = 3) -> tuple[np.ndarray, np.ndarray]:
    """Unit icosphere (subdivisions=3 -> 642 vertices / 1280 faces)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
         (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
         (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    faces = list(f)
    for _ in range(subdivisions):
        cache: dict[tuple[int, int], int] = {}

        def mid(a: int, b: int) -> int:
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    pos = np.asarray(verts, dtype=np.float32)
    fa = np.asarray(faces, dtype=np.uint32)
    nrm = compute_vertex_normals(pos, fa)
    return np.concatenate([pos, nrm], -1).astype(np.float32), fa
