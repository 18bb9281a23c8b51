// Device data layouts shared by the host-side builders and the HIP kernels.
// Sizes are the algorithmic bytes/unit of SURVEY.md 8(d): BVH2 node 32 B,
// triangle record 48 B, instance record 112 B. Reference-defined layouts
// (Medium 88 B, Material 24 B, Geometry 24 B, Vertex 24 B, Index 12 B) are kept
// verbatim so that device addresses handed to users stay meaningful.
#pragma once
#include <stdint.h>

namespace thx {

// BVH2 node, 32 B, fetched as two 16-B loads.
//   inner: count == 0, leftFirst = index of left child (right = left + 1)
//   leaf : count  > 0, leftFirst = first triangle record (BLAS) / first
//          instance slot (TLAS)
struct alignas(16) BvhNode {
    float lo[3];
    uint32_t leftFirst;
    float hi[3];
    uint32_t count;
};
static_assert(sizeof(BvhNode) == 32, "node must be 32 B");

// What the walk reads instead of BvhNode: the two children of an inner node in ONE 32-B record (two
// 16-B loads instead of four -- the walk is bound by the L1's gather rate, one cycle per lane and
// load instruction). The child boxes are quantised to 16 bit per coordinate in the frame of their
// tree (TLAS, or the mesh's BLAS): coordinate = origin + q * step, rounded outwards.
//   w[0] = L.lo.x | L.hi.x << 16   w[1] = L.lo.y | L.hi.y << 16   w[2] = L.lo.z | L.hi.z << 16   w[3] = ref L
//   w[4] = R.lo.x | R.hi.x << 16   w[5] = R.lo.y | R.hi.y << 16   w[6] = R.lo.z | R.hi.z << 16   w[7] = ref R
//   (one word per axis: the walk picks the near / far plane of an axis out of its word with one v_perm_b32, dev_trace.hpp: qPlane)
//   ref = (index << 3) | count: count == 0 -> index of the child's own pair record, count > 0 -> first
//   triangle record (BLAS) / first instance slot (TLAS) of the leaf
struct alignas(16) NodePair {
    uint32_t w[8];
};
static_assert(sizeof(NodePair) == 32, "node pair must be 32 B");
// frame of one tree, 32 B: tree 0 = TLAS, tree 1 + m = BLAS of mesh m
struct alignas(16) TreeFrame {
    float origin[3];
    uint32_t rootRef; // ref of the tree's root (a leaf if the tree is a single leaf)
    float step[3];
    uint32_t pad;
};
static_assert(sizeof(TreeFrame) == 32, "tree frame must be 32 B");

// Triangle record in BVH leaf order, 64 B (half a cache line, never across two). Every vertex is stored as x y z x y:
// the ray/triangle test wants the coordinates in the order (kx, ky, kz) with kz the ray's dominant axis -- a rotation of
// (x, y, z) -- and reads it as three consecutive floats from byte 0, 4 or 8 of the vertex (one 12-B load) instead of
// picking nine values with two selects each (dev_walk.hpp: woopSetup / woopTri).
struct alignas(16) TriRecord {
    float v[3][5]; // vertex k: x y z x y
    uint32_t prim; // index into the mesh's Index buffer
};
static_assert(sizeof(TriRecord) == 64, "triangle record must be 64 B");

// Vertex normals of a triangle in BVH leaf order, 48 B, next to its TriRecord: with them a hit is reconstructed from
// two records addressed by the hit itself, not through geometry -> index triple -> three vertices (three dependent loads).
struct alignas(16) NormalRecord {
    float n0[3]; uint32_t pad0;
    float n1[3]; uint32_t pad1;
    float n2[3]; uint32_t pad2;
};
static_assert(sizeof(NormalRecord) == 48, "normal record must be 48 B");

// Instance record, 112 B = seven 16-B loads. Matrices are row-major 3x4.
struct alignas(16) InstanceRecord {
    float w2o[12];
    float o2w[12];
    uint32_t blasRoot; // node index of the mesh's BLAS root
    int32_t customIndex;
    uint32_t geometry; // index into Geometry[]
    uint32_t frame;    // index of the mesh's TreeFrame
};
static_assert(sizeof(InstanceRecord) == 112, "instance record must be 112 B");

// What the walk reads when it enters an instance, 80 B = five 16-B loads from one address: the world -> object
// matrix and the quantisation frame of the instance's mesh. (A copy of what InstanceRecord and TreeFrame hold: read
// through them, entering an instance is three dependent loads -- leaf slot -> instance id -> record -> frame.)
struct alignas(16) WalkInstance {
    float w2o[12];
    float origin[3];
    uint32_t rootRef;
    float step[3];
    uint32_t pad;
};
static_assert(sizeof(WalkInstance) == 80, "walk instance record must be 80 B");

#pragma pack(push, 4)
// reference: src/theia/material.py:122-140, shader/material.glsl:11-34
struct MediumGLSL {
    float lambda_min, lambda_max;
    uint64_t n, vg, mu_a, mu_s, log_phase, phase_sampling;
    uint64_t phase_m12, phase_m22, phase_m33, phase_m34;
};
static_assert(sizeof(MediumGLSL) == 88, "Medium.GLSL is 88 B");
// reference: src/theia/material.py:587-597, shader/material.glsl:91-97
struct MaterialGLSL {
    uint64_t inside, outside;
    uint32_t flagsInwards, flagsOutwards;
};
static_assert(sizeof(MaterialGLSL) == 24, "Material.GLSL is 24 B");
// reference: src/theia/scene.py:628-635, shader/scene.types.glsl:16-20
struct GeometryGLSL {
    uint64_t vertices, indices, material;
};
static_assert(sizeof(GeometryGLSL) == 24, "Geometry is 24 B");
#pragma pack(pop)

// everything a traversal needs
struct SceneView {
    const NodePair* nodes; // pair records of all trees
    const TriRecord* tris;
    const NormalRecord* normals; // by triangle record
    const InstanceRecord* instances;
    const WalkInstance* walkInstances; // by instance id (the TLAS leaves hold instance ids)
    const GeometryGLSL* geometries;
    const TreeFrame* frames;
    TreeFrame tlas; // frame 0 by value: every query starts with it
    uint32_t nInstances;
    uint32_t wide; // `nodes` holds 4-wide records of 64 B (bvh_build.cpp: emitTree) instead of pair records
};

} // namespace thx
