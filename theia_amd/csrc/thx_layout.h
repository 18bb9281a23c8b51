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

// Triangle record in BVH leaf order, 48 B = three 16-B loads.
struct alignas(16) TriRecord {
    float v0[3];
    uint32_t prim; // index into the mesh's Index buffer
    float v1[3];
    uint32_t pad0;
    float v2[3];
    uint32_t pad1;
};
static_assert(sizeof(TriRecord) == 48, "triangle record must be 48 B");

// Instance record, 112 B = seven 16-B loads. Matrices are row-major 3x4.
struct alignas(16) InstanceRecord {
    float w2o[12];
    float o2w[12];
    uint32_t blasRoot; // node index of the mesh's BLAS root
    int32_t customIndex;
    uint32_t geometry; // index into Geometry[]
    uint32_t pad;
};
static_assert(sizeof(InstanceRecord) == 112, "instance record must be 112 B");

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
    const BvhNode* nodes;
    const TriRecord* tris;
    const InstanceRecord* instances;
    const uint32_t* tlasInstance; // TLAS leaf slot -> instance id
    const GeometryGLSL* geometries;
    uint32_t tlasRoot;
    uint32_t nInstances;
};

} // namespace thx
