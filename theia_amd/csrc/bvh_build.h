// Host-side acceleration-structure builder (replaces what the reference leaves
// to the Vulkan driver: hp.GeometryStore / hp.AccelerationStructure,
// src/theia/scene.py:550, 668). Produces one node array holding the TLAS and
// every mesh's BLAS, the leaf-ordered triangle records and the instance records.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/theia_hip.h"
#include "thx_layout.h"

namespace thx {

// scenes with at least this many pair records (1 MB of them) are built with 4-wide records: their hierarchies are larger than the caches
// near the CUs, and a 4-wide record is one 64-B line where a pair record uses half of one. Measured on config 2 with the detector
// sphere subdivided (profiles/r04_wide_records.txt): 20 480 triangles +2.4 %, 81 920 +-0, 327 680 -3 %, 1.3 M -4.5 %, 5.2 M -6 %.
constexpr size_t kWideFromPairs = 32768;

struct BuiltScene {
    std::vector<BvhNode> nodes;         // [TLAS | BLAS 0 | BLAS 1 | ...] (host only: the source of `pairs`)
    std::vector<NodePair> pairs;        // what the device walks: one record per inner node
    std::vector<TreeFrame> frames;      // quantisation frame + root ref per tree: [TLAS, BLAS 0, BLAS 1, ...]
    std::vector<TriRecord> tris;        // leaf order, all meshes
    std::vector<NormalRecord> normals;  // vertex normals of the same triangles
    std::vector<InstanceRecord> instances;
    std::vector<uint32_t> tlasInstance; // TLAS leaf slot -> instance id (host only: the leaf refs of `pairs` hold the ids)
    std::vector<WalkInstance> walkInstances; // by instance id
    bool wide = false;                  // `pairs` holds 4-wide records (two slots each) instead of pair records (emitTree)
    uint32_t tlasRoot = 0;
    uint32_t maxDepth = 0;              // TLAS depth + 1 + deepest BLAS (stack bound)
    std::vector<uint32_t> blasRoot;     // per mesh
};

// throws std::runtime_error on invalid input
BuiltScene buildScene(const thx_mesh_desc* meshes, uint32_t n_meshes,
                      const thx_instance_desc* instances, uint32_t n_instances,
                      uint32_t max_stack_depth);

// world->object 3x4 from object->world 3x4 (double precision cofactors)
void invertAffine(const float m[12], float out[12]);

} // namespace thx
