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

// scenes with at least this many pair records are built with 4-wide records: they need 32-bit stack entries anyway (api.cpp: stack16),
// their hierarchies are far larger than any cache, and a 4-wide record is one 64-B line where a pair record uses half of one
constexpr size_t kWideFromPairs = 4096;

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
