// Host-visible launch entry points of the HIP kernels (internal to the library).
#pragma once
#include <hip/hip_runtime.h>

#include "dev_trace.hpp"

namespace thx {

constexpr uint32_t kTraceBlock = 256;  // threads per tracing workgroup (4 wavefronts)
constexpr uint32_t kTreeDepth = 48;    // depth bound of the binary trees the builder makes (TLAS + deepest BLAS + 2)
constexpr uint32_t kStackDepth = kTreeDepth;     // traversal stack entries per lane
constexpr uint32_t kRecordSlots = 1;
constexpr uint32_t kStackLds16 = 16;   // staged kernel: entries of them kept in LDS (16-bit / 32-bit entries), the rest in global memory;
// staged kernel: a walk batch sets its last rays aside once fewer lanes than kParkBelow still walk (kParkBelowDeep for
// hierarchies deeper than kParkDeepFrom, whose walks are long and spread more) after at least kParkRounds rounds
// (dev_staged.hpp: walkBatch; measured: profiles/r02_park.txt)
// scenes of many instances spread their walks most (every instance visit is a detour of its own): kParkBelowInstanced
constexpr uint32_t kParkBelow = 10, kParkBelowDeep = 16, kParkDeepFrom = 16, kParkRounds = 2;
constexpr uint32_t kParkBelowInstanced = 24, kParkInstancedFrom = 16;
constexpr uint32_t kStackLds32 = 12;   // measured on MI355X (profiles/r02_stack_spill.txt): 81 920 triangles 7.27 -> 6.06 ms, 5.2 M triangles 13.2 -> 10.9 ms;
                                       // round 3, with the LDS the exact-size rings freed: 10 -> 12 entries, 5.2 M triangles 6.65 -> 6.51 ms

// dynamic LDS bytes of a tracing workgroup (lane kernel)
inline size_t traceLdsBytes(uint32_t kind, uint32_t n_bins, uint32_t block, uint32_t depth, bool stack16) {
    size_t hist = ((size_t)n_bins * 4 + 15) & ~(size_t)15;
    size_t stack = kind == THX_TRACER_SCENE_FORWARD ? (size_t)(block / 64) * depth * 64 * (stack16 ? 2 : 4) : 0;
    return hist + stack;
}
// dynamic LDS bytes of the staged scene kernel (dev_staged.hpp: StagedLayout)
// (nodeBytes: the copy of the scene's pair records the Variant::LDSN kernels keep behind everything else)
// (exactRings: ring capacity = the pool rounded up to 64 instead of to a power of two, dev_staged.hpp: StagedLayout::qcap)
size_t stagedLdsBytes(bool polarized, uint32_t n_bins, uint32_t stackBytesPerThread, uint32_t pool, uint32_t threads, size_t nodeBytes = 0, bool exactRings = false);
uint32_t stagedPoolFor(bool polarized, uint32_t n_bins, uint32_t stackBytesPerThread, size_t budget, uint32_t threads, size_t nodeBytes = 0, bool exactRings = false);
uint32_t stagedRingCap(uint32_t pool, bool exactRings);
// global memory (bytes) of one workgroup's cold photon state (dev_staged.hpp: StagedLayout::coldBytes)
size_t stagedColdBytes(bool polarized, uint32_t pool);
constexpr uint32_t kStagedBlock = 1024;
#ifndef THX_DEBUG_THREADS
#define THX_DEBUG_THREADS 768
#endif
constexpr uint32_t kStagedBlockDebug = THX_DEBUG_THREADS; // instrumented variants: 12 wavefronts, 168 registers per lane (dev_staged.hpp: kSchedThreadsDebug)
constexpr uint32_t kStagedBlockDebugWidePol = 512;        // ... the polarized ones of the 4-wide walk: 8 wavefronts, 256 registers (168 left them 10 short)
// staged kernel: the pair records of a scene go into LDS when they are at most kLdsNodeMaxBytes and still leave a pool of
// kLdsNodeMinPool photons (measured: profiles/r03_lds_nodes.txt)
constexpr size_t kLdsNodeMaxBytes = 48 * 1024;
constexpr uint32_t kLdsNodeMinPool = 1056;
// staged kernel: photons per workgroup beyond which a larger pool no longer pays. The faster the stages, the more photons a
// workgroup can turn over: 1264 at the end of round 2, 1344 with 64 B per photon, and with the kernels of the end of round 3
// config 2 read through the vector cache (THX_LDS_NODES=0) runs 2.58 / 2.48 / 2.48 ms at 1344 / 1472 / 1600; config 5 and the
// large meshes are flat from 1344 on (gpurun_out/r03/sweep20-22.txt)
constexpr uint32_t kStackLds32Wide = 14; // 4-wide records (up to three pushes per visit): 1.3 M / 5.2 M triangles -1.3 / -1.4 % against 12, 16 is slower again (profiles/r04_wide_records.txt)
constexpr uint32_t kPoolBest = 1472;
constexpr uint32_t kPoolMax = 2048; // == THX_POOL_MAX of dev_staged.hpp: the largest pool any sizing (or THX_POOL) can pick

// one function pair per compiled variant (trace_variant.hip, built once per
// KIND x POL x MIS); mode bit 0 selects the instrumented (statistics + track) kernel,
// bit 1 the staged scene kernel
typedef hipError_t (*trace_launch_fn)(uint32_t mode, const LaunchArgs& args, uint32_t grid,
                                      uint32_t block, size_t lds, hipStream_t stream);
typedef int (*trace_occupancy_fn)(uint32_t mode, uint32_t block, size_t lds);

struct TraceVariant {
    trace_launch_fn launch;
    trace_occupancy_fn occupancy;
};
// index = kind * 4 + pol * 2 + mis
const TraceVariant& traceVariant(uint32_t kind, bool pol, bool mis);

// `error` (may be null): device error word copied to out[n_bins]; `mirror` (may be null): second destination of curve + error word
hipError_t launchReduce(const float* slabs, float* out, uint32_t n_bins, uint32_t n_hist,
                        float norm, const uint32_t* error, float* mirror, hipStream_t stream);
// response.replay.glsl: one response() call per item of the input queue
hipError_t launchReplay(const LaunchArgs& args, const uint32_t* in_queue, uint32_t in_capacity, bool in_polarized,
                        uint32_t grid, hipStream_t stream);
constexpr uint32_t kReplayBlock = 256;
// lightsource.sample.glsl: dispatch.batchSize samples into the two output queues
// (with_dim: a third column of the wavelength queue takes every sample's RNG dimension after both draws -- what a tracer that reads
// its sources from these queues goes on with, THX_FLAG_SOURCES_PRESAMPLED)
hipError_t launchSampler(const LaunchArgs& args, bool polarized, uint32_t* light_out, uint32_t* lam_out, uint32_t capacity,
                         bool with_dim, hipStream_t stream);
// target.sample.glsl / target_guide.sample.glsl probes: out[n][25] resp. out[n][13]
hipError_t launchProbe(bool guide, const LaunchArgs& args, const float* lo, const float* hi, uint32_t n, float* out,
                       hipStream_t stream);
// one library function per element (unitProbeKernel): THX_UNIT_* of theia_hip.h
hipError_t launchUnitProbe(uint32_t which, uint64_t addr, const float* in, uint32_t n_in, uint32_t n, float* out, uint32_t n_out,
                           hipStream_t stream);
hipError_t launchRngFill(const thx_philox_params& p, uint32_t base_stream, uint32_t n_streams,
                         uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32,
                         float* out_f32, hipStream_t stream);
hipError_t launchSobolFill(uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim, uint32_t n_samples,
                           uint32_t sample_dim, float* out, hipStream_t stream);
// one idle wavefront for `ticks` x 10 ns; when[0..1] = its start / end on the device's realtime counter (thx_ctx_concurrent_kernels)
hipError_t launchSpin(unsigned long long ticks, unsigned long long* when, hipStream_t stream);
hipError_t launchQuery(const SceneView& scene, const float* rays7, uint32_t n, float* out6,
                       hipStream_t stream);

} // namespace thx
