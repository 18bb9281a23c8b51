// The tracing kernel. Built once per (KIND, POL, MIS) through -D macros so the
// variants compile in parallel; each build holds the production kernel and its
// instrumented twin (event statistics + traversal counters + per-event track).
//
// Execution model (MI355X-first, not the reference's one-thread-one-photon
// dispatch): a fixed grid of 256-thread workgroups; every wavefront owns a
// contiguous photon range and keeps its 64 lanes full -- when photons finish, the
// dead lanes are found with one __ballot, ranked with a popcount prefix and
// refilled from the wave's (scalar) cursor. The RNG stream is the photon index,
// so results do not depend on the lane a photon lands in. Hits are binned with
// LDS float atomics into one partial histogram per workgroup, flushed to a slab
// and summed by reduceKernel (estimator.reduce.glsl).
#include "kernels.h"
#include "dev_staged.hpp"

#ifndef THX_KIND
#error "THX_KIND / THX_POL / THX_MIS must be defined"
#endif
// register budget: minimum wavefronts per SIMD the compiler must leave room for
#ifndef THX_WAVES_PER_SIMD
#define THX_WAVES_PER_SIMD 2
#endif

namespace thx {

template <class V, int KIND>
__global__ void __launch_bounds__(kTraceBlock, THX_WAVES_PER_SIMD) traceKernel(const LaunchArgs args) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t wavesPerBlock = blockDim.x >> 6;
    const uint32_t n_bins = args.n_bins;

    // response.histogram.glsl:20-29 initResponse
    float* hist = ldsHist();
    for (uint32_t i = tid; i < n_bins; i += blockDim.x) hist[i] = 0.0f;
    __syncthreads();

    Lane<V> L;
    L.a = &args;
    L.stackBase = ((n_bins + 3u) & ~3u) + wave * args.stack_wave_words;
    L.cnt.clear();
    L.idx = 0; L.local = 0; L.dim = 0;

    const uint32_t batch = args.p.dispatch.batchSize;
    const uint32_t gw = blockIdx.x * wavesPerBlock + wave;
    const uint64_t b64 = (uint64_t)gw * args.photons_per_wave;
    const uint32_t begin = b64 < batch ? (uint32_t)b64 : batch;
    const uint32_t end = (b64 + args.photons_per_wave) < batch ? (uint32_t)(b64 + args.photons_per_wave) : batch;
    uint32_t cursor = begin; // wave-uniform

    Ray ray;
    ray.position = f3{0, 0, 0}; ray.direction = f3{0, 0, 1}; ray.medium = 0;
    ray.wavelength = 0; ray.time = 0; ray.lin_contrib = 0; ray.log_contrib = 0;
    ray.constants = MediumConstants{1, 1, 0, 0};
    ray.stokes = f4{1, 0, 0, 0}; ray.polRef = f3{0, 0, 0};
    bool active = false;
    bool allowResponse = false;
    uint32_t step = 0;

    while (true) {
        // refill dead lanes from the wave's photon range
        // Refill only once the wave has thinned out (refill_below active lanes or fewer):
        // lanes reborn together stay in lock-step (same step index => same code path),
        // which costs fewer divergent instructions than topping up every iteration.
        const unsigned long long dead = __popcll(__ballot(active)) <= args.refill_below ? __ballot(!active) : 0ull;
        if (!active && dead != 0ull) {
            const uint32_t rank = __popcll(dead & ((1ull << lane) - 1ull));
            const uint32_t mine = cursor + rank;
            if (mine < end) {
                L.local = mine;
                L.idx = args.p.photon_offset + mine;
                L.dim = 0;
                step = 0;
                active = true;
            }
        }
        const uint32_t taken = (uint32_t)__popcll(dead);
        cursor = (end - cursor) < taken ? end : cursor + taken;
        if (__ballot(active) == 0ull) break;
        if (active) {
            if constexpr (KIND == THX_TRACER_SCENE_FORWARD) active = sceneStep(L, ray, step, allowResponse);
            else active = (args.flags & THX_FLAG_PHOTON) ? volPhotonStep(L, ray, step) : volStep(L, ray, step);
        }
    }

    // response.histogram.glsl:35-45 finalizeResponse
    __syncthreads();
    float* slab = args.slabs + (size_t)blockIdx.x * n_bins;
    for (uint32_t i = tid; i < n_bins; i += blockDim.x) slab[i] = hist[i];

    if constexpr (V::STATS) {
#pragma unroll
        for (int k = 0; k < THX_N_STATS; ++k) {
            unsigned long long v = L.cnt.v[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v != 0) atomicAdd(&args.stats[k], v);
        }
    }
}

using Plain = Variant<THX_POL != 0, THX_MIS != 0, false, false>;
using Debug = Variant<THX_POL != 0, THX_MIS != 0, true, true>;
using PlainS = Variant<THX_POL != 0, THX_MIS != 0, false, false, 1>; // 16-bit stack entries known at compile time
using DebugS = Variant<THX_POL != 0, THX_MIS != 0, true, true, 1>;
using PlainT = Variant<THX_POL != 0, THX_MIS != 0, false, false, 2>; // ... and the whole stack in LDS
using DebugT = Variant<THX_POL != 0, THX_MIS != 0, true, true, 2>;
using PlainN = Variant<THX_POL != 0, THX_MIS != 0, false, false, 3>; // ... and the pair records in LDS
using DebugN = Variant<THX_POL != 0, THX_MIS != 0, true, true, 3>;
using PlainM = Variant<THX_POL != 0, THX_MIS != 0, false, false, 4>; // 16-bit entries, deep part in global memory, pair records in LDS
using DebugM = Variant<THX_POL != 0, THX_MIS != 0, true, true, 4>;
using PlainW = Variant<THX_POL != 0, THX_MIS != 0, false, false, 5>; // 32-bit entries, 4-wide records (large meshes)
using DebugW = Variant<THX_POL != 0, THX_MIS != 0, true, true, 5>;

#if THX_KIND == 1
// staged (regrouped-in-LDS) form of the scene tracer, see dev_staged.hpp
template <class V>
__global__ void __launch_bounds__(V::STATS ? (V::WIDE && V::POL ? kStagedBlockDebugWidePol : kSchedThreadsDebug) : kSchedThreads) sceneStagedKernel(const LaunchArgs args) {
    sceneStagedBody<V>(args);
}
#endif

// staged kernel variants: 0/1 nothing known about the stack, 2/3 16-bit entries, 4/5 ... all in LDS, 6/7 ... and the
// pair records too, 8/9 16-bit entries with the deep part in global memory and the pair records in LDS, 10/11 32-bit entries and
// 4-wide records; odd = instrumented
static inline uint32_t stagedVariantIndex(uint32_t mode) {
    if (mode & 32u) return 10u + (mode & 1u); // 4-wide records
    return ((mode & 1u) ? 1u : 0u) | ((mode & 4u) ? ((mode & 8u) ? ((mode & 16u) ? 6u : 4u) : ((mode & 16u) ? 8u : 2u)) : 0u);
}

#define THX_CAT_(a, b, c, d) a##b##_##c##_##d
#define THX_CAT(a, b, c, d) THX_CAT_(a, b, c, d)
#define THX_FN(prefix) THX_CAT(prefix, THX_KIND, THX_POL, THX_MIS)

// mode: bit 0 = instrumented (statistics + track), bit 1 = staged kernel (scene only),
// bit 2 = 16-bit traversal stack entries, bit 3 = the whole stack in LDS, bit 4 = the pair records in LDS, bit 5 = 4-wide records
// (staged kernel only: variants of their own)
hipError_t THX_FN(traceLaunch_)(uint32_t mode, const LaunchArgs& args, uint32_t grid, uint32_t block,
                                size_t lds, hipStream_t stream) {
    const bool debug = mode & 1u;
#if THX_KIND == 1
    if (mode & 2u) {
        // > 64 KB of dynamic LDS needs an explicit opt-in (once per kernel)
        const void* fn[12] = {reinterpret_cast<const void*>(sceneStagedKernel<Plain>), reinterpret_cast<const void*>(sceneStagedKernel<Debug>),
                             reinterpret_cast<const void*>(sceneStagedKernel<PlainS>), reinterpret_cast<const void*>(sceneStagedKernel<DebugS>),
                             reinterpret_cast<const void*>(sceneStagedKernel<PlainT>), reinterpret_cast<const void*>(sceneStagedKernel<DebugT>),
                             reinterpret_cast<const void*>(sceneStagedKernel<PlainN>), reinterpret_cast<const void*>(sceneStagedKernel<DebugN>),
                             reinterpret_cast<const void*>(sceneStagedKernel<PlainM>), reinterpret_cast<const void*>(sceneStagedKernel<DebugM>),
                             reinterpret_cast<const void*>(sceneStagedKernel<PlainW>), reinterpret_cast<const void*>(sceneStagedKernel<DebugW>)};
        static bool optedIn[12] = {false, false, false, false, false, false, false, false, false, false, false, false};
        const uint32_t k = stagedVariantIndex(mode);
        if (!optedIn[k]) {
            hipError_t e = hipFuncSetAttribute(fn[k], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            optedIn[k] = true;
        }
        switch (k) {
        case 0: hipLaunchKernelGGL((sceneStagedKernel<Plain>), dim3(grid), dim3(block), lds, stream, args); break;
        case 1: hipLaunchKernelGGL((sceneStagedKernel<Debug>), dim3(grid), dim3(block), lds, stream, args); break;
        case 2: hipLaunchKernelGGL((sceneStagedKernel<PlainS>), dim3(grid), dim3(block), lds, stream, args); break;
        case 3: hipLaunchKernelGGL((sceneStagedKernel<DebugS>), dim3(grid), dim3(block), lds, stream, args); break;
        case 4: hipLaunchKernelGGL((sceneStagedKernel<PlainT>), dim3(grid), dim3(block), lds, stream, args); break;
        case 5: hipLaunchKernelGGL((sceneStagedKernel<DebugT>), dim3(grid), dim3(block), lds, stream, args); break;
        case 6: hipLaunchKernelGGL((sceneStagedKernel<PlainN>), dim3(grid), dim3(block), lds, stream, args); break;
        case 7: hipLaunchKernelGGL((sceneStagedKernel<DebugN>), dim3(grid), dim3(block), lds, stream, args); break;
        case 8: hipLaunchKernelGGL((sceneStagedKernel<PlainM>), dim3(grid), dim3(block), lds, stream, args); break;
        case 9: hipLaunchKernelGGL((sceneStagedKernel<DebugM>), dim3(grid), dim3(block), lds, stream, args); break;
        case 10: hipLaunchKernelGGL((sceneStagedKernel<PlainW>), dim3(grid), dim3(block), lds, stream, args); break;
        default: hipLaunchKernelGGL((sceneStagedKernel<DebugW>), dim3(grid), dim3(block), lds, stream, args); break;
        }
        return hipGetLastError();
    }
#endif
    if (debug)
        hipLaunchKernelGGL((traceKernel<Debug, THX_KIND>), dim3(grid), dim3(block), lds, stream, args);
    else
        hipLaunchKernelGGL((traceKernel<Plain, THX_KIND>), dim3(grid), dim3(block), lds, stream, args);
    return hipGetLastError();
}

int THX_FN(traceOccupancy_)(uint32_t mode, uint32_t block, size_t lds) {
    int n = 0;
    const bool debug = mode & 1u;
    hipError_t e;
#if THX_KIND == 1
    if (mode & 2u) {
        const uint32_t k = stagedVariantIndex(mode);
        switch (k) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<Plain>, (int)block, lds); break;
        case 1: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<Debug>, (int)block, lds); break;
        case 2: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<PlainS>, (int)block, lds); break;
        case 3: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<DebugS>, (int)block, lds); break;
        case 4: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<PlainT>, (int)block, lds); break;
        case 5: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<DebugT>, (int)block, lds); break;
        case 6: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<PlainN>, (int)block, lds); break;
        case 7: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<DebugN>, (int)block, lds); break;
        case 8: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<PlainM>, (int)block, lds); break;
        case 9: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<DebugM>, (int)block, lds); break;
        case 10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<PlainW>, (int)block, lds); break;
        default: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sceneStagedKernel<DebugW>, (int)block, lds); break;
        }
        return e == hipSuccess ? n : -1;
    }
#endif
    e = debug ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, traceKernel<Debug, THX_KIND>, (int)block, lds)
              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, traceKernel<Plain, THX_KIND>, (int)block, lds);
    return e == hipSuccess ? n : -1;
}

} // namespace thx
