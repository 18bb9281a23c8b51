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

#ifndef THX_KIND
#error "THX_KIND / THX_POL / THX_MIS must be defined"
#endif

namespace thx {

template <class V, int KIND>
__global__ void __launch_bounds__(kTraceBlock) traceKernel(const LaunchArgs args) {
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
    L.stackBase = ((n_bins + 3u) & ~3u) + wave * args.stack_depth * 64u + lane;
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
        const unsigned long long dead = __ballot(!active);
        if (!active) {
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
            else active = volStep(L, ray, step);
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

#define THX_CAT_(a, b, c, d) a##b##_##c##_##d
#define THX_CAT(a, b, c, d) THX_CAT_(a, b, c, d)
#define THX_FN(prefix) THX_CAT(prefix, THX_KIND, THX_POL, THX_MIS)

hipError_t THX_FN(traceLaunch_)(bool debug, const LaunchArgs& args, uint32_t grid, uint32_t block,
                                size_t lds, hipStream_t stream) {
    if (debug)
        hipLaunchKernelGGL((traceKernel<Debug, THX_KIND>), dim3(grid), dim3(block), lds, stream, args);
    else
        hipLaunchKernelGGL((traceKernel<Plain, THX_KIND>), dim3(grid), dim3(block), lds, stream, args);
    return hipGetLastError();
}

int THX_FN(traceOccupancy_)(bool debug, uint32_t block, size_t lds) {
    int n = 0;
    hipError_t e = debug ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, traceKernel<Debug, THX_KIND>, (int)block, lds)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, traceKernel<Plain, THX_KIND>, (int)block, lds);
    return e == hipSuccess ? n : -1;
}

} // namespace thx
