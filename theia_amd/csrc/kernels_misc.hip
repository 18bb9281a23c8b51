// Small kernels: histogram slab reduction, RNG probe, ray-query probe; plus the
// table that maps (kind, polarized, mis) to the separately compiled variants.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "kernels.h"
#include "dev_staged.hpp"
#include "dev_cherenkov.hpp"

namespace thx {

#define THX_DECL(k, p, m)                                                                        \
    hipError_t traceLaunch_##k##_##p##_##m(uint32_t, const LaunchArgs&, uint32_t, uint32_t, size_t, \
                                           hipStream_t);                                         \
    int traceOccupancy_##k##_##p##_##m(uint32_t, uint32_t, size_t);
THX_DECL(0, 0, 0) THX_DECL(0, 0, 1) THX_DECL(0, 1, 0) THX_DECL(0, 1, 1)
THX_DECL(1, 0, 0) THX_DECL(1, 0, 1) THX_DECL(1, 1, 0) THX_DECL(1, 1, 1)
#undef THX_DECL

const TraceVariant& traceVariant(uint32_t kind, bool pol, bool mis) {
#define THX_ENT(k, p, m) TraceVariant{traceLaunch_##k##_##p##_##m, traceOccupancy_##k##_##p##_##m}
    static const TraceVariant table[8] = {THX_ENT(0, 0, 0), THX_ENT(0, 0, 1), THX_ENT(0, 1, 0),
                                          THX_ENT(0, 1, 1), THX_ENT(1, 0, 0), THX_ENT(1, 0, 1),
                                          THX_ENT(1, 1, 0), THX_ENT(1, 1, 1)};
#undef THX_ENT
    return table[(kind & 1u) * 4 + (pol ? 2 : 0) + (mis ? 1 : 0)];
}

size_t stagedLdsBytes(bool polarized, uint32_t n_bins, uint32_t stackBytesPerThread, uint32_t pool, uint32_t threads, size_t nodeBytes, bool exactRings) {
    return polarized ? StagedLayout<true>::bytes(n_bins, stackBytesPerThread, pool, threads, nodeBytes, exactRings)
                     : StagedLayout<false>::bytes(n_bins, stackBytesPerThread, pool, threads, nodeBytes, exactRings);
}
uint32_t stagedPoolFor(bool polarized, uint32_t n_bins, uint32_t stackBytesPerThread, size_t budget, uint32_t threads, size_t nodeBytes, bool exactRings) {
    return polarized ? StagedLayout<true>::poolFor(n_bins, stackBytesPerThread, budget, threads, nodeBytes, exactRings)
                     : StagedLayout<false>::poolFor(n_bins, stackBytesPerThread, budget, threads, nodeBytes, exactRings);
}
uint32_t stagedRingCap(uint32_t pool, bool exactRings) { return StagedLayout<false>::qcap(pool, exactRings); }
size_t stagedColdBytes(bool polarized, uint32_t pool) { return polarized ? StagedLayout<true>::coldBytes(pool) : StagedLayout<false>::coldBytes(pool); }
static_assert(kStagedBlock == kSchedThreads && kStagedBlockDebug == kSchedThreadsDebug, "staged block sizes");
static_assert(kPoolMax == THX_POOL_MAX, "pool bound");

// estimator.reduce.glsl:17-35: out[b] = norm * sum_h slab[h][b] with compensated
// (Kahan) summation. The reference walks the slabs with one thread per bin; with
// a few hundred slabs that serial chain is latency bound (~110 us measured), so
// here one wavefront owns a bin: lane l Kahan-sums slabs l, l+64, ... and the 64
// partials (value + compensation) are combined in a fixed butterfly order, which
// keeps the result deterministic for given slab contents.
// (-ffp-contract=off keeps the compensation from being folded away.)
// `error` (optional): the tracer's device error word; its value lands behind the curve, out[n_bins], so that ONE copy brings the
// host both (thx_tracer_fetch_hist).
// `mirror` (optional): a second destination, the tracer's pinned host buffer (mapped into the device's address space): the curve and
// the error word land there without a copy command behind the kernel.
__global__ void __launch_bounds__(64) reduceKernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                   uint32_t n_bins, uint32_t n_hist, float norm, const uint32_t* __restrict__ error,
                                                   float* __restrict__ mirror) {
    const uint32_t bin = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    float result = 0.0f, c = 0.0f;
    for (uint32_t h = lane; h < n_hist; h += 64) {
        float y = slabs[(size_t)h * n_bins + bin] - c;
        float t = result + y;
        c = (t - result) - y;
        result = t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float other = __shfl_xor(result, off, 64);
        float otherC = __shfl_xor(c, off, 64);
        // Kahan-add the partner's partial sum (its compensation is a correction to it)
        float y = (other - otherC) - c;
        float t = result + y;
        c = (t - result) - y;
        result = t;
    }
    if (lane == 0) {
        out[bin] = result * norm;
        if (mirror != nullptr) mirror[bin] = result * norm;
        if (error != nullptr && bin == 0) {
            const float e = __uint_as_float(*error);
            out[n_bins] = e;
            if (mirror != nullptr) mirror[n_bins] = e;
        }
    }
}

hipError_t launchReduce(const float* slabs, float* out, uint32_t n_bins, uint32_t n_hist,
                        float norm, const uint32_t* error, float* mirror, hipStream_t stream) {
    if (n_bins == 0) return hipSuccess;
    hipLaunchKernelGGL(reduceKernel, dim3(n_bins), dim3(64), 0, stream, slabs, out, n_bins, n_hist, norm, error, mirror);
    return hipGetLastError();
}

// HitReplay (response.replay.glsl:37-52): hit idx of the input queue -> response(hit, idx, dim = 0).
// A fixed grid strides over the queue; the histogram responses bin into the workgroup's LDS
// histogram (flushed to its slab, summed by reduceKernel), the record / store-time responses
// append to the output queue. Input and output are structure-of-arrays queues: a wavefront reads
// and writes 256 contiguous bytes per column.
__global__ void __launch_bounds__(kReplayBlock) replayKernel(const LaunchArgs args, const uint32_t* __restrict__ in,
                                                             uint32_t in_capacity, uint32_t in_polarized) {
    using V = Variant<false, true, false, false, 0, 1>;
    const uint32_t tid = threadIdx.x;
    const uint32_t n_bins = args.n_bins;
    float* hist = ldsHist();
    for (uint32_t i = tid; i < n_bins; i += blockDim.x) hist[i] = 0.0f;
    __syncthreads();

    Lane<V> L;
    L.a = &args;
    L.stackBase = 0;
    L.cnt.clear();
    const uint32_t count = min(in[0], in_capacity);
    const float* cols = reinterpret_cast<const float*>(in + 1);
    const uint32_t tail = in_polarized ? 16u : 9u; // column of the wavelength
    for (uint32_t idx = blockIdx.x * blockDim.x + tid; idx < count; idx += gridDim.x * blockDim.x) {
        auto col = [&](uint32_t c) { return cols[(size_t)c * in_capacity + idx]; };
        L.idx = idx; L.local = idx; L.dim = 0;
        const float time = col(tail + 1), contrib = col(tail + 2);
        if (args.hits == nullptr) {
            response(L, HitItem{time, contrib});
        } else if (args.hit_kind == THX_RESPONSE_STORE_TIME) {
            // response.time.store.glsl:14-25 with response.uniform.glsl: value = contrib
            if (draw(L) < contrib) {
                const uint32_t id = atomicAdd(args.hits, 1u);
                if (id < args.hit_capacity) {
                    float* t = reinterpret_cast<float*>(args.hits + 1) + id;
                    *t = time;
                    if (args.hit_polarized) *reinterpret_cast<int*>(t + args.hit_capacity) = __float_as_int(col(tail + 3));
                }
            }
        } else {
            f4 stokes{1.0f, 0.0f, 0.0f, 0.0f};
            f3 polRef{0.0f, 0.0f, 0.0f};
            if (in_polarized) {
                stokes = f4{col(9), col(10), col(11), col(12)};
                polRef = f3{col(13), col(14), col(15)};
            }
            recordHit(L, f3{col(0), col(1), col(2)}, f3{col(3), col(4), col(5)}, f3{col(6), col(7), col(8)}, stokes, polRef,
                      col(tail), time, contrib, __float_as_int(col(tail + 3)));
        }
    }
    __syncthreads();
    float* slab = args.slabs + (size_t)blockIdx.x * n_bins;
    for (uint32_t i = tid; i < n_bins; i += blockDim.x) slab[i] = hist[i];
}

hipError_t launchReplay(const LaunchArgs& args, const uint32_t* in_queue, uint32_t in_capacity, bool in_polarized,
                        uint32_t grid, hipStream_t stream) {
    const size_t lds = ((size_t)args.n_bins * 4 + 15) & ~(size_t)15;
    hipLaunchKernelGGL(replayKernel, dim3(grid), dim3(kReplayBlock), lds, stream, args, in_queue, in_capacity,
                       in_polarized ? 1u : 0u);
    return hipGetLastError();
}

// LightSampler (lightsource.sample.glsl:52-75): sample idx + base of the wavelength and light source
// -> item idx of the two output queues (count + structure-of-arrays columns)
template <bool POL>
THX_DEV void putSample(uint32_t idx, const SourceRay& ray, const WavelengthSample& photon, uint32_t dim, uint32_t* __restrict__ light_out,
                       uint32_t* __restrict__ lam_out, uint32_t capacity, bool with_dim) {
    float* col = reinterpret_cast<float*>(light_out + 1) + idx;
    auto put = [&](float v) { *col = v; col += capacity; };
    put(ray.position.x); put(ray.position.y); put(ray.position.z);
    put(ray.direction.x); put(ray.direction.y); put(ray.direction.z);
    if constexpr (POL) {
        put(ray.stokes.x); put(ray.stokes.y); put(ray.stokes.z); put(ray.stokes.w);
        put(ray.polRef.x); put(ray.polRef.y); put(ray.polRef.z);
    }
    put(ray.startTime); put(ray.contrib);
    float* w = reinterpret_cast<float*>(lam_out + 1) + idx;
    w[0] = photon.wavelength;
    w[capacity] = photon.contrib;
    if (with_dim) w[2u * capacity] = __uint_as_float(dim); // a tracer's own queue: the RNG dimension the photon goes on with
}

template <bool POL>
__global__ void __launch_bounds__(kReplayBlock) samplerKernel(const LaunchArgs args, uint32_t* __restrict__ light_out,
                                                              uint32_t* __restrict__ lam_out, uint32_t capacity, bool with_dim) {
    using V = Variant<POL, true, false, false, 0, 1>;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t count = args.p.dispatch.batchSize;
    if (idx >= count) return;
    Lane<V> L;
    L.a = &args;
    L.stackBase = 0;
    L.cnt.clear();
    L.idx = idx + args.p.photon_offset;
    L.local = L.idx;
    L.dim = 0;
    const WavelengthSample photon = sampleWavelength(L);
    const uint64_t medium = (uint64_t)args.p.volume.medium[0] | ((uint64_t)args.p.volume.medium[1] << 32); // SampleParams.medium
    const MediumConstants constants = lookUpMedium(medium, photon.wavelength, L.cnt);
    const SourceRay ray = (args.flags & THX_FLAG_LIGHT_KIND_MASK) ? sampleCherenkovFamily(L, photon.wavelength, constants.n) : sampleLight(L);
    putSample<POL>(idx, ray, photon, L.dim, light_out, lam_out, capacity, with_dim);
    if (idx == 0) {
        light_out[0] = count;
        lam_out[0] = count;
    }
}

// The same for a ParticleCascadeLightSource, whose depth sample comes out of a rejection loop (random.gamma.glsl) that the shader's
// constants make long -- 13.5 proposals per sample on average at alpha = 6.4 (8.5 at 4.2), the slowest of 64 lanes a multiple of that: with one
// photon per lane a wavefront runs the loop at a third of its lanes. Here a wavefront works through kCascadePerWave photons: in the
// loop a lane whose proposal was accepted takes the next photon of the wave's range at once (ballot + prefix count: no atomics, and
// which lane draws which photon changes nothing -- a photon's random numbers are addressed by its index), the accepted samples and
// the RNG dimensions behind them wait in LDS, and the rest of the sampling runs afterwards, photon by photon, with all lanes.
// Same arithmetic per photon as samplerKernel (tests/test_reference_cherenkov.py holds both against the oracle item by item).
constexpr uint32_t kCascadePerWave = 1024;
template <bool POL>
__global__ void __launch_bounds__(kReplayBlock) cascadeSamplerKernel(const LaunchArgs args, uint32_t* __restrict__ light_out,
                                                                     uint32_t* __restrict__ lam_out, uint32_t capacity, bool with_dim) {
    using V = Variant<POL, true, false, false, 0, 1>;
    __shared__ float depth[kReplayBlock / 64][kCascadePerWave];
    __shared__ uint32_t dimAfter[kReplayBlock / 64][kCascadePerWave];
    const uint32_t count = args.p.dispatch.batchSize;
    const uint32_t lane = threadIdx.x & 63u, waveInBlock = threadIdx.x >> 6;
    const uint32_t first = (blockIdx.x * (kReplayBlock / 64) + waveInBlock) * kCascadePerWave;
    if (first >= count) return; // (the whole wavefront)
    const uint32_t last = min(first + kCascadePerWave, count);
    float* const zs = depth[waveInBlock];
    uint32_t* const ds = dimAfter[waveInBlock];
    Lane<V> L;
    L.a = &args;
    L.stackBase = 0;
    L.cnt.clear();
    const GammaGA g = gammaSetup(args.p.light.cascade.a_long);
    // the depth samples
    uint32_t next = first; // (the same in every lane)
    uint32_t mine = 0;
    bool have = false;
    float scale = 1.0f;
    while (true) {
        const unsigned long long idle = __ballot(!have);
        if (!have) {
            const uint32_t cand = next + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (cand < last) {
                mine = cand;
                have = true;
                L.idx = mine + args.p.photon_offset;
                L.local = L.idx;
                L.dim = 0;
                (void)sampleWavelength(L); // (its draws come first)
                scale = gammaScale(L, g);
            }
        }
        next += (uint32_t)__popcll(idle);
        if (__ballot(have) == 0ull) break;
        float x;
        if (have && gammaPropose(L, g, x)) {
            zs[mine - first] = scale * x;
            ds[mine - first] = L.dim;
            have = false;
        }
    }
    // everything else, one photon per lane and pass
    const uint64_t medium = (uint64_t)args.p.volume.medium[0] | ((uint64_t)args.p.volume.medium[1] << 32);
    for (uint32_t idx = first + lane; idx < last; idx += 64u) {
        L.idx = idx + args.p.photon_offset;
        L.local = L.idx;
        L.dim = 0;
        const WavelengthSample photon = sampleWavelength(L);
        const MediumConstants constants = lookUpMedium(medium, photon.wavelength, L.cnt);
        L.dim = ds[idx - first];
        const SourceRay ray = cascadeRay<V>(L, photon.wavelength, constants.n, zs[idx - first]);
        putSample<POL>(idx, ray, photon, L.dim, light_out, lam_out, capacity, with_dim);
    }
    if (first == 0u && lane == 0u) {
        light_out[0] = count;
        lam_out[0] = count;
    }
}

hipError_t launchSampler(const LaunchArgs& args, bool polarized, uint32_t* light_out, uint32_t* lam_out, uint32_t capacity,
                         bool with_dim, hipStream_t stream) {
    const uint32_t count = args.p.dispatch.batchSize;
    if (count == 0) return hipSuccess;
    if ((args.flags & THX_FLAG_LIGHT_KIND_MASK) == THX_FLAG_LIGHT_CASCADE) {
        const uint32_t perBlock = (kReplayBlock / 64) * kCascadePerWave;
        const uint32_t grid = (count + perBlock - 1) / perBlock;
        if (polarized)
            hipLaunchKernelGGL(cascadeSamplerKernel<true>, dim3(grid), dim3(kReplayBlock), 0, stream, args, light_out, lam_out, capacity, with_dim);
        else
            hipLaunchKernelGGL(cascadeSamplerKernel<false>, dim3(grid), dim3(kReplayBlock), 0, stream, args, light_out, lam_out, capacity, with_dim);
        return hipGetLastError();
    }
    const uint32_t grid = (count + kReplayBlock - 1) / kReplayBlock;
    if (polarized)
        hipLaunchKernelGGL(samplerKernel<true>, dim3(grid), dim3(kReplayBlock), 0, stream, args, light_out, lam_out, capacity, with_dim);
    else
        hipLaunchKernelGGL(samplerKernel<false>, dim3(grid), dim3(kReplayBlock), 0, stream, args, light_out, lam_out, capacity, with_dim);
    return hipGetLastError();
}

// theia.testing.TargetSampler / TargetGuideSampler (target.sample.glsl, target_guide.sample.glsl)
template <class V>
THX_DEV void probeObserver(Lane<V>& L, const float* lo, const float* hi, f3& observer, f3& direction) {
    float a = draw(L), b = draw(L), c = draw(L);
    observer = f3{mixf(lo[0], hi[0], a), mixf(lo[1], hi[1], b), mixf(lo[2], hi[2], c)};
    float u0, u1;
    draw2(L, u0, u1);
    direction = sphericalToCartessian(TWO_PI * u0, 2.0f * u1 - 1.0f); // util.sample.glsl:84-88
}
// target.sample.glsl:19-47 checkSample / checkIntersect
THX_DEV void putTargetSample(float* o, f3 observer, const TargetSample& ts, bool checkDir, f3 direction) {
    const float expD = distance(ts.position, observer);
    const f3 expPos = mul(ts.worldToObj, ts.position) + ts.offset;
    const f3 expNrm = normalize(mul(ts.worldToObj, ts.normal));
    int error = 0;
    error += (int)(fabsf(expD - ts.dist) > 1e-5f);
    error += (int)(length(expPos - ts.objPosition) > 1e-5f) << 1;
    error += (int)(length(expNrm - ts.objNormal) > 1e-5f) << 2;
    if (checkDir) error += (int)(length(normalize(ts.position - observer) - direction) > 1e-5f) << 3;
    o[0] = ts.position.x; o[1] = ts.position.y; o[2] = ts.position.z;
    o[3] = ts.normal.x; o[4] = ts.normal.y; o[5] = ts.normal.z;
    o[6] = ts.prob;
    o[7] = __uint_as_float(ts.valid ? 1u : 0u);
    o[8] = __int_as_float(error);
}
struct ProbeBox { float lo[3], hi[3]; };
__global__ void probeTargetKernel(const LaunchArgs args, ProbeBox box, uint32_t n, float* out) {
    using V = Variant<false, true, false, false, 0, 1>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Lane<V> L;
    L.a = &args; L.stackBase = 0; L.cnt.clear();
    L.idx = i; L.local = i; L.dim = 0;
    f3 observer, direction;
    probeObserver(L, box.lo, box.hi, observer, direction);
    const TargetSample s = sampleTarget(L, args.p.target, observer, args.flags);
    const TargetSample h = intersectTarget(args.p.target, observer, direction, args.flags);
    float* o = out + (size_t)i * 25;
    o[0] = observer.x; o[1] = observer.y; o[2] = observer.z;
    o[3] = direction.x; o[4] = direction.y; o[5] = direction.z;
    putTargetSample(o + 6, observer, s, false, direction);
    putTargetSample(o + 15, observer, h, true, direction);
    o[24] = __uint_as_float(isOccludedByTarget(args.p.target, observer, args.flags) ? 1u : 0u);
}
__global__ void probeGuideKernel(const LaunchArgs args, ProbeBox box, uint32_t n, float* out) {
    using V = Variant<false, true, false, false, 0, 1>;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Lane<V> L;
    L.a = &args; L.stackBase = 0; L.cnt.clear();
    L.idx = i; L.local = i; L.dim = 0;
    f3 observer, direction;
    probeObserver(L, box.lo, box.hi, observer, direction);
    const GuideSample s = sampleTargetGuide(L, observer);
    const GuideSample e = evalTargetGuide(args.p.guide, observer, direction, args.flags);
    float* o = out + (size_t)i * 13;
    o[0] = observer.x; o[1] = observer.y; o[2] = observer.z;
    o[3] = s.dir.x; o[4] = s.dir.y; o[5] = s.dir.z; o[6] = s.dist; o[7] = s.prob;
    o[8] = e.dir.x; o[9] = e.dir.y; o[10] = e.dir.z; o[11] = e.dist; o[12] = e.prob;
}
hipError_t launchProbe(bool guide, const LaunchArgs& args, const float* lo, const float* hi, uint32_t n, float* out,
                       hipStream_t stream) {
    if (n == 0) return hipSuccess;
    ProbeBox box;
    for (int k = 0; k < 3; ++k) { box.lo[k] = lo[k]; box.hi[k] = hi[k]; }
    const uint32_t grid = (n + 255u) / 256u;
    if (guide) hipLaunchKernelGGL(probeGuideKernel, dim3(grid), dim3(256), 0, stream, args, box, n, out);
    else hipLaunchKernelGGL(probeTargetKernel, dim3(grid), dim3(256), 0, stream, args, box, n, out);
    return hipGetLastError();
}

// Unit probe: one library function of the photon path per element, what the reference's shader unit
// tests do with a tiny test kernel (tests/shader/*.test.glsl; tests/test_lookup.py, test_scatter.py,
// test_polarization.py, test_math.py). in[n][n_in] -> out[n][n_out]; `addr` = table or medium address.
__global__ void unitProbeKernel(uint32_t which, uint64_t addr, const float* __restrict__ in, uint32_t n_in, uint32_t n,
                                float* __restrict__ out, uint32_t n_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* x = in + (size_t)i * n_in;
    float* o = out + (size_t)i * n_out;
    Counters<false> cnt;
    switch (which) {
    case THX_UNIT_LOOKUP: // lookup.glsl:14-28
        o[0] = lookUp(addr, x[0], x[1], cnt);
        break;
    case THX_UNIT_LOOKUP_MEDIUM: { // material.glsl:50-74
        const MediumConstants c = lookUpMedium(addr, x[0], cnt);
        o[0] = c.n; o[1] = c.vg; o[2] = c.mu_s; o[3] = c.mu_e;
        break;
    }
    case THX_UNIT_SCATTER_DIR: { // scatter.volume.glsl:7-26
        const f3 d = scatterDir(f3{x[0], x[1], x[2]}, x[3], x[4]);
        o[0] = d.x; o[1] = d.y; o[2] = d.z;
        break;
    }
    case THX_UNIT_FRESNEL: { // scatter.surface.glsl:21-51
        const Reflectance r = fresnelCoefficients(x[0], x[1], x[2]);
        o[0] = r.r_s; o[1] = r.r_p;
        break;
    }
    case THX_UNIT_SCATTER_PROB: // scatter.volume.glsl:54-65
        o[0] = scatterProb(addr, f3{x[0], x[1], x[2]}, f3{x[3], x[4], x[5]}, cnt);
        break;
    case THX_UNIT_SCATTER_SAMPLE: { // scatter.volume.glsl:47-52
        float p;
        const f3 d = scatter(addr, f3{x[0], x[1], x[2]}, x[3], x[4], p, cnt);
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = p;
        break;
    }
    case THX_UNIT_ROTATE_POLREF: { // polarization.glsl:38-68
        f3 ref;
        float c, s;
        rotatePolRef(f3{x[0], x[1], x[2]}, f3{x[3], x[4], x[5]}, f3{x[6], x[7], x[8]}, ref, c, s);
        o[0] = ref.x; o[1] = ref.y; o[2] = ref.z; o[3] = c; o[4] = s;
        break;
    }
    case THX_UNIT_ALIGN_POLREF: { // polarization.glsl:21-34
        float c, s;
        alignPolRef(f3{x[0], x[1], x[2]}, f3{x[3], x[4], x[5]}, f3{x[6], x[7], x[8]}, c, s);
        o[0] = c; o[1] = s;
        break;
    }
    case THX_UNIT_PHASE_MATRIX: { // polarization.glsl:88-107 with ray.scatter.glsl:46-54
        const f4 v = phaseRotateApply(addr, x[0], x[1], x[2], f4{x[3], x[4], x[5], x[6]}, cnt);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        break;
    }
    case THX_UNIT_POLARIZER: { // polarization.glsl:109-120
        const f4 v = polarizerApply(x[0], x[1], f4{x[2], x[3], x[4], x[5]});
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        break;
    }
    case THX_UNIT_EMISSION_ANGLE: // lightsource.particles.common.glsl:69-97
        o[0] = particleSampleEmissionAngle(x[0], x[1], x[2], x[3]);
        break;
    case THX_UNIT_EMISSION_ANGLE_RANGE: { // lightsource.particles.common.glsl:103-136
        float contrib;
        o[0] = particleSampleEmissionAngleRange(x[0], x[1], x[2], x[3], x[4], x[5], contrib);
        o[1] = contrib;
        break;
    }
    case THX_UNIT_EMISSION_PDF: // lightsource.particles.common.glsl:138-154
        o[0] = particleEvalEmissionAngle(x[0], x[1], x[2], x[3]);
        break;
    case THX_UNIT_FRANK_TAMM: // lightsource.cherenkov.common.glsl:16-21 / :37-42, lightsource.particles.common.glsl:53-62
        o[0] = frankTammCherenkov(x[0], x[1], false);
        o[1] = frankTammCherenkov(x[0], x[1], true);
        o[2] = frankTammParticle(x[1], x[0]);
        break;
    default: // THX_UNIT_SIGN_BIT, math.glsl:18-20
        o[0] = signBit(x[0]);
        break;
    }
}
hipError_t launchUnitProbe(uint32_t which, uint64_t addr, const float* in, uint32_t n_in, uint32_t n, float* out, uint32_t n_out,
                           hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unitProbeKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, which, addr, in, n_in, n, out, n_out);
    return hipGetLastError();
}

// random.sink.glsl equivalent: raw words + floats behind random(stream, dim)
__global__ void rngFillKernel(thx_philox_params p, uint32_t base_stream, uint32_t n_streams,
                              uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32,
                              float* out_f32) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n_streams * n_samples;
    if (i >= total) return;
    uint32_t s = (uint32_t)(i / n_samples), k = (uint32_t)(i % n_samples);
    uint32_t v = random_u32(p, base_stream + s, first_dim + k);
    if (out_u32) out_u32[i] = v;
    if (out_f32) out_f32[i] = normalizeUint(v);
}

__global__ void sobolFillKernel(uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim,
                                uint32_t n_samples, uint32_t sample_dim, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n_streams * n_samples;
    if (i >= total) return;
    uint32_t s = (uint32_t)(i / n_samples), k = (uint32_t)(i % n_samples);
    uint32_t dim = first_dim + k; // random_s / random2D_s: the sink addresses every sample itself
    if (sample_dim == 2) rndSobol2(seed, base_stream + s, dim, out[2 * i], out[2 * i + 1]);
    else out[i] = rndSobol(seed, base_stream + s, dim);
}

hipError_t launchSobolFill(uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim, uint32_t n_samples,
                           uint32_t sample_dim, float* out, hipStream_t stream) {
    size_t total = (size_t)n_streams * n_samples;
    uint32_t grid = (uint32_t)((total + 255) / 256);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(sobolFillKernel, dim3(grid), dim3(256), 0, stream, seed, base_stream, n_streams, first_dim, n_samples,
                       sample_dim, out);
    return hipGetLastError();
}

hipError_t launchRngFill(const thx_philox_params& p, uint32_t base_stream, uint32_t n_streams,
                         uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32,
                         float* out_f32, hipStream_t stream) {
    size_t total = (size_t)n_streams * n_samples;
    uint32_t block = 256;
    uint32_t grid = (uint32_t)((total + block - 1) / block);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(rngFillKernel, dim3(grid), dim3(block), 0, stream, p, base_stream, n_streams,
                       first_dim, n_samples, out_u32, out_f32);
    return hipGetLastError();
}

// One wavefront that does nothing for `ticks` of the 100-MHz realtime counter and notes when it started and ended: the unit of
// thx_ctx_concurrent_kernels, which launches one on each of several streams and counts how many of them were running at once.
__global__ void __launch_bounds__(64) spinKernel(unsigned long long ticks, unsigned long long* when) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t1 = t0;
    for (unsigned int polls = 0; t1 - t0 < ticks && polls < (1u << 22); ++polls) { // (a counter that does not tick must not hang the queue)
        __builtin_amdgcn_s_sleep(32);
        t1 = __builtin_amdgcn_s_memrealtime();
    }
    if (when != nullptr && threadIdx.x == 0) { when[0] = t0; when[1] = t1; }
}
hipError_t launchSpin(unsigned long long ticks, unsigned long long* when, hipStream_t stream) {
    hipLaunchKernelGGL(spinKernel, dim3(1), dim3(64), 0, stream, ticks, when);
    return hipGetLastError();
}

// closest-hit probe for BVH parity tests
__global__ void __launch_bounds__(kTraceBlock) queryKernel(LaunchArgs args, const float* rays7,
                                                           uint32_t n, float* out6) {
    using V = Variant<false, false, false, false>;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Lane<V> L;
    L.a = &args;
    L.stackBase = (threadIdx.x >> 6) * args.stack_wave_words;
    L.idx = 0; L.local = 0; L.dim = 0;
    if (i >= n) return;
    const float* r = rays7 + 7 * (size_t)i;
    QueryHit h = rayQuery(L, f3{r[0], r[1], r[2]}, f3{r[3], r[4], r[5]}, r[6]);
    float* o = out6 + 6 * (size_t)i;
    o[0] = h.valid ? 1.0f : 0.0f;
    o[1] = (float)h.instance;
    o[2] = (float)(h.valid ? args.scene.tris[h.prim].prim : 0u); // the triangle's index in its mesh
    o[3] = h.t;
    o[4] = h.b1;
    o[5] = h.b2;
}

// Development variant: every wavefront owns a chunk of consecutive rays and a lane that
// finishes its ray takes the next one of the chunk (refill when fewer than `thr` lanes
// are busy) -- measures what ray-length divergence costs the plain one-ray-per-lane form.
__global__ void __launch_bounds__(kTraceBlock) queryKernelPersistent(LaunchArgs args, const float* rays7,
                                                                     uint32_t n, float* out6,
                                                                     uint32_t chunk, uint32_t thr) {
    using V = Variant<false, false, false, false>;
    Lane<V> L;
    L.a = &args;
    L.stackBase = (threadIdx.x >> 6) * args.stack_wave_words;
    L.idx = 0; L.local = 0; L.dim = 0;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t next = wave * chunk;                 // wave-uniform
    const uint32_t end = min(n, next + chunk);
    Walker<V> wk(L);
    uint32_t mine = 0xFFFFFFFFu;
    for (;;) {
        const bool idle = wk.done();
        const uint64_t im = __ballot(idle);
        if (next < end && im) {
            const uint32_t rank = __popcll(im & ((1ull << lane) - 1ull));
            const uint32_t cand = next + rank;
            if (idle && mine != 0xFFFFFFFFu) {
                float* o = out6 + 6 * (size_t)mine;
                o[0] = wk.h.valid ? 1.0f : 0.0f; o[1] = (float)wk.h.instance; o[2] = (float)(wk.h.valid ? args.scene.tris[wk.h.prim].prim : 0u);
                o[3] = wk.h.t; o[4] = wk.h.b1; o[5] = wk.h.b2;
                mine = 0xFFFFFFFFu;
            }
            if (idle && cand < end) {
                const float* r = rays7 + 7 * (size_t)cand;
                mine = cand;
                wk.begin(f3{r[0], r[1], r[2]}, f3{r[3], r[4], r[5]}, r[6]);
            }
            next = min(end, next + (uint32_t)__popcll(im));
        }
        uint64_t act = __ballot(!wk.done());
        if (!act) break;
        const uint32_t need = next < end ? thr : 1u;
        while ((uint32_t)__popcll(act) >= need) {
            wk.advance();
            act = __ballot(!wk.done());
            if (!act) break;
        }
    }
    if (mine != 0xFFFFFFFFu) {
        float* o = out6 + 6 * (size_t)mine;
        o[0] = wk.h.valid ? 1.0f : 0.0f; o[1] = (float)wk.h.instance; o[2] = (float)(wk.h.valid ? args.scene.tris[wk.h.prim].prim : 0u);
        o[3] = wk.h.t; o[4] = wk.h.b1; o[5] = wk.h.b2;
    }
}

hipError_t launchQuery(const SceneView& scene, const float* rays7, uint32_t n, float* out6,
                       hipStream_t stream) {
    LaunchArgs a;
    std::memset(&a, 0, sizeof(a));
    a.scene = scene;
    a.stack_depth = kStackDepth;
    a.stack_lds = kStackDepth;
    a.stack_wave_words = kStackDepth * 64;
    a.stack16 = 0;
    a.n_bins = 0;
    uint32_t block = kTraceBlock;
    uint32_t grid = (n + block - 1) / block;
    if (grid == 0) return hipSuccess;
    size_t lds = traceLdsBytes(THX_TRACER_SCENE_FORWARD, 0, block, kStackDepth, false);
    if (const char* e = std::getenv("THX_QUERY_DEPTH")) { // development: smaller u16 stacks -> more waves per CU
        uint32_t d = std::min(32, std::max(4, std::atoi(e)));
        a.stack_depth = d;
        a.stack_lds = d;
        a.stack16 = 1;
        a.stack_wave_words = d * 32;
        lds = (size_t)(block / 64) * d * 128;
    }
    if (const char* e = std::getenv("THX_QUERY_CHUNK")) {
        uint32_t chunk = std::max(64, std::atoi(e)), thr = 48;
        if (const char* t = std::getenv("THX_QUERY_THR")) thr = std::min(64, std::max(1, std::atoi(t)));
        uint32_t waves = (n + chunk - 1) / chunk;
        grid = (waves * 64 + block - 1) / block;
        hipLaunchKernelGGL(queryKernelPersistent, dim3(grid), dim3(block), lds, stream, a, rays7, n, out6, chunk, thr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(queryKernel, dim3(grid), dim3(block), lds, stream, a, rays7, n, out6);
    return hipGetLastError();
}

} // namespace thx
