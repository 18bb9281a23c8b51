// Photon state, propagation, surface interaction, detector response and the
// two-level BVH walk. One lane = one photon; all per-photon state lives in
// registers. References are to /root/reference/src/theia/shader.
#pragma once
#include "dev_core.hpp"

namespace thx {

// compile-time specialisation of one kernel variant (the reference specialises
// through GLSL macros, trace.py:660-666 / :1226-1235)
template <bool POL_, bool MIS_, bool STATS_, bool TRACK_, int STACK_ = 0, int QRNG_ = -1>
struct Variant {
    static constexpr bool POL = POL_;       // POLARIZATION
    static constexpr bool MIS = MIS_;       // !DISABLE_MIS
    static constexpr bool STATS = STATS_;   // EventStatisticCallback + traversal counters
    static constexpr bool TRACK = TRACK_;   // per-event dump
    // The scrambled Sobol sampler (THX_FLAG_RNG_SOBOL) is compiled into the instrumented variants only, which the host picks for
    // it: inlined behind a run-time flag at each of the ~50 places a kernel draws a number it made the production kernels a
    // quarter larger than the 64 KB of instruction cache two CUs share (round 3: 72 KB -> 55 KB). QRNG_ = 1 compiles it into
    // a variant regardless: the small streaming kernels around the path (light sampler, hit replay, probes) have one
    // variant each and take whatever generator the host stage was given.
    static constexpr bool QRNG = QRNG_ < 0 ? STATS_ : QRNG_ != 0;
    // What is known about the traversal stack at compile time. 0: nothing (entry width and the split between LDS and global
    // memory are decided at run time); 1: 16-bit entries (scenes below 4 096 pair records / triangles / instances: every ref = index << 3 | count stays below 0x8000, which the
    // sign-extending pops of Walker<V, true> rely on -- the host checks the scene's largest ref before it picks such a variant); 2: 16-bit
    // entries and the whole stack in LDS (such scenes no deeper than the LDS share). Every push and pop loses a
    // wave-uniform branch with 1 and a lane-divergent one with 2: -1 % and -2 %. 3: as 2, and the walk reads the pair
    // records from a copy in LDS (scenes whose whole node array fits next to the photon pool): an LDS read returns in
    // 64 cycles whatever the other wavefronts do, a 16-B gather through the vector cache in 150 (alone) to 600 (sixteen
    // wavefronts gathering) when it hits and the L2's 500+ when it does not (profiles/r03_mem_latency.txt).
    // 4: as 1 (16-bit entries, deep part of the stack in global memory) with the pair records in LDS: deep scenes whose
    // records still fit.
    // 5: as 0 (32-bit entries) for scenes built with 4-wide records (SceneView::wide: large meshes), which the staged kernel walks with
    // Walker::visitStraight4.
    static constexpr bool S16 = STACK_ != 0 && STACK_ != 5;
    static constexpr bool SPILL = STACK_ != 2 && STACK_ != 3;
    static constexpr bool LDSN = STACK_ == 3 || STACK_ == 4;
    static constexpr bool WIDE = STACK_ == 5;
};

// ray.glsl:22-36 (+163-170 polarized part)
struct Ray {
    f3 position, direction;
    uint64_t medium;
    float wavelength, time, lin_contrib, log_contrib;
    MediumConstants constants;
    f4 stokes;
    f3 polRef;
};

// per-launch constants; lives in kernel-argument (scalar) memory
struct LaunchArgs {
    thx_run_params p;
    uint32_t flags;
    uint32_t path_length;
    uint32_t n_bins;
    uint32_t photons_per_wave; // contiguous photon range owned by one wavefront
    uint32_t track_events;
    uint32_t stack_depth;      // traversal stack entries per lane
    uint32_t stack_lds;        // the first stack_lds of them live in LDS, deeper ones in `spill` (== stack_depth: all in LDS)
    uint32_t stack_wave_words; // LDS words of one wavefront's stack block (stack_lds*64, or stack_lds*32 for 16-bit entries)
    uint32_t stack16;          // 1: stack entries are 16 bit (every ref of the scene < 0x8000: fewer than 4096 pair records / triangles / instances)
    uint32_t lds_node_word;    // staged kernel, Variant::LDSN: LDS word index of the copy of the pair records ...
    uint32_t lds_node_pairs;   // ... and how many records it holds (all of the scene's)
    uint32_t refill_below;     // refill dead lanes when <= this many lanes are still active
    uint32_t pool;             // staged kernel: photons held in LDS per workgroup
    uint32_t ring_cap;         // staged kernel: cells per ring (>= pool; the shadow-ray ring has twice as many) ...
    uint32_t ring_magic;       // ... ceil(2^37 / ring_cap) and
    uint32_t ring_magic_ts;    // ... ceil(2^37 / (2 ring_cap)): ring position -> lap (dev_staged.hpp: Pool::cell)
    float* slabs;              // [gridDim.x][n_bins] partial histograms
    uint32_t* spill;           // [waves of the grid][stack_depth - stack_lds][64] deep stack entries, or null
    uint32_t park_below;       // staged kernel: a walk batch sets its last rays aside once fewer than this many lanes walk (0: never)
    uint32_t park_rounds;      // ... and it has run at least this many rounds
    uint32_t* park;            // staged kernel: [gridDim.x][pool][2][8 + max(stack_depth, 8)] state of walks set aside half way (Walker::save)
    float* cold;               // staged kernel: [gridDim.x][2 or 4][pool] float4 = the cold chunks of the photon pools (dev_staged.hpp)
    uint32_t* hits;            // HitRecorder queue (response.queue.glsl layout) or null: histogram response
    uint32_t hit_capacity;     // entries per SoA column of the queue
    uint32_t hit_polarized;    // queue has the stokes / polRef columns (store-time queue: the objectId column)
    uint32_t hit_kind;         // THX_RESPONSE_RECORD or THX_RESPONSE_STORE_TIME
    const float* light_queue;      // HostLightSource samples (SoA, light_capacity floats per column) or null
    const float* wavelength_queue; // HostWavelengthSource samples (wavelength[cap], contrib[cap]) or null
    uint32_t light_capacity;
    uint32_t wavelength_capacity;
    unsigned long long* stats; // [THX_N_STATS]
    uint32_t* error;           // device error word (THX_DEVERR_* bits), read and cleared by the host
    float* track;              // [batch][track_events][THX_TRACK_WORDS] or null
    unsigned long long* prof;  // staged kernel, THX_STAGE_PROFILE builds: [3][16] cycles/rounds/items per stage
    SceneView scene;
};

// the kernel's LaunchArgs as seen through the laundered kernel-argument pointer (dev_core.hpp)
THX_DEV const THX_KERNARG LaunchArgs& coldArgs() { return *reinterpret_cast<const THX_KERNARG LaunchArgs*>(coldKernarg()); }
// the same as a plain pointer (the compiler still sees the constant address space behind the cast and
// emits scalar loads): what Lane::a points to while a stage runs whose parameters need not stay in SGPRs
THX_DEV const LaunchArgs* coldArgsPtr() { return (const LaunchArgs*)reinterpret_cast<const THX_KERNARG LaunchArgs*>(coldKernarg()); }

// Dynamic LDS of a tracing workgroup: [n_bins floats, padded to 16 B][per-wave
// traversal stacks: stack_depth x 64 words each]. Accessed through the
// __shared__ symbol so that the compiler keeps the LDS address space (ds_*).
extern __shared__ __attribute__((aligned(16))) unsigned char thx_smem[];
THX_DEV float* ldsHist() { return reinterpret_cast<float*>(thx_smem); }
THX_DEV uint32_t* ldsWords() { return reinterpret_cast<uint32_t*>(thx_smem); }

// what a lane needs besides its ray
template <class V>
struct Lane {
    const LaunchArgs* a;
    uint32_t stackBase; // LDS word index of this wavefront's stack block; entry e of lane l at e*64 + l
    Counters<V::STATS> cnt;
    uint32_t idx;   // photon index == RNG stream
    uint32_t local; // index inside the batch
    uint32_t dim;   // RNG dimension
#ifdef THX_STAGE_PROFILE
    unsigned long long* prof = nullptr; // LDS accumulators of the staged kernel's profile
#endif
};

// random(idx, dim) of the RNG the tracer was built with: Philox, or with THX_FLAG_RNG_SOBOL the
// scrambled Sobol sampler (its seed travels in philox.key[0])
template <class V>
THX_DEV float draw(Lane<V>& L) {
    if constexpr (V::QRNG) {
        if (L.a->flags & THX_FLAG_RNG_SOBOL) return rndSobol(L.a->p.philox.key[0], L.idx, L.dim);
    }
    return rnd(L.a->p.philox, L.idx, L.dim);
}
// random2D(idx, dim): two consecutive Philox draws (random.philox.glsl:79-82), or the first two Sobol
// dimensions at one scrambled index (random.sobol.glsl:178-193)
template <class V>
THX_DEV void draw2(Lane<V>& L, float& u0, float& u1) {
    if constexpr (V::QRNG) {
        if (L.a->flags & THX_FLAG_RNG_SOBOL) {
            rndSobol2(L.a->p.philox.key[0], L.idx, L.dim, u0, u1);
            return;
        }
    }
    rnd2(L.a->p.philox, L.idx, L.dim, u0, u1);
}

// ray.glsl:136-143
THX_DEV bool isRayBad(const Ray& r) {
    f3 p = r.position, d = r.direction;
    return isnan(p.x) || isnan(p.y) || isnan(p.z) || isinf(p.x) || isinf(p.y) || isinf(p.z) ||
           isnan(d.x) || isnan(d.y) || isnan(d.z) || isinf(d.x) || isinf(d.y) || isinf(d.z) ||
           length(d) <= 0.0f;
}

// ---------------------------------------------------------------- callbacks

// callback.stat.glsl:29-70 (counted per lane, flushed once per wave) and the
// optional per-event dump
template <class V>
THX_DEV void onEvent(Lane<V>& L, const Ray& r, int code, uint32_t i) {
    if constexpr (V::STATS) {
        switch (code) {
        case RESULT_CODE_RAY_CREATED: L.cnt.add(ST_CREATED); break;
        case RESULT_CODE_RAY_SCATTERED: L.cnt.add(ST_SCATTERED); break;
        case RESULT_CODE_RAY_HIT: L.cnt.add(ST_HIT); break;
        case RESULT_CODE_RAY_DETECTED: L.cnt.add(ST_DETECTED); break;
        case RESULT_CODE_VOLUME_HIT: L.cnt.add(ST_VOLUME); break;
        case RESULT_CODE_RAY_LOST: L.cnt.add(ST_LOST); break;
        case RESULT_CODE_RAY_DECAYED: L.cnt.add(ST_DECAYED); break;
        case RESULT_CODE_RAY_ABSORBED: L.cnt.add(ST_ABSORBED); break;
        case RESULT_CODE_RAY_MISSED: L.cnt.add(ST_MISSED); break;
        case RESULT_CODE_MAX_ITER: L.cnt.add(ST_MAXITER); break;
        case ERROR_CODE_MEDIA_MISMATCH: L.cnt.add(ST_MISMATCH); break;
        default: break;
        }
        if (code <= ERROR_CODE_MAX_VALUE) L.cnt.add(ST_ERROR);
    }
    if constexpr (V::TRACK) {
        if (L.a->track != nullptr && i < L.a->track_events) {
            float* t = L.a->track + ((size_t)L.local * L.a->track_events + i) * THX_TRACK_WORDS;
            t[0] = (float)code;
            t[1] = r.position.x; t[2] = r.position.y; t[3] = r.position.z;
            t[4] = r.direction.x; t[5] = r.direction.y; t[6] = r.direction.z;
            t[7] = r.time; t[8] = r.lin_contrib; t[9] = r.log_contrib;
            t[10] = r.wavelength; t[11] = (float)L.dim;
            if constexpr (V::POL) {
                t[12] = r.stokes.x; t[13] = r.stokes.y; t[14] = r.stokes.z; t[15] = r.stokes.w;
                t[16] = r.polRef.x; t[17] = r.polRef.y; t[18] = r.polRef.z;
            } else { // callback.track.glsl:84-88
                t[12] = 1.0f; t[13] = 0.0f; t[14] = 0.0f; t[15] = 0.0f;
                t[16] = 0.0f; t[17] = 0.0f; t[18] = 0.0f;
            }
            t[19] = 0.0f;
        }
    }
}

// ------------------------------------------------------ response.histogram.glsl

struct HitItem {
    float time, contrib;
};

// where a hit happened, in the object space of what was hit (ray.response.glsl createHit
// arguments). Only the HitRecorder response reads it.
struct HitFrame {
    f3 objPos, objNrm;
    m3 worldToObj;
    int objectId;
};

// response.histogram.glsl:59-68 + response.uniform.glsl:4-6. The bin update is an
// LDS float atomic (ds_add_f32), the partial histogram is flushed per workgroup.
template <class V>
THX_DEV void response(Lane<V>& L, const HitItem& item) {
    if (L.a->flags & THX_FLAG_KERNEL_HISTOGRAM) {
        // response.histogram.kernel.glsl:62-99: the hit is smeared over the bins within the
        // kernel support; Gaussian CDF by the tanh/atan approximation of Vazquez-Leal et al.
        const thx_response_params& rp = L.a->p.response;
        L.cnt.add(ST_RESPONSES);
        auto kernelCdf = [](float x) { return 0.5f + 0.5f * tanhf(7.779374467827938f * x - 55.5f * atanf(0.1257926109373887f * x)); };
        float t = item.time - rp.t0;
        int firstBin = (int)floorf((t - rp.kernelSupport) / rp.binSize);
        int lastBin = (int)ceilf((t + rp.kernelSupport) / rp.binSize);
        firstBin = max(firstBin, 0);
        lastBin = min(lastBin, (int)L.a->n_bins) - 1;
        t = (float)firstBin * rp.binSize + rp.t0 - item.time;
        const float h = 1.0f / rp.kernelBandwidth;
        float prev_cdf = kernelCdf(t * h);
        for (int i = firstBin; i <= lastBin; ++i) {
            t += rp.binSize;
            float cdf = kernelCdf(t * h);
            float w = cdf - prev_cdf;
            prev_cdf = cdf;
            atomicAdd(&ldsHist()[i], w * item.contrib);
        }
        return;
    }
    float f = floorf((item.time - L.a->p.response.t0) / L.a->p.response.binSize);
    L.cnt.add(ST_RESPONSES);
    // `uint bin = int(floor(..))`: negative values wrap and fail bin < N_BINS
    if (f >= 0.0f && f < (float)L.a->n_bins) {
        [[maybe_unused]] uint32_t bin = (uint32_t)f;
        if (bin < L.a->n_bins) atomicAdd(&ldsHist()[bin], item.contrib);
    }
}

// response.record.glsl:23-26 with SAVE_HIT of response.queue.glsl: one global atomic on the
// queue counter, then one store per column (structure of arrays, `capacity` entries each).
// A full queue drops the hit but keeps counting, like the reference's unchecked store would
// overrun; the host clamps the count it reports to the capacity.
template <class V>
THX_DEV void recordHit(Lane<V>& L, f3 pos, f3 dir, f3 nrm, f4 stokes, f3 polRef, float wavelength,
                       float time, float contrib, int objectId) {
    L.cnt.add(ST_RESPONSES);
    const uint32_t cap = L.a->hit_capacity;
    const uint32_t id = atomicAdd(L.a->hits, 1u);
    if (id >= cap) return;
    float* col = reinterpret_cast<float*>(L.a->hits + 1) + id;
    auto put = [&](float v) { *col = v; col += cap; };
    put(pos.x); put(pos.y); put(pos.z);
    put(dir.x); put(dir.y); put(dir.z);
    put(nrm.x); put(nrm.y); put(nrm.z);
    if (L.a->hit_polarized) { // HIT_QUEUE_POLARIZED; an unpolarized tracer stores (1,0,0,0), 0
        put(stokes.x); put(stokes.y); put(stokes.z); put(stokes.w);
        put(polRef.x); put(polRef.y); put(polRef.z);
    }
    put(wavelength); put(time); put(contrib);
    *reinterpret_cast<int*>(col) = objectId;
}

// ray.response.glsl:18-56 / :63-84. Only time and contribution feed the
// histogram response; the polarized contribution picks up S0 after the frame
// alignment (whose first row is (1,0,0,0), so S0 is the ray's S0).
template <class V>
THX_DEV HitItem createHit(const Ray& r) {
    float contrib = r.lin_contrib * thx_exp(r.log_contrib);
    if constexpr (V::POL) contrib *= r.stokes.x;
    return HitItem{r.time, contrib};
}

// createHit + `if (item.contrib > 0.0) response(item)` of the reference's createResponse
// functions. The HitRecorder response exists in the instrumented kernel variants only
// (the host selects them for a recording tracer), so the histogram path of the production
// kernels carries none of it.
template <class V>
THX_DEV void respond(Lane<V>& L, const Ray& r, const HitFrame& fr) {
    if constexpr (V::STATS) {
        if (L.a->hits != nullptr) {
            float contrib = r.lin_contrib * thx_exp(r.log_contrib);
            f3 objHitDir = normalize(mul(fr.worldToObj, r.direction));
            f4 stokes{1.0f, 0.0f, 0.0f, 0.0f};
            f3 hitPolRef{0.0f, 0.0f, 0.0f};
            if constexpr (V::POL) {
                hitPolRef = perpendicularTo(objHitDir, fr.objNrm);
                f3 objPolRef = normalize(mul(fr.worldToObj, r.polRef));
                float c, s;
                alignPolRef(objHitDir, objPolRef, hitPolRef, c, s);
                f4 st = rotApply(c, s, r.stokes);
                contrib *= st.x;
                stokes = f4{st.x / st.x, st.y / st.x, st.z / st.x, st.w / st.x};
            }
            if (contrib > 0.0f) {
                if (L.a->hit_kind == THX_RESPONSE_STORE_TIME) {
                    // response.time.store.glsl:14-25 with response.uniform.glsl: value = contrib is
                    // the detection probability, one draw of the photon's stream decides
                    if (draw(L) < contrib) {
                        L.cnt.add(ST_RESPONSES);
                        const uint32_t cap = L.a->hit_capacity;
                        const uint32_t id = atomicAdd(L.a->hits, 1u);
                        if (id < cap) {
                            float* t = reinterpret_cast<float*>(L.a->hits + 1) + id;
                            *t = r.time;
                            if (L.a->hit_polarized) *reinterpret_cast<int*>(t + cap) = fr.objectId;
                        }
                    }
                } else {
                    recordHit(L, fr.objPos, objHitDir, fr.objNrm, stokes, hitPolRef, r.wavelength, r.time, contrib, fr.objectId);
                }
            }
            return;
        }
    }
    HitItem item = createHit<V>(r);
    if (item.contrib > 0.0f) response(L, item);
}

// ----------------------------------------------------------- ray.scatter.glsl

// ray.scatter.glsl:46-54
template <class V>
THX_DEV void scatterPolRay(Lane<V>& L, Ray& r, uint64_t medium, f3 dir, float cos_theta) {
    f3 polRef;
    float c, s;
    rotatePolRef(r.direction, r.polRef, dir, polRef, c, s);
    r.stokes = phaseRotateApply(medium, cos_theta, c, s, r.stokes, L.cnt);
    r.polRef = polRef;
}
// ray.scatter.glsl:12-15, :70-74
template <class V>
THX_DEV void scatterRayIS(Lane<V>& L, Ray& r, uint64_t medium, f3 dir) {
    if constexpr (V::POL) {
        float cos_theta = dot(r.direction, dir);
        scatterPolRay(L, r, medium, dir, cos_theta);
    }
    r.direction = dir;
    r.lin_contrib *= r.constants.mu_s;
}
// ray.scatter.glsl:31-38, :92-100
template <class V>
THX_DEV void scatterRay(Lane<V>& L, Ray& r, uint64_t medium, float u0, float u1) {
    float cos_theta, phi;
    sampleScatterDir(medium, u0, u1, cos_theta, phi, L.cnt);
    f3 newDir = scatterDir(r.direction, cos_theta, phi);
    if constexpr (V::POL) scatterPolRay(L, r, medium, newDir, cos_theta);
    r.direction = newDir;
    r.lin_contrib *= r.constants.mu_s;
}

// --------------------------------------------------------- ray.propagate.glsl

// ray.propagate.glsl:32-49
THX_DEV float sampleScatterLength(const Ray& r, const thx_propagate_params& p, float u) {
    float dist = p.maxDist;
    float sampleCoef = p.scatterCoefficient >= 0.0f ? p.scatterCoefficient : r.constants.mu_s;
    bool canScatter = r.constants.mu_s > 0.0f;
    if (canScatter && sampleCoef != 0.0f) dist = -thx_log(1.0f - u) / sampleCoef;
    return dist;
}
// ray.propagate.glsl:70-80
THX_DEV int updateRay(Ray& r, float dist, const thx_propagate_params& p) {
    r.log_contrib -= r.constants.mu_e * dist;
    r.time += dist / r.constants.vg;
    return r.time <= p.maxTime ? RESULT_CODE_SUCCESS : RESULT_CODE_RAY_DECAYED;
}
// ray.propagate.glsl:101-130
THX_DEV void updateRayIS(Ray& r, float dist, const thx_propagate_params& p, bool hit) {
    bool canScatter = r.constants.mu_s > 0.0f;
    if (!canScatter) return;
    float sampleCoef = p.scatterCoefficient >= 0.0f ? p.scatterCoefficient : r.constants.mu_s;
    r.log_contrib += sampleCoef * dist;
    if (!hit) r.lin_contrib /= sampleCoef;
}
// ray.propagate.glsl:153-166
THX_DEV int propagateRay(Ray& r, float dist, const thx_propagate_params& p) {
    r.position = r.position + r.direction * dist;
    f3 q = r.position;
    bool outside = q.x < p.lowerBBoxCorner[0] || q.y < p.lowerBBoxCorner[1] ||
                   q.z < p.lowerBBoxCorner[2] || q.x > p.upperBBoxCorner[0] ||
                   q.y > p.upperBBoxCorner[1] || q.z > p.upperBBoxCorner[2];
    int result = updateRay(r, dist, p);
    return outside ? RESULT_CODE_RAY_LOST : result;
}
// ray.propagate.glsl:193-202
template <class V>
THX_DEV void alignRayToHit(Ray& r, f3 hitNormal) {
    if constexpr (V::POL) {
        f3 polRef;
        float c, s;
        rotatePolRef(r.direction, r.polRef, hitNormal, polRef, c, s);
        r.stokes = rotApply(c, s, r.stokes);
        r.polRef = polRef;
    }
}
// ray.propagate.glsl:237-265
template <class V>
THX_DEV int propagateRayToHit(Ray& r, f3 hitPos, f3 hitNormal, const thx_propagate_params& p) {
    alignRayToHit<V>(r, hitNormal);
    float dist = distance(r.position, hitPos);
    r.position = hitPos;
    return updateRay(r, dist, p);
}

// ------------------------------------------------- light / wavelength source

struct WavelengthSample {
    float wavelength, contrib;
};
// lightsource.common.glsl:12-26
struct SourceRay {
    f3 position, direction;
    f4 stokes;
    f3 polRef;
    float startTime, contrib;
};

// wavelengthsource.const.glsl:10-12, wavelengthsource.uniform.glsl:12-16,
// wavelengthsource.function.glsl:13-16, wavelengthsource.host.glsl:10-13 (item L.local of the
// queue: the reference's idx is the position in the batch)
template <class V>
THX_DEV WavelengthSample sampleWavelength(Lane<V>& L) {
    const THX_KERNARG thx_run_params& P = coldArgs().p; // birth happens once per photon: nothing of it needs to stay in SGPRs
    const uint32_t flags = L.a->flags;
    if (flags & THX_FLAG_WAVELENGTH_HOST) {
        const THX_KERNARG LaunchArgs& cold = coldArgs();
        const float* q = cold.wavelength_queue + L.local;
        return WavelengthSample{q[0], q[cold.wavelength_capacity]};
    }
    if (flags & THX_FLAG_WAVELENGTH_FUNCTION) {
        const THX_KERNARG thx_wavelength_params& w = P.wavelength;
        const uint64_t table = (uint64_t)w.table[0] | ((uint64_t)w.table[1] << 32);
        return WavelengthSample{lookUp(table, draw(L), 0.0f, L.cnt), w.p2};
    }
    if (flags & THX_FLAG_WAVELENGTH_UNIFORM) {
        float u = draw(L);
        return WavelengthSample{mixf(P.wavelength.p0, P.wavelength.p1, u), P.wavelength.p2};
    }
    return WavelengthSample{P.wavelength.p0, 1.0f};
}

// lightsource.common.glsl:59-86 (unpolarized source in a polarized environment: Stokes (1,0,0,0),
// any reference perpendicular to the direction) and :88-112
template <class V>
THX_DEV SourceRay sourceRay(f3 position, f3 direction, float startTime, float contrib) {
    SourceRay r;
    r.position = position; r.direction = direction; r.startTime = startTime; r.contrib = contrib;
    r.stokes = f4{1.0f, 0.0f, 0.0f, 0.0f};
    r.polRef = f3{0.0f, 0.0f, 0.0f};
    if constexpr (V::POL) r.polRef = createLocalCOSY(direction).c0;
    return r;
}

THX_DEV float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// lightsource.host.glsl:12-19 (LOAD_SAMPLE of lightsource.queue.glsl), lightsource.cone.glsl:22-72,
// lightsource.pencil.glsl:18-35, lightsource.spherical.glsl:16-34. (The Cherenkov family -- dev_cherenkov.hpp -- is not compiled
// into the tracing kernels: the library samples those sources with the sampler kernel into the tracer's own queues right before the
// launch, which then reads them through the host-queue branch below and takes the photon's RNG dimension from the queue too.)
template <class V>
THX_DEV SourceRay sampleLight(Lane<V>& L) {
    const THX_KERNARG thx_run_params& P = coldArgs().p; // birth happens once per photon: nothing of it needs to stay in SGPRs
    const uint32_t flags = L.a->flags;
    if (flags & THX_FLAG_LIGHT_HOST) {
        const THX_KERNARG LaunchArgs& cold = coldArgs();
        const float* q = cold.light_queue + L.local;
        const size_t cap = cold.light_capacity;
        const f3 pos{q[0], q[cap], q[2 * cap]};
        const f3 dir{q[3 * cap], q[4 * cap], q[5 * cap]};
        if (flags & THX_FLAG_LIGHT_QUEUE_POLARIZED) {
            SourceRay r;
            r.position = pos; r.direction = dir; r.startTime = q[13 * cap]; r.contrib = q[14 * cap];
            r.stokes = f4{1.0f, 0.0f, 0.0f, 0.0f};
            r.polRef = f3{0.0f, 0.0f, 0.0f};
            if constexpr (V::POL) {
                r.stokes = f4{q[6 * cap], q[7 * cap], q[8 * cap], q[9 * cap]};
                r.polRef = f3{q[10 * cap], q[11 * cap], q[12 * cap]};
            }
            return r;
        }
        return sourceRay<V>(pos, dir, q[6 * cap], q[7 * cap]);
    }
    if (flags & THX_FLAG_LIGHT_CONE) {
        const THX_KERNARG thx_light_params& light = P.light;
        float u0, u1;
        draw2(L, u0, u1); // random2D
        float phi = TWO_PI * u0;
        float cos_theta = (1.0f - u1) + light.cosOpeningAngle * u1;
        float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
        f3 localDir{sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta};
        m3 trafo = createLocalCOSY(normalize(ld3(light.direction)));
        f3 rayDir = mul(trafo, localDir);
        float v = draw(L);
        SourceRay r = sourceRay<V>(ld3(light.position), rayDir, mixf(light.t_min, light.t_max, v), light.contribFwd);
        if constexpr (V::POL) {
            if (flags & THX_FLAG_LIGHT_POLARIZED) {
                f3 polRef = ld3(light.polRef);
                polRef = polRef - rayDir * dot(polRef, rayDir);
                r.polRef = normalize(polRef);
                r.stokes = f4{light.stokes[0], light.stokes[1], light.stokes[2], light.stokes[3]};
            }
        }
        return r;
    }
    if (flags & THX_FLAG_LIGHT_PENCIL) {
        float v = draw(L);
        SourceRay r;
        r.position = ld3(P.light.position);
        r.direction = ld3(P.light.direction);
        r.startTime = mixf(P.light.t_min, P.light.t_max, v);
        r.contrib = P.light.contribFwd;
        r.stokes = f4{1.0f, 0.0f, 0.0f, 0.0f};
        r.polRef = f3{0.0f, 0.0f, 0.0f};
        if constexpr (V::POL) {
            r.stokes = f4{P.light.stokes[0], P.light.stokes[1], P.light.stokes[2], P.light.stokes[3]};
            r.polRef = ld3(P.light.polRef);
        }
        return r;
    }
    float u0, u1;
    draw2(L, u0, u1); // random2D
    f3 rayDir = sphericalToCartessian(TWO_PI * u0, 2.0f * u1 - 1.0f);
    float v = draw(L);
    return sourceRay<V>(ld3(P.light.position), rayDir, mixf(P.light.t_min, P.light.t_max, v), P.light.contribFwd);
}

// tracer.*.forward.glsl sampleRay; ray.glsl:38-79, 188-211
template <class V>
THX_DEV Ray sampleRay(Lane<V>& L, uint64_t medium) {
    const WavelengthSample photon = sampleWavelength(L);
    MediumConstants constants = lookUpMedium(medium, photon.wavelength, L.cnt);
    const SourceRay s = sampleLight(L);
    if (L.a->flags & THX_FLAG_SOURCES_PRESAMPLED) { // the draws both samples took were made by the sampler kernel: go on behind them
        const THX_KERNARG LaunchArgs& cold = coldArgs();
        L.dim = __float_as_uint(cold.wavelength_queue[L.local + 2u * cold.wavelength_capacity]);
    }
    Ray r;
    r.position = s.position;
    r.direction = s.direction;
    r.medium = medium;
    r.wavelength = photon.wavelength;
    r.time = s.startTime;
    r.lin_contrib = photon.contrib * s.contrib;
    r.log_contrib = 0.0f;
    r.constants = constants;
    r.stokes = s.stokes;
    r.polRef = s.polRef;
    if (L.a->flags & THX_FLAG_PHOTON) r.lin_contrib = 1.0f; // tracer.scene.photon.init.glsl:49-53
    return r;
}

// tracer.scene.photon.{init,loop}.glsl:84-99 / :68-83: after a step the photon survives with
// probability contrib (one draw) and goes on with contribution 1; the Stokes vector is
// normalised first. Returns the step's result code.
template <class V>
THX_DEV int photonRoulette(Lane<V>& L, Ray& ray, int result) {
    if (result < 0) return result;
    if constexpr (V::POL) {
        ray.lin_contrib *= ray.stokes.x;
        const float s0 = ray.stokes.x;
        ray.stokes = f4{ray.stokes.x / s0, ray.stokes.y / s0, ray.stokes.z / s0, ray.stokes.w / s0};
    }
    if (ray.lin_contrib * thx_exp(ray.log_contrib) <= draw(L)) return RESULT_CODE_RAY_ABSORBED;
    ray.lin_contrib = 1.0f;
    ray.log_contrib = 0.0f;
    return result;
}
// RNG dimension at the start of step `step` (1-based): the reference starts run r of the
// photon tracer at dimension r * rng_per_run (trace.py:2655-2659, Push.dim)
template <class V>
THX_DEV void photonRunDim(Lane<V>& L, uint32_t step) {
    const uint32_t per = L.a->p.photon_steps_per_run;
    if (step > 1u && per != 0u && (step - 1u) % per == 0u) L.dim = ((step - 1u) / per) * L.a->p.photon_rng_per_run;
}

// ============================================================================
// VOLUME TRACER (tracer.volume.forward.glsl, target.sphere.glsl, math.sphere.glsl)
// ============================================================================

struct TargetSample {
    f3 position, normal;
    float dist, prob;
    bool valid;
    // the hit in the target's own frame (target.common.glsl:4-18); read by the HitRecorder response
    // and the probe only
    f3 objPosition, objNormal, offset;
    m3 worldToObj;
};
THX_DEV m3 diag3(float d) { return m3{f3{d, 0.0f, 0.0f}, f3{0.0f, d, 0.0f}, f3{0.0f, 0.0f, d}}; }
THX_DEV HitFrame targetFrame(const thx_run_params& p, const TargetSample& hit) {
    return HitFrame{hit.objPosition, hit.objNormal, hit.worldToObj, p.volume.objectId};
}
THX_DEV bool targetIsPlane(uint32_t flags) { return (flags & (THX_FLAG_TARGET_DISK | THX_FLAG_TARGET_FLAT)) != 0u; }

// math.sphere.glsl:14-39 (nearest root)
THX_DEV float intersectSphereNear(f3 center, float radius, f3 origin, f3 direction) {
    f3 f = origin - center;
    float b2 = dot(f, direction);
    float r2 = radius * radius;
    f3 fd = f - direction * b2;
    float discr = r2 - dot(fd, fd);
    float rx = INFINITY;
    if (discr >= 0.0f) {
        float cc = dot(f, f) - r2;
        float q = -b2 - signBit(b2) * sqrtf(discr);
        rx = cc / q;
        float ry = q;
        if (rx > ry) rx = ry;
    }
    return rx;
}

// math.sphere.glsl:14-39 (farthest root)
THX_DEV float intersectSphereFar(f3 center, float radius, f3 origin, f3 direction) {
    f3 f = origin - center;
    float b2 = dot(f, direction);
    float r2 = radius * radius;
    f3 fd = f - direction * b2;
    float discr = r2 - dot(fd, fd);
    float ry = INFINITY;
    if (discr >= 0.0f) {
        float cc = dot(f, f) - r2;
        float q = -b2 - signBit(b2) * sqrtf(discr);
        float rx = cc / q;
        ry = q;
        if (rx > ry) ry = rx;
    }
    return ry;
}

// util.sample.glsl:31-56: concentric mapping of the unit square onto the unit disk
THX_DEV f3 sampleUnitDisk(float u0, float u1) {
    float rx = 2.0f * u0 - 1.0f, ry = 2.0f * u1 - 1.0f;
    if (rx == 0.0f && ry == 0.0f) return f3{0.0f, 0.0f, 0.0f};
    float phi, r;
    if (fabsf(rx) > fabsf(ry)) {
        r = rx;
        phi = 0.78539816339744830962f * (ry / rx);
    } else {
        r = ry;
        phi = 1.57079632679489661923f - 0.78539816339744830962f * (rx / ry);
    }
    float sp, cp;
    thx_sincos(phi, sp, cp);
    return f3{r * cp, r * sp, 0.0f};
}
THX_DEV m3 ldm3(const float* m) { // column major
    return m3{f3{m[0], m[1], m[2]}, f3{m[3], m[4], m[5]}, f3{m[6], m[7], m[8]}};
}
THX_DEV m3 ldm3(const THX_KERNARG float* m) {
    return m3{f3{m[0], m[1], m[2]}, f3{m[3], m[4], m[5]}, f3{m[6], m[7], m[8]}};
}
// target.disk.glsl:43-66 / target.flat.glsl:47-72
THX_DEV TargetSample intersectTargetPlane(const thx_target_params& tp, f3 observer, f3 direction, bool disk, bool missInfinite) {
    const m3 o2w = ldm3(tp.objToWorld);
    const m3 w2o = transpose(o2w);
    const f3 center = ld3(tp.position);
    f3 localObs = mul(w2o, observer - center);
    f3 localDir = mul(w2o, direction);
    float t = -localObs.z / localDir.z;
    f3 localPos = localObs + localDir * t;
    bool valid = t > 0.0f;
    if (disk) valid = valid && sqrtf(localPos.x * localPos.x + localPos.y * localPos.y) <= tp.radius;
    else valid = valid && 2.0f * fabsf(localPos.x) <= tp.width && 2.0f * fabsf(localPos.y) <= tp.height;
    const float side = signBit(localObs.z);
    TargetSample s;
    s.position = mul(o2w, localPos) + center;
    s.normal = ld3(tp.normal) * side;
    // target.disk.glsl:60 / target.flat.glsl:66 report the distance to the PLANE also when the outline is
    // missed or the plane lies behind the ray, and the tracers take min(hit.dist, free path)
    // (tracer.volume.forward.glsl:167): every ray stops at the infinite plane, rays pointing away from
    // it walk backwards onto it. Kept as the default (results identical to the reference's);
    // THX_FLAG_TARGET_MISS_INFINITE opts into the repaired behaviour: a miss is infinitely far, like
    // createTargetSampleMiss() (oracle: same switch).
    s.dist = (valid || !missInfinite) ? t : INFINITY;
    s.prob = tp.hemisphereProb * (valid ? 1.0f : 0.0f);
    s.valid = valid;
    s.objPosition = localPos;
    s.objNormal = f3{0.0f, 0.0f, side};
    s.worldToObj = w2o;
    s.offset = mul(w2o, center) * -1.0f;
    return s;
}

// target.sphere.glsl:45-78 (objPosition = objNormal = normal, worldToObj = 1/r);
// inner: target.sphere.inner.glsl:36-57 (seen from inside: far root, normal towards the centre)
THX_DEV TargetSample intersectTarget(const thx_target_params& tp, f3 observer, f3 direction, uint32_t flags) {
    if (targetIsPlane(flags)) return intersectTargetPlane(tp, observer, direction, (flags & THX_FLAG_TARGET_DISK) != 0u, (flags & THX_FLAG_TARGET_MISS_INFINITE) != 0u);
    f3 center = ld3(tp.position);
    TargetSample s;
    s.offset = ld3(tp.invPos);
    s.worldToObj = diag3(tp.invRad);
    if (flags & THX_FLAG_TARGET_INNER) {
        float t = intersectSphereFar(center, tp.radius, observer, direction);
        f3 pos = observer + direction * t;
        f3 nrm = normalize(center - pos);
        s.position = pos; s.normal = nrm; s.dist = t; s.prob = tp.hemisphereProb;
        s.valid = (t > 0.0f) && !isinf(t);
        s.objPosition = nrm * -1.0f; s.objNormal = nrm;
        return s;
    }
    float t = intersectSphereNear(center, tp.radius, observer, direction);
    bool hit = (t > 0.0f) && !isinf(t);
    if (!hit) { // target.common.glsl:79-91
        s.position = f3{0, 0, 0};
        s.normal = f3{0, 0, 0};
        s.objPosition = f3{0, 0, 0}; s.objNormal = f3{0, 0, 0};
        s.offset = f3{0, 0, 0}; s.worldToObj = diag3(1.0f);
        s.dist = INFINITY;
        s.prob = 0.0f;
        s.valid = false;
        return s;
    }
    f3 pos = observer + direction * t;
    f3 nrm = normalize(pos - center);
    float d = distance(observer, center);
    float cosOpening = tp.radius / d;
    float prob = tp.hemisphereProb / (1.0f - cosOpening);
    bool valid = !isinf(prob);
    if (!valid) prob = 0.0f;
    s.position = pos; s.normal = nrm; s.dist = t; s.prob = prob; s.valid = valid;
    s.objPosition = nrm; s.objNormal = nrm;
    return s;
}

// target.sphere.glsl:18-43; inner: target.sphere.inner.glsl:18-34 (uniform over the sphere);
// target.disk.glsl:15-41 / target.flat.glsl:17-45
template <class V>
THX_DEV TargetSample sampleTarget(Lane<V>& L, const thx_target_params& tp, f3 observer, uint32_t flags) {
    f3 center = ld3(tp.position);
    TargetSample s;
    if (targetIsPlane(flags)) {
        float u0, u1;
        draw2(L, u0, u1); // random2D
        f3 localPos = (flags & THX_FLAG_TARGET_DISK) ? sampleUnitDisk(u0, u1) * tp.radius
                                                     : f3{tp.width * (u0 - 0.5f), tp.height * (u1 - 0.5f), 0.0f};
        const m3 o2w = ldm3(tp.objToWorld);
        f3 pos = mul(o2w, localPos) + center;
        f3 normal = ld3(tp.normal);
        const float side = signf(dot(normal, observer - pos));
        normal = normal * side;
        s.position = pos; s.normal = normal; s.dist = distance(observer, pos);
        s.prob = tp.hemisphereProb; s.valid = dot(normal, normal) != 0.0f;
        s.objPosition = localPos; s.objNormal = f3{0.0f, 0.0f, side};
        s.worldToObj = transpose(o2w);
        s.offset = mul(s.worldToObj, center) * -1.0f;
        return s;
    }
    s.offset = ld3(tp.invPos);
    s.worldToObj = diag3(tp.invRad);
    if (flags & THX_FLAG_TARGET_INNER) {
        float v0, v1;
        draw2(L, v0, v1); // random2D
        f3 normal = sphericalToCartessian(TWO_PI * v0, 2.0f * v1 - 1.0f); // util.sample.glsl:84-88
        f3 pos = normal * tp.radius + center;
        s.position = pos; s.normal = normal * -1.0f; s.dist = distance(observer, pos);
        s.prob = tp.hemisphereProb; s.valid = true;
        s.objPosition = normal; s.objNormal = normal * -1.0f;
        return s;
    }
    f3 dir = normalize(observer - center);
    float d = distance(observer, center);
    float cosOpening = tp.radius / d;
    float u0, u1;
    draw2(L, u0, u1); // random2D
    float phi = TWO_PI * u0; // util.sample.glsl:23-27
    float cos_theta = (1.0f - u1) + cosOpening * u1;
    f3 normal = mul(createLocalCOSY(dir), sphericalToCartessian(phi, cos_theta));
    f3 pos = normal * tp.radius + center;
    float prob = tp.hemisphereProb / (1.0f - cosOpening);
    bool valid = !isinf(prob);
    if (!valid) prob = 0.0f;
    s.position = pos; s.normal = normal; s.dist = distance(pos, observer); s.prob = prob;
    s.valid = valid;
    s.objPosition = normal; s.objNormal = normal;
    return s;
}

// isOccludedByTarget: target.sphere.glsl:80-83 (inside), target.sphere.inner.glsl:59-61 (outside),
// target.disk.glsl:68-71 / target.flat.glsl:74-77 (never)
THX_DEV bool isOccludedByTarget(const thx_target_params& tp, f3 observer, uint32_t flags) {
    if (targetIsPlane(flags)) return false;
    const float d = distance(observer, ld3(tp.position));
    return (flags & THX_FLAG_TARGET_INNER) ? (d >= tp.radius) : (d <= tp.radius);
}

// tracer.volume.forward.glsl:51-82 (ray by value)
template <class V>
THX_DEV void volCreateResponse(Lane<V>& L, Ray ray, const TargetSample& hit, f3 dir, float weight,
                               bool scattered, uint64_t medium) {
    if (!hit.valid) return;
    if (scattered) scatterRayIS(L, ray, medium, dir);
    int code = propagateRayToHit<V>(ray, hit.position, hit.normal, L.a->p.volume.propagation);
    if (code < 0) return;
    ray.lin_contrib *= weight;
    respond(L, ray, targetFrame(L.a->p, hit));
}

// tracer.volume.forward.glsl:103-114
THX_DEV float jacobian_dAdW(f3 obs, f3 pos, f3 nrm) {
    f3 dir = pos - obs;
    float d2 = dot(dir, dir);
    dir = normalize(dir);
    float factor = d2 / fabsf(dot(dir, nrm));
    if (isinf(factor)) factor = 0.0f;
    return factor;
}

// tracer.volume.forward.glsl:116-149
template <class V>
THX_DEV void volSampleTargetMIS(Lane<V>& L, const Ray& ray, uint64_t med) {
    const thx_target_params& tg = L.a->p.target;
    f3 obs = ray.position, dir = ray.direction;
    float pPP;
    float u0, u1;
    draw2(L, u0, u1); // random2D
    const ScatterTabs tabs = scatterTabs(med);
    f3 dirPhase = scatter(tabs, dir, u0, u1, pPP, L.cnt);
    TargetSample phaseHit = intersectTarget(tg, obs, dirPhase, L.a->flags);
    TargetSample targetHit = sampleTarget(L, tg, obs, L.a->flags);
    f3 dirTarget = normalize(targetHit.position - obs);
    float pTT = targetHit.prob * jacobian_dAdW(obs, targetHit.position, targetHit.normal);
    float pPT = scatterProb(tabs, dir, dirTarget, L.cnt);
    float pTP = phaseHit.prob * jacobian_dAdW(obs, phaseHit.position, phaseHit.normal);
    float wTarget = pTT * pPT / (pTT * pTT + pPT * pPT);
    float wPhase = pPP * pPP / (pPP * pPP + pTP * pTP);
    volCreateResponse(L, ray, phaseHit, dirPhase, wPhase, true, med);
    volCreateResponse(L, ray, targetHit, dirTarget, wTarget, true, med);
}

// tracer.volume.forward.glsl:153-208
template <class V>
THX_DEV int volTrace(Lane<V>& L, Ray& ray, uint64_t med, bool allowResponse) {
    const thx_propagate_params& prop = L.a->p.volume.propagation;
    L.cnt.add(ST_STEPS);
    float u = draw(L);
    float dist = sampleScatterLength(ray, prop, u);
    TargetSample hit = intersectTarget(L.a->p.target, ray.position, ray.direction, L.a->flags);
    bool hitValid = hit.valid && (hit.dist <= dist);
    dist = fminf(hit.dist, dist);
    int result = propagateRay(ray, dist, prop);
    updateRayIS(ray, dist, prop, hitValid);
    if (result < 0) return result;
    if (hitValid && allowResponse) {
        alignRayToHit<V>(ray, hit.normal);
        respond(L, ray, targetFrame(L.a->p, hit));
        return RESULT_CODE_RAY_DETECTED;
    } else if (hitValid) {
        return RESULT_CODE_RAY_ABSORBED;
    }
    if constexpr (V::MIS) volSampleTargetMIS(L, ray, med);
    return RESULT_CODE_RAY_SCATTERED;
}

// One lane-step of tracer.volume.forward.glsl:231-276. step 0 = birth + the first
// (special) trace, step k>0 = scatter + trace; the trace itself has one call
// site. Returns false when the photon is finished.
template <class V>
THX_DEV bool volStep(Lane<V>& L, Ray& ray, uint32_t& step) {
    const LaunchArgs& A = *L.a;
    const uint64_t med = (uint64_t)A.p.volume.medium[0] | ((uint64_t)A.p.volume.medium[1] << 32);
    const bool disable_direct = (A.flags & THX_FLAG_DISABLE_DIRECT) != 0;
    bool allow;
    if (step == 0) {
        ray = sampleRay(L, med);
        onEvent(L, ray, RESULT_CODE_RAY_CREATED, 0);
        // isOccludedByTarget: target.sphere.glsl:80-83 (inside), target.sphere.inner.glsl:59-61 (outside)
        if (isOccludedByTarget(A.p.target, ray.position, A.flags)) {
            onEvent(L, ray, ERROR_CODE_TRACE_ABORT, 0);
            return false;
        }
        if constexpr (V::MIS) {
            if (!disable_direct) {
                TargetSample directHit = intersectTarget(A.p.target, ray.position, ray.direction, A.flags);
                volCreateResponse(L, ray, directHit, ray.direction, 1.0f, false, med);
            }
        }
        allow = !V::MIS && !disable_direct; // DIRECT_LIGHTING, :211-215
    } else {
        float u0, u1;
        draw2(L, u0, u1); // random2D
        scatterRay(L, ray, med, u0, u1);
        allow = !V::MIS; // ALLOW_RESPONSE, :217-221
    }
    int result = volTrace(L, ray, med, allow);
    onEvent(L, ray, result, step + 1);
    if (result < 0) return false;
    if (step == A.path_length) {
        onEvent(L, ray, RESULT_CODE_MAX_ITER, A.path_length + 2);
        return false;
    }
    step += 1;
    return true;
}

// VolumePhotonTracer (THX_FLAG_PHOTON on the volume kind): tracer.volume.photon.common.glsl:37-78
// trace() and :80-121 traceLoop(), tracer.volume.photon.init.glsl:46-62. One lane-step = birth
// (step 0) + one loop iteration: trace, scatter for the next iteration (always, also after a
// stop code), survival roulette, event. The reference restarts the event index with every
// run; the running step index is used here (only a TrackRecordCallback could tell).
template <class V>
THX_DEV bool volPhotonStep(Lane<V>& L, Ray& ray, uint32_t& step) {
    const LaunchArgs& A = *L.a;
    const uint64_t med = (uint64_t)A.p.volume.medium[0] | ((uint64_t)A.p.volume.medium[1] << 32);
    const thx_propagate_params& prop = A.p.volume.propagation;
    if (step == 0) {
        ray = sampleRay(L, med); // contribution 1: survival chance
        onEvent(L, ray, RESULT_CODE_RAY_CREATED, 0);
        if (isOccludedByTarget(A.p.target, ray.position, A.flags)) {
            onEvent(L, ray, ERROR_CODE_TRACE_ABORT, 0);
            return false;
        }
        step = 1;
        if (A.path_length == 0) {
            onEvent(L, ray, RESULT_CODE_MAX_ITER, 1);
            return false;
        }
    }
    photonRunDim(L, step);
    L.cnt.add(ST_STEPS);
    // trace()
    float dist = sampleScatterLength(ray, prop, draw(L));
    TargetSample hit = intersectTarget(A.p.target, ray.position, ray.direction, A.flags);
    const bool hitValid = hit.valid && (hit.dist <= dist);
    dist = fminf(hit.dist, dist);
    int result = propagateRay(ray, dist, prop);
    updateRayIS(ray, dist, prop, hitValid);
    if (result >= 0) {
        if (hitValid) {
            alignRayToHit<V>(ray, hit.normal);
            respond(L, ray, targetFrame(A.p, hit));
            result = RESULT_CODE_RAY_ABSORBED; // so that the photon is not detected twice
        } else {
            result = RESULT_CODE_RAY_SCATTERED;
        }
    }
    // scatter for the next iteration
    float u0, u1;
    draw2(L, u0, u1); // random2D
    scatterRay(L, ray, med, u0, u1);
    result = photonRoulette(L, ray, result);
    onEvent(L, ray, result, step);
    if (result < 0) return false;
    if (step == A.path_length) {
        onEvent(L, ray, RESULT_CODE_MAX_ITER, A.path_length + 1);
        return false;
    }
    step += 1;
    return true;
}

} // namespace thx

#include "dev_walk.hpp" // the ray query: records, triangle test, Walker

namespace thx {

// scene.types.glsl:29-57 (what the path needs of it)
struct SurfaceHit {
    bool valid, inward;
    int customId;
    uint32_t flags;
    uint64_t mat_inside, mat_outside;
    f3 worldPos, rayNrm;
    f3 objPos, objNrm; // read by the HitRecorder response only
    m3 worldToObj;
};

// scene.intersect.glsl:23-99. `rayDir` is ray.state.direction (not normalised).
template <class V>
THX_DEV int processRayQuery(Lane<V>& L, f3 rayDir, uint64_t rayMedium, const QueryHit& q,
                            SurfaceHit& hit) {
    const SceneView& sc = L.a->scene;
    hit.valid = q.valid;
    hit.inward = false; hit.customId = 0; hit.flags = 0;
    hit.mat_inside = 0; hit.mat_outside = 0;
    hit.worldPos = f3{0, 0, 0}; hit.rayNrm = f3{0, 0, 0};
    hit.objPos = f3{0, 0, 0}; hit.objNrm = f3{0, 0, 0};
    hit.worldToObj = m3{f3{0, 0, 0}, f3{0, 0, 0}, f3{0, 0, 0}};
    if (!hit.valid) return RESULT_CODE_RAY_MISSED;
    // everything is addressed by the hit itself: the instance record, its Geometry (geometry k belongs to instance k),
    // and the triangle's vertex positions and normals in leaf order (the same floats the vertex buffer holds)
    const InstanceRecord* inst = sc.instances + q.instance;
    const GeometryGLSL geom = sc.geometries[q.instance];
    hit.customId = inst->customIndex;
    const char* tv = reinterpret_cast<const char*>(sc.tris + q.prim);
    const float4* tn = reinterpret_cast<const float4*>(sc.normals + q.prim);
    const f3 p0 = loadP3(tv), p1 = loadP3(tv + 20), p2 = loadP3(tv + 40);
    const float4 m0 = tn[0], m1 = tn[1], m2 = tn[2];
    f3 n0 = f3{m0.x, m0.y, m0.z}, n1v = f3{m1.x, m1.y, m1.z}, n2v = f3{m2.x, m2.y, m2.z};
    float bx = q.b1, by = q.b2;
    f3 e1 = p1 - p0, e2 = p2 - p0;
    f3 objPos = f3{p0.x + __fmaf_rn(bx, e1.x, by * e2.x), p0.y + __fmaf_rn(bx, e1.y, by * e2.y),
                   p0.z + __fmaf_rn(bx, e1.z, by * e2.z)};
    f3 objNrm = cross(e1, e2);
    f3 n1 = n1v - n0, n2 = n2v - n0;
    f3 intNrm = f3{n0.x + __fmaf_rn(bx, n1.x, by * n2.x), n0.y + __fmaf_rn(bx, n1.y, by * n2.y),
                   n0.z + __fmaf_rn(bx, n1.z, by * n2.z)};
    objNrm = objNrm * signBit(dot(objNrm, intNrm));
    objNrm = normalize(objNrm);

    const float* W = inst->w2o; // row-major 3x4; mat3(world2Obj) columns:
    m3 worldToObj{f3{W[0], W[4], W[8]}, f3{W[1], W[5], W[9]}, f3{W[2], W[6], W[10]}};
    f3 objDir = normalize(mul(worldToObj, rayDir));
    hit.inward = dot(objDir, objNrm) <= 0.0f;
    hit.objPos = objPos; hit.objNrm = objNrm; hit.worldToObj = worldToObj;

    uint32_t flagsIn = 0, flagsOut = 0;
    hit.mat_inside = 0; hit.mat_outside = 0;
    if (geom.material != 0) {
        const THX_GLOBAL MaterialGLSL* mat = reinterpret_cast<const THX_GLOBAL MaterialGLSL*>(geom.material);
        hit.mat_inside = mat->inside; hit.mat_outside = mat->outside;
        flagsIn = mat->flagsInwards; flagsOut = mat->flagsOutwards;
    }
    hit.flags = hit.inward ? flagsIn : flagsOut;
    uint64_t medium = hit.inward ? hit.mat_outside : hit.mat_inside;
    if (rayMedium != medium) return ERROR_CODE_MEDIA_MISMATCH;

    // vec3(objNrm * world2Obj): row vector times matrix (inverse-transpose normal)
    f3 wn = f3{dot(objNrm, worldToObj.c0), dot(objNrm, worldToObj.c1), dot(objNrm, worldToObj.c2)};
    wn = normalize(wn);
    hit.rayNrm = wn * (2.0f * (hit.inward ? 1.0f : 0.0f) - 1.0f);

    const float* M = inst->o2w;
    hit.worldPos.x = M[3] + __fmaf_rn(M[0], objPos.x, __fmaf_rn(M[1], objPos.y, M[2] * objPos.z));
    hit.worldPos.y = M[7] + __fmaf_rn(M[4], objPos.x, __fmaf_rn(M[5], objPos.y, M[6] * objPos.z));
    hit.worldPos.z = M[11] + __fmaf_rn(M[8], objPos.x, __fmaf_rn(M[9], objPos.y, M[10] * objPos.z));
    return RESULT_CODE_SUCCESS;
}

struct Reflectance {
    float n_in, n_tr, r_s, r_p;
};

// scatter.surface.glsl:30-50: the arithmetic of fresnelReflect for given refractive indices
// (cos_i = signed cosine between ray direction and surface normal)
THX_DEV Reflectance fresnelCoefficients(float cos_i, float n_i, float n_t) {
    cos_i = clampf(cos_i, -1.0f, 1.0f);
    float sin_i = sqrtf(fmaxf(1.0f - cos_i * cos_i, 0.0f));
    float sin_t = sin_i * n_i / n_t;
    float cos_t = sqrtf(fmaxf(1.0f - sin_t * sin_t, 0.0f));
    cos_i = fabsf(cos_i);
    Reflectance r;
    r.n_in = n_i; r.n_tr = n_t;
    r.r_s = (n_i * cos_i - n_t * cos_t) / (n_i * cos_i + n_t * cos_t);
    r.r_p = (n_t * cos_i - n_i * cos_t) / (n_t * cos_i + n_i * cos_t);
    return r;
}
// scatter.surface.glsl:21-51
template <class V>
THX_DEV Reflectance fresnelReflect(Lane<V>& L, const Ray& ray, const SurfaceHit& hit) {
    uint64_t otherMed = hit.inward ? hit.mat_inside : hit.mat_outside;
    float n_t = 1.0f;
    if (otherMed != 0) {
        const THX_GLOBAL MediumGLSL* m = reinterpret_cast<const THX_GLOBAL MediumGLSL*>(otherMed);
        float u_t = normalize_lambda(m, ray.wavelength);
        n_t = lookUp(m->n, u_t, 1.0f, L.cnt);
    }
    return fresnelCoefficients(dot(ray.direction, hit.rayNrm), ray.constants.n, n_t);
}

// ray.surface.glsl:22-36 (integer-ULP nudge against self-intersection)
THX_DEV f3 offsetRay(f3 p, f3 n) {
    int ox = (int)(256.0f * n.x), oy = (int)(256.0f * n.y), oz = (int)(256.0f * n.z);
    f3 pi = f3{__int_as_float(__float_as_int(p.x) + ((p.x < 0.0f) ? -ox : ox)),
               __int_as_float(__float_as_int(p.y) + ((p.y < 0.0f) ? -oy : oy)),
               __int_as_float(__float_as_int(p.z) + ((p.z < 0.0f) ? -oz : oz))};
    return f3{fabsf(p.x) < (1.0f / 32.0f) ? p.x + (1.0f / 65536.0f) * n.x : pi.x,
              fabsf(p.y) < (1.0f / 32.0f) ? p.y + (1.0f / 65536.0f) * n.y : pi.y,
              fabsf(p.z) < (1.0f / 32.0f) ? p.z + (1.0f / 65536.0f) * n.z : pi.z};
}

// ray.surface.glsl:44-55
template <class V>
THX_DEV void crossBorder(Lane<V>& L, Ray& ray, const SurfaceHit& hit) {
    ray.position = offsetRay(hit.worldPos, -hit.rayNrm);
    uint64_t medium = hit.inward ? hit.mat_inside : hit.mat_outside;
    ray.medium = medium;
    ray.constants = lookUpMedium(medium, ray.wavelength, L.cnt);
}

THX_DEV f3 glslReflect(f3 I, f3 N) { return I - N * (2.0f * dot(N, I)); }
THX_DEV f3 glslRefract(f3 I, f3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return f3{0.0f, 0.0f, 0.0f};
    return I * eta - N * (eta * d + sqrtf(k));
}

// ray.surface.glsl:65-92, :147-166. IS = reflectance not applied.
template <class V, bool IS>
THX_DEV void reflectRay(Ray& ray, const SurfaceHit& hit, const Reflectance& refl) {
    ray.position = offsetRay(hit.worldPos, hit.rayNrm);
    ray.direction = normalize(glslReflect(ray.direction, hit.rayNrm));
    if constexpr (!IS) {
        float s = refl.r_s, p = refl.r_p;
        ray.lin_contrib *= 0.5f * (s * s + p * p);
    }
    if constexpr (V::POL) ray.stokes = polarizerApply(refl.r_p, refl.r_s, ray.stokes);
}
// ray.surface.glsl:102-132, :193-218
template <class V, bool IS>
THX_DEV void transmitRay(Lane<V>& L, Ray& ray, const SurfaceHit& hit, const Reflectance& refl) {
    crossBorder(L, ray, hit);
    float eta = refl.n_in / refl.n_tr;
    ray.direction = normalize(glslRefract(ray.direction, hit.rayNrm, eta));
    if constexpr (!IS) {
        float s = refl.r_s, p = refl.r_p;
        ray.lin_contrib *= 1.0f - 0.5f * (s * s + p * p);
    }
    if constexpr (V::POL) {
        float t_s = refl.r_s + 1.0f;
        float t_p = (refl.r_p + 1.0f) * eta;
        ray.stokes = polarizerApply(t_p, t_s, ray.stokes);
    }
}

// ------------------------------------------------- target_guide.sphere.glsl

struct GuideSample {
    f3 dir;
    float dist, prob;
};

// target_guide.sphere.glsl:17-31 == :55-69
THX_DEV void guideCone(const thx_guide_params& g, f3 observer, float& d, f3& viewDir,
                       float& cosMin, float& prob) {
    f3 center = ld3(g.position);
    d = distance(center, observer);
    viewDir = normalize(center - observer);
    float sinMax = g.radius / d;
    float sin2Max = sinMax * sinMax;
    cosMin = 1.0f - sqrtf(fmaxf(1.0f - sin2Max, 0.0f));
    if (sin2Max < 0.00068523f) cosMin = 0.5f * sin2Max;
    prob = 1.0f / (TWO_PI * cosMin);
    prob *= (d > g.radius) ? 1.0f : 0.0f;
}
// target_guide.common.glsl:10-32
THX_DEV GuideSample guideSampleFromPoint(f3 observer, f3 targetPos, f3 targetNrm, float targetProb) {
    f3 dir = targetPos - observer;
    const float d2 = dot(dir, dir);
    dir = normalize(dir);
    const float cosNormal = dot(dir, targetNrm);
    float prob = targetProb * d2 / fabsf(cosNormal);
    if (isinf(prob)) prob = 0.0f;
    prob *= (cosNormal < 0.0f) ? 1.0f : 0.0f;
    return GuideSample{dir, distance(targetPos, observer), prob};
}
// target_guide.common.glsl:34-51
THX_DEV GuideSample guideSampleFromDir(f3 dir, float dist, f3 targetNrm, float targetProb) {
    const float cosNormal = dot(dir, targetNrm);
    float prob = targetProb * dist * dist / fabsf(cosNormal);
    if (isinf(prob)) prob = 0.0f;
    prob *= (cosNormal < 0.0f) ? 1.0f : 0.0f;
    return GuideSample{dir, dist, prob};
}
THX_DEV bool guideIsPlane(uint32_t flags) { return (flags & (THX_FLAG_GUIDE_DISK | THX_FLAG_GUIDE_FLAT)) != 0u; }
// target_guide.disk.glsl:25-44 / target_guide.flat.glsl:31-48
THX_DEV GuideSample evalGuidePlane(f3 observer, f3 direction, bool disk) {
    const THX_KERNARG thx_guide_params& g = coldArgs().p.guide;
    const m3 w2o = transpose(ldm3(g.objToWorld));
    f3 localObs = mul(w2o, observer - ld3(g.position));
    f3 localDir = mul(w2o, direction);
    float t = -localObs.z / localDir.z; // +/- inf or NaN if localDir.z = 0
    f3 localHit = localObs + localDir * t;
    bool valid = t > 0.0f && !isinf(t) && !isnan(t);
    if (disk) valid = valid && sqrtf(localHit.x * localHit.x + localHit.y * localHit.y) <= g.radius;
    else valid = valid && 2.0f * fabsf(localHit.x) <= g.width && 2.0f * fabsf(localHit.y) <= g.height;
    return guideSampleFromDir(direction, t, ld3(g.normal), g.prob * (valid ? 1.0f : 0.0f));
}

// target_guide.sphere.glsl:15-51; target_guide.disk.glsl:15-23 / target_guide.flat.glsl:17-29
template <class V>
THX_DEV GuideSample sampleTargetGuide(Lane<V>& L, f3 observer) {
    if (guideIsPlane(L.a->flags)) {
        const THX_KERNARG thx_guide_params& g = coldArgs().p.guide;
        float u0, u1;
        draw2(L, u0, u1); // random2D
        f3 localPos = (L.a->flags & THX_FLAG_GUIDE_DISK) ? sampleUnitDisk(u0, u1) * g.radius
                                                         : f3{g.width * (u0 - 0.5f), g.height * (u1 - 0.5f), 0.0f};
        f3 pos = mul(ldm3(g.objToWorld), localPos) + ld3(g.position);
        return guideSampleFromPoint(observer, pos, ld3(g.normal), g.prob);
    }
    const thx_guide_params& g = L.a->p.guide;
    float d, cosMin, prob;
    f3 viewDir;
    guideCone(g, observer, d, viewDir, cosMin, prob);
    float cosTheta = 1.0f - cosMin * draw(L);
    float sinTheta = sqrtf(fmaxf(1.0f - cosTheta * cosTheta, 0.0f));
    float phi = TWO_PI * draw(L);
    float sp, cp;
    thx_sincos(phi, sp, cp);
    f3 local = f3{sinTheta * sp, sinTheta * cp, cosTheta};
    GuideSample s;
    s.dir = mul(createLocalCOSY(viewDir), local);
    s.dist = d + g.radius;
    s.prob = prob;
    return s;
}
// target_guide.sphere.glsl:53-81
THX_DEV GuideSample evalTargetGuide(const thx_guide_params& g, f3 observer, f3 direction, uint32_t flags) {
    if (guideIsPlane(flags)) return evalGuidePlane(observer, direction, (flags & THX_FLAG_GUIDE_DISK) != 0u);
    float d, cosMin, prob;
    f3 viewDir;
    guideCone(g, observer, d, viewDir, cosMin, prob);
    float cosDir = dot(viewDir, direction);
    prob *= (cosMin >= 1.0f - cosDir) ? 1.0f : 0.0f;
    return GuideSample{direction, d + g.radius, prob};
}

// ----------------------------------------------------- scene.traverse.glsl

// scene.traverse.glsl:33-68 (ray by value)
template <class V>
THX_DEV void sceneCreateResponse(Lane<V>& L, Ray ray, const SurfaceHit& hit,
                                 const Reflectance& surface, bool absorb) {
    if (!absorb) {
        // the transmission factor applies either way; HIT_USE_TRANSMITTED_DIR selects the
        // direction a recorded hit carries (scene.traverse.glsl:42-56)
        const f3 dir = ray.direction;
        transmitRay<V, false>(L, ray, hit, surface);
        if (!(L.a->flags & THX_FLAG_HIT_USE_TRANSMITTED)) ray.direction = dir;
    }
    respond(L, ray, HitFrame{hit.objPos, hit.objNrm, hit.worldToObj, hit.customId});
}

// scene.traverse.glsl:160-180 (ray by value; `hit` already checked for validity)
template <class V>
THX_DEV void processShadowRay(Lane<V>& L, Ray ray, const SurfaceHit& hit) {
    const int targetId = L.a->p.scene.targetId;
    bool isTarget = (hit.flags & MATERIAL_DETECTOR_BIT) != 0;
    bool correctId = targetId < 0 || targetId == hit.customId;
    if (!hit.valid || !correctId || !isTarget) return;
    if (propagateRayToHit<V>(ray, hit.worldPos, hit.rayNrm, L.a->p.scene.propagation) < 0) return;
    Reflectance surface = fresnelReflect(L, ray, hit);
    bool black = (hit.flags & MATERIAL_BLACK_BODY_BIT) != 0;
    sceneCreateResponse(L, ray, hit, surface, black);
}

// scene.traverse.glsl:73-154
template <class V>
THX_DEV int processHit(Lane<V>& L, Ray& ray, const SurfaceHit& hit, bool allowResponse) {
    const LaunchArgs& A = *L.a;
    const int targetId = A.p.scene.targetId;
    int result = propagateRayToHit<V>(ray, hit.worldPos, hit.rayNrm, A.p.scene.propagation);
    if (result < 0) return result;
    result = RESULT_CODE_RAY_HIT;
    bool isAbs = (hit.flags & MATERIAL_BLACK_BODY_BIT) != 0;
    bool isTarget = (hit.flags & MATERIAL_DETECTOR_BIT) != 0;
    bool volBorder = (hit.flags & MATERIAL_VOLUME_BORDER_BIT) != 0;
    bool canTransmit = (hit.flags & MATERIAL_NO_TRANSMIT_BWD_BIT) == 0;
    bool canReflect = (hit.flags & MATERIAL_NO_REFLECT_BWD_BIT) == 0;
    Reflectance surface = fresnelReflect(L, ray, hit);
    if (allowResponse && isTarget && (targetId < 0 || hit.customId == targetId)) {
        sceneCreateResponse(L, ray, hit, surface, isAbs);
        result = RESULT_CODE_RAY_DETECTED;
    }
    if (isAbs) return RESULT_CODE_RAY_ABSORBED;
    if (!(A.flags & THX_FLAG_DISABLE_VOLUME_BORDER)) {
        if (volBorder) {
            crossBorder(L, ray, hit);
            return RESULT_CODE_VOLUME_HIT;
        }
    }
    if (!(A.flags & THX_FLAG_DISABLE_TRANSMISSION)) {
        float r = 0.5f * (surface.r_s * surface.r_s + surface.r_p * surface.r_p);
        if (canReflect && canTransmit) {
            if (draw(L) < r) reflectRay<V, true>(ray, hit, surface);
            else transmitRay<V, true>(L, ray, hit, surface);
        } else if (canReflect) {
            reflectRay<V, false>(ray, hit, surface);
        } else if (canTransmit) {
            transmitRay<V, false>(L, ray, hit, surface);
        } else {
            result = RESULT_CODE_RAY_ABSORBED;
        }
    } else {
        if (canReflect) reflectRay<V, false>(ray, hit, surface);
        else result = RESULT_CODE_RAY_ABSORBED;
    }
    return result;
}

// One lane-step of tracer.scene.forward.glsl:58-106 = trace()
// (scene.traverse.glsl:280-352) + processInteraction() (:361-397).
//
// The up-to-three closest-hit queries of a step (primary, MIS phase shadow ray,
// MIS target shadow ray; :312-323, :185-214) run through ONE walk loop so that
// the traversal code exists once and all lanes of the wave traverse together.
// RNG draw order is the reference's: free path, [reflect/transmit], MIS phase
// (2), MIS guide (2), scatter (2)  (SURVEY.md 3.4).
template <class V>
THX_DEV bool sceneStep(Lane<V>& L, Ray& ray, uint32_t& step, bool& allowResponse) {
    const LaunchArgs& A = *L.a;
    const thx_propagate_params& prop = A.p.scene.propagation;
    if (step == 0) {
        const uint64_t medium =
            (uint64_t)A.p.scene.sceneMedium[0] | ((uint64_t)A.p.scene.sceneMedium[1] << 32);
        ray = sampleRay(L, medium);
        onEvent(L, ray, RESULT_CODE_RAY_CREATED, 0);
        allowResponse = (A.flags & THX_FLAG_DISABLE_DIRECT) == 0;
        step = 1;
        if (A.path_length == 0) {
            onEvent(L, ray, RESULT_CODE_MAX_ITER, 1);
            return false;
        }
    }
    const bool photon = (A.flags & THX_FLAG_PHOTON) != 0;
    const bool last = step == A.path_length;
    const bool lastNoScatter = last && !photon; // the photon tracer always scatters (processInteraction(..., last = false))
    if (photon) photonRunDim(L, step);
    L.cnt.add(ST_STEPS);

    int result = RESULT_CODE_SUCCESS;
    bool pending = true; // step still needs its primary query
    f3 dir = f3{0, 0, 1};
    float dist = 0.0f, sampledDist = 0.0f;
    bool mis_target = false;
    if (isRayBad(ray)) {
        result = ERROR_CODE_RAY_BAD;
        pending = false;
    } else {
        dir = normalize(ray.direction);
        dist = sampleScatterLength(ray, prop, draw(L));
        sampledDist = dist;
        if constexpr (V::MIS) {
            GuideSample gs = evalTargetGuide(A.p.guide, ray.position, dir, A.flags);
            mis_target = allowResponse && gs.prob > 0.0f;
            if (mis_target) {
                mis_target = gs.dist > dist;
                dist = fmaxf(gs.dist, dist);
            }
        }
    }

    // MIS shadow-ray set-up, filled in after the primary query missed
    bool doShadow = false;
    f3 dirPhase = f3{0, 0, 1}, dirTarget = f3{0, 0, 1};
    float distPhase = 0.0f, distTarget = 0.0f, wPhase = 0.0f, wTarget = 0.0f;
    bool scatterAfter = false;

    constexpr int NQ = V::MIS ? 3 : 1;
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
        const bool need = q == 0 ? pending : doShadow;
        f3 qd = q == 0 ? dir : (q == 1 ? dirPhase : dirTarget);
        float qmax = q == 0 ? dist : (q == 1 ? distPhase : distTarget);
        QueryHit qh;
        qh.valid = false;
        if (need) qh = rayQuery(L, ray.position, qd, qmax);
        if (!need) continue;
        if (qh.aborted) { // traversal stack overflow: the photon ends with an error, a shadow ray contributes nothing
            if (q == 0) result = ERROR_CODE_TRACE_ABORT;
            else L.cnt.add(ST_ERROR);
            continue;
        }

        // the ray whose state the hit is evaluated with
        Ray s = ray;
        if (q > 0) scatterRayIS(L, s, ray.medium, qd); // scene.traverse.glsl:195
        SurfaceHit hit;
        int queryResult = processRayQuery(L, s.direction, s.medium, qh, hit);

        if (q > 0) { // traceShadowRay, scene.traverse.glsl:207-213
            if (queryResult < 0) continue;
            s.lin_contrib *= (q == 1 ? wPhase : wTarget);
            processShadowRay(L, s, hit);
            continue;
        }

        // ---- rest of trace(), scene.traverse.glsl:329-351
        if (hit.valid) dist = distance(ray.position, hit.worldPos);
        if constexpr (V::MIS) {
            if (mis_target && hit.valid && dist > sampledDist && queryResult >= 0) {
                processShadowRay(L, ray, hit);
                hit.valid = false;
                dist = sampledDist;
            }
        }
        int propResult = propagateRay(ray, dist, prop);
        updateRayIS(ray, dist, prop, hit.valid);
        result = queryResult <= ERROR_CODE_MAX_VALUE ? queryResult : propResult;
        if (result < 0) continue;

        // ---- processInteraction(), scene.traverse.glsl:361-397
        if (hit.valid) {
            result = processHit(L, ray, hit, allowResponse);
        } else {
            if (!lastNoScatter) {
                if constexpr (V::MIS) { // sampleTargetMIS, :234-269
                    float pPP;
                    float u0, u1;
                    draw2(L, u0, u1); // random2D
                    const ScatterTabs tabs = scatterTabs(ray.medium);
                    dirPhase = scatter(tabs, ray.direction, u0, u1, pPP, L.cnt);
                    GuideSample ts = sampleTargetGuide(L, ray.position);
                    float pTT = ts.prob;
                    GuideSample ps = evalTargetGuide(A.p.guide, ray.position, dirPhase, A.flags);
                    float pTP = ps.prob;
                    float pPT = scatterProb(tabs, ray.direction, ts.dir, L.cnt);
                    wTarget = pTT * pPT / (pTT * pTT + pPT * pPT);
                    wPhase = pPP * pPP / (pPP * pPP + pTP * pTP);
                    dirTarget = ts.dir;
                    distPhase = ps.dist;
                    distTarget = ts.dist;
                    doShadow = true;
                }
                scatterAfter = true;
            }
            result = RESULT_CODE_RAY_SCATTERED;
        }
    }
    if (scatterAfter) { // scene.traverse.glsl:391
        float u0, u1;
        draw2(L, u0, u1); // random2D
        scatterRay(L, ray, ray.medium, u0, u1);
    }
    if (photon) result = photonRoulette(L, ray, result);

    onEvent(L, ray, result, step);
    if (result < 0) return false;
    if constexpr (V::MIS) allowResponse = result != RESULT_CODE_RAY_SCATTERED;
    else allowResponse = true;
    if (last) {
        onEvent(L, ray, RESULT_CODE_MAX_ITER, A.path_length + 1);
        return false;
    }
    step += 1;
    return true;
}

} // namespace thx
