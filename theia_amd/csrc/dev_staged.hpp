// Staged scene tracer: the same per-photon algorithm as sceneStep() (dev_trace.hpp),
// re-organised so that the lanes of a wavefront always run the SAME piece of it.
//
// Why: one-lane-one-photon execution of tracer.scene.forward.glsl is bound by VALU
// issue at ~23 % lane utilisation on MI355X (rocprofv3: SQ_THREAD_CYCLES_VALU /
// (64 * SQ_ACTIVE_INST_VALU)): per step only ~12 % of the photons need a deep BVH
// walk for the primary ray, ~45 % need MIS sampling plus shadow rays, ~50 % are
// being born, and a wavefront pays for every path any of its lanes takes.
//
// How: a workgroup owns a pool of 1100-1472 photons (structure of float4 arrays: 64 B per
// photon in LDS, 48 / 80 B in an L2-resident global array, see StagedLayout) and a queue per stage
//     BIRTH -> A (free path, guide) -> T0 (primary BVH walk + surface hit)
//           -> BM (propagate, no hit) -> M (MIS sampling)
//           -> TS (shadow-ray walk + detector response) -> C (scatter) -> A ...
// Every wavefront schedules itself: it claims up to 64 entries of the fullest ring
// (one CAS on the ring head by lane 0), runs that stage on them, one photon per
// lane, appends the photons to their next rings and looks for work again. There is
// no workgroup barrier inside the loop, so BVH walks (gather/latency heavy) and the
// shading stages (VALU heavy) of different wavefronts overlap on a SIMD and nobody
// waits for the slowest wavefront of a round. Photons are independent and carry
// their own RNG stream (= photon index) and dimension, so the order in which they
// advance does not change any per-photon result; arithmetic, draw order and event
// order are those of the lane kernel (the stage bodies call the same device
// functions). Rings hold 16-bit entries in LDS (as many cells as the pool has photons,
// rounded up to 64) with a cell hand-off (EMPTY <-> entry with a lap tag) between the
// appending and the claiming wavefront; an append is one LDS atomic per lane on the ring
// tail. DESIGN.md section 3.1 has the measurements behind every choice made here.
#pragma once
#include "dev_trace.hpp"

namespace thx {

constexpr uint32_t kSchedThreads = 1024; // upper bound of the workgroup size
// ... of the instrumented variants (statistics, track record, hit queue, Sobol sampler): 12 wavefronts leave every lane 168
// registers instead of 128. At 128 they kept 25-30 values in scratch memory, and three times in round 3 a harmless change elsewhere
// made ONE of them misbehave (photons traced differently, a tripped watchdog, a wild address) while the production variants, which
// never spill, and the other instrumented ones stayed exact (profiles/r03_inner_loop.txt items 10, 16, 22): no spills any more.
#ifndef THX_DEBUG_THREADS
#define THX_DEBUG_THREADS 768
#endif
constexpr uint32_t kSchedThreadsDebug = THX_DEBUG_THREADS;
#ifndef THX_POOL_MAX
#define THX_POOL_MAX 2048 // upper bound of the photon pool of a workgroup (ring entries: 11-bit slot id, bit 11 = a walk taken up again)
#endif
constexpr uint32_t kHitMiss = 0xFFFFFFFFu;
// Walk batches: the last rays of a batch are set aside (Walker::save) and go back into their ring once fewer than
// LaunchArgs::park_below lanes are still walking after at least park_rounds rounds; a later batch takes them up again
// together with fresh rays. A batch lasts as long as its longest ray: without this a wavefront spends the second half
// of most batches on a handful of lanes.
constexpr uint32_t E_RESUME = 1u << 11; // ring entry of a walk that was set aside

// stage ids == queue ids. The BIRTH queue is the free list of pool slots.
enum : int { S_BIRTH = 0, S_A = 1, S_T0 = 2, S_BM = 3, S_M = 4, S_TS = 5, S_C = 6, S_NUM = 7 };
// counters (LDS words), laid out so that ONE 8-byte read per lane fetches everything the
// scheduler looks at: pair q < 7 = {head, tail} of ring q (claimed up to / appended up to,
// monotonic), pair 7 = {photon cursor, abort flag}
enum : int { C_CURSOR = 14, C_ABORT = 15,
#ifdef THX_STAGE_PROFILE
             C_PROF = 32, C_NUM = 32 + 96 // profile builds: 48 64-bit accumulators behind the counters
#else
             C_NUM = 32
#endif
};
THX_DEV constexpr int cHead(int q) { return 2 * q; }
THX_DEV constexpr int cTail(int q) { return 2 * q + 1; }
constexpr uint32_t Q_EMPTY = 0xFFFFu;  // ring cell without an entry (entries: slot < 4096 | which << 12 | lap tag << 14)
// what the abort flag holds (which of the watchdog's checks tripped; reported in bits 8.. of the device error word)
enum : uint32_t { W_IDLE = 1u, W_APPEND = 2u, W_CLAIM = 4u, W_RING_LIMIT = 8u, W_TEST = 16u };
constexpr uint32_t kIdleSpinLimit = 1u << 20; // polls (~150 cycles each): ~70 ms without any work appearing -> abort
constexpr uint32_t kCellSpinLimit = 1u << 22;
#ifndef THX_IDLE_SLEEP
#define THX_IDLE_SLEEP 4 // x64 cycles between two looks of a wavefront that found no work (VALU is ~97 % busy: polling costs)
#endif

// Wavefront priorities (s_setprio; the SIMD's arbiter serves the higher one first): a walking wavefront is bound by its own
// chain of dependent instructions and issues one every ~10 cycles, a shading stage is a dense stream of vector instructions
// that takes every slot it gets -- so walks go first and the shading stages fill the gaps; the scheduling code between two
// stages is short and other wavefronts wait for what it publishes. Measured (profiles/r03_inner_loop.txt): walk 1 / scheduler 3
// over shading 0: config 2 / 3 / 5 -1.8 / -2.7 / -0.9 %; shading above the walks: +2 %. -DTHX_NO_PRIO builds without.
#ifndef THX_NO_PRIO
#ifndef THX_SCHED_PRIO
#define THX_SCHED_PRIO 3
#endif
#ifndef THX_WALK_PRIO
#define THX_WALK_PRIO 1
#endif
#endif

// photon flags (meta.z)
constexpr uint32_t F_ALIVE = 1u, F_ALLOW = 2u, F_MIS_TARGET = 4u;
constexpr uint32_t F_STEP_SHIFT = 8, F_STEP_MASK = 0xFFFFu << 8; // step index
constexpr uint32_t F_PEND_SHIFT = 28, F_PEND_ONE = 1u << 28;     // outstanding shadow walks

template <bool POL>
struct StagedLayout {
    // A photon's state: three HOT chunks (float4) that the scheduler-side stages and the walk read -- position + time,
    // direction + wavelength, meta (index, RNG dimension, flags incl. the pending-walk counter, shadow-ray length) --
    // and COLD chunks only the shading stages touch: contributions + medium, medium constants[, Stokes vector,
    // polarization reference]. Hot chunks and the scratch live in LDS, the cold ones in global memory (one array per
    // workgroup, 32 / 64 B per photon: L2 resident): 112 / 144 -> 80 B of LDS per photon (round 2). Photons in flight per CU are
    // what the throughput follows: pool 800 -> 1232 on config 3, 704 -> 1024 (and 12 -> 16 wavefronts) on
    // config 5, 1024 -> 1264 on config 2; measured -12 % / -15 % / -3 % kernel time.
    // Round 3: of the two scratch chunks (sampled distances of a primary ray / direction + weight of the two MIS shadow
    // rays) the second shadow ray's went the same way: the MIS stage writes it, the one walk batch that takes the ray up
    // reads it once before it starts. 64 B of LDS per photon; what that frees holds the walk's pair records on small
    // scenes (Variant::LDSN) and a larger pool on the others (profiles/r03_lds_nodes.txt).
    static constexpr uint32_t NHOT = 3u;
    static constexpr uint32_t NCOLD = POL ? 5u : 3u;                    // the last one: direction + weight of the second shadow ray
    static constexpr uint32_t NSC = 1u;                                 // scratch chunks per photon in LDS
    static constexpr uint32_t C_SHADOW2 = NCOLD - 1u;
    static constexpr uint32_t NCH = NHOT; // state chunks per photon in LDS
    // Ring capacity of a queue (the TS ring takes two of them): at least the pool. Round 3: the next multiple of 64 instead of
    // the next power of two -- a pool of 1184 photons paid for rings of 2048 cells (32 KB of LDS, 28 B per photon); with
    // 1216-cell rings the same LDS holds ~190 photons more. A ring position then turns into (lap, cell) by a multiplication
    // with a reciprocal (Pool::cell), exact while a ring sees fewer than 2^25 appends per launch; the host keeps the power of
    // two for launches that could get there (`exact` = false).
    static constexpr uint32_t qcap(uint32_t pool, bool exact) {
        if (exact) return (pool + 63u) & ~63u;
        uint32_t c = 64;
        while (c < pool) c <<= 1;
        return c;
    }
    // dynamic LDS: [hist][state+scratch (NCH+NSC)*P float4][stack depth*T]
    //              [rings: 6 x qcap u16, TS ring 2 qcap u16][counters]
    //              [Variant::LDSN: copy of the scene's pair records]
    static size_t bytes(uint32_t n_bins, uint32_t stackBytesPerThread, uint32_t pool, uint32_t threads, size_t nodeBytes = 0, bool exact = false) {
        return nodeOffset(n_bins, stackBytesPerThread, pool, threads, exact) + nodeBytes;
    }
    // byte offset of the node copy = everything in front of it (a multiple of 16: pools are multiples of 16 photons)
    static size_t nodeOffset(uint32_t n_bins, uint32_t stackBytesPerThread, uint32_t pool, uint32_t threads, bool exact = false) {
        size_t hist = ((size_t)n_bins * 4 + 15) & ~(size_t)15;
        return hist + (size_t)(NCH + NSC) * pool * 16 + (size_t)stackBytesPerThread * threads + (size_t)8 * qcap(pool, exact) * 2 + C_NUM * 4;
    }
    // largest pool (multiple of 16, <= THX_POOL_MAX) that fits `budget` bytes of LDS
    static uint32_t poolFor(uint32_t n_bins, uint32_t stackBytesPerThread, size_t budget, uint32_t threads, size_t nodeBytes = 0, bool exact = false) {
        uint32_t p = THX_POOL_MAX;
        while (p > 64 && bytes(n_bins, stackBytesPerThread, p, threads, nodeBytes, exact) > budget) p -= 16;
        return p;
    }
    // bytes of global memory one workgroup needs for its cold chunks
    static constexpr size_t coldBytes(uint32_t pool) { return (size_t)NCOLD * pool * 16; }
};

template <class V>
struct Pool {
    using Lay = StagedLayout<V::POL>;
    float4* st;       // LDS [NHOT][P]: chunk 0 (position, time), 1 (direction, wavelength), 2 (meta)
    float4* sc;       // LDS [NSC][P]
    uint16_t* q;      // rings
    uint32_t* ctr;    // counters
    typedef float vec4 __attribute__((ext_vector_type(4))); // (HIP's float4 has no constructors for address-space qualified objects)
    THX_GLOBAL vec4* cold; // global [P][NCOLD] of this workgroup: the chunks of a photon lie side by side (48 / 80 B)
    // cold chunk k: 0 {lin, log, medium}, 1 {n, vg, mu_s, mu_e}, 2 {stokes}, 3 {polRef}
    THX_DEV float4 ldCold(uint32_t k, uint32_t slot) const {
        const vec4 v = cold[slot * Lay::NCOLD + k];
        return make_float4(v.x, v.y, v.z, v.w);
    }
    THX_DEV void stCold(uint32_t k, uint32_t slot, float4 v) const { cold[slot * Lay::NCOLD + k] = vec4{v.x, v.y, v.z, v.w}; }
    uint32_t P, cap;           // pool size, ring capacity (>= P)
    uint32_t magic, magicTS;   // ceil(2^37 / cap), ceil(2^37 / (2 cap)): position -> lap by multiplication
    // ring s holds slot ids; the TS ring (last) has room for 2 cap entries (two shadow rays per photon)
    uint32_t qHalf, ctrWord; // LDS half-word index of ring 0, LDS word index of the counters
    // LDS half-word index of ring s
    THX_DEV uint32_t queue(int s) const { return qHalf + (uint32_t)(s == S_TS ? 6 : (s > S_TS ? s - 1 : s)) * cap; }
    // LDS half-word index of the cell of ring position `pos` (positions count up for the whole launch) and the lap tag of
    // that position (bit 14 of the cell's entry): a claimer never takes the entry of the previous lap for its own, however
    // late that lap's claimer empties the cell
    THX_DEV uint32_t cell(int s, uint32_t pos, uint32_t& tag) const {
        const uint32_t d = s == S_TS ? 2u * cap : cap;
        const uint32_t lap = (uint32_t)(((uint64_t)pos * (uint64_t)(s == S_TS ? magicTS : magic)) >> 37);
        tag = (lap & 1u) << 14;
        return queue(s) + (pos - lap * d);
    }
    THX_DEV uint32_t* flagsWord(uint32_t slot) const { return reinterpret_cast<uint32_t*>(&st[2 * P + slot]) + 2; }
};

// Scheduler accesses to LDS that other wavefronts change concurrently: relaxed atomics on
// the __shared__ symbol itself. (`volatile` pointers lose the LDS address space -- the compiler
// then emits FLAT loads with system scope and a vmcnt+lgkmcnt wait for every counter or ring
// cell read -- and plain loads could be hoisted out of the spin loops.)
THX_DEV uint32_t ldsRead16(uint32_t half) { return __atomic_load_n(reinterpret_cast<uint16_t*>(thx_smem) + half, __ATOMIC_RELAXED); }
THX_DEV void ldsWrite16(uint32_t half, uint32_t v) { __atomic_store_n(reinterpret_cast<uint16_t*>(thx_smem) + half, (uint16_t)v, __ATOMIC_RELAXED); }
THX_DEV unsigned long long ldsRead64(uint32_t dword2) { return __atomic_load_n(reinterpret_cast<unsigned long long*>(thx_smem) + dword2, __ATOMIC_RELAXED); }
THX_DEV void ldsWrite32(uint32_t word, uint32_t v) { __atomic_store_n(reinterpret_cast<uint32_t*>(thx_smem) + word, v, __ATOMIC_RELAXED); }
constexpr uint32_t kNoCell = 0xFFFFFFFFu;

// ---- state <-> registers -------------------------------------------------
// hot 0 {pos.xyz, time}  hot 1 {dir.xyz, wavelength}  hot 2 {local idx, dim, flags, distShadow}
// cold 0 {lin, log, medium lo, hi}  cold 1 {n, vg, mu_s, mu_e}  cold 2 {stokes}  cold 3 {polRef.xyz, -}
template <class V>
THX_DEV Ray loadRay(const Pool<V>& p, uint32_t s) {
    const float4 c0 = p.st[0 * p.P + s], c1 = p.st[1 * p.P + s], c2 = p.ldCold(0, s), c3 = p.ldCold(1, s);
    Ray r;
    r.position = f3{c0.x, c0.y, c0.z};
    r.time = c0.w;
    r.direction = f3{c1.x, c1.y, c1.z};
    r.wavelength = c1.w;
    r.lin_contrib = c2.x;
    r.log_contrib = c2.y;
    r.medium = (uint64_t)__float_as_uint(c2.z) | ((uint64_t)__float_as_uint(c2.w) << 32);
    r.constants = MediumConstants{c3.x, c3.y, c3.z, c3.w};
    r.stokes = f4{1, 0, 0, 0};
    r.polRef = f3{0, 0, 0};
    if constexpr (V::POL) {
        const float4 c5 = p.ldCold(2, s), c6 = p.ldCold(3, s);
        r.stokes = f4{c5.x, c5.y, c5.z, c5.w};
        r.polRef = f3{c6.x, c6.y, c6.z};
    }
    return r;
}
// `cold`: which cold chunks the caller may have changed (bit k = chunk k); the others are not written again
template <class V>
THX_DEV void storeRay(const Pool<V>& p, uint32_t s, const Ray& r, uint32_t cold = 0xFu) {
    p.st[0 * p.P + s] = make_float4(r.position.x, r.position.y, r.position.z, r.time);
    p.st[1 * p.P + s] = make_float4(r.direction.x, r.direction.y, r.direction.z, r.wavelength);
    if (cold & 1u) p.stCold(0, s, make_float4(r.lin_contrib, r.log_contrib, __uint_as_float((uint32_t)r.medium), __uint_as_float((uint32_t)(r.medium >> 32))));
    if (cold & 2u) p.stCold(1, s, make_float4(r.constants.n, r.constants.vg, r.constants.mu_s, r.constants.mu_e));
    if constexpr (V::POL) {
        if (cold & 4u) p.stCold(2, s, make_float4(r.stokes.x, r.stokes.y, r.stokes.z, r.stokes.w));
        if (cold & 8u) p.stCold(3, s, make_float4(r.polRef.x, r.polRef.y, r.polRef.z, 0.0f));
    }
}

struct Meta {
    uint32_t local, dim, flags;
    float distShadow;
};
template <class V>
THX_DEV Meta loadMeta(const Pool<V>& p, uint32_t s) {
    float4 c = p.st[2 * p.P + s];
    return Meta{__float_as_uint(c.x), __float_as_uint(c.y), __float_as_uint(c.z), c.w};
}
template <class V>
THX_DEV void storeMeta(const Pool<V>& p, uint32_t s, const Meta& m) {
    p.st[2 * p.P + s] = make_float4(__uint_as_float(m.local), __uint_as_float(m.dim), __uint_as_float(m.flags), m.distShadow);
}
template <class V>
THX_DEV void bindLane(Lane<V>& L, const Meta& m) {
    L.local = m.local;
    L.idx = L.a->p.photon_offset + m.local;
    L.dim = m.dim;
}

// TLAS root test done by the requesting stage, so rays that miss the whole scene
// never enter the traversal queue. Same test rayQuery() starts with.
template <class V>
THX_DEV bool rootHit(Lane<V>& L, f3 org, f3 dir, float tmax) {
    const SceneView& sc = L.a->scene;
    bool hit = slabRoot(qRay(tlasFrame0(sc), tlasFrame1(sc), org, dir), tmax);
    if (!hit) { // account like rayQuery would have: one query, one node fetched
        L.cnt.add(ST_QUERIES);
        L.cnt.add(ST_NODES);
    }
    return hit;
}

// end of one loop iteration of tracer.scene.forward.glsl:70-105; returns true if
// the photon goes on to another step
template <class V>
THX_DEV bool endStep(Lane<V>& L, Ray& ray, Meta& m, int result) {
    const uint32_t step = (m.flags & F_STEP_MASK) >> F_STEP_SHIFT;
    if (L.a->flags & THX_FLAG_PHOTON) result = photonRoulette(L, ray, result); // changes the ray: store it afterwards
    onEvent(L, ray, result, step);
    if (result < 0) {
        m.flags &= ~F_ALIVE;
        return false;
    }
    bool allow = true;
    if constexpr (V::MIS) allow = result != RESULT_CODE_RAY_SCATTERED;
    m.flags = (m.flags & ~F_ALLOW) | (allow ? F_ALLOW : 0u);
    if (step == L.a->path_length) {
        onEvent(L, ray, RESULT_CODE_MAX_ITER, L.a->path_length + 1);
        m.flags &= ~F_ALIVE;
        return false;
    }
    m.flags += 1u << F_STEP_SHIFT;
    return true;
}

// trace() after the ray query + processInteraction() (scene.traverse.glsl:324-351,
// 361-397) for one photon; `qh` is its primary query result. Returns the next
// stage: 1 scatter stage(s), 2 next step, 3 finished.
template <class V>
THX_DEV int postPrimary(Lane<V>& L, const Pool<V>& pool, uint32_t slot, const QueryHit& qh) {
    const LaunchArgs& args = *L.a;
    const thx_propagate_params& prop = args.p.scene.propagation;
    Ray ray = loadRay(pool, slot);
    Meta m = loadMeta(pool, slot);
    bindLane(L, m);
    const uint32_t step = (m.flags & F_STEP_MASK) >> F_STEP_SHIFT;
    // the photon tracer always scatters: processInteraction(..., last = false)
    const bool last = step == args.path_length && !(args.flags & THX_FLAG_PHOTON);
    const bool allowResponse = (m.flags & F_ALLOW) != 0u;
    float4 s0 = pool.sc[0 * pool.P + slot];
    float dist = s0.x;
    const float sampledDist = s0.y;
    SurfaceHit hit;
    hit.valid = false;
    int queryResult = RESULT_CODE_RAY_MISSED;
    if (qh.valid) {
        queryResult = processRayQuery(L, ray.direction, ray.medium, qh, hit);
        if (hit.valid) dist = distance(ray.position, hit.worldPos);
        if constexpr (V::MIS) {
            if ((m.flags & F_MIS_TARGET) && hit.valid && dist > sampledDist && queryResult >= 0) {
                processShadowRay(L, ray, hit);
                hit.valid = false;
                dist = sampledDist;
            }
        }
    }
    int propResult = propagateRay(ray, dist, prop);
    updateRayIS(ray, dist, prop, hit.valid);
    int result = queryResult <= ERROR_CODE_MAX_VALUE ? queryResult : propResult;
    bool scatterNeeded = false;
    if (result >= 0) {
        if (hit.valid) {
            result = processHit(L, ray, hit, allowResponse);
        } else {
            scatterNeeded = !last;
            result = RESULT_CODE_RAY_SCATTERED;
        }
    }
    int dest;
    if (scatterNeeded) dest = 1;
    else dest = endStep(L, ray, m, result) ? 2 : 3;
    // without a surface interaction only the contributions changed (and what the photon roulette of endStep rescales)
    storeRay(pool, slot, ray, hit.valid ? 0xFu : ((args.flags & THX_FLAG_PHOTON) ? 0x5u : 0x1u));
    m.dim = L.dim;
    storeMeta(pool, slot, m);
    return dest;
}

// The walk of one batch: one ray per lane, fresh or taken up again (Walker::start), round by round until every
// lane is through -- or until so few are left that they are set aside (`parked`, see LaunchArgs::park_below).
template <class V>
THX_DEV QueryHit walkBatch(Lane<V>& L, f3 org, f3 dir, float tmax, bool resumed, THX_GLOBAL uint32_t* area, bool& parked, uint32_t freeRing2, uint32_t busyMin) {
    Walker<V, true> wk(L);
    wk.start(org, dir, tmax, resumed, area);
    uint32_t rounds = 0;
    while (true) {
        const unsigned long long walking = __ballot(!wk.done());
        if (walking == 0ull) break;
        if (rounds >= L.a->park_rounds && (uint32_t)__popcll(walking) < L.a->park_below) {
            // Setting rays aside frees the wavefront for a fuller batch -- worth it while the workgroup has photons for every
            // wavefront. At the end of a launch, when the pool runs empty, it only adds hand-overs to the life of the last
            // photons, which is what the launch then waits for: with fewer photons in flight than 3/4 of the lanes (of the pool), walk on.
            const unsigned long long fr = ldsRead64(freeRing2); // {head, tail} of the ring of free slots
            const uint32_t inFlight = L.a->pool - ((uint32_t)(fr >> 32) - (uint32_t)fr);
            if (inFlight >= busyMin) {
                if (!wk.done()) {
                    wk.save(area);
                    parked = true;
                }
                break;
            }
        }
#ifdef THX_STAGE_PROFILE
        if (L.prof != nullptr && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(1))) {
            atomicAdd(&L.prof[29], (unsigned long long)__popcll(walking));
            atomicAdd(&L.prof[30], 1ull);
        }
#endif
        wk.advance();
        ++rounds;
    }
    if (wk.h.valid && !parked) L.cnt.add(ST_HITS);
    wk.h.aborted = wk.overflow;
#ifdef THX_STAGE_PROFILE
    wk.h.trips = wk.pLane;
    if ((threadIdx.x & 63u) == 0u && L.prof != nullptr) { // wave-level sums (s_memtime is uniform)
        atomicAdd(&L.prof[24], wk.pInner); atomicAdd(&L.prof[25], wk.pLeaf); atomicAdd(&L.prof[26], wk.pOther);
        atomicAdd(&L.prof[27], wk.pRounds); atomicAdd(&L.prof[28], wk.pTrips);
    }
#endif
    return wk.h;
}

template <class V>
__device__ void sceneStagedBody(const LaunchArgs& args) {
    using Lay = StagedLayout<V::POL>;
    const uint32_t T = blockDim.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t n_bins = args.n_bins;
    const uint32_t P = args.pool;

    float* hist = ldsHist();
    const uint32_t histWords = (n_bins + 3u) & ~3u;
    Pool<V> pool;
    pool.P = P;
    pool.st = reinterpret_cast<float4*>(ldsWords() + histWords);
    pool.sc = pool.st + Lay::NCH * P;
    pool.cold = (THX_GLOBAL typename Pool<V>::vec4*)(args.cold) + (size_t)blockIdx.x * Lay::NCOLD * P;
    const uint32_t stackWord0 = histWords + (Lay::NCH + Lay::NSC) * P * 4;
    pool.q = reinterpret_cast<uint16_t*>(ldsWords() + stackWord0 + args.stack_wave_words * (T >> 6));
    pool.cap = args.ring_cap;
    pool.magic = args.ring_magic;
    pool.magicTS = args.ring_magic_ts;
    pool.ctr = reinterpret_cast<uint32_t*>(pool.q + 8 * pool.cap);
    pool.qHalf = 2u * (stackWord0 + args.stack_wave_words * (T >> 6));
    pool.ctrWord = pool.qHalf / 2u + 4u * pool.cap;

    // response.histogram.glsl:20-29 initResponse; every slot starts free
    for (uint32_t i = tid; i < n_bins; i += T) hist[i] = 0.0f;
    for (uint32_t i = tid; i < 8 * pool.cap; i += T) pool.q[i] = (uint16_t)(i < P ? i : Q_EMPTY);
    if constexpr (V::LDSN) { // the walk's copy of the pair records (two 16-B halves each)
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const THX_GLOBAL u4* src = (const THX_GLOBAL u4*)(args.scene.nodes);
        u4* dst = reinterpret_cast<u4*>(ldsWords() + args.lds_node_word);
        for (uint32_t i = tid; i < 2u * args.lds_node_pairs; i += T) dst[i] = src[i];
    }

    Lane<V> L;
    L.a = &args;
    L.stackBase = stackWord0 + wave * args.stack_wave_words;
    L.cnt.clear();
    L.idx = 0; L.local = 0; L.dim = 0;

    // contiguous photon range of this workgroup
    const uint32_t batch = args.p.dispatch.batchSize;
    const uint64_t per = ((uint64_t)batch + gridDim.x - 1) / gridDim.x;
    const uint64_t b64 = per * blockIdx.x;
    const uint32_t begin = b64 < batch ? (uint32_t)b64 : batch;
    const uint32_t end = (b64 + per) < batch ? (uint32_t)(b64 + per) : batch;
    if (tid < C_NUM) pool.ctr[tid] = tid == (uint32_t)cTail(S_BIRTH) ? P : (tid == C_CURSOR ? begin : 0u);
    if (tid == 0 && (args.flags & (1u << 31))) pool.ctr[C_ABORT] = W_TEST; // THX_TEST_ABORT: exercises the watchdog exit
    __syncthreads();

    const thx_propagate_params& prop = args.p.scene.propagation;
    const uint64_t sceneMedium = (uint64_t)args.p.scene.sceneMedium[0] | ((uint64_t)args.p.scene.sceneMedium[1] << 32);
    auto rl = [](uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k); };

    // What a finished stage hands to the scheduler: every lane appends at most two entries,
    // (dstA, entA) to any ring and (entB) to the TS ring (second shadow ray of the MIS stage).
    int dstA = -1;
    uint32_t entA = 0, entB = 0;
    bool hasB = false;

    // The scheduling code runs ~5 400 times per wavefront and launch between stages of a few
    // thousand cycles, so its INSTRUCTION COUNT matters as much as its LDS round trips (a
    // wavefront issues one instruction every ~10 cycles while three others keep the SIMD busy):
    // appends are one LDS atomic per lane (same-address atomics serialise inside the LDS, one
    // instruction for the wave), the fill levels are compared in lanes 0..7 with a 3-step DPP max.
    // Lane-constant helpers of that comparison (lane q < 7 stands for ring q):
    const uint32_t qLane = lane & 7u;
    //   priority on ties: walk, A, BM, M, C, BIRTH;   0 = lane does not compete (TS, pair 7)
    const uint32_t prioLane = (0x0123456u >> (4u * (qLane == (uint32_t)S_T0 ? 0u : qLane == (uint32_t)S_A ? 1u : qLane == (uint32_t)S_BM ? 2u
                              : qLane == (uint32_t)S_M ? 3u : qLane == (uint32_t)S_C ? 4u : qLane == (uint32_t)S_BIRTH ? 5u : 6u))) & 7u;
    const uint32_t keepMask = prioLane != 0u ? 0xFFFFFFFFu : 0u;
    const uint32_t t0Mask = qLane == (uint32_t)S_T0 ? 0xFFFFFFFFu : 0u;
    const bool birthLane = qLane == (uint32_t)S_BIRTH;

#ifdef THX_STAGE_PROFILE
    const unsigned long long profWg0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long profIdle0 = __builtin_amdgcn_s_memtime();
    unsigned long long* const profLds = reinterpret_cast<unsigned long long*>(pool.ctr + C_PROF);
    L.prof = profLds;
#endif
    uint32_t idleSpins = 0;
    while (true) {
        // ===== LDS round trip 1: reserve a ring cell per entry of the finished stage and
        // fetch the counter block (issued behind the atomics, so it sees them).
#ifdef THX_SCHED_PRIO
        __builtin_amdgcn_s_setprio(THX_SCHED_PRIO);
#endif
        const bool pushA = dstA >= 0;
        uint32_t posA = 0, posB = 0;
        if (__ballot(pushA) != 0ull) {
            if (pushA) posA = atomicAdd(&pool.ctr[cTail(dstA)], 1u);
            if (__ballot(hasB) != 0ull)
                if (hasB) posB = atomicAdd(&pool.ctr[cTail(S_TS)], 1u);
        }
        const unsigned long long hv = ldsRead64(pool.ctrWord / 2u + qLane); // {head, tail} / {cursor, abort}
        const uint32_t hvx = (uint32_t)hv, hvy = (uint32_t)(hv >> 32);
        // (rings that are not a power of two long: Pool::cell is exact below 2^25 appends; the host sizes them so only for
        // launches that stay far below -- should one get there all the same, it ends like a tripped watchdog, loudly)
        if (__ballot(qLane < 7u && hvy >= (1u << 25)) != 0ull && (pool.cap & (pool.cap - 1u)) != 0u) ldsWrite32(pool.ctrWord + C_ABORT, W_RING_LIMIT);

        // ring cells of this lane's entries
        uint32_t cellA = kNoCell, cellB = kNoCell; // LDS half-word indices
        if (pushA) {
            uint32_t tag;
            cellA = pool.cell(dstA, posA, tag);
            entA |= tag;
        }
        if (hasB) {
            uint32_t tag;
            cellB = pool.cell(S_TS, posB, tag);
            entB |= tag;
        }

        // ---- pick the next batch: the fullest ring; both walk rings count as one
        const uint32_t c = hvy - hvx;
        const uint32_t cursor = rl(hvx, 7), abortFlag = rl(hvy, 7);
        const uint32_t freeSlots = rl(c, S_BIRTH), cT0 = rl(c, S_T0), cTS = rl(c, S_TS);
        const uint32_t remaining = cursor < end ? end - cursor : 0u;
        uint32_t e = (c & keepMask) + (cTS & t0Mask);
        e = min(e, birthLane ? remaining : 0xFFFFFFFFu);
        // (letting shading stages go first while a walk batch would not fill the wavefront,
        // or waiting for fuller batches, was measured slower: greedy wins)
        uint32_t key = (e << 3) | prioLane;
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x111, 0xF, 0xF, false)); // row_shr:1
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x112, 0xF, 0xF, false)); // row_shr:2
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x114, 0xF, 0xF, false)); // row_shr:4
        key = rl(key, 7);
        const uint32_t best = key >> 3;
        // cStage: S_NUM = nothing now, S_NUM + 1 = leave (aborted, or every photon is finished)
        uint32_t cStage = S_NUM, pick = S_NUM;
        if (abortFlag != 0u || (cursor >= end && freeSlots == P)) cStage = S_NUM + 1;
        else if (best != 0u) pick = (((uint32_t)S_BIRTH << 4) | ((uint32_t)S_C << 8) | ((uint32_t)S_M << 12) | ((uint32_t)S_BM << 16) |
                                    ((uint32_t)S_A << 20) | ((uint32_t)S_T0 << 24)) >> ((key & 7u) * 4u) & 15u;
        // claim parameters: lane 0 claims ring qa (n0 entries from ha), lane 1 ring TS for a walk
        uint32_t qa = 0, ha = 0, n0 = 0, hb = 0, n1 = 0;
        if (pick == (uint32_t)S_T0) {
            qa = S_T0; ha = rl(hvx, S_T0); n0 = min(cT0, 64u);
            hb = rl(hvx, S_TS); n1 = min(cTS, 64u - n0);
            // A ring that fills the wavefront by itself gives a batch of one kind of ray: what follows the walk differs for
            // primary and shadow rays, and a mixed batch runs both (config 2 / 5 -3 %; smaller thresholds than a full
            // wavefront, or never mixing, are slower: profiles/r02_park.txt).
            if (cTS >= 64u) { n1 = 64u; n0 = 0u; }
        } else if (pick < (uint32_t)S_NUM) {
            qa = pick; ha = rl(hvx, pick); n0 = min(best, 64u);
        }

        // ===== LDS round trip 2: are my ring cells empty (they are, unless the claimer of the
        // previous lap is ~10^4 cycles late), the claim itself (CAS on the ring head), and,
        // speculatively, the entries the claim would give this wavefront.
        uint32_t casOld = 0;
        uint32_t claimCell = kNoCell;
        uint32_t wantTag = 0;
        if (lane < n0) {
            claimCell = pool.cell((int)qa, ha + lane, wantTag);
        } else if (lane < n0 + n1) {
            claimCell = pool.cell(S_TS, hb + (lane - n0), wantTag);
        }
        // (the three reads go out together, for every lane: a lane without a cell reads an unused counter word -- as reads
        // inside `if`s each of them was a branch with its own wait for the LDS, three round trips instead of one)
        const uint32_t noCell = 2u * (pool.ctrWord + 24u);
        uint32_t chkA = ldsRead16(cellA != kNoCell ? cellA : noCell);
        uint32_t chkB = ldsRead16(cellB != kNoCell ? cellB : noCell);
        uint32_t spec = ldsRead16(claimCell != kNoCell ? claimCell : noCell);
        if (lane == 0 && n0 > 0u) casOld = atomicCAS(&pool.ctr[cHead((int)qa)], ha, ha + n0);
        if (lane == 1 && n1 > 0u) casOld = atomicCAS(&pool.ctr[cHead(S_TS)], hb, hb + n1);
        chkA = cellA != kNoCell ? chkA : Q_EMPTY;
        chkB = cellB != kNoCell ? chkB : Q_EMPTY;
        spec = claimCell != kNoCell ? spec : Q_EMPTY;

        // ---- publish the entries of the finished stage. What the stage stored (cold chunks in global memory, hot chunks in
        // LDS) has to be visible before an entry is: the release fence waits for those stores -- here, behind the two LDS round
        // trips above, which need nothing of it (a reserved cell without its entry makes the claimer wait, not read).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (cellA != kNoCell) {
            for (uint32_t spin = 0; chkA != Q_EMPTY; ++spin) {
                if (spin > kCellSpinLimit) { ldsWrite32(pool.ctrWord + C_ABORT, W_APPEND); break; }
                chkA = ldsRead16(cellA);
            }
            ldsWrite16(cellA, entA);
        }
        if (cellB != kNoCell) {
            for (uint32_t spin = 0; chkB != Q_EMPTY; ++spin) {
                if (spin > kCellSpinLimit) { ldsWrite32(pool.ctrWord + C_ABORT, W_APPEND); break; }
                chkB = ldsRead16(cellB);
            }
            ldsWrite16(cellB, entB);
        }
        dstA = -1; hasB = false;

        // ---- outcome of the claim
        const bool ok0 = n0 > 0u && rl(casOld, 0) == ha;
        const bool ok1 = n1 > 0u && rl(casOld, 1) == hb;
        if (cStage != (uint32_t)S_NUM + 1u && (ok0 || ok1)) {
            cStage = pick;
        }
        const uint32_t stage = cStage;
#ifdef THX_STAGE_PROFILE
        const unsigned long long profTb = __builtin_amdgcn_s_memtime();
#endif
        if (stage == (uint32_t)S_NUM + 1u) break;
        if (stage == (uint32_t)S_NUM) { // nothing to do right now (or another wavefront was faster)
#ifdef THX_STAGE_PROFILE
            if (lane == 0) atomicAdd(&profLds[11], 1ull);
#endif
            if (++idleSpins > kIdleSpinLimit) {
                if (lane == 0) ldsWrite32(pool.ctrWord + C_ABORT, W_IDLE);
                break;
            }
            if (best == 0u) __builtin_amdgcn_s_sleep(THX_IDLE_SLEEP);
            continue;
        }
        idleSpins = 0;
        // lanes [0, n0) hold ring qa's entries (if that claim succeeded), lanes [n0, n0 + n1) the
        // shadow rays; a walk batch may consist of the second part only
        bool act = (lane < n0 && ok0) || (lane >= n0 && lane < n0 + n1 && ok1);
        uint32_t entry = spec;
        if (act) {
            // reserved but not written yet: the cell is empty (or, after a very late claimer of
            // the previous lap, still holds that lap's entry: wrong tag)
            for (uint32_t spin = 0; (entry & 0xC000u) != wantTag; ++spin) {
                if (spin > kCellSpinLimit) { ldsWrite32(pool.ctrWord + C_ABORT, W_CLAIM); entry = 0; act = false; break; }
                entry = ldsRead16(claimCell);
            }
            ldsWrite16(claimCell, Q_EMPTY);
            entry &= 0x3FFFu;
        } else {
            entry = 0;
        }
        [[maybe_unused]] const uint32_t n = (ok0 ? n0 : 0u) + (ok1 ? n1 : 0u);
        // photon indices of a birth batch; beyond `end` the slot goes back to the free ring
        uint32_t birthFirst = 0;
        if (stage == (uint32_t)S_BIRTH) {
            if (lane == 0) birthFirst = atomicAdd(&pool.ctr[C_CURSOR], n0);
            birthFirst = __builtin_amdgcn_readfirstlane(birthFirst);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef THX_SCHED_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef THX_STAGE_PROFILE
        const unsigned long long profT0 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            atomicAdd(&profLds[7], profT0 - profIdle0);
            atomicAdd(&profLds[14], profT0 - profTb); // entries
        }
#endif

        // Only the walk keeps its parameters in SGPRs for the whole kernel; the shading stages read theirs
        // again from the kernel-argument segment when they start (a few scalar loads per 64 photons).
        // Otherwise every parameter any stage uses is live across the scheduler loop and the SGPR file
        // overflows into v_writelane / v_readlane pairs inside the walk.
        L.a = stage == S_T0 ? &args : coldArgsPtr();
        bool doA = false, doBM = false, doM = false; // lanes that run the A / BM / M part after the stage body
        Ray rayA;
        Meta mA{0u, 0u, 0u, 0.0f};
        uint32_t slotA = 0;
        switch (stage) {
        case S_BIRTH: { // tracer.scene.forward.glsl:50-67; goes straight on into the A part
            slotA = entry;
            const bool born = act && birthFirst + lane < end;
            if (born) {
                mA = Meta{birthFirst + lane, 0u, 0u, 0.0f};
                bindLane(L, mA);
                rayA = sampleRay(L, sceneMedium);
                onEvent(L, rayA, RESULT_CODE_RAY_CREATED, 0);
                mA.dim = L.dim;
                // (the host guarantees path_length >= 1 for this kernel)
                mA.flags = F_ALIVE | ((args.flags & THX_FLAG_DISABLE_DIRECT) ? 0u : F_ALLOW) | (1u << F_STEP_SHIFT);
                storeRay(pool, slotA, rayA);
                doA = true;
            }
            entA = slotA;
            if (act && !born) dstA = S_BIRTH; // beyond the photon range: the slot goes back
            break;
        }
        case S_A: { // photons that start their next step after a surface interaction
            slotA = entry;
            if (act) {
                rayA = loadRay(pool, slotA);
                mA = loadMeta(pool, slotA);
                doA = true;
            }
            break;
        }
        case S_T0: { // closest-hit walks: primary rays (which 0; a surface hit is finished here,
                     // scene.traverse.glsl:312-351) and shadow rays (which 1, 2; :185-214)
#ifdef THX_WALK_PRIO
            __builtin_amdgcn_s_setprio(THX_WALK_PRIO);
#endif
            const uint32_t slot = entry & 0x7FFu, which = entry >> 12;
            const bool resumed = (entry & E_RESUME) != 0u;
            bool parked = false;
            int dest = 0;
            bool lastShadow = false;
            if (act) {
                float4 c0 = pool.st[0 * P + slot];
                f3 dir;
                float tmax, weight = 0.0f;
                if (which == 0) {
                    float4 c1 = pool.st[1 * P + slot];
                    dir = normalize(f3{c1.x, c1.y, c1.z});
                    tmax = pool.sc[0 * P + slot].x;
                } else {
                    float4 d = which == 1u ? pool.sc[slot] : pool.ldCold(Lay::C_SHADOW2, slot);
                    dir = f3{d.x, d.y, d.z};
                    weight = d.w;
                    tmax = pool.st[2 * P + slot].w;
                }
#ifdef THX_STAGE_PROFILE
                const unsigned long long wt0 = __builtin_amdgcn_s_memtime();
#endif
                QueryHit h = walkBatch(L, f3{c0.x, c0.y, c0.z}, dir, tmax, resumed,
                                       (THX_GLOBAL uint32_t*)(args.park) + ((size_t)(blockIdx.x * P + slot) * 2u + (which == 2u ? 1u : 0u)) * (8u + max(args.stack_depth, 8u)),
                                       parked, pool.ctrWord / 2u + (uint32_t)S_BIRTH, (min(T, P) * 3u) >> 2);
#ifdef THX_STAGE_PROFILE
                // per wave: cycles until the first lane / the whole wave is through the walk
                const unsigned long long wt1 = __builtin_amdgcn_s_memtime();
                if ((tid & 63u) == 0u) {
                    atomicAdd(&profLds[8], wt1 - wt0);
                    atomicAdd(&profLds[10], 1ull);
                }
                { // trips per ray: sums and per-batch maxima by ray kind (0 primary, 1 shadow); histogram of trips
                    const uint32_t kind = which == 0 ? 0u : 1u;
                    atomicAdd(&profLds[40 + 2 * kind], 1ull);
                    atomicAdd(&profLds[41 + 2 * kind], (unsigned long long)h.trips);
                    uint32_t mAll = h.trips;
                    for (int o = 32; o > 0; o >>= 1) mAll = max(mAll, (uint32_t)__shfl_xor((int)mAll, o));
                    const unsigned long long first = __ballot(1);
                    if ((tid & 63u) == (uint32_t)__builtin_ctzll(first)) atomicAdd(&profLds[44], (unsigned long long)mAll);
                    const uint32_t bin = h.trips <= 4 ? 0u : (h.trips <= 16 ? 1u : (h.trips <= 32 ? 2u : 3u));
                    if (bin < 3) atomicAdd(&profLds[45 + bin], 1ull); // <= 4 / 5-16 / 17-32; the rest follows from the totals
                }
#endif
                if (parked) {
                    // nothing to finish yet: the ray goes back into its ring
                } else if (which == 0) {
                    if (h.aborted) { // the traversal stack overflowed: ERROR_CODE_TRACE_ABORT ends the photon
                        Ray ray = loadRay(pool, slot);
                        Meta m = loadMeta(pool, slot);
                        bindLane(L, m);
                        endStep(L, ray, m, ERROR_CODE_TRACE_ABORT);
                        storeMeta(pool, slot, m);
                        dest = 3;
                    } else {
                        dest = h.valid ? postPrimary(L, pool, slot, h) : 4;
                    }
                } else {
                    if (h.aborted) {
                        L.cnt.add(ST_ERROR); // counted; the shadow ray contributes nothing
                    } else if (h.valid) {
                        Ray s = loadRay(pool, slot); // the photon itself is not modified
                        Meta m = loadMeta(pool, slot);
                        bindLane(L, m);
                        scatterRayIS(L, s, s.medium, dir);
                        SurfaceHit hit;
                        if (processRayQuery(L, s.direction, s.medium, h, hit) >= 0) {
                            s.lin_contrib *= weight;
                            processShadowRay(L, s, hit);
                        }
                    }
                    uint32_t old = atomicSub(pool.flagsWord(slot), F_PEND_ONE);
                    lastShadow = ((old >> F_PEND_SHIFT) & 3u) == 1u;
                }
#ifdef THX_STAGE_PROFILE
                if ((tid & 63u) == 0u) atomicAdd(&profLds[9], __builtin_amdgcn_s_memtime() - wt1);
#endif
            }
#if defined(THX_WALK_PRIO)
            __builtin_amdgcn_s_setprio(0);
#endif
            entA = slot;
            dstA = dest == 4 ? S_BM : (dest == 1 ? (V::MIS ? S_M : S_C) : (dest == 2 ? S_A : (dest == 3 ? S_BIRTH : (lastShadow ? S_C : -1))));
            if (parked) {
                dstA = which == 0u ? S_T0 : S_TS;
                entA = slot | E_RESUME | (which << 12);
            }
            break;
        }
        case S_BM: // primary rays whose walk found nothing: BM part below
            slotA = entry;
            doBM = act;
            break;
        case S_M: // photons that scatter after a surface-free walk: M part below
            slotA = entry;
            doM = act;
            break;
        default: { // S_C: scatterRay, scene.traverse.glsl:391, the end of the step and, for
                   // the photons that go on, the A part of the next one
            slotA = entry;
            if (act) {
                rayA = loadRay(pool, slotA);
                mA = loadMeta(pool, slotA);
                bindLane(L, mA);
                float u0, u1;
                draw2(L, u0, u1); // random2D
                scatterRay(L, rayA, rayA.medium, u0, u1);
                doA = endStep(L, rayA, mA, RESULT_CODE_RAY_SCATTERED);
                storeRay(pool, slotA, rayA, 0xDu); // a scatter leaves the medium constants alone
                mA.dim = L.dim;
                if (!doA) {
                    storeMeta(pool, slotA, mA);
                    dstA = S_BIRTH;
                }
            }
            entA = slotA;
            break;
        }
        }

        // ---- A part: trace() up to the ray query, scene.traverse.glsl:280-310. Entered from
        // the A ring, and directly (photon still in registers) after a birth or a scatter.
        if (__ballot(doA) != 0ull) {
            int dest = 0; // 1 traversal, 2 propagate-only, 3 finished
            if (doA) {
                bindLane(L, mA);
                if (args.flags & THX_FLAG_PHOTON) photonRunDim(L, (mA.flags & F_STEP_MASK) >> F_STEP_SHIFT);
                L.cnt.add(ST_STEPS);
                if (isRayBad(rayA)) {
                    endStep(L, rayA, mA, ERROR_CODE_RAY_BAD);
                    dest = 3;
                } else {
                    f3 dir = normalize(rayA.direction);
                    float dist = sampleScatterLength(rayA, prop, draw(L));
                    float sampledDist = dist;
                    bool mis_target = false;
                    if constexpr (V::MIS) {
                        GuideSample gs = evalTargetGuide(args.p.guide, rayA.position, dir, args.flags);
                        mis_target = (mA.flags & F_ALLOW) != 0u && gs.prob > 0.0f;
                        if (mis_target) {
                            mis_target = gs.dist > dist;
                            dist = fmaxf(gs.dist, dist);
                        }
                    }
                    mA.flags = (mA.flags & ~F_MIS_TARGET) | (mis_target ? F_MIS_TARGET : 0u);
                    pool.sc[0 * P + slotA] = make_float4(dist, sampledDist, 0.0f, 0.0f);
                    dest = rootHit(L, rayA.position, dir, dist) ? 1 : 2;
                }
                mA.dim = L.dim;
                storeMeta(pool, slotA, mA);
                entA = slotA;
                // photons that miss the scene's bounding box propagate right here (BM part: short,
                // -2.5 % on config 2); their MIS set-up stays a ring stage of its own
                if (dest == 1) dstA = S_T0;
                else if (dest == 2) doBM = true; // no walk needed: propagate right away (BM part below)
                else dstA = S_BIRTH;
            }
        }

        // ---- BM part: the photon travels its full sampled distance (trace() after a ray query
        // without a hit, scene.traverse.glsl:324-351)
        if (__ballot(doBM) != 0ull) {
            if (doBM) {
                QueryHit none;
                none.valid = false; none.instance = 0; none.prim = 0; none.t = 0; none.b1 = 0; none.b2 = 0;
                const int dest = postPrimary(L, pool, slotA, none);
                entA = slotA;
                // (going straight on into the M part was measured slower: it then runs with the
                // lanes the walk-bound photons leave empty)
                dstA = dest == 1 ? (V::MIS ? S_M : S_C) : (dest == 2 ? S_A : S_BIRTH);
            }
        }

        // ---- M part: sampleTargetMIS set-up, scene.traverse.glsl:234-262
        if (__ballot(doM) != 0ull) {
            const uint32_t slot = slotA;
            bool t1 = false, t2 = false;
            if (doM) {
                Ray ray = loadRay(pool, slot);
                Meta m = loadMeta(pool, slot);
                bindLane(L, m);
                float pPP;
                float u0, u1;
                draw2(L, u0, u1); // random2D
                const ScatterTabs tabs = scatterTabs(ray.medium); // both phase-function evaluations of this stage share the table headers
                f3 dirPhase = scatter(tabs, ray.direction, u0, u1, pPP, L.cnt);
                GuideSample ts = sampleTargetGuide(L, ray.position);
                float pTT = ts.prob;
                GuideSample ps = evalTargetGuide(args.p.guide, ray.position, dirPhase, args.flags);
                float pTP = ps.prob;
                float pPT = scatterProb(tabs, ray.direction, ts.dir, L.cnt);
                float wTarget = pTT * pPT / (pTT * pTT + pPT * pPT);
                float wPhase = pPP * pPP / (pPP * pPP + pTP * pTP);
                pool.sc[0 * P + slot] = make_float4(dirPhase.x, dirPhase.y, dirPhase.z, wPhase);
                pool.stCold(Lay::C_SHADOW2, slot, make_float4(ts.dir.x, ts.dir.y, ts.dir.z, wTarget));
                t1 = rootHit(L, ray.position, dirPhase, ps.dist);
                t2 = rootHit(L, ray.position, ts.dir, ts.dist);
                m.distShadow = ps.dist; // == ts.dist: both are |center - observer| + radius
                m.dim = L.dim;
                m.flags = (m.flags & ~(3u << F_PEND_SHIFT)) | (((t1 ? 1u : 0u) + (t2 ? 1u : 0u)) << F_PEND_SHIFT);
                storeMeta(pool, slot, m);
            }
            if (t1) {
                dstA = S_TS; entA = slot | (1u << 12);
                hasB = t2; entB = slot | (2u << 12);
            } else if (t2) {
                dstA = S_TS; entA = slot | (2u << 12);
            } else if (doM) {
                dstA = S_C; entA = slot;
            }
        }
#ifdef THX_STAGE_PROFILE
        profIdle0 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            atomicAdd(&profLds[stage], profIdle0 - profT0);
            atomicAdd(&profLds[16 + stage], 1ull);
            atomicAdd(&profLds[32 + stage], (unsigned long long)n);
        }
#endif
    }
    // response.histogram.glsl:35-45 finalizeResponse
    __syncthreads();
    // a tripped scheduler watchdog (a bug, never seen) poisons the light curve instead of
    // returning a plausible partial one
    const uint32_t abortCode = pool.ctr[C_ABORT];
    const bool aborted = abortCode != 0u;
    // ... and raises the tracer's device error word: hit queues, tracks, statistics and reductions into
    // external buffers have no slab to poison (the host checks the word on every readback)
    if (aborted && tid == 0 && args.error != nullptr) atomicOr(args.error, THX_DEVERR_WATCHDOG | (abortCode << 8)); // bits 8..12: which check tripped
#ifdef THX_STAGE_PROFILE
    if (tid < 48 && args.prof) atomicAdd(&args.prof[tid], reinterpret_cast<unsigned long long*>(pool.ctr + C_PROF)[tid]);
    if (tid == 0 && args.prof) { // how evenly the workgroups finish: sum and maximum of their run times (100 MHz ticks)
        const unsigned long long ticks = __builtin_amdgcn_s_memrealtime() - profWg0;
        atomicAdd(&args.prof[15], ticks);
        atomicMax(&args.prof[23], ticks);
    }
#endif
    float* slab = args.slabs + (size_t)blockIdx.x * n_bins;
    for (uint32_t i = tid; i < n_bins; i += T) slab[i] = aborted ? __uint_as_float(0x7FC00000u) : hist[i];

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

} // namespace thx
