// The closest-hit ray query of the scene tracers: what the reference asks of the driver (VK_KHR_ray_query,
// scene.traverse.glsl:312-323 / :198-205), done here -- quantised pair records, the watertight ray / triangle test, and
// `Walker`, the resumable walk of one lane through the two-level hierarchy. A part of dev_trace.hpp kept in a file of its own:
// dev_trace.hpp includes it behind the definitions it needs (Variant, LaunchArgs, Lane, the counters) and in front of the surface
// physics that consumes its hits. References are to /root/reference/src/theia/shader.
#pragma once
#ifndef THX_DEV
#error "dev_walk.hpp is included by dev_trace.hpp"
#endif

namespace thx {

// ============================================================================
// SCENE: BVH walk (replaces the driver's VK_KHR_ray_query) + surface physics
// ============================================================================

struct QueryHit {
    bool valid;
    bool aborted = false; // the traversal stack overflowed: the result is not the closest hit (ERROR_CODE_TRACE_ABORT)
    uint32_t instance, prim; // prim = index of the triangle's record (leaf order), not its index in the mesh
    float t, b1, b2; // b1, b2 = barycentric weights of v1, v2
#ifdef THX_STAGE_PROFILE
    uint32_t trips = 0; // inner-node trips of this lane's own ray
#endif
};

// three consecutive floats at a 4-byte aligned address (a vertex of a TriRecord from byte 0, 4 or 8)
struct P3 {
    float x, y, z;
};
THX_DEV f3 loadP3(const char* p) {
    const P3 v = *reinterpret_cast<const P3*>(p);
    return f3{v.x, v.y, v.z};
}

struct WoopRay {
    uint32_t rot;      // 4 kx: byte offset into a stored vertex (x y z x y) at which (kx, ky, kz) start
    float Sx, Sy, Sz;
    f3 o;              // the ray's origin in the order (kx, ky, kz)
};

// Watertight ray/triangle intersection (Woop, Benthin, Wald 2013): shear constants per (object-space) ray. kz is the
// dominant axis of the direction, (kx, ky, kz) a rotation of (x, y, z) -- which the triangle records offer as a byte
// offset (thx_layout.h: TriRecord). The original also exchanges kx and ky when d[kz] < 0, to keep the winding for
// culling; nothing is culled here, and the exchange does nothing but negate U, V, W, det and T together -- exactly
// (a b - c d rounds to the negative of c d - a b) -- so t, the barycentrics and every decision come out bit for bit
// the same without it (tests: hits bitwise those of the oracle, which does exchange).
THX_DEV WoopRay woopSetup(f3 d, f3 org) {
    WoopRay w;
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int kz = 0;
    if (ay > ax) kz = 1;
    if (az > (kz == 0 ? ax : ay)) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    w.rot = 4u * (uint32_t)kx;
    w.Sx = comp(d, kx) / comp(d, kz);
    w.Sy = comp(d, ky) / comp(d, kz);
    w.Sz = 1.0f / comp(d, kz);
    w.o = f3{comp(org, kx), comp(org, ky), comp(org, kz)};
    return w;
}

// p0, p1, p2: the vertices in the order (kx, ky, kz) (loadP3 at TriRecord vertex + w.rot)
THX_DEV bool woopTri(const WoopRay& w, f3 p0, f3 p1, f3 p2, float& t_io, float& b1,
                     float& b2) {
    f3 A = p0 - w.o, B = p1 - w.o, C = p2 - w.o;
    float Akz = A.z, Bkz = B.z, Ckz = C.z;
    float Ax = A.x - w.Sx * Akz, Ay = A.y - w.Sy * Akz;
    float Bx = B.x - w.Sx * Bkz, By = B.y - w.Sy * Bkz;
    float Cx = C.x - w.Sx * Ckz, Cy = C.y - w.Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float Vv = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || Vv == 0.0f || W == 0.0f) { // edge-on: decide in double
        // (The one place where leaving out the exchange could show: a difference that cancels to zero is +0 whichever way
        // round it is taken, not the negative of the other. Where the original exchanges (d[kz] < 0) take the differences
        // its way round and negate the result: the same value when it is not zero, and the zero that keeps the SIGN of a
        // zero barycentric coordinate what the original gives.)
        const bool ex = w.Sz < 0.0f;
        const double cb = (double)Cx * (double)By, bc = (double)Cy * (double)Bx;
        const double ac = (double)Ax * (double)Cy, ca = (double)Ay * (double)Cx;
        const double ba = (double)Bx * (double)Ay, ab = (double)By * (double)Ax;
        U = ex ? -(float)(bc - cb) : (float)(cb - bc);
        Vv = ex ? -(float)(ca - ac) : (float)(ac - ca);
        W = ex ? -(float)(ab - ba) : (float)(ba - ab);
    }
    // (one decision instead of a chain of short-circuit branches: every one of those is a hand-over to the scalar unit and back)
    const int mixed = (int)(fminf(fminf(U, Vv), W) < 0.0f) & (int)(fmaxf(fmaxf(U, Vv), W) > 0.0f);
    float det = (U + Vv) + W;
    if ((mixed | (int)(det == 0.0f)) != 0) return false;
    float Az = w.Sz * Akz, Bz = w.Sz * Bkz, Cz = w.Sz * Ckz;
    float T = (U * Az + Vv * Bz) + W * Cz;
    float rcpDet = 1.0f / det;
    float t = T * rcpDet;
    if (!(t > 0.0f && t < t_io)) return false;
    t_io = t;
    b1 = Vv * rcpDet;
    b2 = W * rcpDet;
    return true;
}

// Slab test against a 32-B node; returns the entry distance or -1 for a miss.
// The ray is kept as (1/d, -o/d) so that a plane distance is one fma, x and y of a
// corner in one v_pk_fma_f32 (the loaded float4 holds them in an aligned register
// pair). The boxes are padded by the builder and the interval test carries a 4e-7
// slack, so the rounding of this form cannot drop a box that holds a hit.
// 1/d is clamped to +-1e24: with an infinite reciprocal (axis-parallel ray, d = 0) the
// far plane of a slab the origin lies in would come out as inf - inf = NaN, v_max would
// return the near plane's -inf for it and the box would be culled. With 1e24 both planes
// stay finite (|plane| * 1e24 and the quantised form's 2^23 * step * 1e24 are far below FLT_MAX
// for any scene measured in metres) and keep their signs: the padding of the boxes (2e-6
// relative) is far above the 6e-8 relative rounding of the product. (A ray flatter than
// 1e-24 is tested as if it were that steep: it reaches a slab it starts outside of EARLIER
// than the true ray would, never later -- conservative.)
typedef float v2f __attribute__((ext_vector_type(2)));
struct SlabRay {
    v2f inv_xy, noi_xy; // 1/d and -(o/d), x and y
    float inv_z, noi_z;
};
// Round 3: the reciprocal is the hardware's (v_rcp_f32, 1 ulp) instead of a correctly rounded division (~10 instructions
// each, six per ray and frame: profiles/r03_inner_loop.txt). The box tests only have to be conservative: an error of the
// reciprocal scales all plane distances of an axis by the same 1 +- 1.2e-7, which the relative slack of the interval test
// (kSlabSlack, raised from 4e-7 to 1e-6 with this) covers whatever the distance; the ray/triangle test -- what decides a
// hit and its coordinates -- keeps its exact divisions (woopSetup), so hits stay bitwise those of the oracle's brute force.
THX_DEV float slabInv(float d) { return fminf(fmaxf(__builtin_amdgcn_rcpf(d), -1.0e24f), 1.0e24f); }
constexpr float kSlabSlack = 1.000001f;
THX_DEV SlabRay slabRay(f3 o, f3 d) {
    SlabRay r;
    const float ix = slabInv(d.x), iy = slabInv(d.y), iz = slabInv(d.z);
    r.inv_xy = v2f{ix, iy};
    r.noi_xy = v2f{-(o.x * ix), -(o.y * iy)};
    r.inv_z = iz;
    r.noi_z = -(o.z * iz);
    return r;
}
THX_DEV float slabNode(const float4 a, const float4 b, const SlabRay& r, float tmax) {
    const v2f t0 = __builtin_elementwise_fma(v2f{a.x, a.y}, r.inv_xy, r.noi_xy);
    const v2f t1 = __builtin_elementwise_fma(v2f{b.x, b.y}, r.inv_xy, r.noi_xy);
    const float tz0 = __builtin_fmaf(a.z, r.inv_z, r.noi_z), tz1 = __builtin_fmaf(b.z, r.inv_z, r.noi_z);
    float tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fmaxf(fminf(tz0, tz1), 0.0f));
    float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fminf(fmaxf(tz0, tz1), tmax));
    return tn <= tf * kSlabSlack ? tn : -1.0f;
}

// The walk reads 16-bit quantised child boxes (thx_layout.h: NodePair): plane = origin + q * step in
// the frame of the tree. Folded into the ray that is one fma per plane again:
//   t = (origin + q * step - o) / d = q * (step / d) + (origin - o) / d = q * a + b.
// Rounding: a and b carry a few 1e-7 relative of |origin / d| resp. |q step / d| -- the same size as
// the errors of the unquantised form above; the builder pads every box by 2e-6 relative and by one
// quantisation step, the interval test carries the 4e-7 slack.
struct QRay {
    v2f a_xy, b_xy;
    float a_z, b_z;
    // byte selectors (v_perm_b32) that turn a box's near / far plane of an axis -- picked by the sign of the ray from the
    // axis' lo | hi << 16 word -- straight into the float 2^23 + q (see qPlane)
    uint32_t snx, sny, snz, sfx, sfy, sfz;
};
// A 16-bit plane index as a float without a conversion: the bytes of q under the exponent of 2^23 ARE the float 2^23 + q
// (one v_perm_b32 from the record's word and the constant; a v_cvt_f32_u32 with a 16-bit source selector issues no faster
// than the v_perm_b32 that had to pick the half-word for it anyway: profiles/r03_valu_rates.txt). The ray's offset takes
// the 2^23 back out: t = (2^23 + q) a + (b - 2^23 a), one fma as before. Price: b - 2^23 a is rounded at the size of
// 2^23 a, i.e. to half a quantisation step -- the planes a ray sees lie up to half a step off. The builder moves every
// plane one whole step outwards (bvh_build.cpp: qlo / qhi, frames with two steps of room at either end), so the boxes stay
// conservative, and hits are decided by the triangle test alone.
constexpr uint32_t kPlaneMagic = 0x4B000000u; // 2^23 as float
constexpr float kPlaneBase = 8388608.0f;
THX_DEV QRay qRay(const float4 f0, const float4 f1, f3 o, f3 d) { // f0 = {origin, rootRef}, f1 = {step, -}
    const float ix = slabInv(d.x), iy = slabInv(d.y), iz = slabInv(d.z);
    QRay r;
    r.a_xy = v2f{f1.x * ix, f1.y * iy};
    r.b_xy = v2f{__builtin_fmaf(f0.x, ix, -(o.x * ix)), __builtin_fmaf(f0.y, iy, -(o.y * iy))};
    r.a_z = f1.z * iz;
    r.b_z = __builtin_fmaf(f0.z, iz, -(o.z * iz));
    r.b_xy = v2f{__builtin_fmaf(-kPlaneBase, r.a_xy.x, r.b_xy.x), __builtin_fmaf(-kPlaneBase, r.a_xy.y, r.b_xy.y)};
    r.b_z = __builtin_fmaf(-kPlaneBase, r.a_z, r.b_z);
    // v_perm_b32 D, word, magic, sel: selector bytes 4..7 = bytes of the word, 0..3 = bytes of the magic (2: 0x00, 3: 0x4B)
    constexpr uint32_t LO = 0x03020504u, HI = 0x03020706u;
    r.snx = r.a_xy.x < 0.0f ? HI : LO; r.sfx = r.a_xy.x < 0.0f ? LO : HI;
    r.sny = r.a_xy.y < 0.0f ? HI : LO; r.sfy = r.a_xy.y < 0.0f ? LO : HI;
    r.snz = r.a_z < 0.0f ? HI : LO;    r.sfz = r.a_z < 0.0f ? LO : HI;
    return r;
}
THX_DEV float qPlane(uint32_t word, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(word, kPlaneMagic, sel)); }
// one child of a pair record: c.x = lo.x | hi.x << 16, c.y = lo.y | hi.y << 16, c.z = lo.z | hi.z << 16
THX_DEV float slabChild(const float4 c, const QRay& r, float tmax) {
    const uint32_t wx = __float_as_uint(c.x), wy = __float_as_uint(c.y), wz = __float_as_uint(c.z);
    // Near and far plane of every axis picked by the sign of the ray instead of computing both distances and taking
    // min / max: t = q a + b is monotonic in q, so the values are the same ones.
    const v2f tn_xy = __builtin_elementwise_fma(v2f{qPlane(wx, r.snx), qPlane(wy, r.sny)}, r.a_xy, r.b_xy);
    const v2f tf_xy = __builtin_elementwise_fma(v2f{qPlane(wx, r.sfx), qPlane(wy, r.sfy)}, r.a_xy, r.b_xy);
    const float tn_z = __builtin_fmaf(qPlane(wz, r.snz), r.a_z, r.b_z), tf_z = __builtin_fmaf(qPlane(wz, r.sfz), r.a_z, r.b_z);
    const float tn = fmaxf(fmaxf(tn_xy.x, tn_xy.y), fmaxf(tn_z, 0.0f));
    const float tf = fminf(fminf(tf_xy.x, tf_xy.y), fminf(tf_z, tmax));
    return tn <= tf * kSlabSlack ? tn : -1.0f;
}
// the same with a miss at infinity (what the 4-wide visit sorts by)
THX_DEV float slabChildFar(const float4 c, const QRay& r, float tmax) {
    const float t = slabChild(c, r, tmax);
    return t >= 0.0f ? t : INFINITY;
}
// ... in one select: entry distance, or +inf for a miss (a hit's distance is finite: the reciprocals are clamped). What
// the branch-free inner-node loop orders the children by: min / max / <= on these do the work of the four-way branch.
// (`inf`: +infinity held in a vector register by the caller -- as a literal the compiler materialises it again in every trip
// of the loop. minRaw: v_min_f32 as it is -- fminf() on a value that comes out of another block, like the closest hit so far,
// is preceded by a v_max_f32 x, x that quiets a signalling NaN it cannot rule out, again in every trip.)
THX_DEV float minRaw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
THX_DEV float maxRaw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
THX_DEV float slabChildInf(const float4 c, const QRay& r, float tmax, float inf) {
    const uint32_t wx = __float_as_uint(c.x), wy = __float_as_uint(c.y), wz = __float_as_uint(c.z);
    const v2f tn_xy = __builtin_elementwise_fma(v2f{qPlane(wx, r.snx), qPlane(wy, r.sny)}, r.a_xy, r.b_xy);
    const v2f tf_xy = __builtin_elementwise_fma(v2f{qPlane(wx, r.sfx), qPlane(wy, r.sfy)}, r.a_xy, r.b_xy);
    const float tn_z = __builtin_fmaf(qPlane(wz, r.snz), r.a_z, r.b_z), tf_z = __builtin_fmaf(qPlane(wz, r.sfz), r.a_z, r.b_z);
    const float tn = fmaxf(fmaxf(tn_xy.x, tn_xy.y), fmaxf(tn_z, 0.0f));
    const float tf = fminf(fminf(tf_xy.x, tf_xy.y), minRaw(tf_z, tmax));
    return tn <= tf * kSlabSlack ? tn : inf;
}
// the root box of a tree = its whole frame, q = 0 .. 65535
THX_DEV bool slabRoot(const QRay& r, float tmax) {
    constexpr float q0 = kPlaneBase, q1 = kPlaneBase + 65535.0f;
    const v2f t0 = __builtin_elementwise_fma(v2f{q0, q0}, r.a_xy, r.b_xy);
    const v2f t1 = __builtin_elementwise_fma(v2f{q1, q1}, r.a_xy, r.b_xy);
    const float tz0 = __builtin_fmaf(q0, r.a_z, r.b_z), tz1 = __builtin_fmaf(q1, r.a_z, r.b_z);
    float tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fmaxf(fminf(tz0, tz1), 0.0f));
    float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fminf(fmaxf(tz0, tz1), tmax));
    return tn <= tf * kSlabSlack;
}
THX_DEV float4 tlasFrame0(const SceneView& sc) { return make_float4(sc.tlas.origin[0], sc.tlas.origin[1], sc.tlas.origin[2], __uint_as_float(sc.tlas.rootRef)); }
THX_DEV float4 tlasFrame1(const SceneView& sc) { return make_float4(sc.tlas.step[0], sc.tlas.step[1], sc.tlas.step[2], 0.0f); }

#ifndef THX_INNER_KEEP
#define THX_INNER_KEEP 2 // inner-node loop of a round ends early once fewer than KEEP / 8 of its starters are still in it (0: never)
#endif
constexpr uint32_t WALK_DONE = 0xFFFFFFFFu;     // low bits 7
constexpr uint32_t WALK_SENTINEL = 0xFFFFFFFEu; // low bits 6: leave the current instance

// Node array of the walk. (Round 1 could stage it in LDS -- a divergent 16-B gather costs the L1 one cycle per lane and
// instruction, an LDS gather about a tenth of that -- but only where that cost no pool slot, which the BASELINE scenes never
// offered: measured -1.4 % / +1.8 % on configs 2 / 3 with the pool it takes. Removed in round 2.)
template <class V>
THX_DEV const float4* walkNodes(const Lane<V>& L) {
    if constexpr (V::LDSN) return reinterpret_cast<const float4*>(ldsWords() + L.a->lds_node_word);
    else return reinterpret_cast<const float4*>(L.a->scene.nodes);
}

// Closest-hit query: opaque, no culling, 0 < t < tmax, committed closest triangle
// (what scene.traverse.glsl:312-323 / :198-205 ask of the driver). Two-level
// BVH2 read as 32-B pair records (both children of an inner node, boxes quantised to 16 bit);
// the per-lane stack lives in LDS (lane-strided so a push/pop of the whole wave touches 64
// consecutive banks). A child's box is fetched exactly once, together with its sibling's, when
// their parent is opened.
//
// Walker = the resumable state of one lane's query: begin() tests the TLAS root,
// advance() opens inner nodes until the lane holds a leaf and then handles that
// leaf (triangles, instance entry or instance exit). rayQuery() runs one ray per
// lane to completion; walkQueue() (dev_staged.hpp) keeps the lanes of a wave fed
// from a queue of rays.
// FAST (the staged kernel's walks): the LDS part of the stack has one more slot BELOW entry 0 that holds WALK_DONE for
// good (the host adds it to the wavefront's block), 16-bit entries are read sign-extended (the host picks them only for
// scenes whose refs stay below 0x8000), and advance() visits inner nodes in straight-line code while a lane's stack
// top is inside the LDS part.
template <class V, bool FAST = false>
struct Walker {
    Lane<V>& L;
    const float4* nodes;
    const char* tris; // TriRecord array, as bytes
    uint32_t depth, ldsN;
    bool s16;
    bool wide; // the scene has 4-wide records (SceneView::wide; the same for every lane of every wavefront of a launch)
    THX_GLOBAL uint32_t* spill; // this wavefront's deep entries (beyond ldsN): entry e of lane l at e * 64 + l (wave-uniform: lives in SGPRs)
    uint32_t* stack32;
    uint16_t* stack16;
    uint32_t sp, cur, curInst;
    bool inBlas;
    bool overflow; // an entry did not fit the stack: the query's photon ends with ERROR_CODE_TRACE_ABORT
    f3 org, dir, ao;
    QRay sr, wsr;
    WoopRay w;
    QueryHit h;
    float infReg; // +inf in a register of its own (slabChildInf)
#ifdef THX_STAGE_PROFILE
    unsigned long long pInner = 0, pLeaf = 0, pOther = 0, pRounds = 0, pTrips = 0;
    uint32_t pLane = 0;
#endif

    THX_DEV explicit Walker(Lane<V>& lane) : L(lane) {
        nodes = walkNodes(L);
        tris = reinterpret_cast<const char*>(L.a->scene.tris);
        depth = L.a->stack_depth;
        ldsN = L.a->stack_lds;
        // (the staged kernel has variants of its own for 4-wide scenes; the one-variant kernels around it -- probes, lane kernel -- look)
        wide = FAST ? V::WIDE : L.a->scene.wide != 0u;
        // Deep entries: the host sizes the LDS part for what rays typically use and lets the rest of the worst case
        // live in global memory (L2), so the depth of a scene's hierarchy does not come out of the photon pool.
        spill = (THX_GLOBAL uint32_t*)(L.a->spill) + (size_t)((blockIdx.x * (blockDim.x >> 6) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * (depth - ldsN)) * 64u;
        // 16-bit entries when the scene is small enough (halves the LDS footprint).
        // (Storing entry distances for pop-time culling was measured slower: -10 % node
        // fetches but +45 % walk time from the extra LDS traffic and the cull loop.)
        s16 = FAST ? V::S16 : (V::S16 || L.a->stack16 != 0u);
        stack32 = ldsWords() + L.stackBase + (threadIdx.x & 63u);
        // 16-bit entries: lanes l and l + 32 share a 32-bit word, so the 32 lanes the LDS serves per cycle
        // touch 32 different banks whatever their stack pointers are
        stack16 = reinterpret_cast<uint16_t*>(ldsWords() + L.stackBase) + (((threadIdx.x & 31u) << 1) | ((threadIdx.x >> 5) & 1u));
        if constexpr (FAST) { // the sentinel slot below entry 0: "what a pop would return" can be read without asking for an empty stack
            if (V::S16) {
                stack16 += 64;
                stack16[-64] = (uint16_t)0xFFFFu;
            } else {
                stack32 += 64;
                stack32[-64] = WALK_DONE;
            }
        }
        asm volatile("v_mov_b32 %0, 0x7f800000" : "=v"(infReg));
        sp = 0; cur = WALK_DONE; curInst = 0; inBlas = false; overflow = false;
        w.rot = 0; w.Sx = 0; w.Sy = 0; w.Sz = 0; w.o = f3{0, 0, 0};
        h.valid = false; h.instance = 0; h.prim = 0; h.t = 0.0f; h.b1 = 0.0f; h.b2 = 0.0f;
    }
    // The host sizes the stack for the scene (hierarchy depth + 2), so a full stack means a broken
    // invariant; it is not silent: the walk goes on without the entry and its photon is ended with
    // ERROR_CODE_TRACE_ABORT, counted like every other device error (callback.stat.glsl:66-69).
    THX_DEV void push(uint32_t v) {
        if (sp >= depth) { overflow = true; return; }
        if (V::SPILL && sp >= ldsN) spill[(sp - ldsN) * 64u + (threadIdx.x & 63u)] = v;
        else if (s16) stack16[sp * 64] = (uint16_t)v;
        else stack32[sp * 64] = v;
        ++sp;
    }
    THX_DEV uint32_t pop() {
        if (sp == 0) return WALK_DONE;
        --sp;
        if (V::SPILL && sp >= ldsN) return spill[(sp - ldsN) * 64u + (threadIdx.x & 63u)];
        if (FAST && s16) return (uint32_t)(int32_t)(int16_t)stack16[sp * 64]; // 0xFFFE -> WALK_SENTINEL
        if (s16) {
            uint32_t v = stack16[sp * 64];
            return v == 0xFFFEu ? WALK_SENTINEL : v;
        }
        return stack32[sp * 64];
    }
    THX_DEV bool done() const { return cur == WALK_DONE; }

    THX_DEV void begin(f3 o, f3 d, float tmax) {
        const SceneView& sc = L.a->scene;
        org = o; dir = d;
        h.valid = false; h.instance = 0; h.prim = 0; h.t = tmax; h.b1 = 0.0f; h.b2 = 0.0f;
        L.cnt.add(ST_QUERIES);
        sp = 0; inBlas = false; curInst = 0; overflow = false;
        // active ray: world space in the TLAS, object space inside an instance
        ao = org;
        sr = qRay(tlasFrame0(sc), tlasFrame1(sc), org, dir);
        wsr = sr;
        L.cnt.add(ST_NODES);
        cur = slabRoot(sr, h.t) ? sc.tlas.rootRef : WALK_DONE;
        // a scene of one instance: its TLAS is a single leaf, every ray that hits the root box enters
        // the instance -- do it here, with all lanes together, instead of spending a walk round on it
        if (cur != WALK_DONE && (cur & 7u) != 0u) enterInstance(cur >> 3);
    }

    // object-space ray of instance `curInst`: ao, sr, w; returns the ref of the instance's BLAS root
    THX_DEV uint32_t instanceFrame() {
        const float4* rec = reinterpret_cast<const float4*>(L.a->scene.walkInstances + curInst);
        const float4 m0 = rec[0], m1 = rec[1], m2 = rec[2], f0 = rec[3], f1 = rec[4];
        ao = f3{m0.w + ((m0.x * org.x + m0.y * org.y) + m0.z * org.z),
                m1.w + ((m1.x * org.x + m1.y * org.y) + m1.z * org.z),
                m2.w + ((m2.x * org.x + m2.y * org.y) + m2.z * org.z)};
        f3 od = f3{(m0.x * dir.x + m0.y * dir.y) + m0.z * dir.z,
                   (m1.x * dir.x + m1.y * dir.y) + m1.z * dir.z,
                   (m2.x * dir.x + m2.y * dir.y) + m2.z * dir.z};
        sr = qRay(f0, f1, ao, od);
        w = woopSetup(od, ao);
        return __float_as_uint(f0.w);
    }

    // TLAS leaves hold exactly one instance: transform the ray into its object space and dive
    THX_DEV void enterInstance(uint32_t first) {
        curInst = first; // TLAS leaves hold instance ids
        L.cnt.add(ST_INSTANCES);
        const uint32_t root = instanceFrame();
        L.cnt.add(ST_NODES);
        // (A ray that misses the mesh's own box -- the instance's world-space box is looser -- takes the sentinel straight
        // off again and leaves through the ordinary exit in the next round. Staying in world space right here is equivalent
        // and was measured no faster: profiles/r02_park.txt.)
        inBlas = true;
        push(WALK_SENTINEL);
        if (slabRoot(sr, h.t))
            cur = root;
        else
            cur = pop(); // pops the sentinel
    }

    // A walk set aside between two rounds and taken up again later, by any lane of any wavefront of the workgroup
    // (dev_staged.hpp: the walk stage parks the last few rays of a batch instead of running the wavefront nearly empty
    // for them). What is kept: the best hit so far, the position in the hierarchy and the stack; the ray frames are
    // computed again from the ray by start(). The walk itself -- order of visits, arithmetic -- is the same as without the break.
    THX_DEV void save(THX_GLOBAL uint32_t* a) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        *(THX_GLOBAL u4*)(a) = u4{__float_as_uint(h.t), __float_as_uint(h.b1), __float_as_uint(h.b2), h.prim};
        *(THX_GLOBAL u4*)(a + 4) = u4{h.instance | (h.valid ? 0x80000000u : 0u), cur, sp | (inBlas ? 0x10000u : 0u) | (overflow ? 0x20000u : 0u), curInst};
        auto entry = [&](uint32_t e) -> uint32_t {
            if (e >= sp) return 0u;
            if (e >= ldsN) return spill[(e - ldsN) * 64u + (threadIdx.x & 63u)];
            if (s16) { const uint32_t v = stack16[e * 64]; return v == 0xFFFEu ? WALK_SENTINEL : v; } // (refs are below 0xFFFE either way)
            return stack32[e * 64];
        };
        // at least eight entries (zeros beyond the saved depth): start() reads the first eight back with two 16-B loads whatever the
        // depth, and must not be handed words nobody wrote (until round 4 it was: harmless -- it never looks at entries beyond the
        // saved depth -- but an uninitialised read all the same). One entry at a time: eight live values more cost one instrumented
        // variant its last free registers (tests/test_capi.py::test_no_kernel_spills_vector_registers).
        const uint32_t n = sp > 8u ? sp : 8u;
#pragma unroll 1
        for (uint32_t e = 0; e < n; ++e) a[8 + e] = entry(e);
    }
    // A batch of the staged kernel's walk stage holds fresh rays (what begin() does for one ray) and rays taken up again (what save() wrote). Their
    // expensive parts are the same -- the world-space slab ray, and the object-space frames of the instance a ray is
    // in (the one it was set aside in, or the only instance of the scene, which begin enters at once) -- so the lanes of
    // both kinds go through them together instead of one kind after the other.
    THX_DEV void start(f3 o, f3 d, float tmax, bool resumed, const THX_GLOBAL uint32_t* a) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const SceneView& sc = L.a->scene;
        org = o; dir = d;
        // A scene of one instance never walks in world space: its TLAS is one leaf, and the stage that queued the ray has
        // tested the TLAS root box already (dev_staged.hpp: rootHit, the same test with the same distance) -- no world-space
        // slab ray then, and no second root test for any scene.
        const bool oneInstance = FAST && (sc.tlas.rootRef & 7u) != 0u;
        if (!oneInstance) {
            wsr = qRay(tlasFrame0(sc), tlasFrame1(sc), org, dir);
            sr = wsr;
        }
        ao = org;
        uint32_t n = 0;       // stack entries to refill (a ray taken up again)
        bool enter = false;   // fresh ray of a one-instance scene: goes straight into the instance
        // (the first eight saved stack entries come with the ray's state, four independent 16-B loads: as "load an entry, push it"
        // the refill was one memory round trip per entry, 4 000-6 000 cycles in front of nearly every batch)
        u4 e0 = u4{0, 0, 0, 0}, e1 = u4{0, 0, 0, 0};
        if (resumed) {
            const u4 r0 = *(const THX_GLOBAL u4*)(a), r1 = *(const THX_GLOBAL u4*)(a + 4);
            e0 = *(const THX_GLOBAL u4*)(a + 8);
            e1 = *(const THX_GLOBAL u4*)(a + 12);
            h.t = __uint_as_float(r0.x); h.b1 = __uint_as_float(r0.y); h.b2 = __uint_as_float(r0.z); h.prim = r0.w;
            h.instance = r1.x & 0x7FFFFFFFu; h.valid = (r1.x >> 31) != 0u;
            cur = r1.y;
            n = r1.z & 0xFFFFu;
            inBlas = (r1.z & 0x10000u) != 0u; overflow = (r1.z & 0x20000u) != 0u;
            curInst = r1.w;
        } else {
            h.valid = false; h.instance = 0; h.prim = 0; h.t = tmax; h.b1 = 0.0f; h.b2 = 0.0f;
            L.cnt.add(ST_QUERIES);
            inBlas = false; curInst = 0; overflow = false;
            L.cnt.add(ST_NODES);
            cur = (FAST || slabRoot(wsr, h.t)) ? sc.tlas.rootRef : WALK_DONE;
            enter = cur != WALK_DONE && (cur & 7u) != 0u;
            if (enter) {
                curInst = cur >> 3;
                L.cnt.add(ST_INSTANCES);
            }
        }
        sp = 0;
        if (inBlas || enter) {
            const uint32_t root = instanceFrame();
            if (enter) { // what enterInstance does after the frames -- without the sentinel entry: the TLAS of a
                         // one-instance scene is this one leaf, so when the stack runs empty inside the instance the
                         // walk is over (enterInstance's way costs every ray one more pop and the way back to world space)
                inBlas = true;
                L.cnt.add(ST_NODES);
                cur = slabRoot(sr, h.t) ? root : WALK_DONE;
            }
        }
        if (n > 0u) push(e0.x);
        if (n > 1u) push(e0.y);
        if (n > 2u) push(e0.z);
        if (n > 3u) push(e0.w);
        if (n > 4u) push(e1.x);
        if (n > 5u) push(e1.y);
        if (n > 6u) push(e1.z);
        if (n > 7u) push(e1.w);
        for (uint32_t e = 8; e < n; ++e) push(a[8 + e]);
    }

    // One inner-node visit, branchy form: the four cases of (left hit, right hit), a push or a pop through push() / pop()
    // (LDS or the global part of the stack, overflow check). What lanes deeper than the LDS part of their stack run, and
    // every walk outside the staged kernel.
    THX_DEV void visitBranchy() {
        const uint32_t pair = cur >> 3;
        const float4 lc = nodes[2 * pair], rc = nodes[2 * pair + 1];
        L.cnt.add(ST_NODES, 2);
        float tl = slabChild(lc, sr, h.t);
        float tr = slabChild(rc, sr, h.t);
        const uint32_t ml = __float_as_uint(lc.w), mr = __float_as_uint(rc.w);
        if (tl >= 0.0f && tr >= 0.0f) {
            bool leftFirst = tl <= tr;
            push(leftFirst ? mr : ml);
            cur = leftFirst ? ml : mr;
        } else if (tl >= 0.0f) {
            cur = ml;
        } else if (tr >= 0.0f) {
            cur = mr;
        } else {
            cur = pop();
        }
    }
    // One inner-node visit as straight-line code, for a lane whose stack top is inside the LDS part (FAST walkers). With the
    // pair records in LDS a trip is bound by its own chain of dependent instructions -- 64 idle cycles per trip cost configs
    // 2 / 3 +3.7 %, 16 more vector instructions +3.6 % (profiles/r03_inner_loop.txt) -- and the four-way branch of
    // visitBranchy() is five hand-overs from the vector unit to the scalar unit and back (~30 cycles each) plus the LDS round
    // trip of a pop behind them. Here: misses are at +inf, so near / far child, "both" and "none" come out of min / max /
    // compare on the two distances; the far child is written to the slot above the stack top whether or not it is pushed
    // (harmless: the slot is free), and what a pop would return is read together with the node (harmless if it does not
    // pop; the slot below entry 0 holds WALK_DONE). Same visits in the same order as the branchy form.
    THX_DEV void visitStraight() {
        // record (cur >> 3) = two float4 = 32 B (cur's low bits are zero: an inner node)
        const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + (cur << 2));
        const float4 lc = rec[0], rc = rec[1];
        const uint32_t below = V::S16 ? (uint32_t)(int32_t)(int16_t)stack16[(int32_t)(sp << 6) - 64] : stack32[(int32_t)(sp << 6) - 64];
        L.cnt.add(ST_NODES, 2);
        const float tl = slabChildInf(lc, sr, h.t, infReg), tr = slabChildInf(rc, sr, h.t, infReg);
        const uint32_t ml = __float_as_uint(lc.w), mr = __float_as_uint(rc.w);
        const bool leftFirst = tl <= tr;
        const uint32_t nearRef = leftFirst ? ml : mr, farRef = leftFirst ? mr : ml;
        // (raw: the distances are selects between a plane distance and the +inf register, which the compiler would canonicalise first)
        const bool both = maxRaw(tl, tr) < INFINITY, none = minRaw(tl, tr) == INFINITY;
        if (V::S16) stack16[sp << 6] = (uint16_t)farRef;
        else stack32[sp << 6] = farRef;
        sp = sp + (both ? 1u : 0u) - (none ? 1u : 0u);
        cur = none ? below : nearRef;
    }

    // ---- 4-wide records (SceneView::wide): four children per visit, the hit ones nearest first ----
    // Order of two children: by entry distance, equal distances (and misses, all at +inf) by their position in the record.
    // rank = how many of the others come first; the nearest hit child is visited next, the other hit ones go on the stack with
    // rank 1 on top. Both forms below make the same visits in the same order.
    struct Quad {
        float t[4];
        uint32_t r[4], rank[4], nhit;
    };
    THX_DEV Quad quadOrder(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, float t0, float t1, float t2, float t3) {
        Quad q;
        q.t[0] = t0; q.t[1] = t1; q.t[2] = t2; q.t[3] = t3;
        q.r[0] = r0; q.r[1] = r1; q.r[2] = r2; q.r[3] = r3;
        const uint32_t b10 = t1 < t0, b20 = t2 < t0, b30 = t3 < t0, b21 = t2 < t1, b31 = t3 < t1, b32 = t3 < t2;
        q.rank[0] = b10 + b20 + b30;
        q.rank[1] = (1u - b10) + b21 + b31;
        q.rank[2] = (1u - b20) + (1u - b21) + b32;
        q.rank[3] = (1u - b30) + (1u - b31) + (1u - b32);
        q.nhit = (uint32_t)(t0 < INFINITY) + (uint32_t)(t1 < INFINITY) + (uint32_t)(t2 < INFINITY) + (uint32_t)(t3 < INFINITY);
        return q;
    }
    // through push() / pop(): every walk outside the staged kernel, and lanes of it whose next three stack slots are not all in LDS
    THX_DEV void visitBranchy4() {
        const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + ((size_t)cur << 3));
        const float4 c0 = rec[0], c1 = rec[1];
        const float t0 = slabChildFar(c0, sr, h.t), t1 = slabChildFar(c1, sr, h.t);
        asm volatile("" ::: "memory"); // (half a record at a time: sixteen registers of boxes at once are more than the instrumented variants have left)
        const float4 c2 = rec[2], c3 = rec[3];
        const Quad q = quadOrder(__float_as_uint(c0.w), __float_as_uint(c1.w), __float_as_uint(c2.w), __float_as_uint(c3.w), t0, t1,
                                 slabChildFar(c2, sr, h.t), slabChildFar(c3, sr, h.t));
        L.cnt.add(ST_NODES, 2u + (q.r[2] != WALK_DONE ? 1u : 0u) + (q.r[3] != WALK_DONE ? 1u : 0u));
        if (q.nhit == 0u) { cur = pop(); return; }
        for (uint32_t want = q.nhit - 1u; want >= 1u; --want) // the farthest first: rank 1 ends up on top
            for (int i = 0; i < 4; ++i)
                if (q.rank[i] == want) push(q.r[i]);
        for (int i = 0; i < 4; ++i)
            if (q.rank[i] == 0u) cur = q.r[i];
    }
    // straight-line form (staged kernel, the next three stack slots in LDS): every child's ref is written -- a hit child of rank k >= 1 to
    // its slot above the stack top, everything else to a slot nobody reads (the first one beyond the LDS part: the host adds it) --
    // and what a pop would return is read together with the record, as in visitStraight()
    THX_DEV void visitStraight4() {
        const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + ((size_t)cur << 3));
        const uint32_t below = V::S16 ? (uint32_t)(int32_t)(int16_t)stack16[(int32_t)(sp << 6) - 64] : stack32[(int32_t)(sp << 6) - 64];
        const float4 c0 = rec[0], c1 = rec[1];
        const float4 c2 = rec[2], c3 = rec[3];
        const float t0 = slabChildInf(c0, sr, h.t, infReg), t1 = slabChildInf(c1, sr, h.t, infReg);
        const Quad q = quadOrder(__float_as_uint(c0.w), __float_as_uint(c1.w), __float_as_uint(c2.w), __float_as_uint(c3.w), t0, t1,
                                 slabChildInf(c2, sr, h.t, infReg), slabChildInf(c3, sr, h.t, infReg));
        L.cnt.add(ST_NODES, 2u + (q.r[2] != WALK_DONE ? 1u : 0u) + (q.r[3] != WALK_DONE ? 1u : 0u));
        uint32_t nearRef = q.r[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool keep = q.t[i] < INFINITY && q.rank[i] != 0u;
            const uint32_t slot = keep ? sp + q.nhit - 1u - q.rank[i] : ldsN;
            if (V::S16) stack16[slot << 6] = (uint16_t)q.r[i];
            else stack32[slot << 6] = q.r[i];
        }
#pragma unroll
        for (int i = 2; i >= 0; --i) nearRef = q.rank[i] == 0u ? q.r[i] : nearRef;
        cur = q.nhit != 0u ? nearRef : below;
        sp = q.nhit != 0u ? sp + q.nhit - 1u : sp - 1u;
    }

    // one round: inner nodes down to a leaf, then that leaf. No-op for a finished lane.
    THX_DEV void advance() {
#ifdef THX_STAGE_PROFILE
        const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
        pRounds += 1;
#endif
        // open inner nodes until this lane holds a leaf -- or until so few lanes of the wave are still at inner
        // nodes (fewer than THX_INNER_KEEP / 8 of those that started the loop) that the others should not wait any
        // longer: everybody leaves then, lanes holding a leaf process it, the rest go on in the next round
        const uint32_t innerKeep = ((uint32_t)__popcll(__ballot((cur & 7u) == 0u)) * (uint32_t)THX_INNER_KEEP) >> 3;
        // Staged kernel: straight-line visits (visitStraight) while a lane's stack top is inside the LDS part. Variants with
        // the whole stack in LDS run nothing else and do not look at the stack pointer at all: the host picks them only
        // when the stack holds the hierarchy's depth + 2 entries, which a walk cannot exceed (a stack made shallower by the
        // THX_STACK_DEPTH test hook runs the split-stack variants, whose push() flags the overflow). On deeper scenes a lane
        // beyond its LDS part takes the branchy visit for those trips.
        if constexpr (FAST) {
            // (one test per trip: it is the loop's condition and what the early exit counts; ONE way out of the loop -- with a
            // `break` for the early exit next to the per-lane condition the compiler kept two sets of masks up to date, 17 scalar
            // instructions per trip, and a wavefront's trip lasts about as many times 8 cycles as it has instructions of ANY kind)
            bool go = (cur & 7u) == 0u;
            while (go) {
#ifdef THX_STAGE_PROFILE
                pLane += 1;
                pTrips += 1;
#ifndef THX_PROF_LEAN // (the per-trip occupancy counters are two LDS atomics in the middle of the loop: -DTHX_PROF_LEAN times the loop without them)
                {
                    const unsigned long long m = __ballot(1);
                    if (L.prof != nullptr && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) {
                        atomicAdd(&L.prof[31], (unsigned long long)__popcll(m));
                        atomicAdd(&L.prof[39], 1ull);
                    }
                }
#endif
#endif
                if constexpr (V::WIDE) {
                    if (sp + 3u <= ldsN) visitStraight4();
                    else visitBranchy4();
                } else {
                    if (!V::SPILL || sp < ldsN) visitStraight();
                    else visitBranchy();
                }
                const bool atInner = (cur & 7u) == 0u;
                const unsigned long long goMask = __builtin_amdgcn_ballot_w64(atInner); // (the comparison's own mask)
                go = atInner && (THX_INNER_KEEP == 0 || (uint32_t)__popcll(goMask) >= innerKeep);
            }
            if (cur == WALK_DONE) sp = 0; // (popping the sentinel slot took the stack pointer below zero)
            goto innerDone; // (lanes still at an inner node -- early exit -- go on in the next round)
        }
        while ((cur & 7u) == 0u) {
#ifdef THX_STAGE_PROFILE
            pLane += 1;
            pTrips += 1; // lane 0's count is what gets reported: trips of the wave while lane 0 is in the loop
            {
                const unsigned long long m = __ballot(1);
                if (L.prof != nullptr && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) {
                    atomicAdd(&L.prof[31], (unsigned long long)__popcll(m));
                    atomicAdd(&L.prof[39], 1ull);
                }
            }
#endif
            if (wide) visitBranchy4();
            else visitBranchy();
            if (THX_INNER_KEEP > 0 && (uint32_t)__popcll(__ballot((cur & 7u) == 0u)) < innerKeep) break;
        }
    innerDone:
#ifdef THX_STAGE_PROFILE
        const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
        pInner += pt1 - pt0;
        struct Tail { // charges the rest of the round to the triangle or the instance phase
            unsigned long long t1; unsigned long long& leaf; unsigned long long& other; bool tri;
            __device__ ~Tail() { const unsigned long long d = __builtin_amdgcn_s_memtime() - t1; if (tri) leaf += d; else other += d; }
        } tail{pt1, pLeaf, pOther, inBlas && cur != WALK_DONE && cur != WALK_SENTINEL};
#endif
        if (cur == WALK_DONE || (cur & 7u) == 0u) return; // finished, or still at an inner node (early exit above)
        if (cur == WALK_SENTINEL) { // back to world space
            inBlas = false;
            sr = wsr;
            cur = pop();
            return;
        }
        uint32_t first = cur >> 3, count = cur & 7u;
        if (!inBlas) {
            enterInstance(first);
        } else {
            // leaves hold <= 4 triangles; fetch them in pairs before testing so that a
            // leaf costs two memory round trips instead of four (tests stay in order)
            const char* tr = tris + ((size_t)first * sizeof(TriRecord) + w.rot);
#pragma unroll
            for (uint32_t i0 = 0; i0 < 4; i0 += 2) {
                if (i0 >= count) break;
                const bool two = i0 + 1 < count;
                const char* t0 = tr + i0 * sizeof(TriRecord);
                // (the record behind is asked for at once, whether or not it belongs to the leaf -- the array ends with a record of
                // padding: as "copy the first, load the second if the leaf has one" the copies waited for the first record
                // before the second was requested, two round trips again)
                const f3 a0 = loadP3(t0), b0 = loadP3(t0 + 20), c0 = loadP3(t0 + 40);
                const f3 a1 = loadP3(t0 + 64), b1 = loadP3(t0 + 84), c1 = loadP3(t0 + 104);
                L.cnt.add(ST_TRIS, two ? 2u : 1u);
                if (woopTri(w, a0, b0, c0, h.t, h.b1, h.b2)) {
                    h.valid = true; h.instance = curInst; h.prim = first + i0;
                }
                if (two && woopTri(w, a1, b1, c1, h.t, h.b1, h.b2)) {
                    h.valid = true; h.instance = curInst; h.prim = first + i0 + 1u;
                }
            }
            cur = pop();
            // the last leaf of an instance visit: leave the instance here instead of spending the next round on it
            if (cur == WALK_SENTINEL) {
                inBlas = false;
                sr = wsr;
                cur = pop();
            }
        }
    }
};

template <class V>
THX_DEV QueryHit rayQuery(Lane<V>& L, f3 org, f3 dir, float tmax) {
    Walker<V> wk(L);
    wk.begin(org, dir, tmax);
#ifdef THX_STAGE_PROFILE
    // lane occupancy of the walk: every round / inner trip is counted once per wave (by its first
    // active lane) together with the number of lanes that take part in it
    while (!wk.done()) {
        const unsigned long long m = __ballot(1);
        if (L.prof != nullptr && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m)) {
            atomicAdd(&L.prof[29], (unsigned long long)__popcll(m));
            atomicAdd(&L.prof[30], 1ull);
        }
        wk.advance();
    }
#else
    while (!wk.done()) wk.advance();
#endif
    if (wk.h.valid) L.cnt.add(ST_HITS);
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

} // namespace thx
