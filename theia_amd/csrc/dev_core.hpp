// Device-side building blocks of the photon path: vector math, Philox4x32-10,
// table look-ups, medium constants, phase-function sampling, Mueller/Stokes
// updates. Written for gfx950 (wave64); every function cites the reference
// shader it restates (paths relative to /root/reference/src/theia/shader).
//
// Arithmetic conventions (the GLSL spec leaves them open; fixed so that the
// kernels and the CPU oracle round identically -- build with -ffp-contract=off):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))
//   mat3 * v      = (c0*v.x + c1*v.y) + c2*v.z
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/theia_hip.h"
#include "thx_layout.h"

namespace thx {

#define THX_DEV __device__ __forceinline__

struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};
struct m3 { // columns
    f3 c0, c1, c2;
};

THX_DEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
THX_DEV m3 transpose(m3 m) { return m3{f3{m.c0.x, m.c1.x, m.c2.x}, f3{m.c0.y, m.c1.y, m.c2.y}, f3{m.c0.z, m.c1.z, m.c2.z}}; }
THX_DEV f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
// Pointers made from the 64-bit device addresses users hand in (medium / material / geometry
// records, tables) carry the global address space: plain pointers would compile to FLAT loads,
// which also tick the LDS counter and are waited for together with every LDS access.
#define THX_GLOBAL __attribute__((address_space(1)))
THX_DEV f3 ld3(const THX_GLOBAL float* p) { return f3{p[0], p[1], p[2]}; }
// Parameter blocks that only rare code paths read (plane guides, host / cone light sources, function
// wavelengths) are read through a laundered constant-address-space pointer: the scalar loads then
// stay inside the branch that needs them. Left to itself the compiler hoists every kernel-argument
// load to the top of the staged kernel, where the values live in SGPRs for its whole run -- the
// kernel is at the SGPR limit, so each extra block spilled (measured: +100 SGPR spills, 36 B of
// scratch per lane, config 2 +3 %).
#define THX_KERNARG __attribute__((address_space(4)))
// start of the kernel-argument segment = the kernel's first parameter (every kernel here takes its
// LaunchArgs first), laundered so that loads through it are not hoisted out of the calling branch
THX_DEV const THX_KERNARG unsigned char* coldKernarg() {
    const THX_KERNARG unsigned char* p = (const THX_KERNARG unsigned char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
THX_DEV f3 ld3(const THX_KERNARG float* p) { return f3{p[0], p[1], p[2]}; }
THX_DEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
THX_DEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
THX_DEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
THX_DEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
THX_DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
THX_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
THX_DEV float distance(f3 a, f3 b) { return length(a - b); }
THX_DEV f3 normalize(f3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
THX_DEV f3 cross(f3 a, f3 b) { // GLSL cross()
    return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
THX_DEV f3 mul(const m3& m, f3 v) { return (m.c0 * v.x + m.c1 * v.y) + m.c2 * v.z; }
THX_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
THX_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
THX_DEV float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

// Transcendentals. Default: the accurate ROCm device-library versions (<= 1-2 ulp), which
// keeps the kernels comparable to the CPU oracle at the libm level. -DTHX_NATIVE_MATH
// switches to the hardware approximations (v_sin/v_cos/v_exp/v_log based, what GLSL
// drivers typically emit for the reference's shaders).
// sine and cosine of one angle: the device library's sinf and cosf each compute both (one argument reduction, the pair of
// polynomials, a selection by quadrant); sincosf does that once and returns the same two values
#ifdef THX_NATIVE_MATH
THX_DEV void thx_sincos(float x, float& s, float& c) { s = __sinf(x); c = __cosf(x); }
#else
THX_DEV void thx_sincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
#endif
#ifdef THX_NATIVE_MATH
THX_DEV float thx_sin(float x) { return __sinf(x); }
THX_DEV float thx_cos(float x) { return __cosf(x); }
THX_DEV float thx_exp(float x) { return __expf(x); }
THX_DEV float thx_log(float x) { return __logf(x); }
THX_DEV float thx_pow(float x, float y) { return __powf(x, y); }
#else
THX_DEV float thx_sin(float x) { return sinf(x); }
THX_DEV float thx_cos(float x) { return cosf(x); }
THX_DEV float thx_exp(float x) { return expf(x); }
THX_DEV float thx_log(float x) { return logf(x); }
THX_DEV float thx_pow(float x, float y) { return powf(x, y); }
#endif

// math.glsl:8-16, material.glsl:6
constexpr float TWO_PI = 6.283185307179586477f;
constexpr float INV_4PI = 0.0795774715459476679f;
constexpr float SPEED_OF_LIGHT = 0.299792458f;
constexpr float INV_SPEED_OF_LIGHT = 3.335640951981521f; // material.glsl:7, ns / m

// result.glsl:10-28
enum : int {
    RESULT_CODE_SUCCESS = 0,
    RESULT_CODE_RAY_CREATED = 1,
    RESULT_CODE_RAY_SCATTERED = 2,
    RESULT_CODE_RAY_HIT = 3,
    RESULT_CODE_RAY_DETECTED = 4,
    RESULT_CODE_VOLUME_HIT = 5,
    RESULT_CODE_RAY_LOST = -1,
    RESULT_CODE_RAY_DECAYED = -2,
    RESULT_CODE_RAY_ABSORBED = -3,
    RESULT_CODE_RAY_MISSED = -4,
    RESULT_CODE_MAX_ITER = -5,
    ERROR_CODE_MAX_VALUE = -10,
    ERROR_CODE_MEDIA_MISMATCH = -11,
    ERROR_CODE_TRACE_ABORT = -12,
    ERROR_CODE_RAY_BAD = -13
};

// material.glsl:79-86
constexpr uint32_t MATERIAL_BLACK_BODY_BIT = 0x01u;
constexpr uint32_t MATERIAL_DETECTOR_BIT = 0x02u;
constexpr uint32_t MATERIAL_NO_REFLECT_BWD_BIT = 0x10u;
constexpr uint32_t MATERIAL_NO_TRANSMIT_BWD_BIT = 0x40u;
constexpr uint32_t MATERIAL_VOLUME_BORDER_BIT = 0x80u;

// counter slots (theia_hip.h: THX_N_STATS)
enum : int {
    ST_CREATED, ST_SCATTERED, ST_HIT, ST_DETECTED, ST_VOLUME, ST_LOST, ST_DECAYED, ST_ABSORBED,
    ST_MISSED, ST_MAXITER, ST_ERROR, ST_MISMATCH,
    ST_QUERIES, ST_NODES, ST_TRIS, ST_INSTANCES, ST_HITS, ST_LUT, ST_RESPONSES, ST_STEPS
};

// Per-lane counters of the instrumented build; empty otherwise.
template <bool ON>
struct Counters {
    uint32_t v[THX_N_STATS];
    THX_DEV void clear() {
#pragma unroll
        for (int i = 0; i < THX_N_STATS; ++i) v[i] = 0;
    }
    THX_DEV void add(int slot, uint32_t n = 1) { v[slot] += n; }
};
template <>
struct Counters<false> {
    THX_DEV void clear() {}
    THX_DEV void add(int, uint32_t = 1) {}
};

// ---------------------------------------------------------------- math.glsl

// math.glsl:18-20
THX_DEV float signBit(float f) {
    return __uint_as_float((__float_as_uint(f) & 0x80000000u) | 0x3F800000u);
}
// math.glsl:24-29 (`precise`: explicit fma, no contraction elsewhere)
THX_DEV float prodDiff(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = __fmaf_rn(a, b, -cd);
    float err = __fmaf_rn(-c, d, cd);
    return diff + err;
}
// math.glsl:33-39
THX_DEV f3 crosser(f3 v, f3 w) {
    return f3{prodDiff(v.y, w.z, v.z, w.y), prodDiff(v.z, w.x, v.x, w.z),
              prodDiff(v.x, w.y, v.y, w.x)};
}
// math.glsl:45-55
THX_DEV m3 createLocalCOSY(f3 vz) {
    float s = signBit(vz.z);
    float a = -1.0f / (s + vz.z);
    float b = vz.x * vz.y * a;
    f3 vx = f3{1.0f + s * vz.x * vz.x * a, s * b, -s * vz.x};
    f3 vy = f3{b, s + vz.y * vz.y * a, -vz.y};
    return m3{normalize(vx), normalize(vy), vz};
}
// math.glsl:58-77
THX_DEV f3 perpendicularTo(f3 v) {
    float s = signBit(v.z);
    float a = -1.0f / (s + v.z);
    float b = v.x * v.y * a;
    return f3{b, s + v.y * v.y * a, -v.y};
}
THX_DEV f3 perpendicularTo(f3 a, f3 b) {
    f3 c = crosser(a, b);
    float len = length(c);
    if (len < 1e-5f) return perpendicularTo(a);
    return f3{c.x / len, c.y / len, c.z / len};
}

// ------------------------------------------------- random.philox.glsl / util

// random.philox.glsl:15-52. counter = baseCount + (i << 2) with carry chain
// and roll-over, key = key + stream; ten rounds.
THX_DEV void philoxCounter(const thx_philox_params& p, uint32_t i, uint32_t& sx, uint32_t& sy, uint32_t& sz, uint32_t& sw) {
    uint32_t carry = i << 2;
    sx = p.offset[0]; sy = p.offset[1]; sz = p.offset[2]; sw = p.offset[3];
    uint32_t t;
    t = sx + carry; carry = t < sx ? 1u : 0u; sx = t;
    t = sy + carry; carry = t < sy ? 1u : 0u; sy = t;
    t = sz + carry; carry = t < sz ? 1u : 0u; sz = t;
    t = sw + carry; carry = t < sw ? 1u : 0u; sw = t;
    sx += carry;
}
THX_DEV void philoxKey(const thx_philox_params& p, uint32_t stream, uint32_t& kx, uint32_t& ky) {
    kx = p.key[0]; ky = p.key[1];
    uint32_t t, carry;
    t = kx + stream; carry = t < kx ? 1u : 0u; kx = t;
    t = ky + carry; carry = t < ky ? 1u : 0u; ky = t;
    kx += carry;
    // (the key words are final here: without this the compiler carries the carry flag through all ten rounds and
    // spends an add-with-carry and a move on every key bump instead of one add)
    asm volatile("" : "+v"(kx), "+v"(ky));
}
constexpr uint32_t kPhiloxM0 = 0xD2511F53u, kPhiloxM1 = 0xCD9E8D57u;
// one round on one block: one 32x32->64 multiply (v_mad_u64_u32) per product instead of mul_hi + mul_lo
THX_DEV void philoxRound(uint32_t& sx, uint32_t& sy, uint32_t& sz, uint32_t& sw, uint32_t kx, uint32_t ky) {
    const uint64_t p0 = (uint64_t)kPhiloxM0 * (uint64_t)sx;
    const uint64_t p1 = (uint64_t)kPhiloxM1 * (uint64_t)sz;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t nx = hi1 ^ sy ^ kx, nz = hi0 ^ sw ^ ky;
    sx = nx; sy = lo1; sz = nz; sw = lo0;
}
THX_DEV void philox4x32(const thx_philox_params& p, uint32_t stream, uint32_t i, uint32_t out[4]) {
    uint32_t sx, sy, sz, sw, kx, ky;
    philoxCounter(p, i, sx, sy, sz, sw);
    philoxKey(p, stream, kx, ky);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philoxRound(sx, sy, sz, sw, kx, ky);
        kx += 0x9E3779B9u;
        ky += 0xBB67AE85u;
    }
    out[0] = sx; out[1] = sy; out[2] = sz; out[3] = sw;
}
// the blocks of dimensions i and i + 1 of one stream together (random2D): one key, one key schedule, two counters
THX_DEV void philox4x32_pair(const thx_philox_params& p, uint32_t stream, uint32_t i, uint32_t a[4], uint32_t b[4]) {
    uint32_t ax, ay, az, aw, bx, by, bz, bw, kx, ky;
    philoxCounter(p, i, ax, ay, az, aw);
    philoxCounter(p, i + 1u, bx, by, bz, bw);
    philoxKey(p, stream, kx, ky);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philoxRound(ax, ay, az, aw, kx, ky);
        philoxRound(bx, by, bz, bw, kx, ky);
        kx += 0x9E3779B9u;
        ky += 0xBB67AE85u;
    }
    a[0] = ax; a[1] = ay; a[2] = az; a[3] = aw;
    b[0] = bx; b[1] = by; b[2] = bz; b[3] = bw;
}

// random.philox.glsl:54-77: the cached block is never reused by the reference
// (philoxBufferCount stays 0), so random_s(stream,i) == philox(stream,i)[i & 3].
THX_DEV uint32_t philoxWord(const uint32_t s[4], uint32_t i) {
    const uint32_t l = i & 3u;
    return l == 0 ? s[0] : (l == 1 ? s[1] : (l == 2 ? s[2] : s[3]));
}
THX_DEV uint32_t random_u32(const thx_philox_params& p, uint32_t stream, uint32_t i) {
    uint32_t s[4];
    philox4x32(p, stream, i, s);
    return philoxWord(s, i);
}
// random.util.glsl:4-9 (uint->float conversion: round to nearest even)
THX_DEV float normalizeUint(uint32_t i) {
    return fminf(__uint_as_float(0x3F7FFFFFu), __uint2float_rn(i) * __uint_as_float(0x2F800000u));
}
// random.philox.glsl:84-94
THX_DEV float rnd(const thx_philox_params& p, uint32_t stream, uint32_t& dim) {
    float r = normalizeUint(random_u32(p, stream, dim));
    dim += 1;
    return r;
}
// two consecutive draws (random.philox.glsl:79-82 random2D): the same two numbers as two calls of rnd()
THX_DEV void rnd2(const thx_philox_params& p, uint32_t stream, uint32_t& dim, float& u0, float& u1) {
    uint32_t a[4], b[4];
    philox4x32_pair(p, stream, dim, a, b);
    u0 = normalizeUint(philoxWord(a, dim));
    u1 = normalizeUint(philoxWord(b, dim + 1u));
    dim += 2;
}

// random.sobol.glsl: shuffled, Owen-scrambled Sobol sampler (Burley 2020), padded to higher
// dimensions: every draw re-scrambles the stream index with a seed derived from the dimension
THX_DEV uint32_t hp_mix32(uint32_t n) { // :92-103
    n ^= n >> 16; n *= 0x21F0AAADu; n ^= n >> 15; n *= 0xD35A2D97u; n ^= n >> 15;
    return n ^ 0xE6FE3BEBu;
}
THX_DEV uint32_t lk_hash(uint32_t n, uint32_t seed) { // :111-119
    n ^= n * 0x3D20ADEAu; n += seed; n *= (seed >> 16) | 1u; n ^= n * 0x05526C56u; n ^= n * 0x53A22864u;
    return n;
}
THX_DEV uint32_t combineHash(uint32_t seed, uint32_t v) { return seed ^ (v + 0x9E3779B9u + (seed << 6) + (seed >> 2)); } // :122-124
// :129-148; the direction numbers of dimension 1 (the table at :34-43) obey v[k+1] = v[k] ^ (v[k] >> 1)
THX_DEV uint32_t sobolBits(uint32_t index, uint32_t dim) {
    if (dim == 0) return __brev(index);
    uint32_t result = 0, v = 0x80000000u;
    for (; index != 0; index >>= 1, v ^= v >> 1)
        if (index & 1u) result ^= v;
    return result;
}
THX_DEV uint32_t nestedScramble(uint32_t v, uint32_t seed) { return __brev(lk_hash(__brev(v), seed)); } // :150-155
THX_DEV float scrambledSobol(uint32_t scrambled_index, uint32_t dim, uint32_t seed) { // :156-161
    return normalizeUint(nestedScramble(sobolBits(scrambled_index, dim), combineHash(seed, dim)));
}
// random.sobol.glsl:165-176 / :178-193
THX_DEV float rndSobol(uint32_t baseSeed, uint32_t stream, uint32_t& dim) {
    const uint32_t seed = combineHash(baseSeed, hp_mix32(dim));
    dim += 1;
    return scrambledSobol(nestedScramble(stream, seed), 0, seed);
}
THX_DEV void rndSobol2(uint32_t baseSeed, uint32_t stream, uint32_t& dim, float& u0, float& u1) {
    const uint32_t seed = combineHash(baseSeed, hp_mix32(dim));
    const uint32_t index = nestedScramble(stream, seed);
    dim += 2;
    u0 = scrambledSobol(index, 0, seed);
    u1 = scrambledSobol(index, 1, seed);
}

// ------------------------------------------------- lookup.glsl / material.glsl

struct MediumConstants {
    float n, vg, mu_s, mu_e;
};

// lookup.glsl:14-28
template <bool ST>
THX_DEV float lookUp(uint64_t table, float u, float nullValue, Counters<ST>& cnt) {
    if (table == 0) return nullValue;
    const THX_GLOBAL int32_t* hdr = reinterpret_cast<const THX_GLOBAL int32_t*>(table);
    const THX_GLOBAL float* samples = reinterpret_cast<const THX_GLOBAL float*>(hdr + 1);
    cnt.add(ST_LUT);
    u = clampf(u, 0.0f, 1.0f);
    u *= (float)(hdr[0] - 1);
    float fl = floorf(u);
    int lo = (int)fl;
    int hi = (int)ceilf(u);
    float l = u - fl;
    return samples[lo] * (1.0f - l) + samples[hi] * l;
}

// The same look-up cut into its dependent load levels (round 3). lookUp() above is three round trips to memory in a
// row -- table pointer, header, samples -- each behind a branch on "no table", and a medium look-up was four of those one
// after the other (the compiler cannot move a load across the branch that guards it): twelve exposed round trips at the
// birth of a photon, six per scatter, twelve more per polarized scatter, at 300-600 cycles each. Here a null table reads
// a dummy region instead of branching (its result is selected away), so the loads of one level of ALL the tables a
// function needs are independent instructions in straight-line code and go out together: three round trips for a medium
// look-up, four for a scatter (two tables, the second indexed by the first's result). Same arithmetic on the same values:
// results are bit for bit those of lookUp().
static __device__ uint32_t thx_null_region[32]; // what the loads of a "null" table / the vacuum medium read (never used)
struct TableRef {
    const THX_GLOBAL float* samples; // samples[0]
    float nm1;                       // float(n - 1); 0 for a null table (index 0 then)
    bool null_;
};
// level 2: the header of the table at address `table` (0: null)
THX_DEV TableRef tableRef(uint64_t table) {
    const bool isNull = table == 0;
    const THX_GLOBAL int32_t* hdr = reinterpret_cast<const THX_GLOBAL int32_t*>(isNull ? (uint64_t)(uintptr_t)thx_null_region : table);
    const int32_t n = hdr[0];
    return TableRef{reinterpret_cast<const THX_GLOBAL float*>(hdr + 1), isNull ? 0.0f : (float)(n - 1), isNull};
}
// level 3: lookup.glsl:14-28 on a table whose header is known
template <bool ST>
THX_DEV float lookUpRef(const TableRef& t, float u, float nullValue, Counters<ST>& cnt) {
    cnt.add(ST_LUT, t.null_ ? 0u : 1u);
    u = clampf(u, 0.0f, 1.0f); // (also turns a NaN from the dummy medium's wavelength range into 0)
    u *= t.nm1;
    float fl = floorf(u);
    int lo = (int)fl;
    int hi = (int)ceilf(u);
    float l = u - fl;
    const float v = t.samples[lo] * (1.0f - l) + t.samples[hi] * l;
    return t.null_ ? nullValue : v;
}
// the medium record a pointer stands for; vacuum (0) reads the dummy region and reports every table as null
THX_DEV const THX_GLOBAL MediumGLSL* mediumRecord(uint64_t medium) {
    return reinterpret_cast<const THX_GLOBAL MediumGLSL*>(medium == 0 ? (uint64_t)(uintptr_t)thx_null_region : medium);
}

// material.glsl:37-39
THX_DEV float normalize_lambda(const THX_GLOBAL MediumGLSL* m, float lambda) {
    return clampf((lambda - m->lambda_min) / (m->lambda_max - m->lambda_min), 0.0f, 1.0f);
}

// material.glsl:50-74 (vacuum = the null medium: n = 1, vg = c, no scattering, no absorption -- what four null tables give)
template <bool ST>
THX_DEV MediumConstants lookUpMedium(uint64_t medium, float lambda, Counters<ST>& cnt) {
    const bool vac = medium == 0;
    const THX_GLOBAL MediumGLSL* m = mediumRecord(medium);
    const float u = normalize_lambda(m, lambda);
    const uint64_t ta = m->mu_a, ts = m->mu_s, tn = m->n, tv = m->vg;                                     // level 1
    const TableRef ra = tableRef(vac ? 0 : ta), rs = tableRef(vac ? 0 : ts), rn = tableRef(vac ? 0 : tn), rv = tableRef(vac ? 0 : tv); // level 2
    const float mu_a = lookUpRef(ra, u, 0.0f, cnt);                                                        // level 3
    const float mu_s = lookUpRef(rs, u, 0.0f, cnt);
    MediumConstants c;
    c.mu_e = mu_a + mu_s;
    c.n = lookUpRef(rn, u, 1.0f, cnt);
    c.vg = lookUpRef(rv, u, SPEED_OF_LIGHT, cnt);
    c.mu_s = mu_s;
    return c;
}

// --------------------------------------------------------- polarization.glsl

// frame-rotation Mueller matrix (polarization.glsl:29-33, column major) times a
// Stokes vector
THX_DEV f4 rotApply(float c, float s, f4 v) {
    return f4{v.x, c * v.y + (-s) * v.z, s * v.y + c * v.z, v.w};
}

// polarization.glsl:38-68 -> (cos 2phi, sin 2phi, new reference)
THX_DEV void rotatePolRef(f3 dir, f3 ref, f3 nw, f3& new_ref, float& c, float& s) {
    f3 nr = crosser(dir, nw);
    float len = length(nr);
    if (len > 1.0e-7f) {
        nr = f3{nr.x / len, nr.y / len, nr.z / len};
    } else {
        new_ref = ref;
        c = 1.0f;
        s = 0.0f;
        return;
    }
    float cos_phi = dot(ref, nr);
    float sin_phi = dot(crosser(ref, nr), dir);
    c = 2.0f * cos_phi * cos_phi - 1.0f;
    s = 2.0f * cos_phi * sin_phi;
    new_ref = nr;
}

// polarization.glsl:21-34
THX_DEV void alignPolRef(f3 rayDir, f3 oldRef, f3 newRef, float& c, float& s) {
    float cos_phi = dot(oldRef, newRef);
    float sin_phi = dot(crosser(oldRef, newRef), rayDir);
    c = 2.0f * cos_phi * cos_phi - 1.0f;
    s = 2.0f * cos_phi * sin_phi;
}

// polarization.glsl:109-120 times a Stokes vector
THX_DEV f4 polarizerApply(float p, float s, f4 v) {
    float att = p * p + s * s;
    float m12 = (p * p - s * s) / att;
    float m33 = (2.0f * p * s) / att;
    return f4{v.x + m12 * v.y, m12 * v.x + v.y, m33 * v.z, m33 * v.w};
}

// ray.scatter.glsl:46-54: (phase * rotate) * stokes with the phase matrix of
// polarization.glsl:88-107. This 4x4 contraction is the one spot the north star
// earmarks for MFMA. With the block-sparse structure it is 14 multiplies per photon on
// the VALU (the default). -DTHX_MUELLER_MFMA builds the measured alternative: the same
// contraction on the matrix pipe (profiles/r02_mueller_mfma.txt has both timings).
#ifdef THX_MUELLER_MFMA
typedef float v4f __attribute__((ext_vector_type(4)));
// out[r] = sum_c M[r][c] * v[c] per LANE with v_mfma_f32_4x4x1f32 (16 blocks of 4 lanes): in a block,
// D[i][j] += A(lane i) * B(lane j); with A = this lane's M[r][c] and B = this lane's v[c] the lane's own
// product sits on the diagonal, i.e. in element (lane & 3) of the lane's accumulator. No data moves between
// lanes; three quarters of the block's products are other lanes' cross terms and are thrown away.
THX_DEV float mfmaDiag(v4f acc) {
    const uint32_t k = threadIdx.x & 3u;
    return k == 0u ? acc.x : (k == 1u ? acc.y : (k == 2u ? acc.z : acc.w));
}
#endif
template <bool ST>
THX_DEV f4 phaseRotateApply(uint64_t medium, float cos_theta, float c, float s, f4 v,
                            Counters<ST>& cnt) {
    // (vacuum: the identity; a medium without one of the tables: 0 for that element, as lookUp() gives)
    const bool vac = medium == 0;
    const THX_GLOBAL MediumGLSL* m = mediumRecord(medium);
    const uint64_t t12 = m->phase_m12, t22 = m->phase_m22, t33 = m->phase_m33, t34 = m->phase_m34;                                  // level 1
    const TableRef r12 = tableRef(vac ? 0 : t12), r22 = tableRef(vac ? 0 : t22), r33 = tableRef(vac ? 0 : t33), r34 = tableRef(vac ? 0 : t34); // level 2
    const float t = 0.5f * (cos_theta + 1.0f);
    const float m12 = lookUpRef(r12, t, 0.0f, cnt);                                                                                   // level 3
    const float m22 = vac ? 1.0f : lookUpRef(r22, t, 0.0f, cnt);
    const float m33 = vac ? 1.0f : lookUpRef(r33, t, 0.0f, cnt);
    const float m34 = lookUpRef(r34, t, 0.0f, cnt);
    f4 o;
#ifdef THX_MUELLER_MFMA
    // M = phase * rotate, row by row (zero entries skipped: rows 0, 1 have no column 3, rows 2, 3 no column 0);
    // the MFMA fuses multiply and add, so the last bits differ from the VALU form below (and from the oracle)
    const v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
    v4f a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, v.x, z, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m12 * c, v.y, a0, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m12 * (-s), v.z, a0, 0, 0, 0);
    v4f a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m12, v.x, z, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m22 * c, v.y, a1, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m22 * (-s), v.z, a1, 0, 0, 0);
    v4f a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(m33 * s, v.y, z, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(m33 * c, v.z, a2, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(m34, v.w, a2, 0, 0, 0);
    v4f a3 = __builtin_amdgcn_mfma_f32_4x4x1f32((-m34) * s, v.y, z, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32((-m34) * c, v.z, a3, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(m33, v.w, a3, 0, 0, 0);
    o.x = mfmaDiag(a0); o.y = mfmaDiag(a1); o.z = mfmaDiag(a2); o.w = mfmaDiag(a3);
#else
    o.x = (v.x + (m12 * c) * v.y) + (m12 * (-s)) * v.z;
    o.y = (m12 * v.x + (m22 * c) * v.y) + (m22 * (-s)) * v.z;
    o.z = ((m33 * s) * v.y + (m33 * c) * v.z) + m34 * v.w;
    o.w = (((-m34) * s) * v.y + ((-m34) * c) * v.z) + m33 * v.w;
#endif
    return o;
}

// --------------------------------------------------------- scatter.volume.glsl

// scatter.volume.glsl:7-26
THX_DEV f3 scatterDir(f3 prevDir, float cos_theta, float phi) {
    prevDir = normalize(prevDir);
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    float sp, cp;
    thx_sincos(phi, sp, cp);
    f3 local = f3{sin_theta * cp, sin_theta * sp, cos_theta};
    local = normalize(local);
    m3 trafo = createLocalCOSY(prevDir);
    return normalize(mul(trafo, local));
}

// The two tables a scatter reads (inverse CDF of cos theta, log phase function), pointers and headers loaded once per stage
// and shared by sampleScatterDir() and scatterProb() (the MIS stage evaluates the phase function twice).
struct ScatterTabs {
    TableRef sampling, logPhase;
    bool canSample, hasPhase; // medium != 0 && phase_sampling != 0; medium != 0 && log_phase != 0
};
THX_DEV ScatterTabs scatterTabs(uint64_t medium) {
    const bool vac = medium == 0;
    const THX_GLOBAL MediumGLSL* m = mediumRecord(medium);
    const uint64_t ts = m->phase_sampling, tl = m->log_phase; // level 1
    ScatterTabs t;
    t.canSample = !vac && ts != 0;
    t.hasPhase = !vac && tl != 0;
    t.sampling = tableRef(t.canSample ? ts : 0);             // level 2
    t.logPhase = tableRef(t.hasPhase ? tl : 0);
    return t;
}

// scatter.volume.glsl:28-45
template <bool ST>
THX_DEV float sampleScatterDir(const ScatterTabs& t, float u0, float u1, float& cos_theta, float& phi, Counters<ST>& cnt) {
    phi = u0 * TWO_PI;
    float ct = lookUpRef(t.sampling, u1, 0.0f, cnt);
    ct = clampf(ct, -1.0f, 1.0f);
    // (a medium with a sampling table but no phase table: lookUp() of the null table gives 0, exp(0) = 1; without a sampling
    // table the phase table is not read)
    TableRef lp = t.logPhase;
    lp.null_ = lp.null_ || !t.canSample;
    const float p = thx_exp(lookUpRef(lp, 0.5f * (ct + 1.0f), 0.0f, cnt));
    cos_theta = t.canSample ? ct : 2.0f * u1 - 1.0f;
    return t.canSample ? p : INV_4PI;
}
template <bool ST>
THX_DEV float sampleScatterDir(uint64_t medium, float u0, float u1, float& cos_theta, float& phi, Counters<ST>& cnt) {
    return sampleScatterDir(scatterTabs(medium), u0, u1, cos_theta, phi, cnt);
}

// scatter.volume.glsl:47-52
template <bool ST>
THX_DEV f3 scatter(const ScatterTabs& t, f3 inDir, float u0, float u1, float& p, Counters<ST>& cnt) {
    float cos_theta, phi;
    p = sampleScatterDir(t, u0, u1, cos_theta, phi, cnt);
    return scatterDir(inDir, cos_theta, phi);
}
template <bool ST>
THX_DEV f3 scatter(uint64_t medium, f3 inDir, float u0, float u1, float& p, Counters<ST>& cnt) {
    return scatter(scatterTabs(medium), inDir, u0, u1, p, cnt);
}

// scatter.volume.glsl:54-65
template <bool ST>
THX_DEV float scatterProb(const ScatterTabs& t, f3 inDir, f3 outDir, Counters<ST>& cnt) {
    float cos_theta = dot(inDir, outDir);
    float log_p = lookUpRef(t.logPhase, 0.5f * (cos_theta + 1.0f), 0.0f, cnt);
    return t.hasPhase ? thx_exp(log_p) : INV_4PI;
}
template <bool ST>
THX_DEV float scatterProb(uint64_t medium, f3 inDir, f3 outDir, Counters<ST>& cnt) {
    return scatterProb(scatterTabs(medium), inDir, outDir, cnt);
}

// util.sample.glsl:10-17
THX_DEV f3 sphericalToCartessian(float phi, float cos_theta) {
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    float sp, cp;
    thx_sincos(phi, sp, cp);
    return f3{sin_theta * sp, sin_theta * cp, cos_theta};
}

} // namespace thx
