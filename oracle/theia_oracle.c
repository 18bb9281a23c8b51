/*
 * theia_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * Plain-C restatement of the inner loop of theia's SceneForwardTracer and
 * VolumeForwardTracer. Every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/theia/shader unless stated).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 * -ffp-contract=off + explicit fmaf() honours the GLSL `precise` spots
 * (SURVEY.md 3.5); all arithmetic is fp32, RNG is uint32.
 *
 * Conventions fixed here because GLSL leaves them to the driver (and mirrored
 * by the HIP kernels so that both sides round identically):
 *   normalize(v) = v * (1/sqrt(dot(v,v)))     dot = ((x*x)+(y*y))+(z*z)
 *   mat*vec      = ((col0*v.x + col1*v.y) + col2*v.z) [+ col3*v.w]
 *   uint->float  = round-to-nearest-even (random.util.glsl:8-9)
 * The ray query (VK_KHR_ray_query in the reference; not part of its sources)
 * is restated as: closest hit, opaque, no culling, 0 < t < tmax, watertight
 * ray/triangle test (Woop, Benthin, Wald 2013) in object space.
 */
#include "theia_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* small vector helpers                                                     */
/* ------------------------------------------------------------------------ */

typedef struct { float x, y, z; } vec3;
typedef struct { float x, y, z, w; } vec4;
typedef struct { vec3 c0, c1, c2; } mat3; /* columns */

static inline vec3 V3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
static inline vec3 v_add(vec3 a, vec3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v_sub(vec3 a, vec3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v_scale(vec3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v_neg(vec3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v_dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v_length(vec3 a) { return sqrtf(v_dot(a, a)); }
static inline float v_distance(vec3 a, vec3 b) { return v_length(v_sub(a, b)); }
static inline vec3 v_normalize(vec3 a) { return v_scale(a, 1.0f / sqrtf(v_dot(a, a))); }
/* GLSL cross() */
static inline vec3 v_cross(vec3 a, vec3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline vec3 m3_mul(mat3 m, vec3 v) {
    return v_add(v_add(v_scale(m.c0, v.x), v_scale(m.c1, v.y)), v_scale(m.c2, v.z));
}
static inline mat3 m3_transpose(mat3 m) {
    mat3 t;
    t.c0.x = m.c0.x; t.c0.y = m.c1.x; t.c0.z = m.c2.x;
    t.c1.x = m.c0.y; t.c1.y = m.c1.y; t.c1.z = m.c2.y;
    t.c2.x = m.c0.z; t.c2.y = m.c1.z; t.c2.z = m.c2.z;
    return t;
}
static inline float f_clamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float f_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline float get3(vec3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int32_t f2i(float f) { int32_t u; memcpy(&u, &f, 4); return u; }
static inline float i2f(int32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* math.glsl:8-16 */
#define PI_F 3.141592653589793238f
#define TWO_PI_F 6.283185307179586477f
#define INV_4PI_F 0.0795774715459476679f
/* material.glsl:6 */
#define SPEED_OF_LIGHT_F 0.299792458f

/* result.glsl:10-28 */
enum {
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
    ERROR_CODE_UNKNOWN = -10,
    ERROR_CODE_MEDIA_MISMATCH = -11,
    ERROR_CODE_TRACE_ABORT = -12,
    ERROR_CODE_RAY_BAD = -13
};

/* material.glsl:79-86 */
#define MATERIAL_BLACK_BODY_BIT 0x01u
#define MATERIAL_DETECTOR_BIT 0x02u
#define MATERIAL_NO_REFLECT_BWD_BIT 0x10u
#define MATERIAL_NO_TRANSMIT_BWD_BIT 0x40u
#define MATERIAL_VOLUME_BORDER_BIT 0x80u

/* stats slots: callback.stat.glsl:7-20, then traversal counters */
enum {
    ST_CREATED, ST_SCATTERED, ST_HIT, ST_DETECTED, ST_VOLUME, ST_LOST, ST_DECAYED,
    ST_ABSORBED, ST_MISSED, ST_MAXITER, ST_ERROR, ST_MISMATCH,
    ST_QUERIES, ST_NODES, ST_TRIS, ST_INSTANCES, ST_HITS, ST_LUT, ST_RESPONSES, ST_STEPS,
    ST_COUNT
};

/* ------------------------------------------------------------------------ */
/* math.glsl                                                                */
/* ------------------------------------------------------------------------ */

/* math.glsl:18-20 */
static inline float signBit(float f) { return u2f((f2u(f) & 0x80000000u) | 0x3F800000u); }
float orc_sign_bit(float f) { return signBit(f); }

/* math.glsl:24-29 (Kahan ab-cd; `precise`) */
static inline float prodDiff(float a, float b, float c, float d) {
    float cd = c * d;
    float diff = fmaf(a, b, -cd);
    float err = fmaf(-c, d, cd);
    return diff + err;
}
/* math.glsl:33-39 */
static inline vec3 crosser(vec3 v, vec3 w) {
    return V3(prodDiff(v.y, w.z, v.z, w.y), prodDiff(v.z, w.x, v.x, w.z),
              prodDiff(v.x, w.y, v.y, w.x));
}
/* math.glsl:45-55 */
static inline mat3 createLocalCOSY(vec3 vz) {
    float s = signBit(vz.z);
    float a = -1.0f / (s + vz.z);
    float b = vz.x * vz.y * a;
    vec3 vx = V3(1.0f + s * vz.x * vz.x * a, s * b, -s * vz.x);
    vec3 vy = V3(b, s + vz.y * vz.y * a, -vz.y);
    mat3 m;
    m.c0 = v_normalize(vx);
    m.c1 = v_normalize(vy);
    m.c2 = vz;
    return m;
}
/* math.glsl:58-63 */
static inline vec3 perpendicularTo1(vec3 v) {
    float s = signBit(v.z);
    float a = -1.0f / (s + v.z);
    float b = v.x * v.y * a;
    return V3(b, s + v.y * v.y * a, -v.y);
}
/* math.glsl:66-77 */
static inline vec3 perpendicularTo2(vec3 a, vec3 b) {
    vec3 c = crosser(a, b);
    float len = v_length(c);
    if (len < 1e-5f) return perpendicularTo1(a);
    return V3(c.x / len, c.y / len, c.z / len);
}

/* ------------------------------------------------------------------------ */
/* random.philox.glsl / random.util.glsl                                    */
/* ------------------------------------------------------------------------ */

static inline uint32_t uaddCarry(uint32_t a, uint32_t b, uint32_t* carry) {
    uint32_t s = a + b;
    *carry = s < a ? 1u : 0u;
    return s;
}

/* random.philox.glsl:15-52 */
static void philox(const orc_philox_params* p, uint32_t stream, uint32_t i, uint32_t out[4]) {
    uint32_t carry = i << 2;
    uint32_t sx = p->offset[0], sy = p->offset[1], sz = p->offset[2], sw = p->offset[3];
    sx = uaddCarry(sx, carry, &carry);
    sy = uaddCarry(sy, carry, &carry);
    sz = uaddCarry(sz, carry, &carry);
    sw = uaddCarry(sw, carry, &carry);
    sx += carry; /* rollover */
    uint32_t kx = p->key[0], ky = p->key[1];
    kx = uaddCarry(kx, stream, &carry);
    ky = uaddCarry(ky, carry, &carry);
    kx += carry; /* rollover */

    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * sx;
        uint64_t p1 = (uint64_t)M1 * sz;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t nx = hi1 ^ sy ^ kx, ny = lo1, nz = hi0 ^ sw ^ ky, nw = lo0;
        sx = nx; sy = ny; sz = nz; sw = nw;
        kx += 0x9E3779B9u;
        ky += 0xBB67AE85u;
    }
    out[0] = sx; out[1] = sy; out[2] = sz; out[3] = sw;
}

/* random.util.glsl:4-9 */
static inline float normalizeUint(uint32_t i) {
    return fminf(u2f(0x3F7FFFFFu), (float)i * u2f(0x2F800000u));
}

/* random.philox.glsl:54-77: the buffer is recomputed on every draw because
 * philoxBufferCount is never updated */
static inline uint32_t random_u32(const orc_philox_params* p, uint32_t stream, uint32_t i) {
    uint32_t s[4];
    philox(p, stream, i, s);
    return s[i & 3u];
}

void orc_rng_fill(const orc_philox_params* p, uint32_t base_stream, uint32_t n_streams,
                  uint32_t first_dim, uint32_t n_samples, uint32_t* out_u32, float* out_f32) {
    for (uint32_t s = 0; s < n_streams; ++s)
        for (uint32_t k = 0; k < n_samples; ++k) {
            uint32_t v = random_u32(p, base_stream + s, first_dim + k);
            if (out_u32) out_u32[(size_t)s * n_samples + k] = v;
            if (out_f32) out_f32[(size_t)s * n_samples + k] = normalizeUint(v);
        }
}

/* ------------------------------------------------------------------------ */
/* lookup.glsl / material.glsl                                              */
/* ------------------------------------------------------------------------ */

/* material.py:122-140 (Medium.GLSL, 88 B), material.glsl:11-34 */
#pragma pack(push, 4)
typedef struct {
    float lambda_min, lambda_max;
    uint64_t n, vg, mu_a, mu_s, log_phase, phase_sampling;
    uint64_t phase_m12, phase_m22, phase_m33, phase_m34;
} MediumGLSL;
/* material.py:587-597 (Material.GLSL, 24 B), material.glsl:91-97 */
typedef struct {
    uint64_t inside, outside;
    uint32_t flagsInwards, flagsOutwards;
} MaterialGLSL;
#pragma pack(pop)

typedef struct { float n, vg, mu_s, mu_e; } MediumConstants;

typedef struct {
    uint64_t stats[ST_COUNT];
} Counters;

/* lookup.glsl:14-28 */
static inline float lookUp(uint64_t table, float u, float nullValue, Counters* cnt) {
    if (table == 0) return nullValue;
    const int32_t* hdr = (const int32_t*)(uintptr_t)table;
    const float* samples = (const float*)(hdr + 1);
    if (cnt) cnt->stats[ST_LUT]++;
    u = f_clamp(u, 0.0f, 1.0f);
    u *= (float)(hdr[0] - 1);
    float fl = floorf(u);
    int lo = (int)fl;
    int hi = (int)ceilf(u);
    float l = u - fl; /* fract */
    return samples[lo] * (1.0f - l) + samples[hi] * l;
}
float orc_lookup(uint64_t table, float u, float null_value) {
    return lookUp(table, u, null_value, NULL);
}

/* material.glsl:37-39 */
static inline float normalize_lambda(const MediumGLSL* m, float lambda) {
    return f_clamp((lambda - m->lambda_min) / (m->lambda_max - m->lambda_min), 0.0f, 1.0f);
}

/* material.glsl:50-74 */
static MediumConstants lookUpMedium(uint64_t medium, float lambda, Counters* cnt) {
    MediumConstants c;
    if (medium == 0) {
        c.n = 1.0f; c.vg = SPEED_OF_LIGHT_F; c.mu_s = 0.0f; c.mu_e = 0.0f;
        return c;
    }
    const MediumGLSL* m = (const MediumGLSL*)(uintptr_t)medium;
    float u = normalize_lambda(m, lambda);
    float mu_a = lookUp(m->mu_a, u, 0.0f, cnt);
    float mu_s = lookUp(m->mu_s, u, 0.0f, cnt);
    c.mu_e = mu_a + mu_s;
    c.n = lookUp(m->n, u, 1.0f, cnt);
    c.vg = lookUp(m->vg, u, SPEED_OF_LIGHT_F, cnt);
    c.mu_s = mu_s;
    return c;
}
void orc_lookup_medium(uint64_t medium, float lambda, float out[4]) {
    MediumConstants c = lookUpMedium(medium, lambda, NULL);
    out[0] = c.n; out[1] = c.vg; out[2] = c.mu_s; out[3] = c.mu_e;
}

/* ------------------------------------------------------------------------ */
/* ray state (ray.glsl:22-36, 163-170, 246-248)                             */
/* ------------------------------------------------------------------------ */

typedef struct {
    vec3 position, direction;
    uint64_t medium;
    float wavelength, time, lin_contrib, log_contrib;
    MediumConstants constants;
    /* polarized part (PolarizedForwardRay) */
    vec4 stokes;
    vec3 polRef;
} Ray;

typedef struct {
    const orc_common* c;
    int polarized;
    uint64_t global_medium; /* volume tracer: params.medium (USE_GLOBAL_MEDIUM) */
    int use_global_medium;
    orc_propagate_params prop;
    Counters cnt;
    float* hist;   /* partial histogram of the current block */
    float* track;  /* per-event dump (may be NULL) */
    uint32_t track_events;
    uint32_t cur_idx;   /* RNG stream / dimension of the photon being traced: responses that */
    uint32_t* cur_dim;  /* draw random numbers (response.time.store.glsl) use them */
} Ctx;

static inline uint64_t rayMedium(const Ctx* x, const Ray* r) {
    return x->use_global_medium ? x->global_medium : r->medium;
}

/* ---- random.sobol.glsl: shuffled, Owen-scrambled Sobol sampler (Burley 2020), padded to higher
 * dimensions: every draw re-scrambles the stream index with a seed derived from the dimension ---- */
static inline uint32_t bitrev32(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
/* :92-103 */
static inline uint32_t hp_mix32(uint32_t n) {
    n ^= n >> 16; n *= 0x21F0AAADu; n ^= n >> 15; n *= 0xD35A2D97u; n ^= n >> 15;
    return n ^ 0xE6FE3BEBu;
}
/* :111-119 */
static inline uint32_t lk_hash(uint32_t n, uint32_t seed) {
    n ^= n * 0x3D20ADEAu; n += seed; n *= (seed >> 16) | 1u; n ^= n * 0x05526C56u; n ^= n * 0x53A22864u;
    return n;
}
/* :122-124 */
static inline uint32_t combineHash(uint32_t seed, uint32_t v) { return seed ^ (v + 0x9E3779B9u + (seed << 6) + (seed >> 2)); }
/* :129-148; the direction numbers of dimension 1 (the table at :34-43) obey v[k+1] = v[k] ^ (v[k] >> 1) */
static inline uint32_t sobolBits(uint32_t index, uint32_t dim) {
    if (dim == 0) return bitrev32(index);
    uint32_t result = 0, v = 0x80000000u;
    for (; index != 0; index >>= 1, v ^= v >> 1)
        if (index & 1u) result ^= v;
    return result;
}
/* :150-155 */
static inline uint32_t nestedScramble(uint32_t v, uint32_t seed) { return bitrev32(lk_hash(bitrev32(v), seed)); }
/* :156-161 */
static inline float scrambledSobol(uint32_t scrambled_index, uint32_t dim, uint32_t seed) {
    return normalizeUint(nestedScramble(sobolBits(scrambled_index, dim), combineHash(seed, dim)));
}

/* random(idx, dim): random.philox.glsl:84-94, or with ORC_FLAG_RNG_SOBOL random.sobol.glsl:165-176 (the
 * seed travels in philox.key[0]) */
static inline float rnd(Ctx* x, uint32_t idx, uint32_t* dim) {
    float r;
    if (x->c->flags & ORC_FLAG_RNG_SOBOL) {
        uint32_t seed = combineHash(x->c->philox.key[0], hp_mix32(*dim));
        r = scrambledSobol(nestedScramble(idx, seed), 0, seed);
    } else {
        r = normalizeUint(random_u32(&x->c->philox, idx, *dim));
    }
    *dim += 1;
    return r;
}
/* random2D(idx, dim): two consecutive Philox draws (random.philox.glsl:79-82, 96-100), or the first two
 * Sobol dimensions at one scrambled index (random.sobol.glsl:178-193) */
static inline void rnd2(Ctx* x, uint32_t idx, uint32_t* dim, float* u0, float* u1) {
    if (x->c->flags & ORC_FLAG_RNG_SOBOL) {
        uint32_t seed = combineHash(x->c->philox.key[0], hp_mix32(*dim));
        uint32_t index = nestedScramble(idx, seed);
        *u0 = scrambledSobol(index, 0, seed);
        *u1 = scrambledSobol(index, 1, seed);
        *dim += 2;
        return;
    }
    *u0 = rnd(x, idx, dim);
    *u1 = rnd(x, idx, dim);
}

/* ray.glsl:136-143 */
static int isRayBad(const Ray* r) {
    vec3 p = r->position, d = r->direction;
    return isnan(p.x) || isnan(p.y) || isnan(p.z) || isinf(p.x) || isinf(p.y) || isinf(p.z) ||
           isnan(d.x) || isnan(d.y) || isnan(d.z) || isinf(d.x) || isinf(d.y) || isinf(d.z) ||
           v_length(d) <= 0.0f;
}

/* ------------------------------------------------------------------------ */
/* polarization.glsl                                                        */
/* ------------------------------------------------------------------------ */

/* reference-frame rotation matrix (column major, polarization.glsl:29-33)
 * applied to a Stokes vector: M*v = ((c0*x + c1*y) + c2*z) + c3*w */
static inline vec4 rotApply(float c, float s, vec4 v) {
    vec4 o;
    o.x = v.x;
    o.y = c * v.y + (-s) * v.z;
    o.z = s * v.y + c * v.z;
    o.w = v.w;
    return o;
}

/* polarization.glsl:38-68; returns 0 if the identity branch was taken */
static int rotatePolRef(vec3 dir, vec3 ref, vec3 nw, vec3* new_ref, float* c, float* s) {
    vec3 nr = crosser(dir, nw);
    float len = v_length(nr);
    if (len > 1.0e-7f) {
        nr = V3(nr.x / len, nr.y / len, nr.z / len);
    } else {
        *new_ref = ref;
        *c = 1.0f; *s = 0.0f;
        return 0;
    }
    float cos_phi = v_dot(ref, nr);
    float sin_phi = v_dot(crosser(ref, nr), dir);
    *c = 2.0f * cos_phi * cos_phi - 1.0f;
    *s = 2.0f * cos_phi * sin_phi;
    *new_ref = nr;
    return 1;
}
void orc_rotate_polref(const float dir[3], const float ref[3], const float nw[3],
                       float out_ref[3], float out_cs[2]) {
    vec3 r;
    rotatePolRef(V3(dir[0], dir[1], dir[2]), V3(ref[0], ref[1], ref[2]),
                 V3(nw[0], nw[1], nw[2]), &r, &out_cs[0], &out_cs[1]);
    out_ref[0] = r.x; out_ref[1] = r.y; out_ref[2] = r.z;
}

/* polarization.glsl:21-34 (alignPolRef) -> (c, s) */
static void alignPolRef(vec3 rayDir, vec3 oldRef, vec3 newRef, float* c, float* s) {
    float cos_phi = v_dot(oldRef, newRef);
    float sin_phi = v_dot(crosser(oldRef, newRef), rayDir);
    *c = 2.0f * cos_phi * cos_phi - 1.0f;
    *s = 2.0f * cos_phi * sin_phi;
}

void orc_align_polref(const float dir[3], const float old_ref[3], const float new_ref[3], float out_cs[2]) {
    alignPolRef(V3(dir[0], dir[1], dir[2]), V3(old_ref[0], old_ref[1], old_ref[2]), V3(new_ref[0], new_ref[1], new_ref[2]),
                &out_cs[0], &out_cs[1]);
}

/* polarization.glsl:109-120 applied to a Stokes vector */
static inline vec4 polarizerApply(float p, float s, vec4 v) {
    float att = p * p + s * s;
    float m12 = (p * p - s * s) / att;
    float m33 = (2.0f * p * s) / att;
    vec4 o;
    o.x = v.x + m12 * v.y;
    o.y = m12 * v.x + v.y;
    o.z = m33 * v.z;
    o.w = m33 * v.w;
    return o;
}
void orc_polarizer(float p, float s, const float in[4], float out[4]) {
    vec4 v = {in[0], in[1], in[2], in[3]};
    vec4 o = polarizerApply(p, s, v);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
}

/* ray.scatter.glsl:46-54: stokes = (phase * rotate) * stokes with the phase
 * matrix of polarization.glsl:88-107. The product phase*rotate is formed first
 * (GLSL is left-associative); structural zeros are dropped. */
static vec4 phaseRotateApply(uint64_t medium, float cos_theta, float c, float s, vec4 v,
                             Counters* cnt) {
    float m12 = 0.0f, m22 = 1.0f, m33 = 1.0f, m34 = 0.0f;
    if (medium != 0) {
        const MediumGLSL* m = (const MediumGLSL*)(uintptr_t)medium;
        float t = 0.5f * (cos_theta + 1.0f);
        m12 = lookUp(m->phase_m12, t, 0.0f, cnt);
        m22 = lookUp(m->phase_m22, t, 0.0f, cnt);
        m33 = lookUp(m->phase_m33, t, 0.0f, cnt);
        m34 = lookUp(m->phase_m34, t, 0.0f, cnt);
    }
    /* columns of P*R: col0 = (1,m12,0,0); col1 = c*P1 + s*P2; col2 = -s*P1 + c*P2;
     * col3 = (0,0,m34,m33), with P1=(m12,m22,0,0), P2=(0,0,m33,-m34) */
    vec4 o;
    o.x = (v.x + (m12 * c) * v.y) + (m12 * (-s)) * v.z;
    o.y = (m12 * v.x + (m22 * c) * v.y) + (m22 * (-s)) * v.z;
    o.z = ((m33 * s) * v.y + (m33 * c) * v.z) + m34 * v.w;
    o.w = (((-m34) * s) * v.y + ((-m34) * c) * v.z) + m33 * v.w;
    return o;
}
void orc_phase_matrix_apply(uint64_t medium, float cos_theta, float c, float s,
                            const float in[4], float out[4]) {
    vec4 v = {in[0], in[1], in[2], in[3]};
    vec4 o = phaseRotateApply(medium, cos_theta, c, s, v, NULL);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
}

/* ------------------------------------------------------------------------ */
/* scatter.volume.glsl                                                      */
/* ------------------------------------------------------------------------ */

/* scatter.volume.glsl:7-26 */
static vec3 scatterDir(vec3 prevDir, float cos_theta, float phi) {
    prevDir = v_normalize(prevDir);
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    vec3 local = V3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    local = v_normalize(local);
    mat3 trafo = createLocalCOSY(prevDir);
    return v_normalize(m3_mul(trafo, local));
}
void orc_scatter_dir(const float prev[3], float cos_theta, float phi, float out[3]) {
    vec3 r = scatterDir(V3(prev[0], prev[1], prev[2]), cos_theta, phi);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* scatter.volume.glsl:28-45 */
static float sampleScatterDir(uint64_t medium, float u0, float u1, float* cos_theta, float* phi,
                              Counters* cnt) {
    *phi = u0 * TWO_PI_F;
    const MediumGLSL* m = (const MediumGLSL*)(uintptr_t)medium;
    if (medium != 0 && m->phase_sampling != 0) {
        float ct = lookUp(m->phase_sampling, u1, 0.0f, cnt);
        ct = f_clamp(ct, -1.0f, 1.0f);
        *cos_theta = ct;
        return expf(lookUp(m->log_phase, 0.5f * (ct + 1.0f), 0.0f, cnt));
    }
    *cos_theta = 2.0f * u1 - 1.0f;
    return INV_4PI_F;
}

/* scatter.volume.glsl:47-52 */
static vec3 scatter(uint64_t medium, vec3 inDir, float u0, float u1, float* p, Counters* cnt) {
    float cos_theta, phi;
    *p = sampleScatterDir(medium, u0, u1, &cos_theta, &phi, cnt);
    return scatterDir(inDir, cos_theta, phi);
}
void orc_scatter_sample(uint64_t medium, const float in[3], float u0, float u1, float out[3],
                        float* p) {
    vec3 r = scatter(medium, V3(in[0], in[1], in[2]), u0, u1, p, NULL);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* scatter.volume.glsl:54-65 */
static float scatterProb(uint64_t medium, vec3 inDir, vec3 outDir, Counters* cnt) {
    const MediumGLSL* m = (const MediumGLSL*)(uintptr_t)medium;
    if (medium == 0 || m->log_phase == 0) return INV_4PI_F;
    float cos_theta = v_dot(inDir, outDir);
    float log_p = lookUp(m->log_phase, 0.5f * (cos_theta + 1.0f), 0.0f, cnt);
    return expf(log_p);
}
float orc_scatter_prob(uint64_t medium, const float in[3], const float out[3]) {
    return scatterProb(medium, V3(in[0], in[1], in[2]), V3(out[0], out[1], out[2]), NULL);
}

/* ------------------------------------------------------------------------ */
/* ray.scatter.glsl                                                         */
/* ------------------------------------------------------------------------ */

/* ray.scatter.glsl:46-54 */
static void scatterPolRay(Ctx* x, Ray* r, vec3 dir, float cos_theta) {
    vec3 polRef;
    float c, s;
    rotatePolRef(r->direction, r->polRef, dir, &polRef, &c, &s);
    r->stokes = phaseRotateApply(rayMedium(x, r), cos_theta, c, s, r->stokes, &x->cnt);
    r->polRef = polRef;
}

/* ray.scatter.glsl:12-15 (+ :70-74 polarized) */
static void scatterRayIS(Ctx* x, Ray* r, vec3 dir) {
    if (x->polarized) {
        float cos_theta = v_dot(r->direction, dir);
        scatterPolRay(x, r, dir, cos_theta);
    }
    r->direction = dir;
    r->lin_contrib *= r->constants.mu_s;
}

/* ray.scatter.glsl:31-38 (+ :92-100 polarized) */
static void scatterRay(Ctx* x, Ray* r, float u0, float u1) {
    float cos_theta, phi;
    sampleScatterDir(rayMedium(x, r), u0, u1, &cos_theta, &phi, &x->cnt);
    vec3 newDir = scatterDir(r->direction, cos_theta, phi);
    if (x->polarized) scatterPolRay(x, r, newDir, cos_theta);
    r->direction = newDir;
    r->lin_contrib *= r->constants.mu_s;
}

/* ------------------------------------------------------------------------ */
/* ray.propagate.glsl                                                       */
/* ------------------------------------------------------------------------ */

/* ray.propagate.glsl:32-49 */
static float sampleScatterLength(const Ray* r, const orc_propagate_params* p, float u) {
    float dist = p->maxDist;
    float sampleCoef = p->scatterCoefficient >= 0.0f ? p->scatterCoefficient : r->constants.mu_s;
    int canScatter = r->constants.mu_s > 0.0f;
    if (canScatter && sampleCoef != 0.0f) dist = -logf(1.0f - u) / sampleCoef;
    return dist;
}

/* ray.propagate.glsl:70-80 */
static int updateRay(Ray* r, float dist, const orc_propagate_params* p) {
    r->log_contrib -= r->constants.mu_e * dist;
    r->time += dist / r->constants.vg;
    return r->time <= p->maxTime ? RESULT_CODE_SUCCESS : RESULT_CODE_RAY_DECAYED;
}

/* ray.propagate.glsl:101-130 */
static void updateRayIS(Ray* r, float dist, const orc_propagate_params* p, int hit) {
    int canScatter = r->constants.mu_s > 0.0f;
    if (!canScatter) return;
    float sampleCoef = p->scatterCoefficient >= 0.0f ? p->scatterCoefficient : r->constants.mu_s;
    r->log_contrib += sampleCoef * dist;
    if (!hit) r->lin_contrib /= sampleCoef;
}

/* ray.propagate.glsl:153-166 */
static int propagateRay(Ray* r, float dist, const orc_propagate_params* p) {
    r->position = v_add(r->position, v_scale(r->direction, dist));
    vec3 q = r->position;
    int outside = q.x < p->lowerBBoxCorner[0] || q.y < p->lowerBBoxCorner[1] ||
                  q.z < p->lowerBBoxCorner[2] || q.x > p->upperBBoxCorner[0] ||
                  q.y > p->upperBBoxCorner[1] || q.z > p->upperBBoxCorner[2];
    int result = updateRay(r, dist, p);
    return outside ? RESULT_CODE_RAY_LOST : result;
}

/* ray.propagate.glsl:193-202 */
static void alignRayToHit(Ctx* x, Ray* r, vec3 hitNormal) {
    if (!x->polarized) return;
    vec3 polRef;
    float c, s;
    rotatePolRef(r->direction, r->polRef, hitNormal, &polRef, &c, &s);
    r->stokes = rotApply(c, s, r->stokes);
    r->polRef = polRef;
}

/* ray.propagate.glsl:237-265 */
static int propagateRayToHit(Ctx* x, Ray* r, vec3 hitPos, vec3 hitNormal,
                             const orc_propagate_params* p) {
    alignRayToHit(x, r, hitNormal);
    float dist = v_distance(r->position, hitPos);
    r->position = hitPos;
    return updateRay(r, dist, p);
}

/* ------------------------------------------------------------------------ */
/* response.histogram.glsl / ray.response.glsl / callbacks                  */
/* ------------------------------------------------------------------------ */

typedef struct {
    vec3 position, direction, normal;
    vec4 stokes;
    vec3 polRef;
    float wavelength, time, contrib;
    int objectId;
} HitItem;

/* response.record.glsl:23-26 + SAVE_HIT of response.queue.glsl: the queue the next traces
 * record into instead of binning (NULL: histogram response). Layout: uint32 count, then one
 * column of `capacity` floats per field. */
static uint32_t* g_hit_queue = NULL;
static uint32_t g_hit_capacity = 0;
static int g_hit_polarized = 0;
void orc_set_hit_queue(uint32_t* queue, uint32_t capacity, int polarized) {
    g_hit_queue = queue; g_hit_capacity = capacity; g_hit_polarized = polarized;
}

/* StoreTimeHitResponse(UniformValueResponse()): queue = uint32 count, float time[cap], int32 objectId[cap] */
static uint32_t* g_time_queue = NULL;
static uint32_t g_time_capacity = 0;
static int g_time_store_id = 0;
void orc_set_time_queue(uint32_t* queue, uint32_t capacity, int store_object_id) {
    g_time_queue = queue; g_time_capacity = capacity; g_time_store_id = store_object_id;
}
/* ScenePhotonTracer (tracer.scene.photon.{init,loop}.glsl): steps per run and RNG dimensions
 * per run (trace.py:2489-2491, 2655-2659); 0 steps = forward tracer */
static uint32_t g_photon_steps_per_run = 0, g_photon_rng_per_run = 0;
void orc_set_photon_mode(uint32_t steps_per_run, uint32_t rng_per_run) {
    g_photon_steps_per_run = steps_per_run; g_photon_rng_per_run = rng_per_run;
}


/* response.time.store.glsl:14-25 with response.uniform.glsl:4-6 */
static void storeTime(Ctx* x, const HitItem* it) {
    float value = it->contrib;
    if (rnd(x, x->cur_idx, x->cur_dim) < value) {
        uint32_t id;
        x->cnt.stats[ST_RESPONSES]++;
#pragma omp atomic capture
        id = g_time_queue[0]++;
        if (id >= g_time_capacity) return;
        float* t = (float*)(g_time_queue + 1) + id;
        *t = it->time;
        if (g_time_store_id) *(int32_t*)(t + g_time_capacity) = it->objectId;
    }
}

static void recordHit(Ctx* x, const HitItem* it) {
    uint32_t id;
    x->cnt.stats[ST_RESPONSES]++;
#pragma omp atomic capture
    id = g_hit_queue[0]++;
    if (id >= g_hit_capacity) return;
    float* col = (float*)(g_hit_queue + 1) + id;
    const uint32_t cap = g_hit_capacity;
#define PUT(v) do { *col = (v); col += cap; } while (0)
    PUT(it->position.x); PUT(it->position.y); PUT(it->position.z);
    PUT(it->direction.x); PUT(it->direction.y); PUT(it->direction.z);
    PUT(it->normal.x); PUT(it->normal.y); PUT(it->normal.z);
    if (g_hit_polarized) {
        if (x->polarized) {
            PUT(it->stokes.x); PUT(it->stokes.y); PUT(it->stokes.z); PUT(it->stokes.w);
            PUT(it->polRef.x); PUT(it->polRef.y); PUT(it->polRef.z);
        } else { /* response.queue.glsl:106-112 */
            PUT(1.0f); PUT(0.0f); PUT(0.0f); PUT(0.0f);
            PUT(0.0f); PUT(0.0f); PUT(0.0f);
        }
    }
    PUT(it->wavelength); PUT(it->time); PUT(it->contrib);
#undef PUT
    *(int32_t*)col = it->objectId;
}

/* response.histogram.glsl:59-68 with response.uniform.glsl:4-6 */
static void response(Ctx* x, const HitItem* item) {
    if (g_time_queue) { storeTime(x, item); return; }
    if (g_hit_queue) { recordHit(x, item); return; }
    if (x->c->flags & ORC_FLAG_KERNEL_HISTOGRAM) { /* response.histogram.kernel.glsl:62-99 */
        const orc_response_params* rp = &x->c->response;
        x->cnt.stats[ST_RESPONSES]++;
        float t = item->time - rp->t0;
        int firstBin = (int)floorf((t - rp->kernelSupport) / rp->binSize);
        int lastBin = (int)ceilf((t + rp->kernelSupport) / rp->binSize);
        if (firstBin < 0) firstBin = 0;
        if (lastBin > (int)x->c->n_bins) lastBin = (int)x->c->n_bins;
        lastBin -= 1;
        t = (float)firstBin * rp->binSize + rp->t0 - item->time;
        const float h = 1.0f / rp->kernelBandwidth;
#define KCDF(v) (0.5f + 0.5f * tanhf(7.779374467827938f * (v) - 55.5f * atanf(0.1257926109373887f * (v))))
        float prev_cdf = KCDF(t * h);
        for (int i = firstBin; i <= lastBin; ++i) {
            t += rp->binSize;
            float cdf = KCDF(t * h);
            float w = cdf - prev_cdf;
            prev_cdf = cdf;
            x->hist[i] += w * item->contrib;
        }
#undef KCDF
        return;
    }
    float value = item->contrib;
    float f = floorf((item->time - x->c->response.t0) / x->c->response.binSize);
    x->cnt.stats[ST_RESPONSES]++;
    /* uint bin = int(floor(..)): negative wraps to a huge uint and is rejected */
    if (f >= 0.0f && f < (float)x->c->n_bins) {
        uint32_t bin = (uint32_t)f;
        if (bin < x->c->n_bins) x->hist[bin] += value;
    }
}

/* ray.response.glsl:18-56 (polarized) / :63-84 (unpolarized) */
static HitItem createHit(Ctx* x, const Ray* r, vec3 objHitPos, vec3 objHitNormal, int objectId,
                         mat3 worldToObj) {
    HitItem it;
    memset(&it, 0, sizeof(it));
    float contrib = r->lin_contrib * expf(r->log_contrib);
    vec3 objHitDir = v_normalize(m3_mul(worldToObj, r->direction));
    if (x->polarized) {
        vec3 hitPolRef = perpendicularTo2(objHitDir, objHitNormal);
        vec3 objPolRef = v_normalize(m3_mul(worldToObj, r->polRef));
        float c, s;
        alignPolRef(objHitDir, objPolRef, hitPolRef, &c, &s);
        vec4 st = rotApply(c, s, r->stokes);
        contrib *= st.x;
        it.stokes.x = st.x / st.x; it.stokes.y = st.y / st.x;
        it.stokes.z = st.z / st.x; it.stokes.w = st.w / st.x;
        it.polRef = hitPolRef;
    }
    it.position = objHitPos;
    it.direction = objHitDir;
    it.normal = objHitNormal;
    it.wavelength = r->wavelength;
    it.time = r->time;
    it.contrib = contrib;
    it.objectId = objectId;
    return it;
}

/* callback.stat.glsl:29-70 + optional per-event dump (callback.track.glsl idea) */
static void onEvent(Ctx* x, const Ray* r, int code, uint32_t local_idx, uint32_t i, uint32_t dim) {
    uint64_t* s = x->cnt.stats;
    switch (code) {
    case RESULT_CODE_RAY_CREATED: s[ST_CREATED]++; break;
    case RESULT_CODE_RAY_SCATTERED: s[ST_SCATTERED]++; break;
    case RESULT_CODE_RAY_HIT: s[ST_HIT]++; break;
    case RESULT_CODE_RAY_DETECTED: s[ST_DETECTED]++; break;
    case RESULT_CODE_VOLUME_HIT: s[ST_VOLUME]++; break;
    case RESULT_CODE_RAY_LOST: s[ST_LOST]++; break;
    case RESULT_CODE_RAY_DECAYED: s[ST_DECAYED]++; break;
    case RESULT_CODE_RAY_ABSORBED: s[ST_ABSORBED]++; break;
    case RESULT_CODE_RAY_MISSED: s[ST_MISSED]++; break;
    case RESULT_CODE_MAX_ITER: s[ST_MAXITER]++; break;
    case ERROR_CODE_MEDIA_MISMATCH: s[ST_MISMATCH]++; break;
    default: break;
    }
    if (code <= ERROR_CODE_MAX_VALUE) s[ST_ERROR]++;
    if (x->track && i < x->track_events) {
        float* t = x->track + ((size_t)local_idx * x->track_events + i) * ORC_TRACK_WORDS;
        t[0] = (float)code;
        t[1] = r->position.x; t[2] = r->position.y; t[3] = r->position.z;
        t[4] = r->direction.x; t[5] = r->direction.y; t[6] = r->direction.z;
        t[7] = r->time; t[8] = r->lin_contrib; t[9] = r->log_contrib;
        t[10] = r->wavelength; t[11] = (float)dim;
        if (x->polarized) {
            t[12] = r->stokes.x; t[13] = r->stokes.y; t[14] = r->stokes.z; t[15] = r->stokes.w;
            t[16] = r->polRef.x; t[17] = r->polRef.y; t[18] = r->polRef.z;
        } else { /* callback.track.glsl:84-88 */
            t[12] = 1.0f; t[13] = 0.0f; t[14] = 0.0f; t[15] = 0.0f;
            t[16] = 0.0f; t[17] = 0.0f; t[18] = 0.0f;
        }
        t[19] = 0.0f;
    }
}

/* ------------------------------------------------------------------------ */
/* light + wavelength source                                                */
/* ------------------------------------------------------------------------ */

/* util.sample.glsl:10-17 */
static vec3 sphericalToCartessian(float phi, float cos_theta) {
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    return V3(sin_theta * sinf(phi), sin_theta * cosf(phi), cos_theta);
}

/* HostLightSource / HostWavelengthSource queues (NULL: not set) */
static const float* g_light_queue = NULL;
static uint32_t g_light_capacity = 0;
static const float* g_lam_queue = NULL;
static uint32_t g_lam_capacity = 0;
void orc_set_light_queue(const float* queue, uint32_t capacity) { g_light_queue = queue; g_light_capacity = capacity; }
void orc_set_wavelength_queue(const float* queue, uint32_t capacity) { g_lam_queue = queue; g_lam_capacity = capacity; }

typedef struct { float wavelength, contrib; } WavelengthSample;
/* lightsource.common.glsl:12-26 */
typedef struct {
    vec3 position, direction;
    vec4 stokes;
    vec3 polRef;
    float startTime, contrib;
} SourceRay;

/* wavelengthsource.const.glsl:10-12, wavelengthsource.uniform.glsl:12-16,
 * wavelengthsource.function.glsl:13-16, wavelengthsource.host.glsl:10-13.
 * `item`: index into a host queue (the reference's idx) */
static WavelengthSample sampleWavelength(Ctx* x, uint32_t idx, uint32_t item, uint32_t* dim) {
    const orc_common* c = x->c;
    WavelengthSample w;
    if (c->flags & ORC_FLAG_WAVELENGTH_HOST) {
        w.wavelength = g_lam_queue[item];
        w.contrib = g_lam_queue[(size_t)g_lam_capacity + item];
    } else if (c->flags & ORC_FLAG_WAVELENGTH_FUNCTION) {
        uint64_t table = (uint64_t)c->wavelength.table[0] | ((uint64_t)c->wavelength.table[1] << 32);
        w.wavelength = lookUp(table, rnd(x, idx, dim), 0.0f, &x->cnt);
        w.contrib = c->wavelength.p2;
    } else if (c->flags & ORC_FLAG_WAVELENGTH_UNIFORM) {
        float u = rnd(x, idx, dim);
        w.wavelength = f_mix(c->wavelength.p0, c->wavelength.p1, u);
        w.contrib = c->wavelength.p2;
    } else {
        w.wavelength = c->wavelength.p0;
        w.contrib = 1.0f;
    }
    return w;
}

/* lightsource.common.glsl:59-86: unpolarized source ray in a polarized environment */
static SourceRay unpolarizedSourceRay(Ctx* x, vec3 position, vec3 direction, float startTime, float contrib) {
    SourceRay r;
    memset(&r, 0, sizeof(r));
    r.position = position; r.direction = direction; r.startTime = startTime; r.contrib = contrib;
    if (x->polarized) {
        r.stokes.x = 1.0f;
        r.polRef = createLocalCOSY(direction).c0;
    }
    return r;
}
/* lightsource.common.glsl:88-112 */
static SourceRay polarizedSourceRay(Ctx* x, vec3 position, vec3 direction, vec4 stokes, vec3 polRef,
                                    float startTime, float contrib) {
    SourceRay r;
    memset(&r, 0, sizeof(r));
    r.position = position; r.direction = direction; r.startTime = startTime; r.contrib = contrib;
    if (x->polarized) { r.stokes = stokes; r.polRef = polRef; }
    return r;
}

/* ---- the Cherenkov family of light sources ---- */
static float f_sign(float v);
#define INV_SPEED_OF_LIGHT_F 3.335640951981521f /* material.glsl:7 */

/* lightsource.cherenkov.common.glsl:6-46: radiance per track length and wavelength, in eV or in photons (d/dphi form) */
static float frankTammCherenkov(float n, float lambda, int photons) {
    lambda *= 1e-3f; /* nm -> um */
    float result = photons ? 7.2973525693f / (lambda * lambda) * (1.0f - (1.0f / (n * n)))
                           : 9.04756408986352f / (lambda * lambda * lambda) * (1.0f - (1.0f / (n * n)));
    return fmaxf(result, 0.0f);
}
/* lightsource.particles.common.glsl:53-62: photons per track length and wavelength, all azimuths */
static float frankTammParticle(float lambda, float n) {
    lambda *= 1e-3f;
    float sin2_theta = 1.0f - 1.0f / (n * n);
    float result = TWO_PI_F * 7.2973525693f / (lambda * lambda) * sin2_theta;
    return fmaxf(result, 0.0f);
}
/* lightsource.particles.common.glsl:69-97 */
static float particleSampleEmissionAngle(float n, float a, float b, float u) {
    float cos_chev = 1.0f / n;
    float int_lower = 1.0f - expf(-b * powf(1.0f - cos_chev, a));
    float int_upper = 1.0f - expf(-b * powf(1.0f + cos_chev, a));
    u = u * (int_upper + int_lower) - int_lower;
    float xx = powf(-logf(1.0f - fabsf(u)) / b, 1.0f / a);
    return cos_chev - f_sign(u) * xx;
}
/* lightsource.particles.common.glsl:103-136: the same profile on [cos_min, cos_max], with the weight f / p */
static float particleSampleEmissionAngleRange(float n, float a, float b, float u, float cos_min, float cos_max, float* contrib) {
    float cos_chev = 1.0f / n;
    float int_upper = 1.0f - expf(-b * powf(fabsf(cos_max - cos_chev), a));
    float int_lower = 1.0f - expf(-b * powf(fabsf(cos_min - cos_chev), a));
    int_upper *= f_sign(cos_max - cos_chev);
    int_lower *= f_sign(cos_min - cos_chev);
    *contrib = fabsf(int_upper - int_lower);
    *contrib /= 2.0f - expf(-b * powf(1.0f - cos_chev, a)) - expf(-b * powf(1.0f + cos_chev, a));
    u = u * (int_upper - int_lower) + int_lower;
    float xx = -powf(-logf(1.0f - fabsf(u)) / b, 1.0f / a);
    return cos_chev - f_sign(u) * xx;
}
/* lightsource.particles.common.glsl:138-154 */
static float particleEvalEmissionAngle(float n, float a, float b, float cos_theta) {
    float cos_chev = 1.0f / n;
    float norm = a * b;
    norm /= 2.0f - expf(-b * powf(1.0f - cos_chev, a)) - expf(-b * powf(1.0f + cos_chev, a));
    float xx = fabsf(cos_theta - cos_chev);
    return expf(-b * powf(xx, a)) * powf(xx, a - 1.0f) * norm;
}
/* random.gamma.glsl:8-31: Cheng's GA rejection sampler; alpha < 1 through alpha + 1 */
static float sampleGamma(Ctx* x, float alpha, uint32_t idx, uint32_t* dim) {
    float scale = 1.0f;
    if (alpha < 1.0f) {
        scale = powf(rnd(x, idx, dim), 1.0f / alpha);
        alpha += 1.0f;
    }
    float a = sqrtf(2.0f * alpha - 1.0f);
    float b = alpha - logf(4.0f);
    float c = alpha + 1.0f / alpha;
    for (;;) {
        float u0, u1;
        rnd2(x, idx, dim, &u0, &u1);
        float V = a * logf(u0 / (1.0f - u0));
        float X = alpha * expf(V);
        if (b + c * V - X >= logf(u0 * u0 * u1)) return scale * X;
        if (!(u0 == u0)) return scale * X; /* (a NaN draw cannot happen; no endless loop if it ever did) */
    }
}
/* a Cherenkov photon: on the cone of half angle acos(1 / n) around the particle's direction, linearly polarized in the plane of the
 * two directions (lightsource.cherenkov.simple.glsl:37-57, lightsource.cherenkov.track.glsl:48-77) */
static SourceRay cherenkovRay(Ctx* x, vec3 pos, vec3 particleDir, float n, float phiDraw, float time, float contrib) {
    float cos_theta = 1.0f / n;
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    float phi = TWO_PI_F * phiDraw;
    vec3 localDir = V3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    vec3 rayDir = m3_mul(createLocalCOSY(particleDir), localDir);
    vec3 polRef = v_normalize(v_cross(rayDir, particleDir));
    vec4 stokes; stokes.x = 1.0f; stokes.y = 1.0f; stokes.z = 0.0f; stokes.w = 0.0f;
    return polarizedSourceRay(x, pos, rayDir, stokes, polRef, time, contrib);
}
static vec3 v_mix(vec3 a, vec3 b, float t) { return V3(f_mix(a.x, b.x, t), f_mix(a.y, b.y, t), f_mix(a.z, b.z, t)); }

/* forward sampleLight of lightsource.cherenkov.simple.glsl:20-58, lightsource.cherenkov.track.glsl:23-78,
 * lightsource.particles.muon.glsl:29-60, lightsource.particles.cascade.glsl:41-69 */
static SourceRay sampleCherenkovFamily(Ctx* x, float wavelength, const MediumConstants* medium, uint32_t idx, uint32_t* dim) {
    const orc_common* c = x->c;
    const int photons = (c->flags & ORC_FLAG_FRANK_TAMM_PHOTONS) != 0;
    const int applyFrankTamm = (c->flags & ORC_FLAG_FRANK_TAMM_IS) == 0;
    switch (c->flags & ORC_FLAG_LIGHT_KIND_MASK) {
    case ORC_FLAG_LIGHT_CHERENKOV: {
        const orc_cherenkov_params* p = &c->light.cherenkov;
        float u = rnd(x, idx, dim);
        vec3 position = v_mix(V3(p->trackStart[0], p->trackStart[1], p->trackStart[2]), V3(p->trackEnd[0], p->trackEnd[1], p->trackEnd[2]), u);
        float startTime = f_mix(p->startTime, p->endTime, u);
        float contrib = TWO_PI_F * frankTammCherenkov(medium->n, wavelength, photons) * p->trackDist;
        float v = rnd(x, idx, dim);
        return cherenkovRay(x, position, V3(p->trackDir[0], p->trackDir[1], p->trackDir[2]), medium->n, v, startTime, contrib);
    }
    case ORC_FLAG_LIGHT_CHERENKOV_TRACK: {
        const uint32_t* track = (const uint32_t*)(uintptr_t)((uint64_t)c->light.cherenkovTrack.track[0] | ((uint64_t)c->light.cherenkovTrack.track[1] << 32));
        const uint32_t trackLength = track[0];
        const float* vertices = (const float*)(track + 1);
        float u = rnd(x, idx, dim);
        u *= (float)trackLength;
        uint32_t vertexIdx = (uint32_t)floorf(u);
        if (vertexIdx > trackLength - 1u) vertexIdx = trackLength - 1u;
        u = u - floorf(u); /* fract */
        const float* s = vertices + 4 * vertexIdx;
        const float* e = s + 4;
        vec3 sp = V3(s[0], s[1], s[2]), ep = V3(e[0], e[1], e[2]);
        vec3 pos = v_mix(sp, ep, u);
        float time = f_mix(s[3], e[3], u);
        float v = rnd(x, idx, dim);
        vec3 particleDir = v_normalize(v_sub(ep, sp));
        float segmentLength = v_distance(sp, ep);
        float contrib = TWO_PI_F * (float)trackLength * segmentLength;
        contrib *= frankTammCherenkov(medium->n, wavelength, photons);
        return cherenkovRay(x, pos, particleDir, medium->n, v, time, contrib);
    }
    case ORC_FLAG_LIGHT_MUON_TRACK: {
        const orc_muon_track_params* p = &c->light.muonTrack;
        vec3 sp = V3(p->startPosition[0], p->startPosition[1], p->startPosition[2]);
        vec3 ep = V3(p->endPosition[0], p->endPosition[1], p->endPosition[2]);
        float u = rnd(x, idx, dim);
        vec3 position = v_mix(sp, ep, u);
        float startTime = f_mix(p->startTime, p->endTime, u);
        float contrib = v_distance(sp, ep);
        contrib *= p->energyScale;
        float v0, v1;
        rnd2(x, idx, dim, &v0, &v1);
        float phi = TWO_PI_F * v0;
        float cos_theta = particleSampleEmissionAngle(medium->n, p->a_angular, p->b_angular, v1);
        vec3 trackDir = v_normalize(v_sub(ep, sp));
        vec3 rayDir = m3_mul(createLocalCOSY(trackDir), sphericalToCartessian(phi, cos_theta));
        if (applyFrankTamm) contrib *= frankTammParticle(wavelength, medium->n);
        return unpolarizedSourceRay(x, position, rayDir, startTime, contrib);
    }
    default: { /* ORC_FLAG_LIGHT_CASCADE */
        const orc_cascade_params* p = &c->light.cascade;
        vec3 dir = V3(p->direction[0], p->direction[1], p->direction[2]);
        float z = sampleGamma(x, p->a_long, idx, dim) * p->b_long;
        vec3 rayPos = v_add(V3(p->startPosition[0], p->startPosition[1], p->startPosition[2]), v_scale(dir, z));
        float time = p->startTime + z * INV_SPEED_OF_LIGHT_F;
        float u0, u1;
        rnd2(x, idx, dim, &u0, &u1);
        float phi = TWO_PI_F * u0;
        float cos_theta = particleSampleEmissionAngle(medium->n, p->a_angular, p->b_angular, u1);
        vec3 rayDir = m3_mul(createLocalCOSY(dir), sphericalToCartessian(phi, cos_theta));
        float contrib = p->effectiveLength;
        if (applyFrankTamm) contrib *= frankTammParticle(wavelength, medium->n);
        return unpolarizedSourceRay(x, rayPos, rayDir, time, contrib);
    }
    }
}

void orc_cherenkov_units(uint32_t which, const float* rows, uint32_t count, float* out) {
    for (uint32_t i = 0; i < count; ++i) {
        if (which == 0) {
            const float* r = rows + 4 * (size_t)i;
            out[i] = particleSampleEmissionAngle(r[0], r[1], r[2], r[3]);
        } else if (which == 1) {
            const float* r = rows + 6 * (size_t)i;
            out[2 * (size_t)i] = particleSampleEmissionAngleRange(r[0], r[1], r[2], r[3], r[4], r[5], &out[2 * (size_t)i + 1]);
        } else if (which == 2) {
            const float* r = rows + 4 * (size_t)i;
            out[i] = particleEvalEmissionAngle(r[0], r[1], r[2], r[3]);
        } else {
            const float* r = rows + 2 * (size_t)i;
            out[3 * (size_t)i] = frankTammCherenkov(r[0], r[1], 0);
            out[3 * (size_t)i + 1] = frankTammCherenkov(r[0], r[1], 1);
            out[3 * (size_t)i + 2] = frankTammParticle(r[1], r[0]);
        }
    }
}

/* lightsource.host.glsl:12-19, lightsource.cone.glsl:22-72, lightsource.pencil.glsl:18-35,
 * lightsource.spherical.glsl:16-34 */
static SourceRay sampleLight(Ctx* x, float wavelength, const MediumConstants* medium, uint32_t idx, uint32_t item, uint32_t* dim) {
    const orc_common* c = x->c;
    if (c->flags & ORC_FLAG_LIGHT_KIND_MASK) return sampleCherenkovFamily(x, wavelength, medium, idx, dim);
    vec3 lightPos = V3(c->light.position[0], c->light.position[1], c->light.position[2]);
    if (c->flags & ORC_FLAG_LIGHT_HOST) { /* LOAD_SAMPLE of lightsource.queue.glsl */
        const float* q = g_light_queue + item;
        const size_t cap = g_light_capacity;
        vec3 pos = V3(q[0], q[cap], q[2 * cap]);
        vec3 dir = V3(q[3 * cap], q[4 * cap], q[5 * cap]);
        if (c->flags & ORC_FLAG_LIGHT_QUEUE_POLARIZED) {
            vec4 stokes; stokes.x = q[6 * cap]; stokes.y = q[7 * cap]; stokes.z = q[8 * cap]; stokes.w = q[9 * cap];
            vec3 polRef = V3(q[10 * cap], q[11 * cap], q[12 * cap]);
            return polarizedSourceRay(x, pos, dir, stokes, polRef, q[13 * cap], q[14 * cap]);
        }
        return unpolarizedSourceRay(x, pos, dir, q[6 * cap], q[7 * cap]);
    }
    if (c->flags & ORC_FLAG_LIGHT_CONE) {
        float u0, u1;
        rnd2(x, idx, dim, &u0, &u1); /* random2D */
        float phi = TWO_PI_F * u0;
        float cos_theta = (1.0f - u1) + c->light.cosOpeningAngle * u1;
        float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
        vec3 localDir = V3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
        mat3 trafo = createLocalCOSY(v_normalize(V3(c->light.direction[0], c->light.direction[1], c->light.direction[2])));
        vec3 rayDir = m3_mul(trafo, localDir);
        float v = rnd(x, idx, dim);
        float startTime = f_mix(c->light.t_min, c->light.t_max, v);
        if (c->flags & ORC_FLAG_LIGHT_POLARIZED) {
            vec3 polRef = V3(c->light.polRef[0], c->light.polRef[1], c->light.polRef[2]);
            polRef = v_sub(polRef, v_scale(rayDir, v_dot(polRef, rayDir)));
            polRef = v_normalize(polRef);
            vec4 stokes; stokes.x = c->light.stokes[0]; stokes.y = c->light.stokes[1];
            stokes.z = c->light.stokes[2]; stokes.w = c->light.stokes[3];
            return polarizedSourceRay(x, lightPos, rayDir, stokes, polRef, startTime, c->light.contribFwd);
        }
        return unpolarizedSourceRay(x, lightPos, rayDir, startTime, c->light.contribFwd);
    }
    if (c->flags & ORC_FLAG_LIGHT_PENCIL) {
        float v = rnd(x, idx, dim);
        vec4 stokes; stokes.x = c->light.stokes[0]; stokes.y = c->light.stokes[1];
        stokes.z = c->light.stokes[2]; stokes.w = c->light.stokes[3];
        return polarizedSourceRay(x, lightPos, V3(c->light.direction[0], c->light.direction[1], c->light.direction[2]),
                                  stokes, V3(c->light.polRef[0], c->light.polRef[1], c->light.polRef[2]),
                                  f_mix(c->light.t_min, c->light.t_max, v), c->light.contribFwd);
    }
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    vec3 rayDir = sphericalToCartessian(TWO_PI_F * u0, 2.0f * u1 - 1.0f);
    float v = rnd(x, idx, dim);
    return unpolarizedSourceRay(x, lightPos, rayDir, f_mix(c->light.t_min, c->light.t_max, v), c->light.contribFwd);
}

/* tracer.*.forward.glsl sampleRay; ray.glsl:38-79, 188-211 */
static Ray sampleRay(Ctx* x, uint64_t medium, uint32_t idx, uint32_t* dim) {
    const uint32_t item = idx - x->c->photon_offset; /* position in the batch */
    WavelengthSample photon = sampleWavelength(x, idx, item, dim);
    MediumConstants constants = lookUpMedium(medium, photon.wavelength, &x->cnt);
    SourceRay s = sampleLight(x, photon.wavelength, &constants, idx, item, dim);
    Ray r;
    memset(&r, 0, sizeof(r));
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
    if (g_photon_steps_per_run) r.lin_contrib = 1.0f; /* tracer.scene.photon.init.glsl:49-53 */
    return r;
}

/* lightsource.sample.glsl:52-75 */
int orc_sample_light(const orc_common* c, uint64_t medium, uint32_t count, uint32_t base, int polarized,
                     uint32_t capacity, uint32_t* light_out, uint32_t* wavelength_out) {
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = c;
    x.polarized = polarized != 0;
    if (count > capacity) return -1;
    for (uint32_t idx = 0; idx < count; ++idx) {
        uint32_t dim = 0;
        WavelengthSample photon = sampleWavelength(&x, idx + base, idx + base, &dim);
        MediumConstants constants = lookUpMedium(medium, photon.wavelength, &x.cnt);
        SourceRay ray = sampleLight(&x, photon.wavelength, &constants, idx + base, idx + base, &dim);
        float* col = (float*)(light_out + 1) + idx;
        const size_t cap = capacity;
#define PUT(v) do { *col = (v); col += cap; } while (0)
        PUT(ray.position.x); PUT(ray.position.y); PUT(ray.position.z);
        PUT(ray.direction.x); PUT(ray.direction.y); PUT(ray.direction.z);
        if (polarized) {
            PUT(ray.stokes.x); PUT(ray.stokes.y); PUT(ray.stokes.z); PUT(ray.stokes.w);
            PUT(ray.polRef.x); PUT(ray.polRef.y); PUT(ray.polRef.z);
        }
        PUT(ray.startTime); PUT(ray.contrib);
#undef PUT
        float* w = (float*)(wavelength_out + 1) + idx;
        w[0] = photon.wavelength;
        w[cap] = photon.contrib;
    }
    light_out[0] = count;
    wavelength_out[0] = count;
    return 0;
}

/* ------------------------------------------------------------------------ */
/* VOLUME TRACER: target.sphere.glsl, math.sphere.glsl, tracer.volume.forward */
/* ------------------------------------------------------------------------ */

typedef struct {
    vec3 position, normal;
    float dist;
    vec3 objPosition, objNormal;
    float prob;
    int valid;
    vec3 offset;
    mat3 worldToObj; /* spheres: mat3(invRad) */
} TargetSample;

/* math.sphere.glsl:14-39; returns nearest t (x component) */
static float intersectSphereNear(vec3 center, float radius, vec3 origin, vec3 direction) {
    vec3 f = v_sub(origin, center);
    float b2 = v_dot(f, direction);
    float r2 = radius * radius;
    vec3 fd = v_sub(f, v_scale(direction, b2));
    float discr = r2 - v_dot(fd, fd);
    float rx = INFINITY, ry = INFINITY;
    if (discr >= 0.0f) {
        float cc = v_dot(f, f) - r2;
        float q = -b2 - signBit(b2) * sqrtf(discr);
        rx = cc / q;
        ry = q;
        if (rx > ry) { float t = rx; rx = ry; ry = t; }
    }
    (void)ry;
    return rx;
}

static mat3 diag3(float d) {
    mat3 m;
    m.c0 = V3(d, 0, 0); m.c1 = V3(0, d, 0); m.c2 = V3(0, 0, d);
    return m;
}

static TargetSample targetMiss(void) { /* target.common.glsl:79-91 */
    TargetSample s;
    memset(&s, 0, sizeof(s));
    s.dist = INFINITY;
    s.worldToObj = diag3(1.0f);
    return s;
}

/* math.sphere.glsl:14-39, farthest root (.y) */
static float intersectSphereFar(vec3 center, float radius, vec3 origin, vec3 direction) {
    vec3 f = v_sub(origin, center);
    float b2 = v_dot(f, direction);
    float r2 = radius * radius;
    vec3 fd = v_sub(f, v_scale(direction, b2));
    float discr = r2 - v_dot(fd, fd);
    float ry = INFINITY;
    if (discr >= 0.0f) {
        float cc = v_dot(f, f) - r2;
        float q = -b2 - signBit(b2) * sqrtf(discr);
        float rx = cc / q;
        ry = q;
        if (rx > ry) ry = rx;
    }
    return ry;
}

static int g_target_inner = 0; /* set per trace from ORC_FLAG_TARGET_INNER */
static int g_target_flat = 0;  /* 1: DiskTarget, 2: FlatTarget (ORC_FLAG_TARGET_DISK / _FLAT) */
static int g_target_miss_inf = 0; /* ORC_FLAG_TARGET_MISS_INFINITE: plane targets report a miss as infinitely far */

/* util.sample.glsl:31-56: concentric mapping of the unit square onto the unit disk */
static vec3 sampleUnitDisk(float u0, float u1) {
    float rx = 2.0f * u0 - 1.0f, ry = 2.0f * u1 - 1.0f;
    if (rx == 0.0f && ry == 0.0f) return V3(0, 0, 0);
    float phi, r;
    if (fabsf(rx) > fabsf(ry)) {
        r = rx;
        phi = 0.78539816339744830962f * (ry / rx);
    } else {
        r = ry;
        phi = 1.57079632679489661923f - 0.78539816339744830962f * (rx / ry);
    }
    return V3(r * cosf(phi), r * sinf(phi), 0.0f);
}

static mat3 targetObjToWorld(const float m[9]) { /* column major */
    mat3 o;
    o.c0 = V3(m[0], m[1], m[2]); o.c1 = V3(m[3], m[4], m[5]); o.c2 = V3(m[6], m[7], m[8]);
    return o;
}
static float f_sign(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

/* target.disk.glsl:15-41 / target.flat.glsl:17-45 */
static TargetSample sampleTargetPlane(Ctx* x, const orc_target_params* tp, vec3 observer, uint32_t idx, uint32_t* dim) {
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    vec3 localPos = g_target_flat == 1 ? v_scale(sampleUnitDisk(u0, u1), tp->radius)
                                       : V3(tp->width * (u0 - 0.5f), tp->height * (u1 - 0.5f), 0.0f);
    mat3 o2w = targetObjToWorld(tp->objToWorld);
    vec3 center = V3(tp->position[0], tp->position[1], tp->position[2]);
    vec3 pos = v_add(m3_mul(o2w, localPos), center);
    vec3 normal = V3(tp->normal[0], tp->normal[1], tp->normal[2]);
    float side = f_sign(v_dot(normal, v_sub(observer, pos)));
    normal = v_scale(normal, side);
    TargetSample s;
    memset(&s, 0, sizeof(s));
    s.position = pos; s.normal = normal; s.dist = v_distance(observer, pos);
    s.objPosition = localPos; s.objNormal = V3(0.0f, 0.0f, side);
    s.prob = tp->hemisphereProb; s.valid = v_dot(normal, normal) != 0.0f;
    s.worldToObj = m3_transpose(o2w);
    s.offset = v_scale(m3_mul(s.worldToObj, center), -1.0f);
    return s;
}
/* target.disk.glsl:43-66 / target.flat.glsl:47-72 */
static TargetSample intersectTargetPlane(const orc_target_params* tp, vec3 observer, vec3 direction) {
    mat3 o2w = targetObjToWorld(tp->objToWorld);
    mat3 w2o = m3_transpose(o2w);
    vec3 center = V3(tp->position[0], tp->position[1], tp->position[2]);
    vec3 localObs = m3_mul(w2o, v_sub(observer, center));
    vec3 localDir = m3_mul(w2o, direction);
    float t = -localObs.z / localDir.z;
    vec3 localPos = v_add(localObs, v_scale(localDir, t));
    int valid;
    if (g_target_flat == 1) valid = t > 0.0f && sqrtf(localPos.x * localPos.x + localPos.y * localPos.y) <= tp->radius;
    else valid = t > 0.0f && 2.0f * fabsf(localPos.x) <= tp->width && 2.0f * fabsf(localPos.y) <= tp->height;
    vec3 pos = v_add(m3_mul(o2w, localPos), center);
    float side = signBit(localObs.z);
    TargetSample s;
    memset(&s, 0, sizeof(s));
    s.position = pos; s.normal = v_scale(V3(tp->normal[0], tp->normal[1], tp->normal[2]), side);
    /* target.disk.glsl:60 / target.flat.glsl:66 report the distance to the PLANE also when the outline is
     * missed or the plane lies behind the ray, and the tracers take min(hit.dist, free path)
     * (tracer.volume.forward.glsl:167): every ray stops at the infinite plane, rays pointing away
     * from it walk backwards onto it (negative distance). That is the default here as well (results
     * identical to the reference's). ORC_FLAG_TARGET_MISS_INFINITE opts into the repaired behaviour:
     * a miss is infinitely far, like createTargetSampleMiss(). */
    s.dist = (valid || !g_target_miss_inf) ? t : INFINITY;
    s.objPosition = localPos; s.objNormal = V3(0.0f, 0.0f, side);
    s.prob = tp->hemisphereProb * (valid ? 1.0f : 0.0f); s.valid = valid;
    s.worldToObj = w2o;
    s.offset = v_scale(m3_mul(w2o, center), -1.0f);
    return s;
}
/* isOccludedByTarget: target.sphere.glsl:80-83 (inside), target.sphere.inner.glsl:59-61 (outside),
 * target.disk.glsl:68-71 / target.flat.glsl:74-77 (never) */
static int isOccludedByTarget(const orc_target_params* tg, vec3 observer) {
    if (g_target_flat) return 0;
    float d = v_distance(observer, V3(tg->position[0], tg->position[1], tg->position[2]));
    return g_target_inner ? d >= tg->radius : d <= tg->radius;
}

/* target.sphere.inner.glsl:36-57 */
static TargetSample intersectTargetInner(const orc_target_params* tp, vec3 observer, vec3 direction) {
    vec3 center = V3(tp->position[0], tp->position[1], tp->position[2]);
    float t = intersectSphereFar(center, tp->radius, observer, direction);
    int hit = (t > 0.0f) && !isinf(t);
    TargetSample s;
    memset(&s, 0, sizeof(s));
    vec3 pos = v_add(observer, v_scale(direction, t));
    vec3 nrm = v_normalize(v_sub(center, pos));
    s.position = pos; s.normal = nrm; s.dist = t;
    s.objPosition = v_scale(nrm, -1.0f); s.objNormal = nrm;
    s.prob = tp->hemisphereProb; s.valid = hit;
    s.offset = V3(tp->invPos[0], tp->invPos[1], tp->invPos[2]);
    s.worldToObj = diag3(tp->invRad);
    return s;
}

/* target.sphere.glsl:45-78 */
static TargetSample intersectTarget(const orc_target_params* tp, vec3 observer, vec3 direction) {
    if (g_target_flat) return intersectTargetPlane(tp, observer, direction);
    if (g_target_inner) return intersectTargetInner(tp, observer, direction);
    vec3 center = V3(tp->position[0], tp->position[1], tp->position[2]);
    float t = intersectSphereNear(center, tp->radius, observer, direction);
    int hit = (t > 0.0f) && !isinf(t);
    if (!hit) return targetMiss();
    TargetSample s;
    vec3 pos = v_add(observer, v_scale(direction, t));
    vec3 nrm = v_normalize(v_sub(pos, center));
    float d = v_distance(observer, center);
    float cosOpening = tp->radius / d;
    float prob = tp->hemisphereProb / (1.0f - cosOpening);
    int valid = !isinf(prob);
    if (!valid) prob = 0.0f;
    s.position = pos; s.normal = nrm; s.dist = t;
    s.objPosition = nrm; s.objNormal = nrm;
    s.prob = prob; s.valid = valid;
    s.offset = V3(tp->invPos[0], tp->invPos[1], tp->invPos[2]);
    s.worldToObj = diag3(tp->invRad);
    return s;
}

/* target.sphere.glsl:18-43 */
static TargetSample sampleTarget(Ctx* x, const orc_target_params* tp, vec3 observer, uint32_t idx,
                                 uint32_t* dim) {
    if (g_target_flat) return sampleTargetPlane(x, tp, observer, idx, dim);
    vec3 center = V3(tp->position[0], tp->position[1], tp->position[2]);
    if (g_target_inner) { /* target.sphere.inner.glsl:18-34: uniform over the sphere */
        float v0, v1;
        rnd2(x, idx, dim, &v0, &v1); /* random2D */
        vec3 normal = sphericalToCartessian(TWO_PI_F * v0, 2.0f * v1 - 1.0f); /* util.sample.glsl:84-88 */
        vec3 pos = v_add(v_scale(normal, tp->radius), center);
        TargetSample s;
        memset(&s, 0, sizeof(s));
        s.position = pos; s.normal = v_scale(normal, -1.0f); s.dist = v_distance(observer, pos);
        s.objPosition = normal; s.objNormal = v_scale(normal, -1.0f);
        s.prob = tp->hemisphereProb; s.valid = 1;
        s.offset = V3(tp->invPos[0], tp->invPos[1], tp->invPos[2]);
        s.worldToObj = diag3(tp->invRad);
        return s;
    }
    vec3 dir = v_normalize(v_sub(observer, center));
    float d = v_distance(observer, center);
    float cosOpening = tp->radius / d;
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    /* util.sample.glsl:23-27 sampleDirectionCone */
    float phi = TWO_PI_F * u0;
    float cos_theta = (1.0f - u1) + cosOpening * u1;
    vec3 normal = m3_mul(createLocalCOSY(dir), sphericalToCartessian(phi, cos_theta));
    vec3 pos = v_add(v_scale(normal, tp->radius), center);
    float prob = tp->hemisphereProb / (1.0f - cosOpening);
    int valid = !isinf(prob);
    if (!valid) prob = 0.0f;
    TargetSample s;
    s.position = pos; s.normal = normal; s.dist = v_distance(pos, observer);
    s.objPosition = normal; s.objNormal = normal;
    s.prob = prob; s.valid = valid;
    s.offset = V3(tp->invPos[0], tp->invPos[1], tp->invPos[2]);
    s.worldToObj = diag3(tp->invRad);
    return s;
}

/* tracer.volume.forward.glsl:51-82 */
static void vol_createResponse(Ctx* x, const orc_volume_trace_params* tp, Ray ray,
                               const TargetSample* hit, vec3 dir, float weight, int scattered) {
    if (!hit->valid) return;
    if (scattered) scatterRayIS(x, &ray, dir);
    int code = propagateRayToHit(x, &ray, hit->position, hit->normal, &x->prop);
    if (code < 0) return;
    ray.lin_contrib *= weight;
    HitItem item = createHit(x, &ray, hit->objPosition, hit->objNormal, tp->objectId,
                             hit->worldToObj);
    if (item.contrib > 0.0f) response(x, &item);
}

/* tracer.volume.forward.glsl:103-114 */
static float jacobian_dAdW(vec3 obs, vec3 pos, vec3 nrm) {
    vec3 dir = v_sub(pos, obs);
    float d2 = v_dot(dir, dir);
    dir = v_normalize(dir);
    float factor = d2 / fabsf(v_dot(dir, nrm));
    if (isinf(factor)) factor = 0.0f;
    return factor;
}

/* tracer.volume.forward.glsl:116-149 */
static void vol_sampleTargetMIS(Ctx* x, const orc_volume_trace_params* tp,
                                const orc_target_params* tg, const Ray* ray, uint32_t idx,
                                uint32_t* dim) {
    uint64_t med = x->global_medium;
    vec3 obs = ray->position, dir = ray->direction;
    float pPP;
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    vec3 dirPhase = scatter(med, dir, u0, u1, &pPP, &x->cnt);
    TargetSample phaseHit = intersectTarget(tg, obs, dirPhase);
    TargetSample targetHit = sampleTarget(x, tg, obs, idx, dim);
    vec3 dirTarget = v_normalize(v_sub(targetHit.position, obs));
    float pTT = targetHit.prob * jacobian_dAdW(obs, targetHit.position, targetHit.normal);
    float pPT = scatterProb(med, dir, dirTarget, &x->cnt);
    float pTP = phaseHit.prob * jacobian_dAdW(obs, phaseHit.position, phaseHit.normal);
    float wTarget = pTT * pPT / (pTT * pTT + pPT * pPT);
    float wPhase = pPP * pPP / (pPP * pPP + pTP * pTP);
    vol_createResponse(x, tp, *ray, &phaseHit, dirPhase, wPhase, 1);
    vol_createResponse(x, tp, *ray, &targetHit, dirTarget, wTarget, 1);
}

/* tracer.volume.forward.glsl:153-208 */
static int vol_trace(Ctx* x, const orc_volume_trace_params* tp, const orc_target_params* tg,
                     Ray* ray, uint32_t idx, uint32_t* dim, int allowResponse) {
    x->cnt.stats[ST_STEPS]++;
    float u = rnd(x, idx, dim);
    float dist = sampleScatterLength(ray, &x->prop, u);
    TargetSample hit = intersectTarget(tg, ray->position, ray->direction);
    int hitValid = hit.valid && (hit.dist <= dist);
    dist = fminf(hit.dist, dist);
    int result = propagateRay(ray, dist, &x->prop);
    updateRayIS(ray, dist, &x->prop, hitValid);
    if (result < 0) return result;
    if (hitValid && allowResponse) {
        alignRayToHit(x, ray, hit.normal);
        HitItem item = createHit(x, ray, hit.objPosition, hit.objNormal, tp->objectId,
                                 hit.worldToObj);
        if (item.contrib > 0.0f) response(x, &item);
        return RESULT_CODE_RAY_DETECTED;
    } else if (hitValid) {
        return RESULT_CODE_RAY_ABSORBED;
    }
    if (!(x->c->flags & ORC_FLAG_DISABLE_MIS)) vol_sampleTargetMIS(x, tp, tg, ray, idx, dim);
    return RESULT_CODE_RAY_SCATTERED;
}

/* VolumePhotonTracer: tracer.volume.photon.common.glsl:37-121, tracer.volume.photon.init.glsl:46-62
 * (all runs in one loop; run r starts at RNG dimension r * rng_per_run; events carry the running
 * step index where the reference restarts it per run) */
static void vol_photonMain(Ctx* x, const orc_volume_trace_params* tp, const orc_target_params* tg,
                           uint32_t local_idx) {
    const orc_common* c = x->c;
    uint32_t dim = 0;
    uint32_t idx = c->photon_offset + local_idx;
    x->cur_idx = idx; x->cur_dim = &dim;
    Ray ray = sampleRay(x, x->global_medium, idx, &dim);
    onEvent(x, &ray, RESULT_CODE_RAY_CREATED, local_idx, 0, dim);
    if (isOccludedByTarget(tg, ray.position)) {
        onEvent(x, &ray, ERROR_CODE_TRACE_ABORT, local_idx, 0, dim);
        return;
    }
    const uint32_t perRun = g_photon_steps_per_run;
    for (uint32_t i = 1; i <= c->path_length; ++i) {
        if (i > 1 && (i - 1) % perRun == 0) dim = ((i - 1) / perRun) * g_photon_rng_per_run;
        x->cnt.stats[ST_STEPS]++;
        float u = rnd(x, idx, &dim);
        float dist = sampleScatterLength(&ray, &x->prop, u);
        TargetSample hit = intersectTarget(tg, ray.position, ray.direction);
        int hitValid = hit.valid && (hit.dist <= dist);
        dist = fminf(hit.dist, dist);
        int result = propagateRay(&ray, dist, &x->prop);
        updateRayIS(&ray, dist, &x->prop, hitValid);
        if (result >= 0) {
            if (hitValid) {
                alignRayToHit(x, &ray, hit.normal);
                HitItem item = createHit(x, &ray, hit.objPosition, hit.objNormal, tp->objectId, hit.worldToObj);
                if (item.contrib > 0.0f) response(x, &item);
                result = RESULT_CODE_RAY_ABSORBED;
            } else {
                result = RESULT_CODE_RAY_SCATTERED;
            }
        }
        float u0, u1;
        rnd2(x, idx, &dim, &u0, &u1); /* random2D */
        scatterRay(x, &ray, u0, u1);
        if (result >= 0) {
            if (x->polarized) {
                ray.lin_contrib *= ray.stokes.x;
                float s0 = ray.stokes.x;
                ray.stokes.x /= s0; ray.stokes.y /= s0; ray.stokes.z /= s0; ray.stokes.w /= s0;
            }
            if (ray.lin_contrib * expf(ray.log_contrib) <= rnd(x, idx, &dim)) {
                result = RESULT_CODE_RAY_ABSORBED;
            } else {
                ray.lin_contrib = 1.0f;
                ray.log_contrib = 0.0f;
            }
        }
        onEvent(x, &ray, result, local_idx, i, dim);
        if (result < 0) return;
    }
    onEvent(x, &ray, RESULT_CODE_MAX_ITER, local_idx, c->path_length + 1, dim);
}

/* tracer.volume.forward.glsl:231-276 */
static void vol_traceMain(Ctx* x, const orc_volume_trace_params* tp, const orc_target_params* tg,
                          uint32_t local_idx) {
    if (g_photon_steps_per_run) { vol_photonMain(x, tp, tg, local_idx); return; }
    const orc_common* c = x->c;
    uint32_t dim = 0;
    uint32_t idx = c->photon_offset + local_idx;
    x->cur_idx = idx; x->cur_dim = &dim;
    int disable_mis = (c->flags & ORC_FLAG_DISABLE_MIS) != 0;
    int disable_direct = (c->flags & ORC_FLAG_DISABLE_DIRECT) != 0;
    int DIRECT_LIGHTING = disable_mis && !disable_direct; /* :211-215 */
    int ALLOW_RESPONSE = disable_mis;                     /* :217-221 */

    Ray ray = sampleRay(x, x->global_medium, idx, &dim);
    onEvent(x, &ray, RESULT_CODE_RAY_CREATED, local_idx, 0, dim);
    /* target.sphere.glsl:80-83 */
    if (isOccludedByTarget(tg, ray.position)) {
        onEvent(x, &ray, ERROR_CODE_TRACE_ABORT, local_idx, 0, dim);
        return;
    }
    if (!disable_direct && !disable_mis) {
        TargetSample directHit = intersectTarget(tg, ray.position, ray.direction);
        vol_createResponse(x, tp, ray, &directHit, ray.direction, 1.0f, 0);
    }
    int result = vol_trace(x, tp, tg, &ray, idx, &dim, DIRECT_LIGHTING);
    onEvent(x, &ray, result, local_idx, 1, dim);
    if (result < 0) return;
    for (uint32_t i = 0; i < c->path_length; ++i) {
        float u0, u1;
        rnd2(x, idx, &dim, &u0, &u1); /* random2D */
        scatterRay(x, &ray, u0, u1);
        result = vol_trace(x, tp, tg, &ray, idx, &dim, ALLOW_RESPONSE);
        onEvent(x, &ray, result, local_idx, i + 2, dim);
        if (result < 0) return;
    }
    onEvent(x, &ray, RESULT_CODE_MAX_ITER, local_idx, c->path_length + 2, dim);
}

/* ------------------------------------------------------------------------ */
/* SCENE: geometry, ray query restatement                                   */
/* ------------------------------------------------------------------------ */

typedef struct { float lo[3], hi[3]; uint32_t left_first, count; } OBvhNode;

typedef struct {
    const float* vertices; /* [n][6] */
    const uint32_t* indices; /* [m][3] */
    uint32_t n_vertices, n_triangles;
    float* vcopy; uint32_t* icopy;
    OBvhNode* nodes; uint32_t n_nodes;
    uint32_t* prim; /* leaf-order permutation */
} OMesh;

typedef struct {
    uint32_t mesh;
    float o2w[12], w2o[12];
    int32_t customIndex;
    uint64_t material;
    float lo[3], hi[3]; /* world AABB */
} OInstance;

struct orc_scene {
    OMesh* meshes; uint32_t n_meshes;
    OInstance* inst; uint32_t n_inst;
};

static void invert34(const float m[12], float out[12]) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double id = 1.0 / det;
    double r[9] = {(e * i - f * h) * id, (c * h - b * i) * id, (b * f - c * e) * id,
                   (f * g - d * i) * id, (a * i - c * g) * id, (c * d - a * f) * id,
                   (d * h - e * g) * id, (b * g - a * h) * id, (a * e - b * d) * id};
    double tx = m[3], ty = m[7], tz = m[11];
    for (int r_ = 0; r_ < 3; ++r_) {
        out[r_ * 4 + 0] = (float)r[r_ * 3 + 0];
        out[r_ * 4 + 1] = (float)r[r_ * 3 + 1];
        out[r_ * 4 + 2] = (float)r[r_ * 3 + 2];
        out[r_ * 4 + 3] = (float)(-(r[r_ * 3 + 0] * tx + r[r_ * 3 + 1] * ty + r[r_ * 3 + 2] * tz));
    }
}

static void tri_bounds(const OMesh* m, uint32_t t, float lo[3], float hi[3], float cen[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    for (int v = 0; v < 3; ++v) {
        const float* p = m->vertices + 6 * (size_t)m->indices[3 * t + v];
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], p[k]);
            hi[k] = fmaxf(hi[k], p[k]);
        }
    }
    for (int k = 0; k < 3; ++k) cen[k] = 0.5f * (lo[k] + hi[k]);
}

/* oracle-private BVH: median split on the longest centroid axis, leaf <= 4.
 * Deliberately a different builder than the product's binned-SAH one. */
static void obvh_build(OMesh* m, uint32_t node, uint32_t first, uint32_t count) {
    OBvhNode* n = &m->nodes[node];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 3; ++k) { n->lo[k] = INFINITY; n->hi[k] = -INFINITY; }
    for (uint32_t i = first; i < first + count; ++i) {
        float lo[3], hi[3], cen[3];
        tri_bounds(m, m->prim[i], lo, hi, cen);
        for (int k = 0; k < 3; ++k) {
            n->lo[k] = fminf(n->lo[k], lo[k]); n->hi[k] = fmaxf(n->hi[k], hi[k]);
            clo[k] = fminf(clo[k], cen[k]); chi[k] = fmaxf(chi[k], cen[k]);
        }
    }
    if (count <= 4) { n->left_first = first; n->count = count; return; }
    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    float split = 0.5f * (clo[axis] + chi[axis]);
    uint32_t i = first, j = first + count;
    while (i < j) {
        float lo[3], hi[3], cen[3];
        tri_bounds(m, m->prim[i], lo, hi, cen);
        if (cen[axis] < split) ++i;
        else { --j; uint32_t t = m->prim[i]; m->prim[i] = m->prim[j]; m->prim[j] = t; }
    }
    uint32_t nleft = i - first;
    if (nleft == 0 || nleft == count) nleft = count / 2;
    uint32_t l = m->n_nodes; m->n_nodes += 2;
    n->left_first = l; n->count = 0;
    obvh_build(m, l, first, nleft);
    obvh_build(m, l + 1, first + nleft, count - nleft);
}

orc_scene* orc_scene_create(const orc_mesh* meshes, uint32_t n_meshes,
                            const orc_instance* instances, uint32_t n_instances) {
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->n_meshes = n_meshes; s->n_inst = n_instances;
    s->meshes = (OMesh*)calloc(n_meshes, sizeof(OMesh));
    s->inst = (OInstance*)calloc(n_instances, sizeof(OInstance));
    for (uint32_t k = 0; k < n_meshes; ++k) {
        OMesh* m = &s->meshes[k];
        m->n_vertices = meshes[k].n_vertices; m->n_triangles = meshes[k].n_triangles;
        m->vcopy = (float*)malloc(sizeof(float) * 6 * m->n_vertices);
        m->icopy = (uint32_t*)malloc(sizeof(uint32_t) * 3 * m->n_triangles);
        memcpy(m->vcopy, meshes[k].vertices, sizeof(float) * 6 * m->n_vertices);
        memcpy(m->icopy, meshes[k].indices, sizeof(uint32_t) * 3 * m->n_triangles);
        m->vertices = m->vcopy; m->indices = m->icopy;
        m->prim = (uint32_t*)malloc(sizeof(uint32_t) * m->n_triangles);
        for (uint32_t t = 0; t < m->n_triangles; ++t) m->prim[t] = t;
        m->nodes = (OBvhNode*)calloc(2 * (size_t)m->n_triangles + 1, sizeof(OBvhNode));
        m->n_nodes = 1;
        obvh_build(m, 0, 0, m->n_triangles);
    }
    for (uint32_t k = 0; k < n_instances; ++k) {
        OInstance* in = &s->inst[k];
        in->mesh = instances[k].mesh;
        memcpy(in->o2w, instances[k].transform, sizeof(float) * 12);
        invert34(in->o2w, in->w2o);
        in->customIndex = instances[k].customIndex;
        in->material = instances[k].material;
        const OMesh* m = &s->meshes[in->mesh];
        for (int a = 0; a < 3; ++a) { in->lo[a] = INFINITY; in->hi[a] = -INFINITY; }
        for (uint32_t v = 0; v < m->n_vertices; ++v) {
            const float* p = m->vertices + 6 * (size_t)v;
            for (int a = 0; a < 3; ++a) {
                float w = in->o2w[a * 4 + 3] + (in->o2w[a * 4 + 0] * p[0] + in->o2w[a * 4 + 1] * p[1] +
                                                in->o2w[a * 4 + 2] * p[2]);
                in->lo[a] = fminf(in->lo[a], w); in->hi[a] = fmaxf(in->hi[a], w);
            }
        }
        for (int a = 0; a < 3; ++a) { /* conservative pad */
            float pad = 1e-5f * fmaxf(fabsf(in->lo[a]), fabsf(in->hi[a])) + 1e-30f;
            in->lo[a] -= pad; in->hi[a] += pad;
        }
    }
    return s;
}

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (uint32_t k = 0; k < s->n_meshes; ++k) {
        free(s->meshes[k].vcopy); free(s->meshes[k].icopy);
        free(s->meshes[k].prim); free(s->meshes[k].nodes);
    }
    free(s->meshes); free(s->inst); free(s);
}

typedef struct { int kx, ky, kz; float Sx, Sy, Sz; vec3 org; } WoopRay;

static WoopRay woop_setup(vec3 o, vec3 d) {
    WoopRay w;
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int kz = 0;
    if (ay > ax) kz = 1;
    if (az > (kz == 0 ? ax : ay)) kz = 2;
    int kx = kz + 1; if (kx == 3) kx = 0;
    int ky = kx + 1; if (ky == 3) ky = 0;
    if (get3(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    w.kx = kx; w.ky = ky; w.kz = kz;
    w.Sx = get3(d, kx) / get3(d, kz);
    w.Sy = get3(d, ky) / get3(d, kz);
    w.Sz = 1.0f / get3(d, kz);
    w.org = o;
    return w;
}

/* watertight ray/triangle test; updates *t,*b1,*b2 (weights of v1, v2) */
static int woop_tri(const WoopRay* w, vec3 p0, vec3 p1, vec3 p2, float* t_io, float* b1, float* b2) {
    vec3 A = v_sub(p0, w->org), B = v_sub(p1, w->org), C = v_sub(p2, w->org);
    float Akz = get3(A, w->kz), Bkz = get3(B, w->kz), Ckz = get3(C, w->kz);
    float Ax = get3(A, w->kx) - w->Sx * Akz, Ay = get3(A, w->ky) - w->Sy * Akz;
    float Bx = get3(B, w->kx) - w->Sx * Bkz, By = get3(B, w->ky) - w->Sy * Bkz;
    float Cx = get3(C, w->kx) - w->Sx * Ckz, Cy = get3(C, w->ky) - w->Sy * Ckz;
    float U = Cx * By - Cy * Bx;
    float V = Ax * Cy - Ay * Cx;
    float W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        double CxBy = (double)Cx * (double)By, CyBx = (double)Cy * (double)Bx;
        U = (float)(CxBy - CyBx);
        double AxCy = (double)Ax * (double)Cy, AyCx = (double)Ay * (double)Cx;
        V = (float)(AxCy - AyCx);
        double BxAy = (double)Bx * (double)Ay, ByAx = (double)By * (double)Ax;
        W = (float)(BxAy - ByAx);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    float Az = w->Sz * Akz, Bz = w->Sz * Bkz, Cz = w->Sz * Ckz;
    float T = (U * Az + V * Bz) + W * Cz;
    float rcpDet = 1.0f / det;
    float t = T * rcpDet;
    if (!(t > 0.0f && t < *t_io)) return 0;
    *t_io = t;
    *b1 = V * rcpDet;
    *b2 = W * rcpDet;
    return 1;
}

static int slab_test(const float lo[3], const float hi[3], vec3 o, vec3 inv, float tmax) {
    float t0 = 0.0f, t1 = tmax;
    float ox[3] = {o.x, o.y, o.z}, iv[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
        float ta = (lo[a] - ox[a]) * iv[a];
        float tb = (hi[a] - ox[a]) * iv[a];
        float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
        /* NaN (0*inf) must not reject: fmaxf/fminf ignore NaN operands */
        t0 = fmaxf(t0, tn);
        t1 = fminf(t1, tf * 1.0000004f);
    }
    return t0 <= t1;
}

typedef struct { int valid; uint32_t instance, prim; float t, b1, b2; } QueryHit;

/* diagnostic: histogram of BVH nodes visited per query (bins of 8, last bin = overflow) */
static uint64_t g_node_hist[64];
void orc_node_histogram(uint64_t out[64], int reset) {
    memcpy(out, g_node_hist, sizeof(g_node_hist));
    if (reset) memset(g_node_hist, 0, sizeof(g_node_hist));
}

static QueryHit ray_query(const orc_scene* s, vec3 origin, vec3 dir, float tmax, int brute,
                          Counters* cnt) {
    QueryHit q;
    memset(&q, 0, sizeof(q));
    q.t = tmax;
    uint64_t nodes_before = cnt ? cnt->stats[ST_NODES] : 0;
    if (cnt) cnt->stats[ST_QUERIES]++;
    vec3 winv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    for (uint32_t k = 0; k < s->n_inst; ++k) {
        const OInstance* in = &s->inst[k];
        if (!brute && !slab_test(in->lo, in->hi, origin, winv, q.t)) continue;
        if (cnt) cnt->stats[ST_INSTANCES]++;
        const float* M = in->w2o;
        vec3 o = V3(M[3] + (M[0] * origin.x + M[1] * origin.y + M[2] * origin.z),
                    M[7] + (M[4] * origin.x + M[5] * origin.y + M[6] * origin.z),
                    M[11] + (M[8] * origin.x + M[9] * origin.y + M[10] * origin.z));
        vec3 d = V3(M[0] * dir.x + M[1] * dir.y + M[2] * dir.z,
                    M[4] * dir.x + M[5] * dir.y + M[6] * dir.z,
                    M[8] * dir.x + M[9] * dir.y + M[10] * dir.z);
        WoopRay w = woop_setup(o, d);
        const OMesh* m = &s->meshes[in->mesh];
        if (brute) {
            for (uint32_t t = 0; t < m->n_triangles; ++t) {
                const uint32_t* ix = m->indices + 3 * (size_t)t;
                const float* a = m->vertices + 6 * (size_t)ix[0];
                const float* b = m->vertices + 6 * (size_t)ix[1];
                const float* c = m->vertices + 6 * (size_t)ix[2];
                if (cnt) cnt->stats[ST_TRIS]++;
                if (woop_tri(&w, V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]),
                             &q.t, &q.b1, &q.b2)) {
                    q.valid = 1; q.instance = k; q.prim = t;
                }
            }
            continue;
        }
        vec3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        uint32_t stack[64];
        int sp = 0;
        stack[sp++] = 0;
        while (sp > 0) {
            const OBvhNode* n = &m->nodes[stack[--sp]];
            if (cnt) cnt->stats[ST_NODES]++;
            if (!slab_test(n->lo, n->hi, o, inv, q.t)) continue;
            if (n->count > 0) {
                for (uint32_t i = 0; i < n->count; ++i) {
                    uint32_t t = m->prim[n->left_first + i];
                    const uint32_t* ix = m->indices + 3 * (size_t)t;
                    const float* a = m->vertices + 6 * (size_t)ix[0];
                    const float* b = m->vertices + 6 * (size_t)ix[1];
                    const float* c = m->vertices + 6 * (size_t)ix[2];
                    if (cnt) cnt->stats[ST_TRIS]++;
                    if (woop_tri(&w, V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]),
                                 V3(c[0], c[1], c[2]), &q.t, &q.b1, &q.b2)) {
                        q.valid = 1; q.instance = k; q.prim = t;
                    }
                }
            } else if (sp < 62) {
                stack[sp++] = n->left_first;
                stack[sp++] = n->left_first + 1;
            }
        }
    }
    if (cnt && q.valid) cnt->stats[ST_HITS]++;
    if (cnt) {
        uint64_t nv = cnt->stats[ST_NODES] - nodes_before;
        uint64_t b = nv / 8 > 63 ? 63 : nv / 8;
#pragma omp atomic
        g_node_hist[b]++;
    }
    return q;
}

int orc_scene_query(const orc_scene* s, const float origin[3], const float dir[3], float tmax,
                    int brute_force, float out[5]) {
    QueryHit q = ray_query(s, V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]), tmax,
                           brute_force, NULL);
    out[0] = (float)q.instance; out[1] = (float)q.prim; out[2] = q.t; out[3] = q.b1; out[4] = q.b2;
    return q.valid;
}

/* ------------------------------------------------------------------------ */
/* scene.intersect.glsl / scatter.surface.glsl / ray.surface.glsl           */
/* ------------------------------------------------------------------------ */

typedef struct {
    int valid;
    uint64_t mat_inside, mat_outside;
    int inward;
    int customId;
    uint32_t flags;
    vec3 worldPos, rayNrm;
    vec3 objPos, objNrm, objDir;
    mat3 worldToObj;
} SurfaceHit;

/* scene.intersect.glsl:23-99 */
static int processRayQuery(const orc_scene* s, const Ray* ray, const QueryHit* q, SurfaceHit* hit) {
    memset(hit, 0, sizeof(*hit));
    hit->valid = q->valid;
    if (!hit->valid) return RESULT_CODE_RAY_MISSED;
    const OInstance* in = &s->inst[q->instance];
    const OMesh* m = &s->meshes[in->mesh];
    float bx = q->b1, by = q->b2;
    hit->customId = in->customIndex;
    const uint32_t* ix = m->indices + 3 * (size_t)q->prim;
    const float* a = m->vertices + 6 * (size_t)ix[0];
    const float* b = m->vertices + 6 * (size_t)ix[1];
    const float* c = m->vertices + 6 * (size_t)ix[2];
    vec3 p0 = V3(a[0], a[1], a[2]), p1 = V3(b[0], b[1], b[2]), p2 = V3(c[0], c[1], c[2]);
    vec3 n0 = V3(a[3], a[4], a[5]), n1v = V3(b[3], b[4], b[5]), n2v = V3(c[3], c[4], c[5]);
    vec3 e1 = v_sub(p1, p0), e2 = v_sub(p2, p0);
    hit->objPos = V3(p0.x + fmaf(bx, e1.x, by * e2.x), p0.y + fmaf(bx, e1.y, by * e2.y),
                     p0.z + fmaf(bx, e1.z, by * e2.z));
    hit->objNrm = v_cross(e1, e2);
    vec3 n1 = v_sub(n1v, n0), n2 = v_sub(n2v, n0);
    vec3 intNrm = V3(n0.x + fmaf(bx, n1.x, by * n2.x), n0.y + fmaf(bx, n1.y, by * n2.y),
                     n0.z + fmaf(bx, n1.z, by * n2.z));
    hit->objNrm = v_scale(hit->objNrm, signBit(v_dot(hit->objNrm, intNrm)));
    hit->objNrm = v_normalize(hit->objNrm);

    const float* W = in->w2o; /* row-major 3x4; GLSL mat3(world2Obj) columns: */
    hit->worldToObj.c0 = V3(W[0], W[4], W[8]);
    hit->worldToObj.c1 = V3(W[1], W[5], W[9]);
    hit->worldToObj.c2 = V3(W[2], W[6], W[10]);
    hit->objDir = v_normalize(m3_mul(hit->worldToObj, ray->direction));
    hit->inward = v_dot(hit->objDir, hit->objNrm) <= 0.0f;

    uint32_t flagsIn = 0, flagsOut = 0;
    if (in->material != 0) {
        const MaterialGLSL* mat = (const MaterialGLSL*)(uintptr_t)in->material;
        hit->mat_inside = mat->inside; hit->mat_outside = mat->outside;
        flagsIn = mat->flagsInwards; flagsOut = mat->flagsOutwards;
    }
    hit->flags = hit->inward ? flagsIn : flagsOut;
    uint64_t medium = hit->inward ? hit->mat_outside : hit->mat_inside;
    if (ray->medium != medium) return ERROR_CODE_MEDIA_MISMATCH;

    /* vec3(objNrm * world2Obj): row vector times matrix */
    vec3 wn = V3(v_dot(hit->objNrm, hit->worldToObj.c0), v_dot(hit->objNrm, hit->worldToObj.c1),
                 v_dot(hit->objNrm, hit->worldToObj.c2));
    wn = v_normalize(wn);
    hit->rayNrm = v_scale(wn, 2.0f * (float)hit->inward - 1.0f);

    const float* M = in->o2w;
    vec3 p = hit->objPos;
    hit->worldPos.x = M[3] + fmaf(M[0], p.x, fmaf(M[1], p.y, M[2] * p.z));
    hit->worldPos.y = M[7] + fmaf(M[4], p.x, fmaf(M[5], p.y, M[6] * p.z));
    hit->worldPos.z = M[11] + fmaf(M[8], p.x, fmaf(M[9], p.y, M[10] * p.z));
    return RESULT_CODE_SUCCESS;
}

typedef struct { float n_in, n_tr, r_s, r_p; } Reflectance;

static Reflectance fresnelCore(float cos_i_in, float n_i, float n_t) {
    /* scatter.surface.glsl:23-50 */
    float cos_i = f_clamp(cos_i_in, -1.0f, 1.0f);
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
void orc_fresnel(float cos_i_signed, float n_i, float n_t, float out[2]) {
    Reflectance r = fresnelCore(cos_i_signed, n_i, n_t);
    out[0] = r.r_s; out[1] = r.r_p;
}

/* scatter.surface.glsl:21-51 */
static Reflectance fresnelReflect(Ctx* x, const Ray* ray, const SurfaceHit* hit) {
    float cos_i = v_dot(ray->direction, hit->rayNrm);
    uint64_t otherMed = hit->inward ? hit->mat_inside : hit->mat_outside;
    float n_t = 1.0f;
    if (otherMed != 0) {
        const MediumGLSL* m = (const MediumGLSL*)(uintptr_t)otherMed;
        float u_t = normalize_lambda(m, ray->wavelength);
        n_t = lookUp(m->n, u_t, 1.0f, &x->cnt);
    }
    return fresnelCore(cos_i, ray->constants.n, n_t);
}

/* ray.surface.glsl:22-36 */
static vec3 offsetRay(vec3 p, vec3 n) {
    int ox = (int)(256.0f * n.x), oy = (int)(256.0f * n.y), oz = (int)(256.0f * n.z);
    vec3 pi = V3(i2f(f2i(p.x) + ((p.x < 0.0f) ? -ox : ox)), i2f(f2i(p.y) + ((p.y < 0.0f) ? -oy : oy)),
                 i2f(f2i(p.z) + ((p.z < 0.0f) ? -oz : oz)));
    return V3(fabsf(p.x) < (1.0f / 32.0f) ? p.x + (1.0f / 65536.0f) * n.x : pi.x,
              fabsf(p.y) < (1.0f / 32.0f) ? p.y + (1.0f / 65536.0f) * n.y : pi.y,
              fabsf(p.z) < (1.0f / 32.0f) ? p.z + (1.0f / 65536.0f) * n.z : pi.z);
}

/* ray.surface.glsl:44-55 */
static void crossBorder(Ctx* x, Ray* ray, const SurfaceHit* hit) {
    ray->position = offsetRay(hit->worldPos, v_neg(hit->rayNrm));
    uint64_t medium = hit->inward ? hit->mat_inside : hit->mat_outside;
    ray->medium = medium;
    ray->constants = lookUpMedium(medium, ray->wavelength, &x->cnt);
}

/* GLSL reflect / refract */
static vec3 g_reflect(vec3 I, vec3 N) { return v_sub(I, v_scale(N, 2.0f * v_dot(N, I))); }
static vec3 g_refract(vec3 I, vec3 N, float eta) {
    float d = v_dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
    return v_sub(v_scale(I, eta), v_scale(N, eta * d + sqrtf(k)));
}

/* ray.surface.glsl:65-73 (+ :147-158 polarized) */
static void reflectRayIS(Ctx* x, Ray* ray, const SurfaceHit* hit, const Reflectance* refl) {
    ray->position = offsetRay(hit->worldPos, hit->rayNrm);
    ray->direction = v_normalize(g_reflect(ray->direction, hit->rayNrm));
    if (x->polarized) ray->stokes = polarizerApply(refl->r_p, refl->r_s, ray->stokes);
}
/* ray.surface.glsl:81-92 (+ :159-166) */
static void reflectRay(Ctx* x, Ray* ray, const SurfaceHit* hit, const Reflectance* refl) {
    ray->position = offsetRay(hit->worldPos, hit->rayNrm);
    ray->direction = v_normalize(g_reflect(ray->direction, hit->rayNrm));
    float s = refl->r_s, p = refl->r_p;
    ray->lin_contrib *= 0.5f * (s * s + p * p);
    if (x->polarized) ray->stokes = polarizerApply(refl->r_p, refl->r_s, ray->stokes);
}
static void transmitPol(Ctx* x, Ray* ray, const Reflectance* refl) {
    if (!x->polarized) return;
    float eta = refl->n_in / refl->n_tr;
    float t_s = refl->r_s + 1.0f;
    float t_p = (refl->r_p + 1.0f) * eta;
    ray->stokes = polarizerApply(t_p, t_s, ray->stokes);
}
/* ray.surface.glsl:102-112 (+ :193-209) */
static void transmitRayIS(Ctx* x, Ray* ray, const SurfaceHit* hit, const Reflectance* refl) {
    crossBorder(x, ray, hit);
    float eta = refl->n_in / refl->n_tr;
    ray->direction = v_normalize(g_refract(ray->direction, hit->rayNrm, eta));
    transmitPol(x, ray, refl);
}
/* ray.surface.glsl:121-132 (+ :210-218) */
static void transmitRay(Ctx* x, Ray* ray, const SurfaceHit* hit, const Reflectance* refl) {
    crossBorder(x, ray, hit);
    float eta = refl->n_in / refl->n_tr;
    ray->direction = v_normalize(g_refract(ray->direction, hit->rayNrm, eta));
    float s = refl->r_s, p = refl->r_p;
    ray->lin_contrib *= 1.0f - 0.5f * (s * s + p * p);
    transmitPol(x, ray, refl);
}

/* ------------------------------------------------------------------------ */
/* target_guide.sphere.glsl                                                 */
/* ------------------------------------------------------------------------ */

typedef struct { vec3 dir; float dist, prob; } GuideSample;

static void guideCone(const orc_guide_params* g, vec3 observer, float* d_out, vec3* viewDir,
                      float* cosMin, float* prob) {
    /* target_guide.sphere.glsl:17-31 == :55-69 */
    vec3 center = V3(g->position[0], g->position[1], g->position[2]);
    float d = v_distance(center, observer);
    *viewDir = v_normalize(v_sub(center, observer));
    float sinMax = g->radius / d;
    float sin2Max = sinMax * sinMax;
    float cm = 1.0f - sqrtf(fmaxf(1.0f - sin2Max, 0.0f));
    if (sin2Max < 0.00068523f) cm = 0.5f * sin2Max;
    float p = 1.0f / (TWO_PI_F * cm);
    p *= (float)(d > g->radius);
    *d_out = d; *cosMin = cm; *prob = p;
}

static int g_guide_flat = 0; /* 1: DiskTargetGuide, 2: FlatTargetGuide (ORC_FLAG_GUIDE_DISK / _FLAT) */

/* target_guide.common.glsl:10-32 */
static GuideSample guideSampleFromPoint(vec3 observer, vec3 targetPos, vec3 targetNrm, float targetProb) {
    vec3 dir = v_sub(targetPos, observer);
    float d2 = v_dot(dir, dir);
    dir = v_normalize(dir);
    float cosNormal = v_dot(dir, targetNrm);
    float prob = targetProb * d2 / fabsf(cosNormal);
    if (isinf(prob)) prob = 0.0f;
    prob *= (float)(cosNormal < 0.0f);
    GuideSample s;
    s.dir = dir; s.dist = v_distance(targetPos, observer); s.prob = prob;
    return s;
}
/* target_guide.common.glsl:34-51 */
static GuideSample guideSampleFromDir(vec3 dir, float dist, vec3 targetNrm, float targetProb) {
    float cosNormal = v_dot(dir, targetNrm);
    float prob = targetProb * dist * dist / fabsf(cosNormal);
    if (isinf(prob)) prob = 0.0f;
    prob *= (float)(cosNormal < 0.0f);
    GuideSample s;
    s.dir = dir; s.dist = dist; s.prob = prob;
    return s;
}
/* target_guide.disk.glsl:15-23 / target_guide.flat.glsl:17-29 */
static GuideSample sampleGuidePlane(Ctx* x, const orc_guide_params* g, vec3 observer, uint32_t idx, uint32_t* dim) {
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    vec3 localPos = g_guide_flat == 1 ? v_scale(sampleUnitDisk(u0, u1), g->radius)
                                      : V3(g->width * (u0 - 0.5f), g->height * (u1 - 0.5f), 0.0f);
    vec3 pos = v_add(m3_mul(targetObjToWorld(g->objToWorld), localPos), V3(g->position[0], g->position[1], g->position[2]));
    return guideSampleFromPoint(observer, pos, V3(g->normal[0], g->normal[1], g->normal[2]), g->prob);
}
/* target_guide.disk.glsl:25-44 / target_guide.flat.glsl:31-48 */
static GuideSample evalGuidePlane(const orc_guide_params* g, vec3 observer, vec3 direction) {
    mat3 w2o = m3_transpose(targetObjToWorld(g->objToWorld));
    vec3 localObs = m3_mul(w2o, v_sub(observer, V3(g->position[0], g->position[1], g->position[2])));
    vec3 localDir = m3_mul(w2o, direction);
    float t = -localObs.z / localDir.z;
    vec3 localHit = v_add(localObs, v_scale(localDir, t));
    int valid = t > 0.0f && !isinf(t) && !isnan(t);
    if (g_guide_flat == 1) valid = valid && sqrtf(localHit.x * localHit.x + localHit.y * localHit.y) <= g->radius;
    else valid = valid && 2.0f * fabsf(localHit.x) <= g->width && 2.0f * fabsf(localHit.y) <= g->height;
    return guideSampleFromDir(direction, t, V3(g->normal[0], g->normal[1], g->normal[2]), g->prob * (valid ? 1.0f : 0.0f));
}

/* target_guide.sphere.glsl:15-51 */
static GuideSample sampleTargetGuide(Ctx* x, const orc_guide_params* g, vec3 observer,
                                     uint32_t idx, uint32_t* dim) {
    if (g_guide_flat) return sampleGuidePlane(x, g, observer, idx, dim);
    float d, cosMin, prob;
    vec3 viewDir;
    guideCone(g, observer, &d, &viewDir, &cosMin, &prob);
    float cosTheta = 1.0f - cosMin * rnd(x, idx, dim);
    float sinTheta = sqrtf(fmaxf(1.0f - cosTheta * cosTheta, 0.0f));
    float phi = TWO_PI_F * rnd(x, idx, dim);
    vec3 local = V3(sinTheta * sinf(phi), sinTheta * cosf(phi), cosTheta);
    GuideSample s;
    s.dir = m3_mul(createLocalCOSY(viewDir), local);
    s.dist = d + g->radius;
    s.prob = prob;
    return s;
}
/* target_guide.sphere.glsl:53-81 */
static GuideSample evalTargetGuide(const orc_guide_params* g, vec3 observer, vec3 direction) {
    if (g_guide_flat) return evalGuidePlane(g, observer, direction);
    float d, cosMin, prob;
    vec3 viewDir;
    guideCone(g, observer, &d, &viewDir, &cosMin, &prob);
    float cosDir = v_dot(viewDir, direction);
    prob *= (float)(cosMin >= 1.0f - cosDir);
    GuideSample s;
    s.dir = direction; s.dist = d + g->radius; s.prob = prob;
    return s;
}
void orc_eval_guide(const orc_guide_params* g, const float obs[3], const float dir[3],
                    float out[2]) {
    GuideSample s = evalTargetGuide(g, V3(obs[0], obs[1], obs[2]), V3(dir[0], dir[1], dir[2]));
    out[0] = s.dist; out[1] = s.prob;
}
void orc_sample_guide(const orc_guide_params* g, const float obs[3], float u0, float u1,
                      float out[5]) {
    /* same arithmetic as sampleTargetGuide with explicit random numbers */
    float d, cosMin, prob;
    vec3 viewDir;
    guideCone(g, V3(obs[0], obs[1], obs[2]), &d, &viewDir, &cosMin, &prob);
    float cosTheta = 1.0f - cosMin * u0;
    float sinTheta = sqrtf(fmaxf(1.0f - cosTheta * cosTheta, 0.0f));
    float phi = TWO_PI_F * u1;
    vec3 dir = m3_mul(createLocalCOSY(viewDir), V3(sinTheta * sinf(phi), sinTheta * cosf(phi), cosTheta));
    out[0] = dir.x; out[1] = dir.y; out[2] = dir.z; out[3] = d + g->radius; out[4] = prob;
}

/* ------------------------------------------------------------------------ */
/* scene.traverse.glsl / tracer.scene.forward.glsl                          */
/* ------------------------------------------------------------------------ */

typedef struct {
    const orc_scene_trace_params* tp;
    const orc_guide_params* guide;
    const orc_scene* scene;
    int brute;
} SceneArgs;

/* scene.traverse.glsl:33-68 */
static void scn_createResponse(Ctx* x, Ray ray, const SurfaceHit* hit, const Reflectance* surface,
                               int absorb) {
    if (!absorb) {
        if (x->c->flags & ORC_FLAG_HIT_USE_TRANSMITTED) {
            transmitRay(x, &ray, hit, surface);
        } else {
            vec3 dir = ray.direction;
            transmitRay(x, &ray, hit, surface);
            ray.direction = dir;
        }
    }
    HitItem item = createHit(x, &ray, hit->objPos, hit->objNrm, hit->customId, hit->worldToObj);
    if (item.contrib > 0.0f) response(x, &item);
}

/* scene.traverse.glsl:73-154 */
static int processHit(Ctx* x, const SceneArgs* a, Ray* ray, const SurfaceHit* hit, uint32_t idx,
                      uint32_t* dim, int allowResponse) {
    int targetId = a->tp->targetId;
    int result = propagateRayToHit(x, ray, hit->worldPos, hit->rayNrm, &x->prop);
    if (result < 0) return result;
    result = RESULT_CODE_RAY_HIT;
    int isAbs = (hit->flags & MATERIAL_BLACK_BODY_BIT) != 0;
    int isTarget = (hit->flags & MATERIAL_DETECTOR_BIT) != 0;
    int volBorder = (hit->flags & MATERIAL_VOLUME_BORDER_BIT) != 0;
    int canTransmit = (hit->flags & MATERIAL_NO_TRANSMIT_BWD_BIT) == 0;
    int canReflect = (hit->flags & MATERIAL_NO_REFLECT_BWD_BIT) == 0;
    Reflectance surface = fresnelReflect(x, ray, hit);
    if (allowResponse && isTarget && (targetId < 0 || hit->customId == targetId)) {
        scn_createResponse(x, *ray, hit, &surface, isAbs);
        result = RESULT_CODE_RAY_DETECTED;
    }
    if (isAbs) return RESULT_CODE_RAY_ABSORBED;
    if (!(x->c->flags & ORC_FLAG_DISABLE_VOLUME_BORDER)) {
        if (volBorder) {
            crossBorder(x, ray, hit);
            return RESULT_CODE_VOLUME_HIT;
        }
    }
    if (!(x->c->flags & ORC_FLAG_DISABLE_TRANSMISSION)) {
        float r = 0.5f * (surface.r_s * surface.r_s + surface.r_p * surface.r_p);
        if (canReflect && canTransmit) {
            if (rnd(x, idx, dim) < r) reflectRayIS(x, ray, hit, &surface);
            else transmitRayIS(x, ray, hit, &surface);
        } else if (canReflect) {
            reflectRay(x, ray, hit, &surface);
        } else if (canTransmit) {
            transmitRay(x, ray, hit, &surface);
        } else {
            result = RESULT_CODE_RAY_ABSORBED;
        }
    } else {
        if (canReflect) reflectRay(x, ray, hit, &surface);
        else result = RESULT_CODE_RAY_ABSORBED;
    }
    return result;
}

/* scene.traverse.glsl:160-180 */
static void processShadowRay(Ctx* x, const SceneArgs* a, Ray ray, const SurfaceHit* hit) {
    int targetId = a->tp->targetId;
    int isTarget = (hit->flags & MATERIAL_DETECTOR_BIT) != 0;
    int correctId = targetId < 0 || targetId == hit->customId;
    if (!hit->valid || !correctId || !isTarget) return;
    if (propagateRayToHit(x, &ray, hit->worldPos, hit->rayNrm, &x->prop) < 0) return;
    Reflectance surface = fresnelReflect(x, &ray, hit);
    int black = (hit->flags & MATERIAL_BLACK_BODY_BIT) != 0;
    scn_createResponse(x, ray, hit, &surface, black);
}

/* scene.traverse.glsl:185-214 */
static void traceShadowRay(Ctx* x, const SceneArgs* a, Ray ray, vec3 dir, float dist, float weight) {
    scatterRayIS(x, &ray, dir);
    QueryHit q = ray_query(a->scene, ray.position, dir, dist, a->brute, &x->cnt);
    SurfaceHit hit;
    if (processRayQuery(a->scene, &ray, &q, &hit) < 0) return;
    ray.lin_contrib *= weight;
    processShadowRay(x, a, ray, &hit);
}

/* scene.traverse.glsl:234-269 */
static void scn_sampleTargetMIS(Ctx* x, const SceneArgs* a, const Ray* ray, uint32_t idx,
                                uint32_t* dim) {
    uint64_t med = ray->medium;
    vec3 obs = ray->position, dir = ray->direction;
    float pPP;
    float u0, u1;
    rnd2(x, idx, dim, &u0, &u1); /* random2D */
    vec3 dirPhase = scatter(med, dir, u0, u1, &pPP, &x->cnt);
    GuideSample targetSample = sampleTargetGuide(x, a->guide, obs, idx, dim);
    float pTT = targetSample.prob;
    GuideSample phaseSample = evalTargetGuide(a->guide, obs, dirPhase);
    float pTP = phaseSample.prob;
    float pPT = scatterProb(med, dir, targetSample.dir, &x->cnt);
    float wTarget = pTT * pPT / (pTT * pTT + pPT * pPT);
    float wPhase = pPP * pPP / (pPP * pPP + pTP * pTP);
    traceShadowRay(x, a, *ray, dirPhase, phaseSample.dist, wPhase);
    traceShadowRay(x, a, *ray, targetSample.dir, targetSample.dist, wTarget);
}

/* scene.traverse.glsl:280-352 */
static int scn_trace(Ctx* x, const SceneArgs* a, Ray* ray, SurfaceHit* hit, uint32_t idx,
                     uint32_t* dim, int allowResponse) {
    int mis = !(x->c->flags & ORC_FLAG_DISABLE_MIS);
    x->cnt.stats[ST_STEPS]++;
    hit->valid = 0;
    if (isRayBad(ray)) return ERROR_CODE_RAY_BAD;
    vec3 dir = v_normalize(ray->direction);
    float dist = sampleScatterLength(ray, &x->prop, rnd(x, idx, dim));
    float sampledDist = dist;
    int mis_target = 0;
    if (mis) {
        GuideSample gs = evalTargetGuide(a->guide, ray->position, dir);
        mis_target = allowResponse && gs.prob > 0.0f;
        if (mis_target) {
            mis_target = gs.dist > dist;
            dist = fmaxf(gs.dist, dist);
        }
    }
    QueryHit q = ray_query(a->scene, ray->position, dir, dist, a->brute, &x->cnt);
    int queryResult = processRayQuery(a->scene, ray, &q, hit);
    if (hit->valid) dist = v_distance(ray->position, hit->worldPos);
    if (mis) {
        if (mis_target && hit->valid && dist > sampledDist && queryResult >= 0) {
            processShadowRay(x, a, *ray, hit);
            hit->valid = 0;
            dist = sampledDist;
        }
    }
    int propResult = propagateRay(ray, dist, &x->prop);
    updateRayIS(ray, dist, &x->prop, hit->valid);
    return queryResult <= ERROR_CODE_MAX_VALUE ? queryResult : propResult;
}

/* scene.traverse.glsl:361-397 */
static int processInteraction(Ctx* x, const SceneArgs* a, Ray* ray, const SurfaceHit* hit,
                              uint32_t idx, uint32_t* dim, int allowResponse, int last) {
    if (hit->valid) return processHit(x, a, ray, hit, idx, dim, allowResponse);
    if (!last) {
        if (!(x->c->flags & ORC_FLAG_DISABLE_MIS)) scn_sampleTargetMIS(x, a, ray, idx, dim);
        float u0, u1;
        rnd2(x, idx, dim, &u0, &u1); /* random2D */
        scatterRay(x, ray, u0, u1);
    }
    return RESULT_CODE_RAY_SCATTERED;
}

/* tracer.scene.forward.glsl:58-106 */
static void scn_traceMain(Ctx* x, const SceneArgs* a, uint32_t local_idx) {
    const orc_common* c = x->c;
    uint32_t dim = 0;
    uint32_t idx = c->photon_offset + local_idx;
    uint64_t medium = (uint64_t)a->tp->sceneMedium[0] | ((uint64_t)a->tp->sceneMedium[1] << 32);
    x->cur_idx = idx; x->cur_dim = &dim;
    Ray ray = sampleRay(x, medium, idx, &dim);
    onEvent(x, &ray, RESULT_CODE_RAY_CREATED, local_idx, 0, dim);
    int allowResponse = !(c->flags & ORC_FLAG_DISABLE_DIRECT);
    const uint32_t perRun = g_photon_steps_per_run; /* != 0: ScenePhotonTracer */
    for (uint32_t i = 1; i <= c->path_length; ++i) {
        int last = i == c->path_length && !perRun; /* photon loop: processInteraction(..., last = false) */
        /* run r of the photon tracer starts at RNG dimension r * rng_per_run (Push.dim) */
        if (perRun && i > 1 && (i - 1) % perRun == 0) dim = ((i - 1) / perRun) * g_photon_rng_per_run;
        SurfaceHit hit;
        int result = scn_trace(x, a, &ray, &hit, idx, &dim, allowResponse);
        if (result >= 0)
            result = processInteraction(x, a, &ray, &hit, idx, &dim, allowResponse, last);
        if (perRun && result >= 0) { /* tracer.scene.photon.loop.glsl:68-83 */
            if (x->polarized) {
                ray.lin_contrib *= ray.stokes.x;
                float s0 = ray.stokes.x;
                ray.stokes.x /= s0; ray.stokes.y /= s0; ray.stokes.z /= s0; ray.stokes.w /= s0;
            }
            if (ray.lin_contrib * expf(ray.log_contrib) <= rnd(x, idx, &dim)) {
                result = RESULT_CODE_RAY_ABSORBED;
            } else {
                ray.lin_contrib = 1.0f;
                ray.log_contrib = 0.0f;
            }
        }
        onEvent(x, &ray, result, local_idx, i, dim);
        if (result < 0) return;
        if (!(c->flags & ORC_FLAG_DISABLE_MIS)) allowResponse = result != RESULT_CODE_RAY_SCATTERED;
        else allowResponse = 1;
    }
    onEvent(x, &ray, RESULT_CODE_MAX_ITER, local_idx, c->path_length + 1, dim);
}

/* ------------------------------------------------------------------------ */
/* drivers: per-block partial histograms + estimator.reduce.glsl            */
/* ------------------------------------------------------------------------ */

typedef void (*photon_fn)(Ctx*, const void*, uint32_t);

/* estimator.reduce.glsl:17-35: one thread per bin walks the histograms in order with a compensated (Kahan) sum, then * norm
 * (`precise` in the shader: no reassociation, no contraction -- volatile keeps the compiler from folding the compensation) */
void orc_reduce(const float* hists, uint32_t n_hist, uint32_t n_bins, float norm, float* out) {
    for (size_t bin = 0; bin < n_bins; ++bin) {
        float result = 0.0f, comp = 0.0f;
        for (uint32_t n = 0; n < n_hist; ++n) {
            volatile float y = hists[(size_t)n * n_bins + bin] - comp;
            volatile float t = result + y;
            comp = (t - result) - y;
            result = t;
        }
        out[bin] = result * norm;
    }
}

static int run_blocks(const orc_common* c, Ctx* proto, photon_fn fn, const void* args,
                      int n_threads, float* hist_out, uint64_t stats[20], float* track,
                      uint32_t track_events) {
    uint32_t bs = c->block_size ? c->block_size : 128;
    uint32_t n_blocks = (c->batch_size + bs - 1) / bs;
    if (n_blocks == 0) n_blocks = 1; /* trace.py:406-407 */
    size_t nb = c->n_bins;
    float* slabs = (float*)calloc((size_t)n_blocks * nb, sizeof(float));
    uint64_t total[ST_COUNT];
    memset(total, 0, sizeof(total));
    if (!slabs) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        Ctx x = *proto;
        memset(&x.cnt, 0, sizeof(x.cnt));
        x.track = track;
        x.track_events = track_events;
#pragma omp for schedule(dynamic, 4)
        for (long b = 0; b < (long)n_blocks; ++b) {
            x.hist = slabs + (size_t)b * nb;
            uint32_t first = (uint32_t)b * bs;
            uint32_t last = first + bs;
            if (last > c->batch_size) last = c->batch_size;
            for (uint32_t i = first; i < last; ++i) fn(&x, args, i);
        }
#pragma omp critical
        for (int k = 0; k < ST_COUNT; ++k) total[k] += x.cnt.stats[k];
    }
    orc_reduce(slabs, n_blocks, (uint32_t)nb, c->normalization, hist_out);
    free(slabs);
    if (stats) memcpy(stats, total, sizeof(uint64_t) * ST_COUNT);
    return 0;
}

typedef struct { const orc_volume_trace_params* tp; const orc_target_params* tg; } VolArgs;
static void vol_photon(Ctx* x, const void* a, uint32_t i) {
    const VolArgs* v = (const VolArgs*)a;
    vol_traceMain(x, v->tp, v->tg, i);
}
static void scn_photon(Ctx* x, const void* a, uint32_t i) { scn_traceMain(x, (const SceneArgs*)a, i); }

int orc_trace_volume(const orc_common* c, const orc_volume_trace_params* t,
                     const orc_target_params* target, int n_threads, float* hist_out,
                     uint64_t stats[20], float* track) {
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = c;
    x.polarized = (c->flags & ORC_FLAG_POLARIZED) != 0;
    x.use_global_medium = 1;
    g_target_inner = (c->flags & ORC_FLAG_TARGET_INNER) != 0;
    g_target_flat = (c->flags & ORC_FLAG_TARGET_DISK) ? 1 : ((c->flags & ORC_FLAG_TARGET_FLAT) ? 2 : 0);
    g_target_miss_inf = (c->flags & ORC_FLAG_TARGET_MISS_INFINITE) != 0;
    x.global_medium = (uint64_t)t->medium[0] | ((uint64_t)t->medium[1] << 32);
    x.prop = t->propagation;
    VolArgs a = {t, target};
    return run_blocks(c, &x, vol_photon, &a, n_threads, hist_out, stats, track, c->path_length + 3);
}

/* response.replay.glsl:37-52: item idx of the input queue (response.queue.glsl layout) goes to
 * response(hit, idx, dim = 0); c->batch_size = capacity of the queue */
typedef struct { const uint32_t* queue; uint32_t capacity; int polarized; } ReplayArgs;
static void replay_item(Ctx* x, const void* a, uint32_t idx) {
    const ReplayArgs* r = (const ReplayArgs*)a;
    uint32_t count = r->queue[0] < r->capacity ? r->queue[0] : r->capacity;
    if (idx >= count) return;
    const float* col = (const float*)(r->queue + 1) + idx;
    const size_t cap = r->capacity;
    HitItem h;
    memset(&h, 0, sizeof(h));
#define GET() (*col); col += cap
    h.position.x = GET(); h.position.y = GET(); h.position.z = GET();
    h.direction.x = GET(); h.direction.y = GET(); h.direction.z = GET();
    h.normal.x = GET(); h.normal.y = GET(); h.normal.z = GET();
    if (r->polarized) {
        h.stokes.x = GET(); h.stokes.y = GET(); h.stokes.z = GET(); h.stokes.w = GET();
        h.polRef.x = GET(); h.polRef.y = GET(); h.polRef.z = GET();
    }
    h.wavelength = GET(); h.time = GET(); h.contrib = GET();
#undef GET
    h.objectId = *(const int32_t*)col;
    uint32_t dim = 0;
    x->cur_idx = idx;
    x->cur_dim = &dim;
    response(x, &h);
}

int orc_replay(const orc_common* c, const uint32_t* queue, uint32_t capacity, int polarized,
               int n_threads, float* hist_out, uint64_t stats[20]) {
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = c;
    x.polarized = polarized != 0;
    ReplayArgs a = {queue, capacity, polarized};
    return run_blocks(c, &x, replay_item, &a, n_threads, hist_out, stats, NULL, 0);
}

void orc_sobol_fill(uint32_t seed, uint32_t base_stream, uint32_t n_streams, uint32_t first_dim, uint32_t n_samples,
                    uint32_t sample_dim, float* out) {
    orc_common c;
    memset(&c, 0, sizeof(c));
    c.flags = ORC_FLAG_RNG_SOBOL;
    c.philox.key[0] = seed;
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = &c;
    for (uint32_t st = 0; st < n_streams; ++st)
        for (uint32_t k = 0; k < n_samples; ++k) {
            uint32_t dim = first_dim + k; /* random_s / random2D_s: the sink addresses every sample itself */
            size_t at = (size_t)st * n_samples + k;
            if (sample_dim == 2) rnd2(&x, base_stream + st, &dim, &out[2 * at], &out[2 * at + 1]);
            else out[at] = rnd(&x, base_stream + st, &dim);
        }
}

/* target.sample.glsl:19-83 */
static void putSample(float* o, vec3 observer, const TargetSample* ts, int checkDir, vec3 direction) {
    mat3 m = ts->worldToObj;
    float expD = v_distance(ts->position, observer);
    vec3 expPos = v_add(m3_mul(m, ts->position), ts->offset);
    vec3 expNrm = v_normalize(m3_mul(m, ts->normal));
    int error = 0;
    error += (int)(fabsf(expD - ts->dist) > 1e-5f);
    error += (int)(v_length(v_sub(expPos, ts->objPosition)) > 1e-5f) << 1;
    error += (int)(v_length(v_sub(expNrm, ts->objNormal)) > 1e-5f) << 2;
    if (checkDir) {
        vec3 sampleDir = v_normalize(v_sub(ts->position, observer));
        error += (int)(v_length(v_sub(sampleDir, direction)) > 1e-5f) << 3;
    }
    o[0] = ts->position.x; o[1] = ts->position.y; o[2] = ts->position.z;
    o[3] = ts->normal.x; o[4] = ts->normal.y; o[5] = ts->normal.z;
    o[6] = ts->prob;
    uint32_t valid = ts->valid ? 1u : 0u;
    memcpy(&o[7], &valid, 4);
    int32_t e = error;
    memcpy(&o[8], &e, 4);
}
static void probeObserver(Ctx* x, uint32_t i, uint32_t* dim, const float lo[3], const float hi[3], vec3* observer, vec3* direction) {
    float a = rnd(x, i, dim), b = rnd(x, i, dim), c = rnd(x, i, dim);
    *observer = V3(f_mix(lo[0], hi[0], a), f_mix(lo[1], hi[1], b), f_mix(lo[2], hi[2], c));
    float u0, u1;
    rnd2(x, i, dim, &u0, &u1);
    *direction = sphericalToCartessian(TWO_PI_F * u0, 2.0f * u1 - 1.0f); /* util.sample.glsl:84-88 */
}
void orc_probe_target(const orc_philox_params* philox, uint32_t flags, const orc_target_params* target,
                      const float box_min[3], const float box_max[3], uint32_t n, float* out) {
    orc_common c;
    memset(&c, 0, sizeof(c));
    c.philox = *philox;
    c.flags = flags & ORC_FLAG_RNG_SOBOL;
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = &c;
    g_target_inner = (flags & ORC_FLAG_TARGET_INNER) != 0;
    g_target_flat = (flags & ORC_FLAG_TARGET_DISK) ? 1 : ((flags & ORC_FLAG_TARGET_FLAT) ? 2 : 0);
    g_target_miss_inf = (flags & ORC_FLAG_TARGET_MISS_INFINITE) != 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t dim = 0;
        vec3 observer, direction;
        probeObserver(&x, i, &dim, box_min, box_max, &observer, &direction);
        TargetSample s = sampleTarget(&x, target, observer, i, &dim);
        TargetSample h = intersectTarget(target, observer, direction);
        float* o = out + (size_t)i * 25;
        o[0] = observer.x; o[1] = observer.y; o[2] = observer.z;
        o[3] = direction.x; o[4] = direction.y; o[5] = direction.z;
        putSample(o + 6, observer, &s, 0, direction);
        putSample(o + 15, observer, &h, 1, direction);
        uint32_t occ = isOccludedByTarget(target, observer) ? 1u : 0u;
        memcpy(&o[24], &occ, 4);
    }
    g_target_inner = 0; g_target_flat = 0; g_target_miss_inf = 0;
}
void orc_probe_guide(const orc_philox_params* philox, uint32_t flags, const orc_guide_params* guide,
                     const float box_min[3], const float box_max[3], uint32_t n, float* out) {
    orc_common c;
    memset(&c, 0, sizeof(c));
    c.philox = *philox;
    c.flags = flags & ORC_FLAG_RNG_SOBOL;
    Ctx x;
    memset(&x, 0, sizeof(x));
    x.c = &c;
    g_guide_flat = (flags & ORC_FLAG_GUIDE_DISK) ? 1 : ((flags & ORC_FLAG_GUIDE_FLAT) ? 2 : 0);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t dim = 0;
        vec3 observer, direction;
        probeObserver(&x, i, &dim, box_min, box_max, &observer, &direction);
        GuideSample s = sampleTargetGuide(&x, guide, observer, i, &dim);
        GuideSample e = evalTargetGuide(guide, observer, direction);
        float* o = out + (size_t)i * 13;
        o[0] = observer.x; o[1] = observer.y; o[2] = observer.z;
        o[3] = s.dir.x; o[4] = s.dir.y; o[5] = s.dir.z; o[6] = s.dist; o[7] = s.prob;
        o[8] = e.dir.x; o[9] = e.dir.y; o[10] = e.dir.z; o[11] = e.dist; o[12] = e.prob;
    }
    g_guide_flat = 0;
}

int orc_trace_scene(const orc_common* c, const orc_scene_trace_params* t,
                    const orc_guide_params* guide, const orc_scene* scene, int n_threads,
                    float* hist_out, uint64_t stats[20], float* track) {
    Ctx x;
    g_guide_flat = (c->flags & ORC_FLAG_GUIDE_DISK) ? 1 : ((c->flags & ORC_FLAG_GUIDE_FLAT) ? 2 : 0);
    memset(&x, 0, sizeof(x));
    x.c = c;
    x.polarized = (c->flags & ORC_FLAG_POLARIZED) != 0;
    x.use_global_medium = 0;
    x.prop = t->propagation;
    SceneArgs a;
    a.tp = t; a.guide = guide; a.scene = scene;
    a.brute = (c->flags & ORC_FLAG_BRUTE_FORCE) != 0;
    if (!(c->flags & ORC_FLAG_DISABLE_MIS) && !guide) return -2;
    return run_blocks(c, &x, scn_photon, &a, n_threads, hist_out, stats, track, c->path_length + 3);
}
