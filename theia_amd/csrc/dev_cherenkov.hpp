// The Cherenkov family of light sources (forward mode): CherenkovLightSource, CherenkovTrackLightSource, MuonTrackLightSource,
// ParticleCascadeLightSource -- shader/lightsource.cherenkov.{common,simple,track}.glsl, lightsource.particles.{common,muon,cascade}.glsl,
// random.gamma.glsl. Compiled into the sampler kernel and the unit probes only (kernels_misc.hip): `pow`, `exp` and the rejection
// loop of the gamma sampler cost the instrumented tracing kernels 50 spilled registers when they were inlined into the birth stage,
// so a tracer with one of these sources lets the sampler kernel fill its light / wavelength queues first (api.cpp: thx_tracer_run).
#pragma once
#ifndef THX_DEV
#error "include dev_trace.hpp first"
#endif

namespace thx {

// lightsource.cherenkov.common.glsl:6-46: radiance per track length and wavelength, in eV or in photons (d/dphi form)
THX_DEV float frankTammCherenkov(float n, float lambda, bool photons) {
    lambda *= 1e-3f; // nm -> um
    const float result = photons ? 7.2973525693f / (lambda * lambda) * (1.0f - (1.0f / (n * n)))
                                 : 9.04756408986352f / (lambda * lambda * lambda) * (1.0f - (1.0f / (n * n)));
    return fmaxf(result, 0.0f);
}
// lightsource.particles.common.glsl:53-62: photons per track length and wavelength, all azimuths
THX_DEV float frankTammParticle(float lambda, float n) {
    lambda *= 1e-3f;
    const float sin2_theta = 1.0f - 1.0f / (n * n);
    const float result = TWO_PI * 7.2973525693f / (lambda * lambda) * sin2_theta;
    return fmaxf(result, 0.0f);
}
// lightsource.particles.common.glsl:69-97
THX_DEV float particleSampleEmissionAngle(float n, float a, float b, float u) {
    const float cos_chev = 1.0f / n;
    const float int_lower = 1.0f - thx_exp(-b * thx_pow(1.0f - cos_chev, a));
    const float int_upper = 1.0f - thx_exp(-b * thx_pow(1.0f + cos_chev, a));
    u = u * (int_upper + int_lower) - int_lower;
    const float x = thx_pow(-thx_log(1.0f - fabsf(u)) / b, 1.0f / a);
    return cos_chev - signf(u) * x;
}
// lightsource.particles.common.glsl:103-136: the same profile on [cos_min, cos_max], with the weight f / p
THX_DEV float particleSampleEmissionAngleRange(float n, float a, float b, float u, float cos_min, float cos_max, float& contrib) {
    const float cos_chev = 1.0f / n;
    float int_upper = 1.0f - thx_exp(-b * thx_pow(fabsf(cos_max - cos_chev), a));
    float int_lower = 1.0f - thx_exp(-b * thx_pow(fabsf(cos_min - cos_chev), a));
    int_upper *= signf(cos_max - cos_chev);
    int_lower *= signf(cos_min - cos_chev);
    contrib = fabsf(int_upper - int_lower);
    contrib /= 2.0f - thx_exp(-b * thx_pow(1.0f - cos_chev, a)) - thx_exp(-b * thx_pow(1.0f + cos_chev, a));
    u = u * (int_upper - int_lower) + int_lower;
    const float x = -thx_pow(-thx_log(1.0f - fabsf(u)) / b, 1.0f / a);
    return cos_chev - signf(u) * x;
}
// lightsource.particles.common.glsl:138-154
THX_DEV float particleEvalEmissionAngle(float n, float a, float b, float cos_theta) {
    const float cos_chev = 1.0f / n;
    float norm = a * b;
    norm /= 2.0f - thx_exp(-b * thx_pow(1.0f - cos_chev, a)) - thx_exp(-b * thx_pow(1.0f + cos_chev, a));
    const float x = fabsf(cos_theta - cos_chev);
    return thx_exp(-b * thx_pow(x, a)) * thx_pow(x, a - 1.0f) * norm;
}
// random.gamma.glsl:8-31: Cheng's GA rejection sampler; alpha < 1 through alpha + 1
// (in two halves -- what is fixed by alpha, and one proposal -- so that the balanced sampler kernel of kernels_misc.hip can run the same
// arithmetic with another photon in every lane and iteration)
struct GammaGA {
    float alpha, a, b, c; // alpha: of the distribution the proposals are made for (the given one, or + 1 below 1)
    bool lifted;          // the given alpha was below 1: a sample is scaled with u^(1 / alpha) of one more draw, made first
    float invAlpha;       // 1 / the given alpha
};
THX_DEV GammaGA gammaSetup(float alpha) {
    GammaGA g;
    g.lifted = alpha < 1.0f;
    g.invAlpha = 1.0f / alpha;
    if (g.lifted) alpha += 1.0f;
    g.alpha = alpha;
    g.a = sqrtf(2.0f * alpha - 1.0f);
    g.b = alpha - thx_log(4.0f);
    g.c = alpha + 1.0f / alpha;
    return g;
}
template <class V>
THX_DEV float gammaScale(Lane<V>& L, const GammaGA& g) { return g.lifted ? thx_pow(draw(L), g.invAlpha) : 1.0f; }
// one proposal: true = accepted, x is the sample (before the scale)
template <class V>
THX_DEV bool gammaPropose(Lane<V>& L, const GammaGA& g, float& x) {
    float u0, u1;
    draw2(L, u0, u1);
    const float v = g.a * thx_log(u0 / (1.0f - u0));
    x = g.alpha * thx_exp(v);
    return g.b + g.c * v - x >= thx_log(u0 * u0 * u1) || !(u0 == u0); // (a NaN draw cannot happen; no endless loop if it ever did)
}
template <class V>
THX_DEV float sampleGamma(Lane<V>& L, float alpha) {
    const GammaGA g = gammaSetup(alpha);
    const float scale = gammaScale(L, g);
    float x;
    while (!gammaPropose(L, g, x)) {}
    return scale * x;
}
// a Cherenkov photon: on the cone of half angle acos(1 / n) around the particle's direction, linearly polarized in the plane of
// the two directions (lightsource.cherenkov.simple.glsl:37-57, lightsource.cherenkov.track.glsl:48-77)
template <class V>
THX_DEV SourceRay cherenkovRay(f3 pos, f3 particleDir, float n, float phiDraw, float time, float contrib) {
    const float cos_theta = 1.0f / n;
    const float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    const float phi = TWO_PI * phiDraw;
    const f3 localDir{sin_theta * thx_cos(phi), sin_theta * thx_sin(phi), cos_theta};
    SourceRay r;
    r.position = pos; r.direction = mul(createLocalCOSY(particleDir), localDir); r.startTime = time; r.contrib = contrib;
    r.stokes = f4{1.0f, 0.0f, 0.0f, 0.0f};
    r.polRef = f3{0.0f, 0.0f, 0.0f};
    if constexpr (V::POL) {
        r.polRef = normalize(cross(r.direction, particleDir));
        r.stokes = f4{1.0f, 1.0f, 0.0f, 0.0f};
    }
    return r;
}
THX_DEV f3 mix3(f3 a, f3 b, float t) { return f3{mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)}; }

// lightsource.particles.cascade.glsl:46-68: the photon of a cascade whose depth sample (gamma distributed, in units of b_long) is given
template <class V>
THX_DEV SourceRay cascadeRay(Lane<V>& L, float wavelength, float n, float gammaSample) {
    const THX_KERNARG thx_cascade_params& cascade = coldArgs().p.light.cascade;
    const f3 dir = ld3(cascade.direction);
    const float z = gammaSample * cascade.b_long;
    const f3 rayPos = ld3(cascade.startPosition) + dir * z;
    const float time = cascade.startTime + z * INV_SPEED_OF_LIGHT;
    float u0, u1;
    draw2(L, u0, u1);
    const float phi = TWO_PI * u0;
    const float cos_theta = particleSampleEmissionAngle(n, cascade.a_angular, cascade.b_angular, u1);
    const f3 rayDir = mul(createLocalCOSY(dir), sphericalToCartessian(phi, cos_theta));
    float contrib = cascade.effectiveLength;
    if ((L.a->flags & THX_FLAG_FRANK_TAMM_IS) == 0u) contrib *= frankTammParticle(wavelength, n);
    return sourceRay<V>(rayPos, rayDir, time, contrib);
}

// forward sampleLight of lightsource.cherenkov.simple.glsl:20-58, lightsource.cherenkov.track.glsl:23-78,
// lightsource.particles.muon.glsl:29-60, lightsource.particles.cascade.glsl:41-69
template <class V>
THX_DEV SourceRay sampleCherenkovFamily(Lane<V>& L, float wavelength, float n) {
    const THX_KERNARG thx_light_params& light = coldArgs().p.light;
    const uint32_t flags = L.a->flags;
    const bool photons = (flags & THX_FLAG_FRANK_TAMM_PHOTONS) != 0u;
    const bool applyFrankTamm = (flags & THX_FLAG_FRANK_TAMM_IS) == 0u;
    switch (flags & THX_FLAG_LIGHT_KIND_MASK) {
    case THX_FLAG_LIGHT_CHERENKOV: {
        const float u = draw(L);
        const f3 position = mix3(ld3(light.cherenkov.trackStart), ld3(light.cherenkov.trackEnd), u);
        const float startTime = mixf(light.cherenkov.startTime, light.cherenkov.endTime, u);
        const float contrib = TWO_PI * frankTammCherenkov(n, wavelength, photons) * light.cherenkov.trackDist;
        const float v = draw(L);
        return cherenkovRay<V>(position, ld3(light.cherenkov.trackDir), n, v, startTime, contrib);
    }
    case THX_FLAG_LIGHT_CHERENKOV_TRACK: {
        const uint32_t* track = reinterpret_cast<const uint32_t*>((uint64_t)light.cherenkovTrack.track[0] | ((uint64_t)light.cherenkovTrack.track[1] << 32));
        const uint32_t trackLength = track[0];
        const float* vertices = reinterpret_cast<const float*>(track + 1);
        float u = draw(L);
        u *= (float)trackLength;
        const uint32_t vertexIdx = min((uint32_t)floorf(u), trackLength - 1u);
        u = u - floorf(u); // fract
        const float* s = vertices + 4u * vertexIdx;
        const f3 sp{s[0], s[1], s[2]}, ep{s[4], s[5], s[6]};
        const f3 pos = mix3(sp, ep, u);
        const float time = mixf(s[3], s[7], u);
        const float v = draw(L);
        const f3 particleDir = normalize(ep - sp);
        const float segmentLength = distance(sp, ep);
        float contrib = TWO_PI * (float)trackLength * segmentLength;
        contrib *= frankTammCherenkov(n, wavelength, photons);
        return cherenkovRay<V>(pos, particleDir, n, v, time, contrib);
    }
    case THX_FLAG_LIGHT_MUON_TRACK: {
        const f3 sp = ld3(light.muonTrack.startPosition), ep = ld3(light.muonTrack.endPosition);
        const float u = draw(L);
        const f3 position = mix3(sp, ep, u);
        const float startTime = mixf(light.muonTrack.startTime, light.muonTrack.endTime, u);
        float contrib = distance(sp, ep);
        contrib *= light.muonTrack.energyScale;
        float v0, v1;
        draw2(L, v0, v1);
        const float phi = TWO_PI * v0;
        const float cos_theta = particleSampleEmissionAngle(n, light.muonTrack.a_angular, light.muonTrack.b_angular, v1);
        const f3 rayDir = mul(createLocalCOSY(normalize(ep - sp)), sphericalToCartessian(phi, cos_theta));
        if (applyFrankTamm) contrib *= frankTammParticle(wavelength, n);
        return sourceRay<V>(position, rayDir, startTime, contrib);
    }
    default: // THX_FLAG_LIGHT_CASCADE
        return cascadeRay<V>(L, wavelength, n, sampleGamma(L, light.cascade.a_long));
    }
}

} // namespace thx
