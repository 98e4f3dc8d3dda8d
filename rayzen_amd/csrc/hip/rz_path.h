// rz_path.h -- the per-pixel path loop of FS:668-773 as a state machine.
//
// The shader nests everything: bounce loop -> calculateLighting -> light loop
// -> shadowVisibility -> up to 32 closest-hit traversals.  Compiled as
// written that is five traversal call sites, each inlined.  Here a path is
// a small record that is advanced from one closest-hit query to the next:
//
//     ray ready --trace--> advance(hit) --> next ray ready | sample finished
//
// with two kinds of ray in flight: a path segment (primary / bounce) or one
// iteration of a shadow query.  There is a single trace call site, lanes of
// a wave can be in different phases while sharing the traversal loop, and the
// same record can be parked in memory and picked up by another lane (the
// compacting launch of rz_kernels.hip).
// The arithmetic, statement by statement, is the shader's.
#pragma once
#include "rz_trace.h"

namespace rz {

#ifdef RZ_PROF
#define RZ_T0() const unsigned long long rzt0_ = __builtin_amdgcn_s_memtime()
#define RZ_T1(c, k) (c).t[k] += __builtin_amdgcn_s_memtime() - rzt0_
#else
#define RZ_T0() do { } while (0)
#define RZ_T1(c, k) do { } while (0)
#endif

enum : int { MODE_SEGMENT = 0, MODE_SHADOW = 1, MODE_DONE = 2 };

struct Path {
    // per pixel
    v2 uv;              // gl_FragCoord.xy / resolution (FS:669)
    float fragSum;      // gl_FragCoord.x + gl_FragCoord.y
    v3 color;           // running sum over samples (FS:672; divide/clamp of FS:772-773 are the resolve's job)
    float ior;          // currentIor: declared OUTSIDE the sample loop (FS:674), so it carries over
    // per sample
    int samp, sampEnd;
    v2 seed;            // FS:688
    int bounce;
    v3 throughput;
    // the ray to trace next
    v3 o, d;
    int mode;
    // bounce-0 direct lighting in progress (FS:569-663), parked while shadow rays fly
    v3 hp, hn, pdir;    // the surface point being lit, its normal, the direction the path arrived with
    int hmat;
    v3 lacc;            // finalColor / specAccum
    int li;             // light index
    float vis, traveled, maxDist;
    int iter;
    // the (at most two) terms this sample adds to the pixel's colour, in the shader's order: FS:717 then FS:709.
    // The one-lane-per-pixel kernel adds them to `color` on the spot and never reads these; the
    // one-lane-per-sample kernel parks them in LDS and replays the adds in sample order (rz_kernels.hip).
    v3 addLight, addSky;
    int usedIor;        // this sample read (and possibly changed) currentIor: FS:727-742 executed
    int gflag;          // pooled paths of a transparent scene (GMODE 2): 1 = P.ior is the value the sequential shader would hand this sample (it may scatter at glass)
};

// FS:204-212 + FS:688-692: the sample's seed and its camera ray direction from (uv, fragCoord.x + y, sample index).
struct CameraRay { v2 seed; v3 d; };
__device__ __forceinline__ CameraRay camera_ray(const float* __restrict__ invProj, const float* __restrict__ invView, const v2 uv, const float fragSum, const int samp) {
    CameraRay r;
    const float sf = ((fragSum + (float)samp) + 1.0f);
    r.seed.x = uv.x * sf;
    r.seed.y = uv.y * sf;
    v2 s1; s1.x = r.seed.x + 1.0f; s1.y = r.seed.y + 1.0f;
    const float jx = rand_(r.seed) * 0.00002f, jy = rand_(s1) * 0.00002f;
    const float ux = uv.x + jx, uy = uv.y + jy;
    const float cx = ux * 2.0f - 1.0f, cy = uy * 2.0f - 1.0f;
    const float* ip = invProj;
    const float ex = ((ip[0] * cx + ip[4] * cy) + ip[8] * -1.0f) + ip[12] * 1.0f;
    const float ey = ((ip[1] * cx + ip[5] * cy) + ip[9] * -1.0f) + ip[13] * 1.0f;
    const v3 world = xform_dir(invView, mk3(ex, ey, -1.0f));
    r.d = normalize(world);
    return r;
}
template <bool COUNT>
__device__ __forceinline__ void begin_sample(const KParams& K, Path& P, Tally& c) {
    if (COUNT) c.samples += 1;
    const CameraRay r = camera_ray(K.invProj, K.invView, P.uv, P.fragSum, P.samp);
    P.seed = r.seed;
    P.o = mk3(K.camPos[0], K.camPos[1], K.camPos[2]);
    P.d = r.d;
    P.throughput = mk3(1.0f, 1.0f, 1.0f);
    P.bounce = 0;
    P.mode = MODE_SEGMENT;
    P.addLight = mk3(0.0f, 0.0f, 0.0f);
    P.addSky = mk3(0.0f, 0.0f, 0.0f);
    P.usedIor = 0;
}

__device__ __forceinline__ void end_sample(Path& P) {
    P.samp += 1;
    P.mode = MODE_DONE;     // caller restarts with begin_sample if samp < sampEnd
}

// FS:533-535
__device__ __forceinline__ v3 fresnel_schlick(float cosTheta, v3 F0) {
    const float p = pow5_(1.0f - cosTheta);
    return mk3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}
// FS:537-539
__device__ __forceinline__ v3 reflect_ray(v3 i, v3 n) { return i - n * (2.0f * dot(i, n)); }
// FS:558-567
__device__ __forceinline__ bool refract_dir(v3 incident, v3 normal, float eta, v3& refr) {
    const float cosi = clamp_(dot(-incident, normal), -1.0f, 1.0f);
    const float sint2 = fmax_(0.0f, 1.0f - cosi * cosi);
    const float k = 1.0f - (eta * eta) * sint2;
    if (k < 0.0f) return false;
    const float w = eta * cosi - __builtin_sqrtf(k);
    refr = normalize(incident * eta + normal * w);
    return true;
}
// FS:192-202.  The first half -- two hash numbers, acos, two sin/cos pairs in binary64 -- depends on the seed alone.
__device__ __forceinline__ v3 hemisphere_local(v2 seed) {
    const float u = rand_(seed);
    v2 s1; s1.x = seed.x + 1.0f; s1.y = seed.y + 1.0f;
    const float v = rand_(s1);
    const float theta = acos_(__builtin_sqrtf(1.0f - u));
    const float phi = (2.0f * 3.14159f) * v;
    float st, ct, sp, cp;
    sincos_(theta, st, ct);
    sincos_(phi, sp, cp);
    return mk3(st * cp, st * sp, ct);
}
// ... and the same OUT OF LINE, for the opaque kernels (round 5, profiles/r05_regs/).  The draw runs for a twelfth of the scatters
// (every other diffuse scatter draws the constant of bounce 0) and is 1 000 binary64 instructions with some forty live values:
// inlined it shares the path's register allocation -- behind a call it has its own frame, the kernel is 2 400 instructions
// shorter and C2 1.8 % faster (c2close -0.9 %, C5 -1.5 %).  The transparent kernels keep it inline: there the call's save /
// restore lands on paths that are live across it more often (c2g +2.2 %, RayZen's scene at 64 spp +4.2 %).  Same operations either way.
// (Flavour 1's draw is ~250 binary32 instructions around calls of the shared sin / cos core: inline everywhere -- behind a call
//  of its own C2 ran 10.78 ms against 10.10.)
#ifndef RZ_HEMI_OUT_OF_LINE
#define RZ_HEMI_OUT_OF_LINE (RZ_MATH_FLAVOUR == 0)
#endif
#ifndef RZ_HEMI_OUT_OF_LINE_GLASS
#define RZ_HEMI_OUT_OF_LINE_GLASS 0
#endif
static __device__ __attribute__((noinline)) v3 hemisphere_local_call(v2 seed) { return hemisphere_local(seed); }
__device__ __forceinline__ v3 hemisphere_world(v3 normal, v3 dir) {
    const v3 up = (__builtin_fabsf(normal.y) < 0.99f) ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
    const v3 tangent = normalize(cross(up, normal));
    const v3 bitangent = cross(normal, tangent);
    return normalize((tangent * dir.x + bitangent * dir.y) + normal * dir.z);
}
// The shader seeds this with tempseed = seed * float(bounce * bounce) * 12793.46 + float(bounce) * 1423.34 (FS:696): at
// bounce 0 that is (+0, +0) for every pixel and every sample (seed > 0), so every first scatter draws the SAME local
// direction.  It is computed once per context by rz_hemi0_kernel with this very function (K.hemi0) and the wave skips
// the five binary64 evaluations -- 1.4 % of the C2 frame -- when all its lanes carry the zero seed (compared by bit pattern).
template <bool GLASS>
__device__ __forceinline__ v3 random_hemisphere_direction(const KParams& K, v3 normal, v2 seed) {
    const bool zero = __float_as_uint(seed.x) == 0u && __float_as_uint(seed.y) == 0u;
    v3 dir = mk3(K.hemi0[0], K.hemi0[1], K.hemi0[2]);
    if (!zero) dir = ((RZ_HEMI_OUT_OF_LINE && !GLASS) || (RZ_HEMI_OUT_OF_LINE_GLASS && GLASS)) ? hemisphere_local_call(seed) : hemisphere_local(seed);
    return hemisphere_world(normal, dir);
}

// Set up the shadow query of light P.li for the parked surface point
// (FS:578-588 transparent branch, FS:622-635 opaque branch).
// GLASS = false: the caller guarantees that no triangle of the scene uses a transparent material, so every
// `transparency > 0` branch of the shader is dead and is compiled out (fewer live registers, less code).
template <bool COUNT, bool GLASS>
__device__ __forceinline__ void start_light(const KParams& K, Path& P, Tally& c) {
    RZ_T0();
    const DevLight L = K.lights[P.li];
    const DevMaterial M = K.materials[P.hmat];
    if (COUNT) c.light_fetches += 1;
    v3 dir;
    if (L.posdir[3] == 1.0f) {
        const v3 lv = mk3(L.posdir[0], L.posdir[1], L.posdir[2]) - P.hp;
        const float dist = fmax_(length(lv), 0.001f);
        dir = (GLASS && M.transparency > 0.0f) ? div3(lv, dist) : normalize(lv);
        P.maxDist = dist;
    } else {
        dir = normalize(mk3(L.posdir[0], L.posdir[1], L.posdir[2]));
        P.maxDist = 1e30f;
    }
    P.o = P.hp + dir * 0.001f;
    P.d = dir;
    P.vis = 1.0f;
    P.traveled = 0.0f;
    P.iter = 0;
    P.mode = MODE_SHADOW;
    RZ_T1(c, 5);
}

// The light `li` is visible with `vis` from the surface point (hp, hn, material hmat) along lightDir: its term (FS:589-607 /
// FS:636-659), or false where the shader returns without one.  Values in, values out -- no path state by reference -- so that the
// same code can sit behind a call (RZ_SHADE_NOINLINE, measured in profiles/r05_flavour/).
template <bool GLASS>
__device__ __forceinline__ bool shade_term(const DevLight* __restrict__ lights, const DevMaterial* __restrict__ materials, const v3 cam,
                                           const v3 hp, const v3 normal, const v3 lightDir, const float maxDist, const float vis,
                                           const int li, const int hmat, v3& term) {
    const DevLight L = lights[li];
    const DevMaterial M = materials[hmat];
    const v3 viewDir = normalize(cam - hp);   // FS:714
    const v3 albedo = mk3(M.albedo[0], M.albedo[1], M.albedo[2]);
    const v3 lcolor = mk3(L.color[0], L.color[1], L.color[2]);
    float attenuation = (L.posdir[3] == 1.0f) ? L.power / (maxDist * maxDist) : L.power;
    attenuation *= vis;
    if (GLASS && M.transparency > 0.0f) {
        const float NdotL = fmax_(dot(normal, lightDir), 0.0f);
        if (NdotL <= 0.0f) return false;
        const float f0 = pow2_((1.0f - M.ior) / (1.0f + M.ior));
        const v3 H = normalize(lightDir + viewDir);
        const float NdotH = fmax_(dot(normal, H), 0.0f);
        const float cosTheta = fmax_(dot(H, viewDir), 0.0f);
        const v3 F = fresnel_schlick(cosTheta, mk3(f0, f0, f0));
        const float rough = fmax_(M.roughness, 0.02f);
        const float a = rough * rough;
        const float a2 = a * a;
        const float dDen = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
        const float D = a2 / ((3.14159f * dDen) * dDen + 1e-6f);
        const float k = (rough + 1.0f) * (rough + 1.0f) / 8.0f;
        const float NdotV = fmax_(dot(normal, viewDir), 0.0f);
        const float Gv = NdotV / ((NdotV * (1.0f - k) + k) + 1e-6f);
        const float Gl = NdotL / ((NdotL * (1.0f - k) + k) + 1e-6f);
        const float denom = fmax_((4.0f * NdotL) * NdotV, 1e-4f);
        const v3 spec = div3(((F * D) * Gv) * Gl, denom);
        term = ((spec * lcolor) * attenuation) * NdotL;
    } else {
        const v3 F0 = mk3(mix_(0.04f, albedo.x, M.metallic), mix_(0.04f, albedo.y, M.metallic),
                          mix_(0.04f, albedo.z, M.metallic));
        const v3 halfwayDir = normalize(lightDir + viewDir);
        const float NdotL = fmax_(dot(normal, lightDir), 0.0f);
        const float NdotV = fmax_(dot(normal, viewDir), 0.0f);
        const v3 F = fresnel_schlick(fmax_(dot(halfwayDir, viewDir), 0.0f), F0);
        const float alpha = M.roughness * M.roughness;
        const float alpha2 = alpha * alpha;
        const float ndh = dot(normal, halfwayDir);
        const float denom = (ndh * ndh) * (alpha2 - 1.0f) + 1.0f;
        const float D = alpha2 / ((3.14159f * denom) * denom);
        const float k = (M.roughness + 1.0f) * (M.roughness + 1.0f) / 8.0f;
        float G = NdotV / (NdotV * (1.0f - k) + k);
        G *= NdotL / (NdotL * (1.0f - k) + k);
        const float denomSpec = fmax_((4.0f * NdotV) * NdotL, 0.0001f);
        const v3 specular = div3((F * D) * G, denomSpec);
        const v3 oneMinusF = mk3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
        const v3 diffuse = div3((oneMinusF * albedo) * NdotL, 3.14159f);
        const v3 t = ((diffuse + specular) * lcolor) * attenuation;
        term = mk3(fmax_(0.0f, t.x), fmax_(0.0f, t.y), fmax_(0.0f, t.z));
    }
    return true;
}
// Behind a call the BRDF term costs the opaque kernels 1.6-1.8 % (C2 10.12 -> 10.30 ms, C3 38.5 -> 39.1) and gains the
// transparent ones 1.0-1.6 % (c2g 18.72 -> 18.53, glassbunny 14.78 -> 14.54): RZ_SHADE_NOINLINE 2 = the transparent kernels only.
#ifndef RZ_SHADE_NOINLINE
#define RZ_SHADE_NOINLINE 2
#endif
struct ShadeOut { v3 term; int ok; };
template <bool GLASS>
static __device__ __attribute__((noinline)) ShadeOut shade_term_call(const DevLight* lights, const DevMaterial* materials, float cx, float cy, float cz,
                                                                     v3 hp, v3 normal, v3 lightDir, float maxDist, float vis, int li, int hmat) {
    ShadeOut o;
    o.term = mk3(0.0f, 0.0f, 0.0f);
    o.ok = shade_term<GLASS>(lights, materials, mk3(cx, cy, cz), hp, normal, lightDir, maxDist, vis, li, hmat, o.term) ? 1 : 0;
    return o;
}
// The light P.li is visible with P.vis: add its term.
template <bool GLASS>
__device__ __forceinline__ void shade_light(const KParams& K, Path& P) {
    if (RZ_SHADE_NOINLINE == 1 || (RZ_SHADE_NOINLINE == 2 && GLASS)) {
        const ShadeOut o = shade_term_call<GLASS>(K.lights, K.materials, K.camPos[0], K.camPos[1], K.camPos[2], P.hp, P.hn, P.d, P.maxDist, P.vis, P.li, P.hmat);
        if (o.ok) P.lacc = P.lacc + o.term;
    } else {
        v3 term;
        if (shade_term<GLASS>(K.lights, K.materials, mk3(K.camPos[0], K.camPos[1], K.camPos[2]), P.hp, P.hn, P.d, P.maxDist, P.vis, P.li, P.hmat, term))
            P.lacc = P.lacc + term;
    }
}

// The PREFIX of a sample that is about to read currentIor for the first time (its first transparent scatter, FS:723-746).
// Everything a sample computes before that point -- the camera ray, the queries so far, the lighting of its first hit, shadow
// walks through glass included -- does not depend on the currentIor it was started with; what follows does.  The speculating
// launch (rz_kernels.hip: render_samples_group<GLASS>) computes a sample a second time when the value it guessed turns out wrong:
// with the state at this point kept (19 dwords per lane in the resident wave's scratch, written once, read by the re-run) the
// second version starts HERE, with the other currentIor, instead of at the camera.  [field][lane]: a field of the wave is one
// 256-B line.  The counting instantiations keep the tallies of the prefix beside it.
template <bool COUNT>
__device__ __forceinline__ void snapshot_store(const KParams& K, const Path& P, const Tally& c) {
    float* const S = K.snap + (size_t)blockIdx.x * K.snapStride + (threadIdx.x & 63);
    S[0 * 64] = P.hp.x; S[1 * 64] = P.hp.y; S[2 * 64] = P.hp.z;
    S[3 * 64] = P.hn.x; S[4 * 64] = P.hn.y; S[5 * 64] = P.hn.z;
    S[6 * 64] = P.pdir.x; S[7 * 64] = P.pdir.y; S[8 * 64] = P.pdir.z;
    S[9 * 64] = P.throughput.x; S[10 * 64] = P.throughput.y; S[11 * 64] = P.throughput.z;
    S[12 * 64] = P.addLight.x; S[13 * 64] = P.addLight.y; S[14 * 64] = P.addLight.z;
    S[15 * 64] = P.seed.x; S[16 * 64] = P.seed.y;
    S[17 * 64] = __int_as_float(P.bounce); S[18 * 64] = __int_as_float(P.hmat);
    if (COUNT) {        // (the scatter that is about to run has counted itself already: the resumed run counts it again)
        const unsigned v[RZ_SNAP_TALLY] = {c.samples, c.traversals, c.tlas_nodes, c.tlas_leaf_indices, c.instances, c.blas_nodes, c.triangles,
                                           c.materials, c.light_fetches, c.scatters - 1u, c.diffuse_scatters, c.hemi_draws, c.lit_lights, c.triangles_past_u};
        for (int k = 0; k < RZ_SNAP_TALLY; ++k) S[(RZ_SNAP_FIELDS + k) * 64] = __uint_as_float(v[k]);
    }
}
// ... and back: the path stands in front of the scatter again (the caller sets P.ior, P.samp and runs scatter<.., GLASS = true>)
template <bool COUNT>
__device__ __forceinline__ void snapshot_load(const KParams& K, Path& P, Tally& c) {
    const float* const S = K.snap + (size_t)blockIdx.x * K.snapStride + (threadIdx.x & 63);
    P.hp = mk3(S[0 * 64], S[1 * 64], S[2 * 64]);
    P.hn = mk3(S[3 * 64], S[4 * 64], S[5 * 64]);
    P.pdir = mk3(S[6 * 64], S[7 * 64], S[8 * 64]);
    P.throughput = mk3(S[9 * 64], S[10 * 64], S[11 * 64]);
    P.addLight = mk3(S[12 * 64], S[13 * 64], S[14 * 64]);
    P.seed.x = S[15 * 64]; P.seed.y = S[16 * 64];
    P.bounce = __float_as_int(S[17 * 64]); P.hmat = __float_as_int(S[18 * 64]);
    P.addSky = mk3(0.0f, 0.0f, 0.0f);
    P.usedIor = 0;
    if (COUNT) {
        unsigned v[RZ_SNAP_TALLY];
        for (int k = 0; k < RZ_SNAP_TALLY; ++k) v[k] = __float_as_uint(S[(RZ_SNAP_FIELDS + k) * 64]);
        c.samples = v[0]; c.traversals = v[1]; c.tlas_nodes = v[2]; c.tlas_leaf_indices = v[3]; c.instances = v[4]; c.blas_nodes = v[5];
        c.triangles = v[6]; c.materials = v[7]; c.light_fetches = v[8]; c.scatters = v[9]; c.diffuse_scatters = v[10]; c.hemi_draws = v[11];
        c.lit_lights = v[12]; c.triangles_past_u = v[13];
    }
}

// FS:720-769: choose the next direction at the parked surface point and move on.
// GMODE (transparent scenes): 1 -- keep the state in front of the sample's first transparent scatter (snapshot_store; only where
// K.snap is set); 2 -- the path must not read currentIor at all (a pooled path of a compacting claim, whose pixel's earlier
// samples may not be through yet): it stops in front of the transparent scatter, P.usedIor tells the caller, nothing is changed.
template <bool COUNT, bool GLASS, int GMODE = 0>
__device__ __forceinline__ void scatter(const KParams& K, Path& P, Tally& c) {
    RZ_T0();
    const DevMaterial M = K.materials[P.hmat];
    if constexpr (GLASS && GMODE == 2) {
        if (M.transparency > 0.0f && !P.gflag) { P.usedIor = 1; P.mode = MODE_DONE; return; }
    }
    if (COUNT) c.scatters += 1;
    const float fb2 = (float)(P.bounce * P.bounce), fb = (float)P.bounce;
    v2 tempseed;
    tempseed.x = (P.seed.x * fb2) * 12793.46f + fb * 1423.34f;
    tempseed.y = (P.seed.y * fb2) * 12793.46f + fb * 1423.34f;
    v2 rs; rs.x = tempseed.x + (float)P.samp; rs.y = tempseed.y + fb;
    const float randVal = rand_(rs);
    const v3 hitNormal = P.hn;
    v3 dir = P.pdir;
    if (GLASS && M.transparency > 0.0f) {
        if constexpr (GMODE == 1) {
            if (K.snap != nullptr && !P.usedIor) snapshot_store<COUNT>(K, P, c);
        }
        P.usedIor = 1;
        const bool entering = dot(-dir, hitNormal) > 0.0f;
        const v3 N = entering ? hitNormal : -hitNormal;
        const float extIor = P.ior;
        const float nextIor = entering ? M.ior : 1.0f;
        const float eta = extIor / nextIor;
        const float cosi = clamp_(dot(-dir, N), 0.0f, 1.0f);
        const float F0 = pow2_((extIor - nextIor) / (extIor + nextIor));
        const float fresnel = F0 + (1.0f - F0) * pow5_(1.0f - cosi);
        v3 refr;
        if (!refract_dir(dir, N, eta, refr)) {
            dir = reflect_ray(dir, N);
            P.throughput = P.throughput * 0.98f;
        } else {
            dir = refr;
            P.ior = nextIor;
            const float tr = M.transparency;
            const v3 tint = mk3(mix_(1.0f, M.albedo[0], tr), mix_(1.0f, M.albedo[1], tr), mix_(1.0f, M.albedo[2], tr));
            const v3 tw = (tint * tr) * (1.0f - fresnel);
            P.throughput = P.throughput * mk3(clamp_(tw.x, 0.0f, 1.0f), clamp_(tw.y, 0.0f, 1.0f), clamp_(tw.z, 0.0f, 1.0f));
        }
    } else {
        if (randVal < M.reflectivity) {
            dir = reflect_ray(dir, hitNormal);
            P.throughput = P.throughput * 0.95f;
        } else {
            {
#ifdef RZ_PROF
                const unsigned long long th0_ = __builtin_amdgcn_s_memtime();
#endif
                dir = random_hemisphere_direction<GLASS>(K, hitNormal, tempseed);
                if (COUNT) { c.diffuse_scatters += 1; if (!(__float_as_uint(tempseed.x) == 0u && __float_as_uint(tempseed.y) == 0u)) c.hemi_draws += 1; }
#ifdef RZ_PROF
                c.t[8] += __builtin_amdgcn_s_memtime() - th0_;
#endif
            }
            P.throughput = P.throughput * (mk3(M.albedo[0], M.albedo[1], M.albedo[2]) * 0.4f);
        }
    }
    const float pushDir = dot(dir, hitNormal) > 0.0f ? 1.0f : -1.0f;
    P.o = P.hp + (hitNormal * pushDir) * 0.003f;
    P.d = dir;
    P.mode = MODE_SEGMENT;
    if (P.bounce > 2) {     // Russian roulette with the SAME random number (FS:764-769)
        const float p = fmax_(P.throughput.x, fmax_(P.throughput.y, P.throughput.z));
        if (randVal > p) { end_sample(P); RZ_T1(c, 7); return; }
        P.throughput = div3(P.throughput, p);
    }
    P.bounce += 1;
    if (P.bounce >= K.maxBounces) end_sample(P);
    RZ_T1(c, 7);
}

// Lighting of the parked point is complete (or there are no lights): FS:717, then scatter.
template <bool COUNT, bool GLASS, int GMODE = 0>
__device__ __forceinline__ void finish_lighting(const KParams& K, Path& P, Tally& c) {
    P.addLight = P.throughput * P.lacc;
    P.color = P.color + P.addLight;
    scatter<COUNT, GLASS, GMODE>(K, P, c);
}

// Advance a path by the result of the closest-hit query of its current ray.
template <bool COUNT, bool GLASS = true, int GMODE = 0>
__device__ __forceinline__ void advance(const KParams& K, Path& P, bool found, const HitRec& h, Tally& c) {
    if (P.mode == MODE_SEGMENT) {
        if (!found) {   // FS:705-711
            RZ_T0();
            const float t = 0.5f * (normalize(P.d).y + 1.0f);
            const v3 sky = mk3(mix_(0.15f, 0.5f, t), mix_(0.25f, 0.7f, t), mix_(0.45f, 1.0f, t));
            P.addSky = P.throughput * sky;
            P.color = P.color + P.addSky;
            end_sample(P);
            RZ_T1(c, 3);
            return;
        }
        if (COUNT) c.materials += 1;        // FS:713
        P.hp = h.p; P.hn = h.n; P.hmat = h.mat; P.pdir = P.d;
        if (P.bounce == 0) {                // FS:716-718
            const DevMaterial M = K.materials[h.mat];
            P.lacc = (GLASS && M.transparency > 0.0f)
                         ? mk3(0.0f, 0.0f, 0.0f)
                         : mk3(0.05f * M.albedo[0], 0.05f * M.albedo[1], 0.05f * M.albedo[2]);
            P.li = 0;
            if (K.nLights > 0) { start_light<COUNT, GLASS>(K, P, c); return; }
            finish_lighting<COUNT, GLASS, GMODE>(K, P, c);
            return;
        }
        scatter<COUNT, GLASS, GMODE>(K, P, c);
        return;
    }
    // MODE_SHADOW: the body of one iteration of FS:511-526
    bool done = false, lit = false;
    if (!found) { done = true; lit = true; }
    else if (h.t < 0.001f) { P.o = P.o + P.d * 0.001f; }
    else {
        P.traveled += h.t;
        if (P.traveled >= P.maxDist) { done = true; lit = true; }
        else {
            if (COUNT) c.materials += 1;
            const float tr = K.materials[h.mat].transparency;
            if (GLASS && tr > 0.0f) { P.vis *= tr; P.o = h.p + P.d * 0.001f; }
            else { P.vis = 0.0f; done = true; lit = false; }
        }
    }
    if (!done) {
        P.iter += 1;
        if (P.iter < 32 && P.vis > 0.05f) return;   // next iteration: trace again
        lit = P.vis > 0.05f;                         // FS:527
    }
    if (lit) { RZ_T0(); if (COUNT) c.lit_lights += 1; shade_light<GLASS>(K, P); RZ_T1(c, 6); }
    P.li += 1;
    if (P.li < K.nLights) { start_light<COUNT, GLASS>(K, P, c); return; }
    finish_lighting<COUNT, GLASS, GMODE>(K, P, c);
}

}  // namespace rz
