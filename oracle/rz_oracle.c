/*
 * rz_oracle.c -- CPU restatement of RayZen's fragment-shader path tracer.
 * TEST INFRASTRUCTURE ONLY.  PINNED against the reference itself: RayZen's own fragment shader run on Mesa llvmpipe
 * (oracle/glref, tests/golden/glref_*.npz, tests/test_glref.py; see rz_oracle.h).
 *
 * Every function cites the lines of RayZen/shaders/fragment_shader.glsl
 * ("FS") it follows.  Statement order, operand order and the literal
 * constants are those of the shader; the built-ins are the pinned ones of
 * rz_oracle_math.h.  Build with -ffp-contract=off.
 */
#include "rz_oracle.h"
#include "rz_oracle_math.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divs3(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* GLSL dot/cross, left to right, no fma */
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float length3(v3 a) { return sqrtf(dot3(a, a)); }
static inline v3 normalize3(v3 a) { return divs3(a, sqrtf(dot3(a, a))); }
static inline v3 ld3(const float* p) { return V3(p[0], p[1], p[2]); }

/* column-major mat4 * vec4(v, 1).xyz : ((c0*x + c1*y) + c2*z) + c3 */
static inline v3 xform_point(const float* m, v3 v) {
    return V3(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12],
              ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
              ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]);
}
/* column-major mat4 * vec4(v, 0).xyz : (c0*x + c1*y) + c2*z */
static inline v3 xform_dir(const float* m, v3 v) {
    return V3((m[0] * v.x + m[4] * v.y) + m[8] * v.z,
              (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
              (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
/* mat3(transpose(m)) * v : component i = dot(column i of m (xyz), v) */
static inline v3 xform_normal(const float* m, v3 v) {
    return V3((m[0] * v.x + m[1] * v.y) + m[2] * v.z,
              (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
              (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}

typedef struct {
    const rzo_scene* sc;
    rzo_counters cnt;
} rctx;

/* FS:188-190 */
static inline float rz_rand(v2 uv) {
    float d = uv.x * 12.9898f + uv.y * 78.233f;
    return rzo_fract(rzo_sin(d) * 43758.5453f);
}

/* FS:192-202 */
static v3 random_hemisphere_direction(v3 normal, v2 seed) {
    float u = rz_rand(seed);
    v2 s1 = {seed.x + 1.0f, seed.y + 1.0f};
    float v = rz_rand(s1);
    float theta = rzo_acos(sqrtf(1.0f - u));
    float phi = (2.0f * 3.14159f) * v;
    float st = rzo_sin(theta), ct = rzo_cos(theta), sp = rzo_sin(phi), cp = rzo_cos(phi);
    v3 dir = V3(st * cp, st * sp, ct);
    v3 up = (fabsf(normal.y) < 0.99f) ? V3(0.0f, 1.0f, 0.0f) : V3(1.0f, 0.0f, 0.0f);
    v3 tangent = normalize3(cross3(up, normal));
    v3 bitangent = cross3(normal, tangent);
    v3 r = add3(add3(scale3(tangent, dir.x), scale3(bitangent, dir.y)), scale3(normal, dir.z));
    return normalize3(r);
}

/* FS:380-388 */
static inline int intersect_aabb(v3 o, v3 invd, const float* bmin, const float* bmax, float* tmin, float* tmax) {
    float t0x = (bmin[0] - o.x) * invd.x, t0y = (bmin[1] - o.y) * invd.y, t0z = (bmin[2] - o.z) * invd.z;
    float t1x = (bmax[0] - o.x) * invd.x, t1y = (bmax[1] - o.y) * invd.y, t1z = (bmax[2] - o.z) * invd.z;
    float sx = rzo_min(t0x, t1x), sy = rzo_min(t0y, t1y), sz = rzo_min(t0z, t1z);
    float bx = rzo_max(t0x, t1x), by = rzo_max(t0y, t1y), bz = rzo_max(t0z, t1z);
    *tmin = rzo_max(rzo_max(sx, sy), sz);
    *tmax = rzo_min(rzo_min(bx, by), bz);
    return *tmax >= rzo_max(*tmin, 0.0f);
}

/* FS:391-416 */
static inline int hit_triangle(const rzo_triangle* tri, v3 o, v3 d, float* tHit, v3* hp, v3* n, int* mat, uint64_t* pastU) {
    v3 v0 = ld3(tri->v0);
    v3 edge1 = sub3(ld3(tri->v1), v0);
    v3 edge2 = sub3(ld3(tri->v2), v0);
    v3 h = cross3(d, edge2);
    float a = dot3(edge1, h);
    if (fabsf(a) < 0.0001f) return 0;
    float f = 1.0f / a;
    v3 s = sub3(o, v0);
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    ++*pastU;                                   /* (tally: tests that reach FS:403, the second half of the test) */
    v3 q = cross3(s, edge1);
    float v = f * dot3(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * dot3(edge2, q);
    if (t > 0.0001f) {
        *tHit = t;
        *hp = add3(o, scale3(d, t));
        *n = normalize3(cross3(edge1, edge2));
        *mat = tri->materialIndex;
        return 1;
    }
    return 0;
}

/* ---- optional analysis hook (tests/analysis/late_bounce_model.py): per camera path, the cost of each of its
 * closest-hit queries in call order (0 primary, 1.. shadow queries, then the bounces), in units of
 * (BLAS nodes popped)/2 + 2 x (triangles tested) + 3.  Off unless rzo_set_trace_recorder() was given a buffer. */
#define RZO_REC_CALLS 8
static uint16_t* g_rec = NULL; static int g_rec_w = 0, g_rec_h = 0, g_rec_spp = 0;
static __thread uint16_t* t_rec = NULL; static __thread int t_rec_call = 0;
void rzo_set_trace_recorder(uint16_t* buf, int width, int height, int spp) { g_rec = buf; g_rec_w = width; g_rec_h = height; g_rec_spp = spp; }
static inline void rec_sample(int x, int y, int s) {
    t_rec = NULL;
    if (g_rec && x >= 0 && y >= 0 && x < g_rec_w && y < g_rec_h && s >= 0 && s < g_rec_spp) {
        t_rec = g_rec + (((size_t)y * g_rec_w + x) * g_rec_spp + s) * RZO_REC_CALLS;
        t_rec_call = 0;
    }
}
static inline void rec_trace_impl(uint64_t dn, uint64_t dt) {
    if (t_rec && t_rec_call < RZO_REC_CALLS) { uint64_t v = dn / 2 + 2 * dt + 3; t_rec[t_rec_call++] = (uint16_t)(v > 65535 ? 65535 : v); }
}
#define rec_trace(c, n0, t0) rec_trace_impl((c)->cnt.blas_nodes - (n0), (c)->cnt.triangles - (t0))

/* FS:419-454 */
static int traverse_blas(rctx* c, v3 o, v3 d, int nodeOff, int triOff, int gTriOff,
                         float* tHitOut, v3* hpOut, v3* nOut, int* matOut) {
    const rzo_scene* sc = c->sc;
    float tHit = 1e30f;
    int hit = 0;
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    v3 invd = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    while (sp > 0) {
        int nidx = stack[--sp];
        const rzo_node* node = &sc->blas_nodes[nodeOff + nidx];
        c->cnt.blas_nodes++;
        float tmin, tmax;
        if (!intersect_aabb(o, invd, node->bmin, node->bmax, &tmin, &tmax) || tmin > tHit) continue;
        if (node->count == 0) continue;      /* empty mesh (BVH.cpp:115-118 writes a count-0 root with inverted bounds).
                                                FS:432 would treat it as internal and push nodes 0 and 1 for ever --
                                                the slab test ACCEPTS an inverted box -- so the reference hangs on it;
                                                here an empty BLAS is simply never hit. */
        if (node->count > 0) {
            for (int i = 0; i < node->count; ++i) {
                int triIdx = gTriOff + sc->blas_indices[triOff + node->leftFirst + i];
                c->cnt.triangles++;
                float t; v3 thp = V3(0, 0, 0), tn = V3(0, 0, 0); int tm = -1;
                if (hit_triangle(&sc->triangles[triIdx], o, d, &t, &thp, &tn, &tm, &c->cnt.triangles_past_u)) {
                    if (t < tHit) { tHit = t; *hpOut = thp; *nOut = tn; *matOut = tm; hit = 1; }
                }
            }
        } else {
            if (sp + 2 > 64) continue;   /* FS has no guard: stack[64] overflow is undefined there */
            stack[sp++] = node->leftFirst;
            stack[sp++] = node->leftFirst + 1;
        }
    }
    *tHitOut = tHit;
    return hit;
}

/* FS:457-503 */
static int traverse_tlas(rctx* c, v3 o, v3 d, float* tHitOut, v3* hpOut, v3* nOut, int* matOut, int* instOut) {
    const rzo_scene* sc = c->sc;
    float tHit = 1e30f;
    int hit = 0;
    c->cnt.traversals++;
    const uint64_t rec_n0 = c->cnt.blas_nodes, rec_t0 = c->cnt.triangles;
    if (sc->n_tlas_nodes == 0) { *tHitOut = tHit; return 0; }
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    v3 invd = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    while (sp > 0) {
        int nidx = stack[--sp];
        const rzo_node* node = &sc->tlas_nodes[nidx];
        c->cnt.tlas_nodes++;
        float tmin, tmax;
        if (!intersect_aabb(o, invd, node->bmin, node->bmax, &tmin, &tmax) || tmin > tHit) continue;
        if (node->count > 0) {
            for (int i = 0; i < node->count; ++i) {
                int instIdx = sc->tlas_indices[node->leftFirst + i];
                c->cnt.tlas_leaf_indices++;
                const rzo_instance* inst = &sc->instances[instIdx];
                c->cnt.instances++;
                v3 lo = xform_point(inst->inverseTransform, o);
                v3 ld = normalize3(xform_dir(inst->inverseTransform, d));
                float tLocal; v3 lhp = V3(0, 0, 0), ln = V3(0, 0, 0); int tm = -1;
                if (traverse_blas(c, lo, ld, inst->blasNodeOffset, inst->blasTriOffset, inst->globalTriOffset,
                                  &tLocal, &lhp, &ln, &tm)) {
                    v3 worldHit = xform_point(inst->transform, lhp);
                    float tWorld = length3(sub3(worldHit, o));
                    if (tWorld < tHit) {
                        tHit = tWorld;
                        *hpOut = worldHit;
                        *nOut = normalize3(xform_normal(inst->inverseTransform, ln));
                        *matOut = tm;
                        *instOut = instIdx;
                        hit = 1;
                    }
                }
            }
        } else {
            if (node->count == 0) continue;  /* empty-scene root written by the builders; FS would never terminate on it */
            if (sp + 2 > 64) continue;
            stack[sp++] = node->leftFirst;
            stack[sp++] = node->leftFirst + 1;
        }
    }
    rec_trace(c, rec_n0, rec_t0);
    *tHitOut = tHit;
    return hit;
}

/* FS:507-528 */
static int shadow_visibility(rctx* c, v3 origin, v3 dir, float maxDist, float* visibility) {
    *visibility = 1.0f;
    float traveled = 0.0f;
    const float EPS = 0.001f;
    for (int iter = 0; iter < 32 && *visibility > 0.05f; ++iter) {
        float tHit; v3 hp = V3(0, 0, 0), n = V3(0, 0, 0); int matIdx = -1, inst = -1;
        if (!traverse_tlas(c, origin, dir, &tHit, &hp, &n, &matIdx, &inst)) return 1;
        if (tHit < EPS) { origin = add3(origin, scale3(dir, EPS)); continue; }
        traveled += tHit;
        if (traveled >= maxDist) return 1;
        const rzo_material* m = &c->sc->materials[matIdx];
        c->cnt.materials++;
        if (m->transparency > 0.0f) {
            *visibility *= m->transparency;
            origin = add3(hp, scale3(dir, EPS));
            continue;
        } else {
            *visibility = 0.0f;
            return 0;
        }
    }
    return *visibility > 0.05f;
}

/* FS:533-535 */
static inline v3 fresnel_schlick(float cosTheta, v3 F0) {
    float p = rzo_pow5(1.0f - cosTheta);
    return V3(F0.x + (1.0f - F0.x) * p, F0.y + (1.0f - F0.y) * p, F0.z + (1.0f - F0.z) * p);
}
/* FS:537-539 */
static inline v3 reflect_ray(v3 i, v3 n) {
    float k = 2.0f * dot3(i, n);
    return sub3(i, scale3(n, k));
}
/* FS:558-567 */
static inline int refract_dir(v3 incident, v3 normal, float eta, v3* refr) {
    float cosi = rzo_clamp(dot3(neg3(incident), normal), -1.0f, 1.0f);
    float sint2 = rzo_max(0.0f, 1.0f - cosi * cosi);
    float eta2 = eta * eta;
    float k = 1.0f - eta2 * sint2;
    if (k < 0.0f) return 0;
    float w = eta * cosi - sqrtf(k);
    *refr = normalize3(add3(scale3(incident, eta), scale3(normal, w)));
    return 1;
}

/* FS:569-663 */
static v3 calculate_lighting(rctx* c, int numLights, v3 hitPoint, v3 normal, const rzo_material* material, v3 viewDir) {
    const rzo_scene* sc = c->sc;
    v3 albedo = ld3(material->albedo);
    if (material->transparency > 0.0f) {                                   /* FS:571-610 */
        float f0 = rzo_pow2((1.0f - material->ior) / (1.0f + material->ior));
        v3 F0 = V3(f0, f0, f0);
        v3 specAccum = V3(0.0f, 0.0f, 0.0f);
        for (int i = 0; i < numLights; ++i) {
            if (i >= (int)sc->n_lights) break;
            const rzo_light* light = &sc->lights[i];
            c->cnt.light_fetches++;
            v3 L; float attenuation; float visibility;
            if (light->posdir[3] == 1.0f) {
                v3 lv = sub3(ld3(light->posdir), hitPoint);
                float dist = rzo_max(length3(lv), 0.001f);
                L = divs3(lv, dist);
                attenuation = light->power / (dist * dist);
                if (!shadow_visibility(c, add3(hitPoint, scale3(L, 0.001f)), L, dist, &visibility)) continue;
            } else {
                L = normalize3(ld3(light->posdir));
                attenuation = light->power;
                if (!shadow_visibility(c, add3(hitPoint, scale3(L, 0.001f)), L, 1e30f, &visibility)) continue;
            }
            attenuation *= visibility;
            c->cnt.lit_lights++;
            float NdotL = rzo_max(dot3(normal, L), 0.0f);
            if (NdotL <= 0.0f) continue;
            v3 H = normalize3(add3(L, viewDir));
            float NdotH = rzo_max(dot3(normal, H), 0.0f);
            float cosTheta = rzo_max(dot3(H, viewDir), 0.0f);
            v3 F = fresnel_schlick(cosTheta, F0);
            float rough = rzo_max(material->roughness, 0.02f);
            float a = rough * rough;
            float a2 = a * a;
            float dDen = (NdotH * NdotH) * (a2 - 1.0f) + 1.0f;
            float D = a2 / ((3.14159f * dDen) * dDen + 1e-6f);
            float k = (rough + 1.0f) * (rough + 1.0f) / 8.0f;
            float NdotV = rzo_max(dot3(normal, viewDir), 0.0f);
            float Gv = NdotV / ((NdotV * (1.0f - k) + k) + 1e-6f);
            float Gl = NdotL / ((NdotL * (1.0f - k) + k) + 1e-6f);
            float denom = rzo_max((4.0f * NdotL) * NdotV, 1e-4f);
            /* spec = (F * D * Gv * Gl) / denom */
            v3 spec = divs3(scale3(scale3(scale3(F, D), Gv), Gl), denom);
            /* specAccum += spec * light.color * attenuation * NdotL */
            v3 t = scale3(scale3(mul3(spec, ld3(light->color)), attenuation), NdotL);
            specAccum = add3(specAccum, t);
        }
        return specAccum;
    }
    /* FS:611-662 */
    v3 F0 = V3(rzo_mix(0.04f, albedo.x, material->metallic),
               rzo_mix(0.04f, albedo.y, material->metallic),
               rzo_mix(0.04f, albedo.z, material->metallic));
    v3 finalColor = V3(0.05f * albedo.x, 0.05f * albedo.y, 0.05f * albedo.z);
    for (int i = 0; i < numLights; ++i) {
        if (i >= (int)sc->n_lights) break;
        const rzo_light* light = &sc->lights[i];
        c->cnt.light_fetches++;
        v3 lightDir;
        float attenuation = 1.0f;
        float visibility;
        if (light->posdir[3] == 1.0f) {
            v3 lightVec = sub3(ld3(light->posdir), hitPoint);
            float distance = rzo_max(length3(lightVec), 0.001f);
            lightDir = normalize3(lightVec);
            attenuation = light->power / (distance * distance);
            if (!shadow_visibility(c, add3(hitPoint, scale3(lightDir, 0.001f)), lightDir, distance, &visibility)) continue;
        } else {
            lightDir = normalize3(ld3(light->posdir));
            attenuation = light->power;
            if (!shadow_visibility(c, add3(hitPoint, scale3(lightDir, 0.001f)), lightDir, 1e30f, &visibility)) continue;
        }
        attenuation *= visibility;
        c->cnt.lit_lights++;
        v3 halfwayDir = normalize3(add3(lightDir, viewDir));
        float NdotL = rzo_max(dot3(normal, lightDir), 0.0f);
        float NdotV = rzo_max(dot3(normal, viewDir), 0.0f);
        v3 F = fresnel_schlick(rzo_max(dot3(halfwayDir, viewDir), 0.0f), F0);
        float alpha = material->roughness * material->roughness;
        float alpha2 = alpha * alpha;
        float ndh = dot3(normal, halfwayDir);
        float denom = ((ndh * ndh) * (alpha2 - 1.0f) + 1.0f);
        float D = alpha2 / ((3.14159f * denom) * denom);
        float k = (material->roughness + 1.0f) * (material->roughness + 1.0f) / 8.0f;
        float G = NdotV / (NdotV * (1.0f - k) + k);
        G *= NdotL / (NdotL * (1.0f - k) + k);
        float denomSpec = rzo_max((4.0f * NdotV) * NdotL, 0.0001f);
        /* specular = (F * D * G) / denomSpec */
        v3 specular = divs3(scale3(scale3(F, D), G), denomSpec);
        /* diffuse = (1.0 - F) * albedo * NdotL / 3.14159 */
        v3 oneMinusF = V3(1.0f - F.x, 1.0f - F.y, 1.0f - F.z);
        v3 diffuse = divs3(scale3(mul3(oneMinusF, albedo), NdotL), 3.14159f);
        /* finalColor += max(vec3(0), (diffuse + specular) * light.color * attenuation) */
        v3 t = scale3(mul3(add3(diffuse, specular), ld3(light->color)), attenuation);
        finalColor = add3(finalColor, V3(rzo_max(0.0f, t.x), rzo_max(0.0f, t.y), rzo_max(0.0f, t.z)));
    }
    return finalColor;
}

/* FS:204-212 */
static void calculate_ray(const rzo_frame* fr, v2 uv, v2 seed, v3* origin, v3* dir) {
    v2 s1 = {seed.x + 1.0f, seed.y + 1.0f};
    float jx = rz_rand(seed) * 0.00002f, jy = rz_rand(s1) * 0.00002f;
    uv.x += jx; uv.y += jy;
    float cx = uv.x * 2.0f - 1.0f, cy = uv.y * 2.0f - 1.0f, cz = -1.0f, cw = 1.0f;
    const float* ip = fr->inv_proj;
    /* ray_eye = invProj * ray_clip ; only .xy survive FS:209 */
    float ex = ((ip[0] * cx + ip[4] * cy) + ip[8] * cz) + ip[12] * cw;
    float ey = ((ip[1] * cx + ip[5] * cy) + ip[9] * cz) + ip[13] * cw;
    v3 eye = V3(ex, ey, -1.0f);
    v3 world = xform_dir(fr->inv_view, eye);       /* w = 0 */
    *origin = ld3(fr->cam_pos);
    *dir = normalize3(world);
}

/* FS:668-773 for one pixel; adds into acc[0..3], updates *ior */
static void shade_pixel(rctx* c, const rzo_frame* fr, int px, int py, float* acc, float* ior) {
    const rzo_scene* sc = c->sc;
    float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
    v2 uv = {fragx / (float)fr->width, fragy / (float)fr->height};
    v3 color = V3(acc[0], acc[1], acc[2]);
    int maxBounces = fr->bounce_budget > 0 ? fr->bounce_budget : 5;
    float currentIor = *ior;
    for (int samp = fr->sample_base; samp < fr->sample_base + fr->spp; ++samp) {
        c->cnt.samples++;
        rec_sample(px, py, samp - fr->sample_base);
        float sf = ((fragx + fragy) + (float)samp) + 1.0f;
        v2 seed = {uv.x * sf, uv.y * sf};
        v3 currentOrigin, currentDirection;
        calculate_ray(fr, uv, seed, &currentOrigin, &currentDirection);
        v3 throughput = V3(1.0f, 1.0f, 1.0f);
        for (int bounce = 0; bounce < maxBounces; ++bounce) {
            float fb2 = (float)(bounce * bounce), fb = (float)bounce;
            v2 tempseed = {(seed.x * fb2) * 12793.46f + fb * 1423.34f,
                           (seed.y * fb2) * 12793.46f + fb * 1423.34f};
            v3 hitPoint = V3(0, 0, 0), hitNormal = V3(0, 0, 0);
            int materialIndex = -1, instanceIdx = -1;
            float closestT = 1e30f;
            int found = traverse_tlas(c, currentOrigin, currentDirection, &closestT, &hitPoint, &hitNormal,
                                      &materialIndex, &instanceIdx);
            if (!found) {
                float t = 0.5f * (normalize3(currentDirection).y + 1.0f);
                v3 sky = V3(rzo_mix(0.15f, 0.5f, t), rzo_mix(0.25f, 0.7f, t), rzo_mix(0.45f, 1.0f, t));
                color = add3(color, mul3(throughput, sky));
                break;
            }
            const rzo_material* hm = &sc->materials[materialIndex];
            c->cnt.materials++;
            v3 viewDir = normalize3(sub3(ld3(fr->cam_pos), hitPoint));
            if (bounce == 0) {
                v3 l = calculate_lighting(c, fr->num_lights, hitPoint, hitNormal, hm, viewDir);
                color = add3(color, mul3(throughput, l));
            }
            v2 rs = {tempseed.x + (float)samp, tempseed.y + (float)bounce};
            float randVal = rz_rand(rs);
            c->cnt.scatters++;
            if (hm->transparency > 0.0f) {                                  /* FS:723-747 */
                int entering = dot3(neg3(currentDirection), hitNormal) > 0.0f;
                v3 N = entering ? hitNormal : neg3(hitNormal);
                float extIor = currentIor;
                float nextIor = entering ? hm->ior : 1.0f;
                float eta = extIor / nextIor;
                float cosi = rzo_clamp(dot3(neg3(currentDirection), N), 0.0f, 1.0f);
                float F0 = rzo_pow2((extIor - nextIor) / (extIor + nextIor));
                float fresnel = F0 + (1.0f - F0) * rzo_pow5(1.0f - cosi);
                v3 refr;
                int ok = refract_dir(currentDirection, N, eta, &refr);
                if (!ok) {
                    currentDirection = reflect_ray(currentDirection, N);
                    throughput = scale3(throughput, 0.98f);
                } else {
                    currentDirection = refr;
                    currentIor = nextIor;
                    float tr = hm->transparency;
                    v3 tint = V3(rzo_mix(1.0f, hm->albedo[0], tr), rzo_mix(1.0f, hm->albedo[1], tr),
                                 rzo_mix(1.0f, hm->albedo[2], tr));
                    float omf = 1.0f - fresnel;
                    v3 tw = scale3(scale3(tint, tr), omf);
                    throughput = mul3(throughput, V3(rzo_clamp(tw.x, 0.0f, 1.0f), rzo_clamp(tw.y, 0.0f, 1.0f),
                                                     rzo_clamp(tw.z, 0.0f, 1.0f)));
                }
            } else {                                                        /* FS:748-757 */
                if (randVal < hm->reflectivity) {
                    currentDirection = reflect_ray(currentDirection, hitNormal);
                    throughput = scale3(throughput, 0.95f);
                } else {
                    currentDirection = random_hemisphere_direction(hitNormal, tempseed);
                    c->cnt.diffuse_scatters++;
                    if (!(tempseed.x == 0.0f && tempseed.y == 0.0f)) c->cnt.hemi_draws++;   /* (at bounce 0 the seed is (+0, +0) for every sample: FS:696) */
                    throughput = mul3(throughput, scale3(ld3(hm->albedo), 0.4f));
                }
            }
            /* FS:759-761 */
            float pushDir = dot3(currentDirection, hitNormal) > 0.0f ? 1.0f : -1.0f;
            currentOrigin = add3(hitPoint, scale3(scale3(hitNormal, pushDir), 0.003f));
            /* FS:764-769 : the SAME rand value as randVal */
            if (bounce > 2) {
                float p = rzo_max(throughput.x, rzo_max(throughput.y, throughput.z));
                if (randVal > p) break;
                throughput = divs3(throughput, p);
            }
        }
    }
    acc[0] = color.x; acc[1] = color.y; acc[2] = color.z;
    acc[3] += (float)fr->spp;
    *ior = currentIor;
}

typedef struct {
    const rzo_scene* sc; const rzo_frame* fr; float* accum; float* ior_state;
    int x0, y0, x1, y1;
    volatile long long* next_chunk;
    rzo_counters cnt;
    long long pixels_done;
} job_t;

/* Work is handed out in chunks of RZO_CHUNK consecutive pixels of the crop rectangle (row-major), not in rows: a
 * short, wide crop (the benchmark's 8-row bands) must still keep every thread busy. */
#define RZO_CHUNK 16

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    rctx c; c.sc = j->sc; memset(&c.cnt, 0, sizeof c.cnt);
    const long long w = j->x1 - j->x0, total = w * (long long)(j->y1 - j->y0);
    for (;;) {
        long long k0 = __sync_fetch_and_add(j->next_chunk, (long long)RZO_CHUNK);
        if (k0 >= total) break;
        long long k1 = k0 + RZO_CHUNK < total ? k0 + RZO_CHUNK : total;
        for (long long k = k0; k < k1; ++k) {
            const int y = j->y0 + (int)(k / w), x = j->x0 + (int)(k % w);
            size_t p = (size_t)y * j->fr->width + x;
            float* acc = j->accum + 4 * p;
            float ior = 1.0f;
            if (j->fr->sample_base == 0) { acc[0] = acc[1] = acc[2] = acc[3] = 0.0f; }
            else if (j->ior_state) ior = j->ior_state[p];
            shade_pixel(&c, j->fr, x, y, acc, &ior);
            if (j->ior_state) j->ior_state[p] = ior;
            c.cnt.pixels++;
            j->pixels_done++;
        }
    }
    j->cnt = c.cnt;
    return NULL;
}

static int g_threads_busy = 0;
int rzo_last_threads_busy(void) { return g_threads_busy; }

int rzo_render(const rzo_scene* scene, const rzo_frame* frame, float* accum, float* ior_state,
               int x0, int y0, int x1, int y1, int nthreads, rzo_counters* counters) {
    if (!scene || !frame || !accum) return -1;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > frame->width) x1 = frame->width;
    if (y1 > frame->height) y1 = frame->height;
    if (nthreads <= 0) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    volatile long long next_chunk = 0;
    if (x1 <= x0 || y1 <= y0) { if (counters) memset(counters, 0, sizeof *counters); return 0; }
    job_t* jobs = (job_t*)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; ++i) {
        jobs[i].sc = scene; jobs[i].fr = frame; jobs[i].accum = accum; jobs[i].ior_state = ior_state;
        jobs[i].x0 = x0; jobs[i].y0 = y0; jobs[i].x1 = x1; jobs[i].y1 = y1; jobs[i].next_chunk = &next_chunk;
    }
    if (nthreads == 1) worker(&jobs[0]);
    else {
        for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &jobs[i]);
        for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    }
    g_threads_busy = 0;
    for (int i = 0; i < nthreads; ++i) g_threads_busy += jobs[i].pixels_done > 0;
    if (counters) {
        memset(counters, 0, sizeof *counters);
        for (int i = 0; i < nthreads; ++i) {
            const uint64_t* s = (const uint64_t*)&jobs[i].cnt; uint64_t* d = (uint64_t*)counters;
            for (size_t k = 0; k < sizeof(rzo_counters) / sizeof(uint64_t); ++k) d[k] += s[k];
        }
    }
    free(jobs); free(th);
    return 0;
}

int rzo_trace(const rzo_scene* scene, const float origin[3], const float dir[3], float out[10]) {
    rctx c; c.sc = scene; memset(&c.cnt, 0, sizeof c.cnt);
    float t; v3 hp = V3(0, 0, 0), n = V3(0, 0, 0); int mat = -1, inst = -1;
    int hit = traverse_tlas(&c, ld3(origin), ld3(dir), &t, &hp, &n, &mat, &inst);
    out[0] = (float)hit; out[1] = t; out[2] = hp.x; out[3] = hp.y; out[4] = hp.z;
    out[5] = n.x; out[6] = n.y; out[7] = n.z; out[8] = (float)mat; out[9] = (float)inst;
    return hit;
}

int rzo_shadow(const rzo_scene* scene, const float origin[3], const float dir[3], float maxDist, float* visibility) {
    rctx c; c.sc = scene; memset(&c.cnt, 0, sizeof c.cnt);
    return shadow_visibility(&c, ld3(origin), ld3(dir), maxDist, visibility);
}

int rzo_math_flavour = 1;      /* rz_oracle_math.h: 1 = llvmpipe's sin / cos / acos (what the product is compiled with by default), 0 = rounds 1-4's binary64 ones; tests/helpers.py sets it to the loaded library's */
void rzo_set_math_flavour(int f) { rzo_math_flavour = f; }
int rzo_get_math_flavour(void) { return rzo_math_flavour; }
float rzo_sin_f(float x) { return rzo_sin(x); }
float rzo_cos_f(float x) { return rzo_cos(x); }
float rzo_acos_f(float x) { return rzo_acos(x); }
float rzo_rand_f(float x, float y) { v2 s = {x, y}; return rz_rand(s); }
void rzo_hemisphere_f(const float n[3], const float seed[2], float out[3]) {
    v2 s = {seed[0], seed[1]};
    v3 r = random_hemisphere_direction(ld3(n), s);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
