/*
 * rz_oracle_present.c -- CPU restatement of the presentation tail of RayZen's fragment shader: everything main()
 * does to `color` after the path loop (RayZen/shaders/fragment_shader.glsl, "FS"):
 *   FS:772-773  color /= numSamples; clamp(0,1)
 *   FS:775-779  BVH wireframe overlay          (overlayBVHWireframe FS:310-373, aabbWireframe FS:229-254,
 *                                               distanceToSegment FS:222-226, hsv2rgb FS:215-219,
 *                                               findBVHBranchIterative FS:257-307)
 *   FS:781-803  light markers
 *   FS:805-819  FPS overlay                    (font FS:118-146, drawFontChar FS:152-161, drawFpsString FS:164-183)
 * TEST INFRASTRUCTURE ONLY; pinned against RayZen's own shader run on Mesa llvmpipe (overlay frames of
 * tests/golden/glref_rayzen_main.npz; see rz_oracle.h).  Same pinned numerics as rz_oracle.c: IEEE binary32,
 * one rounding per operation, no FMA, GLSL operand order; mat*mat and mat*vec sum their terms left to right;
 * smoothstep(e0,e1,x) = t*t*(3 - 2*t), t = clamp((x-e0)/(e1-e0),0,1); float->int conversions truncate.
 */
#include "rz_oracle.h"
#include "rz_oracle_math.h"

#include <math.h>
#include <string.h>

typedef struct { float x, y; } p2;
typedef struct { float x, y, z; } p3;
typedef struct { float x, y, z, w; } p4;

/* FS:123-146 */
static const int kFont[11][8] = {
    {0x3C, 0x66, 0x6E, 0x7E, 0x76, 0x66, 0x3C, 0x00}, {0x18, 0x38, 0x18, 0x18, 0x18, 0x18, 0x3C, 0x00},
    {0x3C, 0x66, 0x06, 0x1C, 0x30, 0x66, 0x7E, 0x00}, {0x3C, 0x66, 0x06, 0x1C, 0x06, 0x66, 0x3C, 0x00},
    {0x0C, 0x1C, 0x3C, 0x6C, 0x7E, 0x0C, 0x0C, 0x00}, {0x7E, 0x60, 0x7C, 0x06, 0x06, 0x66, 0x3C, 0x00},
    {0x1C, 0x30, 0x60, 0x7C, 0x66, 0x66, 0x3C, 0x00}, {0x7E, 0x66, 0x0C, 0x18, 0x18, 0x18, 0x18, 0x00},
    {0x3C, 0x66, 0x66, 0x3C, 0x66, 0x66, 0x3C, 0x00}, {0x3C, 0x66, 0x66, 0x3E, 0x06, 0x0C, 0x38, 0x00},
    {0x00, 0x00, 0x00, 0x00, 0x00, 0x18, 0x18, 0x00}};

static p4 mat_vec(const float* m, p4 v) {          /* column-major mat4 * vec4 */
    p4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
static void mat_mul(const float* a, const float* b, float* out) {   /* out = a * b */
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            out[c * 4 + r] = ((a[0 * 4 + r] * b[c * 4 + 0] + a[1 * 4 + r] * b[c * 4 + 1]) + a[2 * 4 + r] * b[c * 4 + 2]) +
                             a[3 * 4 + r] * b[c * 4 + 3];
}
static p3 mix3(p3 a, p3 b, float t) {
    p3 r = {rzo_mix(a.x, b.x, t), rzo_mix(a.y, b.y, t), rzo_mix(a.z, b.z, t)};
    return r;
}

/* FS:215-219 */
static p3 hsv2rgb(p3 c) {
    const float Kx = 1.0f, Ky = 2.0f / 3.0f, Kz = 1.0f / 3.0f, Kw = 3.0f;
    float px = fabsf(rzo_fract(c.x + Kx) * 6.0f - Kw), py = fabsf(rzo_fract(c.x + Ky) * 6.0f - Kw),
          pz = fabsf(rzo_fract(c.x + Kz) * 6.0f - Kw);
    p3 r;
    r.x = c.z * rzo_mix(Kx, rzo_clamp(px - Kx, 0.0f, 1.0f), c.y);
    r.y = c.z * rzo_mix(Kx, rzo_clamp(py - Kx, 0.0f, 1.0f), c.y);
    r.z = c.z * rzo_mix(Kx, rzo_clamp(pz - Kx, 0.0f, 1.0f), c.y);
    return r;
}

/* FS:222-226 */
static float distance_to_segment(p2 p, p2 a, p2 b) {
    p2 ab = {b.x - a.x, b.y - a.y};
    p2 pa = {p.x - a.x, p.y - a.y};
    float t = rzo_clamp((pa.x * ab.x + pa.y * ab.y) / (ab.x * ab.x + ab.y * ab.y), 0.0f, 1.0f);
    p2 d = {p.x - (a.x + t * ab.x), p.y - (a.y + t * ab.y)};
    return sqrtf(d.x * d.x + d.y * d.y);
}

/* FS:229-254 */
static float aabb_wireframe(const float* bmin, const float* bmax, const float* viewProj, p2 frag, float thickness,
                            float resx, float resy) {
    p3 corners[8] = {{bmin[0], bmin[1], bmin[2]}, {bmax[0], bmin[1], bmin[2]}, {bmax[0], bmax[1], bmin[2]},
                     {bmin[0], bmax[1], bmin[2]}, {bmin[0], bmin[1], bmax[2]}, {bmax[0], bmin[1], bmax[2]},
                     {bmax[0], bmax[1], bmax[2]}, {bmin[0], bmax[1], bmax[2]}};
    static const int edges[24] = {0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 0, 4, 1, 5, 2, 6, 3, 7};
    float minDist = 1e6f;
    for (int i = 0; i < 12; ++i) {
        p3 c0 = corners[edges[i * 2]], c1 = corners[edges[i * 2 + 1]];
        p4 v0 = {c0.x, c0.y, c0.z, 1.0f}, v1 = {c1.x, c1.y, c1.z, 1.0f};
        p4 q0 = mat_vec(viewProj, v0), q1 = mat_vec(viewProj, v1);
        if (q0.w <= 0.0f || q1.w <= 0.0f) continue;
        p2 s0 = {(q0.x / q0.w * 0.5f + 0.5f) * resx, (q0.y / q0.w * 0.5f + 0.5f) * resy};
        p2 s1 = {(q1.x / q1.w * 0.5f + 0.5f) * resx, (q1.y / q1.w * 0.5f + 0.5f) * resy};
        minDist = rzo_min(minDist, distance_to_segment(frag, s0, s1));
    }
    return minDist < thickness ? 1.0f : 0.0f;
}

/* FS:257-307.  Pixel-independent; returns the path length. */
static int find_bvh_branch(const rzo_scene* sc, int nodeOffset, int triOffset, int nodeCount, int selectedTri, int path[32]) {
    int cur = 0, pathLen = 0;
    for (int depth = 0; depth < 32; ++depth) {
        path[pathLen++] = cur;
        if (nodeOffset + cur < 0 || (size_t)(nodeOffset + cur) >= sc->n_blas_nodes) { pathLen = 0; break; }
        const rzo_node* node = &sc->blas_nodes[nodeOffset + cur];
        if (node->count > 0) {
            int found = 0;
            for (int i = 0; i < node->count; ++i)
                if (sc->blas_indices[triOffset + node->leftFirst + i] == selectedTri) { found = 1; break; }
            if (!found) pathLen = 0;
            break;
        } else {
            int leftIdx = node->leftFirst, rightIdx = node->leftFirst + 1, foundInLeft = 0;
            int stack[32], sp = 0;
            stack[sp++] = leftIdx;
            while (sp > 0) {
                int nidx = stack[--sp];
                if (nidx < 0 || nidx >= nodeCount) continue;
                const rzo_node* n = &sc->blas_nodes[nodeOffset + nidx];
                if (n->count > 0) {
                    for (int j = 0; j < n->count; ++j)
                        if (sc->blas_indices[triOffset + n->leftFirst + j] == selectedTri) { foundInLeft = 1; break; }
                } else if (sp + 2 <= 32) {
                    stack[sp++] = n->leftFirst;
                    stack[sp++] = n->leftFirst + 1;
                }
                if (foundInLeft) break;
            }
            cur = foundInLeft ? leftIdx : rightIdx;
        }
    }
    return pathLen;
}

/* FS:152-161 */
static float draw_font_char(int charIdx, p2 frag, p2 pos, float scale) {
    float rx = (frag.x - pos.x) / scale, ry = (frag.y - pos.y) / scale;
    int x = (int)rx;
    int y = 8 - 1 - (int)ry;
    if (x < 0 || x >= 8 || y < 0 || y >= 8) return 0.0f;
    int row = kFont[charIdx][y];
    int mask = 1 << (8 - 1 - x);
    return (row & mask) != 0 ? 1.0f : 0.0f;
}

/* FS:164-183 */
static p3 draw_fps_string(p2 frag, p2 pos, float scale, float fps, p3 fg, p3 bg) {
    int fpsInt = (int)fps;
    int tenths = (int)(rzo_fract(fps) * 10.0f);
    int chars[5] = {(fpsInt / 100) % 10, (fpsInt / 10) % 10, fpsInt % 10, 10, tenths};
    p3 col = bg;
    for (int i = 0; i < 5; ++i) {
        int ci = chars[i];
        if (ci < 0) ci = 0;
        if (ci > 10) ci = 10;              /* FS indexes fontData unchecked: out of range is undefined there */
        p2 cp = {pos.x + (float)(i * (8 + 1)) * scale, pos.y + 0.0f};
        float glyph = draw_font_char(ci, frag, cp, scale);
        col = mix3(col, fg, glyph);
    }
    return col;
}

int rzo_present(const rzo_scene* sc, const rzo_present_params* pp, const float* accum, float* rgb_out, unsigned char* rgba8_out) {
    const int W = pp->width, H = pp->height;
    const float resx = (float)W, resy = (float)H;
    float viewProj[16];
    mat_mul(pp->proj, pp->view, viewProj);         /* camera.projectionMatrix * camera.viewMatrix */
    int path[32], pathLen = 0, selNodeOffset = 0;
    const rzo_instance* selInst = NULL;
    if (pp->show_bvh && pp->bvh_mode == 1 && pp->selected_blas >= 0 && (size_t)pp->selected_blas < sc->n_instances) {
        selInst = &sc->instances[pp->selected_blas];
        selNodeOffset = selInst->blasNodeOffset;
        int nodeCount = ((size_t)pp->selected_blas + 1 < sc->n_instances)
                            ? sc->instances[pp->selected_blas + 1].blasNodeOffset - selNodeOffset
                            : (int)sc->n_blas_nodes - selNodeOffset;
        pathLen = find_bvh_branch(sc, selNodeOffset, selInst->blasTriOffset, nodeCount, pp->selected_tri, path);
    }
    for (int py = 0; py < H; ++py)
        for (int px = 0; px < W; ++px) {
            const size_t pix = (size_t)py * W + px;
            const p2 frag = {(float)px + 0.5f, (float)py + 0.5f};
            const float n = accum[4 * pix + 3] > 0.0f ? accum[4 * pix + 3] : 1.0f;
            p3 color = {rzo_clamp(accum[4 * pix] / n, 0.0f, 1.0f), rzo_clamp(accum[4 * pix + 1] / n, 0.0f, 1.0f),
                        rzo_clamp(accum[4 * pix + 2] / n, 0.0f, 1.0f)};                       /* FS:772-773 */
            if (pp->show_bvh) {                                                                /* FS:683-685, 775-779 */
                float tlasWire = 0.0f, blasWire = 0.0f;
                p3 tlasColor = {0, 0, 0}, blasColor = {0, 0, 0};
                if (pp->bvh_mode == 0) {
                    for (size_t i = 0; i < sc->n_tlas_nodes; ++i) {
                        const rzo_node* node = &sc->tlas_nodes[i];
                        if (node->count > 0) {
                            float w = aabb_wireframe(node->bmin, node->bmax, viewProj, frag, 1.5f, resx, resy);
                            if (w > 0.0f) {
                                int meshIdx = sc->tlas_indices[node->leftFirst];
                                if (meshIdx < 0 || (size_t)meshIdx >= sc->n_instances) continue;
                                float t = (float)meshIdx / (float)sc->n_instances;
                                p3 hsv = {0.0f + t * 0.5f, 1.0f, 1.0f};
                                tlasColor = mix3(tlasColor, hsv2rgb(hsv), w);
                                tlasWire = rzo_max(tlasWire, w);
                            }
                        }
                    }
                    for (size_t i = 0; i < sc->n_instances; ++i) {
                        const rzo_node* node = &sc->blas_nodes[sc->instances[i].blasNodeOffset];
                        float w = aabb_wireframe(node->bmin, node->bmax, viewProj, frag, 2.0f, resx, resy);
                        if (w > 0.0f) {
                            p3 black = {0, 0, 0};
                            blasColor = mix3(blasColor, black, w);
                            blasWire = rzo_max(blasWire, w);
                        }
                    }
                } else if (pp->bvh_mode == 1 && selInst) {
                    for (int i = 0; i < pathLen; ++i) {
                        const rzo_node* node = &sc->blas_nodes[selNodeOffset + path[i]];
                        p4 a = {node->bmin[0], node->bmin[1], node->bmin[2], 1.0f}, b = {node->bmax[0], node->bmax[1], node->bmax[2], 1.0f};
                        p4 ta = mat_vec(selInst->transform, a), tb = mat_vec(selInst->transform, b);
                        float bmin[3] = {ta.x, ta.y, ta.z}, bmax[3] = {tb.x, tb.y, tb.z};
                        float wire = aabb_wireframe(bmin, bmax, viewProj, frag, 2.0f, resx, resy);
                        blasWire = rzo_max(blasWire, wire);
                        p3 hsv = {(float)i / (float)pathLen, 1.0f, 1.0f};
                        blasColor = mix3(blasColor, hsv2rgb(hsv), wire);
                    }
                }
                if (tlasWire > 0.0f || blasWire > 0.0f) {
                    color = mix3(color, tlasColor, 0.5f * tlasWire);
                    color = mix3(color, blasColor, 0.5f * blasWire);
                }
            }
            if (pp->show_lights) {                                                             /* FS:782-803 */
                for (int i = 0; i < pp->num_lights; ++i) {
                    if ((size_t)i >= sc->n_lights) break;
                    const rzo_light* L = &sc->lights[i];
                    if (L->posdir[3] == 1.0f) {
                        p4 wp = {L->posdir[0], L->posdir[1], L->posdir[2], 1.0f};
                        p4 clip = mat_vec(viewProj, wp);
                        if (clip.w > 0.0f) {
                            float nx = clip.x / clip.w, ny = clip.y / clip.w;
                            p2 screen = {(nx * 0.5f + 0.5f) * resx, (ny * 0.5f + 0.5f) * resy};
                            float dx = frag.x - screen.x, dy = frag.y - screen.y;
                            float dist = sqrtf(dx * dx + dy * dy);
                            if (dist < 8.0f) {
                                float t = rzo_clamp((dist - 8.0f) / ((8.0f - 2.0f) - 8.0f), 0.0f, 1.0f);
                                float alpha = (t * t) * (3.0f - 2.0f * t);
                                p3 lc = {L->color[0], L->color[1], L->color[2]};
                                color = mix3(color, lc, alpha);
                            }
                        }
                    }
                }
            }
            if (pp->show_fps) {                                                                /* FS:806-819 */
                const float margin = 8.0f, fpsScale = 2.0f;
                p2 fpsPos = {margin, (resy - margin) - 8.0f * 2.0f};
                p3 white = {1.0f, 1.0f, 1.0f};
                p3 fpsCol = draw_fps_string(frag, fpsPos, fpsScale, pp->fps, white, color);
                float anyFps = 0.0f;
                for (int y = 0; y < 8; ++y)
                    for (int x = 0; x < 8 * 6; ++x) {
                        p2 p = {fpsPos.x + (float)x * fpsScale, fpsPos.y + (float)y * fpsScale};
                        if (fabsf(frag.x - p.x) < 1.0f && fabsf(frag.y - p.y) < 1.0f) anyFps = 1.0f;
                    }
                color = mix3(color, fpsCol, anyFps);
            }
            if (rgb_out) { rgb_out[3 * pix] = color.x; rgb_out[3 * pix + 1] = color.y; rgb_out[3 * pix + 2] = color.z; }
            if (rgba8_out) {
                rgba8_out[4 * pix] = (unsigned char)rintf(rzo_clamp(color.x, 0.0f, 1.0f) * 255.0f);
                rgba8_out[4 * pix + 1] = (unsigned char)rintf(rzo_clamp(color.y, 0.0f, 1.0f) * 255.0f);
                rgba8_out[4 * pix + 2] = (unsigned char)rintf(rzo_clamp(color.z, 0.0f, 1.0f) * 255.0f);
                rgba8_out[4 * pix + 3] = 255;
            }
        }
    return 0;
}
