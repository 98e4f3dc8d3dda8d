// rz_present.hip -- the presentation tail of RayZen's fragment shader as one per-pixel kernel:
//   FS:772-773 resolve (divide by the sample count, clamp), FS:775-779 BVH wireframe (FS:214-373),
//   FS:781-803 light markers, FS:805-819 FPS digits (font FS:118-183), then 8-bit quantisation.
// HBM-bound by construction (16 B read + 4 B written per pixel); the overlays are ALU on data every pixel
// shares (TLAS nodes, instance root boxes, lights, the branch path), which therefore arrives through scalar loads.
// Written independently of the oracle; same pinned numerics as the render kernels (rz_device_math.h).
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_device_math.h"
#include "rz_internal.h"

namespace rz {

// The 8 projected corners of one box (FS:243-249 for both endpoints of every edge) do not depend on the pixel:
// rz_project_boxes computes them once per frame with the very same operations, plus a conservative screen-space
// bounding rectangle (valid corners only, grown by the line thickness and a 1-px guard against rounding) that lets a
// pixel skip a box none of whose edges can come within `thickness` of it.
struct ProjBox { float sx[8], sy[8], sw[8]; float lox, loy, hix, hiy; };

__constant__ int kFontRows[11][8] = {
    {0x3C, 0x66, 0x6E, 0x7E, 0x76, 0x66, 0x3C, 0x00}, {0x18, 0x38, 0x18, 0x18, 0x18, 0x18, 0x3C, 0x00},
    {0x3C, 0x66, 0x06, 0x1C, 0x30, 0x66, 0x7E, 0x00}, {0x3C, 0x66, 0x06, 0x1C, 0x06, 0x66, 0x3C, 0x00},
    {0x0C, 0x1C, 0x3C, 0x6C, 0x7E, 0x0C, 0x0C, 0x00}, {0x7E, 0x60, 0x7C, 0x06, 0x06, 0x66, 0x3C, 0x00},
    {0x1C, 0x30, 0x60, 0x7C, 0x66, 0x66, 0x3C, 0x00}, {0x7E, 0x66, 0x0C, 0x18, 0x18, 0x18, 0x18, 0x00},
    {0x3C, 0x66, 0x66, 0x3C, 0x66, 0x66, 0x3C, 0x00}, {0x3C, 0x66, 0x66, 0x3E, 0x06, 0x0C, 0x38, 0x00},
    {0x00, 0x00, 0x00, 0x00, 0x00, 0x18, 0x18, 0x00}};

__device__ __forceinline__ float4 m4v(const float* m, float x, float y, float z, float w) {
    return make_float4(((m[0] * x + m[4] * y) + m[8] * z) + m[12] * w, ((m[1] * x + m[5] * y) + m[9] * z) + m[13] * w,
                       ((m[2] * x + m[6] * y) + m[10] * z) + m[14] * w, ((m[3] * x + m[7] * y) + m[11] * z) + m[15] * w);
}
__device__ __forceinline__ v3 mixv(v3 a, v3 b, float t) { return mk3(mix_(a.x, b.x, t), mix_(a.y, b.y, t), mix_(a.z, b.z, t)); }

// FS:215-219
__device__ __forceinline__ v3 hsv_to_rgb(float h, float s, float v) {
    const float k1 = 2.0f / 3.0f, k2 = 1.0f / 3.0f;
    const float px = __builtin_fabsf(fract_(h + 1.0f) * 6.0f - 3.0f), py = __builtin_fabsf(fract_(h + k1) * 6.0f - 3.0f),
                pz = __builtin_fabsf(fract_(h + k2) * 6.0f - 3.0f);
    return mk3(v * mix_(1.0f, clamp_(px - 1.0f, 0.0f, 1.0f), s), v * mix_(1.0f, clamp_(py - 1.0f, 0.0f, 1.0f), s),
               v * mix_(1.0f, clamp_(pz - 1.0f, 0.0f, 1.0f), s));
}
// FS:222-226
__device__ __forceinline__ float seg_dist(float px, float py, float ax, float ay, float bx, float by) {
    const float abx = bx - ax, aby = by - ay, pax = px - ax, pay = py - ay;
    const float t = clamp_((pax * abx + pay * aby) / (abx * abx + aby * aby), 0.0f, 1.0f);
    const float dx = px - (ax + t * abx), dy = py - (ay + t * aby);
    return __builtin_sqrtf(dx * dx + dy * dy);
}
// FS:229-254, first half: project the corners
__device__ void project_box(const float* mn, const float* mx, const float* vp, float thickness, float resx, float resy,
                            ProjBox& B) {
    float lox = 3.0e38f, loy = 3.0e38f, hix = -3.0e38f, hiy = -3.0e38f;
    for (int c = 0; c < 8; ++c) {
        // corner order of FS:232-239
        const float x = (c == 1 || c == 2 || c == 5 || c == 6) ? mx[0] : mn[0];
        const float y = (c == 2 || c == 3 || c == 6 || c == 7) ? mx[1] : mn[1];
        const float z = (c >= 4) ? mx[2] : mn[2];
        const float4 q = m4v(vp, x, y, z, 1.0f);
        B.sw[c] = q.w;
        B.sx[c] = (q.x / q.w * 0.5f + 0.5f) * resx;
        B.sy[c] = (q.y / q.w * 0.5f + 0.5f) * resy;
        if (q.w > 0.0f) {
            lox = fmin_(lox, B.sx[c]); hix = fmax_(hix, B.sx[c]);
            loy = fmin_(loy, B.sy[c]); hiy = fmax_(hiy, B.sy[c]);
        }
    }
    // NaN/inf corners: keep the box (rectangle = everything)
    const bool finite = (lox - lox == 0.0f) && (hix - hix == 0.0f) && (loy - loy == 0.0f) && (hiy - hiy == 0.0f);
    const float g = thickness + 1.0f;
    B.lox = finite ? lox - g : -3.0e38f; B.hix = finite ? hix + g : 3.0e38f;
    B.loy = finite ? loy - g : -3.0e38f; B.hiy = finite ? hiy + g : 3.0e38f;
}
// FS:229-254, second half: distance of this pixel to the 12 edges
__device__ __forceinline__ float box_wire(const ProjBox& B, float fx, float fy, float thickness) {
    if (fx < B.lox || fx > B.hix || fy < B.loy || fy > B.hiy) return 0.0f;
    const int e0[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3}, e1[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
    float best = 1e6f;
    for (int i = 0; i < 12; ++i) {
        const int a = e0[i], b = e1[i];
        if (B.sw[a] <= 0.0f || B.sw[b] <= 0.0f) continue;
        best = fmin_(best, seg_dist(fx, fy, B.sx[a], B.sy[a], B.sx[b], B.sy[b]));
    }
    return best < thickness ? 1.0f : 0.0f;
}

// boxes[0 .. nTlasNodes) = TLAS nodes (leaves only are used), then nInstances BLAS roots, then pathLen path boxes
__global__ void rz_project_boxes(const PresentParams P) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const float resx = (float)P.width, resy = (float)P.height;
    const int nT = P.bvhMode == 0 ? P.nTlasNodes : 0, nI = P.bvhMode == 0 ? P.nInstances : 0;
    if (k < nT) {
        const TlasNode nd = P.tlasNodes[k];
        project_box(nd.bmin, nd.bmax, P.viewProj, 1.5f, resx, resy, P.boxes[k]);
    } else if (k < nT + nI) {
        const DevInstance& I = P.instances[k - nT];
        project_box(I.rootMin, I.rootMax, P.viewProj, 2.0f, resx, resy, P.boxes[k]);
    } else if (k < nT + nI + P.pathLen) {
        const int j = k - nT - nI;
        const float4 ta = m4v(P.selTransform, P.pathMin[j][0], P.pathMin[j][1], P.pathMin[j][2], 1.0f);
        const float4 tb = m4v(P.selTransform, P.pathMax[j][0], P.pathMax[j][1], P.pathMax[j][2], 1.0f);
        const float mn[3] = {ta.x, ta.y, ta.z}, mx[3] = {tb.x, tb.y, tb.z};
        project_box(mn, mx, P.viewProj, 2.0f, resx, resy, P.boxes[k]);
    }
}
// FS:152-161
__device__ __forceinline__ float glyph_at(int ch, float fx, float fy, float posx, float posy, float scale) {
    const int x = (int)((fx - posx) / scale);
    const int y = 8 - 1 - (int)((fy - posy) / scale);
    if (x < 0 || x >= 8 || y < 0 || y >= 8) return 0.0f;
    return (kFontRows[ch][y] & (1 << (8 - 1 - x))) != 0 ? 1.0f : 0.0f;
}

__global__ __launch_bounds__(256) void rz_present_kernel(const PresentParams P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.width * P.height) return;
    const int px = i % P.width, py = i / P.width;
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const float resx = (float)P.width, resy = (float)P.height;
    const float4 a = P.accum[i];
    const float n = a.w > 0.0f ? a.w : 1.0f;
    v3 color = mk3(clamp_(a.x / n, 0.0f, 1.0f), clamp_(a.y / n, 0.0f, 1.0f), clamp_(a.z / n, 0.0f, 1.0f));
    if (P.showBvh) {
        float tlasWire = 0.0f, blasWire = 0.0f;
        v3 tlasColor = mk3(0, 0, 0), blasColor = mk3(0, 0, 0);
        if (P.bvhMode == 0) {
            for (int k = 0; k < P.nTlasNodes; ++k) {
                const TlasNode nd = P.tlasNodes[k];
                if (nd.count > 0) {
                    const float w = box_wire(P.boxes[k], fx, fy, 1.5f);
                    if (w > 0.0f) {
                        const int meshIdx = P.tlasIndices[nd.leftFirst];
                        if (meshIdx < 0 || meshIdx >= P.nInstances) continue;
                        const float t = (float)meshIdx / (float)P.nInstances;
                        tlasColor = mixv(tlasColor, hsv_to_rgb(0.0f + t * 0.5f, 1.0f, 1.0f), w);
                        tlasWire = fmax_(tlasWire, w);
                    }
                }
            }
            for (int k = 0; k < P.nInstances; ++k) {
                const float w = box_wire(P.boxes[P.nTlasNodes + k], fx, fy, 2.0f);
                if (w > 0.0f) {
                    blasColor = mixv(blasColor, mk3(0, 0, 0), w);
                    blasWire = fmax_(blasWire, w);
                }
            }
        } else if (P.bvhMode == 1) {
            for (int k = 0; k < P.pathLen; ++k) {
                const float w = box_wire(P.boxes[k], fx, fy, 2.0f);
                blasWire = fmax_(blasWire, w);
                blasColor = mixv(blasColor, hsv_to_rgb((float)k / (float)P.pathLen, 1.0f, 1.0f), w);
            }
        }
        if (tlasWire > 0.0f || blasWire > 0.0f) {
            color = mixv(color, tlasColor, 0.5f * tlasWire);
            color = mixv(color, blasColor, 0.5f * blasWire);
        }
    }
    if (P.showLights) {
        for (int k = 0; k < P.nLights; ++k) {
            const DevLight L = P.lights[k];
            if (L.posdir[3] == 1.0f) {
                const float4 clip = m4v(P.viewProj, L.posdir[0], L.posdir[1], L.posdir[2], 1.0f);
                if (clip.w > 0.0f) {
                    const float sx = ((clip.x / clip.w) * 0.5f + 0.5f) * resx, sy = ((clip.y / clip.w) * 0.5f + 0.5f) * resy;
                    const float dx = fx - sx, dy = fy - sy;
                    const float dist = __builtin_sqrtf(dx * dx + dy * dy);
                    if (dist < 8.0f) {
                        const float t = clamp_((dist - 8.0f) / (6.0f - 8.0f), 0.0f, 1.0f);     // smoothstep(8, 6, dist)
                        color = mixv(color, mk3(L.color[0], L.color[1], L.color[2]), (t * t) * (3.0f - 2.0f * t));
                    }
                }
            }
        }
    }
    if (P.showFps) {
        const float posx = 8.0f, posy = (resy - 8.0f) - 16.0f, scale = 2.0f;
        const int fpsInt = (int)P.fps;
        const int tenths = (int)(fract_(P.fps) * 10.0f);
        int chars[5] = {(fpsInt / 100) % 10, (fpsInt / 10) % 10, fpsInt % 10, 10, tenths};
        v3 col = color;
        for (int k = 0; k < 5; ++k) {
            const int ch = chars[k] < 0 ? 0 : (chars[k] > 10 ? 10 : chars[k]);
            col = mixv(col, mk3(1.0f, 1.0f, 1.0f), glyph_at(ch, fx, fy, posx + (float)(k * 9) * scale, posy + 0.0f, scale));
        }
        // FS:810-818: the 48 x 8 grid of 2-px cells the digits live in
        // (cells sit at posx + 2x, posy + 2y: outside their hull grown by 1 px no cell can match)
        float any = 0.0f;
        if (fx > posx - 2.0f && fx < posx + 96.0f && fy > posy - 2.0f && fy < posy + 16.0f)
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 48; ++x)
                    if (__builtin_fabsf(fx - (posx + (float)x * scale)) < 1.0f && __builtin_fabsf(fy - (posy + (float)y * scale)) < 1.0f) any = 1.0f;
        color = mixv(color, col, any);
    }
    if (P.rgb) { P.rgb[3 * (size_t)i] = color.x; P.rgb[3 * (size_t)i + 1] = color.y; P.rgb[3 * (size_t)i + 2] = color.z; }
    if (P.rgba8)
        P.rgba8[i] = make_uchar4((unsigned char)__builtin_rintf(clamp_(color.x, 0.0f, 1.0f) * 255.0f),
                                 (unsigned char)__builtin_rintf(clamp_(color.y, 0.0f, 1.0f) * 255.0f),
                                 (unsigned char)__builtin_rintf(clamp_(color.z, 0.0f, 1.0f) * 255.0f), 255);
}

void launch_present(const PresentParams& P, hipStream_t s) {
    const int nBoxes = P.showBvh ? (P.bvhMode == 0 ? P.nTlasNodes + P.nInstances : P.pathLen) : 0;
    if (nBoxes > 0) hipLaunchKernelGGL(rz_project_boxes, dim3((nBoxes + 63) / 64), dim3(64), 0, s, P);
    const int n = P.width * P.height;
    hipLaunchKernelGGL(rz_present_kernel, dim3((n + 255) / 256), dim3(256), 0, s, P);
}

}  // namespace rz
