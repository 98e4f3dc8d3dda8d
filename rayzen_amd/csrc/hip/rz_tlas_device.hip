// rz_tlas_device.hip -- updateDynamicBVHAndSSBOs (RayZen/src/main.cpp:1138-1194) on the GPU.
//
// Per frame RayZen re-reads every object's transform, inverts it, recomputes the world AABB of each BLAS root
// (main.cpp:1168-1191), rebuilds the TLAS on the CPU (BVH.cpp:178-240) and re-uploads everything.  Here the frontend
// hands over only the transforms (64 B per instance); one small kernel does the rest where the data lives:
//   phase 1, one lane per instance: inverse (the host library's formula, rz_linalg.h, operation for operation),
//            DevInstance 3x4 blocks, reference-layout BVHInstance, world AABB from the 8 corners;
//   phase 2, lane 0: the TLAS build exactly as BVH.cpp:178-240 (midpoint split on the longest axis, swap partition,
//            count/2 fallback, one instance per leaf, left subtree numbered first) -- it is a sequential algorithm
//            whose output ORDER is part of the result, and a TLAS has tens to thousands of instances, so one lane
//            walking it is microseconds.
// Output is byte-identical to SceneBuffers::updateDynamic (tests/test_gpu_cases.py), so frames rendered from it are
// the same bits as frames rendered from a host-built TLAS.
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_scene_dev.h"

namespace rz {

struct TlasWork {
    const float* transforms;        // n x 16, column-major
    DevInstance* instances;         // in/out: fwd, inv rewritten; root box / bases kept
    rz_bvh_instance* refInstances;  // out: transform + inverseTransform (offsets kept)
    TlasNode* nodes;                // out: 2n-1 nodes
    int32_t* indices;               // out: n
    float* worldMin;                // scratch n x 3
    float* worldMax;                // scratch n x 3
    int32_t* order;                 // scratch n (meshIndices)
    int32_t* stack;                 // scratch 3 x (2n+8)
    int32_t* outCounts;             // [0] = node count, [1] = index count, [2] = depth
    int n;
};

__device__ inline float gmin(float a, float b) { return (b < a) ? b : a; }   // glm::min
__device__ inline float gmax(float a, float b) { return (a < b) ? b : a; }   // glm::max

// rz_linalg.h inverse(): same expressions, same order
__device__ void inverse4(const float* m, float* r) {
    float s0 = m[0] * m[5] - m[1] * m[4], s1 = m[0] * m[6] - m[2] * m[4], s2 = m[0] * m[7] - m[3] * m[4];
    float s3 = m[1] * m[6] - m[2] * m[5], s4 = m[1] * m[7] - m[3] * m[5], s5 = m[2] * m[7] - m[3] * m[6];
    float c5 = m[10] * m[15] - m[11] * m[14], c4 = m[9] * m[15] - m[11] * m[13], c3 = m[9] * m[14] - m[10] * m[13];
    float c2 = m[8] * m[15] - m[11] * m[12], c1 = m[8] * m[14] - m[10] * m[12], c0 = m[8] * m[13] - m[9] * m[12];
    float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    float id = 1.0f / det;
    r[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    r[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    r[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    r[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    r[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    r[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    r[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    r[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    r[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    r[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    r[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    r[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    r[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    r[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    r[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    r[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
}

__global__ __launch_bounds__(256) void rz_tlas_refit(const TlasWork W) {
    const int n = W.n;
    // ---- phase 1
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float m[16], inv[16];
        for (int k = 0; k < 16; ++k) m[k] = W.transforms[(size_t)i * 16 + k];
        inverse4(m, inv);
        DevInstance& D = W.instances[i];
        rz_bvh_instance& R = W.refInstances[i];
        for (int k = 0; k < 16; ++k) { R.transform[k] = m[k]; R.inverseTransform[k] = inv[k]; }
        for (int col = 0; col < 4; ++col)
            for (int row = 0; row < 3; ++row) {
                D.fwd[col * 3 + row] = m[col * 4 + row];
                D.inv[col * 3 + row] = inv[col * 4 + row];
            }
        // main.cpp:974-993: 8 corners of the BLAS root box, (+-1e30) start, glm mat4*vec4 with w = 1
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int c = 0; c < 8; ++c) {
            const float x = (c & 4) ? D.rootMax[0] : D.rootMin[0], y = (c & 2) ? D.rootMax[1] : D.rootMin[1],
                        z = (c & 1) ? D.rootMax[2] : D.rootMin[2];
            for (int r = 0; r < 3; ++r) {
                const float t = m[r] * x + m[4 + r] * y + m[8 + r] * z + m[12 + r] * 1.0f;
                mn[r] = gmin(mn[r], t);
                mx[r] = gmax(mx[r], t);
            }
        }
        for (int r = 0; r < 3; ++r) { W.worldMin[3 * i + r] = mn[r]; W.worldMax[3 * i + r] = mx[r]; }
        W.order[i] = i;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    // ---- phase 2: BVH.cpp:178-240
    const float FMAX = 3.402823466e+38f;
    int32_t* stNode = W.stack;
    int32_t* stStart = W.stack + (2 * n + 8);
    int32_t* stEnd = W.stack + 2 * (2 * n + 8);
    int32_t* stDepth = W.order + n;            // scratch tail: depth of each stacked entry
    int sp = 0, nn = 1, ni = 0, depth = 1;
    stNode[0] = 0; stStart[0] = 0; stEnd[0] = n; stDepth[0] = 1; sp = 1;
    W.nodes[0] = TlasNode{{0, 0, 0}, 0, {0, 0, 0}, 0};
    while (sp > 0) {
        --sp;
        const int nidx = stNode[sp], start = stStart[sp], end = stEnd[sp], count = end - start, d = stDepth[sp];
        if (d > depth) depth = d;
        float bmin[3] = {FMAX, FMAX, FMAX}, bmax[3] = {-FMAX, -FMAX, -FMAX};
        for (int i = start; i < end; ++i) {
            const int o = W.order[i];
            for (int r = 0; r < 3; ++r) {
                bmin[r] = gmin(bmin[r], W.worldMin[3 * o + r]);
                bmax[r] = gmax(bmax[r], W.worldMax[3 * o + r]);
            }
        }
        TlasNode& N = W.nodes[nidx];
        for (int r = 0; r < 3; ++r) { N.bmin[r] = bmin[r]; N.bmax[r] = bmax[r]; }
        if (count == 1) { N.leftFirst = ni; N.count = 1; W.indices[ni++] = W.order[start]; continue; }
        if (count <= 0) { N.leftFirst = 0; N.count = 0; continue; }
        const float ex = bmax[0] - bmin[0], ey = bmax[1] - bmin[1], ez = bmax[2] - bmin[2];
        int axis = 0;
        if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
        const float split = 0.5f * (bmin[axis] + bmax[axis]);
        int mid = start;
        for (int i = start; i < end; ++i) {
            const int o = W.order[i];
            const float cen = (W.worldMin[3 * o + axis] + W.worldMax[3 * o + axis]) * 0.5f;
            if (cen < split) { const int t = W.order[i]; W.order[i] = W.order[mid]; W.order[mid] = t; ++mid; }
        }
        if (mid == start || mid == end) mid = start + (count / 2);
        const int leftIdx = nn, rightIdx = nn + 1;
        N.leftFirst = leftIdx; N.count = -1;
        W.nodes[nn] = TlasNode{{0, 0, 0}, 0, {0, 0, 0}, 0};
        W.nodes[nn + 1] = TlasNode{{0, 0, 0}, 0, {0, 0, 0}, 0};
        nn += 2;
        stNode[sp] = rightIdx; stStart[sp] = mid; stEnd[sp] = end; stDepth[sp] = d + 1; ++sp;
        stNode[sp] = leftIdx; stStart[sp] = start; stEnd[sp] = mid; stDepth[sp] = d + 1; ++sp;
    }
    W.outCounts[0] = nn; W.outCounts[1] = ni; W.outCounts[2] = depth;
}

void launch_tlas_refit(const TlasWork& W, hipStream_t s) {
    hipLaunchKernelGGL(rz_tlas_refit, dim3(1), dim3(256), 0, s, W);
}

}  // namespace rz
