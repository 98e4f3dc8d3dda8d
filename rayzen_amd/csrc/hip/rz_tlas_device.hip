// rz_tlas_device.hip -- updateDynamicBVHAndSSBOs (RayZen/src/main.cpp:1138-1194) on the GPU.
//
// Per frame RayZen re-reads every object's transform, inverts it, recomputes the world AABB of each BLAS root
// (main.cpp:1168-1191), rebuilds the TLAS on the CPU (BVH.cpp:178-240) and re-uploads everything.  Here the frontend
// hands over only the transforms (64 B per instance); one small kernel does the rest where the data lives:
//   phase 1, one lane per instance: inverse (glm::inverse's published sequence, as rz_linalg.h, operation for operation),
//            DevInstance 3x4 blocks, reference-layout BVHInstance, world AABB from the 8 corners;
//   phase 2, 16 waves, level by level: the TLAS build exactly as BVH.cpp:178-240 (midpoint split on the longest axis,
//            swap partition, count/2 fallback, one instance per leaf, left subtree numbered first).  The reference walks
//            a stack, and the ORDER in which it numbers nodes is part of the result -- but that order is a function of
//            the ranges alone: every leaf holds one instance, so a subtree over c instances has c - 1 internal nodes,
//            the k-th internal node in left-first pre-order owns nodes 2k+1 and 2k+2, its left child is internal node
//            k + 1 and its right child internal node k + (mid - start).  So the nodes of one tree level are independent:
//            each is handled by one wave (16 at a time), whose three loops over the node's instances run 64 lanes wide:
//              * bounds: glm::min / glm::max keep the FIRST of equal values (it matters for the sign of a zero), so the
//                reduction carries (value, position) and the earlier position wins ties;
//              * partition: BVH.cpp:214-222 is a Lomuto partition (swap a[i] with a[mid++] when the centre is below the
//                split).  Its result: the "below" elements in their original order, and the others in the order a queue
//                ends up in that pushes every "other" element and, for every "below" element met after the first
//                "other", moves its head to its tail.  Position p (from the first "other", f0) makes push number
//                p - f0; a "below" element's push repeats push number (belows in [f0, p)), an earlier one, so the queue's
//                contents resolve by pointer jumping (6 shuffle rounds inside one 64-element chunk, log2(m) passes
//                through scratch for larger nodes); the last (others) pushes are the final order.
//            (Round 1 walked everything on one lane: 7 ms per frame at 1 025 instances, 4 us per node of dependent
//             memory round trips; a level now costs that once per 16 nodes.)
// Output is byte-identical to SceneBuffers::updateDynamic (tests/test_gpu_cases.py), so frames rendered from it are
// the same bits as frames rendered from a host-built TLAS.
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_internal.h"

namespace rz {


__device__ inline float gmin(float a, float b) { return (b < a) ? b : a; }   // glm::min
__device__ inline float gmax(float a, float b) { return (a < b) ? b : a; }   // glm::max

// glm::inverse(mat4), GLM 0.9.9.8's compute_inverse<4, 4> (the algorithm RayZen's main.cpp:1151 runs per object per frame;
// rz_linalg.h and the oracle restate the same published sequence): 2x2 sub-determinants of rows 1..3, cofactor columns
// (Vec_a * Fac_i - Vec_b * Fac_j) + Vec_c * Fac_l with alternating signs, determinant from the cofactors' first row
// summed pairwise, every cofactor times its reciprocal.  One lane inverts one instance's matrix; -ffp-contract=off keeps
// each product and sum a separate rounding, as on the host.
__device__ void inverse4(const float* m, float* r) {
    // GLM's eighteen coefficients are the 2x2 determinants of two ROWS (a < b) taken from two of the columns 1, 2, 3;
    // Fac_f holds the three of one row pair -- (2,3) (1,3) (1,2) (0,3) (0,2) (0,1) for f = 0..5 -- its first one twice
    const int ra[6] = {2, 1, 1, 0, 0, 0}, rb[6] = {3, 3, 2, 3, 2, 1};
    float fac[6][3];
    for (int f = 0; f < 6; ++f) {
        const int a = ra[f], b = rb[f];
        fac[f][0] = m[8 + a] * m[12 + b] - m[12 + a] * m[8 + b];      // columns 2, 3
        fac[f][1] = m[4 + a] * m[12 + b] - m[12 + a] * m[4 + b];      // columns 1, 3
        fac[f][2] = m[4 + a] * m[8 + b] - m[8 + a] * m[4 + b];        // columns 1, 2
    }
    const int va[4] = {1, 0, 0, 0}, vb[4] = {2, 2, 1, 1}, vc[4] = {3, 3, 3, 2};
    const int fa[4] = {0, 0, 1, 2}, fb[4] = {1, 3, 3, 4}, fc[4] = {2, 4, 5, 5};
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) {
            const int col = j == 0 ? 4 : 0;             // Vec_x[j] = m[1][x] for j = 0, m[0][x] otherwise
            const int fj = j == 0 ? 0 : j - 1;          // Fac_x = (c, c, c', c'')
            const float inv = (m[col + va[k]] * fac[fa[k]][fj] - m[col + vb[k]] * fac[fb[k]][fj]) + m[col + vc[k]] * fac[fc[k]][fj];
            r[4 * k + j] = ((k + j) & 1) ? inv * -1.0f : inv * 1.0f;
        }
    const float det = (m[0] * r[0] + m[1] * r[4]) + (m[2] * r[8] + m[3] * r[12]);
    const float id = 1.0f / det;
    for (int k = 0; k < 16; ++k) r[k] = r[k] * id;
}

__global__ __launch_bounds__(1024) void rz_tlas_refit(const TlasWork W) {
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
                const float t = (m[r] * x + m[4 + r] * y) + (m[8 + r] * z + m[12 + r] * 1.0f);     // glm mat4 * vec4: pairwise
                mn[r] = gmin(mn[r], t);
                mx[r] = gmax(mx[r], t);
            }
        }
        for (int r = 0; r < 3; ++r) { W.worldMin[3 * i + r] = mn[r]; W.worldMax[3 * i + r] = mx[r]; }
        W.order[i] = i;
    }
    __syncthreads();
    // ---- phase 2: BVH.cpp:178-240, level by level, one wave per node
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const float FMAX = 3.402823466e+38f;
    int32_t* rankA = W.scratch;                // (trues before p) << 1 | below, for nodes of more than 64 instances
    int32_t* ptrA = W.scratch + n;
    int32_t* ptrB = W.scratch + 2 * n;
    int32_t* newOrder = W.scratch + 3 * n;
    // level lists: (node index, internal pre-order index k, start, end, pop position, stack entries below it when popped).
    // The pop order (right child first) is a function of the ranges too: a subtree over c instances has 2c - 1 nodes, so
    // the right child is popped at pos + 1, the left child after the right subtree at pos + 2 (end - mid), and what
    // follows the node's subtree at pos + 2c - 1; the left child waits below the right one (rz_context.hip: tlas_pop_order).
    int32_t* listA = W.scratch + 4 * n;
    int32_t* listB = listA + 6 * (n + 1);
    __shared__ int curCount, nextCount;
    if (threadIdx.x == 0) {
        listA[0] = 0; listA[1] = 0; listA[2] = 0; listA[3] = n; listA[4] = 0; listA[5] = 0;
        curCount = 1; nextCount = 0;
    }
    __threadfence_block();
    __syncthreads();
    int depth = 0;
    int32_t* cur = listA; int32_t* nxt = listB;
    for (unsigned level = 0; level < (1u << 24); ++level) {
        const int cnt = curCount;
        if (cnt <= 0) break;
        depth = (int)level + 1;
        for (int item = wave; item < cnt; item += nWaves) {
            const int nidx = cur[6 * item], k = cur[6 * item + 1], start = cur[6 * item + 2], end = cur[6 * item + 3];
            const int dpos = cur[6 * item + 4], below = cur[6 * item + 5];
            const int count = end - start;
            // -- bounds (BVH.cpp:186-191): first-of-equals extremum per axis, 64 lanes wide
            float bmin[3], bmax[3];
            for (int r = 0; r < 3; ++r) {
                float vmin = FMAX, vmax = -FMAX;
                int pmin = 0x7fffffff, pmax = 0x7fffffff;
                for (int p = start + lane; p < end; p += 64) {
                    const int o = W.order[p];
                    const float a = W.worldMin[3 * o + r], b = W.worldMax[3 * o + r];
                    if (a < vmin) { vmin = a; pmin = p; }
                    if (vmax < b) { vmax = b; pmax = p; }
                }
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(vmin, off); const int op = __shfl_xor(pmin, off);
                    if (ov < vmin || (ov == vmin && op < pmin)) { vmin = ov; pmin = op; }
                    const float ow = __shfl_xor(vmax, off); const int oq = __shfl_xor(pmax, off);
                    if (vmax < ow || (ow == vmax && oq < pmax)) { vmax = ow; pmax = oq; }
                }
                bmin[r] = vmin; bmax[r] = vmax;
            }
            TlasNode N;
            TlasDfs R = {};
            for (int r = 0; r < 3; ++r) { N.bmin[r] = bmin[r]; N.bmax[r] = bmax[r]; R.bmin[r] = bmin[r]; R.bmax[r] = bmax[r]; }
            if (count == 1) {
                // every leaf holds exactly one instance, so the number of leaves the reference has written before this one
                // is `start`
                N.leftFirst = start; N.count = 1;
                R.first = start; R.count = 1; R.skip = dpos + 1; R.inst0 = W.order[start];
                if (lane == 0) { W.nodes[nidx] = N; W.indices[start] = W.order[start]; W.dfs[dpos] = R; }
                continue;
            }
            if (count <= 0) {
                N.leftFirst = 0; N.count = 0;
                R.skip = dpos + 1;
                if (lane == 0) { W.nodes[nidx] = N; W.dfs[dpos] = R; }
                continue;
            }
            const float ex = bmax[0] - bmin[0], ey = bmax[1] - bmin[1], ez = bmax[2] - bmin[2];
            int axis = 0;
            if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
            const float split = 0.5f * (bmin[axis] + bmax[axis]);
            int mid;
            if (count <= 64) {
                // -- the whole node in one chunk: lane i holds position start + i
                const bool valid = lane < count;
                const int o = valid ? W.order[start + lane] : 0;
                const bool below = valid && ((W.worldMin[3 * o + axis] + W.worldMax[3 * o + axis]) * 0.5f < split);
                const unsigned long long T = __ballot(below), V = __ballot(valid);
                const int nT = __popcll(T);
                if (nT == 0 || nT == count) {
                    mid = start + (count / 2);                              // BVH.cpp:223: the order stays as it is
                } else {
                    mid = start + nT;
                    const int f0 = __builtin_ctzll(~T & V);                 // first "other"
                    const int rankT = __popcll(T & ((1ull << lane) - 1ull));
                    const int R = nT - f0;                                  // belows after the first other
                    int ptr = below ? (rankT - f0) : (lane - f0);           // push number lane - f0 repeats push `ptr` (or is its own)
                    for (int round = 0; round < 6; ++round) {
                        const int src = (valid && lane >= f0) ? ptr + f0 : lane;
                        const int q = __shfl(ptr, src);
                        if (valid && lane >= f0) ptr = q;
                    }
                    const int srcLane = (valid && lane >= f0) ? ptr + f0 : lane;
                    const int elem = __shfl(o, srcLane);                    // the element push number (lane - f0) carries
                    if (below) W.order[start + rankT] = o;
                    const int kk = lane - f0;
                    if (valid && lane >= f0 && kk >= R) W.order[start + nT + (kk - R)] = elem;
                }
            } else {
                // -- larger nodes: the same in passes through scratch (indexed by position: the nodes of a level are disjoint)
                int running = 0, f0 = 0x7fffffff;
                for (int base = start; base < end; base += 64) {
                    const int p = base + lane;
                    const bool valid = p < end;
                    const int o = valid ? W.order[p] : 0;
                    const bool below = valid && ((W.worldMin[3 * o + axis] + W.worldMax[3 * o + axis]) * 0.5f < split);
                    const unsigned long long T = __ballot(below), V = __ballot(valid);
                    if (valid) rankA[p] = (running + __popcll(T & ((1ull << lane) - 1ull))) * 2 + (below ? 1 : 0);
                    if (f0 == 0x7fffffff && (~T & V) != 0ull) f0 = base + __builtin_ctzll(~T & V);
                    running += __popcll(T);
                }
                const int nT = running;
                __threadfence_block();
                if (nT == 0 || nT == count) {
                    mid = start + (count / 2);
                } else {
                    mid = start + nT;
                    const int R = nT - (f0 - start);
                    for (int p = f0 + lane; p < end; p += 64) {
                        const int v = rankA[p];
                        ptrA[p] = (v & 1) ? f0 + ((v >> 1) - (f0 - start)) : p;      // the earlier push this one repeats, as a position
                    }
                    __threadfence_block();
                    int32_t* src = ptrA; int32_t* dst = ptrB;
                    for (int span = 1; span < end - f0; span <<= 1) {
                        for (int p = f0 + lane; p < end; p += 64) dst[p] = src[src[p]];
                        __threadfence_block();
                        int32_t* t = src; src = dst; dst = t;
                    }
                    for (int p = start + lane; p < end; p += 64) {
                        const int v = rankA[p];
                        if (v & 1) newOrder[start + (v >> 1)] = W.order[p];
                        if (p >= f0 && p - f0 >= R) newOrder[start + nT + (p - f0 - R)] = W.order[src[p]];
                    }
                    __threadfence_block();
                    for (int p = start + lane; p < end; p += 64) W.order[p] = newOrder[p];
                }
            }
            N.leftFirst = 2 * k + 1; N.count = -1;
            R.first = 2 * k + 1; R.count = below + 2 <= 64 ? -1 : 0; R.skip = dpos + 2 * count - 1;
            if (lane == 0) {
                W.nodes[nidx] = N;
                W.dfs[dpos] = R;
                const int pos = atomicAdd(&nextCount, 2);
                int32_t* L = nxt + 6 * pos;
                L[0] = 2 * k + 1; L[1] = k + 1; L[2] = start; L[3] = mid; L[4] = dpos + 2 * (end - mid); L[5] = below;
                L[6] = 2 * k + 2; L[7] = k + (mid - start); L[8] = mid; L[9] = end; L[10] = dpos + 1; L[11] = below + 1;
            }
        }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) { curCount = nextCount; nextCount = 0; }
        int32_t* t = cur; cur = nxt; nxt = t;
        __syncthreads();
    }
    if (threadIdx.x == 0) { W.outCounts[0] = n > 0 ? 2 * n - 1 : 1; W.outCounts[1] = n; W.outCounts[2] = depth; }
}

void launch_tlas_refit(const TlasWork& W, hipStream_t s) {
    hipLaunchKernelGGL(rz_tlas_refit, dim3(1), dim3(1024), 0, s, W);
}

}  // namespace rz
