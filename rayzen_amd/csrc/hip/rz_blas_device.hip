// rz_blas_device.hip -- BVH::buildBLAS (RayZen/src/BVH.cpp:11-175, full-sweep SAH) on the GPU, byte-identical output.
//
// The reference builds top-down on one CPU thread and re-sorts every node's triangles three times
// (O(N log^2 N): 6.5 s for 1 M triangles).  The split decisions are what must be reproduced -- they fix node
// numbering, leaf ranges and index order -- so this is the same algorithm, re-organised for the machine:
//   1. one pass computes each triangle's box and centroid with the reference's expressions;
//   2. three GLOBAL radix sorts (rocPRIM, 64-bit key = orderable(centroid[a]) << 32 | id) give the order
//      std::sort of (centroid[a], id) pairs gives for the whole mesh; that order is total, so any node's sorted
//      list is the global list restricted to the node: it only ever needs STABLE PARTITIONING, never re-sorting;
//   3. level by level (a kernel boundary is the only inter-workgroup synchronisation: nothing spins), one workgroup
//      per node of the level:  bounds (leftmost minimum in index order, so even the sign of a zero matches
//      computeBounds);  per axis a suffix scan storing right-box areas and a prefix scan evaluating
//      cost = (A_l*i + A_r*(N-i)) / (A_parent + 1e-6f) -- the reference's operations in the reference's order
//      (min/max are exact under any association, so a parallel scan gives the sequential boxes);
//      the reference keeps the FIRST strict minimum scanning axis-major, split-minor, which is the lexicographic
//      minimum of (cost, axis, i): a plain block reduction;  then the node's index range becomes the best axis'
//      order, a side flag is set per triangle, and the three lists are stably partitioned into the children;
//   4. nodes get build ids in arrival order (atomics), so a last bottom-up / top-down pair of passes computes the
//      reference's numbering: it allocates children as a pair when a node is popped and builds left subtrees first
//      (BVH.cpp:166-173), i.e. the k-th internal node in left-first pre-order owns nodes 2k+1, 2k+2.
// The midpoint fallback (BVH.cpp:135-149: no finite SAH cost) is a sequential swap-partition whose index order is
// part of the result; it is rare and is done by one lane.  NaN vertex coordinates are not supported (the
// reference's std::sort comparator is undefined on them).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "rayzen_hip.h"
#include "rz_internal.h"

namespace rz {

constexpr int BB_THREADS = 256;
constexpr int BB_WAVES = BB_THREADS / 64;
constexpr int BB_E = 8;                              // elements per thread per chunk: independent gathers in flight
constexpr int BB_CHUNK = BB_THREADS * BB_E;
constexpr float BB_FMAX = 3.402823466e+38f;
constexpr int BB_NOPOS = 0x7fffffff;
constexpr int BB_LARGE = 4096;                       // nodes larger than this are spread over one workgroup per chunk
constexpr int BB_SMALL = 64;                         // nodes this small are handled by one wave, lane = position

struct BuildNode {
    int start, end;          // range in idx / ord[*]
    int left, right;         // build ids of the children (-1: leaf)
    float bmin[3], bmax[3];
    int internals;           // internal nodes in this subtree (incl. itself)
    int depth;
};

struct BlasBuild {
    const rz_triangle* tris;
    int n;
    float* tmin; float* tmax; float* cen;        // 3 floats per triangle each
    unsigned long long* keys; unsigned long long* keysOut;   // sort scratch
    int* ord[2][3];                               // ping-pong: ids sorted along each axis, node ranges aligned
    int* idx;                                     // the reference's triIndices
    unsigned char* side;                          // per triangle: goes left
    float* rarea;                                 // per position: area of the suffix box (one axis at a time, per node)
    BuildNode* nodes;                             // capacity 2n+1
    int* levelNodes[2];                           // build ids of the current / next level's nodes with more than BB_SMALL triangles
    int* smallNodes[2];                           // ... with at most BB_SMALL triangles (leaves included): one wave each
    int* largeNodes[2];                           // ... with more than BB_LARGE triangles: one workgroup per 2048-position chunk
    int* counters;                                // [0] nodes allocated; next level's counts: [1] medium, [3] small, [4] large
    // large-node path: per level, chunk c of large node j is global chunk nodeChunkBase[j] + c
    int* nodeChunkBase;                           // [maxLarge + 1]
    float* chunkBox;                              // [4][maxC][6]: box of the chunk along axis 0..2; [3] = bounds partial in idx order
    int* chunkPos;                                // [maxC][6]: position of the leftmost extremum of the bounds partial
    float* chunkPre; float* chunkSuf;             // [3][maxC][6]: box of everything before / after the chunk
    float* chunkBestCost; int* chunkBestAt;       // [3][maxC]: cheapest split inside the chunk
    int* chunkLefts;                              // [3][maxC]: elements of the chunk that go left
    float* nodeBox;                               // [maxLarge][6]
    int* nodePick;                                // [maxLarge][4]: axis, split, mid, used SAH
    float* rarea3[3];                             // suffix-box areas, all three axes at once
    int maxC;
    rz_bvh_node* outNodes;                        // reference layout, reference order
};

__device__ __forceinline__ float gmin2(float a, float b) { return (b < a) ? b : a; }     // glm::min
__device__ __forceinline__ float gmax2(float a, float b) { return (a < b) ? b : a; }     // glm::max
__device__ __forceinline__ float area2(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return 2.0f * ((dx * dy + dy * dz) + dz * dx);
}
__device__ __forceinline__ unsigned orderable(float f) {
    if (f == 0.0f) f = 0.0f;                     // -0 and +0 compare equal in std::pair's operator<: same key
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ void box_identity(float* b) { b[0] = b[1] = b[2] = BB_FMAX; b[3] = b[4] = b[5] = -BB_FMAX; }
__device__ __forceinline__ void box_merge(float* a, const float* o) {
    a[0] = gmin2(a[0], o[0]); a[1] = gmin2(a[1], o[1]); a[2] = gmin2(a[2], o[2]);
    a[3] = gmax2(a[3], o[3]); a[4] = gmax2(a[4], o[4]); a[5] = gmax2(a[5], o[5]);
}

// ---- 1. per-triangle data + sort keys
__global__ void bb_prepare(BlasBuild B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    const rz_triangle t = B.tris[i];
    for (int k = 0; k < 3; ++k) {
        B.tmin[3 * i + k] = gmin2(t.v0[k], gmin2(t.v1[k], t.v2[k]));     // glm::min(t.v0, glm::min(t.v1, t.v2))
        B.tmax[3 * i + k] = gmax2(t.v0[k], gmax2(t.v1[k], t.v2[k]));
        B.cen[3 * i + k] = ((t.v0[k] + t.v1[k]) + t.v2[k]) / 3.0f;         // (v0 + v1 + v2) / 3.0f
    }
    B.idx[i] = i;
}
__global__ void bb_make_keys(BlasBuild B, int axis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    B.keys[i] = ((unsigned long long)orderable(B.cen[3 * i + axis]) << 32) | (unsigned)i;
}
__global__ void bb_take_ids(BlasBuild B, int axis) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B.n) return;
    B.ord[0][axis][i] = (int)(unsigned)(B.keysOut[i] & 0xFFFFFFFFull);
}
__global__ void bb_init_root(BlasBuild B) {
    BuildNode r{};
    r.start = 0; r.end = B.n; r.left = r.right = -1; r.depth = 1;
    B.nodes[0] = r;
    (B.n <= BB_SMALL ? B.smallNodes[0] : (B.n <= BB_LARGE ? B.levelNodes[0] : B.largeNodes[0]))[0] = 0;
    B.counters[0] = 1; B.counters[1] = 0; B.counters[2] = 0; B.counters[3] = 0; B.counters[4] = 0;
}

struct Best { float cost; int axis; int i; };
__device__ __forceinline__ bool better(const Best& a, const Best& b) {     // a precedes b in the reference's scan?
    if (a.cost < b.cost) return true;
    if (b.cost < a.cost) return false;
    if (a.axis != b.axis) return a.axis < b.axis;
    return a.i < b.i;
}
// leftmost extremum: value first, then the earlier position
__device__ __forceinline__ bool takes_over(bool isMin, float ov, int op, float v, int p) {
    if (isMin ? (ov < v) : (v < ov)) return true;
    return ov == v && op < p;
}

// BVH.cpp:166-173: two children, allocated as a pair; queued for the next level by size class
__device__ __forceinline__ void push_children(const BlasBuild& B, int srcBuf, int nodeId, int start, int mid, int end, int depth) {
    const int id0 = atomicAdd(&B.counters[0], 2);
    BuildNode L{}, R{};
    L.start = start; L.end = mid; L.left = L.right = -1; L.depth = depth + 1;
    R.start = mid; R.end = end; R.left = R.right = -1; R.depth = depth + 1;
    B.nodes[id0] = L; B.nodes[id0 + 1] = R;
    B.nodes[nodeId].left = id0; B.nodes[nodeId].right = id0 + 1;
    for (int k = 0; k < 2; ++k) {
        const int cn = k == 0 ? mid - start : end - mid;
        if (cn <= BB_SMALL) B.smallNodes[srcBuf ^ 1][atomicAdd(&B.counters[3], 1)] = id0 + k;
        else if (cn <= BB_LARGE) B.levelNodes[srcBuf ^ 1][atomicAdd(&B.counters[1], 1)] = id0 + k;
        else B.largeNodes[srcBuf ^ 1][atomicAdd(&B.counters[4], 1)] = id0 + k;
    }
}

// One chunk of a block-wide inclusive box scan.  loc[e] holds this thread's BB_E consecutive boxes on entry and
// their inclusive scan results (carry from earlier chunks included) on exit; sRun becomes the carry for the next
// chunk.  The caller puts a __syncthreads() between calls.
__device__ __forceinline__ void scan_chunk(float (*loc)[6], float (*sWave)[6], float* sRun) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = 1; e < BB_E; ++e) box_merge(loc[e], loc[e - 1]);
    float tot[6];
    for (int c = 0; c < 6; ++c) tot[c] = loc[BB_E - 1][c];
    for (int off = 1; off < 64; off <<= 1) {
        float o[6];
        for (int c = 0; c < 6; ++c) o[c] = __shfl_up(tot[c], off, 64);
        if (lane >= off) box_merge(tot, o);
    }
    float excl[6];
    for (int c = 0; c < 6; ++c) excl[c] = __shfl_up(tot[c], 1, 64);
    if (lane == 0) box_identity(excl);
    if (lane == 63) for (int c = 0; c < 6; ++c) sWave[wv][c] = tot[c];
    __syncthreads();
    float pre[6];
    for (int c = 0; c < 6; ++c) pre[c] = sRun[c];
    for (int w = 0; w < wv; ++w) box_merge(pre, sWave[w]);
    box_merge(excl, pre);
    for (int e = 0; e < BB_E; ++e) box_merge(loc[e], excl);
    __syncthreads();
    if (tid == BB_THREADS - 1) for (int c = 0; c < 6; ++c) sRun[c] = loc[BB_E - 1][c];
}

// ---- 3. one level.  srcBuf: which ping-pong half holds this level's sorted lists.
__global__ __launch_bounds__(BB_THREADS) void bb_level(BlasBuild B, int srcBuf, int levelCount) {
    __shared__ float sWave[BB_WAVES][6];
    __shared__ float sRun[6];
    __shared__ float sBox[6];
    __shared__ float sRedV[BB_WAVES];
    __shared__ int sRedP[BB_WAVES];
    __shared__ Best sBest[BB_WAVES];
    __shared__ int sWaveI[BB_WAVES];
    __shared__ int sCarry;
    if ((int)blockIdx.x >= levelCount) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nodeId = B.levelNodes[srcBuf][blockIdx.x];
    const BuildNode nd = B.nodes[nodeId];
    const int start = nd.start, end = nd.end, N = end - start;
    int* const* ordIn = B.ord[srcBuf];
    int* const* ordOut = B.ord[srcBuf ^ 1];

    // -- bounds in idx order (computeBounds, BVH.cpp:11-19): the sequential loop keeps the FIRST of equal values, so
    //    reduce (value, position) pairs: the winner's own bits travel with it and a zero keeps the right sign.
    {
        float bv[6]; int bp[6];
        for (int c = 0; c < 6; ++c) { bv[c] = c < 3 ? BB_FMAX : -BB_FMAX; bp[c] = BB_NOPOS; }
        for (int p = start + tid; p < end; p += BB_THREADS) {
            const int id = B.idx[p];
            for (int k = 0; k < 3; ++k) {
                const float lo = B.tmin[3 * id + k], hi = B.tmax[3 * id + k];
                if (lo < bv[k]) { bv[k] = lo; bp[k] = p; }
                if (bv[3 + k] < hi) { bv[3 + k] = hi; bp[3 + k] = p; }
            }
        }
        for (int c = 0; c < 6; ++c) {
            const bool isMin = c < 3;
            float v = bv[c]; int p = bp[c];
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_down(v, off, 64); const int op = __shfl_down(p, off, 64);
                if (takes_over(isMin, ov, op, v, p)) { v = ov; p = op; }
            }
            if (lane == 0) { sRedV[wv] = v; sRedP[wv] = p; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < BB_WAVES; ++w)
                    if (takes_over(isMin, sRedV[w], sRedP[w], v, p)) { v = sRedV[w]; p = sRedP[w]; }
                sBox[c] = v;
            }
            __syncthreads();
        }
    }
    if (tid == 0) for (int k = 0; k < 3; ++k) { B.nodes[nodeId].bmin[k] = sBox[k]; B.nodes[nodeId].bmax[k] = sBox[3 + k]; }
    if (N <= 4) return;            // leaf: left/right stay -1 (BVH.cpp:115-118)

    const float parentArea = area2(sBox);
    Best myBest{BB_FMAX, 3, BB_NOPOS};
    for (int a = 0; a < 3; ++a) {
        const int* ord = ordIn[a] + start;
        // suffix scan (BVH.cpp:58-66): q counts from the right end, position p = N-1-q
        __syncthreads();
        if (tid < 6) sRun[tid] = tid < 3 ? BB_FMAX : -BB_FMAX;
        __syncthreads();
        for (int q0 = 0; q0 < N; q0 += BB_CHUNK) {
            float loc[BB_E][6];
            for (int e = 0; e < BB_E; ++e) {
                const int q = q0 + tid * BB_E + e;
                box_identity(loc[e]);
                if (q < N) {
                    const int id = ord[N - 1 - q];
                    for (int k = 0; k < 3; ++k) { loc[e][k] = B.tmin[3 * id + k]; loc[e][3 + k] = B.tmax[3 * id + k]; }
                }
            }
            scan_chunk(loc, sWave, sRun);
            for (int e = 0; e < BB_E; ++e) {
                const int q = q0 + tid * BB_E + e;
                if (q < N) B.rarea[start + (N - 1 - q)] = area2(loc[e]);
            }
            __syncthreads();
        }
        // prefix scan (BVH.cpp:49-57) and the cost of every split (BVH.cpp:68-83)
        if (tid < 6) sRun[tid] = tid < 3 ? BB_FMAX : -BB_FMAX;
        __syncthreads();
        for (int q0 = 0; q0 < N; q0 += BB_CHUNK) {
            float loc[BB_E][6];
            for (int e = 0; e < BB_E; ++e) {
                const int q = q0 + tid * BB_E + e;
                box_identity(loc[e]);
                if (q < N) {
                    const int id = ord[q];
                    for (int k = 0; k < 3; ++k) { loc[e][k] = B.tmin[3 * id + k]; loc[e][3 + k] = B.tmax[3 * id + k]; }
                }
            }
            scan_chunk(loc, sWave, sRun);
            for (int e = 0; e < BB_E; ++e) {
                const int i = q0 + tid * BB_E + e + 1;          // split before position i: left = [0, i)
                if (i < N) {
                    const float leftArea = area2(loc[e]);
                    const float rightArea = B.rarea[start + i];
                    const float cost = (leftArea * (float)i + rightArea * (float)(N - i)) / (parentArea + 1e-6f);
                    const Best cand{cost, a, i};
                    if (cost < BB_FMAX && better(cand, myBest)) myBest = cand;   // from FLT_MAX with `<`: inf and NaN never win
                }
            }
            __syncthreads();
        }
    }
    // block arg-min
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.cost = __shfl_down(myBest.cost, off, 64); o.axis = __shfl_down(myBest.axis, off, 64); o.i = __shfl_down(myBest.i, off, 64);
        if (better(o, myBest)) myBest = o;
    }
    if (lane == 0) sBest[wv] = myBest;
    __syncthreads();
    Best win = sBest[0];
    for (int w = 1; w < BB_WAVES; ++w) if (better(sBest[w], win)) win = sBest[w];
    __syncthreads();

    int mid;
    if (win.axis < 3 && win.i > 0 && win.i < N) {
        // BVH.cpp:128-133: the node's index range becomes the best axis' sorted order
        const int* ord = ordIn[win.axis] + start;
        for (int p = tid; p < N; p += BB_THREADS) {
            const int id = ord[p];
            B.idx[start + p] = id;
            B.side[id] = (unsigned char)(p < win.i);
        }
        mid = start + win.i;
    } else {
        // BVH.cpp:135-149: midpoint fallback, sequential (its swap order is part of the result)
        if (tid == 0) {
            int axis = 0;
            const float ex = sBox[3] - sBox[0], ey = sBox[4] - sBox[1], ez = sBox[5] - sBox[2];
            if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
            const float split = 0.5f * (sBox[axis] + sBox[3 + axis]);
            int m = start;
            for (int p = start; p < end; ++p) {
                if (B.cen[3 * B.idx[p] + axis] < split) { const int t = B.idx[p]; B.idx[p] = B.idx[m]; B.idx[m] = t; ++m; }
            }
            if (m == start || m == end) m = start + (N / 2);
            for (int p = start; p < end; ++p) B.side[B.idx[p]] = (unsigned char)(p < m);
            sCarry = m;
        }
        __syncthreads();
        mid = sCarry;
    }
    __threadfence_block();
    __syncthreads();
    // stable partition of the three sorted lists by side: lefts keep their order, then rights keep theirs
    const int nLeft = mid - start;
    for (int a = 0; a < 3; ++a) {
        const int* in = ordIn[a] + start;
        int* out = ordOut[a] + start;
        __syncthreads();
        if (tid == 0) sCarry = 0;
        __syncthreads();
        for (int q0 = 0; q0 < N; q0 += BB_CHUNK) {
            int ids[BB_E]; int flag[BB_E];
            int cnt = 0;
            for (int e = 0; e < BB_E; ++e) {
                const int q = q0 + tid * BB_E + e;
                ids[e] = q < N ? in[q] : -1;
            }
            for (int e = 0; e < BB_E; ++e) { flag[e] = (ids[e] >= 0 && B.side[ids[e]]) ? 1 : 0; cnt += flag[e]; }
            int incl = cnt;
            for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
            if (lane == 63) sWaveI[wv] = incl;
            __syncthreads();
            int before = sCarry + incl - cnt;                       // lefts among positions before this thread's first
            for (int w = 0; w < wv; ++w) before += sWaveI[w];
            for (int e = 0; e < BB_E; ++e) {
                const int q = q0 + tid * BB_E + e;
                if (q < N) {
                    if (flag[e]) out[before] = ids[e];
                    else out[nLeft + (q - before)] = ids[e];
                    before += flag[e];
                }
            }
            __syncthreads();
            if (tid == BB_THREADS - 1) sCarry = before;
            __syncthreads();
        }
    }
    if (tid == 0) push_children(B, srcBuf, nodeId, start, mid, end, nd.depth);
}

// ---- 3b. the same step for nodes of at most 64 triangles: one wave per node, lane = position, shuffles only.
// Child allocation is aggregated per block (one atomic per counter per 16 nodes): a single word takes ~88 atomics/us.
constexpr int BB_SMALL_THREADS = 1024;
constexpr int BB_SMALL_WAVES = BB_SMALL_THREADS / 64;
__global__ __launch_bounds__(BB_SMALL_THREADS) void bb_level_small(BlasBuild B, int srcBuf, int count) {
    __shared__ int sMid[BB_SMALL_WAVES];          // split position of the wave's node, or -1: no children
    __shared__ int sBase[3];                      // node id base, big-list base, small-list base
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w = blockIdx.x * BB_SMALL_WAVES + wv;
    int nodeId = -1, start = 0, end = 0, depth = 0, mid = -1;
    if (w < count) {                                         // wave-uniform
        nodeId = B.smallNodes[srcBuf][w];
        const BuildNode nd = B.nodes[nodeId];
        start = nd.start; end = nd.end; depth = nd.depth;
    }
    const int N = end - start;
    const bool valid = lane < N;
    if (nodeId >= 0) {
        float box[6];
        {   // bounds, leftmost extremum in idx order
            float b[6];
            box_identity(b);
            if (valid) {
                const int id = B.idx[start + lane];
                for (int k = 0; k < 3; ++k) { b[k] = B.tmin[3 * id + k]; b[3 + k] = B.tmax[3 * id + k]; }
            }
            for (int c = 0; c < 6; ++c) {
                const bool isMin = c < 3;
                const float init = isMin ? BB_FMAX : -BB_FMAX;
                float v = b[c];
                int p = (valid && (isMin ? (v < init) : (init < v))) ? lane : BB_NOPOS;   // a value that does not beat +-FLT_MAX never enters
                if (p == BB_NOPOS) v = init;
                for (int off = 32; off > 0; off >>= 1) {
                    const float ov = __shfl_xor(v, off, 64); const int op = __shfl_xor(p, off, 64);
                    if (takes_over(isMin, ov, op, v, p)) { v = ov; p = op; }
                }
                box[c] = v;
            }
        }
        if (lane == 0) for (int k = 0; k < 3; ++k) { B.nodes[nodeId].bmin[k] = box[k]; B.nodes[nodeId].bmax[k] = box[3 + k]; }
        if (N > 4) {
            const float parentArea = area2(box);
            Best best{BB_FMAX, 3, BB_NOPOS};
            int ida[3];
            for (int a = 0; a < 3; ++a) {
                float pre[6], suf[6];
                box_identity(pre);
                ida[a] = valid ? B.ord[srcBuf][a][start + lane] : -1;
                if (valid) for (int k = 0; k < 3; ++k) { pre[k] = B.tmin[3 * ida[a] + k]; pre[3 + k] = B.tmax[3 * ida[a] + k]; }
                for (int c = 0; c < 6; ++c) suf[c] = pre[c];
                for (int off = 1; off < 64; off <<= 1) {
                    float o[6], q[6];
                    for (int c = 0; c < 6; ++c) { o[c] = __shfl_up(pre[c], off, 64); q[c] = __shfl_down(suf[c], off, 64); }
                    if (lane >= off) box_merge(pre, o);
                    if (lane + off < 64) box_merge(suf, q);
                }
                const float rightArea = __shfl_down(area2(suf), 1, 64);      // the suffix box starting at position lane+1
                const int i = lane + 1;
                if (i < N) {
                    const float leftArea = area2(pre);
                    const float cost = (leftArea * (float)i + rightArea * (float)(N - i)) / (parentArea + 1e-6f);
                    const Best cand{cost, a, i};
                    if (cost < BB_FMAX && better(cand, best)) best = cand;
                }
            }
            for (int off = 32; off > 0; off >>= 1) {
                Best o;
                o.cost = __shfl_xor(best.cost, off, 64); o.axis = __shfl_xor(best.axis, off, 64); o.i = __shfl_xor(best.i, off, 64);
                if (better(o, best)) best = o;
            }
            if (best.axis < 3 && best.i > 0 && best.i < N) {
                const int id = best.axis == 0 ? ida[0] : (best.axis == 1 ? ida[1] : ida[2]);
                if (valid) { B.idx[start + lane] = id; B.side[id] = (unsigned char)(lane < best.i); }
                mid = start + best.i;
            } else {
                int m = 0;
                if (lane == 0) {
                    int axis = 0;
                    const float ex = box[3] - box[0], ey = box[4] - box[1], ez = box[5] - box[2];
                    if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
                    const float split = 0.5f * (box[axis] + box[3 + axis]);
                    m = start;
                    for (int p = start; p < end; ++p) {
                        if (B.cen[3 * B.idx[p] + axis] < split) { const int t = B.idx[p]; B.idx[p] = B.idx[m]; B.idx[m] = t; ++m; }
                    }
                    if (m == start || m == end) m = start + (N / 2);
                    for (int p = start; p < end; ++p) B.side[B.idx[p]] = (unsigned char)(p < m);
                }
                mid = __shfl(m, 0, 64);
            }
            __threadfence_block();                                   // side[] was written by other lanes of this wave
            const int nLeft = mid - start;
            for (int a = 0; a < 3; ++a) {
                const bool isL = valid && B.side[ida[a]] != 0;
                const unsigned long long mask = __ballot(isL);
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                if (valid) B.ord[srcBuf ^ 1][a][start + (isL ? before : nLeft + (lane - before))] = ida[a];
            }
        }
    }
    // children of every splitting node of this block: ids are allocated as pairs; all children are small
    if (lane == 0) sMid[wv] = mid;
    __syncthreads();
    if (threadIdx.x == 0) {
        int splits = 0;
        for (int k = 0; k < BB_SMALL_WAVES; ++k) splits += sMid[k] >= 0;
        sBase[0] = splits ? atomicAdd(&B.counters[0], 2 * splits) : 0;
        sBase[2] = splits ? atomicAdd(&B.counters[3], 2 * splits) : 0;
    }
    __syncthreads();
    if (lane == 0 && mid >= 0) {
        int before = 0;
        for (int k = 0; k < wv; ++k) before += sMid[k] >= 0;
        const int id0 = sBase[0] + 2 * before, slot = sBase[2] + 2 * before;
        BuildNode L{}, R{};
        L.start = start; L.end = mid; L.left = L.right = -1; L.depth = depth + 1;
        R.start = mid; R.end = end; R.left = R.right = -1; R.depth = depth + 1;
        B.nodes[id0] = L; B.nodes[id0 + 1] = R;
        B.nodes[nodeId].left = id0; B.nodes[nodeId].right = id0 + 1;
        B.smallNodes[srcBuf ^ 1][slot] = id0;
        B.smallNodes[srcBuf ^ 1][slot + 1] = id0 + 1;
    }
}

// ---- 3c. nodes of more than BB_LARGE triangles: the same step, one workgroup per 2048-position chunk.  Chunk totals
// are scanned per node into carries, so every chunk can run its part of the prefix / suffix scans on its own.
__device__ __forceinline__ bool locate(const BlasBuild& B, int srcBuf, int nLarge, int c, int& j, int& nodeId, int& start, int& N, int& cl) {
    if (c >= B.nodeChunkBase[nLarge]) return false;
    int lo = 0, hi = nLarge - 1;
    while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (B.nodeChunkBase[m] <= c) lo = m; else hi = m - 1; }
    j = lo; nodeId = B.largeNodes[srcBuf][j];
    start = B.nodes[nodeId].start; N = B.nodes[nodeId].end - start; cl = c - B.nodeChunkBase[j];
    return true;
}
__device__ __forceinline__ void load_box(const BlasBuild& B, int id, float* b) {
    for (int k = 0; k < 3; ++k) { b[k] = B.tmin[3 * id + k]; b[3 + k] = B.tmax[3 * id + k]; }
}
// thread 0 ends up with the block's merged box
__device__ __forceinline__ void block_reduce_box(float* b, float (*sWave)[6]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        float o[6];
        for (int c = 0; c < 6; ++c) o[c] = __shfl_down(b[c], off, 64);
        box_merge(b, o);
    }
    if (lane == 0) for (int c = 0; c < 6; ++c) sWave[wv][c] = b[c];
    __syncthreads();
    if (threadIdx.x == 0) for (int w = 1; w < BB_WAVES; ++w) box_merge(b, sWave[w]);
}
// thread 0 ends up with the leftmost extremum of the block's (value, position) pairs
__device__ __forceinline__ void block_reduce_leftmost(bool isMin, float& v, int& p, float* sRedV, int* sRedP) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(v, off, 64); const int op = __shfl_down(p, off, 64);
        if (takes_over(isMin, ov, op, v, p)) { v = ov; p = op; }
    }
    __syncthreads();
    if (lane == 0) { sRedV[wv] = v; sRedP[wv] = p; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < BB_WAVES; ++w) if (takes_over(isMin, sRedV[w], sRedP[w], v, p)) { v = sRedV[w]; p = sRedP[w]; }
}
__device__ __forceinline__ int block_sum(int v, int* sWaveI) {      // every thread gets the sum
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sWaveI[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < BB_WAVES; ++w) t += sWaveI[w];
    return t;
}

__global__ __launch_bounds__(BB_THREADS) void bbL_setup(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ int sWaveI[BB_WAVES];
    __shared__ int sCarry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) sCarry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < nLarge; j0 += BB_THREADS) {
        const int j = j0 + tid;
        int cnt = 0;
        if (j < nLarge) { const BuildNode nd = B.nodes[B.largeNodes[srcBuf][j]]; cnt = (nd.end - nd.start + BB_CHUNK - 1) / BB_CHUNK; }
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        if (lane == 63) sWaveI[wv] = incl;
        __syncthreads();
        int before = sCarry + incl - cnt;
        for (int w = 0; w < wv; ++w) before += sWaveI[w];
        if (j < nLarge) B.nodeChunkBase[j] = before;
        __syncthreads();
        if (tid == BB_THREADS - 1) sCarry = before + cnt;
        __syncthreads();
    }
    if (tid == 0) B.nodeChunkBase[nLarge] = sCarry;
}

// grid (chunks, 4): y < 3: box of the chunk along axis y;  y == 3: bounds partial of the chunk in idx order
__global__ __launch_bounds__(BB_THREADS) void bbL_totals(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ float sWave[BB_WAVES][6];
    __shared__ float sRedV[BB_WAVES];
    __shared__ int sRedP[BB_WAVES];
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    const int q0 = cl * BB_CHUNK;
    if (a < 3) {
        float b[6];
        box_identity(b);
        const int* ord = B.ord[srcBuf][a] + start;
        for (int e = 0; e < BB_E; ++e) {
            const int q = q0 + tid * BB_E + e;
            if (q < N) { float t[6]; load_box(B, ord[q], t); box_merge(b, t); }
        }
        block_reduce_box(b, sWave);
        if (tid == 0) for (int k = 0; k < 6; ++k) B.chunkBox[((size_t)a * B.maxC + c) * 6 + k] = b[k];
    } else {
        float bv[6]; int bp[6];
        for (int k = 0; k < 6; ++k) { bv[k] = k < 3 ? BB_FMAX : -BB_FMAX; bp[k] = BB_NOPOS; }
        for (int e = 0; e < BB_E; ++e) {
            const int q = q0 + tid * BB_E + e;
            if (q < N) {
                float t[6];
                load_box(B, B.idx[start + q], t);
                for (int k = 0; k < 3; ++k) {
                    if (t[k] < bv[k]) { bv[k] = t[k]; bp[k] = start + q; }
                    if (bv[3 + k] < t[3 + k]) { bv[3 + k] = t[3 + k]; bp[3 + k] = start + q; }
                }
            }
        }
        for (int k = 0; k < 6; ++k) {
            float v = bv[k]; int p = bp[k];
            block_reduce_leftmost(k < 3, v, p, sRedV, sRedP);
            if (tid == 0) { B.chunkBox[((size_t)3 * B.maxC + c) * 6 + k] = v; B.chunkPos[(size_t)c * 6 + k] = p; }
        }
    }
}

// grid (large nodes): node bounds from the partials; exclusive prefix / suffix boxes of the chunk totals per axis
__global__ __launch_bounds__(BB_THREADS) void bbL_node(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ float sWave[BB_WAVES][6];
    __shared__ float sRun[6];
    __shared__ float sRedV[BB_WAVES];
    __shared__ int sRedP[BB_WAVES];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int nodeId = B.largeNodes[srcBuf][j];
    const int c0 = B.nodeChunkBase[j], C = B.nodeChunkBase[j + 1] - c0;
    for (int k = 0; k < 6; ++k) {
        const bool isMin = k < 3;
        float v = isMin ? BB_FMAX : -BB_FMAX; int p = BB_NOPOS;
        for (int cc = tid; cc < C; cc += BB_THREADS) {
            const float ov = B.chunkBox[((size_t)3 * B.maxC + c0 + cc) * 6 + k]; const int op = B.chunkPos[(size_t)(c0 + cc) * 6 + k];
            if (takes_over(isMin, ov, op, v, p)) { v = ov; p = op; }
        }
        block_reduce_leftmost(isMin, v, p, sRedV, sRedP);
        if (tid == 0) { B.nodeBox[j * 6 + k] = v; if (isMin) B.nodes[nodeId].bmin[k] = v; else B.nodes[nodeId].bmax[k - 3] = v; }
    }
    for (int a = 0; a < 3; ++a) {
        const float* tot = B.chunkBox + ((size_t)a * B.maxC + c0) * 6;
        float* pre = B.chunkPre + ((size_t)a * B.maxC + c0) * 6;
        float* suf = B.chunkSuf + ((size_t)a * B.maxC + c0) * 6;
        for (int dir = 0; dir < 2; ++dir) {
            __syncthreads();
            if (tid < 6) sRun[tid] = tid < 3 ? BB_FMAX : -BB_FMAX;
            __syncthreads();
            for (int g0 = 0; g0 < C; g0 += BB_CHUNK) {
                float loc[BB_E][6];
                for (int e = 0; e < BB_E; ++e) {
                    const int r = g0 + tid * BB_E + e;
                    box_identity(loc[e]);
                    if (r < C) { const int g = dir == 0 ? r : C - 1 - r; for (int k = 0; k < 6; ++k) loc[e][k] = tot[(size_t)g * 6 + k]; }
                }
                scan_chunk(loc, sWave, sRun);
                for (int e = 0; e < BB_E; ++e) {
                    const int r = g0 + tid * BB_E + e;
                    if (r + 1 < C) {                           // inclusive up to r = exclusive for the next chunk in scan order
                        const int g = dir == 0 ? r + 1 : C - 2 - r;
                        float* dst = (dir == 0 ? pre : suf) + (size_t)g * 6;
                        for (int k = 0; k < 6; ++k) dst[k] = loc[e][k];
                    }
                }
                __syncthreads();
            }
        }
        if (tid < 6) { pre[tid] = tid < 3 ? BB_FMAX : -BB_FMAX; suf[(size_t)(C - 1) * 6 + tid] = tid < 3 ? BB_FMAX : -BB_FMAX; }
    }
}

// grid (chunks, 3): suffix-box areas of the chunk's positions (BVH.cpp:58-66)
__global__ __launch_bounds__(BB_THREADS) void bbL_suffix(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ float sWave[BB_WAVES][6];
    __shared__ float sRun[6];
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    const int q0 = cl * BB_CHUNK, len = (N - q0 < BB_CHUNK) ? N - q0 : BB_CHUNK;
    if (tid < 6) sRun[tid] = B.chunkSuf[((size_t)a * B.maxC + c) * 6 + tid];
    __syncthreads();
    const int* ord = B.ord[srcBuf][a] + start;
    float loc[BB_E][6];
    for (int e = 0; e < BB_E; ++e) {
        const int r = tid * BB_E + e;
        box_identity(loc[e]);
        if (r < len) load_box(B, ord[q0 + len - 1 - r], loc[e]);
    }
    scan_chunk(loc, sWave, sRun);
    for (int e = 0; e < BB_E; ++e) {
        const int r = tid * BB_E + e;
        if (r < len) B.rarea3[a][start + q0 + len - 1 - r] = area2(loc[e]);
    }
}

// grid (chunks, 3): prefix boxes and the cost of every split inside the chunk (BVH.cpp:49-57, 68-83)
__global__ __launch_bounds__(BB_THREADS) void bbL_cost(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ float sWave[BB_WAVES][6];
    __shared__ float sRun[6];
    __shared__ Best sBest[BB_WAVES];
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    const int q0 = cl * BB_CHUNK, len = (N - q0 < BB_CHUNK) ? N - q0 : BB_CHUNK;
    if (tid < 6) sRun[tid] = B.chunkPre[((size_t)a * B.maxC + c) * 6 + tid];
    __syncthreads();
    const int* ord = B.ord[srcBuf][a] + start;
    float loc[BB_E][6];
    for (int e = 0; e < BB_E; ++e) {
        const int r = tid * BB_E + e;
        box_identity(loc[e]);
        if (r < len) load_box(B, ord[q0 + r], loc[e]);
    }
    scan_chunk(loc, sWave, sRun);
    const float parentArea = area2(B.nodeBox + j * 6);
    Best myBest{BB_FMAX, 3, BB_NOPOS};
    for (int e = 0; e < BB_E; ++e) {
        const int r = tid * BB_E + e, i = q0 + r + 1;
        if (r < len && i < N) {
            const float leftArea = area2(loc[e]);
            const float rightArea = B.rarea3[a][start + i];
            const float cost = (leftArea * (float)i + rightArea * (float)(N - i)) / (parentArea + 1e-6f);
            const Best cand{cost, a, i};
            if (cost < BB_FMAX && better(cand, myBest)) myBest = cand;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.cost = __shfl_down(myBest.cost, off, 64); o.axis = __shfl_down(myBest.axis, off, 64); o.i = __shfl_down(myBest.i, off, 64);
        if (better(o, myBest)) myBest = o;
    }
    if (lane == 0) sBest[wv] = myBest;
    __syncthreads();
    if (tid == 0) {
        Best win = sBest[0];
        for (int w = 1; w < BB_WAVES; ++w) if (better(sBest[w], win)) win = sBest[w];
        B.chunkBestCost[(size_t)a * B.maxC + c] = win.cost;
        B.chunkBestAt[(size_t)a * B.maxC + c] = win.axis < 3 ? win.i : -1;
    }
}

// grid (large nodes): the node's split = lexicographic minimum over its chunks; children
__global__ __launch_bounds__(BB_THREADS) void bbL_pick(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ Best sBest[BB_WAVES];
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nodeId = B.largeNodes[srcBuf][j];
    const BuildNode nd = B.nodes[nodeId];
    const int start = nd.start, end = nd.end, N = end - start;
    const int c0 = B.nodeChunkBase[j], C = B.nodeChunkBase[j + 1] - c0;
    Best myBest{BB_FMAX, 3, BB_NOPOS};
    for (int k = tid; k < 3 * C; k += BB_THREADS) {
        const int a = k / C, cc = k - a * C;
        const int at = B.chunkBestAt[(size_t)a * B.maxC + c0 + cc];
        if (at >= 0) { const Best cand{B.chunkBestCost[(size_t)a * B.maxC + c0 + cc], a, at}; if (better(cand, myBest)) myBest = cand; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.cost = __shfl_down(myBest.cost, off, 64); o.axis = __shfl_down(myBest.axis, off, 64); o.i = __shfl_down(myBest.i, off, 64);
        if (better(o, myBest)) myBest = o;
    }
    if (lane == 0) sBest[wv] = myBest;
    __syncthreads();
    if (tid != 0) return;
    Best win = sBest[0];
    for (int w = 1; w < BB_WAVES; ++w) if (better(sBest[w], win)) win = sBest[w];
    int mid, used = 0;
    if (win.axis < 3 && win.i > 0 && win.i < N) { mid = start + win.i; used = 1; }
    else {          // BVH.cpp:135-149, sequential
        const float* box = B.nodeBox + j * 6;
        int axis = 0;
        const float ex = box[3] - box[0], ey = box[4] - box[1], ez = box[5] - box[2];
        if (ey > ex && ey > ez) axis = 1; else if (ez > ex) axis = 2;
        const float split = 0.5f * (box[axis] + box[3 + axis]);
        int m = start;
        for (int p = start; p < end; ++p) {
            if (B.cen[3 * B.idx[p] + axis] < split) { const int t = B.idx[p]; B.idx[p] = B.idx[m]; B.idx[m] = t; ++m; }
        }
        if (m == start || m == end) m = start + (N / 2);
        for (int p = start; p < end; ++p) B.side[B.idx[p]] = (unsigned char)(p < m);
        mid = m;
    }
    B.nodePick[j * 4 + 0] = win.axis; B.nodePick[j * 4 + 1] = win.i; B.nodePick[j * 4 + 2] = mid; B.nodePick[j * 4 + 3] = used;
    push_children(B, srcBuf, nodeId, start, mid, end, nd.depth);
}

// grid (chunks): BVH.cpp:128-133 -- the index range becomes the best axis' order; side flags
__global__ __launch_bounds__(BB_THREADS) void bbL_mark(BlasBuild B, int srcBuf, int nLarge) {
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, tid = threadIdx.x;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    if (!B.nodePick[j * 4 + 3]) return;
    const int axis = B.nodePick[j * 4 + 0], split = B.nodePick[j * 4 + 1];
    const int* ord = B.ord[srcBuf][axis] + start;
    for (int e = 0; e < BB_E; ++e) {
        const int q = cl * BB_CHUNK + e * BB_THREADS + tid;          // coalesced: order inside the chunk does not matter here
        if (q < N) { const int id = ord[q]; B.idx[start + q] = id; B.side[id] = (unsigned char)(q < split); }
    }
}

// grid (chunks, 3): how many of the chunk's elements go left
__global__ __launch_bounds__(BB_THREADS) void bbL_count(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ int sWaveI[BB_WAVES];
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    const int* ord = B.ord[srcBuf][a] + start;
    int cnt = 0;
    for (int e = 0; e < BB_E; ++e) {
        const int q = cl * BB_CHUNK + e * BB_THREADS + tid;
        if (q < N) cnt += B.side[ord[q]] ? 1 : 0;
    }
    cnt = block_sum(cnt, sWaveI);
    if (tid == 0) B.chunkLefts[(size_t)a * B.maxC + c] = cnt;
}

// grid (chunks, 3): stable partition of the chunk into the children's ranges
__global__ __launch_bounds__(BB_THREADS) void bbL_scatter(BlasBuild B, int srcBuf, int nLarge) {
    __shared__ int sWaveI[BB_WAVES];
    int j, nodeId, start, N, cl;
    const int c = blockIdx.x, a = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!locate(B, srcBuf, nLarge, c, j, nodeId, start, N, cl)) return;
    int part = 0;
    for (int cc = tid; cc < cl; cc += BB_THREADS) part += B.chunkLefts[(size_t)a * B.maxC + (c - cl) + cc];
    const int carry = block_sum(part, sWaveI);                  // lefts in the node's earlier chunks
    __syncthreads();
    const int nLeft = B.nodePick[j * 4 + 2] - start;
    const int* in = B.ord[srcBuf][a] + start;
    int* out = B.ord[srcBuf ^ 1][a] + start;
    const int q0 = cl * BB_CHUNK;
    int ids[BB_E]; int flag[BB_E];
    int cnt = 0;
    for (int e = 0; e < BB_E; ++e) { const int q = q0 + tid * BB_E + e; ids[e] = q < N ? in[q] : -1; }
    for (int e = 0; e < BB_E; ++e) { flag[e] = (ids[e] >= 0 && B.side[ids[e]]) ? 1 : 0; cnt += flag[e]; }
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
    if (lane == 63) sWaveI[wv] = incl;
    __syncthreads();
    int before = carry + incl - cnt;
    for (int w = 0; w < wv; ++w) before += sWaveI[w];
    for (int e = 0; e < BB_E; ++e) {
        const int q = q0 + tid * BB_E + e;
        if (q < N) {
            if (flag[e]) out[before] = ids[e];
            else out[nLeft + (q - before)] = ids[e];
            before += flag[e];
        }
    }
}

// ---- 4. numbering.  bottom-up: internal-node counts; top-down: reference indices.
__global__ void bb_count_internals(BlasBuild B, int nNodes, int depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    BuildNode& nd = B.nodes[i];
    if (nd.depth != depth) return;
    nd.internals = nd.left < 0 ? 0 : 1 + B.nodes[nd.left].internals + B.nodes[nd.right].internals;
}
// rank[i]: position of internal node i among the internal nodes in left-first pre-order (only meaningful for internal
// nodes); the children of the k-th live at 2k+1 and 2k+2.
__global__ void bb_rank_level(BlasBuild B, int* rank, int* refIdx, int nNodes, int depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    const BuildNode nd = B.nodes[i];
    if (nd.depth != depth) return;
    if (depth == 1) { refIdx[i] = 0; rank[i] = 0; }
    if (nd.left >= 0) {
        const int k = rank[i];
        refIdx[nd.left] = 2 * k + 1;
        refIdx[nd.right] = 2 * k + 2;
        rank[nd.left] = k + 1;                                             // the left subtree is built first
        rank[nd.right] = k + 1 + B.nodes[nd.left].internals;
    }
}
__global__ void bb_emit(BlasBuild B, const int* refIdx, int nNodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nNodes) return;
    const BuildNode nd = B.nodes[i];
    rz_bvh_node o;
    for (int k = 0; k < 3; ++k) { o.boundsMin[k] = nd.bmin[k]; o.boundsMax[k] = nd.bmax[k]; }
    if (nd.left < 0) { o.leftFirst = nd.start; o.count = nd.end - nd.start; }
    else { o.leftFirst = refIdx[nd.left]; o.count = -1; }
    B.outNodes[refIdx[i]] = o;
}

// ---------------------------------------------------------------------------------------------------------
static inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

size_t blas_build_workspace_bytes(size_t n) {
    const size_t N = n ? n : 1, M = 2 * N + 2;
    size_t b = 0;
    b += up256(N * sizeof(rz_triangle));
    b += 3 * up256(N * 12);
    b += 2 * up256(N * 8);
    b += 6 * up256(N * 4);
    b += up256(N * 4) + up256(N) + up256(N * 4);
    b += up256(M * sizeof(BuildNode));
    b += 6 * up256(M * 4);
    const size_t ML = N / BB_LARGE + 2, MC = N / BB_CHUNK + ML + 2;
    b += 3 * up256((ML + 1) * 4) + up256(ML * 24) + up256(ML * 16);
    b += up256(4 * MC * 24) + up256(MC * 24) + 2 * up256(3 * MC * 24) + 3 * up256(3 * MC * 4);
    b += 2 * up256(N * 4);
    b += up256(M * sizeof(rz_bvh_node));
    b += 256;
    return b;
}

#define BB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { if (temp) (void)hipFree(temp); return (int)e_; } } while (0)

// Returns 0, a hipError_t (> 0), or -1 (bad arguments / internal limit).  hostTris, nodes_out (capacity 2n+1) and
// idx_out (n) may be HOST or DEVICE pointers (copies use hipMemcpyDefault).  ms (optional): device time from the first kernel to the last, copies excluded.
int blas_build_device(const rz_triangle* hostTris, size_t n, void* workspace, size_t workspaceBytes, rz_bvh_node* nodes_out,
                      int32_t* idx_out, int* nNodesOut, int* depthOut, float* ms, hipStream_t s) {
    void* temp = nullptr;
    if (n == 0 || n > ((size_t)1 << 30) || !workspace || workspaceBytes < blas_build_workspace_bytes(n)) return -1;
    const size_t N = n, M = 2 * N + 2;
    char* w = static_cast<char*>(workspace);
    auto carve = [&](size_t bytes) { char* p = w; w += up256(bytes); return p; };
    BlasBuild B{};
    rz_triangle* devTris = reinterpret_cast<rz_triangle*>(carve(N * sizeof(rz_triangle)));
    B.tris = devTris;
    B.n = (int)n;
    B.tmin = reinterpret_cast<float*>(carve(N * 12)); B.tmax = reinterpret_cast<float*>(carve(N * 12));
    B.cen = reinterpret_cast<float*>(carve(N * 12));
    B.keys = reinterpret_cast<unsigned long long*>(carve(N * 8)); B.keysOut = reinterpret_cast<unsigned long long*>(carve(N * 8));
    for (int h = 0; h < 2; ++h) for (int a = 0; a < 3; ++a) B.ord[h][a] = reinterpret_cast<int*>(carve(N * 4));
    B.idx = reinterpret_cast<int*>(carve(N * 4));
    B.side = reinterpret_cast<unsigned char*>(carve(N));
    B.rarea = reinterpret_cast<float*>(carve(N * 4));
    B.rarea3[0] = B.rarea;
    B.nodes = reinterpret_cast<BuildNode*>(carve(M * sizeof(BuildNode)));
    B.levelNodes[0] = reinterpret_cast<int*>(carve(M * 4)); B.levelNodes[1] = reinterpret_cast<int*>(carve(M * 4));
    B.smallNodes[0] = reinterpret_cast<int*>(carve(M * 4)); B.smallNodes[1] = reinterpret_cast<int*>(carve(M * 4));
    const size_t ML = N / BB_LARGE + 2, MC = N / BB_CHUNK + ML + 2;
    B.maxC = (int)MC;
    B.largeNodes[0] = reinterpret_cast<int*>(carve((ML + 1) * 4)); B.largeNodes[1] = reinterpret_cast<int*>(carve((ML + 1) * 4));
    B.nodeChunkBase = reinterpret_cast<int*>(carve((ML + 1) * 4));
    B.nodeBox = reinterpret_cast<float*>(carve(ML * 24)); B.nodePick = reinterpret_cast<int*>(carve(ML * 16));
    B.chunkBox = reinterpret_cast<float*>(carve(4 * MC * 24)); B.chunkPos = reinterpret_cast<int*>(carve(MC * 24));
    B.chunkPre = reinterpret_cast<float*>(carve(3 * MC * 24)); B.chunkSuf = reinterpret_cast<float*>(carve(3 * MC * 24));
    B.chunkBestCost = reinterpret_cast<float*>(carve(3 * MC * 4)); B.chunkBestAt = reinterpret_cast<int*>(carve(3 * MC * 4));
    B.chunkLefts = reinterpret_cast<int*>(carve(3 * MC * 4));
    B.rarea3[1] = reinterpret_cast<float*>(carve(N * 4)); B.rarea3[2] = reinterpret_cast<float*>(carve(N * 4));
    int* rank = reinterpret_cast<int*>(carve(M * 4));
    int* refIdx = reinterpret_cast<int*>(carve(M * 4));
    B.outNodes = reinterpret_cast<rz_bvh_node*>(carve(M * sizeof(rz_bvh_node)));
    B.counters = reinterpret_cast<int*>(carve(64));

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ms) { BB_HIP(hipEventCreate(&ev0)); BB_HIP(hipEventCreate(&ev1)); }
    BB_HIP(hipMemcpyAsync(devTris, hostTris, N * sizeof(rz_triangle), hipMemcpyDefault, s));
    if (ms) BB_HIP(hipEventRecord(ev0, s));
    const int blocks = (int)((N + 255) / 256);
    hipLaunchKernelGGL(bb_prepare, dim3(blocks), dim3(256), 0, s, B);
    size_t tempBytes = 0;
    BB_HIP(rocprim::radix_sort_keys(nullptr, tempBytes, B.keys, B.keysOut, N, 0, 64, s));
    BB_HIP(hipMalloc(&temp, tempBytes ? tempBytes : 256));
    for (int a = 0; a < 3; ++a) {
        hipLaunchKernelGGL(bb_make_keys, dim3(blocks), dim3(256), 0, s, B, a);
        BB_HIP(rocprim::radix_sort_keys(temp, tempBytes, B.keys, B.keysOut, N, 0, 64, s));
        hipLaunchKernelGGL(bb_take_ids, dim3(blocks), dim3(256), 0, s, B, a);
    }
    hipLaunchKernelGGL(bb_init_root, dim3(1), dim3(1), 0, s, B);
    int largeCount = n > (size_t)BB_LARGE ? 1 : 0, smallCount = n <= (size_t)BB_SMALL ? 1 : 0, bigCount = 1 - largeCount - smallCount;
    int buf = 0, hostCounters[8] = {0}, levels = 0;
    while (largeCount > 0 || bigCount > 0 || smallCount > 0) {
        if (++levels > 4096) { (void)hipFree(temp); return -1; }
        if (largeCount > 0) {
            const unsigned g = (unsigned)(N / BB_CHUNK) + (unsigned)largeCount;       // >= sum of ceil(N_j / chunk)
            const dim3 T(BB_THREADS);
            hipLaunchKernelGGL(bbL_setup, dim3(1), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_totals, dim3(g, 4), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_node, dim3(largeCount), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_suffix, dim3(g, 3), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_cost, dim3(g, 3), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_pick, dim3(largeCount), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_mark, dim3(g), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_count, dim3(g, 3), T, 0, s, B, buf, largeCount);
            hipLaunchKernelGGL(bbL_scatter, dim3(g, 3), T, 0, s, B, buf, largeCount);
        }
        if (bigCount > 0) hipLaunchKernelGGL(bb_level, dim3(bigCount), dim3(BB_THREADS), 0, s, B, buf, bigCount);
        if (smallCount > 0)
            hipLaunchKernelGGL(bb_level_small, dim3((smallCount + BB_SMALL_WAVES - 1) / BB_SMALL_WAVES), dim3(BB_SMALL_THREADS), 0, s, B, buf, smallCount);
        BB_HIP(hipMemcpyAsync(hostCounters, B.counters, 32, hipMemcpyDeviceToHost, s));
        BB_HIP(hipStreamSynchronize(s));
        bigCount = hostCounters[1]; smallCount = hostCounters[3]; largeCount = hostCounters[4];
        BB_HIP(hipMemsetAsync(B.counters + 1, 0, 16, s));
        buf ^= 1;
    }
    const int nNodes = hostCounters[0], depth = levels;      // every level of the loop produced nodes of that depth
    const int nb = (nNodes + 255) / 256;
    for (int d = depth; d >= 1; --d) hipLaunchKernelGGL(bb_count_internals, dim3(nb), dim3(256), 0, s, B, nNodes, d);
    for (int d = 1; d <= depth; ++d) hipLaunchKernelGGL(bb_rank_level, dim3(nb), dim3(256), 0, s, B, rank, refIdx, nNodes, d);
    hipLaunchKernelGGL(bb_emit, dim3(nb), dim3(256), 0, s, B, refIdx, nNodes);
    BB_HIP(hipGetLastError());
    if (ms) BB_HIP(hipEventRecord(ev1, s));
    BB_HIP(hipMemcpyAsync(nodes_out, B.outNodes, (size_t)nNodes * sizeof(rz_bvh_node), hipMemcpyDefault, s));
    BB_HIP(hipMemcpyAsync(idx_out, B.idx, N * 4, hipMemcpyDefault, s));
    BB_HIP(hipStreamSynchronize(s));
    if (ms) { BB_HIP(hipEventElapsedTime(ms, ev0, ev1)); (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); }
    (void)hipFree(temp);
    if (nNodesOut) *nNodesOut = nNodes;
    if (depthOut) *depthOut = depth;
    return 0;
}

}  // namespace rz
