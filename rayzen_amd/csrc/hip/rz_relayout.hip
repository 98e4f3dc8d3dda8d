// rz_relayout.hip -- RayZen's BLAS arrays -> the device layout of rz_scene_dev.h, ON the device.
//
// What it replaces: the host re-layout of rz_context.hip (build_view: a breadth-first walk emitting one DevPair per
// internal node, then a gather of the triangles into leaf order) -- 19 ms at 70 k triangles, 86 ms at 1 M, which made
// the reference's "re-upload all geometry every frame" shape (RayZen/src/main.cpp:1196-1201) a host-bound step.  Same
// output, bit for bit (tests/test_relayout_gpu.py compares the two), so nothing the traversal computes changes:
//   * level by level from the root; the frontier of level d is the list of INTERNAL nodes at depth d in breadth-first
//     order, and the pair index of an internal node is its rank in that order (all levels concatenated) -- exactly the
//     host's queue order.  Per level: flag the internal children (left before right), exclusive scan (rocPRIM), write
//     the level's DevPairs and the next frontier.  Kernel boundaries are the only synchronisation; the next level's
//     size comes back through a pinned word.
//   * then one pass over the leaf slots: slot s of the view is triangle gTriOff + blasTriIndices[triOff + s]; v0 and
//     the two edge vectors (the single-rounding subtractions of FS:392-393), material index checked against the
//     material count, "uses a transparent material" or-ed into a flag.
// Everything the traversal kernels later index with is range-checked HERE, on the device, before it is used: child
// indices, leaf ranges, triangle indices, material indices, and the walk stops if it visits more nodes than the array
// holds (not a tree).  Any violation is reported as a code; the caller then runs the host re-layout, which produces
// the precise error message, and nothing unchecked is ever launched.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include <cstring>

#include "rayzen_hip.h"
#include "rz_device_math.h"
#include "rz_internal.h"

namespace rz {

struct RelayoutErr { int code; int detail; unsigned transparent; int maxSlot; };   // device word block (transparent: bit 0 a transparent material is in use, bit 1 an irregular child box was laid out)
enum : int { RL_OK = 0, RL_BAD_CHILD = 1, RL_BAD_LEAF = 2, RL_BAD_TRI = 3, RL_BAD_MAT = 4, RL_NOT_A_TREE = 5 };

namespace {

__device__ inline void report(RelayoutErr* e, int code, int detail) {
    if (atomicCAS(&e->code, RL_OK, code) == RL_OK) e->detail = detail;
}

// pass 1 of a level: which children of the frontier's nodes are internal?  (left child of entry k -> flags[2k])
__global__ void rl_flags(const rz_bvh_node* __restrict__ nodes, long long nView, const int* __restrict__ frontier, int fsize,
                         int* __restrict__ flags, RelayoutErr* err) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= fsize) return;
    const int n = frontier[k];
    const int L = nodes[n].leftFirst;
    const bool ok = L >= 1 && (long long)L + 1 < nView;
    if (!ok) report(err, RL_BAD_CHILD, n);
    flags[2 * k] = (ok && nodes[L].count < 0) ? 1 : 0;
    flags[2 * k + 1] = (ok && nodes[L + 1].count < 0) ? 1 : 0;
}

// pass 2: write this level's pairs (pair index = levelBase + k), the next frontier and its size
__global__ void rl_write(const rz_bvh_node* __restrict__ nodes, long long nView, long long nIdxAfterTriOff,
                         const int* __restrict__ frontier, int fsize, const int* __restrict__ flags,
                         const int* __restrict__ pos, DevPair* __restrict__ pairs, int levelBase, int nextBase,
                         int* __restrict__ next, int* nextSize, RelayoutErr* err) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= fsize) return;
    if (k == fsize - 1) *nextSize = pos[2 * k + 1] + flags[2 * k + 1];
    const int n = frontier[k];
    const int L = nodes[n].leftFirst;
    if (!(L >= 1 && (long long)L + 1 < nView)) return;          // reported by rl_flags
    DevPair P;
    const rz_bvh_node ln = nodes[L], rn = nodes[L + 1];
    P.lx[0] = ln.boundsMin[0]; P.lx[1] = ln.boundsMax[0]; P.ly[0] = ln.boundsMin[1]; P.ly[1] = ln.boundsMax[1];
    P.lz[0] = ln.boundsMin[2]; P.lz[1] = ln.boundsMax[2];
    P.rx[0] = rn.boundsMin[0]; P.rx[1] = rn.boundsMax[0]; P.ry[0] = rn.boundsMin[1]; P.ry[1] = rn.boundsMax[1];
    P.rz[0] = rn.boundsMin[2]; P.rz[1] = rn.boundsMax[2];
    P.pad[0] = 0; P.pad[1] = 0;
    // a child box with min > max on some axis, or a NaN plane: the octant-specialised slab test (rz_trace.h: slab_finish<OCT>)
    // picks tmin / tmax by the ray's octant instead of by min / max and is only the shader's test for regular boxes
    if (!(ln.boundsMin[0] <= ln.boundsMax[0] && ln.boundsMin[1] <= ln.boundsMax[1] && ln.boundsMin[2] <= ln.boundsMax[2] &&
          rn.boundsMin[0] <= rn.boundsMax[0] && rn.boundsMin[1] <= rn.boundsMax[1] && rn.boundsMin[2] <= rn.boundsMax[2]))
        atomicOr(&err->transparent, 2u);
    int enc[2];
    const rz_bvh_node* ch[2] = {&ln, &rn};
    for (int c = 0; c < 2; ++c) {
        if (ch[c]->count < 0) {
            const int slot = pos[2 * k + c];
            enc[c] = nextBase + slot;           // breadth-first rank of this internal child == its pair index
            next[slot] = L + c;
        } else {
            const int lf = ch[c]->leftFirst, cnt = ch[c]->count;
            if (cnt > 15 || lf < 0 || (long long)lf + cnt > nIdxAfterTriOff) { report(err, RL_BAD_LEAF, L + c); enc[c] = -1; }
            else { enc[c] = ~((lf << 4) | cnt); atomicMax(&err->maxSlot, lf + cnt); }
        }
    }
    P.lenc = enc[0]; P.renc = enc[1];
    pairs[levelBase + k] = P;
}

// leaf-order gather + material check
__global__ void rl_gather(const rz_triangle* __restrict__ tris, long long nTris, const int* __restrict__ idx, int gTriOff,
                          int nSlots, DevTri* __restrict__ out, const rz_material* __restrict__ mats, int nMat, RelayoutErr* err) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nSlots) return;
    const long long src = (long long)gTriOff + idx[s];
    if (src < 0 || src >= nTris) { report(err, RL_BAD_TRI, s); return; }
    const rz_triangle t = tris[src];
    DevTri d;
    d.v0[0] = t.v0[0]; d.v0[1] = t.v0[1]; d.v0[2] = t.v0[2];
    d.e1x = t.v1[0] - t.v0[0]; d.e1y = t.v1[1] - t.v0[1]; d.e1z = t.v1[2] - t.v0[2];   // FS:392
    d.e2x = t.v2[0] - t.v0[0]; d.e2y = t.v2[1] - t.v0[1]; d.e2z = t.v2[2] - t.v0[2];   // FS:393
    d.mat = t.materialIndex;
    d.src = (int32_t)src;
    d.pad = 0;
    out[s] = d;
    if (t.materialIndex < 0 || t.materialIndex >= nMat) { report(err, RL_BAD_MAT, (int)src); return; }
    const float tr = mats[t.materialIndex].transparency;
    if (tr > 0.0f || !(tr == tr)) atomicOr(&err->transparent, 1u);
}

// materials changed under an unchanged layout: re-validate every laid-out triangle's index, recompute "transparent"
__global__ void rl_check_materials(const DevTri* __restrict__ tris, int n, const rz_material* __restrict__ mats, int nMat, RelayoutErr* err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = tris[i].mat;
    if (m < 0 || m >= nMat) { report(err, RL_BAD_MAT, tris[i].src); return; }
    const float tr = mats[m].transparency;
    if (tr > 0.0f || !(tr == tr)) atomicOr(&err->transparent, 1u);
}

}  // namespace

// 0 = fine (*transparentOut set), positive RL_BAD_MAT (*detail = the caller's triangle index), negative hipError_t
int relayout_check_materials_device(const DevTri* tris, long long n, const rz_material* mats, int nMat, void* workspace, int* pinned,
                                    unsigned* transparentOut, int* detail, hipStream_t s) {
    RelayoutErr* err = static_cast<RelayoutErr*>(workspace);
    hipError_t e = hipMemsetAsync(err, 0, sizeof(RelayoutErr), s);
    if (e != hipSuccess) return -(int)e;
    if (n > 0) hipLaunchKernelGGL(rl_check_materials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tris, (int)n, mats, nMat, err);
    if ((e = hipGetLastError()) != hipSuccess) return -(int)e;
    if ((e = hipMemcpyAsync(pinned, err, sizeof(RelayoutErr), hipMemcpyDeviceToHost, s)) != hipSuccess) return -(int)e;
    if ((e = hipStreamSynchronize(s)) != hipSuccess) return -(int)e;
    if (transparentOut) *transparentOut = (unsigned)pinned[2];
    if (detail) *detail = pinned[1];
    return pinned[0];
}

// FS:411: the geometric normal of a triangle, normalize(cross(edge1, edge2)), is the same for every ray that hits it; the
// shader recomputes it per hit.  One pass over the laid-out triangles (either re-layout path) stores it -- the kernels'
// own cross / normalize on the very edges DevTri holds, so the bits are the ones trace_closest used to compute per hit.
__global__ void rl_tri_normals(const DevTri* __restrict__ tris, long long n, DevTriN* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevTri t = tris[i];
    const v3 ln = normalize(cross(mk3(t.e1x, t.e1y, t.e1z), mk3(t.e2x, t.e2y, t.e2z)));
    DevTriN o;
    o.n[0] = ln.x; o.n[1] = ln.y; o.n[2] = ln.z; o.mat = t.mat;
    out[i] = o;
}
int tri_normals_device(const DevTri* tris, long long n, DevTriN* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(rl_tri_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tris, n, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(int)e;
}

size_t relayout_workspace_bytes(size_t nNodes) {
    size_t scanTemp = 0;
    (void)rocprim::exclusive_scan(nullptr, scanTemp, (int*)nullptr, (int*)nullptr, 0, 2 * nNodes + 2, rocprim::plus<int>(), hipStream_t(nullptr));
    // two frontiers (nNodes ints each), flags + positions (2 nNodes + 2 each), scan temp, the error block
    return (2 * nNodes + 2 * (2 * nNodes + 2)) * sizeof(int) + scanTemp + 1024;
}

// One view.  nodes / idx / tris are the whole DEVICE arrays; pairs / trisOut the global device outputs.  hostRoot is
// the view's root node (the caller has it on the host), pinned = 4 pinned ints.  Returns 0, a positive RL_* code
// (the input is inconsistent: fall back to the host path for the message), or a negative hipError_t.
int relayout_view_device(const rz_bvh_node* nodes, long long nNodes, const int32_t* idx, long long nIdx, const rz_triangle* tris,
                         long long nTris, const rz_material* mats, int nMat, const rz_bvh_node& hostRoot, RelayoutView& V,
                         DevPair* pairs, long long pairCap, DevTri* trisOut, long long triCap, void* workspace, size_t workspaceBytes,
                         int* pinned, unsigned* transparentOut, hipStream_t s) {
#define RL_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return -(int)e_; } while (0)
    if (V.nodeOff < 0 || V.nodeOff >= nNodes || V.triOff < 0 || V.triOff > nIdx) return RL_BAD_CHILD;
    const long long nView = nNodes - V.nodeOff, nIdxView = nIdx - V.triOff;
    const rz_bvh_node* vn = nodes + V.nodeOff;
    char* w = static_cast<char*>(workspace);
    int* frontA = reinterpret_cast<int*>(w); w += (size_t)nNodes * sizeof(int);
    int* frontB = reinterpret_cast<int*>(w); w += (size_t)nNodes * sizeof(int);
    int* flags = reinterpret_cast<int*>(w); w += (size_t)(2 * nNodes + 2) * sizeof(int);
    int* pos = reinterpret_cast<int*>(w); w += (size_t)(2 * nNodes + 2) * sizeof(int);
    RelayoutErr* err = reinterpret_cast<RelayoutErr*>(w); w += 256;
    int* nextSize = reinterpret_cast<int*>(w); w += 256;
    void* scanTemp = w;
    size_t scanBytes = workspaceBytes - (size_t)(w - static_cast<char*>(workspace));
    RL_HIP(hipMemsetAsync(err, 0, sizeof(RelayoutErr), s));
    std::memcpy(V.rootMin, hostRoot.boundsMin, 12);
    std::memcpy(V.rootMax, hostRoot.boundsMax, 12);
    V.nPairs = 0; V.depth = 1; V.empty = 0; V.nSlots = 0;
    int maxSlotHost = 0;
    if (hostRoot.count >= 0) {              // the root is a leaf (count 0: an empty mesh, BVH.cpp:115-118)
        if (hostRoot.count > 15 || hostRoot.leftFirst < 0 || (long long)hostRoot.leftFirst + hostRoot.count > nIdxView) return RL_BAD_LEAF;
        V.empty = hostRoot.count == 0;
        V.rootEnc = ~((hostRoot.leftFirst << 4) | hostRoot.count);
        maxSlotHost = hostRoot.leftFirst + hostRoot.count;
    } else {
        V.rootEnc = 0;
        const int zero = 0;
        RL_HIP(hipMemcpyAsync(frontA, &zero, sizeof(int), hipMemcpyHostToDevice, s));
        int fsize = 1, levelBase = 0;
        int* cur = frontA; int* nxt = frontB;
        while (fsize > 0) {
            const int nextBase = levelBase + fsize;
            if ((long long)nextBase > nView || (long long)V.pairBase + nextBase > pairCap) return RL_NOT_A_TREE;
            const int blocks = (fsize + 255) / 256;
            hipLaunchKernelGGL(rl_flags, dim3(blocks), dim3(256), 0, s, vn, nView, cur, fsize, flags, err);
            size_t tb = scanBytes;
            RL_HIP(rocprim::exclusive_scan(scanTemp, tb, flags, pos, 0, (size_t)2 * fsize, rocprim::plus<int>(), s));
            hipLaunchKernelGGL(rl_write, dim3(blocks), dim3(256), 0, s, vn, nView, nIdxView, cur, fsize, flags, pos,
                               pairs + V.pairBase, levelBase, nextBase, nxt, nextSize, err);
            RL_HIP(hipMemcpyAsync(pinned, nextSize, sizeof(int), hipMemcpyDeviceToHost, s));
            RL_HIP(hipMemcpyAsync(pinned + 1, &err->code, sizeof(int), hipMemcpyDeviceToHost, s));
            RL_HIP(hipStreamSynchronize(s));
            if (pinned[1] != RL_OK) return pinned[1];
            V.depth += 1;
            levelBase = nextBase;
            fsize = pinned[0];
            int* t = cur; cur = nxt; nxt = t;
        }
        V.nPairs = levelBase;
        RL_HIP(hipMemcpyAsync(pinned, &err->maxSlot, sizeof(int), hipMemcpyDeviceToHost, s));
        RL_HIP(hipStreamSynchronize(s));
        maxSlotHost = pinned[0];
    }
    V.nSlots = maxSlotHost;
    if ((long long)V.triBase + V.nSlots > triCap) return RL_BAD_LEAF;
    if (V.nSlots > 0)
        hipLaunchKernelGGL(rl_gather, dim3((V.nSlots + 255) / 256), dim3(256), 0, s, tris, nTris, idx + V.triOff, V.gTriOff, V.nSlots,
                           trisOut + V.triBase, mats, nMat, err);
    RL_HIP(hipGetLastError());
    RL_HIP(hipMemcpyAsync(pinned, &err->code, sizeof(int), hipMemcpyDeviceToHost, s));
    RL_HIP(hipMemcpyAsync(pinned + 2, &err->transparent, sizeof(int), hipMemcpyDeviceToHost, s));
    RL_HIP(hipStreamSynchronize(s));
    if (pinned[0] != RL_OK) return pinned[0];
    if (transparentOut) *transparentOut |= (unsigned)pinned[2];
    return 0;
#undef RL_HIP
}

}  // namespace rz
