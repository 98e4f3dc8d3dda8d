// rz_context.hip -- the C-ABI of include/rayzen_hip.h: context, uploads, the
// one-time re-layout of RayZen's SSBO arrays into the device structures of
// rz_scene_dev.h, launches and read-back.  Host code, compiled by hipcc.
//
// Replaces, call for call, what RayZen/src/main.cpp does with OpenGL:
//   rz_upload     <- glGenBuffers+glBufferData+glBindBufferBase (main.cpp:1072-1119)
//   rz_update     <- glBufferSubData                            (main.cpp:1196-1207)
//   rz_set_frame  <- glUniform* in sendSceneDataToShader        (main.cpp:1356-1379)
//   rz_render     <- glDrawArrays(GL_TRIANGLE_FAN,0,4)          (main.cpp:637)
//   rz_sync       <- glFinish                                   (main.cpp:1347)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <tuple>
#include <limits>
#include <vector>

#include "rayzen_hip.h"
#include "rz_internal.h"
#include "rz_device_math.h"      // RZ_MATH_FLAVOUR (rz_math_flavour())


using namespace rz;

namespace {

thread_local std::string g_last_error = "";

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// One BLAS as seen from an instance: RayZen lets every instance name its own
// node / index / triangle offsets (include/BVH.h:14-21); distinct triples are
// laid out once each.
struct BlasView {
    int pairBase = 0, triBase = 0;
    int rootEnc = 0;
    float rootMin[3] = {0, 0, 0}, rootMax[3] = {0, 0, 0};
    int depth = 1;
    bool empty = false;
    bool mayGlass = true;       // a triangle of the view may use a transparent material (a scheduling hint for the kernels: DevInstance.flags bit 1)
};

constexpr int kNumBindings = 10;
size_t elem_size(int b) {
    switch (b) {
        case RZ_BIND_TRIANGLES: return sizeof(rz_triangle);
        case RZ_BIND_MATERIALS: return sizeof(rz_material);
        case RZ_BIND_LIGHTS: return sizeof(rz_light);
        case RZ_BIND_TLAS_NODES: return sizeof(rz_bvh_node);
        case RZ_BIND_TLAS_INDICES: return sizeof(int32_t);
        case RZ_BIND_BLAS_NODES: return sizeof(rz_bvh_node);
        case RZ_BIND_BLAS_INDICES: return sizeof(int32_t);
        case RZ_BIND_INSTANCES: return sizeof(rz_bvh_instance);
        default: return 0;
    }
}

}  // namespace

struct rz_ctx {
    int device = 0;
    unsigned flags = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    // a ring of event pairs: every render launch is bracketed on its stream, and the durations can be
    // collected later without synchronising inside a timed loop
    static constexpr int kRing = 64;
    hipEvent_t evStart[kRing] = {}, evStop[kRing] = {};
    int ringHead = 0;       // next slot to record into
    int ringCount = 0;      // launches recorded since the history was last drained (<= kRing)
    bool timed = false;
    int lastLaunches = 0;
    float hemi0[3] = {0.0f, 0.0f, 0.0f};   // KParams::hemi0, computed at rz_create
    long long triNValid = -1;       // triangles dTriN holds normals for (-1: none; reset with the geometry)
    long long lastGrid = 0;         // workgroups of the last render launch (RZ_PROF: how many wave-log entries are valid)
    rz_launch_plan lastPlan{};      // rz_debug_last_plan
    bool lastCompact = false;       // the last launch took compacting claims (rz_kernels.hip: render_claim_compact)
    bool lastGlobalPool = false;    // the last launch's waves kept their pools of parked paths across claims (rz_kernels.hip: pool_process)
    std::string err;

    // host copies of the caller's arrays (the re-layout needs them; rz_update patches them)
    std::vector<unsigned char> host[kNumBindings];
    bool present[kNumBindings] = {};
    bool geomDirty = true, instDirty = true, tlasDirty = true, matDirty = true, lightDirty = true;

    // device scene
    DevBuf dPairs, dTris, dInst, dTlasNodes, dTlasIdx, dMat, dLight, dCounters, dResolve, dGroupCtr, dBlasOvf;
    std::map<std::tuple<int, int, int>, BlasView> views;
    std::vector<DevPair> hPairs;
    std::vector<DevTri> hTris;
    int maxBlasDepth = 1, tlasDepth = 1;

    // frame
    bool haveFrame = false;
    rz_frame_params frame{};
    DevBuf ownAccum, dIor;
    // device-side dynamic update (rz_update_transforms)
    DevBuf dXforms, dInstRef, dTlasScratch, dProjBoxes, dBuildWs, dTlasDfs, dTriN;
    int nTlasDfs = 0;               // pop positions of the TLAS (rz_trace.h: trace_closest)
    int* tlasHostCounts = nullptr;      // pinned: node count, index count, depth
    bool deviceOwnsTlas = false;        // instances + TLAS on the device are newer than the host copies
    int devTlasNodes = 0;
    bool sceneHasTransparency = true;   // some triangle uses a material with transparency > 0
    const char* lastKernel = "";
    void* extAccum = nullptr;
    size_t extAccumBytes = 0;
    int failAllocCountdown = 0;         // rz_debug_fail_alloc (test hook)
    // device re-layout (rz_relayout.hip): the caller's raw arrays on the device, the fill of dPairs / dTris, scratch
    DevBuf dRawNodes, dRawIdx, dRawTris, dRelayoutWs, dClaimScratch;
    DevBuf dSnap;                                  // transparent scenes: the resident waves' sample prefixes (rz_path.h: snapshot_store)
    DevBuf dWavePools, dWaitMeta;     // the resident waves' pools of parked paths and the bookkeeping of their wait slots (rz_kernels.hip: pool_process; the slots themselves: behind the claim scratch in dClaimScratch)
    size_t lastScratchBytes = 0;                   // what the last compacting launch's waves had in scratch (pools + wait slots + bookkeeping)
    int* relayoutPinned = nullptr;
    bool layoutOnDevice = false;        // dPairs / dTris were produced on the device (hPairs / hTris are empty)
    long long devPairsUsed = 0, devTrisUsed = 0;
    bool matChangedSinceLayout = false; // materials were uploaded after the BLAS views were laid out: their per-view transparency hints are stale
    unsigned devTransparent = 0;       // device re-layout: bit 0 a transparent material is in use, bit 1 an irregular child box
    bool irregularBoxes = false;       // some BLAS child box has min > max or a NaN plane: the traversal keeps the generic slab test
    // rz_build_geometry: BLAS nodes / indices live in dRawNodes / dRawIdx; the host copies are fetched on demand
    bool geomOnDevice = false, geomHostFresh = false;
    size_t devNodes = 0, devIdx = 0;
    std::map<int, rz_bvh_node> devRoots;    // node offset of a mesh -> its root node
};

namespace {

int fail(rz_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try {
        if (c) c->err = buf;
        g_last_error = buf;
    } catch (...) { }     // the code still tells the caller what happened
    return code;
}

// No C++ exception crosses the C-ABI: every exported entry point that can reach a std::vector / std::map /
// std::string growth runs its body through guarded(), which turns std::bad_alloc into RZ_ERR_NO_MEMORY and anything
// else into RZ_ERR_HIP.  (rz_debug_fail_alloc arms a countdown that makes alloc_point() throw std::bad_alloc at the
// n-th host allocation site reached -- the test hook that proves the conversion.)
// It also makes the context's device the calling thread's current one: a context may be driven while another device is
// current (a group of several devices in one process, rz_group.hip: a member reached through rz_group_ctx), and every
// hipMalloc, kernel launch and event below would otherwise land on THAT device.
template <class F>
int guarded(rz_ctx* c, const char* what, F&& body) {
    try {
        if (c) {
            const hipError_t e = hipSetDevice(c->device);
            if (e != hipSuccess) return fail(c, RZ_ERR_HIP, "%s: hipSetDevice(%d): %s", what, c->device, hipGetErrorString(e));
        }
        return body();
    } catch (const std::bad_alloc&) {
        return fail(c, RZ_ERR_NO_MEMORY, "%s: out of host memory", what);
    } catch (const std::exception& e) {
        return fail(c, RZ_ERR_HIP, "%s: %s", what, e.what());
    } catch (...) {
        return fail(c, RZ_ERR_HIP, "%s: unknown C++ exception", what);
    }
}
void alloc_point(rz_ctx* c) {
    if (c->failAllocCountdown > 0 && --c->failAllocCountdown == 0) throw std::bad_alloc();
}

#define RZ_HIP(c, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) return fail((c), RZ_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int ensure(rz_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return RZ_OK;
    b.release();
    size_t want = std::max<size_t>(bytes, 256);
    RZ_HIP(c, hipMalloc(&b.p, want));
    b.cap = want;
    return RZ_OK;
}

// Scratch a launch can do without (the cross-claim pools: without them a claim works its parked paths off by itself, as in
// round 2): taken only if it leaves at least half of the device's free memory to the caller, and a refusal is not an error.
bool ensure_optional(DevBuf& b, size_t bytes) {
    if (bytes <= b.cap && b.p) return true;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (bytes > (freeB + b.cap) / 2) return false;
    b.release();
    if (hipMalloc(&b.p, std::max<size_t>(bytes, 256)) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; b.cap = 0; return false; }
    b.cap = std::max<size_t>(bytes, 256);
    return true;
}

int upload_vec(rz_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc != RZ_OK) return rc;
    if (bytes) RZ_HIP(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return RZ_OK;
}

template <class T> const T* hostArr(const rz_ctx* c, int b) { return reinterpret_cast<const T*>(c->host[b].data()); }
template <class T> size_t hostCount(const rz_ctx* c, int b) { return c->host[b].size() / sizeof(T); }
size_t blasNodeCount(const rz_ctx* c) { return c->geomOnDevice ? c->devNodes : hostCount<rz_bvh_node>(c, RZ_BIND_BLAS_NODES); }
size_t blasIdxCount(const rz_ctx* c) { return c->geomOnDevice ? c->devIdx : hostCount<int32_t>(c, RZ_BIND_BLAS_INDICES); }

int sync_geom_host(rz_ctx* c);

// Lay out one BLAS: breadth-first walk from its root, one DevPair per internal
// node (so the hot top levels are contiguous), triangles gathered to leaf order.
int build_view(rz_ctx* c, int nodeOff, int triOff, int gTriOff, BlasView& V) {
    { int rc = sync_geom_host(c); if (rc != RZ_OK) return rc; }
    const rz_bvh_node* nodes = hostArr<rz_bvh_node>(c, RZ_BIND_BLAS_NODES);
    const int32_t* idx = hostArr<int32_t>(c, RZ_BIND_BLAS_INDICES);
    const rz_triangle* tris = hostArr<rz_triangle>(c, RZ_BIND_TRIANGLES);
    const long long nNodes = (long long)hostCount<rz_bvh_node>(c, RZ_BIND_BLAS_NODES);
    const long long nIdx = (long long)hostCount<int32_t>(c, RZ_BIND_BLAS_INDICES);
    const long long nTris = (long long)hostCount<rz_triangle>(c, RZ_BIND_TRIANGLES);
    if (nodeOff < 0 || nodeOff >= nNodes) return fail(c, RZ_ERR_BAD_SCENE, "instance blasNodeOffset %d outside the BLAS node array (%lld)", nodeOff, nNodes);
    V.pairBase = (int)c->hPairs.size();
    V.triBase = (int)c->hTris.size();
    const rz_bvh_node& root = nodes[nodeOff];
    std::memcpy(V.rootMin, root.boundsMin, 12);
    std::memcpy(V.rootMax, root.boundsMax, 12);

    // slot s of this view's leaf-ordered triangle range <-> blasTriIndices[triOff + s]
    auto leafEnc = [&](const rz_bvh_node& n, int& enc) -> int {
        if (n.count > 15) return fail(c, RZ_ERR_BAD_SCENE, "BLAS leaf with %d triangles (max 15)", n.count);
        if (n.leftFirst < 0 || (long long)triOff + n.leftFirst + n.count > nIdx || triOff < 0)
            return fail(c, RZ_ERR_BAD_SCENE, "BLAS leaf range [%d,+%d) outside the index array", n.leftFirst, n.count);
        enc = ~((n.leftFirst << 4) | n.count);
        return RZ_OK;
    };
    int maxSlot = 0;
    V.depth = 1;
    if (root.count >= 0) {      // the root is a leaf (count 0: empty mesh, BVH.cpp:115-118)
        V.empty = (root.count == 0);
        int rc = leafEnc(root, V.rootEnc);
        if (rc != RZ_OK) return rc;
        maxSlot = root.leftFirst + root.count;
    } else {
        struct Item { int node; int depth; };
        std::vector<Item> queue;
        queue.push_back({0, 1});
        V.rootEnc = 0;          // the root's pair is pair 0 of the view
        size_t head = 0;
        // pair index of an internal node = its rank among internal nodes in BFS order
        while (head < queue.size()) {
            Item it = queue[head++];
            const rz_bvh_node& n = nodes[nodeOff + it.node];
            const int L = n.leftFirst, R = n.leftFirst + 1;
            if (L < 1 || (long long)nodeOff + R >= nNodes)
                return fail(c, RZ_ERR_BAD_SCENE, "BLAS node %d has children %d,%d outside the node array", it.node, L, R);
            if (queue.size() > (size_t)nNodes) return fail(c, RZ_ERR_BAD_SCENE, "BLAS at node offset %d is not a tree", nodeOff);
            DevPair P{};
            const rz_bvh_node& ln = nodes[nodeOff + L];
            const rz_bvh_node& rn = nodes[nodeOff + R];
            P.lx[0] = ln.boundsMin[0]; P.lx[1] = ln.boundsMax[0]; P.ly[0] = ln.boundsMin[1]; P.ly[1] = ln.boundsMax[1];
            P.lz[0] = ln.boundsMin[2]; P.lz[1] = ln.boundsMax[2];
            P.rx[0] = rn.boundsMin[0]; P.rx[1] = rn.boundsMax[0]; P.ry[0] = rn.boundsMin[1]; P.ry[1] = rn.boundsMax[1];
            P.rz[0] = rn.boundsMin[2]; P.rz[1] = rn.boundsMax[2];
            // (an inverted or NaN child box: the octant-specialised slab test is only the shader's test for regular boxes)
            if (!(ln.boundsMin[0] <= ln.boundsMax[0] && ln.boundsMin[1] <= ln.boundsMax[1] && ln.boundsMin[2] <= ln.boundsMax[2] &&
                  rn.boundsMin[0] <= rn.boundsMax[0] && rn.boundsMin[1] <= rn.boundsMax[1] && rn.boundsMin[2] <= rn.boundsMax[2]))
                c->irregularBoxes = true;
            V.depth = std::max(V.depth, it.depth + 1);
            const rz_bvh_node* ch[2] = {&ln, &rn};
            int32_t* encs[2] = {&P.lenc, &P.renc};
            const int chIdx[2] = {L, R};
            for (int k = 0; k < 2; ++k) {
                if (ch[k]->count >= 0) {
                    int e; int rc = leafEnc(*ch[k], e);
                    if (rc != RZ_OK) return rc;
                    *encs[k] = e;
                    maxSlot = std::max(maxSlot, ch[k]->leftFirst + ch[k]->count);
                } else {
                    *encs[k] = (int)queue.size();      // BFS rank of this internal child == its pair index
                    queue.push_back({chIdx[k], it.depth + 1});
                }
            }
            alloc_point(c);
            c->hPairs.push_back(P);
        }
        // queue[i] is the i-th internal node in BFS order and its pair was pushed i-th: enc == i holds by construction
    }
    // gather triangles into leaf order
    bool viewGlass = false;
    for (int s = 0; s < maxSlot; ++s) {
        const long long src = (long long)gTriOff + idx[triOff + s];
        if (src < 0 || src >= nTris) return fail(c, RZ_ERR_BAD_SCENE, "BLAS index %d -> triangle %lld outside the triangle array (%lld)", s, src, nTris);
        const rz_triangle& t = tris[src];
        DevTri d{};
        d.v0[0] = t.v0[0]; d.v0[1] = t.v0[1]; d.v0[2] = t.v0[2];
        d.e1x = t.v1[0] - t.v0[0]; d.e1y = t.v1[1] - t.v0[1]; d.e1z = t.v1[2] - t.v0[2];   // FS:392
        d.e2x = t.v2[0] - t.v0[0]; d.e2y = t.v2[1] - t.v0[1]; d.e2z = t.v2[2] - t.v0[2];   // FS:393
        d.mat = t.materialIndex;
        {   // (hint only: an index the material check will reject later counts as "may be transparent")
            const int nMat_ = (int)hostCount<rz_material>(c, RZ_BIND_MATERIALS);
            const rz_material* mats_ = hostArr<rz_material>(c, RZ_BIND_MATERIALS);
            if (t.materialIndex < 0 || t.materialIndex >= nMat_ || !(mats_[t.materialIndex].transparency <= 0.0f)) viewGlass = true;
        }
        d.src = (int32_t)src;
        if ((s & 1023) == 0) alloc_point(c);
        c->hTris.push_back(d);
    }
    V.mayGlass = viewGlass;
    return RZ_OK;
}

// After rz_update_transforms the device holds newer instances / TLAS than the host copies: bring them back
// (3.4 KB at 16 instances) before anything reads or patches those copies.
// After rz_build_geometry the BLAS nodes / indices exist only on the device: bring them to the host copies before
// anything reads or patches those (rz_read_binding, rz_update, the host re-layout, the wireframe's path walk).
int sync_geom_host(rz_ctx* c) {
    if (!c->geomOnDevice || c->geomHostFresh) return RZ_OK;
    alloc_point(c);
    c->host[RZ_BIND_BLAS_NODES].resize(c->devNodes * sizeof(rz_bvh_node));
    c->host[RZ_BIND_BLAS_INDICES].resize(c->devIdx * sizeof(int32_t));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    if (c->devNodes) RZ_HIP(c, hipMemcpy(c->host[RZ_BIND_BLAS_NODES].data(), c->dRawNodes.p, c->devNodes * sizeof(rz_bvh_node), hipMemcpyDeviceToHost));
    if (c->devIdx) RZ_HIP(c, hipMemcpy(c->host[RZ_BIND_BLAS_INDICES].data(), c->dRawIdx.p, c->devIdx * sizeof(int32_t), hipMemcpyDeviceToHost));
    c->geomHostFresh = true;
    return RZ_OK;
}

int sync_host_from_device(rz_ctx* c) {
    if (!c->deviceOwnsTlas) return RZ_OK;
    const size_t nInst = hostCount<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    const int nn = c->tlasHostCounts[0], ni = c->tlasHostCounts[1];
    c->host[RZ_BIND_TLAS_NODES].resize((size_t)nn * sizeof(rz_bvh_node));
    c->host[RZ_BIND_TLAS_INDICES].resize((size_t)ni * sizeof(int32_t));
    if (nInst) RZ_HIP(c, hipMemcpy(c->host[RZ_BIND_INSTANCES].data(), c->dInstRef.p, nInst * sizeof(rz_bvh_instance), hipMemcpyDeviceToHost));
    if (nn) RZ_HIP(c, hipMemcpy(c->host[RZ_BIND_TLAS_NODES].data(), c->dTlasNodes.p, (size_t)nn * sizeof(rz_bvh_node), hipMemcpyDeviceToHost));
    if (ni) RZ_HIP(c, hipMemcpy(c->host[RZ_BIND_TLAS_INDICES].data(), c->dTlasIdx.p, (size_t)ni * sizeof(int32_t), hipMemcpyDeviceToHost));
    c->deviceOwnsTlas = false;
    return RZ_OK;
}

// The TLAS as the list of its nodes in the order FS:464-501 pops them (right child first), with the position that follows
// each subtree (TlasDfs, rz_scene_dev.h).  `below` is the number of stack entries under a node when it is popped: the
// shader's stack[64] holds no more, and the oracle drops a push that would not fit -- such a node never expands (count 0).
// Call after tlas_depth() has accepted the array (children in range, no more pops than nodes).
void tlas_pop_order(const rz_bvh_node* n, size_t count, const int32_t* idx, std::vector<TlasDfs>& out) {
    out.clear();
    if (count == 0) return;
    struct Item { int node, below, parentPos; };
    std::vector<Item> st;
    std::vector<int> parent;
    st.push_back({0, 0, -1});
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        const rz_bvh_node& N = n[it.node];
        TlasDfs R{};
        std::memcpy(R.bmin, N.boundsMin, 12);
        std::memcpy(R.bmax, N.boundsMax, 12);
        R.first = N.leftFirst;
        R.count = N.count > 0 ? N.count : (N.count < 0 && it.below + 2 <= 64 ? -1 : 0);
        R.inst0 = N.count > 0 ? idx[N.leftFirst] : 0;
        const int pos = (int)out.size();
        out.push_back(R);
        parent.push_back(it.parentPos);
        if (N.count < 0) {                                          // (also under a node that never expands: the list is complete, like the device builder's)
            st.push_back({N.leftFirst, it.below, pos});             // popped after the right subtree, with the same entries below it
            st.push_back({N.leftFirst + 1, it.below + 1, pos});     // popped next, the left child below it
        }
    }
    std::vector<int> size(out.size(), 1);
    for (size_t p = out.size(); p-- > 1;) size[parent[p]] += size[p];
    for (size_t p = 0; p < out.size(); ++p) out[p].skip = (int)p + size[p];
}

int tlas_depth(const rz_bvh_node* n, size_t count) {
    if (count == 0) return 0;
    int best = 1;
    std::vector<std::pair<int, int>> st;
    st.push_back({0, 1});
    size_t visited = 0;
    while (!st.empty()) {
        auto [i, d] = st.back();
        st.pop_back();
        if (++visited > count) return -1;
        best = std::max(best, d);
        if (n[i].count <= 0) {   // the shader treats count <= 0 as internal (FS:470)
            if (n[i].count == 0) continue;     // empty root written by the host builder
            int L = n[i].leftFirst;
            if (L < 1 || (size_t)L + 1 >= count) return -1;
            st.push_back({L, d + 1});
            st.push_back({L + 1, d + 1});
        }
    }
    return best;
}

// Device re-layout of one view (rz_relayout.hip).  RZ_OK: V filled; 1: the arrays are inconsistent or something did
// not fit -- run the host re-layout, which words the error; other negatives: a HIP failure.
bool host_relayout_forced(const rz_ctx* c) {
    if (c->flags & RZ_FLAG_HOST_RELAYOUT) return true;
    const char* e = std::getenv("RZ_HOST_RELAYOUT");
    return e && *e && *e != '0';
}

int prepare_device_relayout(rz_ctx* c) {       // raw arrays + materials on the device, output buffers sized, scratch
    const size_t nNodes = blasNodeCount(c), nIdx = blasIdxCount(c);
    int rc;
    if (!c->geomOnDevice) {                    // (rz_build_geometry left nodes and indices there already)
        rc = upload_vec(c, c->dRawNodes, c->host[RZ_BIND_BLAS_NODES].data(), c->host[RZ_BIND_BLAS_NODES].size());
        if (rc != RZ_OK) return rc;
        rc = upload_vec(c, c->dRawIdx, c->host[RZ_BIND_BLAS_INDICES].data(), c->host[RZ_BIND_BLAS_INDICES].size());
        if (rc != RZ_OK) return rc;
    }
    rc = upload_vec(c, c->dRawTris, c->host[RZ_BIND_TRIANGLES].data(), c->host[RZ_BIND_TRIANGLES].size());
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dPairs, std::max<size_t>(nNodes, 1) * sizeof(DevPair));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dTris, std::max<size_t>(nIdx, 1) * sizeof(DevTri));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dRelayoutWs, relayout_workspace_bytes(nNodes));
    if (rc != RZ_OK) return rc;
    if (!c->relayoutPinned) RZ_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->relayoutPinned), 64, hipHostMallocDefault));
    return RZ_OK;
}

int device_view(rz_ctx* c, int nodeOff, int triOff, int gTriOff, BlasView& V) {
    const long long nNodes = (long long)blasNodeCount(c);
    const long long nIdx = (long long)blasIdxCount(c);
    const long long nTris = (long long)hostCount<rz_triangle>(c, RZ_BIND_TRIANGLES);
    if (nodeOff < 0 || nodeOff >= nNodes) return 1;
    rz_bvh_node root;
    if (c->geomOnDevice) {
        auto it = c->devRoots.find(nodeOff);
        if (it == c->devRoots.end()) return 1;      // an instance that does not start at a mesh's root: let the host path look at it
        root = it->second;
    } else {
        root = hostArr<rz_bvh_node>(c, RZ_BIND_BLAS_NODES)[nodeOff];
    }
    RelayoutView R{};
    unsigned viewFlags = 0;         // bit 0: a triangle of this view uses a transparent material, bit 1: an irregular child box
    R.nodeOff = nodeOff; R.triOff = triOff; R.gTriOff = gTriOff;
    R.pairBase = (int)c->devPairsUsed; R.triBase = (int)c->devTrisUsed;
    const int rc = relayout_view_device(static_cast<const rz_bvh_node*>(c->dRawNodes.p), nNodes, static_cast<const int32_t*>(c->dRawIdx.p), nIdx,
                                        static_cast<const rz_triangle*>(c->dRawTris.p), nTris, static_cast<const rz_material*>(c->dMat.p),
                                        (int)hostCount<rz_material>(c, RZ_BIND_MATERIALS), root, R,
                                        static_cast<DevPair*>(c->dPairs.p), (long long)(c->dPairs.cap / sizeof(DevPair)),
                                        static_cast<DevTri*>(c->dTris.p), (long long)(c->dTris.cap / sizeof(DevTri)), c->dRelayoutWs.p,
                                        c->dRelayoutWs.cap, c->relayoutPinned, &viewFlags, c->stream);
    if (rc < 0) return fail(c, RZ_ERR_HIP, "device re-layout: %s", hipGetErrorString((hipError_t)(-rc)));
    if (rc > 0) return 1;
    c->devTransparent |= viewFlags;
    V.mayGlass = (viewFlags & 1u) != 0;
    V.pairBase = R.pairBase; V.triBase = R.triBase; V.rootEnc = R.rootEnc; V.depth = R.depth; V.empty = R.empty != 0;
    std::memcpy(V.rootMin, R.rootMin, 12); std::memcpy(V.rootMax, R.rootMax, 12);
    c->devPairsUsed += R.nPairs; c->devTrisUsed += R.nSlots;
    return RZ_OK;
}

int finalize_body(rz_ctx* c) {
    if (!(c->geomDirty || c->instDirty || c->tlasDirty || c->matDirty || c->lightDirty)) return RZ_OK;
    for (int b : {RZ_BIND_TRIANGLES, RZ_BIND_MATERIALS, RZ_BIND_LIGHTS, RZ_BIND_TLAS_NODES, RZ_BIND_TLAS_INDICES,
                  RZ_BIND_BLAS_NODES, RZ_BIND_BLAS_INDICES, RZ_BIND_INSTANCES})
        if (!c->present[b]) return fail(c, RZ_ERR_NOT_READY, "binding %d has not been uploaded", b);

    // (the per-view "may hold transparent triangles" hints were taken from the materials of the moment the views were laid out)
    if (c->geomDirty) c->matChangedSinceLayout = false;
    else if (c->matDirty && !c->matChangedSinceLayout) { c->matChangedSinceLayout = true; c->instDirty = true; }
    // materials first: the device re-layout checks triangle material indices against them
    if (c->matDirty) {
        int rc = upload_vec(c, c->dMat, c->host[RZ_BIND_MATERIALS].data(), c->host[RZ_BIND_MATERIALS].size());
        if (rc != RZ_OK) return rc;
    }
    if (c->geomDirty) {
        // the traversal addresses a BLAS's pairs with a 32-bit byte offset (rz_trace.h: sload16_off): < 2^26 pairs per BLAS
        const size_t nodesNow = c->geomOnDevice ? c->devNodes : hostCount<rz_bvh_node>(c, RZ_BIND_BLAS_NODES);
        if (nodesNow >= ((size_t)1 << 27))
            return fail(c, RZ_ERR_BAD_SCENE, "BLAS node array holds %zu nodes; the limit is %zu", nodesNow, ((size_t)1 << 27) - 1);
        c->views.clear(); c->hPairs.clear(); c->hTris.clear(); c->instDirty = true;
        c->triNValid = -1;
        c->devPairsUsed = c->devTrisUsed = 0; c->devTransparent = 0; c->irregularBoxes = false;
        c->layoutOnDevice = !host_relayout_forced(c);
        if (c->layoutOnDevice) { int rc = prepare_device_relayout(c); if (rc != RZ_OK) return rc; }
    }
    if (c->instDirty) {
        const rz_bvh_instance* inst = hostArr<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
        const size_t nInst = hostCount<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
        alloc_point(c);
        std::vector<DevInstance> dev(nInst);
        // Pass 1: lay out every BLAS view the instances name that is not laid out yet -- on the device
        // (rz_relayout.hip), or, if that finds the arrays inconsistent (or is switched off), on the host, which
        // also words the error.
        for (int attempt = 0; attempt < 2; ++attempt) {
            bool redo = false, grew = false;
            for (size_t i = 0; i < nInst && !redo; ++i) {
                auto key = std::make_tuple(inst[i].blasNodeOffset, inst[i].blasTriOffset, inst[i].globalTriOffset);
                if (c->views.find(key) != c->views.end()) continue;
                BlasView V;
                int rc;
                if (c->layoutOnDevice) {
                    rc = device_view(c, inst[i].blasNodeOffset, inst[i].blasTriOffset, inst[i].globalTriOffset, V);
                    if (rc == 1) {          // start over on the host
                        c->layoutOnDevice = false;
                        c->views.clear(); c->hPairs.clear(); c->hTris.clear(); c->triNValid = -1;
                        redo = true;
                        break;
                    }
                } else {
                    rc = build_view(c, inst[i].blasNodeOffset, inst[i].blasTriOffset, inst[i].globalTriOffset, V);
                }
                if (rc != RZ_OK) { c->views.clear(); c->hPairs.clear(); c->hTris.clear(); c->triNValid = -1; c->geomDirty = true; return rc; }
                c->views.emplace(key, V);
                grew = true;
            }
            if (redo) continue;
            if (!c->layoutOnDevice && (grew || c->geomDirty)) {
                int rc = upload_vec(c, c->dPairs, c->hPairs.data(), c->hPairs.size() * sizeof(DevPair));
                if (rc != RZ_OK) return rc;
                rc = upload_vec(c, c->dTris, c->hTris.data(), c->hTris.size() * sizeof(DevTri));
                if (rc != RZ_OK) return rc;
            }
            break;
        }
        {   // per-triangle geometric normals of everything laid out (rz_relayout.hip: rl_tri_normals)
            const long long nLaid = c->layoutOnDevice ? (long long)c->devTrisUsed : (long long)c->hTris.size();
            if (nLaid != c->triNValid) {        // (views are only ever appended between two geometry uploads)
                int rc = ensure(c, c->dTriN, (size_t)std::max<long long>(nLaid, 1) * sizeof(DevTriN));
                if (rc != RZ_OK) return rc;
                const int nrc = tri_normals_device(static_cast<const DevTri*>(c->dTris.p), nLaid, static_cast<DevTriN*>(c->dTriN.p), c->stream);
                if (nrc != 0) return fail(c, RZ_ERR_HIP, "triangle normals: %s", hipGetErrorString((hipError_t)(-nrc)));
                c->triNValid = nLaid;
            }
        }
        c->maxBlasDepth = 1;
        for (size_t i = 0; i < nInst; ++i) {
            const BlasView& V = c->views.at(std::make_tuple(inst[i].blasNodeOffset, inst[i].blasTriOffset, inst[i].globalTriOffset));
            DevInstance& D = dev[i];
            std::memset(&D, 0, sizeof D);
            for (int col = 0; col < 4; ++col)
                for (int row = 0; row < 3; ++row) {
                    D.inv[col * 3 + row] = inst[i].inverseTransform[col * 4 + row];
                    D.fwd[col * 3 + row] = inst[i].transform[col * 4 + row];
                }
            std::memcpy(D.rootMin, V.rootMin, 12);
            std::memcpy(D.rootMax, V.rootMax, 12);
            D.rootEnc = V.rootEnc;
            D.pairBase = V.pairBase;
            D.triBase = V.triBase;
            // bit 0: never hit (empty BLAS).  bit 1: the view may hold transparent triangles -- a hint for the compacting claims
            // (which rays not to park: rz_kernels.hip); when only the materials have changed since the views were laid out
            // nobody knows any more, and every instance says "may"
            D.flags = (V.empty ? 1 : 0) | ((V.mayGlass || c->matChangedSinceLayout) ? 2 : 0);
        }
        for (auto& kv : c->views) c->maxBlasDepth = std::max(c->maxBlasDepth, kv.second.depth);
        int rc = upload_vec(c, c->dInst, dev.data(), dev.size() * sizeof(DevInstance));
        if (rc != RZ_OK) return rc;
        // the staging vector dies at scope exit: the copy must have left it
        RZ_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (c->tlasDirty || c->instDirty) {
        const rz_bvh_node* tn = hostArr<rz_bvh_node>(c, RZ_BIND_TLAS_NODES);
        const size_t nTn = hostCount<rz_bvh_node>(c, RZ_BIND_TLAS_NODES);
        const int32_t* ti = hostArr<int32_t>(c, RZ_BIND_TLAS_INDICES);
        const size_t nTi = hostCount<int32_t>(c, RZ_BIND_TLAS_INDICES);
        const size_t nInst = hostCount<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
        int d = tlas_depth(tn, nTn);
        if (d < 0) return fail(c, RZ_ERR_BAD_SCENE, "TLAS node array is not a tree");
        for (size_t i = 0; i < nTn; ++i)
            if (tn[i].count > 0) {
                if (tn[i].leftFirst < 0 || (size_t)tn[i].leftFirst + (size_t)tn[i].count > nTi)
                    return fail(c, RZ_ERR_BAD_SCENE, "TLAS leaf %zu range outside the TLAS index array", i);
                for (int k = 0; k < tn[i].count; ++k) {
                    int ii = ti[tn[i].leftFirst + k];
                    if (ii < 0 || (size_t)ii >= nInst)
                        return fail(c, RZ_ERR_BAD_SCENE, "TLAS index %d names instance %d of %zu", tn[i].leftFirst + k, ii, nInst);
                }
            }
        c->tlasDepth = std::max(d, 1);
        int rc = upload_vec(c, c->dTlasNodes, tn, nTn * sizeof(rz_bvh_node));
        if (rc != RZ_OK) return rc;
        rc = upload_vec(c, c->dTlasIdx, ti, nTi * sizeof(int32_t));
        if (rc != RZ_OK) return rc;
        std::vector<TlasDfs> dfs;
        tlas_pop_order(tn, nTn, ti, dfs);
        c->nTlasDfs = (int)dfs.size();
        rc = upload_vec(c, c->dTlasDfs, dfs.data(), dfs.size() * sizeof(TlasDfs));
        if (rc != RZ_OK) return rc;
        RZ_HIP(c, hipStreamSynchronize(c->stream));     // the staging vector dies at scope exit
    }
    if (c->lightDirty) {
        int rc = upload_vec(c, c->dLight, c->host[RZ_BIND_LIGHTS].data(), c->host[RZ_BIND_LIGHTS].size());
        if (rc != RZ_OK) return rc;
    }
    // material indices are data the kernels index with: check them once
    if (c->geomDirty || c->matDirty || c->instDirty) {
        const int nMat = (int)hostCount<rz_material>(c, RZ_BIND_MATERIALS);
        const rz_material* mats = hostArr<rz_material>(c, RZ_BIND_MATERIALS);
        bool transparent = false;
        if (c->layoutOnDevice) {
            // the gather of rz_relayout.hip checked the triangles it laid out against the materials uploaded above; when
            // only the materials changed, every laid-out triangle is checked again on the device
            if (c->matDirty && !c->geomDirty) {
                unsigned tr = 0; int detail = 0;
                const int rc = relayout_check_materials_device(static_cast<const DevTri*>(c->dTris.p), c->devTrisUsed, static_cast<const rz_material*>(c->dMat.p),
                                                               nMat, c->dRelayoutWs.p, c->relayoutPinned, &tr, &detail, c->stream);
                if (rc < 0) return fail(c, RZ_ERR_HIP, "material check: %s", hipGetErrorString((hipError_t)(-rc)));
                if (rc > 0) return fail(c, RZ_ERR_BAD_SCENE, "triangle %d has a materialIndex outside the %d materials uploaded", detail, nMat);
                c->devTransparent = (c->devTransparent & 2u) | (tr & 1u);
            }
            transparent = (c->devTransparent & 1u) != 0;
            c->irregularBoxes = (c->devTransparent & 2u) != 0;
        } else {
            for (const DevTri& t : c->hTris) {
                if (t.mat < 0 || t.mat >= nMat)
                    return fail(c, RZ_ERR_BAD_SCENE, "triangle %d has materialIndex %d, %d materials uploaded", t.src, t.mat, nMat);
                transparent = transparent || (mats[t.mat].transparency > 0.0f) || !(mats[t.mat].transparency == mats[t.mat].transparency);
            }
        }
        c->sceneHasTransparency = transparent;
    }
    // uploads read the caller-visible host copies: let them land before rz_update may patch those
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    c->geomDirty = c->instDirty = c->tlasDirty = c->matDirty = c->lightDirty = false;
    return RZ_OK;
}

// A failure half-way through the re-layout (an exception from a growing vector) must not leave a half-built scene
// behind: drop the derived state, keep the caller's arrays, and let the next call redo it.
int finalize(rz_ctx* c) {
    try {
        return finalize_body(c);
    } catch (...) {
        c->views.clear(); c->hPairs.clear(); c->hTris.clear(); c->triNValid = -1;
        c->geomDirty = true;
        throw;
    }
}

#ifndef RZ_SPREAD_MIN_INSTANCES
#define RZ_SPREAD_MIN_INSTANCES 3
#endif
#ifndef RZ_WPOOL_CHUNK
#define RZ_WPOOL_CHUNK 256         // paths a wave collects across claims before it traces them together
#endif
#ifndef RZ_CLAIM_STRIDE_PAD
#define RZ_CLAIM_STRIDE_PAD 0      // extra dwords between the scratch regions of neighbouring resident waves
#endif

// One lane per sample: independent samples when no triangle is transparent, speculated currentIor otherwise
// (rz_kernels.hip).  The one-lane-per-pixel kernel remains as RZ_FLAG_MEGAKERNEL (the literal, sequential form).
bool use_samples(const rz_ctx* c) { return (c->flags & RZ_FLAG_MEGAKERNEL) == 0; }

// The claim counter (word 0, zeroed per launch) and the backstop word (word RZ_ERRWORD, zeroed when made and when reported).
int ensure_group_counter(rz_ctx* c) {
    if (c->dGroupCtr.p) return RZ_OK;
    // (+ a row of 64 zeros at byte 512, read by every wave's ordered sums in place of a dark unit's light rows: rz_kernels.hip, zero_row)
    int rc = ensure(c, c->dGroupCtr, 1024);
    if (rc != RZ_OK) return rc;
    RZ_HIP(c, hipMemsetAsync(c->dGroupCtr.p, 0, 1024, c->stream));
    return RZ_OK;
}

int render_samples(rz_ctx* c, KParams K, bool counted, int evSlot) {
    K.nSlots = K.nLocalTiles * 64;
    int rc = ensure_group_counter(c);
    if (rc != RZ_OK) return rc;
    K.groupCounter = static_cast<unsigned*>(c->dGroupCtr.p);
    // LDS budget: the BLAS stack's LDS window is cut to what keeps the target number of waves on a CU (16 for the
    // opaque variant = its VGPR limit, and for the transparent one too: 5.4 KB of versions leave it a 9-entry window --
    // measured 40.2 -> 34.3 ms on the glass+mirror scene against 12 waves with the whole stack in LDS); deeper entries go to global overflow columns,
    // which are indexed by resident workgroup and therefore only exist for persistent launches.
    const SamplesPlan plan = plan_render_samples(K.spp, K.nSlots, c->sceneHasTransparency);
    const int need = K.blasStackCap;
#ifndef RZ_GLASS_WAVES_PER_CU
#define RZ_GLASS_WAVES_PER_CU 16
#endif
#ifndef RZ_OPAQUE_WAVES_PER_CU
#define RZ_OPAQUE_WAVES_PER_CU 16
#endif
    const size_t budget = (size_t)160 * 1024 / (c->sceneHasTransparency ? RZ_GLASS_WAVES_PER_CU : RZ_OPAQUE_WAVES_PER_CU);
    const size_t fixed = samples_lds_extra(c->sceneHasTransparency, plan.compact) + (size_t)K.tlasStackCap * 256;
    int window = budget > fixed ? (int)((budget - fixed) / 512) : 0;
    if (const char* e = std::getenv("RZ_BLAS_STACK_WINDOW")) window = std::atoi(e);        // test aid: force a small window
    window = std::max(window, 2);
    K.blasOvfCap = 0; K.blasOvf = nullptr;
    if (plan.perClaim > 0 && need > window) {
        K.blasStackCap = window;
        K.blasOvfCap = need - window;
        rc = ensure(c, c->dBlasOvf, (size_t)plan.grid * K.blasOvfCap * 64 * sizeof(uint2));
        if (rc != RZ_OK) return rc;
        K.blasOvf = static_cast<uint2*>(c->dBlasOvf.p);
    }
    // Compacting launches: every resident wave's scratch (rz_kernels.hip: WAIT SLOTS, pool_process) --
    //   * its pool of parked paths: room for the chunk it collects before it traces them + the most one more claim can park;
    //   * its claim scratch (the addends of the claim it is running, 1.5 KB per unit) and behind it its wait slots, one waiting
    //     group's addends (batches x 1.5 KB) each: twice the groups of a claim by default (a claim that
    //     finds fewer free ones than it has groups makes the wave trace its pool first; RZ_WAIT_SLOTS overrides);
    //   * 2 ints of bookkeeping per slot.
    // C2: 80 + 12 + 24 KB per wave, 470 MiB for the grid (round 3: 3.7 GB, of which 3.2 GB an array of 1.5 KB per unit of the launch).
    // The scratch is optional: a launch that cannot have it (or is told so: RZ_DEBUG_NO_POOL_MEMORY=1, a test aid) runs the
    // plain persistent loop instead, same image.
    K.wslots = nullptr; K.wslotStride = 0; K.slotFloats = 0; K.nWaitSlots = 0; K.wmeta = nullptr; K.drainEachClaim = 0; K.claimUnits = 0; K.claimScratchFloats = 0;
    K.wpool = nullptr; K.wpoolStride = 0; K.wpoolChunk = 0;
    if (plan.compact && K.maxBounces < 32768) {     // (a pool entry keeps its path's bounce in 15 bits: rz_kernels.hip, BACK)
        long long chunk = RZ_WPOOL_CHUNK;
        if (const char* e = std::getenv("RZ_WPOOL_CHUNK")) chunk = std::max<long long>(1, std::atoll(e));      // tuning / test aid
        const int nBatches = (K.spp + 63) / 64;
        const int groupsPerClaim = std::max(1, plan.perClaim);
        // (twice a claim's groups: measured on the final build against 24 and 32 slots, C2 / C3 / C4 / C5 / glass all within
        //  0.4 % -- profiles/r04_wait_slots/slots_chunk_final.log -- and 24 KB per wave less to keep: C2's scratch 553 -> 470 MiB)
        int nSlots = std::max(2 * groupsPerClaim, 16 / nBatches);
        if (const char* e = std::getenv("RZ_WAIT_SLOTS")) nSlots = std::max(2 * groupsPerClaim, std::atoi(e));  // tuning / test aid
        nSlots = std::min(nSlots, 64);
        // a pool's capacity: what a wave may hold when it starts a pass -- fewer than `chunk` paths plus a whole claim's -- and what a
        // pass can ADD to that: nothing in an opaque scene (a path survives in place or ends); in a transparent one every sample on the
        // wave's late list (at most RZ_GLATE_CAP, listed in this pass or earlier ones) can be released into the pool behind the survivors
        const size_t stride = (size_t)chunk + (size_t)plan.claimUnits * 64 + 64 + (c->sceneHasTransparency ? (size_t)RZ_GLATE_CAP : 0);
        // (a group's addends; transparent scenes: + one row, the currentIor each pixel ends with)
        const size_t slotFloats = (size_t)nBatches * 384 + (c->sceneHasTransparency ? 64 : 0);
        const size_t claimScratchFloats = (size_t)groupsPerClaim * slotFloats;
        bool snapOn = true;                 // transparent scenes resolve currentIor from the samples' snapshots: no snapshots, no claims
        if (const char* e = std::getenv("RZ_GLASS_SNAPSHOT")) snapOn = std::atoi(e) != 0;
        size_t slotPad = 0;                                                                                      // floats between neighbouring waves' slot regions
        if (const char* e = std::getenv("RZ_SLOT_STRIDE_PAD")) slotPad = (size_t)std::max(0, std::atoi(e));     // tuning aid
        const char* forceNo = std::getenv("RZ_DEBUG_NO_POOL_MEMORY");      // test aid: as if the device had no room for it
        if (nSlots >= groupsPerClaim && !(forceNo && std::atoi(forceNo) != 0) && (!c->sceneHasTransparency || snapOn) &&
            ensure_optional(c->dWavePools, (size_t)plan.grid * stride * RZ_GPOOL_FIELDS * sizeof(unsigned)) &&
            ensure_optional(c->dClaimScratch, (size_t)plan.grid * (claimScratchFloats + nSlots * slotFloats + slotPad) * sizeof(float)) &&
            ensure_optional(c->dWaitMeta, (size_t)plan.grid * 4 * nSlots * sizeof(int32_t))) {
            K.wpool = static_cast<unsigned*>(c->dWavePools.p);
            K.wpoolStride = (uint32_t)stride;
            K.wpoolChunk = (uint32_t)chunk;
            K.wslots = static_cast<float*>(c->dClaimScratch.p);
            K.wslotStride = (uint32_t)(claimScratchFloats + nSlots * slotFloats + slotPad);
            K.claimUnits = plan.claimUnits;
            K.claimScratchFloats = (uint32_t)claimScratchFloats;
            K.glassBoxHint = 1;
            if (const char* e = std::getenv("RZ_GLASS_BOX_HINT")) K.glassBoxHint = std::atoi(e) != 0 ? 1 : 0;      // A/B and test aid
            K.slotFloats = (uint32_t)slotFloats;
            K.nWaitSlots = nSlots;
            K.wmeta = static_cast<int32_t*>(c->dWaitMeta.p);
            K.drainEachClaim = plan.drainEachClaim ? 1 : 0;
        }
    }
    c->lastGlobalPool = K.wpool != nullptr && K.drainEachClaim == 0;
    c->lastCompact = K.wpool != nullptr;
    c->lastScratchBytes = K.wpool ? (size_t)plan.grid * ((size_t)K.wpoolStride * RZ_GPOOL_FIELDS * 4 + (size_t)K.wslotStride * 4 + (size_t)16 * K.nWaitSlots) : 0;
    // transparent scenes, persistent launches: room for every resident wave's sample prefixes (19 + 14 dwords per lane, 34 MB for
    // the grid), so that a sample's second version starts at its first transparent scatter instead of at the camera.
    // RZ_GLASS_SNAPSHOT=0 switches it off (A/B aid: same image either way).
    K.snap = nullptr; K.snapStride = 0;
    if (c->sceneHasTransparency && plan.perClaim > 0) {
        bool on = true;
        if (const char* e = std::getenv("RZ_GLASS_SNAPSHOT")) on = std::atoi(e) != 0;
        const size_t stride = (size_t)(RZ_SNAP_FIELDS + RZ_SNAP_TALLY + RZ_GVER_ROWS) * 64 + (size_t)RZ_GLATE_FIELDS * RZ_GLATE_CAP;
        if (on) {
            rc = ensure(c, c->dSnap, (size_t)plan.grid * stride * sizeof(float));
            if (rc != RZ_OK) return rc;
            K.snap = static_cast<float*>(c->dSnap.p);
            K.snapStride = (uint32_t)stride;
        }
    }
    RZ_HIP(c, hipEventRecord(c->evStart[evSlot], c->stream));
    if (K.nSlots > 0) launch_render_samples(K, counted, c->sceneHasTransparency, c->stream);
    RZ_HIP(c, hipEventRecord(c->evStop[evSlot], c->stream));
    c->lastLaunches = K.nSlots > 0 ? 1 : 0;
    c->lastGrid = plan.grid;
    c->lastPlan = rz_launch_plan{plan.groups, plan.grid, plan.perClaim, (plan.compact && K.wpool) ? plan.claimUnits : 0,
                                 (K.spp + 63) / 64, K.spp >= 64 ? 1 : 64 / K.spp, K.blasStackCap, K.blasOvfCap,
                                 c->sceneHasTransparency ? 1 : 0, (int32_t)((c->lastScratchBytes + (1u << 20) - 1) >> 20)};
    return RZ_OK;
}

int do_render(rz_ctx* c, bool counted, rz_counters* out) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!c->haveFrame) return fail(c, RZ_ERR_NOT_READY, "rz_set_frame has not been called");
    RZ_HIP(c, hipSetDevice(c->device));
    int rc = finalize(c);
    if (rc != RZ_OK) return rc;
    const rz_frame_params& f = c->frame;
    const size_t nPix = (size_t)f.width * f.height;
    float4* accum = nullptr;
    if (c->extAccum) {
        if (c->extAccumBytes < nPix * 16) return fail(c, RZ_ERR_BUFFER_SIZE, "bound accumulation buffer holds %zu bytes, frame needs %zu", c->extAccumBytes, nPix * 16);
        accum = static_cast<float4*>(c->extAccum);
    } else {
        accum = static_cast<float4*>(c->ownAccum.p);
    }
    KParams K{};
    K.pairs = static_cast<const DevPair*>(c->dPairs.p);
    K.tris = static_cast<const DevTri*>(c->dTris.p);
    K.triN = static_cast<const DevTriN*>(c->dTriN.p);
    K.instances = static_cast<const DevInstance*>(c->dInst.p);
    K.tlasDfs = static_cast<const TlasDfs*>(c->dTlasDfs.p);
    K.tlasIndices = static_cast<const int32_t*>(c->dTlasIdx.p);
    K.materials = static_cast<const DevMaterial*>(c->dMat.p);
    K.lights = static_cast<const DevLight*>(c->dLight.p);
    K.accum = accum;
    K.ior = static_cast<float*>(c->dIor.p);
    K.nTlasDfs = c->nTlasDfs;
    K.nLights = std::max(0, std::min<int>(f.num_lights, (int)hostCount<rz_light>(c, RZ_BIND_LIGHTS)));
    K.nMaterials = (int)hostCount<rz_material>(c, RZ_BIND_MATERIALS);
    K.width = f.width; K.height = f.height;
    K.tilesX = (f.width + RZ_TILE_W - 1) / RZ_TILE_W;
    K.tilesY = (f.height + RZ_TILE_H - 1) / RZ_TILE_H;
    const int nTiles = K.tilesX * K.tilesY;
    K.tileRank = f.tile_rank; K.tileNRanks = f.tile_nranks;
    K.nLocalTiles = (nTiles - f.tile_rank + f.tile_nranks - 1) / f.tile_nranks;
    K.maxBounces = f.bounce_budget > 0 ? f.bounce_budget : 5;      // FS:673
    K.spp = f.spp; K.sampleBase = f.sample_base;
    K.blasStackCap = std::max(1, c->maxBlasDepth - 1);
    // a pop followed by two pushes never holds more entries than the tree has levels (FS:460: stack[64])
    K.tlasStackCap = 0;         // the TLAS walk keeps no stack (rz_trace.h: trace_closest)
    std::memcpy(K.invView, f.inv_view, 64);
    std::memcpy(K.invProj, f.inv_proj, 64);
    std::memcpy(K.camPos, f.cam_pos, 12);
    std::memcpy(K.hemi0, c->hemi0, 12);
    {
        const long long nIdx = c->deviceOwnsTlas ? (long long)c->tlasHostCounts[1] : (long long)hostCount<int32_t>(c, RZ_BIND_TLAS_INDICES);
        K.traceRoundCap = (int)std::min<long long>(0x7fffffff, nIdx + (long long)c->nTlasDfs + 64);
        // Spread rays are traced lane by lane when they have several instances to spread over: with two (C2: floor + mesh) the
        // wave-cursor walk's scalar fetches and octant tests are worth more than walking both instances at once.
        // RZ_SPREAD_MIN_INSTANCES overrides the threshold (0 = never; A/B aid).
        long long minInst = RZ_SPREAD_MIN_INSTANCES;
        if (const char* e = std::getenv("RZ_SPREAD_MIN_INSTANCES")) minInst = std::atoll(e);
        K.spreadTrace = (minInst > 0 && nIdx >= minInst) ? 1 : 0;
        K.regularBoxes = c->irregularBoxes ? 0 : 1;
    }
    const size_t perWave = (size_t)K.blasStackCap * 512 + (size_t)K.tlasStackCap * 256 + 4864;
    if (perWave * 4 > 160 * 1024)   // sized for the largest (4-wave) workgroup
        return fail(c, RZ_ERR_BAD_SCENE, "BLAS depth %d needs %zu B of LDS stack per wave; the limit is %d", c->maxBlasDepth, perWave, 40 * 1024);
    if (counted) {
        rc = ensure(c, c->dCounters, sizeof(DevCounters) + 160 * sizeof(unsigned long long));
        if (rc != RZ_OK) return rc;
        RZ_HIP(c, hipMemsetAsync(c->dCounters.p, 0, sizeof(DevCounters) + 160 * sizeof(unsigned long long), c->stream));
        K.counters = static_cast<DevCounters*>(c->dCounters.p);
    }
    // The event pair brackets the render kernels of this call (for the one-lane-per-sample path: the
    // rz_render_samples launches; its small ordered-sum kernel runs after the stop event when unchunked).
    const int slot = c->ringHead;
    if (use_samples(c)) {
        rc = render_samples(c, K, counted, slot);
        if (rc != RZ_OK) return rc;
        c->lastKernel = c->sceneHasTransparency ? (c->lastCompact ? "rz_render_samples<glass>+pool" : "rz_render_samples<glass>")
                                                : (c->lastGlobalPool ? "rz_render_samples+pool" : "rz_render_samples");
    } else {
        RZ_HIP(c, hipEventRecord(c->evStart[slot], c->stream));
        launch_render_pixels(K, counted, c->stream);
        RZ_HIP(c, hipEventRecord(c->evStop[slot], c->stream));
        c->lastLaunches = 1;
        c->lastKernel = "rz_render_pixels";
        c->lastPlan = rz_launch_plan{K.nLocalTiles, K.nLocalTiles, 0, 0, (K.spp + 63) / 64, 64, K.blasStackCap, 0, c->sceneHasTransparency ? 1 : 0, 0};
    }
    RZ_HIP(c, hipGetLastError());
    c->ringHead = (slot + 1) % rz_ctx::kRing;
    c->ringCount = std::min(c->ringCount + 1, (int)rz_ctx::kRing);
    c->timed = true;
    if (counted && out) {
        DevCounters h{};
        RZ_HIP(c, hipMemcpyAsync(&h, c->dCounters.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        out->samples = h.samples; out->traversals = h.traversals; out->tlas_nodes = h.tlas_nodes;
        out->tlas_leaf_indices = h.tlas_leaf_indices; out->instances = h.instances; out->blas_nodes = h.blas_nodes;
        out->triangles = h.triangles; out->materials = h.materials; out->light_fetches = h.light_fetches;
        out->pixels = h.pixels;
        out->scatters = h.scatters; out->diffuse_scatters = h.diffuse_scatters; out->hemi_draws = h.hemi_draws; out->lit_lights = h.lit_lights; out->triangles_past_u = h.triangles_past_u;
#ifdef RZ_PROF
        unsigned long long pr[160];
        RZ_HIP(c, hipMemcpy(pr, static_cast<char*>(c->dCounters.p) + sizeof(DevCounters), sizeof pr, hipMemcpyDeviceToHost));
        for (int r = 0; r < 8; ++r) {       // per query round of a path: 0 primary, 1-2 shadow, 3 first bounce, ...
            const unsigned long long* q = pr + 32 + 11 * r;
            auto avg = [](unsigned long long l, unsigned long long e) { return e ? (double)l / (double)e : 0.0; };
            fprintf(stderr, "[rz_prof] round %d: queries %10llu x %4.1f lanes | descend steps %11llu x %4.1f (uniform %11llu) | triangle tests %11llu x %4.1f | instance entries %10llu x %4.1f | trace wave-cycles %llu\n",
                    r, q[8], avg(q[9], q[8]), q[0], avg(q[1], q[0]), q[6], q[2], avg(q[3], q[2]), q[4], avg(q[5], q[4]), q[10]);
        }
        dump_wave_log(use_samples(c) ? (int)std::min<long long>(c->lastGrid, 1 << 17) : K.nLocalTiles);
        static const char* namesPx[] = {"blas loop iter", "leaf branch", "triangle test", "internal branch", "tlas pop", "instance enter", "outer iter", "uniform pair"};
        const char** names = namesPx;
        for (int k = 0; k < 8; ++k)
            fprintf(stderr, "[rz_prof] %-16s wave-execs %12llu  lanes %14llu  avg active lanes %.1f\n", names[k], pr[2 * k], pr[2 * k + 1], pr[2 * k] ? (double)pr[2 * k + 1] / (double)pr[2 * k] : 0.0);
        fprintf(stderr, "[rz_prof] wave cycles: begin %llu  trace %llu  advance %llu | inside trace: descend loops %llu  leaf phases %llu  whole BLAS walks %llu\n", pr[16], pr[17], pr[18], pr[19], pr[20], pr[21]);
        fprintf(stderr, "[rz_prof] compacting claims: phase 1 (units) %llu  pool rounds %llu wave cycles; %llu rounds with %llu paths = %.1f lanes per round\n", pr[23], pr[28], pr[29], pr[30], pr[29] ? (double)pr[30] / (double)pr[29] : 0.0);
        fprintf(stderr, "[rz_prof] cross-claim pools: %llu traced, %llu queries in them, shade rounds %llu wave cycles\n", pr[29], pr[30], pr[28]);
        fprintf(stderr, "[rz_prof] pool_trace (a claim's pooled queries traced together): %llu wave cycles = T phases %llu + B phases %llu (of which refills %llu); round 7 above = its steps\n", pr[120], pr[121], pr[122], pr[123]);
        fprintf(stderr, "[rz_prof] descend steps of the general (per-lane) walk by children entered: none %llu  one %llu  both %llu (lane-level; the scalar-unit steps are not in here); in the pools' walks: none %llu  one %llu  both %llu\n", pr[132], pr[133], pr[134], pr[136], pr[137], pr[138]);
        fprintf(stderr, "[rz_prof] wait slots: end-of-claim sections %llu wave cycles (of which the claim-end sums %llu); slot_sums inside pool_process %llu\n", pr[124], pr[126], pr[125]);
        fprintf(stderr, "[rz_prof] inside advance: sky %llu  hit %llu  start_light %llu  shade_light %llu  scatter %llu (hemisphere %llu)  shadow step %llu\n", pr[22], pr[23], pr[24], pr[25], pr[26], pr[27], pr[28]);
#endif
    }
    return RZ_OK;
}

}  // namespace

extern "C" {

// The hash of the sources and flags this library was built from (rayzen_amd/build.py: source_hash, passed as
// -DRZ_SOURCE_HASH): lets a benchmark or a test tie the LOADED library to the tree and to a committed counter file.
#ifndef RZ_SOURCE_HASH
#define RZ_SOURCE_HASH "unstamped"
#endif
static const char rz_stamp[] = "RZSRCHASH:" RZ_SOURCE_HASH;
const char* rz_source_hash(void) { return rz_stamp + 10; }
const char* rz_version(void) { return "rayzen_hip 0.5 (gfx950)"; }
int rz_abi_version(void) { return RZ_ABI_VERSION; }
int rz_math_flavour(void) { return RZ_MATH_FLAVOUR; }

int rz_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

size_t rz_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(rz_triangle);
        case 1: return sizeof(rz_bvh_node);
        case 2: return sizeof(rz_bvh_instance);
        case 3: return sizeof(rz_material);
        case 4: return sizeof(rz_light);
        case 5: return sizeof(rz_frame_params);
        case 6: return sizeof(rz_counters);
        default: return 0;
    }
}

const char* rz_last_error(const rz_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

rz_ctx* rz_create(int device, unsigned flags) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(nullptr, RZ_ERR_NO_DEVICE, "no HIP device (%s)", hipGetErrorString(e)); return nullptr; }
    if (device < 0 || device >= n) { fail(nullptr, RZ_ERR_INVALID_ARG, "device %d of %d", device, n); return nullptr; }
    rz_ctx* c = new (std::nothrow) rz_ctx();
    if (!c) { fail(nullptr, RZ_ERR_HIP, "out of host memory"); return nullptr; }
    c->device = device;
    c->flags = flags;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < rz_ctx::kRing; ++i)
        ok = hipEventCreate(&c->evStart[i]) == hipSuccess && hipEventCreate(&c->evStop[i]) == hipSuccess;
    if (!ok) {
        fail(nullptr, RZ_ERR_HIP, "cannot create stream/events on device %d", device);
        rz_destroy(c);
        return nullptr;
    }
    c->stream = c->ownStream;
    const int hrc = compute_hemi0(c->hemi0, c->stream);
    if (hrc != 0) {
        fail(nullptr, RZ_ERR_HIP, "cannot run the set-up kernel on device %d: %s", device, hipGetErrorString((hipError_t)(-hrc)));
        rz_destroy(c);
        return nullptr;
    }
    return c;
}

void rz_destroy(rz_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&c->dPairs, &c->dTris, &c->dInst, &c->dTlasNodes, &c->dTlasIdx, &c->dMat, &c->dLight,
                      &c->dCounters, &c->dResolve, &c->dGroupCtr, &c->dBlasOvf, &c->ownAccum, &c->dIor, &c->dXforms, &c->dInstRef, &c->dTlasScratch, &c->dProjBoxes, &c->dBuildWs, &c->dTlasDfs, &c->dTriN, &c->dRawNodes, &c->dRawIdx, &c->dRawTris, &c->dRelayoutWs, &c->dClaimScratch, &c->dWavePools, &c->dWaitMeta, &c->dSnap
                      })
        b->release();
    if (c->tlasHostCounts) (void)hipHostFree(c->tlasHostCounts);
    if (c->relayoutPinned) (void)hipHostFree(c->relayoutPinned);
    for (int i = 0; i < rz_ctx::kRing; ++i) {
        if (c->evStart[i]) (void)hipEventDestroy(c->evStart[i]);
        if (c->evStop[i]) (void)hipEventDestroy(c->evStop[i]);
    }
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

static int upload_impl(rz_ctx* c, rz_binding binding, const void* data, size_t bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    const size_t es = elem_size((int)binding);
    if (es == 0) return fail(c, RZ_ERR_INVALID_ARG, "unknown binding %d", (int)binding);
    if (bytes % es) return fail(c, RZ_ERR_INVALID_ARG, "binding %d: %zu bytes is not a multiple of the %zu-byte element", (int)binding, bytes, es);
    if (bytes && !data) return fail(c, RZ_ERR_INVALID_ARG, "null data");
    if (c->deviceOwnsTlas) { int rc = sync_host_from_device(c); if (rc != RZ_OK) return rc; }
    if (c->geomOnDevice && (binding == RZ_BIND_BLAS_NODES || binding == RZ_BIND_BLAS_INDICES)) {
        int rc = sync_geom_host(c);         // the other of the two arrays must survive on the host
        if (rc != RZ_OK) return rc;
        c->geomOnDevice = false;
    }
    alloc_point(c);
    c->host[binding].assign(static_cast<const unsigned char*>(data), static_cast<const unsigned char*>(data) + bytes);
    c->present[binding] = true;
    switch (binding) {
        case RZ_BIND_MATERIALS: c->matDirty = true; break;
        case RZ_BIND_LIGHTS: c->lightDirty = true; break;
        case RZ_BIND_TLAS_NODES: case RZ_BIND_TLAS_INDICES: c->tlasDirty = true; break;
        case RZ_BIND_INSTANCES: c->instDirty = true; break;
        default: c->geomDirty = true; break;
    }
    return RZ_OK;
}

static int update_impl(rz_ctx* c, rz_binding binding, size_t offset, const void* data, size_t bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (elem_size((int)binding) == 0) return fail(c, RZ_ERR_INVALID_ARG, "unknown binding %d", (int)binding);
    if (!c->present[binding]) return fail(c, RZ_ERR_NOT_READY, "binding %d has not been uploaded", (int)binding);
    if (bytes && !data) return fail(c, RZ_ERR_INVALID_ARG, "null data");
    if (c->deviceOwnsTlas) { int rc = sync_host_from_device(c); if (rc != RZ_OK) return rc; }
    if (c->geomOnDevice && (binding == RZ_BIND_BLAS_NODES || binding == RZ_BIND_BLAS_INDICES)) {
        int rc = sync_geom_host(c);
        if (rc != RZ_OK) return rc;
        c->geomOnDevice = false;            // the host copies are the truth again
    }
    if (offset > c->host[binding].size() || bytes > c->host[binding].size() - offset)
        return fail(c, RZ_ERR_OUT_OF_RANGE, "binding %d: update [%zu,+%zu) past its %zu bytes", (int)binding, offset, bytes, c->host[binding].size());
    if (bytes == 0) return RZ_OK;
    if (std::memcmp(c->host[binding].data() + offset, data, bytes) == 0) return RZ_OK;   // unchanged: nothing to redo
    std::memcpy(c->host[binding].data() + offset, data, bytes);
    switch (binding) {
        case RZ_BIND_MATERIALS: c->matDirty = true; break;
        case RZ_BIND_LIGHTS: c->lightDirty = true; break;
        case RZ_BIND_TLAS_NODES: case RZ_BIND_TLAS_INDICES: c->tlasDirty = true; break;
        case RZ_BIND_INSTANCES: c->instDirty = true; break;
        default: c->geomDirty = true; break;
    }
    return RZ_OK;
}

static int update_transforms_impl(rz_ctx* c, const float* transforms, size_t n) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!transforms && n) return fail(c, RZ_ERR_INVALID_ARG, "null transforms");
    RZ_HIP(c, hipSetDevice(c->device));
    int rc = finalize(c);           // the device scene must exist (BLAS root boxes live in DevInstance)
    if (rc != RZ_OK) return rc;
    const size_t nInst = hostCount<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
    if (n != nInst) return fail(c, RZ_ERR_INVALID_ARG, "%zu transforms for %zu instances", n, nInst);
    if (n == 0) return RZ_OK;
    if (n > (size_t)1 << 20) return fail(c, RZ_ERR_INVALID_ARG, "too many instances for the device TLAS builder");
    if (!c->tlasHostCounts) RZ_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->tlasHostCounts), 64, hipHostMallocDefault));
    rc = ensure(c, c->dXforms, n * 64);
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dInstRef, n * sizeof(rz_bvh_instance));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dTlasNodes, (2 * n) * sizeof(rz_bvh_node));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dTlasIdx, n * sizeof(int32_t));
    if (rc != RZ_OK) return rc;
    // scratch: worldMin, worldMax (3n floats each), order + depth (2n+8 ints... the depth stack shares order's tail), stack 3*(2n+8), counts 16
    const size_t stackInts = 3 * (2 * n + 8), orderInts = n + (2 * n + 8);
    rc = ensure(c, c->dTlasDfs, (2 * n) * sizeof(TlasDfs));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, c->dTlasScratch, (6 * n) * 4 + orderInts * 4 + stackInts * 4 + 64 + (4 * n + 12 * (n + 1)) * 4);
    if (rc != RZ_OK) return rc;
    // the reference-layout records keep their offsets: seed them from the host copy once per host-side change
    RZ_HIP(c, hipMemcpyAsync(c->dInstRef.p, c->host[RZ_BIND_INSTANCES].data(), n * sizeof(rz_bvh_instance), hipMemcpyHostToDevice, c->stream));
    RZ_HIP(c, hipMemcpyAsync(c->dXforms.p, transforms, n * 64, hipMemcpyHostToDevice, c->stream));
    TlasWork W{};
    char* sc = static_cast<char*>(c->dTlasScratch.p);
    W.transforms = static_cast<const float*>(c->dXforms.p);
    W.instances = static_cast<DevInstance*>(c->dInst.p);
    W.refInstances = static_cast<rz_bvh_instance*>(c->dInstRef.p);
    W.nodes = static_cast<TlasNode*>(c->dTlasNodes.p);
    W.indices = static_cast<int32_t*>(c->dTlasIdx.p);
    W.dfs = static_cast<TlasDfs*>(c->dTlasDfs.p);
    W.worldMin = reinterpret_cast<float*>(sc);
    W.worldMax = W.worldMin + 3 * n;
    W.order = reinterpret_cast<int32_t*>(W.worldMax + 3 * n);
    W.stack = W.order + orderInts;
    W.outCounts = W.stack + stackInts;
    W.scratch = W.outCounts + 16;
    W.n = (int)n;
    launch_tlas_refit(W, c->stream);
    RZ_HIP(c, hipGetLastError());
    RZ_HIP(c, hipMemcpyAsync(c->tlasHostCounts, W.outCounts, 12, hipMemcpyDeviceToHost, c->stream));
    // the caller's transform array may die after this call returns, and the TLAS depth sizes the LDS stack
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    c->devTlasNodes = c->tlasHostCounts[0];
    c->nTlasDfs = c->tlasHostCounts[0];
    c->tlasDepth = std::max(1, c->tlasHostCounts[2]);
    c->deviceOwnsTlas = true;
    return RZ_OK;
}

static int build_geometry_impl(rz_ctx* c, const rz_triangle* triangles, size_t nTris, rz_mesh_build* meshes, size_t nMeshes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if ((nTris && !triangles) || (nMeshes && !meshes)) return fail(c, RZ_ERR_INVALID_ARG, "null argument");
    if (nTris > ((size_t)1 << 30)) return fail(c, RZ_ERR_INVALID_ARG, "too many triangles");
    size_t capNodes = 0, capIdx = 0, maxN = 0;
    for (size_t i = 0; i < nMeshes; ++i) {
        const rz_mesh_build& m = meshes[i];
        if (m.first_triangle > nTris || m.n_triangles > nTris - m.first_triangle)
            return fail(c, RZ_ERR_INVALID_ARG, "mesh %zu: triangles [%zu,+%zu) outside the %zu given", i, m.first_triangle, m.n_triangles, nTris);
        capNodes += m.n_triangles ? 2 * m.n_triangles - 1 : 1;
        capIdx += m.n_triangles;
        maxN = std::max(maxN, m.n_triangles);
    }
    if (capNodes > ((size_t)1 << 30)) return fail(c, RZ_ERR_INVALID_ARG, "too many nodes");
    RZ_HIP(c, hipSetDevice(c->device));
    if (c->deviceOwnsTlas) { int rc = sync_host_from_device(c); if (rc != RZ_OK) return rc; }
    // The build goes into FRESH buffers that replace the context's node / index arrays only when every mesh has been built:
    // a call that fails half-way (HIP error, internal limit, out of host memory) leaves the context exactly as it was -- the
    // arrays of an earlier rz_build_geometry, which devRoots / devNodes / geomOnDevice still describe, are never written to.
    struct Fresh { DevBuf nodes, idx; ~Fresh() { nodes.release(); idx.release(); } } fresh;
    int rc = ensure(c, fresh.nodes, std::max<size_t>(capNodes, 1) * sizeof(rz_bvh_node));
    if (rc != RZ_OK) return rc;
    rc = ensure(c, fresh.idx, std::max<size_t>(capIdx, 1) * sizeof(int32_t));
    if (rc != RZ_OK) return rc;
    if (maxN) { rc = ensure(c, c->dBuildWs, blas_build_workspace_bytes(maxN)); if (rc != RZ_OK) return rc; }
    std::map<int, rz_bvh_node> roots;
    size_t nodeOff = 0, idxOff = 0;
    rz_bvh_node* dNodes = static_cast<rz_bvh_node*>(fresh.nodes.p);
    int32_t* dIdx = static_cast<int32_t*>(fresh.idx.p);
    for (size_t i = 0; i < nMeshes; ++i) {
        rz_mesh_build& m = meshes[i];
        int nn = 1, depth = 1;
        rz_bvh_node root{};
        if (m.n_triangles == 0) {       // BVH.cpp:101-118 on an empty mesh: one root with an inverted box and no triangles
            const float fmax = std::numeric_limits<float>::max();
            for (int k = 0; k < 3; ++k) { root.boundsMin[k] = fmax; root.boundsMax[k] = -fmax; }
            root.leftFirst = 0; root.count = 0;
            RZ_HIP(c, hipMemcpyAsync(dNodes + nodeOff, &root, sizeof root, hipMemcpyHostToDevice, c->stream));
            RZ_HIP(c, hipStreamSynchronize(c->stream));
        } else {
            const int e = blas_build_device(triangles + m.first_triangle, m.n_triangles, c->dBuildWs.p, c->dBuildWs.cap, dNodes + nodeOff,
                                            dIdx + idxOff, &nn, &depth, nullptr, c->stream);
            if (e > 0) return fail(c, RZ_ERR_HIP, "device BLAS build of mesh %zu: %s", i, hipGetErrorString((hipError_t)e));
            if (e < 0) return fail(c, RZ_ERR_HIP, "device BLAS build of mesh %zu: internal limit", i);
            RZ_HIP(c, hipMemcpy(&root, dNodes + nodeOff, sizeof root, hipMemcpyDeviceToHost));
        }
        m.node_offset = (int32_t)nodeOff; m.index_offset = (int32_t)idxOff; m.n_nodes = nn; m.depth = depth; m.root = root;
        alloc_point(c);
        roots[(int)nodeOff] = root;
        nodeOff += (size_t)nn; idxOff += m.n_triangles;
    }
    // the three geometry bindings now are: the caller's triangles (host copy, uploaded by the re-layout as usual) and the
    // device-resident node / index arrays
    alloc_point(c);
    {   // (the last allocation that can fail: a copy first, then nothing below throws)
        std::vector<unsigned char> tcopy(reinterpret_cast<const unsigned char*>(triangles), reinterpret_cast<const unsigned char*>(triangles) + nTris * sizeof(rz_triangle));
        c->host[RZ_BIND_TRIANGLES].swap(tcopy);
    }
    RZ_HIP(c, hipStreamSynchronize(c->stream));        // nothing in flight reads the old arrays
    std::swap(c->dRawNodes, fresh.nodes);               // (the old ones are released with `fresh`)
    std::swap(c->dRawIdx, fresh.idx);
    c->host[RZ_BIND_BLAS_NODES].clear();
    c->host[RZ_BIND_BLAS_INDICES].clear();
    c->present[RZ_BIND_TRIANGLES] = c->present[RZ_BIND_BLAS_NODES] = c->present[RZ_BIND_BLAS_INDICES] = true;
    c->geomOnDevice = true; c->geomHostFresh = false;
    c->devNodes = nodeOff; c->devIdx = idxOff;
    c->devRoots.swap(roots);
    c->geomDirty = true;
    return RZ_OK;
}

static int build_blas_impl(rz_ctx* c, const rz_triangle* tris, size_t n, rz_bvh_node* nodes_out, size_t nodes_cap, int32_t* indices_out,
                  size_t* n_nodes, int* depth, float* device_ms) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (n && !tris) return fail(c, RZ_ERR_INVALID_ARG, "null triangles");
    if (!nodes_out || (n && !indices_out)) return fail(c, RZ_ERR_INVALID_ARG, "null output");
    if (n > (size_t)1 << 30) return fail(c, RZ_ERR_INVALID_ARG, "too many triangles");
    if (nodes_cap < (n ? 2 * n - 1 : 1)) return fail(c, RZ_ERR_BUFFER_SIZE, "nodes_out holds %zu nodes, up to %zu are needed", nodes_cap, n ? 2 * n - 1 : (size_t)1);
    if (device_ms) *device_ms = 0.0f;
    if (n == 0) {        // BVH.cpp:101-118 on an empty mesh: one root with an inverted box and no triangles
        const float fmax = std::numeric_limits<float>::max();
        rz_bvh_node r{};
        for (int k = 0; k < 3; ++k) { r.boundsMin[k] = fmax; r.boundsMax[k] = -fmax; }
        r.leftFirst = 0; r.count = 0;
        nodes_out[0] = r;
        if (n_nodes) *n_nodes = 1;
        if (depth) *depth = 1;
        return RZ_OK;
    }
    RZ_HIP(c, hipSetDevice(c->device));
    const size_t ws = blas_build_workspace_bytes(n);
    int rc = ensure(c, c->dBuildWs, ws);
    if (rc != RZ_OK) return rc;
    std::vector<rz_bvh_node> tmp;
    alloc_point(c);
    tmp.resize(2 * n + 2);
    int nn = 0, dp = 0;
    const int e = blas_build_device(tris, n, c->dBuildWs.p, c->dBuildWs.cap, tmp.data(), indices_out, &nn, &dp, device_ms, c->stream);
    if (e > 0) return fail(c, RZ_ERR_HIP, "device BLAS build: %s", hipGetErrorString((hipError_t)e));
    if (e < 0) return fail(c, RZ_ERR_HIP, "device BLAS build: internal limit");
    if ((size_t)nn > nodes_cap) return fail(c, RZ_ERR_BUFFER_SIZE, "nodes_out holds %zu nodes, %d were built", nodes_cap, nn);
    std::memcpy(nodes_out, tmp.data(), (size_t)nn * sizeof(rz_bvh_node));
    if (n_nodes) *n_nodes = (size_t)nn;
    if (depth) *depth = dp;
    return RZ_OK;
}

static int read_binding_impl(rz_ctx* c, rz_binding binding, void* out, size_t bytes, size_t* needed) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (elem_size((int)binding) == 0) return fail(c, RZ_ERR_INVALID_ARG, "unknown binding %d", (int)binding);
    if (!c->present[binding]) return fail(c, RZ_ERR_NOT_READY, "binding %d has not been uploaded", (int)binding);
    if (c->deviceOwnsTlas) { int rc = sync_host_from_device(c); if (rc != RZ_OK) return rc; c->deviceOwnsTlas = true; }
    if (c->geomOnDevice && (binding == RZ_BIND_BLAS_NODES || binding == RZ_BIND_BLAS_INDICES)) { int rc = sync_geom_host(c); if (rc != RZ_OK) return rc; }
    const size_t have = c->host[binding].size();
    if (needed) *needed = have;
    if (!out) return RZ_OK;
    if (bytes < have) return fail(c, RZ_ERR_BUFFER_SIZE, "binding %d holds %zu bytes, buffer has %zu", (int)binding, have, bytes);
    if (have) std::memcpy(out, c->host[binding].data(), have);
    return RZ_OK;
}

static int set_frame_impl(rz_ctx* c, const rz_frame_params* p) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!p) return fail(c, RZ_ERR_INVALID_ARG, "null params");
    if (p->width <= 0 || p->height <= 0 || (long long)p->width * p->height > (1ll << 28))
        return fail(c, RZ_ERR_INVALID_ARG, "resolution %dx%d", p->width, p->height);
    if (p->spp <= 0) return fail(c, RZ_ERR_INVALID_ARG, "spp %d", p->spp);
    if (p->sample_base < 0) return fail(c, RZ_ERR_INVALID_ARG, "sample_base %d", p->sample_base);
    if (p->tile_nranks < 1 || p->tile_rank < 0 || p->tile_rank >= p->tile_nranks)
        return fail(c, RZ_ERR_INVALID_ARG, "tile_rank %d of %d", p->tile_rank, p->tile_nranks);
    RZ_HIP(c, hipSetDevice(c->device));
    const size_t nPix = (size_t)p->width * p->height;
    const bool resized = !c->haveFrame || c->frame.width != p->width || c->frame.height != p->height;
    if (resized) {
        int rc = ensure(c, c->dIor, nPix * sizeof(float));
        if (rc != RZ_OK) return rc;
        if (!c->extAccum) {
            rc = ensure(c, c->ownAccum, nPix * 16);
            if (rc != RZ_OK) return rc;
            RZ_HIP(c, hipMemsetAsync(c->ownAccum.p, 0, nPix * 16, c->stream));
        }
    }
    // a different tile assignment: pixels this context owned before and no longer owns must read as zero again
    // (the group's reduce sums every member's whole buffer)
    if (!resized && c->haveFrame && (c->frame.tile_rank != p->tile_rank || c->frame.tile_nranks != p->tile_nranks)) {
        void* acc = c->extAccum ? c->extAccum : c->ownAccum.p;
        if (acc && (!c->extAccum || c->extAccumBytes >= nPix * 16)) RZ_HIP(c, hipMemsetAsync(acc, 0, nPix * 16, c->stream));
    }
    c->frame = *p;
    c->haveFrame = true;
    return RZ_OK;
}

int rz_set_stream(rz_ctx* c, void* hip_stream) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    return guarded(c, "rz_set_stream", [&]() -> int {
        (void)hipStreamSynchronize(c->stream);
        c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->ownStream;
        return RZ_OK;
    });
}

static int bind_accum_impl(rz_ctx* c, void* device_rgba, size_t bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    c->extAccum = device_rgba;
    c->extAccumBytes = device_rgba ? bytes : 0;
    if (!device_rgba && c->haveFrame) {
        const size_t nPix = (size_t)c->frame.width * c->frame.height;
        RZ_HIP(c, hipSetDevice(c->device));
        int rc = ensure(c, c->ownAccum, nPix * 16);
        if (rc != RZ_OK) return rc;
        RZ_HIP(c, hipMemsetAsync(c->ownAccum.p, 0, nPix * 16, c->stream));
    }
    return RZ_OK;
}

int rz_render(rz_ctx* c) {
    return guarded(c, "rz_render", [&] { return do_render(c, false, nullptr); });
}
int rz_render_counted(rz_ctx* c, rz_counters* out) {
    return guarded(c, "rz_render_counted", [&] { return do_render(c, true, out); });
}

int rz_sync(rz_ctx* c) {
#ifdef RZ_GSTATS
    if (c) { (void)hipStreamSynchronize(c->stream); dump_gstats(); }
#endif
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    return guarded(c, "rz_sync", [&]() -> int {
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        if (c->dGroupCtr.p) {       // did a kernel run into one of its "cannot happen" bounds?  (rz_kernels.hip: rz_backstop)
            unsigned bits = 0;
            unsigned* w = static_cast<unsigned*>(c->dGroupCtr.p) + RZ_ERRWORD;
            RZ_HIP(c, hipMemcpy(&bits, w, sizeof bits, hipMemcpyDeviceToHost));
            if (bits != 0u) {
                RZ_HIP(c, hipMemset(w, 0, sizeof bits));
                return fail(c, RZ_ERR_INTERNAL, "a render kernel reached a backstop (bits 0x%x:%s%s%s): pixels of the last frame(s) may be missing",
                            bits, (bits & 1u) ? " claim without wait slots" : "", (bits & 2u) ? " pool did not drain" : "", (bits & 4u) ? " currentIor chains did not resolve" : "");
            }
        }
        return RZ_OK;
    });
}

int rz_debug_poke_backstop(rz_ctx* c, unsigned bits) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    return guarded(c, "rz_debug_poke_backstop", [&]() -> int {
        int rc = ensure_group_counter(c);
        if (rc != RZ_OK) return rc;
        RZ_HIP(c, hipStreamSynchronize(c->stream));
        RZ_HIP(c, hipMemcpy(static_cast<unsigned*>(c->dGroupCtr.p) + RZ_ERRWORD, &bits, sizeof bits, hipMemcpyHostToDevice));
        return RZ_OK;
    });
}

void* rz_stream_handle(rz_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }

void* rz_accum_device_ptr(rz_ctx* c) {
    if (!c) return nullptr;
    return c->extAccum ? c->extAccum : c->ownAccum.p;
}

static int clear_accum_impl(rz_ctx* c) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!c->haveFrame) return fail(c, RZ_ERR_NOT_READY, "rz_set_frame has not been called");
    void* p = rz_accum_device_ptr(c);
    RZ_HIP(c, hipMemsetAsync(p, 0, (size_t)c->frame.width * c->frame.height * 16, c->stream));
    return RZ_OK;
}

static int read_accum_impl(rz_ctx* c, float* rgba, size_t bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!c->haveFrame) return fail(c, RZ_ERR_NOT_READY, "rz_set_frame has not been called");
    const size_t need = (size_t)c->frame.width * c->frame.height * 16;
    if (!rgba || bytes < need) return fail(c, RZ_ERR_BUFFER_SIZE, "rz_read_accum needs %zu bytes, got %zu", need, bytes);
    RZ_HIP(c, hipMemcpyAsync(rgba, rz_accum_device_ptr(c), need, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return RZ_OK;
}

static int resolve_rgba8_impl(rz_ctx* c, uint8_t* rgba8, size_t bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!c->haveFrame) return fail(c, RZ_ERR_NOT_READY, "rz_set_frame has not been called");
    const size_t nPix = (size_t)c->frame.width * c->frame.height;
    if (!rgba8 || bytes < nPix * 4) return fail(c, RZ_ERR_BUFFER_SIZE, "rz_resolve_rgba8 needs %zu bytes, got %zu", nPix * 4, bytes);
    int rc = ensure(c, c->dResolve, nPix * 4);
    if (rc != RZ_OK) return rc;
    launch_resolve(static_cast<const float4*>(rz_accum_device_ptr(c)), static_cast<uchar4*>(c->dResolve.p), (int)nPix, c->stream);
    RZ_HIP(c, hipGetLastError());
    RZ_HIP(c, hipMemcpyAsync(rgba8, c->dResolve.p, nPix * 4, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return RZ_OK;
}

static int present_impl(rz_ctx* c, const rz_present_params* pp, uint8_t* rgba8, size_t rgba8_bytes, float* rgb32f, size_t rgb32f_bytes) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!pp) return fail(c, RZ_ERR_INVALID_ARG, "null params");
    if (!c->haveFrame) return fail(c, RZ_ERR_NOT_READY, "rz_set_frame has not been called");
    RZ_HIP(c, hipSetDevice(c->device));
    int rc = finalize(c);
    if (rc != RZ_OK) return rc;
    const size_t nPix = (size_t)c->frame.width * c->frame.height;
    if (rgba8 && rgba8_bytes < nPix * 4) return fail(c, RZ_ERR_BUFFER_SIZE, "rgba8 buffer needs %zu bytes", nPix * 4);
    if (rgb32f && rgb32f_bytes < nPix * 12) return fail(c, RZ_ERR_BUFFER_SIZE, "rgb32f buffer needs %zu bytes", nPix * 12);
    rc = ensure(c, c->dResolve, nPix * 4 + nPix * 12);
    if (rc != RZ_OK) return rc;
    PresentParams P{};
    P.accum = static_cast<const float4*>(rz_accum_device_ptr(c));
    P.rgba8 = static_cast<uchar4*>(c->dResolve.p);
    P.rgb = reinterpret_cast<float*>(static_cast<char*>(c->dResolve.p) + nPix * 4);
    P.tlasNodes = static_cast<const TlasNode*>(c->dTlasNodes.p);
    P.tlasIndices = static_cast<const int32_t*>(c->dTlasIdx.p);
    P.instances = static_cast<const DevInstance*>(c->dInst.p);
    P.lights = static_cast<const DevLight*>(c->dLight.p);
    P.width = c->frame.width; P.height = c->frame.height;
    P.nTlasNodes = c->deviceOwnsTlas ? c->devTlasNodes : (int)hostCount<rz_bvh_node>(c, RZ_BIND_TLAS_NODES);
    P.nInstances = (int)hostCount<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
    P.nLights = std::max(0, std::min<int>(c->frame.num_lights, (int)hostCount<rz_light>(c, RZ_BIND_LIGHTS)));
    // camera.projectionMatrix * camera.viewMatrix, terms summed left to right (glsl:321)
    const float* A = c->frame.proj; const float* B = c->frame.view;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            P.viewProj[col * 4 + row] = ((A[0 * 4 + row] * B[col * 4 + 0] + A[1 * 4 + row] * B[col * 4 + 1]) +
                                         A[2 * 4 + row] * B[col * 4 + 2]) + A[3 * 4 + row] * B[col * 4 + 3];
    P.fps = pp->fps; P.showFps = pp->show_fps; P.showLights = pp->show_lights; P.showBvh = pp->show_bvh; P.bvhMode = pp->bvh_mode;
    P.pathLen = 0;
    if (pp->show_bvh && pp->bvh_mode == 1 && pp->selected_blas >= 0 && pp->selected_blas < P.nInstances) {
        // findBVHBranchIterative (glsl:257-307) does not depend on the pixel: walk it once here
        if (c->deviceOwnsTlas) { rc = sync_host_from_device(c); if (rc != RZ_OK) return rc; c->deviceOwnsTlas = true; }
        rc = sync_geom_host(c);
        if (rc != RZ_OK) return rc;
        const rz_bvh_instance* inst = hostArr<rz_bvh_instance>(c, RZ_BIND_INSTANCES);
        const rz_bvh_node* nodes = hostArr<rz_bvh_node>(c, RZ_BIND_BLAS_NODES);
        const int32_t* idx = hostArr<int32_t>(c, RZ_BIND_BLAS_INDICES);
        const long long nNodes = (long long)hostCount<rz_bvh_node>(c, RZ_BIND_BLAS_NODES), nIdx = (long long)hostCount<int32_t>(c, RZ_BIND_BLAS_INDICES);
        const rz_bvh_instance& S = inst[pp->selected_blas];
        const int nodeOffset = S.blasNodeOffset, triOffset = S.blasTriOffset;
        const int nodeCount = (pp->selected_blas + 1 < P.nInstances) ? inst[pp->selected_blas + 1].blasNodeOffset - nodeOffset : (int)nNodes - nodeOffset;
        auto leafHas = [&](const rz_bvh_node& n) {
            for (int k = 0; k < n.count; ++k) { long long j = (long long)triOffset + n.leftFirst + k; if (j >= 0 && j < nIdx && idx[j] == pp->selected_tri) return true; }
            return false;
        };
        int path[32], len = 0, cur = 0;
        for (int depth = 0; depth < 32; ++depth) {
            path[len++] = cur;
            if ((long long)nodeOffset + cur < 0 || (long long)nodeOffset + cur >= nNodes) { len = 0; break; }
            const rz_bvh_node& node = nodes[nodeOffset + cur];
            if (node.count > 0) { if (!leafHas(node)) len = 0; break; }
            bool inLeft = false;
            int stack[32], sp = 0;
            stack[sp++] = node.leftFirst;
            while (sp > 0) {
                const int nidx = stack[--sp];
                if (nidx < 0 || nidx >= nodeCount || (long long)nodeOffset + nidx >= nNodes) continue;
                const rz_bvh_node& n = nodes[nodeOffset + nidx];
                if (n.count > 0) inLeft = leafHas(n);
                else if (sp + 2 <= 32) { stack[sp++] = n.leftFirst; stack[sp++] = n.leftFirst + 1; }
                if (inLeft) break;
            }
            cur = inLeft ? node.leftFirst : node.leftFirst + 1;
        }
        P.pathLen = len;
        for (int k = 0; k < len; ++k) {
            std::memcpy(P.pathMin[k], nodes[nodeOffset + path[k]].boundsMin, 12);
            std::memcpy(P.pathMax[k], nodes[nodeOffset + path[k]].boundsMax, 12);
        }
        std::memcpy(P.selTransform, S.transform, 64);
    }
    {   // projected-corner cache: one 112-B record per box a pixel may draw
        const size_t nBoxes = (size_t)P.nTlasNodes + (size_t)P.nInstances + 32;
        rc = ensure(c, c->dProjBoxes, nBoxes * 112);
        if (rc != RZ_OK) return rc;
        P.boxes = static_cast<ProjBox*>(c->dProjBoxes.p);
    }
    launch_present(P, c->stream);
    RZ_HIP(c, hipGetLastError());
    if (rgba8) RZ_HIP(c, hipMemcpyAsync(rgba8, P.rgba8, nPix * 4, hipMemcpyDeviceToHost, c->stream));
    if (rgb32f) RZ_HIP(c, hipMemcpyAsync(rgb32f, P.rgb, nPix * 12, hipMemcpyDeviceToHost, c->stream));
    RZ_HIP(c, hipStreamSynchronize(c->stream));
    return RZ_OK;
}

int rz_last_render_ms(rz_ctx* c, float* ms, int* launches) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (!c->timed) return fail(c, RZ_ERR_NOT_READY, "nothing rendered yet");
    return guarded(c, "rz_last_render_ms", [&]() -> int {
        const int slot = (c->ringHead + rz_ctx::kRing - 1) % rz_ctx::kRing;
        RZ_HIP(c, hipEventSynchronize(c->evStop[slot]));
        float t = 0.0f;
        RZ_HIP(c, hipEventElapsedTime(&t, c->evStart[slot], c->evStop[slot]));
        if (ms) *ms = t;
        if (launches) *launches = c->lastLaunches;
        return RZ_OK;
    });
}

const char* rz_last_kernel_name(const rz_ctx* c) { return c ? c->lastKernel : ""; }

int rz_render_history_ms(rz_ctx* c, float* ms, int cap) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    if (cap < 0 || (cap > 0 && !ms)) return fail(c, RZ_ERR_INVALID_ARG, "bad history buffer");
    return guarded(c, "rz_render_history_ms", [&]() -> int {
        const int n = std::min(c->ringCount, cap);
        for (int i = 0; i < n; ++i) {     // oldest of the last n first
            const int slot = (c->ringHead + 2 * rz_ctx::kRing - n + i) % rz_ctx::kRing;
            RZ_HIP(c, hipEventSynchronize(c->evStop[slot]));
            RZ_HIP(c, hipEventElapsedTime(&ms[i], c->evStart[slot], c->evStop[slot]));
        }
        c->ringCount = 0;
        return n;
    });
}

// ---- exported wrappers: no exception leaves the library (guarded(), above) ----
int rz_upload(rz_ctx* c, rz_binding binding, const void* data, size_t bytes) {
    return guarded(c, "rz_upload", [&] { return upload_impl(c, binding, data, bytes); });
}
int rz_update(rz_ctx* c, rz_binding binding, size_t offset, const void* data, size_t bytes) {
    return guarded(c, "rz_update", [&] { return update_impl(c, binding, offset, data, bytes); });
}
int rz_update_transforms(rz_ctx* c, const float* transforms, size_t n) {
    return guarded(c, "rz_update_transforms", [&] { return update_transforms_impl(c, transforms, n); });
}
int rz_build_blas(rz_ctx* c, const rz_triangle* tris, size_t n, rz_bvh_node* nodes_out, size_t nodes_cap, int32_t* indices_out, size_t* n_nodes, int* depth, float* device_ms) {
    return guarded(c, "rz_build_blas", [&] { return build_blas_impl(c, tris, n, nodes_out, nodes_cap, indices_out, n_nodes, depth, device_ms); });
}
int rz_read_binding(rz_ctx* c, rz_binding binding, void* out, size_t bytes, size_t* needed) {
    return guarded(c, "rz_read_binding", [&] { return read_binding_impl(c, binding, out, bytes, needed); });
}
int rz_set_frame(rz_ctx* c, const rz_frame_params* p) {
    return guarded(c, "rz_set_frame", [&] { return set_frame_impl(c, p); });
}
int rz_present(rz_ctx* c, const rz_present_params* pp, uint8_t* rgba8, size_t rgba8_bytes, float* rgb32f, size_t rgb32f_bytes) {
    return guarded(c, "rz_present", [&] { return present_impl(c, pp, rgba8, rgba8_bytes, rgb32f, rgb32f_bytes); });
}
int rz_resolve_rgba8(rz_ctx* c, uint8_t* rgba8, size_t bytes) {
    return guarded(c, "rz_resolve_rgba8", [&] { return resolve_rgba8_impl(c, rgba8, bytes); });
}
int rz_bind_accum(rz_ctx* c, void* device_rgba, size_t bytes) {
    return guarded(c, "rz_bind_accum", [&] { return bind_accum_impl(c, device_rgba, bytes); });
}
int rz_read_accum(rz_ctx* c, float* rgba, size_t bytes) {
    return guarded(c, "rz_read_accum", [&] { return read_accum_impl(c, rgba, bytes); });
}
int rz_clear_accum(rz_ctx* c) {
    return guarded(c, "rz_clear_accum", [&] { return clear_accum_impl(c); });
}

int rz_build_geometry(rz_ctx* c, const rz_triangle* triangles, size_t n_triangles, rz_mesh_build* meshes, size_t n_meshes) {
    return guarded(c, "rz_build_geometry", [&] { return build_geometry_impl(c, triangles, n_triangles, meshes, n_meshes); });
}

int rz_debug_read_layout(rz_ctx* c, int which, void* out, size_t bytes, size_t* needed) {
    return guarded(c, "rz_debug_read_layout", [&]() -> int {
        if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
        if (which != 0 && which != 1) return fail(c, RZ_ERR_INVALID_ARG, "which = %d", which);
        RZ_HIP(c, hipSetDevice(c->device));
        int rc = finalize(c);
        if (rc != RZ_OK) return rc;
        size_t have;
        const void* src;
        if (which == 0) { have = (c->layoutOnDevice ? (size_t)c->devPairsUsed : c->hPairs.size()) * sizeof(DevPair); src = c->dPairs.p; }
        else { have = (c->layoutOnDevice ? (size_t)c->devTrisUsed : c->hTris.size()) * sizeof(DevTri); src = c->dTris.p; }
        if (needed) *needed = have;
        if (!out) return RZ_OK;
        if (bytes < have) return fail(c, RZ_ERR_BUFFER_SIZE, "layout %d holds %zu bytes, buffer has %zu", which, have, bytes);
        if (have) {
            RZ_HIP(c, hipMemcpyAsync(out, src, have, hipMemcpyDeviceToHost, c->stream));
            RZ_HIP(c, hipStreamSynchronize(c->stream));
        }
        return RZ_OK;
    });
}

int rz_debug_last_plan(rz_ctx* c, rz_launch_plan* out) {
    if (!c || !out) return fail(c, RZ_ERR_INVALID_ARG, "rz_debug_last_plan: null argument");
    if (!c->timed) return fail(c, RZ_ERR_NOT_READY, "rz_debug_last_plan: nothing has been rendered yet");
    *out = c->lastPlan;
    return RZ_OK;
}

int rz_debug_fail_alloc(rz_ctx* c, int nth) {
    if (!c) return fail(nullptr, RZ_ERR_INVALID_ARG, "null context");
    c->failAllocCountdown = nth > 0 ? nth : 0;
    return RZ_OK;
}

}  // extern "C"
