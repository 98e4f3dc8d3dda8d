// rz_group.hip -- the multi-GPU group of include/rayzen_hip.h: N contexts, tile-sharded, ONE exchange step per frame.
//
// RayZen is single-GPU; what this replaces is its context lifetime (RayZen/src/main.cpp:228-241 create,
// :681-686 teardown) for N devices of one node.  Pixels shard by 8x8 tiles dealt round-robin (tile t -> rank t % N,
// rz_frame_params.tile_rank / tile_nranks); each member renders ALL samples of its own pixels (currentIor couples a
// pixel's samples, fragment_shader.glsl:674) into a buffer that is zero wherever it owns nothing.  The frame lands on the
// root by ONE exchange step, of which there are two:
//   * "reduce" (THE DEFAULT again since round 5 -- it is what BASELINE.json's north_star names, and the simplest thing RCCL
//     does): one ncclReduce(SUM, float, W * H * 4) of the whole accumulation buffers; tile sets are disjoint and non-owned
//     pixels are zero, so every pixel's single value is added to zeros: bit-identical to one GPU;
//   * "gather" (RZ_GROUP_TRANSPORT=gather or rz_group_set_transport(g, "gather"); round 4's default, demoted because it
//     has never moved a byte between two GPUs): every member packs the tiles it owns (1 KB each, 1 / N of the frame), sends
//     them straight to the root (ncclSend / ncclRecv: xGMI is point to point, the root has a link to every peer, and the
//     seven transfers of an 8-GPU node run side by side), and the root scatters the N packed sets into the frame.  At 1080p
//     on 8 GPUs a member moves 4.1 MB instead of taking part in a 33-MB ring reduce of which 7 / 8 is zeros.
// A gather whose enqueue fails switches the group to "reduce" for the rest of its life and lands the SAME frame that way
// (rz_group_transport() then reads "rccl-reduce(fallback: ...)"); every rank of a group must ask for the same transport --
// one-process groups do by construction, a launcher agrees on it before the first frame (bench.py: all_reduce(MIN)).
//
// RCCL is bound with dlopen when the first group is made: librayzen_hip.so carries no DT_NEEDED on the 570-MB librccl,
// a process that already has an RCCL mapped (e.g. through torch.distributed) shares that copy instead of running two
// collective runtimes side by side, and the types / enums still come from <rccl/rccl.h>, so every call is type-checked.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rayzen_hip.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    std::string path;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;      // (optional: the tile gather)
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
thread_local std::string g_group_error;

int gfail(std::string* where, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    try {
        if (where) *where = buf;
        g_group_error = buf;
    } catch (...) { }
    return code;
}

template <class F> bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

// Order: an RCCL already mapped into the process, $RZ_RCCL_LIBRARY, the ROCm install, the loader's search path.
int bind_rccl(std::string* err) {
    if (g_rccl.handle) return RZ_OK;
    std::vector<std::pair<std::string, int>> tries;
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
    if (const char* e = std::getenv("RZ_RCCL_LIBRARY")) tries.insert(tries.begin(), {e, RTLD_NOW | RTLD_LOCAL});
    tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL});
    tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL});
    tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL});
    std::string last;
    for (auto& t : tries) {
        void* h = dlopen(t.first.c_str(), t.second);
        if (!h) { if (const char* m = dlerror()) last = m; continue; }
        Rccl r;
        r.handle = h;
        r.path = t.first + ((t.second & RTLD_NOLOAD) ? " (already loaded in this process)" : "");
        const bool ok = bind(h, "ncclGetVersion", r.GetVersion) && bind(h, "ncclGetUniqueId", r.GetUniqueId) &&
                        bind(h, "ncclCommInitRank", r.CommInitRank) && bind(h, "ncclCommInitAll", r.CommInitAll) &&
                        bind(h, "ncclCommDestroy", r.CommDestroy) && bind(h, "ncclGroupStart", r.GroupStart) &&
                        bind(h, "ncclGroupEnd", r.GroupEnd) && bind(h, "ncclReduce", r.Reduce) &&
                        bind(h, "ncclGetErrorString", r.GetErrorString);
        if (!ok) { last = t.first + ": an ncclXxx entry point is missing"; dlclose(h); continue; }
        if (!(bind(h, "ncclSend", r.Send) && bind(h, "ncclRecv", r.Recv))) { r.Send = nullptr; r.Recv = nullptr; }
        g_rccl = r;
        return RZ_OK;
    }
    return gfail(err, RZ_ERR_NO_DEVICE, "cannot bind RCCL (set RZ_RCCL_LIBRARY): %s", last.c_str());
}

}  // namespace

struct rz_group {
    int nranks = 0;
    std::vector<rz_ctx*> ctx;           // local members
    std::vector<int> rank;              // their global ranks
    std::vector<int> device;
    std::vector<ncclComm_t> comm;
    // the reduced frame lives on the root member (allocated on first use, on that member's device)
    void* frame = nullptr;
    size_t frameBytes = 0;
    int frameLocal = -1;                // local index of the member that holds `frame`
    int lastRoot = -1;
    // HIP events on each member's stream around its share of the last rz_group_reduce (rz_group_last_reduce_ms)
    std::vector<hipEvent_t> evBefore, evAfter;
    bool reduceTimed = false;
    // the tile gather: each member's packed tiles (on its device), the root's N packed sets (on the root's device)
    bool gather = false;                // true: the tile gather; false: ncclReduce of the whole buffers (the default)
    bool loopback = false;              // RZ_GROUP_LOOPBACK: no communicator, device copies instead of RCCL (always the gather)
    bool distinctDevices = false;       // loopback only: every member on a device of its own (copies cross xGMI), not a one-GPU rehearsal
    std::string fallback;               // why a gather group fell back to the reduce (empty: it did not)
    std::string transportText;          // rz_group_transport()'s answer
    std::vector<void*> packed;
    std::vector<size_t> packedBytes;
    std::vector<hipEvent_t> evSent;     // loopback: a member's packed tiles have reached the root's buffer
    hipEvent_t evScattered = nullptr;   // loopback: the root has read the gathered sets of the last frame
    void* gathered = nullptr;
    size_t gatheredBytes = 0;
    int gatheredLocal = -1;
    int width = 0, height = 0;
    bool haveFrame = false;
    std::string err;
};

namespace {

#define RZG_NCCL(g, call)                                                                                         \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess) return gfail(&(g)->err, RZ_ERR_HIP, "%s: %s", #call, g_rccl.GetErrorString(r_));   \
    } while (0)
#define RZG_HIP(g, call)                                                                                          \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return gfail(&(g)->err, RZ_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));        \
    } while (0)

int member_fail(rz_group* g, int local, int rc, const char* what) {
    return gfail(&g->err, rc, "%s on rank %d (device %d): %s", what, g->rank[local], g->device[local], rz_last_error(g->ctx[local]));
}

int local_of_rank(const rz_group* g, int r) {
    for (size_t i = 0; i < g->rank.size(); ++i)
        if (g->rank[i] == r) return (int)i;
    return -1;
}

// Tile t of the frame (8 x 8 pixels, row-major over the tile grid) belongs to rank t % N and is that rank's local tile t / N
// (the dealing of rz_frame_params.tile_rank / tile_nranks).  A packed set is a rank's local tiles in order, 64 float4 each,
// pixel l of a tile at (l & 7, l >> 3); pixels beyond the frame's edge and tiles beyond the last one are zeros.
__global__ void rz_pack_tiles(const float4* __restrict__ accum, float4* __restrict__ packed, int width, int height, int tilesX, int nTiles,
                              int rank, int nranks, int perRank) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= perRank * 64) return;
    const int lt = idx >> 6, l = idx & 63;
    const int tile = lt * nranks + rank;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (tile < nTiles) {
        const int ty = tile / tilesX, tx = tile - ty * tilesX;
        const int px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        if (px < width && py < height) v = accum[(size_t)py * width + px];
    }
    packed[idx] = v;
}

__global__ void rz_unpack_tiles(const float4* __restrict__ gathered, float4* __restrict__ frame, int width, int height, int tilesX,
                                int nranks, int perRank) {
    const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= width || py >= height) return;
    const int tile = (py >> 3) * tilesX + (px >> 3);
    const int r = tile % nranks, lt = tile / nranks;
    frame[(size_t)py * width + px] = gathered[((size_t)r * perRank + lt) * 64 + ((py & 7) << 3 | (px & 7))];
}

bool transport_is_gather() {            // the environment's wish; "reduce" unless it says gather
    const char* e = std::getenv("RZ_GROUP_TRANSPORT");
    return e && std::strcmp(e, "gather") == 0;
}

void destroy_members(rz_group* g) {
    for (size_t i = 0; i < g->comm.size(); ++i)
        if (g->comm[i]) { (void)hipSetDevice(g->device[i]); (void)g_rccl.CommDestroy(g->comm[i]); }
    if (g->frame && g->frameLocal >= 0) { (void)hipSetDevice(g->device[g->frameLocal]); (void)hipFree(g->frame); }
    if (g->gathered && g->gatheredLocal >= 0) { (void)hipSetDevice(g->device[g->gatheredLocal]); (void)hipFree(g->gathered); }
    for (size_t i = 0; i < g->packed.size(); ++i)
        if (g->packed[i]) { (void)hipSetDevice(g->device[i]); (void)hipFree(g->packed[i]); }
    for (size_t i = 0; i < g->evSent.size(); ++i)
        if (g->evSent[i]) { (void)hipSetDevice(g->device[i]); (void)hipEventDestroy(g->evSent[i]); }
    if (g->evScattered) (void)hipEventDestroy(g->evScattered);
    for (size_t i = 0; i < g->evBefore.size(); ++i) {
        (void)hipSetDevice(g->device[i]);
        if (g->evBefore[i]) (void)hipEventDestroy(g->evBefore[i]);
        if (g->evAfter[i]) (void)hipEventDestroy(g->evAfter[i]);
    }
    for (rz_ctx* c : g->ctx) rz_destroy(c);
}

// The tile gather of one frame (see the head of this file): pack on every local member, move, scatter on the root.
// Everything is enqueued on the members' render streams, behind their kernels; nothing waits on the host.
int gather_tiles(rz_group* g, int root, int rl) {
    const int W = g->width, H = g->height, N = g->nranks;
    const int tilesX = (W + 7) / 8, tilesY = (H + 7) / 8, nTiles = tilesX * tilesY;
    const int perRank = (nTiles + N - 1) / N;
    const size_t setFloats = (size_t)perRank * 64 * 4, setBytes = setFloats * sizeof(float);
    if (g->packed.size() != g->ctx.size()) {
        g->packed.assign(g->ctx.size(), nullptr);
        g->packedBytes.assign(g->ctx.size(), 0);
        g->evSent.assign(g->ctx.size(), nullptr);
    }
    if (rl >= 0 && (g->gatheredLocal != rl || g->gatheredBytes < setBytes * N)) {
        if (g->gathered) { RZG_HIP(g, hipSetDevice(g->device[g->gatheredLocal])); (void)hipFree(g->gathered); g->gathered = nullptr; g->gatheredBytes = 0; }
        RZG_HIP(g, hipSetDevice(g->device[rl]));
        RZG_HIP(g, hipMalloc(&g->gathered, setBytes * N));
        g->gatheredBytes = setBytes * N;
        g->gatheredLocal = rl;
    }
    // (1) pack: the root member straight into its place among the gathered sets, the others into a buffer of their own
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        RZG_HIP(g, hipSetDevice(g->device[i]));
        hipStream_t s = static_cast<hipStream_t>(rz_stream_handle(g->ctx[i]));
        void* dst;
        if ((int)i == rl) {
            dst = static_cast<char*>(g->gathered) + setBytes * (size_t)root;
        } else {
            if (g->packedBytes[i] < setBytes) {
                if (g->packed[i]) { (void)hipFree(g->packed[i]); g->packed[i] = nullptr; g->packedBytes[i] = 0; }
                RZG_HIP(g, hipMalloc(&g->packed[i], setBytes));
                g->packedBytes[i] = setBytes;
            }
            dst = g->packed[i];
        }
        const float4* accum = static_cast<const float4*>(rz_accum_device_ptr(g->ctx[i]));
        if (!accum) return member_fail(g, (int)i, RZ_ERR_NOT_READY, "rz_accum_device_ptr");
        const int threads = perRank * 64;
        rz_pack_tiles<<<(threads + 255) / 256, 256, 0, s>>>(accum, static_cast<float4*>(dst), W, H, tilesX, nTiles, g->rank[i], N, perRank);
        RZG_HIP(g, hipGetLastError());
    }
    // (2) move: every other member's set to its place on the root
    if (g->loopback) {
        for (size_t i = 0; i < g->ctx.size(); ++i) {
            if ((int)i == rl) continue;
            RZG_HIP(g, hipSetDevice(g->device[i]));
            hipStream_t s = static_cast<hipStream_t>(rz_stream_handle(g->ctx[i]));
            if (!g->evSent[i]) RZG_HIP(g, hipEventCreateWithFlags(&g->evSent[i], hipEventDisableTiming));
            if (g->evScattered) RZG_HIP(g, hipStreamWaitEvent(s, g->evScattered, 0));        // (the root is through with the last frame's sets)
            RZG_HIP(g, hipMemcpyAsync(static_cast<char*>(g->gathered) + setBytes * (size_t)g->rank[i], g->packed[i], setBytes, hipMemcpyDeviceToDevice, s));
            RZG_HIP(g, hipEventRecord(g->evSent[i], s));
        }
        RZG_HIP(g, hipSetDevice(g->device[rl]));
        for (size_t i = 0; i < g->ctx.size(); ++i)
            if ((int)i != rl) RZG_HIP(g, hipStreamWaitEvent(static_cast<hipStream_t>(rz_stream_handle(g->ctx[rl])), g->evSent[i], 0));
    } else if (N > 1) {
        // one group of point-to-point calls: a send per non-root member, N - 1 receives on the root (local members included:
        // a one-process group's communicators talk to one another like any others)
        RZG_NCCL(g, g_rccl.GroupStart());
        for (size_t i = 0; i < g->ctx.size(); ++i) {
            hipError_t e = hipSetDevice(g->device[i]);
            hipStream_t s = static_cast<hipStream_t>(rz_stream_handle(g->ctx[i]));
            ncclResult_t r = e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
            if ((int)i == rl) {
                for (int peer = 0; peer < N && r == ncclSuccess; ++peer)
                    if (peer != root)
                        r = g_rccl.Recv(static_cast<char*>(g->gathered) + setBytes * (size_t)peer, setFloats, ncclFloat, peer, g->comm[i], s);
            } else if (r == ncclSuccess) {
                r = g_rccl.Send(g->packed[i], setFloats, ncclFloat, root, g->comm[i], s);
            }
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return gfail(&g->err, RZ_ERR_HIP, "ncclSend / ncclRecv on rank %d: %s", g->rank[i], g_rccl.GetErrorString(r));
            }
        }
        RZG_NCCL(g, g_rccl.GroupEnd());
    }
    // (3) scatter on the root: every pixel of the frame from its owner's set
    if (rl >= 0) {
        RZG_HIP(g, hipSetDevice(g->device[rl]));
        hipStream_t s = static_cast<hipStream_t>(rz_stream_handle(g->ctx[rl]));
        rz_unpack_tiles<<<dim3((W + 127) / 128, H), 128, 0, s>>>(static_cast<const float4*>(g->gathered), static_cast<float4*>(g->frame), W, H, tilesX, N, perRank);
        RZG_HIP(g, hipGetLastError());
        if (g->loopback) {
            if (!g->evScattered) RZG_HIP(g, hipEventCreateWithFlags(&g->evScattered, hipEventDisableTiming));
            RZG_HIP(g, hipEventRecord(g->evScattered, s));
        }
    }
    return RZ_OK;
}

}  // namespace

extern "C" {

int rz_group_rccl_version(int* version) {
    try {
        int rc = bind_rccl(nullptr);
        if (rc != RZ_OK) return rc;
        int v = 0;
        if (g_rccl.GetVersion(&v) != ncclSuccess) return gfail(nullptr, RZ_ERR_HIP, "ncclGetVersion failed");
        if (version) *version = v;
        return RZ_OK;
    } catch (...) { return gfail(nullptr, RZ_ERR_NO_MEMORY, "rz_group_rccl_version: out of host memory"); }
}

int rz_group_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == RZ_GROUP_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id128) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null id buffer");
    try {
        int rc = bind_rccl(nullptr);
        if (rc != RZ_OK) return rc;
        ncclUniqueId id;
        ncclResult_t r = g_rccl.GetUniqueId(&id);
        if (r != ncclSuccess) return gfail(nullptr, RZ_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
        std::memcpy(id128, &id, sizeof id);
        return RZ_OK;
    } catch (...) { return gfail(nullptr, RZ_ERR_NO_MEMORY, "rz_group_unique_id: out of host memory"); }
}

const char* rz_group_last_error(const rz_group* g) { return g ? g->err.c_str() : g_group_error.c_str(); }

rz_group* rz_group_create(int ndev, const int* devices, unsigned flags) {
    rz_group* g = nullptr;
    try {
        if (ndev <= 0 || ndev > 64) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: ndev %d", ndev); return nullptr; }
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { gfail(nullptr, RZ_ERR_NO_DEVICE, "no HIP device"); return nullptr; }
        const bool loopback = (flags & RZ_GROUP_LOOPBACK) != 0u;
        flags &= ~RZ_GROUP_LOOPBACK;
        if (loopback && !devices) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: a loopback group names its devices"); return nullptr; }
        for (int i = 0; i < ndev; ++i) {
            const int d = devices ? devices[i] : i;
            if (d < 0 || d >= have) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: device %d of %d", d, have); return nullptr; }
            for (int j = 0; j < i && !loopback; ++j)
                if ((devices ? devices[j] : j) == d) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: device %d listed twice (one rank per device)", d); return nullptr; }
        }
        if (!loopback && bind_rccl(nullptr) != RZ_OK) return nullptr;
        g = new rz_group();
        g->nranks = ndev;
        g->loopback = loopback;
        g->gather = loopback || (transport_is_gather() && g_rccl.Send && g_rccl.Recv);
        if (loopback) {
            g->distinctDevices = true;
            for (int i = 0; i < ndev; ++i)
                for (int j = 0; j < i; ++j)
                    if (devices[i] == devices[j]) g->distinctDevices = false;
        }
        for (int i = 0; i < ndev; ++i) {
            const int d = devices ? devices[i] : i;
            rz_ctx* c = rz_create(d, flags);
            if (!c) { gfail(nullptr, RZ_ERR_HIP, "rz_group_create: device %d: %s", d, rz_last_error(nullptr)); destroy_members(g); delete g; return nullptr; }
            g->ctx.push_back(c); g->rank.push_back(i); g->device.push_back(d); g->comm.push_back(nullptr);
        }
        if (loopback) {                 // (no communicator: device copies stand in for the links)
            // members on devices of their own: let the copies go straight over the links where the runtime allows it
            // (failure is fine -- the copy is then staged; "already enabled" is fine too)
            if (g->distinctDevices)
                for (int i = 0; i < ndev; ++i)
                    for (int j = 0; j < ndev; ++j)
                        if (i != j && hipSetDevice(g->device[i]) == hipSuccess) { (void)hipDeviceEnablePeerAccess(g->device[j], 0); (void)hipGetLastError(); }
            return g;
        }
        ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), ndev, g->device.data());
        if (r != ncclSuccess) {
            gfail(nullptr, RZ_ERR_HIP, "ncclCommInitAll(%d): %s", ndev, g_rccl.GetErrorString(r));
            for (auto& c : g->comm) c = nullptr;
            destroy_members(g); delete g;
            return nullptr;
        }
        return g;
    } catch (...) {
        gfail(nullptr, RZ_ERR_NO_MEMORY, "rz_group_create: out of host memory");
        if (g) { destroy_members(g); delete g; }
        return nullptr;
    }
}

rz_group* rz_group_create_rank(int device, int rank, int nranks, const void* id128, unsigned flags) {
    rz_group* g = nullptr;
    try {
        if (nranks <= 0 || rank < 0 || rank >= nranks) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create_rank: rank %d of %d", rank, nranks); return nullptr; }
        if (!id128) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create_rank: null id"); return nullptr; }
        if (flags & RZ_GROUP_LOOPBACK) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create_rank: RZ_GROUP_LOOPBACK is for rz_group_create"); return nullptr; }
        if (bind_rccl(nullptr) != RZ_OK) return nullptr;
        rz_ctx* c = rz_create(device, flags);
        if (!c) { gfail(nullptr, RZ_ERR_HIP, "rz_group_create_rank: device %d: %s", device, rz_last_error(nullptr)); return nullptr; }
        g = new rz_group();
        g->nranks = nranks;
        g->gather = transport_is_gather() && g_rccl.Send && g_rccl.Recv;
        g->ctx.push_back(c); g->rank.push_back(rank); g->device.push_back(device); g->comm.push_back(nullptr);
        ncclUniqueId id;
        std::memcpy(&id, id128, sizeof id);
        ncclResult_t r = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&g->comm[0], nranks, id, rank) : ncclUnhandledCudaError;
        if (r != ncclSuccess) {
            gfail(nullptr, RZ_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, nranks, device, g_rccl.GetErrorString(r));
            g->comm[0] = nullptr;
            destroy_members(g); delete g;
            return nullptr;
        }
        return g;
    } catch (...) {
        gfail(nullptr, RZ_ERR_NO_MEMORY, "rz_group_create_rank: out of host memory");
        if (g) { destroy_members(g); delete g; }
        return nullptr;
    }
}

void rz_group_destroy(rz_group* g) {
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) (void)rz_sync(g->ctx[i]);
    destroy_members(g);
    delete g;
}

int rz_group_size(const rz_group* g) { return g ? g->nranks : 0; }
const char* rz_group_transport(const rz_group* g) {
    if (!g) return "";
    try {
        std::string& t = const_cast<rz_group*>(g)->transportText;
        if (g->gather) t = !g->loopback ? "tile-gather(rccl send/recv)" : (g->distinctDevices ? "tile-gather(device copies between the members' GPUs, no RCCL)" : "tile-gather(loopback copies)");
        else t = g->fallback.empty() ? "rccl-reduce" : "rccl-reduce(fallback: " + g->fallback + ")";
        return t.c_str();
    } catch (...) { return g->gather ? "tile-gather" : "rccl-reduce"; }
}
// Which exchange step rz_group_reduce uses from now on: "reduce" or "gather".  EVERY rank of the group must make the same
// call (a launcher agrees on it first); a loopback group has no communicator and always gathers.
int rz_group_set_transport(rz_group* g, const char* name) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    if (!name) return gfail(&g->err, RZ_ERR_INVALID_ARG, "rz_group_set_transport: null name");
    const bool wantGather = std::strcmp(name, "gather") == 0;
    if (!wantGather && std::strcmp(name, "reduce") != 0) return gfail(&g->err, RZ_ERR_INVALID_ARG, "rz_group_set_transport: \"%s\" (reduce | gather)", name);
    if (g->loopback) return wantGather ? RZ_OK : gfail(&g->err, RZ_ERR_INVALID_ARG, "a loopback group has no communicator to reduce over");
    if (wantGather && !(g_rccl.Send && g_rccl.Recv)) return gfail(&g->err, RZ_ERR_NOT_READY, "the bound RCCL (%s) has no ncclSend / ncclRecv", g_rccl.path.c_str());
    try { g->fallback.clear(); } catch (...) { }
    g->gather = wantGather;
    return RZ_OK;
}
int rz_group_local_count(const rz_group* g) { return g ? (int)g->ctx.size() : 0; }
int rz_group_rank(const rz_group* g, int local) { return (g && local >= 0 && local < (int)g->rank.size()) ? g->rank[local] : -1; }
rz_ctx* rz_group_ctx(rz_group* g, int local) { return (g && local >= 0 && local < (int)g->ctx.size()) ? g->ctx[local] : nullptr; }

int rz_group_upload(rz_group* g, rz_binding binding, const void* data, size_t bytes) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = rz_upload(g->ctx[i], binding, data, bytes);
        if (rc != RZ_OK) return member_fail(g, (int)i, rc, "rz_upload");
    }
    return RZ_OK;
}

int rz_group_update(rz_group* g, rz_binding binding, size_t offset, const void* data, size_t bytes) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = rz_update(g->ctx[i], binding, offset, data, bytes);
        if (rc != RZ_OK) return member_fail(g, (int)i, rc, "rz_update");
    }
    return RZ_OK;
}

int rz_group_set_frame(rz_group* g, const rz_frame_params* params) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    if (!params) return gfail(&g->err, RZ_ERR_INVALID_ARG, "null params");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        rz_frame_params p = *params;
        p.tile_rank = g->rank[i];
        p.tile_nranks = g->nranks;
        const int rc = rz_set_frame(g->ctx[i], &p);
        if (rc != RZ_OK) return member_fail(g, (int)i, rc, "rz_set_frame");
    }
    g->width = params->width; g->height = params->height;
    g->haveFrame = true;
    return RZ_OK;
}

int rz_group_render(rz_group* g) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = rz_render(g->ctx[i]);
        if (rc != RZ_OK) return member_fail(g, (int)i, rc, "rz_render");
    }
    return RZ_OK;
}

int rz_group_reduce(rz_group* g, int root) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    if (root < 0 || root >= g->nranks) return gfail(&g->err, RZ_ERR_INVALID_ARG, "rz_group_reduce: root %d of %d", root, g->nranks);
    if (!g->haveFrame) return gfail(&g->err, RZ_ERR_NOT_READY, "rz_group_set_frame has not been called");
    const size_t count = (size_t)g->width * g->height * 4;
    const int rl = local_of_rank(g, root);
    if (rl >= 0 && (g->frameLocal != rl || g->frameBytes < count * 4)) {
        if (g->frame) { RZG_HIP(g, hipSetDevice(g->device[g->frameLocal])); (void)hipFree(g->frame); g->frame = nullptr; g->frameBytes = 0; }
        RZG_HIP(g, hipSetDevice(g->device[rl]));
        RZG_HIP(g, hipMalloc(&g->frame, count * 4));
        g->frameBytes = count * 4;
        g->frameLocal = rl;
    }
    // events around each member's share (created on first use, on the member's device)
    if (g->evBefore.size() != g->ctx.size()) {
        g->evBefore.assign(g->ctx.size(), nullptr);
        g->evAfter.assign(g->ctx.size(), nullptr);
        for (size_t i = 0; i < g->ctx.size(); ++i) {
            RZG_HIP(g, hipSetDevice(g->device[i]));
            RZG_HIP(g, hipEventCreate(&g->evBefore[i]));
            RZG_HIP(g, hipEventCreate(&g->evAfter[i]));
        }
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        RZG_HIP(g, hipSetDevice(g->device[i]));
        RZG_HIP(g, hipEventRecord(g->evBefore[i], static_cast<hipStream_t>(rz_stream_handle(g->ctx[i]))));
    }
    if (g->gather) {
        const int rc = gather_tiles(g, root, rl);
        if (rc != RZ_OK) {
            if (g->loopback) return rc;
            // The gather could not be enqueued (an RCCL error from the send / receive group, a failed allocation): this group
            // reduces from now on, and THIS frame lands through the reduce below -- same bits.  (Other processes of a rank-mode
            // group do not see this rank's error: they would wait in their receive.  An enqueue failure of RCCL's is, in
            // practice, a property of the build or the topology and hits every rank alike; the launcher should still agree
            // on the transport up front.)
            try { g->fallback = "the tile gather failed: " + g->err; } catch (...) { }
            g->gather = false;
        }
    }
    if (!g->gather) {
        // one collective per member, on the stream its render kernel was enqueued on; grouped so that one process
        // driving several devices cannot deadlock on launch order
        RZG_NCCL(g, g_rccl.GroupStart());
        for (size_t i = 0; i < g->ctx.size(); ++i) {
            const void* send = rz_accum_device_ptr(g->ctx[i]);
            void* recv = ((int)i == rl) ? g->frame : const_cast<void*>(send);      // recvbuff is only read on the root
            hipError_t e = hipSetDevice(g->device[i]);
            ncclResult_t r = e == hipSuccess ? g_rccl.Reduce(send, recv, count, ncclFloat, ncclSum, root, g->comm[i],
                                                             static_cast<hipStream_t>(rz_stream_handle(g->ctx[i])))
                                             : ncclUnhandledCudaError;
            if (r != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return gfail(&g->err, RZ_ERR_HIP, "ncclReduce on rank %d: %s", g->rank[i], g_rccl.GetErrorString(r));
            }
        }
        RZG_NCCL(g, g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        RZG_HIP(g, hipSetDevice(g->device[i]));
        RZG_HIP(g, hipEventRecord(g->evAfter[i], static_cast<hipStream_t>(rz_stream_handle(g->ctx[i]))));
    }
    g->reduceTimed = true;
    g->lastRoot = root;
    return RZ_OK;
}

int rz_group_last_reduce_ms(rz_group* g, float* root_ms, float* max_ms) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    if (!g->reduceTimed) return gfail(&g->err, RZ_ERR_NOT_READY, "rz_group_reduce has not been called");
    float worst = 0.0f, atRoot = -1.0f;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        RZG_HIP(g, hipSetDevice(g->device[i]));
        RZG_HIP(g, hipEventSynchronize(g->evAfter[i]));
        float t = 0.0f;
        RZG_HIP(g, hipEventElapsedTime(&t, g->evBefore[i], g->evAfter[i]));
        worst = t > worst ? t : worst;
        if (g->rank[i] == g->lastRoot) atRoot = t;
    }
    if (root_ms) *root_ms = atRoot;
    if (max_ms) *max_ms = worst;
    return RZ_OK;
}

int rz_group_sync(rz_group* g) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int rc = rz_sync(g->ctx[i]);
        if (rc != RZ_OK) return member_fail(g, (int)i, rc, "rz_sync");
    }
    return RZ_OK;
}

void* rz_group_frame_device_ptr(rz_group* g) {
    if (!g || g->lastRoot < 0) return nullptr;
    return local_of_rank(g, g->lastRoot) == g->frameLocal ? g->frame : nullptr;
}

int rz_group_read_frame(rz_group* g, float* rgba, size_t bytes) {
    if (!g) return gfail(nullptr, RZ_ERR_INVALID_ARG, "null group");
    const int rl = g->lastRoot >= 0 ? local_of_rank(g, g->lastRoot) : -1;
    if (rl < 0 || rl != g->frameLocal || !g->frame)
        return gfail(&g->err, RZ_ERR_NOT_READY, "the reduced frame lives on rank %d, which this process does not own", g->lastRoot);
    const size_t need = (size_t)g->width * g->height * 16;
    if (!rgba || bytes < need) return gfail(&g->err, RZ_ERR_BUFFER_SIZE, "rz_group_read_frame needs %zu bytes, got %zu", need, bytes);
    RZG_HIP(g, hipSetDevice(g->device[rl]));
    hipStream_t s = static_cast<hipStream_t>(rz_stream_handle(g->ctx[rl]));
    RZG_HIP(g, hipMemcpyAsync(rgba, g->frame, need, hipMemcpyDeviceToHost, s));
    RZG_HIP(g, hipStreamSynchronize(s));
    return RZ_OK;
}

}  // extern "C"
