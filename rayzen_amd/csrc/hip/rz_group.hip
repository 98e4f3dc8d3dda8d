// rz_group.hip -- the multi-GPU group of include/rayzen_hip.h: N contexts, tile-sharded, ONE RCCL reduce per frame.
//
// RayZen is single-GPU; what this replaces is its context lifetime (RayZen/src/main.cpp:228-241 create,
// :681-686 teardown) for N devices of one node.  Pixels shard by 8x8 tiles dealt round-robin (tile t -> rank t % N,
// rz_frame_params.tile_rank / tile_nranks); each member renders ALL samples of its own pixels (currentIor couples a
// pixel's samples, fragment_shader.glsl:674) into a buffer that is zero wherever it owns nothing, and one
// ncclReduce(sum) over xGMI lands the frame on the root.  The sum adds one value to zeros: bit-identical to one GPU.
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

void destroy_members(rz_group* g) {
    for (size_t i = 0; i < g->comm.size(); ++i)
        if (g->comm[i]) { (void)hipSetDevice(g->device[i]); (void)g_rccl.CommDestroy(g->comm[i]); }
    if (g->frame && g->frameLocal >= 0) { (void)hipSetDevice(g->device[g->frameLocal]); (void)hipFree(g->frame); }
    for (size_t i = 0; i < g->evBefore.size(); ++i) {
        (void)hipSetDevice(g->device[i]);
        if (g->evBefore[i]) (void)hipEventDestroy(g->evBefore[i]);
        if (g->evAfter[i]) (void)hipEventDestroy(g->evAfter[i]);
    }
    for (rz_ctx* c : g->ctx) rz_destroy(c);
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
        for (int i = 0; i < ndev; ++i) {
            const int d = devices ? devices[i] : i;
            if (d < 0 || d >= have) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: device %d of %d", d, have); return nullptr; }
            for (int j = 0; j < i; ++j)
                if ((devices ? devices[j] : j) == d) { gfail(nullptr, RZ_ERR_INVALID_ARG, "rz_group_create: device %d listed twice (one rank per device)", d); return nullptr; }
        }
        if (bind_rccl(nullptr) != RZ_OK) return nullptr;
        g = new rz_group();
        g->nranks = ndev;
        for (int i = 0; i < ndev; ++i) {
            const int d = devices ? devices[i] : i;
            rz_ctx* c = rz_create(d, flags);
            if (!c) { gfail(nullptr, RZ_ERR_HIP, "rz_group_create: device %d: %s", d, rz_last_error(nullptr)); destroy_members(g); delete g; return nullptr; }
            g->ctx.push_back(c); g->rank.push_back(i); g->device.push_back(d); g->comm.push_back(nullptr);
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
        if (bind_rccl(nullptr) != RZ_OK) return nullptr;
        rz_ctx* c = rz_create(device, flags);
        if (!c) { gfail(nullptr, RZ_ERR_HIP, "rz_group_create_rank: device %d: %s", device, rz_last_error(nullptr)); return nullptr; }
        g = new rz_group();
        g->nranks = nranks;
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
