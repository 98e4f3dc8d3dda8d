// Renderer.h -- the frontend-side glue: what RayZen's main.cpp does with
// OpenGL, done with the C-ABI of include/rayzen_hip.h instead.  Header-only;
// link the program with librayzen_host.so and librayzen_hip.so.
//
//   initializeSSBOs(scene)             <- main.cpp:897-1120  (build + 8x glBufferData)
//   updateDynamicBVHAndSSBOs(scene)    <- main.cpp:1123-1208 (TLAS rebuild + glBufferSubData)
//   sendSceneDataToShader(scene, ...)  <- main.cpp:1356-1392 (uniforms)
//   draw() / finish()                  <- main.cpp:637 glDrawArrays / :1347 glFinish
// Unlike the reference's per-frame path, updateDynamicBVHAndSSBOs re-uploads
// only what changed (instances + TLAS, a few KB), not all geometry.
#pragma once
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "RayZenScene.h"
#include "rayzen_hip.h"

namespace rayzen {

class Renderer {
public:
    explicit Renderer(int device = 0, unsigned flags = RZ_FLAG_NONE) : ctx_(rz_create(device, flags)) {
        if (!ctx_) throw std::runtime_error(std::string("rz_create: ") + rz_last_error(nullptr));
    }
    ~Renderer() { rz_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    // Build every BLAS on the device (rz_build_blas: RayZen's full-sweep SAH, same bytes as BVH::buildBLAS).
    void useDeviceBlasBuilder(bool on = true) {
        if (!on) { buffers_.blasBuilder = nullptr; return; }
        buffers_.blasBuilder = [this](const Mesh& mesh, BVH& out) {
            const size_t n = mesh.triangles.size();
            out.nodes.assign(n ? 2 * n - 1 : 1, BVHNode{});
            out.triIndices.assign(n, 0);
            size_t nn = 0;
            if (rz_build_blas(ctx_, reinterpret_cast<const rz_triangle*>(mesh.triangles.data()), n,
                              reinterpret_cast<rz_bvh_node*>(out.nodes.data()), out.nodes.size(), out.triIndices.data(), &nn,
                              nullptr, nullptr) != RZ_OK) return false;
            out.nodes.resize(nn);
            return true;
        };
    }

    void initializeSSBOs(const Scene& scene, bool shareMeshes = false) {
        if (!buffers_.build(scene, shareMeshes)) throw std::runtime_error(std::string("rz_build_blas: ") + rz_last_error(ctx_));
        up(RZ_BIND_TRIANGLES, buffers_.allTriangles);
        up(RZ_BIND_MATERIALS, scene.materials);
        up(RZ_BIND_LIGHTS, scene.lights);
        up(RZ_BIND_TLAS_NODES, buffers_.tlasNodes);
        up(RZ_BIND_TLAS_INDICES, buffers_.tlasTriIndices);
        up(RZ_BIND_BLAS_NODES, buffers_.allBLASNodes);
        up(RZ_BIND_BLAS_INDICES, buffers_.allBLASTriIndices);
        up(RZ_BIND_INSTANCES, buffers_.meshInstances);
    }
    // initializeSSBOs with the geometry half on the GPU (rz_build_geometry): one BLAS per distinct Mesh is built on the
    // device and stays there as bindings 7 / 8; only instances, the TLAS, materials and lights are assembled here and
    // uploaded.  Frames are the same bits as after initializeSSBOs(scene, /*shareMeshes=*/true).
    void initializeSSBOsOnDevice(const Scene& scene) {
        static const Mesh kEmpty;
        std::map<const Mesh*, size_t> index;
        std::vector<rz_mesh_build> built;
        SceneBuffers b;
        std::vector<size_t> meshOf(scene.gameObjects.size());
        for (size_t i = 0; i < scene.gameObjects.size(); ++i) {
            const Mesh* mesh = scene.gameObjects[i].mesh ? scene.gameObjects[i].mesh.get() : &kEmpty;
            auto it = index.find(mesh);
            if (it == index.end()) {
                it = index.emplace(mesh, built.size()).first;
                rz_mesh_build m{};
                m.first_triangle = b.allTriangles.size(); m.n_triangles = mesh->triangles.size();
                built.push_back(m);
                b.allTriangles.insert(b.allTriangles.end(), mesh->triangles.begin(), mesh->triangles.end());
            }
            meshOf[i] = it->second;
        }
        check(rz_build_geometry(ctx_, reinterpret_cast<const rz_triangle*>(b.allTriangles.data()), b.allTriangles.size(), built.data(),
                                built.size()), "rz_build_geometry");
        std::vector<BVHNode> worldRootNodes;
        for (size_t i = 0; i < scene.gameObjects.size(); ++i) {
            const rz_mesh_build& m = built[meshOf[i]];
            BVHNode root;
            std::memcpy(static_cast<void*>(&root), &m.root, sizeof root);
            worldRootNodes.push_back(worldRootNode(root, scene.gameObjects[i].transform));      // main.cpp:974-993
            b.blasRoots.push_back(root);
            BVHInstance inst;
            inst.blasNodeOffset = m.node_offset; inst.blasTriOffset = m.index_offset;
            inst.globalTriOffset = (int)m.first_triangle; inst.meshIndex = (int)i;
            inst.transform = scene.gameObjects[i].transform;
            inst.inverseTransform = inverse(scene.gameObjects[i].transform);
            b.meshInstances.push_back(inst);
            b.maxBLASDepth = std::max(b.maxBLASDepth, (int)m.depth);
        }
        BVH tlas;
        tlas.buildTLAS(b.meshInstances, worldRootNodes);
        b.tlasNodes = tlas.nodes; b.tlasTriIndices = tlas.triIndices; b.tlasDepth = tlas.depth();
        b.blasBuilder = buffers_.blasBuilder;
        buffers_ = std::move(b);
        up(RZ_BIND_MATERIALS, scene.materials);
        up(RZ_BIND_LIGHTS, scene.lights);
        up(RZ_BIND_TLAS_NODES, buffers_.tlasNodes);
        up(RZ_BIND_TLAS_INDICES, buffers_.tlasTriIndices);
        up(RZ_BIND_INSTANCES, buffers_.meshInstances);
    }
    void updateDynamicBVHAndSSBOs(const Scene& scene) {
        buffers_.updateDynamic(scene);
        // the TLAS keeps its node count (2*I-1) and index count (I) for a fixed instance count
        upd(RZ_BIND_INSTANCES, buffers_.meshInstances);
        upd(RZ_BIND_TLAS_NODES, buffers_.tlasNodes);
        upd(RZ_BIND_TLAS_INDICES, buffers_.tlasTriIndices);
    }
    // The same per-frame step with the work done on the GPU: only the transforms (64 B per object) are handed over;
    // inverse, world AABBs and the TLAS rebuild run in the library (rz_update_transforms).
    void updateDynamicBVHAndSSBOsOnDevice(const Scene& scene) {
        std::vector<float> xf(scene.gameObjects.size() * 16);
        for (size_t i = 0; i < scene.gameObjects.size(); ++i) std::memcpy(&xf[i * 16], scene.gameObjects[i].transform.m, 64);
        check(rz_update_transforms(ctx_, xf.data(), scene.gameObjects.size()), "rz_update_transforms");
    }
    void sendSceneDataToShader(const Scene& scene, int width, int height, int bounceBudget, int spp = 1,
                               int sampleBase = 0, int tileRank = 0, int tileNRanks = 1) {
        rz_frame_params p{};
        p.width = width; p.height = height;
        mat4 iv = inverse(scene.camera.viewMatrix), ip = inverse(scene.camera.projectionMatrix);
        std::memcpy(p.inv_view, iv.m, 64); std::memcpy(p.inv_proj, ip.m, 64);
        std::memcpy(p.view, scene.camera.viewMatrix.m, 64); std::memcpy(p.proj, scene.camera.projectionMatrix.m, 64);
        p.cam_pos[0] = scene.camera.position.x; p.cam_pos[1] = scene.camera.position.y; p.cam_pos[2] = scene.camera.position.z;
        p.num_lights = (int)scene.lights.size();
        p.bounce_budget = bounceBudget; p.spp = spp; p.sample_base = sampleBase;
        p.tile_rank = tileRank; p.tile_nranks = tileNRanks;
        check(rz_set_frame(ctx_, &p), "rz_set_frame");
        width_ = width; height_ = height;
    }
    void draw() { check(rz_render(ctx_), "rz_render"); }
    void finish() { check(rz_sync(ctx_), "rz_sync"); }
    std::vector<float> readAccum() {
        std::vector<float> out((size_t)width_ * height_ * 4);
        check(rz_read_accum(ctx_, out.data(), out.size() * sizeof(float)), "rz_read_accum");
        return out;
    }
    std::vector<uint8_t> resolveRGBA8() {
        std::vector<uint8_t> out((size_t)width_ * height_ * 4);
        check(rz_resolve_rgba8(ctx_, out.data(), out.size()), "rz_resolve_rgba8");
        return out;
    }
    float lastRenderMs() { float ms = 0; int n = 0; check(rz_last_render_ms(ctx_, &ms, &n), "rz_last_render_ms"); return ms; }
    rz_ctx* context() { return ctx_; }
    const SceneBuffers& buffers() const { return buffers_; }

private:
    template <class T> void up(rz_binding b, const std::vector<T>& v) {
        check(rz_upload(ctx_, b, v.data(), v.size() * sizeof(T)), "rz_upload");
    }
    template <class T> void upd(rz_binding b, const std::vector<T>& v) {
        check(rz_update(ctx_, b, 0, v.data(), v.size() * sizeof(T)), "rz_update");
    }
    void check(int rc, const char* what) {
        if (rc != RZ_OK) throw std::runtime_error(std::string(what) + ": " + rz_last_error(ctx_));
    }
    rz_ctx* ctx_;
    SceneBuffers buffers_;
    int width_ = 0, height_ = 0;
};

// The same frontend glue for N GPUs of one node driven by ONE process: every device holds the whole scene, renders the
// 8x8-pixel tiles t with t % N == its rank, and one exchange step per frame (a gather of the members' tiles over RCCL) lands the image on rank 0 (rz_group_*).
// A frontend written against Renderer switches by changing the type.
class GroupRenderer {
public:
    explicit GroupRenderer(int ndev, const int* devices = nullptr, unsigned flags = RZ_FLAG_NONE)
        : g_(rz_group_create(ndev, devices, flags)) {
        if (!g_) throw std::runtime_error(std::string("rz_group_create: ") + rz_group_last_error(nullptr));
    }
    ~GroupRenderer() { rz_group_destroy(g_); }
    GroupRenderer(const GroupRenderer&) = delete;
    GroupRenderer& operator=(const GroupRenderer&) = delete;

    int size() const { return rz_group_size(g_); }
    void initializeSSBOs(const Scene& scene, bool shareMeshes = false) {
        if (!buffers_.build(scene, shareMeshes)) throw std::runtime_error("scene build failed");
        up(RZ_BIND_TRIANGLES, buffers_.allTriangles);
        up(RZ_BIND_MATERIALS, scene.materials);
        up(RZ_BIND_LIGHTS, scene.lights);
        up(RZ_BIND_TLAS_NODES, buffers_.tlasNodes);
        up(RZ_BIND_TLAS_INDICES, buffers_.tlasTriIndices);
        up(RZ_BIND_BLAS_NODES, buffers_.allBLASNodes);
        up(RZ_BIND_BLAS_INDICES, buffers_.allBLASTriIndices);
        up(RZ_BIND_INSTANCES, buffers_.meshInstances);
    }
    void updateDynamicBVHAndSSBOs(const Scene& scene) {
        buffers_.updateDynamic(scene);
        upd(RZ_BIND_INSTANCES, buffers_.meshInstances);
        upd(RZ_BIND_TLAS_NODES, buffers_.tlasNodes);
        upd(RZ_BIND_TLAS_INDICES, buffers_.tlasTriIndices);
    }
    void sendSceneDataToShader(const Scene& scene, int width, int height, int bounceBudget, int spp = 1, int sampleBase = 0) {
        rz_frame_params p{};
        p.width = width; p.height = height;
        mat4 iv = inverse(scene.camera.viewMatrix), ip = inverse(scene.camera.projectionMatrix);
        std::memcpy(p.inv_view, iv.m, 64); std::memcpy(p.inv_proj, ip.m, 64);
        std::memcpy(p.view, scene.camera.viewMatrix.m, 64); std::memcpy(p.proj, scene.camera.projectionMatrix.m, 64);
        p.cam_pos[0] = scene.camera.position.x; p.cam_pos[1] = scene.camera.position.y; p.cam_pos[2] = scene.camera.position.z;
        p.num_lights = (int)scene.lights.size();
        p.bounce_budget = bounceBudget; p.spp = spp; p.sample_base = sampleBase;
        p.tile_rank = 0; p.tile_nranks = 1;         // filled in per member by the group
        check(rz_group_set_frame(g_, &p), "rz_group_set_frame");
        width_ = width; height_ = height;
    }
    // glDrawArrays on every device, then the one collective, all asynchronous
    void draw() { check(rz_group_render(g_), "rz_group_render"); check(rz_group_reduce(g_, 0), "rz_group_reduce"); }
    void finish() { check(rz_group_sync(g_), "rz_group_sync"); }
    std::vector<float> readFrame() {
        std::vector<float> out((size_t)width_ * height_ * 4);
        check(rz_group_read_frame(g_, out.data(), out.size() * sizeof(float)), "rz_group_read_frame");
        return out;
    }
    rz_group* group() { return g_; }

private:
    template <class T> void up(rz_binding b, const std::vector<T>& v) {
        check(rz_group_upload(g_, b, v.data(), v.size() * sizeof(T)), "rz_group_upload");
    }
    template <class T> void upd(rz_binding b, const std::vector<T>& v) {
        check(rz_group_update(g_, b, 0, v.data(), v.size() * sizeof(T)), "rz_group_update");
    }
    void check(int rc, const char* what) {
        if (rc != RZ_OK) throw std::runtime_error(std::string(what) + ": " + rz_group_last_error(g_));
    }
    rz_group* g_;
    SceneBuffers buffers_;
    int width_ = 0, height_ = 0;
};

}  // namespace rayzen
