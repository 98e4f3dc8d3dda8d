// capi.cpp -- extern "C" view of the host library (include/rayzen_host.h).
#include "rayzen_host.h"

#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "RayZenScene.h"
#include "SceneCache.h"

using namespace rayzen;

namespace {
mat4 toMat(const float m[16]) { mat4 r; std::memcpy(r.m, m, 64); return r; }
vec3 toVec(const float v[3]) { return vec3(v[0], v[1], v[2]); }
}  // namespace

struct rzh_scene {
    Scene scene;
    std::vector<std::shared_ptr<Mesh>> meshes;
    SceneBuffers buffers;
    bool built = false;
};

extern "C" {

const char* rzh_version(void) { return "rayzen_host 0.1"; }

int rzh_load_obj(const char* path, int materialIndex, rz_triangle* out, int cap) {
    if (!path) return -1;
    Mesh m;
    if (!m.loadFromOBJ(path, materialIndex)) return -1;
    int n = (int)m.triangles.size();
    if (out) std::memcpy(out, m.triangles.data(), sizeof(rz_triangle) * (size_t)(n < cap ? n : cap));
    return n;
}

int rzh_build_blas(const rz_triangle* tris, int n, rz_bvh_node* nodes_out, int32_t* idx_out, int* depth_out) {
    if (n < 0 || !nodes_out || (n > 0 && (!tris || !idx_out))) return -1;
    BVH bvh;
    bvh.buildBLAS(reinterpret_cast<const Triangle*>(tris), n);
    std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(BVHNode));
    if (n > 0) std::memcpy(idx_out, bvh.triIndices.data(), (size_t)n * sizeof(int));
    if (depth_out) *depth_out = bvh.depth();
    return (int)bvh.nodes.size();
}

int rzh_build_tlas(const rz_bvh_node* world_roots, int n, rz_bvh_node* nodes_out, int32_t* idx_out, int* n_idx_out) {
    if (n < 0 || !nodes_out || (n > 0 && (!world_roots || !idx_out))) return -1;
    std::vector<BVHInstance> inst((size_t)n);
    std::vector<BVHNode> roots((size_t)n);
    if (n > 0) std::memcpy(static_cast<void*>(roots.data()), world_roots, (size_t)n * sizeof(BVHNode));
    BVH tlas;
    tlas.buildTLAS(inst, roots);
    std::memcpy(nodes_out, tlas.nodes.data(), tlas.nodes.size() * sizeof(BVHNode));
    if (!tlas.triIndices.empty()) std::memcpy(idx_out, tlas.triIndices.data(), tlas.triIndices.size() * sizeof(int));
    if (n_idx_out) *n_idx_out = (int)tlas.triIndices.size();
    return (int)tlas.nodes.size();
}

void rzh_world_bounds(const rz_bvh_node* root, const float transform[16], float bmin[3], float bmax[3]) {
    BVHNode r;
    std::memcpy(static_cast<void*>(&r), root, sizeof r);
    BVHNode w = worldRootNode(r, toMat(transform));
    bmin[0] = w.boundsMin.x; bmin[1] = w.boundsMin.y; bmin[2] = w.boundsMin.z;
    bmax[0] = w.boundsMax.x; bmax[1] = w.boundsMax.y; bmax[2] = w.boundsMax.z;
}

rzh_scene* rzh_scene_create(void) { return new (std::nothrow) rzh_scene(); }
void rzh_scene_destroy(rzh_scene* s) { delete s; }

int rzh_scene_add_mesh(rzh_scene* s, const rz_triangle* tris, int n) {
    if (!s || n < 0 || (n > 0 && !tris)) return -1;
    auto m = std::make_shared<Mesh>();
    m->triangles.resize((size_t)n);
    if (n > 0) std::memcpy(static_cast<void*>(m->triangles.data()), tris, (size_t)n * sizeof(Triangle));
    s->meshes.push_back(m);
    return (int)s->meshes.size() - 1;
}

int rzh_scene_add_object(rzh_scene* s, int mesh_id, const float transform[16]) {
    if (!s || mesh_id < 0 || mesh_id >= (int)s->meshes.size() || !transform) return -1;
    GameObject o;
    o.mesh = s->meshes[(size_t)mesh_id];
    o.transform = toMat(transform);
    s->scene.gameObjects.push_back(o);
    return (int)s->scene.gameObjects.size() - 1;
}

int rzh_scene_set_transform(rzh_scene* s, int object_id, const float transform[16]) {
    if (!s || object_id < 0 || object_id >= (int)s->scene.gameObjects.size() || !transform) return -1;
    s->scene.gameObjects[(size_t)object_id].transform = toMat(transform);
    return 0;
}

int rzh_scene_build(rzh_scene* s, int share_meshes) {
    if (!s) return -1;
    s->built = s->buffers.build(s->scene, share_meshes != 0);
    return s->built ? 0 : -2;
}

int rzh_scene_set_blas_builder(rzh_scene* s, rzh_blas_builder_fn fn, void* ctx) {
    if (!s) return -1;
    if (!fn) { s->buffers.blasBuilder = nullptr; return 0; }
    s->buffers.blasBuilder = [fn, ctx](const Mesh& mesh, BVH& out) {
        const size_t n = mesh.triangles.size();
        out.nodes.assign(n ? 2 * n - 1 : 1, BVHNode{});
        out.triIndices.assign(n, 0);
        size_t nn = 0;
        const int rc = fn(ctx, reinterpret_cast<const rz_triangle*>(mesh.triangles.data()), n,
                          reinterpret_cast<rz_bvh_node*>(out.nodes.data()), out.nodes.size(), out.triIndices.data(), &nn, nullptr, nullptr);
        if (rc != 0 || nn == 0 || nn > out.nodes.size()) return false;
        out.nodes.resize(nn);
        return true;
    };
    return 0;
}

int rzh_scene_update_dynamic(rzh_scene* s) {
    if (!s || !s->built) return -1;
    s->buffers.updateDynamic(s->scene);
    return 0;
}

const void* rzh_scene_buffer(const rzh_scene* s, rz_binding b, size_t* bytes) {
    if (!s || !s->built) { if (bytes) *bytes = 0; return nullptr; }
    const SceneBuffers& B = s->buffers;
    const void* p = nullptr; size_t n = 0;
    switch (b) {
        case RZ_BIND_TRIANGLES:    p = B.allTriangles.data();      n = B.allTriangles.size() * sizeof(Triangle); break;
        case RZ_BIND_TLAS_NODES:   p = B.tlasNodes.data();         n = B.tlasNodes.size() * sizeof(BVHNode); break;
        case RZ_BIND_TLAS_INDICES: p = B.tlasTriIndices.data();    n = B.tlasTriIndices.size() * sizeof(int); break;
        case RZ_BIND_BLAS_NODES:   p = B.allBLASNodes.data();      n = B.allBLASNodes.size() * sizeof(BVHNode); break;
        case RZ_BIND_BLAS_INDICES: p = B.allBLASTriIndices.data(); n = B.allBLASTriIndices.size() * sizeof(int); break;
        case RZ_BIND_INSTANCES:    p = B.meshInstances.data();     n = B.meshInstances.size() * sizeof(BVHInstance); break;
        default: break;
    }
    if (bytes) *bytes = n;
    return p;
}

int rzh_scene_save_cache(const rzh_scene* s, const char* dir) {
    if (!s || !s->built || !dir) return -1;
    return saveSceneCache(dir, s->buffers) ? 0 : -1;
}

int rzh_scene_load_cache(rzh_scene* s, const char* dir) {
    if (!s || !dir) return -1;
    if (!loadSceneCache(dir, s->buffers)) return -1;
    s->built = true;
    return 0;
}

int rzh_scene_build_cached(rzh_scene* s, const char* dir, int force_rebuild, int report[5]) {
    if (!s || !dir) return -1;
    CacheReport rep;
    if (!initializeSSBOsCached(s->scene, dir, force_rebuild != 0, s->buffers, &rep)) { s->built = false; return -2; }
    s->built = true;
    if (report) {
        report[0] = rep.ssboLoaded; report[1] = rep.ssboInvalidated; report[2] = rep.blasLoaded; report[3] = rep.blasBuilt;
        report[4] = rep.tlasLoaded;
    }
    return 0;
}

void rzh_scene_depths(const rzh_scene* s, int* max_blas_depth, int* tlas_depth) {
    if (max_blas_depth) *max_blas_depth = s ? s->buffers.maxBLASDepth : 0;
    if (tlas_depth) *tlas_depth = s ? s->buffers.tlasDepth : 0;
}

void rzh_camera_matrices(const float position[3], const float target[3], const float up[3], float fov_degrees,
                         float aspect, float z_near, float z_far, float view[16], float proj[16],
                         float inv_view[16], float inv_proj[16]) {
    Camera cam(toVec(position), toVec(target), toVec(up), fov_degrees, aspect, z_near, z_far);
    mat4 iv = inverse(cam.viewMatrix), ip = inverse(cam.projectionMatrix);
    if (view) std::memcpy(view, cam.viewMatrix.m, 64);
    if (proj) std::memcpy(proj, cam.projectionMatrix.m, 64);
    if (inv_view) std::memcpy(inv_view, iv.m, 64);
    if (inv_proj) std::memcpy(inv_proj, ip.m, 64);
}

void rzh_mat_translate(const float m[16], const float v[3], float out[16]) { mat4 r = translate(toMat(m), toVec(v)); std::memcpy(out, r.m, 64); }
void rzh_mat_scale(const float m[16], const float v[3], float out[16]) { mat4 r = scale(toMat(m), toVec(v)); std::memcpy(out, r.m, 64); }
void rzh_mat_rotate(const float m[16], float a, const float axis[3], float out[16]) { mat4 r = rotate(toMat(m), a, toVec(axis)); std::memcpy(out, r.m, 64); }
void rzh_mat_inverse(const float m[16], float out[16]) { mat4 r = inverse(toMat(m)); std::memcpy(out, r.m, 64); }

int rzh_make_cube(int materialIndex, rz_triangle* out, int cap) {
    // geometry of meshes/cube.obj (the Blender 2.76 default cube export, including its
    // 0.999999 / 1.000001 vertices) in the file's face order
    static const float V[8][3] = {{1.0f, -1.0f, -1.0f},      {1.0f, -1.0f, 1.0f},  {-1.0f, -1.0f, 1.0f},
                                  {-1.0f, -1.0f, -1.0f},     {1.0f, 1.0f, -0.999999f},
                                  {0.999999f, 1.0f, 1.000001f}, {-1.0f, 1.0f, 1.0f}, {-1.0f, 1.0f, -1.0f}};
    static const int F[12][3] = {{2, 3, 4}, {8, 7, 6}, {5, 6, 2}, {6, 7, 3}, {3, 7, 8}, {1, 4, 8},
                                 {1, 2, 4}, {5, 8, 6}, {1, 5, 2}, {2, 6, 3}, {4, 3, 8}, {5, 1, 8}};
    if (!out) return 12;
    for (int i = 0; i < 12 && i < cap; ++i) {
        std::memset(&out[i], 0, sizeof(rz_triangle));
        std::memcpy(out[i].v0, V[F[i][0] - 1], 12);
        std::memcpy(out[i].v1, V[F[i][1] - 1], 12);
        std::memcpy(out[i].v2, V[F[i][2] - 1], 12);
        out[i].materialIndex = materialIndex;
    }
    return 12;
}

// Radius of the "bunny" stand-in in direction d (unit): an ellipsoid-ish body
// with low-frequency lobes plus two ear bumps.  Pure function of d.
static double blob_radius(double x, double y, double z, unsigned seed) {
    double ph = 0.37 * (double)(seed % 97u);
    double r = 1.0;
    r += 0.10 * std::sin(3.0 * x + ph) * std::sin(2.0 * y + 1.3) ;
    r += 0.07 * std::sin(4.0 * z + 0.7 + ph) * std::cos(3.0 * x - 0.4);
    r += 0.04 * std::sin(9.0 * y + 2.1) * std::sin(7.0 * z + ph);
    const double ears[2][3] = {{0.32, 0.90, 0.29}, {-0.32, 0.90, 0.29}};
    for (int e = 0; e < 2; ++e) {
        double c = x * ears[e][0] + y * ears[e][1] + z * ears[e][2];   // cos of the angle to the ear axis (axis ~unit)
        double a = (1.0 - c) / 0.045;
        r += 0.55 * std::exp(-a);
    }
    double head = x * 0.0 + y * 0.35 + z * 0.93;
    r += 0.22 * std::exp(-(1.0 - head) / 0.15);
    return r;
}

int rzh_make_blob(int n, float radius, unsigned seed, int materialIndex, rz_triangle* out, int cap) {
    if (n < 1) return -1;
    long long total = 12LL * n * n;
    if (!out) return (int)total;
    // vertex as a pure function of its integer position on the cube surface, so
    // faces share edge vertices bit for bit (watertight)
    auto vertex = [&](int ix, int iy, int iz, float* p) {
        double cx = 2.0 * ix / n - 1.0, cy = 2.0 * iy / n - 1.0, cz = 2.0 * iz / n - 1.0;
        // equal-area-ish warp, then normalise
        double wx = std::tan(cx * 0.78539816339744830962), wy = std::tan(cy * 0.78539816339744830962),
               wz = std::tan(cz * 0.78539816339744830962);
        double l = std::sqrt(wx * wx + wy * wy + wz * wz);
        double dx = wx / l, dy = wy / l, dz = wz / l;
        double r = (double)radius * blob_radius(dx, dy, dz, seed);
        p[0] = (float)(dx * r); p[1] = (float)(dy * r); p[2] = (float)(dz * r);
    };
    int k = 0;
    for (int face = 0; face < 6; ++face) {
        int axis = face >> 1, side = face & 1;         // the fixed coordinate and which end
        int ua = (axis + 1) % 3, va = (axis + 2) % 3;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                int c[4][3];
                const int du[4] = {0, 1, 1, 0}, dv[4] = {0, 0, 1, 1};
                for (int q = 0; q < 4; ++q) {
                    c[q][axis] = side ? n : 0;
                    c[q][ua] = i + du[q];
                    c[q][va] = j + dv[q];
                }
                float p[4][3];
                for (int q = 0; q < 4; ++q) vertex(c[q][0], c[q][1], c[q][2], p[q]);
                // outward winding: (u x v) points along +axis; flip on the low side
                const int t0[3] = {0, 1, 2}, t1[3] = {0, 2, 3};
                for (int t = 0; t < 2; ++t) {
                    const int* id = t ? t1 : t0;
                    if (k < cap) {
                        rz_triangle* tr = &out[k];
                        std::memset(tr, 0, sizeof *tr);
                        std::memcpy(tr->v0, p[id[0]], 12);
                        std::memcpy(tr->v1, p[side ? id[1] : id[2]], 12);
                        std::memcpy(tr->v2, p[side ? id[2] : id[1]], 12);
                        tr->materialIndex = materialIndex;
                    }
                    ++k;
                }
            }
    }
    return k;
}

}  // extern "C"
