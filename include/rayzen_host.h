/*
 * rayzen_host.h -- C view of the host-side input producers (librayzen_host.so).
 *
 * The C++ API is rayzen_amd/csrc/host/RayZenScene.h (Scene, Mesh, BVH ... with
 * RayZen's own class and member names); this header exposes the same code to
 * other languages (the Python tests and bench harness use it through ctypes).
 * Plain C, no GPU needed.  Citations are relative to /root/reference/RayZen.
 */
#ifndef RAYZEN_HOST_H
#define RAYZEN_HOST_H

#include "rayzen_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Mesh::loadFromOBJ (src/Mesh.cpp:6-50).  out == NULL: returns the triangle
 * count; otherwise fills at most cap triangles.  -1: file cannot be opened. */
int rzh_load_obj(const char* path, int materialIndex, rz_triangle* out, int cap);

/* BVH::buildBLAS (src/BVH.cpp:99-175, SAH).  nodes_out: capacity 2*n+1,
 * idx_out: capacity n.  Returns the node count; *depth_out (may be NULL) =
 * longest root-to-leaf path in nodes. */
int rzh_build_blas(const rz_triangle* tris, int n, rz_bvh_node* nodes_out, int32_t* idx_out, int* depth_out);

/* BVH::buildTLAS (src/BVH.cpp:178-240) over world-space instance boxes
 * (only boundsMin/boundsMax of world_roots are read).  nodes_out capacity
 * 2*n, idx_out capacity n.  Returns the node count. */
int rzh_build_tlas(const rz_bvh_node* world_roots, int n, rz_bvh_node* nodes_out, int32_t* idx_out, int* n_idx_out);

/* World AABB of an object-space root box under a column-major transform
 * (src/main.cpp:974-993). */
void rzh_world_bounds(const rz_bvh_node* root, const float transform[16], float bmin[3], float bmax[3]);

/* Scene -> SSBO arrays: initializeSSBOs (src/main.cpp:941-1035, no disk
 * cache) and updateDynamicBVHAndSSBOs (src/main.cpp:1138-1194). */
typedef struct rzh_scene rzh_scene;
rzh_scene* rzh_scene_create(void);
void       rzh_scene_destroy(rzh_scene* s);
/* returns the mesh id (>= 0); triangles are copied */
int  rzh_scene_add_mesh(rzh_scene* s, const rz_triangle* tris, int n);
/* a GameObject: mesh + column-major transform; returns the object id */
int  rzh_scene_add_object(rzh_scene* s, int mesh_id, const float transform[16]);
int  rzh_scene_set_transform(rzh_scene* s, int object_id, const float transform[16]);
/* share_meshes = 0: one BLAS/triangle copy per object, exactly as the
 * reference; 1: one copy per distinct mesh (true instancing). */
int  rzh_scene_build(rzh_scene* s, int share_meshes);        /* -2: the BLAS builder set below failed */
/* Replace BVH::buildBLAS inside rzh_scene_build by a function with rz_build_blas's signature (include/rayzen_hip.h;
 * ctx = its rz_ctx*): the device builder, same bytes.  fn == NULL restores the host builder.  This library stays free
 * of HIP: the caller passes the entry point. */
typedef int (*rzh_blas_builder_fn)(void* ctx, const rz_triangle* tris, size_t n, rz_bvh_node* nodes_out, size_t nodes_cap,
                                   int32_t* indices_out, size_t* n_nodes, int* depth, float* device_ms);
int  rzh_scene_set_blas_builder(rzh_scene* s, rzh_blas_builder_fn fn, void* ctx);
int  rzh_scene_update_dynamic(rzh_scene* s);
/* pointer/size of a geometry array (bindings 0, 5, 6, 7, 8, 9); valid until
 * the next build/update/destroy */
const void* rzh_scene_buffer(const rzh_scene* s, rz_binding b, size_t* bytes);
void rzh_scene_depths(const rzh_scene* s, int* max_blas_depth, int* tlas_depth);
/* RayZen's SSBO disk cache (src/main.cpp:94-115, 914-939, 1037-1043): six files `ssbo_v2_{triangles,blasnodes,
 * blastris,instances,tlasnodes,tlastris}.bin` in `dir`, each a native size_t count + raw POD array.  save: after
 * rzh_scene_build; load: replaces the scene's arrays (rzh_scene_update_dynamic then needs objects added in the same
 * order as the cached instances).  0 on success, -1 on any I/O or consistency failure. */
int rzh_scene_save_cache(const rzh_scene* s, const char* dir);
int rzh_scene_load_cache(rzh_scene* s, const char* dir);
/* initializeSSBOs with RayZen's whole disk cache, step for step (src/main.cpp:897-1060): the ssbo_v2_* set (invalidated
 * only by a changed object count, main.cpp:929-934; transforms refreshed on use, main.cpp:1054-1060), else per object
 * `mesh<i>.nodes.bin` / `mesh<i>.tris.bin` (saveBVHToFile / loadBVHFromFile, main.cpp:117-125, 956-967), then
 * `scene_tlas.nodes.bin` / `.tris.bin` + `instances.bin` (main.cpp:1012-1026, used only if every BLAS came from the cache);
 * what was built is written back and the ssbo_v2_* set rewritten.  `dir` stands for "bvh_cache/v2/" and is created if
 * missing; force_rebuild = the --rebuild-bvh flag.  One BLAS + triangle copy per object, as the reference does.
 * report (may be NULL): {ssbo set used, ssbo set invalidated, BLAS loaded, BLAS built, TLAS+instances loaded}.
 * 0 on success, -2 if the device BLAS builder failed. */
int rzh_scene_build_cached(rzh_scene* s, const char* dir, int force_rebuild, int report[5]);

/* Camera (include/Camera.h:42-48) + the inverses sendSceneDataToShader
 * uploads (src/main.cpp:1363-1364).  target is a direction. All column-major. */
void rzh_camera_matrices(const float position[3], const float target[3], const float up[3],
                         float fov_degrees, float aspect, float z_near, float z_far,
                         float view[16], float proj[16], float inv_view[16], float inv_proj[16]);
/* glm::translate/scale/rotate-style composition helpers: out = m * T|S|R */
void rzh_mat_translate(const float m[16], const float v[3], float out[16]);
void rzh_mat_scale(const float m[16], const float v[3], float out[16]);
void rzh_mat_rotate(const float m[16], float angle_radians, const float axis[3], float out[16]);
void rzh_mat_inverse(const float m[16], float out[16]);

/* Synthetic meshes (no asset files are needed):
 *  cube  -- the 12 triangles of meshes/cube.obj (vertices +-1), same order.
 *  blob  -- closed, watertight cube-sphere with 12*n*n triangles whose radius
 *           is modulated by smooth lobes ("bunny" stand-in: body + two ears);
 *           n = 76 -> 69,312 triangles, n = 289 -> 1,002,252. */
int rzh_make_cube(int materialIndex, rz_triangle* out, int cap);
int rzh_make_blob(int n, float radius, unsigned seed, int materialIndex, rz_triangle* out, int cap);

const char* rzh_version(void);

#ifdef __cplusplus
}
#endif
#endif
