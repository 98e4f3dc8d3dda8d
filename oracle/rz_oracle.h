/*
 * rz_oracle.h -- CPU oracle for RayZen's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing in the product path (rayzen_amd/, include/) may
 * include, link or call it.
 *
 * PARITY.  The reference holds no tests, golden vectors or fixtures for this
 * path.  Its SHADER half (rz_oracle.c, rz_oracle_present.c) is PINNED against
 * the reference itself, run here: oracle/glref loads RayZen's own GLSL from
 * /root/reference and runs it on the OpenGL 4.5 implementation the image ships
 * (Mesa llvmpipe); tests/golden/glref_*.npz hold 26 of its frames + a table of
 * its sin / cos / acos, and tests/test_glref.py compares: pixel by pixel at any
 * budget with math flavour 1 (llvmpipe's three built-ins replayed bit for bit:
 * the default, and the product's), to rounding wherever no random number is
 * drawn with flavour 0 (rounds 1-4's binary64 built-ins).
 * Its HOST half (rz_oracle_bvh.c) stays UNPINNED beyond the node counts the
 * survey recorded: BVH.cpp / Mesh.cpp / main.cpp need GLM (and GLFW / GLEW),
 * neither vendored nor installed, and a stand-in is not allowed.  DESIGN.md 2.
 *
 * Plain C restatement of
 *   RayZen/shaders/fragment_shader.glsl ("FS") :188-212, 380-567, 569-663, 668-773
 *   RayZen/src/BVH.cpp :11-240            (SAH BLAS / midpoint TLAS build)
 *   RayZen/src/Mesh.cpp :6-50             (OBJ reader)
 *   RayZen/src/main.cpp :941-1035         (flatten to the six SSBO arrays)
 * with the struct layouts of RayZen/include/{Mesh,BVH,Material,Light}.h.
 * Self-contained on purpose: it does not include the product's headers.
 */
#ifndef RZ_ORACLE_H
#define RZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/Mesh.h:9-17 */
typedef struct { float v0[3], pad0, v1[3], pad1, v2[3], pad2; int32_t materialIndex; int32_t tail[3]; } rzo_triangle;
/* include/BVH.h:7-12 */
typedef struct { float bmin[3]; int32_t leftFirst; float bmax[3]; int32_t count; } rzo_node;
/* include/BVH.h:14-21 */
typedef struct { int32_t blasNodeOffset, blasTriOffset, meshIndex, globalTriOffset; float transform[16], inverseTransform[16]; } rzo_instance;
/* include/Material.h:6-18 */
typedef struct { float albedo[3], metallic, roughness, reflectivity, transparency, ior; } rzo_material;
/* include/Light.h:6-13 */
typedef struct { float posdir[4], color[3], power; } rzo_light;

typedef struct {
    const rzo_triangle* triangles;   size_t n_triangles;     /* binding 0 */
    const rzo_material* materials;   size_t n_materials;     /* binding 1 */
    const rzo_light*    lights;      size_t n_lights;        /* binding 2 */
    const rzo_node*     tlas_nodes;  size_t n_tlas_nodes;    /* binding 5 */
    const int32_t*      tlas_indices;size_t n_tlas_indices;  /* binding 6 */
    const rzo_node*     blas_nodes;  size_t n_blas_nodes;    /* binding 7 */
    const int32_t*      blas_indices;size_t n_blas_indices;  /* binding 8 */
    const rzo_instance* instances;   size_t n_instances;     /* binding 9 */
} rzo_scene;

/* the uniforms of FS:4-13,100,105 + spp / sample_base (see include/rayzen_hip.h) */
typedef struct {
    int32_t width, height;
    float inv_view[16], inv_proj[16];   /* column-major */
    float cam_pos[3];
    int32_t num_lights;
    int32_t bounce_budget;
    int32_t spp;
    int32_t sample_base;
} rzo_frame;

typedef struct {
    uint64_t samples, traversals, tlas_nodes, tlas_leaf_indices, instances,
             blas_nodes, triangles, materials, light_fetches, pixels,
             /* shading-side units (the benchmark's work model prices them): FS:720-761 executed; of them through FS:755; of
              * those with a non-zero seed (FS:193-195 really evaluated); (point, light) pairs whose BRDF term was evaluated */
             scatters, diffuse_scatters, hemi_draws, lit_lights,
             triangles_past_u;      /* hitTriangle calls that pass the |a| and u-range tests (FS:396-401) and run its second half */
} rzo_counters;

/* Render pixels [x0,x1) x [y0,y1) (row 0 = bottom row, gl_FragCoord origin)
 * into accum (width*height*4 floats, full frame; only the crop is written):
 * rgb += sum over samples of the per-sample radiance in path order (FS:709,717),
 * before FS:772's divide and FS:773's clamp; a += spp.  sample_base == 0
 * zeroes the crop first.  ior_state (width*height floats or NULL) carries
 * FS:674's currentIor across calls with sample_base > 0.
 * nthreads <= 0: one thread.  counters may be NULL.  Returns 0. */
int rzo_render(const rzo_scene* scene, const rzo_frame* frame, float* accum, float* ior_state,
               int x0, int y0, int x1, int y1, int nthreads, rzo_counters* counters);

/* How many of the last rzo_render call's threads rendered at least one pixel (work is handed out in 16-pixel chunks). */
int rzo_last_threads_busy(void);

/* Analysis hook: record, for every camera path of later rzo_render calls, the cost of each of its closest-hit queries
 * (8 uint16 per path: width*height*spp*8 entries, zero-filled by the caller).  NULL switches it off. */
void rzo_set_trace_recorder(uint16_t* buf, int width, int height, int spp);

/* One closest-hit query (FS:457-503), for known-answer tests.
 * out = {hit(0/1), t, px,py,pz, nx,ny,nz, materialIndex, instanceIdx}. */
int rzo_trace(const rzo_scene* scene, const float origin[3], const float dir[3], float out[10]);
/* FS:507-528 */
int rzo_shadow(const rzo_scene* scene, const float origin[3], const float dir[3], float maxDist, float* visibility);

/* Presentation tail of the shader (FS:772-819): resolve + BVH wireframe + light markers + FPS digits. */
typedef struct {
    int32_t width, height;
    float view[16], proj[16];            /* camera.viewMatrix / projectionMatrix, column-major */
    int32_t num_lights;                  /* uniform numLights */
    float fps;                           /* uniformFps */
    int32_t show_fps;                    /* the shader always draws it; 0 switches it off */
    int32_t show_lights;                 /* debugShowLights */
    int32_t show_bvh;                    /* debugShowBVH */
    int32_t bvh_mode;                    /* debugBVHMode: 0 TLAS leaves + BLAS roots, 1 branch to a triangle */
    int32_t selected_blas, selected_tri; /* debugSelectedBLAS / debugSelectedTri */
} rzo_present_params;
/* accum: width*height*4 floats (rgb sums, a = sample count).  rgb_out (3 floats per pixel) and rgba8_out may be NULL. */
int rzo_present(const rzo_scene* scene, const rzo_present_params* pp, const float* accum, float* rgb_out,
                unsigned char* rgba8_out);

/* Which sin / cos / acos the oracle evaluates (rz_oracle_math.h): 1 = Mesa llvmpipe's (the default, and what the product is
 * compiled with: rz_math_flavour()), with which frames of RayZen's own shader rendered by oracle/glref agree pixel by pixel at any
 * bounce budget; 0 = rounds 1-4's binary64, correctly rounded ones (the product's -DRZ_MATH_FLAVOUR=0 build).  Process-wide; set
 * it before rzo_render, not during one. */
void rzo_set_math_flavour(int flavour);
int rzo_get_math_flavour(void);

/* The built-ins of the current flavour, exported for the math tests. */
float rzo_sin_f(float x);
float rzo_cos_f(float x);
float rzo_acos_f(float x);
float rzo_rand_f(float x, float y);                       /* FS:188-190 */
void  rzo_hemisphere_f(const float n[3], const float seed[2], float out[3]); /* FS:192-202 */

/* ---- host half: literal restatement of BVH.cpp / Mesh.cpp / main.cpp flatten ---- */

/* BVH.cpp:99-175 (SAH).  nodes_out: capacity 2*n+1 nodes; idx_out: n ints.
 * Returns the node count (>= 1). */
int rzo_build_blas(const rzo_triangle* tris, int n, rzo_node* nodes_out, int32_t* idx_out);
/* BVH.cpp:178-240.  roots[i] = world-space AABB of instance i (only bmin/bmax read).
 * nodes_out capacity 2*n, idx_out capacity n.  Returns node count; *n_idx_out = index count. */
int rzo_build_tlas(const rzo_node* roots, int n, rzo_node* nodes_out, int32_t* idx_out, int* n_idx_out);
/* main.cpp:974-993: world AABB of a BLAS root box under a column-major transform. */
void rzo_world_bounds(const rzo_node* root, const float transform[16], float bmin[3], float bmax[3]);
/* glm::inverse(mat4) as main.cpp:1001, 1058, 1151 call it: GLM 0.9.9.8 compute_inverse<4,4>, column-major in and out. */
void rzo_mat4_inverse(const float m[16], float out[16]);
/* Mesh.cpp:6-50.  Two-pass: tris_out NULL => returns the triangle count; else fills up to cap. -1: cannot open. */
int rzo_load_obj(const char* path, int materialIndex, rzo_triangle* tris_out, int cap);

#ifdef __cplusplus
}
#endif
#endif
