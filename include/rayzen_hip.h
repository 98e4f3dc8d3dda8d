/*
 * rayzen_hip.h -- C-ABI of the MI355X (gfx950) path-tracing render loop.
 *
 * This is the drop-in boundary for RayZen's render hot path.  RayZen (the
 * reference, cited as file:line relative to /root/reference/RayZen) has no
 * FFI of its own for this path: its frontend talks to the GPU through raw
 * OpenGL calls.  Every entry point below replaces one group of those calls,
 * and every struct is byte-for-byte the element type of one of RayZen's
 * SSBOs (C++ layout == std430 layout), so a frontend hands over the very
 * same std::vector<T>::data() pointers it gives to glBufferData today.
 *
 * Plain C, plain pointers and sizes.  No C++ exception crosses this ABI:
 * every call returns RZ_OK (0) or a negative rz_status and records a message
 * readable through rz_last_error().  A context is externally synchronised
 * (one thread at a time), like the single GL context of the reference.
 */
#ifndef RAYZEN_HIP_H
#define RAYZEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* SSBO element types (the data contract).                                   */
/* ------------------------------------------------------------------------ */

/* include/Mesh.h:9-17, shaders/fragment_shader.glsl:30-38.  64 B, align 16.
 * Vertices are in OBJECT space (src/main.cpp:971-972). */
typedef struct rz_triangle {
    float v0[3]; float pad0;
    float v1[3]; float pad1;
    float v2[3]; float pad2;
    int32_t materialIndex;
    int32_t tail_pad[3];
} rz_triangle;

/* include/BVH.h:7-12, fragment_shader.glsl:40-45.  32 B.
 * internal: count == -1, children at leftFirst and leftFirst+1;
 * leaf:     count in [1,4] (0 for an empty mesh), leftFirst = first slot in
 *           the index array (src/BVH.cpp:115-118,166-171). */
typedef struct rz_bvh_node {
    float boundsMin[3]; int32_t leftFirst;
    float boundsMax[3]; int32_t count;
} rz_bvh_node;

/* include/BVH.h:14-21, fragment_shader.glsl:86-93.  144 B.
 * mat4 are column-major (GLM / GLSL convention). */
typedef struct rz_bvh_instance {
    int32_t blasNodeOffset;
    int32_t blasTriOffset;
    int32_t meshIndex;
    int32_t globalTriOffset;
    float   transform[16];
    float   inverseTransform[16];
} rz_bvh_instance;

/* include/Material.h:6-18, fragment_shader.glsl:15-22.  32 B. */
typedef struct rz_material {
    float albedo[3];
    float metallic;
    float roughness;
    float reflectivity;
    float transparency;
    float ior;
} rz_material;

/* include/Light.h:6-13, fragment_shader.glsl:24-28.  32 B.
 * positionOrDirection.w == 1 -> point light, otherwise directional. */
typedef struct rz_light {
    float positionOrDirection[4];
    float color[3];
    float power;
} rz_light;

/* Binding points: the numeric values ARE the GL SSBO binding indices of the
 * reference (fragment_shader.glsl:47-96, src/main.cpp:1072-1119).  Bindings
 * 3 and 4 are dead in the reference and do not exist here. */
typedef enum rz_binding {
    RZ_BIND_TRIANGLES    = 0,  /* rz_triangle[]      */
    RZ_BIND_MATERIALS    = 1,  /* rz_material[]      */
    RZ_BIND_LIGHTS       = 2,  /* rz_light[]         */
    RZ_BIND_TLAS_NODES   = 5,  /* rz_bvh_node[]      */
    RZ_BIND_TLAS_INDICES = 6,  /* int32[] instance ids */
    RZ_BIND_BLAS_NODES   = 7,  /* rz_bvh_node[] (all meshes concatenated) */
    RZ_BIND_BLAS_INDICES = 8,  /* int32[] mesh-local triangle ids */
    RZ_BIND_INSTANCES    = 9   /* rz_bvh_instance[]  */
} rz_binding;

typedef enum rz_status {
    RZ_OK                 =  0,
    RZ_ERR_INVALID_ARG    = -1,
    RZ_ERR_NO_DEVICE      = -2,
    RZ_ERR_HIP            = -3,  /* a HIP runtime call failed */
    RZ_ERR_OUT_OF_RANGE   = -4,  /* rz_update past the end of a binding */
    RZ_ERR_NOT_READY      = -5,  /* render before all bindings / frame set */
    RZ_ERR_BAD_SCENE      = -6,  /* uploaded arrays are inconsistent */
    RZ_ERR_BUFFER_SIZE    = -7,  /* caller buffer too small */
    RZ_ERR_NO_MEMORY      = -8,  /* a host allocation failed inside the library (std::bad_alloc never crosses the ABI) */
    RZ_ERR_INTERNAL       = -9   /* a render kernel reached one of its "cannot happen" bounds: pixels may be missing (rz_sync reports it) */
} rz_status;

/* Per-frame parameters = the uniforms of sendSceneDataToShader
 * (src/main.cpp:1356-1379; fragment_shader.glsl:4-13,100,105) plus the three
 * knobs the reference hard-codes (spp: `numSamples = 1`, fragment_shader.glsl:675)
 * or does not have (sample_base, tile sharding).  Matrices are column-major.
 * The traced radiance reads only inv_view, inv_proj and cam_pos
 * (fragment_shader.glsl:208-211,714); view/proj ride along for the
 * resolve/overlay stage. */
typedef struct rz_frame_params {
    int32_t width, height;      /* `resolution` */
    float   inv_view[16];
    float   inv_proj[16];
    float   view[16];
    float   proj[16];
    float   cam_pos[3];
    int32_t num_lights;         /* `numLights` */
    int32_t bounce_budget;      /* `uniformBounceBudget`; <= 0 means 5 (glsl:673) */
    int32_t spp;                /* samples per pixel rendered by one rz_render */
    int32_t sample_base;        /* first sample index; 0 resets per-pixel state */
    int32_t tile_rank;          /* this context renders tiles t with          */
    int32_t tile_nranks;        /*   t % tile_nranks == tile_rank (1 => all)  */
} rz_frame_params;

#define RZ_TILE_W 8             /* pixel tile = one wavefront: 8 x 8 pixels */
#define RZ_TILE_H 8

/* Counters of the REFERENCE algorithm's memory touches (what the fragment
 * shader would read from its SSBOs), used to price the roofline. Filled by
 * rz_render_counted().  bytes = 32*tlas_nodes + 4*tlas_leaf_indices +
 * 144*instances + 32*blas_nodes + 68*triangles + 32*materials +
 * 32*light_fetches + 16*pixels  (SURVEY.md section 8d). */
typedef struct rz_counters {
    uint64_t samples;           /* camera paths started                      */
    uint64_t traversals;        /* traverseTLAS calls (primary+shadow+bounce) */
    uint64_t tlas_nodes;        /* tlasNodes[] elements popped               */
    uint64_t tlas_leaf_indices; /* tlasTriIndices[] elements read            */
    uint64_t instances;         /* bvhInstances[] elements entered           */
    uint64_t blas_nodes;        /* blasNodes[] elements popped               */
    uint64_t triangles;         /* hitTriangle calls (index + triangle read) */
    uint64_t materials;         /* materials[] fetches                       */
    uint64_t light_fetches;     /* lights[] fetches                          */
    uint64_t pixels;            /* accumulation-buffer pixels written        */
    /* units of the shading side (round 4: the work model of bench.py prices them instead of bounding them) */
    uint64_t scatters;          /* FS:720-761 executed (a segment that hit)   */
    uint64_t diffuse_scatters;  /* ... of them through FS:755 (hemisphere)    */
    uint64_t hemi_draws;        /* ... of those with a non-zero seed (FS:193-195 evaluated; bounce 0 draws a constant) */
    uint64_t lit_lights;        /* (point, light) pairs whose BRDF term was evaluated (FS:589-607 / 636-659) */
    uint64_t triangles_past_u;  /* hitTriangle calls that pass FS:396-401 (|a|, u range) and run the second half */
} rz_counters;

typedef struct rz_ctx rz_ctx;

/* glfwCreateWindow/MakeContextCurrent (main.cpp:228-241) / teardown (681-686).
 * device = HIP device ordinal.  flags: RZ_FLAG_* below.  Returns NULL on
 * failure (query rz_last_error(NULL)). */
#define RZ_FLAG_NONE          0u  /* default: one lane per sample (rz_render_samples)                         */
#define RZ_FLAG_MEGAKERNEL    1u  /* one lane per pixel, samples in sequence (rz_render_pixels): cross-check  */
#define RZ_FLAG_HOST_RELAYOUT 4u  /* re-lay the scene out on the host instead of on the device (same bytes; cross-check) */
rz_ctx*     rz_create(int device, unsigned flags);
void        rz_destroy(rz_ctx* ctx);
const char* rz_last_error(const rz_ctx* ctx);

/* glGenBuffers + glBufferData + glBindBufferBase (main.cpp:1072-1119).
 * Allocates device storage for the binding and copies `bytes` from host
 * memory; the host pointer is not retained. `bytes` must be a multiple of
 * the binding's element size; 0 is allowed (empty binding). */
int rz_upload(rz_ctx* ctx, rz_binding binding, const void* data, size_t bytes);

/* glBufferSubData (main.cpp:1196-1207): in-place refresh of
 * [offset, offset+bytes) of a binding uploaded before. */
int rz_update(rz_ctx* ctx, rz_binding binding, size_t offset,
              const void* data, size_t bytes);

/* updateDynamicBVHAndSSBOs (main.cpp:1138-1194) done ON THE DEVICE: hand over only the per-instance transforms
 * (n x 16 floats, column-major, n == number of uploaded instances); the library inverts them, recomputes the world
 * AABBs (main.cpp:1168-1191) and rebuilds the TLAS (BVH.cpp:178-240) in one small kernel.  The result is byte-identical
 * to what the host library (SceneBuffers::updateDynamic, librayzen_host.so) produces and rz_read_binding returns it.
 * Inverses and world boxes follow GLM 0.9.9's evaluation order, which is what RayZen's own code executes at
 * main.cpp:974-1001 and 1150-1191: glm::inverse = compute_inverse<4, 4> (the eighteen 2x2 sub-determinants, cofactor
 * columns with alternating signs, the determinant from the cofactors' first row summed pairwise, every cofactor times
 * its reciprocal) and mat4 * vec4 = (m0 x + m1 y) + (m2 z + m3 w).  GLM is not vendored with the reference and is absent
 * from this image, so this is a RESTATEMENT of its published algorithm (rz_linalg.h on the host, rz_tlas_device.hip on
 * the device; the CPU checker states it a third time), pinned by hand-derived known answers in tests/test_linalg_glm.py -- not a run of
 * GLM.  Synchronises the context's stream (the TLAS depth sizes the next launch). */
int rz_update_transforms(rz_ctx* ctx, const float* transforms, size_t n);

/* BVH::buildBLAS (RayZen/src/BVH.cpp:99-175 with the full-sweep SAH of :22-97; called per mesh from main.cpp:954-958)
 * on the device.  Output is byte-identical to the reference builder's `nodes` / `triIndices`: nodes_out receives
 * *n_nodes <= 2n-1 nodes (nodes_cap >= 2n-1, or 1 for n == 0), indices_out n triangle indices.  depth and device_ms
 * (device time, host<->device copies excluded) may be NULL.  tris / outputs are host pointers. */
int rz_build_blas(rz_ctx* ctx, const rz_triangle* tris, size_t n, rz_bvh_node* nodes_out, size_t nodes_cap,
                  int32_t* indices_out, size_t* n_nodes, int* depth, float* device_ms);

/* initializeSSBOs' geometry half (main.cpp:951-972, 1030-1035) without the host in the middle: builds the BLAS of every
 * mesh on the device (the builder of rz_build_blas), concatenates nodes and indices THERE, and makes the results the
 * context's bindings 0, 7 and 8 -- equivalent to BVH::buildBLAS per mesh + rz_upload of the three concatenated arrays,
 * same bytes, but the node / index arrays never visit the host (rz_read_binding fetches them if asked; at 1 M triangles
 * the round trip was 28 of rz_build_blas's 36 ms).  `triangles` holds all meshes' triangles back to back; mesh i is
 * triangles[first_triangle .. +n_triangles).  Per mesh the call returns what an instance of it needs: blasNodeOffset,
 * blasTriOffset (globalTriOffset is first_triangle) and the BLAS root node, whose box main.cpp:974-993 turns into the
 * instance's world box for the TLAS.  The frontend then uploads instances, TLAS, materials and lights as usual. */
typedef struct rz_mesh_build {
    size_t  first_triangle;  /* in  */
    size_t  n_triangles;     /* in  */
    int32_t node_offset;     /* out: blasNodeOffset */
    int32_t index_offset;    /* out: blasTriOffset  */
    int32_t n_nodes;         /* out */
    int32_t depth;           /* out */
    rz_bvh_node root;        /* out */
} rz_mesh_build;
int rz_build_geometry(rz_ctx* ctx, const rz_triangle* triangles, size_t n_triangles, rz_mesh_build* meshes, size_t n_meshes);

/* Copy a binding's current content back to the host in RayZen's own layout (after rz_update_transforms: the
 * instances / TLAS nodes / TLAS indices the device built).  out == NULL: only *needed is set. */
int rz_read_binding(rz_ctx* ctx, rz_binding binding, void* out, size_t bytes, size_t* needed);

/* glUniform* in sendSceneDataToShader (main.cpp:1356-1379).  A change of resolution or of the tile assignment
 * (tile_rank / tile_nranks) zeroes the whole accumulation buffer: pixels a context does not own always read as zero. */
int rz_set_frame(rz_ctx* ctx, const rz_frame_params* params);

/* Optional plumbing for a caller that owns the device memory and the stream
 * (e.g. a torch tensor reduced with RCCL): render on `hip_stream` (a
 * hipStream_t; NULL = the context's own stream) and accumulate into
 * `device_rgba` (width*height*4 floats, device memory; NULL = the context's
 * own buffer). */
int rz_set_stream(rz_ctx* ctx, void* hip_stream);
int rz_bind_accum(rz_ctx* ctx, void* device_rgba, size_t bytes);

/* glDrawArrays(GL_TRIANGLE_FAN,0,4) (main.cpp:637): asynchronous launch of
 * one path-tracing frame: params.spp samples for every pixel of the tiles
 * this context owns, ADDED to the accumulation buffer (RGBA32F; rgb = sum of
 * per-sample radiance before the reference's divide and clamp,
 * fragment_shader.glsl:772-773; a = number of samples).  sample_base == 0
 * first clears the owned pixels. */
int rz_render(rz_ctx* ctx);
/* Same frame, and additionally counts the reference algorithm's memory
 * touches (slower; never used for timing). */
int rz_render_counted(rz_ctx* ctx, rz_counters* out);
/* glFinish (main.cpp:1347). */
int rz_sync(rz_ctx* ctx);

/* Zero the whole accumulation buffer (all pixels, owned or not). */
int rz_clear_accum(rz_ctx* ctx);

/* The reference never reads pixels back; this is new.  Copies the RGBA32F
 * accumulation buffer to host.  Row 0 is the BOTTOM row (gl_FragCoord
 * origin).  Synchronises. */
int rz_read_accum(rz_ctx* ctx, float* rgba, size_t bytes);

/* color /= numSamples; clamp(0,1) (fragment_shader.glsl:772-773), then
 * 8-bit quantisation round(c*255) as the default framebuffer would.
 * Row 0 = bottom row.  Synchronises. */
int rz_resolve_rgba8(rz_ctx* ctx, uint8_t* rgba8, size_t bytes);

/* The rest of the shader's main() after the path loop (fragment_shader.glsl:772-819): resolve, then the overlays the
 * reference draws on top -- BVH wireframe (debugShowBVH/debugBVHMode/debugSelectedBLAS/debugSelectedTri,
 * glsl:98-104,214-373), light markers (debugShowLights, glsl:781-803) and the FPS digits (uniformFps, glsl:805-819;
 * the reference always draws them) -- and 8-bit quantisation.  rgba8 (width*height*4 bytes) and rgb32f
 * (width*height*3 floats: the colour before quantisation) may each be NULL.  Row 0 = bottom row.  Synchronises. */
typedef struct rz_present_params {
    float   fps;            /* uniformFps */
    int32_t show_fps;       /* 1 = as the reference */
    int32_t show_lights;    /* debugShowLights */
    int32_t show_bvh;       /* debugShowBVH */
    int32_t bvh_mode;       /* debugBVHMode: 0 = TLAS leaves + BLAS roots, 1 = branch to one triangle */
    int32_t selected_blas;  /* debugSelectedBLAS (an instance index) */
    int32_t selected_tri;   /* debugSelectedTri (mesh-local triangle id) */
} rz_present_params;
int rz_present(rz_ctx* ctx, const rz_present_params* params, uint8_t* rgba8, size_t rgba8_bytes, float* rgb32f,
               size_t rgb32f_bytes);

/* Wall-clock-free timing: milliseconds the render kernels of the LAST
 * rz_render spent on the GPU (HIP events on the stream they ran on), and how
 * many kernel launches that was.  Synchronises. */
int rz_last_render_ms(rz_ctx* ctx, float* ms, int* launches);
/* Every rz_render launch is bracketed by a HIP event pair on its stream (a ring of 64).  Copies the
 * durations (ms, oldest first) of the launches issued since the previous call -- at most `cap`,
 * at most 64 -- and returns how many; negative on error.  Synchronises on those events only, so a
 * timed loop can issue its launches back to back and collect their GPU times afterwards. */
int rz_render_history_ms(rz_ctx* ctx, float* ms, int cap);
/* Name of the render kernel the last rz_render used (the one those event pairs bracket): the library picks
 * "rz_render_samples" (one lane per sample; "rz_render_samples<glass>" when a triangle uses a transparent material
 * and currentIor is speculated) or "rz_render_pixels" (RZ_FLAG_MEGAKERNEL). */
const char* rz_last_kernel_name(const rz_ctx* ctx);

/* Device pointer of the accumulation buffer currently in use. */
void* rz_accum_device_ptr(rz_ctx* ctx);

/* TEST HOOK: make the nth host-side allocation site reached from now on (scene re-layout, upload copies, staging
 * vectors) fail as if memory had run out; the call in progress returns RZ_ERR_NO_MEMORY and the context stays usable.
 * nth <= 0 disarms. */
int rz_debug_fail_alloc(rz_ctx* ctx, int nth);

/* The HIP stream (hipStream_t) the context's work is enqueued on. */
void* rz_stream_handle(rz_ctx* ctx);

/* ------------------------------------------------------------------------ */
/* Multi-GPU group: tile-sharded rendering + ONE exchange step per frame.     */
/* ------------------------------------------------------------------------ */
/* The reference is single-GPU: its context lifetime is glfwCreateWindow / glfwMakeContextCurrent (main.cpp:228-241)
 * and the teardown at main.cpp:681-686.  A group is that lifetime for N GPUs of one node: it owns one rz_ctx per LOCAL
 * device and the RCCL communicator(s), gives member m the tiles t with t % nranks == rank(m) (rz_frame_params.tile_rank
 * / tile_nranks are filled in by the group), and lands the frame on the root rank with ONE exchange step over xGMI,
 * issued on the members' render streams (no host synchronisation between the render kernel and the exchange).  Two
 * exchange steps exist, both bit-identical to a single-GPU frame:
 *   "reduce" (the default; BASELINE.json's "single RCCL reduce on the accumulation buffer"): ONE ncclReduce(SUM) of the
 *            whole RGBA32F accumulation buffers -- tile sets are disjoint and non-owned pixels are zero, so the sum adds
 *            each pixel's single value to zeros;
 *   "gather" (RZ_GROUP_TRANSPORT=gather in the environment when the group is made, or rz_group_set_transport): every
 *            member packs the tiles it owns (1 / N of the frame) and sends them straight to the root (ncclSend /
 *            ncclRecv), which scatters the N sets into the frame; bits are copied, never added.  A gather whose enqueue
 *            fails makes the group fall back to "reduce" for the rest of its life (the frame still lands).
 * Every rank of a group must use the same one.
 *
 * Two ways to form a group:
 *   rz_group_create       one process drives ndev devices (ncclCommInitAll); ranks = 0..ndev-1, all local.
 *   rz_group_create_rank  one process per GPU (the usual launcher layout): every process passes its own device, its
 *                         rank, the group size and the 128-byte id that rank 0 obtained from rz_group_unique_id and
 *                         distributed by whatever channel the launcher has (ncclCommInitRank; blocks until all ranks
 *                         have called it).
 * RCCL is bound at run time, when the first group is created (librayzen_hip.so has no link-time dependency on the
 * 570-MB librccl, so single-GPU users never load it): an RCCL already loaded in the process is reused, else
 * $RZ_RCCL_LIBRARY, else /opt/rocm/lib/librccl.so.1.  rz_group_rccl_version() reports what was bound (no GPU needed). */
typedef struct rz_group rz_group;
#define RZ_GROUP_ID_BYTES 128
/* rz_group_create only, or-ed into `flags`: a REHEARSAL group -- `devices` may name a device more than once (N ranks on
 * one GPU), no communicator is made and device-to-device copies stand in for the links.  Everything else (the dealing
 * of tiles, packing, the root's scatter, stream ordering) is the code an N-GPU group runs. */
#define RZ_GROUP_LOOPBACK 0x10000u
int       rz_group_rccl_version(int* version);                 /* binds RCCL; *version = ncclGetVersion() */
int       rz_group_unique_id(void* id128);                      /* ncclGetUniqueId into RZ_GROUP_ID_BYTES bytes */
rz_group* rz_group_create(int ndev, const int* devices, unsigned flags);        /* devices NULL: 0..ndev-1 */
rz_group* rz_group_create_rank(int device, int rank, int nranks, const void* id128, unsigned flags);
void      rz_group_destroy(rz_group* g);
const char* rz_group_last_error(const rz_group* g);             /* g NULL: the error of a failed create */
int       rz_group_size(const rz_group* g);                     /* ranks in the communicator */
const char* rz_group_transport(const rz_group* g);              /* how rz_group_reduce moves the frame: "rccl-reduce" | "rccl-reduce(fallback: why)" | "tile-gather(...)" */
int       rz_group_set_transport(rz_group* g, const char* name); /* "reduce" | "gather", from the next rz_group_reduce on; the same call on EVERY rank */
int       rz_group_local_count(const rz_group* g);              /* members owned by this process */
int       rz_group_rank(const rz_group* g, int local);          /* global rank of local member `local` */
rz_ctx*   rz_group_ctx(rz_group* g, int local);                 /* the member's context (for per-device calls) */
/* glBufferData / glBufferSubData on every local member (the scene is replicated: <= 88 MB even for configs[4]). */
int rz_group_upload(rz_group* g, rz_binding binding, const void* data, size_t bytes);
int rz_group_update(rz_group* g, rz_binding binding, size_t offset, const void* data, size_t bytes);
/* sendSceneDataToShader for every local member; tile_rank / tile_nranks of *params are ignored and set per member. */
int rz_group_set_frame(rz_group* g, const rz_frame_params* params);
/* glDrawArrays on every local member: asynchronous, each on its own device and stream. */
int rz_group_render(rz_group* g);
/* The frame's exchange step (tile gather or ncclReduce, see above), enqueued on every member's stream behind its render
 * kernel, from the members' accumulation buffers into the root member's frame buffer (out of place: nobody's
 * accumulation buffer is overwritten, so frames can be continued with sample_base > 0). */
int rz_group_reduce(rz_group* g, int root);
int rz_group_sync(rz_group* g);                                 /* glFinish on every local member */
/* GPU time of the last rz_group_reduce, from HIP events recorded on each local member's stream just before and just
 * after its share of the collective was enqueued: *root_ms on the root member (-1 when the root lives in another
 * process), *max_ms the longest over this process's members.  A member's interval starts when its render kernel ends,
 * so it contains the wait for the slowest rank as well as the transfer.  Synchronises the members' streams. */
int rz_group_last_reduce_ms(rz_group* g, float* root_ms, float* max_ms);
/* Copy the reduced frame (RGBA32F, row 0 = bottom) to host memory.  Only valid in the process that owns `root` of the
 * last rz_group_reduce; RZ_ERR_NOT_READY elsewhere.  Synchronises that member's stream. */
int rz_group_read_frame(rz_group* g, float* rgba, size_t bytes);
/* Device pointer of the reduced frame on the root member (NULL in other processes). */
void* rz_group_frame_device_ptr(rz_group* g);

/* TEST HOOK: set bits of the context's backstop word as a render kernel that ran into one of its bounds would; the next
 * rz_sync returns RZ_ERR_INTERNAL naming them and clears the word. */
int rz_debug_poke_backstop(rz_ctx* ctx, unsigned bits);

/* TEST HOOK: the device-side scene layout as built (which = 0: DevPair[] 64 B each, 1: DevTri[] 48 B each;
 * rayzen_amd/csrc/hip/rz_scene_dev.h).  out NULL: only *needed is set.  Runs the pending re-layout first. */
int rz_debug_read_layout(rz_ctx* ctx, int which, void* out, size_t bytes, size_t* needed);

/* TEST HOOK: how the last rz_render / rz_render_counted of this context was launched (rz_kernels.hip:
 * plan_render_samples).  A (pixel, 64-sample batch) pair is one "unit" of work; persistent launches hand their waves
 * `per_claim` pixel groups per atomic, compacting ones work off `claim_units`-unit claims. */
typedef struct rz_launch_plan {
    int64_t groups;             /* pixel groups of the launch (one wave's pixels: 1 pixel when spp >= 64, else 64 / spp) */
    int64_t grid;               /* workgroups launched (one wave each) */
    int32_t per_claim;          /* groups a persistent wave claims per atomic; 0 = one workgroup per group */
    int32_t claim_units;        /* units of a compacting claim (8 or 16); 0 = the launch does not compact */
    int32_t batches_per_pixel;  /* ceil(spp / 64) */
    int32_t pixels_per_wave;    /* 1, or 64 / spp when spp < 64 */
    int32_t lds_stack_entries;  /* BLAS stack entries per lane kept in LDS */
    int32_t overflow_entries;   /* ... and in the global overflow columns (0: the whole stack fits the LDS window) */
    int32_t transparent;        /* 1: the scene has a transparent material (the speculating variant of the kernel) */
    int32_t scratch_mib;        /* MiB of scratch the launch's resident waves own (pools of parked paths + wait slots); 0: none */
} rz_launch_plan;
int rz_debug_last_plan(rz_ctx* ctx, rz_launch_plan* out);

/* Number of HIP devices visible to the process (0 without a GPU). */
int rz_device_count(void);

/* Library/version probe that needs no GPU. */
const char* rz_version(void);
/* The ABI revision this library was compiled to: bumped whenever a struct of this header changes size or meaning
 * (rz_counters grew in round 4 without one -- a caller built against the older header would have been written past).
 * A binding compares it with RZ_ABI_VERSION of the header it was written against before its first call. */
#define RZ_ABI_VERSION 5
int rz_abi_version(void);
/* Which implementation of the three built-ins GLSL leaves open -- sin, cos, acos; RayZen's hash is fract(sin(x) * 43758.5453)
 * (fragment_shader.glsl:188-190) -- this library was compiled with (rz_device_math.h, RZ_MATH_FLAVOUR): 1 = Mesa llvmpipe's, the
 * OpenGL implementation RayZen's own shader was run on for this project's parity tests (binary32 Cephes sin / cos with fused
 * multiply-adds, Mesa's acos polynomial); 0 = binary64 evaluation rounded once (correctly rounded).  A frame is a function of this
 * choice from the third path segment on.  Needs no GPU. */
int rz_math_flavour(void);
/* sha256 (64 hex digits) of the sources and compiler flags this library was built from (rayzen_amd/build.py:
 * source_hash), or "unstamped": ties the LOADED library to a source tree and to a committed profile.  Needs no GPU. */
const char* rz_source_hash(void);
/* sizeof() of the ABI structs as compiled into the library, for layout
 * checks from other languages: which = 0 triangle, 1 node, 2 instance,
 * 3 material, 4 light, 5 frame_params, 6 counters. */
size_t rz_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* RAYZEN_HIP_H */
