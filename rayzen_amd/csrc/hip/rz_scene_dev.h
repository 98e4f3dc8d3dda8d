// rz_scene_dev.h -- how the scene lives in HBM.
//
// The C-ABI receives RayZen's SSBO arrays verbatim (include/rayzen_hip.h).
// They are re-laid-out ONCE, when a frame is rendered after an upload
// (rz_context.hip: finalize), into the structures below.  Nothing about the
// re-layout changes a single float the traversal computes: boxes and
// vertices are copied bit for bit, edge vectors are the same one-rounding
// subtractions FS:392-393 perform per test, and child / triangle visiting
// order is the reference's.
//
//  * DevPair (64 B, 64-B aligned): the two children of one internal BLAS
//    node -- their boxes and an encoded reference each.  RayZen allocates
//    children as adjacent pairs (BVH.cpp:166-171) and its shader fetches them
//    separately when popped; here one aligned 64-B record (4 x dwordx4 per
//    lane, one cache line) serves both box tests at the moment the parent is
//    expanded.  The two planes of an axis sit side by side so that
//    (plane - o) * inv runs as v_pk_add_f32 / v_pk_mul_f32 on the pair (same
//    IEEE operations, half the issue slots; rz_trace.h: slab_pair).
//  * enc: >= 0  -> internal child: index of ITS DevPair (relative to the
//                  BLAS's pair base);
//         <  0  -> leaf child: ~enc = (firstSlot << 4) | count, count 0..15,
//                  firstSlot relative to the BLAS's triangle base.
//  * DevTri (48 B = 3 x dwordx4): triangles gathered into LEAF ORDER, which
//    removes the blasTriIndices indirection (FS:434) and makes a leaf's <=4
//    triangles contiguous.  v0, e1 = v1 - v0, e2 = v2 - v0, materialIndex.
//  * DevInstance (144 B): the 3x4 parts of transform / inverseTransform the
//    shader actually uses (FS:476-477,484,489), the BLAS root box + enc (so
//    the root test needs no node fetch) and the bases of its pair / triangle
//    ranges.
//  * TLAS nodes and indices keep RayZen's 32-B node layout: a TLAS has a
//    handful of nodes and is traversed literally (FS:457-503).
#pragma once
#include <stdint.h>

namespace rz {

struct alignas(64) DevPair {     // per axis (min, max) adjacent: one packed-f32 op handles both planes of an axis
    float lx[2], ly[2];          // left child's box
    float lz[2], rx[2];          // ... and the right child's
    float ry[2], rz[2];
    int32_t lenc, renc;
    int32_t pad[2];
};
static_assert(sizeof(DevPair) == 64, "DevPair is one 64-B line");

struct alignas(16) DevTri {
    float v0[3]; float e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; int32_t mat; int32_t src; int32_t pad;   // src = index in the caller's triangle array
};
static_assert(sizeof(DevTri) == 48, "DevTri is three 16-B loads");

// What the winner of a closest-hit query needs of its triangle -- FS:411's normalize(cross(edge1, edge2)) and the material
// index -- in one 16-B load (derived from DevTri on the device, rz_relayout.hip: rl_tri_normals).
struct alignas(16) DevTriN { float n[3]; int32_t mat; };
static_assert(sizeof(DevTriN) == 16, "DevTriN is one 16-B load");

struct alignas(16) DevInstance {
    float inv[12];      // inverseTransform columns 0..3, rows 0..2: c0.xyz c1.xyz c2.xyz c3.xyz
    float fwd[12];      // transform, same packing
    float rootMin[3]; int32_t rootEnc;
    float rootMax[3]; int32_t pairBase;
    int32_t triBase;
    int32_t flags;      // bit 0: BLAS is empty / invalid -> never hit
    int32_t pad[2];
};
static_assert(sizeof(DevInstance) == 144, "DevInstance");

struct TlasNode {       // == rz_bvh_node
    float bmin[3]; int32_t leftFirst;
    float bmax[3]; int32_t count;
};

// The TLAS in the order the shader's stack loop pops it (depth first, right child before left; rz_trace.h:
// trace_closest), one 64-B record per pop position -- one s_load_dwordx16.
struct alignas(64) TlasDfs {
    float bmin[3]; int32_t first;   // leaf: first entry in the TLAS index array
    float bmax[3]; int32_t count;   // > 0 leaf (instances), < 0 internal, 0 never expands (empty root / the shader's stack[64] would be full)
    int32_t skip;                   // the position after this node's subtree
    int32_t inst0;                  // leaf: the instance its first entry names
    int32_t pad[6];
};
static_assert(sizeof(TlasDfs) == 64, "TlasDfs is one scalar fetch");

struct DevMaterial { float albedo[3], metallic, roughness, reflectivity, transparency, ior; };
struct DevLight { float posdir[4], color[3], power; };

struct DevCounters {
    unsigned long long samples, traversals, tlas_nodes, tlas_leaf_indices, instances, blas_nodes, triangles,
        materials, light_fetches, pixels, scatters, diffuse_scatters, hemi_draws, lit_lights, triangles_past_u;
};

// Everything a render kernel needs, passed by value.
struct KParams {
    const DevPair* pairs;
    const DevTri* tris;
    const DevInstance* instances;
    const TlasDfs* tlasDfs;     // the TLAS in pop order (derived; the verbatim node array stays with the context for rz_present / rz_read_binding)
    const int32_t* tlasIndices;
    const DevMaterial* materials;
    const DevLight* lights;
    float4* accum;          // width*height RGBA32F, row 0 = bottom
    float* ior;             // width*height: FS:674's currentIor carried across rz_render calls
    DevCounters* counters;  // only for the counting build
    unsigned* groupCounter; // next unclaimed pixel group of this launch (persistent waves, rz_kernels.hip); zeroed per launch
    int32_t nTlasDfs;
    int32_t nLights;        // min(numLights uniform, lights.length())  (FS:574-575)
    int32_t nMaterials;
    int32_t width, height;
    int32_t tilesX, tilesY;
    int32_t nLocalTiles;    // tiles owned by this context
    int32_t tileRank, tileNRanks;
    int32_t maxBounces;
    int32_t spp, sampleBase;
    int32_t nSlots;         // nLocalTiles * 64
    int32_t blasStackCap;   // LDS entries per lane for the BLAS stack: the whole stack (max BLAS depth - 1), or a window of it
    int32_t tlasStackCap;   // 0 since the TLAS walk needs no stack (kept: the kernel-argument layout steers register allocation)
    float invView[16];
    float invProj[16];
    float camPos[3];
    int32_t blasOvfCap;     // entries per lane beyond the LDS window, kept in global memory (persistent launches only; else 0)
    uint2* blasOvf;         // [resident wave][blasOvfCap][64 lanes]
    float* wslots;          // compacting launches (rz_kernels.hip: WAIT SLOTS): per resident wave, the claim scratch [claimUnits][6][64] floats, then nWaitSlots slots of slotFloats floats -- one group's addends [batch][6][64] each
    uint32_t wslotStride;   // floats from one resident wave's scratch to the next (claimUnits x 384 + nWaitSlots x slotFloats)
    const DevTriN* triN;    // [triangle in leaf order] (rz_trace.h: trace_closest's epilogue)
    float hemi0[3];         // rz_path.h: hemisphere_local((+0, +0)), the local direction of every bounce-0 scatter (rz_hemi0_kernel, once per context)
    int32_t traceRoundCap;  // rz_trace.h: trace_spread's backstop -- more rounds than any walk of this TLAS takes (a lane enters an instance at most once per leaf entry)
    int32_t spreadTrace;    // 1: third and later path segments are traced lane by lane (rz_trace.h: trace_spread); 0: always the wave-cursor walk (a scheduling choice: same image)
    // ---- the pool of parked paths a resident wave keeps ACROSS its claims (rz_kernels.hip: pool_process; compacting launches only)
    unsigned* wpool;        // [resident wave][RZ_GPOOL_FIELDS][wpoolStride]
    uint32_t wpoolStride;   // slots per field: wpoolChunk + the most one claim can park
    uint32_t wpoolChunk;    // the wave traces its pool when it holds at least this many paths (and at the end of the launch)
    int32_t* wmeta;         // [resident wave][4 nWaitSlots]: the group in each of the wave's wait slots | its outstanding paths (-1: the slot is free) | transparent scenes: 1 = one of its pooled paths met glass | ... the groups to render again (rz_kernels.hip: redo list)
    uint32_t slotFloats;    // floats per group in a wait slot and in the claim scratch: batches per pixel x 384 (+ 64 in transparent scenes: the currentIor each pixel ends with)
    int32_t nWaitSlots;     // wait slots per resident wave (<= 64, >= twice the groups of a claim)
    int32_t claimUnits;     // units of a claim (8 or 16: the kernel's COMPACT parameter)
    int32_t glassBoxHint;   // transparent scenes: 1 = a late path whose ray passes the box of an instance with glass in it is not parked (rz_kernels.hip: may_hit_glass; RZ_GLASS_BOX_HINT=0 switches it off: same image)
    uint32_t claimScratchFloats;    // floats of a resident wave's claim scratch (groups of a claim x slotFloats); its wait slots follow
    int32_t drainEachClaim; // 1: a wave traces its pool to the end after every claim (RZ_CROSS_CLAIM_POOL=0: the per-claim pools of round 2; a scheduling choice, same image)
    int32_t regularBoxes;   // 1: every BLAS child box has min <= max on every axis (no NaN): the octant-specialised slab test may be used
    // ---- transparent scenes, persistent launches: per resident wave, the state of each lane's sample in front of its first
    //      transparent scatter (rz_path.h: snapshot_store), from which the sample's second version starts; null: re-runs start at the camera
    float* snap;            // [resident wave][snapStride] floats
    uint32_t snapStride;    // >= (RZ_SNAP_FIELDS + RZ_SNAP_TALLY + RZ_GVER_ROWS) * 64 + RZ_GLATE_FIELDS * RZ_GLATE_CAP
};

// Arguments of the device TLAS rebuild (rz_tlas_device.hip: rz_tlas_refit; filled in by rz_context.hip).
struct TlasWork {
    const float* transforms;        // n x 16, column-major
    DevInstance* instances;         // in/out: fwd, inv rewritten; root box / bases kept
    rz_bvh_instance* refInstances;  // out: transform + inverseTransform (offsets kept)
    TlasNode* nodes;                // out: 2n-1 nodes
    int32_t* indices;               // out: n
    TlasDfs* dfs;                   // out: the nodes in the shader's pop order (rz_trace.h: trace_closest), 2n-1 records
    float* worldMin;                // scratch n x 3
    float* worldMax;                // scratch n x 3
    int32_t* order;                 // scratch n (meshIndices)
    int32_t* stack;                 // scratch 3 x (2n+8)
    int32_t* outCounts;             // [0] = node count, [1] = index count, [2] = depth
    int32_t* scratch;               // 4 x n ints (ranks|flags, two pointer buffers, the permuted order) + 2 x 6 x (n + 1) (level lists)
    int n;
};

// How rz_render_samples is launched (rz_kernels.hip: plan_render_samples): groups of pixels, the grid, and the number
// of groups a persistent wave claims per atomic (0: one workgroup per group).
struct SamplesPlan { long long groups, grid; int perClaim; bool compact; int claimUnits; long long nClaims; int runShift; bool drainEachClaim; };   // claimUnits: 8 or 16 when compact; nClaims / runShift: rz_kernels.hip, ClaimMap; drainEachClaim: KParams

// The kernel is instantiated for claims of 8 and of 16 units (rz_kernels.hip: plan_render_samples picks by the size of the launch).
constexpr int RZ_CLAIM_UNITS_SMALL = 8, RZ_CLAIM_UNITS_LARGE = 16;
constexpr int RZ_ERRWORD = 16;          // KParams::groupCounter[RZ_ERRWORD]: the launch's backstop bits (rz_kernels.hip: rz_backstop; read by rz_sync)
constexpr int RZ_POOL_FIELDS = 24;      // the parked path (13), its query (8), the items of a B phase (2), its currentIor (transparent scenes): rz_trace.h, namespace poolf
constexpr int RZ_GPOOL_FIELDS = RZ_POOL_FIELDS + 1;     // a wave's pool: ... + the wait slot of the group a parked path belongs to (field 24)
constexpr int RZ_SNAP_FIELDS = 19, RZ_SNAP_TALLY = 14;  // rz_path.h: snapshot_store -- a sample's state in front of its first transparent scatter (+ its tallies, counting launches)
constexpr int RZ_GLATE_FIELDS = 19, RZ_GLATE_CAP = 128;    // rz_kernels.hip: pool_process -- the pooled samples of a transparent scene that stand in front of a transparent scatter (field-major records behind the version rows)
constexpr int RZ_GVER_ROWS = 12;        // rz_kernels.hip: glass_resolve_unit -- the two versions of a sample's addends, [version][6] rows of 64 floats behind the snapshot rows

}  // namespace rz
