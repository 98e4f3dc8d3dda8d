// rz_internal.h -- what the translation units of librayzen_hip.so share besides the device data layout
// (rz_scene_dev.h): the structs that cross file boundaries and the launch / helper entry points, declared ONCE.
// (Round 1 repeated some of these structs in the files that use them; two copies of one struct are an ODR violation
// waiting for the day one of them changes.)
#pragma once
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_scene_dev.h"

namespace rz {

// ---- rz_kernels.hip
void launch_render_pixels(const KParams& K, bool counted, hipStream_t stream);
SamplesPlan plan_render_samples(int spp, int nSlots, bool glass);
size_t samples_lds_extra(bool glass, bool compact);
void launch_render_samples(const KParams& K, bool counted, bool glass, hipStream_t stream);
void launch_resolve(const float4* accum, uchar4* out, int n, hipStream_t stream);
int compute_hemi0(float out[3], hipStream_t stream);
#ifdef RZ_PROF
void dump_wave_log(int nWaves);
#endif
#ifdef RZ_GSTATS
void dump_gstats();
#endif

// ---- rz_tlas_device.hip
void launch_tlas_refit(const TlasWork& W, hipStream_t s);

// ---- rz_blas_device.hip
size_t blas_build_workspace_bytes(size_t n);
int blas_build_device(const rz_triangle* hostTris, size_t n, void* workspace, size_t workspaceBytes, rz_bvh_node* nodes_out,
                      int32_t* idx_out, int* nNodesOut, int* depthOut, float* ms, hipStream_t s);

// ---- rz_relayout.hip
struct RelayoutView {           // in: the three offsets; out: everything else
    int nodeOff, triOff, gTriOff;
    int pairBase, triBase;      // where this view's pairs / triangles start in the global arrays (in)
    int nPairs, nSlots, depth, rootEnc, empty;
    float rootMin[3], rootMax[3];
};
size_t relayout_workspace_bytes(size_t nNodes);
int relayout_view_device(const rz_bvh_node* nodes, long long nNodes, const int32_t* idx, long long nIdx, const rz_triangle* tris,
                         long long nTris, const rz_material* mats, int nMat, const rz_bvh_node& hostRoot, RelayoutView& V,
                         DevPair* pairs, long long pairCap, DevTri* trisOut, long long triCap, void* workspace, size_t workspaceBytes,
                         int* pinned, unsigned* transparentOut, hipStream_t s);
int tri_normals_device(const DevTri* tris, long long n, DevTriN* out, hipStream_t s);
int relayout_check_materials_device(const DevTri* tris, long long n, const rz_material* mats, int nMat, void* workspace, int* pinned,
                                    unsigned* transparentOut, int* detail, hipStream_t s);

// ---- rz_present.hip
struct ProjBox;                 // screen-space corners of one box (rz_present.hip)
struct PresentParams {
    const float4* accum;
    uchar4* rgba8;              // may be null
    float* rgb;                 // may be null: 3 floats per pixel, the colour before quantisation
    const TlasNode* tlasNodes;
    const int32_t* tlasIndices;
    const DevInstance* instances;
    const DevLight* lights;
    int width, height;
    int nTlasNodes, nInstances, nLights;
    float viewProj[16];         // projectionMatrix * viewMatrix
    float fps;
    int showFps, showLights, showBvh, bvhMode;
    int pathLen;                // bvhMode 1: nodes on the branch to the selected triangle
    float pathMin[32][3], pathMax[32][3];   // their object-space boxes
    float selTransform[16];     // the selected instance's transform
    ProjBox* boxes;             // screen-space corners of every box a pixel may have to draw, projected ONCE
};
void launch_present(const PresentParams& P, hipStream_t s);

}  // namespace rz
