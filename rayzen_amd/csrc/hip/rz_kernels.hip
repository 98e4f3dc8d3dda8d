// rz_kernels.hip -- render kernels for gfx950 (MI355X).  Compile with
// --offload-arch=gfx950 -ffp-contract=off (see rz_device_math.h).
//
// rz_render_pixels<COUNT>: one lane per pixel, one wavefront per 8x8-pixel
// tile (RZ_TILE_W x RZ_TILE_H), 4 wavefronts per workgroup.  Each lane walks
// its pixel's samples in order (the shader's per-pixel state -- currentIor,
// the running colour sum -- makes samples of one pixel sequential) as the
// trace -> advance state machine of rz_path.h, so the 64 lanes of a wave
// share ONE traversal loop whatever phase (primary, shadow iteration,
// bounce) each is in.  Tiles are dealt tile t -> rank t % nranks, so any
// number of GPUs splits a frame into disjoint pixel sets.
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_path.h"

namespace rz {

constexpr int WAVES_PER_BLOCK = 4;
#ifndef RZ_MIN_WAVES_PER_SIMD
#define RZ_MIN_WAVES_PER_SIMD 2
#endif

template <bool COUNT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, RZ_MIN_WAVES_PER_SIMD) void rz_render_pixels(const KParams K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // LDS: per wave, [blasStackCap][64] uint2 then [tlasStackCap][64] int
    const size_t perWave = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int);
    unsigned char* base = lds_raw + perWave * wave;
    uint2* bstk = reinterpret_cast<uint2*>(base) + lane;
    int* tstk = reinterpret_cast<int*>(base + (size_t)K.blasStackCap * 64 * sizeof(uint2)) + lane;

    const int localTile = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (localTile >= K.nLocalTiles) return;
    const int tile = localTile * K.tileNRanks + K.tileRank;
    const int tx = tile % K.tilesX, ty = tile / K.tilesX;
    const int px = tx * RZ_TILE_W + (lane & 7), py = ty * RZ_TILE_H + (lane >> 3);
    const bool inside = px < K.width && py < K.height;
    const size_t pix = (size_t)py * K.width + px;

    Tally c = {};
    Path P;
    P.mode = MODE_DONE;
    P.samp = K.sampleBase;
    P.sampEnd = inside ? K.sampleBase + K.spp : K.sampleBase;
    float alpha = 0.0f;
    if (inside) {
        const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
        P.uv.x = fragx / (float)K.width;
        P.uv.y = fragy / (float)K.height;
        P.fragSum = fragx + fragy;
        if (K.sampleBase == 0) {
            P.color = mk3(0.0f, 0.0f, 0.0f);
            P.ior = 1.0f;
        } else {
            const float4 a = K.accum[pix];
            P.color = mk3(a.x, a.y, a.z);
            alpha = a.w;
            P.ior = K.ior[pix];
        }
    }
#ifdef RZ_PROF
    unsigned long long tTrace = 0, tAdv = 0, tBegin = 0;
#endif
    while (P.samp < P.sampEnd) {
#ifdef RZ_PROF
        RZ_SITE(c, 6);
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (P.mode == MODE_DONE) { RZ_SITE(c, 7); begin_sample<COUNT>(K, P, c); }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        HitRec h;
        const bool found = trace_closest<COUNT>(K, P.o, P.d, h, bstk, tstk, c);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        advance<COUNT>(K, P, found, h, c);
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        tBegin += t1 - t0; tTrace += t2 - t1; tAdv += t3 - t2;
#else
        if (P.mode == MODE_DONE) begin_sample<COUNT>(K, P, c);
        HitRec h;
        const bool found = trace_closest<COUNT>(K, P.o, P.d, h, bstk, tstk, c);
        advance<COUNT>(K, P, found, h, c);
#endif
    }
#ifdef RZ_PROF
    if (COUNT) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) { atomicAdd(&pr[16], tBegin); atomicAdd(&pr[17], tTrace); atomicAdd(&pr[18], tAdv); }
    }
#endif
    if (inside) {
        K.accum[pix] = make_float4(P.color.x, P.color.y, P.color.z, alpha + (float)K.spp);
        K.ior[pix] = P.ior;
    }
    if (COUNT) {
        DevCounters* g = K.counters;
        atomicAdd(&g->samples, (unsigned long long)c.samples);
        atomicAdd(&g->traversals, (unsigned long long)c.traversals);
        atomicAdd(&g->tlas_nodes, (unsigned long long)c.tlas_nodes);
        atomicAdd(&g->tlas_leaf_indices, (unsigned long long)c.tlas_leaf_indices);
        atomicAdd(&g->instances, (unsigned long long)c.instances);
        atomicAdd(&g->blas_nodes, (unsigned long long)c.blas_nodes);
        atomicAdd(&g->triangles, (unsigned long long)c.triangles);
        atomicAdd(&g->materials, (unsigned long long)c.materials);
        atomicAdd(&g->light_fetches, (unsigned long long)c.light_fetches);
        if (inside) atomicAdd(&g->pixels, 1ull);
    }
}

// FS:772-773 + 8-bit quantisation: rgba8 = round(clamp(sum / n, 0, 1) * 255), a = 255.
__global__ void rz_resolve_kernel(const float4* __restrict__ accum, uchar4* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = accum[i];
    const float cnt = a.w > 0.0f ? a.w : 1.0f;
    const float r = clamp_(a.x / cnt, 0.0f, 1.0f), g = clamp_(a.y / cnt, 0.0f, 1.0f), b = clamp_(a.z / cnt, 0.0f, 1.0f);
    out[i] = make_uchar4((unsigned char)__builtin_rintf(r * 255.0f), (unsigned char)__builtin_rintf(g * 255.0f),
                         (unsigned char)__builtin_rintf(b * 255.0f), 255);
}

void launch_render_pixels(const KParams& K, bool counted, hipStream_t stream) {
    const int blocks = (K.nLocalTiles + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks <= 0) return;
    const size_t perWave = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int);
    const size_t lds = perWave * WAVES_PER_BLOCK;
    if (counted)
        hipLaunchKernelGGL(rz_render_pixels<true>, dim3(blocks), dim3(WAVES_PER_BLOCK * 64), lds, stream, K);
    else
        hipLaunchKernelGGL(rz_render_pixels<false>, dim3(blocks), dim3(WAVES_PER_BLOCK * 64), lds, stream, K);
}

void launch_resolve(const float4* accum, uchar4* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(rz_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, out, n);
}

}  // namespace rz
