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

#include <algorithm>
#include <cstdio>
#include <vector>

#include "rayzen_hip.h"
#include "rz_path.h"

namespace rz {

#ifndef RZ_WAVES_PER_BLOCK
#define RZ_WAVES_PER_BLOCK 1   // measured on C2: 4 -> 98.9 ms, 2 -> 88.5 ms, 1 -> 88.0 ms (a 4-wave group holds its CU slots until its slowest tile ends)
#endif
constexpr int WAVES_PER_BLOCK = RZ_WAVES_PER_BLOCK;
#ifndef RZ_MIN_WAVES_PER_SIMD
#define RZ_MIN_WAVES_PER_SIMD 2
#endif

#ifdef RZ_PROF
__device__ unsigned long long rz_wave_log[1 << 17][3];     // diagnostic build: start, end (100 MHz ticks), hw id
#endif

template <bool COUNT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, RZ_MIN_WAVES_PER_SIMD) void rz_render_pixels(const KParams K) {
#ifdef RZ_PROF
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
#endif
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
    if (lane == 0 && localTile < (1 << 17)) {
        rz_wave_log[localTile][0] = wl_t0;
        rz_wave_log[localTile][1] = __builtin_amdgcn_s_memrealtime();
        rz_wave_log[localTile][2] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 0) | (0 << 6) | (31 << 11)) |   // HW_REG_HW_ID? (id 4 on gfx9)
                                    ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // XCC_ID
    }
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

// ---------------------------------------------------------------------------------------------------------
// rz_render_samples<COUNT>: one lane per SAMPLE.  Valid when no triangle of the scene uses a transparent
// material: then FS:674's currentIor never leaves 1.0 and the only coupling between a pixel's samples is the ORDER
// of the colour additions (FS:717 then FS:709, sample after sample).  Each lane runs one sample's path and stores
// its two addends; rz_sum_samples replays the additions in the shader's order, so the sum is bit-identical.
//   item = slot * chunkSpp + s (s fastest): a wave's 64 lanes are consecutive samples of one pixel (spp >= 64), so
//   primary and shadow rays of a wave are near-identical -- uniform traversal, L1 broadcast instead of 64
//   divergent lines -- and a heavy pixel costs ONE sample's latency instead of spp of them: no tail.
#ifndef RZ_SAMPLES_MIN_WAVES
#define RZ_SAMPLES_MIN_WAVES 3   // measured on C2: 2 -> 29.0 ms, 3 -> 22.1 ms, 4 -> 22.1 ms (the kernel is issue-bound; a third wave overlaps scalar/VMEM/LDS issue with VALU)
#endif
template <bool COUNT>
__global__ __launch_bounds__(64, RZ_SAMPLES_MIN_WAVES) void rz_render_samples(const KParams K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    uint2* bstk = reinterpret_cast<uint2*>(lds_raw) + lane;
    int* tstk = reinterpret_cast<int*>(lds_raw + (size_t)K.blasStackCap * 64 * sizeof(uint2)) + lane;
    const long long item = (long long)blockIdx.x * 64 + lane;
    const long long nItems = (long long)K.nSlots * K.chunkSpp;
    Tally c = {};
    Path P;
    P.mode = MODE_DONE;
    bool active = false;
    if (item < nItems) {
        const int slot = (int)(item / K.chunkSpp), s = (int)(item - (long long)slot * K.chunkSpp);
        const int localTile = slot >> 6, l = slot & 63;
        const int tile = localTile * K.tileNRanks + K.tileRank;
        const int tx = tile % K.tilesX, ty = tile / K.tilesX;
        const int px = tx * RZ_TILE_W + (l & 7), py = ty * RZ_TILE_H + (l >> 3);
        if (px < K.width && py < K.height) {
            active = true;
            const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
            P.uv.x = fragx / (float)K.width;
            P.uv.y = fragy / (float)K.height;
            P.fragSum = fragx + fragy;
            P.color = mk3(0.0f, 0.0f, 0.0f);
            P.ior = 1.0f;
            P.samp = K.sampleBase + s;
            begin_sample<COUNT>(K, P, c);
        }
    }
#ifdef RZ_PROF
    unsigned long long tTrace = 0, tAdv = 0;
    while (P.mode != MODE_DONE) {
        RZ_SITE(c, 6);
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        HitRec h;
        const bool found = trace_closest<COUNT>(K, P.o, P.d, h, bstk, tstk, c);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        advance<COUNT, false>(K, P, found, h, c);
        tTrace += t2 - t1; tAdv += __builtin_amdgcn_s_memtime() - t2;
    }
    if (COUNT) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) if (c.p[k]) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) { atomicAdd(&pr[17], tTrace); atomicAdd(&pr[18], tAdv); }
    }
#else
    while (P.mode != MODE_DONE) {
        HitRec h;
        const bool found = trace_closest<COUNT>(K, P.o, P.d, h, bstk, tstk, c);
        advance<COUNT, false>(K, P, found, h, c);
    }
#endif
    if (active) {
        K.contrib[2 * item] = make_float4(P.addLight.x, P.addLight.y, P.addLight.z, 0.0f);
        K.contrib[2 * item + 1] = make_float4(P.addSky.x, P.addSky.y, P.addSky.z, 0.0f);
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
    }
}

// Replays FS:709/717's `color +=` in sample order for every owned pixel (one lane per pixel), starting from the
// colour already in the accumulation buffer.  first: this is the first chunk of the frame (sample_base == 0 and
// chunk 0) -> start from zero.  last: add spp to the sample count.
__global__ __launch_bounds__(256) void rz_sum_samples(const KParams K, const int first, const int countPixels) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= K.nSlots) return;
    const int localTile = slot >> 6, l = slot & 63;
    const int tile = localTile * K.tileNRanks + K.tileRank;
    const int tx = tile % K.tilesX, ty = tile / K.tilesX;
    const int px = tx * RZ_TILE_W + (l & 7), py = ty * RZ_TILE_H + (l >> 3);
    if (px >= K.width || py >= K.height) return;
    const size_t pix = (size_t)py * K.width + px;
    float4 a = first ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : K.accum[pix];
    const float4* __restrict__ cp = K.contrib + 2 * (size_t)slot * K.chunkSpp;
    for (int s = 0; s < K.chunkSpp; ++s) {
        const float4 L = cp[2 * s], S = cp[2 * s + 1];
        a.x = a.x + L.x; a.y = a.y + L.y; a.z = a.z + L.z;
        a.x = a.x + S.x; a.y = a.y + S.y; a.z = a.z + S.z;
    }
    a.w += (float)K.chunkSpp;
    K.accum[pix] = a;
    K.ior[pix] = 1.0f;
    if (countPixels) atomicAdd(&K.counters->pixels, 1ull);
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

#ifdef RZ_PROF
void dump_wave_log(int nWaves) {
    static std::vector<unsigned long long> h;
    h.resize((size_t)3 * (1 << 17));
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(rz_wave_log), h.size() * 8) != hipSuccess) return;
    nWaves = std::min(nWaves, 1 << 17);
    unsigned long long t0 = ~0ull, t1 = 0, sum = 0;
    std::vector<double> dur(nWaves);
    for (int i = 0; i < nWaves; ++i) {
        t0 = std::min(t0, h[3 * i]); t1 = std::max(t1, h[3 * i + 1]);
        sum += h[3 * i + 1] - h[3 * i]; dur[i] = (double)(h[3 * i + 1] - h[3 * i]) * 1e-5;   // ms (100 MHz)
    }
    std::vector<double> sorted = dur; std::sort(sorted.begin(), sorted.end());
    double span = (double)(t1 - t0) * 1e-5;
    fprintf(stderr, "[rz_prof] waves %d  span %.2f ms  sum of lifetimes %.1f ms  avg resident waves %.0f  wave ms: min %.3f med %.3f p90 %.3f p99 %.3f max %.3f\n",
            nWaves, span, (double)sum * 1e-5, (double)sum * 1e-5 / span, sorted[0], sorted[nWaves / 2], sorted[nWaves * 9 / 10], sorted[nWaves * 99 / 100], sorted[nWaves - 1]);
    // residency over time: 20 slices
    const int NS = 20; double res[NS] = {0};
    for (int i = 0; i < nWaves; ++i) {
        double a = (double)(h[3 * i] - t0) * 1e-5, b = (double)(h[3 * i + 1] - t0) * 1e-5;
        for (int s = 0; s < NS; ++s) { double lo = span * s / NS, hi = span * (s + 1) / NS; double o = std::min(b, hi) - std::max(a, lo); if (o > 0) res[s] += o / (hi - lo); }
    }
    fprintf(stderr, "[rz_prof] resident waves per 5%% time slice:");
    for (int s = 0; s < NS; ++s) fprintf(stderr, " %.0f", res[s]);
    fprintf(stderr, "\n");
}
#endif

void launch_render_samples(const KParams& K, bool counted, hipStream_t stream) {
    const long long nItems = (long long)K.nSlots * K.chunkSpp;
    const long long blocks = (nItems + 63) / 64;
    if (blocks <= 0) return;
    const size_t lds = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int);
    if (counted)
        hipLaunchKernelGGL(rz_render_samples<true>, dim3((unsigned)blocks), dim3(64), lds, stream, K);
    else
        hipLaunchKernelGGL(rz_render_samples<false>, dim3((unsigned)blocks), dim3(64), lds, stream, K);
}

void launch_sum_samples(const KParams& K, bool first, bool countPixels, hipStream_t stream) {
    if (K.nSlots <= 0) return;
    hipLaunchKernelGGL(rz_sum_samples, dim3((K.nSlots + 255) / 256), dim3(256), 0, stream, K, first ? 1 : 0,
                       countPixels ? 1 : 0);
}

void launch_resolve(const float4* accum, uchar4* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(rz_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, out, n);
}

}  // namespace rz
