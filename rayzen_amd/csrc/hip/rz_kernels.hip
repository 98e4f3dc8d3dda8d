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
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "rayzen_hip.h"
#include "rz_path.h"
#include "rz_internal.h"

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
#ifdef RZ_GSTATS
__device__ unsigned long long rz_gstats[8];                // diagnostic build (transparent scenes on claims): [0] units [1] units resolved in the wave [2] groups rendered again [3] re-run rounds [4] paths parked [5] pooled paths that met glass
#define RZ_GSTAT(i, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&rz_gstats[i], (unsigned long long)(n)); } while (0)
#else
#define RZ_GSTAT(i, n) do { } while (0)
#endif

template <bool COUNT>
__global__ __launch_bounds__(WAVES_PER_BLOCK * 64, RZ_MIN_WAVES_PER_SIMD) void rz_render_pixels(const KParams K) {
#ifdef RZ_PROF
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // LDS: per wave, [blasStackCap][64] uint2 (tlasStackCap is 0: the TLAS walk keeps no stack)
    const size_t perWave = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int);
    unsigned char* base = lds_raw + perWave * wave;
    const BlasStackT<false> bstk{reinterpret_cast<uint2*>(base) + lane, nullptr, K.blasStackCap};

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
        const bool found = trace_closest<COUNT, false>(K, P.o, P.d, h, bstk, c);
        unsigned long long t2 = __builtin_amdgcn_s_memtime();
        advance<COUNT>(K, P, found, h, c);
        unsigned long long t3 = __builtin_amdgcn_s_memtime();
        tBegin += t1 - t0; tTrace += t2 - t1; tAdv += t3 - t2;
#else
        if (P.mode == MODE_DONE) begin_sample<COUNT>(K, P, c);
        HitRec h;
        const bool found = trace_closest<COUNT, false>(K, P.o, P.d, h, bstk, c);
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
        atomicAdd(&g->scatters, (unsigned long long)c.scatters);
        atomicAdd(&g->diffuse_scatters, (unsigned long long)c.diffuse_scatters);
        atomicAdd(&g->hemi_draws, (unsigned long long)c.hemi_draws);
        atomicAdd(&g->lit_lights, (unsigned long long)c.lit_lights);
        atomicAdd(&g->triangles_past_u, (unsigned long long)c.triangles_past_u);
        if (inside) atomicAdd(&g->pixels, 1ull);
    }
}

// ---------------------------------------------------------------------------------------------------------
// rz_render_samples<COUNT, GLASS>: one lane per SAMPLE.  The shader couples a pixel's samples in two ways only: the
// ORDER of the colour additions (FS:717 then FS:709, sample after sample) and FS:674's currentIor, which is declared
// outside the sample loop and is read / written only where a path scatters at a transparent surface (FS:727-742).
//   GLASS = false (no triangle uses a transparent material): currentIor never leaves 1.0; samples are independent.
//   GLASS = true: the samples of a pixel run in parallel SPECULATING the incoming currentIor; each lane records
//     whether it read it (usedIor) and what it left (P.ior).  A ballot scan then walks the samples in order: the
//     first sample that read currentIor after an earlier one changed it was computed from a wrong value -- every
//     sample before it is final, it and its successors are re-run with the corrected value, and so on.  A lane's
//     final result is computed from exactly the currentIor the sequential shader would have given it.
//   * a wavefront owns whole pixels: ONE pixel, walked in batches of 64 samples, when spp >= 64; floor(64/spp)
//     pixels otherwise.  Its lanes are therefore samples of the same pixel(s): primary and shadow rays of a wave
//     are near-identical -- uniform traversal, scalar-cache node fetches (rz_trace.h) -- and a heavy pixel costs
//     ONE sample's latency per batch instead of spp of them in a row: the 84-ms tail of the per-pixel kernel is gone;
//   * each lane runs one sample's path and parks its two addends in LDS (2 KB per wave); one lane per pixel then
//     replays `color += light; color += sky` in sample order, so the sum is bit-identical to the shader's.
//     Nothing but the final pixel value goes to HBM.
// Waves per SIMD asked of the compiler (the kernel is issue-bound: another resident wave overlaps scalar / VMEM / LDS
// issue with VALU).  Measured on C2: 2 -> 29.0 ms, 3 -> 22.1 ms; 4 paid off only once the SLP vectoriser was switched
// off (build.py: it paired f32 ops into 64-bit register tuples and cost 25 VGPRs): 142 VGPRs -> 128 with 13 spilled,
// 18.65 -> 17.65 ms.  The transparent variant first stayed at 3 (its LDS capped occupancy at 11 waves per CU, so the
// 28 registers spilled at 128 VGPRs bought nothing: 49.4 -> 54.3 ms); with the BLAS stack cut to an LDS window
// (rz_context.hip: render_samples) 16 waves fit and 4 wins: 40.2 -> 34.3 ms.
#ifndef RZ_SAMPLES_MIN_WAVES
#define RZ_SAMPLES_MIN_WAVES 4
#endif
#ifndef RZ_SAMPLES_MIN_WAVES_GLASS
#define RZ_SAMPLES_MIN_WAVES_GLASS 4
#endif
// Slot l of a tile -> pixel (x, y) in the tile.  One pixel per wave (spp >= 64): row-major, a claim of 8 slots is one
// row of the tile.  Several pixels per wave (spp < 64): Morton order (x takes the even bits of l, y the odd ones), so the
// pixels that share a wave form a compact block -- 2x2 at 16 spp, not a 4x1 strip -- and their rays start closer to each
// other: C4 9.30 -> 9.19 ms.  (Morton order for everything moved C2 / C5 / C3 by +0.2 ... +0.5 %: the row-shaped claim
// is at least as good for the compacting launch.)
__device__ __forceinline__ int slot_x(int l, bool morton) { return morton ? ((l & 1) | ((l >> 1) & 2) | ((l >> 2) & 4)) : (l & 7); }
__device__ __forceinline__ int slot_y(int l, bool morton) { return morton ? (((l >> 1) & 1) | ((l >> 2) & 2) | ((l >> 3) & 4)) : (l >> 3); }
#ifdef RZ_PROF
// per-query-round site counters and trace cycles -> pr[32 + 11 r + k] (k < 10: rp, k == 10: rt of lane 0)
__device__ __forceinline__ void rz_prof_rounds(const Tally& c, unsigned long long* pr) {
    for (int k = 0; k < 8; ++k) {       // descend steps by how many children they enter -> pr[132 + k]
        unsigned x = c.ps[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&pr[132 + k], (unsigned long long)x);
    }
    for (int r = 0; r < 8; ++r) {
        for (int k = 0; k < 10; ++k) {
            unsigned x = c.rp[r][k];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&pr[32 + 11 * r + k], (unsigned long long)x);
        }
        if ((threadIdx.x & 63) == 0 && c.rt[r]) atomicAdd(&pr[32 + 11 * r + 10], c.rt[r]);
    }
}
#endif
__device__ __forceinline__ void tally_add(Tally& a, const Tally& b) {
    a.traversals += b.traversals; a.tlas_nodes += b.tlas_nodes; a.tlas_leaf_indices += b.tlas_leaf_indices;
    a.instances += b.instances; a.blas_nodes += b.blas_nodes; a.triangles += b.triangles; a.materials += b.materials;
    a.light_fetches += b.light_fetches; a.samples += b.samples;
    a.triangles_past_u += b.triangles_past_u; a.scatters += b.scatters; a.diffuse_scatters += b.diffuse_scatters; a.hemi_draws += b.hemi_draws; a.lit_lights += b.lit_lights;
#ifdef RZ_PROF
    for (int k = 0; k < 16; ++k) a.p[k] += b.p[k];
    for (int k = 0; k < 8; ++k) a.ps[k] += b.ps[k];
    for (int k = 0; k < 20; ++k) a.t[k] += b.t[k];
    for (int r = 0; r < 8; ++r) { a.rt[r] += b.rt[r]; for (int k = 0; k < 10; ++k) a.rp[r][k] += b.rp[r][k]; }
#endif
}

// A wave's tallies to the launch's counters (the counting instantiations only): one atomic per counter per wave, reduced across
// the lanes first.  DevCounters is an array of 64-bit words in this order; `pixels` (word 9) is added by the code that stores pixels.
__device__ __forceinline__ void tally_flush(const KParams& K, const Tally& c) {
    const unsigned v[14] = {c.samples, c.traversals, c.tlas_nodes, c.tlas_leaf_indices, c.instances, c.blas_nodes,
                            c.triangles, c.materials, c.light_fetches, c.scatters, c.diffuse_scatters, c.hemi_draws, c.lit_lights, c.triangles_past_u};
    unsigned long long* g = reinterpret_cast<unsigned long long*>(K.counters);
    for (int k = 0; k < 14; ++k) {
        unsigned x = v[k];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&g[k < 9 ? k : k + 1], (unsigned long long)x);
    }
}

// Persistent waves: the grid is RZ_PERSIST_WAVES_PER_CU waves per CU (>= what fits) and each wave claims
// RZ_GROUPS_PER_CLAIM consecutive pixel groups at a time from a per-launch counter, instead of one workgroup per pixel:
// at 1080p that was 2 M workgroup launches per frame, each paying dispatch, LDS allocation and kernarg loads for as
// little as one sky lookup.  Measured on C2 (17.2 ms as one workgroup per pixel): static striding 26 / 21 / 18.8 ms
// at 16 / 32 / 64 waves per CU (whole extra rounds: 15, not 16, waves are resident); claiming 2 / 4 / 6 / 8 / 12 /
// 16 / 32 groups: 20.0 / 17.4 / 17.0 / 16.8 / 16.9 / 17.2 / 18.8 ms -- small claims run into the ~88 atomics/us a
// single word sustains, large ones leave a tail.  8 groups = one row of an 8x8 tile.
#ifndef RZ_PERSIST_WAVES_PER_CU
#define RZ_PERSIST_WAVES_PER_CU 16
#endif
#ifndef RZ_GROUPS_PER_CLAIM
#define RZ_GROUPS_PER_CLAIM 8
#endif
#ifndef RZ_GROUPS_PER_CLAIM_SMALL_SPP
#define RZ_GROUPS_PER_CLAIM_SMALL_SPP 6     // (plan_render_samples: launches of several pixels per wave below 3/4 M groups)
#endif
// trace_spread is compiled into the SPREAD flavour of the group code only (COMPACT == 1 of rz_render_samples: launches of
// several pixels per wave): merely present in the 64-spp transparent variant it cost 8 % (c2g 24.6 -> 26.9 ms: registers).
#define RZ_SPREAD_ON(K) (SPREAD && (K).spreadTrace != 0)
#ifndef RZ_CLAIM_RUN_SMALL_SPP
#define RZ_CLAIM_RUN_SMALL_SPP 1
#endif
#ifndef RZ_PARK_BOUNCE
#define RZ_PARK_BOUNCE 2         // a path is parked when it stands in front of this segment (0-based: its third)
#endif
#ifndef RZ_DARK_UNITS
#define RZ_DARK_UNITS 1         // units without a light term keep their three light rows unwritten (ordered_sum_pass: zero_row)
#endif
#ifndef RZ_COMPACT_DEFAULT
#define RZ_COMPACT_DEFAULT 1     // compaction of late bounces across a claim (render_claim_compact); RZ_COMPACT=0/1 overrides at run time
#endif
template <bool COUNT, bool GLASS, bool OVF, bool SPREAD>
__device__ __forceinline__ void render_samples_group(const KParams& K, const unsigned wblock, unsigned char* lds_raw) {
    const int lane = threadIdx.x & 63;
    // (the overflow columns are indexed by the RESIDENT workgroup: blasOvfCap > 0 only in persistent launches)
    const BlasStackT<OVF> bstk{reinterpret_cast<uint2*>(lds_raw) + lane,
                               OVF ? K.blasOvf + ((size_t)blockIdx.x * K.blasOvfCap) * 64 + lane : nullptr, K.blasStackCap};
    float4* addL = reinterpret_cast<float4*>(lds_raw + (size_t)K.blasStackCap * 64 * sizeof(uint2) +
                                             (size_t)K.tlasStackCap * 64 * sizeof(int));

    const int spp = K.spp;
    const int pixPerWave = spp >= 64 ? 1 : 64 / spp;
    const int nBatches = spp >= 64 ? (spp + 63) / 64 : 1;
    // this lane's pixel (as a path-tracing lane) and this lane's pixel as a summing lane (lanes 0..pixPerWave-1)
    const int myPixInWave = spp >= 64 ? 0 : lane / spp;
    const int slot = (int)wblock * pixPerWave + myPixInWave;
    const int sumSlot = (int)wblock * pixPerWave + lane;
    bool inside = false, sumInside = false;
    size_t sumPix = 0;
    Path P;
    if (slot < K.nSlots && myPixInWave < pixPerWave) {
        const int localTile = slot >> 6, l = slot & 63;
        const int tile = localTile * K.tileNRanks + K.tileRank;
        const int tx = tile % K.tilesX, ty = tile / K.tilesX;
        const int px = tx * RZ_TILE_W + slot_x(l, spp < 64), py = ty * RZ_TILE_H + slot_y(l, spp < 64);
        if (px < K.width && py < K.height) {
            inside = true;
            const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
            P.uv.x = fragx / (float)K.width;
            P.uv.y = fragy / (float)K.height;
            P.fragSum = fragx + fragy;
        }
    }
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lane < pixPerWave && sumSlot < K.nSlots) {
        const int localTile = sumSlot >> 6, l = sumSlot & 63;
        const int tile = localTile * K.tileNRanks + K.tileRank;
        const int tx = tile % K.tilesX, ty = tile / K.tilesX;
        const int px = tx * RZ_TILE_W + slot_x(l, spp < 64), py = ty * RZ_TILE_H + slot_y(l, spp < 64);
        if (px < K.width && py < K.height) {
            sumInside = true;
            sumPix = (size_t)py * K.width + px;
            if (K.sampleBase != 0) acc = K.accum[sumPix];
        }
    }
    // currentIor entering this pixel's next sample, held by the pixel's summing lane and fetched by its path lanes
    float iorPix = (GLASS && sumInside && K.sampleBase != 0) ? K.ior[sumPix] : 1.0f;
    const int pixLane = spp >= 64 ? 0 : myPixInWave;         // the summing lane of this path lane's pixel
    // spp >= 64: the wave's single pixel, summed per channel by lanes 0..2 (lane 3 keeps the sample count)
    const bool sumInside0 = __shfl((int)sumInside, 0) != 0;
    const size_t sumPix0 = ((size_t)(unsigned)__shfl((int)(sumPix >> 32), 0) << 32) | (unsigned)__shfl((int)(unsigned)sumPix, 0);
    float chan = 0.0f, alpha0 = 0.0f;
    if (spp >= 64 && sumInside0 && K.sampleBase != 0) {
        const float ax = __shfl(acc.x, 0), ay = __shfl(acc.y, 0), az = __shfl(acc.z, 0);
        chan = lane == 0 ? ax : (lane == 1 ? ay : az);
        alpha0 = __shfl(acc.w, 0);
    }
    Tally c = {};
#ifdef RZ_PROF
    unsigned long long tTrace = 0, tAdv = 0;
#endif
    for (int b = 0; b < nBatches; ++b) {
        const int s = spp >= 64 ? b * 64 + lane : lane - myPixInWave * spp;
        const bool mine = inside && s < spp;          // this lane has a sample in this batch
        P.mode = MODE_DONE;
        P.addLight = mk3(0.0f, 0.0f, 0.0f);
        P.addSky = mk3(0.0f, 0.0f, 0.0f);
        P.usedIor = 0;
        P.ior = 1.0f;
        if constexpr (!GLASS) {
            if (mine) {
                P.color = mk3(0.0f, 0.0f, 0.0f);
                P.samp = K.sampleBase + s;
                begin_sample<COUNT>(K, P, c);
            }
            // (a wave-uniform loop with a predicated body and one exit, not a per-lane `while`: see blas_walk)
            bool anyRun = rz_ballot(P.mode != MODE_DONE) != 0ull;
#ifdef RZ_PROF
            c.rnd = 0;
#endif
            while (anyRun) {
#ifdef RZ_PROF
                const unsigned long long t1 = __builtin_amdgcn_s_memtime();
                unsigned long long t2 = t1;
#endif
                // the wave's rays have spread once every running lane is on its third or a later segment (rz_trace.h: trace_spread)
                const bool spread = RZ_SPREAD_ON(K) && rz_ballot(P.mode != MODE_DONE && !(P.mode == MODE_SEGMENT && P.bounce >= 2)) == 0ull;
                if (P.mode != MODE_DONE) {
                    RZ_SITE(c, 6);
                    HitRec h;
                    const bool found = spread ? trace_spread<COUNT, OVF>(K, P.o, P.d, h, bstk, c) : trace_closest<COUNT, OVF>(K, P.o, P.d, h, bstk, c);
#ifdef RZ_PROF
                    t2 = __builtin_amdgcn_s_memtime();
#endif
                    advance<COUNT, false>(K, P, found, h, c);
                }
#ifdef RZ_PROF
                tTrace += t2 - t1; tAdv += __builtin_amdgcn_s_memtime() - t2;
                c.rt[c.rnd & 7] += t2 - t1;
                if (c.rnd < 7) ++c.rnd;
#endif
                anyRun = rz_ballot(P.mode != MODE_DONE) != 0ull;
            }
            // park the addends (zeros for idle lanes: adding +0 is exact), then replay the adds in sample order
            // (channel-major [3][64] + [3][64] floats: 1.5 KB, and the channel lanes below read consecutive words)
            float* const aL = reinterpret_cast<float*>(addL);
            float* const aS = aL + 3 * 64;
            aL[lane] = P.addLight.x; aL[64 + lane] = P.addLight.y; aL[128 + lane] = P.addLight.z;
            aS[lane] = P.addSky.x; aS[64 + lane] = P.addSky.y; aS[128 + lane] = P.addSky.z;
            __syncthreads();
            if (spp >= 64) {
                // one pixel per wave: the three colour channels are independent chains -> lanes 0,1,2 take one each
                // (a 128-add dependent chain per batch instead of 384 on one lane)
                if (lane < 3 && sumInside0) {
                    const float* Lf = aL + 64 * lane;
                    const float* Sf = aS + 64 * lane;
                    const int n = min(64, spp - b * 64);
                    int k = 0;
                    for (; k + 8 <= n; k += 8) {
                        float l[8], q[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { l[u] = Lf[k + u]; q[u] = Sf[k + u]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { chan = chan + l[u]; chan = chan + q[u]; }   // FS:717, FS:709
                    }
                    for (; k < n; ++k) { chan = chan + Lf[k]; chan = chan + Sf[k]; }
                }
            } else if (sumInside) {
                const int first = lane * spp;
                for (int k = 0; k < spp; ++k) {
                    const int e = first + k;
                    acc.x = acc.x + aL[e]; acc.y = acc.y + aL[64 + e]; acc.z = acc.z + aL[128 + e];     // FS:717
                    acc.x = acc.x + aS[e]; acc.y = acc.y + aS[64 + e]; acc.z = acc.z + aS[128 + e];     // FS:709
                }
            }
            __syncthreads();
        } else {
            // ---- transparent scenes: memoised speculation on currentIor (FS:674).
            // A sample's result depends on the currentIor it starts from only if it scatters at a transparent surface
            // (usedIor).  Each lane keeps up to two VERSIONS of its sample in LDS, keyed by the incoming value it was
            // computed from.  The pixel's summing lane consumes samples in order, taking for each the version whose key
            // equals the currentIor left by the previous one (any version will do if the sample never read it), and
            // stops at the first sample that lacks the version it needs; exactly the lanes that lack the wanted
            // version are then (re)run.  currentIor only ever takes the values 1.0 and the ior of a transparent
            // material, so with one glass material every sample is computed at most twice.  Keys are compared by BIT
            // PATTERN: a NaN ior (the shader just propagates it) would never equal itself as a float and the loop below
            // would re-run the same sample for ever.
            float4* verL = addL;                                  // [2][64]: addend FS:717, .w = incoming ior (key)
            float4* verS = addL + 128;                            // [2][64]: addend FS:709, .w = outgoing ior
            int* vinfo = reinterpret_cast<int*>(addL + 256);      // [2][64]: bit0 valid, bit1 usedIor
            int* chosen = vinfo + 128;                            // [64]: version the summing lane consumed
            vinfo[lane] = 0; vinfo[64 + lane] = 0;
            float keyv[2] = {0.0f, 0.0f};
            int infov[2] = {0, 0};
            int nextSlot = 0;
            int consumed = 0;                                     // summing lanes: samples of my pixel already added
            const int nMine = spp >= 64 ? min(64, spp - b * 64) : spp;
            const int firstLane = spp >= 64 ? 0 : lane * spp;     // summing lanes: first lane of my pixel
            Tally tv[2] = {};
            bool firstPass = true;
            __syncthreads();
            for (;;) {
                const int cons = __shfl(consumed, pixLane);
                const float want = __shfl(iorPix, pixLane);
                const int sIdx = spp >= 64 ? lane : s;            // index of my sample within its pixel's batch
                bool have = false;
#pragma unroll
                for (int v = 0; v < 2; ++v) have = have || ((infov[v] & 1) && (!(infov[v] & 2) || __float_as_uint(keyv[v]) == __float_as_uint(want)));
                const bool run = mine && sIdx >= cons && !have;
                Tally att = {};
                if (run) {
                    P.color = mk3(0.0f, 0.0f, 0.0f);
                    P.ior = want;
                    P.samp = K.sampleBase + s;
                    if (firstPass || K.snap == nullptr) {
                        begin_sample<COUNT>(K, P, COUNT ? att : c);
                    } else {
                        // a second version: only a sample that READ currentIor is ever run again, and its first run has left the
                        // state in front of that read -- its first transparent scatter -- in the wave's scratch (rz_path.h:
                        // snapshot_store): camera ray, queries and first-hit lighting up to there do not depend on currentIor
                        snapshot_load<COUNT>(K, P, COUNT ? att : c);
                        scatter<COUNT, true, false>(K, P, COUNT ? att : c);
                    }
                }
                // (a per-lane loop here: as a wave-uniform loop with a predicated body, which pays in the opaque variant, this
                //  one lost 3 % -- 26.1 -> 27.0 ms on the glass + mirror scene)
#ifdef RZ_PROF
                c.rnd = 0;
#endif
                while (P.mode != MODE_DONE) {
                    HitRec h;
#ifdef RZ_PROF
                    RZ_SITE(c, 6);
                    const unsigned long long tq0_ = __builtin_amdgcn_s_memtime();
                    const bool found = (RZ_SPREAD_ON(K) && rz_ballot(!(P.mode == MODE_SEGMENT && P.bounce >= 2)) == 0ull)
                                           ? trace_spread<false, OVF>(K, P.o, P.d, h, bstk, c) : trace_closest<false, OVF>(K, P.o, P.d, h, bstk, c);
                    c.rt[c.rnd & 7] += __builtin_amdgcn_s_memtime() - tq0_;
                    if (c.rnd < 7) ++c.rnd;
                    advance<false, true, true>(K, P, found, h, c);
#else
                    // (the lanes still in this loop have spread once they all are on their third or a later segment: rz_trace.h, trace_spread)
                    const bool found = (RZ_SPREAD_ON(K) && rz_ballot(!(P.mode == MODE_SEGMENT && P.bounce >= 2)) == 0ull)
                                           ? trace_spread<COUNT, OVF>(K, P.o, P.d, h, bstk, COUNT ? att : c) : trace_closest<COUNT, OVF>(K, P.o, P.d, h, bstk, COUNT ? att : c);
                    advance<COUNT, true, true>(K, P, found, h, COUNT ? att : c);
#endif
                }
                if (run) {
                    const int slot = nextSlot;
                    nextSlot ^= 1;
                    keyv[slot] = want;
                    infov[slot] = 1 | (P.usedIor ? 2 : 0);
                    verL[slot * 64 + lane] = make_float4(P.addLight.x, P.addLight.y, P.addLight.z, want);
                    verS[slot * 64 + lane] = make_float4(P.addSky.x, P.addSky.y, P.addSky.z, P.ior);
                    vinfo[slot * 64 + lane] = infov[slot];
                    if (COUNT) tv[slot] = att;
                }
                __syncthreads();
                if (firstPass && spp >= 64 && rz_ballot(mine && (infov[0] & 2)) == 0ull) {
                    // Nobody in this batch read currentIor (the usual case: most pixels never meet glass): every
                    // sample's only version is final, so add them as the opaque kernel does -- one colour channel per
                    // lane, unrolled -- instead of one lane walking 64 keyed entries (that walk alone was a third of
                    // the variant's instructions).  Same additions in the same order.
                    const float ax = __shfl(acc.x, 0), ay = __shfl(acc.y, 0), az = __shfl(acc.z, 0);
                    float ch = lane == 0 ? ax : (lane == 1 ? ay : az);
                    if (lane < 3 && sumInside0) {
                        const float* Lf = reinterpret_cast<const float*>(verL) + lane;
                        const float* Sf = reinterpret_cast<const float*>(verS) + lane;
                        int k = 0;
                        for (; k + 8 <= nMine; k += 8) {
                            float l[8], q[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { l[u] = Lf[4 * (k + u)]; q[u] = Sf[4 * (k + u)]; }
#pragma unroll
                            for (int u = 0; u < 8; ++u) { ch = ch + l[u]; ch = ch + q[u]; }   // FS:717, FS:709
                        }
                        for (; k < nMine; ++k) { ch = ch + Lf[4 * k]; ch = ch + Sf[4 * k]; }
                    }
                    const float cx = __shfl(ch, 0), cy = __shfl(ch, 1), cz = __shfl(ch, 2);
                    if (sumInside) { acc.x = cx; acc.y = cy; acc.z = cz; consumed = nMine; }
                    chosen[lane] = 0;
                }
                firstPass = false;
                if (sumInside) {
                    while (consumed < nMine) {
                        const int k = firstLane + consumed;
                        int pick = -1;
#pragma unroll
                        for (int v = 0; v < 2; ++v) {
                            const int inf = vinfo[v * 64 + k];
                            if (pick < 0 && (inf & 1) && (!(inf & 2) || __float_as_uint(verL[v * 64 + k].w) == __float_as_uint(iorPix))) pick = v;
                        }
                        if (pick < 0) break;
                        const float4 L = verL[pick * 64 + k], S = verS[pick * 64 + k];
                        acc.x = acc.x + L.x; acc.y = acc.y + L.y; acc.z = acc.z + L.z;     // FS:717
                        acc.x = acc.x + S.x; acc.y = acc.y + S.y; acc.z = acc.z + S.z;     // FS:709
                        if (vinfo[pick * 64 + k] & 2) iorPix = S.w;                        // FS:742
                        chosen[k] = pick;
                        ++consumed;
                    }
                }
                __syncthreads();
                if (rz_ballot(sumInside && consumed < nMine) == 0ull) break;
            }
            if (COUNT && mine) tally_add(c, tv[chosen[lane]]);
            __syncthreads();
        }
    }
    if (!GLASS && spp >= 64) {
        const float cx = __shfl(chan, 0), cy = __shfl(chan, 1), cz = __shfl(chan, 2);
        if (sumInside0 && lane == 0) {          // one whole 16-B pixel store
            K.accum[sumPix0] = make_float4(cx, cy, cz, alpha0 + (float)spp);
            K.ior[sumPix0] = 1.0f;
        }
    } else if (sumInside) {
        acc.w += (float)spp;
        K.accum[sumPix] = acc;
        K.ior[sumPix] = iorPix;
    }
#ifdef RZ_PROF
    if (COUNT) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) if (c.p[k]) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) { atomicAdd(&pr[17], tTrace); atomicAdd(&pr[18], tAdv); atomicAdd(&pr[19], c.t[0]); atomicAdd(&pr[20], c.t[1]); atomicAdd(&pr[21], c.t[2]); for (int k = 3; k < 10; ++k) atomicAdd(&pr[19 + k], c.t[k]); }
        rz_prof_rounds(c, pr);
    }
#endif
    if (COUNT) {
        tally_flush(K, c);
        const unsigned long long pm = rz_ballot(sumInside);   // lanes 0..pixPerWave-1 (lane 0 alone when spp >= 64)
        if (lane == 0 && pm) atomicAdd(&K.counters->pixels, (unsigned long long)__popcll(pm));
    }
}



// Which pixel groups make up claim ci of a persistent launch.  A claim is `perClaim` groups: RUNS of 2^runShift consecutive
// groups, one run from each of perClaim >> runShift equal bands of the launch's groups (band j = runs j * nClaims ...).  With
// one run (runShift = log2 perClaim) a claim is perClaim consecutive groups -- a row of an 8x8 tile at 64 spp, the shape
// round 2 settled on there.  With several, every claim holds a sample of the whole frame, top to bottom, so claims cost
// about the same: a launch cannot end before its last claim does, and a claim that lies wholly inside a mesh takes many
// times the average (C4, 16 spp, 4 pixels per group: 4 / 8 / 16 consecutive groups per claim ran the frame in 10.8 / 12.9 /
// 18.2 ms against 9.4 without claims -- the tail grew with the claim).
struct ClaimMap {
    unsigned nGroups, perClaim, nClaims, runShift;
    __device__ __forceinline__ int group(unsigned ci, int g) const {
        const unsigned j = (unsigned)g >> runShift, w = (unsigned)g & ((1u << runShift) - 1u);
        return (int)(((j * nClaims + ci) << runShift) + w);
    }
    __device__ __forceinline__ int units_of(unsigned ci) const {       // the claim's groups that exist (a prefix: group() grows with g)
        int n = 0;
        for (unsigned g = 0; g < perClaim; ++g) n += (unsigned)group(ci, (int)g) < nGroups ? 1 : 0;
        return n;
    }
};

// ---------------------------------------------------------------------------------------------------------
// WAIT SLOTS (round 4): when a group's pixels are summed, and where its addends wait.
//
// Round 3 deferred per CLAIM: a claim with ONE parked path copied all its units' addends to a per-launch array of 1.5 KB per
// unit (3.2 GB at 1080p / 64 spp, 25 GB for C5) and all its pixels waited for the end of the wave's life.  Now a resident wave
// owns its claim scratch (the addends of the claim it is running, [unit][6][64] floats, as before: the units' stores and the
// sums stay in the cache) and K.nWaitSlots WAIT SLOTS of nBatches x 384 floats -- one GROUP's addends each -- and:
//   * at the end of a claim the wave counts the claim's parked paths per group (ballots over the pool entries it has just
//     written); a group without one is summed at once from the claim scratch (two thirds of C2's pixels);
//   * a group with parked paths gets a slot off the wave's free stack: its addends are copied there (1.5 KB per batch; round 3
//     copied the whole claim), its pool entries are told the slot, `outstanding[slot]` = its paths in the pool;
//   * a path that ends writes its sky term into the slot and decrements the count; the group is summed -- and its slot freed --
//     in the pool pass in which its last path comes back, not at the end of the launch;
//   * a claim may start only when as many slots are free as it has groups: otherwise the wave traces its pool first (every pass
//     ends paths; an empty pool means that every slot is free).
// Scratch per resident wave on C2: pool 80 KB + claim scratch 12 KB + 16 slots x 1.5 KB = 116 KB + the meta block, 470 MiB for
// the grid of 4 096 waves where round 3 took 3.7 GB; C5: 814 MiB where it took 25 GB (rz_debug_last_plan().scratch_mib).
// The meta block of a wave (ints): [0, NS) the launch-order index of the group in the slot | [NS, 2 NS) its outstanding paths
// (-1: the slot is free).  Paths that end decrement with atomics that return nothing; the wave LOOKS at the counts once per pool
// pass and once per claim that has groups to park -- lane s loads slot s's count past the L1 (the atomics act on the L2), a
// ballot gives the free / the ready slots, a popcount of the lower lanes their ranks, ds_permute the k-th of them to lane k.
// No free stack, no ready list, and above all no dependent memory round trip per shade pass or per unit: the first versions
// had an atomic WITH return per pass (is this the group's last path?) and a slot look-up per unit, and each cost the frame
// 5-10 % -- an exposed round trip drains every store the wave has in flight, 500 times per wave.
__device__ __forceinline__ int wmeta_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// lane k <- the number of the k-th set lane of `m` (a scalar loop over its bits: slots are few and this runs once per pass)
__device__ __forceinline__ int kth_set_lane(unsigned long long m) {
    const int lane = threadIdx.x & 63;
    int got = 0, k = 0;
    for (unsigned long long rest = m; rest != 0ull; rest &= rest - 1ull, ++k)
        if (lane == k) got = (int)__builtin_ctzll(rest);
    return got;
}

// One pass of ordered sums: lane 3 q + ch replays the additions of channel ch of the pass's q-th pixel (q < 21) in sample
// order, FS:717 then FS:709, sample after sample, from the group's addends A ([batch][6][64] floats); `slot` is the pixel's index
// in launch order (its tile and place in the tile), `lane0` its first lane in a batch (spp < 64: pixel-in-group x spp).
// GLASS: the currentIor the pixel ends with (FS:674) stands behind the addends, one row of 64, at its first lane.
#ifndef RZ_SUM_UNROLL
#define RZ_SUM_UNROLL 8
#endif
// The addends of a unit are written once and read once (1.5 KB per unit: 6.4 GB per C2 frame through an L2 they do not fit).
// Non-temporal stores and loads for them were measured in round 5 and are 3 % SLOWER (C2 10.65 -> 10.98 ms,
// C5 +1.4 %, c2g +2.9 %: the sums at the claim's end then wait for HBM instead of the L2): plain accesses (profiles/r05_regs/).
#define RZ_ST_ADD(p, v) (*(p) = (v))
#define RZ_LD_ADD(p) (*(p))
// DARK UNITS (round 5).  A unit none of whose samples has a first-hit light term -- every unit whose camera paths all end in the sky:
// 64 % of C2's -- would write three rows of +0 (FS:717's addends) and read them back.  It writes none: `lit` (bit b: the rows of the
// pixel's batch b hold light terms) sends the sums to a row of zeros that all waves share and that never leaves the caches
// (K.groupCounter + 128: 64 floats, zeroed when the context is made, never written).  The additions are the same -- `+ 0.0f` for
// `+ 0.0f` -- so the sums are bit for bit what they were; the claim scratch sees a third less traffic (12.4 -> 9.x GB per C2 frame).
__device__ __forceinline__ const float* zero_row(const KParams& K) { return reinterpret_cast<const float*>(K.groupCounter + 128); }
template <bool COUNT, bool GLASS>
__device__ __forceinline__ void ordered_sum_pass(const KParams& K, const bool valid, const float* __restrict__ A, const int slot, const int lane0, const unsigned lit = ~0u) {
    const int lane = threadIdx.x & 63;
    const int spp = K.spp;
    const int nBatches = (spp + 63) / 64;
    const int q = lane / 3, ch = lane - 3 * q;
    bool inside = false;
    size_t pix = 0;
    float chan = 0.0f, alpha = 0.0f;
    if (valid) {
        const int localTile = slot >> 6, l = slot & 63;
        const int tile = localTile * K.tileNRanks + K.tileRank;
        const int ty = tile / K.tilesX, tx = tile - ty * K.tilesX;
        const int px = tx * RZ_TILE_W + slot_x(l, spp < 64), py = ty * RZ_TILE_H + slot_y(l, spp < 64);
        if (slot < K.nSlots && px < K.width && py < K.height) {
            inside = true;
            pix = (size_t)py * K.width + px;
            if (K.sampleBase != 0) {
                const float4 a = K.accum[pix];
                chan = ch == 0 ? a.x : (ch == 1 ? a.y : a.z);
                alpha = a.w;
            }
        }
    }
    if (inside) {
        for (int b = 0; b < nBatches; ++b) {
            const float* Sf = A + (size_t)b * 384 + 64 * ch + lane0 + 192;
            const float* Lf = ((lit >> b) & 1u) ? Sf - 192 : zero_row(K) + lane0;
            const int cnt = spp >= 64 ? min(64, spp - b * 64) : spp;
            int k = 0;
            for (; k + RZ_SUM_UNROLL <= cnt; k += RZ_SUM_UNROLL) {
                float l[RZ_SUM_UNROLL], q8[RZ_SUM_UNROLL];
#pragma unroll
                for (int u = 0; u < RZ_SUM_UNROLL; ++u) { l[u] = RZ_LD_ADD(Lf + k + u); q8[u] = RZ_LD_ADD(Sf + k + u); }
#pragma unroll
                for (int u = 0; u < RZ_SUM_UNROLL; ++u) { chan = chan + l[u]; chan = chan + q8[u]; }   // FS:717, FS:709
            }
            for (; k < cnt; ++k) { chan = chan + RZ_LD_ADD(Lf + k); chan = chan + RZ_LD_ADD(Sf + k); }
        }
    }
    const int l0 = q < 21 ? 3 * q : 0;
    const float cx = __shfl(chan, l0), cy = __shfl(chan, l0 + 1), cz = __shfl(chan, l0 + 2);
    if (inside && ch == 0) {
        K.accum[pix] = make_float4(cx, cy, cz, alpha + (float)spp);
        K.ior[pix] = GLASS ? A[(size_t)nBatches * 384 + lane0] : 1.0f;
    }
    if (COUNT) {
        const unsigned long long im = rz_ballot(inside && ch == 0);
        if (lane == 0 && im) atomicAdd(&K.counters->pixels, (unsigned long long)__popcll(im));
    }
}

// The ordered sums of the pixels of the groups in the slots of `ready` (a lane mask over the slots), 21 pixels per pass; then
// the slots are free again.
template <bool COUNT, bool GLASS>
__device__ __forceinline__ void slot_sums(const KParams& K, const float* __restrict__ slotsBase, int* __restrict__ meta, const unsigned long long ready, int& freeCount) {
    const int lane = threadIdx.x & 63;
    const int spp = K.spp, NS = K.nWaitSlots;
    const int ppw = spp >= 64 ? 1 : 64 / spp;
    const int n = mask_count(ready);
    const int nPix = n * ppw;
    const int slotOfRank = kth_set_lane(ready);                 // lane r: the r-th ready slot
    const int gidxOfRank = lane < n ? wmeta_load(meta + slotOfRank) : 0;
    for (int p0 = 0; p0 < nPix; p0 += 21) {
        const int q = lane / 3;
        const int p = p0 + q;
        const bool valid = q < 21 && p < nPix;
        const int r = valid ? p / ppw : 0;
        const int pin = p - r * ppw;
        const int sl = __shfl(slotOfRank, r), gidx = __shfl(gidxOfRank, r);       // (by every lane)
        ordered_sum_pass<COUNT, GLASS>(K, valid, slotsBase + (size_t)sl * K.slotFloats, gidx * ppw + pin, pin * (spp >= 64 ? 0 : spp));
    }
    if ((ready >> lane) & 1ull) meta[NS + lane] = -1;
    freeCount += n;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// Transparent scenes on the compacting claims (round 4): FS:674's currentIor, resolved per UNIT.
//
// A pixel's samples are coupled through currentIor: a sample reads it where it scatters at a transparent surface (FS:727-742) and
// may leave another value behind.  The claim runs a unit like an opaque one, every lane from the value its pixel has reached
// (speculating that no earlier sample of the unit changes it).  If NO lane has read currentIor by the time the unit's paths are
// through or stand in front of their third segment, the unit was clean: late paths are parked like an opaque scene's (a parked
// path that later meets glass stops there and marks its group: the group is rendered again, below).  Otherwise the unit is
// resolved HERE, in the wave: every lane runs to its end, and then the chain is walked -- sample after sample in the shader's
// order, carrying currentIor: a sample that never read it has one result; one that did has one result per value it was run
// from (two kept: currentIor is 1.0 or a material's ior); the first sample whose pixel's chain arrives with a value it was not
// run from is run again, FROM ITS SNAPSHOT (rz_path.h: the state in front of its first transparent scatter), and with it,
// speculatively, every later sample that lacks that value -- until every chain walks through.  Exactly the additions the
// sequential shader makes, from exactly the currentIor it would have handed each sample.
// The chain is walked on the SCALAR unit over the lanes that read currentIor (v_readlane of their keys and results): the
// versions' bookkeeping is a few registers per lane, their addends wait in the wave's scratch (K.snap, behind the snapshots) --
// no LDS, so the whole BLAS stack stays there, unlike the group code's 5.4 KB of versions.  The rounds of a resolved unit go
// through the SAME trace / advance loop as the unit's first run (render_claim_compact): a second inlined copy of the walk
// made the kernel 150 KB and its instruction-cache misses eight times the group code's (profiles/r04_glass/).
// Which paths NOT to park in a transparent scene: a parked path that comes to a transparent scatter later costs its whole
// group a second rendering (pool_process).  A ray can only reach glass through the world box of an instance that holds
// transparent triangles (DevInstance.flags bit 1, a hint set at upload): the leaves of the pop-order TLAS list carry those
// boxes, the list is short (RayZen's scenes: a handful of objects), and a path whose next ray passes such a box stays in the
// wave for that segment -- if it does scatter at glass its unit is resolved in the wave, as it must be.  Long lists are not
// walked (every path is parked: the image is the same either way, this only steers the schedule).
#ifndef RZ_GLASS_BOX_MAX_DFS
#define RZ_GLASS_BOX_MAX_DFS 64
#endif
__device__ __forceinline__ bool may_hit_glass(const KParams& K, const bool cand, const v3 o, const v3 d) {
    if (K.glassBoxHint == 0 || K.nTlasDfs > RZ_GLASS_BOX_MAX_DFS || rz_ballot(cand) == 0ull) return false;
    const v3 inv = rcp3(d);
    bool hit = false;
    for (int pos = 0; pos < K.nTlasDfs; ++pos) {
        const f32x16 q = sload16(K.tlasDfs + pos);
        if (__float_as_int(q[7]) <= 0) continue;                                        // internal or never expanding
        if ((sload1(&K.instances[__float_as_int(q[9])].flags) & 2) == 0) continue;      // (a leaf of several instances speaks through its first: hint only)
        float tmin;
        hit = hit || slab(o, inv, q[0], q[1], q[2], q[4], q[5], q[6], tmin);
    }
    return cand && hit;
}

// ---------------------------------------------------------------------------------------------------------
// render_claim_compact: the persistent path with RAY COMPACTION across the pixels of a claim
// (north_star: "wavefront ballot / prefix-sum ray compaction for divergent bounces").
// A claim is up to UNITS (pixel, 64-sample batch) units (spp < 64: groups of 64 / spp pixels).  The wave runs the units one
// after the other, but a path that is about to trace its THIRD segment (bounce >= 2: 5 % of C2's paths, scattered over lanes
// whose neighbours have died) is parked instead: its 13 dwords of state + a back reference (unit, lane) go to the wave's
// pool, at position (paths already there) + (number of parked lanes below it) -- a ballot and a popcount of the lower lanes.  The
// pool is kept ACROSS the wave's claims and traced when it has filled up (pool_process).  The two addends of every sample go to
// the wave's claim scratch; at the end of the claim the groups without a parked path are summed, one lane per (pixel, colour
// channel): 24 lanes at once instead of 3 lanes eight times, and the others move to wait slots (above).  A path's arithmetic
// does not depend on where it runs: same bits.
// GLASS (the scene has a transparent material): units in which a sample reads currentIor are resolved in the wave
// (glass_resolve_unit); nRedo > 0: the "claim" is nRedo groups off the wave's redo list -- groups one of whose pooled paths met
// glass -- rendered again with every unit resolved in the wave.
// All scratch is private to the resident wave (L1 / L2 hits); the __syncthreads() of the one-wave workgroup order its stores
// before its loads.
// BACKSTOPS.  Three loops of the compacting launch carry bounds that "cannot be reached" -- and if an invariant were ever
// violated they would leave pixels unrendered or unsummed with RZ_OK returned.  A wave that reaches one sets a bit of
// the launch's error word (beside the claim counter); rz_sync reads the word and turns it into RZ_ERR_INTERNAL.
//   1: a claim found fewer free wait slots than it has groups after the pool had been drained   2: the pool did not drain in
//   (maxBounces + 1) passes   4: a transparent unit's chains did not resolve in 200 rounds
__device__ __forceinline__ void rz_backstop(const KParams& K, unsigned code) {
    if ((threadIdx.x & 63) == 0) atomicOr(K.groupCounter + RZ_ERRWORD, code);
}

template <bool COUNT, bool OVF, int UNITS, bool GLASS>
__device__ __forceinline__ void render_claim_compact(const KParams& K, const ClaimMap M, const unsigned ci, const int nRedo, const int redoTop, unsigned char* lds_raw, int& wpN, int& freeCount) {
    using namespace poolf;
    const int lane = threadIdx.x & 63;
    const BlasStackT<OVF> bstk{reinterpret_cast<uint2*>(lds_raw) + lane,
                               OVF ? K.blasOvf + ((size_t)blockIdx.x * K.blasOvfCap) * 64 + lane : nullptr, K.blasStackCap};
    float* const addBase = K.wslots + (size_t)blockIdx.x * K.wslotStride;         // the claim scratch: per group [batch][6][64] (+ 64: GLASS)
    unsigned* const W = K.wpool + (size_t)blockIdx.x * K.wpoolStride * RZ_GPOOL_FIELDS;
    int* const meta = K.wmeta + (size_t)blockIdx.x * 4 * K.nWaitSlots;
    const int NS = K.nWaitSlots;
    const size_t WS = K.wpoolStride;
    const size_t GF = K.slotFloats;
    const int spp = K.spp;
    const int nBatches = (spp + 63) / 64;
    const bool redo = GLASS && nRedo > 0;
    // a unit is one wave's worth of samples: a 64-sample batch of one pixel (spp >= 64), or all spp samples of each of the
    // ppw = 64 / spp pixels of a group (spp < 64: lane = pixel-in-group * spp + sample, the pixels a compact block of the tile)
    const int ppw = spp >= 64 ? 1 : 64 / spp;
    const int nGroups = redo ? nRedo : M.units_of(ci);   // <= UNITS / nBatches <= 16 by the launch plan; the caller has seen to it that as many wait slots are free
    const int nUnits = nGroups * nBatches;
    const int pixInUnit = spp >= 64 ? 0 : lane / spp; // (a lane with pixInUnit >= ppw is idle: spp need not divide 64)
    const int sampInUnit = spp >= 64 ? lane : lane - pixInUnit * spp;
    // the g-th group of this claim, as its index in launch order (redo: the list's top nRedo entries)
    auto group_of = [&](int g) -> int { return (GLASS && redo) ? wmeta_load(meta + 3 * NS + (redoTop - nRedo + g)) : M.group(ci, g); };
    int nPool = 0;                                            // wave-uniform: paths this claim has parked so far
    unsigned litMask = 0u;                                    // wave-uniform: bit u -- unit u of this claim wrote its light rows (see zero_row)
    int tileCached = -1, tileX = 0, tileY = 0;                // wave-uniform: the tile of the current unit
    float iorCarry = 1.0f;                                    // GLASS: the currentIor this lane's pixel has reached (carried from batch to batch)
    Tally c = {};
    for (int unit = 0; unit < nUnits; ++unit) {
        Path P;         // (per unit: nothing of a path lives across units)
#ifdef RZ_PROF
        const unsigned long long tph0_ = __builtin_amdgcn_s_memtime();
#endif
        P.mode = MODE_DONE;
        P.addLight = mk3(0.0f, 0.0f, 0.0f);
        P.addSky = mk3(0.0f, 0.0f, 0.0f);
        P.usedIor = 0;
        P.ior = 1.0f;
        const int g = nBatches == 1 ? unit : unit / nBatches, b = unit - g * nBatches;
        bool mine = false;
        float iorStart = 1.0f;
        Tally cu = {};                                  // counting launches of transparent scenes: this unit's own tallies (a snapshot keeps
        Tally& tu = (COUNT && GLASS) ? cu : c;          // its sample's prefix, a resolved unit counts its chosen versions alone)
        {
            const int slot = group_of(g) * ppw + pixInUnit;
            const int localTile = slot >> 6, l = slot & 63;
            int px, py;
            if (spp >= 64 || (64 % ppw) == 0) {
                // (wave-uniform: the group's pixels lie in one tile; a claim lies within one tile unless its size does not divide 64)
                const int ut = __builtin_amdgcn_readfirstlane(localTile);
                if (ut != tileCached) {
                    tileCached = ut;
                    const int tile = ut * K.tileNRanks + K.tileRank;
                    tileY = tile / K.tilesX;        // the integer division costs ~40 instructions: once per claim, not per unit
                    tileX = tile - tileY * K.tilesX;
                }
                px = tileX * RZ_TILE_W + slot_x(l, spp < 64); py = tileY * RZ_TILE_H + slot_y(l, spp < 64);
            } else {                                // 64 / spp pixels per group does not divide the tile's 64: a group can straddle two tiles
                const int tile = localTile * K.tileNRanks + K.tileRank;
                const int ty = tile / K.tilesX, tx = tile - ty * K.tilesX;
                px = tx * RZ_TILE_W + slot_x(l, true); py = ty * RZ_TILE_H + slot_y(l, true);
            }
            const int s = b * 64 + sampInUnit;
            const bool pixInside = pixInUnit < ppw && slot < K.nSlots && px < K.width && py < K.height;
            if constexpr (GLASS) {
                // FS:674: currentIor enters the pixel's first sample of this launch as the last launch left it (sample_base > 0) or as 1
                iorStart = b == 0 ? ((K.sampleBase != 0 && pixInside) ? K.ior[(size_t)py * K.width + px] : 1.0f) : iorCarry;
            }
            if (pixInside && s < spp) {
                mine = true;
                const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
                P.uv.x = fragx / (float)K.width;
                P.uv.y = fragy / (float)K.height;
                P.fragSum = fragx + fragy;
                P.color = mk3(0.0f, 0.0f, 0.0f);
                P.samp = K.sampleBase + s;
                begin_sample<COUNT>(K, P, tu);
                if constexpr (GLASS) P.ior = iorStart;
            }
        }
        // A unit's paths run until they finish or stand in front of their third segment (bounce >= 2).
        // (transparent scenes: ... unless that segment's ray passes the box of an instance with glass in it -- may_hit_glass --
        //  and, once a sample of the unit has read currentIor, to their ends: the unit is then resolved in the wave, round after
        //  round through this same loop -- see "Transparent scenes on the compacting claims" above)
        bool stopLate = !GLASS || !(COUNT || redo);         // late paths stop to be parked (counting launches and redo claims resolve every unit)
        // the versions of this lane's sample (GLASS): the currentIor each was run from / left behind (bit patterns: a NaN must
        // equal itself), bit 0 valid, bit 1 the sample read currentIor; their addends wait in the wave's scratch
        unsigned key0 = __float_as_uint(iorStart), key1 = 0u, out0 = 0u, out1 = 0u, want = __float_as_uint(iorStart);
        int info0 = 0, info1 = 0, nextSlot = 0, chosen = 0;
        bool running = mine;                                // lanes whose version is being computed in this round
        bool resolved = false;                              // wave-uniform: the unit went through the rounds
        float iorEnd = iorStart;
        Tally tv0 = {}, tv1 = {};                           // (counting launches: the tallies of each version's whole run)
        float* const V = GLASS ? K.snap + (size_t)blockIdx.x * K.snapStride + (size_t)(RZ_SNAP_FIELDS + RZ_SNAP_TALLY) * 64 + lane : nullptr;   // [version][6] rows, this lane's column
#ifdef RZ_PROF
        c.rnd = 0;
#endif
        // (every round gives each pixel's first blocked sample the version it lacks: at most 64 rounds per unit; the bound is a backstop)
        for (int round = 0; round < 200; ++round) {
            if (GLASS && round == 199) rz_backstop(K, 4u);
            // (one exit, at the end of the body: see blas_walk)
            bool run, anyRun;
            {
                const bool late = stopLate && P.mode == MODE_SEGMENT && P.bounce >= RZ_PARK_BOUNCE;
                run = P.mode != MODE_DONE && !late;
                if constexpr (GLASS) { if (stopLate) run = run || may_hit_glass(K, late, P.o, P.d); }
                anyRun = rz_ballot(run) != 0ull;
            }
            while (anyRun) {
                if (run) {
                    HitRec h;
#ifdef RZ_PROF
                    RZ_SITE(c, 6);
                    const unsigned long long tq0_ = __builtin_amdgcn_s_memtime();
#endif
                    const bool found = trace_closest<COUNT, OVF>(K, P.o, P.d, h, bstk, tu);
#ifdef RZ_PROF
                    c.rt[c.rnd & 7] += __builtin_amdgcn_s_memtime() - tq0_;
#endif
                    advance<COUNT, GLASS, GLASS ? 1 : 0>(K, P, found, h, tu);
                }
#ifdef RZ_PROF
                if (c.rnd < 7) ++c.rnd;
#endif
                const bool late = stopLate && P.mode == MODE_SEGMENT && P.bounce >= RZ_PARK_BOUNCE;
                if constexpr (GLASS) {
                    // (a lane that did not run in this trip keeps its ray and its verdict: the boxes are looked at once per ray)
                    const bool ran = run;
                    run = P.mode != MODE_DONE && !late;
                    if (stopLate) run = ran && (run || may_hit_glass(K, ran && late, P.o, P.d));
                } else {
                    run = P.mode != MODE_DONE && !late;
                }
                anyRun = rz_ballot(run) != 0ull;
            }
            if constexpr (!GLASS) {
                break;
            } else {
                if (stopLate) {
                    // did a sample of this unit read currentIor?  If not the unit was clean: its late paths are parked
                    if (rz_ballot(P.usedIor != 0) == 0ull) break;
                    stopLate = false;
                    if (rz_ballot(P.mode != MODE_DONE) != 0ull) continue;       // the late paths run to their ends in the wave first
                }
                resolved = true;
                // ---- every lane of this round has its version
                if (running) {
                    const unsigned o = __float_as_uint(P.ior);
                    const int inf = 1 | (P.usedIor ? 2 : 0);
                    float* const R = V + (size_t)nextSlot * 6 * 64;
                    R[0] = P.addLight.x; R[64] = P.addLight.y; R[128] = P.addLight.z;
                    R[192] = P.addSky.x; R[256] = P.addSky.y; R[320] = P.addSky.z;
                    if (nextSlot == 0) { key0 = want; out0 = o; info0 = inf; if (COUNT) tv0 = cu; }
                    else { key1 = want; out1 = o; info1 = inf; if (COUNT) tv1 = cu; }
                    nextSlot ^= 1;
                }
                // ---- walk the chains: the lanes that read currentIor, in sample order, on the scalar unit
                const unsigned long long touch = rz_ballot(mine && (((info0 | info1) & 2) != 0));
                unsigned long long need = 0ull;
                int curPix = -1;
                unsigned cur = 0u;
                bool blocked = false;
                iorEnd = iorStart;
                chosen = 0;
                for (unsigned long long rest = touch; rest != 0ull; rest &= rest - 1ull) {
                    const int k = (int)__builtin_ctzll(rest);
                    const int pix = __builtin_amdgcn_readlane(pixInUnit, k);
                    if (pix != curPix) { curPix = pix; cur = (unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(iorStart), k); blocked = false; }
                    const int i0 = __builtin_amdgcn_readlane(info0, k), i1 = __builtin_amdgcn_readlane(info1, k);
                    const unsigned k0 = (unsigned)__builtin_amdgcn_readlane((int)key0, k), k1 = (unsigned)__builtin_amdgcn_readlane((int)key1, k);
                    const bool has0 = (i0 & 1) && k0 == cur, has1 = (i1 & 1) && k1 == cur;
                    if (!has0 && !has1) {
                        // this sample was not run from the value its chain arrives with: it is run again; the samples behind it are
                        // walked on the guess that it leaves the value alone
                        need |= 1ull << k;
                        if (lane == k) want = cur;
                        blocked = true;
                        continue;
                    }
                    const int v = has0 ? 0 : 1;
                    if (!blocked && lane == k) chosen = v;
                    cur = (unsigned)(v ? __builtin_amdgcn_readlane((int)out1, k) : __builtin_amdgcn_readlane((int)out0, k));
                    if (!blocked && pixInUnit == pix) iorEnd = __uint_as_float(cur);
                }
                if (need == 0ull) break;
                // ---- the lanes that lack a version start it from their snapshots (rz_path.h: in front of their first transparent scatter)
                RZ_GSTAT(3, 1);
                running = ((need >> lane) & 1ull) != 0ull;
                if (running) {
                    P.color = mk3(0.0f, 0.0f, 0.0f);
                    if (COUNT) cu = {};
                    snapshot_load<COUNT>(K, P, tu);
                    P.ior = __uint_as_float(want);
                    P.samp = K.sampleBase + b * 64 + sampInUnit;
                    scatter<COUNT, true, 0>(K, P, tu);
                }
            }
        }
        if constexpr (GLASS) {
            RZ_GSTAT(0, 1);
            if (resolved) {
                RZ_GSTAT(1, 1);
                // every lane takes the version its chain chose (a sample that never read currentIor has the one)
                if (mine) {
                    const float* const R = V + (size_t)chosen * 6 * 64;
                    P.addLight = mk3(R[0], R[64], R[128]);
                    P.addSky = mk3(R[192], R[256], R[320]);
                }
                if (COUNT) { cu = {}; if (mine) tally_add(cu, chosen ? tv1 : tv0); }      // the unit's tallies are those of the versions its chains chose
                P.mode = MODE_DONE;
            }
            if (COUNT) tally_add(c, cu);
            iorCarry = iorEnd;
            if (b == nBatches - 1) addBase[(size_t)g * GF + (size_t)nBatches * 384 + lane] = iorEnd;       // (read at this lane's pixel's first lane by the sums)
        }
        const bool parked = P.mode != MODE_DONE;
        const unsigned long long pm = rz_ballot(parked);
        {
            float* const A = addBase + (size_t)g * GF + (size_t)b * 384;
            // (a dark unit -- every lane's light term is +0, bit for bit -- keeps its three light rows unwritten: see zero_row)
            const bool litUnit = !RZ_DARK_UNITS || rz_ballot((__float_as_uint(P.addLight.x) | __float_as_uint(P.addLight.y) | __float_as_uint(P.addLight.z)) != 0u) != 0ull;
            if (litUnit) {
                litMask |= 1u << unit;
                RZ_ST_ADD(A + lane, P.addLight.x); RZ_ST_ADD(A + 64 + lane, P.addLight.y); RZ_ST_ADD(A + 128 + lane, P.addLight.z);    // FS:717
            }
            RZ_ST_ADD(A + 192 + lane, P.addSky.x); RZ_ST_ADD(A + 256 + lane, P.addSky.y); RZ_ST_ADD(A + 320 + lane, P.addSky.z);       // FS:709 (parked: still 0)
        }
        // compaction: parked lane -> pool position (paths already there) + (number of parked lanes below it)
        if (parked) {
            const size_t sl = (size_t)(wpN + nPool) + __popcll(pm & ((1ull << lane) - 1ull));
            W[OX * WS + sl] = __float_as_uint(P.o.x); W[OY * WS + sl] = __float_as_uint(P.o.y); W[OZ * WS + sl] = __float_as_uint(P.o.z);
            W[DX * WS + sl] = __float_as_uint(P.d.x); W[DY * WS + sl] = __float_as_uint(P.d.y); W[DZ * WS + sl] = __float_as_uint(P.d.z);
            W[TPX * WS + sl] = __float_as_uint(P.throughput.x); W[TPY * WS + sl] = __float_as_uint(P.throughput.y);
            W[TPZ * WS + sl] = __float_as_uint(P.throughput.z);
            W[SEEDX * WS + sl] = __float_as_uint(P.seed.x); W[SEEDY * WS + sl] = __float_as_uint(P.seed.y);
            W[SAMP * WS + sl] = (unsigned)P.samp;
            W[BACK * WS + sl] = ((unsigned)P.bounce << 16) | ((unsigned)unit << 6) | (unsigned)lane;     // (the unit becomes the batch within its group at the end of the claim)
        }
        nPool += mask_count(pm);
        if (GLASS) RZ_GSTAT(4, mask_count(pm));
#ifdef RZ_PROF
        c.t[4] += __builtin_amdgcn_s_memtime() - tph0_;
#endif
    }
    __syncthreads();        // the claim's addends and pool entries are read next
#ifdef RZ_PROF
    const unsigned long long tce0_ = __builtin_amdgcn_s_memtime();
#endif
    // ---- the end of the claim: which groups wait for parked paths?  Lane g counts group g's entries among those the claim has
    // just written (a ballot per group and 64 entries), the groups that wait move to wait slots, the others are summed.
    float* const slotsBase = addBase + K.claimScratchFloats;
    int myCnt = 0;
    for (int e0 = 0; e0 < nPool; e0 += 64) {
        const int e = e0 + lane;
        const int ge = e < nPool ? (int)((W[BACK * WS + (size_t)(wpN + e)] >> 6) & 1023u) / nBatches : -1;
        for (int g = 0; g < nGroups; ++g) {
            const int n = mask_count(rz_ballot(ge == g));
            if (lane == g) myCnt += n;
        }
    }
    const bool waiting = lane < nGroups && myCnt > 0;
    const unsigned long long wm = rz_ballot(waiting);
    int mySlot = 0;
    if (wm != 0ull) {
        // the k-th waiting group takes the k-th free slot (count -1); the caller has seen to it that there are enough
        const unsigned long long freeMask = rz_ballot(lane < NS && wmeta_load(meta + NS + (lane < NS ? lane : 0)) == -1);
        const int freeOfRank = kth_set_lane(freeMask);
        mySlot = __shfl(freeOfRank, __popcll(wm & ((1ull << lane) - 1ull)));     // (by every lane)
        if (waiting) {
            meta[mySlot] = group_of(lane);
            meta[NS + mySlot] = myCnt;
            if (GLASS) meta[2 * NS + mySlot] = 0;
        }
        freeCount -= mask_count(wm);
    }
    // ... their addends move to their slots: a group's rows of 64 floats, one row per step
    for (unsigned long long rest = wm; rest != 0ull; rest &= rest - 1ull) {
        const int g = (int)__builtin_ctzll(rest);
        const int sl = __builtin_amdgcn_readlane(mySlot, g);
        const float* __restrict__ src = addBase + (size_t)g * GF;
        float* __restrict__ dst = slotsBase + (size_t)sl * GF;
        for (int r = 0; r < (int)(GF >> 6); ++r) {
            // (a dark unit's light rows were never written: the slot gets the zeros they stand for)
            const int rb = r / 6, rj = r - 6 * rb;
            const bool dark = rb < nBatches && rj < 3 && ((litMask >> (g * nBatches + rb)) & 1u) == 0u;
            const float* __restrict__ from = dark ? zero_row(K) : src + r * 64;
            RZ_ST_ADD(dst + r * 64 + lane, RZ_LD_ADD(from + lane));
        }
    }
    // ... and their pool entries learn the slot and their batch within the group
    for (int e0 = 0; e0 < nPool; e0 += 64) {
        const int e = e0 + lane;
        const unsigned back = e < nPool ? W[BACK * WS + (size_t)(wpN + e)] : 0u;
        const int unit = (int)((back >> 6) & 1023u), ge = unit / nBatches;
        const int eSlot = __shfl(mySlot, ge);       // (by every lane: a lane that sits a shuffle out reads as zero to the others)
        if (e < nPool) {
            W[BACK * WS + (size_t)(wpN + e)] = (back & 0xffff003fu) | ((unsigned)(unit - ge * nBatches) << 6);
            W[(size_t)(RZ_GPOOL_FIELDS - 1) * WS + (size_t)(wpN + e)] = (unsigned)eSlot;
        }
    }
    wpN += nPool;
    // the groups that do not wait: their ordered sums, straight from the claim scratch
#ifdef RZ_PROF
    const unsigned long long tce1_ = __builtin_amdgcn_s_memtime();
#endif
    const int nPix = nGroups * ppw;
    for (int p0 = 0; p0 < nPix; p0 += 21) {
        const int q = lane / 3;
        const int p = p0 + q;
        const int g = (q < 21 && p < nPix) ? p / ppw : 0;
        const int gCnt = __shfl(myCnt, g);          // (by every lane)
        const bool valid = q < 21 && p < nPix && gCnt == 0;
        const int pin = p - g * ppw;
        ordered_sum_pass<COUNT, GLASS>(K, valid, addBase + (size_t)g * GF, group_of(g) * ppw + pin, pin * (spp >= 64 ? 0 : spp), litMask >> (g * nBatches));
    }
#ifdef RZ_PROF
    c.t[16] += __builtin_amdgcn_s_memtime() - tce0_; c.t[18] += __builtin_amdgcn_s_memtime() - tce1_;
#endif
    if (COUNT) {
        tally_flush(K, c);
#ifdef RZ_PROF
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) if (c.p[k]) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) { for (int k = 0; k < 12; ++k) atomicAdd(&pr[19 + k], c.t[k]); for (int k = 12; k < 20; ++k) atomicAdd(&pr[108 + k], c.t[k]); }
        rz_prof_rounds(c, pr);
#endif
    }
    __syncthreads();        // the next claim overwrites the scratch; the pool pass reads what this one has written
}

// ---------------------------------------------------------------------------------------------------------
// The pool a resident wave keeps ACROSS its claims.
// A claim parks a few dozen paths; lanes can only refill from a list that is much longer than the wave is wide
// (rz_trace.h: pool_trace).  So a wave does not work a claim's parked paths off before its next claim: they collect in the
// wave's own pool -- third, fourth ... segments side by side, a path carries its bounce -- and when the pool holds
// K.wpoolChunk paths (and when the claims have run out, and when a claim finds too few free wait slots) the wave traces ALL of
// them together, shades them 64 at a time (the late part of the shader's bounce loop, FS:705-711 / 720-769: sky and the end, or
// scatter and Russian roulette), writes the sky term of the paths that end to their samples' places in their groups' wait
// slots and keeps the survivors, compacted in place (ballot + prefix popcount), for the next time.  A group whose last path
// has come back is summed at the end of the pass (slot_sums).  Late work and coherent work so run side by side on a CU all
// through the launch -- the late rays keep the texture-address unit busy (64 B per lane and step whatever the ray), the
// coherent ones the issue slots -- and no queue is shared between waves: nothing to synchronise, nothing to wait for.
// A path's arithmetic does not depend on the lane, wave or moment that runs it: same bits as every other launch shape.
// GLASS: a pooled path belongs to a unit in which no sample had read currentIor; if IT comes to a transparent scatter it stops
// there (it may not read a value that an earlier sample of its pixel, still in the pool, could change) and marks its group,
// which is put on the wave's redo list instead of being summed when its last path is back.
template <bool COUNT, bool OVF, bool GLASS>
__device__ __forceinline__ int pool_process(const KParams& K, unsigned* __restrict__ W, const size_t WS, const int n, const BlasStackT<OVF>& bstk, int& freeCount, int& redoCount, int& lateN) {
    using namespace poolf;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    float* const slotsBase = K.wslots + (size_t)blockIdx.x * K.wslotStride + K.claimScratchFloats;      // (behind the wave's claim scratch)
    int* const meta = K.wmeta + (size_t)blockIdx.x * 4 * K.nWaitSlots;
    const int NS = K.nWaitSlots;
    const int spp = K.spp, nBatches = (spp + 63) / 64;
    // transparent scenes: the wave's LATE LIST -- pooled samples that stand in front of a transparent scatter (see below)
    unsigned* const LL = GLASS ? reinterpret_cast<unsigned*>(K.snap + (size_t)blockIdx.x * K.snapStride + (size_t)(RZ_SNAP_FIELDS + RZ_SNAP_TALLY + RZ_GVER_ROWS) * 64) : nullptr;
    constexpr int LC = RZ_GLATE_CAP;
    constexpr unsigned FULL_REDO = 1u << 30;
    Tally c = {};
#ifdef RZ_PROF
    const unsigned long long tl0_ = __builtin_amdgcn_s_memtime();
#endif
    pool_trace<COUNT, OVF>(K, W, WS, n, bstk, c);
#ifdef RZ_PROF
    const unsigned long long tl1_ = __builtin_amdgcn_s_memtime();
    c.t[12] += tl1_ - tl0_;
    c.t[10] += 1;                                   // pools traced
    c.t[11] += (unsigned long long)n;               // queries in them
#endif
    int write = 0;                                  // survivors so far: the write cursor trails the read cursor
    for (int sb = 0; sb < n; sb += 64) {
        const int sl = sb + lane;
        Path P;
        P.mode = MODE_DONE;
        P.addLight = mk3(0.0f, 0.0f, 0.0f);
        P.addSky = mk3(0.0f, 0.0f, 0.0f);
        P.usedIor = 0;
        P.ior = 1.0f;
        P.gflag = 0;
        unsigned back = 0, wslot = 0;
        if (sl < n) {
            P.o = mk3(__uint_as_float(W[OX * WS + sl]), __uint_as_float(W[OY * WS + sl]), __uint_as_float(W[OZ * WS + sl]));
            P.d = mk3(__uint_as_float(W[DX * WS + sl]), __uint_as_float(W[DY * WS + sl]), __uint_as_float(W[DZ * WS + sl]));
            P.throughput = mk3(__uint_as_float(W[TPX * WS + sl]), __uint_as_float(W[TPY * WS + sl]), __uint_as_float(W[TPZ * WS + sl]));
            P.seed.x = __uint_as_float(W[SEEDX * WS + sl]);
            P.seed.y = __uint_as_float(W[SEEDY * WS + sl]);
            P.samp = (int)W[SAMP * WS + sl];
            back = W[BACK * WS + sl];
            wslot = W[(size_t)(RZ_GPOOL_FIELDS - 1) * WS + sl];
            P.bounce = (int)((back >> 16) & 0x7fffu);
            if constexpr (GLASS) {      // (bit 31: a released late sample -- its currentIor is the one the sequential shader would hand it)
                P.gflag = (int)(back >> 31);
                if (P.gflag) P.ior = __uint_as_float(W[IOR * WS + sl]);
            }
            P.color = mk3(0.0f, 0.0f, 0.0f);
            P.mode = MODE_SEGMENT;
            const int qTri = (int)W[QTRI * WS + sl];
            const bool found = qTri >= 0;
            HitRec h;
            if (found) {     // the winner's point, normal and material, as at the end of trace_closest (FS:411-412, 489-491)
                const int qInst = (int)W[QINST * WS + sl];
                const float4 nm = *reinterpret_cast<const float4*>(K.triN + qTri);
                h.t = __uint_as_float(W[QT * WS + sl]);
                h.p = mk3(__uint_as_float(W[QPX * WS + sl]), __uint_as_float(W[QPY * WS + sl]), __uint_as_float(W[QPZ * WS + sl]));
                h.n = normalize(x34_normal(K.instances[qInst].inv, mk3(nm.x, nm.y, nm.z)));
                h.mat = __float_as_int(nm.w);
                h.inst = qInst;
            }
            advance<COUNT, GLASS, GLASS ? 2 : 0>(K, P, found, h, c);      // one segment: sky and the end, or scatter (no shadow queries after bounce 0)
        }
        const bool parked = P.mode != MODE_DONE;
        // GLASS: the path stands in front of a transparent scatter and may not read currentIor yet (an earlier sample of its pixel
        // can still be in the pool).  It waits on the wave's late list -- the state the scatter needs, 19 dwords -- still counted
        // among its group's outstanding paths, and is RELEASED, earliest sample first, once no other path of the group is in
        // flight (below).  Groups of several batches per pixel (units resolved in the wave may stand behind it in the chain) and a
        // full list fall back to rendering the group again.
        bool met = GLASS && sl < n && !P.gflag && P.usedIor != 0;
        bool toList = false;
        if constexpr (GLASS) {
            const unsigned long long mm = rz_ballot(met);
            const bool room = nBatches == 1 && lateN + mask_count(mm) <= LC;
            toList = met && room;
            if (toList) {
                const size_t e = (size_t)lateN + __popcll(mm & below);
                LL[0 * LC + e] = __float_as_uint(P.hp.x); LL[1 * LC + e] = __float_as_uint(P.hp.y); LL[2 * LC + e] = __float_as_uint(P.hp.z);
                LL[3 * LC + e] = __float_as_uint(P.hn.x); LL[4 * LC + e] = __float_as_uint(P.hn.y); LL[5 * LC + e] = __float_as_uint(P.hn.z);
                LL[6 * LC + e] = __float_as_uint(P.pdir.x); LL[7 * LC + e] = __float_as_uint(P.pdir.y); LL[8 * LC + e] = __float_as_uint(P.pdir.z);
                LL[9 * LC + e] = __float_as_uint(P.throughput.x); LL[10 * LC + e] = __float_as_uint(P.throughput.y); LL[11 * LC + e] = __float_as_uint(P.throughput.z);
                LL[12 * LC + e] = __float_as_uint(P.seed.x); LL[13 * LC + e] = __float_as_uint(P.seed.y);
                LL[14 * LC + e] = (unsigned)P.samp; LL[15 * LC + e] = (unsigned)P.bounce; LL[16 * LC + e] = (unsigned)P.hmat;
                LL[17 * LC + e] = back & 0xffffu; LL[18 * LC + e] = wslot;
                atomicAdd(meta + 2 * NS + wslot, 1);       // the group's samples on the late list
            }
            if (room) lateN += mask_count(mm);
        }
        if (sl < n && !parked && !toList) {           // the path has ended: its sky term (FS:709; zero when it ended by roulette or budget) goes to its sample's place
            float* const A = slotsBase + (size_t)wslot * K.slotFloats + (size_t)((back >> 6) & 1023u) * 384;
            const unsigned bl = back & 63u;
            RZ_ST_ADD(A + 192 + bl, P.addSky.x); RZ_ST_ADD(A + 256 + bl, P.addSky.y); RZ_ST_ADD(A + 320 + bl, P.addSky.z);
            if constexpr (GLASS) {
                if (met) atomicOr(reinterpret_cast<unsigned*>(meta) + 2 * NS + wslot, FULL_REDO);       // no room on the list: the group is rendered again
                if (P.gflag) slotsBase[(size_t)wslot * K.slotFloats + (size_t)nBatches * 384 + (spp >= 64 ? 0u : (bl / (unsigned)spp) * (unsigned)spp)] = P.ior;   // what this sample leaves its pixel (FS:742)
            }
            atomicAdd(meta + NS + wslot, -1);      // (nothing comes back: the wave looks at the counts once, after the shade rounds)
        }
        const unsigned long long pm = rz_ballot(parked);
        if (parked) {
            const size_t d = (size_t)write + __popcll(pm & below);        // d <= sl: this round's slots have all been read
            W[OX * WS + d] = __float_as_uint(P.o.x); W[OY * WS + d] = __float_as_uint(P.o.y); W[OZ * WS + d] = __float_as_uint(P.o.z);
            W[DX * WS + d] = __float_as_uint(P.d.x); W[DY * WS + d] = __float_as_uint(P.d.y); W[DZ * WS + d] = __float_as_uint(P.d.z);
            W[TPX * WS + d] = __float_as_uint(P.throughput.x); W[TPY * WS + d] = __float_as_uint(P.throughput.y); W[TPZ * WS + d] = __float_as_uint(P.throughput.z);
            W[SEEDX * WS + d] = __float_as_uint(P.seed.x); W[SEEDY * WS + d] = __float_as_uint(P.seed.y);
            W[SAMP * WS + d] = (unsigned)P.samp;
            W[BACK * WS + d] = ((unsigned)P.gflag << 31) | ((unsigned)P.bounce << 16) | (back & 0xffffu);
            W[(size_t)(RZ_GPOOL_FIELDS - 1) * WS + d] = wslot;
            if constexpr (GLASS) W[IOR * WS + d] = __float_as_uint(P.ior);
        }
        write += mask_count(pm);
    }
#ifdef RZ_PROF
    c.t[9] += __builtin_amdgcn_s_memtime() - tl1_;      // the shade rounds
#endif
    __syncthreads();
#ifdef RZ_PROF
    const unsigned long long tss0_ = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (GLASS) {
        // ---- release: a group none of whose paths is in flight (outstanding == its samples on the late list) lets its EARLIEST
        // late sample go: the sample takes the currentIor its pixel has reached (the slot's ior row), scatters (FS:723-746) and
        // goes on through the pool as a path that may read currentIor -- or ends on the spot.  One sample per group and step: the
        // next one sees what this one leaves.  Repeated until no group can release (a sample that ends on the spot frees the next).
        for (int step = 0; step < 4 * LC && lateN > 0; ++step) {
            const int cnt = lane < NS ? wmeta_load(meta + NS + lane) : 0;
            const unsigned lt = lane < NS ? (unsigned)wmeta_load(meta + 2 * NS + lane) : 0u;
            const unsigned long long rel = rz_ballot(lane < NS && (lt & 0xffffu) != 0u && cnt == (int)(lt & 0xffffu));
            if (rel == 0ull) break;
            for (unsigned long long rest = rel; rest != 0ull; rest &= rest - 1ull) {
                const int s = (int)__builtin_ctzll(rest);
                // the group's earliest sample on the list (key = batch << 6 | lane)
                int best = -1;
                unsigned bestKey = 0xffffffffu;
                for (int e0 = 0; e0 < lateN; e0 += 64) {
                    const int e = e0 + lane;
                    const bool mine_ = e < lateN && LL[18 * LC + e] == (unsigned)s;
                    const unsigned key_ = mine_ ? LL[17 * LC + e] : 0xffffffffu;
                    for (unsigned long long mm = rz_ballot(mine_); mm != 0ull; mm &= mm - 1ull) {
                        const int l_ = (int)__builtin_ctzll(mm);
                        const unsigned k_ = (unsigned)__builtin_amdgcn_readlane((int)key_, l_);
                        if (k_ < bestKey) { bestKey = k_; best = e0 + l_; }
                    }
                }
                if (best < 0) continue;     // (cannot be: the count says the group has samples on the list)
                Path P;
                P.mode = MODE_DONE;
                P.addLight = mk3(0.0f, 0.0f, 0.0f);
                P.addSky = mk3(0.0f, 0.0f, 0.0f);
                P.usedIor = 0; P.gflag = 1; P.ior = 1.0f;
                float* const A = slotsBase + (size_t)s * K.slotFloats;
                const unsigned bl = bestKey & 63u;
                const unsigned iorIdx = (unsigned)nBatches * 384u + (spp >= 64 ? 0u : (bl / (unsigned)spp) * (unsigned)spp);
                if (lane == 0) {
                    const size_t e = (size_t)best;
                    P.hp = mk3(__uint_as_float(LL[0 * LC + e]), __uint_as_float(LL[1 * LC + e]), __uint_as_float(LL[2 * LC + e]));
                    P.hn = mk3(__uint_as_float(LL[3 * LC + e]), __uint_as_float(LL[4 * LC + e]), __uint_as_float(LL[5 * LC + e]));
                    P.pdir = mk3(__uint_as_float(LL[6 * LC + e]), __uint_as_float(LL[7 * LC + e]), __uint_as_float(LL[8 * LC + e]));
                    P.throughput = mk3(__uint_as_float(LL[9 * LC + e]), __uint_as_float(LL[10 * LC + e]), __uint_as_float(LL[11 * LC + e]));
                    P.seed.x = __uint_as_float(LL[12 * LC + e]); P.seed.y = __uint_as_float(LL[13 * LC + e]);
                    P.samp = (int)LL[14 * LC + e]; P.bounce = (int)LL[15 * LC + e]; P.hmat = (int)LL[16 * LC + e];
                    P.color = mk3(0.0f, 0.0f, 0.0f);
                    P.ior = A[iorIdx];
                    scatter<COUNT, true, 0>(K, P, c);
                    LL[18 * LC + e] = 0xffffffffu;         // (off the list)
                    atomicAdd(meta + 2 * NS + s, -1);
                    if (P.mode != MODE_DONE) {             // it goes on: a pooled path that may read currentIor
                        const size_t d = (size_t)write;
                        W[OX * WS + d] = __float_as_uint(P.o.x); W[OY * WS + d] = __float_as_uint(P.o.y); W[OZ * WS + d] = __float_as_uint(P.o.z);
                        W[DX * WS + d] = __float_as_uint(P.d.x); W[DY * WS + d] = __float_as_uint(P.d.y); W[DZ * WS + d] = __float_as_uint(P.d.z);
                        W[TPX * WS + d] = __float_as_uint(P.throughput.x); W[TPY * WS + d] = __float_as_uint(P.throughput.y); W[TPZ * WS + d] = __float_as_uint(P.throughput.z);
                        W[SEEDX * WS + d] = __float_as_uint(P.seed.x); W[SEEDY * WS + d] = __float_as_uint(P.seed.y);
                        W[SAMP * WS + d] = (unsigned)P.samp;
                        W[BACK * WS + d] = (1u << 31) | ((unsigned)P.bounce << 16) | bestKey;
                        W[(size_t)(RZ_GPOOL_FIELDS - 1) * WS + d] = (unsigned)s;
                        W[IOR * WS + d] = __float_as_uint(P.ior);
                    } else {                               // it ended at the scatter (bounce budget, roulette): nothing more to add
                        A[iorIdx] = P.ior;
                        atomicAdd(meta + NS + s, -1);
                    }
                }
                write += __builtin_amdgcn_readfirstlane(P.mode != MODE_DONE ? 1 : 0);
            }
            __syncthreads();
        }
        // (the list is emptied when nothing on it is alive: records are few and short-lived)
        {
            bool any = false;
            for (int e0 = 0; e0 < lateN; e0 += 64) any = any || rz_ballot(e0 + lane < lateN && LL[18 * LC + e0 + lane] != 0xffffffffu) != 0ull;
            if (!any) lateN = 0;
        }
    }
    // the groups whose last path has come back in this pass: slots in use whose count is 0 (read past the L1: the atomics act on the L2)
    unsigned long long ready = rz_ballot(lane < NS && wmeta_load(meta + NS + (lane < NS ? lane : 0)) == 0);
    if constexpr (GLASS) {
        // ... of them, the groups that must be rendered again (a pooled path met glass and could not wait on the list): onto the
        // redo list (the wave renders them before anything else: rz_render_samples), their slots are free
        const unsigned long long dirty = ready & rz_ballot(lane < NS && ((unsigned)wmeta_load(meta + 2 * NS + (lane < NS ? lane : 0)) & FULL_REDO) != 0u);
        if ((dirty >> lane) & 1ull) {
            meta[3 * NS + redoCount + __popcll(dirty & below)] = wmeta_load(meta + lane);
            meta[2 * NS + lane] = 0;
            meta[NS + lane] = -1;
        }
        redoCount += mask_count(dirty);
        RZ_GSTAT(2, mask_count(dirty));
        freeCount += mask_count(dirty);
        ready &= ~dirty;
    }
    slot_sums<COUNT, GLASS>(K, slotsBase, meta, ready, freeCount);
#ifdef RZ_PROF
    c.t[17] += __builtin_amdgcn_s_memtime() - tss0_;
#endif
    if (COUNT) {
        tally_flush(K, c);
#ifdef RZ_PROF
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) if (c.p[k]) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) { for (int k = 9; k < 12; ++k) atomicAdd(&pr[19 + k], c.t[k]); for (int k = 12; k < 20; ++k) atomicAdd(&pr[108 + k], c.t[k]); }
        rz_prof_rounds(c, pr);
#endif
    }
    return write;
}

template <bool COUNT, bool GLASS, bool OVF, int COMPACT>      // COMPACT: 0 the group code, 1 the group code with trace_spread, or the units of a compacting claim (8 / 16)
__global__ __launch_bounds__(64, GLASS ? RZ_SAMPLES_MIN_WAVES_GLASS : RZ_SAMPLES_MIN_WAVES) void rz_render_samples(const KParams K, const unsigned nGroups, const unsigned perClaim,
                                                                                                                  const unsigned nClaims, const unsigned runShift) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
#ifdef RZ_PROF
    const unsigned long long wl_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned wl_claims = 0;
#endif
    if constexpr (COMPACT <= 1) {               // (a compacting launch always claims: its instantiations carry no copy of the group code)
        if (perClaim == 0) {                    // one workgroup per pixel group (small launches, spp < 64)
            render_samples_group<COUNT, GLASS, OVF, COMPACT == 1>(K, blockIdx.x, lds_raw);
            return;
        }
    }
    const ClaimMap M{nGroups, perClaim, nClaims, runShift};
    if constexpr (COMPACT > 1) {
        // every wait slot of this wave is free (count -1), nothing is parked, nothing is to be rendered again
        int* const meta = K.wmeta + (size_t)blockIdx.x * 4 * K.nWaitSlots;
        if ((int)(threadIdx.x & 63) < K.nWaitSlots) meta[K.nWaitSlots + (threadIdx.x & 63)] = -1;
        __syncthreads();
        int wpN = 0, freeCount = K.nWaitSlots, redoCount = 0, lateN = 0;
        const BlasStackT<OVF> bstk{reinterpret_cast<uint2*>(lds_raw) + (threadIdx.x & 63),
                                   OVF ? K.blasOvf + ((size_t)blockIdx.x * K.blasOvfCap) * 64 + (threadIdx.x & 63) : nullptr, K.blasStackCap};
        // (every iteration renders a claim or some redo groups, or ends the wave: the bound is a backstop)
        for (unsigned long long iter = 0; iter < (1ull << 40); ++iter) {
            const bool redo = GLASS && redoCount > 0;
            unsigned ci = 0;
            int nRedo = 0;
            if (redo) {
                nRedo = redoCount < (int)perClaim ? redoCount : (int)perClaim;      // a transparent scene's groups to render again come first
            } else {
                if ((threadIdx.x & 63) == 0) ci = atomicAdd(K.groupCounter, 1u);
                ci = (unsigned)__builtin_amdgcn_readfirstlane((int)ci);
                const bool more = ci < nClaims;         // every wave of the grid gets here with more == false in the end: the counter only grows
                const int need = more ? M.units_of(ci) : 0;
                // ONE place where the wave traces its pool: when it has filled up, when the next claim would find too few free wait
                // slots, after every claim if the launch was told so (RZ_CROSS_CLAIM_POOL=0), and -- generation after generation --
                // when the claims have run out.  Every pass moves its paths one bounce on and ends some of them: a path survives at
                // most maxBounces - 1 scatters, and with an empty pool every slot is free (the bound is a backstop).
                // (transparent scenes: a group's late samples are released one after the other, each for up to maxBounces passes more)
                const int maxPasses = (K.maxBounces + 1) * (GLASS ? RZ_GLATE_CAP + 1 : 1);
                for (int guard = 0; wpN > 0 && (!more || wpN >= (int)K.wpoolChunk || freeCount < need || K.drainEachClaim != 0) && guard <= maxPasses; ++guard)
                    wpN = pool_process<COUNT, OVF, GLASS>(K, K.wpool + (size_t)blockIdx.x * K.wpoolStride * RZ_GPOOL_FIELDS, K.wpoolStride, wpN, bstk, freeCount, redoCount, lateN);
                if (wpN > 0 && (!more || freeCount < need)) rz_backstop(K, 2u);      // (cannot be: every pass ends paths)
                if (!more) {
                    if (GLASS && redoCount > 0) continue;   // the last passes' groups to render again
                    break;
                }
                if (freeCount < need) { rz_backstop(K, 1u); break; }   // (cannot be: see above -- but a claim must not run without its slots; the claim's pixels stay unrendered and rz_sync says so)
            }
#ifdef RZ_PROF
            ++wl_claims;
#endif
            render_claim_compact<COUNT, OVF, COMPACT, GLASS>(K, M, ci, nRedo, redoCount, lds_raw, wpN, freeCount);
            redoCount -= nRedo;
        }
    } else {
        for (;;) {
            unsigned ci = 0;
            if ((threadIdx.x & 63) == 0) ci = atomicAdd(K.groupCounter, 1u);
            ci = (unsigned)__builtin_amdgcn_readfirstlane((int)ci);
            if (ci >= nClaims) break;               // every wave of the grid reaches this: the counter only grows
#ifdef RZ_PROF
            ++wl_claims;
#endif
            const int n = M.units_of(ci);
            for (int g = 0; g < n; ++g) render_samples_group<COUNT, GLASS, OVF, COMPACT == 1>(K, (unsigned)M.group(ci, g), lds_raw);
        }
    }
#ifdef RZ_PROF
    if ((threadIdx.x & 63) == 0 && blockIdx.x < (1u << 17)) {      // the persistent wave's lifetime and the claims it served
        rz_wave_log[blockIdx.x][0] = wl_t0;
        rz_wave_log[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        rz_wave_log[blockIdx.x][2] = wl_claims;
    }
#endif
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

#ifdef RZ_GSTATS
void dump_gstats() {
    unsigned long long h[8] = {};
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(rz_gstats), sizeof h) != hipSuccess) return;
    fprintf(stderr, "[rz_gstats] units %llu  resolved in the wave %llu  groups rendered again %llu  re-run rounds %llu  paths parked %llu  pooled paths that met glass %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
    unsigned long long z[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rz_gstats), z, sizeof z);
}
#endif
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
    {   // third word of the log: claims served (persistent sample kernel) / hardware id (pixel kernel)
        std::vector<unsigned long long> w3(nWaves);
        for (int i = 0; i < nWaves; ++i) w3[i] = h[3 * i + 2];
        std::sort(w3.begin(), w3.end());
        fprintf(stderr, "[rz_prof] third log word per wave (claims served): min %llu  p10 %llu  med %llu  p90 %llu  max %llu\n", w3[0], w3[nWaves / 10], w3[nWaves / 2], w3[nWaves * 9 / 10], w3[nWaves - 1]);
        std::vector<double> ends(nWaves);       // a long tail of ends = the launch waits for the last claims handed out
        for (int i = 0; i < nWaves; ++i) ends[i] = (double)(h[3 * i + 1] - t0) * 1e-5;
        std::sort(ends.begin(), ends.end());
        fprintf(stderr, "[rz_prof] wave END times (ms from the first start): p1 %.3f  p10 %.3f  med %.3f  p90 %.3f  p99 %.3f  max %.3f\n", ends[nWaves / 100], ends[nWaves / 10], ends[nWaves / 2], ends[nWaves * 9 / 10], ends[nWaves * 99 / 100], ends[nWaves - 1]);
    }
    fprintf(stderr, "[rz_prof] resident waves per 5%% time slice:");
    for (int s = 0; s < NS; ++s) fprintf(stderr, " %.0f", res[s]);
    fprintf(stderr, "\n");
}
#endif

// How a frame is launched: one workgroup per pixel group, or a persistent grid claiming `perClaim` groups per atomic.
// Persistent waves pay off when a group is one pixel's 64-sample batches and there are many groups per resident wave
// (C2: 17.2 -> 16.8 ms, C5: 226 -> 203 ms).  With several pixels per wave (spp < 64) or a small frame the hardware
// dispatcher is the better scheduler (C4: 14.6 ms against 15.4-19.9 ms persistent; C1: 0.04 against 0.1-0.2).
SamplesPlan plan_render_samples(int spp, int nSlots, bool glass) {
    SamplesPlan p{};
    const int pixPerWave = spp >= 64 ? 1 : 64 / spp;
    p.groups = ((long long)nSlots + pixPerWave - 1) / pixPerWave;
    static int nCU = 0;
    if (nCU == 0) {
        int dev = 0; hipDeviceProp_t prop;
        nCU = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    const int nBatches = (spp + 63) / 64;
    const long long units = p.groups * nBatches;        // (pixel, 64-sample batch) units of work in this launch
    // A claim is sized in WORK, in units, whatever the spp (sized in pixels -- 8 -- one rank's share of an 8-GPU
    // weak-scaling frame, 1/8 of the tiles at 512 spp, was 4 050 claims for 4 096 waves: 21.9 ms for the work of an 11.3-ms
    // frame), and by the size of the launch:
    //  *  4 units under 3/4 M units (a 1/4 or 1/8 share of a 1080p frame at 64 spp): a claim in the middle of the bunny lasts
    //     ten times the average and the launch cannot end before the last such claim has run -- with 8-unit claims a 1/8-frame
    //     launch took 2.80 ms for 1.41 ms of work, with 4-unit claims 1.93 (rank_share.py; the RZ_PROF wave log shows the
    //     resident waves draining over the second half of such a launch);
    //  *  8 units for a 1080p frame at 64-128 spp: its claims compact better than 4-unit ones (11.3 against 12.3 ms; a half
    //     frame 5.87 against 6.15) and its tail is 0.3 ms; 12 and 16 units: 11.04 -> 11.10 / 11.27 ms;
    //  * 16 units from 4 M units on (C3: 256 spp at 1080p; C5: 128 spp at 4K): more pixels per claim, so more paths to compact
    //     and more lanes in the ordered sums, and the tail no longer shows: C3 44.7 -> 43.0 ms, C5 118.2 -> 115.6.
    //  *  6 groups for launches of several pixels per wave under 3/4 M groups (C4: 16 spp at 1080p, 518 400 groups of four pixels;
    //     round 5, profiles/r05_c4/claim_sweep_c4.log: 1 / 2 / 3 / 4 / 6 / 8 groups per claim -> 9.42 / 6.94 / 6.70 / 6.81 / 6.64 /
    //     7.00 ms; RayZen's scene at 16 spp in its window: 1.89 / 1.41 / 1.37 / 1.40 / 1.41 / 1.45): a claim of four-pixel groups
    //     parks more paths than one of single pixels, the pool fills sooner and the stratified claims (below) keep the tail short.
    const int claimUnits = units >= (4ll << 20) ? RZ_CLAIM_UNITS_LARGE
                         : (units >= (3ll << 18) ? RZ_GROUPS_PER_CLAIM : (spp < 64 ? RZ_GROUPS_PER_CLAIM_SMALL_SPP : std::max(1, RZ_GROUPS_PER_CLAIM / 2)));
    // Launches of several pixels per wave (spp < 64) take the persistent, compacting grid too (round 3): their late bounces
    // are where their time goes -- C4, 16 spp: rounds 4 and 5 of a path (bounces 2 and 3) ran 7 and 3 lanes wide and took 39 %
    // of the traversal's wave cycles (profiles/r03_c4_before/) -- and only a claim of several units has enough parked paths to
    // fill waves with.  RZ_SMALL_SPP_CLAIMS=0 restores one workgroup per group (A/B aid).
    bool smallSppClaims = true;
    if (const char* e = std::getenv("RZ_SMALL_SPP_CLAIMS")) smallSppClaims = std::atoi(e) != 0;
    const bool persistent = (spp >= 64 || smallSppClaims) && p.groups >= (long long)nCU * RZ_PERSIST_WAVES_PER_CU * 16;
    p.perClaim = persistent ? std::max(1, claimUnits / nBatches) : 0;
    if (const char* e = std::getenv("RZ_GROUPS_PER_CLAIM")) p.perClaim = std::max(0, std::atoi(e));      // tuning aid
    // ray compaction across the units of a claim (render_claim_compact, instantiated for 8 and for 16 units): opaque scenes,
    // persistent launches, up to 16 batches per pixel (a unit is ONE pixel's batch of 64 samples)
    bool compact = RZ_COMPACT_DEFAULT != 0;
    if (const char* e = std::getenv("RZ_COMPACT")) compact = std::atoi(e) != 0;                            // A/B aid
    // (measured: C2, 64 spp, 14.29 -> 13.28 ms; 128 spp C5 144.2 -> 143.0.  Several batches per pixel -- a claim is then a few
    //  pixels' batches: at mid-round the gain was gone at 256 spp, on the final code it is back: C3 47.3 -> 44.7 ms, one rank's
    //  share of a 4- / 8-GPU weak-scaling frame (256 / 512 spp) 11.88 -> 11.25 / 11.85 -> 11.70 ms)
    // Transparent scenes compact too (round 4: glass_resolve_unit); RZ_GLASS_CLAIMS=0 keeps them on the speculating group code (A/B aid).
    bool glassClaims = true;
    if (const char* e = std::getenv("RZ_GLASS_CLAIMS")) glassClaims = std::atoi(e) != 0;
    p.compact = compact && (!glass || glassClaims) && p.perClaim > 0 && nBatches <= RZ_CLAIM_UNITS_LARGE;
    if (spp < 64 && !p.compact) p.perClaim = 0;          // (the plain persistent loop never paid for several pixels per wave: C4 9.4 -> 10.8 ... 19.9 ms)
    p.claimUnits = 0;
    if (p.compact) {
        p.claimUnits = (p.perClaim * nBatches > RZ_CLAIM_UNITS_SMALL) ? RZ_CLAIM_UNITS_LARGE : RZ_CLAIM_UNITS_SMALL;
        p.perClaim = std::max(1, std::min(p.perClaim, p.claimUnits / nBatches));
    }
    const long long claims = p.perClaim ? (p.groups + p.perClaim - 1) / p.perClaim : p.groups;
    p.grid = p.perClaim ? std::min<long long>(claims, (long long)nCU * RZ_PERSIST_WAVES_PER_CU) : p.groups;
    p.nClaims = p.perClaim ? claims : 0;
    // The waves of a compacting launch keep their pool of parked paths ACROSS claims (pool_process): a claim's own parked paths
    // are a few dozen, and lanes can only refill from a list much longer than the wave is wide.  RZ_CROSS_CLAIM_POOL=0 makes
    // every wave trace its pool to the end after each claim instead (round 2's per-claim pools; A/B and test aid).
    p.drainEachClaim = false;
    if (const char* e = std::getenv("RZ_CROSS_CLAIM_POOL")) p.drainEachClaim = std::atoi(e) == 0;
    // the shape of a claim (ClaimMap): one run of consecutive groups at 64 spp and more (a tile row; its parked paths are
    // neighbours), runs of RZ_CLAIM_RUN_SMALL_SPP groups from as many bands of the frame as it takes below (a claim of
    // consecutive 4-pixel groups inside a mesh costs many times the average claim, and the launch waits for the last one)
    int run = spp >= 64 ? p.perClaim : RZ_CLAIM_RUN_SMALL_SPP;
    if (const char* e = std::getenv("RZ_CLAIM_RUN")) run = std::max(1, std::atoi(e));                        // tuning aid
    if (run > p.perClaim) run = p.perClaim;
    if (run <= 0 || (run & (run - 1)) != 0 || p.perClaim % run != 0) run = 1;      // runs are a power of two that divides the claim; anything else: single groups
    p.runShift = 0;
    while ((1 << (p.runShift + 1)) <= run) ++p.runShift;
    return p;
}
size_t samples_lds_extra(bool glass, bool compact) {      // LDS per wave besides the two stacks
    if (compact) return 0;                                // addends live in the claim scratch
    return glass ? 4 * 64 * sizeof(float4) + 3 * 64 * sizeof(int) : 6 * 64 * sizeof(float);
}

void launch_render_samples(const KParams& K, bool counted, bool glass, hipStream_t stream) {
    const SamplesPlan plan = plan_render_samples(K.spp, K.nSlots, glass);
    const long long blocks = plan.groups, grid = plan.grid, perClaim = plan.perClaim;
    if (blocks <= 0) return;
    const bool compact = plan.compact && K.wpool != nullptr;
    const size_t lds = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int) + samples_lds_extra(glass, compact);
    const dim3 g((unsigned)grid), b(64);
    const unsigned nGroups = (unsigned)blocks;
    if (perClaim && hipMemsetAsync(K.groupCounter, 0, sizeof(unsigned), stream) != hipSuccess) return;
    const bool ovf = K.blasOvfCap > 0;           // only set for persistent launches (rz_context.hip: render_samples)
#define RZ_LAUNCH_SAMPLES(C, G, O, M) hipLaunchKernelGGL((rz_render_samples<C, G, O, M>), g, b, lds, stream, K, nGroups, (unsigned)perClaim, (unsigned)plan.nClaims, (unsigned)plan.runShift)
    // (trace_spread for the third and later segments: launches of several pixels per wave over scenes of several instances)
    const bool spread = K.spreadTrace != 0 && K.spp < 64;
    if (glass && compact && plan.claimUnits == RZ_CLAIM_UNITS_LARGE) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, true, true, RZ_CLAIM_UNITS_LARGE); else RZ_LAUNCH_SAMPLES(true, true, false, RZ_CLAIM_UNITS_LARGE); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, true, true, RZ_CLAIM_UNITS_LARGE); else RZ_LAUNCH_SAMPLES(false, true, false, RZ_CLAIM_UNITS_LARGE); }
    } else if (glass && compact) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, true, true, RZ_CLAIM_UNITS_SMALL); else RZ_LAUNCH_SAMPLES(true, true, false, RZ_CLAIM_UNITS_SMALL); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, true, true, RZ_CLAIM_UNITS_SMALL); else RZ_LAUNCH_SAMPLES(false, true, false, RZ_CLAIM_UNITS_SMALL); }
    } else if (glass && spread) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, true, true, 1); else RZ_LAUNCH_SAMPLES(true, true, false, 1); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, true, true, 1); else RZ_LAUNCH_SAMPLES(false, true, false, 1); }
    } else if (glass) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, true, true, 0); else RZ_LAUNCH_SAMPLES(true, true, false, 0); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, true, true, 0); else RZ_LAUNCH_SAMPLES(false, true, false, 0); }
    } else if (compact && plan.claimUnits == RZ_CLAIM_UNITS_LARGE) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, false, true, RZ_CLAIM_UNITS_LARGE); else RZ_LAUNCH_SAMPLES(true, false, false, RZ_CLAIM_UNITS_LARGE); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, false, true, RZ_CLAIM_UNITS_LARGE); else RZ_LAUNCH_SAMPLES(false, false, false, RZ_CLAIM_UNITS_LARGE); }
    } else if (compact) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, false, true, RZ_CLAIM_UNITS_SMALL); else RZ_LAUNCH_SAMPLES(true, false, false, RZ_CLAIM_UNITS_SMALL); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, false, true, RZ_CLAIM_UNITS_SMALL); else RZ_LAUNCH_SAMPLES(false, false, false, RZ_CLAIM_UNITS_SMALL); }
    } else if (spread) {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, false, true, 1); else RZ_LAUNCH_SAMPLES(true, false, false, 1); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, false, true, 1); else RZ_LAUNCH_SAMPLES(false, false, false, 1); }
    } else {
        if (counted) { if (ovf) RZ_LAUNCH_SAMPLES(true, false, true, 0); else RZ_LAUNCH_SAMPLES(true, false, false, 0); }
        else { if (ovf) RZ_LAUNCH_SAMPLES(false, false, true, 0); else RZ_LAUNCH_SAMPLES(false, false, false, 0); }
    }
#undef RZ_LAUNCH_SAMPLES
}

// The local hemisphere direction of a zero seed (rz_path.h: random_hemisphere_direction), by the kernels' own arithmetic.
__global__ void rz_hemi0_kernel(float* out) {
    v2 z; z.x = 0.0f; z.y = 0.0f;
    const v3 d = hemisphere_local(z);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
int compute_hemi0(float out[3], hipStream_t stream) {
    float* d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), 3 * sizeof(float));
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(rz_hemi0_kernel, dim3(1), dim3(1), 0, stream, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, 3 * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -(int)e;
}

void launch_resolve(const float4* accum, uchar4* out, int n, hipStream_t stream) {
    hipLaunchKernelGGL(rz_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, out, n);
}

}  // namespace rz
