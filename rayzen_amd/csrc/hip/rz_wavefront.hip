// rz_wavefront.hip -- the queued ("wavefront") form of the render loop for gfx950.
//
// The one-lane-per-pixel kernel (rz_kernels.hip) keeps a path's whole state in registers across the BVH walk:
// 181 VGPRs, 2 waves per SIMD, and measured VALU-busy of 30 % -- the walk is a chain of dependent L1/L2 loads and
// two waves cannot hide it.  Here the loop is cut at its natural seam, the closest-hit query:
//
//   wf_init   all owned pixels: first camera ray of the frame, state parked in HBM, slot queued
//   repeat until the queue is empty ("round"):
//     wf_trace  persistent waves pull rays from the queue; a lane carries ONLY ray + traversal state
//               (~80 VGPRs -> 4+ waves/SIMD); TLAS and BLAS are one flat per-lane state machine, so a lane
//               that finishes early is RE-FILLED from the queue (wave ballot + popcount prefix + one atomicAdd
//               per wave) while its neighbours keep walking; leaves are batched as in rz_trace.h
//     wf_shade  one lane per queued path: rebuild the hit, run the same `advance` state machine as the pixel
//               kernel (rz_path.h), start the pixel's next sample when one ends (so a slot stays queued until
//               all its samples are done), park the state, and COMPACT the survivors into the next queue
//               (ballot / popcount prefix sum / one atomicAdd per wave).
//
// A path's arithmetic does not depend on which lane, wave or round executes it, so results are bit-identical to
// the pixel kernel and to the oracle; only the schedule changes.  Per-pixel sample order (the shader's
// currentIor and colour sum are sequential per pixel, FS:672-674) is kept because a pixel owns exactly one slot.
#include <hip/hip_runtime.h>

#include "rayzen_hip.h"
#include "rz_path.h"
#include "rz_wavefront.h"

namespace rz {

#ifndef RZ_WF_REFILL_MIN
#define RZ_WF_REFILL_MIN 20      // re-fill a wave's idle lanes once this many are idle
#endif
#ifndef RZ_WF_DESCEND_MIN
#define RZ_WF_DESCEND_MIN 12     // leave the descend loop when fewer lanes than this still have an internal node
#endif

enum : int { ST_IDLE = 0, ST_TLAS = 1, ST_BLAS = 2, ST_DONE = 3 };

__device__ __forceinline__ void flush_tally(const KParams& K, const Tally& c) {
    DevCounters* g = K.counters;
    if (c.samples) atomicAdd(&g->samples, (unsigned long long)c.samples);
    if (c.traversals) atomicAdd(&g->traversals, (unsigned long long)c.traversals);
    if (c.tlas_nodes) atomicAdd(&g->tlas_nodes, (unsigned long long)c.tlas_nodes);
    if (c.tlas_leaf_indices) atomicAdd(&g->tlas_leaf_indices, (unsigned long long)c.tlas_leaf_indices);
    if (c.instances) atomicAdd(&g->instances, (unsigned long long)c.instances);
    if (c.blas_nodes) atomicAdd(&g->blas_nodes, (unsigned long long)c.blas_nodes);
    if (c.triangles) atomicAdd(&g->triangles, (unsigned long long)c.triangles);
    if (c.materials) atomicAdd(&g->materials, (unsigned long long)c.materials);
    if (c.light_fetches) atomicAdd(&g->light_fetches, (unsigned long long)c.light_fetches);
}

// slot <-> pixel: slot = localTile * 64 + lane-in-tile
__device__ __forceinline__ bool slot_pixel(const KParams& K, int slot, int& px, int& py) {
    const int localTile = slot >> 6, l = slot & 63;
    const int tile = localTile * K.tileNRanks + K.tileRank;
    const int tx = tile % K.tilesX, ty = tile / K.tilesX;
    px = tx * RZ_TILE_W + (l & 7);
    py = ty * RZ_TILE_H + (l >> 3);
    return px < K.width && py < K.height;
}

__device__ __forceinline__ void park(const WFParams& Q, int slot, const Path& P) {
    Q.rayO[slot] = make_float4(P.o.x, P.o.y, P.o.z, 0.0f);
    Q.rayD[slot] = make_float4(P.d.x, P.d.y, P.d.z, 0.0f);
    Q.s0[slot] = make_float4(P.throughput.x, P.throughput.y, P.throughput.z, P.ior);
    const unsigned packed = (unsigned)P.bounce | ((unsigned)P.mode << 12) | ((unsigned)P.iter << 14) | ((unsigned)P.li << 20);
    Q.s1[slot] = make_float4(P.seed.x, P.seed.y, __int_as_float(P.samp), __uint_as_float(packed));
    Q.s2[slot] = make_float4(P.hp.x, P.hp.y, P.hp.z, __int_as_float(P.hmat));
    Q.s3[slot] = make_float4(P.hn.x, P.hn.y, P.hn.z, P.vis);
    Q.s4[slot] = make_float4(P.pdir.x, P.pdir.y, P.pdir.z, P.traveled);
    Q.s5[slot] = make_float4(P.lacc.x, P.lacc.y, P.lacc.z, P.maxDist);
}

__device__ __forceinline__ void unpark(const WFParams& Q, int slot, Path& P) {
    const float4 o = Q.rayO[slot], d = Q.rayD[slot], a = Q.s0[slot], b = Q.s1[slot], c = Q.s2[slot], e = Q.s3[slot],
                 f = Q.s4[slot], g = Q.s5[slot];
    P.o = mk3(o.x, o.y, o.z); P.d = mk3(d.x, d.y, d.z);
    P.throughput = mk3(a.x, a.y, a.z); P.ior = a.w;
    P.seed.x = b.x; P.seed.y = b.y; P.samp = __float_as_int(b.z);
    const unsigned packed = __float_as_uint(b.w);
    P.bounce = (int)(packed & 0xFFFu); P.mode = (int)((packed >> 12) & 3u); P.iter = (int)((packed >> 14) & 63u);
    P.li = (int)(packed >> 20);
    P.hp = mk3(c.x, c.y, c.z); P.hmat = __float_as_int(c.w);
    P.hn = mk3(e.x, e.y, e.z); P.vis = e.w;
    P.pdir = mk3(f.x, f.y, f.z); P.traveled = f.w;
    P.lacc = mk3(g.x, g.y, g.z); P.maxDist = g.w;
}

// Append `alive` lanes' slots to the out queue: ballot + popcount prefix + one atomicAdd per wave.
__device__ __forceinline__ void enqueue(int* __restrict__ queue, int* __restrict__ count, bool alive, int slot) {
    const unsigned long long m = __ballot(alive);
    if (m == 0ull) return;
    const int lane = (int)__lane_id();
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, leader);
    if (alive) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = slot;
}

// ---------------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(256) void wf_init(const KParams K, const WFParams Q) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    Tally c = {};
    bool alive = false;
    if (slot < Q.nSlots) {
        int px, py;
        if (slot_pixel(K, slot, px, py)) {
            const size_t pix = (size_t)py * K.width + px;
            Path P;
            const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
            P.uv.x = fragx / (float)K.width;
            P.uv.y = fragy / (float)K.height;
            P.fragSum = fragx + fragy;
            P.samp = K.sampleBase;
            P.sampEnd = K.sampleBase + K.spp;
            if (K.sampleBase == 0) {
                K.accum[pix] = make_float4(0.0f, 0.0f, 0.0f, (float)K.spp);
                P.ior = 1.0f;
            } else {
                float4 a = K.accum[pix];
                a.w += (float)K.spp;
                K.accum[pix] = a;
                P.ior = K.ior[pix];
            }
            P.hp = P.hn = P.pdir = P.lacc = mk3(0.0f, 0.0f, 0.0f);
            P.hmat = 0; P.li = 0; P.iter = 0; P.vis = 0.0f; P.traveled = 0.0f; P.maxDist = 0.0f;
            begin_sample<COUNT>(K, P, c);
            park(Q, slot, P);
            alive = true;
            if (COUNT) atomicAdd(&K.counters->pixels, 1ull);
        }
    }
    enqueue(Q.queue[0], &Q.counts[0], alive, slot);
    if (COUNT) flush_tally(K, c);
}

// ---------------------------------------------------------------------------------------------------------
// Closest hit for every queued ray.  One flat state machine per lane:
//   ST_TLAS  pop TLAS nodes / walk the instances of a TLAS leaf until a BLAS is entered (FS:464-500)
//   ST_BLAS  inside one instance's BLAS (FS:426-452, restructured as in rz_trace.h)
//   ST_DONE  result ready to be written; ST_IDLE lane has no ray
template <bool COUNT>
__global__ __launch_bounds__(256) void wf_trace(const KParams K, const WFParams Q, const int round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t perWave = (size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int);
    unsigned char* base = lds_raw + perWave * wave;
    uint2* bstk = reinterpret_cast<uint2*>(base) + lane;
    int* tstk = reinterpret_cast<int*>(base + (size_t)K.blasStackCap * 64 * sizeof(uint2)) + lane;

    const int count = Q.counts[round];
    int* const cursor = &Q.cursors[round];
    const int* __restrict__ queue = Q.queue[round & 1];

    Tally c = {};
    int stage = ST_IDLE, slot = -1;
    v3 o = mk3(0, 0, 0), d = o, inv = o;
    float tHit = 1e30f, bestTLoc = 0.0f;
    int bestTri = -1, bestInst = -1;
    int tsp = 0, tlf = 0, tlc = 0, ti = 0;
    int instIdx = -1;
    const DevInstance* __restrict__ I = K.instances;
    int pairBase = 0, triBase = 0;
    v3 lo = o, ld = o, linv = o;
    float tLoc = 1e30f;
    int best = -1, cur = 0, sp = 0;
    bool more = count > 0;
    // chunk size: enough chunks that every wave of the grid gets several, never less than one wave of rays
    const int totalWaves = gridDim.x * 4;
    int chunk = (count / (totalWaves * 4) + 63) & ~63;
    chunk = max(64, min(chunk, 2048));
    int chunkPos = 0, chunkEnd = 0;

#ifdef RZ_PROF
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};
#define RZ_T0 unsigned long long t_ = __builtin_amdgcn_s_memtime()
#define RZ_T(k) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); tc[k] += n_ - t_; t_ = n_; } while (0)
#else
#define RZ_T0 do { } while (0)
#define RZ_T(k) do { } while (0)
#endif
    for (;;) {
        RZ_T0;
        RZ_SITE(c, 0);
        // ---- re-fill idle lanes from the queue.  The wave owns a private chunk [chunkPos, chunkEnd) of the queue
        // and only goes to the global cursor when that is used up: one word takes ~88 atomics/us on this chip
        // (MI355X_MICROARCH.md, "dequeue"), and one atomic per ~20 rays made the kernel atomic-bound (v1: 479 us
        // per round).
        const unsigned long long idle = __ballot(stage == ST_IDLE);
        const int nIdle = __popcll(idle);
        if (nIdle >= RZ_WF_REFILL_MIN && (more || chunkPos < chunkEnd)) {
            if (chunkPos >= chunkEnd) {          // wave-uniform
                int b = 0;
                if (lane == 0) b = atomicAdd(cursor, chunk);
                b = __shfl(b, 0);
                chunkPos = b;
                chunkEnd = min(b + chunk, count);
                if (b + chunk >= count) more = false;
            }
            if (chunkPos < chunkEnd) {
                const int take = min(nIdle, chunkEnd - chunkPos);
                if (stage == ST_IDLE) {
                    const int r = __popcll(idle & ((1ull << lane) - 1ull));
                    if (r < take) {
                        RZ_SITE(c, 1);
                        slot = queue[chunkPos + r];
                        const float4 ro = Q.rayO[slot], rd = Q.rayD[slot];
                        o = mk3(ro.x, ro.y, ro.z);
                        d = mk3(rd.x, rd.y, rd.z);
                        inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        tHit = 1e30f; bestTri = -1; bestInst = -1; bestTLoc = 0.0f;
                        tlc = 0; ti = 0; tsp = 0;
                        if (K.nTlasNodes > 0) { tstk[0] = 0; tsp = 1; }
                        if (COUNT) c.traversals += 1;
                        stage = ST_TLAS;
                    }
                }
                chunkPos += take;
            }
        }
        if (__ballot(stage != ST_IDLE) == 0ull) {
            if (!more && chunkPos >= chunkEnd) break;
            continue;       // everything idle but rays remain: fetch again (nIdle == 64 >= REFILL_MIN)
        }

        RZ_T(0);
        // ---- TLAS level: runs until the lane has entered a BLAS or finished the query
        while (stage == ST_TLAS) {
            RZ_SITE(c, 2);
            if (ti < tlc) {
                instIdx = K.tlasIndices[tlf + ti];
                ++ti;
                I = K.instances + instIdx;
                if (COUNT) { c.tlas_leaf_indices += 1; c.instances += 1; c.blas_nodes += 1; }
                lo = x34_point(I->inv, o);
                ld = normalize(x34_dir(I->inv, d));
                linv = mk3(1.0f / ld.x, 1.0f / ld.y, 1.0f / ld.z);
                tLoc = 1e30f; best = -1; sp = 0;
                float tminRoot;
                bool go = slab(lo, linv, I->rootMin[0], I->rootMin[1], I->rootMin[2], I->rootMax[0], I->rootMax[1],
                               I->rootMax[2], tminRoot);
                go = go && !(tminRoot > tLoc) && !(I->flags & 1);
                if (go) { cur = I->rootEnc; pairBase = I->pairBase; triBase = I->triBase; stage = ST_BLAS; }
            } else if (tsp > 0) {
                --tsp;
                const int nidx = tstk[tsp * 64];
                const float4* __restrict__ np = reinterpret_cast<const float4*>(K.tlasNodes + nidx);
                const float4 n0 = np[0], n1 = np[1];
                if (COUNT) c.tlas_nodes += 1;
                float tmin;
                if (!slab(o, inv, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z, tmin) || tmin > tHit) continue;
                const int leftFirst = __float_as_int(n0.w), cnt = __float_as_int(n1.w);
                if (cnt > 0) { tlf = leftFirst; tlc = cnt; ti = 0; }
                else if (cnt < 0 && tsp + 2 <= K.tlasStackCap) {
                    tstk[tsp * 64] = leftFirst; ++tsp;
                    tstk[tsp * 64] = leftFirst + 1; ++tsp;
                }
            } else {
                stage = ST_DONE;
            }
        }

        RZ_T(1);
        // ---- BLAS: descend internal nodes (see rz_trace.h for why this equals the shader's order)
        bool finished = false;      // this lane's BLAS walk ended in this iteration
        {
            const DevPair* __restrict__ pairs = K.pairs + pairBase;
            while (stage == ST_BLAS && cur >= 0 && !finished) {
                RZ_SITE(c, 3);
                const float4* __restrict__ pp = reinterpret_cast<const float4*>(pairs + cur);
                const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                if (COUNT) c.blas_nodes += 2;
                float tl, tr;
                bool hl, hr;
                const RayPk RP = make_raypk(lo, linv);
                const f32x2 lx = {p0.x, p0.y}, ly = {p0.z, p0.w}, lz = {p1.x, p1.y};
                const f32x2 rx = {p1.z, p1.w}, ry = {p2.x, p2.y}, rz = {p2.z, p2.w};
                RZ_SLAB_PAIR(-1, RP, lx, ly, lz, rx, ry, rz, hl, tl, hr, tr);
                if (hl) {
                    bstk[sp * 64] = make_uint2((unsigned)__float_as_int(p3.x), __float_as_uint(tl));
                    ++sp;
                }
                if (hr && !(tr > tLoc)) {
                    cur = __float_as_int(p3.y);
                } else {
                    finished = true;
                    while (sp > 0) {
                        --sp;
                        const uint2 e = bstk[sp * 64];
                        if (__uint_as_float(e.y) > tLoc) continue;
                        cur = (int)e.x;
                        finished = false;
                        break;
                    }
                }
                if (__popcll(__ballot(stage == ST_BLAS && cur >= 0 && !finished)) < RZ_WF_DESCEND_MIN) break;
            }
        }
        RZ_T(2);
        // ---- BLAS: leaves, batched
        if (stage == ST_BLAS && !finished && cur < 0) {
            RZ_SITE(c, 4);
            const DevTri* __restrict__ tris = K.tris + triBase;
            const int v = ~cur;
            const int first = v >> 4, cnt = v & 15;
            if (COUNT) c.triangles += (unsigned)cnt;
            for (int i = 0; i < cnt; ++i) {
                RZ_SITE(c, 5);
                const float4* __restrict__ tp = reinterpret_cast<const float4*>(tris + first + i);
                const float4 a = tp[0], b = tp[1], cc = tp[2];
                float t;
                if (moller_trumbore(lo, ld, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, cc.x), t)) {
                    if (t < tLoc) { tLoc = t; best = first + i; }
                }
            }
            finished = true;
            while (sp > 0) {
                --sp;
                const uint2 e = bstk[sp * 64];
                if (__uint_as_float(e.y) > tLoc) continue;
                cur = (int)e.x;
                finished = false;
                break;
            }
        }
        RZ_T(3);
        // ---- this instance is done: FS:482-495, then back to the TLAS level
        if (stage == ST_BLAS && finished) {
            RZ_SITE(c, 6);
            if (best >= 0) {
                const v3 localHit = lo + ld * tLoc;
                const v3 worldHit = x34_point(I->fwd, localHit);
                const float tWorld = length(worldHit - o);
                if (tWorld < tHit) { tHit = tWorld; bestTri = best + triBase; bestInst = instIdx; bestTLoc = tLoc; }
            }
            stage = ST_TLAS;
        }
        // ---- result
        if (stage == ST_DONE) {
            Q.hit[slot] = make_float4(tHit, bestTLoc, __int_as_float(bestTri), __int_as_float(bestInst));
            stage = ST_IDLE;
        }
        RZ_T(4);
    }
#ifdef RZ_PROF
    if (COUNT) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(K.counters + 1);
        for (int k = 0; k < 16; ++k) if (c.p[k]) atomicAdd(&pr[k], (unsigned long long)c.p[k]);
        if (lane == 0) for (int k = 0; k < 5; ++k) atomicAdd(&pr[16 + k], tc[k]);
    }
#endif
    if (COUNT) flush_tally(K, c);
}

// Rebuild the hit record from what wf_trace stored: the same expressions trace_closest evaluates (FS:410-412,
// 484-491), all pure functions of (ray, instance, triangle, local t).
__device__ __forceinline__ bool rebuild_hit(const KParams& K, v3 o, v3 d, float4 hr, HitRec& h) {
    const int tri = __float_as_int(hr.z), inst = __float_as_int(hr.w);
    if (tri < 0) return false;
    const DevInstance* __restrict__ I = K.instances + inst;
    const v3 lo = x34_point(I->inv, o);
    const v3 ld = normalize(x34_dir(I->inv, d));
    const v3 localHit = lo + ld * hr.y;
    const float4* __restrict__ tp = reinterpret_cast<const float4*>(K.tris + tri);
    const float4 a = tp[0], b = tp[1], cc = tp[2];
    const v3 ln = normalize(cross(mk3(a.w, b.x, b.y), mk3(b.z, b.w, cc.x)));
    h.t = hr.x;
    h.p = x34_point(I->fwd, localHit);
    h.n = normalize(x34_normal(I->inv, ln));
    h.mat = __float_as_int(cc.y);
    h.inst = inst;
    return true;
}

// ---------------------------------------------------------------------------------------------------------
constexpr int SHADE_ITEMS = 4;                       // queue entries per thread and block pass
constexpr int SHADE_CHUNK = 256 * SHADE_ITEMS;       // one global atomic per this many entries

template <bool COUNT>
__global__ __launch_bounds__(256) void wf_shade(const KParams K, const WFParams Q, const int round) {
    __shared__ int sSlots[SHADE_CHUNK];
    __shared__ int sCount, sBase;
    const int count = Q.counts[round];
    const int* __restrict__ qin = Q.queue[round & 1];
    int* __restrict__ qout = Q.queue[(round + 1) & 1];
    int* const countOut = &Q.counts[round + 1];
    const int lane = threadIdx.x & 63;
    Tally c = {};
    for (int chunk0 = blockIdx.x * SHADE_CHUNK; chunk0 < count; chunk0 += gridDim.x * SHADE_CHUNK) {
        if (threadIdx.x == 0) sCount = 0;
        __syncthreads();
        for (int it = 0; it < SHADE_ITEMS; ++it) {
            const int i = chunk0 + it * 256 + threadIdx.x;
            bool alive = false;
            int slot = -1;
            if (i < count) {
                slot = qin[i];
                int px, py;
                slot_pixel(K, slot, px, py);
                const size_t pix = (size_t)py * K.width + px;
                Path P;
                unpark(Q, slot, P);
                const float fragx = (float)px + 0.5f, fragy = (float)py + 0.5f;
                P.uv.x = fragx / (float)K.width;
                P.uv.y = fragy / (float)K.height;
                P.fragSum = fragx + fragy;
                P.sampEnd = K.sampleBase + K.spp;
                const float4 acc = K.accum[pix];
                P.color = mk3(acc.x, acc.y, acc.z);
                const v3 color0 = P.color;
                HitRec h;
                const bool found = rebuild_hit(K, P.o, P.d, Q.hit[slot], h);
                advance<COUNT>(K, P, found, h, c);
                if (P.mode == MODE_DONE && P.samp < P.sampEnd) begin_sample<COUNT>(K, P, c);
                alive = P.mode != MODE_DONE;
                if (__float_as_uint(P.color.x) != __float_as_uint(color0.x) ||
                    __float_as_uint(P.color.y) != __float_as_uint(color0.y) ||
                    __float_as_uint(P.color.z) != __float_as_uint(color0.z))
                    K.accum[pix] = make_float4(P.color.x, P.color.y, P.color.z, acc.w);
                if (alive) park(Q, slot, P);
                else K.ior[pix] = P.ior;
            }
            // compaction, level 1: wave ballot + popcount prefix, one LDS atomic per wave
            const unsigned long long m = __ballot(alive);
            if (m != 0ull) {
                const int leader = __ffsll((long long)m) - 1;
                int b = 0;
                if (lane == leader) b = atomicAdd(&sCount, __popcll(m));
                b = __shfl(b, leader);
                if (alive) sSlots[b + __popcll(m & ((1ull << lane) - 1ull))] = slot;
            }
        }
        __syncthreads();
        // level 2: one global atomic per block pass, then a coalesced copy of the survivors
        const int n = sCount;
        if (threadIdx.x == 0 && n > 0) sBase = atomicAdd(countOut, n);
        __syncthreads();
        if (n > 0) {
            const int b = sBase;
            for (int k = threadIdx.x; k < n; k += 256) qout[b + k] = sSlots[k];
        }
        __syncthreads();
    }
    if (COUNT) flush_tally(K, c);
}

// ---------------------------------------------------------------------------------------------------------
size_t wf_trace_lds_bytes(const KParams& K) {
    return ((size_t)K.blasStackCap * 64 * sizeof(uint2) + (size_t)K.tlasStackCap * 64 * sizeof(int)) * 4;
}

void launch_wf_init(const KParams& K, const WFParams& Q, bool counted, hipStream_t s) {
    const int blocks = (Q.nSlots + 255) / 256;
    if (blocks <= 0) return;
    if (counted) hipLaunchKernelGGL(wf_init<true>, dim3(blocks), dim3(256), 0, s, K, Q);
    else hipLaunchKernelGGL(wf_init<false>, dim3(blocks), dim3(256), 0, s, K, Q);
}

void launch_wf_round(const KParams& K, const WFParams& Q, int round, int traceBlocks, int shadeBlocks, bool counted,
                     hipStream_t s) {
    const size_t lds = wf_trace_lds_bytes(K);
    if (counted) {
        hipLaunchKernelGGL(wf_trace<true>, dim3(traceBlocks), dim3(256), lds, s, K, Q, round);
        hipLaunchKernelGGL(wf_shade<true>, dim3(shadeBlocks), dim3(256), 0, s, K, Q, round);
    } else {
        hipLaunchKernelGGL(wf_trace<false>, dim3(traceBlocks), dim3(256), lds, s, K, Q, round);
        hipLaunchKernelGGL(wf_shade<false>, dim3(shadeBlocks), dim3(256), 0, s, K, Q, round);
    }
}

int wf_set_lds_limit(size_t bytes) {
    hipError_t e = hipSuccess;
    if (bytes > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wf_trace<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }
    return e == hipSuccess ? 0 : -1;
}

}  // namespace rz
