// rz_trace.h -- closest-hit query through TLAS and BLAS on gfx950.
//
// Computes exactly what FS:457-503 (traverseTLAS) / FS:419-454 (traverseBLAS)
// / FS:391-416 (hitTriangle) / FS:380-388 (intersectAABB) compute -- same
// float operations, same visiting order, same strict `t < tHit` tie rule --
// but organised for the machine:
//
//  * BLAS: the shader pushes both children and tests each child's box when it
//    is popped.  A box test is a pure function of (ray, box), so it is done
//    here when the PARENT is expanded, on one aligned 64-B DevPair fetch that
//    holds both children.  The shader pops the right child immediately after
//    pushing it (nothing can change tHit in between), so the right child is
//    continued into directly and only the left child is stacked -- and only
//    if the ray hits its box -- together with its entry distance tmin, so the
//    shader's `tmin > tHit` cull is re-evaluated against the CURRENT tHit
//    when the entry is popped, without touching the node again.
//    => one 64-B fetch per internal node instead of two 32-B node fetches per
//       child, and a stack that holds at most one entry per tree level.
//  * the BLAS stack lives in LDS, 8 B per entry, laid out [level][lane]:
//    every lane of a wave addresses its own bank column, so pushes and pops
//    are conflict-free at any mix of depths (ds_write_b64 / ds_read_b64,
//    2 x 32-lane groups).  Capacity = the deepest BLAS of the scene, known on
//    the host at upload time.
//  * TLAS: no stack at all -- the nodes are laid out in the shader's pop order with skip positions and the wave walks
//    that list with one scalar cursor (trace_closest below).
//  * leaves index triangles gathered into leaf order (DevTri, 3 x 16-B loads,
//    no index indirection); only (t, triangle id) are tracked while
//    traversing -- hit point, normal and material are pure functions of the
//    winner and are computed once at the end.
#pragma once
#include "rz_device_math.h"
#include "rz_scene_dev.h"

namespace rz {

#ifndef RZ_UNIFORM_LEAF
#define RZ_UNIFORM_LEAF 1
#endif
#ifndef RZ_ASM_WALK
#define RZ_ASM_WALK 1
#endif
#ifndef RZ_DESCEND_MIN_LANES
#define RZ_DESCEND_MIN_LANES 4   // leave the descend loop when fewer lanes than this still have an internal node (lane=sample kernel on C2: 1 -> 17.6 ms, 2 -> 17.4, 3..6 -> 17.15-17.2, 8 -> 17.3, 12 -> 17.4)
#endif

struct Tally {          // per-thread counts of the REFERENCE algorithm's memory touches
    unsigned traversals, tlas_nodes, tlas_leaf_indices, instances, blas_nodes, triangles, materials, light_fetches,
        samples;
    // units of the shading side (bench.py's work model prices them; the oracle tallies the same events):
    unsigned triangles_past_u;                                      // hitTriangle calls that pass FS:396-401 and run the second half of the test
    unsigned scatters, diffuse_scatters, hemi_draws, lit_lights;    // FS:720-761 executed | of them FS:755 | of those with a non-zero seed (binary64 acos / sin / cos evaluated) | FS:636-659 / 589-607 evaluated
#ifdef RZ_PROF          // diagnostic build only: where do the lanes of a wave spend their iterations?
    unsigned p[16];
    unsigned ps[8];             // descend steps by lane: [0] both child boxes missed or culled, [1] one entered, [2] both (one stacked); [4..6] the same for the pool's walks (pool_trace)
    int rnd;                    // which closest-hit query of its path this lane is in (0 primary, 1-2 shadow, 3.. bounces), capped at 7
    unsigned rp[8][10];         // per query round: wave-execs / lanes of [0,1] descend steps [2,3] triangle tests [4,5] instance entries [6,7] uniform-pair steps [8,9] queries
    unsigned long long rt[8];   // per query round: wave cycles inside trace_closest (lane 0's clock)
    unsigned long long t[20];   // ([16] the end-of-claim section of a compacting claim, [17] slot_sums inside pool_process, [18] the claim-end sums alone) ([4] / [9]: phase 1 / pool rounds of a compacting claim, [10] pool rounds, [11] paths in them)   wave cycles (s_memtime): [0] descend loops, [1] leaf phases, [2] whole BLAS walks; rz_path.h advance(): [3] sky, [4] hit bookkeeping, [5] start_light, [6] shade_light, [7] scatter, [8] of it the hemisphere direction, [9] shadow-step bookkeeping
#endif
};
#ifdef RZ_PROF
// slot 2k counts wave-level executions of a site (added by the first active lane), slot 2k+1 the active lanes
#define RZ_SITE(c, k) do { const bool first_ = __lane_id() == (unsigned)(__ffsll((long long)rz_ballot(1)) - 1); (c).p[2 * (k) + 1] += 1u; if (first_) (c).p[2 * (k)] += 1u; \
        constexpr int rs_ = (k) == 3 ? 0 : (k) == 2 ? 1 : (k) == 5 ? 2 : (k) == 7 ? 3 : (k) == 6 ? 4 : -1; \
        if (rs_ >= 0) { (c).rp[(c).rnd & 7][2 * (rs_ < 0 ? 0 : rs_) + 1] += 1u; if (first_) (c).rp[(c).rnd & 7][2 * (rs_ < 0 ? 0 : rs_)] += 1u; } } while (0)
#else
#define RZ_SITE(c, k) do { } while (0)
#endif

// "These registers are needed now": stops hipcc from splitting a record load and sinking part of it into a later
// branch (seen in the ISA: the left child's ref and each triangle's v0 were re-fetched by a dependent load).
#define RZ_KEEP4(q) asm volatile("" : "+v"((q).x), "+v"((q).y), "+v"((q).z), "+v"((q).w))

// Wave-uniform record fetch through the scalar cache.  In the one-lane-per-sample kernel the 64 lanes of a wave are
// samples of ONE pixel: primary and shadow rays are near-identical, so most of the time every lane wants the same
// BVH node.  A vector load of it costs the texture-address unit 16 cycles per dwordx4 (64 lanes x 16 B at 64 B/clk)
// however many lanes share the address -- measured TA busy 86 % -- while one s_load_dwordx16 fetches the whole
// 64-B record into SGPRs and leaves the vector memory pipe alone.  The data is read-only for the kernel's lifetime,
// which is what the (non-coherent) scalar cache needs.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 sload16(const void* p) {
    f32x16 r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
// ... with the byte offset in a scalar register (unsigned, < 4 GiB): one s_lshl instead of a 64-bit shift and add
__device__ __forceinline__ f32x16 sload16_off(const void* base, unsigned byteOffset) {
    f32x16 r;
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(byteOffset) : "memory");
    return r;
}
// ... and the narrower forms
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x8 sload8(const void* p) {
    f32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
__device__ __forceinline__ f32x4s sload4(const void* p) {
    f32x4s r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
// a 48-B record (a triangle, a 3x4 matrix, an instance's root box and bases) as dwordx8 + dwordx4 behind ONE wait (round 5: issued
// as sload8 then sload4, each with its own wait, the second fetch's latency came on top of the first's: C2 -0.9 %, c2close -1.6 %,
// RayZen's scene at 64 spp -1.5 %; profiles/r05_regs/)
__device__ __forceinline__ void sload12(const void* p, f32x8& a, f32x4s& b) {
    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(p) : "memory");
}
__device__ __forceinline__ int sload1(const void* p) {
    int r;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {      // tells the compiler that p is the same in every lane
    return reinterpret_cast<const T*>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)p >> 32)) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p));
}

#ifndef RZ_SCALAR_UNIFORM
#define RZ_SCALAR_UNIFORM 1
#endif

struct HitRec {
    float t;
    v3 p;       // world-space hit point
    v3 n;       // world-space geometric normal (unit)
    int mat;
    int inst;
};

// Number of lanes for which p holds, compared on the scalar unit (written with __popcll the compiler widened the count
// to 64 bits and compared it with a VECTOR instruction, v_cmp_gt_u64 on an SGPR pair, plus two mask operations).
__device__ __forceinline__ int mask_count(unsigned long long m) {
    int n;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n) : "s"(m) : "scc");
    return n;
}
__device__ __forceinline__ int wave_count(bool p) { return mask_count(rz_ballot(p)); }
// The lanes of a mask, as a predicate: `if (in_mask(m))` is s_and_saveexec with m itself -- the loops below compute the
// mask of the NEXT trip at the end of a trip (their exit test needs it) and enter the body by it, instead of evaluating
// the same comparison a second time at the head (a v_cmp is a 4-cycle instruction here).
__device__ __forceinline__ bool in_mask(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// FS:380-388.  Returns the hit flag; tmin as the shader computes it.
__device__ __forceinline__ bool slab(v3 o, v3 inv, float bx0, float by0, float bz0, float bx1, float by1, float bz1,
                                     float& tmin) {
    float t0x = (bx0 - o.x) * inv.x, t0y = (by0 - o.y) * inv.y, t0z = (bz0 - o.z) * inv.z;
    float t1x = (bx1 - o.x) * inv.x, t1y = (by1 - o.y) * inv.y, t1z = (bz1 - o.z) * inv.z;
    float sx = fmin_(t0x, t1x), sy = fmin_(t0y, t1y), sz = fmin_(t0z, t1z);
    float gx = fmax_(t0x, t1x), gy = fmax_(t0y, t1y), gz = fmax_(t0z, t1z);
    tmin = fmax_(fmax_(sx, sy), sz);
    float tmax = fmin_(fmin_(gx, gy), gz);
    return tmax >= fmax_(tmin, 0.0f);
}

// The same test for the two children of a DevPair, with the subtractions and multiplications of an axis' two planes
// issued as packed FP32 (v_pk_add_f32 with the origin negated, v_pk_mul_f32): the identical IEEE operations on the
// identical operands, two per issue slot.  The origin and the reciprocal direction are broadcast to both halves by
// op_sel, so they stay in the registers they already occupy.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct RayPk { f32x2 oxy, ozz, ixy, izz; };          // (o.x, o.y) (o.z, o.z) (inv.x, inv.y) (inv.z, inv.z)
__device__ __forceinline__ RayPk make_raypk(v3 o, v3 inv) {
    RayPk r;
    r.oxy = f32x2{o.x, o.y}; r.ozz = f32x2{o.z, o.z}; r.ixy = f32x2{inv.x, inv.y}; r.izz = f32x2{inv.z, inv.z};
    return r;
}
// (written as 2-vector IR, not inline asm: the compiler folds the broadcast into op_sel itself, and it then knows the
// products are canonical, so the min/max that follow need no v_max(x, x) quieting first)
#define RZ_PLANES(out, box, o2, i2, SEL)                                                                             \
    do {                                                                                                             \
        const f32x2 ob_ = __builtin_shufflevector(o2, o2, SEL, SEL), ib_ = __builtin_shufflevector(i2, i2, SEL, SEL);\
        out = ((box) - ob_) * ib_;                                                                                   \
    } while (0)
// OCT < 0: the shader's per-axis min / max (FS:384-385).  OCT in 0..7: the wave's rays all point into octant OCT
// (bit k set: direction component k negative) and every reciprocal is finite and nonzero; then for each axis
// (plane_min - o) * inv <= (plane_max - o) * inv when inv > 0 and >= when inv < 0 -- subtraction, multiplication and
// their roundings are monotone -- so min / max of the two products ARE one and the other product: the six v_min / v_max
// per box are compile-time register choices.  (With a zero direction component the products can be 0 * inf = NaN,
// which minNum / maxNum skip: those rays keep the generic form.)  v_min_f32 / v_max_f32 issue at about 3.4 cycles per
// wave on gfx950 against 2 for v_mul / v_add (profiles/r02_valu_issue): 12 of them per pair of boxes were a quarter of
// a descend step's vector-ALU time.
template <int OCT>
__device__ __forceinline__ bool slab_finish(f32x2 tx, f32x2 ty, f32x2 tz, float& tmin) {
    float sx, sy, sz, gx, gy, gz;
    if constexpr (OCT < 0) {
        sx = fmin_(tx.x, tx.y); sy = fmin_(ty.x, ty.y); sz = fmin_(tz.x, tz.y);
        gx = fmax_(tx.x, tx.y); gy = fmax_(ty.x, ty.y); gz = fmax_(tz.x, tz.y);
    } else {
        sx = (OCT & 1) ? tx.y : tx.x; gx = (OCT & 1) ? tx.x : tx.y;
        sy = (OCT & 2) ? ty.y : ty.x; gy = (OCT & 2) ? ty.x : ty.y;
        sz = (OCT & 4) ? tz.y : tz.x; gz = (OCT & 4) ? tz.x : tz.y;
    }
    tmin = fmax_(fmax_(sx, sy), sz);
    const float tmax = fmin_(fmin_(gx, gy), gz);
    return tmax >= fmax_(tmin, 0.0f);
}
#define RZ_SLAB_PAIR(OCT, R, lx, ly, lz, rx, ry, rz, hl, tl, hr, tr)                                                  \
    do {                                                                                                             \
        f32x2 ax_, ay_, az_, bx_, by_, bz_;                                                                          \
        RZ_PLANES(ax_, lx, (R).oxy, (R).ixy, 0); RZ_PLANES(ay_, ly, (R).oxy, (R).ixy, 1);                      \
        RZ_PLANES(az_, lz, (R).ozz, (R).izz, 0);                                                                  \
        RZ_PLANES(bx_, rx, (R).oxy, (R).ixy, 0); RZ_PLANES(by_, ry, (R).oxy, (R).ixy, 1);                      \
        RZ_PLANES(bz_, rz, (R).ozz, (R).izz, 0);                                                                  \
        hl = slab_finish<OCT>(ax_, ay_, az_, tl);                                                                    \
        hr = slab_finish<OCT>(bx_, by_, bz_, tr);                                                                         \
    } while (0)

// FS:391-416 without the outputs that are pure functions of (ray, t, triangle).
// Evaluated without early exits: the accept decision is the conjunction of the
// shader's tests in order, so values computed past a failed test are never used.
__device__ __forceinline__ bool moller_trumbore(v3 o, v3 d, v3 v0, v3 e1, v3 e2, float& t, bool& pastU) {
    v3 h = cross(d, e2);
    float a = dot(e1, h);
    // FS:398.  A lane whose |a| < 0.0001 fails the shader's first test whatever f is, so only |a| <= 2^126 has to hold
    // for the short reciprocal (NaN and infinities fail the comparison and take the division).
    float f;
    if (rz_ballot(!(__builtin_fabsf(a) <= 0x1p126f)) == 0ull) f = rcp_mid(a);
    else f = rcp_ieee(a);
    v3 s = o - v0;
    float u = f * dot(s, h);
    // (every comparison is evaluated for every lane and the results are combined as lane masks: written as
    //  `ok = ok && ...` hipcc wrapped each further pair of compares in its own exec-mask region, four scalar
    //  instructions to spare some lanes two vector ones)
    const bool aSmall = __builtin_fabsf(a) < 0.0001f;
    const bool uLow = u < 0.0f, uHigh = u > 1.0f;
    const bool ok1 = !aSmall && !uLow && !uHigh;
    pastU = ok1;            // (for the counting instantiations' tally)
#ifndef RZ_NO_TRI_EARLY_OUT
    // Wave-level early out after the shader's second test: the lanes of a wave are mostly samples of one pixel testing
    // the same triangle with near-identical rays, so they tend to fail together -- and then the second cross product,
    // two dot products and the remaining comparisons (about 30 of the test's 75 instructions) are skipped.  A lane's
    // own result is unchanged: values computed past a failed test were never used.
    if (rz_ballot(ok1) == 0ull) return false;
#endif
    v3 q = cross(s, e1);
    float v = f * dot(d, q);
    t = f * dot(e2, q);
    const bool vLow = v < 0.0f, uvHigh = u + v > 1.0f, tOk = t > 0.0001f;
    return ok1 && !vLow && !uvHigh && tOk;
}

// 3x4 packed column-major transforms of DevInstance
__device__ __forceinline__ v3 x34_point(const float* m, v3 v) {
    return mk3(((m[0] * v.x + m[3] * v.y) + m[6] * v.z) + m[9],
               ((m[1] * v.x + m[4] * v.y) + m[7] * v.z) + m[10],
               ((m[2] * v.x + m[5] * v.y) + m[8] * v.z) + m[11]);
}
__device__ __forceinline__ v3 x34_dir(const float* m, v3 v) {
    return mk3((m[0] * v.x + m[3] * v.y) + m[6] * v.z,
               (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
               (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ v3 x34_normal(const float* m, v3 v) {   // mat3(transpose(m)) * v
    return mk3((m[0] * v.x + m[1] * v.y) + m[2] * v.z,
               (m[3] * v.x + m[4] * v.y) + m[5] * v.z,
               (m[6] * v.x + m[7] * v.y) + m[8] * v.z);
}

// Pop one stack entry.  The shader's cull `tmin > tHit` (FS:430) is evaluated here, against the CURRENT tLoc; a
// culled entry becomes an empty leaf (enc -1 = ~((0 << 4) | 0)): the next leaf phase tests none of its zero
// triangles and pops again.  Same visits in the same order, without an inner pop-until-unculled loop.
// The stack: entries [0, W) live in LDS (this lane's column, 64 apart), deeper ones -- when the scene's deepest BLAS
// does not fit the LDS budget that keeps 16 waves on a CU -- in a global-memory column of the same shape.  A ray's
// stack holds one entry per level at which BOTH children were hit, typically well under half the tree depth, so the
// overflow is touched by few lanes; W == capacity and ovf == nullptr when everything fits.
// OVF is a compile-time property of the launch: the window test costs the common case (everything in LDS) 6 % if left in.
template <bool OVF>
struct BlasStackT { uint2* lds; uint2* ovf; int W; };
template <bool OVF>
__device__ __forceinline__ void push_entry(const BlasStackT<OVF>& S, int& sp, uint2 e) {
    if (!OVF || sp < S.W) {
        S.lds[sp * 64] = e;
    } else {
        uint2* q = S.ovf + (sp - S.W) * 64;
        asm volatile("" : "+v"(q));      // opaque to the optimiser (hipcc crashed merging this store with the LDS one)
        *q = e;
    }
    ++sp;
}
template <bool OVF>
__device__ __forceinline__ bool pop_entry(const BlasStackT<OVF>& S, int& sp, float tLoc, int& cur) {
    if (sp <= 0) return false;
    --sp;
    uint2 e;
    if (!OVF || sp < S.W) {
        e = S.lds[sp * 64];
    } else {
        const uint2* q = S.ovf + (sp - S.W) * 64;
        asm volatile("" : "+v"(q));
        e = *q;
    }
    asm volatile("" : "+v"(e.x), "+v"(e.y));     // one ds_read_b64
    cur = (__uint_as_float(e.y) > tLoc) ? -1 : (int)e.x;
    return true;
}

// One instance's BLAS (FS:419-454) in the instance's local space.
// bstk: this lane's stack (LDS window + optional overflow).
// Returns the winning triangle (absolute DevTri index) or -1; tLoc = its t.
// OCT: -1 generic, 0..7 the wave's common direction octant (slab_finish).
// MI ("many instances"): the lanes of the wave walk DIFFERENT instances at the same time -- `pairs` / `tris` are then the
// scene's arrays, the lane's instance contributes its bases pbase / tbase, and a node number is only meaningful together
// with its base.  Such waves hold incoherent rays (the late bounces), so the wave-uniform scalar fetch is not even tried.
template <bool COUNT, bool OVF, int OCT, bool MI>
__device__ __forceinline__ int blas_walk(const DevPair* __restrict__ pairs, const DevTri* __restrict__ tris, v3 lo, v3 ld, v3 inv,
                                         bool go, int cur, int pbase, int tbase, float& tLocOut, const BlasStackT<OVF>& bstk, Tally& c) {
    float tLoc = 1e30f;
    int best = -1;
    int sp = 0;
    const RayPk RP = make_raypk(lo, inv);
    // Lane state: `cur` -- >= 0 an internal node (its DevPair), < 0 a leaf (~cur = first << 4 | count; -1 is the empty
    // leaf that a culled stack entry turns into) -- and the stack height `sp`.  A lane is finished when cur == -1 and
    // sp == 0 (an empty leaf with nothing left to pop): no separate flag.
    // The loops below are WAVE-UNIFORM loops with predicated bodies (their trip conditions are ballots, forced into
    // scalar registers), not per-lane `while (go && cur >= 0)` loops: for those hipcc keeps a mask of the lanes that
    // have left each loop level and spent more scalar instructions on exec-mask bookkeeping (~55 per descend step) than
    // vector instructions on the two box tests (~50) -- and the kernel is bound by instruction issue, scalar
    // instructions included.
#ifdef RZ_PROF
    const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#endif
    if (!go) cur = -1;
    // (the bound is a backstop, never reached: a BLAS of n nodes is walked in fewer than 2n rounds and the host has
    //  checked that the node array is a tree -- but a wave that can spin for ever takes the whole device with it)
    // (every loop here has ONE exit, at its end: with a second `break` hipcc turned the wave-uniform exit conditions into
    //  lane masks and tested those -- four to six scalar instructions per trip instead of a compare and a branch)
    unsigned round = 0;
    bool alive;
    do {
        // "while-while": walk internal nodes; a lane that reaches a leaf parks there (its own sequence of operations is
        // unchanged) until the lanes of the wave still descending are few, then the parked lanes test their leaves
        // together.  Without this the wave ran the triangle tests for ~7 of its 64 lanes at a time.
#ifdef RZ_PROF
        const unsigned long long td0_ = __builtin_amdgcn_s_memtime();
#endif
        // (the lane-count test sits at the END of the body: lanes at internal nodes always advance at least one step per
        //  round of the outer loop, or one to three stragglers with nobody at a leaf would never move again)
        unsigned long long actMask = rz_ballot(cur >= 0);
        bool more = actMask != 0ull;
        while (more) {
            if (in_mask(actMask)) {
                RZ_SITE(c, 3);
                const DevPair* pp = pairs + (MI ? pbase + cur : cur);
                if (COUNT) c.blas_nodes += 2;        // the shader pushes, and later pops, both children
                // The shader pushes left, then right, and pops right at once (FS:449-450, 428-430).
                //  * right hit and not culled: continue into it; a hit left child waits on the stack with its entry distance;
                //  * else the next pop IS the left child just pushed: enter it directly (same cull against the same tLoc)
                //    -- no LDS round trip;
                //  * neither box hit: pop an older entry.
                // (a lambda run at the end of EITHER fetch path, not code after their join: joined, the two hit flags
                //  were lane masks merged by six scalar instructions per step)
                auto step = [&](bool hl, float tl, bool hr, float tr, int lenc, int renc) {
                    const bool takeR = hr && !(tr > tLoc);
#ifdef RZ_PROF
                    { const bool takeL_ = hl && !(tl > tLoc); c.ps[(takeR ? 1 : 0) + (takeL_ ? 1 : 0)] += 1u; }
#endif
                    if (hl && takeR) push_entry(bstk, sp, make_uint2((unsigned)lenc, __float_as_uint(tl)));
                    int next = takeR ? renc : ((tl > tLoc) ? -1 : lenc);
                    if (!hl && !takeR) {
                        if (!pop_entry(bstk, sp, tLoc, next)) next = -1;
                    }
                    cur = next;
                };
#if RZ_SCALAR_UNIFORM
                // (`pairs` is wave-uniform -- it comes from the instance record, a scalar fetch -- so comparing the 32-bit
                //  node numbers does what comparing the 64-bit addresses did, and the address is computed on the scalar unit)
                const int ucur = MI ? 0 : __builtin_amdgcn_readfirstlane(cur);
                if (!MI && rz_ballot(cur != ucur) == 0ull) {     // every active lane wants the same pair: one scalar fetch,
                    // box values consumed straight from SGPRs.  And while the lanes also AGREE on where to go -- the samples of a
                    // pixel nearly always do -- the walk stays on the scalar unit: the cursor is an SGPR, the decision two ballots
                    // and a compare, and none of the per-lane selects, exec masks and the uniformity test of the general step are
                    // executed (they were half of a step's 61 instructions).  Same pushes, same culls, same order per lane.
#if RZ_ASM_WALK && !defined(RZ_PROF)
                    if constexpr (!COUNT && !OVF && OCT >= 0) {
                        // The same loop, hand-written.  hipcc keeps the loop's wave-uniform flags as 64-bit lane
                        // masks and spends 19 scalar and branch instructions per step beside the 21 vector ones; this body spends 11:
                        // shift, fetch, wait | the two slab tests in their octant form | mR, "everybody right?" -> (a left child to
                        // stack?) cursor, loop | else "nobody right, everybody left?" -> cursor, loop | else out.  Same fetches,
                        // same arithmetic on the same operands, same pushes per lane, same culls against the lane's own tLoc.
                        // Fixed registers (the instruction set has no way to name half of an operand): s[80:95] the pair,
                        // v[6:17] the six packed plane products, v[18:19] the stack entry (left child, its entry distance), v20-v25.
                        int u = ucur, lenc, renc, mixedI, off;
                        const unsigned long long ex = rz_ballot(true);
                        unsigned long long mHl, mHr, mRr, sv;
                        float tl, tr;
                        const f32x2 ox2 = {lo.x, lo.x}, oy2 = {lo.y, lo.y}, oz2 = {lo.z, lo.z};
                        const f32x2 ix2 = {inv.x, inv.x}, iy2 = {inv.y, inv.y}, iz2 = {inv.z, inv.z};
                        const unsigned ldsCol = (unsigned)(unsigned long long)bstk.lds;      // (a flat LDS address: its low half is the LDS offset)
#define RZ_WALK_ASM(NLX, FLX, NLY, FLY, NLZ, FLZ, NRX, FRX, NRY, FRY, NRZ, FRZ)                                                        \
                        asm volatile(                                                                                                  \
                            "1:\n\t"                                                                                                   \
                            "s_lshl_b32 %[off], %[u], 6\n\t"                                                                           \
                            "s_load_dwordx16 s[80:95], %[base], %[off]\n\t"                                                            \
                            "s_waitcnt lgkmcnt(0)\n\t"                                                                                 \
                            "v_pk_add_f32 v[6:7], s[80:81], %[ox] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
                            "v_pk_add_f32 v[8:9], s[82:83], %[oy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                       \
                            "v_pk_add_f32 v[10:11], s[84:85], %[oz] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                     \
                            "v_pk_mul_f32 v[6:7], %[ix], v[6:7]\n\t"                                                                   \
                            "v_pk_mul_f32 v[8:9], %[iy], v[8:9]\n\t"                                                                   \
                            "v_pk_mul_f32 v[10:11], %[iz], v[10:11]\n\t"                                                               \
                            "v_pk_add_f32 v[12:13], s[86:87], %[ox] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                     \
                            "v_pk_add_f32 v[14:15], s[88:89], %[oy] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                     \
                            "v_pk_add_f32 v[16:17], s[90:91], %[oz] neg_lo:[0,1] neg_hi:[0,1]\n\t"                                     \
                            "v_max3_f32 v19, " NLX ", " NLY ", " NLZ "\n\t"                                                            \
                            "v_pk_mul_f32 v[12:13], %[ix], v[12:13]\n\t"                                                               \
                            "v_pk_mul_f32 v[14:15], %[iy], v[14:15]\n\t"                                                               \
                            "v_pk_mul_f32 v[16:17], %[iz], v[16:17]\n\t"                                                               \
                            "v_min3_f32 v20, " FLX ", " FLY ", " FLZ "\n\t"                                                            \
                            "v_max_f32_e32 v21, 0, v19\n\t"                                                                            \
                            "v_cmp_ge_f32_e64 %[mhl], v20, v21\n\t"                                                                    \
                            "v_max3_f32 v22, " NRX ", " NRY ", " NRZ "\n\t"                                                            \
                            "v_min3_f32 v23, " FRX ", " FRY ", " FRZ "\n\t"                                                            \
                            "v_max_f32_e32 v24, 0, v22\n\t"                                                                            \
                            "v_cmp_ge_f32_e64 %[mhr], v23, v24\n\t"                                                                    \
                            "v_cmp_gt_f32_e32 vcc, v22, %[tloc]\n\t"                                                                   \
                            "s_andn2_b64 %[mr], %[mhr], vcc\n\t"                                                                       \
                            "s_cmp_eq_u64 %[mr], %[ex]\n\t"                                                                            \
                            "s_cbranch_scc0 3f\n\t"                                                                                    \
                            "s_cmp_eq_u64 %[mhl], 0\n\t"                                                                               \
                            "s_cbranch_scc1 2f\n\t"                                                                                    \
                            "s_and_saveexec_b64 %[sv], %[mhl]\n\t"                                                                     \
                            "v_lshl_add_u32 v25, %[sp], 9, %[lds]\n\t"                                                                 \
                            "v_add_u32_e32 %[sp], 1, %[sp]\n\t"                                                                        \
                            "v_mov_b32_e32 v18, s92\n\t"                                                                               \
                            "ds_write_b64 v25, v[18:19]\n\t"                                                                           \
                            "s_mov_b64 exec, %[sv]\n"                                                                                  \
                            "2:\n\t"                                                                                                   \
                            "s_mov_b32 %[u], s93\n\t"                                                                                  \
                            "s_cmp_gt_i32 s93, -1\n\t"                                                                                 \
                            "s_cbranch_scc1 1b\n\t"                                                                                    \
                            "s_mov_b32 %[mixed], 0\n\t"                                                                                \
                            "s_branch 9f\n"                                                                                            \
                            "3:\n\t"                                                                                                   \
                            "s_mov_b32 %[mixed], 1\n\t"                                                                                \
                            "s_cmp_lg_u64 %[mr], 0\n\t"                                                                                \
                            "s_cbranch_scc1 9f\n\t"                                                                                    \
                            "v_cmp_gt_f32_e32 vcc, v19, %[tloc]\n\t"                                                                   \
                            "s_andn2_b64 %[mr], %[mhl], vcc\n\t"                                                                       \
                            "s_cmp_eq_u64 %[mr], %[ex]\n\t"                                                                            \
                            "s_cbranch_scc0 9f\n\t"                                                                                    \
                            "s_mov_b32 %[u], s92\n\t"                                                                                  \
                            "s_cmp_gt_i32 s92, -1\n\t"                                                                                 \
                            "s_cbranch_scc1 1b\n\t"                                                                                    \
                            "s_mov_b32 %[mixed], 0\n"                                                                                  \
                            "9:\n\t"                                                                                                   \
                            "v_mov_b32_e32 %[tl], v19\n\t"                                                                             \
                            "v_mov_b32_e32 %[tr], v22\n\t"                                                                             \
                            "s_mov_b32 %[lenc], s92\n\t"                                                                               \
                            "s_mov_b32 %[renc], s93"                                                                                   \
                            : [u] "+s"(u), [sp] "+v"(sp), [off] "=&s"(off), [mhl] "=&s"(mHl), [mhr] "=&s"(mHr), [mr] "=&s"(mRr),         \
                              [sv] "=&s"(sv), [mixed] "=&s"(mixedI), [tl] "=&v"(tl), [tr] "=&v"(tr), [lenc] "=&s"(lenc), [renc] "=&s"(renc) \
                            : [base] "s"(pairs), [ox] "v"(ox2), [oy] "v"(oy2), [oz] "v"(oz2), [ix] "v"(ix2), [iy] "v"(iy2), [iz] "v"(iz2), \
                              [tloc] "v"(tLoc), [ex] "s"(ex), [lds] "v"(ldsCol)                                                         \
                            : "memory", "vcc", "scc", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", \
                              "s92", "s93", "s94", "s95", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", \
                              "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25")
                        if constexpr (OCT == 0) { RZ_WALK_ASM("v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17"); }
                        else if constexpr (OCT == 1) { RZ_WALK_ASM("v7", "v6", "v8", "v9", "v10", "v11", "v13", "v12", "v14", "v15", "v16", "v17"); }
                        else if constexpr (OCT == 2) { RZ_WALK_ASM("v6", "v7", "v9", "v8", "v10", "v11", "v12", "v13", "v15", "v14", "v16", "v17"); }
                        else if constexpr (OCT == 3) { RZ_WALK_ASM("v7", "v6", "v9", "v8", "v10", "v11", "v13", "v12", "v15", "v14", "v16", "v17"); }
                        else if constexpr (OCT == 4) { RZ_WALK_ASM("v6", "v7", "v8", "v9", "v11", "v10", "v12", "v13", "v14", "v15", "v17", "v16"); }
                        else if constexpr (OCT == 5) { RZ_WALK_ASM("v7", "v6", "v8", "v9", "v11", "v10", "v13", "v12", "v14", "v15", "v17", "v16"); }
                        else if constexpr (OCT == 6) { RZ_WALK_ASM("v6", "v7", "v9", "v8", "v11", "v10", "v12", "v13", "v15", "v14", "v17", "v16"); }
                        else if constexpr (OCT == 7) { RZ_WALK_ASM("v7", "v6", "v9", "v8", "v11", "v10", "v13", "v12", "v15", "v14", "v17", "v16"); }
#undef RZ_WALK_ASM
                        if (mixedI != 0) step(in_mask(mHl), tl, in_mask(mHr), tr, lenc, renc);      // the lanes part ways here: the general step, on the pair just tested
                        else cur = u;
                    } else
#endif
                    {
                    int u = ucur, lenc, renc;
                    const unsigned long long ex = rz_ballot(true);
                    float tl, tr;
                    bool hl, hr, mixed;
                    int steps = 0;
                    do {
                        RZ_SITE(c, 7);
                        if (COUNT && steps > 0) c.blas_nodes += 2;
                        ++steps;
                        const f32x16 q = sload16_off(pairs, (unsigned)u << 6);     // (a BLAS's pairs span < 4 GiB: rz_context.hip, finalize)
                        const f32x2 lx = {q[0], q[1]}, ly = {q[2], q[3]}, lz = {q[4], q[5]};
                        const f32x2 rx = {q[6], q[7]}, ry = {q[8], q[9]}, rz = {q[10], q[11]};
                        RZ_SLAB_PAIR(OCT, RP, lx, ly, lz, rx, ry, rz, hl, tl, hr, tr);
                        lenc = __float_as_int(q[12]); renc = __float_as_int(q[13]);
                        // (lane masks combined as integers: a ballot of `hr && !(tr > tLoc)` is materialised per lane and compared again)
                        const unsigned long long mHl = rz_ballot(hl);
                        const unsigned long long mR = rz_ballot(hr) & ~rz_ballot(tr > tLoc);
                        mixed = true;
                        if (mR == ex) {                 // everybody goes on into the right child; a hit left child waits on the lane's stack
                            if (mHl != 0ull) {
                                if (in_mask(mHl)) push_entry(bstk, sp, make_uint2((unsigned)lenc, __float_as_uint(tl)));
                            }
                            u = renc; mixed = false;
                        } else if (mR == 0ull && (mHl & ~rz_ballot(tl > tLoc)) == ex) {     // nobody does, and everybody enters the left child directly
                            u = lenc; mixed = false;
                        }
                    } while (!mixed && u >= 0);
                    if (mixed) step(hl, tl, hr, tr, lenc, renc);    // the lanes part ways here: the general step, on the pair just tested
                    else cur = u;
                    }
                } else
#endif
                {
                    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(pp);
                    float4 p0 = p4[0], p1 = p4[1], p2 = p4[2], p3 = p4[3];
#ifndef RZ_EXP_NOKEEP
                    RZ_KEEP4(p0); RZ_KEEP4(p3);
#endif
                    const f32x2 lx = {p0.x, p0.y}, ly = {p0.z, p0.w}, lz = {p1.x, p1.y};
                    const f32x2 rx = {p1.z, p1.w}, ry = {p2.x, p2.y}, rz = {p2.z, p2.w};
                    float tl, tr;
                    bool hl, hr;
                    RZ_SLAB_PAIR(OCT, RP, lx, ly, lz, rx, ry, rz, hl, tl, hr, tr);
                    step(hl, tl, hr, tr, __float_as_int(p3.x), __float_as_int(p3.y));
                }
            }
            actMask = rz_ballot(cur >= 0);
            more = mask_count(actMask) >= RZ_DESCEND_MIN_LANES;
        }
#ifdef RZ_PROF
        const unsigned long long td1_ = __builtin_amdgcn_s_memtime();
        c.t[0] += td1_ - td0_;
#endif
        alive = rz_ballot((cur != -1) || (sp > 0)) != 0ull;
        if (alive) {
            // leaves: <= 4 triangles each, contiguous in leaf order, tested in order (count 0: a culled stack entry, or a
            // finished lane); a wave-uniform loop over the triangle slot, lanes with fewer triangles sit out
            const bool leaf = cur < 0;
            const int v = ~cur;
            const int first = (v >> 4) + (MI ? tbase : 0), count = leaf ? (v & 15) : 0;
            if (COUNT) c.triangles += (unsigned)count;
            if (leaf) RZ_SITE(c, 1);
            int i = 0;
            unsigned long long triMask = rz_ballot(0 < count);
            bool any = triMask != 0ull;
#if RZ_UNIFORM_LEAF
            // Every lane that has triangles to test stands at the SAME leaf (the samples of a pixel mostly do: they walked there
            // together): its triangles come through the scalar cache, 48 B once per wave instead of three 16-B loads per lane,
            // and the test takes them from scalar registers.  The same tests in the same order on the same values.
            if constexpr (!MI) {
                if (any) {
                    const int ucur = __builtin_amdgcn_readlane(cur, (int)__builtin_ctzll(triMask));
                    if (rz_ballot(0 < count && cur != ucur) == 0ull) {
                        const int uv = ~ucur, ufirst = uv >> 4, ucount = uv & 15;
                        if (in_mask(triMask)) {
                            for (int k = 0; k < ucount; ++k) {
                                RZ_SITE(c, 2);
                                const float* __restrict__ tf = reinterpret_cast<const float*>(tris + ufirst + k);
                                f32x8 a;
                                f32x4s b;
                                sload12(tf, a, b);
                                float t;
                                bool pastU;
                                const bool hit = moller_trumbore(lo, ld, mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(a[6], a[7], b[0]), t, pastU);
                                if (COUNT && pastU) c.triangles_past_u += 1;
                                if (hit && t < tLoc) { tLoc = t; best = ufirst + k; }
                            }
                        }
                        any = false;
                    }
                }
            }
#endif
            while (any) {
                if (in_mask(triMask)) {
                    RZ_SITE(c, 2);
                    const float4* __restrict__ tp = reinterpret_cast<const float4*>(tris + first + i);
                    float4 a = tp[0], b = tp[1], cc = tp[2];
                    RZ_KEEP4(a);
                    float t;
                    bool pastU;
                    const bool hit = moller_trumbore(lo, ld, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, cc.x), t, pastU);
                    if (COUNT && pastU) c.triangles_past_u += 1;
                    if (hit && t < tLoc) { tLoc = t; best = first + i; }
                }
                ++i;
                triMask = rz_ballot(i < count);
                any = triMask != 0ull;
            }
            if (leaf) {
                if (!pop_entry(bstk, sp, tLoc, cur)) cur = -1;
            }
        }
#ifdef RZ_PROF
        c.t[1] += __builtin_amdgcn_s_memtime() - td1_;
#endif
        RZ_SITE(c, 0);
    } while (alive && ++round < (1u << 24));
#ifdef RZ_PROF
    c.t[2] += __builtin_amdgcn_s_memtime() - tw0_;
#endif
    tLocOut = tLoc;
    return best;
}


#ifndef RZ_OCTANT_SLAB
#define RZ_OCTANT_SLAB 1
#endif
// The lanes of the wave enter ONE instance, `I` (wave-uniform: its record comes through the scalar cache).
template <bool COUNT, bool OVF>
__device__ __forceinline__ int traverse_blas(const KParams& K, const DevInstance* __restrict__ I, v3 lo, v3 ld,
                                             float& tLocOut, const BlasStackT<OVF>& bstk, Tally& c) {
    const v3 inv = rcp3(ld);
    if (COUNT) c.blas_nodes += 1;            // the shader pops the root
    // (I is wave-uniform: the instance's root box, root reference and bases come through the scalar cache)
    f32x8 r0;                                // rootMin[3], rootEnc, rootMax[3], pairBase
    f32x4s r1;                               // triBase, flags
    sload12(I->rootMin, r0, r1);
    float tminRoot;
    bool go = slab(lo, inv, r0[0], r0[1], r0[2], r0[4], r0[5], r0[6], tminRoot);
    const int iflags = __float_as_int(r1[1]);
    go = go && !(tminRoot > 1e30f) && !(iflags & 1);
    const int cur = __float_as_int(r0[3]);
    const int triBase = __float_as_int(r1[0]);
    const DevPair* __restrict__ pairs = K.pairs + __float_as_int(r0[7]);
    const DevTri* __restrict__ tris = K.tris + triBase;
    int best;
#if RZ_OCTANT_SLAB
    if constexpr (!COUNT) {
        // the lanes that will walk this BLAS: do they share a direction octant, with every reciprocal finite and nonzero?
        const float big = __builtin_huge_valf();
        const bool fin = (__builtin_fabsf(inv.x) < big) && (__builtin_fabsf(inv.y) < big) && (__builtin_fabsf(inv.z) < big) &&
                         inv.x != 0.0f && inv.y != 0.0f && inv.z != 0.0f;
        const int oct = (inv.x < 0.0f ? 1 : 0) | (inv.y < 0.0f ? 2 : 0) | (inv.z < 0.0f ? 4 : 0);
        const unsigned long long walkers = rz_ballot(go);
        int uoct = -1;
        if (walkers != 0ull && K.regularBoxes != 0) {       // (an inverted or NaN child box would be hit by the shader's min / max and missed by the octant's choice)
            const int first = __builtin_amdgcn_readlane(oct, (int)__builtin_ctzll(walkers));
            if (rz_ballot(go && (!fin || oct != first)) == 0ull) uoct = first;
        }
        switch (uoct) {
            case 0: best = blas_walk<COUNT, OVF, 0, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 1: best = blas_walk<COUNT, OVF, 1, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 2: best = blas_walk<COUNT, OVF, 2, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 3: best = blas_walk<COUNT, OVF, 3, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 4: best = blas_walk<COUNT, OVF, 4, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 5: best = blas_walk<COUNT, OVF, 5, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 6: best = blas_walk<COUNT, OVF, 6, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            case 7: best = blas_walk<COUNT, OVF, 7, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
            default: best = blas_walk<COUNT, OVF, -1, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c); break;
        }
    } else
#endif
    {
        best = blas_walk<COUNT, OVF, -1, false>(pairs, tris, lo, ld, inv, go, cur, 0, 0, tLocOut, bstk, c);
    }
    return best < 0 ? -1 : best + triBase;
}

// Every lane enters ITS OWN instance (record `inst` of K.instances, fetched per lane): one BLAS walk serves lanes that
// stand in different instances.  Per lane the same operations as traverse_blas.
template <bool COUNT, bool OVF>
__device__ __forceinline__ int traverse_blas_mi(const KParams& K, int inst, v3 lo, v3 ld, float& tLocOut, const BlasStackT<OVF>& bstk, Tally& c) {
    const v3 inv = rcp3(ld);
    if (COUNT) c.blas_nodes += 1;            // the shader pops the root
    const float4* __restrict__ I4 = reinterpret_cast<const float4*>(K.instances + inst);
    const float4 r0 = I4[6], r1 = I4[7];     // rootMin[3], rootEnc | rootMax[3], pairBase
    const int2 r2 = *reinterpret_cast<const int2*>(I4 + 8);      // triBase, flags
    float tminRoot;
    bool go = slab(lo, inv, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tminRoot);
    go = go && !(tminRoot > 1e30f) && !(r2.y & 1);
    return blas_walk<COUNT, OVF, -1, true>(K.pairs, K.tris, lo, ld, inv, go, __float_as_int(r0.w), __float_as_int(r1.w), r2.x, tLocOut, bstk, c);
}

// FS:457-503 without a stack.  The shader's loop pops nodes in an order that does not depend on the ray: depth first,
// right child before left (it pushes leftFirst, then leftFirst + 1, and pops the top).  A ray only decides which
// subtrees are SKIPPED -- box missed, or tmin > tHit at the moment of the pop.  So the TLAS is laid out once per
// upload / rebuild as the list of its nodes in that order (TlasDfs, rz_scene_dev.h), each with the position that
// follows its subtree, and a lane's whole traversal state is one integer: the position it visits next.  The WAVE walks
// the list front to back with a scalar cursor `pos`: the record at pos is fetched once through the scalar cache, the
// lanes standing at pos test its box against their own ray and their own tHit, and move on to pos + 1 (internal, box
// passed) or to the record's skip position.  Every lane that is not at pos stands at or behind pos's skip position
// (it left pos's subtree, or never entered it), so the next position anybody needs is pos + 1 if some lane descended
// and skip[pos] otherwise: no wave-wide minimum, no LDS stack, no per-lane node fetch, no uniformity test.
// All lanes that ever enter an instance do so in the same step, with the instance record in scalar registers.
// Same pops in the same order per lane, same culls against the same tHit: same result, same tallies.
// (The shader's stack[64] has no overflow guard; the oracle skips a push that would not fit.  The number of entries
//  below a node when it is popped is a property of the tree -- one per ancestor whose right subtree holds it -- so
//  the layout marks such nodes as never expanding: count 0.)
template <bool COUNT, bool OVF>
__device__ __forceinline__ bool trace_closest(const KParams& K, v3 o, v3 d, HitRec& h, const BlasStackT<OVF>& bstk, Tally& c) {
    float tHit = 1e30f;
    int bestTri = -1, bestInst = -1;
    v3 bestP = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT) c.traversals += 1;
    const v3 inv = rcp3(d);
    int idx = 0;                                    // the position this lane visits next
    const int nDfs = K.nTlasDfs;
    // (pos strictly grows -- a skip position lies behind its node, by construction on the host and on the device -- so the
    //  list is walked in at most nDfs steps; `step` is the backstop against a list that says otherwise: a wave that never
    //  leaves this loop takes the device with it)
    for (int pos = 0, step = 0; pos < nDfs && step < nDfs; ++step) {
        const f32x16 q = sload16(K.tlasDfs + pos);
        const bool at = idx == pos;
        RZ_SITE(c, 4);
        if (COUNT && at) c.tlas_nodes += 1;
        // (evaluated by every lane and combined without short-circuits: the test is straight-line code either way, and a
        //  flag defined under `at && ...` came back from its branch as a lane mask that the ballots below had to rebuild)
        float tmin;
        const bool box = slab(o, inv, q[0], q[1], q[2], q[4], q[5], q[6], tmin);
        const bool pass = at & box & !(tmin > tHit);
        const int count = __float_as_int(q[7]), skip = __float_as_int(q[8]);
        if (count > 0) {
            if (rz_ballot(pass) != 0ull) {
                const int first = __float_as_int(q[3]);
                for (int i = 0; i < count; ++i) {
                    const int instIdx = i == 0 ? __float_as_int(q[9]) : sload1(K.tlasIndices + first + i);
                    const DevInstance* __restrict__ I = K.instances + instIdx;
                    if (pass) {
                        if (COUNT) { c.tlas_leaf_indices += 1; c.instances += 1; }
                        RZ_SITE(c, 5);
                        f32x8 m0;
                        f32x4s m1;
                        sload12(I->inv, m0, m1);
                        const float mi[12] = {m0[0], m0[1], m0[2], m0[3], m0[4], m0[5], m0[6], m0[7], m1[0], m1[1], m1[2], m1[3]};
                        const v3 lo = x34_point(mi, o);
                        const v3 ld = normalize(x34_dir(mi, d));
                        float tLoc;
                        const int tri = traverse_blas<COUNT, OVF>(K, I, lo, ld, tLoc, bstk, c);
                        if (rz_ballot(tri >= 0) != 0ull) {
                            f32x8 f0;
                            f32x4s f1;
                            sload12(I->fwd, f0, f1);
                            const float mf[12] = {f0[0], f0[1], f0[2], f0[3], f0[4], f0[5], f0[6], f0[7], f1[0], f1[1], f1[2], f1[3]};
                            if (tri >= 0) {
                                const v3 localHit = lo + ld * tLoc;              // FS:410
                                const v3 worldHit = x34_point(mf, localHit);     // FS:484
                                const float tWorld = length(worldHit - o);       // FS:485
                                if (tWorld < tHit) { tHit = tWorld; bestP = worldHit; bestTri = tri; bestInst = instIdx; }
                            }
                        }
                    }
                }
            }
            if (at) idx = skip;
            pos = skip;
        } else if (count < 0) {                     // internal: the right child is the next record
            if (at) idx = pass ? pos + 1 : skip;
            pos = rz_ballot(pass) != 0ull ? pos + 1 : skip;
        } else {                                    // count == 0: the host builder's empty root, or a node the shader's stack could not expand
            if (at) idx = skip;
            pos = skip;
        }
    }
    if (bestTri < 0) return false;
    // the winner's normal and material (FS:411-412, 489-491): the triangle's own normal comes precomputed (DevTriN)
    const float4 nm = *reinterpret_cast<const float4*>(K.triN + bestTri);
    const v3 ln = mk3(nm.x, nm.y, nm.z);
    h.t = tHit;
    h.p = bestP;
    h.n = normalize(x34_normal(K.instances[bestInst].inv, ln));
    h.mat = __float_as_int(nm.w);
    h.inst = bestInst;
    return true;
}

// The same query for a wave whose rays have SPREAD over the scene -- the third and later segments of the paths (the pool
// rounds of a compacting claim, the late rounds of a group) -- where trace_closest's economy turns into its cost: its scalar
// cursor visits every list position some lane needs, one after the other, and serves the instances its lanes enter one after
// the other too.  Measured on C4 (16 instances, 16 spp; profiles/r03_c4_before/): on the third segment 38 lanes of a wave
// were alive, 15 of them entered an instance together, 7 were descending at any step -- the wave walked 4.9 instances per
// query in turn -- and the third and fourth segments took 39 % of the traversal's wave cycles for 6 % of its queries.
// Here every lane is on its own: it walks the pop-order list by itself (its own record fetch per step: the number of steps
// is the longest lane's pops, not the number of positions the wave touches), parks at the first leaf it passes, and when
// every lane is parked or through, ALL parked lanes run their BLAS walks at once, each in its own instance
// (traverse_blas_mi); then they go on behind their leaves.  Per lane the pops, the culls against its own tHit, the instance
// entries and their order are exactly trace_closest's: same result, same tallies; which of the two runs is a scheduling
// decision of the caller (wave-uniform) that cannot change a bit of the image.
template <bool COUNT, bool OVF>
__device__ __forceinline__ bool trace_spread(const KParams& K, v3 o, v3 d, HitRec& h, const BlasStackT<OVF>& bstk, Tally& c) {
    float tHit = 1e30f;
    int bestTri = -1, bestInst = -1;
    v3 bestP = mk3(0.0f, 0.0f, 0.0f);
    if (COUNT) c.traversals += 1;
    const v3 inv = rcp3(d);
    const int nDfs = K.nTlasDfs;
    int idx = 0;                    // the list position this lane visits next (>= nDfs: through)
    int pinst = -1;                 // the instance this lane is about to enter
    int pnext = 0, pend = 0;        // further entries of the same leaf in the TLAS index array (RayZen's own TLAS has one instance per leaf, BVH.cpp:204-208)
    // (a lane enters an instance at most once per leaf entry of the list and every round enters at least one: the cap is a
    //  backstop against a list that says otherwise -- a wave that never leaves this loop takes the device with it)
    bool again;
    int round = 0;
    do {
        // ---- the list: every lane without an instance to enter moves on until it has one or is through
        unsigned long long wm = rz_ballot(pinst < 0 && idx < nDfs);
        int steps = 0;
        while (wm != 0ull) {
            if (in_mask(wm)) {
                RZ_SITE(c, 4);
                const float4* __restrict__ R = reinterpret_cast<const float4*>(K.tlasDfs + idx);
                const float4 r0 = R[0], r1 = R[1];                              // bmin, first | bmax, count
                const int2 r2 = *reinterpret_cast<const int2*>(R + 2);          // skip, inst0
                if (COUNT) c.tlas_nodes += 1;
                float tmin;
                const bool box = slab(o, inv, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tmin);
                const bool pass = box & !(tmin > tHit);
                const int count = __float_as_int(r1.w), first = __float_as_int(r0.w);
                if (count > 0 && pass) { pinst = r2.y; pnext = first + 1; pend = first + count; }
                idx = (count < 0 && pass) ? idx + 1 : r2.x;
            }
            wm = (++steps <= nDfs) ? rz_ballot(pinst < 0 && idx < nDfs) : 0ull;     // (a lane's position strictly grows: at most nDfs steps)
        }
        // ---- the instances: all parked lanes at once, each in its own
        again = rz_ballot(pinst >= 0) != 0ull;
        if (pinst >= 0) {
            if (COUNT) { c.tlas_leaf_indices += 1; c.instances += 1; }
            RZ_SITE(c, 5);
            const float4* __restrict__ I4 = reinterpret_cast<const float4*>(K.instances + pinst);
            const float4 a0 = I4[0], a1 = I4[1], a2 = I4[2];
            const float mi[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            const v3 lo = x34_point(mi, o);
            const v3 ld = normalize(x34_dir(mi, d));
            float tLoc;
            const int tri = traverse_blas_mi<COUNT, OVF>(K, pinst, lo, ld, tLoc, bstk, c);
            if (tri >= 0) {
                const float4 b0 = I4[3], b1 = I4[4], b2 = I4[5];
                const float mf[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
                const v3 localHit = lo + ld * tLoc;              // FS:410
                const v3 worldHit = x34_point(mf, localHit);     // FS:484
                const float tWorld = length(worldHit - o);       // FS:485
                if (tWorld < tHit) { tHit = tWorld; bestP = worldHit; bestTri = tri; bestInst = pinst; }
            }
            if (pnext < pend) { pinst = K.tlasIndices[pnext]; ++pnext; } else pinst = -1;
        }
    } while (again && ++round < K.traceRoundCap);
    if (bestTri < 0) return false;
    const float4 nm = *reinterpret_cast<const float4*>(K.triN + bestTri);
    const v3 ln = mk3(nm.x, nm.y, nm.z);
    h.t = tHit;
    h.p = bestP;
    h.n = normalize(x34_normal(K.instances[bestInst].inv, ln));
    h.mat = __float_as_int(nm.w);
    h.inst = bestInst;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// The closest-hit queries of a POOL of paths, traced together (the pool rounds of a compacting claim, rz_kernels.hip).
//
// Why.  The third and later segments of the paths of a claim are compacted into a pool; round 2 worked it off 64 paths at a
// time with trace_closest.  Those rays share nothing, and their BLAS walks differ wildly in length -- a ray that clips a
// corner of an instance's box is through in three steps, one that grazes the mesh takes two hundred -- while a wave's walk
// lasts as long as its longest lane's: measured on C2, descend steps of the pool rounds ran 14.7 and 9.5 lanes wide with 58
// paths in the wave; on C4 (sixteen instances, entered one after the other) 7.1 and 3.3.  A fifth (C2) to two fifths (C4) of
// the frame went there for 4-6 % of its queries.
//
// How.  A query is a sequence of list steps and BLAS walks of ONE path; nothing orders the queries of different paths.  So
// the pool's queries advance in alternating phases:
//   * T phase, 64 queries at a time, one lane each: walk the pop-order list (as trace_spread does) to the next leaf the ray
//     passes and emit an ITEM (query, instance) -- or finish the query;
//   * B phase: the items are walked by a wave whose lanes REFILL: a lane whose walk is over writes its result to its query
//     (world-space hit, kept if strictly nearer: FS:484-486) and takes the next item from the list, so the wave's lanes stay
//     busy until the list runs out instead of waiting for the longest walk of a fixed set of 64.
// Per query the pops, the culls against its own tHit, the instance entries and their order are exactly trace_closest's (a
// query has one item in flight at a time); only WHICH lane runs a step, and when, differs.  Same results, same tallies.
// All state lives in the claim's scratch (rz_scene_dev.h: pool fields), read and written only by this wave.
namespace poolf {      // field numbers of the pool (each field is PS consecutive dwords)
enum : int { OX = 0, OY, OZ, DX, DY, DZ, TPX, TPY, TPZ, SEEDX, SEEDY, SAMP, BACK,      // the parked path (rz_kernels.hip)
             QT = 13, QPX, QPY, QPZ, QTRI, QINST, QIDX, QSUB,                            // its query: tHit, hit point, winner, list position, entry within a multi-instance leaf
             ITSLOT = 21, ITINST = 22,                                                   // the items of the current B phase
             IOR = 23 };                                                                 // transparent scenes: the currentIor of a path that may read it (rz_kernels.hip: released late samples)
}
static_assert(poolf::IOR + 1 == RZ_POOL_FIELDS, "pool fields");

#ifndef RZ_REFILL_MIN_LANES
#define RZ_REFILL_MIN_LANES 8      // idle lanes it takes to interrupt the walk for a refill (unless nobody walks at all)
#endif
#ifndef RZ_POOL_DESCEND_MIN_LANES
#define RZ_POOL_DESCEND_MIN_LANES 4    // the pool's walks leave the descend loop for the leaves when fewer lanes than this still descend
#endif
#ifndef RZ_TAIL_LANES
#define RZ_TAIL_LANES 16           // a B phase whose list has run out ends when fewer lanes than this still walk (they go on in the next one)
#endif

// pool: field f of slot s is pool[f * PS + s] (PS: the capacity of the wave's pool).
template <bool COUNT, bool OVF>
__device__ __forceinline__ void pool_trace(const KParams& K, unsigned* __restrict__ pool, const size_t PS, const int nQ, const BlasStackT<OVF>& bstk, Tally& c) {
    using namespace poolf;
    const int lane = threadIdx.x & 63;
    const int nDfs = K.nTlasDfs;
    const unsigned long long below = (1ull << lane) - 1ull;
    constexpr unsigned BUSY = 0x80000000u;      // in QIDX: an item of this query is being walked (the T phase leaves the query alone)
    for (int s = lane; s < nQ; s += 64) {
        pool[QT * PS + s] = __float_as_uint(1e30f);
        pool[QTRI * PS + s] = 0xffffffffu;
        pool[QIDX * PS + s] = 0u;
        pool[QSUB * PS + s] = 0u;
    }
    if (COUNT) for (int s = lane; s < nQ; s += 64) c.traversals += 1;
    __syncthreads();
    // the walk a lane is in lives across the phases: a B phase ends when its list has run out and only a few long walks are
    // left -- those go on in the next B phase, beside the items the T phase has made in the meantime, instead of keeping
    // sixty lanes waiting for them (measured with a B phase that ran to its last walk: chunks of 512 queries walked 8 lanes wide)
    int slot = -1, inst = 0;                    // the item this lane holds (slot < 0: none)
    int cur = -1, sp = 0, best = -1, pbase = 0, tbase = 0;
    float tLoc = 1e30f;
    v3 lo = mk3(0.0f, 0.0f, 0.0f), ld = mk3(1.0f, 1.0f, 1.0f);
    RayPk RP = make_raypk(lo, ld);
    // (every iteration retires at least one item or finishes: a query enters at most K.traceRoundCap instances and there are
    //  nQ of them -- the bound is a backstop; a wave that never leaves this loop takes the device with it)
    const long long maxIter = (long long)K.traceRoundCap * (nQ > 0 ? nQ : 1) + 8;
    for (long long iter = 0; iter < maxIter; ++iter) {
        // ---- T phase: every query that is neither finished nor being walked moves on to the next leaf its ray passes
#ifdef RZ_PROF
        const unsigned long long tT0_ = __builtin_amdgcn_s_memtime();
        c.rnd = 7;      // (the per-round site counters of the diagnostic build: everything pool_trace does is filed under round 7)
#endif
        int nItems = 0;
        for (int sb = 0; sb < nQ; sb += 64) {
            const int s = sb + lane;
            const unsigned raw = s < nQ ? pool[QIDX * PS + s] : (unsigned)nDfs;
            const bool mine = s < nQ && (raw & BUSY) == 0u;
            int idx = mine ? (int)raw : nDfs;
            unsigned long long wm = rz_ballot(idx < nDfs);
            if (wm == 0ull) continue;
            int sub = 0, pinst = -1;
            v3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(1.0f, 1.0f, 1.0f);
            float tHit = 1e30f;
            if (idx < nDfs) {
                o = mk3(__uint_as_float(pool[OX * PS + s]), __uint_as_float(pool[OY * PS + s]), __uint_as_float(pool[OZ * PS + s]));
                d = mk3(__uint_as_float(pool[DX * PS + s]), __uint_as_float(pool[DY * PS + s]), __uint_as_float(pool[DZ * PS + s]));
                tHit = __uint_as_float(pool[QT * PS + s]);
                sub = (int)pool[QSUB * PS + s];
            }
            const v3 inv = rcp3(d);
            int steps = 0;
            while (wm != 0ull) {
                if (in_mask(wm)) {
                    RZ_SITE(c, 4);
                    const float4* __restrict__ R = reinterpret_cast<const float4*>(K.tlasDfs + idx);
                    const float4 r0 = R[0], r1 = R[1];                              // bmin, first | bmax, count
                    const int2 r2 = *reinterpret_cast<const int2*>(R + 2);          // skip, inst0
                    const int count = __float_as_int(r1.w), first = __float_as_int(r0.w);
                    if (sub > 0) {      // inside a multi-instance leaf: FS:471-496 enters its instances one after the other without testing the box again
                        pinst = K.tlasIndices[first + sub];
                        sub = sub + 1 < count ? sub + 1 : 0;
                        if (sub == 0) idx = r2.x;
                    } else {
                        if (COUNT) c.tlas_nodes += 1;
                        float tmin;
                        const bool box = slab(o, inv, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tmin);
                        const bool pass = box & !(tmin > tHit);
                        if (count > 0 && pass) {
                            pinst = r2.y;
                            if (count > 1) sub = 1; else idx = r2.x;
                        } else {
                            idx = (count < 0 && pass) ? idx + 1 : r2.x;
                        }
                    }
                }
                wm = (++steps <= nDfs) ? rz_ballot(pinst < 0 && idx < nDfs) : 0ull;     // (a lane's position strictly grows: at most nDfs steps)
            }
            if (mine) { pool[QIDX * PS + s] = (unsigned)idx | (pinst >= 0 ? BUSY : 0u); pool[QSUB * PS + s] = (unsigned)sub; }
            const unsigned long long hm = rz_ballot(pinst >= 0);
            if (pinst >= 0) {
                const int k = nItems + __popcll(hm & below);
                pool[ITSLOT * PS + k] = (unsigned)s;
                pool[ITINST * PS + k] = (unsigned)pinst;
            }
            nItems += mask_count(hm);
        }
#ifdef RZ_PROF
        const unsigned long long tB0_ = __builtin_amdgcn_s_memtime();
        c.t[13] += tB0_ - tT0_;
#endif
        if (nItems == 0 && rz_ballot(slot >= 0) == 0ull) break;      // no query has anywhere to go and nobody walks: the pool is traced
        __syncthreads();
        // ---- B phase: walk the items, lanes refilling from the list
        int next = 0;                               // wave-uniform: the next item to hand out
        int retired = 0;                            // items retired in this phase
        // (an item is walked in fewer than 2 x its BLAS's nodes rounds -- the host has checked that the node arrays are trees --
        //  and every refill hands out or retires an item: the bound is a backstop)
        bool again;
        unsigned guard = 0;
        do {
            // -- refill: idle lanes retire the item they hold and take the next one
            bool idle = cur == -1 && sp == 0;
            unsigned long long im = rz_ballot(idle);
            const int walkers = mask_count(rz_ballot(!idle));
            // (when the list has run out the phase is about to end -- see below -- and the finished items are retired on the way out)
            bool refill = (next < nItems && (mask_count(im) >= RZ_REFILL_MIN_LANES || walkers == 0)) ||
                          (next >= nItems && rz_ballot(idle && slot >= 0) != 0ull && (walkers < RZ_TAIL_LANES || mask_count(im) >= RZ_REFILL_MIN_LANES));
#ifdef RZ_PROF
            const unsigned long long tR0_ = __builtin_amdgcn_s_memtime();
            RZ_SITE(c, 0);              // rounds of the B phase and the lanes in them
            if (refill && idle) RZ_SITE(c, 1);     // refill events and the idle lanes they serve
#endif
            while (refill) {
                const int pending = nItems - next;
                retired += mask_count(rz_ballot(idle && slot >= 0));       // (wave-uniform: counted outside the lanes' branch)
                if (idle) {
                    if (slot >= 0) {
                        if (best >= 0) {       // FS:410, 484-486: the hit in world space, kept if strictly nearer
                            const float4* __restrict__ I4 = reinterpret_cast<const float4*>(K.instances + inst);
                            const float4 b0 = I4[3], b1 = I4[4], b2 = I4[5];
                            const float mf[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
                            const v3 o = mk3(__uint_as_float(pool[OX * PS + slot]), __uint_as_float(pool[OY * PS + slot]), __uint_as_float(pool[OZ * PS + slot]));
                            const v3 localHit = lo + ld * tLoc;
                            const v3 worldHit = x34_point(mf, localHit);
                            const float tWorld = length(worldHit - o);
                            if (tWorld < __uint_as_float(pool[QT * PS + slot])) {
                                pool[QT * PS + slot] = __float_as_uint(tWorld);
                                pool[QPX * PS + slot] = __float_as_uint(worldHit.x); pool[QPY * PS + slot] = __float_as_uint(worldHit.y); pool[QPZ * PS + slot] = __float_as_uint(worldHit.z);
                                pool[QTRI * PS + slot] = (unsigned)best;
                                pool[QINST * PS + slot] = (unsigned)inst;
                            }
                        }
                        pool[QIDX * PS + slot] &= ~BUSY;        // the T phase may move this query on
                    }
                    slot = -1;
                    const int rank = __popcll(im & below);
                    if (rank < pending) {
                        const int k = next + rank;
                        slot = (int)pool[ITSLOT * PS + k];
                        inst = (int)pool[ITINST * PS + k];
                        if (COUNT) { c.tlas_leaf_indices += 1; c.instances += 1; c.blas_nodes += 1; }
                        RZ_SITE(c, 5);
                        const v3 o = mk3(__uint_as_float(pool[OX * PS + slot]), __uint_as_float(pool[OY * PS + slot]), __uint_as_float(pool[OZ * PS + slot]));
                        const v3 d = mk3(__uint_as_float(pool[DX * PS + slot]), __uint_as_float(pool[DY * PS + slot]), __uint_as_float(pool[DZ * PS + slot]));
                        const float4* __restrict__ I4 = reinterpret_cast<const float4*>(K.instances + inst);
                        const float4 a0 = I4[0], a1 = I4[1], a2 = I4[2];
                        const float mi[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
                        lo = x34_point(mi, o);
                        ld = normalize(x34_dir(mi, d));
                        const v3 linv = rcp3(ld);
                        const float4 r0 = I4[6], r1 = I4[7];                         // rootMin[3], rootEnc | rootMax[3], pairBase
                        const int2 r2 = *reinterpret_cast<const int2*>(I4 + 8);      // triBase, flags
                        float tminRoot;
                        bool go = slab(lo, linv, r0.x, r0.y, r0.z, r1.x, r1.y, r1.z, tminRoot);
                        go = go && !(tminRoot > 1e30f) && !(r2.y & 1);
                        cur = go ? __float_as_int(r0.w) : -1;
                        pbase = __float_as_int(r1.w); tbase = r2.x;
                        RP = make_raypk(lo, linv);
                        tLoc = 1e30f; best = -1; sp = 0;
                    }
                }
                next += pending < mask_count(im) ? pending : mask_count(im);
                idle = cur == -1 && sp == 0;
                im = rz_ballot(idle);
                // (again at once if the lanes just served are idle already -- their rays missed the root box -- and items remain)
                refill = next < nItems && (mask_count(im) >= RZ_REFILL_MIN_LANES || rz_ballot(!idle) == 0ull);
            }
#ifdef RZ_PROF
            c.t[15] += __builtin_amdgcn_s_memtime() - tR0_;
#endif
            // -- descend (the loops of blas_walk<.., MI = true>)
            unsigned long long actMask = rz_ballot(cur >= 0);
            bool more = actMask != 0ull;
            while (more) {
                if (in_mask(actMask)) {
                    RZ_SITE(c, 3);
                    if (COUNT) c.blas_nodes += 2;
                    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(K.pairs + (pbase + cur));
                    float4 p0 = p4[0], p1 = p4[1], p2 = p4[2], p3 = p4[3];
                    RZ_KEEP4(p0); RZ_KEEP4(p3);
                    const f32x2 lx = {p0.x, p0.y}, ly = {p0.z, p0.w}, lz = {p1.x, p1.y};
                    const f32x2 rx = {p1.z, p1.w}, ry = {p2.x, p2.y}, rz = {p2.z, p2.w};
                    float tl, tr;
                    bool hl, hr;
                    RZ_SLAB_PAIR(-1, RP, lx, ly, lz, rx, ry, rz, hl, tl, hr, tr);
                    const int lenc = __float_as_int(p3.x), renc = __float_as_int(p3.y);
                    const bool takeR = hr && !(tr > tLoc);
#ifdef RZ_PROF
                    { const bool takeL_ = hl && !(tl > tLoc); c.ps[4 + (takeR ? 1 : 0) + (takeL_ ? 1 : 0)] += 1u; }
#endif
                    if (hl && takeR) push_entry(bstk, sp, make_uint2((unsigned)lenc, __float_as_uint(tl)));
                    int nxt = takeR ? renc : ((tl > tLoc) ? -1 : lenc);
                    if (!hl && !takeR) {
                        if (!pop_entry(bstk, sp, tLoc, nxt)) nxt = -1;
                    }
                    cur = nxt;
                }
                actMask = rz_ballot(cur >= 0);
                more = mask_count(actMask) >= RZ_POOL_DESCEND_MIN_LANES;
            }
            // -- leaves
            const unsigned long long alive = rz_ballot((cur != -1) || (sp > 0));
            if (alive != 0ull) {
                const bool leaf = cur < 0;
                const int v = ~cur;
                const int first = (v >> 4) + tbase, count = leaf ? (v & 15) : 0;
                if (COUNT) c.triangles += (unsigned)count;
                int i = 0;
                unsigned long long triMask = rz_ballot(0 < count);
                bool any = triMask != 0ull;
                while (any) {
                    if (in_mask(triMask)) {
                        RZ_SITE(c, 2);
                        const float4* __restrict__ tp = reinterpret_cast<const float4*>(K.tris + first + i);
                        float4 a = tp[0], b = tp[1], cc = tp[2];
                        RZ_KEEP4(a);
                        float t;
                        bool pastU;
                        const bool hit = moller_trumbore(lo, ld, mk3(a.x, a.y, a.z), mk3(a.w, b.x, b.y), mk3(b.z, b.w, cc.x), t, pastU);
                        if (COUNT && pastU) c.triangles_past_u += 1;
                        if (hit && t < tLoc) { tLoc = t; best = first + i; }
                    }
                    ++i;
                    triMask = rz_ballot(i < count);
                    any = triMask != 0ull;
                }
                if (leaf) {
                    if (!pop_entry(bstk, sp, tLoc, cur)) cur = -1;
                }
            }
            // The phase goes on while its list lasts, and after that while many lanes still walk; it ends -- once something has
            // been retired, so that the next T phase has a query to move -- when at most RZ_TAIL_LANES long walks are left: they
            // continue in the next B phase.  Lanes that hold a finished item have retired it above by then (refill's second case).
            const int stillWalking = mask_count(rz_ballot((cur != -1) || (sp > 0)));
            const bool holding = rz_ballot(slot >= 0 && cur == -1 && sp == 0) != 0ull;
            again = next < nItems || stillWalking >= RZ_TAIL_LANES || holding || (stillWalking > 0 && retired == 0);
            // (holding: one more round, whose refill retires the finished items before the T phase looks at their queries)
        } while (again && ++guard < (1u << 28));
#ifdef RZ_PROF
        c.t[14] += __builtin_amdgcn_s_memtime() - tB0_;
#endif
        __syncthreads();
    }
}

}  // namespace rz
