// rz_device_math.h -- arithmetic of the path tracer on gfx950.
//
// The renderer must reproduce RayZen's fragment shader (RayZen/shaders/
// fragment_shader.glsl, "FS") decision for decision: its random numbers are
// fract(sin(x)*43758.5453) of arguments up to ~1e10, so one ulp anywhere
// upstream flips whole paths.  Hence:
//   * every float operation below is a single IEEE binary32 operation in the
//     shader's operand order; the translation unit is compiled with
//     -ffp-contract=off so hipcc emits v_mul_f32 + v_add_f32, never v_fmac;
//     division and sqrt are the correctly rounded forms (hipcc default
//     -fhip-fp32-correctly-rounded-divide-sqrt);
//   * GLSL min/max/clamp = v_min_f32 / v_max_f32 (IEEE minNum/maxNum: the
//     non-NaN operand wins), which is also what a GPU running FS does;
//   * sin/cos/acos, which GLSL leaves to the driver, are those of the driver
//     RayZen's own shader was run on for this project (Mesa llvmpipe; see
//     "sin / cos / acos" below): with them this renderer draws the random
//     numbers that run of the reference drew.  Rounds 1-4's definition
//     (binary64, correctly rounded) stays behind -DRZ_MATH_FLAVOUR=0.
#pragma once
#include <hip/hip_runtime.h>

namespace rz {

// Lane mask of a predicate (HIP's __ballot takes an int: the bool -> int -> bool round trip was materialised as
// v_cndmask + v_cmp_ne in the loops whose predicate is an AND of lane masks).
__device__ __forceinline__ unsigned long long rz_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

struct v3 { float x, y, z; };
struct v2 { float x, y; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
// GLSL dot / cross, evaluated left to right
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) {
    return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float clamp_(float x, float lo, float hi) { return fmin_(fmax_(x, lo), hi); }
__device__ __forceinline__ float mix_(float a, float b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ float fract_(float x) { return x - __builtin_floorf(x); }
__device__ __forceinline__ float pow2_(float x) { return x * x; }
__device__ __forceinline__ float pow5_(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

// column-major mat4 (16 floats) applied to (v,1) / (v,0); rows 0..2 only
__device__ __forceinline__ v3 xform_point(const float* m, v3 v) {
    return mk3(((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12],
               ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
               ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]);
}
__device__ __forceinline__ v3 xform_dir(const float* m, v3 v) {
    return mk3((m[0] * v.x + m[4] * v.y) + m[8] * v.z,
               (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
               (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}
// mat3(transpose(m)) * v
__device__ __forceinline__ v3 xform_normal(const float* m, v3 v) {
    return mk3((m[0] * v.x + m[1] * v.y) + m[2] * v.z,
               (m[4] * v.x + m[5] * v.y) + m[6] * v.z,
               (m[8] * v.x + m[9] * v.y) + m[10] * v.z);
}

// ---- reciprocals ------------------------------------------------------------
// 1.0f / x must be the correctly rounded IEEE quotient (section "pinned numerics" of DESIGN.md); hipcc expands it to 11
// instructions (two v_div_scale, v_rcp, five FMA-class steps, v_div_fmas, v_div_fixup: ~31 issue cycles).  For
// 2^-126 <= |x| <= 2^126 -- x and 1/x both normal -- ONE Newton step on the hardware's 1-ulp v_rcp_f32 already rounds
// correctly: checked for every one of the 2^32 bit patterns (profiles/scripts/rcp_exhaustive.hip; tests/test_rcp_gpu.py
// repeats the sweep on the GPU the suite runs on), so the callers use it when a wave-level range test passes and the
// full division otherwise.  Outside that range the short form is wrong (denormal quotients are rounded twice).
__device__ __forceinline__ float rcp_mid(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ bool rcp_mid_ok(float x) {       // false for NaN, zero, denormals, infinities too
    const float a = __builtin_fabsf(x);
    return (a >= 0x1p-126f) & (a <= 0x1p126f);
}
// The IEEE expansions behind the short forms' range votes are COLD (a wave takes them when one lane's operand is denormal, huge, zero
// or NaN) and big (11-13 instructions per quotient or root, dozens of sites): out of line since round 5 -- the C2 kernel is 1 400
// instructions shorter, 32 instead of 37 spilled VGPRs, C2 -0.8 %, C3 -0.9 %, C4 -1.0 %, c2g -1.3 % (RayZen's scene at 64 spp +1.0 %;
// profiles/r05_regs/).  RZ_SLOW_PATHS_INLINE=1 restores the inlined form (A/B aid).
#if defined(RZ_SLOW_PATHS_INLINE) && RZ_SLOW_PATHS_INLINE
#define RZ_SLOWFN __device__ __forceinline__
#else
#define RZ_SLOWFN static __device__ __attribute__((noinline))
#endif
RZ_SLOWFN v3 rcp3_ieee(v3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
__device__ __forceinline__ float rcp_ieee(float x) { return 1.0f / x; }
RZ_SLOWFN v3 div3_ieee(v3 a, float b) { return a / b; }
RZ_SLOWFN float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
RZ_SLOWFN v3 normalize_ieee(v3 a, float d2) { return a / __builtin_sqrtf(d2); }
// (1/x, 1/y, 1/z) of a ray direction
__device__ __forceinline__ v3 rcp3(v3 d) {
    const bool okx = rcp_mid_ok(d.x), oky = rcp_mid_ok(d.y), okz = rcp_mid_ok(d.z);
    const bool ok = okx & oky & okz;
    if (rz_ballot(!ok) == 0ull) return mk3(rcp_mid(d.x), rcp_mid(d.y), rcp_mid(d.z));
    return rcp3_ieee(d);
}

// ---- quotients and square roots ---------------------------------------------
// a / b and sqrt(x) must be the correctly rounded IEEE results (pinned numerics); hipcc's expansions are 11 and 13
// instructions (31 and 42 issue cycles).  Shorter forms, each admitted only behind a range test on its operands (a wave
// votes: one lane outside the range sends the whole wave through the compiler's expansion -- same result either way):
//   * div_mid(a, b, r) with r = rcp_mid(b), the correctly rounded reciprocal: q0 = a r; e = fma(-q0, b, a) -- the exact
//     residual; q = fma(e, r, q0).  Markstein's correction step; his theorem wants q0 within an ulp of a / b, which
//     RN(a RN(1 / b)) is not proven to be, so the claim is a MEASURED one: profiles/scripts/div_sqrt_proof.hip tries 2^33
//     random pairs, every mantissa of either operand against 64 values of the other and the exponent boundaries
//     (tests/test_div_sqrt_gpu.py repeats it on the GPU of the test run).  Operands within [2^-60, 2^60] keep the quotient,
//     and the residual's last bit, normal; a numerator of +0 is let through (q0 = +-0 with the quotient's sign, the residual
//     is +0 and the final fma returns q0); -0 is not (the residual +0 would turn a quotient of -0 into +0 when b > 0).
//     Several quotients by ONE divisor share r: normalize costs one reciprocal, not three divisions.
//   * sqrt_mid(x): s = v_sqrt_f32(x) (1 ulp); e = fma(-s, s, x); fma(e, h, s) with h = 0.5 v_rsq_f32(x) -- PROVEN by
//     enumeration of every x in [2^-100, 2^100] (the same program: no mismatch from 2^-102 up to the largest finite x).
__device__ __forceinline__ bool range_ok(unsigned absbits, unsigned lo, unsigned hi) { return absbits - lo <= hi - lo; }    // lo <= absbits <= hi, one unsigned compare
constexpr unsigned RZ_F32_2M60 = (127u - 60u) << 23, RZ_F32_2P60 = (127u + 60u) << 23, RZ_F32_2M100 = (127u - 100u) << 23, RZ_F32_2P100 = (127u + 100u) << 23;
__device__ __forceinline__ bool div_mid_den_ok(float b) { return range_ok(__float_as_uint(b) & 0x7fffffffu, RZ_F32_2M60, RZ_F32_2P60); }
__device__ __forceinline__ bool div_mid_num_ok(float a) {
    const unsigned raw = __float_as_uint(a);
    return raw == 0u || range_ok(raw & 0x7fffffffu, RZ_F32_2M60, RZ_F32_2P60);
}
__device__ __forceinline__ float div_mid(float a, float b, float r) {     // r = rcp_mid(b)
    const float q0 = a * r;
    const float e = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(e, r, q0);
}
__device__ __forceinline__ bool sqrt_mid_ok(float x) { return range_ok(__float_as_uint(x), RZ_F32_2M100, RZ_F32_2P100); }   // (negative, zero, NaN, infinity: no)
__device__ __forceinline__ float sqrt_mid(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);        // (0.5 v_rcp(s) instead fails for the 100 inputs (2 - 2^-23) 4^k of the range)
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

// Three numerators at once, the SAME admitted set as div_mid_num_ok per component: +0, or magnitude within [2^-60, 2^60].
// Two folded integer tests: (1) min(|bits| - lo, bits) is 0 for +0, stays out of range for -0 and for everything above the
// range -- but a POSITIVE magnitude below 2^-60 (denormals included) wraps high in the first operand and is let through by
// the second (round 3 shipped this test alone: ADVICE r3; for such numerators the residual fma(-q0, b, a) is no longer exact
// and div_mid misses the IEEE quotient); (2) min3(|bits| - 1) >= lo - 1 holds iff no component has 0 < |bits| < lo (zero
// wraps to the top).  profiles/scripts/div_sqrt_proof.hip compares this predicate with div_mid_num_ok on every exponent,
// both signs, zeros, denormals, infinities and NaNs, and drives div3 / normalize themselves.
__device__ __forceinline__ bool div_mid_num3_ok(v3 a) {
    const unsigned rx = __float_as_uint(a.x), ry = __float_as_uint(a.y), rz = __float_as_uint(a.z);
    const unsigned ax = rx & 0x7fffffffu, ay = ry & 0x7fffffffu, az = rz & 0x7fffffffu;
    const unsigned tx = min(ax - RZ_F32_2M60, rx), ty = min(ay - RZ_F32_2M60, ry), tz = min(az - RZ_F32_2M60, rz);
    const bool inOrZero = max(max(tx, ty), tz) <= RZ_F32_2P60 - RZ_F32_2M60;
    const bool noTiny = min(min(ax - 1u, ay - 1u), az - 1u) >= RZ_F32_2M60 - 1u;
    return inOrZero & noTiny;
}
// (a.x, a.y, a.z) / b, each the correctly rounded quotient: one reciprocal and three correction steps when the wave's operands
// are all in range, three IEEE divisions otherwise.
__device__ __forceinline__ v3 div3(v3 a, float b) {
    if (rz_ballot(!(div_mid_den_ok(b) && div_mid_num3_ok(a))) == 0ull) {
        const float r = rcp_mid(b);
        return mk3(div_mid(a.x, b, r), div_mid(a.y, b, r), div_mid(a.z, b, r));
    }
    return div3_ieee(a, b);
}
// GLSL length / normalize with their pinned definitions, sqrt(dot(v, v)) and v / sqrt(dot(v, v)): the square root and the three
// quotients correctly rounded, by the short forms when the wave's operands allow (dot in [2^-100, 2^100] puts the length within
// [2^-50, 2^50], inside the divisor's range), by the compiler's expansions otherwise.  (Round 2 paid 42 + 3 x 31 issue cycles
// per normalize for them; a closest-hit query through two instances normalizes three vectors and takes two lengths.)
__device__ __forceinline__ float length(v3 a) {
    const float d2 = dot(a, a);
    if (rz_ballot(!sqrt_mid_ok(d2)) == 0ull) return sqrt_mid(d2);
    return sqrt_ieee(d2);
}
__device__ __forceinline__ v3 normalize(v3 a) {
    const float d2 = dot(a, a);
    if (rz_ballot(!(sqrt_mid_ok(d2) && div_mid_num3_ok(a))) == 0ull) {
        const float s = sqrt_mid(d2);
        const float r = rcp_mid(s);
        return mk3(div_mid(a.x, s, r), div_mid(a.y, s, r), div_mid(a.z, s, r));
    }
    return normalize_ieee(a, d2);
}

// ---- sin / cos / acos -----------------------------------------------------
// GLSL leaves these three to the implementation; RZ_MATH_FLAVOUR chooses which implementation the product IS:
//   1: Mesa llvmpipe's -- the OpenGL implementation RayZen's own shader was RUN on for this project (the test harness beside the
//      oracle; tests/test_glref.py): sin / cos = Cephes' single-precision routine as in sse_mathfun / gallivm (octant
//      j = (trunc(|x| 4/pi) + 1) & ~1 with x86's out-of-range conversion, three-constant Cody-Waite reduction and both polynomials
//      with FUSED multiply-adds, result clamped to [-1, 1]), acos = Mesa's GLSL polynomial in unfused binary32.  With it the
//      renderer draws the random numbers RayZen's shader draws on that implementation (FS:188-190: fract(sin(x) * 43758.5453),
//      x to 1e11 -- the range reduction decides every bit), so its frames can be held against the reference's own frames pixel
//      by pixel at every bounce budget.  ~25 binary32 instructions per sin.
//      THE DEFAULT since round 5.  Same box, against flavour 0 (profiles/r05_flavour/): C2 10.31 -> 10.10 ms although the frame
//      does 2 % more traversal work with these random numbers, C4 7.41 -> 6.78, c2g 21.3 -> 18.7, RayZen's own frame 0.310 ->
//      0.251 ms, c2close +0.8 %, glassbunny +1.7 %.
//   0: rounds 1-4's definition: binary64 evaluation (Cody-Waite by pi/2 in three fma steps, fdlibm's k_sin / k_cos / e_acos
//      polynomials) rounded once to binary32 -- correctly rounded, ~77 binary64 instructions per sin at half rate.  Still
//      built by -DRZ_MATH_FLAVOUR=0 (its hemisphere draw then goes behind a call again: rz_path.h) and tested the same way: the oracle holds
//      both definitions and the suite asks the loaded library which one it is (rz_math_flavour()).
// The reduction + both polynomials (sincos_core, ~25 instructions, five values out) sit behind ONE call shared by sin_ / cos_ /
// sincos_ (RZ_SINCOS_NOINLINE): inlined at its seven sites it cost the opaque kernels 1-3 % (C2 10.37 vs 10.10 ms, C3 39.8 vs
// 38.5), the same effect as every other piece of cold, stateless code moved out of the one big function (profiles/r05_regs/).
#ifndef RZ_MATH_FLAVOUR
#define RZ_MATH_FLAVOUR 1
#endif
#if RZ_MATH_FLAVOUR == 1
struct SinCosF { float xr, z, ps, pc; unsigned j; };
#ifndef RZ_SINCOS_NOINLINE
#define RZ_SINCOS_NOINLINE 1
#endif
#if RZ_SINCOS_NOINLINE
static __device__ __attribute__((noinline)) SinCosF sincos_core(float x) {
#else
__device__ __forceinline__ SinCosF sincos_core(float x) {
#endif      // x >= 0 (the callers pass |x|)
    const float scale = x * 1.27323954473516f;
    // cvttps2dq: truncation, and 0x80000000 for everything out of range (v_cvt_i32_f32 would saturate)
    const int jt = (scale < 2147483648.0f) ? (int)scale : (int)0x80000000;
    SinCosF o;
    o.j = ((unsigned)jt + 1u) & ~1u;
    const float y = (float)(int)o.j;
    float xr = __builtin_fmaf(y, -0.78515625f, x);
    xr = __builtin_fmaf(y, -2.4187564849853515625e-4f, xr);
    xr = __builtin_fmaf(y, -3.77489497744594108e-8f, xr);
    const float z = xr * xr;
    float c = __builtin_fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    c = __builtin_fmaf(c, z, 4.166664568298827E-002f);
    c = c * z;
    c = c * z;
    c = c - z * 0.5f;
    c = c + 1.0f;
    float sn = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    sn = __builtin_fmaf(sn, z, -1.6666654611E-1f);
    sn = sn * z;
    sn = __builtin_fmaf(sn, xr, xr);
    o.xr = xr; o.z = z; o.ps = sn; o.pc = c;
    return o;
}
__device__ __forceinline__ float sincos_finish(float v, unsigned signbit) {
    float r = __uint_as_float(__float_as_uint(v) ^ signbit);
    r = (r < 1.0f) ? r : 1.0f;          // (x86 minps / maxps: a NaN -- x^2 overflowed -- yields the bound)
    r = (r > -1.0f) ? r : -1.0f;
    return r;
}
__device__ __forceinline__ float sin_(float x) {
    const SinCosF t = sincos_core(__builtin_fabsf(x));
    const unsigned sign = (__float_as_uint(x) & 0x80000000u) ^ ((t.j & 4u) << 29);
    return sincos_finish((t.j & 2u) == 0u ? t.ps : t.pc, sign);
}
__device__ __forceinline__ float cos_(float x) {
    const SinCosF t = sincos_core(__builtin_fabsf(x));
    const unsigned e = t.j - 2u;
    return sincos_finish((e & 2u) == 0u ? t.ps : t.pc, ((~e) & 4u) << 29);
}
// sin and cos of the same angle from one reduction and one pair of polynomials: the bits of sin_(x) and cos_(x).
__device__ __forceinline__ void sincos_(float x, float& s, float& c) {
    const SinCosF t = sincos_core(__builtin_fabsf(x));
    const unsigned e = t.j - 2u;
    s = sincos_finish((t.j & 2u) == 0u ? t.ps : t.pc, (__float_as_uint(x) & 0x80000000u) ^ ((t.j & 4u) << 29));
    c = sincos_finish((e & 2u) == 0u ? t.ps : t.pc, ((~e) & 4u) << 29);
}
__device__ __forceinline__ float acos_(float x) {
    const float PIO2 = 1.57079632679489661923f, PIO4M1 = 0.78539816339744830962f - 1.0f;
    const float ax = __builtin_fabsf(x);
    float e = 0.08132463f + ax * -0.02363318f;
    e = PIO4M1 + ax * e;
    e = PIO2 + ax * e;
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    const float t = 1.0f - ax;
    const float rt = (rz_ballot(!sqrt_mid_ok(t)) == 0ull) ? sqrt_mid(t) : sqrt_ieee(t);
    const float as = sgn * (PIO2 - rt * e);
    return PIO2 - as;
}
#else
// A binary64 literal cannot be an inline operand on gfx950, so hipcc materialises each one in a register pair -- and,
// left alone, hoists all ~24 of them out of the sample loop and keeps them in VGPRs across the whole BVH walk
// (measured: the fused kernel wanted 193 VGPRs where trace and shading need 82 and 84 on their own).  KD() pins a
// constant to an SGPR pair created where it is used (two s_mov_b32 on the scalar unit; v_fma_f64 takes one scalar
// operand) and, being volatile, stays inside the function.
__device__ __forceinline__ double KD(double v) { asm volatile("" : "+s"(v)); return v; }

struct SinCos { double s, c; int q; };

__device__ __forceinline__ SinCos sincos_core(float xf) {
    const double x = (double)xf;
    const double k = __builtin_rint(x * KD(6.36619772367581382433e-01));
    double r = __builtin_fma(-k, KD(1.57079632673412561417e+00), x);
    r = __builtin_fma(-k, KD(6.07710050630396597660e-11), r);
    r = __builtin_fma(-k, KD(2.02226624871116645580e-21), r);
    const double q = k - 4.0 * __builtin_floor(k * 0.25);
    const double z = r * r;
    double ps = __builtin_fma(z, KD(1.58969099521155010221e-10), KD(-2.50507602534068634195e-08));
    ps = __builtin_fma(z, ps, KD(2.75573137070700676789e-06));
    ps = __builtin_fma(z, ps, KD(-1.98412698298579493134e-04));
    ps = __builtin_fma(z, ps, KD(8.33333333332248946124e-03));
    ps = __builtin_fma(z, ps, KD(-1.66666666666666324348e-01));
    double pc = __builtin_fma(z, KD(-1.13596475577881948265e-11), KD(2.08757232129817482790e-09));
    pc = __builtin_fma(z, pc, KD(-2.75573143513906633035e-07));
    pc = __builtin_fma(z, pc, KD(2.48015872894767294178e-05));
    pc = __builtin_fma(z, pc, KD(-1.38888888888741095749e-03));
    pc = __builtin_fma(z, pc, KD(4.16666666666666019037e-02));
    SinCos o;
    o.s = __builtin_fma(z * r, ps, r);
    o.c = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    o.q = (int)q;
    return o;
}
__device__ __forceinline__ float sin_(float x) {
    SinCos t = sincos_core(x);
    double v = (t.q & 1) ? t.c : t.s;
    if (t.q & 2) v = -v;
    return (float)v;
}
__device__ __forceinline__ float cos_(float x) {
    SinCos t = sincos_core(x);
    double v = (t.q & 1) ? t.s : t.c;
    if ((t.q + 1) & 2) v = -v;
    return (float)v;
}
// sin and cos of the same angle from ONE argument reduction and ONE pair of polynomials: the values are those of
// sin_(x) and cos_(x) bit for bit (same core, same selection).  Written out because KD() is volatile, which keeps
// the compiler from merging the two cores by itself.
__device__ __forceinline__ void sincos_(float x, float& s, float& c) {
    SinCos t = sincos_core(x);
    double vs = (t.q & 1) ? t.c : t.s;
    if (t.q & 2) vs = -vs;
    double vc = (t.q & 1) ? t.s : t.c;
    if ((t.q + 1) & 2) vc = -vc;
    s = (float)vs;
    c = (float)vc;
}
__device__ __forceinline__ float acos_(float xf) {
    const double x = (double)xf;
    const double ax = __builtin_fabs(x);
    if (!(ax < 1.0)) {
        if (x != x) return xf;
        return (x > 0.0) ? 0.0f : (float)KD(3.14159265358979311600e+00);
    }
    const bool small = ax < 0.5;
    const double z = small ? x * x : (1.0 - ax) * 0.5;
    double p = __builtin_fma(z, KD(3.47933107596021167570e-05), KD(7.91534994289814532176e-04));
    p = __builtin_fma(z, p, KD(-4.00555345006794114027e-02));
    p = __builtin_fma(z, p, KD(2.01212532134862925881e-01));
    p = __builtin_fma(z, p, KD(-3.25565818622400915405e-01));
    p = __builtin_fma(z, p, KD(1.66666666666666657415e-01));
    p = p * z;
    double q = __builtin_fma(z, KD(7.70381505559019352791e-02), KD(-6.88283971605453293030e-01));
    q = __builtin_fma(z, q, KD(2.02094576023350569471e+00));
    q = __builtin_fma(z, q, KD(-2.40339491173441421878e+00));
    q = __builtin_fma(z, q, 1.0);
    const double R = p / q;
    double res;
    if (small) {
        res = KD(1.57079632679489655800e+00) - __builtin_fma(x, R, x);
    } else {
        const double s = __builtin_sqrt(z);
        const double t = 2.0 * __builtin_fma(s, R, s);
        res = (x > 0.0) ? t : KD(3.14159265358979311600e+00) - t;
    }
    return (float)res;
}

#endif      // RZ_MATH_FLAVOUR

// FS:188-190
#if defined(RZ_RAND_NOINLINE) && RZ_RAND_NOINLINE
static __device__ __attribute__((noinline)) float rand_(v2 uv) {
#else
__device__ __forceinline__ float rand_(v2 uv) {
#endif
    float d = uv.x * 12.9898f + uv.y * 78.233f;
    return fract_(sin_(d) * 43758.5453f);
}

}  // namespace rz
