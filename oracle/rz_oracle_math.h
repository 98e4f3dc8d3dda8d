/*
 * rz_oracle_math.h -- the definitions of the GLSL built-ins used by RayZen's
 * path tracer (shaders/fragment_shader.glsl, "FS").
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/.
 *
 * Rules (SURVEY.md section 8a "numerics"), confirmed against RayZen's own
 * shader run on Mesa llvmpipe (oracle/glref; llvmpipe's + - * / sqrt
 * inversesqrt are IEEE and unfused: probe_math.glsl):
 *   - all scene arithmetic is IEEE binary32, one rounding per operation,
 *     evaluated left to right, NO fused multiply-add (compile with
 *     -ffp-contract=off);
 *   - division and sqrt are correctly rounded;
 *   - min/max are IEEE-754 minNum/maxNum (the non-NaN operand wins), which
 *     is what GPU min/max instructions do;
 *   - sin/cos/acos -- left to the implementation by GLSL, and decisive: FS's
 *     hash is fract(sin(x) * 43758.5453) with x to 1e11 -- come in two
 *     FLAVOURS (rzo_math_flavour, at the end of this file):
 *       1 (the default, and the product's since round 5): Mesa llvmpipe's,
 *         replayed bit for bit -- the implementation RayZen's shader was RUN
 *         on here, so that oracle, product and the reference's own frames
 *         agree pixel by pixel at every bounce budget (tests/test_glref.py);
 *       0 (rounds 1-4): computed in binary64 from + - * / sqrt fma floor rint
 *         only and rounded once to binary32 (fdlibm k_sin / k_cos / e_acos
 *         coefficients, 3-term Cody-Waite split of pi/2 applied with fma):
 *         correctly rounded.  Agrees with the shader wherever no random
 *         number is drawn.
 */
#ifndef RZ_ORACLE_MATH_H
#define RZ_ORACLE_MATH_H

#include <math.h>

static inline float rzo_min(float a, float b) {     /* minNum */
    if (a != a) return b;
    if (b != b) return a;
    return (b < a) ? b : a;
}
static inline float rzo_max(float a, float b) {     /* maxNum */
    if (a != a) return b;
    if (b != b) return a;
    return (a < b) ? b : a;
}
static inline float rzo_clamp(float x, float lo, float hi) {
    return rzo_min(rzo_max(x, lo), hi);
}
static inline float rzo_mix(float a, float b, float t) {
    return a * (1.0f - t) + b * t;
}
static inline float rzo_fract(float x) { return x - floorf(x); }
static inline float rzo_pow2(float x) { return x * x; }
static inline float rzo_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }

/* x -> (r, quadrant): r = x - k*pi/2, k = rint(x*2/pi), quadrant = k mod 4. */
static inline double rzo_reduce_pio2(double x, int* quadrant) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double P2 = 6.07710050630396597660e-11;   /* next 33 bits */
    const double P3 = 2.02226624871116645580e-21;   /* next 33 bits */
    double k = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-k, P1, x);
    r = __builtin_fma(-k, P2, r);
    r = __builtin_fma(-k, P3, r);
    double q = k - 4.0 * __builtin_floor(k * 0.25);  /* exact, in {0,1,2,3} */
    *quadrant = (int)q;
    return r;
}
static inline double rzo_ksin(double r) {           /* |r| <= pi/4 (+slack) */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = __builtin_fma(z, S6, S5);
    p = __builtin_fma(z, p, S4);
    p = __builtin_fma(z, p, S3);
    p = __builtin_fma(z, p, S2);
    p = __builtin_fma(z, p, S1);
    return __builtin_fma(z * r, p, r);
}
static inline double rzo_kcos(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = __builtin_fma(z, C6, C5);
    p = __builtin_fma(z, p, C4);
    p = __builtin_fma(z, p, C3);
    p = __builtin_fma(z, p, C2);
    p = __builtin_fma(z, p, C1);
    double h = __builtin_fma(z, -0.5, 1.0);
    return __builtin_fma(z * z, p, h);
}
static inline float rzo_sin_pinned(float x) {
    int q; double r = rzo_reduce_pio2((double)x, &q);
    double s = rzo_ksin(r), c = rzo_kcos(r);
    double v = (q & 1) ? c : s;
    if (q & 2) v = -v;
    return (float)v;
}
static inline float rzo_cos_pinned(float x) {
    int q; double r = rzo_reduce_pio2((double)x, &q);
    double s = rzo_ksin(r), c = rzo_kcos(r);
    double v = (q & 1) ? s : c;
    if (((q + 1) & 2) != 0) v = -v;
    return (float)v;
}
static inline float rzo_acos_pinned(float xf) {
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double PIO2 = 1.57079632679489655800e+00, PI = 3.14159265358979311600e+00;
    double x = (double)xf;
    double ax = (x < 0.0) ? -x : x;
    if (!(ax < 1.0)) {                       /* |x| >= 1 or NaN */
        if (x != x) return xf;
        return (x > 0.0) ? 0.0f : (float)PI;
    }
    int small = ax < 0.5;
    double z = small ? x * x : (1.0 - ax) * 0.5;
    double p = __builtin_fma(z, pS5, pS4);
    p = __builtin_fma(z, p, pS3);
    p = __builtin_fma(z, p, pS2);
    p = __builtin_fma(z, p, pS1);
    p = __builtin_fma(z, p, pS0);
    p = p * z;
    double qd = __builtin_fma(z, qS4, qS3);
    qd = __builtin_fma(z, qd, qS2);
    qd = __builtin_fma(z, qd, qS1);
    qd = __builtin_fma(z, qd, 1.0);
    double R = p / qd;
    double res;
    if (small) {
        res = PIO2 - __builtin_fma(x, R, x);
    } else {
        double s = __builtin_sqrt(z);
        double t = 2.0 * __builtin_fma(s, R, s);
        res = (x > 0.0) ? t : PI - t;
    }
    return (float)res;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * MATH FLAVOUR 1 (the default; the product's RZ_MATH_FLAVOUR=1): sin / cos / acos as the GL implementation computes them that
 * runs RayZen's own shader in oracle/glref (Mesa 23.2 llvmpipe on an x86-64 with FMA).  With it the oracle, the product and
 * RayZen's shader draw the SAME random numbers (FS:188-190: fract(sin(x) * 43758.5453), x up to 1e11 -- where an
 * implementation's range reduction decides every bit), so frames of any bounce budget agree pixel by pixel
 * (tests/test_glref.py), not only in distribution.  Flavour 0 -- the binary64 definitions above, rounds 1-4's specification and
 * still a build option of the product -- differs in these three functions and nowhere else.
 *   sin, cos: gallivm's lp_build_sin_or_cos -- Cephes' sinf / cosf as in J. Pommier's sse_mathfun: octant j = (trunc(|x| * 4/pi)
 *     + 1) & ~1 (x86 cvttps2dq: 0x80000000 out of range), three-constant Cody-Waite reduction and both polynomials with
 *     FUSED multiply-adds (llvm.fmuladd on a machine with FMA), the result clamped to [-1, 1];
 *   acos: Mesa's GLSL built-in, pi/2 - sign(x) (pi/2 - sqrt(1 - |x|) (pi/2 + |x| (pi/4 - 1 + |x| (p0 + |x| p1)))) in unfused binary32.
 * Both were checked BIT FOR BIT against tables made by llvmpipe itself (oracle/glref/probe_math.glsl,
 * tests/test_glref.py::test_flavour_1_is_llvmpipes_sin_cos_acos): 23 000 + 40 000 arguments, all ranges, all equal. */
extern int rzo_math_flavour;

static inline float rzo_lp_sincos(float xin, int want_cos) {
    union { float f; unsigned u; } v, r;
    v.f = xin;
    unsigned sign_bit = want_cos ? 0u : (v.u & 0x80000000u);
    v.u &= 0x7fffffffu;
    const float x = v.f;
    const float scale_y = x * 1.27323954473516f;
    int j = (scale_y >= 2147483648.0f || scale_y != scale_y) ? (int)0x80000000 : (int)scale_y;
    j = (int)(((unsigned)j + 1u) & ~1u);
    const float y = (float)j;
    unsigned swap, poly;
    if (want_cos) {
        const unsigned e = (unsigned)j - 2u;
        swap = ((~e) & 4u) << 29;
        poly = (e & 2u) == 0u;
    } else {
        swap = ((unsigned)j & 4u) << 29;
        poly = ((unsigned)j & 2u) == 0u;
    }
    sign_bit ^= swap;
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
    float s = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    s = __builtin_fmaf(s, z, -1.6666654611E-1f);
    s = s * z;
    s = __builtin_fmaf(s, xr, xr);
    r.f = poly ? s : c;
    r.u ^= sign_bit;
    r.f = (r.f < 1.0f) ? r.f : 1.0f;        /* x86 minps / maxps: a NaN (x^2 overflowed) yields the bound */
    r.f = (r.f > -1.0f) ? r.f : -1.0f;
    return r.f;
}
static inline float rzo_lp_acos(float x) {
    const float PIO2 = 1.57079632679489661923f, PIO4M1 = 0.78539816339744830962f - 1.0f;
    const float ax = fabsf(x);
    float e = 0.08132463f + ax * -0.02363318f;
    e = PIO4M1 + ax * e;
    e = PIO2 + ax * e;
    const float sgn = (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f);
    const float as = sgn * (PIO2 - sqrtf(1.0f - ax) * e);
    return PIO2 - as;
}
static inline float rzo_sin(float x) { return rzo_math_flavour == 1 ? rzo_lp_sincos(x, 0) : rzo_sin_pinned(x); }
static inline float rzo_cos(float x) { return rzo_math_flavour == 1 ? rzo_lp_sincos(x, 1) : rzo_cos_pinned(x); }
static inline float rzo_acos(float x) { return rzo_math_flavour == 1 ? rzo_lp_acos(x) : rzo_acos_pinned(x); }

#endif
