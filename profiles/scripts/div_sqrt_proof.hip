// div_sqrt_proof.hip -- short instruction sequences for the correctly rounded square root and quotient of binary32, checked on
// the GPU against the compiler's IEEE expansions (v_sqrt / v_rcp + the full fix-up sequences, -fhip-fp32-correctly-rounded-divide-sqrt):
//   * square root: ALL 2^31 non-negative bit patterns (an exhaustive proof, like rcp_exhaustive.hip);
//   * quotient a / b with r = the correctly rounded reciprocal of b (rz::rcp_mid): q0 = a r; e = fma(-q0, b, a); q = fma(e, r, q0)
//     (Markstein's correction; correctly rounded by his theorem when q0 is within an ulp of a / b, which RN(a RN(1/b)) is not
//     PROVEN to be -- hence the sweep): 2^33 random operand pairs, every mantissa of b against 64 numerators, every mantissa of a
//     against 64 divisors, and the exponent boundaries of the admitted range.  Not exhaustive (2^64 pairs): a measured claim.
// The PRODUCT lines check rz_device_math.h's own functions under their own admission tests (the three-numerator test of div3 /
// normalize included: its admitted set is compared with the scalar one pattern by pattern); tests/test_div_sqrt_gpu.py
// builds and runs this file on the GPU of the test run and asserts on them.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rayzen_amd/csrc/hip -I include -o /tmp/div_sqrt_proof profiles/scripts/div_sqrt_proof.hip
#include <hip/hip_runtime.h>
#include "rz_device_math.h"
#include <cstdio>
#include <cstring>

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {      // splitmix64
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}

// ---- square root candidates
template <int CAND> __device__ __forceinline__ float sqrt_cand(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);                 // v_sqrt_f32, 1 ulp
    if (CAND == 0) {                                           // one residual step with h = 0.5 / s from v_rcp
        const float h = 0.5f * __builtin_amdgcn_rcpf(s);
        const float e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, h, s);
    }
    if (CAND == 1) {                                           // ... with h = 0.5 * v_rsq(x)
        const float h = 0.5f * __builtin_amdgcn_rsqf(x);
        const float e = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(e, h, s);
    }
    return s;                                                  // CAND 2: the bare instruction
}
struct Res { unsigned long long tried, bad; unsigned lo, hi; };
template <int CAND> __global__ void sweep_sqrt(Res* out, unsigned loBits, unsigned hiBits) {      // patterns in [loBits, hiBits]
    unsigned long long bad = 0, tried = 0; unsigned lo = 0xffffffffu, hi = 0;
    for (unsigned long long i = (unsigned long long)loBits + blockIdx.x * blockDim.x + threadIdx.x; i <= hiBits; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        ++tried;
        if (__float_as_uint(__builtin_sqrtf(x)) != __float_as_uint(sqrt_cand<CAND>(x))) { ++bad; lo = (unsigned)i < lo ? (unsigned)i : lo; hi = (unsigned)i > hi ? (unsigned)i : hi; }
    }
    atomicAdd(&out->tried, tried); atomicAdd(&out->bad, bad); atomicMin(&out->lo, lo); atomicMax(&out->hi, hi);
}
__global__ void sweep_sqrt_product(unsigned long long* out) {          // out[0] admitted, out[1] mismatches
    unsigned long long admitted = 0, bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        if (!rz::sqrt_mid_ok(x)) continue;
        ++admitted;
        if (__float_as_uint(__builtin_sqrtf(x)) != __float_as_uint(rz::sqrt_mid(x))) ++bad;
    }
    atomicAdd(&out[0], admitted); atomicAdd(&out[1], bad);
}

// ---- quotient candidates
template <int CAND> __device__ __forceinline__ float div_cand(float a, float b) {
    const float r = rz::rcp_mid(b);
    const float q0 = a * r;
    const float e = __builtin_fmaf(-q0, b, a);
    const float q1 = __builtin_fmaf(e, r, q0);
    if (CAND == 0) return q1;
    const float e1 = __builtin_fmaf(-q1, b, a);                // CAND 1: a second correction
    return __builtin_fmaf(e1, r, q1);
}
// operand from 64 random bits: sign, exponent in [expLo, expHi] (biased), mantissa; special mantissas now and then
__device__ __forceinline__ float operand(unsigned long long z, int expLo, int expHi) {
    const unsigned sign = (unsigned)(z >> 63) << 31;
    const unsigned e = (unsigned)expLo + (unsigned)((z >> 40) % (unsigned)(expHi - expLo + 1));
    unsigned m = (unsigned)z & 0x7fffffu;
    const unsigned sel = (unsigned)(z >> 24) & 15u;
    if (sel == 0) m = 0; else if (sel == 1) m = 0x7fffffu; else if (sel == 2) m &= 0xffu; else if (sel == 3) m |= 0x7fff00u; else if (sel == 4) m = 1u << ((z >> 28) % 23);
    return __uint_as_float(sign | (e << 23) | m);
}
template <int CAND> __global__ void sweep_div_random(Res* out, unsigned long long n, unsigned long long seed, int expLo, int expHi) {
    unsigned long long bad = 0, tried = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = operand(mix(seed + 2 * i), expLo, expHi), b = operand(mix(seed + 2 * i + 1), expLo, expHi);
        ++tried;
        if (__float_as_uint(a / b) != __float_as_uint(div_cand<CAND>(a, b))) ++bad;
    }
    atomicAdd(&out->tried, tried); atomicAdd(&out->bad, bad);
}
// every mantissa of one operand against 64 values of the other (which: 0 = all mantissas of b, 1 = all mantissas of a)
template <int CAND> __global__ void sweep_div_mantissa(Res* out, int which, unsigned long long seed) {
    unsigned long long bad = 0, tried = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (64ull << 23); i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned m = (unsigned)i & 0x7fffffu, k = (unsigned)(i >> 23);
        const float full = __uint_as_float((127u << 23) | m);
        const float other = k < 8 ? __uint_as_float((127u << 23) | (k == 0 ? 0u : k == 1 ? 0x7fffffu : k == 2 ? 0x400000u : k == 3 ? 1u : k == 4 ? 0x7ffffeu : k == 5 ? 0x555555u : k == 6 ? 0x2aaaaau : 0x3fffffu))
                                  : operand(mix(seed + k), 100, 154);
        const float a = which == 0 ? other : full, b = which == 0 ? full : other;
        ++tried;
        if (__float_as_uint(a / b) != __float_as_uint(div_cand<CAND>(a, b))) ++bad;
    }
    atomicAdd(&out->tried, tried); atomicAdd(&out->bad, bad);
}
__global__ void sweep_div_product(unsigned long long* out, unsigned long long n, unsigned long long seed) {       // out[0] admitted pairs, out[1] mismatches, out[2] rejected
    unsigned long long admitted = 0, bad = 0, rejected = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        // exponents over the WHOLE range, zeros, infinities and NaNs included: the admission test has to keep the bad ones out
        float a = operand(mix(seed + 2 * i), 0, 255), b = operand(mix(seed + 2 * i + 1), 0, 255);
        if ((i & 1023u) == 0) a = 0.0f;
        if ((i & 1023u) == 1) a = -0.0f;
        if (!(rz::div_mid_num_ok(a) && rz::div_mid_den_ok(b))) { ++rejected; continue; }
        ++admitted;
        if (__float_as_uint(a / b) != __float_as_uint(rz::div_mid(a, b, rz::rcp_mid(b)))) ++bad;
    }
    atomicAdd(&out[0], admitted); atomicAdd(&out[1], bad); atomicAdd(&out[2], rejected);
}

// The admission test the product SHIPS for three numerators at once (div3, normalize): (1) for every one of the 2^32 bit patterns
// v, in each of the three positions beside two harmless components, div_mid_num3_ok agrees with div_mid_num_ok(v) -- the admitted
// sets are equal (round 3's version admitted positive magnitudes below 2^-60: ADVICE r3); (2) random vectors with components over
// the WHOLE range (zeros of both signs, denormals, infinities, NaNs): whatever num3_ok admits, div_mid divides correctly.
__global__ void sweep_num3_sets(unsigned long long* out) {            // out[0] patterns tried, out[1] disagreements, out[2] admitted
    unsigned long long bad = 0, adm = 0, tried = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x) {
        const float v = __uint_as_float((unsigned)i);
        const bool one = rz::div_mid_num_ok(v);
        const bool x = rz::div_mid_num3_ok(rz::mk3(v, 1.0f, -3.5f)), y = rz::div_mid_num3_ok(rz::mk3(0.0f, v, 0x1p-60f)), z = rz::div_mid_num3_ok(rz::mk3(0x1p60f, -1.0f, v));
        ++tried; adm += one ? 1 : 0;
        if (x != one || y != one || z != one) ++bad;
    }
    atomicAdd(&out[0], tried); atomicAdd(&out[1], bad); atomicAdd(&out[2], adm);
}
__global__ void sweep_div3_product(unsigned long long* out, unsigned long long n, unsigned long long seed) {     // out[0] admitted vectors, out[1] mismatching components, out[2] rejected
    unsigned long long admitted = 0, bad = 0, rejected = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        float c[3];
        for (int k = 0; k < 3; ++k) {
            const unsigned long long z = mix(seed + 4 * i + k);
            const unsigned sel = (unsigned)(z >> 56) & 7u;         // a third of the components from the edges of the admitted set
            c[k] = sel == 0 ? 0.0f : sel == 1 ? -0.0f : sel == 2 ? operand(z, 0, 70) : operand(z, 60, 194);
        }
        const float b = operand(mix(seed + 4 * i + 3), 60, 194);
        const rz::v3 a = rz::mk3(c[0], c[1], c[2]);
        if (!(rz::div_mid_num3_ok(a) && rz::div_mid_den_ok(b))) { ++rejected; continue; }
        ++admitted;
        const float r = rz::rcp_mid(b);
        for (int k = 0; k < 3; ++k) if (__float_as_uint(c[k] / b) != __float_as_uint(rz::div_mid(c[k], b, r))) ++bad;
    }
    atomicAdd(&out[0], admitted); atomicAdd(&out[1], bad); atomicAdd(&out[2], rejected);
}

int main(int argc, char** argv) {
    const unsigned long long nRandom = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 33);
    Res* d; hipMalloc(&d, sizeof(Res));
    auto run = [&](const char* name, auto launch) {
        Res h{0, 0, 0xffffffffu, 0}; hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
        launch();
        hipDeviceSynchronize(); hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
        if (h.bad && h.lo != 0xffffffffu) { float flo, fhi; std::memcpy(&flo, &h.lo, 4); std::memcpy(&fhi, &h.hi, 4);
            printf("%-64s tried %llu mismatches %llu, x in [%.9g (0x%08x), %.9g (0x%08x)]\n", name, h.tried, h.bad, flo, h.lo, fhi, h.hi); }
        else printf("%-64s tried %llu mismatches %llu\n", name, h.tried, h.bad);
        return h.bad;
    };
    // square roots: normal inputs [2^-126, max finite]; then zero, denormals, infinity separately
    run("sqrt: v_sqrt + residual step, h = 0.5 v_rcp(s); normal x", [&] { sweep_sqrt<0><<<4096, 256>>>(d, 0x00800000u, 0x7f7fffffu); });
    run("sqrt: v_sqrt + residual step, h = 0.5 v_rsq(x); normal x", [&] { sweep_sqrt<1><<<4096, 256>>>(d, 0x00800000u, 0x7f7fffffu); });
    run("sqrt: bare v_sqrt_f32; normal x", [&] { sweep_sqrt<2><<<4096, 256>>>(d, 0x00800000u, 0x7f7fffffu); });
    run("sqrt: residual step (v_rcp); zero and denormal x", [&] { sweep_sqrt<0><<<4096, 256>>>(d, 0x00000000u, 0x007fffffu); });
    run("sqrt: residual step (v_rcp); x in [2^-100, 2^100]", [&] { sweep_sqrt<0><<<4096, 256>>>(d, 0x0d800000u, 0x71800000u); });
    // quotients
    run("div: one correction; 2^33 random pairs, exponents 2^-60..2^60", [&] { sweep_div_random<0><<<8192, 256>>>(d, nRandom, 12345ull, 67, 187); });
    run("div: two corrections; 2^31 random pairs, exponents 2^-60..2^60", [&] { sweep_div_random<1><<<8192, 256>>>(d, nRandom >> 2, 777ull, 67, 187); });
    run("div: one correction; every mantissa of b x 64 numerators", [&] { sweep_div_mantissa<0><<<8192, 256>>>(d, 0, 99ull); });
    run("div: one correction; every mantissa of a x 64 divisors", [&] { sweep_div_mantissa<0><<<8192, 256>>>(d, 1, 4242ull); });
    run("div: one correction; random pairs at the admitted exponent boundaries", [&] { sweep_div_random<0><<<8192, 256>>>(d, nRandom >> 4, 31337ull, 67, 69); });
    run("div: one correction; random pairs at the upper boundaries", [&] { sweep_div_random<0><<<8192, 256>>>(d, nRandom >> 4, 271828ull, 185, 187); });
    // the product's own functions
    unsigned long long* d2; hipMalloc(&d2, 24);
    unsigned long long h2[3];
    hipMemset(d2, 0, 24); sweep_sqrt_product<<<4096, 256>>>(d2); hipDeviceSynchronize(); hipMemcpy(h2, d2, 24, hipMemcpyDeviceToHost);
    printf("PRODUCT sqrt_mid: admitted %llu inputs, mismatches %llu\n", h2[0], h2[1]);
    const bool sqrtOk = h2[1] == 0 && h2[0] > (1ull << 30);
    hipMemset(d2, 0, 24); sweep_div_product<<<8192, 256>>>(d2, nRandom, 5551212ull); hipDeviceSynchronize(); hipMemcpy(h2, d2, 24, hipMemcpyDeviceToHost);
    printf("PRODUCT div_mid: admitted %llu pairs (rejected %llu), mismatches %llu\n", h2[0], h2[2], h2[1]);
    const bool divOk = h2[1] == 0 && h2[0] > nRandom / 16;
    hipMemset(d2, 0, 24); sweep_num3_sets<<<8192, 256>>>(d2); hipDeviceSynchronize(); hipMemcpy(h2, d2, 24, hipMemcpyDeviceToHost);
    printf("PRODUCT div_mid_num3_ok vs div_mid_num_ok: %llu bit patterns x 3 positions (admitted %llu), disagreements %llu\n", h2[0], h2[2], h2[1]);
    const bool setsOk = h2[1] == 0 && h2[0] == (1ull << 32);
    hipMemset(d2, 0, 24); sweep_div3_product<<<8192, 256>>>(d2, nRandom >> 2, 8675309ull); hipDeviceSynchronize(); hipMemcpy(h2, d2, 24, hipMemcpyDeviceToHost);
    printf("PRODUCT div3 admission: admitted %llu vectors (rejected %llu), mismatches %llu\n", h2[0], h2[2], h2[1]);
    const bool div3Ok = h2[1] == 0 && h2[0] > nRandom / 256;
    return sqrtOk && divOk && setsOk && div3Ok ? 0 : 1;
}
