// rcp_exhaustive.hip -- which short instruction sequences give the correctly rounded 1/x (== the IEEE division 1.0f / x
// that hipcc expands to 11 instructions) for EVERY binary32 x?  All 2^32 bit patterns are tried on the GPU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/rcp_exhaustive profiles/scripts/rcp_exhaustive.hip && /tmp/rcp_exhaustive
// Candidates (r0 = v_rcp_f32(x), 1 ulp):
//   A: e = fma(-x, r0, 1); r = fma(e, r0, r0)                                   + v_div_fixup_f32(r, x, 1)
//   B: A, then e = fma(-x, r, 1); r = fma(e, r, r)                               + fixup
//   C: A, then q = r; e = fma(-x, q, 1); r = fma(e, r, q)   (residual of the QUOTIENT, as the library sequence does) + fixup
// Output: per candidate, the number of mismatching x and the range of |x| in which they lie.
// The last line checks the PRODUCT's own functions (rz_device_math.h): for every x that rcp_mid_ok() admits, rcp_mid(x)
// must equal 1.0f / x bit for bit.  tests/test_rcp_gpu.py builds and runs this file on the GPU of the test run.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I rayzen_amd/csrc/hip -o /tmp/rcp_exhaustive profiles/scripts/rcp_exhaustive.hip
#include <hip/hip_runtime.h>
#include "rz_device_math.h"
#include <cstdio>
#include <cstring>
#include <cmath>

__device__ __forceinline__ float rcp_hw(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fixup(float q, float den, float num) { return __builtin_amdgcn_div_fixupf(q, den, num); }

template <int CAND> __device__ __forceinline__ float cand(float x) {
    const float r0 = rcp_hw(x);
    float e = __builtin_fmaf(-x, r0, 1.0f);
    float r = __builtin_fmaf(e, r0, r0);
    if (CAND == 1) { e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r); }
    if (CAND == 2) { const float q = r; e = __builtin_fmaf(-x, q, 1.0f); r = __builtin_fmaf(e, r, q); }
    if (CAND == 3) return r;                                    // A without the fixup
    return fixup(r, x, 1.0f);
}

struct Res { unsigned long long bad; unsigned lo, hi; unsigned long long badMid; };   // |x| bit patterns of the smallest / largest mismatch; mismatches with 2^-126 <= |x| <= 2^126

template <int CAND> __global__ void sweep(Res* out) {
    const unsigned long long n = 1ull << 32;
    unsigned long long bad = 0, badMid = 0; unsigned lo = 0xffffffffu, hi = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        const float want = 1.0f / x;                            // the IEEE division (11-instruction expansion)
        const float got = cand<CAND>(x);
        const bool same = __float_as_uint(want) == __float_as_uint(got);       // bit patterns, NaN payloads included
        if (!same) { ++bad; const unsigned a = (unsigned)i & 0x7fffffffu; lo = a < lo ? a : lo; hi = a > hi ? a : hi; if (a >= 0x00800000u && a <= 0x7e800000u) ++badMid; }
    }
    atomicAdd(&out->bad, bad);
    atomicAdd(&out->badMid, badMid);
    atomicMin(&out->lo, lo);
    atomicMax(&out->hi, hi);
}

__global__ void sweep_product(unsigned long long* out) {        // out[0] admitted inputs, out[1] mismatches among them
    const unsigned long long n = 1ull << 32;
    unsigned long long admitted = 0, bad = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((unsigned)i);
        if (!rz::rcp_mid_ok(x)) continue;
        ++admitted;
        if (__float_as_uint(1.0f / x) != __float_as_uint(rz::rcp_mid(x))) ++bad;
    }
    atomicAdd(&out[0], admitted);
    atomicAdd(&out[1], bad);
}

int main() {
    Res* d; hipMalloc(&d, sizeof(Res));
    const char* names[4] = {"A (one Newton step) + fixup", "B (two Newton steps) + fixup", "C (Newton + quotient residual) + fixup", "A without fixup"};
    for (int c = 0; c < 4; ++c) {
        Res h{0, 0xffffffffu, 0, 0}; hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
        if (c == 0) sweep<0><<<4096, 256>>>(d); else if (c == 1) sweep<1><<<4096, 256>>>(d); else if (c == 2) sweep<2><<<4096, 256>>>(d); else sweep<3><<<4096, 256>>>(d);
        hipDeviceSynchronize(); hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
        float flo, fhi; std::memcpy(&flo, &h.lo, 4); std::memcpy(&fhi, &h.hi, 4);
        if (h.bad) printf("%-42s mismatches %llu of 2^32, |x| in [%.9g (0x%08x), %.9g (0x%08x)]; with 2^-126 <= |x| <= 2^126: %llu\n", names[c], h.bad, flo, h.lo, fhi, h.hi, h.badMid);
        else printf("%-42s mismatches 0 of 2^32\n", names[c]);
    }
    unsigned long long* d2; hipMalloc(&d2, 16); hipMemset(d2, 0, 16);
    sweep_product<<<4096, 256>>>(d2);
    unsigned long long h2[2] = {0, 0};
    hipDeviceSynchronize(); hipMemcpy(h2, d2, 16, hipMemcpyDeviceToHost);
    printf("PRODUCT rcp_mid: admitted %llu inputs, mismatches %llu\n", h2[0], h2[1]);
    return h2[1] == 0 && h2[0] == 2ull * 253ull * (1ull << 23) - 2ull * ((1ull << 23) - 1ull) ? 0 : 1;
}
