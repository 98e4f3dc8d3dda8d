// Issue cost of f64 VALU ops on gfx950, one wave per SIMD, 8 independent accumulators (no dependency stalls).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double* out, float* outf, int iters, long long* cyc) {
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
    const double m = 0.999999, c = 1e-7;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], m, c);
            if (OP == 1) a[i] = a[i] * m;
            if (OP == 2) a[i] = a[i] + c;
            if (OP == 3) f[i] = __builtin_fmaf(f[i], 0.999999f, 1e-7f);
            if (OP == 4) a[i] = __builtin_rint(a[i] * m);
            if (OP == 5) a[i] = (double)(float)a[i] ;
            if (OP == 6) a[i] = __builtin_floor(a[i]) + c;
        }
    }
    const long long t1 = clock64();
    double s = 0; float sf = 0;
    for (int i = 0; i < 8; ++i) { s += a[i]; sf += f[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; outf[blockIdx.x * blockDim.x + threadIdx.x] = sf;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double* o; float* of; long long* cy; hipMalloc(&o, 1 << 20); hipMalloc(&of, 1 << 20); hipMalloc(&cy, 8);
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "mul+rndne f64", "cvt f64->f32->f64", "floor+add f64"};
    const int iters = 20000;
    for (int op = 0; op < 7; ++op) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            // 1024 workgroups of one wave: 1 wave per SIMD on 256 CUs
            switch (op) {
                case 0: k<0><<<1024, 64>>>(o, of, iters, cy); break; case 1: k<1><<<1024, 64>>>(o, of, iters, cy); break;
                case 2: k<2><<<1024, 64>>>(o, of, iters, cy); break; case 3: k<3><<<1024, 64>>>(o, of, iters, cy); break;
                case 4: k<4><<<1024, 64>>>(o, of, iters, cy); break; case 5: k<5><<<1024, 64>>>(o, of, iters, cy); break;
                case 6: k<6><<<1024, 64>>>(o, of, iters, cy); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cy, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-20s %.3f ms  %.2f ns per wave-op (8 per iter)  clock64 ticks/op %.2f\n", names[op], ms, ms * 1e6 / (iters * 8.0), (double)c / (iters * 8.0));
        }
    }
    return 0;
}
