// Issue cost of packed FP32 VALU ops on gfx950 relative to plain ones: 4 waves per SIMD, 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 a[8]; float s[16];
    for (int i = 0; i < 8; ++i) { a[i] = f2{1.0f + threadIdx.x * 1e-3f + i, 2.0f + i}; }
    for (int i = 0; i < 16; ++i) s[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const f2 m = {0.999999f, 1.000001f}, c = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 2) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[2 * i]) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[2 * i + 1]) : "v"(m.y)); }
            if (OP == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[2 * i]) : "v"(c.x)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[2 * i + 1]) : "v"(c.y)); }
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* o; hipMalloc(&o, 1 << 22);
    const char* names[] = {"v_pk_mul_f32 (2 mul)", "v_pk_add_f32 (2 add)", "2 x v_mul_f32", "2 x v_add_f32", "v_pk_mul_f32 op_sel", "v_pk_fma_f32"};
    const int iters = 20000;
    for (int op = 0; op < 6; ++op)
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            switch (op) {   // 256 CUs x 4 SIMDs x 4 waves = 1024 blocks of 256 threads
                case 0: k<0><<<1024, 256>>>(o, iters); break; case 1: k<1><<<1024, 256>>>(o, iters); break;
                case 2: k<2><<<1024, 256>>>(o, iters); break; case 3: k<3><<<1024, 256>>>(o, iters); break;
                case 4: k<4><<<1024, 256>>>(o, iters); break; case 5: k<5><<<1024, 256>>>(o, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: 4 waves x iters x 8 groups; time per group of (2 f32 ops) per wave in cycles at 2.4 GHz
            if (rep) printf("%-24s %.3f ms   %.2f cycles per pair of f32 ops per wave (at 2.4 GHz, 4 waves/SIMD)\n", names[op], ms, ms * 1e-3 * 2.4e9 / (4.0 * iters * 8.0));
        }
    return 0;
}
