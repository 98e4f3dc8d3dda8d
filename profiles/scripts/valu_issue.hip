// valu_issue.hip -- what does a gfx950 SIMD sustain, in cycles per wave64 instruction, for the instruction classes the
// render kernel is made of?  (task 2: calibrate the "issue limit" claim.)
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue profiles/scripts/valu_issue.hip && ./valu_issue
//
// One workgroup per CU (forced by a 100-KB LDS request), 4*W waves each, so every SIMD hosts exactly W waves that run
// the same loop of 16 INDEPENDENT instructions of one class (16 separate destination registers: no dependent-chain
// stall at W >= 1 for 4-cycle latencies).  Cycles are the shader clock read in-kernel with s_memtime around the loop
// (median over all waves), so DVFS does not enter.  Reported per class and W:
//     cyc/inst/SIMD = loop cycles of one wave / (instructions it issued * W)
// i.e. the SIMD's sustained cost of one wave-instruction with W waves competing.  The same binary is the target of a
// `rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass: that gives what
// SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles reads at saturation (run with an argument = one class index to profile only it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

enum { N_OPS = 66 };
static const char* kNames[N_OPS] = {
    "v_fma_f32",
    "v_mul_f32",
    "v_add_f32",
    "v_sub_f32 (sgpr operand)",
    "v_mul_f32 (sgpr operand)",
    "v_mul_f32 (literal operand)",
    "v_min_f32",
    "v_max_f32",
    "v_min3_f32",
    "v_max3_f32",
    "v_med3_f32",
    "v_pk_min_f32? (v_pk_max_f16 stand-in)",
    "v_cndmask_b32 vcc (vcc set once)",
    "v_cndmask_b32 e64 sgpr-pair mask",
    "v_cmp_lt_f32 vcc (e32)",
    "v_cmp_lt_f32 e64 -> sgpr pair (no reader)",
    "v_cmp e32 + v_cndmask vcc (select)",
    "v_cmpx_lt_f32 (writes exec)+restore",
    "v_mov_b32",
    "v_mov_b32 (sgpr src)",
    "v_add_u32",
    "v_lshlrev_b32",
    "v_and_b32 (inline const)",
    "v_and_b32 (literal)",
    "v_lshl_add_u32",
    "v_mad_u32_u24",
    "v_mul_lo_u32",
    "v_lshl_add_u64",
    "v_add_co_u32 + v_addc_co_u32 (64-bit add)",
    "v_pk_mul_f32",
    "v_pk_add_f32",
    "v_pk_add_f32 (sgpr pair, op_sel bcast)",
    "v_fma_f64",
    "v_fma_f64 (sgpr-pair operand)",
    "v_mul_f64",
    "v_add_f64",
    "v_cvt_f64_f32",
    "v_cvt_f32_f64",
    "v_cvt_f32_i32",
    "v_floor_f32",
    "v_rndne_f64",
    "v_rcp_f32",
    "v_sqrt_f32",
    "v_rsq_f32",
    "v_rcp_f64",
    "v_sqrt_f64 (v_rsq_f64)",
    "v_div_scale_f32",
    "v_div_fmas_f32",
    "v_div_fixup_f32",
    "a / b (IEEE f32 divide, hipcc sequence)",
    "1.0f / a (IEEE f32 reciprocal, hipcc)",
    "sqrtf(a) (IEEE f32, hipcc sequence)",
    "normalize(v3) = v / sqrt(dot) (5 per iter)",
    "f64 divide p / q (hipcc sequence)",
    "v_fma_f32 + s_add_u32 (1:1)",
    "s_add_u32 alone",
    "v_readfirstlane_b32 (no reader)",
    "v_readlane_b32 (no reader)",
    "v_mov_b32 dpp row_shr:1",
    "ds_bpermute_b32 (+wait)",
    "s_ballot: v_cmp e64 + s_cmp reader",
    "ds_write_b64 + ds_read_b64 (+wait)",
    "ds_read_b64 x16 then one wait",
    "s_load_dwordx16 (+wait), same line",
    "global_load_dwordx4 uniform addr (+wait)",
    "global_load_dwordx4 x4 uniform then wait"
};
// wave-instructions per "unit" (a unit is repeated 16x per loop iteration); negative: the whole 16-step body holds -n units of 1
static const int kInstPerUnit[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -5, 1, 2, 1, 1, 1, 1, 1, 2, 2, -16, 1, 1, -4};

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters, float seed, const float* gmem) {
    extern __shared__ unsigned long long lds[];
    float x[16]; double d[8]; f2 p[8]; unsigned u[16]; unsigned long long q[8]; unsigned s32[4] = {1, 2, 3, 4};
    const float one = 1.0f + seed * 1e-9f, eps = 1e-7f + seed * 1e-12f;
    for (int i = 0; i < 16; ++i) { x[i] = 1.0f + (float)threadIdx.x * 1e-3f + (float)i; u[i] = threadIdx.x * 7 + i; }
    for (int i = 0; i < 8; ++i) { d[i] = 1.0 + threadIdx.x * 1e-3 + i; p[i] = f2{1.0f + i, 2.0f + threadIdx.x}; q[i] = threadIdx.x + i; }
    const double done = 1.0 + seed * 1e-12;
    const f2 pm = {0.999999f, 1.000001f};
    const unsigned u0 = threadIdx.x | 1u;
    const unsigned long long q0 = threadIdx.x;
    // wave-uniform scalars (readfirstlane: the values are computed on the vector ALU)
    float sone; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sone) : "v"(one));
    double sdone = 1.000000000001; asm volatile("" : "+s"(sdone));      // a literal pinned to an SGPR pair (what KD() does in rz_device_math.h)
    f2 spm; spm.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pm.x))); spm.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(pm.y)));
    const unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)(unsigned)iters;    // scalar (kernel argument)
    const float* gptr = gmem; asm volatile("" : "+s"(gptr));
    const float* gvp = gmem; asm volatile("" : "+v"(gvp));
    asm volatile("s_mov_b64 vcc, %0" : : "s"(smask) : "vcc");
    const unsigned ldsOff = (threadIdx.x & 1023) * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "v"(eps)); }
            if (OP == 1) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(one)); }
            if (OP == 2) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(eps)); }
            if (OP == 3) { asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sone)); }
            if (OP == 4) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sone)); }
            if (OP == 5) { asm volatile("v_mul_f32 %0, 0x3f800001, %0" : "+v"(x[i])); }
            if (OP == 6) { asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(one)); }
            if (OP == 7) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(eps)); }
            if (OP == 8) { asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "v"(eps)); }
            if (OP == 9) { asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "v"(eps)); }
            if (OP == 10) { asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "v"(eps)); }
            if (OP == 11) { asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(one)); }
            if (OP == 12) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(one) : ); }
            if (OP == 13) { asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "s"(smask)); }
            if (OP == 14) { asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(one) : "vcc"); }
            if (OP == 15) { { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(x[i]), "v"(one)); } }
            if (OP == 16) { asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(one) : "vcc"); }
            if (OP == 17) { asm volatile("v_cmpx_lt_f32 %0, %1\n\ts_mov_b64 exec, -1" : : "v"(eps), "v"(x[i])); }
            if (OP == 18) { asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(one)); }
            if (OP == 19) { asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "s"(sone)); }
            if (OP == 20) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u0)); }
            if (OP == 21) { asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i])); }
            if (OP == 22) { asm volatile("v_and_b32 %0, 63, %0" : "+v"(u[i])); }
            if (OP == 23) { asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(u[i])); }
            if (OP == 24) { asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u0)); }
            if (OP == 25) { asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u0)); }
            if (OP == 26) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u0)); }
            if (OP == 27) { asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i & 7]) : "v"(q0)); }
            if (OP == 28) { asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(u[i]), "+v"(u[(i + 8) & 15]) : "v"(u0) : "vcc"); }
            if (OP == 29) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pm)); }
            if (OP == 30) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(pm)); }
            if (OP == 31) { asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[0,1]" : "+v"(p[i & 7]) : "s"(spm)); }
            if (OP == 32) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(done)); }
            if (OP == 33) { asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i & 7]) : "s"(sdone)); }
            if (OP == 34) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(done)); }
            if (OP == 35) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(done)); }
            if (OP == 36) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i & 7]) : "v"(x[i])); }
            if (OP == 37) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x[i]) : "v"(d[i & 7])); }
            if (OP == 38) { asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(x[i]) : "v"(u[i])); }
            if (OP == 39) { asm volatile("v_floor_f32 %0, %0" : "+v"(x[i])); }
            if (OP == 40) { asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i & 7])); }
            if (OP == 41) { asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])); }
            if (OP == 42) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[i])); }
            if (OP == 43) { asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i])); }
            if (OP == 44) { asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i & 7])); }
            if (OP == 45) { asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i & 7])); }
            if (OP == 46) { asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x[i]) : "v"(one) : "vcc"); }
            if (OP == 47) { asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(one) : "vcc"); }
            if (OP == 48) { asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(one)); }
            if (OP == 49) { { x[i] = x[i] / one; asm volatile("" : "+v"(x[i])); } }
            if (OP == 50) { { x[i] = 1.0f / x[i]; asm volatile("" : "+v"(x[i])); } }
            if (OP == 51) { { x[i] = __builtin_sqrtf(x[i]) + one; asm volatile("" : "+v"(x[i])); } }
            if (OP == 52) { if (i < 5) { float a = x[3 * i], b = x[3 * i + 1], c = x[(3 * i + 2) & 15];
                    const float len = __builtin_sqrtf((a * a + b * b) + c * c);
                    a = a / len; b = b / len; c = c / len;
                    x[3 * i] = a + one; x[3 * i + 1] = b + one; x[(3 * i + 2) & 15] = c + one;
                    asm volatile("" : "+v"(x[3 * i]), "+v"(x[3 * i + 1]), "+v"(x[(3 * i + 2) & 15])); } }
            if (OP == 53) { { d[i & 7] = d[i & 7] / done; asm volatile("" : "+v"(d[i & 7])); } }
            if (OP == 54) { { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(one), "v"(eps)); asm volatile("s_add_u32 %0, %0, 1" : "+s"(s32[i & 3]) : : "scc"); } }
            if (OP == 55) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(s32[i & 3]) : : "scc"); }
            if (OP == 56) { { unsigned s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(u[i])); } }
            if (OP == 57) { { unsigned s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(u[i])); } }
            if (OP == 58) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(one)); }
            if (OP == 59) { asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(u[i]) : "v"(ldsOff)); }
            if (OP == 60) { { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(x[i]), "v"(one)); asm volatile("s_cmp_eq_u64 %0, 0" : : "s"(m) : "scc"); } }
            if (OP == 61) { { asm volatile("ds_write_b64 %0, %1" : : "v"(ldsOff), "v"(d[i & 7]) : "memory"); asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d[(i + 1) & 7]) : "v"(ldsOff) : "memory"); } }
            if (OP == 62) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[i & 7]) : "v"(ldsOff), "i"(0) : "memory"); if (i == 15) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (OP == 63) { { f16v r; asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(gptr) : "memory"); } }
            if (OP == 64) { { float4 r; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(gvp) : "memory"); } }
            if (OP == 65) { if (i < 4) { float4 r; asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(gvp), "i"(0) : "memory"); if (i == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float sink = 0.0f;
    for (int i = 0; i < 16; ++i) sink += x[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) sink += (float)d[i] + p[i].x + p[i].y + (float)q[i];
    for (int i = 0; i < 4; ++i) sink += (float)s32[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[3 * wave] = t1 - t0; out[3 * wave + 1] = r1 - r0; out[3 * wave + 2] = (unsigned long long)sink; }
    (void)lds;
}

template <int OP>
static void launch(unsigned long long* out, int grid, int block, int iters, const float* gmem) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(block), 100 * 1024, 0, out, iters, 1.0f, gmem);
}
typedef void (*LaunchFn)(unsigned long long*, int, int, int, const float*);
template <int... I> struct Seq {};
template <int N, int... I> struct MakeSeq : MakeSeq<N - 1, N - 1, I...> {};
template <int... I> struct MakeSeq<0, I...> { typedef Seq<I...> type; };
template <int... I> static void fill(LaunchFn* t, Seq<I...>) { LaunchFn a[] = {&launch<I>...}; std::memcpy(t, a, sizeof a); }

int main(int argc, char** argv) {
    const int only = argc > 1 ? std::atoi(argv[1]) : -1;
    const int onlyW = argc > 2 ? std::atoi(argv[2]) : 0;
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    const int nCU = prop.multiProcessorCount;
    LaunchFn table[N_OPS]; fill(table, MakeSeq<N_OPS>::type());
    unsigned long long* out; (void)hipMalloc(&out, sizeof(unsigned long long) * 3 * nCU * 16);
    float* gmem; (void)hipMalloc(&gmem, 4096); (void)hipMemset(gmem, 0, 4096);
    std::vector<unsigned long long> h(3 * nCU * 16);
    printf("# %s, %d CUs; cycles per wave64 instruction per SIMD with W waves resident on the SIMD (shader clock, s_memtime)\n", prop.name, nCU);
    printf("%-44s %8s %8s %8s %8s   %s\n", "instruction class", "W=1", "W=2", "W=3", "W=4", "in-kernel clock (GHz) at W=4");
    for (int op = 0; op < N_OPS; ++op) {
        if (only >= 0 && op != only) continue;
        double cpi[4] = {0, 0, 0, 0}, ghz = 0;
        for (int W = 1; W <= 4; ++W) {
            if (onlyW && W != onlyW) continue;
            const int block = 256 * W;
            const int iters = 1024;
            for (int rep = 0; rep < 2; ++rep) {
                table[op](out, nCU, block, iters, gmem);
                (void)hipDeviceSynchronize();
            }
            const int nWaves = nCU * block / 64;
            (void)hipMemcpy(h.data(), out, sizeof(unsigned long long) * 3 * nWaves, hipMemcpyDeviceToHost);
            std::vector<double> cyc(nWaves), clk(nWaves);
            for (int w = 0; w < nWaves; ++w) { cyc[w] = (double)h[3 * w]; clk[w] = (double)h[3 * w] / ((double)h[3 * w + 1] * 10.0); }   // realtime = 100 MHz
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double perIter = kInstPerUnit[op] > 0 ? 16.0 * kInstPerUnit[op] : (double)-kInstPerUnit[op];
            cpi[W - 1] = cyc[nWaves / 2] / ((double)iters * perIter * W);
            ghz = clk[nWaves / 2];
        }
        printf("%-44s %8.2f %8.2f %8.2f %8.2f   %.2f\n", kNames[op], cpi[0], cpi[1], cpi[2], cpi[3], ghz);
        fflush(stdout);
    }
    (void)hipFree(out); (void)hipFree(gmem);
    return 0;
}
