// What a vector load costs the texture-address path of a gfx950 CU as a function of WHICH lanes are live.
// Every wave walks a pseudo-random chain of 64-byte records (4 x dwordx4 per step, as a descend step of the BLAS walk fetches a
// DevPair) out of a 4-MB table (L2-resident, larger than the 32-KB vector L1), with only the lanes of `mask` active.  The chain
// is dependent (next index from the loaded data), 16 waves per CU hide the latency; what is reported is wave-steps per second
// and ns per step per CU.  Build: hipcc --offload-arch=gfx950 -O3 -o ta_lanes ta_lanes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void walk(const float4* __restrict__ table, unsigned nRec, unsigned long long mask, int steps, unsigned* out) {
    const int lane = threadIdx.x & 63;
    unsigned cur = (blockIdx.x * 64u + lane) * 2654435761u % nRec;
    float acc = 0.0f;
    if ((mask >> lane) & 1ull) {
        for (int s = 0; s < steps; ++s) {
            const float4* p = table + (size_t)cur * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z;
            cur = __float_as_uint(d.w) % nRec;
        }
    }
    if (acc == 12345.0f) out[0] = cur;
}

int main() {
    const unsigned nRec = 1u << 16;        // 64 B each: 4 MB
    std::vector<float4> h((size_t)nRec * 4);
    unsigned x = 12345u;
    for (unsigned i = 0; i < nRec; ++i) {
        for (int k = 0; k < 4; ++k) h[(size_t)i * 4 + k] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
        x = x * 1664525u + 1013904223u;
        unsigned nxt = x % nRec;
        h[(size_t)i * 4 + 3].w = *reinterpret_cast<float*>(&nxt);
    }
    float4* d = nullptr; unsigned* out = nullptr;
    hipMalloc(&d, h.size() * sizeof(float4)); hipMalloc(&out, 64);
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int nCU = prop.multiProcessorCount;
    const int grid = nCU * 16, steps = 4000;
    struct { const char* name; unsigned long long m; } cases[] = {
        {"64 lanes", ~0ull},
        {"32 lanes, low half", 0xffffffffull},
        {"32 lanes, every other", 0x5555555555555555ull},
        {"16 lanes, lanes 0-15", 0xffffull},
        {"16 lanes, every 4th (one per quad)", 0x1111111111111111ull},
        {"16 lanes, 4 quads spread (0-3, 16-19, 32-35, 48-51)", 0x000f000f000f000full},
        {"8 lanes, lanes 0-7", 0xffull},
        {"8 lanes, every 8th", 0x0101010101010101ull},
        {"4 lanes, lanes 0-3", 0xfull},
        {"4 lanes, every 16th", 0x0001000100010001ull},
        {"1 lane", 0x1ull},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& cs : cases) {
        hipLaunchKernelGGL(walk, dim3(grid), dim3(64), 0, 0, d, nRec, cs.m, 200, out);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(walk, dim3(grid), dim3(64), 0, 0, d, nRec, cs.m, steps, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double waveSteps = (double)grid * steps;
        printf("%-52s %8.3f ms  %7.2f G wave-steps/s  %6.1f ns per wave-step per CU (= %5.1f cycles at 2.4 GHz; 4 loads per step)\n",
               cs.name, ms, waveSteps / ms * 1e-6, ms * 1e6 / (steps * 16.0), ms * 1e6 / (steps * 16.0) * 2.4);
    }
    return 0;
}
