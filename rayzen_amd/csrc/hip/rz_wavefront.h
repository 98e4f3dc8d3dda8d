// rz_wavefront.h -- buffers of the queued pipeline (rz_wavefront.hip).  One slot per owned pixel
// (slot = localTile*64 + lane-in-tile); all arrays are SoA of float4 indexed by slot, so a wave's accesses are
// 1 KiB contiguous when the queue is in slot order (it starts that way and compaction preserves order per wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rz {

struct WFParams {
    float4* rayO;       // o.xyz
    float4* rayD;       // d.xyz
    float4* hit;        // world t, local t, triangle (bits, -1 = miss), instance (bits)
    float4* s0;         // throughput.xyz, currentIor
    float4* s1;         // seed.xy, samp (bits), bounce | mode<<12 | iter<<14 | light<<20 (bits)
    float4* s2;         // parked surface point hp.xyz, material (bits)
    float4* s3;         // its normal hn.xyz, shadow visibility
    float4* s4;         // arriving direction pdir.xyz, shadow distance travelled
    float4* s5;         // lighting accumulator lacc.xyz, shadow maxDist
    int* queue[2];      // ping-pong queues of slots
    int* counts;        // counts[r] = rays queued for round r   (zeroed per frame)
    int* cursors;       // cursors[r] = next unfetched entry of round r's queue
    int nSlots;
    int maxRounds;      // capacity of counts[] / cursors[] minus one
};

struct KParams;
size_t wf_trace_lds_bytes(const KParams& K);
void launch_wf_init(const KParams& K, const WFParams& Q, bool counted, hipStream_t s);
void launch_wf_round(const KParams& K, const WFParams& Q, int round, int traceBlocks, int shadeBlocks, bool counted,
                     hipStream_t s);
int wf_set_lds_limit(size_t bytes);

}  // namespace rz
