// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ on gfx950 in PatchMatch's OWN access shape.
// Two kernels over a 2 GiB buffer (8x the 256 MiB Infinity Cache, touched once each => every byte comes from HBM, the byte count is known):
//   k_stream : every lane reads 16 B, fully coalesced (the shape the guide's x2 correction was measured on; k_normalize in the pipeline)
//   k_pm_rows: every 8-lane group reads one candidate-tile ROW as k_pm_step<1,...> does at C = 64: 3 pixels x 256 B = 768 contiguous bytes at a pseudo-random
//              pixel, as 6 loads of 8 lanes x 16 B (128 contiguous bytes per group and load; a wave = 8 groups at 8 unrelated addresses). Triplets are disjoint
//              (a multiplicative permutation of the triplet index), so nothing is read twice.
// usage: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o c --output-format csv -- ./fetch_calib
// prints the known byte counts; scripts/probes/fetch_calib_report.py divides the counter values by them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ buf, size_t n4, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    for (size_t j = i; j < n4; j += (size_t)gridDim.x * 256) { const float4 v = buf[j]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
// ntrip triplets of 768 B in the buffer; group g reads triplet (g * MULT) mod ntrip (MULT odd and coprime to ntrip = a power of two => a permutation)
__global__ __launch_bounds__(256) void k_pm_rows(const float4* __restrict__ buf, unsigned ntrip_mask, unsigned ngroups, float* __restrict__ out) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x, g = t >> 3, l = t & 7;
    if (g >= ngroups) return;
    const unsigned trip = (g * 2654435761u) & ntrip_mask;
    const float4* row = buf + (size_t)trip * 48;                  // 768 B = 48 float4
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const float4 v = row[k * 8 + l]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    float4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    const size_t n4 = bytes / 16;
    // 1 GiB streamed (the first half), then 768 MiB of tile rows out of the SECOND half (cold: the memset's tail in the Infinity Cache is at most 256 MiB of it,
    // so a third pass reads rows from the first half again after the stream pass has flushed it) — three dispatches, each with a known byte count
    hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const float4*)buf, n4 / 2, out);
    CK(hipDeviceSynchronize());
    const unsigned ntrip = 1u << 20;                               // 2^20 triplets x 768 B = 768 MiB
    hipLaunchKernelGGL(k_pm_rows, dim3(ntrip * 8 / 256), dim3(256), 0, 0, (const float4*)(buf + n4 / 2), ntrip - 1, ntrip, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const float4*)(buf + n4 / 2), n4 / 2, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_pm_rows, dim3(ntrip * 8 / 256), dim3(256), 0, 0, (const float4*)buf, ntrip - 1, ntrip, out);
    CK(hipDeviceSynchronize());
    printf("{\"k_stream_bytes\": %zu, \"k_pm_rows_bytes\": %zu}\n", bytes / 2, (size_t)ntrip * 768);
    return 0;
}
