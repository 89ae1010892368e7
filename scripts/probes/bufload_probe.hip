// Probe of raw buffer load semantics on gfx950 (design input for the VGG conv B-operand loads): per-dword range check of a 12-byte load that straddles num_records,
// 32-bit wrap of a "negative" voffset, and whether soffset takes part in the range check.   hipcc --offload-arch=gfx950 -O2 bufload_probe.hip -o bufload_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__global__ void probe(const float* base, unsigned nbytes, const unsigned* voff, const unsigned* soff, float* out, int n) {
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, nbytes, 0x00020000);
    int i = threadIdx.x;
    if (i >= n) return;
    u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, voff[i], __builtin_amdgcn_readfirstlane(soff[0]), 0);
    out[3 * i] = __uint_as_float(v.x); out[3 * i + 1] = __uint_as_float(v.y); out[3 * i + 2] = __uint_as_float(v.z);
}
int main() {
    const int N = 1024;            // floats in the buffer; the descriptor covers only the first 256 (1024 bytes) of a 4096-byte allocation
    float h[N]; for (int i = 0; i < N; ++i) h[i] = 100.f + i;
    float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    float* base = d + 256;         // descriptor base in the middle: "negative" reads land in mapped memory if they are executed at all
    unsigned hv[8] = {0u, 4u, 1024u - 8u, 1024u - 4u, 1024u, 0xFFFFFFFCu, 0xFFFFFFF8u, 0xFFFFFFF4u};
    unsigned *dv, *ds; float* dout; hipMalloc(&dv, sizeof(hv)); hipMalloc(&ds, 4); hipMalloc(&dout, 8 * 3 * 4);
    hipMemcpy(dv, hv, sizeof(hv), hipMemcpyHostToDevice);
    for (unsigned so : {0u, 512u, 1024u, 2048u}) {
        hipMemcpy(ds, &so, 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, base, 1024u, dv, ds, dout, 8);
        float ho[24]; hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        printf("soffset %u (base value %.0f, first value past num_records %.0f):\n", so, h[256], h[256 + 256]);
        for (int i = 0; i < 8; ++i) printf("  voffset %08x -> %.0f %.0f %.0f\n", hv[i], ho[3 * i], ho[3 * i + 1], ho[3 * i + 2]);
    }
    return 0;
}
