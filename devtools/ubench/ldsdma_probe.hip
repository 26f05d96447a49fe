// Probe of buffer_load_dwordx4 ... lds on gfx950: lane -> LDS mapping and out-of-range lanes.
//   hipcc --offload-arch=gfx950 -O3 -o devtools/ubench/ldsdma_probe devtools/ubench/ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ void probe(const float* src, unsigned nbytes, float* out, int soff) {
    __shared__ float lds[2 * 64 * 4 + 64];
    for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += 64) lds[i] = -7.0f;     // stale marker
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    // lanes 0..59 read unit (63 - lane) (reversed), lanes 60..63 are out of range
    unsigned voff = lane < 60 ? (unsigned)(63 - lane) * 16u : 0xFFFFFFF0u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 16), 16, voff, soff, 0, 0);
    // second instruction with an immediate offset of 1024 bytes into LDS? (imm applies to the global side)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 16 + 256), 16, (unsigned)lane * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 64 * 4 + 64; i += 64) out[i] = lds[i];
}

int main() {
    const int n = 64 * 4 + 64;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, (2 * 64 * 4 + 64) * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, 64 * 16, o, 0);
    std::vector<float> r(2 * 64 * 4 + 64);
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("pad before: %g %g\n", r[0], r[15]);
    printf("instr1 lane0 unit: %g %g %g %g (expect 252..255)\n", r[16], r[17], r[18], r[19]);
    printf("instr1 lane1 unit: %g (expect 248)\n", r[20]);
    printf("instr1 lane59 unit: %g (expect 16)\n", r[16 + 59 * 4]);
    printf("instr1 lane60..63 (OOB): %g %g %g %g (0 = zero fill, -7 = skipped)\n", r[16 + 60 * 4], r[16 + 61 * 4], r[16 + 62 * 4], r[16 + 63 * 4 + 3]);
    printf("instr2 lane0: %g lane63: %g (expect 0, 252)\n", r[16 + 256], r[16 + 256 + 63 * 4]);
    printf("pad after: %g\n", r[16 + 512]);
    // soffset participates in the range check?
    probe<<<1, 64>>>(d, 64 * 16, o, 64);   // soffset 64 bytes: lanes reading units near the end go past num_records
    hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("soffset=64: lane0 (unit 63 + 4 units = past the end): %g (0 => soffset is range checked; 268 => not)\n", r[16]);
    printf("soffset=64: lane59 (unit 4+4=8): %g (expect 32)\n", r[16 + 59 * 4]);
    return 0;
}
