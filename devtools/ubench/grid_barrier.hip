// Grid-barrier microbenchmark (round 6): what does a device-wide barrier inside a 256-block kernel cost on MI355X, and do
// cross-XCD hand-overs through system-scope (sc0 sc1) stores / loads arrive?  One block per CU (112 KB of LDS), 8 waves.
//   hipcc --offload-arch=gfx950 -O3 -o devtools/ubench/grid_barrier devtools/ubench/grid_barrier.hip && devtools/ubench/grid_barrier
// Each block: spin `work` iterations of MFMAs (skewed by block id so that blocks arrive at different times), write one entry,
// barrier, read the entries (the entry check of this harness is NOT trusted: hipcc folded its second store operand; only
// the timing columns and the time-out / generation counters of profiles/r06_grid_barrier.txt are used).  Modes: 0 = no barrier (baseline, entries
// not checked), 1 = sense-reversing barrier on two global words (agent-scope atomics), spin with s_sleep and a timeout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bar { unsigned count, gen, timeout, pad; };

// MODE 1: flat barrier, one counter, every waiter polls the generation word (s_sleep 8)
// MODE 2: flat barrier, waiters poll with s_sleep 64
// MODE 3: two levels: per-XCD counter / generation (XCD = blockIdx & 7; L2-local traffic), one leader per XCD arrives at the
//         global counter and polls the global generation, then releases its XCD
#ifndef FENCE
#define FENCE 0
#endif
#ifndef AUX
#define AUX 17
#endif
template <int MODE>
__device__ __forceinline__ bool grid_barrier(Bar* bar, unsigned nblocks) {
    __shared__ unsigned ok_s;
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // my stores (entries) visible device-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ok = 1, it = 0;
        if (MODE <= 2) {
            const unsigned gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned prev = __hip_atomic_fetch_add(&bar->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == nblocks - 1) {
                __hip_atomic_store(&bar->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&bar->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                    if (MODE == 1) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
                    if (++it > (1u << 22)) { ok = 0; __hip_atomic_store(&bar->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
        } else {
            Bar* xb = bar + 1 + (blockIdx.x & 7);                // this XCD's counter / generation
            const unsigned per = nblocks >> 3;
            const unsigned xgen = __hip_atomic_load(&xb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned prev = __hip_atomic_fetch_add(&xb->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == per - 1) {                               // this XCD's last arriver = its leader
                __hip_atomic_store(&xb->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned gen = __hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned p2 = __hip_atomic_fetch_add(&bar->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p2 == 7) {
                    __hip_atomic_store(&bar->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_add(&bar->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(&bar->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++it > (1u << 22)) { ok = 0; __hip_atomic_store(&bar->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
                __hip_atomic_fetch_add(&xb->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(&xb->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xgen) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++it > (1u << 22)) { ok = 0; __hip_atomic_store(&bar->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // drop stale lines before reading the others' entries
    return ok_s != 0;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(Bar* bar, f32x4* entries, float* out, int work, int stamp) {
    float* dbg = out + 300;
    __shared__ half8 lds[7168];                    // 112 KB: one block per CU
    const int tid = threadIdx.x;
    lds[tid] = half8{(_Float16)tid, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    half8 a = lds[tid], b = lds[(tid + 1) & 511];
    const int n = work + (blockIdx.x & 15) * (work / 64);     // skew: up to +25 %
    for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    // one entry per block, written so that another XCD can read it inside this kernel: system-scope store
    if (tid == 0) {
        const f32x4 e = {(float)blockIdx.x, (float)stamp, s, 1.0f};
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)entries, 0, gridDim.x * 16u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e.x), rs, blockIdx.x * 16u, 0, AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e.y), rs, blockIdx.x * 16u + 4, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bool ok = true;
    if (MODE >= 1) ok = grid_barrier<MODE>(bar, gridDim.x);
    int bad = 0;
    if (MODE >= 1) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)entries, 0, gridDim.x * 16u, 0x00020000);
        for (int e = tid; e < (int)gridDim.x; e += 512) {
            const unsigned x0 = __builtin_amdgcn_raw_buffer_load_b32(rs, e * 16u, 0, AUX);
            const unsigned x1 = __builtin_amdgcn_raw_buffer_load_b32(rs, e * 16u + 4, 0, AUX);
            if (__builtin_bit_cast(float, x0) != (float)e || __builtin_bit_cast(float, x1) != (float)stamp) { bad++; if (blockIdx.x == 3 && e == 5) { dbg[0] = __builtin_bit_cast(float, x0); dbg[1] = __builtin_bit_cast(float, x1); dbg[2] = (float)stamp; dbg[3] = (float)e; } }
        }
    }
    if (bad || !ok) atomicAdd(&out[1], 1.0f);
    if (tid == 0) out[2 + blockIdx.x] = s;
}

int main() {
    const int nb = 256;
    Bar* bar; f32x4* ent; float* out;
    hipMalloc(&bar, 9 * sizeof(Bar)); hipMemset(bar, 0, 9 * sizeof(Bar));
    hipMalloc(&ent, nb * 16); hipMemset(ent, 0, nb * 16);
    hipMalloc(&out, 400 * 4); hipMemset(out, 0, 400 * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {50, 400, 1600}) {
        for (int mode = 0; mode < 4; ++mode) {
            // capture 200 dependent launches in a graph, replay 5 times
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int i = 0; i < 200; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), 0, st, bar, ent, out, work, i);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), 0, st, bar, ent, out, work, i);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(512), 0, st, bar, ent, out, work, i);
                else hipLaunchKernelGGL(k<3>, dim3(nb), dim3(512), 0, st, bar, ent, out, work, i);
            }
            hipStreamEndCapture(st, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphLaunch(ge, st); hipStreamSynchronize(st);
            hipMemset(out, 0, 8);
            hipEventRecord(e0, st);
            for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float h[2]; Bar hb; float dbg[4];
            hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(dbg, out + 300, 16, hipMemcpyDeviceToHost); hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost);
            printf("work %5d mode %d: %.2f us per launch  (bad blocks %.0f, timeouts %u, gen %u) dbg x0 %g x1 %g stamp %g e %g\n", work, mode, ms * 1e3 / 1000, h[1], hb.timeout, hb.gen, dbg[0], dbg[1], dbg[2], dbg[3]);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
