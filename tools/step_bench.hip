// step_bench — what does ONE dependent traversal step cost on this chip, whatever the traversal does with it?
// A wave walks a chain of K steps; a step = one 128-byte node line per lane (7 x buffer_load_dwordx4; the same line
// for every lane, or a different one per lane), F dependent VALU instructions on the loaded data, D dependent LDS
// round trips, then the next index from the data (a dependent chain, like pop -> fetch -> test -> next).
// Prints ns per step for 1 wave per SIMD-quarter (one wave per CU) and for 4 waves per SIMD (16 per CU), node arrays
// of 64 KB (L1), 2 MB (L2), 13 MB (the bench scene's nodes), 256 MB.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/step_bench tools/step_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld16(__amdgpu_buffer_rsrc_t r, uint32_t off)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int F, int D, bool UNIFORM>
__global__ __launch_bounds__(64) void k_chain(const float* nodes, uint32_t n_nodes, int K, float* sink, unsigned long long* ticks)
{
    __shared__ float lds[1024];
    const uint32_t lane = threadIdx.x;
    lds[lane] = (float)lane, lds[lane + 64] = 1.0f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(nodes), 0, n_nodes * 128u, 0x00020000);
    uint32_t idx = (blockIdx.x * 977u + (UNIFORM ? 0u : lane * 131u)) % n_nodes;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < K; s++) {
        const uint32_t off = idx * 128u;
        const float4 a = ld16(rs, off), b = ld16(rs, off + 16), c = ld16(rs, off + 32), d = ld16(rs, off + 48);
        const float4 e = ld16(rs, off + 64), f = ld16(rs, off + 80), g = ld16(rs, off + 96);
        float x = a.x + b.y + c.z + d.w + e.x + f.y;
#pragma unroll
        for (int i = 0; i < F; i++)
            x = __builtin_fmaf(x, 1.0000001f, 0.5f); // dependent chain of F VALU instructions
#pragma unroll
        for (int i = 0; i < D; i++) { // dependent LDS round trips
            const uint32_t j = (__float_as_uint(x) >> 3) & 127u;
            x += lds[j];
        }
        acc += x;
        uint32_t nx = __float_as_uint(g.x) ^ (__float_as_uint(x) & 1u);
        if (UNIFORM)
            nx = __builtin_amdgcn_readfirstlane(nx);
        idx = nx % n_nodes;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    sink[blockIdx.x * 64 + lane] = acc;
    if (lane == 0)
        ticks[blockIdx.x] = t1 - t0;
}
template <int F, int D, bool U>
static void run(const char* name, const float* d_nodes, uint32_t n_nodes, int blocks, float* sink, unsigned long long* d_ticks)
{
    const int K = 400;
    hipLaunchKernelGGL((k_chain<F, D, U>), dim3(blocks), dim3(64), 0, 0, d_nodes, n_nodes, K, sink, d_ticks);
    hipLaunchKernelGGL((k_chain<F, D, U>), dim3(blocks), dim3(64), 0, 0, d_nodes, n_nodes, K, sink, d_ticks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(blocks);
    hipMemcpy(t.data(), d_ticks, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : t)
        sum += (double)v;
    printf("  %-34s %7.0f ns/step\n", name, sum / blocks * 10.0 / K); // 100 MHz ticks
}
int main()
{
    const size_t sizes[4] = { 64u << 10, 2u << 20, 13u << 20, 256u << 20 };
    float* sink;
    unsigned long long* ticks;
    hipMalloc(&sink, 4096 * 64 * 4 * 4);
    hipMalloc(&ticks, 16384 * 8);
    for (size_t bytes : sizes) {
        const uint32_t n = (uint32_t)(bytes / 128);
        std::vector<uint32_t> h(bytes / 4);
        uint32_t st = 12345u;
        for (auto& v : h) {
            st = st * 1664525u + 1013904223u;
            v = (st >> 4) & 0x3fffffu; // small positive float bit patterns / next indices
        }
        float* d;
        hipMalloc(&d, bytes);
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        for (int blocks : { 256, 4096, 16384 }) {
            printf("node array %zu KB, %d waves (%s)\n", bytes >> 10, blocks, blocks == 256 ? "1 per CU" : blocks == 4096 ? "16 per CU, one round" : "16 per CU, four rounds");
            run<0, 0, true>("uniform line, F=0", d, n, blocks, sink, ticks);
            run<0, 0, false>("line per lane, F=0", d, n, blocks, sink, ticks);
            run<100, 0, true>("uniform line, F=100", d, n, blocks, sink, ticks);
            run<100, 0, false>("line per lane, F=100", d, n, blocks, sink, ticks);
            run<300, 0, false>("line per lane, F=300", d, n, blocks, sink, ticks);
            run<500, 0, false>("line per lane, F=500", d, n, blocks, sink, ticks);
            run<100, 4, false>("line per lane, F=100, 4 LDS trips", d, n, blocks, sink, ticks);
            run<100, 8, false>("line per lane, F=100, 8 LDS trips", d, n, blocks, sink, ticks);
        }
        hipFree(d);
    }
    return 0;
}
