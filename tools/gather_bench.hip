// gather_bench.hip — what does a divergent 64-byte gather cost on MI355X?
// Each lane follows a pseudo-random chain through a table of 64-byte records
// (4 x global_load_dwordx4 per record), like the traversal's node phase.
// Prints records per cycle per CU for several table sizes and waves per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench && ./gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct alignas(64) Rec { float4 q[4]; };

template <int ILP>
__global__ __launch_bounds__(256) void k_gather(const Rec* __restrict__ tab, uint32_t mask, int iters, float* out)
{
    uint32_t idx[ILP];
    float acc = 0.f;
    for (int k = 0; k < ILP; k++)
        idx[k] = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + k * 40503u;
    for (int i = 0; i < iters; i++) {
        float4 a[ILP][4];
        for (int k = 0; k < ILP; k++) {
            const Rec* r = tab + (idx[k] & mask);
            a[k][0] = r->q[0], a[k][1] = r->q[1], a[k][2] = r->q[2], a[k][3] = r->q[3];
        }
        for (int k = 0; k < ILP; k++) {
            acc += a[k][0].x + a[k][1].y + a[k][2].z;
            idx[k] = idx[k] * 1664525u + 1013904223u + __float_as_uint(a[k][3].w); // dependent on the data
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Cooperative variant: 4 adjacent lanes fetch the 4 quads of one record (16 records per
// load instruction, 4 instructions for the wave's 64 records), transposed back through LDS.
__global__ __launch_bounds__(256) void k_gather_quad(const Rec* __restrict__ tab, uint32_t mask, int iters, float* out)
{
    __shared__ float4 stage[4][64 * 4 + 4];
    float4* st = stage[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float4 v[4];
        for (int k = 0; k < 4; k++) {
            const uint32_t src = 16u * k + (lane >> 2);
            const uint32_t r = __shfl(idx, src, 64) & mask;
            v[k] = tab[r].q[lane & 3u];
        }
        for (int k = 0; k < 4; k++)
            st[(16u * k + (lane >> 2)) * 4u + (lane & 3u)] = v[k];
        __builtin_amdgcn_wave_barrier();
        const float4 a0 = st[lane * 4u], a1 = st[lane * 4u + 1], a2 = st[lane * 4u + 2], a3 = st[lane * 4u + 3];
        __builtin_amdgcn_wave_barrier();
        acc += a0.x + a1.y + a2.z;
        idx = idx * 1664525u + 1013904223u + __float_as_uint(a3.w);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Narrow variant: one dword per lane per record (is the cost per access or per byte?)
__global__ __launch_bounds__(256) void k_gather_dword(const Rec* __restrict__ tab, uint32_t mask, int iters, float* out)
{
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        const float w = tab[idx & mask].q[3].w;
        acc += w;
        idx = idx * 1664525u + 1013904223u + __float_as_uint(w);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const size_t sizes_mb[] = { 1, 2, 8, 64 };
    for (size_t mb : sizes_mb) {
        const uint32_t n = (uint32_t)(mb << 20) / 64;
        std::vector<Rec> h(n);
        for (uint32_t i = 0; i < n; i++)
            for (int k = 0; k < 4; k++)
                h[i].q[k] = make_float4(1.f, 2.f, 3.f, 0.f);
        Rec* tab;
        hipMalloc(&tab, (size_t)n * 64);
        hipMemcpy(tab, h.data(), (size_t)n * 64, hipMemcpyHostToDevice);
        for (int blocks_per_cu : { 2, 4 }) {
            for (int ilp : { 1, 2, 3, 4 }) {
                const int iters = 400;
                const int grid = cus * blocks_per_cu;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (ilp == 3)
                        hipLaunchKernelGGL(k_gather_quad, dim3(grid), dim3(256), 0, 0, tab, n - 1, iters, out);
                    else if (ilp == 4)
                        hipLaunchKernelGGL(k_gather_dword, dim3(grid), dim3(256), 0, 0, tab, n - 1, iters, out);
                    else if (ilp == 1)
                        hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, tab, n - 1, iters, out);
                    else
                        hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, tab, n - 1, iters, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double recs = (double)grid * 256 * iters * (ilp == 2 ? 2 : 1);
                const double cyc = ms * 1e-3 * 2.4e9;
                printf("table %3zu MB  waves/CU %2d  mode %d (1,2 = per-lane ILP; 3 = quad-cooperative; 4 = one dword) : %.3f ms  %.4f records/cycle/CU  (%.1f cycles per 64-lane gather step)  %.2f TB/s useful\n",
                    mb, blocks_per_cu * 4, ilp, ms, recs / cyc / cus, 64.0 / (recs / cyc / cus) , recs * 64 / (ms * 1e-3) / 1e12);
            }
        }
        hipFree(tab);
    }
    return 0;
}
