// ta_bench.hip — cost model of one global_load_dwordx4 wave instruction on MI355X as a
// function of how its 64 lane addresses are distributed. All patterns read a table
// that is L2-resident (1 MB) or L1-resident (8 KB); 16 waves per CU; the loads of one
// iteration are independent, iterations are chained through the data.
//   k distinct 64-byte records per instruction, lanes grouped (g) or interleaved (i);
//   "coal": 64 lanes x 16 B contiguous; "quad": 4 adjacent lanes share a 64-byte record,
//   16 random records; "stride96": 32+32 lanes walking two 96-byte-stride arrays (leaf phase).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { P_DISTINCT_G, P_DISTINCT_I, P_COAL, P_QUAD, P_STRIDE96, P_STRIDE48, P_DWORD_DISTINCT };

template <int PAT>
__global__ __launch_bounds__(256) void k_ta(const float4* __restrict__ tab, uint32_t mask16, int k, int iters, float* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t seed = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u; // per wave
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t a; // index in 16-byte units
            const uint32_t s = seed + j * 977u;
            if (PAT == P_DISTINCT_G) {
                const uint32_t grp = lane / (64u / k);
                a = (((s + grp * 40503u) * 1664525u) >> 4 << 2) + (j & 3);
            } else if (PAT == P_DISTINCT_I) {
                const uint32_t grp = lane % k;
                a = (((s + grp * 40503u) * 1664525u) >> 4 << 2) + (j & 3);
            } else if (PAT == P_COAL) {
                a = ((s * 1664525u) >> 4 << 6) + lane;
            } else if (PAT == P_QUAD) {
                a = (((s + (lane >> 2) * 40503u) * 1664525u) >> 4 << 2) + (lane & 3u);
            } else if (PAT == P_STRIDE96) {
                a = (((s + (lane >> 5) * 40503u) * 1664525u) >> 4 << 8) + (lane & 31u) * 6u + j;
            } else if (PAT == P_STRIDE48) {
                a = (((s + (lane >> 5) * 40503u) * 1664525u) >> 4 << 8) + (lane & 31u) * 3u + (j % 3);
            } else {
                a = (((s + lane * 40503u) * 1664525u) >> 4 << 2) + (j & 3);
            }
            if (PAT == P_DWORD_DISTINCT) {
                v[j].x = reinterpret_cast<const float*>(tab)[(size_t)(a & mask16) * 4];
                v[j].y = v[j].z = v[j].w = 0.f;
            } else
                v[j] = tab[a & mask16];
        }
        acc += v[0].x + v[1].y + v[2].z + v[3].w;
        seed = seed * 1664525u + 1013904223u + __float_as_uint(v[3].w + v[0].w + v[1].w + v[2].w);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int PAT>
static void run(const char* name, const float4* tab, uint32_t mask16, int k, int cus, float* out, const char* tname)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 2000, grid = cus * 4;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_ta<PAT>, dim3(grid), dim3(256), 0, 0, tab, mask16, k, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double instr_per_cu = 16.0 * iters * 4; // wave-level load instructions per CU
    printf("%-6s %-14s k=%2d : %8.1f cycles per wave-load per CU\n", tname, name, k, ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, 4 << 20);
    for (size_t bytes : { (size_t)8 << 10, (size_t)1 << 20, (size_t)64 << 20 }) {
        const uint32_t n16 = (uint32_t)(bytes / 16);
        std::vector<float4> h(n16, make_float4(1.f, 2.f, 3.f, 0.f));
        float4* tab;
        hipMalloc(&tab, bytes);
        hipMemcpy(tab, h.data(), bytes, hipMemcpyHostToDevice);
        const char* tname = bytes == (8 << 10) ? "8KB" : bytes == (1 << 20) ? "1MB" : "64MB";
        for (int k : { 1, 2, 4, 8, 16, 32, 64 })
            run<P_DISTINCT_G>("distinct-group", tab, n16 - 1, k, cus, out, tname);
        for (int k : { 2, 8, 32 })
            run<P_DISTINCT_I>("distinct-inter", tab, n16 - 1, k, cus, out, tname);
        run<P_COAL>("coalesced-1KB", tab, n16 - 1, 1, cus, out, tname);
        run<P_QUAD>("quad-64B", tab, n16 - 1, 16, cus, out, tname);
        run<P_STRIDE96>("stride-96B", tab, n16 - 1, 2, cus, out, tname);
        run<P_STRIDE48>("stride-48B", tab, n16 - 1, 2, cus, out, tname);
        run<P_DWORD_DISTINCT>("dword-distinct", tab, n16 - 1, 64, cus, out, tname);
        hipFree(tab);
    }
    return 0;
}
