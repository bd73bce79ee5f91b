// How many HIP streams of one process really run kernels side by side (hardware queues)?
// Each stream gets one single-workgroup kernel that spins ~1 ms; the wall time of S such launches tells
// how many ran concurrently. Build: hipcc --offload-arch=gfx950 -O2 tools/queue_bench.hip -o /tmp/queue_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main()
{
    const long long ticks = 100000; // 100 MHz -> 1 ms
    for (int S : { 1, 2, 3, 4, 6, 8, 12, 16, 24, 32 }) {
        std::vector<hipStream_t> st(S);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& s : st) spin<<<1, 64, 0, s>>>(1000, nullptr);
        hipDeviceSynchronize();
        const auto a = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 4; rep++)
            for (auto& s : st) spin<<<1, 64, 0, s>>>(ticks, nullptr);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        printf("streams %2d: %7.2f ms for %3d ms of kernels -> %.1f side by side\n", S, ms, 4 * S, 4 * S / ms);
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
