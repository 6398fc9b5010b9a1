// Micro-benchmark (tools/, not shipped): what does executing code for the FIRST time cost on
// gfx950?  A kernel runs a straight line of KB KiB of `s_nop 0` (4 bytes, one issue cycle each)
// three times inside one launch and stamps each pass with s_memtime; pass 0 fetches the code
// through a cold instruction cache, passes 1-2 run it warm.  The launch is repeated: if the
// instruction cache survived between launches, pass 0 of the second launch would be warm too.
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_icache tools/ubench_icache.hip && /tmp/ubench_icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KB>
__global__ void __launch_bounds__(64) k_line(unsigned long long *out, int passes)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {
        if (KB == 4) asm volatile(".rept 1024\n s_nop 0\n .endr" ::: "memory");
        if (KB == 16) asm volatile(".rept 4096\n s_nop 0\n .endr" ::: "memory");
        if (KB == 32) asm volatile(".rept 8192\n s_nop 0\n .endr" ::: "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * 4 + p] = t1 - t0;
        t0 = t1;
    }
}

template <int KB>
static int run(int grid)
{
    unsigned long long *d;
    CHECK(hipMalloc(&d, (size_t)grid * 4 * 8));
    std::vector<unsigned long long> h((size_t)grid * 4);
    for (int launch = 0; launch < 3; ++launch) {
        CHECK(hipMemset(d, 0, (size_t)grid * 4 * 8));
        hipLaunchKernelGGL(k_line<KB>, dim3(grid), dim3(64), 0, 0, d, 3);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        double s[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
        for (int b = 0; b < grid; ++b)
            for (int p = 0; p < 3; ++p) { s[p] += (double)h[(size_t)b * 4 + p]; if ((double)h[(size_t)b * 4 + p] > mx[p]) mx[p] = (double)h[(size_t)b * 4 + p]; }
        printf("%2d KiB line, %5d waves, launch %d: cycles per pass mean %.0f %.0f %.0f  max %.0f %.0f %.0f  (cold - warm = %.0f cycles = %.1f per 64-byte line)\n", KB, grid, launch,
               s[0] / grid, s[1] / grid, s[2] / grid, mx[0], mx[1], mx[2], (s[0] - s[2]) / grid, (s[0] - s[2]) / grid / (KB * 16));
    }
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    for (int grid : {1, 256, 1024, 4096}) {
        if (run<4>(grid)) return 1;
        if (run<16>(grid)) return 1;
        if (run<32>(grid)) return 1;
    }
    return 0;
}
