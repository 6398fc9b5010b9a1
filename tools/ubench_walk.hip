// Microbenchmark: how does a short dependent ALU loop (the shape of the ray walk) scale
// with the number of workgroups on MI355X?  Diagnostic tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int VARIANT>
__global__ void __launch_bounds__(1024) k_walk(int steps, double derr0, unsigned *out, int lds_cells)
{
    extern __shared__ unsigned win[];
    if (lds_cells > 0) for (int i = threadIdx.x; i < lds_cells; i += blockDim.x) win[i] = 0;
    __syncthreads();
    double error = 0.0, derr = derr0 + 1e-3 * (threadIdx.x & 63);
    int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    unsigned acc = 0;
    int mysteps = steps;
    if (VARIANT & 4) mysteps = steps / 2 + ((threadIdx.x * 37) % (steps / 2 + 1));   // divergent trip counts
    for (int k = 0; k < mysteps; ++k) {
        unsigned wx = (unsigned)lx, wy = (unsigned)ly;
        bool in = wx < 4096u && wy < 4096u;
        if (VARIANT & 1) { if (in && lds_cells > 0) atomicAdd(&win[(wx * 97 + wy) % lds_cells], 1u); }
        else acc += in ? wx + wy : 0;
        error += derr;
        bool stepy = error >= 0.5;
        lx += 1;
        ly += stepy ? 1 : 0;
        error = stepy ? error - 1.0 : error;
        if (VARIANT & 2) { if ((k & 7) == (int)(threadIdx.x & 7)) acc ^= lx; }        // a divergent branch per step
    }
    __syncthreads();
    if (lds_cells > 0) acc += win[threadIdx.x % lds_cells];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lx + ly;
}

template <int V>
float run(int grid, int block, int steps, size_t lds, unsigned *out, int reps)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_walk<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_walk<V>, dim3(grid), dim3(block), lds, 0, steps, 0.37, out, (int)(lds / 4));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_walk<V>, dim3(grid), dim3(block), lds, 0, steps, 0.37, out, (int)(lds / 4));
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;   // us
}

int main()
{
    unsigned *out; hipMalloc(&out, 4096 * 1024 * 4);
    const int steps = 110;
    printf("steps=%d; time per launch in us\n", steps);
    printf("%-34s", "config \\ grid");
    int grids[] = {1, 8, 64, 128, 256, 512, 1024, 2048};
    for (int g : grids) printf("%8d", g);
    printf("\n");
    struct Cfg { const char *name; int block; size_t lds; int variant; } cfgs[] = {
        {"768thr lds0   plain", 768, 0, 0}, {"768thr lds72K plain", 768, 72 * 1024, 0},
        {"768thr lds72K +ldsatomic", 768, 72 * 1024, 1}, {"768thr lds0   +branch", 768, 0, 2},
        {"768thr lds0   +divergent trips", 768, 0, 4}, {"256thr lds0   plain", 256, 0, 0},
        {"1024thr lds0  plain", 1024, 0, 0}, {"768thr lds144K plain", 768, 144 * 1024, 0},
    };
    for (auto &c : cfgs) {
        printf("%-34s", c.name);
        for (int g : grids) {
            float us = 0;
            switch (c.variant) {
            case 0: us = run<0>(g, c.block, steps, c.lds, out, 20); break;
            case 1: us = run<1>(g, c.block, steps, c.lds, out, 20); break;
            case 2: us = run<2>(g, c.block, steps, c.lds, out, 20); break;
            case 4: us = run<4>(g, c.block, steps, c.lds, out, 20); break;
            }
            printf("%8.1f", us);
        }
        printf("\n");
    }
    return 0;
}
