// Microbenchmark 2: the ray-walk loop body of k_grid_update_win with synthetic rays,
// ablated piece by piece.  Diagnostic tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct P { int W, H, Hp2, wx0, wy0, xw, yw; };

// bit 0: LDS atomic; 1: hx/hy tracking; 2: nvis count; 3: per-lane trip count (dx varies); 4: steep/ystep selects (ax/ay variables)
template <int V>
__global__ void __launch_bounds__(1024) k_walk(P p, int steps, unsigned *out)
{
    extern __shared__ unsigned win[];
    for (int i = threadIdx.x; i < p.W * p.Hp2; i += blockDim.x) win[i] = 0;
    __syncthreads();
    const int t = threadIdx.x;
    int dx = (V & 8) ? steps / 2 + (t * 37) % (steps / 2 + 1) : steps;
    const int klast = (t & 1) ? 0 : dx;
    const bool steep = (t >> 1) & 1;
    const int ystep = (t & 4) ? 1 : -1;
    int ax_x = 1, ax_y = 0, ay_x = 0, ay_y = 1;
    if (V & 16) { ax_x = steep ? 0 : 1; ax_y = steep ? 1 : 0; ay_x = steep ? ystep : 0; ay_y = steep ? 0 : ystep; }
    int lx = 100 + (t & 7), ly = 80 + ((t >> 3) & 7);
    int hx = -1, hy = -1;
    double error = 0.0, derr = 0.05 + 0.9 * ((t * 13) % 64) / 64.0;
    unsigned nvis = 0;
    for (int k = 0; k <= dx; ++k) {
        bool last = k == klast;
        unsigned wx = (unsigned)(lx - p.wx0), wy = (unsigned)(ly - p.wy0);
        bool inwin = wx < (unsigned)p.W && wy < (unsigned)p.H;
        if (V & 4) nvis += inwin ? 1u : 0u;
        if (V & 2) { hx = last ? lx : hx; hy = last ? ly : hy; }
        if (V & 1) { if (inwin && !last) atomicAdd(&win[wx * p.Hp2 + (wy >> 1)], 1u << ((wy & 1u) * 16u)); }
        else nvis += (inwin && !last) ? wx : 0;
        error += derr;
        bool stepy = error >= 0.5;
        lx += ax_x + (stepy ? ay_x : 0);
        ly += ax_y + (stepy ? ay_y : 0);
        error = stepy ? error - 1.0 : error;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + t] = nvis + hx + hy + lx + ly + win[t % (p.W * p.Hp2)];
}

template <int V>
float run(int grid, int block, int steps, unsigned *out, int reps)
{
    P p{200, 160, 80, 0, 0, 400, 400};
    size_t lds = (size_t)p.W * p.Hp2 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_walk<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_walk<V>, dim3(grid), dim3(block), lds, 0, p, steps, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_walk<V>, dim3(grid), dim3(block), lds, 0, p, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;
}

#define ROW(V, name) { printf("%-44s", name); for (int g : grids) printf("%8.1f", run<V>(g, 768, steps, out, 20)); printf("\n"); }
int main()
{
    unsigned *out; hipMalloc(&out, 4096 * 1024 * 4);
    const int steps = 110;
    int grids[] = {1, 64, 256, 512, 1024};
    printf("steps=%d 768 threads, 64KB LDS; us per launch\n%-44s", steps, "variant \\ grid");
    for (int g : grids) printf("%8d", g);
    printf("\n");
    ROW(0, "bare (error chain + coords + inwin)")
    ROW(1, "+LDS atomic")
    ROW(3, "+LDS atomic +hx/hy")
    ROW(7, "+LDS atomic +hx/hy +nvis")
    ROW(15, "+ ... + per-lane trip counts")
    ROW(31, "+ ... + steep/ystep selects (full body)")
    ROW(30, "full body without LDS atomic")
    return 0;
}
