// Micro-benchmark: what does the memory system give for the particle maps' sweep traffic alone?
// P maps of 400 x 400 uint32 counters (+ int8 pmap); per map 236 rows x 53 quads (16 B) are read, incremented and
// written back in place, 4 B of pmap per quad are read - the byte-window owner kernel's sweep without its ray cast.
// Modes: 0 read-modify-write, 1 read only, 2 write only, 3 RMW + pmap read, 4 RMW + pmap read + 360 scattered atomics.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_rmw tools/ubench_rmw.hip && ./ubench_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int XW = 400, YW = 400, ROWS = 236, X0 = 80;

template <int MODE, int BATCH, int ORDER, int AUXL, int AUXS>
__global__ void k_sweep(unsigned *pass, unsigned *hit, signed char *pm, int P, int maps_per_wg, unsigned *sink, int Y0, int QROW)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
    unsigned acc = 0;
    for (int it = 0; it < maps_per_wg; ++it) {
        const int m = blockIdx.x + it * gridDim.x;
        if (m >= P) break;
        unsigned *pp = pass + (size_t)m * XW * YW;
        signed char *pmm = pm + (size_t)m * XW * YW;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(pp, 0, XW * YW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(pmm, 0, XW * YW, 0x00020000);
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hit + (size_t)m * XW * YW, 0, XW * YW * 4, 0x00020000);
        const unsigned vo = lane < QROW ? lane * 16u : 0x80000000u;
        for (int r0 = ORDER ? wv * BATCH : wv; r0 < ROWS; r0 += NW * BATCH) {
            u32x4 p[BATCH];
            unsigned om[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = ORDER ? r0 + u : r0 + u * NW;
                const unsigned ro = (unsigned)((X0 + min(r, ROWS - 1)) * YW + Y0);
                const unsigned v = r < ROWS ? vo : 0x80000000u;
                p[u] = u32x4{1, 2, 3, 4};
                om[u] = 0;
                if (MODE != 2) p[u] = __builtin_amdgcn_raw_buffer_load_b128(rp, v, ro * 4, AUXL);
                if (MODE >= 3) om[u] = __builtin_amdgcn_raw_buffer_load_b32(rm, v >> 2, ro, 0);
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = ORDER ? r0 + u : r0 + u * NW;
                const unsigned ro = (unsigned)((X0 + min(r, ROWS - 1)) * YW + Y0);
                const unsigned v = r < ROWS ? vo : 0x80000000u;
                u32x4 q = p[u];
                q.x += 1; q.y += 2; q.z += 1; q.w += om[u];
                if (MODE == 5) __builtin_amdgcn_raw_buffer_store_b128(q, rh, v, ro * 4, 0);
                else if (MODE != 1) __builtin_amdgcn_raw_buffer_store_b128(q, rp, v, ro * 4, AUXS);
                else acc += q.x ^ q.y ^ q.z ^ q.w;
            }
        }
        if (MODE == 4 && threadIdx.x < 360) {
            const unsigned cell = (unsigned)((X0 + (threadIdx.x * 37) % ROWS) * YW + Y0 + (threadIdx.x * 11) % 200);
            atomicAdd(hit + (size_t)m * XW * YW + cell, 1u);
        }
    }
    if (MODE == 1 && acc == 0x12345678u) sink[0] = acc;
}

// streaming reference: every 16-byte quad of a buffer of n quads; mode 0 in-place RMW, 1 copy src -> dst, 2 read, 3 write
template <int MODE>
__global__ void k_stream(u32x4 *a, u32x4 *b, size_t n, unsigned *sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 0) { u32x4 v = a[i]; v.x += 1; a[i] = v; }
        if (MODE == 1) { u32x4 v = a[i]; v.x += 1; b[i] = v; }
        if (MODE == 2) { u32x4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
        if (MODE == 3) { a[i] = u32x4{1, 2, 3, (unsigned)i}; }
    }
    if (MODE == 2 && acc == 0x12345678u) sink[0] = acc;
}
template <int MODE>
static void stream(const char *name, unsigned *a, unsigned *b, size_t bytes, unsigned *sink)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n = bytes / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_stream<MODE>), dim3(256 * 8), dim3(512), 0, 0, (u32x4 *)a, (u32x4 *)b, n, sink);
    hipEventRecord(e0, 0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<MODE>), dim3(256 * 8), dim3(512), 0, 0, (u32x4 *)a, (u32x4 *)b, n, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double moved = (MODE <= 1 ? 2.0 : 1.0) * bytes;
    printf("%-34s %.3f ms  %.2f TB/s (%.2f GB)\n", name, ms, moved / ms / 1e9, moved / 1e9);
}

template <int MODE, int BATCH, int ORDER = 0, int AUXL = 0, int AUXS = 0>
static void run(const char *name, unsigned *pass, unsigned *hit, signed char *pm, int P, int threads, int grid, unsigned *sink, int Y0 = 92, int QROW = 53)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int mpw = (P + grid - 1) / grid;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_sweep<MODE, BATCH, ORDER, AUXL, AUXS>), dim3(grid), dim3(threads), 0, 0, pass, hit, pm, P, mpw, sink, Y0, QROW);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_sweep<MODE, BATCH, ORDER, AUXL, AUXS>), dim3(grid), dim3(threads), 0, 0, pass, hit, pm, P, mpw, sink, Y0, QROW);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double quads = (double)P * ROWS * QROW;
    double bytes = 0;
    if (MODE != 2) bytes += quads * 16;
    if (MODE != 1) bytes += quads * 16;
    if (MODE >= 3) bytes += quads * 4;
    printf("%-34s aux ld %2d st %2d order %d y0 %3d quads/row %2d threads %4d grid %5d batch %d: %.3f ms  %.2f TB/s (requested bytes %.2f GB)\n", name, AUXL, AUXS, ORDER, Y0, QROW, threads, grid, BATCH, ms, bytes / ms / 1e9, bytes / 1e9);
}

int main(int argc, char **argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 10000;
    unsigned *pass, *hit, *sink;
    signed char *pm;
    const size_t cells = (size_t)P * XW * YW;
    if (hipMalloc(&pass, cells * 4) != hipSuccess || hipMalloc(&hit, cells * 4) != hipSuccess || hipMalloc(&pm, cells) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipMemset(pass, 0, cells * 4); hipMemset(hit, 0, cells * 4); hipMemset(pm, 50, cells);
    // streaming references (whole 6.4 GB buffers)
    stream<2>("stream: read", pass, hit, cells * 4, sink);
    stream<3>("stream: write", pass, hit, cells * 4, sink);
    stream<1>("stream: copy pass -> hit", pass, hit, cells * 4, sink);
    stream<0>("stream: in-place RMW", pass, hit, cells * 4, sink);
    // the sweep's pattern: rows of 52 quads starting on a 64-byte piece (y0 = 80) or 53 quads starting mid-piece (y0 = 92)
    run<1, 4, 0, 0, 0>("read only", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<2, 4, 0, 0, 0>("write only", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<2, 4, 0, 0, 2>("write only", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 0, 0>("RMW in place", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 0, 2>("RMW in place", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 2, 2>("RMW in place", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 17, 17>("RMW in place", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 0, 0>("RMW in place", pass, hit, pm, P, 384, P, sink, 92, 53);
    run<0, 4, 0, 2, 2>("RMW in place", pass, hit, pm, P, 384, P, sink, 92, 53);
    run<0, 8, 0, 2, 2>("RMW in place", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<0, 4, 0, 2, 2>("RMW in place", pass, hit, pm, P, 768, 512, sink, 80, 52);
    run<3, 4, 0, 2, 2>("RMW + pmap read", pass, hit, pm, P, 384, P, sink, 80, 52);
    run<4, 4, 0, 2, 2>("RMW + pmap read + 360 atomics", pass, hit, pm, P, 384, P, sink, 80, 52);
    return 0;
}
