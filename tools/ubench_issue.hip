// Issue cost of the vector instruction classes k_icp is made of, and the LDS atomic rate the window ray casts run on,
// measured on the chip (gfx950): what `roofline.frac` of bench.py prices instruction counts with.
//
//   hipcc -O3 --offload-arch=gfx950 -o ubench_issue tools/ubench_issue.hip && ./ubench_issue
//
// Part 1: per class, a loop of 256 independent instructions (8 rotating destinations) run by W waves per SIMD on every CU;
//         cycles per wave-instruction per SIMD = elapsed x clock / (instructions per wave x waves per SIMD).  The clock is
//         taken from s_memtime against s_memrealtime (100 MHz) inside the kernel.
// Part 3: a chain of dependent v_add_f64 in one wave alone (latency per step of a serial recurrence: k_pose_compose).
// Part 2: ds_add_u32 (no return) from 64 lanes into a 36 864-cell (72 KiB) window of 16-bit counters, two per dword, as
//         k_grid_update_win / k_wedge_cast issue it: (a) every lane a random dword, (b) every lane walks its own line
//         through the window (the address pattern of the walk), (c) conflict-free (lane-linear); 1-16 waves per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int kRep = 32;        // instruction groups of 8 per loop trip (written out: R32)
constexpr int kTrips = 200;
#define R4(x) x x x x
#define R32(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x) R4(x)

#define GROUP8(op)                                                                                                               \
    asm volatile(op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc")

#define OP_ADD_F64(d) "v_add_f64 " #d ", " #d ", %8\n"
#define OP_MUL_F64(d) "v_mul_f64 " #d ", " #d ", %8\n"
#define OP_FMA_F64(d) "v_fma_f64 " #d ", " #d ", %8, %9\n"
#define OP_MIN_F64(d) "v_min_f64 " #d ", " #d ", %8\n"
#define OP_CMP_F64(d) "v_cmp_lt_f64 vcc, " #d ", %8\n"

template <int KIND>
__global__ void __launch_bounds__(256) k_issue_f64(double *out, unsigned long long *clk, double b, double c)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < kTrips; ++t) {
        if (KIND == 0) { R32(GROUP8(OP_ADD_F64);) }
        if (KIND == 1) { R32(GROUP8(OP_MUL_F64);) }
        if (KIND == 2) { R32(GROUP8(OP_FMA_F64);) }
        if (KIND == 3) { R32(GROUP8(OP_MIN_F64);) }
        if (KIND == 4) { R32(GROUP8(OP_CMP_F64);) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

#define GROUP8I(op)                                                                                                              \
    asm volatile(op(%0) op(%1) op(%2) op(%3) op(%4) op(%5) op(%6) op(%7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c))
#define OP_ADD_U32(d) "v_add_u32 " #d ", " #d ", %8\n"
#define OP_CNDMASK(d) "v_cndmask_b32 " #d ", " #d ", %8, vcc\n"
#define OP_DPP(d) "v_mov_b32_dpp " #d ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define OP_FMA_F32(d) "v_fma_f32 " #d ", " #d ", %8, %9\n"
#define OP_RCP_F32(d) "v_rcp_f32 " #d ", " #d "\n"
#define OP_MULLO(d) "v_mul_lo_u32 " #d ", " #d ", %8\n"
#define OP_CVT(d) "v_cvt_f32_i32 " #d ", " #d "\n"
#define OP_CNDMASK64(d) "v_cndmask_b32_e64 " #d ", " #d ", %8, vcc\n"

template <int KIND>
__global__ void __launch_bounds__(256) k_issue_b32(unsigned *out, unsigned long long *clk, unsigned b, unsigned c)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a0), "v"(b) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < kTrips; ++t) {
        if (KIND == 0) { R32(GROUP8I(OP_ADD_U32);) }
        if (KIND == 1) { R32(GROUP8I(OP_CNDMASK);) }
        if (KIND == 2) { R32(GROUP8I(OP_DPP);) }
        if (KIND == 3) { R32(GROUP8I(OP_FMA_F32);) }
        if (KIND == 4) { R32(GROUP8I(OP_RCP_F32);) }
        if (KIND == 5) { R32(GROUP8I(OP_MULLO);) }
        if (KIND == 6) { R32(GROUP8I(OP_CVT);) }
        if (KIND == 7) { R32(GROUP8I(OP_CNDMASK64);) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// ---- LDS atomics --------------------------------------------------------------------------------------------
constexpr int kWinDwords = 36864 / 2;
typedef __attribute__((address_space(3))) unsigned lds_u32_t;

template <int PATTERN>
__global__ void __launch_bounds__(1024) k_lds_add(unsigned *out, unsigned long long *clk, int steps, unsigned seed)
{
    extern __shared__ unsigned win[];
    for (int i = threadIdx.x; i < kWinDwords; i += blockDim.x) win[i] = 0u;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)win;
    unsigned x = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u) ^ seed;
    // (b) a line through a 192 x 192 window of halfwords: start cell and a per-step halfword stride of (dx, dy) with |dx| <= 1, |dy| <= 1
    unsigned a2 = base + 2u * ((x >> 8) % 36864u);
    const int dirs[8] = {2, -2, 384, -384, 386, -382, 382, -386};
    const int da = dirs[(x >> 3) & 7];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        unsigned addr;
        if (PATTERN == 0) {                                          // random dword every step
            x = x * 1664525u + 1013904223u;
            addr = base + 4u * ((x >> 10) % (unsigned)kWinDwords);
            (void)__hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)addr, 1u << ((x >> 5) & 16u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (PATTERN == 1) {                                   // the walk: one running halfword address per lane
            a2 += (unsigned)da;
            if (a2 - base >= 2u * 36864u) a2 = base + (a2 - base) % (2u * 36864u);
            (void)__hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)(a2 & ~3u), 1u << ((a2 << 3) & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {                                                     // conflict-free: lane-linear dwords
            addr = base + 4u * ((threadIdx.x + 64u * (unsigned)s) % (unsigned)kWinDwords);
            (void)__hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)addr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned sum = 0;
    for (int i = threadIdx.x; i < kWinDwords; i += blockDim.x) sum += win[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// ---- part 3: dependent chains (what one lane pays per step of a serial recurrence, k_pose_compose) ----------------
template <int KIND>
__global__ void __launch_bounds__(64) k_dep(double *out, unsigned long long *clk, double b, double c, int trips)
{
    if (KIND == 2) __builtin_amdgcn_s_setprio(3);
    double a = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < trips; ++t) {
        if (KIND == 0 || KIND == 2) { R32(asm volatile("v_add_f64 %0, %0, %1\nv_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (KIND == 1) { R32(asm volatile("v_add_f64 %0, %0, %1\nv_add_f64 %0, %0, -%2" : "+v"(a) : "v"(b), "v"(c));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

template <typename F>
static double time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs\n", prop.gcnArchName, cus);
    void *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)cus * 16 * 1024 * 8));
    CHECK(hipMalloc(&clk, 16));
    unsigned long long hclk[2];
    const double insts_per_wave = (double)kTrips * kRep * 8;
    printf("# part 1: cycles per wave-instruction per SIMD (256-thread workgroups = one wave per SIMD each)\n");
    auto report = [&](const char *name, int wps, double ms) {
        CHECK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0) ;      // cycles per ns: s_memtime ticks / (100 MHz ticks x 10 ns)
        const double cyc = ms * 1e6 * ghz / (insts_per_wave * wps);
        printf("%-16s waves/SIMD %d: %.3f ms, clock %.2f GHz, %.2f cycles per instruction per SIMD\n", name, wps, ms, ghz, cyc);
    };
    for (int wps : {1, 2, 4}) {
        const dim3 grid(cus * wps), block(256);
#define RUN_F64(K, NAME) report(NAME, wps, time_ms([&] { hipLaunchKernelGGL(k_issue_f64<K>, grid, block, 0, 0, (double *)out, clk, 1.0000001, 0.5); }))
#define RUN_B32(K, NAME) report(NAME, wps, time_ms([&] { hipLaunchKernelGGL(k_issue_b32<K>, grid, block, 0, 0, (unsigned *)out, clk, 3u, 5u); }))
        RUN_F64(0, "v_add_f64");
        RUN_F64(1, "v_mul_f64");
        RUN_F64(2, "v_fma_f64");
        RUN_F64(3, "v_min_f64");
        RUN_F64(4, "v_cmp_lt_f64");
        RUN_B32(0, "v_add_u32");
        // (the 4-byte VOP2 encoding of the select, back to back, issues four to ten times slower than its 8-byte VOP3 encoding
        // reading the same vcc.  It does not show in the real kernels: libslamhip built with every v_cndmask_b32_e32
        // re-encoded as _e64 - 9 504 of them in icp_kernels.hip - ran k_icp in 0.104 ms, the same as before: round 4.)
        RUN_B32(1, "v_cndmask_b32 (VOP2)");
        RUN_B32(7, "v_cndmask_b32_e64");
        RUN_B32(2, "v_mov_b32_dpp");
        RUN_B32(3, "v_fma_f32");
        RUN_B32(4, "v_rcp_f32");
        RUN_B32(5, "v_mul_lo_u32");
        RUN_B32(6, "v_cvt_f32_i32");
    }
    printf("# part 2: ds_add_u32 (no return), 64 lanes, 72 KiB window of 16-bit counters; adds per cycle per CU\n");
    const int steps = 4096;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds_add<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kWinDwords * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds_add<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kWinDwords * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lds_add<2>), hipFuncAttributeMaxDynamicSharedMemorySize, kWinDwords * 4));
    for (int waves : {1, 2, 4, 8, 12, 16}) {
        const dim3 grid(cus), block(64 * waves);
        const char *names[3] = {"random dwords", "line walk", "conflict-free"};
        for (int pat = 0; pat < 3; ++pat) {
            double ms = time_ms([&] {
                if (pat == 0) hipLaunchKernelGGL(k_lds_add<0>, grid, block, kWinDwords * 4, 0, (unsigned *)out, clk, steps, 12345u);
                if (pat == 1) hipLaunchKernelGGL(k_lds_add<1>, grid, block, kWinDwords * 4, 0, (unsigned *)out, clk, steps, 12345u);
                if (pat == 2) hipLaunchKernelGGL(k_lds_add<2>, grid, block, kWinDwords * 4, 0, (unsigned *)out, clk, steps, 12345u);
            });
            CHECK(hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0);
            const double adds = (double)steps * 64 * waves;          // per CU
            const double cycles = (double)hclk[0];                   // in-kernel, workgroup 0 (zeroing and the final sum excluded)
            printf("%-14s waves/CU %2d: %.3f ms, %.2f GHz, %.2f adds per cycle per CU, %.3e adds/s on %d CUs\n", names[pat], waves, ms, ghz,
                   adds / cycles, adds / cycles * ghz * 1e9 * cus, cus);
        }
    }
    printf("# part 3: one wave alone, a chain of dependent float64 adds; cycles per instruction\n");
    {
        const int trips = 1000;
        const char *names[3] = {"v_add_f64, dependent", "(v + a) - b, dependent pair", "v_add_f64, dependent, s_setprio 3"};
        for (int kind = 0; kind < 3; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k_dep<0>, dim3(1), dim3(64), 0, 0, (double *)out, clk, 1.0000001, 0.5, trips);
                if (kind == 1) hipLaunchKernelGGL(k_dep<1>, dim3(1), dim3(64), 0, 0, (double *)out, clk, 1.0000001, 0.5, trips);
                if (kind == 2) hipLaunchKernelGGL(k_dep<2>, dim3(1), dim3(64), 0, 0, (double *)out, clk, 1.0000001, 0.5, trips);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(hclk, clk, 8, hipMemcpyDeviceToHost));
            printf("%-36s %.2f cycles per add\n", names[kind], (double)hclk[0] / (trips * 64.0));
        }
    }
    return 0;
}
