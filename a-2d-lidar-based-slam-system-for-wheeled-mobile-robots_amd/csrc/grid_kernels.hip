// Occupancy-grid kernels for gfx950 (MI355X): the float-error Bresenham ray walk of the
// reference with integer evidence counters, the threshold/finalize pass, and the
// stand-alone line rasteriser.
//
// Functional spec = the reference's Python (W12m = "W12_LiDAR SLAM/w12-mapping/
// course_agv_slam/scripts" under /root/reference):
//   Mapping.update   W12m/mapping.py:22-51    -> cast_ray / k_grid_update*
//   bresenham        W12m/bresenham.py:2-58   -> ray_setup + the walk loops, k_bresenham
//   pmap threshold   W12m/mapping.py:47-50    -> k_grid_finalize
//   publishMap       W12m/slam_ekf.py:270-271 -> k_grid_transpose
//
// Evidence model (DESIGN.md "K4"): the reference adds 0.01 per pass-through cell and 20
// on the last cell of each ray into a float64 map and thresholds at > 10.  Those sums
// commute only as integers, so the device keeps uint32 pass / hit counters per cell
// (atomicAdd, order-free, bit-reproducible) and the finalize kernel applies the rule the
// float sums obey: untouched -> 50; hit >= 1 (20 > 10) or pass >= 1001 (the 1001st
// sequential +0.01 is the first float64 sum above 10) -> 100; else 0.
//
// The ray walk keeps the reference's float64 error accumulation verbatim
// (error += dy/float(dx); if error >= 0.5: y += ystep; error -= 1.0): 15 % of lines
// differ from integer Bresenham, so anything else breaks cell parity.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>
#include <climits>
#include <type_traits>

#include "slam_internal.h"
#include "slam_stamps.h"

namespace slam {

struct Ray {
    int x0, y0, dx, ystep;   // walk origin (after steep / endpoint swaps), run length, y direction
    double derr;
    bool steep, flag;        // flag: the walk runs end -> start, i.e. path order is reversed (:57-58)
};

// bresenham.__init__ up to the loop (bresenham.py:10-43).  Returns false for identical
// endpoints (empty path, :10-11).
__device__ __forceinline__ bool ray_setup(int sx, int sy, int ex, int ey, Ray &r)
{
    if (sx == ex && sy == ey) return false;
    int adx = abs(ex - sx), ady = abs(ey - sy);
    r.steep = ady > adx;                                             // :14
    if (r.steep) { int t = sx; sx = sy; sy = t; t = ex; ex = ey; ey = t; }   // :15-17
    r.flag = sx > ex;                                                // :19
    if (r.flag) { int t = sx; sx = ex; ex = t; t = sy; sy = ey; ey = t; }    // :20-29
    r.x0 = sx; r.y0 = sy;
    r.dx = ex - sx;                                                  // :32
    int dy = abs(ey - sy);                                           // :33
    r.derr = (double)dy / (double)r.dx;                              // :35  (IEEE division)
    r.ystep = sy < ey ? 1 : -1;                                      // :40-43
    return true;
}

// World coordinate -> cell index, int(scale * (v + off)) truncated toward zero
// (mapping.py:33-36).  Flags what Python would raise on (NaN: ValueError, inf: OverflowError).
__device__ __forceinline__ int to_cell(double v, double scale, double off, int &bad)
{
    double c = scale * (v + off);
    if (c != c) { bad |= kStatusNaN; return 0; }
    if (!(fabs(c) < (double)kMaxRayCells)) { bad |= kStatusOverflow; return 0; }
    return (int)c;
}

// One ray of Mapping.update (mapping.py:38-50): +1 pass on every in-bounds cell of the
// path except the last, +1 hit on the last.  Returns the number of in-bounds cells.
// The first cell of the path (the ray origin, shared by every ray of the scan) is not
// written here when `skip_first` is set: the caller adds it once per wave.
__device__ __forceinline__ unsigned cast_ray(uint32_t *__restrict__ pass, uint32_t *__restrict__ hit, int xw, int yw,
                                             int pcx, int pcy, int pox, int poy, bool &first_pending)
{
    Ray r;
    first_pending = false;
    if (!ray_setup(pcx, pcy, pox, poy, r)) return 0u;
    unsigned nvis = 0;
    double error = 0.0;                                              // :34
    int y = r.y0;
    for (int k = 0; k <= r.dx; ++k) {                                // :45
        int x = r.x0 + k;
        int lx = r.steep ? y : x, ly = r.steep ? x : y;              // :46-49
        bool last = r.flag ? (k == 0) : (k == r.dx);                 // last cell in PATH order
        bool first = r.flag ? (k == r.dx) : (k == 0);
        if ((unsigned)lx < (unsigned)xw && (unsigned)ly < (unsigned)yw) {   // mapping.py:41
            ++nvis;
            if (first && lx == pcx && ly == pcy) {
                first_pending = true;                                // aggregated by the caller
            } else {
                size_t c = (size_t)lx * yw + ly;
                atomicAdd(last ? &hit[c] : &pass[c], 1u);            // :42-45
            }
        }
        error += r.derr;                                             // :51
        if (error >= 0.5) { y += r.ystep; error -= 1.0; }            // :53-55
    }
    return nvis;
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Shared tail of the two update kernels: per-wave aggregation of the origin cell and of
// the visit counter, and the sticky status word.
__device__ __forceinline__ void finish_wave(const GridDev &g, uint32_t *pass, int pcx, int pcy, bool first_pending,
                                            unsigned nvis, int bad)
{
    unsigned long long m = __ballot(first_pending);
    unsigned tot = wave_sum_u32(nvis);
    int anybad = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anybad |= __shfl_xor(anybad, off, kWave);
    if ((threadIdx.x & 63) == 0) {
        if (m) atomicAdd(&pass[(size_t)pcx * g.yw + pcy], (unsigned)__popcll(m));
        if (tot) atomicAdd(visit_slot(g.visits), (unsigned long long)tot);
        if (anybad) atomicOr(g.status, anybad);
    }
}

// Mapping.update for B scans given world-frame endpoints: one workgroup per scan, one
// lane per beam (a workgroup never spans two scans, so the origin cell is wave-uniform).
__global__ void __launch_bounds__(256) k_grid_update(GridDev g, const double *__restrict__ ox, const double *__restrict__ oy,
                                                     const double *__restrict__ cx, const double *__restrict__ cy, int n,
                                                     const int32_t *__restrict__ gob)
{
    __shared__ int first_bad;
    const int b = blockIdx.x;
    const int gi = gob ? gob[b] : 0;
    uint32_t *pass = g.pass + (size_t)gi * g.xw * g.yw, *hit = g.hit + (size_t)gi * g.xw * g.yw;
    int cbad = 0;                                                    // the origin is the same for every beam;
    const int pcx = to_cell(cx[b], g.scale, g.off_x, cbad);          // :35  a bad origin only raises if some
    const int pcy = to_cell(cy[b], g.scale, g.off_y, cbad);          // :36  beam is actually cast
    // the scan stops at its first beam that Python's int() would raise on (mapping.py:29-36: the beams before it have been
    // applied when the exception leaves update(), and the error is that beam's): find it before anything is cast
    if (threadIdx.x == 0) first_bad = INT_MAX;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = ox[(size_t)b * n + i], y = oy[(size_t)b * n + i];
        if (fabs(x) == INFINITY) continue;                           // mapping.py:30: only ox is tested
        int bad = cbad;
        (void)to_cell(x, g.scale, g.off_x, bad);
        (void)to_cell(y, g.scale, g.off_y, bad);
        if (bad) atomicMin(&first_bad, i);
    }
    __syncthreads();
    const int stop = first_bad;
    for (int base = 0; base < n; base += blockDim.x) {
        int i = base + threadIdx.x;
        int bad = 0;
        bool first_pending = false;
        unsigned nvis = 0;
        if (i < n && i <= stop) {
            double x = ox[(size_t)b * n + i], y = oy[(size_t)b * n + i];
            if (!(fabs(x) == INFINITY)) {                            // mapping.py:30: only ox is tested
                int pox = to_cell(x, g.scale, g.off_x, bad);        // :33
                int poy = to_cell(y, g.scale, g.off_y, bad);        // :34
                bad |= cbad;
                if (!bad) nvis = cast_ray(pass, hit, g.xw, g.yw, pcx, pcy, pox, poy, first_pending);
            }
        }
        finish_wave(g, pass, pcx, pcy, first_pending, nvis, bad);
    }
}

// Replay form: the world-frame endpoints are formed here from the raw ranges and the
// dead-reckoned pose (slam_ekf.py:89 with u2T of :130-137 and laserToNumpy of :115-123),
// all in float64 whatever storage type the ICP point buffers use, so cells never depend
// on that choice.  Block (k-1, l) handles scan k of trajectory l.
__global__ void __launch_bounds__(256) k_grid_update_replay(GridDev g, const float *__restrict__ ranges,
                                                            const double *__restrict__ cos_t, const double *__restrict__ sin_t,
                                                            const double *__restrict__ poses, int n_scan, int n,
                                                            const int32_t *__restrict__ got)
{
    __shared__ int first_bad;
    const int km1 = blockIdx.x, l = blockIdx.y;
    const int gi = got ? got[l] : 0;
    uint32_t *pass = g.pass + (size_t)gi * g.xw * g.yw, *hit = g.hit + (size_t)gi * g.xw * g.yw;
    const double *pose = poses + 3 * ((size_t)l * (n_scan - 1) + km1);
    const double px = pose[0], py = pose[1];
    const double c = cos(pose[2]), s = sin(pose[2]);
    const float *r = ranges + ((size_t)l * n_scan + km1 + 1) * n;
    int cbad = 0;
    const int pcx = to_cell(px, g.scale, g.off_x, cbad);
    const int pcy = to_cell(py, g.scale, g.off_y, cbad);
    auto world = [&](int i, double &x, double &y) {
        double rr = (double)r[i];
        if (rr == INFINITY) rr = 30.0;                               // slam_ekf.py:119
        double lx = cos_t[i] * rr, ly = sin_t[i] * rr;               // :122
        x = c * lx + (-s) * ly + px * 1.0;                           // u2T(pose).dot(pc), :89
        y = s * lx + c * ly + py * 1.0;
    };
    // (the scan stops at its first beam int() would raise on, see k_grid_update)
    if (threadIdx.x == 0) first_bad = INT_MAX;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double x, y;
        world(i, x, y);
        if (fabs(x) == INFINITY) continue;
        int bad = cbad;
        (void)to_cell(x, g.scale, g.off_x, bad);
        (void)to_cell(y, g.scale, g.off_y, bad);
        if (bad) atomicMin(&first_bad, i);
    }
    __syncthreads();
    const int stop = first_bad;
    for (int base = 0; base < n; base += blockDim.x) {
        int i = base + threadIdx.x;
        int bad = 0;
        bool first_pending = false;
        unsigned nvis = 0;
        if (i < n && i <= stop) {
            double x, y;
            world(i, x, y);
            if (!(fabs(x) == INFINITY)) {
                int pox = to_cell(x, g.scale, g.off_x, bad);
                int poy = to_cell(y, g.scale, g.off_y, bad);
                bad |= cbad;
                if (!bad) nvis = cast_ray(pass, hit, g.xw, g.yw, pcx, pcy, pox, poy, first_pending);
            }
        }
        finish_wave(g, pass, pcx, pcy, first_pending, nvis, bad);
    }
}

static inline int ray_block(int n)
{
    int blk = ((n + kWave - 1) / kWave) * kWave;
    return blk > 256 ? 256 : (blk < kWave ? kWave : blk);
}

hipError_t launch_grid_update(const GridDev &g, const double *ox, const double *oy, const double *cx, const double *cy,
                              int B, int n, const int32_t *gob, hipStream_t s)
{
    if (g.pmap_live && g.live_dirty) *g.live_dirty = true;
    SLAM_LAUNCH(k_grid_update, dim3(B), dim3(ray_block(n)), 0, s, g, ox, oy, cx, cy, n, gob);
    return hipGetLastError();
}

hipError_t launch_grid_update_replay(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                     const double *poses, int L, int n_scan, int n, const int32_t *got, hipStream_t s)
{
    if (n_scan < 2) return hipSuccess;
    if (g.pmap_live && g.live_dirty) *g.live_dirty = true;
    SLAM_LAUNCH(k_grid_update_replay, dim3(n_scan - 1, L), dim3(ray_block(n)), 0, s, g, ranges, cos_t, sin_t,
                       poses, n_scan, n, got);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// LDS-window ray casting (DESIGN.md "K4 v2").
//
// Direct global atomics run at the chip's scattered memory-side atomic rate (~2e10 /s,
// measured: profiles/r01_v1_*), far below what the walks themselves cost.  All rays of a
// scan start in the same cell and consecutive scans of a stream overlap almost entirely,
// so one workgroup takes a GROUP of consecutive scans of one stream, accumulates their
// pass-through evidence in a window of the map held in LDS and flushes the non-zero cells
// once, row by row, so the global atomics that remain are few and address-contiguous.
// Window cells are 16-bit pass counters, two per dword, updated with ds_add_u32 of
// 1 << 16*(cell & 1): a ray visits a cell at most once, so a count never exceeds the
// group's ray count (<= 65,535 by construction) and never carries into its neighbour.
// 36,864 cells = 72 KiB, so two workgroups share a CU's 160 KiB.  Hits (one per ray) go
// straight to the global counters.  The window is the bounding box of the group's ray
// origins and endpoints (clamped to the map) when that fits, else a sub-rectangle of it
// around the first origin; cells of a ray outside the window fall back to direct global
// atomics, so the result is exact for any window.
// ---------------------------------------------------------------------------------
// Between a workgroup's own global atomics and its plain loads of the same cells: every wave waits
// for its atomics to be acknowledged, then the workgroup barrier.  The map belongs to this
// workgroup for the whole launch and none of its lines has been loaded before (no stale L1 copy
// can exist), so nothing has to be written back or invalidated: a device-wide __threadfence()
// here made every owner write the XCD's whole L2 back.
__device__ __forceinline__ void owner_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
constexpr int kWinCells = 36864;     // 16-bit cells: 72 KiB of LDS
constexpr int kWinMaxGroup = 64;     // scans per workgroup
constexpr int kSortBins = 128;        // ray-length histogram (4 cells per bin)
constexpr int kMaxSortRays = 8192;    // rays per workgroup that can be length-sorted (u16 ids in LDS)
constexpr int kRaysPerLane = 4;     // lanes per workgroup = rays / kRaysPerLane (rays are dealt to waves dynamically)

// mapping.py:47-50 applied to the integer counters (see the header comment).
struct OccRule {
    int hit_levels;
    uint32_t pass_thresh[kMaxHitLevels];
    __host__ __device__ static OccRule of(const GridDev &g)
    {
        OccRule r;
        r.hit_levels = g.hit_levels;
        for (int k = 0; k < kMaxHitLevels; ++k) r.pass_thresh[k] = g.pass_thresh[k];
        return r;
    }
    __device__ __forceinline__ uint32_t value(uint32_t p, uint32_t h) const
    {
        if ((p | h) == 0) return 50u;
        // (as a sum of per-level tests with constant table indices: written as a select chain over the table the
        // compiler turns it into a per-lane indexed load from a private copy, i.e. scratch memory)
        bool occ = h >= (uint32_t)hit_levels;
#pragma unroll
        for (int k = 0; k < kMaxHitLevels; ++k) occ |= (h == (uint32_t)k) & (p >= pass_thresh[k]) & (k < hit_levels);
        return occ ? 100u : 0u;
    }
};

struct ScanConst {
    double px, py, c, s;   // ray origin (world) and heading cos / sin (replay source only)
    int pcx, pcy, cbad, pad;
};

// Source of rays: scans of a replay (ranges + poses) ...
struct ReplaySource {
    const float *ranges;
    const double *cos_t, *sin_t, *poses;
    int n_scan, n;
    long traj_stride;      // floats between trajectories' scan blocks (0: every trajectory reads the same scans)
    int grid_per_traj;     // trajectory l casts into map l (particle hypotheses, BASELINE configs[2])
    const double *centres; // nullable [L][n_scan-1][2]: ray origins that are not the pose (w12-mapping-online,
                           // W12o/slam_ekf.py:71-77,104: the centre comes from /tf, the points from xEst)
    const double *heading_cs = nullptr;   // nullable [L][n_scan-1][2]: cos / sin of the poses' headings, where the pose kernel left them
    __device__ int scans_per_traj() const { return n_scan - 1; }
    __device__ int own_grid(int l) const { return grid_per_traj ? l : 0; }
    bool maps_are_private() const { return grid_per_traj != 0; }
    __device__ void scan_const(int l, int k, const GridDev &g, ScanConst &sc) const
    {
        const double *pose = poses + 3 * ((size_t)l * (n_scan - 1) + k);
        sc.px = pose[0]; sc.py = pose[1];
        if (heading_cs) {
            const double *cs = heading_cs + 2 * ((size_t)l * (n_scan - 1) + k);
            sc.c = cs[0]; sc.s = cs[1];
        } else {
            sc.c = cos(pose[2]); sc.s = sin(pose[2]);
        }
        sc.cbad = 0;
        double ox = sc.px, oy = sc.py;
        if (centres) {
            const double *ctr = centres + 2 * ((size_t)l * (n_scan - 1) + k);
            ox = ctr[0]; oy = ctr[1];
        }
        sc.pcx = to_cell(ox, g.scale, g.off_x, sc.cbad);
        sc.pcy = to_cell(oy, g.scale, g.off_y, sc.cbad);
    }
    // a beam's inputs (fetch) and its end cell (ray) are separate so that a kernel can have the loads in
    // flight while it does something else
    struct Beam { float r; double ct, st; };
    __device__ Beam fetch(int l, int k, int i) const
    {
        return Beam{ranges[(size_t)l * traj_stride + (size_t)(k + 1) * n + i], cos_t[i], sin_t[i]};
    }
    // false: beam skipped (mapping.py:30) or flagged in `bad`
    __device__ bool ray(const Beam &b, const ScanConst &sc, const GridDev &g, int &pox, int &poy, int &bad) const
    {
        double rr = (double)b.r;
        if (rr == INFINITY) rr = 30.0;                               // slam_ekf.py:119
        double lx = b.ct * rr, ly = b.st * rr;                       // :122
        double x = sc.c * lx + (-sc.s) * ly + sc.px * 1.0;           // u2T(pose).dot(pc), :89
        double y = sc.s * lx + sc.c * ly + sc.py * 1.0;
        if (fabs(x) == INFINITY) return false;
        pox = to_cell(x, g.scale, g.off_x, bad);
        poy = to_cell(y, g.scale, g.off_y, bad);
        bad |= sc.cbad;
        return !bad;
    }
    __device__ bool ray(int l, int k, int i, const ScanConst &sc, const GridDev &g, int &pox, int &poy, int &bad) const
    {
        return ray(fetch(l, k, i), sc, g, pox, poy, bad);
    }
};

// ... or explicit world-frame endpoints (Mapping.update's own arguments).
struct ExplicitSource {
    const double *ox, *oy, *cx, *cy;
    int B, n;
    __device__ int scans_per_traj() const { return B; }
    __device__ int own_grid(int) const { return 0; }
    bool maps_are_private() const { return false; }
    __device__ void scan_const(int, int k, const GridDev &g, ScanConst &sc) const
    {
        sc.px = cx[k]; sc.py = cy[k]; sc.c = 1.0; sc.s = 0.0;
        sc.cbad = 0;
        sc.pcx = to_cell(sc.px, g.scale, g.off_x, sc.cbad);
        sc.pcy = to_cell(sc.py, g.scale, g.off_y, sc.cbad);
    }
    struct Beam { double x, y; };
    __device__ Beam fetch(int, int k, int i) const { return Beam{ox[(size_t)k * n + i], oy[(size_t)k * n + i]}; }
    __device__ bool ray(const Beam &b, const ScanConst &sc, const GridDev &g, int &pox, int &poy, int &bad) const
    {
        if (fabs(b.x) == INFINITY) return false;                     // mapping.py:30
        pox = to_cell(b.x, g.scale, g.off_x, bad);
        poy = to_cell(b.y, g.scale, g.off_y, bad);
        bad |= sc.cbad;
        return !bad;
    }
    __device__ bool ray(int l, int k, int i, const ScanConst &sc, const GridDev &g, int &pox, int &poy, int &bad) const
    {
        return ray(fetch(l, k, i), sc, g, pox, poy, bad);
    }
};

// LDS byte address of a pointer into shared memory, and a fire-and-forget 32-bit add at such an address
typedef __attribute__((address_space(3))) unsigned lds_u32_t;
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ void lds_add_u32(unsigned addr, unsigned v)
{
    (void)__hip_atomic_fetch_add((lds_u32_t *)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, kWave));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, kWave));
    return v;
}

// Pass 2 of the window kernel: walk a workgroup's rays.  Rays are handed to waves 64 at a
// time from an LDS counter, in order of decreasing length when the group was sorted in pass 1
// (a wave runs as long as its longest ray: unsorted, 54 % of the lane-steps were idle).  The walk
// itself is VALU-issue bound (measured: ~4 cycles per instruction per wave, insensitive to
// occupancy, memory traffic and padding), so the body is kept short: the cell is kept as
// (lx, ly) and advanced incrementally, and the path's last cell (the hit, mapping.py:45) is
// remembered and written once after the loop.  It is the reference's loop
// (bresenham.py:45-55) step for step.
template <bool COVERS, class Src>
__device__ __forceinline__ unsigned cast_rays(const GridDev &g, const Src &src, const ScanConst *sc, int l, int s0, int n,
                                              int nrays, int *next_ray, const unsigned short *order, unsigned *win, int wx0,
                                              int wy0, int W, int H, int Hp2, uint32_t *__restrict__ pass,
                                              uint32_t *__restrict__ hit, const int *first_bad)
{
    unsigned nvis = 0;
    const int lane = threadIdx.x & 63;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(next_ray, kWave);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= nrays) break;
        if (base + lane >= nrays) continue;
        const int r = order ? (int)order[base + lane] : base + lane;   // longest rays first when sorted
        int s = r / n, i = r - s * n, pox, poy, b2 = 0;
        Ray ry;
        if (i >= first_bad[s]) continue;                             // beams from the scan's first bad one on are not cast
        if (!src.ray(l, s0 + s, i, sc[s], g, pox, poy, b2)) continue;
        if (!ray_setup(sc[s].pcx, sc[s].pcy, pox, poy, ry)) continue;
        const int klast = ry.flag ? 0 : ry.dx;                       // walk step of the path's LAST cell
        const int ax_x = ry.steep ? 0 : 1, ax_y = ry.steep ? 1 : 0;  // cell step per walk step ...
        const int ay_x = ry.steep ? ry.ystep : 0, ay_y = ry.steep ? 0 : ry.ystep;   // ... and per y step
        int lx = ry.steep ? ry.y0 : ry.x0, ly = ry.steep ? ry.x0 : ry.y0;
        int hx = -1, hy = -1;
        double error = 0.0;                                           // bresenham.py:34
        // A ray whose two end cells lie in the window stays in it (a Bresenham path never leaves the box
        // of its ends): no bounds checks, the cell kept as ONE running halfword index, four steps per
        // wave-wide loop test.  The path's last cell is the endpoint's cell (mapping.py:44-45); it takes the
        // hit and no pass count, so the walk leaves it out: a reversed path (flag) starts one step in.
        const int Hs = 2 * Hp2;
        const bool safe = (unsigned)(sc[s].pcx - wx0) < (unsigned)W && (unsigned)(sc[s].pcy - wy0) < (unsigned)H &&
                    (unsigned)(pox - wx0) < (unsigned)W && (unsigned)(poy - wy0) < (unsigned)H;
        // (a wave that also holds rays which leave the window walks all its rays the checked way: two loops
        // one after the other made the slowest workgroups - those whose half does not fit - slower)
        if (!__any(!safe)) {
            {
                // the cell as the LDS byte address of its 16-bit counter: dword = address & ~3, the count's
                // position in it from address bit 1 (9 VALU + 1 LDS instructions per step; 14 + 1 + the
                // per-step lane mask before)
                unsigned a2 = lds_addr(win) + 2u * (unsigned)((lx - wx0) * Hs + (ly - wy0));
                const int da_k = 2 * (ry.steep ? 1 : Hs), da_y = 2 * (ry.steep ? ry.ystep * Hs : ry.ystep);
                int rem = ry.dx;                                      // pass cells: all but the path's last
                auto advance = [&]() {
                    error += ry.derr;                                 // bresenham.py:51
                    const bool stepy = error >= 0.5;                  // :53
                    a2 += (unsigned)(da_k + (stepy ? da_y : 0));
                    error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);   // :55 (minus 1.0, or minus 0.0: exact)
                };
                if (ry.flag) advance();                               // step 0 is the hit cell: start one step in
                nvis += (unsigned)ry.dx + 1u;                         // every cell of the path is in the map
                auto step = [&]() {
                    lds_add_u32(a2 & ~3u, 1u << ((a2 << 3) & 31u));   // mapping.py:43
                    advance();
                };
                for (;;) {                                            // four steps per wave-wide test, no lane mask inside
                    const bool full = rem >= 4;
                    if (!__any(full)) break;
                    if (full) { step(); step(); step(); step(); rem -= 4; }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rem > u) step();
                atomicAdd(&hit[(size_t)pox * g.yw + poy], 1u);        // mapping.py:45
            }
            continue;
        }
        for (int k = 0; k <= ry.dx; ++k) {                            // :45
            bool last = k == klast;
            unsigned wx = (unsigned)(lx - wx0), wy = (unsigned)(ly - wy0);
            bool inwin = wx < (unsigned)W && wy < (unsigned)H;
            bool inmap = COVERS ? inwin : ((unsigned)lx < (unsigned)g.xw && (unsigned)ly < (unsigned)g.yw);   // mapping.py:41
            nvis += inmap ? 1u : 0u;
            hx = last ? lx : hx; hy = last ? ly : hy;
            if (inwin && !last) atomicAdd(&win[wx * Hp2 + (wy >> 1)], 1u << ((wy & 1u) * 16u));          // mapping.py:43
            if (!COVERS) { if (inmap && !inwin && !last) atomicAdd(&pass[(size_t)lx * g.yw + ly], 1u); }
            error += ry.derr;                                         // bresenham.py:51
            bool stepy = error >= 0.5;                                // :53
            lx += ax_x + (stepy ? ay_x : 0);
            ly += ax_y + (stepy ? ay_y : 0);
            error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);    // :55 (minus 1.0, or minus 0.0: exact)
        }
        if ((unsigned)hx < (unsigned)g.xw && (unsigned)hy < (unsigned)g.yw) atomicAdd(&hit[(size_t)hx * g.yw + hy], 1u);   // mapping.py:45
    }
    return nvis;
}

// Single-scan owner form: the rays' bounding box is cut into strips of rows (x ranges) that each fit
// the window, and the scan is walked once per strip, counting only the strip's cells.  A ray whose
// x range misses the strip is skipped; a walk cannot be entered midway (its state is a rounded
// running sum), so a ray that reaches a later strip is walked again from its start.  The path's
// last cell gets its hit in the strip it lies in: the global counter by a fire-and-forget atomic
// and bit 15 of its window cell, so that the sweep knows where this scan's hits fell.
template <class Src>
__device__ __forceinline__ unsigned cast_rays_strip(const GridDev &g, const Src &src, const ScanConst &sc, int l, int s0,
                                                    int nrays, int *next_ray, const unsigned short *order, unsigned *win,
                                                    int wx0, int wy0, int W, int H, int Hp2, uint32_t *__restrict__ hit,
                                                    int first_bad)
{
    unsigned nvis = 0;
    const int lane = threadIdx.x & 63;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(next_ray, kWave);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= nrays) break;
        if (base + lane >= nrays) continue;
        const int i = order ? (int)order[base + lane] : base + lane;   // longest rays first when sorted
        int pox, poy, b2 = 0;
        Ray ry;
        if (i >= first_bad) continue;                                // beams from the first bad one on are not cast
        if (!src.ray(l, s0, i, sc, g, pox, poy, b2)) continue;
        if (max(sc.pcx, pox) < wx0 || min(sc.pcx, pox) >= wx0 + W) continue;   // never enters this strip
        if (!ray_setup(sc.pcx, sc.pcy, pox, poy, ry)) continue;
        const int klast = ry.flag ? 0 : ry.dx;                       // walk step of the path's LAST cell
        const int ax_x = ry.steep ? 0 : 1, ax_y = ry.steep ? 1 : 0;  // cell step per walk step ...
        const int ay_x = ry.steep ? ry.ystep : 0, ay_y = ry.steep ? 0 : ry.ystep;   // ... and per y step
        int lx = ry.steep ? ry.y0 : ry.x0, ly = ry.steep ? ry.x0 : ry.y0;
        double error = 0.0;                                           // bresenham.py:34
        for (int k = 0; k <= ry.dx; ++k) {                            // :45
            const unsigned wx = (unsigned)(lx - wx0), wy = (unsigned)(ly - wy0);
            if (wx < (unsigned)W && wy < (unsigned)H) {               // in this strip (every in-map cell of the ray is in some strip)
                ++nvis;                                               // mapping.py:41
                unsigned *cell = &win[wx * Hp2 + (wy >> 1)];
                const unsigned sh = (wy & 1u) * 16u;
                if (k != klast) atomicAdd(cell, 1u << sh);            // mapping.py:43
                else { atomicOr(cell, 0x8000u << sh); atomicAdd(&hit[(size_t)lx * g.yw + ly], 1u); }   // :45
            }
            error += ry.derr;                                         // bresenham.py:51
            const bool stepy = error >= 0.5;                          // :53
            lx += ax_x + (stepy ? ay_x : 0);
            ly += ax_y + (stepy ? ay_y : 0);
            error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);    // :55 (minus 1.0, or minus 0.0: exact)
        }
    }
    return nvis;
}

constexpr int kWinPhase = 48 + 64;     // box[]: from here the phases of k_grid_update_win (windows [4][8], phase of a quadrant [4], boundaries in the sorted order [5])
constexpr int kWinBoxInts = kWinPhase + 48;   // bbox[4], flags; from [16]: the boxes of the four direction quadrants; from [48]: every scan's first bad beam (k_grid_update_win)
__host__ __device__ inline size_t win_sc_bytes(int group) { return ((size_t)group * sizeof(ScanConst) + 15) & ~(size_t)15; }
__host__ __device__ inline int win_sort_cap(long rays) { return rays <= kMaxSortRays ? (int)((rays + 7) & ~7L) : 0; }   // 16-byte multiple
__host__ __device__ inline size_t win_lds_bytes(int group, int sort_cap, int win_cells)
{
    return win_sc_bytes(group) + kWinBoxInts * 4 + 4 * kSortBins * 4 + (size_t)sort_cap * 2 + (size_t)win_cells * 2 + kLdsGuard;
}
// Window capacity (16-bit cells) of a launch: kWinCells, or, for small groups, whatever still lets two
// workgroups share a CU's 160 KiB (a single 360-beam scan: 40 288 cells instead of 36 864 - the
// bounding box of a scan of the 10 m x 8 m benchmark room, padded to 16-cell pieces, is 35-37 k cells)
inline int win_cells_for(int group, int sort_cap)
{
    const long half = 80 * 1024 - 512;                               // (a little room for allocation granules)
    long other = (long)win_lds_bytes(group, sort_cap, 0);
    long cells = ((half - other) / 2) & ~15L;
    return cells > kWinCells ? (int)cells : kWinCells;
}

template <class Src>
__global__ void __launch_bounds__(1024) k_grid_update_win(GridDev g, Src src, int group_size, const int32_t *__restrict__ got,
                                                          int exclusive, int sort_cap, int win_cells, int split)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS is carved for this launch's group size (win_lds_bytes): a small group leaves room for a
    // second workgroup on the CU
    ScanConst *sc = reinterpret_cast<ScanConst *>(smem);                                  // [group_size]
    int *box = reinterpret_cast<int *>(smem + win_sc_bytes(group_size));                  // bbox[4], window[4], flags
    int *hist = box + kWinBoxInts;                                                        // [4][kSortBins]: (phase, length bin)
    unsigned short *order = reinterpret_cast<unsigned short *>(hist + 4 * kSortBins);     // [sort_cap]
    unsigned *win = reinterpret_cast<unsigned *>(order + sort_cap);                       // [W][Hp/2] dwords
    char *guard = reinterpret_cast<char *>(win) + (size_t)win_cells * 2;
    lds_guard_fill(guard);
    STAMP_DECL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int l = blockIdx.y;
    // split: TWO workgroups per group of scans, one per direction half - a ray never crosses the column of its origin,
    // so the rays running towards larger x and those running towards smaller x touch disjoint halves of the group's
    // box and can be cast by different CUs: each walks its half's rays into its own window and flushes it
    const int my_half = split ? (int)(blockIdx.x & 1u) : -1;
    const int s0 = (int)(split ? blockIdx.x >> 1 : blockIdx.x) * group_size;
    const int cnt = min(group_size, src.scans_per_traj() - s0);
    const int gi = got ? got[l] : src.own_grid(l);
    uint32_t *pass = g.pass + (size_t)gi * g.xw * g.yw, *hit = g.hit + (size_t)gi * g.xw * g.yw;
    const int n = src.n, nrays = cnt * n;

    // exclusive owner of the map + live pmap + rows that are a multiple of 4 cells: the flush is a
    // plain vectorised read-modify-write of the touched rectangle that also re-thresholds pmap
    const bool fused = exclusive && g.pmap_live && (g.yw & 3) == 0 && (((size_t)gi * g.xw * g.yw) & 3) == 0;
    // rows of an even number of cells: two neighbouring counters are one aligned 8-byte word, and the two
    // 16-bit counts of a window dword are flushed by ONE 64-bit atomic (see the flush)
    const bool pair64 = (g.yw & 1) == 0 && (((size_t)gi * g.xw * g.yw) & 1) == 0;
    int *fb = box + 48;                                              // [cnt] first beam of a scan that Python's int() would raise on
    if (tid < cnt) { src.scan_const(l, s0 + tid, g, sc[tid]); fb[tid] = INT_MAX; }
    unsigned long long *wg_visits = reinterpret_cast<unsigned long long *>(box + 12);
    if (tid == 0) {
        box[0] = box[1] = INT_MAX; box[2] = box[3] = INT_MIN; box[9] = 0; *wg_visits = 0ull;
        for (int q = 0; q < 4; ++q) { box[16 + 4 * q] = box[17 + 4 * q] = INT_MAX; box[18 + 4 * q] = box[19 + 4 * q] = INT_MIN; }
        box[34] = 1; box[36] = 0;
    }
    const bool sorted = nrays <= sort_cap;
    // Direction quadrants (DESIGN.md "K4a"): a Bresenham path stays in the box of its two ends, so a ray never crosses the
    // column or the row of its origin, and the rays of the four direction quadrants (end cell right / left of, above / below
    // the origin) touch four nearly disjoint parts of the group's bounding box (up to the few cells the origins of the
    // group's scans differ by).  When the box of a workgroup's rays does not fit the window - a 10 m x 8 m room seen at an
    // angle spans 250 x 250 cells, 1.5 windows - it is cut at the origins' column, and a half that still does not fit at their
    // row: every part gets the window to itself, one PHASE after the other; every ray is still walked once, and none of
    // its cells takes the checked walk with scattered global atomics (round 4 stamps: 44 of a replay's 250 workgroups had a
    // half that did not fit; their walk took 109 k cycles against 36 k, and the launch lasted as long as they did).
    const bool quads = sorted && !exclusive;
    unsigned short *bins = reinterpret_cast<unsigned short *>(win);   // scratch until the window is zeroed
    for (int k = tid; k < 4 * kSortBins; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    STAMP(0);                                   // scan constants

    // pass 1: bounding box of everything the group's rays can touch, and of the four quadrants' rays
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN, bad = 0;
    for (int r = tid; r < nrays; r += blockDim.x) {
        int s = r / n, i = r - s * n, pox, poy, len = 0, b2 = 0, quad = 0;
        const bool valid = src.ray(l, s0 + s, i, sc[s], g, pox, poy, b2);
        if (valid) {
            quad = (pox >= sc[s].pcx ? 1 : 0) | (poy >= sc[s].pcy ? 2 : 0);
            const int rx0 = min(pox, sc[s].pcx), rx1 = max(pox, sc[s].pcx), ry0 = min(poy, sc[s].pcy), ry1 = max(poy, sc[s].pcy);
            bx0 = min(bx0, rx0); bx1 = max(bx1, rx1);
            by0 = min(by0, ry0); by1 = max(by1, ry1);
            len = max(abs(pox - sc[s].pcx), abs(poy - sc[s].pcy));
        }
        // a scan stops at its first beam that Python's int() would raise on (mapping.py:29-36: the beams before
        // it have been applied when the exception leaves update(), and the error is that beam's)
        if (b2) atomicMin(&fb[s], i);
        if (sorted) {                                                // bin by (phase,) length, longest first
            const int lbin = kSortBins - 1 - min(len >> 2, kSortBins - 1);
            if (quads) {
                // which phase of which workgroup casts this ray is decided behind the barrier, when the boxes are known
                // (bits 7-8: quadrant, bit 9: beam parity); rays with nothing to cast are nobody's
                bins[r] = (valid && len > 0) ? (unsigned short)(lbin | (quad << 7) | ((i & 1) << 9)) : (unsigned short)0xffffu;
            } else {
                bins[r] = (unsigned short)lbin;                      // parked in the (not yet zeroed) window
                atomicAdd(&hist[lbin], 1);
            }
        }
    }
    bx0 = wave_min_i32(bx0); by0 = wave_min_i32(by0); bx1 = wave_max_i32(bx1); by1 = wave_max_i32(by1);
    if (lane == 0 && bx0 <= bx1) {
        atomicMin(&box[0], bx0); atomicMin(&box[1], by0); atomicMax(&box[2], bx1); atomicMax(&box[3], by1);
    }
    __syncthreads();
    if (tid < cnt && fb[tid] != INT_MAX) {                           // the first bad beam's own error (NaN or overflow)
        int pox, poy, b2 = 0;
        (void)src.ray(l, s0 + tid, fb[tid], sc[tid], g, pox, poy, b2);
        atomicOr(g.status, b2);
    }
    int *phw = box + kWinPhase;                                      // [4][8] window of a phase: x0, y0, W, H, covers
    int *phq = box + kWinPhase + 32;                                 // [4] phase of a quadrant's rays (-1: not this workgroup's)
    int *seg = box + kWinPhase + 36;                                 // [5] boundaries of the phases in the sorted order
    if (tid == 0 && quads && box[0] <= box[2]) {
        // A quadrant's rays lie between the group's origins and the edges of its bounding box on the quadrant's side (every ray
        // lies in the box of its two ends): the box of quadrant q from the box of everything and the extremes of the origins
        int ox0 = INT_MAX, oy0 = INT_MAX, ox1 = INT_MIN, oy1 = INT_MIN;
        for (int k = 0; k < cnt; ++k) { ox0 = min(ox0, sc[k].pcx); ox1 = max(ox1, sc[k].pcx); oy0 = min(oy0, sc[k].pcy); oy1 = max(oy1, sc[k].pcy); }
        for (int q = 0; q < 4; ++q) {
            box[16 + 4 * q] = (q & 1) ? max(ox0, box[0]) : box[0]; box[18 + 4 * q] = (q & 1) ? box[2] : min(ox1, box[2]);
            box[17 + 4 * q] = (q & 2) ? max(oy0, box[1]) : box[1]; box[19 + 4 * q] = (q & 2) ? box[3] : min(oy1, box[3]);
        }
    }
    if (tid == 0 && quads) {
        // union of the quadrants in mask -> clamped box; false: empty
        auto box_of = [&](unsigned mask, int *o) -> bool {
            o[0] = o[1] = INT_MAX; o[2] = o[3] = INT_MIN;
            for (int q = 0; q < 4; ++q)
                if ((mask >> q & 1u) && box[16 + 4 * q] <= box[18 + 4 * q]) {
                    o[0] = min(o[0], box[16 + 4 * q]); o[1] = min(o[1], box[17 + 4 * q]); o[2] = max(o[2], box[18 + 4 * q]); o[3] = max(o[3], box[19 + 4 * q]);
                }
            o[0] = max(o[0], 0); o[1] = max(o[1], 0); o[2] = min(o[2], g.xw - 1); o[3] = min(o[3], g.yw - 1);
            if (pair64) o[1] &= ~1;                                  // the window's dwords line up with 8-byte pairs of counters
            return o[0] <= o[2] && o[1] <= o[3];
        };
        auto fits = [&](unsigned mask) -> bool {
            int o[4];
            return !box_of(mask, o) || (long)(o[2] - o[0] + 1) * ((o[3] - o[1] + 2) & ~1) <= win_cells;
        };
        // the parts (sets of quadrants) this workgroup casts, one phase each
        unsigned part[4] = {0u, 0u, 0u, 0u};
        int nph = 1, parity = 0;
        if (split) {
            // Two workgroups per group.  When the group's whole box fits one window, both take that window and share the
            // rays evenly, by beam parity (neighbouring beams are about equally long; the direction halves of a scan taken
            // off-centre differ up to 3 : 1 in cells to walk).  Else a workgroup per direction half, cut again at the
            // origins' row if the half does not fit.
            const unsigned mine = my_half ? 0xAu : 0x5u;             // quadrants right / left of the origins' column
            if (fits(0xFu)) { part[0] = 0xFu; parity = 1; }
            else if (fits(mine)) part[0] = mine;
            else { part[0] = mine & 0x3u; part[1] = mine & 0xCu; nph = 2; }
        } else {
            if (fits(0xFu)) part[0] = 0xFu;
            else if (fits(0x5u) && fits(0xAu)) { part[0] = 0x5u; part[1] = 0xAu; nph = 2; }
            else { part[0] = 1u; part[1] = 2u; part[2] = 4u; part[3] = 8u; nph = 4; }
        }
        for (int q = 0; q < 4; ++q) phq[q] = -1;
        for (int ph = 0; ph < nph; ++ph) {
            for (int q = 0; q < 4; ++q)
                if (part[ph] >> q & 1u) phq[q] = ph;
            int o[4], W = 0, H = 0, cov = 1;
            if (box_of(part[ph], o)) {
                W = o[2] - o[0] + 1; H = o[3] - o[1] + 1;
                if ((long)W * ((H + 1) & ~1) > win_cells) {
                    // still too large: the sub-rectangle next to the origins (the rays start there; clamped into the part's
                    // box, i.e. in its corner for a quadrant) and the rest by direct atomics
                    const int Hd = min(H, 192), Wd = min(W, win_cells / ((Hd + 1) & ~1));   // rows are stored padded to an even height
                    int cx0 = min(max(sc[0].pcx - Wd / 2, o[0]), o[2] - Wd + 1), cy0 = min(max(sc[0].pcy - Hd / 2, o[1]), o[3] - Hd + 1);
                    if (pair64) cy0 &= ~1;
                    o[0] = cx0; o[1] = cy0; W = Wd; H = Hd; cov = 0;
                }
                // the window is W rows of (H + 1) / 2 dwords: it must fit the win_cells 16-bit cells carved for it.  Should the
                // sizing above ever be wrong (it once used the unpadded height: a 193 x 191 box wrote 96 dwords past the
                // window), fall back to no window at all - every cell then takes the direct-atomic path, still exact - and
                // raise the internal-error status bit.
                if ((long)W * ((H + 1) >> 1) > win_cells / 2) { atomicOr(g.status, kStatusGuard); W = 0; H = 0; cov = 0; }
            } else {
                o[0] = o[1] = 0;
            }
            int *w = phw + 8 * ph;
            w[0] = o[0]; w[1] = o[1]; w[2] = W; w[3] = H; w[4] = cov;
        }
        box[34] = nph; box[36] = parity;
        box[10] = 0; box[11] = 0;
    }
    if (tid == 0 && !quads) {
        int x0 = max(box[0], 0), y0 = max(box[1], 0), x1 = min(box[2], g.xw - 1), y1 = min(box[3], g.yw - 1);
        int W = 0, H = 0, covers = 1;
        int fastwin = 0, strip_w = 0;                                // strips of the single-scan owner form
        if (x0 <= x1 && y0 <= y1) {
            // Fast owner sweep (see the end of the kernel): one scan, one hit occupies, and the window,
            // widened to whole 64-byte pieces of the counter rows (16 cells), still holds every cell.
            if (fused && cnt == 1 && nrays < 32768 && g.hit_levels == 1 && (g.yw & 15) == 0) {
                const int ya = y0 & ~15, Ha = ((y1 | 15) + 1) - ya, Wb = x1 - x0 + 1;
                int S = (int)(((long)Wb * Ha + win_cells - 1) / win_cells), Ws = (Wb + S - 1) / S;
                while ((long)Ws * Ha > win_cells && S < Wb) { ++S; Ws = (Wb + S - 1) / S; }
                if ((long)Ws * Ha <= win_cells) { y0 = ya; y1 = ya + Ha - 1; fastwin = S; strip_w = Ws; }
            }
            if (fused) y0 &= ~3;                                      // quads of the fused flush line up with the window's dwords
            else if (pair64) y0 &= ~1;                                // the window's dwords line up with 8-byte pairs of counters
            W = x1 - x0 + 1; H = y1 - y0 + 1;
            if (!fastwin && (long)W * ((H + 1) & ~1) > win_cells) {   // keep a sub-rectangle around the first origin
                int Hd = min(H, 192), Wd = min(W, win_cells / ((Hd + 1) & ~1));   // rows are stored padded to an even height
                int cx0 = min(max(sc[0].pcx - Wd / 2, x0), x1 - Wd + 1), cy0 = min(max(sc[0].pcy - Hd / 2, y0), y1 - Hd + 1);
                if (fused) cy0 &= ~3;
                else if (pair64) cy0 &= ~1;
                x0 = cx0; y0 = cy0; W = Wd; H = Hd;
                covers = 0;
            }
            if ((long)(fastwin ? strip_w : W) * ((H + 1) >> 1) > win_cells / 2) {   // (see above)
                atomicOr(g.status, kStatusGuard);
                fastwin = 0;
                W = 0; H = 0; covers = 0;
            }
        }
        phw[0] = x0; phw[1] = y0; phw[2] = W; phw[3] = H; phw[4] = covers; box[10] = fastwin; box[11] = strip_w;
        box[34] = 1; box[36] = 0;
        seg[0] = 0; seg[1] = nrays;
    }
    __syncthreads();
    STAMP(1);                                   // pass 1: endpoints, bounding boxes, phases
    int wx0 = phw[0], wy0 = phw[1], W = phw[2], H = phw[3];
    bool covers = phw[4] != 0;            // the window holds every in-map cell the phase's rays can touch
    int Hp2 = (H + 1) >> 1;               // dwords per window row
    const int phases = box[34];
    if (sorted) {
        if (quads) {                                                 // this workgroup's rays and their phases
            const bool parity = box[36] != 0;
            for (int r = tid; r < nrays; r += blockDim.x) {
                const unsigned b = bins[r];
                if (b == 0xffffu) continue;
                const int ph = phq[(b >> 7) & 3u];
                const bool mine = ph >= 0 && (!parity || (int)((b >> 9) & 1u) == my_half);
                const unsigned fin = (unsigned)ph * kSortBins + (b & 127u);
                bins[r] = mine ? (unsigned short)fin : (unsigned short)0xffffu;
                if (mine) atomicAdd(&hist[fin], 1);
            }
            __syncthreads();
        }
        // counting sort of the ray ids by (phase, length bin): exclusive scan of the histogram (wave 0), then every ray
        // claims a slot in its bin.  Order inside a bin is arbitrary; the map update does not depend on ray order.
        if (wave == 0) {
            int h8[8], tot = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) { h8[u] = hist[8 * lane + u]; tot += h8[u]; }
            int inc = tot;
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) { int v = __shfl_up(inc, off, kWave); if (lane >= off) inc += v; }
            int run = inc - tot;
            if (quads && (lane & 15) == 0) seg[lane >> 4] = run;     // first bin of a phase: kSortBins / 8 = 16 lanes per phase
            if (quads && lane == kWave - 1) seg[4] = inc;
#pragma unroll
            for (int u = 0; u < 8; ++u) { hist[8 * lane + u] = run; run += h8[u]; }
        }
        __syncthreads();
        for (int r = tid; r < nrays; r += blockDim.x)
            if (bins[r] != 0xffffu) order[atomicAdd(&hist[bins[r]], 1)] = (unsigned short)r;
        __syncthreads();
    }
    unsigned nvis = 0;
    const unsigned short *ord = sorted ? order : nullptr;
    const int strips = box[10], strip_w = box[11];
    const bool fast = strips != 0;        // single-scan owner form (the sweep at the end of the kernel)
    for (int ph = 0; ph < (fast ? 0 : phases); ++ph) {
        const int seg0 = seg[ph], seg1 = seg[ph + 1];
        if (seg0 == seg1) continue;                                   // (uniform) a part without rays
        {
            const int *o = phw + 8 * ph;
            wx0 = o[0]; wy0 = o[1]; W = o[2]; H = o[3]; covers = o[4] != 0; Hp2 = (H + 1) >> 1;
            __syncthreads();                                          // the previous phase's flush has read the window
            if (tid == 0) box[9] = 0;
        }
        for (int w = tid; w < W * Hp2; w += blockDim.x) win[w] = 0u;
        __syncthreads();
        STAMP(2);                                   // sort + zero
        // pass 2: walk the rays (the reference's float-error Bresenham, bresenham.py:45-55)
        const int *first_bad = fb;
        if (covers) nvis += cast_rays<true>(g, src, sc, l, s0, n, seg1 - seg0, &box[9], ord ? ord + seg0 : nullptr, win, wx0, wy0, W, H, Hp2, pass, hit, first_bad);
        else        nvis += cast_rays<false>(g, src, sc, l, s0, n, seg1 - seg0, &box[9], ord ? ord + seg0 : nullptr, win, wx0, wy0, W, H, Hp2, pass, hit, first_bad);
        __syncthreads();
        STAMP(3);                                   // walk

    // flush: one wave per window row, lanes along y (contiguous in the [x][y] map), two cells per lane
    // every workgroup flushes the same part of the map: start each one at a different row so
    // that concurrent flushes do not queue on the same addresses
    const int rot = W > 0 ? (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)W) : 0;
    for (int rr = wave; rr < (fused ? 0 : W); rr += nwaves) {
        const int row = rr + rot < W ? rr + rot : rr + rot - W;
        size_t gbase = (size_t)(wx0 + row) * g.yw + wy0;
        // The flush is bound by the chip's rate of global atomics (every workgroup of a replay adds its
        // ~25 k touched cells to the same map: 6 M atomics per 1 000 scans at 4 scans per group, 20 us
        // chip-wide): a dword of the window - two cells that are neighbours in the map row - goes out as ONE
        // 64-bit add of (count0, count1 << 32).  A carry out of the low counter would need 2^32 passes.
        if (pair64 && (wy0 & 1) == 0) {
            for (int d = lane; d < Hp2; d += kWave) {
                unsigned v = win[row * Hp2 + d];
                if (v) atomicAdd(reinterpret_cast<unsigned long long *>(&pass[gbase + 2 * d]),
                                 (unsigned long long)(v & 0xffffu) | ((unsigned long long)(v >> 16) << 32));
            }
            continue;
        }
        for (int d = lane; d < Hp2; d += kWave) {
            unsigned v = win[row * Hp2 + d];
            unsigned p0 = v & 0xffffu, p1 = v >> 16;
            if (p0) atomicAdd(&pass[gbase + 2 * d], p0);
            if (p1) atomicAdd(&pass[gbase + 2 * d + 1], p1);          // p1 != 0 implies 2d+1 < H
        }
    }
    }   // phases
    // Live pmap: this workgroup is the only writer of its map during the launch, so once its own
    // atomics (hits, out-of-window passes) have landed it finishes every cell its rays could have
    // touched - the bounding box of pass 1, clamped to the map - in one sweep: add the window's
    // pass counts with plain 16-byte read-modify-writes (no flush atomics) and re-threshold pmap,
    // the finalize pass restricted to what changed.
    if (fast) {
        // Single-scan owner form.  Nothing of this map is read that an atomic of this launch wrote:
        // pass counts come from the window, "a hit fell here" from the window's flag bits (the hit
        // COUNTERS are bumped by the fire-and-forget atomics of the walk and never read), and the
        // cell's earlier state from the live pmap itself: with one hit occupying, pmap == 100 says
        // "hit before or pass >= threshold" and counters only grow, so an occupied cell stays
        // occupied; otherwise hit == 0 and the new pass count decides (mapping.py:47-50).
        // The window's rows are whole 64-byte pieces of the counter rows (16 cells = 4 quads = 4
        // lanes): a piece the scan touched is read and written back whole, an untouched piece is
        // neither read nor written; pmap is stored only where it changes.  No cell is ever updated
        // by a global atomic: a bounding box larger than the window is cut into strips of rows, each
        // walked and swept in turn (a rotated 10 m x 8 m room spans ~250 x 250 cells: two strips).
        const uint32_t pthr = g.pass_thresh[0];
        int8_t *pm = g.pmap_live + (size_t)gi * g.xw * g.yw;
        const int qrow = H >> 2;                                      // H is a multiple of 16 here
        const unsigned qinv = (unsigned)((0x100000000ull + (unsigned)qrow - 1) / (unsigned)qrow);   // q / qrow == umulhi(q, qinv), q < 2^16
        constexpr int kBatch = 4;                                     // 16-byte read-modify-writes a lane keeps in flight
        for (int strip = 0; strip < strips; ++strip) {
        const int sx0 = wx0 + strip * strip_w, SW = min(strip_w, wx0 + W - sx0), total = SW * qrow;
        if (strip) __syncthreads();                                   // the previous strip's sweep has read the window
        for (int w = tid; w < SW * Hp2; w += blockDim.x) win[w] = 0u;
        if (tid == 0) box[9] = 0;
        __syncthreads();
        STAMP(2);                                   // (sort +) zero
        nvis += cast_rays_strip(g, src, sc[0], l, s0, nrays, &box[9], ord, win, sx0, wy0, SW, H, Hp2, hit, fb[0]);
        __syncthreads();
        STAMP(3);                                   // walk
        for (int q0 = tid; q0 < total; q0 += kBatch * blockDim.x) {
            uint4 p[kBatch];
            uint32_t om[kBatch], d0[kBatch], d1[kBatch];
            size_t at[kBatch];
            bool live[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int q = q0 + u * (int)blockDim.x;
                d0[u] = d1[u] = 0u;
                if (q < total) {
                    const int r = (int)__umulhi((unsigned)q, qinv), c = q - r * qrow;
                    const uint2 d = *reinterpret_cast<const uint2 *>(win + (r * Hp2 + 2 * c));
                    d0[u] = d.x; d1[u] = d.y;
                    at[u] = (size_t)(sx0 + r) * g.yw + (wy0 + 4 * c);
                }
                // the four lanes of a 64-byte piece decide together (total and blockDim are multiples of 4)
                const unsigned long long m = __ballot((d0[u] | d1[u]) != 0u);
                live[u] = ((m >> (lane & 60)) & 0xFull) != 0ull;
                if (live[u]) {
                    p[u] = *reinterpret_cast<const uint4 *>(pass + at[u]);
                    om[u] = *reinterpret_cast<const uint32_t *>(pm + at[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (!live[u]) continue;
                const uint32_t c0 = d0[u] & 0x7fffu, c1 = (d0[u] >> 16) & 0x7fffu, c2 = d1[u] & 0x7fffu, c3 = (d1[u] >> 16) & 0x7fffu;
                p[u].x += c0; p[u].y += c1; p[u].z += c2; p[u].w += c3;
                *reinterpret_cast<uint4 *>(pass + at[u]) = p[u];
                if (!(d0[u] | d1[u])) continue;
                const uint32_t o = om[u];
                // per byte: 1 where the cell is occupied afterwards - it was (bit 6 is set in 100 only, not
                // in 0 or 50), a hit fell on it, or its pass count reached the threshold
                uint32_t occ = (o >> 6) & 0x01010101u;
                occ |= ((d0[u] >> 15) & 1u) | ((d0[u] >> 31) << 8) | (((d1[u] >> 15) & 1u) << 16) | ((d1[u] >> 31) << 24);
                occ |= (p[u].x >= pthr ? 1u : 0u) | (p[u].y >= pthr ? 0x100u : 0u) | (p[u].z >= pthr ? 0x10000u : 0u) | (p[u].w >= pthr ? 0x1000000u : 0u);
                // per byte: 0xff where this scan touched the cell (passed through or hit)
                const uint32_t tm = ((d0[u] & 0xffffu) ? 0xffu : 0u) | ((d0[u] >> 16) ? 0xff00u : 0u) | ((d1[u] & 0xffffu) ? 0xff0000u : 0u) |
                                    ((d1[u] >> 16) ? 0xff000000u : 0u);
                const uint32_t out = (o & ~tm) | ((occ * 100u) & tm);
                if (out != o) *reinterpret_cast<uint32_t *>(pm + at[u]) = out;
            }
        }
        STAMP_SYNC();
        STAMP(4);                                   // sweep
        }   // strips
        STAMP_COUNT(7, strips);
    } else if (exclusive && g.pmap_live) {
        owner_fence();
        __syncthreads();
        const int x0 = max(box[0], 0), y0 = max(box[1], 0), x1 = min(box[2], g.xw - 1), y1 = min(box[3], g.yw - 1);
        if (x0 <= x1 && y0 <= y1) {
            const OccRule rule = OccRule::of(g);
            int8_t *pm = g.pmap_live + (size_t)gi * g.xw * g.yw;
            // plain loads are safe: the counters were only touched by this workgroup's atomics (done,
            // fenced) and nothing of this map has been read into this CU's L1 during the launch
            if (fused) {
                const int ya = y0 & ~3, qrow = ((y1 | 3) + 1 - ya) >> 2, rows = x1 - x0 + 1;
                const int total = rows * qrow;
                // (pm is a char pointer and may alias anything for the compiler: the loads of a batch
                // are issued before its stores by hand)
                constexpr int kBatch = 4;                                     // 16-byte read-modify-writes a lane keeps in flight
                for (int q0 = tid; q0 < total; q0 += kBatch * blockDim.x) {
                    uint4 p[kBatch], h[kBatch];
                    size_t at[kBatch];
                    bool add[kBatch];
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        int q = min(q0 + u * (int)blockDim.x, total - 1);
                        int r = q / qrow, c = q - r * qrow;
                        int x = x0 + r, y = ya + 4 * c;
                        at[u] = (size_t)x * g.yw + y;
                        p[u] = *reinterpret_cast<const uint4 *>(pass + at[u]);
                        h[u] = *reinterpret_cast<const uint4 *>(hit + at[u]);
                        unsigned wx = (unsigned)(x - wx0), wy = (unsigned)(y - wy0);   // wy is a multiple of 4 when inside
                        unsigned d0 = 0, d1 = 0;
                        if (wx < (unsigned)W && wy < (unsigned)H) {
                            unsigned di = wx * Hp2 + (wy >> 1);
                            d0 = win[di];
                            d1 = (wy >> 1) + 1 < (unsigned)Hp2 ? win[di + 1] : 0u;
                        }
                        add[u] = (d0 | d1) != 0;
                        p[u].x += d0 & 0xffffu; p[u].y += d0 >> 16;
                        p[u].z += d1 & 0xffffu; p[u].w += d1 >> 16;
                    }
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) {
                        if (q0 + u * (int)blockDim.x >= total) continue;
                        if (add[u]) *reinterpret_cast<uint4 *>(pass + at[u]) = p[u];
                        uint32_t out = rule.value(p[u].x, h[u].x) | rule.value(p[u].y, h[u].y) << 8 |
                                       rule.value(p[u].z, h[u].z) << 16 | rule.value(p[u].w, h[u].w) << 24;
                        *reinterpret_cast<uint32_t *>(pm + at[u]) = out;
                    }
                }
            } else {
                for (int x = x0 + wave; x <= x1; x += nwaves) {
                    size_t rowb = (size_t)x * g.yw;
                    for (int y = y0 + lane; y <= y1; y += kWave) pm[rowb + y] = (int8_t)rule.value(pass[rowb + y], hit[rowb + y]);
                }
            }
        }
    }
    unsigned tot = wave_sum_u32(nvis);
    int anybad = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anybad |= __shfl_xor(anybad, off, kWave);
    if (lane == 0) {
        if (tot) atomicAdd(wg_visits, (unsigned long long)tot);      // summed per workgroup in LDS: ONE global add, at the very end
        if (anybad) atomicOr(g.status, anybad);
    }
    __syncthreads();
    if (tid == 0 && *wg_visits) atomicAdd(visit_slot(g.visits), *wg_visits);
    STAMP(4);                                   // flush / sweep
    STAMP_VAL((box[36] << 8) | (covers ? 1 : 0) | (phases << 4) | ((unsigned)(W * ((H + 1) & ~1)) << 12));
    STAMP_END(5);                               // [5] lifetime, [6] workgroups
    lds_guard_check(guard, g.status);
}


// ---------------------------------------------------------------------------------
// Single-scan owner kernel (DESIGN.md "K4 owner"): ONE scan cast into a map that this workgroup
// owns for the whole launch - the per-particle maps of BASELINE.json configs[2], and
// Mapping.update of one scan - with the live pmap kept current.  No cell is updated by a global
// atomic and nothing an atomic wrote is read back:
//   * every ray's walk state lives in registers (up to kOwnerRays rays per lane, longest rays in
//     the lowest waves), so a bounding box larger than the LDS window is handled by cutting it
//     into strips of rows (x ranges) that are walked and swept in turn, in ascending x.  The
//     reference walks along +x (or +y for steep lines) after its endpoint swaps, so the map x of
//     a walk never decreases except for steep lines that step towards -x: every other ray resumes
//     in a strip where it left the previous one and is walked exactly ONCE; those (a quarter of
//     the rays, and only when there is more than one strip) restart from their first cell in
//     every strip, because a walk cannot be entered midway - its state is a rounded running sum
//     (bresenham.py:45-55: the same operations in the same order either way);
//   * the path's last cell is the endpoint cell itself (the path runs start -> end inclusive):
//     it takes the hit without being walked to - a fire-and-forget atomic on the hit counter,
//     which is never read, and bit 15 of its 16-bit window cell;
//   * the sweep of a strip reads the window, adds the pass counts into whole 64-byte pieces of the
//     counter rows (plain 16-byte loads / stores of touched pieces only) and re-thresholds pmap
//     from the window's flag bits, the new pass counts and the live pmap's previous value: with
//     one hit occupying (the reference's +20 > 10), pmap == 100 says "hit before or pass >=
//     threshold" and counters only grow, so an occupied cell stays occupied; otherwise hit == 0
//     and the new pass count decides (mapping.py:42-50).
// Rays that leave the map (mapping.py:41 drops their outside cells) are rare and take a plain
// bounds-checked walk per strip.
// ---------------------------------------------------------------------------------
constexpr int kOwnerThreads = 512;   // 8 waves: two workgroups per CU at up to 128 VGPRs
constexpr int kOwnerMaxRays = 2;                    // rays per lane (template parameter: 1 up to 512 beams)

struct OwnRay {
    int lx, h;            // current cell: map x, halfword index in the strip's window
    int k, kend, klast;   // next walk step, last walk step, walk step of the path's last cell (takes the hit instead)
    int dlx_k, dlx_y;     // map-x step per walk step / per y advance
    int dh_k, dh_y;       // window halfword step per walk step / per y advance
    int ly;               // map y of the current cell (kept current between strips only)
    double error, derr;
    __device__ __forceinline__ void start(const Ray &rr, int H)
    {
        k = 0; kend = rr.dx; klast = rr.flag ? 0 : rr.dx;
        lx = rr.steep ? rr.y0 : rr.x0;
        ly = rr.steep ? rr.x0 : rr.y0;
        dlx_k = rr.steep ? 0 : 1; dlx_y = rr.steep ? rr.ystep : 0;
        dh_k = rr.steep ? 1 : H;  dh_y = rr.steep ? rr.ystep * H : rr.ystep;
        error = 0.0; derr = rr.derr;                                  // bresenham.py:34-35
        h = 0;
    }
};

template <class Src, int kOwnerRays>
__device__ __forceinline__ void owner_cast(const GridDev &g, const Src &src, int sort_cap, int win_cells, const int l)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ScanConst *sc = reinterpret_cast<ScanConst *>(smem);
    int *box = reinterpret_cast<int *>(smem + win_sc_bytes(1));
    int *hist = box + kWinBoxInts;                                   // [2][kSortBins]: (direction half,) length bin
    unsigned short *order = reinterpret_cast<unsigned short *>(hist + 2 * kSortBins);
    unsigned *win = reinterpret_cast<unsigned *>(order + sort_cap);
    char *guard = reinterpret_cast<char *>(win) + (size_t)win_cells * 2;
    lds_guard_fill(guard);
    STAMP_DECL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = src.own_grid(l), n = src.n;
    uint32_t *pass = g.pass + (size_t)gi * g.xw * g.yw, *hit = g.hit + (size_t)gi * g.xw * g.yw;
    int8_t *pm = g.pmap_live + (size_t)gi * g.xw * g.yw;
    unsigned long long *wg_visits = reinterpret_cast<unsigned long long *>(box + 12);
    unsigned short *bins = reinterpret_cast<unsigned short *>(win);   // scratch until the window is first zeroed

    if (tid == 0) {
        src.scan_const(l, 0, g, sc[0]);
        box[0] = box[1] = INT_MAX; box[2] = box[3] = INT_MIN; *wg_visits = 0ull; box[14] = 0; box[15] = INT_MAX;
    }
    if (tid < 2 * kSortBins) hist[tid] = 0;
    __syncthreads();
    STAMP(0);
    const ScanConst c0 = sc[0];

    // pass 1: endpoints, bounding box, length histogram
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = INT_MIN, by1 = INT_MIN, bad = 0;
    for (int r = tid; r < n; r += blockDim.x) {
        int pox = 0, poy = 0, len = 0, b2 = 0;
        if (src.ray(l, 0, r, c0, g, pox, poy, b2)) {
            bx0 = min(bx0, min(pox, c0.pcx)); bx1 = max(bx1, max(pox, c0.pcx));
            by0 = min(by0, min(poy, c0.pcy)); by1 = max(by1, max(poy, c0.pcy));
            len = max(abs(pox - c0.pcx), abs(poy - c0.pcy));
        }
        // the scan stops at its first beam that Python's int() would raise on (mapping.py:29-36: the
        // beams before it have been applied when the exception leaves update(), and the error is that
        // beam's)
        if (b2) atomicMin(&box[15], r);
        // longest first; rays that run towards larger x behind those that run towards smaller x (see `halves`)
        int bin = kSortBins - 1 - min(len >> 2, kSortBins - 1) + (pox >= c0.pcx ? kSortBins : 0);
        bins[r] = (unsigned short)bin;
        atomicAdd(&hist[bin], 1);
    }
    bx0 = wave_min_i32(bx0); by0 = wave_min_i32(by0); bx1 = wave_max_i32(bx1); by1 = wave_max_i32(by1);
    if (lane == 0 && bx0 <= bx1) {
        atomicMin(&box[0], bx0); atomicMin(&box[1], by0); atomicMax(&box[2], bx1); atomicMax(&box[3], by1);
    }
    __syncthreads();
    if (tid == 0) {
        if (box[15] != INT_MAX) {                                    // the first bad beam's own error (NaN or overflow)
            int pox, poy, b2 = 0;
            (void)src.ray(l, 0, box[15], c0, g, pox, poy, b2);
            atomicOr(g.status, b2);
        }
        // window = bounding box clamped to the map, rows widened to whole 16-cell pieces, cut into
        // S strips of at most Ws rows that each fit the window
        int x0 = max(box[0], 0), y0 = max(box[1], 0), x1 = min(box[2], g.xw - 1), y1 = min(box[3], g.yw - 1);
        int S = 0, Ws = 0, ya = 0, Ha = 0;
        if (x0 <= x1 && y0 <= y1) {
            ya = y0 & ~15; Ha = ((y1 | 15) + 1) - ya;
            const int Wb = x1 - x0 + 1;
            S = (int)(((long)Wb * Ha + win_cells - 1) / win_cells); Ws = (Wb + S - 1) / S;
            while ((long)Ws * Ha > win_cells) { ++S; Ws = (Wb + S - 1) / S; }   // (the host made sure one row fits: yw <= win_cells)
        }
        // Direction halves (as in k_grid_update_win): a box that needs two or more strips is cut at the origin's
        // column instead when both sides then fit - a ray never crosses that column, so each ray is walked once,
        // in the half it runs into, without a bounds test; strips walk some rays again and test every step.
        int halves = 0;
        const int split = min(max(c0.pcx, x0), x1);
        if (S >= 2 && (long)max(split - x0 + 1, x1 - split + 1) * Ha <= win_cells) { halves = 1; S = 2; }
        box[4] = x0; box[5] = ya; box[6] = x1 - x0 + 1; box[7] = Ha; box[10] = S; box[11] = Ws; box[8] = halves; box[9] = split;
    }
    if (wave == 0) {                                                 // counting sort by (half, length bin): scan of the histogram
        int h4[4], tot = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { h4[u] = hist[4 * lane + u]; tot += h4[u]; }
        int inc = tot;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { int v = __shfl_up(inc, off, kWave); if (lane >= off) inc += v; }
        int run = inc - tot;
#pragma unroll
        for (int u = 0; u < 4; ++u) { hist[4 * lane + u] = run; run += h4[u]; }
    }
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) order[atomicAdd(&hist[bins[r]], 1)] = (unsigned short)r;
    __syncthreads();
    STAMP(1);
    const int wx0 = box[4], wy0 = box[5], W = box[6], H = box[7], strips = box[10], strip_w = box[11];
    const bool halves = box[8] != 0;
    const int split = box[9];
    const int Hp2 = H >> 1;

    // this lane's rays: sorted rays tid, tid + blockDim, ...
    OwnRay ry[kOwnerRays];
    int pox[kOwnerRays], poy[kOwnerRays];
    // 0: nothing to walk; 1: all cells in the map, map x never decreases along the walk; 2: all cells in
    // the map, map x decreases (steep, towards -x): restarted in every strip; 3: leaves the map
    unsigned kind[kOwnerRays];
    unsigned nvis = 0;
    int unsafe = 0;
#pragma unroll
    for (int j = 0; j < kOwnerRays; ++j) {
        kind[j] = 0u; pox[j] = poy[j] = 0;
        Ray rr;
        rr.x0 = rr.y0 = 0; rr.dx = -1; rr.ystep = 1; rr.derr = 0.0; rr.steep = rr.flag = false;
        ry[j].start(rr, H);                                          // kend = -1: never active
        const int sr = tid + j * (int)blockDim.x;
        if (sr >= n || strips == 0) continue;
        int b2 = 0;
        if ((int)order[sr] >= box[15]) continue;                     // at or after the first bad beam: not cast
        if (!src.ray(l, 0, (int)order[sr], c0, g, pox[j], poy[j], b2)) continue;
        if (!ray_setup(c0.pcx, c0.pcy, pox[j], poy[j], rr)) continue;          // identical cells: empty path (bresenham.py:10-11)
        const bool inmap = (unsigned)c0.pcx < (unsigned)g.xw && (unsigned)c0.pcy < (unsigned)g.yw &&
                           (unsigned)pox[j] < (unsigned)g.xw && (unsigned)poy[j] < (unsigned)g.yw;
        if (!inmap) { kind[j] = 3u; unsafe = 1; continue; }
        kind[j] = (rr.steep && rr.ystep < 0 && strips > 1 && !halves) ? 2u : 1u;
        nvis += (unsigned)rr.dx + 1u;                                // every cell of the path is in the map (mapping.py:41)
        ry[j].start(rr, H);
    }
    if (unsafe) box[14] = 1;
    __syncthreads();
    const bool any_unsafe = box[14] != 0;

    const uint32_t pthr = g.pass_thresh[0];
    const int qrow = H >> 2;                                         // quads per window row (H is a multiple of 16)
    const unsigned qinv = qrow ? (unsigned)((0x100000000ull + (unsigned)qrow - 1) / (unsigned)qrow) : 0u;   // q / qrow == umulhi(q, qinv)
    for (int strip = 0; strip < strips; ++strip) {
        const int sx0 = halves ? (strip ? split : wx0) : wx0 + strip * strip_w;
        const int SW = halves ? (strip ? wx0 + W - split : split - wx0 + 1) : min(strip_w, wx0 + W - sx0), total = SW * qrow;
        if (strip) __syncthreads();                                  // the previous strip's sweep has read the window
        {
            uint4 *w4 = reinterpret_cast<uint4 *>(win);
            for (int w = tid; w < (SW * Hp2) >> 2; w += blockDim.x) w4[w] = make_uint4(0u, 0u, 0u, 0u);   // H % 16 == 0: whole uint4s
        }
        __syncthreads();
        STAMP(2);
        // hits: the path's last cell is the endpoint cell (mapping.py:44-45)
#pragma unroll
        for (int j = 0; j < kOwnerRays; ++j) {
            if (kind[j] == 0u) continue;
            if (halves && (pox[j] >= c0.pcx ? 1 : 0) != strip) continue;     // (the origin's column lies in both halves)
            const unsigned wx = (unsigned)(pox[j] - sx0), wy = (unsigned)(poy[j] - wy0);
            if (wx < (unsigned)SW && wy < (unsigned)H) {             // (the strips tile the bounding box clamped to the map)
                atomicOr(&win[wx * Hp2 + (wy >> 1)], 0x8000u << ((wy & 1u) * 16u));
                atomicAdd(&hit[(size_t)pox[j] * g.yw + poy[j]], 1u);
                if (kind[j] == 3u) ++nvis;
            }
        }
        // walk (bresenham.py:45-55)
        if (halves) {
            // every ray of this half, start to end, unchecked (see cast_rays): all its cells lie in the half's window
#pragma unroll
            for (int j = 0; j < kOwnerRays; ++j) {
                const OwnRay &o = ry[j];
                const bool mine = kind[j] == 1u && (pox[j] >= c0.pcx ? 1 : 0) == strip;
                if (!__any(mine)) continue;
                unsigned a2 = lds_addr(win) + 2u * (unsigned)((o.lx - sx0) * H + (o.ly - wy0));
                const int da_k = 2 * o.dh_k, da_y = 2 * o.dh_y;
                double error = 0.0;
                int rem = mine ? o.kend : 0;                         // pass cells: all but the path's last
                auto advance = [&]() {
                    error += o.derr;
                    const bool stepy = error >= 0.5;
                    a2 += (unsigned)(da_k + (stepy ? da_y : 0));
                    error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);
                };
                if (mine && o.klast == 0) advance();                 // a reversed path: its first walk step is the hit cell
                auto step = [&]() {
                    lds_add_u32(a2 & ~3u, 1u << ((a2 << 3) & 31u));   // mapping.py:43
                    advance();
                };
                for (;;) {
                    const bool full = rem >= 4;
                    if (!__any(full)) break;
                    if (full) { step(); step(); step(); step(); rem -= 4; }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rem > u) step();
            }
        } else
#pragma unroll
        for (int j = 0; j < kOwnerRays; ++j) {
            OwnRay &o = ry[j];
            const bool desc = kind[j] == 2u;
            if (desc) {                                              // restarts from its first cell (which lies to the right)
                Ray rr;
                ray_setup(c0.pcx, c0.pcy, pox[j], poy[j], rr);
                o.start(rr, H);
            }
            // the window halfword of the current cell, relative to this strip
            o.h = (o.lx - sx0) * H + (o.ly - wy0);
            // ascending: walks until it leaves the strip to the right; descending: until it has left it to the left
            const int lo = desc ? sx0 : INT_MIN, hi = desc ? INT_MAX : sx0 + SW;
            for (;;) {
                bool moved = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {                        // (four steps per wave-wide check; branch-free conditions)
                    const bool act = (o.k <= o.kend) & (o.lx >= lo) & (o.lx < hi);
                    moved |= act;
                    if (act) {
                        const bool cnt = ((unsigned)(o.lx - sx0) < (unsigned)SW) & (o.k != o.klast);
                        if (cnt) atomicAdd(&win[o.h >> 1], 1u << ((o.h & 1) * 16));   // mapping.py:43
                        o.error += o.derr;                           // bresenham.py:51
                        const bool stepy = o.error >= 0.5;           // :53
                        o.h += o.dh_k + (stepy ? o.dh_y : 0);
                        o.lx += o.dlx_k + (stepy ? o.dlx_y : 0);
                        o.error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);   // :55 (minus 1.0, or minus 0.0: exact)
                        ++o.k;
                    }
                }
                if (!__any(moved)) break;
            }
            o.ly = wy0 + (o.h - (o.lx - sx0) * H);                   // map y of the cell the ray stands on now
        }
        if (any_unsafe) {
            // rays with cells outside the map: plain walk of the whole ray, counting this strip's cells
#pragma unroll
            for (int j = 0; j < kOwnerRays; ++j) {
                if (kind[j] != 3u) continue;
                if (halves && (pox[j] >= c0.pcx ? 1 : 0) != strip) continue;   // (walked once, in the half they run into)
                Ray rr;
                ray_setup(c0.pcx, c0.pcy, pox[j], poy[j], rr);
                const int klast = rr.flag ? 0 : rr.dx;
                double error = 0.0;
                int y = rr.y0;
                for (int k = 0; k <= rr.dx; ++k) {
                    const int x = rr.x0 + k;
                    const unsigned wx = (unsigned)((rr.steep ? y : x) - sx0), wy = (unsigned)((rr.steep ? x : y) - wy0);
                    if (k != klast && wx < (unsigned)SW && wy < (unsigned)H) {
                        ++nvis;
                        atomicAdd(&win[wx * Hp2 + (wy >> 1)], 1u << ((wy & 1u) * 16u));
                    }
                    error += rr.derr;
                    if (error >= 0.5) { y += rr.ystep; error -= 1.0; }
                }
            }
        }
        __syncthreads();
        STAMP(3);
        // sweep: 4 lanes per 64-byte piece of a counter row
        constexpr int kBatch = 4;                                     // 16-byte read-modify-writes a lane keeps in flight
        for (int q0 = tid; q0 < total; q0 += kBatch * blockDim.x) {
            uint4 p[kBatch];
            uint32_t om[kBatch], d0[kBatch], d1[kBatch];
            size_t at[kBatch];
            bool live[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int q = q0 + u * (int)blockDim.x;
                d0[u] = d1[u] = 0u;
                if (q < total) {
                    const int r = (int)__umulhi((unsigned)q, qinv), c = q - r * qrow;
                    const uint2 d = *reinterpret_cast<const uint2 *>(win + (r * Hp2 + 2 * c));
                    d0[u] = d.x; d1[u] = d.y;
                    at[u] = (size_t)(sx0 + r) * g.yw + (wy0 + 4 * c);
                }
                // the four lanes of a piece decide together (total and blockDim are multiples of 4)
                const unsigned long long m = __ballot((d0[u] | d1[u]) != 0u);
                live[u] = ((m >> (lane & 60)) & 0xFull) != 0ull;
                if (live[u]) {
                    p[u] = *reinterpret_cast<const uint4 *>(pass + at[u]);
                    om[u] = *reinterpret_cast<const uint32_t *>(pm + at[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                if (!live[u]) continue;
                const uint32_t d0v = d0[u], d1v = d1[u];
                p[u].x += d0v & 0x7fffu; p[u].y += (d0v >> 16) & 0x7fffu; p[u].z += d1v & 0x7fffu; p[u].w += (d1v >> 16) & 0x7fffu;
                *reinterpret_cast<uint4 *>(pass + at[u]) = p[u];
                if (!(d0v | d1v)) continue;
                const uint32_t o = om[u];
                // one bit per byte (cell): touched by this scan / hit by this scan / occupied before (bit 6
                // is set in 100 only) / never touched before (bit 4 is set in 50 only)
                const uint32_t tb = ((d0v & 0xffffu) ? 1u : 0u) | ((d0v >> 16) ? 0x100u : 0u) | ((d1v & 0xffffu) ? 0x10000u : 0u) | ((d1v >> 16) ? 0x1000000u : 0u);
                const uint32_t fb = ((d0v >> 15) & 1u) | (((d0v >> 31) & 1u) << 8) | (((d1v >> 15) & 1u) << 16) | ((d1v >> 31) << 24);
                const uint32_t was = (o >> 6) & 0x01010101u, fresh = (o >> 4) & 0x01010101u;
                const uint32_t pmax = max(max(p[u].x, p[u].y), max(p[u].z, p[u].w));
                // a byte changes iff the cell was touched, was not occupied, and is fresh (50 -> 0 or 100),
                // hit now (-> 100) or at / over the pass threshold now (-> 100)
                if (((tb & ~was) & (fresh | fb)) != 0u || pmax >= pthr) {
                    uint32_t occ = was | fb;
                    occ |= (p[u].x >= pthr ? 1u : 0u) | (p[u].y >= pthr ? 0x100u : 0u) | (p[u].z >= pthr ? 0x10000u : 0u) | (p[u].w >= pthr ? 0x1000000u : 0u);
                    const uint32_t tm = tb * 255u;
                    const uint32_t out = (o & ~tm) | ((occ * 100u) & tm);
                    if (out != o) *reinterpret_cast<uint32_t *>(pm + at[u]) = out;
                }
            }
        }
        STAMP_SYNC();
        STAMP(4);
    }
    unsigned tot = wave_sum_u32(nvis);
    int anybad = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anybad |= __shfl_xor(anybad, off, kWave);
    if (lane == 0) {
        if (tot) atomicAdd(wg_visits, (unsigned long long)tot);
        if (anybad) atomicOr(g.status, anybad);
    }
    __syncthreads();
    if (tid == 0 && *wg_visits) atomicAdd(visit_slot(g.visits), *wg_visits);
    STAMP_COUNT(7, strips);
    STAMP_END(5);
    lds_guard_check(guard, g.status);
}

// The general owner kernel has two launch forms: one workgroup per map (blockIdx.y), or - behind the
// byte-window kernel below - a fixed number of workgroups that work off the list of maps that kernel
// left for it (redo[0]: count, redo[1]: workgroups done, redo[2..]: map numbers; normally empty).
template <class Src, int kOwnerRays>
__global__ void __launch_bounds__(kOwnerThreads, kOwnerRays == 1 ? 2 * kOwnerThreads / 256 : kOwnerThreads / 256)
k_grid_update_owner(GridDev g, Src src, int sort_cap, int win_cells)
{
    owner_cast<Src, kOwnerRays>(g, src, sort_cap, win_cells, (int)blockIdx.y);
}

template <class Src, int kOwnerRays>
__global__ void __launch_bounds__(kOwnerThreads) k_grid_update_owner_redo(GridDev g, Src src, int sort_cap, int win_cells, int32_t *redo)
{
    const int count = redo[0];
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        owner_cast<Src, kOwnerRays>(g, src, sort_cap, win_cells, redo[2 + it]);
        __syncthreads();                                             // the next map re-uses the LDS regions
    }
    // every workgroup has read the count before it reports here: the last one empties the list for the next launch
    if (threadIdx.x == 0 && atomicAdd(&redo[1], 1) == (int)gridDim.x - 1) { redo[0] = 0; redo[1] = 0; }
}

// ---------------------------------------------------------------------------------
// Single-scan owner kernel with a BYTE window (DESIGN.md "K4 owner8"): the plain case of the kernel above -
// every beam valid, every ray inside the map, at most one ray per lane - with 8-bit window cells, so that the
// whole bounding box of a scan (236 x 212 cells for the 10 m x 8 m benchmark room at 0.05 m) is ONE window of
// ~50 KB and three workgroups share a CU.  One phase per map: no strips, no direction halves, every ray walked
// once without bounds tests, one sweep.
//   * a window byte = 7-bit pass count + bit 7 "a hit fell here".  The scan's origin cell is the first cell of
//     every path and would count all n rays: it is left out of the walk and its count (the number of non-empty
//     rays) is added by the sweep.  Any other cell sees only the rays whose direction lies within a cell's width
//     of it: 54 of 360 next to the origin on the benchmark scan.
//   * nothing bounds a byte for arbitrary input (n identical rays), and a count of 128 would spill into the flag
//     and the neighbouring cell.  So the walk is CHECKED before anything is written to the map: the sum of all
//     window bytes (flags masked) must equal the number of cells the rays passed, which the set-up knows
//     (sum of dx - 1).  Every overflow event lowers the masked byte sum (by 128, 127 or 126 ...) and nothing
//     raises it, so equality proves that no byte overflowed.  On a mismatch - and for every case this kernel
//     does not handle: a bad beam, a ray leaving the map, a box larger than the window - the map's number goes
//     on the re-do list and the general kernel above casts it; this kernel has then written nothing to it.
//   * set-up with two barriers: the beam loads are in flight while LDS is initialised; the counting sort by
//     length (longest rays in the lowest waves) takes the rank inside a bin from the histogram atomic's return
//     value and the bin's offset from a wave-level scan that every wave does for itself; the sorted end cells
//     travel through LDS as packed 16-bit pairs instead of being loaded and computed a second time.
//   * the sweep goes by window rows: a wave takes a row (64 quads of it at a time), so the map addresses are a
//     scalar row offset plus a per-lane constant, and reads / writes the counters and pmap with non-temporal buffer
//     loads and stores whose per-lane offset lies beyond the buffer for lanes with nothing to do: BATCH rows per
//     wave in flight, no branches, no lane masks.  Counters and pmap rule exactly as in the kernel above
//     (mapping.py:42-50); the pmap arithmetic runs only for quads where something can change.
//   * what bounds it (tools/ubench_rmw.hip, profiles/r03_ubench_rmw.txt): the chip moves this traffic - 64-byte
//     pieces read and written back in place, scattered over 14 GB of maps - at 3.8 TB/s with the default cache
//     policy and 4.75 TB/s non-temporal; a plain streaming copy reaches 4.5 TB/s (reads alone 6.3, writes alone 4.4).
// ---------------------------------------------------------------------------------
constexpr int kOwn8Bins = 128;
constexpr int kOwn8BoxInts = 16;
constexpr int kOwn8MaxLen = 2048;
constexpr int kOwn8Threads = 384;       // >= kOwn8Bins, a multiple of 64: six waves, one ray per lane up to 384 beams
constexpr int kOwn8Batch = 8;           // window rows a wave keeps in flight in the sweep (4: 15 % slower, 12: 10 % slower; 512 threads: equal)
// LDS of a workgroup: half a CU's 160 KiB.  (Three workgroups per CU with 52.5 KiB each - the benchmark scan's box just fit -
// were measured no faster: the kernel goes at the pace of the memory system, not of the workgroups in flight.)
constexpr int kOwn8LdsBytes = 81920;
// cache policy of the sweep's counter and pmap traffic (buffer aux bits: 1 sc0, 2 nt, 16 sc1): every byte is read and
// written once per launch, and non-temporal loads AND stores run the in-place read-modify-write 24 % faster than the
// default policy (tools/ubench_rmw.hip: 4.75 against 3.83 TB/s on this access pattern; nt on the stores alone: no gain)
constexpr int kOwn8Aux = 2;
constexpr int kOwn8PerCU = 163840 / kOwn8LdsBytes;      // workgroups per CU the register budget is set for
__host__ __device__ inline size_t own8_fixed_bytes(int threads) { return (size_t)(kOwn8BoxInts + kOwn8Bins) * 4 + (size_t)threads * 4; }

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kBufferRsrcWord3 = 0x00020000;     // raw buffer, 32-bit data format (gfx9 / CDNA)
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ void lds_or_u32(unsigned addr, unsigned v)
{
    (void)__hip_atomic_fetch_or((lds_u32_t *)(uintptr_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// New value of a quad's four pmap bytes (mapping.py:47-50) from their old value and what the scan did to the cells (one bit
// per byte - 7: touched, 6: hit, 5: pass count at or over the threshold now): a touched cell shows 100 if it was occupied
// (bit 6 is set in 100 only, not in 0 or 50), is hit now or is over the threshold, else 0; an untouched cell keeps its byte.
__device__ __forceinline__ uint32_t own8_pmap_bytes(uint32_t o, uint32_t e)
{
    const uint32_t tb = (e >> 7) & 0x01010101u;
    const uint32_t occ = ((o >> 6) | (e >> 6) | (e >> 5)) & 0x01010101u;
    const uint32_t tm = tb * 255u;
    return (o & ~tm) | ((occ * 100u) & tm);
}

template <class Src, int THREADS, int BATCH>
__global__ void __launch_bounds__(THREADS, (kOwn8PerCU * THREADS + 255) / 256) k_grid_update_owner8(GridDev g, Src src, int win_bytes, int32_t *__restrict__ redo)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *box = reinterpret_cast<int *>(smem);                        // [0..3] bounding box, [4] odd case seen, [5] cells the rays pass, [6] window byte sum
    int *hist = box + kOwn8BoxInts;                                  // [kOwn8Bins] rays per length bin
    unsigned *slots = reinterpret_cast<unsigned *>(hist + kOwn8Bins);   // [THREADS] end cells in sorted order
    unsigned *win = slots + THREADS;                                 // [W][Hs] bytes
    char *guard = reinterpret_cast<char *>(win) + win_bytes;
    lds_guard_fill(guard);
    STAMP_DECL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int l = blockIdx.y, gi = src.own_grid(l), n = src.n;
    auto give_up = [&]() {                                           // (uniform) leave the map to the general kernel
        if (tid == 0) redo[2 + atomicAdd(&redo[0], 1)] = l;
    };

    // this lane's beam: loads first, LDS set-up while they are in flight
    typename Src::Beam beam = src.fetch(l, 0, tid < n ? tid : 0);
    ScanConst c0;
    src.scan_const(l, 0, g, c0);
    if (tid < kOwn8BoxInts) box[tid] = tid < 2 ? INT_MAX : (tid < 4 ? INT_MIN : 0);
    if (tid < kOwn8Bins) hist[tid] = 0;
    {
        uint4 *w4 = reinterpret_cast<uint4 *>(win);
        for (int w = tid; w < (win_bytes >> 4); w += THREADS) w4[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    STAMP(0);

    // pass 1: end cell, length bin, bounding box (mapping.py:30-36)
    int pox = 0, poy = 0, b2 = 0;
    const bool ok = tid < n && src.ray(beam, c0, g, pox, poy, b2);
    const bool inmap = (unsigned)c0.pcx < (unsigned)g.xw && (unsigned)c0.pcy < (unsigned)g.yw &&
                       (unsigned)pox < (unsigned)g.xw && (unsigned)poy < (unsigned)g.yw;
    // int() would raise / the ray leaves the map / a ray longer than the lengths for which "the float walk ends
    // in the cell of its other end" has been checked exhaustively (dx <= 2100: tests/test_walk_ends.py)
    const int len = max(abs(pox - c0.pcx), abs(poy - c0.pcy));
    const bool odd = (tid < n && b2 != 0) || (ok && (!inmap || len > kOwn8MaxLen));
    const bool valid = ok && inmap && len > 0 && len <= kOwn8MaxLen;   // identical cells: empty path (bresenham.py:10-11)
    int bin = 0, idx = 0;
    if (valid) {
        bin = kOwn8Bins - 1 - min(len >> 2, kOwn8Bins - 1);          // longest first
        idx = atomicAdd(&hist[bin], 1);
    }
    const unsigned pk = (unsigned)pox | ((unsigned)poy << 16);       // (both < 65 536 when valid)
    {
        unsigned lo = valid ? pk : 0xffffffffu, hi = valid ? pk : 0u;
        int cells = valid ? len - 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = pk_min_u16(lo, (unsigned)__shfl_xor((int)lo, off, kWave));
            hi = pk_max_u16(hi, (unsigned)__shfl_xor((int)hi, off, kWave));
            cells += __shfl_xor(cells, off, kWave);
        }
        const unsigned long long anyv = __ballot(valid), anyodd = __ballot(odd);
        if (lane == 0) {
            if (anyv) {
                atomicMin(&box[0], (int)(lo & 0xffffu)); atomicMin(&box[1], (int)(lo >> 16));
                atomicMax(&box[2], (int)(hi & 0xffffu)); atomicMax(&box[3], (int)(hi >> 16));
                atomicAdd(&box[5], cells);
            }
            if (anyodd) box[4] = 1;
        }
    }
    __syncthreads();
    STAMP(1);
    if (box[4]) { give_up(); return; }
    // counting sort: every wave scans the histogram for itself (two bins per lane)
    const int h0 = hist[2 * lane], h1 = hist[2 * lane + 1];
    int inc = h0 + h1;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) { const int v = __shfl_up(inc, off, kWave); if (lane >= off) inc += v; }
    const int nv = __shfl(inc, kWave - 1, kWave);                    // non-empty rays
    if (nv == 0) return;                                             // nothing to cast (mapping.py:38-39: empty paths)
    const int x0 = min(box[0], c0.pcx), y0 = min(box[1], c0.pcy), x1 = max(box[2], c0.pcx), y1 = max(box[3], c0.pcy);
    // window rows of whole 64-byte pieces of the counter rows (16 cells = 4 quads = 4 lanes of the sweep): a piece
    // the scan touched is read and written back whole - runs that start or end inside a piece cost the memory
    // system 11 % of its rate for this traffic (tools/ubench_rmw.hip: 4.29 against 4.84 TB/s)
    const int y4 = y0 & ~15, Hs = ((y1 | 15) + 1) - y4, W = x1 - x0 + 1;
    if ((long)W * Hs > (long)win_bytes) { give_up(); return; }
    {
        const int excl = inc - (h0 + h1);
        int base = __shfl(excl, bin >> 1, kWave);
        const int first = __shfl(h0, bin >> 1, kWave);
        if (bin & 1) base += first;
        if (valid) slots[base + idx] = pk;
    }
    __syncthreads();
    STAMP(2);

    // this lane's ray: the tid-th longest
    const bool have = tid < nv;
    const unsigned e = have ? slots[tid] : 0u;
    const int ex = (int)(e & 0xffffu), ey = (int)(e >> 16);
    Ray rr;
    rr.x0 = rr.y0 = 0; rr.dx = 1; rr.ystep = 1; rr.derr = 0.0; rr.steep = rr.flag = false;
    if (have) (void)ray_setup(c0.pcx, c0.pcy, ex, ey, rr);
    const unsigned wbase = lds_addr(win);
    if (have) {                                                      // the path's last cell is the end cell: the hit (mapping.py:44-45)
        const unsigned b = (unsigned)((ex - x0) * Hs + (ey - y4));
        lds_or_u32(wbase + (b & ~3u), 0x80u << ((b & 3u) * 8u));
    }
    {
        // the walk (bresenham.py:45-55) without its two end steps: walk step 0 is the origin cell (or, for a
        // reversed path, the hit cell) and step dx the hit cell (or the origin): dx - 1 cells in between
        const int lx = rr.steep ? rr.y0 : rr.x0, ly = rr.steep ? rr.x0 : rr.y0;
        unsigned a = wbase + (unsigned)((lx - x0) * Hs + (ly - y4));
        const int da_k = rr.steep ? 1 : Hs, da_y = rr.steep ? rr.ystep * Hs : rr.ystep;
        double error = 0.0;                                          // bresenham.py:34
        int rem = have ? rr.dx - 1 : 0;
        auto advance = [&]() {
            error += rr.derr;                                        // :51
            const bool stepy = error >= 0.5;                         // :53
            a += (unsigned)(da_k + (stepy ? da_y : 0));
            error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);    // :55 (minus 1.0, or minus 0.0: exact)
        };
        auto step = [&]() {
            lds_add_u32(a & ~3u, 1u << ((a << 3) & 31u));            // mapping.py:43
            advance();
        };
        if (have) advance();
        for (;;) {                                                   // four steps per wave-wide test, no lane mask inside
            const bool full = rem >= 4;
            if (!__any(full)) break;
            if (full) { step(); step(); step(); step(); rem -= 4; }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (rem > u) step();
    }
    __syncthreads();
    STAMP(3);

    // From here on a wave works on whole window rows (map x = x0 + r), 64 quads of a row at a time (a quad: 4 cells,
    // one window dword, 16 bytes of counters, 4 of pmap): row and segment are wave-uniform, so the window index and the
    // map addresses are a scalar base plus a per-lane constant and cost no vector instruction.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = THREADS / kWave;
    const int qrow = Hs >> 2, nseg = (qrow + kWave - 1) >> 6;

    // the check: masked byte sum of the window == cells the rays passed
    {
        unsigned acc = 0u;
        for (int seg = 0; seg < nseg; ++seg) {
            const int c = seg * kWave + lane;
            if (c < qrow)
                for (int r = wv; r < W; r += NW) acc = __builtin_amdgcn_sad_u8(win[r * qrow + c] & 0x7f7f7f7fu, 0u, acc);
        }
        acc = wave_sum_u32(acc);
        if (lane == 0 && acc) atomicAdd(&box[6], (int)acc);
    }
    __syncthreads();
    if (box[6] != box[5]) { give_up(); return; }

    uint32_t *pass = g.pass + (size_t)gi * g.xw * g.yw, *hit = g.hit + (size_t)gi * g.xw * g.yw;
    int8_t *pm = g.pmap_live + (size_t)gi * g.xw * g.yw;
    if (tid == 0) atomicAdd(visit_slot(g.visits), (unsigned long long)(box[5] + 2 * nv));   // every cell of every path is in the map (:41)

    // sweep.  Counters and pmap go through buffer loads / stores: the map's row is the scalar offset, the lane's quad
    // the vector offset, and a lane with nothing to do gets an offset beyond the buffer - the hardware then neither
    // reads nor writes for it - so a batch of rows is straight-line code without branches or lane masks.
    const uint32_t pthr = g.pass_thresh[0];
    const int org_r = c0.pcx - x0, org_q = (c0.pcy - y4) >> 2;       // the origin cell's row, quad ...
    const unsigned org_b = (unsigned)(c0.pcy - y4) & 3u;             // ... and byte in the quad
    const __amdgpu_buffer_rsrc_t rs_pass = __builtin_amdgcn_make_buffer_rsrc(pass, 0, g.xw * g.yw * 4, kBufferRsrcWord3);
    const __amdgpu_buffer_rsrc_t rs_pm = __builtin_amdgcn_make_buffer_rsrc(pm, 0, g.xw * g.yw, kBufferRsrcWord3);
    constexpr unsigned kSkip = 0x80000000u;                          // (beyond any map: xw * yw * 4 < 2^31, checked by the launcher)
    for (int seg = 0; seg < nseg; ++seg) {
        const int c = seg * kWave + lane;
        const bool incol = c < qrow;
        const bool org_lane = c == org_q;
        for (int r0 = wv; r0 < W; r0 += NW * BATCH) {
            u32x4_t p[BATCH];
            uint32_t om[BATCH], dd[BATCH], vo[BATCH], vp[BATCH], ev[BATCH];
            auto rowoff_of = [&](int u) -> uint32_t {                // (wave-uniform) first cell of the segment in map row x0 + r
                const int rc = min(r0 + u * NW, W - 1);
                return (uint32_t)((x0 + rc) * g.yw + y4 + seg * 4 * kWave);
            };
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = r0 + u * NW;                           // (wave-uniform)
                const int rc = min(r, W - 1);
                dd[u] = (incol && r < W) ? win[rc * qrow + c] : 0u;
                // the four lanes of a piece decide together (rows start on a piece, 64 lanes are 16 pieces)
                const unsigned long long m = __ballot(dd[u] != 0u || (r == org_r && org_lane));
                const bool live = ((m >> (lane & 60)) & 0xFull) != 0ull && incol;
                vo[u] = live ? (unsigned)lane << 4 : kSkip;         // byte offset of the lane's quad in the counter row
                p[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_pass, vo[u], rowoff_of(u) << 2, kOwn8Aux);
            }
            // pmap (mapping.py:47-50).  A touched cell shows 0 or 100 afterwards; its byte CHANGES only if the cell was never
            // touched before (then its old pass count is 0), or is hit now and was not occupied, or its pass count is at or
            // over the threshold now.  The old bytes are read where they can matter.  Up front: quads a hit fell in.  The
            // other two cases are known once the counters are here and are rare once a map has seen a scan or two: those
            // quads read pmap late (below).  The up-front reads are issued BEHIND the batch's counter loads: a wave's loads
            // return in order, and a lone lane's pmap line coming from memory between two rows' counters held every later
            // row of the batch back.  (Per 10 000 maps: pmap read for every live quad, as until round 3, 0.671 ms and
            // 3.20 GB; for the quads that need it, each between its row's counters and the next row's, 0.722 ms; behind the
            // counter loads 0.640 ms and 2.94 GB; listed in LDS and settled after the sweep in one round 0.718 ms - that
            // round trip is paid at the end of every workgroup's life, with nothing left to overlap it.)
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                vp[u] = (vo[u] != kSkip && (dd[u] & 0x80808080u) != 0u) ? (unsigned)lane << 2 : kSkip;
                om[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_pm, vp[u], rowoff_of(u), kOwn8Aux);
            }
            bool late_any = false;
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int r = r0 + u * NW;
                const uint32_t d = dd[u], dm = d & 0x7f7f7f7fu;
                u32x4_t q = p[u];
                // one bit per cell (bit 7 of its byte): its pass count was 0 before this scan
                const uint32_t z7 = (q.x == 0u ? 0x80u : 0u) | (q.y == 0u ? 0x8000u : 0u) | (q.z == 0u ? 0x800000u : 0u) | (q.w == 0u ? 0x80000000u : 0u);
                q.x += dm & 0xffu; q.y += (dm >> 8) & 0xffu; q.z += (dm >> 16) & 0xffu; q.w += dm >> 24;   // mapping.py:43
                // one bit per cell (bit 7 of its byte): passed or hit by this scan
                uint32_t t7 = ((dm + 0x7f7f7f7fu) | d) & 0x80808080u;
                if (r == org_r) {                                    // (wave-uniform) every non-empty path starts in the origin cell
                    if (org_lane) {
                        q.x += org_b == 0u ? (uint32_t)nv : 0u; q.y += org_b == 1u ? (uint32_t)nv : 0u;
                        q.z += org_b == 2u ? (uint32_t)nv : 0u; q.w += org_b == 3u ? (uint32_t)nv : 0u;
                        t7 |= 0x80u << (8u * org_b);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(q, rs_pass, vo[u], rowoff_of(u) << 2, kOwn8Aux);
                // what the scan did to the quad's cells, one bit per byte - 7: touched, 6: hit, 5: count at or over the threshold
                const uint32_t ge5 = (q.x >= pthr ? 0x20u : 0u) | (q.y >= pthr ? 0x2000u : 0u) | (q.z >= pthr ? 0x200000u : 0u) | (q.w >= pthr ? 0x20000000u : 0u);
                ev[u] = t7 | ((d >> 1) & 0x40404040u) | ge5;
                if (vp[u] == kSkip && vo[u] != kSkip && ((t7 & z7) != 0u || (ge5 & (t7 >> 2)) != 0u)) {
                    vp[u] = ((unsigned)lane << 2) | 1u;              // (bit 0: read late)
                    late_any = true;
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {                        // the quads read up front
                if (vp[u] == kSkip || (vp[u] & 1u)) continue;
                const uint32_t out = own8_pmap_bytes(om[u], ev[u]);
                if (out != om[u]) __builtin_amdgcn_raw_buffer_store_b32(out, rs_pm, vp[u], rowoff_of(u), kOwn8Aux);
            }
            if (__any(late_any)) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    om[u] = __builtin_amdgcn_raw_buffer_load_b32(rs_pm, (vp[u] != kSkip && (vp[u] & 1u)) ? vp[u] & ~3u : kSkip, rowoff_of(u), kOwn8Aux);
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    if (vp[u] == kSkip || !(vp[u] & 1u)) continue;
                    const uint32_t out = own8_pmap_bytes(om[u], ev[u]);
                    if (out != om[u]) __builtin_amdgcn_raw_buffer_store_b32(out, rs_pm, vp[u] & ~3u, rowoff_of(u), kOwn8Aux);
                }
            }
        }
    }
    // the hits last (mapping.py:45): fire-and-forget atomics, which the sweep's loads would otherwise queue behind
    if (have) atomicAdd(&hit[(size_t)ex * g.yw + ey], 1u);
    STAMP_SYNC();
    STAMP(4);
    STAMP_END(5);
    lds_guard_check(guard, g.status);
}

// CUs and LDS bytes per CU of the current device (cached per device): the launch shapes below count workgroups against them
struct ChipShape { int cus; long lds_per_cu; };
static ChipShape chip_shape()
{
    static std::mutex mu;
    static std::vector<std::pair<int, ChipShape>> seen;
    int dev = 0;
    ChipShape c{256, 160 * 1024};                                    // MI355X, should a query fail
    if (hipGetDevice(&dev) != hipSuccess) return c;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &e : seen)
        if (e.first == dev) return e.second;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) c.cus = v;
    // (gfx950 reports 160 KiB here; a workgroup may ask for all of it)
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) c.lds_per_cu = v;
    (void)hipGetLastError();
    seen.push_back({dev, c});
    return c;
}

template <class Src>
static hipError_t launch_win(const GridDev &g, const Src &src, int L, int scans, int n, int group, const int32_t *got,
                             hipStream_t s, int split_pref = -1)
{
    const size_t lds_max = win_lds_bytes(kWinMaxGroup, kMaxSortRays, kWinCells);
    {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(&k_grid_update_win<Src>), (int)lds_max);
        if (e != hipSuccess) return e;
    }
    int groups = (scans + group - 1) / group;
    long rays = (long)group * n;
    long want = (rays + kRaysPerLane - 1) / kRaysPerLane;
    int threads = want >= kMaxWaves * kWave ? kMaxWaves * kWave : (int)(((want + kWave - 1) / kWave) * kWave);
    if (threads < 128) threads = 128;
    // one workgroup per map and no other writer: single stream (L == 1) or one map per stream
    const int exclusive = g.pmap_live && groups == 1 && !got && (L == 1 || src.maps_are_private());
    if (g.pmap_live && !exclusive && g.live_dirty) *g.live_dirty = true;
    // the window takes most of a CU's LDS, so a CU runs one or two workgroups: the flush / the
    // exclusive sweep (a streaming pass over the touched rectangle) need lanes for memory
    // operations in flight even when there are few rays (measured on 10 000 single-scan groups:
    // 2.88 ms with 128 threads, 2.05 ms with 512)
    constexpr int kWinMinThreads = 512;
    if (threads < kWinMinThreads) threads = kWinMinThreads;
    const int sort_cap = win_sort_cap((long)group * n);
    const int win_cells = win_cells_for(group, sort_cap);
    // one scan per map, cast by the map's only writer, live pmap, the reference's one-hit-occupies rule
    if (exclusive && group == 1 && scans == 1 && g.hit_levels == 1 && (g.yw & 15) == 0 && (((size_t)g.xw * g.yw) & 3) == 0 &&
        g.yw <= win_cells && n <= kOwnerMaxRays * kOwnerThreads) {
        for (const void *f : {reinterpret_cast<const void *>(&k_grid_update_owner<Src, 1>), reinterpret_cast<const void *>(&k_grid_update_owner<Src, 2>),
                              reinterpret_cast<const void *>(&k_grid_update_owner_redo<Src, 1>)}) {
            hipError_t e = allow_dynamic_lds(f, (int)lds_max);
            if (e != hipSuccess) return e;
        }
        const size_t lds = win_lds_bytes(1, sort_cap, win_cells);
        // the plain case (all of a closed room's scans) goes through the byte-window kernel, three workgroups per CU;
        // whatever it leaves on the re-do list is cast by the general kernel behind it
        if (g.redo && n <= kOwn8Threads && n <= kOwnerThreads && g.xw <= 65535 && g.yw <= 65535 && (long)g.xw * g.yw < (1L << 29)) {
            const int own8_lds = kOwn8LdsBytes + kLdsGuard, own8_win = (int)(kOwn8LdsBytes - own8_fixed_bytes(kOwn8Threads));
            SLAM_LAUNCH((k_grid_update_owner8<Src, kOwn8Threads, kOwn8Batch>), dim3(1, L), dim3(kOwn8Threads), own8_lds, s,
                        g, src, own8_win, g.redo);
            SLAM_LAUNCH((k_grid_update_owner_redo<Src, 1>), dim3(std::min(L, 256)), dim3(kOwnerThreads), lds, s, g, src, sort_cap, win_cells, g.redo);
            return hipGetLastError();
        }
        if (n <= kOwnerThreads) SLAM_LAUNCH((k_grid_update_owner<Src, 1>), dim3(1, L), dim3(kOwnerThreads), lds, s, g, src, sort_cap, win_cells);
        else                    SLAM_LAUNCH((k_grid_update_owner<Src, 2>), dim3(1, L), dim3(kOwnerThreads), lds, s, g, src, sort_cap, win_cells);
        return hipGetLastError();
    }
    // Two workgroups per group where the rays are sorted and the map is not the workgroup's own - a group whose box fits one
    // window is shared by beam parity, else one workgroup takes the rays right of the origins' column and one those left of it
    // (k_grid_update_win): a launch that cannot fill the chip on its own (a 1 000-scan replay is 125 groups on 256 CUs) runs
    // 11 % shorter that way (0.088 -> 0.078 ms); when launches of several contexts share the chip the duplicated first pass
    // costs 4 % of the throughput, so callers that overlap replays switch it off (context option "grid_split")
    const ChipShape chip = chip_shape();
    const int split = (!exclusive && sort_cap > 0 && (split_pref > 0 || (split_pref < 0 && (long)groups * L <= (3L * chip.cus) / 4))) ? 1 : 0;
    // A launch of at most one workgroup per CU asks for more than half a CU's LDS, so that no CU gets two of its workgroups
    // while others stay empty: the dispatcher does not spread a small grid by itself (stamps, round 4: of 250 workgroups on
    // 256 CUs the slowest took 1.8 x the mean with the same number of cells to walk - it shared its CU with another one).
    size_t lds = win_lds_bytes(group, sort_cap, win_cells);
    const long wgs = (long)(split ? 2 * groups : groups) * L;
    // (only where a CU's LDS really is more than twice a workgroup's need, and the larger request is one the device grants)
    const size_t half_cu = (size_t)(chip.lds_per_cu / 2);
    if (split && wgs <= chip.cus && lds <= half_cu && half_cu + 512 <= (size_t)chip.lds_per_cu) lds = half_cu + 512;
    if (lds > lds_max) {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(&k_grid_update_win<Src>), (int)lds);
        if (e != hipSuccess) return e;
    }
    SLAM_LAUNCH((k_grid_update_win<Src>), dim3(split ? 2 * groups : groups, L), dim3(threads), lds, s, g, src, group, got,
                exclusive, sort_cap, win_cells, split);
    return hipGetLastError();
}

// group == 0: automatic.  Measured on the 1k-scan replay (profiles/r01_*): the kernel is
// fastest with about one workgroup per two CUs (8 scans each); more, smaller groups re-flush
// the same map region more often, fewer leave CUs idle.
static int pick_group(int group, long total_scans, int scans_per_traj, int n)
{
    int gmax = std::min(kWinMaxGroup, 65535 / std::max(n, 1));
    if (gmax < 1) return 0;                                          // n too large for the packed counters
    if (group <= 0) {
        long want = (total_scans + 127) / 128;
        group = (int)std::min<long>(std::max<long>(want, 1), 12);   // (5 000-scan replays, four overlapping: 12 -> 9.3, 16 -> 8.9 M scans/s)
    }
    group = std::min(std::min(group, gmax), std::max(scans_per_traj, 1));
    return group;
}

hipError_t launch_grid_update_win(const GridDev &g, const double *ox, const double *oy, const double *cx, const double *cy,
                                  int B, int n, int group, hipStream_t s, int split_pref)
{
    int G = pick_group(group, B, B, n);
    if (G == 0) return launch_grid_update(g, ox, oy, cx, cy, B, n, nullptr, s);
    ExplicitSource src{ox, oy, cx, cy, B, n};
    return launch_win(g, src, 1, B, n, G, nullptr, s, split_pref);
}

hipError_t launch_grid_update_replay_win(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                         const double *poses, int L, int n_scan, int n, const int32_t *got, int group,
                                         hipStream_t s, int shared_scans, int grid_per_traj, const double *heading_cs, int split_pref)
{
    if (n_scan < 2) return hipSuccess;
    int G = pick_group(group, (long)L * (n_scan - 1), n_scan - 1, n);
    if (G == 0) {
        if (shared_scans || grid_per_traj) return hipErrorInvalidValue;   // n too large for the window kernel
        return launch_grid_update_replay(g, ranges, cos_t, sin_t, poses, L, n_scan, n, got, s);
    }
    ReplaySource src{ranges, cos_t, sin_t, poses, n_scan, n, shared_scans ? 0L : (long)n_scan * n, grid_per_traj, nullptr, heading_cs};
    return launch_win(g, src, L, n_scan - 1, n, G, got, s, split_pref);
}

// S scans cast from given poses, optionally with ray origins of their own.
hipError_t launch_grid_update_scans(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                    const double *poses, const double *centres, int S, int n, int group, hipStream_t s, int split_pref)
{
    int G = pick_group(group, S, S, n);
    if (G == 0) return hipErrorInvalidValue;                          // n too large for the window kernel
    // the replay source reads scan k+1 of a stream of n_scan = S+1: shift the base by one scan
    ReplaySource src{ranges - n, cos_t, sin_t, poses, S + 1, n, 0L, 0, centres};
    return launch_win(g, src, 1, S, n, G, nullptr, s, split_pref);
}

// ---------------------------------------------------------------------------------
// Tiled ray casting for maps far larger than one LDS window (DESIGN.md "K4 tiles"; the
// 2000x2000 @ 0.02 m map of BASELINE.json configs[4], where a scan's rays cover ~800 k cells
// and the window kernel fell back to scattered global atomics for most of them).
//
// The float-error Bresenham walk (bresenham.py:45-55) cannot be entered in the middle: its
// state is a rounded running sum.  So every ray is walked ONCE (k_ray_bits), touching no map
// cell, and the walk is recorded as one bit per step - "y advanced after this step" - plus a
// running count of set bits per 32-step word.  With those, the cell of step k is
// (x0 + k, y0 + ystep * popcount(bits[0..k))) for any k, so a second kernel (k_tile_cast) can
// give every (map tile, scan group) pair its own workgroup: it enters each ray where it
// crosses the tile, accumulates the tile's pass counts in LDS exactly like the window kernel,
// and flushes the non-zero cells row by row.  The cells visited are the reference's by
// construction (same walk, recorded instead of applied).  Hits (one per ray) and the visit
// counter are handled by the first kernel; rays of 2048 steps or more take the direct path
// there.
// ---------------------------------------------------------------------------------
constexpr int kTileSteps = 2048;                 // rays with dx < kTileSteps are recorded
constexpr int kTileWords = kTileSteps / 32;
constexpr int kTileSide = 192;        // 192 x 192 16-bit cells = the 72 KiB window

struct RayRec {
    int x0, y0, dx, yend;                        // walk coordinates; yend = y of the cell at step dx
    uint32_t flags;                              // 1: steep, 2: reversed (path runs end -> start), 4: ystep > 0, 8: valid
    int pad[3];
};

struct TileScratch {
    RayRec *recs;                                // [rays]
    uint32_t *bits;                              // [rays][kTileWords]
    unsigned short *prefix;                      // [rays][kTileWords]: set bits before each word
    int *gbox;                                   // [groups][4]: map-space bounding box of a group's recorded rays
};

__global__ void __launch_bounds__(256) k_tile_init(int *gbox, int groups)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < groups) { gbox[4 * i] = gbox[4 * i + 1] = INT_MAX; gbox[4 * i + 2] = gbox[4 * i + 3] = INT_MIN; }
}

// Pass A: block (scan, stream) walks the scan's rays once.
template <class Src>
__global__ void __launch_bounds__(256) k_ray_bits(GridDev g, Src src, TileScratch ts, int group_size)
{
    __shared__ ScanConst sc;
    __shared__ int box[4];
    const int s = blockIdx.x, l = blockIdx.y, n = src.n;
    const int scans = src.scans_per_traj();
    const int groups_per_traj = (scans + group_size - 1) / group_size;
    const int group = l * groups_per_traj + s / group_size;
    uint32_t *pass = g.pass, *hit = g.hit;       // single shared map (the launcher guarantees it)
    __shared__ int first_bad;
    if (threadIdx.x == 0) {
        src.scan_const(l, s, g, sc);
        box[0] = box[1] = INT_MAX; box[2] = box[3] = INT_MIN;
        first_bad = INT_MAX;
    }
    __syncthreads();
    // the scan stops at its first beam that Python's int() would raise on (mapping.py:29-36: the beams before it have been
    // applied when the exception leaves update(), and the error is that beam's): find it before anything is recorded or cast
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int pox, poy, b2 = 0;
        (void)src.ray(l, s, i, sc, g, pox, poy, b2);
        if (b2) atomicMin(&first_bad, i);
    }
    __syncthreads();
    const int stop = first_bad;
    unsigned nvis = 0;
    int bad = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const size_t rid = ((size_t)l * scans + s) * n + i;
        RayRec rec;
        rec.flags = 0; rec.x0 = rec.y0 = rec.dx = rec.yend = 0; rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
        int pox, poy, b2 = 0;
        Ray ry;
        const bool valid = i <= stop && src.ray(l, s, i, sc, g, pox, poy, b2) && ray_setup(sc.pcx, sc.pcy, pox, poy, ry);
        // the common ray: recorded, and both ends (hence every cell) inside the map.  Only the walk's
        // decisions are needed - one bit per step - so the loop is the error recurrence alone
        // (bresenham.py:51-55), all lanes at the same step; the path's last cell is the endpoint cell
        // itself and takes its hit without being walked to (mapping.py:44-45).
        const bool plain = valid && ry.dx < kTileSteps && (unsigned)sc.pcx < (unsigned)g.xw && (unsigned)sc.pcy < (unsigned)g.yw &&
                           (unsigned)pox < (unsigned)g.xw && (unsigned)poy < (unsigned)g.yw;
        {
            uint32_t *bw = ts.bits + rid * kTileWords;
            unsigned short *pw = ts.prefix + rid * kTileWords;
            const int dx = plain ? ry.dx : -1;
            const double derr = plain ? ry.derr : 0.0;
            double error = 0.0;
            uint32_t count = 0, before_last = 0;
            for (int k0 = 0; __any(k0 <= dx); k0 += 32) {
                if (k0 <= dx) {
                    uint32_t word = 0;
                    const int nb = min(32, dx - k0 + 1);
                    // (whole words: the steps past the ray's end are computed and masked off - nothing reads
                    // the recurrence after them - so there is no lane mask per step)
#pragma unroll 8
                    for (int b = 0; b < 32; ++b) {
                        error += derr;                                // bresenham.py:51
                        const bool stepy = error >= 0.5;              // :53
                        error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);   // :55 (minus 1.0, or minus 0.0: exact)
                        word |= stepy ? (1u << b) : 0u;
                    }
                    if (nb < 32) word &= (1u << nb) - 1u;
                    pw[k0 >> 5] = (unsigned short)count;
                    bw[k0 >> 5] = word;
                    if (k0 + 32 > dx) before_last = count + __popc(word & ((1u << (dx - k0)) - 1u));   // y advances before step dx
                    count += __popc(word);
                }
            }
            if (plain) {
                nvis += (unsigned)ry.dx + 1u;                         // mapping.py:41: every cell is in the map
                atomicAdd(&hit[(size_t)pox * g.yw + poy], 1u);
                rec.yend = ry.y0 + ry.ystep * (int)before_last;
                rec.x0 = ry.x0; rec.y0 = ry.y0; rec.dx = ry.dx;
                rec.flags = 8u | (ry.steep ? 1u : 0u) | (ry.flag ? 2u : 0u) | (ry.ystep > 0 ? 4u : 0u);
                int a0 = ry.x0, a1 = ry.x0 + ry.dx, b0 = min(ry.y0, rec.yend), b1 = max(ry.y0, rec.yend);
                int mx0 = ry.steep ? b0 : a0, mx1 = ry.steep ? b1 : a1, my0 = ry.steep ? a0 : b0, my1 = ry.steep ? a1 : b1;
                atomicMin(&box[0], mx0); atomicMin(&box[1], my0); atomicMax(&box[2], mx1); atomicMax(&box[3], my1);
            }
        }
        if (valid && !plain) {                                       // leaves the map, or too long to record: the step-by-step form
            const bool record = ry.dx < kTileSteps;
            const int klast = ry.flag ? 0 : ry.dx;
            uint32_t *bw = ts.bits + rid * kTileWords;
            unsigned short *pw = ts.prefix + rid * kTileWords;
            double error = 0.0;
            int y = ry.y0, hx = -1, hy = -1;
            uint32_t word = 0, count = 0;
            for (int k = 0; k <= ry.dx; ++k) {
                if (record && (k & 31) == 0) pw[k >> 5] = (unsigned short)count;
                int x = ry.x0 + k;
                int lx = ry.steep ? y : x, ly = ry.steep ? x : y;
                bool inmap = (unsigned)lx < (unsigned)g.xw && (unsigned)ly < (unsigned)g.yw;
                nvis += inmap ? 1u : 0u;
                if (k == klast) { hx = lx; hy = ly; }
                else if (!record && inmap) atomicAdd(&pass[(size_t)lx * g.yw + ly], 1u);
                if (k == ry.dx) rec.yend = y;
                error += ry.derr;
                bool stepy = error >= 0.5;
                if (stepy) { y += ry.ystep; error -= 1.0; word |= 1u << (k & 31); ++count; }
                if (record && ((k & 31) == 31 || k == ry.dx)) { bw[k >> 5] = word; word = 0; }
            }
            if ((unsigned)hx < (unsigned)g.xw && (unsigned)hy < (unsigned)g.yw) atomicAdd(&hit[(size_t)hx * g.yw + hy], 1u);
            if (record) {
                rec.x0 = ry.x0; rec.y0 = ry.y0; rec.dx = ry.dx;
                rec.flags = 8u | (ry.steep ? 1u : 0u) | (ry.flag ? 2u : 0u) | (ry.ystep > 0 ? 4u : 0u);
                int a0 = ry.x0, a1 = ry.x0 + ry.dx, b0 = min(ry.y0, rec.yend), b1 = max(ry.y0, rec.yend);
                int mx0 = ry.steep ? b0 : a0, mx1 = ry.steep ? b1 : a1, my0 = ry.steep ? a0 : b0, my1 = ry.steep ? a1 : b1;
                atomicMin(&box[0], mx0); atomicMin(&box[1], my0); atomicMax(&box[2], mx1); atomicMax(&box[3], my1);
            }
        }
        bad |= b2;
        ts.recs[rid] = rec;
    }
    __syncthreads();
    if (threadIdx.x == 0 && box[0] <= box[2]) {
        atomicMin(&ts.gbox[4 * group], box[0]); atomicMin(&ts.gbox[4 * group + 1], box[1]);
        atomicMax(&ts.gbox[4 * group + 2], box[2]); atomicMax(&ts.gbox[4 * group + 3], box[3]);
    }
    unsigned tot = wave_sum_u32(nvis);
    int anybad = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anybad |= __shfl_xor(anybad, off, kWave);
    if ((threadIdx.x & 63) == 0) {
        if (tot) atomicAdd(visit_slot(g.visits), (unsigned long long)tot);
        if (anybad) atomicOr(g.status, anybad);
    }
}

// Pass B: workgroup (tile, group): the pass counts of one map tile from one group of scans.
__global__ void __launch_bounds__(1024) k_tile_cast(GridDev g, TileScratch ts, int tiles_x, int rays_per_group, long total_rays)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned *win = reinterpret_cast<unsigned *>(smem);
    __shared__ int next_ray;
    const int tile = blockIdx.x, group = blockIdx.y;
    const int tx0 = (tile % tiles_x) * kTileSide, ty0 = (tile / tiles_x) * kTileSide;
    const int W = min(kTileSide, g.xw - tx0), H = min(kTileSide, g.yw - ty0);
    const int tx1 = tx0 + W - 1, ty1 = ty0 + H - 1;
    const int *gb = ts.gbox + 4 * group;
    if (gb[0] > tx1 || gb[2] < tx0 || gb[1] > ty1 || gb[3] < ty0) return;   // nothing of this group comes near the tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int Hp2 = (H + 1) >> 1;
    const int Hs = Hp2 | 1;                      // odd row stride: rows start in different LDS banks
    char *guard = smem + (size_t)kTileSide * ((kTileSide / 2) | 1) * 4;
    lds_guard_fill(guard);
    for (int w = tid; w < W * Hs; w += blockDim.x) win[w] = 0u;
    if (tid == 0) next_ray = 0;
    __syncthreads();
    const long rbase = (long)group * rays_per_group;
    const int nrays = (int)min((long)rays_per_group, total_rays - rbase);
    uint32_t *pass = g.pass;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&next_ray, kWave);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= nrays) break;
        if (base + lane >= nrays) continue;
        const size_t rid = (size_t)(rbase + base + lane);
        const RayRec rec = ts.recs[rid];
        if (!(rec.flags & 8u)) continue;
        const bool steep = rec.flags & 1u;
        const int ystep = (rec.flags & 4u) ? 1 : -1;
        const int klast = (rec.flags & 2u) ? 0 : rec.dx;
        // tile range along the walk axis (a) and the other axis (b), in walk coordinates
        const int a0 = steep ? ty0 : tx0, a1 = steep ? ty1 : tx1, b0 = steep ? tx0 : ty0, b1 = steep ? tx1 : ty1;
        const int k0 = max(0, a0 - rec.x0), k1 = min(rec.dx, a1 - rec.x0);
        if (k0 > k1 || min(rec.y0, rec.yend) > b1 || max(rec.y0, rec.yend) < b0) continue;
        const uint32_t *bw = ts.bits + rid * kTileWords;
        int w = k0 >> 5;
        uint32_t word = bw[w];
        int y = rec.y0 + ystep * ((int)ts.prefix[rid * kTileWords + w] + __popc(word & ((1u << (k0 & 31)) - 1u)));
        word >>= (k0 & 31);
        // The cell of walk step k as the LDS byte address of its 16-bit counter, advanced incrementally (while
        // the ray is outside the tile's rows the address is virtual: it is only used inside them).  The path's
        // last cell takes the hit and no pass count (mapping.py:44-45): it is the ray's first or last step and is
        // trimmed off the range here.
        const int HW = 2 * Hs;                                       // 16-bit counters per window row
        const int da_k = 2 * (steep ? 1 : HW), da_y = 2 * (steep ? ystep * HW : ystep);
        unsigned a2 = lds_addr(win) + 2u * (unsigned)(steep ? (y - b0) * HW + (rec.x0 + k0 - a0) : (rec.x0 + k0 - a0) * HW + (y - b0));
        int k = k0, kk1 = k1 - (klast == k1 ? 1 : 0);
        if (klast == k0) {                                           // (klast == 0 == k0: the word holds bit 0 still)
            const unsigned bit = word & 1u;
            y += bit ? ystep : 0;
            a2 += (unsigned)(da_k + (bit ? da_y : 0));
            word >>= 1;
            ++k;
            if ((k & 31) == 0 && k <= kk1) word = bw[++w];           // (only for a one-bit first word: dx >= 1, so never; kept for safety)
        }
        bool gone = false;
        while (k <= kk1 && !gone) {
            const int kend = min(kk1, (w << 5) + 31);
            const uint32_t next = (w + 1) * 32 <= k1 ? bw[w + 1] : 0u;   // in flight while this word is walked
            const int nb = kend - k + 1;
            const uint32_t bits = nb == 32 ? word : (word & ((1u << nb) - 1u));
            const int pc = __popc(bits);
            const int yafter = y + ystep * pc;                       // y of the step after this word's last
            if (ystep > 0 ? yafter < b0 : yafter > b1) {             // the whole word stays short of the tile's rows
                y = yafter; k = kend + 1; a2 += (unsigned)(nb * da_k + pc * da_y); word = next; ++w;
                continue;
            }
            // every step of the word inside the rows (y is monotone: both ends inside): no checks, four steps
            // per wave-wide test
            const int ylast = yafter - ((bits >> (nb - 1)) & 1u ? ystep : 0);   // y AT the word's last step
            const bool inside = y >= b0 && y <= b1 && ylast >= b0 && ylast <= b1;
            if (__any(inside)) {
                int rem = inside ? nb : 0;
                auto step = [&]() {
                    lds_add_u32(a2 & ~3u, 1u << ((a2 << 3) & 31u));
                    a2 += (unsigned)(da_k + ((word & 1u) ? da_y : 0));
                    word >>= 1;
                };
                for (;;) {
                    const bool full = rem >= 4;
                    if (!__any(full)) break;
                    if (full) { step(); step(); step(); step(); rem -= 4; }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rem > u) step();
                if (inside) { y = yafter; k = kend + 1; word = next; ++w; }
            }
            if (!inside) {                                           // the word in which the ray enters or leaves the rows
                for (; k <= kend; ++k) {
                    if (ystep > 0 ? y > b1 : y < b0) { gone = true; break; }   // left the tile for good
                    if (y >= b0 && y <= b1) lds_add_u32(a2 & ~3u, 1u << ((a2 << 3) & 31u));
                    const unsigned bit = word & 1u;
                    y += bit ? ystep : 0;
                    a2 += (unsigned)(da_k + (bit ? da_y : 0));
                    word >>= 1;
                }
                word = next;
                ++w;
            }
        }
    }
    __syncthreads();
    const int rot = (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)W);
    for (int rr = wave; rr < W; rr += nwaves) {
        const int row = rr + rot < W ? rr + rot : rr + rot - W;
        size_t gbase = (size_t)(tx0 + row) * g.yw + ty0;
        if ((gbase & 1) == 0) {                             // pairs of counters are aligned 8-byte words (see the window kernel)
            for (int d = lane; d < Hp2; d += kWave) {
                unsigned v = win[row * Hs + d];
                if (v) atomicAdd(reinterpret_cast<unsigned long long *>(&pass[gbase + 2 * d]),
                                 (unsigned long long)(v & 0xffffu) | ((unsigned long long)(v >> 16) << 32));
            }
            continue;
        }
        for (int d = lane; d < Hp2; d += kWave) {
            unsigned v = win[row * Hs + d];
            unsigned p0 = v & 0xffffu, p1 = v >> 16;
            if (p0) atomicAdd(&pass[gbase + 2 * d], p0);
            if (p1) atomicAdd(&pass[gbase + 2 * d + 1], p1);
        }
    }
    lds_guard_check(guard, g.status);
}

size_t tile_scratch_bytes(long rays, long groups)
{
    // (groups: an upper bound is enough - one per scan)
    return (size_t)rays * (sizeof(RayRec) + kTileWords * 4 + kTileWords * 2) + (size_t)groups * 16 + 1024;
}

bool tiles_apply(const GridDev &g, int n, const int32_t *got, int grid_per_traj, int wedges)
{
    // maps much larger than a window; a group's ray count must fit the 16-bit counters.  The recorded-walk tiles cast into ONE
    // shared map; the direction wedges also take a map per trajectory (`got`: dense replays in batches, bench.py --config dense --traj)
    return (!got || wedges) && !grid_per_traj && (long)g.xw * g.yw > 8L * kWinCells && n <= kTileMaxBeams;
}

template <class Src>
static hipError_t launch_tiles(const GridDev &g, const Src &src, int L, int scans, int n, int group, void *scratch, hipStream_t s)
{
    if (scans < 1) return hipSuccess;
    // scans per (tile, group) workgroup: 16 by default (1080 beams, 2000 x 2000 cells, three overlapping replays:
    // 8 -> 0.79, 12 -> 0.85, 16 -> 0.86, 24 -> 0.88, 32 -> 0.83 M scans/s; alone 16 equals 8 and 32 is 14 % slower)
    int G = group > 0 ? group : 16;
    G = std::min(G, std::max(1, 65535 / n));
    G = std::min(G, scans);
    if ((long)G * n > 65535) return hipErrorInvalidValue;   // (the tile counters and a group's ray numbers are 16-bit)
    if (L > 1)                                    // a group is a contiguous range of ray ids: it must not straddle streams
        while (scans % G != 0) --G;
    const int groups_per_traj = (scans + G - 1) / G;
    const long groups = (long)L * groups_per_traj, rays = (long)L * scans * n;
    if (groups > 65535) return hipErrorInvalidValue;
    TileScratch ts;
    char *p = static_cast<char *>(scratch);
    ts.recs = reinterpret_cast<RayRec *>(p); p += (size_t)rays * sizeof(RayRec);
    ts.bits = reinterpret_cast<uint32_t *>(p); p += (size_t)rays * kTileWords * 4;
    ts.prefix = reinterpret_cast<unsigned short *>(p); p += (size_t)rays * kTileWords * 2;
    ts.gbox = reinterpret_cast<int *>(p);
    if (g.pmap_live && g.live_dirty) *g.live_dirty = true;
    // the family is three launches: bracket it with recorded events when timing is armed
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const bool timed = launch_events(&ev0, &ev1);
    if (timed) (void)hipEventRecord(ev0, s);
    hipLaunchKernelGGL(k_tile_init, dim3((groups + 255) / 256), dim3(256), 0, s, ts.gbox, (int)groups);
    hipLaunchKernelGGL((k_ray_bits<Src>), dim3(scans, L), dim3(256), 0, s, g, src, ts, G);
    const int tiles_x = (g.xw + kTileSide - 1) / kTileSide, tiles_y = (g.yw + kTileSide - 1) / kTileSide;
    size_t lds = (size_t)kTileSide * ((kTileSide / 2) | 1) * 4 + kLdsGuard;
    {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(&k_tile_cast), (int)lds);
        if (e != hipSuccess) return e;
    }
    // 1024 lanes: the kernel is bound by the latency of its loads and LDS atomics (measured 2.35 ms
    // with 512, 2.14 ms with 1024 on the 1080-beam / 2000x2000 replay; tiles of 128 or 224 cells are slower)
    hipLaunchKernelGGL(k_tile_cast, dim3(tiles_x * tiles_y, (unsigned)groups), dim3(1024), lds, s, g, ts, tiles_x, G * n, rays);
    if (timed) (void)hipEventRecord(ev1, s);
    return hipGetLastError();
}

template <class Src>
static hipError_t launch_wedges(const GridDev &g, const Src &src, int L, int scans, int n, int group, void *scratch, hipStream_t s, const int32_t *got = nullptr);

hipError_t launch_grid_update_tiles(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                    const double *poses, const double *centres, int L, int n_scan, int n, int group,
                                    void *scratch, hipStream_t s, int wedges, const int32_t *got)
{
    if (n_scan < 2) return hipSuccess;
    ReplaySource src{ranges, cos_t, sin_t, poses, n_scan, n, (long)n_scan * n, 0, centres};
    if (wedges) return launch_wedges(g, src, L, n_scan - 1, n, group, scratch, s, got);
    if (got) return hipErrorInvalidValue;                            // (the recorded-walk tiles know one shared map)
    return launch_tiles(g, src, L, n_scan - 1, n, group, scratch, s);
}

// explicit world-frame endpoints (Mapping.update's own arguments), B scans into one map
hipError_t launch_grid_update_tiles_explicit(const GridDev &g, const double *ox, const double *oy, const double *cx,
                                             const double *cy, int B, int n, int group, void *scratch, hipStream_t s, int wedges)
{
    ExplicitSource src{ox, oy, cx, cy, B, n};
    if (wedges) return launch_wedges(g, src, 1, B, n, group, scratch, s);
    return launch_tiles(g, src, 1, B, n, group, scratch, s);
}

// ---------------------------------------------------------------------------------
// Wedge ray casting for maps far larger than one LDS window (DESIGN.md "K4 wedges"): the rays of a group of
// scans are dealt to workgroups by DIRECTION - steep or not, pointing up or down the walk axis, slope class - so
// that a workgroup's rays fan out from (nearly) one origin inside a narrow wedge, and the wedge is swept in BANDS
// along the walk axis: zero the window, walk every ray through the band, flush, next band.  A ray's walk
// (bresenham.py:45-55) always ascends along its walk axis after the reference's endpoint swaps, which is the
// band axis of its own steepness class: every ray is walked exactly ONCE, start to end, its state (step, y, float
// error) resting in registers between bands - nothing is recorded, nothing re-entered.
// The window of a band is a PARALLELOGRAM: rows follow the wedge with an integer shear M in {-1, 0, 1} per step
// of the walk axis (the slope class's nearest integer), so the LDS index stays LINEAR in the cell,
//     index = (a - a_lo) * ca + (y - M a - v_lo) * cv          (a: walk axis, y: the other axis)
// and the unchecked walk of cast_rays applies unchanged with strides (ca - M cv) per step and (+-cv) per y step.
// A band takes as many rows as its parallelogram leaves room for (the wedge's width at the band's far end plus the
// drift |slope - M| <= 1/2 per row).  k_wedge_sort prepares the group: end cells, direction class and a counting
// sort by (class, length) so that lanes hold the longest rays of their class first; hits, the visit counter and
// the odd rays (leaving the map, bad beams) are its business, as they were k_ray_bits'.
// ---------------------------------------------------------------------------------
constexpr int kWedgeSlopes = 4;                       // slope classes per (steepness, direction): equal parts of [-1, 1] (8: 10 % slower)
constexpr int kWedgeClasses = 4 * kWedgeSlopes;       // x (steep?) x (walk runs away from / towards the origin)
constexpr int kWedgeLenBins = 64;
constexpr int kWedgeThreads = 512;
constexpr int kWedgeSlots = 4;                        // rays per lane and pass
constexpr int kWedgeCells = 37888;                    // 16-bit window cells (74 KiB): two workgroups per CU
constexpr int kWedgePartMin = kWedgeThreads * kWedgeSlots;   // rays per part of a unit (the scratch is sized for it)

struct WedgeScratch {
    uint32_t *ends;            // [rays] end cell, x | y << 16
    uint32_t *orgs;            // [scans] origin cell
    unsigned short *list;      // [rays] ray numbers inside their group, sorted by (class, length descending)
    int *offs;                 // [groups][kWedgeClasses + 1] class boundaries in the group's list
    uint32_t *cost;            // [groups][kWedgeClasses] cells the class's rays pass (from the length bins: to +-8 cells a ray)
    uint32_t *order;           // [groups * kWedgeClasses] units (group * kWedgeClasses + class), the most cells first
    const int32_t *got;        // nullable [L]: map of trajectory l (null: every trajectory casts into map 0)
};

__device__ __forceinline__ int wedge_class(const Ray &r, int ddx, int ddy)
{
    // minor over major displacement in [-1, 1], as the walk sees it (after the swaps the walk ascends along its axis)
    const int maj = r.steep ? ddy : ddx, mnr = r.steep ? ddx : ddy;
    const int q = min(kWedgeSlopes - 1, (int)(((long)(mnr * (maj < 0 ? -1 : 1) + abs(maj)) * kWedgeSlopes) / (2L * abs(maj))));
    return ((r.steep ? 2 : 0) + (r.flag ? 1 : 0)) * kWedgeSlopes + q;
}
__host__ __device__ inline int wedge_shear(int cls)      // nearest integer of the class's slopes
{
    const int q = cls % kWedgeSlopes;                     // slopes [-1 + 2 q / Q, -1 + 2 (q + 1) / Q): middle (2 q + 1 - Q) / Q
    const int mid2 = 2 * (2 * q + 1 - kWedgeSlopes);      // twice the middle, times Q
    return mid2 >= kWedgeSlopes ? 1 : (mid2 <= -kWedgeSlopes ? -1 : 0);
}

template <class Src>
__global__ void __launch_bounds__(1024) k_wedge_sort(GridDev g, Src src, WedgeScratch ws, int group_size)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ScanConst *sc = reinterpret_cast<ScanConst *>(smem);                                   // [group_size]
    int *hist = reinterpret_cast<int *>(smem + win_sc_bytes(group_size));                  // [kWedgeClasses * kWedgeLenBins]
    int *wsum = hist + kWedgeClasses * kWedgeLenBins;                                       // [64] cross-wave scan (16 used); before it: every scan's first bad beam
    unsigned short *keys = reinterpret_cast<unsigned short *>(wsum + kWinMaxGroup);         // [group_size * n]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = blockIdx.y, n = src.n, scans = src.scans_per_traj();
    const int groups_per_traj = (scans + group_size - 1) / group_size;
    const int s0 = blockIdx.x * group_size, cnt = min(group_size, scans - s0);
    const long group = (long)l * groups_per_traj + blockIdx.x;
    const long ray0 = ((long)l * scans + s0) * n;                     // first ray of the group in ends[]
    const size_t map_off = (size_t)(ws.got ? ws.got[l] : 0) * g.xw * g.yw;   // the trajectory's map (one shared map without `got`)
    uint32_t *pass = g.pass + map_off, *hit = g.hit + map_off;
    if (tid < cnt) {
        src.scan_const(l, s0 + tid, g, sc[tid]);
        ws.orgs[(long)l * scans + s0 + tid] = (uint32_t)(sc[tid].pcx & 0xffff) | ((uint32_t)(sc[tid].pcy & 0xffff) << 16);
    }
    for (int k = tid; k < kWedgeClasses * kWedgeLenBins; k += blockDim.x) hist[k] = 0;
    int *fb = wsum;                                                  // [cnt <= 64 ints, ahead of the scan that uses wsum] every scan's first bad beam
    if (tid < kWinMaxGroup) fb[tid] = INT_MAX;
    __syncthreads();
    const int nrays = cnt * n;
    // every scan stops at its first beam that Python's int() would raise on (mapping.py:29-36: the beams before it have been
    // applied when the exception leaves update(), and the error is that beam's): find it before any hit or count is added
    for (int r = tid; r < nrays; r += blockDim.x) {
        const int s = r / n, i = r - s * n;
        int pox, poy, b2 = 0;
        (void)src.ray(l, s0 + s, i, sc[s], g, pox, poy, b2);
        if (b2) atomicMin(&fb[s], i);
    }
    __syncthreads();
    unsigned nvis = 0;
    int bad = 0;
    for (int r = tid; r < nrays; r += blockDim.x) {
        const int s = r / n, i = r - s * n;
        int pox, poy, b2 = 0;
        Ray ry;
        const bool valid = i <= fb[s] && src.ray(l, s0 + s, i, sc[s], g, pox, poy, b2) && ray_setup(sc[s].pcx, sc[s].pcy, pox, poy, ry);
        const bool plain = valid && (unsigned)sc[s].pcx < (unsigned)g.xw && (unsigned)sc[s].pcy < (unsigned)g.yw &&
                           (unsigned)pox < (unsigned)g.xw && (unsigned)poy < (unsigned)g.yw;
        unsigned short key = 0xffffu;
        if (plain) {                                                 // every cell in the map: the path's last cell takes the hit (mapping.py:44-45)
            nvis += (unsigned)ry.dx + 1u;
            atomicAdd(&hit[(size_t)pox * g.yw + poy], 1u);
            ws.ends[ray0 + r] = (uint32_t)pox | ((uint32_t)poy << 16);
            const int cls = wedge_class(ry, pox - sc[s].pcx, poy - sc[s].pcy);
            key = (unsigned short)(cls * kWedgeLenBins + (kWedgeLenBins - 1 - min(ry.dx >> 4, kWedgeLenBins - 1)));   // longest first
            atomicAdd(&hist[key], 1);
        } else if (valid) {                                          // leaves the map: the step-by-step form with direct atomics
            const int klast = ry.flag ? 0 : ry.dx;
            double error = 0.0;
            int y = ry.y0, hx = -1, hy = -1;
            for (int k = 0; k <= ry.dx; ++k) {
                const int x = ry.x0 + k;
                const int lx = ry.steep ? y : x, ly = ry.steep ? x : y;
                const bool inmap = (unsigned)lx < (unsigned)g.xw && (unsigned)ly < (unsigned)g.yw;
                nvis += inmap ? 1u : 0u;
                if (k == klast) { hx = lx; hy = ly; }
                else if (inmap) atomicAdd(&pass[(size_t)lx * g.yw + ly], 1u);
                error += ry.derr;
                if (error >= 0.5) { y += ry.ystep; error -= 1.0; }
            }
            if ((unsigned)hx < (unsigned)g.xw && (unsigned)hy < (unsigned)g.yw) atomicAdd(&hit[(size_t)hx * g.yw + hy], 1u);
        }
        keys[r] = key;
        bad |= b2;
    }
    __syncthreads();
    // exclusive scan of the bins: kBinsPerThread consecutive bins per thread (1 024 threads), wave scans, then the waves' totals
    {
        constexpr int kBinsPerThread = kWedgeClasses * kWedgeLenBins / 1024;
        static_assert(kBinsPerThread >= 1 && kBinsPerThread * 1024 == kWedgeClasses * kWedgeLenBins && kWedgeLenBins % kBinsPerThread == 0, "bins per thread");
        int v[kBinsPerThread], sum = 0;
#pragma unroll
        for (int u = 0; u < kBinsPerThread; ++u) { v[u] = hist[tid * kBinsPerThread + u]; sum += v[u]; }
        {
            // cells a class's rays pass, from its length bins (16 cells wide): what k_wedge_order sorts the units by.  One bin
            // per thread and kWedgeLenBins = 64 bins per class: a class is a wave.
            static_assert(kBinsPerThread == 1 && kWedgeLenBins == kWave, "a class's bins are one wave");
            const unsigned c = wave_sum_u32((unsigned)v[0] * (unsigned)((kWedgeLenBins - 1 - lane) * 16 + 8));
            if (lane == 0) ws.cost[group * kWedgeClasses + wave] = c;
        }
        int inc = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int t = __shfl_up(inc, off, kWave); if (lane >= off) inc += t; }
        if (lane == kWave - 1) wsum[wave] = inc;
        __syncthreads();
        int run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int u = 0; u < kBinsPerThread; ++u) {
            const int bin = tid * kBinsPerThread + u;
            hist[bin] = run;
            if ((bin & (kWedgeLenBins - 1)) == 0) ws.offs[group * (kWedgeClasses + 1) + bin / kWedgeLenBins] = run;
            run += v[u];
        }
        if (tid == blockDim.x - 1) ws.offs[group * (kWedgeClasses + 1) + kWedgeClasses] = run;
    }
    __syncthreads();
    for (int r = tid; r < nrays; r += blockDim.x)
        if (keys[r] != 0xffffu) ws.list[ray0 + atomicAdd(&hist[keys[r]], 1)] = (unsigned short)r;
    unsigned tot = wave_sum_u32(nvis);
    int anybad = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) anybad |= __shfl_xor(anybad, off, kWave);
    if (lane == 0) {
        if (tot) atomicAdd(visit_slot(g.visits), (unsigned long long)tot);
        if (anybad) atomicOr(g.status, anybad);
    }
}

// The work of a launch in the order k_wedge_cast takes it: a (group, class) unit is cut into PARTS of at most `part_rays` rays
// of its length-sorted list, and the parts are sorted by decreasing work; workgroup b casts part order[1 + b] and the
// hardware starts workgroups in index order, so the longest parts start first.  Before (stamps, round 4: 1 008 units on 512
// workgroup slots) unit lifetimes ranged to 2.7 x the mean and the longest unit alone lasted 400 of the launch's 480 us.
// Counting sort by the logarithm of a part's share of its unit's cost (256 buckets, order inside a bucket arbitrary).
// order[0]: parts listed; an entry is unit | part << 20.
__global__ void __launch_bounds__(1024) k_wedge_order(WedgeScratch ws, int units, int part_rays)
{
    __shared__ int hist[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    auto parts_of = [&](int u) -> int {
        const int g = u / kWedgeClasses, c = u - g * kWedgeClasses;
        const int nr = ws.offs[(size_t)g * (kWedgeClasses + 1) + c + 1] - ws.offs[(size_t)g * (kWedgeClasses + 1) + c];
        return (nr + part_rays - 1) / part_rays;                      // (a class without rays: no part)
    };
    auto bucket = [](uint32_t c) -> int { return 255 - min(255, (int)(__log2f((float)c + 1.0f) * 9.0f)); };   // log2 < 32: 9 per octave
    for (int u = tid; u < units; u += blockDim.x) {
        const int np = parts_of(u);
        if (np > 0) atomicAdd(&hist[bucket(ws.cost[u] / (uint32_t)np)], np);
    }
    __syncthreads();
    if (tid < 64) {                                                  // exclusive scan of the 256 buckets by one wave
        int h4[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { h4[k] = hist[4 * tid + k]; tot += h4[k]; }
        int inc = tot;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) { const int v = __shfl_up(inc, off, kWave); if (tid >= off) inc += v; }
        int run = inc - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hist[4 * tid + k] = run; run += h4[k]; }
        if (tid == 63) ws.order[0] = (uint32_t)inc;
    }
    __syncthreads();
    for (int u = tid; u < units; u += blockDim.x) {
        const int np = parts_of(u);
        if (np <= 0) continue;
        const int base = atomicAdd(&hist[bucket(ws.cost[u] / (uint32_t)np)], np);
        for (int k = 0; k < np; ++k) ws.order[1 + base + k] = (uint32_t)u | ((uint32_t)k << 20);
    }
}

struct WedgeRay {
    int x0, y;             // walk origin along the axis; current y (the other axis, in walk coordinates)
    int k, kend;           // next walk step that passes a cell, last such step (the path's last cell - the hit - is neither)
    int ystep;
    double error, derr;
};

__global__ void __launch_bounds__(kWedgeThreads, 2 * kWedgeThreads / 256) k_wedge_cast(GridDev g, WedgeScratch ws, int n, int group_size, int scans, int part_rays)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *box = reinterpret_cast<int *>(smem);                          // [0] a_min [1] a_max [2] v_lo [3] v_hi [4] rows of the band
    unsigned *win = reinterpret_cast<unsigned *>(smem + 64);
    char *guard = reinterpret_cast<char *>(win) + (size_t)kWedgeCells * 2;
    lds_guard_fill(guard);
    STAMP_DECL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = kWedgeThreads >> 6;
    if (blockIdx.x >= ws.order[0]) return;                               // (the grid is sized for the largest possible number of parts)
    const uint32_t entry = ws.order[1 + blockIdx.x];                     // the parts with the most cells first
    const uint32_t unit = entry & 0xfffffu, part = entry >> 20;
    const int cls = (int)(unit % kWedgeClasses);
    const long group = unit / kWedgeClasses;
    const int groups_per_traj = (scans + group_size - 1) / group_size;
    const int l = (int)(group / groups_per_traj), s0 = (int)(group % groups_per_traj) * group_size;
    const long ray0 = ((long)l * scans + s0) * n;
    const int lo = ws.offs[group * (kWedgeClasses + 1) + cls] + (int)part * part_rays;
    const int hi = min(ws.offs[group * (kWedgeClasses + 1) + cls + 1], lo + part_rays);
    if (lo >= hi) return;
    const bool steep = (cls / kWedgeSlopes) >= 2;
    const int M = wedge_shear(cls);
    uint32_t *pass = g.pass + (size_t)(ws.got ? ws.got[l] : 0) * g.xw * g.yw;
    const unsigned wbase = lds_addr(win);

    for (int c0 = lo; c0 < hi; c0 += kWedgeThreads * kWedgeSlots) {     // (one pass unless a class holds more than 2 048 rays)
        WedgeRay wr[kWedgeSlots];
        int amin = INT_MAX, amax = INT_MIN;
#pragma unroll
        for (int j = 0; j < kWedgeSlots; ++j) {
            const int pos = c0 + j * kWedgeThreads + tid;                // slot 0 holds the class's longest rays
            WedgeRay &w = wr[j];
            w.x0 = 0; w.y = 0; w.k = 1; w.kend = 0; w.ystep = 1; w.error = 0.0; w.derr = 0.0;
            if (pos < hi) {
                const int r = (int)ws.list[ray0 + pos];
                const uint32_t e = ws.ends[ray0 + r], o = ws.orgs[(long)l * scans + s0 + r / n];
                Ray ry;
                (void)ray_setup((int)(o & 0xffffu), (int)(o >> 16), (int)(e & 0xffffu), (int)(e >> 16), ry);
                w.x0 = ry.x0; w.y = ry.y0; w.ystep = ry.ystep; w.derr = ry.derr;
                w.k = 0; w.kend = ry.dx - 1;                             // walk steps 0 .. dx - 1 pass, step dx is the hit ...
                if (ry.flag) {                                           // ... or, for a reversed path, step 0 is the hit: start one step in
                    w.error += w.derr;                                   // bresenham.py:51-55
                    if (w.error >= 0.5) { w.y += w.ystep; w.error -= 1.0; }
                    w.k = 1; w.kend = ry.dx;
                }
                if (w.k <= w.kend) { amin = min(amin, w.x0 + w.k); amax = max(amax, w.x0 + w.kend); }
            }
        }
        if (tid == 0) { box[0] = INT_MAX; box[1] = INT_MIN; }
        __syncthreads();
        amin = wave_min_i32(amin); amax = wave_max_i32(amax);
        if (lane == 0 && amin <= amax) { atomicMin(&box[0], amin); atomicMax(&box[1], amax); }
        __syncthreads();
        STAMP(0);                                   // ray set-up
        const int a_first = box[0] & ~1, a_end = box[1];               // (even band starts: pairs of counters line up for the flush)
        for (int a_lo = a_first; a_lo <= a_end;) {
            // rows of this band (even): try what is left (at most 512), shrink to what the parallelogram's width allows.
            // Along a ray the sheared coordinate v = y - M a moves one way only (each step changes it by 0 or by the
            // class's sign), so a ray's range in the band is spanned by its values on entry and on exit; the walk stays
            // within half a cell of its line (|error| <= 1/2), which bounds the exit value before the ray is walked.
            int rows = (min(a_end - a_lo + 1, 512) + 1) & ~1;
            int v_lo = 0, v_hi = -1, Hw = 0;
            for (int attempt = 0; attempt < 2; ++attempt) {
                if (tid == 0) { box[2] = INT_MAX; box[3] = INT_MIN; }
                __syncthreads();
                int vl = INT_MAX, vh = INT_MIN;
#pragma unroll
                for (int j = 0; j < kWedgeSlots; ++j) {
                    const WedgeRay &w = wr[j];
                    const int a_in = w.x0 + w.k, a_out = min(w.x0 + w.kend, a_lo + rows - 1);
                    if (w.k <= w.kend && a_in <= a_out) {
                        const double adv = (double)(a_out - a_in) * w.derr + w.error;     // y steps up to the exit, +- 1/2
                        const int yo_lo = w.y + (w.ystep > 0 ? (int)floor(adv - 0.5) : -(int)ceil(adv + 0.5));
                        const int yo_hi = w.y + (w.ystep > 0 ? (int)ceil(adv + 0.5) : -(int)floor(adv - 0.5));
                        const int vi = w.y - M * a_in, vo0 = yo_lo - M * a_out, vo1 = yo_hi - M * a_out;
                        vl = min(vl, min(vi, vo0)); vh = max(vh, max(vi, vo1));
                    }
                }
                vl = wave_min_i32(vl); vh = wave_max_i32(vh);
                if (lane == 0 && vl <= vh) { atomicMin(&box[2], vl); atomicMax(&box[3], vh); }
                __syncthreads();
                v_lo = box[2]; v_hi = box[3];
                if (v_lo > v_hi) break;                                  // no ray has a cell in these rows
                v_lo &= ~1;                                              // (even: pairs of counters line up for the flush)
                Hw = ((v_hi - v_lo + 1) + 1) & ~1;                       // parallelogram width, even
                const int fit = (kWedgeCells / Hw) & ~1;
                if (fit < 2) break;                                      // not even two rows fit: the band goes by direct atomics (below)
                if (rows <= fit) break;
                rows = fit;                                              // fewer rows: the extents can only shrink
                if (attempt == 1) break;
                __syncthreads();
            }
            if (v_lo > v_hi) { a_lo += rows; __syncthreads(); continue; }
            if ((long)Hw * 2 > kWedgeCells) {
                // rays of one class that lie too far apart for any window (scans of a group cast from origins all over a
                // large map): this band's cells go straight to the counters, one atomic each - slow, exact, rare
#pragma unroll
                for (int j = 0; j < kWedgeSlots; ++j) {
                    WedgeRay &w = wr[j];
                    while (w.k <= w.kend && w.x0 + w.k < a_lo + rows) {
                        const int a = w.x0 + w.k;
                        atomicAdd(&pass[steep ? (size_t)w.y * g.yw + a : (size_t)a * g.yw + w.y], 1u);   // mapping.py:43
                        w.error += w.derr;                               // bresenham.py:51-55
                        if (w.error >= 0.5) { w.y += w.ystep; w.error -= 1.0; }
                        ++w.k;
                    }
                }
                a_lo += rows;
                __syncthreads();
                continue;
            }
            // physical layout: rows of C halfwords.  Not steep: a row per walk-axis step (map x), columns along map y.
            // Steep: a row per sheared column (map x again), columns along the walk axis (map y) - contiguous in the map either way
            const int C = steep ? rows : Hw, P = steep ? Hw : rows;
            const int ca = steep ? 1 : C, cv = steep ? C : 1;
            for (int w4 = tid; w4 < (P * C) >> 3; w4 += kWedgeThreads) reinterpret_cast<uint4 *>(win)[w4] = make_uint4(0u, 0u, 0u, 0u);
            for (int w1 = ((P * C) >> 3) * 4 + tid; w1 < (P * C) >> 1; w1 += kWedgeThreads) win[w1] = 0u;
            __syncthreads();
            STAMP(1);                               // band geometry + zero
            const int dh_a = ca - M * cv;
#pragma unroll
            for (int j = 0; j < kWedgeSlots; ++j) {
                WedgeRay &w = wr[j];
                const int a_in = w.x0 + w.k;
                int rem = (w.k <= w.kend && a_in < a_lo + rows) ? min(w.x0 + w.kend, a_lo + rows - 1) - a_in + 1 : 0;
                if (!__any(rem > 0)) continue;
                unsigned a2 = wbase + 2u * (unsigned)((a_in - a_lo) * ca + (w.y - M * a_in - v_lo) * cv);
                const int da_k = 2 * dh_a, da_y = 2 * w.ystep * cv;
                double error = w.error;
                int y = w.y;
                const int took = rem;
                auto step = [&]() {
                    lds_add_u32(a2 & ~3u, 1u << ((a2 << 3) & 31u));      // mapping.py:43
                    error += w.derr;                                     // bresenham.py:51
                    const bool stepy = error >= 0.5;                     // :53
                    a2 += (unsigned)(da_k + (stepy ? da_y : 0));
                    y += stepy ? w.ystep : 0;
                    error -= __hiloint2double(stepy ? 0x3ff00000 : 0, 0);   // :55 (minus 1.0, or minus 0.0: exact)
                };
                for (;;) {
                    const bool full = rem >= 4;
                    if (!__any(full)) break;
                    if (full) { step(); step(); step(); step(); rem -= 4; }
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (rem > u) step();
                w.error = error; w.y = y; w.k += took;
            }
            __syncthreads();
            STAMP(3);                               // walk
            // flush: a wave per MAP row (map x), lanes along map y, two cells per lane as one 64-bit add where the pair is
            // aligned.  Not steep, or steep without shear: a map row is a physical row of the window.  Steep with shear: the
            // cells of map row x lie on a diagonal of the window (column c = map y - a_lo in physical row x - M y - v_lo),
            // read with per-lane addresses.
            const bool diag = steep && M != 0;
            const int nrows = diag ? P + C - 1 : P;                      // map rows the parallelogram touches
            const int x_first = !steep ? a_lo : (M == 0 ? v_lo : v_lo + (M > 0 ? M * a_lo : M * (a_lo + C - 1)));
            const unsigned short *winh = reinterpret_cast<const unsigned short *>(win);
            const int rot = (int)((blockIdx.x * 37u + blockIdx.y * 11u) % (unsigned)nrows);
            for (int pr = wave; pr < nrows; pr += nwaves) {
                const int p = pr + rot < nrows ? pr + rot : pr + rot - nrows;
                const int mx = x_first + p;
                if ((unsigned)mx >= (unsigned)g.xw) continue;            // (rows beyond the rays' reach hold nothing)
                const int my0 = steep ? a_lo : v_lo + M * mx;            // map y of column 0
                const size_t gbase = (size_t)mx * g.yw + my0;
                const bool pair = ((gbase & 1) == 0);
                for (int d = lane; d < (C >> 1); d += kWave) {
                    unsigned v;
                    if (!diag) v = win[p * (C >> 1) + d];
                    else {
                        const int p0 = mx - v_lo - M * (a_lo + 2 * d), p1 = p0 - M;     // physical rows of the pair's two cells
                        const unsigned c0 = (unsigned)p0 < (unsigned)P ? winh[p0 * C + 2 * d] : 0u;
                        const unsigned c1 = (unsigned)p1 < (unsigned)P ? winh[p1 * C + 2 * d + 1] : 0u;
                        v = c0 | (c1 << 16);
                    }
                    if (!v) continue;
                    if (pair) atomicAdd(reinterpret_cast<unsigned long long *>(&pass[gbase + 2 * d]), (unsigned long long)(v & 0xffffu) | ((unsigned long long)(v >> 16) << 32));
                    else {
                        if (v & 0xffffu) atomicAdd(&pass[gbase + 2 * d], v & 0xffffu);
                        if (v >> 16) atomicAdd(&pass[gbase + 2 * d + 1], v >> 16);
                    }
                }
            }
            a_lo += rows;
            __syncthreads();
            STAMP(4);                               // flush
        }
    }
    STAMP_END(5);
    lds_guard_check(guard, g.status);
}

size_t wedge_scratch_bytes(long rays, long scans, long groups)
{
    return (size_t)rays * 6 + (size_t)scans * 4 + (size_t)groups * (kWedgeClasses + 1) * 4 + (size_t)groups * kWedgeClasses * 8 + (size_t)(rays / kWedgePartMin + 1) * 4 + 4096;
}

template <class Src>
static hipError_t launch_wedges(const GridDev &g, const Src &src, int L, int scans, int n, int group, void *scratch, hipStream_t s, const int32_t *got)
{
    if (scans < 1) return hipSuccess;
    int G = group > 0 ? group : 16;
    G = std::min(G, std::max(1, 65535 / n));
    G = std::min(G, scans);
    if ((long)G * n > 65535) return hipErrorInvalidValue;   // (ray numbers inside a group are 16-bit: keys[], ws.list[])
    const int groups_per_traj = (scans + G - 1) / G;
    const long groups = (long)L * groups_per_traj, rays = (long)L * scans * n;
    if (groups > 65535 || g.xw > 65535 || g.yw > 65535) return hipErrorInvalidValue;   // (end cells travel as 16-bit pairs)
    WedgeScratch ws;
    char *p = static_cast<char *>(scratch);
    ws.ends = reinterpret_cast<uint32_t *>(p); p += (size_t)rays * 4;
    ws.orgs = reinterpret_cast<uint32_t *>(p); p += (size_t)L * scans * 4;
    ws.offs = reinterpret_cast<int *>(p); p += (size_t)groups * (kWedgeClasses + 1) * 4;
    ws.cost = reinterpret_cast<uint32_t *>(p); p += (size_t)groups * kWedgeClasses * 4;
    ws.order = reinterpret_cast<uint32_t *>(p); p += ((size_t)groups * kWedgeClasses + (size_t)rays / kWedgePartMin + 1) * 4;
    ws.list = reinterpret_cast<unsigned short *>(p);
    ws.got = got;
    if (g.pmap_live && g.live_dirty) *g.live_dirty = true;
    const size_t lds_a = win_sc_bytes(G) + (size_t)(kWedgeClasses * kWedgeLenBins + kWinMaxGroup) * 4 + (size_t)G * n * 2;
    const size_t lds_b = 64 + (size_t)kWedgeCells * 2 + kLdsGuard;
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(&k_wedge_sort<Src>), (int)lds_a);
    if (e == hipSuccess) e = allow_dynamic_lds(reinterpret_cast<const void *>(&k_wedge_cast), (int)lds_b);
    if (e != hipSuccess) return e;
    SLAM_LAUNCH((k_wedge_sort<Src>), dim3(groups_per_traj, L), dim3(1024), lds_a, s, g, src, ws, G);
    // rays per part: what a workgroup's lanes hold in one pass.  (Smaller parts balance the launch better and cost more than they
    // gain - every part of a class zeroes and flushes the class's bands again: 1 024 rays 0.533 ms, 512 rays 0.667 ms against 0.516.)
    const int part_rays = kWedgeThreads * kWedgeSlots;
    const long max_parts = groups * kWedgeClasses + rays / part_rays;      // (every unit's last part may be short)
    SLAM_LAUNCH(k_wedge_order, dim3(1), dim3(1024), 0, s, ws, (int)(groups * kWedgeClasses), part_rays);
    SLAM_LAUNCH(k_wedge_cast, dim3((unsigned)max_parts), dim3(kWedgeThreads), lds_b, s, g, ws, n, G, scans, part_rays);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_grid_finalize(const uint32_t *__restrict__ pass, const uint32_t *__restrict__ hit,
                                                       size_t cells, OccRule rule, int8_t *__restrict__ pmap)
{
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t c = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; c < cells; c += stride) {
        if (c + 4 <= cells) {
            uint4 p = *reinterpret_cast<const uint4 *>(pass + c);
            uint4 h = *reinterpret_cast<const uint4 *>(hit + c);
            uint32_t out = rule.value(p.x, h.x) | rule.value(p.y, h.y) << 8 | rule.value(p.z, h.z) << 16 |
                           rule.value(p.w, h.w) << 24;
            *reinterpret_cast<uint32_t *>(pmap + c) = out;
        } else {
            for (size_t e = c; e < cells; ++e) pmap[e] = (int8_t)rule.value(pass[e], hit[e]);
        }
    }
}

hipError_t launch_grid_finalize(const GridDev &g, int g0, int gcount, int8_t *pmap, hipStream_t s)
{
    size_t per = (size_t)g.xw * g.yw, cells = per * gcount;
    size_t blocks = (cells / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    OccRule rule = OccRule::of(g);
    SLAM_LAUNCH(k_grid_finalize, dim3(blocks), dim3(256), 0, s, g.pass + per * g0, g.hit + per * g0, cells, rule, pmap);
    return hipGetLastError();
}

// datamap view of the counters: free_inc*pass + hit_inc*hit (mapping.py:43,45).
__global__ void __launch_bounds__(256) k_grid_datamap(const uint32_t *__restrict__ pass, const uint32_t *__restrict__ hit,
                                                      size_t cells, double free_inc, double hit_inc, double *__restrict__ out)
{
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x)
        out[c] = free_inc * (double)pass[c] + hit_inc * (double)hit[c];
}

hipError_t launch_grid_datamap(const GridDev &g, int gi, double *datamap, hipStream_t s)
{
    size_t per = (size_t)g.xw * g.yw;
    size_t blocks = (per + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    SLAM_LAUNCH(k_grid_datamap, dim3(blocks), dim3(256), 0, s, g.pass + per * gi, g.hit + per * gi, per, g.free_inc,
                       g.hit_inc, datamap);
    return hipGetLastError();
}

// publishMap layout (slam_ekf.py:270-271): data[y*xw + x] = pmap[x][y]; 32x32 LDS tile
// transpose so both sides are coalesced.
__global__ void __launch_bounds__(256) k_grid_transpose(const int8_t *__restrict__ pmap, int xw, int yw, int8_t *__restrict__ data)
{
    __shared__ int8_t tile[32][33];
    int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        int x = x0 + r, y = y0 + tx;
        if (x < xw && y < yw) tile[r][tx] = pmap[(size_t)x * yw + y];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int y = y0 + r, x = x0 + tx;
        if (x < xw && y < yw) data[(size_t)y * xw + x] = tile[tx][r];
    }
}

hipError_t launch_grid_transpose(const int8_t *pmap, int xw, int yw, int8_t *data, hipStream_t s)
{
    SLAM_LAUNCH(k_grid_transpose, dim3((xw + 31) / 32, (yw + 31) / 32), dim3(256), 0, s, pmap, xw, yw, data);
    return hipGetLastError();
}

// bresenham(start, end).path (bresenham.py:2-58): one lane per line.
__global__ void __launch_bounds__(256) k_bresenham(const int32_t *__restrict__ starts, const int32_t *__restrict__ ends, int B,
                                                   const int64_t *__restrict__ offsets, int32_t *__restrict__ lens,
                                                   int32_t *__restrict__ cells)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    Ray r;
    if (!ray_setup(starts[2 * b], starts[2 * b + 1], ends[2 * b], ends[2 * b + 1], r)) { lens[b] = 0; return; }
    lens[b] = r.dx + 1;
    if (!cells) return;
    int32_t *out = cells + 2 * offsets[b];
    double error = 0.0;
    int y = r.y0;
    for (int k = 0; k <= r.dx; ++k) {
        int x = r.x0 + k;
        int j = r.flag ? r.dx - k : k;                               // path.reverse(), :57-58
        out[2 * (size_t)j] = r.steep ? y : x;
        out[2 * (size_t)j + 1] = r.steep ? x : y;
        error += r.derr;
        if (error >= 0.5) { y += r.ystep; error -= 1.0; }
    }
}

hipError_t launch_bresenham(const int32_t *starts, const int32_t *ends, int B, const int64_t *offsets, int32_t *lens,
                            int32_t *cells, hipStream_t s)
{
    SLAM_LAUNCH(k_bresenham, dim3((B + 255) / 256), dim3(256), 0, s, starts, ends, B, offsets, lens, cells);
    return hipGetLastError();
}

}  // namespace slam
