// Internal declarations shared by the kernel files and the C-ABI layer of libslamhip.so.
// gfx950 (MI355X) only; wavefront = 64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "slam_hip.h"

#include <hip/hip_ext.h>

namespace slam {

// Kernel timing: when the ABI layer has opened a timing bracket (slam_timing_enable; `Timed` in
// slam_abi.hip), every launch inside it asks the bracket for a fresh pair of events and carries
// them as the dispatch's own start / stop events (hipExtLaunchKernelGGL), so they bracket exactly
// the kernel's execution - what rocprofv3's kernel trace reports - and no marker packets are put
// into the queue.  A family that is several launches (the byte-window ray cast and its fallback,
// the wedge sort and cast) is the sum of its launches.
struct LaunchTimer {
    void *self;
    bool (*next)(void *self, hipEvent_t *e0, hipEvent_t *e1);
};
extern thread_local LaunchTimer g_launch_timer;
inline bool launch_events(hipEvent_t *e0, hipEvent_t *e1)
{
    return g_launch_timer.self && g_launch_timer.next(g_launch_timer.self, e0, e1);
}

#define SLAM_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                          \
    do {                                                                                                              \
        hipEvent_t e0_ = nullptr, e1_ = nullptr;                                                                      \
        if (::slam::launch_events(&e0_, &e1_))                                                                        \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, e0_, e1_, 0, __VA_ARGS__);                      \
        else                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                      \
    } while (0)


// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: the largest size set so far is remembered
// per (device, function), so a second device used by the same process gets the attribute too and a later launch that
// needs more LDS than any before it raises it.
hipError_t allow_dynamic_lds(const void *kernel, int bytes);

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;            // 1024-thread workgroups
constexpr int kMaxRayCells = 1 << 20;    // longest ray the grid kernels will walk (cells)

enum : int { kStatusNaN = 1, kStatusOverflow = 2, kStatusGuard = 4 };

// Debug build (-DSLAM_LDS_GUARD, `make guard`): every kernel with dynamic LDS gets kLdsGuard
// extra bytes behind its last region, fills them with a pattern and checks them before it
// exits; a changed word raises kStatusGuard (SLAM_ERR_HIP from slam_check_status).  GPU
// AddressSanitizer is not available on the target pool; this catches the overrun class that
// matters here (a region sized too small, e.g. a window row count rounded the wrong way).
#ifdef SLAM_LDS_GUARD
constexpr int kLdsGuard = 512;
#else
constexpr int kLdsGuard = 0;
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void lds_guard_fill(char *p)
{
    for (int i = threadIdx.x; i < kLdsGuard / 4; i += blockDim.x) reinterpret_cast<unsigned *>(p)[i] = 0xA5A5A5A5u;
}
__device__ __forceinline__ void lds_guard_check(const char *p, int *status)
{
    if (kLdsGuard == 0) return;
    __syncthreads();
    bool bad = false;
    for (int i = threadIdx.x; i < kLdsGuard / 4; i += blockDim.x) bad |= reinterpret_cast<const unsigned *>(p)[i] != 0xA5A5A5A5u;
    if (bad && status) atomicOr(status, kStatusGuard);
}
#endif

// ---- ICP ---------------------------------------------------------------------------
struct IcpArgs {
    // Either point buffers ...
    const void *tar, *src;     // [2][n] per set, storage type = template T
    // ... or raw scans (fused polar->Cartesian, used by the replay / particle pipelines): scan k
    // of a stream is the source, scan k-1 the target; the points are formed in registers with the
    // same arithmetic as k_scan_to_points (one multiply, then rounding to T) and never touch HBM.
    const float *ranges;       // nullable; [.. scans ..][n]
    const double *cos_t, *sin_t;
    long tar_scan_stride, src_scan_stride;   // floats between consecutive pairs' scans (0: shared)
    const double *prior;       // nullable [B][6]
    long tar_stride, src_stride;  // elements between consecutive sets (0: shared)
    int ppt;                   // replay addressing: pairs per trajectory (0: plain batch)
    int B, n_tar, n_src, max_iter;
    double tol;
    double *T_out;             // [B][9]
    int32_t *iters_out;        // nullable [B]
    double *err_out;           // nullable [B]
    int *status = nullptr;     // sticky status word of the context (LDS guard builds)
    int qpt_pref = 0;          // queries per lane in batched launches: 0 = by batch size (context option "icp_qpt")
    void *zero_ptr = nullptr;  // nullable: zero_bytes bytes (a multiple of 4, 16-byte aligned start) that the launch clears on its way -
    size_t zero_bytes = 0;     //   the counters of the map the replay's ray cast fills next (slam_replay_dev, option "replay_reset")
    int polar_copy = 0;        // set by launch_icp: the kernel carves the unpadded second copy of the target (nn_polar)
    int team_cap = 0;          // set by launch_icp: room in the LDS list of first-iteration queries without a beam window (nn_listed), 0: none
    int team_mode = 0;         // context option "icp_team": 0 = on where it applies, 1 = off (the box search takes every such query)
};

hipError_t launch_icp(const IcpArgs &a, int dtype, hipStream_t s);
hipError_t launch_nn(const void *src, const void *tar, int B, int n_src, int n_tar, int dtype, double *dist,
                     int32_t *idx, hipStream_t s);
hipError_t launch_kabsch(const double *src, const double *tar, int B, int n, double *T_out, hipStream_t s);
hipError_t launch_scan_to_points(const float *ranges, const double *cos_t, const double *sin_t, long total,
                                 int n, int clip_inf, int dtype, void *pts, hipStream_t s);
// prior: nullable [L][6] (n must be 1): the step composed is T.[prior; 0 0 1] (particle hypotheses).
// heading_cs: nullable [L][2] (n must be 1): cos / sin of the new headings, for the ray cast that follows.
hipError_t launch_pose_compose(const double *T, const double *pose0, int L, int n, double *poses, hipStream_t s,
                               const double *prior = nullptr, double *heading_cs = nullptr);

#ifdef SLAM_STAMPS_ICP
hipError_t debug_polar_lanes(unsigned long long out[4], bool clear);   // diagnostic build: lane efficiency counters of nn_polar (slam_stamps.h)
#endif

// ---- grid --------------------------------------------------------------------------
constexpr int kMaxHitLevels = 8;
constexpr int kVisitSlots = 256;         // power of two
constexpr int kVisitStride = 8;          // unsigned long longs between slots (64 bytes)
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned long long *visit_slot(unsigned long long *visits)
{
    return visits + (size_t)((blockIdx.x + blockIdx.y * 131u) & (unsigned)(kVisitSlots - 1)) * kVisitStride;
}
#endif

struct GridDev {
    int G, xw, yw;
    double scale, off_x, off_y;
    uint32_t *pass, *hit;          // [G][xw][yw]
    // In-bounds cell visits since reset: kVisitSlots counters, one per 64-byte line; a workgroup adds
    // to the slot of its block index and slam_grid_visits sums them.  (One shared counter made every
    // wave of a 10 000-workgroup launch queue on one address: 80 000 serialised atomics, 1.6 ms.)
    unsigned long long *visits;
    int *status;                   // sticky kStatus* bits (context-wide)
    // Occupied rule on the integer counters (see slam_grid_create): a cell with h hits and p
    // passes is occupied iff h >= hit_levels or p >= pass_thresh[h].
    int hit_levels;                // 1 for the reference's +20 (one hit occupies), 3 for the online variant's +4
    uint32_t pass_thresh[kMaxHitLevels];
    // Live pmap (slam_grid_live_pmap): [G][xw][yw] int8 kept current by ray casts that own their
    // map exclusively (one workgroup per map); any other update sets *live_dirty (host flag) and
    // the next finalize / read refreshes it with a full pass.
    int8_t *pmap_live;
    bool *live_dirty;
    // Re-do list of the single-scan owner kernels (allocated with the live pmap): [0] maps listed, [1] workgroups
    // of the general kernel that have read the list, [2 + k] map numbers.  Empty between launches.
    int32_t *redo;
    double free_inc, hit_inc;
};

hipError_t launch_grid_update(const GridDev &g, const double *ox, const double *oy, const double *cx,
                              const double *cy, int B, int n, const int32_t *grid_of_batch, hipStream_t s);
// world points are formed in-kernel from ranges and poses (slam_ekf.py:89): scan k>=1 of
// trajectory l uses poses[l][k-1].
hipError_t launch_grid_update_replay(const GridDev &g, const float *ranges, const double *cos_t,
                                     const double *sin_t, const double *poses, int L, int n_scan, int n,
                                     const int32_t *grid_of_traj, hipStream_t s);
// LDS-window variants (group = scans per workgroup, 0 = automatic); they fall back to the
// direct-atomic kernels when a scan has too many beams for the packed window counters.
// split_pref: two workgroups per group of scans, one per direction half: -1 = when the launch cannot fill the chip, 0 = never, 1 = always
hipError_t launch_grid_update_win(const GridDev &g, const double *ox, const double *oy, const double *cx,
                                  const double *cy, int B, int n, int group, hipStream_t s, int split_pref = -1);
hipError_t launch_grid_update_replay_win(const GridDev &g, const float *ranges, const double *cos_t,
                                         const double *sin_t, const double *poses, int L, int n_scan, int n,
                                         const int32_t *grid_of_traj, int group, hipStream_t s, int shared_scans = 0,
                                         int grid_per_traj = 0, const double *heading_cs = nullptr, int split_pref = -1);
hipError_t launch_grid_update_scans(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                    const double *poses, const double *centres, int S, int n, int group, hipStream_t s, int split_pref = -1);
// Tiled path for maps much larger than an LDS window (single shared map, ReplaySource only).
constexpr int kTileMaxBeams = 8192;      // beams per scan the tiled / wedge paths take (ray numbers inside a group are 16-bit)
size_t tile_scratch_bytes(long rays, long groups);
size_t wedge_scratch_bytes(long rays, long scans, long groups);
bool tiles_apply(const GridDev &g, int n, const int32_t *got, int grid_per_traj, int wedges = 0);
// got: nullable [L] map of every trajectory (wedges only; the recorded-walk tiles cast into one shared map)
hipError_t launch_grid_update_tiles(const GridDev &g, const float *ranges, const double *cos_t, const double *sin_t,
                                    const double *poses, const double *centres, int L, int n_scan, int n, int group,
                                    void *scratch, hipStream_t s, int wedges = 0, const int32_t *got = nullptr);
hipError_t launch_grid_update_tiles_explicit(const GridDev &g, const double *ox, const double *oy, const double *cx,
                                             const double *cy, int B, int n, int group, void *scratch, hipStream_t s, int wedges = 0);
hipError_t launch_grid_finalize(const GridDev &g, int g0, int gcount, int8_t *pmap, hipStream_t s);
hipError_t launch_grid_datamap(const GridDev &g, int gi, double *datamap, hipStream_t s);
hipError_t launch_grid_transpose(const int8_t *pmap, int xw, int yw, int8_t *data, hipStream_t s);
hipError_t launch_bresenham(const int32_t *starts, const int32_t *ends, int B, const int64_t *offsets,
                            int32_t *lens, int32_t *cells, hipStream_t s);

// ---- scan-to-map observation ---------------------------------------
hipError_t launch_map_obstacles(const int8_t *map, int width, int height, int wire, double resolution, double origin_x,
                                double origin_y, double *ox, double *oy, int cap, int *count, hipStream_t s);
hipError_t launch_virtual_scan(const double *ox, const double *oy, int K, const double *poses, int B, double angle_min,
                               double angle_increment, int n, double *ranges, hipStream_t s);
hipError_t launch_ranges64_to_points(const double *ranges, const double *cos_t, const double *sin_t, int B, int n,
                                     double *pts, hipStream_t s);

}  // namespace slam
