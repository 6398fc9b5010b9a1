// Scan-to-map observation kernels for gfx950: the map's obstacle cells are
// projected into the LaserScan the robot would see from a pose hypothesis, which is then
// matched against the real scan by the ICP kernel.
//
// Functional spec = W9 = "W9_Fusion Localization (LiDAR Odometry)/course_agv_slam/scripts"
// (under /root/reference):
//   Localization.updateMap        localization.py:54-60    -> k_map_obstacles
//   Localization.laserEstimation  localization.py:128-150  -> k_virtual_scan
//   Localization.laserToNumpy     localization.py:168-174  -> k_ranges64_to_points
//   Localization.calc_map_observation :152-157            -> slam_map_observation (ABI layer)
//
// The projection is a scatter-min: every obstacle drops its distance into one beam bin and
// the bin keeps the smallest.  Distances are non-negative float64, whose bit patterns order
// like the values, so the minimum is an integer atomicMin on the bits: order-free and
// reproducible.  One lane per (obstacle, pose hypothesis), bins privatised in LDS.
#include <hip/hip_runtime.h>

#include "slam_internal.h"

namespace slam {

// updateMap (localization.py:54-60): cells > 20 or < -0.5 are obstacles (occupied AND
// unknown: pmap's 50 counts).  `wire` selects the OccupancyGrid layout data[y*width + x]
// (what the reference receives); otherwise the map is [x][y] as Mapping.pmap.  The list
// order is arbitrary (atomic append); the consumers are order-free.
__global__ void __launch_bounds__(256) k_map_obstacles(const int8_t *__restrict__ map, int width, int height, int wire,
                                                       double resolution, double origin_x, double origin_y,
                                                       double *__restrict__ ox, double *__restrict__ oy, int cap,
                                                       int *__restrict__ count)
{
    const long cells = (long)width * height;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (long)gridDim.x * blockDim.x) {
        int x, y;
        if (wire) { y = (int)(c / width); x = (int)(c - (long)y * width); }
        else      { x = (int)(c / height); y = (int)(c - (long)x * height); }
        int v = map[c];
        if (v > 20 || v < 0) {
            int k = atomicAdd(count, 1);
            if (k < cap) {
                ox[k] = (x * resolution + origin_x) * 1.0;           // :57
                oy[k] = (y * resolution + origin_y) * 1.0;           // :58
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_fill_u64(unsigned long long *p, long n, unsigned long long v)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

// laserEstimation (localization.py:128-150).  blockIdx.y = pose hypothesis, blockIdx.x = slice
// of the obstacle list.  The beam bins of the hypothesis live in LDS (n x 8 B) and take the
// scatter-min there (ds_min_u64); a workgroup that owns the whole list stores its bins,
// otherwise the slices meet in global memory with one atomicMin per touched bin.
__global__ void __launch_bounds__(256) k_virtual_scan(const double *__restrict__ ox, const double *__restrict__ oy, int K,
                                                      const double *__restrict__ poses, double angle_min,
                                                      double angle_increment, int n, unsigned long long empty,
                                                      unsigned long long *__restrict__ ranges)
{
    extern __shared__ unsigned long long bins[];
    const int b = blockIdx.y;
    const double px = poses[3 * b], py = poses[3 * b + 1], pth = poses[3 * b + 2];
    unsigned long long *r = ranges + (long)b * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) bins[i] = empty;
    __syncthreads();
    const int per = (K + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(K, lo + per);
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        double x = ox[i], y = oy[i];
        double dist = hypot(px - x, py - y);                                          // :138
        double q = (atan2(y - py, x - px) - angle_min - pth) / angle_increment;       // :139
        if (!(fabs(q) < 2.0e9)) continue;                // NaN / absurd: the reference would raise or spin
        long index = (long)q;                            // int(): truncation toward zero
        index %= n;                                      // the two while-loops of :141-144
        if (index < 0) index += n;
        atomicMin(&bins[index], (unsigned long long)__double_as_longlong(dist));      // :145-146 (strict '<' = min)
    }
    __syncthreads();
    if (gridDim.x == 1) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) r[i] = bins[i];
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            if (bins[i] < empty) atomicMin(&r[i], bins[i]);
    }
}

// laserToNumpy on float64 ranges (localization.py:168-174): [B][n] -> points [B][2][n].
__global__ void __launch_bounds__(256) k_ranges64_to_points(const double *__restrict__ ranges, const double *__restrict__ cos_t,
                                                            const double *__restrict__ sin_t, long total, int n,
                                                            double *__restrict__ pts)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long b = e / n;
        int i = (int)(e - b * n);
        double r = ranges[e];
        pts[b * 2 * n + i] = cos_t[i] * r;
        pts[b * 2 * n + n + i] = sin_t[i] * r;
    }
}

hipError_t launch_map_obstacles(const int8_t *map, int width, int height, int wire, double resolution, double origin_x,
                                double origin_y, double *ox, double *oy, int cap, int *count, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    long cells = (long)width * height;
    long blocks = (cells + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SLAM_LAUNCH(k_map_obstacles, dim3(blocks), dim3(256), 0, s, map, width, height, wire, resolution, origin_x,
                       origin_y, ox, oy, cap, count);
    return hipGetLastError();
}

hipError_t launch_virtual_scan(const double *ox, const double *oy, int K, const double *poses, int B, double angle_min,
                               double angle_increment, int n, double *ranges, hipStream_t s)
{
    const double hundred = 100.0;                                    // data.ranges = [100.0]*total_num (:135)
    unsigned long long bits;
    memcpy(&bits, &hundred, sizeof bits);
    // enough workgroups to fill 256 CUs a few times over; a slice is at least 1024 obstacles
    int slices = (2048 + B - 1) / B;
    int max_slices = (K + 1023) / 1024;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    long total = (long)B * n;
    if (slices > 1)
        SLAM_LAUNCH(k_fill_u64, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0, s,
                           reinterpret_cast<unsigned long long *>(ranges), total, bits);
    SLAM_LAUNCH(k_virtual_scan, dim3(slices, B), dim3(256), (size_t)n * 8, s, ox, oy, K, poses, angle_min,
                       angle_increment, n, bits, reinterpret_cast<unsigned long long *>(ranges));
    return hipGetLastError();
}

hipError_t launch_ranges64_to_points(const double *ranges, const double *cos_t, const double *sin_t, int B, int n,
                                     double *pts, hipStream_t s)
{
    long total = (long)B * n;
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    SLAM_LAUNCH(k_ranges64_to_points, dim3(blocks), dim3(256), 0, s, ranges, cos_t, sin_t, total, n, pts);
    return hipGetLastError();
}

}  // namespace slam
