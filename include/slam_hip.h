/*
 * slam_hip.h - C ABI of libslamhip.so: the MI355X (gfx950) implementation of the
 * per-scan SLAM hot path of zjwzcx/A-2D-LiDAR-based-SLAM-System-for-Wheeled-Mobile-Robots.
 *
 * The reference has no FFI layer (it is pure Python 2 / NumPy); its boundary for this
 * path is the Python class API of course_agv_slam/scripts.  Each entry
 * point below names the reference interface it replaces.  Reference paths:
 *   W12m = "W12_LiDAR SLAM/w12-mapping/course_agv_slam/scripts"
 *   W7   = "W7_Dead Reckoning (ICP)/course_agv_slam/scripts"
 * The ctypes binding a maintainer of the reference would add is shown in INTEGRATION.md
 * and implemented in the package's _abi.py.
 *
 * Conventions
 *  - plain C types only; every function returns 0 (SLAM_OK) or a negative SLAM_ERR_*;
 *    slam_last_error() returns the message of the calling thread's last failure.
 *  - a slam_ctx owns one HIP stream (or borrows the caller's) and a device workspace;
 *    use one context per host thread.  Nothing here falls back to the CPU: without a
 *    usable gfx950 device slam_create fails.
 *  - functions without suffix take HOST pointers, copy in, run the kernels, copy out and
 *    synchronise; *_dev functions take DEVICE pointers (e.g. torch.Tensor.data_ptr()),
 *    only enqueue work on the context's stream and do not synchronise.
 *  - point sets are structure-of-arrays: one set is 2*n values, the n x coordinates
 *    then the n y coordinates ("[2][n]").  The reference's third row of ones
 *    (icp.py:42-49) is not stored.  `dtype` is the STORAGE type of a point buffer
 *    (SLAM_F64 / SLAM_F32 / SLAM_F16); all arithmetic is float64 regardless.
 *  - T is a 3x3 row-major float64 matrix [[R, t], [0, 0, 1]] exactly as ICP.process
 *    returns it; poses are (x, y, theta) float64.
 *  - grids are [xw][yw] row-major (x outer) as mapping.py:14-15.
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAM_ABI_VERSION 1

typedef struct slam_ctx slam_ctx;
typedef struct slam_grid slam_grid;

enum {
    SLAM_OK = 0,
    SLAM_ERR_INVALID = -1,  /* bad argument (null pointer, non-positive size, unknown dtype) */
    SLAM_ERR_HIP = -2,      /* a HIP runtime call or kernel launch failed                    */
    SLAM_ERR_NOMEM = -3,    /* device or host allocation failed                              */
    SLAM_ERR_NAN = -4,      /* NaN world coordinate: the reference raises ValueError from
                               int(nan) at mapping.py:33                                     */
    SLAM_ERR_OVERFLOW = -5, /* infinite / unrepresentable cell index: the reference raises
                               OverflowError from int(inf) at mapping.py:33-36 (or would walk
                               a ray of more than 2^20 cells)                                */
    SLAM_ERR_NODEVICE = -6  /* no gfx950 device visible                                      */
};

enum { SLAM_F64 = 0, SLAM_F32 = 1, SLAM_F16 = 2 };

/* kernel families timed by slam_timing_* */
enum { SLAM_K_POINTS = 0, SLAM_K_ICP = 1, SLAM_K_COMPOSE = 2, SLAM_K_GRID = 3, SLAM_K_FINALIZE = 4,
       SLAM_K_NN = 5, SLAM_K_KABSCH = 6, SLAM_K_BRESENHAM = 7, SLAM_K_COUNT = 8 };

int slam_abi_version(void);
const char *slam_last_error(void);

/* ---- context ------------------------------------------------------------------ */
/* device: HIP ordinal.  stream: a hipStream_t to enqueue on (e.g. torch's current
 * stream), or NULL to create a private one. */
int slam_create(int device, void *stream, slam_ctx **out);
int slam_destroy(slam_ctx *ctx);
int slam_synchronize(slam_ctx *ctx);
/* Order the context's work against another HIP stream of the caller WITHOUT a host synchronise (stream: a
 * hipStream_t, e.g. torch's current stream, on which the RCCL collectives of torch.distributed are ordered; NULL: the
 * legacy default stream).  direction 0: `stream` waits for everything this context has enqueued so far - on its own
 * stream and on its internal ones ("pipeline" map / pose stages, chunked particle ray casts); direction 1: everything
 * the context enqueues from now on waits for what `stream` holds at this moment.  The reference has no counterpart
 * (one Python thread, W12m/slam_ekf.py:63-95); this is what makes a device pointer handed out by
 * slam_grid_counters_dev / slam_grid_live_pmap safe to use from a stream the library does not own:
 *   slam_grid_counters_dev(...); slam_stream_order(ctx, s, 0); <all_reduce on s>; slam_stream_order(ctx, s, 1); */
int slam_stream_order(slam_ctx *ctx, void *stream, int direction);
/* Synchronise and return-and-clear the sticky data error raised by kernels since the
 * last call (SLAM_OK, SLAM_ERR_NAN or SLAM_ERR_OVERFLOW). */
int slam_check_status(slam_ctx *ctx);
/* Tuning knobs.  Maps, cell indices, nearest-neighbour indices and iteration counts never depend on them; the launch
 * shape of the scan matcher ("icp_qpt", and the batch size itself) decides the order in which a pair's sums are added,
 * so transforms and poses of two shapes agree to rounding (1e-13), not bit for bit.
 * "grid_mode": 1 = automatic (default): ray casting through an LDS window per group of scans,
 *   or - for one shared map much larger than a window - rays dealt by direction and swept in
 *   bands through a sheared window (wedges); 0 = direct global atomics; 2 = walks recorded once
 *   and cast tile by tile, wherever that applies; 3 = always the window; 4 = wedges wherever
 *   they apply.
 * "grid_group": scans per workgroup / per tile group, 0 = automatic.
 * "grid_split": LDS-window ray cast with two workgroups per group of scans, one per direction
 *   half (a ray never crosses the column of its origin): -1 = automatic (default: when the launch
 *   cannot fill the chip on its own - shorter launch, a little more total work), 0 = never
 *   (callers that overlap launches of several contexts), 1 = always.
 * "icp_qpt": queries per lane of batched scan matching, 1..3; 0 = by batch size (two for
 *   launches that cannot fill the chip on their own, three from 2 500 pairs of 360 beams or
 *   834 of 1 080 - 7 500 waves at two queries per lane; callers that overlap several smaller
 *   launches set 3).
 * "replay_reset": 1 = slam_replay_dev starts its map from zero (as slam_grid_reset before it would),
 *   clearing the counters inside its scan-matching launch: one dispatch less per replay.  0
 *   (default): the map accumulates across replays until slam_grid_reset.
 * "icp_team": first-iteration queries of a scan without a usable beam window (range jumps between
 *   the two scans): 0 = compacted into a list and searched apart from the lanes that own them
 *   (default), 1 = by the box search of the owning lane.  Same results; an A/B switch.
 * "pipeline": 1 = slam_replay_dev with a map runs as three stages on three streams of the
 *   context (scan matching | pose composition | slam_grid_reset -> ray cast ->
 *   slam_grid_finalize_dev), so the map stage of one replay overlaps the scan matching of the
 *   next; give consecutive replays different poses_out and T_out buffers to benefit (a buffer an
 *   earlier stage is still reading is waited for).  Results of the later stages (poses_out,
 *   pmap_dev) are visible after slam_synchronize / a device synchronise, or to any later call
 *   on this context.  0 = everything on the context's stream (default).
 * "particle_chunks": k > 1 = slam_particles_dev cuts its batch into k chunks: scan matching and pose step of chunk
 *   i + 1 on the context's stream beside the ray cast of chunk i on a second stream.  The call still returns with the
 *   context's stream ordered behind ALL of its work (the second stream is joined at the end), so buffers may be
 *   reused by later calls as with any *_dev entry point.  Measured slower than one piece on MI355X;
 *   0 / 1 = off (default). */
int slam_set_option(slam_ctx *ctx, const char *name, double value);
/* Per-kernel-family timing with HIP events on the context's stream (bench.py roofline).
 * on: 0 = off, 1 = every family, 2 * mask = only the families in mask (bit SLAM_K_*): events on a
 * dispatch cost throughput (3-4 % with four replays overlapping), a caller that needs one family's
 * times pays for one.  read: synchronises, adds up elapsed ms and launch counts since the last reset. */
int slam_timing_enable(slam_ctx *ctx, int on);
int slam_timing_read(slam_ctx *ctx, double ms_out[SLAM_K_COUNT], int64_t launches_out[SLAM_K_COUNT]);

/* ---- ICP ------------------------------------------------------------------------ */
/* Replaces ICP.laserToNumpy (W7/icp.py:182-195) and SLAM_EKF.laserToNumpy
 * (W12m/slam_ekf.py:115-123; clip_inf != 0 applies its inf -> 30 m rule, :119).
 * ranges [B][n] float32; cos_t, sin_t [n] = cos/sin(numpy.linspace(angle_min, angle_max, n))
 * computed by the caller; pts_out [B][2][n] of `dtype`. */
int slam_scan_to_points(slam_ctx *ctx, const float *ranges, const double *cos_t, const double *sin_t,
                        int B, int n, int clip_inf, int dtype, void *pts_out);
int slam_scan_to_points_dev(slam_ctx *ctx, const float *ranges, const double *cos_t, const double *sin_t,
                            int B, int n, int clip_inf, int dtype, void *pts_out);

/* Replaces ICP.findNearest(src, tar) (W12m/icp.py:90-114): brute-force nearest
 * neighbour, lowest index on ties OF THE DISTANCE - the reference compares sqrt of the
 * fused square, so squares that differ in the last places but share a square root tie too
 * (tests/golden/g9, g10) - and (distance 0, index 0) when nothing compares less
 * than inf.  src [B][2][n_src], tar [B][2][n_tar]; dist [B][n_src], idx [B][n_src]. */
int slam_nn(slam_ctx *ctx, const void *src, const void *tar, int B, int n_src, int n_tar, int dtype,
            double *dist, int32_t *idx);
int slam_nn_dev(slam_ctx *ctx, const void *src, const void *tar, int B, int n_src, int n_tar, int dtype,
                double *dist, int32_t *idx);

/* Replaces ICP.getTransform(src, tar) (W12m/icp.py:149-179): rigid 2-D fit of paired
 * rows; src, tar [B][2][n] float64; T_out [B][9].
 * DOCUMENTED DEVIATION: when every row of src or of tar is ONE point (bitwise), W = BB^T.AA is
 * mathematically zero and every rotation is optimal; the reference's SVD then sees the rounding
 * noise of np.mean and returns an arbitrary rotation (tests/golden/g8_collapsed.npz holds eight,
 * 16 .. 170 degrees).  This function returns the canonical R = I, t = centroid_tar - centroid_src;
 * both answers move the source centroid onto the matched point. */
int slam_kabsch2d(slam_ctx *ctx, const double *src, const double *tar, int B, int n, double *T_out);
int slam_kabsch2d_dev(slam_ctx *ctx, const double *src, const double *tar, int B, int n, double *T_out);

/* Replaces ICP.process(tar_pc, src_pc) (W12m/icp.py:38-88) and the loop body of
 * ICP.laserCallback (W7/icp.py:74-92) for B independent pairs in one launch.
 * tar [B][2][n_tar] (or one shared set when tar_shared != 0), src likewise.
 * prior: NULL, or [B][6] row-major 2x3 matrices applied to the source points first
 * (x' = p0*x + p1*y + p2; y' = p3*x + p4*y + p5): the perturbed-prior particle batch of
 * BASELINE.json configs[2]; the returned T maps the perturbed source.
 * T_out [B][9]; iters_out [B] and mean_err_out [B] may be NULL.
 * n_tar, n_src <= 8192.  The rigid fit of every iteration follows slam_kabsch2d, including its
 * canonical R = I for collapsed correspondences (every source point matched to one target). */
int slam_icp_batch(slam_ctx *ctx, const void *tar, const void *src, int B, int n_tar, int n_src,
                   int dtype, int tar_shared, int src_shared, const double *prior, int max_iter,
                   double tol, double *T_out, int32_t *iters_out, double *mean_err_out);
int slam_icp_batch_dev(slam_ctx *ctx, const void *tar, const void *src, int B, int n_tar, int n_src,
                       int dtype, int tar_shared, int src_shared, const double *prior, int max_iter,
                       double tol, double *T_out, int32_t *iters_out, double *mean_err_out);

/* Replaces the pose part of ICP.publishResult(T) (W7/icp.py:153-158 = W12m/icp.py:185-190)
 * applied along L trajectories of n steps: T [L][n][9], pose0 [L][3] -> poses_out [L][n][3]
 * (pose after each step; theta is not wrapped). */
int slam_pose_compose(slam_ctx *ctx, const double *T, const double *pose0, int L, int n, double *poses_out);
int slam_pose_compose_dev(slam_ctx *ctx, const double *T, const double *pose0, int L, int n, double *poses_out);

/* ---- occupancy grid --------------------------------------------------------------- */
/* Replaces Mapping.__init__(xw, yw, xyreso) (W12m/mapping.py:8-20) for G independent
 * maps.  Cell index rule: int(scale * (x + off)) truncated toward zero; the reference
 * hard-codes scale = off_x = off_y = 10 (:33-36).  free_inc / hit_inc / thresh are the
 * +0.01 / +20 / >10 of :43-47 (hit_inc = 4 gives the w12-mapping-online variant,
 * W12o/mapping.py:46).  Evidence is held as integer pass / hit counters; a cell with h hits
 * and p passes is occupied iff the float64 running sum "h x hit_inc, then p x free_inc"
 * (sequential IEEE adds, hits first) exceeds thresh.  For hit_inc > thresh that is the
 * reference's answer whatever the order of arrival; for the +4 variant the reference's own
 * answer is order-dependent when p sits exactly on a threshold, and this canonical order
 * decides.  At most 8 hits may be needed to exceed thresh (else SLAM_ERR_INVALID). */
int slam_grid_create(slam_ctx *ctx, int G, int xw, int yw, double scale, double off_x, double off_y,
                     double free_inc, double hit_inc, double thresh, slam_grid **out);
int slam_grid_destroy(slam_ctx *ctx, slam_grid *grid);
int slam_grid_reset(slam_ctx *ctx, slam_grid *grid);

/* Replaces Mapping.update(ox, oy, center_x, center_y) (W12m/mapping.py:22-51) for B
 * scans: ox, oy [B][n] world-frame beam endpoints, cx, cy [B] ray origins;
 * grid_of_batch [B] selects the map each scan is cast into (NULL: all into map 0). */
int slam_grid_update(slam_ctx *ctx, slam_grid *grid, const double *ox, const double *oy, const double *cx,
                     const double *cy, int B, int n, const int32_t *grid_of_batch);
int slam_grid_update_dev(slam_ctx *ctx, slam_grid *grid, const double *ox, const double *oy,
                         const double *cx, const double *cy, int B, int n, const int32_t *grid_of_batch);

/* Replaces the map-building lines of SLAM_EKF.laserCallback for S scans at once:
 * obs = u2T(xEst).dot(laserToNumpy(msg)) (inf -> 30 m) and mapping.update(obs[0], obs[1],
 * centre) (W12m/slam_ekf.py:88-90, :115-123).  ranges float32 [S][n]; poses [S][3] = xEst of
 * each scan; centres [S][2] = the ray origins, or NULL to cast from the pose as w12-mapping
 * does.  w12-mapping-online takes the centre from /tf instead (W12o/slam_ekf.py:71-77,104).
 * All S scans go into map 0. */
int slam_grid_update_scans(slam_ctx *ctx, slam_grid *grid, const float *ranges, const double *cos_t,
                           const double *sin_t, const double *poses, const double *centres, int S, int n);
int slam_grid_update_scans_dev(slam_ctx *ctx, slam_grid *grid, const float *ranges, const double *cos_t,
                               const double *sin_t, const double *poses, const double *centres, int S, int n);

/* Read map g back (what Mapping.update returns, mapping.py:51, plus the state behind it).
 * Any output may be NULL.  pmap [xw][yw] int8 in {0, 50, 100}; datamap [xw][yw] float64
 * = free_inc*pass + hit_inc*hit; pass, hit [xw][yw] uint32. */
int slam_grid_read(slam_ctx *ctx, slam_grid *grid, int g, int8_t *pmap, double *datamap, uint32_t *pass,
                   uint32_t *hit);
/* Device addresses of the evidence counters, pass and hit, each uint32 [G][xw][yw]: for
 * checkpoint / restore and for merging maps that several GPUs built from disjoint scans - one
 * all_reduce(SUM) over the ranks, in place (integer sums commute, so the merged map is
 * bit-identical for any rank count).  Work enqueued on the context's stream after
 * this call sees every earlier update; a live pmap is marked stale.  The call orders nothing against
 * OTHER streams: a consumer on a stream of its own (a collective on torch's current stream) first makes
 * that stream wait for the context, and the context wait for it afterwards, with slam_stream_order -
 * dist.all_reduce_grid of the Python package does exactly that. */
int slam_grid_counters_dev(slam_ctx *ctx, slam_grid *grid, uint32_t **pass_dev, uint32_t **hit_dev);

/* Keep pmap [G][xw][yw] int8 resident and current on the device and return its address.
 * Ray casts that are the only writer of their map in a launch (one scan group per map: the
 * per-particle maps of slam_particles, Mapping.update of one scan) re-threshold just the cells
 * they could have touched, so no finalize pass over the whole map is needed afterwards; any
 * other update marks it stale and the next finalize / read refreshes it with a full pass.
 * slam_grid_finalize_dev(pmap_dev == that address) then costs nothing when it is current. */
int slam_grid_live_pmap(slam_ctx *ctx, slam_grid *grid, int8_t **pmap_dev_out);
/* Device-side finalize of all G maps into pmap_dev [G][xw][yw] int8 (no synchronise). */
int slam_grid_finalize_dev(slam_ctx *ctx, slam_grid *grid, int8_t *pmap_dev);
/* Replaces the data layout of SLAM_EKF.publishMap (W12m/slam_ekf.py:270-271):
 * data[y*xw + x] = int8(pmap[x][y]). */
int slam_grid_occupancy_data(slam_ctx *ctx, slam_grid *grid, int g, int8_t *data);
/* Number of in-bounds cell visits accumulated since creation / reset (SURVEY.md 8d "C"). */
int slam_grid_visits(slam_ctx *ctx, slam_grid *grid, uint64_t *visits_out);

/* Replaces bresenham(start, end).path (W12m/bresenham.py:2-58), B lines at once.
 * starts, ends [B][2] int32.  lens_out [B] receives each path length; cells_out (may be
 * NULL) receives the paths, line b at cells_out + 2*offsets[b] (x, y interleaved), with
 * room for max(|dx|,|dy|)+1 cells each. */
int slam_bresenham_batch(slam_ctx *ctx, const int32_t *starts, const int32_t *ends, int B,
                         const int64_t *offsets, int32_t *lens_out, int32_t *cells_out, int64_t total_cells);

/* ---- fused replay ------------------------------------------------------------------ */
/* The per-scan unit of BASELINE.json (one ICP.process + one Mapping.update) over L scan
 * streams of n_scan scans: SLAM_EKF.laserCallback (W12m/slam_ekf.py:63-95) without its
 * out-of-scope EKF / landmark steps, the map being cast from the dead-reckoned ICP pose.
 * ranges [L][n_scan][n] float32; pose0 [L][3]; grid may be NULL (ICP + poses only);
 * grid_of_traj [L] or NULL (all into map 0).
 * poses_out [L][n_scan-1][3]; T_out [L][n_scan-1][9], iters_out [L][n_scan-1] may be NULL.
 * dtype: storage type of the intermediate point buffers the ICP reads. */
int slam_replay(slam_ctx *ctx, const float *ranges, const double *cos_t, const double *sin_t, int L,
                int n_scan, int n, int dtype, int max_iter, double tol, const double *pose0,
                slam_grid *grid, const int32_t *grid_of_traj, double *poses_out, double *T_out,
                int32_t *iters_out);
/* Device form.  pts_ws is unused (may be NULL): polar->Cartesian is fused into the ICP
 * kernel, which forms the points of storage type `dtype` in registers; the parameter is
 * kept for ABI stability. */
int slam_replay_dev(slam_ctx *ctx, const float *ranges, const double *cos_t, const double *sin_t, int L,
                    int n_scan, int n, int dtype, int max_iter, double tol, const double *pose0,
                    slam_grid *grid, const int32_t *grid_of_traj, void *pts_ws, double *poses_out,
                    double *T_out, int32_t *iters_out);

/* ---- particle hypotheses (BASELINE.json configs[2]) --------------------------------- */
/* The same two operators batched over P pose hypotheses of ONE scan pair (the reference
 * has no particle filter; this is ICP.process + Mapping.update evaluated P times with
 * perturbed priors).  ranges2 [2][n] float32: previous scan (target) then current scan
 * (source).  prior [P][6] (nullable): 2x3 matrix applied to the source points before the
 * solve; the hypothesis' motion is then M = T.[prior; 0 0 1].  pose_prev [P][3].
 * Per hypothesis p: T_p = ICP.process(tar, prior_p . src) (W12m/icp.py:38-88);
 * pose_p = pose_prev_p (+) M_p (icp.py:185-190); the current scan is ray-cast from pose_p
 * into map p of `grid` (needs G >= P maps; W12m/mapping.py:22-51).
 * poses_out [P][3]; T_out [P][9]; iters_out [P] (host form: T_out / iters_out nullable). */
int slam_particles(slam_ctx *ctx, const float *ranges2, const double *cos_t, const double *sin_t, int n, int dtype,
                   const double *prior, const double *pose_prev, int P, int max_iter, double tol, slam_grid *grid,
                   double *poses_out, double *T_out, int32_t *iters_out);
/* Device form: T_out is required; pts_ws is unused (may be NULL), as in slam_replay_dev. */
int slam_particles_dev(slam_ctx *ctx, const float *ranges2, const double *cos_t, const double *sin_t, int n, int dtype,
                       const double *prior, const double *pose_prev, int P, int max_iter, double tol, slam_grid *grid,
                       void *pts_ws, double *poses_out, double *T_out, int32_t *iters_out);

/* ---- scan-to-map observation -------------------------------------- */
/* W9 = "W9_Fusion Localization (LiDAR Odometry)/course_agv_slam/scripts".
 * Replaces the obstacle extraction of Localization.updateMap (W9/localization.py:54-60):
 * cells > 20 or < -0.5 (occupied and unknown) -> (tx*resolution + origin_x, ty*resolution +
 * origin_y).  map: int8 [height*width]; wire_layout != 0: OccupancyGrid order data[y*width + x]
 * (what the reference receives), else [x][y] (Mapping.pmap order).  Up to cap points are
 * written (in arbitrary order); *count_out receives the number found. */
int slam_map_obstacles(slam_ctx *ctx, const int8_t *map, int width, int height, int wire_layout, double resolution,
                       double origin_x, double origin_y, double *ox, double *oy, int cap, int *count_out);
int slam_map_obstacles_dev(slam_ctx *ctx, const int8_t *map, int width, int height, int wire_layout, double resolution,
                           double origin_x, double origin_y, double *ox, double *oy, int cap, int *count_dev);

/* Replaces Localization.laserEstimation(msg, x) (W9/localization.py:128-150) for B pose
 * hypotheses: obstacle points (ox, oy)[K], poses [B][3] -> ranges_out [B][n] float64, the
 * scan the map would produce (100.0 where no obstacle falls into a beam's bin). */
int slam_virtual_scan(slam_ctx *ctx, const double *ox, const double *oy, int K, const double *poses, int B,
                      double angle_min, double angle_increment, int n, double *ranges_out);
int slam_virtual_scan_dev(slam_ctx *ctx, const double *ox, const double *oy, int K, const double *poses, int B,
                          double angle_min, double angle_increment, int n, double *ranges_out);

/* Replaces Localization.laserToNumpy (W9/localization.py:168-174) for float64 ranges (the
 * virtual scan above is float64, not a float32 wire message): ranges [B][n] ->
 * pts_out [B][2][n] float64. */
int slam_scan_to_points_f64(slam_ctx *ctx, const double *ranges, const double *cos_t, const double *sin_t, int B, int n,
                            double *pts_out);
int slam_scan_to_points_f64_dev(slam_ctx *ctx, const double *ranges, const double *cos_t, const double *sin_t, int B,
                                int n, double *pts_out);

/* Replaces Localization.calc_map_observation(msg) (W9/localization.py:152-157) for B pose
 * hypotheses: virtual scan of the map from poses[b] -> laserToNumpy (:168-174) -> ICP.process
 * against the current scan's points src ([B][2][n] float64, or one shared [2][n] set when
 * src_shared != 0).  cos_t, sin_t [n] as for slam_scan_to_points.  T_out [B][9];
 * iters_out [B] nullable. */
int slam_map_observation(slam_ctx *ctx, const double *ox, const double *oy, int K, const double *poses,
                         const double *src, int B, int n, int src_shared, const double *cos_t, const double *sin_t,
                         double angle_min, double angle_increment, int max_iter, double tol, double *T_out,
                         int32_t *iters_out);
/* Device form: needs workspaces vranges_ws [B][n] and vpts_ws [B][2][n] (float64). */
int slam_map_observation_dev(slam_ctx *ctx, const double *ox, const double *oy, int K, const double *poses,
                             const double *src, int B, int n, int src_shared, const double *cos_t,
                             const double *sin_t, double angle_min, double angle_increment, int max_iter, double tol,
                             double *vranges_ws, double *vpts_ws, double *T_out, int32_t *iters_out);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
