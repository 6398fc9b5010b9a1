/*
 * CPU oracle (plain C, float64) for the ICP + occupancy-grid hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the package, the HIP library)
 * links, loads or calls this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and there only as the checker / the timed CPU
 * baseline ("kind": "port").
 *
 * It restates the reference's algorithm (paths relative to /root/reference):
 *   W12m = "W12_LiDAR SLAM/w12-mapping/course_agv_slam/scripts"
 *   W7   = "W7_Dead Reckoning (ICP)/course_agv_slam/scripts"
 * Each function cites the lines it follows.  Parity is PINNED: tests/test_oracle_golden.py
 * checks every function here against golden vectors produced by running the
 * reference's own Python (oracle/gen_golden.py -> tests/golden/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the compiler fuses nothing, so
 * the arithmetic is the plain IEEE double arithmetic CPython/NumPy perform; the ONE
 * fused multiply-add is written out, in orc_find_nearest, where the reference's BLAS dot has it).
 *
 * Point sets are structure-of-arrays (x[], y[]); the reference's 3xN matrices with a
 * row of ones (icp.py:42-49) carry no extra information.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LASER_RANGE 30.0 /* W12m/slam_ekf.py:18 */

/* a-7: W7/icp.py:182-195, W12m/slam_ekf.py:115-123.  cos_t/sin_t are cos/sin of
 * numpy.linspace(angle_min, angle_max, n), computed by the caller with NumPy so the
 * trigonometry is the reference's own. clip_inf reproduces slam_ekf.py:119. */
void orc_laser_to_points(const float *ranges, const double *cos_t, const double *sin_t, int n,
                         int clip_inf, double *x, double *y)
{
    for (int i = 0; i < n; ++i) {
        double r = (double)ranges[i];
        if (clip_inf && isinf(r) && r > 0) r = ORC_MAX_LASER_RANGE;
        x[i] = cos_t[i] * r;
        y[i] = sin_t[i] * r;
    }
}

/* a-4: W12m/icp.py:90-114.  For every source point the first target point at the
 * smallest Euclidean distance (strict '<', :103).  A NaN or inf distance never wins,
 * leaving distance 0 / index 0 (:96-97). */
/* Which ordering decides: 0 = the reference's (the distance, sqrt of the fused square), 1 = the device
 * kernels' (the fused square itself; DESIGN.md "ordering of sub-ulp near-ties").  They pick the same target
 * unless two squares differ in the last places and share a square root; orc_nn_rule_splits() counts the
 * queries for which they did not (since the last reset), whatever the rule in force - so a test can tell
 * "the documented difference occurred in this input" from "something else is wrong". */
static int g_nn_rule = 0;
static long g_nn_splits = 0;
void orc_set_nn_rule(int rule) { g_nn_rule = rule; }
long orc_nn_rule_splits(int reset)
{
    long v = g_nn_splits;
    if (reset) g_nn_splits = 0;
    return v;
}

void orc_find_nearest(const double *sx, const double *sy, int n, const double *tx, const double *ty,
                      int m, double *dist, int32_t *idx)
{
    long splits = 0;
    for (int i = 0; i < n; ++i) {
        double min_dist = INFINITY, min_d2 = INFINITY;
        int j_ref = 0, j_dev = 0;
        double d_ref = 0.0, d_dev = 0.0;
        for (int j = 0; j < m; ++j) {
            double dx = sx[i] - tx[j];
            double dy = sy[i] - ty[j];
            /* np.linalg.norm of a 2-vector (:102) is sqrt(x.dot(x)), and the two-element dot is BLAS
             * arithmetic: fma(x1, x1, x0*x0) in the NumPy / OpenBLAS the golden vectors were made with
             * (tests/test_nn_near_ties.py checks that it still is; the fused and the unfused square differ
             * in the last place for a quarter of all inputs, which decides near-ties of symmetric clouds) */
            double d2 = fma(dy, dy, dx * dx);
            double d = sqrt(d2);
            if (d < min_dist) {
                min_dist = d;
                j_ref = j;
                d_ref = d;
            }
            if (d2 < min_d2 && d2 < INFINITY) {                     /* (an infinite distance never wins, as above) */
                min_d2 = d2;
                j_dev = j;
                d_dev = d;
            }
        }
        splits += j_ref != j_dev;
        idx[i] = g_nn_rule ? j_dev : j_ref;
        dist[i] = g_nn_rule ? d_dev : d_ref;
    }
    if (splits) {
#pragma omp atomic
        g_nn_splits += splits;
    }
}

/* 2x2 singular value decomposition W = U diag(s) Vt with s[0] >= s[1] >= 0, by the
 * closed-form two-angle factorisation (stands in for numpy.linalg.svd at icp.py:161;
 * U.Vt is independent of the sign/ordering conventions LAPACK happens to use). */
static void svd2x2(const double w[4], double u[4], double s[2], double vt[4])
{
    double a = w[0], b = w[1], c = w[2], d = w[3];
    double e = 0.5 * (a + d), f = 0.5 * (a - d), g = 0.5 * (c + b), h = 0.5 * (c - b);
    double q = hypot(e, h), r = hypot(f, g);
    double a1 = atan2(g, f), a2 = atan2(h, e);
    double th = 0.5 * (a2 - a1), ph = 0.5 * (a2 + a1);
    double sy = q - r;
    double sg = sy < 0 ? -1.0 : 1.0;
    u[0] = cos(ph); u[1] = -sin(ph); u[2] = sin(ph); u[3] = cos(ph);
    s[0] = q + r; s[1] = fabs(sy);
    vt[0] = cos(th); vt[1] = -sin(th); vt[2] = sg * sin(th); vt[3] = sg * cos(th);
}

/* a-5: W12m/icp.py:149-179 (the Vt[1,:] reflection fix of :168; W7/icp.py:136 indexes
 * Vt[2,:] and cannot run).  a = source rows, b = target rows, paired.  T is 3x3
 * row-major [[R, t], [0, 0, 1]]. */
void orc_get_transform(const double *ax, const double *ay, const double *bx, const double *by, int n,
                       double T[9])
{
    double cax = 0, cay = 0, cbx = 0, cby = 0;
    for (int k = 0; k < n; ++k) { cax += ax[k]; cay += ay[k]; cbx += bx[k]; cby += by[k]; }
    cax /= n; cay /= n; cbx /= n; cby /= n;                       /* :154-155 */
    double w[4] = {0, 0, 0, 0};                                   /* W = BB^T . AA  (:160) */
    for (int k = 0; k < n; ++k) {
        double aax = ax[k] - cax, aay = ay[k] - cay, bbx = bx[k] - cbx, bby = by[k] - cby;
        w[0] += bbx * aax; w[1] += bbx * aay; w[2] += bby * aax; w[3] += bby * aay;
    }
    /* DOCUMENTED DEVIATION (collapsed sets): when every row of a or of b is ONE point, W is
     * mathematically zero and every rotation is optimal; the reference's centred rows are
     * rounding noise instead (np.mean of n equal values is not that value), W ~ 1e-31, and
     * the rotation its SVD returns is arbitrary.  Oracles and product return the canonical
     * R = I, t = centroid_B - centroid_A (the SVD of an exact zero matrix);
     * tests/golden/g8_collapsed.npz records what the reference itself returns. */
    {
        int same_a = 1, same_b = 1;
        for (int k = 1; k < n; ++k) {
            if (!(ax[k] == ax[0] && ay[k] == ay[0])) same_a = 0;
            if (!(bx[k] == bx[0] && by[k] == by[0])) same_b = 0;
        }
        if (same_a || same_b) w[0] = w[1] = w[2] = w[3] = 0.0;
    }
    double u[4], s[2], vt[4], r[4];
    svd2x2(w, u, s, vt);                                          /* :161 */
    r[0] = u[0] * vt[0] + u[1] * vt[2]; r[1] = u[0] * vt[1] + u[1] * vt[3];
    r[2] = u[2] * vt[0] + u[3] * vt[2]; r[3] = u[2] * vt[1] + u[3] * vt[3];   /* :162 */
    if (r[0] * r[3] - r[1] * r[2] < 0) {                          /* :164-169 */
        vt[2] = -vt[2]; vt[3] = -vt[3];
        r[0] = u[0] * vt[0] + u[1] * vt[2]; r[1] = u[0] * vt[1] + u[1] * vt[3];
        r[2] = u[2] * vt[0] + u[3] * vt[2]; r[3] = u[2] * vt[1] + u[3] * vt[3];
    }
    T[0] = r[0]; T[1] = r[1]; T[2] = cbx - (r[0] * cax + r[1] * cay);   /* :172 */
    T[3] = r[2]; T[4] = r[3]; T[5] = cby - (r[2] * cax + r[3] * cay);
    T[6] = 0; T[7] = 0; T[8] = 1;
}

/* a-3: W12m/icp.py:38-88.  Returns the number of iterations run; *mean_err receives
 * the last mean nearest-neighbour distance (:75). */
int orc_icp_process(const double *tx, const double *ty, int m, const double *sx0, const double *sy0,
                    int n, int max_iter, double tol, double T[9], double *mean_err)
{
    double *buf = (double *)malloc(sizeof(double) * 5 * (size_t)n);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    double *sx = buf, *sy = buf + n, *dist = buf + 2 * n, *mx = buf + 3 * n, *my = buf + 4 * n;
    memcpy(sx, sx0, sizeof(double) * n);
    memcpy(sy, sy0, sizeof(double) * n);
    double pre_error = 0.0, me = 0.0;
    int it = 0;
    for (int i = 0; i < max_iter; ++i) {
        orc_find_nearest(sx, sy, n, tx, ty, m, dist, idx);        /* :67 */
        for (int k = 0; k < n; ++k) { mx[k] = tx[idx[k]]; my[k] = ty[idx[k]]; }
        double Ti[9];
        orc_get_transform(sx, sy, mx, my, n, Ti);                 /* :69 */
        for (int k = 0; k < n; ++k) {                             /* src = T . src (:71) */
            double x = Ti[0] * sx[k] + Ti[1] * sy[k] + Ti[2];
            double y = Ti[3] * sx[k] + Ti[4] * sy[k] + Ti[5];
            sx[k] = x; sy[k] = y;
        }
        ++it;
        double sum = 0;
        for (int k = 0; k < n; ++k) sum += dist[k];
        me = sum / n;                                             /* :75 */
        if (fabs(pre_error - me) < tol) break;                    /* :76-77 */
        pre_error = me;
    }
    orc_get_transform(sx0, sy0, sx, sy, n, T);                    /* :81 */
    if (mean_err) *mean_err = me;
    free(buf); free(idx);
    return it;
}

/* B independent pairs, OpenMP across pairs (the cpu_baseline of bench.py).  tar/src are
 * [B][2][n] (x row then y row). */
void orc_icp_batch(const double *tar, const double *src, int B, int m, int n, int max_iter, double tol,
                   double *T_out, int32_t *iters_out, double *err_out)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const double *t = tar + (size_t)b * 2 * m, *s = src + (size_t)b * 2 * n;
        iters_out[b] = orc_icp_process(t, t + m, m, s, s + n, n, max_iter, tol, T_out + 9 * (size_t)b,
                                       err_out ? err_out + b : 0);
    }
}

/* a-6: W7/icp.py:153-158 = W12m/icp.py:185-190; theta is not wrapped. */
void orc_compose_pose(double sta[3], const double T[9])
{
    double dyaw = atan2(T[3], T[0]);
    double x = sta[0] + cos(sta[2]) * T[2] - sin(sta[2]) * T[5];
    double y = sta[1] + sin(sta[2]) * T[2] + cos(sta[2]) * T[5];
    sta[0] = x; sta[1] = y; sta[2] = sta[2] + dyaw;
}

/* a-8: W12m/slam_ekf.py:89 with u2T of :130-137: world = [[c,-s,x],[s,c,y]] . [px;py;1]. */
void orc_world_points(const double pose[3], const double *px, const double *py, int n, double *ox, double *oy)
{
    double c = cos(pose[2]), s = sin(pose[2]);
    for (int i = 0; i < n; ++i) {
        ox[i] = c * px[i] + (-s) * py[i] + pose[0] * 1.0;
        oy[i] = s * px[i] + c * py[i] + pose[1] * 1.0;
    }
}

/* a-11: W12m/bresenham.py:2-58.  Writes up to cap cells (x,y interleaved) in path
 * order start -> end and returns the path length (0 for identical endpoints, :10-11).
 * The error term is a float64 running sum of dy/float(dx) (:35,:51), compared with 0.5
 * and decremented by 1.0 (:53-55): NOT integer Bresenham. */
int orc_bresenham(int x0, int y0, int x1, int y1, int32_t *xy, int cap)
{
    if (x0 == x1 && y0 == y1) return 0;
    int steep = abs(y1 - y0) > abs(x1 - x0);                      /* :14 */
    if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
    int flag = 0;
    if (x0 > x1) { flag = 1; int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }   /* :19-29 */
    int dx = x1 - x0, dy = abs(y1 - y0);
    double error = 0.0, derr = (double)dy / (double)dx;           /* :34-35 */
    int y = y0, ystep = y0 < y1 ? 1 : -1;                         /* :40-43 */
    int len = dx + 1;
    for (int k = 0; k <= dx; ++k) {
        int x = x0 + k;
        int j = flag ? dx - k : k;                                /* reverse() of :57-58 */
        if (j < cap) {
            xy[2 * j] = steep ? y : x;
            xy[2 * j + 1] = steep ? x : y;
        }
        error += derr;                                            /* :51 */
        if (error >= 0.5) { y += ystep; error -= 1.0; }           /* :53-55 */
    }
    return len;
}

/* Property of the walk above that the byte-window ray cast relies on (k_grid_update_owner8): the float-error walk
 * of bresenham.py:45-55 ends in the cell of its other end, i.e. after dx error updates exactly dy of them stepped y.
 * Counts the (dx, dy) pairs with 1 <= dx <= max_dx, 0 <= dy <= dx for which it does not (expected: none). */
long orc_walk_end_mismatches(int max_dx)
{
    long bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : bad)
    for (int dx = 1; dx <= max_dx; ++dx)
        for (int dy = 0; dy <= dx; ++dy) {
            double error = 0.0, derr = (double)dy / (double)dx;       /* :34-35 */
            int y = 0;
            for (int k = 0; k < dx; ++k) {
                error += derr;                                    /* :51 */
                if (error >= 0.5) { ++y; error -= 1.0; }          /* :53-55 */
            }
            bad += y != dy;
        }
    return bad;
}

/* a-9/a-10: W12m/mapping.py:8-51 with the index rule generalised to
 * int(scale*(x+offset)) (the reference hard-codes 10 and 10, :33-36).  State: datamap
 * (float64, the reference's own accumulator), pmap, and integer pass / hit counters
 * that restate the same evidence.  Arrays are [xw][yw] row-major as in mapping.py:14. */
typedef struct {
    int xw, yw;
    double scale, off_x, off_y, free_inc, hit_inc, thresh;
    double *datamap;
    int8_t *pmap;
    uint32_t *pass_cnt, *hit_cnt;
} orc_grid;

orc_grid *orc_grid_create(int xw, int yw, double scale, double off_x, double off_y, double free_inc,
                          double hit_inc, double thresh)
{
    orc_grid *g = (orc_grid *)calloc(1, sizeof(orc_grid));
    size_t c = (size_t)xw * yw;
    g->xw = xw; g->yw = yw; g->scale = scale; g->off_x = off_x; g->off_y = off_y;
    g->free_inc = free_inc; g->hit_inc = hit_inc; g->thresh = thresh;
    g->datamap = (double *)calloc(c, sizeof(double));
    g->pmap = (int8_t *)malloc(c);
    memset(g->pmap, 50, c);                                       /* :14 */
    g->pass_cnt = (uint32_t *)calloc(c, sizeof(uint32_t));
    g->hit_cnt = (uint32_t *)calloc(c, sizeof(uint32_t));
    return g;
}

void orc_grid_destroy(orc_grid *g)
{
    if (!g) return;
    free(g->datamap); free(g->pmap); free(g->pass_cnt); free(g->hit_cnt); free(g);
}

double *orc_grid_datamap(orc_grid *g) { return g->datamap; }
int8_t *orc_grid_pmap(orc_grid *g) { return g->pmap; }
uint32_t *orc_grid_pass(orc_grid *g) { return g->pass_cnt; }
uint32_t *orc_grid_hit(orc_grid *g) { return g->hit_cnt; }

/* mapping.py:22-51.  Returns the number of in-bounds cell visits (the C of SURVEY.md
 * 8(d): algorithmic grid bytes = 9*C). */
long orc_grid_update(orc_grid *g, const double *ox, const double *oy, int n, double cx, double cy)
{
    long visits = 0;
    for (int i = 0; i < n; ++i) {
        if (isinf(ox[i])) continue;                               /* :30 (x only) */
        int px_o = (int)(g->scale * (ox[i] + g->off_x));          /* :33-36: truncation toward zero */
        int py_o = (int)(g->scale * (oy[i] + g->off_y));
        int px_c = (int)(g->scale * (cx + g->off_x));
        int py_c = (int)(g->scale * (cy + g->off_y));
        if (px_o == px_c && py_o == py_c) continue;               /* empty path */
        /* walk the path in path order without materialising it */
        int x0 = px_c, y0 = py_c, x1 = px_o, y1 = py_o;
        int steep = abs(y1 - y0) > abs(x1 - x0);
        if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
        int flag = 0;
        if (x0 > x1) { flag = 1; int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
        int dx = x1 - x0, dy = abs(y1 - y0);
        double error = 0.0, derr = (double)dy / (double)dx;
        int y = y0, ystep = y0 < y1 ? 1 : -1;
        /* datamap must be accumulated in path order (start -> end) to reproduce the
         * reference's float sums, so a reversed walk is buffered first. */
        int32_t *cells = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(dx + 1));
        for (int k = 0; k <= dx; ++k) {
            int x = x0 + k, j = flag ? dx - k : k;
            cells[2 * j] = steep ? y : x;
            cells[2 * j + 1] = steep ? x : y;
            error += derr;
            if (error >= 0.5) { y += ystep; error -= 1.0; }
        }
        for (int j = 0; j <= dx; ++j) {                           /* :39-50 */
            int lpx = cells[2 * j], lpy = cells[2 * j + 1];
            if (lpx >= 0 && lpx < g->xw && lpy >= 0 && lpy < g->yw) {
                size_t c = (size_t)lpx * g->yw + lpy;
                if (j < dx) { g->datamap[c] += g->free_inc; g->pass_cnt[c] += 1; }
                else        { g->datamap[c] += g->hit_inc;  g->hit_cnt[c] += 1; }
                g->pmap[c] = g->datamap[c] > g->thresh ? 100 : 0;
                ++visits;
            }
        }
        free(cells);
    }
    return visits;
}

/* Smallest k whose float64 running sum of k additions of free_inc exceeds thresh
 * (mapping.py:43,47): 1001 for (0.01, 10). */
static int pass_threshold_from(double base, double free_inc, double thresh)
{
    double acc = base;
    int k = 0;
    while (!(acc > thresh)) { acc += free_inc; ++k; }
    return k;
}

int orc_pass_count_threshold(double free_inc, double thresh) { return pass_threshold_from(0.0, free_inc, thresh); }

/* Canonical integer rule behind pmap (include/slam_hip.h, slam_grid_create): h hits and p
 * passes are occupied iff h >= levels or p >= table[h]; the float sum is taken hits first.
 * Returns levels (<= cap). */
int orc_occupied_rule(double free_inc, double hit_inc, double thresh, int *table, int cap)
{
    double base = 0.0;
    int levels = 0;
    while (!(base > thresh) && levels < cap) {
        table[levels++] = pass_threshold_from(base, free_inc, thresh);
        base += hit_inc;
    }
    return levels;
}

static int8_t occupied_value(uint32_t ps, uint32_t ht, const int *table, int levels)
{
    if (!(ps | ht)) return 50;
    if (ht >= (uint32_t)levels) return 100;
    return ps >= (uint32_t)table[ht] ? 100 : 0;
}

/* W12m/slam_ekf.py:270-271: data[y*width + x] = int8(trunc(pmap[x][y])). */
void orc_occupancy_grid_data(const orc_grid *g, int8_t *data)
{
    for (int y = 0; y < g->yw; ++y)
        for (int x = 0; x < g->xw; ++x)
            data[(size_t)y * g->xw + x] = g->pmap[(size_t)x * g->yw + y];
}

/* Pipeline a-7 -> a-3 -> a-6 -> a-8 -> a-10 over one scan stream (the unit bench.py
 * counts: one ICP.process + one Mapping.update per scan after the first;
 * slam_ekf.py:63-95 minus the out-of-scope EKF / landmark steps).  ranges is
 * [n_scan][n] float32.  With threads > 1 the ICP solves (independent per pair) run under
 * OpenMP; pose composition and the grid stay serial and in stream order.  Returns the
 * number of in-bounds grid cell visits. */
long orc_replay(const float *ranges, int n_scan, int n, const double *cos_t, const double *sin_t,
                int max_iter, double tol, const double pose0[3], orc_grid *g, double *poses_out /*[n_scan-1][3]*/,
                double *T_out /*[n_scan-1][9]*/, int32_t *iters_out, int threads)
{
    size_t np = (size_t)n_scan * n;
    double *px = (double *)malloc(sizeof(double) * np), *py = (double *)malloc(sizeof(double) * np);
    for (int k = 0; k < n_scan; ++k)
        orc_laser_to_points(ranges + (size_t)k * n, cos_t, sin_t, n, 1, px + (size_t)k * n, py + (size_t)k * n);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int k = 1; k < n_scan; ++k)
        iters_out[k - 1] = orc_icp_process(px + (size_t)(k - 1) * n, py + (size_t)(k - 1) * n, n,
                                           px + (size_t)k * n, py + (size_t)k * n, n, max_iter, tol,
                                           T_out + 9 * (size_t)(k - 1), 0);
    double sta[3] = {pose0[0], pose0[1], pose0[2]};
    double *ox = (double *)malloc(sizeof(double) * n), *oy = (double *)malloc(sizeof(double) * n);
    long visits = 0;
    for (int k = 1; k < n_scan; ++k) {
        orc_compose_pose(sta, T_out + 9 * (size_t)(k - 1));
        memcpy(poses_out + 3 * (size_t)(k - 1), sta, sizeof(sta));
        if (g) {
            orc_world_points(sta, px + (size_t)k * n, py + (size_t)k * n, n, ox, oy);
            visits += orc_grid_update(g, ox, oy, n, sta[0], sta[1]);
        }
    }
    free(px); free(py); free(ox); free(oy);
    return visits;
}

/* Counters-only, OpenMP-parallel ray casting of one scan (used by orc_replay_mt):
 * the same walk as orc_grid_update, integer evidence only, atomic adds. */
static long grid_update_counts(orc_grid *g, const double *ox, const double *oy, int n, double cx, double cy)
{
    long visits = 0;
    for (int i = 0; i < n; ++i) {
        if (isinf(ox[i])) continue;
        int x1 = (int)(g->scale * (ox[i] + g->off_x)), y1 = (int)(g->scale * (oy[i] + g->off_y));
        int x0 = (int)(g->scale * (cx + g->off_x)), y0 = (int)(g->scale * (cy + g->off_y));
        if (x0 == x1 && y0 == y1) continue;
        int steep = abs(y1 - y0) > abs(x1 - x0);
        if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
        int flag = 0;
        if (x0 > x1) { flag = 1; int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
        int dx = x1 - x0, dy = abs(y1 - y0);
        double error = 0.0, derr = (double)dy / (double)dx;
        int y = y0, ystep = y0 < y1 ? 1 : -1;
        for (int k = 0; k <= dx; ++k) {
            int x = x0 + k;
            int lx = steep ? y : x, ly = steep ? x : y;
            int last = flag ? (k == 0) : (k == dx);
            if (lx >= 0 && lx < g->xw && ly >= 0 && ly < g->yw) {
                uint32_t *c = (last ? g->hit_cnt : g->pass_cnt) + (size_t)lx * g->yw + ly;
#pragma omp atomic
                *c += 1;
                ++visits;
            }
            error += derr;
            if (error >= 0.5) { y += ystep; error -= 1.0; }
        }
    }
    return visits;
}

/* All-cores CPU baseline of the replay for bench.py ("cpu_baseline", kind "port"): ICP
 * solves in parallel over pairs, poses composed serially, rays cast in parallel over scans
 * into the integer counters, pmap from the counter rule (pass >= k* or hit >= 1).  Same
 * results as orc_replay for poses, counters and pmap; datamap is not maintained. */
long orc_replay_mt(const float *ranges, int n_scan, int n, const double *cos_t, const double *sin_t, int max_iter,
                   double tol, const double pose0[3], orc_grid *g, double *poses_out, double *T_out,
                   int32_t *iters_out, int threads)
{
    if (threads < 1) threads = 1;
    size_t np = (size_t)n_scan * n;
    double *px = (double *)malloc(sizeof(double) * np), *py = (double *)malloc(sizeof(double) * np);
#pragma omp parallel for num_threads(threads)
    for (int k = 0; k < n_scan; ++k)
        orc_laser_to_points(ranges + (size_t)k * n, cos_t, sin_t, n, 1, px + (size_t)k * n, py + (size_t)k * n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int k = 1; k < n_scan; ++k)
        iters_out[k - 1] = orc_icp_process(px + (size_t)(k - 1) * n, py + (size_t)(k - 1) * n, n,
                                           px + (size_t)k * n, py + (size_t)k * n, n, max_iter, tol,
                                           T_out + 9 * (size_t)(k - 1), 0);
    double sta[3] = {pose0[0], pose0[1], pose0[2]};
    for (int k = 1; k < n_scan; ++k) {
        orc_compose_pose(sta, T_out + 9 * (size_t)(k - 1));
        memcpy(poses_out + 3 * (size_t)(k - 1), sta, sizeof(sta));
    }
    long visits = 0;
    if (g) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : visits)
        for (int k = 1; k < n_scan; ++k) {
            double *ox = (double *)malloc(sizeof(double) * 2 * (size_t)n), *oy = ox + n;
            const double *p = poses_out + 3 * (size_t)(k - 1);
            orc_world_points(p, px + (size_t)k * n, py + (size_t)k * n, n, ox, oy);
            visits += grid_update_counts(g, ox, oy, n, p[0], p[1]);
            free(ox);
        }
        int table[64];
        int levels = orc_occupied_rule(g->free_inc, g->hit_inc, g->thresh, table, 64);
        size_t cells = (size_t)g->xw * g->yw;
#pragma omp parallel for num_threads(threads)
        for (size_t c = 0; c < cells; ++c) {
            uint32_t ps = g->pass_cnt[c], ht = g->hit_cnt[c];
            g->pmap[c] = occupied_value(ps, ht, table, levels);
        }
    }
    free(px); free(py);
    return visits;
}

/* f-1: W9 localization.py:128-150 (laserEstimation).  W9 = "W9_Fusion Localization (LiDAR
 * Odometry)/course_agv_slam/scripts".  obstacle points (ox, oy)[K], pose (x, y, theta) ->
 * ranges[total_num] (float64; 100.0 where no obstacle falls into the beam's bin). */
void orc_laser_estimation(const double *ox, const double *oy, int K, const double pose[3], double angle_min,
                          double angle_increment, int total_num, double *ranges)
{
    for (int i = 0; i < total_num; ++i) ranges[i] = 100.0;
    for (int i = 0; i < K; ++i) {
        double dist = hypot(pose[0] - ox[i], pose[1] - oy[i]);
        double q = (atan2(oy[i] - pose[1], ox[i] - pose[0]) - angle_min - pose[2]) / angle_increment;
        if (!(fabs(q) < 2.0e9)) continue;               /* NaN / huge: Python would raise or loop; skipped */
        long index = (long)q;                             /* int(): truncation toward zero */
        while (index > total_num - 1) index -= total_num;
        while (index < 0) index += total_num;
        if (dist < ranges[index]) ranges[index] = dist;
    }
}

/* W9 localization.py:54-60 (updateMap): OccupancyGrid data[y*width + x] -> obstacle list, in
 * numpy.nonzero order of map_data[x][y]; cells > 20 or < -0.5 (occupied and unknown). */
int orc_map_obstacles(const int8_t *data, int width, int height, double resolution, double origin_x, double origin_y,
                      double *ox, double *oy)
{
    int k = 0;
    for (int x = 0; x < width; ++x)
        for (int y = 0; y < height; ++y) {
            int v = data[(size_t)y * width + x];
            if (v > 20 || v < 0) { ox[k] = (x * resolution + origin_x) * 1.0; oy[k] = (y * resolution + origin_y) * 1.0; ++k; }
        }
    return k;
}
