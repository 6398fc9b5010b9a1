// ICP scan-matching kernels for gfx950 (MI355X): brute-force nearest neighbour,
// closed-form 2-D Kabsch, the whole ICP.process loop in one launch, polar->Cartesian
// conversion and dead-reckoning pose composition.
//
// Functional spec = the reference's Python (paths relative to /root/reference,
// W12m = "W12_LiDAR SLAM/w12-mapping/course_agv_slam/scripts"):
//   ICP.process      W12m/icp.py:38-88      -> k_icp
//   ICP.findNearest  W12m/icp.py:90-114     -> nn_polar / nn_listed / nn_search / nn_exact / k_nn
//   ICP.getTransform W12m/icp.py:149-179    -> kabsch_from_sums / k_kabsch
//   laserToNumpy     W12m/slam_ekf.py:115-123 -> k_scan_to_points
//   publishResult    W12m/icp.py:185-190    -> k_pose_compose
//
// Design (DESIGN.md "K2"): one workgroup per scan pair, one to three query points per lane.  The
// target is staged once in LDS as float64 (x, y) pairs; the source points, their original copies
// and the running best live in registers for the whole solve, so HBM sees each range once.
// Nearest neighbours of a scan are searched in a window of beam indices around a guess (nn_polar),
// first-iteration queries whose guess bounds nothing are compacted into an LDS list and searched
// apart from their lanes (nn_listed), the box search (nn_search) takes what remains.  The first
// iteration forms centroids and centred products in two reductions as the reference does, the later
// ones in one reduction about the previous matches' centroid.  All arithmetic is float64 (the
// reference is float64 and a float32 distance would flip near-tied neighbours, moving the pose by
// ~1e-4).  The library is compiled with -ffp-contract=off: fused multiply-adds appear
// only where written as fma().
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <type_traits>

#include "slam_internal.h"
#include "slam_stamps.h"

namespace slam {

#ifdef SLAM_STAMPS_ICP
__device__ unsigned long long g_polar_lanes[4];
hipError_t debug_polar_lanes(unsigned long long out[4], bool clear)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_polar_lanes), sizeof(unsigned long long) * 4);
    if (e == hipSuccess && clear) {
        const unsigned long long zero[4] = {0ull, 0ull, 0ull, 0ull};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_polar_lanes), zero, sizeof zero);
    }
    return e;
}
#endif

__device__ __forceinline__ double ld(const double *p, long i) { return p[i]; }
__device__ __forceinline__ double ld(const float *p, long i) { return (double)p[i]; }
__device__ __forceinline__ double ld(const __half *p, long i) { return (double)__half2float(p[i]); }
__device__ __forceinline__ void st(double *p, long i, double v) { p[i] = v; }
__device__ __forceinline__ void st(float *p, long i, double v) { p[i] = (float)v; }
// float64 -> float16 with ONE rounding (as numpy.astype(float16) does): go through a
// float32 rounded to odd, so the final round-to-nearest sees the sticky information.
__device__ __forceinline__ void st(__half *p, long i, double v)
{
    float f = (float)v;
    if ((double)f != v && v == v) {
        unsigned u = __float_as_uint(f);
        if (fabs((double)f) > fabs(v)) u -= 1u;   // back to the truncated magnitude
        f = __uint_as_float(u | 1u);
    }
    p[i] = __float2half_rn(f);
}

// value of x after a store to / load from storage type T (the rounding a point buffer applies)
__device__ __forceinline__ double round_as(double v, const double *) { return v; }
__device__ __forceinline__ double round_as(double v, const float *) { return (double)(float)v; }
__device__ __forceinline__ double round_as(double v, const __half *)
{
    __half h;
    st(&h, 0, v);
    return (double)__half2float(h);
}

// Wave-wide sum of a double, result in every lane, without touching LDS: four DPP
// exchange steps inside each row of 16 lanes (xor 1, xor 2, half-row mirror, row mirror;
// addition is commutative, so both partners of an exchange hold the same bits), then the
// four row sums are read with v_readlane and added in a fixed order.  Needs all 64 lanes
// active.  Deterministic run to run and identical in all lanes.
template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // (bound_ctrl set: every lane has a source lane in these patterns, and the compiler need not
    // initialise the destination first - that was one extra v_mov per exchange)
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane),
                            __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ double wave_sum_f64(double x)
{
    x += dpp_xchg<0xB1>(x);    // quad_perm [1,0,3,2]
    x += dpp_xchg<0x4E>(x);    // quad_perm [2,3,0,1]
    x += dpp_xchg<0x141>(x);   // row_half_mirror
    x += dpp_xchg<0x140>(x);   // row_mirror
    return (readlane_f64(x, 0) + readlane_f64(x, 16)) + (readlane_f64(x, 32) + readlane_f64(x, 48));
}

// Sum of nwaves doubles in LDS, `stride` apart, added in ascending order to 0.0 - the cross-wave stage of the
// reductions below.  The reads are issued together (a loop over a run-time count waits for every read before it
// issues the next: 150 cycles a wave, twice per iteration on the critical path of a solve and eight times behind it).
template <int NW>
__device__ __forceinline__ double lds_column_n(const double *p, int stride)
{
    double x[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) x[w] = p[w * stride];
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += x[w];
    return t;
}
__device__ __forceinline__ double lds_column(const double *p, int stride, int nwaves)
{
    switch (nwaves) {                      // (the workgroup shapes of the benchmark configurations first)
    case 3: return lds_column_n<3>(p, stride);
    case 2: return lds_column_n<2>(p, stride);
    case 6: return lds_column_n<6>(p, stride);
    case 9: return lds_column_n<9>(p, stride);
    case 4: return lds_column_n<4>(p, stride);
    case 5: return lds_column_n<5>(p, stride);
    case 8: return lds_column_n<8>(p, stride);
    case 16: return lds_column_n<16>(p, stride);
    }
    double t = 0.0;
    int w = 0;
    for (; w + 4 <= nwaves; w += 4) {
        const double x0 = p[w * stride], x1 = p[(w + 1) * stride], x2 = p[(w + 2) * stride], x3 = p[(w + 3) * stride];
        t += x0; t += x1; t += x2; t += x3;
    }
    for (; w < nwaves; ++w) t += p[w * stride];
    return t;
}

// Sum NV doubles over the workgroup; every thread receives the (bitwise identical)
// totals.  Fixed exchange pattern + fixed wave order: deterministic run to run.  `scratch`
// must alternate between two buffers on consecutive calls (no trailing
// barrier).  ([NV][nwaves] doubles.)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch, int nwaves, int wave, int lane)
{
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_f64(v[k]);
    if (nwaves == 1) return;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) scratch[k * nwaves + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lds_column(scratch + k * nwaves, 1, nwaves);
}

// Transposed wave reduction: N = 8 (or 4) values are summed over the wave's 64 lanes in 58 (33)
// instructions instead of N x 23, by halving the number of live values at every exchange: at a
// step whose partner differs in selection bit sigma, a lane KEEPS one value of each pair and sends
// the other.  The exchanges are the DPP patterns of wave_sum_f64 (xor 1, xor 2, xor 7, xor 15
// inside a row of 16 lanes), then the rows are added with v_permlane16_swap / v_permlane32_swap.
// For a partner to hold the same value indices at every later step, the selection bits are
// sigma1 = b0^b2, sigma2 = b1^b2, sigma3 = b2^b3 of the lane number: each flips under its own step's
// xor mask and under none of the later ones.  Afterwards lane l holds the wave total of value
// sigma1 + 2 sigma2 (+ 4 sigma3): lanes 0..3 hold values 0..3, lanes 7, 6, 5, 4 values 4..7.
// Fixed order: deterministic, the same in every lane that holds the same value.
struct LaneSel { bool s1, s2, s3; int idx8, idx4; };
__device__ __forceinline__ LaneSel lane_sel(int lane)
{
    LaneSel s;
    s.s1 = ((lane ^ (lane >> 2)) & 1) != 0;
    s.s2 = (((lane >> 1) ^ (lane >> 2)) & 1) != 0;
    s.s3 = (((lane >> 2) ^ (lane >> 3)) & 1) != 0;
    s.idx4 = (s.s1 ? 1 : 0) + (s.s2 ? 2 : 0);
    s.idx8 = s.idx4 + (s.s3 ? 4 : 0);
    return s;
}
template <int CTRL>
__device__ __forceinline__ double tstep(double a, double b, bool sel)    // sel: keep b, send a
{
    const double keep = sel ? b : a, send = sel ? a : b;
    return keep + dpp_xchg<CTRL>(send);
}
__device__ __forceinline__ double rows_sum_f64(double x)                 // x holds a 16-lane row total in every lane of the row
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)lo, (unsigned)lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap((unsigned)hi, (unsigned)hi, false, false);
    x = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);     // rows 0+1 | 2+3
    lo = __double2loint(x); hi = __double2hiint(x);
    a = __builtin_amdgcn_permlane32_swap((unsigned)lo, (unsigned)lo, false, false);
    b = __builtin_amdgcn_permlane32_swap((unsigned)hi, (unsigned)hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]); // (0+1) + (2+3)
}
__device__ __forceinline__ double wave_reduce8(const double (&v)[8], const LaneSel &s)
{
    const double r0 = tstep<0xB1>(v[0], v[1], s.s1), r1 = tstep<0xB1>(v[2], v[3], s.s1);
    const double r2 = tstep<0xB1>(v[4], v[5], s.s1), r3 = tstep<0xB1>(v[6], v[7], s.s1);
    const double t0 = tstep<0x4E>(r0, r1, s.s2), t1 = tstep<0x4E>(r2, r3, s.s2);
    double u = tstep<0x141>(t0, t1, s.s3);
    u += dpp_xchg<0x140>(u);
    return rows_sum_f64(u);
}
__device__ __forceinline__ double wave_reduce4(const double (&v)[4], const LaneSel &s)
{
    const double r0 = tstep<0xB1>(v[0], v[1], s.s1), r1 = tstep<0xB1>(v[2], v[3], s.s1);
    double u = tstep<0x4E>(r0, r1, s.s2);
    u += dpp_xchg<0x141>(u);
    u += dpp_xchg<0x140>(u);
    return rows_sum_f64(u);
}
// Workgroup total of the value this lane holds after wave_reduce8 / 4: lanes 0..7 of every wave
// leave theirs in scratch[wave][value], and after the barrier every lane adds its
// value's column in wave order.  `scratch` ([nwaves][NV]) alternates between two buffers on consecutive calls.
template <int NV = 8>
__device__ __forceinline__ double block_total(double mine, int idx, double *scratch, int nwaves, int wave, int lane)
{
    if (nwaves == 1) return mine;
    if (lane < 8 && idx < NV) scratch[wave * NV + idx] = mine;
    __syncthreads();
    return lds_column(scratch + min(idx, NV - 1), NV, nwaves);
}

struct Rigid2 {
    double c, s, tx, ty;
};

// ICP.getTransform (icp.py:149-179) from reduced sums.  The reference takes the SVD of
// the 2x2 matrix W = BB^T.AA (:160-161), R = U.Vt with the reflection fix (:162-169);
// that R is exactly rot(atan2(W10 - W01, W00 + W11)) (checked against the
// SVD form in tests/test_oracle_golden.py::test_get_transform), evaluated here without
// trigonometry as (A, B)/hypot(A, B).
__device__ __forceinline__ Rigid2 kabsch_from_sums(double cax, double cay, double cbx, double cby, double w00,
                                                   double w01, double w10, double w11)
{
    double A = w00 + w11, B = w10 - w01;
    double h = sqrt(A * A + B * B);
    Rigid2 r;
    r.c = h == 0.0 ? 1.0 : A / h;           // W = 0: U.Vt of a zero matrix is the identity; NaN propagates
    r.s = h == 0.0 ? 0.0 : B / h;
    r.tx = cbx - (r.c * cax - r.s * cay);   // t = centroid_B - R.centroid_A (:172)
    r.ty = cby - (r.s * cax + r.c * cay);
    return r;
}

// The same with the two divisions done as ONE sequence (lane 0: A / h, lane 1: B / h) and broadcast:
// identical bits, a dozen instructions fewer per iteration.  Needs all 64 lanes active.
__device__ __forceinline__ Rigid2 kabsch_from_sums_wave(double cax, double cay, double cbx, double cby, double w00,
                                                        double w01, double w10, double w11, int lane)
{
    const double A = w00 + w11, B = w10 - w01;
    const double h = sqrt(A * A + B * B);
    const double qd = ((lane & 1) ? B : A) / h;
    Rigid2 r;
    r.c = h == 0.0 ? 1.0 : readlane_f64(qd, 0);
    r.s = h == 0.0 ? 0.0 : readlane_f64(qd, 1);
    r.tx = cbx - (r.c * cax - r.s * cay);
    r.ty = cby - (r.s * cax + r.c * cay);
    return r;
}

// ICP.findNearest (icp.py:90-114) for one query against the LDS-resident target cloud.
// Strict '<' keeps the lowest index on ties (:103); NaN never wins.  Squared distances
// are compared (sqrt is monotone); the distance itself is sqrt of the winner.
//
// Exact pruning.  The cloud is cut into blocks of kNNBlock consecutive points (beam order,
// hence spatially compact) whose bounding boxes sit in LDS.  A candidate index `seed` (the
// same beam index, or the previous iteration's match) gives an upper bound U on the
// answer.  Phase A tests every box against U (a straight-line loop of broadcast LDS reads)
// and leaves each LANE a bit mask of the blocks that could hold something closer than U;
// phase B lets every lane scan ITS OWN marked blocks, lowest first, with per-lane LDS
// addresses (on the benchmark scans a lane marks 1.2 blocks on average, the worst lane of a
// wave 2.4, while the union over the wave's 64 lanes is 5.2 - the earlier wave-uniform scan
// evaluated that union).  The box distance lb is evaluated with the same operation sequence
// as a point distance (sub, mul, fma) and IEEE rounding is monotone, so lb <= d2 holds for
// every point of the block as computed: an unmarked block holds only points strictly
// farther than U >= the final minimum, marked blocks are scanned in index order with strict
// '<', and the result (index and distance) is bit-identical to the exhaustive scan, ties
// included.
constexpr int kNNBlock = 16;
constexpr int kNNStride = kNNBlock + 1;   // LDS slots per block: the pad spreads the blocks over the banks
static_assert(kNNBlock == 16, "tslot() assumes 16-point blocks");

struct Box { double x0, x1, y0, y1; };

__device__ __forceinline__ int tslot(int j) { return j + (j >> 4); }   // LDS slot of target point j

__device__ __forceinline__ double dist2(double sx, double sy, double tx, double ty)
{
    double dx = sx - tx, dy = sy - ty;
    return fma(dy, dy, dx * dx);
}

// ---------------------------------------------------------------------------------
// The reference orders candidates by DISTANCE, sqrt of the square (icp.py:102-103), and sqrt maps up to
// three neighbouring doubles to one value: two candidates whose squares differ in the last places can be
// a TIE for it - the lower index wins - where the squares are ordered.  The searches below order by the
// square (no square root per candidate) and watch for the one event that can make the two orderings
// disagree: a candidate replacing a best that is less than 2^-50 above it (a class of equal roots spans
// at most 2^-51 of its value).  k_nn re-does the query of a lane that saw one the reference's way over the
// whole target (nn_exact); k_icp marks the PAIR (an LDS flag) and its workgroup then re-does it with
// nn_exact in every iteration.  Rare - mathematically tied neighbours of symmetric or quantised scans whose
// squares round differently - and exact; tests/golden/g10_sqrt_ties.npz holds replays whose iteration count
// depends on it.
// ---------------------------------------------------------------------------------
constexpr double kTieAbove = 1.0 + 0x1p-50;

struct Best {
    double d2;                // smallest square so far
    int j;
    unsigned long long ambm;  // lanes that saw the event, as a wave-wide mask (two scalar instructions per candidate: a
                              // per-lane bool costs a dozen vector ones in what the compiler makes of it; a per-lane
                              // counter fed by the compares as carries - two vector adds - was slower still: 10 000
                              // pairs 0.505 against 0.460 ms, and 0.395 ms without any bookkeeping; collecting the
                              // masks of a trip and combining them behind a scheduling barrier changed nothing)
    __device__ __forceinline__ void start() { d2 = INFINITY; j = 0; ambm = 0ull; }
    __device__ __forceinline__ void take(double d, int k)
    {
        const bool c = d < d2;
        const double dk = d * kTieAbove;                             // (off the compare chain: it depends on the candidate alone)
        ambm |= __ballot(c) & ~__ballot(dk < d2);
        d2 = fmin(d2, d);                                            // NaN never lowers it
        j = c ? k : j;
    }
    __device__ __forceinline__ bool amb() const { return (ambm >> (threadIdx.x & 63)) & 1ull; }
};

// the reference's own loop for one query (icp.py:99-105) over an LDS cloud (slot of point j: j + (j >> 4)
// when `padded`).  Its square root per candidate costs ~20 registers: it lives in k_nn and in the EXACT
// variant of k_icp, not in the hot one (there it took occupancy from 5 to 4 waves per SIMD, 10 000 pairs
// 0.39 -> 0.47 ms).
struct NNHit { double d2; int j; };
__device__ __forceinline__ NNHit nn_exact(const double2 *tab, bool padded, int n_tar, double sx, double sy)
{
    double ms = INFINITY, md2 = INFINITY;
    int mj = 0;
    for (int j = 0; j < n_tar; ++j) {
        const double2 t = tab[padded ? j + (j >> 4) : j];
        const double d2 = dist2(sx, sy, t.x, t.y);
        const double s = sqrt(d2);
        const bool c = s < ms;                                       // strict: the first of equal distances; NaN, inf never win
        ms = c ? s : ms;
        md2 = c ? d2 : md2;
        mj = c ? j : mj;
    }
    return NNHit{md2, mj};
}

__device__ __forceinline__ void nn_search(const double2 *__restrict__ tarL, const Box *__restrict__ boxes,
                                          const Box *__restrict__ boxes4, int nblocks, int n_tar, double sx, double sy,
                                          int seed, bool first_iter, bool active, double &best_d2, int &best_j, bool &amb)
{
    seed = min(max(seed, 0), n_tar - 1);
    double U;
    if (first_iter) {
        // no previous match yet: the same-index guess can be far off, so take the best of the
        // guess's whole block as the bound (16 evaluations that save several block scans)
        const double2 *t = tarL + (seed >> 4) * kNNStride;
        U = INFINITY;
#pragma unroll
        for (int k = 0; k < kNNBlock; ++k) U = fmin(U, dist2(sx, sy, t[k].x, t[k].y));   // fmin ignores NaN
    } else {
        double2 ts = tarL[tslot(seed)];
        U = dist2(sx, sy, ts.x, ts.y);
    }
    // (a block is marked if its box comes within the bound WIDENED by a class of equal distances: a point that
    // ties with the guess for the reference - its square a last place above the guess's - may sit exactly on the
    // edge of an otherwise farther box, and has to be seen for the tie to be noticed)
    double bound = (U == U) ? U * (1.0 + 0x1p-49) : INFINITY;       // a NaN seed distance bounds nothing
    if (!active) bound = -1.0;                   // padding lanes never ask for a block
    Best b;
    b.start();
    for (int base = 0; base < nblocks; base += 32) {
        const int cnt = min(32, nblocks - base);
        unsigned mask = 0u;
        for (int sb = 0; sb < cnt; sb += 4) {    // phase A, two levels: a box of 4 blocks first
            Box sx4 = boxes4[(base + sb) >> 2];
            double dx4 = fmax(fmax(sx4.x0 - sx, sx - sx4.x1), 0.0);
            double dy4 = fmax(fmax(sx4.y0 - sy, sy - sx4.y1), 0.0);
            if (!__any(fma(dy4, dy4, dx4 * dx4) <= bound)) continue;   // wave-uniform
#pragma unroll
            for (int u = 0; u < 4; ++u) {        // boxes[] is padded to a multiple of 4 with empty boxes
                Box bx = boxes[base + sb + u];
                double dx = fmax(fmax(bx.x0 - sx, sx - bx.x1), 0.0);
                double dy = fmax(fmax(bx.y0 - sy, sy - bx.y1), 0.0);
                double lb = fma(dy, dy, dx * dx);
                mask |= (lb <= bound) ? (1u << (sb + u)) : 0u;
            }
        }
        if (cnt < 32) mask &= (1u << cnt) - 1u;  // padding boxes are never scanned
        while (__any(mask != 0u)) {              // phase B: per-lane scan, lowest marked block first
            if (mask != 0u) {
                const int blk = base + __ffs((int)mask) - 1;
                mask &= mask - 1u;
                const double2 *t = tarL + blk * kNNStride;
                const double before = b.d2;
                int kk = 0;
#pragma unroll
                for (int k = 0; k < kNNBlock; ++k) {
                    double2 tk = t[k];
                    const double d = dist2(sx, sy, tk.x, tk.y);
                    const bool c = d < b.d2;
                    b.ambm |= __ballot(c) & ~__ballot(d * kTieAbove < b.d2);
                    b.d2 = fmin(b.d2, d);                            // NaN never lowers it
                    kk = c ? k : kk;                                 // (the index within the block: constants)
                }
                b.j = (b.d2 < before) ? blk * kNNBlock + kk : b.j;
            }
        }
    }
    best_d2 = b.d2;
    best_j = b.j;
    amb = b.amb();
}

// ---------------------------------------------------------------------------------
// Exact nearest neighbour in a cloud that IS a scan (nn_polar): target point k lies on the ray of
// beam k from the origin of the target frame, t_k = r_k (cos_t[k], sin_t[k]), r_k >= 0.  For a
// query s with |s| = rs and a candidate at distance sqrt(U) (the seed), a target closer than
// that must lie on a ray that passes within sqrt(U) of s: |angle(s) - beta_k| <= asin(sqrt(U) /
// rs).  With beams ordered by angle and spaced at least dbeta apart that is a window of beam
// INDICES around the seed's beam j: angle(s) = beta_j + delta with tan(delta) = cross(t_j, s) /
// dot(t_j, s), so k - j lies in [(delta - alpha) / dbeta, (delta + alpha) / dbeta] - typically 1-3
// beams instead of the 30-40 points + 15 boxes the box search touches (measured on the benchmark
// scans in the iterations after the first: median 2 candidates per query, mean 2.4, 90 % <= 4).  The window is evaluated in float32 with
// every rounding pushed outwards (plus the angular slack
// of points rounded to their storage type), the candidates inside it are compared exactly as
// everywhere else (float64 dist2, ascending index, strict '<'), and a ray outside it holds only
// points strictly farther than U: the result is bit-identical to the exhaustive scan.  The
// scan may wrap (beam n-1 next to beam 0: a full-circle lidar); wrapped candidates are always
// included, which is merely unnecessary for a scan that does not wrap.
// Queries whose window is wide (no good match: newly visible surfaces) or whose bound does not
// apply (closer to the origin than 2 sqrt(U), NaN) are left to the box search (`big`).
// ---------------------------------------------------------------------------------
constexpr int kPolarMax = 96;                 // widest window (beams) a lane searches itself before it asks the box search ...
constexpr int kPolarMaxFirst = 16;            // ... and in a first iteration whose wider ones are listed (nn_listed: 16 / 24 / 32 within 1 %),
constexpr int kPolarMaxListed = 48;           // where a window may be this wide after the re-guess
constexpr int kPolarProbe = 8;                // beams either side of a useless guess that are tried for a better one
constexpr bool kOnePass = true;               // iterations after the first: centroids and centred products in one reduction (k_icp)
constexpr int kPolarTail = 4;                 // NaN points behind the beam-window search's copy of the target

template <typename T> struct StoreSlack { static constexpr float ang = 2e-7f; };              // float64 points
template <> struct StoreSlack<float> { static constexpr float ang = 1e-6f; };                // 2 x 2^-24 sqrt(2), doubled
template <> struct StoreSlack<__half> { static constexpr float ang = 3e-3f; };               // 2 x 2^-11 sqrt(2), doubled

struct PolarGeo {
    float inv_db;     // 1 / (smallest angle between neighbouring beams), rounded up; 0: the cloud is not a usable scan
    float slack;      // angular slack of the stored points (radians)
};

template <int UNROLL, bool PROBE>
__device__ __forceinline__ void nn_polar(const double2 *__restrict__ tarP, int n_tar, double sx, double sy, int seed,
                                         bool active, const PolarGeo &geo, int wmax, double &best_d2, int &best_j, bool &big,
                                         bool &amb, unsigned long long &ambm)
{
    // (seed is a valid target index: the same beam clamped to the target's size before the first iteration, a match after it)
    const float fsx = (float)sx, fsy = (float)sy;
    const float rs2 = fsx * fsx + fsy * fsy;
    int lo, hi;
    // the window of beam indices around beam j that holds every target closer than t_j; false: no usable window
    auto window = [&](int j, int &wlo, int &whi) -> bool {
        const double2 ts = tarP[j];
        const double U = dist2(sx, sy, ts.x, ts.y);
        const float ftx = (float)ts.x, fty = (float)ts.y;
        // (v_rcp_f32 / v_sqrt_f32: one unit in the last place, 6e-8, against margins of 1e-6 and more; the IEEE
        // division the compiler makes of `/` or __fdividef here is ten instructions, twice per window)
        const float x2 = ((float)U * 1.000002f + 1e-30f) * __builtin_amdgcn_rcpf(rs2 * 0.999998f);   // (sqrt(U) / rs)^2, rounded up
        const bool small = x2 < 0.25f;                               // NaN: false
        const float x = __builtin_amdgcn_sqrtf(x2) * 1.000001f;
        const float alpha = x * (1.0f + 0.19f * x2) * 1.000002f + geo.slack;   // >= asin(x) for x < 0.5: (asin x - x) / x^3 grows from 1/6 to 0.1888 there
        const float y = (ftx * fsy - fty * fsx) * __builtin_amdgcn_rcpf(ftx * fsx + fty * fsy);       // tan(delta), |delta| <= 30 degrees
        const float y3 = y * y * y * 0.33333334f;
        const float dhi = (y >= 0.0f ? y : y - y3) + 4e-6f;          // y - y^3/3 <= atan(y) <= y for y >= 0 (mirrored below 0)
        const float dlo = (y >= 0.0f ? y - y3 : y) - 4e-6f;
        // With D = beta_k - beta_j in [delta - alpha, delta + alpha] and every gap between neighbouring beams at
        // least 1 / inv_db, a beam k > j has k - j <= D * inv_db, hence k <= j + floor(D * inv_db); every factor of
        // the computed product is rounded outwards by 1e-6 and more against 1e-7 of arithmetic rounding, so the
        // computed product is not below the true one and its floor not below the true floor.  Likewise below j
        // with ceil.  (Until late in round 2 there was one more beam either side "for safety", until round 4 the
        // roundings went the other way - ceil above, floor below: one beam either side of nearly every window,
        // 4.2 instead of 2.2 candidates per query on the benchmark replay.  Checked against the exhaustive search
        // by tests/test_polar_window_bound.py on the CPU and by every GPU parity test.)
        // (clamped to one turn either side: a window that wide is refused below, and the sums cannot wrap)
        const float fn = (float)n_tar;
        wlo = j + (int)ceilf(fmaxf(-fn, fminf(0.0f, (dlo - alpha) * geo.inv_db)));
        whi = j + (int)floorf(fminf(fn, fmaxf(0.0f, (dhi + alpha) * geo.inv_db)));
        return small && whi - wlo < wmax;
    };
    auto scan = [&](int a0, int a1, Best &b) {
        // UNROLL 4: four candidates per trip, their LDS reads in flight together.  A launch that cannot fill
        // the chip is bound by the LATENCY of a trip (read, distance, compare chain, wave-wide loop
        // test): 999 pairs alone 0.121 against 0.125 ms.  (With the padded copy and clamped indices the wider
        // trip cost a full chip 2.6 % - the windows are rounded to whole trips; reading the unpadded copy it
        // no longer does, and every launch shape up to three queries per lane takes four.)
        // (the candidates of a trip are read from the unpadded copy at p, p + 1, ...: no index arithmetic per
        // candidate.  A trip may run past a1: what lies there is a target outside the window - strictly
        // farther than the bound, it cannot win - or one of the NaN points behind the last beam.)
        // Every lane runs every trip of its wave - no lane mask inside the loop, so the tie bookkeeping stays in
        // scalar registers - but a lane past its own range reads the NaN points behind the last beam (one
        // address for all such lanes: a broadcast, where reading on through real targets cost LDS bandwidth).
        int trips = 0;                                               // (used by the diagnostic build only)
        if (UNROLL == 4) for (int k = a0; __any(k <= a1); k += 4) {
            ++trips;
            const int kc = k <= a1 ? k : n_tar;
            const double2 *t = tarP + kc;
            const double2 t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
            const double d0 = dist2(sx, sy, t0.x, t0.y), d1 = dist2(sx, sy, t1.x, t1.y);
            const double d2 = dist2(sx, sy, t2.x, t2.y), d3 = dist2(sx, sy, t3.x, t3.y);
            b.take(d0, kc);
            b.take(d1, kc + 1);
            b.take(d2, kc + 2);
            b.take(d3, kc + 3);
        }
        else for (int k = a0; __any(k <= a1); k += 2) {
            ++trips;
            const int kc = k <= a1 ? k : n_tar;
            const double2 *t = tarP + kc;
            const double2 t0 = t[0], t1 = t[1];
            const double d0 = dist2(sx, sy, t0.x, t0.y);
            const double d1 = dist2(sx, sy, t1.x, t1.y);
            b.take(d0, kc);
            b.take(d1, kc + 1);
        }
        ISTAMP_SCAN(wmax != kPolarMax, a1 >= a0 ? a1 - a0 + 1 : 0, trips, UNROLL);
        (void)trips;
    };
    bool fits = window(seed, lo, hi);
    // (PROBE: for the listed queries of a first iteration, nn_listed.  Until the end of round 3 the lanes of the
    // three-queries launch shape re-guessed for themselves, in every iteration; with the first iteration's
    // window-less queries listed, that shape is 16 % shorter without it when replays overlap: 0.183 against 0.218 ms.)
    if (PROBE && __any(active && !fits)) {
        // A guess that bounds nothing useful - before the first update source and target point of one beam
        // lie on ONE ray, and where the two scans see different surfaces there (14 % of the lanes, in 84 %
        // of the wave-queries) the bound is the range jump - is replaced by the best of the 17 beams around
        // it: the surface the query lies on is usually seen a few beams away (1.5 % of the lanes, 18 % of the
        // wave-queries remain for the box search).  Any index is a valid guess: it only supplies the bound.
        const bool need = active && !fits;
        Best pb;
        pb.start();
        pb.j = seed;
        scan(need ? max(seed - kPolarProbe, 0) : 1, need ? min(seed + kPolarProbe, n_tar - 1) : 0, pb);
        int lo1, hi1;
        const bool fits1 = window(pb.j, lo1, hi1) && need;
        lo = fits1 ? lo1 : lo;
        hi = fits1 ? hi1 : hi;
        fits = fits || fits1;
    }
    big = active && !fits;
    const bool go = active && !big;
    // three index ranges in ascending order: wrapped from above | the window | wrapped from below
    const int m0 = max(lo, 0), m1 = min(hi, n_tar - 1);
    // (the scan may close on itself with its last beam up to half a beam spacing PAST its first - polar_probe admits
    // that much: wrapped beam m lies at least n - 1 + m - 1/2 spacings above beam 0's index, so it belongs to the
    // window when m <= hi - (n - 1) + 1, and likewise below)
    // [0, e0] = [0, min(hi - (n - 1) + 1, m0 - 1)] when hi >= n - 2, and [s2, n - 1] = [max(n - 1 + lo - 1, m1 + 1), n - 1] when
    // lo <= 1; non-empty iff (hi >= n - 2 and lo >= 1) resp. (lo <= 1 and hi <= n - 2).  Rare: ONE wave-uniform test on lo
    // and hi, and the two ranges are formed only behind it (they were 10 vector instructions per query and iteration).
    Best b;
    b.start();
    const bool wraps = __any(go && ((hi >= n_tar - 2 && lo >= 1) || (lo <= 1 && hi <= n_tar - 2)));
    if (wraps) {
        const int e0 = hi >= n_tar - 2 ? min(hi - (n_tar - 1) + 1, m0 - 1) : -1;    // [0, e0]
        scan(go ? 0 : 1, go ? e0 : 0, b);
    }
    scan(go ? m0 : 1, go ? m1 : 0, b);
    if (wraps) {
        const int s2 = lo <= 1 ? max(n_tar - 1 + lo - 1, m1 + 1) : n_tar;           // [s2, n_tar - 1]
        scan(go ? s2 : 1, go ? n_tar - 1 : 0, b);
    }
    best_d2 = b.d2;
    best_j = b.j;
    amb = go && b.amb();
    ambm = b.ambm;                       // the same as a wave-wide mask (a lane that is not `go` compared nothing: its bit is never set)
}

// ---------------------------------------------------------------------------------
// nn_listed: the FIRST iteration's queries without a usable beam window, taken out of the lanes that own them.
// Before the first update a query and its guess lie on one ray, and where the two scans see different
// surfaces there the bound is the range jump: a window of dozens of beams for 9-14 % of the queries,
// scattered over 71 % of the wave-queries - and a wave whose lane re-guesses or takes the box search waits for
// it (a quarter of a 999-pair launch's duration).  k_icp lists such queries in LDS; here they are dealt out
// again, one per lane, to as few waves as hold them (the others wait at the barrier and leave the SIMDs to
// other pairs): each lane re-guesses among the beams around the first guess and searches the window of the
// better guess (nn_polar with PROBE).  What is still left - no surface near the ray at all, 0.4 % of the
// queries - is searched exhaustively, four queries at a time, by the rows of 16 lanes of the wave: lane r of a
// row looks at targets r, r + 16, ... in ascending order under Best's rule, then the 16 partial results are
// merged by (square, index): the smallest square, of equal squares the lowest index - what the reference's
// strict '<' over ascending indices keeps.  The tie bookkeeping carries over: a merge raises the flag when its
// winner replaces a lower-indexed candidate less than a class of equal roots above it (the event Best
// watches for), and flags are inherited from both sides; a candidate of lower index inside the winner's
// class always meets such a merge on its way up, whatever the order of the merges.
// qlist[e]: in (x, y) of the query, out (square, index | flag << 31 in the low word of .y); qseed[e]: its guess.
// ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_xchg_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }

template <int CTRL>
__device__ __forceinline__ void team_merge(double &d, int &jf)
{
    const double od = dpp_xchg<CTRL>(d);
    const int ojf = dpp_xchg_i<CTRL>(jf);
    const int j = jf & 0x7fffffff, oj = ojf & 0x7fffffff;
    const bool owin = od < d || (od == d && oj < j);
    const double dw = owin ? od : d, dl = owin ? d : od;
    const int jw = owin ? oj : j, jl = owin ? j : oj;
    const bool ev = dw < dl && jl < jw && !(dw * kTieAbove < dl);
    d = dw;
    jf = jw | ((jf | ojf) & (int)0x80000000) | (ev ? (int)0x80000000 : 0);
}

// one query against the whole target by the 16 lanes of a DPP row; every lane of the row returns the result
__device__ __forceinline__ void nn_row(const double2 *__restrict__ tarP, int n_tar, double2 qp, double &bd, int &jf)
{
    const int r = threadIdx.x & 15;
    bd = INFINITY;
    int bj = 0;
    bool f = false;
    for (int jb = r; jb < n_tar + r; jb += 64) {                     // a lane past the last target reads the NaN point behind it
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = jb + 16 * u;
            const double2 t = tarP[min(j, n_tar)];
            const double d = dist2(qp.x, qp.y, t.x, t.y);
            const bool c = d < bd;
            f |= c && !(d * kTieAbove < bd);
            bd = fmin(bd, d);                                        // NaN never lowers it
            bj = c ? j : bj;
        }
    }
    jf = bj | (f ? (int)0x80000000 : 0);
    team_merge<0xB1>(bd, jf);     // quad_perm [1,0,3,2]
    team_merge<0x4E>(bd, jf);     // quad_perm [2,3,0,1]
    team_merge<0x141>(bd, jf);    // row_half_mirror
    team_merge<0x140>(bd, jf);    // row_mirror
}

template <int UNROLL>
__device__ __forceinline__ void nn_listed(const double2 *__restrict__ tarP, int n_tar, const PolarGeo &geo, double2 *qlist,
                                          const int *qseed, int nq)
{
    const int lane = threadIdx.x & 63, row = lane >> 4;
    for (int base = (threadIdx.x >> 6) * 64; base < nq; base += blockDim.x) {     // wave-uniform
        const int e = base + lane;
        const bool act = e < nq;
        const double2 qp = qlist[act ? e : nq - 1];
        double d2;
        int j;
        bool big, amb;
        unsigned long long ambm;
        nn_polar<UNROLL, true>(tarP, n_tar, qp.x, qp.y, qseed[act ? e : nq - 1], act, geo, kPolarMaxListed, d2, j, big, amb, ambm);
        if (act && !big) qlist[e] = make_double2(d2, __hiloint2double(0, j | (amb ? (int)0x80000000 : 0)));
        unsigned long long left = __ballot(big);
        while (left != 0ull) {                                       // four of the remaining queries, one per row
            int src = 0, cnt = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (left != 0ull) {
                    const int l = __ffsll((long long)left) - 1;
                    left &= left - 1ull;
                    src = row == k ? l : src;
                    cnt = k + 1;
                }
            }
            const bool valid = row < cnt;
            double bd;
            int jf;
            nn_row(tarP, n_tar, qlist[base + src], bd, jf);           // (rows without a query repeat lane 0's entry)
            if (valid && (lane & 15) == 0) qlist[base + src] = make_double2(bd, __hiloint2double(0, jf));
        }
    }
}

// LDS image of the target: float64 (x, y) pairs padded with NaN points to a whole number
// of blocks (NaN never wins a comparison), then one bounding box per block.
__host__ __device__ inline int nn_blocks(int n_tar) { return (n_tar + kNNBlock - 1) / kNNBlock; }
__host__ __device__ inline int nn_boxes_padded(int n_tar) { return (nn_blocks(n_tar) + 3) / 4 * 4; }   // multiple of 4
__host__ __device__ inline int nn_boxes4(int n_tar) { return nn_boxes_padded(n_tar) / 4; }
__host__ __device__ inline size_t nn_lds_bytes(int n_tar)
{
    return (size_t)nn_blocks(n_tar) * kNNStride * sizeof(double2) +
           (size_t)(nn_boxes_padded(n_tar) + nn_boxes4(n_tar)) * sizeof(Box);
}

// A cloud read either from a point buffer or straight from a raw scan (laserToNumpy fused:
// slam_ekf.py:115-123, x = cos(angle_i) * r with inf -> 30 m, rounded to the buffer type T).
template <typename T>
struct Cloud {
    const T *pts;              // [2][n] or null
    const float *ranges;       // [n] or null
    const double *cos_t, *sin_t;
    int n;
    __device__ __forceinline__ double2 at(int j) const
    {
        if (ranges) {
            double r = (double)ranges[j];
            if (r == INFINITY) r = 30.0;
            return make_double2(round_as(cos_t[j] * r, (const T *)nullptr), round_as(sin_t[j] * r, (const T *)nullptr));
        }
        return make_double2(ld(pts, j), ld(pts, (long)n + j));
    }
};

// Is the target a usable scan?  Beam directions must be unit vectors, ordered by angle with
// neighbours less than 30 degrees apart, spanning at most one turn (plus half a beam), and no
// range may be negative.  One workgroup-wide pass over the trig tables per pair (they are shared
// by all pairs of a launch and sit in L2).  Every wave leaves its part in slots[wave]: [0] smallest
// cross product of neighbouring directions as float bits (a lower bound of their angle: asin(x) >= x),
// [1] upper bound of the span it saw (float bits), [2] ok flag; polar_combine() puts them together behind
// a barrier.  (No atomics, no initialisation to order against: the probe runs ahead of the pair's first
// barrier, its loads in flight together with those that stage the target.)
template <typename T>
__device__ __forceinline__ void polar_probe(const Cloud<T> &tar, int n_tar, unsigned *slots)
{
    // per-lane partials, one butterfly per wave
    unsigned mn = 0x7f800000u;
    float sum = 0.0f;
    bool bad = false;
    for (int j = threadIdx.x; j < n_tar; j += blockDim.x) {
        const double c0 = tar.cos_t[j], s0 = tar.sin_t[j];
        bool ok = fabs(c0 * c0 + s0 * s0 - 1.0) < 1e-6 && !(tar.ranges[j] < 0.0f);
        if (j + 1 < n_tar) {
            const double c1 = tar.cos_t[j + 1], s1 = tar.sin_t[j + 1];
            const double cr = c0 * s1 - s0 * c1, dt = c0 * c1 + s0 * s1;
            ok = ok && cr > 0.0 && cr <= 0.5 && dt > 0.0;
            if (ok) {
                mn = min(mn, __float_as_uint((float)cr * 0.999999f));
                sum += (float)(cr * (1.0 + cr * cr * (1.0 / 6.0 + 0.1 * cr * cr))) * 1.000001f;
            }
        }
        bad |= !ok;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = min(mn, (unsigned)__shfl_xor((int)mn, off));
        sum += __shfl_xor(sum, off);
    }
    const bool wave_bad = __any(bad);
    if ((threadIdx.x & 63) == 0) {
        unsigned *s = slots + 4 * (threadIdx.x >> 6);
        s[0] = mn;
        s[1] = __float_as_uint(sum * 1.00001f);
        s[2] = wave_bad ? 0u : 1u;
    }
}

struct PolarProbe { float dmin, span; bool ok; };
__device__ __forceinline__ PolarProbe polar_combine(const unsigned *slots, int nwaves)
{
    const int lane = threadIdx.x & 63;
    unsigned mn = 0x7f800000u, okw = 1u;
    float sum = 0.0f;
    if (lane < nwaves) { mn = slots[4 * lane]; sum = __uint_as_float(slots[4 * lane + 1]); okw = slots[4 * lane + 2]; }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {                          // (at most kMaxWaves = 16 slots)
        mn = min(mn, (unsigned)__shfl_xor((int)mn, off));
        sum += __shfl_xor(sum, off);
        okw &= (unsigned)__shfl_xor((int)okw, off);
    }
    PolarProbe r;
    r.dmin = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)mn));
    r.span = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(sum * 1.00001f)));
    r.ok = __builtin_amdgcn_readfirstlane((int)okw) != 0;
    return r;
}

template <typename T>
__device__ __forceinline__ void stage_points(const Cloud<T> &tar, int n_tar, double2 *tarL, double2 *tarP = nullptr)
{
    const int nb = nn_blocks(n_tar), npad = nb * kNNBlock;
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int j = threadIdx.x; j < npad; j += blockDim.x) {
        const double2 t = j < n_tar ? tar.at(j) : make_double2(qnan, qnan);
        tarL[tslot(j)] = t;
        if (tarP && j < n_tar) tarP[j] = t;
    }
    if (tarP && threadIdx.x < kPolarTail) tarP[n_tar + threadIdx.x] = make_double2(qnan, qnan);
}

__device__ __forceinline__ void stage_boxes(int n_tar, const double2 *tarL, Box *boxes, Box *boxes4)
{
    const int nb = nn_blocks(n_tar);
    for (int b = threadIdx.x; b < nn_boxes_padded(n_tar); b += blockDim.x) {
        Box bx{INFINITY, -INFINITY, INFINITY, -INFINITY};            // stays empty for the padding boxes
        if (b < nb) {
#pragma unroll
            for (int k = 0; k < kNNBlock; ++k) {
                double2 t = tarL[b * kNNStride + k];
                bx.x0 = fmin(bx.x0, t.x); bx.x1 = fmax(bx.x1, t.x);  // fmin / fmax ignore NaN
                bx.y0 = fmin(bx.y0, t.y); bx.y1 = fmax(bx.y1, t.y);
            }
        }
        boxes[b] = bx;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nn_boxes4(n_tar); b += blockDim.x) {
        Box bx = boxes[4 * b];
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            Box o = boxes[4 * b + u];
            bx.x0 = fmin(bx.x0, o.x0); bx.x1 = fmax(bx.x1, o.x1);
            bx.y0 = fmin(bx.y0, o.y0); bx.y1 = fmax(bx.y1, o.y1);
        }
        boxes4[b] = bx;
    }
}

template <typename T>
__device__ __forceinline__ void stage_target(const Cloud<T> &tar, int n_tar, double2 *tarL, Box *boxes, Box *boxes4)
{
    stage_points(tar, n_tar, tarL);
    __syncthreads();
    stage_boxes(n_tar, tarL, boxes, boxes4);
}

// ---------------------------------------------------------------------------------
// k_icp: ICP.process (icp.py:38-88), one workgroup per pair, QPT queries per lane.
// ---------------------------------------------------------------------------------
constexpr int kIcpExtraLds = 32;               // flag words: source set collapsed, count of listed queries, re-do
__host__ __device__ inline size_t icp_polar_bytes(int n_tar) { return (size_t)(n_tar + kPolarTail) * sizeof(double2); }
// cross-wave stage of the reductions ([2][nwaves][8] doubles: seven values in the one-pass iteration; 128 B a wave, so
// what follows stays 16-byte aligned) and of the collapsed-set test ([2][nwaves][4]: matched point of the wave's first
// query, "this wave saw another"), two alternating buffers each
constexpr int kRedStride = 8;
__host__ __device__ inline size_t icp_red_bytes(int nwaves) { return (size_t)2 * nwaves * (kRedStride + 4) * sizeof(double) + 8 * sizeof(double); }   // + what a pair's first wave hands the others (kLead)

// EXACT: the second pass over a pair in which the first saw a best undercut its predecessor by less than a class of
// equal distances (see "Best"): the same solve with the reference's own nearest-neighbour loop (nn_exact).
// Returns whether the pair needs that second pass (the same value in every lane; false from the second pass).
template <typename T, int QPT, int UNROLL, bool EXACT>
__device__ __forceinline__ bool icp_pair(const IcpArgs &a, const int b, char *smem)
{
    ISTAMP_DECL;
    const int nblocks = nn_blocks(a.n_tar);
    double2 *tarL = reinterpret_cast<double2 *>(smem);                                           // [nblocks * kNNStride]
    Box *boxes = reinterpret_cast<Box *>(smem + (size_t)nblocks * kNNStride * sizeof(double2));  // [padded to x4]
    Box *boxes4 = boxes + nn_boxes_padded(a.n_tar);                                              // one per 4 blocks
    // the target again, not padded, for the beam-window search (nn_polar): carved when the target is a scan and
    // both copies fit the CU's LDS (launch_icp_t); without it the padded copy serves every purpose
    const bool has_p = a.polar_copy != 0;
    const size_t p_bytes = has_p ? icp_polar_bytes(a.n_tar) : 0;
    double2 *tarP = has_p ? reinterpret_cast<double2 *>(smem + nn_lds_bytes(a.n_tar)) : nullptr;   // [n_tar + kPolarTail]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    double *red = reinterpret_cast<double *>(smem + nn_lds_bytes(a.n_tar) + p_bytes);             // [2][nwaves][kRedStride]
    double *cref = red + 2 * nwaves * kRedStride;                                                // [2][nwaves][4]
    double *lead = cref + 2 * nwaves * 4;                                                        // [8]: rotation, translation, centroid, mean error from the pair's first wave (kLead)
    unsigned *geo = reinterpret_cast<unsigned *>(lead + 8);                          // [8] flags: [3] source set collapsed, [4] queries listed for nn_listed, [5] re-do (the others unused)
    double2 *qlist = reinterpret_cast<double2 *>(geo + 8);                                      // [a.team_cap] nn_listed
    int *qseed = reinterpret_cast<int *>(qlist + a.team_cap);                                    // [a.team_cap]
    char *guard = reinterpret_cast<char *>(qseed + a.team_cap);
    lds_guard_fill(guard);

    const long be = (long)b + (a.ppt ? b / a.ppt : 0);
    const int n_src = a.n_src, n_tar = a.n_tar;
    Cloud<T> tar{nullptr, nullptr, a.cos_t, a.sin_t, n_tar}, src{nullptr, nullptr, a.cos_t, a.sin_t, n_src};
    if (a.ranges) {
        tar.ranges = a.ranges + be * a.tar_scan_stride;
        src.ranges = a.ranges + be * a.src_scan_stride + n_tar;      // the scan after the target's
    } else {
        tar.pts = static_cast<const T *>(a.tar) + be * a.tar_stride;
        src.pts = static_cast<const T *>(a.src) + be * a.src_stride;
    }

    if (threadIdx.x == 0) { geo[3] = 1u; geo[4] = 0u; geo[5] = 0u; }

    // the source points first: their loads are in flight together with those that stage the target
    double sx[QPT], sy[QPT], ax[QPT], ay[QPT];
    int seed[QPT];
    bool ok[QPT];
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        int i = tid + q * blockDim.x;
        ok[q] = i < n_src;
        double2 pt = ok[q] ? src.at(i) : make_double2(0.0, 0.0);
        double x = pt.x, y = pt.y;
        if (a.prior) {
            const double *p = a.prior + 6 * (long)b;
            double xp = p[0] * x + p[1] * y + p[2];
            double yp = p[3] * x + p[4] * y + p[5];
            x = xp; y = yp;
        }
        sx[q] = ax[q] = x;
        sy[q] = ay[q] = y;
        seed[q] = min(i, n_tar - 1);             // first guess: the same beam index
    }
    bool src_differs = false;
    {
        // Collapsed sets (every point of a set is ONE point): W = BB^T.AA is mathematically zero and
        // the canonical answer is R = I (the SVD of a zero matrix), t = centroid_B - centroid_A.  The
        // reference's centred rows are rounding noise there (np.mean of n equal values is not that
        // value) and its rotation arbitrary: documented deviation, tests/golden/g8_collapsed.npz.
        double2 p0 = src.at(0);
        if (a.prior) {
            const double *p = a.prior + 6 * (long)b;
            p0 = make_double2(p[0] * p0.x + p[1] * p0.y + p[2], p[3] * p0.x + p[4] * p0.y + p[5]);
        }
#pragma unroll
        for (int q = 0; q < QPT; ++q) src_differs |= ok[q] && !(ax[q] == p0.x && ay[q] == p0.y);
    }
    ISTAMP(13);
    stage_points(tar, n_tar, tarL, tarP);
    const bool probed = a.ranges && has_p;
    unsigned *pslots = reinterpret_cast<unsigned *>(cref);           // [nwaves][4]: the iterations use this space later
    if (probed) polar_probe(tar, n_tar, pslots);
    __syncthreads();                                                 // (geo is initialised)
    ISTAMP(10);
    if (src_differs) geo[3] = 0u;
    // the target is a scan with usable beam geometry: nearest neighbours by beam window (nn_polar)
    PolarGeo pg;
    {
        PolarProbe pp{1.0f, 0.0f, false};
        if (probed) pp = polar_combine(pslots, nwaves);
        const float dmin = pp.dmin, span = pp.span;
        // (beams closer together than 1e-5 rad are no usable geometry: the window's index bound (dhi + alpha) * inv_db
        // must stay far inside the int range)
        const bool polar = probed && pp.ok && n_tar >= 2 && dmin >= 1e-5f && dmin < 1.0f && span <= 6.2831855f + 0.5f * dmin;
        pg.inv_db = polar ? __fdividef(1.000002f, dmin) : 0.0f;
        pg.slack = StoreSlack<T>::ang;
    }
    ISTAMP(12);
    stage_boxes(n_tar, tarL, boxes, boxes4);
    ISTAMP(11);
    __syncthreads();
    const bool src_collapsed = geo[3] != 0u;

    const double dn = (double)n_src;
    const LaneSel ls = lane_sel(lane);
    ISTAMP(14);
    double pre_error = 0.0, mean_error = 0.0, pcx = 0.0, pcy = 0.0, ca0x = 0.0, ca0y = 0.0;   // (ca0: centroid of the source before it moves)
    int iters = 0, par = 0;
    unsigned long long amb_mask = 0ull;   // lanes of this wave that saw a best undercut its predecessor by less than a class of equal distances:
                                          // kept as a wave-wide mask in scalar registers (a per-lane flag cost four vector instructions a query)
    // one iteration; FIRST: the instance for iteration 0 (two reductions as the reference; the only one that lists queries
    // for nn_listed), the loop behind it takes the one-pass form; returns true when the solve has converged (icp.py:76-77)
    auto iterate = [&](auto first_tag, const int it) __attribute__((always_inline)) -> bool {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool kLead = QPT >= 3 && !EXACT;                   // launch shapes for a full chip: see `finish` below
        double mx[QPT], my[QPT], dq[QPT];
        // first iteration over a scan: queries without a usable beam window are listed (nn_listed)
        const bool team_it = FIRST && !EXACT && a.team_cap > 0 && pg.inv_db > 0.0f;
        int slot[QPT];
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            double d2; int j;
            slot[q] = -1;
            if (EXACT) {
                const NNHit h = nn_exact(has_p ? tarP : tarL, !has_p, ok[q] ? n_tar : 0, sx[q], sy[q]);
                d2 = h.d2; j = h.j;
            } else if (pg.inv_db > 0.0f) {                           // wave-uniform: the target is a scan
                bool big, amb_lane;
                unsigned long long am;
                nn_polar<UNROLL, false>(tarP, n_tar, sx[q], sy[q], seed[q], ok[q], pg, team_it ? kPolarMaxFirst : kPolarMax, d2, j,
                                        big, amb_lane, am);          // icp.py:67
                (void)amb_lane;
                amb_mask |= am;
                if (FIRST && team_it) {
                    const unsigned long long bm = __ballot(big);
                    if (bm != 0ull) {
                        int base = 0;
                        if (lane == 0) base = (int)atomicAdd(&geo[4], (unsigned)__popcll(bm));
                        base = __builtin_amdgcn_readfirstlane(base);
                        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
                        if (big && pos < a.team_cap) {
                            slot[q] = pos;
                            qlist[pos] = make_double2(sx[q], sy[q]);
                            qseed[pos] = seed[q];
                            big = false;                             // (a list that is full leaves the rest to the box search)
                        }
                    }
                }
                // The few queries without a good match (newly visible surfaces; they come in runs of
                // neighbouring beams: measured 1.3 % of the queries, in 10 % of the wave-queries, 8 lanes at
                // a time) take the box search.  Tried and dropped: scanning the whole cloud for them
                // with the wave, one query after the other (4.37e7 instead of 4.00e7 instructions per
                // launch), and searching a lane's queries together in one wave-wide loop (4.55e7).
                ISTAMP_BIG(it, big);
                if (__any(big)) {
                    double d2b; int jb; bool ambb;
                    nn_search(tarL, boxes, boxes4, nblocks, n_tar, sx[q], sy[q], seed[q], FIRST, big, d2b, jb, ambb);
                    d2 = big ? d2b : d2;
                    j = big ? jb : j;
                    amb_mask |= __ballot(big && ambb);
                }
            } else {
                bool ambs;
                nn_search(tarL, boxes, boxes4, nblocks, n_tar, sx[q], sy[q], seed[q], FIRST, ok[q], d2, j, ambs);   // icp.py:67
                amb_mask |= __ballot(ambs);
            }                                                        // (amb_mask != 0 -> the pair is re-done: icp_pair<EXACT>)
            seed[q] = j;                                             // next iteration's guess
            double2 m = has_p ? tarP[j] : tarL[tslot(j)];
            mx[q] = m.x; my[q] = m.y;
            dq[q] = (d2 < INFINITY) ? sqrt(d2) : 0.0;               // never-won query: distance 0 (:97)
        }
        if (FIRST && team_it) {
            __syncthreads();
            nn_listed<UNROLL>(tarP, n_tar, pg, qlist, qseed, min((int)geo[4], a.team_cap));
            __syncthreads();
#pragma unroll
            for (int q = 0; q < QPT; ++q) {
                bool flagged = false;
                if (slot[q] >= 0) {
                    const double2 rs = qlist[slot[q]];
                    const int jf = __double2loint(rs.y), j = jf & 0x7fffffff;
                    flagged = jf < 0;
                    seed[q] = j;
                    const double2 m = tarP[j];
                    mx[q] = m.x; my[q] = m.y;
                    dq[q] = (rs.x < INFINITY) ? sqrt(rs.x) : 0.0;
                }
                amb_mask |= __ballot(flagged);
            }
        }
        // every source point matched to ONE target point (same coordinates)?  see "collapsed sets" above.  Every wave
        // compares its matches with that of its first query (which always exists) and leaves point and verdict in LDS
        // ahead of the iteration's first barrier; behind it the waves' points are compared with one another.
        par ^= 1;
        double *cr = cref + par * nwaves * 4;
        bool wave_differs;
        {
            const double m0x = readlane_f64(mx[0], 0), m0y = readlane_f64(my[0], 0);
            // The usual case is settled on ONE lane: some first query of the wave is matched to another beam than lane 0's
            // (an integer compare per lane), and that beam's point differs from lane 0's.  Only when no such lane exists, or
            // its point has the same coordinates (targets that coincide: several beams of range 0), every match is compared.
            const int j0 = __builtin_amdgcn_readfirstlane(seed[0]);
            const unsigned long long other = __ballot(ok[0] && seed[0] != j0);
            bool found = false;
            if (other != 0ull) {
                const int l = __ffsll((long long)other) - 1;
                const double ox = readlane_f64(mx[0], l), oy = readlane_f64(my[0], l);
                found = !(ox == m0x && oy == m0y);
            }
            if (found) wave_differs = true;
            else {
                bool differs = false;
#pragma unroll
                for (int q = 0; q < QPT; ++q) differs |= ok[q] && !(mx[q] == m0x && my[q] == m0y);
                wave_differs = __any(differs);
            }
            if (nwaves > 1 && lane == 0) { cr[4 * wave] = m0x; cr[4 * wave + 1] = m0y; cr[4 * wave + 2] = wave_differs ? 1.0 : 0.0; }
        }
        auto targets_collapsed = [&]() -> bool {                     // (call behind the barrier)
            if (nwaves == 1) return !wave_differs;
            const double *mine = cr + 4 * min(lane, nwaves - 1);
            return !__any(mine[2] != 0.0 || !(mine[0] == cr[0] && mine[1] == cr[1]));
        };
        double cax, cay, cbx, cby, w[4];
        Rigid2 r;
        if (FIRST || !kOnePass) {
            // Two passes, as the reference (centroids, then centred products, icp.py:154-160).
            double v[5] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < QPT; ++q)
                if (ok[q]) { v[0] += sx[q]; v[1] += sy[q]; v[2] += mx[q]; v[3] += my[q]; v[4] += dq[q]; }
            ISTAMP(FIRST ? 6 : 1);
            // sums of the five quantities over the pair (transposed reduction: lane l ends up with the total
            // of value idx8(l)), ONE division sequence for all of them, then five broadcasts
            {
                const double v8[8] = {v[0], v[1], v[2], v[3], v[4], 0.0, 0.0, 0.0};
                const double tot = block_total(wave_reduce8(v8, ls), ls.idx8, red + par * nwaves * kRedStride, nwaves, wave, lane);
                const double qv = tot / dn;                          // icp.py:154-155, :75
                cax = readlane_f64(qv, 0); cay = readlane_f64(qv, 1); cbx = readlane_f64(qv, 2); cby = readlane_f64(qv, 3);
                mean_error = readlane_f64(qv, 7);                    // value 4 lives in lane 7
            }
            ISTAMP(FIRST ? 7 : 2);
            const bool tar_collapsed = targets_collapsed();
            w[0] = w[1] = w[2] = w[3] = 0.0;
#pragma unroll
            for (int q = 0; q < QPT; ++q) {
                if (ok[q]) {
                    double aax = sx[q] - cax, aay = sy[q] - cay, bbx = mx[q] - cbx, bby = my[q] - cby;
                    w[0] += bbx * aax; w[1] += bbx * aay; w[2] += bby * aax; w[3] += bby * aay;   // :160
                }
            }
            par ^= 1;
            {
                const double tot = block_total(wave_reduce4(w, ls), ls.idx4, red + par * nwaves * kRedStride, nwaves, wave, lane);
                w[0] = readlane_f64(tot, 0); w[1] = readlane_f64(tot, 1); w[2] = readlane_f64(tot, 2); w[3] = readlane_f64(tot, 3);
            }
            ISTAMP(FIRST ? 8 : 3);
            if (tar_collapsed || src_collapsed) w[0] = w[1] = w[2] = w[3] = 0.0;
            if (FIRST) { ca0x = cax; ca0y = cay; }
        } else {
            // From the second iteration on: ONE pass and one barrier.  The update of the iteration before moved the
            // source's centroid onto the centroid of its matches (t = c_B - R c_A), so that point p is within rounding
            // of this iteration's source centroid and close to the new matches': sums and products are formed about p
            // and the exact centroids and centred products follow algebraically,
            //   c_A = p + S_a / N,  c_B = p + S_b / N,  W = sum (b - p)(a - p)^T - S_b S_a^T / N,
            // with S_a = sum (a - p) at rounding level - the correction is of the order of the last place of W.
            // The rotation needs W only through A = W00 + W11 and B = W10 - W01 (kabsch_from_sums): the two are summed
            // directly - seven values instead of nine, one exchange level less in the wave reduction.
            double u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < QPT; ++q) {
                if (ok[q]) {
                    const double dax = sx[q] - pcx, day = sy[q] - pcy, dbx = mx[q] - pcx, dby = my[q] - pcy;
                    u[0] += dax; u[1] += day; u[2] += dbx; u[3] += dby; u[4] += dq[q];
                    // (fused: two instructions a sum instead of four - these sums are this kernel's own form of W anyway, K2)
                    u[5] = fma(dbx, dax, fma(dby, day, u[5])); u[6] = fma(dby, dax, fma(-dbx, day, u[6]));
                }
            }
            ISTAMP(1);
            // what follows the sums: the quotients, the rotation, the translation - a hundred instructions on wave-uniform
            // values.  In the launch shapes for a full chip (kLead: three queries per lane and more) the pair's first wave
            // does them alone and leaves the result in LDS behind a second barrier - the waves that wait leave their issue
            // slots to other pairs (10 000 pairs: 0.388 -> 0.381 ms); a launch that cannot fill the chip is bound by a pair's
            // own latency, and there every wave computes them for itself behind the ONE barrier.
            auto finish = [&](const double tot) __attribute__((always_inline)) {
                const double qv = tot / dn;                          // icp.py:154-155, :75
                const double qax = readlane_f64(qv, 0), qay = readlane_f64(qv, 1);
                cax = pcx + qax; cay = pcy + qay;
                cbx = pcx + readlane_f64(qv, 2); cby = pcy + readlane_f64(qv, 3);
                mean_error = readlane_f64(qv, 7);                    // value 4 lives in lane 7, values 5, 6 in lanes 6, 5
                const double sbx = readlane_f64(tot, 2), sby = readlane_f64(tot, 3);
                w[0] = readlane_f64(tot, 6) - (sbx * qax + sby * qay);   // A
                w[2] = readlane_f64(tot, 5) - (sby * qax - sbx * qay);   // B
                w[1] = w[3] = 0.0;
                ISTAMP(3);
                if (targets_collapsed() || src_collapsed) w[0] = w[2] = 0.0;
                r = kabsch_from_sums_wave(cax, cay, cbx, cby, w[0], w[1], w[2], w[3], lane);    // :69
            };
            if (kLead && nwaves > 1) {
                double *sc = red + par * nwaves * kRedStride;
                const double mine = wave_reduce8(u, ls);
                if (lane < 8) sc[wave * 8 + ls.idx8] = mine;
                __syncthreads();
                if (__builtin_amdgcn_readfirstlane(wave) == 0) {
                    finish(lds_column(sc + ls.idx8, 8, nwaves));
                    if (lane == 0) {
                        double2 *l2 = reinterpret_cast<double2 *>(lead);
                        l2[0] = make_double2(r.c, r.s); l2[1] = make_double2(r.tx, r.ty);
                        l2[2] = make_double2(cbx, cby); l2[3] = make_double2(mean_error, 0.0);
                    }
                }
                __syncthreads();
                const double2 *l2 = reinterpret_cast<const double2 *>(lead);
                const double2 v0 = l2[0], v1 = l2[1], v2 = l2[2], v3 = l2[3];
                r.c = v0.x; r.s = v0.y; r.tx = v1.x; r.ty = v1.y;
                cbx = readlane_f64(v2.x, 0); cby = readlane_f64(v2.y, 0);
                mean_error = readlane_f64(v3.x, 0);
            } else {
                finish(block_total<8>(wave_reduce8(u, ls), ls.idx8, red + par * nwaves * kRedStride, nwaves, wave, lane));
            }
        }
        if (FIRST || !kOnePass) r = kabsch_from_sums_wave(cax, cay, cbx, cby, w[0], w[1], w[2], w[3], lane);    // :69
#pragma unroll
        for (int q = 0; q < QPT; ++q) {                              // src = T.src (:71)
            double nx = r.c * sx[q] + (-r.s) * sy[q] + r.tx;
            double ny = r.s * sx[q] + r.c * sy[q] + r.ty;
            sx[q] = nx; sy[q] = ny;
        }
        pcx = cbx; pcy = cby;                                        // the centroid the source now has (up to rounding)
        ++iters;
        ISTAMP(FIRST ? 9 : 4);
        if (fabs(pre_error - mean_error) < a.tol) return true;       // :76-77
        pre_error = mean_error;
        return false;
    };
    if (a.max_iter > 0 && !iterate(std::true_type{}, 0))
        for (int it = 1; it < a.max_iter; ++it)
            if (iterate(std::false_type{}, it)) break;

    // (a pair in which some lane saw a best undercut its predecessor by less than a class of equal distances is
    // re-done: the flag travels through LDS, behind the barriers of the sums below)
    if (!EXACT && amb_mask != 0ull) geo[5] = 1u;
    // final T = getTransform(A_original, src_final) (icp.py:81).  ONE reduction when an iteration has run: the originals'
    // centroid c_A is the first iteration's source centroid (the same sums in the same order - the source had not moved yet),
    // and the last update put the source's centroid on p = (pcx, pcy) up to rounding, so with S = sum (s - p)
    //   c_S = p + S / N,   W = sum (s - p)(a - c_A)^T   (- S . sum (a - c_A)^T / N: rounding level times rounding level),
    // of which the rotation needs A = W00 + W11 and B = W10 - W01 (as in the iterations).  Only the pair's first wave goes on
    // behind the barrier: it alone stores the result.
    Rigid2 r;
    if (iters > 0) {
        double u[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            if (ok[q]) {
                const double dsx = sx[q] - pcx, dsy = sy[q] - pcy, dax = ax[q] - ca0x, day = ay[q] - ca0y;
                u[0] += dsx; u[1] += dsy;
                u[2] += dsx * dax + dsy * day; u[3] += dsy * dax - dsx * day;
            }
        }
        par ^= 1;
        const double mine = wave_reduce4(u, ls);
        double tot = mine;
        bool mine_to_finish = true;
        if (nwaves > 1) {
            double *sc = red + par * nwaves * kRedStride;
            if (lane < 8 && ls.idx4 < 8) sc[wave * 8 + ls.idx4] = mine;
            __syncthreads();
            mine_to_finish = __builtin_amdgcn_readfirstlane(wave) == 0;
            if (mine_to_finish) tot = lds_column(sc + ls.idx4, 8, nwaves);
        }
        if (mine_to_finish) {
            const double qv = tot / dn;
            const double csx = pcx + readlane_f64(qv, 0), csy = pcy + readlane_f64(qv, 1);
            double wa = readlane_f64(tot, 2), wb = readlane_f64(tot, 3);
            if (src_collapsed) wa = wb = 0.0;
            r = kabsch_from_sums_wave(ca0x, ca0y, csx, csy, wa, 0.0, wb, 0.0, lane);
        }
    } else {
        // no iteration (max_iter 0): centroids, then centred products, as the reference
        double v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < QPT; ++q)
            if (ok[q]) { v[0] += ax[q]; v[1] += ay[q]; v[2] += sx[q]; v[3] += sy[q]; }
        par ^= 1;
        double cax, cay, cbx, cby;
        {
            const double tot = block_total(wave_reduce4(v, ls), ls.idx4, red + par * nwaves * kRedStride, nwaves, wave, lane);
            const double qv = tot / dn;
            cax = readlane_f64(qv, 0); cay = readlane_f64(qv, 1); cbx = readlane_f64(qv, 2); cby = readlane_f64(qv, 3);
        }
        double w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < QPT; ++q) {
            if (ok[q]) {
                double aax = ax[q] - cax, aay = ay[q] - cay, bbx = sx[q] - cbx, bby = sy[q] - cby;
                w[0] += bbx * aax; w[1] += bbx * aay; w[2] += bby * aax; w[3] += bby * aay;
            }
        }
        par ^= 1;
        {
            const double tot = block_total(wave_reduce4(w, ls), ls.idx4, red + par * nwaves * kRedStride, nwaves, wave, lane);
            w[0] = readlane_f64(tot, 0); w[1] = readlane_f64(tot, 1); w[2] = readlane_f64(tot, 2); w[3] = readlane_f64(tot, 3);
        }
        if (src_collapsed) w[0] = w[1] = w[2] = w[3] = 0.0;
        r = kabsch_from_sums_wave(cax, cay, cbx, cby, w[0], w[1], w[2], w[3], lane);
    }
    if (tid == 0) {
        double *To = a.T_out + 9 * (long)b;
        To[0] = r.c; To[1] = -r.s; To[2] = r.tx;
        To[3] = r.s; To[4] = r.c;  To[5] = r.ty;
        To[6] = 0.0; To[7] = 0.0;  To[8] = 1.0;
        if (a.iters_out) a.iters_out[b] = iters;
        if (a.err_out) a.err_out[b] = mean_error;
    }
    ISTAMP(5);
    ISTAMP_END(iters);
    lds_guard_check(guard, a.status);
    return !EXACT && (nwaves > 1 ? geo[5] != 0u : amb_mask != 0ull);
}

// k_icp: one workgroup per pair.  None of the pairs of noisy scans and a few per cent of those with quantised ranges
// need the second pass (icp_pair<EXACT>); the workgroup of such a pair runs it right away - until round 3 a
// second launch did, which cost every batch 4 us whether or not a pair was flagged.  The second pass overwrites
// the outputs of the first.  (Registers are the larger of the two passes' needs, not their sum.)
template <typename T, int QPT, int UNROLL>
__global__ void __launch_bounds__(1024) k_icp(IcpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (a.zero_ptr) {
        // the replay's map counters, cleared on the way (16 bytes a lane, fire and forget) instead of by a fill kernel of
        // its own ahead of this launch: one dispatch less on the replay's stream
        uint4 *p = static_cast<uint4 *>(a.zero_ptr);
        const size_t n16 = a.zero_bytes >> 4;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(0u, 0u, 0u, 0u);
        if (blockIdx.x == 0 && threadIdx.x < ((a.zero_bytes & 15) >> 2)) static_cast<unsigned *>(a.zero_ptr)[(n16 << 2) + threadIdx.x] = 0u;
    }
    if (icp_pair<T, QPT, UNROLL, false>(a, blockIdx.x, smem)) {
        __syncthreads();                                             // the second pass re-uses the LDS
        icp_pair<T, QPT, 2, true>(a, blockIdx.x, smem);
    }
}

static inline int icp_block(int n_src, int qpt)
{
    int per = (n_src + qpt - 1) / qpt;
    int blk = ((per + kWave - 1) / kWave) * kWave;
    return blk < kWave ? kWave : blk;
}

template <typename T>
static hipError_t launch_icp_t(const IcpArgs &a_in, hipStream_t s)
{
    IcpArgs a = a_in;
    int qpt = (a.n_src + 1023) / 1024;
    // Queries per lane for batches (a handful of pairs cannot fill the chip anyway: one query per lane
    // gives the lowest latency, 0.13 instead of 0.15 ms for the drop-in ICP.process call).  Fewer waves per
    // pair repeat the per-iteration fixed work less often: three queries per lane are fastest when
    // the chip is full (10 000 pairs: 0.510 against 0.536 ms; four overlapping 999-pair replays: 6.2
    // against 6.0 M scans/s); a launch that cannot fill the chip on its own runs shorter with two
    // (999 pairs alone: 0.121 against 0.140 ms).  a.qpt_pref: 0 = by batch size, else 1..3.  "Full" goes by the waves the
    // batch would have at two queries per lane: from 7 500 on - 2 500 pairs of 360 beams, 834 pairs of 1 080 (999 such pairs
    // alone: 0.338 ms with three queries per lane against 0.364 with two) - the chip holds them in more than one round.
    const long waves_at_two = (long)a.B * ((a.n_src + 127) / 128);
    int pref = a.qpt_pref > 0 ? a.qpt_pref : (waves_at_two >= 7500 ? 3 : 2);
    if (a.B > 64 && qpt < pref && a.n_src > 64 * pref) qpt = pref;
    if (qpt > 4) qpt = 8;                     // the shapes that exist: 1, 2, 3, 4, 8 queries per lane
    const int block = icp_block(a.n_src, qpt);
    const size_t lds_base = nn_lds_bytes(a.n_tar) + icp_red_bytes(block / kWave) + kIcpExtraLds + kLdsGuard;
    // second, unpadded copy of the target for the beam-window search: only for scans, and only while both copies fit
    // (up to 4 544 beams; larger scans, up to the documented 8 192, and point clouds go by the box search alone)
    a.polar_copy = (a.ranges && lds_base + icp_polar_bytes(a.n_tar) <= 160 * 1024) ? 1 : 0;
    size_t lds = lds_base + (a.polar_copy ? icp_polar_bytes(a.n_tar) : 0);
    // the list of first-iteration queries without a usable window (nn_listed): room for half of the queries - on the
    // benchmark scans a fifth of them is listed - as far as the CU's LDS goes; a full list leaves the rest to the box search
    a.team_cap = 0;
    if (a.polar_copy && a.team_mode == 0) {
        long cap = ((a.n_src + 1) / 2 + 15) / 16 * 16;
        const long room = ((long)160 * 1024 - (long)lds) / (long)(sizeof(double2) + sizeof(int));
        cap = cap < room ? cap : room / 16 * 16;
        a.team_cap = cap > 0 ? (int)cap : 0;
    }
    lds += (size_t)a.team_cap * (sizeof(double2) + sizeof(int));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(a.B);
#define SLAM_ICP_CASE(Q, U)                                                                                     \
    {                                                                                                           \
        if (lds > 64 * 1024) {                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_icp<T, Q, U>),                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            if (e != hipSuccess) return e;                                                                      \
        }                                                                                                       \
        SLAM_LAUNCH((k_icp<T, Q, U>), grid, dim3(block), lds, s, a);                                            \
    }
    // (one wave per pair with six queries per lane - no barriers, the per-iteration fixed work paid
    // once per pair - was measured: 12 % fewer instructions, but 0.28 instead of 0.20 ms alone and
    // no faster with replays overlapping: dropped)
    // (candidates per trip of the beam-window search: see nn_polar.  Four
    // per trip are as fast as two when the chip is full and faster when it is not - four overlapping 999-pair
    // replays 8.5 -> 8.9 M scans/s - since the candidates come from the unpadded copy.)
    if (qpt <= 1) SLAM_ICP_CASE(1, 4)
    else if (qpt <= 2) SLAM_ICP_CASE(2, 4)
    else if (qpt <= 3) SLAM_ICP_CASE(3, 4)
    else if (qpt <= 4) SLAM_ICP_CASE(4, 2)
    else if (qpt <= 8) SLAM_ICP_CASE(8, 2)
    else return hipErrorInvalidValue;   // n_src > 8192
#undef SLAM_ICP_CASE
    return hipGetLastError();
}

hipError_t launch_icp(const IcpArgs &a, int dtype, hipStream_t s)
{
    switch (dtype) {
    case SLAM_F64: return launch_icp_t<double>(a, s);
    case SLAM_F32: return launch_icp_t<float>(a, s);
    case SLAM_F16: return launch_icp_t<__half>(a, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// k_nn: ICP.findNearest as a stand-alone operator (icp.py:90-114).
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_nn(const T *src, const T *tar, int n_src, int n_tar, double *dist, int32_t *idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nblocks = nn_blocks(n_tar);
    double2 *tarL = reinterpret_cast<double2 *>(smem);
    Box *boxes = reinterpret_cast<Box *>(smem + (size_t)nblocks * kNNStride * sizeof(double2));
    Box *boxes4 = boxes + nn_boxes_padded(n_tar);
    const int b = blockIdx.y;
    stage_target(Cloud<T>{tar + (long)b * 2 * n_tar, nullptr, nullptr, nullptr, n_tar}, n_tar, tarL, boxes, boxes4);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = i < n_src;
    const T *s = src + (long)b * 2 * n_src;
    double sx = ok ? ld(s, i) : 0.0, sy = ok ? ld(s, (long)n_src + i) : 0.0;
    double d2; int j;
    bool amb;
    nn_search(tarL, boxes, boxes4, nblocks, n_tar, sx, sy, i, true, ok, d2, j, amb);
    if (amb) { const NNHit h = nn_exact(tarL, true, n_tar, sx, sy); d2 = h.d2; j = h.j; }
    if (ok) {
        dist[(long)b * n_src + i] = (d2 < INFINITY) ? sqrt(d2) : 0.0;
        idx[(long)b * n_src + i] = j;
    }
}

template <typename T>
static hipError_t launch_nn_t(const void *src, const void *tar, int B, int n_src, int n_tar, double *dist,
                              int32_t *idx, hipStream_t s)
{
    size_t lds = nn_lds_bytes(n_tar);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nn<T>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((n_src + 255) / 256, B);
    SLAM_LAUNCH((k_nn<T>), grid, dim3(256), lds, s, static_cast<const T *>(src), static_cast<const T *>(tar),
                       n_src, n_tar, dist, idx);
    return hipGetLastError();
}

hipError_t launch_nn(const void *src, const void *tar, int B, int n_src, int n_tar, int dtype, double *dist,
                     int32_t *idx, hipStream_t s)
{
    switch (dtype) {
    case SLAM_F64: return launch_nn_t<double>(src, tar, B, n_src, n_tar, dist, idx, s);
    case SLAM_F32: return launch_nn_t<float>(src, tar, B, n_src, n_tar, dist, idx, s);
    case SLAM_F16: return launch_nn_t<__half>(src, tar, B, n_src, n_tar, dist, idx, s);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------
// k_kabsch: ICP.getTransform on paired rows (icp.py:149-179), one workgroup per pair.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_kabsch(const double *src, const double *tar, int n, double *T_out)
{
    __shared__ double red[2 * 4 * kMaxWaves];
    __shared__ int same[2];                                          // every row of src / of tar is ONE point (see k_icp)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x;
    const double *a = src + (long)b * 2 * n, *bb = tar + (long)b * 2 * n;
    if (tid == 0) same[0] = same[1] = 1;
    __syncthreads();
    double v[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += blockDim.x) {
        v[0] += a[i]; v[1] += a[n + i]; v[2] += bb[i]; v[3] += bb[n + i];
        if (!(a[i] == a[0] && a[n + i] == a[n])) same[0] = 0;
        if (!(bb[i] == bb[0] && bb[n + i] == bb[n])) same[1] = 0;
    }
    block_sum<4>(v, red, nwaves, wave, lane);
    double dn = (double)n;
    double cax = v[0] / dn, cay = v[1] / dn, cbx = v[2] / dn, cby = v[3] / dn;
    double w[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += blockDim.x) {
        double aax = a[i] - cax, aay = a[n + i] - cay, bbx = bb[i] - cbx, bby = bb[n + i] - cby;
        w[0] += bbx * aax; w[1] += bbx * aay; w[2] += bby * aax; w[3] += bby * aay;
    }
    block_sum<4>(w, red + 4 * kMaxWaves, nwaves, wave, lane);
    if (same[0] || same[1]) w[0] = w[1] = w[2] = w[3] = 0.0;          // collapsed set: canonical R = I (barriers of block_sum passed)
    if (tid == 0) {
        Rigid2 r = kabsch_from_sums(cax, cay, cbx, cby, w[0], w[1], w[2], w[3]);
        double *To = T_out + 9 * (long)b;
        To[0] = r.c; To[1] = -r.s; To[2] = r.tx;
        To[3] = r.s; To[4] = r.c;  To[5] = r.ty;
        To[6] = 0.0; To[7] = 0.0;  To[8] = 1.0;
    }
}

hipError_t launch_kabsch(const double *src, const double *tar, int B, int n, double *T_out, hipStream_t s)
{
    SLAM_LAUNCH(k_kabsch, dim3(B), dim3(256), 0, s, src, tar, n, T_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// k_scan_to_points: laserToNumpy (W7/icp.py:182-195; W12m/slam_ekf.py:115-123).
// x = cos(angle_i) * r, y = sin(angle_i) * r in float64, one IEEE multiply each, with the
// caller's NumPy-computed cos/sin tables: bit-identical to the reference's products.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_scan_to_points(const float *__restrict__ ranges, const double *__restrict__ cos_t,
                                                        const double *__restrict__ sin_t, long total, int n, int clip_inf,
                                                        T *__restrict__ pts)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long b = e / n;
        int i = (int)(e - b * n);
        double r = (double)ranges[e];
        if (clip_inf && r == INFINITY) r = 30.0;                    // MAX_LASER_RANGE, slam_ekf.py:18,119
        st(pts, b * 2 * n + i, cos_t[i] * r);
        st(pts, b * 2 * n + n + i, sin_t[i] * r);
    }
}

hipError_t launch_scan_to_points(const float *ranges, const double *cos_t, const double *sin_t, long total, int n,
                                 int clip_inf, int dtype, void *pts, hipStream_t s)
{
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    switch (dtype) {
    case SLAM_F64:
        SLAM_LAUNCH((k_scan_to_points<double>), dim3(blocks), dim3(256), 0, s, ranges, cos_t, sin_t, total, n, clip_inf, static_cast<double *>(pts));
        break;
    case SLAM_F32:
        SLAM_LAUNCH((k_scan_to_points<float>), dim3(blocks), dim3(256), 0, s, ranges, cos_t, sin_t, total, n, clip_inf, static_cast<float *>(pts));
        break;
    case SLAM_F16:
        SLAM_LAUNCH((k_scan_to_points<__half>), dim3(blocks), dim3(256), 0, s, ranges, cos_t, sin_t, total, n, clip_inf, static_cast<__half *>(pts));
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// k_pose_compose: the pose update of ICP.publishResult (icp.py:185-190) chained over a
// trajectory.  The recurrence is serial, so the transcendental work is done by all lanes
// (delta_yaw; cos/sin of the heading BEFORE each step) and only the two running sums are
// walked by one lane, in the reference's evaluation order:
//   x = (x + cos(th)*tx) - sin(th)*ty;  y = (y + sin(th)*tx) + cos(th)*ty;  th = th + dyaw.
// ---------------------------------------------------------------------------------
// A small footprint on purpose (256 lanes, 9 KB of LDS; 1 024 lanes and 96 KB until the end of round 3): the kernel is one
// workgroup per trajectory that other kernels' workgroups have to make room for - when replays overlap on several contexts
// its launch waited for a CU with that much free LDS, 57 us instead of 27 from dispatch to end, and its stream stood still
// meanwhile (four overlapping 999-pair replays: 9.2 -> 10.0 M scans/s with the small one).
//
// Round 4: a pipeline of the four waves over chunks of 64 steps, one barrier per stage.  The two recurrences are chains of
// dependent float64 adds that ONE lane has to walk in the reference's order, and what they cost is that lane's own
// instruction stream (a lone wave issues an instruction every ~5 cycles: until round 4 a step cost 72 cycles, of which the
// adds themselves are ~27).  Now wave 0 walks the heading chain of chunk c while wave 1 walks the position chains of chunk
// c - 2, wave 2 computes delta_yaw (atan2) of chunk c + 1 and stores the positions of chunk c - 3, and wave 3 computes
// cos / sin of chunk c - 1's headings: every wave on a SIMD of its own, every value through LDS, operands read two per
// instruction (ds_read_b128) a batch ahead of their use, the matrices' elements loaded from memory a stage ahead.  Same
// operations in the same order as before: results are bit-identical.
constexpr int kComposeChunk = 64;
constexpr int kComposeThreads = 256;
constexpr int kComposeBatch = 16;

__global__ void __launch_bounds__(kComposeThreads) k_pose_compose(const double *__restrict__ T, const double *__restrict__ pose0, int n,
                                                                  double *__restrict__ poses)
{
    __shared__ __attribute__((aligned(16))) double dyaw[3][kComposeChunk];           // delta_yaw of a chunk's steps (icp.py:185)
    __shared__ __attribute__((aligned(16))) double thb[2][kComposeChunk];            // heading BEFORE each step
    __shared__ __attribute__((aligned(16))) double ab[2][kComposeChunk][4];          // c tx, s ty | s tx, -(c ty)
    __shared__ __attribute__((aligned(16))) double xo[2][2][kComposeChunk];          // [x | y] after each step
    // (latency chains on single waves: on a SIMD shared with other kernels' waves they would get every fourth issue slot)
    __builtin_amdgcn_s_setprio(3);
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Tl = T + 9 * (long)l * n;
    double *Pl = poses + 3 * (long)l * n;
    const int nc = (n + kComposeChunk - 1) / kComposeChunk;
    // chain state, in the registers of the lanes that walk the chains
    double th = pose0[3 * l + 2];
    double v = pose0[3 * l + (lane & 1)];                             // wave 1: lane 0 walks x, lane 1 walks y
    // matrix elements a stage ahead: wave 2 wants T10, T00 of chunk s, wave 3 T02, T12 of chunk s - 2
    auto load2 = [&](int chunk, int e0, int e1, double &u0, double &u1) {
        const int k = chunk * kComposeChunk + lane;
        const bool in = chunk >= 0 && chunk < nc && k < n;
        u0 = in ? Tl[9 * (long)k + e0] : 0.0;
        u1 = in ? Tl[9 * (long)k + e1] : 0.0;
    };
    double m0 = 0.0, m1 = 0.0;
    if (wave == 2) load2(0, 3, 0, m0, m1);
    if (wave == 3) load2(-2, 2, 5, m0, m1);
    for (int s = 0; s < nc + 4; ++s) {
        if (wave == 0) {
            // heading chain of chunk s - 1: th_before[k] = th; th = th + dyaw[k] (icp.py:190)
            const int c = s - 1;
            if (c >= 0 && c < nc && lane == 0) {
                const double2 *d2 = reinterpret_cast<const double2 *>(dyaw[c % 3]);
                double2 *o2 = reinterpret_cast<double2 *>(thb[c & 1]);
                double2 cur[kComposeBatch / 2], nxt[kComposeBatch / 2];
#pragma unroll
                for (int u = 0; u < kComposeBatch / 2; ++u) cur[u] = d2[u];
#pragma unroll
                for (int b0 = 0; b0 < kComposeChunk; b0 += kComposeBatch) {
                    if (b0 + kComposeBatch < kComposeChunk) {
#pragma unroll
                        for (int u = 0; u < kComposeBatch / 2; ++u) nxt[u] = d2[(b0 + kComposeBatch) / 2 + u];
                    }
#pragma unroll
                    for (int u = 0; u < kComposeBatch / 2; ++u) {
                        double2 o;
                        o.x = th; th = th + cur[u].x;
                        o.y = th; th = th + cur[u].y;
                        o2[b0 / 2 + u] = o;
                    }
#pragma unroll
                    for (int u = 0; u < kComposeBatch / 2; ++u) cur[u] = nxt[u];
                }
            }
        } else if (wave == 1) {
            // position chains of chunk s - 3, in the reference's evaluation order (icp.py:188-189):
            //   x = (x + c tx) - s ty;   y = (y + s tx) + c ty = (y + s tx) - (-(c ty))
            const int c = s - 3;
            if (c >= 0 && c < nc && lane < 2) {
                const double2 *p2 = reinterpret_cast<const double2 *>(&ab[c & 1][0][0]) + lane;     // step k: p2[2 k]
                double2 *o2 = reinterpret_cast<double2 *>(xo[c & 1][lane]);
                double2 cur[kComposeBatch], nxt[kComposeBatch];
#pragma unroll
                for (int u = 0; u < kComposeBatch; ++u) cur[u] = p2[2 * u];
#pragma unroll
                for (int b0 = 0; b0 < kComposeChunk; b0 += kComposeBatch) {
                    if (b0 + kComposeBatch < kComposeChunk) {
#pragma unroll
                        for (int u = 0; u < kComposeBatch; ++u) nxt[u] = p2[2 * (b0 + kComposeBatch + u)];
                    }
#pragma unroll
                    for (int u = 0; u < kComposeBatch; u += 2) {
                        double2 o;
                        v = (v + cur[u].x) - cur[u].y;
                        o.x = v;
                        v = (v + cur[u + 1].x) - cur[u + 1].y;
                        o.y = v;
                        o2[(b0 + u) / 2] = o;
                    }
#pragma unroll
                    for (int u = 0; u < kComposeBatch; ++u) cur[u] = nxt[u];
                }
            }
        } else if (wave == 2) {
            // delta_yaw of chunk s (this stage's matrix elements were loaded a stage ago; the next chunk's are requested
            // first), then the finished positions of chunk s - 4
            double n0, n1;
            load2(s + 1, 3, 0, n0, n1);
            if (s < nc) dyaw[s % 3][lane] = (s * kComposeChunk + lane < n) ? atan2(m0, m1) : 0.0;   // icp.py:185
            m0 = n0; m1 = n1;
            const int c = s - 4, k = c * kComposeChunk + lane;
            if (c >= 0 && k < n) {
                Pl[3 * (long)k] = xo[c & 1][0][lane];
                Pl[3 * (long)k + 1] = xo[c & 1][1][lane];
            }
        } else {
            // cos / sin of the heading before each step of chunk s - 2, the four products the position chains add, and the
            // heading after the step
            double n0, n1;
            load2(s - 1, 2, 5, n0, n1);
            const int c = s - 2, k = c * kComposeChunk + lane;
            if (c >= 0 && c < nc) {
                double a_ = 0.0, b_ = 0.0, c_ = 0.0, d_ = 0.0;
                if (k < n) {
                    const double h = thb[c & 1][lane];
                    const double cs = cos(h), sn = sin(h);
                    a_ = cs * m0; b_ = sn * m1; c_ = sn * m0; d_ = cs * m1;
                    Pl[3 * (long)k + 2] = h + dyaw[c % 3][lane];     // icp.py:190
                }
                double2 *q = reinterpret_cast<double2 *>(&ab[c & 1][lane][0]);
                q[0] = make_double2(a_, b_);
                q[1] = make_double2(c_, -d_);
            }
            m0 = n0; m1 = n1;
        }
        __syncthreads();
    }
}

// One dead-reckoning step per hypothesis (n = 1, many trajectories): pose = pose0 (+) M with
// M = T.[prior; 0 0 1] when a prior was applied to the source before the solve (the solve's T
// maps the PERTURBED source, so the motion of the original scan is the product).
__global__ void __launch_bounds__(256) k_pose_step(const double *__restrict__ T, const double *__restrict__ pose0,
                                                   const double *__restrict__ prior, int L, double *__restrict__ poses,
                                                   double *__restrict__ heading_cs)
{
    int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const double *t = T + 9 * (long)l;
    double m00 = t[0], m10 = t[3], m02 = t[2], m12 = t[5];
    if (prior) {
        const double *p = prior + 6 * (long)l;
        m00 = t[0] * p[0] + t[1] * p[3];
        m10 = t[3] * p[0] + t[4] * p[3];
        m02 = t[0] * p[2] + t[1] * p[5] + t[2];
        m12 = t[3] * p[2] + t[4] * p[5] + t[5];
    }
    double x = pose0[3 * l], y = pose0[3 * l + 1], th = pose0[3 * l + 2];
    double dyaw = atan2(m10, m00);                                   // icp.py:185
    double c = cos(th), s = sin(th);
    poses[3 * l] = (x + c * m02) - s * m12;                          // :188
    poses[3 * l + 1] = (y + s * m02) + c * m12;                      // :189
    const double th1 = th + dyaw;
    poses[3 * l + 2] = th1;                                          // :190
    // cos / sin of the NEW heading for the ray cast that follows (u2T, slam_ekf.py:130-137): one evaluation here
    // instead of one per lane of the map's workgroup there; same functions, same argument bits, same results
    if (heading_cs) { heading_cs[2 * l] = cos(th1); heading_cs[2 * l + 1] = sin(th1); }
}

hipError_t launch_pose_compose(const double *T, const double *pose0, int L, int n, double *poses, hipStream_t s,
                               const double *prior, double *heading_cs)
{
    // (heading_cs is written by k_pose_step only: a caller that asks for it - the particle pipeline's ray cast reads it -
    // takes that kernel whatever the batch size and whether or not there are priors)
    if (prior || heading_cs || (n == 1 && L > 64)) {
        if (n != 1) return hipErrorInvalidValue;
        SLAM_LAUNCH(k_pose_step, dim3((L + 255) / 256), dim3(256), 0, s, T, pose0, prior, L, poses, heading_cs);
        return hipGetLastError();
    }
    SLAM_LAUNCH(k_pose_compose, dim3(L), dim3(kComposeThreads), 0, s, T, pose0, n, poses);
    return hipGetLastError();
}

}  // namespace slam
