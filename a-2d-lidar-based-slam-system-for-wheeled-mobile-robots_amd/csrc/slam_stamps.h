// Diagnostic builds only (never shipped): in-kernel phase timers of the ray-cast and scan-matching kernels.
//   -DSLAM_STAMPS      grid kernels: thread 0 of every workgroup adds the shader-clock cycles of each phase to 64-bit
//                      counters behind the context's status word (status + 8 ints)
//   -DSLAM_STAMPS_ICP  k_icp: the same per scan pair (not together with SLAM_STAMPS: same counters)
// slam_debug_read (slam_abi.hip, compiled in only with one of the two) returns and clears the counters.  Without the
// switches every macro below is empty.
#pragma once

// ---- grid kernels: STAMP_DECL at kernel entry, STAMP(k) closes phase k, STAMP_END(k): [k] lifetime, [k + 1] workgroups
#ifdef SLAM_STAMPS
// (behind the 256-byte status block the stamps build allocates kStampRecords records of 8 dwords, one per workgroup in the
// order they finish: phases 0..4 in shader cycles, lifetime, blockIdx.x | blockIdx.y << 16, a value of the kernel's choice (STAMP_VAL; else the start on the 100 MHz clock))
constexpr int kStampRecords = 32768;
#define STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_first = st_t0; unsigned st_rec[5] = {0u, 0u, 0u, 0u, 0u}; unsigned st_real = (unsigned)__builtin_amdgcn_s_memrealtime()
#define STAMP(k)                                                                                         \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
            atomicAdd(reinterpret_cast<unsigned long long *>(g.status + 8) + (k), t_ - st_t0);           \
            st_rec[(k) < 5 ? (k) : 4] += (unsigned)(t_ - st_t0);                                         \
            st_t0 = t_;                                                                                  \
        }                                                                                                \
    } while (0)
#define STAMP_END(k)                                                                                     \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
            atomicAdd(reinterpret_cast<unsigned long long *>(g.status + 8) + (k), t_ - st_first);        \
            const unsigned long long id_ = atomicAdd(reinterpret_cast<unsigned long long *>(g.status + 8) + (k) + 1, 1ull); \
            if (id_ < (unsigned long long)kStampRecords) {                                               \
                unsigned *r_ = reinterpret_cast<unsigned *>(g.status) + 64 + 8 * id_;                    \
                for (int q_ = 0; q_ < 5; ++q_) r_[q_] = st_rec[q_];                                      \
                r_[5] = (unsigned)(t_ - st_first); r_[6] = blockIdx.x | (blockIdx.y << 16); r_[7] = st_real; \
            }                                                                                            \
        }                                                                                                \
    } while (0)
#define STAMP_VAL(v) do { st_real = (unsigned)(v); } while (0)
#define STAMP_SYNC() __syncthreads()        /* so that a phase's time is the workgroup's, not thread 0's */
#define STAMP_COUNT(k, v)                                                                                \
    do {                                                                                                 \
        if (threadIdx.x == 0) atomicAdd(reinterpret_cast<unsigned long long *>(g.status + 8) + (k), (unsigned long long)(v)); \
    } while (0)
#define STAMP_WAVE_COUNT(k, v)   /* lane 0 of every wave adds v (wave-uniform) to counter k */                     \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0) atomicAdd(reinterpret_cast<unsigned long long *>(g.status + 8) + (k), (unsigned long long)(v)); \
    } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_END(k)
#define STAMP_SYNC()
#define STAMP_COUNT(k, v)
#define STAMP_WAVE_COUNT(k, v)
#define STAMP_VAL(v)
#endif

// Diagnostic build (-DSLAM_STAMPS_ICP, never shipped; not together with the grid kernels' SLAM_STAMPS: same counters): thread 0 of every pair adds the shader-clock cycles
// of each phase to the 64-bit counters behind the status word (slam_debug_read): [0] staging, [1]
// search, [2] centroid reduction, [3] products + reduction, [4] Kabsch + transform, [5] final T;
// [6..9]: phases 1..4 of the FIRST iteration (the others hold iterations >= 1), [10] iterations, [11] pairs, [12] lifetimes,
// [13] lifetimes on the 100 MHz clock, [14] 2^62 - earliest start, [15] latest end (100 MHz clock).
// Lane efficiency of the beam-window search (nn_polar::scan), -DSLAM_STAMPS_ICP only: candidates inside the lanes' own
// windows against the candidate slots the waves ran (64 lanes x trips x candidates per trip), for the first iteration
// ([0], [1]) and the later ones ([2], [3]); slam_debug_lanes (slam_abi.hip) reads and clears them.
// (-DSLAM_STAMPS_ICP=3: the phase timers alone - the lane counters' atomics, four addresses for every wave of the launch,
// stretch a pair's lifetime tenfold and with it every phase)
#if defined(SLAM_STAMPS_ICP) && SLAM_STAMPS_ICP != 3
extern __device__ unsigned long long g_polar_lanes[4];
#define ISTAMP_SCAN(first, own, trips, per_trip)                                                          \
    do {                                                                                                 \
        unsigned o_ = (unsigned)(own);                                                                   \
        for (int off_ = 32; off_ > 0; off_ >>= 1) o_ += __shfl_xor(o_, off_);                            \
        if ((threadIdx.x & 63) == 0) {                                                                   \
            atomicAdd(&g_polar_lanes[(first) ? 0 : 2], (unsigned long long)o_);                          \
            atomicAdd(&g_polar_lanes[(first) ? 1 : 3], (unsigned long long)(trips) * (per_trip) * 64ull); \
        }                                                                                                \
    } while (0)
#else
#define ISTAMP_SCAN(first, own, trips, per_trip)
#endif

#ifdef SLAM_STAMPS_ICP
#define ISTAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_first = st_t0, st_acc[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_big[4] = {0, 0, 0, 0}, st_real = __builtin_amdgcn_s_memrealtime()
#define ISTAMP(k)                                                                                        \
    do {                                                                                                 \
        unsigned long long t_ = __builtin_amdgcn_s_memtime();                                            \
        st_acc[k] += t_ - st_t0;                                                                         \
        st_t0 = t_;                                                                                      \
    } while (0)
#if SLAM_STAMPS_ICP == 2   /* when do pairs start and end?  [0..11], [12..23]: histograms in 15 us buckets from the first start; [24]: that start */
#define ISTAMP_END(iters)                                                                                \
    do {                                                                                                 \
        if (threadIdx.x == 0) {                                                                          \
            unsigned long long *c_ = reinterpret_cast<unsigned long long *>(a.status + 8);              \
            unsigned long long base_ = atomicCAS(c_ + 24, 0ull, st_real);                                \
            if (base_ == 0ull) base_ = st_real;                                                          \
            unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                    \
            long long s0_ = (long long)(st_real - base_) / 1500, s1_ = (long long)(r_ - base_) / 1500;   \
            atomicAdd(c_ + min(max(s0_, 0ll), 11ll), 1ull);                                              \
            atomicAdd(c_ + 12 + min(max(s1_, 0ll), 11ll), 1ull);                                         \
        }                                                                                                \
    } while (0)
#define ISTAMP_BIG(it, big)
#else
#define ISTAMP_BIG(it, big)                                                                              \
    do {                                                                                                 \
        st_big[(it) == 0 ? 0 : 2] += __popcll(__ballot(big));                                            \
        st_big[(it) == 0 ? 1 : 3] += __any(big) ? 1 : 0;                                                 \
    } while (0)
#define ISTAMP_END(iters)                                                                                \
    do {                                                                                                 \
        if ((threadIdx.x & 63) == 0)                                                                     \
            for (int k_ = 0; k_ < 4; ++k_) atomicAdd(reinterpret_cast<unsigned long long *>(a.status + 8) + 21 + k_, st_big[k_]); \
        if (threadIdx.x == 0) {                                                                          \
            unsigned long long *c_ = reinterpret_cast<unsigned long long *>(a.status + 8);              \
            for (int k_ = 0; k_ < 10; ++k_) atomicAdd(c_ + k_, st_acc[k_]);                              \
            for (int k_ = 0; k_ < 5; ++k_) atomicAdd(c_ + 16 + k_, st_acc[10 + k_]);                      \
            atomicAdd(c_ + 10, (unsigned long long)(iters));                                             \
            atomicAdd(c_ + 11, 1ull);                                                                    \
            atomicAdd(c_ + 12, __builtin_amdgcn_s_memtime() - st_first);                                 \
            unsigned long long r_ = __builtin_amdgcn_s_memrealtime();                                    \
            atomicAdd(c_ + 13, r_ - st_real);                                                            \
            atomicMax(c_ + 14, (1ull << 62) - st_real);                                                  \
            atomicMax(c_ + 15, r_);                                                                      \
        }                                                                                                \
    } while (0)
#endif
#else
#define ISTAMP_DECL
#define ISTAMP(k)
#define ISTAMP_BIG(it, big)
#define ISTAMP_END(iters)
#endif

