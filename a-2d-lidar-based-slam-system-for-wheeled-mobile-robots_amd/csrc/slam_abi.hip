// C-ABI layer of libslamhip.so (include/slam_hip.h): contexts, workspaces, error
// reporting, host<->device staging and the launch sequences.  No compute happens on the
// host; without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <initializer_list>
#include <mutex>
#include <utility>
#include <vector>

#include "slam_internal.h"
#include "slam_stamps.h"

using namespace slam;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(SLAM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

#define REQUIRE(cond, msg)                                                                             \
    do {                                                                                               \
        if (!(cond)) return fail(SLAM_ERR_INVALID, "%s: %s", __func__, msg);                           \
    } while (0)

size_t dtype_size(int dtype)
{
    switch (dtype) {
    case SLAM_F64: return 8;
    case SLAM_F32: return 4;
    case SLAM_F16: return 2;
    }
    return 0;
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
};

struct Pending {
    int kind;
    hipEvent_t e0, e1;
    bool first;      // first launch of its timing bracket: the family's launch count goes by brackets
};

}  // namespace

struct slam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Arena staging;   // device copies of host arguments (host-pointer entry points)
    Arena scratch;   // temporaries of *_dev sequences
    Arena tiles;     // recorded walks of the tiled ray cast
    char *pinned = nullptr;   // page-locked host block: small calls go through it in one copy each way
    size_t pinned_cap = 0;
    int *status = nullptr;
    bool timing = false;
    unsigned timing_mask = ~0u;   // families (bit SLAM_K_*) whose launches carry events while timing is on
    std::vector<hipEvent_t> pool;
    std::vector<Pending> pending;
    double ms[SLAM_K_COUNT] = {0};
    int64_t launches[SLAM_K_COUNT] = {0};
    int grid_mode = 1;    // 0: direct global atomics, 1: automatic, 2: tiles, 3: LDS window, 4: wedges
    int grid_group = 0;   // scans per workgroup in window mode (0: automatic)
    int grid_split = -1;  // window mode: two workgroups per group, one per direction half (-1: when the launch cannot fill the chip)
    int icp_qpt = 0;      // queries per lane of batched scan matching (0: by batch size)
    int replay_reset = 0; // 1: slam_replay_dev clears its map's counters itself (inside the scan-matching launch)
    int icp_team = 0;     // first-iteration queries without a beam window: 0 = listed and searched apart from their lanes (nn_listed), 1 = box search
    // "pipeline" option: the map stage of slam_replay_dev (reset -> ray cast -> finalize) runs on
    // a second stream, so the map stage of one replay overlaps the scan matching of the next.
    int pipeline = 0;
    hipStream_t gstream = nullptr;                   // map stage
    hipStream_t cstream = nullptr;                   // pose composition (a serial chain in one workgroup)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_icp = nullptr, ev_pose = nullptr;
    hipEvent_t ev_cast[2] = {nullptr, nullptr};     // "compose + ray cast of that replay have read T / poses"
    const void *cast_poses[2] = {nullptr, nullptr};
    const void *cast_T[2] = {nullptr, nullptr};
    double *pipe_T[2] = {nullptr, nullptr};          // T buffers when the caller passes none
    size_t pipe_T_bytes = 0;
    int cast_pos = 0;
    bool gdirty = false;                             // work on gstream the main stream has not waited for
    bool mgrid = false;                              // map work on the main stream gstream has not waited for
    // Particle batches in chunks (option "particle_chunks"): scan matching and pose steps of chunk k on the main stream,
    // its ray cast on pstream behind an event - the vector-issue-bound matcher of chunk k + 1 shares the chip with the
    // memory-bound ray cast of chunk k (slam_particles_dev).
    int particle_chunks = 0;                         // 0 / 1: off, else the number of chunks
    hipStream_t pstream = nullptr;
    std::vector<hipEvent_t> pev_pose, pev_cast;      // per chunk: "poses written" (main -> pstream), "ray cast done" (pstream -> main)
    hipEvent_t pev_join = nullptr;
    bool pdirty = false;                             // ray casts on pstream the main stream has not waited for
    hipEvent_t ev_order = nullptr;                   // slam_stream_order: ordering with a caller's stream
};

struct slam_grid {
    GridDev d;
    unsigned long long *visits = nullptr;
    size_t state_bytes = 0;        // pass[] + hit[] + visit counter, one allocation
    int8_t *pmap_one = nullptr;    // [xw][yw] read-back staging
    double *datamap_one = nullptr;
    int8_t *pmap_live = nullptr;   // [G][xw][yw], slam_grid_live_pmap
    bool live_dirty = false;       // pmap_live is behind the counters
    bool pristine = true;          // nothing has been cast since creation / the last reset: every cell is 50
};

namespace slam {
thread_local LaunchTimer g_launch_timer = {nullptr, nullptr};

hipError_t allow_dynamic_lds(const void *kernel, int bytes)
{
    struct Entry { int dev; const void *kernel; int bytes; };
    static std::mutex mu;
    static std::vector<Entry> done;       // largest size the attribute has been set to, per (device, kernel)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    Entry *hit = nullptr;
    for (auto &d : done)
        if (d.dev == dev && d.kernel == kernel) hit = &d;
    if (hit && hit->bytes >= bytes) return hipSuccess;
    // (a kernel whose LDS need depends on the launch - k_wedge_sort: beams x scans per group - asks again with more)
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    if (hit) hit->bytes = bytes;
    else done.push_back(Entry{dev, kernel, bytes});
    return hipSuccess;
}
}  // namespace slam

namespace {

int arena_reserve(slam_ctx *c, Arena &a, size_t bytes)
{
    a.used = 0;
    if (bytes <= a.cap) return SLAM_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (a.base) HIPCHK(hipFree(a.base));
    a.base = nullptr;
    a.cap = 0;
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&a.base), want);
    if (e != hipSuccess) return fail(SLAM_ERR_NOMEM, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e));
    a.cap = want;
    return SLAM_OK;
}

// Next piece of a reserved arena.  A piece that does not fit (a reservation computed too small:
// an internal error) comes back null, which the entry point's null checks / the first copy
// into it turn into an error code instead of a silent overrun.
template <typename T>
T *carve(Arena &a, size_t count)
{
    size_t off = align_up(a.used);
    if (off + count * sizeof(T) > a.cap) {
        a.used = a.cap + 1;
        (void)fail(SLAM_ERR_NOMEM, "internal: workspace reservation too small (%zu > %zu bytes)", off + count * sizeof(T), a.cap);
        return nullptr;
    }
    a.used = off + count * sizeof(T);
    return reinterpret_cast<T *>(a.base + off);
}

// RAII bracket around the kernel launch(es) of a family: every SLAM_LAUNCH inside it draws a
// fresh pair of events from it (slam_internal.h), and the family's time is the sum of its launches.
struct Timed {
    slam_ctx *c;
    int kind;
    LaunchTimer saved;
    bool used = false;
    // counts: the bracket's first launch counts as a launch of its family (false: one more piece of a family whose
    // launches are counted elsewhere - the later chunks of a particle batch)
    Timed(slam_ctx *ctx, int k, hipStream_t = nullptr, bool counts = true) : c(ctx), kind(k), saved(g_launch_timer), used(!counts)
    {
        if (c->timing && (c->timing_mask >> k & 1u)) g_launch_timer = {this, &Timed::next};
    }
    ~Timed() { g_launch_timer = saved; }
    static bool next(void *self, hipEvent_t *e0, hipEvent_t *e1)
    {
        Timed *t = static_cast<Timed *>(self);
        *e0 = t->get();
        *e1 = t->get();
        if (!*e0 || !*e1) {
            if (*e0) t->c->pool.push_back(*e0);
            if (*e1) t->c->pool.push_back(*e1);
            return false;
        }
        t->c->pending.push_back({t->kind, *e0, *e1, !t->used});
        t->used = true;
        return true;
    }
    hipEvent_t get()
    {
        if (!c->pool.empty()) {
            hipEvent_t e = c->pool.back();
            c->pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

int use(slam_ctx *c)
{
    if (!c) return fail(SLAM_ERR_INVALID, "null context");
    HIPCHK(hipSetDevice(c->device));
    return SLAM_OK;
}

// maps much larger than a window: direction wedges (grid_mode 1 automatic, 4) unless the recorded-walk tiles are asked
// for (2) or the map's sides exceed what the wedges' packed end cells hold
int wedges_ok(const slam_ctx *c, const slam_grid *g) { return c->grid_mode != 2 && g->d.xw <= 65535 && g->d.yw <= 65535; }

// stream of the pipelined map stage
hipStream_t gs(slam_ctx *c) { return c->pipeline ? c->gstream : c->stream; }

// the map stream must see everything enqueued on the main stream so far
int fork_to_grid(slam_ctx *c)
{
    if (!c->pipeline) return SLAM_OK;
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->gstream, c->ev_fork, 0));
    c->gdirty = true;
    return SLAM_OK;
}

// the main stream must see the ray casts of chunked particle batches (pstream) enqueued so far
int join_particles(slam_ctx *c)
{
    if (!c->pdirty) return SLAM_OK;
    HIPCHK(hipEventRecord(c->pev_join, c->pstream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->pev_join, 0));
    c->pdirty = false;
    return SLAM_OK;
}

// the main stream must see everything enqueued on the map stream so far
int join_from_grid(slam_ctx *c)
{
    if (int rc = join_particles(c)) return rc;
    if (!c->pipeline || !c->gdirty) return SLAM_OK;
    HIPCHK(hipEventRecord(c->ev_join, c->gstream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->gdirty = false;
    return SLAM_OK;
}

constexpr size_t kPinnedBytes = 256 * 1024;

bool ensure_pinned(slam_ctx *c)
{
    if (c->pinned) return true;
    if (hipHostMalloc(reinterpret_cast<void **>(&c->pinned), kPinnedBytes, hipHostMallocDefault) != hipSuccess) {
        c->pinned = nullptr;
        (void)hipGetLastError();
        return false;
    }
    c->pinned_cap = kPinnedBytes;
    return true;
}

// Host <-> staging copies of an entry point.  The pieces lie in ONE arena in ascending order, so
// when they are small (a drop-in call: one scan pair, one scan) they cross the bus in one copy
// through the page-locked block, laid out like the device pieces; otherwise one copy per piece.
struct Seg {
    void *dev;
    void *host;          // source (copy_in) or destination (copy_out_sync); null pieces are skipped
    size_t bytes;
};

int copy_in(slam_ctx *c, std::initializer_list<Seg> segs)
{
    char *lo = nullptr, *hi = nullptr;
    for (const Seg &g : segs) {
        if (!g.host || !g.bytes) continue;
        if (!g.dev) return fail(SLAM_ERR_NOMEM, "internal: workspace");
        char *d = static_cast<char *>(g.dev);
        if (!lo || d < lo) lo = d;
        if (!hi || d + g.bytes > hi) hi = d + g.bytes;
    }
    if (!lo) return SLAM_OK;
    const size_t span = (size_t)(hi - lo);
    if (span <= kPinnedBytes / 2 && ensure_pinned(c)) {
        for (const Seg &g : segs)
            if (g.host && g.bytes) memcpy(c->pinned + (static_cast<char *>(g.dev) - lo), g.host, g.bytes);
        HIPCHK(hipMemcpyAsync(lo, c->pinned, span, hipMemcpyHostToDevice, c->stream));
        return SLAM_OK;
    }
    for (const Seg &g : segs)
        if (g.host && g.bytes) HIPCHK(hipMemcpyAsync(g.dev, g.host, g.bytes, hipMemcpyHostToDevice, c->stream));
    return SLAM_OK;
}

int copy_out_sync(slam_ctx *c, std::initializer_list<Seg> segs)
{
    char *lo = nullptr, *hi = nullptr;
    for (const Seg &g : segs) {
        if (!g.host || !g.bytes) continue;
        if (!g.dev) return fail(SLAM_ERR_NOMEM, "internal: workspace");
        char *d = static_cast<char *>(g.dev);
        if (!lo || d < lo) lo = d;
        if (!hi || d + g.bytes > hi) hi = d + g.bytes;
    }
    if (lo) {
        const size_t span = (size_t)(hi - lo);
        if (span <= kPinnedBytes / 2 && ensure_pinned(c)) {
            char *h = c->pinned + kPinnedBytes / 2;
            HIPCHK(hipMemcpyAsync(h, lo, span, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (const Seg &g : segs)
                if (g.host && g.bytes) memcpy(g.host, h + (static_cast<char *>(g.dev) - lo), g.bytes);
            return SLAM_OK;
        }
        for (const Seg &g : segs)
            if (g.host && g.bytes) HIPCHK(hipMemcpyAsync(g.host, g.dev, g.bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

// entry points that touch a map on the MAIN stream call this first
int grid_on_main(slam_ctx *c)
{
    if (int rc = join_particles(c)) return rc;
    if (!c->pipeline) return SLAM_OK;
    c->mgrid = true;
    return join_from_grid(c);
}

int status_to_code(int st)
{
    if (st & kStatusGuard) return fail(SLAM_ERR_HIP, "internal: a kernel wrote past its LDS regions (SLAM_LDS_GUARD build)");
    if (st & kStatusNaN) return fail(SLAM_ERR_NAN, "cannot convert float NaN to integer (mapping.py:33-36)");
    if (st & kStatusOverflow)
        return fail(SLAM_ERR_OVERFLOW, "cannot convert float infinity to integer / cell index beyond 2^20 (mapping.py:33-36)");
    return SLAM_OK;
}

int check_status_sync(slam_ctx *c)
{
    int st = 0;
    if (int rc = join_from_grid(c)) return rc;
    HIPCHK(hipMemcpyAsync(&st, c->status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (st) {
        HIPCHK(hipMemsetAsync(c->status, 0, sizeof(int), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return status_to_code(st);
}

#define H2D(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream))
#define D2H(dst, src, bytes) HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream))
#define TRY(expr)                                                                                      \
    do {                                                                                               \
        int rc_ = (expr);                                                                              \
        if (rc_ != SLAM_OK) return rc_;                                                                \
    } while (0)

}  // namespace

extern "C" {

int slam_abi_version(void) { return SLAM_ABI_VERSION; }

const char *slam_last_error(void) { return g_err.c_str(); }

int slam_create(int device, void *stream, slam_ctx **out)
{
    if (!out) return fail(SLAM_ERR_INVALID, "slam_create: out is null");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(SLAM_ERR_NODEVICE, "no HIP device visible (%s); this library has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(SLAM_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SLAM_ERR_NODEVICE, "device %d is %s; libslamhip is built for gfx950 (MI355X) only", device,
                    prop.gcnArchName);
    slam_ctx *c = new slam_ctx();
    c->device = device;
    if (stream) {
        c->stream = static_cast<hipStream_t>(stream);
    } else {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(SLAM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
#ifdef SLAM_STAMPS
    const size_t status_bytes = 256 + (size_t)kStampRecords * 32;   // diagnostic build: per-workgroup phase records behind the status block
#else
    const size_t status_bytes = 256;
#endif
    e = hipMalloc(reinterpret_cast<void **>(&c->status), status_bytes);
    if (e == hipSuccess) e = hipMemsetAsync(c->status, 0, status_bytes, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        delete c;
        return fail(SLAM_ERR_HIP, "context init: %s", hipGetErrorString(e));
    }
    *out = c;
    return SLAM_OK;
}

int slam_destroy(slam_ctx *c)
{
    if (!c) return SLAM_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->pstream) {
        (void)hipStreamSynchronize(c->pstream);
        (void)hipStreamDestroy(c->pstream);
        for (hipEvent_t e : c->pev_pose) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->pev_cast) (void)hipEventDestroy(e);
        if (c->pev_join) (void)hipEventDestroy(c->pev_join);
    }
    if (c->gstream) {
        (void)hipStreamSynchronize(c->cstream);
        (void)hipStreamSynchronize(c->gstream);
        (void)hipStreamDestroy(c->cstream);
        (void)hipStreamDestroy(c->gstream);
        for (hipEvent_t e : {c->ev_fork, c->ev_join, c->ev_icp, c->ev_pose, c->ev_cast[0], c->ev_cast[1]})
            if (e) (void)hipEventDestroy(e);
        for (double *t : c->pipe_T)
            if (t) (void)hipFree(t);
    }
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    for (auto &p : c->pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->staging.base) (void)hipFree(c->staging.base);
    if (c->scratch.base) (void)hipFree(c->scratch.base);
    if (c->tiles.base) (void)hipFree(c->tiles.base);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->status) (void)hipFree(c->status);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SLAM_OK;
}

int slam_synchronize(slam_ctx *c)
{
    TRY(use(c));
    TRY(join_from_grid(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_stream_order(slam_ctx *c, void *stream, int direction)
{
    TRY(use(c));
    REQUIRE(direction == 0 || direction == 1, "direction is 0 (the stream waits for the context) or 1 (the context waits for the stream)");
    hipStream_t other = static_cast<hipStream_t>(stream);
    if (!c->ev_order) HIPCHK(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    if (direction == 0) {
        TRY(join_from_grid(c));                    // the map / pose / particle streams into the context's stream first
        if (other == c->stream) return SLAM_OK;
        HIPCHK(hipEventRecord(c->ev_order, c->stream));
        HIPCHK(hipStreamWaitEvent(other, c->ev_order, 0));
        return SLAM_OK;
    }
    if (other != c->stream) {
        HIPCHK(hipEventRecord(c->ev_order, other));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_order, 0));
    }
    // with the "pipeline" option the map stage runs on a stream of its own: it forks from the context's stream again before
    // its next piece of work (as after any map work on the context's stream)
    if (c->pipeline) {
        TRY(join_from_grid(c));
        c->mgrid = true;
    }
    return SLAM_OK;
}

int slam_check_status(slam_ctx *c)
{
    TRY(use(c));
    return check_status_sync(c);
}

int slam_set_option(slam_ctx *c, const char *name, double value)
{
    TRY(use(c));
    REQUIRE(name, "null name");
    if (!strcmp(name, "grid_mode")) { REQUIRE(value == 0 || value == 1 || value == 2 || value == 3 || value == 4, "grid_mode is 0..4"); c->grid_mode = (int)value; }
    else if (!strcmp(name, "grid_group")) { REQUIRE(value >= 0 && value <= 64, "grid_group in [0, 64]"); c->grid_group = (int)value; }
    else if (!strcmp(name, "grid_split")) { REQUIRE(value == -1 || value == 0 || value == 1, "grid_split is -1, 0 or 1"); c->grid_split = (int)value; }
    else if (!strcmp(name, "replay_reset")) { REQUIRE(value == 0 || value == 1, "replay_reset is 0 or 1"); c->replay_reset = (int)value; }
    else if (!strcmp(name, "icp_team")) { REQUIRE(value == 0 || value == 1, "icp_team is 0 or 1"); c->icp_team = (int)value; }
    else if (!strcmp(name, "particle_chunks")) { REQUIRE(value >= 0 && value <= 64 && value == (int)value, "particle_chunks in [0, 64]"); TRY(join_particles(c)); c->particle_chunks = (int)value; }
    else if (!strcmp(name, "icp_qpt")) { REQUIRE(value >= 0 && value <= 3, "icp_qpt in [0, 3]"); c->icp_qpt = (int)value; }
    else if (!strcmp(name, "pipeline")) {
        REQUIRE(value == 0 || value == 1, "pipeline is 0 or 1");
        TRY(join_from_grid(c));
        if (value == 1 && !c->gstream) {
            HIPCHK(hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking));
            HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
            for (hipEvent_t *e : {&c->ev_fork, &c->ev_join, &c->ev_icp, &c->ev_pose, &c->ev_cast[0], &c->ev_cast[1]})
                HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        if (value == 0 && c->pipeline) HIPCHK(hipStreamSynchronize(c->gstream));
        c->pipeline = (int)value;
        c->cast_poses[0] = c->cast_poses[1] = nullptr;
        c->cast_T[0] = c->cast_T[1] = nullptr;
    }
    else return fail(SLAM_ERR_INVALID, "unknown option %s", name);
    return SLAM_OK;
}

int slam_timing_enable(slam_ctx *c, int on)
{
    TRY(use(c));
    TRY(join_from_grid(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) { c->pool.push_back(p.e0); c->pool.push_back(p.e1); }
    c->pending.clear();
    for (int k = 0; k < SLAM_K_COUNT; ++k) { c->ms[k] = 0; c->launches[k] = 0; }
    c->timing = on != 0;
    c->timing_mask = on > 1 ? (unsigned)on >> 1 : ~0u;   // on = 1: every family; on = 2 * mask: the families in mask (bit SLAM_K_*)
    return SLAM_OK;
}

int slam_timing_read(slam_ctx *c, double ms_out[SLAM_K_COUNT], int64_t launches_out[SLAM_K_COUNT])
{
    TRY(use(c));
    TRY(join_from_grid(c));                 // (timed launches on the map / particle streams)
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            c->ms[p.kind] += ms;
            c->launches[p.kind] += p.first ? 1 : 0;
        }
        c->pool.push_back(p.e0);
        c->pool.push_back(p.e1);
    }
    c->pending.clear();
    for (int k = 0; k < SLAM_K_COUNT; ++k) {
        if (ms_out) ms_out[k] = c->ms[k];
        if (launches_out) launches_out[k] = c->launches[k];
        c->ms[k] = 0;
        c->launches[k] = 0;
    }
    return SLAM_OK;
}

#if defined(SLAM_STAMPS) || defined(SLAM_STAMPS_ICP)
/* diagnostic builds only: the 256-byte status block (word 0: status bits; from word 8: phase counters) */
int slam_debug_read(slam_ctx *c, void *out256, int clear)
{
    TRY(use(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out256, c->status, 256, hipMemcpyDeviceToHost));
    if (clear) HIPCHK(hipMemset(reinterpret_cast<char *>(c->status) + 32, 0, 224));
    return SLAM_OK;
}
#endif
#ifdef SLAM_STAMPS_ICP
/* diagnostic build: lane efficiency counters of the beam-window search (slam_stamps.h, ISTAMP_SCAN): own candidates and
   candidate slots of the first iteration, of the later ones */
int slam_debug_lanes(slam_ctx *c, unsigned long long *out4, int clear)
{
    TRY(use(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(slam::debug_polar_lanes(out4, clear != 0));
    return SLAM_OK;
}
#endif
#ifdef SLAM_STAMPS
/* diagnostic build: the per-workgroup records (8 dwords each, slam_stamps.h) of the launches since the last clear */
int slam_debug_records(slam_ctx *c, void *out, int max_records)
{
    TRY(use(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    const int n = max_records < kStampRecords ? max_records : kStampRecords;
    HIPCHK(hipMemcpy(out, reinterpret_cast<char *>(c->status) + 256, (size_t)n * 32, hipMemcpyDeviceToHost));
    return SLAM_OK;
}
#endif

/* ---- ICP ------------------------------------------------------------------------ */

int slam_scan_to_points_dev(slam_ctx *c, const float *ranges, const double *cos_t, const double *sin_t, int B, int n,
                            int clip_inf, int dtype, void *pts_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pts_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "B and n must be positive");
    REQUIRE(dtype_size(dtype), "unknown dtype");
    Timed t(c, SLAM_K_POINTS);
    HIPCHK(launch_scan_to_points(ranges, cos_t, sin_t, (long)B * n, n, clip_inf, dtype, pts_out, c->stream));
    return SLAM_OK;
}

int slam_scan_to_points(slam_ctx *c, const float *ranges, const double *cos_t, const double *sin_t, int B, int n,
                        int clip_inf, int dtype, void *pts_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pts_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "B and n must be positive");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    size_t nr = (size_t)B * n;
    TRY(arena_reserve(c, c->staging, align_up(nr * 4) + 2 * align_up((size_t)n * 8) + align_up(2 * nr * ds) + 1024));
    float *d_r = carve<float>(c->staging, nr);
    double *d_c = carve<double>(c->staging, n), *d_s = carve<double>(c->staging, n);
    char *d_p = carve<char>(c->staging, 2 * nr * ds);
    TRY(copy_in(c, {{d_r, const_cast<float *>(ranges), nr * 4}, {d_c, const_cast<double *>(cos_t), (size_t)n * 8},
                    {d_s, const_cast<double *>(sin_t), (size_t)n * 8}}));
    TRY(slam_scan_to_points_dev(c, d_r, d_c, d_s, B, n, clip_inf, dtype, d_p));
    return copy_out_sync(c, {{d_p, pts_out, 2 * nr * ds}});
}

int slam_nn_dev(slam_ctx *c, const void *src, const void *tar, int B, int n_src, int n_tar, int dtype, double *dist,
                int32_t *idx)
{
    TRY(use(c));
    REQUIRE(src && tar && dist && idx, "null pointer");
    REQUIRE(B > 0 && n_src > 0 && n_tar > 0, "sizes must be positive");
    REQUIRE(dtype_size(dtype), "unknown dtype");
    REQUIRE(n_tar <= 8192, "n_tar too large for the LDS-resident target (max 8192 points)");
    Timed t(c, SLAM_K_NN);
    HIPCHK(launch_nn(src, tar, B, n_src, n_tar, dtype, dist, idx, c->stream));
    return SLAM_OK;
}

int slam_nn(slam_ctx *c, const void *src, const void *tar, int B, int n_src, int n_tar, int dtype, double *dist,
            int32_t *idx)
{
    TRY(use(c));
    REQUIRE(src && tar && dist && idx, "null pointer");
    REQUIRE(B > 0 && n_src > 0 && n_tar > 0, "sizes must be positive");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    size_t bs = (size_t)B * 2 * n_src * ds, bt = (size_t)B * 2 * n_tar * ds, q = (size_t)B * n_src;
    TRY(arena_reserve(c, c->staging, align_up(bs) + align_up(bt) + align_up(q * 8) + align_up(q * 4) + 1024));
    char *d_s = carve<char>(c->staging, bs), *d_t = carve<char>(c->staging, bt);
    double *d_d = carve<double>(c->staging, q);
    int32_t *d_i = carve<int32_t>(c->staging, q);
    H2D(d_s, src, bs);
    H2D(d_t, tar, bt);
    TRY(slam_nn_dev(c, d_s, d_t, B, n_src, n_tar, dtype, d_d, d_i));
    D2H(dist, d_d, q * 8);
    D2H(idx, d_i, q * 4);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_kabsch2d_dev(slam_ctx *c, const double *src, const double *tar, int B, int n, double *T_out)
{
    TRY(use(c));
    REQUIRE(src && tar && T_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "sizes must be positive");
    Timed t(c, SLAM_K_KABSCH);
    HIPCHK(launch_kabsch(src, tar, B, n, T_out, c->stream));
    return SLAM_OK;
}

int slam_kabsch2d(slam_ctx *c, const double *src, const double *tar, int B, int n, double *T_out)
{
    TRY(use(c));
    REQUIRE(src && tar && T_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "sizes must be positive");
    size_t bp = (size_t)B * 2 * n * 8;
    TRY(arena_reserve(c, c->staging, 2 * align_up(bp) + align_up((size_t)B * 72) + 1024));
    double *d_s = carve<double>(c->staging, (size_t)B * 2 * n), *d_t = carve<double>(c->staging, (size_t)B * 2 * n);
    double *d_T = carve<double>(c->staging, (size_t)B * 9);
    H2D(d_s, src, bp);
    H2D(d_t, tar, bp);
    TRY(slam_kabsch2d_dev(c, d_s, d_t, B, n, d_T));
    D2H(T_out, d_T, (size_t)B * 72);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_icp_batch_dev(slam_ctx *c, const void *tar, const void *src, int B, int n_tar, int n_src, int dtype,
                       int tar_shared, int src_shared, const double *prior, int max_iter, double tol, double *T_out,
                       int32_t *iters_out, double *mean_err_out)
{
    TRY(use(c));
    REQUIRE(tar && src && T_out, "null pointer");
    REQUIRE(B > 0 && n_src > 0 && n_tar > 0, "sizes must be positive");
    REQUIRE(max_iter >= 0, "max_iter must be >= 0");
    REQUIRE(dtype_size(dtype), "unknown dtype");
    REQUIRE(n_src <= 8192, "n_src > 8192 not supported");
    REQUIRE(n_tar <= 8192, "n_tar too large for the LDS-resident target (max 8192 points)");
    IcpArgs a;
    a.tar = tar; a.src = src; a.prior = prior;
    a.ranges = nullptr; a.cos_t = a.sin_t = nullptr; a.tar_scan_stride = a.src_scan_stride = 0;
    a.tar_stride = tar_shared ? 0 : 2L * n_tar;
    a.src_stride = src_shared ? 0 : 2L * n_src;
    a.ppt = 0;
    a.B = B; a.n_tar = n_tar; a.n_src = n_src; a.max_iter = max_iter; a.tol = tol;
    a.T_out = T_out; a.iters_out = iters_out; a.err_out = mean_err_out;
    a.status = c->status; a.qpt_pref = c->icp_qpt; a.team_mode = c->icp_team;
    Timed t(c, SLAM_K_ICP);
    HIPCHK(launch_icp(a, dtype, c->stream));
    return SLAM_OK;
}

int slam_icp_batch(slam_ctx *c, const void *tar, const void *src, int B, int n_tar, int n_src, int dtype, int tar_shared,
                   int src_shared, const double *prior, int max_iter, double tol, double *T_out, int32_t *iters_out,
                   double *mean_err_out)
{
    TRY(use(c));
    REQUIRE(tar && src && T_out, "null pointer");
    REQUIRE(B > 0 && n_src > 0 && n_tar > 0, "sizes must be positive");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    size_t bt = (size_t)(tar_shared ? 1 : B) * 2 * n_tar * ds, bs = (size_t)(src_shared ? 1 : B) * 2 * n_src * ds;
    TRY(arena_reserve(c, c->staging, align_up(bt) + align_up(bs) + align_up((size_t)B * 48) + align_up((size_t)B * 72) +
                                         align_up((size_t)B * 4) + align_up((size_t)B * 8) + 2048));
    char *d_t = carve<char>(c->staging, bt), *d_s = carve<char>(c->staging, bs);
    double *d_p = prior ? carve<double>(c->staging, (size_t)B * 6) : nullptr;
    double *d_T = carve<double>(c->staging, (size_t)B * 9);
    int32_t *d_i = carve<int32_t>(c->staging, B);
    double *d_e = carve<double>(c->staging, B);
    TRY(copy_in(c, {{d_t, const_cast<void *>(tar), bt}, {d_s, const_cast<void *>(src), bs},
                    {d_p, const_cast<double *>(prior), prior ? (size_t)B * 48 : 0}}));
    TRY(slam_icp_batch_dev(c, d_t, d_s, B, n_tar, n_src, dtype, tar_shared, src_shared, d_p, max_iter, tol, d_T, d_i, d_e));
    return copy_out_sync(c, {{d_T, T_out, (size_t)B * 72}, {d_i, iters_out, (size_t)B * 4}, {d_e, mean_err_out, (size_t)B * 8}});
}

int slam_pose_compose_dev(slam_ctx *c, const double *T, const double *pose0, int L, int n, double *poses_out)
{
    TRY(use(c));
    REQUIRE(T && pose0 && poses_out, "null pointer");
    REQUIRE(L > 0 && n > 0, "sizes must be positive");
    Timed t(c, SLAM_K_COMPOSE);
    HIPCHK(launch_pose_compose(T, pose0, L, n, poses_out, c->stream));
    return SLAM_OK;
}

int slam_pose_compose(slam_ctx *c, const double *T, const double *pose0, int L, int n, double *poses_out)
{
    TRY(use(c));
    REQUIRE(T && pose0 && poses_out, "null pointer");
    REQUIRE(L > 0 && n > 0, "sizes must be positive");
    size_t nt = (size_t)L * n;
    TRY(arena_reserve(c, c->staging, align_up(nt * 72) + align_up((size_t)L * 24) + align_up(nt * 24) + 1024));
    double *d_T = carve<double>(c->staging, nt * 9), *d_0 = carve<double>(c->staging, (size_t)L * 3);
    double *d_P = carve<double>(c->staging, nt * 3);
    H2D(d_T, T, nt * 72);
    H2D(d_0, pose0, (size_t)L * 24);
    TRY(slam_pose_compose_dev(c, d_T, d_0, L, n, d_P));
    D2H(poses_out, d_P, nt * 24);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

/* ---- occupancy grid --------------------------------------------------------------- */

int slam_grid_create(slam_ctx *c, int G, int xw, int yw, double scale, double off_x, double off_y, double free_inc,
                     double hit_inc, double thresh, slam_grid **out)
{
    TRY(use(c));
    REQUIRE(out, "out is null");
    *out = nullptr;
    REQUIRE(G > 0 && xw > 0 && yw > 0, "G, xw, yw must be positive");
    REQUIRE(scale > 0 && std::isfinite(scale) && std::isfinite(off_x) && std::isfinite(off_y), "bad index rule");
    REQUIRE(free_inc > 0 && hit_inc > 0 && std::isfinite(thresh), "increments must be positive");
    // Occupied rule on the integer counters.  The reference keeps a float64 running sum per
    // cell (mapping.py:43,45) and tests it against thresh (:47).  With h hits and p passes the
    // sum is evaluated here in ONE canonical order, hits first, with the same sequential IEEE
    // adds: hit_levels = fewest hits that exceed thresh on their own (1 for +20; 3 for the +4
    // of w12-mapping-online), pass_thresh[h] = fewest passes that exceed it after h hits
    // (1001 for (0.01, 10), because 1000 sequential adds of 0.01 give 9.99999999999983).
    // For +20 the rule is exactly the reference's in any order.  For +4 the reference's own
    // answer depends on the order in which a cell's +4 and +0.01 arrived when p is exactly on
    // a threshold (p = pass_thresh[h] - 1 or pass_thresh[h]; the rounding of <= 1001 adds is
    // ~1e-13 << 0.01, so no other p is affected): those cells follow the canonical order.
    uint32_t levels = 0, table[kMaxHitLevels];
    {
        volatile double base = 0.0;
        while (levels < (uint32_t)kMaxHitLevels && !(base > thresh)) {
            volatile double acc = base;
            uint32_t k = 0;
            while (!(acc > thresh) && k < 0xfffffff0u) { acc = acc + free_inc; ++k; }
            table[levels++] = k;
            base = base + hit_inc;
        }
        REQUIRE(base > thresh, "hit_inc is too small: more than 8 hits would be needed to exceed thresh");
    }
    slam_grid *g = new slam_grid();
    size_t cells = (size_t)G * xw * yw;
    g->d.G = G; g->d.xw = xw; g->d.yw = yw;
    g->d.scale = scale; g->d.off_x = off_x; g->d.off_y = off_y;
    g->d.free_inc = free_inc; g->d.hit_inc = hit_inc;
    g->d.hit_levels = (int)levels;
    for (uint32_t k = 0; k < (uint32_t)kMaxHitLevels; ++k) g->d.pass_thresh[k] = k < levels ? table[k] : 0;
    g->d.status = c->status;
    // one allocation [pass | hit | visit counter] so that a reset is a single memset
    g->state_bytes = align_up(cells * 4) * 2 + align_up((size_t)kVisitSlots * kVisitStride * sizeof(unsigned long long));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&g->d.pass), g->state_bytes);
    if (e == hipSuccess) {
        g->d.hit = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(g->d.pass) + align_up(cells * 4));
        g->visits = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(g->d.pass) + 2 * align_up(cells * 4));
    }
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g->pmap_one), align_up((size_t)xw * yw));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&g->datamap_one), (size_t)xw * yw * 8);
    if (e != hipSuccess) {
        slam_grid_destroy(c, g);
        return fail(SLAM_ERR_NOMEM, "grid allocation (%zu cells): %s", cells, hipGetErrorString(e));
    }
    g->d.visits = g->visits;
    g->d.pmap_live = nullptr;
    g->d.redo = nullptr;
    g->d.live_dirty = &g->live_dirty;
    *out = g;
    return slam_grid_reset(c, g);
}

int slam_grid_destroy(slam_ctx *c, slam_grid *g)
{
    if (!g) return SLAM_OK;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->gstream) { (void)hipStreamSynchronize(c->cstream); (void)hipStreamSynchronize(c->gstream); }
        if (c->pstream) (void)hipStreamSynchronize(c->pstream);
    }
    if (g->pmap_live) (void)hipFree(g->pmap_live);
    if (g->d.redo) (void)hipFree(g->d.redo);
    if (g->d.pass) (void)hipFree(g->d.pass);   // also holds hit[] and the visit counter
    if (g->pmap_one) (void)hipFree(g->pmap_one);
    if (g->datamap_one) (void)hipFree(g->datamap_one);
    delete g;
    return SLAM_OK;
}

int slam_grid_reset(slam_ctx *c, slam_grid *g)
{
    TRY(use(c));
    REQUIRE(g, "null grid");
    TRY(join_particles(c));
    if (c->pipeline && c->mgrid) {
        TRY(fork_to_grid(c));
        c->mgrid = false;
    }
    HIPCHK(hipMemsetAsync(g->d.pass, 0, g->state_bytes, gs(c)));
    g->pristine = true;
    if (g->pmap_live) {
        HIPCHK(hipMemsetAsync(g->pmap_live, 50, (size_t)g->d.G * g->d.xw * g->d.yw, gs(c)));
        g->live_dirty = false;
    }
    if (c->pipeline) c->gdirty = true;
    return SLAM_OK;
}

int slam_grid_update_dev(slam_ctx *c, slam_grid *g, const double *ox, const double *oy, const double *cx, const double *cy,
                         int B, int n, const int32_t *grid_of_batch)
{
    TRY(use(c));
    TRY(grid_on_main(c));
    REQUIRE(g && ox && oy && cx && cy, "null pointer");
    REQUIRE(B > 0 && n > 0, "sizes must be positive");
    g->pristine = false;
    Timed t(c, SLAM_K_GRID);
    // (forced tiles / wedges keep the bound of the automatic choice on the beam count: ray numbers inside a group are 16-bit)
    const bool tiles_ok = !grid_of_batch && (tiles_apply(g->d, n, nullptr, 0) || ((c->grid_mode == 2 || c->grid_mode == 4) && n <= kTileMaxBeams));
    if ((c->grid_mode == 1 || c->grid_mode == 2 || c->grid_mode == 4) && tiles_ok) {
        size_t need = wedges_ok(c, g) ? wedge_scratch_bytes((long)B * n, B, B) : tile_scratch_bytes((long)B * n, B);
        if (need > c->tiles.cap) TRY(arena_reserve(c, c->tiles, need));
        HIPCHK(launch_grid_update_tiles_explicit(g->d, ox, oy, cx, cy, B, n, c->grid_group, c->tiles.base, c->stream, wedges_ok(c, g)));
    } else if (c->grid_mode != 0 && !grid_of_batch) {
        HIPCHK(launch_grid_update_win(g->d, ox, oy, cx, cy, B, n, c->grid_group, c->stream, c->grid_split));
    } else {
        HIPCHK(launch_grid_update(g->d, ox, oy, cx, cy, B, n, grid_of_batch, c->stream));
    }
    return SLAM_OK;
}

int slam_grid_update(slam_ctx *c, slam_grid *g, const double *ox, const double *oy, const double *cx, const double *cy,
                     int B, int n, const int32_t *grid_of_batch)
{
    TRY(use(c));
    REQUIRE(g && ox && oy && cx && cy, "null pointer");
    REQUIRE(B > 0 && n > 0, "sizes must be positive");
    if (grid_of_batch)
        for (int b = 0; b < B; ++b) REQUIRE(grid_of_batch[b] >= 0 && grid_of_batch[b] < g->d.G, "grid_of_batch out of range");
    size_t np = (size_t)B * n;
    TRY(arena_reserve(c, c->staging, 2 * align_up(np * 8) + 2 * align_up((size_t)B * 8) + align_up((size_t)B * 4) + 2048));
    double *d_x = carve<double>(c->staging, np), *d_y = carve<double>(c->staging, np);
    double *d_cx = carve<double>(c->staging, B), *d_cy = carve<double>(c->staging, B);
    int32_t *d_g = grid_of_batch ? carve<int32_t>(c->staging, B) : nullptr;
    TRY(copy_in(c, {{d_x, const_cast<double *>(ox), np * 8}, {d_y, const_cast<double *>(oy), np * 8},
                    {d_cx, const_cast<double *>(cx), (size_t)B * 8}, {d_cy, const_cast<double *>(cy), (size_t)B * 8},
                    {d_g, const_cast<int32_t *>(grid_of_batch), grid_of_batch ? (size_t)B * 4 : 0}}));
    TRY(slam_grid_update_dev(c, g, d_x, d_y, d_cx, d_cy, B, n, d_g));
    return check_status_sync(c);
}

// Ray cast of L streams x (n_scan - 1) scans into `g` on stream st, choosing the kernel:
// grid_mode 0 direct atomics, 1 automatic (LDS window; recorded walks + tiles for maps much
// larger than a window), 2 always tiles (where they apply), 3 always the window.
static int cast_replay(slam_ctx *c, slam_grid *g, const float *ranges, const double *cos_t, const double *sin_t,
                       const double *poses, const double *centres, int L, int n_scan, int n, const int32_t *got,
                       hipStream_t st)
{
    g->pristine = false;
    const bool tiles_ok = tiles_apply(g->d, n, got, 0, wedges_ok(c, g)) ||
                          ((c->grid_mode == 2 || c->grid_mode == 4) && (!got || (c->grid_mode == 4 && wedges_ok(c, g))) && n <= kTileMaxBeams);
    if ((c->grid_mode == 1 || c->grid_mode == 2 || c->grid_mode == 4) && tiles_ok) {
        long rays = (long)L * (n_scan - 1) * n, groups = (long)L * (n_scan - 1);
        // the wedges keep 6 bytes per ray (end cell, sorted ray number); the recorded walks of grid_mode 2 ~400
        size_t need = wedges_ok(c, g) ? wedge_scratch_bytes(rays, groups, groups) : tile_scratch_bytes(rays, groups);
        if (need > c->tiles.cap) {
            if (c->gstream) HIPCHK(hipStreamSynchronize(c->gstream));
            TRY(arena_reserve(c, c->tiles, need));
        }
        HIPCHK(launch_grid_update_tiles(g->d, ranges, cos_t, sin_t, poses, centres, L, n_scan, n, c->grid_group, c->tiles.base, st, wedges_ok(c, g), got));
        return SLAM_OK;
    }
    if (c->grid_mode != 0) {
        if (centres) HIPCHK(launch_grid_update_scans(g->d, ranges + n, cos_t, sin_t, poses, centres, n_scan - 1, n, c->grid_group, st, c->grid_split));
        else HIPCHK(launch_grid_update_replay_win(g->d, ranges, cos_t, sin_t, poses, L, n_scan, n, got, c->grid_group, st, 0, 0, nullptr, c->grid_split));
        return SLAM_OK;
    }
    REQUIRE(!centres, "grid_mode 0 has no separate ray origins");
    HIPCHK(launch_grid_update_replay(g->d, ranges, cos_t, sin_t, poses, L, n_scan, n, got, st));
    return SLAM_OK;
}

int slam_grid_update_scans_dev(slam_ctx *c, slam_grid *g, const float *ranges, const double *cos_t, const double *sin_t,
                               const double *poses, const double *centres, int S, int n)
{
    TRY(use(c));
    TRY(grid_on_main(c));
    REQUIRE(g && ranges && cos_t && sin_t && poses, "null pointer");
    REQUIRE(S > 0 && n > 0 && n <= 65535, "bad sizes");
    Timed t(c, SLAM_K_GRID);
    // scan k of cast_replay's stream is ranges[k + 1]: shift the base by one scan
    return cast_replay(c, g, ranges - n, cos_t, sin_t, poses, centres, 1, S + 1, n, nullptr, c->stream);
}

int slam_grid_update_scans(slam_ctx *c, slam_grid *g, const float *ranges, const double *cos_t, const double *sin_t,
                           const double *poses, const double *centres, int S, int n)
{
    TRY(use(c));
    REQUIRE(g && ranges && cos_t && sin_t && poses, "null pointer");
    REQUIRE(S > 0 && n > 0 && n <= 65535, "bad sizes");
    size_t nr = (size_t)S * n;
    TRY(arena_reserve(c, c->staging, align_up(nr * 4) + 2 * align_up((size_t)n * 8) + align_up((size_t)S * 24) +
                                         align_up((size_t)S * 16) + 2048));
    float *d_r = carve<float>(c->staging, nr);
    double *d_c = carve<double>(c->staging, n), *d_s = carve<double>(c->staging, n);
    double *d_p = carve<double>(c->staging, (size_t)S * 3);
    double *d_o = centres ? carve<double>(c->staging, (size_t)S * 2) : nullptr;
    H2D(d_r, ranges, nr * 4);
    H2D(d_c, cos_t, (size_t)n * 8);
    H2D(d_s, sin_t, (size_t)n * 8);
    H2D(d_p, poses, (size_t)S * 24);
    if (centres) H2D(d_o, centres, (size_t)S * 16);
    TRY(slam_grid_update_scans_dev(c, g, d_r, d_c, d_s, d_p, d_o, S, n));
    return check_status_sync(c);
}

static int refresh_live(slam_ctx *c, slam_grid *g, hipStream_t st)
{
    if (!g->pmap_live || !g->live_dirty) return SLAM_OK;
    Timed t(c, SLAM_K_FINALIZE, st);
    HIPCHK(launch_grid_finalize(g->d, 0, g->d.G, g->pmap_live, st));
    g->live_dirty = false;
    return SLAM_OK;
}

int slam_grid_live_pmap(slam_ctx *c, slam_grid *g, int8_t **pmap_dev_out)
{
    TRY(use(c));
    REQUIRE(g && pmap_dev_out, "null pointer");
    TRY(grid_on_main(c));
    if (!g->pmap_live) {
        size_t bytes = (size_t)g->d.G * g->d.xw * g->d.yw;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&g->pmap_live), align_up(bytes));
        if (e != hipSuccess) {
            g->pmap_live = nullptr;
            return fail(SLAM_ERR_NOMEM, "live pmap (%zu bytes): %s", bytes, hipGetErrorString(e));
        }
        g->d.pmap_live = g->pmap_live;
        if (hipMalloc(reinterpret_cast<void **>(&g->d.redo), ((size_t)g->d.G + 2) * sizeof(int32_t)) != hipSuccess) {
            g->d.redo = nullptr;                  // (the owner kernels then run without the byte-window fast path)
            (void)hipGetLastError();
        } else {
            HIPCHK(hipMemsetAsync(g->d.redo, 0, ((size_t)g->d.G + 2) * sizeof(int32_t), c->stream));
        }
        if (g->pristine) {                    // untouched maps: every cell is 50 (no finalize pass over G maps of zeros)
            HIPCHK(hipMemsetAsync(g->pmap_live, 50, bytes, c->stream));
            g->live_dirty = false;
        } else {
            g->live_dirty = true;
            TRY(refresh_live(c, g, c->stream));
        }
    }
    *pmap_dev_out = g->pmap_live;
    return SLAM_OK;
}

int slam_grid_finalize_dev(slam_ctx *c, slam_grid *g, int8_t *pmap_dev)
{
    TRY(use(c));
    REQUIRE(g && pmap_dev, "null pointer");
    TRY(join_particles(c));        // (a chunked particle batch casts on a stream of its own)
    TRY(fork_to_grid(c));          // pmap_dev may still be in use by work on the main stream
    if (g->pmap_live) {            // kept current by the ray casts; refresh only if something bypassed that
        TRY(refresh_live(c, g, gs(c)));
        if (pmap_dev != g->pmap_live)
            HIPCHK(hipMemcpyAsync(pmap_dev, g->pmap_live, (size_t)g->d.G * g->d.xw * g->d.yw, hipMemcpyDeviceToDevice, gs(c)));
        return SLAM_OK;
    }
    Timed t(c, SLAM_K_FINALIZE, gs(c));
    HIPCHK(launch_grid_finalize(g->d, 0, g->d.G, pmap_dev, gs(c)));
    return SLAM_OK;
}

int slam_grid_read(slam_ctx *c, slam_grid *g, int gi, int8_t *pmap, double *datamap, uint32_t *pass, uint32_t *hit)
{
    TRY(use(c));
    TRY(grid_on_main(c));
    REQUIRE(g, "null grid");
    REQUIRE(gi >= 0 && gi < g->d.G, "grid index out of range");
    size_t per = (size_t)g->d.xw * g->d.yw;
    if (pmap && g->pmap_live) {
        TRY(refresh_live(c, g, c->stream));
        D2H(pmap, g->pmap_live + per * gi, per);
    } else if (pmap) {
        {
            Timed t(c, SLAM_K_FINALIZE);
            HIPCHK(launch_grid_finalize(g->d, gi, 1, g->pmap_one, c->stream));
        }
        D2H(pmap, g->pmap_one, per);
    }
    if (datamap) {
        HIPCHK(launch_grid_datamap(g->d, gi, g->datamap_one, c->stream));
        D2H(datamap, g->datamap_one, per * 8);
    }
    if (pass) D2H(pass, g->d.pass + per * gi, per * 4);
    if (hit) D2H(hit, g->d.hit + per * gi, per * 4);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_grid_counters_dev(slam_ctx *c, slam_grid *g, uint32_t **pass_dev, uint32_t **hit_dev)
{
    TRY(use(c));
    REQUIRE(g && pass_dev && hit_dev, "null pointer");
    TRY(grid_on_main(c));                 // later work on the context's stream sees every update so far
    if (g->pmap_live) g->live_dirty = true;   // the caller may change the counters behind the library's back
    g->pristine = false;
    *pass_dev = g->d.pass;
    *hit_dev = g->d.hit;
    return SLAM_OK;
}

int slam_grid_occupancy_data(slam_ctx *c, slam_grid *g, int gi, int8_t *data)
{
    TRY(use(c));
    TRY(grid_on_main(c));
    REQUIRE(g && data, "null pointer");
    REQUIRE(gi >= 0 && gi < g->d.G, "grid index out of range");
    size_t per = (size_t)g->d.xw * g->d.yw;
    TRY(arena_reserve(c, c->scratch, align_up(per) + 1024));
    int8_t *d_data = carve<int8_t>(c->scratch, per);
    const int8_t *pm = g->pmap_one;
    if (g->pmap_live) {                       // kept current by the ray casts: no finalize pass
        TRY(refresh_live(c, g, c->stream));
        pm = g->pmap_live + per * gi;
    } else {
        HIPCHK(launch_grid_finalize(g->d, gi, 1, g->pmap_one, c->stream));
    }
    HIPCHK(launch_grid_transpose(pm, g->d.xw, g->d.yw, d_data, c->stream));
    D2H(data, d_data, per);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_grid_visits(slam_ctx *c, slam_grid *g, uint64_t *visits_out)
{
    TRY(use(c));
    TRY(grid_on_main(c));
    REQUIRE(g && visits_out, "null pointer");
    static thread_local unsigned long long slots[kVisitSlots * kVisitStride];
    D2H(slots, g->visits, sizeof slots);
    HIPCHK(hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    for (int k = 0; k < kVisitSlots; ++k) v += slots[(size_t)k * kVisitStride];
    *visits_out = v;
    return SLAM_OK;
}

int slam_bresenham_batch(slam_ctx *c, const int32_t *starts, const int32_t *ends, int B, const int64_t *offsets,
                         int32_t *lens_out, int32_t *cells_out, int64_t total_cells)
{
    TRY(use(c));
    REQUIRE(starts && ends && lens_out, "null pointer");
    REQUIRE(B > 0, "B must be positive");
    REQUIRE(!cells_out || (offsets && total_cells >= 0), "cells_out needs offsets and total_cells");
    for (int b = 0; b < B; ++b) {
        int64_t dx = (int64_t)ends[2 * b] - starts[2 * b], dy = (int64_t)ends[2 * b + 1] - starts[2 * b + 1];
        int64_t len = std::max(std::llabs(dx), std::llabs(dy)) + 1;
        if (len > 2 * (int64_t)kMaxRayCells) return fail(SLAM_ERR_OVERFLOW, "line %d longer than 2^21 cells", b);
        if (cells_out && (offsets[b] < 0 || offsets[b] + len > total_cells))
            return fail(SLAM_ERR_INVALID, "line %d does not fit cells_out", b);
    }
    size_t tc = cells_out ? (size_t)total_cells : 0;
    TRY(arena_reserve(c, c->staging, 2 * align_up((size_t)B * 8) + align_up((size_t)B * 8) + align_up((size_t)B * 4) +
                                         align_up(tc * 8) + 2048));
    int32_t *d_s = carve<int32_t>(c->staging, (size_t)B * 2), *d_e = carve<int32_t>(c->staging, (size_t)B * 2);
    int64_t *d_o = carve<int64_t>(c->staging, B);
    int32_t *d_l = carve<int32_t>(c->staging, B);
    int32_t *d_c = cells_out ? carve<int32_t>(c->staging, tc * 2) : nullptr;
    H2D(d_s, starts, (size_t)B * 8);
    H2D(d_e, ends, (size_t)B * 8);
    if (cells_out) H2D(d_o, offsets, (size_t)B * 8);
    {
        Timed t(c, SLAM_K_BRESENHAM);
        HIPCHK(launch_bresenham(d_s, d_e, B, d_o, d_l, d_c, c->stream));
    }
    D2H(lens_out, d_l, (size_t)B * 4);
    if (cells_out && tc) D2H(cells_out, d_c, tc * 8);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

/* ---- fused replay ------------------------------------------------------------------ */

int slam_replay_dev(slam_ctx *c, const float *ranges, const double *cos_t, const double *sin_t, int L, int n_scan, int n,
                    int dtype, int max_iter, double tol, const double *pose0, slam_grid *grid, const int32_t *grid_of_traj,
                    void *pts_ws, double *poses_out, double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pose0 && poses_out, "null pointer");
    (void)pts_ws;   // kept in the ABI; the point buffers are no longer materialised
    REQUIRE(L > 0 && n_scan >= 2 && n > 0, "need L > 0, n_scan >= 2, n > 0");
    REQUIRE(max_iter >= 0, "max_iter must be >= 0");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    REQUIRE(n <= 8192, "n too large (max 8192 beams)");
    const long pairs = (long)L * (n_scan - 1);
    REQUIRE(pairs < (1L << 31), "too many scan pairs for one launch");
    const bool piped = c->pipeline && grid;     // three stages on three streams: ICP | compose | map
    double *T = T_out;
    if (!T) {
        if (piped) {                            // two alternating buffers: compose of replay k reads T while ICP k+1 runs
            size_t need = align_up((size_t)pairs * 72);
            if (need > c->pipe_T_bytes) {
                TRY(join_from_grid(c));
                HIPCHK(hipStreamSynchronize(c->stream));
                for (double *&t : c->pipe_T) {
                    if (t) HIPCHK(hipFree(t));
                    t = nullptr;
                    if (hipMalloc(reinterpret_cast<void **>(&t), need) != hipSuccess)
                        return fail(SLAM_ERR_NOMEM, "hipMalloc(%zu bytes) for T", need);
                }
                c->pipe_T_bytes = need;
            }
            T = c->pipe_T[c->cast_pos];
        } else {
            TRY(arena_reserve(c, c->scratch, align_up((size_t)pairs * 72) + 1024));
            T = carve<double>(c->scratch, (size_t)pairs * 9);
        }
    }
    if (c->pipeline)        // is an earlier replay's compose / ray cast still reading these buffers?
        for (int k = 0; k < 2; ++k)
            if (c->cast_poses[k] && (c->cast_poses[k] == poses_out || c->cast_T[k] == T)) {
                HIPCHK(hipStreamWaitEvent(c->stream, c->ev_cast[k], 0));
                HIPCHK(hipStreamWaitEvent(c->cstream, c->ev_cast[k], 0));
                c->cast_poses[k] = c->cast_T[k] = nullptr;
            }
    {
        // polar -> Cartesian is fused into the ICP kernel: scan k-1 / k of stream l are read as raw
        // ranges and turned into points (of storage type `dtype`) in registers
        IcpArgs a;
        a.tar = a.src = nullptr; a.prior = nullptr;
        a.ranges = ranges; a.cos_t = cos_t; a.sin_t = sin_t;
        a.tar_scan_stride = a.src_scan_stride = n;
        a.tar_stride = a.src_stride = 0;
        a.ppt = n_scan - 1;
        a.B = (int)pairs; a.n_tar = n; a.n_src = n; a.max_iter = max_iter; a.tol = tol;
        a.T_out = T; a.iters_out = iters_out; a.err_out = nullptr;
        a.status = c->status; a.qpt_pref = c->icp_qpt; a.team_mode = c->icp_team;
        if (grid && c->replay_reset) {
            // option "replay_reset": the map starts from zero for this replay.  On one stream the scan-matching launch
            // clears the counters on its way (they are next touched by the ray cast behind it); maps with a live pmap and
            // the three-stream pipeline go through slam_grid_reset.
            if (piped || grid->pmap_live) {
                TRY(slam_grid_reset(c, grid));
            } else {
                a.zero_ptr = grid->d.pass; a.zero_bytes = grid->state_bytes;
                grid->pristine = true;
            }
        }
        Timed t(c, SLAM_K_ICP);
        HIPCHK(launch_icp(a, dtype, c->stream));
    }
    if (!piped) {
        {
            Timed t(c, SLAM_K_COMPOSE);
            HIPCHK(launch_pose_compose(T, pose0, L, n_scan - 1, poses_out, c->stream));
        }
        if (grid) {
            TRY(grid_on_main(c));
            Timed t(c, SLAM_K_GRID);
            TRY(cast_replay(c, grid, ranges, cos_t, sin_t, poses_out, nullptr, L, n_scan, n, grid_of_traj, c->stream));
        }
        return SLAM_OK;
    }
    // compose waits for the ICP (and everything before it on the main stream: pose0, ranges)
    HIPCHK(hipEventRecord(c->ev_icp, c->stream));
    HIPCHK(hipStreamWaitEvent(c->cstream, c->ev_icp, 0));
    {
        Timed t(c, SLAM_K_COMPOSE, c->cstream);
        HIPCHK(launch_pose_compose(T, pose0, L, n_scan - 1, poses_out, c->cstream));
    }
    HIPCHK(hipEventRecord(c->ev_pose, c->cstream));
    HIPCHK(hipStreamWaitEvent(c->gstream, c->ev_pose, 0));
    c->gdirty = true;
    c->mgrid = false;
    {
        Timed t(c, SLAM_K_GRID, c->gstream);
        TRY(cast_replay(c, grid, ranges, cos_t, sin_t, poses_out, nullptr, L, n_scan, n, grid_of_traj, c->gstream));
    }
    HIPCHK(hipEventRecord(c->ev_cast[c->cast_pos], c->gstream));
    c->cast_poses[c->cast_pos] = poses_out;
    c->cast_T[c->cast_pos] = T;
    c->cast_pos ^= 1;
    return SLAM_OK;
}

int slam_replay(slam_ctx *c, const float *ranges, const double *cos_t, const double *sin_t, int L, int n_scan, int n,
                int dtype, int max_iter, double tol, const double *pose0, slam_grid *grid, const int32_t *grid_of_traj,
                double *poses_out, double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pose0 && poses_out, "null pointer");
    REQUIRE(L > 0 && n_scan >= 2 && n > 0, "need L > 0, n_scan >= 2, n > 0");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    if (grid && grid_of_traj)
        for (int l = 0; l < L; ++l) REQUIRE(grid_of_traj[l] >= 0 && grid_of_traj[l] < grid->d.G, "grid_of_traj out of range");
    size_t nr = (size_t)L * n_scan * n, pairs = (size_t)L * (n_scan - 1);
    TRY(arena_reserve(c, c->staging, align_up(nr * 4) + 2 * align_up((size_t)n * 8) + align_up((size_t)L * 24) +
                                         align_up(pairs * 72) + align_up(pairs * 24) + align_up(pairs * 4) +
                                         align_up((size_t)L * 4) + 4096));
    float *d_r = carve<float>(c->staging, nr);
    double *d_c = carve<double>(c->staging, n), *d_s = carve<double>(c->staging, n);
    double *d_0 = carve<double>(c->staging, (size_t)L * 3);
    int32_t *d_g = (grid && grid_of_traj) ? carve<int32_t>(c->staging, L) : nullptr;
    double *d_T = carve<double>(c->staging, pairs * 9), *d_P = carve<double>(c->staging, pairs * 3);
    int32_t *d_it = carve<int32_t>(c->staging, pairs);
    TRY(copy_in(c, {{d_r, const_cast<float *>(ranges), nr * 4}, {d_c, const_cast<double *>(cos_t), (size_t)n * 8},
                    {d_s, const_cast<double *>(sin_t), (size_t)n * 8}, {d_0, const_cast<double *>(pose0), (size_t)L * 24},
                    {d_g, const_cast<int32_t *>(grid_of_traj), d_g ? (size_t)L * 4 : 0}}));
    TRY(slam_replay_dev(c, d_r, d_c, d_s, L, n_scan, n, dtype, max_iter, tol, d_0, grid, d_g, nullptr, d_P, d_T, d_it));
    // "pipeline" option with a map: compose ran on cstream and the ray cast on gstream (which
    // waited for compose): the copies below are issued on the main stream, which must first see
    // both (join_from_grid records on gstream, i.e. after ev_pose of cstream).
    TRY(join_from_grid(c));
    TRY(copy_out_sync(c, {{d_P, poses_out, pairs * 24}, {d_T, T_out, pairs * 72}, {d_it, iters_out, pairs * 4}}));
    return check_status_sync(c);
}

/* ---- particle hypotheses ----------------------------------------------------------- */

int slam_particles_dev(slam_ctx *c, const float *ranges2, const double *cos_t, const double *sin_t, int n, int dtype,
                       const double *prior, const double *pose_prev, int P, int max_iter, double tol, slam_grid *grid,
                       void *pts_ws, double *poses_out, double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE(ranges2 && cos_t && sin_t && pose_prev && poses_out && T_out, "null pointer");
    (void)pts_ws;   // kept in the ABI; the point buffers are no longer materialised
    REQUIRE(P > 0 && n > 0 && n <= 8192, "need P > 0 and 0 < n <= 8192");
    REQUIRE(max_iter >= 0, "max_iter must be >= 0");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    REQUIRE(!grid || grid->d.G >= P, "the grid object needs one map per particle");
    REQUIRE(!grid || P <= 65535, "at most 65535 particles per call when ray casting");
    // Chunks (option "particle_chunks", off by default): matcher and pose step of chunk k on the context's stream, its ray
    // cast on a second stream behind an event, beside the matcher of chunk k + 1.  The matcher is bound by vector issue and
    // the ray cast by memory traffic, so the two should share the chip well - they do not (measured, round 4: 10 000
    // hypotheses 1.14 ms per step in one piece, 1.19 / 1.35 / 1.61 in 2 / 4 / 8 chunks): a ray-cast workgroup holds half a
    // CU's LDS for its window and the matcher's workgroups a quarter of its registers each, both for their whole lives, so
    // workgroups of the two kernels displace each other on a CU instead of filling each other's stalls.
    // Kept as an option because the chunks are also the unit a caller could overlap with work of its own; hypotheses are
    // independent (ICP.process / Mapping.update per hypothesis), so chunking changes no result.
    int chunks = c->particle_chunks > 1 ? c->particle_chunks : 1;
    if (!grid || chunks > P) chunks = 1;
    if (chunks == 1) TRY(grid_on_main(c));
    else if (c->pipeline) TRY(join_from_grid(c));
    if (chunks > 1) {
        if (!c->pstream) {
            HIPCHK(hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->pev_join, hipEventDisableTiming));
        }
        while ((int)c->pev_pose.size() < chunks) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIPCHK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            c->pev_pose.push_back(e0);
            c->pev_cast.push_back(e1);
        }
    }
    double *heading_cs = nullptr;               // cos / sin of the new headings: written by the pose step, read by the ray cast
    if (grid) {
        const size_t need = align_up((size_t)P * 16) + 1024;
        TRY(arena_reserve(c, c->scratch, need));
        heading_cs = carve<double>(c->scratch, (size_t)P * 2);
        grid->pristine = false;
    }
    for (int k = 0; k < chunks; ++k) {
        const int p0 = (int)((long)P * k / chunks), pc = (int)((long)P * (k + 1) / chunks) - p0;
        {
            IcpArgs a;
            a.tar = a.src = nullptr; a.prior = prior ? prior + 6 * (size_t)p0 : nullptr;
            a.ranges = ranges2; a.cos_t = cos_t; a.sin_t = sin_t;
            a.tar_scan_stride = a.src_scan_stride = 0;   // every hypothesis matches the same scan pair
            a.tar_stride = a.src_stride = 0;
            a.ppt = 0;
            a.B = pc; a.n_tar = n; a.n_src = n; a.max_iter = max_iter; a.tol = tol;
            a.T_out = T_out + 9 * (size_t)p0; a.iters_out = iters_out ? iters_out + p0 : nullptr; a.err_out = nullptr;
            a.status = c->status; a.team_mode = c->icp_team;
            // (queries per lane by the size of the BATCH: its chunks share the chip with one another's ray casts)
            a.qpt_pref = c->icp_qpt > 0 ? c->icp_qpt : (P >= 2500 ? 3 : 0);
            Timed t(c, SLAM_K_ICP, nullptr, k == 0);
            HIPCHK(launch_icp(a, dtype, c->stream));
        }
        {
            Timed t(c, SLAM_K_COMPOSE, nullptr, k == 0);
            HIPCHK(launch_pose_compose(T_out + 9 * (size_t)p0, pose_prev + 3 * (size_t)p0, pc, 1, poses_out + 3 * (size_t)p0, c->stream,
                                       prior ? prior + 6 * (size_t)p0 : nullptr, heading_cs ? heading_cs + 2 * (size_t)p0 : nullptr));
        }
        if (grid) {
            hipStream_t st = c->stream;
            if (chunks > 1) {
                HIPCHK(hipEventRecord(c->pev_pose[k], c->stream));
                HIPCHK(hipStreamWaitEvent(c->pstream, c->pev_pose[k], 0));
                st = c->pstream;
            }
            GridDev d = grid->d;                                       // the chunk's maps
            const size_t off = (size_t)p0 * d.xw * d.yw;
            d.G = pc; d.pass += off; d.hit += off;
            if (d.pmap_live) d.pmap_live += off;
            Timed t(c, SLAM_K_GRID, nullptr, k == 0);
            HIPCHK(launch_grid_update_replay_win(d, ranges2, cos_t, sin_t, poses_out + 3 * (size_t)p0, pc, 2, n, nullptr, 1, st,
                                                 /*shared_scans=*/1, /*grid_per_traj=*/1, heading_cs + 2 * (size_t)p0));
            if (chunks > 1) HIPCHK(hipEventRecord(c->pev_cast[k], c->pstream));
        }
    }
    if (chunks > 1) {
        // The ray casts on the second stream read the caller's ranges2 / cos_t / sin_t / poses_out and the heading scratch: the
        // context's stream waits for them here, so that - as for every *_dev entry point - work enqueued behind this call is
        // ordered behind ALL of it (a caller overwriting ranges2 for the next scan pair, slam_replay_dev carving
        // the same scratch).  The overlap this option is for is the one INSIDE a batch: cast of chunk k beside the matcher of k + 1.
        c->pdirty = true;
        TRY(join_particles(c));
        if (c->pipeline) c->mgrid = true;      // map work the map stream has not seen: its next piece forks from the main stream
    }
    return SLAM_OK;
}

int slam_particles(slam_ctx *c, const float *ranges2, const double *cos_t, const double *sin_t, int n, int dtype,
                   const double *prior, const double *pose_prev, int P, int max_iter, double tol, slam_grid *grid,
                   double *poses_out, double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE(ranges2 && cos_t && sin_t && pose_prev && poses_out, "null pointer");
    REQUIRE(P > 0 && n > 0, "sizes must be positive");
    size_t ds = dtype_size(dtype);
    REQUIRE(ds, "unknown dtype");
    TRY(arena_reserve(c, c->staging, align_up((size_t)2 * n * 4) + 2 * align_up((size_t)n * 8) + align_up((size_t)P * 48) +
                                         align_up((size_t)P * 24) * 2 + align_up(4 * (size_t)n * ds) + align_up((size_t)P * 72) +
                                         align_up((size_t)P * 4) + 4096));
    float *d_r = carve<float>(c->staging, 2 * (size_t)n);
    double *d_c = carve<double>(c->staging, n), *d_s = carve<double>(c->staging, n);
    double *d_pr = prior ? carve<double>(c->staging, (size_t)P * 6) : nullptr;
    double *d_p0 = carve<double>(c->staging, (size_t)P * 3), *d_P = carve<double>(c->staging, (size_t)P * 3);
    char *d_pts = carve<char>(c->staging, 4 * (size_t)n * ds);
    double *d_T = carve<double>(c->staging, (size_t)P * 9);
    int32_t *d_it = carve<int32_t>(c->staging, P);
    H2D(d_r, ranges2, 2 * (size_t)n * 4);
    H2D(d_c, cos_t, (size_t)n * 8);
    H2D(d_s, sin_t, (size_t)n * 8);
    if (prior) H2D(d_pr, prior, (size_t)P * 48);
    H2D(d_p0, pose_prev, (size_t)P * 24);
    TRY(slam_particles_dev(c, d_r, d_c, d_s, n, dtype, d_pr, d_p0, P, max_iter, tol, grid, d_pts, d_P, d_T, d_it));
    D2H(poses_out, d_P, (size_t)P * 24);
    if (T_out) D2H(T_out, d_T, (size_t)P * 72);
    if (iters_out) D2H(iters_out, d_it, (size_t)P * 4);
    return check_status_sync(c);
}

/* ---- scan-to-map observation ------------------------------------- */

int slam_map_obstacles(slam_ctx *c, const int8_t *map, int width, int height, int wire_layout, double resolution,
                       double origin_x, double origin_y, double *ox, double *oy, int cap, int *count_out)
{
    TRY(use(c));
    REQUIRE(map && ox && oy && count_out, "null pointer");
    REQUIRE(width > 0 && height > 0 && cap >= 0, "bad sizes");
    size_t cells = (size_t)width * height;
    TRY(arena_reserve(c, c->staging, align_up(cells) + 2 * align_up((size_t)cap * 8) + 1024));
    int8_t *d_m = carve<int8_t>(c->staging, cells);
    double *d_x = carve<double>(c->staging, cap), *d_y = carve<double>(c->staging, cap);
    int *d_k = carve<int>(c->staging, 1);
    H2D(d_m, map, cells);
    HIPCHK(launch_map_obstacles(d_m, width, height, wire_layout, resolution, origin_x, origin_y, d_x, d_y, cap, d_k, c->stream));
    int k = 0;
    D2H(&k, d_k, sizeof(int));
    HIPCHK(hipStreamSynchronize(c->stream));
    *count_out = k;
    int got = k < cap ? k : cap;
    if (got > 0) {
        D2H(ox, d_x, (size_t)got * 8);
        D2H(oy, d_y, (size_t)got * 8);
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return SLAM_OK;
}

int slam_map_obstacles_dev(slam_ctx *c, const int8_t *map, int width, int height, int wire_layout, double resolution,
                           double origin_x, double origin_y, double *ox, double *oy, int cap, int *count_dev)
{
    TRY(use(c));
    REQUIRE(map && ox && oy && count_dev, "null pointer");
    REQUIRE(width > 0 && height > 0 && cap >= 0, "bad sizes");
    HIPCHK(launch_map_obstacles(map, width, height, wire_layout, resolution, origin_x, origin_y, ox, oy, cap, count_dev, c->stream));
    return SLAM_OK;
}

int slam_virtual_scan_dev(slam_ctx *c, const double *ox, const double *oy, int K, const double *poses, int B,
                          double angle_min, double angle_increment, int n, double *ranges_out)
{
    TRY(use(c));
    REQUIRE((K == 0 || (ox && oy)) && poses && ranges_out, "null pointer");
    REQUIRE(K >= 0 && B > 0 && n > 0 && n <= 8192 && B <= 65535, "bad sizes");
    REQUIRE(angle_increment != 0.0 && std::isfinite(angle_increment) && std::isfinite(angle_min), "bad angles");
    HIPCHK(launch_virtual_scan(ox, oy, K, poses, B, angle_min, angle_increment, n, ranges_out, c->stream));
    return SLAM_OK;
}

int slam_virtual_scan(slam_ctx *c, const double *ox, const double *oy, int K, const double *poses, int B, double angle_min,
                      double angle_increment, int n, double *ranges_out)
{
    TRY(use(c));
    REQUIRE((K == 0 || (ox && oy)) && poses && ranges_out, "null pointer");
    REQUIRE(K >= 0 && B > 0 && n > 0, "bad sizes");
    TRY(arena_reserve(c, c->staging, 2 * align_up((size_t)K * 8) + align_up((size_t)B * 24) + align_up((size_t)B * n * 8) + 2048));
    double *d_x = carve<double>(c->staging, K), *d_y = carve<double>(c->staging, K);
    double *d_p = carve<double>(c->staging, (size_t)B * 3), *d_r = carve<double>(c->staging, (size_t)B * n);
    if (K) { H2D(d_x, ox, (size_t)K * 8); H2D(d_y, oy, (size_t)K * 8); }
    H2D(d_p, poses, (size_t)B * 24);
    TRY(slam_virtual_scan_dev(c, d_x, d_y, K, d_p, B, angle_min, angle_increment, n, d_r));
    D2H(ranges_out, d_r, (size_t)B * n * 8);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_scan_to_points_f64_dev(slam_ctx *c, const double *ranges, const double *cos_t, const double *sin_t, int B, int n,
                                double *pts_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pts_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "bad sizes");
    HIPCHK(launch_ranges64_to_points(ranges, cos_t, sin_t, B, n, pts_out, c->stream));
    return SLAM_OK;
}

int slam_scan_to_points_f64(slam_ctx *c, const double *ranges, const double *cos_t, const double *sin_t, int B, int n,
                            double *pts_out)
{
    TRY(use(c));
    REQUIRE(ranges && cos_t && sin_t && pts_out, "null pointer");
    REQUIRE(B > 0 && n > 0, "bad sizes");
    TRY(arena_reserve(c, c->staging, align_up((size_t)B * n * 8) + 2 * align_up((size_t)n * 8) + align_up((size_t)B * 2 * n * 8) + 2048));
    double *d_r = carve<double>(c->staging, (size_t)B * n);
    double *d_c = carve<double>(c->staging, n), *d_s = carve<double>(c->staging, n);
    double *d_p = carve<double>(c->staging, (size_t)B * 2 * n);
    H2D(d_r, ranges, (size_t)B * n * 8);
    H2D(d_c, cos_t, (size_t)n * 8);
    H2D(d_s, sin_t, (size_t)n * 8);
    TRY(slam_scan_to_points_f64_dev(c, d_r, d_c, d_s, B, n, d_p));
    D2H(pts_out, d_p, (size_t)B * 2 * n * 8);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

int slam_map_observation_dev(slam_ctx *c, const double *ox, const double *oy, int K, const double *poses, const double *src,
                             int B, int n, int src_shared, const double *cos_t, const double *sin_t, double angle_min,
                             double angle_increment, int max_iter, double tol, double *vranges_ws, double *vpts_ws,
                             double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE(poses && src && cos_t && sin_t && vranges_ws && vpts_ws && T_out, "null pointer");
    REQUIRE(B > 0 && n > 0 && n <= 8192, "bad sizes");
    TRY(slam_virtual_scan_dev(c, ox, oy, K, poses, B, angle_min, angle_increment, n, vranges_ws));
    HIPCHK(launch_ranges64_to_points(vranges_ws, cos_t, sin_t, B, n, vpts_ws, c->stream));
    return slam_icp_batch_dev(c, vpts_ws, src, B, n, n, SLAM_F64, 0, src_shared, nullptr, max_iter, tol, T_out, iters_out,
                              nullptr);
}

int slam_map_observation(slam_ctx *c, const double *ox, const double *oy, int K, const double *poses, const double *src,
                         int B, int n, int src_shared, const double *cos_t, const double *sin_t, double angle_min,
                         double angle_increment, int max_iter, double tol, double *T_out, int32_t *iters_out)
{
    TRY(use(c));
    REQUIRE((K == 0 || (ox && oy)) && poses && src && cos_t && sin_t && T_out, "null pointer");
    REQUIRE(K >= 0 && B > 0 && n > 0, "bad sizes");
    size_t bs = (size_t)(src_shared ? 1 : B) * 2 * n * 8;
    TRY(arena_reserve(c, c->staging, 2 * align_up((size_t)K * 8) + align_up((size_t)B * 24) + align_up(bs) + 2 * align_up((size_t)n * 8) +
                                         align_up((size_t)B * n * 8) + align_up((size_t)B * 2 * n * 8) + align_up((size_t)B * 72) +
                                         align_up((size_t)B * 4) + 4096));
    double *d_x = carve<double>(c->staging, K), *d_y = carve<double>(c->staging, K);
    double *d_p = carve<double>(c->staging, (size_t)B * 3);
    double *d_s = carve<double>(c->staging, bs / 8);
    double *d_c = carve<double>(c->staging, n), *d_sn = carve<double>(c->staging, n);
    double *d_vr = carve<double>(c->staging, (size_t)B * n), *d_vp = carve<double>(c->staging, (size_t)B * 2 * n);
    double *d_T = carve<double>(c->staging, (size_t)B * 9);
    int32_t *d_it = carve<int32_t>(c->staging, B);
    if (K) { H2D(d_x, ox, (size_t)K * 8); H2D(d_y, oy, (size_t)K * 8); }
    H2D(d_p, poses, (size_t)B * 24);
    H2D(d_s, src, bs);
    H2D(d_c, cos_t, (size_t)n * 8);
    H2D(d_sn, sin_t, (size_t)n * 8);
    TRY(slam_map_observation_dev(c, d_x, d_y, K, d_p, d_s, B, n, src_shared, d_c, d_sn, angle_min, angle_increment, max_iter,
                                 tol, d_vr, d_vp, d_T, d_it));
    D2H(T_out, d_T, (size_t)B * 72);
    if (iters_out) D2H(iters_out, d_it, (size_t)B * 4);
    HIPCHK(hipStreamSynchronize(c->stream));
    return SLAM_OK;
}

}  // extern "C"
