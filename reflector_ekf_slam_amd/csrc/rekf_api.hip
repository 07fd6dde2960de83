// rekf_api.hip -- host side of the C ABI declared in include/rekf.h.
//
// Owns the HIP device buffers and the stream of one filter handle and turns
// each reference method (reflector_ekf_slam.cc) into an asynchronous kernel
// chain.  Time bookkeeping (State::time, the "drop old odometry" test at
// reflector_ekf_slam.cc:211-212) is pure host data and stays on the host; every
// size that depends on device results (n, m, the match lists) stays on the
// device so no call here waits for the GPU except the getters.
#include "../../include/rekf.h"
#include "../../include/rekf_debug.h"
#include "ekf_dev.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

static_assert(REKF_MAX_OBS == REKF_MAX_OBS_WIDE, "host/device observation capacity mismatch");

namespace {

struct ProfSlot {
    hipEvent_t a, b;
    int kernel;
};

}  // namespace

struct rekf {
    rekf_options opt;
    int device;
    int max_landmarks;
    hipStream_t stream;
    RekfDev dev;
    // the Kn / HPt panels exist twice, by scan parity: k_mid of scan t writes one set, the held-back downdate of scan t reads it --
    // inside the launch in which scan t+1's mid role writes the other (one launch per scan)
    double *panel_base[2] = {nullptr, nullptr};     // HPt, Kn (2 x ld x 64 each)
    int panel_par = 0;
    // ... and so does P: dev.P is the STORED covariance (what every reader reads), P_alt the buffer the next in-launch downdate writes
    double *P_alt = nullptr;
    double time;
    double vt[3];
    int n_ub;                  // host upper bound of the device-resident n
    int n_det = 3;             // the same bound WITHOUT what peek_n learns on the way: it moves only where the call sequence says so (a scan adds its
                               // 2K, a wait for the device makes it exact).  Decisions that change the ARITHMETIC a scan sees -- auto-grow's wait for the
                               // exact n makes the next scan host-predicted: cos / sin of the wrapped heading, a last-place difference -- go by this one,
                               // so that results do not depend on how far the device happens to have got when the host looks
    int last_m_ub = 64;        // innovation-row bound of the last scan
    bool full;                 // n == n_max is known: no landmark can be added (k_augment is not launched)
    bool n_exact = true;       // n_ub IS the device's n (nothing that can append landmarks is in flight past the last read-back)
    bool auto_grow = false;    // rekf_set_auto_grow: re-reserve instead of dropping reflectors
    // ---- results the kernels publish: 16 tagged slots in pinned, device-mapped host memory ----
    RekfHostSlot *host_slots;
    RekfHostSlot *host_slots_dev;
    int slot_seq;
    bool pub_valid;            // a kernel already enqueued publishes (or has published) the CURRENT device pose under tag pub_seq
    int pub_seq;
    // every publisher's tag with the growth bound at its enqueue: lets a later call turn a published n into a bound (or the exact value)
    // of the CURRENT n without waiting for the device
    struct PubRec { int seq; long cum; } pub_ring[64];
    long cum_growth = 0;       // sum of 2 K over every scan enqueued while the state could still grow
    int last_pub_enq = 0;      // tag of the youngest publisher enqueued
    // ---- host mirror of the pose (reference reflector_ekf_slam.cc:154-223: Predict is O(1) on the pose and O(n) on two rows of P) ----
    // mir_mu / mir_P = pose mean and 3 x 3 pose block of the state once everything enqueued has run AND the predicts below are applied.
    // Valid after any pose read-back; HandleOdometryMessage / an empty scan then advance it on the host (same source as the kernels,
    // glibc libm) and only accumulate the composite G = I + a e0 e2^T + b e1 e2^T for P's landmark rows: no launch.
    bool mir_valid = false;
    double mir_mu[3], mir_P[9];
    bool lazy_pending = false; // the mirror is ahead of the device by the composite (lazy_a, lazy_b)
    double lazy_a = 0, lazy_b = 0;
    int flags_last = 0;        // sticky device flags as of the last read-back
    unsigned long scan_count = 0;   // parity = the RekfCtl::pred slot of the scan's Predict
    unsigned front_total = 0;       // observations handed to the front end so far = RekfCtl::front_count once they are all matched
    // LAZY DOWNDATE.  A scan's last k_downdate2 is not enqueued with the scan but held back: if the next call is another scan, it goes
    // out as k_dd_front, with that scan's front end (Predict's pose, ReflectorMatch: they need the mean, nothing of P) in workgroups of
    // its own beside it -- two launches per scan instead of three, and the match off the critical path.  Anything else that looks at
    // the device state enqueues it first (flush_dd).
    bool dd_pending = false;
    unsigned dd_scan = 0;           // ... the scan it belongs to (its write-ahead correction carries the id) ...
    RekfDev dd_dev;                 // the held-back launch: device view (publisher tag included) ...
    int dd_n_ub = 0;                // ... and the bound of n it was planned with
    bool dd_aug = false;            // the scan's k_augment is held back with it (the state can still grow): it runs right behind the downdate
    RekfFrontArgs dd_aug_args;      // ... with the scan's launch packet (the new reflectors' observations)
    bool dd_aug_inline_ok = false;  // ... and it may run inside the next scan's k_mid instead of a launch of its own (RekfCtl::augrec: whole scans only)
    // ONE LAUNCH PER SCAN (round 5; REKF_SCAN_LAUNCH=0 turns it off): the held-back downdate is not applied in front of the next scan's
    // k_mid but BESIDE it -- as a role of the same launch, from the stored P into the other P buffer -- while the mid role corrects what
    // it gathers by the pending panels (k_mid): the rank-m downdate is off the update's critical path.  For a filter that cannot grow
    // (h->full) and whole scans; everything else goes through the two-launch chain (k_dd_front, k_mid).
    bool scan_launch = true;
    // EXCLUSIVE (opt-in: rekf_set_exclusive / REKF_EXCLUSIVE=1): two hand-overs put WAITING workgroups into a launch (the mid role waits for
    // an in-grid front end; for the previous scan's in-grid augmentation).  That is deadlock-free only while nothing else competes for the
    // GPU's CUs, which a library cannot see: the caller says so, and even then the hand-overs are used only while this is the process's
    // only live handle.  Off: the front end is a launch of its own (k_front_mb / k_dd_front), the augmentation k_augment.
    bool exclusive = false;
    bool live_counted = false;      // this handle is in g_live_handles
    // SPECULATIVE MATCH (round 5; REKF_SPEC=0 turns it off).  What a scan's update waits for -- Predict's pose and ReflectorMatch against the
    // mean the PREVIOUS update left -- cannot start before that update has ended; run one scan EARLY it can: scan t + 1's front end as a
    // role of scan t's launch, against the mean that launch starts from, with the margins that let scan t + 1's k_mid prove (or repair)
    // every decision (RekfCtl::spec).  That needs scan t + 1's observations when scan t is launched: a caller that enqueues scan after
    // scan (no read-back in between) has its newest scan HELD on the host until the next call brings the one after it; every other
    // call (getters, odometry, sync) sends the held scan first.  Same results, bit for bit, as the exact front end.
    bool spec_enable = true;
    bool held = false;              // a scan waits on the host: ...
    double held_t = 0;
    int held_K = 0;
    bool held_gps = false;
    double held_gps3[3] = {0, 0, 0};
    float held_xy[2 * 32];
    bool spec_ready = false;        // the last launch ran the speculative front end of scan spec_scan
    bool scan_committed = false;    // process_scan: the host's bookkeeping has moved (an error before that leaves the handle as it was: the scan can be handed over again)
    // EARLY n (round 6): on a filter that can still grow the host does not know, when it enqueues scan t + 1, whether scan t appended reflectors --
    // and everything that makes an update ONE launch (n in the launch packet, the downdate as a role beside the mid role) needs to know.
    // k_mid's workgroup 0 therefore publishes the n the state will have behind the scan as soon as the scan's match record is final
    // (a few microseconds into the launch: host slot 13), and the NEXT scan's call waits for that slot before it plans its launch: the
    // host of a growing filter runs at most one launch ahead of the device (the device is never idle for it: the launch it waits for has
    // only just started).  A filter that cannot grow (n == n_max) publishes nothing and waits for nothing.
    // MATCH GRID (round 6; ekf_dev.h RekfCtl::grid_state): a hash grid over the landmark means lets k_mid match a HOST-PREDICTED scan itself
    // (the reference node's pattern: the pose read back after every scan, odometry in between) -- no front-end launch, no kernel boundary on
    // the scan-to-pose path.  Built lazily (k_grid_build, in stream order in front of the first scan that wants it), kept by the kernels
    // (new reflectors are binned where they appear; a landmark that drifts too far from its binning position invalidates the grid and
    // says so through host slot 14: the scans in between match by the full sweep, the host rebuilds).
    bool grid_on = true;            // rekf_debug_set_grid(h, 0) turns it off (the twin of the parity tests: k_front_mb as a launch of its own)
    bool grid_dev_valid = false;    // the device grid is built and the kernels are maintaining it (h->dev.grid_* set)
    int grid_note_seen = 0;         // tag of the last invalidation note taken from host slot 14
    int grid_builds = 0;
    int *grid_bucket = nullptr;
    float *grid_p0 = nullptr;
    int grid_mask = 0;
    float grid_drift = REKF_GRID_DRIFT;
    int grid_mask_override = -1;    // rekf_debug_set_grid: a smaller table (tests: bucket overflow)
    bool early_valid = false;       // the last scan's k_mid publishes its n under tag early_seq ...
    int early_seq = 0;
    int early_n_before = -1;        // ... and this was the (exact) n it started from (-1: not known): equal = the scan appended nothing
    unsigned spec_scan = 0;
    // WHO PUBLISHES pose, pose block, n and flags of a scan.  A caller that reads the pose back after its scans (the reference's node,
    // src/ros_node.cc:514-515) gets them from k_mid's workgroup 0 -- a kernel earlier: GetPose does not wait for the downdate -- which
    // costs that kernel 0.8 us (it ends when the PCIe writes are through); a caller that enqueues scan after scan gets them from the
    // first workgroup of the downdate, at its start, off every critical path.  Decided per scan from what the caller did since the last one.
    bool pose_read_since_scan = true;
    bool last_scan_empty = false;   // the last HandleObservationMessage had no points: its (empty) match record lives here, not on the device
    RekfCtl *ctl_staging;      // pinned copy of the control block
    std::string hip_error;
    int flags_seen = 0;        // sticky device flags already reported on stderr
    int inject_failure = 0;    // rekf_debug_inject_failure: the next HandleObservationMessage fails at this stage
    double *dev_ell;           // device scratch for k_ellipses (5 doubles per landmark of capacity)
    double *dev_pred;          // device scratch for k_predict_rows (4 * ld + 12 doubles)
    float *dev_obs = nullptr;  // a wide scan's observations (REKF_MAX_OBS_WIDE x 2 floats)
    float *obs_staging = nullptr;   // pinned: a wide scan is copied here before the call returns, and from here to the device
    hipEvent_t obs_staging_ev = nullptr;
    bool obs_staging_busy = false;
    double *dev_mu_lin = nullptr;   // the mean a wide scan is linearised at (copy taken before its first block step)
    // profiling
    bool prof_on;
    int prof_mask;
    std::vector<ProfSlot> prof_slots;
    size_t prof_used;
    int prof_open = 0;         // ProfScopes currently open (they nest): the slot table is never flushed while one is
    double prof_total_us[REKF_K_COUNT];
    long prof_count[REKF_K_COUNT];
    std::vector<float> prof_update_us;   // every REKF_K_UPDATE reading since the last reset
};

namespace {

// How many handles this process holds: the in-launch hand-overs of an EXCLUSIVE handle (struct rekf) are used only while it is the only one
std::atomic<int> g_live_handles{0};
bool in_grid_ok(const rekf_t *h) { return h->exclusive && g_live_handles.load(std::memory_order_seq_cst) == 1; }

#define HIP_TRY(h, expr)                                                         \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            if (h) (h)->hip_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return REKF_ERR_HIP;                                                 \
        }                                                                        \
    } while (0)

int round_up(int x, int q) { return (x + q - 1) / q * q; }

static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("isb" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
// wait until the `count` slots from `first` carry `seq` (a kernel on the handle's stream stores them).  Polls with a pause; a stream
// that has drained without them is an error at once; a stream that is still busy (a shared or oversubscribed GPU, a profiler, a
// debugger: the kernel is merely slow) is polled up to a hard bound of REKF_WAIT_SECONDS (default 30) -- never an unbounded
// hipStreamSynchronize: a hung kernel must come back as REKF_ERR_HIP, not block GetPose and odometry for ever.
template <class H> int wait_slots(H *h, int first, int count, int seq)
{
    static const double limit_s = [] { const char *e = std::getenv("REKF_WAIT_SECONDS"); const double v = e ? std::atof(e) : 0.0; return v > 0.0 ? v : 30.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (int k = first; k < first + count; ++k) {
        const int *tag = &h->host_slots[k].seq;
        while (__atomic_load_n(tag, __ATOMIC_ACQUIRE) != seq) {
            cpu_relax();
            // (a slot that has not come within ~40 us of polling is a launch queued behind other work -- other sessions of this process,
            // whose host threads may be waiting for this CPU: from there on every poll gives the CPU away first.  Measured on the GPU box's
            // two logical CPUs with four sessions on four threads: 39.5 k -> 45.0 k updates/s in aggregate, profiles/r06_experiments.txt 13)
            if (spins > 4096u) std::this_thread::yield();
            if ((++spins & 0xffffu) == 0) {
                const bool drained = hipStreamQuery(h->stream) != hipErrorNotReady;
                if (drained) {
                    if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) == seq) break;
                    h->hip_error = "a pose kernel finished without publishing its result";
                    return REKF_ERR_HIP;
                }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
                    h->hip_error = "timed out waiting for a kernel to publish the pose (REKF_WAIT_SECONDS)";
                    return REKF_ERR_HIP;
                }
                if ((spins & 0xfffffu) == 0) std::this_thread::yield();
            }
        }
    }
    return REKF_OK;
}

int prof_flush(rekf_t *h)
{
    if (h->prof_used == 0) return REKF_OK;
    if (h->prof_open > 0) return REKF_ERR_INVALID;    // an open scope has not recorded its end event yet
    const size_t used = h->prof_used;
    h->prof_used = 0;                                 // whatever happens below, the table is reused from the start
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < used; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_slots[i].a, h->prof_slots[i].b));
        h->prof_total_us[h->prof_slots[i].kernel] += 1e3 * (double)ms;
        h->prof_count[h->prof_slots[i].kernel] += 1;
        if (h->prof_slots[i].kernel == REKF_K_UPDATE && h->prof_update_us.size() < (1u << 20))
            h->prof_update_us.push_back(1e3f * ms);
    }
    return REKF_OK;
}

struct ProfScope {
    // Scopes nest (REKF_K_UPDATE around the per-kernel brackets): a scope keeps the INDEX of its slot -- the vector may
    // reallocate while it is open -- and the slot table is only flushed (which synchronises and reuses it from the
    // start) when no scope is open; a scope that finds the table full while another is open records nothing.
    rekf_t *h;
    long slot;
    ProfScope(rekf_t *h_, int kernel) : h(h_), slot(-1)
    {
        if (!h->prof_on || !((h->prof_mask >> kernel) & 1)) return;
        if (h->prof_used == h->prof_slots.size()) {
            if (h->prof_slots.size() >= 65536) {
                if (h->prof_open > 0 || prof_flush(h) != REKF_OK || h->prof_used != 0) return;
            } else {
                ProfSlot s;
                if (hipEventCreate(&s.a) != hipSuccess) return;
                if (hipEventCreate(&s.b) != hipSuccess) { (void)hipEventDestroy(s.a); return; }
                s.kernel = kernel;
                h->prof_slots.push_back(s);
            }
        }
        if (h->prof_used >= h->prof_slots.size()) return;
        slot = (long)h->prof_used++;
        h->prof_slots[(size_t)slot].kernel = kernel;
        ++h->prof_open;
        (void)hipEventRecord(h->prof_slots[(size_t)slot].a, h->stream);
    }
    ~ProfScope()
    {
        if (slot < 0) return;
        (void)hipEventRecord(h->prof_slots[(size_t)slot].b, h->stream);
        --h->prof_open;
    }
};

void fill_front_args(const rekf_t *h, RekfFrontArgs &a, double dt)
{
    std::memset(&a, 0, sizeof(a));
    a.pair0 = -1;
    a.dt = dt;
    a.vt[0] = h->vt[0]; a.vt[1] = h->vt[1]; a.vt[2] = h->vt[2];
    a.lin_cov = h->opt.linear_velocity_cov;
    a.ang_cov = h->opt.angular_velocity_cov;
    a.obs_cov = h->opt.observation_cov;
    a.model = (h->opt.odom_model == REKF_ODOM_DIFF) ? 0 : 1;   // cc:13-32: anything else -> OMNI
}

int device_flags_to_code(int flags)
{
    if (flags & REKF_FLAG_STARVED) return REKF_ERR_HIP;
    if (flags & REKF_FLAG_SINGULAR) return REKF_ERR_SINGULAR;
    if (flags & REKF_FLAG_CAPACITY) return REKF_ERR_CAPACITY;
    return REKF_OK;
}

// One loud line per newly seen sticky bit: the reference never drops a reflector (cc:311-364), so a caller that only
// uses the getters must still hear about it.
void report_flags(rekf_t *h, int flags)
{
    const int fresh = flags & ~h->flags_seen;
    if (fresh & REKF_FLAG_CAPACITY)
        std::fprintf(stderr, "rekf: landmark capacity (max_landmarks = %d) exceeded: new reflectors are being DROPPED "
                             "(the reference grows its state without bound)\n", h->max_landmarks);
    if (fresh & REKF_FLAG_STARVED) {
        std::fprintf(stderr, "rekf: a hand-over inside a launch gave up waiting (other work held the GPU's CUs): the filter state is not meaningful any more; "
                             "an EXCLUSIVE handle (rekf_set_exclusive / REKF_EXCLUSIVE=1) must have the GPU to itself\n");
        h->hip_error = "an in-launch hand-over starved (REKF_FLAGBIT_STARVED)";
    }
    if (fresh & REKF_FLAG_SINGULAR)
        std::fprintf(stderr, "rekf: innovation covariance was not positive definite in some scan; the update was applied as computed\n");
    h->flags_seen |= flags;
}

// ---- publishers and the n they carry ------------------------------------------------------------------------------
// Every call that changes the device state ends in ONE kernel that stores pose mean, 3 x 3 pose block, n and the sticky flags as
// tagged slots into pinned host memory (the tile-(0,0) workgroup of a scan's last k_downdate2, else a 64-thread publish kernel).  new_publisher hands out the tag and remembers the growth bound at that point.
int new_publisher(rekf_t *h)
{
    const int seq = ++h->slot_seq;
    h->pub_ring[seq & 63] = {seq, h->cum_growth};
    h->last_pub_enq = seq;
    h->pub_valid = true; h->pub_seq = seq;
    return seq;
}

// Non-blocking: whatever n the device has published so far bounds the current n (each scan enqueued behind that publisher can
// have appended at most its K reflectors), and IS the current n when nothing that can append was enqueued behind it.
void peek_n(rekf_t *h)
{
    const RekfHostSlot *s = &h->host_slots[12];
    const int s1 = __atomic_load_n(&s->seq, __ATOMIC_ACQUIRE);
    const double v = *(volatile const double *)&s->v;
    const int s2 = __atomic_load_n(&s->seq, __ATOMIC_ACQUIRE);
    if (s1 != s2 || s1 <= 0) return;
    const rekf::PubRec &r = h->pub_ring[s1 & 63];
    if (r.seq != s1) return;
    const long slack = h->cum_growth - r.cum;
    const long nb = (long)v + slack;
    if (nb < h->n_ub || slack == 0) h->n_ub = (int)(nb < h->dev.n_max ? nb : h->dev.n_max);
    if (slack == 0) { h->n_exact = true; h->full = h->n_ub >= h->dev.n_max; }
}

// EARLY n (struct rekf): the n the last scan leaves, from its k_mid's workgroup 0 (host slot 13) -- waits for that launch to have got
// as far as its match record (a few microseconds in).  Tells, too, whether that scan appended reflectors: if it did not, nothing of it is
// pending but its downdate (no k_augment, no rows to append), and the next scan may take it along as a role of its own launch.
int learn_early_n(rekf_t *h)
{
    if (!h->early_valid) return REKF_OK;
    const int rc = wait_slots(h, 13, 1, h->early_seq);
    h->early_valid = false;                            // (whatever happened: a launch that never published must not fail every later call)
    if (rc != REKF_OK) return rc;
    int n = (int)h->host_slots[13].v;
    if (n > h->dev.n_max) n = h->dev.n_max;
    h->n_ub = n; h->n_det = n; h->n_exact = true;
    h->full = n >= h->dev.n_max;
    if (h->early_n_before >= 0 && n == h->early_n_before) h->dd_aug = false;     // (the held-back k_augment would find nothing to do)
    return REKF_OK;
}

// MATCH GRID (struct rekf): an invalidation note from the device (host slot 14: a landmark drifted out of its bin -- rebuild; a bucket ran
// over -- this world does not fit the table, stop using the grid), and the lazy build in front of a scan that wants the grid.
void grid_take_note(rekf_t *h)
{
    if (!h->grid_dev_valid) return;
    const RekfHostSlot *s = &h->host_slots[14];
    const int tag = __atomic_load_n(&s->seq, __ATOMIC_ACQUIRE);
    if (tag == h->grid_note_seen) return;
    h->grid_note_seen = tag;
    if (*(volatile const double *)&s->v >= 2.0) h->grid_on = false;
    h->grid_dev_valid = false;
    h->dev.grid_bucket = nullptr; h->dev.grid_p0 = nullptr; h->dev.grid_note = nullptr;
}
void grid_ensure(rekf_t *h)          // (call with n exact: h->n_ub landmarks' means are in h->dev.mu)
{
    if (h->grid_dev_valid) return;
    h->dev.grid_bucket = h->grid_bucket; h->dev.grid_p0 = h->grid_p0;
    h->dev.grid_mask = (h->grid_mask_override >= 0 && h->grid_mask_override < h->grid_mask) ? h->grid_mask_override : h->grid_mask;
    h->dev.grid_drift = h->grid_drift;
    h->dev.grid_note = h->host_slots_dev + 14;
    rekf_launch_grid_build(h->dev, h->n_ub, -(++h->grid_builds), h->stream);      // (note tags of a build are negative: never a scan id)
    h->grid_dev_valid = true;
}

// the held-back downdate (struct rekf: LAZY DOWNDATE) goes out on its own
int flush_dd(rekf_t *h)
{
    if (!h->dd_pending) return REKF_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    h->dd_pending = false;
    h->dd_dev.P_out = h->dd_dev.P;                     // in place (the stored P catches up with the filter)
    { ProfScope ps(h, REKF_K_DOWNDATE); rekf_launch_downdate(h->dd_dev, h->dd_n_ub, h->stream); }
    if (h->dd_aug) { ProfScope ps(h, REKF_K_AUGMENT); rekf_launch_augment(h->dd_dev, h->dd_aug_args, h->stream); h->dd_aug = false; }
    HIP_TRY(h, hipGetLastError());
    return REKF_OK;
}

int pull_ctl(rekf_t *h)
{
    HIP_TRY(h, hipSetDevice(h->device));
    { int rcf = flush_dd(h); if (rcf != REKF_OK) return rcf; }
    HIP_TRY(h, hipMemcpyAsync(h->ctl_staging, h->dev.ctl, sizeof(RekfCtl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_ub = h->ctl_staging->n;
    h->n_det = h->n_ub;
    h->n_exact = true;
    h->full = h->ctl_staging->n >= h->dev.n_max;
    h->flags_last = h->ctl_staging->err;
    report_flags(h, h->ctl_staging->err);
    return REKF_OK;
}

// ---- the pose mirror ----------------------------------------------------------------------------------------------
// Bring the mirror up to date with the device: wait for the slots of the publisher in flight (or enqueue a publish kernel behind
// whatever is enqueued).  No copy engine, no wait for a completion signal.  n and the flags arrive with the pose.
int refresh_mirror(rekf_t *h)
{
    if (h->mir_valid) return REKF_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // the held-back downdate goes out only if IT is the publisher being waited for (a caller that reads the pose after every scan gets it
    // from k_mid: the downdate then stays pending across the read-back and runs beside the next scan's k_mid) or nothing publishes at all
    if (h->dd_pending && (h->dd_dev.pub || !h->pub_valid)) { int rcf = flush_dd(h); if (rcf != REKF_OK) return rcf; }
    if (!h->pub_valid) {
        const int seq = new_publisher(h);
        rekf_launch_publish_pose(h->dev, h->host_slots_dev, seq, h->stream);
        HIP_TRY(h, hipGetLastError());
    }
    h->pose_read_since_scan = true;
    int rc = wait_slots(h, 0, 13, h->pub_seq);        // all of them: the last kernels of a call store different slots
    if (rc != REKF_OK) return rc;
    for (int q = 0; q < 3; ++q) h->mir_mu[q] = h->host_slots[q].v;
    for (int q = 0; q < 9; ++q) h->mir_P[q] = h->host_slots[3 + q].v;
    peek_n(h);
    if (h->n_exact) h->n_det = h->n_ub;              // (the publisher just waited for is the last one enqueued: n is exact here, by construction)
    h->flags_last = h->host_slots[12].aux;
    report_flags(h, h->flags_last);
    h->mir_valid = true;
    h->lazy_pending = false; h->lazy_a = 0; h->lazy_b = 0;
    return REKF_OK;
}

// Predict (cc:154-206) on the mirror: the kernels' own source (ekf_dev.h) with the host's libm -- the same libm a CPU build of the
// reference calls.  The O(n) part, rows / columns 0, 1 of P against row / column 2, is deferred: G_k ... G_1 =
// I + (sum a) e0 e2^T + (sum b) e1 e2^T exactly, because e2^T (a, b, 0)^T = 0.
Motion host_predict(rekf_t *h, const RekfFrontArgs &a, double *mu3, double *P9, bool accumulate)
{
#pragma clang fp contract(off)
    Motion mo;
    motion_terms(a, mu3[2], mo);
    mu3[0] = mu3[0] + mo.d[0];
    mu3[1] = mu3[1] + mo.d[1];
    const double th = mu3[2] + mo.d[2];
    mu3[2] = atan2(sin(th), cos(th));                  // cc:181 / :205
    corner_predict(P9, 3, mo);
    if (accumulate) { h->lazy_a += mo.a; h->lazy_b += mo.b; h->lazy_pending = true; }
    return mo;
}

void put_host_prediction(const rekf_t *h, RekfFrontArgs &a)
{
    a.host_pred = 1;
    a.pre_pose[0] = h->mir_mu[0]; a.pre_pose[1] = h->mir_mu[1]; a.pre_pose[2] = h->mir_mu[2];
    a.pre_pose[3] = cos(h->mir_mu[2]); a.pre_pose[4] = sin(h->mir_mu[2]);     // cc:252-253: of the wrapped heading
    a.pre_ab[0] = h->lazy_a; a.pre_ab[1] = h->lazy_b;
    for (int q = 0; q < 9; ++q) a.pre_C9[q] = h->mir_P[q];
}

// The device's P / mu catch up with the mirror (needed before anything but a scan reads them there).
int flush_lazy(rekf_t *h)
{
    { int rcf = flush_dd(h); if (rcf != REKF_OK) return rcf; }
    if (!h->lazy_pending) return REKF_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    RekfFrontArgs a;
    fill_front_args(h, a, 0.0);
    put_host_prediction(h, a);
    h->dev.n_known = h->n_exact ? h->n_ub : -1;
    rekf_launch_apply_predict(h->dev, a, h->stream);
    HIP_TRY(h, hipGetLastError());
    h->lazy_pending = false; h->lazy_a = 0; h->lazy_b = 0;
    h->pub_valid = false;                              // the slots hold the pose from before these predicts (the mirror is what is current)
    return REKF_OK;
}

// The device keeps P as its lower triangle (ekf_dev.h): the callers' n x n copies get the upper triangle from it (blocked: the
// strided side of the transpose stays in cache).
void mirror_lower(double *sg, int n)
{
    constexpr int B = 32;
    for (int j0 = 0; j0 < n; j0 += B)
        for (int i0 = 0; i0 <= j0; i0 += B) {
            const int j1 = j0 + B < n ? j0 + B : n, i1 = i0 + B < n ? i0 + B : n;
            for (int j = j0; j < j1; ++j)
                for (int i = i0; i < i1 && i < j; ++i) sg[(size_t)i + (size_t)j * n] = sg[(size_t)j + (size_t)i * n];
        }
}

struct DevBuffers {            // everything whose size depends on max_landmarks (rekf_create, rekf_reserve)
    double *mu = nullptr, *mu_out = nullptr, *P = nullptr, *P2 = nullptr, *HPt = nullptr, *Kn = nullptr, *dev_pred = nullptr, *dev_mu_lin = nullptr,
           *dev_ell = nullptr;
    int *grid_bucket = nullptr;
    float *grid_p0 = nullptr;
    int grid_mask = 0;
    int ld = 0, n_max = 0;
};
void free_buffers(DevBuffers &b)
{
    (void)hipFree(b.mu); (void)hipFree(b.mu_out); (void)hipFree(b.P); (void)hipFree(b.P2); (void)hipFree(b.HPt); (void)hipFree(b.Kn);
    (void)hipFree(b.dev_pred); (void)hipFree(b.dev_mu_lin); (void)hipFree(b.dev_ell); (void)hipFree(b.grid_bucket); (void)hipFree(b.grid_p0);
    b = DevBuffers();
}
int alloc_buffers(rekf_t *h, int max_landmarks, DevBuffers &b)
{
    const int n_max = 3 + 2 * max_landmarks;
    const int ld = round_up(n_max, 64);
    if ((double)ld * ld * 8.0 >= 4294967296.0) return REKF_ERR_UNSUPPORTED;   // the kernels address P with 32-bit byte offsets
    b.ld = ld; b.n_max = n_max;
    HIP_TRY(h, hipMalloc(&b.mu, sizeof(double) * ld));
    HIP_TRY(h, hipMalloc(&b.mu_out, sizeof(double) * ld));
    HIP_TRY(h, hipMalloc(&b.P, sizeof(double) * (size_t)ld * ld));
    HIP_TRY(h, hipMalloc(&b.P2, sizeof(double) * (size_t)ld * ld));     // (the in-launch downdate's destination: 2 x 36 MB at C3 of 288 GB)
    HIP_TRY(h, hipMalloc(&b.HPt, sizeof(double) * 2 * (size_t)ld * REKF_PANEL_COLS));      // (both parities, struct rekf)
    HIP_TRY(h, hipMalloc(&b.Kn, sizeof(double) * 2 * (size_t)ld * REKF_PANEL_COLS));
    HIP_TRY(h, hipMalloc(&b.dev_pred, sizeof(double) * (4 * (size_t)ld + 16)));
    HIP_TRY(h, hipMalloc(&b.dev_mu_lin, sizeof(double) * ld));
    HIP_TRY(h, hipMalloc(&b.dev_ell, sizeof(double) * 5 * (size_t)(max_landmarks > 0 ? max_landmarks : 1)));
    {   // the match grid: at least four buckets per reflector of the capacity (a power of two), 8 ints each
        int gh = 1024;
        while (gh < 4 * max_landmarks) gh *= 2;
        b.grid_mask = gh - 1;
        HIP_TRY(h, hipMalloc(&b.grid_bucket, sizeof(int) * REKF_GRID_INTS_PER_BUCKET * (size_t)gh));
        HIP_TRY(h, hipMemsetAsync(b.grid_bucket, 0, sizeof(int) * REKF_GRID_INTS_PER_BUCKET * (size_t)gh, h->stream));
        HIP_TRY(h, hipMalloc(&b.grid_p0, sizeof(float) * 3 * (size_t)(max_landmarks + 8)));         // (p0[Lcap][2] | slot[Lcap])
        HIP_TRY(h, hipMemsetAsync(b.grid_p0, 0xff, sizeof(float) * 3 * (size_t)(max_landmarks + 8), h->stream));     // (slot = -1: no entry)
    }
    HIP_TRY(h, hipMemsetAsync(b.mu, 0, sizeof(double) * ld, h->stream));
    HIP_TRY(h, hipMemsetAsync(b.mu_out, 0, sizeof(double) * ld, h->stream));
    HIP_TRY(h, hipMemsetAsync(b.P, 0, sizeof(double) * (size_t)ld * ld, h->stream));              // cc:10-11
    HIP_TRY(h, hipMemsetAsync(b.P2, 0, sizeof(double) * (size_t)ld * ld, h->stream));             // (rows >= n stay zero in both: the tiles are read whole)
    HIP_TRY(h, hipMemsetAsync(b.HPt, 0, sizeof(double) * 2 * (size_t)ld * REKF_PANEL_COLS, h->stream));
    HIP_TRY(h, hipMemsetAsync(b.Kn, 0, sizeof(double) * 2 * (size_t)ld * REKF_PANEL_COLS, h->stream));
    return REKF_OK;
}
void adopt_buffers(rekf_t *h, const DevBuffers &b, int max_landmarks)
{
    h->dev.mu = b.mu; h->dev.mu_out = b.mu_out; h->dev.P = b.P; h->dev.P_out = b.P; h->P_alt = b.P2; h->dev.HPt = b.HPt; h->dev.Kn = b.Kn;
    h->panel_base[0] = b.HPt; h->panel_base[1] = b.Kn;
    h->panel_par = 0;
    h->dev_pred = b.dev_pred; h->dev_mu_lin = b.dev_mu_lin; h->dev_ell = b.dev_ell;
    h->dev.ld = b.ld; h->dev.n_max = b.n_max;
    h->dev.mu_lin = nullptr;
    h->max_landmarks = max_landmarks;
    h->grid_bucket = b.grid_bucket; h->grid_p0 = b.grid_p0; h->grid_mask = b.grid_mask;
    h->grid_dev_valid = false;                        // (new buffers: built again in front of the first scan that wants it)
    h->dev.grid_bucket = nullptr; h->dev.grid_p0 = nullptr; h->dev.grid_mask = 0; h->dev.grid_note = nullptr;
}
// the other set of panels for the next k_mid (call once the scan's downdate has taken its copy of the device view)
void flip_panels(rekf_t *h)
{
    h->panel_par ^= 1;
    const size_t po = (size_t)h->panel_par * (size_t)h->dev.ld * REKF_PANEL_COLS;
    h->dev.HPt = h->panel_base[0] + po; h->dev.Kn = h->panel_base[1] + po;
}
DevBuffers current_buffers(const rekf_t *h)
{
    DevBuffers b;
    b.mu = h->dev.mu; b.mu_out = h->dev.mu_out; b.P = h->dev.P; b.P2 = h->P_alt; b.HPt = h->panel_base[0]; b.Kn = h->panel_base[1];
    b.dev_pred = h->dev_pred; b.dev_mu_lin = h->dev_mu_lin; b.dev_ell = h->dev_ell; b.ld = h->dev.ld; b.n_max = h->dev.n_max;
    b.grid_bucket = h->grid_bucket; b.grid_p0 = h->grid_p0; b.grid_mask = h->grid_mask;
    return b;
}

}  // namespace

extern "C" {

struct NextScan;
static int flush_held(rekf_t *h);
#define FLUSH_HELD(h) do { if ((h)->held) { int rch_ = flush_held(h); if (rch_ != REKF_OK) return rch_; } } while (0)

int rekf_abi_version(void) { return REKF_ABI_VERSION; }

const char *rekf_strerror(int code)
{
    switch (code) {
    case REKF_OK: return "ok";
    case REKF_ERR_INVALID: return "invalid argument";
    case REKF_ERR_HIP: return "HIP runtime error";
    case REKF_ERR_TOO_MANY_OBS: return "too many observations in one scan";
    case REKF_ERR_CAPACITY: return "landmark capacity exceeded; new reflectors dropped";
    case REKF_ERR_SINGULAR: return "innovation covariance not positive definite";
    case REKF_ERR_BUFFER: return "caller buffer too small";
    case REKF_ERR_UNSUPPORTED: return "unsupported size or option";
    default: return "unknown error";
    }
}

const char *rekf_last_hip_error(rekf_t *h) { return h ? h->hip_error.c_str() : ""; }

int rekf_create(const rekf_options *opt, int max_landmarks, int device, rekf_t **out)
{
    if (!opt || !out || max_landmarks < 1) return REKF_ERR_INVALID;
    *out = nullptr;
    rekf_t *h = new (std::nothrow) rekf();
    if (!h) return REKF_ERR_INVALID;
    h->opt = *opt;
    h->device = device;
    h->max_landmarks = max_landmarks;
    h->time = opt->init_time;                         // cc:8
    h->vt[0] = h->vt[1] = h->vt[2] = 0.0;             // cc:6
    h->n_ub = 3;
    h->full = false;
    { const char *e = std::getenv("REKF_SCAN_LAUNCH"); h->scan_launch = !(e && e[0] == '0'); }
    { const char *e = std::getenv("REKF_EXCLUSIVE"); h->exclusive = e && e[0] == '1'; }
    { const char *e = std::getenv("REKF_SPEC"); h->spec_enable = !(e && e[0] == '0'); }
    h->prof_on = false;
    h->prof_mask = -1;
    h->prof_used = 0;
    for (int k = 0; k < REKF_K_COUNT; ++k) { h->prof_total_us[k] = 0; h->prof_count[k] = 0; }
    h->stream = nullptr;
    h->host_slots = nullptr; h->host_slots_dev = nullptr; h->slot_seq = 0; h->pub_valid = false; h->pub_seq = 0;
    for (auto &r : h->pub_ring) r = {0, 0};
    h->ctl_staging = nullptr;
    h->dev_ell = nullptr;
    h->dev_pred = nullptr;
    std::memset(&h->dev, 0, sizeof(h->dev));

    int rc = [&]() -> int {
        HIP_TRY(h, hipSetDevice(device));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_TRY(h, hipMalloc(&h->dev.ctl, sizeof(RekfCtl)));
        DevBuffers b;
        int rb = alloc_buffers(h, max_landmarks, b);
        if (rb != REKF_OK) { free_buffers(b); return rb; }
        adopt_buffers(h, b, max_landmarks);
        HIP_TRY(h, hipMalloc(&h->dev.cp, sizeof(double) * 2 * REKF_CP_LD * REKF_CP_LD));      // (write-ahead correction panels, RekfCtl::cp_*)
        HIP_TRY(h, hipMemsetAsync(h->dev.cp, 0, sizeof(double) * 2 * REKF_CP_LD * REKF_CP_LD, h->stream));
        HIP_TRY(h, hipMalloc(&h->dev_obs, sizeof(float) * 2 * REKF_MAX_OBS_WIDE));
        HIP_TRY(h, hipHostMalloc(&h->obs_staging, sizeof(float) * 2 * REKF_MAX_OBS_WIDE));
        HIP_TRY(h, hipEventCreateWithFlags(&h->obs_staging_ev, hipEventDisableTiming));
        // results come back as tagged slots in pinned, device-mapped host memory (no copy engine on the pose path)
        HIP_TRY(h, hipHostMalloc(&h->host_slots, sizeof(RekfHostSlot) * 16, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->host_slots, 0, sizeof(RekfHostSlot) * 16);
        void *dv = nullptr;
        HIP_TRY(h, hipHostGetDevicePointer(&dv, h->host_slots, 0));
        h->host_slots_dev = (RekfHostSlot *)dv;
        HIP_TRY(h, hipHostMalloc(&h->ctl_staging, sizeof(RekfCtl)));
        h->dev.M_map = 0;
        HIP_TRY(h, hipMemsetAsync(h->dev.ctl, 0, sizeof(RekfCtl), h->stream));
        std::memset(h->ctl_staging, 0, sizeof(RekfCtl));
        h->ctl_staging->n = 3;
        HIP_TRY(h, hipMemcpyAsync(h->dev.ctl, h->ctl_staging, sizeof(int) * 2, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemcpy(h->dev.mu, opt->init_pose, sizeof(double) * 3, hipMemcpyHostToDevice));   // cc:9
        return REKF_OK;
    }();
    if (rc != REKF_OK) {
        std::fprintf(stderr, "rekf_create: %s\n", rc == REKF_ERR_UNSUPPORTED ? "max_landmarks too large (P is addressed with 32-bit byte offsets)" : h->hip_error.c_str());
        rekf_destroy(h);
        return rc;
    }
    // the mirror starts current: init_pose and a zero pose block (cc:9-11)
    for (int q = 0; q < 3; ++q) h->mir_mu[q] = opt->init_pose[q];
    for (int q = 0; q < 9; ++q) h->mir_P[q] = 0.0;
    h->mir_valid = true;
    g_live_handles.fetch_add(1, std::memory_order_seq_cst);
    h->live_counted = true;
    *out = h;
    return REKF_OK;
}

void rekf_destroy(rekf_t *h)
{
    if (!h) return;
    h->held = false;
    if (h->live_counted) g_live_handles.fetch_sub(1, std::memory_order_seq_cst);
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &s : h->prof_slots) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    DevBuffers b = current_buffers(h);
    free_buffers(b);
    (void)hipFree(h->dev.ctl); (void)hipFree(h->dev.cp);
    (void)hipFree(h->dev.map_xy); (void)hipFree(h->dev.map_cov); (void)hipFree(h->dev_obs);
    if (h->obs_staging) (void)hipHostFree(h->obs_staging);
    if (h->obs_staging_ev) (void)hipEventDestroy(h->obs_staging_ev);
    if (h->host_slots) (void)hipHostFree(h->host_slots);
    if (h->ctl_staging) (void)hipHostFree(h->ctl_staging);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rekf_reserve(rekf_t *h, int new_max_landmarks)
{
    if (!h || new_max_landmarks < 1) return REKF_ERR_INVALID;
    if (new_max_landmarks <= h->max_landmarks) return REKF_OK;
    FLUSH_HELD(h);
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = flush_lazy(h);
    if (rc != REKF_OK) return rc;
    rc = pull_ctl(h);                                  // drains the stream; n is exact afterwards
    if (rc != REKF_OK) return rc;
    const int n = h->ctl_staging->n;
    DevBuffers nb;
    rc = alloc_buffers(h, new_max_landmarks, nb);
    if (rc != REKF_OK) { free_buffers(nb); (void)hipGetLastError(); return rc; }
    // the re-layout: column c of P moves from stride ld to stride ld', everything past n stays zero (the reference resizes and
    // copies on EVERY augment, cc:360-363; here once per doubling)
    rc = [&]() -> int {
        HIP_TRY(h, hipMemcpyAsync(nb.mu, h->dev.mu, sizeof(double) * n, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpy2DAsync(nb.P, sizeof(double) * nb.ld, h->dev.P, sizeof(double) * h->dev.ld, sizeof(double) * n, n,
                                    hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return REKF_OK;
    }();
    if (rc != REKF_OK) { free_buffers(nb); return rc; }
    DevBuffers old = current_buffers(h);
    adopt_buffers(h, nb, new_max_landmarks);
    free_buffers(old);
    h->full = false;
    h->n_ub = n; h->n_det = n; h->n_exact = true;
    return REKF_OK;
}

int rekf_set_exclusive(rekf_t *h, int on)
{
    if (!h) return REKF_ERR_INVALID;
    h->exclusive = on != 0;
    return REKF_OK;
}

int rekf_set_auto_grow(rekf_t *h, int on)
{
    if (!h) return REKF_ERR_INVALID;
    h->auto_grow = on != 0;
    return REKF_OK;
}

int rekf_get_capacity(rekf_t *h, int *max_landmarks)
{
    if (!h || !max_landmarks) return REKF_ERR_INVALID;
    *max_landmarks = h->max_landmarks;
    return REKF_OK;
}

int rekf_set_map(rekf_t *h, const float *xy, const double *cov, int M)
{
    if (!h || M < 0 || (M > 0 && (!xy || !cov))) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->dev.map_xy); (void)hipFree(h->dev.map_cov);
    h->dev.map_xy = nullptr; h->dev.map_cov = nullptr; h->dev.M_map = 0; h->dev.map_lip = 0.0;
    if (M == 0) return REKF_OK;
    // the map branch of the match measures sqrt(e^T S e) with S the stored 2 x 2 COVARIANCE (cc:411, quirk Q3).  For symmetric positive
    // semi-definite S that is a seminorm: it moves by at most sqrt(lambda_max) |de| when the observation moves by de -- the bound the
    // speculative match's margin proof needs (k_mid).  Any other S: no bound, scans are not speculated for (map_lip < 0)
    double lip2 = 0.0;
    bool psd = true;
    for (int j = 0; j < M; ++j) {
        const double a = cov[4 * j], b = cov[4 * j + 1], c = cov[4 * j + 2], d = cov[4 * j + 3];
        const double tol = 1e-12 * (std::fabs(a) + std::fabs(d) + 1e-300);
        if (!(std::fabs(b - c) <= tol)) { psd = false; break; }
        const double tr = a + d, det = a * d - b * c;
        const double disc = std::sqrt(std::fmax(0.25 * tr * tr - det, 0.0));
        const double lmax = 0.5 * tr + disc, lmin = 0.5 * tr - disc;
        if (!(lmin >= -tol) || !(lmax == lmax)) { psd = false; break; }
        if (lmax > lip2) lip2 = lmax;
    }
    h->dev.map_lip = psd ? std::sqrt(lip2) : -1.0;
    HIP_TRY(h, hipMalloc(&h->dev.map_xy, sizeof(float) * 2 * (size_t)M));
    HIP_TRY(h, hipMalloc(&h->dev.map_cov, sizeof(double) * 4 * (size_t)M));
    HIP_TRY(h, hipMemcpy(h->dev.map_xy, xy, sizeof(float) * 2 * (size_t)M, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dev.map_cov, cov, sizeof(double) * 4 * (size_t)M, hipMemcpyHostToDevice));
    h->dev.M_map = M;
    return REKF_OK;
}

int rekf_handle_odometry(rekf_t *h, double t, double vx, double vy, double wz)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    if (t < h->time) return REKF_OK;                  // drop old data, cc:211-212
    if (h->opt.use_imu) return REKF_OK;               // cc:213-223: the use_imu branch is empty -- nothing happens
    // Predict(dt) on the host's pose mirror: no launch.  If the pose of the last scan has not been read back yet this waits for the
    // slots its last kernel stores (the reference's handler is synchronous anyway, and its node reads the pose after every scan,
    // src/ros_node.cc:514-515, so in that call pattern nothing is ever waited for).
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    h->vt[0] = vx; h->vt[1] = vy; h->vt[2] = wz;      // cc:216
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:217
    host_predict(h, a, h->mir_mu, h->mir_P, true);    // cc:218 Predict(dt)
    h->time = t;                                      // cc:219
    return REKF_OK;
}

// the scan a held scan's launch speculates for (struct rekf: SPECULATIVE MATCH)
struct NextScan { double t; const float *xy; int K; bool gps; };
static int process_scan(rekf_t *h, double t, const float *xy, int K, const double *gps_pose3, const NextScan *next);
// the held scan goes out (without a successor to speculate for)
static int flush_held(rekf_t *h)
{
    if (!h->held) return REKF_OK;
    h->held = false;
    const int rc = process_scan(h, h->held_t, h->held_xy, h->held_K, h->held_gps ? h->held_gps3 : nullptr, nullptr);
    if (rc != REKF_OK && !h->scan_committed) h->held = true;       // nothing has moved: the scan is still held, the caller's next call sends it again
    return rc;
}

int rekf_handle_observation(rekf_t *h, double t, const float *xy, int K, const double *gps_pose3)
{
    if (!h || K < 0 || (K > 0 && !xy)) return REKF_ERR_INVALID;
    if (K > REKF_MAX_OBS) return REKF_ERR_TOO_MANY_OBS;
    // a scan that can take part in the speculation pipeline: a whole scan (a filter that can still grow included: struct rekf, EARLY n; a
    // pre-loaded map whose covariances give its branch of the match a margin proof included: RekfDev::map_lip), handed over without a
    // pose read-back since the last scan
    const bool holdable = h->spec_enable && h->scan_launch && K >= 1 && K <= 32 && (h->dev.M_map == 0 || h->dev.map_lip >= 0.0) &&
                          !h->mir_valid && !h->prof_on;
    if (h->held) {
        if (!holdable) { FLUSH_HELD(h); return process_scan(h, t, xy, K, gps_pose3, nullptr); }
        const NextScan nx = {t, xy, K, gps_pose3 != nullptr};
        h->held = false;
        const int rc = process_scan(h, h->held_t, h->held_xy, h->held_K, h->held_gps ? h->held_gps3 : nullptr, &nx);
        if (rc != REKF_OK) {
            // The held scan failed; THIS call's scan has not been touched, and the error is this call's: hand it over again.  An error in
            // front of the host's bookkeeping (everything that can fail comes first, process_scan) leaves the held scan held -- the retry
            // sends both, nothing is applied twice, nothing is lost; behind it (a refused launch) the held scan counts as applied, like
            // any scan whose call fails there
            if (!h->scan_committed) h->held = true;
            return rc;
        }
    } else if (!holdable || !h->dd_pending) return process_scan(h, t, xy, K, gps_pose3, nullptr);
    h->held = true; h->held_t = t; h->held_K = K; h->held_gps = gps_pose3 != nullptr;
    if (gps_pose3) { h->held_gps3[0] = gps_pose3[0]; h->held_gps3[1] = gps_pose3[1]; h->held_gps3[2] = gps_pose3[2]; }
    std::memcpy(h->held_xy, xy, sizeof(float) * 2 * (size_t)K);
    return REKF_OK;
}

static int process_scan(rekf_t *h, double t, const float *xy, int K, const double *gps_pose3, const NextScan *next)
{
    const bool staged = K > REKF_MAX_OBS_DEV;                       // too many observations for the launch packet
    const bool blocks = 2 * K + (gps_pose3 ? 3 : 0) > 64;           // more innovation rows than one pass of k_mid takes
    h->scan_committed = false;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->early_valid) { int rce = learn_early_n(h); if (rce != REKF_OK) return rce; }      // (a growing filter: struct rekf, EARLY n)
    peek_n(h);                                        // whatever the device has published meanwhile tightens the bound on n, for free
    grid_take_note(h);
    if (K == 0) {                                     // cc:235-236: Predict only -- on the mirror, like an odometry message
        int rc = refresh_mirror(h);
        if (rc != REKF_OK) return rc;
        RekfFrontArgs a0;
        fill_front_args(h, a0, t - h->time);
        host_predict(h, a0, h->mir_mu, h->mir_P, true);
        h->time = t;                                  // cc:234
        h->last_scan_empty = true;
        return REKF_OK;
    }
    if (h->mir_valid && h->n_exact) h->n_det = h->n_ub;
    if (h->auto_grow && h->n_det + 2 * K > h->dev.n_max) {
        // room for K new reflectors is not certain.  Learn the exact n (waits for the publisher in flight: only when the BOUND says
        // so -- and a doubling leaves room for many scans), then re-reserve rather than let k_mid drop reflectors (cc:311-364 never does).
        int rc = refresh_mirror(h);
        if (rc != REKF_OK) return rc;
        if (h->n_ub + 2 * K > h->dev.n_max) {
            const int need = (h->n_ub - 3) / 2 + K;
            rc = rekf_reserve(h, need > 2 * h->max_landmarks ? need : 2 * h->max_landmarks);
            if (rc != REKF_OK && rc != REKF_ERR_UNSUPPORTED) return rc;     // (too large to address: fall back to the capacity flag)
        }
    }
    // ---- everything that can FAIL comes first, so that a failing call leaves the handle exactly as it found it: the host's counters
    // (scan parity, front-end target, growth bound, publisher tags), the pose mirror and the time all move only once nothing but
    // kernel launches is left (the ABI's promise: an error code, and the state stays valid -- hand the scan over again)
    ProfScope upd(h, REKF_K_UPDATE);                  // one bracket around the whole chain (per-update latency)
    if (h->inject_failure == 1) { h->inject_failure = 0; h->hip_error = "injected failure (staging)"; return REKF_ERR_HIP; }
    if (staged) {                                     // the scan does not fit the launch packet: through a pinned staging buffer into HBM
        if (h->obs_staging_busy) HIP_TRY(h, hipEventSynchronize(h->obs_staging_ev));   // the previous wide scan's copy (long done)
        std::memcpy(h->obs_staging, xy, sizeof(float) * 2 * (size_t)K);                 // the caller's buffer is free when we return
        HIP_TRY(h, hipMemcpyAsync(h->dev_obs, h->obs_staging, sizeof(float) * 2 * (size_t)K, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipEventRecord(h->obs_staging_ev, h->stream));
        h->obs_staging_busy = true;
    }
    // ONE LAUNCH PER SCAN (struct rekf): the held-back downdate runs beside this scan's k_mid, from the stored P into the other buffer, and
    // the mid role corrects what it gathers -- with n known exactly and no landmark augmentation pending on the stored P (a filter that
    // cannot grow, or one whose last scan is known to have appended nothing: struct rekf, EARLY n), whole scans on this side
    const bool fast = h->dd_pending && h->scan_launch && h->n_exact && !h->dd_aug && !blocks && !staged && K <= 32 &&
                      rekf_scan_launch_fits(h->n_ub, next ? next->K : K);
    // else the previous scan's downdate and this scan's front end go out as ONE launch (k_dd_front) -- also behind a read-back (the scan is
    // host-predicted: its front end is a match only, which hides under the downdate), unless the handle is exclusive: there the front end
    // runs inside k_mid's own grid and the downdate goes out first.  (Until round 5 every read-back sent the downdate out alone and
    // the front end as a launch of its own behind it: 8 us per update of a caller that reads the pose after every scan of a growing filter.)
    const bool with_dd = h->dd_pending && (fast || !h->mir_valid || !in_grid_ok(h)) &&
                         h->inject_failure != 2;          // (rekf_debug_inject_failure(2): the downdate goes out alone, and that launch "fails")
    if (!with_dd) {
        if (h->inject_failure == 2) { h->inject_failure = 0; h->hip_error = "injected failure (held-back downdate)"; return REKF_ERR_HIP; }
        int rcf = flush_dd(h);
        if (rcf != REKF_OK) return rcf;
    }
    // ---- from here on: host bookkeeping and launches only
    h->scan_committed = true;
    h->early_valid = false;                           // (set again below by a whole scan on a growing filter)
    const bool alone = in_grid_ok(h);                 // (in-launch hand-overs only on an exclusive, lone handle: struct rekf, EXCLUSIVE)
    h->last_scan_empty = false;
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:232 (dt may be negative, Q8)
    a.is_obs = 1;
    a.K = K;
    if (!staged) std::memcpy(a.obs, xy, sizeof(float) * 2 * (size_t)K);
    else a.obs_ext = h->dev_obs;
    if (gps_pose3) {
        a.has_gps = 1;
        a.gps[0] = gps_pose3[0]; a.gps[1] = gps_pose3[1]; a.gps[2] = gps_pose3[2];
    }
    if (h->mir_valid) {
        // the pose is known here: this scan's own Predict (cc:233) runs on the mirror too, and the front kernel receives the
        // predicted pose, the pose block and the composite of every predict since the device last saw P -- no motion model there
        host_predict(h, a, h->mir_mu, h->mir_P, true);
        put_host_prediction(h, a);
        h->lazy_pending = false; h->lazy_a = 0; h->lazy_b = 0;
    }
    h->mir_valid = false;                             // the update moves the pose: current again after the next read-back
    h->dev.n_known = h->n_exact ? h->n_ub : -1;
    // this scan's Predict travels through the control block (RekfCtl::pred): the front kernel writes the slot, k_mid and the scan's
    // (first) k_downdate2 apply it to what they read of P
    const int pred_slot = (int)(h->scan_count++ & 1);
    a.pred_slot = pred_slot; a.apply_pred = 1;
    a.pred_ix = (int)(h->scan_count & 3);             // (scan id mod 4: RekfCtl::pred, dmmax)
    h->dev.pred_ix = a.pred_ix;
    a.dd_par = pred_slot;                             // (RekfCtl::dd_queue: every k_mid zeroes the OTHER parity's counter for the next launch)
    // the front end counts matched observations (RekfCtl::front_count); the workgroup that reaches this scan's target compacts the
    // results into the record k_mid starts from (whole scans only: a wide scan goes through k_compact_wide)
    // SPECULATIVE MATCH (struct rekf): the previous launch has run this scan's front end -- no front end now; k_mid proves the record
    const bool use_spec = fast && h->spec_ready && h->spec_scan == (unsigned)h->scan_count && !a.host_pred;
    h->spec_ready = false;
    // a whole scan whose front end is a launch in front of k_mid leaves its raw results; k_mid compacts for itself (RekfFrontArgs::compact_in_mid).
    // (Decided further down -- where the front end goes -- but the count it would have added to belongs here.)
    const bool front_in_grid = !use_spec && alone &&
                               ((fast && !next) || (!fast && !with_dd && a.host_pred && !blocks && !staged && K <= 32));   // (as decided below)
    // (not when the front end rides inside k_dd_front: there the previous scan's downdate sets the launch's length, the election is hidden,
    // and k_mid would pay 0.9 us for the compaction)
    const bool cim = !use_spec && !blocks && !staged && K <= 32 && !front_in_grid && (fast || !with_dd);
    a.compact_in_mid = cim ? 1 : 0;
    // a host-predicted whole scan whose front end would be a launch of its own: k_mid matches it itself through the match grid
    const bool use_grid = h->grid_on && cim && a.host_pred && h->n_exact && h->grid_bucket != nullptr;
    if (!use_spec && !cim) h->front_total += (unsigned)K;    // (the front end counts the observations it matches; the speculative one has counted these)
    a.front_target = h->front_total;
    a.spec = use_spec ? 1 : 0;
    a.compact_in_front = blocks ? 0 : 1;
    a.cp_write = blocks ? 0 : 1;   // (whole scans leave their write-ahead correction: k_mid phase G)
    h->dev.pred_slot = -1;
    h->dev.post_slot = pred_slot;                     // (RekfCtl::post_C9: k_mid writes the scan's slot, the scan's downdate stores it)
    h->dev.P_out = h->dev.P;
    h->dev.kc_ub = round_up(2 * K + (gps_pose3 ? 3 : 0), 16);
    // one kernel of the call publishes pose, pose block, flags and the n the state will have once the k_augment behind the chain has run
    // (which changes none of the others): k_mid, or the first workgroup of the call's last downdate (struct rekf: WHO PUBLISHES)
    const bool aug = !h->full;
    if (aug) h->cum_growth += 2 * K;
    const int pub_seq = new_publisher(h);
    a.scan_id = (unsigned)h->scan_count;
    int front_wgs = 0;                                // (fast: the front end's workgroups inside the scan's launch)
    if (fast) {
        // the pending downdate stays pending until the k_mid launch below takes it along; the mid role sees it as a correction
        a.corr = 1; a.corr_pred = h->dd_dev.pred_slot; a.corr_post = h->dd_dev.post_slot; a.corr_pred_ix = h->dd_dev.pred_ix; a.corr_scan = h->dd_scan;
        if (use_spec) { /* nothing: the record is there */ }
        else if (alone && !next) front_wgs = K;
        else if (use_grid) { grid_ensure(h); a.grid_match = 1; }      // (k_mid matches the scan itself: struct rekf, MATCH GRID)
        else { ProfScope ps(h, REKF_K_FRONT); rekf_launch_front_mb(h->dev, a, h->n_ub, h->stream); }
    } else if (with_dd) {
        h->dd_pending = false;
        // the previous scan's augmentation: inside this scan's k_mid on an exclusive handle (its workgroup 0 appends the rows first thing;
        // whole scans on both sides); else by the LAST downdate workgroup of this launch to finish (RekfDev::aug_tail: the previous scan
        // was a whole scan, its record is in RekfCtl::augrec); else as k_augment right behind the downdate
        const bool inline_aug = alone && h->dd_aug && h->dd_aug_inline_ok && !blocks;
        const bool tail_aug = h->dd_aug && !inline_aug && h->dd_aug_inline_ok;
        a.aug_pending = h->dd_aug ? (tail_aug ? 2 : 1) : 0;
        h->dd_dev.P_out = h->dd_dev.P;                // (in place)
        h->dd_dev.aug_tail = tail_aug ? 1 + ((pred_slot ^ 1) & 1) : 0;
        { ProfScope ps(h, REKF_K_DOWNDATE); rekf_launch_dd_front(h->dd_dev, h->dd_n_ub, h->dev, a, h->stream); }
        h->dd_dev.aug_tail = 0;
        if (h->dd_aug && !inline_aug && !tail_aug) { ProfScope ps(h, REKF_K_AUGMENT); rekf_launch_augment(h->dd_dev, h->dd_aug_args, h->stream); }
        h->dd_aug = false;
        a.aug_pending = 0;
        a.aug_in_mid = inline_aug ? 1 : 0;
    } else if (alone && a.host_pred && !blocks && !staged && K <= 32) {
        // behind a pose read-back the front end is a match only (pose, cos / sin, pose block go by value): it runs as the first
        // workgroups of k_mid's own grid, one observation each, and hands the record over inside the launch
        a.front_in_mid = K;
    } else if (use_grid) {
        grid_ensure(h); a.grid_match = 1;             // (k_mid matches the scan itself: struct rekf, MATCH GRID)
    } else {
        ProfScope ps(h, REKF_K_FRONT);
        rekf_launch_front_mb(h->dev, a, h->n_ub, h->stream);
    }
    h->time = t;                                      // cc:234
    const int n_ub = h->n_ub;
    const int m_ub = 2 * K + (gps_pose3 ? 3 : 0);
    h->dev.mu_lin = h->dev.mu;
    // lazy downdate (struct rekf): the scan's last downdate -- and its k_augment -- go out with the next call.  (Not for a staged scan:
    // its k_augment reads the observations from a device buffer the next staged scan overwrites.)
    const bool hold_back = !staged;
    hipError_t enq_err = hipSuccess;
    const bool early_pub = h->pose_read_since_scan;   // (struct rekf: WHO PUBLISHES)
    h->pose_read_since_scan = false;
    auto downdate = [&](bool first, bool last) {
        RekfDev dd = h->dev;
        dd.pred_slot = first ? pred_slot : -1;        // the scan's first downdate commits its Predict
        if (last && !early_pub) { dd.pub = h->host_slots_dev; dd.pub_seq = pub_seq; dd.pub_aug = aug ? 1 : 0; }     // ... its last one publishes (at its start)
        flip_panels(h);                               // (dd has its own copy of the view: the next k_mid writes the other set)
        if (last && hold_back) { h->dd_pending = true; h->dd_dev = dd; h->dd_n_ub = n_ub; h->dd_scan = a.scan_id; return; }     // (lazy downdate: with the next call)
        ProfScope ps(h, REKF_K_DOWNDATE);
        rekf_launch_downdate(dd, n_ub, h->stream);
    };
    if (blocks) {
        // More than 32 observations (the reference has no limit, cc:397): matched once, then the joint update runs as
        // exact block steps of at most 32 pairs through the same two kernels (k_mid explains why that is the same
        // update).  The host cannot know how many observations matched, so it enqueues ceil(K / stride) steps; a step
        // past the last pair only carries the mean over and adds zero panels.
        const int stride = gps_pose3 ? 30 : 32;
        a.pair_stride = stride;
        h->dev.kc_ub = 64;
        rekf_launch_compact_wide(h->dev, a, h->stream);
        enq_err = hipMemcpyAsync(h->dev_mu_lin, h->dev.mu, sizeof(double) * (size_t)h->dev.ld, hipMemcpyDeviceToDevice, h->stream);   // (checked behind the chain)
        for (int p0 = 0; p0 < K; p0 += stride) {
            a.pair0 = p0;
            a.apply_pred = (p0 == 0) ? 1 : 0;
            h->dev.mu_lin = h->dev_mu_lin;
            {
                RekfDev dm = h->dev;
                if (early_pub && p0 + stride >= K) { dm.pub = h->host_slots_dev; dm.pub_seq = pub_seq; }
                ProfScope ps(h, REKF_K_MID);
                rekf_launch_mid(dm, a, n_ub, 64, aug, h->stream);
            }
            std::swap(h->dev.mu, h->dev.mu_out); h->dev.grid_par ^= 1;
            downdate(p0 == 0, p0 + stride >= K);      // the last step commits the final pose
        }
        a.pair0 = -1;
        h->dev.mu_lin = h->dev.mu;
    } else {
        // the whole innovation fits one pass: gather + solve + gain as ONE launch (k_mid), which leaves the updated
        // mean in the other mean buffer
        {
            RekfDev dm = h->dev;
            if (early_pub) { dm.pub = h->host_slots_dev; dm.pub_seq = pub_seq; }
            if (aug) {
                // a filter that can still grow: workgroup 0 publishes the n this scan leaves as soon as its match record is final, for the
                // next scan's call (struct rekf, EARLY n), and leaves the scan's augmentation record whatever form the launch has
                dm.early = h->host_slots_dev + 13; dm.early_seq = ++h->early_seq; dm.aug_write = 1;
                h->early_valid = true; h->early_n_before = h->n_exact ? n_ub : -1;
            }
            ProfScope ps(h, REKF_K_MID);
            if (fast) {
                // [front end |] mid role | the held-back downdate, from the stored P into the other buffer -- which then IS the stored P
                h->dd_pending = false;
                RekfDev ddv = h->dd_dev;
                ddv.P_out = h->P_alt;
                // ... and, for a caller that enqueues scan after scan, the NEXT scan's front end, speculatively (struct rekf)
                RekfFrontArgs an;
                const bool with_next = next && front_wgs == 0 && h->spec_enable;
                if (with_next) {
                    fill_front_args(h, an, next->t - t);
                    an.is_obs = 1; an.K = next->K;
                    std::memcpy(an.obs, next->xy, sizeof(float) * 2 * (size_t)next->K);
                    an.has_gps = next->gps ? 1 : 0;
                    an.pred_slot = pred_slot ^ 1; an.pred_ix = (a.pred_ix + 1) & 3; an.scan_id = a.scan_id + 1u;
                    an.compact_in_front = 1;
                    an.prev_dt = a.dt; an.prev_vt[0] = a.vt[0]; an.prev_vt[1] = a.vt[1]; an.prev_vt[2] = a.vt[2];
                    h->front_total += (unsigned)next->K;
                    an.front_target = h->front_total;
                    h->spec_ready = true; h->spec_scan = an.scan_id;
                }
                (void)rekf_launch_scan(ddv, dm, a, n_ub, m_ub, front_wgs, with_next ? &an : nullptr, h->stream);
                std::swap(h->dev.P, h->P_alt);
                h->dev.P_out = h->dev.P;
            } else rekf_launch_mid(dm, a, n_ub, m_ub, aug || a.aug_in_mid != 0, h->stream);
        }
        std::swap(h->dev.mu, h->dev.mu_out); h->dev.grid_par ^= 1;
        downdate(true, true);
    }
    h->last_m_ub = m_ub;
    // the state only grows: once it is known full, k_augment can never have work again
    // (k_mid / k_compact_wide drop the extra reflectors and raise REKF_FLAG_CAPACITY)
    if (aug) {
        if (h->dd_pending) { h->dd_aug = true; h->dd_aug_args = a; h->dd_aug_args.aug_in_mid = 0; h->dd_aug_inline_ok = !blocks && !staged; }      // (held back with the scan's downdate)
        else {
            ProfScope ps(h, REKF_K_AUGMENT);
            rekf_launch_augment(h->dev, a, h->stream);
        }
    }
    { ProfScope ps(h, REKF_K_EMPTY); }
    if (aug) {
        // the scan may have appended up to K reflectors; the exact n stays on the device until it is published
        const int grown = n_ub + 2 * K;
        h->n_ub = grown > h->dev.n_max ? h->dev.n_max : grown;
        h->n_det = (h->n_det + 2 * K > h->dev.n_max) ? h->dev.n_max : h->n_det + 2 * K;
        h->n_exact = false;
    }
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = enq_err;
    if (h->inject_failure == 3) { h->inject_failure = 0; le = hipErrorLaunchFailure; }
    if (le != hipSuccess) {
        // a launch was refused: which kernels of the chain run is no longer what the host planned.  Let the stream drain and take the
        // counters the kernels keep (n, the front end's count) from the device, so that the next scan starts from what really happened
        h->hip_error = std::string("kernel launch: ") + hipGetErrorString(le);
        (void)hipStreamSynchronize(h->stream);
        if (hipMemcpy(h->ctl_staging, h->dev.ctl, sizeof(RekfCtl), hipMemcpyDeviceToHost) == hipSuccess) {
            h->front_total = h->ctl_staging->front_count;
            h->n_ub = h->ctl_staging->n; h->n_det = h->n_ub; h->n_exact = true; h->full = h->n_ub >= h->dev.n_max;
        }
        (void)hipGetLastError();
        h->pub_valid = false; h->mir_valid = false; h->early_valid = false;
        return REKF_ERR_HIP;
    }
    return REKF_OK;
}

int rekf_predict_state(rekf_t *h, double t, double mu3[3], double sigma3x3[9])
{
    if (!h || !mu3) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:100
    double m3[3], P9[9];
    for (int q = 0; q < 3; ++q) m3[q] = h->mir_mu[q];
    for (int q = 0; q < 9; ++q) P9[q] = h->mir_P[q];
    host_predict(h, a, m3, P9, false);                // non-mutating: on a copy of the mirror
    for (int q = 0; q < 3; ++q) mu3[q] = m3[q];
    if (sigma3x3) for (int q = 0; q < 9; ++q) sigma3x3[q] = P9[q];
    return REKF_OK;
}

int rekf_get_time(rekf_t *h, double *t)
{
    if (!h || !t) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    *t = h->time;
    return REKF_OK;
}

int rekf_get_pose(rekf_t *h, double *t, double mu3[3], double sigma3x3[9])
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    // the kernels of the last call have published (or are about to publish) the pose they committed; between scans the mirror
    // simply is the pose -- either way no copy engine and no stream wait
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    if (t) *t = h->time;
    if (mu3) for (int q = 0; q < 3; ++q) mu3[q] = h->mir_mu[q];
    if (sigma3x3) for (int q = 0; q < 9; ++q) sigma3x3[q] = h->mir_P[q];
    return REKF_OK;
}

int rekf_get_marker_ellipses(rekf_t *h, double *out5, int cap, int *count)
{
    if (!h || !count || cap < 0 || (cap > 0 && !out5)) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    HIP_TRY(h, hipSetDevice(h->device));
    { int rcf = flush_dd(h); if (rcf != REKF_OK) return rcf; }
    const int lim = cap < h->max_landmarks ? cap : h->max_landmarks;
    rekf_launch_ellipses(h->dev, h->dev_ell, lim, h->stream);       // (landmark blocks and means only: pending predicts do not touch them)
    int rc = pull_ctl(h);                              // synchronises the stream; n is exact afterwards
    if (rc != REKF_OK) return rc;
    const int L = (h->ctl_staging->n - 3) / 2;
    const int k = L < lim ? L : lim;
    if (k > 0) HIP_TRY(h, hipMemcpy(out5, h->dev_ell, sizeof(double) * 5 * (size_t)k, hipMemcpyDeviceToHost));
    *count = k;
    return (L > cap) ? REKF_ERR_BUFFER : REKF_OK;
}

int rekf_get_n(rekf_t *h, int *n)
{
    if (!h || !n) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = refresh_mirror(h);                        // n arrives with the pose slots
    if (rc != REKF_OK) return rc;
    if (!h->n_exact) { rc = pull_ctl(h); if (rc != REKF_OK) return rc; }
    *n = h->n_ub;
    return REKF_OK;
}

int rekf_get_state(rekf_t *h, double *t, int *n_out, double *mu, long mu_cap, double *sigma, long sigma_cap)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = flush_lazy(h);
    if (rc != REKF_OK) return rc;
    rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    const int n = h->ctl_staging->n;
    if (t) *t = h->time;
    if (n_out) *n_out = n;
    if (mu) {
        if (mu_cap < n) return REKF_ERR_BUFFER;
        HIP_TRY(h, hipMemcpyAsync(mu, h->dev.mu, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    }
    if (sigma) {
        if (sigma_cap < (long)n * n) return REKF_ERR_BUFFER;
        HIP_TRY(h, hipMemcpy2DAsync(sigma, sizeof(double) * n, h->dev.P, sizeof(double) * h->dev.ld,
                                    sizeof(double) * n, n, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (sigma) mirror_lower(sigma, n);
    return REKF_OK;
}

int rekf_set_state(rekf_t *h, double t, int n, const double *mu, const double *sigma, const double *vt3)
{
    if (!h || !mu || !sigma || n < 3 || n > h->dev.n_max || ((n - 3) & 1)) return REKF_ERR_INVALID;
    if (h->held) (void)flush_held(h);
    h->held = false;                                              // (a held scan that could not be sent belonged to the state being replaced)
    h->early_valid = false;
    h->grid_dev_valid = false; h->dev.grid_bucket = nullptr; h->dev.grid_p0 = nullptr; h->dev.grid_note = nullptr;
    h->spec_ready = false;
    h->pub_valid = false;
    for (auto &r : h->pub_ring) r = {0, 0};                       // an n published before this call says nothing about the new state
    h->lazy_pending = false; h->lazy_a = 0; h->lazy_b = 0;       // whatever was pending belonged to the state being replaced
    h->mir_valid = false;
    h->dd_pending = false; h->dd_aug = false;                      // (a held-back downdate belonged to the state being replaced)
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int ld = h->dev.ld;
    HIP_TRY(h, hipMemsetAsync(h->dev.P, 0, sizeof(double) * (size_t)ld * ld, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->dev.mu, 0, sizeof(double) * ld, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->dev.mu, mu, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpy2DAsync(h->dev.P, sizeof(double) * ld, sigma, sizeof(double) * n, sizeof(double) * n, n,
                                hipMemcpyHostToDevice, h->stream));
    std::memset(h->ctl_staging, 0, sizeof(RekfCtl));
    h->ctl_staging->n = n;
    h->ctl_staging->front_count = h->front_total;                  // (the front end's count of matched observations goes on)
    HIP_TRY(h, hipMemcpyAsync(h->dev.ctl, h->ctl_staging, sizeof(RekfCtl), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->time = t;
    h->n_ub = n;
    h->n_det = n;
    h->n_exact = true;
    h->full = n >= h->dev.n_max;
    h->flags_last = 0;
    h->last_scan_empty = false;
    if (vt3) { h->vt[0] = vt3[0]; h->vt[1] = vt3[1]; h->vt[2] = vt3[2]; }
    for (int q = 0; q < 3; ++q) h->mir_mu[q] = mu[q];
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) h->mir_P[i + 3 * j] = rekf_plower(sigma, n, i, j);   // (the lower triangle is what counts)
    h->mir_valid = true;
    return REKF_OK;
}

int rekf_get_last_match(rekf_t *h, int *n_state, int *state_pairs, int *n_map, int *map_pairs, int *n_new,
                        int *new_ids)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    if (h->last_scan_empty) {                          // cc:235-236: the record of an empty scan is empty (kept on the host)
        if (n_state) *n_state = 0;
        if (n_map) *n_map = 0;
        if (n_new) *n_new = 0;
        return REKF_OK;
    }
    const RekfCtl *c = h->ctl_staging;
    if (n_state) *n_state = c->n_state;
    if (n_map) *n_map = c->n_map;
    if (n_new) *n_new = c->n_new;
    if (state_pairs) std::memcpy(state_pairs, c->state_pairs, sizeof(int) * 2 * (size_t)c->n_state);
    if (map_pairs) std::memcpy(map_pairs, c->map_pairs, sizeof(int) * 2 * (size_t)c->n_map);
    if (new_ids) std::memcpy(new_ids, c->new_ids, sizeof(int) * (size_t)c->n_new);
    return REKF_OK;
}

int rekf_sync(rekf_t *h)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    rc = flush_lazy(h);                                // predicts the host applied to its mirror only: the device state catches up (one small kernel, only when pending)
    if (rc != REKF_OK) return rc;
    // the published pose can arrive a few microseconds before the last workgroups of k_downdate2 are through: Sync means done
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int flags = h->flags_last;
    if (flags) {
        int zero = 0;
        HIP_TRY(h, hipMemcpy(&h->dev.ctl->err, &zero, sizeof(int), hipMemcpyHostToDevice));
        h->flags_last = 0;
    }
    return device_flags_to_code(flags);
}

int rekf_get_flags(rekf_t *h, int *flags)
{
    if (!h || !flags) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    *flags = h->flags_last;
    return REKF_OK;
}

int rekf_predict_state_full(rekf_t *h, double t, double *time_out, int *n_out, double *mu, long mu_cap, double *sigma,
                            long sigma_cap)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = refresh_mirror(h);
    if (rc != REKF_OK) return rc;
    rc = flush_lazy(h);
    if (rc != REKF_OK) return rc;
    rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    const int n = h->ctl_staging->n;
    const int ld = h->dev.ld;
    if (time_out) *time_out = h->time;                // `State result = state_` keeps the state's time (cc:99)
    if (n_out) *n_out = n;
    if ((mu && mu_cap < n) || (sigma && sigma_cap < (long)n * n)) return REKF_ERR_BUFFER;
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:100
    double m3[3], P9[9];                               // pose and pose block: on a copy of the mirror, like rekf_predict_state
    for (int q = 0; q < 3; ++q) m3[q] = h->mir_mu[q];
    for (int q = 0; q < 9; ++q) P9[q] = h->mir_P[q];
    const Motion mo = host_predict(h, a, m3, P9, false);
    a.host_pred = 1;
    a.pre_ab[0] = mo.a; a.pre_ab[1] = mo.b;
    rekf_launch_predict_rows(h->dev, a, h->dev_pred, h->stream);
    std::vector<double> pred(4 * (size_t)ld);
    HIP_TRY(h, hipMemcpyAsync(pred.data(), h->dev_pred, sizeof(double) * pred.size(), hipMemcpyDeviceToHost, h->stream));
    if (mu) HIP_TRY(h, hipMemcpyAsync(mu, h->dev.mu, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    if (sigma)
        HIP_TRY(h, hipMemcpy2DAsync(sigma, sizeof(double) * n, h->dev.P, sizeof(double) * ld, sizeof(double) * n, n,
                                    hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // layout of pred: row0[ld] | row1[ld] | col0[ld] | col1[ld]
    const double *row0 = pred.data(), *row1 = row0 + ld, *col0 = row1 + ld, *col1 = col0 + ld;
    if (mu) for (int i = 0; i < 3; ++i) mu[i] = m3[i];
    if (sigma) {
        mirror_lower(sigma, n);
        for (int c = 3; c < n; ++c) {
            sigma[0 + (size_t)c * n] = row0[c]; sigma[1 + (size_t)c * n] = row1[c];
            sigma[c + (size_t)0 * n] = col0[c]; sigma[c + (size_t)1 * n] = col1[c];
        }
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) sigma[i + (size_t)j * n] = P9[i + 3 * j];
    }
    return REKF_OK;
}

int rekf_profile_enable(rekf_t *h, int on)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    if (!on) { int rc = prof_flush(h); if (rc != REKF_OK) return rc; }
    h->prof_on = on != 0;
    h->prof_mask = on;
    return REKF_OK;
}

int rekf_profile_read(rekf_t *h, int k, double *total_us, long *count)
{
    if (!h || k < 0 || k >= REKF_K_COUNT) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = prof_flush(h);
    if (rc != REKF_OK) return rc;
    if (total_us) *total_us = h->prof_total_us[k];
    if (count) *count = h->prof_count[k];
    return REKF_OK;
}

int rekf_profile_reset(rekf_t *h)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = prof_flush(h);
    if (rc != REKF_OK) return rc;
    for (int k = 0; k < REKF_K_COUNT; ++k) { h->prof_total_us[k] = 0; h->prof_count[k] = 0; }
    h->prof_update_us.clear();
    return REKF_OK;
}

int rekf_profile_samples(rekf_t *h, float *out_us, long cap, long *count)
{
    if (!h || !count || cap < 0 || (cap > 0 && !out_us)) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = prof_flush(h);
    if (rc != REKF_OK) return rc;
    const long have = (long)h->prof_update_us.size();
    const long k = have < cap ? have : cap;
    if (k > 0) std::memcpy(out_us, h->prof_update_us.data(), sizeof(float) * (size_t)k);
    *count = have;
    return REKF_OK;
}

void *rekf_stream(rekf_t *h) { return h ? (void *)h->stream : nullptr; }

/* measurement hook: launch ONE kernel of the chain `reps` times back to back on the handle's
 * stream (operating on whatever the last observation left in the scratch buffers) and return the
 * average device time per launch in microseconds (hipEvents).  The state is NOT meaningful
 * afterwards (P -= K HP applied repeatedly): callers snapshot/restore with get/set_state. */
int rekf_debug_time_kernel(rekf_t *h, int kernel, int reps, int ablate, double *avg_us)
{
    if (!h || !avg_us || reps < 1) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    if (kernel != REKF_K_DOWNDATE || ablate != 0) return REKF_ERR_INVALID;      // the only kernel this hook knows; `ablate` is reserved
    int rc = flush_lazy(h);
    if (rc != REKF_OK) return rc;
    rc = pull_ctl(h);                                  // n exact: the launches below carry it
    if (rc != REKF_OK) return rc;
    h->pub_valid = false;
    h->mir_valid = false;                              // P is meaningless afterwards
    RekfDev dev = h->dev;
    dev.dbg = 0;
    dev.pub = nullptr;
    dev.pred_slot = -1;
    dev.n_known = h->n_ub;
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t a, b;
    HIP_TRY(h, hipEventCreate(&a));
    HIP_TRY(h, hipEventCreate(&b));
    const int n_ub = h->n_ub;
    for (int i = 0; i < 3; ++i) rekf_launch_downdate(dev, n_ub, h->stream);
    HIP_TRY(h, hipEventRecord(a, h->stream));
    for (int i = 0; i < reps; ++i) rekf_launch_downdate(dev, n_ub, h->stream);
    HIP_TRY(h, hipEventRecord(b, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, a, b));
    *avg_us = 1e3 * (double)ms / reps;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return REKF_OK;
}

/* debug builds only: the 8 scratch counters kernels may fill (see REKF_DEBUG_TIMING) */
int rekf_debug_counters(rekf_t *h, long long out8[32])
{
    if (!h || !out8) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    for (int i = 0; i < 32; ++i) out8[i] = h->ctl_staging->dbg[i];
#ifndef REKF_DEBUG_TIMING
    out8[24] = (long long)h->ctl_staging->dd_queue[0] + (long long)h->ctl_staging->dd_queue[1];    // work items the in-launch downdate roles of the last two launches asked for
    out8[17] = h->grid_builds;                         // k_grid_build launches so far (host count)
    out8[16] = h->grid_on ? 1 : 0;
#endif
    return REKF_OK;
}

/* MATCH GRID test hook: on = 0 switches the grid match off (the scan's front end is a launch again: the twin of the parity tests);
 * drift_limit > 0 replaces the drift bound (metres; tiny values make every update invalidate the grid: rebuilds and sweep fall-backs);
 * mask >= 0 shrinks the hash table to mask + 1 buckets (mask = 2^k - 1; overflow: the library must stop using the grid by itself). */
int rekf_debug_set_grid(rekf_t *h, int on, double drift_limit, int mask)
{
    if (!h || (mask >= 0 && (mask & (mask + 1)) != 0)) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    h->grid_on = on != 0;
    h->grid_drift = drift_limit > 0.0 ? (float)drift_limit : REKF_GRID_DRIFT;
    h->grid_mask_override = mask;
    h->grid_dev_valid = false;
    h->dev.grid_bucket = nullptr; h->dev.grid_p0 = nullptr; h->dev.grid_note = nullptr;
    return REKF_OK;
}

int rekf_debug_inject_failure(rekf_t *h, int stage)
{
    if (!h || stage < 0 || stage > 3) return REKF_ERR_INVALID;
    h->inject_failure = stage;
    return REKF_OK;
}

int rekf_device_layout(rekf_t *h, int *ld, int *n_max, void **P_dev, void **mu_dev)
{
    if (!h) return REKF_ERR_INVALID;
    FLUSH_HELD(h);
    // the caller is about to look at P / mu: the held-back downdate AND the predicts the host has applied to its mirror only go out
    { int rcf = flush_lazy(h); if (rcf != REKF_OK) return rcf; }
    if (ld) *ld = h->dev.ld;
    if (n_max) *n_max = h->dev.n_max;
    if (P_dev) *P_dev = h->dev.P;
    if (mu_dev) *mu_dev = h->dev.mu;       // (double-buffered: k_mid alternates between two buffers, this is the current one)
    return REKF_OK;
}

}  // extern "C"
