// rekf_api.hip -- host side of the C ABI declared in include/rekf.h.
//
// Owns the HIP device buffers and the stream of one filter handle and turns
// each reference method (reflector_ekf_slam.cc) into an asynchronous kernel
// chain.  Time bookkeeping (State::time, the "drop old odometry" test at
// reflector_ekf_slam.cc:211-212) is pure host data and stays on the host; every
// size that depends on device results (n, m, the match lists) stays on the
// device so no call here waits for the GPU except the getters.
#include "../../include/rekf.h"
#include "ekf_dev.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
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
    double time;
    double vt[3];
    int n_ub;                  // host upper bound of the device-resident n
    int last_m_ub = 64;        // innovation-row bound of the last scan (sizes the k_solve launch)
    bool full;                 // a readback showed n == n_max: no landmark can ever be added again
    bool n_exact = true;       // n_ub IS the device's n (nothing that can append landmarks was enqueued since it was read back)
    double *pose_staging;      // pinned, 12 doubles
    RekfHostSlot *host_slots;  // pinned + mapped: 16 tagged slots the pose kernels store into (null: copy-engine path)
    RekfHostSlot *host_slots_dev;
    int slot_seq;
    bool pose_read;            // GetPose / Sync / GetFlags was called since the last odometry message: the caller reads the pose at odometry rate
    bool pub_valid;            // the slots hold (or will hold, once the enqueued kernels have run) the CURRENT pose under tag pub_seq
    int pub_seq;
    double *dev_out12;         // device scratch for k_predict_pose
    double *dev_ell;           // device scratch for k_ellipses (5 doubles per landmark of capacity)
    RekfCtl *ctl_staging;      // pinned copy of the control block
    std::string hip_error;
    int flags_seen = 0;        // sticky device flags already reported on stderr
    double *dev_pred;          // device scratch for k_predict_rows (4 * ld + 12 doubles)
    float *dev_obs = nullptr;  // a wide scan's observations (REKF_MAX_OBS_WIDE x 2 floats)
    double *dev_mu_lin = nullptr;   // the mean a wide scan is linearised at (copy taken before its first block step)
    // profiling
    bool prof_on;
    int prof_mask;
    std::vector<ProfSlot> prof_slots;
    size_t prof_used;
    double prof_total_us[REKF_K_COUNT];
    long prof_count[REKF_K_COUNT];
    std::vector<float> prof_update_us;   // every REKF_K_UPDATE reading since the last reset
};

namespace {

#define HIP_TRY(h, expr)                                                         \
    do {                                                                         \
        hipError_t e_ = (expr);                                                  \
        if (e_ != hipSuccess) {                                                  \
            if (h) (h)->hip_error = std::string(#expr) + ": " + hipGetErrorString(e_); \
            return REKF_ERR_HIP;                                                 \
        }                                                                        \
    } while (0)

int round_up(int x, int q) { return (x + q - 1) / q * q; }

// wait until the `count` slots from `first` carry `seq` (a kernel on the handle's stream stores them).  Falls back to a stream
// synchronisation when the stream drains without them (an earlier kernel failed) and gives up after five seconds.
template <class H> int wait_slots(H *h, int first, int count, int seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (int k = first; k < first + count; ++k) {
        const int *tag = &h->host_slots[k].seq;
        while (__atomic_load_n(tag, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0xfffffu) == 0) {
                if (hipStreamQuery(h->stream) != hipErrorNotReady) {
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    if (__atomic_load_n(tag, __ATOMIC_ACQUIRE) == seq) break;
                    h->hip_error = "a pose kernel finished without publishing its result";
                    return REKF_ERR_HIP;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    h->hip_error = "no pose result after 5 s";
                    return REKF_ERR_HIP;
                }
            }
        }
    }
    return REKF_OK;
}

int prof_flush(rekf_t *h)
{
    if (h->prof_used == 0) return REKF_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->prof_used; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_slots[i].a, h->prof_slots[i].b));
        h->prof_total_us[h->prof_slots[i].kernel] += 1e3 * (double)ms;
        h->prof_count[h->prof_slots[i].kernel] += 1;
        if (h->prof_slots[i].kernel == REKF_K_UPDATE && h->prof_update_us.size() < (1u << 20))
            h->prof_update_us.push_back(1e3f * ms);
    }
    h->prof_used = 0;
    return REKF_OK;
}

struct ProfScope {
    rekf_t *h;
    ProfSlot *slot;
    ProfScope(rekf_t *h_, int kernel) : h(h_), slot(nullptr)
    {
        if (!h->prof_on || !((h->prof_mask >> kernel) & 1)) return;
        if (h->prof_used == h->prof_slots.size()) {
            if (h->prof_slots.size() >= 65536) {
                prof_flush(h);
            } else {
                ProfSlot s;
                if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return;
                s.kernel = kernel;
                h->prof_slots.push_back(s);
            }
        }
        slot = &h->prof_slots[h->prof_used++];
        slot->kernel = kernel;
        (void)hipEventRecord(slot->a, h->stream);
    }
    ~ProfScope()
    {
        if (slot) (void)hipEventRecord(slot->b, h->stream);
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
    if (fresh & REKF_FLAG_SINGULAR)
        std::fprintf(stderr, "rekf: innovation covariance was not positive definite in some scan; the update was applied as computed\n");
    h->flags_seen |= flags;
}

// Synchronise and refresh the host copy of the control block.
// the next launches publish the pose they commit (k_front; k_mid + k_downdate2) under a fresh tag
int begin_publish(rekf_t *h)
{
    if (!h->host_slots) { h->pub_valid = false; return 0; }
    h->dev.pub = h->host_slots_dev;
    h->dev.pub_seq = ++h->slot_seq;
    return h->dev.pub_seq;
}
void end_publish(rekf_t *h, int seq)
{
    h->dev.pub = nullptr;
    if (seq) { h->pub_valid = true; h->pub_seq = seq; }
}

int pull_ctl(rekf_t *h)
{
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->ctl_staging, h->dev.ctl, sizeof(RekfCtl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_ub = h->ctl_staging->n;
    h->n_exact = true;
    h->full = h->ctl_staging->n >= h->dev.n_max;
    report_flags(h, h->ctl_staging->err);
    return REKF_OK;
}

// n and the sticky flags alone (rekf_sync, rekf_get_flags, rekf_get_n): one small kernel behind everything enqueued so far stores
// them into pinned host memory, polled by tag -- not the 7 KB control block through a copy engine and a stream wait (31 us on an
// idle stream; this: ~6 us).  Everything enqueued before it has finished when its slot arrives (one in-order stream).
int pull_n_flags(rekf_t *h, int *flags)
{
    if (!h->host_slots) {
        int rc = pull_ctl(h);
        if (rc == REKF_OK && flags) *flags = h->ctl_staging->err;
        return rc;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    h->pose_read = true;
    int seq = h->pub_seq;
    if (!h->pub_valid) {                              // nothing in flight publishes: one small kernel does
        seq = ++h->slot_seq;
        rekf_launch_publish_pose(h->dev, h->host_slots_dev, seq, h->stream);
        HIP_TRY(h, hipGetLastError());
        h->pub_valid = true; h->pub_seq = seq;
    }
    int rc = wait_slots(h, 0, 13, seq);               // all of them: the last kernels of the call store different slots
    if (rc != REKF_OK) return rc;
    const int n = (int)h->host_slots[12].v, err = h->host_slots[12].aux;
    h->n_ub = n;
    h->n_exact = true;
    h->full = n >= h->dev.n_max;
    report_flags(h, err);
    if (flags) *flags = err;
    return REKF_OK;
}

}  // namespace

extern "C" {

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
    case REKF_ERR_UNSUPPORTED: return "unsupported option";
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
    h->prof_on = false;
    h->prof_mask = -1;
    h->prof_used = 0;
    for (int k = 0; k < REKF_K_COUNT; ++k) { h->prof_total_us[k] = 0; h->prof_count[k] = 0; }
    h->stream = nullptr;
    h->pose_staging = nullptr;
    h->host_slots = nullptr; h->host_slots_dev = nullptr; h->slot_seq = 0; h->pub_valid = false; h->pub_seq = 0; h->pose_read = false;
    h->ctl_staging = nullptr;
    h->dev_out12 = nullptr;
    h->dev_ell = nullptr;
    h->dev_pred = nullptr;
    std::memset(&h->dev, 0, sizeof(h->dev));

    const int n_max = 3 + 2 * max_landmarks;
    const int ld = round_up(n_max, 64);
    int rc = [&]() -> int {
        HIP_TRY(h, hipSetDevice(device));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        HIP_TRY(h, hipMalloc(&h->dev.ctl, sizeof(RekfCtl)));
        HIP_TRY(h, hipMalloc(&h->dev.mu, sizeof(double) * ld));
        HIP_TRY(h, hipMalloc(&h->dev.mu_out, sizeof(double) * ld));
        HIP_TRY(h, hipMalloc(&h->dev.P, sizeof(double) * (size_t)ld * ld));
        HIP_TRY(h, hipMalloc(&h->dev.HPt, sizeof(double) * (size_t)ld * REKF_PANEL_COLS));
        HIP_TRY(h, hipMalloc(&h->dev.Kn, sizeof(double) * (size_t)ld * REKF_PANEL_COLS));
        HIP_TRY(h, hipMalloc(&h->dev.KnB, sizeof(double) * REKF_STRIP_MAX * REKF_MR_PAD));
        HIP_TRY(h, hipMalloc(&h->dev.HPtB, sizeof(double) * REKF_STRIP_MAX * REKF_MR_PAD));
        HIP_TRY(h, hipMalloc(&h->dev_out12, sizeof(double) * 16));
        HIP_TRY(h, hipMalloc(&h->dev_pred, sizeof(double) * (4 * (size_t)ld + 16)));
        HIP_TRY(h, hipMalloc(&h->dev_obs, sizeof(float) * 2 * REKF_MAX_OBS_WIDE));
        HIP_TRY(h, hipMalloc(&h->dev_mu_lin, sizeof(double) * ld));
        HIP_TRY(h, hipMalloc(&h->dev_ell, sizeof(double) * 5 * (size_t)(max_landmarks > 0 ? max_landmarks : 1)));
        HIP_TRY(h, hipHostMalloc(&h->pose_staging, sizeof(double) * 16));
        if (hipHostMalloc(&h->host_slots, sizeof(RekfHostSlot) * 16, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            std::memset(h->host_slots, 0, sizeof(RekfHostSlot) * 16);
            void *dv = nullptr;
            if (hipHostGetDevicePointer(&dv, h->host_slots, 0) == hipSuccess) h->host_slots_dev = (RekfHostSlot *)dv;
            else { (void)hipHostFree(h->host_slots); h->host_slots = nullptr; }
        }
        (void)hipGetLastError();
        HIP_TRY(h, hipHostMalloc(&h->ctl_staging, sizeof(RekfCtl)));
        h->dev.ld = ld;
        h->dev.n_max = n_max;
        h->dev.M_map = 0;
        HIP_TRY(h, hipMemsetAsync(h->dev.ctl, 0, sizeof(RekfCtl), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.mu, 0, sizeof(double) * ld, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.mu_out, 0, sizeof(double) * ld, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.P, 0, sizeof(double) * (size_t)ld * ld, h->stream));   // cc:10-11
        HIP_TRY(h, hipMemsetAsync(h->dev.HPt, 0, sizeof(double) * (size_t)ld * REKF_PANEL_COLS, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.Kn, 0, sizeof(double) * (size_t)ld * REKF_PANEL_COLS, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.KnB, 0, sizeof(double) * REKF_STRIP_MAX * REKF_MR_PAD, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dev.HPtB, 0, sizeof(double) * REKF_STRIP_MAX * REKF_MR_PAD, h->stream));
        std::memset(h->ctl_staging, 0, sizeof(RekfCtl));
        h->ctl_staging->n = 3;
        HIP_TRY(h, hipMemcpyAsync(h->dev.ctl, h->ctl_staging, sizeof(int) * 2, hipMemcpyHostToDevice, h->stream));
        h->pose_staging[0] = opt->init_pose[0];       // cc:9
        h->pose_staging[1] = opt->init_pose[1];
        h->pose_staging[2] = opt->init_pose[2];
        HIP_TRY(h, hipMemcpyAsync(h->dev.mu, h->pose_staging, sizeof(double) * 3, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return REKF_OK;
    }();
    if (rc != REKF_OK) {
        std::fprintf(stderr, "rekf_create: %s\n", h->hip_error.c_str());
        rekf_destroy(h);
        return rc;
    }
    *out = h;
    return REKF_OK;
}

void rekf_destroy(rekf_t *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &s : h->prof_slots) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    (void)hipFree(h->dev.ctl); (void)hipFree(h->dev.mu); (void)hipFree(h->dev.mu_out); (void)hipFree(h->dev.P);
    (void)hipFree(h->dev.HPt); (void)hipFree(h->dev.Kn); (void)hipFree(h->dev.KnB); (void)hipFree(h->dev.HPtB);
    (void)hipFree(h->dev.map_xy); (void)hipFree(h->dev.map_cov); (void)hipFree(h->dev_out12); (void)hipFree(h->dev_ell); (void)hipFree(h->dev_pred); (void)hipFree(h->dev_obs); (void)hipFree(h->dev_mu_lin);
    if (h->pose_staging) (void)hipHostFree(h->pose_staging);
    if (h->host_slots) (void)hipHostFree(h->host_slots);
    if (h->ctl_staging) (void)hipHostFree(h->ctl_staging);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int rekf_set_map(rekf_t *h, const float *xy, const double *cov, int M)
{
    if (!h || M < 0 || (M > 0 && (!xy || !cov))) return REKF_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    (void)hipFree(h->dev.map_xy); (void)hipFree(h->dev.map_cov);
    h->dev.map_xy = nullptr; h->dev.map_cov = nullptr; h->dev.M_map = 0;
    if (M == 0) return REKF_OK;
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
    if (t < h->time) return REKF_OK;                  // drop old data, cc:211-212
    if (h->opt.use_imu) return REKF_OK;               // cc:213-223: the use_imu branch is empty -- nothing happens
    h->vt[0] = vx; h->vt[1] = vy; h->vt[2] = wz;      // cc:216
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:217
    a.is_obs = 0;
    h->dev.n_known = h->n_exact ? h->n_ub : -1;
    HIP_TRY(h, hipSetDevice(h->device));
    {
        ProfScope ps(h, REKF_K_PREDICT);
        // k_front publishes the pose it commits only for a caller that reads it back between odometry messages (the
        // reference's node does: HandleOdometryMessage -> GetState, src/ros_node.cc:627-660): the kernel is one workgroup and
        // would otherwise end 1.5 us later, waiting for its PCIe writes, for nobody
        const bool want = h->pose_read;
        h->pose_read = false;
        h->pub_valid = false;
        const int seq = want ? begin_publish(h) : 0;
        rekf_launch_front(h->dev, a, h->stream);      // cc:218 Predict(dt)
        end_publish(h, seq);
    }
    h->time = t;                                      // cc:219
    return REKF_OK;
}

int rekf_handle_observation(rekf_t *h, double t, const float *xy, int K, const double *gps_pose3)
{
    if (!h || K < 0 || (K > 0 && !xy)) return REKF_ERR_INVALID;
    if (K > REKF_MAX_OBS) return REKF_ERR_TOO_MANY_OBS;
    const bool staged = K > REKF_MAX_OBS_DEV;                       // too many observations for the launch packet
    const bool blocks = 2 * K + (gps_pose3 ? 3 : 0) > 64;           // more innovation rows than one pass of k_mid takes
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:232 (dt may be negative, Q8)
    a.is_obs = 1;
    a.K = K;
    if (K > 0 && !staged) std::memcpy(a.obs, xy, sizeof(float) * 2 * (size_t)K);
    if (gps_pose3) {
        a.has_gps = 1;
        a.gps[0] = gps_pose3[0]; a.gps[1] = gps_pose3[1]; a.gps[2] = gps_pose3[2];
    }
    h->dev.n_known = h->n_exact ? h->n_ub : -1;
    HIP_TRY(h, hipSetDevice(h->device));
    ProfScope upd(h, REKF_K_UPDATE);                  // one bracket around the whole chain (per-update latency)
    if (K == 0) {                                     // cc:235-236: predict only, single-workgroup kernel
        ProfScope ps(h, REKF_K_FRONT);
        const int seq = begin_publish(h);
        rekf_launch_front(h->dev, a, h->stream);
        end_publish(h, seq);
        h->time = t;
        return REKF_OK;
    }
    h->dev.kc_ub = round_up(2 * K + (gps_pose3 ? 3 : 0), 16);
    h->dev.n_known = h->n_exact ? h->n_ub : -1;
    if (staged) {                                     // the scan does not fit the launch packet: stage it in HBM (copied before this call returns)
        HIP_TRY(h, hipMemcpyAsync(h->dev_obs, xy, sizeof(float) * 2 * (size_t)K, hipMemcpyHostToDevice, h->stream));
        a.obs_ext = h->dev_obs;
    }
    h->pub_valid = false;
    // in steady state (state full: no k_augment behind the chain) the last k_mid / k_downdate2 of the call publish the pose
    const bool fold = h->full && h->host_slots && rekf_downdate_publishes();
    int pub_seq = 0;
    { ProfScope ps(h, REKF_K_FRONT); rekf_launch_front_mb(h->dev, a, h->n_ub, h->stream); }
    h->time = t;                                      // cc:234
    const int n_ub = h->n_ub;
    const int m_ub = 2 * K + (gps_pose3 ? 3 : 0);
    h->dev.mu_lin = h->dev.mu;
    if (blocks) {
        // More than 32 observations (the reference has no limit, cc:397): matched once, then the joint update runs as
        // exact block steps of at most 32 pairs through the same two kernels (k_mid explains why that is the same
        // update).  The host cannot know how many observations matched, so it enqueues ceil(K / stride) steps; a step
        // past the last pair only carries the mean over and adds zero panels.
        const int stride = gps_pose3 ? 30 : 32;
        a.pair_stride = stride;
        h->dev.kc_ub = 64;
        rekf_launch_compact_wide(h->dev, a, h->stream);
        HIP_TRY(h, hipMemcpyAsync(h->dev_mu_lin, h->dev.mu, sizeof(double) * (size_t)h->dev.ld, hipMemcpyDeviceToDevice, h->stream));
        for (int p0 = 0; p0 < K; p0 += stride) {
            a.pair0 = p0;
            h->dev.mu_lin = h->dev_mu_lin;
            if (fold && p0 + stride >= K) pub_seq = begin_publish(h);          // the last step commits the final pose
            { ProfScope ps(h, REKF_K_MID); rekf_launch_mid(h->dev, a, n_ub, 64, h->stream); }
            std::swap(h->dev.mu, h->dev.mu_out);
            { ProfScope ps(h, REKF_K_DOWNDATE); rekf_launch_downdate(h->dev, n_ub, h->stream); }
        }
        end_publish(h, pub_seq);
        a.pair0 = -1;
        h->dev.mu_lin = h->dev.mu;
    } else {
        // the whole innovation fits one pass: gather + solve + gain as ONE launch (k_mid), which leaves the updated
        // mean in the other mean buffer
        if (fold) pub_seq = begin_publish(h);
        { ProfScope ps(h, REKF_K_MID); rekf_launch_mid(h->dev, a, n_ub, m_ub, h->stream); }
        std::swap(h->dev.mu, h->dev.mu_out);
        { ProfScope ps(h, REKF_K_DOWNDATE); rekf_launch_downdate(h->dev, n_ub, h->stream); }
        end_publish(h, pub_seq);
    }
    h->last_m_ub = m_ub;
    // the state only grows: once a readback has shown it full, k_augment can never have work again
    // (k_mid / k_compact_wide drop the extra reflectors and raise REKF_FLAG_CAPACITY)
    if (!h->full) { ProfScope ps(h, REKF_K_AUGMENT); rekf_launch_augment(h->dev, a, h->stream); }
    { ProfScope ps(h, REKF_K_EMPTY); }
    // the scan may have appended up to K reflectors; the exact n stays on the device
    int grown = n_ub + 2 * K;
    h->n_ub = grown > h->dev.n_max ? h->dev.n_max : grown;
    if (!h->full) h->n_exact = false;                 // k_augment may have appended: only the device knows by how much
    HIP_TRY(h, hipGetLastError());
    return REKF_OK;
}

int rekf_predict_state(rekf_t *h, double t, double mu3[3], double sigma3x3[9])
{
    if (!h || !mu3) return REKF_ERR_INVALID;
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:100
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->host_slots) {                               // one launch; the result arrives in pinned memory, polled by tag
        const int seq = ++h->slot_seq;
        rekf_launch_predict_pose(h->dev, a, h->dev_out12, h->host_slots_dev, seq, h->stream);
        HIP_TRY(h, hipGetLastError());
        h->pub_valid = false;                          // the slots now hold the PREDICTED pose
        int rc = wait_slots(h, 0, 12, seq);
        if (rc != REKF_OK) return rc;
        for (int q = 0; q < 3; ++q) mu3[q] = h->host_slots[q].v;
        if (sigma3x3) for (int q = 0; q < 9; ++q) sigma3x3[q] = h->host_slots[3 + q].v;
        return REKF_OK;
    }
    rekf_launch_predict_pose(h->dev, a, h->dev_out12, nullptr, 0, h->stream);
    HIP_TRY(h, hipMemcpyAsync(h->pose_staging, h->dev_out12, sizeof(double) * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::memcpy(mu3, h->pose_staging, sizeof(double) * 3);
    if (sigma3x3) std::memcpy(sigma3x3, h->pose_staging + 3, sizeof(double) * 9);
    return REKF_OK;
}

int rekf_get_time(rekf_t *h, double *t)
{
    if (!h || !t) return REKF_ERR_INVALID;
    *t = h->time;
    return REKF_OK;
}

int rekf_get_pose(rekf_t *h, double *t, double mu3[3], double sigma3x3[9])
{
    if (!h) return REKF_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->host_slots) {
        // the kernels of the last call have published (or are about to publish) the pose they committed; otherwise one small
        // launch behind whatever is enqueued does -- either way no copy engine and no stream wait
        h->pose_read = true;
        int seq = h->pub_seq;
        if (!h->pub_valid) {
            seq = ++h->slot_seq;
            rekf_launch_publish_pose(h->dev, h->host_slots_dev, seq, h->stream);
            HIP_TRY(h, hipGetLastError());
            h->pub_valid = true; h->pub_seq = seq;
        }
        int rc = wait_slots(h, 0, 13, seq);
        if (rc != REKF_OK) return rc;
        report_flags(h, h->host_slots[12].aux);
        if (t) *t = h->time;
        if (mu3) for (int q = 0; q < 3; ++q) mu3[q] = h->host_slots[q].v;
        if (sigma3x3) for (int q = 0; q < 9; ++q) sigma3x3[q] = h->host_slots[3 + q].v;
        return REKF_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(h->pose_staging, h->dev.mu, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpy2DAsync(h->pose_staging + 3, sizeof(double) * 3, h->dev.P, sizeof(double) * h->dev.ld,
                                sizeof(double) * 3, 3, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->pose_staging + 12, h->dev.ctl, sizeof(int) * 2, hipMemcpyDeviceToHost, h->stream));   // n, err
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    report_flags(h, ((const int *)(h->pose_staging + 12))[1]);
    if (t) *t = h->time;
    if (mu3) std::memcpy(mu3, h->pose_staging, sizeof(double) * 3);
    if (sigma3x3) std::memcpy(sigma3x3, h->pose_staging + 3, sizeof(double) * 9);
    return REKF_OK;
}

int rekf_get_marker_ellipses(rekf_t *h, double *out5, int cap, int *count)
{
    if (!h || !count || cap < 0 || (cap > 0 && !out5)) return REKF_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int lim = cap < h->max_landmarks ? cap : h->max_landmarks;
    rekf_launch_ellipses(h->dev, h->dev_ell, lim, h->stream);
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
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    *n = h->ctl_staging->n;
    return REKF_OK;
}

int rekf_get_state(rekf_t *h, double *t, int *n_out, double *mu, long mu_cap, double *sigma, long sigma_cap)
{
    if (!h) return REKF_ERR_INVALID;
    int rc = pull_ctl(h);
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
    return REKF_OK;
}

int rekf_set_state(rekf_t *h, double t, int n, const double *mu, const double *sigma, const double *vt3)
{
    if (!h || !mu || !sigma || n < 3 || n > h->dev.n_max || ((n - 3) & 1)) return REKF_ERR_INVALID;
    h->pub_valid = false;
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
    HIP_TRY(h, hipMemcpyAsync(h->dev.ctl, h->ctl_staging, sizeof(RekfCtl), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->time = t;
    h->n_ub = n;
    h->n_exact = true;
    h->full = n >= h->dev.n_max;
    if (vt3) { h->vt[0] = vt3[0]; h->vt[1] = vt3[1]; h->vt[2] = vt3[2]; }
    return REKF_OK;
}

int rekf_get_last_match(rekf_t *h, int *n_state, int *state_pairs, int *n_map, int *map_pairs, int *n_new,
                        int *new_ids)
{
    if (!h) return REKF_ERR_INVALID;
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
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
    int flags = 0;
    int rc = pull_n_flags(h, &flags);
    if (rc != REKF_OK) return rc;
    // the published pose can arrive a few microseconds before the last workgroups of k_downdate2 are through: Sync means done
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (flags) {
        h->pub_valid = false;                          // the slot that carried the flags is stale once they are cleared
        int zero = 0;
        HIP_TRY(h, hipMemcpy(&h->dev.ctl->err, &zero, sizeof(int), hipMemcpyHostToDevice));
    }
    return device_flags_to_code(flags);
}

int rekf_get_flags(rekf_t *h, int *flags)
{
    if (!h || !flags) return REKF_ERR_INVALID;
    return pull_n_flags(h, flags);
}

int rekf_predict_state_full(rekf_t *h, double t, double *time_out, int *n_out, double *mu, long mu_cap, double *sigma,
                            long sigma_cap)
{
    if (!h) return REKF_ERR_INVALID;
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    const int n = h->ctl_staging->n;
    const int ld = h->dev.ld;
    if (time_out) *time_out = h->time;                // `State result = state_` keeps the state's time (cc:99)
    if (n_out) *n_out = n;
    if ((mu && mu_cap < n) || (sigma && sigma_cap < (long)n * n)) return REKF_ERR_BUFFER;
    RekfFrontArgs a;
    fill_front_args(h, a, t - h->time);               // cc:100
    rekf_launch_predict_rows(h->dev, a, h->dev_pred, h->stream);
    std::vector<double> pred(4 * (size_t)ld + 16);
    HIP_TRY(h, hipMemcpyAsync(pred.data(), h->dev_pred, sizeof(double) * pred.size(), hipMemcpyDeviceToHost, h->stream));
    if (mu) HIP_TRY(h, hipMemcpyAsync(mu, h->dev.mu, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    if (sigma)
        HIP_TRY(h, hipMemcpy2DAsync(sigma, sizeof(double) * n, h->dev.P, sizeof(double) * ld, sizeof(double) * n, n,
                                    hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    // layout of pred: row0[ld] | row1[ld] | col0[ld] | col1[ld] | mu3 | corner 3x3 column-major
    const double *row0 = pred.data(), *row1 = row0 + ld, *col0 = row1 + ld, *col1 = col0 + ld, *tail = col1 + ld;
    if (mu) for (int i = 0; i < 3; ++i) mu[i] = tail[i];
    if (sigma) {
        for (int c = 3; c < n; ++c) {
            sigma[0 + (size_t)c * n] = row0[c]; sigma[1 + (size_t)c * n] = row1[c];
            sigma[c + (size_t)0 * n] = col0[c]; sigma[c + (size_t)1 * n] = col1[c];
        }
        for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) sigma[i + (size_t)j * n] = tail[3 + i + 3 * j];
    }
    return REKF_OK;
}

int rekf_profile_enable(rekf_t *h, int on)
{
    if (!h) return REKF_ERR_INVALID;
    if (!on) { int rc = prof_flush(h); if (rc != REKF_OK) return rc; }
    h->prof_on = on != 0;
    h->prof_mask = on;
    return REKF_OK;
}

int rekf_profile_read(rekf_t *h, int k, double *total_us, long *count)
{
    if (!h || k < 0 || k >= REKF_K_COUNT) return REKF_ERR_INVALID;
    int rc = prof_flush(h);
    if (rc != REKF_OK) return rc;
    if (total_us) *total_us = h->prof_total_us[k];
    if (count) *count = h->prof_count[k];
    return REKF_OK;
}

int rekf_profile_reset(rekf_t *h)
{
    if (!h) return REKF_ERR_INVALID;
    int rc = prof_flush(h);
    if (rc != REKF_OK) return rc;
    for (int k = 0; k < REKF_K_COUNT; ++k) { h->prof_total_us[k] = 0; h->prof_count[k] = 0; }
    h->prof_update_us.clear();
    return REKF_OK;
}

int rekf_profile_samples(rekf_t *h, float *out_us, long cap, long *count)
{
    if (!h || !count || cap < 0 || (cap > 0 && !out_us)) return REKF_ERR_INVALID;
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
    h->pub_valid = false;
    RekfDev dev = h->dev;
    dev.dbg = ablate;
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t a, b;
    HIP_TRY(h, hipEventCreate(&a));
    HIP_TRY(h, hipEventCreate(&b));
    const int n_ub = h->n_ub;
    auto launch = [&]() {
        if (kernel == REKF_K_DOWNDATE) rekf_launch_downdate(dev, n_ub, h->stream);
    };
    for (int i = 0; i < 3; ++i) launch();
    HIP_TRY(h, hipEventRecord(a, h->stream));
    for (int i = 0; i < reps; ++i) launch();
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
    int rc = pull_ctl(h);
    if (rc != REKF_OK) return rc;
    for (int i = 0; i < 32; ++i) out8[i] = h->ctl_staging->dbg[i];
    return REKF_OK;
}

int rekf_device_layout(rekf_t *h, int *ld, int *n_max, void **P_dev, void **mu_dev)
{
    if (!h) return REKF_ERR_INVALID;
    if (ld) *ld = h->dev.ld;
    if (n_max) *n_max = h->dev.n_max;
    if (P_dev) *P_dev = h->dev.P;
    if (mu_dev) *mu_dev = h->dev.mu;
    return REKF_OK;
}

}  // extern "C"
