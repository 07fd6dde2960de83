// ekf_dev.h -- device-side data layout shared by the EKF kernels and the C-ABI host.
//
// HBM layout of one filter handle (all FP64 unless noted), sized once at
// rekf_create for n_max = 3 + 2*max_landmarks and never reallocated:
//
//   mu, mu_out [ld]      state mean  [x, y, theta, l0x, l0y, l1x, ...], double-buffered (k_mid reads one, writes the other)
//   P    [ld x ld] x 2   covariance, COLUMN-major (Eigen::MatrixXd order,
//                        reference ekf_slam_interface.h:47), leading dimension
//                        ld = roundup(n_max, 64) so every 64x64 tile is in bounds
//                        and every column starts on a 512-byte boundary.
//                        Stored as its LOWER TRIANGLE: element (i, j) is valid iff i >= j; no kernel ever reads the memory
//                        above the diagonal (k_downdate2 happens to leave mirror images there inside its diagonal tiles,
//                        rekf_set_state uploads whatever the caller passes): the covariance cannot be anything but exactly
//                        symmetric, and the rank-m downdate moves half the bytes.  The host getters mirror on the way out.
//   HPt  [ld x 64]       HPt = (H P)^T gathered from the ROWS of P, column-major
//   Kn   [ld x 64]       Kn = -K = -W S^-1, column-major
//                        TWO buffers (round 5): the stored covariance is ONE SCAN BEHIND the filter -- a scan's rank-m downdate (and its
//                        Predict) stay PENDING as the panels below and are applied by the NEXT scan's launch, whose downdate role reads one
//                        buffer and writes the other while its mid role reads the first and corrects what it gathers (k_mid)
//   ctl                  RekfCtl below: n, error flags, the scan record
// Rows >= n of HPt and Kn are kept exactly zero, and so are their columns [m, kc_ub), so that the tile kernel
// never needs bounds checks on P.  W = P H^T and S^-1 never leave the chip (k_mid).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define REKF_MAX_OBS_DEV 64                          // observations of a scan that travel by value with the launch and update jointly
#define REKF_MAX_OBS_WIDE 256                        // most observations per scan at all (wide scans: staged in HBM, updated in exact block steps)
#define REKF_CP_LD 68                               // row stride (and rows) of a write-ahead correction panel: sub-block rows 0 .. 66
#define REKF_PANEL_COLS 64                          // columns of Kn / HPt: one block step of the update has at most 64 innovation rows

enum { REKF_FLAG_CAPACITY = 1, REKF_FLAG_SINGULAR = 2, REKF_FLAG_STARVED = 4 };

struct RekfCtl {
    int n;                    // committed state dimension (3 + 2 L)
    int err;                  // sticky REKF_FLAG_* bits
    // ---- record of the last observation (ReflectorMatchResult + H rows) ----
    int K;                    // observations in the scan
    int n_state, n_map, n_new;
    int m;                    // innovation rows = 2*(n_state+n_map) (+3 with a pose observation)
    int m_pad;                // m rounded up to 16; W/Kn columns [m, m_pad) are zero
    int state_pairs[2 * REKF_MAX_OBS_WIDE];
    int map_pairs[2 * REKF_MAX_OBS_WIDE];
    int new_ids[REKF_MAX_OBS_WIDE];
    // ---- hand-off between the multi-workgroup front kernel and k_gather / k_gain ----
    double pose_pred[5];          // x, y, theta, cos(theta), sin(theta) after Predict (mu[0..2] is committed by k_mid)
    double pose_next[2][5];       // ... of a SPECULATIVE scan (by scan parity): evaluated by workgroup 0 of the PREVIOUS scan's k_mid, at its end
    int pose_pending;
    int obs_kind[REKF_MAX_OBS_WIDE];  // per observation: 0 map match, 1 state match, 2 new
    int obs_idx[REKF_MAX_OBS_WIDE];
    // ---- the pending Predict of a scan (round 3): Predict's O(n) part -- columns 0, 1 of P against column 2 -- is not a pass over
    // memory any more.  The front kernel leaves (a, b) of G = I + a e0 e2^T + b e1 e2^T and the predicted 3 x 3 pose block here, and
    // the two kernels that read P before the scan's downdate has rewritten it apply them to what they read: k_mid to its gathers,
    // k_downdate2 to its tiles of column 0 (which then STORE predicted-and-updated values: the scan's downdate commits its Predict).
    // Two slots, by scan parity: the front kernel of scan t+1 may run beside the downdate of scan t (lazy downdate, rekf_api.hip).
    struct Pred { double ab[2]; double C9[9]; } pred[4];      // (by scan id mod 4, RekfFrontArgs::pred_ix: one launch may read the pending scan's, the current scan's, and write the next scan's)
    // ---- the 3 x 3 pose block AFTER the update (round 3): evaluated once, by k_mid's workgroup 0 (which has K's pose rows and the pose
    // columns of H P in LDS), published to the host from there -- GetPose does not wait for the downdate -- and taken over BY VALUE by
    // the downdate's tile (0, 0), so that the published block and the stored one are the same bits
    double post_C9[2][9];             // (by scan parity: the held-back downdate of scan t reads scan t's block while scan t + 1's k_mid writes its own)
    // ---- the scan's match results in the form k_mid works with (round 3): ordered compaction of the per-observation results, the rank of
    // every matched landmark among the matched ones, the slots of the sub-block.  Written by the front end -- the workgroup whose
    // observation is the last to be matched (front_count reaches the launch's front_target) -- so that k_mid starts with one load of
    // this record instead of a control-block round trip, ballots and a 64 x 64 rank count on its critical path.  Whole scans only
    // (at most 32 observations); the block steps of a wide scan build theirs in k_mid from k_compact_wide's lists.
    struct Rec {
        int cnt[8];                   // pairs, m, m_pad, state pairs, pose rows?, new reflectors (clamped to the capacity), map pairs, K
        int pair_obs[32], pair_id[32], pair_state[32];
        int urank[32];                // state pair -> rank of its landmark among the DISTINCT matched landmarks (two observations on one landmark, Q6, share it)
        int uid[32];                  // the distinct matched landmarks in ascending order (uid[urank] = landmark id)
        int nu, pad_[3];              // how many: the sub-block k_mid gathers has 3 + 2 nu rows, in ascending global order
        int newid[64];                // observation indices of the new reflectors
    } rec[2];                         // (by scan parity: the SPECULATIVE front end of scan t + 1 runs inside scan t's launch, beside the mid role that reads scan t's)
    // SPECULATIVE MATCH (round 5): scan t + 1's ReflectorMatch runs inside scan t's launch, against the mean and pose scan t STARTS from (the
    // update it would have to wait for moves them by millimetres).  Per observation it leaves the result, the distance to the nearest
    // reflector and to the runner-up; scan t + 1's k_mid accepts the record when every decision is PROVABLY the one the exact match would
    // take -- |d1 - gate| and d2 - d1 against a bound of how far the update moved the pose and the reflectors -- and re-matches the
    // observations that do not pass
    struct Spec { int kind[32], idx[32]; double d1[32], d2[32]; double dm1[32]; double pose[3]; int n; unsigned scan; } spec[2];     // (dm1: with a pre-loaded map, the distance to the nearest MAP point of an observation that did not match the map)
    unsigned long long dmmax[4];      // bits of max |mu_new - mu_old| over the landmark rows of a scan's update (by scan id mod 4: a launch writes its own, reads the previous scan's, zeroes the next one's)
    unsigned front_count;             // observations matched so far, over the life of the handle (never reset)
    // ---- a scan's landmark augmentation deferred into the NEXT scan's k_mid (round 4): while the state can grow, every scan used to
    // be followed by a k_augment launch that found nothing to do (2.4 us of kernel boundary per scan).  k_mid's workgroup 0 leaves what
    // the augmentation needs here -- the state dimension it appends to, the number of new reflectors, their observations -- by scan
    // parity (the record of scan t is read all through k_mid of scan t + 1, while that kernel's workgroup 0 writes scan t + 1's), and the
    // next k_mid's workgroup 0 does the appending first thing; the other workgroups wait for aug_done only when there IS something
    struct AugRec { int n_before, n2; float obs[2 * REKF_MAX_OBS_DEV]; } augrec[2];
    unsigned aug_done;                // scan id of the last k_mid whose workgroup 0 has appended its predecessor's new reflectors
    int aug_arrive;                   // k_dd_front, RekfDev::aug_tail: downdate workgroups of the launch that are through (the last one appends; back to 0 by it)
    unsigned rec_seq;                 // scan id of the last scan whose match record (rec) the front role INSIDE k_mid's grid has completed
    // WRITE-AHEAD CORRECTION (round 5): at the end of a scan's k_mid the workgroups that own rows of the scan's sub-block R write the scan's
    // rank-m correction of P(R, R) -- sum_k HPt(lo, k) Kn(hi, k), the downdate's own arithmetic -- to RekfDev::cp (by scan parity).  The
    // next scan's k_mid, which meets the stored P one scan behind, adds it to what it gathers when its own R is the same set of landmarks
    // (usually: the 32 nearest reflectors change every 5-10 scans); else it computes the correction itself
    int cp_uid[2][32];                // the landmarks the panel of that parity covers, ascending
    int cp_nu[2];
    unsigned cp_scan[2];              // ... and the scan (RekfFrontArgs::scan_id) that wrote it
    unsigned dd_queue[2];             // the downdate role INSIDE k_mid's grid takes its tiles from a queue: next work item, by launch parity (the
                                      // mid role's workgroup 0 zeroes the other parity's counter for the next launch)
    // MATCH GRID (round 6): a hash grid over the landmarks' float32 means -- 1 m cells, buckets of {count, 7 x (landmark id, its CURRENT float32
    // mean)}, RekfDev::grid_bucket (layout: rekf_grid_* below); RekfDev::grid_p0 = where each landmark stood when it was binned, and its entry.
    // The entries' means are kept current by whoever updates the mean (k_mid phase F, double-buffered by RekfDev::grid_par like the mean itself).  ReflectorMatch's state branch accepts a landmark only inside 0.6 m
    // (cc:446), and such a landmark -- as long as it has drifted less than REKF_GRID_DRIFT from its binning position -- sits in one of the
    // 3 x 3 cells around the observation's global point: k_mid can match a host-predicted scan ITSELF, exactly, from ~10 candidates per
    // observation instead of all L (no front-end launch).  grid_state: 1 = built and every landmark within the drift bound; 0 = not
    // (k_mid then matches by the full sweep, slowly, and the host rebuilds: k_grid_build).
    int grid_state;
    int grid_overflow;            // a bucket ran over: the table is too small for this world (the host stops using the grid)
    long long dbg[32];            // scratch for in-kernel timing experiments (REKF_DEBUG_TIMING builds)
};
#define REKF_GRID_SLOTS 7
#define REKF_GRID_DRIFT 0.3f

// By-value kernel argument of the front kernel: one scan's worth of host input.
struct RekfFrontArgs {
    double dt;                // t - state time (host tracks time)
    double vt[3];             // vt_ used for this predict
    double lin_cov, ang_cov, obs_cov;
    double gps[3];
    int model;                // 0 DIFF, 1 OMNI
    int is_obs;               // 0: odometry path (predict only, scan record untouched)
    int K;
    int has_gps;
    float obs[2 * REKF_MAX_OBS_DEV];
    // wide scans (K > REKF_MAX_OBS_DEV): the observations live in HBM, and k_mid takes the matched pairs [pair0, pair0 + stride)
    // of the record k_compact_wide left in the control block -- one exact block step of the joint update per launch
    const float *obs_ext;     // K x 2 floats on the device, or null (obs[] above holds them)
    int pair0;                // first pair of this block step, or -1: the whole scan (k_mid compacts the match results itself)
    int pair_stride;          // pairs per block step (32, or 30 with a pose observation: its 3 rows ride on the last step)
    // Predict evaluated by the HOST (rekf_api.hip keeps a mirror of the pose mean and the 3 x 3 pose block whenever the caller has
    // read the pose back; HandleOdometryMessage then costs no launch at all).  host_pred = 1: the device evaluates no motion model --
    // pre_pose is the predicted pose, pre_ab the composite G = I + a e0 e2^T + b e1 e2^T of every predict since the device last
    // saw P (consecutive predicts compose exactly: e2^T (a, b, 0)^T = 0), pre_C9 the pose block after them.
    int pred_slot;            // which RekfCtl::pred slot this scan's Predict goes through (front kernel writes, k_mid reads)
    unsigned front_target;    // RekfCtl::front_count once all K observations of THIS scan are matched: the workgroup that reaches it compacts
    int compact_in_front;     // 1: whole scan (K <= 32, one pass of k_mid): the front end leaves RekfCtl::rec; 0: wide scan (k_compact_wide)
    int compact_in_mid;       // whole scan whose front end is a LAUNCH in front of k_mid (k_front_mb, k_dd_front): its workgroups leave their results in
                              // RekfCtl::obs_kind / obs_idx and end -- no count, no compacting workgroup; k_mid, behind the kernel boundary, compacts them
                              // for itself (every workgroup: 2 loads + one wave's ballots).  The count-and-elect scheme put a 255 -> 1 fan-in and a
                              // second pass through memory (~4 us) on the path of every scan that is not speculated
    int aug_pending;          // front role inside k_dd_front: the previous scan's k_augment has not run yet -- the state the match sees has
                              // n + 2 n_new rows (the new reflectors' means are there: k_mid writes them)
    int front_in_mid;         // k_mid: the first front_in_mid workgroups of its grid are this scan's front end (a host-predicted scan behind a
                              // pose read-back: no launch of its own for the match); the others wait for RekfCtl::rec_seq
    int dd_in_mid;            // k_mid: the workgroups from dd_first on are the PREVIOUS scan's downdate (four of their eight waves; tiles from
    int dd_first;             // RekfCtl::dd_queue[dd_par]); 0: none.  The front role's workgroups join them when their observation is matched
    int dd_par;
    int n_mid;                // k_mid: workgroups of the mid role (behind the front_in_mid front workgroups)
    int corr;                 // k_mid: the stored P is one scan behind -- what is gathered takes the pending downdate (dp's panels, kc_ub columns) ...
    int corr_pred;            // ... and, first, the pending Predict (RekfCtl::pred[corr_pred], -1: none); the pose block is RekfCtl::post_C9[corr_post]
    int corr_post;
    int corr_pred_ix;         // ... (its RekfCtl::pred index)
    int pred_ix;              // this scan's RekfCtl::pred / dmmax index (scan id mod 4)
    unsigned corr_scan;       // ... and that scan's id (its write-ahead correction, RekfCtl::cp_scan, must carry it)
    int cp_write;             // k_mid: leave this scan's write-ahead correction (whole scans)
    int spec;                 // k_mid: the scan's match record is SPECULATIVE (RekfCtl::spec[pred_slot]): prove it or re-match; Predict is evaluated here
    int spec_front;           // k_mid: that many workgroups behind the mid role are the NEXT scan's speculative front end (its launch packet: An)
    double prev_dt, prev_vt[3];   // front role as speculation for scan t + 1 inside scan t's launch: scan t's own motion comes first
    int aug_in_mid;           // k_mid: the previous scan's augmentation has not run: workgroup 0 appends its rows first (RekfCtl::augrec), n = n_before + 2 n2
    unsigned scan_id;         // running number of the scan (RekfCtl::aug_done)
    int apply_pred;           // k_mid: apply the pending Predict to the gathered P (whole scan or FIRST block step of a wide scan)
    int host_pred;
    int grid_match;           // k_mid (a host-predicted whole scan): no front end has run -- every workgroup matches the scan itself through the match grid (RekfCtl::grid_state)
    double pre_pose[5];       // x, y, theta (wrapped, cc:181/:205), cos(theta), sin(theta) of the WRAPPED heading as the reference takes them (cc:252-253)
    double pre_ab[2];
    double pre_C9[9];         // column-major 3 x 3
};
__host__ __device__ static inline float rekf_obs(const RekfFrontArgs &A, int i) { return A.obs_ext ? A.obs_ext[i] : A.obs[i]; }

// a value the kernels hand to the host in place: one 16-byte system-scope store into pinned host memory, polled by its tag.
// Slots 0..2 = pose mean, 3..11 = the 3 x 3 pose block (column-major), 12 = {n, aux = sticky flags}.
struct RekfHostSlot { double v; int seq; int aux; };

struct RekfDev {
    RekfCtl *ctl;
    double *mu;         // the current mean
    double *mu_out;     // k_mid writes the updated mean here; the host swaps mu / mu_out behind that launch
    const double *mu_lin; // the mean the scan is linearised at (= mu, except in the later block steps of a wide scan)
    double *P;          // the stored covariance (every reader's source)
    double *P_out;      // k_downdate2 / the downdate role: where the downdated tiles go (= P: in place; the other buffer: one launch per scan)
    double *cp;         // write-ahead corrections, 2 x REKF_CP_LD x REKF_CP_LD (by scan parity; RekfCtl::cp_*)
    double *HPt;        // (H P)^T, column-major, REKF_PANEL_COLS columns
    double *Kn;         // -K, column-major, REKF_PANEL_COLS columns
    float *map_xy;      // M_map x 2
    double *map_cov;    // M_map x 4 row-major
    int M_map;
    int ld;
    int n_max;
    int dbg;            // ablation bits for rekf_debug_time_kernel; 0 in normal operation
    int n_known;        // the exact state dimension when the host knows it (state full, or nothing enqueued since a read-back), else -1:
                        // spares the kernels a dependent read of ctl->n at their start
    RekfHostSlot *pub;  // non-null (only in the launch of the LAST k_downdate2 / k_dd_front of a call): its first workgroup publishes, at its
                        // START, the pose mean, the 3 x 3 pose block after the update (RekfCtl::post_C9), n and the flags ...
    int pub_seq;        //   ... under this tag, so that GetPose / Sync after the call need no kernel of their own
    int pub_aug;        //   ... and a k_augment follows: the n to publish is n + 2 ctl->n_new (neither pose nor pose block change there)
    int dd_lo, dd_x;    // k_downdate2, class B: tiles per workgroup -- dd_lo each, the first dd_x workgroups one more (set by rekf_launch_downdate)
    int dd_sub;         // k_downdate2: class B holds the tiles with I >= J + dd_sub (2: a class-A workgroup also takes the tile below its
                        // diagonal tile; 1: it does not; 0: the host does not know n exactly -- the kernel derives the schedule itself)
    int dd_grid;        // k_dd_front: workgroups [0, dd_grid) are the downdate, the rest the next scan's front end (0: k_downdate2, the whole grid)
    int aug_tail;       // k_dd_front: this view's scan may have met new reflectors (RekfCtl::augrec[aug_tail - 1]): the LAST downdate workgroup to
                        // finish appends their covariance rows -- k_augment without a launch of its own; 0: no
    int pred_ix;        // the RekfCtl::pred index of the scan this view belongs to (scan id mod 4)
    int post_slot;      // k_downdate2 / k_mid: the RekfCtl::post_C9 slot of the scan this view belongs to (k_mid writes it, the scan's downdate stores it)
    int pred_slot;      // k_downdate2: >= 0: the scan's pending Predict (RekfCtl::pred[pred_slot]) is applied to the tiles of column 0 as they are
                        // read (and so committed by this launch); -1: nothing pending (later block steps of a wide scan, timing hook)
    int kc_ub;          // host bound of m_pad for the current scan, rounded up to 16 (0: unknown); columns [m, kc_ub) of HPt / Kn are zero
    int aug_write;      // k_mid: leave the scan's augmentation record (RekfCtl::augrec) whatever the kernel's MODE (a growing filter in the one-launch form)
    RekfHostSlot *early;  // k_mid (whole scans on a filter that can still grow): workgroup 0 publishes the n the state has behind this scan -- n + 2 (new
    int early_seq;        // reflectors) -- under this tag AS SOON AS the scan's match record is final (rekf_api.hip, struct rekf: EARLY n)
    int pad_;
    int *grid_bucket;     // the match grid (RekfCtl::grid_state), one allocation: cnt[GH] | id[7 GH] | xy[2 parities][7 GH][2] (GH = grid_mask + 1), or null
    float *grid_p0;       // ... and per landmark: its binning position p0[Lcap][2] | its entry slot[Lcap] (bucket * 7 + slot, -1: none); Lcap = (n_max - 3) / 2 + 8
    RekfHostSlot *grid_note;  // where a kernel that invalidates the grid tells the host (tag = the scan's id, v = 1 drift / 2 overflow)
    int grid_mask;
    float grid_drift;     // how far a landmark may stand from its binning position before the grid is invalid (REKF_GRID_DRIFT; tests lower it)
    int grid_par;         // which half of xy holds the means a launch STARTS from (it writes the updated ones into the other half; the host flips it with mu / mu_out)
    int pad3_;
    double map_lip;       // pre-loaded map: sqrt of the largest eigenvalue over its covariances -- sqrt(e^T S e) moves by at most map_lip |de| when the
                          // observation's global point moves by de (the speculative match's margin proof, k_mid); < 0: some covariance is not
                          // symmetric positive semi-definite, no bound (scans are then not speculated for)
};

// ----------------------------------------------------------------------------
// Predict's scalar part (reference reflector_ekf_slam.cc:154-206), shared by the kernels and by the host's pose mirror
// (rekf_api.hip): the motion increment d, the entries a, b of G = I + a e0 e2^T + b e1 e2^T and V = Gu Qu Gu^T.
// No FMA contraction: the same doubles as a plain x86-64 build of the reference.
// ----------------------------------------------------------------------------
// one libm call for both (device: ocml, host: glibc -- whose sincos returns exactly what sin and cos return)
__host__ __device__ static inline void rekf_sincos(double x, double *s, double *c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(x, s, c);
#else
    ::sincos(x, s, c);
#endif
}

struct Motion {
    double d[3];
    double a, b;
    double V[9];
};

__host__ __device__ static inline void motion_terms_of(int model, double dt, double vx, double vy, double w, double lin_cov, double ang_cov, double theta, Motion &mo);
__host__ __device__ static inline void motion_terms(const RekfFrontArgs &A, double theta, Motion &mo)
{
    motion_terms_of(A.model, A.dt, A.vt[0], A.vt[1], A.vt[2], A.lin_cov, A.ang_cov, theta, mo);
}
__host__ __device__ static inline void motion_terms_of(int model, double dt, double vx, double vy, double w, double lin_cov, double ang_cov, double theta, Motion &mo)
{
#pragma clang fp contract(off)
    double Gu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double Qu[3];
    int q;
    if (model == 0) {                                     // DIFF  cc:156-183
        const double delta_theta = w * dt;
        const double half = theta + delta_theta / 2;
        double sh, ch;
        rekf_sincos(half, &sh, &ch);
        mo.d[0] = vx * dt * ch;
        mo.d[1] = vx * dt * sh;
        mo.d[2] = delta_theta;
        mo.a = -vx * dt * sh;
        mo.b = vx * dt * ch;
        q = 2;
        Gu[0] = dt * ch; Gu[1] = -vx * dt * dt * sh / 2;
        Gu[3] = dt * sh; Gu[4] = vx * dt * dt * ch / 2;
        Gu[6] = 0;       Gu[7] = dt;
        Qu[0] = lin_cov; Qu[1] = ang_cov; Qu[2] = 0;
    } else {                                              // OMNI  cc:184-205
        const double delta_theta = w * dt;
        double st, ct;
        rekf_sincos(theta, &st, &ct);
        mo.d[0] = vx * dt * ct - vy * dt * st;
        mo.d[1] = vx * dt * st + vy * dt * ct;
        mo.d[2] = delta_theta;
        mo.a = -vx * dt * st - vy * dt * ct;
        mo.b = vx * dt * ct - vy * dt * st;
        q = 3;
        Gu[0] = dt * ct; Gu[1] = -dt * st; Gu[2] = 0.;
        Gu[3] = dt * st; Gu[4] = dt * ct;  Gu[5] = 0.;
        Gu[6] = 0.;      Gu[7] = 0.;       Gu[8] = dt;
        Qu[0] = lin_cov; Qu[1] = lin_cov; Qu[2] = ang_cov;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < q; ++k)
                s += Gu[i * 3 + k] * Qu[k] * Gu[j * 3 + k];
            mo.V[i * 3 + j] = s;
        }
}

// 3x3 pose block of G P G^T + Gu Qu Gu^T (row ops, then column ops, then + V),
// in place on a column-major 3x3 with leading dimension ld.
__host__ __device__ static inline void corner_predict(double *P, int ld, const Motion &mo)
{
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
        const double p2 = P[2 + (size_t)c * ld];
        P[0 + (size_t)c * ld] = P[0 + (size_t)c * ld] + mo.a * p2;
        P[1 + (size_t)c * ld] = P[1 + (size_t)c * ld] + mo.b * p2;
    }
    for (int r = 0; r < 3; ++r) {
        const double p2 = P[r + (size_t)2 * ld];
        P[r + (size_t)0 * ld] = P[r + (size_t)0 * ld] + mo.a * p2;
        P[r + (size_t)1 * ld] = P[r + (size_t)1 * ld] + mo.b * p2;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            P[i + (size_t)j * ld] += mo.V[i * 3 + j];
    // the stored covariance is EXACTLY symmetric (k_downdate2 computes the lower triangle and mirrors it): the upper
    // elements take the lower ones' bits (the reference's two differ in the last place at most)
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j) P[i + (size_t)j * ld] = P[j + (size_t)i * ld];
}

// sigma(i, j) of the lower-triangle storage
__host__ __device__ static inline double rekf_plower(const double *P, int ld, int i, int j)
{
    return (i >= j) ? P[(size_t)i + (size_t)j * (size_t)ld] : P[(size_t)j + (size_t)i * (size_t)ld];
}

// bucket of the 1 m cell (cx, cy) of the match grid
__host__ __device__ static inline int rekf_grid_hash(int cx, int cy, int mask)
{
    return (int)(((unsigned)cx * 73856093u) ^ ((unsigned)cy * 19349663u)) & mask;
}

__host__ __device__ static inline int *rekf_grid_cnt(const RekfDev &d) { return d.grid_bucket; }
__host__ __device__ static inline int *rekf_grid_id(const RekfDev &d) { return d.grid_bucket + (d.grid_mask + 1); }
__host__ __device__ static inline float *rekf_grid_xy(const RekfDev &d, int par) { return (float *)(d.grid_bucket + 8 * (size_t)(d.grid_mask + 1)) + (size_t)(par & 1) * 14 * (size_t)(d.grid_mask + 1); }
__host__ __device__ static inline int *rekf_grid_slot(const RekfDev &d) { return (int *)(d.grid_p0 + 2 * (size_t)((d.n_max - 3) / 2 + 8)); }
#define REKF_GRID_INTS_PER_BUCKET 36          // 1 + 7 + 2 * 14

// launch wrappers (ekf_kernels.hip)
void rekf_launch_apply_predict(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s);
void rekf_launch_front_mb(const RekfDev &d, const RekfFrontArgs &a, int n_ub, hipStream_t s);
void rekf_launch_compact_wide(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s);
void rekf_launch_mid(const RekfDev &d, RekfFrontArgs &a, int n_ub, int m_ub, bool mode_grow, hipStream_t s);   // mode_grow: the filter can still grow, or the previous scan's augmentation rides in this launch
void rekf_launch_downdate(const RekfDev &d, int n_ub, hipStream_t s);
void rekf_launch_dd_front(const RekfDev &d, int n_ub, const RekfDev &dn, const RekfFrontArgs &an, hipStream_t s);
bool rekf_scan_launch_fits(int n_ub, int K_front);   // the one-launch form leaves the downdate role enough CUs (and its header fields hold the grid)
int rekf_launch_scan(const RekfDev &dd, const RekfDev &d, RekfFrontArgs &a, int n_ub, int m_ub, int front_wgs, const RekfFrontArgs *an, hipStream_t s);   // ONE launch per scan: [front end |] mid role (corrects what it gathers by dd's pending panels) | dd's downdate from dd.P into dd.P_out
void rekf_launch_augment(const RekfDev &d, const RekfFrontArgs &a, hipStream_t s);
void rekf_launch_ellipses(const RekfDev &d, double *out5, int cap, hipStream_t s);
void rekf_launch_grid_build(const RekfDev &d, int n, int note_tag, hipStream_t s);
void rekf_launch_publish_pose(const RekfDev &d, RekfHostSlot *hout, int seq, hipStream_t s);
void rekf_launch_predict_rows(const RekfDev &d, const RekfFrontArgs &a, double *out, hipStream_t s);
