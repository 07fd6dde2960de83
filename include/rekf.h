/*
 * rekf.h -- C ABI of the MI355X-native reflector EKF-SLAM core (librekf.so).
 *
 * This is the drop-in boundary underneath the reference's C++ interface
 * ekf::ReflectorEKFSLAMInterface (reference include/reflector_ekf_slam/
 * ekf_slam_interface.h:50-67).  The reference has no FFI of its own; each entry
 * point below names the reference method it replaces.  A header-only C++ adapter
 * with the reference's class and method names lives in
 * include/reflector_ekf_slam_amd/ekf_slam_adapter.hpp; INTEGRATION.md shows the
 * five-line change in the reference's ros_node.cc.
 *
 * Conventions
 *  - opaque handle, one HIP stream per handle, NOT thread-safe: the caller
 *    serialises calls (the reference's caller is the single-threaded
 *    ros::spin(), reference src/ros_node.cc:68).
 *  - plain pointers and sizes only; all buffers are caller-owned HOST memory and
 *    are borrowed for the duration of the call.
 *  - every function returns 0 (REKF_OK) or a negative REKF_ERR_* code; nothing
 *    calls exit() or throws across this boundary (the reference LOG(ERROR)+exit(-1)s,
 *    reflector_ekf_slam.cc:373-378; the C++ adapter may restore that).
 *  - rekf_handle_observation only ENQUEUEs device work and returns.  rekf_handle_odometry launches NOTHING: Predict
 *    (reflector_ekf_slam.cc:154-206) is O(1) on the pose and touches two rows of P, so the library keeps a host mirror
 *    of the pose mean and the 3 x 3 pose block (current after any pose read-back), advances it with the host's libm and
 *    hands the composite of all predicts to the next scan's first kernel.  If the last scan's pose has not been read
 *    back yet, the first odometry message behind it waits for that pose to be published (never, in the reference's
 *    call pattern: its node reads the pose after every scan, src/ros_node.cc:514-515).  rekf_get_pose and
 *    rekf_predict_state between scans are answered from the mirror without touching the device.
 *  - PENDING WORK (rounds 3-6).  Of the work a scan enqueues, the covariance downdate (reflector_ekf_slam.cc:308) and the landmark
 *    augmentation (:311-364) are held back and go out with the NEXT call: with the next scan they run beside that scan's update (the
 *    stored covariance is kept one scan behind the filter; the update corrects what it reads of it -- same bits).  A caller that hands
 *    scan after scan over WITHOUT reading anything back in between also has its NEWEST SCAN held on the host until the next call:
 *    that scan's launch then carries the next scan's ReflectorMatch with it, speculatively, one launch early (proved or repaired by
 *    the next scan's update: identical association lists by construction).  Every entry point that reads the state (the getters,
 *    rekf_handle_odometry, rekf_sync, rekf_predict_state*, rekf_reserve, rekf_device_layout, ...) sends whatever is held first: no
 *    caller can observe a state without it, and a caller that reads the pose after every scan (the reference's node) is never held.
 *    ERRORS of a held scan (ABI 7): the call that sends it reports them, and that call's OWN scan (if it brought one) has then not been
 *    touched -- repeat the call.  If the failure came before anything of the handle had moved (every HIP call that can fail comes first)
 *    the held scan is still held and the repeated call sends both, each exactly once; after a refused kernel launch the held scan counts
 *    as applied, like any scan whose call fails there (the handle resynchronises with the device).
 *    A filter that can still GROW (n < 3 + 2 max_landmarks) waits, at the start of each scan's call, until the PREVIOUS scan's launch
 *    has published the n that scan leaves (a few microseconds into that launch: its match is final long before its update is) -- the
 *    host of a growing filter runs at most one launch ahead of the device, and in exchange knows n exactly when it plans a launch.
 *    Callers that read device memory through their own HIP calls must call rekf_device_layout or rekf_sync each time.
 *    REKF_SPEC=0 / REKF_SCAN_LAUNCH=0 in the environment (read at rekf_create) switch the speculation / the one-launch form off: the
 *    twins the bit-identity tests compare against (same results); REKF_EXCLUSIVE=1 = rekf_set_exclusive.
 *  - the covariance lives in HBM for the life of the handle, column-major like
 *    Eigen::MatrixXd (ekf_slam_interface.h:47) with a fixed leading dimension, as its LOWER TRIANGLE (element (i, j) is
 *    valid iff i >= j; nothing reads the memory above the diagonal); the getters mirror it into the caller's n x n buffer.
 */
#ifndef REKF_H_
#define REKF_H_

#ifdef __cplusplus
extern "C" {
#endif

#define REKF_ABI_VERSION 7

/* Most observations one scan may carry (K).  The reference has no limit (reflector_ekf_slam.cc:397 loops over
 * obs.cloud_.size()); this one is a buffer size, equal to what the detectors of rdet.h can emit (RDET_MAX_CENTERS).
 * Up to 32 observations (64 innovation rows: a 64 x 64 innovation covariance is what one pass of k_mid inverts) update jointly in
 * one pass; wider scans are matched once and updated in exact block steps of 32 matched pairs (same posterior: see k_mid in
 * csrc/ekf_kernels.hip).  Scans of up to 64 observations travel by value in the launch packet, wider ones through a staging buffer. */
#define REKF_MAX_OBS 256

enum {
    REKF_OK = 0,
    REKF_ERR_INVALID = -1,        /* bad argument / null handle */
    REKF_ERR_HIP = -2,            /* a HIP runtime call failed; see rekf_last_hip_error */
    REKF_ERR_TOO_MANY_OBS = -3,   /* K > REKF_MAX_OBS */
    REKF_ERR_CAPACITY = -4,       /* state grew past max_landmarks; extra reflectors were dropped */
    REKF_ERR_SINGULAR = -5,       /* innovation covariance had a non-positive pivot */
    REKF_ERR_BUFFER = -6,         /* caller buffer too small */
    REKF_ERR_UNSUPPORTED = -7     /* max_landmarks too large: P is addressed with 32-bit byte offsets (about 11.5 k landmarks) */
};

/* Sticky device-side condition bits (rekf_get_flags): the state kept growing past max_landmarks and the extra
 * reflectors of a scan were dropped (the reference grows without bound, reflector_ekf_slam.cc:311-364); the
 * innovation covariance S of some scan had a non-positive pivot (the update was applied as computed); a hand-over INSIDE a launch
 * (workgroups of k_mid waiting for the scan's front end or the previous scan's augmentation, which run as other workgroups of the same
 * launch) gave up waiting -- the state is no longer meaningful; rekf_sync returns REKF_ERR_HIP.  Such hand-overs exist ONLY on a handle
 * the caller has declared EXCLUSIVE (rekf_set_exclusive, below): by default no launch of this library contains a workgroup that waits
 * for another one. */
enum { REKF_FLAGBIT_CAPACITY = 1, REKF_FLAGBIT_SINGULAR = 2, REKF_FLAGBIT_STARVED = 4 };

enum { REKF_ODOM_DIFF = 0, REKF_ODOM_OMNI = 1 };   /* sensor::OdometryModel, sensor_data.h:56-60 */

/* ekf::EKFOptions (ekf_slam_interface.h:28-41).  The three *_cov fields are
 * variances: the reference's caller squares the launch-file sigmas
 * (src/ros_node.cc:207-238).  map_path is handled by the host side
 * (rekf_set_map).  use_imu != 0 is accepted and behaves like the reference: odometry
 * messages are then ignored (reflector_ekf_slam.cc:213-223; the IMU branch is empty, and the
 * reference's caller forces use_imu = false, src/ros_node.cc:186). */
typedef struct rekf_options {
    int odom_model;
    int use_imu;
    double init_time;
    double init_pose[3];
    double linear_velocity_cov;
    double angular_velocity_cov;
    double observation_cov;
} rekf_options;

typedef struct rekf rekf_t;

/* ReflectorEKFSLAM::ReflectorEKFSLAM(options)  (reflector_ekf_slam.cc:6-37).
 * Allocates mu / P / scratch for up to max_landmarks reflectors on HIP device
 * `device`.  P is n_max x n_max FP64 with n_max = 3 + 2*max_landmarks. */
int rekf_create(const rekf_options *opt, int max_landmarks, int device, rekf_t **out);
void rekf_destroy(rekf_t *h);

/* The reference resizes mu / sigma on every augment and never runs out of room (reflector_ekf_slam.cc:316-363).  Here the
 * buffers are sized once; rekf_reserve re-lays the state out for a larger max_landmarks (new buffers with the larger
 * leading dimension, one strided device-to-device copy of the n x n covariance, old buffers freed; no-op when the
 * capacity already suffices).  Synchronises.  With rekf_set_auto_grow(h, 1) HandleObservationMessage calls it itself
 * (capacity doubling) whenever a scan COULD overflow the capacity -- it then first waits for the exact n -- so that no
 * reflector is ever dropped and REKF_FLAGBIT_CAPACITY never fires.  At THIS level it is off until switched on (a handle fresh from
 * rekf_create has a fixed capacity and the sticky flag); every wrapper shipped with the library -- the C++ EkfSlam / adapter classes,
 * the Python ReflectorEKFSLAM, the replay node -- switches it on by default, because that is the reference's behaviour (max_landmarks
 * is then the INITIAL capacity), and takes auto_grow = false for a fixed one. */
int rekf_reserve(rekf_t *h, int new_max_landmarks);
int rekf_set_auto_grow(rekf_t *h, int on);
/* EXCLUSIVE (off by default; REKF_EXCLUSIVE=1 in the environment at rekf_create switches it on): the caller promises that this handle
 * has the GPU to itself -- no other process, no other stream of this process keeps its CUs busy.  The library then lets a scan's front
 * end (Predict's pose, ReflectorMatch) run as the first workgroups of the scan's own launch, with the update's workgroups waiting for it
 * INSIDE the launch (one launch per scan, the match off the launch boundary).  Safe only under that promise: with other work on the
 * GPU the waiting workgroups can hold the CUs the working ones need (REKF_FLAGBIT_STARVED).  Even on an exclusive handle the
 * hand-overs are used only while it is the process's only live handle.  Results are the same bits either way. */
int rekf_set_exclusive(rekf_t *h, int on);
int rekf_get_capacity(rekf_t *h, int *max_landmarks);

/* map_ as LoadMapFromTxtFile leaves it (reflector_ekf_slam.cc:80-94): M points
 * (float32 xy) with M row-major 2x2 covariances.  M = 0 clears the map. */
int rekf_set_map(rekf_t *h, const float *xy, const double *cov, int M);

/* HandleOdometryMessage (reflector_ekf_slam.cc:208-223): drops t < state time,
 * stores (vx, vy, wz), predicts by t - state time -- on the host's pose mirror, no kernel launch (see Conventions). */
int rekf_handle_odometry(rekf_t *h, double t, double vx, double vy, double wz);

/* HandleObservationMessage (reflector_ekf_slam.cc:229-368): predict to t,
 * ReflectorMatch, EKF update, landmark augmentation.  xy = K robot-frame points
 * (sensor::Observation::cloud_).  gps_pose3 = nullable (x, y, yaw): the pose
 * observation of the USE_GPS build (reflector_ekf_slam_gps.cc:305-340).
 * Asynchronous: xy is copied before the call returns (an empty scan, K = 0, is a Predict: handled like an odometry message). */
int rekf_handle_observation(rekf_t *h, double t, const float *xy, int K,
                            const double *gps_pose3);

/* PredictState (reflector_ekf_slam.cc:97-152), pose block only (the only part the reference's caller
 * reads, src/ros_node.cc:455-470): non-mutating; evaluated on a copy of the host's pose mirror (no device access once
 * the pose of the last scan has been read back). */
int rekf_predict_state(rekf_t *h, double t, double mu3[3], double sigma3x3[9]);

/* PredictState as the interface returns it (ekf_slam_interface.h:59): the FULL predicted State --
 * `State result = state_` with result.sigma = G sigma G^T + Gu Qu Gu^T over all n (rows/columns 0, 1 and the
 * pose block change; computed on the device with Predict's own arithmetic) and result.mu[0..2] advanced;
 * *time_out = the state's time, which the reference leaves unchanged in the copy (:99).  Buffers as in
 * rekf_get_state.  Non-mutating.  Synchronises; an n x n D2H copy like GetState. */
int rekf_predict_state_full(rekf_t *h, double t, double *time_out, int *n, double *mu, long mu_cap,
                            double *sigma, long sigma_cap);

/* GetLatestTime (reflector_ekf_slam.h:33-36).  No device access. */
int rekf_get_time(rekf_t *h, double *t);

/* Pose-only fast path for the GetState() call the reference's caller makes
 * after every callback (src/ros_node.cc:515,592,638): the last kernel of a scan's chain stores pose, pose block, n and the
 * flags as tagged slots into pinned host memory and this call polls them (no copy engine, no stream wait); after an
 * odometry message it is answered from the host mirror.  sigma3x3 is column-major. */
int rekf_get_pose(rekf_t *h, double *t, double mu3[3], double sigma3x3[9]);

/* Current state dimension n = 3 + 2*landmarks.  Synchronises. */
int rekf_get_n(rekf_t *h, int *n);

/* GetState (reflector_ekf_slam.h:37-40): full copy.  mu (n doubles) and sigma
 * (n*n doubles, column-major, leading dimension n) may each be NULL.
 * mu_cap / sigma_cap are the buffer capacities in doubles.  Synchronises. */
int rekf_get_state(rekf_t *h, double *t, int *n, double *mu, long mu_cap,
                   double *sigma, long sigma_cap);

/* Landmark covariance ellipses for the caller's visualisation (Node::ReflectorToRosMarkers,
 * src/ros_node.cc:736-765): per landmark i the 2x2 block sigma(3+2i.., 3+2i..) is eigen-decomposed ON THE
 * DEVICE and only 5 doubles come back -- {mx, my, angle, x_len, y_len} with angle = atan2 of the first
 * pseudo-eigenvector (:763) and x_len/y_len = 2 sqrt(5.991 * eigenvalue) (:764-765) -- instead of the
 * n x n GetState() copy the reference makes for this (33 MB at 1024 landmarks vs 40 KB).
 * out5 holds cap landmarks (5*cap doubles); *count = landmarks written.  Synchronises. */
int rekf_get_marker_ellipses(rekf_t *h, double *out5, int cap, int *count);

/* Restore a full state (checkpoint resume / tests).  sigma column-major, ld = n.  vt3 = nullable last odometry velocity.
 * Only the LOWER triangle of sigma (i >= j) is used: the device stores the covariance as its lower triangle -- the rank-m
 * downdate then reads and writes half the bytes, and the stored covariance cannot be anything but exactly symmetric -- and
 * rekf_get_state / rekf_predict_state_full return that triangle mirrored. */
int rekf_set_state(rekf_t *h, double t, int n, const double *mu, const double *sigma,
                   const double *vt3);

/* The ReflectorMatchResult of the last observation (ekf_slam_interface.h:18-26):
 * pairs are (observation index, landmark/map index).  Buffers hold up to
 * REKF_MAX_OBS entries (pairs: 2*REKF_MAX_OBS ints).  Any pointer may be NULL.
 * Synchronises. */
int rekf_get_last_match(rekf_t *h, int *n_state, int *state_pairs, int *n_map,
                        int *map_pairs, int *n_new, int *new_ids);

/* Wait for all enqueued work; returns a sticky device-side error
 * (REKF_ERR_CAPACITY / REKF_ERR_SINGULAR) once, then clears it. */
int rekf_sync(rekf_t *h);

/* The sticky REKF_FLAGBIT_* bits, without clearing them.  Every getter that synchronises (rekf_get_pose,
 * rekf_get_n, rekf_get_state, rekf_get_last_match, ...) also refreshes them and prints ONE line to stderr the
 * first time a bit appears, so a caller that never calls rekf_sync still hears about dropped reflectors.
 * Synchronises. */
int rekf_get_flags(rekf_t *h, int *flags);

/* The hipStream_t of the handle (as void*), for callers that enqueue their own work behind the filter's. */
void *rekf_stream(rekf_t *h);
/* Leading dimension (doubles) of the device covariance (lower triangle valid, see Conventions) and its device pointer; mu_dev is the CURRENT mean buffer (the
 * mean is double-buffered: every update flips between two buffers).  Enqueues whatever is held back or pending -- the last scan's
 * downdate / augmentation and the predicts the host has applied to its pose mirror only -- so that the buffers, once the stream has
 * drained (rekf_sync), hold the very state the getters return. */
int rekf_device_layout(rekf_t *h, int *ld, int *n_max, void **P_dev, void **mu_dev);

/* Measurement hooks (per-kernel hipEvent brackets, kernel timing, debug counters, fault injection) live in rekf_debug.h: they are
 * for bench.py, the profiling scripts and the tests, not part of the drop-in surface. */

const char *rekf_strerror(int code);
const char *rekf_last_hip_error(rekf_t *h);
int rekf_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* REKF_H_ */
