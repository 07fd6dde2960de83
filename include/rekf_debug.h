/*
 * rekf_debug.h -- measurement and test hooks of librekf.so (bench.py, scripts/, tests/).  NOT part of the drop-in boundary
 * (include/rekf.h): nothing a replacement of ekf::ReflectorEKFSLAMInterface needs is declared here.
 */
#ifndef REKF_DEBUG_H_
#define REKF_DEBUG_H_

#include "rekf.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    REKF_K_PREDICT = 0,   /* (never recorded since ABI 4: odometry messages launch nothing) */
    REKF_K_FRONT = 1,     /* predict + ReflectorMatch + H rows       */
    REKF_K_GATHER = 2,    /* (round 1's separate kernels; never recorded since k_mid fused them) */
    REKF_K_SOLVE = 3,
    REKF_K_GAIN = 4,
    REKF_K_DOWNDATE = 5,  /* P -= K (H P)  (the roofline kernel)     */
    REKF_K_AUGMENT = 6,   /* new landmarks                           */
    REKF_K_EMPTY = 7,     /* an event pair around nothing: the bracket's own cost, in situ */
    REKF_K_UPDATE = 8,    /* ONE bracket around the whole HandleObservationMessage chain (per-update latency);
                           * its individual readings are kept, see rekf_profile_samples */
    REKF_K_MID = 9,       /* gather + solve + gain in one launch: W = P H^T, (H P)^T, S^-1, K = W S^-1, mu += K dz */
    REKF_K_COUNT = 10
};
/* When on, kernel launches are bracketed by hipEvents on the handle's stream: `on` is a bit mask
 * over the REKF_K_* ids (1 << id); -1 = all.  Brackets perturb the stream (each costs a few
 * microseconds of command-processor time), so measure one kernel at a time for absolute numbers. */
int rekf_profile_enable(rekf_t *h, int on);
/* Sum (microseconds) and count of the recorded launches of kernel k since the
 * last rekf_profile_reset.  Synchronises. */
int rekf_profile_read(rekf_t *h, int k, double *total_us, long *count);
int rekf_profile_reset(rekf_t *h);
/* The individual REKF_K_UPDATE readings (microseconds, in call order) since the last reset: up to cap values
 * into out_us, *count = how many exist.  For the median / p99 per-update latency of SURVEY 8(d).  Synchronises. */
int rekf_profile_samples(rekf_t *h, float *out_us, long cap, long *count);

/* Time `reps` back-to-back launches of the covariance downdate (kernel = REKF_K_DOWNDATE) on the
 * panels left by the last observation, between ONE hipEvent pair; the filter state is not meaningful afterwards
 * (snapshot / restore it with rekf_get_state / rekf_set_state).  `ablate` must be 0 (reserved). */
int rekf_debug_time_kernel(rekf_t *h, int kernel, int reps, int ablate, double *avg_us);

/* Debug builds (-DREKF_DEBUG_TIMING) let kernels drop cycle counters here.  In a release build: out32[20] / [21] = scans whose
 * SPECULATIVE match record k_mid proved / of those, scans with observations it had to re-match; [22] / [23] = scans that met a pending
 * downdate / of those, scans that computed its correction themselves (no write-ahead panel); [24] = work items the in-launch downdate
 * roles of the last two launches asked for (> 0: the one-launch form ran); [26] / [27] = device time stamps (100 MHz) of the downdate
 * role's first start / last end in the last launch.  The environment knobs of the library (read at rekf_create; twins for the
 * bit-identity tests and A/B measurements, same results): REKF_SPEC=0 (no speculative match), REKF_SCAN_LAUNCH=0 (two-launch chain),
 * REKF_EXCLUSIVE=1 (= rekf_set_exclusive). */
int rekf_debug_counters(rekf_t *h, long long out32[32]);

/* Fault injection (tests): the NEXT rekf_handle_observation fails with REKF_ERR_HIP at `stage` as if a HIP call had:
 *   1 = while staging a wide scan's observations, 2 = while enqueueing the held-back downdate, 3 = at the launch check behind the
 * chain (the kernels HAVE been enqueued).  The contract under test: the state stays valid -- after stages 1 and 2 the scan was not
 * applied at all (hand it over again), after stage 3 it was. */
int rekf_debug_inject_failure(rekf_t *h, int stage);

/* The match grid (csrc/ekf_dev.h, RekfCtl::grid_state: k_mid matches a host-predicted scan itself from the few landmarks in the 3 x 3 grid
 * cells around each observation).  on = 0: off (the scan's front end is a launch of its own again: the twin the parity tests compare with);
 * drift_limit > 0: replaces the 0.3 m bound on how far a landmark may stand from its binning position (tiny values force rebuilds and the
 * full-sweep fall-back); mask >= 0 (2^k - 1): a table of mask + 1 buckets (overflow: the library stops using the grid by itself).
 * rekf_debug_counters: out32[18] = scans k_mid matched itself, out32[19] = of those, by the full sweep (grid invalid at that moment),
 * out32[17] = k_grid_build launches so far, out32[16] = the grid is still in use (1) or was given up / switched off (0). */
int rekf_debug_set_grid(rekf_t *h, int on, double drift_limit, int mask);

#ifdef __cplusplus
}
#endif
#endif /* REKF_DEBUG_H_ */
