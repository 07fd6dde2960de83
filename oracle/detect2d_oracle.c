/*
 * oracle/detect2d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's 2D reflector detector:
 *   src/reflector_detect/laser/laser_reflector_detect.cc  (HandleLaserScan, :23-316)
 *   src/reflector_detect/laser/pose_extrapolator.cc       (default build: no USE_UNIFORM_VELOCITY)
 *   include/transform/{rigid_transform,transform}.h       (Rigid2 algebra, Project2D)
 * All file:line citations are into /root/reference.  It is a literal sequential
 * state machine, float32 where the reference is float32 (geometry, the accumulated
 * beam angle, the point TIME stored in a Vector3f -- sensor_data.h:18) and double
 * where it is double (poses, odometry).
 *
 * PARITY UNPINNED: the reference has no tests/fixtures for this path and cannot be
 * built here (ROS sensor_msgs, Eigen, glog absent).  Pinned against drift and against a
 * second implementation only: hand-checkable micro cases in tests/, and the independent
 * data-parallel numpy statement tests/witness/detect2d_witness.py, which must agree bit for
 * bit (tests/test_witness_cpu.py; its vectors: tests/golden/witness_frontends.npz).
 *
 * Defined behaviour where the reference has UB (DESIGN.md quirk register):
 *   Q13  no bright beam in range  -> empty observation (reference: front() on an empty deque)
 *        no beam inside the message range -> empty observation (reference: back() on empty vector)
 *        a bright beam before any valid point -> skipped (reference: point_cloud.back() on empty)
 *
 * Compile with -ffp-contract=off (the reference build has no FMA).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    double time, px, py, qz, qw, vx, vy, wz;   /* sensor::OdometryData fields the extrapolator reads */
} od2_odom;

typedef struct od2 {
    double intensity_min, reflector_min_length, reflector_length_error;   /* laser_reflector_detect.h:8-15 */
    float range_min, range_max;
    double s2b[3];                 /* Project2D(sensor_to_base_link): x, y, yaw (transform.h:93-98) */
    od2_odom *odom; int n_odom, cap_odom;   /* PoseExtrapolator::odometry_data_ */
    float *returns; int n_returns, cap_returns;   /* range_data_.returns */
} od2_t;

typedef struct { double x, y, a; } rigid2d;   /* transform::Rigid2d: translation + Rotation2D angle */
typedef struct { float x, y, a; } rigid2f;

/* ---- Rigid2 algebra: rigid_transform.h:46-51,62-67,87-102 ------------------ */
static rigid2d r2_inverse(rigid2d r)
{
    rigid2d o;
    const double c = cos(-r.a), s = sin(-r.a);      /* rotation_.inverse() = Rotation2D(-angle) */
    o.a = -r.a;
    o.x = -(c * r.x + (-s) * r.y);                  /* -(rotation * translation_) */
    o.y = -(s * r.x + c * r.y);
    return o;
}
static rigid2d r2_mul(rigid2d l, rigid2d r)
{
    rigid2d o;
    const double c = cos(l.a), s = sin(l.a);
    o.x = (c * r.x + (-s) * r.y) + l.x;             /* lhs.rotation() * rhs.translation() + lhs.translation() */
    o.y = (s * r.x + c * r.y) + l.y;
    o.a = l.a + r.a;                                /* Rotation2D * Rotation2D adds the angles */
    return o;
}
static rigid2f r2_cast(rigid2d r)
{
    rigid2f o; o.x = (float)r.x; o.y = (float)r.y; o.a = (float)r.a; return o;
}
static void r2f_apply(rigid2f r, float px, float py, float *ox, float *oy)
{
    const float c = cosf(r.a), s = sinf(r.a);        /* Rotation2D<float>::toRotationMatrix */
    *ox = (c * px + (-s) * py) + r.x;
    *oy = (s * px + c * py) + r.y;
}

/* ---- PoseExtrapolator (pose_extrapolator.cc) -------------------------------- */
static rigid2d interpolator(const od2_odom *st, double time)       /* :102-129 */
{
    rigid2d o;
    if (st->time <= time) {
        const double delta_t = st->time - time;
        const double odom_yaw = 2 * atan2(st->qz, st->qw);
        const double now_yaw = odom_yaw - st->wz * delta_t;
        o.x = st->px - st->vx * delta_t * cos(now_yaw) + st->vy * delta_t * sin(now_yaw);
        o.y = st->py - st->vx * delta_t * sin(now_yaw) - st->vy * delta_t * cos(now_yaw);
        o.a = now_yaw;
        return o;
    }
    {
        const double delta_t = time - st->time;
        const double odom_yaw = 2 * atan2(st->qz, st->qw);
        const double now_yaw = odom_yaw - st->wz * delta_t;   /* sign as in the reference (Q14) */
        o.x = st->px + st->vx * delta_t * cos(now_yaw) - st->vy * delta_t * sin(now_yaw);
        o.y = st->py + st->vx * delta_t * sin(now_yaw) + st->vy * delta_t * cos(now_yaw);
        o.a = now_yaw;
        return o;
    }
}
static rigid2d extrapolator_pose(const od2_t *d, double time)      /* :34-84 */
{
    rigid2d id = {0, 0, 0};
    if (d->n_odom == 0) return id;
    const od2_odom *front = &d->odom[0], *back = &d->odom[d->n_odom - 1];
    if (time <= front->time) return interpolator(front, time);
    if (time >= back->time) return interpolator(back, time);
    /* :76-82: the loop overwrites without break -> always ends on the LAST sample (Q14) */
    return interpolator(back, time);
}
static void trim_by_time(od2_t *d, double time)                    /* :12-26 */
{
    int drop = 0;
    while (d->n_odom - drop > 1 && d->odom[drop].time < time) drop++;
    if (drop) {
        memmove(d->odom, d->odom + drop, sizeof(od2_odom) * (size_t)(d->n_odom - drop));
        d->n_odom -= drop;
    }
}

/* ---- API ---------------------------------------------------------------------- */
od2_t *od2_create(double intensity_min, double reflector_min_length, double reflector_length_error,
                  float range_min, float range_max, const double s2b_xyyaw[3])
{
    od2_t *d = (od2_t *)calloc(1, sizeof(od2_t));
    d->intensity_min = intensity_min;
    d->reflector_min_length = reflector_min_length;
    d->reflector_length_error = reflector_length_error;
    d->range_min = range_min; d->range_max = range_max;
    memcpy(d->s2b, s2b_xyyaw, sizeof(double) * 3);
    return d;
}
void od2_destroy(od2_t *d)
{
    if (!d) return;
    free(d->odom); free(d->returns); free(d);
}
void od2_handle_odometry(od2_t *d, double t, double px, double py, double qz, double qw,
                         double vx, double vy, double wz)           /* pose_extrapolator.cc:28-32 */
{
    if (d->n_odom == d->cap_odom) {
        d->cap_odom = d->cap_odom ? 2 * d->cap_odom : 64;
        d->odom = (od2_odom *)realloc(d->odom, sizeof(od2_odom) * (size_t)d->cap_odom);
    }
    od2_odom o = {t, px, py, qz, qw, vx, vy, wz};
    d->odom[d->n_odom++] = o;
}
int od2_get_returns(const od2_t *d, float *out, int cap)
{
    const int n = d->n_returns < cap ? d->n_returns : cap;
    if (out && n > 0) memcpy(out, d->returns, sizeof(float) * 2 * (size_t)n);
    return d->n_returns;
}

typedef struct { float x, y, t; } tpoint;                             /* Eigen::Vector3f (x, y, time) */
typedef struct { tpoint *p; int *id; int n, cap; } run_t;

static void run_push(run_t *r, tpoint p, int id)
{
    if (r->n == r->cap) {
        r->cap = r->cap ? 2 * r->cap : 32;
        r->p = (tpoint *)realloc(r->p, sizeof(tpoint) * (size_t)r->cap);
        r->id = (int *)realloc(r->id, sizeof(int) * (size_t)r->cap);
    }
    r->p[r->n] = p; r->id[r->n] = id; r->n++;
}
static run_t run_copy(const run_t *r)
{
    run_t o; o.n = r->n; o.cap = r->n ? r->n : 1;
    o.p = (tpoint *)malloc(sizeof(tpoint) * (size_t)o.cap);
    o.id = (int *)malloc(sizeof(int) * (size_t)o.cap);
    memcpy(o.p, r->p, sizeof(tpoint) * (size_t)r->n);
    memcpy(o.id, r->id, sizeof(int) * (size_t)r->n);
    return o;
}

/*
 * HandleLaserScan (laser_reflector_detect.cc:23-316).
 * Returns the number of reflector centres written to centers_xy (<= max_centers), or
 *   -1 invalid message (the reference LOG(ERROR)+exit(-1)s, :27-38)
 *   -2 max_centers too small.
 * obs_time gets observation.time_ (= stamp; USE_CORRECT_TIME is never defined, :308-310).
 */
int od2_handle_scan(od2_t *d, double stamp, float angle_min, float angle_max, float angle_increment,
                    float scan_time, float msg_range_min, float msg_range_max,
                    const float *ranges, const float *intensities, int N,
                    float *centers_xy, int max_centers, double *obs_time)
{
    if (obs_time) *obs_time = stamp;                                        /* :26 */
    if (msg_range_min < 0 || msg_range_max <= msg_range_min) return -1;      /* :27-32 */
    if (angle_increment < 0.f && angle_max <= angle_min) return -1;          /* :33-38 */
    d->n_returns = 0;
    if (N <= 0) return 0;

    run_t *clusters = NULL; int n_clusters = 0, cap_clusters = 0;            /* reflector_points */
    int *cluster_first_id = NULL;                                            /* reflector_ids[k].front() */
    int n_cluster_ids = 0;                                                   /* reflector_ids.size() */
    run_t cur = {0};                                                         /* reflector / reflector_id */
    tpoint *cloud = (tpoint *)malloc(sizeof(tpoint) * (size_t)N);            /* point_cloud */
    int n_cloud = 0;

    const double last_point_time = stamp;                                    /* :48 */
    const double point_delta_t = (double)(scan_time / (float)N);             /* :49 float / size_t -> float */
    const double first_point_time = last_point_time - scan_time;            /* :50 */
    float angle = angle_min;                                                 /* :51 */
    trim_by_time(d, first_point_time);                                       /* :52-53 */
    rigid2d s2b_d = {d->s2b[0], d->s2b[1], d->s2b[2]};
    const rigid2f s2b = r2_cast(s2b_d);                                      /* :54 */
    const int is_circle_scan = (angle_max - angle_min - 2 * M_PI) < 1e-6;    /* :55 (no fabs: Q13) */

#define PUSH_CLUSTER(run, with_id)                                                         \
    do {                                                                                   \
        if (n_clusters == cap_clusters) {                                                  \
            cap_clusters = cap_clusters ? 2 * cap_clusters : 16;                           \
            clusters = (run_t *)realloc(clusters, sizeof(run_t) * (size_t)cap_clusters);   \
            cluster_first_id = (int *)realloc(cluster_first_id, sizeof(int) * (size_t)cap_clusters); \
        }                                                                                  \
        clusters[n_clusters] = run_copy(run);                                              \
        if (with_id) { cluster_first_id[n_cluster_ids++] = (run)->id[0]; }                  \
        n_clusters++;                                                                      \
    } while (0)

    for (int i = 0; i < N; ++i) {                                            /* :60 */
        const float range = ranges[i];
        if (range >= msg_range_min && range <= msg_range_max) {              /* :65 */
            const float nx = range * cosf(angle), ny = range * sinf(angle);  /* :68 */
            tpoint p;
            r2f_apply(s2b, nx, ny, &p.x, &p.y);                              /* :70 */
            p.t = (float)(first_point_time + i * point_delta_t);             /* :71-72: stored in a Vector3f */
            cloud[n_cloud++] = p;
        }
        if (d->range_min <= range && range <= d->range_max) {                /* :77 */
            const double intensity = intensities[i];                         /* :80 */
            if (intensity > d->intensity_min && n_cloud > 0) {               /* :82 (+ our n_cloud guard) */
                const tpoint back = cloud[n_cloud - 1];                      /* point_cloud.back() */
                if (cur.n == 0) {                                            /* :85-90 */
                    run_push(&cur, back, i);
                } else {
                    const int last_id = cur.id[cur.n - 1];                   /* :93 */
                    if (i - last_id == 1) {                                  /* :96-101 */
                        run_push(&cur, back, i);
                    } else {
                        int detected_gap = 0;
                        if (i - last_id < 4 && fabs(ranges[i] - ranges[last_id]) < 0.3 &&
                            intensities[i + 1 < N ? i + 1 : i] > d->intensity_min) {   /* :111 */
                            detected_gap = 1;
                            for (int j = last_id + 1; j < i; ++j) {          /* :115-130 */
                                const float range_gap = ranges[j];
                                const float angle_gap = angle - angle_increment * (i - j);
                                if (isinf(range_gap)) continue;
                                tpoint g;
                                r2f_apply(s2b, range_gap * cosf(angle_gap), range_gap * sinf(angle_gap), &g.x, &g.y);
                                g.t = (float)(first_point_time + j * point_delta_t);
                                run_push(&cur, g, j);
                            }
                            run_push(&cur, back, i);                         /* :135-136 */
                        }
                        if (!detected_gap) {                                 /* :140-169 */
                            const float len = hypotf(cur.p[0].x - cur.p[cur.n - 1].x, cur.p[0].y - cur.p[cur.n - 1].y);
                            if ((is_circle_scan && cur.id[0] == 0) ||
                                fabs(len - d->reflector_min_length) < d->reflector_length_error)   /* :151 */
                                PUSH_CLUSTER(&cur, 1);                        /* :153-154 */
                            cur.n = 0;                                       /* :159-160 */
                            run_push(&cur, back, i);                         /* :163-164 */
                        }
                    }
                }
            }
        }
        angle += angle_increment;                                            /* :175 float accumulation (Q15) */
    }

    /* ---- last / first reflector (:178-236) ---- */
    if (cur.n > 0) {
        if (n_clusters > 0) {
            const int first_reflector_first_point_id = (n_cluster_ids > 0) ? cluster_first_id[0] : -1;
            const int last_reflector_last_point_id = cur.id[cur.n - 1];
            const tpoint first_point = clusters[0].p[0];
            const tpoint first_reflector_last_point = clusters[0].p[clusters[0].n - 1];
            const tpoint last_point = cur.p[cur.n - 1];
            const tpoint last_reflector_first_point = cur.p[0];
            const float dxl = last_point.x - first_point.x, dyl = last_point.y - first_point.y;
            if (is_circle_scan && first_reflector_first_point_id == 0 &&
                last_reflector_last_point_id == N - 1 && sqrtf(dxl * dxl + dyl * dyl) < 0.1) {   /* :188-190 */
                for (int q = 0; q < cur.n; ++q) run_push(&clusters[0], cur.p[q], cur.id[q]);   /* :193-194 */
            } else {
                const float len = hypotf(cur.p[0].x - cur.p[cur.n - 1].x, cur.p[0].y - cur.p[cur.n - 1].y);
                if (fabs(len - d->reflector_min_length) < d->reflector_length_error)
                    PUSH_CLUSTER(&cur, 0);                                   /* :202 (ids not pushed) */
            }
            if (is_circle_scan && last_reflector_last_point_id == 0) {       /* :205-214 */
                const float fx = first_reflector_last_point.x - last_reflector_first_point.x;
                const float fy = first_reflector_last_point.y - last_reflector_first_point.y;
                const float first_len = sqrtf(fx * fx + fy * fy);
                if (fabs(first_len - d->reflector_min_length) >= d->reflector_length_error) {
                    free(clusters[0].p); free(clusters[0].id);
                    memmove(clusters, clusters + 1, sizeof(run_t) * (size_t)(n_clusters - 1));
                    n_clusters--;
                }
            }
        } else {
            const float len = hypotf(cur.p[0].x - cur.p[cur.n - 1].x, cur.p[0].y - cur.p[cur.n - 1].y);   /* :218-223 */
            if (fabs(len - d->reflector_min_length) < d->reflector_length_error)
                PUSH_CLUSTER(&cur, 0);
        }
    }
    /* else: no bright beam at all -- the reference dereferences an empty deque (:226, Q13);
     * defined here as "no reflectors". */

    int result = 0;
    if (n_cloud == 0) goto done;                                             /* :239 would be UB; CHECK at :251 */

    /* ---- motion distortion correction of the whole scan (:239-259) ---- */
    {
        if (d->cap_returns < n_cloud) {
            d->cap_returns = n_cloud;
            d->returns = (float *)realloc(d->returns, sizeof(float) * 2 * (size_t)n_cloud);
        }
        const rigid2d max_time_pose = extrapolator_pose(d, (double)cloud[n_cloud - 1].t);   /* :252 */
        const rigid2d last_pose_inverse = r2_inverse(max_time_pose);         /* :253 */
        for (int i = 0; i < n_cloud; ++i) {
            const rigid2d pose = extrapolator_pose(d, (double)cloud[i].t);   /* :249 */
            const rigid2f rel = r2_cast(r2_mul(last_pose_inverse, pose));    /* :257 */
            r2f_apply(rel, cloud[i].x, cloud[i].y, &d->returns[2 * i], &d->returns[2 * i + 1]);
        }
        d->n_returns = n_cloud;

        if (n_clusters == 0) goto done;                                      /* :272-275 */
        if (n_clusters > max_centers) { result = -2; goto done; }
        const rigid2f to_base = r2_cast(r2_inverse(max_time_pose));          /* :299 */
        for (int k = 0; k < n_clusters; ++k) {                                /* :277-306 */
            float cx = 0.f, cy = 0.f;
            for (int q = 0; q < clusters[k].n; ++q) {
                const tpoint p = clusters[k].p[q];
                const rigid2f pose = r2_cast(extrapolator_pose(d, (double)p.t));   /* :287,:293 */
                float ox, oy, bx, by;
                r2f_apply(pose, p.x, p.y, &ox, &oy);                          /* point in odom */
                r2f_apply(to_base, ox, oy, &bx, &by);                         /* back to base_link at scan end */
                cx += bx; cy += by;                                           /* :300-304 float32 running sum */
            }
            centers_xy[2 * k] = cx / (float)clusters[k].n;                    /* :305 */
            centers_xy[2 * k + 1] = cy / (float)clusters[k].n;
        }
        result = n_clusters;
    }
done:
    for (int k = 0; k < n_clusters; ++k) { free(clusters[k].p); free(clusters[k].id); }
    free(clusters); free(cluster_first_id); free(cur.p); free(cur.id); free(cloud);
    return result;
}
