/*
 * rdet.h -- C ABI of the MI355X-native reflector detectors (librdet.so).
 *
 * Drop-in boundary underneath the reference's C++ interface
 * reflector_detect::ReflectorDetectInterface
 * (reference include/reflector_detect/reflector_detect_interface.h:23-38) and its two
 * implementations LaserReflectorDetect (src/reflector_detect/laser/laser_reflector_detect.cc)
 * and PointCloudReflectorDetect (src/reflector_detect/point_cloud/point_cloud_reflector_detect.cc).
 * The ROS message types do not cross this boundary: the C++ adapter
 * (include/reflector_ekf_slam_amd/detect_adapter.hpp) unpacks sensor_msgs::LaserScan /
 * PointCloud2 into the plain arrays below.
 *
 * Conventions: opaque handles, caller-owned HOST buffers borrowed for the call, 0 / negative
 * error code, never exit() (the reference exit(-1)s on malformed scans,
 * laser_reflector_detect.cc:27-38), not thread-safe, one HIP stream per handle.  Both
 * handle_* calls are synchronous because the reference interface returns the Observation.
 */
#ifndef RDET_H_
#define RDET_H_

#ifdef __cplusplus
extern "C" {
#endif

#define RDET_ABI_VERSION 1
#define RDET_MAX_CENTERS 256     /* most reflectors one scan / cloud may yield */

enum {
    RDET_OK = 0,
    RDET_ERR_INVALID = -1,       /* bad argument */
    RDET_ERR_HIP = -2,
    RDET_ERR_BAD_SCAN = -3,      /* range_min/max or angle fields malformed (reference: exit(-1)) */
    RDET_ERR_CAPACITY = -4,      /* more beams / points / reflectors than the handle was created for */
    RDET_ERR_BUFFER = -5
};

/* reflector_detect::ReflectorDetectOptions (laser_reflector_detect.h:8-15) */
typedef struct rdet2d_options {
    double intensity_min;
    double reflector_min_length;
    double reflector_length_error;
    float range_min;
    float range_max;
} rdet2d_options;

typedef struct rdet2d rdet2d_t;

/* LaserReflectorDetect(options) + SetSensorToBaseLinkTransform(pose): the transform is passed
 * already projected to 2D (transform::Project2D, transform.h:93-98): x, y, yaw. */
int rdet2d_create(const rdet2d_options *opt, const double sensor_to_base_link_xyyaw[3],
                  int max_beams, int device, rdet2d_t **out);
void rdet2d_destroy(rdet2d_t *h);
int rdet2d_set_sensor_to_base_link(rdet2d_t *h, const double xyyaw[3]);

/* HandleOdometryData (laser_reflector_detect.cc:318-322 -> pose_extrapolator.cc:28-32).
 * quat_zw = (orientation.z, orientation.w): the only components the extrapolator reads. */
int rdet2d_handle_odometry(rdet2d_t *h, double t, const double pos_xy[2], const double quat_zw[2],
                           double vx, double vy, double wz);

/* HandleLaserScan (laser_reflector_detect.cc:23-316).  The scalar arguments are the
 * sensor_msgs::LaserScan header fields; ranges / intensities hold N beams.
 * Out: K reflector centres (base_link frame, de-skewed to the scan end) in centers_xy
 * (capacity max_centers pairs), obs_time = observation.time_. */
int rdet2d_handle_scan(rdet2d_t *h, double stamp, float angle_min, float angle_max,
                       float angle_increment, float scan_time, float range_min, float range_max,
                       const float *ranges, const float *intensities, int N,
                       float *centers_xy, int max_centers, int *K, double *obs_time);

/* GetRangeData (laser_reflector_detect.h:24): origin + de-skewed returns of the last scan. */
int rdet2d_get_range_data(rdet2d_t *h, float origin_xy[2], float *returns_xy, int cap_points,
                          int *n_returns);

/* reflector_detect::PointCloudOptions (point_cloud_reflector_detect.h:37-40) */
typedef struct rdet3d_options {
    double intensity_min;
} rdet3d_options;

typedef struct rdet3d rdet3d_t;

int rdet3d_create(const rdet3d_options *opt, const double sensor_to_base_link_xyyaw[3],
                  int max_points, int device, rdet3d_t **out);
void rdet3d_destroy(rdet3d_t *h);

/* HandlePointCloud (point_cloud_reflector_detect.cc:9-106): xyzi = N points (x, y, z,
 * intensity) as pcl::fromROSMsg would deliver them.  Out: K centres in base_link. */
int rdet3d_handle_cloud(rdet3d_t *h, double stamp, const float *xyzi, int N,
                        float *centers_xy, int max_centers, int *K, double *obs_time);

/* The same in two halves, for a caller that wants the next cloud's copy and launches to run while the device is still on this one
 * (a node whose callback hands clouds over back to back; the reference's own callback, point_cloud_reflector_detect.cc:9-106, is the
 * synchronous call above).  rdet3d_submit: the cloud into device memory and the kernels enqueued, no waiting.  rdet3d_collect: the
 * centres of the OLDEST cloud submitted and not yet collected (blocks until they are there).  At most two clouds may be submitted and
 * not collected (a third submit returns RDET_ERR_INVALID, as does a collect with nothing submitted, or rdet3d_handle_cloud in
 * between); results are those of rdet3d_handle_cloud called in the same order. */
int rdet3d_submit(rdet3d_t *h, double stamp, const float *xyzi, int N, int max_centers);
int rdet3d_collect(rdet3d_t *h, float *centers_xy, int max_centers, int *K, double *obs_time);

const char *rdet_strerror(int code);
int rdet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RDET_H_ */
