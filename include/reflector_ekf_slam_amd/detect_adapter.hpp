// detect_adapter.hpp -- drop-in replacements for reflector_detect::LaserReflectorDetect and
// PointCloudReflectorDetect, compiled INSIDE the reference tree (needs ROS sensor_msgs, PCL
// conversions and the reference's headers; guarded, see ekf_slam_adapter.hpp).
#pragma once
#if __has_include(<sensor_msgs/LaserScan.h>) && __has_include("reflector_detect/reflector_detect_interface.h")
#include <cstdlib>
#include <iostream>

#include <sensor_msgs/point_cloud2_iterator.h>

#include "reflector_detect/laser/laser_reflector_detect.h"
#include "reflector_detect/point_cloud/point_cloud_reflector_detect.h"
#include "rekf.hpp"

namespace reflector_detect {

class LaserReflectorDetectHip : public ReflectorDetectInterface {
public:
    explicit LaserReflectorDetectHip(const ReflectorDetectOptions &o, int max_beams = 8192, int device = 0)
    {
        rdet2d_options c{o.intensity_min, o.reflector_min_length, o.reflector_length_error, o.range_min, o.range_max};
        const double s2b[3] = {0, 0, 0};
        if (rdet2d_create(&c, s2b, max_beams, device, &h_) != RDET_OK) std::exit(-1);
    }
    ~LaserReflectorDetectHip() override { rdet2d_destroy(h_); }
    void SetSensorToBaseLinkTransform(const transform::Rigid3d &pose) override
    {
        sensor_to_base_link_transform_ = pose;
        const auto p2 = transform::Project2D(pose);                       // transform.h:93-98
        const double s2b[3] = {p2.translation().x(), p2.translation().y(), p2.rotation().angle()};
        rdet2d_set_sensor_to_base_link(h_, s2b);
    }
    sensor::Observation HandleLaserScan(const sensor_msgs::LaserScanConstPtr &msg) override
    {
        std::vector<float> centers(2 * RDET_MAX_CENTERS);
        int K = 0;
        double t = 0;
        const int rc = rdet2d_handle_scan(h_, msg->header.stamp.toSec(), msg->angle_min, msg->angle_max,
                                          msg->angle_increment, msg->scan_time, msg->range_min, msg->range_max,
                                          msg->ranges.data(), msg->intensities.data(), (int)msg->ranges.size(),
                                          centers.data(), RDET_MAX_CENTERS, &K, &t);
        if (rc != RDET_OK) { std::cerr << "HandleLaserScan: " << rdet_strerror(rc) << std::endl; std::exit(-1); }   // cc:27-38
        sensor::Observation obs;
        obs.time_ = t;
        for (int k = 0; k < K; ++k) obs.cloud_.push_back(Eigen::Vector2f(centers[2 * k], centers[2 * k + 1]));
        return obs;
    }
    sensor::RangeData GetRangeData() override
    {
        int n = 0;
        float origin[2];
        rdet2d_get_range_data(h_, origin, nullptr, 0, &n);
        std::vector<float> r(2 * (size_t)(n > 0 ? n : 1));
        rdet2d_get_range_data(h_, origin, r.data(), n, &n);
        sensor::RangeData out{Eigen::Vector2f(origin[0], origin[1]), {}, {}};
        for (int i = 0; i < n; ++i) out.returns.push_back(Eigen::Vector2f(r[2 * i], r[2 * i + 1]));
        return out;
    }
    void HandleOdometryData(const sensor::OdometryData &m) override
    {
        const double p[2] = {m.position.x(), m.position.y()}, q[2] = {m.orientation.z(), m.orientation.w()};
        rdet2d_handle_odometry(h_, m.time, p, q, m.linear_velocity.x(), m.linear_velocity.y(), m.angular_velocity.z());
    }

private:
    rdet2d_t *h_ = nullptr;
};

class PointCloudReflectorDetectHip : public ReflectorDetectInterface {
public:
    explicit PointCloudReflectorDetectHip(const PointCloudOptions &o, int max_points = 1 << 18, int device = 0) : opt_(o), max_points_(max_points), device_(device) {}
    ~PointCloudReflectorDetectHip() override { rdet3d_destroy(h_); }
    void SetSensorToBaseLinkTransform(const transform::Rigid3d &pose) override
    {
        sensor_to_base_link_transform_ = pose;
        rdet3d_destroy(h_);
        h_ = nullptr;
    }
    sensor::Observation HandlePointCloud(const sensor_msgs::PointCloud2ConstPtr &msg) override
    {
        if (!h_) {
            const auto p2 = transform::Project2D(sensor_to_base_link_transform_);
            const double s2b[3] = {p2.translation().x(), p2.translation().y(), p2.rotation().angle()};
            rdet3d_options c{opt_.intensity_min};
            if (rdet3d_create(&c, s2b, max_points_, device_, &h_) != RDET_OK) std::exit(-1);
        }
        // what pcl::fromROSMsg + PointXYZI delivers (cc:30): x, y, z, intensity per point
        std::vector<float> xyzi;
        sensor_msgs::PointCloud2ConstIterator<float> ix(*msg, "x"), iy(*msg, "y"), iz(*msg, "z"), ii(*msg, "intensity");
        for (; ix != ix.end(); ++ix, ++iy, ++iz, ++ii) { xyzi.push_back(*ix); xyzi.push_back(*iy); xyzi.push_back(*iz); xyzi.push_back(*ii); }
        std::vector<float> centers(2 * RDET_MAX_CENTERS);
        int K = 0;
        double t = 0;
        const int rc = rdet3d_handle_cloud(h_, msg->header.stamp.toSec(), xyzi.data(), (int)(xyzi.size() / 4), centers.data(),
                                           RDET_MAX_CENTERS, &K, &t);
        if (rc != RDET_OK) { std::cerr << "HandlePointCloud: " << rdet_strerror(rc) << std::endl; std::exit(-1); }
        sensor::Observation obs;
        obs.time_ = t;
        for (int k = 0; k < K; ++k) obs.cloud_.push_back(Eigen::Vector2f(centers[2 * k], centers[2 * k + 1]));
        return obs;
    }

private:
    PointCloudOptions opt_;
    int max_points_, device_;
    rdet3d_t *h_ = nullptr;
};

}  // namespace reflector_detect
#endif
