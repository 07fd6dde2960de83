// ekf_slam_adapter.hpp -- drop-in replacement for ekf::ReflectorEKFSLAM that a maintainer of the
// reference compiles INSIDE the reference tree (it needs the reference's own headers and Eigen;
// neither exists on the build image of this repo, so this file is guarded and is exercised only by
// tests/cpp/adapter_smoke.cpp through its Eigen-free base, rekf.hpp).
//
// Usage in /root/reference/src/ros_node.cc:
//     slam_ = common::make_unique<ekf::ReflectorEKFSLAMHip>(options, /*max_landmarks=*/1024);
// Every virtual of ekf::ReflectorEKFSLAMInterface (ekf_slam_interface.h:50-67) is implemented;
// method names, argument meaning and the exit(-1) error behaviour of the reference are kept.
#pragma once
#if __has_include(<Eigen/Core>) && __has_include("reflector_ekf_slam/ekf_slam_interface.h")
#include <cstdlib>
#include <iostream>

#include "reflector_ekf_slam/ekf_slam_interface.h"
#include "rekf.hpp"

namespace ekf {

class ReflectorEKFSLAMHip : public ReflectorEKFSLAMInterface {
public:
    explicit ReflectorEKFSLAMHip(const EKFOptions &options, int max_landmarks = 1024, int device = 0)
        : impl_(to_c(options), max_landmarks, device)
    {
        // LoadMapFromTxtFile (reflector_ekf_slam.cc:43-95) stays host code: parse, then rekf_set_map
        LoadMapFromTxtFile(options.map_path);
    }
    ~ReflectorEKFSLAMHip() override {}

    void HandleOdometryMessage(const sensor::OdometryData &odometry) override
    {
        guard([&] { impl_.HandleOdometry(odometry.time, odometry.linear_velocity.x(), odometry.linear_velocity.y(),
                                         odometry.angular_velocity.z()); });
        mirror_valid_ = false;
    }
    void HandleImuMessage(const sensor::ImuData &) override {}          // empty in the reference (cc:224-227)
    void HandleObservationMessage(const sensor::Observation &observation) override
    {
        static_assert(sizeof(Eigen::Vector2f) == 2 * sizeof(float), "PointCloud must be contiguous float pairs");
        const float *xy = observation.cloud_.empty() ? nullptr : observation.cloud_.front().data();
        double gps[3];
        const double *gp = nullptr;
        if (observation.gps_pose_) {                                        // USE_GPS build (gps.cc:305-340)
            gps[0] = observation.gps_pose_->translation().x();
            gps[1] = observation.gps_pose_->translation().y();
            gps[2] = observation.gps_pose_->rotation().angle();
            gp = gps;
        }
        guard([&] { impl_.HandleObservation(observation.time_, xy, (int)observation.cloud_.size(), gp); });
        mirror_valid_ = false;
    }
    State PredictState(const double &time) override
    {
        // the full predicted State, as reflector_ekf_slam.cc:97-152 returns it: every mean, the n x n covariance with
        // rows/columns 0, 1 and the pose block propagated on the device, and the state's own time (`State result = state_`)
        State s;
        std::vector<double> mu, sig;
        guard([&] { impl_.PredictStateFull(time, s.time, mu, sig); });
        const int n = (int)mu.size();
        s.mu = Eigen::Map<Eigen::VectorXd>(mu.data(), n);
        s.sigma = Eigen::Map<Eigen::MatrixXd>(sig.data(), n, n);
        return s;
    }
    // 96-byte fast path for the only part src/ros_node.cc:455-470 reads of it
    void PredictPose(const double &time, Eigen::Vector3d &mu, Eigen::Matrix3d &sigma)
    {
        double c9[9];
        guard([&] { impl_.PredictPose(time, mu.data(), c9); });
        sigma = Eigen::Map<Eigen::Matrix3d>(c9);
    }
    Eigen::VectorXd &GetStateVector() override { refresh(); return mirror_.mu; }       // never called by the reference
    Eigen::MatrixXd &GetCoviarance() override { refresh(); return mirror_.sigma; }
    double GetLatestTime() override { return impl_.LatestTime(); }
    State GetState() override { refresh(); return mirror_; }
    sensor::Map GetGlobalMap() override { return map_; }

    // fast path for the pose-only consumers in ros_node.cc:515-545,638-658 (no n x n copy)
    void GetPose(Eigen::Vector3d &mu, Eigen::Matrix3d &sigma)
    {
        double c9[9];
        guard([&] { impl_.Pose(mu.data(), c9); });
        sigma = Eigen::Map<Eigen::Matrix3d>(c9);
    }

    // Node::ReflectorToRosMarkers (ros_node.cc:736-790) without the n x n GetState(): the 2x2 eigen-solves run on
    // the device; marker i gets position (x, y), orientation (0, 0, sin(angle/2), cos(angle/2)) and
    // scale (s*x_len, s*y_len, 0.1*s*(x_len+y_len)).
    std::vector<rekfpp::EkfSlam::MarkerEllipse> GetMarkerEllipses()
    {
        std::vector<rekfpp::EkfSlam::MarkerEllipse> e;
        guard([&] { e = impl_.MarkerEllipses(); });
        return e;
    }

private:
    static rekf_options to_c(const EKFOptions &o)
    {
        rekf_options c;
        c.odom_model = (o.odom_model == sensor::OdometryModel::DIFF) ? REKF_ODOM_DIFF : REKF_ODOM_OMNI;
        c.use_imu = o.use_imu ? 1 : 0;
        c.init_time = o.init_time;
        for (int i = 0; i < 3; ++i) c.init_pose[i] = o.init_pose(i);
        c.linear_velocity_cov = o.linear_velocity_cov;
        c.angular_velocity_cov = o.angular_velocity_cov;
        c.observation_cov = o.observation_cov;
        return c;
    }
    template <class F> void guard(F &&f)
    {
        try { f(); }
        catch (const std::exception &e) { std::cerr << e.what() << std::endl; std::exit(-1); }   // reference: LOG(ERROR) + exit(-1)
    }
    void refresh()
    {
        if (mirror_valid_) return;
        std::vector<double> mu, sig;
        guard([&] { impl_.State(mirror_.time, mu, sig); });
        // (the getters refresh the sticky device flags and librekf prints one loud line per new bit: a state that
        // outgrew max_landmarks drops reflectors, which the reference never does -- size max_landmarks for the site)
        const int n = (int)mu.size();
        mirror_.mu = Eigen::Map<Eigen::VectorXd>(mu.data(), n);
        mirror_.sigma = Eigen::Map<Eigen::MatrixXd>(sig.data(), n, n);      // both column-major
        mirror_valid_ = true;
    }
    void LoadMapFromTxtFile(const std::string &file)
    {
        if (file.empty() || !IsFileExist(file)) return;                     // cc:45-46
        std::ifstream in(file.c_str());
        std::string line;
        std::vector<std::vector<double>> result;
        while (getline(in, line)) {
            if (line.empty()) continue;
            std::vector<double> vec;
            for (auto &p : SplitString(line, ',')) if (!p.empty()) vec.push_back(std::stod(p));
            result.push_back(vec);
        }
        if (result.size() != 2 || result.back().size() != 2 * result.front().size()) return;   // cc:74-79
        std::vector<float> xy(result[0].begin(), result[0].end());
        impl_.SetMap(xy, result[1]);       // covariances from line 1 (the reference indexes line 0: UB, Q9)
        for (size_t i = 0; i + 1 < result[0].size(); i += 2) {
            map_.reflector_map_.push_back(Eigen::Vector2f(result[0][i], result[0][i + 1]));
            Eigen::Matrix2d c;
            c << result[1][2 * i], result[1][2 * i + 1], result[1][2 * i + 2], result[1][2 * i + 3];
            map_.reflector_map_coviarance_.push_back(c);
        }
    }

    rekfpp::EkfSlam impl_;
    State mirror_;
    bool mirror_valid_ = false;
    sensor::Map map_;
};

}  // namespace ekf
#endif
