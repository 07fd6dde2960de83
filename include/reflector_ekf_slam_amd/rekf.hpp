// rekf.hpp -- header-only C++ RAII wrappers over the C ABI (include/rekf.h, include/rdet.h).
// Plain std types only, so it compiles without Eigen/ROS; the Eigen/ROS-typed adapter that
// derives from the reference's own abstract classes sits on top of this in
// ekf_slam_adapter.hpp / detect_adapter.hpp.
#pragma once
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../rdet.h"
#include "../rekf.h"

namespace rekfpp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &where) : std::runtime_error(where + ": " + rekf_strerror(c)), code(c) {}
};

struct MatchResult {   // ekf::ReflectorMatchResult (ekf_slam_interface.h:18-26)
    std::vector<std::pair<int, int>> map_obs_match_ids, state_obs_match_ids;
    std::vector<int> new_ids;
};

class EkfSlam {
public:
    // auto_grow: the reference never runs out of room (it resizes on every augment, reflector_ekf_slam.cc:316-363);
    // with it max_landmarks is only the initial reservation and the capacity doubles on demand (rekf_reserve)
    EkfSlam(const rekf_options &opt, int max_landmarks, int device = 0, bool auto_grow = true)
    {
        const int rc = rekf_create(&opt, max_landmarks, device, &h_);
        if (rc != REKF_OK) throw Error(rc, "rekf_create");
        const int rg = rekf_set_auto_grow(h_, auto_grow ? 1 : 0);
        if (rg != REKF_OK) { rekf_destroy(h_); h_ = nullptr; throw Error(rg, "rekf_set_auto_grow"); }
    }
    ~EkfSlam() { rekf_destroy(h_); }
    EkfSlam(const EkfSlam &) = delete;
    EkfSlam &operator=(const EkfSlam &) = delete;

    void HandleOdometry(double t, double vx, double vy, double wz) { chk(rekf_handle_odometry(h_, t, vx, vy, wz), "HandleOdometryMessage"); }
    void HandleObservation(double t, const float *xy, int K, const double *gps_pose3 = nullptr)
    {
        chk(rekf_handle_observation(h_, t, xy, K, gps_pose3), "HandleObservationMessage");
    }
    void SetMap(const std::vector<float> &xy, const std::vector<double> &cov)
    {
        chk(rekf_set_map(h_, xy.data(), cov.data(), (int)(xy.size() / 2)), "rekf_set_map");
    }
    void Reserve(int max_landmarks) { chk(rekf_reserve(h_, max_landmarks), "rekf_reserve"); }
    int Capacity() { int c = 0; chk(rekf_get_capacity(h_, &c), "rekf_get_capacity"); return c; }
    double LatestTime() const { double t = 0; rekf_get_time(h_, &t); return t; }
    int Dim() { int n = 0; chk(rekf_get_n(h_, &n), "rekf_get_n"); return n; }
    void Pose(double mu3[3], double sigma3x3[9]) { double t; chk(rekf_get_pose(h_, &t, mu3, sigma3x3), "rekf_get_pose"); }
    void PredictPose(double t, double mu3[3], double sigma3x3[9]) { chk(rekf_predict_state(h_, t, mu3, sigma3x3), "PredictState"); }
    // PredictState as the interface returns it (ekf_slam_interface.h:59): full mu / sigma, time = the state's own (cc:99)
    void PredictStateFull(double t_query, double &t_state, std::vector<double> &mu, std::vector<double> &sigma)
    {
        int n = Dim();
        mu.resize((size_t)n);
        sigma.resize((size_t)n * n);
        chk(rekf_predict_state_full(h_, t_query, &t_state, &n, mu.data(), (long)mu.size(), sigma.data(), (long)sigma.size()),
            "PredictState");
    }
    // sticky REKF_FLAGBIT_* bits (capacity overflow = reflectors dropped; non-PD innovation covariance), not cleared
    int Flags() { int f = 0; chk(rekf_get_flags(h_, &f), "rekf_get_flags"); return f; }
    // full state: mu (n) and sigma (n*n, column-major like Eigen::MatrixXd)
    void State(double &t, std::vector<double> &mu, std::vector<double> &sigma)
    {
        int n = Dim();
        mu.resize((size_t)n);
        sigma.resize((size_t)n * n);
        chk(rekf_get_state(h_, &t, &n, mu.data(), (long)mu.size(), sigma.data(), (long)sigma.size()), "GetState");
    }
    // Node::ReflectorToRosMarkers' numbers (src/ros_node.cc:750-765), one per landmark, computed on the device
    struct MarkerEllipse { double x, y, angle, x_len, y_len; };
    std::vector<MarkerEllipse> MarkerEllipses()
    {
        const int cap = (Dim() - 3) / 2;
        std::vector<MarkerEllipse> e((size_t)(cap > 0 ? cap : 0));
        int k = 0;
        static_assert(sizeof(MarkerEllipse) == 5 * sizeof(double), "packed");
        chk(rekf_get_marker_ellipses(h_, cap > 0 ? &e[0].x : nullptr, cap, &k), "rekf_get_marker_ellipses");
        e.resize((size_t)k);
        return e;
    }
    MatchResult LastMatch()
    {
        int ns = 0, nm = 0, nn = 0;
        std::vector<int> sp(2 * REKF_MAX_OBS), mp(2 * REKF_MAX_OBS), nw(REKF_MAX_OBS);
        chk(rekf_get_last_match(h_, &ns, sp.data(), &nm, mp.data(), &nn, nw.data()), "rekf_get_last_match");
        MatchResult r;
        for (int i = 0; i < ns; ++i) r.state_obs_match_ids.emplace_back(sp[2 * i], sp[2 * i + 1]);
        for (int i = 0; i < nm; ++i) r.map_obs_match_ids.emplace_back(mp[2 * i], mp[2 * i + 1]);
        r.new_ids.assign(nw.begin(), nw.begin() + nn);
        return r;
    }
    void Sync() { chk(rekf_sync(h_), "rekf_sync"); }
    rekf_t *handle() { return h_; }

private:
    static void chk(int rc, const char *where) { if (rc != REKF_OK) throw Error(rc, where); }
    rekf_t *h_ = nullptr;
};

class LaserDetector {
public:
    LaserDetector(const rdet2d_options &opt, const double s2b_xyyaw[3], int max_beams = 8192, int device = 0)
    {
        const int rc = rdet2d_create(&opt, s2b_xyyaw, max_beams, device, &h_);
        if (rc != RDET_OK) throw std::runtime_error(std::string("rdet2d_create: ") + rdet_strerror(rc));
    }
    ~LaserDetector() { rdet2d_destroy(h_); }
    LaserDetector(const LaserDetector &) = delete;
    LaserDetector &operator=(const LaserDetector &) = delete;
    void HandleOdometry(double t, double px, double py, double qz, double qw, double vx, double vy, double wz)
    {
        const double p[2] = {px, py}, q[2] = {qz, qw};
        rdet2d_handle_odometry(h_, t, p, q, vx, vy, wz);
    }
    // returns observation.time_; centres (x, y pairs) in `centers`
    double HandleScan(double stamp, float angle_min, float angle_max, float angle_increment, float scan_time,
                      float range_min, float range_max, const std::vector<float> &ranges,
                      const std::vector<float> &intensities, std::vector<float> &centers)
    {
        centers.assign(2 * RDET_MAX_CENTERS, 0.f);
        int K = 0;
        double t = stamp;
        const int rc = rdet2d_handle_scan(h_, stamp, angle_min, angle_max, angle_increment, scan_time, range_min, range_max,
                                          ranges.data(), intensities.data(), (int)ranges.size(), centers.data(),
                                          RDET_MAX_CENTERS, &K, &t);
        if (rc != RDET_OK) throw std::runtime_error(std::string("HandleLaserScan: ") + rdet_strerror(rc));
        centers.resize((size_t)2 * K);
        return t;
    }
    rdet2d_t *handle() { return h_; }

private:
    rdet2d_t *h_ = nullptr;
};

}  // namespace rekfpp
