// Compiles the Eigen-free C++ wrappers (include/reflector_ekf_slam_amd/rekf.hpp) and, on a GPU box,
// runs one create -> odometry -> observation -> state round trip.  Built and run by
// tests/test_cpp_adapter.py (compile-only without a GPU).
#include <cmath>
#include <cstdio>
#include <vector>

#include "reflector_ekf_slam_amd/detect_adapter.hpp"
#include "reflector_ekf_slam_amd/ekf_slam_adapter.hpp"
#include "reflector_ekf_slam_amd/rekf.hpp"

int main(int argc, char **)
{
    if (argc > 1) return 0;                       // compile/link check only
    rekf_options o{};
    o.odom_model = REKF_ODOM_DIFF;
    o.linear_velocity_cov = 0.0025; o.angular_velocity_cov = 0.0064; o.observation_cov = 0.0025;
    rekfpp::EkfSlam f(o, 16);
    f.HandleOdometry(0.1, 1.0, 0.0, 0.2);
    const float obs[4] = {3.0f, 1.0f, -2.0f, 0.5f};
    f.HandleObservation(0.2, obs, 2);
    f.HandleObservation(0.3, obs, 2);
    double t;
    std::vector<double> mu, sig;
    f.State(t, mu, sig);
    const auto m = f.LastMatch();
    if (mu.size() != 7 || m.state_obs_match_ids.size() != 2 || !m.new_ids.empty()) { std::printf("FAIL\n"); return 1; }
    if (!(std::fabs(sig[0] - sig[0]) == 0.0)) return 1;
    const auto ell = f.MarkerEllipses();
    if (ell.size() != 2 || !(ell[0].x_len > 0.0) || std::fabs(ell[0].x - mu[3]) > 0.0) { std::printf("FAIL ellipses\n"); return 1; }
    std::printf("ADAPTER_OK n=%zu t=%.2f x=%.6f\n", mu.size(), t, mu[0]);
    return 0;
}
