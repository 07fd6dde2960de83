// Compiles the Eigen-free C++ wrappers (include/reflector_ekf_slam_amd/rekf.hpp) and, on a GPU box,
// runs one create -> odometry -> observation -> state round trip.  Built and run by
// tests/test_cpp_adapter.py (compile-only without a GPU).
#include <cmath>
#include <cstdio>
#include <vector>

#include "reflector_ekf_slam_amd/detect_adapter.hpp"
#include "reflector_ekf_slam_amd/ekf_slam_adapter.hpp"
#include "reflector_ekf_slam_amd/rekf.hpp"
#include "reflector_ekf_slam_amd/rgrid.hpp"

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
    {   // PredictState, full (ekf_slam_interface.h:59): non-mutating, n means, state's own time, pose block = the fast path's
        double ts = -1, mu3[3], c9[9];
        std::vector<double> pm, ps;
        f.PredictStateFull(0.45, ts, pm, ps);
        f.PredictPose(0.45, mu3, c9);
        std::vector<double> mu2, sig2;
        double t2;
        f.State(t2, mu2, sig2);
        if (pm.size() != 7 || ts != t || pm[0] != mu3[0] || ps[0] != c9[0] || ps[1 + 7 * 0] != c9[1] || mu2 != mu || sig2 != sig ||
            pm[3] != mu[3] || ps[0 + 7 * 3] == sig[0 + 7 * 3] || f.Flags() != 0) { std::printf("FAIL predict_state_full\n"); return 1; }
    }
    const auto ell = f.MarkerEllipses();
    if (ell.size() != 2 || !(ell[0].x_len > 0.0) || std::fabs(ell[0].x - mu[3]) > 0.0) { std::printf("FAIL ellipses\n"); return 1; }
    // grid front-end: insert a short wall into an unknown grid, then find it again with the matcher
    {
        rekfpp::GridFrontEnd gf(4096, 200 * 200, 1 << 16);
        gf.SetGrid(std::vector<uint16_t>(200 * 200, 0), 200, 200, 0.05, 5.0, 5.0);
        rekfpp::GridFrontEnd::Cloud wall;
        for (int i = 0; i < 120; ++i) { wall.push_back(2.0f); wall.push_back(-1.5f + 0.025f * i); }
        for (int rep = 0; rep < 3; ++rep) gf.Insert({0.f, 0.f}, wall, {});
        const auto g = gf.GetGrid();
        size_t known = 0;
        for (uint16_t v : g) known += v != 0;
        const auto thin = gf.VoxelFilter(wall, 0.05f);
        rgrid_match_options mo{0.2, 0.1, 1e-1, 1e-1};
        std::array<double, 3> pe{};
        const double score = gf.Match(mo, {0.1, 0.05, 0.0}, thin, pe);
        if (known < 1000 || thin.size() >= wall.size() || !(score > 0.3) || std::fabs(pe[0]) > 0.051) {
            std::printf("FAIL grid known=%zu thin=%zu score=%f pe=%f\n", known, thin.size() / 2, score, pe[0]);
            return 1;
        }
        {   // MapBuilder::AddRangeData as one call on a fresh handle: two scans of the same wall
            rekfpp::GridFrontEnd mbf(4096, 512 * 512, 1 << 16);
            rgrid_map_builder_options mo2{0.05f, 0.025f, 0.9, 100, 50.0, {0.2, 0.26, 1e-1, 1e-1}, {1.0, 0.1, 0.4, 100, 1}, 0.55f, 0.49f, 1};
            std::array<double, 3> lp{};
            rekfpp::GridFrontEnd::Cloud local;
            const bool ok1 = mbf.AddRangeData(mo2, {0.f, 0.f}, wall, {}, {0.0, 0.0, 0.0}, lp, &local);
            const bool ok2 = mbf.AddRangeData(mo2, {0.f, 0.f}, wall, {}, {0.02, -0.01, 0.0}, lp, &local);
            if (!ok1 || !ok2 || local.size() != wall.size() || std::fabs(lp[0]) > 0.06) {
                std::printf("FAIL add_range_data ok=%d,%d x=%f\n", (int)ok1, (int)ok2, lp[0]);
                return 1;
            }
        }
        rgrid_refine_options ro{1.0, 0.1, 0.4, 100, 1};
        std::array<double, 3> refined{};
        const rgrid_refine_summary rs = gf.RefineMatch(ro, {0.1, 0.05}, pe, thin, refined);
        const auto lim = gf.GetLimits();
        if (rs.termination != 0 || !(rs.final_cost <= rs.initial_cost) || std::fabs(refined[0]) > 0.06 || lim.num_x_cells != 200) {
            std::printf("FAIL refine term=%d cost %f -> %f x=%f nx=%d\n", rs.termination, rs.initial_cost, rs.final_cost, refined[0], lim.num_x_cells);
            return 1;
        }
    }
    std::printf("ADAPTER_OK n=%zu t=%.2f x=%.6f\n", mu.size(), t, mu[0]);
    return 0;
}
