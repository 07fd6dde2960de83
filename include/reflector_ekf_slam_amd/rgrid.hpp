// rgrid.hpp -- header-only C++ RAII wrapper over include/rgrid.h (grid-mapper front-end): the calls
// mapping::MapBuilder makes to sensor::VoxelFilter / AdaptiveVoxelFilter, RealTimeCorrelativeScanMatcher2D::Match and
// ProbabilityGridRangeDataInserter2D::Insert (src/mapping/map_builder.cc:30-31,43,73), with plain std types.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rgrid.h"

namespace rekfpp {

struct GridError : std::runtime_error {
    int code;
    GridError(int c, const std::string &where) : std::runtime_error(where + ": " + rgrid_strerror(c)), code(c) {}
};

class GridFrontEnd {
public:
    using Cloud = std::vector<float>;                 // x0, y0, x1, y1, ... (sensor::PointCloud = std::vector<Eigen::Vector2f>)

    explicit GridFrontEnd(int max_points = 8192, int max_cells = 4096 * 4096, int max_candidates = 1 << 20, int device = 0)
    {
        const int rc = rgrid_create(max_points, max_cells, max_candidates, device, &h_);
        if (rc != RGRID_OK) throw GridError(rc, "rgrid_create");
    }
    ~GridFrontEnd() { rgrid_destroy(h_); }
    GridFrontEnd(const GridFrontEnd &) = delete;
    GridFrontEnd &operator=(const GridFrontEnd &) = delete;

    // sensor::VoxelFilter(size).Filter(cloud)  (voxel_filter.cc:81-95)
    Cloud VoxelFilter(const Cloud &cloud, float size)
    {
        Cloud out(cloud.size());
        int m = 0;
        chk(rgrid_voxel_filter(h_, cloud.data(), (int)(cloud.size() / 2), size, out.data(), (int)(out.size() / 2), &m), "VoxelFilter");
        out.resize(2 * (size_t)m);
        return out;
    }
    // sensor::AdaptiveVoxelFilter(options).Filter(cloud)  (voxel_filter.cc:116-120)
    Cloud AdaptiveVoxelFilter(const Cloud &cloud, double max_length, double min_num_points, double max_range)
    {
        Cloud out(cloud.size());
        int m = 0;
        chk(rgrid_adaptive_voxel_filter(h_, cloud.data(), (int)(cloud.size() / 2), max_length, min_num_points, max_range,
                                        out.data(), (int)(out.size() / 2), &m), "AdaptiveVoxelFilter");
        out.resize(2 * (size_t)m);
        return out;
    }
    // the ProbabilityGrid: grid.correspondence_cost_cells(), grid.limits()  (grid_2d.h:83-106, map_limits.h:24-45)
    void SetGrid(const std::vector<uint16_t> &cells, int num_x_cells, int num_y_cells, double resolution, double max_x, double max_y)
    {
        if ((size_t)num_x_cells * num_y_cells != cells.size()) throw GridError(RGRID_ERR_INVALID, "SetGrid");
        chk(rgrid_set_grid(h_, cells.data(), num_x_cells, num_y_cells, resolution, max_x, max_y), "SetGrid");
        ncells_ = cells.size();
    }
    std::vector<uint16_t> GetGrid()
    {
        std::vector<uint16_t> out(ncells_);
        chk(rgrid_get_grid(h_, out.data(), (long)out.size()), "GetGrid");
        return out;
    }
    // MapLimits of the resident grid
    struct Limits { int num_x_cells, num_y_cells; double resolution, max_x, max_y; };
    Limits GetLimits()
    {
        Limits l{};
        chk(rgrid_get_limits(h_, &l.num_x_cells, &l.num_y_cells, &l.resolution, &l.max_x, &l.max_y), "GetLimits");
        return l;
    }
    // GrowAsNeeded  (probability_grid_range_data_inserter_2d.cc:20-38 -> Grid2D::GrowLimits, grid_2d.cc:59-99)
    void GrowAsNeeded(const std::array<float, 2> &origin, const Cloud &returns, const Cloud &misses)
    {
        chk(rgrid_grow_as_needed(h_, origin.data(), returns.data(), (int)(returns.size() / 2), misses.data(), (int)(misses.size() / 2)),
            "GrowAsNeeded");
        const Limits l = GetLimits();
        ncells_ = (size_t)l.num_x_cells * l.num_y_cells;
    }
    // ProbabilityGridRangeDataInserter2D::Insert  (probability_grid_range_data_inserter_2d.cc:103-114); grows the grid
    // first, as the reference's CastRays does (:45)
    void Insert(const std::array<float, 2> &origin, const Cloud &returns, const Cloud &misses, float hit_probability = 0.55f,
                float miss_probability = 0.49f, bool insert_free_space = true)
    {
        GrowAsNeeded(origin, returns, misses);
        chk(rgrid_insert(h_, origin.data(), returns.data(), (int)(returns.size() / 2), misses.data(), (int)(misses.size() / 2),
                         hit_probability, miss_probability, insert_free_space ? 1 : 0), "Insert");
    }
    // RealTimeCorrelativeScanMatcher2D::Match  (real_time_correlative_scan_matcher_2d.cc:84-118): returns the score
    double Match(const rgrid_match_options &opt, const std::array<double, 3> &initial_pose, const Cloud &cloud,
                 std::array<double, 3> &pose_estimate)
    {
        double score = 0;
        chk(rgrid_match(h_, &opt, initial_pose.data(), cloud.data(), (int)(cloud.size() / 2), pose_estimate.data(), &score,
                        nullptr, nullptr), "Match");
        return score;
    }
    // CeresScanMatcher2D::Match  (ceres_scan_matcher_2d.cc:26-62): returns the summary, pose in `pose_estimate`
    rgrid_refine_summary RefineMatch(const rgrid_refine_options &opt, const std::array<double, 2> &target_translation,
                                     const std::array<double, 3> &initial_pose, const Cloud &cloud, std::array<double, 3> &pose_estimate)
    {
        rgrid_refine_summary s{};
        chk(rgrid_refine_match(h_, &opt, target_translation.data(), initial_pose.data(), cloud.data(), (int)(cloud.size() / 2),
                               pose_estimate.data(), &s), "RefineMatch");
        return s;
    }
    // ProbabilityGrid::DrawToSubmapTexture  (probability_grid.cc:86-131): (value, alpha) bytes of the known-cells window
    struct Texture { std::vector<uint8_t> cells; int offset_x, offset_y, width, height; double slice_max_x, slice_max_y; };
    Texture DrawTexture()
    {
        Texture t;
        t.cells.resize(2 * ncells_);
        int box[4];
        double sm[2];
        chk(rgrid_draw_texture(h_, t.cells.data(), (long)t.cells.size(), box, sm), "DrawTexture");
        t.offset_x = box[0]; t.offset_y = box[1]; t.width = box[2]; t.height = box[3];
        t.slice_max_x = sm[0]; t.slice_max_y = sm[1];
        t.cells.resize(2 * (size_t)t.width * t.height);
        return t;
    }
    // mapping::MapBuilder::AddRangeData  (map_builder.cc:57-108) in one call; false where the reference returns nullptr
    bool AddRangeData(const rgrid_map_builder_options &opt, const std::array<float, 2> &origin, const Cloud &returns, const Cloud &misses,
                      const std::array<double, 3> &ekf_pose, std::array<double, 3> &local_pose, Cloud *returns_in_local = nullptr)
    {
        int status = 0;
        if (returns_in_local) returns_in_local->assign(returns.size(), 0.f);
        chk(rgrid_add_range_data(h_, &opt, origin.data(), returns.data(), (int)(returns.size() / 2), misses.data(), (int)(misses.size() / 2),
                                 ekf_pose.data(), local_pose.data(), returns_in_local ? returns_in_local->data() : nullptr, &status),
            "AddRangeData");
        if (status == RGRID_SCAN_INSERTED) { const Limits l = GetLimits(); ncells_ = (size_t)l.num_x_cells * l.num_y_cells; }
        return status == RGRID_SCAN_INSERTED;
    }
    rgrid_t *handle() { return h_; }

private:
    static void chk(int rc, const char *where) { if (rc != RGRID_OK) throw GridError(rc, where); }
    rgrid_t *h_ = nullptr;
    size_t ncells_ = 0;
};

}  // namespace rekfpp
