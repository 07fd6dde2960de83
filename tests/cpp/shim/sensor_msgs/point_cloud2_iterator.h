// syntax-only stand-in for ROS's sensor_msgs/point_cloud2_iterator.h; NOT ROS
#pragma once
#include "PointCloud2.h"
namespace sensor_msgs {
template <class T> class PointCloud2ConstIterator {
    const T *p_ = nullptr;
public:
    PointCloud2ConstIterator(const PointCloud2 &, const std::string &) {}
    PointCloud2ConstIterator end() const { return *this; }
    bool operator!=(const PointCloud2ConstIterator &o) const { return p_ != o.p_; }
    PointCloud2ConstIterator &operator++() { return *this; }
    const T &operator*() const { return *p_; }
};
}
