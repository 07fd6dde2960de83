// syntax-only stand-in for ROS's sensor_msgs/PointCloud2.h; NOT ROS
#pragma once
#include "LaserScan.h"
namespace sensor_msgs {
struct PointField { std::string name; unsigned offset; unsigned char datatype; unsigned count; };
struct PointCloud2 {
    ros_shim::Header header;
    unsigned height, width;
    std::vector<PointField> fields;
    bool is_bigendian;
    unsigned point_step, row_step;
    std::vector<unsigned char> data;
    bool is_dense;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}
