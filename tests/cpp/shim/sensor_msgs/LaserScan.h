// syntax-only stand-in for ROS's sensor_msgs/LaserScan.h (field names and types of the message definition); NOT ROS
#pragma once
#include <memory>
#include <string>
#include <vector>
namespace ros_shim { struct Time { double toSec() const { return 0.0; } }; struct Header { unsigned seq; Time stamp; std::string frame_id; }; }
namespace sensor_msgs {
struct LaserScan {
    ros_shim::Header header;
    float angle_min, angle_max, angle_increment, time_increment, scan_time, range_min, range_max;
    std::vector<float> ranges, intensities;
};
typedef std::shared_ptr<const LaserScan> LaserScanConstPtr;
}
