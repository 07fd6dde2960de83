// Translation unit of tests/test_adapter_syntax_cpu.py: our Eigen/ROS-typed adapters against the REAL reference headers
// (-I /root/reference/include) with syntax-only stand-ins for Eigen / sensor_msgs / PCL (tests/cpp/shim).  `override` on every
// method + !is_abstract = each adapter implements its reference interface with exactly the reference's signatures.
#include <type_traits>

#include "ekf_slam_adapter.hpp"
#include "detect_adapter.hpp"

static_assert(std::is_base_of<ekf::ReflectorEKFSLAMInterface, ekf::ReflectorEKFSLAMHip>::value, "derives from the reference's interface");
static_assert(!std::is_abstract<ekf::ReflectorEKFSLAMHip>::value, "every pure virtual of ekf_slam_interface.h:50-67 is implemented");
static_assert(std::is_base_of<reflector_detect::ReflectorDetectInterface, reflector_detect::LaserReflectorDetectHip>::value, "");
static_assert(std::is_base_of<reflector_detect::ReflectorDetectInterface, reflector_detect::PointCloudReflectorDetectHip>::value, "");
static_assert(!std::is_abstract<reflector_detect::LaserReflectorDetectHip>::value && !std::is_abstract<reflector_detect::PointCloudReflectorDetectHip>::value, "");

// the calls ros_node.cc makes (src/ros_node.cc:440,455-470,514-515,577,627-660), spelled against the adapters
void node_calls(const ekf::EKFOptions &options, const sensor::OdometryData &odo, const sensor::Observation &obs,
                const sensor_msgs::LaserScanConstPtr &scan, const sensor_msgs::PointCloud2ConstPtr &cloud)
{
    std::unique_ptr<ekf::ReflectorEKFSLAMInterface> slam = common::make_unique<ekf::ReflectorEKFSLAMHip>(options, 1024, 0);
    slam->HandleOdometryMessage(odo);
    slam->HandleObservationMessage(obs);
    ekf::State st = slam->GetState();
    ekf::State ps = slam->PredictState(st.time + 0.1);
    (void)slam->GetLatestTime();
    sensor::Map m = slam->GetGlobalMap();
    (void)ps; (void)m;
    reflector_detect::ReflectorDetectOptions lo{160., 0.18, 0.06, 0.3f, 10.f};
    std::unique_ptr<reflector_detect::ReflectorDetectInterface> det = common::make_unique<reflector_detect::LaserReflectorDetectHip>(lo);
    det->SetSensorToBaseLinkTransform(transform::Rigid3d::Identity());
    det->HandleOdometryData(odo);
    sensor::Observation o2 = det->HandleLaserScan(scan);
    sensor::RangeData rd = det->GetRangeData();
    (void)rd;
    reflector_detect::PointCloudOptions po{160.};
    std::unique_ptr<reflector_detect::ReflectorDetectInterface> det3 = common::make_unique<reflector_detect::PointCloudReflectorDetectHip>(po);
    sensor::Observation o3 = det3->HandlePointCloud(cloud);
    (void)o2; (void)o3;
}
