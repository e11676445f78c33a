// pcop_types.hpp — ROS-free, PCL-free stand-ins for the message and cloud types that cross the
// hot-path boundary of the reference node (minibot_cr18/src/obstacle_detection.cpp, "od.cpp").
// Field names and layouts follow the originals so that a ROS build can map them one to one
// (see ros1/obstacle_detection_mi355x.cpp and INTEGRATION.md):
//   sensor_msgs/PointCloud2, sensor_msgs/PointField      (subscribed at od.cpp:1001)
//   pcl::PointXYZ / pcl::PointCloud<pcl::PointXYZ>        (od.cpp:271-296, 430-455)
//   pcl::PointIndices                                     (od.cpp:793)
//   pointcloud_obstacle_processing/PointWithRad, PointIndicesArray   (msg/*.msg, od.cpp:1009)
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pcop_host {

struct Header {
    uint32_t seq = 0;
    double stamp = 0.0;
    std::string frame_id;
};

struct PointField {
    enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name;
    uint32_t offset = 0;
    uint8_t datatype = FLOAT32;
    uint32_t count = 1;
};

struct PointCloud2 {   // sensor_msgs/PointCloud2
    Header header;
    uint32_t height = 1, width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = true;
};

struct alignas(16) PointXYZ {   // pcl::PointXYZ in memory: 16 bytes, pad = 1.0f
    float x = 0.0f, y = 0.0f, z = 0.0f, pad = 1.0f;
};
static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is 16 bytes");

struct PointCloudXYZ {   // pcl::PointCloud<pcl::PointXYZ>
    typedef std::shared_ptr<PointCloudXYZ> Ptr;
    Header header;
    std::vector<PointXYZ> points;
    uint32_t width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
};

struct PointIndices {   // pcl::PointIndices
    typedef std::shared_ptr<PointIndices> Ptr;
    Header header;
    std::vector<int> indices;
};

struct ModelCoefficients {   // pcl::ModelCoefficients
    typedef std::shared_ptr<ModelCoefficients> Ptr;
    Header header;
    std::vector<float> values;
};

struct PointWithRad {   // msg/PointWithRad.msg:1-4
    float x = 0, y = 0, z = 0, r = 0;
};

struct PointIndicesArray {   // msg/PointIndicesArray.msg:1
    std::vector<PointWithRad> points;
};

// Stand-in for ros::Publisher: keeps the last message published on a topic.
template <class Msg>
struct Publisher {
    std::string topic;
    uint32_t queue_size = 1;
    uint64_t published = 0;
    Msg last;
    void publish(const Msg& m) { last = m; ++published; }
};

// The reference passes a pcl::search::KdTree handle into extract_euclidian_clusters (od.cpp:430-433,
// built at od.cpp:791-792).  Here the spatial index lives on the GPU and is rebuilt per call, so the
// handle only keeps the call shape; setInputCloud records what od.cpp:792 hands over.
struct KdTreeHandle {
    typedef std::shared_ptr<KdTreeHandle> Ptr;
    PointCloudXYZ::Ptr input;
    void setInputCloud(const PointCloudXYZ::Ptr& cloud) { input = cloud; }
};

}  // namespace pcop_host
