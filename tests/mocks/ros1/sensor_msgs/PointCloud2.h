// tests/mocks/ros1: see README.md (the fields of sensor_msgs/PointCloud2 and PointField)
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include <ros/ros.h>
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
namespace sensor_msgs {
struct PointField {
    enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0;
};
struct PointCloud2 {
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::vector<PointField> fields;
    uint8_t is_bigendian = 0;
    uint32_t point_step = 0, row_step = 0;
    std::vector<uint8_t> data;
    uint8_t is_dense = 0;
    typedef std::shared_ptr<PointCloud2> Ptr;
    typedef std::shared_ptr<PointCloud2 const> ConstPtr;
};
typedef std::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
}  // namespace sensor_msgs
