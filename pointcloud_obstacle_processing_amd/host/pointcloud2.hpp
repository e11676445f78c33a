// pointcloud2.hpp — sensor_msgs/PointCloud2 <-> XYZ, the conversions of od.cpp:688-689
// (pcl_conversions::toPCL + pcl::fromPCLPointCloud2) and od.cpp:288-295 (pcl::toROSMsg).
#pragma once

#include <string>

#include "pcop_types.hpp"

namespace pcop_host {

// Byte offsets of the FLOAT32 fields "x","y","z" (pcl::fromPCLPointCloud2 matches fields by name
// and datatype).  Returns false when one is missing or not FLOAT32.
bool find_xyz_fields(const PointCloud2& msg, uint32_t* off_x, uint32_t* off_y, uint32_t* off_z, std::string* error);

// Host-side conversion (used by the stage API wrappers; the frame API converts on the GPU).
bool from_pointcloud2(const PointCloud2& msg, PointCloudXYZ* cloud, std::string* error);

// pcl::toROSMsg for PointXYZ clouds: fields x,y,z FLOAT32 at 0,4,8, point_step 16, height 1.
void to_pointcloud2(const PointCloudXYZ& cloud, PointCloud2* msg);

// A small self-describing file format for the ROS-free harness (text header + raw data blob).
bool write_pointcloud2_file(const std::string& path, const PointCloud2& msg, std::string* error);
bool read_pointcloud2_file(const std::string& path, PointCloud2* msg, std::string* error);

}  // namespace pcop_host
