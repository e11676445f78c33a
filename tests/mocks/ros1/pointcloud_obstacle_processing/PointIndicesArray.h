// tests/mocks/ros1: see README.md (what genmsg makes of msg/PointIndicesArray.msg: PointWithRad[] points)
#pragma once
#include <vector>
#include <pointcloud_obstacle_processing/PointWithRad.h>
namespace pointcloud_obstacle_processing { struct PointIndicesArray { std::vector<PointWithRad> points; }; }
