// tests/mocks/ros1: see README.md (what genmsg makes of msg/PointWithRad.msg: float32 x, y, z, r)
#pragma once
namespace pointcloud_obstacle_processing { struct PointWithRad { float x = 0, y = 0, z = 0, r = 0; }; }
