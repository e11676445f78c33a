// params_yaml.hpp — the node's parameters, read from the same yaml tree and with the same
// defaults as od.cpp:940-975 (file: minibot_cr18/params.yaml, loaded by launch/main.launch:2).
#pragma once

#include <map>
#include <string>

namespace pcop_host {

// Flat view of a yaml file made of nested maps of scalars: "a/b/c" -> "value".
// Enough for the reference's parameter file; not a general YAML parser.
std::map<std::string, std::string> load_yaml_scalars(const std::string& path, std::string* error);

struct NodeParams {
    // private namespace  /pointcloud_obstacle_processing/obstacle_detection/*  (nh("~"), od.cpp:934)
    int accumulate_count = 2;                    // od.cpp:940
    bool downsample_input_data = true;           // od.cpp:943 (yaml misspells it "downsame_input_data": never matched)
    bool passthrough_filter_enable = true;       // od.cpp:944
    bool publish_point_clouds = true;            // od.cpp:945
    int grid_opacity = 0;                        // od.cpp:946
    float block_size = 0.15f;                    // od.cpp:955
    float dev_percent = 0.5f;                    // od.cpp:956
    float downsample_size = 0.015f;              // od.cpp:964   -> pcop_params.leaf
    int statistical_outlier_meanK = 15;          // od.cpp:966
    float statistical_outlier_stdDevThres = 1.0f;// od.cpp:967
    float plane_segment_dist_thres = 0.040f;     // od.cpp:969
    int plane_segment_angle = 20;                // od.cpp:970
    float euc_cluster_tolerance = 0.4f;          // od.cpp:972   -> pcop_params.tol
    int euc_min_cluster_size = 5;                // od.cpp:973   -> pcop_params.min_cluster
    int euc_max_cluster_size = 20000;            // od.cpp:974   -> pcop_params.max_cluster
    float convex_hull_alpha = 180.0f;            // od.cpp:975
    // namespace level  /pointcloud_obstacle_processing/*  (nh_pub, od.cpp:935, 948-953)
    float x_min = -1.0f, x_max = 1.0f, y_min = -0.5f, y_max = 0.6f, z_min = 0.0f, z_max = -0.5f;
};

// nh.param<T>(key, var, default): yaml value if present (doubles narrowed to float like
// XmlRpc -> float does), else the default in code.  Returns false only when the file is unreadable.
bool load_node_params(const std::string& yaml_path, NodeParams* out, std::string* error);

}  // namespace pcop_host
