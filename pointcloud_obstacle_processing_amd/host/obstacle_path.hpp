// obstacle_path.hpp — host-side mirror of the reference's hot path (od.cpp = minibot_cr18/src/
// obstacle_detection.cpp), same function names and argument meaning, on top of the C ABI
// (include/pcop.h).  No PCL, no ROS: a ROS build wraps these with the adapter in
// ros1/obstacle_detection_mi355x.cpp.
//
//   downsample_cloud            od.cpp:271-296     VoxelGrid + optional publish on "voxel_grid"
//   extract_euclidian_clusters  od.cpp:430-455     EuclideanClusterExtraction
//   build_initial_occupancy_grid_dataset  od.cpp:175-269   bounds filter + occupancy binning (the step before)
//   remove_statistical_outliers  od.cpp:316-340    StatisticalOutlierRemoval + optional publish (between the two)
//   segment_plane_and_extract_indices  od.cpp:342-428   RANSAC plane removal + optional publishes (after it)
//   ObstacleNode::cloud_cb      od.cpp:674-928     the part of the callback the north star names:
//                                                  PointCloud2 in -> voxel grid -> clusters ->
//                                                  "voxel_grid", "centroids", "euc_clusters" out
#pragma once

#include <string>
#include <vector>

#include "../../include/pcop.h"
#include "params_yaml.hpp"
#include "pcop_types.hpp"
#include "pointcloud2.hpp"

namespace pcop_host {

// One GPU context shared by the free functions below (the reference keeps its state in globals too,
// od.cpp:69-125).  Throws std::runtime_error when no GPU / libpcop is unusable: there is no CPU path.
pcop_ctx* default_context(size_t min_points);
void release_default_context();

// od.cpp:271-296.  Same signature with the stand-in types; leaf_size is the cubic VoxelGrid leaf.
void downsample_cloud(const PointCloudXYZ::Ptr& input_cloud, PointCloudXYZ& output_cloud, float leaf_size,
                      Publisher<PointCloud2>& pub, bool publish);

// od.cpp:430-455.  search_method is kept for call compatibility (the index is built on the GPU).
void extract_euclidian_clusters(PointCloudXYZ::Ptr& input_cloud, std::vector<PointIndices>& cluster_indices,
                                float cluster_tolerance, int min_cluster_size, int max_cluster_size,
                                KdTreeHandle::Ptr& search_method);

// od.cpp:175-269, same argument list (the reference reads block_size and dev_percent from globals, od.cpp:87-88;
// here they are the two trailing arguments).  grid_data gets 100 for a "hole" and 0 otherwise, point_count_array and
// row_point_averages the per-cell counts and the integer row means, output_cloud the points inside the box in input
// order.  The kept cloud also stays resident on the GPU of the default context (pcop_process_filtered).
void build_initial_occupancy_grid_dataset(int grid_size, int grid_height, int grid_width, char* grid_data, float x_min,
                                          float x_max, float y_min, float y_max, float z_min, float z_max,
                                          long long* point_count_array, long long* row_point_averages,
                                          const PointCloudXYZ& input_cloud, PointCloudXYZ& output_cloud,
                                          float block_size, float dev_percent);

// od.cpp:316-340.  Same signature with the stand-in types (meanK arrives as a float there too and is narrowed
// by setMeanK(int)); publishes on `pub` with frame_id "world" when asked (od.cpp:331-338).
void remove_statistical_outliers(PointCloudXYZ::Ptr& input_cloud, PointCloudXYZ& output_cloud, float meanK, float stdDevThres,
                                 Publisher<PointCloud2>& pub, bool publish);

// od.cpp:342-428.  Same argument list with the stand-in types: planar_cloud is the input and ends up as the points outside
// every plane found, indices_cloud as the points of the last plane removed, cloud_f as planar_cloud; coefficients holds the
// refined coefficients of the last plane.  inliers / outliers keep the call shape (the reference's ExtractIndices consumes
// them internally; here the selection happens on the GPU).  axis, dist_thresh, eps_angle as the node passes them (od.cpp:768,
// 110-111: the angle is an int that PCL reads as radians).  Returns false when a round found no plane (the reference prints
// "Couldn't estimate a planar model for the given dataset." and leaves the loop).
bool segment_plane_and_extract_indices(PointCloudXYZ::Ptr& planar_cloud, PointCloudXYZ::Ptr& indices_cloud, PointCloudXYZ::Ptr& cloud_f,
                                       ModelCoefficients::Ptr& coefficients, PointIndices::Ptr& inliers, PointIndices::Ptr& outliers,
                                       const float axis[3], float dist_thresh, int eps_angle, Publisher<PointCloud2>& pub1,
                                       Publisher<PointCloud2>& pub2, Publisher<PointCloud2>& pub3, bool publish_planar, bool publish_indices,
                                       bool publish_filtered);

// The node surface of the path: same topic names, queue sizes and parameter keys as od.cpp:930-1015.
class ObstacleNode {
public:
    static constexpr const char* kPointTopic = "/kinect2/qhd/points";   // od.cpp:80
    explicit ObstacleNode(const NodeParams& params, size_t max_points = 1u << 20, int device = 0);
    ~ObstacleNode();
    ObstacleNode(const ObstacleNode&) = delete;
    ObstacleNode& operator=(const ObstacleNode&) = delete;

    // The hot path of cloud_cb for one (already accumulated) cloud.  Returns a pcop_status.
    int cloud_cb(const PointCloud2& cloud_msg);
    const std::string& last_error() const { return error_; }

    NodeParams params;
    // false (default): the chain BASELINE.json's north star names, voxel grid -> clusters.  true: the node's own order,
    // build_initial_occupancy_grid_dataset (limits, block_size, dev_percent of `params`) -> voxel grid ->
    // remove_statistical_outliers (statistical_outlier_meanK / _stdDevThres) -> segment_plane_and_extract_indices
    // (plane_segment_dist_thres, plane_segment_angle, axis z) -> clusters (od.cpp:727-797), one pcop_process_node call; voxel_cloud is then the cloud the cluster indices
    // refer to (the inliers), occupancy_grid / _counts / _row_averages the grid outputs.
    bool node_order = false;
    bool remove_planes = true;                          // node order only: segment_plane_and_extract_indices between the outlier removal and the clustering (od.cpp:778-781)
    std::vector<signed char> occupancy_grid;            // occupancy_grid_data (od.cpp:725)
    std::vector<long long> occupancy_counts, occupancy_row_averages;
    pcop_node_stats stats{};
    Publisher<PointCloud2> downsample_publisher;        // "voxel_grid",   queue 1   (od.cpp:1004): in node order the cloud BEFORE the outlier removal, as downsample_cloud publishes it
    Publisher<PointCloud2> statistical_outlier_publisher;   // "statistical_outliers", queue 1    (od.cpp:1005) - node order
    Publisher<PointCloud2> indices_cloud_publisher;          // "indices_cloud",        queue 1000 (od.cpp:1006) - node order, once a plane was removed
    Publisher<PointCloud2> planar_cloud_publisher;           // "planar_cloud",         queue 1000 (od.cpp:1007) - node order
    Publisher<PointCloud2> filtered_cloud_publisher;         // "cloud_f",              queue 1000 (od.cpp:1008) - node order, once a plane was removed
    Publisher<PointIndicesArray> centroid_publisher;    // "centroids",    queue 1   (od.cpp:1009)
    Publisher<PointCloud2> euc_cluster_publisher;       // "euc_clusters", queue 5   (od.cpp:1010)
    std::vector<PointIndices> euc_cluster_indices;      // od.cpp:793
    PointCloudXYZ voxel_cloud;                          // statistical_outlier_input_cloud in od.cpp:740
    int passthrough = 0;

private:
    pcop_ctx* ctx_ = nullptr;
    std::string error_;
    bool node_voxel_grid_sent_ = false;
};

}  // namespace pcop_host
