// C-callable hooks into the host-side mirror, so its CPU-only logic (yaml tree, PointCloud2
// field lookup, file format) can be exercised from the Python test-suite without a GPU.
#include <cstring>

#include "params_yaml.hpp"
#include "pointcloud2.hpp"

using namespace pcop_host;

extern "C" {

struct pcop_host_params {
    int accumulate_count, downsample_input_data, passthrough_filter_enable, publish_point_clouds, grid_opacity;
    float block_size, dev_percent, downsample_size;
    int statistical_outlier_meanK;
    float statistical_outlier_stdDevThres, plane_segment_dist_thres;
    int plane_segment_angle;
    float euc_cluster_tolerance;
    int euc_min_cluster_size, euc_max_cluster_size;
    float convex_hull_alpha, x_min, x_max, y_min, y_max, z_min, z_max;
};

int pcop_host_load_params(const char* yaml_path, pcop_host_params* out) {
    NodeParams p;
    std::string err;
    if (!load_node_params(yaml_path, &p, &err)) return -1;
    out->accumulate_count = p.accumulate_count; out->downsample_input_data = p.downsample_input_data;
    out->passthrough_filter_enable = p.passthrough_filter_enable; out->publish_point_clouds = p.publish_point_clouds;
    out->grid_opacity = p.grid_opacity; out->block_size = p.block_size; out->dev_percent = p.dev_percent;
    out->downsample_size = p.downsample_size; out->statistical_outlier_meanK = p.statistical_outlier_meanK;
    out->statistical_outlier_stdDevThres = p.statistical_outlier_stdDevThres;
    out->plane_segment_dist_thres = p.plane_segment_dist_thres; out->plane_segment_angle = p.plane_segment_angle;
    out->euc_cluster_tolerance = p.euc_cluster_tolerance; out->euc_min_cluster_size = p.euc_min_cluster_size;
    out->euc_max_cluster_size = p.euc_max_cluster_size; out->convex_hull_alpha = p.convex_hull_alpha;
    out->x_min = p.x_min; out->x_max = p.x_max; out->y_min = p.y_min; out->y_max = p.y_max; out->z_min = p.z_min; out->z_max = p.z_max;
    return 0;
}

// Round trip of the harness file format + field lookup + host-side conversion:
// reads `path`, returns n points and copies up to cap xyz triples.
long pcop_host_read_cloud_xyz(const char* path, float* xyz, long cap, int* is_dense, unsigned* off) {
    PointCloud2 msg;
    std::string err;
    if (!read_pointcloud2_file(path, &msg, &err)) return -1;
    PointCloudXYZ cloud;
    if (!from_pointcloud2(msg, &cloud, &err)) return -2;
    if (off && !find_xyz_fields(msg, &off[0], &off[1], &off[2], &err)) return -2;
    if (is_dense) *is_dense = msg.is_dense ? 1 : 0;
    const long n = (long)cloud.points.size();
    for (long i = 0; i < n && i < cap; ++i) { xyz[3 * i] = cloud.points[i].x; xyz[3 * i + 1] = cloud.points[i].y; xyz[3 * i + 2] = cloud.points[i].z; }
    return n;
}

}  // extern "C"
