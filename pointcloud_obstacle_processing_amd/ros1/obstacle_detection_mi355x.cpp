// obstacle_detection_mi355x.cpp — ROS-1 adapter: the reference node's surface (od.cpp:930-1015) on
// top of the C ABI in include/pcop.h.  NOT built in this repository's CI (no catkin / roscpp in the
// build image); it compiles inside the reference's catkin workspace with
//     target_link_libraries(pointcloud_obstacle_detection ${catkin_LIBRARIES} pcop)
// and needs the generated message headers of msg/PointWithRad.msg and msg/PointIndicesArray.msg.
// PCL is not linked: the three PCL call clusters of the hot path are replaced.
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include <pointcloud_obstacle_processing/PointIndicesArray.h>
#include <pointcloud_obstacle_processing/PointWithRad.h>

#include <cstring>
#include <vector>

#include "pcop.h"

namespace {

pcop_ctx* g_ctx = nullptr;
ros::Publisher downsample_publisher, centroid_publisher, euc_cluster_publisher;   // od.cpp:69-76
ros::Publisher statistical_outlier_publisher, indices_cloud_publisher, planar_cloud_publisher, filtered_cloud_publisher;
bool node_order = true;                                                           // the node's own stage order in one fused call
float pt_lower_lim_x, pt_upper_lim_x, pt_lower_lim_y, pt_upper_lim_y, pt_lower_lim_z, pt_upper_lim_z;   // od.cpp:98-103
float block_size, dev_percent;                                                    // od.cpp:87-88
int statistical_outlier_meanK, plane_segment_angle;                               // od.cpp:107,111
float statistical_outlier_stdDevThres, plane_segment_dist_thres;                  // od.cpp:108,110
const char* point_topic = "/kinect2/qhd/points";                                  // od.cpp:80
float downsample_leaf_size, euc_cluster_tolerance;                                // od.cpp:105,113
int euc_min_cluster_size, euc_max_cluster_size;                                   // od.cpp:114-115
bool publish_point_clouds;                                                        // od.cpp:118

bool xyz_offsets(const sensor_msgs::PointCloud2& m, uint32_t off[3]) {
    bool have[3] = {false, false, false};
    for (const auto& f : m.fields) {
        if (f.datatype != sensor_msgs::PointField::FLOAT32) continue;
        const int a = f.name == "x" ? 0 : f.name == "y" ? 1 : f.name == "z" ? 2 : -1;
        if (a >= 0) { off[a] = f.offset; have[a] = true; }
    }
    return have[0] && have[1] && have[2];
}

// xyz16: n records of 16 bytes {x, y, z, 1.0f} as the GPU wrote them = the data blob pcl::toROSMsg produces
sensor_msgs::PointCloud2 xyz_cloud_msg(const float* xyz16, size_t n, const char* frame) {
    sensor_msgs::PointCloud2 m;
    m.header.frame_id = frame;
    m.header.stamp = ros::Time::now();
    m.height = 1; m.width = n; m.is_bigendian = false; m.is_dense = true;
    m.point_step = 16; m.row_step = 16 * n;
    const char* names[3] = {"x", "y", "z"};
    for (int a = 0; a < 3; ++a) {
        sensor_msgs::PointField f;
        f.name = names[a]; f.offset = 4 * a; f.datatype = sensor_msgs::PointField::FLOAT32; f.count = 1;
        m.fields.push_back(f);
    }
    m.data.resize(16 * n);
    if (n) std::memcpy(&m.data[0], xyz16, 16 * n);
    return m;
}

// The hot-path part of cloud_cb (od.cpp:674-928): conversion (688-689), VoxelGrid (740-741),
// kd-tree + Euclidean clusters (791-797), and the three topics of the path.
void cloud_cb(const sensor_msgs::PointCloud2ConstPtr& cloud_msg) {
    uint32_t off[3];
    if (!xyz_offsets(*cloud_msg, off)) { ROS_ERROR("cloud without FLOAT32 x/y/z"); return; }
    pcop_params p{};
    p.leaf = downsample_leaf_size; p.tol = euc_cluster_tolerance;
    p.min_cluster = euc_min_cluster_size; p.max_cluster = euc_max_cluster_size; p.publish_clouds = 1;
    pcop_set_params(g_ctx, &p);
    pcop_set_cloud_layout(g_ctx, PCOP_LAYOUT_AOS16);
    pcop_result r{};
    int rc;
    if (node_order) {   // od.cpp:727-797 in the node's order, one call; the clouds of the topics between the steps come out of it
        // The message goes to the GPU as it is (pcop_process_node_pointcloud2 unpacks x / y / z by field offset there: od.cpp:688-689);
        // the library checks its geometry, the size of the blob is checked here, where the vector is.
        const size_t n = (size_t)cloud_msg->width * cloud_msg->height;
        if (n && cloud_msg->data.size() < (size_t)cloud_msg->height * cloud_msg->row_step) {
            ROS_ERROR("cloud message truncated: %zu bytes for %u rows of %u", cloud_msg->data.size(), cloud_msg->height, cloud_msg->row_step);
            return;
        }
        pcop_occ_params occ{};
        occ.x_min = pt_lower_lim_x; occ.x_max = pt_upper_lim_x; occ.y_min = pt_lower_lim_y; occ.y_max = pt_upper_lim_y;
        occ.z_min = pt_lower_lim_z; occ.z_max = pt_upper_lim_z; occ.block_size = block_size; occ.dev_percent = dev_percent;
        pcop_occupancy_grid_dims(occ.x_min, occ.x_max, occ.y_min, occ.y_max, occ.block_size, &occ.grid_width, &occ.grid_height);   // od.cpp:958-959
        pcop_plane_params pp{};
        pp.dist_thresh = plane_segment_dist_thres; pp.axis[2] = 1.0f; pp.eps_angle = (double)plane_segment_angle;                  // od.cpp:768, 778-781
        const size_t cells = (size_t)(occ.grid_width > 0 ? occ.grid_width : 0) * (size_t)(occ.grid_height > 0 ? occ.grid_height : 0);
        // scratch kept across callbacks (the queue depth is 1 and the callback is single-threaded: od.cpp:1001, 1014)
        static std::vector<int64_t> counts, rows;
        static std::vector<int8_t> grid;
        static std::vector<float> vg, so, pl, ix, cf;
        if (counts.size() < cells + 1) { counts.resize(cells + 1); grid.resize(cells + 1); }
        if (rows.size() < (size_t)(occ.grid_height > 0 ? occ.grid_height : 0) + 1) rows.resize((size_t)(occ.grid_height > 0 ? occ.grid_height : 0) + 1);
        for (std::vector<float>* v : {&vg, &so, &pl, &ix, &cf}) if (v->size() < 4 * n + 4) v->resize(4 * n + 4);
        pcop_node_clouds cl{};
        cl.voxel_grid = vg.data(); cl.statistical_outliers = so.data(); cl.planar_cloud = pl.data(); cl.indices_cloud = ix.data(); cl.cloud_f = cf.data();
        rc = pcop_process_node_pointcloud2(g_ctx, cloud_msg->data.data(), cloud_msg->width, cloud_msg->height, cloud_msg->point_step, cloud_msg->row_step,
                                           off[0], off[1], off[2], cloud_msg->is_bigendian, cloud_msg->is_dense, &occ, counts.data(), rows.data(),
                                           grid.data(), statistical_outlier_meanK, statistical_outlier_stdDevThres, &pp, &r, nullptr, &cl);
        if (rc != PCOP_OK) { ROS_ERROR("pcop: %s (%s)", pcop_status_string(rc), pcop_last_error(g_ctx)); return; }
        if (publish_point_clouds) {   // od.cpp:288-295, 331-338, 401-426 (a count of -1: the call did not produce that cloud)
            if (cl.n_voxel_grid >= 0) downsample_publisher.publish(xyz_cloud_msg(cl.voxel_grid, (size_t)cl.n_voxel_grid, "world"));
            if (cl.n_statistical_outliers >= 0) statistical_outlier_publisher.publish(xyz_cloud_msg(cl.statistical_outliers, (size_t)cl.n_statistical_outliers, "world"));
            if (cl.n_planar_cloud >= 0) planar_cloud_publisher.publish(xyz_cloud_msg(cl.planar_cloud, (size_t)cl.n_planar_cloud, "world"));
            if (cl.n_indices_cloud >= 0) indices_cloud_publisher.publish(xyz_cloud_msg(cl.indices_cloud, (size_t)cl.n_indices_cloud, "world"));
            if (cl.n_cloud_f >= 0) filtered_cloud_publisher.publish(xyz_cloud_msg(cl.cloud_f, (size_t)cl.n_cloud_f, "world"));
        }
    } else {
        if ((size_t)cloud_msg->width * cloud_msg->height && cloud_msg->data.size() < (size_t)cloud_msg->height * cloud_msg->row_step) {
            ROS_ERROR("cloud message truncated: %zu bytes for %u rows of %u", cloud_msg->data.size(), cloud_msg->height, cloud_msg->row_step);
            return;
        }
        rc = pcop_process_pointcloud2(g_ctx, cloud_msg->data.data(), cloud_msg->width, cloud_msg->height,
                                      cloud_msg->point_step, cloud_msg->row_step, off[0], off[1], off[2],
                                      cloud_msg->is_bigendian, cloud_msg->is_dense, &r);
        if (rc != PCOP_OK) { ROS_ERROR("pcop: %s (%s)", pcop_status_string(rc), pcop_last_error(g_ctx)); return; }
        if (publish_point_clouds) downsample_publisher.publish(xyz_cloud_msg(r.voxel_xyz, r.num_voxels, "world"));
    }
    pointcloud_obstacle_processing::PointIndicesArray centroids;
    std::vector<float> xyzr(4 * (size_t)r.num_clusters + 4);
    if (r.num_clusters) pcop_cluster_summaries(g_ctx, xyzr.data(), r.num_clusters);
    for (int c = 0; c < r.num_clusters; ++c) {
        pointcloud_obstacle_processing::PointWithRad q;
        q.x = xyzr[4 * c]; q.y = xyzr[4 * c + 1]; q.z = xyzr[4 * c + 2]; q.r = xyzr[4 * c + 3];
        centroids.points.push_back(q);
    }
    centroid_publisher.publish(centroids);
    const float* clu = nullptr;
    int32_t nclu = 0;
    if (pcop_cluster_cloud(g_ctx, 0, &clu, &nclu) == PCOP_OK) euc_cluster_publisher.publish(xyz_cloud_msg(clu, (size_t)nclu, "/world"));
}

}  // namespace

int main(int argc, char** argv) {
    ros::init(argc, argv, "pointcloud_obstacle_detection");          // od.cpp:933
    ros::NodeHandle nh("~");
    ros::NodeHandle nh_pub;                                          // the reference's namespace-level handle for the field limits (od.cpp:935)
    nh.param("publish_point_clouds", publish_point_clouds, true);    // od.cpp:945
    nh.param("downsample_size", downsample_leaf_size, (float)0.015); // od.cpp:964
    nh.param("euc_cluster_tolerance", euc_cluster_tolerance, (float)0.4);
    nh.param("euc_min_cluster_size", euc_min_cluster_size, 5);
    nh.param("euc_max_cluster_size", euc_max_cluster_size, 20000);
    nh_pub.param("x_min", pt_lower_lim_x, (float)-1.0); nh_pub.param("x_max", pt_upper_lim_x, (float)1.0);       // od.cpp:948-953
    nh_pub.param("y_min", pt_lower_lim_y, (float)-0.5); nh_pub.param("y_max", pt_upper_lim_y, (float)0.6);
    nh_pub.param("z_min", pt_lower_lim_z, (float)0); nh_pub.param("z_max", pt_upper_lim_z, (float)-0.5);
    nh.param("block_size", block_size, (float)0.15);                 // od.cpp:955-956
    nh.param("dev_percent", dev_percent, (float)0.5);
    nh.param("statistical_outlier_meanK", statistical_outlier_meanK, 15);                    // od.cpp:966-967
    nh.param("statistical_outlier_stdDevThres", statistical_outlier_stdDevThres, (float)1.0);
    nh.param("plane_segment_dist_thres", plane_segment_dist_thres, (float)0.040);            // od.cpp:969-970
    nh.param("plane_segment_angle", plane_segment_angle, 20);
    nh.param("pcop_node_order", node_order, true);
    int max_points = 1 << 22, device = 0;
    nh.param("pcop_max_points", max_points, max_points);
    nh.param("pcop_device", device, device);
    if (pcop_create(&g_ctx, device, (size_t)max_points) != PCOP_OK) { ROS_FATAL("no MI355X context"); return 1; }
    ros::Subscriber sub = nh.subscribe(point_topic, 1, cloud_cb);                                           // od.cpp:1001
    downsample_publisher = nh.advertise<sensor_msgs::PointCloud2>("voxel_grid", 1);                         // od.cpp:1004
    statistical_outlier_publisher = nh.advertise<sensor_msgs::PointCloud2>("statistical_outliers", 1);      // od.cpp:1005
    indices_cloud_publisher = nh.advertise<sensor_msgs::PointCloud2>("indices_cloud", 1000);                // od.cpp:1006
    planar_cloud_publisher = nh.advertise<sensor_msgs::PointCloud2>("planar_cloud", 1000);                  // od.cpp:1007
    filtered_cloud_publisher = nh.advertise<sensor_msgs::PointCloud2>("cloud_f", 1000);                     // od.cpp:1008
    centroid_publisher = nh.advertise<pointcloud_obstacle_processing::PointIndicesArray>("centroids", 1);   // od.cpp:1009
    euc_cluster_publisher = nh.advertise<sensor_msgs::PointCloud2>("euc_clusters", 5);                      // od.cpp:1010
    ros::spin();
    pcop_destroy(g_ctx);
    return 0;
}
