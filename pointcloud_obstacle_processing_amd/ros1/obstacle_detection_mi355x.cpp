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
    const int rc = pcop_process_pointcloud2(g_ctx, cloud_msg->data.data(), cloud_msg->width, cloud_msg->height,
                                            cloud_msg->point_step, cloud_msg->row_step, off[0], off[1], off[2],
                                            cloud_msg->is_bigendian, cloud_msg->is_dense, &r);
    if (rc != PCOP_OK) { ROS_ERROR("pcop: %s (%s)", pcop_status_string(rc), pcop_last_error(g_ctx)); return; }
    if (publish_point_clouds) downsample_publisher.publish(xyz_cloud_msg(r.voxel_xyz, r.num_voxels, "world"));
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
    nh.param("publish_point_clouds", publish_point_clouds, true);    // od.cpp:945
    nh.param("downsample_size", downsample_leaf_size, (float)0.015); // od.cpp:964
    nh.param("euc_cluster_tolerance", euc_cluster_tolerance, (float)0.4);
    nh.param("euc_min_cluster_size", euc_min_cluster_size, 5);
    nh.param("euc_max_cluster_size", euc_max_cluster_size, 20000);
    int max_points = 1 << 22, device = 0;
    nh.param("pcop_max_points", max_points, max_points);
    nh.param("pcop_device", device, device);
    if (pcop_create(&g_ctx, device, (size_t)max_points) != PCOP_OK) { ROS_FATAL("no MI355X context"); return 1; }
    ros::Subscriber sub = nh.subscribe(point_topic, 1, cloud_cb);                                           // od.cpp:1001
    downsample_publisher = nh.advertise<sensor_msgs::PointCloud2>("voxel_grid", 1);                         // od.cpp:1004
    centroid_publisher = nh.advertise<pointcloud_obstacle_processing::PointIndicesArray>("centroids", 1);   // od.cpp:1009
    euc_cluster_publisher = nh.advertise<sensor_msgs::PointCloud2>("euc_clusters", 5);                      // od.cpp:1010
    ros::spin();
    pcop_destroy(g_ctx);
    return 0;
}
