#include "obstacle_path.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace pcop_host {

namespace {
pcop_ctx* g_ctx = nullptr;
size_t g_cap = 0;

void check(pcop_ctx* c, int rc, const char* what) {
    if (rc == PCOP_OK) return;
    std::string msg = std::string(what) + ": " + pcop_status_string(rc);
    if (c) msg += std::string(" (") + pcop_last_error(c) + ")";
    throw std::runtime_error(msg);
}
}  // namespace

pcop_ctx* default_context(size_t min_points) {
    if (g_ctx && g_cap >= min_points) return g_ctx;
    release_default_context();
    size_t cap = 1u << 20;
    while (cap < min_points) cap <<= 1;
    pcop_ctx* c = nullptr;
    check(nullptr, pcop_create(&c, 0, cap), "pcop_create");
    g_ctx = c; g_cap = cap;
    return g_ctx;
}

void release_default_context() {
    if (g_ctx) pcop_destroy(g_ctx);
    g_ctx = nullptr; g_cap = 0;
}

void downsample_cloud(const PointCloudXYZ::Ptr& input_cloud, PointCloudXYZ& output_cloud, float leaf_size,
                      Publisher<PointCloud2>& pub, bool publish) {
    const size_t n = input_cloud->points.size();
    pcop_ctx* c = default_context(n);
    pcop_params p{};
    pcop_get_params(c, &p);
    p.leaf = leaf_size;                                   // sor.setLeafSize(leaf, leaf, leaf)   od.cpp:284
    check(c, pcop_set_params(c, &p), "pcop_set_params");
    output_cloud.points.resize(n);
    size_t m = 0;
    int passthrough = 0;
    check(c, pcop_voxel_downsample(c, n ? &input_cloud->points[0].x : nullptr, PCOP_LAYOUT_AOS16, n,
                                   input_cloud->is_dense ? 1 : 0, n ? &output_cloud.points[0].x : nullptr, &m, &passthrough),
          "pcop_voxel_downsample");                       // sor.filter(output_cloud)               od.cpp:285
    output_cloud.points.resize(m);
    output_cloud.header = input_cloud->header;            // Filter::filter copies the header
    output_cloud.width = (uint32_t)m; output_cloud.height = 1;
    output_cloud.is_dense = passthrough ? input_cloud->is_dense : true;
    if (publish) {                                        // od.cpp:288-295
        PointCloud2 publish_cloud;
        to_pointcloud2(output_cloud, &publish_cloud);
        publish_cloud.header.frame_id = "world";
        pub.publish(publish_cloud);
    }
}

void extract_euclidian_clusters(PointCloudXYZ::Ptr& input_cloud, std::vector<PointIndices>& cluster_indices,
                                float cluster_tolerance, int min_cluster_size, int max_cluster_size,
                                KdTreeHandle::Ptr& search_method) {
    (void)search_method;   // ec.setSearchMethod(tree): the neighbour index is rebuilt on the GPU per call
    const size_t m = input_cloud->points.size();
    pcop_ctx* c = default_context(m);
    pcop_params p{};
    pcop_get_params(c, &p);
    p.tol = cluster_tolerance; p.min_cluster = min_cluster_size; p.max_cluster = max_cluster_size;   // od.cpp:447-449
    if (!(p.leaf > 0.0f)) p.leaf = 1.0f;
    check(c, pcop_set_params(c, &p), "pcop_set_params");
    std::vector<int32_t> offsets(m + 1), indices(m ? m : 1);
    int32_t k = 0;
    check(c, pcop_euclidean_clusters(c, m ? &input_cloud->points[0].x : nullptr, PCOP_LAYOUT_AOS16, m, offsets.data(),
                                     indices.data(), &k),
          "pcop_euclidean_clusters");                     // ec.extract(cluster_indices)           od.cpp:454
    cluster_indices.clear();
    cluster_indices.resize((size_t)k);
    for (int32_t cidx = 0; cidx < k; ++cidx) {
        PointIndices& pi = cluster_indices[(size_t)cidx];
        pi.header = input_cloud->header;
        pi.indices.assign(indices.begin() + offsets[cidx], indices.begin() + offsets[cidx + 1]);
    }
}

void build_initial_occupancy_grid_dataset(int grid_size, int grid_height, int grid_width, char* grid_data, float x_min,
                                          float x_max, float y_min, float y_max, float z_min, float z_max,
                                          long long* point_count_array, long long* row_point_averages,
                                          const PointCloudXYZ& input_cloud, PointCloudXYZ& output_cloud,
                                          float block_size, float dev_percent) {
    if (grid_size != grid_height * grid_width) throw std::runtime_error("build_initial_occupancy_grid_dataset: grid_size != height * width");
    const size_t n = input_cloud.points.size();
    pcop_ctx* c = default_context(n);
    pcop_occ_params p{};
    p.x_min = x_min; p.x_max = x_max; p.y_min = y_min; p.y_max = y_max; p.z_min = z_min; p.z_max = z_max;
    p.block_size = block_size; p.dev_percent = dev_percent; p.grid_width = grid_width; p.grid_height = grid_height;
    std::vector<float> kept(n ? 3 * n : 3);
    std::vector<int64_t> counts((size_t)grid_size), rows((size_t)grid_height);
    size_t n_out = 0;
    int32_t bad = 0;
    check(c, pcop_occupancy_filter(c, n ? &input_cloud.points[0].x : nullptr, PCOP_LAYOUT_AOS16, n, PCOP_MEM_HOST, &p, kept.data(),
                                   &n_out, counts.data(), rows.data(), reinterpret_cast<int8_t*>(grid_data), &bad),
          "pcop_occupancy_filter");
    for (int i = 0; i < grid_size; ++i) point_count_array[i] = counts[(size_t)i];
    for (int i = 0; i < grid_height; ++i) row_point_averages[i] = rows[(size_t)i];
    for (size_t i = 0; i < n_out; ++i) {   // output_cloud.push_back(input_cloud.points[i])   od.cpp:214
        PointXYZ q{};
        q.x = kept[3 * i]; q.y = kept[3 * i + 1]; q.z = kept[3 * i + 2];
        output_cloud.points.push_back(q);
    }
    output_cloud.width = (uint32_t)output_cloud.points.size();
    output_cloud.height = 1;
    if (bad) fprintf(stderr, "OUT OF BOUNDS INDEX ACCESS (%d points)\n", bad);   // ROS_ERROR at od.cpp:206
}

void remove_statistical_outliers(PointCloudXYZ::Ptr& input_cloud, PointCloudXYZ& output_cloud, float meanK, float stdDevThres,
                                 Publisher<PointCloud2>& pub, bool publish) {
    const size_t n = input_cloud->points.size();
    pcop_ctx* c = default_context(n);
    std::vector<PointXYZ> kept(n ? n : 1);
    size_t n_out = 0;
    check(c, pcop_statistical_outlier_removal(c, n ? &input_cloud->points[0].x : nullptr, PCOP_LAYOUT_AOS16, n, (int)meanK,   // sor2.setMeanK(meanK)
                                              stdDevThres, &kept[0].x, &n_out, nullptr, nullptr),                              // sor2.filter(output_cloud)
          "pcop_statistical_outlier_removal");
    kept.resize(n_out);
    output_cloud.points.swap(kept);
    output_cloud.header = input_cloud->header;
    output_cloud.width = (uint32_t)n_out;
    output_cloud.height = 1;
    output_cloud.is_dense = true;
    if (publish) {                                       // od.cpp:331-338
        PointCloud2 msg;
        to_pointcloud2(output_cloud, &msg);
        msg.header.frame_id = "world";
        pub.publish(msg);
    }
}

bool segment_plane_and_extract_indices(PointCloudXYZ::Ptr& planar_cloud, PointCloudXYZ::Ptr& indices_cloud, PointCloudXYZ::Ptr& cloud_f,
                                       ModelCoefficients::Ptr& coefficients, PointIndices::Ptr& inliers, PointIndices::Ptr& outliers,
                                       const float axis[3], float dist_thresh, int eps_angle, Publisher<PointCloud2>& pub1,
                                       Publisher<PointCloud2>& pub2, Publisher<PointCloud2>& pub3, bool publish_planar, bool publish_indices,
                                       bool publish_filtered) {
    const size_t n = planar_cloud->points.size();
    pcop_ctx* c = default_context(n);
    pcop_plane_params p{};
    p.dist_thresh = dist_thresh; p.axis[0] = axis[0]; p.axis[1] = axis[1]; p.axis[2] = axis[2];
    p.eps_angle = (double)eps_angle;                     // seg.setEpsAngle(eps_angle): the int goes in as radians (od.cpp:371)
    std::vector<PointXYZ> rest(n ? n : 1), last(n ? n : 1), before(n ? n : 1);
    float last_coeffs[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    size_t n_rest = 0, n_last = 0, n_before = 0;
    int32_t n_planes = 0, found_all = 0;
    check(c, pcop_segment_planes_clouds(c, n ? &planar_cloud->points[0].x : nullptr, PCOP_LAYOUT_AOS16, n, &p, &rest[0].x, &n_rest, &last[0].x, &n_last,
                                        &before[0].x, &n_before, nullptr, 0, &n_planes, &found_all, last_coeffs), "pcop_segment_planes_clouds");
    rest.resize(n_rest); last.resize(n_last); before.resize(n_before);
    const Header hdr = planar_cloud->header;
    auto fill = [&](PointCloudXYZ::Ptr& dst, const std::vector<PointXYZ>& pts) {
        if (!dst) dst.reset(new PointCloudXYZ);
        dst->points = pts; dst->header = hdr; dst->width = (uint32_t)pts.size(); dst->height = 1; dst->is_dense = true;
    };
    if (n_planes) {                                      // (with no plane found the reference leaves all three clouds as they were)
        fill(planar_cloud, rest);
        fill(indices_cloud, last);
        fill(cloud_f, before);                           // extract.filter(*cloud_f); planar_cloud.swap(cloud_f): the cloud the last plane was cut from (od.cpp:389-392)
        if (coefficients) coefficients->values.assign(last_coeffs, last_coeffs + 4);
    }
    if (inliers) inliers->indices.clear();
    if (outliers) outliers->indices.clear();
    auto pub = [&](Publisher<PointCloud2>& pb, const PointCloudXYZ::Ptr& cl) {   // od.cpp:400-426
        PointCloud2 msg;
        to_pointcloud2(*cl, &msg);
        msg.header.frame_id = "world";
        pb.publish(msg);
    };
    if (publish_planar && planar_cloud) pub(pub1, planar_cloud);
    if (publish_indices && indices_cloud) pub(pub2, indices_cloud);
    if (publish_filtered && cloud_f) pub(pub3, cloud_f);
    return found_all != 0;
}

ObstacleNode::ObstacleNode(const NodeParams& prm, size_t max_points, int device) : params(prm) {
    downsample_publisher.topic = "voxel_grid"; downsample_publisher.queue_size = 1;
    centroid_publisher.topic = "centroids"; centroid_publisher.queue_size = 1;
    euc_cluster_publisher.topic = "euc_clusters"; euc_cluster_publisher.queue_size = 5;
    check(nullptr, pcop_create(&ctx_, device, max_points), "pcop_create");
}

ObstacleNode::~ObstacleNode() {
    if (ctx_) pcop_destroy(ctx_);
}

int ObstacleNode::cloud_cb(const PointCloud2& cloud_msg) {
    uint32_t ox = 0, oy = 0, oz = 0;
    if (!find_xyz_fields(cloud_msg, &ox, &oy, &oz, &error_)) return PCOP_ERR_ARG;
    pcop_params p{};
    p.leaf = params.downsample_size;                 // od.cpp:740
    p.tol = params.euc_cluster_tolerance;            // od.cpp:796
    p.min_cluster = params.euc_min_cluster_size;
    p.max_cluster = params.euc_max_cluster_size;
    p.publish_clouds = 1;                            // the voxel cloud is needed on the host for euc_clusters
    int rc = pcop_set_params(ctx_, &p);
    if (rc) { error_ = pcop_last_error(ctx_); return rc; }
    // both clouds come off the GPU as the 16-byte records of pcl::PointXYZ, the layout pcl::toROSMsg publishes (od.cpp:288-295)
    rc = pcop_set_cloud_layout(ctx_, PCOP_LAYOUT_AOS16);
    if (rc) { error_ = pcop_last_error(ctx_); return rc; }
    pcop_result r{};
    node_voxel_grid_sent_ = false;
    if (node_order) {
        // pcl::fromPCLPointCloud2 (od.cpp:688-689) happens on the GPU: the blob goes over as it is (pcop_process_node_pointcloud2); the
        // library checks the message's geometry, the blob's size is checked here, where the vector is
        const size_t n_in = (size_t)cloud_msg.width * cloud_msg.height;
        if (n_in && cloud_msg.data.size() < (size_t)cloud_msg.row_step * cloud_msg.height) { error_ = "PointCloud2 data shorter than height * row_step"; return PCOP_ERR_ARG; }
        pcop_occ_params occ{};
        occ.x_min = params.x_min; occ.x_max = params.x_max; occ.y_min = params.y_min; occ.y_max = params.y_max;
        occ.z_min = params.z_min; occ.z_max = params.z_max; occ.block_size = params.block_size; occ.dev_percent = params.dev_percent;
        pcop_occupancy_grid_dims(occ.x_min, occ.x_max, occ.y_min, occ.y_max, occ.block_size, &occ.grid_width, &occ.grid_height);   // od.cpp:958-959
        pcop_plane_params plane{};                           // od.cpp:768, 778-781: axis (0, 0, 1), the yaml's threshold and "angle"
        plane.dist_thresh = params.plane_segment_dist_thres; plane.axis[0] = 0.0f; plane.axis[1] = 0.0f; plane.axis[2] = 1.0f;
        plane.eps_angle = (double)params.plane_segment_angle;
        const size_t cells = (size_t)std::max(occ.grid_width, 0) * (size_t)std::max(occ.grid_height, 0);
        occupancy_grid.assign(cells, 0); occupancy_counts.assign(cells, 0); occupancy_row_averages.assign((size_t)std::max(occ.grid_height, 0), 0);
        static_assert(sizeof(long long) == sizeof(int64_t), "counts are 64-bit");
        // one fused call feeds every topic of the node's order: the clouds between the steps come back as pcl::PointXYZ records
        const size_t room = std::max<size_t>(n_in, 1);
        std::vector<PointXYZ> c_vg(room), c_so(room), c_pl(room), c_ix(room), c_cf(room);
        pcop_node_clouds clouds{};
        clouds.voxel_grid = &c_vg[0].x; clouds.statistical_outliers = &c_so[0].x; clouds.planar_cloud = &c_pl[0].x;
        clouds.indices_cloud = &c_ix[0].x; clouds.cloud_f = &c_cf[0].x;
        rc = pcop_process_node_pointcloud2(ctx_, cloud_msg.data.data(), cloud_msg.width, cloud_msg.height, cloud_msg.point_step, cloud_msg.row_step,
                                      ox, oy, oz, cloud_msg.is_bigendian ? 1 : 0, 1, &occ, reinterpret_cast<int64_t*>(occupancy_counts.data()), reinterpret_cast<int64_t*>(occupancy_row_averages.data()),
                                      reinterpret_cast<int8_t*>(occupancy_grid.data()), params.statistical_outlier_meanK,
                                      params.statistical_outlier_stdDevThres, remove_planes ? &plane : nullptr, &r, &stats, &clouds);
        if (rc == PCOP_OK && params.publish_point_clouds) {
            auto send = [&](Publisher<PointCloud2>& pub, std::vector<PointXYZ>& pts, int32_t cnt) {   // pcl::toROSMsg + frame_id "world" (od.cpp:288-295, 331-338, 401-426)
                if (cnt < 0) return;
                PointCloudXYZ cl;
                pts.resize((size_t)cnt);
                cl.points.swap(pts);
                cl.width = (uint32_t)cnt; cl.height = 1; cl.is_dense = true;
                PointCloud2 m;
                to_pointcloud2(cl, &m);
                m.header.frame_id = "world";
                pub.publish(m);
            };
            if (clouds.n_voxel_grid >= 0) node_voxel_grid_sent_ = true;
            send(downsample_publisher, c_vg, clouds.n_voxel_grid);
            send(statistical_outlier_publisher, c_so, clouds.n_statistical_outliers);
            send(planar_cloud_publisher, c_pl, clouds.n_planar_cloud);
            send(indices_cloud_publisher, c_ix, clouds.n_indices_cloud);
            send(filtered_cloud_publisher, c_cf, clouds.n_cloud_f);
        }
    } else {
        rc = pcop_process_pointcloud2(ctx_, cloud_msg.data.data(), cloud_msg.width, cloud_msg.height, cloud_msg.point_step,
                                      cloud_msg.row_step, ox, oy, oz, cloud_msg.is_bigendian ? 1 : 0, cloud_msg.is_dense ? 1 : 0, &r);
    }
    if (rc) { error_ = pcop_last_error(ctx_); return rc; }
    passthrough = r.passthrough;
    // the downsampled cloud (statistical_outlier_input_cloud, od.cpp:740-741)
    voxel_cloud.header = cloud_msg.header;
    static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ is {x, y, z, pad = 1.0f}");
    voxel_cloud.points.resize((size_t)r.num_voxels);
    if (r.num_voxels) std::memcpy(static_cast<void*>(&voxel_cloud.points[0]), r.voxel_xyz, (size_t)r.num_voxels * sizeof(PointXYZ));
    voxel_cloud.width = (uint32_t)r.num_voxels; voxel_cloud.height = 1; voxel_cloud.is_dense = true;
    if (params.publish_point_clouds && !node_voxel_grid_sent_) {   // od.cpp:288-295 (node order: sent above, the cloud before the filters)
        PointCloud2 msg;
        to_pointcloud2(voxel_cloud, &msg);
        msg.header.frame_id = "world";
        downsample_publisher.publish(msg);
    }
    // std::vector<pcl::PointIndices> euc_cluster_indices (od.cpp:793-797)
    euc_cluster_indices.assign((size_t)r.num_clusters, PointIndices());
    for (int32_t c = 0; c < r.num_clusters; ++c)
        euc_cluster_indices[(size_t)c].indices.assign(r.indices + r.offsets[c], r.indices + r.offsets[c + 1]);
    // "centroids": one PointWithRad per cluster (the commented-out publish of od.cpp:857-858)
    PointIndicesArray centroids;
    centroids.points.resize((size_t)r.num_clusters);
    if (r.num_clusters) {
        rc = pcop_cluster_summaries(ctx_, &centroids.points[0].x, r.num_clusters);
        if (rc) { error_ = pcop_last_error(ctx_); return rc; }
    }
    centroid_publisher.publish(centroids);
    // "euc_clusters": the points of all kept clusters, frame "/world" (od.cpp:861-866)
    PointCloudXYZ clustered;
    const float* clu = nullptr;
    int32_t nclu = 0;
    rc = pcop_cluster_cloud(ctx_, 0, &clu, &nclu);   // gathered on the GPU, cluster after cluster
    if (rc) { error_ = pcop_last_error(ctx_); return rc; }
    clustered.points.resize((size_t)nclu);
    if (nclu) std::memcpy(static_cast<void*>(&clustered.points[0]), clu, (size_t)nclu * sizeof(PointXYZ));
    PointCloud2 clusters;
    to_pointcloud2(clustered, &clusters);
    clusters.header.frame_id = "/world";
    euc_cluster_publisher.publish(clusters);
    return PCOP_OK;
}

}  // namespace pcop_host
