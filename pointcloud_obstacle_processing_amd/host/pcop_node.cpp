// pcop_node — ROS-free stand-in for the node's hot path: reads the same params.yaml tree and a
// serialized sensor_msgs/PointCloud2, runs the path on the GPU and writes what the node would
// publish on "voxel_grid", "centroids" and "euc_clusters", plus the cluster indices.
//
//   pcop_node --params params.yaml --cloud in.pc2 --out outdir [--device 0] [--stage-api] [--occupancy] [--sor] [--node-order]
//   pcop_node --params params.yaml --cloud in.pc2 --out outdir --gpus G [--same-gpu] [--frames F] [--batch B]
//   pcop_node --params params.yaml --cloud in.pc2 --gpus G --bench [--frames F] [--batch B]
//
// --gpus G runs F frames (frame i = the cloud without its last i * (N / 4F) points, so that no two are alike) through the
// C++ pool of host/pcop_pool.hpp: one feeder thread and one context per GPU, frame f on GPU f mod G, the parameters broadcast
// from rank 0 (RCCL), results in frame order; writes pool_frame_<i>.txt (cluster indices) and pool_frame_<i>.vox (float32
// xyz of the voxel cloud).  --same-gpu maps every worker to device 0 (rehearsal of G > 1 on a one-GPU box).
// --bench (with --gpus) measures the pool instead of writing files: F frames of the whole cloud, each written straight into a
// pinned staging slot the pool hands out (no copy on the caller's side), results taken as they come; prints one JSON line with
// frames/s from host memory (the PCIe-inclusive rate: 12 bytes a point cross the bus inside the timed region).
//
// --node-order (frame mode) runs the node's own order in one call that keeps every intermediate cloud on the GPU:
// bounds filter + occupancy binning -> voxel grid -> statistical outlier removal -> RANSAC plane removal -> clusters
// (od.cpp:727-797), parameters from the yaml tree (--no-planes leaves the plane removal out); voxel_grid.pc2 is then the cloud the cluster indices refer to.
//
// --stage-api runs the two free functions (downsample_cloud, extract_euclidian_clusters) the way
// cloud_cb calls them (od.cpp:740, od.cpp:791-797) instead of the fused frame entry.
// --occupancy (with --stage-api) runs build_initial_occupancy_grid_dataset first, as cloud_cb does (od.cpp:725-731),
// with the limits, block_size and dev_percent of the yaml tree, and writes occupancy.txt (grid size, kept points,
// counts, row means, hole marks).
// --sor (with --stage-api) runs remove_statistical_outliers between the voxel grid and the clustering, as cloud_cb
// does (od.cpp:751-754), with statistical_outlier_meanK / _stdDevThres of the yaml tree, and writes sor.pc2.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <chrono>

#include "obstacle_path.hpp"
#include "pcop_pool.hpp"

using namespace pcop_host;

static void write_indices(const std::string& path, const std::vector<PointIndices>& clusters) {
    std::ofstream f(path);
    f << clusters.size() << "\n";
    for (const PointIndices& c : clusters) {
        f << c.indices.size();
        for (int i : c.indices) f << " " << i;
        f << "\n";
    }
}

int main(int argc, char** argv) {
    std::string params_path, cloud_path, out_dir = ".";
    int device = 0, gpus = 0, frames = 8, batch = 4;
    bool same_gpu = false, bench = false, bench_fill = false;
    bool stage_api = false, occupancy = false, sor = false, node_order = false, planes = true;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--params" && i + 1 < argc) params_path = argv[++i];
        else if (a == "--cloud" && i + 1 < argc) cloud_path = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_dir = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (a == "--batch" && i + 1 < argc) batch = std::atoi(argv[++i]);
        else if (a == "--same-gpu") same_gpu = true;
        else if (a == "--bench") bench = true;
        else if (a == "--bench-fill") { bench = true; bench_fill = true; }
        else if (a == "--stage-api") stage_api = true;
        else if (a == "--occupancy") occupancy = true;
        else if (a == "--sor") sor = true;
        else if (a == "--node-order") node_order = true;
        else if (a == "--no-planes") planes = false;
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (params_path.empty() || cloud_path.empty()) {
        std::fprintf(stderr, "usage: pcop_node --params params.yaml --cloud in.pc2 --out outdir [--device N] [--stage-api]\n");
        return 2;
    }
    std::string err;
    NodeParams prm;
    if (!load_node_params(params_path, &prm, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    PointCloud2 msg;
    if (!read_pointcloud2_file(cloud_path, &msg, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    try {
        if (gpus > 0) {   // the frame-sharded pool
            PointCloudXYZ cloud;
            if (!from_pointcloud2(msg, &cloud, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            const size_t n = cloud.size();
            pcop_params pp{};
            pp.leaf = prm.downsample_size; pp.tol = prm.euc_cluster_tolerance; pp.min_cluster = prm.euc_min_cluster_size;
            pp.max_cluster = prm.euc_max_cluster_size; pp.publish_clouds = 1;
            if (bench) pp.publish_clouds = 0;
            GpuPool pool(gpus, std::max<size_t>(n, 1024), batch, pp, same_gpu);
            if (!pool.ok()) { std::fprintf(stderr, "pool: %s\n", pool.error().c_str()); return 1; }
            for (int g = 0; g < pool.gpus(); ++g) {   // every worker must hold rank 0's struct, bit for bit
                const pcop_params q = pool.params_of(g);
                if (std::memcmp(&q, &pp, sizeof(pcop_params)) != 0) { std::fprintf(stderr, "pool: worker %d did not receive the parameters\n", g); return 1; }
            }
            frames = std::max(1, frames);
            if (bench) {   // throughput of the pool from host memory: frames filled in place, nothing written to disk
                std::vector<float> soa(3 * n);
                for (size_t k = 0; k < n; ++k) { soa[k] = cloud.points[k].x; soa[n + k] = cloud.points[k].y; soa[2 * n + k] = cloud.points[k].z; }
                size_t got = 0, voxels = 0, clusters = 0;
                int bad = 0;
                auto take = [&]() -> bool {
                    PoolResult r;
                    if (!pool.next(&r)) return false;
                    if (r.status) { if (!bad) std::fprintf(stderr, "frame %llu on worker %d: %s (%s)\n", (unsigned long long)r.frame, r.gpu, pcop_status_string(r.status), r.error.c_str()); bad = 1; }
                    voxels = (size_t)r.num_voxels; clusters = r.offsets.empty() ? 0 : r.offsets.size() - 1;
                    ++got;
                    return true;
                };
                const int inflight = 3 * batch * pool.gpus();
                // warm-up: one round of every slot
                size_t sent = 0;
                // A slot is written once, on its first use: after that the points are where a driver's DMA would have put them, and the
                // timed loop measures the pool (queueing, 12 MB a frame over PCIe, the GPU, the results), not this thread's memcpy.
                // --bench-fill writes every frame (one pass over the cloud at memcpy speed on the caller's thread).
                std::vector<float*> filled;
                auto push = [&](int count) {
                    for (int i = 0; i < count && !bad; ++i) {
                        float* dst = pool.acquire();
                        if (!dst) { bad = 1; break; }
                        if (bench_fill || std::find(filled.begin(), filled.end(), dst) == filled.end()) { std::memcpy(dst, soa.data(), 3 * n * sizeof(float)); if (!bench_fill) filled.push_back(dst); }
                        if (!pool.commit(n)) { bad = 1; break; }
                        ++sent;
                        while (!bad && sent - got > (size_t)inflight) take();
                    }
                };
                push(inflight);
                while (got < sent && take()) {}
                const auto b0 = std::chrono::steady_clock::now();
                const size_t before = got;
                push(frames);
                while (got < sent && take()) {}
                const double bdt = std::chrono::duration<double>(std::chrono::steady_clock::now() - b0).count();
                if (bad) { std::fprintf(stderr, "pool bench failed (%s)\n", pool.error().c_str()); return 1; }
                std::printf("{\"pool_bench\": {\"gpus\": %d, \"same_gpu\": %s, \"batch\": %d, \"contexts_per_gpu\": 2, \"frames\": %zu, \"points_per_frame\": %zu, \"seconds\": %.6f, "
                            "\"frames_per_s\": %.1f, \"pcie_GBs\": %.2f, \"voxels\": %zu, \"clusters\": %zu, \"input\": \"%s\", \"params_by\": \"%s\"}}\n",
                            pool.gpus(), same_gpu ? "true" : "false", batch, got - before, n, bdt, (got - before) / bdt, (got - before) / bdt * 12.0 * n / 1e9, voxels, clusters,
                            bench_fill ? "host memory, every frame written into its pinned staging slot by the caller (one memcpy pass, acquire / commit)" : "pinned staging slots already holding the frame (acquire / commit; a driver's DMA target): no copy on the caller's thread",
                            pool.broadcast_kind());
                return 0;
            }
            const size_t cut = n / (4u * (size_t)frames);
            std::vector<float> soa;
            const auto t0 = std::chrono::steady_clock::now();
            size_t done = 0;
            int failed = 0;   // (an error is remembered and returned from main once the pool has been drained: its destructor joins the workers)
            auto drain_one = [&]() -> bool {
                PoolResult r;
                if (!pool.next(&r)) return false;
                if (r.status) { std::fprintf(stderr, "frame %llu on worker %d: %s (%s)\n", (unsigned long long)r.frame, r.gpu, pcop_status_string(r.status), r.error.c_str()); failed = 1; ++done; return true; }
                if (r.gpu != (int)(r.frame % (uint64_t)pool.gpus())) { std::fprintf(stderr, "frame %llu ran on worker %d\n", (unsigned long long)r.frame, r.gpu); failed = 1; ++done; return true; }
                std::vector<PointIndices> clusters(r.offsets.size() - 1);
                for (size_t c = 0; c + 1 < r.offsets.size(); ++c) clusters[c].indices.assign(r.indices.begin() + r.offsets[c], r.indices.begin() + r.offsets[c + 1]);
                write_indices(out_dir + "/pool_frame_" + std::to_string(r.frame) + ".txt", clusters);
                std::ofstream f(out_dir + "/pool_frame_" + std::to_string(r.frame) + ".vox", std::ios::binary);
                f.write(reinterpret_cast<const char*>(r.voxel_xyz.data()), (std::streamsize)(r.voxel_xyz.size() * sizeof(float)));
                ++done;
                return true;
            };
            for (int i = 0; i < frames; ++i) {
                const size_t ni = n - (size_t)i * cut;
                soa.resize(3 * ni);
                for (size_t k = 0; k < ni; ++k) { soa[k] = cloud.points[k].x; soa[ni + k] = cloud.points[k].y; soa[2 * ni + k] = cloud.points[k].z; }
                if (!pool.submit(soa.data(), ni)) { std::fprintf(stderr, "pool: submit failed (%s)\n", pool.error().c_str()); return 1; }
                while ((size_t)i + 1 - done > (size_t)(3 * batch * pool.gpus())) drain_one();   // (results are taken as they come, so that the staging never fills)
            }
            while (drain_one()) {}
            if (failed) return 1;
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("pool: %d worker(s)%s, batch %d, %zu frames in %.3f s (%.1f frames/s, host-memory input), parameters by %s\n", pool.gpus(),
                        same_gpu ? " on one device" : "", batch, done, dt, done / dt, pool.broadcast_kind());
            return 0;
        }
        if (stage_api) {
            PointCloudXYZ::Ptr in(new PointCloudXYZ);
            if (!from_pointcloud2(msg, in.get(), &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            if (occupancy) {   // od.cpp:725-731 with the grid size of od.cpp:958-960
                int32_t gw = 0, gh = 0;
                pcop_occupancy_grid_dims(prm.x_min, prm.x_max, prm.y_min, prm.y_max, prm.block_size, &gw, &gh);
                const int gsize = gw * gh;
                std::vector<char> grid((size_t)gsize);
                std::vector<long long> cnt((size_t)gsize), rows((size_t)gh);
                PointCloudXYZ::Ptr kept(new PointCloudXYZ);
                build_initial_occupancy_grid_dataset(gsize, gh, gw, grid.data(), prm.x_min, prm.x_max, prm.y_min, prm.y_max,
                                                     prm.z_min, prm.z_max, cnt.data(), rows.data(), *in, *kept,
                                                     prm.block_size, prm.dev_percent);
                std::ofstream f(out_dir + "/occupancy.txt");
                f << gw << " " << gh << " " << kept->size() << "\n";
                for (long long v : cnt) f << v << " ";
                f << "\n";
                for (long long v : rows) f << v << " ";
                f << "\n";
                for (char v : grid) f << (int)v << " ";
                f << "\n";
                kept->header = in->header;
                in = kept;   // downsample_input_cloud
            }
            Publisher<PointCloud2> pub;
            pub.topic = "voxel_grid";
            PointCloudXYZ::Ptr vox(new PointCloudXYZ);
            downsample_cloud(in, *vox, prm.downsample_size, pub, prm.publish_point_clouds);          // od.cpp:740
            if (sor) {                                                                                // od.cpp:751-754
                PointCloudXYZ::Ptr filtered(new PointCloudXYZ);
                Publisher<PointCloud2> sor_pub;
                sor_pub.topic = "statistical_outliers";
                remove_statistical_outliers(vox, *filtered, (float)prm.statistical_outlier_meanK, prm.statistical_outlier_stdDevThres,
                                            sor_pub, true);
                if (!write_pointcloud2_file(out_dir + "/sor.pc2", sor_pub.last, &err)) throw std::runtime_error(err);
                vox = filtered;
            }
            KdTreeHandle::Ptr tree(new KdTreeHandle);
            tree->setInputCloud(vox);                                                                 // od.cpp:791-792
            std::vector<PointIndices> clusters;
            extract_euclidian_clusters(vox, clusters, prm.euc_cluster_tolerance, prm.euc_min_cluster_size,
                                       prm.euc_max_cluster_size, tree);                              // od.cpp:796
            if (pub.published && !write_pointcloud2_file(out_dir + "/voxel_grid.pc2", pub.last, &err)) throw std::runtime_error(err);
            write_indices(out_dir + "/cluster_indices.txt", clusters);
            std::printf("stage-api: N=%zu M=%zu K=%zu\n", in->size(), vox->size(), clusters.size());
            release_default_context();
            return 0;
        }
        ObstacleNode node(prm, std::max<size_t>((size_t)msg.width * msg.height, 1024), device);
        node.node_order = node_order;
        node.remove_planes = planes;
        const int rc = node.cloud_cb(msg);
        if (rc) { std::fprintf(stderr, "cloud_cb failed: %s (%s)\n", pcop_status_string(rc), node.last_error().c_str()); return 1; }
        if (node.downsample_publisher.published &&
            !write_pointcloud2_file(out_dir + "/voxel_grid.pc2", node.downsample_publisher.last, &err)) throw std::runtime_error(err);
        if (!write_pointcloud2_file(out_dir + "/euc_clusters.pc2", node.euc_cluster_publisher.last, &err)) throw std::runtime_error(err);
        if (node_order) {   // the topics between the steps (od.cpp:1005-1008), from the same fused call; clustered.pc2 = the cloud the cluster indices refer to
            auto put = [&](const char* name, const Publisher<PointCloud2>& pub) {
                if (pub.published && !write_pointcloud2_file(out_dir + "/" + name + ".pc2", pub.last, &err)) throw std::runtime_error(err);
            };
            put("statistical_outliers", node.statistical_outlier_publisher);
            put("planar_cloud", node.planar_cloud_publisher);
            put("indices_cloud", node.indices_cloud_publisher);
            put("cloud_f", node.filtered_cloud_publisher);
            PointCloud2 m;
            to_pointcloud2(node.voxel_cloud, &m);
            if (!write_pointcloud2_file(out_dir + "/clustered.pc2", m, &err)) throw std::runtime_error(err);
        }
        {
            std::ofstream f(out_dir + "/centroids.txt");   // PointIndicesArray: one "x y z r" line per PointWithRad
            f.precision(9);
            for (const PointWithRad& p : node.centroid_publisher.last.points) f << p.x << " " << p.y << " " << p.z << " " << p.r << "\n";
        }
        write_indices(out_dir + "/cluster_indices.txt", node.euc_cluster_indices);
        std::printf("frame: N=%u M=%zu K=%zu passthrough=%d\n", msg.width * msg.height, node.voxel_cloud.size(),
                    node.euc_cluster_indices.size(), node.passthrough);
        if (node_order) std::printf("node order: inside the box %d, voxels %d, inliers %d, planes %d, clustered %d, threshold %.17g\n", node.stats.n_filtered,
                                    node.stats.n_voxels, node.stats.n_inliers, node.stats.n_planes, node.stats.n_clustered, node.stats.threshold);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
