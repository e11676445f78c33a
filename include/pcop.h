/*
 * pcop.h — C ABI of the MI355X-native obstacle hot path
 * (VoxelGrid downsample -> fixed-radius neighbour graph -> Euclidean clusters).
 *
 * This is the drop-in boundary for the three call clusters of the reference
 * node minibot_cr18/src/obstacle_detection.cpp ("od.cpp"):
 *
 *   od.cpp:271-296  downsample_cloud()            -> pcop_voxel_downsample()
 *   od.cpp:791-792  search::KdTree + setInputCloud -> (internal: device cell grid, no entry point)
 *   od.cpp:430-455  extract_euclidian_clusters()   -> pcop_euclidean_clusters()
 *   od.cpp:688-689  PointCloud2 -> PointXYZ        -> pcop_process_pointcloud2()
 *   od.cpp:740-797  the chain inside cloud_cb()    -> pcop_process_pointcloud2() / pcop_process_xyz()
 *   od.cpp:806-814  (commented) centroids msg      -> pcop_cluster_summaries()
 *
 * Everything behind these entry points is hand-written HIP for gfx950; there
 * is no CPU backend in this library.  All functions return 0 on success or a
 * negative pcop_status; nothing throws across the ABI.  All buffers are owned
 * by the caller unless stated.  A context is bound to one GPU and one HIP
 * stream, is not thread-safe, and different contexts are independent.
 */
#ifndef PCOP_H
#define PCOP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCOP_VERSION_MAJOR 0
#define PCOP_VERSION_MINOR 1

/* ---- status codes -------------------------------------------------------- */
typedef enum pcop_status {
    PCOP_OK              =  0,
    PCOP_ERR_ARG         = -1,  /* null pointer / bad layout / bad size                     */
    PCOP_ERR_PARAM       = -2,  /* leaf or tolerance not finite-positive, min>max ...       */
    PCOP_ERR_NO_DEVICE   = -3,  /* no HIP device / bad device id (there is no CPU fallback) */
    PCOP_ERR_HIP         = -4,  /* a HIP runtime call failed (see pcop_last_error)          */
    PCOP_ERR_CAPACITY    = -5,  /* n > max_points given to pcop_create                      */
    PCOP_ERR_NONFINITE   = -6,  /* NaN/Inf where the reference would be undefined           */
    PCOP_ERR_RANGE       = -7,  /* coarse neighbour grid would exceed 2^32 cells            */
    PCOP_ERR_STATE       = -8   /* call order (e.g. summaries before a frame)               */
} pcop_status;

/* ---- point memory layouts ------------------------------------------------ */
#define PCOP_LAYOUT_AOS16 0   /* pcl::PointXYZ in memory: float x,y,z,pad (16 B/pt)     */
#define PCOP_LAYOUT_SOA   1   /* three planes x[n] y[n] z[n], plane stride = n floats   */
#define PCOP_LAYOUT_AOS12 2   /* packed float x,y,z (12 B/pt)                           */

/* ---- memory spaces for the *_dev entry points ---------------------------- */
#define PCOP_MEM_HOST   0
#define PCOP_MEM_DEVICE 1

/* The parameters of od.cpp:105,113-115,118 (yaml keys downsample_size,
 * euc_cluster_tolerance, euc_min_cluster_size, euc_max_cluster_size,
 * publish_point_clouds).  POD, 32 bytes: this struct is the only thing that is
 * broadcast between GPUs (RCCL) in a multi-GPU run. */
typedef struct pcop_params {
    float   leaf;            /* downsample_size          (od.cpp:964) */
    float   tol;             /* euc_cluster_tolerance    (od.cpp:972) */
    int32_t min_cluster;     /* euc_min_cluster_size     (od.cpp:973) */
    int32_t max_cluster;     /* euc_max_cluster_size     (od.cpp:974) */
    int32_t publish_clouds;  /* publish_point_clouds     (od.cpp:945) */
    int32_t reserved[3];
} pcop_params;

/* Result of one frame.  Pointers refer to pinned host memory owned by the
 * context; they stay valid until the next frame call on the same context. */
typedef struct pcop_result {
    int32_t        passthrough;   /* 1 = VoxelGrid overflow rule fired: voxel cloud == input cloud */
    int32_t        num_voxels;    /* M: points in the downsampled cloud                            */
    int32_t        num_clusters;  /* K: clusters kept, ordered as PCL orders them                  */
    int32_t        num_indices;   /* offsets[K]                                                    */
    const int32_t* offsets;       /* K+1 CSR offsets                                               */
    const int32_t* indices;       /* indices into the downsampled cloud, ascending per cluster     */
    const float*   voxel_xyz;     /* M centroids in the context's cloud layout (AOS12 unless
                                     pcop_set_cloud_layout chose another); NULL unless publish_clouds != 0 */
} pcop_result;

/* Per-stage device times of the last frame (milliseconds, HIP events on the
 * context's stream).  Filled only when profiling was enabled on the context. */
#define PCOP_MAX_STAGES 128
typedef struct pcop_stage_times {
    int32_t     count;
    const char* name[PCOP_MAX_STAGES];
    float       ms[PCOP_MAX_STAGES];
} pcop_stage_times;

typedef struct pcop_ctx pcop_ctx;

/* ---- lifecycle ----------------------------------------------------------- */
/* device_id >= 0 selects a HIP device; anything else is PCOP_ERR_NO_DEVICE.
 * max_points bounds n (and m) of every later call on this context. */
int  pcop_create(pcop_ctx** ctx, int device_id, size_t max_points);
/* Same, with room for max_frames frames that are processed together (pcop_submit_batch). */
int  pcop_create_batch(pcop_ctx** ctx, int device_id, size_t max_points, int max_frames);
int  pcop_max_frames(const pcop_ctx* ctx);
void pcop_destroy(pcop_ctx* ctx);
int  pcop_set_params(pcop_ctx* ctx, const pcop_params* p);
int  pcop_get_params(const pcop_ctx* ctx, pcop_params* p);
const char* pcop_last_error(const pcop_ctx* ctx);
const char* pcop_status_string(int status);
int  pcop_set_profiling(pcop_ctx* ctx, int enable);
int  pcop_get_stage_times(const pcop_ctx* ctx, pcop_stage_times* out);
/* The HIP stream (hipStream_t) the context launches on, as an opaque pointer. */
void* pcop_stream(const pcop_ctx* ctx);

/* ---- stage API: mirrors downsample_cloud / extract_euclidian_clusters ---- */
/* VoxelGrid<PointXYZ>::filter with a cubic leaf (od.cpp:282-285).
 * is_dense==0 drops non-finite points first, as PCL does.  Output has the
 * layout of the input; for SOA the output plane stride is n.  *passthrough is
 * set when PCL's "leaf size too small" rule copies the input unchanged. */
int pcop_voxel_downsample(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int is_dense,
                          float* out_xyz, size_t* out_m, int* passthrough);

/* EuclideanClusterExtraction::extract (od.cpp:446-454) on an m-point cloud.
 * offsets has room for m+1 entries, indices for m entries. */
int pcop_euclidean_clusters(pcop_ctx* ctx, const float* xyz, int layout, size_t m,
                            int32_t* offsets, int32_t* indices, int32_t* k);

/* ---- fused frame API: what cloud_cb's replacement calls ------------------- */
/* data/width/height/point_step/row_step/is_bigendian/is_dense are the fields of
 * sensor_msgs/PointCloud2; off_x/off_y/off_z are the byte offsets of its
 * FLOAT32 fields "x","y","z" (od.cpp:688-689 resolves them by name). */
int pcop_process_pointcloud2(pcop_ctx* ctx, const uint8_t* data, uint32_t width, uint32_t height,
                             uint32_t point_step, uint32_t row_step, uint32_t off_x, uint32_t off_y,
                             uint32_t off_z, int is_bigendian, int is_dense, pcop_result* out);

/* Same chain on an XYZ array; mem says where xyz lives (device pointers must
 * belong to the context's GPU).  This is the entry the benchmark times with
 * the cloud already resident in HBM. */
int pcop_process_xyz(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int is_dense, int mem,
                     pcop_result* out);

/* Asynchronous split of pcop_process_xyz for several frames in flight (one
 * context per frame): submit enqueues the whole frame on the context's stream
 * and returns; wait blocks until it is done and fills *out. */
int pcop_submit_xyz(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int is_dense, int mem);
int pcop_wait(pcop_ctx* ctx, pcop_result* out);

/* Batched form for throughput: nframes independent clouds (xyz[f], n[f]) go through the chain in
 * the same kernel launches (frames are independent: od.cpp:699-797 keeps no cross-frame state).
 * pcop_wait_batch fills out[0..nframes).  Results per frame are those of pcop_process_xyz. */
int pcop_submit_batch(pcop_ctx* ctx, const float* const* xyz, int layout, const size_t* n, int nframes,
                      int is_dense, int mem);
int pcop_wait_batch(pcop_ctx* ctx, pcop_result* out, int nframes);

/* PointWithRad[] of the `centroids` topic (msg/PointWithRad.msg:1-4,
 * msg/PointIndicesArray.msg:1) for the last frame / last cluster call:
 * xyzr[4*c..] = centre x,y,z and radius r of cluster c (definition: DESIGN.md). */
int pcop_cluster_summaries(pcop_ctx* ctx, float* xyzr, int32_t cap_clusters);
int pcop_cluster_summaries_frame(pcop_ctx* ctx, int frame, float* xyzr, int32_t cap_clusters);

/* ---- egress in the layout pcl::toROSMsg gives a PointCloud<PointXYZ> (od.cpp:288-295, 861-866) -------------
 * The frame API's voxel cloud (pcop_result.voxel_xyz, the `voxel_grid` topic) is written by the GPU in `layout`:
 * PCOP_LAYOUT_AOS16 = the 16-byte {x, y, z, 1.0f} records of pcl::PointXYZ, i.e. the data blob of the PointCloud2
 * message as it stands (no host re-pack); PCOP_LAYOUT_AOS12 (the default) = packed xyz.  SOA is not offered here. */
int pcop_set_cloud_layout(pcop_ctx* ctx, int layout);
/* The `euc_clusters` cloud of a finished frame (od.cpp:861-866, commented intent): the points of all kept clusters,
 * cluster after cluster in the order of pcop_result.offsets, gathered on the GPU as 16-byte {x, y, z, 1.0f} records.
 * *xyz16 points to pinned host memory owned by the context (valid until the next call on it); *n_points =
 * pcop_result.num_indices. */
int pcop_cluster_cloud(pcop_ctx* ctx, int frame, const float** xyz16, int32_t* n_points);

/* ---- the step before the hot path: bounds filter + occupancy binning -------------
 * Replaces build_initial_occupancy_grid_dataset (od.cpp:175-269) with its helpers
 * get_occupancy_grid_x_y (od.cpp:134-150) and get_occupancy_grid_location (od.cpp:153-157):
 * drops the points outside the box (and those whose x is NaN), counts the kept points per 2-D grid
 * cell, takes the integer mean of every grid row and marks the cells below (1 - dev_percent) of
 * their row's mean as holes (100).  The kept points, in input order, are the cloud the reference
 * hands to downsample_cloud (od.cpp:740); here they stay resident in HBM for pcop_process_filtered. */
typedef struct pcop_occ_params {
    float   x_min, x_max;      /* pt_lower_lim_x, pt_upper_lim_x   (od.cpp:947-948) */
    float   y_min, y_max;      /* pt_lower_lim_y, pt_upper_lim_y   (od.cpp:949-950) */
    float   z_min, z_max;      /* pt_lower_lim_z, pt_upper_lim_z   (od.cpp:951-952) */
    float   block_size;        /* block_size                       (od.cpp:955)     */
    float   dev_percent;       /* dev_percent                      (od.cpp:956)     */
    int32_t grid_width;        /* occupancy_grid_width             (od.cpp:958)     */
    int32_t grid_height;       /* occupancy_grid_height            (od.cpp:959)     */
} pcop_occ_params;

/* occupancy_grid_width / _height as main() derives them from the limits (od.cpp:958-959). */
void pcop_occupancy_grid_dims(float x_min, float x_max, float y_min, float y_max, float block_size,
                              int32_t* grid_width, int32_t* grid_height);

/* xyz/layout/n/mem as in pcop_process_xyz.  Outputs (each may be NULL): out_xyz = the kept points
 * (AOS12, caller provides room for n points); counts[w*h] = occupancy_grid_pt_counts; row_avg[h] =
 * row_averages; grid[w*h] = occupancy_grid_data (100 or 0).  *n_out = kept points; *index_errors =
 * kept points whose cell index was >= w*h (the reference logs them, does not count them and keeps
 * them, od.cpp:205-213). */
int pcop_occupancy_filter(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int mem, const pcop_occ_params* p,
                          float* out_xyz, size_t* n_out, int64_t* counts, int64_t* row_avg, int8_t* grid,
                          int32_t* index_errors);

/* The chain of pcop_process_xyz on the cloud the last pcop_occupancy_filter kept (it never leaves
 * HBM): build_initial_occupancy_grid_dataset -> downsample_cloud -> ... of cloud_cb (od.cpp:727-797). */
int pcop_process_filtered(pcop_ctx* ctx, pcop_result* out);

/* ---- between voxel grid and clustering in the real node: statistical outlier removal -------
 * Replaces remove_statistical_outliers (od.cpp:316-340: pcl::StatisticalOutlierRemoval with setMeanK(mean_k),
 * setStddevMulThresh(stddev_mul), called at od.cpp:751-754).  The GPU computes, for every point, the mean
 * distance to its mean_k nearest neighbours (exact k-NN on a dense cell grid, float32 L2_Simple distances,
 * square roots summed in double in ascending order); the global mean / standard deviation of those values is a
 * sequential double sum in the reference and is formed on the host in the same order, as is the final
 * `distance > threshold` selection.  xyz / layout as in pcop_voxel_downsample (host memory); out_xyz receives
 * the kept points in input order and the same layout (room for n points).  distances (n floats) and threshold
 * may be NULL.  1 <= mean_k <= 31.  An empty cloud gives an empty cloud (as sor2.filter does: the robot sees
 * nothing in range and cloud_cb carries on); a cloud with 1 .. mean_k points, or with a non-finite coordinate,
 * is refused (PCOP_ERR_PARAM / PCOP_ERR_NONFINITE): the reference reads past its search result on the former. */
int pcop_statistical_outlier_removal(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int mean_k, float stddev_mul,
                                     float* out_xyz, size_t* n_out, float* distances, double* threshold);

/* ---- RANSAC plane removal --------------------------------------------------------------------
 * Replaces segment_plane_and_extract_indices (od.cpp:342-428, called at od.cpp:778-781): pcl::SACSegmentation with
 * SACMODEL_PERPENDICULAR_PLANE, SAC_RANSAC, optimised coefficients, PCL's defaults (50 iterations, probability 0.99, the
 * fixed seed of a non-random model) inside "while more than 30 % of the cloud is left: find a plane, stop if none, keep the
 * points outside it" with pcl::ExtractIndices.  The GPU counts the inliers of all candidate planes of a round in one pass
 * and splits the cloud; the sampling, the plane through three points and the least-squares refit are sequential by nature
 * and run on the host, in the reference's order of float operations.
 *   dist_thresh  plane_segment_dist_thres (od.cpp:110, 969)   axis  (0, 0, 1) in the node (od.cpp:768)
 *   eps_angle    plane_segment_angle as the node passes it: an int that PCL reads as radians (od.cpp:111, 371, 970) */
typedef struct pcop_plane_params {
    float  dist_thresh;
    float  axis[3];
    double eps_angle;
} pcop_plane_params;
/* Stage form (host memory in and out, like the reference's function): out_xyz receives what is left (layout of the input,
 * room for n points), last_plane_xyz (may be NULL; same layout, room for n points) the points of the last plane removed
 * ("indices_cloud" in the reference), coeffs (may be NULL, room for 4 * cap_planes floats) the refined coefficients of
 * every plane removed, *n_planes their number, *found_all 0 when a round found no plane (the node prints "Couldn't
 * estimate a planar model" and leaves the loop). */
int pcop_segment_planes(pcop_ctx* ctx, const float* xyz, int layout, size_t n, const pcop_plane_params* p,
                        float* out_xyz, size_t* n_out, float* last_plane_xyz, size_t* n_last_plane,
                        float* coeffs, int32_t cap_planes, int32_t* n_planes, int32_t* found_all);
/* The same with the reference's third cloud: before_last_xyz (may be NULL; same layout, room for n points) receives the cloud
 * the last plane was cut from - "cloud_f" after the loop's planar_cloud.swap(cloud_f) (od.cpp:389-392), which is what the node
 * publishes on `cloud_f` - and last_coeffs (may be NULL, 4 floats) the coefficients of the last plane whatever cap_planes is.
 * *n_planes is the number of planes removed and may exceed cap_planes; only min(*n_planes, cap_planes) sets are in coeffs. */
int pcop_segment_planes_clouds(pcop_ctx* ctx, const float* xyz, int layout, size_t n, const pcop_plane_params* p,
                               float* out_xyz, size_t* n_out, float* last_plane_xyz, size_t* n_last_plane,
                               float* before_last_xyz, size_t* n_before_last,
                               float* coeffs, int32_t cap_planes, int32_t* n_planes, int32_t* found_all, float* last_coeffs);

/* ---- the node's chain in one call, nothing but results leaving HBM (od.cpp:727-797) -----------------------------
 * build_initial_occupancy_grid_dataset (od.cpp:727-731) -> downsample_cloud (740-741) -> remove_statistical_outliers
 * (751-754) -> segment_plane_and_extract_indices (778-781) -> kd-tree + extract_euclidian_clusters (791-797), in the
 * node's order.
 *   occ      NULL: no bounds filter / occupancy binning, the chain starts at the voxel grid.  Otherwise as in
 *            pcop_occupancy_filter; counts / row_avg / grid (each may be NULL) receive its grid outputs.
 *   mean_k   0: no outlier removal.  Otherwise as in pcop_statistical_outlier_removal, applied to the voxel cloud.
 *   plane    NULL: no plane removal.  Otherwise as in pcop_segment_planes, applied to what the outlier removal kept.
 * The kept cloud of every step stays on the device; the point counts flow through device memory.  With outlier removal
 * the call synchronises once in the middle: the n mean neighbour distances go to the host, which forms the two
 * sequential double sums of the reference (a parallel sum would round differently) and hands back the threshold; the
 * plane removal synchronises once per candidate-plane count (its sampling and refit are the host's).
 * out->voxel_xyz / num_voxels describe the cloud the cluster indices refer to (the voxel cloud, or what the later steps
 * kept of it); stats (may be NULL) reports the point counts after every step. */
typedef struct pcop_node_stats {
    int32_t n_in;          /* points handed in                                   */
    int32_t n_filtered;    /* points inside the box (== n_in without occ)         */
    int32_t index_errors;  /* as in pcop_occupancy_filter                         */
    int32_t n_voxels;      /* voxel grid output                                   */
    int32_t n_inliers;     /* after the outlier removal (== n_voxels without it)  */
    int32_t n_planes;      /* planes removed                                      */
    int32_t n_clustered;   /* points that entered the clustering                  */
    int32_t planes_found_all; /* 0: a round of the plane removal found no plane   */
    double  threshold;     /* mean + stddev_mul * sigma (0 without outlier removal) */
} pcop_node_stats;
int pcop_process_node(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int is_dense, int mem,
                      const pcop_occ_params* occ, int64_t* counts, int64_t* row_avg, int8_t* grid,
                      int mean_k, float stddev_mul, const pcop_plane_params* plane, pcop_result* out, pcop_node_stats* stats);

/* The same call with the clouds the node publishes on its way (od.cpp:1004-1008), so that one fused call feeds every topic:
 *   voxel_grid            the voxel cloud BEFORE the outlier removal          (downsample_cloud's publish, od.cpp:288-295)
 *   statistical_outliers  what the outlier removal kept                       (od.cpp:331-338; only with mean_k != 0)
 *   planar_cloud          what the plane removal left                         (od.cpp:401-408; only with plane != NULL)
 *   indices_cloud         the points of the LAST plane removed                (od.cpp:410-417)
 *   cloud_f               the cloud before the last plane was removed: the reference's loop ends every round with
 *                         planar_cloud.swap(cloud_f) (od.cpp:389-392), so this - not what is left - is what
 *                         filtered_cloud_publisher sends (od.cpp:419-426)
 * Every buffer may be NULL; one that is not receives 16-byte records {x, y, z, 1.0f} (pcl::PointXYZ, what pcl::toROSMsg lays
 * out) and needs room for n points; n_* report the point counts (a cloud the call did not produce - its step was not asked
 * for - : -1).  When the plane removal was asked for and no plane went (the first fit found no inliers, or the cloud was empty),
 * indices_cloud and cloud_f both carry the cloud the plane stage was given: cloud_cb initialises indices_cloud_y and
 * filtered_cloud_y as copies of the outlier removal's output (od.cpp:774-775) and publishes them whatever happened (od.cpp:410-426). */
typedef struct pcop_node_clouds {
    float* voxel_grid;           int32_t n_voxel_grid;
    float* statistical_outliers; int32_t n_statistical_outliers;
    float* planar_cloud;         int32_t n_planar_cloud;
    float* indices_cloud;        int32_t n_indices_cloud;
    float* cloud_f;              int32_t n_cloud_f;
} pcop_node_clouds;
int pcop_process_node_clouds(pcop_ctx* ctx, const float* xyz, int layout, size_t n, int is_dense, int mem,
                      const pcop_occ_params* occ, int64_t* counts, int64_t* row_avg, int8_t* grid,
                      int mean_k, float stddev_mul, const pcop_plane_params* plane, pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds);

/* The same call fed with the sensor_msgs/PointCloud2 blob itself (replaces pcl_conversions::toPCL + pcl::fromPCLPointCloud2,
 * od.cpp:688-689, in front of the node's chain): the bytes go to the GPU as they are and k_unpack_points picks x / y / z by
 * field offset there - any point_step, row padding, big-endian floats - so a caller has no per-point loop of its own.  The
 * message geometry is checked before anything is read (PCOP_ERR_ARG: a field offset beyond point_step - 4, row_step below
 * width * point_step; PCOP_ERR_CAPACITY: width * height above the context's max_points); `data` must hold
 * height * row_step bytes (the caller checks its vector's size: the ABI sees a pointer). */
int pcop_process_node_pointcloud2(pcop_ctx* ctx, const uint8_t* data, uint32_t width, uint32_t height, uint32_t point_step,
                      uint32_t row_step, uint32_t off_x, uint32_t off_y, uint32_t off_z, int is_bigendian, int is_dense,
                      const pcop_occ_params* occ, int64_t* counts, int64_t* row_avg, int8_t* grid,
                      int mean_k, float stddev_mul, const pcop_plane_params* plane, pcop_result* out, pcop_node_stats* stats, pcop_node_clouds* clouds);

/* Device-side allocation helpers so a host program without its own HIP code
 * (e.g. a ctypes caller) can keep clouds resident in HBM. */
int pcop_dev_alloc(pcop_ctx* ctx, size_t bytes, void** dev_ptr);
int pcop_dev_free(pcop_ctx* ctx, void* dev_ptr);
int pcop_dev_upload(pcop_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int pcop_dev_download(pcop_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int pcop_device_count(void);
/* Keeps the CALLING thread (and the threads it starts afterwards) on the CPUs next to GPU `device_id`
 * (/sys/bus/pci/devices/<bus id>/local_cpulist, intersected with the CPUs the thread may run on now), so that the pinned buffers it
 * allocates and first touches - a context's mailboxes and result buffers, a pool's staging slots - sit on the GPU's NUMA node.
 * Call it before pcop_create on that thread.  Returns the number of CPUs the thread is bound to, 0 when nothing was changed
 * (no NUMA information, or fewer than four of those CPUs are allowed to this thread), PCOP_ERR_NO_DEVICE for a device that does not exist.
 * No counterpart in the reference (one GPU-less process, od.cpp:995-1001); used by host/pcop_pool.cpp's workers and bench.py's ranks. */
int pcop_bind_thread_near_device(int device_id);

/* Test hook: device address and size of a named internal buffer of the last frame
 * ("cen_x", "rec_a", "cell_key", "parent", "label", ... ; "state" = the frame's DevState; "name@k": frame slot k of a batch).
 * Used by the stage-wise parity tests only; not part of the drop-in surface. */
int pcop_debug_buffer(pcop_ctx* ctx, const char* name, void** dev_ptr, size_t* bytes);
/* Test hook: a named host-side counter of the context ("internal_retries": batches queued a second time after a bounded
 * device-side wait ran out; "replans": batches re-run with larger launch parameters or on another path; "stale_reruns": batches re-run on scrubbed
 * occupancy flags after a kernel found a flag no point of the batch had set; "lds_ordered": 1 while
 * ranks come from lane-ordered LDS atomics, 0 once a self-check sent the context back to peer masks; "stream_checked": the
 * streaming voxel path's partition has been verified against the peer-mask ranks). */
int pcop_debug_counter(pcop_ctx* ctx, const char* name, uint64_t* value);
/* Test hook: the CPU-list parser behind pcop_bind_thread_near_device ("0-47,96-143"); returns the number of CPUs or PCOP_ERR_ARG. */
int pcop_debug_parse_cpulist(const char* txt, int32_t* cpus, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PCOP_H */
