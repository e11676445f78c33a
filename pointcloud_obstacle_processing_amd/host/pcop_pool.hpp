// pcop_pool.hpp — frames over the GPUs of one node, in C++ (the reference has no multi-GPU path:
// od.cpp:1014 is one ros::spin() thread).
//
// The hot path is embarrassingly frame-parallel (a frame touches no other frame's data: od.cpp:699-797, the accumulator
// is cleared at od.cpp:926), so the pool shards by frame: frame f goes to GPU f mod G.  Per GPU there is one FEEDER
// THREAD that owns everything of that GPU - TWO pcop contexts (created with max_frames = batch), its pinned staging
// buffers (allocated by the thread itself, so that first touch puts them on the thread's NUMA node) and its queue.
// The thread queues batch k + 1 on one context (host-to-device copies and kernels are asynchronous) before it collects
// batch k from the other: the copy of one batch runs under the kernels of the other, which is what keeps a GPU fed from
// host memory.  A caller fills a staging slot IN PLACE (acquire / commit: no copy on the caller's thread; submit() is the
// convenience form with one).
// The only bytes that cross GPUs are the 32-byte pcop_params: rank 0's struct is broadcast with ncclBroadcast (RCCL)
// when the pool starts; there is no data-path collective.  Results come back in frame order.
// Start-up is all-or-nothing: every worker finishes the steps that can fail (device, staging, stream, contexts) and the
// workers agree on the outcome BEFORE anyone enters the collective, so a worker that fails cannot leave the others
// waiting in a broadcast it never joins.
//
// `same_gpu` maps every worker to one device (a one-GPU box rehearsing G > 1): RCCL cannot put two ranks of one
// process on one device, so that mode hands the struct over in host memory instead and says so in broadcast_kind().
#pragma once

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pcop.h"

namespace pcop_host {

struct PoolResult {
    uint64_t frame = 0;
    int gpu = -1;                       // worker that processed it
    int status = 0;                     // pcop_status
    std::string error;
    int passthrough = 0;
    int32_t num_voxels = 0;
    std::vector<int32_t> offsets;       // K + 1, PCL's order
    std::vector<int32_t> indices;
    std::vector<float> voxel_xyz;       // 3 floats per voxel (AOS12) when the params ask for the cloud
};

class GpuPool {
public:
    // gpus workers; worker g runs on HIP device (same_gpu ? 0 : g).  params: what rank 0 holds; every other worker starts
    // from zeros and receives it.  batch: frames a worker puts through the launches of one pcop_submit_batch.
    GpuPool(int gpus, size_t max_points, int batch, const pcop_params& params, bool same_gpu = false);
    ~GpuPool();
    GpuPool(const GpuPool&) = delete;
    GpuPool& operator=(const GpuPool&) = delete;

    bool ok() const { return error_.empty(); }
    const std::string& error() const { return error_; }
    int gpus() const { return (int)workers_.size(); }
    const char* broadcast_kind() const { return bcast_; }
    pcop_params params_of(int worker) const;          // what the worker ended up with (tests)

    // One frame, filled in place: acquire() hands out the pinned staging slot of the NEXT frame (room for max_points points;
    // blocks while that frame's worker has none free), the caller writes XYZ SoA planes x[n] y[n] z[n] into it, commit(n) queues
    // it.  Frames are numbered 0, 1, 2, ... in commit order; frame f goes to worker f mod G.  One acquire per commit.
    float* acquire();
    bool commit(size_t n);
    // The same with a copy: n points, planes in the caller's own memory.
    bool submit(const float* xyz_soa, size_t n);
    // The next result in frame order (blocks until it is there); false once every submitted frame has been handed out.
    bool next(PoolResult* out);

private:
    struct Job { uint64_t frame; size_t n; int slot; };
    struct Worker {
        int index = 0, device = 0;
        std::thread thread;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Job> queue;
        std::vector<int> free_slots;
        std::vector<float*> staging;          // pinned, 3 * max_points floats each
        pcop_params params{};
        bool stop = false, ready = false, failed = false;
        std::string error;
    };
    // start-up agreement: workers through their fallible set-up, and how many of them failed
    std::mutex smu_;
    std::condition_variable scv_;
    int setup_done_ = 0, setup_failed_ = 0;
    int acquired_slot_ = -1;                  // the slot acquire() handed out for frame submitted_
    void run(Worker* w);
    void deliver(PoolResult&& r);

    std::vector<Worker*> workers_;
    size_t max_points_;
    int batch_;
    pcop_params root_params_;
    bool same_gpu_;
    const char* bcast_ = "none";
    void* comms_ = nullptr;                   // ncclComm_t[G]
    std::string error_;
    uint64_t submitted_ = 0, handed_out_ = 0;
    std::mutex rmu_;
    std::condition_variable rcv_;
    std::map<uint64_t, PoolResult> results_;
};

}  // namespace pcop_host
