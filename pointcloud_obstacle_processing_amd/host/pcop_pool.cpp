// pcop_pool.cpp — see pcop_pool.hpp.
#include "pcop_pool.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

namespace pcop_host {

namespace {
std::string hip_msg(const char* what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); }
}  // namespace

GpuPool::GpuPool(int gpus, size_t max_points, int batch, const pcop_params& params, bool same_gpu)
    : max_points_(max_points), batch_(std::max(1, batch)), root_params_(params), same_gpu_(same_gpu) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { error_ = "no HIP device (the pool has no CPU path)"; return; }
    if (gpus < 1) { error_ = "gpus must be >= 1"; return; }
    if (!same_gpu && gpus > ndev) { error_ = "more workers than visible GPUs (same_gpu maps them all to device 0)"; return; }
    // the one collective of the design: rank 0's 32 bytes to everyone (RCCL); G ranks of one process, one per device
    std::vector<int> devs(gpus);
    for (int g = 0; g < gpus; ++g) devs[g] = same_gpu ? 0 : g;
    if (!same_gpu || gpus == 1) {
        ncclComm_t* comms = new ncclComm_t[gpus];
        const ncclResult_t r = ncclCommInitAll(comms, gpus, devs.data());
        if (r != ncclSuccess) { error_ = std::string("ncclCommInitAll: ") + ncclGetErrorString(r); delete[] comms; return; }
        comms_ = comms;
        bcast_ = "ncclBroadcast (RCCL), 32 bytes from rank 0";
    } else {
        bcast_ = "host memory (same_gpu rehearsal: RCCL cannot place two ranks of one process on one device)";
    }
    for (int g = 0; g < gpus; ++g) {
        Worker* w = new Worker();
        w->index = g; w->device = devs[g];
        if (g == 0 || comms_ == nullptr) w->params = params;   // (without RCCL the struct travels in host memory)
        workers_.push_back(w);
    }
    for (Worker* w : workers_) w->thread = std::thread(&GpuPool::run, this, w);
    for (Worker* w : workers_) {   // wait until every feeder has its context, its staging and its parameters
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w] { return w->ready || w->failed; });
        if (w->failed && error_.empty()) error_ = "worker " + std::to_string(w->index) + ": " + w->error;
    }
}

GpuPool::~GpuPool() {
    for (Worker* w : workers_) {
        { std::lock_guard<std::mutex> lk(w->mu); w->stop = true; }
        w->cv.notify_all();
    }
    for (Worker* w : workers_) if (w->thread.joinable()) w->thread.join();
    if (comms_) {
        ncclComm_t* comms = static_cast<ncclComm_t*>(comms_);
        for (size_t g = 0; g < workers_.size(); ++g) ncclCommDestroy(comms[g]);
        delete[] comms;
    }
    for (Worker* w : workers_) delete w;
}

pcop_params GpuPool::params_of(int worker) const { return workers_[(size_t)worker]->params; }

bool GpuPool::submit(const float* xyz_soa, size_t n) {
    if (!ok() || n > max_points_ || (!xyz_soa && n)) return false;
    Worker* w = workers_[submitted_ % workers_.size()];
    int slot;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w] { return !w->free_slots.empty() || w->failed; });
        if (w->failed) return false;
        slot = w->free_slots.back();
        w->free_slots.pop_back();
    }
    // planes x[n] y[n] z[n], packed: what pcop_submit_batch reads as PCOP_LAYOUT_SOA
    if (n) std::memcpy(w->staging[(size_t)slot], xyz_soa, 3 * n * sizeof(float));
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->queue.push_back(Job{submitted_, n, slot});
    }
    w->cv.notify_all();
    ++submitted_;
    return true;
}

void GpuPool::deliver(PoolResult&& r) {
    { std::lock_guard<std::mutex> lk(rmu_); results_.emplace(r.frame, std::move(r)); }
    rcv_.notify_all();
}

bool GpuPool::next(PoolResult* out) {
    if (handed_out_ >= submitted_) return false;
    std::unique_lock<std::mutex> lk(rmu_);
    rcv_.wait(lk, [this] { return results_.count(handed_out_) != 0; });
    auto it = results_.find(handed_out_);
    *out = std::move(it->second);
    results_.erase(it);
    ++handed_out_;
    return true;
}

void GpuPool::run(Worker* w) {
    auto fail = [&](const std::string& msg) {
        { std::lock_guard<std::mutex> lk(w->mu); w->failed = true; w->error = msg; }
        w->cv.notify_all();
    };
    hipError_t he = hipSetDevice(w->device);
    if (he != hipSuccess) return fail(hip_msg("hipSetDevice", he));
    // pinned staging, allocated (and touched) by the feeder thread itself: two batches' worth, so that the caller fills the
    // next batch while this one is on the GPU
    const int nslots = 2 * batch_;
    for (int s = 0; s < nslots; ++s) {
        float* p = nullptr;
        he = hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(3 * max_points_, 1) * sizeof(float), hipHostMallocDefault);
        if (he != hipSuccess) return fail(hip_msg("hipHostMalloc", he));
        std::memset(p, 0, std::max<size_t>(3 * max_points_, 1) * sizeof(float));
        w->staging.push_back(p);
    }
    // parameters: rank 0 holds them, everyone receives the 32-byte struct
    if (comms_) {
        ncclComm_t comm = static_cast<ncclComm_t*>(comms_)[w->index];
        void* dbuf = nullptr;
        hipStream_t st = nullptr;
        if ((he = hipMalloc(&dbuf, sizeof(pcop_params))) != hipSuccess) return fail(hip_msg("hipMalloc", he));
        if ((he = hipStreamCreate(&st)) != hipSuccess) return fail(hip_msg("hipStreamCreate", he));
        if ((he = hipMemcpyAsync(dbuf, &w->params, sizeof(pcop_params), hipMemcpyHostToDevice, st)) != hipSuccess) return fail(hip_msg("hipMemcpyAsync", he));
        const ncclResult_t r = ncclBroadcast(dbuf, dbuf, sizeof(pcop_params), ncclChar, 0, comm, st);
        if (r != ncclSuccess) return fail(std::string("ncclBroadcast: ") + ncclGetErrorString(r));
        if ((he = hipMemcpyAsync(&w->params, dbuf, sizeof(pcop_params), hipMemcpyDeviceToHost, st)) != hipSuccess) return fail(hip_msg("hipMemcpyAsync", he));
        if ((he = hipStreamSynchronize(st)) != hipSuccess) return fail(hip_msg("hipStreamSynchronize", he));
        (void)hipStreamDestroy(st);
        (void)hipFree(dbuf);
    }
    pcop_ctx* ctx = nullptr;
    int rc = pcop_create_batch(&ctx, w->device, max_points_, batch_);
    if (rc) return fail(std::string("pcop_create_batch: ") + pcop_status_string(rc));
    if ((rc = pcop_set_params(ctx, &w->params))) { pcop_destroy(ctx); return fail(std::string("pcop_set_params: ") + pcop_status_string(rc)); }
    {
        std::lock_guard<std::mutex> lk(w->mu);
        for (int s = 0; s < nslots; ++s) w->free_slots.push_back(s);
        w->ready = true;
    }
    w->cv.notify_all();

    std::vector<Job> jobs;
    std::vector<const float*> ptrs((size_t)batch_);
    std::vector<size_t> counts((size_t)batch_);
    std::vector<pcop_result> res((size_t)batch_);
    for (;;) {
        jobs.clear();
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return !w->queue.empty() || w->stop; });
            if (w->queue.empty() && w->stop) break;
            while (!w->queue.empty() && (int)jobs.size() < batch_) { jobs.push_back(w->queue.front()); w->queue.pop_front(); }
        }
        const int nf = (int)jobs.size();
        for (int f = 0; f < nf; ++f) { ptrs[(size_t)f] = w->staging[(size_t)jobs[(size_t)f].slot]; counts[(size_t)f] = jobs[(size_t)f].n; }
        rc = pcop_submit_batch(ctx, ptrs.data(), PCOP_LAYOUT_SOA, counts.data(), nf, 1, PCOP_MEM_HOST);
        if (!rc) rc = pcop_wait_batch(ctx, res.data(), nf);
        for (int f = 0; f < nf; ++f) {
            PoolResult r;
            r.frame = jobs[(size_t)f].frame; r.gpu = w->index; r.status = rc;
            if (rc) {
                r.error = pcop_last_error(ctx);
            } else {
                const pcop_result& q = res[(size_t)f];
                r.passthrough = q.passthrough; r.num_voxels = q.num_voxels;
                r.offsets.assign(q.offsets, q.offsets + q.num_clusters + 1);
                r.indices.assign(q.indices, q.indices + q.num_indices);
                if (q.voxel_xyz && q.num_voxels) r.voxel_xyz.assign(q.voxel_xyz, q.voxel_xyz + 3 * (size_t)q.num_voxels);
            }
            deliver(std::move(r));
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            for (int f = 0; f < nf; ++f) w->free_slots.push_back(jobs[(size_t)f].slot);
        }
        w->cv.notify_all();
    }
    pcop_destroy(ctx);
    for (float* p : w->staging) (void)hipHostFree(p);
}

}  // namespace pcop_host
