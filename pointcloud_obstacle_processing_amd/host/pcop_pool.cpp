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

float* GpuPool::acquire() {
    if (!ok()) return nullptr;
    Worker* w = workers_[submitted_ % workers_.size()];
    if (acquired_slot_ >= 0) return w->staging[(size_t)acquired_slot_];   // (asked twice for the same frame)
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [w] { return !w->free_slots.empty() || w->failed; });
    if (w->failed) return nullptr;
    acquired_slot_ = w->free_slots.back();
    w->free_slots.pop_back();
    return w->staging[(size_t)acquired_slot_];
}

bool GpuPool::commit(size_t n) {
    if (!ok() || acquired_slot_ < 0 || n > max_points_) return false;
    Worker* w = workers_[submitted_ % workers_.size()];
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->queue.push_back(Job{submitted_, n, acquired_slot_});
    }
    w->cv.notify_all();
    acquired_slot_ = -1;
    ++submitted_;
    return true;
}

bool GpuPool::submit(const float* xyz_soa, size_t n) {
    if (!ok() || n > max_points_ || (!xyz_soa && n)) return false;
    float* dst = acquire();
    if (!dst) return false;
    // planes x[n] y[n] z[n], packed: what pcop_submit_batch reads as PCOP_LAYOUT_SOA
    if (n) std::memcpy(dst, xyz_soa, 3 * n * sizeof(float));
    return commit(n);
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
    // ---- everything that can fail, before anyone enters the collective ------------------------------------------------------
    std::string setup_error;
    pcop_ctx* ctx[2] = {nullptr, nullptr};
    void* dbuf = nullptr;
    hipStream_t st = nullptr;
    // pinned staging, allocated (and touched) by the feeder thread itself: three batches' worth - one the caller fills, one on
    // its way to the GPU, one being worked on
    const int nslots = 3 * batch_;
    {
        pcop_bind_thread_near_device(w->device);   // the staging slots and the contexts' pinned buffers on the GPU's NUMA node (0 = stays where it is)
        hipError_t he = hipSetDevice(w->device);
        if (he != hipSuccess) setup_error = hip_msg("hipSetDevice", he);
        for (int s = 0; s < nslots && setup_error.empty(); ++s) {
            float* p = nullptr;
            he = hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(3 * max_points_, 1) * sizeof(float), hipHostMallocDefault);
            if (he != hipSuccess) { setup_error = hip_msg("hipHostMalloc", he); break; }
            std::memset(p, 0, std::max<size_t>(3 * max_points_, 1) * sizeof(float));
            w->staging.push_back(p);
        }
        if (setup_error.empty() && comms_) {
            if ((he = hipMalloc(&dbuf, sizeof(pcop_params))) != hipSuccess) setup_error = hip_msg("hipMalloc", he);
            else if ((he = hipStreamCreate(&st)) != hipSuccess) setup_error = hip_msg("hipStreamCreate", he);
        }
        for (int k = 0; k < 2 && setup_error.empty(); ++k) {
            const int rc = pcop_create_batch(&ctx[k], w->device, max_points_, batch_);
            if (rc) setup_error = std::string("pcop_create_batch: ") + pcop_status_string(rc);
        }
    }
    bool all_ok;
    {   // the workers agree: did everyone get this far?
        std::unique_lock<std::mutex> lk(smu_);
        ++setup_done_;
        if (!setup_error.empty()) ++setup_failed_;
        scv_.notify_all();
        scv_.wait(lk, [this] { return setup_done_ == (int)workers_.size(); });
        all_ok = setup_failed_ == 0;
    }
    auto release = [&]() {
        for (int k = 0; k < 2; ++k) if (ctx[k]) { pcop_destroy(ctx[k]); ctx[k] = nullptr; }
        if (st) (void)hipStreamDestroy(st);
        if (dbuf) (void)hipFree(dbuf);
        for (float* p : w->staging) (void)hipHostFree(p);
        w->staging.clear();
    };
    if (!all_ok) {   // nobody broadcasts: the pool reports the first failure and has no workers
        release();
        return fail(setup_error.empty() ? "another worker failed to start" : setup_error);
    }
    // ---- parameters: rank 0 holds them, everyone receives the 32-byte struct ---------------------------------------------------
    if (comms_) {
        ncclComm_t comm = static_cast<ncclComm_t*>(comms_)[w->index];
        hipError_t he;
        std::string e;
        if ((he = hipMemcpyAsync(dbuf, &w->params, sizeof(pcop_params), hipMemcpyHostToDevice, st)) != hipSuccess) e = hip_msg("hipMemcpyAsync", he);
        // (every rank enters the broadcast, whatever its copy did: a rank that stays out would leave the others waiting)
        const ncclResult_t r = ncclBroadcast(dbuf, dbuf, sizeof(pcop_params), ncclChar, 0, comm, st);
        if (r != ncclSuccess && e.empty()) e = std::string("ncclBroadcast: ") + ncclGetErrorString(r);
        if (e.empty() && (he = hipMemcpyAsync(&w->params, dbuf, sizeof(pcop_params), hipMemcpyDeviceToHost, st)) != hipSuccess) e = hip_msg("hipMemcpyAsync", he);
        if ((he = hipStreamSynchronize(st)) != hipSuccess && e.empty()) e = hip_msg("hipStreamSynchronize", he);
        if (!e.empty()) { release(); return fail(e); }
    }
    for (int k = 0; k < 2; ++k) {
        const int rc = pcop_set_params(ctx[k], &w->params);
        if (rc) { release(); return fail(std::string("pcop_set_params: ") + pcop_status_string(rc)); }
    }
    {
        std::lock_guard<std::mutex> lk(w->mu);
        for (int s = 0; s < nslots; ++s) w->free_slots.push_back(s);
        w->ready = true;
    }
    w->cv.notify_all();

    // ---- the feeder loop: batch k + 1 is queued on one context before batch k is collected from the other --------------------
    struct InFlight { std::vector<Job> jobs; int rc = 0; bool active = false; };
    InFlight fl[2];
    std::vector<const float*> ptrs((size_t)batch_);
    std::vector<size_t> counts((size_t)batch_);
    std::vector<pcop_result> res((size_t)batch_);
    auto collect = [&](int k) {
        InFlight& q = fl[k];
        const int nf = (int)q.jobs.size();
        int rc = q.rc;
        if (!rc) rc = pcop_wait_batch(ctx[k], res.data(), nf);
        for (int f = 0; f < nf; ++f) {
            PoolResult r;
            r.frame = q.jobs[(size_t)f].frame; r.gpu = w->index; r.status = rc;
            if (rc) {
                r.error = pcop_last_error(ctx[k]);
            } else {
                const pcop_result& o = res[(size_t)f];
                r.passthrough = o.passthrough; r.num_voxels = o.num_voxels;
                r.offsets.assign(o.offsets, o.offsets + o.num_clusters + 1);
                r.indices.assign(o.indices, o.indices + o.num_indices);
                if (o.voxel_xyz && o.num_voxels) r.voxel_xyz.assign(o.voxel_xyz, o.voxel_xyz + 3 * (size_t)o.num_voxels);
            }
            deliver(std::move(r));
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            for (int f = 0; f < nf; ++f) w->free_slots.push_back(q.jobs[(size_t)f].slot);
        }
        w->cv.notify_all();
        q.jobs.clear(); q.active = false;
    };
    int cur = 0;
    for (;;) {
        std::vector<Job> jobs;
        bool stop = false;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            // with a batch in flight the thread does not sleep on an empty queue: it goes and collects
            w->cv.wait(lk, [&] { return !w->queue.empty() || w->stop || fl[0].active || fl[1].active; });
            // a full batch is taken at once; a partial one only when the GPU would otherwise idle (nothing in flight) or the pool is
            // being drained - with a batch in flight the thread collects that first and lets the caller fill more slots meanwhile
            const bool idle = !fl[0].active && !fl[1].active;
            if ((int)w->queue.size() >= batch_ || idle || w->stop)
                while (!w->queue.empty() && (int)jobs.size() < batch_) { jobs.push_back(w->queue.front()); w->queue.pop_front(); }
            stop = w->stop && jobs.empty();
        }
        if (!jobs.empty()) {
            if (fl[cur].active) collect(cur);   // (its turn came round again)
            const int nf = (int)jobs.size();
            for (int f = 0; f < nf; ++f) { ptrs[(size_t)f] = w->staging[(size_t)jobs[(size_t)f].slot]; counts[(size_t)f] = jobs[(size_t)f].n; }
            fl[cur].rc = pcop_submit_batch(ctx[cur], ptrs.data(), PCOP_LAYOUT_SOA, counts.data(), nf, 1, PCOP_MEM_HOST);
            fl[cur].jobs = std::move(jobs); fl[cur].active = true;
            cur ^= 1;
            if (fl[cur].active) collect(cur);   // the older batch, while the new one's copies and kernels run
        } else if (fl[cur].active) collect(cur);
        else if (fl[cur ^ 1].active) collect(cur ^ 1);
        else if (stop) break;
    }
    release();
}

}  // namespace pcop_host
