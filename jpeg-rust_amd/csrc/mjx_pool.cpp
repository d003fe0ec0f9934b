// mjx_pool.cpp -- multi-GPU front of the C ABI: per-GPU host thread + work queue, no collective.
//
// The reference decodes one file on one thread (src/jpeg/mod.rs:202-417, src/jpeg/decoder.rs:162-343 touch only `self`):
// pictures are independent, so a list of files shards over the GPUs of a node with no exchange step at all -- file i goes
// to device slot i mod N (BASELINE.json north_star: "per-GPU work queues, no RCCL").  A pool owns one mjx_ctx and one
// persistent host thread per slot; mjx_pool_decode_batch hands every thread its share of the list, each runs the
// pipelined mjx_decode_batch on its own device (its own parse threads, upload stream and decode streams), and the
// outputs stay on the device that produced them.  Only the C ABI of include/mjx.h is used here: this file is plain C++.
#include "mjx.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

struct Worker {
    mjx_ctx *ctx = nullptr;
    int device = 0;
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;       // one job at a time (the pool's callers are serialised)
    bool has_job = false, quit = false, done = false;
};

void worker_loop(Worker *w)
{
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->has_job || w->quit; });
            if (w->quit) return;
            job = std::move(w->job);
            w->has_job = false;
        }
        try {
            job();
        } catch (...) {          // (a job reports through its own state; one that throws has set nothing: see mjx_pool_decode_batch)
        }
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->done = true;
        }
        w->cv.notify_all();
    }
}

}   // namespace

struct mjx_pool {
    std::vector<Worker *> workers;
    std::mutex call_mu;
    int deal = MJX_POOL_DEAL_BY_BYTES;
    int fault_slot = -1;             // test knob (MJX_POOL_FAULT_SLOT=k): slot k's device "fails" -- its job returns MJX_ERR_DEVICE
};

struct mjx_pool_result {
    std::vector<mjx_batch *> batches;                 // per slot (null: the slot had no file, or its call failed)
    std::vector<int> slot_rc;                         // per slot: return code of its mjx_decode_batch
    std::vector<uint32_t> slot_of, index_in_slot;     // per file
};

extern "C" int mjx_pool_create(const int *devices, size_t n_devices, mjx_pool **out)
{
    if (!out || !devices || n_devices == 0 || n_devices > 64) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        mjx_pool *p = new mjx_pool;
        int rc = MJX_OK;
        for (size_t k = 0; k < n_devices && rc == MJX_OK; k++) {
            Worker *w = new Worker;
            w->device = devices[k];
            rc = mjx_ctx_create(devices[k], &w->ctx);
            if (rc != MJX_OK) { delete w; break; }
            w->thread = std::thread(worker_loop, w);
            p->workers.push_back(w);
        }
        if (rc != MJX_OK) { mjx_pool_destroy(p); return rc; }
        if (const char *e = std::getenv("MJX_POOL_FAULT_SLOT")) p->fault_slot = std::atoi(e);
        if (const char *e = std::getenv("MJX_POOL_DEAL")) p->deal = std::strcmp(e, "rr") == 0 ? MJX_POOL_DEAL_ROUND_ROBIN : MJX_POOL_DEAL_BY_BYTES;
        *out = p;
        return MJX_OK;
    } catch (...) {
        return MJX_ERR_NOMEM;
    }
}

extern "C" void mjx_pool_destroy(mjx_pool *pool)
{
    if (!pool) return;
    for (Worker *w : pool->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
        mjx_ctx_destroy(w->ctx);
        delete w;
    }
    delete pool;
}

extern "C" size_t mjx_pool_devices(const mjx_pool *pool) { return pool ? pool->workers.size() : 0; }

extern "C" int mjx_pool_device(const mjx_pool *pool, size_t slot)
{
    return (pool && slot < pool->workers.size()) ? pool->workers[slot]->device : -1;
}

extern "C" int mjx_pool_set_deal(mjx_pool *pool, int deal)
{
    if (!pool || (deal != MJX_POOL_DEAL_BY_BYTES && deal != MJX_POOL_DEAL_ROUND_ROBIN)) return MJX_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> serial(pool->call_mu);
    pool->deal = deal;
    return MJX_OK;
}

namespace {
// Everything the workers of one call touch lives here, on the heap, and the call does not leave before every worker it has
// posted a job to is done with it.
struct PoolCall {
    std::vector<std::vector<const uint8_t *>> ptrs;
    std::vector<std::vector<size_t>> sizes;
    std::vector<std::vector<int>> st;
    std::vector<std::vector<uint8_t *>> rgb;
    mjx_pool_result res;
};
}   // namespace

extern "C" int mjx_pool_decode_batch(mjx_pool *pool, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *opts,
                                     unsigned threads_per_device, int *slot_of, uint8_t **rgb_dev, int *status,
                                     mjx_pool_result **out)
{
    if (!pool || !out || ((!jpegs || !lens) && n)) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> serial(pool->call_mu);
    const size_t N = pool->workers.size();
    PoolCall *call = nullptr;
    size_t posted = 0;
    int rc = MJX_OK;
    try {
        call = new PoolCall;
        mjx_pool_result &r = call->res;
        r.batches.assign(N, nullptr);
        r.slot_rc.assign(N, MJX_OK);
        r.slot_of.resize(n);
        r.index_in_slot.resize(n);
        call->ptrs.resize(N);
        call->sizes.resize(N);
        // Dealing.  Pictures are independent, so any assignment is correct; what matters is that the slots
        // finish together.  By compressed bytes (default): file i goes to the slot that has been dealt the fewest bytes so
        // far, the lowest slot on a tie -- for a list of equal files that is i mod N, for a skewed one (large files at
        // every N-th place, a run of small ones) the queues stay level where round robin would not.  Round robin: i mod N.
        std::vector<uint64_t> load(N, 0);
        for (size_t i = 0; i < n; i++) {
            size_t s = i % N;
            if (pool->deal == MJX_POOL_DEAL_BY_BYTES) {
                s = 0;
                for (size_t k = 1; k < N; k++) if (load[k] < load[s]) s = k;
            }
            load[s] += uint64_t(lens[i]) + 4096;             // (+ a constant per file: lists of tiny files are dealt by count)
            r.slot_of[i] = uint32_t(s);
            r.index_in_slot[i] = uint32_t(call->ptrs[s].size());
            call->ptrs[s].push_back(jpegs[i]);
            call->sizes[s].push_back(lens[i]);
        }
        call->st.resize(N);
        call->rgb.resize(N);
        for (size_t s = 0; s < N; s++) {
            call->st[s].assign(call->ptrs[s].size(), MJX_OK);
            call->rgb[s].assign(call->ptrs[s].size(), nullptr);
        }
        // every allocation is behind us: hand out the jobs (one per slot that has files; a std::function of this size
        // allocates nothing)
        const int fault_slot = pool->fault_slot;
        for (size_t s = 0; s < N; s++) {
            Worker *w = pool->workers[s];
            if (call->ptrs[s].empty()) continue;
            r.slot_rc[s] = MJX_ERR_DEVICE;                   // (stays, should the job die before it has a return code)
            std::lock_guard<std::mutex> lk(w->mu);
            w->done = false;
            w->job = [call, s, w, opts, threads_per_device, fault_slot] {
                if (int(s) == fault_slot) return;            // this slot's device has "failed": MJX_ERR_DEVICE stands
                call->res.slot_rc[s] = mjx_decode_batch(w->ctx, call->ptrs[s].data(), call->sizes[s].data(), call->ptrs[s].size(), opts,
                                                        threads_per_device, call->rgb[s].data(), call->st[s].data(), &call->res.batches[s]);
            };
            w->has_job = true;
            w->cv.notify_all();
            posted |= size_t(1) << s;
        }
    } catch (...) {
        rc = MJX_ERR_NOMEM;
    }
    for (size_t s = 0; s < N; s++) {                               // the host aggregates: wait for every queue that got a job
        if (!((posted >> s) & 1)) continue;
        Worker *w = pool->workers[s];
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done; });
    }
    if (rc != MJX_OK || !call) {
        if (call) for (mjx_batch *b : call->res.batches) mjx_batch_free(b);
        delete call;
        return rc != MJX_OK ? rc : MJX_ERR_NOMEM;
    }
    // A slot that failed fails its own files only: the others keep their results, the call returns the failure.
    mjx_pool_result *r = new (std::nothrow) mjx_pool_result(std::move(call->res));
    if (!r) {
        for (mjx_batch *b : call->res.batches) mjx_batch_free(b);
        delete call;
        return MJX_ERR_NOMEM;
    }
    for (size_t i = 0; i < n; i++) {
        const size_t s = r->slot_of[i], k = r->index_in_slot[i];
        const int slot_rc = r->slot_rc[s];
        if (slot_rc != MJX_OK) rc = slot_rc;
        if (slot_of) slot_of[i] = int(s);
        if (status) status[i] = slot_rc != MJX_OK ? slot_rc : call->st[s][k];
        if (rgb_dev) rgb_dev[i] = slot_rc != MJX_OK ? nullptr : call->rgb[s][k];
    }
    delete call;
    *out = r;
    return rc;
}

extern "C" int mjx_pool_result_locate(const mjx_pool_result *r, size_t i, size_t *slot, mjx_batch **batch, size_t *index)
{
    if (!r || i >= r->slot_of.size()) return MJX_ERR_INVALID_ARG;
    const size_t s = r->slot_of[i];
    if (slot) *slot = s;
    if (batch) *batch = r->batches[s];
    if (index) *index = r->index_in_slot[i];
    return r->batches[s] ? MJX_OK : (r->slot_rc[s] != MJX_OK ? r->slot_rc[s] : MJX_ERR_INVALID_ARG);
}

extern "C" void mjx_pool_result_free(mjx_pool_result *r)
{
    if (!r) return;
    for (mjx_batch *b : r->batches) mjx_batch_free(b);
    delete r;
}
