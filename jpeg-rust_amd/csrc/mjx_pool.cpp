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
        } catch (...) {
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

extern "C" int mjx_pool_decode_batch(mjx_pool *pool, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *opts,
                                     unsigned threads_per_device, int *slot_of, uint8_t **rgb_dev, int *status,
                                     mjx_pool_result **out)
{
    if (!pool || !out || ((!jpegs || !lens) && n)) return MJX_ERR_INVALID_ARG;
    *out = nullptr;
    try {
        std::lock_guard<std::mutex> serial(pool->call_mu);
        const size_t N = pool->workers.size();
        mjx_pool_result *r = new mjx_pool_result;
        r->batches.assign(N, nullptr);
        r->slot_rc.assign(N, MJX_OK);
        r->slot_of.resize(n);
        r->index_in_slot.resize(n);
        // file i -> slot i mod N (north_star); the slot's list keeps the files in order
        std::vector<std::vector<const uint8_t *>> ptrs(N);
        std::vector<std::vector<size_t>> sizes(N), files(N);
        for (size_t i = 0; i < n; i++) {
            const size_t s = i % N;
            r->slot_of[i] = uint32_t(s);
            r->index_in_slot[i] = uint32_t(ptrs[s].size());
            ptrs[s].push_back(jpegs[i]);
            sizes[s].push_back(lens[i]);
            files[s].push_back(i);
        }
        std::vector<std::vector<int>> st(N);
        std::vector<std::vector<uint8_t *>> rgb(N);
        for (size_t s = 0; s < N; s++) {
            Worker *w = pool->workers[s];
            st[s].assign(ptrs[s].size(), MJX_OK);
            rgb[s].assign(ptrs[s].size(), nullptr);
            std::lock_guard<std::mutex> lk(w->mu);
            w->done = false;
            w->job = [&, s, w] {
                if (ptrs[s].empty()) return;
                r->slot_rc[s] = mjx_decode_batch(w->ctx, ptrs[s].data(), sizes[s].data(), ptrs[s].size(), opts, threads_per_device,
                                                 rgb[s].data(), st[s].data(), &r->batches[s]);
            };
            w->has_job = true;
            w->cv.notify_all();
        }
        for (size_t s = 0; s < N; s++) {                           // the host aggregates: wait for every queue
            Worker *w = pool->workers[s];
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->done; });
        }
        int rc = MJX_OK;
        for (size_t i = 0; i < n; i++) {
            const size_t s = r->slot_of[i], k = r->index_in_slot[i];
            const int slot_rc = r->slot_rc[s];
            if (slot_rc != MJX_OK) rc = slot_rc;
            if (slot_of) slot_of[i] = int(s);
            if (status) status[i] = slot_rc != MJX_OK ? slot_rc : st[s][k];
            if (rgb_dev) rgb_dev[i] = slot_rc != MJX_OK ? nullptr : rgb[s][k];
        }
        *out = r;
        return rc;
    } catch (...) {
        return MJX_ERR_NOMEM;
    }
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
