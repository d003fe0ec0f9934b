// mjx_pool.cpp -- multi-GPU front of the C ABI: per-GPU host thread + work queue, no collective.
//
// The reference decodes one file on one thread (src/jpeg/mod.rs:202-417, src/jpeg/decoder.rs:162-343 touch only `self`):
// pictures are independent, so a list of files shards over the GPUs of a node with no exchange step at all (BASELINE.json
// north_star: "per-GPU work queues, no RCCL").  A pool owns one mjx_ctx and one persistent host thread per slot;
// mjx_pool_decode_batch hands every thread its share of the list, each runs the pipelined mjx_decode_batch on its own
// device (its own parse threads, upload stream and decode streams), and the outputs stay on the device that produced them.
// The slots share one host: the parse threads are budgeted over the slots, and a slot's host thread is bound to its GPU's
// NUMA node where the platform names one (see slot_affinity).  Only the C ABI of include/mjx.h is used here: plain C++.
#include "mjx.h"

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Worker {
    mjx_ctx *ctx = nullptr;
    int device = 0;
    bool ready = false;              // the thread has placed itself (slot_affinity); under mu
    std::atomic<int> numa_node{-1};  // node the thread is bound to (-1: not bound); set by the thread when it starts
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;       // one job at a time (the pool's callers are serialised)
    bool has_job = false, quit = false, done = false;
};

// The processors of NUMA node `node` that this process may run on (empty: unknown node, or none of them allowed).
// /sys/devices/system/node/node<k>/cpulist is a list like "0-31,128-159".
bool node_cpus(int node, cpu_set_t *out)
{
    CPU_ZERO(out);
    if (node < 0) return false;
    const std::string path = "/sys/devices/system/node/node" + std::to_string(node) + "/cpulist";
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) { std::fclose(f); return false; }
    int a = 0, b = 0, any = 0;
    for (;;) {
        if (std::fscanf(f, "%d", &a) != 1) break;
        b = a;
        int c = std::fgetc(f);
        if (c == '-') { if (std::fscanf(f, "%d", &b) != 1) break; c = std::fgetc(f); }
        for (int k = a; k <= b && k < CPU_SETSIZE; k++)
            if (k >= 0 && CPU_ISSET(k, &allowed)) { CPU_SET(k, out); any++; }
        if (c != ',') break;
    }
    std::fclose(f);
    return any > 0;
}

// A slot's host thread runs mjx_decode_batch for its GPU: it starts the slot's parse threads (they inherit its affinity)
// and makes the context's first pinned allocations (first touch: the pages land on the node the thread runs on).  Bound to the
// processors of the GPU's NUMA node, the de-stuffed scans sit in memory next to the PCIe root the DMA engine reads them
// through, and eight slots do not crowd onto one socket.  Only when the node is known, the process may run there, and the
// node has room for the slot's threads.
void slot_affinity(Worker *w, unsigned threads_wanted)
{
    const char *e = std::getenv("MJX_POOL_NUMA");
    if (e && std::atoi(e) == 0) return;
    const int node = mjx_ctx_numa_node(w->ctx);
    cpu_set_t set;
    if (!node_cpus(node, &set) || unsigned(CPU_COUNT(&set)) < std::max(1u, threads_wanted)) return;
    if (sched_setaffinity(0, sizeof set, &set) == 0) w->numa_node = node;
}

void worker_loop(Worker *w, unsigned threads_wanted)
{
    slot_affinity(w, threads_wanted);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->ready = true;                 // (mjx_pool_create waits for it: a call then finds every slot where it will stay)
    }
    w->cv.notify_all();
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
    std::vector<unsigned> slot_threads;               // per slot: parse threads of its call (0: no file)
    std::vector<int> slot_node;                       // per slot: NUMA node its host thread is bound to (-1: not bound)
    std::vector<double> slot_ms;                      // per slot: wall clock of its mjx_decode_batch (0: no file)
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
            w->thread = std::thread(worker_loop, w, std::max(2u, mjx_host_processors() / unsigned(2 * n_devices)));
            p->workers.push_back(w);
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->ready; });
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
    std::vector<char> posted;
    int rc = MJX_OK;
    // The slots share the host: with no count given each takes its share of the processors the process may use -- half of them
    // parse in all, as in a single mjx_decode_batch, at least two per slot -- instead of every slot sizing itself as if it
    // were alone (eight slots on a 16-processor quota would start 64 parse threads).
    const unsigned threads = threads_per_device ? threads_per_device : std::max(2u, mjx_host_processors() / unsigned(2 * N));
    try {
        posted.assign(N, 0);
        call = new PoolCall;
        mjx_pool_result &r = call->res;
        r.batches.assign(N, nullptr);
        r.slot_rc.assign(N, MJX_OK);
        r.slot_threads.assign(N, 0);
        r.slot_node.assign(N, -1);
        r.slot_ms.assign(N, 0.0);
        r.slot_of.resize(n);
        r.index_in_slot.resize(n);
        call->ptrs.resize(N);
        call->sizes.resize(N);
        // Dealing.  Pictures are independent, so any assignment is correct; what matters is that the slots
        // finish together.  By compressed bytes (default): the files are taken largest first (a stable order: equal files keep
        // their list order) and each goes to the slot that has been dealt the fewest bytes so far, the lowest slot on a tie
        // -- longest-processing-time-first.  For a list of equal files that is i mod N; for a skewed one (a few large files
        // anywhere in a run of small ones) the queues end level, which dealing in list order does not promise.  Round robin:
        // i mod N.
        std::vector<size_t> order(n);
        std::iota(order.begin(), order.end(), size_t(0));
        if (pool->deal == MJX_POOL_DEAL_BY_BYTES)
            std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return lens[a] > lens[b]; });
        std::vector<uint64_t> load(N, 0);
        for (size_t j = 0; j < n; j++) {
            const size_t i = order[j];
            size_t s = i % N;
            if (pool->deal == MJX_POOL_DEAL_BY_BYTES) {
                s = 0;
                for (size_t k = 1; k < N; k++) if (load[k] < load[s]) s = k;
            }
            load[s] += uint64_t(lens[i]) + 4096;             // (+ a constant per file: lists of tiny files are dealt by count)
            r.slot_of[i] = uint32_t(s);
        }
        for (size_t i = 0; i < n; i++) {                     // (inside a slot the files keep their list order)
            const size_t s = r.slot_of[i];
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
        // hand out the jobs, one per slot that has files.  (Assigning the job can allocate -- its captures are larger than
        // std::function's inline buffer -- and so can throw: a slot counts as posted, and its `done` is cleared, only once its
        // job is in place; the wait below covers exactly the posted ones.)
        const int fault_slot = pool->fault_slot;
        for (size_t s = 0; s < N; s++) {
            Worker *w = pool->workers[s];
            if (call->ptrs[s].empty()) continue;
            r.slot_rc[s] = MJX_ERR_DEVICE;                   // (stays, should the job die before it has a return code)
            r.slot_threads[s] = threads;
            r.slot_node[s] = w->numa_node;
            std::lock_guard<std::mutex> lk(w->mu);
            w->job = [call, s, w, opts, threads, fault_slot] {
                if (int(s) == fault_slot) return;            // this slot's device has "failed": MJX_ERR_DEVICE stands
                const auto t0 = std::chrono::steady_clock::now();
                call->res.slot_rc[s] = mjx_decode_batch(w->ctx, call->ptrs[s].data(), call->sizes[s].data(), call->ptrs[s].size(), opts,
                                                        threads, call->rgb[s].data(), call->st[s].data(), &call->res.batches[s]);
                call->res.slot_ms[s] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            };
            w->done = false;
            w->has_job = true;
            w->cv.notify_all();
            posted[s] = 1;
        }
    } catch (...) {
        rc = MJX_ERR_NOMEM;
    }
    for (size_t s = 0; s < N; s++) {                               // the host aggregates: wait for every queue that got a job
        if (s >= posted.size() || !posted[s]) continue;
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

extern "C" int mjx_pool_result_host(const mjx_pool_result *r, size_t slot, unsigned *threads, int *numa_node)
{
    if (!r || slot >= r->slot_threads.size()) return MJX_ERR_INVALID_ARG;
    if (threads) *threads = r->slot_threads[slot];
    if (numa_node) *numa_node = r->slot_node[slot];
    return MJX_OK;
}

extern "C" int mjx_pool_result_slot_ms(const mjx_pool_result *r, size_t slot, double *ms)
{
    if (!r || slot >= r->slot_ms.size() || !ms) return MJX_ERR_INVALID_ARG;
    *ms = r->slot_ms[slot];
    return MJX_OK;
}

extern "C" void mjx_pool_result_free(mjx_pool_result *r)
{
    if (!r) return;
    for (mjx_batch *b : r->batches) mjx_batch_free(b);
    delete r;
}
