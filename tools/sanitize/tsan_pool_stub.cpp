// mjx_pool.cpp (the multi-GPU front: one host thread + work queue per device slot) under ThreadSanitizer, against
// stubs of the single-device C ABI it is written on -- no HIP, no GPU: what is checked is the pool's own hand-over of jobs and
// results between the caller's thread and the slots' threads.  Scenarios: eight slots sharing two "devices", lists of 1 / 7 / 4096
// files, an empty list, a slot whose device fails (MJX_POOL_FAULT_SLOT), dealing by bytes and round robin, calls from two
// threads at once (the pool serialises them), destroy right after a call.  Built and run by tests/test_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -Iinclude tools/sanitize/tsan_pool_stub.cpp jpeg-rust_amd/csrc/mjx_pool.cpp -pthread
#include "mjx.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

// ---- stubs of the single-device ABI ------------------------------------------------------------------------------------
struct mjx_ctx { int device; std::atomic<int> calls{0}; };
struct mjx_batch { std::vector<uint8_t> rgb; size_t n; };
static std::atomic<int> g_live_batches{0};

extern "C" int mjx_ctx_create(int device, mjx_ctx **out) { *out = new mjx_ctx; (*out)->device = device; return MJX_OK; }
extern "C" void mjx_ctx_destroy(mjx_ctx *ctx) { delete ctx; }
extern "C" int mjx_ctx_numa_node(const mjx_ctx *) { return -1; }
extern "C" unsigned mjx_host_processors(void) { return 8; }
extern "C" void mjx_batch_free(mjx_batch *b) { if (b) { g_live_batches--; delete b; } }
extern "C" int mjx_decode_batch(mjx_ctx *ctx, const uint8_t *const *jpegs, const size_t *lens, size_t n, const mjx_opts *, unsigned,
                                uint8_t **rgb_dev, int *status, mjx_batch **out)
{
    ctx->calls++;
    mjx_batch *b = new mjx_batch;
    b->n = n;
    b->rgb.assign(n ? n : 1, 0);
    g_live_batches++;
    for (size_t i = 0; i < n; i++) {                       // "decode": read the file's first byte, as a slot's parse threads would
        b->rgb[i] = lens[i] ? jpegs[i][0] : 0;
        status[i] = lens[i] >= 2 && jpegs[i][0] == 0xff ? MJX_OK : MJX_ERR_TRUNCATED;
        rgb_dev[i] = status[i] == MJX_OK ? &b->rgb[i] : nullptr;
    }
    *out = b;
    return MJX_OK;
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); fails++; } } while (0)

static void one_call(mjx_pool *pool, size_t n, int expect_fault_slot)
{
    std::vector<std::vector<uint8_t>> files(n);
    std::vector<const uint8_t *> ptrs(n);
    std::vector<size_t> lens(n);
    for (size_t i = 0; i < n; i++) {
        files[i].assign(2 + (i * 37) % 5000, uint8_t(i));
        files[i][0] = (i % 11 == 3) ? 0x00 : 0xff;         // (some files "do not parse")
        ptrs[i] = files[i].data();
        lens[i] = files[i].size();
    }
    std::vector<int> slot_of(n ? n : 1), status(n ? n : 1);
    std::vector<uint8_t *> rgb(n ? n : 1);
    mjx_pool_result *res = nullptr;
    mjx_opts o{};
    const int rc = mjx_pool_decode_batch(pool, ptrs.data(), lens.data(), n, &o, 0, slot_of.data(), rgb.data(), status.data(), &res);
    CHECK(res != nullptr);
    if (expect_fault_slot < 0) CHECK(rc == MJX_OK); else if (n >= mjx_pool_devices(pool)) CHECK(rc == MJX_ERR_DEVICE);
    for (size_t i = 0; i < n; i++) {
        CHECK(slot_of[i] >= 0 && size_t(slot_of[i]) < mjx_pool_devices(pool));
        if (slot_of[i] == expect_fault_slot) { CHECK(status[i] == MJX_ERR_DEVICE && rgb[i] == nullptr); continue; }
        CHECK(status[i] == (i % 11 == 3 ? MJX_ERR_TRUNCATED : MJX_OK));
        if (status[i] == MJX_OK) CHECK(rgb[i] && *rgb[i] == 0xff);
        size_t slot, index; mjx_batch *b;
        CHECK(mjx_pool_result_locate(res, i, &slot, &b, &index) == MJX_OK && slot == size_t(slot_of[i]) && b && index < b->n);
    }
    for (size_t s = 0; s < mjx_pool_devices(pool); s++) {
        unsigned t; int node; double ms;
        CHECK(mjx_pool_result_host(res, s, &t, &node) == MJX_OK && mjx_pool_result_slot_ms(res, s, &ms) == MJX_OK && ms >= 0.0);
    }
    mjx_pool_result_free(res);
}

int main()
{
    const int devices[8] = {0, 1, 0, 1, 0, 1, 0, 1};
    for (int fault = -1; fault <= 5; fault += 6) {          // no fault, then slot 5 fails
        if (fault >= 0) setenv("MJX_POOL_FAULT_SLOT", "5", 1); else unsetenv("MJX_POOL_FAULT_SLOT");
        mjx_pool *pool = nullptr;
        CHECK(mjx_pool_create(devices, 8, &pool) == MJX_OK && mjx_pool_devices(pool) == 8 && mjx_pool_device(pool, 7) == 1);
        for (int deal = 0; deal < 2; deal++) {
            CHECK(mjx_pool_set_deal(pool, deal) == MJX_OK);
            one_call(pool, 0, fault);
            one_call(pool, 1, fault);
            one_call(pool, 7, fault);
            one_call(pool, 4096, fault);
        }
        // two callers at once: the pool serialises them (call_mu); the slots' threads see one job at a time
        std::thread a([&] { for (int k = 0; k < 8; k++) one_call(pool, 64 + k, fault); });
        std::thread b([&] { for (int k = 0; k < 8; k++) one_call(pool, 33 + k, fault); });
        a.join();
        b.join();
        mjx_pool_destroy(pool);
    }
    CHECK(g_live_batches.load() == 0);
    std::printf("tsan pool stub: %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
