// huff_emul.cpp -- CPU emulation of the GPU entropy-decode *algorithm* (test infrastructure only).
//
// Runs the same per-lane routine the HIP kernels run (jpeg-rust_amd/csrc/mjx_huff.h) over every
// subsequence sequentially, in the kernels' phase order: speculative pass, intra-workgroup
// synchronisation, inter-workgroup fix passes, block-count scan, write pass, DC prefix sum.
// It lets the CPU test-suite check the decode tables, the symbol step and the convergence of the
// self-synchronising scheme against the oracle without a GPU.  It is never linked into libmjx.so.
#include "mjx.h"
#include "mjx_huff.h"
#include "mjx_plan.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace mjx;

namespace {
struct HostBits {
    const uint8_t *p;
    size_t n;
    uint32_t be32(uint32_t i) const
    {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) {
            const size_t idx = size_t(i) * 4 + k;
            w = (w << 8) | (idx < n ? p[idx] : 0xaau);
        }
        return w;
    }
};
struct CoefSink {
    int16_t *coef;
    int16_t *dcbuf;
    int *bad;
    void dc(uint32_t b, int v) const { dcbuf[b] = int16_t(v); }
    void ac(uint32_t b, unsigned pos, int v) const { coef[size_t(b) * 64 + pos] = int16_t(v); }
    void bad_code(uint32_t) const { *bad = 1; }
    void tick() const {}
};
}   // namespace

namespace {
struct TickSink {
    mutable long ticks = 0;
    void dc(uint32_t, int) const {}
    void ac(uint32_t, unsigned, int) const {}
    void bad_code(uint32_t) const {}
    void tick() const { ticks++; }
};
struct HostCps {
    uint32_t *w;
    uint32_t get(uint32_t k) const { return w[k]; }
    void set(uint32_t k, uint32_t v) const { w[k] = v; }
};
}   // namespace

// wg_lanes = subsequence slots per workgroup, warm = slots at the front that re-decode the tail of the previous
// workgroup's range (results discarded) so that the first owned slot usually starts from a synchronised state.
extern "C" int emul_decode_coefs(const uint8_t *jpeg, size_t len, int layout, int wg_lanes, int16_t *out,
                                 size_t cap_blocks, size_t *nblocks, int *stats /* [8] */)
{
    int warm = wg_lanes >= 64 ? 4 : 1;
    if (wg_lanes < 0) { wg_lanes = -wg_lanes; warm = 0; }
    const uint32_t own = uint32_t(wg_lanes - warm);
    mjx_opts opts{};
    opts.layout = uint8_t(layout);
    mjx_scan_desc d;
    int rc = mjx_parse(jpeg, len, &opts, &d);
    if (rc) return rc;
    ImagePlan plan;
    rc = plan_image(d, opts, plan);
    if (rc) { mjx_free_scan(&d); return rc; }
    const HuffImage &img = plan.himg;
    const HostBits bits{plan.scan, plan.scan_len};
    const uint32_t nsub = img.nsub;
    std::vector<SubseqState> g_entry(nsub), g_exit(nsub);
    TickSink ns;
    NoCheckpoints nocp;
    std::vector<std::vector<long>> iter_ticks;   // per iteration index: symbols of every decode
    const char *dump = std::getenv("MJX_EMUL_DUMP");
    auto end_of = [&](uint32_t s) { uint64_t e = uint64_t(s + 1) * kSubseqBits; return uint32_t(e < img.total_bits ? e : img.total_bits); };
    long redecodes = 0, merged = 0, max_local_iters = 0, fix_passes = 0, fix_mismatch_first = 0;

    // Work-list synchronisation of the slots [0, nslot) of one workgroup whose slot l is subsequence base + l.
    auto wg_sync = [&](uint32_t base, uint32_t nslot, std::vector<SubseqState> &entry, std::vector<SubseqState> &exit_,
                       std::vector<uint32_t> &cps, bool use_cp, std::vector<uint32_t> work) {
        long iters = 0;
        while (!work.empty()) {
            if (iter_ticks.size() <= size_t(iters)) iter_ticks.resize(iters + 1);
            for (uint32_t l : work) {
                const long t0 = ns.ticks;
                const uint32_t s = base + l;
                SubseqState old = exit_[l];
                if (use_cp) {
                    HostCps hc{cps.data() + size_t(l) * kNumCp};
                    exit_[l] = decode_subseq<false, true>(bits, plan.lut.data(), img, entry[l], end_of(s), 0, ns, hc, s * kSubseqBits, old);
                } else {
                    exit_[l] = decode_subseq<false, false>(bits, plan.lut.data(), img, entry[l], end_of(s), 0, ns, nocp, 0, old);
                }
                redecodes++;
                iter_ticks[iters].push_back(ns.ticks - t0);
            }
            work.clear();
            for (uint32_t l = 1; l < nslot; l++)
                if (!same_entry(exit_[l - 1], entry[l])) {
                    entry[l].p = exit_[l - 1].p; entry[l].z = exit_[l - 1].z; entry[l].c = exit_[l - 1].c; entry[l].n = 0;
                    work.push_back(l);
                }
            iters++;
        }
        if (iters > max_local_iters) max_local_iters = iters;
    };

    // k_huff_sync
    const uint32_t nwg = (nsub + own - 1) / own;
    for (uint32_t w = 0; w < nwg; w++) {
        const uint32_t own0 = w * own, own1 = std::min(nsub, own0 + own);
        const uint32_t base = own0 >= uint32_t(warm) ? own0 - warm : 0;
        const uint32_t nslot = own1 - base;
        std::vector<SubseqState> entry(nslot), exit_(nslot);
        std::vector<uint32_t> cps(size_t(nslot) * kNumCp, 0), work(nslot);
        for (uint32_t l = 0; l < nslot; l++) { entry[l] = SubseqState{(base + l) * uint32_t(kSubseqBits), 0, 0, 0}; exit_[l] = SubseqState{0, 0, 0, 0}; work[l] = l; }
        wg_sync(base, nslot, entry, exit_, cps, true, work);
        for (uint32_t l = own0 - base; l < nslot; l++) { g_entry[base + l] = entry[l]; g_exit[base + l] = exit_[l]; }
    }
    redecodes -= nsub;   // first decodes are not re-decodes
    // k_huff_fix passes until one finds nothing
    for (;;) {
        long mism = 0;
        std::vector<SubseqState> snap(g_exit);
        for (uint32_t w = 1; w < nwg; w++) {
            const uint32_t own0 = w * own, own1 = std::min(nsub, own0 + own);
            if (same_entry(snap[own0 - 1], g_entry[own0])) continue;
            mism++;
            const uint32_t nslot = own1 - own0;
            std::vector<SubseqState> entry(g_entry.begin() + own0, g_entry.begin() + own1), exit_(g_exit.begin() + own0, g_exit.begin() + own1);
            std::vector<uint32_t> cps;
            entry[0].p = snap[own0 - 1].p; entry[0].z = snap[own0 - 1].z; entry[0].c = snap[own0 - 1].c; entry[0].n = 0;
            wg_sync(own0, nslot, entry, exit_, cps, false, std::vector<uint32_t>{0});
            for (uint32_t l = 0; l < nslot; l++) { g_entry[own0 + l] = entry[l]; g_exit[own0 + l] = exit_[l]; }
        }
        if (fix_passes == 0) fix_mismatch_first = mism;
        fix_passes++;
        if (mism == 0) break;
    }
    (void)merged;
    if (dump) {
        for (size_t it = 0; it < iter_ticks.size(); it++) {
            auto &v = iter_ticks[it];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            long sum = 0; for (long x : v) sum += x;
            std::fprintf(stderr, "iter %zu: items %zu mean %.1f p50 %ld p90 %ld p99 %ld max %ld\n", it, v.size(), double(sum) / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
        }
    }
    // block-count scan + write pass
    std::vector<uint32_t> blkbase(nsub);
    uint32_t acc = 0;
    for (uint32_t s = 0; s < nsub; s++) { blkbase[s] = acc; acc += g_exit[s].n; }
    const size_t nb = img.total_blocks;
    *nblocks = nb;
    int bad = 0;
    if (nb <= cap_blocks) {
        std::memset(out, 0, nb * 64 * sizeof(int16_t));
        std::vector<int16_t> dcb(nb, 0);
        CoefSink sink{out, dcb.data(), &bad};
        for (uint32_t s = 0; s < nsub; s++)
            decode_subseq<true, false>(bits, plan.lut.data(), img, g_entry[s], end_of(s), blkbase[s], sink, nocp, 0, g_exit[s]);
        int32_t pred[3] = {0, 0, 0};
        for (size_t b = 0; b < nb; b++) {
            const int c = plan.blk_comp[b % plan.bpm];
            pred[c] += dcb[b];
            out[b * 64] = int16_t(pred[c]);
        }
    }
    if (stats) {
        stats[0] = int(nsub); stats[1] = int(max_local_iters); stats[2] = int(redecodes); stats[3] = int(fix_passes);
        stats[4] = int(plan.bpm); stats[5] = int(plan.lut.size()); stats[6] = bad; stats[7] = int(fix_mismatch_first);
    }
    mjx_free_scan(&d);
    return nb <= cap_blocks ? (bad ? MJX_ERR_BAD_HUFFMAN : MJX_OK) : MJX_ERR_NOMEM;
}
