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
};
}   // namespace

extern "C" int emul_decode_coefs(const uint8_t *jpeg, size_t len, int layout, int wg_lanes, int16_t *out,
                                 size_t cap_blocks, size_t *nblocks, int *stats /* [8] */)
{
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
    std::vector<SubseqState> entry(nsub), exit_(nsub);
    NullSink ns;
    auto end_of = [&](uint32_t s) { uint64_t e = uint64_t(s + 1) * kSubseqBits; return uint32_t(e < img.total_bits ? e : img.total_bits); };
    long redecodes = 0, max_local_iters = 0, fix_passes = 0;

    // pass 0
    for (uint32_t s = 0; s < nsub; s++) {
        entry[s] = SubseqState{uint32_t(s) * kSubseqBits, 0, 0, 0};
        exit_[s] = decode_subseq<false>(bits, plan.lut.data(), img, entry[s], end_of(s), 0, ns);
    }
    auto local_sync = [&](uint32_t first, uint32_t last) {   // subsequences [first, last)
        long iters = 0;
        for (;;) {
            std::vector<uint32_t> work;
            for (uint32_t s = first + 1; s < last; s++)
                if (!same_entry(exit_[s - 1], entry[s])) work.push_back(s);
            if (work.empty()) break;
            // all lanes read their predecessor's exit before anyone re-decodes (barrier in the kernel)
            std::vector<SubseqState> ne(work.size());
            for (size_t k = 0; k < work.size(); k++) ne[k] = exit_[work[k] - 1];
            for (size_t k = 0; k < work.size(); k++) {
                const uint32_t s = work[k];
                entry[s] = ne[k];
                entry[s].n = 0;
                exit_[s] = decode_subseq<false>(bits, plan.lut.data(), img, entry[s], end_of(s), 0, ns);
                redecodes++;
            }
            iters++;
        }
        if (iters > max_local_iters) max_local_iters = iters;
    };
    for (uint32_t f = 0; f < nsub; f += wg_lanes) local_sync(f, f + wg_lanes < nsub ? f + wg_lanes : nsub);
    for (;;) {
        long changed_last = 0;
        std::vector<SubseqState> snapshot(exit_);
        for (uint32_t f = wg_lanes; f < nsub; f += wg_lanes) {
            const uint32_t l = f + wg_lanes < nsub ? f + wg_lanes : nsub;
            if (same_entry(snapshot[f - 1], entry[f])) continue;
            const SubseqState before = exit_[l - 1];
            entry[f] = snapshot[f - 1];
            entry[f].n = 0;
            exit_[f] = decode_subseq<false>(bits, plan.lut.data(), img, entry[f], end_of(f), 0, ns);
            redecodes++;
            local_sync(f, l);
            if (!same_entry(before, exit_[l - 1])) changed_last++;
        }
        fix_passes++;
        bool consistent = true;
        for (uint32_t f = wg_lanes; f < nsub; f += wg_lanes)
            if (!same_entry(exit_[f - 1], entry[f])) consistent = false;
        if (consistent) break;
        (void)changed_last;
    }
    // block-count scan + write pass
    std::vector<uint32_t> blkbase(nsub);
    uint32_t acc = 0;
    for (uint32_t s = 0; s < nsub; s++) { blkbase[s] = acc; acc += exit_[s].n; }
    const size_t nb = img.total_blocks;
    *nblocks = nb;
    int bad = 0;
    if (nb <= cap_blocks) {
        std::memset(out, 0, nb * 64 * sizeof(int16_t));
        std::vector<int16_t> dcb(nb, 0);
        CoefSink sink{out, dcb.data(), &bad};
        for (uint32_t s = 0; s < nsub; s++)
            decode_subseq<true>(bits, plan.lut.data(), img, entry[s], end_of(s), blkbase[s], sink);
        int32_t pred[3] = {0, 0, 0};
        for (size_t b = 0; b < nb; b++) {
            const int c = plan.blk_comp[b % plan.bpm];
            pred[c] += dcb[b];
            out[b * 64] = int16_t(pred[c]);
        }
    }
    if (stats) {
        stats[0] = int(nsub); stats[1] = int(max_local_iters); stats[2] = int(redecodes); stats[3] = int(fix_passes);
        stats[4] = int(plan.bpm); stats[5] = int(plan.lut.size()); stats[6] = bad; stats[7] = int(acc);
    }
    mjx_free_scan(&d);
    return nb <= cap_blocks ? (bad ? MJX_ERR_BAD_HUFFMAN : MJX_OK) : MJX_ERR_NOMEM;
}
