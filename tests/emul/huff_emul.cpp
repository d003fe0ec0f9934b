// huff_emul.cpp -- CPU emulation of the GPU entropy-decode *algorithm* (test infrastructure only).
//
// Runs the same per-lane routine the HIP kernels run (jpeg-rust_amd/csrc/mjx_huff.h) over every
// subsequence sequentially, in the kernels' phase order: speculative pass, intra-workgroup
// synchronisation, inter-workgroup fix passes, block-count scan, write pass, DC prefix sum.
// It lets the CPU test-suite check the decode tables, the symbol step and the convergence of the
// self-synchronising scheme against the oracle without a GPU.  It is never linked into libmjx.so.
#include "mjx.h"
#include "mjx_huff.h"
#include "mjx_kernels.h"
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
    uint32_t be32(uint32_t byte_off) const
    {
        uint32_t w = 0;
        for (int k = 0; k < 4; k++) {
            const size_t idx = size_t(byte_off) + k;
            w = (w << 8) | (idx < n ? p[idx] : 0xaau);
        }
        return w;
    }
    uint32_t ahead(uint32_t byte_off) const { return be32(byte_off); }
    void advance() const {}
};
// Mirrors the device sink of k_huff_write: AC entries appended to the compact stream, DC differences per block,
// the stream offset of every tile's first block.
struct StreamSink {
    uint32_t *entries;
    int16_t *dcbuf;
    uint32_t *tile_eoff;
    uint32_t tile_blocks, total_blocks;
    uint32_t off;
    int *bad;
    void dc(uint32_t b, int v)
    {
        dcbuf[b] = int16_t(v);
        if (b % tile_blocks == 0) tile_eoff[b / tile_blocks] = off;
    }
    void ac(uint32_t b, uint32_t r_scaled, int v) { entries[off++] = coef_entry(v, 63u - (r_scaled >> kRShift), b); }
    void block_done(uint32_t next_blk)
    {
        if (next_blk == total_blocks) tile_eoff[(total_blocks + tile_blocks - 1) / tile_blocks] = off;
    }
    void bad_code(uint32_t, uint32_t) const { *bad = 1; }
    void tick() const {}
};
// The same for the quad-interleaved stream (pictures of one scan on the device, mjx_kernels.h: stream_phys): the lane's entries go
// to its own column, a tile offset is the virtual index subsequence * column capacity + entry.
struct QuadSink {
    uint32_t *col;          // the columns (behind the run-length head of the region)
    int16_t *dcbuf;
    uint32_t *tile_eoff;
    uint32_t tile_blocks, total_blocks;
    uint32_t s, rows, off;
    int *bad;
    void dc(uint32_t b, int v)
    {
        dcbuf[b] = int16_t(v);
        if (b % tile_blocks == 0) tile_eoff[b / tile_blocks] = s * (rows * 8u) + off;
    }
    void ac(uint32_t b, uint32_t r_scaled, int v)
    {
        if (off >= rows * 8u) { *bad = 6; return; }                 // (a column holds any run: an entry takes two bits of scan at least)
        col[stream_phys(s, off++, rows)] = coef_entry(v, 63u - (r_scaled >> kRShift), b);
    }
    void block_done(uint32_t next_blk)
    {
        if (next_blk == total_blocks) tile_eoff[(total_blocks + tile_blocks - 1) / tile_blocks] = s * (rows * 8u) + off;
    }
    void bad_code(uint32_t, uint32_t) const { *bad = 1; }
    void tick() const {}
};
struct TickSink {
    mutable long ticks = 0;
    void dc(uint32_t, int) const {}
    void ac(uint32_t, uint32_t, int) const {}
    void block_done(uint32_t) const {}
    void bad_code(uint32_t, uint32_t) const {}
    void tick() const { ticks++; }
};
struct HostCps {
    uint32_t *w;     // 2 words per checkpoint
    uint32_t get(uint32_t k) const { return w[2 * k]; }
    uint32_t get_m(uint32_t k) const { return w[2 * k + 1]; }
    CpPair get_pair(uint32_t k) const { return CpPair{w[2 * k], w[2 * k + 1]}; }
    void set(uint32_t k, uint32_t v, uint32_t m) const { w[2 * k] = v; w[2 * k + 1] = m; }
};
}   // namespace

// Emulates the kernel sequence  k_huff_spec -> k_huff_merge rounds -> k_huff_scan -> k_huff_write -> DC prediction.
// `mode`: 0 = rounds see a snapshot of the previous round's exits (what fully concurrent lanes see in the worst case),
//         1 = rounds update in place in subsequence order (what a single lane walking the items would see).
// `sub_base_bits`: 0 = the subsequence length plan_image chooses; otherwise the scan is re-cut into subsequences of about
// that many bits (replan_subsequences: what build_batch does for batches too small to fill the device).
extern "C" int emul_decode_coefs_sub(const uint8_t *jpeg, size_t len, int layout, int mode, unsigned sub_base_bits, int16_t *out,
                                     size_t cap_blocks, size_t *nblocks, int *stats /* [8] */);
extern "C" int emul_decode_coefs(const uint8_t *jpeg, size_t len, int layout, int mode, int16_t *out,
                                 size_t cap_blocks, size_t *nblocks, int *stats /* [8] */)
{
    return emul_decode_coefs_sub(jpeg, len, layout, mode, 0, out, cap_blocks, nblocks, stats);
}
extern "C" int emul_decode_coefs_sub(const uint8_t *jpeg, size_t len, int layout, int mode, unsigned sub_base_bits, int16_t *out,
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
    if (sub_base_bits) replan_subsequences(plan, sub_base_bits);
    if (plan.restart_mcus) { mjx_free_scan(&d); return MJX_ERR_DRI_UNSUPPORTED; }   // (restart intervals: GPU tests only)
    const HuffImage &img = plan.himg;
    // the counting passes (k_huff_spec, the merge rounds) use the image's second table set: AC tables with a pair part, two
    // symbols per step where they can (mjx_huff.h); the write pass the plain one
    HuffImage img2 = img;
    for (uint32_t k = 0; k < uint32_t(kMaxBlocksPerMcu); k++) img2.btab[k].tabs = img.tabs_pair[k];
    const LutEntry *lut2 = plan.lut.data() + plan.lut_plain_n;
    const HostBits bits{plan.scan, plan.scan_len};
    const uint32_t nsub = img.nsub;
    std::vector<SubseqState> g_entry(nsub), g_exit(nsub);
    std::vector<uint32_t> g_cps(size_t(nsub) * kMaxCp * 2, 0xdeadbeefu);  // uninitialised on the device
    TickSink ns;
    NoCheckpoints nocp;
    std::vector<std::vector<long>> iter_ticks;
    const char *dump = std::getenv("MJX_EMUL_DUMP");
    auto end_of = [&](uint32_t s) { uint64_t e = uint64_t(s + 1) * img.sub_bits; return uint32_t(e < img.total_bits ? e : img.total_bits); };
    long redecodes = 0, rounds = 0;

    iter_ticks.emplace_back();
    for (uint32_t s = 0; s < nsub; s++) {                                  // k_huff_spec
        const SubseqState e = make_state(s * img.sub_bits, 0, 0);
        HostCps hc{g_cps.data() + size_t(s) * kMaxCp * 2};
        const long t0 = ns.ticks;
        g_exit[s] = decode_subseq<false, 1, true>(bits, lut2, img2, e, end_of(s), 0, ns, hc, s * img.sub_bits, e);
        g_entry[s] = e;
        iter_ticks[0].push_back(ns.ticks - t0);
    }
    // Two generations per subsequence (Gen2, mjx_kernels.hip; MJX_MERGE_MEMO=0 or mode 1: one, the loop below): the decode before the
    // last is kept -- entry, exit, checkpoints -- in a second set, gen[s] bit 0 = the current set, bit 1 = the other one holds a
    // decode.  A round: every workgroup's worth of lanes first takes remembered decodes back in a sweep (round_begin), then the lanes
    // that still do not start where their predecessor ended decode again, recording in a scratch set and comparing with both
    // recorded paths (merge_slice), and settle as merge_finish does.  Rounds see the previous round's exits (mode 0).
    const char *memo_env = std::getenv("MJX_MERGE_MEMO");
    const bool two = mode == 0 && !(memo_env && std::atoi(memo_env) == 0);
    if (two) {
        std::vector<SubseqState> ent[3], ext[2];
        std::vector<uint32_t> cp[3];
        std::vector<uint8_t> gen(nsub, 0);
        ent[0] = g_entry; ent[1].assign(nsub, make_state(0, 0, 0)); ent[2] = ent[1];
        ext[0] = g_exit; ext[1].assign(nsub, make_state(0, 0, 0));
        cp[0] = g_cps; cp[1].assign(g_cps.size(), 0xdeadbeefu); cp[2] = cp[1];
        auto cps_of = [&](int k, uint32_t s) { return HostCps{cp[k].data() + size_t(s) * kMaxCp * 2}; };
        for (;;) {
            long changed = 0;
            iter_ticks.emplace_back();
            std::vector<SubseqState> snap_exit(nsub);
            for (uint32_t s = 0; s < nsub; s++) snap_exit[s] = ext[gen[s] & 1][s];
            std::vector<char> active(nsub, 0);
            std::vector<SubseqState> start(nsub);
            for (uint32_t w0 = 1; w0 < nsub; w0 += uint32_t(kMergeWg)) {      // a merge workgroup: lanes w0 .. w0 + kMergeWg - 1
                const uint32_t w1 = std::min<uint32_t>(nsub, w0 + uint32_t(kMergeWg));
                std::vector<SubseqState> cur_exit(w1 - w0);
                std::vector<char> flipped(w1 - w0, 0);
                for (uint32_t s = w0; s < w1; s++) cur_exit[s - w0] = ext[gen[s] & 1][s];
                for (uint32_t iter = 0; iter < uint32_t(kMergeWg); iter++) {
                    std::vector<char> flip(w1 - w0, 0);
                    bool any = false;
                    for (uint32_t s = w0; s < w1; s++) {
                        const SubseqState prev = s == w0 ? snap_exit[s - 1] : cur_exit[s - 1 - w0];
                        const uint32_t c = gen[s] & 1;
                        if ((gen[s] & 2) && !same_entry(prev, ent[c][s]) && same_entry(prev, ent[c ^ 1][s])) { flip[s - w0] = 1; any = true; }
                    }
                    if (!any) break;
                    for (uint32_t s = w0; s < w1; s++)
                        if (flip[s - w0]) { gen[s] = uint8_t((gen[s] & 1) ^ 1) | 2; cur_exit[s - w0] = ext[gen[s] & 1][s]; flipped[s - w0] = 1; }
                }
                for (uint32_t s = w0; s < w1; s++) {
                    const SubseqState prev = s == w0 ? snap_exit[s - 1] : cur_exit[s - 1 - w0];
                    active[s] = !same_entry(prev, ent[gen[s] & 1][s]);
                    start[s] = make_state(prev.p, prev.z, prev.c);
                    if (active[s] || flipped[s - w0]) changed++;
                }
            }
            for (uint32_t s = 1; s < nsub; s++) {
                if (!active[s]) continue;
                const uint32_t cset = gen[s] & 1, oset = cset ^ 1;
                const bool other_valid = (gen[s] & 2) != 0;
                const SubseqState e = start[s];
                ent[2][s] = e;
                const HostCps cc = cps_of(int(cset), s), co = cps_of(int(oset), s), cw = cps_of(2, s);
                uint32_t met = 0, nrec = 0;
                SubseqState x = make_state(e.p, e.z, e.c);
                const long t0 = ns.ticks;
                if (e.p <= end_of(s)) {
                    LaneState st; LaneEvents ev;
                    lane_begin(st, bits, img2, e);
                    events_begin<2>(ev, s * img.sub_bits, end_of(s), img2.cp_bits);
                    uint32_t blk = 0;
                    for (;;) {
                        const bool crossed = symbol_step<false, true>(st, bits, lut2, img2, blk, ns);
                        if (!crossed || st.wn < ev.next_wn) continue;
                        if (st.wn >= ev.end_wn) break;
                        const uint32_t state = cp_state_word(st);
                        const uint32_t w_c = cc.get(ev.k), w_o = other_valid ? co.get(ev.k) : 0u;
                        if ((w_c & kCpStateMask) == state) {
                            met = 1;
                            x = make_state(ext[cset][s].p, ext[cset][s].z, ext[cset][s].c, st.n + ((w_c >> 16) & 0x7fffu), lane_m(st) + cc.get_m(ev.k));
                            break;
                        }
                        if (other_valid && (w_o & kCpStateMask) == state) {
                            met = 2;
                            x = make_state(ext[oset][s].p, ext[oset][s].z, ext[oset][s].c, st.n + ((w_o >> 16) & 0x7fffu), lane_m(st) + co.get_m(ev.k));
                            break;
                        }
                        cw.set(ev.k, state | (st.n << 16), lane_m(st));
                        ev.k++;
                        ev.next_wn += ev.cp_bytes;
                        if (ev.next_wn > ev.end_wn) ev.next_wn = ev.end_wn;
                    }
                    nrec = ev.k;
                    if (!met) x = make_state(lane_pos(st), lane_z(st), lane_c(st, img2), st.n, lane_m(st));
                }
                iter_ticks.back().push_back(ns.ticks - t0);
                const uint32_t dset = met == 1 ? cset : oset;
                const HostCps dst = cps_of(int(dset), s);
                for (uint32_t j = 0; j < nrec; j++) {
                    const CpPair v = cw.get_pair(j);
                    dst.set(j, (v.w & kCpStateMask) | ((x.n - ((v.w >> 16) & 0x7fffu)) << 16), x.m - v.m);
                }
                ent[dset][s] = ent[2][s];
                ext[dset][s] = x;
                if (met != 1) gen[s] = uint8_t(oset | 2);
                redecodes++;
            }
            rounds++;
            if (changed == 0) break;
            if (rounds > 100000) return MJX_ERR_INVALID_ARG;
        }
        for (uint32_t s = 0; s < nsub; s++) { g_entry[s] = ent[gen[s] & 1][s]; g_exit[s] = ext[gen[s] & 1][s]; }      // (k_huff_scan folds the sets)
    }
    for (; !two;) {                                                        // k_huff_merge rounds, one recorded decode per subsequence
        long redone = 0;
        iter_ticks.emplace_back();
        std::vector<SubseqState> snap;
        if (mode == 0) snap = g_exit;
        for (uint32_t s = 1; s < nsub; s++) {
            const SubseqState prev = mode == 0 ? snap[s - 1] : g_exit[s - 1];
            if (same_entry(prev, g_entry[s])) continue;
            const SubseqState e = make_state(prev.p, prev.z, prev.c);
            g_entry[s] = e;
            HostCps hc{g_cps.data() + size_t(s) * kMaxCp * 2};
            const long t0 = ns.ticks;
            g_exit[s] = decode_subseq<false, 2, true>(bits, lut2, img2, e, end_of(s), 0, ns, hc, s * img.sub_bits, g_exit[s]);
            iter_ticks.back().push_back(ns.ticks - t0);
            redone++;
        }
        rounds++;
        redecodes += redone;
        if (redone == 0) break;
    }
    if (dump) {
        for (size_t it = 0; it < iter_ticks.size(); it++) {
            auto &v = iter_ticks[it];
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            long sum = 0; for (long x : v) sum += x;
            std::fprintf(stderr, "pass %zu: items %zu mean %.1f p50 %ld p90 %ld p99 %ld max %ld\n", it, v.size(), double(sum) / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
        }
    }
    // k_huff_scan + k_huff_write + DC prediction + expansion of the compact stream
    std::vector<uint32_t> blkbase(nsub), ebase(nsub);
    uint32_t acc = 0, eacc = 0;
    for (uint32_t s = 0; s < nsub; s++) { blkbase[s] = acc; ebase[s] = eacc; acc += g_exit[s].n; eacc += g_exit[s].m; }
    const size_t nb = img.total_blocks;
    *nblocks = nb;
    int bad = 0;
    if (nb <= cap_blocks) {
        const uint32_t tile_blocks = tile_mcus(plan.bpm, plan.hmax) * plan.bpm;
        const uint32_t ntiles = uint32_t((nb + tile_blocks - 1) / tile_blocks);
        std::vector<uint32_t> entries(size_t(eacc) + 1, 0xffffffffu), tile_eoff(ntiles + 1, 0xffffffffu);
        std::vector<int16_t> dcb(nb, 0);
        for (uint32_t s = 0; s < nsub; s++) {
            StreamSink sink{entries.data(), dcb.data(), tile_eoff.data(), tile_blocks, uint32_t(nb), ebase[s], &bad};
            decode_subseq<true, 0>(bits, plan.lut.data(), img, g_entry[s], end_of(s), blkbase[s], sink, nocp, 0, g_exit[s]);
            // the entry count of the synchronisation passes must be exact for every subsequence that lies inside the image
            if (blkbase[s] + g_exit[s].n < nb && sink.off - ebase[s] != g_exit[s].m) { if (dump && !bad) std::fprintf(stderr, "m mismatch s=%u got %u want %u n=%u\n", s, sink.off - ebase[s], g_exit[s].m, g_exit[s].n); bad = 2; }
        }
        std::memset(out, 0, nb * 64 * sizeof(int16_t));
        for (uint32_t t = 0; t < ntiles; t++) {
            if (tile_eoff[t] == 0xffffffffu || tile_eoff[t + 1] == 0xffffffffu) { bad = 3; break; }
            for (uint32_t j = tile_eoff[t]; j < tile_eoff[t + 1]; j++) {
                const uint32_t e = entries[j];
                const uint32_t blk = t * tile_blocks + (((e >> 22) - t * tile_blocks) & 0xffu);
                if (blk >= nb) { bad = 4; break; }
                out[size_t(blk) * 64 + ((e >> 16) & 63)] = int16_t(e & 0xffff);
            }
        }
        // The write pass once more into the quad-interleaved layout, expanded the way stage B reads it (quad_prepare + quad_cell,
        // the functions the kernel runs): every group of every tile, masked at the tile's ends -- must give the same blocks.
        if (!bad) {
            const uint32_t rows = stream_rows_for(img.sub_bits), cap = rows * 8u, hdr = stream_hdr_entries(nsub);
            std::vector<uint32_t> region(size_t(hdr) + size_t(stream_quad_entries(nsub, rows)), 0xdeadbeefu), eoff2(ntiles + 1, 0xffffffffu);
            std::vector<int16_t> dcb2(nb, 0), out2(nb * 64, 0);
            uint32_t *runs = region.data();                                        // one run word per subsequence (mjx_kernels.h: run_word)
            for (uint32_t s = 0; s < nsub; s++) {
                QuadSink sink{region.data() + hdr, dcb2.data(), eoff2.data(), tile_blocks, uint32_t(nb), s, rows, 0u, &bad};
                decode_subseq<true, 0>(bits, plan.lut.data(), img, g_entry[s], end_of(s), blkbase[s], sink, nocp, 0, g_exit[s]);
                while (sink.off & 7u) region[hdr + stream_phys(s, sink.off++, rows)] = 0u;       // null entries up to the group boundary
                runs[s] = run_word(0u, sink.off >> 3, 0u);
            }
            for (uint32_t t = 0; t < ntiles && !bad; t++) {
                if (eoff2[t] == 0xffffffffu || eoff2[t + 1] == 0xffffffffu) { bad = 7; break; }
                const uint32_t sub[2] = {eoff2[t] / cap, eoff2[t + 1] / cap};
                const uint16_t at[2] = {uint16_t(eoff2[t] % cap), uint16_t(eoff2[t + 1] % cap)};
                QuadCum cum;
                QuadView q{sub, at, &cum, runs, rows, nsub};
                cum = quad_prepare(q, 0);
                QuadCell cell;
                const uint32_t total = quad_cell(q, 0, 0, cell);
                for (uint32_t o = 0; o < total; o++) {
                    if (quad_cell(q, 0, o, cell) != total || cell.phys == 0xffffffffu) { bad = 8; break; }
                    for (uint32_t k = cell.k_lo; k < cell.k_hi; k++) {
                        const uint32_t e = region[hdr + cell.phys + k];
                        if (((e >> 16) & 63) == 0) continue;                                   // null entry
                        const uint32_t blk = t * tile_blocks + (((e >> 22) + cell.label - t * tile_blocks) & 0xffu);
                        if (blk >= nb || blk / tile_blocks != t) { bad = 9; break; }
                        out2[size_t(blk) * 64 + ((e >> 16) & 63)] = int16_t(e & 0xffff);
                    }
                }
            }
            if (!bad && (std::memcmp(out2.data(), out, nb * 64 * sizeof(int16_t)) != 0 || dcb2 != dcb)) bad = 10;
        }
        int32_t pred[3] = {0, 0, 0};
        for (size_t b = 0; b < nb; b++) {
            const int c = plan.blk_comp[b % plan.bpm];
            pred[c] += dcb[b];
            out[b * 64] = int16_t(pred[c]);
        }
    }
    if (stats) {
        stats[0] = int(nsub); stats[1] = int(rounds); stats[2] = int(redecodes); stats[3] = int(rounds);
        stats[4] = int(plan.bpm); stats[5] = int(plan.lut.size()); stats[6] = bad; stats[7] = 0;
    }
    mjx_free_scan(&d);
    return nb <= cap_blocks ? (bad ? MJX_ERR_BAD_HUFFMAN : MJX_OK) : MJX_ERR_NOMEM;
}

// ---- single decode (round 5): the counting pass emits ------------------------------------------------------------------
// Emulates  k_huff_emit -> k_huff_merge rounds (count only, with the merge depth kept) -> k_huff_scan -> k_huff_prefix -> k_block_gather
// and the expansion stage B performs, on the CPU:
//   * every lane warms up over the last `warm_bits` bits of the subsequence before its own (count only, from the guess "a block
//     starts here") and then decodes its own subsequence ONCE, emitting -- entries into its column from index H on, labelled with
//     the number of blocks it has completed (not the block's index in the picture, which it cannot know), one word {DC difference,
//     column index where the block's entries start} per block into its block column from index Hb on -- and recording checkpoints;
//   * the merge rounds find the true entries as before; per subsequence the deepest merge point of its re-decodes is kept (kfix);
//   * lanes whose entry was wrong re-decode from the true entry up to that checkpoint and write the prefix RIGHT-ALIGNED against
//     the first entry / block word of the first decode that is valid, with labels continuing into that decode's;
//   * block bases (scan) give every subsequence's label -> block offset; the gather turns block columns into dcdiff[] and tile offsets.
// out/nblocks as emul_decode_coefs_sub.  stats: [0] nsub [1] rounds [2] lanes re-decoded by the prefix pass [3] symbols of the prefix pass
// [4] symbols of the merge rounds [5] max(m_prefix - j0K) [6] bad [7] max(n_prefix - n0K)
namespace {
struct EmitSink {                  // first decode and prefix pass: the column of one subsequence (entries upwards, block words from the top downwards)
    uint32_t *col;
    uint32_t s, rows, eoff, boff;             // next entry index in the column; block word index = boff + label
    uint32_t label_add;                       // what is added to the decode's own block count to give the label
    uint32_t e_lo, e_hi;                      // bounds the entry indices must stay inside (a violation = fall back to the two-pass path)
    uint32_t bad_label; int *overflow;
    uint32_t bad_pos = 0;
    uint32_t lowest_bword = 0xffffffffu;      // lowest column index a block word went to (entries must stay below it)
    long ticks = 0;
    void dc(uint32_t b, int v)
    {
        const uint32_t j = block_word_index(boff + b, rows);
        if (j >= rows * 8u || j < eoff) { *overflow = 1; return; }
        lowest_bword = std::min(lowest_bword, j);
        col[stream_phys(s, j, rows)] = (uint32_t(v) & 0xffffu) | (eoff << 16);
    }
    void ac(uint32_t b, uint32_t r_scaled, int v)
    {
        if (eoff < e_lo || eoff >= e_hi || eoff >= lowest_bword) { *overflow = 1; return; }
        col[stream_phys(s, eoff++, rows)] = coef_entry(v, 63u - (r_scaled >> kRShift), b + label_add);
    }
    void block_done(uint32_t) const {}
    void bad_code(uint32_t b, uint32_t pos) { if (bad_label == 0xffffffffu) { bad_label = b; bad_pos = pos; } }
    void tick() { ticks++; }
};
}   // namespace

extern "C" int emul_single_decode_cp(const uint8_t *jpeg, size_t len, int layout, int mode, unsigned sub_base_bits, unsigned warm_bits,
                                     unsigned headroom_groups, unsigned cp_bits, int16_t *out, size_t cap_blocks, size_t *nblocks, int *stats /* [8] */);
extern "C" int emul_single_decode(const uint8_t *jpeg, size_t len, int layout, int mode, unsigned sub_base_bits, unsigned warm_bits,
                                  unsigned headroom_groups, int16_t *out, size_t cap_blocks, size_t *nblocks, int *stats /* [8] */)
{
    return emul_single_decode_cp(jpeg, len, layout, mode, sub_base_bits, warm_bits, headroom_groups, 0, out, cap_blocks, nblocks, stats);
}
// cp_bits: bits between the checkpoints the emitting decode records (0: kCpBits; a multiple of kCpBits)
extern "C" int emul_single_decode_cp(const uint8_t *jpeg, size_t len, int layout, int mode, unsigned sub_base_bits, unsigned warm_bits,
                                     unsigned headroom_groups, unsigned cp_bits, int16_t *out, size_t cap_blocks, size_t *nblocks, int *stats /* [8] */)
{
    mjx_opts opts{};
    opts.layout = uint8_t(layout);
    mjx_scan_desc d;
    int rc = mjx_parse(jpeg, len, &opts, &d);
    if (rc) return rc;
    ImagePlan plan;
    rc = plan_image(d, opts, plan);
    if (rc) { mjx_free_scan(&d); return rc; }
    if (sub_base_bits) replan_subsequences(plan, sub_base_bits);
    if (plan.restart_mcus) { mjx_free_scan(&d); return MJX_ERR_DRI_UNSUPPORTED; }
    const uint32_t CPB = cp_bits ? std::max<uint32_t>(uint32_t(kCpBits), cp_bits / uint32_t(kCpBits) * uint32_t(kCpBits)) : uint32_t(kCpBits);
    plan.himg.cp_bits = CPB;
    const HuffImage &img = plan.himg;
    HuffImage img2 = img;                                                     // counting passes: the table set with pair parts
    for (uint32_t k = 0; k < uint32_t(kMaxBlocksPerMcu); k++) img2.btab[k].tabs = img.tabs_pair[k];
    const LutEntry *lut2 = plan.lut.data() + plan.lut_plain_n;
    HuffImage imgw = img;                                                     // emitting passes: no "last block" to stop at
    imgw.total_blocks = 0xffffffffu;
    const HostBits bits{plan.scan, plan.scan_len};
    const uint32_t nsub = img.nsub, L = img.sub_bits;
    const uint32_t W = std::min<uint32_t>(warm_bits / 32u * 32u, L);
    headroom_groups = std::min<unsigned>(headroom_groups, kEmitHeadGroups);
    const uint32_t H = headroom_groups * 8u, Hb = headroom_groups * 4u;       // head room in front of the first decode's entries / block words (DevImage::emit_head)
    const uint32_t rows = stream_rows_for(L), cap = rows * 8u;
    auto end_of = [&](uint32_t s) { uint64_t e = uint64_t(s + 1) * L; return uint32_t(e < img.total_bits ? e : img.total_bits); };
    const uint32_t hdr = stream_hdr_entries(nsub);
    std::vector<uint32_t> whole(size_t(hdr) + size_t(stream_quad_entries(nsub, rows)), 0xdeadbeefu);
    uint32_t *runs = whole.data(), *region = whole.data() + hdr;
    std::vector<SubseqState> g_entry(nsub), g_exit(nsub);
    std::vector<uint32_t> g_cps(size_t(nsub) * kMaxCp * 2, 0xdeadbeefu), d0n(nsub), d0m(nsub), kfix(nsub, 0), badl(nsub, 0xffffffffu), badp(nsub, 0);
    constexpr uint32_t kAll = kEmitAll;
    NoCheckpoints nocp;
    TickSink ns;
    int overflow = 0, bad = 0;
    long merge_ticks = 0, prefix_ticks = 0, prefix_lanes = 0, rounds = 0;
    // ---- k_huff_emit
    for (uint32_t s = 0; s < nsub; s++) {
        SubseqState e = make_state(0, 0, 0);
        if (s > 0) {
            const uint32_t from = s * L - W;
            e = W ? decode_subseq<false, 0>(bits, plan.lut.data(), img, make_state(from, 0, 0), s * L, 0, ns, nocp, from, make_state(0, 0, 0))
                  : make_state(s * L, 0, 0);
            e.n = e.m = 0;
        }
        EmitSink sink{region, s, rows, H, Hb, 0u, H, cap, 0xffffffffu, &overflow};
        HostCps hc{g_cps.data() + size_t(s) * kMaxCp * 2};
        const SubseqState x = decode_subseq<true, 1>(bits, plan.lut.data(), imgw, e, end_of(s), 0, sink, hc, s * L, e);
        while (sink.eoff & 7u) region[stream_phys(s, sink.eoff++, rows)] = 0u;     // null entries up to the group boundary
        g_entry[s] = make_state(e.p, e.z, e.c);
        g_exit[s] = x;
        d0n[s] = x.n; d0m[s] = x.m;
        badl[s] = sink.bad_label;
        badp[s] = sink.bad_pos - s * L;
    }
    // ---- merge rounds (count only); the deepest merge point of a subsequence's re-decodes is kept
    for (;;) {
        long redone = 0;
        std::vector<SubseqState> snap;
        if (mode == 0) snap = g_exit;
        for (uint32_t s = 1; s < nsub; s++) {
            const SubseqState prev = mode == 0 ? snap[s - 1] : g_exit[s - 1];
            if (same_entry(prev, g_entry[s])) continue;
            const SubseqState e = make_state(prev.p, prev.z, prev.c);
            g_entry[s] = e;
            HostCps hc{g_cps.data() + size_t(s) * kMaxCp * 2};
            uint32_t depth = kAll;
            if (e.p > end_of(s)) { g_exit[s] = make_state(e.p, e.z, e.c); }
            else {
                LaneState st; LaneEvents ev;
                lane_begin(st, bits, img2, e);
                events_begin<2>(ev, s * L, end_of(s), CPB);
                uint32_t blk = 0;
                const long t0 = ns.ticks;
                for (;;) {
                    const bool crossed = symbol_step<false, true>(st, bits, lut2, img2, blk, ns);
                    if (crossed && lane_event<2>(st, ev, img2, hc)) break;
                }
                merge_ticks += ns.ticks - t0;
                checkpoint_fixup(hc, ev.k, st.n, lane_m(st));
                if (ev.merged) depth = ev.k + 1;
                g_exit[s] = lane_exit(st, ev, img2, g_exit[s]);
            }
            kfix[s] = std::max(kfix[s], depth);
            redone++;
        }
        rounds++;
        if (redone == 0) break;
        if (rounds > 64) { bad = 11; break; }
    }
    // ---- k_huff_scan: blocks completed before every subsequence
    std::vector<uint32_t> B(nsub + 1, 0);
    for (uint32_t s = 0; s < nsub; s++) B[s + 1] = B[s] + g_exit[s].n;
    const size_t nb = img.total_blocks;
    *nblocks = nb;
    // ---- k_huff_prefix: re-decode from the true entry up to the merge point, right-aligned against the first decode's valid part
    std::vector<uint32_t> gs(nsub, H / 8u), ge(nsub);
    std::vector<int32_t> lbl(nsub, 0);
    int64_t worst_e = 0, worst_b = 0;
    for (uint32_t s = 0; s < nsub; s++) {
        ge[s] = (H + d0m[s] + 7u) / 8u;
        if (!kfix[s]) continue;
        prefix_lanes++;
        uint32_t n0r = 0, m0r = 0, stop_bit = end_of(s);
        if (kfix[s] != kAll) {
            const uint32_t k = kfix[s] - 1;
            n0r = (g_cps[(size_t(s) * kMaxCp + k) * 2] >> 16) & 0x7fffu;
            m0r = g_cps[(size_t(s) * kMaxCp + k) * 2 + 1];
            stop_bit = s * L + (k + 1) * CPB;
        }
        const uint32_t n0K = d0n[s] - n0r, j0K = d0m[s] - m0r, n_p = g_exit[s].n - n0r, m_p = g_exit[s].m - m0r;
        worst_e = std::max<int64_t>(worst_e, int64_t(m_p) - int64_t(j0K));
        worst_b = std::max<int64_t>(worst_b, int64_t(n_p) - int64_t(n0K));
        if (int64_t(H) + j0K < int64_t(m_p) || int64_t(Hb) + n0K < int64_t(n_p)) { overflow = 1; continue; }
        const uint32_t off_e = H + j0K - m_p;
        lbl[s] = int32_t(n0K) - int32_t(n_p);
        EmitSink sink{region, s, rows, off_e, uint32_t(int32_t(Hb) + lbl[s]), uint32_t(lbl[s]), off_e, H + j0K, 0xffffffffu, &overflow};
        for (uint32_t i = off_e & ~7u; i < off_e; i++) region[stream_phys(s, i, rows)] = 0u;      // null entries in front of the run's first entry
        gs[s] = off_e >> 3;
        const SubseqState e = g_entry[s];
        if (e.p <= end_of(s)) {
            LaneState st;
            lane_begin(st, bits, imgw, e);
            uint32_t blk = 0;
            const uint32_t stop_wn = wn_after(stop_bit);
            while (st.wn < stop_wn) (void)symbol_step<true, false>(st, bits, plan.lut.data(), imgw, blk, sink);
            prefix_ticks += sink.ticks;
            if (sink.eoff != H + j0K || st.n != n_p) bad = 12;                   // the counts of the merge rounds must be exact
            if (kfix[s] != kAll) {
                const uint32_t rec = g_cps[(size_t(s) * kMaxCp + kfix[s] - 1) * 2];
                if ((rec & kCpStateMask) != cp_state_word(st)) bad = 13;        // ... and the prefix must meet the first decode's path there
            }
            if (sink.bad_label != 0xffffffffu && B[s] + sink.bad_label < nb) bad = 1;
        }
        if (badl[s] != 0xffffffffu) {       // the first decode's invalid code counts if it lies at or behind the merge point
            if (kfix[s] == kAll) badl[s] = 0xffffffffu;
            else {
                const uint32_t rec = g_cps[(size_t(s) * kMaxCp + kfix[s] - 1) * 2];
                const uint32_t pK = 8u * (wn_after(kfix[s] * CPB) - 8u) - (rec & 31u);
                if (badp[s] < pK) badl[s] = 0xffffffffu;
            }
        }
    }
    for (uint32_t s = 0; s < nsub; s++) runs[s] = run_word(gs[s], ge[s], uint32_t(int32_t(B[s]) - lbl[s]) & 0xffu);
    if (overflow) bad = 14;                        // (the device hands the picture to the two-pass kernels: nothing else counts)
    for (uint32_t s = 0; s < nsub; s++)
        if (badl[s] != 0xffffffffu && int64_t(B[s]) + int64_t(badl[s]) - lbl[s] < int64_t(nb)) bad = bad ? bad : 1;
    // ---- k_block_gather + stage B's expansion (quad_prepare / quad_cell: the functions the kernel runs)
    if (!bad && nb <= cap_blocks && B[nsub] >= nb) {
        const uint32_t tile_blocks = tile_mcus(plan.bpm, plan.hmax) * plan.bpm;
        const uint32_t ntiles = uint32_t((nb + tile_blocks - 1) / tile_blocks);
        std::vector<uint32_t> eoff(ntiles + 1, 0xffffffffu);
        std::vector<int16_t> dcb(nb, 0);
        for (uint32_t s = 0; s < nsub; s++) {
            const uint32_t fs = g_entry[s].z ? 1u : 0u, fs1 = s + 1 < nsub ? (g_entry[s + 1].z ? 1u : 0u) : (g_exit[s].z ? 1u : 0u);
            for (uint64_t a = uint64_t(B[s]) + fs; a < uint64_t(B[s + 1]) + fs1 && a <= nb; a++) {
                const uint32_t w = region[stream_phys(s, block_word_index(uint32_t(int64_t(Hb) + int64_t(a - B[s]) + lbl[s]), rows), rows)];
                if (a < nb) dcb[a] = int16_t(w & 0xffffu);
                if (a == nb) eoff[ntiles] = s * cap + (w >> 16);
                else if (a % tile_blocks == 0) eoff[a / tile_blocks] = s * cap + (w >> 16);
            }
            if (s == nsub - 1 && uint64_t(B[s + 1]) + fs1 <= nb) eoff[ntiles] = s * cap + H + d0m[s];
        }
        std::memset(out, 0, nb * 64 * sizeof(int16_t));
        for (uint32_t t = 0; t < ntiles && !bad; t++) {
            if (eoff[t] == 0xffffffffu || eoff[t + 1] == 0xffffffffu) { bad = 7; break; }
            const uint32_t sub[2] = {eoff[t] / cap, eoff[t + 1] / cap};
            const uint16_t at[2] = {uint16_t(eoff[t] % cap), uint16_t(eoff[t + 1] % cap)};
            QuadCum cum;
            QuadView q{sub, at, &cum, runs, rows, nsub};
            cum = quad_prepare(q, 0);
            QuadCell cell;
            const uint32_t total = quad_cell(q, 0, 0, cell);
            const uint32_t first_lo = (t * tile_blocks) & 0xffu;
            for (uint32_t o = 0; o < total && !bad; o++) {
                if (quad_cell(q, 0, o, cell) != total || cell.phys == 0xffffffffu) { bad = 8; break; }
                for (uint32_t k = cell.k_lo; k < cell.k_hi; k++) {
                    const uint32_t e = region[cell.phys + k];
                    if (((e >> 16) & 63) == 0) continue;                        // null entry
                    const uint32_t bi = ((e >> 22) - (first_lo - cell.label)) & 0xffu;      // (as scatter_batch)
                    const uint32_t blk = t * tile_blocks + bi;
                    if (bi >= tile_blocks || blk >= nb) { bad = 9; break; }
                    out[size_t(blk) * 64 + ((e >> 16) & 63)] = int16_t(e & 0xffff);
                }
            }
        }
        int32_t pred[3] = {0, 0, 0};
        for (size_t b = 0; b < nb; b++) {
            const int c = plan.blk_comp[b % plan.bpm];
            pred[c] += dcb[b];
            out[b * 64] = int16_t(pred[c]);
        }
    } else if (!bad && B[nsub] < nb) bad = 15;                                 // truncated
    if (stats) {
        stats[0] = int(nsub); stats[1] = int(rounds); stats[2] = int(prefix_lanes); stats[3] = int(prefix_ticks);
        stats[4] = int(merge_ticks); stats[5] = int(worst_e); stats[6] = bad; stats[7] = int(worst_b);
    }
    mjx_free_scan(&d);
    return nb <= cap_blocks ? (bad ? MJX_ERR_BAD_HUFFMAN : MJX_OK) : MJX_ERR_NOMEM;
}

// ---- round 5: multi-scan pictures read without the gather, and what the planner decides per scan ------------------------------------
// The cuts of a scan's stream as k_huff_write records them (StreamSink::next_cut): the first cut at or after block 0, then always
// the first one behind the last.  Returns the number of cuts (mcu_out / slot_out hold the first `cap`).
extern "C" int emul_planar_cuts(unsigned scan_mcux, unsigned scan_mcuy, unsigned T, unsigned pic_mcux, unsigned hs, unsigned vs,
                                unsigned *mcu_out, unsigned *slot_out, int cap)
{
    const uint32_t S = planar_row_slots(pic_mcux, T), nmcu = scan_mcux * scan_mcuy;
    int n = 0;
    for (uint32_t q = 0; q < nmcu;) {
        const PlanarCut c = planar_cut(q, scan_mcux, S, T, pic_mcux, hs, vs);
        if (c.mcu >= nmcu) break;
        if (n < cap) { mcu_out[n] = c.mcu; slot_out[n] = c.slot; }
        n++;
        q = c.mcu + 1;
    }
    return n;
}
// plan_input on a file: per plan {role, part_idx, ncomp, bpm, himg.bpm (the decoder's block-in-MCU period), sub_bits, nsub, status}
extern "C" int emul_plan_parts(const uint8_t *jpeg, size_t len, int *out /* [cap][8] */, int cap)
{
    mjx_opts opts{};
    mjx_scan_desc d;
    const int rc = mjx_parse(jpeg, len, &opts, &d);
    if (rc) return -rc;
    std::vector<ImagePlan> plans;
    plan_input(d, opts, plans);
    int n = 0;
    for (const ImagePlan &p : plans) {
        if (n < cap) {
            int *o = out + 8 * n;
            o[0] = int(p.role); o[1] = int(p.part_idx); o[2] = int(p.ncomp); o[3] = int(p.bpm); o[4] = int(p.himg.bpm);
            o[5] = int(p.himg.sub_bits); o[6] = int(p.himg.nsub); o[7] = p.status;
        }
        n++;
    }
    mjx_free_scan(&d);
    return n;
}
