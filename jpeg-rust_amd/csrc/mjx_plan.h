// mjx_plan.h -- host-side per-image decode plan: geometry (reference src/jpeg/decoder.rs:164-192,
// 239-250), decode tables, dequantisation multipliers.  Shared by the device API (mjx_device.hip) and the
// CPU emulation harness (tests/emul).
#ifndef MJX_PLAN_H
#define MJX_PLAN_H

#include "mjx.h"
#include "mjx_huff.h"

#include <vector>

namespace mjx {

struct ImagePlan {
    int status = MJX_OK;
    uint32_t width = 0, height = 0;
    uint32_t ncomp = 0;
    uint32_t h[3] = {1, 1, 1}, v[3] = {1, 1, 1};   // effective sampling factors (scan order)
    uint32_t tq[3] = {0, 0, 0};
    uint32_t hmax = 1, vmax = 1;
    uint32_t bpm = 0;                              // blocks per MCU
    uint32_t mcux = 0, mcuy = 0;                   // MCU grid of the standard layout
    uint32_t nmcu = 0;                             // MCUs to decode (Q2 count in REF_COMPAT)
    uint32_t layout = MJX_LAYOUT_STANDARD;
    uint8_t blk_comp[kMaxBlocksPerMcu] = {0};      // block position in MCU -> component
    uint8_t blk_bx[kMaxBlocksPerMcu] = {0};        //                       -> block column inside the MCU
    uint8_t blk_by[kMaxBlocksPerMcu] = {0};        //                       -> block row inside the MCU
    HuffImage himg{};
    std::vector<LutEntry> lut;                     // all decode tables of the image, concatenated: the plain set (what the write pass
    uint32_t lut_plain_n = 0;                      // uses), then from lut_plain_n on the set with pair parts (the counting passes)
    float qmult[3][64];                            // per component, zig-zag order: q[k] * idct prescale
    const uint8_t *scan = nullptr;
    size_t scan_len = 0;
    // The scan still holds FF00 pairs (and RSTn markers): it is de-stuffed on the device at upload (k_destuff_*), scan_len is
    // the stuffed length -- an upper bound; himg.total_bits, himg.nsub and the segment table are bounds and placeholders here,
    // the device writes the exact values into the DevImage (k_destuff_prefix, k_restart_geometry).
    bool stuffed = false;
    uint32_t wg_lanes = uint32_t(kHuffWg);         // lanes of the entropy workgroups the scan is cut for: 512, or 256 / 128 for a scan that fills no more
                                                   // (replan_subsequences; k_huff_spec / k_huff_write run a chunk at its pictures' largest)
    uint32_t nsub_layout = 0;                      // subsequences the scan pool region is laid out for (0: himg.nsub; a batch tiled from
                                                   // one that was de-stuffed on the device keeps the region of the bound)
    // REF_COMPAT placement (decoder.rs:239-250): replication factors per component, block grid of the image
    uint32_t ref_xf[3] = {1, 1, 1}, ref_yf[3] = {1, 1, 1};
    uint32_t nbx = 0, nby = 0;
    // Restart intervals: the scan is a sequence of independent segments of restart_mcus MCUs each.
    // seg[g] = (first subsequence, first bit) of segment g, plus a sentinel (nsub, total_bits).  One segment when the
    // image has no restart interval.
    uint32_t restart_mcus = 0;
    std::vector<uint32_t> seg;                     // 2 * (nseg + 1) words
    uint32_t nseg = 1;
    // Multi-scan files (one component per scan): every scan is planned as a one-component picture of
    // its own (role 1: entropy decode and DC prediction only, blocks in the component's raster order), the picture
    // itself as role 2 (no scan: its coefficient stream is gathered from the role-1 streams, then stage B as usual).
    uint32_t role = 0;
    uint32_t cbw[3] = {0, 0, 0}, cbh[3] = {0, 0, 0};   // role 2: the components' own block grids (T.81 A.2.2)
    uint32_t nparts = 0;                               // role 2: scans in front of the picture; role 1: scans of its file
    uint32_t part_idx = 0;                             // role 1: which of them this is (the picture's plan lies nparts - part_idx plans behind)
    uint32_t src_part[3] = {0, 0, 0}, src_comp[3] = {0, 0, 0};   // ... which of them carries component c, as its n-th component
};

// decoder.rs:259-288 get_indices: raster counter (x, y) of a component's blocks -> block position (bug-for-bug, Q3).
// Returns false where the reference's usize arithmetic underflows (panic).
bool ref_get_indices(long x, long y, long max_x, long x_factor, long y_factor, long max_x_factor, long max_y_factor,
                     long *ox, long *oy);

// Validates `d` and fills `plan`.  Returns plan.status.  `scan_part`: d is one scan of a multi-scan file (two interleaved
// components are then allowed).
int plan_image(const mjx_scan_desc &d, const mjx_opts &opts, ImagePlan &plan, bool scan_part = false);

// Re-cuts a planned picture's scan into subsequences of about `base_bits` bits (at most kSubseqBits; plan_image uses
// kSubseqBits).  Batches too small to fill the device are re-planned with shorter subsequences (mjx_api.hip).
void replan_subsequences(ImagePlan &plan, uint32_t base_bits, bool allow_long = true);     // allow_long: kLongSubseqBits for long scans (base_bits == kSubseqBits only)

// The plans of one input: `plan_image` for an ordinary file; for a multi-scan file one role-1 plan per scan (in file
// order) followed by the role-2 plan of the picture.  The last plan appended is the picture's.
void plan_input(const mjx_scan_desc &d, const mjx_opts &opts, std::vector<ImagePlan> &out);

// mjx_parse.cpp: mjx_parse with caller-lent storage for the de-stuffed scan (see there)
int parse_into(const uint8_t *jpeg, size_t len, const mjx_opts *opts, mjx_scan_desc *out, uint8_t *storage, size_t cap);

extern const uint8_t kZigZag[64];                  // decoder.rs:404-407 ZIGZAG_INDICES

}   // namespace mjx
#endif
