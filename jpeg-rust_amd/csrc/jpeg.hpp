// jpeg.hpp -- C++ mirror of the reference's decode surface over the C ABI (include/mjx.h).
//
//   reference                                             here
//   JPEGImage::parse(Vec<u8>) -> Result<JPEGImage,String>  jpeg::JPEGImage::parse(bytes) -> Result
//   .width() / .height() / .image_data()                   same names (src/jpeg/mod.rs:467-477)
//   JPEGDecoder::new(data).frame_header()...decode()       jpeg::JPEGDecoder (src/jpeg/decoder.rs:55-162)
//   HuffmanTable::from_size_data_tables                     jpeg::HuffmanTable (src/jpeg/huffman.rs:37)
//
// The reference's panics become error codes carried by Result (never exceptions across the ABI).
#ifndef MJX_JPEG_HPP
#define MJX_JPEG_HPP

#include "mjx.h"

#include <array>
#include <cstdint>
#include <cstring>
#include <string>
#include <tuple>
#include <vector>

namespace jpeg {

using Pixel = std::tuple<uint8_t, uint8_t, uint8_t>;

struct FrameComponentHeader { uint8_t component_id, horizontal_sampling_factor, vertical_sampling_factor, quantization_selector; };
struct FrameHeader { uint8_t sample_precision; uint16_t num_lines, samples_per_line; std::vector<FrameComponentHeader> frame_components; };
struct ScanComponentHeader { uint8_t component_id, dc_table_selector, ac_table_selector; };
struct ScanHeader { std::vector<ScanComponentHeader> scan_components; };

struct HuffmanTable {
    std::array<uint8_t, 16> size_data{};
    std::vector<uint8_t> data_table;
    static HuffmanTable from_size_data_tables(const uint8_t *size_data, const uint8_t *data_table, size_t n)
    {
        HuffmanTable t;
        std::memcpy(t.size_data.data(), size_data, 16);
        t.data_table.assign(data_table, data_table + (n > 256 ? 256 : n));
        return t;
    }
};

class JPEGDecoder {
public:
    explicit JPEGDecoder(const uint8_t *data, size_t len) : data_(data, data + len) { std::memset(&d_, 0, sizeof d_); }
    JPEGDecoder &dimensions(std::pair<size_t, size_t> dims) { d_.width = uint16_t(dims.first); d_.height = uint16_t(dims.second); return *this; }
    JPEGDecoder &frame_header(const FrameHeader &fh) { frame_ = fh; return *this; }
    JPEGDecoder &scan_header(const ScanHeader &sh) { scan_ = sh; return *this; }
    void huffman_ac_tables(uint8_t id, const HuffmanTable &t) { set(d_.ac[id & 3], t); d_.ac_present |= uint8_t(1u << (id & 3)); }
    void huffman_dc_tables(uint8_t id, const HuffmanTable &t) { set(d_.dc[id & 3], t); d_.dc_present |= uint8_t(1u << (id & 3)); }
    void quantization_table(uint8_t id, const std::vector<uint16_t> &t)
    {
        for (size_t k = 0; k < 64 && k < t.size(); k++) d_.qt[id & 3][k] = t[k];
        d_.qt_present |= uint8_t(1u << (id & 3));
    }
    // decode(): (pixels, bytes_read); bytes_read is bookkeeping the reference's caller ignores (mod.rs:415-417).
    int decode(mjx_ctx *ctx, std::vector<Pixel> &out, size_t &bytes_read)
    {
        if (scan_.scan_components.size() != 1 && scan_.scan_components.size() != 3) return MJX_ERR_UNSUPPORTED_FORMAT;
        d_.ncomp = uint8_t(scan_.scan_components.size());
        for (size_t i = 0; i < scan_.scan_components.size(); i++) {          // scan order, decoder.rs:141-150
            const ScanComponentHeader &sc = scan_.scan_components[i];
            const FrameComponentHeader *fc = nullptr;
            for (const auto &f : frame_.frame_components) if (f.component_id == sc.component_id) { fc = &f; break; }
            if (!fc) return MJX_ERR_REF_PANIC;
            d_.comp[i] = mjx_comp{sc.component_id, fc->horizontal_sampling_factor, fc->vertical_sampling_factor,
                                  fc->quantization_selector, sc.dc_table_selector, sc.ac_table_selector};
        }
        std::vector<uint8_t> padded(data_);
        padded.insert(padded.end(), 32, 0xaa);
        d_.scan = padded.data();
        d_.scan_len = data_.size();
        mjx_batch *b = nullptr;
        int st = MJX_OK;
        int rc = mjx_batch_create(ctx, &d_, 1, nullptr, &b, &st);
        if (rc != MJX_OK) return rc;
        if (st == MJX_OK) rc = mjx_batch_decode(b, MJX_STAGE_ALL);
        if (st == MJX_OK && rc == MJX_OK) rc = mjx_batch_wait(b);
        if (st == MJX_OK && rc == MJX_OK) st = mjx_batch_status(b, 0);
        if (st == MJX_OK && rc == MJX_OK) {
            std::vector<uint8_t> rgb(size_t(d_.width) * d_.height * 3);
            rc = mjx_batch_copy_rgb(b, 0, rgb.data());
            out.resize(size_t(d_.width) * d_.height);
            for (size_t i = 0; i < out.size(); i++) out[i] = Pixel(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
            bytes_read = data_.size();
        }
        mjx_batch_free(b);
        return rc != MJX_OK ? rc : st;
    }

private:
    static void set(mjx_hufftab &dst, const HuffmanTable &t)
    {
        std::memset(&dst, 0, sizeof dst);
        std::memcpy(dst.bits, t.size_data.data(), 16);
        std::memcpy(dst.vals, t.data_table.data(), t.data_table.size());
    }
    std::vector<uint8_t> data_;
    mjx_scan_desc d_;
    FrameHeader frame_{};
    ScanHeader scan_{};
};

class JPEGImage {
public:
    struct Result {
        int code = MJX_OK;                    // MJX_OK or the status that replaces the reference's panic
        std::string message;
        bool ok() const { return code == MJX_OK; }
    };
    // JPEGImage::parse, src/jpeg/mod.rs:202
    static Result parse(const std::vector<uint8_t> &vec, JPEGImage &image, const mjx_opts *opts = nullptr)
    {
        mjx_image img{0, 0, nullptr};
        Result r;
        r.code = mjx_decode(vec.data(), vec.size(), opts, &img);
        r.message = mjx_strerror(r.code);
        if (!r.ok()) return r;
        image.dimensions_ = {uint16_t(img.width), uint16_t(img.height)};
        image.image_data_.resize(size_t(img.width) * img.height);
        for (size_t i = 0; i < image.image_data_.size(); i++)
            image.image_data_[i] = Pixel(img.rgb[3 * i], img.rgb[3 * i + 1], img.rgb[3 * i + 2]);
        image.has_data_ = true;
        mjx_free_image(&img);
        return r;
    }
    size_t width() const { return dimensions_.first; }                          // mod.rs:467
    size_t height() const { return dimensions_.second; }                        // mod.rs:471
    const std::vector<Pixel> *image_data() const { return has_data_ ? &image_data_ : nullptr; }   // mod.rs:475 Option<&Vec>

private:
    std::pair<uint16_t, uint16_t> dimensions_{0, 0};
    std::vector<Pixel> image_data_;
    bool has_data_ = false;
};

}   // namespace jpeg
#endif
