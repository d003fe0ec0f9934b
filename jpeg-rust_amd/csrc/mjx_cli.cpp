// mjx_cli.cpp -- counterpart of the reference's CLI (src/main.rs:24-40):  mjx_cli <in.jpeg> <out.ppm> [--p6] [--strict]
// Writes the same ASCII P3 file ("P3\n{w} {h}\n255\n" then "r g b\n" per pixel, main.rs:35-39), buffered; --p6 writes
// binary PPM instead.  Exit code = MJX_* status.
#include "jpeg.hpp"

#include <cstdio>
#include <cstring>
#include <string>

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <input.jpeg> <output.ppm> [--p6] [--strict] [--ref-compat]\n", argv[0]);   // main.rs:26-28 expect()
        return MJX_ERR_INVALID_ARG;
    }
    bool p6 = false;
    mjx_opts opts{};
    for (int i = 3; i < argc; i++) {
        if (!std::strcmp(argv[i], "--p6")) p6 = true;
        else if (!std::strcmp(argv[i], "--strict")) opts.strict_ref = 1;
        else if (!std::strcmp(argv[i], "--ref-compat")) opts.layout = MJX_LAYOUT_REF_COMPAT;
    }
    std::FILE *f = std::fopen(argv[1], "rb");                                   // file_to_bytes, main.rs:16-22
    if (!f) { std::perror(argv[1]); return MJX_ERR_INVALID_ARG; }
    std::vector<uint8_t> bytes;
    uint8_t buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    jpeg::JPEGImage image;
    const jpeg::JPEGImage::Result r = jpeg::JPEGImage::parse(bytes, image, &opts);   // main.rs:31
    if (!r.ok()) { std::fprintf(stderr, "decode failed: %s (%d)\n", r.message.c_str(), r.code); return r.code; }
    std::FILE *o = std::fopen(argv[2], "wb");
    if (!o) { std::perror(argv[2]); return MJX_ERR_INVALID_ARG; }
    std::fprintf(o, "%s\n%zu %zu\n255\n", p6 ? "P6" : "P3", image.width(), image.height());   // main.rs:35
    std::string out;
    for (const jpeg::Pixel &px : *image.image_data()) {                          // main.rs:36-39
        if (p6) { out.push_back(char(std::get<0>(px))); out.push_back(char(std::get<1>(px))); out.push_back(char(std::get<2>(px))); }
        else { char line[24]; out.append(line, size_t(std::snprintf(line, sizeof line, "%u %u %u\n", std::get<0>(px), std::get<1>(px), std::get<2>(px)))); }
        if (out.size() > (1u << 20)) { std::fwrite(out.data(), 1, out.size(), o); out.clear(); }
    }
    std::fwrite(out.data(), 1, out.size(), o);
    std::fclose(o);
    return MJX_OK;
}
