// statmc_pfm.hpp -- the statistics-dump wire format: 32-bit float PFM, as the reference writes it
// with cv::imwrite (src/statistics/buffer.cpp:40-53: RGB->BGR + imwrite, i.e. a standard "PF"
// file with RGB triples; 1-channel buffers as "Pf"; `n` converted to float first, buffer.h:51-54)
// and reads it back with cv::imread + BGR->RGB (src/statistics/statpath.cpp:449-454).
// PFM: header "PF|Pf\n<w> <h>\n<scale>\n", rows stored bottom-to-top, scale < 0 = little endian.
#ifndef STATMC_PFM_HPP
#define STATMC_PFM_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace statmc {

struct PfmImage {
    int width = 0, height = 0, channels = 0;
    std::vector<float> data;  // top-to-bottom, interleaved
};

inline PfmImage readPfm(const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    char magic[3] = {0, 0, 0};
    PfmImage im;
    double scale = 0;
    if (std::fscanf(f, "%2s %d %d %lf", magic, &im.width, &im.height, &scale) != 4 || magic[0] != 'P' ||
        (magic[1] != 'F' && magic[1] != 'f') || im.width <= 0 || im.height <= 0 || scale == 0) {
        std::fclose(f);
        throw std::runtime_error("not a PFM file: " + path);
    }
    std::fgetc(f);  // the single whitespace byte after the scale
    im.channels = magic[1] == 'F' ? 3 : 1;
    const size_t row = (size_t)im.width * im.channels;
    im.data.resize(row * im.height);
    for (int y = im.height - 1; y >= 0; y--)  // file is bottom-to-top
        if (std::fread(im.data.data() + row * y, sizeof(float), row, f) != row) {
            std::fclose(f);
            throw std::runtime_error("truncated PFM file: " + path);
        }
    std::fclose(f);
    if (scale > 0) {  // big endian payload
        for (float &v : im.data) {
            uint32_t u;
            std::memcpy(&u, &v, 4);
            u = (u >> 24) | ((u >> 8) & 0xFF00u) | ((u << 8) & 0xFF0000u) | (u << 24);
            std::memcpy(&v, &u, 4);
        }
    }
    return im;
}

inline void writePfm(const std::string &path, int width, int height, int channels, const float *data) {
    if (channels != 1 && channels != 3) throw std::runtime_error("PFM holds 1 or 3 channels");
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    std::fprintf(f, "%s\n%d %d\n-1.000000\n", channels == 3 ? "PF" : "Pf", width, height);
    const size_t row = (size_t)width * channels;
    for (int y = height - 1; y >= 0; y--) std::fwrite(data + row * y, sizeof(float), row, f);
    std::fclose(f);
}

}  // namespace statmc
#endif
