// statmc_denoiser.hpp -- header-only C++17 host side above the C ABI of statmc.h.
//
// Mirrors the reference's operator interface for this path, with the OpenCV types swapped for
// minimal stand-alone ones (no OpenCV, no HIP headers needed by the including TU):
//
//   statmc::HostImage / DeviceImage      <- cv::Mat / cv::cuda::GpuMat as used by Buffer
//   statmc::Stream                       <- cv::cuda::Stream          (estimator.h:326)
//   statmc::stat_denoiser::setup / filter<T> / calculateMeanVars<T> / synchronize
//                                        <- cv::cuda::stat_denoiser::* (estimator.h:280,
//                                           estimator.cpp:437-487, 501-521, 572)
//   statmc::Buffer, BufferRegistry, OutputBufferSelection
//                                        <- src/statistics/buffer.h:19-108, buffer.cpp:12-79
//   statmc::StatTypeConfig(s), enums     <- src/statistics/estimator.h:61-102, statpath.h:20-36
//   statmc::makeStatTypeConfigs          <- CreateStatPathIntegrator's rules, statpath.cpp:1013-1173
//   statmc::Estimator                    <- src/statistics/estimator.h:241-380, estimator.cpp:86-289,
//                                           409-489, 571-573 (same member names, same buffer names,
//                                           same aliasing and upload/download sets)
//   statmc::StatTile<T>, Estimator::GetTiles / Merge[Transform]Tile(s)
//                                        <- src/statistics/estimator.h:36-59,147-239, estimator.cpp:297-407:
//                                           the tiles StatPathIntegrator::Render feeds sample by sample
//                                           (statpath.cpp:355-388).  Here a tile RECORDS its samples; the
//                                           merge hands them to the device, where statmc_accumulate_tiles
//                                           runs the Add*Sample* arithmetic (Estimator::EnableDeviceAccumulation).
//
// Error behaviour: the reference has none at these call sites (OpenCV throws cv::Exception);
// here every failing C call throws statmc::Error carrying statmc_last_error().
#ifndef STATMC_DENOISER_HPP
#define STATMC_DENOISER_HPP

#include <algorithm>
#include <array>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <tuple>
#include <vector>

#include "statmc.h"
#include "statmc_bands.hpp"
#include "statmc_pfm.hpp"

namespace statmc {

struct float3 {
    float x, y, z;
};

class Error : public std::runtime_error {
  public:
    Error(int code, const std::string &what) : std::runtime_error(what), code(code) {}
    int code;
};

inline void check(int rc) {
    if (rc != STATMC_OK) throw Error(rc, std::string("statmc: ") + statmc_last_error());
}

enum ImageType { I32C1 = 0, F32C1 = 1, F32C3 = 3 };
inline int channelsOf(ImageType t) { return t == F32C3 ? 3 : 1; }

// Row-major, interleaved, tightly packed host image with shared ownership (copies alias the
// same pixels, like cv::Mat).
class HostImage {
  public:
    HostImage() = default;
    // Page-locked when the HIP runtime can provide it (Upload / Download then run at the full PCIe rate and stay
    // asynchronous on the stream, which pageable cv::Mat storage does not); plain zeroed memory otherwise, e.g. when
    // the buffer catalogue is inspected on a machine without a GPU.
    HostImage(int rows, int cols, ImageType type) : rows(rows), cols(cols), type(type) {
        const size_t n = (size_t)rows * cols * channelsOf(type) * 4;
        void *p = nullptr;
        if (n && statmc_malloc_host(&p, n) == STATMC_OK && p) {
            std::memset(p, 0, n);
            store = std::shared_ptr<void>(p, [](void *q) { statmc_free_host(q); });
            pinned_ = true;
        } else {
            p = std::calloc(n ? n : 1, 1);
            if (!p) throw std::bad_alloc();
            store = std::shared_ptr<void>(p, [](void *q) { std::free(q); });
        }
    }
    int channels() const { return channelsOf(type); }
    size_t bytes() const { return (size_t)rows * cols * channels() * 4; }
    bool empty() const { return !store; }
    bool pinned() const { return pinned_; }
    void *ptr() const { return store.get(); }
    template <class T>
    T *ptr() const { return reinterpret_cast<T *>(ptr()); }
    bool sameStorage(const HostImage &o) const { return store == o.store; }
    int rows = 0, cols = 0;
    ImageType type = F32C1;

  private:
    std::shared_ptr<void> store;
    bool pinned_ = false;
};

class Stream {
  public:
    Stream() = default;                       // the default (null) stream
    explicit Stream(void *handle) : h(handle) {}
    void *handle() const { return h; }
    void waitForCompletion() { check(statmc_synchronize(h)); }

  private:
    void *h = nullptr;
};

// Device image with shared ownership (the GpuMat role).  `allocate == false` builds a
// descriptor without device memory: used to inspect the buffer catalogue without a GPU.
// Opt-in for a host whose samples live on the device (a GPU renderer feeding statmc_accumulate from HBM): device images come
// from statmc_malloc_placed(STATMC_MEM_STATE) and the sample arenas of the device-side accumulation from
// statmc_malloc_placed(STATMC_MEM_STREAM), so that the moments and the samples lie in different interference classes of the card's
// memory (include/statmc.h).  Off by default: the reference's flow uploads 76 B/px per iteration over PCIe and gains
// nothing from it.  Process-wide; set it before the first Estimator is built.
inline bool &usePlacedMemory() {
    static bool on = false;
    return on;
}

class DeviceImage {
  public:
    DeviceImage() = default;
    DeviceImage(int rows, int cols, ImageType type, bool allocate = true) : rows(rows), cols(cols), type(type) {
        if (allocate) {
            void *p = nullptr;
            if (usePlacedMemory()) check(statmc_malloc_placed(&p, bytes(), STATMC_MEM_STATE));
            else check(statmc_malloc(&p, bytes()));
            mem = std::shared_ptr<void>(p, [](void *q) { statmc_free(q); });
        }
    }
    int channels() const { return channelsOf(type); }
    size_t bytes() const { return (size_t)rows * cols * channels() * 4; }
    void *data() const { return mem.get(); }
    bool sameStorage(const DeviceImage &o) const { return mem && mem == o.mem; }
    statmc_image desc() const { return statmc_image{mem.get(), (size_t)cols * channels() * 4, cols, rows}; }
    void upload(const HostImage &m, Stream &s) { check(statmc_upload(mem.get(), m.ptr(), bytes(), s.handle())); }
    void download(HostImage &m, Stream &s) const { check(statmc_download(m.ptr(), mem.get(), bytes(), s.handle())); }
    int rows = 0, cols = 0;
    ImageType type = F32C1;

  private:
    std::shared_ptr<void> mem;
};

// ------------------------------------------------------------------------------------------
namespace stat_denoiser {

inline void setup(int device = 0) { check(statmc_setup(device)); }
inline void synchronize(Stream &s) { check(statmc_synchronize(s.handle())); }
// The reference fixes the significance level at compile time by choosing one of three
// `t_quantiles` tables (README.md:149,158); here it is a run-time setting: 0 -> 0.005 (default),
// 1 -> 0.002, 2 -> 0.05, and a table can be replaced by the caller's own quantiles (dof 1..n).
inline void setSignificance(int alphaIndex) { check(statmc_set_significance(alphaIndex)); }
inline void setTQuantiles(int alphaIndex, const std::vector<float> &q) {
    check(statmc_set_t_quantiles(alphaIndex, q.data(), (int)q.size()));
}
// What the tree does not fix about filter<T> (statmc_filter_spec in statmc.h): gate form, channel rule,
// one- or two-sided quantile, per-pixel or Welch dof, border policy, n < 2.  Default-constructed = spec v2.
inline void setFilterSpec(const statmc_filter_spec &spec) { check(statmc_set_filter_spec(&spec)); }
inline statmc_filter_spec getFilterSpec() {
    statmc_filter_spec s;
    check(statmc_get_filter_spec(&s));
    return s;
}
// "gate=asym,channels=joint,sides=one,dof=welch,border=clamp,small_n=exclude" (any subset, any order;
// the other value of each field is its default: sym / and / two / pixel / clip / accept; gate has a third value, centre)
inline statmc_filter_spec parseFilterSpec(const std::string &text) {
    statmc_filter_spec s = {0, 0, 0, 0, 0, 0};
    size_t pos = 0;
    while (pos < text.size()) {
        size_t end = text.find(',', pos);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        if (eq == std::string::npos) throw std::runtime_error("filter spec: expected key=value, got '" + item + "'");
        const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
        auto pick = [&](const char *zero, const char *one) -> int32_t {
            if (v == zero || v == "0") return 0;
            if (v == one || v == "1") return 1;
            throw std::runtime_error("filter spec: " + k + " is '" + zero + "' or '" + one + "', not '" + v + "'");
        };
        if (k == "gate") s.gate = (v == "centre" || v == "2") ? 2 : pick("sym", "asym");   // centre: Moon et al. 2013 (-DMEMFNC=1)
        else if (k == "channels") s.channel_rule = pick("and", "joint");
        else if (k == "sides") s.sides = pick("two", "one");
        else if (k == "dof") s.dof = pick("pixel", "welch");
        else if (k == "border") s.border = pick("clip", "clamp");
        else if (k == "small_n") s.small_n = pick("accept", "exclude");
        else throw std::runtime_error("filter spec: unknown field '" + k + "'");
    }
    return s;
}

namespace detail {
inline std::vector<statmc_image> descs(const std::vector<DeviceImage> &v) {
    std::vector<statmc_image> d;
    d.reserve(v.size());
    for (const auto &i : v) d.push_back(i.desc());
    return d;
}
template <class T>
struct channels;
template <>
struct channels<float> { static constexpr int value = 1; };
template <>
struct channels<float3> { static constexpr int value = 3; };
}  // namespace detail

// The argument block of one filter<T> call with the descriptor tables it points into.
struct FilterCall {
    std::vector<statmc_image> n, mean, m2, m3, film, g, mc, dc, ff;
    statmc_filter_args a;
    int channels = 3;
    void prepassRows(int y0, int y1) const { bands::prepassRows(a, channels, y0, y1); }   // rows [y0, y1) only
    void filterRows(int y0, int y1) const { bands::filterRows(a, channels, y0, y1); }     // output rows [y0, y1)
    void run() const { check(channels == 3 ? statmc_filter_f32x3(&a) : statmc_filter_f32(&a)); }
};

// Argument order of cv::cuda::stat_denoiser::filter<T> (estimator.cpp:437-459).  Where the
// reference passes a GpuMat holding a device array of PtrStepSzb, this takes the images.
template <class T>
void fillFilterCall(FilterCall &c, unsigned char nBuffers, unsigned short width, unsigned short height, float filterDSFactor,
                    unsigned char filterRadius, bool denoiseFilm, const std::vector<DeviceImage> &nPtrs,
                    const std::vector<DeviceImage> &meanPtrs, const std::vector<DeviceImage> &m2Ptrs,
                    const std::vector<DeviceImage> &m3Ptrs, const std::vector<DeviceImage> &filmPtrs,
                    const DeviceImage &filmBuffer, const std::vector<DeviceImage> &gBufferPtrs,
                    const std::vector<unsigned char> &gBufferChannelCounts, const std::vector<float> &gBufferDRFactors,
                    size_t nGBuffers, const std::vector<DeviceImage> &meanCorrPtrs,
                    const std::vector<DeviceImage> &discriminatorPtrs, const std::vector<DeviceImage> &filmFilteredPtrs,
                    const DeviceImage &filmFilteredBuffer, Stream &stream) {
    c.n = detail::descs(nPtrs); c.mean = detail::descs(meanPtrs); c.m2 = detail::descs(m2Ptrs);
    c.m3 = detail::descs(m3Ptrs); c.film = detail::descs(filmPtrs); c.g = detail::descs(gBufferPtrs);
    c.mc = detail::descs(meanCorrPtrs); c.dc = detail::descs(discriminatorPtrs); c.ff = detail::descs(filmFilteredPtrs);
    c.channels = detail::channels<T>::value;
    statmc_filter_args &a = c.a;
    std::memset(&a, 0, sizeof(a));
    a.n_buffers = nBuffers;
    a.width = width;
    a.height = height;
    a.filter_ds_factor = filterDSFactor;
    a.filter_radius = filterRadius;
    a.denoise_film = denoiseFilm ? 1 : 0;
    a.n = c.n.data();
    a.mean = c.mean.data();
    a.m2 = c.m2.data();
    a.m3 = c.m3.data();
    a.film = c.film.data();
    a.film_buffer = filmBuffer.desc();
    a.g_buffers = c.g.data();
    a.g_channel_counts = gBufferChannelCounts.data();
    a.g_dr_factors = gBufferDRFactors.data();
    a.n_g_buffers = nGBuffers;
    a.mean_corr = c.mc.data();
    a.discriminator = c.dc.data();
    a.film_filtered = c.ff.data();
    a.film_filtered_buffer = filmFilteredBuffer.desc();
    a.stream = stream.handle();
}
template <class T>
void filter(unsigned char nBuffers, unsigned short width, unsigned short height, float filterDSFactor,
            unsigned char filterRadius, bool denoiseFilm, const std::vector<DeviceImage> &nPtrs,
            const std::vector<DeviceImage> &meanPtrs, const std::vector<DeviceImage> &m2Ptrs,
            const std::vector<DeviceImage> &m3Ptrs, const std::vector<DeviceImage> &filmPtrs,
            const DeviceImage &filmBuffer, const std::vector<DeviceImage> &gBufferPtrs,
            const std::vector<unsigned char> &gBufferChannelCounts, const std::vector<float> &gBufferDRFactors,
            size_t nGBuffers, const std::vector<DeviceImage> &meanCorrPtrs,
            const std::vector<DeviceImage> &discriminatorPtrs, const std::vector<DeviceImage> &filmFilteredPtrs,
            const DeviceImage &filmFilteredBuffer, Stream &stream) {
    FilterCall c;
    fillFilterCall<T>(c, nBuffers, width, height, filterDSFactor, filterRadius, denoiseFilm, nPtrs, meanPtrs, m2Ptrs, m3Ptrs,
                      filmPtrs, filmBuffer, gBufferPtrs, gBufferChannelCounts, gBufferDRFactors, nGBuffers, meanCorrPtrs,
                      discriminatorPtrs, filmFilteredPtrs, filmFilteredBuffer, stream);
    c.run();
}

// cv::cuda::stat_denoiser::calculateMeanVars<T> (estimator.cpp:501-521, commented-out call).
template <class T>
void calculateMeanVars(unsigned char nBuffers, unsigned short width, unsigned short height,
                       const std::vector<DeviceImage> &nPtrs, const std::vector<DeviceImage> &filmM2Ptrs,
                       const std::vector<DeviceImage> &filmVarPtrs, Stream &stream, bool rowNQuirk = true) {
    const auto n = detail::descs(nPtrs), m2 = detail::descs(filmM2Ptrs), var = detail::descs(filmVarPtrs);
    check(statmc_calculate_mean_vars(nBuffers, width, height, detail::channels<T>::value, n.data(), m2.data(),
                                     var.data(), rowNQuirk ? 1 : 0, stream.handle()));
}

}  // namespace stat_denoiser

// ------------------------------------------------------------------------------------------
// src/statistics/buffer.h:19-71
class Buffer {
  public:
    Buffer() {}
    Buffer(const std::string &name, HostImage mat, bool allocateDevice = true)
        : Buffer(name, mat, DeviceImage(mat.rows, mat.cols, mat.type, allocateDevice)) {}
    Buffer(const std::string &name, HostImage mat, DeviceImage gpuMat) : name(name), mat(mat), gpuMat(gpuMat) {
        if (mat.channels() == 1)
            channelNames = {name};
        else
            channelNames = {name + ".R", name + ".G", name + ".B"};
    }
    void upload(Stream &stream) { gpuMat.upload(mat, stream); }
    void download(Stream &stream) { gpuMat.download(mat, stream); }
    const uint8_t *matPtr() const { return mat.ptr<uint8_t>(); }

    std::string name;
    std::vector<std::string> channelNames;
    HostImage mat;
    DeviceImage gpuMat;
};

// src/statistics/buffer.h:73-80
class BufferRegistry {
  public:
    explicit BufferRegistry(const Buffer &filmBuffer) { buffers.push_back(filmBuffer); }
    void Register(const Buffer &b) { buffers.push_back(b); }
    const Buffer *find(const std::string &name) const {
        for (const auto &b : buffers)
            if (b.name == name) return &b;
        return nullptr;
    }
    std::vector<Buffer> buffers;
};

// src/statistics/buffer.h:82-108, buffer.cpp:12-53: the buffers of a registry whose names match a
// regular expression (the integrator's "outputregex", statpath.cpp:1003-1011), written as
// "<stem>[-<suffix>]-<buffer name>.<ext>".  The reference writes through cv::imwrite (any format
// OpenCV knows; the statistics workflow uses .pfm, README.md "outputregex"); this build writes PFM
// only.  Display() (tev, src/display) is out of scope.
class OutputBufferSelection {
  public:
    OutputBufferSelection(const BufferRegistry &reg, const std::string &filename) {
        SetFilename(filename);
        for (const Buffer &b : reg.buffers) buffers.push_back(b);
    }
    OutputBufferSelection(const BufferRegistry &reg, const std::regex &regex, const std::string &filename) {
        SetFilename(filename);
        for (const Buffer &b : reg.buffers)
            if (std::regex_match(b.name, regex)) buffers.push_back(b);
    }
    // buffer.cpp:31-35: everything that is not float (the int32 counts) is converted to float
    void PrepareOutput() const {
        outMats.assign(buffers.size(), std::vector<float>());
        for (size_t i = 0; i < buffers.size(); i++) {
            const Buffer &b = buffers[i];
            if (b.mat.type != I32C1) continue;
            const int32_t *src = b.mat.ptr<int32_t>();
            outMats[i].resize((size_t)b.mat.rows * b.mat.cols);
            for (size_t k = 0; k < outMats[i].size(); k++) outMats[i][k] = (float)src[k];
        }
    }
    void Write(const std::string &filenameSuffix = "") const {  // buffer.cpp:37-53
        if (filenameExtension != "pfm") throw Error(STATMC_ERR_UNSUPPORTED, "OutputBufferSelection::Write: only .pfm is written");
        if (outMats.size() != buffers.size()) PrepareOutput();
        for (size_t i = 0; i < buffers.size(); i++) {
            const Buffer &b = buffers[i];
            const std::string filename = (filenameSuffix.empty() ? filenameStem : filenameStem + "-" + filenameSuffix) + "-" +
                                         b.name + "." + filenameExtension;
            const float *data = b.mat.type == I32C1 ? outMats[i].data() : b.mat.ptr<float>();
            writePfm(filename, b.mat.cols, b.mat.rows, b.mat.channels(), data);
        }
    }
    std::string GetFilenameStem() const { return filenameStem; }
    const std::vector<Buffer> &selected() const { return buffers; }

  private:
    void SetFilename(const std::string &filename) {  // buffer.cpp:75-79
        const size_t pos = filename.find_last_of(".");
        filenameStem = filename.substr(0, pos);
        filenameExtension = pos == std::string::npos ? std::string() : filename.substr(pos + 1);
    }
    std::vector<Buffer> buffers;
    mutable std::vector<std::vector<float>> outMats;
    std::string filenameStem, filenameExtension;
};

// src/statistics/statpath.h:28-36 and estimator.h:68-71
enum StatTypeIndex {
    Radiance = 0, MISBSDFWinRate = 1, MISLightWinRate = 2, StatMaterialID = 3, StatDepth = 4,
    StatNormal = 5, StatAlbedo = 6, ItRadiance = 7
};
enum CUDAGroupIndex { DenoiseGroup = 0, CalculateMeanVarianceGroup = 1 };
static constexpr unsigned char nCUDAGroupIndices = 2;

// src/statistics/estimator.h:74-90
struct StatTypeConfig {
    unsigned char type = 0;
    unsigned char index = 0;
    bool enable = false;
    unsigned char nBounces = 0, bounceStart = 0, bounceEnd = 0;
    unsigned char nChannels = 1;
    bool transform = false;
    unsigned char maxMoment = 1;
    bool gBuffer = false;
    bool enableForFilter = false;
    float filterSD = 0.f;
    std::vector<unsigned char> cudaGroups;
};

struct StatTypeConfigs {
    StatTypeConfig &operator[](size_t i) { return configs[i]; }
    const StatTypeConfig &operator[](size_t i) const { return configs[i]; }
    unsigned char nEnabled = 0;
    std::vector<StatTypeConfig> configs;
};

// The Integrator "statpath" parameters that shape the statistics path (statpath.cpp:966-1001).
struct StatPathParams {
    int maxDepth = 5;
    int trackedBounces = -1;  // default: maxDepth
    bool multiChannelStats = true;
    bool acrr = false, smis = false, calcProDenStats = false, calcMoonStats = false, calcGBuffers = false,
         calcStats = false, denoiseImage = false, calcItStats = false;
    float filterSD = 10.f;
    unsigned char filterRadius = 20;
    std::vector<std::string> filterBuffers = {"albedo", "normal"};
    std::vector<float> filterBufferSDs = {0.02f, 0.1f};
};

// CreateStatPathIntegrator's configuration rules, statpath.cpp:1013-1173.  One deliberate
// deviation: the float G-buffers (materialid, depth) get their own slot
// counter instead of borrowing the RGB one.
inline StatTypeConfigs makeStatTypeConfigs(const StatPathParams &p) {
    if (p.filterBuffers.size() != p.filterBufferSDs.size())
        throw Error(STATMC_ERR_INVALID, "Size of filterbuffers and filterbuffersds must match.");  // statpath.cpp:1090-1093
    const unsigned char nTracked = (unsigned char)(p.trackedBounces >= 0 ? p.trackedBounces : p.maxDepth);
    StatTypeConfigs c;
    c.configs.assign(8, StatTypeConfig());
    if (p.acrr || p.calcProDenStats || p.denoiseImage || p.calcStats || p.calcMoonStats) {
        auto &cfg = c[Radiance];
        cfg.type = Radiance;
        cfg.index = c.nEnabled++;
        cfg.enable = true;
        cfg.bounceStart = 0;
        cfg.bounceEnd = p.acrr ? nTracked : 1;
        cfg.nBounces = cfg.bounceEnd - cfg.bounceStart;
        if (p.multiChannelStats) cfg.nChannels = 3;
        if (p.calcProDenStats || p.calcMoonStats) cfg.maxMoment = 2;
        if (p.acrr || p.denoiseImage || p.calcStats) {
            cfg.transform = true;
            cfg.maxMoment = 3;
        }
        if (p.acrr || p.denoiseImage) cfg.cudaGroups.push_back(DenoiseGroup);
        if (p.calcProDenStats) cfg.cudaGroups.push_back(CalculateMeanVarianceGroup);
    }
    if (p.smis) {
        for (unsigned char t : {(unsigned char)MISBSDFWinRate, (unsigned char)MISLightWinRate}) {
            auto &cfg = c[t];
            cfg.type = t;
            cfg.index = c.nEnabled++;
            cfg.enable = true;
            cfg.bounceStart = 0;
            cfg.bounceEnd = nTracked;
            cfg.nBounces = nTracked;
            cfg.nChannels = 1;
            cfg.transform = false;
            cfg.maxMoment = 3;
            cfg.cudaGroups.push_back(DenoiseGroup);
        }
    }
    if (p.acrr || p.denoiseImage || p.smis || p.calcProDenStats || p.calcGBuffers || p.calcStats || p.calcMoonStats) {
        struct G { const char *name; unsigned char type, channels; };
        for (const G &g : {G{"materialid", StatMaterialID, 1}, G{"depth", StatDepth, 1}, G{"normal", StatNormal, 3},
                           G{"albedo", StatAlbedo, 3}}) {
            auto &cfg = c[g.type];
            const auto it = std::find(p.filterBuffers.begin(), p.filterBuffers.end(), g.name);
            if (it == p.filterBuffers.end()) continue;
            cfg.enable = true;
            if (p.acrr || p.denoiseImage || p.smis) {
                cfg.enableForFilter = true;
                cfg.filterSD = p.filterBufferSDs[it - p.filterBuffers.begin()];
            }
            cfg.type = g.type;
            cfg.index = c.nEnabled++;
            cfg.bounceStart = 0;
            cfg.bounceEnd = 1;
            cfg.nBounces = 1;
            cfg.nChannels = g.channels;
            cfg.gBuffer = true;
            cfg.transform = false;
            cfg.maxMoment = 1;
            if (p.calcProDenStats) {
                cfg.maxMoment = 2;
                cfg.cudaGroups.push_back(CalculateMeanVarianceGroup);
            }
        }
    }
    if (p.calcItStats) {
        auto &cfg = c[ItRadiance];
        cfg.type = ItRadiance;
        cfg.index = c.nEnabled++;
        cfg.enable = true;
        cfg.bounceStart = 0;
        cfg.bounceEnd = 1;
        cfg.nBounces = 1;
        cfg.nChannels = 3;
        cfg.transform = false;
        cfg.maxMoment = 2;
    }
    return c;
}

// ------------------------------------------------------------------------------------------
// pbrt's Point2i / Bounds2i (src/core/geometry.h) as far as the tile interface uses them.
struct Point2i {
    Point2i() = default;
    Point2i(int x, int y) : x(x), y(y) {}
    int x = 0, y = 0;
};
struct Bounds2i {  // pMax exclusive
    Bounds2i() = default;
    Bounds2i(const Point2i &a, const Point2i &b) : pMin(a), pMax(b) {}
    int Area() const { return (pMax.x - pMin.x) * (pMax.y - pMin.y); }
    Point2i pMin, pMax;
};
struct Vector2f {
    Vector2f() = default;
    Vector2f(float x, float y) : x(x), y(y) {}
    float x = 0.f, y = 0.f;
};
using Vec3 = float3;  // statpbrt.h:22

// StatTile<T> (estimator.h:147-239) with the reference's constructor and Add*Sample* names, so
// that Render<T>'s member-function pointers (statpath.cpp:97-116,166-170) bind unchanged.  The
// reference updates StatTilePixel moments on the calling thread; this tile only records the
// sample -- planes[s][pixel], sample-major, already the block layout of statmc_accumulate_tiles --
// and the moments are updated on the device when the tile is merged.  Which of the six methods is
// called does not matter here: the stat type's (transform, maxMoment) is applied at the merge, as
// GetAddSampleFn derives the method from that same configuration.
template <typename T>
class StatTile {
  public:
    explicit StatTile(const Bounds2i &pixelBounds)
        : pixelBounds(pixelBounds), tileWidth(std::max(0, pixelBounds.pMax.x - pixelBounds.pMin.x)),
          nPixels((size_t)tileWidth * std::max(0, pixelBounds.pMax.y - pixelBounds.pMin.y)), counts(nPixels, 0u) {}
    // The tile GetTilesF hands out (estimator.h:152-161): it also carries the pixel reconstruction filter.  None of
    // the reference's Add*Sample* methods reads it (estimator.h:162-232), so it is kept and not interpreted here either.
    StatTile(const Bounds2i &pixelBounds, const Vector2f &filterRadius, const float *filterTable, int filterTableSize)
        : StatTile(pixelBounds) {
        this->filterRadius = filterRadius;
        this->filterTable = filterTable;
        this->filterTableSize = filterTableSize;
    }
    Vector2f GetFilterRadius() const { return filterRadius; }
    const float *GetFilterTable() const { return filterTable; }
    int GetFilterTableSize() const { return filterTableSize; }
    void AddSampleM1(const Point2i p, const T sample) { record(p, sample); }
    void AddSampleM2(const Point2i p, const T sample) { record(p, sample); }
    void AddSampleM3(const Point2i p, const T sample) { record(p, sample); }
    void AddTransformSampleM1(const Point2i p, const T sample) { record(p, sample); }
    void AddTransformSampleM2(const Point2i p, const T sample) { record(p, sample); }
    void AddTransformSampleM3(const Point2i p, const T sample) { record(p, sample); }
    Bounds2i GetPixelBounds() const { return pixelBounds; }
    // samples recorded for pixel p since the last merge
    unsigned pending(const Point2i p) const { return counts[index(p)]; }

  private:
    friend class Estimator;
    size_t index(const Point2i p) const {
        return (size_t)(p.y - pixelBounds.pMin.y) * tileWidth + (size_t)(p.x - pixelBounds.pMin.x);
    }
    void record(const Point2i p, const T &sample) {
        const size_t i = index(p);
        const uint32_t s = counts[i]++;
        if (s >= planes.size()) planes.emplace_back(nPixels);
        planes[s][i] = sample;
    }
    Bounds2i pixelBounds;
    Vector2f filterRadius;
    const float *filterTable = nullptr;
    int filterTableSize = 0;
    int tileWidth;
    size_t nPixels;
    // the merge interface is const in the reference (estimator.h:290-301); emptying the recorder is
    // not part of the tile's observable state
    mutable std::vector<uint32_t> counts;
    std::vector<std::vector<T>> planes;
};

// ------------------------------------------------------------------------------------------
// src/statistics/estimator.h:241-380.  Public members keep the reference's names because
// StatPathIntegrator indexes them directly (statpath.cpp:308-311, 504-511).
class Estimator {
  public:
    Estimator(const Buffer &filmBuffer, const StatTypeConfigs &statTypeConfigs, const float filterSD,
              const unsigned char filterRadius, const bool denoiseFilm, const bool acrrEnabled, const bool smisEnabled,
              BufferRegistry &reg, bool allocateDevice = true, int device = 0)
        : width((unsigned short)filmBuffer.mat.cols), height((unsigned short)filmBuffer.mat.rows),
          filterDSFactor(-.5f / (filterSD * filterSD)), filterRadius(filterRadius), denoiseFilm(denoiseFilm),
          acrrEnabled(acrrEnabled), smisEnabled(smisEnabled), filmBuffer(filmBuffer),
          filmFilteredBuffer("film-f", HostImage(filmBuffer.mat.rows, filmBuffer.mat.cols, F32C3), allocateDevice),
          allocateDevice(allocateDevice), device(device) {
        floatBufferCounts.assign(nCUDAGroupIndices, 0);
        rgbBufferCounts.assign(nCUDAGroupIndices, 0);
        // only the enabled configs are kept (estimator.h:269-270)
        for (const auto &cfg : statTypeConfigs.configs)
            if (cfg.enable) this->statTypeConfigs.configs.push_back(cfg);
        this->statTypeConfigs.nEnabled = (unsigned char)this->statTypeConfigs.configs.size();
        reg.Register(filmFilteredBuffer);
        if (denoiseFilm) {
            uploadBuffers.push_back(&this->filmBuffer);
            downloadBuffers.push_back(&this->filmFilteredBuffer);
        }
        if (allocateDevice) {
            stat_denoiser::setup(device);
            deviceCUs = statmc_device_cus();      // of THIS Estimator's device (setup made it current): what the band plan is fitted to
        }
    }
    Estimator(const Estimator &) = delete;
    Estimator &operator=(const Estimator &) = delete;

    void RegisterGBuffer(Buffer &b, const float filterSD) {  // estimator.cpp:14-17
        gBuffers.push_back(b);
        gBufferDRFactors.emplace_back(-.5f / (filterSD * filterSD));
    }

    // estimator.cpp:86-289: 10 named images per (type, bounce), aliasing rules, upload / download
    // sets, per-group buffer counts and the per-group image tables handed to filter<T>.
    void AllocateBuffers(BufferRegistry &reg) {
        auto &cfgs = statTypeConfigs;
        const size_t nT = cfgs.nEnabled;
        for (auto *v : {&nBuffers, &meanBuffers, &m2Buffers, &m3Buffers, &meanCorrBuffers, &discriminatorBuffers,
                        &filmBuffers, &filmFilteredBuffers, &filmM2Buffers, &filmVarBuffers})
            v->assign(nT, std::vector<Buffer>());
        // raw Buffer* go into the upload/download sets: reserve so they stay valid (estimator.cpp:22)
        for (size_t i = 0; i < nT; i++)
            for (auto *v : {&nBuffers, &meanBuffers, &m2Buffers, &m3Buffers, &meanCorrBuffers, &discriminatorBuffers,
                            &filmBuffers, &filmFilteredBuffers, &filmM2Buffers, &filmVarBuffers})
                (*v)[i].reserve(cfgs.configs[i].nBounces);

        auto has = [](const StatTypeConfig &c, unsigned char g) {
            return std::find(c.cudaGroups.begin(), c.cudaGroups.end(), g) != c.cudaGroups.end();
        };
        for (unsigned char i = 0; i < cfgs.nEnabled; i++) {
            auto &cfg = cfgs.configs[i];
            const ImageType T = cfg.nChannels == 3 ? F32C3 : F32C1;
            for (unsigned char j = cfg.bounceStart; j < cfg.bounceEnd; j++) {
                const std::string pre = "t" + std::to_string(i) + "-b" + std::to_string(j);
                auto alloc = [&](std::vector<std::vector<Buffer>> &bufs, const char *suffix, HostImage mat) -> Buffer & {
                    bufs[cfg.index].emplace_back(pre + suffix, mat, allocateDevice);
                    reg.Register(bufs[cfg.index].back());
                    return bufs[cfg.index].back();
                };
                auto allocGpu = [&](std::vector<std::vector<Buffer>> &bufs, const char *suffix, HostImage mat,
                                    DeviceImage gpu) -> Buffer & {
                    bufs[cfg.index].emplace_back(pre + suffix, mat, gpu);
                    reg.Register(bufs[cfg.index].back());
                    return bufs[cfg.index].back();
                };
                alloc(nBuffers, "-n", HostImage(height, width, I32C1));
                if (cfg.transform) {
                    alloc(meanBuffers, "-mean", HostImage(height, width, T));
                    alloc(m2Buffers, "-m2", HostImage(height, width, T));
                    alloc(filmBuffers, "-film-mean", HostImage(height, width, T));
                    alloc(filmM2Buffers, "-film-m2", HostImage(height, width, T));
                } else {  // mean / m2 share storage with their film counterparts (estimator.cpp:127-137)
                    HostImage mean(height, width, T), m2(height, width, T);
                    DeviceImage meanGPU(height, width, T, allocateDevice), m2GPU(height, width, T, allocateDevice);
                    allocGpu(meanBuffers, "-mean", mean, meanGPU);
                    allocGpu(m2Buffers, "-m2", m2, m2GPU);
                    allocGpu(filmBuffers, "-film-mean", mean, meanGPU);
                    allocGpu(filmM2Buffers, "-film-m2", m2, m2GPU);
                }
                alloc(m3Buffers, "-m3", HostImage(height, width, T));
                alloc(meanCorrBuffers, "-mean-corr", HostImage(height, width, T));
                alloc(discriminatorBuffers, "-discriminator", HostImage(height, width, T));
                alloc(filmVarBuffers, "-film-mean-var", HostImage(height, width, T));
                // radiance, bounce 0 with denoiseFilm: host image shared with "film-f", own device image
                // (estimator.cpp:143-146)
                if (cfg.nChannels == 3 && denoiseFilm && cfg.type == Radiance && j == 0)
                    alloc(filmFilteredBuffers, "-film-mean-f", filmFilteredBuffer.mat);
                else
                    alloc(filmFilteredBuffers, "-film-mean-f", HostImage(height, width, T));

                const unsigned char jj = j - cfg.bounceStart;
                if (cfg.gBuffer && cfg.enableForFilter) {
                    RegisterGBuffer(filmBuffers[i][jj], cfg.filterSD);
                    addUnique(uploadBuffers, &filmBuffers[i][jj]);
                }
                auto &counts = cfg.nChannels == 3 ? rgbBufferCounts : floatBufferCounts;
                for (unsigned char k : cfg.cudaGroups)
                    if (k != CalculateMeanVarianceGroup) {
                        counts[k]++;
                        runCUDA = true;
                    } else if (j == 0) {
                        counts[k]++;
                    }
                if (has(cfg, DenoiseGroup)) {
                    addUnique(uploadBuffers, &nBuffers[i][jj]);
                    addUnique(uploadBuffers, &meanBuffers[i][jj]);
                    addUnique(uploadBuffers, &m2Buffers[i][jj]);
                    addUnique(uploadBuffers, &m3Buffers[i][jj]);
                    const bool coveredByFilm = cfg.nChannels == 3 && denoiseFilm && cfg.type == Radiance && j == 0;
                    const bool moveFilm = cfg.nChannels == 3 ? !coveredByFilm : (acrrEnabled || smisEnabled);
                    if (moveFilm) {  // estimator.cpp:168-172 (RGB), 225-229 (float)
                        if (cfg.transform) addUnique(uploadBuffers, &filmBuffers[i][jj]);
                        addUnique(downloadBuffers, &filmFilteredBuffers[i][jj]);
                    }
                }
                if (has(cfg, CalculateMeanVarianceGroup) && j == 0) {
                    addUnique(uploadBuffers, &nBuffers[i][jj]);
                    addUnique(uploadBuffers, &filmM2Buffers[i][jj]);
                    addUnique(downloadBuffers, &filmVarBuffers[i][jj]);
                }
            }
        }
        // per-group image tables (PREPARE_STAT_BUFFER_GPU_PTRS, estimator.cpp:35-69)
        auto table = [&](const std::vector<std::vector<Buffer>> &bufs, unsigned char channels, unsigned char group) {
            std::vector<DeviceImage> t;
            for (unsigned char i = 0; i < cfgs.nEnabled; i++) {
                const auto &cfg = cfgs.configs[i];
                if (cfg.nChannels != channels) continue;
                for (unsigned char k : cfg.cudaGroups) {
                    if (k != group) continue;
                    if (k != CalculateMeanVarianceGroup)
                        for (unsigned char j = 0; j < cfg.nBounces; j++) t.push_back(bufs[i][j].gpuMat);
                    else
                        t.push_back(bufs[i][0].gpuMat);
                }
            }
            return t;
        };
        for (unsigned char g = 0; g < nCUDAGroupIndices; g++) {
            for (int rgb = 0; rgb < 2; rgb++) {
                Tables &t = rgb ? rgbTables[g] : floatTables[g];
                const unsigned char ch = rgb ? 3 : 1;
                t.n = table(nBuffers, ch, g);
                t.mean = table(meanBuffers, ch, g);
                t.m2 = table(m2Buffers, ch, g);
                t.m3 = table(m3Buffers, ch, g);
                t.film = table(filmBuffers, ch, g);
                t.filmM2 = table(filmM2Buffers, ch, g);
                t.filmVar = table(filmVarBuffers, ch, g);
                t.meanCorr = table(meanCorrBuffers, ch, g);
                t.discriminator = table(discriminatorBuffers, ch, g);
                t.filmFiltered = table(filmFilteredBuffers, ch, g);
            }
        }
        gBufferImages.clear();
        gBufferChannelCounts.clear();
        for (const auto &b : gBuffers) {  // PREPARE_G_BUFFER_GPU_PTRS, estimator.cpp:72-84
            gBufferImages.push_back(b.gpuMat);
            gBufferChannelCounts.push_back((unsigned char)b.gpuMat.channels());
        }
    }

    // Upload / Denoise / Download / Synchronize (estimator.cpp:409-489, 571-573; the sequence StatPathIntegrator times
    // as "CUDA time", statpath.cpp:409-417).  All four only enqueue.  With more than one pipeline band (SetPipelineBands;
    // automatic: for images of 512 rows and more, bands fitted to the window filter's rounds -- bands::plan) the image is cut into bands of rows and the three phases
    // run on their own streams (copies in on two) ordered by events: band k travels with the r rows below it (its lower halo), is pre-passed
    // and filtered as soon as it has arrived, and copied back as soon as it is filtered -- the PCIe copies in both
    // directions and the kernels overlap instead of adding up; what is left after the last copy in is one band's
    // filter.  Results are the same bits as without bands: the pre-pass is per pixel, and the window filter forms a
    // pixel's sums in the same order for any output region.
    void SetPipelineBands(int n) { bandsRequested = n; }   // 0 = automatic, 1 = off
    // Copy queues the transfers of a band are dealt over (statmc_bands.hpp): 1 by default -- 3.9 ms for the 1080p
    // bracket, every iteration.  2 (SetUploadQueues / STATMC_UPLOAD_QUEUES=2) gives 3.5 ms when nothing stalls, but about
    // one iteration in twelve then takes 7 - 10 ms because the host thread blocks inside a hipMemcpyAsync enqueue
    // (statmc_cv.hpp, note (b); tools/experiments/iter_times.py): 4.1 ms on average, with a tail.
    // (Two further transports -- a pulling kernel as the second queue, or instead of the copy engine altogether -- were
    // measured in round 3 and gained nothing, profiles/r03_upload_modes.log; they are no longer part of the library.)
    void SetUploadQueues(int n) { uploadQueues = n < 2 ? 1 : 2; }
    int PipelineBands() const {
        if (!allocateDevice || acc.dry) return 1;
        return bandPlan().count();
    }
    void Upload() {  // estimator.cpp:409-416
        if (acc.enabled) FlushSamples();  // statistics are produced on the device: only the rest moves
        if (acc.dry) return;
        const int nb = PipelineBands();
        if (nb <= 1) {
            for (Buffer *b : uploadBuffers)
                if (!acc.enabled || !acc.deviceProduced.count(b)) b->upload(stream);
            return;
        }
        ensurePipeline(nb);
        // the copies overwrite images that earlier work on `stream` may still read
        pipe.beginUploads(stream.handle());
        std::vector<Buffer *> moving;
        std::vector<size_t> rowBytes;
        for (Buffer *b : uploadBuffers) {
            if (acc.enabled && acc.deviceProduced.count(b)) continue;
            moving.push_back(b);
            rowBytes.push_back((size_t)b->mat.cols * b->mat.channels() * 4);
        }
        const std::vector<int> queue = bands::Streams::deal(rowBytes, uploadQueues);
        for (int k = 0; k < nb; k++) {
            const int y0 = arrivalEdge(k, nb), y1 = arrivalEdge(k + 1, nb);
            pipe.beginTransfer(k);
            for (size_t i = 0; i < moving.size(); i++) {
                Buffer *b = moving[i];
                const size_t row = rowBytes[i];
                pipe.upload(queue[i], static_cast<char *>(b->gpuMat.data()) + y0 * row, b->mat.ptr<char>() + y0 * row,
                            (size_t)(y1 - y0) * row);
            }
            pipe.markArrived(k);
        }
        pipe.uploaded = pipe.pendingJoin = nb;
    }
    void Download() {  // estimator.cpp:418-425
        joinUploads();
        const int nb = pipe.denoised;
        pipe.denoised = 0;
        for (Buffer *b : downloadBuffers) {
            if (nb <= 1 || !pipe.outputs.count(b)) {
                b->download(stream);   // behind everything enqueued on `stream`
                continue;
            }
        }
        if (nb <= 1) return;
        for (int k = 0; k < nb; k++) {
            const int y0 = bandEdge(k, nb), y1 = bandEdge(k + 1, nb);
            check(statmc_stream_wait_event(pipe.down, pipe.filtered[k]));
            for (Buffer *b : downloadBuffers) {
                if (!pipe.outputs.count(b)) continue;
                const size_t row = (size_t)b->mat.cols * b->mat.channels() * 4;
                check(statmc_download(b->mat.ptr<char>() + y0 * row, static_cast<char *>(b->gpuMat.data()) + y0 * row,
                                      (size_t)(y1 - y0) * row, pipe.down));
            }
        }
        pipe.downloading = true;
    }
    // Placed memory (usePlacedMemory()): hands the GiB slots of the classes nobody asked for back to the driver
    // (statmc_placement_trim; synchronises the device).  Denoise() calls it once, after the first iteration's images and sample
    // arenas have been dealt; a host whose arenas keep growing (the progressive schedule doubles them) may call it again later.
    int TrimPlacedMemory() {
        if (!usePlacedMemory()) return 0;
        check(statmc_set_device(device));
        const int n = statmc_placement_trim();
        if (n < 0) check(n);
        placementTrimmed = true;
        return n;
    }
    bool placementTrimmed = false;

    void Denoise() {  // estimator.cpp:427-489
        if (usePlacedMemory() && !placementTrimmed) TrimPlacedMemory();   // (the trim synchronises the device itself)
        stat_denoiser::FilterCall calls[2];
        int nCalls = 0;
        if (floatBufferCounts[DenoiseGroup] > 0) {
            const Tables &t = floatTables[DenoiseGroup];
            stat_denoiser::fillFilterCall<float>(calls[nCalls++], floatBufferCounts[DenoiseGroup], width, height, filterDSFactor,
                                                 filterRadius, denoiseFilm, t.n, t.mean, t.m2, t.m3, t.film, filmBuffer.gpuMat,
                                                 gBufferImages, gBufferChannelCounts, gBufferDRFactors, gBuffers.size(),
                                                 t.meanCorr, t.discriminator, t.filmFiltered, filmFilteredBuffer.gpuMat, stream);
        }
        if (rgbBufferCounts[DenoiseGroup] > 0) {
            const Tables &t = rgbTables[DenoiseGroup];
            stat_denoiser::fillFilterCall<float3>(calls[nCalls++], rgbBufferCounts[DenoiseGroup], width, height, filterDSFactor,
                                                  filterRadius, denoiseFilm, t.n, t.mean, t.m2, t.m3, t.film, filmBuffer.gpuMat,
                                                  gBufferImages, gBufferChannelCounts, gBufferDRFactors, gBuffers.size(),
                                                  t.meanCorr, t.discriminator, t.filmFiltered, filmFilteredBuffer.gpuMat, stream);
        }
        if (pipe.downloading) {   // copies out of an earlier Download() still read the images this call rewrites
            check(statmc_event_record(pipe.join, pipe.down));
            check(statmc_stream_wait_event(stream.handle(), pipe.join));
        }
        const int nb = pipe.uploaded;
        pipe.uploaded = 0;
        if (nb <= 1) {
            joinUploads();
            for (int c = 0; c < nCalls; c++) calls[c].run();
            return;
        }
        // transfer k brought rows [arrivalEdge(k), arrivalEdge(k + 1)): band k and its lower halo.  Pre-pass those rows,
        // then filter band k (its upper halo came with the transfers before).
        for (int k = 0; k < nb; k++) {
            pipe.waitArrived(stream.handle(), k);
            for (int c = 0; c < nCalls; c++) calls[c].prepassRows(arrivalEdge(k, nb), arrivalEdge(k + 1, nb));
            for (int c = 0; c < nCalls; c++) calls[c].filterRows(bandEdge(k, nb), bandEdge(k + 1, nb));
            check(statmc_event_record(pipe.filtered[k], stream.handle()));
        }
        pipe.denoised = nb;
        pipe.pendingJoin = 0;   // `stream` has waited for every band
    }
    // estimator.cpp:491-569.  The reference runs this loop on the CPU between rendering and
    // Upload(); here it is the device kernel, enqueued on the stream: call it after Upload().
    void CalculateMeanVars(bool rowNQuirk = true) {
        joinUploads();
        if (floatBufferCounts[CalculateMeanVarianceGroup] > 0) {
            const Tables &t = floatTables[CalculateMeanVarianceGroup];
            stat_denoiser::calculateMeanVars<float>(floatBufferCounts[CalculateMeanVarianceGroup], width, height, t.n,
                                                    t.filmM2, t.filmVar, stream, rowNQuirk);
        }
        if (rgbBufferCounts[CalculateMeanVarianceGroup] > 0) {
            const Tables &t = rgbTables[CalculateMeanVarianceGroup];
            stat_denoiser::calculateMeanVars<float3>(rgbBufferCounts[CalculateMeanVarianceGroup], width, height, t.n,
                                                     t.filmM2, t.filmVar, stream, rowNQuirk);
        }
    }
    void Synchronize() {  // estimator.cpp:571-573
        joinUploads();
        stat_denoiser::synchronize(stream);
        if (pipe.downloading) {
            check(statmc_synchronize(pipe.down));
            pipe.downloading = false;
        }
    }
    int deviceIndex() const { return device; }

    // Tile-local pooled moments of a device image -- per tileSize x tileSize tile and channel {count, mean, M2}, by one
    // wave per tile (Welford per lane, Chan merges through wave shuffles: statmc_tile_moments).  On the untransformed
    // radiance mean (filmBuffers[type][bounce]) this is the local noise level of the estimate, tile by tile: what an
    // adaptive sampler or a progress display reads instead of the full-resolution images.  Returns a host image of
    // tiles_y rows and tiles_x * channels columns whose "pixels" are the three moments; blocks until it is there.
    HostImage TileMoments(const Buffer &b, int tileSize = 16) {
        joinUploads();
        const int C = b.gpuMat.channels();
        const int tx = (width + tileSize - 1) / tileSize, ty = (height + tileSize - 1) / tileSize;
        DeviceImage dev(ty, tx * C, F32C3);
        HostImage host(ty, tx * C, F32C3);
        check(statmc_tile_moments((uint16_t)width, (uint16_t)height, C, static_cast<const float *>(b.gpuMat.data()), tileSize,
                                  static_cast<float *>(dev.data()), stream.handle()));
        dev.download(host, stream);
        Synchronize();
        return host;
    }

    // The noise level of the estimate per render tile, for a sampler that spends its next samples where they are needed:
    // per tileSize x tileSize tile the mean over its pixels of the variance of the mean, film-m2 / ((n - 1) n)
    // (statmc_calculate_mean_vars, per pixel: no row quirk), summed over the channels in channel order -- one launch of the
    // per-pixel kernel and one of the wave-level tile reduction (statmc_tile_moments), tiles_x * tiles_y floats back.
    // Row-major over the tiles; blocks until it is there.  tools/statmc_render_sim.cpp --adaptive is the consumer.
    std::vector<float> TileNoise(unsigned char type = 0, unsigned char bounce = 0, int tileSize = 16) {
        joinUploads();
        const Buffer &m2 = filmM2Buffers.at(type).at(bounce), &n = nBuffers.at(type).at(bounce);
        const int C = m2.gpuMat.channels();
        const int tx = (width + tileSize - 1) / tileSize, ty = (height + tileSize - 1) / tileSize;
        DeviceImage var(height, width, C == 3 ? F32C3 : F32C1), dev(ty, tx * C, F32C3);
        HostImage host(ty, tx * C, F32C3);
        const statmc_image dn = n.gpuMat.desc(), dm2 = m2.gpuMat.desc(), dvar = var.desc();
        check(statmc_calculate_mean_vars(1, (uint16_t)width, (uint16_t)height, C, &dn, &dm2, &dvar, /*row_n_quirk=*/0, stream.handle()));
        check(statmc_tile_moments((uint16_t)width, (uint16_t)height, C, static_cast<const float *>(var.data()), tileSize,
                                  static_cast<float *>(dev.data()), stream.handle()));
        dev.download(host, stream);
        Synchronize();
        std::vector<float> noise((size_t)tx * ty);
        for (size_t t = 0; t < noise.size(); t++) {
            float s = 0.f;
            for (int c = 0; c < C; c++) s += host.ptr<float>()[(t * C + c) * 3 + 1];   // {count, mean, M2}: the tile mean
            noise[t] = s;
        }
        return noise;
    }

    // ---- the accumulation side: tiles in, statistics images on the device ------------------
    // The reference's constructor also takes the film's cropped pixel bounds and its pixel reconstruction filter
    // (estimator.h:253,264-265); only GetTilesF reads them, so here they are optional settings.  Defaults: the whole
    // image, a box filter of radius 0.5 with no table.
    void SetPixelFilter(const Vector2f &radius, const float *table, int tableWidth) {
        pixelFilterRadius = radius;
        filterTable = table;
        filterTableWidth = tableWidth;
    }
    void SetCroppedPixelBounds(const Bounds2i &b) { croppedPixelBounds = b; }
    // estimator.cpp:312-338: tiles over the pixels that samples inside sampleBounds contribute to under the pixel
    // filter -- Ceil(pMin - 0.5 - radius) .. Floor(pMax - 0.5 + radius) + 1, cut to the cropped pixel bounds.
    // (Not used by Render, which asks GetTiles for unfiltered tiles.)
    Bounds2i FilteredTileBounds(const Bounds2i &sampleBounds) const {
        const Bounds2i crop = croppedPixelBounds.Area() > 0 ? croppedPixelBounds : Bounds2i(Point2i(0, 0), Point2i(width, height));
        const int x0 = (int)std::ceil((float)sampleBounds.pMin.x - 0.5f - pixelFilterRadius.x);
        const int y0 = (int)std::ceil((float)sampleBounds.pMin.y - 0.5f - pixelFilterRadius.y);
        const int x1 = (int)std::floor((float)sampleBounds.pMax.x - 0.5f + pixelFilterRadius.x) + 1;
        const int y1 = (int)std::floor((float)sampleBounds.pMax.y - 0.5f + pixelFilterRadius.y) + 1;
        Bounds2i b(Point2i(std::max(x0, crop.pMin.x), std::max(y0, crop.pMin.y)), Point2i(std::min(x1, crop.pMax.x), std::min(y1, crop.pMax.y)));
        if (b.pMax.x < b.pMin.x || b.pMax.y < b.pMin.y) b = Bounds2i();   // pbrt's Intersect of disjoint boxes is degenerate too
        return b;
    }
    template <typename T>
    std::vector<StatTile<T>> GetTilesF(const Bounds2i &sampleBounds, const unsigned char bounceEnd) const {
        return std::vector<StatTile<T>>(bounceEnd, StatTile<T>(FilteredTileBounds(sampleBounds), pixelFilterRadius, filterTable, filterTableWidth));
    }
    template <typename T>
    std::vector<std::vector<StatTile<T>>> GetTilesF(const Bounds2i &sampleBounds, const unsigned char bounceEnd,
                                                    const unsigned char n) const {
        return std::vector<std::vector<StatTile<T>>>(
            bounceEnd, std::vector<StatTile<T>>(n, StatTile<T>(FilteredTileBounds(sampleBounds), pixelFilterRadius, filterTable, filterTableWidth)));
    }
    // estimator.cpp:297-309
    template <typename T>
    std::vector<StatTile<T>> GetTiles(const Bounds2i &tilePixelBounds, const unsigned char bounceEnd) const {
        return std::vector<StatTile<T>>(bounceEnd, StatTile<T>(tilePixelBounds));
    }
    template <typename T>
    std::vector<std::vector<StatTile<T>>> GetTiles(const Bounds2i &tilePixelBounds, const unsigned char bounceEnd,
                                                   const unsigned char n) const {
        return std::vector<std::vector<StatTile<T>>>(bounceEnd, std::vector<StatTile<T>>(n, StatTile<T>(tilePixelBounds)));
    }

    // Switches Merge*Tile(s) from "nothing to merge into" to the device path: the n / mean / m2 / m3 /
    // film-mean / film-m2 images of every enabled (type, bounce) live on the device, start at zero and
    // are updated by statmc_accumulate_tiles; Upload() then moves only what the host still produces
    // (the "film" image).  maxHostBytes bounds the page-locked staging the merges fill between flushes.
    // dryRun (tests of the staging logic on machines without a GPU): plain host memory, and a flush
    // only counts what it would have handed to the device (stagedMerges()).
    void EnableDeviceAccumulation(size_t maxHostBytes = (size_t)2 << 30, bool dryRun = false) {
        if (!allocateDevice && !dryRun) throw Error(STATMC_ERR_INVALID, "device accumulation needs device images");
        std::lock_guard<std::mutex> lk(acc.mu);
        acc.enabled = true;
        acc.dry = dryRun;
        acc.arenas.clear();
        // every (type, bounce) arena holds the same number of pixel-samples: the page-locked
        // staging is split in proportion to the channel counts
        size_t floatsPerPixelSample = 0;
        for (unsigned char i = 0; i < statTypeConfigs.nEnabled; i++)
            floatsPerPixelSample += (size_t)statTypeConfigs.configs[i].nChannels * statTypeConfigs.configs[i].nBounces;
        acc.capacity = (int64_t)(maxHostBytes / (sizeof(float) * std::max<size_t>(floatsPerPixelSample, 1))) / 4 * 4;
        for (unsigned char i = 0; i < statTypeConfigs.nEnabled; i++) {
            acc.arenas.emplace_back(statTypeConfigs.configs[i].nBounces);
            for (unsigned char j = 0; j < statTypeConfigs.configs[i].nBounces; j++) {
                for (auto *bufs : {&nBuffers, &meanBuffers, &m2Buffers, &m3Buffers, &filmBuffers, &filmM2Buffers}) {
                    Buffer &b = (*bufs)[i][j];
                    if (!dryRun) check(statmc_memset(b.gpuMat.data(), 0, b.gpuMat.bytes(), stream.handle()));
                    acc.deviceProduced.insert(&b);
                }
            }
        }
    }
    // Statistics of one stat type back to zero (the reference re-creates the it-radiance tiles
    // every iteration, statpath.cpp:194-207).
    void ResetStatistics(const unsigned char statTypeIndex) {
        for (auto *bufs : {&nBuffers, &meanBuffers, &m2Buffers, &m3Buffers, &filmBuffers, &filmM2Buffers})
            for (Buffer &b : (*bufs)[statTypeIndex])
                check(statmc_memset(b.gpuMat.data(), 0, b.gpuMat.bytes(), stream.handle()));
    }

    // estimator.cpp:341-407.  Thread-safe (Render calls them from its worker threads, one tile
    // each, statpath.cpp:381-388).  The samples recorded in the tile since its last merge are copied
    // to the staging arena of (statTypeIndex, bounceIndex) and the tile is emptied.
    template <typename T>
    void MergeTile(const StatTile<T> &tile, const unsigned char statTypeIndex, const unsigned char bounceIndex) const {
        mergeRecorded(tile, statTypeIndex, bounceIndex);
    }
    template <typename T>
    void MergeTiles(const std::vector<StatTile<T>> &tiles, const StatTypeConfig &cfg) const {
        for (unsigned char j = 0; j < cfg.nBounces; j++) MergeTile(tiles[j + cfg.bounceStart], cfg.index, j);
    }
    template <typename T>
    void MergeTiles(const std::vector<std::vector<StatTile<T>>> &tiles, const std::vector<StatTypeConfig> &cfgs) const {
        for (unsigned char i = 0; i < cfgs.size(); i++)
            for (unsigned char j = 0; j < cfgs[i].nBounces; j++) MergeTile(tiles[j + cfgs[i].bounceStart][i], cfgs[i].index, j);
    }
    // The transform variants differ from the plain ones only in also writing film-mean / film-m2
    // (estimator.cpp:385-386); on the device that follows from the type's `transform` flag.
    template <typename T>
    void MergeTransformTile(const StatTile<T> &tile, const unsigned char statTypeIndex, const unsigned char bounceIndex) const {
        mergeRecorded(tile, statTypeIndex, bounceIndex);
    }
    template <typename T>
    void MergeTransformTiles(const std::vector<StatTile<T>> &tiles, const StatTypeConfig &cfg) const {
        MergeTiles(tiles, cfg);
    }
    template <typename T>
    void MergeTransformTiles(const std::vector<std::vector<StatTile<T>>> &tiles, const std::vector<StatTypeConfig> &cfgs) const {
        MergeTiles(tiles, cfgs);
    }

    // Uploads what the merges have staged and runs the accumulation (asynchronous on `stream`).
    // Upload() calls it; call it yourself to bound the staging memory of a long iteration.
    void FlushSamples() const {
        std::unique_lock<std::mutex> lk(acc.mu);
        flushLocked(lk);
    }
    // (tile, buffer) merges handed over by flushes so far, and the number of flushes
    size_t stagedMerges() const { std::lock_guard<std::mutex> lk(acc.mu); return acc.flushedMerges; }
    size_t flushes() const { std::lock_guard<std::mutex> lk(acc.mu); return acc.nFlushes; }
    // Statistics images device -> host mats (dumps, OutputBufferSelection::Write); asynchronous.
    void DownloadStatistics() {
        joinUploads();
        for (unsigned char i = 0; i < statTypeConfigs.nEnabled; i++)
            for (unsigned char j = 0; j < statTypeConfigs.configs[i].nBounces; j++)
                for (auto *bufs : {&nBuffers, &meanBuffers, &m2Buffers, &m3Buffers, &filmBuffers, &filmM2Buffers})
                    (*bufs)[i][j].download(stream);
    }

    const unsigned short width, height;
    const float filterDSFactor;
    const unsigned char filterRadius;
    const bool denoiseFilm, acrrEnabled, smisEnabled;
    std::vector<unsigned char> floatBufferCounts, rgbBufferCounts;
    bool runCUDA = false;
    Stream stream;
    Buffer filmBuffer, filmFilteredBuffer;
    StatTypeConfigs statTypeConfigs;
    // insertion-ordered, duplicate-free (the reference's unordered_set<Buffer*> has no defined
    // order: SURVEY.md App. D.5)
    std::vector<Buffer *> uploadBuffers, downloadBuffers;
    std::vector<std::vector<Buffer>> nBuffers, meanBuffers, m2Buffers, m3Buffers;
    std::vector<std::vector<Buffer>> filmBuffers, filmM2Buffers, filmFilteredBuffers, filmVarBuffers;
    std::vector<std::vector<Buffer>> meanCorrBuffers, discriminatorBuffers;
    std::vector<Buffer> gBuffers;
    std::vector<float> gBufferDRFactors;

    struct Tables {
        std::vector<DeviceImage> n, mean, m2, m3, film, filmM2, filmVar, meanCorr, discriminator, filmFiltered;
    };
    Tables floatTables[nCUDAGroupIndices], rgbTables[nCUDAGroupIndices];
    std::vector<DeviceImage> gBufferImages;
    std::vector<unsigned char> gBufferChannelCounts;

  private:
    static void addUnique(std::vector<Buffer *> &v, Buffer *b) {
        if (std::find(v.begin(), v.end(), b) == v.end()) v.push_back(b);
    }
    // ---- the Upload / Denoise / Download pipeline (see Upload())
    struct Pipeline : bands::Streams {                // copy streams (non-blocking) + per-band events; the kernels run on `stream`
        std::unordered_set<const Buffer *> outputs;   // what Denoise() writes: copied back band by band
        int uploaded = 0, denoised = 0, pendingJoin = 0;
        bool downloading = false;
    } pipe;
    int bandsRequested = 0;
    int uploadQueues = [] { const char *e = std::getenv("STATMC_UPLOAD_QUEUES"); return e && std::atoi(e) >= 2 ? 2 : 1; }();
    // where the bands lie (statmc_bands.hpp: automatic = fitted to the window filter's rounds); the same plan serves Upload,
    // Denoise and Download of an iteration because it depends on the image, the radius and the request only
    // (cached: the plan is asked for at every band edge, and building it reads the environment and the device's CU count)
    // The CU count is the Estimator's own device's, read once in the constructor -- not the calling thread's current device
    // (a render worker that never called statmc_set_device would plan for the 256-CU default) -- and part of the key; the
    // cache is guarded: Upload / Denoise / Download may be driven from different host threads.
    // (by value: a reference to the cache would outlive the lock that guards it -- ADVICE r5; a plan is a few dozen ints)
    bands::Plan bandPlan() const {
        std::lock_guard<std::mutex> lk(planMutex);
        const auto key = std::make_tuple(width, height, (int)filterRadius, bandsRequested, deviceCUs);
        if (planCache.height != height || planCacheKey != key) {
            planCache = bands::plan(width, height, filterRadius, bandsRequested, deviceCUs);
            planCacheKey = key;
        }
        return planCache;
    }
    mutable bands::Plan planCache;
    mutable std::tuple<int, int, int, int, int> planCacheKey{-1, -1, -1, -1, -1};
    mutable std::mutex planMutex;
    int deviceCUs = 0;
    int bandEdge(int k, int) const { return bandPlan().edge(k); }
    int arrivalEdge(int k, int) const { return bandPlan().arrival(k); }
    void ensurePipeline(int nb) {
        if (!pipe.up) {
            for (auto *bufs : {&meanCorrBuffers, &discriminatorBuffers, &filmFilteredBuffers})
                for (auto &perType : *bufs)
                    for (Buffer &b : perType) pipe.outputs.insert(&b);
            pipe.outputs.insert(&filmFilteredBuffer);
        }
        pipe.ensure(nb);
    }
    // everything enqueued on `stream` from here on sees the uploaded images
    void joinUploads() {
        for (int k = 0; k < pipe.pendingJoin; k++) pipe.waitArrived(stream.handle(), k);
        pipe.pendingJoin = 0;
    }
    Vector2f pixelFilterRadius{0.5f, 0.5f};   // estimator.h:313 `filter` (pbrt's default box filter)
    const float *filterTable = nullptr;       // film.cpp:56-65: the film's 16 x 16 table of filter weights
    int filterTableWidth = 0;
    Bounds2i croppedPixelBounds;              // empty = the whole image
    const bool allocateDevice;
    const int device;

    // ---- staging of recorded samples -----------------------------------------------------
    // Between two flushes every merged tile owns one slot: a block of S x pixels "pixel-samples"
    // at the same offset in the arena of every (type, bounce) -- all types of a tile see the same
    // samples per pixel (statpath.cpp:355-371), so one tile table serves them all.
    struct PinnedFloats {  // page-locked, allocated once (its address must not move: merges copy into it unlocked)
        ~PinnedFloats() {
            if (ptr && pinned) statmc_free_host(ptr);
            if (ptr && !pinned) std::free(ptr);
        }
        void allocate(size_t n, bool pin) {
            if (ptr) return;
            void *p = nullptr;
            if (pin) check(statmc_malloc_host(&p, n * sizeof(float)));
            else if (!(p = std::malloc(n * sizeof(float)))) throw std::bad_alloc();
            ptr = static_cast<float *>(p);
            cap = n;
            pinned = pin;
        }
        float *ptr = nullptr;
        size_t cap = 0;
        bool pinned = true;
    };
    struct DeviceBytes {
        ~DeviceBytes() { if (ptr) statmc_free(ptr); }
        void reserve(size_t bytes) {
            if (bytes <= cap) return;
            if (ptr) statmc_free(ptr);
            ptr = nullptr;
            cap = 0;
            if (usePlacedMemory()) check(statmc_malloc_placed(&ptr, bytes, STATMC_MEM_STREAM));   // a sample arena: read once per flush
            else check(statmc_malloc(&ptr, bytes));
            cap = bytes;
        }
        void *ptr = nullptr;
        size_t cap = 0;
    };
    struct Arena {
        PinnedFloats host;
        DeviceBytes dev;
        std::vector<uint32_t> slots;     // slots this (type, bounce) has merged since the last flush
        std::vector<uint8_t> merged;     // merged[slot] != 0: the same, as a flag per slot (O(1) duplicate check)
    };
    struct Slot {
        int32_t x0, y0, x1, y1, samples;
        int64_t offset;                  // in pixel-samples, a multiple of 4
    };
    struct Accumulation {
        std::mutex mu;
        std::condition_variable idle;    // signalled when the last unlocked copy has finished
        int writers = 0;                 // merges copying into the arenas right now (outside the lock)
        bool enabled = false, uploadInFlight = false, dry = false;
        size_t flushedMerges = 0, nFlushes = 0;
        int64_t capacity = 0;            // pixel-samples per arena
        std::vector<std::vector<Arena>> arenas;  // [statTypeIndex][bounceIndex]
        std::vector<Slot> slots;
        std::map<std::array<int32_t, 4>, uint32_t> slotOf;
        int64_t nextOffset = 0;
        DeviceBytes tables;
        std::unordered_set<const Buffer *> deviceProduced;
    };
    mutable Accumulation acc;

    // host staging may be rewritten only after the copies of the previous flush have finished
    void beginEpochLocked() const {
        if (!acc.uploadInFlight) return;
        if (acc.dry) {
            acc.uploadInFlight = false;
            return;
        }
        check(statmc_set_device(device));
        check(statmc_synchronize(stream.handle()));
        acc.uploadInFlight = false;
    }

    template <typename T>
    void mergeRecorded(const StatTile<T> &tile, const unsigned char ti, const unsigned char bj) const {
        constexpr int C = stat_denoiser::detail::channels<T>::value;
        // every pixel that received samples received the same number, and those pixels form a
        // rectangle (the whole tile, or its part inside the integrator's pixelbounds, statpath.cpp:261)
        uint32_t S = 0;
        for (uint32_t c : tile.counts) S = std::max(S, c);
        if (S == 0) return;
        const int tw = tile.tileWidth, th = tw ? (int)(tile.nPixels / tw) : 0;
        int rx0 = tw, ry0 = th, rx1 = 0, ry1 = 0;
        for (int y = 0; y < th; y++)
            for (int x = 0; x < tw; x++)
                if (tile.counts[(size_t)y * tw + x]) {
                    rx0 = std::min(rx0, x); rx1 = std::max(rx1, x + 1);
                    ry0 = std::min(ry0, y); ry1 = std::max(ry1, y + 1);
                }
        for (int y = 0; y < th; y++)
            for (int x = 0; x < tw; x++) {
                const bool inside = x >= rx0 && x < rx1 && y >= ry0 && y < ry1;
                if (tile.counts[(size_t)y * tw + x] != (inside ? S : 0u))
                    throw Error(STATMC_ERR_UNSUPPORTED, "Merge*Tile: pixels of one tile hold different sample counts");
            }
        const std::array<int32_t, 4> key = {tile.pixelBounds.pMin.x + rx0, tile.pixelBounds.pMin.y + ry0,
                                            tile.pixelBounds.pMin.x + rx1, tile.pixelBounds.pMin.y + ry1};
        if (key[0] < 0 || key[1] < 0 || key[2] > width || key[3] > height)
            throw Error(STATMC_ERR_INVALID, "Merge*Tile: tile outside the image");
        const size_t npx = (size_t)(rx1 - rx0) * (ry1 - ry0);
        const int64_t need = (int64_t)((S * npx + 3) / 4 * 4);
        float *dst = nullptr;
        {   // ---- under the lock: find or make the tile's slot, claim it for this buffer
            std::unique_lock<std::mutex> lk(acc.mu);
            if (!acc.enabled)
                throw Error(STATMC_ERR_INVALID, "Merge*Tile: call EnableDeviceAccumulation() first (tiles record samples; "
                                                "the moments are computed on the device)");
            if (ti >= acc.arenas.size() || bj >= acc.arenas[ti].size()) throw Error(STATMC_ERR_INVALID, "Merge*Tile: no such buffer");
            if (statTypeConfigs.configs[ti].nChannels != C) throw Error(STATMC_ERR_INVALID, "Merge*Tile: channel count mismatch");
            if (need > acc.capacity) throw Error(STATMC_ERR_INVALID, "Merge*Tile: one tile's samples exceed the staging size");
            auto it = acc.slotOf.find(key);
            if (it == acc.slotOf.end() && acc.nextOffset + need > acc.capacity) {
                flushLocked(lk);  // staging full: hand over what is there (waits for the copies in progress)
                it = acc.slotOf.end();
            }
            beginEpochLocked();
            uint32_t slot;
            if (it == acc.slotOf.end()) {
                slot = (uint32_t)acc.slots.size();
                acc.slots.push_back(Slot{key[0], key[1], key[2], key[3], (int32_t)S, acc.nextOffset});
                acc.slotOf.emplace(key, slot);
                acc.nextOffset += need;
            } else {
                slot = it->second;
                if (acc.slots[slot].samples != (int32_t)S)
                    throw Error(STATMC_ERR_UNSUPPORTED, "Merge*Tile: the stat types of one tile hold different sample counts");
            }
            Arena &A = acc.arenas[ti][bj];
            if (A.merged.size() <= slot) A.merged.resize((size_t)slot + 1, 0);
            if (A.merged[slot]) throw Error(STATMC_ERR_INVALID, "Merge*Tile: tile merged twice");
            A.merged[slot] = 1;
            A.host.allocate((size_t)acc.capacity * C, !acc.dry);
            A.slots.push_back(slot);
            dst = A.host.ptr + (size_t)acc.slots[slot].offset * C;
            acc.writers++;
        }
        // ---- outside the lock: the copy (the arena does not move; a flush waits for writers == 0)
        const int rw = rx1 - rx0;
        for (uint32_t s = 0; s < S; s++)
            for (int y = ry0; y < ry1; y++)
                std::memcpy(dst + ((size_t)s * npx + (size_t)(y - ry0) * rw) * C, &tile.planes[s][(size_t)y * tw + rx0],
                            (size_t)rw * C * sizeof(float));
        std::fill(tile.counts.begin(), tile.counts.end(), 0u);
        {
            std::lock_guard<std::mutex> lk(acc.mu);
            if (--acc.writers == 0) acc.idle.notify_all();
        }
    }

    statmc_stat_type statTypeFor(unsigned char i, unsigned char j, const float *samples) const {
        const StatTypeConfig &cfg = statTypeConfigs.configs[i];
        statmc_stat_type t;
        std::memset(&t, 0, sizeof(t));
        t.channels = cfg.nChannels;
        t.transform = cfg.transform ? 1 : 0;
        t.max_moment = cfg.maxMoment;
        t.samples = samples;
        t.n = static_cast<int32_t *>(nBuffers[i][j].gpuMat.data());
        t.mean = static_cast<float *>(meanBuffers[i][j].gpuMat.data());
        t.m2 = static_cast<float *>(m2Buffers[i][j].gpuMat.data());
        t.m3 = static_cast<float *>(m3Buffers[i][j].gpuMat.data());
        if (cfg.transform) {  // otherwise film-mean / film-m2 are the mean / m2 images themselves
            t.film_mean = static_cast<float *>(filmBuffers[i][j].gpuMat.data());
            t.film_m2 = static_cast<float *>(filmM2Buffers[i][j].gpuMat.data());
        }
        return t;
    }

    void flushLocked(std::unique_lock<std::mutex> &lk) const {
        acc.idle.wait(lk, [&] { return acc.writers == 0; });  // merges still copying into the arenas
        if (acc.slots.empty()) return;
        acc.nFlushes++;
        for (const auto &per_type : acc.arenas)
            for (const Arena &a : per_type) acc.flushedMerges += a.slots.size();
        if (acc.dry) {
            resetEpochLocked();
            return;
        }
        check(statmc_set_device(device));  // a render worker thread may be the one that flushes
        const size_t used = (size_t)acc.nextOffset;  // pixel-samples staged in every arena
        const size_t n = acc.slots.size();
        // tile tables: bounds (int32 x 4), offsets (int64), samples (int32), one device block
        std::vector<int64_t> offsets(n);
        std::vector<int32_t> bounds(4 * n), samples(n);
        for (size_t k = 0; k < n; k++) {
            const Slot &s = acc.slots[k];
            bounds[4 * k] = s.x0; bounds[4 * k + 1] = s.y0; bounds[4 * k + 2] = s.x1; bounds[4 * k + 3] = s.y1;
            offsets[k] = s.offset;
            samples[k] = s.samples;
        }
        const size_t tabBytes = n * (8 + 16 + 4);
        acc.tables.reserve(tabBytes);
        char *tab = static_cast<char *>(acc.tables.ptr);
        // (synchronous copies of a few KB: the vectors die with this call)
        check(statmc_upload(tab, offsets.data(), n * 8, stream.handle()));
        check(statmc_upload(tab + n * 8, bounds.data(), n * 16, stream.handle()));
        check(statmc_upload(tab + n * 24, samples.data(), n * 4, stream.handle()));
        check(statmc_synchronize(stream.handle()));
        std::vector<statmc_stat_type> full;
        for (unsigned char i = 0; i < acc.arenas.size(); i++)
            for (unsigned char j = 0; j < acc.arenas[i].size(); j++) {
                Arena &A = acc.arenas[i][j];
                if (A.slots.empty()) continue;
                const size_t bytes = used * statTypeConfigs.configs[i].nChannels * sizeof(float);
                A.dev.reserve(bytes);
                check(statmc_upload(A.dev.ptr, A.host.ptr, bytes, stream.handle()));
                const statmc_stat_type t = statTypeFor(i, j, static_cast<const float *>(A.dev.ptr));
                if (A.slots.size() == n) {
                    full.push_back(t);
                } else {  // this buffer saw only some of the tiles: its own, filtered table
                    std::vector<int64_t> o;
                    std::vector<int32_t> b, sm;
                    for (uint32_t k : A.slots) {
                        o.push_back(offsets[k]);
                        sm.push_back(samples[k]);
                        b.insert(b.end(), bounds.begin() + 4 * k, bounds.begin() + 4 * k + 4);
                    }
                    DeviceBytes own;
                    const size_t m = o.size();
                    own.reserve(m * 28);
                    char *p = static_cast<char *>(own.ptr);
                    check(statmc_upload(p, o.data(), m * 8, stream.handle()));
                    check(statmc_upload(p + m * 8, b.data(), m * 16, stream.handle()));
                    check(statmc_upload(p + m * 24, sm.data(), m * 4, stream.handle()));
                    check(statmc_accumulate_tiles(width, height, &t, 1, reinterpret_cast<const int32_t *>(p + m * 8),
                                                  reinterpret_cast<const int64_t *>(p), reinterpret_cast<const int32_t *>(p + m * 24),
                                                  (int)m, stream.handle()));
                    check(statmc_synchronize(stream.handle()));  // `own` is freed on scope exit
                }
            }
        for (size_t first = 0; first < full.size(); first += 8) {  // kMaxStatTypes per launch
            const int cnt = (int)std::min<size_t>(8, full.size() - first);
            check(statmc_accumulate_tiles(width, height, full.data() + first, cnt, reinterpret_cast<const int32_t *>(tab + n * 8),
                                          reinterpret_cast<const int64_t *>(tab), reinterpret_cast<const int32_t *>(tab + n * 24),
                                          (int)n, stream.handle()));
        }
        resetEpochLocked();
    }

    void resetEpochLocked() const {
        acc.uploadInFlight = true;
        acc.slots.clear();
        acc.slotOf.clear();
        acc.nextOffset = 0;
        for (auto &per_type : acc.arenas)
            for (Arena &a : per_type) {
                a.slots.clear();
                a.merged.clear();
            }
    }
};

// ------------------------------------------------------------------------------------------
// The RGB denoise pass of an Estimator cut into a gx x gy grid of film blocks, one per device of a list (new
// capability: the reference is single-GPU).  Per block: cut the seven filter inputs out of the
// Estimator's device images (device-to-device rectangle copies), pre-pass + pack into the block + halo image, fetch the
// r-pixel halo from the neighbouring blocks (statmc_halo_exchange: two phases of peer copies ordered by events), window
// filter of the owned pixels, paste into "film-f".  Everything is enqueued asynchronously on one stream per block; the
// work tiles of the filter sit on a grid fixed in film coordinates, so the result is bit-identical to the unsharded
// Estimator::Denoise().  With every block on the device of the Estimator this is the single-GPU emulation the tests
// use; with blocks on different devices the copies cross xGMI.
class FilmShards {
  public:
    FilmShards(Estimator &est, int gx, int gy, std::vector<int> devices = {}) : est(est), gx(gx), gy(gy) {
        if (gx < 1 || gy < 1 || est.width % gx || est.height % gy)
            throw Error(STATMC_ERR_INVALID, "FilmShards: the film does not split into equal blocks");
        int nRgb = 0, nSc = 0;
        for (size_t g = 0; g < est.gBuffers.size(); g++) (est.gBufferChannelCounts[g] == 3 ? nRgb : nSc)++;
        if (est.rgbBufferCounts[DenoiseGroup] != 1 || nRgb > 2 || nSc > 2)
            throw Error(STATMC_ERR_UNSUPPORTED, "FilmShards: one RGB radiance buffer under at most two RGB and two 1-channel G-buffers");
        plainTwoRgb = nRgb == 2 && nSc == 0;
        bw = est.width / gx;
        bh = est.height / gy;
        r = est.filterRadius;
        if (devices.empty()) devices.push_back(est.deviceIndex());
        blocks.resize((size_t)gx * gy);
        for (int b = 0; b < gx * gy; b++) {
            Block &B = blocks[b];
            const int bx = b % gx, by = b / gx;
            B.device = devices[b % devices.size()];
            B.pl = bx > 0 ? r : 0; B.pr = bx + 1 < gx ? r : 0; B.pt = by > 0 ? r : 0; B.pb = by + 1 < gy ? r : 0;
            B.x0 = bx * bw; B.y0 = by * bh;
            check(statmc_setup(B.device));               // idempotent; makes the device current for the allocations
            B.n = DeviceImage(bh, bw, I32C1);
            for (DeviceImage *im : {&B.mean, &B.m2, &B.m3, &B.colour}) *im = DeviceImage(bh, bw, F32C3);
            for (size_t g = 0; g < est.gBuffers.size(); g++) {
                B.g.push_back(DeviceImage(bh, bw, est.gBufferChannelCounts[g] == 3 ? F32C3 : F32C1));
                B.gch.push_back(est.gBufferChannelCounts[g]);
            }
            B.dg.resize(B.g.size());
            const int pw = bw + B.pl + B.pr, ph = bh + B.pt + B.pb;
            B.out = DeviceImage(ph, pw, F32C3);
            void *st = nullptr;
            check(statmc_stream_create(&st));
            B.stream = std::shared_ptr<void>(st, [](void *q) { statmc_stream_destroy(q); });
        }
        check(statmc_set_device(est.deviceIndex()));
        layOutPacked();
    }

    // Estimator::Denoise() for the RGB buffer, sharded.  The inputs must be on the Estimator's device (after Upload()
    // or a flush of the device accumulation); returns with the work enqueued and the Estimator's stream waiting for it.
    void Denoise() {
        est.Synchronize();   // the cuts below read what the Estimator's stream wrote
        layOutPacked();      // (the filter spec may have changed since the last call)
        // filter spec, significance level and quantile tables are per-device state: every block device filters under the
        // Estimator's device's rules (a seam between blocks under different specs would not be the unsharded result)
        for (const Block &B : blocks)
            if (B.device != est.deviceIndex()) check(statmc_copy_device_settings(est.deviceIndex(), B.device));
        const Estimator::Tables &t = est.rgbTables[DenoiseGroup];
        const DeviceImage &colour = est.denoiseFilm ? est.filmBuffer.gpuMat : t.film[0];
        const DeviceImage &out = est.denoiseFilm ? est.filmFilteredBuffer.gpuMat : t.filmFiltered[0];
        const int src = est.deviceIndex();
        std::vector<statmc_block> desc(blocks.size());
        for (size_t b = 0; b < blocks.size(); b++) {
            Block &B = blocks[b];
            check(statmc_set_device(B.device));
            void *st = B.stream.get();
            auto cut = [&](const DeviceImage &whole, DeviceImage &blk, int elem) {
                const statmc_image d = blk.desc(), w = whole.desc();
                check(statmc_copy_rect(&d, B.device, 0, 0, &w, src, B.x0, B.y0, bw, bh, elem, st));
            };
            cut(t.n[0], B.n, 4); cut(t.mean[0], B.mean, 12); cut(t.m2[0], B.m2, 12); cut(t.m3[0], B.m3, 12);
            cut(colour, B.colour, 12);
            for (size_t g = 0; g < B.g.size(); g++) cut(est.gBufferImages[g], B.g[g], 4 * B.gch[g]);
            statmc_filter_args a = args(B);
            check(statmc_prepass_pack(&a, &B.packedDesc, B.pl, B.pt));
            desc[b] = statmc_block{B.device, B.packedDesc, st};
        }
        check(statmc_halo_exchange(desc.data(), gx, gy, bw, bh, r));
        for (Block &B : blocks) {
            check(statmc_set_device(B.device));
            statmc_filter_args a = args(B);
            a.packed_inputs = B.packedDesc;
            a.width = (uint16_t)B.packedDesc.cols;
            a.height = (uint16_t)B.packedDesc.rows;
            a.roi_x0 = B.pl; a.roi_y0 = B.pt; a.roi_x1 = B.pl + bw; a.roi_y1 = B.pt + bh;
            a.film_x0 = B.x0 - B.pl; a.film_y0 = B.y0 - B.pt;
            const statmc_image o = B.out.desc();
            a.film_filtered = &o;
            check(statmc_window_filter(&a, 3));
            const statmc_image w = out.desc();
            check(statmc_copy_rect(&w, src, B.x0, B.y0, &o, B.device, B.pl, B.pt, bw, bh, 12, B.stream.get()));
        }
        for (Block &B : blocks) {
            check(statmc_set_device(B.device));
            check(statmc_synchronize(B.stream.get()));
        }
        check(statmc_set_device(src));
    }

    int nBlocks() const { return (int)blocks.size(); }
    int packedChannels() const { return pch; }

  private:
    // The block + halo images for the filter spec of the Estimator's device: 15 channels for the shipped two RGB G-buffers,
    // 17 with depth / material id, 16 / 18 under Welch degrees of freedom (+ the sample count, which the pair test reads).
    void layOutPacked() {
        statmc_filter_spec spec;
        check(statmc_set_device(est.deviceIndex()));
        check(statmc_get_filter_spec(&spec));
        const bool welch = spec.dof == STATMC_DOF_WELCH;
        const int want = welch ? (plainTwoRgb ? 16 : 18) : plainTwoRgb ? 15 : 17;
        if (want == pch) return;
        pch = want;
        for (Block &B : blocks) {
            check(statmc_set_device(B.device));
            check(statmc_synchronize(B.stream.get()));
            const int pw = bw + B.pl + B.pr, ph = bh + B.pt + B.pb;
            void *p = nullptr;
            check(statmc_malloc(&p, (size_t)pw * ph * pch * 4));
            B.packed = std::shared_ptr<void>(p, [](void *q) { statmc_free(q); });
            B.packedDesc = statmc_image{p, (size_t)pw * pch * 4, pw, ph};
        }
        check(statmc_set_device(est.deviceIndex()));
    }
    struct Block {
        int device = 0, pl = 0, pr = 0, pt = 0, pb = 0, x0 = 0, y0 = 0;
        DeviceImage n, mean, m2, m3, colour, out;
        std::vector<DeviceImage> g;          // the Estimator's G-buffers, in its order
        std::shared_ptr<void> packed, stream;
        statmc_image packedDesc{};
        // descriptor storage the argument block points into
        statmc_image dn{}, dmean{}, dm2{}, dm3{}, dcol{};
        std::vector<statmc_image> dg;
        std::vector<uint8_t> gch;
    };
    // the argument block of filter<float3> for one block (reference order, estimator.cpp:465-487)
    statmc_filter_args args(Block &B) {
        statmc_filter_args a;
        std::memset(&a, 0, sizeof(a));
        B.dn = B.n.desc(); B.dmean = B.mean.desc(); B.dm2 = B.m2.desc(); B.dm3 = B.m3.desc(); B.dcol = B.colour.desc();
        for (size_t g = 0; g < B.g.size(); g++) B.dg[g] = B.g[g].desc();
        a.n_buffers = 1;
        a.width = (uint16_t)bw;
        a.height = (uint16_t)bh;
        a.filter_ds_factor = est.filterDSFactor;
        a.filter_radius = est.filterRadius;
        a.denoise_film = 0;
        a.n = &B.dn; a.mean = &B.dmean; a.m2 = &B.dm2; a.m3 = &B.dm3; a.film = &B.dcol;
        a.g_buffers = B.dg.data();
        a.g_channel_counts = B.gch.data();
        a.g_dr_factors = est.gBufferDRFactors.data();
        a.n_g_buffers = B.g.size();
        a.stream = B.stream.get();
        return a;
    }
    Estimator &est;
    int gx, gy, bw = 0, bh = 0, r = 0, pch = 0;
    bool plainTwoRgb = true;
    std::vector<Block> blocks;
};

}  // namespace statmc

#endif  // STATMC_DENOISER_HPP
