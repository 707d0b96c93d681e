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
//   statmc::Buffer, BufferRegistry       <- src/statistics/buffer.h:19-80
//   statmc::StatTypeConfig(s), enums     <- src/statistics/estimator.h:61-102, statpath.h:20-36
//   statmc::makeStatTypeConfigs          <- CreateStatPathIntegrator's rules, statpath.cpp:1013-1173
//   statmc::Estimator                    <- src/statistics/estimator.h:241-380, estimator.cpp:86-289,
//                                           409-489, 571-573 (same member names, same buffer names,
//                                           same aliasing and upload/download sets)
//
// Error behaviour: the reference has none at these call sites (OpenCV throws cv::Exception);
// here every failing C call throws statmc::Error carrying statmc_last_error().
#ifndef STATMC_DENOISER_HPP
#define STATMC_DENOISER_HPP

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "statmc.h"

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
    HostImage(int rows, int cols, ImageType type)
        : rows(rows), cols(cols), type(type),
          store(std::make_shared<std::vector<uint32_t>>((size_t)rows * cols * channelsOf(type), 0u)) {}
    int channels() const { return channelsOf(type); }
    size_t bytes() const { return (size_t)rows * cols * channels() * 4; }
    bool empty() const { return !store; }
    void *ptr() const { return store ? store->data() : nullptr; }
    template <class T>
    T *ptr() const { return reinterpret_cast<T *>(ptr()); }
    bool sameStorage(const HostImage &o) const { return store == o.store; }
    int rows = 0, cols = 0;
    ImageType type = F32C1;

  private:
    std::shared_ptr<std::vector<uint32_t>> store;
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
class DeviceImage {
  public:
    DeviceImage() = default;
    DeviceImage(int rows, int cols, ImageType type, bool allocate = true) : rows(rows), cols(cols), type(type) {
        if (allocate) {
            void *p = nullptr;
            check(statmc_malloc(&p, bytes()));
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

// Argument order of cv::cuda::stat_denoiser::filter<T> (estimator.cpp:437-459).  Where the
// reference passes a GpuMat holding a device array of PtrStepSzb, this takes the images.
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
    const auto n = detail::descs(nPtrs), mean = detail::descs(meanPtrs), m2 = detail::descs(m2Ptrs),
               m3 = detail::descs(m3Ptrs), film = detail::descs(filmPtrs), g = detail::descs(gBufferPtrs),
               mc = detail::descs(meanCorrPtrs), dc = detail::descs(discriminatorPtrs),
               ff = detail::descs(filmFilteredPtrs);
    statmc_filter_args a;
    std::memset(&a, 0, sizeof(a));
    a.n_buffers = nBuffers;
    a.width = width;
    a.height = height;
    a.filter_ds_factor = filterDSFactor;
    a.filter_radius = filterRadius;
    a.denoise_film = denoiseFilm ? 1 : 0;
    a.n = n.data();
    a.mean = mean.data();
    a.m2 = m2.data();
    a.m3 = m3.data();
    a.film = film.data();
    a.film_buffer = filmBuffer.desc();
    a.g_buffers = g.data();
    a.g_channel_counts = gBufferChannelCounts.data();
    a.g_dr_factors = gBufferDRFactors.data();
    a.n_g_buffers = nGBuffers;
    a.mean_corr = mc.data();
    a.discriminator = dc.data();
    a.film_filtered = ff.data();
    a.film_filtered_buffer = filmFilteredBuffer.desc();
    a.stream = stream.handle();
    check(detail::channels<T>::value == 3 ? statmc_filter_f32x3(&a) : statmc_filter_f32(&a));
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
          allocateDevice(allocateDevice) {
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
        if (allocateDevice) stat_denoiser::setup(device);
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

    void Upload() {  // estimator.cpp:409-416
        for (Buffer *b : uploadBuffers) b->upload(stream);
    }
    void Download() {  // estimator.cpp:418-425
        for (Buffer *b : downloadBuffers) b->download(stream);
    }
    void Denoise() {  // estimator.cpp:427-489
        if (floatBufferCounts[DenoiseGroup] > 0) {
            const Tables &t = floatTables[DenoiseGroup];
            stat_denoiser::filter<float>(floatBufferCounts[DenoiseGroup], width, height, filterDSFactor, filterRadius,
                                         denoiseFilm, t.n, t.mean, t.m2, t.m3, t.film, filmBuffer.gpuMat, gBufferImages,
                                         gBufferChannelCounts, gBufferDRFactors, gBuffers.size(), t.meanCorr,
                                         t.discriminator, t.filmFiltered, filmFilteredBuffer.gpuMat, stream);
        }
        if (rgbBufferCounts[DenoiseGroup] > 0) {
            const Tables &t = rgbTables[DenoiseGroup];
            stat_denoiser::filter<float3>(rgbBufferCounts[DenoiseGroup], width, height, filterDSFactor, filterRadius,
                                          denoiseFilm, t.n, t.mean, t.m2, t.m3, t.film, filmBuffer.gpuMat, gBufferImages,
                                          gBufferChannelCounts, gBufferDRFactors, gBuffers.size(), t.meanCorr,
                                          t.discriminator, t.filmFiltered, filmFilteredBuffer.gpuMat, stream);
        }
    }
    // estimator.cpp:491-569.  The reference runs this loop on the CPU between rendering and
    // Upload(); here it is the device kernel, enqueued on the stream: call it after Upload().
    void CalculateMeanVars(bool rowNQuirk = true) {
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
    void Synchronize() { stat_denoiser::synchronize(stream); }  // estimator.cpp:571-573

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
    const bool allocateDevice;
};

}  // namespace statmc

#endif  // STATMC_DENOISER_HPP
