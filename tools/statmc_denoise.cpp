// statmc_denoise -- offline denoising of a statistics dump, the counterpart of
// `pbrt --denoise --writeimages scene.pbrt` (StatPathIntegrator::Denoise<T>,
// src/statistics/statpath.cpp:456-550) on top of statmc::Estimator and libstatmc_hip.so.
//
//   statmc_denoise --stem out/scene --spp 4,8,16 [--filtersd 10] [--filterradius 20]
//                  [--filterbuffers albedo,normal --filterbuffersds 0.02,0.1]
//                  [--output 'film-f,t0-b0-mean-corr'] [--warmup]
//                  [--significance 0|1|2] [--tquantiles table.txt] [--compare other/stem] [--no-write]
//                  [--spec gate=sym|asym,channels=and|joint,sides=two|one,dof=pixel|welch,border=clip|clamp,small_n=accept|exclude]
//                  [--grid GXxGY [--devices 0,1,..]]   the denoise pass over film blocks with a halo exchange (C++ only)
//                  [--placed]                          device images from statmc_malloc_placed (statmc::usePlacedMemory())
//   statmc_denoise --catalogue [--config denoise|acrr|smis|proden|ours] [--width W --height H]
//
// Per iteration it reads "<stem>-<spp>-film.pfm" and every "<stem>-<spp>-t<i>-b<j>-<suffix>.pfm"
// whose suffix is one of n, mean, m2, m3, film-m2, mean-corr, discriminator, film-mean
// (statpath.cpp:476-511), runs Upload / Denoise / Download / Synchronize (timed like the
// reference's "CUDA time [ns]" bracket, statpath.cpp:520-527) and writes the selected buffers as
// "<stem>-<spp>-<buffer>.pfm".  --catalogue prints the buffer catalogue of a shipped
// configuration without touching a device (used by the CPU tests).
// --compare S: after each iteration every written buffer is compared with "S-<spp>-<buffer>.pfm"
// when that file exists (e.g. film-f / mean-corr / discriminator dumps written by the CUDA
// StatMC) and the relative L2 error per channel is printed -- the check BASELINE's 1e-5 parity
// bound is stated in.  --tquantiles loads whitespace-separated t quantiles for dof 1..n into the
// selected significance slot (for users who have the reference's own tables).
#include <chrono>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "statmc_denoiser.hpp"
#include "statmc_pfm.hpp"

using namespace statmc;

#include "statmc_debug.h"   // --kernel general: the A/B switch of the library (per device; the split is a declared setting of statmc.h)

static std::vector<std::string> split(const std::string &s) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, ','))
        if (!item.empty()) out.push_back(item);
    return out;
}

static bool fileExists(const std::string &p) {
    FILE *f = std::fopen(p.c_str(), "rb");
    if (f) std::fclose(f);
    return f != nullptr;
}

// ReadFile (statpath.cpp:449-454): imread -> convertTo(buffer type) -> BGR2RGB
static void readInto(const std::string &path, Buffer &b) {
    PfmImage im = readPfm(path);
    if (im.width != b.mat.cols || im.height != b.mat.rows || im.channels != b.mat.channels())
        throw std::runtime_error(path + ": image shape does not match buffer " + b.name);
    if (b.mat.type == I32C1) {
        int32_t *dst = b.mat.ptr<int32_t>();
        for (size_t i = 0; i < im.data.size(); i++) dst[i] = (int32_t)im.data[i];  // n is stored as float
    } else {
        std::memcpy(b.mat.ptr(), im.data.data(), im.data.size() * sizeof(float));
    }
}

// relative L2 error and largest absolute difference per channel of `got` against `want`
static void compareImages(const std::string &name, const PfmImage &want, int width, int height, int channels,
                          const float *got) {
    if (want.width != width || want.height != height || want.channels != channels) {
        std::printf("compare %s: shape mismatch (%dx%dx%d vs %dx%dx%d)\n", name.c_str(), want.width, want.height,
                    want.channels, width, height, channels);
        return;
    }
    for (int c = 0; c < channels; c++) {
        double num = 0, den = 0, worst = 0;
        for (size_t i = 0; i < (size_t)width * height; i++) {
            const double a = got[i * channels + c], b = want.data[i * channels + c];
            if (std::isnan(a) && std::isnan(b)) continue;
            num += (a - b) * (a - b);
            den += b * b;
            worst = std::max(worst, std::fabs(a - b));
        }
        std::printf("compare %s ch%d rel_l2 %.6e max_abs %.6e\n", name.c_str(), c, den > 0 ? std::sqrt(num / den) : std::sqrt(num),
                    worst);
    }
}

static StatPathParams shippedConfig(const std::string &name) {
    StatPathParams p;
    p.maxDepth = 65;
    p.trackedBounces = 0;
    if (name == "denoise") {  // scenes/render-denoise.pbrt, scenes/denoise.pbrt
        p.denoiseImage = true;
    } else if (name == "acrr") {  // scenes/acrr.pbrt
        p.acrr = true;
        p.trackedBounces = 5;
        p.multiChannelStats = false;
    } else if (name == "smis") {  // scenes/smis.pbrt
        p.smis = true;
        p.trackedBounces = 6;
        p.multiChannelStats = false;
    } else if (name == "proden") {  // scenes/render-for-proden.pbrt
        p.calcProDenStats = true;
    } else if (name == "ours") {  // scenes/render-for-ours.pbrt
        p.calcStats = true;
    } else {
        throw std::runtime_error("unknown config " + name);
    }
    return p;
}

int main(int argc, char **argv) {
    try {
        std::string stem, sppList, output = "film-f", config = "denoise", compareStem, tqFile, specText, gridText, devicesText;
        bool noWrite = false;
        int forceParts = 0, bands = 0;
        std::string kernel, sweep, sweepSignificance = "0,1,2";
        int significance = 0;
        StatPathParams params = shippedConfig("denoise");
        bool catalogue = false, warmup = false, configGiven = false;
        int width = 64, height = 48;
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto next = [&]() -> std::string {
                if (i + 1 >= argc) throw std::runtime_error("missing value after " + a);
                return argv[++i];
            };
            if (a == "--stem") stem = next();
            else if (a == "--spp") sppList = next();
            else if (a == "--output") output = next();
            else if (a == "--filtersd") params.filterSD = std::stof(next());
            else if (a == "--filterradius") params.filterRadius = (unsigned char)std::stoi(next());
            else if (a == "--filterbuffers") params.filterBuffers = split(next());
            else if (a == "--filterbuffersds") {
                params.filterBufferSDs.clear();
                for (const auto &s : split(next())) params.filterBufferSDs.push_back(std::stof(s));
            } else if (a == "--config") { config = next(); configGiven = true; }
            else if (a == "--catalogue") catalogue = true;
            else if (a == "--warmup") warmup = true;
            else if (a == "--compare") compareStem = next();
            else if (a == "--no-write") noWrite = true;   // compare / time only: no output dumps
            else if (a == "--significance") significance = std::stoi(next());
            else if (a == "--tquantiles") tqFile = next();
            else if (a == "--spec") specText = next();
            else if (a == "--sweep") sweep = next();            // "all" | "quick": every filter spec x significance level in ONE process (tools/fit_spec.py)
            else if (a == "--sweep-significance") sweepSignificance = next();
            else if (a == "--grid") gridText = next();          // GXxGY: the denoise pass sharded over film blocks
            else if (a == "--devices") devicesText = next();    // devices the blocks go to, round robin (default: one)
            else if (a == "--placed") statmc::usePlacedMemory() = true;   // device images from statmc_malloc_placed (the blocks of --grid too)
            else if (a == "--parts") forceParts = std::stoi(next());
            else if (a == "--kernel") kernel = next();   // "general": window_filter_generic for every call (one arithmetic for every spec)
            else if (a == "--bands") bands = std::stoi(next());  // Upload / Denoise / Download as a pipeline of row bands (0 = automatic, 1 = off)
            else if (a == "--width") width = std::stoi(next());
            else if (a == "--height") height = std::stoi(next());
            else throw std::runtime_error("unknown option " + a);
        }
        if (configGiven) {
            StatPathParams c = shippedConfig(config);
            c.filterSD = params.filterSD;
            c.filterRadius = params.filterRadius;
            c.filterBuffers = params.filterBuffers;
            c.filterBufferSDs = params.filterBufferSDs;
            params = c;
        }
        const StatTypeConfigs cfgs = makeStatTypeConfigs(params);

        if (catalogue) {
            Buffer film("film", HostImage(height, width, F32C3), false);
            BufferRegistry reg(film);
            Estimator est(film, cfgs, params.filterSD, params.filterRadius, params.denoiseImage, params.acrr,
                          params.smis, reg, /*allocateDevice=*/false);
            est.AllocateBuffers(reg);
            for (const auto &b : reg.buffers)
                std::printf("buffer %s %s %d\n", b.name.c_str(), b.mat.type == I32C1 ? "i32" : "f32", b.mat.channels());
            for (const Buffer *b : est.uploadBuffers) std::printf("upload %s\n", b->name.c_str());
            for (const Buffer *b : est.downloadBuffers) std::printf("download %s\n", b->name.c_str());
            for (size_t i = 0; i < est.gBuffers.size(); i++)
                std::printf("gbuffer %s %d %.9g\n", est.gBuffers[i].name.c_str(), est.gBuffers[i].mat.channels(),
                            est.gBufferDRFactors[i]);
            std::printf("counts float %d %d rgb %d %d runCUDA %d ds %.9g radius %d\n", est.floatBufferCounts[0],
                        est.floatBufferCounts[1], est.rgbBufferCounts[0], est.rgbBufferCounts[1], (int)est.runCUDA,
                        est.filterDSFactor, (int)est.filterRadius);
            // aliasing facts the reference relies on
            for (unsigned char i = 0; i < est.statTypeConfigs.nEnabled; i++)
                std::printf("alias t%d mean==film-mean %d m2==film-m2 %d\n", i,
                            (int)est.meanBuffers[i][0].mat.sameStorage(est.filmBuffers[i][0].mat),
                            (int)est.m2Buffers[i][0].mat.sameStorage(est.filmM2Buffers[i][0].mat));
            if (!est.filmFilteredBuffers.empty() && !est.filmFilteredBuffers[0].empty())
                std::printf("alias t0-b0-film-mean-f==film-f %d\n",
                            (int)est.filmFilteredBuffers[0][0].mat.sameStorage(est.filmFilteredBuffer.mat));
            return 0;
        }

        if (stem.empty() || sppList.empty()) throw std::runtime_error("--stem and --spp are required");
        const std::vector<std::string> spps = split(sppList);
        const std::string first = stem + "-" + spps[0] + "-film.pfm";
        {
            PfmImage im = readPfm(first);
            width = im.width;
            height = im.height;
        }
        Buffer film("film", HostImage(height, width, F32C3));
        BufferRegistry reg(film);
        Estimator est(film, cfgs, params.filterSD, params.filterRadius, params.denoiseImage, params.acrr, params.smis, reg);
        est.AllocateBuffers(reg);
        est.SetPipelineBands(bands);
        std::cout << "pipeline bands: " << est.PipelineBands() << std::endl;
        std::vector<float> tq;
        if (!tqFile.empty()) {
            std::ifstream in(tqFile);
            if (!in) throw std::runtime_error("cannot open " + tqFile);
            for (float v; in >> v;) tq.push_back(v);
            if (tq.empty()) throw std::runtime_error(tqFile + ": no quantiles");
        }
        // one (significance level, spec) per run; --sweep: the whole grid of tools/fit_spec.py in this process
        std::vector<std::pair<int, std::string>> runs;
        if (sweep.empty()) {
            runs.push_back({significance, specText});
        } else {
            if (sweep != "all" && sweep != "quick") throw std::runtime_error("--sweep all | quick");
            const char *names[6] = {"gate", "channels", "sides", "dof", "border", "small_n"};
            const char *vals[6][2] = {{"sym", "asym"}, {"and", "joint"}, {"two", "one"}, {"pixel", "welch"}, {"clip", "clamp"}, {"accept", "exclude"}};
            const char *gates[3] = {"sym", "asym", "centre"};
            const int nFree = sweep == "quick" ? 3 : 6;
            for (const auto &sg : split(sweepSignificance))
                for (int g = 0; g < 3; g++)          // first field slowest, like itertools.product
                    for (int m = 0; m < (1 << (nFree - 1)); m++) {
                        std::string t = std::string("gate=") + gates[g];
                        for (int f = 1; f < nFree; f++) t += std::string(",") + names[f] + "=" + vals[f][(m >> (nFree - 1 - f)) & 1];
                        runs.push_back({std::stoi(sg), t});
                    }
        }
        auto applyRun = [&](int sig, const std::string &spec) {
            const statmc_filter_spec sp = stat_denoiser::parseFilterSpec(spec);
            // the table the pre-pass will read: one-sided tables sit behind the two-sided ones
            if (!tq.empty()) stat_denoiser::setTQuantiles(sig + (sp.sides ? 3 : 0), tq);
            stat_denoiser::setSignificance(sig);
            stat_denoiser::setFilterSpec(sp);
        };
        applyRun(runs[0].first, runs[0].second);
        const std::vector<std::string> outputs = split(output);
        if (forceParts > 0 && statmc_set_filter_split(forceParts) != STATMC_OK) throw std::runtime_error(statmc_last_error());
        if (kernel == "general") statmc_debug_force_filter_variant(1);
        else if (!kernel.empty()) throw std::runtime_error("--kernel general (or nothing)");
        std::unique_ptr<FilmShards> shards;
        if (!gridText.empty()) {
            const size_t x = gridText.find('x');
            if (x == std::string::npos) throw std::runtime_error("--grid wants GXxGY");
            std::vector<int> devs;
            for (const auto &d : split(devicesText)) devs.push_back(std::stoi(d));
            shards.reset(new FilmShards(est, std::stoi(gridText.substr(0, x)), std::stoi(gridText.substr(x + 1)), devs));
        }

        auto iteration = [&](const std::string &spp, bool write) {
            using clk = std::chrono::steady_clock;
            auto t0 = clk::now();
            const std::string prefix = stem + "-" + spp + "-";
            if (fileExists(prefix + "film.pfm")) readInto(prefix + "film.pfm", est.filmBuffer);
            struct Slot { const char *suffix; std::vector<std::vector<Buffer>> *bufs; };
            const Slot slots[] = {{"n", &est.nBuffers}, {"mean", &est.meanBuffers}, {"m2", &est.m2Buffers},
                                  {"m3", &est.m3Buffers}, {"film-m2", &est.filmM2Buffers},
                                  {"mean-corr", &est.meanCorrBuffers}, {"discriminator", &est.discriminatorBuffers},
                                  {"film-mean", &est.filmBuffers}};
            for (const Slot &s : slots)
                for (auto &perType : *s.bufs)
                    for (Buffer &b : perType) {
                        const std::string path = prefix + b.name + ".pfm";
                        if (fileExists(path)) readInto(path, b);
                    }
            auto t1 = clk::now();
            std::cout << "I/O time [ns]: " << std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() << std::endl;
            t0 = clk::now();
            est.Upload();
            const auto tu = clk::now();
            if (shards) shards->Denoise();   // film blocks + halo exchange (statmc_halo_exchange), same result bit for bit
            else est.Denoise();
            const auto td = clk::now();
            est.Download();
            const auto tl = clk::now();
            est.Synchronize();
            t1 = clk::now();
            std::cout << "HIP time [ns]: " << std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() << std::endl;
            // how long the host spent ENQUEUEING each phase (all four calls only enqueue; a long time here is the runtime
            // blocking the caller) and waiting in Synchronize
            auto ns = [](clk::time_point a, clk::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count(); };
            std::cout << "host phases [ns]: upload " << ns(t0, tu) << " denoise " << ns(tu, td) << " download " << ns(td, tl) << " synchronize "
                      << ns(tl, t1) << std::endl;
            if (!write) return;
            for (const auto &name : outputs) {
                const Buffer *b = reg.find(name);
                if (!b) throw std::runtime_error("no buffer named " + name);
                HostImage host = b->mat;
                if (std::find(est.downloadBuffers.begin(), est.downloadBuffers.end(), b) == est.downloadBuffers.end() &&
                    (name.find("mean-corr") != std::string::npos || name.find("discriminator") != std::string::npos)) {
                    // device-only outputs: fetch them explicitly
                    b->gpuMat.download(host, est.stream);
                    est.Synchronize();
                }
                std::vector<float> tmp;
                const float *pixels = host.type == I32C1 ? nullptr : host.ptr<float>();
                if (host.type == I32C1) {
                    tmp.resize((size_t)width * height);
                    for (size_t i = 0; i < tmp.size(); i++) tmp[i] = (float)host.ptr<int32_t>()[i];
                    pixels = tmp.data();
                }
                // compare first, and never write over the file a comparison reads (--compare with the dumps' own stem: the
                // CUDA build's outputs sit next to its inputs, tools/pin_from_dumps.sh)
                const std::string mine = prefix + name + ".pfm", other = compareStem + "-" + spp + "-" + name + ".pfm";
                const bool comparing = !compareStem.empty() && fileExists(other);
                if (comparing) compareImages(name, readPfm(other), width, height, host.channels(), pixels);
                if (!noWrite && !(comparing && other == mine)) writePfm(mine, width, height, host.channels(), pixels);
            }
        };
        for (const auto &run : runs) {
            if (!sweep.empty()) {
                std::cout << "==== sweep significance " << run.first << " spec " << run.second << std::endl;
                applyRun(run.first, run.second);
            }
            if (warmup) {  // --warmup (statpath.cpp:543-547)
                std::cout << "==== Warm-Up Start ====" << std::endl;
                iteration(spps[0], false);
                std::cout << "==== Warm-Up End ====" << std::endl;
            }
            for (size_t i = 0; i < spps.size(); i++) {
                std::cout << "Iteration: " << (i + 1) << std::endl;
                iteration(spps[i], true);
            }
        }
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "statmc_denoise: %s\n", e.what());
        return 1;
    }
}
