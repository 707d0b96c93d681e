// statmc_render_sim -- the render loop of StatPathIntegrator::Render<T> (src/statistics/statpath.cpp:
// 120-445) around statmc::Estimator, with the path tracer replaced by a counter-based synthetic
// sample source.  It exercises the accumulation side of the drop-in exactly the way the reference
// drives it:
//
//   * 16 x 16 tiles (statpath.cpp:132), one set of StatTile<T> per tile from Estimator::GetTiles
//     (statpath.cpp:173-190), worker threads taking tiles (ParallelFor2D);
//   * per pixel and sample the Add*Sample* method selected from the stat type's configuration is
//     called through a member-function pointer (GetAddSampleFn, statpath.cpp:97-116, 355-371);
//   * Merge[Transform]Tiles once per tile and iteration from the worker thread (statpath.cpp:381-388);
//   * the exponential iteration schedule spp, spp, 2 spp, 4 spp, ... (statpath.cpp:272-279);
//   * Upload / Denoise / Download / Synchronize on the main thread, timed and printed like the
//     reference's "CUDA time [ns]" bracket (statpath.cpp:397-417);
//   * dumps named <stem>-<total spp>-<buffer>.pfm (OutputBufferSelection::Write, statpath.cpp:421-427).
//
// The tiles record samples; the moments are accumulated on the device (statmc_accumulate_tiles).
// The sample source is a pure function of (seed, x, y, sample index), restated in
// tests/test_render_sim_gpu.py so that the dumps can be checked against the CPU oracle.
//
//   statmc_render_sim --width 96 --height 56 --spp 4 --iterations 3 --stem out/sim [--threads 4]
//                     [--seed 1] [--filtersd 10] [--filterradius 20] [--stage-mb 2048] [--no-denoise]
//                     [--config denoise|acrr|smis] [--trackedbounces 5] [--outputregex '.*'] [--warmup] [--tilestats]
//                     [--adaptive] [--placed]
// --adaptive: per-tile sample budgets from the tile-local noise level (Estimator::TileNoise); see RenderLoop.
// --config denoise: Render<Vec3>, RGB radiance + normal + albedo, filter<float3> (scenes/render-denoise.pbrt).
// --config acrr:    Render<Float>, "multichannelstats" false: the luminance of the path prefix up to each
//                   of the tracked bounces is one float stat buffer, filtered together by filter<float>
//                   (scenes/acrr.pbrt; estimator.cpp:434-460).
// --config smis:    no radiance statistics; per tracked bounce two float tallies (BSDF / light win rate,
//                   plain M3) through the nested MergeTiles overload, filter<float> over 2 x bounces buffers.
//   statmc_render_sim [--seed 1] --print-sample x y s     (prints one generated sample, no device)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>

#include "statmc_denoiser.hpp"
#include "statmc_pfm.hpp"

using namespace statmc;

// ---- synthetic sample source (all arithmetic in fp32, one rounding per operation) --------------
static inline uint32_t mix(uint32_t a) {  // lowbias32
    a ^= a >> 16;
    a *= 0x7feb352dU;
    a ^= a >> 15;
    a *= 0x846ca68bU;
    a ^= a >> 16;
    return a;
}
static inline uint32_t draw(uint32_t seed, int x, int y, uint32_t sample, uint32_t stream) {
    return mix(mix(mix(seed ^ (0x9e3779b9U * (uint32_t)(x + 65536 * y))) + sample) + stream * 0x85ebca6bU);
}
static inline float unit(uint32_t key) { return (float)(key >> 8) * (1.0f / 16777216.0f); }

static const float kAlbedo[3][3] = {{0.80f, 0.25f, 0.20f}, {0.15f, 0.55f, 0.85f}, {0.60f, 0.60f, 0.10f}};
static const float kNormal[3][3] = {{0.0f, 0.0f, 1.0f}, {0.6f, -0.8f, 0.0f}, {-0.7071f, 0.0f, 0.7071f}};

struct PixelSample {
    Vec3 radiance, normal, albedo;
};
// GetStatSample<T> (statpath.h): the RGB triple, or its luminance (RGBSpectrum::y(), spectrum.h)
template <typename T>
static T GetStatSample(const Vec3 &L);
template <>
Vec3 GetStatSample<Vec3>(const Vec3 &L) { return L; }
template <>
float GetStatSample<float>(const Vec3 &L) { return 0.212671f * L.x + 0.715160f * L.y + 0.072169f * L.z; }
static PixelSample makeSample(uint32_t seed, int x, int y, uint32_t s) {
    const int region = ((x / 24) + (y / 20)) % 3;
    const float e = 0.5f + 0.5f * ((float)((x * 7 + y * 3) % 32) / 32.0f);
    const bool black = (draw(seed, x, y, s, 3) & 7u) == 0u;  // zero-radiance paths
    float rad[3], nrm[3], alb[3];
    for (int c = 0; c < 3; c++) {
        const float u = unit(draw(seed, x, y, s, (uint32_t)c));
        const float t = 4.0f * (u * u);
        rad[c] = black ? 0.0f : (kAlbedo[region][c] * e) * t;
        nrm[c] = kNormal[region][c] + (unit(draw(seed, x, y, s, 4u + c)) - 0.5f) * 0.02f;
        alb[c] = kAlbedo[region][c] + (unit(draw(seed, x, y, s, 7u + c)) - 0.5f) * 0.01f;
    }
    return PixelSample{Vec3{rad[0], rad[1], rad[2]}, Vec3{nrm[0], nrm[1], nrm[2]}, Vec3{alb[0], alb[1], alb[2]}};
}

// GetAddSampleFn (statpath.cpp:97-116)
template <typename T>
using AddSampleFn = void (StatTile<T>::*)(const Point2i, const T);
template <typename T>
static AddSampleFn<T> GetAddSampleFn(const StatTypeConfig &cfg) {
    if (cfg.transform) {
        if (cfg.maxMoment == 3) return &StatTile<T>::AddTransformSampleM3;
        if (cfg.maxMoment == 2) return &StatTile<T>::AddTransformSampleM2;
        if (cfg.maxMoment == 1) return &StatTile<T>::AddTransformSampleM1;
    } else {
        if (cfg.maxMoment == 3) return &StatTile<T>::AddSampleM3;
        if (cfg.maxMoment == 2) return &StatTile<T>::AddSampleM2;
        if (cfg.maxMoment == 1) return &StatTile<T>::AddSampleM1;
    }
    return nullptr;
}

struct Options {
    int width = 96, height = 56, spp = 4, iterations = 3, threads = 4, stageMb = 2048, trackedBounces = 5;
    unsigned seed = 1;
    float filterSD = 10.f;
    int filterRadius = 20;
    bool denoise = true, acrr = false, smis = false, warmUp = false, tileStats = false, adaptive = false;
    std::string stem, outputRegex = ".*";
};

template <typename T>
static void Render(const Options &o) {
    const int width = o.width, height = o.height;
    StatPathParams params;
    params.acrr = o.acrr;
    params.smis = o.smis;
    params.multiChannelStats = !o.acrr;
    params.trackedBounces = o.trackedBounces;
    params.denoiseImage = o.denoise && !o.acrr && !o.smis;
    params.calcStats = !o.denoise && !o.acrr && !o.smis;
    params.filterSD = o.filterSD;
    params.filterRadius = (unsigned char)o.filterRadius;
    const StatTypeConfigs sCfgs = makeStatTypeConfigs(params);
    // `film` itself is not denoised here (no Film in this harness): the colour images are the
    // t0-b<j>-film-mean buffers, the results t0-b<j>-film-mean-f.
    Buffer film("film", HostImage(height, width, F32C3));
    BufferRegistry reg(film);
    Estimator estimator(film, sCfgs, o.filterSD, (unsigned char)o.filterRadius, /*denoiseFilm=*/false, o.acrr, o.smis, reg);
    estimator.AllocateBuffers(reg);
    estimator.EnableDeviceAccumulation((size_t)o.stageMb << 20);

    std::vector<StatTypeConfig> enabledRGBFeatureCfgs;  // statpath.cpp:160-163
    for (unsigned char t : {(unsigned char)StatMaterialID, (unsigned char)StatDepth, (unsigned char)StatNormal,
                            (unsigned char)StatAlbedo})
        if (sCfgs[t].enable && sCfgs[t].nChannels == 3) enabledRGBFeatureCfgs.push_back(sCfgs[t]);
    const unsigned char nRGBBuffers = (unsigned char)enabledRGBFeatureCfgs.size();
    if (nRGBBuffers != 2) throw std::runtime_error("expected the normal and albedo feature types");

    AddSampleFn<T> AddLSampleFn = GetAddSampleFn<T>(sCfgs[Radiance]);
    AddSampleFn<float> AddMISWinRateSampleFn = GetAddSampleFn<float>(sCfgs[MISBSDFWinRate]);
    AddSampleFn<Vec3> AddRGBGBufferSampleFn = GetAddSampleFn<Vec3>(sCfgs[StatNormal]);
    const bool misEnabled = sCfgs[MISBSDFWinRate].enable && sCfgs[MISLightWinRate].enable;
    const std::vector<StatTypeConfig> misCfgs = {sCfgs[MISBSDFWinRate], sCfgs[MISLightWinRate]};
    const unsigned char nLs = sCfgs[Radiance].bounceEnd;

    const int tileSize = 16;  // statpath.cpp:132
    const int nTilesX = (width + tileSize - 1) / tileSize, nTilesY = (height + tileSize - 1) / tileSize;
    const int nTilesTotal = nTilesX * nTilesY;
    std::vector<std::vector<StatTile<T>>> lTiles(nTilesTotal);
    std::vector<std::vector<std::vector<StatTile<Vec3>>>> rgbFeatureTiles(nTilesTotal);
    std::vector<std::vector<std::vector<StatTile<float>>>> misTallyTiles(nTilesTotal);
    auto tileBoundsOf = [&](int tileIndex) {
        const int tx = tileIndex % nTilesX, ty = tileIndex / nTilesX;
        return Bounds2i(Point2i(tx * tileSize, ty * tileSize),
                        Point2i(std::min((tx + 1) * tileSize, width), std::min((ty + 1) * tileSize, height)));
    };
    for (int t = 0; t < nTilesTotal; t++) {  // statpath.cpp:173-190
        lTiles[t] = estimator.GetTiles<T>(tileBoundsOf(t), sCfgs[Radiance].bounceEnd);
        rgbFeatureTiles[t] = estimator.GetTiles<Vec3>(tileBoundsOf(t), 1, nRGBBuffers);
        misTallyTiles[t] = estimator.GetTiles<float>(tileBoundsOf(t), sCfgs[MISBSDFWinRate].bounceEnd, 2);
    }

    const OutputBufferSelection outBufSel(reg, std::regex(o.outputRegex), (o.stem.empty() ? std::string("out") : o.stem) + ".pfm");

    unsigned done = 0;  // samples per pixel so far (the nominal count: the dumps' names)
    // --adaptive (no counterpart in the reference: the consumer of the tile-local moments): after every iteration the tiles
    // are ranked by the noise of their radiance estimate (Estimator::TileNoise: tile mean of the variance of the mean, from
    // the wave-level reduction statmc_tile_moments), and in the next iteration the noisiest quarter gets twice the
    // schedule's samples per pixel, the quietest quarter half of them (at least one).  Counts then differ from tile to tile.
    std::vector<unsigned> tileDone(nTilesTotal, 0u);      // samples per pixel so far, tile by tile
    std::vector<float> tileNoise;                          // of the last iteration (empty: no ranking yet)
    auto RenderLoop = [&](const int nIterations, const bool writeOutput) {
    for (int i = 1; i <= nIterations; i++) {
        const unsigned target = i == 1 ? (unsigned)o.spp : (unsigned)o.spp << std::max(i - 2, 0);  // statpath.cpp:272-279
        std::vector<unsigned> tileTarget(nTilesTotal, target);
        if (o.adaptive && !tileNoise.empty()) {
            std::vector<int> order(nTilesTotal);
            for (int t = 0; t < nTilesTotal; t++) order[t] = t;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tileNoise[a] < tileNoise[b]; });   // ties: tile order
            const int q = nTilesTotal / 4;
            for (int k = 0; k < q; k++) {
                tileTarget[order[k]] = std::max(1u, target / 2);
                tileTarget[order[nTilesTotal - 1 - k]] = 2 * target;
            }
        }
        auto begin = std::chrono::steady_clock::now();
        std::atomic<int> nextTile{0};
        std::atomic<bool> failed{false};
        std::string failure;
        std::mutex failMu;
        auto worker = [&]() {
            try {
                std::vector<Vec3> Ls(nLs);
                for (int t = nextTile.fetch_add(1); t < nTilesTotal; t = nextTile.fetch_add(1)) {
                    const Bounds2i tb = tileBoundsOf(t);
                    const unsigned target = tileTarget[t], done = tileDone[t];   // (shadow the schedule's: this tile's)
                    std::vector<StatTile<T>> &tileLs = lTiles[t];
                    std::vector<std::vector<StatTile<Vec3>>> &tileRGBFeatures = rgbFeatureTiles[t];
                    std::vector<std::vector<StatTile<float>>> &tileMISTallies = misTallyTiles[t];
                    for (int y = tb.pMin.y; y < tb.pMax.y; y++)
                        for (int x = tb.pMin.x; x < tb.pMax.x; x++) {
                            const Point2i actualPixel(x, y);
                            for (unsigned s = 0; s < target; s++) {  // do { ... } while (StartNextSample())
                                const PixelSample smp = makeSample(o.seed, x, y, done + s);
                                // Li() leaves the radiance gathered up to bounce j in Ls[j]; here a fixed share
                                for (unsigned char j = 0; j < nLs; j++) {
                                    const float share = (float)(j + 1) / (float)nLs;
                                    Ls[j] = Vec3{smp.radiance.x * share, smp.radiance.y * share, smp.radiance.z * share};
                                }
                                for (unsigned char j = sCfgs[Radiance].bounceStart; j < sCfgs[Radiance].bounceEnd; j++)
                                    (tileLs[j].*AddLSampleFn)(actualPixel, GetStatSample<T>(Ls[j]));
                                // MIS tallies per bounce: which technique won the sample (statpath.cpp:361-364)
                                for (unsigned char j = sCfgs[MISBSDFWinRate].bounceStart; j < sCfgs[MISBSDFWinRate].bounceEnd; j++) {
                                    const float bsdf = (float)((draw(o.seed, x, y, done + s, 20u + j) >> 3) & 1u);
                                    const float light = (float)((draw(o.seed, x, y, done + s, 40u + j) >> 3) & 1u);
                                    (tileMISTallies[j][0].*AddMISWinRateSampleFn)(actualPixel, bsdf);
                                    (tileMISTallies[j][1].*AddMISWinRateSampleFn)(actualPixel, light);
                                }
                                (tileRGBFeatures[0][0].*AddRGBGBufferSampleFn)(actualPixel, smp.normal);
                                (tileRGBFeatures[0][1].*AddRGBGBufferSampleFn)(actualPixel, smp.albedo);
                            }
                        }
                    // Merge tiles into buffers (statpath.cpp:381-388)
                    if (sCfgs[Radiance].enable) estimator.MergeTransformTiles(tileLs, sCfgs[Radiance]);
                    if (misEnabled) estimator.MergeTiles(tileMISTallies, misCfgs);
                    estimator.MergeTiles(tileRGBFeatures, enabledRGBFeatureCfgs);
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(failMu);
                failed = true;
                failure = e.what();
            }
        };
        std::vector<std::thread> pool;
        for (int t = 0; t < o.threads; t++) pool.emplace_back(worker);
        for (auto &t : pool) t.join();
        if (failed) throw std::runtime_error(failure);
        done += target;
        for (int t = 0; t < nTilesTotal; t++) tileDone[t] += tileTarget[t];
        auto end = std::chrono::steady_clock::now();
        std::cout << "Iteration: " << i << std::endl;
        std::cout << "SPP: " << target << std::endl;
        std::cout << "Rendering time [ns]: " << std::chrono::duration_cast<std::chrono::nanoseconds>(end - begin).count() << std::endl;

        begin = std::chrono::steady_clock::now();
        estimator.Upload();  // stages + accumulates the iteration's samples on the device
        if (estimator.runCUDA) {
            estimator.Denoise();
            estimator.Download();
        }
        estimator.Synchronize();
        end = std::chrono::steady_clock::now();
        std::cout << "CUDA time [ns]: " << std::chrono::duration_cast<std::chrono::nanoseconds>(end - begin).count() << std::endl;
        if (o.adaptive && sCfgs[Radiance].enable) {
            tileNoise = estimator.TileNoise(sCfgs[Radiance].index, 0, tileSize);
            unsigned lo = ~0u, hi = 0;
            for (unsigned v : tileDone) { lo = std::min(lo, v); hi = std::max(hi, v); }
            std::cout << "Adaptive: samples per pixel so far " << lo << " .. " << hi << " over " << nTilesTotal << " tiles" << std::endl;
        }

        begin = std::chrono::steady_clock::now();
        if (!o.stem.empty() && writeOutput) {  // statpath.cpp:419-427: outBufSel.PrepareOutput(); outBufSel.Write(total spp)
            // what lives only on the device comes to the host mats first: the statistics, and the
            // filter's device-only by-products (mean-corr, discriminator)
            estimator.DownloadStatistics();
            if (estimator.runCUDA)
                for (auto *bufs : {&estimator.meanCorrBuffers, &estimator.discriminatorBuffers})
                    for (auto &perType : *bufs)
                        for (Buffer &b : perType) b.download(estimator.stream);
            estimator.Synchronize();
            outBufSel.PrepareOutput();
            outBufSel.Write(std::to_string(done));
            if (o.tileStats && !estimator.filmBuffers.empty() && !estimator.filmBuffers[0].empty()) {
                // --tilestats: the local noise level of the radiance estimate, one value per 16 x 16 tile (the tile
                // size of the render loop): "<stem>-<spp>-t0-b0-tile-mean.pfm" and "-tile-var.pfm" (pooled sample
                // variance M2 / (count - 1) of the per-pixel means inside the tile), from wave-level reductions
                const Buffer &fm = estimator.filmBuffers[0][0];
                const int C = fm.gpuMat.channels();
                const HostImage tm = estimator.TileMoments(fm, 16);
                const int ty = tm.rows, tx = tm.cols / C;
                std::vector<float> mean((size_t)ty * tx * C), var((size_t)ty * tx * C);
                float worst = -1.f;
                int wx = 0, wy = 0;
                for (int y = 0; y < ty; y++)
                    for (int x = 0; x < tx; x++)
                        for (int c = 0; c < C; c++) {
                            const float *m = tm.ptr<float>() + (((size_t)y * tx + x) * C + c) * 3;
                            const size_t k = ((size_t)y * tx + x) * C + c;
                            mean[k] = m[1];
                            var[k] = m[0] > 1.f ? m[2] / (m[0] - 1.f) : 0.f;
                            if (var[k] > worst) { worst = var[k]; wx = x; wy = y; }
                        }
                const std::string prefix = o.stem + "-" + std::to_string(done) + "-" + fm.name + "-";
                writePfm(prefix.substr(0, prefix.size() - std::string("film-mean-").size()) + "tile-mean.pfm", tx, ty, C, mean.data());
                writePfm(prefix.substr(0, prefix.size() - std::string("film-mean-").size()) + "tile-var.pfm", tx, ty, C, var.data());
                std::cout << "Noisiest tile: (" << wx << ", " << wy << ") variance " << worst << std::endl;
            }
        }
        end = std::chrono::steady_clock::now();
        std::cout << "Output time [ns]: " << std::chrono::duration_cast<std::chrono::nanoseconds>(end - begin).count() << std::endl;
    }
    };

    if (o.warmUp) {  // statpath.cpp:433-437: one throw-away iteration, then everything starts over
        std::cout << "==== Warm-Up Start ====" << std::endl;
        RenderLoop(1, false);
        std::cout << "==== Warm-Up End ====" << std::endl;
        // the reference re-creates its tiles, i.e. all statistics, for the real run
        for (unsigned char t = 0; t < estimator.statTypeConfigs.nEnabled; t++) estimator.ResetStatistics(t);
        done = 0;
        std::fill(tileDone.begin(), tileDone.end(), 0u);
        tileNoise.clear();
    }
    RenderLoop(o.iterations, true);
}

int main(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--width") o.width = std::atoi(next());
        else if (a == "--height") o.height = std::atoi(next());
        else if (a == "--spp") o.spp = std::atoi(next());
        else if (a == "--iterations") o.iterations = std::atoi(next());
        else if (a == "--threads") o.threads = std::atoi(next());
        else if (a == "--seed") o.seed = (unsigned)std::strtoul(next(), nullptr, 10);
        else if (a == "--filtersd") o.filterSD = (float)std::atof(next());
        else if (a == "--filterradius") o.filterRadius = std::atoi(next());
        else if (a == "--stage-mb") o.stageMb = std::atoi(next());
        else if (a == "--trackedbounces") o.trackedBounces = std::atoi(next());
        else if (a == "--stem") o.stem = next();
        else if (a == "--outputregex") o.outputRegex = next();
        else if (a == "--no-denoise") o.denoise = false;
        else if (a == "--warmup") o.warmUp = true;
        else if (a == "--tilestats") o.tileStats = true;
        else if (a == "--adaptive") o.adaptive = true;
        else if (a == "--placed") statmc::usePlacedMemory() = true;   // device images and sample arenas from statmc_malloc_placed
        else if (a == "--config") {
            const std::string c = next();
            if (c != "denoise" && c != "acrr" && c != "smis") {
                std::fprintf(stderr, "unknown config %s\n", c.c_str());
                return 2;
            }
            o.acrr = c == "acrr";
            o.smis = c == "smis";
        } else if (a == "--print-sample") {  // x y s: the generator alone, no device (CPU test of the restatement)
            const int x = std::atoi(next()), y = std::atoi(next());
            const unsigned s = (unsigned)std::strtoul(next(), nullptr, 10);
            const PixelSample p = makeSample(o.seed, x, y, s);
            std::printf("%a %a %a %a %a %a %a %a %a\n", p.radiance.x, p.radiance.y, p.radiance.z, p.normal.x, p.normal.y,
                        p.normal.z, p.albedo.x, p.albedo.y, p.albedo.z);
            return 0;
        } else {
            std::fprintf(stderr, "unknown option %s\n", a.c_str());
            return 2;
        }
    }
    if (o.width <= 0 || o.height <= 0 || o.width > 65535 || o.height > 65535 || o.spp <= 0 || o.iterations <= 0 ||
        o.threads <= 0 || o.trackedBounces < 1 || o.trackedBounces > 16) {
        std::fprintf(stderr, "bad size / spp / iterations / threads / trackedbounces\n");
        return 2;
    }
    try {
        if (o.acrr) Render<float>(o);  // StatPathIntegrator::Render dispatches on enableMultiChannelStats
        else Render<Vec3>(o);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "statmc_render_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
