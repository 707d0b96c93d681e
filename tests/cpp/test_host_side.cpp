// CPU-only checks of the C++ host side (include/statmc_denoiser.hpp), built with
// -fsanitize=address,undefined by tests/test_host_cpu.py: the StatTile recorder, Estimator::GetTiles,
// the error paths that need no device, OutputBufferSelection + the PFM codec.  Links against
// libstatmc_hip.so only for the symbols; nothing here touches a GPU.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <thread>
#include <atomic>

#include "statmc_denoiser.hpp"

using namespace statmc;

#define REQUIRE(cond)                                                              \
    do {                                                                           \
        if (!(cond)) {                                                             \
            std::fprintf(stderr, "%s:%d: REQUIRE(%s) failed\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                          \
        }                                                                          \
    } while (0)

template <class F>
static bool throwsError(F f, int code) {
    try {
        f();
    } catch (const Error &e) {
        return e.code == code;
    }
    return false;
}

int main(int argc, char **argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";

    // ---- StatTile: same index rule as Tile<T>::GetPixel (estimator.h:44-48), one plane per sample
    {
        StatTile<Vec3> tile(Bounds2i(Point2i(16, 32), Point2i(29, 40)));  // 13 x 8, clipped edge tile
        REQUIRE(tile.GetPixelBounds().Area() == 13 * 8);
        for (int s = 0; s < 3; s++)
            for (int y = 32; y < 40; y++)
                for (int x = 16; x < 29; x++) tile.AddTransformSampleM3(Point2i(x, y), Vec3{(float)x, (float)y, (float)s});
        REQUIRE(tile.pending(Point2i(16, 32)) == 3 && tile.pending(Point2i(28, 39)) == 3);
        void (StatTile<Vec3>::*fn)(const Point2i, const Vec3) = &StatTile<Vec3>::AddSampleM1;  // binds like statpath.cpp:166
        (tile.*fn)(Point2i(20, 35), Vec3{1, 2, 3});
        REQUIRE(tile.pending(Point2i(20, 35)) == 4);
        StatTile<float> empty(Bounds2i(Point2i(5, 5), Point2i(5, 9)));  // degenerate bounds: no pixels
        REQUIRE(empty.GetPixelBounds().Area() == 0);
    }

    // ---- Estimator without a device: catalogue, GetTiles, merge refused, config errors
    {
        StatPathParams p;
        p.denoiseImage = true;
        const StatTypeConfigs cfgs = makeStatTypeConfigs(p);
        Buffer film("film", HostImage(24, 40, F32C3), /*allocateDevice=*/false);
        BufferRegistry reg(film);
        Estimator est(film, cfgs, 10.f, 20, false, false, false, reg, /*allocateDevice=*/false);
        est.AllocateBuffers(reg);
        REQUIRE(reg.find("t0-b0-m3") && reg.find("t2-b0-film-mean") && !reg.find("t3-b0-n"));
        auto tiles = est.GetTiles<Vec3>(Bounds2i(Point2i(0, 0), Point2i(16, 16)), cfgs[Radiance].bounceEnd);
        REQUIRE(tiles.size() == 1);
        auto nested = est.GetTiles<Vec3>(Bounds2i(Point2i(0, 0), Point2i(16, 16)), 1, 2);
        // GetTilesF (estimator.cpp:312-338): the pixels that samples of [16, 32)^2 reach under the pixel filter, cut to
        // the cropped bounds (film 40 x 24).  Box filter of radius 0.5: Ceil(16 - 1) .. Floor(32) + 1 = [15, 33);
        // radius 2: Ceil(13.5) .. Floor(33.5) + 1 = [14, 34).
        {
            auto f = est.GetTilesF<Vec3>(Bounds2i(Point2i(16, 16), Point2i(32, 32)), 2);
            const Bounds2i b = f[0].GetPixelBounds();
            REQUIRE(f.size() == 2 && b.pMin.x == 15 && b.pMin.y == 15 && b.pMax.x == 33 && b.pMax.y == 24);
            static const float table[4] = {1.f, .5f, .5f, .25f};
            est.SetPixelFilter(statmc::Vector2f(2.f, 2.f), table, 2);
            auto g = est.GetTilesF<float>(Bounds2i(Point2i(16, 16), Point2i(32, 32)), 1, 3);
            const Bounds2i c = g[0][2].GetPixelBounds();
            REQUIRE(g.size() == 1 && g[0].size() == 3 && c.pMin.x == 14 && c.pMin.y == 14 && c.pMax.x == 34 && c.pMax.y == 24);
            REQUIRE(g[0][0].GetFilterTable() == table && g[0][0].GetFilterTableSize() == 2 && g[0][0].GetFilterRadius().x == 2.f);
            g[0][1].AddSampleM2(Point2i(14, 23), 1.5f);           // a filtered tile records like any other
            REQUIRE(g[0][1].pending(Point2i(14, 23)) == 1);
            est.SetCroppedPixelBounds(Bounds2i(Point2i(0, 0), Point2i(20, 20)));
            const Bounds2i d = est.GetTilesF<float>(Bounds2i(Point2i(16, 16), Point2i(32, 32)), 1)[0].GetPixelBounds();
            REQUIRE(d.pMin.x == 14 && d.pMin.y == 14 && d.pMax.x == 20 && d.pMax.y == 20);
            REQUIRE(est.GetTilesF<float>(Bounds2i(Point2i(30, 30), Point2i(32, 32)), 1)[0].GetPixelBounds().Area() == 0);
            est.SetCroppedPixelBounds(Bounds2i());
            est.SetPixelFilter(statmc::Vector2f(.5f, .5f), nullptr, 0);
        }
        REQUIRE(nested.size() == 1 && nested[0].size() == 2);
        tiles[0].AddTransformSampleM3(Point2i(3, 4), Vec3{1, 1, 1});
        REQUIRE(throwsError([&] { est.MergeTransformTiles(tiles, cfgs[Radiance]); }, STATMC_ERR_INVALID));   // not enabled
        REQUIRE(throwsError([&] { est.EnableDeviceAccumulation(); }, STATMC_ERR_INVALID));                    // no device images
        StatPathParams bad;
        bad.denoiseImage = true;
        bad.filterBufferSDs = {0.1f};
        REQUIRE(throwsError([&] { makeStatTypeConfigs(bad); }, STATMC_ERR_INVALID));

        // ---- OutputBufferSelection: regex over the registry, PrepareOutput converts n, Write -> PFM
        HostImage &nMat = est.nBuffers[0][0].mat;
        for (int i = 0; i < 24 * 40; i++) nMat.ptr<int32_t>()[i] = i % 7;
        float *mean = est.meanBuffers[0][0].mat.ptr<float>();
        for (int i = 0; i < 24 * 40 * 3; i++) mean[i] = 0.25f * (float)i;
        const OutputBufferSelection sel(reg, std::regex("t0-b0-(n|mean)"), tmp + "/host.pfm");
        REQUIRE(sel.selected().size() == 2 && sel.GetFilenameStem() == tmp + "/host");
        sel.PrepareOutput();
        sel.Write("8");
        const PfmImage n = readPfm(tmp + "/host-8-t0-b0-n.pfm"), m = readPfm(tmp + "/host-8-t0-b0-mean.pfm");
        REQUIRE(n.channels == 1 && n.width == 40 && n.height == 24 && m.channels == 3);
        for (int i = 0; i < 24 * 40; i++) REQUIRE(n.data[i] == (float)(i % 7));
        for (int i = 0; i < 24 * 40 * 3; i++) REQUIRE(m.data[i] == 0.25f * (float)i);
        const OutputBufferSelection all(reg, tmp + "/all.pfm");
        REQUIRE(all.selected().size() == reg.buffers.size());
        const OutputBufferSelection png(reg, std::regex("film"), tmp + "/x.png");
        REQUIRE(throwsError([&] { png.Write(); }, STATMC_ERR_UNSUPPORTED));
    }
    // ---- the staging logic under threads (dry run: no device): 8 workers merge 3 buffers of every tile
    // of a 200 x 120 film for 3 iterations through a staging so small that it flushes many times
    {
        StatPathParams p;
        p.denoiseImage = true;
        const StatTypeConfigs cfgs = makeStatTypeConfigs(p);
        const int W = 200, H = 120, ts = 16, tx = (W + ts - 1) / ts, ty = (H + ts - 1) / ts, nTiles = tx * ty;
        Buffer film("film", HostImage(H, W, F32C3), false);
        BufferRegistry reg(film);
        Estimator est(film, cfgs, 10.f, 20, false, false, false, reg, false);
        est.AllocateBuffers(reg);
        est.EnableDeviceAccumulation((size_t)1 << 20, /*dryRun=*/true);
        std::vector<StatTypeConfig> feat = {cfgs[StatNormal], cfgs[StatAlbedo]};
        std::vector<std::vector<StatTile<Vec3>>> lTiles(nTiles);
        std::vector<std::vector<std::vector<StatTile<Vec3>>>> fTiles(nTiles);
        auto bounds = [&](int t) {
            const int x = t % tx, y = t / tx;
            return Bounds2i(Point2i(x * ts, y * ts), Point2i(std::min((x + 1) * ts, W), std::min((y + 1) * ts, H)));
        };
        for (int t = 0; t < nTiles; t++) {
            lTiles[t] = est.GetTiles<Vec3>(bounds(t), 1);
            fTiles[t] = est.GetTiles<Vec3>(bounds(t), 1, 2);
        }
        for (int it = 0; it < 3; it++) {
            std::atomic<int> next{0};
            std::vector<std::thread> pool;
            for (int w = 0; w < 8; w++)
                pool.emplace_back([&] {
                    for (int t = next.fetch_add(1); t < nTiles; t = next.fetch_add(1)) {
                        const Bounds2i b = bounds(t);
                        for (int y = b.pMin.y; y < b.pMax.y; y++)
                            for (int x = b.pMin.x; x < b.pMax.x; x++)
                                for (int s = 0; s < 5; s++) {
                                    lTiles[t][0].AddTransformSampleM3(Point2i(x, y), Vec3{1.f, 2.f, (float)s});
                                    fTiles[t][0][0].AddSampleM1(Point2i(x, y), Vec3{0.f, 0.f, 1.f});
                                    fTiles[t][0][1].AddSampleM1(Point2i(x, y), Vec3{.5f, .5f, .5f});
                                }
                        est.MergeTransformTiles(lTiles[t], cfgs[Radiance]);
                        est.MergeTiles(fTiles[t], feat);
                    }
                });
            for (auto &th : pool) th.join();
            est.Upload();  // flushes the rest
            REQUIRE(est.stagedMerges() == (size_t)(it + 1) * nTiles * 3);
        }
        REQUIRE(est.flushes() > 10);  // the 1 MiB staging filled up many times
        REQUIRE(lTiles[0][0].pending(Point2i(0, 0)) == 0);
    }
    std::puts("host side ok");
    return 0;
}
