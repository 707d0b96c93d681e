// The reference's OWN StatTile<T> (src/statistics/estimator.h:147-239, patched by patches/0001-0003 so that
// statpbrt.h includes include/statmc_cv.hpp instead of OpenCV) run on the CPU: what this exercises is PRODUCT code --
// the cv::Vec arithmetic of include/statmc_cv.hpp, on which the patched reference's CPU-side accumulation runs --
// through the reference's own update sequence.  tools/check_reference_compiles.sh builds and runs it (container only:
// the reference's sources never travel), tests/test_host_cpu.py compares what it writes with oracle_add_sample bit for
// bit.  It constructs only StatTile<Float> / StatTile<Vec3> -- no Estimator, no setup(), no GPU.
//
// NOT a pin of the oracle: the build needs a logging stub for <glog/logging.h> and this repository's cv:: stand-in,
// which is exactly what the rules exclude as `oracle/_ref`; `parity` stays "unpinned".
//
// in:  argv[1] = int32 W, H, S; int32 count[H][W]; float sample[S][H][W][3]
// out: argv[2] = for T in (Float, Vec3), transform in (0, 1), maxMoment in (1, 2, 3): StatTilePixel<T>[H][W], raw
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "statistics/estimator.h"

using namespace pbrt;

template <class T> T sampleOf(const float *s);
template <> Float sampleOf<Float>(const float *s) { return s[0]; }
template <> Vec3 sampleOf<Vec3>(const float *s) { return Vec3(s[0], s[1], s[2]); }

template <class T>
static void run(int W, int H, int S, const std::vector<int32_t> &count, const std::vector<float> &smp, bool transform, int moment, FILE *out) {
    StatTile<T> tile(Bounds2i(Point2i(0, 0), Point2i(W, H)));
    // the member-function pointers StatPathIntegrator::Render selects (statpath.cpp:97-116)
    void (StatTile<T>::*fn)(const Point2i, const T) =
        transform ? (moment == 1 ? &StatTile<T>::AddTransformSampleM1 : moment == 2 ? &StatTile<T>::AddTransformSampleM2 : &StatTile<T>::AddTransformSampleM3)
                  : (moment == 1 ? &StatTile<T>::AddSampleM1 : moment == 2 ? &StatTile<T>::AddSampleM2 : &StatTile<T>::AddSampleM3);
    for (int s = 0; s < S; s++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++)
                if (s < count[(size_t)y * W + x]) (tile.*fn)(Point2i(x, y), sampleOf<T>(&smp[(((size_t)s * H + y) * W + x) * 3]));
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const StatTilePixel<T> &px = tile.GetPixel(Point2i(x, y));
            std::fwrite(&px, sizeof(px), 1, out);
        }
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *in = std::fopen(argv[1], "rb");
    if (!in) return 3;
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, in) != 3) return 4;
    const int W = hdr[0], H = hdr[1], S = hdr[2];
    std::vector<int32_t> count((size_t)W * H);
    std::vector<float> smp((size_t)S * H * W * 3);
    if (std::fread(count.data(), 4, count.size(), in) != count.size() || std::fread(smp.data(), 4, smp.size(), in) != smp.size()) return 5;
    std::fclose(in);
    static_assert(sizeof(StatTilePixel<Float>) == 64 && sizeof(StatTilePixel<Vec3>) == 128, "StatTilePixel<T> layout (SURVEY 8 a1)");
    FILE *out = std::fopen(argv[2], "wb");
    if (!out) return 6;
    for (int transform = 0; transform < 2; transform++)
        for (int moment = 1; moment <= 3; moment++) run<Float>(W, H, S, count, smp, transform != 0, moment, out);
    for (int transform = 0; transform < 2; transform++)
        for (int moment = 1; moment <= 3; moment++) run<Vec3>(W, H, S, count, smp, transform != 0, moment, out);
    std::fclose(out);
    return 0;
}
