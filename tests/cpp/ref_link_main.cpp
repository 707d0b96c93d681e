// Link check of the drop-in claim (tools/check_reference_compiles.sh, container only): the reference's OWN
// src/statistics/estimator.cpp + buffer.cpp, patched by patches/0001-0003 and compiled against include/statmc_cv.hpp,
// link with this main against libstatmc_hip.so alone -- no OpenCV, no CUDA.  The calls follow the reference's use of
// its Estimator (statpath.cpp:33-84 construction, 397-413 per-iteration bracket).  Never run on the CPU box (the
// Estimator constructor calls stat_denoiser::setup(), which needs a gfx950 device) and never shipped to the GPU box
// (the reference's sources do not travel): it exists to be compiled and linked.
#include "statistics/estimator.h"

using namespace pbrt;

int main() {
    const int w = 64, h = 48;
    Buffer film("film", Mat3(h, w));
    BufferRegistry reg(film);
    StatTypeConfigs cfgs;
    // shipped default (scenes/render-denoise.pbrt): radiance RGB transform M3 in the denoise group, normal / albedo as G-buffers
    StatTypeConfig rad;
    rad.type = 0; rad.index = 0; rad.enable = true; rad.nBounces = 1; rad.bounceEnd = 1; rad.nChannels = 3;
    rad.transform = true; rad.maxMoment = 3; rad.cudaGroups = {DenoiseGroup};
    cfgs.configs.push_back(rad);
    for (int g = 0; g < 2; g++) {
        StatTypeConfig c;
        c.type = (unsigned char)(1 + g); c.index = (unsigned char)(1 + g); c.enable = true; c.nBounces = 1; c.bounceEnd = 1; c.nChannels = 3;
        c.gBuffer = true; c.enableForFilter = true; c.filterSD = g == 0 ? 0.1f : 0.02f;
        cfgs.configs.push_back(c);
    }
    cfgs.nEnabled = 3;
    Estimator est(film, cfgs, 10.f, 20, true, false, false, 4, reg, Bounds2i(Point2i(0, 0), Point2i(w, h)), nullptr);
    est.AllocateBuffers(reg);
    auto tiles = est.GetTiles<Vec3>(Bounds2i(Point2i(0, 0), Point2i(16, 16)), 1);
    tiles[0].AddTransformSampleM3(Point2i(3, 4), Vec3(0.25f, 1.5f, 0.f));
    est.MergeTransformTiles(tiles, cfgs[0]);
    est.Upload();
    est.Denoise();
    est.Download();
    est.Synchronize();
    est.CalculateMeanVars();
    return 0;
}
