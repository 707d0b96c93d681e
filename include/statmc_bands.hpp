// statmc_bands.hpp -- Upload / Denoise / Download as a pipeline of row bands (shared by statmc::Estimator in
// statmc_denoiser.hpp and by the cv::cuda adaptor in statmc_cv.hpp).
//
// The reference brackets Estimator::Upload(); Denoise(); Download(); Synchronize() as its "CUDA time"
// (src/statistics/statpath.cpp:409-417; estimator.cpp:409-489, 571-573).  All four only enqueue work, so nothing
// obliges them to run one after the other: the image is cut into bands of rows, a transfer carries one band plus the
// r rows below it (its lower halo), the band is pre-passed and filtered as soon as that transfer has arrived and copied
// back as soon as it is filtered.  Copies in (two queues), kernels and copies out sit on their own streams, ordered by events
// (statmc_event_record / statmc_stream_wait_event); what is left after the last copy in is one band's filter.  Where the
// bands lie is a Plan (below): by default fitted to the window filter's rounds, the last band the short one.
// The results are the same bits as with one stream: the pre-pass is per pixel, and the window filter forms a pixel's
// sums in the same order for any output region (the library chooses its window-sweep parts for the whole image).
#ifndef STATMC_BANDS_HPP
#define STATMC_BANDS_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "statmc.h"

namespace statmc {
namespace bands {

inline void ok(int rc, const char *what) {
    if (rc != STATMC_OK) throw std::runtime_error(std::string(what) + ": " + statmc_last_error());
}

// halo rows of a transfer: the filter radius, rounded up to whole 8-row filter tiles
inline int halo(int radius) { return (radius + 7) & ~7; }
// Number of bands for an image of `height` rows when the caller asks for a count: 1 -> off; never a band shorter than 64
// rows or than its own halo.
inline int count(int height, int radius, int requested) {
    const int n = std::max(1, requested);
    const int minRows = std::max(64, halo(radius));
    return std::max(1, std::min(n, height / minRows));
}

// Where the bands of an image lie.  Edges are multiples of 8 rows (whole filter tiles, 16-byte aligned sub-images).
//   requested > 1: that many bands of equal height (as far as the minimum height allows).
//   requested = 1: no bands.
//   requested = 0: automatic -- none below 512 rows; above, bands FITTED TO THE FILTER'S ROUNDS.  The window filter runs
//     one workgroup per 128 x 8 tile and CU, and to filter rows [y0, y1) the pair-symmetric kernel also sweeps the
//     ceil(r / 8) tile rows above y0 (their taps reach into the band), so a band of b tile rows costs
//     ceil((b + ceil(r / 8)) * tiles_per_row / CUs) rounds of ~0.2 ms whatever it produces: six equal bands of a 1080p film
//     are 345 + 45 workgroups each = two rounds for 22.5 tile rows, where two rounds hold 31 (+ 3).  The automatic plan
//     picks the number of rounds per band that gives at most six bands and makes every band but the last exactly as
//     tall as those rounds hold; the last band takes the rest and is the short one -- it is what remains after the last copy
//     in.  (1080p: 248, 248, 248, 248, 88 rows: the bracket 3.88 -> 3.72 ms; tools/experiments/upload_modes.py.)
struct Plan {
    std::vector<int> edges{0, 0};   // edges[k] .. edges[k + 1]: band k
    int height = 0, radius = 0;
    int count() const { return (int)edges.size() - 1; }
    int edge(int k) const { return k <= 0 ? 0 : k >= count() ? height : edges[k]; }
    // transfer k carries rows [arrival(k), arrival(k + 1)): band k shifted down by its halo
    int arrival(int k) const {
        if (k <= 0) return 0;
        if (k >= count()) return height;
        return std::min(height, edge(k) + halo(radius));
    }
};
constexpr int kTileW = 128, kTileH = 8, kAutoBandsMax = 6;   // the pair-symmetric kernel's tile
// cus: compute units of the device the filter runs on (statmc_device_cus(); 256 on a whole MI355X, fewer on a partitioned one)
inline Plan plan(int width, int height, int radius, int requested, int cus = 0) {
    const int devCUs = cus > 0 ? cus : statmc_device_cus();
    const int kFilterCUs = devCUs > 0 ? devCUs : 256;   // (no device set up: the plan of a whole MI355X)
    Plan p;
    p.height = height;
    p.radius = radius;
    p.edges = {0, height};
    const int minRows = std::max(64, halo(radius));
    if (requested == 1 || height < 2 * minRows || (requested == 0 && height < 512)) return p;
    p.edges.clear();
    if (requested > 1) {
        const int n = count(height, radius, requested);
        for (int k = 0; k < n; k++) p.edges.push_back((int)(((long long)height * k / n) & ~7LL));
        p.edges.push_back(height);
        return p;
    }
    if (const char *s = std::getenv("STATMC_BANDS_EDGES"); s && *s) {   // experiment: "y1,y2,..." places the edges by hand
        p.edges.push_back(0);
        for (const char *q = s; *q;) {
            const int y = std::atoi(q) & ~7;
            if (y > p.edges.back() && y < height) p.edges.push_back(y);
            while (*q && *q != ',') q++;
            if (*q == ',') q++;
        }
        p.edges.push_back(height);
        return p;
    }
    const int tilesPerRow = (width + kTileW - 1) / kTileW, tileRows = (height + kTileH - 1) / kTileH;
    const int above = (radius + kTileH - 1) / kTileH;
    int rounds = std::max(1, (int)(((long long)tileRows * tilesPerRow + (long long)kFilterCUs * kAutoBandsMax - 1) / ((long long)kFilterCUs * kAutoBandsMax)));
    int bandTileRows = 0;
    for (;; rounds++) {   // the tallest band those rounds hold; more rounds while that is less than the minimum or gives too many bands
        bandTileRows = rounds * kFilterCUs / tilesPerRow - above;
        if (bandTileRows * kTileH >= minRows && (tileRows + bandTileRows - 1) / bandTileRows <= kAutoBandsMax) break;
    }
    // (the end of the image in one-round bands -- 248, 248, 248, 112, 112, 112 -- was tried: best iterations 3.63 ms, but a
    // bimodal 3.63 / 3.85 and the same mean as the plain plan's steady 3.74)
    for (int y = 0; y < height; y += bandTileRows * kTileH) p.edges.push_back(y);
    if (p.edges.size() > 1 && height - p.edges.back() < minRows) p.edges.pop_back();   // a last band below the minimum joins the one before
    p.edges.push_back(height);
    return p;
}

// rows [y0, y1) of an image as an image of its own (rows are contiguous: a band is a sub-array)
inline statmc_image rows(const statmc_image &im, int y0, int y1) {
    statmc_image r = im;
    r.data = static_cast<char *>(im.data) + (size_t)y0 * im.step;
    r.rows = y1 - y0;
    return r;
}
// pre-pass of rows [y0, y1) of the buffers of `a` (per-pixel work: the band is passed as a shorter image)
inline void prepassRows(const statmc_filter_args &a, int channels, int y0, int y1) {
    const size_t nb = a.n_buffers;
    std::vector<statmc_image> n(nb), mean(nb), m2(nb), m3(nb), mc(nb), dc(nb);
    for (size_t b = 0; b < nb; b++) {
        n[b] = rows(a.n[b], y0, y1); mean[b] = rows(a.mean[b], y0, y1); m2[b] = rows(a.m2[b], y0, y1);
        m3[b] = rows(a.m3[b], y0, y1); mc[b] = rows(a.mean_corr[b], y0, y1); dc[b] = rows(a.discriminator[b], y0, y1);
    }
    statmc_filter_args p = a;
    p.height = (uint16_t)(y1 - y0);
    p.n = n.data(); p.mean = mean.data(); p.m2 = m2.data(); p.m3 = m3.data();
    p.mean_corr = mc.data(); p.discriminator = dc.data();
    ok(statmc_prepass(&p, channels), "statmc_prepass");
}
// window filter of rows [y0, y1) (the window still reads the whole image)
inline void filterRows(const statmc_filter_args &a, int channels, int y0, int y1) {
    statmc_filter_args f = a;
    f.roi_x0 = 0; f.roi_x1 = a.width; f.roi_y0 = y0; f.roi_y1 = y1;
    ok(statmc_window_filter(&f, channels), "statmc_window_filter");
}

// The copy streams and the per-band events of one pipeline.  Copies in are dealt over TWO streams: behind one queue
// every copy pays ~9.5 us between its predecessor's completion and its own start (42 pieces of the 157.6 MB of a
// 1080p upload: 3.21 ms instead of 2.88 for 7 whole images); on two queues the gaps of one hide behind the other's
// transfers (2.77 ms, 56.8 GB/s -- tools/experiments/copy_granularity.py).
struct Streams {
    void *up = nullptr, *up2 = nullptr, *down = nullptr, *join = nullptr;
    std::vector<void *> arrived, arrived2, filtered;   // per band: transfer landed (per copy stream) / band filtered
    // Which of the two copy streams carries each image of a transfer: bytes balanced (largest first onto the lighter
    // queue), so that both queues finish a transfer together.
    static std::vector<int> deal(const std::vector<size_t> &bytes, int queues = 2) {
        if (queues < 2) return std::vector<int>(bytes.size(), 0);
        std::vector<size_t> order(bytes.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return bytes[a] > bytes[b]; });
        std::vector<int> q(bytes.size(), 0);
        size_t load[2] = {0, 0};
        for (size_t i : order) {
            const int s = load[1] < load[0] ? 1 : 0;
            q[i] = s;
            load[s] += bytes[i];
        }
        return q;
    }
    void *upStream(int q) const { return q ? up2 : up; }
    void upload(int q, void *dst, const void *src, size_t bytes) const { ok(statmc_upload(dst, src, bytes, upStream(q)), "statmc_upload"); }
    // transfer k starts on both queues only when transfer k - 1 has landed on both: the queues stay in step, a band is
    // complete when it would have been on one queue (minus the gaps)
    void beginTransfer(int k) const {
        if (k <= 0) return;
        static const bool freeRunning = [] { const char *e = std::getenv("STATMC_BANDS_FREE"); return e && e[0] == '1'; }();   // experiment
        if (freeRunning) return;
        ok(statmc_stream_wait_event(up, arrived2[k - 1]), "statmc_stream_wait_event");
        ok(statmc_stream_wait_event(up2, arrived[k - 1]), "statmc_stream_wait_event");
    }
    // enqueue on `stream`: wait until transfer k has landed
    void waitArrived(void *stream, int k) const {
        ok(statmc_stream_wait_event(stream, arrived[k]), "statmc_stream_wait_event");
        ok(statmc_stream_wait_event(stream, arrived2[k]), "statmc_stream_wait_event");
    }
    // the copies of a new round must not overtake what `stream` has been given so far
    void beginUploads(void *stream) const {
        ok(statmc_event_record(join, stream), "statmc_event_record");
        ok(statmc_stream_wait_event(up, join), "statmc_stream_wait_event");
        ok(statmc_stream_wait_event(up2, join), "statmc_stream_wait_event");
    }
    void markArrived(int k) const {
        ok(statmc_event_record(arrived[k], up), "statmc_event_record");
        ok(statmc_event_record(arrived2[k], up2), "statmc_event_record");
    }
    Streams() = default;
    Streams(const Streams &) = delete;
    Streams &operator=(const Streams &) = delete;
    ~Streams() { destroy(); }
    void ensure(int nb) {
        if (!up) {
            // The copy streams live in other priority classes than the (normal-priority) kernel stream: the runtime keeps a
            // pool of GPU_MAX_HW_QUEUES = 4 hardware queues per priority level and lets a fifth stream of a level share a
            // queue with an arbitrary other one (which one depends on address order, i.e. on the process); a copy
            // stream that shares the kernel stream's hardware queue parks its event-wait barrier packets in front of the
            // kernels, and the pipeline degenerates (7.0 instead of 4.0 ms for the 1080p bracket in one process out of
            // three; always with GPU_MAX_HW_QUEUES=2 -- tools/experiments/diagnose_queues.py).  STATMC_BANDS_SAME_PRIORITY=1
            // restores the old behaviour for A/B runs.
            static const bool flat = [] { const char *e = std::getenv("STATMC_BANDS_SAME_PRIORITY"); return e && e[0] == '1'; }();
            int pu = flat ? 0 : 1, pu2 = flat ? 0 : 1, pd = flat ? 0 : -1;
            if (const char *e = std::getenv("STATMC_BANDS_PRIO")) std::sscanf(e, "%d,%d,%d", &pu, &pu2, &pd);   // experiments
            ok(statmc_stream_create_with_priority(&up, pu), "statmc_stream_create");
            ok(statmc_stream_create_with_priority(&up2, pu2), "statmc_stream_create");
            ok(statmc_stream_create_with_priority(&down, pd), "statmc_stream_create");
            ok(statmc_event_create(&join), "statmc_event_create");
        }
        while ((int)arrived.size() < nb) {
            void *a = nullptr, *a2 = nullptr, *f = nullptr;
            ok(statmc_event_create(&a), "statmc_event_create");
            ok(statmc_event_create(&a2), "statmc_event_create");
            ok(statmc_event_create(&f), "statmc_event_create");
            arrived.push_back(a);
            arrived2.push_back(a2);
            filtered.push_back(f);
        }
    }
    void destroy() {
        if (!up) return;
        statmc_synchronize(up);
        statmc_synchronize(up2);
        statmc_synchronize(down);
        for (void *e : arrived) statmc_event_destroy(e);
        for (void *e : arrived2) statmc_event_destroy(e);
        for (void *e : filtered) statmc_event_destroy(e);
        statmc_event_destroy(join);
        statmc_stream_destroy(up);
        statmc_stream_destroy(up2);
        statmc_stream_destroy(down);
        up = up2 = down = join = nullptr;
        arrived.clear();
        arrived2.clear();
        filtered.clear();
    }
};

}  // namespace bands
}  // namespace statmc

#endif  // STATMC_BANDS_HPP
