// statmc_cv.hpp -- the OpenCV names StatMC's statistics path uses, on top of libstatmc_hip.so.
//
// StatMC's src/statistics/{statpbrt.h,buffer.{h,cpp},estimator.{h,cpp},statpath.{h,cpp}} and core/film.{h,cpp} use
// OpenCV for exactly five things: `cv::Vec<Float,3>` as the RGB value type, `cv::Mat` /
// `cv::Mat_<T>` as refcounted host images, `cv::cuda::GpuMat` / `cv::cuda::Stream` as device images and the one
// stream, the device tables of `cv::cuda::PtrStepSzb` handed to the denoiser, and PFM file I/O
// (`imwrite / imread / cvtColor / merge / glob`); and they call `cv::cuda::stat_denoiser::{setup, filter<T>,
// synchronize}`, which lives in the authors' fork of opencv_contrib (src/ext/opencv_contrib, .gitmodules:19-21).
// This header provides that subset -- own code, API-compatible for the calls the tree makes -- so that the
// reference's sources compile UNCHANGED once statpbrt.h includes this file instead of <opencv2/...>
// (patches/0001, 0002): the reference's Estimator, its Buffer registry and StatPathIntegrator : SamplerIntegrator
// (the pbrt-v3 plugin surface, src/statistics/statpath.h:48-138) stay what they are, Upload / Denoise / Download run
// on the MI355X.  No OpenCV, no CUDA headers.
//
// Reference call sites this is written against: statpbrt.h:11-28; buffer.h:19-71; buffer.cpp:34-71;
// estimator.h:127-145,252-281,326-373; estimator.cpp:35-84,127-146,287-288,409-489,524-573; core/film.h:79-90;
// statpath.cpp:306-311,370,449-454,479-481; statpath.h:140-147.
#ifndef STATMC_CV_HPP
#define STATMC_CV_HPP

#include <glob.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include <map>
#include <set>

#include "statmc.h"
#include "statmc_bands.hpp"
#include "statmc_pfm.hpp"

typedef unsigned char uchar;
typedef unsigned short ushort;

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) (((depth) & 7) + (((cn) - 1) << CV_CN_SHIFT))
#define CV_MAT_DEPTH(type) ((type) & 7)
#define CV_MAT_CN(type) ((((type) >> CV_CN_SHIFT) & 511) + 1)
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC(n) CV_MAKETYPE(CV_8U, (n))
#define CV_32SC1 CV_MAKETYPE(CV_32S, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)

namespace cv {

typedef std::string String;

inline void statmcCheck(int rc, const char *what) {
    if (rc != STATMC_OK) throw std::runtime_error(std::string(what) + ": " + statmc_last_error());  // cv::Exception's role
}

// ---- cv::Vec<T, N>: the element-wise value type (estimator.h:127-145 adds `*` and `/ uint64` itself)
template <typename T, int N>
class Vec {
  public:
    Vec() { for (int i = 0; i < N; i++) val[i] = T(0); }
    Vec(T v0) { val[0] = v0; for (int i = 1; i < N; i++) val[i] = T(0); }   // cv semantics: the rest is zero
    Vec(T v0, T v1, T v2) { static_assert(N == 3, ""); val[0] = v0; val[1] = v1; val[2] = v2; }
    explicit Vec(const T *p) { for (int i = 0; i < N; i++) val[i] = p[i]; }
    const T &operator[](int i) const { return val[i]; }
    T &operator[](int i) { return val[i]; }
    T val[N];
};
template <typename T, int N> inline Vec<T, N> operator+(const Vec<T, N> &a, const Vec<T, N> &b) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = a[i] + b[i]; return r; }
template <typename T, int N> inline Vec<T, N> operator-(const Vec<T, N> &a, const Vec<T, N> &b) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = a[i] - b[i]; return r; }
template <typename T, int N> inline Vec<T, N> operator-(const Vec<T, N> &a) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = -a[i]; return r; }
template <typename T, int N> inline Vec<T, N> &operator+=(Vec<T, N> &a, const Vec<T, N> &b) { for (int i = 0; i < N; i++) a[i] += b[i]; return a; }
template <typename T, int N> inline Vec<T, N> &operator-=(Vec<T, N> &a, const Vec<T, N> &b) { for (int i = 0; i < N; i++) a[i] -= b[i]; return a; }
template <typename T, int N> inline Vec<T, N> operator*(const Vec<T, N> &a, T s) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = a[i] * s; return r; }
template <typename T, int N> inline Vec<T, N> operator*(T s, const Vec<T, N> &a) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = a[i] * s; return r; }
template <typename T, int N> inline Vec<T, N> operator/(const Vec<T, N> &a, T s) { Vec<T, N> r; for (int i = 0; i < N; i++) r[i] = a[i] / s; return r; }
typedef Vec<float, 3> Vec3f;

template <typename T> struct DataType;
template <> struct DataType<float> { enum { type = CV_32FC1 }; };
template <> struct DataType<int> { enum { type = CV_32SC1 }; };
template <> struct DataType<uchar> { enum { type = CV_8UC1 }; };
template <> struct DataType<Vec<float, 3>> { enum { type = CV_32FC3 }; };

inline size_t elemSizeOf(int type) { return (size_t)CV_MAT_CN(type) * (CV_MAT_DEPTH(type) == CV_8U ? 1 : 4); }

// ---- cv::Mat: refcounted, row-major, interleaved, tightly packed host image.  Page-locked when the HIP runtime
// can provide it, so that Buffer::upload / download (buffer.h:57-63) run at the PCIe rate and stay asynchronous.
class Mat {
  public:
    Mat() {}
    Mat(int rows, int cols, int type) { create(rows, cols, type); }
    explicit Mat(const std::vector<float> &v) {   // Mat gBufferDRFactorsMat(gBufferDRFactors), estimator.cpp:287
        create((int)v.size(), 1, CV_32FC1);
        if (!v.empty()) std::memcpy(data_.get(), v.data(), v.size() * sizeof(float));
    }
    void create(int r, int c, int t) {
        if (data_ && r == rows && c == cols && t == type_) return;
        rows = r; cols = c; type_ = t;
        const size_t n = (size_t)r * c * elemSizeOf(t);
        void *p = nullptr;
        if (n && statmc_malloc_host(&p, n) == STATMC_OK && p) {
            data_ = std::shared_ptr<void>(p, [](void *q) { statmc_free_host(q); });
        } else {
            p = std::malloc(n ? n : 1);
            if (!p) throw std::bad_alloc();
            data_ = std::shared_ptr<void>(p, [](void *q) { std::free(q); });
        }
        std::memset(p, 0, n);
    }
    int type() const { return type_; }
    int depth() const { return CV_MAT_DEPTH(type_); }
    int channels() const { return CV_MAT_CN(type_); }
    size_t elemSize() const { return elemSizeOf(type_); }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return !data_ || rows == 0 || cols == 0; }
    uchar *ptr(int row = 0) { return static_cast<uchar *>(data_.get()) + (size_t)row * cols * elemSize(); }
    const uchar *ptr(int row = 0) const { return static_cast<const uchar *>(data_.get()) + (size_t)row * cols * elemSize(); }
    template <typename T> T *ptr(int row = 0) { return reinterpret_cast<T *>(ptr(row)); }
    template <typename T> const T *ptr(int row = 0) const { return reinterpret_cast<const T *>(ptr(row)); }
    // depth conversion with unchanged channel count (int32 <-> float32 is all the path needs: buffer.h:51-54,
    // statpath.cpp:450); converting a Mat into itself is allowed
    void convertTo(Mat &dst, int rtype) const {
        const int ddepth = CV_MAT_DEPTH(rtype), cn = channels();
        Mat out = (&dst == this) ? Mat() : dst;
        out.create(rows, cols, CV_MAKETYPE(ddepth, cn));
        const size_t n = total() * cn;
        if (depth() == ddepth) std::memcpy(out.ptr(), ptr(), n * (ddepth == CV_8U ? 1 : 4));
        else if (depth() == CV_32S && ddepth == CV_32F) { const int *s = ptr<int>(); float *d = out.ptr<float>(); for (size_t i = 0; i < n; i++) d[i] = (float)s[i]; }
        else if (depth() == CV_32F && ddepth == CV_32S) { const float *s = ptr<float>(); int *d = out.ptr<int>(); for (size_t i = 0; i < n; i++) d[i] = (int)std::lrintf(s[i]); }
        else throw std::runtime_error("statmc_cv: convertTo supports int32 <-> float32 only");
        dst = out;
    }
    // OpenCV takes the destination as an OutputArray, which binds to a const Mat as well (a Mat is a handle to shared
    // pixels): buffer.cpp:34-38 converts into the outMat of a `const Buffer &`
    void convertTo(const Mat &dst, int rtype) const { convertTo(const_cast<Mat &>(dst), rtype); }
    int rows = 0, cols = 0;

  private:
    int type_ = CV_8UC1;
    std::shared_ptr<void> data_;
};

template <typename T>
class Mat_ : public Mat {
  public:
    Mat_() {}
    Mat_(int rows, int cols) : Mat(rows, cols, DataType<T>::type) {}
    Mat_(const Mat &m) : Mat(m) {}
};
typedef Mat_<float> Mat1f;
typedef Mat_<int> Mat1i;
typedef Mat_<Vec3f> Mat3f;

namespace cuda {

// Upload / filter / download as a pipeline of row bands (statmc_bands.hpp): GpuMat::upload of an image only notes the
// copy; filter<T> then issues the copies band by band on a copy stream and filters each band as soon as it has landed;
// GpuMat::download of an image that filter wrote follows band by band on a second copy stream.  The reference's
// Estimator::Upload / Denoise / Download / Synchronize (estimator.cpp:409-489) run unchanged and overlap.  Anything
// else that touches the stream flushes the noted copies first.  STATMC_CV_BANDS=1 in the environment switches the
// pipeline off (0 / unset: automatic, n: n bands).
// Contract that differs from OpenCV's pageable cudaMemcpyAsync: the host image is read when its copy is ENQUEUED (at the
// next filter<T> / download / waitForCompletion on the stream), not inside upload() -- the pixels of a Mat handed to
// upload() must stay unchanged until the stream has been synchronised.  The reference's render loop satisfies this
// (statpath.cpp:397-417: Upload .. Synchronize with no host write in between).  The noted copy holds a reference to both
// the host Mat and the device allocation, so destroying or re-creating either side before the flush is safe.
namespace detail {
struct PendingUpload {
    uchar *dst;
    Mat src;          // keeps the host image alive until the copy has been enqueued and completed
    size_t rowBytes;
    int rows;
    std::shared_ptr<void> dstMem;   // ... and the device image: a GpuMat destroyed or re-created before the copy is enqueued
                                    // must not leave a dangling destination
};
struct StreamState : statmc::bands::Streams {
    std::vector<PendingUpload> pending;     // noted by GpuMat::upload, not yet enqueued
    std::vector<Mat> inflight;              // host images of enqueued band copies (released at synchronisation)
    std::set<const uchar *> outputs;        // images the last banded filter call wrote
    int outBands = 0, outHeight = 0;
    statmc::bands::Plan outPlan;            // the bands of that call (the copies out follow them)
    bool downloading = false;
};
inline int requestedBands() {
    static const int v = [] { const char *e = std::getenv("STATMC_CV_BANDS"); return e ? std::atoi(e) : 0; }();
    return v;
}
}  // namespace detail

class Stream {   // one asynchronous queue (estimator.h:326)
  public:
    Stream() : st_(std::make_shared<detail::StreamState>()) {
        void *s = nullptr;
        if (statmc_stream_create(&s) == STATMC_OK) h_ = std::shared_ptr<void>(s, [](void *q) { statmc_stream_destroy(q); });
    }
    void *handle() const { return h_.get(); }
    detail::StreamState &state() const { return *st_; }
    // the copies GpuMat::upload has noted go onto the stream itself
    void flushUploads() {
        for (auto &p : st_->pending) {
            statmcCheck(statmc_upload(p.dst, p.src.ptr(), p.rowBytes * p.rows, h_.get()), "GpuMat::upload");
            st_->inflight.push_back(p.src);
        }
        st_->pending.clear();
    }
    void waitForCompletion() {
        flushUploads();
        statmcCheck(statmc_synchronize(h_.get()), "Stream::waitForCompletion");
        if (st_->downloading) {
            statmcCheck(statmc_synchronize(st_->down), "Stream::waitForCompletion");
            st_->downloading = false;
        }
        st_->inflight.clear();
    }

  private:
    std::shared_ptr<void> h_;
    std::shared_ptr<detail::StreamState> st_;
};

template <typename T>
struct PtrStepSz {   // what a kernel would index: the descriptor one table entry carries (estimator.cpp:35-84)
    T *data;
    size_t step;
    int cols, rows;
};
typedef PtrStepSz<uchar> PtrStepSzb;

// Device image, tightly packed rows (OpenCV pitches its rows; nothing in the tree depends on the pitch).  Small
// byte tables (the PtrStepSzb tables, channel counts, range factors) keep a host shadow of what was uploaded: the
// HIP entry points take descriptors by value, so the denoiser reads the tables from there.
class GpuMat {
  public:
    GpuMat() {}
    GpuMat(int rows, int cols, int type) { create(rows, cols, type); }
    void create(int r, int c, int t) {
        if (mem_ && r == rows && c == cols && t == type_) return;
        rows = r; cols = c; type_ = t;
        step = (size_t)c * elemSizeOf(t);
        void *p = nullptr;
        statmcCheck(statmc_malloc(&p, step * r > 0 ? step * r : 1), "GpuMat");
        mem_ = std::shared_ptr<void>(p, [](void *q) { statmc_free(q); });
        data = static_cast<uchar *>(p);
    }
    int type() const { return type_; }
    int channels() const { return CV_MAT_CN(type_); }
    bool empty() const { return !mem_; }
    void upload(const Mat &m, Stream &s) {
        create(m.rows, m.cols, m.type());
        const size_t n = step * rows;
        if (n <= 65536) {
            shadow_ = std::make_shared<std::vector<uchar>>(m.ptr(), m.ptr() + n);
            if (n) statmcCheck(statmc_upload(data, shadow_->data(), n, s.handle()), "GpuMat::upload");
        } else {   // an image: noted, enqueued by filter<T> band by band (or by whatever touches the stream next)
            shadow_.reset();
            auto &pending = s.state().pending;
            for (auto &p : pending)
                if (p.dst == data) { p.src = m; return; }
            pending.push_back(detail::PendingUpload{data, m, step, rows, mem_});
        }
    }
    void download(Mat &m, Stream &s) const {
        m.create(rows, cols, type_);
        detail::StreamState &st = s.state();
        if (st.outBands > 1 && rows == st.outHeight && st.outputs.count(data)) {   // behind each band's filter
            for (int k = 0; k < st.outBands; k++) {
                const int y0 = st.outPlan.edge(k), y1 = st.outPlan.edge(k + 1);
                statmcCheck(statmc_stream_wait_event(st.down, st.filtered[k]), "GpuMat::download");
                statmcCheck(statmc_download(m.ptr(y0), data + (size_t)y0 * step, (size_t)(y1 - y0) * step, st.down), "GpuMat::download");
            }
            st.downloading = true;
            return;
        }
        s.flushUploads();
        statmcCheck(statmc_download(m.ptr(), data, step * rows, s.handle()), "GpuMat::download");
    }
    template <typename T> operator PtrStepSz<T>() const { return PtrStepSz<T>{reinterpret_cast<T *>(data), step, cols, rows}; }
    const uchar *shadow() const { return shadow_ ? shadow_->data() : nullptr; }
    statmc_image desc() const { return statmc_image{data, step, cols, rows}; }
    int rows = 0, cols = 0;
    size_t step = 0;
    uchar *data = nullptr;

  private:
    int type_ = CV_8UC1;
    std::shared_ptr<void> mem_;
    std::shared_ptr<std::vector<uchar>> shadow_;
};

namespace stat_denoiser {

inline void setup() { statmcCheck(statmc_setup(0), "stat_denoiser::setup"); }
inline void synchronize(Stream &s) { s.waitForCompletion(); }

namespace detail {
// a table of n PtrStepSzb entries (uploaded by the caller, read back from the shadow) -> descriptors
inline std::vector<statmc_image> table(const GpuMat &t, size_t n, int channels) {
    std::vector<statmc_image> v(n);
    if (n == 0) return v;
    const PtrStepSzb *e = reinterpret_cast<const PtrStepSzb *>(t.shadow());
    if (!e || (size_t)t.cols * t.rows < n) throw std::runtime_error("statmc_cv: pointer table was not uploaded through GpuMat::upload");
    (void)channels;   // GpuMat -> PtrStepSz<uchar> keeps `cols` in pixels of the source image; only `data` is retyped
    for (size_t i = 0; i < n; i++) v[i] = statmc_image{e[i].data, e[i].step, e[i].cols, e[i].rows};
    return v;
}
}  // namespace detail

// The argument list of the reference's call sites, position by position (estimator.cpp:437-459, 465-487).
// T = float or any 12-byte struct of three floats (the reference passes its own `struct float3`, estimator.cpp:8-10).
template <typename T>
void filter(uchar nBuffers, ushort width, ushort height, float filterDSFactor, uchar filterRadius, bool denoiseFilm,
            const GpuMat &nPtrs, const GpuMat &meanPtrs, const GpuMat &m2Ptrs, const GpuMat &m3Ptrs, const GpuMat &filmPtrs,
            const GpuMat &filmBuffer, const GpuMat &gBufferPtrs, const GpuMat &gBufferChannelCounts,
            const GpuMat &gBufferDRFactors, size_t nGBuffers, const GpuMat &meanCorrPtrs, const GpuMat &discriminatorPtrs,
            const GpuMat &filmFilteredPtrs, const GpuMat &filmFilteredBuffer, Stream &stream) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 12, "filter<T>: T is float or three floats");
    constexpr int C = sizeof(T) == 12 ? 3 : 1;
    const auto n = detail::table(nPtrs, nBuffers, 1), mean = detail::table(meanPtrs, nBuffers, C), m2 = detail::table(m2Ptrs, nBuffers, C),
               m3 = detail::table(m3Ptrs, nBuffers, C), film = detail::table(filmPtrs, nBuffers, C),
               mc = detail::table(meanCorrPtrs, nBuffers, C), dc = detail::table(discriminatorPtrs, nBuffers, C),
               ff = detail::table(filmFilteredPtrs, nBuffers, C);
    std::vector<uint8_t> gch(nGBuffers);
    std::vector<float> gdr(nGBuffers);
    std::vector<statmc_image> g(nGBuffers);
    if (nGBuffers) {
        const uchar *cc = gBufferChannelCounts.shadow();
        const float *dr = reinterpret_cast<const float *>(gBufferDRFactors.shadow());
        if (!cc || !dr) throw std::runtime_error("statmc_cv: G-buffer tables were not uploaded through GpuMat::upload");
        for (size_t i = 0; i < nGBuffers; i++) { gch[i] = cc[i]; gdr[i] = dr[i]; }
        const PtrStepSzb *e = reinterpret_cast<const PtrStepSzb *>(gBufferPtrs.shadow());
        if (!e) throw std::runtime_error("statmc_cv: G-buffer pointer table was not uploaded through GpuMat::upload");
        for (size_t i = 0; i < nGBuffers; i++) g[i] = statmc_image{e[i].data, e[i].step, e[i].cols, e[i].rows};
    }
    statmc_filter_args a;
    std::memset(&a, 0, sizeof(a));
    a.n_buffers = nBuffers;
    a.width = width;
    a.height = height;
    a.filter_ds_factor = filterDSFactor;
    a.filter_radius = filterRadius;
    a.denoise_film = denoiseFilm ? 1 : 0;
    a.n = n.data(); a.mean = mean.data(); a.m2 = m2.data(); a.m3 = m3.data(); a.film = film.data();
    a.film_buffer = filmBuffer.desc();
    a.g_buffers = g.data();
    a.g_channel_counts = gch.data();
    a.g_dr_factors = gdr.data();
    a.n_g_buffers = nGBuffers;
    a.mean_corr = mc.data(); a.discriminator = dc.data(); a.film_filtered = ff.data();
    a.film_filtered_buffer = filmFilteredBuffer.desc();
    a.stream = stream.handle();
    cuda::detail::StreamState &st = stream.state();
    for (const auto *tab : {&mc, &dc, &ff})
        for (const auto &im : *tab) st.outputs.erase(static_cast<const uchar *>(im.data));
    st.outputs.erase(filmFilteredBuffer.data);
    if (st.downloading) {   // band copies out of an earlier download still read the images this call rewrites
        statmcCheck(statmc_event_record(st.join, st.down), "stat_denoiser::filter");
        statmcCheck(statmc_stream_wait_event(stream.handle(), st.join), "stat_denoiser::filter");
    }
    namespace B = statmc::bands;
    const B::Plan plan = B::plan(width, height, filterRadius, cuda::detail::requestedBands());
    const int nb = plan.count();
    bool banded = nb > 1 && !st.pending.empty();
    for (const auto &p : st.pending) banded = banded && p.rows == height;
    if (!banded) {
        stream.flushUploads();
        statmcCheck(C == 3 ? statmc_filter_f32x3(&a) : statmc_filter_f32(&a), "stat_denoiser::filter");
        return;
    }
    // the noted uploads travel band by band on the copy stream (a transfer = a band + the r rows below it) ...
    st.ensure(nb);
    st.beginUploads(stream.handle());   // earlier work may still read the images
    std::vector<size_t> rowBytes;
    for (const auto &p : st.pending) rowBytes.push_back(p.rowBytes);
    // One copy queue by default (4.38 ms for the 1080p bracket in 48 of 48 processes).  With two
    // (STATMC_CV_UPLOAD_QUEUES=2) it is 4.0 ms -- when nothing stalls.  Two stalls were found (round 3,
    // tools/experiments/diagnose_queues.py, iter_times.py):
    //  (a) 7.0 ms for the life of one process in three: the runtime multiplexes streams over GPU_MAX_HW_QUEUES = 4
    //      hardware queues per priority level, the pipeline's streams were one too many, and a copy stream that shared
    //      the kernel stream's hardware queue parked its event-wait barrier packets in front of the kernels (16 of 16
    //      processes with GPU_MAX_HW_QUEUES=2).  Gone since the copy streams live in other priority classes
    //      (statmc_bands.hpp) -- also with GPU_MAX_HW_QUEUES=2.
    //  (b) what is left: with two copy streams feeding four or more bands the HOST thread now and then blocks 6.6 - 8 ms
    //      inside one hipMemcpyAsync enqueue (5.8 ms brackets in 1 - 5 processes of 16 here; single 7 - 10 ms iterations in
    //      the Estimator); the copies themselves run at the same rate in slow and fast processes, three bands never
    //      showed it (0 of 32), HSA_ENABLE_INTERRUPT=0 nearly removes it (1 of 96 iterations against 8 of 96): a
    //      wake-up path inside the runtime, not something this side of the API can order differently.
    static const int nQueues = [] { const char *e = std::getenv("STATMC_CV_UPLOAD_QUEUES"); return e && std::atoi(e) >= 2 ? 2 : 1; }();
    const std::vector<int> queue = B::Streams::deal(rowBytes, nQueues);
    for (int k = 0; k < nb; k++) {
        const int y0 = plan.arrival(k), y1 = plan.arrival(k + 1);
        st.beginTransfer(k);
        for (size_t i = 0; i < st.pending.size(); i++) {
            const auto &p = st.pending[i];
            st.upload(queue[i], p.dst + (size_t)y0 * p.rowBytes, p.src.ptr() + (size_t)y0 * p.rowBytes, (size_t)(y1 - y0) * p.rowBytes);
        }
        st.markArrived(k);
    }
    for (auto &p : st.pending) st.inflight.push_back(p.src);
    st.pending.clear();
    // ... and every band is pre-passed and filtered as soon as its transfer has landed
    for (int k = 0; k < nb; k++) {
        st.waitArrived(stream.handle(), k);
        B::prepassRows(a, C, plan.arrival(k), plan.arrival(k + 1));
        B::filterRows(a, C, plan.edge(k), plan.edge(k + 1));
        statmcCheck(statmc_event_record(st.filtered[k], stream.handle()), "stat_denoiser::filter");
    }
    st.outputs.clear();
    for (const auto *tab : {&mc, &dc, &ff})
        for (const auto &im : *tab) st.outputs.insert(static_cast<const uchar *>(im.data));
    st.outputs.insert(filmFilteredBuffer.data);
    st.outBands = nb;
    st.outPlan = plan;
    st.outHeight = height;
}

// The dormant GPU form of Estimator::CalculateMeanVars (the call at estimator.cpp:501-521 is commented out in favour of
// the CPU loop that follows it): film_var = film_m2 / ((n - 1) n), n read per pixel.
template <typename T>
void calculateMeanVars(uchar nBuffers, ushort width, ushort height, const GpuMat &nPtrs, const GpuMat &filmM2Ptrs,
                       const GpuMat &filmVarPtrs, Stream &stream) {
    constexpr int C = sizeof(T) == 12 ? 3 : 1;
    stream.flushUploads();
    const auto n = detail::table(nPtrs, nBuffers, 1), m2 = detail::table(filmM2Ptrs, nBuffers, C), var = detail::table(filmVarPtrs, nBuffers, C);
    statmcCheck(statmc_calculate_mean_vars(nBuffers, width, height, C, n.data(), m2.data(), var.data(), 0, stream.handle()),
                "stat_denoiser::calculateMeanVars");
}

}  // namespace stat_denoiser
}  // namespace cuda

// ---- file I/O of the dumps (buffer.cpp:40-53, statpath.cpp:449-454, 479-481): PFM, 32-bit float, 1 or 3 channels.
// OpenCV hands images around in BGR order; the file holds RGB.
enum { IMREAD_UNCHANGED = -1 };
enum { COLOR_RGB2BGR = 4, COLOR_BGR2RGB = 4 };
inline void cvtColor(const Mat &src, Mat &dst, int) {   // swaps channels 0 and 2; in place allowed
    if (src.channels() != 3 || src.depth() != CV_32F) throw std::runtime_error("statmc_cv: cvtColor wants float x 3");
    Mat out(src.rows, src.cols, src.type());
    const float *s = src.ptr<float>();
    float *d = out.ptr<float>();
    for (size_t i = 0; i < src.total(); i++) { d[3 * i] = s[3 * i + 2]; d[3 * i + 1] = s[3 * i + 1]; d[3 * i + 2] = s[3 * i]; }
    dst = out;
}
inline bool imwrite(const String &filename, const Mat &bgr) {
    if (bgr.depth() != CV_32F) throw std::runtime_error("statmc_cv: imwrite writes 32-bit float PFM only");
    if (bgr.channels() == 3) {
        Mat rgb;
        cvtColor(bgr, rgb, COLOR_BGR2RGB);
        statmc::writePfm(filename, bgr.cols, bgr.rows, 3, rgb.ptr<float>());
    } else {
        statmc::writePfm(filename, bgr.cols, bgr.rows, 1, bgr.ptr<float>());
    }
    return true;
}
inline Mat imread(const String &filename, int) {
    const statmc::PfmImage im = statmc::readPfm(filename);
    Mat m(im.height, im.width, CV_MAKETYPE(CV_32F, im.channels));
    std::memcpy(m.ptr(), im.data.data(), im.data.size() * sizeof(float));
    if (im.channels == 3) cvtColor(m, m, COLOR_RGB2BGR);
    return m;
}
inline void merge(const std::vector<Mat> &mv, Mat &dst) {   // single-channel float images -> one interleaved image
    if (mv.empty()) { dst = Mat(); return; }
    const int cn = (int)mv.size();
    Mat out(mv[0].rows, mv[0].cols, CV_MAKETYPE(CV_32F, cn));
    for (int c = 0; c < cn; c++) {
        if (mv[c].channels() != 1 || mv[c].depth() != CV_32F || mv[c].rows != out.rows || mv[c].cols != out.cols)
            throw std::runtime_error("statmc_cv: merge wants equal single-channel float images");
        const float *s = mv[c].ptr<float>();
        float *d = out.ptr<float>();
        for (size_t i = 0; i < out.total(); i++) d[i * cn + c] = s[i];
    }
    dst = out;
}
inline void glob(const String &pattern, std::vector<String> &result, bool = false) {
    result.clear();
    glob_t g;
    if (::glob(pattern.c_str(), 0, nullptr, &g) == 0)
        for (size_t i = 0; i < g.gl_pathc; i++) result.emplace_back(g.gl_pathv[i]);
    globfree(&g);
}

}  // namespace cv

#endif  // STATMC_CV_HPP
