// The OpenCV-shaped adaptor (include/statmc_cv.hpp) used the way StatMC's Estimator uses OpenCV: refcounted host
// images, device images, PtrStepSzb tables built on the host and uploaded, stat_denoiser::filter<float3> with the
// argument list of src/statistics/estimator.cpp:465-487, PFM round trip in BGR order.  Prints the result so that the
// Python test can compare it with the library called directly.
//   test_cv_adaptor <stem> <spp> <out.pfm>      (dump files as written by tests: film, t0-b0-{n,mean,m2,m3}, t1/t2-b0-film-mean)
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <iostream>

#include "statmc_cv.hpp"

struct float3 { float x, y, z; };   // the reference's own local type (estimator.cpp:8-10)

using cv::Mat;
using cv::cuda::GpuMat;
using cv::cuda::PtrStepSzb;
using Vec3 = cv::Vec<float, 3>;
using Mat3 = cv::Mat_<Vec3>;

struct Buffer {   // the shape of src/statistics/buffer.h:19-71
    Buffer(const std::string &name, Mat mat) : name(name), mat(mat), gpuMat(mat.rows, mat.cols, mat.type()) {}
    void upload(cv::cuda::Stream &s) { gpuMat.upload(mat, s); }
    void download(cv::cuda::Stream &s) { gpuMat.download(mat, s); }
    std::string name;
    Mat mat;
    GpuMat gpuMat;
};

static Mat readDump(const std::string &path, int type) {   // StatPathIntegrator::ReadFile, statpath.cpp:449-454
    Mat m(1, 1, type);
    cv::imread(path, cv::IMREAD_UNCHANGED).convertTo(m, type);
    if (m.channels() == 3) cv::cvtColor(m, m, cv::COLOR_BGR2RGB);
    return m;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const std::string prefix = std::string(argv[1]) + "-" + argv[2] + "-";
    try {
        // value type: the operators estimator.h relies on
        Vec3 a(1.f, 2.f, 3.f), b = 0.5f;
        if (b[0] != 0.5f || b[1] != 0.f) return 3;
        Vec3 c = a - b;
        c += a * 2.f;
        c = -c / 2.f;
        if (c[0] != -1.25f || c[2] != -4.5f) return 4;
        const float rgb[3] = {4.f, 5.f, 6.f};
        if (Vec3(rgb)[1] != 5.f) return 5;

        cv::cuda::stat_denoiser::setup();
        cv::cuda::Stream stream;
        Buffer film("film", readDump(prefix + "film.pfm", CV_32FC3));
        const int h = film.mat.rows, w = film.mat.cols;
        Buffer n("n", readDump(prefix + "t0-b0-n.pfm", CV_32SC1)), mean("mean", readDump(prefix + "t0-b0-mean.pfm", CV_32FC3)),
            m2("m2", readDump(prefix + "t0-b0-m2.pfm", CV_32FC3)), m3("m3", readDump(prefix + "t0-b0-m3.pfm", CV_32FC3)),
            normal("normal", readDump(prefix + "t1-b0-film-mean.pfm", CV_32FC3)),
            albedo("albedo", readDump(prefix + "t2-b0-film-mean.pfm", CV_32FC3));
        Buffer meanCorr("mean-corr", Mat3(h, w)), disc("discriminator", Mat3(h, w)), filmMeanF("film-mean-f", Mat3(h, w)),
            filmF("film-f", Mat3(h, w)), filmMean("film-mean", Mat3(h, w));
        if (n.mat.depth() != CV_32S || n.mat.ptr<int>(h / 2)[w / 2] < 1) return 6;
        for (Buffer *bptr : {&film, &n, &mean, &m2, &m3, &normal, &albedo}) bptr->upload(stream);

        // pointer tables: one entry each, built like PREPARE_STAT_BUFFER_GPU_PTRS (estimator.cpp:35-69)
        auto table = [&](Buffer &buf) {
            Mat cpu(1, 1, CV_8UC(sizeof(PtrStepSzb)));
            PtrStepSzb *p = cpu.ptr<PtrStepSzb>();
            *p = buf.gpuMat;
            GpuMat g;
            g.upload(cpu, stream);
            return g;
        };
        GpuMat nP = table(n), meanP = table(mean), m2P = table(m2), m3P = table(m3), filmP = table(filmMean),
               mcP = table(meanCorr), dcP = table(disc), ffP = table(filmMeanF);
        // G-buffers: PREPARE_G_BUFFER_GPU_PTRS (estimator.cpp:72-84) and the range factors (estimator.cpp:16,287-288)
        std::vector<Buffer *> gBuffers = {&normal, &albedo};
        Mat ptrsCPU(1, 2, CV_8UC(sizeof(PtrStepSzb))), countsCPU(1, 2, CV_8UC1);
        for (int i = 0; i < 2; i++) {
            ptrsCPU.ptr<PtrStepSzb>()[i] = gBuffers[i]->gpuMat;
            countsCPU.ptr<unsigned char>()[i] = (unsigned char)gBuffers[i]->gpuMat.channels();
        }
        GpuMat gPtrs, gCounts, gDR;
        gPtrs.upload(ptrsCPU, stream);
        gCounts.upload(countsCPU, stream);
        const std::vector<float> drFactors = {-.5f / (0.1f * 0.1f), -.5f / (0.02f * 0.02f)};
        Mat drMat(drFactors);
        gDR.upload(drMat, stream);

        const float sd = 10.f;
        cv::cuda::stat_denoiser::filter<float3>(1, (ushort)w, (ushort)h, -.5f / (sd * sd), 20, true, nP, meanP, m2P, m3P, filmP,
                                                film.gpuMat, gPtrs, gCounts, gDR, gBuffers.size(), mcP, dcP, ffP, filmF.gpuMat, stream);
        filmF.download(stream);
        cv::cuda::stat_denoiser::synchronize(stream);
        // the same bracket again, timed: Upload + Denoise + Download + Synchronize in the reference's call order
        // (statpath.cpp:409-417); with STATMC_CV_BANDS unset the adaptor runs it as a pipeline of row bands
        long long best_ns = -1;
        for (int it = 0; it < 4; it++) {
            const auto t0 = std::chrono::steady_clock::now();
            for (Buffer *bptr : {&film, &n, &mean, &m2, &m3, &normal, &albedo}) bptr->upload(stream);
            cv::cuda::stat_denoiser::filter<float3>(1, (ushort)w, (ushort)h, -.5f / (sd * sd), 20, true, nP, meanP, m2P, m3P, filmP,
                                                    film.gpuMat, gPtrs, gCounts, gDR, gBuffers.size(), mcP, dcP, ffP, filmF.gpuMat, stream);
            filmF.download(stream);
            meanCorr.download(stream);
            cv::cuda::stat_denoiser::synchronize(stream);
            const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            if (best_ns < 0 || ns < best_ns) best_ns = ns;
        }
        std::printf("bracket_ns %lld bands %d\n", best_ns, stream.state().outBands);
        if (std::getenv("STATMC_CV_DIAG")) {
            // diagnosis of slow processes (tools/experiments/diagnose_queues.py): the raw copies of the bracket on the
            // pipeline's own streams, without kernels -- one queue, two queues, the download
            auto &st = stream.state();
            st.ensure(1);
            std::vector<Buffer *> ups = {&film, &n, &mean, &m2, &m3, &normal, &albedo};
            auto best_of = [&](auto &&fn) {
                long long best = -1;
                for (int it = 0; it < 5; it++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    fn();
                    const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                    if (best < 0 || ns < best) best = ns;
                }
                return best;
            };
            auto upload = [&](int queues) {
                int i = 0;
                for (Buffer *b : ups) {
                    void *q = (queues == 2 && (i++ & 1)) ? st.up2 : st.up;
                    statmc_upload(b->gpuMat.data, b->mat.ptr(), b->gpuMat.step * b->gpuMat.rows, q);
                }
                statmc_synchronize(st.up);
                statmc_synchronize(st.up2);
            };
            const long long u1 = best_of([&] { upload(1); }), u2 = best_of([&] { upload(2); });
            const long long dn = best_of([&] {
                statmc_download(filmF.mat.ptr(), filmF.gpuMat.data, filmF.gpuMat.step * filmF.gpuMat.rows, st.down);
                statmc_synchronize(st.down);
            });
            // uploads on two queues while the download runs on the third
            const long long both = best_of([&] {
                statmc_download(filmF.mat.ptr(), filmF.gpuMat.data, filmF.gpuMat.step * filmF.gpuMat.rows, st.down);
                upload(2);
                statmc_synchronize(st.down);
            });
            std::printf("diag upload_1q_ns %lld upload_2q_ns %lld download_ns %lld upload_2q_plus_download_ns %lld\n", u1, u2, dn, both);
        }
        if (argc > 4) {   // the corrected means of the last iteration, for a bit-for-bit comparison between band counts
            Mat mcOut;
            cv::cvtColor(meanCorr.mat, mcOut, cv::COLOR_RGB2BGR);
            cv::imwrite(argv[4], mcOut);
        }

        Mat out;   // OutputBufferSelection::Write, buffer.cpp:40-53
        cv::cvtColor(filmF.mat, out, cv::COLOR_RGB2BGR);
        cv::imwrite(argv[3], out);
        std::vector<cv::String> found;
        cv::glob(prefix + "*.pfm", found, false);
        std::printf("ok %dx%d dumps %zu\n", w, h, found.size());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "test_cv_adaptor: %s\n", e.what());
        return 1;
    }
}
