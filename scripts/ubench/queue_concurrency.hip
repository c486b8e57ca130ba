// queue_concurrency.hip -- how many HIP streams does the device really run side by side, and what happens at 5 and 6?
// (round-5 verdict item 1: bench.py with 5 / 6 frames in flight is 15 % / 8 % slower than with 4 or 8, unexplained.)
//
// Every stream gets a chain of K dependent kernels.  A kernel is ONE workgroup of 64 threads that spins on the
// 100-MHz s_memrealtime clock for `spin_us` microseconds: it occupies no resource that another chain could want, so
// if S chains of K kernels take as long as one chain, S streams run concurrently; if they take S times as long they
// are serialised.  Each kernel also writes its (start, end) stamps and the XCC it ran on, so the per-stream
// timelines can be laid side by side: which streams NEVER overlap tells which hardware queues share a pipe.
//   form A  "spin":   S streams x K kernels x spin_us                    -> wall / (K * spin_us)  = serialisation factor
//   form B  "frame":  per stream a repeating pattern long (80 us) + 20 short (3 us) kernels -- the shape of a frame
//                     (one blend, many small dependent launches) -> frames/s equivalent per stream count
// The number of hardware queues is GPU_MAX_HW_QUEUES (read by the HIP runtime at initialisation): run the binary once
// per value.   hipcc --offload-arch=gfx950 -O3 scripts/ubench/queue_concurrency.hip -o queue_concurrency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                                \
        }                                                                \
    } while (0)

struct Stamp {
    unsigned long long t0, t1;
    uint32_t xcc, pad;
};

__global__ void k_spin(Stamp* out, uint32_t ticks) {
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        asm volatile("s_sleep 4\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    } while (t - t0 < ticks);
    if (threadIdx.x == 0) {
        uint32_t x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out->t0 = t0;
        out->t1 = t;
        out->xcc = x & 15u;
    }
}

static double run(int S, int K, const std::vector<uint32_t>& pattern_us, std::vector<hipStream_t>& streams, Stamp* d_st,
                  std::vector<Stamp>* host_out) {
    const int P = (int)pattern_us.size();
    CK(hipDeviceSynchronize());
    const auto a = std::chrono::steady_clock::now();
    for (int k = 0; k < K; ++k)
        for (int s = 0; s < S; ++s)
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, streams[s], d_st + (size_t)s * K + k, pattern_us[k % P] * 100u);
    CK(hipDeviceSynchronize());
    const auto b = std::chrono::steady_clock::now();
    if (host_out) {
        host_out->resize((size_t)S * K);
        CK(hipMemcpy(host_out->data(), d_st, sizeof(Stamp) * (size_t)S * K, hipMemcpyDeviceToHost));
    }
    return std::chrono::duration<double, std::micro>(b - a).count();
}

int main(int argc, char** argv) {
    const int maxS = argc > 1 ? std::atoi(argv[1]) : 12;
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    std::vector<hipStream_t> streams(maxS);
    for (auto& s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Stamp* d_st;
    const int KMAX = 4096;
    CK(hipMalloc(&d_st, sizeof(Stamp) * (size_t)maxS * KMAX));
    const std::vector<uint32_t> spin = {20u};
    std::vector<uint32_t> frame = {80u};
    for (int i = 0; i < 20; ++i) frame.push_back(3u);
    // warm up every stream
    run(maxS, 8, spin, streams, d_st, nullptr);
    for (int S = 1; S <= maxS; ++S) {
        // A: equal kernels
        const int K = 200;
        double best = 1e30;
        std::vector<Stamp> st;
        for (int rep = 0; rep < 3; ++rep) best = std::min(best, run(S, K, spin, streams, d_st, rep == 2 ? &st : nullptr));
        const double one_chain = K * 20.0;
        // pairwise overlap of the streams' busy intervals (share of stream i's busy time during which stream j is busy too)
        std::vector<double> never;
        int never_pairs = 0;
        for (int i = 0; i < S; ++i)
            for (int j = i + 1; j < S; ++j) {
                double ov = 0.0;
                int bj = 0;
                for (int a = 0; a < K; ++a) {
                    const Stamp& x = st[(size_t)i * K + a];
                    while (bj < K && st[(size_t)j * K + bj].t1 <= x.t0) ++bj;
                    for (int b = bj; b < K && st[(size_t)j * K + b].t0 < x.t1; ++b) {
                        const Stamp& y = st[(size_t)j * K + b];
                        const double lo = (double)std::max(x.t0, y.t0), hi = (double)std::min(x.t1, y.t1);
                        if (hi > lo) ov += (hi - lo) / 100.0;
                    }
                }
                if (ov < 0.02 * one_chain) ++never_pairs;
            }
        if (std::getenv("QC_TIMELINE") && (S == 4 || S == 5 || S == 6 || S == 8)) {
            // per stream: span and busy time; the overlap matrix (share of stream i's busy time with stream j busy too); the
            // first kernels of every stream as (start, end) in us from the first start: who runs while who waits
            unsigned long long t00 = ~0ull;
            for (auto& x : st) t00 = std::min(t00, x.t0);
            std::printf("{\"timeline_streams\": %d, \"hw_queues\": \"%s\", \"streams\": [", S, q ? q : "default");
            for (int i = 0; i < S; ++i) {
                double busy = 0.0;
                for (int a = 0; a < K; ++a) busy += (double)(st[(size_t)i * K + a].t1 - st[(size_t)i * K + a].t0) / 100.0;
                std::printf("%s{\"first_start_us\": %.1f, \"last_end_us\": %.1f, \"busy_us\": %.0f, \"overlap_with\": [", i ? ", " : "",
                            (double)(st[(size_t)i * K].t0 - t00) / 100.0, (double)(st[(size_t)i * K + K - 1].t1 - t00) / 100.0, busy);
                for (int j = 0; j < S; ++j) {
                    double ov = 0.0;
                    if (j != i)
                        for (int a = 0; a < K; ++a) {
                            const Stamp& x = st[(size_t)i * K + a];
                            for (int b = 0; b < K; ++b) {
                                const Stamp& y = st[(size_t)j * K + b];
                                if (y.t0 >= x.t1) break;
                                if (y.t1 <= x.t0) continue;
                                ov += (double)(std::min(x.t1, y.t1) - std::max(x.t0, y.t0)) / 100.0;
                            }
                        }
                    std::printf("%s%.2f", j ? ", " : "", ov / busy);
                }
                std::printf("], \"first_kernels_us\": [");
                for (int a = 0; a < 12; ++a)
                    std::printf("%s[%.0f, %.0f]", a ? ", " : "", (double)(st[(size_t)i * K + a + 100].t0 - t00) / 100.0,
                                (double)(st[(size_t)i * K + a + 100].t1 - t00) / 100.0);
                std::printf("]}");
            }
            std::printf("]}\n");
        }
        // B: frame-shaped chains
        const int KF = 21 * 30;
        double bestf = 1e30;
        for (int rep = 0; rep < 3; ++rep) bestf = std::min(bestf, run(S, KF, frame, streams, d_st, nullptr));
        const double frames = 30.0 * S;
        std::printf("{\"hw_queues\": \"%s\", \"streams\": %d, \"spin_wall_us\": %.0f, \"serialisation\": %.2f, \"concurrency\": %.2f, "
                    "\"pairs_that_never_overlap\": %d, \"frame_chain_us_per_frame\": %.1f, \"frame_chain_ideal_us\": %.1f}\n",
                    q ? q : "default", S, best, best / one_chain, S * one_chain / best, never_pairs, bestf / frames, 140.0 / S);
        std::fflush(stdout);
    }
    return 0;
}
