// drivers.cpp -- the reference's offline front-ends on top of the C ABI (SURVEY 8f, N2 + N4).
//
// Mirrors src/bin/render.rs:33-128 (render_views: resolution cap, fit_near_far, SplattingArgs of the offline
// callers, Rgba16Float target cleared to TRANSPARENT, PNG per view), :187-246 (download_texture: f16 -> clamp ->
// * 255 -> `as u8`), src/bin/measure.rs:27-154 (fixed 2048x2048 Rgba8Unorm target, 1 + 10 x cameras frames, one
// wait, "average FPS") and src/renderer.rs:548-582 (Display::render).  Everything here only CALLS the hot path
// through the public entry points (ws_renderer_prepare / ws_renderer_render); nothing is re-implemented.
#include <sys/stat.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ws_internal.h"

using namespace ws;

namespace ws {
int launch_display(const void* src, int src_format, size_t src_pitch, uint32_t w, uint32_t h, const float bg[4],
                   int dst_format, void* dst, size_t dst_pitch, hipStream_t stream);
}

namespace {

size_t texel_bytes(ws_color_format f) { return f == WS_FORMAT_RGBA8_UNORM ? 4 : (f == WS_FORMAT_RGBA16_FLOAT ? 8 : 16); }

// SplattingArgs of the offline callers (bin/render.rs:88-103, bin/measure.rs:63-78)
void offline_args(const ws_scene_camera& sc, const ws_pointcloud* pc, uint32_t vw, uint32_t vh, ws_splatting_args* a) {
    std::memset(a, 0, sizeof *a);
    ws_camera_from_scene(sc.position, sc.rotation, sc.fx, sc.fy, sc.width, sc.height, &a->camera);
    ws_aabb box;
    ws_pointcloud_bbox(pc, &box);
    ws_camera_fit_near_far(&a->camera, &box);
    a->viewport[0] = vw;
    a->viewport[1] = vh;
    a->gaussian_scaling = 1.0f;
    a->max_sh_deg = ws_pointcloud_sh_deg(pc);
    a->walltime_secs = 100.0;
    // background_color: TRANSPARENT (all zero)
}

bool make_dir(const std::string& p) {
    struct stat st;
    if (stat(p.c_str(), &st) == 0) return S_ISDIR(st.st_mode);
    return mkdir(p.c_str(), 0777) == 0;
}
bool make_dirs(const std::string& p) {  // create_dir_all
    for (size_t i = 1; i < p.size(); ++i)
        if (p[i] == '/' && !make_dir(p.substr(0, i))) return false;
    return make_dir(p);
}

}  // namespace

extern "C" {

int ws_download_texture_rgba8(ws_context* ctx, const void* d_image, ws_color_format format, uint32_t width,
                              uint32_t height, size_t row_pitch_bytes, uint8_t* out, void* stream_v) {
    if (!ctx || !d_image || !out || width == 0 || height == 0) return fail(WS_ERR_INVALID, "ws_download_texture_rgba8: bad argument");
    const size_t tb = texel_bytes(format);
    if (row_pitch_bytes < tb * width) return fail(WS_ERR_INVALID, "ws_download_texture_rgba8: row pitch too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    std::vector<uint8_t> raw;
    try {
        raw.resize(row_pitch_bytes * height);
    } catch (...) {
        return fail(WS_ERR_OOM, "ws_download_texture_rgba8: host allocation failed");
    }
    WS_HIP(hipMemcpyAsync(raw.data(), d_image, raw.size(), hipMemcpyDeviceToHost, stream));
    WS_HIP(hipStreamSynchronize(stream));
    for (uint32_t y = 0; y < height; ++y) {
        const uint8_t* row = raw.data() + (size_t)y * row_pitch_bytes;
        uint8_t* o = out + (size_t)y * width * 4;
        for (uint32_t i = 0; i < width * 4; ++i) {
            if (format == WS_FORMAT_RGBA8_UNORM) {
                o[i] = row[i];
                continue;
            }
            float v;
            if (format == WS_FORMAT_RGBA16_FLOAT) {
                uint16_t h;
                std::memcpy(&h, row + (size_t)i * 2, 2);
                v = host_f16_to_f32(h);
            } else {
                std::memcpy(&v, row + (size_t)i * 4, 4);
            }
            // f32::clamp(0., 1.) * 255. as u8  (bin/render.rs:232; NaN -> 0 like Rust's saturating cast)
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            const float s = v * 255.0f;
            o[i] = (s != s) ? 0 : (uint8_t)s;
        }
    }
    return WS_OK;
}

int ws_render_views(ws_context* ctx, const ws_pointcloud* pc, const ws_scene* scene, int split, const char* out_dir,
                    uint32_t* rendered) {
    if (!ctx || !pc || !scene || !out_dir) return fail(WS_ERR_INVALID, "ws_render_views: null argument");
    if (split != WS_SPLIT_TRAIN && split != WS_SPLIT_TEST) return fail(WS_ERR_INVALID, "ws_render_views: split must be train or test");
    if (rendered) *rendered = 0;
    const std::string dir = std::string(out_dir) + "/" + (split == WS_SPLIT_TEST ? "test" : "train");
    if (!make_dirs(dir)) return fail(WS_ERR_IO, "ws_render_views: cannot create " + dir);
    const uint32_t n = ws_scene_cameras(scene, split, 0, nullptr);
    std::vector<ws_scene_camera> cams(n);
    ws_scene_cameras(scene, split, n, cams.data());
    ws_renderer* r = nullptr;
    int rc = ws_renderer_create(ctx, WS_FORMAT_RGBA16_FLOAT, ws_pointcloud_sh_deg(pc), ws_pointcloud_compressed(pc), &r);
    if (rc) return rc;
    // bin/render.rs:154 draws into an Rgba16Float target: the fixed-function blender rounds the destination to f16 after
    // every splat.  The target-precision blend mode reproduces that (ws_context_config::render_views_fast_blend: the throughput blend,
    // one rounding at the store).
    if (!ctx->render_views_fast_blend) (void)ws_renderer_set_blend_mode(r, WS_BLEND_TARGET_PRECISION);
    void* target = nullptr;
    size_t target_bytes = 0;
    std::vector<uint8_t> rgba;
    const float clear[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n && rc == WS_OK; ++i) {
        uint32_t w = cams[i].width, h = cams[i].height;
        if (w > 1600) {  // bin/render.rs:58-62
            const float s = (float)w / 1600.0f;
            w = 1600;
            h = (uint32_t)((float)h / s);
        }
        if (w == 0 || h == 0) {
            rc = fail(WS_ERR_INVALID, "ws_render_views: camera with an empty image");
            break;
        }
        const size_t need = (size_t)w * h * 8;
        if (need > target_bytes) {
            if (target) ws_device_free(ctx, target);
            target = nullptr;
            if ((rc = ws_device_malloc(ctx, need, &target))) break;
            target_bytes = need;
        }
        ws_splatting_args a;
        offline_args(cams[i], pc, w, h, &a);
        // Entries are emitted far -> near and truncated at the capacity, so an overflowing frame would silently lose
        // its NEAREST splats: render, look at the frame's error bits, grow the entry list and render again if needed.
        for (int attempt = 0; attempt < 3; ++attempt) {
            if ((rc = ws_renderer_prepare(r, pc, &a, nullptr))) break;
            if ((rc = ws_renderer_render(r, pc, clear, target, (size_t)w * 8, nullptr))) break;
            uint32_t bits = 0, needed = 0;
            if ((rc = ws_renderer_errors(r, &bits, &needed, 1))) break;
            if (bits == 0) break;
            if ((bits & ~1u) != 0 || attempt == 2) {
                rc = fail(WS_ERR_OVERFLOW, "ws_render_views: the frame reported device-side errors (tile-entry overflow or a "
                                           "look-back time-out)");
                break;
            }
            ws_renderer_set_tile_entry_capacity(r, (uint64_t)needed + needed / 4 + 4096);
        }
        if (rc) break;
        rgba.resize((size_t)w * h * 4);
        if ((rc = ws_download_texture_rgba8(ctx, target, WS_FORMAT_RGBA16_FLOAT, w, h, (size_t)w * 8, rgba.data(), nullptr))) break;
        char name[32];
        std::snprintf(name, sizeof name, "/%05u.png", i);
        if ((rc = ws_png_write_rgba8((dir + name).c_str(), w, h, rgba.data(), (size_t)w * 4))) break;
        if (rendered) *rendered = i + 1;
    }
    if (target) ws_device_free(ctx, target);
    ws_renderer_destroy(r);
    return rc;
}

int ws_measure(ws_context* ctx, const ws_pointcloud* pc, const ws_scene* scene, uint32_t num_samples,
               uint32_t frames_in_flight, float* fps) {
    if (!ctx || !pc || !scene || !fps) return fail(WS_ERR_INVALID, "ws_measure: null argument");
    if (num_samples == 0) num_samples = 10;  // bin/measure.rs:98
    if (frames_in_flight == 0) frames_in_flight = 1;
    const uint32_t n = ws_scene_cameras(scene, WS_SPLIT_TRAIN, 0, nullptr);
    if (n == 0) return fail(WS_ERR_INVALID, "ws_measure: the scene has no training cameras");
    std::vector<ws_scene_camera> cams(n);
    ws_scene_cameras(scene, WS_SPLIT_TRAIN, n, cams.data());
    const uint32_t W = 2048, H = 2048;  // bin/measure.rs:34
    std::vector<ws_renderer*> rs(frames_in_flight, nullptr);
    std::vector<void*> targets(frames_in_flight, nullptr);
    std::vector<hipStream_t> streams(frames_in_flight, nullptr);
    int rc = WS_OK;
    for (uint32_t k = 0; k < frames_in_flight && rc == WS_OK; ++k) {
        rc = ws_renderer_create(ctx, WS_FORMAT_RGBA8_UNORM, ws_pointcloud_sh_deg(pc), ws_pointcloud_compressed(pc), &rs[k]);
        if (rc == WS_OK) ws_internal_renderer_set_throughput_mode(rs[k], frames_in_flight > 1);
        if (rc == WS_OK) rc = ws_device_malloc(ctx, (size_t)W * H * 4, &targets[k]);
        if (rc == WS_OK && k > 0 && hipStreamCreateWithFlags(&streams[k], hipStreamNonBlocking) != hipSuccess)
            rc = fail(WS_ERR_HIP, "ws_measure: hipStreamCreate failed");
    }
    const float clear[4] = {0, 0, 0, 0};
    // The measurement, at most three times: a run whose frames overflowed the (automatic) tile-entry capacity is not a rate --
    // every slot's renderer is given 1.25 x the largest demand any slot saw and the WHOLE procedure runs again (a
    // scene heavier than the automatic capacity used to fail here for good, each call starting from fresh renderers).
    for (int attempt = 0; rc == WS_OK; ++attempt) {
        const auto start = std::chrono::steady_clock::now();  // before the warm-up frame (bin/measure.rs:50)
        ws_splatting_args a;
        offline_args(cams[0], pc, W, H, &a);
        rc = ws_renderer_prepare(rs[0], pc, &a, streams[0]);  // "first render to lazy init sorter stuff"
        if (rc == WS_OK) rc = ws_renderer_render(rs[0], pc, clear, targets[0], (size_t)W * 4, streams[0]);
        uint32_t f = 0;
        for (uint32_t i = 0; i < n && rc == WS_OK; ++i) {
            offline_args(cams[i], pc, W, H, &a);
            for (uint32_t s = 0; s < num_samples && rc == WS_OK; ++s, ++f) {
                const uint32_t k = f % frames_in_flight;
                rc = ws_renderer_prepare(rs[k], pc, &a, streams[k]);
                if (rc == WS_OK) rc = ws_renderer_render(rs[k], pc, clear, targets[k], (size_t)W * 4, streams[k]);
            }
        }
        if (rc == WS_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(WS_ERR_HIP, "ws_measure: device sync failed");  // device.poll(Wait)
        const float secs = std::chrono::duration<float>(std::chrono::steady_clock::now() - start).count();
        if (rc == WS_OK) *fps = 1.0f / (secs / ((float)n * (float)num_samples));
        uint32_t all_bits = 0, demand = 0;
        for (uint32_t k = 0; k < frames_in_flight && rc == WS_OK; ++k) {  // a rate over frames that dropped entries is not a rate
            uint32_t bits = 0;
            rc = ws_renderer_errors(rs[k], &bits, nullptr, 1);
            all_bits |= bits;
            demand = std::max(demand, ws_internal_renderer_demand(rs[k]));
        }
        if (rc != WS_OK || all_bits == 0) break;
        if ((all_bits & ~1u) != 0 || attempt == 2 || demand == 0) {
            rc = fail(WS_ERR_OVERFLOW, "ws_measure: frames reported device-side errors (tile-entry overflow or a look-back time-out)");
            break;
        }
        for (uint32_t k = 0; k < frames_in_flight; ++k)
            ws_renderer_set_tile_entry_capacity(rs[k], (uint64_t)demand + demand / 4 + 4096);
    }
    for (uint32_t k = 0; k < frames_in_flight; ++k) {
        if (rs[k]) ws_renderer_destroy(rs[k]);
        if (targets[k]) ws_device_free(ctx, targets[k]);
        if (streams[k]) (void)hipStreamDestroy(streams[k]);
    }
    return rc;
}

// ---- view batches: several frames in flight ---------------------------------------------------------------------
// A frame is a pure function of (scene, camera); the views of a batch are independent.  Every frame in flight gets its
// own renderer (private scratch), and its own HIP stream; the scene is shared, read-only.  Small latency-bound kernels
// of one frame (the 12 launches of the depth sort) then overlap the wide kernels of another: 1.5x the frames/s of
// strictly one frame at a time on MI355X.
// HIP maps user streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, the null stream included): export
// GPU_MAX_HW_QUEUES=8 before the process makes its first HIP call, or two of four slots may share a queue and
// serialise (4800 instead of 5600 frames/s on c2).
// SUBMISSION THREADS (round 4).  A frame is 22 launches + a memset, 65-90 us of one host thread; a small scene (10 k ... 250 k
// Gaussians at 800x600) needs less GPU time per frame than that, so one thread enqueueing for four slots is the limit:
// 13.7 k frames/s on the 10 k scene, 10.9 k at 250 k (profiles/r04/mt_enqueue_probe.txt).  The slots are independent -- own
// renderer, own stream -- and HIP launches into different streams from different threads in parallel, so every slot gets a
// worker thread that enqueues ITS frames of a call, in order (the order on each stream, and therefore every frame, is what
// the single thread produces): 31.4 k and 18.8 k frames/s on the same scenes; nothing changes where the GPU is the limit
// (1 M Gaussians at 1080p: 7.16 k either way), so the workers are used for point clouds of at most BATCH_THREADS_MAX_POINTS
// Gaussians (WS_BATCH_THREADS=0 / 1 forces them off / on).  They cost host cores while they run (about one per slot).
constexpr uint32_t BATCH_THREADS_MAX_POINTS = 512u * 1024u;
// HOST RUN-AHEAD (round 5).  A caller that enqueues a thousand frames in one call gets them all queued: the host needs 77 us per
// frame where the GPU needs 140, so it runs ahead until the runtime's queues are full and then WAITS INSIDE THE HIP RUNTIME
// for queue space -- spinning: the enqueue thread at 100 % of a core and a runtime helper thread beside it at 93 %
// (profiles/r05/host_threads.txt; 1.9 cores per rank, 8 ranks on a 16-CPU quota).  Nothing is gained by being a thousand frames
// ahead.  Every slot therefore keeps its host side at most `queue_depth` frames (default 5, ws_context_config::batch_queue_depth; 0 =
// unbounded; measured 2 / 3 / 5: -2 / -1.6 ... -3 / 0 ... -1 % frames/s against unbounded, profiles/r05/host_run_ahead_ab.txt)
// ahead of the device: the compositing kernel of every frame posts the frame's number to pinned host memory when
// it STARTS (ws_internal_renderer_progress: one store by one thread, no event, no extra packet), and before a slot's next
// frame is enqueued the host polls that word, sleeping 20 us between looks -- no runtime call, no spinning (a HIP event wait,
// blocking flavour included, spins for ~200 us before it sleeps, i.e. always at these frame times: measured, 0.99 of a core).
// With 4 slots x 5 frames the GPU always has > 2 ms of work queued; the image stream is unchanged.
struct SlotWindow {
    uint32_t waits = 0;      // times the host had to wait for this slot (ws_view_batch_host_waits)
    bool stalled = false;    // the slot's progress word did not move for 10 s: windowing is off for it, the batch reports WS_ERR_STATE
};
struct ws_view_batch {
    ws_context* ctx = nullptr;
    std::vector<ws_renderer*> renderers;
    std::vector<hipStream_t> streams;
    std::vector<SlotWindow> windows;
    uint32_t queue_depth = 5;
    bool threads_disabled = false;  // a host that could not start the submission threads: this batch enqueues from the caller's
    uint64_t next = 0;  // frames enqueued so far: frame i runs on slot i % frames_in_flight

    // one call's work, shared by the workers (valid while `pending` != 0)
    struct Job {
        const ws_pointcloud* pc = nullptr;
        const ws_splatting_args* views = nullptr;
        void* const* targets = nullptr;
        uint32_t num_views = 0;
        size_t pitch = 0;
        const float* background = nullptr;
        uint64_t first = 0;  // value of `next` at the start of the call
    } job;
    struct Worker {
        std::thread th;
        int rc = WS_OK;
        std::string err;
    };
    std::vector<Worker> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    uint64_t generation = 0;  // bumped per call: a worker runs when it sees a new generation
    uint32_t pending = 0;     // workers still busy with the current generation
    bool quit = false;
};

namespace {
// before a slot's next frame is enqueued: sleep until the slot's host side is fewer than queue_depth frames ahead of the device
int slot_window_admit(ws_view_batch* b, size_t slot) {
    if (b->queue_depth == 0) return WS_OK;
    const ws_renderer* r = b->renderers[slot];
    const uint32_t enq = ws_internal_renderer_frames_enqueued(r);
    uint32_t started = 0;
    if (!ws_internal_renderer_progress(r, &started)) return WS_OK;  // (no mailbox: unbounded, as before)
    SlotWindow& win = b->windows[slot];
    if (win.stalled) return WS_OK;  // (reported once, below; later frames are enqueued unbounded, as without a mailbox)
    bool waited = false;
    int spins = 0;
    // (sequence numbers wrap: compare differences) -- and never wait forever: a frame that failed to launch posts nothing
    for (; (int32_t)(enq - started) >= (int32_t)b->queue_depth && spins < 500000; ++spins) {
        struct timespec ts = {0, 20000};
        nanosleep(&ts, nullptr);
        waited = true;
        if (!ws_internal_renderer_progress(r, &started)) break;
    }
    if (waited) ++win.waits;
    if (spins >= 500000) {  // 10 s without progress (a lost launch, a device fault): say so instead of proceeding silently
        win.stalled = true;
        return fail(WS_ERR_STATE, "ws_view_batch_render: a slot's frames made no progress for 10 s (device fault or lost launch); "
                                  "the frames enqueued so far are NOT known to have been drawn");
    }
    return WS_OK;
}

void batch_worker(ws_view_batch* b, size_t slot) {
    (void)hipSetDevice(b->ctx->device);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(b->m);
            b->cv_work.wait(lk, [&] { return b->quit || b->generation != seen; });
            if (b->quit) return;
            seen = b->generation;
        }
        const ws_view_batch::Job& j = b->job;
        const size_t slots = b->renderers.size();
        int rc = WS_OK;
        for (uint32_t i = 0; i < j.num_views && rc == WS_OK; ++i) {
            if ((j.first + i) % slots != slot) continue;
            rc = slot_window_admit(b, slot);
            if (rc == WS_OK) rc = ws_renderer_prepare(b->renderers[slot], j.pc, &j.views[i], b->streams[slot]);
            if (rc == WS_OK) rc = ws_renderer_render(b->renderers[slot], j.pc, j.background, j.targets[i], j.pitch, b->streams[slot]);
        }
        b->workers[slot].rc = rc;
        if (rc) b->workers[slot].err = ws_last_error();  // (the error text is thread-local: hand it to the caller)
        {
            std::lock_guard<std::mutex> lk(b->m);
            if (--b->pending == 0) b->cv_done.notify_one();
        }
    }
}
}  // namespace

int ws_view_batch_create(ws_context* ctx, ws_color_format format, uint32_t sh_deg, int compressed,
                         uint32_t frames_in_flight, ws_view_batch** out) {
    if (!ctx || !out) return fail(WS_ERR_INVALID, "ws_view_batch_create: null argument");
    *out = nullptr;
    if (frames_in_flight == 0 || frames_in_flight > 64) return fail(WS_ERR_INVALID, "ws_view_batch_create: frames_in_flight must be 1..64");
    ws_view_batch* b = new (std::nothrow) ws_view_batch();
    if (!b) return fail(WS_ERR_OOM, "ws_view_batch_create: host allocation failed");
    b->ctx = ctx;
    int rc = WS_OK;
    for (uint32_t k = 0; k < frames_in_flight && rc == WS_OK; ++k) {
        ws_renderer* r = nullptr;
        rc = ws_renderer_create(ctx, format, sh_deg, compressed, &r);
        if (rc) break;
        b->renderers.push_back(r);
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            rc = fail(WS_ERR_HIP, "ws_view_batch_create: hipStreamCreate failed");
            break;
        }
        b->streams.push_back(s);
    }
    if (rc == WS_OK) {
        if (ctx->batch_queue_depth >= 0) b->queue_depth = (uint32_t)ctx->batch_queue_depth;
        if (b->queue_depth > 64u) b->queue_depth = 64u;
        b->windows.resize(b->renderers.size());
        // several frames in flight: the slots' blends keep the image order of their workgroups (ws_api.cpp)
        for (ws_renderer* r : b->renderers) ws_internal_renderer_set_throughput_mode(r, b->renderers.size() > 1);
    }
    if (rc != WS_OK) {
        ws_view_batch_destroy(b);
        return rc;
    }
    *out = b;
    return WS_OK;
}

void ws_view_batch_destroy(ws_view_batch* b) {
    if (!b) return;
    {
        std::lock_guard<std::mutex> lk(b->m);
        b->quit = true;
    }
    b->cv_work.notify_all();
    for (auto& w : b->workers)
        if (w.th.joinable()) w.th.join();
    (void)hipDeviceSynchronize();
    for (ws_renderer* r : b->renderers) ws_renderer_destroy(r);
    for (hipStream_t s : b->streams) (void)hipStreamDestroy(s);
    delete b;
}

uint32_t ws_view_batch_frames_in_flight(const ws_view_batch* b) { return b ? (uint32_t)b->renderers.size() : 0u; }

int ws_view_batch_render(ws_view_batch* b, const ws_pointcloud* pc, const ws_splatting_args* views, uint32_t num_views,
                         void* const* d_targets, size_t row_pitch_bytes, const float background[4]) {
    if (!b || !pc || (!views && num_views) || (!d_targets && num_views)) return fail(WS_ERR_INVALID, "ws_view_batch_render: null argument");
    const size_t slots = b->renderers.size();
    // WS_BATCH_K1=g: consecutive frames in groups of g share ONE K1 launch (the scene is read once per group instead of
    // once per frame); groups are aligned to the slot ring, so with slots = 2 g one group's K1 overlaps the other's blends
    size_t group = (size_t)b->ctx->batch_k1;
    if (group > 1 && slots % group != 0) group = 1;
    // every slot's frames enqueued by its own thread (see ws_view_batch): small scenes, enough frames to be worth a wake-up
    const int want_threads = b->ctx->batch_threads;
    if (group == 1 && slots >= 2 && num_views >= 2 * slots && want_threads != 0 && !b->threads_disabled &&
        (want_threads > 0 || pc->num_points <= BATCH_THREADS_MAX_POINTS)) {
        bool have_workers = !b->workers.empty();
        if (!have_workers) {
            try {  // (no C++ exception crosses the ABI: a host that cannot start threads keeps the one-thread path)
                std::vector<ws_view_batch::Worker> ws_(slots);
                b->workers.swap(ws_);
                for (size_t s2 = 0; s2 < slots; ++s2) b->workers[s2].th = std::thread(batch_worker, b, s2);
                have_workers = true;
            } catch (...) {
                {
                    std::lock_guard<std::mutex> lk(b->m);
                    b->quit = true;
                }
                b->cv_work.notify_all();
                for (auto& w : b->workers)
                    if (w.th.joinable()) w.th.join();
                b->workers.clear();
                b->quit = false;
                b->threads_disabled = true;  // (this batch keeps the one-thread path; the context is shared state: untouched)
            }
        }
        if (have_workers) {
            {
                std::lock_guard<std::mutex> lk(b->m);
                b->job.pc = pc;
                b->job.views = views;
                b->job.targets = d_targets;
                b->job.num_views = num_views;
                b->job.pitch = row_pitch_bytes;
                b->job.background = background;
                b->job.first = b->next;
                for (auto& w : b->workers) w.rc = WS_OK;
                b->pending = (uint32_t)slots;
                ++b->generation;
            }
            b->cv_work.notify_all();
            {
                std::unique_lock<std::mutex> lk(b->m);
                b->cv_done.wait(lk, [&] { return b->pending == 0; });
            }
            b->next += num_views;
            for (auto& w : b->workers)
                if (w.rc) return fail(w.rc, w.err);
            return WS_OK;
        }
    }
    uint32_t i = 0;
    while (i < num_views) {
        const size_t k = (size_t)(b->next % slots);
        if (group > 1 && k % group == 0 && num_views - i >= group) {
            ws_renderer* rs[K1_MAX_VIEWS];
            hipStream_t ss[K1_MAX_VIEWS];
            for (size_t j = 0; j < group; ++j) {
                rs[j] = b->renderers[k + j];
                ss[j] = b->streams[k + j];
            }
            int rc = WS_OK;
            for (size_t j = 0; j < group && rc == WS_OK; ++j) rc = slot_window_admit(b, k + j);
            if (rc) return rc;
            rc = ws_internal_prepare_group(rs, (uint32_t)group, pc, &views[i], ss);
            if (rc == WS_ERR_UNSUPPORTED) {  // (timers, capture, a frame graph ...: every frame its own K1)
                group = 1;
                continue;
            }
            if (rc) return rc;
            for (size_t j = 0; j < group; ++j) {
                rc = ws_renderer_render(rs[j], pc, background, d_targets[i + j], row_pitch_bytes, ss[j]);
                if (rc) return rc;
            }
            i += (uint32_t)group;
            b->next += group;
            continue;
        }
        // a ragged head (the ring is not at a group boundary) or tail, or no grouping: frame by frame
        // a target that an earlier frame of this call still writes must be on the same slot (same stream: ordered)
        int rc = slot_window_admit(b, k);
        if (rc == WS_OK) rc = ws_renderer_prepare(b->renderers[k], pc, &views[i], b->streams[k]);
        if (rc == WS_OK) rc = ws_renderer_render(b->renderers[k], pc, background, d_targets[i], row_pitch_bytes, b->streams[k]);
        if (rc) return rc;
        ++b->next;
        ++i;
    }
    return WS_OK;
}

int ws_view_batch_sync(ws_view_batch* b) {
    if (!b) return fail(WS_ERR_INVALID, "ws_view_batch_sync: null batch");
    for (hipStream_t s : b->streams) WS_HIP(hipStreamSynchronize(s));
    return WS_OK;
}

int ws_view_batch_errors(ws_view_batch* b, uint32_t* bits, int reset) {
    if (!b || !bits) return fail(WS_ERR_INVALID, "ws_view_batch_errors: null argument");
    *bits = 0;
    for (ws_renderer* r : b->renderers) {
        uint32_t one = 0;
        int rc = ws_renderer_errors(r, &one, nullptr, reset);
        if (rc) return rc;
        *bits |= one;
    }
    return WS_OK;
}

uint32_t ws_view_batch_host_waits(const ws_view_batch* b) {
    uint32_t n = 0;
    if (b)
        for (const SlotWindow& w : b->windows) n += w.waits;
    return n;
}

ws_renderer* ws_view_batch_renderer(ws_view_batch* b, uint32_t slot) {
    return (b && slot < b->renderers.size()) ? b->renderers[slot] : nullptr;
}

int ws_display_composite(ws_context* ctx, const void* d_src, ws_color_format src_format, size_t src_pitch_bytes,
                         uint32_t width, uint32_t height, const float background[4], ws_surface_format dst_format,
                         void* d_dst, size_t dst_pitch_bytes, void* stream) {
    if (!ctx || !d_src || !d_dst || width == 0 || height == 0) return fail(WS_ERR_INVALID, "ws_display_composite: bad argument");
    if (src_pitch_bytes < texel_bytes(src_format) * width || dst_pitch_bytes < (size_t)width * 4 ||
        (src_pitch_bytes % texel_bytes(src_format)) != 0 || (dst_pitch_bytes % 4) != 0)
        return fail(WS_ERR_INVALID, "ws_display_composite: row pitch does not fit the format");
    if (dst_format != WS_SURFACE_RGBA8_UNORM && dst_format != WS_SURFACE_BGRA8_UNORM)
        return fail(WS_ERR_INVALID, "ws_display_composite: unknown surface format");
    const float zero[4] = {0, 0, 0, 0};
    return launch_display(d_src, (int)src_format, src_pitch_bytes, width, height, background ? background : zero,
                          (int)dst_format, d_dst, dst_pitch_bytes, static_cast<hipStream_t>(stream));
}

}  // extern "C"
