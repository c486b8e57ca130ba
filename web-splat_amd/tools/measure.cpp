// websplat_measure -- the reference's `measure` binary (src/bin/measure.rs:15-24, 156-197) on libwebsplat_hip:
//   websplat_measure <input.ply|.npz> <cameras.json> [frames in flight, default 1 = the reference's procedure]
// prints "average FPS: <f>" over 10 frames of every training camera at 2048x2048.
#include <cstdio>
#include <cstdlib>

#include "websplat.h"
#include "websplat_env.h"  // the harness-side translation of WS_* switches (the library reads no environment)

int main(int argc, char** argv) {
    if (argc < 3 || argc > 4) {
        std::fprintf(stderr, "usage: %s <input.ply|.npz> <scene cameras.json> [frames_in_flight]\n", argv[0]);
        return 2;
    }
    const unsigned inflight = argc == 4 ? (unsigned)std::atoi(argv[3]) : 1u;
    ws_context* ctx = nullptr;
    ws_pointcloud* pc = nullptr;
    ws_scene* scene = nullptr;
    float fps = 0.0f;
    ws_context_config cfg;
    ws_context_config_from_env(&cfg);
    int rc = ws_context_create_with_config(0, &cfg, &ctx);
    std::printf("reading scene file '%s'\n", argv[2]);
    if (rc == WS_OK) rc = ws_scene_load_json(argv[2], &scene);
    std::printf("reading point cloud file '%s'\n", argv[1]);
    if (rc == WS_OK) rc = ws_pointcloud_load(ctx, argv[1], &pc);
    if (rc == WS_OK) rc = ws_measure(ctx, pc, scene, 10, inflight ? inflight : 1, &fps);
    if (rc == WS_OK) std::printf("average FPS: %g\n", fps);
    else std::fprintf(stderr, "error %d: %s\n", rc, ws_last_error());
    if (pc) ws_pointcloud_destroy(pc);
    if (scene) ws_scene_destroy(scene);
    if (ctx) ws_context_destroy(ctx);
    return rc == WS_OK ? 0 : 1;
}
