// websplat_render -- the reference's `render` binary (src/bin/render.rs:15-31, 129-183) on libwebsplat_hip:
//   websplat_render <input.ply|.npz> <cameras.json> <img_out>
// renders the test split, then the train split, to <img_out>/{test,train}/<index:05>.png.
#include <cstdio>

#include "websplat.h"
#include "websplat_env.h"  // the harness-side translation of WS_* switches (the library reads no environment)

int main(int argc, char** argv) {
    if (argc != 4) {
        std::fprintf(stderr, "usage: %s <input.ply|.npz> <scene cameras.json> <img_out dir>\n", argv[0]);
        return 2;
    }
    ws_context* ctx = nullptr;
    ws_pointcloud* pc = nullptr;
    ws_scene* scene = nullptr;
    ws_context_config cfg;
    ws_context_config_from_env(&cfg);
    int rc = ws_context_create_with_config(0, &cfg, &ctx);
    std::printf("reading scene file '%s'\n", argv[2]);
    if (rc == WS_OK) rc = ws_scene_load_json(argv[2], &scene);
    std::printf("reading point cloud file '%s'\n", argv[1]);
    if (rc == WS_OK) rc = ws_pointcloud_load(ctx, argv[1], &pc);
    const int splits[2] = {WS_SPLIT_TEST, WS_SPLIT_TRAIN};
    for (int split : splits) {
        uint32_t n = 0;
        if (rc == WS_OK) {
            std::printf("saving images to '%s/%s'\n", argv[3], split == WS_SPLIT_TEST ? "test" : "train");
            rc = ws_render_views(ctx, pc, scene, split, argv[3], &n);
            std::printf("rendered %u views\n", n);
        }
    }
    if (rc != WS_OK) std::fprintf(stderr, "error %d: %s\n", rc, ws_last_error());
    else std::printf("done!\n");
    if (pc) ws_pointcloud_destroy(pc);
    if (scene) ws_scene_destroy(scene);
    if (ctx) ws_context_destroy(ctx);
    return rc == WS_OK ? 0 : 1;
}
