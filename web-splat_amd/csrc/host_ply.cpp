// host_ply.cpp -- INRIA 3D-Gaussian-Splatting PLY loader on the host (the step in front of the hot path).
//
// Mirrors src/io/ply.rs:28-48 (header: sh degree from the number of f_* properties, vertex count,
// comments `mip=..`, `kernel_size=..`, `background_color=r,g,b`), :164-196 (binary little/big endian
// bodies, ascii unsupported like the reference's todo!()), and io/mod.rs:45-105 (magic-byte sniffing,
// bbox grown from Aabb::zeroed(), centroid, plane fit).  ply-rs 0.1.3 (header parser) is not vendored in
// the reference; the PLY header grammar is written out here.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ws_internal.h"

using namespace ws;

namespace {

struct PlyHeader {
    bool little_endian = true;
    uint32_t num_vertices = 0;
    uint32_t num_props = 0;
    uint32_t num_f_props = 0;
    bool all_float = true;
    std::vector<std::string> comments;
    long body_offset = 0;
};

bool read_line(FILE* f, std::string* out) {
    out->clear();
    int c;
    while ((c = std::fgetc(f)) != EOF) {
        if (c == '\n') return true;
        if (c != '\r') out->push_back((char)c);
    }
    return !out->empty();
}

int parse_header(FILE* f, PlyHeader* h) {
    std::string line;
    if (!read_line(f, &line) || line != "ply") return fail(WS_ERR_IO, "ply: missing magic");
    bool in_vertex = false, have_format = false;
    while (read_line(f, &line)) {
        if (line == "end_header") {
            h->body_offset = std::ftell(f);
            if (!have_format) return fail(WS_ERR_IO, "ply: missing format line");
            return WS_OK;
        }
        if (line.rfind("format ", 0) == 0) {
            have_format = true;
            if (line.find("binary_little_endian") != std::string::npos)
                h->little_endian = true;
            else if (line.find("binary_big_endian") != std::string::npos)
                h->little_endian = false;
            else
                return fail(WS_ERR_UNSUPPORTED, "ply: ascii format not supported (as in the reference)");
        } else if (line.rfind("comment ", 0) == 0) {
            h->comments.push_back(line.substr(8));
        } else if (line.rfind("element ", 0) == 0) {
            char name[64];
            unsigned long long cnt = 0;
            if (std::sscanf(line.c_str(), "element %63s %llu", name, &cnt) != 2) return fail(WS_ERR_IO, "ply: bad element line");
            in_vertex = std::strcmp(name, "vertex") == 0;
            if (in_vertex) {
                // the device side indexes Gaussians with 30 bits; a count that does not fit must not be truncated
                if (cnt >= (1ull << 30)) return fail(WS_ERR_UNSUPPORTED, "ply: more than 2^30-1 vertices");
                h->num_vertices = (uint32_t)cnt;
            }
        } else if (line.rfind("property ", 0) == 0 && in_vertex) {
            char type[32], name[64];
            if (std::sscanf(line.c_str(), "property %31s %63s", type, name) != 2) return fail(WS_ERR_IO, "ply: bad property line");
            if (std::strcmp(type, "float") != 0 && std::strcmp(type, "float32") != 0) h->all_float = false;
            h->num_props++;
            if (std::strncmp(name, "f_", 2) == 0) h->num_f_props++;
        }
    }
    return fail(WS_ERR_IO, "ply: end_header not found");
}

bool comment_value(const PlyHeader& h, const char* key, std::string* value) {
    for (const std::string& c : h.comments)
        if (c.find(key) != std::string::npos) {
            const size_t eq = c.rfind('=');
            *value = eq == std::string::npos ? c : c.substr(eq + 1);
            return true;
        }
    return false;
}

}  // namespace

struct ws_ply_cloud_impl {
    ws_ply_cloud pub;  // first member: the public view
    std::vector<uint8_t> gaussians, sh;
};

static int ply_read_impl(const char* path, ws_ply_cloud** out);

// No C++ exception crosses the C ABI.
extern "C" int ws_ply_read(const char* path, ws_ply_cloud** out) {
    if (!path || !out) return fail(WS_ERR_INVALID, "ws_ply_read: null argument");
    *out = nullptr;
    try {
        return ply_read_impl(path, out);
    } catch (const std::bad_alloc&) {
        return fail(WS_ERR_OOM, "ply: host allocation failed");
    } catch (...) {
        return fail(WS_ERR_IO, "ply: malformed file");
    }
}

// Header + raw vertex rows (native endianness) of an INRIA 3DGS PLY: the part of PlyReader::new / read that touches
// the file.  rows: num_vertices x (14 + 3 * (sh_deg + 1)^2) f32.
static int ply_read_rows(const char* path, PlyHeader* h, uint32_t* sh_deg_out, std::vector<float>* rows) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(WS_ERR_IO, std::string("ws_ply_read: cannot open ") + path);
    int rc = parse_header(f, h);
    if (rc) {
        std::fclose(f);
        return rc;
    }
    // io/ply.rs:102-113: sh degree from the number of f_* properties / 3
    const uint32_t ncoef = h->num_f_props / 3;
    const uint32_t root = (uint32_t)std::lround(std::sqrt((double)ncoef));
    if (h->num_f_props % 3 != 0 || root * root != ncoef || root == 0 || root > 4 || !h->all_float) {
        std::fclose(f);
        return fail(WS_ERR_IO, "ply: number of sh coefficients cannot be mapped to an sh degree <= 3");
    }
    *sh_deg_out = root - 1;
    const uint32_t row_len = 3 + 3 + 3 * ncoef + 1 + 3 + 4;
    if (h->num_props != row_len) {
        std::fclose(f);
        return fail(WS_ERR_IO, "ply: vertex layout is not the INRIA 3DGS layout (x,y,z,n*,f_dc*,f_rest*,opacity,scale*,rot*)");
    }
    // the body must actually be there before anything is sized by the header's vertex count
    std::fseek(f, 0, SEEK_END);
    const long file_size = std::ftell(f);
    const uint64_t need = (uint64_t)h->num_vertices * row_len * sizeof(float);
    if (file_size < 0 || (uint64_t)(file_size - h->body_offset) < need) {
        std::fclose(f);
        return fail(WS_ERR_IO, "ply: truncated vertex data");
    }
    try {
        rows->resize((size_t)h->num_vertices * row_len);
    } catch (...) {
        std::fclose(f);
        return fail(WS_ERR_OOM, "ply: host allocation failed");
    }
    std::fseek(f, h->body_offset, SEEK_SET);
    const size_t got = rows->empty() ? 0 : std::fread(rows->data(), sizeof(float), rows->size(), f);
    std::fclose(f);
    if (got != rows->size()) return fail(WS_ERR_IO, "ply: truncated vertex data");
    if (!h->little_endian) {
        uint32_t* w = reinterpret_cast<uint32_t*>(rows->data());
        const ws::OmpQuietWorkers omp_quiet;  // (ws_internal.h: the region's workers sleep at once instead of spinning 200 ms)
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < (int64_t)rows->size(); ++i) w[i] = __builtin_bswap32(w[i]);
    }
    return WS_OK;
}

// the optional header comments: `mip=..`, `kernel_size=..`, `background_color=r,g,b` (io/ply.rs:123-163)
static int ply_header_meta(const PlyHeader& h, int32_t* has_mip, int32_t* mip, int32_t* has_ks, float* ks, int32_t* has_bg,
                           float bg[3]) {
    std::string v;
    *has_mip = *mip = *has_ks = *has_bg = 0;
    *ks = 0.0f;
    if (comment_value(h, "mip", &v)) {  // io/ply.rs:123-130 (parse::<bool>)
        if (v != "true" && v != "false") return fail(WS_ERR_IO, "ply: bad mip comment");
        *has_mip = 1;
        *mip = (v == "true") ? 1 : 0;
    }
    if (comment_value(h, "kernel_size", &v)) {  // io/ply.rs:131-138: parse::<f32>()? -- junk is an error, not 0.0
        char* end = nullptr;
        const float k = std::strtof(v.c_str(), &end);
        if (v.empty() || v[0] == ' ' || v[0] == '\t' || end == v.c_str() || *end != 0) return fail(WS_ERR_IO, "ply: bad kernel_size comment");
        *has_ks = 1;
        *ks = k;
    }
    if (comment_value(h, "background_color", &v)) {
        float c[3];
        if (std::sscanf(v.c_str(), "%f,%f,%f", &c[0], &c[1], &c[2]) == 3) {
            *has_bg = 1;
            std::memcpy(bg, c, sizeof c);
        }  // parse failures are only warned about in the reference (io/ply.rs:36-38)
    }
    return WS_OK;
}

static int ply_read_impl(const char* path, ws_ply_cloud** out) {
    PlyHeader h;
    uint32_t sh_deg = 0;
    std::vector<float> rows;
    int rc = ply_read_rows(path, &h, &sh_deg, &rows);
    if (rc) return rc;
    auto* impl = new (std::nothrow) ws_ply_cloud_impl();
    if (!impl) return fail(WS_ERR_OOM, "ply: host allocation failed");
    struct Guard {
        ws_ply_cloud_impl* p;
        ~Guard() { delete p; }
    } guard{impl};
    impl->gaussians.resize((size_t)h.num_vertices * 28);
    impl->sh.resize((size_t)h.num_vertices * 96);
    // the per-vertex conversion (io/ply.rs:50-100) runs over all host cores (OpenMP inside ws_ply_rows_convert)
    if ((rc = ws_ply_rows_convert(rows.data(), h.num_vertices, sh_deg, impl->gaussians.data(), impl->sh.data()))) return rc;
    rows.clear();
    rows.shrink_to_fit();

    ws_ply_cloud& d = impl->pub;
    std::memset(&d, 0, sizeof d);
    d.num_points = h.num_vertices;
    d.sh_deg = sh_deg;
    d.gaussians = impl->gaussians.data();
    d.gaussians_bytes = impl->gaussians.size();
    d.sh_coefs = impl->sh.data();
    d.sh_coefs_bytes = impl->sh.size();
    ws_aabb zero;  // Aabb::zeroed(), io/mod.rs:74
    std::memset(&zero, 0, sizeof zero);
    if (h.num_vertices == 0) {
        // The reference's reader accepts an empty vertex list (io/ply.rs:164-196 loops zero times): bbox stays
        // Aabb::zeroed() and the centroid is 0/0 (io/mod.rs:74-84).  Such a cloud can be read, not uploaded
        // (ws_pointcloud_create rejects it, as wgpu rejects zero-sized bindings).
        d.bbox = zero;
        d.center[0] = d.center[1] = d.center[2] = std::nanf("");
    } else if ((rc = ws_pointcloud_stats(impl->gaussians.data(), h.num_vertices, 28, &zero, &d.bbox, d.center, &d.has_up, d.up))) {
        return rc;
    }
    if ((rc = ply_header_meta(h, &d.has_mip_splatting, &d.mip_splatting, &d.has_kernel_size, &d.kernel_size,
                              &d.has_background_color, d.background_color)))
        return rc;
    guard.p = nullptr;
    *out = &impl->pub;
    return WS_OK;
}

extern "C" void ws_ply_free(ws_ply_cloud* pc) {
    if (pc) delete reinterpret_cast<ws_ply_cloud_impl*>(pc);  // pub is the first member
}

// PlyReader::read + PointCloud::new.  The vertex rows go to the device as they sit in the file and are converted
// there (ws_pointcloud_create_from_ply_rows -> k_ply_decode); ws_context_config::ply_decode_host selects the host conversion
// (ws_ply_read + ws_pointcloud_create) for comparison.
static int load_ply_impl(ws_context* ctx, const char* path, ws_pointcloud** out) {
    if (ctx && ctx->ply_decode_host) {
        ws_ply_cloud* h = nullptr;
        int rc = ws_ply_read(path, &h);
        if (rc) return rc;
        ws_pointcloud_desc d;
        std::memset(&d, 0, sizeof d);
        d.num_points = h->num_points;
        d.sh_deg = h->sh_deg;
        d.compressed = 0;
        d.gaussians = h->gaussians;
        d.gaussians_bytes = h->gaussians_bytes;
        d.sh_coefs = h->sh_coefs;
        d.sh_coefs_bytes = h->sh_coefs_bytes;
        d.bbox = h->bbox;
        std::memcpy(d.center, h->center, sizeof d.center);
        d.has_up = h->has_up;
        std::memcpy(d.up, h->up, sizeof d.up);
        d.has_mip_splatting = h->has_mip_splatting;
        d.mip_splatting = h->mip_splatting;
        d.has_kernel_size = h->has_kernel_size;
        d.kernel_size = h->kernel_size;
        d.has_background_color = h->has_background_color;
        std::memcpy(d.background_color, h->background_color, sizeof d.background_color);
        rc = ws_pointcloud_create(ctx, &d, out);
        ws_ply_free(h);
        return rc;
    }
    PlyHeader h;
    uint32_t sh_deg = 0;
    std::vector<float> rows;
    int rc = ply_read_rows(path, &h, &sh_deg, &rows);
    if (rc) return rc;
    ws_pointcloud_desc meta;
    std::memset(&meta, 0, sizeof meta);
    if ((rc = ply_header_meta(h, &meta.has_mip_splatting, &meta.mip_splatting, &meta.has_kernel_size, &meta.kernel_size,
                              &meta.has_background_color, meta.background_color)))
        return rc;
    return ws_pointcloud_create_from_ply_rows(ctx, rows.data(), h.num_vertices, sh_deg, &meta, out);
}

extern "C" int ws_pointcloud_load_ply(ws_context* ctx, const char* path, ws_pointcloud** out) {
    if (!ctx || !path || !out) return fail(WS_ERR_INVALID, "ws_pointcloud_load_ply: null argument");
    *out = nullptr;
    try {
        return load_ply_impl(ctx, path, out);
    } catch (const std::bad_alloc&) {
        return fail(WS_ERR_OOM, "ply: host allocation failed");
    } catch (...) {
        return fail(WS_ERR_IO, "ply: malformed file");
    }
}
