// host_scene.cpp -- dataset scene files and image write-out on the host: the callers' side of the
// hot path in the reference's offline front-ends.
//
// Mirrors src/scene.rs:13-24 (SceneCamera), :113-154 (Scene::from_cameras / from_json: every 8th camera of the file
// is a test view, duplicates by id replaced by the later entry, extend = largest camera-to-camera distance),
// :156-194 (camera / cameras(split) sorted by id / nearest_camera) and the PNG write-out of bin/render.rs:127
// (`image` crate, RGBA8).  serde_json and the `image` crate are not vendored in the reference: the JSON grammar
// (RFC 8259) and the PNG container (ISO/IEC 15948: IHDR / IDAT / IEND, filter type 0, zlib stream, CRC-32) are
// written out here; DEFLATE and CRC come from the system zlib.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "ws_internal.h"

using namespace ws;

struct ws_scene {
    std::map<uint32_t, ws_scene_camera> cameras;  // by id (scene.rs:115 HashMap<usize, SceneCamera>)
    float extend = 0.0f;
};

namespace {

// ---- a small JSON reader (RFC 8259); values are kept as a tagged tree ----------------------------------
struct JValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double num = 0.0;
    bool b = false;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue* get(const char* key) const {
        for (const auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* p;
    const char* end;
    std::string err;
    void ws_() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    bool fail_(const char* m) {
        if (err.empty()) err = m;
        return false;
    }
    bool parse_string(std::string* out) {
        if (p >= end || *p != '"') return fail_("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail_("bad escape");
                switch (*p) {
                    case 'n': out->push_back('\n'); break;
                    case 't': out->push_back('\t'); break;
                    case 'r': out->push_back('\r'); break;
                    case 'b': out->push_back('\b'); break;
                    case 'f': out->push_back('\f'); break;
                    case 'u': {
                        if (p + 4 >= end) return fail_("bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; ++i) {
                            const char c = p[i];
                            cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : 99);
                        }
                        p += 4;
                        if (cp < 0x80) {
                            out->push_back((char)cp);
                        } else if (cp < 0x800) {
                            out->push_back((char)(0xC0 | (cp >> 6)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        } else {
                            out->push_back((char)(0xE0 | (cp >> 12)));
                            out->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                            out->push_back((char)(0x80 | (cp & 0x3F)));
                        }
                        break;
                    }
                    default: out->push_back(*p);  // \" \\ \/
                }
                ++p;
            } else {
                out->push_back(*p++);
            }
        }
        if (p >= end) return fail_("unterminated string");
        ++p;
        return true;
    }
    bool parse(JValue* v, int depth = 0) {
        if (depth > 64) return fail_("nesting too deep");
        ws_();
        if (p >= end) return fail_("unexpected end of input");
        const char c = *p;
        if (c == '{') {
            v->kind = JValue::Object;
            ++p;
            ws_();
            if (p < end && *p == '}') { ++p; return true; }
            while (true) {
                ws_();
                std::string key;
                if (!parse_string(&key)) return false;
                ws_();
                if (p >= end || *p != ':') return fail_("expected ':'");
                ++p;
                JValue child;
                if (!parse(&child, depth + 1)) return false;
                v->obj.emplace_back(std::move(key), std::move(child));
                ws_();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail_("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v->kind = JValue::Array;
            ++p;
            ws_();
            if (p < end && *p == ']') { ++p; return true; }
            while (true) {
                JValue child;
                if (!parse(&child, depth + 1)) return false;
                v->arr.push_back(std::move(child));
                ws_();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail_("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v->kind = JValue::String;
            return parse_string(&v->str);
        }
        if (end - p >= 4 && std::memcmp(p, "true", 4) == 0) { v->kind = JValue::Bool; v->b = true; p += 4; return true; }
        if (end - p >= 5 && std::memcmp(p, "false", 5) == 0) { v->kind = JValue::Bool; v->b = false; p += 5; return true; }
        if (end - p >= 4 && std::memcmp(p, "null", 4) == 0) { v->kind = JValue::Null; p += 4; return true; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const std::string tmp(p, std::min<size_t>((size_t)(end - p), 64));
            char* e = nullptr;
            v->num = std::strtod(tmp.c_str(), &e);
            if (e == tmp.c_str()) return fail_("bad number");
            p += e - tmp.c_str();
            v->kind = JValue::Number;
            return true;
        }
        return fail_("unexpected character");
    }
};

bool num_field(const JValue& o, const char* key, double* out) {
    const JValue* v = o.get(key);
    if (!v || v->kind != JValue::Number) return false;
    *out = v->num;
    return true;
}

int camera_from_json(const JValue& o, size_t index, ws_scene_camera* c) {
    std::memset(c, 0, sizeof *c);
    const std::string where = " in camera " + std::to_string(index);
    if (o.kind != JValue::Object) return fail(WS_ERR_IO, "cameras.json: expected an object" + where);
    double id, w, h, fx, fy;
    if (!num_field(o, "id", &id) || !num_field(o, "width", &w) || !num_field(o, "height", &h) ||
        !num_field(o, "fx", &fx) || !num_field(o, "fy", &fy))
        return fail(WS_ERR_IO, "cameras.json: missing id / width / height / fx / fy" + where);
    if (id < 0 || id != std::floor(id) || w < 0 || h < 0 || w != std::floor(w) || h != std::floor(h))
        return fail(WS_ERR_IO, "cameras.json: id / width / height must be non-negative integers" + where);
    const JValue* name = o.get("img_name");
    if (!name || name->kind != JValue::String) return fail(WS_ERR_IO, "cameras.json: missing img_name" + where);
    const JValue* pos = o.get("position");
    const JValue* rot = o.get("rotation");
    if (!pos || pos->kind != JValue::Array || pos->arr.size() != 3 || !rot || rot->kind != JValue::Array || rot->arr.size() != 3)
        return fail(WS_ERR_IO, "cameras.json: position must be [3] and rotation [3][3]" + where);
    c->id = (uint32_t)id;
    std::strncpy(c->img_name, name->str.c_str(), sizeof(c->img_name) - 1);
    c->width = (uint32_t)w;
    c->height = (uint32_t)h;
    c->fx = (float)fx;
    c->fy = (float)fy;
    for (int k = 0; k < 3; ++k) {
        if (pos->arr[k].kind != JValue::Number) return fail(WS_ERR_IO, "cameras.json: bad position" + where);
        c->position[k] = (float)pos->arr[k].num;
        const JValue& row = rot->arr[k];
        if (row.kind != JValue::Array || row.arr.size() != 3) return fail(WS_ERR_IO, "cameras.json: bad rotation" + where);
        for (int j = 0; j < 3; ++j) {
            if (row.arr[j].kind != JValue::Number) return fail(WS_ERR_IO, "cameras.json: bad rotation" + where);
            c->rotation[k * 3 + j] = (float)row.arr[j].num;
        }
    }
    return WS_OK;
}

void be32(uint8_t* p, uint32_t v) {
    p[0] = (uint8_t)(v >> 24);
    p[1] = (uint8_t)(v >> 16);
    p[2] = (uint8_t)(v >> 8);
    p[3] = (uint8_t)v;
}

bool png_chunk(FILE* f, const char type[4], const uint8_t* data, uint32_t len) {
    uint8_t hdr[8];
    be32(hdr, len);
    std::memcpy(hdr + 4, type, 4);
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, hdr + 4, 4);
    if (len) crc = crc32(crc, data, len);
    uint8_t tail[4];
    be32(tail, (uint32_t)crc);
    return std::fwrite(hdr, 1, 8, f) == 8 && (len == 0 || std::fwrite(data, 1, len, f) == len) && std::fwrite(tail, 1, 4, f) == 4;
}

}  // namespace

extern "C" {

// Scene::from_json (scene.rs:140-154) on an in-memory document
static int scene_load_json_impl(const char* path, ws_scene** out);
static int scene_from_json_text_impl(const char* text, size_t len, ws_scene** out) {
    JParser jp{text, text + len, {}};
    JValue root;
    if (!jp.parse(&root)) return fail(WS_ERR_IO, "cameras.json: " + jp.err);
    jp.ws_();
    if (jp.p != jp.end) return fail(WS_ERR_IO, "cameras.json: trailing characters");
    if (root.kind != JValue::Array) return fail(WS_ERR_IO, "cameras.json: top level must be an array of cameras");
    std::vector<ws_scene_camera> cams(root.arr.size());
    for (size_t i = 0; i < root.arr.size(); ++i) {
        int rc = camera_from_json(root.arr[i], i, &cams[i]);
        if (rc) return rc;
        // "7 out of 8 cameras are taken as training images" (scene.rs:143-151): file position, not id
        cams[i].split = (i % 8 == 0) ? WS_SPLIT_TEST : WS_SPLIT_TRAIN;
    }
    ws_scene* s = new (std::nothrow) ws_scene();
    if (!s) return fail(WS_ERR_OOM, "ws_scene_from_json_text: host allocation failed");
    struct Guard {
        ws_scene* p;
        ~Guard() { delete p; }
    } guard{s};
    // max_distance (scene.rs:189-201): O(n^2) over ALL cameras of the file, squared distances, one sqrt
    float max_d2 = 0.0f;
    for (size_t i = 0; i < cams.size(); ++i)
        for (size_t j = i + 1; j < cams.size(); ++j) {
            const float dx = cams[i].position[0] - cams[j].position[0], dy = cams[i].position[1] - cams[j].position[1],
                        dz = cams[i].position[2] - cams[j].position[2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            max_d2 = std::fmax(max_d2, d2);
        }
    s->extend = std::sqrt(max_d2);
    for (const ws_scene_camera& c : cams) s->cameras[c.id] = c;  // a later duplicate replaces the earlier one
    guard.p = nullptr;
    *out = s;
    return WS_OK;
}

// No C++ exception crosses the C ABI (allocation failures while building the DOM of a huge or hostile file included).
int ws_scene_from_json_text(const char* text, size_t len, ws_scene** out) {
    if (!text || !out) return fail(WS_ERR_INVALID, "ws_scene_from_json_text: null argument");
    *out = nullptr;
    try {
        return scene_from_json_text_impl(text, len, out);
    } catch (const std::bad_alloc&) {
        return fail(WS_ERR_OOM, "ws_scene_from_json_text: host allocation failed");
    } catch (...) {
        return fail(WS_ERR_IO, "cameras.json: malformed input");
    }
}

int ws_scene_load_json(const char* path, ws_scene** out) {
    if (!path || !out) return fail(WS_ERR_INVALID, "ws_scene_load_json: null argument");
    *out = nullptr;
    try {
        return scene_load_json_impl(path, out);
    } catch (const std::bad_alloc&) {
        return fail(WS_ERR_OOM, "ws_scene_load_json: host allocation failed");
    } catch (...) {
        return fail(WS_ERR_IO, "ws_scene_load_json: malformed input");
    }
}

static int scene_load_json_impl(const char* path, ws_scene** out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(WS_ERR_IO, std::string("ws_scene_load_json: cannot open ") + path);
    std::string text;
    char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, got);
    std::fclose(f);
    return ws_scene_from_json_text(text.data(), text.size(), out);
}

void ws_scene_destroy(ws_scene* s) { delete s; }
uint32_t ws_scene_num_cameras(const ws_scene* s) { return s ? (uint32_t)s->cameras.size() : 0u; }
float ws_scene_extend(const ws_scene* s) { return s ? s->extend : 0.0f; }

// Scene::cameras(split) (scene.rs:164-177): filtered by split, sorted by id
uint32_t ws_scene_cameras(const ws_scene* s, int split, uint32_t capacity, ws_scene_camera* out) {
    if (!s) return 0;
    uint32_t n = 0;
    for (const auto& kv : s->cameras) {  // std::map iterates in ascending id
        if (split != WS_SPLIT_ALL && kv.second.split != split) continue;
        if (out && n < capacity) out[n] = kv.second;
        ++n;
    }
    return n;
}

int ws_scene_get_camera(const ws_scene* s, uint32_t id, ws_scene_camera* out) {
    if (!s) return 0;
    auto it = s->cameras.find(id);
    if (it == s->cameras.end()) return 0;
    if (out) *out = it->second;
    return 1;
}

// Scene::nearest_camera (scene.rs:180-194): min over (distance2 * 1e6) as u32; the reference walks a HashMap, so ties
// are broken arbitrarily there -- here by the smallest id
int ws_scene_nearest_camera(const ws_scene* s, const float pos[3], int split, uint32_t* id) {
    if (!s || !pos) return 0;
    bool found = false;
    uint32_t best_key = 0, best_id = 0;
    for (const auto& kv : s->cameras) {
        const ws_scene_camera& c = kv.second;
        if (split != WS_SPLIT_ALL && c.split != split) continue;
        const float dx = c.position[0] - pos[0], dy = c.position[1] - pos[1], dz = c.position[2] - pos[2];
        const float d2 = (dx * dx + dy * dy + dz * dz) * 1e6f;
        const uint32_t key = d2 >= 4294967296.0f ? 0xFFFFFFFFu : (d2 > 0.0f ? (uint32_t)d2 : 0u);  // Rust `as u32` saturates
        if (!found || key < best_key) {
            found = true;
            best_key = key;
            best_id = c.id;
        }
    }
    if (found && id) *id = best_id;
    return found ? 1 : 0;
}

// RGBA8 PNG, no interlace, filter type 0 on every scanline (ImageBuffer::save, bin/render.rs:127)
int ws_png_write_rgba8(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba, size_t row_stride_bytes) {
    if (!path || !rgba || width == 0 || height == 0 || row_stride_bytes < (size_t)width * 4)
        return fail(WS_ERR_INVALID, "ws_png_write_rgba8: bad argument");
    std::vector<uint8_t> raw;
    try {
        raw.resize(((size_t)width * 4 + 1) * height);
    } catch (...) {
        return fail(WS_ERR_OOM, "ws_png_write_rgba8: host allocation failed");
    }
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t* row = raw.data() + (size_t)y * (width * 4 + 1);
        row[0] = 0;
        std::memcpy(row + 1, rgba + (size_t)y * row_stride_bytes, (size_t)width * 4);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return fail(WS_ERR_IO, "ws_png_write_rgba8: deflate failed");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(WS_ERR_IO, std::string("ws_png_write_rgba8: cannot create ") + path);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    uint8_t ihdr[13];
    be32(ihdr, width);
    be32(ihdr + 4, height);
    ihdr[8] = 8;   // bit depth
    ihdr[9] = 6;   // colour type: RGBA
    ihdr[10] = 0;  // deflate
    ihdr[11] = 0;  // adaptive filtering (type 0 used)
    ihdr[12] = 0;  // no interlace
    bool ok = std::fwrite(sig, 1, 8, f) == 8 && png_chunk(f, "IHDR", ihdr, 13) &&
              png_chunk(f, "IDAT", comp.data(), (uint32_t)clen) && png_chunk(f, "IEND", nullptr, 0);
    ok = (std::fclose(f) == 0) && ok;
    return ok ? WS_OK : fail(WS_ERR_IO, "ws_png_write_rgba8: write failed");
}

}  // extern "C"
