// host_npz.cpp -- c3dgs ("compressed 3D Gaussian splatting") .npz loader on the host.
//
// Mirrors src/io/npz.rs:29-56 (NpzReader::new: sh degree from features_rest.shape[1] + 1, optional kernel_size /
// mip_splatting / background_color), :59-225 (read: de-quantisation parameters, xyz f16 -> f32, scaling / rotation
// codebooks -> normalised quaternion + scale -> covariance f16 x 6, GaussianCompressed 24-B records, packed int8 SH
// records dc RGB + rest), src/io/mod.rs:107-150 (new_compressed: bbox grown from Aabb::unit(), centroid, plane fit,
// up only when the bbox radius is >= 10) and src/io/mod.rs:45-61 (magic-byte sniffing: "ply" vs "PK\x03\x04").
//
// The reference reads the archive with the `npyz` crate (0.8, features ["npz","half"]; not vendored under
// /root/reference): an .npz is a ZIP archive of .npy members.  Both container formats are written out here from
// their published specifications (PKWARE APPNOTE 4.3-4.5 incl. the ZIP64 records numpy emits; NEP 1 .npy header);
// DEFLATE members (np.savez_compressed) are inflated with the system zlib.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "ws_internal.h"

using namespace ws;

namespace {

struct NpyArray {
    char kind = 0;      // 'f' float, 'i' signed, 'u' unsigned, 'b' bool
    int item = 0;       // bytes per element
    bool little = true;
    std::vector<uint64_t> shape;
    std::vector<uint8_t> data;  // raw element bytes (C order)
    // product of the dimensions; npy_parse() rejects headers whose product (times the item size) overflows or
    // exceeds the member, so this never wraps for an array that was returned to a caller
    size_t count() const {
        size_t n = 1;
        for (uint64_t s : shape) n *= (size_t)s;
        return n;
    }
};

uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// ---- ZIP --------------------------------------------------------------------------------------------
struct ZipMember {
    std::string name;
    uint16_t method = 0;
    uint32_t crc = 0;
    uint64_t csize = 0, usize = 0, local_off = 0;
};

int zip_directory(const std::vector<uint8_t>& f, std::vector<ZipMember>* out) {
    const size_t n = f.size();
    if (n < 22) return fail(WS_ERR_IO, "npz: file too small to be a zip archive");
    // end-of-central-directory record: last occurrence of PK\5\6 within the final 64 KiB + 22 bytes
    size_t eocd = (size_t)-1;
    const size_t lo = n > 65557 ? n - 65557 : 0;
    for (size_t i = n - 22 + 1; i-- > lo;)
        if (rd32(&f[i]) == 0x06054b50u) {
            eocd = i;
            break;
        }
    if (eocd == (size_t)-1) return fail(WS_ERR_IO, "npz: end-of-central-directory record not found");
    uint64_t entries = rd16(&f[eocd + 10]), cd_size = rd32(&f[eocd + 12]), cd_off = rd32(&f[eocd + 16]);
    if (entries == 0xFFFFu || cd_size == 0xFFFFFFFFu || cd_off == 0xFFFFFFFFu) {
        // ZIP64: locator (PK\6\7) sits right before the EOCD and points at the zip64 EOCD record (PK\6\6)
        if (eocd < 20 || rd32(&f[eocd - 20]) != 0x07064b50u) return fail(WS_ERR_IO, "npz: zip64 locator missing");
        const uint64_t z = rd64(&f[eocd - 20 + 8]);
        if (z > n || n - z < 56 || rd32(&f[z]) != 0x06064b50u) return fail(WS_ERR_IO, "npz: zip64 end record missing");
        entries = rd64(&f[z + 32]);
        cd_size = rd64(&f[z + 40]);
        cd_off = rd64(&f[z + 48]);
    }
    if (cd_off > n || cd_size > n - cd_off) return fail(WS_ERR_IO, "npz: central directory out of bounds");
    if (entries > cd_size / 46) return fail(WS_ERR_IO, "npz: central directory entry count does not fit its size");
    size_t p = (size_t)cd_off;
    for (uint64_t e = 0; e < entries; ++e) {
        if (p > n || n - p < 46 || rd32(&f[p]) != 0x02014b50u) return fail(WS_ERR_IO, "npz: bad central directory entry");
        ZipMember m;
        m.method = rd16(&f[p + 10]);
        m.crc = rd32(&f[p + 16]);
        m.csize = rd32(&f[p + 20]);
        m.usize = rd32(&f[p + 24]);
        const uint16_t nlen = rd16(&f[p + 28]), xlen = rd16(&f[p + 30]), clen = rd16(&f[p + 32]);
        m.local_off = rd32(&f[p + 42]);
        if ((size_t)nlen + xlen + clen > n - p - 46) return fail(WS_ERR_IO, "npz: central directory entry out of bounds");
        m.name.assign(reinterpret_cast<const char*>(&f[p + 46]), nlen);
        // zip64 extended information (header id 1): the fields that overflowed, in this fixed order
        size_t x = p + 46 + nlen;
        const size_t xend = x + xlen;
        while (x + 4 <= xend) {
            const uint16_t id = rd16(&f[x]), sz = rd16(&f[x + 2]);
            if ((size_t)sz > xend - x - 4) break;  // a field that runs past the extra area: ignore the rest
            if (id == 1) {
                size_t q = x + 4;
                if (m.usize == 0xFFFFFFFFu && q + 8 <= xend) { m.usize = rd64(&f[q]); q += 8; }
                if (m.csize == 0xFFFFFFFFu && q + 8 <= xend) { m.csize = rd64(&f[q]); q += 8; }
                if (m.local_off == 0xFFFFFFFFu && q + 8 <= xend) { m.local_off = rd64(&f[q]); q += 8; }
            }
            x += 4 + sz;
        }
        out->push_back(m);
        p += 46 + nlen + xlen + clen;
    }
    return WS_OK;
}

int zip_extract(const std::vector<uint8_t>& f, const ZipMember& m, std::vector<uint8_t>* out) {
    const size_t n = f.size();
    if (m.local_off > n || n - m.local_off < 30 || rd32(&f[m.local_off]) != 0x04034b50u)
        return fail(WS_ERR_IO, "npz: bad local header of " + m.name);
    const size_t data = (size_t)m.local_off + 30 + rd16(&f[m.local_off + 26]) + rd16(&f[m.local_off + 28]);
    if (data > n || m.csize > n - data) return fail(WS_ERR_IO, "npz: member data out of bounds: " + m.name);
    // a DEFLATE stream expands at most 1032:1; anything larger is a forged size, not a big array
    if (m.usize > (1ull << 40) || (m.method == 8 && m.usize / 1032 > m.csize + 1)) return fail(WS_ERR_IO, "npz: implausible member size: " + m.name);
    try {
        out->resize((size_t)m.usize);
    } catch (...) {
        return fail(WS_ERR_OOM, "npz: host allocation failed");
    }
    if (m.method == 0) {
        if (m.csize != m.usize) return fail(WS_ERR_IO, "npz: stored member with differing sizes: " + m.name);
        if (m.usize) std::memcpy(out->data(), &f[data], (size_t)m.usize);
    } else if (m.method == 8) {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) return fail(WS_ERR_IO, "npz: inflateInit2 failed");  // raw deflate
        size_t in_pos = 0, out_pos = 0;
        int zr = Z_OK;
        while (zr != Z_STREAM_END) {  // uInt is 32 bit: feed / drain in < 4 GiB pieces
            const size_t in_left = (size_t)m.csize - in_pos, out_left = (size_t)m.usize - out_pos;
            zs.next_in = const_cast<Bytef*>(&f[data + in_pos]);
            zs.avail_in = (uInt)std::min<size_t>(in_left, 1u << 30);
            zs.next_out = out->data() + out_pos;
            zs.avail_out = (uInt)std::min<size_t>(out_left, 1u << 30);
            const uInt ai = zs.avail_in, ao = zs.avail_out;
            zr = inflate(&zs, Z_NO_FLUSH);
            in_pos += ai - zs.avail_in;
            out_pos += ao - zs.avail_out;
            if (zr != Z_OK && zr != Z_STREAM_END) break;
            if (zr == Z_OK && ai == zs.avail_in && ao == zs.avail_out) break;  // no progress
        }
        inflateEnd(&zs);
        if (zr != Z_STREAM_END || out_pos != m.usize) return fail(WS_ERR_IO, "npz: inflate failed for " + m.name);
    } else {
        return fail(WS_ERR_UNSUPPORTED, "npz: unsupported zip compression method in " + m.name);
    }
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), out->data(), (uInt)out->size());
    if (out->size() < (1ull << 32) && crc != m.crc) return fail(WS_ERR_IO, "npz: CRC mismatch in " + m.name);
    return WS_OK;
}

// ---- NPY (NEP 1) ------------------------------------------------------------------------------------
int npy_parse(const std::vector<uint8_t>& b, const std::string& name, NpyArray* a) {
    if (b.size() < 10 || std::memcmp(b.data(), "\x93NUMPY", 6) != 0) return fail(WS_ERR_IO, "npz: " + name + " is not an .npy member");
    const int major = b[6];
    size_t hlen, hoff;
    if (major == 1) {
        hlen = rd16(&b[8]);
        hoff = 10;
    } else if (major == 2 || major == 3) {
        if (b.size() < 12) return fail(WS_ERR_IO, "npz: truncated npy header in " + name);
        hlen = rd32(&b[8]);
        hoff = 12;
    } else {
        return fail(WS_ERR_UNSUPPORTED, "npz: unknown .npy version in " + name);
    }
    if (hlen > b.size() - hoff) return fail(WS_ERR_IO, "npz: truncated npy header in " + name);
    const std::string h(reinterpret_cast<const char*>(&b[hoff]), hlen);
    auto value_after = [&](const char* key) -> size_t {
        const size_t k = h.find(key);
        if (k == std::string::npos) return std::string::npos;
        const size_t c = h.find(':', k);
        return c == std::string::npos ? std::string::npos : c + 1;
    };
    size_t p = value_after("'descr'");
    if (p == std::string::npos) return fail(WS_ERR_IO, "npz: descr missing in " + name);
    const size_t q0 = h.find('\'', p), q1 = h.find('\'', q0 + 1);
    if (q0 == std::string::npos || q1 == std::string::npos) return fail(WS_ERR_UNSUPPORTED, "npz: structured dtype in " + name);
    std::string descr = h.substr(q0 + 1, q1 - q0 - 1);
    a->little = true;
    if (!descr.empty() && (descr[0] == '<' || descr[0] == '>' || descr[0] == '|' || descr[0] == '=')) {
        a->little = descr[0] != '>';
        descr = descr.substr(1);
    }
    if (descr == "?") descr = "b1";
    if (descr.size() < 2) return fail(WS_ERR_UNSUPPORTED, "npz: dtype of " + name);
    a->kind = descr[0];
    a->item = std::atoi(descr.c_str() + 1);
    if (!(a->kind == 'f' || a->kind == 'i' || a->kind == 'u' || a->kind == 'b') || a->item < 1 || a->item > 8)
        return fail(WS_ERR_UNSUPPORTED, "npz: dtype '" + descr + "' of " + name);
    p = value_after("'fortran_order'");
    if (p != std::string::npos) {
        const size_t v = h.find_first_not_of(' ', p);
        if (v == std::string::npos) return fail(WS_ERR_IO, "npz: fortran_order without a value in " + name);
        if (h.compare(v, 4, "True") == 0) return fail(WS_ERR_UNSUPPORTED, "npz: fortran-ordered array " + name);
    }
    p = value_after("'shape'");
    if (p == std::string::npos) return fail(WS_ERR_IO, "npz: shape missing in " + name);
    const size_t s0 = h.find('(', p), s1 = h.find(')', s0);
    if (s0 == std::string::npos || s1 == std::string::npos) return fail(WS_ERR_IO, "npz: bad shape in " + name);
    a->shape.clear();
    for (size_t i = s0 + 1; i < s1;) {
        while (i < s1 && (h[i] == ' ' || h[i] == ',')) ++i;
        if (i >= s1) break;
        a->shape.push_back(std::strtoull(h.c_str() + i, nullptr, 10));
        while (i < s1 && h[i] != ',') ++i;
    }
    // element count and byte size with overflow checks: no dimension may exceed what the member can hold
    const size_t avail = b.size() - hoff - hlen;
    size_t bytes = (size_t)a->item;
    bool empty = false;
    for (uint64_t dim : a->shape) empty = empty || dim == 0;  // e.g. features_rest of a degree-0 cloud: (n, 0, 3)
    if (empty) bytes = 0;
    for (uint64_t dim : a->shape) {
        if (empty) break;
        if (dim > avail || __builtin_mul_overflow(bytes, (size_t)dim, &bytes) || bytes > avail) {
            a->shape.clear();
            return fail(WS_ERR_IO, "npz: truncated data in " + name);
        }
    }
    a->data.assign(b.begin() + hoff + hlen, b.begin() + hoff + hlen + bytes);
    if (!a->little && a->item > 1)
        for (size_t i = 0; i < bytes; i += a->item)
            for (int k = 0; k < a->item / 2; ++k) std::swap(a->data[i + k], a->data[i + a->item - 1 - k]);
    return WS_OK;
}

struct Npz {
    std::vector<uint8_t> file;
    std::map<std::string, ZipMember> members;  // by array name (".npy" stripped)
    bool has(const std::string& n) const { return members.count(n) != 0; }
    int get(const std::string& n, NpyArray* a) const {
        auto it = members.find(n);
        if (it == members.end()) return fail(WS_ERR_IO, "npz: array " + n + " missing");
        std::vector<uint8_t> raw;
        int rc = zip_extract(file, it->second, &raw);
        if (rc) return rc;
        return npy_parse(raw, n, a);
    }
};

int npz_open(const char* path, Npz* z) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(WS_ERR_IO, std::string("npz: cannot open ") + path);
    std::fseek(f, 0, SEEK_END);
    const long sz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    try {
        z->file.resize(sz > 0 ? (size_t)sz : 0);
    } catch (...) {
        std::fclose(f);
        return fail(WS_ERR_OOM, "npz: host allocation failed");
    }
    const size_t got = z->file.empty() ? 0 : std::fread(z->file.data(), 1, z->file.size(), f);
    std::fclose(f);
    if (got != z->file.size()) return fail(WS_ERR_IO, "npz: short read");
    std::vector<ZipMember> dir;
    int rc = zip_directory(z->file, &dir);
    if (rc) return rc;
    for (const ZipMember& m : dir) {
        std::string n = m.name;
        if (n.size() > 4 && n.compare(n.size() - 4, 4, ".npy") == 0) n.resize(n.size() - 4);
        z->members[n] = m;
    }
    return WS_OK;
}

// element i of an array as f32 / i32, accepting the numeric dtypes numpy may have written the scalar with
// (npyz's into_vec::<f32> only takes '<f4'; python floats saved through np.savez arrive as '<f8')
float elem_f32(const NpyArray& a, size_t i) {
    const uint8_t* p = a.data.data() + i * a.item;
    if (a.kind == 'f') {
        if (a.item == 4) { float v; std::memcpy(&v, p, 4); return v; }
        if (a.item == 8) { double v; std::memcpy(&v, p, 8); return (float)v; }
        if (a.item == 2) return host_f16_to_f32(rd16(p));
    }
    if (a.kind == 'i') {
        if (a.item == 1) return (float)(int8_t)p[0];
        if (a.item == 2) return (float)(int16_t)rd16(p);
        if (a.item == 4) return (float)(int32_t)rd32(p);
        if (a.item == 8) return (float)(int64_t)rd64(p);
    }
    if (a.kind == 'u' || a.kind == 'b') {
        if (a.item == 1) return (float)p[0];
        if (a.item == 2) return (float)rd16(p);
        if (a.item == 4) return (float)rd32(p);
        if (a.item == 8) return (float)rd64(p);
    }
    return 0.0f;
}
int32_t elem_i32(const NpyArray& a, size_t i) {
    const uint8_t* p = a.data.data() + i * a.item;
    if (a.kind == 'f') return (int32_t)elem_f32(a, i);
    if (a.item == 1) return a.kind == 'i' ? (int32_t)(int8_t)p[0] : (int32_t)p[0];
    if (a.item == 2) return a.kind == 'i' ? (int32_t)(int16_t)rd16(p) : (int32_t)rd16(p);
    if (a.item == 4) return (int32_t)rd32(p);
    return (int32_t)(int64_t)rd64(p);
}

// get_npz_value (npz.rs:262-275): first element of an optional array
int scalar_f32(const Npz& z, const char* name, float dflt, float* out, bool* present = nullptr) {
    *out = dflt;
    if (present) *present = false;
    if (!z.has(name)) return WS_OK;
    NpyArray a;
    int rc = z.get(name, &a);
    if (rc) return rc;
    if (a.count() == 0) return fail(WS_ERR_IO, std::string("npz: array empty: ") + name);
    *out = elem_f32(a, 0);
    if (present) *present = true;
    return WS_OK;
}
int scalar_i32(const Npz& z, const char* name, int32_t dflt, int32_t* out, bool* present = nullptr) {
    *out = dflt;
    if (present) *present = false;
    if (!z.has(name)) return WS_OK;
    NpyArray a;
    int rc = z.get(name, &a);
    if (rc) return rc;
    if (a.count() == 0) return fail(WS_ERR_IO, std::string("npz: array empty: ") + name);
    *out = elem_i32(a, 0);
    if (present) *present = true;
    return WS_OK;
}

int expect_i8(const NpyArray& a, const char* name) {
    if (a.item != 1 || !(a.kind == 'i' || a.kind == 'u')) return fail(WS_ERR_IO, std::string("npz: ") + name + " must be int8");
    return WS_OK;
}

}  // namespace

struct ws_npz_cloud_impl {
    ws_npz_cloud pub;
    std::vector<uint8_t> gaussians, sh, covars;
};

extern "C" {

static int npz_read_impl(const char* path, ws_npz_cloud** out) {
    Npz z;
    int rc = npz_open(path, &z);
    if (rc) return rc;

    // ---- NpzReader::new (npz.rs:29-56) ----
    uint32_t sh_deg = 0;
    NpyArray features_rest;
    if (z.has("features_rest")) {
        if ((rc = z.get("features_rest", &features_rest))) return rc;
        if (features_rest.shape.size() < 2) return fail(WS_ERR_IO, "npz: features_rest must be at least 2-D");
        const uint32_t ncoef = (uint32_t)features_rest.shape[1] + 1;  // utils.rs:180-190 sh_deg_from_num_coefs
        const uint32_t root = (uint32_t)std::lround(std::sqrt((double)ncoef));
        if (root * root != ncoef || root == 0) return fail(WS_ERR_IO, "npz: num sh coefs not valid");
        sh_deg = root - 1;
        if (sh_deg > 3) return fail(WS_ERR_UNSUPPORTED, "npz: sh degree > 3");
    }
    auto* impl = new (std::nothrow) ws_npz_cloud_impl();
    if (!impl) return fail(WS_ERR_OOM, "ws_npz_read: host allocation failed");
    struct Guard {  // frees the cloud on every early exit, exceptions included; released on success
        ws_npz_cloud_impl* p;
        ~Guard() { delete p; }
    } guard{impl};
    ws_npz_cloud& pc = impl->pub;
    std::memset(&pc, 0, sizeof pc);
#define NPZ_TRY(expr)        \
    do {                     \
        if ((rc = (expr))) { \
            return rc;       \
        }                    \
    } while (0)
    bool present = false;
    NPZ_TRY(scalar_f32(z, "kernel_size", 0.0f, &pc.kernel_size, &present));
    pc.has_kernel_size = present;
    int32_t mip = 0;
    NPZ_TRY(scalar_i32(z, "mip_splatting", 0, &mip, &present));
    pc.has_mip_splatting = present;
    pc.mip_splatting = mip != 0;
    if (z.has("background_color")) {
        NpyArray bg;
        NPZ_TRY(z.get("background_color", &bg));
        if (bg.count() != 3) {
            return fail(WS_ERR_IO, "npz: background_color must have 3 elements");
        }
        pc.has_background_color = 1;
        for (int k = 0; k < 3; ++k) pc.background_color[k] = elem_f32(bg, k);
    }

    // ---- read (npz.rs:59-225) ----
    float opacity_scale, scaling_scale, rotation_scale, dc_scale, rest_scale, sf_scale = 1.0f;
    int32_t opacity_zp, scaling_zp_i, rotation_zp_i, dc_zp, rest_zp, sf_zp = 0;
    NPZ_TRY(scalar_f32(z, "opacity_scale", 1.0f, &opacity_scale));
    NPZ_TRY(scalar_i32(z, "opacity_zero_point", 0, &opacity_zp));
    NPZ_TRY(scalar_f32(z, "scaling_scale", 1.0f, &scaling_scale));
    NPZ_TRY(scalar_i32(z, "scaling_zero_point", 0, &scaling_zp_i));
    NPZ_TRY(scalar_f32(z, "rotation_scale", 1.0f, &rotation_scale));
    NPZ_TRY(scalar_i32(z, "rotation_zero_point", 0, &rotation_zp_i));
    NPZ_TRY(scalar_f32(z, "features_dc_scale", 1.0f, &dc_scale));
    NPZ_TRY(scalar_i32(z, "features_dc_zero_point", 0, &dc_zp));
    NPZ_TRY(scalar_f32(z, "features_rest_scale", 1.0f, &rest_scale));
    NPZ_TRY(scalar_i32(z, "features_rest_zero_point", 0, &rest_zp));
    const float scaling_zp = (float)scaling_zp_i, rotation_zp = (float)rotation_zp_i;
    const bool has_sf = z.has("scaling_factor_scale");  // npz.rs:87-95
    NpyArray scaling_factor;
    if (has_sf) {
        NPZ_TRY(scalar_f32(z, "scaling_factor_scale", 1.0f, &sf_scale));
        NPZ_TRY(scalar_i32(z, "scaling_factor_zero_point", 0, &sf_zp));
        NPZ_TRY(z.get("scaling_factor", &scaling_factor));
        NPZ_TRY(expect_i8(scaling_factor, "scaling_factor"));
    }
    NpyArray xyz, scaling, rotation, opacity, features_dc, fidx, gidx;
    NPZ_TRY(z.get("xyz", &xyz));
    if (xyz.kind != 'f' || xyz.count() % 3 != 0) {
        return fail(WS_ERR_IO, "npz: xyz must be a float array of 3-vectors");
    }
    NPZ_TRY(z.get("scaling", &scaling));
    NPZ_TRY(expect_i8(scaling, "scaling"));
    NPZ_TRY(z.get("rotation", &rotation));
    NPZ_TRY(expect_i8(rotation, "rotation"));
    NPZ_TRY(z.get("opacity", &opacity));
    NPZ_TRY(expect_i8(opacity, "opacity"));
    NPZ_TRY(z.get("features_dc", &features_dc));
    NPZ_TRY(expect_i8(features_dc, "features_dc"));
    if (features_rest.data.empty() && !z.has("features_rest")) {
        return fail(WS_ERR_IO, "npz: array features_rest missing");  // try_get_npz_array, npz.rs:152
    }
    NPZ_TRY(expect_i8(features_rest, "features_rest"));
    const bool has_fidx = z.has("feature_indices"), has_gidx = z.has("gaussian_indices");
    if (has_fidx) NPZ_TRY(z.get("feature_indices", &fidx));
    if (has_gidx) NPZ_TRY(z.get("gaussian_indices", &gidx));

    const size_t num_points = xyz.count() / 3;
    const size_t ncoef = (size_t)(sh_deg + 1) * (sh_deg + 1);
    const size_t rest_len = ncoef * 3 - 3;
    const size_t n_sh = features_dc.count() / 3;
    const size_t n_geo = rotation.count() / 4;
    if (num_points == 0 || num_points >= (1u << 30) || opacity.count() < num_points || scaling.count() / 3 < n_geo ||
        features_rest.count() < n_sh * rest_len || (has_sf && scaling_factor.count() < num_points) ||
        (has_fidx && fidx.count() < num_points) || (has_gidx && gidx.count() < num_points)) {
        return fail(WS_ERR_IO, "npz: array lengths are inconsistent");
    }
    try {
        impl->gaussians.resize(num_points * 24);
        impl->sh.resize(n_sh * ncoef * 3);
        impl->covars.resize(n_geo * 12);
    } catch (...) {
        return fail(WS_ERR_OOM, "ws_npz_read: host allocation failed");
    }
    // GaussianCompressed (pointcloud.rs:14-22): xyz f32 x3, opacity i8, scale_factor i8, pad, geometry_idx, sh_idx
    for (size_t i = 0; i < num_points; ++i) {
        uint8_t* g = impl->gaussians.data() + i * 24;
        for (int k = 0; k < 3; ++k) {
            const float v = elem_f32(xyz, i * 3 + k);
            std::memcpy(g + 4 * k, &v, 4);
        }
        g[12] = opacity.data[i];
        g[13] = has_sf ? scaling_factor.data[i] : 0;
        g[14] = g[15] = 0;
        const uint32_t gi = has_gidx ? (uint32_t)elem_i32(gidx, i) : (uint32_t)i;
        const uint32_t si = has_fidx ? (uint32_t)elem_i32(fidx, i) : (uint32_t)i;
        std::memcpy(g + 16, &gi, 4);
        std::memcpy(g + 20, &si, 4);
    }
    // packed SH records (npz.rs:183-196): dc RGB, then the (C-1)*3 rest bytes of the same entry
    for (size_t i = 0; i < n_sh; ++i) {
        uint8_t* s = impl->sh.data() + i * ncoef * 3;
        s[0] = features_dc.data[i * 3 + 0];
        s[1] = features_dc.data[i * 3 + 1];
        s[2] = features_dc.data[i * 3 + 2];
        if (rest_len) std::memcpy(s + 3, features_rest.data.data() + i * rest_len, rest_len);
    }
    // covariance codebook (npz.rs:102-130, 197-202): de-quantise, normalise, build_cov, round to f16
    for (size_t i = 0; i < n_geo; ++i) {
        float s[3], q[4];
        for (int k = 0; k < 3; ++k) {
            const float v = ((float)(int8_t)scaling.data[i * 3 + k] - scaling_zp) * scaling_scale;
            s[k] = has_sf ? std::fmax(v, 0.0f) : std::exp(v);
        }
        if (has_sf) {  // Vector3::normalize
            const float mag = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            const float inv = 1.0f / mag;
            for (float& v : s) v = v * inv;
        }
        for (int k = 0; k < 4; ++k) q[k] = ((float)(int8_t)rotation.data[i * 4 + k] - rotation_zp) * rotation_scale;
        {  // Quaternion::new(c0, c1, c2, c3).normalize(): magnitude2 = s*s + v.dot(v)
            const float mag = std::sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
            const float inv = 1.0f / mag;
            for (float& v : q) v = v * inv;
        }
        float cov[6];
        build_cov(q, s, cov);
        for (int k = 0; k < 6; ++k) {
            const uint16_t h = host_f32_to_f16(cov[k]);
            std::memcpy(impl->covars.data() + i * 12 + 2 * k, &h, 2);
        }
    }
#undef NPZ_TRY
    pc.num_points = (uint32_t)num_points;
    pc.sh_deg = sh_deg;
    pc.gaussians = impl->gaussians.data();
    pc.gaussians_bytes = impl->gaussians.size();
    pc.sh_coefs = impl->sh.data();
    pc.sh_coefs_bytes = impl->sh.size();
    pc.covars = impl->covars.data();
    pc.covars_bytes = impl->covars.size();
    pc.quantization.color_dc = {dc_zp, dc_scale, {0, 0}};
    pc.quantization.color_rest = {rest_zp, rest_scale, {0, 0}};
    pc.quantization.opacity = {opacity_zp, opacity_scale, {0, 0}};
    pc.quantization.scaling_factor = {sf_zp, sf_scale, {0, 0}};
    guard.p = nullptr;
    *out = &impl->pub;
    return WS_OK;
}

// No C++ exception may cross the C ABI: allocation failures and any std:: range error raised while picking a crafted
// archive apart come back as error codes.
int ws_npz_read(const char* path, ws_npz_cloud** out) {
    if (!path || !out) return fail(WS_ERR_INVALID, "ws_npz_read: null argument");
    *out = nullptr;
    try {
        return npz_read_impl(path, out);
    } catch (const std::bad_alloc&) {
        return fail(WS_ERR_OOM, "ws_npz_read: host allocation failed");
    } catch (...) {
        return fail(WS_ERR_IO, "ws_npz_read: malformed archive");
    }
}

void ws_npz_free(ws_npz_cloud* pc) {
    if (pc) delete reinterpret_cast<ws_npz_cloud_impl*>(pc);  // pub is the first member
}

// GenericGaussianPointCloud::new_compressed (io/mod.rs:107-150) + PointCloud::new
int ws_pointcloud_load_npz(ws_context* ctx, const char* path, ws_pointcloud** out) {
    if (!ctx || !path || !out) return fail(WS_ERR_INVALID, "ws_pointcloud_load_npz: null argument");
    *out = nullptr;
    ws_npz_cloud* c = nullptr;
    int rc = ws_npz_read(path, &c);
    if (rc) return rc;
    ws_pointcloud_desc d;
    std::memset(&d, 0, sizeof d);
    d.num_points = c->num_points;
    d.sh_deg = c->sh_deg;
    d.compressed = 1;
    d.gaussians = c->gaussians;
    d.gaussians_bytes = c->gaussians_bytes;
    d.sh_coefs = c->sh_coefs;
    d.sh_coefs_bytes = c->sh_coefs_bytes;
    d.covars = c->covars;
    d.covars_bytes = c->covars_bytes;
    d.quantization = &c->quantization;
    ws_aabb unit;  // Aabb::unit(), io/mod.rs:119
    for (int k = 0; k < 3; ++k) {
        unit.min[k] = -1.0f;
        unit.max[k] = 1.0f;
    }
    rc = ws_pointcloud_stats(c->gaussians, c->num_points, 24, &unit, &d.bbox, d.center, &d.has_up, d.up);
    if (rc == WS_OK) {
        d.has_kernel_size = c->has_kernel_size;
        d.kernel_size = c->kernel_size;
        d.has_mip_splatting = c->has_mip_splatting;
        d.mip_splatting = c->mip_splatting;
        d.has_background_color = c->has_background_color;
        std::memcpy(d.background_color, c->background_color, sizeof d.background_color);
        rc = ws_pointcloud_create(ctx, &d, out);
    }
    ws_npz_free(c);
    return rc;
}

// GenericGaussianPointCloud::load (io/mod.rs:45-61): the reader is chosen by the file's magic bytes
int ws_pointcloud_load(ws_context* ctx, const char* path, ws_pointcloud** out) {
    if (!ctx || !path || !out) return fail(WS_ERR_INVALID, "ws_pointcloud_load: null argument");
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(WS_ERR_IO, std::string("ws_pointcloud_load: cannot open ") + path);
    unsigned char sig[4] = {0, 0, 0, 0};
    const size_t got = std::fread(sig, 1, 4, f);
    std::fclose(f);
    if (got >= 3 && std::memcmp(sig, "ply", 3) == 0) return ws_pointcloud_load_ply(ctx, path, out);
    if (got == 4 && std::memcmp(sig, "PK\x03\x04", 4) == 0) return ws_pointcloud_load_npz(ctx, path, out);
    return fail(WS_ERR_IO, "ws_pointcloud_load: Unknown file format");
}

}  // extern "C"
