// ply_decode.hip -- raw INRIA 3DGS PLY vertex rows -> the resident 8-plane scene layout, on the GPU.
//
// Replaces, for the load path, the per-vertex work of src/io/ply.rs:50-100 (read_line: SH transpose, sigmoid, exp of
// the log-scales, quaternion normalisation, build_cov (utils.rs:194-203), f32 -> f16) and the re-layout that
// ws_pointcloud_create does on the host: the vertex rows (62 f32 = 248 B at SH degree 3) are uploaded as they sit in
// the file and one kernel writes the eight 16-B planes the preprocess kernel reads (ws_internal.h).  No AoS blobs, no
// host-side conversion, no host-side re-layout.
//
// Arithmetic: f32 in the reference's operation order (this file is compiled with -ffp-contract=off, like the host
// twin ws_ply_rows_convert in host_math.cpp), IEEE division and square root.  The one function that is not an IEEE
// operation is exp(): the reference calls Rust's f32::exp (libm expf), whose last bit is implementation-defined --
// glibc's expf differs from the correctly rounded value for 0.06 % of arguments.  Here exp is evaluated in f64 and
// rounded once to f32 (correctly rounded for all but ~2^-29 of the arguments); after the rounding to f16 that every
// affected field gets, the opacity / covariance halves agree with the host twin except for a 1-ulp difference in
// about 1 of 10^4 values (tests/test_gpu_ply_decode.py states and checks the bound); positions and SH coefficients are
// byte-exact.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "ws_internal.h"

namespace ws {

namespace {

constexpr int PLY_ROWS_PER_BLOCK = 128;

__device__ __forceinline__ float exp_f32(float x) { return (float)exp((double)x); }
__device__ __forceinline__ uint32_t f2h(float f) { return (uint32_t)__half_as_ushort(__float2half_rn(f)); }

// utils.rs:206-212 sigmoid ("numerical stable")
__device__ __forceinline__ float sigmoid_f32(float x) {
    if (x >= 0.0f) return 1.0f / (1.0f + exp_f32(-x));
    const float e = exp_f32(x);
    return e / (1.0f + e);
}

__global__ __launch_bounds__(PLY_ROWS_PER_BLOCK) void k_ply_decode(const float* __restrict__ rows, uint32_t n,
                                                                  uint32_t num_coefs, uint32_t row_len,
                                                                  uint4* __restrict__ planes) {
    extern __shared__ float s_rows[];  // PLY_ROWS_PER_BLOCK x row_len: the block's rows, read as one contiguous run
    const uint32_t first = blockIdx.x * PLY_ROWS_PER_BLOCK;
    const uint32_t count = (n - first) < (uint32_t)PLY_ROWS_PER_BLOCK ? (n - first) : (uint32_t)PLY_ROWS_PER_BLOCK;
    const size_t base = (size_t)first * row_len;
    const uint32_t total = count * row_len;
    for (uint32_t i = threadIdx.x; i < total; i += PLY_ROWS_PER_BLOCK) s_rows[i] = rows[base + i];
    __syncthreads();
    if (threadIdx.x >= count) return;
    const float* r = s_rows + (size_t)threadIdx.x * row_len;  // x y z | nx ny nz | f_dc[3] | f_rest[3][C-1] | opacity | scale[3] | rot[4]
    const float* rest = r + 9;
    const float* tail = rest + (num_coefs - 1u) * 3u;
    const uint32_t i = first + threadIdx.x;

    // opacity, scales, rotation -> covariance (io/ply.rs:77-96)
    const float opacity = sigmoid_f32(tail[0]);
    const float sc[3] = {exp_f32(tail[1]), exp_f32(tail[2]), exp_f32(tail[3])};
    float q0 = tail[4], q1 = tail[5], q2 = tail[6], q3 = tail[7];
    {  // Quaternion::normalize: magnitude2 = s*s + v.dot(v), v * (1 / magnitude)
        const float mag = sqrtf(q0 * q0 + (q1 * q1 + q2 * q2 + q3 * q3));
        const float inv = 1.0f / mag;
        q0 = q0 * inv;
        q1 = q1 * inv;
        q2 = q2 * inv;
        q3 = q3 * inv;
    }
    // cgmath: Matrix3::from(Quaternion), columns m[c][r]
    float m[3][3];
    {
        const float s = q0, x = q1, y = q2, z = q3;
        const float x2 = x + x, y2 = y + y, z2 = z + z;
        const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z;
        const float yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
        const float sy2 = y2 * s, sz2 = z2 * s, sx2 = x2 * s;
        m[0][0] = 1.0f - yy2 - zz2;
        m[0][1] = xy2 + sz2;
        m[0][2] = xz2 - sy2;
        m[1][0] = xy2 - sz2;
        m[1][1] = 1.0f - xx2 - zz2;
        m[1][2] = yz2 + sx2;
        m[2][0] = xz2 + sy2;
        m[2][1] = yz2 - sx2;
        m[2][2] = 1.0f - xx2 - yy2;
    }
    // utils.rs:194-203: l = R * diag(scale) (full row.column products, zero terms included: they decide the sign of
    // an exact zero), cov = l * l^T, upper triangle
    float l[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float s = m[0][rr] * (0 == c ? sc[c] : 0.0f);
            s += m[1][rr] * (1 == c ? sc[c] : 0.0f);
            s += m[2][rr] * (2 == c ? sc[c] : 0.0f);
            l[c][rr] = s;
        }
    float cv[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float s = l[0][rr] * l[0][c];
            s += l[1][rr] * l[1][c];
            s += l[2][rr] * l[2][c];
            cv[c][rr] = s;
        }
    const float cov6[6] = {cv[0][0], cv[0][1], cv[0][2], cv[1][1], cv[1][2], cv[2][2]};

    // plane 0: x, y, z, opacity f16 | pad;  plane 1: cov f16 x 6 | pad
    planes[(size_t)0 * n + i] = make_uint4(__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), f2h(opacity));
    planes[(size_t)1 * n + i] = make_uint4(f2h(cov6[0]) | (f2h(cov6[1]) << 16), f2h(cov6[2]) | (f2h(cov6[3]) << 16),
                                           f2h(cov6[4]) | (f2h(cov6[5]) << 16), 0u);
    // planes 2..7: [[f16; 3]; 16], coefficient-major; PLY f_rest is channel-major [3][C-1] (io/ply.rs:63-75);
    // coefficients above the file's degree are zero
    uint32_t hw[24];
#pragma unroll
    for (int e = 0; e < 48; e += 2) {
        uint32_t pair = 0u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = (e + k) / 3, j = (e + k) % 3;
            float v = 0.0f;
            if (c == 0) v = r[6 + j];
            else if ((uint32_t)c < num_coefs) v = rest[(uint32_t)j * (num_coefs - 1u) + (uint32_t)(c - 1)];
            pair |= f2h(v) << (16 * k);
        }
        hw[e / 2] = pair;
    }
#pragma unroll
    for (int p = 0; p < 6; ++p)
        planes[(size_t)(2 + p) * n + i] = make_uint4(hw[4 * p], hw[4 * p + 1], hw[4 * p + 2], hw[4 * p + 3]);
}

}  // namespace

// d_rows: n x row_len f32 in device memory; planes: PC_PLANES x n x 16 B
int launch_ply_decode(const float* d_rows, uint32_t n, uint32_t sh_deg, uint4* planes, hipStream_t stream) {
    if (n == 0) return WS_OK;
    const uint32_t num_coefs = (sh_deg + 1u) * (sh_deg + 1u);
    const uint32_t row_len = 14u + 3u * num_coefs;
    const uint32_t blocks = (n + PLY_ROWS_PER_BLOCK - 1) / PLY_ROWS_PER_BLOCK;
    const size_t lds = (size_t)PLY_ROWS_PER_BLOCK * row_len * sizeof(float);
    hipLaunchKernelGGL(k_ply_decode, dim3(blocks), dim3(PLY_ROWS_PER_BLOCK), lds, stream, d_rows, n, num_coefs, row_len, planes);
    WS_HIP(hipGetLastError());
    return WS_OK;
}

}  // namespace ws
