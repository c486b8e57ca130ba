// Micro-benchmark (analysis tooling, not product): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the
// access patterns this library uses?  MI355X_MICROARCH.md calibrates only the wide coalesced read (FETCH_SIZE = 1/2 of the
// bytes); the blend gathers 20-byte records at random and the sorts read dwords.  Every kernel here moves a KNOWN
// number of bytes; scripts/ubench/fetch_calib.py divides those by the counters of a `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
// pass and writes the factors scripts/pmc_traffic.py applies per kernel class.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fetch_calib.hip -o /tmp/fetch_calib && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_calib_read16(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_read4(const uint32_t* __restrict__ src, size_t n4, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) acc ^= src[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// 20-byte records (dwordx4 + dword, 4-byte aligned) at random indices: the blend's Splat gather
__global__ __launch_bounds__(256) void k_calib_gather20(const uint8_t* __restrict__ table, const uint32_t* __restrict__ idx, size_t n,
                                                       uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        const char* p = reinterpret_cast<const char*>(table) + (size_t)idx[i] * 20;
        uint4 a;
        uint32_t b;
        __builtin_memcpy(&a, p, 16);
        __builtin_memcpy(&b, p + 16, 4);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_gather4(const uint32_t* __restrict__ table, const uint32_t* __restrict__ idx, size_t n,
                                                      uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc ^= table[idx[i]];
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_calib_write16(uint4* __restrict__ dst, size_t n16) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256ull)
        dst[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_calib_write4(uint32_t* __restrict__ dst, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) dst[i] = (uint32_t)i;
}
// dword stores to random positions (a permutation): the radix scatter's worst case
__global__ __launch_bounds__(256) void k_calib_scatter4(uint32_t* __restrict__ dst, const uint32_t* __restrict__ idx, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[idx[i]] = (uint32_t)i;
}

int main() {
    // tables far larger than the 256-MiB Infinity Cache, so that the counters see HBM traffic
    const size_t BYTES = 1536ull << 20;          // streaming buffers
    const size_t RECS = BYTES / 20;              // 20-B records in the gather table
    const size_t NG = 32ull << 20;               // gathers / scatters per launch
    uint8_t* buf;
    uint32_t *idx20, *idx4, *perm, *sink, *dst;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&dst, BYTES));
    CK(hipMalloc(&idx20, NG * 4));
    CK(hipMalloc(&idx4, NG * 4));
    CK(hipMalloc(&perm, NG * 4));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, BYTES));
    CK(hipMemset(dst, 0, BYTES));
    std::vector<uint32_t> h(NG);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < NG; ++i) h[i] = (uint32_t)(rnd() % RECS);
    CK(hipMemcpy(idx20, h.data(), NG * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < NG; ++i) h[i] = (uint32_t)(rnd() % (BYTES / 4));
    CK(hipMemcpy(idx4, h.data(), NG * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < NG; ++i) h[i] = (uint32_t)i;
    for (size_t i = NG - 1; i > 0; --i) { const size_t j = rnd() % (i + 1); std::swap(h[i], h[j]); }
    CK(hipMemcpy(perm, h.data(), NG * 4, hipMemcpyHostToDevice));
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        k_calib_read16<<<grid, 256>>>(reinterpret_cast<const uint4*>(buf), BYTES / 16, sink);
        k_calib_read4<<<grid, 256>>>(reinterpret_cast<const uint32_t*>(buf), BYTES / 4, sink);
        k_calib_gather20<<<grid, 256>>>(buf, idx20, NG, sink);
        k_calib_gather4<<<grid, 256>>>(reinterpret_cast<const uint32_t*>(buf), idx4, NG, sink);
        k_calib_write16<<<grid, 256>>>(reinterpret_cast<uint4*>(dst), BYTES / 16);
        k_calib_write4<<<grid, 256>>>(dst, BYTES / 4);
        k_calib_scatter4<<<grid, 256>>>(dst, perm, NG);
        CK(hipDeviceSynchronize());
    }
    // what each launch moves, by the program's own count (useful bytes; the index streams are listed separately)
    printf("{\"k_calib_read16\": {\"read\": %zu}, \"k_calib_read4\": {\"read\": %zu}, "
           "\"k_calib_gather20\": {\"read\": %zu, \"index_read\": %zu, \"gathers\": %zu, \"table_bytes\": %zu}, "
           "\"k_calib_gather4\": {\"read\": %zu, \"index_read\": %zu, \"gathers\": %zu, \"table_bytes\": %zu}, "
           "\"k_calib_write16\": {\"write\": %zu}, \"k_calib_write4\": {\"write\": %zu}, "
           "\"k_calib_scatter4\": {\"write\": %zu, \"index_read\": %zu}}\n",
           BYTES, BYTES, NG * 20, NG * 4, NG, RECS * 20, NG * 4, NG * 4, NG, BYTES, BYTES, BYTES, NG * 4, NG * 4);
    return 0;
}
