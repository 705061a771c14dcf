// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes this library uses.
// Every kernel touches a known set of 128-byte lines of a 1 GiB buffer (4x the Infinity Cache) exactly once, so the
// bytes that have to cross the L2's memory side are known: lines * 128 (or lines * 64 if half lines are fetched).
// Build:  hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
// Run:    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib -- tools/pmc_calib     (and again with WRITE_SIZE)
// tools/pmc_calib prints one line per kernel with the bytes it must move; tools/pmc_calib_report.py joins them with
// the counters.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static constexpr size_t BYTES = size_t(1) << 30;

// 16 B per lane, fully coalesced stream (k_classify's density stream, row stores)
__global__ void calib_read16(const uint4* __restrict__ p, uint32_t* sink, size_t n16) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (; i < n16; i += stride) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 4 B per lane, coalesced
__global__ void calib_read4(const uint32_t* __restrict__ p, uint32_t* sink, size_t n4) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (; i < n4; i += stride) acc ^= p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// 1 B per lane, consecutive bytes (64 B per wave instruction)
__global__ void calib_read1(const uint8_t* __restrict__ p, uint32_t* sink, size_t n1) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (; i < n1; i += stride) acc += p[i];
    if (acc == 0x12345678u) sink[0] = acc;
}
// 1 B per lane, byte stride S (S = 2, 4, 8: the old level >= 1 gathers; S = 128: one byte per line)
template <int S>
__global__ void calib_gather1(const uint8_t* __restrict__ p, uint32_t* sink, size_t count) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (; i < count; i += stride) acc += p[i * S];
    if (acc == 0x12345678u) sink[0] = acc;
}
// 24-byte row segments at a 1024-byte pitch (the 19^3 neighbourhood staging of one 16^3 block: 3 x 8 B per row)
__global__ void calib_rows24(const uint8_t* __restrict__ p, uint32_t* sink, size_t rows) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    uint32_t acc = 0;
    for (; i < rows * 3; i += stride) {
        size_t r = i / 3, k = i % 3;
        uint2 v = *reinterpret_cast<const uint2*>(p + r * 1024 + 504 + k * 8);   // bytes 504..527: straddles a line boundary
        acc ^= v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// 16 B per lane coalesced stores
__global__ void calib_write16(uint4* __restrict__ p, size_t n16) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n16; i += stride) p[i] = make_uint4(uint32_t(i), 1, 2, 3);
}
// 4 B per lane coalesced stores (index lists)
__global__ void calib_write4(uint32_t* __restrict__ p, size_t n4) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < n4; i += stride) p[i] = uint32_t(i);
}
// 48-byte records, one per lane, three 16-B stores (the vertex pool)
__global__ void calib_write48(uint4* __restrict__ p, size_t recs) {
    size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    size_t stride = size_t(gridDim.x) * blockDim.x;
    for (; i < recs; i += stride) {
        p[i * 3 + 0] = make_uint4(uint32_t(i), 1, 2, 3);
        p[i * 3 + 1] = make_uint4(uint32_t(i), 4, 5, 6);
        p[i * 3 + 2] = make_uint4(uint32_t(i), 7, 8, 9);
    }
}

int main() {
    uint8_t* buf = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&buf, BYTES));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, BYTES));
    CHECK(hipDeviceSynchronize());
    const int B = 256, G = 256 * 16;
    // name, bytes that must be fetched (lines touched * 128), bytes written
    printf("calib_read16 read %zu write 0\n", BYTES);
    calib_read16<<<G, B>>>(reinterpret_cast<const uint4*>(buf), sink, BYTES / 16);
    printf("calib_read4 read %zu write 0\n", BYTES);
    calib_read4<<<G, B>>>(reinterpret_cast<const uint32_t*>(buf), sink, BYTES / 4);
    printf("calib_read1 read %zu write 0\n", BYTES);
    calib_read1<<<G, B>>>(buf, sink, BYTES);
    printf("calib_gather1<2> read %zu write 0\n", BYTES);
    calib_gather1<2><<<G, B>>>(buf, sink, BYTES / 2);
    printf("calib_gather1<8> read %zu write 0\n", BYTES);
    calib_gather1<8><<<G, B>>>(buf, sink, BYTES / 8);
    printf("calib_gather1<128> read %zu write 0\n", BYTES);
    calib_gather1<128><<<G, B>>>(buf, sink, BYTES / 128);
    printf("calib_rows24 read %zu write 0\n", (BYTES / 1024) * 256);   // two lines per row
    calib_rows24<<<G, B>>>(buf, sink, BYTES / 1024);
    printf("calib_write16 read 0 write %zu\n", BYTES);
    calib_write16<<<G, B>>>(reinterpret_cast<uint4*>(buf), BYTES / 16);
    printf("calib_write4 read 0 write %zu\n", BYTES);
    calib_write4<<<G, B>>>(reinterpret_cast<uint32_t*>(buf), BYTES / 4);
    printf("calib_write48 read 0 write %zu\n", (BYTES / 48) * 48);
    calib_write48<<<G, B>>>(reinterpret_cast<uint4*>(buf), BYTES / 48);
    CHECK(hipDeviceSynchronize());
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
