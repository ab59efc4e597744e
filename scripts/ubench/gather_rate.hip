// gather_rate.hip -- how fast does one CU serve scattered 4/8-byte loads, as a function of how many distinct 128-B lines a
// wave instruction touches and of how the lanes that share a line are placed?  (input for the hash-grid gather layout)
//   hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// G lanes share a line.  contiguous: lanes [g*G, g*G+G) share; interleaved: lanes with equal (lane % (64/G)) share.
// MODE: 0 plain, 1 relaxed agent-scope atomic load (sc1: served by L2, bypasses L1), 2 nontemporal (nt)
template <int MODE>
__global__ void __launch_bounds__(256) kmode(const uint8_t* __restrict__ table, uint32_t n_lines, int G, int iters, uint32_t* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t grp = lane / G;
    const uint32_t within = (mix(lane * 977u + 13u) % 32) * 4;
    uint32_t acc = 0;
    uint32_t h = mix(wave * 64u + grp + 1u);
#pragma unroll 8
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t* p = reinterpret_cast<const uint32_t*>(table + (size_t)((h >> 8) % n_lines) * 128 + within);
        if (MODE == 0) acc += *p;
        if (MODE == 1) acc += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) acc += __builtin_nontemporal_load(p);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int BYTES>
__global__ void __launch_bounds__(256) k(const uint8_t* __restrict__ table, uint32_t n_lines, int G, int interleaved, int iters,
                                         uint32_t* __restrict__ sink) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t grp = interleaved ? lane % (64 / G) : lane / G;
    const uint32_t within = (mix(lane * 977u + 13u) % (128 / BYTES)) * BYTES;
    uint32_t acc = 0;
    uint32_t h = mix(wave * 64u + grp + 1u);
#pragma unroll 8
    for (int i = 0; i < iters; i++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t line = (h >> 8) % n_lines;
        const uint8_t* p = table + (size_t)line * 128 + within;
        if (BYTES == 4) acc += *reinterpret_cast<const uint32_t*>(p);
        else { const uint2 v = *reinterpret_cast<const uint2*>(p); acc += v.x ^ v.y; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t max_bytes = 64u << 20;
    uint8_t* table;
    uint32_t* sink;
    CK(hipMalloc(&table, max_bytes));
    CK(hipMemset(table, 1, max_bytes));
    CK(hipMalloc(&sink, 4));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1000.0;
    printf("CUs %d clock %.0f MHz\n", cus, mhz);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 2048, blocks = cus * 4;  // 4 blocks x 4 waves = 16 waves per CU
    const size_t sizes[] = {8u << 10, 1u << 20, 25u << 20};
    printf("%-6s %-8s %-4s %-12s %10s %14s %14s\n", "bytes", "table", "G", "placement", "ms", "cyc/instr/CU", "Glane-ld/s");
    for (int bytes : {4, 8})
        for (size_t sz : sizes)
            for (int G : {1, 2, 4, 8, 16, 64})
                for (int inter : {0, 1}) {
                    if ((G == 1 || G == 64) && inter) continue;
                    const uint32_t n_lines = (uint32_t)(sz / 128);
                    for (int rep = 0; rep < 2; rep++) {
                        CK(hipEventRecord(a));
                        if (bytes == 4) k<4><<<blocks, 256>>>(table, n_lines, G, inter, iters, sink);
                        else k<8><<<blocks, 256>>>(table, n_lines, G, inter, iters, sink);
                        CK(hipEventRecord(b));
                        CK(hipEventSynchronize(b));
                    }
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    const double instr_per_cu = (double)blocks * 4 * iters / cus;
                    const double cyc = ms * 1e-3 * mhz * 1e6 / instr_per_cu;
                    printf("%-6d %-8zu %-4d %-12s %10.3f %14.1f %14.2f\n", bytes, sz >> 10, G, inter ? "interleaved" : "contiguous", ms, cyc,
                           (double)blocks * 4 * iters * 64 / (ms * 1e-3) / 1e9);
                }
    printf("\ncache-policy variants, 4-byte loads, 8 in flight per wave\n%-8s %-4s %-10s %10s %14s\n", "table", "G", "mode", "ms", "cyc/instr/CU");
    const char* names[3] = {"plain", "sc1(atomic)", "nt"};
    for (size_t sz : sizes)
        for (int G : {1, 4})
            for (int mode = 0; mode < 3; mode++) {
                const uint32_t n_lines = (uint32_t)(sz / 128);
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(a));
                    if (mode == 0) kmode<0><<<blocks, 256>>>(table, n_lines, G, iters, sink);
                    if (mode == 1) kmode<1><<<blocks, 256>>>(table, n_lines, G, iters, sink);
                    if (mode == 2) kmode<2><<<blocks, 256>>>(table, n_lines, G, iters, sink);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                }
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                const double instr_per_cu = (double)blocks * 4 * iters / cus;
                printf("%-8zu %-4d %-10s %10.3f %14.1f\n", sz >> 10, G, names[mode], ms, ms * 1e-3 * mhz * 1e6 / instr_per_cu);
            }
    return 0;
}
