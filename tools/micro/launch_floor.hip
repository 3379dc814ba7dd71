// What a dependent launch costs on this GPU, priced the way the chunked-prefill launch pair is: kernels with
// the grid, block and LDS footprint of the key-range-split attention launch (256 workgroups of 256 threads,
// 152 KiB of LDS) and of its merge launch (512 workgroups), doing nothing or moving only the state bytes
// (17,408 per wave and key range, written once, read once), back to back on one stream.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor.bin launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kUnit = 17408;

__global__ __launch_bounds__(256, 1) void empty_big_lds(int* sink) {
    extern __shared__ unsigned char lds[];
    if (sink == nullptr) lds[threadIdx.x] = 1;          // keeps the allocation
}
__global__ __launch_bounds__(256) void empty_small(int* sink) {}

// every wave writes one unit (17 x 1 KiB, 16 bytes per lane and instruction)
__global__ __launch_bounds__(256, 1) void write_states(uint4* part, int nsplit) {
    extern __shared__ unsigned char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int split = blockIdx.x / 64, bid = blockIdx.x % 64;
    uint4* u = part + ((size_t)(bid * 4 + wave) * nsplit + split) * (kUnit / 16);
    const uint4 v = make_uint4(blockIdx.x, wave, lane, 1);
    for (int i = 0; i < 17; ++i) u[i * 64 + lane] = v;
    if (part == nullptr) lds[threadIdx.x] = 1;
}
// 512 workgroups: (bid, wave, half) reads its share of the four units and writes 32 rows of 256 bytes
__global__ __launch_bounds__(256) void read_states(const uint4* part, uint4* out, int nsplit) {
    const int sub = blockIdx.x / 64, bid = blockIdx.x % 64;
    const int wave = sub >> 1, qb = sub & 1, lane = threadIdx.x & 63, d = threadIdx.x >> 6;
    const uint4* u0 = part + ((size_t)(bid * 4 + wave) * nsplit) * (kUnit / 16);
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < nsplit; ++s)
        for (int hf = 0; hf < 2; ++hf) {
            const uint4 v = u0[(size_t)s * (kUnit / 16) + ((qb * 4 + d) * 2 + hf) * 64 + lane];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    out[((size_t)blockIdx.x * 256 + threadIdx.x) / 2] = acc;    // half the bytes read per key range
}

// straight-line code of a given size (8-byte instructions): what a kernel pays for running through code that
// another kernel's body has pushed out of the instruction cache since its last launch
template <int N>
__global__ __launch_bounds__(256) void straight_line(int* sink) {
    int v = threadIdx.x;
    asm volatile(".rept %1\n\tv_add3_u32 %0, %0, 1, 2\n\t.endr" : "+v"(v) : "n"(N));
    if (v == 0x7fffffff) *sink = v;
}

template <typename F>
static float time_us(F&& f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main() {
    const int lds = 155936, nsplit = 4;
    CK(hipFuncSetAttribute((const void*)empty_big_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)write_states, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    uint4 *part, *out; int* sink;
    CK(hipMalloc(&part, (size_t)64 * 4 * nsplit * kUnit));
    CK(hipMalloc(&out, (size_t)512 * 256 * 16));
    CK(hipMalloc(&sink, 4));
    const int reps = 400;
    printf("empty kernel, 256 workgroups x 256 threads, 152 KiB LDS:        %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(empty_big_lds, dim3(256), dim3(256), lds, 0, sink); }, reps));
    printf("empty kernel, 512 workgroups x 256 threads:                     %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(empty_small, dim3(512), dim3(256), 0, 0, sink); }, reps));
    printf("pair of the two:                                                %6.2f us per pair\n",
           time_us([&] { hipLaunchKernelGGL(empty_big_lds, dim3(256), dim3(256), lds, 0, sink);
                         hipLaunchKernelGGL(empty_small, dim3(512), dim3(256), 0, 0, sink); }, reps));
    printf("state bytes only: write launch (17.8 MB)                        %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(write_states, dim3(256), dim3(256), lds, 0, part, nsplit); }, reps));
    printf("state bytes only: read launch (17.8 MB in, 2.1 MB out)          %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(read_states, dim3(512), dim3(256), 0, 0, part, out, nsplit); }, reps));
    printf("state bytes only: pair                                          %6.2f us per pair\n",
           time_us([&] { hipLaunchKernelGGL(write_states, dim3(256), dim3(256), lds, 0, part, nsplit);
                         hipLaunchKernelGGL(read_states, dim3(512), dim3(256), 0, 0, part, out, nsplit); }, reps));
    printf("straight-line code, 512 workgroups: 400 instructions            %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(straight_line<400>, dim3(512), dim3(256), 0, 0, sink); }, reps));
    printf("straight-line code, 256 workgroups: 6000 instructions (48 KB)   %6.2f us per launch\n",
           time_us([&] { hipLaunchKernelGGL(straight_line<6000>, dim3(256), dim3(256), 0, 0, sink); }, reps));
    printf("the two alternating                                             %6.2f us per pair\n",
           time_us([&] { hipLaunchKernelGGL(straight_line<6000>, dim3(256), dim3(256), 0, 0, sink);
                         hipLaunchKernelGGL(straight_line<400>, dim3(512), dim3(256), 0, 0, sink); }, reps));
    return 0;
}
