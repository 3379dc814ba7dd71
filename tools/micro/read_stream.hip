// What this GPU streams from HBM when a kernel only READS (the decode attention launch reads 667 MB and writes 1 MB):
// every workgroup sums a contiguous slice of a buffer larger than the last-level cache with 16-byte loads, a few
// launch shapes; the buffer is swept once per launch and the launches rotate over several buffers.
//   hipcc --offload-arch=gfx950 -O3 -o read_stream.bin read_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void read_sum(const uint4* __restrict__ src4, size_t n_vec, unsigned* out) {
    const v4u* src = reinterpret_cast<const v4u*>(src4);
    const size_t per_wg = n_vec / gridDim.x;
    const v4u* p = src + (size_t)blockIdx.x * per_wg;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per_wg; i += UNROLL * 256) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// The paged cache's pattern: the buffer is [token rows][8 kv heads][256 B]; a workgroup streams ONE head of a run of
// consecutive rows (as the decode launch does for a (sequence, kv head) pair): one wave-load = 4 rows x 256 B at a
// 2 KiB stride.
template <int UNROLL>
__global__ __launch_bounds__(256) void read_head_rows(const uint4* __restrict__ src4, size_t n_rows, unsigned* out) {
    const v4u* src = reinterpret_cast<const v4u*>(src4);
    const int head = blockIdx.x & 7;
    const size_t groups = gridDim.x >> 3;
    const size_t rows_per_wg = n_rows / groups;
    const size_t row0 = (size_t)(blockIdx.x >> 3) * rows_per_wg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    // wave w takes rows 4 (4 i + w) .. + 3 of the run per step i; lane: row lane / 16, 16-byte piece lane % 16
    for (size_t r = 4 * wave; r + 16 * (UNROLL - 1) + 3 < rows_per_wg; r += 16 * UNROLL) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            v[u] = __builtin_nontemporal_load(src + ((row0 + r + 16 * u + (lane >> 4)) * 8 + head) * 16 + (lane & 15));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

// The decode launch's own mapping: the four waves of a workgroup stream four NEIGHBOURING heads of the same run of rows
// (1 KiB contiguous per row and workgroup; the other four heads belong to the next workgroup, i.e. another XCD).
__global__ __launch_bounds__(256) void read_four_heads(const uint4* __restrict__ src4, size_t n_rows, unsigned* out) {
    const v4u* src = reinterpret_cast<const v4u*>(src4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int head = (blockIdx.x & 1) * 4 + wave;
    const size_t groups = gridDim.x >> 1;
    const size_t rows_per_wg = n_rows / groups;
    const size_t row0 = (size_t)(blockIdx.x >> 1) * rows_per_wg;
    unsigned acc = 0;
    for (size_t r = 0; r + 15 < rows_per_wg; r += 16) {
        v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_nontemporal_load(src + ((row0 + r + 4 * u + (lane >> 4)) * 8 + head) * 16 + (lane & 15));
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = 640ull << 20;
    const int nbuf = 4;
    uint4* buf[nbuf]; unsigned* out;
    for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemset(buf[i], 1, bytes)); }
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int grids[] = {256, 512, 1024, 2048, 4096};
    for (int g : grids) {
        for (int variant = 0; variant < 2; ++variant) {
            auto launch = [&](int i) {
                if (variant == 0) hipLaunchKernelGGL(read_sum<4>, dim3(g), dim3(256), 0, 0, buf[i % nbuf], bytes / 16, out);
                else hipLaunchKernelGGL(read_sum<8>, dim3(g), dim3(256), 0, 0, buf[i % nbuf], bytes / 16, out);
            };
            for (int i = 0; i < 8; ++i) launch(i);
            CK(hipDeviceSynchronize());
            const int reps = 40;
            CK(hipEventRecord(a));
            for (int i = 0; i < reps; ++i) launch(i);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("read-only stream, %5d workgroups x 256 threads, %d loads in flight per lane: %7.1f us per 640 MiB  %6.0f GB/s\n",
                   g, variant ? 8 : 4, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
        }
    }
    for (int g : {2048, 4096, 8192}) {
        auto launch = [&](int i) { hipLaunchKernelGGL(read_head_rows<4>, dim3(g), dim3(256), 0, 0, buf[i % nbuf], bytes / 2048, out); };
        for (int i = 0; i < 8; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("one head of 2 KiB rows per workgroup (256-byte pieces), %5d workgroups: %7.1f us per 640 MiB  %6.0f GB/s\n",
               g, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    }
    for (int g : {512, 1024, 2048, 4096}) {
        auto launch = [&](int i) { hipLaunchKernelGGL(read_four_heads, dim3(g), dim3(256), 0, 0, buf[i % nbuf], bytes / 2048, out); };
        for (int i = 0; i < 8; ++i) launch(i);
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("four neighbouring heads per workgroup, one per wave (1 KiB per row), %5d workgroups: %7.1f us per 640 MiB  %6.0f GB/s\n",
               g, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e9);
    }
    return 0;
}
