// What the first instructions of a short kernel cost: launches of 512 workgroups x 256 threads that do nothing,
// read their kernel arguments, and then follow one or two dependent loads from global memory before one store
// per workgroup -- the chain every small kernel of a serving step starts with (arguments -> sequence bounds ->
// data).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel durations; prints launch periods itself.
//   hipcc --offload-arch=gfx950 -O3 -o latency_chain.bin latency_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { int* out; const int* idx; const int* table; int pad[40]; };

__global__ __launch_bounds__(256) void hop0(Args a) {}
__global__ __launch_bounds__(256) void hop1(Args a) {
    if (threadIdx.x == 0) a.out[blockIdx.x] = a.pad[3];
}
__global__ __launch_bounds__(256) void hop2(Args a) {
    const int i = a.idx[blockIdx.x & 7];
    if (threadIdx.x == 0) a.out[blockIdx.x] = i;
}
__global__ __launch_bounds__(256) void hop3(Args a) {
    const int i = a.idx[blockIdx.x & 7];
    const int j = a.table[i + (threadIdx.x & 63)];
    if ((threadIdx.x & 63) == 0) a.out[blockIdx.x * 4 + (threadIdx.x >> 6)] = j;
}
__global__ __launch_bounds__(256) void hop4(Args a) {
    const int i = a.idx[blockIdx.x & 7];
    const int j = a.table[i + (threadIdx.x & 63)];
    const int k = a.table[(j & 1023) + (threadIdx.x & 63)];
    if ((threadIdx.x & 63) == 0) a.out[blockIdx.x * 4 + (threadIdx.x >> 6)] = k;
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
    Args a{};
    int *out, *idx, *table;
    CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&idx, 4096)); CK(hipMalloc(&table, 1 << 20));
    CK(hipMemset(idx, 0, 4096)); CK(hipMemset(table, 0, 1 << 20));
    a.out = out; a.idx = idx; a.table = table;
    const int reps = 400;
    printf("period, no argument read:                      %6.2f us\n", time_us([&] { hipLaunchKernelGGL(hop0, dim3(512), dim3(256), 0, 0, a); }, reps));
    printf("period, arguments -> store:                    %6.2f us\n", time_us([&] { hipLaunchKernelGGL(hop1, dim3(512), dim3(256), 0, 0, a); }, reps));
    printf("period, arguments -> load -> store:            %6.2f us\n", time_us([&] { hipLaunchKernelGGL(hop2, dim3(512), dim3(256), 0, 0, a); }, reps));
    printf("period, arguments -> load -> load -> store:    %6.2f us\n", time_us([&] { hipLaunchKernelGGL(hop3, dim3(512), dim3(256), 0, 0, a); }, reps));
    printf("period, arguments -> 3 dependent loads -> store: %6.2f us\n", time_us([&] { hipLaunchKernelGGL(hop4, dim3(512), dim3(256), 0, 0, a); }, reps));
    return 0;
}
