// Microbenchmark: streaming read bandwidth on gfx950 as a function of the bytes a lane loads per instruction, in the
// launch shape of the tiled phases: ONE 1024-thread workgroup per CU (160 KiB of LDS requested), 8 loads in flight per
// lane, every wave streaming its own contiguous range.
//   hipcc --offload-arch=gfx950 -O3 -o load_width load_width.hip && ./load_width
// Question it answers: are phases A and D, whose per-edge streams are 2- and 4-byte elements
// (one element per lane and instruction), bound by the rate at which a CU processes vector-memory instructions rather
// than by bytes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kThreads = 1024, kU = 8;

template <typename T>
__device__ inline float value_of(T v);
template <> __device__ inline float value_of<uint16_t>(uint16_t v) { return (float)v; }
template <> __device__ inline float value_of<uint32_t>(uint32_t v) { return (float)v; }
template <> __device__ inline float value_of<uint2>(uint2 v) { return (float)(v.x ^ v.y); }
template <> __device__ inline float value_of<uint4>(uint4 v) { return (float)(v.x ^ v.y ^ v.z ^ v.w); }

// every wave reads a contiguous range of `per_wave` elements, 64 consecutive elements per instruction
template <typename T>
__global__ __launch_bounds__(kThreads) void k_stream(const T* __restrict__ src, int64_t per_wave, float* out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x % 64;
  const int64_t wave = (int64_t)blockIdx.x * (kThreads / 64) + threadIdx.x / 64;
  const T* p = src + wave * per_wave;
  float acc = 0.f;
  for (int64_t i = 0; i + kU * 64 <= per_wave; i += kU * 64) {
    T v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) v[u] = p[i + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < kU; ++u) acc += value_of<T>(v[u]);
  }
  lds[threadIdx.x] = acc;
  __syncthreads();
  out[(int64_t)blockIdx.x * kThreads + threadIdx.x] = lds[threadIdx.x ^ 1];
}

template <typename T>
void run(const char* name, const void* src, size_t bytes, float* out, int wgs_per_cu) {
  const int grid = 256 * wgs_per_cu;
  const int64_t per_wave = (int64_t)(bytes / sizeof(T)) / ((int64_t)grid * (kThreads / 64)) / (kU * 64) * (kU * 64);
  const size_t lds = wgs_per_cu == 1 ? 160 * 1024 : 64 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(k_stream<T>, dim3(grid), dim3(kThreads), lds, 0, (const T*)src, per_wave, out);
  hipEventRecord(a);
  const int reps = 5;
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_stream<T>, dim3(grid), dim3(kThreads), lds, 0, (const T*)src, per_wave, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double moved = (double)per_wave * sizeof(T) * grid * (kThreads / 64);
  printf("%-28s %d wg/CU  %7.1f us  %6.2f TB/s  (%.0f MB)\n", name, wgs_per_cu, 1e3 * ms / reps,
         moved / (1e-3 * ms / reps) / 1e12, moved / 1e6);
}

int main() {
  const size_t bytes = (size_t)1 << 30;      // 1 GiB: far beyond the Infinity Cache
  void* src;
  float* out;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc(&out, sizeof(float) * 1024 * 512);
  for (int w = 1; w <= 2; ++w) {
    run<uint16_t>("2 bytes per lane (ushort)", src, bytes / 8, out, w);
    run<uint32_t>("4 bytes per lane (dword)", src, bytes / 4, out, w);
    run<uint2>("8 bytes per lane (dwordx2)", src, bytes / 2, out, w);
    run<uint4>("16 bytes per lane (dwordx4)", src, bytes, out, w);
  }
  hipFree(src);
  hipFree(out);
  return 0;
}
