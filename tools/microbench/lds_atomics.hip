// Microbenchmark: rate of random-address LDS operations on gfx950, one 1024-thread workgroup per CU.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kN = 32768;      // LDS words touched
constexpr int kIters = 256;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ idx, float* out, int iters) {
  __shared__ unsigned long long lds64[kN / 2];              // 128 KiB, viewed as fp32 or u64
  float* lds = reinterpret_cast<float*>(lds64);
  for (int i = threadIdx.x; i < kN; i += 1024) lds[i] = 0.f;
  __syncthreads();
  float acc = 0.f;
  uint32_t a = idx[blockIdx.x * 1024 + threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    a = a * 1664525u + 1013904223u;
    const uint32_t j = (a >> 8) % kN;
    if (MODE == 0) atomicAdd(&lds[j], 1.0f);                                   // ds_add_f32
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&lds[j]), 3u);        // ds_add_u32
    if (MODE == 2) atomicAdd(&lds64[j >> 1], 3ull);                            // ds_add_u64
    if (MODE == 3) acc += lds[j];                                              // ds_read_b32
    if (MODE == 4) lds[j] = acc + it;                                          // ds_write_b32
    if (MODE == 5) atomicAdd(&lds[(j & ~63u) + (threadIdx.x & 63)], 1.0f);     // conflict-free f32 atomics
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = acc + lds[threadIdx.x] + (MODE == 2 ? (float)lds64[threadIdx.x % (kN / 2)] : 0.f);
}

template <int MODE>
void run(const char* name, const uint32_t* idx, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, idx, out, kIters);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, idx, out, kIters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double ops = 256.0 * 1024 * kIters;
  printf("%-28s %.3f ms  %.1f G lane-ops/s  = %.2f lanes/clk/CU @2.4GHz\n", name, ms, ops / ms / 1e6,
         ops / 256 / (ms * 1e-3) / 2.4e9);
}

int main() {
  uint32_t* idx; float* out;
  hipMalloc(&idx, 256 * 1024 * 4); hipMalloc(&out, 256 * 1024 * 4);
  std::vector<uint32_t> h(256 * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u);
  hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>("ds_add_f32 random", idx, out);
  run<5>("ds_add_f32 conflict-free", idx, out);
  run<1>("ds_add_u32 random", idx, out);
  run<2>("ds_add_u64 random", idx, out);
  run<3>("ds_read_b32 random", idx, out);
  run<4>("ds_write_b32 random", idx, out);
  return 0;
}
