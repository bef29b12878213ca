/* Calling libgradjune_hip.so from plain C through include/gradjune_hip.h - no Python, no torch.
 *
 *   gcc examples/abi_demo.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       -Lgradabm-june_amd/grad_june_amd/lib -lgradjune_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/gradabm-june_amd/grad_june_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/abi_demo && /tmp/abi_demo
 *
 * Rows a8+a9 (IsInfectedSampler + infect_people) on one million agents with p(not infected) = 0.7, then the
 * per-step result reduction (row f2).  Every buffer is a caller-owned device pointer; calls are asynchronous
 * on the stream passed in; return codes are checked with gj_error_string().
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "gradjune_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_GJ(x) do { int rc_ = (x); if (rc_ != GJ_OK) { fprintf(stderr, "%s: [%d] %s\n", #x, rc_, gj_error_string(rc_)); return 1; } } while (0)

int main(void) {
  const int64_t n = 1000000;
  printf("libgradjune_hip ABI version %d\n", gj_version());
  float *h = (float*)malloc(n * sizeof(float));
  unsigned char* hc = (unsigned char*)malloc(n);
  float *p_not, *susc, *inf, *t_inf, *new_inf, *stage;
  unsigned char* cls;
  double* out;
  CHECK_HIP(hipMalloc((void**)&p_not, n * 4)); CHECK_HIP(hipMalloc((void**)&susc, n * 4));
  CHECK_HIP(hipMalloc((void**)&inf, n * 4));   CHECK_HIP(hipMalloc((void**)&t_inf, n * 4));
  CHECK_HIP(hipMalloc((void**)&new_inf, n * 4)); CHECK_HIP(hipMalloc((void**)&stage, n * 4));
  CHECK_HIP(hipMalloc((void**)&cls, n)); CHECK_HIP(hipMalloc((void**)&out, 5 * sizeof(double)));
  for (int64_t i = 0; i < n; ++i) h[i] = 0.7f;
  CHECK_HIP(hipMemcpy(p_not, h, n * 4, hipMemcpyHostToDevice));
  for (int64_t i = 0; i < n; ++i) h[i] = 1.0f;
  CHECK_HIP(hipMemcpy(susc, h, n * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(stage, h, n * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemset(inf, 0, n * 4)); CHECK_HIP(hipMemset(t_inf, 0, n * 4)); CHECK_HIP(hipMemset(out, 0, 5 * sizeof(double)));
  for (int64_t i = 0; i < n; ++i) hc[i] = (unsigned char)(i % 100);          /* age = i % 100, sex 0 */
  CHECK_HIP(hipMemcpy(cls, hc, n, hipMemcpyHostToDevice));

  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  /* a8 + a9: Philox noise keyed by (seed 42, step 0, agent id) */
  CHECK_GJ(gj_sample_infect(n, p_not, NULL, 42, 0, 0, 3.0f, new_inf, susc, inf, t_inf, stream));
  /* f2: cases, cases by age bin (0,18), (18,65), (65,100), deaths */
  const int32_t edges[4] = {0, 18, 65, 100};
  CHECK_GJ(gj_step_stats(n, cls, inf, stage, 3, edges, 7, out, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  double res[5];
  CHECK_HIP(hipMemcpy(res, out, sizeof(res), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h, t_inf, n * 4, hipMemcpyDeviceToHost));
  printf("infected %.0f of %lld (expected ~30%%): by age bin %.0f / %.0f / %.0f, deaths %.0f\n", res[0], (long long)n,
         res[1], res[2], res[3], res[4]);
  const double frac = res[0] / (double)n;
  if (frac < 0.295 || frac > 0.305) { fprintf(stderr, "unexpected infected fraction %f\n", frac); return 1; }
  /* argument errors never touch the device */
  if (gj_sample_infect(-1, p_not, NULL, 0, 0, 0, 0.f, new_inf, susc, inf, t_inf, stream) != GJ_E_RANGE) return 1;
  printf("ok\n");
  return 0;
}
