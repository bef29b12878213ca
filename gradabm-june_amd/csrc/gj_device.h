// Device-side helpers shared by the gfx950 kernels: wave64 reductions, Philox4x32-10, the
// per-agent Gumbel-softmax decision (a8) and the infection state update (a9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gj {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kThreads = 256;      // 4 waves per workgroup: one per SIMD of a CU
constexpr int kEdgesPerThread = 8; // STREAM block = 2048 edges = GJ_STREAM_EDGES

// Sum over the lanes of an aligned group of G lanes (G = 4, 16, 64); every lane gets the total
// of ITS group.  Butterfly order: run-to-run deterministic.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// ---- Philox4x32-10 (Salmon et al., SC'11), counter = (pair lo, pair hi, step lo, step hi), key = seed.
// One call yields the two Exponential(1) draws of each agent of a pair (see exp_pair).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#ifndef GJ_PHILOX_MULHI
  // one 32x32->64 product per word pair (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: half the quarter-rate
  // integer multiplies, which are what the sampler costs
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]);
  const uint32_t lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]);
  const uint32_t lo1 = 0xCD9E8D57u * c[2];
#endif
  const uint32_t n0 = hi1 ^ c[1] ^ k0;
  const uint32_t n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0;
  c[1] = lo1;
  c[2] = n2;
  c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key,
                                              uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi,
                   (uint32_t)(ctr_hi >> 32)};
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0];
  out[1] = c[1];
  out[2] = c[2];
  out[3] = c[3];
}

// 23 random bits -> uniform in (0,1): (k + 0.5) * 2^-23 for k < 2^23 is exactly representable in fp32 (24 significant
// bits), so the result is never 0 or 1 and -log(u) is finite and positive.  (With 24 bits the largest value,
// 16777215.5, rounds up to 2^24 and the draw becomes exactly 1: an Exponential draw of 0, once per ~1.7e7 draws -
// about once per step of a 10 M-agent world.)
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-7f; }

// ---- the library's own noise for a8 (no reference draw to reproduce) ---------------------------------------------
// The reference draws two Exponential(1) variates (e0, e1) per agent and infects iff log(1-p) - log(e1) > log(p) -
// log(e0), i.e. iff p < e0 / (e0 + e1).  For two iid exponentials theta = e0 / (e0 + e1) is Uniform(0,1) and
// INDEPENDENT of s = e0 + e1 ~ Gamma(2,1).  So:
//   forward : theta = one 23-bit uniform; infected iff p < theta.  One Philox4x32-10 block (four words) serves the FOUR
//             agents 4k .. 4k+3 of GLOBAL ids (counter = agent >> 2) - a quarter of the integer multiplies that are the
//             sampler's cost, no logarithm, no division.
//   backward: the straight-through softmax needs both draws: e0 = theta * s, e1 = (1 - theta) * s with s = -log(u1 u2)
//             from a second block (counter = agent >> 1, stream bit 62 set) - the same joint law as two iid draws,
//             and by construction the same decision as the forward's.
__device__ __forceinline__ float infection_uniform_from_block(const uint32_t (&r)[4], int which) {
  return u01(which == 0 ? r[0] : (which == 1 ? r[1] : (which == 2 ? r[2] : r[3])));
}
__device__ __forceinline__ float infection_uniform(uint64_t seed, uint64_t step, int64_t agent) {
  uint32_t r[4];
  philox4x32_10((uint64_t)agent >> 2, step, seed, r);
  return infection_uniform_from_block(r, (int)(agent & 3));
}
__device__ __forceinline__ float own_new_infected(float p, float theta) { return (p < theta) ? 1.0f : 0.0f; }
// (e0, e1) with e0 / (e0 + e1) == theta up to rounding: what k_adjoint_sample feeds the softmax derivative
__device__ __forceinline__ void exp_pair(uint64_t seed, uint64_t step, int64_t agent, float& e0, float& e1) {
  const float theta = infection_uniform(seed, step, agent);
  uint32_t r[4];
  philox4x32_10((uint64_t)agent >> 1, step | (1ull << 62), seed, r);
  const float u1 = u01((agent & 1) ? r[2] : r[0]), u2 = u01((agent & 1) ? r[3] : r[1]);
  const float s = -logf(u1) - logf(u2);
  e0 = theta * s;
  e1 = (1.0f - theta) * s;
}

// a8: IsInfectedSampler.forward = F.gumbel_softmax(vstack(p, 1-p).log(), tau=0.1, hard=True, dim=0)
// (reference grad_june/infection.py:13-18).  e0/e1 are the Exponential(1) draws of rows 0/1.
// Returns the forward value 1 - (y_hard - y_soft + y_soft)[0], computed op for op.
__device__ __forceinline__ float gumbel_new_infected(float p, float e0, float e1) {
  const float l0 = logf(p);
  const float l1 = logf(1.0f - p);
  const float z0 = (l0 + (-logf(e0))) / 0.1f;
  const float z1 = (l1 + (-logf(e1))) / 0.1f;
  const float m = fmaxf(z0, z1);
  const float x0 = expf(z0 - m);
  const float x1 = expf(z1 - m);
  const float s = x0 + x1;
  const float y0 = x0 / s;
  const float y1 = x1 / s;
  const float h0 = (y1 > y0) ? 0.0f : 1.0f;  // argmax, ties -> row 0 (not infected)
  const float ret0 = (h0 - y0) + y0;
  return 1.0f - ret0;
}

// a9: GradJune.infect_people (reference grad_june/model.py:103-110)
__device__ __forceinline__ void infect(float nw, float now, float& susc, float& inf, float& t_inf) {
  susc = fmaxf(0.0f, susc - nw);
  inf = inf + nw;
  t_inf = t_inf + nw * (now - t_inf);
}

// a7 (reference base.py:136-140): p = clamp(exp(-clamp(ts, 1e-6, 100) * dt), 0, 1).  torch.clamp passes a NaN
// through; fminf / fmaxf would swallow it (and turn a poisoned agent into "never infected").
__device__ __forceinline__ float clamp_keep_nan(float x, float lo, float hi) {
  return (x != x) ? x : fminf(fmaxf(x, lo), hi);
}
__device__ __forceinline__ float not_infected_prob(float ts, float dt) {
  return clamp_keep_nan(expf(-clamp_keep_nan(ts, 1e-6f, 100.0f) * dt), 0.0f, 1.0f);
}

__device__ __forceinline__ float sgnf(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

}  // namespace gj
