// gfx950 (MI355X / CDNA4) kernels + C ABI for the per-timestep infection message-passing path
// of GradABM-JUNE.  ABI: include/gradjune_hip.h.  Reference semantics: SURVEY.md section 8a.
//
// Per step (gj_step): three dependent launches on the caller's stream
//   k_transmission    a1+a2   elementwise over agents           (HBM-bound, streaming)
//   k_venue_reduce    a3-a5   segmented sum per venue (pass 1)   (HBM/MALL-bound, index stream + gather)
//   [k_combine_long]          ordered combine of >2048-edge venues' partial sums
//   k_agent_gather    a4,a6-a9 per-agent gather over its venues + epilogue + Gumbel decision
//                              + state update (pass 2, fused)
// The path is a sparse segmented reduction: no MFMA.  The rules that matter are coalesced index
// streams, many independent gathers in flight, LDS staging of per-venue partial sums and wave64
// shuffles for the segmented combine.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gradjune_hip.h"
#include "gj_device.h"
#include "gj_tiled.h"

namespace gj {

// ------------------------------------------------------------------------------------------
// kernel-argument blocks (passed by value; < 4 KiB)
// ------------------------------------------------------------------------------------------
struct SetP1 {
  const int32_t* v_rowptr;
  const int32_t* v_agent;
  const float* v_pc;
  float* cum;
  int32_t stride;
  int32_t nk;                              // networks active on this set in this step
  float beta[GJ_MAX_NETS_PER_SET];
  int32_t table[GJ_MAX_NETS_PER_SET];      // -1: no leisure table
  int32_t raw;                             // 1: household (raw transmission), 0: q*transmission
  int32_t leisure;                         // 1: any network of the set uses a table
};

struct P1Args {
  SetP1 sets[GJ_MAX_SETS];
  const gj_block* blocks;
  const float* trans;
  const float* qtrans;
  const uint8_t* cls;
  const float* tables;
  float* partial;
  int32_t day_type;
};

struct SetP2 {
  const int32_t* a_rowptr;
  const int32_t* a_venue;
  const float* cum;
  int32_t stride;
  int32_t nk;
  int32_t mask[GJ_MAX_NETS_PER_SET];
  int32_t table[GJ_MAX_NETS_PER_SET];
};

struct P2Args {
  SetP2 groups[GJ_MAX_SETS];   // active edge sets in accumulation order
  int32_t n_groups;
  int32_t any_leisure;
  int64_t n_agents;
  const uint8_t* cls;
  const float* tables;
  const float* stage;
  float* susceptibility;
  float* is_infected;
  float* infection_time;
  float* not_infected_probs;
  float* new_infected;
  float* trans_susc;
  const float* exp_noise;
  float now, dt, q_thr;
  int32_t day_type, has_q, sample;
  uint64_t seed, step;
  int64_t agent_offset;
  const gj_clock* clock;
};

// ------------------------------------------------------------------------------------------
// a1 + a2   transmission profile and quarantine-masked copy
//   reference: grad_june/transmission.py:39-51, grad_june/policies/quarantine_policies.py:13-33
// ------------------------------------------------------------------------------------------
// The profile's transcendental part, exp(-lgamma(shape)) * pow(x, shape - 1) * exp(e), costs ~690 instructions with
// libm's lgammaf (~420) and powf (~200) - and a wave pays them for all 64 lanes whenever one lane holds an infected
// agent, which made k_transmission arithmetic-bound at the 5-25 % prevalence of the timed steps.  Round 3:
//   1 / Gamma(shape)  by the recurrence to [1, 2] and Abramowitz & Stegun 6.1.36 (degree 8, |eps| <= 3e-7 there; measured
//                     3.5e-7 relative in fp32), libm only for shapes outside (0.25, 16);
//   pow(x, y)         = exp2(y * log2(x)) on the hardware's v_log_f32 / v_exp_f32 (1 ulp each): the error of the
//                     exponent, |y log2 x| * 2^-23, is ~1e-6 relative in the result; 0 / inf at x == 0 as powf; x < 0: NaN for a
//                     non-integer y and +-|x|^y for an integer one, as powf / torch.pow;
//   exp(e)            = exp2(e * log2(e)): ~|e| * 1e-7.
// Together ~2e-6 relative against the reference's fp32 torch ops (themselves ~1e-7); the parity tests hold the
// transmissions to 2e-5.  ~60 instructions.
__device__ __forceinline__ float inv_gamma(float x) {
  if (!(x > 0.25f && x < 16.0f)) return expf(-lgammaf(x));        // (also NaN)
  float up = 1.0f, down = 1.0f;           // Gamma(x_in) = Gamma(x) * up / down
  while (x > 2.0f) {
    x -= 1.0f;
    up *= x;
  }
  while (x < 1.0f) {
    down *= x;
    x += 1.0f;
  }
  const float z = x - 1.0f;
  float g = 0.035868343f;
  g = g * z - 0.193527818f;
  g = g * z + 0.482199394f;
  g = g * z - 0.756704078f;
  g = g * z + 0.918206857f;
  g = g * z - 0.897056937f;
  g = g * z + 0.988205891f;
  g = g * z - 0.577191652f;
  g = g * z + 1.0f;                       // Gamma(1 + z), 0 <= z <= 1
  return down / (g * up);
}
__device__ __forceinline__ float fast_pow(float x, float y) {
  if (y == 0.0f) return 1.0f;             // powf(x, 0) == 1 for every x
  if (x < 0.0f) {
    // torch.pow / powf of a negative base: finite for an INTEGER exponent (sign by its parity), NaN otherwise.  A
    // constant integer shape with t < shift is the case that matters (transmission.py:45-49: sign == 0 there, and
    // 0 * finite == 0 where 0 * NaN would poison the venue sums).  Rare: not worth libm's powf in the
    // instruction stream of a launch that lives on its occupancy.
    const float yi = truncf(y);
    if (yi != y) return __builtin_nanf("");
    const float r = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(-x));
    const float h = 0.5f * yi;
    return (truncf(h) != h) ? -r : r;     // odd exponent: the base's sign survives
  }
  return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));      // v_exp_f32(y * v_log_f32(x))
}
__device__ __forceinline__ float fast_exp(float e) { return __builtin_amdgcn_exp2f(e * 1.44269504088896341f); }

__device__ __forceinline__ float transmission_value(float mx, float shp, float rt, float sh, float t_inf,
                                                    float inf, float now) {
  const float t = now - t_inf;
  const float d = t - sh;
  const float sign = (sgnf(d + 1e-10f) + 1.0f) / 2.0f;
  const float aux = inv_gamma(shp) * fast_pow(d * rt, shp - 1.0f);
  const float aux2 = fast_exp((sh - t) * rt) * rt;
  return mx * sign * aux * aux2 * inf;
}

// (Round 3, measured and not adopted: listing a chunk's infected agents in LDS and evaluating them densely, one per
// lane, from scattered parameter loads - a tenth of the arithmetic at 3 % prevalence, but the list can only be worked
// off behind a barrier and a second, uncoalesced round trip: 110 us against 70 on C3, whose timed steps run at 5-25 %
// prevalence.)
__global__ __launch_bounds__(kThreads) void k_transmission(
    int64_t n, const float* __restrict__ mx, const float* __restrict__ shp, const float* __restrict__ rt,
    const float* __restrict__ sh, const float* __restrict__ t_inf, const float* __restrict__ inf,
    const float* __restrict__ stage, float* __restrict__ trans, float* __restrict__ qtrans, float now_arg,
    int has_q, float q_thr, const gj_clock* __restrict__ clock) {
  const float now = clock ? clock->now : now_arg;      // device clock: a captured step replayed for later timesteps
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 f = load_nt(reinterpret_cast<const float4*>(inf) + i);      // (non-temporal: read once per step, see gj_tiled.h)
    float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // is_infected == 0 makes the product 0 whatever the profile (finite for every agent the reference gives a
    // finite value for), so the five parameter streams are only read where someone is infected: early in an
    // epidemic most 128-byte lines of them are never touched
    unsigned m = (f.x != 0.0f ? 1u : 0u) | (f.y != 0.0f ? 2u : 0u) | (f.z != 0.0f ? 4u : 0u) | (f.w != 0.0f ? 8u : 0u);
    if (m) {
      const float4 a = load_nt(reinterpret_cast<const float4*>(mx) + i);
      const float4 b = load_nt(reinterpret_cast<const float4*>(shp) + i);
      const float4 c = load_nt(reinterpret_cast<const float4*>(rt) + i);
      const float4 d = load_nt(reinterpret_cast<const float4*>(sh) + i);
      const float4 e = load_nt(reinterpret_cast<const float4*>(t_inf) + i);
      // The profile (~100 instructions since round 3, ~690 with libm - see inv_gamma) is evaluated by a wave for all 64
      // lanes whenever one of them needs it.  Each lane therefore takes ITS infected agents one after the other: a wave makes
      // as many evaluations as its busiest lane has infected agents (at 1 % prevalence ~1 instead of ~2 with one
      // evaluation per component, at 30 % 3.4 of 4): 67 -> 62 us on C3.  (Measured, not adopted: exp(-lgamma(shape)),
      // 60 % of those instructions and a per-agent constant, cached in a sixth parameter array - 6 us SLOWER: with
      // the evaluations compacted the launch is bound by its reads again, and the cache adds 4 bytes per agent.)
      while (m) {
        const int k = __builtin_ctz(m);
        m &= m - 1u;
#define GJ_PICK(v) (k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w)
        const float val = transmission_value(GJ_PICK(a), GJ_PICK(b), GJ_PICK(c), GJ_PICK(d), GJ_PICK(e), GJ_PICK(f), now);
#undef GJ_PICK
        r.x = k == 0 ? val : r.x;
        r.y = k == 1 ? val : r.y;
        r.z = k == 2 ? val : r.z;
        r.w = k == 3 ? val : r.w;
      }
    }
    reinterpret_cast<float4*>(trans)[i] = r;
    if (has_q) {
      const float4 s = reinterpret_cast<const float4*>(stage)[i];
      float4 q;
      q.x = (s.x < q_thr ? 1.0f : 0.0f) * r.x;
      q.y = (s.y < q_thr ? 1.0f : 0.0f) * r.y;
      q.z = (s.z < q_thr ? 1.0f : 0.0f) * r.z;
      q.w = (s.w < q_thr ? 1.0f : 0.0f) * r.w;
      reinterpret_cast<float4*>(qtrans)[i] = q;
    }
  }
  // tail (n % 4 agents): first threads of block 0
  if (blockIdx.x == 0) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    if (threadIdx.x < (n & 3)) {
      const float r = inf[i] != 0.0f ? transmission_value(mx[i], shp[i], rt[i], sh[i], t_inf[i], inf[i], now) : 0.0f;
      trans[i] = r;
      if (has_q) qtrans[i] = (stage[i] < q_thr ? 1.0f : 0.0f) * r;
    }
  }
}

// ------------------------------------------------------------------------------------------
// a3 + a4 + a5   pass 1: cum_n[v] = sum_{e in venue v} trans_n[agent(e)] * (beta_n * p_contact[v])
//   reference: grad_june/infection_networks/base.py:61-79,86-87 (+ leisure_network.py:61-72)
//
// One workgroup per schedule entry (gj_block).
//  STREAM: the block owns whole venues [v0,v1) with <= 2048 edges in total.  All 256 threads
//          stream the block's contiguous edge range with coalesced index loads, 8 independent
//          gathers in flight per thread, and stage the gathered values in LDS; then `lanes`
//          lanes per venue sum each venue's LDS segment (lanes == 1: in edge order, i.e. the
//          reference's scatter_add_ order) and combine with wave64 shuffles.
//  LONG:   the block owns edges [e0,e1) of ONE venue; register accumulation, wave shuffles,
//          cross-wave combine through LDS, partial sum to plan.partial[slot].
// ------------------------------------------------------------------------------------------
template <bool LEISURE>
__device__ __forceinline__ float gather_x(const P1Args& A, const float* __restrict__ src, int32_t a,
                                          const float* __restrict__ tab) {
  float x = src[a];
  if (LEISURE) x = tab[A.cls[a]] * x;   // (q*L)*t == L*(q*t) exactly for q in {0,1}
  return x;
}

template <int G>
__device__ __forceinline__ void stream_reduce(const SetP1& S, const gj_block& b, int k, const float* lds,
                                              int tid) {
  const int R = b.v1 - b.v0;
  const int grp = tid / G, lane = tid % G;
  constexpr int kGroups = kThreads / G;
  const float beta = S.beta[k];
  for (int r = grp; r < R; r += kGroups) {
    const int v = b.v0 + r;
    const int s = S.v_rowptr[v] - b.e0;
    const int t = S.v_rowptr[v + 1] - b.e0;
    const float y = beta * S.v_pc[v];
    float sum = 0.0f;
    for (int i = s + lane; i < t; i += G) sum += lds[i] * y;
    if (G > 1) sum = group_sum<G>(sum);
    if (lane == 0) S.cum[(int64_t)v * S.stride + k] = sum;
  }
}

__global__ __launch_bounds__(kThreads) void k_venue_reduce(const P1Args A) {
  __shared__ float lds[GJ_STREAM_EDGES];
  __shared__ float wsum[kThreads / kWave][GJ_MAX_NETS_PER_SET];
  const gj_block b = A.blocks[blockIdx.x];
  const SetP1& S = A.sets[b.set];
  const int nk = S.nk;
  if (nk == 0) return;  // set not active in this step (block-uniform)
  const int tid = threadIdx.x;
  const float* __restrict__ src = S.raw ? A.trans : A.qtrans;

  if (b.kind == 0) {
    const int ne = b.e1 - b.e0;
    for (int k = 0; k < nk; ++k) {
      const float* tab = S.leisure ? A.tables + (int64_t)S.table[k] * GJ_TABLE_SIZE + A.day_type * 200 : nullptr;
      // stage: coalesced index stream, kEdgesPerThread independent gathers per thread
      int32_t ag[kEdgesPerThread];
#pragma unroll
      for (int j = 0; j < kEdgesPerThread; ++j) {
        const int i = j * kThreads + tid;
        ag[j] = (i < ne) ? S.v_agent[b.e0 + i] : -1;
      }
#pragma unroll
      for (int j = 0; j < kEdgesPerThread; ++j) {
        const int i = j * kThreads + tid;
        if (ag[j] >= 0) lds[i] = S.leisure ? gather_x<true>(A, src, ag[j], tab) : gather_x<false>(A, src, ag[j], tab);
      }
      __syncthreads();
      switch (b.lanes) {
        case 1: stream_reduce<1>(S, b, k, lds, tid); break;
        case 4: stream_reduce<4>(S, b, k, lds, tid); break;
        case 16: stream_reduce<16>(S, b, k, lds, tid); break;
        default: stream_reduce<64>(S, b, k, lds, tid); break;
      }
      if (k + 1 < nk) __syncthreads();
    }
    return;
  }

  // LONG: one venue, edges [e0,e1)
  float acc[GJ_MAX_NETS_PER_SET];
  float y[GJ_MAX_NETS_PER_SET];
  const float pc = S.v_pc[b.v0];
#pragma unroll
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
    acc[k] = 0.0f;
    y[k] = (k < nk) ? S.beta[k] * pc : 0.0f;
  }
  if (!S.leisure) {
    float a0 = 0.0f;
    int e = b.e0 + tid;
    for (; e + 3 * kThreads < b.e1; e += 4 * kThreads) {
      const int32_t i0 = S.v_agent[e], i1 = S.v_agent[e + kThreads], i2 = S.v_agent[e + 2 * kThreads],
                    i3 = S.v_agent[e + 3 * kThreads];
      const float x0 = src[i0], x1 = src[i1], x2 = src[i2], x3 = src[i3];
      a0 += x0 * y[0];
      a0 += x1 * y[0];
      a0 += x2 * y[0];
      a0 += x3 * y[0];
    }
    for (; e < b.e1; e += kThreads) a0 += src[S.v_agent[e]] * y[0];
    acc[0] = a0;
  } else {
    const float* tabs = A.tables + A.day_type * 200;
    for (int e = b.e0 + tid; e < b.e1; e += kThreads) {
      const int32_t a = S.v_agent[e];
      const float x = src[a];
      const int c = A.cls[a];
#pragma unroll
      for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k)
        if (k < nk) acc[k] += (tabs[(int64_t)S.table[k] * GJ_TABLE_SIZE + c] * x) * y[k];
    }
  }
  const int wave = tid / kWave, lane = tid % kWave;
#pragma unroll
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
    if (k < nk) {
      const float w = group_sum<kWave>(acc[k]);
      if (lane == 0) wsum[wave][k] = w;
    }
  }
  __syncthreads();
  if (tid < nk) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) s += wsum[w][tid];
    A.partial[(int64_t)b.slot * GJ_MAX_NETS_PER_SET + tid] = s;
  }
}

// ordered combine of the partial sums of one long venue (deterministic: chunk order)
struct CombineArgs {
  struct { float* cum; int32_t stride; int32_t nk; } sets[GJ_MAX_SETS];
  const gj_long_row* rows;
  const float* partial;
  int32_t n_rows;
};

__global__ __launch_bounds__(kThreads) void k_combine_long(const CombineArgs C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = i / GJ_MAX_NETS_PER_SET, k = i % GJ_MAX_NETS_PER_SET;
  if (row >= C.n_rows) return;
  const gj_long_row r = C.rows[row];
  const auto& S = C.sets[r.set];
  if (k >= S.nk) return;
  float s = 0.0f;
  for (int slot = r.slot0; slot < r.slot1; ++slot) s += C.partial[(int64_t)slot * GJ_MAX_NETS_PER_SET + k];
  S.cum[(int64_t)r.v * S.stride + k] = s;
}

// ------------------------------------------------------------------------------------------
// a4 + a6 + a7 (+ a8 + a9)   pass 2, one thread per agent:
//   ts[a] = sum_n  sum_{e in row a of set(n)} cum_n[venue(e)] * susc_n[a]      (network order = params order)
//   p = clamp(exp(-clamp(ts,1e-6,100) * dt), 0, 1);  Gumbel decision;  state update
//   reference: base.py:80-83,118-141; leisure_network.py:74-85,107-120; infection.py:13-18; model.py:103-110
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_agent_gather(const P2Args P) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= P.n_agents) return;
  float susc = P.susceptibility[a];
  float q = 1.0f;
  if (P.has_q) q = (P.stage[a] < P.q_thr) ? 1.0f : 0.0f;
  int c = 0;
  if (P.any_leisure) c = P.cls[a];
  const float* tabs = P.tables + P.day_type * 200;

  float ts = 0.0f;
  for (int g = 0; g < P.n_groups; ++g) {
    const SetP2& S = P.groups[g];
    const int r0 = S.a_rowptr[a], r1 = S.a_rowptr[a + 1];
    if (S.nk == 1 && S.mask[0] <= GJ_MASK_Q) {
      const float w = (S.mask[0] == GJ_MASK_RAW) ? susc : q * susc;
      float acc = 0.0f;
      for (int e = r0; e < r1; ++e) acc += S.cum[(int64_t)S.a_venue[e] * S.stride] * w;
      ts += acc;
    } else {
      float w[GJ_MAX_NETS_PER_SET], acc[GJ_MAX_NETS_PER_SET];
#pragma unroll
      for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
        acc[k] = 0.0f;
        w[k] = 0.0f;
        if (k < S.nk) {
          const int m = S.mask[k];
          float wk = (m == GJ_MASK_RAW) ? susc : q;
          if (m >= GJ_MASK_QL) wk = wk * tabs[(int64_t)S.table[k] * GJ_TABLE_SIZE + c];
          if (m != GJ_MASK_RAW) wk = wk * susc;
          if (m == GJ_MASK_QL_AGE75) wk = wk * (((c % 100) > 75) ? 1.0f : 0.0f);
          w[k] = wk;
        }
      }
      for (int e = r0; e < r1; ++e) {
        const float* cv = S.cum + (int64_t)S.a_venue[e] * S.stride;
#pragma unroll
        for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k)
          if (k < S.nk) acc[k] += cv[k] * w[k];
      }
#pragma unroll
      for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k)
        if (k < S.nk) ts += acc[k];
    }
  }
  if (P.trans_susc) P.trans_susc[a] = ts;
  const float p = not_infected_prob(ts, P.dt);
  if (P.not_infected_probs) P.not_infected_probs[a] = p;
  if (!P.sample) return;

  float e0, e1;
  if (P.exp_noise) {
    e0 = P.exp_noise[a];
    e1 = P.exp_noise[P.n_agents + a];
  } else {
    e0 = e1 = 1.0f;
  }
  const float nw = P.exp_noise ? gumbel_new_infected(p, e0, e1)
                               : own_new_infected(p, infection_uniform(P.seed, P.clock ? P.clock->step : P.step,
                                                                       P.agent_offset + a));
  if (P.new_infected) P.new_infected[a] = nw;
  if (nw != 0.0f) {   // unchanged values are not rewritten
    float inf = P.is_infected[a], t_inf = P.infection_time[a];
    infect(nw, P.clock ? P.clock->now : P.now, susc, inf, t_inf);
    P.susceptibility[a] = susc;
    P.is_infected[a] = inf;
    P.infection_time[a] = t_inf;
  }
}

// a8 + a9 on a caller-supplied probability vector
__global__ __launch_bounds__(kThreads) void k_sample_infect(int64_t n, const float* __restrict__ p_not,
                                                            const float* __restrict__ noise, uint64_t seed,
                                                            uint64_t step, int64_t agent_offset, float now,
                                                            float* __restrict__ new_inf, float* __restrict__ susc,
                                                            float* __restrict__ inf, float* __restrict__ t_inf) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  float e0, e1;
  if (noise) {
    e0 = noise[a];
    e1 = noise[n + a];
  } else {
    e0 = e1 = 1.0f;
  }
  const float nw = noise ? gumbel_new_infected(p_not[a], e0, e1)
                         : own_new_infected(p_not[a], infection_uniform(seed, step, agent_offset + a));
  if (new_inf) new_inf[a] = nw;
  if (susc != nullptr && nw != 0.0f) {   // state pointers NULL: sample only
    float s = susc[a], i = inf[a], t = t_inf[a];
    infect(nw, now, s, i, t);
    susc[a] = s;
    inf[a] = i;
    t_inf[a] = t;
  }
}

// f1: disease-stage progression (reference grad_june/symptoms.py:204-247, 82-128), one lane per agent
struct SymptomsArgs {
  gj_symptoms_params P;
  int64_t n;
  const uint8_t* cls;
  const float* new_inf;
  float* cur;
  float* nxt;
  float* ttn;
  const float* progresses;
  const float* dwell;
};

__device__ __forceinline__ float dwell_sample(int kind, float loc, float scale, float z) {
  const float v = loc + scale * z;
  return kind == 1 ? __builtin_amdgcn_exp2f(v * 1.44269504088896341f) : v;      // LogNormal: exp on v_exp_f32
}
// The library's own draw for an agent at stage s that is due (no reference stream to reproduce: symptoms.py:82-128 draws
// torch.bernoulli + rsample): progress with the table's probability, dwell time = LogNormal / Normal of one Box-Muller
// normal.  On the hardware's transcendental units (v_log_f32, v_sqrt_f32, v_cos_f32 - whose argument is in revolutions -
// v_exp_f32): the draws of a wave's due agents were half of the fused symptoms launch with libm's logf / cosf / expf.
// ONE definition, shared by the update and its adjoint (which must replay the same draw).
__device__ __forceinline__ void stage_draw(const gj_symptoms_params& P, int64_t a, int s, int age, bool& onward, float& d) {
  uint32_t r[4];
  philox4x32_10((uint64_t)(P.agent_offset + a), P.step | (1ull << 63), P.seed, r);
  onward = u01(r[0]) < P.progress[s * 100 + age];
  const float ln_u = __builtin_amdgcn_logf(u01(r[1])) * 0.693147180559945309f;        // ln(u) = log2(u) * ln(2)
  const float z = __builtin_amdgcn_sqrtf(-2.0f * ln_u) * __builtin_amdgcn_cosf(u01(r[2]));   // cos(2 pi u)
  d = onward ? dwell_sample(P.next_kind[s], P.next_loc[s], P.next_scale[s], z)
             : dwell_sample(P.rec_kind[s], P.rec_loc[s], P.rec_scale[s], z);
}

// One agent's stage update (symptoms.py:204-247, 82-128).  Returns true when any of the three values changed.
__device__ __forceinline__ bool symptoms_agent(const SymptomsArgs& S, int64_t a, float nw, int cls, float& cur,
                                               float& nx, float& tt) {
  const int n_stages = S.P.n_stages;
  const float time = S.P.time;
  const float cur0 = cur, nx0 = nx, tt0 = tt;
  nx = nx + nw * (2.0f - nx);                       // newly infected: next stage = exposed, due now
  tt = tt + nw * (time - tt);
  const bool moving = (time >= tt) && (cur < (float)(n_stages - 1));
  cur = cur - (cur - nx) * (moving ? 1.0f : 0.0f);
  int s = (int)cur;
  s = min(max(s, 0), n_stages - 1);
  const int age = cls % 100;
  if (moving && s >= 2 && s <= n_stages - 2 && cur == (float)s) {
    bool onward;
    float d;
    if (S.progresses) {
      onward = S.progresses[a] != 0.0f;
      d = S.dwell[a];
    } else {
      stage_draw(S.P, a, s, age, onward, d);
    }
    if (onward) {
      nx = nx + 1.0f;
    } else {
      nx = nx - nx;
    }
    tt = tt + d;
  }
  // (bit comparison: a NaN that stays a NaN has not changed)
  return __float_as_uint(cur) != __float_as_uint(cur0) || __float_as_uint(nx) != __float_as_uint(nx0) ||
         __float_as_uint(tt) != __float_as_uint(tt0);
}

__global__ __launch_bounds__(kThreads) void k_symptoms(const SymptomsArgs S) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= S.n) return;
  float cur = S.cur[a], nx = S.nxt[a], tt = S.ttn[a];
  symptoms_agent(S, a, S.new_inf[a], (int)S.cls[a], cur, nx, tt);
  S.cur[a] = cur;
  S.nxt[a] = nx;
  S.ttn[a] = tt;
}

// f3: adjoint of k_symptoms w.r.t. the stage values and new_infected (oracle/gj_oracle.py:adjoint_symptoms).
// Recomputes the branch the agent took from the PRE-step state and the same randomness.
struct SymptomsAdjointArgs {
  gj_symptoms_params P;
  int64_t n;
  const uint8_t* cls;
  const float* new_inf;
  const float* cur0;
  const float* nxt0;
  const float* ttn0;
  const float* progresses;
  const float* dwell;
  const float* g_cur;
  const float* g_nxt;
  const float* g_ttn;
  float* g_cur_in;
  float* g_nxt_in;
  float* g_ttn_in;
  float* g_new;
};

__global__ __launch_bounds__(kThreads) void k_adjoint_symptoms(const SymptomsAdjointArgs S) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= S.n) return;
  const int n_stages = S.P.n_stages;
  const float time = S.P.time;
  const float nw = S.new_inf[a];
  const float c0 = S.cur0[a], x0 = S.nxt0[a], t0 = S.ttn0[a];
  const float x1 = x0 + nw * (2.0f - x0);
  const float t1 = t0 + nw * (time - t0);
  const bool moving = (time >= t1) && (c0 < (float)(n_stages - 1));
  const float m = moving ? 1.0f : 0.0f;
  const float c1 = c0 - (c0 - x1) * m;
  int s = (int)c1;
  s = min(max(s, 0), n_stages - 1);
  float gc1 = S.g_cur ? S.g_cur[a] : 0.0f;
  float gx1 = S.g_nxt ? S.g_nxt[a] : 0.0f;
  const float gt1 = S.g_ttn ? S.g_ttn[a] : 0.0f;
  if (moving && s >= 2 && s <= n_stages - 2 && c1 == (float)s) {
    bool onward;
    float d;
    if (S.progresses) {
      onward = S.progresses[a] != 0.0f;
      d = S.dwell[a];
    } else {
      stage_draw(S.P, a, s, (int)(S.cls[a] % 100), onward, d);
    }
    gc1 += gt1 * d / (float)s;              // time += dwell * (current == s) * current / s  (either branch)
    if (onward) {
      gc1 += gx1 / (float)s;                // next += (current == s) * current / s
    } else {
      gc1 -= gx1 * x1 / (float)s;           // next -= next * (current == s) * current / s
      gx1 = 0.0f;
    }
  }
  gx1 += gc1 * m;                           // current -= (current - next) * moving
  S.g_cur_in[a] = gc1 * (1.0f - m);
  S.g_nxt_in[a] = gx1 * (1.0f - nw);        // next += new_infected * (2 - next)
  if (S.g_ttn_in) S.g_ttn_in[a] = gt1 * (1.0f - nw);   // time += new_infected * (now - time)
  S.g_new[a] = gx1 * (2.0f - x0) + gt1 * (time - t0);
}

// f3: elementwise adjoints (see include/gradjune_hip.h)
__global__ __launch_bounds__(kThreads) void k_adjoint_sample(
    int64_t n, const float* __restrict__ susc0, const float* __restrict__ time0, const float* __restrict__ acc,
    const float* __restrict__ noise, uint64_t seed, uint64_t step, int64_t agent_offset, float now, float dt,
    const float* __restrict__ g_susc, const float* __restrict__ g_inf, const float* __restrict__ g_time,
    const float* __restrict__ g_new, float* __restrict__ x_out, float* __restrict__ grad_susc,
    float* __restrict__ grad_time) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const float s0 = susc0[a];
  const float ac = acc[a];
  const float ts = s0 * ac;
  const bool inside = (ts >= 1e-6f) && (ts <= 100.0f);
  const float p = not_infected_prob(ts, dt);
  float e0, e1;
  if (noise) {
    e0 = noise[a];
    e1 = noise[n + a];
  } else {
    exp_pair(seed, step, agent_offset + a, e0, e1);
  }
  // forward of the straight-through sampler (same op sequence as gumbel_new_infected)
  const float z0 = (logf(p) + (-logf(e0))) / 0.1f;
  const float z1 = (logf(1.0f - p) + (-logf(e1))) / 0.1f;
  const float m = fmaxf(z0, z1);
  const float x0 = expf(z0 - m), x1 = expf(z1 - m);
  const float y0 = x0 / (x0 + x1), y1 = x1 / (x0 + x1);
  const float nu = noise ? ((y1 > y0) ? 1.0f : 0.0f)                                   // the forward's rule
                         : own_new_infected(p, infection_uniform(seed, step, agent_offset + a));
  const float gs = g_susc ? g_susc[a] : 0.0f, gi = g_inf ? g_inf[a] : 0.0f, gt = g_time ? g_time[a] : 0.0f;
  const float gn = g_new ? g_new[a] : 0.0f;
  const float x = s0 - nu;                                   // torch.maximum(0, x): tie splits the gradient
  const float h = (x > 0.0f) ? 1.0f : ((x == 0.0f) ? 0.5f : 0.0f);
  const float nu_bar = gi + gt * (now - time0[a]) - gs * h + gn;
  float dnu_dp = -(y0 * y1 / 0.1f) * (1.0f / p + 1.0f / (1.0f - p));
  if (!(fabsf(dnu_dp) < 3.0e38f)) dnu_dp = 0.0f;             // p == 0 or 1: y0*y1 == 0 there
  const float ts_bar = inside ? nu_bar * dnu_dp * (-dt * p) : 0.0f;
  x_out[a] = s0 * ts_bar;
  grad_susc[a] = gs * h + ts_bar * ac;
  grad_time[a] = gt * (1.0f - nu);
}

__global__ __launch_bounds__(kThreads) void k_adjoint_transmission(
    int64_t n, const float* __restrict__ mx, const float* __restrict__ shp, const float* __restrict__ rt,
    const float* __restrict__ sh, const float* __restrict__ time0, const float* __restrict__ inf0, float now,
    const float* __restrict__ trans_bar, const float* __restrict__ g_inf, float* __restrict__ grad_inf,
    float* __restrict__ grad_time) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const float tb = trans_bar[a];
  const float t = now - time0[a];
  const float d = t - sh[a];
  const float sign = (sgnf(d + 1e-10f) + 1.0f) / 2.0f;
  const float aux = inv_gamma(shp[a]) * fast_pow(d * rt[a], shp[a] - 1.0f);
  const float aux2 = fast_exp((sh[a] - t) * rt[a]) * rt[a];
  const float base = mx[a] * sign * aux * aux2;              // d trans / d is_infected
  const float inf = inf0[a];
  float dtdt = 0.0f;                                         // d trans / d t
  if (inf != 0.0f) dtdt = base * inf * ((shp[a] - 1.0f) / d - rt[a]);
  grad_inf[a] = (g_inf ? g_inf[a] : 0.0f) + tb * base;
  grad_time[a] = grad_time[a] - tb * dtdt;                   // d t / d infection_time = -1
}

// f3: d loss / d log_beta of the networks on one edge set (include/gradjune_hip.h, gj_adjoint_beta_*)
struct AdjBetaArgs {
  int64_t n_venues;
  int32_t stride, nk;
  const float* cum_fwd;
  const float* cum_bwd;
  const float* v_pc;
  const double* weights;
  float beta[GJ_MAX_NETS_PER_SET];
  int32_t cols[GJ_MAX_NETS_PER_SET];
  double* partial;
};

// (1024 lanes per workgroup, two venues' loads in flight per lane: a lane's terms are added in venue order, one after the
// other in fp64, so its loop is a chain of memory round trips - 6 M households over 256 x 256 lanes were 92 of them)
constexpr int kAdjBetaThreads = 1024;
__global__ __launch_bounds__(kAdjBetaThreads) void k_adjoint_beta_partial(const AdjBetaArgs A) {
  __shared__ double part[kAdjBetaThreads / kWave][GJ_MAX_NETS_PER_SET];
  double acc[GJ_MAX_NETS_PER_SET];
#pragma unroll
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) acc[k] = 0.0;
  const int64_t step = (int64_t)gridDim.x * blockDim.x;
  auto term = [&](double pc, double w, const float (&f)[GJ_MAX_NETS_PER_SET], const float (&b)[GJ_MAX_NETS_PER_SET]) {
    if (!(pc > 0.0)) return;
#pragma unroll
    for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
      if (k >= A.nk) break;
      const double beta = (double)A.beta[k];
      if (beta == 0.0) continue;
      acc[k] += (double)f[k] * (double)b[k] / (beta * pc) * w;
    }
  };
  for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < A.n_venues; v0 += 2 * step) {
    const int64_t v1 = v0 + step;
    const bool two = v1 < A.n_venues;
    const int64_t u1 = two ? v1 : v0;
    const double pc0 = (double)A.v_pc[v0], pc1 = (double)A.v_pc[u1];
    const double w0 = A.weights ? A.weights[v0] : 1.0, w1 = A.weights ? A.weights[u1] : 1.0;
    float f0[GJ_MAX_NETS_PER_SET], b0[GJ_MAX_NETS_PER_SET], f1[GJ_MAX_NETS_PER_SET], b1[GJ_MAX_NETS_PER_SET];
#pragma unroll
    for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
      const bool on = k < A.nk;
      f0[k] = on ? A.cum_fwd[v0 * A.stride + k] : 0.0f;
      b0[k] = on ? A.cum_bwd[v0 * A.stride + k] : 0.0f;
      f1[k] = on ? A.cum_fwd[u1 * A.stride + k] : 0.0f;
      b1[k] = on ? A.cum_bwd[u1 * A.stride + k] : 0.0f;
    }
    term(pc0, w0, f0, b0);
    if (two) term(pc1, w1, f1, b1);
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
    double x = acc[k];
    for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    if (lane == 0) part[wave][k] = x;
  }
  __syncthreads();
  if ((int)threadIdx.x < A.nk) {
    double x = 0.0;
    for (int w = 0; w < kAdjBetaThreads / kWave; ++w) x += part[w][threadIdx.x];
    A.partial[(int64_t)blockIdx.x * GJ_MAX_NETS + A.cols[threadIdx.x]] += x;     // this (row, column) is this workgroup's alone
  }
}

__global__ void k_adjoint_beta_finish(int32_t n_cols, const double* partial, const float* scale, double* out) {
  const int c = threadIdx.x;
  if (c >= n_cols) return;
  double x = 0.0;
  for (int b = 0; b < GJ_ADJ_BETA_BLOCKS; ++b) x += partial[(int64_t)b * GJ_MAX_NETS + c];
  out[c] = x * (double)(*scale) * 2.302585092994046;      // ln(10)
}

// f2: per-step result reductions (reference grad_june/runner.py:167,198-224), one streaming pass
struct StatsArgs {
  int64_t n;
  const uint8_t* cls;
  const float* inf;
  const float* stage;
  int32_t n_bins;
  int32_t edges[GJ_MAX_AGE_BINS + 1];
  int32_t dead;
  int32_t vec4;     // all three arrays 16-byte (cls: 4-byte) aligned
  double* out;
};

__global__ __launch_bounds__(kThreads) void k_step_stats(const StatsArgs S) {
  constexpr int kOut = GJ_MAX_AGE_BINS + 2;
  __shared__ double part[kThreads / kWave][kOut];
  double acc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) acc[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto take = [&](float inf, float stage, int cls) {
    const int age = cls % 100;
    acc[0] += inf;
#pragma unroll
    for (int b = 0; b < GJ_MAX_AGE_BINS; ++b)
      if (b < S.n_bins && age > S.edges[b] && age < S.edges[b + 1]) acc[1 + b] += inf;
    if (stage == (float)S.dead) acc[GJ_MAX_AGE_BINS + 1] += 1.0;
  };
  int64_t first_scalar = 0;
  if (S.vec4) {   // 16-byte aligned arrays: four agents per lane and load (the scalar form is latency-bound)
    const int64_t n4 = S.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 f = reinterpret_cast<const float4*>(S.inf)[i];
      const float4 g = reinterpret_cast<const float4*>(S.stage)[i];
      const uint32_t c = reinterpret_cast<const uint32_t*>(S.cls)[i];
      take(f.x, g.x, (int)(c & 0xFF));
      take(f.y, g.y, (int)((c >> 8) & 0xFF));
      take(f.z, g.z, (int)((c >> 16) & 0xFF));
      take(f.w, g.w, (int)(c >> 24));
    }
    first_scalar = n4 << 2;
  }
  for (int64_t a = first_scalar + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < S.n; a += stride)
    take(S.inf[a], S.stage[a], (int)S.cls[a]);
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int k = 0; k < kOut; ++k) {
    double v = acc[k];
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) part[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
    for (int w = 0; w < kThreads / kWave; ++w) v += part[w][threadIdx.x];
    const int k = threadIdx.x;
    int dst = -1;
    if (k == 0) dst = 0;
    else if (k <= GJ_MAX_AGE_BINS) dst = (k - 1 < S.n_bins) ? k : -1;
    else dst = 1 + S.n_bins;
    if (dst >= 0 && v != 0.0) atomicAdd(&S.out[dst], v);
  }
}

// f1 + f2 in one pass (gj_symptoms_step_stats): the stage update of four agents per lane, written back only where a
// value changed (early in an epidemic almost nobody moves: the three arrays are then read, not rewritten), and the
// Runner's reductions taken from the registers that hold the updated stages.
__global__ __launch_bounds__(kThreads) void k_symptoms_stats(const SymptomsArgs S, const StatsArgs R) {
  constexpr int kOut = GJ_MAX_AGE_BINS + 2;
  __shared__ double part[kThreads / kWave][kOut];
  double acc[kOut];
#pragma unroll
  for (int k = 0; k < kOut; ++k) acc[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto take = [&](float inf, float stage, int cls) {
    const int age = cls % 100;
    acc[0] += inf;
#pragma unroll
    for (int b = 0; b < GJ_MAX_AGE_BINS; ++b)
      if (b < R.n_bins && age > R.edges[b] && age < R.edges[b + 1]) acc[1 + b] += inf;
    if (stage == (float)R.dead) acc[GJ_MAX_AGE_BINS + 1] += 1.0;
  };
  int64_t first_scalar = 0;
  if (R.vec4) {
    const int64_t n4 = S.n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      // all six loads issued before the first use
      const float4 nw = reinterpret_cast<const float4*>(S.new_inf)[i];
      float4 c = reinterpret_cast<const float4*>(S.cur)[i];
      float4 x = reinterpret_cast<const float4*>(S.nxt)[i];
      float4 t = reinterpret_cast<const float4*>(S.ttn)[i];
      const float4 f = reinterpret_cast<const float4*>(R.inf)[i];
      const uint32_t cl = reinterpret_cast<const uint32_t*>(S.cls)[i];
      bool ch = symptoms_agent(S, 4 * i, nw.x, (int)(cl & 0xFF), c.x, x.x, t.x);
      ch |= symptoms_agent(S, 4 * i + 1, nw.y, (int)((cl >> 8) & 0xFF), c.y, x.y, t.y);
      ch |= symptoms_agent(S, 4 * i + 2, nw.z, (int)((cl >> 16) & 0xFF), c.z, x.z, t.z);
      ch |= symptoms_agent(S, 4 * i + 3, nw.w, (int)(cl >> 24), c.w, x.w, t.w);
      if (ch) {
        reinterpret_cast<float4*>(S.cur)[i] = c;
        reinterpret_cast<float4*>(S.nxt)[i] = x;
        reinterpret_cast<float4*>(S.ttn)[i] = t;
      }
      take(f.x, c.x, (int)(cl & 0xFF));
      take(f.y, c.y, (int)((cl >> 8) & 0xFF));
      take(f.z, c.z, (int)((cl >> 16) & 0xFF));
      take(f.w, c.w, (int)(cl >> 24));
    }
    first_scalar = n4 << 2;
  }
  for (int64_t a = first_scalar + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < S.n; a += stride) {
    float cur = S.cur[a], nx = S.nxt[a], tt = S.ttn[a];
    const int cls = (int)S.cls[a];
    if (symptoms_agent(S, a, S.new_inf[a], cls, cur, nx, tt)) {
      S.cur[a] = cur;
      S.nxt[a] = nx;
      S.ttn[a] = tt;
    }
    take(R.inf[a], cur, cls);
  }
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
#pragma unroll
  for (int k = 0; k < kOut; ++k) {
    double v = acc[k];
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    if (lane == 0) part[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < kOut) {
    double v = 0.0;
    for (int w = 0; w < kThreads / kWave; ++w) v += part[w][threadIdx.x];
    const int k = threadIdx.x;
    int dst = -1;
    if (k == 0) dst = 0;
    else if (k <= GJ_MAX_AGE_BINS) dst = (k - 1 < R.n_bins) ? k : -1;
    else dst = 1 + R.n_bins;
    if (dst >= 0 && v != 0.0) atomicAdd(&R.out[dst], v);
  }
}

// a2 alone: q*transmission for a caller-supplied transmission vector
__global__ __launch_bounds__(kThreads) void k_quarantine_transmission(int64_t n, const float* __restrict__ stage,
                                                                      const float* __restrict__ trans,
                                                                      float* __restrict__ qtrans, float q_thr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) qtrans[i] = (stage[i] < q_thr ? 1.0f : 0.0f) * trans[i];
}

// halo pack / unpack
__global__ __launch_bounds__(kThreads) void k_pack(int64_t n, const int32_t* __restrict__ idx,
                                                   const float* __restrict__ src, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}
__global__ __launch_bounds__(kThreads) void k_unpack(int64_t n, const int32_t* __restrict__ idx,
                                                     const float* __restrict__ in, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = in[i];
}

// ------------------------------------------------------------------------------------------
// host side: argument checking and launch
// ------------------------------------------------------------------------------------------
static inline int launch_status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? GJ_OK : (int)e;
}

static int check_tiled(const gj_plan* plan);

static int check_plan(const gj_plan* plan) {
  if (!plan) return GJ_E_NULL;
  if (plan->n_agents < 0 || plan->n_ext_agents < plan->n_agents) return GJ_E_RANGE;
  if (plan->n_ext_agents > INT32_MAX) return GJ_E_RANGE;
  if (plan->n_sets < 0 || plan->n_sets > GJ_MAX_SETS) return GJ_E_RANGE;
  for (int s = 0; s < plan->n_sets; ++s) {
    const gj_edge_set& S = plan->sets[s];
    if (S.n_edges < 0 || S.n_edges > INT32_MAX || S.n_venues < 0 || S.n_venues > INT32_MAX) return GJ_E_RANGE;
    if (S.cum_stride < 1 || S.cum_stride > GJ_MAX_NETS_PER_SET) return GJ_E_PLAN;
    if (!S.v_pcontact && S.n_venues > 0) return GJ_E_NULL;
    if (S.n_venues > 0 && !S.cum) return GJ_E_NULL;
    if (plan->tiled) continue;
    if (!S.v_rowptr || !S.a_rowptr) return GJ_E_NULL;
    if (S.n_edges > 0 && (!S.v_agent || !S.a_venue)) return GJ_E_NULL;
  }
  if (plan->tiled) return check_tiled(plan);
  if (plan->n_blocks < 0 || (plan->n_blocks > 0 && !plan->blocks)) return GJ_E_PLAN;
  if (plan->n_long_rows < 0 || (plan->n_long_rows > 0 && (!plan->long_rows || !plan->partial))) return GJ_E_PLAN;
  return GJ_OK;
}

// group the step's networks by edge set, keeping order; networks of one set must be adjacent
struct Groups {
  int n;
  int set[GJ_MAX_SETS];
  int first[GJ_MAX_SETS];
  int nk[GJ_MAX_SETS];
};

static int group_networks(const gj_plan* plan, const gj_step_params* p, Groups* G) {
  if (!p) return GJ_E_NULL;
  if (p->transpose && !plan->tiled) return GJ_E_PLAN;   // the backward passes exist for the tiled layout
  if (p->n_nets < 0 || p->n_nets > GJ_MAX_NETS) return GJ_E_RANGE;
  if (p->day_type < 0 || p->day_type > 1) return GJ_E_RANGE;
  G->n = 0;
  bool seen[GJ_MAX_SETS] = {false};
  for (int i = 0; i < p->n_nets; ++i) {
    const gj_network& N = p->nets[i];
    if (N.set < 0 || N.set >= plan->n_sets) return GJ_E_RANGE;
    if (N.mask_kind < GJ_MASK_RAW || N.mask_kind > GJ_MASK_QL_AGE75) return GJ_E_RANGE;
    if (N.mask_kind >= GJ_MASK_QL) {
      if (N.table < 0 || N.table >= plan->n_tables || !plan->tables || !plan->agent_class) return GJ_E_PLAN;
    }
    if (G->n > 0 && G->set[G->n - 1] == N.set) {
      if (G->nk[G->n - 1] >= plan->sets[N.set].cum_stride) return GJ_E_PLAN;
      // transmissions of one set come from one source array: RAW and masked kinds cannot mix
      const int m0 = p->nets[G->first[G->n - 1]].mask_kind;
      if ((m0 == GJ_MASK_RAW) != (N.mask_kind == GJ_MASK_RAW)) return GJ_E_PLAN;
      G->nk[G->n - 1]++;
    } else {
      if (seen[N.set]) return GJ_E_PLAN;  // not adjacent
      seen[N.set] = true;
      G->set[G->n] = N.set;
      G->first[G->n] = i;
      G->nk[G->n] = 1;
      G->n++;
    }
  }
  return GJ_OK;
}

// full: the launch reads/writes is_infected and infection_time too (a1, a9)
static int check_state(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, bool full = true) {
  if (!st) return GJ_E_NULL;
  if (plan->n_agents == 0) return GJ_OK;
  if (!st->transmission || !st->susceptibility) return GJ_E_NULL;
  if (full && (!st->is_infected || !st->infection_time)) return GJ_E_NULL;
  if (p->has_quarantine && (!st->current_stage || !st->q_transmission)) return GJ_E_NULL;
  return GJ_OK;
}

static int do_transmission(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p,
                           hipStream_t stream) {
  if (!st->max_infectiousness || !st->shape || !st->rate || !st->shift) return GJ_E_NULL;
  const int64_t n = plan->n_agents;
  if (n == 0) return GJ_OK;
  const int64_t n4 = (n + 3) / 4;
  int64_t blocks = (n4 + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_transmission, dim3((unsigned)blocks), dim3(kThreads), 0, stream, n, st->max_infectiousness,
                     st->shape, st->rate, st->shift, st->infection_time, st->is_infected, st->current_stage,
                     st->transmission, st->q_transmission, p->now, p->has_quarantine, p->q_threshold, p->clock);
  return launch_status();
}

static int do_venue_reduce(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G,
                           hipStream_t stream) {
  if (plan->n_blocks == 0 || G.n == 0) return GJ_OK;
  P1Args A;
  for (int s = 0; s < GJ_MAX_SETS; ++s) {
    A.sets[s] = SetP1{};
    A.sets[s].nk = 0;
  }
  for (int g = 0; g < G.n; ++g) {
    const gj_edge_set& E = plan->sets[G.set[g]];
    SetP1& S = A.sets[G.set[g]];
    S.v_rowptr = E.v_rowptr;
    S.v_agent = E.v_agent;
    S.v_pc = E.v_pcontact;
    S.cum = E.cum;
    S.stride = E.cum_stride;
    S.nk = G.nk[g];
    S.raw = p->nets[G.first[g]].mask_kind == GJ_MASK_RAW;
    S.leisure = 0;
    for (int k = 0; k < G.nk[g]; ++k) {
      const gj_network& N = p->nets[G.first[g] + k];
      S.beta[k] = N.beta;
      S.table[k] = N.mask_kind >= GJ_MASK_QL ? N.table : -1;
      if (N.mask_kind >= GJ_MASK_QL) S.leisure = 1;
    }
    if (S.leisure) {
      // a set either is a leisure set (every network has a table) or is not
      for (int k = 0; k < G.nk[g]; ++k)
        if (S.table[k] < 0) return GJ_E_PLAN;
    }
  }
  A.blocks = plan->blocks;
  A.trans = st->transmission;
  A.qtrans = p->has_quarantine ? st->q_transmission : st->transmission;
  A.cls = plan->agent_class;
  A.tables = plan->tables;
  A.partial = plan->partial;
  A.day_type = p->day_type;
  hipLaunchKernelGGL(k_venue_reduce, dim3((unsigned)plan->n_blocks), dim3(kThreads), 0, stream, A);
  int rc = launch_status();
  if (rc != GJ_OK) return rc;
  if (plan->n_long_rows > 0) {
    CombineArgs C;
    for (int s = 0; s < GJ_MAX_SETS; ++s) {
      C.sets[s].cum = A.sets[s].cum;
      C.sets[s].stride = A.sets[s].stride;
      C.sets[s].nk = A.sets[s].nk;
    }
    C.rows = plan->long_rows;
    C.partial = plan->partial;
    C.n_rows = plan->n_long_rows;
    const int64_t threads = (int64_t)plan->n_long_rows * GJ_MAX_NETS_PER_SET;
    hipLaunchKernelGGL(k_combine_long, dim3((unsigned)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       stream, C);
    rc = launch_status();
  }
  return rc;
}

static int do_agent_gather(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G,
                           const gj_step_io* io, int sample, hipStream_t stream) {
  const int64_t n = plan->n_agents;
  if (n == 0) return GJ_OK;
  if (io && io->agent_sums) return GJ_E_PLAN;      // the CSR kernel multiplies by the susceptibility per term: tiled layout only
  P2Args P;
  P.n_groups = G.n;
  P.any_leisure = 0;
  for (int g = 0; g < G.n; ++g) {
    const gj_edge_set& E = plan->sets[G.set[g]];
    SetP2& S = P.groups[g];
    S.a_rowptr = E.a_rowptr;
    S.a_venue = E.a_venue;
    S.cum = E.cum;
    S.stride = E.cum_stride;
    S.nk = G.nk[g];
    for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
      S.mask[k] = 0;
      S.table[k] = 0;
    }
    for (int k = 0; k < G.nk[g]; ++k) {
      const gj_network& N = p->nets[G.first[g] + k];
      S.mask[k] = N.mask_kind;
      S.table[k] = N.mask_kind >= GJ_MASK_QL ? N.table : 0;
      if (N.mask_kind >= GJ_MASK_QL) P.any_leisure = 1;
    }
  }
  P.n_agents = n;
  P.cls = plan->agent_class;
  P.tables = plan->tables;
  P.stage = st->current_stage;
  P.susceptibility = st->susceptibility;
  P.is_infected = st->is_infected;
  P.infection_time = st->infection_time;
  P.not_infected_probs = io ? io->not_infected_probs : nullptr;
  P.new_infected = io ? io->new_infected : nullptr;
  P.trans_susc = io ? io->trans_susc : nullptr;
  P.exp_noise = io ? io->exp_noise : nullptr;
  P.now = p->now;
  P.dt = p->delta_time;
  P.q_thr = p->q_threshold;
  P.day_type = p->day_type;
  P.has_q = p->has_quarantine;
  P.sample = sample;
  P.seed = p->seed;
  P.step = p->step;
  P.agent_offset = p->agent_offset;
  P.clock = p->clock;
  const int64_t blocks = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(k_agent_gather, dim3((unsigned)blocks), dim3(kThreads), 0, stream, P);
  return launch_status();
}

// ------------------------------------------------------------------------------------------
// tiled layout: host side
// ------------------------------------------------------------------------------------------
static int check_tiled(const gj_plan* plan) {
  const gj_tiled* T = plan->tiled;
  if (T->n_slices < 1 || T->slice_agents < 64 || T->slice_agents % 64 || T->slice_agents > kMaxSliceAgents) return GJ_E_PLAN;
  // slices cover owned + halo agents; halo agents start on a slice boundary (phase D runs on the
  // slices of owned agents only)
  if ((int64_t)T->n_slices * T->slice_agents < plan->n_ext_agents) return GJ_E_PLAN;
  if (T->n_work < 0 || (T->n_work > 0 && !T->work)) return GJ_E_PLAN;
  for (int s = 0; s < plan->n_sets; ++s) {
    const gj_tiled_set& S = T->sets[s];
    if (S.n_blocks < 0) return GJ_E_PLAN;
    if (S.n_blocks == 0) continue;
    if (!S.blk_v0 || !S.blk_e0 || !S.tile_sptr || !S.tile_jpos || !S.chunk_ptr) return GJ_E_NULL;
    const bool run = S.run_pv_blk || S.run_pv_win || S.run_blk_r0 || S.run_win_lo || S.run_win_n;
    if (run) {   // all or none; one window of cum per slice must fit phase D's table region
      if (!S.run_pv_blk || !S.run_pv_win || !S.run_blk_r0 || !S.run_win_lo || !S.run_win_n) return GJ_E_NULL;
      if (S.run_max_window < 0 || S.run_max_window > 32768 || S.run_tiled_edges < 0 ||
          S.run_tiled_edges > plan->sets[s].n_edges || S.ell_k)
        return GJ_E_PLAN;
    }
    if (S.presum && (!S.ell_k || T->presum_wgs < 1 || T->presum_wgs > 4096)) return GJ_E_PLAN;
    const int64_t held = run ? (int64_t)S.run_tiled_edges : plan->sets[s].n_edges;
    if (held > 0 && (!S.e_lv || !S.a_la || !S.val || !S.chunk_desc)) return GJ_E_NULL;
    if (S.max_block_venues < 1 || S.max_block_venues > 65536) return GJ_E_PLAN;
    // phases A and D address val / a_la / chunk_desc with 32-bit byte offsets: < 2^30 edges per set (a larger set
    // is split over several edge sets of the same venue type)
    if (plan->sets[s].n_edges >= ((int64_t)1 << 30) - 8 * (int64_t)S.n_blocks) return GJ_E_RANGE;
    if (S.ell_k) {   // direct form of pass 2: 16-bit venue ids, one 2/4/8/16-byte row per agent
      if (S.ell_k != 2 && S.ell_k != 4 && S.ell_k != 8) return GJ_E_PLAN;
      if (!S.ell) return GJ_E_NULL;
      if (plan->sets[s].n_venues > 65534) return GJ_E_PLAN;
    }
  }
  return GJ_OK;
}

// edges the tiled arrays of a set hold (a run form keeps one edge per owned agent out of them)
static inline int64_t tiled_edges(const gj_plan* plan, int s) {
  const gj_tiled_set& S = plan->tiled->sets[s];
  return S.run_pv_blk ? (int64_t)S.run_tiled_edges : plan->sets[s].n_edges;
}

// pass 1 of this set runs in the direct form (k_tile_presum + k_presum_reduce) instead of phases A + B
static inline bool uses_presum(const gj_plan* plan, int s) {
  const gj_tiled* T = plan->tiled;
  return T->presum_wgs > 0 && T->sets[s].presum != nullptr && T->sets[s].ell_k != 0 && T->sets[s].n_blocks > 0 &&
         plan->sets[s].n_edges > 0;
}

template <typename K>
static int allow_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return GJ_E_PLAN;
  if (bytes > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return GJ_OK;
}

static void fill_set_a(const gj_plan* plan, const gj_step_params* p, const Groups& G, TSetA* sets) {
  const gj_tiled* T = plan->tiled;
  for (int s = 0; s < GJ_MAX_SETS; ++s) sets[s] = TSetA{};
  for (int g = 0; g < G.n; ++g) {
    const int s = G.set[g];
    const gj_tiled_set& S = T->sets[s];
    sets[s].a_la = S.a_la;
    sets[s].tile_sptr = S.tile_sptr;
    sets[s].tile_jpos = S.tile_jpos;
    sets[s].chunk_ptr = S.chunk_ptr;
    sets[s].chunk_desc = reinterpret_cast<const int4*>(S.chunk_desc);
    sets[s].val = S.val;
    sets[s].J = S.n_blocks;
    sets[s].active = (S.n_blocks > 0 && tiled_edges(plan, s) > 0) ? G.nk[g] : 0;
    sets[s].presum = uses_presum(plan, s);
    sets[s].raw = p->nets[G.first[g]].mask_kind == GJ_MASK_RAW;
    sets[s].wide = S.desc_wide;          // 0 / 1: descriptor format, 2: explicit slots
    sets[s].direct = S.ell_k != 0;
    sets[s].multi_slots = S.multi_slots;
  }
}

// LDS of phases A and D: one slice (fp32 values in A, 64-bit fixed-point sums in D)
static size_t slice_lds(const gj_tiled* T, size_t elem) { return (size_t)T->slice_agents * elem; }
// phase D: 64-bit sums + two flag bits per agent (saturated up / down, gj_tiled.h fx_flag)
static size_t agents_lds(const gj_tiled* T) { return (size_t)T->slice_agents * sizeof(fx_t) + 2 * ((size_t)T->slice_agents / 8); }

// pass 1 of the sets in its direct form: LDS tables of fixed-point sums per workgroup (k_tile_presum)
static int presum_fill(const gj_plan* plan, const gj_step_params* p, const Groups& G, int g, TPSet* X) {
  const gj_tiled* T = plan->tiled;
  const int s = G.set[g];
  const gj_tiled_set& S = T->sets[s];
  const gj_edge_set& E = plan->sets[s];
  const int64_t owned_slices = (plan->n_agents + T->slice_agents - 1) / T->slice_agents;
  X->ell = S.ell;
  X->plane_stride = owned_slices * (int64_t)T->slice_agents * 2;
  X->partial = reinterpret_cast<fx_t*>(S.presum);
  X->planes = S.ell_k / 2;
  X->V = (int32_t)E.n_venues;
  X->stride = E.cum_stride;
  X->nk = G.nk[g];
  X->raw = p->nets[G.first[g]].mask_kind == GJ_MASK_RAW;
  X->leisure = 0;
  X->_pad = 0;
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) X->table[k] = X->age75[k] = 0;
  for (int k = 0; k < G.nk[g]; ++k) {
    const gj_network& N = p->nets[G.first[g] + k];
    X->table[k] = N.mask_kind >= GJ_MASK_QL ? N.table : -1;
    X->age75[k] = N.mask_kind == GJ_MASK_QL_AGE75;
    if (N.mask_kind >= GJ_MASK_QL) X->leisure = 1;
  }
  if (X->leisure) {
    if (!plan->agent_class || (uintptr_t)plan->agent_class % 4 != 0) return GJ_E_PLAN;
    for (int k = 0; k < G.nk[g]; ++k)
      if (X->table[k] < 0) return GJ_E_PLAN;
  } else if (G.nk[g] != 1) {
    return GJ_E_PLAN;
  }
  return GJ_OK;
}

static int tiled_presum(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G,
                        hipStream_t stream) {
  const gj_tiled* T = plan->tiled;
  TilePArgs P;
  P.n_sets = 0;
  for (int g = 0; g < G.n; ++g) {
    if (!uses_presum(plan, G.set[g])) continue;
    if (P.n_sets == GJ_MAX_PRESUM) return GJ_E_PLAN;
    const int rc = presum_fill(plan, p, G, g, &P.sets[P.n_sets++]);
    if (rc) return rc;
  }
  if (P.n_sets == 0 || plan->n_agents == 0) return GJ_OK;
  // the LDS table of a set: (venues of a group x networks + 64 scratch) 8-byte sums, leisure weights, one flag bit per sum
  const int64_t budget = 160 * 1024;
  size_t lds = 0;
  for (int t = 0; t < P.n_sets; ++t) {
    TPSet& X = P.sets[t];
    const int64_t fixed = 64 * 8 + (X.leisure ? (int64_t)X.nk * 200 * 4 : 0) + 64;
    int64_t per_venue = (int64_t)X.nk * 8;                  // + its flag bits: 1/8 byte per sum, counted below
    int64_t gv = (budget - fixed) * 8 / (per_venue * 8 + X.nk);
    if (T->direct_table_floats > 0 && gv > T->direct_table_floats / X.nk) gv = T->direct_table_floats / X.nk;   // tests: several groups
    if (gv > X.V) gv = X.V;
    if (gv < 1) gv = 1;
    X.group_venues = (int32_t)gv;
    const size_t need = (size_t)(gv * X.nk + 64) * 8 + (X.leisure ? (size_t)X.nk * 200 * 4 : 0) + ((size_t)(gv * X.nk + 64 + 31) / 32) * 4;
    if (need > lds) lds = need;
  }
  int64_t apw = (plan->n_agents + T->presum_wgs - 1) / T->presum_wgs;
  apw = (apw + 63) / 64 * 64;
  P.agents_per_wg = (int32_t)apw;
  P.n_agents = plan->n_agents;
  P.trans = st->transmission;
  P.qtrans = p->has_quarantine ? st->q_transmission : st->transmission;
  if ((uintptr_t)P.trans % 16 != 0 || (uintptr_t)P.qtrans % 16 != 0) return GJ_E_PLAN;
  P.cls = plan->agent_class;
  P.tables = plan->tables;
  P.day_type = p->day_type;
  P.transpose = p->transpose;
  int rc = allow_lds(k_tile_presum, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tile_presum, dim3((unsigned)T->presum_wgs), dim3(kTileThreads), lds, stream, P);
  return launch_status();
}

static int tiled_presum_reduce(const gj_plan* plan, const gj_step_params* p, const Groups& G, hipStream_t stream) {
  const gj_tiled* T = plan->tiled;
  PReduceArgs R;
  R.n_sets = 0;
  R.n_wgs = T->presum_wgs;
  R._pad = 0;
  int64_t total = 0;
  for (int g = 0; g < G.n; ++g) {
    const int s = G.set[g];
    if (!uses_presum(plan, s)) continue;
    if (R.n_sets == GJ_MAX_PRESUM) return GJ_E_PLAN;
    PReduceSet& X = R.sets[R.n_sets++];
    const gj_edge_set& E = plan->sets[s];
    X.partial = reinterpret_cast<const fx_t*>(T->sets[s].presum);
    X.v_pc = E.v_pcontact;
    X.cum = E.cum;
    X.V = (int32_t)E.n_venues;
    X.stride = E.cum_stride;
    X.nk = G.nk[g];
    X.first = (int32_t)total;
    for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) X.beta[k] = k < G.nk[g] ? p->nets[G.first[g] + k].beta : 0.0f;
    total += (int64_t)X.V * X.nk;
  }
  if (R.n_sets == 0 || total == 0) return GJ_OK;
  if (total > INT32_MAX) return GJ_E_RANGE;
  R.total = (int32_t)total;
  hipLaunchKernelGGL(k_presum_reduce, dim3((unsigned)((total + kWave - 1) / kWave)), dim3(kThreads), 0, stream, R);
  return launch_status();
}

static int tiled_scatter(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G,
                         hipStream_t stream) {
  const gj_tiled* T = plan->tiled;
  if (plan->n_ext_agents == 0 || G.n == 0) return GJ_OK;
  TileAArgs A;
  fill_set_a(plan, p, G, A.sets);
  A.n_sets = plan->n_sets;
  A.slice_agents = T->slice_agents;
  A.n_agents = plan->n_ext_agents;     // phase A also scatters the halo agents' values
  A.trans = st->transmission;
  A.qtrans = p->has_quarantine ? st->q_transmission : st->transmission;
  const size_t lds = slice_lds(T, sizeof(float));
  int rc = allow_lds(k_tile_scatter, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)T->n_slices), dim3(kTileThreads), lds, stream, A);
  rc = launch_status();
  if (rc) return rc;
  return tiled_presum(plan, st, p, G, stream);
}

// Bit pattern of the largest |term| a venue sum of this set takes: 2^14 (the window of the 2^-36 fixed point) while the
// largest venue has at most 4 096 edges, else 2^26 / next_pow2(edges) - terms x edges then stays below 2^62 fixed-point
// units and no sum can wrap.  A term beyond it saturates its venue (gj_tiled.h).  0 edges = not stated: 2^14.
static uint32_t venue_term_limit(int32_t max_venue_edges) {
  int e = 14;
  int64_t p2 = 4096;
  while (p2 < (int64_t)max_venue_edges && e > -20) {
    p2 <<= 1;
    --e;
  }
  const float lim = ldexpf(1.0f, e);
  uint32_t bits;
  memcpy(&bits, &lim, sizeof(bits));
  return bits;
}

static int tiled_venues(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G, int mode,
                        hipStream_t stream) {
  const gj_tiled* T = plan->tiled;
  if (G.n == 0) return GJ_OK;
  if (T->n_work == 0) return mode == 2 ? GJ_OK : tiled_presum_reduce(plan, p, G, stream);
  TileBArgs B;
  for (int s = 0; s < GJ_MAX_SETS; ++s) B.sets[s] = TSetB{};
  size_t lds = 16;
  for (int g = 0; g < G.n; ++g) {
    const int s = G.set[g];
    const gj_tiled_set& S = T->sets[s];
    const gj_edge_set& E = plan->sets[s];
    TSetB& X = B.sets[s];
    X.blk_v0 = S.blk_v0;
    X.blk_e0 = S.blk_e0;
    X.e_lv = S.e_lv;
    X.e_cls = S.e_cls;
    X.val = S.val;
    X.v_pc = E.v_pcontact;
    X.cum = E.cum;
    X.stride = E.cum_stride;
    X.nk = S.n_blocks > 0 ? G.nk[g] : 0;
    if (uses_presum(plan, s)) X.nk = 0;          // pass 1 of the set is k_tile_presum's, pass 2 phase D's: nothing here
    X.direct = S.ell_k != 0;
    X.leisure = 0;
    for (int k = 0; k < G.nk[g]; ++k) {
      const gj_network& N = p->nets[G.first[g] + k];
      X.beta[k] = N.beta;
      X.table[k] = N.mask_kind >= GJ_MASK_QL ? N.table : -1;
      X.age75[k] = N.mask_kind == GJ_MASK_QL_AGE75;
      if (N.mask_kind >= GJ_MASK_QL) X.leisure = 1;
    }
    if (X.leisure) {
      if (!S.e_cls && E.n_edges > 0) return GJ_E_PLAN;
      for (int k = 0; k < G.nk[g]; ++k)
        if (X.table[k] < 0) return GJ_E_PLAN;
    } else if (G.nk[g] != 1) {
      return GJ_E_PLAN;   // several networks on one set need per-network tables
    }
    X.pv_blk = nullptr;
    X.blk_r0 = nullptr;
    X.x = nullptr;
    X.n_x = 0;
    if (S.run_pv_blk && mode != 2) {     // run form: phase B reads the primary edges' values from the per-agent array
      if (X.leisure) return GJ_E_PLAN;
      const bool raw = p->nets[G.first[g]].mask_kind == GJ_MASK_RAW;
      X.x = (p->has_quarantine && !raw) ? st->q_transmission : st->transmission;
      if (!X.x) return GJ_E_NULL;
      if ((uintptr_t)X.x % 16 != 0 || (uintptr_t)S.run_pv_blk % 16 != 0) return GJ_E_PLAN;
      X.pv_blk = S.run_pv_blk;
      X.blk_r0 = S.run_blk_r0;
      X.n_x = plan->n_ext_agents;
    }
    const size_t n_sums = (size_t)X.nk * S.max_block_venues + 64;               // + a scratch sum per lane of a wave
    const size_t need = n_sums * sizeof(fx_t) + (X.leisure ? 2 * 200 * (size_t)X.nk : 0) * sizeof(float) +
                        2 * ((n_sums + 31) / 32 * 4);                                 // two flag bits per sum
    X.term_limit = venue_term_limit(S.max_venue_edges);
    if (X.nk && need > lds) lds = need;
  }
  B.work = T->work;
  B.tables = plan->tables;
  B.day_type = p->day_type;
  B.mode = mode;
  B.transpose = p->transpose;
  int rc = allow_lds(k_tile_venues, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tile_venues, dim3((unsigned)T->n_work), dim3(kTileThreads), lds, stream, B);
  rc = launch_status();
  if (rc || mode == 2) return rc;
  return tiled_presum_reduce(plan, p, G, stream);
}

static int tiled_agents(const gj_plan* plan, const gj_agent_state* st, const gj_step_params* p, const Groups& G,
                        const gj_step_io* io, int sample, hipStream_t stream) {
  const gj_tiled* T = plan->tiled;
  if (plan->n_agents == 0) return GJ_OK;
  TileDArgs D;
  fill_set_a(plan, p, G, D.sets);
  D.n_sets = plan->n_sets;
  D.slice_agents = T->slice_agents;
  D.n_agents = plan->n_agents;
  D.stage = st->current_stage;
  D.susceptibility = st->susceptibility;
  D.is_infected = st->is_infected;
  D.infection_time = st->infection_time;
  D.not_infected_probs = io ? io->not_infected_probs : nullptr;
  D.new_infected = io ? io->new_infected : nullptr;
  D.trans_susc = io ? io->trans_susc : nullptr;
  D.agent_sums = io ? io->agent_sums : nullptr;
  D.exp_noise = io ? io->exp_noise : nullptr;
  D.now = p->now;
  D.dt = p->delta_time;
  D.q_thr = p->q_threshold;
  D.has_q = p->has_quarantine;
  D.sample = sample;
  D.seed = p->seed;
  D.step = p->step;
  D.agent_offset = p->agent_offset;
  D.clock = p->clock;
  D.acc_scratch = T->agent_scratch;
  size_t lds = agents_lds(T);
  // direct form of pass 2: the sets whose cum is read from an LDS table behind the slice's (compacted) sums
  D.n_direct = 0;
  D.day_type = p->day_type;
  D.transpose = p->transpose;
  D.cls = plan->agent_class;
  D.tables = plan->tables;
  const int64_t owned_slices = (plan->n_agents + T->slice_agents - 1) / T->slice_agents;
  for (int g = 0; g < G.n; ++g) {
    const int s = G.set[g];
    const gj_tiled_set& S = T->sets[s];
    const gj_edge_set& E = plan->sets[s];
    const bool run = S.run_pv_win != nullptr;
    if ((!S.ell_k && !run) || S.n_blocks == 0 || E.n_edges == 0) continue;
    if (D.n_direct == GJ_MAX_DIRECT) return GJ_E_PLAN;
    TDirect& X = D.direct[D.n_direct++];
    X.ell = run ? S.run_pv_win : S.ell;
    X.cum = E.cum;
    X.K = 2;
    X.planes = run ? 1 : S.ell_k / 2;
    X.plane_stride = owned_slices * (int64_t)T->slice_agents * 2;
    X.V = run ? S.run_max_window : (int32_t)E.n_venues;      // (run form: the table is one slice's window of cum)
    X.win_lo = run ? S.run_win_lo : nullptr;
    X.win_n = run ? S.run_win_n : nullptr;
    if (run && (uintptr_t)S.run_pv_win % 8 != 0) return GJ_E_PLAN;
    X.stride = E.cum_stride;
    X.nk = G.nk[g];
    X.raw = p->nets[G.first[g]].mask_kind == GJ_MASK_RAW;
    X.leisure = 0;
    for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) X.table[k] = X.age75[k] = 0;
    for (int k = 0; k < G.nk[g]; ++k) {
      const gj_network& N = p->nets[G.first[g] + k];
      X.table[k] = N.mask_kind >= GJ_MASK_QL ? N.table : -1;
      X.age75[k] = N.mask_kind == GJ_MASK_QL_AGE75;
      if (N.mask_kind >= GJ_MASK_QL) X.leisure = 1;
    }
    if (X.leisure) {
      // the direct form reads four agents' classes as one dword (see gj_plan.agent_class)
      if (!plan->agent_class || (uintptr_t)plan->agent_class % 4 != 0) return GJ_E_PLAN;
      for (int k = 0; k < G.nk[g]; ++k)
        if (X.table[k] < 0) return GJ_E_PLAN;
    } else if (G.nk[g] != 1) {
      return GJ_E_PLAN;
    }
  }
  D.table_floats = D.table1_floats = 0;
  D._pad3 = 0;
  if (D.n_direct) {
    // LDS of the direct form (the slice's sums are in registers by then): two class-weight buffers, then two table
    // regions, each followed by 64 floats of slack for the last DMA piece.  Region 0 takes the largest table (in
    // groups of venues if it is larger than everything), region 1 what is left: a set whose whole table fits there is
    // staged while the previous set - in region 0 - is still being read.
    const int64_t budget = 160 * 1024 / 4 - 2 * kClassWeightFloats - 2 * kWave;
    int64_t largest = 0;
    for (int t = 0; t < D.n_direct; ++t) {
      const int64_t sz = (int64_t)D.direct[t].V * D.direct[t].stride;
      largest = largest > sz ? largest : sz;
    }
    int64_t cap0 = largest < budget ? largest : budget;
    if (T->direct_table_floats > 0 && T->direct_table_floats < cap0) cap0 = T->direct_table_floats;
    if (cap0 < GJ_MAX_NETS_PER_SET) cap0 = GJ_MAX_NETS_PER_SET;
    int64_t cap1 = budget - cap0;
    if (T->direct_table_floats > 0 && T->direct_table_floats < cap1) cap1 = T->direct_table_floats;
    int prev_region = -1;
    for (int t = 0; t < D.n_direct; ++t) {
      TDirect& X = D.direct[t];
      const int64_t sz = (int64_t)X.V * X.stride;
      X.region = (prev_region == 0 && sz <= cap1) ? 1 : 0;
      const int64_t cap = X.region ? cap1 : cap0;
      X.group_venues = sz <= cap ? X.V : (int32_t)(cap / X.stride);
      if (X.group_venues < 1 || (X.win_lo && sz > cap)) return GJ_E_PLAN;   // a window is staged whole
      X._pad = 0;
      prev_region = X.region;
    }
    D.table_floats = (int32_t)cap0;
    D.table1_floats = (int32_t)cap1;
    const size_t direct_lds = 4 * ((size_t)2 * kClassWeightFloats + (size_t)cap0 + kWave + (size_t)cap1 + kWave);
    if (direct_lds > lds) lds = direct_lds;
  }
  {
    uintptr_t bits = (uintptr_t)D.susceptibility | (uintptr_t)D.not_infected_probs | (uintptr_t)D.new_infected |
                     (uintptr_t)D.trans_susc | (uintptr_t)D.acc_scratch | (uintptr_t)D.agent_sums;
    D.io_vec4 = (bits % 16 == 0) ? 1 : 0;
  }
  int rc = allow_lds(k_tile_agents, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(k_tile_agents, dim3((unsigned)owned_slices), dim3(kTileThreads), lds, stream, D);
  rc = launch_status();
  if (rc || !D.acc_scratch) return rc;
  hipLaunchKernelGGL(k_tile_epilogue, dim3((unsigned)((plan->n_agents + 255) / 256)), dim3(256), 0, stream, D);
  return launch_status();
}

}  // namespace gj


// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

#ifdef GJ_DIAG_STAMPS
// diagnostics build only: the venue launch's per-workgroup timestamps, [3 * 8192] uint64 (start, end, XCD) to host memory
int gj_diag_venue_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gj::gj_diag_venue), sizeof(unsigned long long) * 3 * gj::kDiagVenueSlots);
}
#endif

int gj_version(void) { return GJ_ABI_VERSION; }

const char* gj_error_string(int code) {
  switch (code) {
    case GJ_OK: return "ok";
    case GJ_E_NULL: return "required pointer is NULL";
    case GJ_E_RANGE: return "count or index out of range";
    case GJ_E_PLAN: return "inconsistent plan or network list";
    case GJ_E_NODEVICE: return "no HIP device, or not a gfx950 (MI355X) device";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}

int gj_check_device(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return GJ_E_NODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GJ_E_NODEVICE;
  // the code object holds gfx950 (CDNA4, MI355X) kernels only, sized for its 160 KiB of LDS per CU
  const char* want = "gfx950";
  for (int i = 0; want[i]; ++i)
    if (prop.gcnArchName[i] != want[i]) return GJ_E_NODEVICE;
  return GJ_OK;
}

int gj_transmission_update(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
                           void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  if (!params) return GJ_E_NULL;
  rc = gj::check_state(plan, state, params);
  if (rc) return rc;
  return gj::do_transmission(plan, state, params, (hipStream_t)stream);
}

int gj_quarantine_transmission(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
                               void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  if (!params) return GJ_E_NULL;
  rc = gj::check_state(plan, state, params, false);
  if (rc) return rc;
  if (!params->has_quarantine || plan->n_agents == 0) return GJ_OK;
  const int64_t n = plan->n_agents;
  hipLaunchKernelGGL(gj::k_quarantine_transmission, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)),
                     dim3(gj::kThreads), 0, (hipStream_t)stream, n, state->current_stage, state->transmission,
                     state->q_transmission, params->q_threshold);
  return gj::launch_status();
}

int gj_venue_reduce(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params, void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  gj::Groups G;
  rc = gj::group_networks(plan, params, &G);
  if (rc) return rc;
  rc = gj::check_state(plan, state, params, false);
  if (rc) return rc;
  if (plan->tiled) {
    rc = gj::tiled_scatter(plan, state, params, G, (hipStream_t)stream);
    if (rc) return rc;
    return gj::tiled_venues(plan, state, params, G, 1, (hipStream_t)stream);
  }
  return gj::do_venue_reduce(plan, state, params, G, (hipStream_t)stream);
}

int gj_agent_gather(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
                    const gj_step_io* io, int sample, void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  gj::Groups G;
  rc = gj::group_networks(plan, params, &G);
  if (rc) return rc;
  rc = gj::check_state(plan, state, params, sample != 0);
  if (rc) return rc;
  if (plan->tiled) {
    rc = gj::tiled_venues(plan, state, params, G, 2, (hipStream_t)stream);
    if (rc) return rc;
    return gj::tiled_agents(plan, state, params, G, io, sample, (hipStream_t)stream);
  }
  return gj::do_agent_gather(plan, state, params, G, io, sample, (hipStream_t)stream);
}

int gj_sample_infect(int64_t n_agents, const float* not_infected_probs, const float* exp_noise, uint64_t seed,
                     uint64_t step, int64_t agent_offset, float now, float* new_infected, float* susceptibility,
                     float* is_infected, float* infection_time, void* stream) {
  if (n_agents < 0) return GJ_E_RANGE;
  if (n_agents == 0) return GJ_OK;
  if (!not_infected_probs) return GJ_E_NULL;
  const int n_state = (susceptibility != nullptr) + (is_infected != nullptr) + (infection_time != nullptr);
  if (n_state != 0 && n_state != 3) return GJ_E_NULL;   // all three or none (sample only)
  if (n_state == 0 && !new_infected) return GJ_E_NULL;
  const int64_t blocks = (n_agents + gj::kThreads - 1) / gj::kThreads;
  hipLaunchKernelGGL(gj::k_sample_infect, dim3((unsigned)blocks), dim3(gj::kThreads), 0, (hipStream_t)stream, n_agents,
                     not_infected_probs, exp_noise, seed, step, agent_offset, now, new_infected, susceptibility,
                     is_infected, infection_time);
  return gj::launch_status();
}

int gj_adjoint_sample(int64_t n, const float* susceptibility0, const float* infection_time0, const float* acc,
                      const float* exp_noise, uint64_t seed, uint64_t step, int64_t agent_offset, float now,
                      float delta_time, const float* g_susc, const float* g_inf, const float* g_time,
                      const float* g_new, float* x_out, float* grad_susc_out, float* grad_time_out, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!susceptibility0 || !infection_time0 || !acc || !x_out || !grad_susc_out || !grad_time_out) return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_adjoint_sample, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)), dim3(gj::kThreads),
                     0, (hipStream_t)stream, n, susceptibility0, infection_time0, acc, exp_noise, seed, step,
                     agent_offset, now, delta_time, g_susc, g_inf, g_time, g_new, x_out, grad_susc_out, grad_time_out);
  return gj::launch_status();
}

int gj_adjoint_transmission(int64_t n, const gj_agent_state* st, float now, const float* trans_bar,
                            const float* g_inf, float* grad_inf_out, float* grad_time_inout, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!st || !trans_bar || !grad_inf_out || !grad_time_inout) return GJ_E_NULL;
  if (!st->max_infectiousness || !st->shape || !st->rate || !st->shift || !st->infection_time || !st->is_infected)
    return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_adjoint_transmission, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)),
                     dim3(gj::kThreads), 0, (hipStream_t)stream, n, st->max_infectiousness, st->shape, st->rate,
                     st->shift, st->infection_time, st->is_infected, now, trans_bar, g_inf, grad_inf_out,
                     grad_time_inout);
  return gj::launch_status();
}

int gj_adjoint_beta_partial(int64_t n_venues, int32_t stride, int32_t nk, const float* cum_fwd, const float* cum_bwd,
                            const float* v_pcontact, const double* weights, const float* beta, const int32_t* cols,
                            double* partial, void* stream) {
  if (n_venues < 0 || stride < 1 || stride > GJ_MAX_NETS_PER_SET || nk < 0 || nk > stride) return GJ_E_RANGE;
  if (!partial || (nk > 0 && (!beta || !cols))) return GJ_E_NULL;
  if (n_venues == 0 || nk == 0) return GJ_OK;
  if (!cum_fwd || !cum_bwd || !v_pcontact) return GJ_E_NULL;
  gj::AdjBetaArgs A;
  A.n_venues = n_venues;
  A.stride = stride;
  A.nk = nk;
  A.cum_fwd = cum_fwd;
  A.cum_bwd = cum_bwd;
  A.v_pc = v_pcontact;
  A.weights = weights;
  for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {
    A.beta[k] = k < nk ? beta[k] : 0.0f;
    A.cols[k] = k < nk ? cols[k] : 0;
    if (k < nk && (cols[k] < 0 || cols[k] >= GJ_MAX_NETS)) return GJ_E_RANGE;
  }
  A.partial = partial;
  hipLaunchKernelGGL(gj::k_adjoint_beta_partial, dim3(GJ_ADJ_BETA_BLOCKS), dim3(gj::kAdjBetaThreads), 0, (hipStream_t)stream, A);
  return gj::launch_status();
}

int gj_adjoint_beta_finish(int32_t n_cols, const double* partial, const float* scale, double* out, void* stream) {
  if (n_cols < 0 || n_cols > GJ_MAX_NETS) return GJ_E_RANGE;
  if (n_cols == 0) return GJ_OK;
  if (!partial || !scale || !out) return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_adjoint_beta_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, n_cols, partial, scale, out);
  return gj::launch_status();
}

int gj_symptoms_update(int64_t n, const uint8_t* agent_class, const float* new_infected, float* current_stage,
                       float* next_stage, float* time_to_next_stage, const gj_symptoms_params* params,
                       const float* progresses, const float* dwell, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!agent_class || !new_infected || !current_stage || !next_stage || !time_to_next_stage || !params)
    return GJ_E_NULL;
  if (params->n_stages < 3 || params->n_stages > GJ_MAX_STAGES) return GJ_E_RANGE;
  if ((progresses == nullptr) != (dwell == nullptr)) return GJ_E_NULL;   // inject both or neither
  if (!progresses && !params->progress) return GJ_E_NULL;
  gj::SymptomsArgs S;
  S.P = *params;
  S.n = n;
  S.cls = agent_class;
  S.new_inf = new_infected;
  S.cur = current_stage;
  S.nxt = next_stage;
  S.ttn = time_to_next_stage;
  S.progresses = progresses;
  S.dwell = dwell;
  hipLaunchKernelGGL(gj::k_symptoms, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)), dim3(gj::kThreads), 0,
                     (hipStream_t)stream, S);
  return gj::launch_status();
}

int gj_adjoint_symptoms(int64_t n, const uint8_t* agent_class, const float* new_infected,
                        const float* current_stage0, const float* next_stage0, const float* time_to_next_stage0,
                        const gj_symptoms_params* params, const float* progresses, const float* dwell,
                        const float* g_current, const float* g_next, const float* g_time, float* g_current_in,
                        float* g_next_in, float* g_time_in, float* g_new_infected, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!agent_class || !new_infected || !current_stage0 || !next_stage0 || !time_to_next_stage0 || !params)
    return GJ_E_NULL;
  if (!g_current_in || !g_next_in || !g_new_infected) return GJ_E_NULL;
  if (params->n_stages < 3 || params->n_stages > GJ_MAX_STAGES) return GJ_E_RANGE;
  if ((progresses == nullptr) != (dwell == nullptr)) return GJ_E_NULL;   // inject both or neither
  if (!progresses && !params->progress) return GJ_E_NULL;
  gj::SymptomsAdjointArgs S;
  S.P = *params;
  S.n = n;
  S.cls = agent_class;
  S.new_inf = new_infected;
  S.cur0 = current_stage0;
  S.nxt0 = next_stage0;
  S.ttn0 = time_to_next_stage0;
  S.progresses = progresses;
  S.dwell = dwell;
  S.g_cur = g_current;
  S.g_nxt = g_next;
  S.g_ttn = g_time;
  S.g_cur_in = g_current_in;
  S.g_nxt_in = g_next_in;
  S.g_ttn_in = g_time_in;
  S.g_new = g_new_infected;
  hipLaunchKernelGGL(gj::k_adjoint_symptoms, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)),
                     dim3(gj::kThreads), 0, (hipStream_t)stream, S);
  return gj::launch_status();
}

int gj_step_stats(int64_t n, const uint8_t* agent_class, const float* is_infected, const float* current_stage,
                  int32_t n_bins, const int32_t* bin_edges, int32_t dead_stage, double* out, void* stream) {
  if (n < 0 || n_bins < 0 || n_bins > GJ_MAX_AGE_BINS) return GJ_E_RANGE;
  if (!out || (n_bins > 0 && !bin_edges)) return GJ_E_NULL;
  if (n == 0) return GJ_OK;
  if (!agent_class || !is_infected || !current_stage) return GJ_E_NULL;
  gj::StatsArgs S;
  S.n = n;
  S.cls = agent_class;
  S.inf = is_infected;
  S.stage = current_stage;
  S.n_bins = n_bins;
  for (int b = 0; b <= GJ_MAX_AGE_BINS; ++b) S.edges[b] = (b <= n_bins) ? bin_edges[b] : 0;
  S.dead = dead_stage;
  S.out = out;
  S.vec4 = (((uintptr_t)is_infected | (uintptr_t)current_stage) % 16 == 0 && (uintptr_t)agent_class % 4 == 0) ? 1 : 0;
  int64_t blocks = ((S.vec4 ? (n >> 2) + 3 : n) + gj::kThreads - 1) / gj::kThreads;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gj::k_step_stats, dim3((unsigned)blocks), dim3(gj::kThreads), 0, (hipStream_t)stream, S);
  return gj::launch_status();
}

int gj_symptoms_step_stats(int64_t n, const uint8_t* agent_class, const float* new_infected, float* current_stage,
                           float* next_stage, float* time_to_next_stage, const gj_symptoms_params* params,
                           const float* progresses, const float* dwell, const float* is_infected, int32_t n_bins,
                           const int32_t* bin_edges, int32_t dead_stage, double* out, void* stream) {
  if (n < 0 || n_bins < 0 || n_bins > GJ_MAX_AGE_BINS) return GJ_E_RANGE;
  if (!out || (n_bins > 0 && !bin_edges)) return GJ_E_NULL;
  if (n == 0) return GJ_OK;
  if (!agent_class || !new_infected || !current_stage || !next_stage || !time_to_next_stage || !params || !is_infected)
    return GJ_E_NULL;
  if (params->n_stages < 3 || params->n_stages > GJ_MAX_STAGES) return GJ_E_RANGE;
  if ((progresses == nullptr) != (dwell == nullptr)) return GJ_E_NULL;   // inject both or neither
  if (!progresses && !params->progress) return GJ_E_NULL;
  gj::SymptomsArgs S;
  S.P = *params;
  S.n = n;
  S.cls = agent_class;
  S.new_inf = new_infected;
  S.cur = current_stage;
  S.nxt = next_stage;
  S.ttn = time_to_next_stage;
  S.progresses = progresses;
  S.dwell = dwell;
  gj::StatsArgs R;
  R.n = n;
  R.cls = agent_class;
  R.inf = is_infected;
  R.stage = current_stage;
  R.n_bins = n_bins;
  for (int b = 0; b <= GJ_MAX_AGE_BINS; ++b) R.edges[b] = (b <= n_bins) ? bin_edges[b] : 0;
  R.dead = dead_stage;
  R.out = out;
  const uintptr_t bits = (uintptr_t)new_infected | (uintptr_t)current_stage | (uintptr_t)next_stage |
                         (uintptr_t)time_to_next_stage | (uintptr_t)is_infected;
  R.vec4 = (bits % 16 == 0 && (uintptr_t)agent_class % 4 == 0) ? 1 : 0;
  int64_t blocks = ((R.vec4 ? (n >> 2) + 3 : n) + gj::kThreads - 1) / gj::kThreads;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gj::k_symptoms_stats, dim3((unsigned)blocks), dim3(gj::kThreads), 0, (hipStream_t)stream, S, R);
  return gj::launch_status();
}

int gj_step(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params, const gj_step_io* io,
            void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  gj::Groups G;
  rc = gj::group_networks(plan, params, &G);
  if (rc) return rc;
  rc = gj::check_state(plan, state, params);
  if (rc) return rc;
  rc = gj::do_transmission(plan, state, params, (hipStream_t)stream);
  if (rc) return rc;
  if (plan->tiled) {
    rc = gj::tiled_scatter(plan, state, params, G, (hipStream_t)stream);
    if (rc) return rc;
    rc = gj::tiled_venues(plan, state, params, G, 0, (hipStream_t)stream);
    if (rc) return rc;
    return gj::tiled_agents(plan, state, params, G, io, 1, (hipStream_t)stream);
  }
  rc = gj::do_venue_reduce(plan, state, params, G, (hipStream_t)stream);
  if (rc) return rc;
  return gj::do_agent_gather(plan, state, params, G, io, 1, (hipStream_t)stream);
}

int gj_step_phase(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
                  const gj_step_io* io, int phase, void* stream) {
  int rc = gj::check_plan(plan);
  if (rc) return rc;
  gj::Groups G;
  rc = gj::group_networks(plan, params, &G);
  if (rc) return rc;
  rc = gj::check_state(plan, state, params, /*full=*/phase == 0 || phase == 3);   // only a1 / a9 touch the infection state
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (phase) {
    case 0: return gj::do_transmission(plan, state, params, st);
    case 1: return plan->tiled ? gj::tiled_scatter(plan, state, params, G, st)
                               : gj::do_venue_reduce(plan, state, params, G, st);
    case 2: return plan->tiled ? gj::tiled_venues(plan, state, params, G, 0, st) : GJ_OK;
    case 3: return plan->tiled ? gj::tiled_agents(plan, state, params, G, io, 1, st)
                               : gj::do_agent_gather(plan, state, params, G, io, 1, st);
    case 4: return plan->tiled ? gj::tiled_agents(plan, state, params, G, io, 0, st)
                               : gj::do_agent_gather(plan, state, params, G, io, 0, st);
    case 5: return plan->tiled ? gj::tiled_venues(plan, state, params, G, 1, st) : GJ_OK;
    case 6: return plan->tiled ? gj::tiled_venues(plan, state, params, G, 2, st) : GJ_OK;
    case 7:   // 1 then 5 (A + B) in one call: the launch path of a partial-sum group
    case 8:   // 1 then 2 (A + B + C)
      if (!plan->tiled) return gj::do_venue_reduce(plan, state, params, G, st);
      rc = gj::tiled_scatter(plan, state, params, G, st);
      if (rc) return rc;
      return gj::tiled_venues(plan, state, params, G, phase == 7 ? 1 : 0, st);
    default: return GJ_E_RANGE;
  }
}

namespace gj {
__global__ void k_clock_advance(gj_clock* clock, double delta_now) {
  const uint64_t step = clock->step + 1;
  clock->step = step;
  clock->now = (float)(clock->now0 + (double)(int64_t)(step - clock->step0) * delta_now);
}
}  // namespace gj

int gj_clock_advance(gj_clock* clock, double delta_now, void* stream) {
  if (!clock) return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_clock_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, clock, delta_now);
  return gj::launch_status();
}

int gj_pack_f32(int64_t n, const int32_t* index, const float* src, float* out, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!index || !src || !out) return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_pack, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)), dim3(gj::kThreads), 0,
                     (hipStream_t)stream, n, index, src, out);
  return gj::launch_status();
}

int gj_unpack_f32(int64_t n, const int32_t* index, const float* in, float* dst, void* stream) {
  if (n < 0) return GJ_E_RANGE;
  if (n == 0) return GJ_OK;
  if (!index || !in || !dst) return GJ_E_NULL;
  hipLaunchKernelGGL(gj::k_unpack, dim3((unsigned)((n + gj::kThreads - 1) / gj::kThreads)), dim3(gj::kThreads), 0,
                     (hipStream_t)stream, n, index, in, dst);
  return gj::launch_status();
}

int gj_event_create(void** event) {
  if (!event) return GJ_E_NULL;
  hipEvent_t e;
  const hipError_t rc = hipEventCreate(&e);
  if (rc != hipSuccess) return (int)rc;
  *event = (void*)e;
  return GJ_OK;
}
int gj_event_record(void* event, void* stream) {
  if (!event) return GJ_E_NULL;
  return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
}
int gj_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return GJ_E_NULL;
  hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
int gj_event_destroy(void* event) {
  if (!event) return GJ_E_NULL;
  return (int)hipEventDestroy((hipEvent_t)event);
}

}  // extern "C"
