// Tiled ("propagation-blocked") kernels: phases A-D of include/gradjune_hip.h, struct gj_tiled.
// Every random access hits LDS; HBM sees only coalesced streams.  Included by gradjune_hip.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gradjune_hip.h"
#include "gj_device.h"

namespace gj {

constexpr int kTileThreads = 1024;  // 16 waves: one workgroup per CU when the slice fills LDS
constexpr int kTileWaves = kTileThreads / kWave;

struct TSetA {            // what phases A and D need of one set
  const uint16_t* a_la;
  const int32_t* tile_sptr;
  const int32_t* tile_jpos;
  float* val;
  int32_t J;
  int32_t active;         // networks active on the set in this step (0: skip)
  int32_t raw;            // 1: reads raw transmission / susceptibility weight (household)
  int32_t _pad;
};

struct TileAArgs {
  TSetA sets[GJ_MAX_SETS];
  int32_t n_sets;
  int32_t slice_agents;
  int64_t n_agents;
  const float* trans;
  const float* qtrans;    // == trans when no quarantine collection
};

// ---- phase A: scatter the slice's transmissions to every edge, in block-major tile order -------
__device__ __forceinline__ void load_slice(float* lds, const float* __restrict__ src, int64_t base, int n_local,
                                           int tid) {
  const int n4 = n_local >> 2;   // base is a multiple of 64 floats: 16-byte aligned
  const float4* s4 = reinterpret_cast<const float4*>(src + base);
  float4* d4 = reinterpret_cast<float4*>(lds);
  for (int i = tid; i < n4; i += kTileThreads) d4[i] = s4[i];
  for (int i = (n4 << 2) + tid; i < n_local; i += kTileThreads) lds[i] = src[base + i];
}

__global__ __launch_bounds__(kTileThreads) void k_tile_scatter(const TileAArgs A) {
  extern __shared__ __align__(16) float lds_x[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
  const int s = blockIdx.x;
  const int64_t base = (int64_t)s * A.slice_agents;
  const int n_local = (int)min((int64_t)A.slice_agents, A.n_agents - base);
  const bool two_sources = A.qtrans != A.trans;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 0: sets that read q*transmission (or everything when there is one source); pass 1: raw sets
    if (pass == 1 && !two_sources) break;
    bool any = false;
    for (int t = 0; t < A.n_sets; ++t)
      if (A.sets[t].active && (!two_sources || A.sets[t].raw == pass)) any = true;
    if (!any) continue;
    if (pass == 1) __syncthreads();
    load_slice(lds_x, pass == 0 ? A.qtrans : A.trans, base, n_local, tid);
    __syncthreads();
    for (int t = 0; t < A.n_sets; ++t) {
      const TSetA& T = A.sets[t];
      if (!T.active || (two_sources && T.raw != pass)) continue;
      const int row = s * T.J;
      for (int j = wave; j < T.J; j += kTileWaves) {
        const int a = T.tile_sptr[row + j], b = T.tile_sptr[row + j + 1];
        const int p = T.tile_jpos[row + j] - a;
        for (int i = a + lane; i < b; i += kWave) T.val[p + i] = lds_x[T.a_la[i]];
      }
    }
  }
}

// ---- phases B + C: per venue block, LDS sums; cum = beta * p_contact * sums; gather back ---------
struct TSetB {
  const int32_t* blk_v0;
  const int32_t* blk_e0;
  const uint16_t* e_lv;
  const uint8_t* e_cls;
  float* val;
  const float* v_pc;
  float* cum;
  int32_t stride;
  int32_t nk;                              // active networks on the set
  float beta[GJ_MAX_NETS_PER_SET];
  int32_t table[GJ_MAX_NETS_PER_SET];      // leisure table index or -1
  int32_t age75[GJ_MAX_NETS_PER_SET];      // susceptibility additionally * (age > 75)
  int32_t leisure;
  int32_t _pad;
};

struct TileBArgs {
  TSetB sets[GJ_MAX_SETS];
  const int32_t* work;     // (set, block) pairs
  const float* tables;
  int32_t day_type;
  int32_t mode;            // 0: B then C (fused);  1: B only (cum written);  2: C only (cum read)
};

// segmented (by run of equal keys) inclusive sum across the wave; returns true on the last lane of a run
__device__ __forceinline__ bool run_sum(int key, float& x, int lane) {
  const int prev = __shfl_up(key, 1, kWave);
  int head = (lane == 0) || (prev != key);
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const float y = __shfl_up(x, off, kWave);
    const int hy = __shfl_up(head, off, kWave);
    if (lane >= off && !head) {
      x += y;
      head |= hy;
    }
  }
  const int next = __shfl_down(key, 1, kWave);
  return (lane == kWave - 1) || (next != key);
}

__global__ __launch_bounds__(kTileThreads) void k_tile_venues(const TileBArgs B) {
  extern __shared__ __align__(16) float lds_s[];
  const int tid = threadIdx.x, lane = tid % kWave;
  const int set = B.work[2 * blockIdx.x], j = B.work[2 * blockIdx.x + 1];
  const TSetB& T = B.sets[set];
  const int nk = T.nk;
  if (nk == 0) return;
  const int v0 = T.blk_v0[j], nv = T.blk_v0[j + 1] - v0;
  const int e0 = T.blk_e0[j], e1 = T.blk_e0[j + 1];
  float* sums = lds_s;                       // [nk][nv]
  float* tabs = lds_s + (size_t)nk * nv;     // [nk][200] pass-1 tables, then [nk][200] pass-2 weights
  if (T.leisure) {
    for (int i = tid; i < nk * 200; i += kTileThreads) {
      const int k = i / 200, c = i % 200;
      const float l = B.tables[(int64_t)T.table[k] * GJ_TABLE_SIZE + B.day_type * 200 + c];
      tabs[i] = l;
      tabs[nk * 200 + i] = T.age75[k] ? l * (((c % 100) > 75) ? 1.0f : 0.0f) : l;
    }
  }
  if (B.mode != 2) {
    for (int i = tid; i < nk * nv; i += kTileThreads) sums[i] = 0.0f;
    __syncthreads();
    // B: stream the block's edges; whole waves stay converged (uniform trip count) for the shuffles
    const int n_iter = (e1 - e0 + kTileThreads - 1) / kTileThreads;
    for (int it = 0; it < n_iter; ++it) {
      const int i = e0 + it * kTileThreads + tid;
      const bool ok = i < e1;
      const int lv = ok ? (int)T.e_lv[i] : -1 - lane;
      float x = ok ? T.val[i] : 0.0f;
      if (!T.leisure) {
        const bool tail = run_sum(lv, x, lane);
        if (ok && tail) atomicAdd(&sums[lv], x);
      } else if (ok) {
        const int c = T.e_cls[i];
        for (int k = 0; k < nk; ++k) atomicAdd(&sums[k * nv + lv], tabs[k * 200 + c] * x);
      }
    }
    __syncthreads();
    for (int i = tid; i < nk * nv; i += kTileThreads) {
      const int k = i / nv, lv = i - k * nv;
      const float c = (T.beta[k] * T.v_pc[v0 + lv]) * sums[i];
      T.cum[(int64_t)(v0 + lv) * T.stride + k] = c;
      sums[i] = c;
    }
    if (B.mode == 1) return;
  } else {
    for (int i = tid; i < nk * nv; i += kTileThreads) {
      const int k = i / nv, lv = i - k * nv;
      sums[i] = T.cum[(int64_t)(v0 + lv) * T.stride + k];
    }
  }
  __syncthreads();
  // C: per edge, the venue's cum (leisure: weighted over the set's networks by the agent's class)
  for (int i = e0 + tid; i < e1; i += kTileThreads) {
    const int lv = T.e_lv[i];
    float r;
    if (!T.leisure) {
      r = sums[lv];
    } else {
      const int c = T.e_cls[i];
      r = 0.0f;
      for (int k = 0; k < nk; ++k) r += tabs[nk * 200 + k * 200 + c] * sums[k * nv + lv];
    }
    T.val[i] = r;
  }
}

// ---- phase D: per slice, accumulate the edges' values per agent in LDS; epilogue a7-a9 -----------
struct TileDArgs {
  TSetA sets[GJ_MAX_SETS];
  int32_t n_sets;
  int32_t slice_agents;
  int64_t n_agents;
  const float* stage;
  float* susceptibility;
  float* is_infected;
  float* infection_time;
  float* not_infected_probs;
  float* new_infected;
  float* trans_susc;
  const float* exp_noise;
  float now, dt, q_thr;
  int32_t has_q, sample;
  uint64_t seed, step;
  int64_t agent_offset;
};

__global__ __launch_bounds__(kTileThreads) void k_tile_agents(const TileDArgs D) {
  extern __shared__ __align__(16) float lds_acc[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
  const int s = blockIdx.x;
  const int64_t base = (int64_t)s * D.slice_agents;
  const int n_local = (int)min((int64_t)D.slice_agents, D.n_agents - base);
  for (int i = tid; i < n_local; i += kTileThreads) lds_acc[i] = 0.0f;
  __syncthreads();
  // ts = susc * (q * sum over masked sets + sum over raw sets): masked sets first, scale by q, raw sets last
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = 0; t < D.n_sets; ++t) {
      const TSetA& T = D.sets[t];
      if (!T.active || T.raw != pass) continue;
      const int row = s * T.J;
      for (int j = wave; j < T.J; j += kTileWaves) {
        const int a = T.tile_sptr[row + j], b = T.tile_sptr[row + j + 1];
        const int p = T.tile_jpos[row + j] - a;
        for (int i = a + lane; i < b; i += kWave) atomicAdd(&lds_acc[T.a_la[i]], T.val[p + i]);
      }
    }
    __syncthreads();
    if (pass == 0 && D.has_q) {
      for (int i = tid; i < n_local; i += kTileThreads)
        lds_acc[i] = ((D.stage[base + i] < D.q_thr) ? 1.0f : 0.0f) * lds_acc[i];
      __syncthreads();
    }
  }
  for (int i = tid; i < n_local; i += kTileThreads) {
    const int64_t a = base + i;
    float susc = D.susceptibility[a];
    float ts = susc * lds_acc[i];
    if (D.trans_susc) D.trans_susc[a] = ts;
    ts = fminf(fmaxf(ts, 1e-6f), 100.0f);
    float p = expf(-ts * D.dt);
    p = fminf(fmaxf(p, 0.0f), 1.0f);
    if (D.not_infected_probs) D.not_infected_probs[a] = p;
    if (!D.sample) continue;
    float e0, e1;
    if (D.exp_noise) {
      e0 = D.exp_noise[a];
      e1 = D.exp_noise[D.n_agents + a];
    } else {
      exp_pair(D.seed, D.step, D.agent_offset + a, e0, e1);
    }
    const float nw = gumbel_new_infected(p, e0, e1);
    if (D.new_infected) D.new_infected[a] = nw;
    if (nw != 0.0f) {
      float inf = D.is_infected[a], t_inf = D.infection_time[a];
      infect(nw, D.now, susc, inf, t_inf);
      D.susceptibility[a] = susc;
      D.is_infected[a] = inf;
      D.infection_time[a] = t_inf;
    }
  }
}

}  // namespace gj
