// Tiled ("propagation-blocked") kernels: phases A-D of include/gradjune_hip.h, struct gj_tiled.
// Every random access hits LDS; HBM sees only coalesced streams.  Included by gradjune_hip.hip.
//
// All three kernels are pure streaming + LDS, so what matters is bytes in flight per CU: one
// workgroup of 16 waves owns up to 158 KiB of LDS (one per CU), and every lane keeps kUnroll
// independent loads in flight before it touches LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gradjune_hip.h"
#include "gj_device.h"

namespace gj {

constexpr int kTileThreads = 1024;  // 16 waves: one workgroup per CU when the slice fills LDS
constexpr int kTileWaves = kTileThreads / kWave;
constexpr int kUnroll = 8;          // 64-edge chunks a wave keeps in flight: phase D, and phase A with wide descriptors
#ifndef GJ_UNROLL_NARROW
#define GJ_UNROLL_NARROW 16
#endif
constexpr int kUnrollNarrow = GJ_UNROLL_NARROW;   // phase A with 16-byte descriptors (16 of them = one wave-wide load):
                                                  // 0.182 -> 0.149 ms on C3 against 8 (phase A is bound by loads in flight)
#ifndef GJ_VENUE_UNROLL
#define GJ_VENUE_UNROLL 2
#endif
constexpr int kVenueUnroll = GJ_VENUE_UNROLL;  // 8-slot groups a lane keeps in flight (phase B)
#ifndef GJ_VENUE_UNROLL_C
#define GJ_VENUE_UNROLL_C GJ_VENUE_UNROLL
#endif
constexpr int kVenueUnrollC = GJ_VENUE_UNROLL_C;  // same, phase C

// LDS float atomics run at 0.33 lanes/clk/CU on gfx950 (measured, tools/microbench/lds_atomics.hip)
// against 4.9 for ds_add_u64 and 7.3 for ds_add_u32, so the per-venue and per-agent sums are kept
// in 64-bit fixed point: integer adds are order-independent, which also makes both passes bitwise
// reproducible, and exact (no rounding inside a sum).
//   pass 1 (sums of transmissions per venue):  2^-40 resolution (9e-13), |sum| < 8.3e6
//   pass 2 (sums of cum per agent):            2^-36 resolution (1.5e-11), |sum| < 1.3e8
typedef unsigned long long fx_t;
template <int BITS>
__device__ __forceinline__ fx_t to_fx(float x) {
  constexpr float scale = (float)(1ull << BITS);
  constexpr float vmax = (float)(1ull << (62 - BITS));
  x = fminf(fmaxf(x, -vmax), vmax);                   // also maps NaN to -vmax: finite, flagged by tests
  return (fx_t)__float2ll_rn(x * scale);
}
template <int BITS>
__device__ __forceinline__ float from_fx(fx_t v) {
  return (float)((double)(long long)v * (1.0 / (double)(1ull << BITS)));
}
constexpr int kFxVenue = 40, kFxAgent = 36;

struct TSetA {            // what phases A and D need of one set
  const uint16_t* a_la;
  const int32_t* tile_sptr;
  const int32_t* tile_jpos;
  const int32_t* chunk_ptr;
  const int4* chunk_desc;
  float* val;
  int32_t J;
  int32_t active;         // networks active on the set in this step (0: skip)
  int32_t raw;            // 1: reads raw transmission / susceptibility weight (household)
  int32_t wide;           // chunk_desc holds two int4 per chunk (up to 6 tiles per chunk), see tiling.py
};

struct TileAArgs {
  TSetA sets[GJ_MAX_SETS];
  int32_t n_sets;
  int32_t slice_agents;
  int64_t n_agents;
  const float* trans;
  const float* qtrans;    // == trans when no quarantine collection
};

__device__ __forceinline__ void load_slice(float* lds, const float* __restrict__ src, int64_t base, int n_local,
                                           int tid) {
  const int n4 = n_local >> 2;   // base is a multiple of 64 floats: 16-byte aligned
  const float4* s4 = reinterpret_cast<const float4*>(src + base);
  float4* d4 = reinterpret_cast<float4*>(lds);
  for (int i = tid; i < n4; i += kTileThreads) d4[i] = s4[i];
  for (int i = (n4 << 2) + tid; i < n_local; i += kTileThreads) lds[i] = src[base + i];
}

// Block-major slot of lane `lane` of a 64-edge chunk (descriptor d: slot0, slot1, split | multi<<16, j0;
// see tiling.py).  Fast form: pure register arithmetic on a wave-uniform descriptor (straight-line
// code, so that a batch of chunks keeps all its loads in flight).
__device__ __forceinline__ int chunk_slot_fast(const int4 d, int lane) {
  const int split = d.z & 0xFFFF;
  return (lane < split) ? d.x + lane : d.y + (lane - split);
}
// General form for the rare chunk that spans more than two (tiny) tiles: walk the tile tables.
__device__ __noinline__ int chunk_slot_slow(const TSetA& T, const int4 d, int row, int i, int lane) {
  const int split = d.z & 0xFFFF;
  if (lane < split) return d.x + lane;
  int j = d.w;
  while (i >= T.tile_sptr[row + j + 1]) ++j;
  return T.tile_jpos[row + j] + (i - T.tile_sptr[row + j]);
}

// Wide descriptors (sets with small tiles): d0 = base_0..3, d1 = base_4, base_5, start_1..4 (bytes),
// start_5 | multi << 8 | j0 << 9.  Lanes [start_k, start_k+1) of the chunk map to slots base_k + lane.
__device__ __forceinline__ int chunk_slot_wide(const int4 d0, const int4 d1, int lane) {
  int b = d0.x;
  b = (lane >= (d1.z & 0xFF)) ? d0.y : b;
  b = (lane >= ((d1.z >> 8) & 0xFF)) ? d0.z : b;
  b = (lane >= ((d1.z >> 16) & 0xFF)) ? d0.w : b;
  b = (lane >= ((d1.z >> 24) & 0xFF)) ? d1.x : b;
  b = (lane >= (d1.w & 0xFF)) ? d1.y : b;
  return b + lane;
}
// More than six tiles in one chunk (tiles of a few edges): walk the tile tables from block j0.
__device__ __noinline__ int chunk_slot_walk(const TSetA& T, int row, int i, int j) {
  while (i >= T.tile_sptr[row + j + 1]) ++j;
  return T.tile_jpos[row + j] + (i - T.tile_sptr[row + j]);
}

// One batch of chunks of a set: the block-major slot of this lane's edge in each of them.  The batch's
// descriptors are consecutive in memory, so ONE coalesced load brings them in (lane l holds dword l of
// the batch); each chunk's words are then broadcast with v_readlane into scalars.  Straight-line, so the
// whole batch is in flight; a wave-uniform branch takes the table walk for batches that contain a chunk
// the descriptor cannot express.
template <bool WIDE, int U>
__device__ __forceinline__ int batch_desc_load(const TSetA& T, int c_base, int n_chunks, int c0, int lane) {
  constexpr int W = WIDE ? 8 : 4;                       // dwords per descriptor
  static_assert(U * W <= kWave, "a batch's descriptors must fit one wave-wide load");
  const int32_t* desc = reinterpret_cast<const int32_t*>(T.chunk_desc);
  const int u_l = min(lane / W, U - 1), k_l = lane % W;
  return desc[(int64_t)(c_base + min(c0 + u_l, n_chunks - 1)) * W + k_l];
}

template <bool WIDE, int U>
__device__ __forceinline__ void batch_slots(const TSetA& T, const int word, int row, int seg0, int seg1, int c0,
                                            int lane, int (&slot)[U]) {
  constexpr int W = WIDE ? 8 : 4;
  const int k_l = lane % W;
  const bool flag = WIDE ? (k_l == 7 && (word & 0x100)) : (k_l == 2 && (word >> 16));
  const bool any_multi = __builtin_amdgcn_ballot_w64(flag) != 0ull;
#define GJ_DW(u, k) __builtin_amdgcn_readlane(word, (u) * W + (k))
  if (!any_multi) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!WIDE) {
        slot[u] = chunk_slot_fast(make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), 0), lane);
      } else {
        slot[u] = chunk_slot_wide(make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), GJ_DW(u, 3)),
                                  make_int4(GJ_DW(u, 4), GJ_DW(u, 5), GJ_DW(u, 6), GJ_DW(u, 7)), lane);
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(seg0 + (c0 + u) * kWave + lane, seg1 - 1);
      if (!WIDE) {
        const int4 d = make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), GJ_DW(u, 3));
        slot[u] = (d.z >> 16) ? chunk_slot_slow(T, d, row, i, lane) : chunk_slot_fast(d, lane);
      } else {
        const int4 d0 = make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), GJ_DW(u, 3));
        const int4 d1 = make_int4(GJ_DW(u, 4), GJ_DW(u, 5), GJ_DW(u, 6), GJ_DW(u, 7));
        slot[u] = (d1.w & 0x100) ? chunk_slot_walk(T, row, i, (int)((unsigned)d1.w >> 9))
                                 : chunk_slot_wide(d0, d1, lane);
      }
    }
  }
#undef GJ_DW
}

// Phase A's inner loop for one set.  The kernel is bound by loads in flight (one workgroup of 16 waves per CU),
// so besides the 16-chunk batches the loop is software-pipelined: batch k+1's index / descriptor loads are
// issued before batch k's values are read from LDS and stored (vmcnt counts in order, so waiting for batch k
// leaves them in flight).  Two register sets alternate so that nothing is copied.
#ifndef GJ_SCATTER_PIPELINE
#define GJ_SCATTER_PIPELINE 1
#endif
template <bool WIDE, int kU>
__device__ __forceinline__ void scatter_batches(const TSetA& T, const float* lds_x, int s, int wave, int lane) {
  const int row = s * T.J;
  const int seg0 = T.tile_sptr[row], seg1 = T.tile_sptr[row + T.J];
  const int c_base = T.chunk_ptr[s];
  const int n_chunks = T.chunk_ptr[s + 1] - c_base;
  constexpr int kStride = kTileWaves * kU;
  auto stage1 = [&](int c, int& word, int (&la)[kU]) {        // descriptors + local agent indices of batch c
    const int cc = min(c, n_chunks - 1);                      // past the end: a harmless re-load of the last chunk
    word = batch_desc_load<WIDE, kU>(T, c_base, n_chunks, cc, lane);   // first: the slots wait on it
#pragma unroll
    for (int u = 0; u < kU; ++u) la[u] = T.a_la[min(seg0 + (cc + u) * kWave + lane, seg1 - 1)];
  };
  auto stage2 = [&](int c, int word, const int (&la)[kU]) {
    int slot[kU];
    batch_slots<WIDE, kU>(T, word, row, seg0, seg1, c, lane, slot);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c + u) * kWave + lane;
      if ((c + u < n_chunks) && (i < seg1)) T.val[slot[u]] = lds_x[la[u]];
    }
  };
  int c0 = wave * kU;
  if (c0 >= n_chunks) return;
  int laA[kU], wordA;
#if GJ_SCATTER_PIPELINE
  int laB[kU], wordB;
  stage1(c0, wordA, laA);
  while (true) {
    const int c1 = c0 + kStride;
    stage1(c1, wordB, laB);
    stage2(c0, wordA, laA);
    if (c1 >= n_chunks) break;
    const int c2 = c1 + kStride;
    stage1(c2, wordA, laA);
    stage2(c1, wordB, laB);
    if (c2 >= n_chunks) break;
    c0 = c2;
  }
#else
  for (; c0 < n_chunks; c0 += kStride) {
    stage1(c0, wordA, laA);
    stage2(c0, wordA, laA);
  }
#endif
}

// 16-chunk batches pay off when a slice has several hundred chunks of the set to stream (C3 at 10 M agents: 459,
// phase A 0.182 -> 0.149 ms); on short segments they lose (C2, 309 chunks: 0.041 -> 0.071 ms), so the batch size
// is chosen per slice and set (wave-uniform).
constexpr int kLongSegmentChunks = 384;
template <bool WIDE>
__device__ __forceinline__ void scatter_set(const TSetA& T, const float* lds_x, int s, int wave, int lane) {
  const int n_chunks = T.chunk_ptr[s + 1] - T.chunk_ptr[s];
  if (!WIDE && kUnrollNarrow != kUnroll && n_chunks >= kLongSegmentChunks) {
    scatter_batches<WIDE, WIDE ? kUnroll : kUnrollNarrow>(T, lds_x, s, wave, lane);
  } else {
    scatter_batches<WIDE, kUnroll>(T, lds_x, s, wave, lane);
  }
}

// Phase D's inner loop for one set.  (Measured, not adopted: software-pipelining this loop - the next batch's
// index / descriptor loads issued behind the current value loads - and 16 instead of 8 chunks per batch both
// left the kernel at 0.25 ms on C3: it is bound by the ~0.5 KB granularity of the per-tile value reads.)
template <bool WIDE>
__device__ __forceinline__ void gather_set(const TSetA& T, fx_t* lds_acc, int s, int wave, int lane) {
  const int row = s * T.J;
  const int seg0 = T.tile_sptr[row], seg1 = T.tile_sptr[row + T.J];
  const int c_base = T.chunk_ptr[s];
  const int n_chunks = T.chunk_ptr[s + 1] - c_base;
  constexpr int kU = kUnroll;
  for (int c0 = wave * kU; c0 < n_chunks; c0 += kTileWaves * kU) {
    int la[kU], slot[kU];
    float v[kU];
    const int word = batch_desc_load<WIDE, kU>(T, c_base, n_chunks, c0, lane);   // first: the val loads wait on it
#pragma unroll
    for (int u = 0; u < kU; ++u)    // unconditional (clamped) loads: straight-line, all in flight
      la[u] = T.a_la[min(seg0 + (c0 + u) * kWave + lane, seg1 - 1)];
    batch_slots<WIDE, kU>(T, word, row, seg0, seg1, c0, lane, slot);
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // the slot depends on the position only: these loads overlap the ones above
      const int i = seg0 + (c0 + u) * kWave + lane;
      const bool ok = (c0 + u < n_chunks) && (i < seg1);
      v[u] = T.val[ok ? slot[u] : 0];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c0 + u) * kWave + lane;
      if ((c0 + u < n_chunks) && (i < seg1)) atomicAdd(&lds_acc[la[u]], to_fx<kFxAgent>(v[u]));
    }
  }
}

// ---- phase A: scatter the slice's transmissions to every edge, in block-major tile order -------
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
      if (T.wide) {
        scatter_set<true>(T, lds_x, s, wave, lane);
      } else {
        scatter_set<false>(T, lds_x, s, wave, lane);
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
  int32_t transpose;       // 1: pass-1 / pass-2 per-network tables exchanged (backward pass)
  int32_t _pad;
};

struct Slots8 {           // 8 consecutive block-major slots: 16 bytes of local venue indices
  uint32_t w[4];
  __device__ __forceinline__ int lv(int q) const { return (w[q >> 1] >> ((q & 1) * 16)) & 0xFFFF; }
};

__global__ __launch_bounds__(kTileThreads) void k_tile_venues(const TileBArgs B) {
  extern __shared__ __align__(16) float lds_s[];
  const int tid = threadIdx.x;
  const int set = B.work[2 * blockIdx.x], j = B.work[2 * blockIdx.x + 1];
  const TSetB& T = B.sets[set];
  const int nk = T.nk;
  if (nk == 0) return;
  const int v0 = T.blk_v0[j], nv = T.blk_v0[j + 1] - v0;
  const int g0 = T.blk_e0[j] >> 3, g1 = T.blk_e0[j + 1] >> 3;   // groups of 8 slots
  fx_t* sums = reinterpret_cast<fx_t*>(lds_s);                    // [nk][nv] fixed-point sums (phase B)
  float* cumf = lds_s;                                            // cum of (k, lv) at float index 2*(k*nv+lv) (phase C)
  float* tabs = lds_s + 2 * (size_t)nk * nv;  // [nk][200] pass-1 tables, then [nk][200] pass-2 weights
  if (T.leisure) {
    for (int i = tid; i < nk * 200; i += kTileThreads) {
      const int k = i / 200, c = i % 200;
      const float l = B.tables[(int64_t)T.table[k] * GJ_TABLE_SIZE + B.day_type * 200 + c];
      const float lw = T.age75[k] ? l * (((c % 100) > 75) ? 1.0f : 0.0f) : l;
      tabs[i] = B.transpose ? lw : l;                 // weights of the transmitting side (pass 1)
      tabs[nk * 200 + i] = B.transpose ? l : lw;      // weights of the receiving side (pass 2)
    }
  }
  const uint4* lv8 = reinterpret_cast<const uint4*>(T.e_lv);
  const uint2* cls8 = reinterpret_cast<const uint2*>(T.e_cls);
  float4* val4 = reinterpret_cast<float4*>(T.val);
  if (B.mode != 2) {
    for (int i = tid; i < nk * nv; i += kTileThreads) sums[i] = 0;
    __syncthreads();
    // B: each lane takes 8 consecutive slots (48 bytes), merges runs of one venue in registers and adds
    // each run to the block's LDS sums; kVenueUnroll such groups are loaded before the first is used
    auto add_group = [&](const uint4 raw, const float4 xa, const float4 xb, const uint2 craw) {
      const Slots8 L{{raw.x, raw.y, raw.z, raw.w}};
      const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      if (!T.leisure) {
        int cur = L.lv(0);
        float acc = x[0];
#pragma unroll
        for (int q = 1; q < 8; ++q) {
          const int lv = L.lv(q);
          if (lv == cur) {
            acc += x[q];
          } else {
            if (cur != 0xFFFF) atomicAdd(&sums[cur], to_fx<kFxVenue>(acc));
            cur = lv;
            acc = x[q];
          }
        }
        if (cur != 0xFFFF) atomicAdd(&sums[cur], to_fx<kFxVenue>(acc));
      } else {
        const uint32_t cw[2] = {craw.x, craw.y};
        for (int k = 0; k < nk; ++k) {
          const float* tk = tabs + k * 200;
          int cur = L.lv(0);
          float acc = tk[cw[0] & 0xFF] * x[0];
#pragma unroll
          for (int q = 1; q < 8; ++q) {
            const int lv = L.lv(q);
            const float xl = tk[(cw[q >> 2] >> ((q & 3) * 8)) & 0xFF] * x[q];
            if (lv == cur) {
              acc += xl;
            } else {
              if (cur != 0xFFFF) atomicAdd(&sums[k * nv + cur], to_fx<kFxVenue>(acc));
              cur = lv;
              acc = xl;
            }
          }
          if (cur != 0xFFFF) atomicAdd(&sums[k * nv + cur], to_fx<kFxVenue>(acc));
        }
      }
    };
    for (int g = g0 + tid; g < g1; g += kVenueUnroll * kTileThreads) {
      uint4 raw[kVenueUnroll];
      float4 xa[kVenueUnroll], xb[kVenueUnroll];
      uint2 craw[kVenueUnroll];
#pragma unroll
      for (int u = 0; u < kVenueUnroll; ++u) {     // clamped, unconditional: all loads in flight together
        const int gu = min(g + u * kTileThreads, g1 - 1);
        raw[u] = lv8[gu];
        xa[u] = val4[2 * gu];
        xb[u] = val4[2 * gu + 1];
        craw[u] = T.leisure ? cls8[gu] : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < kVenueUnroll; ++u)
        if (g + u * kTileThreads < g1) add_group(raw[u], xa[u], xb[u], craw[u]);
    }
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
      const float beta = T.beta[k];
      for (int lv = tid; lv < nv; lv += kTileThreads) {
        const float c = (beta * T.v_pc[v0 + lv]) * from_fx<kFxVenue>(sums[k * nv + lv]);
        T.cum[(int64_t)(v0 + lv) * T.stride + k] = c;
        cumf[2 * (k * nv + lv)] = c;      // low half of the lane's own 8-byte slot
      }
    }
    if (B.mode == 1) return;
  } else {
    for (int k = 0; k < nk; ++k)
      for (int lv = tid; lv < nv; lv += kTileThreads)
        cumf[2 * (k * nv + lv)] = T.cum[(int64_t)(v0 + lv) * T.stride + k];
  }
  __syncthreads();
  // C: per slot, the venue's cum (leisure: weighted over the set's networks by the agent's class)
  for (int g = g0 + tid; g < g1; g += kVenueUnrollC * kTileThreads) {
    uint4 raw[kVenueUnrollC];
    uint2 craw[kVenueUnrollC];
#pragma unroll
    for (int u = 0; u < kVenueUnrollC; ++u) {
      const int gu = min(g + u * kTileThreads, g1 - 1);
      raw[u] = lv8[gu];
      craw[u] = T.leisure ? cls8[gu] : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < kVenueUnrollC; ++u) {
      const int gu = g + u * kTileThreads;
      if (gu >= g1) continue;
      const Slots8 L{{raw[u].x, raw[u].y, raw[u].z, raw[u].w}};
      float r[8];
      if (!T.leisure) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int lv = L.lv(q);
          r[q] = (lv != 0xFFFF) ? cumf[2 * lv] : 0.0f;
        }
      } else {
        const uint32_t cw[2] = {craw[u].x, craw[u].y};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int lv = L.lv(q);
          const int c = (cw[q >> 2] >> ((q & 3) * 8)) & 0xFF;
          float a = 0.0f;
          if (lv != 0xFFFF)
            for (int k = 0; k < nk; ++k) a += tabs[nk * 200 + k * 200 + c] * cumf[2 * (k * nv + lv)];
          r[q] = a;
        }
      }
      val4[2 * gu] = make_float4(r[0], r[1], r[2], r[3]);
      val4[2 * gu + 1] = make_float4(r[4], r[5], r[6], r[7]);
    }
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
  float* acc_scratch;     // non-NULL: write the per-agent sums here and leave a7-a9 to k_tile_epilogue
};

// a7-a9 for one agent per lane, from the per-agent sums of phase D (split form)
__global__ __launch_bounds__(256) void k_tile_epilogue(const TileDArgs D) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= D.n_agents) return;
  float susc = D.susceptibility[a];
  float ts = susc * D.acc_scratch[a];
  if (D.trans_susc) D.trans_susc[a] = ts;
  ts = fminf(fmaxf(ts, 1e-6f), 100.0f);
  float p = expf(-ts * D.dt);
  p = fminf(fmaxf(p, 0.0f), 1.0f);
  if (D.not_infected_probs) D.not_infected_probs[a] = p;
  if (!D.sample) return;
  float e0, e1;
  if (D.exp_noise) {
    e0 = D.exp_noise[a];
    e1 = D.exp_noise[D.n_agents + a];
  } else {
    exp_pair(D.seed, D.step, D.agent_offset + a, e0, e1);
  }
  const float nw = D.exp_noise ? gumbel_new_infected(p, e0, e1) : ratio_new_infected(p, e0, e1);
  if (D.new_infected) D.new_infected[a] = nw;
  if (nw != 0.0f) {
    float inf = D.is_infected[a], t_inf = D.infection_time[a];
    infect(nw, D.now, susc, inf, t_inf);
    D.susceptibility[a] = susc;
    D.is_infected[a] = inf;
    D.infection_time[a] = t_inf;
  }
}

__global__ __launch_bounds__(kTileThreads) void k_tile_agents(const TileDArgs D) {
  extern __shared__ __align__(16) fx_t lds_acc[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
  const int s = blockIdx.x;
  const int64_t base = (int64_t)s * D.slice_agents;
  const int n_local = (int)min((int64_t)D.slice_agents, D.n_agents - base);
  for (int i = tid; i < n_local; i += kTileThreads) lds_acc[i] = 0;
  __syncthreads();
  // ts = susc * (q * sum over masked sets + sum over raw sets): masked sets first, scale by q, raw sets last
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = 0; t < D.n_sets; ++t) {
      const TSetA& T = D.sets[t];
      if (!T.active || T.raw != pass) continue;
      if (T.wide) {
        gather_set<true>(T, lds_acc, s, wave, lane);
      } else {
        gather_set<false>(T, lds_acc, s, wave, lane);
      }
    }
    __syncthreads();
    if (pass == 0 && D.has_q) {
      for (int i = tid; i < n_local; i += kTileThreads)
        if (!(D.stage[base + i] < D.q_thr)) lds_acc[i] = 0;
      __syncthreads();
    }
  }
  if (D.acc_scratch) {   // split form: hand the per-agent sums to k_tile_epilogue (runs at full occupancy)
    for (int i = tid; i < n_local; i += kTileThreads) D.acc_scratch[base + i] = from_fx<kFxAgent>(lds_acc[i]);
    return;
  }
  // Epilogue: each lane takes adjacent agent pairs (2j, 2j+1), kPairs of them per iteration with their loads
  // issued together; in Philox mode one block serves both agents of a pair.
  constexpr int kPairs = 2;
  const bool pair_aligned = ((D.agent_offset + base) & 1) == 0;   // local pairs are global pairs (else: per agent)
  for (int j0 = tid; 2 * j0 < n_local; j0 += kPairs * kTileThreads) {
    float susc_b[2 * kPairs], e0_b[2 * kPairs], e1_b[2 * kPairs];
#pragma unroll
    for (int u = 0; u < kPairs; ++u) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * (j0 + u * kTileThreads) + h;
        const bool ok = i < n_local;
        susc_b[2 * u + h] = ok ? D.susceptibility[base + i] : 0.0f;
        e0_b[2 * u + h] = (ok && D.sample && D.exp_noise) ? D.exp_noise[base + i] : 1.0f;
        e1_b[2 * u + h] = (ok && D.sample && D.exp_noise) ? D.exp_noise[D.n_agents + base + i] : 1.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < kPairs; ++u) {
      const int i_pair = 2 * (j0 + u * kTileThreads);
      if (i_pair >= n_local) continue;
      uint32_t r[4] = {0u, 0u, 0u, 0u};
      const bool block = D.sample && !D.exp_noise && pair_aligned;
      if (block) philox4x32_10((uint64_t)(D.agent_offset + base + i_pair) >> 1, D.step, D.seed, r);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = i_pair + h;
        if (i >= n_local) continue;
        const int64_t a = base + i;
        float susc = susc_b[2 * u + h];
        float ts = susc * from_fx<kFxAgent>(lds_acc[i]);
        if (D.trans_susc) D.trans_susc[a] = ts;
        ts = fminf(fmaxf(ts, 1e-6f), 100.0f);
        float p = expf(-ts * D.dt);
        p = fminf(fmaxf(p, 0.0f), 1.0f);
        if (D.not_infected_probs) D.not_infected_probs[a] = p;
        if (!D.sample) continue;
        float e0 = e0_b[2 * u + h], e1 = e1_b[2 * u + h];
        if (block) {
          exp_from_block(r, h, e0, e1);
        } else if (!D.exp_noise) {
          exp_pair(D.seed, D.step, D.agent_offset + a, e0, e1);
        }
        const float nw = D.exp_noise ? gumbel_new_infected(p, e0, e1) : ratio_new_infected(p, e0, e1);
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
  }
}

}  // namespace gj
