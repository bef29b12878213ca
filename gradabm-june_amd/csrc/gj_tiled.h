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
constexpr int kMaxSliceAgents = 19840;   // 8-byte sums + two flag bits per agent within 160 KiB of LDS
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
#ifndef GJ_CUM_BATCH
#define GJ_CUM_BATCH 8
#endif
constexpr int kCumBatch = GJ_CUM_BATCH;        // venues per lane whose p_contact loads are in flight together (cum write-out)
#ifndef GJ_DMA_WIDE
#define GJ_DMA_WIDE 0
#endif

// LDS float atomics run at 0.33 lanes/clk/CU on gfx950 (measured, tools/microbench/lds_atomics.hip)
// against 4.9 for ds_add_u64 and 7.3 for ds_add_u32, so the per-venue and per-agent sums are kept
// in 64-bit fixed point: integer adds are order-independent, which also makes both passes bitwise
// reproducible, and exact (no rounding inside a sum).
//   pass 1 (sums of transmissions per venue):  2^-36 resolution (1.5e-11), |value| <= 16384, |sum| < 1.3e8
//   pass 2 (sums of cum per agent):            2^-32 resolution (2.3e-10), |value| <= 2.6e5, |sum| < 2.1e9
// float -> fixed point is the classic magic-number conversion in fp64 (two instructions + a 64-bit subtract instead
// of the dozen a float -> int64 conversion compiles to; phase D issues instructions, it does not wait for memory):
// x * 2^BITS + 1.5 * 2^52 is exact up to the final round-to-nearest-even at integer granularity, and the double's bit
// pattern minus the magic's is the integer in two's complement, for |x * 2^BITS| < 2^50.
typedef unsigned long long fx_t;
template <int BITS>
__device__ __forceinline__ constexpr float fx_max() { return (float)(1ull << (50 - BITS)); }
template <int BITS>
__device__ __forceinline__ fx_t to_fx(float x) {      // |x| <= fx_max<BITS>() (fx_add checks)
  constexpr double scale = (double)(1ull << BITS);
  constexpr double magic = 6755399441055744.0;       // 1.5 * 2^52
  const double d = __builtin_fma((double)x, scale, magic);
  return (fx_t)(__double_as_longlong(d) - 0x4338000000000000LL);
}
template <int BITS>
__device__ __forceinline__ float from_fx(fx_t v) {
  return (float)((double)(long long)v * (1.0 / (double)(1ull << BITS)));
}
// What cannot be summed as an integer raises a flag of its element instead of adding (round 4: two flags per element,
// so that a finite value beyond the window SATURATES instead of poisoning - the reference has no window: its sums stay
// finite and the epilogue clamps them, base.py:136-138):
//   x >  window (also +inf)  -> flag POS;   x < -window (also -inf) -> flag NEG;   NaN -> both.
// An element reads back as +-kSaturated (1e30: finite, so that a zero factor downstream - p_contact of an empty venue, a
// susceptibility of 0, a class weight of 0 - still gives 0 as it does in the reference, and large enough that whatever
// is left saturates every later window and ends at the epilogue's clamp: exp(-100 dt)), or NaN with both flags.
// Flags are bit arrays `flags[0 .. words)` = POS and `flags[words .. 2 words)` = NEG.
// Branch-free on the common path: control flow around every add lets hipcc sink each global load that feeds one into
// the add's own block, behind an s_waitcnt of its own - a batch of eight loads then costs eight round trips, not one.
constexpr float kSaturated = 1.0e30f;
__device__ __forceinline__ void fx_flag(uint32_t* flags, int words, int i, float x) {
  // (only ever called with |x| beyond the window or x not a number: a zero would raise both flags)
  const uint32_t bit = 1u << (i & 31);
  if (!(x < 0.0f)) atomicOr(&flags[i >> 5], bit);             // positive, +inf or NaN
  if (!(x > 0.0f)) atomicOr(&flags[words + (i >> 5)], bit);   // negative, -inf or NaN
}
template <int BITS>
__device__ __forceinline__ void fx_add(fx_t* sums, uint32_t* flags, int words, int i, float x) {
  const bool ok = fabsf(x) <= fx_max<BITS>();
  atomicAdd(&sums[i], to_fx<BITS>(ok ? x : 0.0f));
  if (__builtin_expect(!ok, 0)) fx_flag(flags, words, i, x);
}
__device__ __forceinline__ float fx_special(uint32_t pos, uint32_t neg, float v) {
  return (pos & neg) ? __builtin_nanf("") : (pos ? kSaturated : (neg ? -kSaturated : v));
}
template <int BITS>
__device__ __forceinline__ float fx_read(const fx_t* sums, const uint32_t* flags, int words, int i) {
  const uint32_t pos = (flags[i >> 5] >> (i & 31)) & 1u, neg = (flags[words + (i >> 5)] >> (i & 31)) & 1u;
  return fx_special(pos, neg, from_fx<BITS>(sums[i]));
}
constexpr int kFxVenue = 36, kFxAgent = 32;

// Element `idx` of a wave-uniform array through a 32-bit BYTE offset (scalar base + vector offset addressing: no
// 64-bit address arithmetic per lane - phases A and D issue instructions, they do not wait).  The array must be
// smaller than 4 GiB (check_tiled).
template <typename T>
__device__ __forceinline__ T at32(const T* base, int idx) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint64_t)((uint32_t)idx * (uint32_t)sizeof(T)));
}
template <typename T>
__device__ __forceinline__ void put32(T* base, int idx, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint64_t)((uint32_t)idx * (uint32_t)sizeof(T))) = v;
}

// Non-temporal accesses for the streams that are written once and read once per step and are larger than the caches:
// the per-edge workspace `val` (B's loads, C's stores, D's loads), the ELL rows of the direct form, the transmission
// kernel's parameter arrays.  Measured per stream (tools/ab.py, round 3, one call each): together -12 ... -24 us per C3
// step (venue launch -10 %, transmission -10 %).  NOT for phase A's 4-byte scattered stores (partial lines: 118 -> 176 us,
// although the venue launch that reads them gains 33 us), nor for the 2-byte index streams, whose unaligned 64-element
// pieces share cache lines between consecutive loads (a_la in phase A: 118 -> 134 us).
typedef float gj_v4f __attribute__((ext_vector_type(4)));
typedef unsigned int gj_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int gj_v2u __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T at32nt(const T* base, int idx) {
  return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint64_t)((uint32_t)idx * (uint32_t)sizeof(T))));
}
template <typename T>
__device__ __forceinline__ void put32nt(T* base, int idx, T v) {
  __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint64_t)((uint32_t)idx * (uint32_t)sizeof(T))));
}
__device__ __forceinline__ uint4 load_nt(const uint4* p) {
  const gj_v4u v = __builtin_nontemporal_load(reinterpret_cast<const gj_v4u*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 load_nt(const float4* p) {
  const gj_v4f v = __builtin_nontemporal_load(reinterpret_cast<const gj_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_nt(float4* p, float4 x) {
  gj_v4f v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<gj_v4f*>(p));
}
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
  int32_t wide;           // 1: chunk_desc holds two int4 per chunk (up to 6 tiles per chunk), see tiling.py;
                          // 2: no descriptors - chunk_desc is int32 [E], the block-major slot of every slice-major edge
  int32_t direct;         // pass 2 of the set is taken by phase D's direct form (TDirect): no val / a_la reads there
  int32_t presum;         // pass 1 of the set is taken by k_tile_presum: phase A has nothing to scatter
  const int32_t* multi_slots;   // rows of 64 explicit slots for the chunks a descriptor cannot express, or NULL (walk the tables)
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
  // all of a lane's loads are issued before the first LDS write: as a rolled loop (load, wait, write) the slice arrived
  // in five dependent memory round trips at the head of every workgroup
  constexpr int kQ = (kMaxSliceAgents + 4 * kTileThreads - 1) / (4 * kTileThreads);
  if (n4 > 0) {
    float4 v[kQ];
#pragma unroll
    for (int u = 0; u < kQ; ++u) v[u] = s4[min(tid + u * kTileThreads, n4 - 1)];      // clamped, unconditional
#pragma unroll
    for (int u = 0; u < kQ; ++u)
      if (tid + u * kTileThreads < n4) d4[tid + u * kTileThreads] = v[u];
  }
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
  return at32(desc, (c_base + min(c0 + u_l, n_chunks - 1)) * W + k_l);
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
  } else if (T.multi_slots) {
    // A batch that holds a chunk its descriptor cannot express (more tiles than segments): that chunk's lanes take
    // their slots from row j0 of multi_slots.  Branch-free: EVERY chunk of the batch issues the load (row 0 where it is
    // not needed) before the first select, so the batch's loads are in flight together - under a per-chunk branch hipcc
    // waits for each one with vmcnt(0), which also drains the software pipeline's loads of the next batch.
    int ms[U];
    bool mu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int w = WIDE ? GJ_DW(u, 7) : GJ_DW(u, 2);
      mu[u] = WIDE ? ((w & 0x100) != 0) : ((w >> 16) != 0);
      const int j0 = WIDE ? (int)((unsigned)w >> 9) : GJ_DW(u, 3);
      ms[u] = at32(T.multi_slots, (mu[u] ? j0 : 0) * kWave + lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int f;
      if (!WIDE) {
        f = chunk_slot_fast(make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), 0), lane);
      } else {
        f = chunk_slot_wide(make_int4(GJ_DW(u, 0), GJ_DW(u, 1), GJ_DW(u, 2), GJ_DW(u, 3)),
                            make_int4(GJ_DW(u, 4), GJ_DW(u, 5), GJ_DW(u, 6), GJ_DW(u, 7)), lane);
      }
      slot[u] = mu[u] ? ms[u] : f;
    }
  } else {
    // a plan without the rows (rounds 1-3): the lanes of such a chunk walk the tile tables
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
template <bool WIDE, int kU, bool NT = false>
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
    for (int u = 0; u < kU; ++u) la[u] = at32(T.a_la, min(seg0 + (cc + u) * kWave + lane, seg1 - 1));
  };
  auto stage2 = [&](int c, int word, const int (&la)[kU]) {
    int slot[kU];
    batch_slots<WIDE, kU>(T, word, row, seg0, seg1, c, lane, slot);
#ifdef GJ_DIAG_EXTRA_VALU      // diagnostics (tools/ab.py): N no-op VALU instructions per edge - is the launch bound by instruction issue?
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int r = 0; r < GJ_DIAG_EXTRA_VALU; ++r) asm volatile("v_add_u32 %0, %0, 0" : "+v"(slot[u]));
#endif
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c + u) * kWave + lane;
      if ((c + u < n_chunks) && (i < seg1)) {
        if (NT) put32nt(T.val, slot[u], lds_x[la[u]]); else put32(T.val, slot[u], lds_x[la[u]]);
      }
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
#ifndef GJ_NT_TILE_EDGES
#define GJ_NT_TILE_EDGES 1024
#endif
constexpr int kNtTileEdges = GJ_NT_TILE_EDGES;   // (64 with tiles padded to 16 edges: tiling.build_tiled(tile_pad=16), experiment)
template <bool WIDE>
__device__ __forceinline__ void scatter_set(const TSetA& T, const float* lds_x, int s, int wave, int lane) {
  const int n_chunks = T.chunk_ptr[s + 1] - T.chunk_ptr[s];
  // Non-temporal stores where this slice's tiles of the set are long (mean >= kNtTileEdges edges): a tile is one
  // contiguous run of `val`, and the partial lines at the ends of a run's pieces are what makes such stores slow
  // (every set non-temporal: phase A 118 -> 176 us; C3's schools / universities / leisure / care homes only: -3 us, and
  // -2 us in the venue launch that reads them - its loads no longer meet phase A's dirty lines on their way out of L2)
  if (!WIDE && kUnrollNarrow != kUnroll && n_chunks >= kLongSegmentChunks && n_chunks * kWave >= kNtTileEdges * T.J) {
    scatter_batches<WIDE, WIDE ? kUnroll : kUnrollNarrow, true>(T, lds_x, s, wave, lane);
  } else if (!WIDE && kUnrollNarrow != kUnroll && n_chunks >= kLongSegmentChunks) {
    scatter_batches<WIDE, WIDE ? kUnroll : kUnrollNarrow>(T, lds_x, s, wave, lane);
  } else {
    scatter_batches<WIDE, kUnroll>(T, lds_x, s, wave, lane);
  }
}

// Phase D's inner loop for one set.  (Measured, not adopted: software-pipelining this loop - the next batch's
// index / descriptor loads issued behind the current value loads - and 16 instead of 8 chunks per batch both
// left the kernel at 0.25 ms on C3: it is bound by the ~0.5 KB granularity of the per-tile value reads.)
#ifndef GJ_UNROLL_D_NARROW
#define GJ_UNROLL_D_NARROW 8       // chunks in flight per wave on sets with 16-byte descriptors (16: 8 % slower, registers)
#endif
#ifndef GJ_GATHER_PIPELINE
#define GJ_GATHER_PIPELINE 1
#endif
template <bool WIDE>
__device__ __forceinline__ void gather_set(const TSetA& T, fx_t* lds_acc, uint32_t* lds_flags, int flag_words, int s,
                                           int wave, int lane) {
  const int row = s * T.J;
  const int seg0 = T.tile_sptr[row], seg1 = T.tile_sptr[row + T.J];
  const int c_base = T.chunk_ptr[s];
  const int n_chunks = T.chunk_ptr[s + 1] - c_base;
  constexpr int kU = WIDE ? kUnroll : GJ_UNROLL_D_NARROW;
  constexpr int kStride = kTileWaves * kU;
  // A batch costs two dependent memory round trips (descriptors + local agent indices, then the values at the slots
  // the descriptors give), ~2 us each while every CU streams.  The next batch's first trip is issued before the
  // current batch's values are waited for (vmcnt counts in order), so a batch costs one.
  auto stage1 = [&](int c, int& word, int (&la)[kU]) {        // descriptors + local agent indices of batch c
    const int cc = min(c, n_chunks - 1);                      // past the end: a harmless re-load of the last chunk
    word = batch_desc_load<WIDE, kU>(T, c_base, n_chunks, cc, lane);
#pragma unroll
    for (int u = 0; u < kU; ++u) la[u] = at32(T.a_la, min(seg0 + (cc + u) * kWave + lane, seg1 - 1));
  };
  auto values = [&](int c, int word, float (&v)[kU]) {        // the batch's value loads (issued, not waited for)
    int slot[kU];
    batch_slots<WIDE, kU>(T, word, row, seg0, seg1, c, lane, slot);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c + u) * kWave + lane;
      const bool ok = (c + u < n_chunks) && (i < seg1);
      v[u] = at32nt(T.val, ok ? slot[u] : 0);
    }
  };
  auto add = [&](int c, const int (&la)[kU], const float (&v)[kU]) {
    // straight-line: every lane adds (0 where there is no edge; la is a valid local index either way).  Whether a value
    // can be summed at all - |x| <= 262144, not a NaN - is checked once per batch on the bit patterns (as unsigned
    // integers they order like the magnitudes, NaN and the infinities above every finite value); the batch that holds
    // such a value takes the per-element form and flags the agent
    constexpr uint32_t kLimit = 0x48800000u;            // bit pattern of fx_max<kFxAgent>() = 262144.0f
    static_assert(kFxAgent == 32, "kLimit is the pattern of 2^(50 - kFxAgent)");
    float x[kU];
    uint32_t m = 0u;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c + u) * kWave + lane;
      const bool edge = (c + u < n_chunks) && (i < seg1);
      x[u] = edge ? v[u] : 0.0f;
      m = max(m, __float_as_uint(x[u]) & 0x7FFFFFFFu);
    }
#ifdef GJ_DIAG_EXTRA_VALU
#pragma unroll
    for (int u = 0; u < kU; ++u)
#pragma unroll
      for (int r = 0; r < GJ_DIAG_EXTRA_VALU; ++r) asm volatile("v_add_f32 %0, %0, 0" : "+v"(x[u]));
#endif
    if (__builtin_expect(m <= kLimit, 1)) {
#pragma unroll
      for (int u = 0; u < kU; ++u) atomicAdd(&lds_acc[la[u]], to_fx<kFxAgent>(x[u]));
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if ((__float_as_uint(x[u]) & 0x7FFFFFFFu) <= kLimit) {
          atomicAdd(&lds_acc[la[u]], to_fx<kFxAgent>(x[u]));
        } else {
          fx_flag(lds_flags, flag_words, la[u], x[u]);
        }
      }
    }
  };
  int c0 = wave * kU;
  if (c0 >= n_chunks) return;
  int laA[kU], wordA;
  float v[kU];
#if GJ_GATHER_PIPELINE
  int laB[kU], wordB;
  stage1(c0, wordA, laA);
  while (true) {
    const int c1 = c0 + kStride;
    values(c0, wordA, v);
    stage1(c1, wordB, laB);           // behind the value loads: in flight while they are waited for and added
    add(c0, laA, v);
    if (c1 >= n_chunks) break;
    const int c2 = c1 + kStride;
    values(c1, wordB, v);
    stage1(c2, wordA, laA);
    add(c1, laB, v);
    if (c2 >= n_chunks) break;
    c0 = c2;
  }
#else
  for (; c0 < n_chunks; c0 += kStride) {
    stage1(c0, wordA, laA);
    values(c0, wordA, v);
    add(c0, laA, v);
  }
#endif
}

// Sets whose tiles hold a handful of edges (a rank's halo half of a heavy-tailed set, a 1e8-agent world in one
// partition): a 64-edge chunk then spans more tiles than a descriptor can express and every lane would walk the tile
// tables (dependent loads, the launch falls off a cliff - round 2 measured 1.48 ms for such a phase A next to 0.25 ms
// for its phase B).  Such a set carries the slot of every edge explicitly instead: 4 more bytes per edge and pass,
// coalesced, straight-line.
template <int kU>
__device__ __forceinline__ void scatter_explicit(const TSetA& T, const float* lds_x, int s, int wave, int lane) {
  const int row = s * T.J;
  const int seg0 = T.tile_sptr[row], seg1 = T.tile_sptr[row + T.J];
  const int n_chunks = T.chunk_ptr[s + 1] - T.chunk_ptr[s];
  const int32_t* slots = reinterpret_cast<const int32_t*>(T.chunk_desc);
  for (int c0 = wave * kU; c0 < n_chunks; c0 += kTileWaves * kU) {
    int la[kU], sl[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = min(seg0 + (c0 + u) * kWave + lane, seg1 - 1);
      la[u] = at32(T.a_la, i);
      sl[u] = at32(slots, i);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c0 + u) * kWave + lane;
      if ((c0 + u < n_chunks) && (i < seg1)) put32(T.val, sl[u], lds_x[la[u]]);
    }
  }
}

template <int kU>
__device__ __forceinline__ void gather_explicit(const TSetA& T, fx_t* lds_acc, uint32_t* lds_flags, int flag_words, int s,
                                                int wave, int lane) {
  const int row = s * T.J;
  const int seg0 = T.tile_sptr[row], seg1 = T.tile_sptr[row + T.J];
  const int n_chunks = T.chunk_ptr[s + 1] - T.chunk_ptr[s];
  const int32_t* slots = reinterpret_cast<const int32_t*>(T.chunk_desc);
  constexpr uint32_t kLimit = 0x48800000u;              // fx_max<kFxAgent>() = 262144.0f (see gather_set)
  for (int c0 = wave * kU; c0 < n_chunks; c0 += kTileWaves * kU) {
    int la[kU], sl[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = min(seg0 + (c0 + u) * kWave + lane, seg1 - 1);
      la[u] = at32(T.a_la, i);
      sl[u] = at32(slots, i);
    }
    float x[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) x[u] = at32(T.val, sl[u]);
    uint32_t m = 0u;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int i = seg0 + (c0 + u) * kWave + lane;
      x[u] = ((c0 + u < n_chunks) && (i < seg1)) ? x[u] : 0.0f;
      m = max(m, __float_as_uint(x[u]) & 0x7FFFFFFFu);
    }
    if (__builtin_expect(m <= kLimit, 1)) {
#pragma unroll
      for (int u = 0; u < kU; ++u) atomicAdd(&lds_acc[la[u]], to_fx<kFxAgent>(x[u]));
    } else {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if ((__float_as_uint(x[u]) & 0x7FFFFFFFu) <= kLimit) {
          atomicAdd(&lds_acc[la[u]], to_fx<kFxAgent>(x[u]));
        } else {
          fx_flag(lds_flags, flag_words, la[u], x[u]);
        }
      }
    }
  }
}

// ---- phase A: scatter the slice's transmissions to every edge, in block-major tile order -------
#ifndef GJ_SCATTER_WAVES_PER_SIMD
#define GJ_SCATTER_WAVES_PER_SIMD 4      // 8: two workgroups per CU (<= 64 VGPRs; the slice's 80 KB of LDS allow it)
#endif
__global__ __launch_bounds__(kTileThreads, GJ_SCATTER_WAVES_PER_SIMD) void k_tile_scatter(const TileAArgs A) {
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
      if (A.sets[t].active && !A.sets[t].presum && (!two_sources || A.sets[t].raw == pass)) any = true;
    if (!any) continue;
    if (pass == 1) __syncthreads();
    load_slice(lds_x, pass == 0 ? A.qtrans : A.trans, base, n_local, tid);
    __syncthreads();
    for (int t = 0; t < A.n_sets; ++t) {
      const TSetA& T = A.sets[t];
      if (!T.active || T.presum || (two_sources && T.raw != pass)) continue;
      if (T.wide == 2) {
        scatter_explicit<kUnroll>(T, lds_x, s, wave, lane);
      } else if (T.wide) {
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
  int32_t direct;                          // pass 2 runs in phase D's direct form: phase C has nothing to do
  uint32_t term_limit;                     // bit pattern of the largest |term| a venue sum takes (<= 16384): chosen per set
                                           // so that the largest venue's sum cannot leave the 64 bits (fill_set_b)
  int32_t _pad_b;
  // run form (gj_tiled_set.run_*): the primary edge of the agents [blk_r0[j], blk_r0[j+1]) - value x[a], venue pv_blk[a]
  const uint16_t* pv_blk;
  const int32_t* blk_r0;
  const float* x;                          // transmission or q * transmission [n_x], 16-byte aligned
  int64_t n_x;
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

// Phase B, a group of 8 slots that holds a value which cannot be summed (rare; kept out of line so that its registers
// do not count against the launch's two workgroups per CU).
__device__ __noinline__ void venue_group_slow(fx_t* sums, uint32_t* vflags, int words, uint32_t limit, int base_k, int l0,
                                              int l1, int l2, int l3, int l4, int l5, int l6, int l7, float x0, float x1,
                                              float x2, float x3, float x4, float x5, float x6, float x7) {
  const int lv[8] = {l0, l1, l2, l3, l4, l5, l6, l7};
  const float x[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (lv[q] == 0xFFFF) continue;
    if ((__float_as_uint(x[q]) & 0x7FFFFFFFu) <= limit) {      // |x| inside the set's window (<= 16384 = fx_max<kFxVenue>())
      atomicAdd(&sums[base_k + lv[q]], to_fx<kFxVenue>(x[q]));
    } else {
      fx_flag(vflags, words, base_k + lv[q], x[q]);
    }
  }
}

struct Slots8 {           // 8 consecutive block-major slots: 16 bytes of local venue indices
  uint32_t w[4];
  __device__ __forceinline__ int lv(int q) const { return (w[q >> 1] >> ((q & 1) * 16)) & 0xFFFF; }
};

#ifdef GJ_DIAG_STAMPS
// timing diagnostics (tools/venue_timeline.py): start / end time and XCD of every workgroup of the last venue launch
constexpr int kDiagVenueSlots = 8192;
__device__ unsigned long long gj_diag_venue[3 * kDiagVenueSlots];
struct VenueStamp {
  unsigned long long t0;
  __device__ VenueStamp() : t0(__builtin_amdgcn_s_memrealtime()) {}      // the constant 100 MHz counter: one clock for the chip
  __device__ ~VenueStamp() {                                            // (s_memtime counts per XCD, unsynchronised)
    if (threadIdx.x == 0 && blockIdx.x < kDiagVenueSlots) {
      gj_diag_venue[3 * blockIdx.x] = t0;
      gj_diag_venue[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
      gj_diag_venue[3 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;     // HW_REG_XCC_ID
    }
  }
};
#endif

// One workgroup per (set, venue block) work item, heaviest first.  (Measured with per-workgroup timestamps,
// tools/venue_timeline.py: 1 899 workgroups of 10-80 us on 512 slots, ~80 % of the slot-time used - a finished
// workgroup's slot idles for a few us until its successor's 16 waves are up.  Persistent workgroups that work through
// several items each closed those gaps and took as long: with every slot busy the items stretch, the launch is bound
// by the memory system at ~4.8 TB/s of measured traffic, not by the slots.)
#ifndef GJ_VENUE_WAVES_PER_SIMD
#define GJ_VENUE_WAVES_PER_SIMD 4      // 72 VGPRs, no scratch.  (8 = two workgroups per CU needs <= 64 VGPRs and spills 12 dwords
                                       // with the exact run merging: measured 207 vs 204 us - the second workgroup, worth 25 % in
                                       // round 1, no longer pays now that the direct form took half of phase C away)
#endif
__global__ __launch_bounds__(kTileThreads, GJ_VENUE_WAVES_PER_SIMD) void k_tile_venues(const TileBArgs B) {
  extern __shared__ __align__(16) float lds_s[];
#ifdef GJ_DIAG_STAMPS
  VenueStamp gj_stamp;
#endif
  const int tid = threadIdx.x;
  const int set = B.work[2 * blockIdx.x], j = B.work[2 * blockIdx.x + 1];
  const TSetB& T = B.sets[set];
  const int nk = T.nk;
  if (nk == 0) return;
  const int v0 = T.blk_v0[j], nv = T.blk_v0[j + 1] - v0;
  const int g0 = T.blk_e0[j] >> 3, g1 = T.blk_e0[j + 1] >> 3;   // groups of 8 slots
  fx_t* sums = reinterpret_cast<fx_t*>(lds_s);                    // [nk][nv] fixed-point sums (phase B)
  float* cumf = lds_s;                                            // cum of (k, lv) at float index 2*(k*nv+lv) (phase C)
  float* tabs = lds_s + 2 * ((size_t)nk * nv + 64);  // (64 scratch sums, one per lane of a wave, follow the sums)
                                                     // [nk][200] pass-1 tables, then [nk][200] pass-2 weights
  uint32_t* vflags = reinterpret_cast<uint32_t*>(tabs + (T.leisure ? 2 * nk * 200 : 0));   // two bits per sum (fx_flag)
  const int vwords = (nk * nv + 64 + 31) / 32;
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
    for (int i = tid; i < nk * nv + 64; i += kTileThreads) sums[i] = 0;
    for (int i = tid; i < 2 * vwords; i += kTileThreads) vflags[i] = 0u;
    __syncthreads();
    // B: each lane takes 8 consecutive slots (48 bytes), merges runs of one venue in registers and adds
    // each run to the block's LDS sums; kVenueUnroll such groups are loaded before the first is used
    // Straight-line: a run of one venue is summed left to right in a register and added where the run ends; every
    // slot position issues an add, the ones that are not the end of a run (or are padding) add 0 to a scratch slot of
    // the lane's own.  (Branches per slot made this launch issue three times the instructions: it is bound by
    // instruction issue, SQ_ACTIVE_INST_ANY x waves per SIMD ~ 0.8.)
    const int dummy = nk * nv + (tid & 63);
    // Every slot's value goes to fixed point FIRST and a run of one venue is merged as integers: exact, so a venue's sum
    // does not depend on where the block boundaries and the padding put its runs relative to the 8-slot groups - the
    // tile geometry (eb_target, sv_max, slices) cannot change a single bit of `cum`.  (Round 2 merged the runs in fp32
    // and converted the run totals; the launch is bound by the memory system, the extra integer adds are free.)
#if defined(GJ_DIAG_FLOAT_RUNS)      // round 2's form, kept for A/B timing only (tools/ab.py): runs merged in fp32
    auto run_sums = [&](const int (&lv)[8], const float (&x)[8], int base_k) {
      float s8 = x[0];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool last = (q == 7) || (lv[q + (q < 7)] != lv[q]);
        const bool take = last && lv[q] != 0xFFFF;
        fx_add<kFxVenue>(sums, vflags, vwords, take ? base_k + lv[q] : dummy, take ? s8 : 0.0f);
        if (q < 7) s8 = last ? x[q + 1] : s8 + x[q + 1];
      }
    };
#elif defined(GJ_DIAG_SLOT_ADDS)     // A/B: no run merging at all - every slot adds its own value to its venue
    auto run_sums = [&](const int (&lv)[8], const float (&x)[8], int base_k) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool valid = lv[q] != 0xFFFF;
        const bool ok = fabsf(x[q]) <= fx_max<kFxVenue>();
        atomicAdd(&sums[valid ? base_k + lv[q] : dummy], to_fx<kFxVenue>((ok && valid) ? x[q] : 0.0f));
        if (__builtin_expect(!ok && valid, 0)) fx_flag(vflags, vwords, base_k + lv[q], x[q]);
      }
    };
#else
    // Every slot's value goes to fixed point FIRST and a run of one venue is merged as integers: exact, so a venue's sum
    // does not depend on where the block boundaries and the padding put its runs relative to the 8-slot groups - the
    // tile geometry (eb_target, sv_max, slices) and the partition cannot change a single bit of `cum`.  (Round 2 merged
    // the runs in fp32 and converted the run totals: GJ_DIAG_FLOAT_RUNS, 8-18 us faster on C3 from box to box and not
    // exact.  Measured and ruled out as the cause of that gap, tools/ab.py: instruction count - this form has 12 per slot
    // against 19 -, zero / non-zero scratch adds, LDS operations in flight, registers reused behind an LDS operation.)
    // The range check is made ONCE per group on the bit patterns - as unsigned integers |x| <= 16384, NaN and the
    // infinities order like their patterns - and a slot that is not the end of a run adds whatever the running sum is
    // to the lane's scratch sum instead of selecting a zero.
    auto run_sums = [&](const int (&lv)[8], const float (&x)[8], int base_k) {
      // (wave-uniform, at most the pattern of fx_max<kFxVenue>() = 16384.0f; smaller for a set with venues of more than
      // 4 096 attendees, so that attendees x window stays inside the 64-bit sum: no sum can wrap)
      const uint32_t kLimit = T.term_limit;
      static_assert(kFxVenue == 36, "term_limit is derived from 2^(50 - kFxVenue) (fill_set_b)");
      uint32_t m = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) m = max(m, __float_as_uint(x[q]) & 0x7FFFFFFFu);
      if (__builtin_expect(m > kLimit, 0)) {            // a NaN / infinity / out-of-window term: saturates or poisons its venue
        venue_group_slow(sums, vflags, vwords, kLimit, base_k, lv[0], lv[1], lv[2], lv[3], lv[4], lv[5], lv[6], lv[7], x[0],
                         x[1], x[2], x[3], x[4], x[5], x[6], x[7]);
        return;
      }
      fx_t s8 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const fx_t f = to_fx<kFxVenue>(x[q]);
        const bool first = (q == 0) || (lv[q] != lv[q - (q > 0)]);
        s8 = first ? f : s8 + f;
        const bool last = (q == 7) || (lv[q + (q < 7)] != lv[q]);
        const bool take = last && lv[q] != 0xFFFF;
        atomicAdd(&sums[take ? base_k + lv[q] : dummy], s8);
      }
    };
#endif
#ifndef GJ_WAVE_RUNS
#define GJ_WAVE_RUNS 1
#endif
    // 512 consecutive slots of ONE venue (a wave's 64 groups): the slots' fixed-point terms are added up per lane, then
    // across the wave (integer adds: the same sum, bit for bit), and ONE lane adds the total - instead of 512 LDS
    // atomics per network of which 448 go to scratch sums and 64 to one address.  What makes the case common: venues far
    // larger than a tile's width - the leisure venues of a JUNE world (every resident of the k nearest super areas:
    // 15 000 attendees, six networks per slot), the giant venues of BASELINE config 5, schools on a world with a geography.
    auto wave_sum = [&](const float (&xl)[8], int target) -> bool {      // false: a term outside the window - generic path
      uint32_t m = 0u;
#pragma unroll
      for (int q = 0; q < 8; ++q) m = max(m, __float_as_uint(xl[q]) & 0x7FFFFFFFu);
      if (__builtin_amdgcn_ballot_w64(m > T.term_limit) != 0ull) return false;
      fx_t s = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += to_fx<kFxVenue>(xl[q]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
      if ((tid & (kWave - 1)) == 0) atomicAdd(&sums[target], s);
      return true;
    };
    auto add_group = [&](const uint4 raw, const float4 xa, const float4 xb, const uint2 craw) {
      const Slots8 L{{raw.x, raw.y, raw.z, raw.w}};
      const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
      int lv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) lv[q] = L.lv(q);
      bool one_venue = false;
#if GJ_WAVE_RUNS
      {
        // (raw.x == raw.y == raw.z == raw.w and both halves of a word equal: all eight local venues are the same)
        const bool mine = (raw.x == raw.y) && (raw.y == raw.z) && (raw.z == raw.w) && ((raw.x >> 16) == (raw.x & 0xFFFFu)) &&
                          (lv[0] != 0xFFFF) && (lv[0] == __builtin_amdgcn_readfirstlane(lv[0]));
        one_venue = __builtin_amdgcn_ballot_w64(mine) == ~0ull;         // every lane of the wave, all of them active
      }
#endif
      if (!T.leisure) {
        if (!(one_venue && wave_sum(x, lv[0]))) run_sums(lv, x, 0);
      } else {
        const uint32_t cw[2] = {craw.x, craw.y};
        for (int k = 0; k < nk; ++k) {
          const float* tk = tabs + k * 200;
          float xl[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) xl[q] = tk[(cw[q >> 2] >> ((q & 3) * 8)) & 0xFF] * x[q];
          if (!(one_venue && wave_sum(xl, k * nv + lv[0]))) run_sums(lv, xl, k * nv);
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
        xa[u] = load_nt(val4 + 2 * gu);           // read once, written by phase A: non-temporal
        xb[u] = load_nt(val4 + 2 * gu + 1);
        craw[u] = T.leisure ? cls8[gu] : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < kVenueUnroll; ++u)
        if (g + u * kTileThreads < g1) add_group(raw[u], xa[u], xb[u], craw[u]);
    }
    if (T.pv_blk) {
      // Run form: the agents [r0, r1) have their primary venue in this block; agent a is a slot with value x[a] and
      // local venue pv_blk[a] - the same straight-line groups of 8 as above, read from the per-agent arrays themselves
      // (consecutive agents of one venue are a run).  Agents of the neighbouring blocks in the first / last group are
      // masked; the last group of the world may reach past x.
      const int r0 = T.blk_r0[j], r1 = T.blk_r0[j + 1];
      const uint4* pv8 = reinterpret_cast<const uint4*>(T.pv_blk);
      const float4* x4 = reinterpret_cast<const float4*>(T.x);
      const int ga = r0 >> 3, gb = (r1 + 7) >> 3;
      const int g_in = (int)(T.n_x >> 3);            // groups wholly inside x
      for (int g = ga + tid; g < gb; g += kVenueUnroll * kTileThreads) {
        uint4 raw[kVenueUnroll];
        float4 xa[kVenueUnroll], xb[kVenueUnroll];
#pragma unroll
        for (int u = 0; u < kVenueUnroll; ++u) {
          const int gu = min(g + u * kTileThreads, gb - 1);
          raw[u] = pv8[gu];
          if (gu < g_in) {
            xa[u] = x4[2 * gu];
            xb[u] = x4[2 * gu + 1];
          } else {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = ((int64_t)gu * 8 + q < T.n_x) ? T.x[(int64_t)gu * 8 + q] : 0.0f;
            xa[u] = make_float4(t[0], t[1], t[2], t[3]);
            xb[u] = make_float4(t[4], t[5], t[6], t[7]);
          }
        }
#pragma unroll
        for (int u = 0; u < kVenueUnroll; ++u) {
          const int gu = g + u * kTileThreads;
          if (gu >= gb) continue;
          const Slots8 L{{raw[u].x, raw[u].y, raw[u].z, raw[u].w}};
          const float x[8] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w, xb[u].x, xb[u].y, xb[u].z, xb[u].w};
          int lv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int a = gu * 8 + q;
            lv[q] = (a >= r0 && a < r1) ? L.lv(q) : 0xFFFF;
          }
          run_sums(lv, x, 0);
        }
      }
    }
    __syncthreads();
    // cum = (beta * p_contact) * sum per venue and network.  A lane's p_contact loads are issued kCumBatch at a time:
    // one venue per iteration (load, wait, convert, store) made this write-out a chain of nv / 1024 = 16 memory round
    // trips per workgroup - most of what a household item (25 us) spent its time on.
    for (int l0 = tid; l0 < nv; l0 += kCumBatch * kTileThreads) {
      float pc[kCumBatch];
#pragma unroll
      for (int u = 0; u < kCumBatch; ++u) pc[u] = T.v_pc[v0 + min(l0 + u * kTileThreads, nv - 1)];   // clamped, unconditional
      for (int k = 0; k < nk; ++k) {
        const float beta = T.beta[k];
#pragma unroll
        for (int u = 0; u < kCumBatch; ++u) {
          const int lv = l0 + u * kTileThreads;
          if (lv < nv) {
            const float c = (beta * pc[u]) * fx_read<kFxVenue>(sums, vflags, vwords, k * nv + lv);
            T.cum[(int64_t)(v0 + lv) * T.stride + k] = c;
            cumf[2 * (k * nv + lv)] = c;      // low half of the lane's own 8-byte slot
          }
        }
      }
    }
    if (B.mode == 1 || T.direct) return;
  } else {
    if (T.direct) return;
    for (int k = 0; k < nk; ++k) {
      for (int l0 = tid; l0 < nv; l0 += kCumBatch * kTileThreads) {
        float c[kCumBatch];
#pragma unroll
        for (int u = 0; u < kCumBatch; ++u)
          c[u] = T.cum[(int64_t)(v0 + min(l0 + u * kTileThreads, nv - 1)) * T.stride + k];
#pragma unroll
        for (int u = 0; u < kCumBatch; ++u) {
          const int lv = l0 + u * kTileThreads;
          if (lv < nv) cumf[2 * (k * nv + lv)] = c[u];
        }
      }
    }
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
        for (int q = 0; q < 8; ++q) {      // straight-line: a pad slot reads venue 0 and keeps 0
          const int lv = L.lv(q);
          const float c = cumf[2 * (lv != 0xFFFF ? lv : 0)];
          r[q] = (lv != 0xFFFF) ? c : 0.0f;
        }
      } else {
        const uint32_t cw[2] = {craw[u].x, craw[u].y};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int lv = L.lv(q);
          const int c = (cw[q >> 2] >> ((q & 3) * 8)) & 0xFF;
          const int li = lv != 0xFFFF ? lv : 0;
          float a = 0.0f;
          for (int k = 0; k < nk; ++k) a += tabs[nk * 200 + k * 200 + c] * cumf[2 * (k * nv + li)];
          r[q] = (lv != 0xFFFF) ? a : 0.0f;
        }
      }
      store_nt(val4 + 2 * gu, make_float4(r[0], r[1], r[2], r[3]));      // whole lines, read once by phase D
      store_nt(val4 + 2 * gu + 1, make_float4(r[4], r[5], r[6], r[7]));
    }
  }
}

// ---- pass 1 in the "direct" form (sets with few venues whose edges all belong to owned agents) ----------------------
// A set whose pass 2 is direct has, per owned agent, the venue ids of its edges (ELL rows) - which serve pass 1 as
// well: a workgroup takes a contiguous range of agents, reads their transmissions (coalesced), rows and classes, and
// adds every edge's term into an LDS table of 64-bit fixed-point sums, one per (venue, network) - exact, so the order
// and the split over workgroups cannot change a bit - and writes its table to partial[workgroup][venue][network].
// k_presum_reduce then adds the tables up and applies beta * p_contact.  Per edge: ~3 bytes of rows instead of the
// ~12.3 of phases A + B (a_la + val write + descriptors, e_lv + val read), and the LDS atomics leave the venue launch.
constexpr int GJ_MAX_PRESUM = 6;              // (opt-in experiment: at most six sets take pass 1 in this form)
constexpr int64_t kPresumBad = (int64_t)0x8000000000000000ll;      // a sum that holds a value which cannot be summed
struct TPSet {
  const uint16_t* ell;    // [planes][rows][2]
  int64_t plane_stride;
  fx_t* partial;          // [workgroups][V * stride]
  int32_t planes, V, stride, nk;
  int32_t group_venues;   // venues per LDS table
  int32_t raw, leisure, _pad;
  int32_t table[GJ_MAX_NETS_PER_SET];
  int32_t age75[GJ_MAX_NETS_PER_SET];
};
struct TilePArgs {
  TPSet sets[GJ_MAX_PRESUM];
  int32_t n_sets;
  int32_t agents_per_wg;  // multiple of 4
  int64_t n_agents;
  const float* trans;
  const float* qtrans;
  const uint8_t* cls;
  const float* tables;
  int32_t day_type, transpose;
};

constexpr int kPresumQuads = 3;     // quads of agents a lane keeps in flight
// (the opt-in experiment keeps round 3's rule: one flag per sum, anything that cannot be summed reads back NaN)
__device__ __forceinline__ void fx_add_or_poison(fx_t* sums, uint32_t* flags, int i, float x) {
  const bool ok = fabsf(x) <= fx_max<kFxVenue>();
  atomicAdd(&sums[i], to_fx<kFxVenue>(ok ? x : 0.0f));
  if (__builtin_expect(!ok, 0)) atomicOr(&flags[i >> 5], 1u << (i & 31));
}

__global__ __launch_bounds__(kTileThreads) void k_tile_presum(const TilePArgs P) {
  extern __shared__ __align__(16) fx_t lds_p[];
  const int tid = threadIdx.x;
  const int64_t a_begin = (int64_t)blockIdx.x * P.agents_per_wg;
  const int64_t a_end = min(P.n_agents, a_begin + P.agents_per_wg);
  const int n_quads = (int)((max(a_end - a_begin, (int64_t)0) + 3) >> 2);
  for (int t = 0; t < P.n_sets; ++t) {
    const TPSet& T = P.sets[t];
    const int nk = T.nk;
    const float* x = T.raw ? P.trans : P.qtrans;
    for (int v0 = 0; v0 < T.V; v0 += T.group_venues) {
      const int nv = min(T.group_venues, T.V - v0);
      const int n_sums = nv * nk;
      const int dummy = n_sums + (tid & 63);             // a scratch sum per lane of a wave: what does not count adds 0 there
      float* wtab = reinterpret_cast<float*>(lds_p + n_sums + 64);                  // [nk][200] weights of the transmitting side
      uint32_t* flags = reinterpret_cast<uint32_t*>(wtab + (T.leisure ? nk * 200 : 0));
      __syncthreads();                                   // the previous table has been written out
      for (int i = tid; i < n_sums + 64; i += kTileThreads) lds_p[i] = 0;
      for (int i = tid; i < (n_sums + 64 + 31) / 32; i += kTileThreads) flags[i] = 0u;
      if (T.leisure) {
        for (int i = tid; i < nk * 200; i += kTileThreads) {
          const int k = i / 200, c = i % 200;
          const float l = P.tables[(int64_t)T.table[k] * GJ_TABLE_SIZE + P.day_type * 200 + c];
          const float lw = T.age75[k] ? l * (((c % 100) > 75) ? 1.0f : 0.0f) : l;
          wtab[i] = P.transpose ? lw : l;
        }
      }
      __syncthreads();
      for (int plane = 0; plane < T.planes; ++plane) {
        const uint16_t* ell = T.ell + plane * T.plane_stride + a_begin * 2;
        // software pipeline over batches of kPresumQuads quads per lane: the next batch's rows / transmissions / classes
        // are in flight while the current batch's terms are added (one workgroup per CU - the table fills the LDS - so
        // nothing else hides the round trip); two register sets alternate
        struct Batch {
          uint4 rows[kPresumQuads];
          float4 xs[kPresumQuads];
          uint32_t cl[kPresumQuads];
        };
        auto load = [&](int b0, Batch& B) {
#pragma unroll
          for (int u = 0; u < kPresumQuads; ++u) {       // clamped, unconditional: every load in flight together
            const int q = min(b0 + tid + u * kTileThreads, n_quads - 1);
            const int64_t a0 = a_begin + 4 * (int64_t)q;
            B.rows[u] = *reinterpret_cast<const uint4*>(ell + 8 * (int64_t)q);     // rows are padded to whole slices
            if (a0 + 4 <= P.n_agents) {
              B.xs[u] = *reinterpret_cast<const float4*>(x + a0);
            } else {
              B.xs[u] = make_float4(a0 < P.n_agents ? x[a0] : 0.0f, a0 + 1 < P.n_agents ? x[a0 + 1] : 0.0f,
                                    a0 + 2 < P.n_agents ? x[a0 + 2] : 0.0f, 0.0f);
            }
            B.cl[u] = T.leisure ? *reinterpret_cast<const uint32_t*>(P.cls + a0) : 0u;
          }
        };
        auto add = [&](int b0, const Batch& B) {
#pragma unroll
          for (int u = 0; u < kPresumQuads; ++u) {
            const int q = b0 + tid + u * kTileThreads;
            if (q >= n_quads) continue;
            const uint32_t w[4] = {B.rows[u].x, B.rows[u].y, B.rows[u].z, B.rows[u].w};
            const float xv[4] = {B.xs[u].x, B.xs[u].y, B.xs[u].z, B.xs[u].w};
            const int n_ok = (int)min((int64_t)4, a_end - (a_begin + 4 * (int64_t)q));   // agents of the quad in this range
            // straight-line: every entry adds - what is empty, another group's or past the range adds 0 to the scratch sum
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int c = 0; c < 2; ++c) {
                const int lv = (int)((w[j] >> (16 * c)) & 0xFFFF) - v0;
                const bool in = ((unsigned)lv < (unsigned)nv) && (j < n_ok);
                if (!T.leisure) {
                  fx_add_or_poison(lds_p, flags, in ? lv : dummy, in ? xv[j] : 0.0f);
                } else {
                  const int cj = (B.cl[u] >> (8 * j)) & 0xFF;
                  for (int k = 0; k < nk; ++k)
                    fx_add_or_poison(lds_p, flags, in ? lv * nk + k : dummy, in ? wtab[k * 200 + cj] * xv[j] : 0.0f);
                }
              }
            }
          }
        };
        constexpr int kStep = kPresumQuads * kTileThreads;
        if (n_quads > 0) {
          Batch ba, bb;
          int b0 = 0;
          load(b0, ba);
          while (true) {
            const bool more1 = b0 + kStep < n_quads;
            if (more1) load(b0 + kStep, bb);
            add(b0, ba);
            if (!more1) break;
            b0 += kStep;
            const bool more2 = b0 + kStep < n_quads;
            if (more2) load(b0 + kStep, ba);
            add(b0, bb);
            if (!more2) break;
            b0 += kStep;
          }
        }
      }
      __syncthreads();
      fx_t* out = T.partial + (int64_t)blockIdx.x * T.V * T.stride;
      for (int i = tid; i < n_sums; i += kTileThreads) {
        const int lv = i / nk, k = i % nk;
        const bool bad = (flags[i >> 5] >> (i & 31)) & 1u;
        out[(int64_t)(v0 + lv) * T.stride + k] = bad ? (fx_t)kPresumBad : lds_p[i];
      }
    }
  }
}

struct PReduceSet {
  const fx_t* partial;
  const float* v_pc;
  float* cum;
  int32_t V, stride, nk, first;      // first: index of the set's first (venue, network) entry in the launch
  float beta[GJ_MAX_NETS_PER_SET];
};
struct PReduceArgs {
  PReduceSet sets[GJ_MAX_PRESUM];
  int32_t n_sets, n_wgs;
  int32_t total;                     // entries of all sets
  int32_t _pad;
};

// cum[v][k] = (beta_k * p_contact[v]) * sum over the workgroups' tables - the same expression as phase B's.  A workgroup
// takes 64 consecutive (venue, network) entries; its four waves each add up a quarter of the tables (for one table the
// 64 entries are 512 contiguous bytes), eight loads in flight per lane, and the four partial sums meet in LDS.
__global__ __launch_bounds__(kThreads) void k_presum_reduce(const PReduceArgs R) {
  __shared__ fx_t part[kThreads / kWave][kWave];
  __shared__ uint32_t bad_any[kWave];
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int i = blockIdx.x * kWave + lane;
  const bool in = i < R.total;
  int t = 0;
  while (t + 1 < R.n_sets && i >= R.sets[t + 1].first) ++t;
  const PReduceSet& T = R.sets[t];
  const int e = in ? i - T.first : 0;
  const int v = e / T.nk, k = e % T.nk;
  const int64_t idx = (int64_t)v * T.stride + k, table = (int64_t)T.V * T.stride;
  fx_t s = 0;
  bool bad = false;
  if (threadIdx.x < kWave) bad_any[threadIdx.x] = 0u;
  constexpr int kWaves = kThreads / kWave;
  for (int w0 = wave; w0 < R.n_wgs; w0 += kWaves * 8) {
    fx_t p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int w = w0 + u * kWaves;
      p[u] = (in && w < R.n_wgs) ? T.partial[w * table + idx] : (fx_t)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bad |= p[u] == (fx_t)kPresumBad;
      s += p[u];
    }
  }
  part[wave][lane] = s;
  __syncthreads();
  if (bad) atomicOr(&bad_any[lane], 1u);
  __syncthreads();
  if (wave == 0 && in) {
    fx_t total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) total += part[w][lane];
    T.cum[idx] = (T.beta[k] * T.v_pc[v]) * (bad_any[lane] ? __builtin_nanf("") : from_fx<kFxVenue>(total));
  }
}

// ---- phase D: per slice, accumulate the edges' values per agent in LDS; epilogue a7-a9 -----------
constexpr int GJ_MAX_DIRECT = GJ_MAX_SETS;      // every set may be in the direct or the run form (a partition's split halves:
                                                // up to 12 sets; 6 until round 3)
constexpr int kClassWeightFloats = GJ_MAX_NETS_PER_SET * 200;
struct TDirect {          // a set whose pass 2 is taken straight from the venues' cum
  const uint16_t* ell;    // [planes][owned agents, padded to slices][K] venue ids, 0xFFFF = none
  const float* cum;       // [V * stride]
  int64_t plane_stride;   // elements of one plane
  int32_t K, planes;      // K entries per agent and plane (1, or 2: a pair of columns per plane)
  int32_t V, stride, nk;
  int32_t region;         // which of the two LDS table regions its venue values are staged in
  int32_t group_venues;   // venues per staging group (== V unless the table is larger than region 0)
  int32_t _pad;
  int32_t raw, leisure;
  int32_t table[GJ_MAX_NETS_PER_SET];
  int32_t age75[GJ_MAX_NETS_PER_SET];
  // run form (gj_tiled_set.run_*): `ell` holds ONE window-relative index per agent, the table is the slice's window
  // [win_lo[s], win_lo[s] + win_n[s]) of cum - one group, one plane
  const int32_t* win_lo;
  const int32_t* win_n;
};

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
  float* agent_sums;      // non-NULL: optional output, the per-agent sums before the susceptibility factor (gj_step_io)
  // "direct" form of pass 2 (sets with few venues, tiling.py build_ell): no per-edge workspace
  TDirect direct[GJ_MAX_DIRECT];
  int32_t n_direct;
  int32_t table_floats;   // LDS floats of table region 0 (+ 64 of slack); region 1 follows
  int32_t table1_floats;  // LDS floats of table region 1 (0: none - every table is staged in region 0, one after the other)
  int32_t _pad3;
  int32_t day_type, transpose;
  const uint8_t* cls;
  const float* tables;
  const gj_clock* clock;  // non-NULL: now / step are read from device memory (a captured step replayed)
  int32_t _pad2;
  int32_t io_vec4;        // susceptibility and the optional per-agent outputs are 16-byte aligned
};

// a7-a9 for one agent per lane, from the per-agent sums of phase D (split form)
__global__ __launch_bounds__(256) void k_tile_epilogue(const TileDArgs D) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= D.n_agents) return;
  const float now = D.clock ? D.clock->now : D.now;
  const uint64_t step = D.clock ? D.clock->step : D.step;
  float susc = D.susceptibility[a];
  float ts = susc * D.acc_scratch[a];
  if (D.trans_susc) D.trans_susc[a] = ts;
  const float p = not_infected_prob(ts, D.dt);
  if (D.not_infected_probs) D.not_infected_probs[a] = p;
  if (!D.sample) return;
  float e0, e1;
  if (D.exp_noise) {
    e0 = D.exp_noise[a];
    e1 = D.exp_noise[D.n_agents + a];
  } else {
    e0 = e1 = 1.0f;
  }
  const float nw = D.exp_noise ? gumbel_new_infected(p, e0, e1)
                               : own_new_infected(p, infection_uniform(D.seed, step, D.agent_offset + a));
  if (D.new_infected) D.new_infected[a] = nw;
  if (nw != 0.0f) {
    float inf = D.is_infected[a], t_inf = D.infection_time[a];
    infect(nw, now, susc, inf, t_inf);
    D.susceptibility[a] = susc;
    D.is_infected[a] = inf;
    D.infection_time[a] = t_inf;
  }
}

// ---- phase D, direct form (sets with few venues; tiling.py build_ell) --------------------------------------------
// acc32[i] += sum over agent i's ELL entries of the venue's cum (leisure sets: weighted over the set's networks by the
// agent's class).  The venues' values pass through LDS in groups of table_floats / nk venues.  No atomics: a lane owns
// QUADS of consecutive agents, 4 * (tid + m * 1024) + 0..3, so one quad's rows are a single 8K-byte load, its classes
// one dword and its sums one 16-byte LDS access.  A lane's quads are loaded in batches that fit its registers (K <= 2:
// all of the slice's at once, BEFORE the table is staged, so that the two latencies overlap and a set with several
// venue groups reads its rows once).
constexpr int kQuadsPerLane = (kMaxSliceAgents + 4 * kTileThreads - 1) / (4 * kTileThreads);   // 5
static_assert(kQuadsPerLane * 4 * kTileThreads >= kMaxSliceAgents, "a lane's quads must cover the largest slice");

struct DirectBatch {            // a lane's five quads of one set and plane
  uint32_t w[kQuadsPerLane][4]; // quad's ELL rows: agent j, column c = half-word 2 * j + c
  uint32_t cls[kQuadsPerLane];  // the quad's four classes
};

__device__ __forceinline__ void direct_load(const TileDArgs& D, const TDirect& T, int64_t base, int n_local, int tid,
                                            int plane, DirectBatch& b) {
  const unsigned last_quad = (unsigned)(n_local - 1) >> 2;   // rows are padded to whole slices: any quad of the slice is readable
  // wave-uniform bases + 32-bit lane offsets
  const uint16_t* ell = T.ell + plane * T.plane_stride + base * 2;
  const uint8_t* cls = D.cls + base;
#pragma unroll
  for (int u = 0; u < kQuadsPerLane; ++u) {
    unsigned q = min((unsigned)(tid + u * kTileThreads), last_quad);           // clamped: unconditional, all in flight
    // (opaque to the optimiser: hipcc otherwise hoists the five 64-bit class addresses out of the item loop as
    // invariants and pays for the ten registers with scratch spills - 144 bytes per lane, more than the XCD's L2 holds
    // for its 32 workgroups, i.e. HBM traffic)
    asm volatile("" : "+v"(q));
    if (!T.win_lo) {
      const uint4 r = load_nt(reinterpret_cast<const uint4*>(ell + 8u * q));
      b.w[u][0] = r.x;
      b.w[u][1] = r.y;
      b.w[u][2] = r.z;
      b.w[u][3] = r.w;
    } else {          // one index per agent: the quad's four in 8 bytes; the second column is empty
      const uint2 r = *reinterpret_cast<const uint2*>(T.ell + base + 4u * q);
      b.w[u][0] = (r.x & 0xFFFFu) | 0xFFFF0000u;
      b.w[u][1] = (r.x >> 16) | 0xFFFF0000u;
      b.w[u][2] = (r.y & 0xFFFFu) | 0xFFFF0000u;
      b.w[u][3] = (r.y >> 16) | 0xFFFF0000u;
    }
    // the quad's four classes: one dword (agent_class is 4-byte aligned and padded to a multiple of 4 agents)
    b.cls[u] = T.leisure ? *reinterpret_cast<const uint32_t*>(cls + 4u * q) : 0u;
  }
}

__device__ __forceinline__ void direct_add(const TileDArgs& D, const TDirect& T, float (&acc)[kQuadsPerLane][4],
                                           uint32_t qmask, const float* wtab, const float* tab, int v0, int nv,
                                           int n_local, int tid, const DirectBatch& b) {
  constexpr int K = 2;
  const int nk = T.nk, stride = T.stride;       // tab is the venues' cum in memory order: [venue][stride]
#pragma unroll
  for (int u = 0; u < kQuadsPerLane; ++u) {
    const int q = tid + u * kTileThreads;
    if (4 * q >= n_local) continue;
    // quarantined agents (bit set) take nothing from masked sets
    const uint32_t quarantined = T.raw ? 0u : (qmask >> (4 * u)) & 0xFu;
    // branch-free: every entry reads LDS (entry 0 of the table when it is empty, another group's or quarantined),
    // the quad's eight reads issued back to back; what does not count is zeroed afterwards
    int idx[4][K];
    bool ok[4][K];
    float x[4][K];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < K; ++c) {
        const int h = j * K + c;
        const int lv = (int)((b.w[u][h >> 1] >> (16 * (h & 1))) & 0xFFFF) - v0;
        ok[j][c] = ((unsigned)lv < (unsigned)nv) && !((quarantined >> j) & 1u);   // not 0xFFFF, this group's
        idx[j][c] = ok[j][c] ? lv : 0;
      }
    }
    if (!T.leisure) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < K; ++c) x[j][c] = tab[idx[j][c] * stride];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < K; ++c) x[j][c] = 0.0f;
      // (fully unrolled with a uniform early exit: network k is then a constant offset of the LDS reads; as a rolled
      // loop hipcc keeps a dozen strength-reduced address registers alive across it and spills around it)
#pragma unroll
      for (int k = 0; k < GJ_MAX_NETS_PER_SET; ++k) {   // per entry: sum over the set's networks, in network order
        if (k >= nk) break;
        float w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = wtab[k * 200 + ((b.cls[u] >> (8 * j)) & 0xFF)];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int c = 0; c < K; ++c) x[j][c] += w[j] * tab[idx[j][c] * stride + k];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // agents beyond n_local in the last quad add 0
      float sj = 0.0f;
#pragma unroll
      for (int c = 0; c < K; ++c) sj += ok[j][c] ? x[j][c] : 0.0f;
      acc[u][j] += sj;
    }
  }
}

// Staging of one group of a direct set's venue values: LDS-DMA (global_load_lds: no registers, asynchronous until the
// next vmcnt(0), which hipcc places in front of __syncthreads()), in memory order [venue][stride].  64-float pieces,
// one per wave-instruction; the last piece re-reads the last element into the region's slack.  With the first group
// of a leisure set the receiving side's class weights are computed into that set's weight buffer.
__device__ __forceinline__ void direct_stage(const TileDArgs& D, const TDirect& T, float* wtab, float* tab, int v0,
                                             int nv, int tid) {
  if (T.leisure && v0 == 0) {
    for (int i = tid; i < T.nk * 200; i += kTileThreads) {
      const int k = i / 200, c = i % 200;
      const float l = D.tables[(int64_t)T.table[k] * GJ_TABLE_SIZE + D.day_type * 200 + c];
      const float lw = T.age75[k] ? l * (((c % 100) > 75) ? 1.0f : 0.0f) : l;
      wtab[i] = D.transpose ? l : lw;       // weights of the receiving side (pass 2)
    }
  }
#ifndef GJ_DIAG_NO_DIRECT_TABLE
  const float* src = T.cum + (int64_t)v0 * T.stride;
  const int n = nv * T.stride;
  const int lane = tid % kWave, wave = __builtin_amdgcn_readfirstlane(tid / kWave);
#if GJ_DMA_WIDE
  // gfx950's 16-byte LDS-DMA (global_load_lds_dwordx4): 1 KiB per wave-instruction instead of 256 bytes - a quarter of
  // the instructions for the whole pieces, the 4-byte form for the remainder (a lane must not read past the table).
  // Only when both ends are 16-byte aligned (wave-uniform; a run-form window starts at any venue)
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(tab)) & 15u) == 0;
  const int n_wide = aligned ? (n / (4 * kWave)) * (4 * kWave) : 0;
  for (int p = wave; (p + 1) * 4 * kWave <= n_wide; p += kTileWaves)
    __builtin_amdgcn_global_load_lds(src + p * 4 * kWave + 4 * lane, tab + p * 4 * kWave, 16, 0, 0);
  for (int p = wave; n_wide + p * kWave < n; p += kTileWaves)
    __builtin_amdgcn_global_load_lds(src + min(n_wide + p * kWave + lane, n - 1), tab + n_wide + p * kWave, 4, 0, 0);
#else
  for (int p = wave; p * kWave < n; p += kTileWaves)
    __builtin_amdgcn_global_load_lds(src + min(p * kWave + lane, n - 1), tab + p * kWave, 4, 0, 0);
#endif
#endif
}

// All direct sets of the slice.  Work items are (set, venue group, plane) in order.  While item i is summed, the rows
// of item i + 1 are on their way to registers and - when it opens a new venue group that was given the other LDS
// region - its venue values on their way to LDS (DMA): a memory round trip costs 4-10 us while every CU streams, and
// with one workgroup per CU nothing else would hide it.  `cur` holds the rows of the first item, issued before the
// caller's own LDS phase.  LDS: two class-weight buffers (sets alternate), table region 0, table region 1.
// The barrier that publishes a DMA-staged table to the other waves: global_load_lds completes under vmcnt, and a
// workgroup barrier by itself does not wait for it (hipcc happens to place the wait; this does not rely on that).
__device__ __forceinline__ void publish_staged() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

__device__ __forceinline__ bool direct_next(const TileDArgs& D, int& t, int& v0, int& plane) {
  const TDirect& T = D.direct[t];
  if (plane + 1 < T.planes) {
    ++plane;
    return true;
  }
  plane = 0;
  if (!T.win_lo && v0 + T.group_venues < T.V) {
    v0 += T.group_venues;
    return true;
  }
  v0 = 0;
  ++t;
  return t < D.n_direct;
}

__device__ __forceinline__ void direct_sets(const TileDArgs& D, float (&acc)[kQuadsPerLane][4], uint32_t qmask,
                                            float* lds, int64_t base, int n_local, int tid, DirectBatch& cur,
                                            uint64_t diag_t0 = 0) {
  auto wtab_of = [&](int t) { return lds + (t & 1) * kClassWeightFloats; };
  auto tab_of = [&](const TDirect& T) {
    return lds + 2 * kClassWeightFloats + (T.region ? D.table_floats + kWave : 0);
  };
  // a group of venue values: [lo, lo + nv) of the set's cum; the rows index it relative to `v0` (run form: the slice's
  // window, indexed from 0)
  auto group_nv = [&](const TDirect& T, int v0) { return T.win_lo ? T.win_n[blockIdx.x] : min(T.group_venues, T.V - v0); };
  auto group_lo = [&](const TDirect& T, int v0) { return T.win_lo ? T.win_lo[blockIdx.x] : v0; };
  int t = 0, v0 = 0, plane = 0;
  __syncthreads();                              // the LDS is free: every lane has its sums in registers
  direct_stage(D, D.direct[0], wtab_of(0), tab_of(D.direct[0]), group_lo(D.direct[0], 0), group_nv(D.direct[0], 0), tid);
  bool more = true;
  while (more) {
    const TDirect& T = D.direct[t];
    const int nv = group_nv(T, v0);
    if (plane == 0) publish_staged();           // this group's values have landed (vmcnt(0) + barrier) - and its rows
#ifdef GJ_DIAG_STAMPS
    if (threadIdx.x == 0 && D.trans_susc && t < 4)
      D.trans_susc[(int64_t)blockIdx.x * D.slice_agents + 8 + 2 * t] = (float)(__builtin_amdgcn_s_memtime() - diag_t0);
#endif
    int tn = t, vn = v0, pn = plane;
    more = direct_next(D, tn, vn, pn);
    const bool new_group = more && pn == 0;
    // the next group's values can be staged while this one is read if they go to the other region
    const bool overlap = new_group && D.direct[tn].region != T.region;
    if (overlap)
      direct_stage(D, D.direct[tn], wtab_of(tn), tab_of(D.direct[tn]), group_lo(D.direct[tn], vn), group_nv(D.direct[tn], vn), tid);
    DirectBatch nxt;
    if (more) direct_load(D, D.direct[tn], base, n_local, tid, pn, nxt);     // in flight while this item is summed
#ifndef GJ_DIAG_NO_DIRECT_ADD
    direct_add(D, T, acc, qmask, wtab_of(t), tab_of(T), v0, nv, n_local, tid, cur);
#else
    if (cur.w[0][0] == 0x12345678u && cur.cls[0] == 77u) acc[0][0] = 1.0f;     // keep the loads alive
#endif
#ifdef GJ_DIAG_STAMPS
    if (threadIdx.x == 0 && D.trans_susc && t < 4)
      D.trans_susc[(int64_t)blockIdx.x * D.slice_agents + 9 + 2 * t] = (float)(__builtin_amdgcn_s_memtime() - diag_t0);
#endif
    if (new_group && !overlap) {                // same region: only once every wave is done reading this group
      __syncthreads();
      direct_stage(D, D.direct[tn], wtab_of(tn), tab_of(D.direct[tn]), group_lo(D.direct[tn], vn), group_nv(D.direct[tn], vn), tid);
    }
    if (more) cur = nxt;
    t = tn;
    v0 = vn;
    plane = pn;
  }
}

// four consecutive per-agent values: one 16-byte access when the arrays allow it, else guarded scalars
__device__ __forceinline__ void load_quad(const float* p, int64_t a0, int n_ok, bool vec, float (&v)[4]) {
  if (vec && n_ok >= 4) {
    const float4 x = *reinterpret_cast<const float4*>(p + a0);
    v[0] = x.x;
    v[1] = x.y;
    v[2] = x.z;
    v[3] = x.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (j < n_ok) ? p[a0 + j] : 0.0f;
  }
}
// NT: a dense per-agent OUTPUT of the step (probabilities, new cases, sums) - nobody reads it before the next launches
// have streamed hundreds of MB, so it is stored non-temporally instead of sitting in L2 as 80 MB of dirty lines that the
// transmission kernel and phase A of the next step then meet (C3: -6 ... -12 us per step, three rounds of tools/ab.py)
template <bool NT = false>
__device__ __forceinline__ void store_quad(float* p, int64_t a0, int n_ok, bool vec, const float (&v)[4]) {
  if (vec && n_ok >= 4) {
    if (NT) {
      store_nt(reinterpret_cast<float4*>(p + a0), make_float4(v[0], v[1], v[2], v[3]));
    } else {
      *reinterpret_cast<float4*>(p + a0) = make_float4(v[0], v[1], v[2], v[3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < n_ok) p[a0 + j] = v[j];
  }
}

// GJ_DIAG_STAMPS (timing diagnostics, tools/ab.py): lane 0 of every workgroup writes the cycles since its start at
// marked points into trans_susc[slice base + k] (the epilogue then leaves trans_susc alone)
#ifdef GJ_DIAG_STAMPS
#define GJ_STAMP(k)                                                                                         \
  do {                                                                                                      \
    if (threadIdx.x == 0 && D.trans_susc)                                                                    \
      D.trans_susc[(int64_t)blockIdx.x * D.slice_agents + (k)] = (float)(__builtin_amdgcn_s_memtime() - gj_t0); \
  } while (0)
#else
#define GJ_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(kTileThreads) void k_tile_agents(const TileDArgs D) {
  extern __shared__ __align__(16) fx_t lds_acc[];
#ifdef GJ_DIAG_STAMPS
  const uint64_t gj_t0 = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave), lane = tid % kWave;
  const int s = blockIdx.x;
  const int64_t base = (int64_t)s * D.slice_agents;
  const int n_local = (int)min((int64_t)D.slice_agents, D.n_agents - base);
  uint32_t* lds_flags = reinterpret_cast<uint32_t*>(lds_acc + D.slice_agents);   // two bits per agent (fx_flag): POS words, NEG words
  const int fwords = D.slice_agents / 32;
  for (int i = tid; i < n_local; i += kTileThreads) lds_acc[i] = 0;
  for (int i = tid; i < 2 * fwords; i += kTileThreads) lds_flags[i] = 0u;
  DirectBatch first;                       // rows of the first direct item: their round trip hides behind the tiled sets
  if (D.n_direct > 0) direct_load(D, D.direct[0], base, n_local, tid, 0, first);
  __syncthreads();
  // ts = susc * (q * sum over masked sets + sum over raw sets): masked sets first, scale by q, raw sets last
  for (int pass = 0; pass < 2; ++pass) {
    for (int t = 0; t < D.n_sets; ++t) {
      const TSetA& T = D.sets[t];
      if (!T.active || T.direct || T.raw != pass) continue;
      if (T.wide == 2) {
        gather_explicit<kUnroll>(T, lds_acc, lds_flags, fwords, s, wave, lane);
      } else if (T.wide) {
        gather_set<true>(T, lds_acc, lds_flags, fwords, s, wave, lane);
      } else {
        gather_set<false>(T, lds_acc, lds_flags, fwords, s, wave, lane);
      }
    }
    __syncthreads();
    if (pass == 0 && D.has_q) {
      // (a lane's stage loads are issued ten at a time: one per iteration made this a chain of 20 memory round trips)
      constexpr int kStageBatch = 10;
      for (int i0 = tid; i0 < n_local; i0 += kStageBatch * kTileThreads) {
        float stg[kStageBatch];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) stg[u] = D.stage[base + min(i0 + u * kTileThreads, n_local - 1)];
#pragma unroll
        for (int u = 0; u < kStageBatch; ++u) {
          const int i = i0 + u * kTileThreads;
          if (i < n_local && !(stg[u] < D.q_thr)) {
            lds_acc[i] = 0;
            // a sum that merely saturated is finite in the reference: its quarantine factor of 0 makes it 0 (NaN stays NaN)
            const uint32_t bit = 1u << (i & 31);
            const bool pos = lds_flags[i >> 5] & bit, neg = lds_flags[fwords + (i >> 5)] & bit;
            if (pos != neg) {
              atomicAnd(&lds_flags[i >> 5], ~bit);
              atomicAnd(&lds_flags[fwords + (i >> 5)], ~bit);
            }
          }
        }
      }
      __syncthreads();
    }
  }
  GJ_STAMP(1);
  // ---- per-agent tail.  Each lane owns QUADS of consecutive agents, 4 * (tid + m * 1024) + 0..3 (m < 5): their sums
  // leave LDS for registers, so that the whole LDS is free for the direct sets' venue tables, and the epilogue moves
  // 16 bytes per lane and array.
  constexpr int kQ = kQuadsPerLane;
  float acc[kQ][4];
#pragma unroll
  for (int m = 0; m < kQ; ++m) {
    const int i0 = 4 * (tid + m * kTileThreads);
    // (the quad's four "not summable" bits sit in one word: i0 is a multiple of 4)
    const int iq = min(i0, D.slice_agents - 4);           // unconditional, clamped LDS reads (the whole quad is in LDS)
    const uint32_t pos = (lds_flags[iq >> 5] >> (iq & 31)) & 0xFu, neg = (lds_flags[fwords + (iq >> 5)] >> (iq & 31)) & 0xFu;
    fx_t raw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) raw[j] = lds_acc[iq + j];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = from_fx<kFxAgent>(raw[j]);
      acc[m][j] = (i0 < n_local) ? (((pos | neg) >> j) & 1u ? fx_special((pos >> j) & 1u, (neg >> j) & 1u, v) : v) : 0.0f;
    }
  }
  if (D.n_direct > 0) {
    uint32_t qmask = 0u;                                  // bit 4m + j: the agent is quarantined
    if (D.has_q) {
      float stg[kQ][4];
#pragma unroll
      for (int m = 0; m < kQ; ++m) {
        const int i0 = 4 * (tid + m * kTileThreads);
#pragma unroll
        for (int j = 0; j < 4; ++j) stg[m][j] = (i0 + j < n_local) ? D.stage[base + i0 + j] : -INFINITY;
      }
#pragma unroll
      for (int m = 0; m < kQ; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) qmask |= (stg[m][j] < D.q_thr) ? 0u : (1u << (4 * m + j));
    }
    GJ_STAMP(2);
#ifndef GJ_DIAG_NO_DIRECT_SETS       // timing diagnostics only (tools/ab.py): what the direct sets cost in phase D
#ifdef GJ_DIAG_STAMPS
    direct_sets(D, acc, qmask, reinterpret_cast<float*>(lds_acc), base, n_local, tid, first, gj_t0);
#else
    direct_sets(D, acc, qmask, reinterpret_cast<float*>(lds_acc), base, n_local, tid, first);
#endif
#endif
    GJ_STAMP(3);
  }
  const bool vec = D.io_vec4 != 0;
  if (D.agent_sums) {    // the forward's sums, kept for a backward pass
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
      const int i0 = 4 * (tid + m * kTileThreads);
      if (i0 < n_local) store_quad<true>(D.agent_sums, base + i0, n_local - i0, vec, acc[m]);
    }
  }
  if (D.acc_scratch) {   // split form: hand the per-agent sums to k_tile_epilogue (runs at full occupancy)
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
      const int i0 = 4 * (tid + m * kTileThreads);
      if (i0 < n_local) store_quad(D.acc_scratch, base + i0, n_local - i0, vec, acc[m]);
    }
    return;
  }
  // Epilogue a7-a9.  ALL of a lane's susceptibility loads are issued together (one memory round trip, not one per
  // quad); ts = susceptibility * sum goes to LDS (a lane reads back only what it wrote), so that the final loop is
  // rolled.  In it a quad's infection thresholds are drawn (Philox: the epilogue's arithmetic, independent of the
  // data) between its stores - measured: as a phase of its own in front of the loop the same arithmetic costs 40 %
  // more, there is nothing in flight for it to hide behind.
  const float now = D.clock ? D.clock->now : D.now;
  const uint64_t step = D.clock ? D.clock->step : D.step;
  const bool own_noise = D.sample && !D.exp_noise;
  const int mis = (int)((D.agent_offset + base) & 3);       // (wave-uniform: base is a multiple of 64)
  float4* ts4 = reinterpret_cast<float4*>(lds_acc);
  {
    float sq[kQ][4];
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
      int i0 = 4 * (tid + m * kTileThreads);
      asm volatile("" : "+v"(i0));          // (computed here, not carried through the direct sets as five 64-bit addresses)
#pragma unroll
      for (int j = 0; j < 4; ++j) sq[m][j] = 0.0f;
      if (i0 < n_local) load_quad(D.susceptibility, base + i0, n_local - i0, vec, sq[m]);
    }
    __syncthreads();          // every lane has its sums in registers, the last venue table has been read by every wave
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
      const int q = tid + m * kTileThreads;
      if (4 * q < n_local)
        ts4[q] = make_float4(sq[m][0] * acc[m][0], sq[m][1] * acc[m][1], sq[m][2] * acc[m][2], sq[m][3] * acc[m][3]);
    }
  }
  GJ_STAMP(4);
  uint32_t infected = 0u;                       // bit 4m + j: agent 4 * (tid + m * 1024) + j was infected in this step
  int m_run = 0;
  for (int q = tid; 4 * q < n_local; q += kTileThreads, ++m_run) {
    const int i0 = 4 * q;
    const float4 tq = ts4[q];
    const float ts[4] = {tq.x, tq.y, tq.z, tq.w};
    const int64_t a0 = base + i0;
    const int n_ok = n_local - i0;
    float p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = not_infected_prob(ts[j], D.dt);
#ifndef GJ_DIAG_STAMPS
    if (D.trans_susc) store_quad<true>(D.trans_susc, a0, n_ok, vec, ts);
#endif
    if (D.not_infected_probs) store_quad<true>(D.not_infected_probs, a0, n_ok, vec, p);
    if (!D.sample) continue;
    float nw[4], th[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (own_noise) {
      // one Philox block serves the GLOBAL agents 4k .. 4k+3: a quad that starts on a multiple of four takes one
      // block, any other quad two (wave-uniform: base is a multiple of 64)
      const uint64_t g0 = (uint64_t)(D.agent_offset + a0);
      uint32_t r0[4], r1[4] = {0u, 0u, 0u, 0u};
      philox4x32_10(g0 >> 2, step, D.seed, r0);
      if (mis) philox4x32_10((g0 >> 2) + 1, step, D.seed, r1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = mis + j;                       // word of the pair of blocks
        th[j] = u01(w == 0 ? r0[0] : w == 1 ? r0[1] : w == 2 ? r0[2] : w == 3 ? r0[3] : w == 4 ? r1[0] : w == 5 ? r1[1] : r1[2]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (own_noise) {
        nw[j] = own_new_infected(p[j], th[j]);
      } else {
        float e0 = 1.0f, e1 = 1.0f;
        if (j < n_ok) {
          e0 = D.exp_noise[a0 + j];
          e1 = D.exp_noise[D.n_agents + a0 + j];
        }
        nw[j] = gumbel_new_infected(p[j], e0, e1);
      }
      if (j < n_ok && nw[j] != 0.0f) infected |= 1u << (4 * m_run + j);
    }
    if (D.new_infected) store_quad<true>(D.new_infected, a0, n_ok, vec, nw);
    // a9 where nw is neither 0 nor 1 (a NaN probability under injected noise): on the spot, with the value itself
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < n_ok && nw[j] != 0.0f && nw[j] != 1.0f) {
        infected &= ~(1u << (4 * m_run + j));
        const int64_t a = a0 + j;
        float sc = D.susceptibility[a], inf = D.is_infected[a], t_inf = D.infection_time[a];
        infect(nw[j], now, sc, inf, t_inf);
        D.susceptibility[a] = sc;
        D.is_infected[a] = inf;
        D.infection_time[a] = t_inf;
      }
    }
  }
  // a9 for the agents infected in this step (new_infected == 1), all of a lane's at once: one memory round trip for
  // the whole slice instead of one per quad that holds a new case (a wave covers 256 agents per iteration: at one
  // new case per thousand agents a quarter of the iterations would wait on these loads)
  GJ_STAMP(6);
#ifndef GJ_TAIL_BATCHED
  // Each lane takes its new cases one after the other: one memory round trip per case of the wave's busiest lane (a wave
  // of 1 280 agents has a handful of new cases per step, two in one lane are rare), three loads issued together, ~30
  // instructions.  (The batched form below - every (quad, agent) slot of the lane loaded straight-line, idle slots
  // reading the slice's first agent - pays one round trip too, but 60 loads, 60 predicated stores and their address
  // selects per lane whether there is a case or not: 8 % of the launch.)
  if (__builtin_amdgcn_ballot_w64(infected != 0u) != 0ull) {
    uint32_t todo = infected;
    while (todo) {
      const int k = __builtin_ctz(todo);
      todo &= todo - 1u;
      const int64_t a = base + 4 * (tid + (k >> 2) * kTileThreads) + (k & 3);
      float sc = D.susceptibility[a], inf = D.is_infected[a], tinf = D.infection_time[a];
      infect(1.0f, now, sc, inf, tinf);
      D.susceptibility[a] = sc;
      D.is_infected[a] = inf;
      D.infection_time[a] = tinf;
    }
  }
#else
  if (__builtin_amdgcn_ballot_w64(infected != 0u) != 0ull) {
    // straight-line loads: a lane without a case at (m, j) reads the slice's first agent instead (one line for the
    // whole wave) - a load under its own branch would be waited for on its own
    float sc[kQ][4], inf[kQ][4], tinf[kQ][4];
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool hit = (infected >> (4 * m + j)) & 1u;
        const int64_t a = hit ? base + 4 * (tid + m * kTileThreads) + j : base;
        sc[m][j] = D.susceptibility[a];
        inf[m][j] = D.is_infected[a];
        tinf[m][j] = D.infection_time[a];
      }
    }
#pragma unroll
    for (int m = 0; m < kQ; ++m) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t a = base + 4 * (tid + m * kTileThreads) + j;
        infect(1.0f, now, sc[m][j], inf[m][j], tinf[m][j]);
        if ((infected >> (4 * m + j)) & 1u) {
          D.susceptibility[a] = sc[m][j];
          D.is_infected[a] = inf[m][j];
          D.infection_time[a] = tinf[m][j];
        }
      }
    }
  }
#endif
  GJ_STAMP(5);
}

}  // namespace gj
