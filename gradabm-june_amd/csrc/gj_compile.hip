// Graph compile on the device (include/gradjune_hip.h, "graph compile"): the tiled layout
// of one edge set built in HBM from the reference's unsorted COO edge_index.  The arrays are specified by
// grad_june_amd/tiling.py (build_tiled / wide_descriptors / build_ell, numpy) and come out identical bit for bit
// (tests/test_gpu_compile_native.py).  All O(E) work is streaming kernels around one radix sort of 64-bit keys
// (tile, local venue, local agent) and a handful of prefix sums (rocPRIM through hipcub); nothing is allocated here.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>

#include "../../include/gradjune_hip.h"

namespace gjc {

constexpr int kThreads = 256;
constexpr int kPad = 8;       // tiling.PAD
constexpr int kChunk = 64;    // tiling.CHUNK
constexpr int kWideSegments = 6;

static inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kThreads - 1) / kThreads, 1 << 18)); }
// launches that end in one atomic per wave on ONE counter: few, long-running workgroups (a grid of n / 256 workgroups
// serialises ~n / 64 atomics on that address: 2.8 ms for 15 M edges, rocprof round 3)
static inline int grid_for_reduction(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + kThreads - 1) / kThreads, 2048)); }
static inline int64_t align_up(int64_t x) { return (x + 255) & ~(int64_t)255; }
static inline int bits_for(uint64_t n) {      // bits needed to represent values < n
  int b = 0;
  while (b < 64 && (n > ((uint64_t)1 << b))) ++b;
  return std::max(b, 1);
}

// first index i in [0, n) with a[i] > x, or n (numpy searchsorted side="right")
__device__ __forceinline__ int upper_bound(const int32_t* a, int n, int64_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)a[mid] > x) {
      hi = mid;
    } else {
      lo = mid + 1;
    }
  }
  return lo;
}

// ---- stage 1: degrees, range checks, venue blocks -------------------------------------------------------------
__global__ void k_degrees(const int64_t* agent, const int64_t* venue, int64_t E, int64_t n_ext, int32_t V,
                          int32_t* degree, int32_t* counts) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e], v = venue[e];
    if (a < 0 || a >= n_ext) {
      counts[GJ_CC_ERROR] = 1;
      continue;
    }
    if (v < 0 || v >= V) {
      counts[GJ_CC_ERROR] = 2;
      continue;
    }
    atomicAdd(&degree[v], 1);
  }
}

// tiling.venue_blocks: a greedy walk over the venues' edge prefix sums, one wave, 64-ary searches
__global__ void k_venue_blocks(const int32_t* rp, int32_t V, int32_t sv_max, int32_t eb_target, int32_t* blk_v0,
                               int32_t cap, int32_t* counts) {
  const int lane = threadIdx.x;
  int v = 0, J = 0;
  if (lane == 0) blk_v0[0] = 0;
  while (v < V) {
    const int64_t target = (int64_t)rp[v] + eb_target;
    int lo = v, hi = V + 1;                         // upper bound of target in rp[0 .. V] lies in [lo, hi]
    while (lo < hi) {
      const int step = (hi - lo + 63) / 64;
      const int idx = lo + lane * step;
      const bool gt = idx < hi ? (int64_t)rp[idx] > target : true;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(gt);
      if (b == 0ull) {
        lo = lo + 63 * step + 1;
      } else {
        const int f = __builtin_ctzll(b);
        hi = min(hi, lo + f * step);
        lo = f == 0 ? lo : lo + (f - 1) * step + 1;
      }
    }
    int end = lo - 1;
    end = max(end, v + 1);
    end = min(end, min(V, v + sv_max));
    if (J >= cap) {
      if (lane == 0) counts[GJ_CC_ERROR] = 3;
      break;
    }
    ++J;
    if (lane == 0) blk_v0[J] = end;
    v = end;
  }
  if (lane == 0) counts[GJ_CC_BLOCKS] = J;
}

// ---- stage 2 --------------------------------------------------------------------------------------------------
// key = (tile = j * S + s, local venue, local agent): block-major tile order, inside a tile by venue then agent
__global__ void k_keys(const int64_t* agent, const int64_t* venue, int64_t E, const int32_t* blk_v0, int32_t J,
                       int32_t S, int32_t SA, uint64_t* keys, uint32_t* vals, int32_t* tile_len_js) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e], v = venue[e];
    const int j = upper_bound(blk_v0, J + 1, v) - 1;
    const uint32_t lv = (uint32_t)(v - blk_v0[j]);
    const int s = (int)(a / SA);
    const uint32_t la = (uint32_t)(a - (int64_t)s * SA);
    const uint32_t tile = (uint32_t)j * (uint32_t)S + (uint32_t)s;
    keys[e] = ((uint64_t)tile << 32) | ((uint64_t)lv << 16) | la;
    vals[e] = (uint32_t)e;
    atomicAdd(&tile_len_js[tile], 1);
  }
}

__global__ void k_block_slots(const int32_t* upos_js, int32_t J, int32_t S, int32_t* blk_slots) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > J) return;
  if (j == J) {
    blk_slots[j] = 0;
    return;
  }
  const int len = upos_js[(int64_t)(j + 1) * S] - upos_js[(int64_t)j * S];
  blk_slots[j] = (len + kPad - 1) / kPad * kPad;
}

// slice-major tile tables: len_sj[s * J + j], tile_jpos[s * J + j]
__global__ void k_tile_tables(const int32_t* tile_len_js, const int32_t* upos_js, const int32_t* blk_e0, int32_t J,
                              int32_t S, int32_t* len_sj, int32_t* tile_jpos) {
  const int64_t n = (int64_t)S * J;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x) {
    if (t == n) {
      len_sj[t] = 0;
      continue;
    }
    const int s = (int)(t / J), j = (int)(t - (int64_t)s * J);
    const int64_t js = (int64_t)j * S + s;
    len_sj[t] = tile_len_js[js];
    tile_jpos[t] = upos_js[js] + (blk_e0[j] - upos_js[(int64_t)j * S]);
  }
}

__global__ void k_fill(const uint64_t* keys, const uint32_t* order, int64_t E, const int64_t* agent,
                       const uint8_t* agent_class, const int32_t* upos_js, const int32_t* sptr,
                       const int32_t* tile_jpos, int32_t J, int32_t S, uint16_t* a_la, uint16_t* e_lv, uint8_t* e_cls) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
    const uint32_t tile = (uint32_t)(key >> 32);
    const int j = (int)(tile / (uint32_t)S), s = (int)(tile - (uint32_t)j * (uint32_t)S);
    const int within = (int)i - upos_js[tile];
    const int64_t t = (int64_t)s * J + j;
    a_la[sptr[t] + within] = (uint16_t)(key & 0xFFFF);
    const int pos_bm = tile_jpos[t] + within;
    e_lv[pos_bm] = (uint16_t)((key >> 16) & 0xFFFF);
    if (e_cls) e_cls[pos_bm] = agent_class[agent[order[i]]];
  }
}

__global__ void k_chunk_counts(const int32_t* sptr, int32_t J, int32_t S, int32_t* n_chunks) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s > S) return;
  n_chunks[s] = s == S ? 0 : (sptr[(int64_t)(s + 1) * J] - sptr[(int64_t)s * J] + kChunk - 1) / kChunk;
}

struct ChunkGeom {
  int s, first_edge, chunk_end, t0;
};
__device__ __forceinline__ ChunkGeom chunk_geom(int c, const int32_t* chunk_ptr, const int32_t* sptr, int32_t J, int32_t S) {
  ChunkGeom g;
  g.s = upper_bound(chunk_ptr, S + 1, c) - 1;
  const int seg0 = sptr[(int64_t)g.s * J], seg1 = sptr[(int64_t)(g.s + 1) * J];
  g.first_edge = seg0 + kChunk * (c - chunk_ptr[g.s]);
  g.chunk_end = min(g.first_edge + kChunk, seg1);
  g.t0 = upper_bound(sptr, S * J + 1, g.first_edge) - 1;
  return g;
}

// tiling.build_tiled's 4-word descriptors: slot0, slot1, split | multi << 16, j0
__global__ void k_descriptors(const int32_t* chunk_ptr, const int32_t* sptr, const int32_t* tile_jpos, int32_t J,
                              int32_t S, int32_t E, int4* desc, int32_t* counts) {
  const int total = chunk_ptr[S];
  if (blockIdx.x == 0 && threadIdx.x == 0) counts[GJ_CC_CHUNKS] = total;
  int n_multi = 0;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const ChunkGeom g = chunk_geom(c, chunk_ptr, sptr, J, S);
    const int end0 = sptr[g.t0 + 1];
    const int split = min(end0, g.chunk_end) - g.first_edge;
    const int slot0 = tile_jpos[g.t0] + (g.first_edge - sptr[g.t0]);
    const bool two = end0 < g.chunk_end;
    const int t1 = two ? upper_bound(sptr, S * J + 1, min(end0, E - 1)) - 1 : g.t0;
    const int slot1 = two ? tile_jpos[t1] + (end0 - sptr[t1]) : 0;
    const bool multi = two && sptr[t1 + 1] < g.chunk_end;
    n_multi += multi ? 1 : 0;
    desc[c] = make_int4(slot0, slot1, split + (multi ? (1 << 16) : 0), g.t0 - g.s * J);
  }
  if (n_multi) atomicAdd(&counts[GJ_CC_MULTI], n_multi);
}

// tiling.wide_descriptors: base_0..5, start_1..4 (bytes), start_5 | multi << 8 | j0 << 9
__global__ void k_wide_descriptors(const int32_t* chunk_ptr, const int32_t* sptr, const int32_t* tile_jpos, int32_t J,
                                   int32_t S, int32_t total, int32_t* desc, int32_t* counts) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const ChunkGeom g = chunk_geom(c, chunk_ptr, sptr, J, S);
    uint32_t base[kWideSegments] = {0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t start[kWideSegments] = {0u, 64u, 64u, 64u, 64u, 64u};
    base[0] = (uint32_t)(tile_jpos[g.t0] + (g.first_edge - sptr[g.t0]));
    int p = sptr[g.t0 + 1];                                   // where the next non-empty tile begins
    int k = 1;
    while (p < g.chunk_end && k < kWideSegments) {
      const int t = upper_bound(sptr, S * J + 1, p) - 1;      // the non-empty tile that starts at p
      start[k] = (uint32_t)(p - g.first_edge);
      base[k] = (uint32_t)(tile_jpos[t] - (int)start[k]);
      p = sptr[t + 1];
      ++k;
    }
    const uint32_t multi = (p < g.chunk_end) ? 1u : 0u;
    if (multi) atomicAdd(&counts[GJ_CC_WIDE_MULTI], 1);      // chunks whose lanes walk the tile tables (> 6 tiles)
    const uint32_t j0 = (uint32_t)(g.t0 - g.s * J);
    if (j0 >= (1u << 22)) counts[GJ_CC_ERROR] = 4;
    int32_t* d = desc + (int64_t)c * 8;
    for (int q = 0; q < kWideSegments; ++q) d[q] = (int32_t)base[q];
    d[6] = (int32_t)(start[1] | (start[2] << 8) | (start[3] << 16) | (start[4] << 24));
    d[7] = (int32_t)(start[5] | (multi << 8) | (j0 << 9));
  }
}

// ---- ELL rows of the direct form --------------------------------------------------------------------------------
__global__ void k_ell_degrees(const int64_t* agent, int64_t E, int64_t n_agents, int32_t* degree, int32_t* counts) {
  int owned = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e];
    if (a >= 0 && a < n_agents) {
      atomicAdd(&degree[a], 1);
      ++owned;
    }
  }
  for (int off = 32; off > 0; off >>= 1) owned += __shfl_xor(owned, off);
  if ((threadIdx.x & 63) == 0 && owned) atomicAdd(&counts[GJ_CC_OWNED_EDGES], owned);
}

__global__ void k_max_degree(const int32_t* degree, int64_t n, int32_t* counts) {
  int m = 0;
  for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < n; a += (int64_t)gridDim.x * blockDim.x)
    m = max(m, degree[a]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(&counts[GJ_CC_MAX_DEGREE], m);
}

__global__ void k_ell_keys(const int64_t* agent, int64_t E, int64_t n_agents, uint32_t* keys, uint32_t* vals) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e];
    keys[e] = (a >= 0 && a < n_agents) ? (uint32_t)a : (uint32_t)n_agents;     // halo edges sort behind the owned ones
    vals[e] = (uint32_t)e;
  }
}

__global__ void k_ell_fill(const uint32_t* keys, const uint32_t* order, int64_t E, int64_t n_agents,
                           const int32_t* rowptr, const int64_t* venue, int64_t rows, int32_t ell_k, uint16_t* ell,
                           int32_t* counts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t a = keys[i];
    if ((int64_t)a >= n_agents) continue;
    const int col = (int)i - rowptr[a];                     // the agent's col-th edge in COO order (stable sort)
    if (col < 0 || col >= ell_k) {                          // ell_k too small for this agent (or a stale `degree`):
      if (counts) counts[GJ_CC_ERROR] = 5;                  // never a write outside the caller's table
      continue;
    }
    ell[((int64_t)(col >> 1) * rows + a) * 2 + (col & 1)] = (uint16_t)venue[order[i]];
  }
}

// explicit slots (tiling.build_tiled, slot_idx): the block-major slot of every slice-major edge position
__global__ void k_explicit_slots(const int32_t* sptr, const int32_t* tile_jpos, int64_t n_tiles, int64_t E, int32_t* slots) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_tiles + 1;                 // last tile t with sptr[t] <= i (empty tiles share a start: the last wins)
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((int64_t)sptr[mid] > i) {
        hi = mid;
      } else {
        lo = mid + 1;
      }
    }
    const int64_t t = lo - 1;
    slots[i] = tile_jpos[t] + (int32_t)(i - sptr[t]);
  }
}

// multi chunks (tiling.attach_multi_slots): a row of 64 explicit slots for every chunk its descriptor cannot express
__global__ void k_multi_flags(const int32_t* desc, int32_t n_chunks, int32_t wide, int32_t* flags) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_chunks) return;
  if (c == n_chunks) {
    flags[c] = 0;
    return;
  }
  const uint32_t w = (uint32_t)desc[(int64_t)c * (wide ? 8 : 4) + (wide ? 7 : 2)];
  flags[c] = wide ? (int32_t)((w >> 8) & 1u) : (int32_t)((w >> 16) != 0u);
}
// one wave per chunk: lane l resolves slice-major position first_edge + l; the descriptor's j0 field becomes the row
__global__ void k_multi_fill(const int32_t* chunk_ptr, const int32_t* sptr, const int32_t* tile_jpos, int32_t J, int32_t S,
                             int32_t n_chunks, int32_t wide, const int32_t* row_of, int32_t* desc, int32_t* multi_slots,
                             int32_t* counts) {
  const int lane = threadIdx.x % 64;
  const int waves = (gridDim.x * blockDim.x) / 64;
  for (int c = (blockIdx.x * blockDim.x + threadIdx.x) / 64; c < n_chunks; c += waves) {
    if (row_of[c + 1] == row_of[c]) continue;          // (wave-uniform)
    const int m = row_of[c];
    const ChunkGeom g = chunk_geom(c, chunk_ptr, sptr, J, S);
    const int pos = g.first_edge + lane;
    int slot = 0;
    if (pos < g.chunk_end) {
      const int t = upper_bound(sptr, S * J + 1, pos) - 1;
      slot = tile_jpos[t] + (pos - sptr[t]);
    }
    multi_slots[(int64_t)m * 64 + lane] = slot;
    if (lane == 0) {
      if (wide) {
        if (m >= (1 << 22)) counts[GJ_CC_ERROR] = 4;
        int32_t* d7 = desc + (int64_t)c * 8 + 7;
        *d7 = (int32_t)(((uint32_t)*d7 & 0x1FFu) | ((uint32_t)m << 9));
      } else {
        desc[(int64_t)c * 4 + 3] = m;
      }
    }
  }
}

// ---- run form (tiling.split_primary_runs / finish_run_form) ---------------------------------------------------------
constexpr int32_t kNoVenue = 0x7FFFFFFF;

__global__ void k_run_vmin(const int64_t* agent, const int64_t* venue, int64_t E, int64_t n_agents, int32_t* vmin) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e];
    if (a >= 0 && a < n_agents) atomicMin(&vmin[a], (int32_t)venue[e]);
  }
}
// the primary edge of an agent: the FIRST edge in COO order to its smallest venue
__global__ void k_run_pick(const int64_t* agent, const int64_t* venue, int64_t E, int64_t n_agents, const int32_t* vmin,
                           int32_t* pick) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e];
    if (a >= 0 && a < n_agents && (int32_t)venue[e] == vmin[a]) atomicMin(&pick[a], (int32_t)e);
  }
}
__global__ void k_run_keep(const int64_t* agent, int64_t E, int64_t n_agents, const int32_t* pick, uint8_t* keep,
                           int32_t* counts) {
  __shared__ int32_t part[2];
  if (threadIdx.x < 2) part[threadIdx.x] = 0;
  __syncthreads();
  int32_t primary = 0, owned = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = agent[e];
    const bool own = a >= 0 && a < n_agents;
    const bool prim = own && pick[a] == (int32_t)e;
    keep[e] = prim ? 0 : 1;
    primary += prim ? 1 : 0;
    owned += own ? 1 : 0;
  }
  if (primary) atomicAdd(&part[0], primary);
  if (owned) atomicAdd(&part[1], owned);
  __syncthreads();
  if (threadIdx.x == 0 && part[0]) atomicAdd(&counts[GJ_CC_RUN_PRIMARY], part[0]);
  if (threadIdx.x == 0 && part[1]) atomicAdd(&counts[GJ_CC_OWNED_EDGES], part[1]);
}
// vmin must be non-decreasing over the owned agents (agents without an edge - kNoVenue - last)
__global__ void k_run_sorted(const int32_t* vmin, int64_t n_agents, int32_t* counts) {
  for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a + 1 < n_agents; a += (int64_t)gridDim.x * blockDim.x)
    if (vmin[a] > vmin[a + 1]) counts[GJ_CC_RUN_UNSORTED] = 1;
}
// the window of venues of every owned slice: [vmin of its first agent, vmin of its last agent with an edge]
__global__ void k_run_windows(const int32_t* vmin, int64_t n_agents, int32_t slice_agents, int32_t n_own_slices,
                              int32_t* win_lo, int32_t* win_n, int32_t* counts) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_own_slices) return;
  const int64_t a0 = (int64_t)s * slice_agents, a1 = min(n_agents, a0 + slice_agents);
  int32_t lo = 0, n = 0;
  if (a0 < a1 && vmin[a0] != kNoVenue) {
    int64_t last = a1 - 1;
    while (last > a0 && vmin[last] == kNoVenue) --last;      // only the world's last agents have no edge (sorted)
    lo = vmin[a0];
    n = vmin[last] - lo + 1;
  }
  win_lo[s] = lo;
  win_n[s] = n;
  if (n > 0) atomicMax(&counts[GJ_CC_RUN_WINDOW], n);
}
__global__ void k_run_index(const int32_t* vmin, int64_t n_agents, int64_t rows, int32_t slice_agents, const int32_t* blk_v0,
                            int32_t J, const int32_t* win_lo, uint16_t* pv_blk, uint16_t* pv_win) {
  for (int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; a < rows; a += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = a < n_agents ? vmin[a] : kNoVenue;
    uint16_t b = 0xFFFF, w = 0xFFFF;
    if (v != kNoVenue) {
      const int j = upper_bound(blk_v0, J + 1, v) - 1;
      b = (uint16_t)(v - blk_v0[j]);
      w = (uint16_t)(v - win_lo[a / slice_agents]);
    }
    pv_blk[a] = b;
    pv_win[a] = w;
  }
}
// blk_r0[j] = first owned agent whose smallest venue is >= blk_v0[j] (numpy searchsorted side="left" on the sorted vmin)
__global__ void k_run_blk_r0(const int32_t* vmin, int64_t n_agents, const int32_t* blk_v0, int32_t J, int32_t* blk_r0) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > J) return;
  const int32_t x = blk_v0[j];
  int64_t lo = 0, hi = n_agents;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (vmin[mid] < x) {
      lo = mid + 1;
    } else {
      hi = mid;
    }
  }
  blk_r0[j] = (int32_t)lo;
}

// ---- workspace ----------------------------------------------------------------------------------------------
struct Geometry {
  int64_t E, V, S, J, n_agents;
  int64_t tiles() const { return S * J; }
};

static int64_t blocks_upper_bound(const gj_compile_set* c) {
  // every block but the last closes on sv_max venues, on eb_target edges or on a single venue above eb_target
  return (c->n_venues + (int64_t)c->sv_max - 1) / c->sv_max + 2 * (c->n_edges / std::max(1, c->eb_target)) + 2;
}

static Geometry geometry_of(const gj_compile_set* c) {
  Geometry g;
  g.E = c->n_edges;
  g.V = c->n_venues;
  g.S = c->n_slices;
  g.J = c->n_blocks > 0 ? c->n_blocks : blocks_upper_bound(c);
  g.n_agents = c->n_agents;
  return g;
}

static size_t sort64_temp(int64_t E, int end_bit) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)E, 0, end_bit, (hipStream_t)0);
  return b;
}
static size_t sort32_temp(int64_t E, int end_bit) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)E, 0, end_bit, (hipStream_t)0);
  return b;
}
static size_t scan_temp(int64_t n) {
  size_t b = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, (hipStream_t)0);
  return b;
}

static size_t select_temp(int64_t E) {
  size_t b = 0;
  (void)hipcub::DeviceSelect::Flagged(nullptr, b, (const int64_t*)nullptr, (const uint8_t*)nullptr, (int64_t*)nullptr,
                                      (int32_t*)nullptr, (int)E, (hipStream_t)0);
  return b;
}

struct Carver {        // hands out 256-byte aligned pieces of the workspace; only counts when base is NULL
  char* base;
  int64_t used = 0;
  template <typename T>
  T* take(int64_t n) {
    T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
    used += align_up(n * (int64_t)sizeof(T));
    return p;
  }
};

struct TilesWs {
  uint64_t *keys_a, *keys_b;
  uint32_t *vals_a, *vals_b;
  int32_t *tile_len_js, *upos_js, *len_sj, *blk_slots, *n_chunks;
  void* temp;
  size_t temp_bytes;
};
static int64_t carve_tiles(const Geometry& g, void* ws, TilesWs* out) {
  Carver c{(char*)ws};
  TilesWs w;
  const int64_t E1 = std::max<int64_t>(g.E, 1);
  w.keys_a = c.take<uint64_t>(E1);
  w.keys_b = c.take<uint64_t>(E1);
  w.vals_a = c.take<uint32_t>(E1);
  w.vals_b = c.take<uint32_t>(E1);
  w.tile_len_js = c.take<int32_t>(g.tiles() + 1);
  w.upos_js = c.take<int32_t>(g.tiles() + 1);
  w.len_sj = c.take<int32_t>(g.tiles() + 1);
  w.blk_slots = c.take<int32_t>(g.J + 1);
  w.n_chunks = c.take<int32_t>(g.S + 1);
  w.temp_bytes = std::max(sort64_temp(E1, 64), scan_temp(g.tiles() + 1));
  w.temp = c.take<char>((int64_t)w.temp_bytes);
  if (out) *out = w;
  return c.used;
}

struct BlocksWs {
  int32_t *degree, *rp;
  void* temp;
  size_t temp_bytes;
};
static int64_t carve_blocks(const Geometry& g, void* ws, BlocksWs* out) {
  Carver c{(char*)ws};
  BlocksWs w;
  w.degree = c.take<int32_t>(g.V + 1);
  w.rp = c.take<int32_t>(g.V + 1);
  w.temp_bytes = scan_temp(g.V + 1);
  w.temp = c.take<char>((int64_t)w.temp_bytes);
  if (out) *out = w;
  return c.used;
}

struct EllWs {
  uint32_t *keys_a, *keys_b, *vals_a, *vals_b;
  int32_t* rowptr;
  void* temp;
  size_t temp_bytes;
};
static int64_t carve_ell(const Geometry& g, void* ws, EllWs* out) {
  Carver c{(char*)ws};
  EllWs w;
  const int64_t E1 = std::max<int64_t>(g.E, 1);
  w.keys_a = c.take<uint32_t>(E1);
  w.keys_b = c.take<uint32_t>(E1);
  w.vals_a = c.take<uint32_t>(E1);
  w.vals_b = c.take<uint32_t>(E1);
  w.rowptr = c.take<int32_t>(g.n_agents + 1);
  w.temp_bytes = std::max(sort32_temp(E1, 32), scan_temp(g.n_agents + 1));
  w.temp = c.take<char>((int64_t)w.temp_bytes);
  if (out) *out = w;
  return c.used;
}

static int check_set(const gj_compile_set* c) {
  if (!c) return GJ_E_NULL;
  if (c->n_edges < 0 || c->n_edges >= ((int64_t)1 << 30) || c->n_venues < 0 || c->n_agents < 0) return GJ_E_RANGE;
  if (c->n_slices < 1 || c->slice_agents < 1 || c->slice_agents > 65536) return GJ_E_RANGE;
  // local venue indices are 16-bit and 0xFFFF marks a pad slot: at most 65535 venues per block (local ids 0..65534)
  if (c->sv_max < 1 || c->sv_max > 65535 || c->eb_target < 1) return GJ_E_RANGE;
  if (c->n_edges > 0 && (!c->agent || !c->venue)) return GJ_E_NULL;
  if (c->n_agents > c->n_ext_agents || c->n_ext_agents > (int64_t)c->n_slices * c->slice_agents) return GJ_E_RANGE;
  return 0;
}

#define GJC_HIP(expr)                      \
  do {                                     \
    const hipError_t e_ = (expr);          \
    if (e_ != hipSuccess) return (int)e_;  \
  } while (0)

static int exclusive_scan(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, int64_t n, hipStream_t st) {
  return (int)hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, (int)n, st);
}

}  // namespace gjc

extern "C" {

int gj_compile_capacity(const gj_compile_set* set, int64_t* blk_cap, int64_t* slots_cap, int64_t* chunks_cap) {
  if (const int rc = gjc::check_set(set)) return rc;
  const gjc::Geometry g = gjc::geometry_of(set);
  if (blk_cap) *blk_cap = gjc::blocks_upper_bound(set);
  if (slots_cap) *slots_cap = g.E + gjc::kPad * g.J;
  if (chunks_cap) *chunks_cap = g.E / gjc::kChunk + g.S;
  return 0;
}

int gj_compile_workspace_bytes(const gj_compile_set* set, int64_t* bytes) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!bytes) return GJ_E_NULL;
  const gjc::Geometry g = gjc::geometry_of(set);
  if (g.tiles() >= ((int64_t)1 << 31) - 1) return GJ_E_RANGE;
  *bytes = std::max({gjc::carve_blocks(g, nullptr, nullptr), gjc::carve_tiles(g, nullptr, nullptr),
                     gjc::carve_ell(g, nullptr, nullptr),
                     gjc::align_up((int64_t)gjc::select_temp(std::max<int64_t>(g.E, 1))) + 256});
  return 0;
}

int gj_compile_blocks(const gj_compile_set* set, int32_t* blk_v0, int32_t blk_cap, int32_t* counts, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!blk_v0 || !counts || !workspace) return GJ_E_NULL;
  if (blk_cap < 1) return GJ_E_RANGE;
  hipStream_t st = (hipStream_t)stream;
  const gjc::Geometry g = gjc::geometry_of(set);
  gjc::BlocksWs w;
  if (gjc::carve_blocks(g, workspace, &w) > workspace_bytes) return GJ_E_RANGE;
  GJC_HIP(hipMemsetAsync(counts, 0, GJ_COMPILE_COUNTS * sizeof(int32_t), st));
  GJC_HIP(hipMemsetAsync(w.degree, 0, (g.V + 1) * sizeof(int32_t), st));
  if (g.E > 0) {
    gjc::k_degrees<<<gjc::grid_for(g.E), gjc::kThreads, 0, st>>>(
        set->agent, set->venue, g.E, set->n_ext_agents, (int32_t)g.V, w.degree, counts);
  }
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, w.degree, w.rp, g.V + 1, st)) return rc;
  gjc::k_venue_blocks<<<1, 64, 0, st>>>(w.rp, (int32_t)g.V, set->sv_max, set->eb_target, blk_v0, blk_cap, counts);
  return (int)hipGetLastError();
}

int gj_compile_tiles(const gj_compile_set* set, const int32_t* blk_v0, const gj_compile_out* out, int32_t* counts,
                     void* workspace, int64_t workspace_bytes, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!blk_v0 || !out || !counts || !workspace) return GJ_E_NULL;
  if (set->n_blocks < 1) return GJ_E_RANGE;
  if (!out->blk_e0 || !out->e_lv || !out->a_la || !out->tile_sptr || !out->tile_jpos || !out->chunk_ptr || !out->chunk_desc)
    return GJ_E_NULL;
  if (out->e_cls && !set->agent_class) return GJ_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const gjc::Geometry g = gjc::geometry_of(set);
  const int32_t J = (int32_t)g.J, S = (int32_t)g.S;
  if (g.tiles() >= ((int64_t)1 << 31) - 1) return GJ_E_RANGE;
  if (out->slots_cap < g.E + gjc::kPad * g.J || out->chunks_cap < g.E / gjc::kChunk + g.S) return GJ_E_RANGE;
  gjc::TilesWs w;
  if (gjc::carve_tiles(g, workspace, &w) > workspace_bytes) return GJ_E_RANGE;
  GJC_HIP(hipMemsetAsync(w.tile_len_js, 0, (g.tiles() + 1) * sizeof(int32_t), st));
  GJC_HIP(hipMemsetAsync(out->e_lv, 0xFF, out->slots_cap * sizeof(uint16_t), st));
  if (out->e_cls) GJC_HIP(hipMemsetAsync(out->e_cls, 0, out->slots_cap, st));
  const uint64_t* keys = w.keys_a;
  const uint32_t* order = w.vals_a;
  if (g.E > 0) {
    gjc::k_keys<<<gjc::grid_for(g.E), gjc::kThreads, 0, st>>>(set->agent, set->venue, g.E, blk_v0, J, S,
                                                             set->slice_agents, w.keys_a, w.vals_a, w.tile_len_js);
    const int end_bit = 32 + gjc::bits_for((uint64_t)g.tiles());
    size_t tb = w.temp_bytes;
    GJC_HIP(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (int)g.E, 0,
                                                std::min(end_bit, 64), st));
    keys = w.keys_b;
    order = w.vals_b;
  }
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, w.tile_len_js, w.upos_js, g.tiles() + 1, st)) return rc;
  gjc::k_block_slots<<<(J + 1 + gjc::kThreads - 1) / gjc::kThreads, gjc::kThreads, 0, st>>>(w.upos_js, J, S, w.blk_slots);
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, w.blk_slots, out->blk_e0, J + 1, st)) return rc;
  GJC_HIP(hipMemcpyAsync(counts + GJ_CC_SLOTS, out->blk_e0 + J, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  gjc::k_tile_tables<<<gjc::grid_for(g.tiles() + 1), gjc::kThreads, 0, st>>>(w.tile_len_js, w.upos_js, out->blk_e0, J, S,
                                                                            w.len_sj, out->tile_jpos);
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, w.len_sj, out->tile_sptr, g.tiles() + 1, st)) return rc;
  if (g.E > 0) {
    gjc::k_fill<<<gjc::grid_for(g.E), gjc::kThreads, 0, st>>>(keys, order, g.E, set->agent,
                                                             out->e_cls ? set->agent_class : nullptr, w.upos_js,
                                                             out->tile_sptr, out->tile_jpos, J, S, out->a_la, out->e_lv,
                                                             out->e_cls);
  }
  gjc::k_chunk_counts<<<(S + 1 + gjc::kThreads - 1) / gjc::kThreads, gjc::kThreads, 0, st>>>(out->tile_sptr, J, S, w.n_chunks);
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, w.n_chunks, out->chunk_ptr, S + 1, st)) return rc;
  GJC_HIP(hipMemsetAsync(counts + GJ_CC_MULTI, 0, sizeof(int32_t), st));
  gjc::k_descriptors<<<gjc::grid_for(out->chunks_cap), gjc::kThreads, 0, st>>>(
      out->chunk_ptr, out->tile_sptr, out->tile_jpos, J, S, (int32_t)g.E, reinterpret_cast<int4*>(out->chunk_desc), counts);
  return (int)hipGetLastError();
}

int gj_compile_wide_descriptors(const gj_compile_set* set, const gj_compile_out* out, int32_t n_chunks, int32_t* desc8,
                                int32_t* counts, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!out || !counts || !out->tile_sptr || !out->tile_jpos || !out->chunk_ptr) return GJ_E_NULL;
  if (set->n_blocks < 1 || n_chunks < 0) return GJ_E_RANGE;
  if (n_chunks == 0) return 0;
  if (!desc8) return GJ_E_NULL;
  gjc::k_wide_descriptors<<<gjc::grid_for(n_chunks), gjc::kThreads, 0, (hipStream_t)stream>>>(
      out->chunk_ptr, out->tile_sptr, out->tile_jpos, set->n_blocks, set->n_slices, n_chunks, desc8, counts);
  return (int)hipGetLastError();
}

int gj_compile_ell_degrees(const gj_compile_set* set, int32_t* degree, int32_t* counts, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!degree || !counts) return GJ_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  GJC_HIP(hipMemsetAsync(degree, 0, (set->n_agents + 1) * sizeof(int32_t), st));
  GJC_HIP(hipMemsetAsync(counts + GJ_CC_OWNED_EDGES, 0, 2 * sizeof(int32_t), st));
  if (set->n_edges > 0) {
    gjc::k_ell_degrees<<<gjc::grid_for_reduction(set->n_edges), gjc::kThreads, 0, st>>>(set->agent, set->n_edges, set->n_agents,
                                                                             degree, counts);
    if (set->n_agents > 0)
      gjc::k_max_degree<<<gjc::grid_for_reduction(set->n_agents), gjc::kThreads, 0, st>>>(degree, set->n_agents, counts);
  }
  return (int)hipGetLastError();
}

int gj_compile_ell(const gj_compile_set* set, int32_t ell_k, int64_t rows, const int32_t* degree, uint16_t* ell,
                   void* workspace, int64_t workspace_bytes, int32_t* counts, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!degree || !ell || !workspace) return GJ_E_NULL;
  if (ell_k < 2 || (ell_k & (ell_k - 1)) || rows < set->n_agents || set->n_venues > 65535) return GJ_E_RANGE;
  hipStream_t st = (hipStream_t)stream;
  const gjc::Geometry g = gjc::geometry_of(set);
  gjc::EllWs w;
  if (gjc::carve_ell(g, workspace, &w) > workspace_bytes) return GJ_E_RANGE;
  GJC_HIP(hipMemsetAsync(ell, 0xFF, (size_t)rows * ell_k * sizeof(uint16_t), st));
  if (g.E == 0) return 0;
  gjc::k_ell_keys<<<gjc::grid_for(g.E), gjc::kThreads, 0, st>>>(set->agent, g.E, g.n_agents, w.keys_a, w.vals_a);
  size_t tb = w.temp_bytes;
  GJC_HIP(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b, (int)g.E, 0,
                                              std::min(32, gjc::bits_for((uint64_t)g.n_agents + 1)), st));
  if (const int rc = gjc::exclusive_scan(w.temp, w.temp_bytes, degree, w.rowptr, g.n_agents + 1, st)) return rc;
  gjc::k_ell_fill<<<gjc::grid_for(g.E), gjc::kThreads, 0, st>>>(w.keys_b, w.vals_b, g.E, g.n_agents, w.rowptr, set->venue,
                                                               rows, ell_k, ell, counts);
  return (int)hipGetLastError();
}

int gj_compile_explicit_slots(const gj_compile_set* set, const gj_compile_out* out, int32_t* slots, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!out || !out->tile_sptr || !out->tile_jpos) return GJ_E_NULL;
  if (set->n_blocks < 1) return GJ_E_RANGE;
  if (set->n_edges == 0) return 0;
  if (!slots) return GJ_E_NULL;
  const int64_t n_tiles = (int64_t)set->n_slices * set->n_blocks;
  gjc::k_explicit_slots<<<gjc::grid_for(set->n_edges), gjc::kThreads, 0, (hipStream_t)stream>>>(
      out->tile_sptr, out->tile_jpos, n_tiles, set->n_edges, slots);
  return (int)hipGetLastError();
}

int gj_compile_multi_slots(const gj_compile_set* set, const gj_compile_out* out, int32_t n_chunks, int32_t wide,
                           int32_t* desc, int32_t n_multi, int32_t* multi_slots, int32_t* counts, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!out || !out->tile_sptr || !out->tile_jpos || !out->chunk_ptr || !counts) return GJ_E_NULL;
  if (set->n_blocks < 1 || n_chunks < 0 || n_multi < 0) return GJ_E_RANGE;
  if (n_chunks == 0 || n_multi == 0) return 0;
  if (!desc || !multi_slots || !workspace) return GJ_E_NULL;
  // workspace: flags [n_chunks + 1] | rows [n_chunks + 1] | scan temp
  const int64_t n1 = (int64_t)n_chunks + 1;
  size_t temp_bytes = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n1);
  const int64_t need = 2 * gjc::align_up(n1 * 4) + gjc::align_up((int64_t)temp_bytes);
  if (need > workspace_bytes) return GJ_E_RANGE;
  char* base = static_cast<char*>(workspace);
  int32_t* flags = reinterpret_cast<int32_t*>(base);
  int32_t* rows = reinterpret_cast<int32_t*>(base + gjc::align_up(n1 * 4));
  void* temp = base + 2 * gjc::align_up(n1 * 4);
  hipStream_t st = (hipStream_t)stream;
  gjc::k_multi_flags<<<gjc::grid_for(n1), gjc::kThreads, 0, st>>>(desc, n_chunks, wide, flags);
  if (const int rc = gjc::exclusive_scan(temp, temp_bytes, flags, rows, n1, st)) return rc;
  gjc::k_multi_fill<<<gjc::grid_for((int64_t)n_chunks * 64), gjc::kThreads, 0, st>>>(
      out->chunk_ptr, out->tile_sptr, out->tile_jpos, set->n_blocks, set->n_slices, n_chunks, wide, rows, desc, multi_slots,
      counts);
  return (int)hipGetLastError();
}

// ---- run form ---------------------------------------------------------------------------------------------------------
int gj_compile_runs_pick(const gj_compile_set* set, int32_t* vmin, int32_t* pick, uint8_t* keep, int32_t* win_lo,
                         int32_t* win_n, int32_t* counts, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!vmin || !pick || !counts || !win_lo || !win_n || (set->n_edges > 0 && !keep)) return GJ_E_NULL;
  if (set->n_venues >= gjc::kNoVenue) return GJ_E_RANGE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t A = set->n_agents, E = set->n_edges;
  const int32_t n_own_slices = (int32_t)std::max<int64_t>(1, (A + set->slice_agents - 1) / set->slice_agents);
  GJC_HIP(hipMemsetAsync(counts, 0, GJ_COMPILE_COUNTS * sizeof(int32_t), st));
  if (A > 0) {
    GJC_HIP(hipMemsetD32Async((hipDeviceptr_t)vmin, gjc::kNoVenue, (size_t)A, st));
    GJC_HIP(hipMemsetD32Async((hipDeviceptr_t)pick, gjc::kNoVenue, (size_t)A, st));
  }
  if (E > 0 && A > 0) {
    gjc::k_run_vmin<<<gjc::grid_for(E), gjc::kThreads, 0, st>>>(set->agent, set->venue, E, A, vmin);
    gjc::k_run_pick<<<gjc::grid_for(E), gjc::kThreads, 0, st>>>(set->agent, set->venue, E, A, vmin, pick);
  }
  if (E > 0) gjc::k_run_keep<<<gjc::grid_for_reduction(E), gjc::kThreads, 0, st>>>(set->agent, E, A, pick, keep, counts);
  if (A > 1) gjc::k_run_sorted<<<gjc::grid_for(A), gjc::kThreads, 0, st>>>(vmin, A, counts);
  gjc::k_run_windows<<<(n_own_slices + gjc::kThreads - 1) / gjc::kThreads, gjc::kThreads, 0, st>>>(
      vmin, A, set->slice_agents, n_own_slices, win_lo, win_n, counts);
  return (int)hipGetLastError();
}

int gj_compile_runs_rest(const gj_compile_set* set, const uint8_t* keep, int64_t* agent_out, int64_t* venue_out,
                         void* workspace, int64_t workspace_bytes, int32_t* counts, void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!counts || !workspace) return GJ_E_NULL;
  if (set->n_edges == 0) return 0;
  if (!keep || !agent_out || !venue_out) return GJ_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  size_t tb = gjc::select_temp(set->n_edges);
  if ((int64_t)gjc::align_up((int64_t)tb) + 256 > workspace_bytes) return GJ_E_RANGE;
  int32_t* n_out = reinterpret_cast<int32_t*>(workspace);            // the selection's own counter, then its scratch
  void* temp = reinterpret_cast<char*>(workspace) + 256;
  GJC_HIP(hipcub::DeviceSelect::Flagged(temp, tb, set->agent, keep, agent_out, n_out, (int)set->n_edges, st));
  GJC_HIP(hipcub::DeviceSelect::Flagged(temp, tb, set->venue, keep, venue_out, n_out, (int)set->n_edges, st));
  return (int)hipGetLastError();
}

int gj_compile_runs_index(const gj_compile_set* set, const int32_t* vmin, const int32_t* blk_v0, int32_t n_blocks,
                          int64_t rows, const int32_t* win_lo, uint16_t* pv_blk, uint16_t* pv_win, int32_t* blk_r0,
                          void* stream) {
  if (const int rc = gjc::check_set(set)) return rc;
  if (!vmin || !blk_v0 || !win_lo || !pv_blk || !pv_win || !blk_r0) return GJ_E_NULL;
  if (n_blocks < 1 || rows < set->n_agents) return GJ_E_RANGE;
  hipStream_t st = (hipStream_t)stream;
  if (rows > 0)
    gjc::k_run_index<<<gjc::grid_for(rows), gjc::kThreads, 0, st>>>(vmin, set->n_agents, rows, set->slice_agents, blk_v0,
                                                                    n_blocks, win_lo, pv_blk, pv_win);
  gjc::k_run_blk_r0<<<(n_blocks + 1 + gjc::kThreads - 1) / gjc::kThreads, gjc::kThreads, 0, st>>>(vmin, set->n_agents, blk_v0,
                                                                                                  n_blocks, blk_r0);
  return (int)hipGetLastError();
}

}  // extern "C"
