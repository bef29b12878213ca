/* The graph compile of libgradjune_hip.so from plain C (include/gradjune_hip.h, "graph compile"):
 * the reference's unsorted COO edge_index of one edge set -> the tiled layout, on the device, caller-owned buffers.
 *
 *   gcc examples/compile_demo.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       -Lgradabm-june_amd/grad_june_amd/lib -lgradjune_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/gradabm-june_amd/grad_june_amd/lib -Wl,-rpath,/opt/rocm/lib -o /tmp/compile_demo && /tmp/compile_demo
 *
 * 200 000 agents, 5 000 venues, 300 000 random memberships.  Checks what the layout promises: every edge appears once in
 * each of the two tile orders, a venue block's slots hold local venue ids below its venue count, the slice-major local
 * agent ids are below the slice size, the tile prefix sums end at the edge count, and the ELL rows of the direct form
 * hold every owned agent's venues; and the run form of a household-ordered set (gj_compile_runs_*).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gradjune_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_GJ(x) do { int rc_ = (x); if (rc_ != GJ_OK) { fprintf(stderr, "%s: [%d] %s\n", #x, rc_, gj_error_string(rc_)); return 1; } } while (0)
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static void* dmalloc(size_t bytes) {
  void* p = NULL;
  return hipMalloc(&p, bytes ? bytes : 4) == hipSuccess ? p : NULL;
}

int main(void) {
  const int64_t A = 200000, E = 300000;
  const int32_t V = 5000, SA = 4096, S = (int32_t)((A + SA - 1) / SA);
  int64_t *agent = (int64_t*)malloc(E * 8), *venue = (int64_t*)malloc(E * 8);
  uint64_t x = 88172645463325252ull;                    /* xorshift: any reproducible edge list will do */
  for (int64_t e = 0; e < E; ++e) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    agent[e] = (int64_t)(x % (uint64_t)A);
    venue[e] = (int64_t)((x >> 32) % (uint64_t)V);
  }
  int64_t *d_agent = (int64_t*)dmalloc(E * 8), *d_venue = (int64_t*)dmalloc(E * 8);
  CHECK(d_agent && d_venue);
  CHECK_HIP(hipMemcpy(d_agent, agent, E * 8, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_venue, venue, E * 8, hipMemcpyHostToDevice));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  gj_compile_set cs = {d_agent, d_venue, NULL, E, A, A, V, S, SA, 1024 /* venues per block */, 2048 /* edges per block: tiles of ~40 edges, so that some 64-edge chunks span more than two tiles */, 0};
  int64_t blk_cap = 0, slots_cap = 0, chunks_cap = 0, ws_bytes = 0;
  CHECK_GJ(gj_compile_capacity(&cs, &blk_cap, NULL, NULL));
  CHECK_GJ(gj_compile_workspace_bytes(&cs, &ws_bytes));
  void* ws = dmalloc((size_t)ws_bytes);
  int32_t *counts = (int32_t*)dmalloc(GJ_COMPILE_COUNTS * 4), *blk_v0 = (int32_t*)dmalloc((blk_cap + 1) * 4);
  CHECK(ws && counts && blk_v0);
  int32_t c[GJ_COMPILE_COUNTS];

  /* stage 1: venue degrees, range checks, venue blocks */
  CHECK_GJ(gj_compile_blocks(&cs, blk_v0, (int32_t)blk_cap, counts, ws, ws_bytes, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost));
  CHECK(c[GJ_CC_ERROR] == 0 && c[GJ_CC_BLOCKS] >= 1);
  const int32_t J = c[GJ_CC_BLOCKS];
  cs.n_blocks = J;

  /* stage 2: the arrays of gj_tiled_set */
  CHECK_GJ(gj_compile_capacity(&cs, NULL, &slots_cap, &chunks_cap));
  gj_compile_out out = {(int32_t*)dmalloc((J + 1) * 4), (uint16_t*)dmalloc(slots_cap * 2), NULL, (uint16_t*)dmalloc(E * 2),
                        (int32_t*)dmalloc(((int64_t)S * J + 1) * 4), (int32_t*)dmalloc((int64_t)S * J * 4),
                        (int32_t*)dmalloc((S + 1) * 4), (int32_t*)dmalloc(chunks_cap * 4 * 4), slots_cap, chunks_cap};
  CHECK(out.blk_e0 && out.e_lv && out.a_la && out.tile_sptr && out.tile_jpos && out.chunk_ptr && out.chunk_desc);
  CHECK_GJ(gj_compile_tiles(&cs, blk_v0, &out, counts, ws, ws_bytes, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost));
  CHECK(c[GJ_CC_ERROR] == 0);
  const int32_t n_slots = c[GJ_CC_SLOTS], n_chunks = c[GJ_CC_CHUNKS];
  printf("J = %d venue blocks, %d slices, %d block-major slots for %lld edges, %d chunks (%d spanning > 2 tiles)\n", J, S,
         n_slots, (long long)E, n_chunks, c[GJ_CC_MULTI]);

  int32_t *h_v0 = (int32_t*)malloc((J + 1) * 4), *h_e0 = (int32_t*)malloc((J + 1) * 4);
  int32_t* h_sptr = (int32_t*)malloc(((int64_t)S * J + 1) * 4);
  uint16_t *h_lv = (uint16_t*)malloc((size_t)n_slots * 2), *h_la = (uint16_t*)malloc(E * 2);
  CHECK_HIP(hipMemcpy(h_v0, blk_v0, (J + 1) * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_e0, out.blk_e0, (J + 1) * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_sptr, out.tile_sptr, ((int64_t)S * J + 1) * 4, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_lv, out.e_lv, (size_t)n_slots * 2, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_la, out.a_la, E * 2, hipMemcpyDeviceToHost));
  CHECK(h_v0[0] == 0 && h_v0[J] == V && h_e0[0] == 0 && h_e0[J] == n_slots && h_sptr[(int64_t)S * J] == E);
  /* block-major order: per venue, as many slots as the venue has edges; pads carry 0xFFFF */
  int64_t* deg = (int64_t*)calloc(V, 8);
  int64_t real = 0;
  for (int32_t j = 0; j < J; ++j) {
    CHECK(h_e0[j] % 8 == 0 && h_v0[j + 1] - h_v0[j] <= 1024);
    for (int32_t i = h_e0[j]; i < h_e0[j + 1]; ++i) {
      if (h_lv[i] == 0xFFFF) continue;
      CHECK(h_lv[i] < h_v0[j + 1] - h_v0[j]);
      deg[h_v0[j] + h_lv[i]]++;
      real++;
    }
  }
  CHECK(real == E);
  for (int64_t e = 0; e < E; ++e) deg[venue[e]]--;
  for (int32_t v = 0; v < V; ++v) CHECK(deg[v] == 0);
  /* slice-major order: per agent, as many entries as the agent has edges */
  int64_t* adeg = (int64_t*)calloc(A, 8);
  for (int32_t s = 0; s < S; ++s)
    for (int32_t i = h_sptr[(int64_t)s * J]; i < h_sptr[(int64_t)(s + 1) * J]; ++i) {
      CHECK(h_la[i] < SA && (int64_t)s * SA + h_la[i] < A);
      adeg[(int64_t)s * SA + h_la[i]]++;
    }
  int64_t max_deg = 0;
  for (int64_t e = 0; e < E; ++e) adeg[agent[e]]--;
  for (int64_t a = 0; a < A; ++a) CHECK(adeg[a] == 0);

  /* the chunks a 4-word descriptor cannot express (they span more than two tiles) get a row of 64 explicit slots each
   * (gj_tiled_set.multi_slots, ABI 6); every row is checked against the tile tables here */
  const int32_t n_multi = c[GJ_CC_MULTI];
  if (n_multi > 0) {
    int32_t* multi_slots = (int32_t*)dmalloc((size_t)n_multi * 64 * 4);
    const int64_t ws2_bytes = 2 * (4 * ((int64_t)n_chunks + 1) + 256) + (1 << 20);
    void* ws2 = dmalloc((size_t)ws2_bytes);
    CHECK(multi_slots && ws2);
    CHECK_GJ(gj_compile_multi_slots(&cs, &out, n_chunks, 0, out.chunk_desc, n_multi, multi_slots, counts, ws2, ws2_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    int32_t* h_desc = (int32_t*)malloc((size_t)n_chunks * 16);
    int32_t* h_rows = (int32_t*)malloc((size_t)n_multi * 64 * 4);
    int32_t* h_jpos = (int32_t*)malloc((int64_t)S * J * 4);
    int32_t* h_cptr = (int32_t*)malloc((S + 1) * 4);
    CHECK_HIP(hipMemcpy(h_desc, out.chunk_desc, (size_t)n_chunks * 16, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_rows, multi_slots, (size_t)n_multi * 64 * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_jpos, out.tile_jpos, (int64_t)S * J * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_cptr, out.chunk_ptr, (S + 1) * 4, hipMemcpyDeviceToHost));
    int32_t seen = 0;
    for (int32_t s = 0; s < S; ++s)
      for (int32_t ch = h_cptr[s]; ch < h_cptr[s + 1]; ++ch) {
        if (((uint32_t)h_desc[4 * ch + 2] >> 16) == 0) continue;
        const int32_t row = h_desc[4 * ch + 3];                 /* the j0 field of such a chunk: its row */
        CHECK(row == seen);                                    /* rows are in chunk order */
        ++seen;
        const int64_t first = h_sptr[(int64_t)s * J] + 64 * (int64_t)(ch - h_cptr[s]), seg_end = h_sptr[(int64_t)(s + 1) * J];
        int64_t t = (int64_t)s * J;
        for (int lane = 0; lane < 64; ++lane) {
          const int64_t pos = first + lane;
          if (pos >= seg_end) { CHECK(h_rows[64 * row + lane] == 0); continue; }
          while (h_sptr[t + 1] <= pos) ++t;                    /* the tile that holds the position */
          CHECK(h_rows[64 * row + lane] == h_jpos[t] + (int32_t)(pos - h_sptr[t]));
        }
      }
    CHECK(seen == n_multi);
    printf("multi chunks: %d rows of explicit slots, every slot = tile_jpos + offset\n", n_multi);
    free(h_desc); free(h_rows); free(h_jpos); free(h_cptr);
  }

  /* the ELL rows of the direct form of pass 2 */
  int32_t* degree = (int32_t*)dmalloc((A + 1) * 4);
  CHECK(degree);
  CHECK_GJ(gj_compile_ell_degrees(&cs, degree, counts, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost));
  CHECK(c[GJ_CC_OWNED_EDGES] == E);
  max_deg = c[GJ_CC_MAX_DEGREE];
  int32_t K = 2;
  while (K < max_deg) K *= 2;
  const int64_t rows = (int64_t)S * SA;
  uint16_t* ell = (uint16_t*)dmalloc((size_t)rows * K * 2);
  CHECK(ell);
  CHECK_GJ(gj_compile_ell(&cs, K, rows, degree, ell, ws, ws_bytes, counts, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  uint16_t* h_ell = (uint16_t*)malloc((size_t)rows * K * 2);
  CHECK_HIP(hipMemcpy(h_ell, ell, (size_t)rows * K * 2, hipMemcpyDeviceToHost));
  int64_t entries = 0, vsum = 0, vref = 0;
  for (int64_t i = 0; i < rows * K; ++i)
    if (h_ell[i] != 0xFFFF) { entries++; vsum += h_ell[i]; }
  for (int64_t e = 0; e < E; ++e) vref += venue[e];
  CHECK(entries == E && vsum == vref);
  printf("ELL: K = %d columns (largest agent degree %lld), %lld entries\n", K, (long long)max_deg, (long long)entries);

  /* the run form of a set the agents are ordered by: a household world - agent a lives in household a / 3 and a third of
   * the agents visit a second, random one.  One edge per agent (its first edge to its smallest venue) leaves the tiled
   * arrays; the others are compacted, in COO order, as the input of gj_compile_blocks / _tiles. */
  {
    const int64_t Eh = A + A / 3;
    const int32_t Vh = (int32_t)((A + 2) / 3);
    int64_t *ha = (int64_t*)malloc(Eh * 8), *hv = (int64_t*)malloc(Eh * 8);
    for (int64_t a = 0; a < A; ++a) { ha[a] = a; hv[a] = a / 3; }
    for (int64_t i = 0; i < A / 3; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      ha[A + i] = 3 * i;
      hv[A + i] = (3 * i) / 3 + 1 + (int64_t)(x % (uint64_t)(Vh - (3 * i) / 3 - 1 > 0 ? Vh - (3 * i) / 3 - 1 : 1));   /* a LARGER venue id */
      if (hv[A + i] >= Vh) hv[A + i] = Vh - 1;
    }
    int64_t *d_ha = (int64_t*)dmalloc(Eh * 8), *d_hv = (int64_t*)dmalloc(Eh * 8);
    CHECK(d_ha && d_hv);
    CHECK_HIP(hipMemcpy(d_ha, ha, Eh * 8, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_hv, hv, Eh * 8, hipMemcpyHostToDevice));
    gj_compile_set hs = {d_ha, d_hv, NULL, Eh, A, (int64_t)S * SA, Vh, S, SA, 16384, 131072, 0};
    int64_t hws_bytes = 0;
    CHECK_GJ(gj_compile_workspace_bytes(&hs, &hws_bytes));
    void* hws = dmalloc((size_t)hws_bytes);
    int32_t *vmin = (int32_t*)dmalloc(A * 4), *pick = (int32_t*)dmalloc(A * 4), *win_lo = (int32_t*)dmalloc(S * 4),
            *win_n = (int32_t*)dmalloc(S * 4);
    uint8_t* keep = (uint8_t*)dmalloc(Eh);
    CHECK(hws && vmin && pick && win_lo && win_n && keep);
    CHECK_GJ(gj_compile_runs_pick(&hs, vmin, pick, keep, win_lo, win_n, counts, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(c, counts, sizeof(c), hipMemcpyDeviceToHost));
    CHECK(c[GJ_CC_RUN_PRIMARY] == A && c[GJ_CC_RUN_UNSORTED] == 0 && c[GJ_CC_OWNED_EDGES] == Eh);
    CHECK(c[GJ_CC_RUN_WINDOW] >= SA / 3 && c[GJ_CC_RUN_WINDOW] <= SA / 3 + 2);       /* three agents per household */
    int64_t *rest_a = (int64_t*)dmalloc((Eh - A) * 8), *rest_v = (int64_t*)dmalloc((Eh - A) * 8);
    CHECK(rest_a && rest_v);
    CHECK_GJ(gj_compile_runs_rest(&hs, keep, rest_a, rest_v, hws, hws_bytes, counts, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    int64_t* h_rest = (int64_t*)malloc((Eh - A) * 8);
    CHECK_HIP(hipMemcpy(h_rest, rest_a, (Eh - A) * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < Eh - A; ++i) CHECK(h_rest[i] == 3 * i);                   /* the second visits, in COO order */
    printf("run form: %d primary edges leave the tiled arrays, %lld stay; widest slice window %d venues\n",
           c[GJ_CC_RUN_PRIMARY], (long long)(Eh - A), c[GJ_CC_RUN_WINDOW]);
  }

  /* argument errors never touch the device */
  cs.slice_agents = 70000;
  CHECK(gj_compile_capacity(&cs, NULL, NULL, NULL) == GJ_E_RANGE);
  printf("ok\n");
  return 0;
}
