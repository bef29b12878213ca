/*
 * gradjune_hip.h - C ABI of the MI355X (gfx950) infection message-passing library.
 *
 * This is the drop-in boundary for ONE path of GradABM-JUNE: the per-timestep infection
 * step.  The reference has no native code and no FFI for this path:
 * it runs as eager PyTorch + torch_geometric ops on the CPU.  Each entry point below
 * therefore names the reference PYTHON interface it replaces (paths relative to the
 * reference tree, file:line) - these are the call sites a maintainer re-binds with the
 * ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes.  No torch types, no C++ types.
 *   - Every pointer marked "device" is a HIP device pointer valid on the current device.
 *     Every buffer (inputs, outputs, workspace) is allocated and owned by the caller; the
 *     library never allocates, frees or retains device memory and keeps no global state,
 *     so it is re-entrant and one host thread per process/GPU may call it freely.
 *   - All launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     legacy default stream).  No entry point synchronises, so calls may be captured into a
 *     hipGraph.
 *   - Return value: 0 = ok; negative = argument error (GJ_E_*); positive = hipError_t of a
 *     failed launch.  Nothing throws across the ABI.  gj_error_string() describes a code.
 *   - Index arrays are int32 (the reference's int64 COO is compiled once on the host into
 *     int32 CSR, see gj_plan below); all values are IEEE fp32, computed with FMA contraction
 *     off so that the op sequence matches the reference's ATen ops.
 */
#ifndef GRADJUNE_HIP_H
#define GRADJUNE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GJ_ABI_VERSION 6

#define GJ_MAX_SETS 12        /* distinct agent<->venue edge sets in a world (reference: 6)   */
#define GJ_MAX_NETS 16        /* infection networks active in one step (reference: <= 11)      */
#define GJ_MAX_NETS_PER_SET 8 /* networks sharing one edge set (reference: 6 leisure classes)  */
#define GJ_TABLE_SIZE 400     /* one leisure table: [day_type 2][sex 2][age 100] floats        */

/* error codes (negative) */
#define GJ_OK 0
#define GJ_E_NULL (-1)     /* required pointer is NULL                      */
#define GJ_E_RANGE (-2)    /* count / index out of the documented range     */
#define GJ_E_PLAN (-3)     /* inconsistent plan (sizes, strides, schedule)  */
#define GJ_E_NODEVICE (-4) /* no HIP device / wrong architecture            */

/* how a network derives its per-agent transmission / susceptibility from the raw values
 * (reference: grad_june/infection_networks/base.py:47-59,144-149;
 *             grad_june/infection_networks/leisure_network.py:61-85,107-120)           */
typedef enum gj_mask_kind {
  GJ_MASK_RAW = 0,      /* household: raw transmission / susceptibility, ignores quarantine */
  GJ_MASK_Q = 1,        /* q[a] * value                                                      */
  GJ_MASK_QL = 2,       /* q[a] * L[day, sex[a], age[a]] * value                             */
  GJ_MASK_QL_AGE75 = 3  /* as QL; susceptibility additionally * (age[a] > 75) (care_visit)   */
} gj_mask_kind;

/* One agent<->venue edge set, compiled once from the reference's COO
 * `data["attends_<set>"].edge_index` (int64 [2,E], row 0 agent, row 1 venue;
 * grad_june/june_world_loader/network_loader.py:30-44) into two int32 CSR views.
 * Edge order inside every CSR row is the COO order (stable sort), so sums taken in row
 * order reproduce the reference's scatter_add_ order.                                      */
typedef struct gj_edge_set {
  int64_t n_venues;
  int64_t n_edges;
  const int32_t* v_rowptr;  /* device [n_venues+1]  CSR by venue                            */
  const int32_t* v_agent;   /* device [n_edges]     agent of each edge, venue-major         */
  const float* v_pcontact;  /* device [n_venues]    clamp(1/(people-1),0,1), base.py:63-69  */
  const int32_t* a_rowptr;  /* device [n_agents+1]  CSR by agent                            */
  const int32_t* a_venue;   /* device [n_edges]     venue of each edge, agent-major         */
  float* cum;               /* device [n_venues*cum_stride] workspace: pass-1 out, pass-2 in */
  int32_t cum_stride;       /* floats per venue in `cum` (>= networks active on this set)   */
  int32_t _pad;
} gj_edge_set;

/* One workgroup's share of pass 1 (built on the host by the graph compiler).
 * kind 0: STREAM - venues [v0,v1) whose edges [e0,e1) (<= GJ_STREAM_EDGES) are staged through
 *         LDS and reduced with `lanes` (1,4,16 or 64) lanes per venue.
 * kind 1: LONG   - edges [e0,e1) of the single venue v0; the partial sum goes to
 *         partial[slot*cum_stride + k] and is combined in chunk order by a second kernel.  */
typedef struct gj_block {
  int32_t set;
  int32_t kind;
  int32_t v0, v1;
  int32_t e0, e1;
  int32_t slot;
  int32_t lanes;
} gj_block;

/* venue whose degree exceeds the stream capacity: partial slots [slot0,slot1) -> cum[v] */
typedef struct gj_long_row {
  int32_t set;
  int32_t v;
  int32_t slot0, slot1;
} gj_long_row;

#define GJ_STREAM_EDGES 2048 /* max edges of a STREAM block (256 threads x 8)               */

/* ---- tiled ("propagation-blocked") layout: the fast path of both passes -------------------
 * Agents are cut into n_slices slices of slice_agents consecutive agents; the venues of a set
 * into blocks of consecutive venues; edge (a, v) belongs to tile (slice(a), block(v)).  Every
 * random access of the two passes then hits LDS (a slice of transmissions / a block of venue
 * sums) and HBM only sees four coalesced streams per step (24 B per edge):
 *   A  one workgroup per slice :  val[block-major pos] = x_slice[a_la]
 *   B  one workgroup per block :  sums[e_lv] += val   (LDS integer atomics); cum = beta*p_contact*sums
 *   C  same workgroup (fused)  :  val[i] = cum[e_lv[i]]          (in place)
 *   D  one workgroup per slice :  acc[a_la] += val[block-major pos]; epilogue a7-a9
 * A tile is contiguous in both the slice-major (s, j) and the block-major (j, s) edge order.
 * Sums are accumulated with 64-bit fixed-point LDS atomics (resolution 2^-36 per venue, 2^-32 per agent).  The
 * reference has no window (its sums stay finite and the epilogue clamps them, base.py:136-138); here a TERM beyond
 * +-16384 (venue; less for sets with venues of > 4096 attendees, see max_venue_edges) / +-262144 (agent), or an
 * infinity, SATURATES its sum: the element reads back +-1e30, which every later stage carries to the epilogue's clamp
 * (exp(-100 dt): infected with certainty) while a zero factor - p_contact of an empty venue, susceptibility 0, a
 * quarantine mask - still gives 0, as in the reference.  A NaN term, or saturation in both directions, reads back NaN.
 * No sum can wrap: terms x attendees is bounded per set (max_venue_edges), edges per agent by 4096.
 * integer adds are order-independent, so results are bitwise reproducible from run to run, and
 * agree with the CSR path to fp32 rounding.                                                  */
typedef struct gj_tiled_set {
  int32_t n_blocks;          /* J: venue blocks of this set                                  */
  int32_t max_block_venues;  /* largest block of this set (sizes the LDS of phases B/C)      */
  int32_t desc_wide;         /* 0: chunk_desc holds 4, 1: 8 int32 per chunk (see chunk_desc); 2: no descriptors -
                                chunk_desc holds int32 [E], the block-major slot of every slice-major edge
                                (sets with tiles of a few edges)                                  */
  int32_t ell_k;             /* 0: pass 2 of this set runs through phases C + D.  2, 4 or 8: "direct"
                                form - phase C is skipped, phase D reads the venues' cum from an LDS
                                table through `ell` (sets with <= 65534 venues, see `ell`)        */
  const int32_t* blk_v0;     /* device [J+1]   venue range of block j                        */
  const int32_t* blk_e0;     /* device [J+1]   block-major SLOT range of block j; every block is
                                               padded to a multiple of 8 slots (16-byte accesses) */
  const uint16_t* e_lv;      /* device [slots] venue index local to its block, block-major;
                                               0xFFFF marks a pad slot                        */
  const uint8_t* e_cls;      /* device [slots] agent_class of the edge's agent, block-major
                                               (sets that carry leisure tables; else NULL)    */
  const uint16_t* a_la;      /* device [E]     agent index local to its slice, slice-major   */
  const int32_t* tile_sptr;  /* device [S*J+1] slice-major prefix: tile (s,j) = [sptr[s*J+j], sptr[s*J+j+1]) */
  const int32_t* tile_jpos;  /* device [S*J]   block-major start slot of tile (s,j)          */
  const int32_t* chunk_ptr;  /* device [S+1]   first 64-edge chunk of slice s's slice-major segment */
  const int32_t* chunk_desc; /* device [4*chunks] per 64-edge chunk: slot0, slot1, split|multi<<16, j0:
                                the first `split` edges map to block-major slots slot0.., the rest
                                (next non-empty tile) to slot1..; multi: spans > 2 tiles, lanes
                                resolve through tile_sptr/tile_jpos starting at block j0.
                                desc_wide (sets with small tiles): [8*chunks], base_0..base_5,
                                start_1|start_2<<8|start_3<<16|start_4<<24, start_5|multi<<8|j0<<9:
                                lanes [start_k, start_k+1) lie in one tile and map to slots
                                base_k + lane (start_0 = 0, unused segments start at 64);
                                multi: more than 6 tiles                                        */
  float* val;                /* device [slots] workspace: per-edge value (phase A->B, C->D)  */
  const uint16_t* ell;       /* device [ell_k / 2 planes][owned slices * slice_agents][2] or NULL: the venue
                                ids (whole-set numbering) of each OWNED agent's edges in COO order, columns in pairs (plane p holds an agent's edges 2p and
                                2p + 1), 0xFFFF = none.  Replaces a_la / val / chunk_desc reads of phase D and
                                all of phase C for this set: 2 * ell_k bytes per agent                    */
  /* "Run form" of ONE edge per owned agent (NULL pointers: none).  When the owned agents are ordered by their smallest
   * venue of this set (the household-major order of graph.locality_order; agents without an edge last), the agents
   * whose smallest venue is v are consecutive, and that "primary" edge (a, vmin(a)) needs no entry in the arrays
   * above: phase B reads the transmissions of block j's agents [run_blk_r0[j], run_blk_r0[j+1]) straight from the
   * per-agent array next to run_pv_blk (phase A has nothing to scatter for them); phase D stages the window
   * [run_win_lo[s], run_win_lo[s] + run_win_n[s]) of the venues' cum through LDS and reads it through run_pv_win
   * (phase C has nothing to write for them).  The set's other edges stay in the tiled arrays.                     */
  const uint16_t* run_pv_blk; /* device [owned slices * slice_agents] vmin(a) - blk_v0[block of vmin(a)], 0xFFFF = none */
  const uint16_t* run_pv_win; /* device [owned slices * slice_agents] vmin(a) - run_win_lo[slice of a], 0xFFFF = none    */
  const int32_t* run_blk_r0;  /* device [J+1]  first agent whose primary venue lies in block j (non-decreasing)        */
  const int32_t* run_win_lo;  /* device [owned slices] first venue of the slice's window                              */
  const int32_t* run_win_n;   /* device [owned slices] venues in the window (0: no primary edge in the slice)         */
  int32_t run_max_window;     /* max run_win_n (sizes the LDS table of phase D), <= 32768                             */
  int32_t run_tiled_edges;    /* with a run form: the edges the tiled arrays above hold (the set's n_edges minus the
                                 primary ones; may be 0 - every person lives in exactly one household)               */
  /* Pass 1 in the "direct" form (NULL: through phases A + B).  For a set in the direct form of pass 2 (ell_k != 0)
   * whose edges ALL belong to owned agents, the ELL rows serve pass 1 too: gj_tiled.presum_wgs workgroups each take a
   * contiguous range of owned agents, add every edge's term into an LDS table of 64-bit fixed-point sums per (venue,
   * network) and write it to presum[workgroup][venue][network]; a second launch adds the tables up and applies
   * beta * p_contact.  Exact (integer sums), so cum is the same bit for bit as through phases A + B.              */
  int64_t* presum;            /* device [presum_wgs][n_venues * cum_stride] workspace or NULL                         */
  const int32_t* multi_slots; /* device [n_multi * 64] or NULL: the block-major slots of the 64 edges of every chunk that spans
                                 more tiles than its descriptor has segments ("multi"); the descriptor's j0 field holds the
                                 chunk's row.  NULL: such a chunk's lanes walk tile_sptr / tile_jpos from block j0
                                 (rounds 1-3: ~50x the cost of a chunk - 1 % of them doubled phase A on a world with a
                                 geography).  ABI 6                                                                     */
  int32_t max_venue_edges;    /* edges of the set's LARGEST venue (run-form primaries included).  Bounds the terms of one
                                 venue sum: the window of a term is min(16384, 2^26 / next_pow2(max_venue_edges)), so that
                                 no sum can leave its 64 bits.  0 = not stated: 16384, no such guarantee (ABI 6)        */
  int32_t _pad_mve;
} gj_tiled_set;

typedef struct gj_tiled {
  int32_t n_slices;          /* S                                                            */
  int32_t slice_agents;      /* SA (multiple of 64, <= 19840: one slice of 8-byte sums + two flag bits per agent fits LDS) */
  int32_t direct_table_floats; /* 0: the direct form of pass 2 stages as many venue values through LDS at once
                                as fit; > 0: at most this many (tests: forces several groups)   */
  int32_t n_work;            /* entries of `work`                                            */
  const int32_t* work;       /* device [2*n_work] (set, block) pairs, heaviest first         */
  float* agent_scratch;      /* device [n_agents] workspace or NULL.  Non-NULL: phase D hands its
                                per-agent sums to a separate full-occupancy epilogue launch  */
  int32_t presum_wgs;        /* workgroups (= partial tables) of the direct form of pass 1, see gj_tiled_set.presum;
                                0: no set uses it                                                 */
  int32_t _pad_presum;
  gj_tiled_set sets[GJ_MAX_SETS];
} gj_tiled;

/* The compiled, immutable contact graph ("plan").  Host struct; arrays it points to are device
 * memory owned by the caller.                                                              */
typedef struct gj_plan {
  int64_t n_agents;           /* agents OWNED by this rank (all per-agent outputs)           */
  int64_t n_ext_agents;       /* owned + halo agents: length of transmission, q_transmission
                                 and agent_class, the arrays pass 1 gathers from (== n_agents
                                 on one GPU); v_agent indexes this extended range            */
  int32_t n_sets;
  int32_t n_blocks;           /* entries of `blocks`                                        */
  int32_t n_long_rows;        /* entries of `long_rows`                                     */
  int32_t n_partial_slots;    /* slots of `partial`                                         */
  gj_edge_set sets[GJ_MAX_SETS];
  const gj_block* blocks;     /* device [n_blocks]                                          */
  const gj_long_row* long_rows; /* device [n_long_rows]                                     */
  float* partial;             /* device [n_partial_slots * GJ_MAX_NETS_PER_SET] workspace   */
  const uint8_t* agent_class; /* device [n_ext_agents]  sex*100 + age  (0..199).  4-byte aligned and
                                 readable up to the next multiple of 4 agents when a leisure set is
                                 in the direct form (gj_tiled_set.ell): four classes are one load   */
  const float* tables;        /* device [n_tables * GJ_TABLE_SIZE] leisure tables           */
  int32_t n_tables;
  int32_t _pad;
  const gj_tiled* tiled;      /* HOST pointer or NULL.  Non-NULL: the passes run on the tiled
                                 layout and the CSR arrays of `sets` (v_rowptr, v_agent, a_rowptr,
                                 a_venue, blocks, long_rows) may be NULL                      */
} gj_plan;

/* One infection network active in this step
 * (reference: one InfectionNetwork module, base.py:11-87 / leisure_network.py:7-120).      */
typedef struct gj_network {
  float beta;          /* 10**log_beta * active SocialDistancing factors, fp32 (base.py:36-42) */
  int32_t set;         /* index into gj_plan.sets                                             */
  int32_t mask_kind;   /* gj_mask_kind                                                        */
  int32_t table;       /* index into gj_plan.tables, or -1                                    */
} gj_network;

/* The two scalars that change from one timestep to the next, in DEVICE memory: with gj_step_params.clock set, the
 * kernels read `now` and `step` from there instead of from the launch arguments, so a step captured once in a
 * hipGraph (kernels + the multi-GPU collectives) can be replayed for every following timestep of the same kind
 * (same networks, betas, duration) - gj_clock_advance is the graph's first node.                              */
typedef struct gj_clock {
  float now;      /* timer.now, days: what the kernels read                                        */
  float _pad;
  uint64_t step;  /* Philox stream id: timestep counter                                            */
  double now0;    /* (now0, step0): the origin the host set; gj_clock_advance derives               */
  uint64_t step0; /* now = (float)(now0 + (step - step0) * delta) - no accumulated rounding, the    */
                  /* same value an eager run computes on the host for that step (timer.py:92-95)    */
} gj_clock;

/* Scalars of one timestep.  `nets` MUST be in the reference's accumulation order
 * (activity hierarchy, grad_june/timer.py:14-26,139-157) with the networks of one edge set
 * adjacent; the kernels add the per-network terms in exactly this order.                   */
typedef struct gj_step_params {
  float now;            /* timer.now, days (timer.py:92-95)                                 */
  float delta_time;     /* timer.duration, days (timer.py:101-103)                          */
  int32_t day_type;     /* 0 weekday, 1 weekend (timer.py:84-90)                            */
  int32_t has_quarantine; /* 0: no quarantine-policy collection (mask is scalar 1.0)        */
  float q_threshold;    /* min stage_threshold over ACTIVE quarantine policies, +inf if none */
  int32_t n_nets;
  uint64_t seed;        /* Philox key (perf mode)                                           */
  uint64_t step;        /* Philox stream id: timestep counter                               */
  int64_t agent_offset; /* global id of local agent 0 (multi-GPU: partition-invariant RNG)  */
  int32_t transpose;    /* 0: forward.  1 (tiled layout, backward pass): the two passes run with the
                           pass-1 / pass-2 per-network weights exchanged - the adjoint of the
                           aggregation w.r.t. the transmissions (row f3)                      */
  int32_t _pad;
  gj_network nets[GJ_MAX_NETS];
  const gj_clock* clock; /* device pointer or NULL.  Non-NULL: `now` and `step` above are ignored, the kernels read
                            them from *clock (see gj_clock)                                                 */
} gj_step_params;

/* Per-agent state, all device fp32 [n_agents] unless noted.
 * (reference: data["agent"].*, grad_june/runner.py:72-90)                                  */
typedef struct gj_agent_state {
  const float* max_infectiousness; /* infection_parameters["max_infectiousness"]            */
  const float* shape;
  const float* rate;
  const float* shift;
  float* infection_time;
  float* is_infected;
  float* susceptibility;
  float* transmission;       /* out of a1; in of pass 1                                     */
  float* q_transmission;     /* workspace: qmask*transmission; may alias `transmission` when
                                has_quarantine == 0                                          */
  const float* current_stage; /* symptoms["current_stage"] as fp32 (NULL iff !has_quarantine) */
} gj_agent_state;

int gj_version(void);
const char* gj_error_string(int code);

/* GJ_OK when the CURRENT HIP device can run this library (a gfx950 / MI355X: the kernels are built for that
 * architecture only and size their workgroups for its 160 KiB of LDS per CU), GJ_E_NODEVICE otherwise.
 * The host mirror calls it once per process before the first launch (there is no other path to fall back to). */
int gj_check_device(void);

/* Which optional outputs / inputs the fused step uses. NULL = not wanted / not supplied.  */
typedef struct gj_step_io {
  float* not_infected_probs; /* out [n_agents] (a7), optional                               */
  float* new_infected;       /* out [n_agents] (a8), optional                               */
  const float* exp_noise;    /* in  [2*n_agents] Exponential(1) draws, row 0 = "not infected";
                                NULL = draw in-kernel with Philox4x32-10(seed, step, agent)  */
  float* trans_susc;         /* out [n_agents] pre-clamp sum over networks, optional (tests) */
  float* agent_sums;         /* out [n_agents] the same sum WITHOUT the agent's susceptibility factor (trans_susc =
                                susceptibility * agent_sums), optional: what a backward pass needs of the forward
                                (grad_june_amd/autograd.py keeps it instead of recomputing both passes).  Tiled
                                layout only: the CSR kernels multiply per term, as the reference does (GJ_E_PLAN) */
} gj_step_io;

/* a1 + a2: replaces TransmissionUpdater.forward (grad_june/transmission.py:38-51) and the
 * quarantine mask of QuarantinePolicies.apply (grad_june/policies/quarantine_policies.py:26-33).
 * Writes state->transmission and, when has_quarantine, state->q_transmission.              */
int gj_transmission_update(const gj_plan* plan, const gj_agent_state* state,
                           const gj_step_params* params, void* stream);

/* a2 alone: q_transmission[a] = (current_stage[a] < q_threshold) * transmission[a] for a
 * transmission vector the caller filled itself (the contract of InfectionNetworks.forward,
 * base.py:118-141, which reads data["agent"].transmission).  No-op when !has_quarantine.     */
int gj_quarantine_transmission(const gj_plan* plan, const gj_agent_state* state,
                               const gj_step_params* params, void* stream);

/* a3 + a4 + a5 (pass 1): replaces, for every active network, the first
 * `self.propagate(edge_index, x=transmissions, y=beta*p_contact)` of InfectionNetwork.forward
 * (grad_june/infection_networks/base.py:61-79).  Fills plan->sets[s].cum.                   */
int gj_venue_reduce(const gj_plan* plan, const gj_agent_state* state,
                    const gj_step_params* params, void* stream);

/* a4 + a6 + a7 (+ a8 + a9 when `sample` != 0): replaces the second propagate
 * (base.py:80-83), the sum over networks and clamp/exp/clamp of InfectionNetworks.forward
 * (base.py:118-141) and, fused, IsInfectedSampler.forward (grad_june/infection.py:13-18) and
 * GradJune.infect_people (grad_june/model.py:90-110).                                       */
int gj_agent_gather(const gj_plan* plan, const gj_agent_state* state,
                    const gj_step_params* params, const gj_step_io* io, int sample,
                    void* stream);

/* a8 + a9 alone on a given probability vector: replaces IsInfectedSampler.forward +
 * infect_people for callers that hold `not_infected_probs` (e.g. infect_fraction_of_people,
 * grad_june/infection.py:31-42).  The three state pointers may ALL be NULL: then only
 * `new_infected` is produced (IsInfectedSampler.forward alone).                             */
int gj_sample_infect(int64_t n_agents, const float* not_infected_probs, const float* exp_noise,
                     uint64_t seed, uint64_t step, int64_t agent_offset, float now,
                     float* new_infected, float* susceptibility, float* is_infected,
                     float* infection_time, void* stream);

/* ---- row f1 ("next"): disease-stage progression, run right after the hot path each step -------
 * replaces SymptomsUpdater.forward + SymptomsSampler.sample_next_stage
 * (grad_june/symptoms.py:204-247, 82-128) as one fused per-agent kernel.                        */
#define GJ_MAX_STAGES 16
typedef struct gj_symptoms_params {
  int32_t n_stages;                 /* len(symptoms.stages), <= GJ_MAX_STAGES (reference: 8)     */
  int32_t _pad;
  const float* progress;            /* device [n_stages*100] P(progress | stage, age) (symptoms.py:39-51) */
  int32_t next_kind[GJ_MAX_STAGES]; /* dwell time before progressing: 0 none, 1 LogNormal, 2 Normal */
  float next_loc[GJ_MAX_STAGES];
  float next_scale[GJ_MAX_STAGES];
  int32_t rec_kind[GJ_MAX_STAGES];  /* dwell time before recovering                              */
  float rec_loc[GJ_MAX_STAGES];
  float rec_scale[GJ_MAX_STAGES];
  float time;                       /* timer.now                                                 */
  float _pad2;
  uint64_t seed, step;              /* Philox key / stream (when no noise is injected)           */
  int64_t agent_offset;
} gj_symptoms_params;

/* current_stage / next_stage / time_to_next_stage: device fp32 [n], updated in place.
 * progresses / dwell: optional injected randomness (device fp32 [n]): the torch.bernoulli outcome
 * and the stage-time sample each agent consumes; NULL = Philox4x32-10(seed, step, agent).       */
int gj_symptoms_update(int64_t n, const uint8_t* agent_class, const float* new_infected,
                       float* current_stage, float* next_stage, float* time_to_next_stage,
                       const gj_symptoms_params* params, const float* progresses, const float* dwell,
                       void* stream);

/* ---- row f2 ("next"): the Runner's per-step result reductions in one pass -----------------------
 * replaces, per timestep, `is_infected.sum()` (grad_june/runner.py:167), get_cases_by_age
 * (runner.py:217-224: one masked sum per age bin, OPEN intervals lo < age < hi) and the deaths
 * count of store_differentiable_deaths (runner.py:198-215).
 * out: device double [2 + n_bins], ZEROED by the caller:  out[0] = sum is_infected,
 * out[1 .. n_bins] = sum is_infected over each age bin, out[1 + n_bins] = #agents at `dead_stage`.
 * Accumulated in fp64 (exact for these integer-valued sums), so the result is order-independent. */
#define GJ_MAX_AGE_BINS 8
int gj_step_stats(int64_t n, const uint8_t* agent_class, const float* is_infected,
                  const float* current_stage, int32_t n_bins, const int32_t* bin_edges /* host [n_bins+1] */,
                  int32_t dead_stage, double* out, void* stream);

/* ---- rows f1 + f2 in ONE pass: gj_symptoms_update followed by gj_step_stats of the updated state -----------------
 * What the reference's time loop does after the hot path of every step (grad_june/model.py:141 then
 * runner.py:167-171): the stage progression, then the result reductions over the post-update is_infected /
 * current_stage.  One kernel, four agents per lane: the three symptom arrays are read once and written back only
 * where an agent's values changed, the reductions are taken from the registers that hold the updated stages.
 * Arguments as for the two calls it replaces; `out` (device double [2 + n_bins]) must be ZEROED by the caller.      */
int gj_symptoms_step_stats(int64_t n, const uint8_t* agent_class, const float* new_infected,
                           float* current_stage, float* next_stage, float* time_to_next_stage,
                           const gj_symptoms_params* params, const float* progresses, const float* dwell,
                           const float* is_infected, int32_t n_bins, const int32_t* bin_edges /* host [n_bins+1] */,
                           int32_t dead_stage, double* out, void* stream);

/* ---- row f3: adjoint (backward) of one hot-path step, forward-only kernels reused ---------------
 * The aggregation ts = susc * sum_n w_n * (M_n^T diag(beta_n p_contact) M_n)(m_n * transmission) is
 * self-adjoint up to the exchange of the masks m_n <-> w_n, so its backward is the same four tiled
 * phases run with gj_step_params.transpose = 1 on the vector susc0 * ts_bar.  What remains is
 * elementwise:
 * gj_adjoint_sample: through infect_people (model.py:103-110), the straight-through Gumbel-softmax
 *   (infection.py:13-18) and clamp/exp/clamp (base.py:136-140).  Inputs: the step's pre-state, `acc`
 *   (= pre-susceptibility sum, i.e. gj_agent_gather's trans_susc run with susceptibility == 1), the
 *   step's noise (or Philox key) and the gradients w.r.t. the step's outputs (NULL = zeros).
 *   Outputs: x_out = susc0 * ts_bar (input of the transposed passes), grad_susc_out (complete),
 *   grad_time_out = g_time * (1 - new_infected).
 * gj_adjoint_transmission: through the transmission profile (transmission.py:39-51):
 *   grad_inf_out = g_inf + trans_bar * d trans/d is_infected;
 *   grad_time_inout += trans_bar * d trans/d infection_time.                                        */
int gj_adjoint_sample(int64_t n, const float* susceptibility0, const float* infection_time0, const float* acc,
                      const float* exp_noise, uint64_t seed, uint64_t step, int64_t agent_offset, float now,
                      float delta_time, const float* g_susc, const float* g_inf, const float* g_time,
                      const float* g_new, float* x_out, float* grad_susc_out, float* grad_time_out,
                      void* stream);
int gj_adjoint_transmission(int64_t n, const gj_agent_state* state0, float now, const float* trans_bar,
                            const float* g_inf, float* grad_inf_out, float* grad_time_inout, void* stream);

/* d loss / d log_beta of the networks on ONE edge set, from the forward's and the transposed passes' per-venue sums:
 *   col0 + k :  ln(10) * scale * sum_v [p_contact[v] > 0]  cum_fwd[v][k] * cum_bwd[v][k] / (beta[k] * p_contact[v]) * weight[v]
 * (reference: autograd through base.py:36-42,78-83; `weights` - fp64 [n_venues] or NULL = 1 - is a rank's share of
 * each venue in a multi-GPU run).  Two launches, deterministic: gj_adjoint_beta_partial ADDS every workgroup's fp64
 * partial sums into partial[GJ_ADJ_BETA_BLOCKS][GJ_MAX_NETS] (zeroed by the caller before a step's first set; a twin
 * network on the second half of a split set adds into its network's column), gj_adjoint_beta_finish sums the rows in
 * order into out[0 .. n_cols) and applies ln(10) * *scale (a device float: the power of two the cotangent was
 * normalised by).  beta[k] == 0 contributes 0.                                                                   */
#define GJ_ADJ_BETA_BLOCKS 256
int gj_adjoint_beta_partial(int64_t n_venues, int32_t stride, int32_t nk, const float* cum_fwd, const float* cum_bwd,
                            const float* v_pcontact, const double* weights, const float* beta /* host [nk] */,
                            const int32_t* cols /* host [nk] */, double* partial, void* stream);
int gj_adjoint_beta_finish(int32_t n_cols, const double* partial, const float* scale, double* out, void* stream);

/* gj_adjoint_symptoms: adjoint of gj_symptoms_update - what makes a loss on the symptom stages (the
 * deaths series, grad_june/runner.py:198-215; test/unit/test_runner.py:82-90 and
 * test_symptoms.py:208-231 assert the gradient exists) differentiable w.r.t. log_beta.  Inputs: the
 * PRE-update current / next stage and time_to_next_stage, new_infected, the SAME params (time, seed,
 * step) or the same injected `progresses` + `dwell`, and the gradients w.r.t. the updated current /
 * next stage / time (each may be NULL = 0).  Outputs ([n] each; g_time_in may be NULL): gradients
 * w.r.t. the incoming current stage, next stage, time_to_next_stage and w.r.t. new_infected.  The
 * gradient paths are those of symptoms.py:98,105-124,231-236: next += new*(2-next), time +=
 * new*(now-time), current -= (current-next)*mask, and the value-1 factor (current==i)*current/i on
 * the +1 / -next updates of next_stage and on the dwell time added to time_to_next_stage.          */
int gj_adjoint_symptoms(int64_t n, const uint8_t* agent_class, const float* new_infected,
                        const float* current_stage0, const float* next_stage0, const float* time_to_next_stage0,
                        const gj_symptoms_params* params, const float* progresses, const float* dwell,
                        const float* g_current, const float* g_next, const float* g_time, float* g_current_in,
                        float* g_next_in, float* g_time_in, float* g_new_infected, void* stream);

/* The production step: a1..a9 = the middle of GradJune.forward (grad_june/model.py:125-138)
 * as three dependent launches on `stream` (+1 when the plan has long rows).                */
int gj_step(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
            const gj_step_io* io, void* stream);

/* One launch group of gj_step at a time, for per-kernel timing (bench.py brackets each with HIP
 * events): phase 0 = transmission; 1 = pass-1 front (tiled: phase A scatter; CSR: venue reduce);
 * 2 = tiled phases B+C (CSR: nothing); 3 = pass-2 + epilogue + decision + state update.
 * Calling phases 0,1,2,3 in order is exactly gj_step.  Diagnostic variants: 4 = phase 3 without
 * the decision/state update (probabilities only); 5 / 6 = the tiled venue launch split into its
 * B half (sums -> cum) and its C half (cum -> per-edge values) - the multi-GPU step all-reduces
 * cum between the two; 7 = phases 1 then 5, 8 = phases 1 then 2, in one call.                  */
int gj_step_phase(const gj_plan* plan, const gj_agent_state* state, const gj_step_params* params,
                  const gj_step_io* io, int phase, void* stream);

/* clock->step += 1; clock->now = (float)(now0 + (step - step0) * delta_now), in double: a step length that is not a
 * power of two (hours / 24) does not drift over the replays.  One tiny launch on `stream` (the first node of a
 * captured step).                                                                                                 */
int gj_clock_advance(gj_clock* clock, double delta_now, void* stream);

/* Multi-GPU halo exchange helpers (one process per GPU; the all-to-all itself is issued by
 * the host through torch.distributed/RCCL between the two calls).
 * pack:   out[i] = src[index[i]]            for i < n      (send buffer of owned agents)
 * unpack: dst[index[i]] = in[i]             for i < n      (halo slots of remote agents)   */
int gj_pack_f32(int64_t n, const int32_t* index, const float* src, float* out, void* stream);
int gj_unpack_f32(int64_t n, const int32_t* index, const float* in, float* dst, void* stream);

/* ================= graph compile on the device ("native graph compile") =================
 * The tiled layout of ONE edge set (the arrays of gj_tiled_set) built in HBM from the reference's data format:
 * the unsorted COO `data["attends_<set>"].edge_index` (int64 [2, E]; grad_june/ has no compile step of its own -
 * torch_geometric walks this COO on every propagate, infection_networks/base.py:78-80).  Same arrays, bit for bit, as
 * the numpy specification grad_june_amd/tiling.py (build_tiled, wide_descriptors, build_ell), which documents them.
 * Kernels: key construction, rocPRIM radix sort / scans (hipcub), scatter of the 16-bit local indices, chunk
 * descriptors.  The library allocates nothing: the caller passes every output at its upper-bound size and one
 * workspace (gj_compile_workspace_bytes), reads `counts` (device int32[GJ_COMPILE_COUNTS]) after a stage and trims.
 *   1. gj_compile_blocks        venue degrees, range checks, the venue blocks         -> counts[GJ_CC_BLOCKS] = J
 *   2. gj_compile_tiles         everything else, with 4-word chunk descriptors        -> SLOTS, CHUNKS, MULTI
 *   3. gj_compile_wide_descriptors   (optional) the 8-word descriptor format for sets with small tiles
 *   4. gj_compile_ell_degrees / gj_compile_ell    the ELL rows of the direct form of pass 2
 * counts[GJ_CC_ERROR] != 0 after a stage: 1 agent index out of range, 2 venue index out of range, 3 more venue
 * blocks than blk_cap, 4 wide descriptor field overflow, 5 an owned agent has more edges than ell_k columns (the
 * entry is skipped, nothing is written outside the table).                                                        */
#define GJ_COMPILE_COUNTS 16
#define GJ_CC_BLOCKS 0       /* J: venue blocks                                                        */
#define GJ_CC_SLOTS 1        /* length of the block-major arrays (every block padded to 8 slots)       */
#define GJ_CC_CHUNKS 2       /* 64-edge chunks of the slice-major order                                */
#define GJ_CC_MULTI 3        /* chunks that span more than two tiles (share > 5 %: use wide descriptors; else rows, gj_compile_multi_slots) */
#define GJ_CC_OWNED_EDGES 4  /* edges whose agent is owned (< n_agents)                                */
#define GJ_CC_MAX_DEGREE 5   /* largest number of edges of one owned agent                             */
#define GJ_CC_ERROR 7
#define GJ_CC_RUN_PRIMARY 8   /* gj_compile_runs_pick: owned agents with an edge = primary edges               */
#define GJ_CC_RUN_UNSORTED 9  /* != 0: the owned agents are NOT ordered by their smallest venue - no run form  */
#define GJ_CC_RUN_WINDOW 10   /* venues in the widest slice window                                            */
#define GJ_CC_WIDE_MULTI 11   /* gj_compile_wide_descriptors: chunks that span more than six tiles (share > 2 %:
                                 use gj_compile_explicit_slots)                                             */

typedef struct gj_compile_set {
  const int64_t* agent;        /* [n_edges] edge_index[0]: agent ids, owned then halo, < n_ext_agents            */
  const int64_t* venue;        /* [n_edges] edge_index[1]: venue ids < n_venues                                  */
  const uint8_t* agent_class;  /* [owned + halo agents] or NULL; non-NULL: e_cls is written (leisure sets)        */
  int64_t n_edges;             /* < 2^30                                                                         */
  int64_t n_agents;            /* owned agents (the ELL rows)                                                    */
  int64_t n_ext_agents;        /* owned + halo agents, <= n_slices * slice_agents                                */
  int32_t n_venues;
  int32_t n_slices, slice_agents;   /* S, SA <= 65536                                                            */
  int32_t sv_max, eb_target;        /* venues per block <= 65536, edges per block aimed for                      */
  int32_t n_blocks;                 /* J, as stage 1 returned it (stage 1 itself: ignored)                       */
} gj_compile_set;

typedef struct gj_compile_out {       /* device buffers at their upper-bound sizes, see gj_compile_capacity      */
  int32_t* blk_e0;       /* [J + 1]                                                                              */
  uint16_t* e_lv;        /* [slots_cap]   0xFFFF = pad                                                            */
  uint8_t* e_cls;        /* [slots_cap] or NULL                                                                   */
  uint16_t* a_la;        /* [n_edges]                                                                             */
  int32_t* tile_sptr;    /* [S * J + 1]                                                                           */
  int32_t* tile_jpos;    /* [S * J]                                                                               */
  int32_t* chunk_ptr;    /* [S + 1]                                                                               */
  int32_t* chunk_desc;   /* [chunks_cap * 4]                                                                      */
  int64_t slots_cap, chunks_cap;
} gj_compile_out;

/* Upper bounds for the caller's allocations: blocks (before stage 1), slots and chunks (J known or not).          */
int gj_compile_capacity(const gj_compile_set* set, int64_t* blk_cap, int64_t* slots_cap, int64_t* chunks_cap);
/* Bytes of workspace that suffice for every stage of this set (J = set->n_blocks, or its upper bound when 0).      */
int gj_compile_workspace_bytes(const gj_compile_set* set, int64_t* bytes);
int gj_compile_blocks(const gj_compile_set* set, int32_t* blk_v0 /* [blk_cap + 1] */, int32_t blk_cap,
                      int32_t* counts, void* workspace, int64_t workspace_bytes, void* stream);
int gj_compile_tiles(const gj_compile_set* set, const int32_t* blk_v0, const gj_compile_out* out, int32_t* counts,
                     void* workspace, int64_t workspace_bytes, void* stream);
int gj_compile_wide_descriptors(const gj_compile_set* set, const gj_compile_out* out, int32_t n_chunks,
                                int32_t* desc8 /* [n_chunks * 8] */, int32_t* counts, void* stream);
/* degree[a] (int32 [n_agents + 1], caller-owned) = edges of owned agent a; counts[OWNED_EDGES], counts[MAX_DEGREE]. */
int gj_compile_ell_degrees(const gj_compile_set* set, int32_t* degree, int32_t* counts, void* stream);
/* ell: uint16 [ell_k / 2 planes][rows][2] (gj_tiled_set.ell), rows = owned slices * slice_agents; `degree` as
 * returned by gj_compile_ell_degrees; ell_k >= counts[GJ_CC_MAX_DEGREE] (checked per entry: counts[GJ_CC_ERROR] = 5
 * when an agent does not fit; `counts` may be NULL).                                                               */
int gj_compile_ell(const gj_compile_set* set, int32_t ell_k, int64_t rows, const int32_t* degree, uint16_t* ell,
                   void* workspace, int64_t workspace_bytes, int32_t* counts, void* stream);

/* (optional, after gj_compile_tiles) the explicit-slot form of a set with tiles of a few edges: slots[i] (int32 [E]) = the
 * block-major slot of slice-major edge i; gj_tiled_set.desc_wide = 2 and chunk_desc = slots.                       */
int gj_compile_explicit_slots(const gj_compile_set* set, const gj_compile_out* out, int32_t* slots, void* stream);
/* (after gj_compile_tiles / gj_compile_wide_descriptors, for a set NOT in the explicit-slot form) the chunks a descriptor
 * cannot express - counts[GJ_CC_MULTI] of the 4-word form, counts[GJ_CC_WIDE_MULTI] of the 8-word form = `n_multi` - get
 * a row of 64 explicit slots each, in chunk order (gj_tiled_set.multi_slots, int32 [n_multi * 64], caller-owned; lanes
 * past a chunk's end: 0), and their descriptors the row number in the j0 field.  `desc` = the descriptors in use
 * ([n_chunks * 4] or, `wide` != 0, [n_chunks * 8]); workspace >= 2 * (n_chunks + 1) * 4 bytes + 64 KiB.
 * Specification: tiling.attach_multi_slots.  (ABI 6; before, such a chunk's lanes walked the tile tables.)              */
int gj_compile_multi_slots(const gj_compile_set* set, const gj_compile_out* out, int32_t n_chunks, int32_t wide,
                           int32_t* desc, int32_t n_multi, int32_t* multi_slots, int32_t* counts, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* ---- run form of the set that orders the agents (gj_tiled_set.run_*; specification: tiling.split_primary_runs /
 * finish_run_form).  Three steps around the compile of the set's remaining edges:
 *   gj_compile_runs_pick   vmin[a] (int32 [n_agents], 0x7FFFFFFF = no edge), pick[a] (COO position of the agent's
 *                          primary edge: the first edge to its smallest venue), keep[e] (uint8 [E]: 1 = the edge stays
 *                          in the tiled arrays), the slices' windows (int32 [owned slices] each) and
 *                          counts[RUN_PRIMARY / RUN_UNSORTED / RUN_WINDOW / OWNED_EDGES] - the caller decides;
 *   gj_compile_runs_rest   the kept edges, compacted in COO order (int64 [E - primary] each) - the input of
 *                          gj_compile_blocks / _tiles for this set;
 *   gj_compile_runs_index  once the venue blocks are known: run_pv_blk, run_pv_win (uint16 [rows]), run_blk_r0.     */
int gj_compile_runs_pick(const gj_compile_set* set, int32_t* vmin, int32_t* pick, uint8_t* keep, int32_t* win_lo,
                         int32_t* win_n, int32_t* counts, void* stream);
int gj_compile_runs_rest(const gj_compile_set* set, const uint8_t* keep, int64_t* agent_out, int64_t* venue_out,
                         void* workspace, int64_t workspace_bytes, int32_t* counts, void* stream);
int gj_compile_runs_index(const gj_compile_set* set, const int32_t* vmin, const int32_t* blk_v0, int32_t n_blocks,
                          int64_t rows, const int32_t* win_lo, uint16_t* pv_blk, uint16_t* pv_win, int32_t* blk_r0,
                          void* stream);

/* Average device time of the last-timed dominant kernel is measured by the caller with
 * hipEvents; these two helpers let a ctypes caller do that on the stream it launches on
 * (torch.cuda.Event only sees torch's current stream).                                      */
int gj_event_create(void** event);
int gj_event_record(void* event, void* stream);
int gj_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop`       */
int gj_event_destroy(void* event);

#ifdef __cplusplus
}
#endif
#endif /* GRADJUNE_HIP_H */
