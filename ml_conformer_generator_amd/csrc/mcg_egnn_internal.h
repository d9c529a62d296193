// Types shared by the translation units of the EGNN denoiser (EGNNDynamics.forward, reference egnn.py:472-513):
//   mcg_egnn_model.hip   weights: repacking into MFMA fragment order, upload, precision / option switches
//   mcg_plan_host.cpp    HOST half of a batch plan: offsets, the edge-row table, the unit tables (no HIP; `make asan`)
//   mcg_egnn_plan.hip    device half of a plan: uploads, workspace, molecule ranges / streams
//   mcg_edge_exact.hip   exact-fp32 fused edge-MLP kernels (four-tile / quarter-tile units, k_edge_ns)
//   mcg_edge_bf16.hip    bf16-operand and f32x6 split-operand edge kernels
//   mcg_egnn_api.hip     one denoiser call: embedding, 9 blocks (node GEMMs + edge kernels), output head, HIP graph
//
// Data layout in HBM (all fp32):
//   * nodes are COMPACT: molecule b owns rows node_off[b] .. node_off[b]+n_b-1 (padded
//     slots of the reference's [B,N] layout are never materialised - they contribute
//     exactly zero there, egnn.py:51,83,127,148);
//   * node features h[M_r][HP], HP = 432 = 27 MFMA column tiles (420 real + 12 zero);
//   * edges are never materialised.  The real directed edges (i != j) of all molecules
//     form one flat row list  row = row_off[b] + i*(n_b-1) + jj,  j = jj + (jj >= i);
//     a wave owns 16 consecutive rows and output columns of the edge MLP's second layer in registers, so the
//     gate dot product, the mask and the per-node sum over j happen in the epilogue without the message
//     tensor m_ij[E,420] (reference: 78 MB per layer at config 2) ever touching memory;
//   * first edge-MLP layer is factorised per node:
//        W1 [h_i | h_j | d2 | d0] + b1 = (Wa h_i + b1) + Wb h_j + wd*d2 + wd0*d0
//     Pab[M_r][2*HP] holds (Wa h + b1 | Wb h); the per-edge sum + SiLU is generated
//     straight into the MFMA A-operand registers.
#pragma once
#include "mcg_common.h"
#include "mcg_api_internal.h"
#include "mcg_plan_host.h"

#include <vector>

constexpr int MCG_H = 420;        // hidden_nf (conformer_generator.py:70)
constexpr int MCG_HP = 432;       // padded to 27 column tiles of 16
constexpr int MCG_PAB_BLOCKED_FLOATS = 2 * 14 * 32;      // per atom, blocked layer-1 input layout of the bf16 mode (896 >= 2 * HP)
constexpr int MCG_NT = 27;
constexpr int MCG_KSTEPS = MCG_H / 4;   // 105 MFMA k-steps
constexpr int MCG_IN_NF = 12;     // 8 classes + time + 3 context
constexpr float MCG_NORM = 100.0f;  // egnn.py:435
// one 16-k group of the edge kernels' B-pack in LDS: 7 x 1 KiB pieces per wave x 4 waves = 28 KiB (27 used)
constexpr int MCG_GROUP_LDS_FLOATS = 28 * 256;

struct EdgeArgs {
    const float* pab;       // [M_r][2*HP]
    const float* x;         // [M_r][4] current coordinates
    const float* x0;        // [M_r][4] coordinates at network input (d0)
    const float* wd;        // [HP] layer-1 weights of d2 (current squared distance)
    const float* wd0;       // [HP] layer-1 weights of d0 (initial squared distance)
    const float* Bp;        // packed second-layer weights (KSTEPS x NT x 64, + 1 KB pad)
    const float* b2;        // [HP]
    const float* wv;        // [HP] attention weights (GCL) or coordinate head w5 (equiv)
    float bv;               // attention bias (GCL)
    const int2* row_ij;     // [n_mtiles*16] (i, j | seg << 24) of every edge row: compact node ids + the row's
                            // segment inside its unit; (-1,-1) on the padded tail
    const int* wave_poff;   // prefix offsets of (unit, node) partial slots (per-unit partial sums: k_edge_ns, bf16 kernels)
    int n_mtiles; int n_waves;
    float* P;               // per-unit partial sums.  GCL: [n_pslots][HP];  equiv: [n_pslots][4]
    // workgroup-level sums (k_edge_lds): the 4 waves of a workgroup fold their per-tile sums in LDS and
    // write ONE row per atom and workgroup, already divided by 100:  U [n_uslots + 1][HP] (GCL) / [..][4] (equiv)
    const int4* wg_info;    // per workgroup: {first global slot, slots, ws0 of its 4 waves (8 bits each), nseg of its 4 waves
                            // (8 bits each)} - ONE 16-byte scalar load per workgroup
    float* U;
    int n_full_wg;          // workgroups [0, n_full_wg) take four tiles each (LDS-staged body), the rest ONE tile (quarter-tile body)
    const float* Bp4;       // the same second-layer weights as B-pack4 (mcg_gemm.h): 16 B per lane and 16-k group
    int pab_blocked;        // 64-row bf16 kernel: pab is [2 parts][14 k-blocks][8 pieces][M][4] (mcg_gemm.h c_blocked) instead of [M][864]
    int M;                  // atoms (rows of pab)
};

// launchers of the edge kernel families (each returns a hipError_t from hipGetLastError).  `t0` / `t1` (optional): events
// that receive the KERNEL's own begin / end timestamps (hipExtLaunchKernelGGL) - mcg_bench_edge_incall
hipError_t mcg_launch_edge_exact(const EdgeArgs& a, bool equiv, int n_units, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);   // k_edge_lds: four-tile + quarter-tile units
hipError_t mcg_launch_edge_ns(const EdgeArgs& a, bool equiv, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);                    // k_edge_ns: one workgroup per 16-row tile
hipError_t mcg_launch_edge_bf16_16(const EdgeArgs& a, bool equiv, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);               // k_edge_lds_bf16: 16 rows per wave
hipError_t mcg_launch_edge_w64(const EdgeArgs& a, bool equiv, bool x6, hipStream_t s, hipEvent_t t0 = nullptr, hipEvent_t t1 = nullptr);          // k_edge_bf16_w64: 64-row units, bf16 / f32x6

struct EdgeLayer {      // second layer + head of an edge MLP, and its factorised first layer
    float *pab_Bp = nullptr, *pab_bias = nullptr, *wd = nullptr, *wd0 = nullptr;
    float *w2_Bp = nullptr, *w2_Bp4 = nullptr, *b2 = nullptr, *wv = nullptr;
    float bv = 0.f;
    uint16_t *pab_Bp16 = nullptr, *w2_Bp16 = nullptr;     // bf16 operand packs
    uint16_t* w2_Bp16x3 = nullptr;                        // W2 as three bf16 parts, [k-block][part][nt][lane][8] (f32x6 mode)
    uint16_t* pab_Bp16x3 = nullptr;                       // first-layer weights, same three-part form
};
struct NodeLayer {
    float *w3_Bp = nullptr, *b3 = nullptr, *w4_Bp = nullptr, *b4 = nullptr;
    uint16_t *w3_Bp16 = nullptr, *w4_Bp16 = nullptr;
    uint16_t *w3_Bp16x3 = nullptr, *w4_Bp16x3 = nullptr;  // three-part packs for the f32x6 GEMM
};

struct mcg_egnn {
    uint64_t uid = 0;           // unique per created model (captured graphs are keyed by it, never by the host address:
                                // a destroyed model's address is routinely handed out again by the allocator)
    int n_blocks = 0;
    bool bf16 = false;          // MFMA operands rounded to bf16 (opt-in, mcg_egnn_set_precision)
    bool x6 = false;            // f32x6: edge second layer as six bf16 partial products of three-part operands (fp32-accurate)
    // options (mcg_egnn_set_option)
    bool x6_gemm = true;        // f32x6 mode: node-side GEMMs on the split-operand kernel too
    int gemm_rn = 0, gemm_x6_rn = 0;   // wave tile width of the node GEMMs (0 = the launcher's cost model)
    int gemm_bf16_lds = 0;             // MCG_OPT_GEMM_BF16_LDS: 0 auto, 1 never, 2 whenever the shape allows
    int node_fused = 0;                // MCG_OPT_NODE_FUSED: 0 auto (from 32 row blocks on), 1 never, 2 whenever the bf16 gather path runs
    uint32_t opt_epoch = 0;     // bumped by mcg_egnn_set_precision / mcg_egnn_set_option: part of the captured graph's key,
                                // so a plan that already captured its launches re-captures after a change
    float *emb_wT = nullptr, *emb_b = nullptr, *out_w = nullptr, *out_b = nullptr;
    std::vector<EdgeLayer> gcl_edge;   // 2 per block
    std::vector<NodeLayer> gcl_node;   // 2 per block
    std::vector<EdgeLayer> equiv;      // 1 per block
    std::vector<void*> allocs;
};

struct mcg_plan {
    int B = 0, N = 0, M = 0, n_rows = 0, n_mtiles = 0, MT = 1, n_waves = 0, n_pslots = 0;
    int2* row_ij = nullptr;
    int *n_nodes = nullptr, *node_off = nullptr, *wave_poff = nullptr, *node_mol = nullptr,
        *node_slots = nullptr;              // node_slots: [M][8] per-unit partial slots of every atom
    float *x = nullptr, *x0 = nullptr, *h = nullptr, *h2 = nullptr, *pab = nullptr, *agg = nullptr, *t1 = nullptr,
          *P = nullptr, *Px = nullptr;
    // workgroup-level sums of the throughput edge kernel (MT = 1; see edge_epilogue_wg): one row per (workgroup, atom)
    bool wgc = false;                       // tables below are valid and every workgroup touches <= 16 atoms
    struct UnitTables {
        int n_units = 0, n_full_wg = 0;     // workgroups of the throughput kernel; the first n_full_wg take four tiles, the rest one
        int n_uslots = 0;                   // rows of U / Ux this set writes (the common zero row sits behind the larger set's)
        int4* wg_info = nullptr;
        int4* node_slots = nullptr;         // per atom: its one to four rows of U (unused = the zero row)
        int max_span = 2;                   // most rows of U any atom has (3 / 4 only with quarter-tile units)
    };
    UnitTables ut[2];                       // [0]: the automatic split into four-tile and quarter-tile units,
    bool have_alt = false;                  // [1]: four-tile units only (latency_mode 0), built when it differs from [0]
    const UnitTables& units() const { return ut[(latency_mode == 0 && have_alt) ? 1 : 0]; }
    float *U = nullptr, *Ux = nullptr;
    // per-unit partial sums of the 64-row / 16-row bf16 and the column-split kernels (P / Px, one row per unit and atom, NOT
    // divided by 100): an atom's first four slots as one int4 (unused = the zero row n_pslots), so that the bf16 node GEMM
    // can gather them like the fp32 one gathers U and the coordinate update can ride along as a side job
    int4* pslots4 = nullptr;
    int pspan = 0;                          // most slots any atom has; the gathers need <= 4
    bool x_pending = false;                 // host-side: a coordinate update sits in pending_u / pending_slots, not yet applied to x
    bool pab_ready = false;                 // host-side: the next edge layer's first-layer projections are already in `pab` (fused node launch)
    const float* pending_u = nullptr;       // Ux (workgroup-level sums) or Px (per-unit partial sums)
    const int4* pending_slots = nullptr;
    std::vector<void*> allocs;              // blocks of the plan pool (mcg_dev_alloc)
    bool is_sub = false;                    // a molecule range of another plan
    // destruction: the plan's blocks go back to a pool, so everything launched on them must have finished - every entry
    // point that enqueues work on this plan records `ev_done` on the caller's stream behind it (mcg_plan_mark; the molecule
    // ranges' streams have joined that stream by then) and mcg_plan_destroy waits for THAT event on the plan's OWN device,
    // instead of stalling the whole (current) device
    int dev = -1;                           // device the plan lives on (current device of mcg_plan_create)
    hipEvent_t ev_done = nullptr;
    mutable bool ev_pending = false;
    // optional split into independent molecule ranges that run on separate HIP streams
    // (the latency-bound node GEMMs of one range overlap the edge kernels of the other)
    std::vector<mcg_plan*> subs;
    std::vector<int> sub_b0;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> ev_join;
    hipEvent_t ev_fork = nullptr;
    // the whole denoiser call (~95 launches) captured once as a HIP graph and replayed: the host then
    // issues one graph launch per call instead of ~95 kernel launches
    float* t_buf = nullptr;                 // fixed device copy of t[B] read by the captured graph
    hipStream_t cap_stream = nullptr;       // capture happens here (the caller's stream may be the null stream)
    hipGraphExec_t graph_exec = nullptr;
    std::vector<hipGraphExec_t> retired_graphs;     // executables replaced by a re-capture: destroyed with the plan (mcg_egnn_api.hip)
    const void* g_key[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // xh, context, out, model uid, precision mode | option epoch << 8
    int graph_failed = 0;
    // Callers whose tensors move between calls (the reference's own sampler loop allocates a fresh xh / out
    // every step) would force a re-capture per call: after the second key change the graph is captured on
    // plan-owned staging buffers instead and each call adds three small device-to-device copies around it.
    int key_changes = 0;
    float *xh_stage = nullptr, *ctx_stage = nullptr, *out_stage = nullptr;
    int latency_mode = -1;                  // -1 auto; 0: four-tile units only; 1: the stand-alone column-split kernel (k_edge_ns)
    // mcg_bench_edge_incall: when set, every edge launch of a (plain, un-captured) denoiser call is bracketed by a pair
    // of events that receive the kernel's own begin / end timestamps
    struct EdgeTiming { hipEvent_t t0, t1; bool equiv, used; };
    std::vector<EdgeTiming>* edge_timing = nullptr;       // pool of pre-created event pairs (no API call between launches)
    size_t* edge_timing_next = nullptr;                   // shared cursor into the pool
};

// records "everything enqueued on this plan so far" on the caller's stream (see mcg_plan::ev_done)
inline void mcg_plan_mark(const mcg_plan* pl, hipStream_t s) {
    if (pl && pl->ev_done && hipEventRecord(pl->ev_done, s) == hipSuccess) pl->ev_pending = true;
    else (void)hipGetLastError();
}

struct mcg_plan_mark_guard {              // marks at scope exit, whatever the return path
    const mcg_plan* p; hipStream_t s;
    ~mcg_plan_mark_guard() { mcg_plan_mark(p, s); }
};

// device-memory pool of the plans (mcg_devmem.hip): blocks of destroyed plans are reused by new ones
int mcg_dev_alloc(size_t bytes, void** out);
void mcg_dev_free(void* p);
void mcg_dev_trim();

// small device-memory helpers shared by the model and plan builders
int mcg_upload_f(const std::vector<float>& v, float** d);
int mcg_upload_u16(const std::vector<uint16_t>& v, uint16_t** d);
int mcg_upload_i(const std::vector<int>& v, int** d);
