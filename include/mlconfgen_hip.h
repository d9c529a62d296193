/* libmlconfgen_hip.so - C ABI of the MI355X (gfx950) denoising hot path of mlconfgen.
 *
 * The reference (Membrizard/ml_conformer_generator @ 2025-07-04) is pure Python/PyTorch and has no
 * FFI; its operator boundary is the pair of module-call seams that its own ONNX export treats as
 * operators.  Each entry point below names the reference interface it
 * replaces.  Plain pointers and sizes only; every float buffer is fp32, row-major, contiguous and
 * lives in DEVICE memory unless the name ends in `_host`.  `stream` is a hipStream_t (0 = default).
 * All functions return 0 on success (MCG_OK) or a non-zero code; mcg_last_error() gives the text.
 * No global mutable state except the opaque handles.  Behaviour is chosen through arguments: `mcg_plan_opts`
 * (per plan), `mcg_egnn_set_precision` / `mcg_egnn_set_option` (per model), `mcg_plan_set_latency_mode`.  Two PROCESS
 * switches remain, read from the environment, neither needed for normal operation:
 *   MCG_GRAPH=0     plain launches instead of the captured HIP graph per denoiser call (read once, on first use)
 *   MCG_VERBOSE=1   graph capture diagnostics on stderr
 * (Python side: MCG_LIB_PATH = alternative library file, MCG_FORCE_COLLECTIVE=1 = run the final gather on a 1-rank
 * group.)  Threading: like the reference (single Python thread, one stream) - a handle carries workspace, so one
 * mcg_plan / mcg_gcn must not run on two host threads or two streams at once; different handles are independent,
 * mcg_egnn weights are read-only after creation (mcg_egnn_set_precision / mcg_egnn_set_option excepted) and may be
 * shared by several plans.  mcg_last_error() is thread-local.
 */
#ifndef MLCONFGEN_HIP_H
#define MLCONFGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mcg_egnn mcg_egnn;   /* packed EGNN weights (one per checkpoint)            */
typedef struct mcg_plan mcg_plan;   /* batch geometry + workspace (one per batch of sizes)  */
typedef struct mcg_gcn mcg_gcn;     /* packed AdjMatSeer weights + workspace                */

const char* mcg_last_error(void);
int mcg_abi_version(void);

/* ---- EGNN weights.  Replaces `EquivariantDiffusion.load_state_dict(...)` of the dynamics
 * (conformer_generator.py:90-95).  `tensors_host` are HOST pointers to the reference-layout
 * fp32 tensors ([out,in] row-major, as nn.Linear stores them) in this order:
 *   embedding.weight, embedding.bias, embedding_out.weight, embedding_out.bias, then for each
 *   block k: gcl_0{edge_mlp.0.w, .0.b, edge_mlp.2.w, .2.b, node_mlp.0.w, .0.b, node_mlp.2.w, .2.b,
 *   att_mlp.0.w, att_mlp.0.b}, gcl_1{same 10}, gcl_equiv{coord_mlp.0.w, .0.b, .2.w, .2.b, .4.w}
 * (n_tensors = 4 + 25*n_blocks; hidden must be 420).  The library repacks them into MFMA
 * fragment order and uploads them. */
int mcg_egnn_create(const float* const* tensors_host, int n_tensors, int hidden, int n_blocks, mcg_egnn** out);
void mcg_egnn_destroy(mcg_egnn* m);
/* mode 0 (default): exact fp32 MFMA everywhere.
 * mode 1 "bf16":  the MFMA operands of the EGNN contractions are rounded to bf16 (fp32 accumulate, fp32
 *                 coordinates / aggregates / epilogues) - BASELINE.json configs[4].
 * mode 2 "f32x6": fp32-accurate edge-MLP contraction on the bf16 matrix pipe: each fp32 operand is carried as the
 *                 exact sum of three bf16 parts and the six partial products of weight >= 2^-16 are accumulated
 *                 in fp32 (dropped terms <= 2^-23 relative); everything else as mode 0.  Uses edge_mt = 4 plans
 *                 (other plans run the exact kernels). */
int mcg_egnn_set_precision(mcg_egnn* m, int mode);
/* Model-level measurement options (defaults in brackets; none is needed for normal operation):
 *   MCG_OPT_X6_GEMM     [1]  f32x6 mode: 1 = node-side GEMMs on the split-operand kernel too, 0 = exact fp32 GEMMs
 *   MCG_OPT_GEMM_RN     [0]  wave tile width (column tiles) of the 32-row node GEMM kernel: 0 = cost model, 1..3
 *   MCG_OPT_GEMM_X6_RN  [0]  same for the split-operand GEMM kernel
 *   MCG_OPT_GEMM_BF16_LDS [0] bf16 mode: 0 = the LDS-staged 9-wave node GEMM from 80 row blocks (2 560 atoms) on, 1 = never (32-row kernel),
 *                            2 = whenever its shape limits allow (results are bit-identical either way)
 *   MCG_OPT_NODE_FUSED  [0]  bf16 mode: the node phase of a GCL layer (node MLP + the next edge layer's first-layer projections) as ONE
 *                            launch: 0 = from 32 row blocks (1 024 atoms) on, 1 = never (three launches),
 *                            2 = whenever the bf16 gather path runs (results are bit-identical either way) */
enum { MCG_OPT_X6_GEMM = 1, MCG_OPT_GEMM_RN = 2, MCG_OPT_GEMM_X6_RN = 3, MCG_OPT_GEMM_BF16_LDS = 4, MCG_OPT_NODE_FUSED = 5 };
/* Both setters may be called at any time between denoiser calls: a plan that has already captured its launches as a HIP
 * graph re-captures on its next call (the graph is keyed by the model's option epoch). */
int mcg_egnn_set_option(mcg_egnn* m, int option, int value);

/* Measurement / test hook: node and GCN GEMM launches ISSUED (plainly or into a graph capture) by this process since the
 * last reset, counts_host[family * 8 + rn] with family 0 = 32-row fp32 kernel, 1 = 32-row bf16 kernel, 2 = 16-row-tile fp32
 * kernel, 3 = split-operand kernel and rn = column tiles per wave (family 1: slot 7 = the LDS-staged 9-wave bf16 kernel, slot 6 = the
 * fused node launch of MCG_OPT_NODE_FUSED) - how a
 * test sees that an option changed launch shapes. */
int mcg_debug_gemm_launches(int64_t* counts_host, int reset);

/* ---- Batch plan.  Replaces the per-call `get_adj_matrix` edge-list rebuild (egnn.py:475,515-541)
 * and the node/edge masks (mol_utils.py:226-252): node_mask[b] is the prefix of n_nodes_host[b]
 * ones, edge_mask = outer product minus diagonal.  edge_mt: 0 = auto, 1 = 16-row edge tiles (exact-fp32 kernels,
 * 16-row bf16 kernel), 4 = 64-row units (bf16 / f32x6 kernels; needs <= 16 atoms' rows per unit). */
int mcg_plan_create(int B, int N, const int32_t* n_nodes_host, int edge_mt, mcg_plan** out);
/* The same with every choice the library would make itself open to the caller (a zero-initialised struct = defaults;
 * `opts` may be NULL):
 *   edge_mt          as above
 *   n_ranges         independent molecule ranges, each running the whole denoiser on its own HIP stream inside
 *                    mcg_egnn_dynamics: 0 = the library's choice (16-row-tile plans: 1 / 2 / 3 / 4 ranges below 3 600 /
 *                    5 200 / 14 000 / from 14 000 edge tiles on; 64-row-unit plans: 2 from 8 192 tiles on), 1..4 given
 *   four_tile_units  workgroups of the exact-fp32 throughput edge kernel that take four 16-row tiles each (the rest take
 *                    ONE tile, its columns split over the 4 waves): 0 = every complete round of the chip (default),
 *                    -1 = none, n > 0 = the first n (rounded up to a multiple of 8, capped at all of them;
 *                    MCG_ALL_FOUR_TILE = all) */
typedef struct mcg_plan_opts {
    int32_t edge_mt;
    int32_t n_ranges;
    int32_t four_tile_units;
    int32_t reserved[5];          /* must be zero */
} mcg_plan_opts;
#define MCG_ALL_FOUR_TILE 0x3fffffff
int mcg_plan_create_ex(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, mcg_plan** out);
void mcg_plan_destroy(mcg_plan* p);
/* Plans take their device memory (one block of tables, one of workspace) from a per-device pool inside the library: the
 * blocks of a destroyed plan are handed to the next one instead of going back to the driver, so a caller that meets a new
 * size vector on every call (and evicts an old plan on every call) allocates nothing in steady state.
 * stats_host[4] (may be NULL) = {bytes in use by live plans, bytes cached for reuse, driver allocations so far, pool hits so
 * far} of the current device; trim != 0 first returns every cached block to the driver. */
int mcg_pool_stats(int64_t* stats_host, int trim);
/* Edge-kernel choice (exact-fp32 mode, edge_mt 1): -1 auto - the throughput kernel, whose workgroups take four 16-row
 * tiles each for every complete round of the chip and ONE tile each (columns split over the 4 waves) for the rest, i.e.
 * for the whole of a small batch; 0 = four-tile workgroups only; 1 = the stand-alone column-split kernel with per-wave
 * partial sums (the fallback for plans without workgroup-level tables). */
int mcg_plan_set_latency_mode(mcg_plan* p, int mode);
/* Host-only self-check of the tables mcg_plan_create_ex would build for a device with `cus` compute units (no GPU
 * call; same `opts`): every edge row's (unit, tile, segment) must land in a slot its
 * atom lists, no slot may be shared by two atoms, a four-tile unit parks <= 16 rows, the row table names the right
 * (i, j).  info[8] = {table sets, units, four-tile units, slots, most slots per atom of set 0; units, slots, most slots
 * per atom of set 1 (four-tile units only; zeros when there is one set)}.  Returns MCG_OK or an error with text. */
int mcg_plan_check_tables(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, int cus, int32_t* info_host);
/* info[8] = {real nodes, real edges, edge_mt, edge waves, partial slots, B, N, 16-row edge tiles} */
int mcg_plan_info(const mcg_plan* p, int32_t* info_host);
/* number of molecule ranges (HIP streams) the plan actually runs: 1 = unsplit (small batches are never split) */
int mcg_plan_ranges(const mcg_plan* p);

/* ---- Op seam 1: out[B,N,11] = EGNNDynamics.forward(t[B,1], xh[B,N,11], node_mask, edge_mask,
 * context[B,N,3])  (egnn.py:472-513; called from equivariant_diffusion.py:187).  Masks are those of
 * the plan.  Inputs are not modified; `out` may not alias `xh`. */
int mcg_egnn_dynamics(const mcg_egnn* m, mcg_plan* p, const float* t, const float* xh, const float* context,
                      float* out, void* stream);

/* Kernel-level pin: ONE EquivariantBlock (egnn.py:188-222) on compact node arrays
 * h_io[M,420], x_io[M,3], x0[M,3] (M = real nodes of the plan, molecule-major). */
int mcg_egnn_block_debug(const mcg_egnn* m, mcg_plan* p, int block, float* h_io, float* x_io, const float* x0,
                         void* stream);

/* Debug hooks: one GCL layer (egnn.py:70-85) on compact arrays (result stays in the plan), and a
 * copy-out of the plan's internal buffers: which = 0 h[M][432], 1 pab[M][864], 2 agg[M][432],
 * 3 node-MLP hidden[M][432], 4 x[M][4]. */
int mcg_egnn_gcl_debug(const mcg_egnn* m, mcg_plan* p, int layer, const float* h_in, const float* x_in,
                       const float* x0, void* stream);
int mcg_plan_peek(const mcg_plan* p, int which, float* dst, void* stream);

/* Measurement hook: the edge-MLP kernel of GCL layer `layer` (equiv = 0, 0..2*n_blocks-1) or of the
 * coordinate update of block `layer` (equiv = 1) launched `iters` times on the plan's current state. */
int mcg_bench_edge(const mcg_egnn* m, mcg_plan* p, int layer, int equiv, int iters, void* stream);
/* Measurement hook: `calls` whole denoiser calls (arguments as mcg_egnn_dynamics) issued as plain launches with every
 * edge-MLP launch bracketed by events that receive the kernel's own begin / end timestamps, i.e. the dominant kernel timed
 * in the context it runs in.  us_host[4] = {mean us of the 18 GCL edge launches per call, mean us of the 9 coordinate-layer
 * ones, how many of each were timed}.  Synchronises `stream`. */
int mcg_bench_edge_incall(const mcg_egnn* m, mcg_plan* p, const float* t, const float* xh, const float* context, float* out,
                          int calls, float* us_host, void* stream);

/* ---- Sampler arithmetic (equivariant_diffusion.py).  randn_x[B,N,3] / randn_h[B,N,8] are RAW
 * standard-normal draws (the caller draws them in the reference's order: x first, then h);
 * masking and centre-of-gravity removal (:56-76) happen inside. */
/* eps[B,N,11] = sample_combined_position_feature_noise (:341-363) */
int mcg_sampler_noise(const mcg_plan* p, const float* randn_x, const float* randn_h, float* eps, void* stream);
/* z <- sample_p_zs_given_zt (:295-339): one network call + ancestral update + mean removal.
 * alpha_ts, c_eps = sigma2_ts/alpha_ts/sigma_t, c_noise = sigma_ts*sigma_s/sigma_t are the fp32
 * scalars of :308-326; t_dev[B] holds t. */
int mcg_sampler_step(const mcg_egnn* m, mcg_plan* p, float* z, const float* context, const float* t_dev,
                     const float* randn_x, const float* randn_h, float alpha_ts, float c_eps, float c_noise,
                     float* eps_hat_scratch, void* stream);
/* x[B,N,3], h[B,N,8] (one-hot, float) = sample_p_xh_given_z0 (:261-285) */
int mcg_sampler_decode(const mcg_egnn* m, mcg_plan* p, const float* z0, const float* context, const float* t_zero_dev,
                       const float* randn_x, float inv_alpha0, float sigma0, float sigma_x, float norm_x, float norm_h,
                       float* eps_hat_scratch, float* x_out, float* h_out, void* stream);
/* mode 0: z = alpha_s*z_known + sigma_s*eps            (merge_fragments :548-559)
 * mode 1: re-noise z_known, align the fixed fragment's centre of mass (:79-105), blend (:489-493) */
int mcg_sampler_blend(const mcg_plan* p, float* z, const float* z_known, const float* fixed_mask, const float* randn_x,
                      const float* randn_h, float alpha_s, float sigma_s, float blend, int mode, void* stream);

/* ---- Stand-alone aggregate (unsorted_segment_sum of gate*m, egnn.py:49-64,418-437) over the compact
 * real-edge list: node v owns n_rows[v] consecutive rows of m[E_r, D] starting at first_row[v];
 * out[v,:] = sum_j gate[row]*m[row,:] / 100.  HBM-roofline probe; production fuses this. */
int mcg_egnn_aggregate(const float* m, const float* gate, const int32_t* first_row, const int32_t* n_rows, float* out,
                       int n_nodes, int D, void* stream);

/* ---- Op seam 2: AdjMatSeer (adj_mat_seer.py:104-165).  tensors_host order (22):
 *   gcn1..4.linear.{weight,bias}, resize.{weight,bias}, nodes_embedding.weight,
 *   nodes_coord_fc.{weight,bias}, gcn1_dm..gcn3_dm.linear.{weight,bias}, dm_resize.{weight,bias},
 *   dm_nodes_embedding.weight */
int mcg_gcn_create(const float* const* tensors_host, int n_tensors, mcg_gcn** out);
void mcg_gcn_destroy(mcg_gcn* g);
/* logits[B,42,42,5] = adj_mat_seer(elements[B,42] int64, dist_mat[B,42,42], adj_mat[B,42,42]);
 * bond[B,42,42] int8 = argmax over the 5 bond classes (the consumer's reduction, mol_utils.py:210).
 * Either output may be NULL. */
int mcg_gcn_forward(mcg_gcn* g, const int64_t* elements, const float* dist_mat, const float* adj_mat, float* logits,
                    int8_t* bond, int B, void* stream);
int mcg_gcn_check(mcg_gcn* g);
/* EDM -> GCN hand-off (replaces samples_to_rdkit_mol + prepare_adj_mat_seer_input, utils/mol_utils.py:18-57,146-194,
 * for the tensor part): elements[B,42] int64 (atomic numbers, 0 padded), dist_mat[B,42,42] (distances + I),
 * adj_mat[B,42,42] ({0,1} covalent-radius connectivity + I) from x[B,N,3], h[B,N,8] one-hot and n_nodes[B] (device
 * int32).  Atoms keep their generation order.  dist_mat is the reference's BIT FOR BIT: coordinates through the "%.9f" text
 * round trip (rint(x * 1e9) / 1e9, exact in fp64), fp64 differences / sum / sqrt in `distance_matrix`'s order, ONE rounding
 * to fp32 at the store (utils/mol_utils.py:46-51,129-143,168,176-187); the covalent rule compares in fp64 as well
 * (ABI 4: cov_factor is a double). */
int mcg_handoff(const float* x, const float* h, const int32_t* n_nodes_dev, int B, int N, double cov_factor,
                int64_t* elements, float* dist_mat, float* adj_mat, void* stream);
/* The same with the two RDKit-owned decisions of the reference open to the caller (`canonicalise`,
 * utils/mol_utils.py:110-126: DetermineConnectivity + `_smilesAtomOutputOrder` + RenumberAtoms; the MolGraph adjacency
 * of `prepare_adj_mat_seer_input`, :146-194).  AdjMatSeer is NOT permutation-equivariant (nodes_coord_fc is a dense layer
 * over the position index, adj_mat_seer.py:135-138), so a checkpoint trained on canonical-SMILES order needs that order:
 *   order[B,42] int32 (device; NULL = generation order): order[b][p] = generation index of the atom placed at
 *                position p - the argument of `Chem.RenumberAtoms(mol, order)`; the first n_nodes[b] entries of row b must
 *                be a permutation of 0..n_nodes[b]-1 (out-of-range entries are clamped and reported through
 *                bad_order_flag), the rest is ignored;
 *   conn_in[B,42,42] uint8 (device; NULL = covalent-radius rule d < cov_factor*(r_i + r_j)): symmetric {0,1}
 *                1-order connectivity in GENERATION order (what DetermineConnectivity perceives before renumbering);
 *                the diagonal is ignored, +I is added here; it is permuted with the atoms;
 *   x_out[B,N,3] (NULL = not wanted; may not alias x): the coordinates in the order of the GCN input, zero padded -
 *                the reference's `canonicalised_samples`, whose order the bond write-back and the returned
 *                molecules follow (conformer_generator.py:357-366);
 *   bad_order_flag (device int32, NULL = not wanted): set to 1 if an order entry was out of range; never cleared here.
 * elements / dist_mat / adj_mat come out in the permuted order. */
int mcg_handoff_ex(const float* x, const float* h, const int32_t* n_nodes_dev, int B, int N, double cov_factor,
                   const int32_t* order, const uint8_t* conn_in, int64_t* elements, float* dist_mat, float* adj_mat,
                   float* x_out, int32_t* bad_order_flag, void* stream);

/* Bond write-back + validity pre-filter behind the GCN (replaces the tensor half of `redefine_bonds`,
 * utils/mol_utils.py:197-223, and stands in for `standardize_mol(...) is not None`, conformer_generator.py:362-366 /
 * utils/standardizer.py:83-111, which is RDKit sanitisation + MMFF and cannot run without RDKit):
 * bond[B,42,42] int8 = argmax bond classes from mcg_gcn_forward; bond_sym[B,42,42] int8 = its strict lower triangle
 * mirrored, zero outside the molecule's n x n block; valid[B] uint8 = 1 when no atom exceeds its maximum valence and
 * the bond graph is one connected fragment (a labelled PROXY, not RDKit's gate). */
int mcg_bond_writeback(const int8_t* bond, const int64_t* elements, const int32_t* n_nodes_dev, int B, int8_t* bond_sym,
                       uint8_t* valid, void* stream);
/* Inertial fragment matching, the tensor work between its two sampler runs (`inverse_coord_transform`,
 * utils/mol_utils.py:508-524, then `ifm_prepare_fragments_for_merge`, :460-505) in one launch:
 * z_known[B,N,11] = [fixed fragment (ff_x[n_ff,3] | ff_h[n_ff,8]) ; gen_x[B,n_gen,3] @ rotation[B,3,3]^T - shift[B,3] |
 * gen_h[B,n_gen,8]], zero rows up to N; fixed_mask[B,N,1] = 1 on the first n_ff rows.  n_ff + n_gen <= N. */
int mcg_ifm_merge(const float* ff_x, const float* ff_h, int n_ff, const float* gen_x, const float* gen_h, int n_gen,
                  const float* shift, const float* rotation, int B, int N, float* z_known, float* fixed_mask, void* stream);

/* ---- Evaluation (grid half): Gaussian-volume shape Tanimoto of one reference against B
 * candidates in R orientations - `tanimoto_score` (cheminformatics/shape_similarity.py:468-492) for every
 * (candidate, rotation) pair of `evaluate_samples` (cheminformatics/pipeline.py:64-85).
 * ref[n_ref,3]; cand[B,N,3] with n_nodes[B] real atoms; rot[R,9] row-major matrices applied as coord @ M;
 * axes[3,n] = the three linspace axes of the n^3 grid (built by the caller exactly as the reference's
 * `Grid`); f_scratch[n^3]; score[B*R]. */
int mcg_shape_tanimoto(const float* ref, int n_ref, const float* cand, const int32_t* n_nodes_dev, int B, int N,
                       const float* rot, int R, const float* axes, int n, float alpha, float amplitude,
                       float* f_scratch, float* score, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MLCONFGEN_HIP_H */
