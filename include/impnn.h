/*
 * impnn.h - C ABI of libimpnn.so: the MI355X (gfx950) implementation of the ionic-mpnn
 * message-passing forward path.
 *
 * The reference (goalheart/ionic-mpnn) is pure Python on TensorFlow/Keras and has no FFI of
 * its own; the interface replaced here is the set of TF ops each Keras layer's call() issues.
 * Every entry point cites the reference lines (relative to the reference root) it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM), caller-allocated, row-major contiguous;
 *     float = IEEE fp32, ids / connectivity = int32 exactly as the reference's
 *     Input(dtype=tf.int32) tensors (train_viscosity.py:150-157);
 *   - B batch (one ion each), N padded atoms, E padded directed edge slots, D atom_dim,
 *     K bond_dim, S message-passing steps, Va/Vb vocabulary sizes including padding id 0;
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous enqueues that never
 *     allocate, free, synchronise or retain pointers past return (hipGraph-capturable);
 *   - return value: 0 = ok, negative = error (IMPNN_E_*); impnn_last_error_string() gives the
 *     thread-local text of the last failure.  No C++ exception crosses this boundary;
 *   - re-entrant: results depend on the arguments only.  The library's only mutable state is thread-local
 *     (last error text, the opt-in event profiler and debug stamp buffer of the calling thread) plus write-once
 *     per-device caches of device attributes;
 *   - indices are never trusted: an edge whose src or tgt is outside [0,N) is treated as a
 *     padding edge (contributes nothing), an embedding id outside [0,V) yields a zero row -
 *     the behaviour of TF's GPU gather/scatter kernels; tf-CPU's "raise" behaviour is
 *     available through impnn_validate_indices + the Python wrapper's debug mode.
 */
#ifndef IMPNN_H_
#define IMPNN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMPNN_OK 0
#define IMPNN_E_BADARG (-1)      /* null pointer, non-positive size, misaligned buffer */
#define IMPNN_E_UNSUPPORTED (-2) /* shape outside what the kernel set covers */
#define IMPNN_E_LAUNCH (-3)      /* hipLaunch / hipGetLastError failure */
#define IMPNN_E_WORKSPACE (-4)   /* workspace too small */

typedef void* impnn_stream_t; /* hipStream_t */

/* library identity: returns IMPNN_ABI_VERSION */
#define IMPNN_ABI_VERSION 3
int impnn_abi_version(void);
/* text of the calling thread's last error ("" if none) */
const char* impnn_last_error_string(void);
/* name of the code object target the library was built for ("gfx950") */
const char* impnn_target_arch(void);

/* ---- a1/a2: keras Embedding lookup (train_viscosity.py:163-164,171-172;
 *      train_melting_point.py:149-150,157-158).  out[r,:] = table[ids[r],:], r < rows. */
int impnn_embed_gather(const int32_t* ids, const float* table, float* out, int64_t rows,
                       int32_t vocab, int32_t dim, impnn_stream_t stream);

/* ---- a4: BondMatrixMessage.call with a dense bond_state (models/layers.py:100-117).
 *      h (B,N,D), bond_state (B,E,K), conn (B,E,2)=[src,tgt], W=bond_transform (K,D,D)
 *      -> messages (B,E,D); rows with src==0 or tgt==0 are zero (models/layers.py:114-115). */
int impnn_bmm_message(const float* h, const float* bond_state, const int32_t* conn, const float* W,
                      float* messages, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K,
                      impnn_stream_t stream);

/* ---- per-bond-type matrices (schedule A): A[v,i,j] = sum_k bond_table[v,k]*W[k,i,j].
 *      Equals tf.tensordot(bond_state, W) of models/layers.py:108 evaluated once per bond type
 *      instead of once per edge, valid because bond_state is always an Embedding lookup
 *      (train_viscosity.py:172, train_melting_point.py:158). out (Vb,D,D). */
int impnn_bond_type_matrices(const float* bond_table, const float* W, float* out, int32_t Vb,
                             int32_t K, int32_t D, impnn_stream_t stream);

/* ---- a4 from bond ids: messages[b,e,:] = A[bond_ids[b,e]] @ h[b,src[b,e],:], masked as a4. */
int impnn_bmm_message_typed(const float* h, const int32_t* bond_ids, const int32_t* conn,
                            const float* type_mats, float* messages, int32_t B, int32_t N,
                            int32_t E, int32_t D, int32_t Vb, impnn_stream_t stream);

/* ---- a5: Reduce.call (models/layers.py:57-83): agg[b,t,:] = sum_{e: tgt[b,e]=t, t>0} m[b,e,:],
 *      accumulated in edge-slot order (deterministic, == sequential scatter_nd on CPU).
 *      tgt is read as tgt[(b*E+e)*tgt_stride]: stride 1 for a contiguous (B,E) tensor, 2 to read
 *      conn[:,:,1] in place (pass conn+1), as the caller does at train_viscosity.py:182. */
int impnn_reduce_scatter_add(const float* messages, const int32_t* tgt, int32_t tgt_stride,
                             float* agg, int32_t B, int32_t N, int32_t E, int32_t D,
                             impnn_stream_t stream);

/* ---- a10: the orphan models/bond_matrix_message.py:37-65 signature: a4 then a5 in one launch,
 *      [h, bond_state, conn] -> agg (B,N,D).  W is (K,D*D) flat == (K,D,D) row-major. */
int impnn_bmm_fused(const float* h, const float* bond_state, const int32_t* conn, const float* W,
                    float* agg, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K,
                    impnn_stream_t stream);

/* ---- a7: GatedUpdate.call (models/layers.py:142-156) over `rows` = B*N atom rows (padding rows
 *      included, as the reference computes them).  Wz/Wr/Wh are keras Dense kernels (2D,D)
 *      (input-major), b* (D,), LayerNormalization gamma/beta (D,), eps = 1e-3 by keras default. */
int impnn_gated_update(const float* h, const float* agg, const float* Wz, const float* bz,
                       const float* Wr, const float* br, const float* Wh, const float* bh,
                       const float* gamma, const float* beta, float ln_eps, float* out,
                       int64_t rows, int32_t D, impnn_stream_t stream);

/* ---- a7 on a row list: the same arithmetic on rows row_index[0 .. *n_rows) of h / agg / out only (all other rows of
 *      `out` are left untouched).  The reference computes GatedUpdate on padding atoms too; their values can reach
 *      neither a message (no valid edge names them) nor the pool, so a caller that owns the whole encode() loop may
 *      skip them: impnn_kept_rows gives, per molecule, r_b = 1 + max(last n with atom_ids[b,n] > 0, largest atom
 *      index on a valid edge) (the kept rows are closed under "is a source of": skipping is exact), and
 *      impnn_row_index_fill turns r and its inclusive prefix sum into the flat list b*N + [0, r_b) and its length
 *      (*n_rows, device memory - no host round trip; the launch is sized for max_rows = B*N).  atom_dim 32 / 64 / 128. */
int impnn_gated_update_rows(const float* h, const float* agg, const float* Wz, const float* bz,
                            const float* Wr, const float* br, const float* Wh, const float* bh,
                            const float* gamma, const float* beta, float ln_eps, float* out,
                            const int32_t* row_index, const int32_t* n_rows, int64_t max_rows, int32_t D,
                            impnn_stream_t stream);
int impnn_kept_rows(const int32_t* atom_ids, const int32_t* bond_ids, const int32_t* conn, int32_t* rows_out,
                    int32_t B, int32_t N, int32_t E, int32_t Vb, impnn_stream_t stream);
int impnn_row_index_fill(const int32_t* kept_rows, const int32_t* kept_rows_inclusive_prefix,
                         int32_t* row_index, int32_t* n_rows, int32_t B, int32_t N, impnn_stream_t stream);

/* ---- a8: GlobalSumPool.call (models/layers.py:161-164): out[b,:] = sum_n h[b,n,:]*[ids[b,n]>0]. */
int impnn_global_sum_pool(const float* h, const int32_t* atom_ids, float* out, int32_t B,
                          int32_t N, int32_t D, impnn_stream_t stream);

/* ---- a9: encode() (train_viscosity.py:166-187; train_melting_point.py:152-171) up to and
 *      including GlobalSumPool, for `n_ions` independent ion branches in ONE launch
 *      (cation and anion: train_viscosity.py:193-194).
 *
 *  Per ion g < n_ions:  atom_ids[g] (B,N), bond_ids[g] (B,E), conn[g] (B,E,2), step weights
 *  weights[g] in the canonical packed layout below, pooled[g] (B,D).
 *  atom_table (Va,D) and bond_table (Vb,K) are shared by the ions (train_viscosity.py:163-164).
 *
 *  Canonical packed step-weight layout (floats), repeated S times per ion:
 *     bond_transform K*D*D | Wz 2D*D | bz D | Wr 2D*D | br D | Wh 2D*D | bh D | gamma D | beta D
 *  impnn_encoder_step_floats(D,K) returns that per-step count.
 *
 *  The arrays of pointers are HOST arrays of device pointers (length n_ions, n_ions <= 2). */
/*  `mode` - schedule and arithmetic of the encoder (an argument of every encoder entry; the library keeps no
 *  encoder state between calls):
 *    IMPNN_ENCODER_F32 (0):       "pull" form (agg = sum_k W_k G_k), v_mfma_f32_16x16x4_f32, exact f32 products;
 *                                 atom_dim 32, bond_dim <= 8.
 *    IMPNN_ENCODER_F16X2 (1):     pull form with each f32 operand split into two fp16 numbers, 3 fp16 MFMA products
 *                                 per f32 product, f32 accumulation (product error ~2^-21; narrower than f32).  Valid
 *                                 while |h|, |agg|, |G| < 4094 and |weights| < 255 - the caller's static bound
 *                                 (ionic_mpnn_amd.model checks it when weights are packed).
 *    IMPNN_ENCODER_F32_TYPED (2): per-bond-type form, the reference's own order of operations
 *                                 (models/layers.py:108-112): m_e = A[bond id of e] h[src_e] with
 *                                 A[v] = sum_k bond_table[v,k] W[k], on v_mfma_f32_4x4x1_16b_f32, exact f32 products;
 *                                 in-edge messages summed in edge-slot order.  atom_dim 32, ANY bond_dim (K = D^2 of
 *                                 train_melting_point.py:146 included), Vb <= 256, ANY padded shape N, E < 65536 (the
 *                                 explicit-hydrogen molecules of src/featurize.py:45 pad to E = 4 max_bonds,
 *                                 train_viscosity.py:288-289): one persistent kernel with the node state in LDS.  What
 *                                 bounds a molecule is what it HOLDS - kept rows <= 256, valid edges <= 512,
 *                                 in-degrees <= 255 - checked per batch by the plan kernels: a batch with a molecule
 *                                 beyond that gets NaN outputs and a nonzero int32 at byte
 *                                 impnn_encoder_plan_overflow_offset() of the workspace (device memory; a caller whose
 *                                 shapes allow it - N > 256 or E > 255 - reads the word back and takes the layer-at-a-
 *                                 time entries for that batch, as ionic_mpnn_amd.model does).  atom_dim 64 / 128 (train_viscosity.py with atom_dim=128,
 *                                 num_steps=6), N <= 256, E <= 1024, Vb <= 512: the same arithmetic as a short sequence
 *                                 of launches per call on compact kept rows (per type-run GEMMs for the messages,
 *                                 slot-order sums, GatedUpdate on 64-row tiles; csrc/encoder_wide.hip); `workgroups`
 *                                 is ignored there.
 *    IMPNN_ENCODER_F32X3_TYPED (3): mode 2 with the GatedUpdate GEMMs on the bf16 matrix pipe: every f32 operand is carried
 *                                 EXACTLY as three bf16 terms (3 x 8 significant bits, fp32's exponent range) and all nine
 *                                 cross products are accumulated in f32 (9 x v_mfma_f32_16x16x32_bf16 per 8 f32 MFMAs):
 *                                 the products are the f32 products, only their summation order differs.  atom_dim 32:
 *                                 messages stay on the f32 4x4x1 MFMA; atom_dim 64 / 128: the per-type message GEMMs
 *                                 run the same way.  Same shapes as mode 2 (padded E > 512 included), same records,
 *                                 prepared buffer of its own (impnn_encoder_prepared_bytes with this mode).  Opt-in.
 *  `workgroups` - persistent workgroups of the launch: 0 = default (environment IMPNN_ENCODER_WORKGROUPS if set - a
 *  process-wide diagnostics override, read ONCE at the first call - else one per compute unit), n = min(max(n, 16), CUs);
 *  a multiple of that for very large batches or padded shapes (the size query, plan and run agree on it by themselves).  It fixes the workspace layout, so the size query, the plan and
 *  the run of one batch must agree - impnn_encoder_plan returns what it used in its plan info and impnn_encoder_run
 *  takes it from there.  Why it is a knob: a caller that keeps several batches in flight on several streams gets more
 *  out of the chip with fewer, longer-running workgroups per launch (bench.py --streams 3: 128). */
#define IMPNN_ENCODER_F32 0
#define IMPNN_ENCODER_F16X2 1
#define IMPNN_ENCODER_F32_TYPED 2
#define IMPNN_ENCODER_F32X3_TYPED 3
int64_t impnn_encoder_step_floats(int32_t D, int32_t K);
size_t impnn_encoder_plan_overflow_offset(void);
int impnn_encoder_workspace_bytes(int32_t n_ions, int32_t B, int32_t N, int32_t E, int32_t D,
                                  int32_t K, int32_t S, int32_t Vb, int32_t mode, int32_t workgroups,
                                  size_t* bytes);
int impnn_encoder_fused(int32_t n_ions, const int32_t* const* atom_ids,
                        const int32_t* const* bond_ids, const int32_t* const* conn,
                        const float* atom_table, int32_t Va, const float* bond_table, int32_t Vb,
                        const float* const* weights, int32_t mode, float* const* pooled, int32_t B,
                        int32_t N, int32_t E, int32_t D, int32_t K, int32_t S, float ln_eps,
                        int32_t workgroups, void* workspace, size_t workspace_bytes,
                        impnn_stream_t stream);

/* ---- a9 with weights prepared once.  The kernel-side weight image (mode 0: transposed / padded f32; mode 1:
 *      pre-split fp16 hi/lo blocks; mode 2: the Vb per-bond-type matrices of every step in MFMA operand order plus
 *      the transposed GatedUpdate kernels) depends on the weights only, so a caller whose weights are fixed between
 *      calls (inference; every step of an epoch's evaluation) builds it once per ion and per mode and passes it to
 *      impnn_encoder_fused_prepared, which then launches only the two plan kernels and the encoder.
 *      impnn_encoder_fused (above) takes the canonical weights and rebuilds the image in the workspace on every call.
 *      `prepared`: device buffer of impnn_encoder_prepared_bytes(D, S, Vb, mode) bytes (0: shape not covered by the
 *      mode), 16B aligned;
 *      `bond_table` (Vb,K) is read by mode 2 only (may be NULL otherwise). */
size_t impnn_encoder_prepared_bytes(int32_t D, int32_t S, int32_t Vb, int32_t mode);
int impnn_encoder_prepare_weights(const float* weights, const float* bond_table, int32_t D, int32_t K,
                                  int32_t S, int32_t Vb, int32_t mode, void* prepared,
                                  size_t prepared_bytes, impnn_stream_t stream);
int impnn_encoder_fused_prepared(int32_t n_ions, const int32_t* const* atom_ids,
                                 const int32_t* const* bond_ids, const int32_t* const* conn,
                                 const float* atom_table, int32_t Va, const float* bond_table,
                                 int32_t Vb, const void* const* prepared, int32_t mode,
                                 float* const* pooled, int32_t B, int32_t N, int32_t E, int32_t D,
                                 int32_t K, int32_t S, float ln_eps, int32_t workgroups, void* workspace,
                                 size_t workspace_bytes, impnn_stream_t stream);

/* ---- f1: everything after GlobalSumPool in one launch.
 *      kind 0 (viscosity, train_viscosity.py:189,197-214 + models/layers.py:10-49):
 *        fp_g = relu(pooled_g Wfp_g + bfp_g); mixed = relu(fp_cat Wp_cat + bp_cat) + relu(fp_an Wp_an + bp_an);
 *        [A,b,c] = mixed Wv + bv; out = A + clip(softplus(b),0,20) / (T/100 + clip(softplus(c),0.1,50) + 1e-6)
 *      kind 1 (melting point, train_melting_point.py:173,191-198): out = relu(mixed Wh + bh) Wo + bo
 *      head_weights (keras Dense kernels (in,out) then bias, in this order; impnn_model_head_floats):
 *        Wfp_cat D*F | bfp_cat F | Wfp_an | bfp_an | Wp_cat F*Mx | bp_cat Mx | Wp_an | bp_an |
 *        kind 0: Wv Mx*3 | bv 3        kind 1: Wh Mx*F | bh F | Wo F | bo 1
 *      pooled_* (B,D), temperature (B,1) in kelvin (kind 0 only), out (B,1).  D <= 128 (atom_dim of the encoder:
 *      train_viscosity.py's 32, or 128 for the wider models), F, Mx <= 64 (fp_size 32, mixing_size 20 in the reference). */
int64_t impnn_model_head_floats(int32_t kind, int32_t D, int32_t F, int32_t Mx);
int impnn_model_head(int32_t kind, const float* pooled_cat, const float* pooled_an,
                     const float* temperature, const float* head_weights, float* out, int32_t B,
                     int32_t D, int32_t F, int32_t Mx, impnn_stream_t stream);

/* ---- f1 for training (f4): the same head read from the INDIVIDUAL weight tensors - `weights` is a host array of
 *      device pointers in the order of impnn_model_head's packed layout (10 tensors for kind 0, 12 for kind 1), so a
 *      training step does not re-pack the head after every optimizer update - and its backward in one launch.
 *      impnn_model_head_bwd recomputes the forward per sample, writes dpooled_cat / dpooled_an (B,D) and ADDS the
 *      parameter gradients to dweights[i] (same order and shapes as weights[i]; float atomics, so the sum order
 *      over workgroups is not fixed - fp32 rounding-level run-to-run differences in these ~5k values).
 *      Gradient of clip follows tf.clip_by_value / torch.clamp: passes where min <= x <= max.
 *      D <= 128, F, Mx <= 64 (these entries and the loss entries below).
 *      Replaces the autograd tape of train_viscosity.py:189-214 / train_melting_point.py:173-198. */
int impnn_model_head_tensors(int32_t kind, const float* pooled_cat, const float* pooled_an,
                             const float* temperature, const float* const* weights, float* out,
                             int32_t B, int32_t D, int32_t F, int32_t Mx, impnn_stream_t stream);
int impnn_model_head_bwd(int32_t kind, const float* pooled_cat, const float* pooled_an,
                         const float* temperature, const float* const* weights, const float* dout,
                         float* dpooled_cat, float* dpooled_an, float* const* dweights, int32_t B,
                         int32_t D, int32_t F, int32_t Mx, impnn_stream_t stream);

/* ---- the head with the training loss folded in (train_viscosity.py:189,227-230):
 *        loss = mean_b (pred_b - y_b)^2 + sum_t l2[t] * sum(W_t^2)
 *      keras "mse" plus the kernel_regularizer=l2(...) penalties; `l2` is a HOST array with one lambda per weight tensor
 *      in the order of `weights` (0 for tensors without a penalty).  impnn_model_head_loss writes the scalar `loss`
 *      (device) and, when `pred` is not null, the predictions (B); the squared errors are summed per workgroup in
 *      sample order and by the last workgroup to arrive in workgroup order, so the value is reproducible.  `workspace`:
 *      impnn_model_head_loss_workspace_floats(B) floats, the first 4 bytes ZERO before the first call (every call
 *      leaves them zero); one workspace per stream in flight.
 *      impnn_model_head_loss_bwd: `dloss` is a device scalar (the gradient of the loss value, 1 for a plain
 *      loss.backward(), n_local/n_global on a data-parallel rank); the kernel forms 2 (pred_b - y_b) / B * dloss
 *      itself and adds 2 l2[t] W_t dloss to the parameter gradients once.  Replaces ~25 elementwise / reduction
 *      launches of the autograd tape per training step. */
int64_t impnn_model_head_loss_workspace_floats(int32_t B);
int impnn_model_head_loss(int32_t kind, const float* pooled_cat, const float* pooled_an,
                          const float* temperature, const float* const* weights, const float* l2,
                          const float* y, float* pred, float* loss, float* workspace,
                          int64_t workspace_floats, int32_t B, int32_t D, int32_t F, int32_t Mx,
                          impnn_stream_t stream);
int impnn_model_head_loss_bwd(int32_t kind, const float* pooled_cat, const float* pooled_an,
                              const float* temperature, const float* const* weights, const float* l2,
                              const float* y, const float* dloss, float* dpooled_cat, float* dpooled_an,
                              float* const* dweights, int32_t B, int32_t D, int32_t F, int32_t Mx,
                              impnn_stream_t stream);

/* ---- a9 in two halves, for callers that pipeline batches.  impnn_encoder_plan runs only the
 *      graph-dependent plan kernels (row counts, shares, chunk records) of a batch into `workspace`;
 *      impnn_encoder_run runs only the encoder kernel from a planned workspace.  The plan needs no
 *      weights, so a caller may plan batch i+1 (any stream) before or while batch i is encoded and
 *      order the two with events; every batch in flight needs its own workspace.
 *      impnn_encoder_plan fills `info` (host memory, plain data) with the shape, record kind and workgroup count
 *      it planned for; impnn_encoder_run takes its launch geometry from `info` and returns IMPNN_E_BADARG when its
 *      own shape arguments or the record kind of `mode` (modes 0/1 share one, modes 2/3 share the other) differ.  The
 *      workspace itself starts with the same facts (device side): an encoder kernel that finds a plan made for
 *      another geometry writes NaN to `pooled` instead of reading records at wrong offsets.
 *      impnn_encoder_fused_prepared(...) == impnn_encoder_plan(...) then impnn_encoder_run(...). */
typedef struct impnn_encoder_plan_info {
  int32_t v[12];
} impnn_encoder_plan_info;
int impnn_encoder_plan(int32_t n_ions, const int32_t* const* atom_ids, const int32_t* const* bond_ids,
                       const int32_t* const* conn, int32_t B, int32_t N, int32_t E, int32_t D,
                       int32_t K, int32_t S, int32_t Va, int32_t Vb, int32_t mode, int32_t workgroups,
                       void* workspace, size_t workspace_bytes, impnn_stream_t stream,
                       impnn_encoder_plan_info* info);
int impnn_encoder_run(int32_t n_ions, const int32_t* const* atom_ids, const float* atom_table,
                      int32_t Va, const float* bond_table, int32_t Vb, const void* const* prepared,
                      int32_t mode, float* const* pooled, int32_t B, int32_t N, int32_t E, int32_t D,
                      int32_t K, int32_t S, float ln_eps, const impnn_encoder_plan_info* info,
                      void* workspace, size_t workspace_bytes, impnn_stream_t stream);

/* ---- f2: batch assembly on the GPU - the step before the path.  Replaces, per batch, the host list
 *      handling of train_viscosity.py:291-314: np.array(list)[idx] of id lists shifted by +1
 *      (train_viscosity.py:255-262), pad_sequences_1d (utils/mp_utils.py:12-16) and
 *      preprocess_edges_and_bonds (utils/mp_utils.py:18-45: every (src,tgt) followed by (tgt,src) with
 *      the same bond id, then [0,0]/0 padding or truncation to L = 2*max_edges slots).
 *      The id dataset of M samples is flattened once, per ion g, into ragged device arrays:
 *        atom_flat[g] raw atom ids, atom_off[g] (M+1) offsets; edge_flat[g] (src,tgt) pairs and
 *        bond_flat[g] raw bond ids, both indexed by edge_off[g] (M+1) offsets (a sample's edge and bond
 *        lists are cut to the shorter of the two, as the reference's zip() does).
 *      For b < B and s = sample_idx[b]:
 *        atom_ids[g][b,:]  = atom_flat[g][sample s] + id_shift, right-padded with 0 to N;
 *        conn[g][b,2e,:]   = edge e, conn[g][b,2e+1,:] = its reverse (NOT shifted), bond_ids[g][b,2e..2e+1] =
 *        bond e + id_shift; slots >= min(2*n_edges, L) are 0;  t_out[b] = t_flat[s] (optional, may be NULL).
 *      A sample index outside [0,M) yields an all-padding sample; a sample with more than N atoms is cut
 *      at N (the reference raises on both: the Python wrapper checks them on the host).
 *      Pointer arrays are HOST arrays of device pointers; edge_flat / conn 8-byte aligned. */
int impnn_batch_assemble(int32_t n_ions, const int32_t* sample_idx, int32_t B, int32_t M,
                         const int32_t* const* atom_flat, const int32_t* const* atom_off,
                         const int32_t* const* edge_flat, const int32_t* const* bond_flat,
                         const int32_t* const* edge_off, int32_t id_shift, int32_t N, int32_t L,
                         int32_t* const* atom_ids, int32_t* const* bond_ids, int32_t* const* conn,
                         const float* t_flat, float* t_out, impnn_stream_t stream);

/*  Mini-batch gather from a device-resident, already padded data set (model.fit over the arrays of
 *  train_viscosity.py:288-314): row rows[r] of tensor t -> row r of dst[t], for up to 8 tensors in one launch.
 *  src / dst / row_bytes are HOST arrays (device pointers, bytes per row: positive multiples of 4); `rows` is a device
 *  int64 array of n_rows indices, not range-checked (the caller built them from a permutation of the data set).
 *  As the first node of a captured training step the host only refreshes `rows` between replays. */
int impnn_gather_rows(int32_t n_tensors, const void* const* src, void* const* dst, const int64_t* row_bytes,
                      const int64_t* rows, int32_t n_rows, impnn_stream_t stream);

/* ---- f4: backward of the layer-at-a-time path and the optimizer step - what Keras autodiff and
 *      keras.optimizers.Adam(1e-3, clipnorm=1.0) do inside model.fit (train_viscosity.py:227-230,328-338;
 *      train_melting_point.py:205-208).  Every kernel is the adjoint of the forward entry of the same name,
 *      with the same masks and the same "out-of-range index == padding" rule.  Buffers marked (+=) must be
 *      zeroed (or hold a running sum) before the call: they are accumulated with float atomics.
 *
 *  Embedding (a1/a2):        dtable[ids[r],:] (+=) dout[r,:]
 *  Reduce (a5, :57-83):      dmessages[b,e,:] = tgt > 0 ? dagg[b,tgt,:] : 0
 *  GlobalSumPool (a8):       dh[b,n,:] = atom_ids[b,n] > 0 ? dpooled[b,:] : 0
 *  BondMatrixMessage (a4) in the per-bond-type schedule (impnn_bond_type_matrices + impnn_bmm_message_typed):
 *      dh[b,src,:] (+=) A[type]^T dmessages[b,e,:];   dtype_mats[type] (+=) dmessages[b,e,:] (x) h[b,src,:]
 *      (the batch's valid edges are counting-sorted by type in `workspace`, one workgroup per <=64 edges of a type;
 *      sorted_ready != 0: `workspace` still holds the sort of the same (conn, bond_ids) from an earlier call)
 *      then   dW[k] = sum_v Tb[v,k] dtype_mats[v];     dbond_table[v,k] = <dtype_mats[v], W[k]>
 *  GatedUpdate (a7, :142-156): dh, dagg (rows,D) and dparams in the canonical order
 *      Wz 2D*D | bz D | Wr | br | Wh | bh | gamma | beta  (impnn_gated_update_param_floats(D) floats, overwritten);
 *      intermediates are recomputed from (h, agg); the kernel gradients are split-K GEMMs over row chunks and
 *      all parameter sums go through partials in `workspace` (impnn_gated_update_bwd_workspace_floats) that
 *      are added in a fixed order: bitwise reproducible.  atom_dim must divide 256.
 *  `accumulate` != 0 (GatedUpdate dparams; dW and dbond_table of impnn_bond_type_matrices_bwd with K < 64): the
 *      results are ADDED to the output buffers - the caller points them at the optimizer's gradient buffer and
 *      saves one add per parameter tensor. */
int impnn_embed_gather_bwd(const int32_t* ids, const float* dout, float* dtable, int64_t rows, int32_t vocab,
                           int32_t dim, impnn_stream_t stream);
int impnn_reduce_scatter_bwd(const float* dagg, const int32_t* tgt, int32_t tgt_stride, float* dmessages, int32_t B,
                             int32_t N, int32_t E, int32_t D, impnn_stream_t stream);
int impnn_global_sum_pool_bwd(const float* dpooled, const int32_t* atom_ids, float* dh, int32_t B, int32_t N,
                              int32_t D, impnn_stream_t stream);
int64_t impnn_bmm_message_typed_bwd_workspace_bytes(int32_t B, int32_t E, int32_t Vb);
/*  impnn_bmm_message_typed (forward) over the same type-sorted edge segments, any D <= 128: A[type] staged in LDS,
 *  one workgroup per <= 64 edges of a type.  Same workspace (and size query) as the backward entry: a sort made here
 *  serves the backward call of the layer and every other layer of the ion (sorted_ready = 1 there).
 *  sorted_ready | 2: `messages` is the buffer an earlier call on the SAME (bond_ids, conn) wrote and nothing else
 *  touched since - the zero rows of masked / out-of-range edges (models/layers.py:114-115) are still in place and
 *  the pass that writes them is skipped (a training loop that keeps one message buffer per ion for all its layers). */
int impnn_bmm_message_typed_sorted(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                   const float* type_mats, float* messages, void* workspace, int64_t workspace_bytes,
                                   int32_t B, int32_t N, int32_t E, int32_t D, int32_t Vb, int32_t sorted_ready,
                                   impnn_stream_t stream);
int impnn_bmm_message_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                const float* type_mats, const float* dmessages, float* dh, float* dtype_mats,
                                void* workspace, int64_t workspace_bytes, int32_t B, int32_t N, int32_t E, int32_t D,
                                int32_t Vb, int32_t sorted_ready, impnn_stream_t stream);
/*  Backward of Reduce o BondMatrixMessage in one launch (models/layers.py:57-83 after :100-117): the gradient of a
 *  message IS the gradient of the aggregate row it was added to, so the (B,E,D) message gradient is never
 *  materialised - the kernel reads dagg (B,N,D) at row tgt(e).  Otherwise identical to impnn_bmm_message_typed_bwd. */
int impnn_message_reduce_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                   const float* type_mats, const float* dagg, float* dh, float* dtype_mats,
                                   void* workspace, int64_t workspace_bytes, int32_t B, int32_t N, int32_t E,
                                   int32_t D, int32_t Vb, int32_t sorted_ready, impnn_stream_t stream);
/*  The same with a (B,E,D) buffer whose rows at masked / out-of-range edges are ZERO - the message buffer an
 *  impnn_bmm_message_typed_sorted call of the same (bond_ids, conn) left behind is one (its other rows are overwritten
 *  here).  The per-edge vectors A_t^T g_e go to their edge slot's row and a slot-order pass adds them into dh at the
 *  source rows - no float atomics on dh (22 M of them at atom_dim 128, batch 4096: 260 of the kernel's 349 us), and
 *  dh becomes bitwise reproducible. */
int impnn_message_reduce_typed_bwd_scratch(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                           const float* type_mats, const float* dagg, float* dh, float* dtype_mats,
                                           void* workspace, int64_t workspace_bytes, float* edge_scratch, int32_t B,
                                           int32_t N, int32_t E, int32_t D, int32_t Vb, int32_t sorted_ready,
                                           impnn_stream_t stream);
int impnn_bond_type_matrices_bwd(const float* bond_table, const float* W, const float* dtype_mats, float* dW,
                                 float* dbond_table, int32_t Vb, int32_t K, int32_t D, int32_t accumulate,
                                 impnn_stream_t stream);
/*  The per-bond-type matrices of ALL message layers of a model (both ions, every step) in one launch, and their
 *  backward in two: `W`, `type_mats`, `dtype_mats`, `dW` are host arrays of n device pointers (layer p: W_p (K,D,D),
 *  A_p (Vb,D,D)); the bond embedding table is shared (train_viscosity.py:172), so dbond_table (Vb,K) is the sum over
 *  the layers, accumulated in layer order.  accumulate as in impnn_bond_type_matrices_bwd.  Same arithmetic per layer
 *  as the single-layer entries (bitwise equal A_p and dW_p).  A training step at the reference's batch 32 is bound by
 *  the number of launches: 3 launches here replace 3 per layer. */
int impnn_bond_type_matrices_multi(const float* bond_table, const float* const* W, float* const* type_mats,
                                   int32_t n, int32_t Vb, int32_t K, int32_t D, impnn_stream_t stream);
int impnn_bond_type_matrices_multi_bwd(const float* bond_table, const float* const* W,
                                       const float* const* dtype_mats, float* const* dW, float* dbond_table,
                                       int32_t n, int32_t Vb, int32_t K, int32_t D, int32_t accumulate,
                                       impnn_stream_t stream);
/* The same with a workspace (impnn_bond_type_matrices_multi_bwd_workspace_floats floats, 16-byte aligned): the bond-table
 * gradient - a (Vb x K) result over a contraction of n D^2 - runs on the matrix cores, per-wave partials in the
 * workspace, added in a fixed order (bitwise reproducible; 116-170 us -> ~15 us at atom_dim 128, all of it on the
 * critical path of a training step).  K <= 16, Vb <= 128, D % 16 == 0; other shapes take the entry above's kernels. */
int64_t impnn_bond_type_matrices_multi_bwd_workspace_floats(int32_t n, int32_t Vb, int32_t K, int32_t D);
int impnn_bond_type_matrices_multi_bwd_ws(const float* bond_table, const float* const* W,
                                          const float* const* dtype_mats, float* const* dW, float* dbond_table,
                                          int32_t n, int32_t Vb, int32_t K, int32_t D, int32_t accumulate,
                                          float* workspace, int64_t workspace_floats, impnn_stream_t stream);
int64_t impnn_gated_update_param_floats(int32_t D);
int64_t impnn_gated_update_bwd_workspace_floats(int64_t rows, int32_t D);
int impnn_gated_update_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                           const float* br, const float* Wh, const float* bh, const float* gamma, float ln_eps,
                           const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                           int64_t workspace_floats, int64_t rows, int32_t D, int32_t accumulate,
                           impnn_stream_t stream);
/* a7 backward on a row list (the adjoint of impnn_gated_update_rows; atom_dim 64 / 128): gradients of the rows
 * row_index[0 .. *n_rows) only - dh / dagg rows outside the list are left untouched (a caller that reads them zeroes
 * them first), the parameter gradients are sums over the listed rows.  Padding atoms of an encode() loop carry no
 * gradient (nothing they compute reaches a message or the pool), so this is exact for impnn_kept_rows' list.
 * The launch is sized for max_rows; *n_rows lives on the device (no host round trip, capturable). */
int64_t impnn_gated_update_rows_bwd_workspace_floats(int64_t max_rows, int32_t D);
int impnn_gated_update_rows_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                                const float* br, const float* Wh, const float* bh, const float* gamma, float ln_eps,
                                const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                                int64_t workspace_floats, const int32_t* row_index, const int32_t* n_rows,
                                int64_t max_rows, int32_t D, int32_t accumulate, impnn_stream_t stream);
/* The same pair for a training loop that keeps activations instead of recomputing them (atom_dim 64 / 128, and 32
 * without a row list - workspace of impnn_gated_update_bwd_workspace_floats there):
 * impnn_gated_update_rows_train is impnn_gated_update_rows that also writes, per LISTED row (by list position), the
 * gates z, r, the candidate tanh(.) and r * h of models/layers.py:145-152 into `saved`
 * (impnn_gated_update_rows_saved_floats(max_rows, D) = 4 D max_rows floats, 16-byte aligned);
 * impnn_gated_update_rows_bwd_saved is impnn_gated_update_rows_bwd without its two recompute GEMM passes (half of its
 * matrix work): same arguments, same workspace size, plus that buffer - which it CONSUMES (it comes back holding the
 * pre-activation gradients; a second backward over it needs a second forward).  Both take row_index = n_rows = NULL
 * for "all max_rows rows" (the small batches a training loop does not build a list for). */
int64_t impnn_gated_update_rows_saved_floats(int64_t max_rows, int32_t D);
int impnn_gated_update_rows_train(const float* h, const float* agg, const float* Wz, const float* bz,
                                  const float* Wr, const float* br, const float* Wh, const float* bh,
                                  const float* gamma, const float* beta, float ln_eps, float* out,
                                  const int32_t* row_index, const int32_t* n_rows, int64_t max_rows, int32_t D,
                                  float* saved, impnn_stream_t stream);
int impnn_gated_update_rows_bwd_saved(const float* h, const float* agg, const float* Wz, const float* bz,
                                      const float* Wr, const float* br, const float* Wh, const float* bh,
                                      const float* gamma, float ln_eps, const float* dout, float* dh, float* dagg,
                                      float* dparams, float* workspace, int64_t workspace_floats,
                                      const int32_t* row_index, const int32_t* n_rows, int64_t max_rows, int32_t D,
                                      int32_t accumulate, float* saved, impnn_stream_t stream);

/*  Optimizer step, one launch for all variables (train_viscosity.py:227-230):
 *      g <- g * clipnorm / max(||g||_2, clipnorm)      per variable (tf.clip_by_norm); clipnorm <= 0: off
 *      m <- b1 m + (1-b1) g;   v <- b2 v + (1-b2) g^2
 *      w <- w - lr * sqrt(1 - b2^step) / (1 - b1^step) * m / (sqrt(v) + eps)         (step counts from 1)
 *  var_table: DEVICE array of 4*n_vars device pointers (w, g, m, v per variable, all f32);
 *  sizes: DEVICE array of n_vars element counts. */
int impnn_adam_clipnorm_step(const void* var_table, const int64_t* sizes, int32_t n_vars, int64_t step, float lr,
                             float beta1, float beta2, float eps, float clipnorm, impnn_stream_t stream);
/*  The same with the step number in DEVICE memory: *step_counter is incremented by one (on the stream) and the
 *  new value is the `step` of this update.  A training step captured in a hipGraph (forward, backward and this
 *  call) then advances its own counter on every replay. */
int impnn_adam_clipnorm_step_counted(const void* var_table, const int64_t* sizes, int32_t n_vars,
                                     int64_t* step_counter, float lr, float beta1, float beta2, float eps,
                                     float clipnorm, impnn_stream_t stream);

/* ---- measurement: HIP-event timing of the dominant kernel (encoder_fused_kernel / encoder_typed_kernel), recorded on the
 *      stream the kernel is launched on.  After impnn_profile_enable(capacity) every
 *      impnn_encoder_fused call of this thread records one (start, stop) event pair around that
 *      kernel alone (the two small plan kernels are outside the pair) until `capacity` pairs exist.
 *      impnn_profile_collect synchronises the recorded events, writes up to max_n durations in
 *      milliseconds, returns their count in *n_out and rewinds.  Used by bench.py's roofline leg. */
int impnn_profile_enable(int32_t capacity);
int impnn_profile_collect(float* ms_out, int32_t max_n, int32_t* n_out);
int impnn_profile_disable(void);

/* ---- diagnostics: when a device buffer of >= 256 bytes per encoder workgroup is set, the encoder
 *      kernel's lane 0 writes s_memtime stamps into it (entry, after prologue, after each of the
 *      first 5 steps, exit, and the phase boundaries of wave 0's first tile in one step) -
 *      32 uint64 per workgroup.  NULL (the default) disables it; with NULL
 *      no stamp instruction executes.  Never set during a timed run. */
int impnn_debug_set_stamp_buffer(void* device_buffer, size_t bytes);

/* ---- debug: counts indices the reference's CPU path would raise on.  counts[0] += #conn
 *      entries outside [0,N), counts[1] += #atom ids outside [0,Va), counts[2] += #bond ids
 *      outside [0,Vb).  Any of conn/atom_ids/bond_ids may be NULL.  counts: 3 device int32,
 *      zeroed by the caller. */
int impnn_validate_indices(const int32_t* conn, const int32_t* atom_ids, const int32_t* bond_ids,
                           int32_t* counts, int32_t B, int32_t N, int32_t E, int32_t Va,
                           int32_t Vb, impnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMPNN_H_ */
