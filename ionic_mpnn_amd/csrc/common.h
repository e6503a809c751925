// Shared host-side helpers for libimpnn.so (gfx950 only; no other target is supported).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/impnn.h"

namespace impnn {

// thread-local last-error text (impnn_last_error_string)
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(impnn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(IMPNN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return IMPNN_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;  // CDNA wavefront

// ---- layer-at-a-time launches (layer_kernels.hip)
int launch_embed_gather(const int32_t* ids, const float* table, float* out, int64_t rows, int vocab,
                        int dim, hipStream_t s);
int launch_bmm_message(const float* h, const float* bs, const int32_t* conn, const float* W, float* m,
                       float* agg, int B, int N, int E, int D, int K, hipStream_t s);
int launch_bond_type_matrices(const float* tb, const float* W, float* out, int Vb, int K, int D,
                              hipStream_t s);
int launch_bmm_message_typed(const float* h, const int32_t* bond_ids, const int32_t* conn,
                             const float* type_mats, float* m, int B, int N, int E, int D, int Vb,
                             hipStream_t s);
int launch_reduce_scatter_add(const float* m, const int32_t* tgt, int tgt_stride, float* agg, int B,
                              int N, int E, int D, hipStream_t s, int accumulate = 0);
int launch_gated_update(const float* h, const float* agg, const float* Wz, const float* bz,
                        const float* Wr, const float* br, const float* Wh, const float* bh,
                        const float* gamma, const float* beta, float eps, float* out, int64_t rows,
                        int D, hipStream_t s, const int32_t* ridx = nullptr, const int32_t* nrows_dev = nullptr,
                        float* save = nullptr);
int launch_kept_rows(const int32_t* atom_ids, const int32_t* bond_ids, const int32_t* conn, int32_t* rows_out, int B,
                     int N, int E, int Vb, hipStream_t s);
int launch_row_index_fill(const int32_t* r, const int32_t* incl, int32_t* idx, int32_t* count, int B, int N,
                          hipStream_t s);
int launch_global_sum_pool(const float* h, const int32_t* ids, float* out, int B, int N, int D,
                           hipStream_t s);
int64_t model_head_loss_workspace_floats(int B);
int launch_model_head_tensors(int kind, const float* pc, const float* pa, const float* T, const float* const* weights,
                              float* out, int B, int D, int F, int Mx, hipStream_t s, const float* l2 = nullptr,
                              const float* y = nullptr, float* loss_out = nullptr, float* workspace = nullptr);
int launch_model_head_bwd(int kind, const float* pc, const float* pa, const float* T, const float* const* weights,
                          const float* dout, float* dpc, float* dpa, float* const* grads, int B, int D, int F, int Mx,
                          hipStream_t s, const float* l2 = nullptr, const float* y = nullptr,
                          const float* dloss = nullptr);
int launch_gather_rows(int n, const void* const* src, void* const* dst, const int64_t* row_bytes, const int64_t* rows,
                       int n_rows, hipStream_t s);
int launch_model_head(int kind, const float* pc, const float* pa, const float* T, const float* w, float* out, int B,
                      int D, int F, int Mx, hipStream_t s);
int launch_validate_indices(const int32_t* conn, const int32_t* atom_ids, const int32_t* bond_ids,
                            int32_t* counts, int B, int N, int E, int Va, int Vb, hipStream_t s);

// ---- backward + optimizer (train_kernels.hip)
int launch_embed_gather_bwd(const int32_t* ids, const float* dout, float* dtable, int64_t rows, int vocab, int dim,
                            hipStream_t s);
int launch_reduce_scatter_bwd(const float* dagg, const int32_t* tgt, int tgt_stride, float* dm, int B, int N, int E,
                              int D, hipStream_t s);
int launch_global_sum_pool_bwd(const float* dp, const int32_t* ids, float* dh, int B, int N, int D, hipStream_t s);
int64_t bmm_message_typed_bwd_workspace_ints(int B, int E, int Vb);
int launch_bmm_message_typed_sorted(const float* h, const int32_t* bond_ids, const int32_t* conn, const float* A,
                                    float* m, int32_t* workspace, int B, int N, int E, int D, int Vb, int sorted_ready,
                                    hipStream_t s);
int launch_bmm_message_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn, const float* A,
                                 const float* dm, float* dh, float* dA, int32_t* workspace, int B, int N, int E,
                                 int D, int Vb, int sorted_ready, int from_agg, hipStream_t s, float* du = nullptr);
int launch_strided_gemm(const float* A, const float* B, float* out, int64_t rows, int M, int N, int64_t a_rs,
                        int64_t a_cs, int64_t b_rs, int64_t b_cs, hipStream_t s);
int launch_bond_type_matrices_multi(const float* tb, const float* const* W, float* const* out, int n, int Vb, int K,
                                    int D, hipStream_t s);
int launch_bond_type_matrices_multi_bwd(const float* tb, const float* const* W, const float* const* dA,
                                        float* const* dW, float* dtb, int n, int Vb, int K, int D, int accumulate,
                                        hipStream_t s, float* workspace = nullptr);
int64_t bond_type_matrices_multi_bwd_workspace(int n, int Vb, int K, int D);
int launch_bond_type_matrices_bwd(const float* tb, const float* W, const float* dA, float* dW, float* dtb, int Vb,
                                  int K, int D, int accumulate, hipStream_t s);
int gated_update_bwd_blocks(int64_t rows, int D);
int64_t gated_update_param_floats(int D);
int64_t gated_update_bwd_workspace(int64_t rows, int D, bool row_list = false);
int launch_gated_update_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                            const float* br, const float* Wh, const float* bh, const float* gamma, float eps,
                            const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                            int64_t rows, int D, int accumulate, hipStream_t s, const int32_t* ridx = nullptr,
                            const int32_t* nrows_dev = nullptr, float* saved = nullptr);
int launch_adam_clipnorm(const void* table, const void* sizes, int n_vars, int64_t step, int64_t* step_dev, float lr,
                         float b1, float b2, float eps, float clipnorm, hipStream_t s);

// ---- batch assembly (loader_kernels.hip)
int launch_batch_assemble(int n_ions, const int32_t* sample_idx, int B, int M, const int32_t* const* atom_flat,
                          const int32_t* const* atom_off, const int32_t* const* edge_flat,
                          const int32_t* const* bond_flat, const int32_t* const* edge_off, int shift, int N, int L,
                          int32_t* const* atom_ids, int32_t* const* bond_ids, int32_t* const* conn,
                          const float* t_flat, float* t_out, hipStream_t s);

// ---- event-pair profiler (api.hip); record_* are no-ops unless enabled on this thread
void profile_record_start(hipStream_t s);
void profile_record_stop(hipStream_t s);

void* debug_stamp_buffer(size_t* bytes);  // thread-local diagnostics buffer (api.hip), usually null

// ---- fused encoders (encoder_fused.hip: pull form, modes 0/1; encoder_typed.hip: per-bond-type form, mode 2)
struct EncoderArgs {
  int n_ions;
  const int32_t* atom_ids[2];
  const int32_t* bond_ids[2];
  const int32_t* conn[2];
  const float* weights[2];   // canonical packed step weights (used when prepared[g] is null)
  const void* prepared[2];   // impnn_encoder_prepare_weights output for `mode`, or null
  int mode;                  // 0 f32, 1 f16x2, 2 f32 typed
  int phases;                // bit 0: plan kernels, bit 1: encoder kernel
  int nwg;                   // resolved persistent workgroups (encoder_workgroups), same for plan and run
  float* pooled[2];
  const float* atom_table;
  const float* bond_table;
  int Va, Vb, B, N, E, D, K, S;
  float ln_eps;
  void* workspace;
  size_t workspace_bytes;
};
bool encoder_fused_supported(int mode, int N, int E, int D, int K, int S, int Vb);
int encoder_workgroups(int n_ions, int B, int requested, int N, int E, int mode);
size_t encoder_fused_workspace_bytes(int mode, int n_ions, int B, int N, int E, int D, int S, int Vb, int nwg);
int launch_encoder_fused(const EncoderArgs& a, hipStream_t s);
int launch_encoder_phase(const EncoderArgs& a, hipStream_t s, bool plan_phase);
size_t encoder_prepared_bytes(int mode, int D, int S, int Vb);
int launch_encoder_prepare(const float* weights, const float* bond_table, int D, int K, int S, int Vb, int mode,
                           void* prepared, hipStream_t s);
int ensure_lds_limit(const void* kern, int slot);
int device_compute_units();  // CUs of the current device (cached per device index)
// Rows per workgroup of the wide (atom_dim 64 / 128) GatedUpdate forward / backward kernels: 64, or 16 below 8 K rows,
// where 64-row tiles cannot fill the chip (one definition: the launchers and the workspace sizing must agree).
inline int gu_wide_tile_rows(int64_t rows) { return rows < 8192 ? 16 : 64; }
// ---- wide states (encoder_wide.hip: atom_dim 64 / 128 behind the same entries, mode 2)
bool encoder_wide_supported(int N, int E, int D, int K, int S, int Vb);
size_t encoder_wide_workspace_bytes(int n_ions, int B, int N, int E, int D, int S, int Vb, bool x3);
size_t encoder_wide_prepared_bytes(int D, int S, int Vb, bool x3);
int launch_encoder_wide_prepare(const float* weights, const float* bond_table, int D, int K, int S, int Vb, bool x3,
                                void* prepared, hipStream_t s);
int launch_encoder_wide(const EncoderArgs& a, hipStream_t s);

}  // namespace impnn
