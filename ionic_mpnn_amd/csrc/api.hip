// extern "C" boundary of libimpnn.so: argument checks, then launches.  See include/impnn.h.
#include <cstring>
#include <vector>

#include <cstddef>

#include "common.h"
#include "encoder_layout.h"

namespace impnn {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct Profiler {
  std::vector<hipEvent_t> start, stop;
  int used = 0;
  bool enabled = false;
  bool open = false;  // a start without its stop
};
thread_local Profiler g_prof;
}  // namespace

namespace {
thread_local void* g_stamp_ptr = nullptr;
thread_local size_t g_stamp_bytes = 0;
}  // namespace

void* debug_stamp_buffer(size_t* bytes) {
  *bytes = g_stamp_bytes;
  return g_stamp_ptr;
}

void profile_record_start(hipStream_t s) {
  Profiler& p = g_prof;
  if (!p.enabled || p.used >= (int)p.start.size()) return;
  if (hipEventRecord(p.start[p.used], s) == hipSuccess) p.open = true;
}

void profile_record_stop(hipStream_t s) {
  Profiler& p = g_prof;
  if (!p.enabled || !p.open) return;
  (void)hipEventRecord(p.stop[p.used], s);
  p.open = false;
  ++p.used;
}

}  // namespace impnn

using namespace impnn;

#define REQUIRE(cond, what)                                              \
  do {                                                                   \
    if (!(cond)) return fail(IMPNN_E_BADARG, "%s: %s", __func__, what); \
  } while (0)

extern "C" {

int impnn_abi_version(void) { return IMPNN_ABI_VERSION; }
const char* impnn_last_error_string(void) { return error_buffer(); }
const char* impnn_target_arch(void) { return "gfx950"; }

int impnn_embed_gather(const int32_t* ids, const float* table, float* out, int64_t rows, int32_t vocab,
                       int32_t dim, impnn_stream_t stream) {
  REQUIRE(rows >= 0 && vocab > 0 && dim > 0, "rows>=0, vocab>0, dim>0 required");
  REQUIRE(rows == 0 || (ids && table && out), "null pointer");
  return launch_embed_gather(ids, table, out, rows, vocab, dim, as_stream(stream));
}

int impnn_bmm_message(const float* h, const float* bond_state, const int32_t* conn, const float* W,
                      float* messages, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K,
                      impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && K > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_state && conn && W && messages, "null pointer");
  return launch_bmm_message(h, bond_state, conn, W, messages, nullptr, B, N, E, D, K, as_stream(stream));
}

int impnn_bond_type_matrices(const float* bond_table, const float* W, float* out, int32_t Vb, int32_t K,
                             int32_t D, impnn_stream_t stream) {
  REQUIRE(Vb > 0 && K > 0 && D > 0, "bad shape");
  REQUIRE(bond_table && W && out, "null pointer");
  return launch_bond_type_matrices(bond_table, W, out, Vb, K, D, as_stream(stream));
}

int impnn_bmm_message_typed(const float* h, const int32_t* bond_ids, const int32_t* conn,
                            const float* type_mats, float* messages, int32_t B, int32_t N, int32_t E,
                            int32_t D, int32_t Vb, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && Vb > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_ids && conn && type_mats && messages, "null pointer");
  return launch_bmm_message_typed(h, bond_ids, conn, type_mats, messages, B, N, E, D, Vb, as_stream(stream));
}

int impnn_reduce_scatter_add(const float* messages, const int32_t* tgt, int32_t tgt_stride, float* agg,
                             int32_t B, int32_t N, int32_t E, int32_t D, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0, "bad shape");
  REQUIRE(tgt_stride >= 1, "tgt_stride must be >= 1");
  if (B == 0) return IMPNN_OK;
  REQUIRE(agg && (E == 0 || (messages && tgt)), "null pointer");
  return launch_reduce_scatter_add(messages, tgt, tgt_stride, agg, B, N, E, D, as_stream(stream));
}

int impnn_bmm_fused(const float* h, const float* bond_state, const int32_t* conn, const float* W, float* agg,
                    int32_t B, int32_t N, int32_t E, int32_t D, int32_t K, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E > 0 && D > 0 && K > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(h && bond_state && conn && W && agg, "null pointer");
  return launch_bmm_message(h, bond_state, conn, W, nullptr, agg, B, N, E, D, K, as_stream(stream));
}

int impnn_gated_update(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                       const float* br, const float* Wh, const float* bh, const float* gamma,
                       const float* beta, float ln_eps, float* out, int64_t rows, int32_t D,
                       impnn_stream_t stream) {
  REQUIRE(rows >= 0 && D > 0, "bad shape");
  if (rows == 0) return IMPNN_OK;
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && beta && out, "null pointer");
  REQUIRE(ln_eps >= 0.f, "ln_eps must be >= 0");
  return launch_gated_update(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, ln_eps, out, rows, D,
                             as_stream(stream));
}

int impnn_gated_update_rows(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                            const float* br, const float* Wh, const float* bh, const float* gamma,
                            const float* beta, float ln_eps, float* out, const int32_t* row_index,
                            const int32_t* n_rows, int64_t max_rows, int32_t D, impnn_stream_t stream) {
  REQUIRE(max_rows >= 0 && D > 0, "bad shape");
  if (max_rows == 0) return IMPNN_OK;
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && beta && out && row_index && n_rows, "null pointer");
  REQUIRE(ln_eps >= 0.f, "ln_eps must be >= 0");
  return launch_gated_update(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, ln_eps, out, max_rows, D, as_stream(stream),
                             row_index, n_rows);
}

int impnn_kept_rows(const int32_t* atom_ids, const int32_t* bond_ids, const int32_t* conn, int32_t* rows_out,
                    int32_t B, int32_t N, int32_t E, int32_t Vb, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && Vb > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(atom_ids && rows_out && (E == 0 || conn), "null pointer");
  return launch_kept_rows(atom_ids, bond_ids, conn, rows_out, B, N, E, Vb, as_stream(stream));
}

int impnn_row_index_fill(const int32_t* kept_rows, const int32_t* kept_rows_inclusive_prefix, int32_t* row_index,
                         int32_t* n_rows, int32_t B, int32_t N, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(kept_rows && kept_rows_inclusive_prefix && row_index && n_rows, "null pointer");
  return launch_row_index_fill(kept_rows, kept_rows_inclusive_prefix, row_index, n_rows, B, N, as_stream(stream));
}

int impnn_global_sum_pool(const float* h, const int32_t* atom_ids, float* out, int32_t B, int32_t N,
                          int32_t D, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && D > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(h && atom_ids && out, "null pointer");
  return launch_global_sum_pool(h, atom_ids, out, B, N, D, as_stream(stream));
}

size_t impnn_encoder_plan_overflow_offset(void) { return offsetof(enc::PlanHeader, overflow); }

int64_t impnn_encoder_step_floats(int32_t D, int32_t K) {
  if (D <= 0 || K <= 0) return -1;
  const int64_t d = D, k = K;
  return k * d * d + 3 * (2 * d * d + d) + 2 * d;
}

namespace {
constexpr int32_t kInfoMagic = 0x706c616e;  // "plan"
// impnn_encoder_plan_info.v: magic, mode class (0 pull records / 1 typed records), n_ions, B, N, E, S, Vb, nwg, D, K,
// record kind (0 pull, 1 typed atom_dim 32, 2 wide atom_dim 64 / 128: three different workspace layouts)
inline int mode_class(int mode) { return mode >= 2 ? 1 : 0; }
inline int record_kind(int mode, int D) { return mode >= 2 ? (D == 32 ? 1 : 2) : 0; }
}  // namespace

int impnn_encoder_workspace_bytes(int32_t n_ions, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K,
                                  int32_t S, int32_t Vb, int32_t mode, int32_t workgroups, size_t* bytes) {
  REQUIRE(bytes, "null pointer");
  REQUIRE(n_ions >= 1 && n_ions <= 2 && B >= 0 && N > 0 && E >= 0 && D > 0 && K > 0 && S >= 0 && Vb > 0,
          "bad shape");
  REQUIRE(mode >= 0 && mode <= 3, "mode must be 0 (f32), 1 (f16x2), 2 (f32 typed) or 3 (f32x3 typed)");
  REQUIRE(workgroups >= 0, "workgroups must be >= 0 (0: default)");
  if (!encoder_fused_supported(mode, N, E, D, K, S, Vb))
    return fail(IMPNN_E_UNSUPPORTED, "encoder_fused: mode=%d shape N=%d E=%d D=%d K=%d S=%d Vb=%d not covered", mode,
                N, E, D, K, S, Vb);
  *bytes = encoder_fused_workspace_bytes(mode, n_ions, B, N, E, D, S, Vb, encoder_workgroups(n_ions, B, workgroups, N, E, mode));
  return IMPNN_OK;
}

static int encoder_common(const char* fn, int32_t n_ions, const int32_t* const* atom_ids,
                          const int32_t* const* bond_ids, const int32_t* const* conn, const float* atom_table,
                          int32_t Va, const float* bond_table, int32_t Vb, const float* const* weights,
                          const void* const* prepared, int32_t mode, float* const* pooled, int32_t B, int32_t N,
                          int32_t E, int32_t D, int32_t K, int32_t S, float ln_eps, int32_t workgroups,
                          const impnn_encoder_plan_info* info_in, impnn_encoder_plan_info* info_out, void* workspace,
                          size_t workspace_bytes, impnn_stream_t stream, int phases = 3) {
#define REQ(cond, what)                                           \
  do {                                                            \
    if (!(cond)) return fail(IMPNN_E_BADARG, "%s: %s", fn, what); \
  } while (0)
  REQ(n_ions >= 1 && n_ions <= 2, "n_ions must be 1 or 2");
  REQ(mode >= 0 && mode <= 3, "mode must be 0 (f32), 1 (f16x2), 2 (f32 typed) or 3 (f32x3 typed)");
  REQ(workgroups >= 0, "workgroups must be >= 0 (0: default)");
  REQ(B >= 0 && N > 0 && E >= 0 && D > 0 && K > 0 && S >= 0 && Va > 0 && Vb > 0, "bad shape");
  const bool planning = (phases & 1) != 0, running = (phases & 2) != 0;
  REQ(atom_ids && (!planning || (bond_ids && conn)), "null pointer");
  REQ(!running || ((weights || prepared) && pooled && atom_table && bond_table), "null pointer");
  if (!encoder_fused_supported(mode, N, E, D, K, S, Vb))
    return fail(IMPNN_E_UNSUPPORTED, "encoder_fused: mode=%d shape N=%d E=%d D=%d K=%d S=%d Vb=%d not covered", mode,
                N, E, D, K, S, Vb);
  int nwg = encoder_workgroups(n_ions, B, workgroups, N, E, mode);
  if (info_in) {  // run half: the plan's geometry is authoritative, and must be the geometry of this call
    const int32_t* v = info_in->v;
    REQ(v[0] == kInfoMagic, "plan info was not filled by impnn_encoder_plan");
    if (v[1] != mode_class(mode) || v[2] != n_ions || v[3] != B || v[4] != N || v[5] != E || v[6] != S || v[7] != Vb ||
        v[9] != D || v[10] != K || v[11] != record_kind(mode, D))
      return fail(IMPNN_E_BADARG, "%s: the workspace was planned for another batch shape or record kind "
                  "(planned: kind %d/%d, n_ions %d, B %d, N %d, E %d, S %d, Vb %d, D %d, K %d)", fn, v[1], v[11], v[2],
                  v[3], v[4], v[5], v[6], v[7], v[9], v[10]);
    nwg = v[8];
  }
  if (info_out) {
    int32_t* v = info_out->v;
    for (int i = 0; i < 12; ++i) v[i] = 0;
    v[0] = kInfoMagic; v[1] = mode_class(mode); v[2] = n_ions; v[3] = B; v[4] = N; v[5] = E; v[6] = S; v[7] = Vb;
    v[8] = nwg; v[9] = D; v[10] = K; v[11] = record_kind(mode, D);
  }
  if (B == 0) return IMPNN_OK;
  EncoderArgs a{};
  a.n_ions = n_ions;
  a.mode = mode;
  a.phases = phases;
  a.nwg = nwg;
  for (int g = 0; g < n_ions; ++g) {
    const bool have_w = !running || S == 0 || (prepared && prepared[g]) || (weights && weights[g]);
    REQ(atom_ids[g] && have_w, "null per-ion pointer");
    REQ(!planning || E == 0 || (bond_ids[g] && conn[g]), "null per-ion pointer");
    REQ(!running || pooled[g], "null per-ion pointer");
    a.atom_ids[g] = atom_ids[g];
    a.bond_ids[g] = bond_ids ? bond_ids[g] : nullptr;
    a.conn[g] = conn ? conn[g] : nullptr;
    a.weights[g] = weights ? weights[g] : nullptr;
    a.prepared[g] = prepared ? prepared[g] : nullptr;
    a.pooled[g] = pooled ? pooled[g] : nullptr;
  }
#undef REQ
  a.atom_table = atom_table;
  a.bond_table = bond_table;
  a.Va = Va; a.Vb = Vb; a.B = B; a.N = N; a.E = E; a.D = D; a.K = K; a.S = S;
  a.ln_eps = ln_eps;
  a.workspace = workspace;
  a.workspace_bytes = workspace_bytes;
  const size_t need = encoder_fused_workspace_bytes(mode, n_ions, B, N, E, D, S, Vb, nwg);
  if (need > 0 && (!workspace || workspace_bytes < need))
    return fail(IMPNN_E_WORKSPACE, "%s: workspace %zu < %zu bytes", fn, workspace_bytes, need);
  return launch_encoder_fused(a, as_stream(stream));
}

int impnn_encoder_fused(int32_t n_ions, const int32_t* const* atom_ids, const int32_t* const* bond_ids,
                        const int32_t* const* conn, const float* atom_table, int32_t Va,
                        const float* bond_table, int32_t Vb, const float* const* weights, int32_t mode,
                        float* const* pooled, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K, int32_t S,
                        float ln_eps, int32_t workgroups, void* workspace, size_t workspace_bytes,
                        impnn_stream_t stream) {
  return encoder_common(__func__, n_ions, atom_ids, bond_ids, conn, atom_table, Va, bond_table, Vb, weights, nullptr,
                        mode, pooled, B, N, E, D, K, S, ln_eps, workgroups, nullptr, nullptr, workspace,
                        workspace_bytes, stream);
}

int impnn_encoder_plan(int32_t n_ions, const int32_t* const* atom_ids, const int32_t* const* bond_ids,
                       const int32_t* const* conn, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K, int32_t S,
                       int32_t Va, int32_t Vb, int32_t mode, int32_t workgroups, void* workspace,
                       size_t workspace_bytes, impnn_stream_t stream, impnn_encoder_plan_info* info) {
  REQUIRE(info, "null plan info");
  return encoder_common(__func__, n_ions, atom_ids, bond_ids, conn, nullptr, Va, nullptr, Vb, nullptr, nullptr, mode,
                        nullptr, B, N, E, D, K, S, 0.f, workgroups, nullptr, info, workspace, workspace_bytes, stream,
                        1);
}

int impnn_encoder_run(int32_t n_ions, const int32_t* const* atom_ids, const float* atom_table, int32_t Va,
                      const float* bond_table, int32_t Vb, const void* const* prepared, int32_t mode,
                      float* const* pooled, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K, int32_t S,
                      float ln_eps, const impnn_encoder_plan_info* info, void* workspace, size_t workspace_bytes,
                      impnn_stream_t stream) {
  REQUIRE(info, "null plan info");
  return encoder_common(__func__, n_ions, atom_ids, nullptr, nullptr, atom_table, Va, bond_table, Vb, nullptr,
                        prepared, mode, pooled, B, N, E, D, K, S, ln_eps, 0, info, nullptr, workspace,
                        workspace_bytes, stream, 2);
}

size_t impnn_encoder_prepared_bytes(int32_t D, int32_t S, int32_t Vb, int32_t mode) {
  if (mode < 0 || mode > 3 || Vb <= 0 || D <= 0) return 0;
  if (!encoder_fused_supported(mode, 1, 0, D, 1, S, mode >= 2 ? Vb : 1)) return 0;
  return encoder_prepared_bytes(mode, D, S, Vb);
}

int impnn_encoder_prepare_weights(const float* weights, const float* bond_table, int32_t D, int32_t K, int32_t S,
                                  int32_t Vb, int32_t mode, void* prepared, size_t prepared_bytes,
                                  impnn_stream_t stream) {
  REQUIRE(D > 0 && K > 0 && S >= 0 && Vb > 0, "bad shape");
  REQUIRE(mode >= 0 && mode <= 3, "mode must be 0 (f32), 1 (f16x2), 2 (f32 typed) or 3 (f32x3 typed)");
  if (!encoder_fused_supported(mode, 1, 0, D, K, S, mode >= 2 ? Vb : 1))
    return fail(IMPNN_E_UNSUPPORTED, "encoder_prepare_weights: mode=%d D=%d K=%d Vb=%d not covered", mode, D, K, Vb);
  if (S == 0) return IMPNN_OK;
  REQUIRE(weights && prepared && (mode < 2 || bond_table), "null pointer");
  REQUIRE(aligned16(prepared), "prepared buffer must be 16B aligned");
  if (prepared_bytes < encoder_prepared_bytes(mode, D, S, Vb))
    return fail(IMPNN_E_WORKSPACE, "encoder_prepare_weights: buffer %zu < %zu bytes", prepared_bytes,
                encoder_prepared_bytes(mode, D, S, Vb));
  return launch_encoder_prepare(weights, bond_table, D, K, S, Vb, mode, prepared, as_stream(stream));
}

int impnn_encoder_fused_prepared(int32_t n_ions, const int32_t* const* atom_ids, const int32_t* const* bond_ids,
                                 const int32_t* const* conn, const float* atom_table, int32_t Va,
                                 const float* bond_table, int32_t Vb, const void* const* prepared, int32_t mode,
                                 float* const* pooled, int32_t B, int32_t N, int32_t E, int32_t D, int32_t K,
                                 int32_t S, float ln_eps, int32_t workgroups, void* workspace, size_t workspace_bytes,
                                 impnn_stream_t stream) {
  return encoder_common(__func__, n_ions, atom_ids, bond_ids, conn, atom_table, Va, bond_table, Vb, nullptr, prepared,
                        mode, pooled, B, N, E, D, K, S, ln_eps, workgroups, nullptr, nullptr, workspace,
                        workspace_bytes, stream);
}

int64_t impnn_model_head_floats(int32_t kind, int32_t D, int32_t F, int32_t Mx) {
  if (D <= 0 || F <= 0 || Mx <= 0 || (kind != 0 && kind != 1)) return -1;
  const int64_t common = 2 * ((int64_t)D * F + F) + 2 * ((int64_t)F * Mx + Mx);
  return common + (kind == 0 ? (int64_t)Mx * 3 + 3 : (int64_t)Mx * F + F + F + 1);
}

int impnn_model_head(int32_t kind, const float* pooled_cat, const float* pooled_an, const float* temperature,
                     const float* head_weights, float* out, int32_t B, int32_t D, int32_t F, int32_t Mx,
                     impnn_stream_t stream) {
  REQUIRE(kind == 0 || kind == 1, "kind must be 0 (viscosity) or 1 (melting point)");
  REQUIRE(B >= 0 && D > 0 && F > 0 && Mx > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(pooled_cat && pooled_an && head_weights && out && (kind == 1 || temperature), "null pointer");
  return launch_model_head(kind, pooled_cat, pooled_an, temperature, head_weights, out, B, D, F, Mx, as_stream(stream));
}

int impnn_model_head_tensors(int32_t kind, const float* pooled_cat, const float* pooled_an, const float* temperature,
                             const float* const* weights, float* out, int32_t B, int32_t D, int32_t F, int32_t Mx,
                             impnn_stream_t stream) {
  REQUIRE(kind == 0 || kind == 1, "kind must be 0 (viscosity) or 1 (melting point)");
  REQUIRE(B >= 0 && D > 0 && F > 0 && Mx > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(pooled_cat && pooled_an && weights && out && (kind == 1 || temperature), "null pointer");
  return launch_model_head_tensors(kind, pooled_cat, pooled_an, temperature, weights, out, B, D, F, Mx,
                                   as_stream(stream));
}

int impnn_model_head_bwd(int32_t kind, const float* pooled_cat, const float* pooled_an, const float* temperature,
                         const float* const* weights, const float* dout, float* dpooled_cat, float* dpooled_an,
                         float* const* dweights, int32_t B, int32_t D, int32_t F, int32_t Mx, impnn_stream_t stream) {
  REQUIRE(kind == 0 || kind == 1, "kind must be 0 (viscosity) or 1 (melting point)");
  REQUIRE(B >= 0 && D > 0 && F > 0 && Mx > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(pooled_cat && pooled_an && weights && dout && dpooled_cat && dpooled_an && dweights &&
              (kind == 1 || temperature),
          "null pointer");
  return launch_model_head_bwd(kind, pooled_cat, pooled_an, temperature, weights, dout, dpooled_cat, dpooled_an,
                               dweights, B, D, F, Mx, as_stream(stream));
}

int64_t impnn_model_head_loss_workspace_floats(int32_t B) { return B > 0 ? model_head_loss_workspace_floats(B) : 4; }

int impnn_model_head_loss(int32_t kind, const float* pooled_cat, const float* pooled_an, const float* temperature,
                          const float* const* weights, const float* l2, const float* y, float* pred, float* loss,
                          float* workspace, int64_t workspace_floats, int32_t B, int32_t D, int32_t F, int32_t Mx,
                          impnn_stream_t stream) {
  REQUIRE(kind == 0 || kind == 1, "kind must be 0 (viscosity) or 1 (melting point)");
  REQUIRE(B > 0 && D > 0 && F > 0 && Mx > 0, "bad shape");
  REQUIRE(pooled_cat && pooled_an && weights && l2 && y && loss && workspace && (kind == 1 || temperature),
          "null pointer");
  if (workspace_floats < impnn_model_head_loss_workspace_floats(B))
    return fail(IMPNN_E_WORKSPACE, "model_head_loss: workspace of %lld floats is too small", (long long)workspace_floats);
  return launch_model_head_tensors(kind, pooled_cat, pooled_an, temperature, weights, pred, B, D, F, Mx,
                                   as_stream(stream), l2, y, loss, workspace);
}

int impnn_model_head_loss_bwd(int32_t kind, const float* pooled_cat, const float* pooled_an, const float* temperature,
                              const float* const* weights, const float* l2, const float* y, const float* dloss,
                              float* dpooled_cat, float* dpooled_an, float* const* dweights, int32_t B, int32_t D,
                              int32_t F, int32_t Mx, impnn_stream_t stream) {
  REQUIRE(kind == 0 || kind == 1, "kind must be 0 (viscosity) or 1 (melting point)");
  REQUIRE(B > 0 && D > 0 && F > 0 && Mx > 0, "bad shape");
  REQUIRE(pooled_cat && pooled_an && weights && l2 && y && dloss && dpooled_cat && dpooled_an && dweights &&
              (kind == 1 || temperature),
          "null pointer");
  return launch_model_head_bwd(kind, pooled_cat, pooled_an, temperature, weights, nullptr, dpooled_cat, dpooled_an,
                               dweights, B, D, F, Mx, as_stream(stream), l2, y, dloss);
}

int impnn_gather_rows(int32_t n_tensors, const void* const* src, void* const* dst, const int64_t* row_bytes,
                      const int64_t* rows, int32_t n_rows, impnn_stream_t stream) {
  REQUIRE(n_tensors >= 0 && n_rows >= 0, "bad shape");
  if (n_tensors == 0 || n_rows == 0) return IMPNN_OK;
  REQUIRE(src && dst && row_bytes && rows, "null pointer");
  return launch_gather_rows(n_tensors, src, dst, row_bytes, rows, n_rows, as_stream(stream));
}

int impnn_profile_enable(int32_t capacity) {
  REQUIRE(capacity > 0 && capacity <= (1 << 20), "capacity out of range");
  impnn_profile_disable();
  g_prof.start.resize(capacity);
  g_prof.stop.resize(capacity);
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&g_prof.start[i]) != hipSuccess || hipEventCreate(&g_prof.stop[i]) != hipSuccess) {
      g_prof.start.resize(i);
      g_prof.stop.resize(i);
      impnn_profile_disable();
      return fail(IMPNN_E_LAUNCH, "impnn_profile_enable: hipEventCreate failed");
    }
  }
  g_prof.used = 0;
  g_prof.open = false;
  g_prof.enabled = true;
  return IMPNN_OK;
}

int impnn_profile_collect(float* ms_out, int32_t max_n, int32_t* n_out) {
  REQUIRE(n_out && (ms_out || max_n == 0) && max_n >= 0, "bad arguments");
  int n = g_prof.used < max_n ? g_prof.used : max_n;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess ||
        hipEventElapsedTime(&ms_out[i], g_prof.start[i], g_prof.stop[i]) != hipSuccess)
      return fail(IMPNN_E_LAUNCH, "impnn_profile_collect: event %d not readable", i);
  }
  *n_out = n;
  g_prof.used = 0;
  g_prof.open = false;
  return IMPNN_OK;
}

int impnn_profile_disable(void) {
  for (auto e : g_prof.start) (void)hipEventDestroy(e);
  for (auto e : g_prof.stop) (void)hipEventDestroy(e);
  g_prof.start.clear();
  g_prof.stop.clear();
  g_prof.used = 0;
  g_prof.enabled = false;
  g_prof.open = false;
  return IMPNN_OK;
}

int impnn_debug_set_stamp_buffer(void* device_buffer, size_t bytes) {
  g_stamp_ptr = device_buffer;
  g_stamp_bytes = device_buffer ? bytes : 0;
  return IMPNN_OK;
}

int impnn_embed_gather_bwd(const int32_t* ids, const float* dout, float* dtable, int64_t rows, int32_t vocab,
                           int32_t dim, impnn_stream_t stream) {
  REQUIRE(rows >= 0 && vocab > 0 && dim > 0, "bad shape");
  if (rows == 0) return IMPNN_OK;
  REQUIRE(ids && dout && dtable, "null pointer");
  return launch_embed_gather_bwd(ids, dout, dtable, rows, vocab, dim, as_stream(stream));
}

int impnn_reduce_scatter_bwd(const float* dagg, const int32_t* tgt, int32_t tgt_stride, float* dmessages, int32_t B,
                             int32_t N, int32_t E, int32_t D, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && tgt_stride >= 1, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(dagg && tgt && dmessages, "null pointer");
  return launch_reduce_scatter_bwd(dagg, tgt, tgt_stride, dmessages, B, N, E, D, as_stream(stream));
}

int impnn_global_sum_pool_bwd(const float* dpooled, const int32_t* atom_ids, float* dh, int32_t B, int32_t N,
                              int32_t D, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && D > 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  REQUIRE(dpooled && atom_ids && dh, "null pointer");
  return launch_global_sum_pool_bwd(dpooled, atom_ids, dh, B, N, D, as_stream(stream));
}

int64_t impnn_bmm_message_typed_bwd_workspace_bytes(int32_t B, int32_t E, int32_t Vb) {
  if (B < 0 || E < 0 || Vb <= 0) return 0;
  return 4 * bmm_message_typed_bwd_workspace_ints(B, E, Vb);
}

int impnn_bmm_message_typed_sorted(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                   const float* type_mats, float* messages, void* workspace, int64_t workspace_bytes,
                                   int32_t B, int32_t N, int32_t E, int32_t D, int32_t Vb, int32_t sorted_ready,
                                   impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && Vb > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_ids && conn && type_mats && messages && workspace, "null pointer");
  if (workspace_bytes < impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    return fail(IMPNN_E_WORKSPACE, "bmm_message_typed_sorted: workspace of %lld bytes is too small",
                (long long)workspace_bytes);
  return launch_bmm_message_typed_sorted(h, bond_ids, conn, type_mats, messages, static_cast<int32_t*>(workspace), B,
                                         N, E, D, Vb, sorted_ready & 3, as_stream(stream));
}

int impnn_bmm_message_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                const float* type_mats, const float* dmessages, float* dh, float* dtype_mats,
                                void* workspace, int64_t workspace_bytes, int32_t B, int32_t N, int32_t E, int32_t D,
                                int32_t Vb, int32_t sorted_ready, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && Vb > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_ids && conn && type_mats && dmessages && dh && dtype_mats && workspace, "null pointer");
  if (workspace_bytes < impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    return fail(IMPNN_E_WORKSPACE, "bmm_message_typed_bwd: workspace of %lld bytes is too small", (long long)workspace_bytes);
  return launch_bmm_message_typed_bwd(h, bond_ids, conn, type_mats, dmessages, dh, dtype_mats,
                                      static_cast<int32_t*>(workspace), B, N, E, D, Vb, sorted_ready != 0, 0,
                                      as_stream(stream));
}

int impnn_message_reduce_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                   const float* type_mats, const float* dagg, float* dh, float* dtype_mats,
                                   void* workspace, int64_t workspace_bytes, int32_t B, int32_t N, int32_t E, int32_t D,
                                   int32_t Vb, int32_t sorted_ready, impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && Vb > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_ids && conn && type_mats && dagg && dh && dtype_mats && workspace, "null pointer");
  if (workspace_bytes < impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    return fail(IMPNN_E_WORKSPACE, "message_reduce_typed_bwd: workspace of %lld bytes is too small",
                (long long)workspace_bytes);
  return launch_bmm_message_typed_bwd(h, bond_ids, conn, type_mats, dagg, dh, dtype_mats,
                                      static_cast<int32_t*>(workspace), B, N, E, D, Vb, sorted_ready != 0, 1,
                                      as_stream(stream));
}

int impnn_message_reduce_typed_bwd_scratch(const float* h, const int32_t* bond_ids, const int32_t* conn,
                                           const float* type_mats, const float* dagg, float* dh, float* dtype_mats,
                                           void* workspace, int64_t workspace_bytes, float* edge_scratch, int32_t B,
                                           int32_t N, int32_t E, int32_t D, int32_t Vb, int32_t sorted_ready,
                                           impnn_stream_t stream) {
  REQUIRE(B >= 0 && N > 0 && E >= 0 && D > 0 && Vb > 0, "bad shape");
  if (B == 0 || E == 0) return IMPNN_OK;
  REQUIRE(h && bond_ids && conn && type_mats && dagg && dh && dtype_mats && workspace && edge_scratch, "null pointer");
  if (workspace_bytes < impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    return fail(IMPNN_E_WORKSPACE, "message_reduce_typed_bwd_scratch: workspace of %lld bytes is too small",
                (long long)workspace_bytes);
  return launch_bmm_message_typed_bwd(h, bond_ids, conn, type_mats, dagg, dh, dtype_mats,
                                      static_cast<int32_t*>(workspace), B, N, E, D, Vb, sorted_ready != 0, 1,
                                      as_stream(stream), edge_scratch);
}

int impnn_bond_type_matrices_bwd(const float* bond_table, const float* W, const float* dtype_mats, float* dW,
                                 float* dbond_table, int32_t Vb, int32_t K, int32_t D, int32_t accumulate,
                                 impnn_stream_t stream) {
  REQUIRE(Vb > 0 && K > 0 && D > 0, "bad shape");
  REQUIRE(bond_table && W && dtype_mats && dW && dbond_table, "null pointer");
  return launch_bond_type_matrices_bwd(bond_table, W, dtype_mats, dW, dbond_table, Vb, K, D, accumulate != 0,
                                       as_stream(stream));
}

int impnn_bond_type_matrices_multi(const float* bond_table, const float* const* W, float* const* type_mats, int32_t n,
                                   int32_t Vb, int32_t K, int32_t D, impnn_stream_t stream) {
  REQUIRE(n >= 0 && Vb > 0 && K > 0 && D > 0, "bad shape");
  if (n == 0) return IMPNN_OK;
  REQUIRE(bond_table && W && type_mats, "null pointer");
  return launch_bond_type_matrices_multi(bond_table, W, type_mats, n, Vb, K, D, as_stream(stream));
}

int impnn_bond_type_matrices_multi_bwd(const float* bond_table, const float* const* W, const float* const* dtype_mats,
                                       float* const* dW, float* dbond_table, int32_t n, int32_t Vb, int32_t K,
                                       int32_t D, int32_t accumulate, impnn_stream_t stream) {
  REQUIRE(n >= 0 && Vb > 0 && K > 0 && D > 0, "bad shape");
  if (n == 0) return IMPNN_OK;
  REQUIRE(bond_table && W && dtype_mats && dW && dbond_table, "null pointer");
  return launch_bond_type_matrices_multi_bwd(bond_table, W, dtype_mats, dW, dbond_table, n, Vb, K, D, accumulate != 0,
                                             as_stream(stream));
}

int64_t impnn_bond_type_matrices_multi_bwd_workspace_floats(int32_t n, int32_t Vb, int32_t K, int32_t D) {
  if (n <= 0 || Vb <= 0 || K <= 0 || D <= 0) return 0;
  return bond_type_matrices_multi_bwd_workspace(n, Vb, K, D);
}

int impnn_bond_type_matrices_multi_bwd_ws(const float* bond_table, const float* const* W,
                                          const float* const* dtype_mats, float* const* dW, float* dbond_table,
                                          int32_t n, int32_t Vb, int32_t K, int32_t D, int32_t accumulate,
                                          float* workspace, int64_t workspace_floats, impnn_stream_t stream) {
  REQUIRE(n >= 0 && Vb > 0 && K > 0 && D > 0, "bad shape");
  if (n == 0) return IMPNN_OK;
  REQUIRE(bond_table && W && dtype_mats && dW && dbond_table && workspace, "null pointer");
  if (workspace_floats < impnn_bond_type_matrices_multi_bwd_workspace_floats(n, Vb, K, D))
    return fail(IMPNN_E_WORKSPACE, "bond_type_matrices_multi_bwd_ws: workspace of %lld floats is too small",
                (long long)workspace_floats);
  return launch_bond_type_matrices_multi_bwd(bond_table, W, dtype_mats, dW, dbond_table, n, Vb, K, D, accumulate != 0,
                                             as_stream(stream), workspace);
}

int64_t impnn_gated_update_param_floats(int32_t D) { return D > 0 ? gated_update_param_floats(D) : 0; }

int64_t impnn_gated_update_bwd_workspace_floats(int64_t rows, int32_t D) {
  if (rows < 0 || D <= 0) return 0;
  return gated_update_bwd_workspace(rows, D);
}

int impnn_gated_update_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                           const float* br, const float* Wh, const float* bh, const float* gamma, float ln_eps,
                           const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                           int64_t workspace_floats, int64_t rows, int32_t D, int32_t accumulate,
                           impnn_stream_t stream) {
  REQUIRE(rows >= 0 && D > 0 && D <= 256 && 256 % D == 0, "atom_dim must divide 256");
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && dout && dh && dagg && dparams && workspace,
          "null pointer");
  if (workspace_floats < impnn_gated_update_bwd_workspace_floats(rows, D))
    return fail(IMPNN_E_WORKSPACE, "gated_update_bwd: workspace of %lld floats is too small", (long long)workspace_floats);
  return launch_gated_update_bwd(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, ln_eps, dout, dh, dagg, dparams, workspace,
                                 rows, D, accumulate != 0, as_stream(stream));
}

int64_t impnn_gated_update_rows_bwd_workspace_floats(int64_t max_rows, int32_t D) {
  if (max_rows < 0 || (D != 64 && D != 128)) return 0;
  return gated_update_bwd_workspace(max_rows, D, true);
}

int impnn_gated_update_rows_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                                const float* br, const float* Wh, const float* bh, const float* gamma, float ln_eps,
                                const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                                int64_t workspace_floats, const int32_t* row_index, const int32_t* n_rows,
                                int64_t max_rows, int32_t D, int32_t accumulate, impnn_stream_t stream) {
  REQUIRE(max_rows >= 0, "bad shape");
  if (D != 64 && D != 128)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd: atom_dim %d (the row-list form covers 64 and 128)", D);
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && dout && dh && dagg && dparams && workspace &&
          row_index && n_rows, "null pointer");
  if (workspace_floats < impnn_gated_update_rows_bwd_workspace_floats(max_rows, D))
    return fail(IMPNN_E_WORKSPACE, "gated_update_rows_bwd: workspace of %lld floats is too small", (long long)workspace_floats);
  if (max_rows == 0) return IMPNN_OK;
  return launch_gated_update_bwd(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, ln_eps, dout, dh, dagg, dparams, workspace,
                                 max_rows, D, accumulate != 0, as_stream(stream), row_index, n_rows);
}

int64_t impnn_gated_update_rows_saved_floats(int64_t max_rows, int32_t D) {
  if (max_rows < 0 || (D != 32 && D != 64 && D != 128)) return 0;
  return max_rows * 4 * D;
}

int impnn_gated_update_rows_train(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                                  const float* br, const float* Wh, const float* bh, const float* gamma,
                                  const float* beta, float ln_eps, float* out, const int32_t* row_index,
                                  const int32_t* n_rows, int64_t max_rows, int32_t D, float* saved,
                                  impnn_stream_t stream) {
  REQUIRE(max_rows >= 0, "bad shape");
  if (D != 32 && D != 64 && D != 128)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_train: atom_dim %d (the saving forward covers 32, 64 and 128)", D);
  if (max_rows == 0) return IMPNN_OK;
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && beta && out && saved, "null pointer");
  REQUIRE((row_index != nullptr) == (n_rows != nullptr), "row_index and n_rows: both or neither");
  REQUIRE((reinterpret_cast<uintptr_t>(saved) & 15u) == 0, "saved must be 16-byte aligned");
  REQUIRE(ln_eps >= 0.f, "ln_eps must be >= 0");
  return launch_gated_update(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, ln_eps, out, max_rows, D, as_stream(stream),
                             row_index, n_rows, saved);
}

int impnn_gated_update_rows_bwd_saved(const float* h, const float* agg, const float* Wz, const float* bz,
                                      const float* Wr, const float* br, const float* Wh, const float* bh,
                                      const float* gamma, float ln_eps, const float* dout, float* dh, float* dagg,
                                      float* dparams, float* workspace, int64_t workspace_floats,
                                      const int32_t* row_index, const int32_t* n_rows, int64_t max_rows, int32_t D,
                                      int32_t accumulate, float* saved, impnn_stream_t stream) {
  REQUIRE(max_rows >= 0, "bad shape");
  if (D != 32 && D != 64 && D != 128)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd_saved: atom_dim %d (covers 32, 64 and 128)", D);
  REQUIRE(h && agg && Wz && bz && Wr && br && Wh && bh && gamma && dout && dh && dagg && dparams && workspace && saved,
          "null pointer");
  REQUIRE((row_index != nullptr) == (n_rows != nullptr), "row_index and n_rows: both or neither");
  if (D == 32 && row_index)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd_saved: atom_dim 32 takes no row list");
  if (workspace_floats < (D == 32 ? impnn_gated_update_bwd_workspace_floats(max_rows, D)
                                  : impnn_gated_update_rows_bwd_workspace_floats(max_rows, D)))
    return fail(IMPNN_E_WORKSPACE, "gated_update_rows_bwd_saved: workspace of %lld floats is too small",
                (long long)workspace_floats);
  if (max_rows == 0) return IMPNN_OK;
  return launch_gated_update_bwd(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, ln_eps, dout, dh, dagg, dparams, workspace,
                                 max_rows, D, accumulate != 0, as_stream(stream), row_index, n_rows, saved);
}

int impnn_adam_clipnorm_step(const void* var_table, const int64_t* sizes, int32_t n_vars, int64_t step, float lr,
                             float beta1, float beta2, float eps, float clipnorm, impnn_stream_t stream) {
  REQUIRE(n_vars >= 0 && step >= 1, "bad arguments (step counts from 1)");
  if (n_vars == 0) return IMPNN_OK;
  REQUIRE(var_table && sizes, "null pointer");
  return launch_adam_clipnorm(var_table, sizes, n_vars, step, nullptr, lr, beta1, beta2, eps, clipnorm,
                              as_stream(stream));
}

int impnn_adam_clipnorm_step_counted(const void* var_table, const int64_t* sizes, int32_t n_vars, int64_t* step_counter,
                                     float lr, float beta1, float beta2, float eps, float clipnorm,
                                     impnn_stream_t stream) {
  REQUIRE(n_vars >= 0, "bad arguments");
  REQUIRE(var_table && sizes && step_counter, "null pointer");
  return launch_adam_clipnorm(var_table, sizes, n_vars, 0, step_counter, lr, beta1, beta2, eps, clipnorm,
                              as_stream(stream));
}

int impnn_batch_assemble(int32_t n_ions, const int32_t* sample_idx, int32_t B, int32_t M,
                         const int32_t* const* atom_flat, const int32_t* const* atom_off,
                         const int32_t* const* edge_flat, const int32_t* const* bond_flat,
                         const int32_t* const* edge_off, int32_t id_shift, int32_t N, int32_t L,
                         int32_t* const* atom_ids, int32_t* const* bond_ids, int32_t* const* conn,
                         const float* t_flat, float* t_out, impnn_stream_t stream) {
  REQUIRE(n_ions >= 1 && n_ions <= 2, "n_ions must be 1 or 2");
  REQUIRE(B >= 0 && M >= 1 && N >= 1 && L >= 0, "bad shape");
  REQUIRE(atom_flat && atom_off && edge_flat && bond_flat && edge_off && atom_ids && bond_ids && conn,
          "null pointer array");
  if (B == 0) return IMPNN_OK;
  REQUIRE(sample_idx, "null sample_idx");
  for (int g = 0; g < n_ions; ++g) {
    REQUIRE(atom_flat[g] && atom_off[g] && edge_off[g] && atom_ids[g], "null per-ion pointer");
    REQUIRE(L == 0 || (edge_flat[g] && bond_flat[g] && bond_ids[g] && conn[g]), "null per-ion edge pointer");
    REQUIRE((reinterpret_cast<uintptr_t>(edge_flat[g]) & 7u) == 0 && (reinterpret_cast<uintptr_t>(conn[g]) & 7u) == 0,
            "edge_flat / conn must be 8-byte aligned");
  }
  REQUIRE(!t_out || t_flat, "t_out without t_flat");
  return launch_batch_assemble(n_ions, sample_idx, B, M, atom_flat, atom_off, edge_flat, bond_flat, edge_off,
                               id_shift, N, L, atom_ids, bond_ids, conn, t_flat, t_out, as_stream(stream));
}

int impnn_validate_indices(const int32_t* conn, const int32_t* atom_ids, const int32_t* bond_ids,
                           int32_t* counts, int32_t B, int32_t N, int32_t E, int32_t Va, int32_t Vb,
                           impnn_stream_t stream) {
  REQUIRE(counts, "null pointer");
  REQUIRE(B >= 0 && N > 0 && E >= 0, "bad shape");
  if (B == 0) return IMPNN_OK;
  return launch_validate_indices(conn, atom_ids, bond_ids, counts, B, N, E, Va, Vb, as_stream(stream));
}

}  // extern "C"
