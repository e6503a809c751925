// Fused message-passing encoder for gfx950 (MI355X): encode() of train_viscosity.py:166-187 up to
// and including GlobalSumPool, for both ions in one launch, D = 32, K <= 8.
//
// Shape of the computation:
//   * molecules are cut into CHUNKS of <= 256 packed atom rows (padding atoms are not carried;
//     see "rows" below); one 1024-thread workgroup (16 waves, one 16-atom tile each) owns a chunk
//     for all S steps, its node state h lives in LDS (double-buffered), the weights of the current
//     (ion, step) live in LDS;
//   * two arithmetic modes for the GEMMs, same data flow:
//       mode 0 "f32":      v_mfma_f32_16x16x4_f32, exact f32 products.  On gfx950 this instruction
//                          runs at the VALU rate and does not overlap VALU work of the same SIMD
//                          (tools/ubench/mfma_valu_overlap.hip: 4.49 ms + 0.92 ms -> 5.32 ms);
//       mode 1 "f16x2":    every f32 operand x is split x*2^s = hi + lo into two fp16 numbers
//                          (hi = x*2^s rounded toward zero to 11 bits, lo = the next 11 bits), and
//                          a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_16x16x32_f16 with f32
//                          accumulation: 3 matrix-pipe instructions replace 8, product error
//                          ~2^-21, measured end-to-end error vs fp64 2-3e-7 (plain f32: 2e-7).
//                          Weights are split once per call into the LDS image, activations on the
//                          fly (2 VALU ops / value).  Requires |h|,|agg|,|G| < 4094 and
//                          |W| < 255 (fp16 range after scaling); the caller checks a static bound.
//   * atoms sit on the MFMA N dimension (lane & 15), features on M: every GEMM is computed
//     transposed, out^T = W^T * in^T, so an accumulator tile (feature = 4*(lane>>4)+reg) is
//     directly the B operand of the next GEMM - no LDS round trip between message, gates,
//     candidate and LayerNorm;
//   * message + Reduce (models/layers.py:100-117, 57-83) in "pull" form: each atom row walks its
//     in-edges in edge-slot order (CSR built once per chunk, deterministic), forming
//         G[k][j] = sum_{e -> atom} bond_table[bond_id_e][k] * h[src_e][j]
//     in registers, then agg^T = sum_k W_k * G_k on the matrix cores (128 MFMA / 16 atoms);
//   * GatedUpdate (models/layers.py:142-156): 96 MFMA / 16 atoms, sigmoid/tanh/LayerNorm on the
//     accumulator registers, LayerNorm row reduction = 8 in-lane adds + 2 cross-lane steps;
//   * GlobalSumPool (models/layers.py:161-164): segmented sum over the chunk's rows from LDS.
//
// rows: molecule b keeps rows [0, r_b), r_b = 1 + max(last n with atom_ids[b,n] > 0, largest atom
// index on a valid edge).  Rows >= r_b can never send (no valid edge names them) and are masked
// by the pool, so dropping them cannot change the output; the kept set is closed under
// "is a source of", which makes the skip exact, not approximate.
//
// Three launches per call: plan_stats (+ weight image prep), plan_scan, encoder_fused.
#include "common.h"

#ifndef IMPNN_OPT_PF_LATE
#define IMPNN_OPT_PF_LATE 1
#endif
#ifndef IMPNN_OPT_PSORT
#define IMPNN_OPT_PSORT 1
#endif

namespace impnn {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// mode 1 scaling (powers of two, exact): weights are stored as W*kSW, B operands as x*kSX
constexpr float kSW = 256.0f;
constexpr float kSX = 16.0f;
constexpr float kAcc = kSW * kSX;  // scale of every accumulator in mode 1

constexpr int kD = 32;
constexpr int kKMax = 8;
constexpr int kRCap = 256;   // packed rows per chunk
constexpr int kECap = 1024;  // valid edges per chunk (= 4 * kRCap, enforced through "virtual rows")
constexpr int kHS = 36;      // LDS row stride of h (floats): 16B aligned, conflict-free b128 tile writes
constexpr int kMsgRS = 36;   // row stride of the message-weight image
constexpr int kUpdRS = 68;   // row stride of the update-weight image (2D + 4)
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kTbCapFloats = 2048;  // bond table copy in LDS (Vb*K floats)

__host__ __device__ constexpr int img_msg_floats(int K) { return K * kD * kMsgRS; }
__host__ __device__ constexpr int img_upd_floats() { return 3 * kD * kUpdRS; }
__host__ __device__ constexpr int img_vec_floats() { return 5 * kD; }
__host__ __device__ constexpr int img_floats(int K) {
  return img_msg_floats(K) + img_upd_floats() + img_vec_floats();
}
// mode 1 image (halfs): per (k, T) / (gate, T, half) two 512-half blocks (hi, lo); a lane's 8 halfs of
// a block are contiguous, so the A operand of one MFMA is one conflict-free ds_read_b128.
__host__ __device__ constexpr int img16_msg_halfs(int K) { return K * 2 * 2 * 512; }
__host__ __device__ constexpr int img16_upd_halfs() { return 3 * 2 * 2 * 2 * 512; }
__host__ __device__ constexpr int img16_vec_float_off(int K) { return (img16_msg_halfs(K) + img16_upd_halfs()) / 2; }
// feature held by element j (0..7) of lane quarter q: the accumulator layout of a 16x16 MFMA tile pair
__host__ __device__ constexpr int feat_of(int q, int j) { return 16 * (j >> 2) + 4 * q + (j & 3); }
// Every image is stored (HBM workspace and LDS) in a slot of kImgSlot floats so that the
// register prefetch is kPf unconditional 16-byte loads per thread (no per-load branch / wait).
constexpr int kPf = 4;
constexpr int kImgSlot = kPf * kThreads * 4;  // 16384 floats = 64 KiB >= img_floats(8) = 15904
static_assert(kThreads == 4 * kRCap, "prologue maps 4 threads to a row");
static_assert(img_floats(kKMax) <= kImgSlot, "weight image does not fit its slot");
static_assert(img16_vec_float_off(kKMax) + img_vec_floats() <= kImgSlot, "split weight image does not fit");

// workspace layout (bytes, all 256-aligned sections)
struct Ws {
  size_t img_off, rows_off, vr_off, start_off, first_off, nchunks_off, desc_off, total;
  int ub;  // upper bound of chunks per ion
};

__host__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__host__ inline int vr_max_of(int N, int E) {
  int v = (E + 3) / 4;
  int m = N > v ? N : v;
  return m < 1 ? 1 : m;
}

__host__ inline Ws ws_layout(int n_ions, int B, int N, int E, int K, int S) {
  Ws w{};
  const int vrmax = vr_max_of(N, E);
  const int win = kRCap - vrmax + 1;
  w.ub = (int)(((int64_t)B * vrmax) / win) + 1;
  size_t off = 0;
  w.img_off = off;
  off = align_up(off + (size_t)n_ions * (S > 0 ? S : 1) * kImgSlot * sizeof(float), 256);
  w.rows_off = off;
  off = align_up(off + (size_t)n_ions * B * sizeof(int32_t), 256);
  w.vr_off = off;
  off = align_up(off + (size_t)n_ions * B * sizeof(int32_t), 256);
  w.start_off = off;
  off = align_up(off + (size_t)n_ions * (B + 1) * sizeof(int32_t), 256);
  w.first_off = off;
  off = align_up(off + (size_t)n_ions * (w.ub + 2) * sizeof(int32_t), 256);
  w.nchunks_off = off;
  off = align_up(off + 2 * sizeof(int32_t), 256);
  w.desc_off = off;
  off = align_up(off + (size_t)n_ions * w.ub * 4 * sizeof(int32_t), 256);
  w.total = off;
  return w;
}

struct PlanParams {
  const int32_t* atom_ids[2];
  const int32_t* bond_ids[2];
  const int32_t* conn[2];
  int32_t* rows;   // [n_ions][B]
  int32_t* vr;     // [n_ions][B]
  int32_t* start;  // [n_ions][B+1]
  int32_t* first;  // [n_ions][ub+2]
  int32_t* nchunks;  // [2]
  int32_t* desc;     // [n_ions][ub][4] = {first molecule, molecules, first virtual row, rows}; zeros if empty
  int n_ions, B, N, E, K, S, Vb, win, ub;
};

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int t = __shfl_xor(v, o);
    v = v > t ? v : t;
  }
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ bool edge_valid(int s, int t, int bid, int N, int Vb) {
  // models/layers.py:114-115 (src>0 & tgt>0); out-of-range indices behave as padding (impnn.h)
  return s > 0 && t > 0 && s < N && t < N && (unsigned)bid < (unsigned)Vb;
}

// -----------------------------------------------------------------------------------------
// plan_stats: one wave per (ion, molecule): kept rows r_b, valid edges v_b, virtual rows
// vr_b = max(1, r_b, ceil(v_b/4)).  Extra blocks convert the canonical packed step weights into
// the LDS image the encoder copies verbatim (message rows padded to 36, gate kernels transposed).
// -----------------------------------------------------------------------------------------
__global__ void plan_stats_kernel(PlanParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= (int64_t)p.n_ions * p.B) return;
  const int g = (int)(item / p.B);
  const int b = (int)(item - (int64_t)g * p.B);
  const int32_t* ids = p.atom_ids[g] + (int64_t)b * p.N;
  const int32_t* cn = p.conn[g] + (int64_t)b * p.E * 2;
  const int32_t* bd = p.bond_ids[g] + (int64_t)b * p.E;
  // wave-level reductions through ballots (scalar unit), no cross-lane data movement
  int rmax = 0, cnt = 0;
  for (int n0 = 0; n0 < p.N; n0 += 64) {
    const int n = n0 + lane;
    const unsigned long long hit = __ballot(n < p.N && ids[n] > 0);
    if (hit) rmax = n0 + 64 - __builtin_clzll(hit);  // 1 + highest n with ids[n] > 0
  }
  int emax = 0;  // largest atom index on a valid edge (lane-local)
  for (int e0 = 0; e0 < p.E; e0 += 64) {
    const int e = e0 + lane;
    bool ok = false;
    if (e < p.E) {
      const int2 st = *reinterpret_cast<const int2*>(cn + 2 * e);
      ok = edge_valid(st.x, st.y, bd[e], p.N, p.Vb);
      if (ok) {
        const int m = st.x > st.y ? st.x : st.y;
        emax = emax > m ? emax : m;
      }
    }
    cnt += __builtin_popcountll(__ballot(ok));
  }
  if (cnt > 0) {  // wave max of emax, bit by bit from the top (indices < 65536)
    bool alive = true;
    int res = 0;
#pragma unroll
    for (int bit = 15; bit >= 0; --bit) {
      const bool one = alive && ((emax >> bit) & 1);
      if (__ballot(one)) {
        res |= 1 << bit;
        alive = one;
      }
    }
    rmax = rmax > res + 1 ? rmax : res + 1;
  }
  if (lane == 0) {
    int vr = (cnt + 3) >> 2;
    vr = vr > rmax ? vr : rmax;
    vr = vr < 1 ? 1 : vr;
    p.rows[item] = rmax;
    p.vr[item] = vr;
  }
}

// -----------------------------------------------------------------------------------------
// weight_image: canonical packed step weights -> the image the encoder copies verbatim into LDS
// (mode 0: f32, message rows padded to 36, gate kernels transposed; mode 1: fp16 hi/lo blocks in
// MFMA A-operand order).  grid = (slices, steps); depends on the weights only, so callers that
// keep weights fixed run it once (impnn_encoder_prepare_weights).
// -----------------------------------------------------------------------------------------
struct ImageParams {
  const float* weights;  // S steps, canonical layout
  float* img;            // S slots of kImgSlot floats
  int K, mode;
  int64_t step_floats;
};

__global__ void weight_image_kernel(ImageParams p) {
  {
    const int s = blockIdx.y;
    const int t_begin = blockIdx.x * blockDim.x + threadIdx.x, t_stride = gridDim.x * blockDim.x;
    const float* w = p.weights + (int64_t)s * p.step_floats;
    const int K = p.K;
    const float* W = w;                                 // (K,32,32)
    const float* Wz = W + (int64_t)K * kD * kD;          // (64,32)
    const float* bz = Wz + 2 * kD * kD;
    const float* Wr = bz + kD;
    const float* br = Wr + 2 * kD * kD;
    const float* Wh = br + kD;
    const float* bh = Wh + 2 * kD * kD;
    const float* gamma = bh + kD;
    const float* beta = gamma + kD;
    float* img = p.img + (int64_t)s * kImgSlot;
    if (p.mode == 1) {
      _Float16* hi_lo = reinterpret_cast<_Float16*>(img);
      const int nm = img16_msg_halfs(K), nu = img16_upd_halfs();
      for (int t = t_begin; t < nm + nu; t += t_stride) {
        // t = ((blk * 2 + part) * 64 + lane) * 8 + j
        const int j = t & 7, ln = (t >> 3) & 63, part = (t >> 9) & 1;
        const int q = ln >> 4, i = ln & 15, f = feat_of(q, j);
        float wv;
        if (t < nm) {
          const int blk = t >> 10;  // k*2 + T
          const int k = blk >> 1, T = blk & 1;
          wv = W[((int64_t)k * kD + 16 * T + i) * kD + f];
        } else {
          const int blk = (t - nm) >> 10;  // (gate*2 + T)*2 + half
          const int half = blk & 1, T = (blk >> 1) & 1, gate = blk >> 2;
          const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
          wv = Wg[(int64_t)(half * kD + f) * kD + 16 * T + i];
        }
        wv *= kSW;
        const _Float16 hi = __builtin_amdgcn_cvt_pkrtz(wv, 0.f)[0];
        const _Float16 lo = __builtin_amdgcn_cvt_pkrtz(wv - (float)hi, 0.f)[0];
        hi_lo[t] = part == 0 ? hi : lo;
      }
      float* vec = img + img16_vec_float_off(K);
      for (int t = t_begin; t < img_vec_floats(); t += t_stride) {
        const int v = t / kD, i = t - v * kD;
        const float* src = v == 0 ? bz : v == 1 ? br : v == 2 ? bh : v == 3 ? gamma : beta;
        vec[t] = v < 3 ? src[i] * kAcc : src[i];  // biases seed the (scaled) accumulators
      }
      for (int t = img16_vec_float_off(K) + img_vec_floats() + t_begin; t < kImgSlot; t += t_stride) img[t] = 0.f;
      return;
    }
    const int nmsg = img_msg_floats(K), nupd = img_upd_floats();
    for (int t = t_begin; t < nmsg; t += t_stride) {
      const int row = t / kMsgRS, j = t - row * kMsgRS;  // row = k*32 + i_out
      img[t] = j < kD ? W[(int64_t)row * kD + j] : 0.f;
    }
    for (int t = t_begin; t < nupd; t += t_stride) {
      const int row = t / kUpdRS, jj = t - row * kUpdRS;  // row = gate*32 + i_out
      const int gate = row / kD, io = row - gate * kD;
      const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
      img[nmsg + t] = jj < 2 * kD ? Wg[(int64_t)jj * kD + io] : 0.f;
    }
    for (int t = t_begin; t < img_vec_floats(); t += t_stride) {
      const int v = t / kD, i = t - v * kD;
      const float* src = v == 0 ? bz : v == 1 ? br : v == 2 ? bh : v == 3 ? gamma : beta;
      img[nmsg + nupd + t] = src[i];
    }
    for (int t = img_floats(K) + t_begin; t < kImgSlot; t += t_stride) img[t] = 0.f;
  }
}

// -----------------------------------------------------------------------------------------
// plan_scan: one 1024-thread workgroup per ion: start[b] = exclusive prefix of vr, chunk of
// molecule b = start[b] / win (win = 256 - vr_max + 1 guarantees <= 256 rows per chunk), and
// first[c] = first molecule of chunk c (empty chunks get an empty range).
// -----------------------------------------------------------------------------------------
__global__ void plan_scan_kernel(PlanParams p) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int g = blockIdx.x;
  const int B = p.B;
  const int32_t* vr = p.vr + (int64_t)g * B;
  int32_t* start = p.start + (int64_t)g * (B + 1);
  int32_t* first = p.first + (int64_t)g * (p.ub + 2);
  const int T = blockDim.x;
  const int per = (B + T - 1) / T;
  const int b0 = threadIdx.x * per;
  const int b1 = (b0 + per) < B ? (b0 + per) : B;
  int local = 0;
  for (int b = b0; b < b1; ++b) local += vr[b];
  // block exclusive scan of `local`
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int i = 0; i < (T >> 6); ++i) {
      int t = wsum[i];
      wsum[i] = c;
      c += t;
    }
    carry_s = c;
  }
  __syncthreads();
  int run = wsum[wv] + incl - local;
  int prev_chunk = (b0 == 0 || b0 >= B) ? -1 : 0;
  // chunk of the molecule just before b0 (needed to detect a chunk boundary at b0)
  if (b0 > 0 && b0 < B) prev_chunk = (run - vr[b0 - 1]) / p.win;
  for (int b = b0; b < b1; ++b) {
    start[b] = run;
    const int c = run / p.win;
    if (c != prev_chunk) {
      for (int cc = prev_chunk + 1; cc <= c; ++cc) first[cc] = b;
      prev_chunk = c;
    }
    run += vr[b];
  }
  const int total = carry_s;
  const int last_chunk = B > 0 ? (total - vr[B - 1]) / p.win : -1;
  if (threadIdx.x == 0) {
    start[B] = total;
    p.nchunks[g] = last_chunk + 1;
    first[last_chunk + 1] = B;
  }
  // chunk descriptors: one 16-byte record per launched workgroup (the encoder's first load)
  __threadfence_block();
  __syncthreads();
  int4* desc = reinterpret_cast<int4*>(p.desc) + (int64_t)g * p.ub;
  for (int c = threadIdx.x; c < p.ub; c += T) {
    int4 d = make_int4(0, 0, 0, 0);
    if (c <= last_chunk) {
      const int f0 = first[c], f1 = first[c + 1];
      if (f1 > f0) {
        const int s0 = start[f0];
        d = make_int4(f0, f1 - f0, s0, start[f1] - s0);
      }
    }
    desc[c] = d;
  }
}

// -----------------------------------------------------------------------------------------
// the encoder
// -----------------------------------------------------------------------------------------
struct EncParams {
  const int32_t* atom_ids[2];
  const int32_t* bond_ids[2];
  const int32_t* conn[2];
  float* pooled[2];
  const float* atom_table;
  const float* bond_table;
  const float* img[2];  // per ion: S weight images (kImgSlot floats each)
  const int32_t* rows;
  const int32_t* start;
  const int32_t* first;
  const int32_t* nchunks;
  const int32_t* desc;
  int n_ions, B, N, E, K, S, Va, Vb, ub;
  float ln_eps;
  unsigned long long* stamps;  // diagnostics only (impnn_debug_set_stamp_buffer): 16 words per workgroup
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * x));
}
// the same on an accumulator that carries the mode-1 scale kAcc (the division is folded into the constant)
__device__ __forceinline__ float fast_sigmoid_scaled(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((-1.44269504088896f / kAcc) * x));
}
__device__ __forceinline__ float fast_tanh_scaled(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.88539008177793f / kAcc) * x));
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr int kDegBins = 18;  // in-degree 0..15, ">= 16", and "row beyond the chunk" (placed last)

// LDS carve (floats unless noted)
struct Lds {
  float* wimg;       // kImgSlot
  float* hbuf0;      // kRCap*kHS
  float* hbuf1;      // kRCap*kHS
  float* tb;         // kTbCapFloats (row stride 8)
  uint32_t* ent;     // kECap : in-edge lists, edge-slot order
  uint32_t* ent2;    // kECap : fill order (before the per-row sort)
  int32_t* rowptr;   // kRCap+4 : CSR over PLACED rows
  int32_t* cursor;   // kRCap   : fill cursors (placed rows)
  int32_t* cnt;      // kRCap   : in-degree per LOGICAL row
  int32_t* place;    // kRCap   : logical row -> placed row (rows are placed by descending in-degree)
  int32_t* rowinfo;  // kRCap   : per logical row (mol_local<<16 | [atom id > 0]<<15 | n) or -1
  int32_t* moloff;   // kRCap+4
  int32_t* molrows;  // kRCap
  int32_t* tilemax;  // 16      : largest in-degree inside each 16-row tile
  int32_t* bins;     // 2*kDegBins (+pad): histogram / bin cursors
  int32_t* scratch;  // 32
};

__host__ __device__ inline size_t lds_bytes(int K) {
  (void)K;
  return sizeof(float) * ((size_t)kImgSlot + 2 * kRCap * kHS + kTbCapFloats) + sizeof(uint32_t) * 2 * kECap +
         sizeof(int32_t) * ((kRCap + 4) + 5 * kRCap + (kRCap + 4) + 16 + 48 + 32);
}

struct H8 {
  half8 hi, lo;
};
// v (already scaled into fp16 range) = hi + lo, both rounded toward zero: 2 VALU ops per value
// (one v_cvt_pkrtz per pair for hi, one v_fma_mix_f32 per value for the exact residual v - hi,
// one v_cvt_pkrtz per pair for lo).
__device__ __forceinline__ H8 split8(const float* v) {
  typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
  union U {
    h2_t h;
    unsigned u;
  };
  union {
    half8 v8;
    unsigned u[4];
  } hi, lo;
#pragma unroll
  for (int pr = 0; pr < 4; ++pr) {
    U h2;
    h2.h = __builtin_amdgcn_cvt_pkrtz(v[2 * pr], v[2 * pr + 1]);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(h2.u), "v"(v[2 * pr]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(h2.u), "v"(v[2 * pr + 1]));
    U l2;
    l2.h = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi.u[pr] = h2.u;
    lo.u[pr] = l2.u;
  }
  H8 r;
  r.hi = hi.v8;
  r.lo = lo.v8;
  return r;
}
__device__ __forceinline__ H8 split8(f32x4 a, f32x4 b) {
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return split8(v);
}
__device__ __forceinline__ half8 ldh8(const _Float16* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ f32x4 mfma16(half8 a, half8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// acc += A * (b.hi + b.lo) with A = ah + al, dropping al*b.lo (2^-22 relative).  The two correction
// products go into the same f32 accumulator (callers interleave >= 2 independent accumulators).
__device__ __forceinline__ void mma3(f32x4& acc, const _Float16* blk, int lane, const H8& b) {
  const half8 ah = ldh8(blk + lane * 8), al = ldh8(blk + 512 + lane * 8);
  acc = mfma16(ah, b.hi, acc);
  acc = mfma16(ah, b.lo, acc);
  acc = mfma16(al, b.hi, acc);
}

// KT = compile-time bond_dim (0: run-time K <= 8); SPLIT: mode 1 (fp16 hi/lo products)
template <int KT, bool SPLIT>
__global__ __launch_bounds__(kThreads, kThreads / 256) void encoder_fused_kernel(EncParams p) {
  extern __shared__ __align__(16) float smem[];
  const int K = KT ? KT : p.K;
  Lds L;
  {
    float* f = smem;
    L.wimg = f; f += kImgSlot;
    L.hbuf0 = f; f += kRCap * kHS;
    L.hbuf1 = f; f += kRCap * kHS;
    L.tb = f; f += kTbCapFloats;
    L.ent = reinterpret_cast<uint32_t*>(f); f += kECap;
    L.ent2 = reinterpret_cast<uint32_t*>(f); f += kECap;
    L.rowptr = reinterpret_cast<int32_t*>(f); f += kRCap + 4;
    L.cursor = reinterpret_cast<int32_t*>(f); f += kRCap;
    L.cnt = reinterpret_cast<int32_t*>(f); f += kRCap;
    L.place = reinterpret_cast<int32_t*>(f); f += kRCap;
    L.rowinfo = reinterpret_cast<int32_t*>(f); f += kRCap;
    L.moloff = reinterpret_cast<int32_t*>(f); f += kRCap + 4;
    L.molrows = reinterpret_cast<int32_t*>(f); f += kRCap;
    L.tilemax = reinterpret_cast<int32_t*>(f); f += 16;
    L.bins = reinterpret_cast<int32_t*>(f); f += 48;
    L.scratch = reinterpret_cast<int32_t*>(f);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long* stamp = p.stamps ? p.stamps + (size_t)blockIdx.x * 32 : nullptr;
  if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memtime();
  const int g = p.n_ions == 2 ? (blockIdx.x & 1) : 0;
  const int c = p.n_ions == 2 ? (blockIdx.x >> 1) : blockIdx.x;
  const int4 dsc = reinterpret_cast<const int4*>(p.desc)[(int64_t)g * p.ub + c];
  // workgroup-uniform: keep them in SGPRs so every loop bound / branch below stays scalar
  const int m0 = __builtin_amdgcn_readfirstlane(dsc.x), M = __builtin_amdgcn_readfirstlane(dsc.y);
  const int base = __builtin_amdgcn_readfirstlane(dsc.z);
  const int R = __builtin_amdgcn_readfirstlane(dsc.w);  // <= kRCap by construction of the plan
  if (M <= 0) return;
  const int32_t* start = p.start + (int64_t)g * (p.B + 1);
  const int32_t* rows_g = p.rows + (int64_t)g * p.B;
  const int ntiles = (R + 15) >> 4;
  const int N = p.N, E = p.E;
  const int32_t* ids_g = p.atom_ids[g];
  const int32_t* conn_g = p.conn[g];
  const int32_t* bond_g = p.bond_ids[g];
  const float* img_g = p.img[g];

  // ---- prologue ------------------------------------------------------------------------
  // P0: chunk tables; this thread's first edge slot starts its flight
  f32x4 pf[kPf];
#if IMPNN_OPT_PF_LATE == 0
  if (p.S > 0) {
#pragma unroll
    for (int i = 0; i < kPf; ++i) pf[i] = ld4(img_g + 4 * (tid + i * kThreads));
  }
#endif
  const int n_slots = M * E;
  // edge slot owned by this thread in the first pass (slots beyond kThreads are re-read in loops)
  int s_m = 0, s_e = 0, s_bid = -1;
  int2 s_st = make_int2(0, 0);
  if (tid < n_slots) {
    s_m = tid / E;
    s_e = tid - s_m * E;
    const int64_t b = m0 + s_m;
    s_st = *reinterpret_cast<const int2*>(conn_g + (b * E + s_e) * 2);
    s_bid = bond_g[b * E + s_e];
  }
  const bool s_valid = tid < n_slots && edge_valid(s_st.x, s_st.y, s_bid, N, p.Vb);
  for (int m = tid; m <= M; m += kThreads) {
    L.moloff[m] = start[m0 + m] - base;
    if (m < M) L.molrows[m] = rows_g[m0 + m];
  }
  for (int t = tid; t < p.Vb * kKMax; t += kThreads) {
    const int v = t >> 3, k = t & 7;
    L.tb[t] = k < K ? p.bond_table[v * K + k] * (SPLIT ? kSX : 1.0f) : 0.f;  // mode 1: G comes out pre-scaled
  }
  if (tid < kRCap) L.cnt[tid] = 0;
  if (tid < 48) L.bins[tid] = 0;
  if (tid < 16) L.tilemax[tid] = 0;
  __syncthreads();
  if (stamp && tid == 0) stamp[16] = __builtin_amdgcn_s_memtime();

  // P1: in-degree of every logical row (edge-parallel); logical row -> (molecule, n, id>0);
  //     h0 rows (train_viscosity.py:171) start their flight: 4 threads per row, 2 x 16 B each
  if (s_valid) atomicAdd(&L.cnt[L.moloff[s_m] + s_st.y], 1);
  for (int slot = tid + kThreads; slot < n_slots; slot += kThreads) {
    const int m = slot / E, e = slot - m * E;
    const int64_t b = m0 + m;
    const int2 st = *reinterpret_cast<const int2*>(conn_g + (b * E + e) * 2);
    if (edge_valid(st.x, st.y, bond_g[b * E + e], N, p.Vb)) atomicAdd(&L.cnt[L.moloff[m] + st.y], 1);
  }
  f32x4 hv0 = {0.f, 0.f, 0.f, 0.f}, hv1 = hv0;
  {
    const int row = tid >> 2, sub = tid & 3;  // kThreads == 4 * kRCap
    int info = -1;
    if (row < R) {
      int lo = 0, hi = M - 1;  // largest m with moloff[m] <= row
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (L.moloff[mid] <= row) lo = mid; else hi = mid - 1;
      }
      const int n = row - L.moloff[lo];
      if (n < L.molrows[lo]) {
        const int id = ids_g[(int64_t)(m0 + lo) * N + n];
        info = (lo << 16) | (id > 0 ? 0x8000 : 0) | n;
        if ((unsigned)id < (unsigned)p.Va) {
          hv0 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub);
          hv1 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub + 4);
        }
      }
    }
    if (sub == 0) L.rowinfo[row] = info;
  }
  // the step-0 weight image starts its flight behind the latency-critical loads (loads retire in
  // order: issued first, it would hold up every small dependent load above)
#if IMPNN_OPT_PF_LATE == 1
  if (p.S > 0) {
#pragma unroll
    for (int i = 0; i < kPf; ++i) pf[i] = ld4(img_g + 4 * (tid + i * kThreads));
  }
#endif
  __syncthreads();
  if (stamp && tid == 0) stamp[17] = __builtin_amdgcn_s_memtime();

  // P2: place rows by descending in-degree (counting sort over 18 bins) so that a tile's lanes
  //     walk in-edge lists of similar length.  The placement inside a bin comes from an LDS
  //     atomic and may differ run to run - harmless: no result depends on where a row sits
  //     (MFMA columns, the gather and LayerNorm are per row; the pool walks logical rows).
  int my_bin = 0, my_deg = 0;
  if (tid < kRCap) {
    my_deg = L.cnt[tid];
    my_bin = tid >= R ? 0 : (my_deg >= 16 ? 1 : 17 - my_deg);  // bin 0 = beyond chunk (placed last) ... 17 = degree 0
    atomicAdd(&L.bins[my_bin], 1);
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of bins 1..17, then bin 0
    const int bidx = lane < kDegBins ? (lane == kDegBins - 1 ? 0 : lane + 1) : 0;  // placement order
    const int v = lane < kDegBins ? L.bins[bidx] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane < kDegBins) L.bins[24 + bidx] = incl - v;
  }
  __syncthreads();
  if (tid < kRCap) {
    const int pos = L.bins[24 + my_bin] + atomicAdd(&L.bins[my_bin], -1) - 1;
    L.place[tid] = pos;
    L.cursor[pos] = my_deg;  // in-degree per placed row (scanned below)
    if (my_deg > 0) atomicMax(&L.tilemax[pos >> 4], my_deg);
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[18] = __builtin_amdgcn_s_memtime();

  // P3: exclusive scan of the placed in-degrees -> rowptr; cursor = fill position
  {
    const int my_cnt = tid < kRCap ? L.cursor[tid] : 0;
    int incl = my_cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (tid < kRCap && lane == 63) L.scratch[wave] = incl;
    __syncthreads();
    if (tid < kRCap) {
      int off = 0;
      for (int w = 0; w < wave; ++w) off += L.scratch[w];
      const int excl = off + incl - my_cnt;
      L.rowptr[tid] = excl;
      L.cursor[tid] = excl;
      if (tid == kRCap - 1) L.rowptr[kRCap] = excl + my_cnt;
    }
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[19] = __builtin_amdgcn_s_memtime();

  // P4: fill.  entry = edge slot (16b) | bond id (8b) | placed source row (8b); the slot in the
  //     top bits lets P5 restore edge-slot order, so the accumulation order is fixed run to run.
  if (s_valid) {
    const int mo = L.moloff[s_m];
    const int pos = atomicAdd(&L.cursor[L.place[mo + s_st.y]], 1);
    L.ent2[pos] = ((uint32_t)s_e << 16) | ((uint32_t)s_bid << 8) | (uint32_t)L.place[mo + s_st.x];
  }
  for (int slot = tid + kThreads; slot < n_slots; slot += kThreads) {
    const int m = slot / E, e = slot - m * E;
    const int64_t b = m0 + m;
    const int2 st = *reinterpret_cast<const int2*>(conn_g + (b * E + e) * 2);
    const int bid = bond_g[b * E + e];
    if (edge_valid(st.x, st.y, bid, N, p.Vb)) {
      const int mo = L.moloff[m];
      const int pos = atomicAdd(&L.cursor[L.place[mo + st.y]], 1);
      L.ent2[pos] = ((uint32_t)e << 16) | ((uint32_t)bid << 8) | (uint32_t)L.place[mo + st.x];
    }
  }
  {  // h0 -> both LDS buffers at the row's placed position (slack rows: zeros)
    const int row = tid >> 2, sub = tid & 3;
    const int pr = L.place[row];
    st4(L.hbuf0 + pr * kHS + 8 * sub, hv0);
    st4(L.hbuf0 + pr * kHS + 8 * sub + 4, hv1);
    st4(L.hbuf1 + pr * kHS + 8 * sub, hv0);
    st4(L.hbuf1 + pr * kHS + 8 * sub + 4, hv1);
  }
  if (p.S > 0) {
#pragma unroll
    for (int i = 0; i < kPf; ++i) st4(L.wimg + 4 * (tid + i * kThreads), pf[i]);
  }
  __syncthreads();
  if (stamp && tid == 0) stamp[20] = __builtin_amdgcn_s_memtime();

  // P5: every row's in-edge list in edge-slot order: entry-parallel rank sort ent2 -> ent
#if IMPNN_OPT_PSORT == 0
  if (tid < kRCap) {
    const int lo = L.rowptr[tid], hi = L.rowptr[tid + 1];
    for (int i = lo; i < hi; ++i) {
      const uint32_t v = L.ent2[i];
      int rank = 0;
      for (int j = lo; j < hi; ++j) rank += L.ent2[j] < v;
      L.ent[lo + rank] = v;
    }
  }
#else
  {
    const int total = L.rowptr[kRCap];
    for (int i = tid; i < total; i += kThreads) {
      int lo = 0, hi = kRCap - 1;  // largest row with rowptr[row] <= i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (L.rowptr[mid] <= i) lo = mid; else hi = mid - 1;
      }
      const int b0 = L.rowptr[lo], b1 = L.rowptr[lo + 1];
      const uint32_t v = L.ent2[i];
      int rank = 0;
      for (int j = b0; j < b1; ++j) rank += L.ent2[j] < v;
      L.ent[b0 + rank] = v;
    }
  }
#endif
  __syncthreads();

  if (stamp && tid == 0) stamp[1] = __builtin_amdgcn_s_memtime();
  // ---- message-passing steps -----------------------------------------------------------
  // One 16-atom tile per wave (16 waves, 4 per SIMD): while one wave of a SIMD gathers or runs
  // its sigmoid/LayerNorm VALU work, the other three keep the matrix pipe fed.
  // Rows are placed by descending in-degree, so tile w (= wave w) has the longest gather of its
  // SIMD's four tiles when w is small: give the long-gather waves issue priority so the critical
  // path (heaviest tile) is not slowed by its lighter partners.
  {
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    if (wu < 4) __builtin_amdgcn_s_setprio(3);
    else if (wu < 8) __builtin_amdgcn_s_setprio(2);
    else if (wu < 12) __builtin_amdgcn_s_setprio(1);
  }
  const int a = lane & 15, q = lane >> 4;
  const float* wmsg = L.wimg;
  const float* wupd = L.wimg + img_msg_floats(K);
  const float* wvec = SPLIT ? L.wimg + img16_vec_float_off(K) : wupd + img_upd_floats();
  const _Float16* hmsg = reinterpret_cast<const _Float16*>(L.wimg);
  const _Float16* hupd = hmsg + img16_msg_halfs(K);
  for (int s = 0; s < p.S; ++s) {
    const float* hcur = (s & 1) ? L.hbuf1 : L.hbuf0;
    float* hnext = (s & 1) ? L.hbuf0 : L.hbuf1;
    const int sn = (s + 1) < p.S ? (s + 1) : s;  // the last step re-reads its own image: no branch
    const float* nxt = img_g + (int64_t)sn * kImgSlot;
    bool pf_issued = false;

    // Tile -> wave map.  Waves w, w+4, w+8, w+12 share a SIMD.  With ntiles = 4q + r the first r
    // SIMD groups carry q+1 tiles and the rest q; tiles are ordered heavy -> light (rows are placed by
    // descending in-degree), so the q-tile groups take the heaviest tiles and the (q+1)-tile groups
    // the light ones: the busiest SIMD is not also the one with the longest gathers.
    int my_tile = -1;
    {
      const int grp = wave & 3, slot = wave >> 2, q4 = ntiles >> 2, r4 = ntiles & 3;
      const int heavy = (4 - r4) * q4;  // tiles given to the q-tile groups
      if (grp >= r4) {
        if (slot < q4) my_tile = slot * (4 - r4) + (grp - r4);
      } else if (slot <= q4) {
        my_tile = heavy + slot * r4 + grp;
      }
    }
    for (int tile = my_tile; tile >= 0 && tile < ntiles; tile = -1) {
      const bool tstamp = stamp != nullptr && tile == 0 && s == (p.S > 1 ? 1 : 0);  // wave 0 only (tile 0)
      if (tstamp && lane == 0) stamp[8] = __builtin_amdgcn_s_memtime();
      const int row = tile * 16 + a;
      const f32x4 h0 = ld4(hcur + row * kHS + 4 * q);
      const f32x4 h1 = ld4(hcur + row * kHS + 16 + 4 * q);

      // ---- pull gather: G[k][j] = sum_{in-edges} tb[bond][k] * h[src][j]   (edge-slot order)
      const int p0 = L.rowptr[row];
      const int deg = L.rowptr[row + 1] - p0;
      const int maxdeg = __builtin_amdgcn_readfirstlane(L.tilemax[tile]);
      float G[kKMax][8];
      {
        // first in-edge initialises G (rows without in-edges use a zero coefficient vector)
        const uint32_t ent = L.ent[deg > 0 ? p0 : 0];
        const int src = ent & 0xffu, bid = (ent >> 8) & 0xffu;
        const f32x4 x0 = ld4(hcur + src * kHS + 4 * q);
        const f32x4 x1 = ld4(hcur + src * kHS + 16 + 4 * q);
        f32x4 c0 = ld4(L.tb + bid * kKMax);
        f32x4 c1 = ld4(L.tb + bid * kKMax + 4);
        if (deg <= 0) {
          c0 = f32x4{0.f, 0.f, 0.f, 0.f};
          c1 = c0;
        }
#pragma unroll
        for (int k = 0; k < kKMax; ++k) {
          const float ck = k < 4 ? c0[k & 3] : c1[k & 3];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            G[k][i] = ck * x0[i];
            G[k][4 + i] = ck * x1[i];
          }
        }
      }
      for (int d = 1; d < maxdeg; ++d) {
        if (d < deg) {
          const uint32_t ent = L.ent[p0 + d];
          const int src = ent & 0xffu, bid = (ent >> 8) & 0xffu;
          const f32x4 x0 = ld4(hcur + src * kHS + 4 * q);
          const f32x4 x1 = ld4(hcur + src * kHS + 16 + 4 * q);
          const f32x4 c0 = ld4(L.tb + bid * kKMax);
          const f32x4 c1 = ld4(L.tb + bid * kKMax + 4);
#pragma unroll
          for (int k = 0; k < kKMax; ++k) {
            const float ck = k < 4 ? c0[k & 3] : c1[k & 3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              G[k][i] = fmaf(ck, x0[i], G[k][i]);
              G[k][4 + i] = fmaf(ck, x1[i], G[k][4 + i]);
            }
          }
        }
      }
      if (tstamp && lane == 0) stamp[9] = __builtin_amdgcn_s_memtime();

      // ---- agg^T = sum_k W_k * G_k   (models/layers.py:108-112 + 78-82, reassociated)
      f32x4 agg0 = {0.f, 0.f, 0.f, 0.f}, agg1 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT) {
#pragma unroll
        for (int k = 0; k < kKMax; ++k) {
          if (k < K) {
            const H8 g = split8(G[k]);
            mma3(agg0, hmsg + (k * 2 + 0) * 1024, lane, g);
            mma3(agg1, hmsg + (k * 2 + 1) * 1024, lane, g);
          }
        }
        // accumulators carry kAcc; keep agg as agg*kSX: the B-operand scale of the next GEMMs
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          agg0[i] *= (kSX / kAcc);
          agg1[i] *= (kSX / kAcc);
        }
      } else {
#pragma unroll
        for (int k = 0; k < kKMax; ++k) {
          if (k < K) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const f32x4 A0 = ld4(wmsg + (k * kD + a) * kMsgRS + 16 * u + 4 * q);
              const f32x4 A1 = ld4(wmsg + (k * kD + 16 + a) * kMsgRS + 16 * u + 4 * q);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                agg0 = mfma4(A0[r], G[k][4 * u + r], agg0);
                agg1 = mfma4(A1[r], G[k][4 * u + r], agg1);
              }
            }
          }
        }
      }
      if (tstamp && lane == 0) stamp[10] = __builtin_amdgcn_s_memtime();
      // next step's weight image starts its flight now (G is dead: registers are free)
      if (!pf_issued) {
#pragma unroll
        for (int i = 0; i < kPf; ++i) pf[i] = ld4(nxt + 4 * (tid + i * kThreads));
        pf_issued = true;
      }

      // ---- gates z, r (models/layers.py:144-147) and candidate (:150-151)
      f32x4 z0 = ld4(wvec + 0 * kD + 4 * q), z1 = ld4(wvec + 0 * kD + 16 + 4 * q);
      f32x4 r0 = ld4(wvec + 1 * kD + 4 * q), r1 = ld4(wvec + 1 * kD + 16 + 4 * q);
      f32x4 t0 = ld4(wvec + 2 * kD + 4 * q), t1 = ld4(wvec + 2 * kD + 16 + 4 * q);
      f32x4 rh0, rh1;
      if constexpr (SPLIT) {
        f32x4 hs0, hs1;  // h * kSX
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hs0[i] = h0[i] * kSX;
          hs1[i] = h1[i] * kSX;
        }
        const H8 sh = split8(hs0, hs1);
        const H8 sa = split8(agg0, agg1);
        // block index = ((gate*2 + T)*2 + half), 1024 halfs each
        mma3(z0, hupd + ((0 * 2 + 0) * 2 + 0) * 1024, lane, sh);
        mma3(z1, hupd + ((0 * 2 + 1) * 2 + 0) * 1024, lane, sh);
        mma3(r0, hupd + ((1 * 2 + 0) * 2 + 0) * 1024, lane, sh);
        mma3(r1, hupd + ((1 * 2 + 1) * 2 + 0) * 1024, lane, sh);
        mma3(z0, hupd + ((0 * 2 + 0) * 2 + 1) * 1024, lane, sa);
        mma3(z1, hupd + ((0 * 2 + 1) * 2 + 1) * 1024, lane, sa);
        mma3(r0, hupd + ((1 * 2 + 0) * 2 + 1) * 1024, lane, sa);
        mma3(r1, hupd + ((1 * 2 + 1) * 2 + 1) * 1024, lane, sa);
        if (tstamp && lane == 0) stamp[11] = __builtin_amdgcn_s_memtime();
        f32x4 rs0, rs1;  // sigmoid(r) * h * kSX
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          z0[i] = fast_sigmoid_scaled(z0[i]);  // accumulators carry kAcc: folded into the exp2 constant
          z1[i] = fast_sigmoid_scaled(z1[i]);
          rs0[i] = fast_sigmoid_scaled(r0[i]) * hs0[i];  // :149
          rs1[i] = fast_sigmoid_scaled(r1[i]) * hs1[i];
        }
        const H8 srh = split8(rs0, rs1);
        mma3(t0, hupd + ((2 * 2 + 0) * 2 + 0) * 1024, lane, srh);
        mma3(t1, hupd + ((2 * 2 + 1) * 2 + 0) * 1024, lane, srh);
        mma3(t0, hupd + ((2 * 2 + 0) * 2 + 1) * 1024, lane, sa);
        mma3(t1, hupd + ((2 * 2 + 1) * 2 + 1) * 1024, lane, sa);
      } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int col = 32 * half + 16 * u + 4 * q;
            const f32x4 Az0 = ld4(wupd + (0 * kD + a) * kUpdRS + col);
            const f32x4 Az1 = ld4(wupd + (0 * kD + 16 + a) * kUpdRS + col);
            const f32x4 Ar0 = ld4(wupd + (1 * kD + a) * kUpdRS + col);
            const f32x4 Ar1 = ld4(wupd + (1 * kD + 16 + a) * kUpdRS + col);
            const f32x4 Bv = half == 0 ? (u == 0 ? h0 : h1) : (u == 0 ? agg0 : agg1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              z0 = mfma4(Az0[r], Bv[r], z0);
              z1 = mfma4(Az1[r], Bv[r], z1);
              r0 = mfma4(Ar0[r], Bv[r], r0);
              r1 = mfma4(Ar1[r], Bv[r], r1);
            }
          }
        }
        if (tstamp && lane == 0) stamp[11] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          z0[i] = fast_sigmoid(z0[i]);
          z1[i] = fast_sigmoid(z1[i]);
          rh0[i] = fast_sigmoid(r0[i]) * h0[i];  // :149
          rh1[i] = fast_sigmoid(r1[i]) * h1[i];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int col = 32 * half + 16 * u + 4 * q;
            const f32x4 Ah0 = ld4(wupd + (2 * kD + a) * kUpdRS + col);
            const f32x4 Ah1 = ld4(wupd + (2 * kD + 16 + a) * kUpdRS + col);
            const f32x4 Bv = half == 0 ? (u == 0 ? rh0 : rh1) : (u == 0 ? agg0 : agg1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              t0 = mfma4(Ah0[r], Bv[r], t0);
              t1 = mfma4(Ah1[r], Bv[r], t1);
            }
          }
        }
      }
      if (tstamp && lane == 0) stamp[12] = __builtin_amdgcn_s_memtime();
      // ---- blend, LayerNorm, residual  (models/layers.py:153-155)
      f32x4 n0, n1;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (1-z) h + z t == h + z (t - h)
        n0[i] = fmaf(z0[i], (SPLIT ? fast_tanh_scaled(t0[i]) : fast_tanh(t0[i])) - h0[i], h0[i]);
        n1[i] = fmaf(z1[i], (SPLIT ? fast_tanh_scaled(t1[i]) : fast_tanh(t1[i])) - h1[i], h1[i]);
        sum += n0[i] + n1[i];
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float mean = sum * (1.0f / kD);
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        n0[i] -= mean;
        n1[i] -= mean;
        var = fmaf(n0[i], n0[i], var);
        var = fmaf(n1[i], n1[i], var);
      }
      var += __shfl_xor(var, 16);
      var += __shfl_xor(var, 32);
      const float inv = 1.0f / sqrtf(var * (1.0f / kD) + p.ln_eps);
      const f32x4 g0 = ld4(wvec + 3 * kD + 4 * q), g1 = ld4(wvec + 3 * kD + 16 + 4 * q);
      const f32x4 b0 = ld4(wvec + 4 * kD + 4 * q), b1 = ld4(wvec + 4 * kD + 16 + 4 * q);
      f32x4 o0, o1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o0[i] = n0[i] * inv * g0[i] + b0[i] + h0[i];
        o1[i] = n1[i] * inv * g1[i] + b1[i] + h1[i];
      }
      st4(hnext + row * kHS + 4 * q, o0);
      st4(hnext + row * kHS + 16 + 4 * q, o1);
      if (tstamp && lane == 0) stamp[13] = __builtin_amdgcn_s_memtime();
    }
    if (!pf_issued) {  // waves without a tile in this chunk still carry their share of the image
#pragma unroll
      for (int i = 0; i < kPf; ++i) pf[i] = ld4(nxt + 4 * (tid + i * kThreads));
    }
    if (stamp && tid == 0 && s == (p.S > 1 ? 1 : 0)) stamp[14] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (stamp && tid == 0 && s == (p.S > 1 ? 1 : 0)) stamp[15] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < kPf; ++i) st4(L.wimg + 4 * (tid + i * kThreads), pf[i]);
    __syncthreads();
    if (p.stamps && tid == 0 && s < 5) stamp[2 + s] = __builtin_amdgcn_s_memtime();
  }

  // ---- GlobalSumPool (models/layers.py:161-164): rows whose atom id > 0.  Four lanes share one
  //      (molecule, 4 features): each sums every 4th row in ascending order, then a fixed 2-step
  //      butterfly - a wavefront segmented reduction with a run-to-run fixed order.
  const float* hfin = (p.S & 1) ? L.hbuf1 : L.hbuf0;
  float* out_g = p.pooled[g];
  for (int t0 = 0; t0 < M * 32; t0 += kThreads) {
    const int t = t0 + tid;
    const int part = t & 3, f4 = (t >> 2) & 7, m = t >> 5;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const int nr = L.molrows[m], mo = L.moloff[m];
      for (int n = part; n < nr; n += 4)
        if (L.rowinfo[mo + n] & 0x8000) acc += ld4(hfin + L.place[mo + n] * kHS + 4 * f4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += __shfl_xor(acc[i], 1);
      acc[i] += __shfl_xor(acc[i], 2);
    }
    if (m < M && part == 0) st4(out_g + (int64_t)(m0 + m) * kD + 4 * f4, acc);
  }
  if (stamp && tid == 0) {
    stamp[7] = __builtin_amdgcn_s_memtime();
    stamp[6] = ((unsigned long long)R << 32) | (unsigned)M;
  }
}

}  // namespace

bool encoder_fused_supported(int N, int E, int D, int K, int S, int Vb) {
  if (D != kD || K < 1 || K > kKMax || S < 0) return false;
  if (N < 1 || N > 0xffff || E < 0) return false;
  if (Vb < 1 || Vb > 0xffff || (int64_t)Vb * kKMax > kTbCapFloats) return false;
  const int vrmax = vr_max_of(N, E);
  if (vrmax > kRCap / 2) return false;  // keep the packing window >= half a chunk
  return true;
}

size_t encoder_fused_workspace_bytes(int n_ions, int B, int N, int E, int D, int K, int S, int Vb) {
  (void)D; (void)Vb;
  return ws_layout(n_ions, B, N, E, K, S).total;
}

size_t encoder_prepared_bytes(int S) { return (size_t)(S > 0 ? S : 1) * kImgSlot * sizeof(float); }

int launch_encoder_prepare(const float* weights, int D, int K, int S, int mode, void* prepared, hipStream_t s) {
  if (S <= 0) return IMPNN_OK;
  ImageParams ip{};
  ip.weights = weights;
  ip.img = static_cast<float*>(prepared);
  ip.K = K;
  ip.mode = mode == 1 ? 1 : 0;
  ip.step_floats = impnn_encoder_step_floats(D, K);
  weight_image_kernel<<<dim3(16, S), 256, 0, s>>>(ip);
  return check_launch("weight_image");
}

int launch_encoder_fused(const EncoderArgs& a, hipStream_t s) {
  const Ws w = ws_layout(a.n_ions, a.B, a.N, a.E, a.K, a.S);
  if (!aligned16(a.workspace)) return fail(IMPNN_E_BADARG, "encoder_fused: workspace must be 16B aligned");
  if (!aligned16(a.atom_table)) return fail(IMPNN_E_BADARG, "encoder_fused: atom_table must be 16B aligned");
  char* base = static_cast<char*>(a.workspace);
  const int mode = a.mode == 1 ? 1 : 0;
  PlanParams pp{};
  EncParams ep{};
  for (int g = 0; g < a.n_ions; ++g) {
    if ((reinterpret_cast<uintptr_t>(a.conn[g]) & 7u) != 0)
      return fail(IMPNN_E_BADARG, "encoder_fused: connectivity must be 8B aligned");
    pp.atom_ids[g] = ep.atom_ids[g] = a.atom_ids[g];
    pp.bond_ids[g] = ep.bond_ids[g] = a.bond_ids[g];
    pp.conn[g] = ep.conn[g] = a.conn[g];
    ep.pooled[g] = a.pooled[g];
    if (a.prepared[g]) {
      if (!aligned16(a.prepared[g])) return fail(IMPNN_E_BADARG, "encoder_fused: prepared weights must be 16B aligned");
      ep.img[g] = static_cast<const float*>(a.prepared[g]);
    } else {  // canonical weights: build the image into the workspace first
      float* img = reinterpret_cast<float*>(base + w.img_off) + (size_t)g * (a.S > 0 ? a.S : 1) * kImgSlot;
      if (int rc = launch_encoder_prepare(a.weights[g], a.D, a.K, a.S, mode, img, s)) return rc;
      ep.img[g] = img;
    }
  }
  pp.rows = reinterpret_cast<int32_t*>(base + w.rows_off);
  pp.vr = reinterpret_cast<int32_t*>(base + w.vr_off);
  pp.start = reinterpret_cast<int32_t*>(base + w.start_off);
  pp.first = reinterpret_cast<int32_t*>(base + w.first_off);
  pp.nchunks = reinterpret_cast<int32_t*>(base + w.nchunks_off);
  pp.desc = reinterpret_cast<int32_t*>(base + w.desc_off);
  pp.n_ions = a.n_ions; pp.B = a.B; pp.N = a.N; pp.E = a.E; pp.K = a.K; pp.S = a.S; pp.Vb = a.Vb;
  pp.win = kRCap - vr_max_of(a.N, a.E) + 1;
  pp.ub = w.ub;
  const int waves_per_block = 4;
  const int mol_blocks = (int)(((int64_t)a.n_ions * a.B + waves_per_block - 1) / waves_per_block);
  plan_stats_kernel<<<mol_blocks, 64 * waves_per_block, 0, s>>>(pp);
  if (int rc = check_launch("plan_stats")) return rc;
  plan_scan_kernel<<<a.n_ions, 1024, 0, s>>>(pp);
  if (int rc = check_launch("plan_scan")) return rc;

  ep.atom_table = a.atom_table;
  ep.bond_table = a.bond_table;
  ep.rows = pp.rows; ep.start = pp.start; ep.first = pp.first; ep.nchunks = pp.nchunks; ep.desc = pp.desc;
  ep.n_ions = a.n_ions; ep.B = a.B; ep.N = a.N; ep.E = a.E; ep.K = a.K; ep.S = a.S;
  ep.Va = a.Va; ep.Vb = a.Vb; ep.ub = w.ub; ep.ln_eps = a.ln_eps;
  ep.stamps = nullptr;
  {
    size_t sb = 0;
    void* sp = debug_stamp_buffer(&sb);
    if (sp && sb >= (size_t)w.ub * a.n_ions * 32 * sizeof(unsigned long long))
      ep.stamps = static_cast<unsigned long long*>(sp);
  }
  const size_t lds = lds_bytes(a.K);
  const int variant = (a.K == 8 ? 1 : 0) + 2 * mode;
  void (*kern)(EncParams) = variant == 0   ? encoder_fused_kernel<0, false>
                            : variant == 1 ? encoder_fused_kernel<8, false>
                            : variant == 2 ? encoder_fused_kernel<0, true>
                                           : encoder_fused_kernel<8, true>;
  static bool attr_set[4] = {false, false, false, false};
  if (!attr_set[variant]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return fail(IMPNN_E_LAUNCH, "encoder_fused: cannot raise LDS limit: %s", hipGetErrorString(e));
    attr_set[variant] = true;
  }
  profile_record_start(s);
  kern<<<w.ub * a.n_ions, kThreads, lds, s>>>(ep);
  profile_record_stop(s);
  return check_launch("encoder_fused");
}

}  // namespace impnn
