// Layer-at-a-time kernels for the drop-in Keras-layer surface (gfx950).
//
// These are the HBM-bound "one reference layer = one launch" kernels behind
// BondMatrixMessage / Reduce / GatedUpdate / GlobalSumPool / Embedding.  The graphs/sec
// headline runs through encoder_fused.hip instead; these exist so that each reference layer
// has a bit-checkable counterpart with the reference's own tensor boundaries.
//
// Reference lines each kernel follows are cited at the kernel.
#include "common.h"

namespace impnn {

namespace {

constexpr int kBlock = 256;

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t ldv4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ f32x4_t mfma_f32(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float fsig(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
__device__ __forceinline__ float ftanh(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * x));
}


__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// a1/a2  Embedding lookup: out[r,:] = table[ids[r],:]  (train_viscosity.py:171-172)
// One thread per 16-byte piece of an output row; out-of-range id -> zero row.
// ---------------------------------------------------------------------------------------
template <int VEC>
__global__ void embed_gather_kernel(const int32_t* __restrict__ ids, const float* __restrict__ table,
                                    float* __restrict__ out, int64_t rows, int vocab, int dim) {
  const int pieces = dim / VEC;
  const int64_t total = rows * pieces;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / pieces;
    const int c = (int)(t - r * pieces) * VEC;
    const int id = ids[r];
    const bool ok = (unsigned)id < (unsigned)vocab;
    if constexpr (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(table + (int64_t)id * dim + c);
      *reinterpret_cast<float4*>(out + r * dim + c) = v;
    } else {
      out[r * dim + c] = ok ? table[(int64_t)id * dim + c] : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a4  BondMatrixMessage.call (models/layers.py:100-117), dense bond_state, any D/K.
//     m[b,e,i] = [src>0 & tgt>0] * sum_k bs[b,e,k] * sum_j W[k,i,j] * h[b,src,j]
// One workgroup per molecule; h[b] staged in LDS; optional fused Reduce (a10) when agg != null:
// messages are kept in LDS and scattered in edge-slot order by column-owner threads.
// ---------------------------------------------------------------------------------------
__global__ void bmm_message_kernel(const float* __restrict__ h, const float* __restrict__ bs,
                                   const int32_t* __restrict__ conn, const float* __restrict__ W,
                                   float* __restrict__ m_out, float* __restrict__ agg_out, int N,
                                   int E, int D, int K, int h_in_lds) {
  extern __shared__ __align__(16) float smem[];
  const int b = blockIdx.x;
  const float* hb = h + (int64_t)b * N * D;
  float* hs = smem;                                  // N*D (if h_in_lds)
  float* ms = smem + (h_in_lds ? (size_t)N * D : 0);  // E*D (only when fusing the reduce)
  if (h_in_lds) {
    for (int t = threadIdx.x; t < N * D; t += blockDim.x) hs[t] = hb[t];
    __syncthreads();
  }
  const float* hsrc = h_in_lds ? hs : hb;
  const int32_t* cb = conn + (int64_t)b * E * 2;
  const float* bsb = bs + (int64_t)b * E * K;
  for (int t = threadIdx.x; t < E * D; t += blockDim.x) {
    const int e = t / D, i = t - e * D;
    const int src = cb[2 * e], tgt = cb[2 * e + 1];
    float acc = 0.f;
    if (src > 0 && tgt > 0 && src < N && tgt < N) {
      const float* s = hsrc + (size_t)src * D;
      for (int k = 0; k < K; ++k) {
        const float* w = W + ((int64_t)k * D + i) * D;
        float dot = 0.f;
        for (int j = 0; j < D; ++j) dot = fmaf(w[j], s[j], dot);
        acc = fmaf(bsb[(int64_t)e * K + k], dot, acc);
      }
    }
    if (m_out) m_out[((int64_t)b * E + e) * D + i] = acc;
    if (agg_out) ms[t] = acc;
  }
  if (agg_out) {
    __syncthreads();
    // Reduce.call (models/layers.py:57-83): column owner walks the edge slots in order.
    float* ab = agg_out + (int64_t)b * N * D;
    for (int t = threadIdx.x; t < N * D; t += blockDim.x) ab[t] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += blockDim.x) {
      for (int e = 0; e < E; ++e) {
        const int tgt = cb[2 * e + 1];
        if (tgt > 0 && tgt < N) ab[(int64_t)tgt * D + i] += ms[e * D + i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// schedule A: per-bond-type matrices  A[v,i,j] = sum_k Tb[v,k] * W[k,i,j]
// (tf.tensordot of models/layers.py:108 evaluated once per vocabulary entry)
// ---------------------------------------------------------------------------------------
__global__ void bond_type_matrices_kernel(const float* __restrict__ tb, const float* __restrict__ W,
                                          float* __restrict__ out, int Vb, int K, int DD) {
  const int v = blockIdx.y;
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < DD; ij += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(tb[(int64_t)v * K + k], W[(int64_t)k * DD + ij], acc);
    out[(int64_t)v * DD + ij] = acc;
  }
}

// The same for GEMM-shaped bond_dim (K = D^2 = 1024 of train_melting_point.py:146): out (Vb x DD) = Tb (Vb x K) W (K x DD)
// on v_mfma_f32_16x16x4_f32 (exact f32 products).  One 256-thread workgroup per 16 output columns; its four waves split
// K and meet in LDS in wave order (a fixed summation order: bitwise reproducible); a lane's four k of a 16-k block are
// consecutive (one 16-byte load of its Tb row), the MFMA step s takes component s of every lane - A and B agree on that
// order, which is all a dot product asks.  VT = 16-row tiles of the vocabulary (Vb <= 16 VT).
template <int VT>
__global__ __launch_bounds__(256) void bond_type_matrices_mfma_kernel(const float* __restrict__ tb, const float* __restrict__ W,
                                                                      float* __restrict__ out, int Vb, int K, int DD) {
  __shared__ f32x4_t part[3][VT][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 15, q = lane >> 4;
  const int j0 = blockIdx.x * 16;
  const int kw = K >> 2, k_lo = wave * kw;  // this wave's quarter of K (a multiple of 16: launch_bond_type_matrices)
  f32x4_t acc[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) acc[vt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float* arow[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int v = 16 * vt + a;
    arow[vt] = tb + (int64_t)(v < Vb ? v : Vb - 1) * K + 4 * q;  // (rows past the vocabulary: multiplied, never stored)
  }
  const float* bcol = W + (int64_t)(4 * q) * DD + j0 + a;
  for (int k0 = k_lo; k0 < k_lo + kw; k0 += 16) {
    f32x4_t av[VT];
    float bv[4];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) av[vt] = *reinterpret_cast<const f32x4_t*>(arow[vt] + k0);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) bv[s4] = bcol[(int64_t)(k0 + s4) * DD];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) acc[vt] = mfma_f32(av[vt][s4], bv[s4], acc[vt]);
  }
  if (wave > 0) {
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) part[wave - 1][vt][lane] = acc[vt];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      f32x4_t r = acc[vt];
      r += part[0][vt][lane];
      r += part[1][vt][lane];
      r += part[2][vt][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = 16 * vt + 4 * q + i;
        if (v < Vb) out[(int64_t)v * DD + j0 + a] = r[i];
      }
    }
  }
}

// a4 from bond ids: m[b,e,:] = A[bond_ids[b,e]] @ h[b,src,:], masked like models/layers.py:114-115
__global__ void bmm_message_typed_kernel(const float* __restrict__ h, const int32_t* __restrict__ bond_ids,
                                         const int32_t* __restrict__ conn, const float* __restrict__ A,
                                         float* __restrict__ m_out, int N, int E, int D, int Vb) {
  extern __shared__ __align__(16) float smem[];
  const int b = blockIdx.x;
  const float* hb = h + (int64_t)b * N * D;
  for (int t = threadIdx.x; t < N * D; t += blockDim.x) smem[t] = hb[t];
  __syncthreads();
  const int32_t* cb = conn + (int64_t)b * E * 2;
  for (int t = threadIdx.x; t < E * D; t += blockDim.x) {
    const int e = t / D, i = t - e * D;
    const int src = cb[2 * e], tgt = cb[2 * e + 1];
    const int ty = bond_ids[(int64_t)b * E + e];
    float acc = 0.f;
    if (src > 0 && tgt > 0 && src < N && tgt < N && (unsigned)ty < (unsigned)Vb) {
      const float* a = A + ((int64_t)ty * D + i) * D;
      const float* s = smem + (size_t)src * D;
      for (int j = 0; j < D; ++j) acc = fmaf(a[j], s[j], acc);
    }
    m_out[((int64_t)b * E + e) * D + i] = acc;
  }
}

// ---------------------------------------------------------------------------------------
// a4 from bond ids for D = 32 (schedule A) on the matrix cores.
// A workgroup takes kTM molecules, counting-sorts their valid edges by bond type in LDS and cuts
// every type's run into tiles of 16 edges; a tile is the GEMM  m^T (32x16) = A_type (32x32) * h_src^T
// (32x16) on v_mfma_f32_16x16x4_f32 (exact f32), with A_type fetched once per tile from L2 instead of
// once per edge.  Masked / out-of-range edges get zero rows (models/layers.py:114-115).
// ---------------------------------------------------------------------------------------
constexpr int kTM = 32;          // molecules per workgroup
constexpr int kTSlots = 4096;    // edge slots per workgroup held in LDS (kTM * E)
constexpr int kTVb = 1024;       // bond types handled by the LDS histogram

__global__ __launch_bounds__(512) void bmm_message_typed_d32_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ bond_ids, const int32_t* __restrict__ conn,
    const float* __restrict__ A, float* __restrict__ m_out, int B, int N, int E, int Vb, int tm) {
  constexpr int D = 32;
  __shared__ int32_t hist[kTVb + 1], cursor[kTVb], tbase[kTVb + 1];
  __shared__ uint32_t sorted[kTSlots];
  __shared__ uint16_t tile_type[kTSlots / 16 + kTVb];
  __shared__ int32_t wtot[8], carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * tm;  // tm <= kTM molecules per workgroup (fewer for small batches: more workgroups)
  const int nm = (B - b0) < tm ? (B - b0) : tm;
  const int n_slots = nm * E;
  for (int t = tid; t <= Vb; t += blockDim.x) hist[t] = 0;
  __syncthreads();
  // pass 1: classify every edge slot; zero rows for masked edges; histogram of types
  for (int slot = tid; slot < n_slots; slot += blockDim.x) {
    const int ml = slot / E, e = slot - ml * E;
    const int64_t g = (int64_t)(b0 + ml) * E + e;
    const int2 st = *reinterpret_cast<const int2*>(conn + g * 2);
    const int ty = bond_ids[g];
    const bool ok = st.x > 0 && st.y > 0 && st.x < N && st.y < N && (unsigned)ty < (unsigned)Vb;
    if (ok) {
      atomicAdd(&hist[ty], 1);
    } else {
      f32x4_t z = {0.f, 0.f, 0.f, 0.f};
      float* o = m_out + g * D;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4_t*>(o + 4 * i) = z;
    }
  }
  __syncthreads();
  // exclusive scans over types: edge offsets (hist -> cursor) and tile offsets (tbase)
  {
    const int per = (Vb + (int)blockDim.x - 1) / (int)blockDim.x;
    const int t0 = tid * per, t1 = (t0 + per) < Vb ? (t0 + per) : Vb;
    int le = 0, lt = 0;
    for (int t = t0; t < t1; ++t) {
      le += hist[t];
      lt += (hist[t] + 15) >> 4;
    }
    int ie = le, it = lt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int ue = __shfl_up(ie, o), ut = __shfl_up(it, o);
      if (lane >= o) {
        ie += ue;
        it += ut;
      }
    }
    if (lane == 63) wtot[wave] = (ie << 16) | it;  // <= 4096 edges, <= 1280 tiles: 16 bits each
    __syncthreads();
    int oe = 0, ot = 0;
    for (int w = 0; w < wave; ++w) {
      oe += wtot[w] >> 16;
      ot += wtot[w] & 0xffff;
    }
    int re = oe + ie - le, rt = ot + it - lt;
    for (int t = t0; t < t1; ++t) {
      const int c = hist[t];
      cursor[t] = re;
      tbase[t] = rt;
      for (int j = 0; j < ((c + 15) >> 4); ++j) tile_type[rt + j] = (uint16_t)t;
      re += c;
      rt += (c + 15) >> 4;
    }
    if (tid == (int)blockDim.x - 1) carry_s = rt;  // total tiles (last thread sees the full prefix)
  }
  __syncthreads();
  // pass 2: scatter the valid edges into their type's run (order inside a run is irrelevant: every
  // edge's row is computed independently)
  for (int slot = tid; slot < n_slots; slot += blockDim.x) {
    const int ml = slot / E, e = slot - ml * E;
    const int64_t g = (int64_t)(b0 + ml) * E + e;
    const int2 st = *reinterpret_cast<const int2*>(conn + g * 2);
    const int ty = bond_ids[g];
    if (st.x > 0 && st.y > 0 && st.x < N && st.y < N && (unsigned)ty < (unsigned)Vb) {
      const int pos = atomicAdd(&cursor[ty], 1);
      sorted[pos] = ((uint32_t)ml << 27) | ((uint32_t)e << 12) | (uint32_t)st.x;  // 5 | 15 | 12 bits
    }
  }
  __syncthreads();
  // tiles: contiguous ranges per wave, so consecutive tiles mostly share their type's matrix
  const int ntile = carry_s, nw = blockDim.x >> 6;
  const int a = lane & 15, q = lane >> 4;
  int cur_ty = -1;
  f32x4_t A0[2], A1[2];
  for (int tile = (ntile * wave) / nw; tile < (ntile * (wave + 1)) / nw; ++tile) {
    const int ty = tile_type[tile];
    if (ty != cur_ty) {  // wave-uniform
      cur_ty = ty;
      const float* At = A + (int64_t)ty * D * D;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        A0[u] = ldv4(At + a * D + 16 * u + 4 * q);         // rows 0..15 of A_type, columns 16u+4q..
        A1[u] = ldv4(At + (16 + a) * D + 16 * u + 4 * q);  // rows 16..31
      }
    }
    const int run_end = cursor[ty];                 // after pass 2: one past the type's run
    const int idx = run_end - hist[ty] + (tile - tbase[ty]) * 16 + a;
    const bool live = idx < run_end;
    const uint32_t ent = sorted[live ? idx : run_end - 1];
    const int ml = ent >> 27, e = (ent >> 12) & 0x7fff, src = ent & 0xfff;
    const float* hs = h + ((int64_t)(b0 + ml) * N + src) * D;
    const f32x4_t x0 = ldv4(hs + 4 * q), x1 = ldv4(hs + 16 + 4 * q);
    f32x4_t o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0 = mfma_f32(A0[0][r], x0[r], o0);
      o1 = mfma_f32(A1[0][r], x0[r], o1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0 = mfma_f32(A0[1][r], x1[r], o0);
      o1 = mfma_f32(A1[1][r], x1[r], o1);
    }
    if (live) {
      float* o = m_out + ((int64_t)(b0 + ml) * E + e) * D;
      *reinterpret_cast<f32x4_t*>(o + 4 * q) = o0;
      *reinterpret_cast<f32x4_t*>(o + 16 + 4 * q) = o1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a5  Reduce.call (models/layers.py:57-83).  Thread (molecule, column) walks the E edge slots in
// order and adds into its own column of agg[b] - no atomics, bitwise equal to a sequential
// scatter_nd.  agg[b] lives in LDS when it fits, else directly in HBM/L2.
// ---------------------------------------------------------------------------------------
// accumulate: the sums start from what agg holds instead of from zero (the message adjoint's source-row sums on top
// of the GatedUpdate's dh: launch_bmm_message_typed_bwd with a per-edge buffer).
// blockIdx.y: a range of `rows_per` target rows.  A row's sum only depends on the order of ITS in-edges, so workgroups that
// each walk all edge slots and add the rows of their own range give the same bits as one walk - and small batches of
// large molecules (the explicit-hydrogen shape at the reference's batch of 32: 16 workgroups, accumulators in HBM
// because 160 x 128 floats x 2 molecules do not fit LDS) get gridDim.y times the threads and LDS accumulators again.
__global__ void reduce_scatter_kernel(const float* __restrict__ m, const int32_t* __restrict__ tgt,
                                      int tgt_stride, float* __restrict__ agg, int B, int N, int E,
                                      int D, int mols_per_block, int use_lds, int accumulate, int rows_per,
                                      int use_list) {
  extern __shared__ __align__(16) float smem[];
  const int cols = D < (int)blockDim.x ? D : (int)blockDim.x;  // threads per molecule
  const int ml = threadIdx.x / cols;
  const int c0 = threadIdx.x - ml * cols;
  const int b = blockIdx.x * mols_per_block + ml;
  const bool active = ml < mols_per_block && b < B;
  const int n_lo = blockIdx.y * rows_per, n_hi = n_lo + rows_per < N ? n_lo + rows_per : N;
  // acc[(t - n_lo) * D + column]: the rows [n_lo, n_hi) of this molecule
  float* acc = use_lds ? smem + (size_t)ml * rows_per * D : (active ? agg + ((int64_t)b * N + n_lo) * D : nullptr);
  // use_list (cols == D, E and N below 65536): a wave per molecule first compacts the slots whose target lies in this
  // workgroup's row range - (slot << 16 | row - n_lo), in slot order - so the column threads walk only those instead
  // of testing all E indices each (E = 640, sixteen ranges: 40 tests per row that is added)
  int32_t* list = reinterpret_cast<int32_t*>(smem + (use_lds ? (size_t)mols_per_block * rows_per * D : 0));
  int32_t* cnts = list + (size_t)mols_per_block * E;
  if (use_list) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwave = (int)blockDim.x >> 6;
    for (int mw = wave; mw < mols_per_block; mw += nwave) {
      const int bb = blockIdx.x * mols_per_block + mw;
      int cnt = 0;
      if (bb < B) {
        const int32_t* tbb = tgt + (int64_t)bb * E * tgt_stride;
        for (int e0 = 0; e0 < E; e0 += 64) {
          const int e = e0 + lane;
          const int t = e < E ? tbb[(int64_t)e * tgt_stride] : 0;
          const bool ok = t > 0 && t >= n_lo && t < n_hi;
          const unsigned long long mask = __ballot(ok);
          if (ok) list[(size_t)mw * E + cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (e << 16) | (t - n_lo);
          cnt += __popcll(mask);
        }
      }
      if (lane == 0) cnts[mw] = cnt;
    }
  }
  if (active && (use_lds || !accumulate)) {
    const float* ab0 = agg + ((int64_t)b * N + n_lo) * D;
    for (int n = 0; n < n_hi - n_lo; ++n)
      for (int i = c0; i < D; i += cols) acc[(size_t)n * D + i] = accumulate ? ab0[(size_t)n * D + i] : 0.f;
  }
  if (use_list) __syncthreads();
  if (active) {
    const float* mb = m + (int64_t)b * E * D;
    const int32_t* tb = tgt + (int64_t)b * E * tgt_stride;
    if (use_list) {  // (cols == D)
      constexpr int kU = 16;
      const int32_t* L = list + (size_t)ml * E;
      const int n = cnts[ml];
      int i = 0;
      for (; i + kU <= n; i += kU) {
        float v[kU];
        int p[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) p[u] = L[i + u];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = mb[(int64_t)((unsigned)p[u] >> 16) * D + c0];
#pragma unroll
        for (int u = 0; u < kU; ++u) acc[(size_t)(p[u] & 0xffff) * D + c0] += v[u];
      }
      for (; i < n; ++i) {
        const int p = L[i];
        acc[(size_t)(p & 0xffff) * D + c0] += mb[(int64_t)((unsigned)p >> 16) * D + c0];
      }
    } else if (cols == D) {  // one column per thread: keep 16 edge rows in flight; adds stay in edge-slot order
      constexpr int kU = 16;
      int e = 0;
      for (; e + kU <= E; e += kU) {
        float v[kU];
        int t[kU];
        // the indices first (the same words for every lane of a molecule), then only the rows that will be added: padded
        // edge slots - nearly half of them in the benchmark's batches - are not read at all
#pragma unroll
        for (int u = 0; u < kU; ++u) t[u] = tb[(int64_t)(e + u) * tgt_stride];
#pragma unroll
        for (int u = 0; u < kU; ++u) v[u] = (t[u] > 0 && t[u] >= n_lo && t[u] < n_hi) ? mb[(int64_t)(e + u) * D + c0] : 0.f;
#pragma unroll
        for (int u = 0; u < kU; ++u)
          if (t[u] > 0 && t[u] >= n_lo && t[u] < n_hi) acc[(size_t)(t[u] - n_lo) * D + c0] += v[u];
      }
      for (; e < E; ++e) {
        const int t = tb[(int64_t)e * tgt_stride];
        if (t > 0 && t >= n_lo && t < n_hi) acc[(size_t)(t - n_lo) * D + c0] += mb[(int64_t)e * D + c0];
      }
    } else {
      for (int e = 0; e < E; ++e) {
        const int t = tb[(int64_t)e * tgt_stride];
        if (t > 0 && t >= n_lo && t < n_hi)
          for (int i = c0; i < D; i += cols) acc[(size_t)(t - n_lo) * D + i] += mb[(int64_t)e * D + i];
      }
    }
    if (use_lds) {
      float* ab = agg + ((int64_t)b * N + n_lo) * D;
      for (int n = 0; n < n_hi - n_lo; ++n)
        for (int i = c0; i < D; i += cols) ab[(size_t)n * D + i] = acc[(size_t)n * D + i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// a7  GatedUpdate.call (models/layers.py:142-156), any D.  R rows per workgroup.
//   c=[h|agg]; z=sig(cWz+bz); r=sig(cWr+br); ht=tanh([r*h|agg]Wh+bh);
//   n=(1-z)h+z*ht; n=LN(n)*gamma+beta (eps, biased var); out=n+h.
// ---------------------------------------------------------------------------------------
// Thread (row group rg, column i) owns column i of RB rows (rg, rg + G, ...; G = blockDim / D row groups): every
// weight element it loads from L2 is used for RB rows, so a pass over the 3 x 2D x D kernels serves RB * G rows.
template <int RB>
__global__ void gated_update_kernel(const float* __restrict__ h, const float* __restrict__ agg,
                                    const float* __restrict__ Wz, const float* __restrict__ bz,
                                    const float* __restrict__ Wr, const float* __restrict__ br,
                                    const float* __restrict__ Wh, const float* __restrict__ bh,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float eps, float* __restrict__ out, int64_t rows, int D, int R) {
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;            // R*D
  float* as = hs + R * D;      // R*D
  float* zs = as + R * D;      // R*D
  float* rh = zs + R * D;      // R*D
  float* ns = rh + R * D;      // R*D
  float* st = ns + R * D;      // 2*R (mean, inv)
  const int G = blockDim.x / D;  // row groups; R == RB * G
  const int i = threadIdx.x % D, rg = threadIdx.x / D;
  const bool worker = rg < G;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const int nr = (int)((rows - row0) < R ? (rows - row0) : R);
  for (int t = threadIdx.x; t < R * D; t += blockDim.x) {
    const bool in = t < nr * D;
    hs[t] = in ? h[row0 * D + t] : 0.f;
    as[t] = in ? agg[row0 * D + t] : 0.f;
  }
  __syncthreads();
  if (worker) {
    float az[RB], ar[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      az[u] = bz[i];
      ar[u] = br[i];
    }
    for (int j = 0; j < D; ++j) {
      const float wz = Wz[(int64_t)j * D + i], wr = Wr[(int64_t)j * D + i];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const float x = hs[(rg + u * G) * D + j];
        az[u] = fmaf(x, wz, az[u]);
        ar[u] = fmaf(x, wr, ar[u]);
      }
    }
    for (int j = 0; j < D; ++j) {
      const float wz = Wz[(int64_t)(D + j) * D + i], wr = Wr[(int64_t)(D + j) * D + i];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const float x = as[(rg + u * G) * D + j];
        az[u] = fmaf(x, wz, az[u]);
        ar[u] = fmaf(x, wr, ar[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int t = (rg + u * G) * D + i;
      zs[t] = sigmoidf_(az[u]);
      rh[t] = sigmoidf_(ar[u]) * hs[t];
    }
  }
  __syncthreads();
  if (worker) {
    float ah[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) ah[u] = bh[i];
    for (int j = 0; j < D; ++j) {
      const float w = Wh[(int64_t)j * D + i];
#pragma unroll
      for (int u = 0; u < RB; ++u) ah[u] = fmaf(rh[(rg + u * G) * D + j], w, ah[u]);
    }
    for (int j = 0; j < D; ++j) {
      const float w = Wh[(int64_t)(D + j) * D + i];
#pragma unroll
      for (int u = 0; u < RB; ++u) ah[u] = fmaf(as[(rg + u * G) * D + j], w, ah[u]);
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int t = (rg + u * G) * D + i;
      const float z = zs[t];
      ns[t] = (1.0f - z) * hs[t] + z * tanhf(ah[u]);
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < nr; r += blockDim.x) {
    float mean = 0.f;
    for (int j = 0; j < D; ++j) mean += ns[r * D + j];
    mean /= (float)D;
    float var = 0.f;
    for (int j = 0; j < D; ++j) {
      const float d = ns[r * D + j] - mean;
      var = fmaf(d, d, var);
    }
    var /= (float)D;
    st[2 * r] = mean;
    st[2 * r + 1] = 1.0f / sqrtf(var + eps);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nr * D; t += blockDim.x) {
    const int r = t / D, c = t - r * D;
    out[row0 * D + t] = (ns[t] - st[2 * r]) * st[2 * r + 1] * gamma[c] + beta[c] + hs[t];
  }
}

// ---------------------------------------------------------------------------------------
// a7 for D = 32 on the matrix cores (exact f32 products, v_mfma_f32_16x16x4_f32): 16 atom rows per
// wave and iteration, atoms on the MFMA N dimension, features on M, so the gate accumulators are
// directly the operands of the candidate GEMM and of LayerNorm (same scheme as encoder_fused.hip).
// Weights are transposed into LDS once per workgroup ((gate, out) rows of 2D inputs, stride 68).
// ---------------------------------------------------------------------------------------
constexpr int kGuRS = 68;  // LDS row stride of the transposed gate kernels

__global__ __launch_bounds__(256) void gated_update_d32_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ out, int64_t rows,
    const int32_t* __restrict__ ridx, const int32_t* __restrict__ nrows_dev, float* __restrict__ save) {
  // save (optional; impnn_gated_update_rows_train): [z | r | tanh(t)] per row by list position, then r * h from float
  // 3 D max_rows on - see gated_update_wide16_kernel
  constexpr int D = 32;
  const int64_t max_rows = rows;  // what the launch and every buffer are sized for
  if (nrows_dev) {  // row list (impnn_gated_update_rows): rows ridx[0 .. *nrows_dev) only; never beyond the sizing
    const int64_t n = *nrows_dev;
    rows = n < 0 ? 0 : (n < max_rows ? n : max_rows);
  }
  __shared__ __align__(16) float wimg[3 * D * kGuRS + 5 * D];
  for (int t = threadIdx.x; t < 3 * D * 2 * D; t += blockDim.x) {
    const int gate = t / (2 * D * D), rem = t - gate * 2 * D * D;
    const int jj = rem / D, io = rem - jj * D;  // keras kernel (in=jj, out=io), read coalesced along io
    const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
    wimg[(gate * D + io) * kGuRS + jj] = Wg[rem];
  }
  for (int t = threadIdx.x; t < 5 * D; t += blockDim.x) {
    const int v = t / D, i = t - v * D;
    const float* src = v == 0 ? bz : v == 1 ? br : v == 2 ? bh : v == 3 ? gamma : beta;
    wimg[3 * D * kGuRS + t] = src[i];
  }
  __syncthreads();
  const float* wvec = wimg + 3 * D * kGuRS;
  const int lane = threadIdx.x & 63, a = lane & 15, q = lane >> 4;
  const int64_t ntiles = (rows + 15) >> 4;
  const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = wave_id; tile < ntiles; tile += nwaves) {
    const int64_t row = tile * 16 + a;
    int64_t rl = row < rows ? row : rows - 1;  // clamped load address; the store is masked
    if (ridx) {
      rl = ridx[rl];
      rl = rl < 0 ? 0 : (rl < max_rows ? rl : max_rows - 1);
    }
    const f32x4_t h0 = ldv4(h + rl * D + 4 * q), h1 = ldv4(h + rl * D + 16 + 4 * q);
    const f32x4_t a0 = ldv4(agg + rl * D + 4 * q), a1 = ldv4(agg + rl * D + 16 + 4 * q);
    f32x4_t z0 = ldv4(wvec + 4 * q), z1 = ldv4(wvec + 16 + 4 * q);
    f32x4_t r0 = ldv4(wvec + D + 4 * q), r1 = ldv4(wvec + D + 16 + 4 * q);
    f32x4_t t0 = ldv4(wvec + 2 * D + 4 * q), t1 = ldv4(wvec + 2 * D + 16 + 4 * q);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 32 * half + 16 * u + 4 * q;
        const f32x4_t Az0 = ldv4(wimg + (0 * D + a) * kGuRS + col), Az1 = ldv4(wimg + (0 * D + 16 + a) * kGuRS + col);
        const f32x4_t Ar0 = ldv4(wimg + (1 * D + a) * kGuRS + col), Ar1 = ldv4(wimg + (1 * D + 16 + a) * kGuRS + col);
        const f32x4_t Bv = half == 0 ? (u == 0 ? h0 : h1) : (u == 0 ? a0 : a1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z0 = mfma_f32(Az0[r], Bv[r], z0);
          z1 = mfma_f32(Az1[r], Bv[r], z1);
          r0 = mfma_f32(Ar0[r], Bv[r], r0);
          r1 = mfma_f32(Ar1[r], Bv[r], r1);
        }
      }
    }
    f32x4_t rh0, rh1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z0[i] = fsig(z0[i]);
      z1[i] = fsig(z1[i]);
      r0[i] = fsig(r0[i]);
      r1[i] = fsig(r1[i]);
      rh0[i] = r0[i] * h0[i];  // models/layers.py:149
      rh1[i] = r1[i] * h1[i];
    }
    if (save && row < rows) {
      float* sv = save + row * 3 * D + 4 * q;
      *reinterpret_cast<f32x4_t*>(sv) = z0;
      *reinterpret_cast<f32x4_t*>(sv + 16) = z1;
      *reinterpret_cast<f32x4_t*>(sv + D) = r0;
      *reinterpret_cast<f32x4_t*>(sv + D + 16) = r1;
      float* sr = save + max_rows * 3 * D + row * D + 4 * q;
      *reinterpret_cast<f32x4_t*>(sr) = rh0;
      *reinterpret_cast<f32x4_t*>(sr + 16) = rh1;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 32 * half + 16 * u + 4 * q;
        const f32x4_t Ah0 = ldv4(wimg + (2 * D + a) * kGuRS + col), Ah1 = ldv4(wimg + (2 * D + 16 + a) * kGuRS + col);
        const f32x4_t Bv = half == 0 ? (u == 0 ? rh0 : rh1) : (u == 0 ? a0 : a1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t0 = mfma_f32(Ah0[r], Bv[r], t0);
          t1 = mfma_f32(Ah1[r], Bv[r], t1);
        }
      }
    }
    f32x4_t n0, n1;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t0[i] = ftanh(t0[i]);
      t1[i] = ftanh(t1[i]);
      n0[i] = fmaf(z0[i], t0[i] - h0[i], h0[i]);  // (1-z) h + z tanh(.)
      n1[i] = fmaf(z1[i], t1[i] - h1[i], h1[i]);
      sum += n0[i] + n1[i];
    }
    if (save && row < rows) {
      float* sv = save + row * 3 * D + 2 * D + 4 * q;
      *reinterpret_cast<f32x4_t*>(sv) = t0;
      *reinterpret_cast<f32x4_t*>(sv + 16) = t1;
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.0f / D);
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
    const float inv = 1.0f / sqrtf(var * (1.0f / D) + eps);
    const f32x4_t g0 = ldv4(wvec + 3 * D + 4 * q), g1 = ldv4(wvec + 3 * D + 16 + 4 * q);
    const f32x4_t b0 = ldv4(wvec + 4 * D + 4 * q), b1 = ldv4(wvec + 4 * D + 16 + 4 * q);
    f32x4_t o0, o1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o0[i] = n0[i] * inv * g0[i] + b0[i] + h0[i];
      o1[i] = n1[i] * inv * g1[i] + b1[i] + h1[i];
    }
    if (row < rows) {  // (rl is the row itself, or its entry of the row list)
      *reinterpret_cast<f32x4_t*>(out + rl * D + 4 * q) = o0;
      *reinterpret_cast<f32x4_t*>(out + rl * D + 16 + 4 * q) = o1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// a7 for wider states (D a multiple of 16, 48 <= D <= 128) on the matrix cores, exact f32 products.
// A workgroup owns 64 rows (wave w: rows 16w..16w+15) for which c = [h | agg] and r*h stay in LDS; the gate kernels
// (3 x 2D x D floats: 393 KB at D = 128) stream through LDS in slices of 16 input rows.  Orientation
// out (rows x features) = c (rows x 2D) W (2D x D): rows on the MFMA M dimension, features on N, so the keras
// kernels are read as stored.  K index ordered 16u + 4q + r: one 16-byte LDS read of c feeds four steps.
// Row reductions of LayerNorm: 16-lane DPP rows (features) x the feature tiles in registers.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum_f(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
  return v;
}

template <int NT>  // NT = D / 16 feature tiles
__global__ __launch_bounds__(256) void gated_update_wide_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ out, int64_t rows) {
  constexpr int D = 16 * NT, LDC = 2 * D + 4, LDR = D + 4;
  constexpr int LDW = 2 * D;  // slice layout: element (input row 4*qq + r, column c) at ((qq * LDW + c) * 4 + r)
  extern __shared__ __align__(16) float smem[];
  float* cs = smem;                 // 64 x LDC : [h | agg]
  float* rhs = cs + 64 * LDC;       // 64 x LDR : r * h
  float* ws = rhs + 64 * LDR;       // 2 x 16 x LDW : slices of 16 input rows of the gate kernels, double-buffered
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 64;
  for (int t = tid; t < 64 * D; t += 256) {
    const int r = t / D, c = t - r * D;
    const bool in = row0 + r < rows;
    cs[r * LDC + c] = in ? h[(row0 + r) * D + c] : 0.f;
    cs[r * LDC + D + c] = in ? agg[(row0 + r) * D + c] : 0.f;
  }
  f32x4_t z[NT], rg[NT];
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    const float b0 = bz[16 * T + a], b1 = br[16 * T + a];
    z[T] = f32x4_t{b0, b0, b0, b0};
    rg[T] = f32x4_t{b1, b1, b1, b1};
  }
  const float* crow = cs + (16 * wave + a) * LDC + 4 * q;
  // kernel slices: global -> registers one slice ahead (in flight under the MFMAs) -> the other LDS buffer
  constexpr int kP1 = 16 * 2 * D / 256, kP2 = 16 * D / 256;
  float pre[kP1];
  auto fetch1 = [&](int u) {
#pragma unroll
    for (int i = 0; i < kP1; ++i) {
      const int t = tid + 256 * i, jj = t / (2 * D), c = t - jj * 2 * D;
      pre[i] = c < D ? Wz[(int64_t)(16 * u + jj) * D + c] : Wr[(int64_t)(16 * u + jj) * D + c - D];
    }
  };
  auto park1 = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < kP1; ++i) {
      const int t = tid + 256 * i, jj = t / (2 * D), c = t - jj * 2 * D;
      dst[(((jj >> 2) * LDW + c) << 2) + (jj & 3)] = pre[i];
    }
  };
  fetch1(0);
  park1(ws);
  __syncthreads();
  for (int u = 0; u < 2 * NT; ++u) {
    float* cur = ws + (u & 1) * 16 * LDW;
    float* nxt = ws + ((u + 1) & 1) * 16 * LDW;
    if (u + 1 < 2 * NT) fetch1(u + 1);
    const f32x4_t av = ldv4(crow + 16 * u);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const f32x4_t bzv = ldv4(cur + ((q * LDW + 16 * T + a) << 2)), brv = ldv4(cur + ((q * LDW + D + 16 * T + a) << 2));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[T] = mfma_f32(av[r], bzv[r], z[T]);
        rg[T] = mfma_f32(av[r], brv[r], rg[T]);
      }
    }
    if (u + 1 < 2 * NT) park1(nxt);  // nxt was last read two iterations ago: the barrier below orders it
    __syncthreads();
  }
  // z, r -> sigmoid; r * h into LDS (every wave only touches its own 16 rows)
#pragma unroll
  for (int T = 0; T < NT; ++T)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * wave + 4 * q + g, f = 16 * T + a;
      z[T][g] = sigmoidf_(z[T][g]);
      rhs[rl * LDR + f] = sigmoidf_(rg[T][g]) * cs[rl * LDC + f];
    }
  f32x4_t tt[NT];
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    const float b2 = bh[16 * T + a];
    tt[T] = f32x4_t{b2, b2, b2, b2};
  }
  const float* rrow = rhs + (16 * wave + a) * LDR + 4 * q;
  auto fetch2 = [&](int u) {
#pragma unroll
    for (int i = 0; i < kP2; ++i) {
      const int t = tid + 256 * i, jj = t / D, c = t - jj * D;
      pre[i] = Wh[(int64_t)(16 * u + jj) * D + c];
    }
  };
  auto park2 = [&](float* dst) {
#pragma unroll
    for (int i = 0; i < kP2; ++i) {
      const int t = tid + 256 * i, jj = t / D, c = t - jj * D;
      dst[(((jj >> 2) * LDW + c) << 2) + (jj & 3)] = pre[i];
    }
  };
  fetch2(0);
  park2(ws);
  __syncthreads();
  for (int u = 0; u < 2 * NT; ++u) {
    float* cur = ws + (u & 1) * 16 * LDW;
    float* nxt = ws + ((u + 1) & 1) * 16 * LDW;
    if (u + 1 < 2 * NT) fetch2(u + 1);
    const f32x4_t av = u < NT ? ldv4(rrow + 16 * u) : ldv4(crow + 16 * u);  // [r*h | agg]
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      const f32x4_t bv = ldv4(cur + ((q * LDW + 16 * T + a) << 2));
#pragma unroll
      for (int r = 0; r < 4; ++r) tt[T] = mfma_f32(av[r], bv[r], tt[T]);
    }
    if (u + 1 < 2 * NT) park2(nxt);
    __syncthreads();
  }
  // blend, LayerNorm over the D features of each row, residual
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < NT; ++T)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float hv = cs[(16 * wave + 4 * q + g) * LDC + 16 * T + a];
      const float n = (1.0f - z[T][g]) * hv + z[T][g] * tanhf(tt[T][g]);
      tt[T][g] = n;
      sum[g] += n;
    }
  float mean[4], inv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) mean[g] = row16_sum_f(sum[g]) * (1.0f / D);
  float var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < NT; ++T)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float d = tt[T][g] - mean[g];
      var[g] = fmaf(d, d, var[g]);
    }
#pragma unroll
  for (int g = 0; g < 4; ++g) inv[g] = 1.0f / sqrtf(row16_sum_f(var[g]) * (1.0f / D) + eps);
#pragma unroll
  for (int T = 0; T < NT; ++T) {
    const int f = 16 * T + a;
    const float gm = gamma[f], bt = beta[f];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * wave + 4 * q + g;
      if (row0 + rl < rows)
        out[(row0 + rl) * D + f] = (tt[T][g] - mean[g]) * inv[g] * gm + bt + cs[rl * LDC + f];
    }
  }
}

// The same with 16 waves (4 per SIMD, so LDS / MFMA latencies overlap): wave (row tile w & 3, feature group w >> 2)
// owns NT/4 feature tiles of 16 rows; LayerNorm's row sums are completed across the 4 feature groups through LDS.
template <int NT>  // NT = D / 16 feature tiles, NT % 4 == 0
__global__ __launch_bounds__(1024) void gated_update_wide16_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float* __restrict__ out, int64_t rows,
    const int32_t* __restrict__ ridx, const int32_t* __restrict__ nrows_dev, int tile_rows,
    float* __restrict__ save) {
  // save (optional; the training forward, impnn_gated_update_rows_train): what the backward would otherwise recompute
  // with two of its four GEMM passes, by LIST POSITION - [z | r | tanh(t)] in 3 D floats per row, and from float
  // 3 D max_rows on r * h (the layout gated_update_bwd_wide16_kernel keeps its pre-activation gradients in).
  // ridx / nrows_dev (optional): the kernel works on the rows ridx[0 .. *nrows_dev) of h / agg / out instead of on
  // rows [0, rows) - the model's layered path skips padding atoms this way (impnn_kept_row_index); the grid is
  // sized for `rows`, workgroups beyond the list leave at once.
  // tile_rows = 64, or 16 when the launch has too few rows to fill the chip with 64-row tiles (the reference trains
  // with 32 pairs per step: 1 280 rows = 20 tiles on 256 CUs): only the four waves of row tile 0 - one per SIMD -
  // multiply then, the others just help moving the weight slices, and a tile's MFMA time drops 4x.
  const int64_t max_rows = rows;  // what the launch and every buffer are sized for
  if (nrows_dev) {  // a stale or foreign device count never reaches beyond the sizing (indices are never trusted)
    const int64_t n = *nrows_dev;
    rows = n < 0 ? 0 : (n < max_rows ? n : max_rows);
  }
  constexpr int D = 16 * NT, LDC = 2 * D + 4, LDR = D + 4;
  constexpr int LDW = 2 * D;  // slice layout: element (input row 4*qq + r, column c) at ((qq * LDW + c) * 4 + r)
  extern __shared__ __align__(16) float smem[];
  float* cs = smem;                 // 64 x LDC : [h | agg]
  float* rhs = cs + 64 * LDC;       // 64 x LDR : r * h
  float* ws = rhs + 64 * LDR;       // 3 x 16 x LDW : slices of 16 input rows of the gate kernels, a ring of three
  constexpr int NL = NT / 4;  // feature tiles of this wave
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = lane & 15, q = lane >> 4;
  const int wave = wv >> 2, fg = wv & 3;  // row tile, feature group (the four waves of a row tile sit on four SIMDs)
  const bool act = 16 * wave < tile_rows;  // (wave-uniform) does this wave own rows of the tile?
  float* part = ws + 3 * 16 * LDW;  // 2 x 4 x 64 row partials (sum, squared deviation) of LayerNorm
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;
  if (row0 >= rows) return;
  {  // the tile of [h | agg]: 16-byte loads, all of a thread's requests in flight before the first LDS store
    constexpr int kQ = 64 * (D / 4) / 1024;  // quads of h (and of agg) per thread
    f32x4_t hv[kQ], av[kQ];
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      const int t = tid + 1024 * i, r = t / (D / 4), c4 = t - r * (D / 4);
      const bool in = row0 + r < rows && r < tile_rows;
      int64_t src = in ? (ridx ? (int64_t)ridx[row0 + r] : row0 + r) : 0;
      src = src < 0 ? 0 : (src < max_rows ? src : max_rows - 1);
      hv[i] = in ? ldv4(h + src * D + 4 * c4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
      av[i] = in ? ldv4(agg + src * D + 4 * c4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      const int t = tid + 1024 * i, r = t / (D / 4), c4 = t - r * (D / 4);
      *reinterpret_cast<f32x4_t*>(cs + r * LDC + 4 * c4) = hv[i];
      *reinterpret_cast<f32x4_t*>(cs + r * LDC + D + 4 * c4) = av[i];
    }
  }
  f32x4_t z[NL], rg[NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int T = fg * NL + TL;
    const float b0 = bz[16 * T + a], b1 = br[16 * T + a];
    z[TL] = f32x4_t{b0, b0, b0, b0};
    rg[TL] = f32x4_t{b1, b1, b1, b1};
  }
  const float* crow = cs + (16 * wave + a) * LDC + 4 * q;
  // Kernel slices: global -> registers TWO slices ahead -> a ring of three LDS buffers.  One slice ahead was not
  // enough: all 256 CUs stream the same 16 KB slice at the same time (one L2 channel group), a slice took ~5.5 K cycles
  // against 2 K of MFMA work, each iteration waiting out its own L2 round trip in front of the barrier.
  // A thread moves the four input rows 4qq .. 4qq+3 of one column: four coalesced 4-byte loads, ONE conflict-free 16-byte
  // LDS store (the slice layout keeps those four side by side).  Inside an iteration the order is: LDS reads of this
  // slice's operands, LDS store of the next slice, MFMAs - the four waves of a SIMD finish their MFMAs together, so a
  // store placed after them was fully exposed in front of the barrier (55 us of 439 at D = 128).
  constexpr int kI1 = 4 * 2 * D, kI2 = 4 * D;  // (qq, column) items of a slice: phase 1 [Wz | Wr], phase 2 Wh
  static_assert(kI1 <= 1024, "one item per thread");
  const int it_qq = tid / (2 * D), it_c = tid - it_qq * (2 * D);
  f32x4_t preA, preB;
  auto fetch1 = [&](int u, f32x4_t& pre) {
    if (tid < kI1) {
      const float* src = (it_c < D ? Wz + it_c : Wr + (it_c - D)) + (int64_t)(16 * u + 4 * it_qq) * D;
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[r] = src[r * D];
    }
  };
  auto park1 = [&](float* dst, const f32x4_t& pre) {
    if (tid < kI1) *reinterpret_cast<f32x4_t*>(dst + ((it_qq * LDW + it_c) << 2)) = pre;
  };
  struct Ops1 {
    f32x4_t av, bzv[NL], brv[NL];
  };
  auto read1 = [&](int u, Ops1& o) {
    const float* cur = ws + (u % 3) * 16 * LDW;
    o.av = ldv4(crow + 16 * u);
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const int T = fg * NL + TL;
      o.bzv[TL] = ldv4(cur + ((q * LDW + 16 * T + a) << 2));
      o.brv[TL] = ldv4(cur + ((q * LDW + D + 16 * T + a) << 2));
    }
  };
  auto mma1 = [&](const Ops1& o) {
    if (!act) return;
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z[TL] = mfma_f32(o.av[r], o.bzv[TL][r], z[TL]);
        rg[TL] = mfma_f32(o.av[r], o.brv[TL][r], rg[TL]);
      }
  };
  fetch1(0, preA);
  fetch1(1, preB);
  park1(ws, preA);  // slice 0
  __syncthreads();
  // iteration u: slice u is in LDS buffer u % 3, slice u + 1 in registers (requested one iteration ago), slice u + 2 is
  // requested; buffer (u + 1) % 3 was last read at iteration u - 2, two barriers ago
#ifdef IMPNN_DIAG_GU_NOLOOPS
  for (int u = 0; u < 0; u += 2) {
#else
  for (int u = 0; u < 2 * NT; u += 2) {
#endif
    Ops1 o;
    // (scheduling fences: left alone, the compiler sinks the store and the requests below the MFMAs again)
    if (u + 2 < 2 * NT) fetch1(u + 2, preA);          // preA was parked in the previous iteration (or above)
    read1(u, o);
    __builtin_amdgcn_sched_barrier(0);
    park1(ws + ((u + 1) % 3) * 16 * LDW, preB);
    __builtin_amdgcn_sched_barrier(0);
    mma1(o);
    __syncthreads();
    if (u + 3 < 2 * NT) fetch1(u + 3, preB);
    read1(u + 1, o);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < 2 * NT) park1(ws + ((u + 2) % 3) * 16 * LDW, preA);
    __builtin_amdgcn_sched_barrier(0);
    mma1(o);
    __syncthreads();
  }
  // z, r -> sigmoid; r * h into LDS (every wave only touches its own 16 rows)
  if (act) {
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * wave + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        z[TL][g] = fsig(z[TL][g]);
        const float rv = fsig(rg[TL][g]), rhv = rv * cs[rl * LDC + f];
        rhs[rl * LDR + f] = rhv;
        if (save && row0 + rl < rows) {
          float* sv = save + (row0 + rl) * 3 * D + f;
          sv[0] = z[TL][g];
          sv[D] = rv;
          save[max_rows * 3 * D + (row0 + rl) * D + f] = rhv;
        }
      }
  }
  f32x4_t tt[NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const float b2 = bh[16 * (fg * NL + TL) + a];
    tt[TL] = f32x4_t{b2, b2, b2, b2};
  }
  const float* rrow = rhs + (16 * wave + a) * LDR + 4 * q;
  const int it2_qq = tid / D, it2_c = tid - it2_qq * D;
  auto fetch2 = [&](int u, f32x4_t& pre) {
    if (tid < kI2) {
      const float* src = Wh + (int64_t)(16 * u + 4 * it2_qq) * D + it2_c;
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[r] = src[r * D];
    }
  };
  auto park2 = [&](float* dst, const f32x4_t& pre) {
    if (tid < kI2) *reinterpret_cast<f32x4_t*>(dst + ((it2_qq * LDW + it2_c) << 2)) = pre;
  };
  struct Ops2 {
    f32x4_t av, bv[NL];
  };
  auto read2 = [&](int u, Ops2& o) {
    const float* cur = ws + (u % 3) * 16 * LDW;
    o.av = u < NT ? ldv4(rrow + 16 * u) : ldv4(crow + 16 * u);  // [r*h | agg]
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) o.bv[TL] = ldv4(cur + ((q * LDW + 16 * (fg * NL + TL) + a) << 2));
  };
  auto mma2 = [&](const Ops2& o) {
    if (!act) return;
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int r = 0; r < 4; ++r) tt[TL] = mfma_f32(o.av[r], o.bv[TL][r], tt[TL]);
  };
  fetch2(0, preA);
  fetch2(1, preB);
  park2(ws, preA);
  __syncthreads();
#ifdef IMPNN_DIAG_GU_NOLOOPS
  for (int u = 0; u < 0; u += 2) {
#else
  for (int u = 0; u < 2 * NT; u += 2) {
#endif
    Ops2 o;
    if (u + 2 < 2 * NT) fetch2(u + 2, preA);
    read2(u, o);
    __builtin_amdgcn_sched_barrier(0);
    park2(ws + ((u + 1) % 3) * 16 * LDW, preB);
    __builtin_amdgcn_sched_barrier(0);
    mma2(o);
    __syncthreads();
    if (u + 3 < 2 * NT) fetch2(u + 3, preB);
    read2(u + 1, o);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < 2 * NT) park2(ws + ((u + 2) % 3) * 16 * LDW, preA);
    __builtin_amdgcn_sched_barrier(0);
    mma2(o);
    __syncthreads();
  }
  // blend, LayerNorm over the D features of each row (partials of the 4 feature groups meet in LDS), residual
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  if (act) {
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float hv = cs[(16 * wave + 4 * q + g) * LDC + 16 * (fg * NL + TL) + a];
        const float tv = ftanh(tt[TL][g]);
        const float n = (1.0f - z[TL][g]) * hv + z[TL][g] * tv;
        if (save && row0 + 16 * wave + 4 * q + g < rows)
          save[(row0 + 16 * wave + 4 * q + g) * 3 * D + 2 * D + 16 * (fg * NL + TL) + a] = tv;
        tt[TL][g] = n;
        sum[g] += n;
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = row16_sum_f(sum[g]);
      if (a == 0) part[fg * 64 + 16 * wave + 4 * q + g] = v;
    }
  }
  __syncthreads();
  float mean[4], inv[4];
  if (act) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * wave + 4 * q + g;
      mean[g] = ((part[rl] + part[64 + rl]) + (part[128 + rl] + part[192 + rl])) * (1.0f / D);
    }
    float var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float d = tt[TL][g] - mean[g];
        var[g] = fmaf(d, d, var[g]);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = row16_sum_f(var[g]);
      if (a == 0) part[256 + fg * 64 + 16 * wave + 4 * q + g] = v;
    }
  }
  __syncthreads();
  if (!act) return;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int rl = 256 + 16 * wave + 4 * q + g;
    inv[g] = 1.0f / sqrtf(((part[rl] + part[64 + rl]) + (part[128 + rl] + part[192 + rl])) * (1.0f / D) + eps);
  }
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = 16 * (fg * NL + TL) + a;
    const float gm = gamma[f], bt = beta[f];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * wave + 4 * q + g;
      if (row0 + rl < rows) {
        int64_t dst = ridx ? (int64_t)ridx[row0 + rl] : row0 + rl;
        dst = dst < 0 ? 0 : (dst < max_rows ? dst : max_rows - 1);
        out[dst * D + f] = (tt[TL][g] - mean[g]) * inv[g] * gm + bt + cs[rl * LDC + f];
      }
    }
  }
}

// a8  GlobalSumPool.call (models/layers.py:161-164)
__global__ void global_sum_pool_kernel(const float* __restrict__ h, const int32_t* __restrict__ ids,
                                       float* __restrict__ out, int B, int N, int D) {
  const int64_t total = (int64_t)B * D;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / D;
    const int i = (int)(t - b * D);
    float acc = 0.f;
    constexpr int kU = 8;
    int n = 0;
    for (; n + kU <= N; n += kU) {  // 8 rows in flight; the sum stays in ascending n
      float v[kU];
      int id[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        v[u] = h[(b * N + n + u) * D + i];
        id[u] = ids[b * N + n + u];
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (id[u] > 0) acc += v[u];
    }
    for (; n < N; ++n)
      if (ids[b * N + n] > 0) acc += h[(b * N + n) * D + i];
    out[t] = acc;
  }
}

__global__ void validate_indices_kernel(const int32_t* conn, const int32_t* atom_ids,
                                        const int32_t* bond_ids, int32_t* counts, int64_t n_conn,
                                        int64_t n_atom, int64_t n_bond, int N, int Va, int Vb) {
  int c0 = 0, c1 = 0, c2 = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (conn)
    for (int64_t t = t0; t < n_conn; t += stride) c0 += (unsigned)conn[t] >= (unsigned)N;
  if (atom_ids)
    for (int64_t t = t0; t < n_atom; t += stride) c1 += (unsigned)atom_ids[t] >= (unsigned)Va;
  if (bond_ids)
    for (int64_t t = t0; t < n_bond; t += stride) c2 += (unsigned)bond_ids[t] >= (unsigned)Vb;
  if (c0) atomicAdd(&counts[0], c0);
  if (c1) atomicAdd(&counts[1], c1);
  if (c2) atomicAdd(&counts[2], c2);
}

// ---------------------------------------------------------------------------------------
// f1  model head after GlobalSumPool, one launch (train_viscosity.py:189,197-214 + models/layers.py:10-49;
// train_melting_point.py:173,191-198).
//   fp_g  = relu(pooled_g @ Wfp_g + bfp_g)         (D -> F)     g in {cat, an}
//   mixed = relu(fp_cat @ Wp_cat + bp_cat) + relu(fp_an @ Wp_an + bp_an)      (F -> Mx)
//   kind 0: vp = mixed @ Wv + bv (Mx -> 3); A = vp0; Bc = clip(softplus(vp1), 0, 20);
//           Cc = clip(softplus(vp2), 0.1, 50); out = A + Bc / (T/100 + Cc + 1e-6)
//   kind 1: out = relu(mixed @ Wh + bh) @ Wo + bo   (Mx -> F -> 1)
// ---------------------------------------------------------------------------------------
constexpr int kHeadMaxDim = 64;   // fp_size, mixing_size
constexpr int kHeadMaxX = 128;    // pooled width (atom_dim 128: train_viscosity.py with a wider encoder)

__device__ __forceinline__ float softplus_exact(float x) { return x > 20.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }

// 8 samples per 256-thread workgroup, 32 threads per sample: thread (s, jj) owns outputs jj, jj+32 of
// every layer; the sample's vectors and all weights sit in LDS (13.6 KB of weights at the defaults).
constexpr int kHeadSPB = 8;

__global__ __launch_bounds__(256) void model_head_kernel(int kind, const float* __restrict__ pc,
                                                         const float* __restrict__ pa, const float* __restrict__ T,
                                                         const float* __restrict__ w, float* __restrict__ out, int B, int D,
                                                         int F, int Mx, int wfloats) {
  extern __shared__ __align__(16) float hsm[];
  float* ws = hsm;                                   // all head weights
  float* xs = ws + ((wfloats + 3) & ~3);             // [kHeadSPB][2][kHeadMaxX] pooled rows
  float* fp = xs + kHeadSPB * 2 * kHeadMaxX;         // [kHeadSPB][2][kHeadMaxDim]
  float* mix = fp + kHeadSPB * 2 * kHeadMaxDim;      // [kHeadSPB][kHeadMaxDim]
  float* hid = mix + kHeadSPB * kHeadMaxDim;         // [kHeadSPB][kHeadMaxDim]
  const int tid = threadIdx.x, sl = tid >> 5, jj = tid & 31;
  const int b = blockIdx.x * kHeadSPB + sl;
  const bool live = b < B;
  for (int t = tid; t < wfloats; t += blockDim.x) ws[t] = w[t];
  for (int g = 0; g < 2; ++g)
    for (int i = jj; i < D; i += 32) xs[(sl * 2 + g) * kHeadMaxX + i] = live ? (g == 0 ? pc : pa)[(int64_t)b * D + i] : 0.f;
  __syncthreads();
  const float* Wfp[2] = {ws, ws + D * F + F};
  const float* wp = ws + 2 * (D * F + F);
  const float* Wp[2] = {wp, wp + F * Mx + Mx};
  const float* wt = wp + 2 * (F * Mx + Mx);
  for (int g = 0; g < 2; ++g)
    for (int j = jj; j < F; j += 32) {
      float acc = Wfp[g][D * F + j];
      const float* x = xs + (sl * 2 + g) * kHeadMaxX;
      for (int i = 0; i < D; ++i) acc = fmaf(x[i], Wfp[g][i * F + j], acc);
      fp[(sl * 2 + g) * kHeadMaxDim + j] = fmaxf(acc, 0.f);
    }
  __syncthreads();
  for (int j = jj; j < Mx; j += 32) {
    float m = 0.f;
    for (int g = 0; g < 2; ++g) {
      float acc = Wp[g][F * Mx + j];
      const float* x = fp + (sl * 2 + g) * kHeadMaxDim;
      for (int i = 0; i < F; ++i) acc = fmaf(x[i], Wp[g][i * Mx + j], acc);
      m += fmaxf(acc, 0.f);  // AddTwoTensors / keras Add
    }
    mix[sl * kHeadMaxDim + j] = m;
  }
  __syncthreads();
  const float* mx = mix + sl * kHeadMaxDim;
  if (kind == 0) {
    if (jj < 3) {
      float acc = wt[Mx * 3 + jj];
      for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], wt[i * 3 + jj], acc);
      hid[sl * kHeadMaxDim + jj] = acc;
    }
    __syncthreads();
    if (jj == 0 && live) {
      const float* vp = hid + sl * kHeadMaxDim;
      const float Bc = fminf(fmaxf(softplus_exact(vp[1]), 0.f), 20.f);
      const float Cc = fminf(fmaxf(softplus_exact(vp[2]), 0.1f), 50.f);
      out[b] = vp[0] + Bc / (T[b] / 100.0f + Cc + 1e-6f);
    }
  } else {
    const float* Wh = wt;
    const float* bh = Wh + Mx * F;
    const float* Wo = bh + F;
    for (int j = jj; j < F; j += 32) {
      float acc = bh[j];
      for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], Wh[i * F + j], acc);
      hid[sl * kHeadMaxDim + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (jj == 0 && live) {
      float acc = Wo[F];
      for (int j = 0; j < F; ++j) acc = fmaf(hid[sl * kHeadMaxDim + j], Wo[j], acc);
      out[b] = acc;
    }
  }
}

inline int grid_for(int64_t items, int block = kBlock, int cap = 256 * 8) {
  int64_t g = (items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

constexpr size_t kMaxLds = 160 * 1024;

}  // namespace

int launch_embed_gather(const int32_t* ids, const float* table, float* out, int64_t rows, int vocab,
                        int dim, hipStream_t s) {
  if (rows == 0) return IMPNN_OK;
  if (dim % 4 == 0 && aligned16(table) && aligned16(out)) {
    embed_gather_kernel<4><<<grid_for(rows * (dim / 4)), kBlock, 0, s>>>(ids, table, out, rows, vocab, dim);
  } else {
    embed_gather_kernel<1><<<grid_for(rows * dim), kBlock, 0, s>>>(ids, table, out, rows, vocab, dim);
  }
  return check_launch("embed_gather");
}

int launch_bmm_message(const float* h, const float* bs, const int32_t* conn, const float* W, float* m,
                       float* agg, int B, int N, int E, int D, int K, hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  size_t hbytes = (size_t)N * D * sizeof(float);
  size_t mbytes = agg ? (size_t)E * D * sizeof(float) : 0;
  int h_in_lds = (hbytes + mbytes) <= kMaxLds;
  size_t lds = (h_in_lds ? hbytes : 0) + mbytes;
  if (lds > kMaxLds) return fail(IMPNN_E_UNSUPPORTED, "bmm_fused: E*D=%d floats exceed LDS", E * D);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)bmm_message_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  bmm_message_kernel<<<B, kBlock, lds, s>>>(h, bs, conn, W, m, agg, N, E, D, K, h_in_lds);
  return check_launch("bmm_message");
}

int launch_bond_type_matrices(const float* tb, const float* W, float* out, int Vb, int K, int D,
                              hipStream_t s) {
  const int DD = D * D;
  if (K >= 64) {  // K = D*D (train_melting_point.py:146): a real GEMM, out (Vb x DD) = Tb (Vb x K) W (K x DD)
    const bool al = (reinterpret_cast<uintptr_t>(tb) & 15u) == 0;
    if (K % 64 == 0 && DD % 16 == 0 && Vb >= 1 && Vb <= 128 && al) {  // 79 -> ~10 us at Vb = 72, K = DD = 1024
      const int vt = (Vb + 15) / 16;
      const dim3 grid(DD / 16);
      if (vt <= 2) bond_type_matrices_mfma_kernel<2><<<grid, 256, 0, s>>>(tb, W, out, Vb, K, DD);
      else if (vt <= 5) bond_type_matrices_mfma_kernel<5><<<grid, 256, 0, s>>>(tb, W, out, Vb, K, DD);
      else bond_type_matrices_mfma_kernel<8><<<grid, 256, 0, s>>>(tb, W, out, Vb, K, DD);
      return check_launch("bond_type_matrices_mfma");
    }
    return launch_strided_gemm(tb, W, out, K, Vb, DD, 1, K, DD, 1, s);
  }
  dim3 grid((DD + kBlock - 1) / kBlock, Vb);
  bond_type_matrices_kernel<<<grid, kBlock, 0, s>>>(tb, W, out, Vb, K, DD);
  return check_launch("bond_type_matrices");
}

int launch_bmm_message_typed(const float* h, const int32_t* bond_ids, const int32_t* conn,
                             const float* type_mats, float* m, int B, int N, int E, int D, int Vb,
                             hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  if (D == 32 && Vb <= kTVb && (int64_t)kTM * E <= kTSlots && E < (1 << 15) && N < (1 << 12) && aligned16(h) &&
      aligned16(type_mats) && aligned16(m) && (reinterpret_cast<uintptr_t>(conn) & 7u) == 0) {
    int tm = B / 512;  // small batches (training with 32): fewer molecules per workgroup, more workgroups
    tm = tm < 1 ? 1 : (tm > kTM ? kTM : tm);
    bmm_message_typed_d32_kernel<<<(B + tm - 1) / tm, 512, 0, s>>>(h, bond_ids, conn, type_mats, m, B, N, E, Vb, tm);
    return check_launch("bmm_message_typed_d32");
  }
  size_t lds = (size_t)N * D * sizeof(float);
  if (lds > kMaxLds) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed: N*D=%d floats exceed LDS", N * D);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)bmm_message_typed_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  bmm_message_typed_kernel<<<B, kBlock, lds, s>>>(h, bond_ids, conn, type_mats, m, N, E, D, Vb);
  return check_launch("bmm_message_typed");
}

// a5 for small batches (training at the reference's batch 32): one molecule per workgroup and P thread groups per
// column, group r owning the targets t with t % P == r.  The valid edges are first dealt into P lists in edge-slot
// order (wave ballots give every edge its rank), then thread (r, column) walks list r with 16 message rows in flight.
// Every (target, column) sum is formed by one thread in edge-slot order: bitwise equal to reduce_scatter_kernel.
constexpr int kRsP = 8;
__global__ __launch_bounds__(256) void reduce_scatter_small_kernel(const float* __restrict__ m,
                                                                   const int32_t* __restrict__ tgt, int tgt_stride,
                                                                   float* __restrict__ agg, int N, int E, int D,
                                                                   int P) {
  extern __shared__ __align__(16) float smem[];
  float* acc = smem;                                            // N*D
  uint16_t* list = reinterpret_cast<uint16_t*>(acc + (size_t)N * D);  // P lists of <= E edge slots
  uint16_t* tg = list + (size_t)P * E;                          // target row of every edge slot (0: skipped)
  __shared__ int cnt[kRsP][4], len[kRsP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* mb = m + (int64_t)b * E * D;
  const int32_t* tb = tgt + (int64_t)b * E * tgt_stride;
  for (int i = tid; i < N * D; i += 256) acc[i] = 0.f;
  if (tid < kRsP) len[tid] = 0;
  __syncthreads();
  for (int e0 = 0; e0 < E; e0 += 256) {
    const int e = e0 + tid;
    const int t = e < E ? tb[(int64_t)e * tgt_stride] : 0;
    const bool ok = t > 0 && t < N;
    if (e < E) tg[e] = (uint16_t)(ok ? t : 0);
    const int re = ok ? t % P : -1;
    int rank = 0;
    for (int r = 0; r < P; ++r) {
      const unsigned long long mask = __ballot(re == r);
      if (re == r) rank = __popcll(mask & ((1ull << lane) - 1ull));
      if (lane == 0) cnt[r][wave] = __popcll(mask);
    }
    __syncthreads();
    if (ok) {
      int base = len[re];
      for (int w = 0; w < wave; ++w) base += cnt[re][w];
      list[(size_t)re * E + base + rank] = (uint16_t)e;
    }
    __syncthreads();
    if (tid < P) len[tid] += cnt[tid][0] + cnt[tid][1] + cnt[tid][2] + cnt[tid][3];
    __syncthreads();
  }
  const int r = tid / D, c = tid - r * D;
  if (r < P) {
    const int n = len[r];
    const uint16_t* L = list + (size_t)r * E;
    for (int k = 0; k < n; k += 16) {
      float v[16];
      int tt[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        tt[u] = -1;
        v[u] = 0.f;
        if (k + u < n) {
          const int e = L[k + u];
          v[u] = mb[(int64_t)e * D + c];
          tt[u] = tg[e];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (tt[u] >= 0) acc[(size_t)tt[u] * D + c] += v[u];
    }
  }
  __syncthreads();
  float* ab = agg + (int64_t)b * N * D;
  for (int i = tid; i < N * D; i += 256) ab[i] = acc[i];
}

int launch_reduce_scatter_add(const float* m, const int32_t* tgt, int tgt_stride, float* agg, int B,
                              int N, int E, int D, hipStream_t s, int accumulate) {
  if (B == 0) return IMPNN_OK;
  if (!accumulate && B < 2048 && D <= 128 && E > 0 && E < 65536 && N < 65536) {  // too few (molecule, column) threads to hide latency
    int P = 256 / D;
    P = P > kRsP ? kRsP : P;
    const size_t l = sizeof(float) * (size_t)N * D + sizeof(uint16_t) * ((size_t)P * E + E);
    if (P >= 2 && l <= 64 * 1024) {
      reduce_scatter_small_kernel<<<B, 256, l, s>>>(m, tgt, tgt_stride, agg, N, E, D, P);
      return check_launch("reduce_scatter_small");
    }
  }
  const int cols = D < kBlock ? D : kBlock;
  const int mpb = kBlock / cols;
  // row ranges (gridDim.y): enough threads for ~2 waves per SIMD of the chip, ranges of at least 8 rows, and - whatever the
  // batch - ranges small enough that the accumulators of a workgroup's molecules fit 64 KB of LDS (several workgroups per CU)
  int K = 1;
  const int64_t threads = (int64_t)B * cols;
  if (threads < 131072) K = (int)((131072 + threads - 1) / threads);
  while ((size_t)mpb * ((N + K - 1) / K) * D * sizeof(float) > 64 * 1024 && (N + K - 1) / K > 8) ++K;
  if (K > 16) K = 16;
  if (K > N / 8) K = N / 8 > 0 ? N / 8 : 1;
  const int rows_per = (N + K - 1) / K;
  K = (N + rows_per - 1) / rows_per;
  size_t lds = (size_t)mpb * rows_per * D * sizeof(float);
  int use_lds = lds <= kMaxLds;
  if (!use_lds) lds = 0;
  const size_t lbytes = sizeof(int32_t) * ((size_t)mpb * E + mpb);
  const int use_list = cols == D && E > 0 && E < 65536 && N < 65536 && lbytes <= 32 * 1024 && lds + lbytes <= kMaxLds;
  if (use_list) lds += lbytes;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)reduce_scatter_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  reduce_scatter_kernel<<<dim3((B + mpb - 1) / mpb, K), kBlock, lds, s>>>(m, tgt, tgt_stride, agg, B, N, E, D, mpb,
                                                                        use_lds, accumulate, rows_per, use_list);
  return check_launch("reduce_scatter_add");
}

// ---------------------------------------------------------------------------------------
// Kept rows of a padded batch (the encoder's rule, encoder_plan.hip): molecule b keeps rows [0, r_b),
// r_b = 1 + max(last n with atom_ids[b,n] > 0, largest atom index on a valid edge); rows beyond can neither send a
// message nor be pooled.  One wave per molecule.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kept_rows_kernel(const int32_t* __restrict__ atom_ids,
                                                        const int32_t* __restrict__ bond_ids,
                                                        const int32_t* __restrict__ conn, int32_t* __restrict__ rows_out,
                                                        int B, int N, int E, int Vb) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  int r = 0;
  for (int n = lane; n < N; n += 64)
    if (atom_ids[(int64_t)b * N + n] > 0) r = n + 1;
  for (int e = lane; e < E; e += 64) {
    const int sv = conn[((int64_t)b * E + e) * 2], tv = conn[((int64_t)b * E + e) * 2 + 1];
    const int bid = bond_ids ? bond_ids[(int64_t)b * E + e] : 0;
    if (sv > 0 && tv > 0 && sv < N && tv < N && (unsigned)bid < (unsigned)Vb) {
      const int m = (sv > tv ? sv : tv) + 1;
      r = r > m ? r : m;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(r, o);
    r = r > t ? r : t;
  }
  if (lane == 0) rows_out[b] = r;
}

// row list: molecule b's kept rows b*N + [0, r_b) at positions start_b + [0, r_b); start = exclusive prefix of r
__global__ void row_index_fill_kernel(const int32_t* __restrict__ r, const int32_t* __restrict__ incl,
                                      int32_t* __restrict__ idx, int32_t* __restrict__ count, int B, int N) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) *count = B > 0 ? incl[B - 1] : 0;
  if (t >= (int64_t)B * N) return;
  const int b = (int)(t / N), n = (int)(t - (int64_t)b * N);
  if (n < r[b]) idx[incl[b] - r[b] + n] = (int32_t)t;
}

int launch_kept_rows(const int32_t* atom_ids, const int32_t* bond_ids, const int32_t* conn, int32_t* rows_out, int B,
                     int N, int E, int Vb, hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  kept_rows_kernel<<<(B + 3) / 4, 256, 0, s>>>(atom_ids, bond_ids, conn, rows_out, B, N, E, Vb);
  return check_launch("kept_rows");
}

int launch_row_index_fill(const int32_t* r, const int32_t* incl, int32_t* idx, int32_t* count, int B, int N,
                          hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  const int64_t n = (int64_t)B * N;
  row_index_fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(r, incl, idx, count, B, N);
  return check_launch("row_index_fill");
}

int launch_gated_update(const float* h, const float* agg, const float* Wz, const float* bz,
                        const float* Wr, const float* br, const float* Wh, const float* bh,
                        const float* gamma, const float* beta, float eps, float* out, int64_t rows,
                        int D, hipStream_t s, const int32_t* ridx, const int32_t* nrows_dev, float* save) {
  if (rows == 0) return IMPNN_OK;
  if (save && !(D == 32 || D == 64 || D == 128))
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_train: atom_dim %d (the saving forward covers 32, 64 and 128)", D);
  if (save && D == 32 && !(aligned16(h) && aligned16(agg) && aligned16(out) && aligned16(save)))
    return fail(IMPNN_E_BADARG, "gated_update_rows_train: needs 16-byte aligned tensors");
  if (ridx && !(D == 32 || (D % 64 == 0 && D <= 128)))
    return fail(IMPNN_E_UNSUPPORTED, "gated_update: a row list is supported for atom_dim 32, 64 and 128 only");
  if (ridx && D == 32 && !(aligned16(h) && aligned16(agg) && aligned16(out)))
    return fail(IMPNN_E_BADARG, "gated_update: row-list variant needs 16-byte aligned tensors");
  if (D == 32 && aligned16(h) && aligned16(agg) && aligned16(out)) {
    const int64_t tiles = (rows + 15) / 16;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;  // grid-stride: the weight transpose is paid once per workgroup
    gated_update_d32_kernel<<<(unsigned)blocks, 256, 0, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, out, rows,
                                                             ridx, nrows_dev, save);
    return check_launch("gated_update_d32");
  }
  if (D % 16 == 0 && D >= 48 && D <= 128) {  // matrix cores; the kernels stream through LDS in 16-row slices
    const size_t lw = sizeof(float) * ((size_t)64 * (2 * D + 4) + 64 * (D + 4) + 2 * 16 * (2 * D));
    const unsigned blocks = (unsigned)((rows + 63) / 64);
    // 16 waves / 64-row tiles fill the chip from ~8 K rows; below that 16-row tiles (four multiplying waves)
    const int tile_rows = gu_wide_tile_rows(rows);
    const unsigned blocks16 = (unsigned)((rows + tile_rows - 1) / tile_rows);
#define WIDE(NT_)                                                                                                  \
    do {                                                                                                            \
      if (lw > 48 * 1024)                                                                                           \
        (void)hipFuncSetAttribute((const void*)gated_update_wide_kernel<NT_>,                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);                             \
      gated_update_wide_kernel<NT_><<<blocks, 256, lw, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, out, rows); \
      return check_launch("gated_update_wide");                                                                     \
    } while (0)
    if (D % 64 == 0) {  // 16 waves per workgroup: 4 per SIMD
      const size_t l16 = lw + sizeof(float) * (512 + 16 * 2 * D);  // a third slice buffer + the LayerNorm partials
#define WIDE16(NT_)                                                                                                \
      do {                                                                                                          \
        (void)hipFuncSetAttribute((const void*)gated_update_wide16_kernel<NT_>,                                     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16);                            \
        gated_update_wide16_kernel<NT_><<<blocks16, 1024, l16, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, out, \
                                                                   rows, ridx, nrows_dev, tile_rows, save);         \
        return check_launch("gated_update_wide16");                                                                 \
      } while (0)
      if (D == 64) WIDE16(4);
      WIDE16(8);
#undef WIDE16
    }
    switch (D / 16) {
      case 3: WIDE(3);
      case 4: WIDE(4);
      case 5: WIDE(5);
      case 6: WIDE(6);
      case 7: WIDE(7);
      case 8: WIDE(8);
    }
#undef WIDE
  }
  if (D > kBlock) return fail(IMPNN_E_UNSUPPORTED, "gated_update: D=%d too large", D);
  const int G = kBlock / D;
  const bool big = D >= 64;  // more rows per weight pass where the kernels no longer sit in L1
  const int R = (big ? 8 : 4) * G;
  const size_t lds = ((size_t)5 * R * D + 2 * R) * sizeof(float);
  if (lds > kMaxLds) return fail(IMPNN_E_UNSUPPORTED, "gated_update: D=%d too large", D);
  const int64_t blocks = (rows + R - 1) / R;
  if (big) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gated_update_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    gated_update_kernel<8><<<(unsigned)blocks, kBlock, lds, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, out,
                                                                rows, D, R);
  } else {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gated_update_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    gated_update_kernel<4><<<(unsigned)blocks, kBlock, lds, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps, out,
                                                                rows, D, R);
  }
  return check_launch("gated_update");
}

int launch_global_sum_pool(const float* h, const int32_t* ids, float* out, int B, int N, int D,
                           hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  global_sum_pool_kernel<<<grid_for((int64_t)B * D), kBlock, 0, s>>>(h, ids, out, B, N, D);
  return check_launch("global_sum_pool");
}

int launch_model_head(int kind, const float* pc, const float* pa, const float* T, const float* w, float* out, int B,
                      int D, int F, int Mx, hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  if (D > kHeadMaxX || F > kHeadMaxDim || Mx > kHeadMaxDim)
    return fail(IMPNN_E_UNSUPPORTED, "model_head: dims D=%d (<= %d) F=%d Mx=%d (<= %d)", D, kHeadMaxX, F, Mx, kHeadMaxDim);
  const int wfloats = (int)impnn_model_head_floats(kind, D, F, Mx);
  const size_t lds = sizeof(float) * (((size_t)wfloats + 3) / 4 * 4 + (size_t)kHeadSPB * (2 * kHeadMaxX + 4 * kHeadMaxDim));
  // every (D <= 128, F <= 64, Mx <= 64) fits the 160 KB of a gfx950 CU: 29 057 weight floats + 16 KB of sample scratch
  if (lds > 160 * 1024) return fail(IMPNN_E_UNSUPPORTED, "model_head: weights do not fit LDS");
  if (lds > 64 * 1024)
    if (int rc = ensure_lds_limit((const void*)model_head_kernel, 8)) return rc;
  model_head_kernel<<<(B + kHeadSPB - 1) / kHeadSPB, 256, lds, s>>>(kind, pc, pa, T, w, out, B, D, F, Mx, wfloats);
  return check_launch("model_head");
}

int launch_validate_indices(const int32_t* conn, const int32_t* atom_ids, const int32_t* bond_ids,
                            int32_t* counts, int B, int N, int E, int Va, int Vb, hipStream_t s) {
  const int64_t n_conn = (int64_t)B * E * 2, n_atom = (int64_t)B * N, n_bond = (int64_t)B * E;
  validate_indices_kernel<<<grid_for(n_conn), kBlock, 0, s>>>(conn, atom_ids, bond_ids, counts, n_conn,
                                                             n_atom, n_bond, N, Va, Vb);
  return check_launch("validate_indices");
}

}  // namespace impnn
