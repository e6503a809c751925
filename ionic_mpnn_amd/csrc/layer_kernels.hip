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
// a5  Reduce.call (models/layers.py:57-83).  Thread (molecule, column) walks the E edge slots in
// order and adds into its own column of agg[b] - no atomics, bitwise equal to a sequential
// scatter_nd.  agg[b] lives in LDS when it fits, else directly in HBM/L2.
// ---------------------------------------------------------------------------------------
__global__ void reduce_scatter_kernel(const float* __restrict__ m, const int32_t* __restrict__ tgt,
                                      int tgt_stride, float* __restrict__ agg, int B, int N, int E,
                                      int D, int mols_per_block, int use_lds) {
  extern __shared__ __align__(16) float smem[];
  const int cols = D < (int)blockDim.x ? D : (int)blockDim.x;  // threads per molecule
  const int ml = threadIdx.x / cols;
  const int c0 = threadIdx.x - ml * cols;
  const int b = blockIdx.x * mols_per_block + ml;
  const bool active = ml < mols_per_block && b < B;
  float* acc = use_lds ? smem + (size_t)ml * N * D : (active ? agg + (int64_t)b * N * D : nullptr);
  if (active) {
    for (int n = 0; n < N; ++n)
      for (int i = c0; i < D; i += cols) acc[(size_t)n * D + i] = 0.f;
    const float* mb = m + (int64_t)b * E * D;
    const int32_t* tb = tgt + (int64_t)b * E * tgt_stride;
    for (int e = 0; e < E; ++e) {
      const int t = tb[(int64_t)e * tgt_stride];
      if (t > 0 && t < N)
        for (int i = c0; i < D; i += cols) acc[(size_t)t * D + i] += mb[(int64_t)e * D + i];
    }
    if (use_lds) {
      float* ab = agg + (int64_t)b * N * D;
      for (int n = 0; n < N; ++n)
        for (int i = c0; i < D; i += cols) ab[(size_t)n * D + i] = acc[(size_t)n * D + i];
    }
  }
}

// ---------------------------------------------------------------------------------------
// a7  GatedUpdate.call (models/layers.py:142-156), any D.  R rows per workgroup.
//   c=[h|agg]; z=sig(cWz+bz); r=sig(cWr+br); ht=tanh([r*h|agg]Wh+bh);
//   n=(1-z)h+z*ht; n=LN(n)*gamma+beta (eps, biased var); out=n+h.
// ---------------------------------------------------------------------------------------
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
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const int nr = (int)((rows - row0) < R ? (rows - row0) : R);
  for (int t = threadIdx.x; t < nr * D; t += blockDim.x) {
    hs[t] = h[row0 * D + t];
    as[t] = agg[row0 * D + t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nr * D; t += blockDim.x) {
    const int r = t / D, i = t - r * D;
    float az = bz[i], ar = br[i];
    for (int j = 0; j < D; ++j) {
      const float x = hs[r * D + j];
      az = fmaf(x, Wz[(int64_t)j * D + i], az);
      ar = fmaf(x, Wr[(int64_t)j * D + i], ar);
    }
    for (int j = 0; j < D; ++j) {
      const float x = as[r * D + j];
      az = fmaf(x, Wz[(int64_t)(D + j) * D + i], az);
      ar = fmaf(x, Wr[(int64_t)(D + j) * D + i], ar);
    }
    zs[t] = sigmoidf_(az);
    rh[t] = sigmoidf_(ar) * hs[t];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nr * D; t += blockDim.x) {
    const int r = t / D, i = t - r * D;
    float ah = bh[i];
    for (int j = 0; j < D; ++j) ah = fmaf(rh[r * D + j], Wh[(int64_t)j * D + i], ah);
    for (int j = 0; j < D; ++j) ah = fmaf(as[r * D + j], Wh[(int64_t)(D + j) * D + i], ah);
    const float z = zs[t];
    ns[t] = (1.0f - z) * hs[t] + z * tanhf(ah);
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
    const int r = t / D, i = t - r * D;
    out[row0 * D + t] = (ns[t] - st[2 * r]) * st[2 * r + 1] * gamma[i] + beta[i] + hs[t];
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
    for (int n = 0; n < N; ++n)
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
  dim3 grid((DD + kBlock - 1) / kBlock, Vb);
  bond_type_matrices_kernel<<<grid, kBlock, 0, s>>>(tb, W, out, Vb, K, DD);
  return check_launch("bond_type_matrices");
}

int launch_bmm_message_typed(const float* h, const int32_t* bond_ids, const int32_t* conn,
                             const float* type_mats, float* m, int B, int N, int E, int D, int Vb,
                             hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  size_t lds = (size_t)N * D * sizeof(float);
  if (lds > kMaxLds) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed: N*D=%d floats exceed LDS", N * D);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)bmm_message_typed_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  bmm_message_typed_kernel<<<B, kBlock, lds, s>>>(h, bond_ids, conn, type_mats, m, N, E, D, Vb);
  return check_launch("bmm_message_typed");
}

int launch_reduce_scatter_add(const float* m, const int32_t* tgt, int tgt_stride, float* agg, int B,
                              int N, int E, int D, hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  const int cols = D < kBlock ? D : kBlock;
  const int mpb = kBlock / cols;
  size_t lds = (size_t)mpb * N * D * sizeof(float);
  int use_lds = lds <= kMaxLds;
  if (!use_lds) lds = 0;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)reduce_scatter_kernel,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  reduce_scatter_kernel<<<(B + mpb - 1) / mpb, kBlock, lds, s>>>(m, tgt, tgt_stride, agg, B, N, E, D, mpb,
                                                                 use_lds);
  return check_launch("reduce_scatter_add");
}

int launch_gated_update(const float* h, const float* agg, const float* Wz, const float* bz,
                        const float* Wr, const float* br, const float* Wh, const float* bh,
                        const float* gamma, const float* beta, float eps, float* out, int64_t rows,
                        int D, hipStream_t s) {
  if (rows == 0) return IMPNN_OK;
  int R = kBlock / D;
  if (R < 1) R = 1;
  size_t lds = ((size_t)5 * R * D + 2 * R) * sizeof(float);
  if (lds > kMaxLds) return fail(IMPNN_E_UNSUPPORTED, "gated_update: D=%d too large", D);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)gated_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  const int64_t blocks = (rows + R - 1) / R;
  gated_update_kernel<<<(unsigned)blocks, kBlock, lds, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, beta, eps,
                                                           out, rows, D, R);
  return check_launch("gated_update");
}

int launch_global_sum_pool(const float* h, const int32_t* ids, float* out, int B, int N, int D,
                           hipStream_t s) {
  if (B == 0) return IMPNN_OK;
  global_sum_pool_kernel<<<grid_for((int64_t)B * D), kBlock, 0, s>>>(h, ids, out, B, N, D);
  return check_launch("global_sum_pool");
}

int launch_validate_indices(const int32_t* conn, const int32_t* atom_ids, const int32_t* bond_ids,
                            int32_t* counts, int B, int N, int E, int Va, int Vb, hipStream_t s) {
  const int64_t n_conn = (int64_t)B * E * 2, n_atom = (int64_t)B * N, n_bond = (int64_t)B * E;
  validate_indices_kernel<<<grid_for(n_conn), kBlock, 0, s>>>(conn, atom_ids, bond_ids, counts, n_conn,
                                                             n_atom, n_bond, N, Va, Vb);
  return check_launch("validate_indices");
}

}  // namespace impnn
