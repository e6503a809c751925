// Backward kernels of the layer-at-a-time path and the optimizer step: what Keras
// autodiff + Adam(1e-3, clipnorm=1.0) do for the reference's model.fit (train_viscosity.py:227-230,
// 328-338; train_melting_point.py:205-208).  One kernel per reference layer, the adjoint of the forward
// in layer_kernels.hip with the same masks (models/layers.py:114-115, :70 tgt > 0) and the same
// "out-of-range index == padding" rule.  Written for any D (VALU, f32); the parameter-gradient sums run
// over per-workgroup partial buffers that a second kernel adds in a fixed order, except where noted.
#include "common.h"

namespace impnn {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// a1/a2 backward: dtable[ids[r], :] += dout[r, :]   (float atomics: rows of one id meet in any order)
// ---------------------------------------------------------------------------------------
__global__ void embed_gather_bwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                        float* __restrict__ dtable, int64_t rows, int vocab, int dim) {
  const int64_t total = rows * dim;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / dim;
    const int c = (int)(t - r * dim);
    const int id = ids[r];
    if ((unsigned)id < (unsigned)vocab) atomicAdd(&dtable[(int64_t)id * dim + c], dout[t]);
  }
}

// ---------------------------------------------------------------------------------------
// a5 backward (models/layers.py:57-83): dmessages[b,e,:] = tgt > 0 ? dagg[b,tgt,:] : 0
// ---------------------------------------------------------------------------------------
__global__ void reduce_scatter_bwd_kernel(const float* __restrict__ dagg, const int32_t* __restrict__ tgt,
                                          int tgt_stride, float* __restrict__ dm, int64_t BE, int N, int E, int D) {
  const int64_t total = BE * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t be = t / D;
    const int c = (int)(t - be * D);
    const int64_t b = be / E;
    const int tg = tgt[be * tgt_stride];
    dm[t] = (tg > 0 && tg < N) ? dagg[(b * N + tg) * D + c] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// a8 backward (models/layers.py:161-164): dh[b,n,:] = atom_ids[b,n] > 0 ? dpooled[b,:] : 0
// ---------------------------------------------------------------------------------------
__global__ void global_sum_pool_bwd_kernel(const float* __restrict__ dp, const int32_t* __restrict__ ids,
                                           float* __restrict__ dh, int64_t BN, int N, int D) {
  const int64_t total = BN * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bn = t / D;
    const int c = (int)(t - bn * D);
    dh[t] = ids[bn] > 0 ? dp[(bn / N) * D + c] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// a4 backward in the per-bond-type schedule.  Forward: m[b,e,:] = A[type_e] h[b,src_e,:] on valid edges.
//   dh[b,src,:]  += A[type]^T dm[b,e,:]
//   dA[type,i,j] += dm[b,e,i] h[b,src,j]
// A workgroup takes kMol molecules, counting-sorts their valid edges by type in LDS, and walks the type
// runs: A[type] is staged in LDS once per run; dA of the run is summed in registers ((i,j) entries dealt
// over the threads) and leaves with one atomicAdd per entry and run; dh goes out with float atomics
// (several edges share a source row).  dh and dA must be zeroed by the caller.
// ---------------------------------------------------------------------------------------
constexpr int kMol = 32;
constexpr int kMaxSlots = 4096;   // kMol * E edge slots per workgroup
constexpr int kMaxTypes = 1024;

template <int ACC>  // ACC = ceil(D*D / kBlock) accumulators per thread
__global__ __launch_bounds__(kBlock) void bmm_message_typed_bwd_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ bond_ids, const int32_t* __restrict__ conn,
    const float* __restrict__ A, const float* __restrict__ dm, float* __restrict__ dh, float* __restrict__ dA,
    int B, int N, int E, int D, int Vb) {
  extern __shared__ __align__(16) float smem[];
  __shared__ int cnt[kMaxTypes + 1];
  __shared__ int order[kMaxSlots];
  float* As = smem;  // D*D
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * kMol;
  const int nb = min(kMol, B - b0);
  const int slots = nb * E;
  for (int t = tid; t <= Vb; t += kBlock) cnt[t] = 0;
  __syncthreads();
  auto valid_type = [&](int slot) -> int {
    const int64_t be = (int64_t)b0 * E + slot;
    const int src = conn[be * 2], tgt = conn[be * 2 + 1], ty = bond_ids[be];
    return (src > 0 && tgt > 0 && src < N && tgt < N && (unsigned)ty < (unsigned)Vb) ? ty : -1;
  };
  for (int s = tid; s < slots; s += kBlock) {
    const int ty = valid_type(s);
    if (ty >= 0) atomicAdd(&cnt[ty + 1], 1);
  }
  __syncthreads();
  if (tid == 0) {  // exclusive prefix over <= Vb+1 counters (Vb is small next to the edge work)
    int run = 0;
    for (int t = 1; t <= Vb; ++t) {
      const int c = cnt[t];
      cnt[t] = run;
      run += c;
    }
    cnt[0] = run;  // total valid edges
  }
  __syncthreads();
  const int total = cnt[0];
  __syncthreads();
  for (int s = tid; s < slots; s += kBlock) {
    const int ty = valid_type(s);
    if (ty >= 0) order[atomicAdd(&cnt[ty + 1], 1)] = s;  // cnt[ty+1] ends as the END of type ty's run
  }
  __syncthreads();
  const int DD = D * D;
  int pos = 0;
  while (pos < total) {  // workgroup-uniform walk over the type runs
    const int s0 = order[pos];
    const int ty = bond_ids[(int64_t)b0 * E + s0];
    const int end = cnt[ty + 1];
    for (int t = tid; t < DD; t += kBlock) As[t] = A[(int64_t)ty * DD + t];
    __syncthreads();
    // dh: thread (edge lane, column j)
    const int lanes = kBlock / D > 0 ? kBlock / D : 1;
    if (tid < lanes * D) {
      const int j = tid % D, el = tid / D;
      for (int p = pos + el; p < end; p += lanes) {
        const int s = order[p];
        const int64_t be = (int64_t)b0 * E + s;
        const int64_t b = be / E;
        const int src = conn[be * 2];
        const float* g = dm + be * D;
        float u = 0.f;
        for (int i = 0; i < D; ++i) u = fmaf(g[i], As[i * D + j], u);
        atomicAdd(&dh[(b * N + src) * D + j], u);
      }
    }
    // dA of this run
    float acc[ACC];
#pragma unroll
    for (int a = 0; a < ACC; ++a) acc[a] = 0.f;
    for (int p = pos; p < end; ++p) {
      const int s = order[p];
      const int64_t be = (int64_t)b0 * E + s;
      const int64_t b = be / E;
      const float* g = dm + be * D;
      const float* x = h + (b * N + conn[be * 2]) * D;
#pragma unroll
      for (int a = 0; a < ACC; ++a) {
        const int q = tid + a * kBlock;
        if (q < DD) acc[a] = fmaf(g[q / D], x[q % D], acc[a]);
      }
    }
#pragma unroll
    for (int a = 0; a < ACC; ++a) {
      const int q = tid + a * kBlock;
      if (q < DD) atomicAdd(&dA[(int64_t)ty * DD + q], acc[a]);
    }
    __syncthreads();
    pos = end;
  }
}

// ---------------------------------------------------------------------------------------
// schedule A backward: A[v] = sum_k Tb[v,k] W[k]  =>  dW[k] = sum_v Tb[v,k] dA[v];  dTb[v,k] = <dA[v], W[k]>
// ---------------------------------------------------------------------------------------
__global__ void bond_type_matrices_bwd_w_kernel(const float* __restrict__ tb, const float* __restrict__ dA,
                                                float* __restrict__ dW, int Vb, int K, int DD) {
  const int k = blockIdx.y;
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < DD; ij += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int v = 0; v < Vb; ++v) acc = fmaf(tb[(int64_t)v * K + k], dA[(int64_t)v * DD + ij], acc);
    dW[(int64_t)k * DD + ij] = acc;
  }
}
__global__ void bond_type_matrices_bwd_t_kernel(const float* __restrict__ W, const float* __restrict__ dA,
                                                float* __restrict__ dtb, int Vb, int K, int DD) {
  // one wave per (v,k)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= Vb * K) return;
  const int v = wave / K, k = wave - v * K;
  float acc = 0.f;
  for (int ij = lane; ij < DD; ij += 64) acc = fmaf(dA[(int64_t)v * DD + ij], W[(int64_t)k * DD + ij], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) dtb[(int64_t)v * K + k] = acc;
}

// ---------------------------------------------------------------------------------------
// a7 backward (models/layers.py:142-156).  Forward per row, c = [h|agg]:
//   z = sig(c Wz + bz); r = sig(c Wr + br); t = tanh([r*h|agg] Wh + bh); n = (1-z) h + z t;
//   x = (n - mean) * inv; out = gamma x + beta + h
// Persistent workgroups walk tiles of R = kBlock/D rows; intermediates are recomputed from (h, agg);
// every workgroup owns one slice of `partial` (P floats: dWz 2D*D | dbz | dWr | dbr | dWh | dbh | dgamma |
// dbeta, the canonical order) that it updates with plain read-modify-writes, and
// reduce_partials_kernel adds the slices in a fixed order: parameter gradients are bitwise reproducible.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gated_update_bwd_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma, float eps,
    const float* __restrict__ dout, float* __restrict__ dh, float* __restrict__ dagg,
    float* __restrict__ partial, int64_t rows, int D, int R) {
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;            // R*D each
  float* as = hs + R * D;
  float* zs = as + R * D;
  float* rs = zs + R * D;
  float* rhs = rs + R * D;     // r * h
  float* ts = rhs + R * D;     // tanh
  float* xs = ts + R * D;      // n, then x-hat
  float* g1 = xs + R * D;      // dx-hat, then dzp
  float* g2 = g1 + R * D;      // dx-hat * x-hat, then drp
  float* g3 = g2 + R * D;      // dtp
  float* st = g3 + R * D;      // 4*R: mean, inv, m1, m2
  const int tid = threadIdx.x;
  const int DD2 = 2 * D * D;
  const int P = 3 * (DD2 + D) + 2 * D;
  float* mine = partial + (int64_t)blockIdx.x * P;
  for (int q = tid; q < P; q += kBlock) mine[q] = 0.f;
  const int64_t ntile = (rows + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    const int nr = (int)((rows - row0) < R ? (rows - row0) : R);
    __syncthreads();
    for (int t = tid; t < nr * D; t += kBlock) {
      hs[t] = h[row0 * D + t];
      as[t] = agg[row0 * D + t];
    }
    __syncthreads();
    for (int t = tid; t < nr * D; t += kBlock) {
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
      const float z = sigmoid_exact(az), rr = sigmoid_exact(ar);
      zs[t] = z;
      rs[t] = rr;
      rhs[t] = rr * hs[t];
    }
    __syncthreads();
    for (int t = tid; t < nr * D; t += kBlock) {
      const int r = t / D, i = t - r * D;
      float ah = bh[i];
      for (int j = 0; j < D; ++j) ah = fmaf(rhs[r * D + j], Wh[(int64_t)j * D + i], ah);
      for (int j = 0; j < D; ++j) ah = fmaf(as[r * D + j], Wh[(int64_t)(D + j) * D + i], ah);
      const float tt = tanhf(ah);
      ts[t] = tt;
      xs[t] = (1.0f - zs[t]) * hs[t] + zs[t] * tt;
    }
    __syncthreads();
    for (int r = tid; r < nr; r += kBlock) {
      float mean = 0.f;
      for (int j = 0; j < D; ++j) mean += xs[r * D + j];
      mean /= (float)D;
      float var = 0.f;
      for (int j = 0; j < D; ++j) {
        const float d = xs[r * D + j] - mean;
        var = fmaf(d, d, var);
      }
      st[4 * r] = mean;
      st[4 * r + 1] = 1.0f / sqrtf(var / (float)D + eps);
    }
    __syncthreads();
    for (int t = tid; t < nr * D; t += kBlock) {
      const int r = t / D, i = t - r * D;
      const float xh = (xs[t] - st[4 * r]) * st[4 * r + 1];
      xs[t] = xh;
      const float dxh = dout[row0 * D + t] * gamma[i];
      g1[t] = dxh;
      g2[t] = dxh * xh;
    }
    __syncthreads();
    for (int r = tid; r < nr; r += kBlock) {
      float m1 = 0.f, m2 = 0.f;
      for (int j = 0; j < D; ++j) {
        m1 += g1[r * D + j];
        m2 += g2[r * D + j];
      }
      st[4 * r + 2] = m1 / (float)D;
      st[4 * r + 3] = m2 / (float)D;
    }
    // dgamma / dbeta of this tile (thread i < D owns column i)
    for (int i = tid; i < D; i += kBlock) {
      float dg = 0.f, db = 0.f;
      for (int r = 0; r < nr; ++r) {
        const float dy = dout[(row0 + r) * D + i];
        dg = fmaf(dy, xs[r * D + i], dg);
        db += dy;
      }
      mine[3 * (DD2 + D) + i] += dg;
      mine[3 * (DD2 + D) + D + i] += db;
    }
    __syncthreads();
    // dn -> (dzp, dtp), first part of dh
    for (int t = tid; t < nr * D; t += kBlock) {
      const int r = t / D;
      const float dn = st[4 * r + 1] * (g1[t] - st[4 * r + 2] - xs[t] * st[4 * r + 3]);
      const float z = zs[t], tt = ts[t];
      g1[t] = dn * (tt - hs[t]) * z * (1.0f - z);   // dzp
      g3[t] = dn * z * (1.0f - tt * tt);            // dtp
      xs[t] = dout[row0 * D + t] + dn * (1.0f - z);  // dh so far (x-hat is dead)
    }
    __syncthreads();
    // dc2 = dtp Wh^T: lower half -> through r*h, upper half -> dagg
    for (int t = tid; t < nr * D; t += kBlock) {
      const int r = t / D, i = t - r * D;
      float lo = 0.f, hi = 0.f;
      for (int j = 0; j < D; ++j) {
        const float d = g3[r * D + j];
        lo = fmaf(d, Wh[(int64_t)i * D + j], lo);
        hi = fmaf(d, Wh[(int64_t)(D + i) * D + j], hi);
      }
      const float rr = rs[t];
      g2[t] = lo * hs[t] * rr * (1.0f - rr);  // drp
      xs[t] += lo * rr;
      zs[t] = hi;                               // dagg so far (z is dead)
    }
    __syncthreads();
    // dc = dzp Wz^T + drp Wr^T
    for (int t = tid; t < nr * D; t += kBlock) {
      const int r = t / D, i = t - r * D;
      float lo = 0.f, hi = 0.f;
      for (int j = 0; j < D; ++j) {
        const float dz = g1[r * D + j], dr = g2[r * D + j];
        lo = fmaf(dz, Wz[(int64_t)i * D + j], lo);
        lo = fmaf(dr, Wr[(int64_t)i * D + j], lo);
        hi = fmaf(dz, Wz[(int64_t)(D + i) * D + j], hi);
        hi = fmaf(dr, Wr[(int64_t)(D + i) * D + j], hi);
      }
      dh[row0 * D + t] = xs[t] + lo;
      dagg[row0 * D + t] = zs[t] + hi;
    }
    // parameter gradients of this tile: dW_g[i'][j] += sum_r in_g[r][i'] dpre_g[r][j]
    for (int q = tid; q < DD2; q += kBlock) {
      const int ip = q / D, j = q - ip * D;
      float az = 0.f, ar = 0.f, ah = 0.f;
      for (int r = 0; r < nr; ++r) {
        const float c = ip < D ? hs[r * D + ip] : as[r * D + ip - D];
        const float c2 = ip < D ? rhs[r * D + ip] : c;
        az = fmaf(c, g1[r * D + j], az);
        ar = fmaf(c, g2[r * D + j], ar);
        ah = fmaf(c2, g3[r * D + j], ah);
      }
      mine[q] += az;
      mine[(DD2 + D) + q] += ar;
      mine[2 * (DD2 + D) + q] += ah;
    }
    for (int j = tid; j < D; j += kBlock) {
      float sz = 0.f, sr = 0.f, sh = 0.f;
      for (int r = 0; r < nr; ++r) {
        sz += g1[r * D + j];
        sr += g2[r * D + j];
        sh += g3[r * D + j];
      }
      mine[DD2 + j] += sz;
      mine[(DD2 + D) + DD2 + j] += sr;
      mine[2 * (DD2 + D) + DD2 + j] += sh;
    }
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int P) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < P; q += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < nblk; ++b) acc += partial[(int64_t)b * P + q];
    out[q] = acc;
  }
}

// ---------------------------------------------------------------------------------------
// Optimizer step: keras.optimizers.Adam(lr, clipnorm) as the trainers configure it
// (train_viscosity.py:227-230).  One workgroup per variable:
//   g <- g * clipnorm / max(||g||_2, clipnorm)          (tf.clip_by_norm, per variable; clipnorm <= 0: off)
//   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2
//   w <- w - lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
// The variable table holds device pointers: 4 per variable (w, g, m, v) and the element count.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void adam_clipnorm_kernel(const unsigned long long* __restrict__ table,
                                                             const long long* __restrict__ sizes, float lr,
                                                             float b1, float b2, float eps, float clipnorm,
                                                             float corr1, float corr2) {
  __shared__ float red[16];
  __shared__ float scale_s;
  const int var = blockIdx.x;
  float* w = reinterpret_cast<float*>(table[4 * var + 0]);
  const float* g = reinterpret_cast<const float*>(table[4 * var + 1]);
  float* m = reinterpret_cast<float*>(table[4 * var + 2]);
  float* v = reinterpret_cast<float*>(table[4 * var + 3]);
  const long long n = sizes[var];
  float ss = 0.f;
  for (long long t = threadIdx.x; t < n; t += blockDim.x) ss = fmaf(g[t], g[t], ss);
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    const float norm = sqrtf(tot);
    scale_s = clipnorm > 0.f ? clipnorm / fmaxf(norm, clipnorm) : 1.0f;
  }
  __syncthreads();
  const float sc = scale_s;
  const float alpha = lr * sqrtf(corr2) / corr1;  // corr1 = 1 - b1^t, corr2 = 1 - b2^t
  for (long long t = threadIdx.x; t < n; t += blockDim.x) {
    const float gg = g[t] * sc;
    const float mm = b1 * m[t] + (1.0f - b1) * gg;
    const float vv = b2 * v[t] + (1.0f - b2) * gg * gg;
    m[t] = mm;
    v[t] = vv;
    w[t] -= alpha * mm / (sqrtf(vv) + eps);
  }
}

inline int grid_for(int64_t items, int cap = 256 * 8) {
  int64_t g = (items + kBlock - 1) / kBlock;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace

int launch_embed_gather_bwd(const int32_t* ids, const float* dout, float* dtable, int64_t rows, int vocab, int dim,
                            hipStream_t s) {
  embed_gather_bwd_kernel<<<grid_for(rows * dim), kBlock, 0, s>>>(ids, dout, dtable, rows, vocab, dim);
  return check_launch("embed_gather_bwd");
}

int launch_reduce_scatter_bwd(const float* dagg, const int32_t* tgt, int tgt_stride, float* dm, int B, int N, int E,
                              int D, hipStream_t s) {
  reduce_scatter_bwd_kernel<<<grid_for((int64_t)B * E * D), kBlock, 0, s>>>(dagg, tgt, tgt_stride, dm,
                                                                              (int64_t)B * E, N, E, D);
  return check_launch("reduce_scatter_bwd");
}

int launch_global_sum_pool_bwd(const float* dp, const int32_t* ids, float* dh, int B, int N, int D, hipStream_t s) {
  global_sum_pool_bwd_kernel<<<grid_for((int64_t)B * N * D), kBlock, 0, s>>>(dp, ids, dh, (int64_t)B * N, N, D);
  return check_launch("global_sum_pool_bwd");
}

int launch_bmm_message_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn, const float* A,
                                 const float* dm, float* dh, float* dA, int B, int N, int E, int D, int Vb,
                                 hipStream_t s) {
  if ((int64_t)kMol * E > kMaxSlots) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_bwd: E=%d too large", E);
  if (Vb > kMaxTypes) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_bwd: Vb=%d too large", Vb);
  if (D > 128) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_bwd: D=%d > 128", D);
  const int grid = (B + kMol - 1) / kMol;
  const size_t lds = sizeof(float) * (size_t)D * D;
  const int acc = (D * D + kBlock - 1) / kBlock;
#define LAUNCH(ACC)                                                                                      \
  do {                                                                                                   \
    if (lds > 48 * 1024)                                                                                 \
      (void)hipFuncSetAttribute((const void*)bmm_message_typed_bwd_kernel<ACC>,                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    bmm_message_typed_bwd_kernel<ACC><<<grid, kBlock, lds, s>>>(h, bond_ids, conn, A, dm, dh, dA, B, N, E, D, Vb); \
  } while (0)
  if (acc <= 1) LAUNCH(1);
  else if (acc <= 4) LAUNCH(4);
  else if (acc <= 16) LAUNCH(16);
  else LAUNCH(64);
#undef LAUNCH
  return check_launch("bmm_message_typed_bwd");
}

int launch_bond_type_matrices_bwd(const float* tb, const float* W, const float* dA, float* dW, float* dtb, int Vb,
                                  int K, int D, hipStream_t s) {
  const int DD = D * D;
  bond_type_matrices_bwd_w_kernel<<<dim3((DD + kBlock - 1) / kBlock, K), kBlock, 0, s>>>(tb, dA, dW, Vb, K, DD);
  if (int rc = check_launch("bond_type_matrices_bwd_w")) return rc;
  const int64_t waves = (int64_t)Vb * K;
  bond_type_matrices_bwd_t_kernel<<<(int)((waves * 64 + kBlock - 1) / kBlock), kBlock, 0, s>>>(W, dA, dtb, Vb, K, DD);
  return check_launch("bond_type_matrices_bwd_t");
}

int gated_update_bwd_blocks(int64_t rows, int D) {
  const int R = kBlock / D > 0 ? kBlock / D : 1;
  const int64_t ntile = (rows + R - 1) / R;
  return (int)(ntile < 256 ? (ntile < 1 ? 1 : ntile) : 256);
}

int64_t gated_update_param_floats(int D) { return 3 * ((int64_t)2 * D * D + D) + 2 * D; }

int launch_gated_update_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                            const float* br, const float* Wh, const float* bh, const float* gamma, float eps,
                            const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                            int64_t rows, int D, hipStream_t s) {
  const int R = kBlock / D > 0 ? kBlock / D : 1;
  const int nblk = gated_update_bwd_blocks(rows, D);
  const int P = (int)gated_update_param_floats(D);
  const size_t lds = sizeof(float) * ((size_t)10 * R * D + 4 * R);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)gated_update_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  gated_update_bwd_kernel<<<nblk, kBlock, lds, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh, dagg,
                                                    workspace, rows, D, R);
  if (int rc = check_launch("gated_update_bwd")) return rc;
  reduce_partials_kernel<<<grid_for(P, 64), kBlock, 0, s>>>(workspace, dparams, nblk, P);
  return check_launch("reduce_partials");
}

int launch_adam_clipnorm(const void* table, const void* sizes, int n_vars, int64_t step, float lr, float b1, float b2,
                         float eps, float clipnorm, hipStream_t s) {
  const float corr1 = 1.0f - powf(b1, (float)step), corr2 = 1.0f - powf(b2, (float)step);
  adam_clipnorm_kernel<<<n_vars, 1024, 0, s>>>(static_cast<const unsigned long long*>(table),
                                               static_cast<const long long*>(sizes), lr, b1, b2, eps, clipnorm,
                                               corr1, corr2);
  return check_launch("adam_clipnorm");
}

}  // namespace impnn
