// Backward kernels of the layer-at-a-time path and the optimizer step: what Keras
// autodiff + Adam(1e-3, clipnorm=1.0) do for the reference's model.fit (train_viscosity.py:227-230,
// 328-338; train_melting_point.py:205-208).  One kernel per reference layer, the adjoint of the forward
// in layer_kernels.hip with the same masks (models/layers.py:114-115, :70 tgt > 0) and the same
// "out-of-range index == padding" rule.  Written for any D (VALU, f32); the parameter-gradient sums run
// over per-workgroup partial buffers that a second kernel adds in a fixed order, except where noted.
#include <cstdlib>

#include "common.h"

namespace impnn {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t ldv4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ void stv4(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ f32x4_t mfma_f32(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// a1/a2 backward: dtable[ids[r], :] += dout[r, :]   (float atomics: rows of one id meet in any order).
// Vocabularies are small (~10^2 rows), so thousands of rows collide on the same table row: when the table
// fits LDS every workgroup first sums its rows there (LDS atomics) and touches HBM once per entry.
// ---------------------------------------------------------------------------------------
__global__ void embed_gather_bwd_kernel(const int32_t* __restrict__ ids, const float* __restrict__ dout,
                                        float* __restrict__ dtable, int64_t rows, int vocab, int dim, int use_lds) {
  extern __shared__ __align__(16) float smem[];
  const int64_t total = rows * dim;
  if (use_lds) {
    const int tsize = vocab * dim;
    for (int t = threadIdx.x; t < tsize; t += blockDim.x) smem[t] = 0.f;
    __syncthreads();
    // contiguous slice of rows per workgroup
    const int64_t per = (total + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < total ? lo + per : total;
    constexpr int kU = 8;  // independent (id, value) loads in flight per thread
    for (int64_t t0 = lo + threadIdx.x; t0 < hi; t0 += (int64_t)kU * blockDim.x) {
      float v[kU];
      int at[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t t = t0 + (int64_t)u * blockDim.x;
        at[u] = -1;
        v[u] = 0.f;
        if (t < hi) {
          const int64_t r = t / dim;
          const int id = ids[r];
          v[u] = dout[t];
          if ((unsigned)id < (unsigned)vocab) at[u] = id * dim + (int)(t - r * dim);
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (at[u] >= 0 && v[u] != 0.f) atomicAdd(&smem[at[u]], v[u]);  // (padding atoms: half of the rows, all zeros)
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tsize; t += blockDim.x) {
      const float v = smem[t];
      if (v != 0.f) atomicAdd(&dtable[t], v);
    }
    return;
  }
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
// The valid edges of the WHOLE batch are counting-sorted by bond type (histogram, prefix, scatter - three
// small launches), and every workgroup of the main kernel takes one segment of <= kSeg edges of one type:
// A[type] and the segment's dm / h rows are staged in LDS, dA of the segment is summed in registers ((i,j)
// entries dealt over the threads) and leaves with one atomicAdd per entry, dh goes out with float atomics
// (several edges share a source row).  Parallelism is edges/kSeg workgroups at any batch size (a batch of
// 32 molecules still gives ~100).  dh and dA must be zeroed by the caller.
// workspace (int32): cnt Vb+1 | start Vb+1 | cursor Vb | segbase Vb+1 | order B*E
// ---------------------------------------------------------------------------------------
constexpr int kSeg = 64;
constexpr int kMaxTypes = 4096;

__device__ __forceinline__ int edge_type_or_neg(const int32_t* conn, const int32_t* bond_ids, int64_t be, int N, int Vb) {
  const int src = conn[be * 2], tgt = conn[be * 2 + 1], ty = bond_ids[be];
  return (src > 0 && tgt > 0 && src < N && tgt < N && (unsigned)ty < (unsigned)Vb) ? ty : -1;
}

__global__ void zero_floats_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) p[t] = 0.f;
}

__global__ void zero_ints_kernel(int32_t* __restrict__ p, int n) {
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) p[t] = 0;
}

// Histogram of the valid edges by type.  Each workgroup counts a contiguous slice in LDS first and touches the
// global counters once per type it saw: with ~10^2 types and ~10^5 edges, per-edge global atomics serialise.
__global__ __launch_bounds__(kBlock) void edge_type_hist_kernel(const int32_t* __restrict__ conn,
                                                                const int32_t* __restrict__ bond_ids,
                                                                int32_t* __restrict__ cnt, int64_t BE, int N, int Vb) {
  __shared__ int32_t lh[kMaxTypes];
  for (int t = threadIdx.x; t < Vb; t += kBlock) lh[t] = 0;
  __syncthreads();
  const int64_t per = (BE + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < BE ? lo + per : BE;
  for (int64_t be = lo + threadIdx.x; be < hi; be += kBlock) {
    const int ty = edge_type_or_neg(conn, bond_ids, be, N, Vb);
    if (ty >= 0) atomicAdd(&lh[ty], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < Vb; t += kBlock)
    if (lh[t]) atomicAdd(&cnt[t], lh[t]);
}

// one workgroup: start[t] = exclusive prefix of cnt, cursor = start, segbase[t] = exclusive prefix of
// ceil(cnt[t] / kSeg); start[Vb] = valid edges, segbase[Vb] = segments
__global__ void edge_type_prefix_kernel(const int32_t* __restrict__ cnt, int32_t* __restrict__ start,
                                        int32_t* __restrict__ cursor, int32_t* __restrict__ segbase, int Vb) {
  if (threadIdx.x == 0) {
    int run = 0, segs = 0;
    for (int t = 0; t < Vb; ++t) {
      const int c = cnt[t];
      start[t] = run;
      cursor[t] = run;
      segbase[t] = segs;
      run += c;
      segs += (c + kSeg - 1) / kSeg;
    }
    start[Vb] = run;
    segbase[Vb] = segs;
  }
}

// Scatter of the valid edges into their type's run: a workgroup counts its slice in LDS, reserves one range per type
// with a single global atomic, and places its edges inside the reserved ranges with LDS atomics.
__global__ __launch_bounds__(kBlock) void edge_type_scatter_kernel(const int32_t* __restrict__ conn,
                                                                   const int32_t* __restrict__ bond_ids,
                                                                   int32_t* __restrict__ cursor,
                                                                   int32_t* __restrict__ order, int64_t BE, int N,
                                                                   int Vb) {
  __shared__ int32_t lh[kMaxTypes];  // count, then the next free position of the reserved range
  for (int t = threadIdx.x; t < Vb; t += kBlock) lh[t] = 0;
  __syncthreads();
  const int64_t per = (BE + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < BE ? lo + per : BE;
  for (int64_t be = lo + threadIdx.x; be < hi; be += kBlock) {
    const int ty = edge_type_or_neg(conn, bond_ids, be, N, Vb);
    if (ty >= 0) atomicAdd(&lh[ty], 1);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < Vb; t += kBlock) {
    const int c = lh[t];
    lh[t] = c ? atomicAdd(&cursor[t], c) : 0;
  }
  __syncthreads();
  for (int64_t be = lo + threadIdx.x; be < hi; be += kBlock) {
    const int ty = edge_type_or_neg(conn, bond_ids, be, N, Vb);
    if (ty >= 0) order[atomicAdd(&lh[ty], 1)] = (int32_t)be;
  }
}

// The whole sort in ONE workgroup for small batches (the reference trains with 32 pairs: 8 K edge slots): every
// thread keeps the types of its <= 16 slots in registers between the histogram and the scatter, the scan over the
// types runs one type per thread.  Replaces four launches (zero, hist, prefix, scatter).
constexpr int kSortSmallPer = 16;
__global__ __launch_bounds__(1024) void edge_type_sort_small_kernel(const int32_t* __restrict__ conn,
                                                                    const int32_t* __restrict__ bond_ids,
                                                                    int32_t* __restrict__ cnt, int32_t* __restrict__ start,
                                                                    int32_t* __restrict__ cursor,
                                                                    int32_t* __restrict__ segbase,
                                                                    int32_t* __restrict__ order, int BE, int N, int Vb) {
  __shared__ int32_t lh[1024];  // counts, then the next free position of every type's run
  __shared__ int32_t wc[16], wsg[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  lh[tid] = 0;
  __syncthreads();
  int ty[kSortSmallPer];
#pragma unroll
  for (int u = 0; u < kSortSmallPer; ++u) {
    const int be = tid + u * 1024;
    ty[u] = be < BE ? edge_type_or_neg(conn, bond_ids, be, N, Vb) : -1;
  }
#pragma unroll
  for (int u = 0; u < kSortSmallPer; ++u)
    if (ty[u] >= 0) atomicAdd(&lh[ty[u]], 1);
  __syncthreads();
  const int c = tid < Vb ? lh[tid] : 0;
  const int sg = (c + kSeg - 1) / kSeg;
  int ic = c, is = sg;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int uc = __shfl_up(ic, o), us = __shfl_up(is, o);
    if (lane >= o) {
      ic += uc;
      is += us;
    }
  }
  if (lane == 63) {
    wc[wave] = ic;
    wsg[wave] = is;
  }
  __syncthreads();
  int oc = 0, os = 0;
  for (int w = 0; w < wave; ++w) {
    oc += wc[w];
    os += wsg[w];
  }
  ic += oc;
  is += os;
  if (tid < Vb) {
    cnt[tid] = c;
    start[tid] = ic - c;
    cursor[tid] = ic - c;
    segbase[tid] = is - sg;
    lh[tid] = ic - c;
  }
  if (tid == 1023) {  // threads past Vb carry zeros: the last inclusive value is the total
    start[Vb] = ic;
    segbase[Vb] = is;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kSortSmallPer; ++u)
    if (ty[u] >= 0) order[atomicAdd(&lh[ty[u]], 1)] = tid + u * 1024;
}

template <int ACC>  // ACC = ceil(D*D / blockDim.x) accumulators per thread
__global__ __launch_bounds__(1024) void bmm_message_typed_bwd_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ conn, const float* __restrict__ A,
    const float* __restrict__ dm, float* __restrict__ dh, float* __restrict__ dA, const int32_t* __restrict__ start,
    const int32_t* __restrict__ segbase, const int32_t* __restrict__ order, int N, int E, int D, int Vb,
    int from_agg, float* __restrict__ du) {  // from_agg: dm is the gradient of Reduce's output (B,N,D) and dm of edge e is its row tgt(e)
  // du (optional): per-edge vectors to their edge slot's row instead of atomics on dh (see the matrix-core kernel below)
  extern __shared__ __align__(16) float smem[];
  __shared__ int64_t srcrow[kSeg];
  __shared__ int64_t slot[kSeg];
  const int seg = blockIdx.x;
  if (seg >= segbase[Vb]) return;  // the grid is an upper bound on the number of segments
  int lo = 0, hi = Vb - 1;          // type of this segment: largest t with segbase[t] <= seg
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segbase[mid] <= seg) lo = mid; else hi = mid - 1;
  }
  const int ty = lo;
  const int p0 = start[ty] + (seg - segbase[ty]) * kSeg;
  const int n = min(kSeg, start[ty + 1] - p0);
  if (n <= 0) return;
  const int tid = threadIdx.x;
  const int DD = D * D;
  float* As = smem;            // D*D
  float* gm = As + DD;         // kSeg x D: dm rows of the segment's edges
  float* xm = gm + kSeg * D;   // kSeg x D: their source rows of h
  for (int t = tid; t < DD; t += (int)blockDim.x) As[t] = A[(int64_t)ty * DD + t];
  for (int t = tid; t < n * D; t += (int)blockDim.x) {
    const int e = t / D, c = t - e * D;
    const int64_t be = order[p0 + e];
    const int64_t row = (be / E) * N + conn[be * 2];
    const int64_t grow = from_agg ? (be / E) * N + conn[be * 2 + 1] : be;
    gm[e * D + c] = dm[grow * D + c];
    xm[e * D + c] = h[row * D + c];
    if (c == 0) {
      srcrow[e] = row;
      slot[e] = be;
    }
  }
  __syncthreads();
  const int lanes = (int)blockDim.x / D > 0 ? (int)blockDim.x / D : 1;
  if (tid < lanes * D) {  // dh: thread (edge lane, column j)
    const int j = tid % D, el = tid / D;
    for (int e = el; e < n; e += lanes) {
      float u = 0.f;
      for (int i = 0; i < D; ++i) u = fmaf(gm[e * D + i], As[i * D + j], u);
      if (du) du[slot[e] * D + j] = u;
      else atomicAdd(&dh[srcrow[e] * D + j], u);
    }
  }
#pragma unroll
  for (int a = 0; a < ACC; ++a) {  // dA of this segment: entry q = (i, j)
    const int q = tid + a * (int)blockDim.x;
    if (q < DD) {
      const int i = q / D, j = q - i * D;
      float v = 0.f;
      for (int e = 0; e < n; ++e) v = fmaf(gm[e * D + i], xm[e * D + j], v);
      atomicAdd(&dA[(int64_t)ty * DD + q], v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same adjoint on the matrix cores for wide states (D = 64, 128; exact f32 products).  Per 64-edge segment two
// GEMMs: dh rows = G A_t (features on M from the TRANSPOSED type matrix in LDS, edges on N) and dA_t += G^T X (both
// operands straight from the row-major edge tiles: one 4-byte LDS read per operand and k step, a 2x2 block of output
// tiles per wave), G = dm rows (or dagg rows at the edges' targets), X = h rows at their sources.  A workgroup walks a
// contiguous range of segments: the type's transposed matrix stays in LDS and its dA accumulators (16 registers per
// thread) in registers until the type changes, when they are added to dA with float atomics (dh likewise, as in the
// VALU kernel above: several edges share a source row).  Edge indices run three segments ahead of the MFMAs (sorted
// position -> edge slot -> its rows -> the two 512-byte rows), one stage per iteration, so that no request waits on a
// load issued in the same iteration; segments past the workgroup's range are clamped to its last one.
// The VALU kernel took 1.7 ms per call at 4096 molecules x D = 128 (1.3 % of the f32 MFMA peak for 10.7 GFLOP).
// ---------------------------------------------------------------------------------------
constexpr int kBwdMfmaMaxTypes = 1024;

template <int NT>
__global__ __launch_bounds__(1024) void bmm_message_typed_bwd_mfma_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ conn, const float* __restrict__ A,
    const float* __restrict__ dm, float* __restrict__ dh, float* __restrict__ dA, const int32_t* __restrict__ start,
    const int32_t* __restrict__ segbase, const int32_t* __restrict__ order, int N, int E, int Vb, int from_agg,
    int owner_mode, float* __restrict__ du) {
  // du (optional, (B,E,D) with zero rows at masked edges): the per-edge vectors A_t^T g_e go to their edge slot's row with
  // plain 16-byte stores and a slot-order pass (reduce_scatter_kernel keyed by the source index) adds them into dh -
  // instead of float atomics on dh from here: 22 M of them at batch 4096 were 260 of the kernel's 349 us.
  // owner_mode (small batches): workgroup t takes ALL segments of bond type t and is the only one that touches dA_t,
  // which it updates with plain loads / adds / stores - flushing 64 KB of accumulators with float atomics after a
  // single segment costs ~50 us per workgroup (one 256-byte atomic wave-instruction per ~50 ns and CU).
  constexpr int D = 16 * NT, LD = D + 4, QD = D / 4, NLW = NT / 4, TI = NT / 4;
  constexpr int kX = kSeg * QD / 1024;  // 16-byte pieces of an edge tile per thread
  constexpr int kB = D * QD / 1024;     // ... of the matrix
  static_assert(kX >= 1 && kB >= 1 && NLW >= 1, "tile shape");
  extern __shared__ __align__(16) float smem[];
  float* AmT = smem;                // D x LD : AmT[j][i] = A_t[i][j]
  float* G = AmT + D * LD;          // kSeg x LD
  float* X = G + kSeg * LD;         // kSeg x LD
  int32_t* hrow_s = reinterpret_cast<int32_t*>(X + kSeg * LD);  // kSeg source rows (b * N + src)
  int32_t* sb_s = hrow_s + kSeg;    // segbase[0 .. Vb]
  int32_t* st_s = sb_s + Vb + 1;    // start[0 .. Vb]
  int32_t* be_s = st_s + Vb + 1;    // kSeg edge slots (b * E + e; -1 past the segment's edges) - with du
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int nseg = segbase[Vb];
  const int per = (nseg + (int)gridDim.x - 1) / (int)gridDim.x;
  int s0 = blockIdx.x * per, s1 = s0 + per < nseg ? s0 + per : nseg;
  if (owner_mode) {
    if ((int)blockIdx.x >= Vb) return;
    s0 = segbase[blockIdx.x];
    s1 = segbase[blockIdx.x + 1];
  }
  if (s0 >= s1) return;
  for (int t = tid; t <= Vb; t += 1024) {
    sb_s[t] = segbase[t];
    st_s[t] = start[t];
  }
  __syncthreads();
  struct Seg {
    int ty, p0, n;
  };
  auto seg_of = [&](int seg, int ty) {  // ty: a type at or before the segment's
    while (sb_s[ty + 1] <= seg) ++ty;
    Seg d;
    d.ty = ty;
    d.p0 = st_s[ty] + (seg - sb_s[ty]) * kSeg;
    d.n = min(kSeg, st_s[ty + 1] - d.p0);
    return d;
  };
  int ty0;
  {
    int lo = 0, hi = Vb - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sb_s[mid] <= s0) lo = mid; else hi = mid - 1;
    }
    ty0 = lo;
  }
  // pipeline stages, per thread and piece i < kX (edge e_i = (tid + 1024 i) / QD of the segment):
  //   stage C (3 ahead): be  = order[p0 + min(e, n - 1)]
  //   stage B (2 ahead): src, tgt = conn[be]                      -> rows
  //   stage A (1 ahead): the two 16-byte pieces of dm / h          -> registers -> LDS after the MFMAs
  int be_c[kX], be_b[kX], ok_b[kX];
  int hrow_a[kX], grow_a[kX], ok_a[kX];
  int64_t hrow_x[kX];
  f32x4_t gr[kX], xr[kX];
  int hrow_n[kX], ok_n[kX];  // of the pieces held in gr / xr
  int bes_a[kX], bes_n[kX];  // their edge slots (with du)
  auto stage_c = [&](const Seg& d, int* be, int* ok) {
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      const int e = (tid + 1024 * i) / QD;
      ok[i] = e < d.n;
      // (slots past the segment's edges borrow the rows of its real edges in turn: their zero sums are then spread
      //  over the segment's source rows - all of them on the LAST edge's row made 46 of 64 slots contend for one
      //  row at the reference's batch of 32: 99 vs 15 us per call)
      be[i] = order[d.p0 + (e < d.n ? e : e % d.n)];
    }
  };
  auto stage_b = [&](const int* be, int* hrow, int* grow, int* bes) {
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      const int b = be[i] / E;
      const int2 st = *reinterpret_cast<const int2*>(conn + (int64_t)be[i] * 2);
      hrow[i] = b * N + st.x;
      grow[i] = from_agg ? b * N + st.y : be[i];
      bes[i] = be[i];
    }
  };
  auto stage_a = [&](const int* hrow, const int* grow) {
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      const int c4 = (tid + 1024 * i) % QD;
      gr[i] = ldv4(dm + (int64_t)grow[i] * D + 4 * c4);
      xr[i] = ldv4(h + (int64_t)hrow[i] * D + 4 * c4);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      const int idx = tid + 1024 * i, e = idx / QD, c4 = idx - e * QD;
      const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
      stv4(G + e * LD + 4 * c4, ok_n[i] ? gr[i] : zero);  // rows past the segment's edges are zero: they add nothing to dA
      stv4(X + e * LD + 4 * c4, ok_n[i] ? xr[i] : zero);
      if (c4 == 0) {
        hrow_s[e] = hrow_n[i];
        be_s[e] = ok_n[i] ? bes_n[i] : -1;
      }
    }
  };
  auto load_matrix = [&](int ty) {  // transposing copy: lanes run along i (conflict-free LDS stores)
#ifdef IMPNN_DIAG_BWD_NOMATRIX
    return;
#endif
    const float* At = A + (int64_t)ty * D * D;
#pragma unroll
    for (int i2 = 0; i2 < kB; ++i2) {
      const int idx = tid + 1024 * i2, i = idx % D, c4 = idx / D;
      const f32x4_t v = ldv4(At + (int64_t)i * D + 4 * c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) AmT[(4 * c4 + r) * LD + i] = v[r];
    }
  };
  const int last = s1 - 1;
  Seg d0 = seg_of(s0, ty0);
  Seg d1 = seg_of(min(s0 + 1, last), d0.ty), d2 = seg_of(min(s0 + 2, last), d1.ty), d3 = seg_of(min(s0 + 3, last), d2.ty);
  // prologue: bring segment s0 into LDS, s0 + 1 to stage A, s0 + 2 to stage B, s0 + 3 to stage C
  {
    int be0[kX], ok0[kX], hr0[kX], gr0[kX];
    stage_c(d0, be0, ok0);
    stage_b(be0, hr0, gr0, bes_n);
    stage_a(hr0, gr0);
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      hrow_n[i] = hr0[i];
      ok_n[i] = ok0[i];
    }
    load_matrix(d0.ty);
    park();
    stage_c(d1, be0, ok0);
    stage_b(be0, hrow_a, grow_a, bes_a);
#pragma unroll
    for (int i = 0; i < kX; ++i) ok_a[i] = ok0[i];
    stage_c(d2, be_b, ok_b);
    stage_c(d3, be_c, ok_n);  // (ok of stage C travels with it below)
  }
  int ok_c[kX];
#pragma unroll
  for (int i = 0; i < kX; ++i) ok_c[i] = ok_n[i];
#pragma unroll
  for (int i = 0; i < kX; ++i) ok_n[i] = 1;  // placeholder until the first stage-A request below
  __syncthreads();
  const int et = wave & 3, fg = wave >> 2;          // GEMM 1: edge tile, feature group
  const int wi = wave & 3, wj = wave >> 2;          // GEMM 2: block of i tiles, block of j tiles
  f32x4_t acc2[TI][TI];
#pragma unroll
  for (int x = 0; x < TI; ++x)
#pragma unroll
    for (int y = 0; y < TI; ++y) acc2[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto flush_dA = [&](int ty) {
#ifdef IMPNN_DIAG_BWD_NOFLUSH
    return;
#endif
    float* dst = dA + (int64_t)ty * D * D;
#pragma unroll
    for (int x = 0; x < TI; ++x)
#pragma unroll
      for (int y = 0; y < TI; ++y) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float* pd = dst + (16 * (wi * TI + x) + 4 * q + g) * D + 16 * (wj * TI + y) + a;
          if (owner_mode) *pd += acc2[x][y][g];
          else atomicAdd(pd, acc2[x][y][g]);
        }
        acc2[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      }
  };
  for (int seg = s0; seg < s1; ++seg) {
    // requests for the segments ahead (each consumes what the previous iteration requested)
    int hrow_t[kX], ok_t[kX], bes_t[kX];
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      hrow_t[i] = hrow_a[i];
      ok_t[i] = ok_a[i];
      bes_t[i] = bes_a[i];
    }
    stage_a(hrow_a, grow_a);                 // rows of seg + 1
    stage_b(be_b, hrow_a, grow_a, bes_a);    // row indices of seg + 2
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      ok_a[i] = ok_b[i];
      be_b[i] = be_c[i];
      ok_b[i] = ok_c[i];
    }
    const Seg d4 = seg_of(min(seg + 4, last), d3.ty);
    stage_c(d4, be_c, ok_c);                 // edge slots of seg + 4
    __builtin_amdgcn_sched_barrier(0);
    // ---- GEMM 1: dh rows of this segment's edges
    {
      f32x4_t acc1[NLW];
#pragma unroll
      for (int TL = 0; TL < NLW; ++TL) acc1[TL] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const float* grow_p = G + (16 * et + a) * LD + 4 * q;
      const float* arow_p = AmT + (16 * (fg * NLW) + a) * LD + 4 * q;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const f32x4_t gv = ldv4(grow_p + 16 * u);
        f32x4_t av[NLW];
#pragma unroll
        for (int TL = 0; TL < NLW; ++TL) av[TL] = ldv4(arow_p + 16 * TL * LD + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int TL = 0; TL < NLW; ++TL) acc1[TL] = mfma_f32(av[TL][r], gv[r], acc1[TL]);
      }
      // (unconditional: rows past the segment's edges are zero in G, so their sums are exact zeros added to rows of
      //  the segment's real edges - a conditional atomic would keep the compiler from counting outstanding
      //  memory operations, and the LDS stores below would wait for every atomic of the tile)
#ifndef IMPNN_DIAG_BWD_NODH
      if (du) {  // (workgroup-uniform)
        const int bes = be_s[16 * et + a];
        if (bes >= 0) {
          float* dst = du + (int64_t)bes * D + 16 * (fg * NLW) + 4 * q;
#pragma unroll
          for (int TL = 0; TL < NLW; ++TL) stv4(dst + 16 * TL, acc1[TL]);
        }
      } else {
        float* dst = dh + (int64_t)hrow_s[16 * et + a] * D + 16 * (fg * NLW) + 4 * q;
#pragma unroll
        for (int TL = 0; TL < NLW; ++TL)
#pragma unroll
          for (int g = 0; g < 4; ++g) atomicAdd(dst + 16 * TL + g, acc1[TL][g]);
      }
#endif
    }
    // ---- GEMM 2: dA_t += G^T X over the segment's edges (k = edge)
#pragma unroll 4
    for (int sx = 0; sx < kSeg / 4; ++sx) {
      float gi[TI], xj[TI];
#pragma unroll
      for (int x = 0; x < TI; ++x) gi[x] = G[(4 * sx + q) * LD + 16 * (wi * TI + x) + a];
#pragma unroll
      for (int y = 0; y < TI; ++y) xj[y] = X[(4 * sx + q) * LD + 16 * (wj * TI + y) + a];
#pragma unroll
      for (int x = 0; x < TI; ++x)
#pragma unroll
        for (int y = 0; y < TI; ++y) acc2[x][y] = mfma_f32(gi[x], xj[y], acc2[x][y]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (seg + 1 >= s1) break;
    const bool new_type = d1.ty != d0.ty;  // (workgroup-uniform)
    if (new_type) flush_dA(d0.ty);
    __syncthreads();                       // every wave is done with G, X, hrow_s (and AmT)
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      hrow_n[i] = hrow_t[i];
      ok_n[i] = ok_t[i];
      bes_n[i] = bes_t[i];
    }
    park();
    if (new_type) load_matrix(d1.ty);
    __syncthreads();
    d0 = d1;
    d1 = d2;
    d2 = d3;
    d3 = d4;
  }
  flush_dA(d0.ty);
}

// ---------------------------------------------------------------------------------------
// a4 forward over the same type-sorted segments, for any D <= 128 (the D = 32 MFMA kernel of layer_kernels.hip
// keeps its own in-workgroup sort): A[type] sits in LDS with row stride D+1, so the lanes (output feature i) read
// their rows without bank conflicts - the per-molecule kernel reads A[type][i][:] with a stride of D floats between
// lanes, 64 cache lines per load.  Thread (edge lane, i) computes 4 edges at a time from one pass over its row.
// Masked / out-of-range edges get zero rows from a separate pass (models/layers.py:114-115).
// ---------------------------------------------------------------------------------------
__global__ void zero_invalid_messages_kernel(const int32_t* __restrict__ conn, const int32_t* __restrict__ bond_ids,
                                             float* __restrict__ m, int64_t BE, int N, int D, int Vb) {
  const int64_t total = BE * D;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t be = t / D;
    if (edge_type_or_neg(conn, bond_ids, be, N, Vb) < 0) m[t] = 0.f;
  }
}

__global__ __launch_bounds__(kBlock) void bmm_message_typed_seg_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ conn, const float* __restrict__ A,
    float* __restrict__ m_out, const int32_t* __restrict__ start, const int32_t* __restrict__ segbase,
    const int32_t* __restrict__ order, int N, int E, int D, int Vb) {
  extern __shared__ __align__(16) float smem[];
  __shared__ int64_t outrow[kSeg];
  const int seg = blockIdx.x;
  if (seg >= segbase[Vb]) return;
  int lo = 0, hi = Vb - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segbase[mid] <= seg) lo = mid; else hi = mid - 1;
  }
  const int ty = lo;
  const int p0 = start[ty] + (seg - segbase[ty]) * kSeg;
  const int n = min(kSeg, start[ty + 1] - p0);
  if (n <= 0) return;
  const int tid = threadIdx.x;
  const int LD = D + 1;
  float* As = smem;              // D x (D+1)
  float* xm = As + D * LD;       // kSeg x D, rows beyond n are zero
  for (int t = tid; t < D * D; t += kBlock) As[(t / D) * LD + (t % D)] = A[(int64_t)ty * D * D + t];
  for (int t = tid; t < kSeg * D; t += kBlock) {
    const int e = t / D, c = t - e * D;
    float v = 0.f;
    if (e < n) {
      const int64_t be = order[p0 + e];
      v = h[((be / E) * N + conn[be * 2]) * D + c];
      if (c == 0) outrow[e] = be;
    }
    xm[t] = v;
  }
  __syncthreads();
  const int lanes = kBlock / D > 0 ? kBlock / D : 1;
  if (tid < lanes * D) {
    const int i = tid % D, el = tid / D;
    const float* arow = As + i * LD;
    for (int e0 = 4 * el; e0 < n; e0 += 4 * lanes) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      const float* x = xm + e0 * D;
      for (int j = 0; j < D; ++j) {
        const float w = arow[j];
        a0 = fmaf(w, x[j], a0);
        a1 = fmaf(w, x[D + j], a1);
        a2 = fmaf(w, x[2 * D + j], a2);
        a3 = fmaf(w, x[3 * D + j], a3);
      }
      m_out[outrow[e0] * D + i] = a0;
      if (e0 + 1 < n) m_out[outrow[e0 + 1] * D + i] = a1;
      if (e0 + 2 < n) m_out[outrow[e0 + 2] * D + i] = a2;
      if (e0 + 3 < n) m_out[outrow[e0 + 3] * D + i] = a3;
    }
  }
}

// The same on the matrix cores for D a multiple of 16 (exact f32 products): per segment the GEMM
// m^T (D x 64) = A[type] (D x D) * x^T (D x 64) in 16x16 output tiles, K index ordered as 16u + 4q + r so that one
// 16-byte LDS read per lane feeds four MFMA steps of both operands.  Wave w owns the output tiles w, w+4, ...
__global__ __launch_bounds__(1024) void bmm_message_typed_seg_mfma_kernel(
    const float* __restrict__ h, const int32_t* __restrict__ conn, const float* __restrict__ A,
    float* __restrict__ m_out, const int32_t* __restrict__ start, const int32_t* __restrict__ segbase,
    const int32_t* __restrict__ order, int N, int E, int D, int Vb, int segs_per_wg) {
  extern __shared__ __align__(16) float smem[];
  __shared__ int64_t outrow[kSeg];
  const int nseg = segbase[Vb];
  const int tid = threadIdx.x;
  const int LD = D + 4;            // 16-byte aligned rows, bank-staggered
  float* As = smem;                // D x LD
  float* xm = As + D * LD;         // kSeg x LD, rows beyond n are zero
  const int D4 = D >> 2;
  const int lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  // A workgroup walks `segs_per_wg` consecutive segments: segments of one type are numbered consecutively, so the
  // type's matrix (D*D*4 bytes - 64 KB at D = 128: the segment's dominant cost, not its MFMAs) stays in LDS until the
  // type changes instead of being copied once per 64 edges.
  int held = -1;
  for (int seg = blockIdx.x * segs_per_wg; seg < (blockIdx.x + 1) * segs_per_wg && seg < nseg; ++seg) {
    int lo = 0, hi = Vb - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (segbase[mid] <= seg) lo = mid; else hi = mid - 1;
    }
    const int ty = lo;
    const int p0 = start[ty] + (seg - segbase[ty]) * kSeg;
    const int n = min(kSeg, start[ty + 1] - p0);
    if (n <= 0) continue;  // (workgroup-uniform)
    __syncthreads();       // the previous segment's MFMAs are done with xm / outrow (and As, if the type changes)
    // 16-byte loads throughout (D is a multiple of 16, rows of A / h / the LDS tiles are 16-byte aligned)
    if (ty != held) {
      held = ty;
      const float* Aty = A + (int64_t)ty * D * D;
      for (int t = tid; t < D * D4; t += (int)blockDim.x) {
        const int r = t / D4, c4 = t - r * D4;
        stv4(As + r * LD + 4 * c4, ldv4(Aty + (int64_t)r * D + 4 * c4));
      }
    }
    if (tid < kSeg) outrow[tid] = tid < n ? (int64_t)order[p0 + tid] : 0;
    __syncthreads();
    for (int t = tid; t < kSeg * D4; t += (int)blockDim.x) {
      const int e = t / D4, c4 = t - e * D4;
      f32x4_t v = {0.f, 0.f, 0.f, 0.f};
      if (e < n) {
        const int64_t be = outrow[e];
        v = ldv4(h + ((be / E) * N + conn[be * 2]) * D + 4 * c4);
      }
      stv4(xm + e * LD + 4 * c4, v);
    }
    __syncthreads();
    const int mt = D >> 4, et = (n + 15) >> 4;          // output tiles: features x edges
    for (int tile = wave; tile < mt * et; tile += (int)blockDim.x >> 6) {
      const int T = tile % mt, Et = tile / mt;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      const float* arow = As + (16 * T + a) * LD + 4 * q;
      const float* xrow = xm + (16 * Et + a) * LD + 4 * q;
      for (int u = 0; u < mt; ++u) {
        const f32x4_t av = ldv4(arow + 16 * u), xv = ldv4(xrow + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma_f32(av[r], xv[r], acc);
      }
      const int e = 16 * Et + a;                        // accumulator: feature 16T + 4q + reg of edge e
      if (e < n) stv4(m_out + outrow[e] * D + 16 * T + 4 * q, acc);
    }
  }
}

// ---------------------------------------------------------------------------------------
// schedule A backward: A[v] = sum_k Tb[v,k] W[k]  =>  dW[k] = sum_v Tb[v,k] dA[v];  dTb[v,k] = <dA[v], W[k]>
// ---------------------------------------------------------------------------------------
__global__ void bond_type_matrices_bwd_w_kernel(const float* __restrict__ tb, const float* __restrict__ dA,
                                                float* __restrict__ dW, int Vb, int K, int DD, int accumulate) {
  const int k = blockIdx.y;
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < DD; ij += gridDim.x * blockDim.x) {
    float acc = 0.f;
    int v = 0;
    for (; v + 16 <= Vb; v += 16) {  // 16 independent rows in flight; the adds stay in vocabulary order
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = dA[(int64_t)(v + u) * DD + ij];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(tb[(int64_t)(v + u) * K + k], x[u], acc);
    }
    for (; v < Vb; ++v) acc = fmaf(tb[(int64_t)v * K + k], dA[(int64_t)v * DD + ij], acc);
    dW[(int64_t)k * DD + ij] = accumulate ? dW[(int64_t)k * DD + ij] + acc : acc;
  }
}
__global__ void bond_type_matrices_bwd_t_kernel(const float* __restrict__ W, const float* __restrict__ dA,
                                                float* __restrict__ dtb, int Vb, int K, int DD, int accumulate) {
  // one wave per (v,k)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= Vb * K) return;
  const int v = wave / K, k = wave - v * K;
  float acc = 0.f;
  const float* da = dA + (int64_t)v * DD;
  const float* w = W + (int64_t)k * DD;
  int ij = lane;
  for (; ij + 7 * 64 < DD; ij += 8 * 64) {  // 16 independent loads in flight per lane
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x[u] = da[ij + 64 * u];
      y[u] = w[ij + 64 * u];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(x[u], y[u], acc);
  }
  for (; ij < DD; ij += 64) acc = fmaf(da[ij], w[ij], acc);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) dtb[(int64_t)v * K + k] = accumulate ? dtb[(int64_t)v * K + k] + acc : acc;
}

// ---------------------------------------------------------------------------------------
// The same for ALL message layers of a model in one launch each (training at the reference's batch 32 is bound by
// the number of launches, and these depend on the weights only): problem p = (ion, step) has its own W_p, the bond
// embedding table is shared, so dTb sums over the problems.
// ---------------------------------------------------------------------------------------
constexpr int kBtmMax = 16;
struct BtmBatch {
  const float* W[kBtmMax];
  const float* dA[kBtmMax];
  float* out[kBtmMax];  // forward: A_p (Vb x DD); backward: dW_p (K x DD)
  int n;
};
// kernel-argument tables are indexed with constants only (a dynamic index is a dependent kernarg load per use)
#define BTM_STAGE(bt)                                                       \
  __shared__ const float* sW[kBtmMax];                                      \
  __shared__ const float* sdA[kBtmMax];                                     \
  __shared__ float* sout[kBtmMax];                                          \
  _Pragma("unroll") for (int q = 0; q < kBtmMax; ++q) if ((int)threadIdx.x == q) {  \
    sW[q] = bt.W[q];                                                        \
    sdA[q] = bt.dA[q];                                                      \
    sout[q] = bt.out[q];                                                    \
  }                                                                         \
  __syncthreads();

__global__ void bond_type_matrices_multi_kernel(const float* __restrict__ tb, BtmBatch bt, int Vb, int K, int DD) {
  BTM_STAGE(bt)
  const int v = blockIdx.y;
  const float* W = sW[blockIdx.z];
  float* out = sout[blockIdx.z];
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < DD; ij += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(tb[(int64_t)v * K + k], W[(int64_t)k * DD + ij], acc);  // as the single kernel
    out[(int64_t)v * DD + ij] = acc;
  }
}
// bond_dim <= 8 (the reference's 8): the K values W[k][ij] of a thread's element stay in registers while it walks the
// bond types, so W is read once instead of once per type (at atom_dim 128: 6 MB instead of 436 MB of L2 reads over the
// 12 layers; 86 -> ~15 us).  Same fmaf chain per output element as the kernel above: identical bits.
constexpr int kBtmSmallK = 8;
constexpr int kBtmTbMax = 4096;  // bond_table floats staged in LDS (Vb * K)
__global__ __launch_bounds__(kBlock) void bond_type_matrices_multi_smallk_kernel(const float* __restrict__ tb, BtmBatch bt,
                                                                                 int Vb, int K, int DD) {
  BTM_STAGE(bt)
  __shared__ float tb_s[kBtmTbMax];
  for (int t = threadIdx.x; t < Vb * K; t += kBlock) tb_s[t] = tb[t];
  __syncthreads();
  const float* W = sW[blockIdx.y];
  float* out = sout[blockIdx.y];
  const int ij = blockIdx.x * kBlock + threadIdx.x;
  if (ij >= DD) return;
  float w[kBtmSmallK];
#pragma unroll
  for (int k = 0; k < kBtmSmallK; ++k) w[k] = k < K ? W[(int64_t)k * DD + ij] : 0.f;
  for (int v = 0; v < Vb; ++v) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kBtmSmallK; ++k)
      if (k < K) acc = fmaf(tb_s[v * K + k], w[k], acc);
    out[(int64_t)v * DD + ij] = acc;
  }
}

__global__ void bond_type_matrices_multi_bwd_w_kernel(const float* __restrict__ tb, BtmBatch bt, int Vb, int K, int DD,
                                                      int accumulate) {
  BTM_STAGE(bt)
  const int k = blockIdx.y;
  const float* dA = sdA[blockIdx.z];
  float* dW = sout[blockIdx.z];
  for (int ij = blockIdx.x * blockDim.x + threadIdx.x; ij < DD; ij += gridDim.x * blockDim.x) {
    float acc = 0.f;
    int v = 0;
    for (; v + 16 <= Vb; v += 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) x[u] = dA[(int64_t)(v + u) * DD + ij];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(tb[(int64_t)(v + u) * K + k], x[u], acc);
    }
    for (; v < Vb; ++v) acc = fmaf(tb[(int64_t)v * K + k], dA[(int64_t)v * DD + ij], acc);
    dW[(int64_t)k * DD + ij] = accumulate ? dW[(int64_t)k * DD + ij] + acc : acc;
  }
}
__global__ void bond_type_matrices_multi_bwd_t_kernel(BtmBatch bt, float* __restrict__ dtb, int Vb, int K, int DD,
                                                      int accumulate) {
  BTM_STAGE(bt)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= Vb * K) return;
  const int v = wave / K, k = wave - v * K;
  float acc = 0.f;
  for (int p = 0; p < bt.n; ++p) {  // problems in order, then the lanes: a fixed summation order
    const float* da = sdA[p] + (int64_t)v * DD;
    const float* w = sW[p] + (int64_t)k * DD;
    int ij = lane;
    for (; ij + 7 * 64 < DD; ij += 8 * 64) {
      float x[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x[u] = da[ij + 64 * u];
        y[u] = w[ij + 64 * u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(x[u], y[u], acc);
    }
    for (; ij < DD; ij += 64) acc = fmaf(da[ij], w[ij], acc);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) dtb[(int64_t)v * K + k] = accumulate ? dtb[(int64_t)v * K + k] + acc : acc;
}


// ---------------------------------------------------------------------------------------
// dtb[v,k] = sum_p sum_ij dA_p[v,ij] W_p[k,ij] on the matrix cores (exact f32 products): bond types on M (VT tiles of
// 16), k on N (K <= 16, padded with zeros), the contraction over (p, ij) cut into chunks of kBtC per WAVE - a lane loads
// 16 bytes of a dA row and of a W row per 16 contraction indices (the MFMA's k = the lane's quarter q, component r of
// the quad: any bijection serves as long as both operands use it).  The per-wave partials land in `part`
// ([wave][v][k]) and bond_type_matrices_t_sum_kernel adds them in wave order: bitwise reproducible.
// The one-wave-per-output kernel above walked 12 x 16 K products per lane with eight loads in flight: 116-170 us at
// atom_dim 128 - alone on the stream at the end of every backward pass, so all of it on the critical path of a step.
// ---------------------------------------------------------------------------------------
constexpr int kBtC = 256;
template <int VT>
__global__ __launch_bounds__(256) void bond_type_matrices_multi_bwd_t_mfma_kernel(BtmBatch bt, float* __restrict__ part,
                                                                                  int Vb, int K, int DD, int waves) {
  BTM_STAGE(bt)
  const int w = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63, a = lane & 15, q = lane >> 4;
  if (w >= waves) return;
  const int cpp = DD / kBtC, p = w / cpp, c0 = (w - p * cpp) * kBtC + 4 * q;
  const float* Wp = sW[p] + (int64_t)(a < K ? a : 0) * DD + c0;
  const float* dAp = sdA[p] + c0;
  const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
  f32x4_t acc[VT];
#pragma unroll
  for (int t = 0; t < VT; ++t) acc[t] = zero;
#pragma unroll 2
  for (int st = 0; st < kBtC / 16; ++st) {
    const f32x4_t wv = a < K ? ldv4(Wp + 16 * st) : zero;
    f32x4_t x[VT];
#pragma unroll
    for (int t = 0; t < VT; ++t) {
      const int v = 16 * t + a;
      x[t] = v < Vb ? ldv4(dAp + (int64_t)v * DD + 16 * st) : zero;
    }
#pragma unroll
    for (int t = 0; t < VT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t] = mfma_f32(x[t][r], wv[r], acc[t]);
  }
  float* mine = part + (int64_t)w * Vb * K;
#pragma unroll
  for (int t = 0; t < VT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int v = 16 * t + 4 * q + g;
      if (v < Vb && a < K) mine[v * K + a] = acc[t][g];
    }
}
__global__ void bond_type_matrices_t_sum_kernel(const float* __restrict__ part, float* __restrict__ dtb, int VK, int waves,
                                                int accumulate) {
  const int e = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (e >= VK) return;
  float acc = 0.f;
  for (int w = lane; w < waves; w += 64) acc += part[(int64_t)w * VK + e];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane == 0) dtb[e] = accumulate ? dtb[e] + acc : acc;
}

// ---------------------------------------------------------------------------------------
// a7 backward (models/layers.py:142-156).  Forward per row, c = [h|agg]:
//   z = sig(c Wz + bz); r = sig(c Wr + br); t = tanh([r*h|agg] Wh + bh); n = (1-z) h + z t;
//   x = (n - mean) * inv; out = gamma x + beta + h
// Three launches:
//   1. gated_update_bwd_kernel: tiles of R = kBlock/D rows, intermediates recomputed from (h, agg) ->
//      dh, dagg, the pre-activation gradients dpre = [dzp|drp|dtp] (rows x 3D), r*h (rows x D), and per
//      workgroup the column sums that give dbz, dbr, dbh, dgamma, dbeta (5D floats);
//   2. tn_gemm_splitk_kernel: the three kernel gradients dW_g = in_g^T dpre_g (in_z = in_r = [h|agg],
//      in_h = [r*h|agg]) as split-K GEMMs over row chunks (64x32 output tiles, 4x2 per thread);
//   3. gated_update_reduce_kernel: adds the per-workgroup / per-chunk partials in a fixed order into dparams
//      (Wz 2D*D | bz | Wr | br | Wh | bh | gamma | beta): parameter gradients are bitwise reproducible.
// ---------------------------------------------------------------------------------------
// WLDS: the three gate kernels staged in LDS (row stride D+1: both access directions conflict-free).
// BS: workgroup size = rows per tile x D; large D uses 1024 threads so that one pass over the (L2-resident)
// kernels serves 4x more rows.
template <bool WLDS, int BS>
__global__ __launch_bounds__(BS) void gated_update_bwd_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma, float eps,
    const float* __restrict__ dout, float* __restrict__ dh, float* __restrict__ dagg, float* __restrict__ dpre,
    float* __restrict__ rh_out, float* __restrict__ small, const float* __restrict__ WT, int64_t rows, int D, int R) {
  extern __shared__ __align__(16) float smem[];
  float* hs = smem;            // R*D each
  float* as = hs + R * D;
  float* zs = as + R * D;
  float* rs = zs + R * D;
  float* rhs = rs + R * D;     // r * h
  float* ts = rhs + R * D;     // tanh
  float* xs = ts + R * D;      // n, then x-hat, then dh so far
  float* g1 = xs + R * D;      // dx-hat, then dzp
  float* g2 = g1 + R * D;      // dx-hat * x-hat, then drp
  float* g3 = g2 + R * D;      // dtp
  float* st = g3 + R * D;      // 4*R: mean, inv, m1, m2
  const int LD = D + 1;
  float* wz_s = st + 4 * R;    // 2D*LD each (WLDS only)
  float* wr_s = wz_s + 2 * D * LD;
  float* wh_s = wr_s + 2 * D * LD;
  const int tid = threadIdx.x;
#define WZ(r_, c_) (WLDS ? wz_s[(r_) * LD + (c_)] : Wz[(int64_t)(r_) * D + (c_)])
#define WR(r_, c_) (WLDS ? wr_s[(r_) * LD + (c_)] : Wr[(int64_t)(r_) * D + (c_)])
#define WH(r_, c_) (WLDS ? wh_s[(r_) * LD + (c_)] : Wh[(int64_t)(r_) * D + (c_)])
  // element (r_, c_) for loops whose LANES walk r_: from LDS (padded stride) or from the transposed copies
  // WT = [Wz^T | Wr^T | Wh^T], each D x 2D, so that consecutive lanes read consecutive addresses
  const int D2 = 2 * D;
#define WZ_T(r_, c_) (WLDS ? wz_s[(r_) * LD + (c_)] : WT[(int64_t)(c_) * D2 + (r_)])
#define WR_T(r_, c_) (WLDS ? wr_s[(r_) * LD + (c_)] : WT[(int64_t)D * D2 + (int64_t)(c_) * D2 + (r_)])
#define WH_T(r_, c_) (WLDS ? wh_s[(r_) * LD + (c_)] : WT[(int64_t)2 * D * D2 + (int64_t)(c_) * D2 + (r_)])
  float s_bz = 0.f, s_br = 0.f, s_bh = 0.f, s_dg = 0.f, s_db = 0.f;  // this thread's (row slot, column) sums
  if (WLDS) {
    for (int t = threadIdx.x; t < 2 * D * D; t += BS) {
      const int rw = t / D, c = t - rw * D;
      wz_s[rw * LD + c] = Wz[t];
      wr_s[rw * LD + c] = Wr[t];
      wh_s[rw * LD + c] = Wh[t];
    }
  }
  const int64_t ntile = (rows + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    const int nr = (int)((rows - row0) < R ? (rows - row0) : R);
    __syncthreads();
    for (int t = tid; t < nr * D; t += BS) {
      hs[t] = h[row0 * D + t];
      as[t] = agg[row0 * D + t];
    }
    __syncthreads();
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D, i = t - r * D;
      float az = bz[i], ar = br[i];
      for (int j = 0; j < D; ++j) {
        const float x = hs[r * D + j];
        az = fmaf(x, WZ(j, i), az);
        ar = fmaf(x, WR(j, i), ar);
      }
      for (int j = 0; j < D; ++j) {
        const float x = as[r * D + j];
        az = fmaf(x, WZ(D + j, i), az);
        ar = fmaf(x, WR(D + j, i), ar);
      }
      const float z = sigmoid_exact(az), rr = sigmoid_exact(ar);
      zs[t] = z;
      rs[t] = rr;
      const float rhv = rr * hs[t];
      rhs[t] = rhv;
      rh_out[row0 * D + t] = rhv;
    }
    __syncthreads();
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D, i = t - r * D;
      float ah = bh[i];
      for (int j = 0; j < D; ++j) ah = fmaf(rhs[r * D + j], WH(j, i), ah);
      for (int j = 0; j < D; ++j) ah = fmaf(as[r * D + j], WH(D + j, i), ah);
      const float tt = tanhf(ah);
      ts[t] = tt;
      xs[t] = (1.0f - zs[t]) * hs[t] + zs[t] * tt;
    }
    __syncthreads();
    for (int r = tid; r < nr; r += BS) {
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
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D, i = t - r * D;
      const float xh = (xs[t] - st[4 * r]) * st[4 * r + 1];
      xs[t] = xh;
      const float dy = dout[row0 * D + t];
      const float dxh = dy * gamma[i];
      g1[t] = dxh;
      g2[t] = dxh * xh;
    }
    __syncthreads();
    for (int r = tid; r < nr; r += BS) {
      float m1 = 0.f, m2 = 0.f;
      for (int j = 0; j < D; ++j) {
        m1 += g1[r * D + j];
        m2 += g2[r * D + j];
      }
      st[4 * r + 2] = m1 / (float)D;
      st[4 * r + 3] = m2 / (float)D;
    }
    __syncthreads();
    // dn -> (dzp, dtp), first part of dh; column sums for dgamma / dbeta
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D;
      const float dy = dout[row0 * D + t];
      const float xh = xs[t];
      const float dn = st[4 * r + 1] * (g1[t] - st[4 * r + 2] - xh * st[4 * r + 3]);
      const float z = zs[t], tt = ts[t];
      const float dzp = dn * (tt - hs[t]) * z * (1.0f - z);
      const float dtp = dn * z * (1.0f - tt * tt);
      g1[t] = dzp;
      g3[t] = dtp;
      xs[t] = dy + dn * (1.0f - z);  // dh so far (x-hat is dead)
      if (t == tid) {                 // BS == R*D: one element per thread, fixed (slot, column)
        s_dg = fmaf(dy, xh, s_dg);
        s_db += dy;
        s_bz += dzp;
        s_bh += dtp;
      }
    }
    __syncthreads();
    // dc2 = dtp Wh^T: lower half -> through r*h, upper half -> dagg
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D, i = t - r * D;
      float lo = 0.f, hi = 0.f;
      for (int j = 0; j < D; ++j) {
        const float d = g3[r * D + j];
        lo = fmaf(d, WH_T(i, j), lo);
        hi = fmaf(d, WH_T(D + i, j), hi);
      }
      const float rr = rs[t];
      const float drp = lo * hs[t] * rr * (1.0f - rr);
      g2[t] = drp;
      xs[t] += lo * rr;
      zs[t] = hi;  // dagg so far (z is dead)
      if (t == tid) s_br += drp;
    }
    __syncthreads();
    // dc = dzp Wz^T + drp Wr^T; dpre leaves for the kernel-gradient GEMMs
    for (int t = tid; t < nr * D; t += BS) {
      const int r = t / D, i = t - r * D;
      float lo = 0.f, hi = 0.f;
      for (int j = 0; j < D; ++j) {
        const float dz = g1[r * D + j], dr = g2[r * D + j];
        lo = fmaf(dz, WZ_T(i, j), lo);
        lo = fmaf(dr, WR_T(i, j), lo);
        hi = fmaf(dz, WZ_T(D + i, j), hi);
        hi = fmaf(dr, WR_T(D + i, j), hi);
      }
      dh[row0 * D + t] = xs[t] + lo;
      dagg[row0 * D + t] = zs[t] + hi;
      float* dp = dpre + (row0 + r) * 3 * D;
      dp[i] = g1[t];
      dp[D + i] = g2[t];
      dp[2 * D + i] = g3[t];
    }
  }
  // column sums: thread tid holds (slot tid / D, column tid % D); add the slots in a fixed order
  __syncthreads();
  float* red = smem;  // 5 * BS
  red[tid] = s_bz;
  red[BS + tid] = s_br;
  red[2 * BS + tid] = s_bh;
  red[3 * BS + tid] = s_dg;
  red[4 * BS + tid] = s_db;
  __syncthreads();
  float* mine = small + (int64_t)blockIdx.x * 5 * D;
  for (int q = tid; q < 5 * D; q += BS) {
    const int which = q / D, i = q - which * D;
    float acc = 0.f;
    for (int slot = 0; slot < R; ++slot) acc += red[which * BS + slot * D + i];
    mine[q] = acc;
  }
}
#undef WZ
#undef WR
#undef WH
#undef WZ_T
#undef WR_T
#undef WH_T

// ---------------------------------------------------------------------------------------
// a7 backward for D = 32 on the matrix cores (exact f32 products, v_mfma_f32_16x16x4_f32), the adjoint of
// gated_update_d32_kernel in layer_kernels.hip with the same layout: 16 rows per wave and iteration, rows on the
// MFMA N dimension (lane & 15), features on M, so every accumulator tile (feature 4*(lane>>4)+reg) is directly the
// B operand of the next product.  The gate kernels sit in LDS twice: transposed ((gate, out) rows of 2D inputs,
// stride 68) for the forward recompute, and as stored ((gate, in) rows of D outputs, stride 36) for the products
// with the pre-activation gradients.  Outputs and partial sums are those of gated_update_bwd_kernel.
// ---------------------------------------------------------------------------------------
// sum over the 16 lanes of a DPP row (the rows of one feature quarter), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));  // row_ror:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));  // row_ror:8
  return v;
}
constexpr int kBwT = 68;  // stride of the transposed kernels (rows: gate*32 + out, cols: 2D inputs)
constexpr int kBwN = 36;  // stride of the natural kernels (rows: gate*64 + in, cols: D outputs)

// SAVED: dpre / rh_out arrive holding the training forward's z, r, tanh(t) / r * h (gated_update_d32_kernel's `save`):
// no recompute, no transposed kernels in LDS - 24 of the kernel's 48 MFMAs per 16 rows.
template <bool SAVED>
__global__ __launch_bounds__(kBlock, SAVED ? 2 : 1) void gated_update_bwd_d32_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma, float eps,
    const float* __restrict__ dout, float* __restrict__ dh, float* __restrict__ dagg, float* __restrict__ dpre,
    float* __restrict__ rh_out, float* __restrict__ small, int64_t rows) {
  constexpr int D = 32;
  extern __shared__ __align__(16) float smem[];
  float* wt = smem;                       // 3*D rows x kBwT (not with SAVED)
  float* wn = wt + (SAVED ? 0 : 3 * D * kBwT);  // 3*2D rows x kBwN
  float* wvec = wn + 3 * 2 * D * kBwN;    // bz | br | bh | gamma
  float* red = wvec + 4 * D;              // 4 waves x 5 x D column sums
  for (int t = threadIdx.x; t < 3 * 2 * D * D; t += kBlock) {
    const int gate = t / (2 * D * D), rem = t - gate * 2 * D * D;
    const int jj = rem / D, io = rem - jj * D;  // keras kernel (in = jj, out = io)
    const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
    const float w = Wg[rem];
    if (!SAVED) wt[(gate * D + io) * kBwT + jj] = w;
    wn[(gate * 2 * D + jj) * kBwN + io] = w;
  }
  for (int t = threadIdx.x; t < 4 * D; t += kBlock) {
    const int v = t / D, i = t - v * D;
    wvec[t] = (v == 0 ? bz : v == 1 ? br : v == 2 ? bh : gamma)[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = lane & 15, q = lane >> 4;
  const int64_t ntiles = (rows + 15) >> 4;
  const int64_t wave_id = (int64_t)blockIdx.x * (kBlock >> 6) + wave;
  const int64_t nwaves = (int64_t)gridDim.x * (kBlock >> 6);
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t s_bz[2] = {zero4, zero4}, s_br[2] = {zero4, zero4}, s_bh[2] = {zero4, zero4};
  f32x4_t s_dg[2] = {zero4, zero4}, s_db[2] = {zero4, zero4};
  const f32x4_t gm0 = ldv4(wvec + 3 * D + 4 * q), gm1 = ldv4(wvec + 3 * D + 16 + 4 * q);
  for (int64_t tile = wave_id; tile < ntiles; tile += nwaves) {
    const int64_t row = tile * 16 + a;
    const bool live = row < rows;
    const int64_t rl = live ? row : rows - 1;  // clamped load address; stores and sums are masked through dy = 0
    const f32x4_t h0 = ldv4(h + rl * D + 4 * q), h1 = ldv4(h + rl * D + 16 + 4 * q);
    const f32x4_t a0 = ldv4(agg + rl * D + 4 * q), a1 = ldv4(agg + rl * D + 16 + 4 * q);
    f32x4_t dy0 = ldv4(dout + rl * D + 4 * q), dy1 = ldv4(dout + rl * D + 16 + 4 * q);
    if (!live) {
      dy0 = zero4;
      dy1 = zero4;
    }
    // ---- forward recompute (as gated_update_d32_kernel), or what that kernel kept
    f32x4_t z0, z1, r0, r1, t0, t1;
    if constexpr (SAVED) {
      const float* sv = dpre + rl * 3 * D + 4 * q;
      z0 = ldv4(sv); z1 = ldv4(sv + 16);
      r0 = ldv4(sv + D); r1 = ldv4(sv + D + 16);
      t0 = ldv4(sv + 2 * D); t1 = ldv4(sv + 2 * D + 16);
    } else {
    z0 = ldv4(wvec + 4 * q); z1 = ldv4(wvec + 16 + 4 * q);
    r0 = ldv4(wvec + D + 4 * q); r1 = ldv4(wvec + D + 16 + 4 * q);
    t0 = ldv4(wvec + 2 * D + 4 * q); t1 = ldv4(wvec + 2 * D + 16 + 4 * q);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 32 * half + 16 * u + 4 * q;
        const f32x4_t Az0 = ldv4(wt + (0 * D + a) * kBwT + col), Az1 = ldv4(wt + (0 * D + 16 + a) * kBwT + col);
        const f32x4_t Ar0 = ldv4(wt + (1 * D + a) * kBwT + col), Ar1 = ldv4(wt + (1 * D + 16 + a) * kBwT + col);
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
      z0[i] = sigmoid_exact(z0[i]);
      z1[i] = sigmoid_exact(z1[i]);
      r0[i] = sigmoid_exact(r0[i]);
      r1[i] = sigmoid_exact(r1[i]);
      rh0[i] = r0[i] * h0[i];
      rh1[i] = r1[i] * h1[i];
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 32 * half + 16 * u + 4 * q;
        const f32x4_t Ah0 = ldv4(wt + (2 * D + a) * kBwT + col), Ah1 = ldv4(wt + (2 * D + 16 + a) * kBwT + col);
        const f32x4_t Bv = half == 0 ? (u == 0 ? rh0 : rh1) : (u == 0 ? a0 : a1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t0 = mfma_f32(Ah0[r], Bv[r], t0);
          t1 = mfma_f32(Ah1[r], Bv[r], t1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t0[i] = tanhf(t0[i]);
      t1[i] = tanhf(t1[i]);
    }
    if (live) {
      stv4(rh_out + row * D + 4 * q, rh0);
      stv4(rh_out + row * D + 16 + 4 * q, rh1);
    }
    }
    f32x4_t n0, n1;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      n0[i] = fmaf(z0[i], t0[i] - h0[i], h0[i]);
      n1[i] = fmaf(z1[i], t1[i] - h1[i], h1[i]);
      sum += n0[i] + n1[i];
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
    // ---- LayerNorm backward
    f32x4_t dx0, dx1;
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      n0[i] *= inv;  // x-hat
      n1[i] *= inv;
      dx0[i] = dy0[i] * gm0[i];
      dx1[i] = dy1[i] * gm1[i];
      m1 += dx0[i] + dx1[i];
      m2 = fmaf(dx0[i], n0[i], m2);
      m2 = fmaf(dx1[i], n1[i], m2);
      s_dg[0][i] = fmaf(dy0[i], n0[i], s_dg[0][i]);
      s_dg[1][i] = fmaf(dy1[i], n1[i], s_dg[1][i]);
      s_db[0][i] += dy0[i];
      s_db[1][i] += dy1[i];
    }
    m1 += __shfl_xor(m1, 16);
    m1 += __shfl_xor(m1, 32);
    m2 += __shfl_xor(m2, 16);
    m2 += __shfl_xor(m2, 32);
    m1 *= (1.0f / D);
    m2 *= (1.0f / D);
    f32x4_t dzp0, dzp1, dtp0, dtp1, dh0, dh1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dn0 = inv * (dx0[i] - m1 - n0[i] * m2), dn1 = inv * (dx1[i] - m1 - n1[i] * m2);
      dzp0[i] = dn0 * (t0[i] - h0[i]) * z0[i] * (1.0f - z0[i]);
      dzp1[i] = dn1 * (t1[i] - h1[i]) * z1[i] * (1.0f - z1[i]);
      dtp0[i] = dn0 * z0[i] * (1.0f - t0[i] * t0[i]);
      dtp1[i] = dn1 * z1[i] * (1.0f - t1[i] * t1[i]);
      dh0[i] = dy0[i] + dn0 * (1.0f - z0[i]);
      dh1[i] = dy1[i] + dn1 * (1.0f - z1[i]);
    }
    // ---- dc2 = Wh dtp : rows 0..31 of Wh act on r*h, rows 32..63 on agg
    f32x4_t lo0 = zero4, lo1 = zero4, hi0 = zero4, hi1 = zero4;
    const float* whn = wn + 2 * 2 * D * kBwN;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = 16 * u + 4 * q;
      const f32x4_t A0 = ldv4(whn + (a) * kBwN + col), A1 = ldv4(whn + (16 + a) * kBwN + col);
      const f32x4_t A2 = ldv4(whn + (32 + a) * kBwN + col), A3 = ldv4(whn + (48 + a) * kBwN + col);
      const f32x4_t Bv = u == 0 ? dtp0 : dtp1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        lo0 = mfma_f32(A0[r], Bv[r], lo0);
        lo1 = mfma_f32(A1[r], Bv[r], lo1);
        hi0 = mfma_f32(A2[r], Bv[r], hi0);
        hi1 = mfma_f32(A3[r], Bv[r], hi1);
      }
    }
    f32x4_t drp0, drp1, da0 = hi0, da1 = hi1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      drp0[i] = lo0[i] * h0[i] * r0[i] * (1.0f - r0[i]);
      drp1[i] = lo1[i] * h1[i] * r1[i] * (1.0f - r1[i]);
      dh0[i] = fmaf(lo0[i], r0[i], dh0[i]);
      dh1[i] = fmaf(lo1[i], r1[i], dh1[i]);
      s_bz[0][i] += dzp0[i]; s_bz[1][i] += dzp1[i];
      s_br[0][i] += drp0[i]; s_br[1][i] += drp1[i];
      s_bh[0][i] += dtp0[i]; s_bh[1][i] += dtp1[i];
    }
    // ---- dc = Wz dzp + Wr drp
    lo0 = zero4; lo1 = zero4; hi0 = zero4; hi1 = zero4;
#pragma unroll
    for (int gsel = 0; gsel < 2; ++gsel) {
      const float* wg = wn + gsel * 2 * D * kBwN;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = 16 * u + 4 * q;
        const f32x4_t A0 = ldv4(wg + (a) * kBwN + col), A1 = ldv4(wg + (16 + a) * kBwN + col);
        const f32x4_t A2 = ldv4(wg + (32 + a) * kBwN + col), A3 = ldv4(wg + (48 + a) * kBwN + col);
        const f32x4_t Bv = gsel == 0 ? (u == 0 ? dzp0 : dzp1) : (u == 0 ? drp0 : drp1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo0 = mfma_f32(A0[r], Bv[r], lo0);
          lo1 = mfma_f32(A1[r], Bv[r], lo1);
          hi0 = mfma_f32(A2[r], Bv[r], hi0);
          hi1 = mfma_f32(A3[r], Bv[r], hi1);
        }
      }
    }
    if (live) {
      stv4(dh + row * D + 4 * q, dh0 + lo0);
      stv4(dh + row * D + 16 + 4 * q, dh1 + lo1);
      stv4(dagg + row * D + 4 * q, da0 + hi0);
      stv4(dagg + row * D + 16 + 4 * q, da1 + hi1);
      float* dp = dpre + row * 3 * D;
      stv4(dp + 4 * q, dzp0);
      stv4(dp + 16 + 4 * q, dzp1);
      stv4(dp + D + 4 * q, drp0);
      stv4(dp + D + 16 + 4 * q, drp1);
      stv4(dp + 2 * D + 4 * q, dtp0);
      stv4(dp + 2 * D + 16 + 4 * q, dtp1);
    }
  }
  // ---- column sums: over the 16 rows of a DPP row, then over the 4 waves (fixed order)
  f32x4_t* sums[5] = {s_bz, s_br, s_bh, s_dg, s_db};
#pragma unroll
  for (int w5 = 0; w5 < 5; ++w5)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = row16_sum(sums[w5][blk][i]);
        if (a == 0) red[(wave * 5 + w5) * D + 16 * blk + 4 * q + i] = v;
      }
  __syncthreads();
  float* mine = small + (int64_t)blockIdx.x * 5 * D;
  for (int t = threadIdx.x; t < 5 * D; t += kBlock)
    mine[t] = (red[t] + red[5 * D + t]) + (red[10 * D + t] + red[15 * D + t]);
}

// ---------------------------------------------------------------------------------------
// a7 backward for wide states (D = 64, 128) on the matrix cores: the adjoint of gated_update_wide16_kernel
// (layer_kernels.hip) with its layout - a workgroup of 16 waves owns 64 rows at a time, wave (row tile rt,
// feature group fg) the NT/4 feature tiles 16*(fg*NT/4 + TL) + a of 16 rows; rows on the MFMA M dimension, the
// kernels stream through LDS in double-buffered slices of 16 contraction indices, read as stored for the forward
// recompute and column-wise (W^T) for the products with the pre-activation gradients:
//   P1  z, r   = [h|agg] [Wz|Wr]            P2  t = [r*h|agg] Wh
//   P3  dc2    = dtp Wh^T                    P4  dc = [dzp|drp] [Wz^T ; Wr^T]
// LDS: c = [h|agg] (later [dzp|drp]), r*h (later dtp), two slice buffers, LayerNorm row partials, column sums.
// Outputs and partial sums are those of gated_update_bwd_kernel.
// ---------------------------------------------------------------------------------------
// SAVED: dpre / rh_out arrive holding what the training forward kept (gated_update_wide16_kernel's `save`: z, r, tanh(t)
// in the slots that receive dzp, drp, dtp; r * h) - the recompute passes P1 and P2, half of the kernel's MFMAs, go.
template <int NT, bool SAVED>
__global__ __launch_bounds__(1024) void gated_update_bwd_wide16_kernel(
    const float* __restrict__ h, const float* __restrict__ agg, const float* __restrict__ Wz,
    const float* __restrict__ bz, const float* __restrict__ Wr, const float* __restrict__ br,
    const float* __restrict__ Wh, const float* __restrict__ bh, const float* __restrict__ gamma, float eps,
    const float* __restrict__ dout, float* __restrict__ dh, float* __restrict__ dagg, float* __restrict__ dpre,
    float* __restrict__ rh_out, float* __restrict__ small, int64_t rows, const int32_t* __restrict__ ridx,
    const int32_t* __restrict__ nrows_dev, float* __restrict__ hc, float* __restrict__ aggc, int tile_rows,
    const float* __restrict__ wt) {
  // wt = [Wz^T | Wr^T | Wh^T], each D x 2D (transpose3_kernel, once per call): the slices of P3 / P4 are then rows of
  // 2D consecutive floats - fetched and parked like the forward's (coalesced 4-byte loads two slices ahead, one
  // conflict-free 16-byte LDS store, a ring of three buffers) instead of 64-byte pieces and 4-way conflicting stores.
  // tile_rows = 64, or 16 for launches with too few rows to fill the chip with 64-row tiles (the reference trains with
  // 32 pairs per step): then only the four waves of row tile 0 - one per SIMD - run the four GEMM passes.
  // ridx / nrows_dev (optional): the kernel works on the rows ridx[0 .. *nrows_dev) of h / agg / dout / dh / dagg
  // (impnn_gated_update_rows_bwd: the kept rows of an encode() loop); dpre, r*h and the copies hc / aggc of the rows'
  // inputs are written compactly (list position), which is what the weight-gradient GEMMs behind this kernel read.
  const int64_t max_rows = rows;  // what the launch and every buffer are sized for
  if (nrows_dev) {  // a stale or foreign device count never reaches beyond the sizing (indices are never trusted)
    const int64_t n = *nrows_dev;
    rows = n < 0 ? 0 : (n < max_rows ? n : max_rows);
  }
  constexpr int D = 16 * NT, LDC = 2 * D + 4, LDR = D + 4, LDW = 2 * D, NL = NT / 4;
  extern __shared__ __align__(16) float smem[];
  float* cs = smem;                   // 64 x LDC
  float* rhs = cs + 64 * LDC;         // 64 x LDR
  float* ws = rhs + 64 * LDR;         // 3 x 16 x LDW, element (k = 4*qq + r, column c) at ((qq * LDW + c) * 4 + r)
  float* part = ws + 3 * 16 * LDW;    // 4 x (4 x 64): LayerNorm row partials (sum, sq. deviation, m1, m2)
  float* red = ws;                    // 4 row tiles x 5 x D column sums (after the last tile: the slices are dead)
  static_assert(4 * 5 * D <= 3 * 16 * LDW, "column sums alias the slice ring");
  int32_t* grow_s = reinterpret_cast<int32_t*>(part + 4 * 256);  // 64 global rows of the tile
  const int tid = threadIdx.x;
  int a = tid & 15, q = (tid >> 4) & 3, fg = (tid >> 6) & 3, rt = tid >> 8;  // lane = 16 q + a, wave = 4 rt + fg: a row tile's four waves sit on four SIMDs
  const bool act = 16 * (tid >> 8) < tile_rows;  // (wave-uniform) does this wave's row tile hold rows of the tile?
  constexpr int kPre = 16 * 2 * D / 1024;
  float pre[kPre];
  // (`opaque(tid)`: the slice addresses are recomputed where they are used - a few integer ops - instead of being
  //  hoisted out of the GEMM loops by the compiler: ~60 loop-invariant per-thread addresses had been spilled to scratch
  //  in the prologue and reloaded in every iteration)
  auto opaque = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto opaque0 = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto relane = [&]() {  // same values, re-derived: every phase's element addresses start from here
    const int t_ = opaque0(threadIdx.x);
    a = t_ & 15; q = (t_ >> 4) & 3; fg = (t_ >> 6) & 3; rt = t_ >> 8;
  };
  auto park = [&](float* dst, int ncols) {  // slice element t -> (k = t / ncols, column t % ncols)
    const int tid = opaque(threadIdx.x);
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
      const int t = tid + 1024 * i;
      if (t < 16 * ncols) {
        const int jj = t / ncols, c = t - jj * ncols;
        dst[(((jj >> 2) * LDW + c) << 2) + (jj & 3)] = pre[i];
      }
    }
  };
  auto park_t = [&](float* dst) {           // transposed slices: element t -> (column t / 16, k = t % 16)
    const int tid = opaque(threadIdx.x);
#pragma unroll
    for (int i = 0; i < kPre; ++i) {
      const int t = tid + 1024 * i, c = t >> 4, jj = t & 15;
      dst[(((jj >> 2) * LDW + c) << 2) + (jj & 3)] = pre[i];
    }
  };
  float s_bz[NL] = {}, s_br[NL] = {}, s_bh[NL] = {}, s_dg[NL] = {}, s_db[NL] = {};
  const int64_t ntile = (rows + tile_rows - 1) / tile_rows;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t row0 = tile * tile_rows;
    const int nrt = (int)((rows - row0) < tile_rows ? (rows - row0) : tile_rows);  // rows of this tile
    // per-tile base pointers (scalar) + 32-bit offsets: keeps the per-element addresses out of the register file
    float* rh_t = rh_out + row0 * D;
    float* dpre_t = dpre + row0 * 3 * D;
    __syncthreads();
    if (tid < 64) {
      int64_t gr = tid < nrt ? (ridx ? (int64_t)ridx[row0 + tid] : row0 + tid) : 0;
      grow_s[tid] = (int32_t)(gr < 0 ? 0 : (gr < max_rows ? gr : max_rows - 1));
    }
    __syncthreads();
    for (int t = tid; t < 64 * D; t += 1024) {
      const int r = t / D, c = t - r * D;
      const bool in = r < nrt;
      const int64_t src = (int64_t)grow_s[r] * D + c;
      const float hv = in ? h[src] : 0.f, av = in ? agg[src] : 0.f;
      cs[r * LDC + c] = hv;
      cs[r * LDC + D + c] = av;
#ifndef IMPNN_DIAG_GUB_NOCOPY
      if (hc && in) {
        hc[row0 * D + t] = hv;
        aggc[row0 * D + t] = av;
      }
#endif
    }
    const float* crow = cs + (16 * rt + a) * LDC + 4 * q;
    const float* rrow = rhs + (16 * rt + a) * LDR + 4 * q;
    f32x4_t tt[NL];
    if constexpr (SAVED) {
      __syncthreads();  // the tile of [h | agg] is in LDS
    } else {
    // ---- P1: z, r
    f32x4_t z[NL], rr[NL];
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const int f = 16 * (fg * NL + TL) + a;
      z[TL] = f32x4_t{bz[f], bz[f], bz[f], bz[f]};
      rr[TL] = f32x4_t{br[f], br[f], br[f], br[f]};
    }
    auto fetch1 = [&](int u) {
      const int tid = opaque(threadIdx.x);
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        const int t = tid + 1024 * i, jj = t / (2 * D), c = t - jj * 2 * D;
        pre[i] = c < D ? Wz[(int64_t)(16 * u + jj) * D + c] : Wr[(int64_t)(16 * u + jj) * D + c - D];
      }
    };
    fetch1(0);
    park(ws, 2 * D);
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < 2 * NT; ++u) {
      float* cur = ws + (u & 1) * 16 * LDW;
      float* nxt = ws + ((u + 1) & 1) * 16 * LDW;
      if (u + 1 < 2 * NT) fetch1(u + 1);
      if (act) {
      const f32x4_t av = ldv4(crow + 16 * u);
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const int col = 16 * (fg * NL + TL) + a;
        const f32x4_t b0 = ldv4(cur + ((q * LDW + col) << 2)), b1 = ldv4(cur + ((q * LDW + D + col) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z[TL] = mfma_f32(av[r], b0[r], z[TL]);
          rr[TL] = mfma_f32(av[r], b1[r], rr[TL]);
        }
      }
      }
      if (u + 1 < 2 * NT) park(nxt, 2 * D);
      __syncthreads();
    }
    relane();
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        z[TL][g] = sigmoid_exact(z[TL][g]);
        rr[TL][g] = sigmoid_exact(rr[TL][g]);
        const float v = rr[TL][g] * cs[rl * LDC + f];
        rhs[rl * LDR + f] = v;
        if (rl < nrt) rh_t[rl * D + f] = v;
        // Register relief (the kernel ran 156 VGPRs over its 128 with everything held): r, z and tanh(t) are parked
        // in the tile's own dpre rows - slots that receive drp / dzp / dtp later anyway - and read back by the same
        // thread where they are needed again (L2-resident, same address: program order).
        if (rl < nrt) {
          dpre_t[rl * 3 * D + D + f] = rr[TL][g];
          dpre_t[rl * 3 * D + f] = z[TL][g];
        }
      }
    // ---- P2: t
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const float b2 = bh[16 * (fg * NL + TL) + a];
      tt[TL] = f32x4_t{b2, b2, b2, b2};
    }
    auto fetch2 = [&](int u) {
      const int tid = opaque(threadIdx.x);
#pragma unroll
      for (int i = 0; i < kPre; ++i) {
        const int t = tid + 1024 * i;
        if (t < 16 * D) pre[i] = Wh[(int64_t)(16 * u + t / D) * D + t % D];
      }
    };
    fetch2(0);
    park(ws, D);
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < 2 * NT; ++u) {
      float* cur = ws + (u & 1) * 16 * LDW;
      float* nxt = ws + ((u + 1) & 1) * 16 * LDW;
      if (u + 1 < 2 * NT) fetch2(u + 1);
      if (act) {
      const f32x4_t av = u < NT ? ldv4(rrow + 16 * u) : ldv4(crow + 16 * u);
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const f32x4_t bv = ldv4(cur + ((q * LDW + 16 * (fg * NL + TL) + a) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) tt[TL] = mfma_f32(av[r], bv[r], tt[TL]);
      }
      }
      if (u + 1 < 2 * NT) park(nxt, D);
      __syncthreads();
    }
    }
    // ---- blend, LayerNorm forward statistics
    relane();
    f32x4_t xh[NL];
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        const float hv = cs[rl * LDC + f];
        float tv;
        if constexpr (SAVED) tv = rl < nrt ? dpre_t[rl * 3 * D + 2 * D + f] : 0.f;
        else tv = tanhf(tt[TL][g]);
        const float zz = rl < nrt ? dpre_t[rl * 3 * D + f] : 0.f;
        xh[TL][g] = (1.0f - zz) * hv + zz * tv;
        sum[g] += xh[TL][g];
        if (!SAVED && rl < nrt) dpre_t[rl * 3 * D + 2 * D + f] = tv;
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = row16_sum(sum[g]);
      if (a == 0) part[fg * 64 + 16 * rt + 4 * q + g] = v;
    }
    __syncthreads();
    float mean[4], inv[4], var[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * rt + 4 * q + g;
      mean[g] = ((part[rl] + part[64 + rl]) + (part[128 + rl] + part[192 + rl])) * (1.0f / D);
    }
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        xh[TL][g] -= mean[g];
        var[g] = fmaf(xh[TL][g], xh[TL][g], var[g]);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v = row16_sum(var[g]);
      if (a == 0) part[256 + fg * 64 + 16 * rt + 4 * q + g] = v;
    }
    __syncthreads();
    float m1[4] = {0.f, 0.f, 0.f, 0.f}, m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 256 + 16 * rt + 4 * q + g;
      inv[g] = 1.0f / sqrtf(((part[rl] + part[64 + rl]) + (part[128 + rl] + part[192 + rl])) * (1.0f / D) + eps);
    }
    // ---- LayerNorm backward
    relane();
    f32x4_t dxh[NL], dy[NL];
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const float gm = gamma[16 * (fg * NL + TL) + a];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g;
        dy[TL][g] = rl < nrt ? dout[(int64_t)grow_s[rl] * D + 16 * (fg * NL + TL) + a] : 0.f;
        xh[TL][g] *= inv[g];
        dxh[TL][g] = dy[TL][g] * gm;
        m1[g] += dxh[TL][g];
        m2[g] = fmaf(dxh[TL][g], xh[TL][g], m2[g]);
        s_dg[TL] = fmaf(dy[TL][g], xh[TL][g], s_dg[TL]);
        s_db[TL] += dy[TL][g];
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float v1 = row16_sum(m1[g]), v2 = row16_sum(m2[g]);
      if (a == 0) {
        part[512 + fg * 64 + 16 * rt + 4 * q + g] = v1;
        part[768 + fg * 64 + 16 * rt + 4 * q + g] = v2;
      }
    }
    __syncthreads();  // (also: every wave is done reading r*h from `rhs`)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int rl = 16 * rt + 4 * q + g;
      m1[g] = ((part[512 + rl] + part[576 + rl]) + (part[640 + rl] + part[704 + rl])) * (1.0f / D);
      m2[g] = ((part[768 + rl] + part[832 + rl]) + (part[896 + rl] + part[960 + rl])) * (1.0f / D);
    }
    relane();
    f32x4_t dhA[NL];
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        const float hv = cs[rl * LDC + f];
        const float dn = inv[g] * (dxh[TL][g] - m1[g] - xh[TL][g] * m2[g]);
        const float zz = rl < nrt ? dpre_t[rl * 3 * D + f] : 0.f, tv = rl < nrt ? dpre_t[rl * 3 * D + 2 * D + f] : 0.f;
        const float dzp = dn * (tv - hv) * zz * (1.0f - zz);
        const float dtp = dn * zz * (1.0f - tv * tv);
        dhA[TL][g] = dy[TL][g] + dn * (1.0f - zz);
        s_bz[TL] += dzp;
        s_bh[TL] += dtp;
        rhs[rl * LDR + f] = dtp;  // A operand of P3
        if (rl < nrt) {
          dpre_t[rl * 3 * D + f] = dzp;
          dpre_t[rl * 3 * D + 2 * D + f] = dtp;
        }
      }
    // ---- P3: dc2 = dtp Wh^T  (lo: through r*h, hi: to agg)
    f32x4_t lo[NL], hi[NL];
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) lo[TL] = hi[TL] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // slices of a transposed kernel (rows k0 .. k0 + 15 of a D x 2D block of wt): thread (qq, c) moves rows 4qq .. 4qq + 3
    // of column c
    constexpr int kIT = 4 * 2 * D;
    static_assert(kIT <= 1024, "one item per thread");
    auto fetchT = [&](const float* base, int k0, f32x4_t& pre4) {
      const int t_ = opaque(threadIdx.x);
      if (t_ < kIT) {
        const int qq = t_ / (2 * D), c = t_ - qq * (2 * D);
        const float* src = base + (int64_t)(k0 + 4 * qq) * (2 * D) + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) pre4[r] = src[r * 2 * D];
      }
    };
    auto parkT = [&](float* dst, const f32x4_t& pre4) {
      const int t_ = opaque(threadIdx.x);
      if (t_ < kIT) {
        const int qq = t_ / (2 * D), c = t_ - qq * (2 * D);
        *reinterpret_cast<f32x4_t*>(dst + ((qq * LDW + c) << 2)) = pre4;
      }
    };
    struct OpsT {
      f32x4_t av, b0[NL], b1[NL];
    };
    auto readT = [&](const float* arow, int u, OpsT& o) {
      const float* cur = ws + (u % 3) * 16 * LDW;
      o.av = ldv4(arow + 16 * u);
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const int col = 16 * (fg * NL + TL) + a;
        o.b0[TL] = ldv4(cur + ((q * LDW + col) << 2));
        o.b1[TL] = ldv4(cur + ((q * LDW + D + col) << 2));
      }
    };
    auto mmaT = [&](const OpsT& o) {
      if (!act) return;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[TL] = mfma_f32(o.av[r], o.b0[TL][r], lo[TL]);
          hi[TL] = mfma_f32(o.av[r], o.b1[TL][r], hi[TL]);
        }
    };
    // [lo | hi] += A (rows of `arow`, 16 nsl contraction indices) x the nsl slices that start at `base`
    auto passT = [&](const float* base, int nsl, const float* arow) {
#ifdef IMPNN_DIAG_GUB_NOGEMM  // (knock-out timing builds: tools/gu_pair_bench.py with IMPNN_LIB)
      return;
#endif
      f32x4_t preA, preB;
      fetchT(base, 0, preA);
      fetchT(base, 16, preB);
      parkT(ws, preA);
      __syncthreads();
#pragma unroll 1
      for (int u = 0; u < nsl; u += 2) {
        OpsT o;
        if (u + 2 < nsl) fetchT(base, 16 * (u + 2), preA);
        readT(arow, u, o);
        __builtin_amdgcn_sched_barrier(0);
        parkT(ws + ((u + 1) % 3) * 16 * LDW, preB);
        __builtin_amdgcn_sched_barrier(0);
        mmaT(o);
        __syncthreads();
        if (u + 3 < nsl) fetchT(base, 16 * (u + 3), preB);
        readT(arow, u + 1, o);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 2 < nsl) parkT(ws + ((u + 2) % 3) * 16 * LDW, preA);
        __builtin_amdgcn_sched_barrier(0);
        mmaT(o);
        __syncthreads();
      }
    };
    passT(wt + (int64_t)2 * 2 * D * D, NT, rrow);
    relane();
    // lo / hi go on as P4's accumulators: they start from the direct terms (dh: dy + dn (1 - z) + r dc2_lo, dagg: dc2_hi).
    // c = [h | agg] becomes [dzp | drp] element by element - a thread reads and replaces its own h, nobody reads agg
    // after P2 - so only the barrier in front of P4's first operand read is needed.
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        const float hv = cs[rl * LDC + f], rv = rl < nrt ? dpre_t[rl * 3 * D + D + f] : 0.f;
        const float dzp = rl < nrt ? dpre_t[rl * 3 * D + f] : 0.f;  // parked above
        const float drp = lo[TL][g] * hv * rv * (1.0f - rv);
        s_br[TL] += drp;
        if (rl < nrt) dpre_t[rl * 3 * D + D + f] = drp;
        cs[rl * LDC + f] = dzp;
        cs[rl * LDC + D + f] = drp;
        lo[TL][g] = fmaf(lo[TL][g], rv, dhA[TL][g]);
      }
    // ---- P4: dc = [dzp|drp] [Wz^T ; Wr^T]
    passT(wt, 2 * NT, crow);  // [Wz^T ; Wr^T] are consecutive rows of wt
    relane();
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rl = 16 * rt + 4 * q + g, f = 16 * (fg * NL + TL) + a;
        if (rl < nrt) {
          const int64_t dst = (int64_t)grow_s[rl] * D + f;
          dh[dst] = lo[TL][g];
          dagg[dst] = hi[TL][g];
        }
      }
  }
  // ---- column sums: the 4 row quarters of a wave, then the 4 row tiles (fixed order)
  float* sums[5] = {s_bz, s_br, s_bh, s_dg, s_db};
#pragma unroll
  for (int w5 = 0; w5 < 5; ++w5)
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      float v = sums[w5][TL];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q == 0) red[(rt * 5 + w5) * D + 16 * (fg * NL + TL) + a] = v;
    }
  __syncthreads();
  float* mine = small + (int64_t)blockIdx.x * 5 * D;
  for (int t = tid; t < 5 * D; t += 1024)
    mine[t] = (red[t] + red[5 * D + t]) + (red[10 * D + t] + red[15 * D + t]);
}

// WT = [Wz^T | Wr^T | Wh^T] (each D x 2D) for the large-D variant above (32x32 LDS tiles)
__global__ void transpose3_kernel(const float* __restrict__ Wz, const float* __restrict__ Wr,
                                  const float* __restrict__ Wh, float* __restrict__ WT, int D) {
  __shared__ float tile[32][33];
  const float* W = blockIdx.z == 0 ? Wz : (blockIdx.z == 1 ? Wr : Wh);
  float* T = WT + (int64_t)blockIdx.z * 2 * D * D;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;  // W is (2D x D): rows r, cols c
  for (int y = threadIdx.y; y < 32; y += blockDim.y) {
    const int r = r0 + y, c = c0 + threadIdx.x;
    tile[y][threadIdx.x] = (r < 2 * D && c < D) ? W[(int64_t)r * D + c] : 0.f;
  }
  __syncthreads();
  for (int y = threadIdx.y; y < 32; y += blockDim.y) {
    const int c = c0 + y, r = r0 + threadIdx.x;       // T is (D x 2D): T[c][r]
    if (c < D && r < 2 * D) T[(int64_t)c * 2 * D + r] = tile[threadIdx.x][y];
  }
}

// C[M x N] (per chunk) = sum over the chunk's rows r of A(r,m) * B(r,n), every operand addressed with a row and
// a column stride; the M axis may be the concatenation [A1 | A2] of two operands (split at Mh).
// blockIdx = (chunk, tile, problem); 64 x 32 output tile, thread owns 4 x 2, 32 rows staged in LDS per iteration.
// Output of chunk c of problem p: out + (p * nchunk + c) * M * N, row-major.
struct GemmProblems {
  const float* A1[3];
  const float* A2[3];
  const float* B[3];
};
constexpr int kGM = 64, kGN = 32, kGR = 32;  // output tile, rows staged per iteration

__global__ __launch_bounds__(kBlock) void strided_gemm_splitk_kernel(GemmProblems ga, float* __restrict__ out,
                                                                     int64_t rows, int M, int Mh, int N,
                                                                     int64_t a_rs, int64_t a_cs, int64_t b_rs,
                                                                     int64_t b_cs, int nchunk, int tilesN,
                                                                     const int32_t* __restrict__ rows_dev) {
  if (rows_dev) {  // (the contraction length lives on the device: kept rows of a batch; never beyond the sizing)
    const int64_t n = *rows_dev;
    rows = n < 0 ? 0 : (n < rows ? n : rows);
  }
  // LDS tiles of kGR contraction rows; row strides chosen so that the four k-rows of one MFMA step fall into
  // disjoint bank groups (80 = 64 + 16, 48 = 32 + 16 floats)
  constexpr int kLA = kGM + 16, kLB = kGN + 16;
  __shared__ __align__(16) float As[kGR * kLA];
  __shared__ __align__(16) float Bs[kGR * kLB];
  const int chunk = blockIdx.x, tile = blockIdx.y, prob = blockIdx.z;
  const int tm0 = (tile / tilesN) * kGM, tn0 = (tile % tilesN) * kGN;
  const float* A1 = ga.A1[prob];
  const float* A2 = ga.A2[prob];
  const float* B = ga.B[prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int64_t per = (rows + nchunk - 1) / nchunk;
  const int64_t r_lo = (int64_t)chunk * per, r_hi = r_lo + per < rows ? r_lo + per : rows;
  // exact-f32 MFMA: wave w owns output rows m = 16w .. 16w+15 of the 64 x 32 tile (two 16 x 16 tiles);
  // contraction rows on K: lane (a, q) feeds A[row 4s+q][16w + a] and B[row 4s+q][16t + a] straight from the tiles
  f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // Fast path (the GatedUpdate weight gradients: both operands row-major, whole tiles): 16-byte loads, the next
  // kGR rows in flight in registers under the MFMAs of the current ones.
  const bool fast = a_cs == 1 && b_cs == 1 && tm0 + kGM <= M && tn0 + kGN <= N && (Mh & 3) == 0 && (a_rs & 3) == 0 &&
                    (b_rs & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(A1) | reinterpret_cast<uintptr_t>(A2) | reinterpret_cast<uintptr_t>(B)) & 15u) == 0;
  if (fast) {
    constexpr int kA4 = kGR * kGM / 4 / kBlock, kB4 = kGR * kGN / 4 / kBlock;  // float4 per thread: 2 and 1
    static_assert(kA4 * kBlock * 4 == kGR * kGM && kB4 * kBlock * 4 == kGR * kGN, "tile / block mismatch");
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t ra[kA4], rb[kB4];
    const float* ap[kA4];
    const float* bp[kB4];
    int ar[kA4], br_[kB4], ao[kA4], bo[kB4];
#pragma unroll
    for (int i = 0; i < kA4; ++i) {
      const int t = tid + kBlock * i, r = t / (kGM / 4), m = tm0 + 4 * (t % (kGM / 4));
      ar[i] = r;
      ao[i] = r * kLA + 4 * (t % (kGM / 4));
      ap[i] = (m < Mh ? A1 + m : A2 + (m - Mh)) + (int64_t)r * a_rs;
    }
#pragma unroll
    for (int i = 0; i < kB4; ++i) {
      const int t = tid + kBlock * i, r = t / (kGN / 4), n = tn0 + 4 * (t % (kGN / 4));
      br_[i] = r;
      bo[i] = r * kLB + 4 * (t % (kGN / 4));
      bp[i] = B + n + (int64_t)r * b_rs;
    }
    auto fetch = [&](int64_t r0) {
      const int nr = (int)((r_hi - r0) < kGR ? (r_hi - r0) : kGR);
#pragma unroll
      for (int i = 0; i < kA4; ++i)
        ra[i] = ar[i] < nr ? *reinterpret_cast<const f32x4_t*>(ap[i] + r0 * a_rs) : zero4;
#pragma unroll
      for (int i = 0; i < kB4; ++i)
        rb[i] = br_[i] < nr ? *reinterpret_cast<const f32x4_t*>(bp[i] + r0 * b_rs) : zero4;
    };
    if (r_lo < r_hi) fetch(r_lo);
    for (int64_t r0 = r_lo; r0 < r_hi; r0 += kGR) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kA4; ++i) *reinterpret_cast<f32x4_t*>(As + ao[i]) = ra[i];
#pragma unroll
      for (int i = 0; i < kB4; ++i) *reinterpret_cast<f32x4_t*>(Bs + bo[i]) = rb[i];
      __syncthreads();
      if (r0 + kGR < r_hi) fetch(r0 + kGR);
#pragma unroll
      for (int st = 0; st < kGR / 4; ++st) {
        const float av = As[(4 * st + q) * kLA + 16 * wave + a];
        const float b0 = Bs[(4 * st + q) * kLB + a], b1 = Bs[(4 * st + q) * kLB + 16 + a];
        acc0 = mfma_f32(av, b0, acc0);
        acc1 = mfma_f32(av, b1, acc1);
      }
    }
  } else
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += kGR) {
    const int nr = (int)((r_hi - r0) < kGR ? (r_hi - r0) : kGR);
    __syncthreads();
    if (a_cs == 1) {  // columns contiguous: lanes walk m
      for (int t = tid; t < kGR * kGM; t += kBlock) {
        const int r = t / kGM, mm = t % kGM, m = tm0 + mm;
        float v = 0.f;
        if (r < nr && m < M) v = m < Mh ? A1[(r0 + r) * a_rs + m] : A2[(r0 + r) * a_rs + (m - Mh)];
        As[r * kLA + mm] = v;
      }
    } else {          // rows contiguous (a transposed operand): lanes walk r
      for (int t = tid; t < kGR * kGM; t += kBlock) {
        const int mm = t / kGR, r = t % kGR, m = tm0 + mm;
        float v = 0.f;
        if (r < nr && m < M) v = m < Mh ? A1[(r0 + r) * a_rs + m * a_cs] : A2[(r0 + r) * a_rs + (m - Mh) * a_cs];
        As[r * kLA + mm] = v;
      }
    }
    if (b_cs == 1) {
      for (int t = tid; t < kGR * kGN; t += kBlock) {
        const int r = t / kGN, nn = t % kGN, n = tn0 + nn;
        Bs[r * kLB + nn] = (r < nr && n < N) ? B[(r0 + r) * b_rs + n] : 0.f;
      }
    } else {
      for (int t = tid; t < kGR * kGN; t += kBlock) {
        const int nn = t / kGR, r = t % kGR, n = tn0 + nn;
        Bs[r * kLB + nn] = (r < nr && n < N) ? B[(r0 + r) * b_rs + n * b_cs] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < kGR / 4; ++st) {
      const float av = As[(4 * st + q) * kLA + 16 * wave + a];
      const float b0 = Bs[(4 * st + q) * kLB + a], b1 = Bs[(4 * st + q) * kLB + 16 + a];
      acc0 = mfma_f32(av, b0, acc0);
      acc1 = mfma_f32(av, b1, acc1);
    }
  }
  float* o = out + ((int64_t)prob * nchunk + chunk) * M * N;
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // accumulator: row 16w + 4q + g, column 16t + a
    const int m = tm0 + 16 * wave + 4 * q + g;
    if (m < M) {
      if (tn0 + a < N) o[(int64_t)m * N + tn0 + a] = acc0[g];
      if (tn0 + 16 + a < N) o[(int64_t)m * N + tn0 + 16 + a] = acc1[g];
    }
  }
}

// The same product for wide states (M = 2D >= 128, N = D >= 64, both operands row-major, whole tiles): a 128 x 64
// output tile per workgroup, a 32 x 64 block (2 x 4 accumulator tiles) per wave - 6 LDS reads feed 8 MFMAs per k step
// instead of 3 feeding 2, and a 32-row stage holds 64 MFMAs per wave between its two barriers instead of 16 (the
// 64 x 32 tiles above ran the D = 128 weight gradients at 27 % of the f32 MFMA peak: 433 us per call at batch 4096).
constexpr int kBM = 128, kBN = 64;
__global__ __launch_bounds__(kBlock) void strided_gemm_splitk_big_kernel(GemmProblems ga, float* __restrict__ out,
                                                                         int64_t rows, int M, int Mh, int N,
                                                                         int64_t a_rs, int64_t b_rs, int nchunk,
                                                                         int tilesN, const int32_t* __restrict__ rows_dev) {
  if (rows_dev) {
    const int64_t n = *rows_dev;
    rows = n < 0 ? 0 : (n < rows ? n : rows);
  }
  constexpr int kLA = kBM + 16, kLB = kBN + 16;  // row strides: the four k rows of an MFMA step on disjoint bank groups
  __shared__ __align__(16) float As[kGR * kLA];
  __shared__ __align__(16) float Bs[kGR * kLB];
  const int chunk = blockIdx.x, tile = blockIdx.y, prob = blockIdx.z;
  const int tm0 = (tile / tilesN) * kBM, tn0 = (tile % tilesN) * kBN;
  const float* A1 = ga.A1[prob];
  const float* A2 = ga.A2[prob];
  const float* B = ga.B[prob];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int64_t per = (rows + nchunk - 1) / nchunk;
  const int64_t r_lo = (int64_t)chunk * per, r_hi = r_lo + per < rows ? r_lo + per : rows;
  constexpr int kA4 = kGR * kBM / 4 / kBlock, kB4 = kGR * kBN / 4 / kBlock;  // float4 per thread: 4 and 2
  static_assert(kA4 * kBlock * 4 == kGR * kBM && kB4 * kBlock * 4 == kGR * kBN, "tile / block mismatch");
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = zero4;
  f32x4_t ra[kA4], rb[kB4];
  auto fetch = [&](int64_t r0) {
    const int nr = (int)((r_hi - r0) < kGR ? (r_hi - r0) : kGR);
#pragma unroll
    for (int i = 0; i < kA4; ++i) {
      const int t = tid + kBlock * i, r = t / (kBM / 4), m = tm0 + 4 * (t % (kBM / 4));
      const float* ap = (m < Mh ? A1 + m : A2 + (m - Mh)) + (r0 + r) * a_rs;
      ra[i] = r < nr ? ldv4(ap) : zero4;
    }
#pragma unroll
    for (int i = 0; i < kB4; ++i) {
      const int t = tid + kBlock * i, r = t / (kBN / 4), n = tn0 + 4 * (t % (kBN / 4));
      rb[i] = r < nr ? ldv4(B + n + (r0 + r) * b_rs) : zero4;
    }
  };
  if (r_lo < r_hi) fetch(r_lo);
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += kGR) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kA4; ++i) {
      const int t = tid + kBlock * i;
      stv4(As + (t / (kBM / 4)) * kLA + 4 * (t % (kBM / 4)), ra[i]);
    }
#pragma unroll
    for (int i = 0; i < kB4; ++i) {
      const int t = tid + kBlock * i;
      stv4(Bs + (t / (kBN / 4)) * kLB + 4 * (t % (kBN / 4)), rb[i]);
    }
    __syncthreads();
    if (r0 + kGR < r_hi) fetch(r0 + kGR);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < kGR / 4; ++st) {
      float av[2], bv[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[(4 * st + q) * kLA + 32 * wave + 16 * i + a];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[(4 * st + q) * kLB + 16 * j + a];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f32(av[i], bv[j], acc[i][j]);
    }
  }
  float* o = out + ((int64_t)prob * nchunk + chunk) * M * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // accumulator (i, j): row 32w + 16i + 4q + g, column 16j + a
      const int m = tm0 + 32 * wave + 16 * i + 4 * q + g;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[(int64_t)m * N + tn0 + 16 * j + a] = acc[i][j][g];
    }
}

// dparams = fixed-order sums of the partials (canonical layout, see above).  One wave per output: lane l adds
// partials l, l+64, ... in order, then a fixed butterfly - the same association on every run.
__global__ void gated_update_reduce_kernel(const float* __restrict__ small, const float* __restrict__ gpart,
                                           float* __restrict__ dparams, int nblk, int nchunk, int D, int accumulate,
                                           int wblocks) {
  const int DD2 = 2 * D * D;
  if ((int)blockIdx.x < wblocks) {
    // the three kernel gradients.  Few chunk partials (wide states: <= 64): one THREAD per element, the partials added
    // in chunk order, consecutive threads on consecutive addresses (one wave per element read its 20 partials 128 KB
    // apart: 27 us per call for 8 MB).  Many partials (atom_dim 32 at large batches: 342): one wave per element.
    if (nchunk <= 64) {
      const int t = blockIdx.x * blockDim.x + threadIdx.x;
      if (t >= 3 * DD2) return;
      const int gate = t / DD2, off = t - gate * DD2;
      const float* src = gpart + (int64_t)gate * nchunk * DD2 + off;
      float acc = 0.f;
      int c = 0;
      for (; c + 8 <= nchunk; c += 8) {  // 8 partials in flight, added in chunk order
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = src[(int64_t)(c + u) * DD2];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += x[u];
      }
      for (; c < nchunk; ++c) acc += src[(int64_t)c * DD2];
      const int q = gate * (DD2 + D) + off;
      dparams[q] = accumulate ? dparams[q] + acc : acc;
      return;
    }
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (t >= 3 * DD2) return;
    const int gate = t / DD2, off = t - gate * DD2;
    const float* src = gpart + (int64_t)gate * nchunk * DD2 + off;
    float acc = 0.f;
    for (int c = lane; c < nchunk; c += 64) acc += src[(int64_t)c * DD2];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const int q = gate * (DD2 + D) + off;
    if (lane == 0) dparams[q] = accumulate ? dparams[q] + acc : acc;
    return;
  }
  // the five vectors (bz, br, bh, gamma, beta): one wave per element over the nblk per-workgroup column sums
  const int e = (((int)blockIdx.x - wblocks) * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (e >= 5 * D) return;
  const int q = e < 3 * D ? (e / D) * (DD2 + D) + DD2 + (e % D) : 3 * (DD2 + D) + (e - 3 * D);
  const float* src = small + e;
  float acc = 0.f;
  int c = lane;
  for (; c + 7 * 64 < nblk; c += 8 * 64) {  // eight slices in flight (a lane's loads are 5 D floats apart), added in order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src[(int64_t)(c + 64 * u) * 5 * D];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += x[u];
  }
  for (; c < nblk; c += 64) acc += src[(int64_t)c * 5 * D];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) dparams[q] = accumulate ? dparams[q] + acc : acc;
}

// ---------------------------------------------------------------------------------------
// Optimizer step: keras.optimizers.Adam(lr, clipnorm) as the trainers configure it
// (train_viscosity.py:227-230).  kAdamSplit workgroups per variable: each forms the variable's gradient norm itself
// (the same thread-strided sum in the same order in every workgroup, so all of them scale by the identical factor - no
// partials buffer, no second launch) and updates its own 1/kAdamSplit of the elements; with one workgroup per variable
// the step took as long as one workgroup needs for the largest variable (172 us at atom_dim 128: bond_transform, 131 K
// floats through one workgroup's 20 GB/s):
//   g <- g * clipnorm / max(||g||_2, clipnorm)          (tf.clip_by_norm, per variable; clipnorm <= 0: off)
//   m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2
//   w <- w - lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)
// The variable table holds device pointers: 4 per variable (w, g, m, v) and the element count.
// ---------------------------------------------------------------------------------------
__global__ void incr_step_kernel(long long* step) { *step += 1; }
constexpr int kAdamSplit = 16;

__global__ __launch_bounds__(1024) void adam_clipnorm_kernel(const unsigned long long* __restrict__ table,
                                                             const long long* __restrict__ sizes, float lr,
                                                             float b1, float b2, float eps, float clipnorm,
                                                             float corr1, float corr2,
                                                             const long long* __restrict__ step_dev) {
  if (step_dev) {  // step counter in device memory (a captured hipGraph replays with a new step every time)
    const float t = (float)*step_dev;
    corr1 = 1.0f - powf(b1, t);
    corr2 = 1.0f - powf(b2, t);
  }
  __shared__ float red[16];
  __shared__ float scale_s;
  const int var = blockIdx.x;
  float* w = reinterpret_cast<float*>(table[4 * var + 0]);
  const float* g = reinterpret_cast<const float*>(table[4 * var + 1]);
  float* m = reinterpret_cast<float*>(table[4 * var + 2]);
  float* v = reinterpret_cast<float*>(table[4 * var + 3]);
  const long long n = sizes[var];
  if ((long long)blockIdx.y * 4096 >= n && blockIdx.y > 0) return;  // no elements for this workgroup
  float ss = 0.f;
  if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {  // 16-byte loads (every workgroup of the variable takes this path or none)
    const long long n4 = n >> 2;
    for (long long t = threadIdx.x; t < n4; t += blockDim.x) {
      const f32x4_t x = ldv4(g + 4 * t);
      ss = fmaf(x[0], x[0], ss);
      ss = fmaf(x[1], x[1], ss);
      ss = fmaf(x[2], x[2], ss);
      ss = fmaf(x[3], x[3], ss);
    }
    for (long long t = 4 * n4 + threadIdx.x; t < n; t += blockDim.x) ss = fmaf(g[t], g[t], ss);
  } else {
    for (long long t = threadIdx.x; t < n; t += blockDim.x) ss = fmaf(g[t], g[t], ss);
  }
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
  long long per = (n + gridDim.y - 1) / gridDim.y;
  if (per < 4096) per = 4096;  // small variables stay with their first workgroup(s)
  const long long lo = (long long)blockIdx.y * per, hi = lo + per < n ? lo + per : n;
  for (long long t = lo + threadIdx.x; t < hi; t += blockDim.x) {
    const float gg = g[t] * sc;
    const float mm = b1 * m[t] + (1.0f - b1) * gg;
    const float vv = b2 * v[t] + (1.0f - b2) * gg * gg;
    m[t] = mm;
    v[t] = vv;
    w[t] -= alpha * mm / (sqrtf(vv) + eps);
  }
}

// ---------------------------------------------------------------------------------------
// f1 for training: everything after GlobalSumPool, forward from the individual weight tensors (no packing) and
// its backward, one launch each (train_viscosity.py:189,197-214 + models/layers.py:10-49;
// train_melting_point.py:173,191-198).  Tensor order = the packed order of impnn_model_head:
//   Wfp_cat | bfp_cat | Wfp_an | bfp_an | Wp_cat | bp_cat | Wp_an | bp_an | kind 0: Wv | bv ; kind 1: Wh | bh | Wo | bo
// Backward: 8 samples per workgroup, 32 threads per sample; the forward is recomputed; parameter gradients are
// summed in LDS per workgroup and ADDED to the individual gradient buffers with float atomics.
// ---------------------------------------------------------------------------------------
constexpr int kHdMax = 64, kHdSPB = 8, kHdTensors = 12;
constexpr int kHdXMax = 128;  // widest pooled state (config 5: atom_dim 128); fp_size and mixing_size stay <= kHdMax
struct HeadTensors {
  const float* w[kHdTensors];
  float* g[kHdTensors];
  int off[kHdTensors + 1];
  int n;
  float l2[kHdTensors];  // keras l2(lambda) per tensor (0: none); used by the loss entries only
};
// loss = mean_b (pred_b - y_b)^2 + sum_t l2_t * sum(W_t^2)   (keras "mse" + kernel_regularizer, train_viscosity.py:189,229)
struct HeadLoss {
  const float* y;        // (B); null: the kernels behave as the plain head entries
  const float* dloss;    // backward: device scalar, the gradient of the loss value
  float* loss_out;       // forward: device scalar
  float* partial;        // forward: one squared-error sum per workgroup
  unsigned int* counter; // forward: arrival ticket, zero before the first call, left at zero by every call
  float inv_B;
};
__device__ __forceinline__ float head_l2(const HeadTensors& ht, int sgm) {
  float v = ht.l2[0];
#pragma unroll
  for (int q = 1; q < kHdTensors; ++q) v = sgm == q ? ht.l2[q] : v;
  return v;
}
// deterministic workgroup sum of one value per thread (256 threads); result valid in every thread
__device__ __forceinline__ float head_block_sum(float v, float* red) {
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float softplus_stable(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ int head_total(const HeadTensors& ht) {
  int tot = ht.off[1];
#pragma unroll
  for (int q = 2; q <= kHdTensors; ++q) tot = ht.n == q ? ht.off[q] : tot;
  return tot;
}
// segment of packed index t.  Constant indices only: the table is a kernel argument, a dynamic index into it would
// be a dependent load from the kernarg segment per probe.
__device__ __forceinline__ int head_segment(const HeadTensors& ht, int t, int* base) {
  int sgm = 0, b = 0;
#pragma unroll
  for (int q = 1; q < kHdTensors; ++q)
    if (q < ht.n && t >= ht.off[q]) sgm = q, b = ht.off[q];
  *base = b;
  return sgm;
}
__device__ __forceinline__ const float* head_wptr(const HeadTensors& ht, int sgm) {
  const float* p = ht.w[0];
#pragma unroll
  for (int q = 1; q < kHdTensors; ++q) p = sgm == q ? ht.w[q] : p;
  return p;
}
__device__ __forceinline__ float* head_gptr(const HeadTensors& ht, int sgm) {
  float* p = ht.g[0];
#pragma unroll
  for (int q = 1; q < kHdTensors; ++q) p = sgm == q ? ht.g[q] : p;
  return p;
}

__device__ __forceinline__ void head_load_weights(const HeadTensors& ht, float* ws) {
  const int total = head_total(ht);
  constexpr int kU = 8;  // independent loads in flight per thread
  for (int t0 = threadIdx.x; t0 < total; t0 += blockDim.x * kU) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int t = t0 + u * blockDim.x;
      int base;
      const int sgm = head_segment(ht, t, &base);
      v[u] = t < total ? head_wptr(ht, sgm)[t - base] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int t = t0 + u * blockDim.x;
      if (t < total) ws[t] = v[u];
    }
  }
}

// forward up to the mixed vector; returns through LDS: fpre (pre-activation of the fingerprint Dense), fp, ppre, mix
__device__ __forceinline__ void head_forward_mix(const float* ws, const float* xs, float* fpre, float* ppre, float* mix,
                                                 int sl, int jj, int D, int F, int Mx) {
  const float* Wfp[2] = {ws, ws + D * F + F};
  const float* wp = ws + 2 * (D * F + F);
  const float* Wp[2] = {wp, wp + F * Mx + Mx};
  for (int g = 0; g < 2; ++g)
    for (int j = jj; j < F; j += 32) {
      float acc = Wfp[g][D * F + j];
      const float* x = xs + (sl * 2 + g) * kHdXMax;
      for (int i = 0; i < D; ++i) acc = fmaf(x[i], Wfp[g][i * F + j], acc);
      fpre[(sl * 2 + g) * kHdMax + j] = acc;
    }
  __syncthreads();
  for (int j = jj; j < Mx; j += 32) {
    float m = 0.f;
    for (int g = 0; g < 2; ++g) {
      float acc = Wp[g][F * Mx + j];
      const float* x = fpre + (sl * 2 + g) * kHdMax;
      for (int i = 0; i < F; ++i) acc = fmaf(fmaxf(x[i], 0.f), Wp[g][i * Mx + j], acc);
      ppre[(sl * 2 + g) * kHdMax + j] = acc;
      m += fmaxf(acc, 0.f);
    }
    mix[sl * kHdMax + j] = m;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void model_head_tensors_kernel(int kind, const float* __restrict__ pc,
                                                                 const float* __restrict__ pa,
                                                                 const float* __restrict__ T, HeadTensors ht,
                                                                 float* __restrict__ out, int B, int D, int F, int Mx,
                                                                 HeadLoss hl) {
  extern __shared__ __align__(16) float hsm[];
  __shared__ float red[256];
  __shared__ float sq[kHdSPB];
  __shared__ int is_last;
  const int total = head_total(ht);
  float* ws = hsm;
  float* xs = ws + ((total + 3) & ~3);
  float* fpre = xs + kHdSPB * 2 * kHdXMax;
  float* ppre = fpre + kHdSPB * 2 * kHdMax;
  float* mix = ppre + kHdSPB * 2 * kHdMax;
  float* hid = mix + kHdSPB * kHdMax;
  const int tid = threadIdx.x, sl = tid >> 5, jj = tid & 31;
  const int b = blockIdx.x * kHdSPB + sl;
  const bool live = b < B;
  head_load_weights(ht, ws);
  for (int g = 0; g < 2; ++g)
    for (int i = jj; i < D; i += 32) xs[(sl * 2 + g) * kHdXMax + i] = live ? (g == 0 ? pc : pa)[(int64_t)b * D + i] : 0.f;
  __syncthreads();
  head_forward_mix(ws, xs, fpre, ppre, mix, sl, jj, D, F, Mx);
  const float* wt = ws + 2 * (D * F + F) + 2 * (F * Mx + Mx);
  const float* mx = mix + sl * kHdMax;
  if (kind == 0) {
    if (jj < 3) {
      float acc = wt[Mx * 3 + jj];
      for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], wt[i * 3 + jj], acc);
      hid[sl * kHdMax + jj] = acc;
    }
    __syncthreads();
    if (jj == 0) {
      float pred = 0.f;
      if (live) {
        const float* vp = hid + sl * kHdMax;
        const float Bc = fminf(fmaxf(softplus_stable(vp[1]), 0.f), 20.f);
        const float Cc = fminf(fmaxf(softplus_stable(vp[2]), 0.1f), 50.f);
        pred = vp[0] + Bc / (T[b] / 100.0f + Cc + 1e-6f);
        if (out) out[b] = pred;
      }
      if (hl.y) sq[sl] = live ? (pred - hl.y[b]) * (pred - hl.y[b]) : 0.f;
    }
  } else {
    const float* Wh = wt;
    const float* bh = Wh + Mx * F;
    const float* Wo = bh + F;
    for (int j = jj; j < F; j += 32) {
      float acc = bh[j];
      for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], Wh[i * F + j], acc);
      hid[sl * kHdMax + j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (jj == 0) {
      float acc = Wo[F];
      for (int j = 0; j < F; ++j) acc = fmaf(hid[sl * kHdMax + j], Wo[j], acc);
      if (live && out) out[b] = acc;
      if (hl.y) sq[sl] = live ? (acc - hl.y[b]) * (acc - hl.y[b]) : 0.f;
    }
  }
  if (!hl.y) return;
  // ---- loss: workgroup sums in sample order, then the LAST workgroup to arrive adds them in workgroup order
  __syncthreads();
  if (tid == 0) {
    float sum = 0.f;
    for (int q = 0; q < kHdSPB; ++q) sum += sq[q];
    __hip_atomic_store(&hl.partial[blockIdx.x], sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned int ticket = atomicAdd(hl.counter, 1u);
    is_last = ticket == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  float v = 0.f;
  for (int i = tid; i < (int)gridDim.x; i += 256)
    v += __hip_atomic_load(&hl.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float se = head_block_sum(v, red);
  float reg = 0.f;
  for (int t = tid; t < total; t += 256) {
    int base;
    const float lam = head_l2(ht, head_segment(ht, t, &base));
    reg = fmaf(lam * ws[t], ws[t], reg);
  }
  reg = head_block_sum(reg, red);
  if (tid == 0) {
    hl.loss_out[0] = se * hl.inv_B + reg;
    *hl.counter = 0u;
  }
}

// per-sample vectors of the backward, kHdMax floats each, in LDS: the parameter gradients are outer products of these
enum { kVX0, kVX1, kVFp0, kVFp1, kVDfp0, kVDfp1, kVDpr0, kVDpr1, kVMix, kVTop, kVHid, kVOne, kHdVecs };
constexpr int kHdVecStride = 2 * kHdXMax + (kHdVecs - 2) * kHdMax;
__device__ __forceinline__ int head_vec_off(int which) {
  return which < 2 ? which * kHdXMax : 2 * kHdXMax + (which - 2) * kHdMax;
}

// 1024 threads: the first 256 walk the samples (8 samples x 32 lanes, as the forward kernel), all of them stage the
// weights, form the parameter gradients' outer products and flush them - the three phases that scale with the packed
// weight count (25 K floats at atom_dim 128) and made the kernel ~96 us at every batch below 2048, alone on the stream
// between the two halves of a training step.
__global__ __launch_bounds__(1024) void model_head_bwd_kernel(int kind, const float* __restrict__ pc,
                                                             const float* __restrict__ pa, const float* __restrict__ T,
                                                             HeadTensors ht, const float* __restrict__ dout,
                                                             float* __restrict__ dpc, float* __restrict__ dpa, int B,
                                                             int D, int F, int Mx, HeadLoss hl) {
  extern __shared__ __align__(16) float hsm[];
  const int total = head_total(ht);
  const int tpad = (total + 3) & ~3;
  float* ws = hsm;
  float* dws = ws + tpad;  // parameter-gradient sums of this workgroup; element t is owned by thread t % 256
  float* vec = dws + tpad;  // [kHdSPB][kHdVecStride]: the two pooled states (kHdXMax each), then 10 vectors of kHdMax
  float* fpre = vec + kHdSPB * kHdVecStride;
  float* ppre = fpre + kHdSPB * 2 * kHdMax;
  const int tid = threadIdx.x;
  const bool worker = tid < 32 * kHdSPB;                     // a lane of a sample; the others skip the per-sample loops
  const int sl = worker ? tid >> 5 : 0, jj = worker ? (tid & 31) : (1 << 30);
  float* my = vec + sl * kHdVecStride;
  auto V = [&](int which) { return my + head_vec_off(which); };
  head_load_weights(ht, ws);
  for (int t = tid; t < tpad; t += blockDim.x) dws[t] = 0.f;
  const int o_fp[2] = {0, D * F + F};
  const int o_p0 = 2 * (D * F + F);
  const int o_p[2] = {o_p0, o_p0 + F * Mx + Mx};
  const int o_t = o_p0 + 2 * (F * Mx + Mx);

  for (int b0 = blockIdx.x * kHdSPB; b0 < B; b0 += gridDim.x * kHdSPB) {
    const int b = b0 + sl;
    const bool live = worker && b < B;
    __syncthreads();  // the previous group's outer products are done with vec
    // xs of head_forward_mix = vectors kVX0,kVX1 (contiguous)
    for (int g = 0; g < 2; ++g)
      for (int i = jj; i < D; i += 32) V(kVX0 + g)[i] = live ? (g == 0 ? pc : pa)[(int64_t)b * D + i] : 0.f;
    __syncthreads();
    {  // forward (same arithmetic as head_forward_mix, on this kernel's vector layout)
      for (int g = 0; g < 2; ++g)
        for (int j = jj; j < F; j += 32) {
          float acc = ws[o_fp[g] + D * F + j];
          const float* x = V(kVX0 + g);
          for (int i = 0; i < D; ++i) acc = fmaf(x[i], ws[o_fp[g] + i * F + j], acc);
          fpre[(sl * 2 + g) * kHdMax + j] = acc;
          V(kVFp0 + g)[j] = fmaxf(acc, 0.f);
        }
      __syncthreads();
      for (int j = jj; j < Mx; j += 32) {
        float m = 0.f;
        for (int g = 0; g < 2; ++g) {
          float acc = ws[o_p[g] + F * Mx + j];
          const float* x = V(kVFp0 + g);
          for (int i = 0; i < F; ++i) acc = fmaf(x[i], ws[o_p[g] + i * Mx + j], acc);
          ppre[(sl * 2 + g) * kHdMax + j] = acc;
          m += fmaxf(acc, 0.f);
        }
        V(kVMix)[j] = m;
      }
      __syncthreads();
    }
    const float* mx = V(kVMix);
    // gradient of the prediction: given (plain head), or 2 (pred - y) / B * dloss once pred is known (loss entries)
    const float gscale = hl.y ? 2.0f * hl.inv_B * hl.dloss[0] : 0.f;
    float d = (live && !hl.y) ? dout[b] : 0.f;
    float* top = V(kVTop);
    // ---- top of the head: kVTop = gradient of [A,b,c] (kind 0) / of the hidden pre-activation (kind 1)
    if (kind == 0) {
      if (jj < 3) {
        float acc = ws[o_t + Mx * 3 + jj];
        for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], ws[o_t + i * 3 + jj], acc);
        V(kVHid)[jj] = acc;
      }
      __syncthreads();
      const float* vp = V(kVHid);
      const float sp1 = softplus_stable(vp[1]), sp2 = softplus_stable(vp[2]);
      const float Bc = fminf(fmaxf(sp1, 0.f), 20.f), Cc = fminf(fmaxf(sp2, 0.1f), 50.f);
      const float den = (live ? T[b] : 300.f) / 100.0f + Cc + 1e-6f;
      if (hl.y && live) d = gscale * (vp[0] + Bc / den - hl.y[b]);
      if (jj == 0) V(kVOne)[0] = d;
      float dvp[3];
      dvp[0] = d;
      dvp[1] = (sp1 >= 0.f && sp1 <= 20.f) ? d / den / (1.0f + expf(-vp[1])) : 0.f;  // clamp passes inside [min,max]
      dvp[2] = (sp2 >= 0.1f && sp2 <= 50.f) ? -d * Bc / (den * den) / (1.0f + expf(-vp[2])) : 0.f;
      if (jj < 3) top[jj] = jj == 0 ? dvp[0] : (jj == 1 ? dvp[1] : dvp[2]);
      for (int i = jj; i < Mx; i += 32) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc = fmaf(ws[o_t + i * 3 + c], dvp[c], acc);
        V(kVDpr0)[i] = ppre[(sl * 2 + 0) * kHdMax + i] > 0.f ? acc : 0.f;
        V(kVDpr1)[i] = ppre[(sl * 2 + 1) * kHdMax + i] > 0.f ? acc : 0.f;
      }
    } else {
      const int o_bh = o_t + Mx * F, o_wo = o_bh + F;
      for (int j = jj; j < F; j += 32) {
        float acc = ws[o_bh + j];
        for (int i = 0; i < Mx; ++i) acc = fmaf(mx[i], ws[o_t + i * F + j], acc);
        V(kVHid)[j] = fmaxf(acc, 0.f);
      }
      __syncthreads();
      if (hl.y && live) {
        float pred = ws[o_wo + F];
        for (int j = 0; j < F; ++j) pred = fmaf(V(kVHid)[j], ws[o_wo + j], pred);  // as the forward kernel
        d = gscale * (pred - hl.y[b]);
      }
      if (jj == 0) V(kVOne)[0] = d;
      for (int j = jj; j < F; j += 32) top[j] = V(kVHid)[j] > 0.f ? ws[o_wo + j] * d : 0.f;
      __syncthreads();
      for (int i = jj; i < Mx; i += 32) {
        float acc = 0.f;
        for (int j = 0; j < F; ++j) acc = fmaf(ws[o_t + i * F + j], top[j], acc);
        V(kVDpr0)[i] = ppre[(sl * 2 + 0) * kHdMax + i] > 0.f ? acc : 0.f;
        V(kVDpr1)[i] = ppre[(sl * 2 + 1) * kHdMax + i] > 0.f ? acc : 0.f;
      }
    }
    __syncthreads();
    // ---- projections (relu) -> fingerprints (relu) -> pooled
    for (int g = 0; g < 2; ++g) {
      const float* dpr = V(kVDpr0 + g);
      for (int i = jj; i < F; i += 32) {
        float acc = 0.f;
        for (int j = 0; j < Mx; ++j) acc = fmaf(ws[o_p[g] + i * Mx + j], dpr[j], acc);
        V(kVDfp0 + g)[i] = fpre[(sl * 2 + g) * kHdMax + i] > 0.f ? acc : 0.f;
      }
    }
    __syncthreads();
    for (int g = 0; g < 2; ++g) {
      const float* dfg = V(kVDfp0 + g);
      float* dx = g == 0 ? dpc : dpa;
      for (int i = jj; i < D; i += 32) {
        float acc = 0.f;
        for (int j = 0; j < F; ++j) acc = fmaf(ws[o_fp[g] + i * F + j], dfg[j], acc);
        if (live) dx[(int64_t)b * D + i] = acc;
      }
    }
    // ---- parameter gradients: element t of the packed layout = sum over the samples of a[i] * b[j]
    for (int t = tid; t < total; t += blockDim.x) {
      int base;
      const int sgm = head_segment(ht, t, &base);
      const int loc = t - base;
      int va, vb, ncols;  // va < 0: a bias (sum of b[j])
      switch (sgm) {
        case 0: va = kVX0, vb = kVDfp0, ncols = F; break;
        case 1: va = -1, vb = kVDfp0, ncols = F; break;
        case 2: va = kVX1, vb = kVDfp1, ncols = F; break;
        case 3: va = -1, vb = kVDfp1, ncols = F; break;
        case 4: va = kVFp0, vb = kVDpr0, ncols = Mx; break;
        case 5: va = -1, vb = kVDpr0, ncols = Mx; break;
        case 6: va = kVFp1, vb = kVDpr1, ncols = Mx; break;
        case 7: va = -1, vb = kVDpr1, ncols = Mx; break;
        case 8: va = kVMix, vb = kVTop, ncols = kind == 0 ? 3 : F; break;
        case 9: va = -1, vb = kVTop, ncols = kind == 0 ? 3 : F; break;
        case 10: va = kVHid, vb = kVOne, ncols = 1; break;  // Wo (F,1): hidden * dout
        default: va = -1, vb = kVOne, ncols = 1; break;     // bo
      }
      const int i = loc / ncols, j = loc - i * ncols;
      float acc = 0.f;
      if (va < 0) {
#pragma unroll
        for (int q = 0; q < kHdSPB; ++q) acc += vec[q * kHdVecStride + head_vec_off(vb) + j];
      } else {
#pragma unroll
        for (int q = 0; q < kHdSPB; ++q)
          acc = fmaf(vec[q * kHdVecStride + head_vec_off(va) + i], vec[q * kHdVecStride + head_vec_off(vb) + j], acc);
      }
      dws[t] += acc;
    }
  }
  __syncthreads();
  const float reg_scale = (hl.y && blockIdx.x == 0) ? 2.0f * hl.dloss[0] : 0.f;  // d/dW of l2 * sum(W^2), added once
  for (int t = tid; t < total; t += blockDim.x) {
    int base;
    const int sgm = head_segment(ht, t, &base);
    const float v = dws[t] + reg_scale * head_l2(ht, sgm) * ws[t];
    if (v != 0.f) atomicAdd(head_gptr(ht, sgm) + (t - base), v);
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
  const size_t lds = sizeof(float) * (size_t)vocab * dim;
  const int use_lds = lds <= 64 * 1024 && rows * dim >= 8 * (int64_t)vocab * dim;
  if (use_lds && lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)embed_gather_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  embed_gather_bwd_kernel<<<grid_for(rows * dim, use_lds ? 512 : 256 * 8), kBlock, use_lds ? lds : 0, s>>>(
      ids, dout, dtable, rows, vocab, dim, use_lds);
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

int64_t bmm_message_typed_bwd_workspace_ints(int B, int E, int Vb) { return (int64_t)4 * (Vb + 1) + (int64_t)B * E; }

// the batch's valid edges, counting-sorted by bond type, in `workspace` (layout at the kernels above)
static int launch_edge_type_sort(const int32_t* bond_ids, const int32_t* conn, int32_t* workspace, int B, int N, int E,
                                 int Vb, hipStream_t s) {
  const int64_t BE = (int64_t)B * E;
  int32_t* cnt = workspace;
  int32_t* start = cnt + (Vb + 1);
  int32_t* cursor = start + (Vb + 1);
  int32_t* segbase = cursor + (Vb + 1);
  int32_t* order = segbase + (Vb + 1);
  if (BE <= 1024 * kSortSmallPer && Vb <= 1024) {
    edge_type_sort_small_kernel<<<1, 1024, 0, s>>>(conn, bond_ids, cnt, start, cursor, segbase, order, (int)BE, N, Vb);
    return check_launch("edge_type_sort_small");
  }
  // (a kernel, not hipMemsetAsync: the call must behave the same inside a captured hipGraph)
  zero_ints_kernel<<<grid_for(Vb + 1), kBlock, 0, s>>>(cnt, Vb + 1);
  if (int rc = check_launch("zero_ints")) return rc;
  const int sort_grid = grid_for(BE / 8 + 1, 512);
  edge_type_hist_kernel<<<sort_grid, kBlock, 0, s>>>(conn, bond_ids, cnt, BE, N, Vb);
  if (int rc = check_launch("edge_type_hist")) return rc;
  edge_type_prefix_kernel<<<1, 64, 0, s>>>(cnt, start, cursor, segbase, Vb);
  if (int rc = check_launch("edge_type_prefix")) return rc;
  edge_type_scatter_kernel<<<sort_grid, kBlock, 0, s>>>(conn, bond_ids, cursor, order, BE, N, Vb);
  return check_launch("edge_type_scatter");
}

int launch_bmm_message_typed_sorted(const float* h, const int32_t* bond_ids, const int32_t* conn, const float* A,
                                    float* m, int32_t* workspace, int B, int N, int E, int D, int Vb, int sorted_ready,
                                    hipStream_t s) {
  if (Vb > kMaxTypes) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_sorted: Vb=%d too large", Vb);
  if (D > 128) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_sorted: D=%d > 128", D);
  const int64_t BE = (int64_t)B * E;
  if (!(sorted_ready & 1))
    if (int rc = launch_edge_type_sort(bond_ids, conn, workspace, B, N, E, Vb, s)) return rc;
  if (!(sorted_ready & 2)) {  // (bit 1: the caller's buffer still holds the zero rows of an earlier call on this batch)
    zero_invalid_messages_kernel<<<grid_for(BE * D), kBlock, 0, s>>>(conn, bond_ids, m, BE, N, D, Vb);
    if (int rc = check_launch("zero_invalid_messages")) return rc;
  }
  const int32_t* start = workspace + (Vb + 1);
  const int32_t* segbase = workspace + 3 * (Vb + 1);
  const int32_t* order = workspace + 4 * (Vb + 1);
  const int64_t max_segs = (BE + kSeg - 1) / kSeg + Vb;
  if (D % 16 == 0 && ((reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(h)) & 15u) == 0) {
    const size_t lm = sizeof(float) * ((size_t)D * (D + 4) + (size_t)kSeg * (D + 4));
    if (lm > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)bmm_message_typed_seg_mfma_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
    // wide states: 16 waves (4 per SIMD) share the segment's 32 output tiles, so LDS reads overlap the MFMAs
    // wide states: several segments per workgroup (the matrix copy is amortised); keep >= ~4 workgroups per CU of work
    int spw = D >= 64 ? (int)(max_segs / 1024) : 1;
    spw = spw < 1 ? 1 : (spw > 8 ? 8 : spw);
    bmm_message_typed_seg_mfma_kernel<<<(int)((max_segs + spw - 1) / spw), D >= 64 ? 1024 : kBlock, lm, s>>>(
        h, conn, A, m, start, segbase, order, N, E, D, Vb, spw);
    return check_launch("bmm_message_typed_seg_mfma");
  }
  const size_t lds = sizeof(float) * ((size_t)D * (D + 1) + (size_t)kSeg * D);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)bmm_message_typed_seg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  bmm_message_typed_seg_kernel<<<(int)max_segs, kBlock, lds, s>>>(h, conn, A, m, start, segbase, order, N, E, D, Vb);
  return check_launch("bmm_message_typed_seg");
}

int launch_bmm_message_typed_bwd(const float* h, const int32_t* bond_ids, const int32_t* conn, const float* A,
                                 const float* dm, float* dh, float* dA, int32_t* workspace, int B, int N, int E,
                                 int D, int Vb, int sorted_ready, int from_agg, hipStream_t s, float* du) {
  if (Vb > kMaxTypes) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_bwd: Vb=%d too large", Vb);
  if (D > 128) return fail(IMPNN_E_UNSUPPORTED, "bmm_message_typed_bwd: D=%d > 128", D);
  const int64_t BE = (int64_t)B * E;
  int32_t* cnt = workspace;
  int32_t* start = cnt + (Vb + 1);
  int32_t* cursor = start + (Vb + 1);
  int32_t* segbase = cursor + (Vb + 1);
  int32_t* order = segbase + (Vb + 1);
  if (!sorted_ready)  // the sort depends on (conn, bond_ids) only: forward and backward of the S layers of an ion share it
    if (int rc = launch_edge_type_sort(bond_ids, conn, workspace, B, N, E, Vb, s)) return rc;
  const int64_t max_segs = (BE + kSeg - 1) / kSeg + Vb;
  const bool al16 = ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(dm)) & 15u) == 0 &&
                    (reinterpret_cast<uintptr_t>(conn) & 7u) == 0;
  const char* force = getenv("IMPNN_MESSAGE_BWD");  // diagnostics: "valu" / "mfma"
  const bool want_mfma = force ? force[0] == 'm' : true;
  if ((D == 64 || D == 128) && Vb <= kBwdMfmaMaxTypes && al16 && want_mfma) {
    const size_t lm = sizeof(float) * ((size_t)D * (D + 4) + 2 * (size_t)kSeg * (D + 4)) + sizeof(int32_t) * (2 * kSeg + 2 * (size_t)(Vb + 1));
    if (du && (reinterpret_cast<uintptr_t>(du) & 15u)) return fail(IMPNN_E_BADARG, "message backward: the per-edge buffer must be 16B aligned");
    // one workgroup per type without atomics on dA ("mo", diagnostics) was never faster than balanced segment ranges:
    // 26.8 vs 26.1 us per call at batch 32, 423 vs 331 us at batch 4096 (VALU kernel: 41.7 / 887 us)
    const int owner = force && force[1] == 'o' ? 1 : 0;
    const int grid = owner ? Vb : (int)(max_segs < 256 ? max_segs : 256);
    if (D == 128) {
      (void)hipFuncSetAttribute((const void*)bmm_message_typed_bwd_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
      bmm_message_typed_bwd_mfma_kernel<8><<<grid, 1024, lm, s>>>(h, conn, A, dm, dh, dA, start, segbase, order, N, E, Vb, from_agg, owner, du);
    } else {
      (void)hipFuncSetAttribute((const void*)bmm_message_typed_bwd_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm);
      bmm_message_typed_bwd_mfma_kernel<4><<<grid, 1024, lm, s>>>(h, conn, A, dm, dh, dA, start, segbase, order, N, E, Vb, from_agg, owner, du);
    }
    if (int rc = check_launch("bmm_message_typed_bwd (mfma)")) return rc;
    // the per-edge vectors, added into dh at their source rows in edge-slot order (conn[b, e, 0]: stride 2)
    if (du) return launch_reduce_scatter_add(du, conn, 2, dh, B, N, E, D, s, 1);
    return IMPNN_OK;
  }
  const size_t lds = sizeof(float) * ((size_t)D * D + 2 * (size_t)kSeg * D);
  // wide states need > 64 KB of LDS (one workgroup per CU): 16 waves instead of 4 keep every SIMD busy
  const int threads = D >= 64 ? 1024 : kBlock;
  const int acc = (D * D + threads - 1) / threads;
#define LAUNCH(ACC)                                                                                          \
  do {                                                                                                       \
    if (lds > 48 * 1024)                                                                                     \
      (void)hipFuncSetAttribute((const void*)bmm_message_typed_bwd_kernel<ACC>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    bmm_message_typed_bwd_kernel<ACC><<<(int)max_segs, threads, lds, s>>>(h, conn, A, dm, dh, dA, start, segbase, \
                                                                           order, N, E, D, Vb, from_agg, du);\
  } while (0)
  if (acc <= 1) LAUNCH(1);
  else if (acc <= 4) LAUNCH(4);
  else if (acc <= 16) LAUNCH(16);
  else LAUNCH(64);
#undef LAUNCH
  if (du) {
    if (int rc = check_launch("bmm_message_typed_bwd")) return rc;
    return launch_reduce_scatter_add(du, conn, 2, dh, B, N, E, D, s, 1);
  }
  return check_launch("bmm_message_typed_bwd");
}

// out[M x N] = sum_r A(r,m) B(r,n) in one pass (no split: deterministic, no workspace)
int launch_strided_gemm(const float* A, const float* B, float* out, int64_t rows, int M, int N, int64_t a_rs,
                        int64_t a_cs, int64_t b_rs, int64_t b_cs, hipStream_t s) {
  GemmProblems ga{};
  ga.A1[0] = A; ga.A2[0] = A; ga.B[0] = B;
  const int tiles_n = (N + kGN - 1) / kGN, tiles_m = (M + kGM - 1) / kGM;
  strided_gemm_splitk_kernel<<<dim3(1, tiles_m * tiles_n, 1), kBlock, 0, s>>>(ga, out, rows, M, M, N, a_rs, a_cs, b_rs,
                                                                              b_cs, 1, tiles_n, nullptr);
  return check_launch("strided_gemm");
}

int launch_bond_type_matrices_bwd(const float* tb, const float* W, const float* dA, float* dW, float* dtb, int Vb,
                                  int K, int D, int accumulate, hipStream_t s) {
  const int DD = D * D;
  if (K >= 64) {
    if (accumulate) return fail(IMPNN_E_UNSUPPORTED, "bond_type_matrices_bwd: accumulate needs K < 64");  // GEMM-shaped: dW (K x DD) = Tb^T dA over Vb rows; dTb (Vb x K) = dA W^T over DD rows
    if (int rc = launch_strided_gemm(tb, dA, dW, Vb, K, DD, K, 1, DD, 1, s)) return rc;
    return launch_strided_gemm(dA, W, dtb, DD, Vb, K, 1, DD, 1, DD, s);
  }
  bond_type_matrices_bwd_w_kernel<<<dim3((DD + kBlock - 1) / kBlock, K), kBlock, 0, s>>>(tb, dA, dW, Vb, K, DD, accumulate);
  if (int rc = check_launch("bond_type_matrices_bwd_w")) return rc;
  const int64_t waves = (int64_t)Vb * K;
  bond_type_matrices_bwd_t_kernel<<<(int)((waves * 64 + kBlock - 1) / kBlock), kBlock, 0, s>>>(W, dA, dtb, Vb, K, DD,
                                                                                               accumulate);
  return check_launch("bond_type_matrices_bwd_t");
}

static int gu_main_blocks(int64_t rows, int D) {
  // tiles of the main kernel: kBlock / D rows (generic kernels), 16 or 64 rows (the wide matrix-core kernel)
  const int R = (D == 64 || D == 128) ? gu_wide_tile_rows(rows) : kBlock / D;
  const int64_t ntile = (rows + R - 1) / R;
  return (int)(ntile < 1024 ? (ntile < 1 ? 1 : ntile) : 1024);
}
// 128 x 64 output tiles: atom_dim 64 / 128 from 8 K rows (below that the 64 x 32 tiles give 4x the workgroups: 1.445
// vs 1.478 ms per step at batch 32)
static bool gu_big_tiles(int D, int64_t rows) { return D % kBN == 0 && (2 * D) % kBM == 0 && rows >= 8192; }
static int gu_tiles(int D, int64_t rows, int* tiles_n) {
  const int gn = gu_big_tiles(D, rows) ? kBN : kGN, gm = gu_big_tiles(D, rows) ? kBM : kGM;
  const int tn = (D + gn - 1) / gn, tm = (2 * D + gm - 1) / gm;
  if (tiles_n) *tiles_n = tn;
  return tm * tn;
}
static int gu_chunks(int64_t rows, int D) {
  const int tiles = gu_tiles(D, rows, nullptr);
  // ~1024 workgroups of the 64 x 32 tiles, ~768 (three resident per CU) of the 128 x 64 ones
  int64_t want = gu_big_tiles(D, rows) ? (768 + 3 * tiles - 1) / (3 * tiles) : (1024 + 3 * tiles - 1) / (3 * tiles);
  const int64_t cap = (rows + 255) / 256;  // at least 256 contraction rows per chunk: fewer partials to write and add
  if (want > cap) want = cap;
  return (int)(want < 1 ? 1 : want);
}

int gated_update_bwd_blocks(int64_t rows, int D) { return gu_main_blocks(rows, D); }

int64_t gated_update_param_floats(int D) { return 3 * ((int64_t)2 * D * D + D) + 2 * D; }

// workspace (floats): dpre rows*3D | r*h rows*D | small nblk*5D | GEMM partials 3*nchunk*2D*D | W^T 3*2D*D
// (row-list form: + compact copies of the rows' h and agg, rows*2D)
int64_t gated_update_bwd_workspace(int64_t rows, int D, bool row_list) {
  return rows * 4 * D + (int64_t)gu_main_blocks(rows, D) * 5 * D + (int64_t)3 * gu_chunks(rows, D) * 2 * D * D +
         (int64_t)3 * 2 * D * D + (row_list ? rows * 2 * D : 0);
}

int launch_gated_update_bwd(const float* h, const float* agg, const float* Wz, const float* bz, const float* Wr,
                            const float* br, const float* Wh, const float* bh, const float* gamma, float eps,
                            const float* dout, float* dh, float* dagg, float* dparams, float* workspace,
                            int64_t rows, int D, int accumulate, hipStream_t s, const int32_t* ridx,
                            const int32_t* nrows_dev, float* saved) {
  if (D > kBlock || kBlock % D != 0)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_bwd: atom_dim %d must divide %d", D, kBlock);
  if (ridx && D != 64 && D != 128)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd: atom_dim %d (the row-list form covers 64 and 128)", D);
  const int R = kBlock / D;
  const int nblk = gu_main_blocks(rows, D), nchunk = gu_chunks(rows, D);
  int nsmall = nblk;  // slices of `small` the reduction reads
  if (saved && !(D == 32 || D == 64 || D == 128))
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd_saved: atom_dim %d (covers 32, 64 and 128)", D);
  if (saved && D == 32 && ridx)
    return fail(IMPNN_E_UNSUPPORTED, "gated_update_rows_bwd_saved: atom_dim 32 takes no row list");
  // saved (impnn_gated_update_rows_train's buffer, [z | r | tanh(t)] then r * h): used in place of the workspace's
  // first two regions and CONSUMED - it leaves holding the pre-activation gradients
  float* dpre = saved ? saved : workspace;
  float* rh = dpre + rows * 3 * D;
  float* small = workspace + rows * 4 * D;
  float* gpart = small + (int64_t)nblk * 5 * D;
  float* wt = gpart + (int64_t)3 * nchunk * 2 * D * D;
  float* hc = ridx ? wt + (int64_t)3 * 2 * D * D : nullptr;  // compact copies of the listed rows (row-list form)
  float* aggc = ridx ? hc + rows * D : nullptr;
  size_t lds = sizeof(float) * ((size_t)10 * R * D + 4 * R);
  if (lds < sizeof(float) * 5 * kBlock) lds = sizeof(float) * 5 * kBlock;
  const size_t wlds = sizeof(float) * (size_t)3 * 2 * D * (D + 1);
  const bool al16 = ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(agg) | reinterpret_cast<uintptr_t>(dout) |
                      reinterpret_cast<uintptr_t>(dh) | reinterpret_cast<uintptr_t>(dagg) |
                      reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(dpre)) & 15u) == 0;
  if ((ridx || saved) && !al16) return fail(IMPNN_E_BADARG, "gated_update_rows_bwd: tensors must be 16B aligned");
  if ((D == 64 || D == 128) && al16) {
    const size_t lw = sizeof(float) * ((size_t)64 * (2 * D + 4) + 64 * (D + 4) + 3 * 16 * 2 * D + 4 * 256) + 64 * sizeof(int32_t);
    const int tile_rows = gu_wide_tile_rows(rows);  // (as the forward kernel: 16-row tiles below ~8 K rows)
    transpose3_kernel<<<dim3((D + 31) / 32, (2 * D + 31) / 32, 3), dim3(32, 8), 0, s>>>(Wz, Wr, Wh, wt, D);
    if (int rc = check_launch("transpose3")) return rc;
    const int64_t tiles64 = (rows + tile_rows - 1) / tile_rows;
    const int nb = (int)(tiles64 < nblk ? tiles64 : nblk);  // every launched workgroup writes its slice of `small`
    nsmall = nb;                                            // (zeros when the row list ends before its tiles)
#define BWD16(NT_, SV_)                                                                                              \
    do {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)gated_update_bwd_wide16_kernel<NT_, SV_>,                                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lw);                                 \
      gated_update_bwd_wide16_kernel<NT_, SV_><<<nb, 1024, lw, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh, \
                                                                   dagg, dpre, rh, small, rows, ridx, nrows_dev, hc,  \
                                                                   aggc, tile_rows, wt);                              \
    } while (0)
    if (D == 64) {
      if (saved) BWD16(4, true); else BWD16(4, false);
    } else {
      if (saved) BWD16(8, true); else BWD16(8, false);
    }
#undef BWD16
  } else if (D == 32 && al16) {
    const size_t l32 = sizeof(float) * ((size_t)3 * 32 * kBwT + 3 * 64 * kBwN + 4 * 32 + 4 * 5 * 32);
    if (l32 > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gated_update_bwd_d32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)l32);
    if (saved) {
      const size_t l32s = l32 - sizeof(float) * 3 * 32 * kBwT;  // no transposed kernels
      gated_update_bwd_d32_kernel<true><<<nblk, kBlock, l32s, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh, dagg,
                                                                 dpre, rh, small, rows);
    } else {
      gated_update_bwd_d32_kernel<false><<<nblk, kBlock, l32, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh, dagg,
                                                                  dpre, rh, small, rows);
    }
  } else if (lds + wlds <= 120 * 1024) {
    lds += wlds;
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gated_update_bwd_kernel<true, kBlock>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    gated_update_bwd_kernel<true, kBlock><<<nblk, kBlock, lds, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh,
                                                                    dagg, dpre, rh, small, nullptr, rows, D, R);
  } else {
    constexpr int kBig = 1024;
    const int Rb = kBig / D;
    size_t lb = sizeof(float) * ((size_t)10 * Rb * D + 4 * Rb);
    if (lb < sizeof(float) * 5 * kBig) lb = sizeof(float) * 5 * kBig;
    if (lb > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)gated_update_bwd_kernel<false, kBig>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
    transpose3_kernel<<<dim3((D + 31) / 32, (2 * D + 31) / 32, 3), dim3(32, 8), 0, s>>>(Wz, Wr, Wh, wt, D);
    if (int rc = check_launch("transpose3")) return rc;
    gated_update_bwd_kernel<false, kBig><<<nblk, kBig, lb, s>>>(h, agg, Wz, bz, Wr, br, Wh, bh, gamma, eps, dout, dh,
                                                                dagg, dpre, rh, small, wt, rows, D, Rb);
  }
  if (int rc = check_launch("gated_update_bwd")) return rc;
  int tiles_n = 1;
  const int tiles = gu_tiles(D, rows, &tiles_n);
  GemmProblems ga;
  const float* gh = ridx ? hc : h;      // the GEMMs contract over the listed rows: their compact copies
  const float* ga_ = ridx ? aggc : agg;
  ga.A1[0] = gh; ga.A2[0] = ga_; ga.B[0] = dpre;
  ga.A1[1] = gh; ga.A2[1] = ga_; ga.B[1] = dpre + D;
  ga.A1[2] = rh; ga.A2[2] = ga_; ga.B[2] = dpre + 2 * D;
  const bool al16g = ((reinterpret_cast<uintptr_t>(gh) | reinterpret_cast<uintptr_t>(ga_) | reinterpret_cast<uintptr_t>(rh) |
                       reinterpret_cast<uintptr_t>(dpre)) & 15u) == 0;
  if (gu_big_tiles(D, rows) && al16g)
    strided_gemm_splitk_big_kernel<<<dim3(nchunk, tiles, 3), kBlock, 0, s>>>(ga, gpart, rows, 2 * D, D, D, D, 3 * D, nchunk,
                                                                             tiles_n, ridx ? nrows_dev : nullptr);
  else if (gu_big_tiles(D, rows))
    return fail(IMPNN_E_BADARG, "gated_update_bwd: tensors must be 16B aligned at atom_dim %d", D);
  else
    strided_gemm_splitk_kernel<<<dim3(nchunk, tiles, 3), kBlock, 0, s>>>(ga, gpart, rows, 2 * D, D, D, D, 1, 3 * D, 1,
                                                                         nchunk, tiles_n, ridx ? nrows_dev : nullptr);
  if (int rc = check_launch("strided_gemm_splitk")) return rc;
  const int64_t welems = (int64_t)3 * 2 * D * D * (nchunk <= 64 ? 1 : 64);  // a thread or a wave per kernel element
  const int wblocks = (int)((welems + kBlock - 1) / kBlock), vblocks = (5 * D * 64 + kBlock - 1) / kBlock;
  gated_update_reduce_kernel<<<wblocks + vblocks, kBlock, 0, s>>>(small, gpart, dparams, nsmall, nchunk, D, accumulate,
                                                                 wblocks);
  return check_launch("gated_update_reduce");
}

int launch_adam_clipnorm(const void* table, const void* sizes, int n_vars, int64_t step, int64_t* step_dev, float lr,
                         float b1, float b2, float eps, float clipnorm, hipStream_t s) {
  float corr1 = 1.0f, corr2 = 1.0f;
  if (step_dev) {
    incr_step_kernel<<<1, 1, 0, s>>>(reinterpret_cast<long long*>(step_dev));
    if (int rc = check_launch("incr_step")) return rc;
  } else {
    corr1 = 1.0f - powf(b1, (float)step);
    corr2 = 1.0f - powf(b2, (float)step);
  }
  adam_clipnorm_kernel<<<dim3(n_vars, kAdamSplit), 1024, 0, s>>>(static_cast<const unsigned long long*>(table),
                                               static_cast<const long long*>(sizes), lr, b1, b2, eps, clipnorm,
                                               corr1, corr2, reinterpret_cast<const long long*>(step_dev));
  return check_launch("adam_clipnorm");
}

static int head_tensor_table(int kind, const float* const* weights, float* const* grads, int D, int F, int Mx,
                             HeadTensors* ht, const float* l2 = nullptr) {
  const int sizes0[10] = {D * F, F, D * F, F, F * Mx, Mx, F * Mx, Mx, Mx * 3, 3};
  const int sizes1[12] = {D * F, F, D * F, F, F * Mx, Mx, F * Mx, Mx, Mx * F, F, F, 1};
  ht->n = kind == 0 ? 10 : 12;
  int off = 0;
  for (int i = 0; i < ht->n; ++i) {
    if (!weights[i]) return fail(IMPNN_E_BADARG, "model_head: null weight tensor %d", i);
    ht->w[i] = weights[i];
    ht->g[i] = grads ? grads[i] : nullptr;
    if (grads && !grads[i]) return fail(IMPNN_E_BADARG, "model_head_bwd: null gradient tensor %d", i);
    ht->off[i] = off;
    ht->l2[i] = l2 ? l2[i] : 0.f;
    off += kind == 0 ? sizes0[i] : sizes1[i];
  }
  ht->off[ht->n] = off;
  return IMPNN_OK;
}

int64_t model_head_loss_workspace_floats(int B) { return (B + kHdSPB - 1) / kHdSPB + 4; }

int launch_model_head_tensors(int kind, const float* pc, const float* pa, const float* T, const float* const* weights,
                              float* out, int B, int D, int F, int Mx, hipStream_t s, const float* l2, const float* y,
                              float* loss_out, float* workspace) {
  if (D > kHdXMax || F > kHdMax || Mx > kHdMax)
    return fail(IMPNN_E_UNSUPPORTED, "model_head: dims D=%d (<= %d) F=%d Mx=%d (<= %d)", D, kHdXMax, F, Mx, kHdMax);
  HeadTensors ht{};
  if (int rc = head_tensor_table(kind, weights, nullptr, D, F, Mx, &ht, l2)) return rc;
  HeadLoss hl{};
  if (y) {  // workspace: [0] arrival counter (zero between calls) | [4...] one partial per workgroup
    hl.y = y;
    hl.loss_out = loss_out;
    hl.counter = reinterpret_cast<unsigned int*>(workspace);
    hl.partial = workspace + 4;
    hl.inv_B = 1.0f / (float)B;
  }
  const size_t lds =
      sizeof(float) * (((size_t)ht.off[ht.n] + 3) / 4 * 4 + (size_t)kHdSPB * (2 * kHdXMax + 6 * kHdMax));
  if (lds > 156 * 1024) return fail(IMPNN_E_UNSUPPORTED, "model_head: weights do not fit LDS");
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)model_head_tensors_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  model_head_tensors_kernel<<<(B + kHdSPB - 1) / kHdSPB, 256, lds, s>>>(kind, pc, pa, T, ht, out, B, D, F, Mx, hl);
  return check_launch("model_head_tensors");
}

int launch_model_head_bwd(int kind, const float* pc, const float* pa, const float* T, const float* const* weights,
                          const float* dout, float* dpc, float* dpa, float* const* grads, int B, int D, int F, int Mx,
                          hipStream_t s, const float* l2, const float* y, const float* dloss) {
  if (D > kHdXMax || F > kHdMax || Mx > kHdMax)
    return fail(IMPNN_E_UNSUPPORTED, "model_head_bwd: dims D=%d (<= %d) F=%d Mx=%d (<= %d)", D, kHdXMax, F, Mx, kHdMax);
  HeadTensors ht{};
  if (int rc = head_tensor_table(kind, weights, grads, D, F, Mx, &ht, l2)) return rc;
  HeadLoss hl{};
  if (y) {
    hl.y = y;
    hl.dloss = dloss;
    hl.inv_B = 1.0f / (float)B;
  }
  const size_t lds =
      sizeof(float) * (2 * (((size_t)ht.off[ht.n] + 3) / 4 * 4) + (size_t)kHdSPB * (kHdVecStride + 4 * kHdMax));
  if (lds > 156 * 1024) return fail(IMPNN_E_UNSUPPORTED, "model_head_bwd: weights do not fit LDS");
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)model_head_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int groups = (B + kHdSPB - 1) / kHdSPB;  // bounded grid: every workgroup flushes ~|weights| atomics once
  model_head_bwd_kernel<<<groups < 512 ? groups : 512, 1024, lds, s>>>(kind, pc, pa, T, ht, dout, dpc, dpa, B, D, F, Mx, hl);
  return check_launch("model_head_bwd");
}

int launch_bond_type_matrices_multi(const float* tb, const float* const* W, float* const* out, int n, int Vb, int K,
                                    int D, hipStream_t s) {
  const int DD = D * D;
  for (int p0 = 0; p0 < n; p0 += kBtmMax) {
    BtmBatch bt{};
    bt.n = n - p0 < kBtmMax ? n - p0 : kBtmMax;
    for (int q = 0; q < bt.n; ++q) {
      if (!W[p0 + q] || !out[p0 + q]) return fail(IMPNN_E_BADARG, "bond_type_matrices_multi: null tensor %d", p0 + q);
      bt.W[q] = W[p0 + q];
      bt.out[q] = out[p0 + q];
    }
    if (K <= kBtmSmallK && Vb * K <= kBtmTbMax && DD >= 4096)  // (small matrices: too few workgroups this way)
      bond_type_matrices_multi_smallk_kernel<<<dim3((DD + kBlock - 1) / kBlock, bt.n), kBlock, 0, s>>>(tb, bt, Vb, K, DD);
    else
      bond_type_matrices_multi_kernel<<<dim3((DD + kBlock - 1) / kBlock, Vb, bt.n), kBlock, 0, s>>>(tb, bt, Vb, K, DD);
    if (int rc = check_launch("bond_type_matrices_multi")) return rc;
  }
  return IMPNN_OK;
}

int64_t bond_type_matrices_multi_bwd_workspace(int n, int Vb, int K, int D) {
  const int nb = n < kBtmMax ? n : kBtmMax;
  return (int64_t)nb * ((int64_t)D * D / kBtC + 1) * Vb * K;
}

int launch_bond_type_matrices_multi_bwd(const float* tb, const float* const* W, const float* const* dA,
                                        float* const* dW, float* dtb, int n, int Vb, int K, int D, int accumulate,
                                        hipStream_t s, float* workspace) {
  const int DD = D * D;
  // with a workspace: the bond-table gradient on the matrix cores (partials per wave, summed in a fixed order)
  const bool mfma_t = workspace && K <= 16 && Vb <= 128 && DD % kBtC == 0 && DD >= 4096 &&  // (small matrices: one launch less)
                      (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0;
  for (int p0 = 0; p0 < n; p0 += kBtmMax) {
    BtmBatch bt{};
    bt.n = n - p0 < kBtmMax ? n - p0 : kBtmMax;
    for (int q = 0; q < bt.n; ++q) {
      if (!W[p0 + q] || !dA[p0 + q] || !dW[p0 + q])
        return fail(IMPNN_E_BADARG, "bond_type_matrices_multi_bwd: null tensor %d", p0 + q);
      bt.W[q] = W[p0 + q];
      bt.dA[q] = dA[p0 + q];
      bt.out[q] = dW[p0 + q];
    }
    bond_type_matrices_multi_bwd_w_kernel<<<dim3((DD + kBlock - 1) / kBlock, K, bt.n), kBlock, 0, s>>>(tb, bt, Vb, K, DD,
                                                                                                     accumulate);
    if (int rc = check_launch("bond_type_matrices_multi_bwd_w")) return rc;
    bool al = mfma_t;
    for (int q = 0; q < bt.n; ++q)
      al = al && ((reinterpret_cast<uintptr_t>(bt.W[q]) | reinterpret_cast<uintptr_t>(bt.dA[q])) & 15u) == 0;
    if (al) {
      const int nw = bt.n * (DD / kBtC), nwg = (nw + 3) / 4, acc_t = (accumulate || p0 > 0) ? 1 : 0;
      switch ((Vb + 15) / 16) {
#define BTM_T(VT_)                                                                                                  \
        case VT_:                                                                                                     \
          bond_type_matrices_multi_bwd_t_mfma_kernel<VT_><<<nwg, 256, 0, s>>>(bt, workspace, Vb, K, DD, nw);          \
          break;
        BTM_T(1) BTM_T(2) BTM_T(3) BTM_T(4) BTM_T(5) BTM_T(6) BTM_T(7) BTM_T(8)
#undef BTM_T
      }
      if (int rc = check_launch("bond_type_matrices_multi_bwd_t_mfma")) return rc;
      bond_type_matrices_t_sum_kernel<<<(Vb * K * 64 + kBlock - 1) / kBlock, kBlock, 0, s>>>(workspace, dtb, Vb * K, nw,
                                                                                           acc_t);
      if (int rc = check_launch("bond_type_matrices_t_sum")) return rc;
      continue;
    }
    const int64_t waves = (int64_t)Vb * K;
    // (tried for bond_dim <= 8: one workgroup per bond type with the K partial dot products per thread, dA read once
    //  instead of once per k - 71 workgroups are far too few: 116 us -> 1.3 ms at atom_dim 128)
    bond_type_matrices_multi_bwd_t_kernel<<<(int)((waves * 64 + kBlock - 1) / kBlock), kBlock, 0, s>>>(
        bt, dtb, Vb, K, DD, (accumulate || p0 > 0) ? 1 : 0);
    if (int rc = check_launch("bond_type_matrices_multi_bwd_t")) return rc;
  }
  return IMPNN_OK;
}

}  // namespace impnn
