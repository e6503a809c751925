// encode() for wide atom states (atom_dim 64 / 128: BASELINE config 5, train_viscosity.py:166-190 with atom_dim=128,
// num_steps=6) behind the encoder entries of include/impnn.h, mode IMPNN_ENCODER_F32_TYPED.
//
// At D = 128 a row of node state is 512 bytes and a GatedUpdate weight set 394 KB: nothing of the D = 32 encoder's
// "everything of a chunk in LDS" scheme carries over, and the work is GEMM-shaped and compute-bound in exact f32
// (12 D^2 flop per kept row and step for the update, 2 D^2 per valid edge for the message; 300 GFLOP per 4096-pair
// forward against ~0.3 GB of compulsory traffic per step).  So the path is a short sequence of launches per call, BOTH
// ions in every launch, all of it on a compact row space:
//
//   plan (graph only, once per batch)
//     wide_count      one wave per molecule: kept rows r_b (rows that can send, receive or be pooled - the same rule as
//                     the D = 32 encoders, encoder_plan.hip) and the histogram of valid edges by (ion, bond type)
//     wide_scan       one workgroup: compact row base of every molecule (an ion starts at a multiple of 128 rows),
//                     per-type runs of the type-sorted edge list and their tiles
//     wide_place      valid edges into their type's run (source row per sorted position) and, per kept row, the
//                     positions of its in-edges IN EDGE-SLOT ORDER (the reference's sequential scatter_nd order,
//                     models/layers.py:74-82)
//   run
//     wide_embed      h[row] = atom_table[atom id]                               (a1)
//     S x  wide_message   m[p] = A[type_p] h[src_p]: one GEMM per type run, 64-edge tiles, the type's matrix resident
//                         in LDS, next tile's rows in flight under the MFMAs      (a2 + a4, models/layers.py:100-117)
//                         (mode f32x3 at D = 128: wide_message_x3 - bf16x9, matrix operands in registers)
//          wide_reduce    agg[row] = sum of its in-edge messages, slot order      (a5); rows with <= 2 in-edges are
//                         left to the update, which adds up to two messages itself (wide_iota + wide_place)
//          wide_update    GatedUpdate on 64-row tiles (two workgroups per CU), [h|agg] and the gate kernels
//                         streamed through LDS in 16-deep k slices, h updated in place (a7, models/layers.py:142-156)
//                         (mode f32x3: wide_update_x3 on 64-row tiles, wide_update_x3b on 128-row tiles once a batch
//                         fills the chip - bf16x9, kernel slices straight from global into LDS)
//     wide_pool       pooled[b] = sum_n h[b,n] [atom_ids[b,n] > 0], ascending n   (a8)
//
// Every product is an exact f32 product on v_mfma_f32_16x16x4_f32; every sum has a fixed order that does not depend
// on where a molecule sits in the batch, so results are bitwise reproducible and independent of sharding.
#include <atomic>
#include <climits>
#include <cstdlib>

#include "common.h"

namespace impnn {
namespace wide {

constexpr int kRT = 64;        // rows of a GatedUpdate tile (the exact-f32 kernel; mode 3's large-batch kernel: 128)
constexpr int kRowAlign = 128; // an ion's rows start at a multiple of it (a tile never holds rows of two ions)
constexpr int kMaxN = 256;     // atoms per molecule (LDS tables of wide_place)
constexpr int kMaxE = 1024;    // edge slots per molecule (LDS tables of wide_place; the explicit-hydrogen data sets pad to E = 4 max_bonds = 640)
constexpr int kMaxVb = 512;    // bond vocabulary (types of both ions: one per thread of wide_scan)
constexpr int kMolPerWg = 16;  // molecules of a wide_count / wide_place workgroup (4 waves x 4); launches of up to
                               // 1024 molecules take one molecule per wave (Inputs::mpw: latency, not atomics, bounds them)

// meta words (device): rows of ion g, first compact row of ion g, valid edges, message tiles
enum { kMetaRows = 0, kMetaBase = 2, kMetaEnd = 4, kMetaValid = 5, kMetaTiles = 6, kMetaWords = 16 };

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t ldv4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
__device__ __forceinline__ void stv4(float* p, f32x4_t v) { *reinterpret_cast<f32x4_t*>(p) = v; }
__device__ __forceinline__ f32x4_t mfma_f32(float a, float b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float fsig(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}
__device__ __forceinline__ float ftanh(float x) {
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * x)), 1.0f);
}
// The elementwise arithmetic of the GatedUpdate, shared by every update kernel of this file with the fusion of multiply
// and add spelled out: a batch and its shards may run different kernels (tile sizes) and must agree bit for bit, which
// they do not if the compiler is left to contract `a * b + c` one way in one kernel and another way in the next.
__device__ __forceinline__ float gu_rh(float r_pre, float h) {  // sigmoid(r) * h (models/layers.py:147-148)
#pragma clang fp contract(off)
  return fsig(r_pre) * h;
}
__device__ __forceinline__ float gu_blend(float z, float h, float t_pre) {  // (1 - z) h + z tanh(t) (models/layers.py:150)
#pragma clang fp contract(off)
  const float keep = (1.0f - z) * h;
  return fmaf(z, ftanh(t_pre), keep);
}
__device__ __forceinline__ float gu_inv_std(float sq_dev_sum, float inv_d, float eps) {  // 1 / sqrt(var + eps): v_rsq_f32, 1 ulp
  return __builtin_amdgcn_rsqf(fmaf(sq_dev_sum, inv_d, eps));
}
__device__ __forceinline__ float gu_out(float x, float mean, float inv, float gamma, float beta, float h) {  // LayerNorm + residual
#pragma clang fp contract(off)
  const float n = (x - mean) * inv;
  return fmaf(n, gamma, beta) + h;
}
__device__ __forceinline__ float row16_sum_f(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));
  return v;
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4_t mfma_bf16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// two f32 -> their three packed bf16 pairs (low half <- x, high half <- y)
__device__ __forceinline__ void split_pair_w(float x, float y, unsigned& w0, unsigned& w1, unsigned& w2) {
  const unsigned xb = __builtin_bit_cast(unsigned, x), yb = __builtin_bit_cast(unsigned, y);
  const float x1 = x - __builtin_bit_cast(float, xb & 0xffff0000u), y1 = y - __builtin_bit_cast(float, yb & 0xffff0000u);
  const unsigned x1b = __builtin_bit_cast(unsigned, x1), y1b = __builtin_bit_cast(unsigned, y1);
  const float x2 = x1 - __builtin_bit_cast(float, x1b & 0xffff0000u), y2 = y1 - __builtin_bit_cast(float, y1b & 0xffff0000u);
  w0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
  w1 = __builtin_amdgcn_perm(y1b, x1b, 0x07060302u);
  w2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y2), __builtin_bit_cast(unsigned, x2), 0x07060302u);
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;  // destination of global_load_lds (a wave-uniform LDS address)

constexpr size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// In-kernel stamps (diagnostics builds only; tools/wide_stamps.py): thread 0 of a workgroup writes s_memtime into
// word `slot` of its 8-word record.  In the product build the macro is empty and no stamp executes.
#ifdef IMPNN_DIAG_WIDE_STAMPS
#define WIDE_STAMP(buf, slot)                                                                     \
  do {                                                                                            \
    if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define WIDE_STAMP_REAL(buf, slot)                                                                    \
  do {                                                                                                \
    if ((buf) && threadIdx.x == 0) (buf)[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define WIDE_STAMP(buf, slot) do { } while (0)
#define WIDE_STAMP_REAL(buf, slot) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------------------
struct Ws {
  size_t meta, kept, rowbase, cnt, tstart, tilebase, cursor, srcrow, rowinfo, csr, aggc2, h, agg, m, img, total;
  int64_t rmax, vmax;
  int nT;
};

inline int tile_edges(int D) { return D >= 128 ? 64 : 128; }

// floats of one step of a prepared image: Vb type matrices (D x D, row-major [i][j]) | [Wz|Wr] slices | Wh slices |
// bz br bh gamma beta
// (mode 3: the gate kernels as three bf16 planes: 9 D^2 floats' worth of bytes instead of 6 D^2, and behind the vectors
//  the type matrices once more as three bf16 planes in MFMA operand order: 1.5 Vb D^2 floats' worth - mat_planes_off)
inline size_t mat_planes_off(int D, int Vb) { return (size_t)Vb * D * D + 9 * (size_t)D * D + 5 * (size_t)D; }
inline size_t step_floats(int D, int Vb, bool x3 = false) {
  return x3 ? mat_planes_off(D, Vb) + (size_t)Vb * D * D / 2 * 3 : (size_t)Vb * D * D + 6 * (size_t)D * D + 5 * (size_t)D;
}
inline size_t prepared_bytes(int D, int S, int Vb, bool x3 = false) {
  return align_up((size_t)(S > 0 ? S : 1) * step_floats(D, Vb, x3) * 4, 256);
}

inline Ws ws_layout(int n_ions, int B, int N, int E, int D, int S, int Vb, bool x3 = false) {
  Ws w{};
  const int64_t mols = (int64_t)n_ions * B;
  w.nT = n_ions * Vb;
  w.rmax = (mols * N + (int64_t)n_ions * kRowAlign + kRowAlign - 1) / kRowAlign * kRowAlign;  // whole tiles
  w.vmax = mols * E + (int64_t)(w.nT + 2) * tile_edges(D);  // a type's run is padded to whole tiles
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += align_up(bytes, 256);
    return at;
  };
  w.meta = take(kMetaWords * 4);
  w.cnt = take((size_t)(w.nT + 1) * 4);  // (meta and cnt are zeroed together)
  w.kept = take((size_t)mols * 4);
  w.rowbase = take((size_t)mols * 4);
  w.tstart = take((size_t)(w.nT + 1) * 4);
  w.tilebase = take((size_t)(w.nT + 1) * 4);
  w.cursor = take((size_t)(w.nT + 1) * 4);
  w.srcrow = take((size_t)w.vmax * 4);
  w.rowinfo = take((size_t)w.rmax * 8);
  w.csr = take((size_t)w.vmax * 4);
  w.aggc2 = take((size_t)w.rmax * 8);  // two sources per row (wide_iota_kernel)
  w.h = take((size_t)w.rmax * D * 4);
  w.agg = take((size_t)(w.rmax + 1) * D * 4);  // + a row of zeros at index rmax
  w.m = take((size_t)w.vmax * D * 4);
  w.img = take((size_t)n_ions * prepared_bytes(D, S, Vb, x3));
  w.total = o;
  return w;
}

struct Inputs {
  const int32_t* atom_ids[2];
  const int32_t* bond_ids[2];
  const int32_t* conn[2];
  int n_ions, B, N, E, Va, Vb;
  int mpw;  // molecules per wave of wide_count / wide_place: kMolPerWg / 4, or 1 for small launches
};

__device__ __forceinline__ int valid_type(const int32_t* conn, const int32_t* bond_ids, int64_t be, int N, int Vb,
                                          int& src, int& tgt) {
  src = conn[be * 2];
  tgt = conn[be * 2 + 1];
  const int ty = bond_ids[be];
  return (src > 0 && tgt > 0 && src < N && tgt < N && (unsigned)ty < (unsigned)Vb) ? ty : -1;
}

// ------------------------------------------------------------------------------------------------------------
// plan kernels
// ------------------------------------------------------------------------------------------------------------
// Where the GatedUpdate finds a row's aggregated messages: TWO sources per row, c2a[row] + c2b[row] (added where the
// update parks the slice, first slot first: the Reduce's order).  A source is a row of `agg` (code >= 0) or ~position of
// a message in `m`.  A row with one in-edge names that message and the row of zeros at index n of `agg`; a row with two
// names both messages; a row with none the zeros twice; every other row itself (written by wide_reduce) and the zeros.
// wide_reduce then only sums rows with three in-edges and more - the leaves of a tree, every hydrogen of an
// explicit-hydrogen molecule, every chain atom cost neither a read nor a write of an aggregated copy.
// Defaults here, rows with <= 2 in-edges from wide_place.
__global__ void wide_iota_kernel(int32_t* __restrict__ c2a, int32_t* __restrict__ c2b, float* __restrict__ agg, int n,
                                 int D, int32_t* __restrict__ zero, int nz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    c2a[i] = i;
    c2b[i] = n;
  }
  if (i < D) agg[(int64_t)n * D + i] = 0.f;
  if (i < nz) zero[i] = 0;  // meta and the type counters (what wide_zero_kernel did in a launch of its own)
}


// One wave per molecule (4 in turn): kept rows, and the workgroup's histogram of valid edges by (ion, type) - counted
// in LDS, one global atomic per type the workgroup saw.
__global__ __launch_bounds__(256) void wide_count_kernel(Inputs in, int32_t* __restrict__ kept,
                                                         int32_t* __restrict__ cnt) {
  __shared__ int32_t lh[2 * kMaxVb];
  const int nT = in.n_ions * in.Vb;
  for (int t = threadIdx.x; t < nT; t += 256) lh[t] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mols = in.n_ions * in.B;
  for (int i = 0; i < in.mpw; ++i) {
    const int mol = (blockIdx.x * 4 + wave) * in.mpw + i;
    if (mol >= mols) break;
    const int g = mol >= in.B ? 1 : 0, b = mol - g * in.B;
    const int32_t* ids = in.atom_ids[g] + (int64_t)b * in.N;
    int r = 0;
    for (int n = lane; n < in.N; n += 64)
      if (ids[n] > 0) r = n + 1;
    for (int e = lane; e < in.E; e += 64) {
      int sv, tv;
      const int ty = valid_type(in.conn[g], in.bond_ids[g], (int64_t)b * in.E + e, in.N, in.Vb, sv, tv);
      if (ty >= 0) {
        const int mx = (sv > tv ? sv : tv) + 1;
        r = r > mx ? r : mx;
        atomicAdd(&lh[g * in.Vb + ty], 1);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int t = __shfl_xor(r, o);
      r = r > t ? r : t;
    }
    if (lane == 0) kept[mol] = r;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nT; t += 256)
    if (lh[t]) atomicAdd(&cnt[t], lh[t]);
}

// One workgroup of 1024 threads: (a) exclusive scan of the kept rows per ion (an ion's first row is a multiple of kRowAlign),
// (b) per-type runs and tiles.
__global__ __launch_bounds__(1024) void wide_scan_kernel(const int32_t* __restrict__ kept, int32_t* __restrict__ rowbase,
                                                         const int32_t* __restrict__ cnt, int32_t* __restrict__ tstart,
                                                         int32_t* __restrict__ cursor, int32_t* __restrict__ tilebase,
                                                         int32_t* __restrict__ srcrow, int32_t* __restrict__ meta,
                                                         int n_ions, int B, int nT, int te) {
  __shared__ int32_t wsum[16], wsum2[16];
  __shared__ int32_t carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int g = 0; g < n_ions; ++g) {
    const int per = (B + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < B ? lo + per : B;
    int s = 0;
    for (int b = lo; b < hi; ++b) s += kept[g * B + b];
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    int run = base + off + inc - s;
    for (int b = lo; b < hi; ++b) {
      rowbase[g * B + b] = run;
      run += kept[g * B + b];
    }
    if (tid == 1023) carry = off + inc;
    __syncthreads();
    const int rows = carry;
    if (tid == 0) {
      meta[kMetaRows + g] = rows;
      meta[kMetaBase + g] = base;
      meta[kMetaEnd] = base + rows;
    }
    base = (base + rows + kRowAlign - 1) / kRowAlign * kRowAlign;
    __syncthreads();
  }
  {  // types: nT <= 1024, one per thread
    const int c = tid < nT ? cnt[tid] : 0;
    const int tl = (c + te - 1) / te;
    int ic = c, it = tl;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int uc = __shfl_up(ic, o), ut = __shfl_up(it, o);
      if (lane >= o) {
        ic += uc;
        it += ut;
      }
    }
    if (lane == 63) {
      wsum[wave] = ic;
      wsum2[wave] = it;
    }
    __syncthreads();
    int oc = 0, ot = 0;
    for (int w = 0; w < wave; ++w) {
      oc += wsum[w];
      ot += wsum2[w];
    }
    ic += oc;
    it += ot;
    if (tid < nT) {  // a type's run starts at a whole tile: position = te x tile
      tstart[tid] = (it - tl) * te;
      cursor[tid] = (it - tl) * te;
      tilebase[tid] = it - tl;
      // the padding positions behind the run read row 0 in wide_message (any row inside the workspace would do)
      for (int pz = (it - tl) * te + c; pz < it * te; ++pz) srcrow[pz] = 0;
    }
    if (tid == 1023) {  // threads past nT carry zeros: the last inclusive values are the totals
      tstart[nT] = it * te;
      tilebase[nT] = it;
      meta[kMetaValid] = ic;
      meta[kMetaTiles] = it;
    }
  }
}

// Places the valid edges of kMolPerWg molecules: a range per (ion, type) is reserved with one global atomic per
// workgroup, positions inside it come from LDS atomics (where an edge lands inside its run does not matter: nothing
// is summed across sorted positions).  Then, per molecule, the in-edge lists of its kept rows in edge-slot order:
// rowinfo[row] = (first entry, in-degree), entries at the molecule's own E-slot segment of `csr`.
__global__ __launch_bounds__(256) void wide_place_kernel(Inputs in, const int32_t* __restrict__ kept,
                                                         const int32_t* __restrict__ rowbase,
                                                         int32_t* __restrict__ cursor, int32_t* __restrict__ srcrow,
                                                         int2* __restrict__ rowinfo, int32_t* __restrict__ csr,
                                                         int32_t* __restrict__ c2a,
                                                         int32_t* __restrict__ c2b, int zero_row, int direct_ok) {
  __shared__ int32_t lh[2 * kMaxVb];
  __shared__ int16_t tg_s[4][kMaxE];   // target row of a slot, -1 = not a valid edge
  __shared__ int32_t pos_s[4][kMaxE];  // its sorted position
  __shared__ int32_t deg_s[4][kMaxN], off_s[4][kMaxN], cnt_s[4][kMaxN];
  const int nT = in.n_ions * in.Vb;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mols = in.n_ions * in.B;
  for (int t = threadIdx.x; t < nT; t += 256) lh[t] = 0;
  __syncthreads();
  for (int i = 0; i < in.mpw; ++i) {
    const int mol = (blockIdx.x * 4 + wave) * in.mpw + i;
    if (mol >= mols) break;
    const int g = mol >= in.B ? 1 : 0, b = mol - g * in.B;
    for (int e = lane; e < in.E; e += 64) {
      int sv, tv;
      const int ty = valid_type(in.conn[g], in.bond_ids[g], (int64_t)b * in.E + e, in.N, in.Vb, sv, tv);
      if (ty >= 0) atomicAdd(&lh[g * in.Vb + ty], 1);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nT; t += 256) {
    const int c = lh[t];
    lh[t] = c ? atomicAdd(&cursor[t], c) : 0;
  }
  __syncthreads();
  for (int i = 0; i < in.mpw; ++i) {
    const int mol = (blockIdx.x * 4 + wave) * in.mpw + i;
    if (mol >= mols) break;
    const int g = mol >= in.B ? 1 : 0, b = mol - g * in.B;
    const int r = kept[mol], rb = rowbase[mol];
    for (int n = lane; n < r; n += 64) deg_s[wave][n] = 0;
    for (int e = lane; e < in.E; e += 64) {
      int sv, tv;
      const int ty = valid_type(in.conn[g], in.bond_ids[g], (int64_t)b * in.E + e, in.N, in.Vb, sv, tv);
      int16_t tg = -1;
      if (ty >= 0) {
        const int p = atomicAdd(&lh[g * in.Vb + ty], 1);
        srcrow[p] = rb + sv;
        pos_s[wave][e] = p;
        tg = (int16_t)tv;
        atomicAdd(&deg_s[wave][tv], 1);
      }
      tg_s[wave][e] = tg;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // this wave's LDS writes and atomics have landed
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of the in-degrees over the kept rows (r <= kMaxN = 4 x 64)
    int run = 0;
    for (int n0 = 0; n0 < r; n0 += 64) {
      const int n = n0 + lane;
      const int d = n < r ? deg_s[wave][n] : 0;
      int inc = d;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
      }
      if (n < r) {
        off_s[wave][n] = run + inc - d;
        rowinfo[rb + n] = make_int2((int)((int64_t)mol * in.E) + run + inc - d, d);
        if (d == 0) c2a[rb + n] = zero_row;  // nothing to add
      }
      run += __shfl(inc, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // slot order inside a row's list: rank = earlier valid slots with the same target = those of earlier 64-slot groups
    // (a running count per target in LDS) + the lower lanes of this group that name the same target (63 readlanes).
    // (Walking all earlier slots per slot was E^2 / 64 LDS reads per lane: 390 us per call at the explicit-hydrogen
    //  shape E = 640, a tenth of the whole encode.)
    for (int n = lane; n < r; n += 64) cnt_s[wave][n] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < in.E; e0 += 64) {
      const int e = e0 + lane;
      const int tg = e < in.E ? tg_s[wave][e] : -1;
      int rank = 0;
#pragma unroll
      for (int j = 0; j < 63; ++j) {
        const int tj = __builtin_amdgcn_readlane(tg, j);
        rank += (j < lane && tj == tg) ? 1 : 0;
      }
      if (tg >= 0) {
        rank += cnt_s[wave][tg];
        csr[(int64_t)mol * in.E + off_s[wave][tg] + rank] = pos_s[wave][e];
        const int dg = deg_s[wave][tg];  // (wide_iota_kernel: the sources of rows with one or two in-edges)
        if (dg <= 2 && direct_ok) (rank == 0 ? c2a : c2b)[rb + tg] = ~pos_s[wave][e];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // every lane has read the counts of the earlier groups
      __builtin_amdgcn_wave_barrier();
      if (tg >= 0) atomicAdd(&cnt_s[wave][tg], 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// run kernels
// ------------------------------------------------------------------------------------------------------------
// a1: one wave per molecule (4 in turn), a row per D/4 lanes; an id outside the vocabulary gives a zero row.
__global__ __launch_bounds__(256) void wide_embed_kernel(Inputs in, const int32_t* __restrict__ kept,
                                                         const int32_t* __restrict__ rowbase,
                                                         const float* __restrict__ table, float* __restrict__ h, int D) {
  // one workgroup per molecule, D / 4 threads per row (a wave per molecule and four molecules per wave, as the plan
  // kernels have it, left a batch of 32 pairs with 4 workgroups walking 24 rows each in turn: 23 us)
  const int mol = blockIdx.x;
  const int qd = D >> 2, rpw = 256 / qd;
  const int sub = threadIdx.x / qd, c4 = threadIdx.x - sub * qd;
  const int g = mol >= in.B ? 1 : 0, b = mol - g * in.B;
  const int r = kept[mol], rb = rowbase[mol];
  const int32_t* ids = in.atom_ids[g] + (int64_t)b * in.N;
  for (int n = sub; n < r; n += rpw) {
    const int id = ids[n];
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)id < (unsigned)in.Va) v = ldv4(table + (int64_t)id * D + 4 * c4);
    stv4(h + (int64_t)(rb + n) * D + 4 * c4, v);
  }
}

// a4 over the type-sorted edge list.  A workgroup walks a contiguous range of TE-edge tiles; tiles of one type are
// consecutive, so the type's D x D matrix (64 KB at D = 128) stays in LDS until the type changes.  Output tile =
// (features on M) x (edges on N): lane (a, q) of the accumulator of feature tile T holds features 16T + 4q .. +3 of
// edge a - one 16-byte store per tile.  The next tile's source rows (and, at a type change, the next matrix) are
// requested before the MFMAs of the current tile and stored to the other LDS buffer after them.
struct MsgParams {
  const float* h;
  float* m;
  const float* img[2];      // prepared images; the type matrices of this step start at img[g] + mat_off
  size_t mat_off;
  size_t planes_off;        // mode 3: the same matrices as bf16 planes (wide_mat_planes_kernel), img[g] + planes_off
  const int32_t* srcrow;
  const int32_t* tilebase;
  const int32_t* meta;
  int nT, Vb;
  unsigned long long* stamps;  // diagnostics builds only (IMPNN_DIAG_WIDE_STAMPS)
};

template <int NT, int TE>
__global__ __launch_bounds__(1024) void wide_message_kernel(MsgParams p) {
  constexpr int D = 16 * NT, LD = D + 4, QD = D / 4;
  constexpr int EG = TE / 16, FG = 16 / EG, NLW = NT / FG;  // edge tiles, feature groups, feature tiles per wave
  constexpr int kX = TE * QD / 1024, kB = D * QD / 1024;    // 16-byte pieces per thread: a tile of rows, the matrix
  static_assert(kX >= 1 && kB >= 1 && NLW >= 1, "tile shape");
  extern __shared__ __align__(16) float smem[];
  float* Bm = smem;               // D x LD
  float* Xb = Bm + D * LD;        // 2 x TE x LD
  int32_t* tb_s = reinterpret_cast<int32_t*>(Xb + 2 * TE * LD);  // tilebase[0 .. nT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int et = wave % EG, fg = wave / EG;
  const int ntiles = p.meta[kMetaTiles];
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  if (t0 >= t1) return;
  WIDE_STAMP(p.stamps, 0);
  WIDE_STAMP_REAL(p.stamps, 5);
  for (int t = tid; t <= p.nT; t += 1024) tb_s[t] = p.tilebase[t];
  __syncthreads();
  auto mat_of = [&](int t) {
    const int g = t >= p.Vb ? 1 : 0;
    return p.img[g] + p.mat_off + (size_t)(t - g * p.Vb) * D * D;
  };
  // A type's run starts at a multiple of TE sorted positions (wide_scan), so tile t is positions [t TE, (t + 1) TE):
  // positions past the type's last edge are padding - their source row is row 0 (wide_scan), their messages are
  // computed and stored like any other and never read.  No load or store of the loop is conditional, which lets the
  // compiler count outstanding memory operations instead of draining them: source rows are requested TWO tiles ahead
  // (sr2), the rows themselves one tile ahead (xr), the stores of a tile drain under the next tile's MFMAs.
  int sr1[kX], sr2[kX];
  f32x4_t xr[kX], br[kB];
  auto fetch_sr = [&](int tile, int* sr) {
#pragma unroll
    for (int i = 0; i < kX; ++i) sr[i] = p.srcrow[tile * TE + (tid + 1024 * i) / QD];
  };
  auto fetch_x = [&](const int* sr) {
#pragma unroll
    for (int i = 0; i < kX; ++i) xr[i] = ldv4(p.h + (int64_t)sr[i] * D + 4 * ((tid + 1024 * i) % QD));
  };
  auto park_x = [&](float* X) {
#pragma unroll
    for (int i = 0; i < kX; ++i) {
      const int idx = tid + 1024 * i, e = idx / QD, c4 = idx - e * QD;
      stv4(X + e * LD + 4 * c4, xr[i]);
    }
  };
  auto fetch_b = [&](int t) {
    const float* A = mat_of(t);
#pragma unroll
    for (int i = 0; i < kB; ++i) br[i] = ldv4(A + (size_t)(tid + 1024 * i) * 4);
  };
  auto park_b = [&]() {
#pragma unroll
    for (int i = 0; i < kB; ++i) {
      const int idx = tid + 1024 * i, r = idx / QD, c4 = idx - r * QD;
      stv4(Bm + r * LD + 4 * c4, br[i]);
    }
  };
  int ty;
  {  // type of the first tile: largest t with tilebase[t] <= t0 (empty types share a base with their successor)
    int lo = 0, hi = p.nT - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tb_s[mid] <= t0) lo = mid; else hi = mid - 1;
    }
    ty = lo;
  }
  int run_end = tb_s[ty + 1];  // first tile of the next type
  fetch_sr(t0, sr1);
  fetch_b(ty);
  fetch_x(sr1);
  fetch_sr(min(t0 + 1, t1 - 1), sr1);
  park_x(Xb);
  park_b();
  __syncthreads();
  WIDE_STAMP(p.stamps, 1);
  int cur = 0;
  for (int tile = t0; tile < t1; ++tile) {
    // the next tile (the last tile is simply requested again: no branch around the requests)
    const int nxt = min(tile + 1, t1 - 1);
    int ty2 = ty, run_end2 = run_end;
    if (nxt >= run_end) {  // (workgroup-uniform) a new type: step over empty ones
      do {
        ++ty2;
        run_end2 = tb_s[ty2 + 1];
      } while (run_end2 <= nxt);
      fetch_b(ty2);
    }
    fetch_x(sr1);
    fetch_sr(min(tile + 2, t1 - 1), sr2);
    __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the requests below the MFMAs, next to their use)
    {
      const float* X = Xb + cur * TE * LD;
      f32x4_t acc[NLW];
#pragma unroll
      for (int TL = 0; TL < NLW; ++TL) acc[TL] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      const float* xrow = X + (16 * et + a) * LD + 4 * q;
      const float* arow = Bm + (16 * (fg * NLW) + a) * LD + 4 * q;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
#ifdef IMPNN_DIAG_WIDE_NOLDS
        const f32x4_t xv = {1.f + u, 2.f, 3.f, 4.f};
        f32x4_t av[NLW];
#pragma unroll
        for (int TL = 0; TL < NLW; ++TL) av[TL] = f32x4_t{0.5f, 0.25f + TL, 0.125f, 2.f};
#else
        const f32x4_t xv = ldv4(xrow + 16 * u);
        f32x4_t av[NLW];
#pragma unroll
        for (int TL = 0; TL < NLW; ++TL) av[TL] = ldv4(arow + 16 * TL * LD + 16 * u);
#endif
#ifdef IMPNN_DIAG_WIDE_NOMMA
        acc[0] += xv + av[0] + av[NLW - 1];
#else
#pragma unroll
        for (int TL = 0; TL < NLW; ++TL)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[TL] = mfma_f32(av[TL][r], xv[r], acc[TL]);
#endif
      }
      float* dst = p.m + ((int64_t)tile * TE + 16 * et + a) * D + 16 * (fg * NLW) + 4 * q;
#pragma unroll
      for (int TL = 0; TL < NLW; ++TL) stv4(dst + 16 * TL, acc[TL]);
    }
    __builtin_amdgcn_sched_barrier(0);
    park_x(Xb + (cur ^ 1) * TE * LD);
    if (ty2 != ty) {     // (workgroup-uniform)
      __syncthreads();   // every wave is done with the old matrix
      park_b();
    }
    __syncthreads();
    cur ^= 1;
    ty = ty2;
    run_end = run_end2;
#pragma unroll
    for (int i = 0; i < kX; ++i) sr1[i] = sr2[i];
  }
  WIDE_STAMP(p.stamps, 4);
  WIDE_STAMP_REAL(p.stamps, 6);
#ifdef IMPNN_DIAG_WIDE_STAMPS
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)(t1 - t0);
#endif
}

// ------------------------------------------------------------------------------------------------------------
// a2 + a4 in mode IMPNN_ENCODER_F32X3_TYPED: the per-type GEMMs m = A[type] h[src] on the bf16 matrix pipe, every f32
// operand carried exactly as three bf16 terms and all nine cross products accumulated in f32 (as the GatedUpdate of this
// mode).  8 waves: a wave multiplies 32 edges x 32 features (2 x 2 MFMA tiles; 64-edge tiles at D = 128, 128-edge tiles
// at D = 64 - as the plan cuts them).
//   * a wave keeps ITS operands of the type's matrix - 32 feature rows, all k, three planes: 96 VGPRs - in registers
//     for the whole run of the type (a type's run is ~40 tiles; the planes come pre-split and in operand order from
//     the prepared image, wide_mat_planes_kernel), so a tile costs LDS traffic for the rows only;
//   * the rows of the next tile are gathered under the MFMAs, split (three planes of bf16) and parked in the other of
//     two LDS stages between the MFMAs of the second half of the tile: one barrier per tile.
// Tiles, runs and the unconditional requests as in wide_message_kernel.
// ------------------------------------------------------------------------------------------------------------
constexpr int kMsgX3Threads = 512;
constexpr size_t msg_x3_lds_bytes(int D, int TE, int nT) { return 2 * (size_t)3 * (D / 32) * 4 * (TE + 1) * 16 + (size_t)(nT + 1) * 4; }

template <int NT, int TE>
__global__ __launch_bounds__(kMsgX3Threads, 1) void wide_message_x3_kernel(MsgParams p) {
  constexpr int D = 16 * NT, QD = D / 4, KB = D / 32, T = kMsgX3Threads;
  constexpr int UM = 3 * KB * 4 * D;   // 16-byte units of a type's matrix (three planes): [plane][k block][k octet][feature]
  constexpr int XS = TE + 1;           // units between the (k block, k octet) rows of a tile's planes: one unit of padding, so
                                       // that the 16 k octets a wave parks at once fall into different LDS banks
  constexpr int UX = 3 * KB * 4 * XS;  // ... of a tile of rows: [plane][k block][k octet][edge]
  constexpr int kX = TE * QD / T;      // 16-byte pieces of f32 rows per thread
  constexpr int EGN = TE / 32, FGN = 8 / EGN;  // 8 waves = EGN groups of 32 edges x FGN groups of 32 features
  static_assert(EGN * FGN == 8 && NT == 2 * FGN && kX >= 2 && kX % 2 == 0 && KB % 2 == 0, "tile shape");
  extern __shared__ __align__(16) unsigned char smem_b[];
  uint4* const Xb = reinterpret_cast<uint4*>(smem_b);             // 2 x UX units
  int32_t* const tb_s = reinterpret_cast<int32_t*>(Xb + 2 * UX);  // tilebase[0 .. nT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = lane & 15, q = lane >> 4;
  const int eg = wave % EGN, fg = wave / EGN;  // 32 edges x 32 features
  const int ntiles = p.meta[kMetaTiles];
  const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = blockIdx.x * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  if (t0 >= t1) return;
  WIDE_STAMP(p.stamps, 0);
  WIDE_STAMP_REAL(p.stamps, 5);
  bf16x8_t am[KB][2][3];  // the wave's matrix operands: [k block][feature tile][plane]
  auto load_mat = [&](int t) {
    const int g = t >= p.Vb ? 1 : 0;
    const uint4* src = reinterpret_cast<const uint4*>(p.img[g] + p.planes_off) + (size_t)(t - g * p.Vb) * UM;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int TL = 0; TL < 2; ++TL)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          am[kb][TL][pl] = __builtin_bit_cast(bf16x8_t, src[((pl * KB + kb) * 4 + q) * D + 16 * (fg * 2 + TL) + a]);
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // rows are requested TWO tiles ahead (two sets of staging registers, alternating), their source-row indices three
  int srn[kX];
  f32x4_t xa[kX], xb[kX];
  auto fetch_sr = [&](int tile) {
#pragma unroll
    for (int i = 0; i < kX; ++i) srn[i] = p.srcrow[tile * TE + (tid + T * i) / QD];
  };
  auto fetch_x = [&](f32x4_t (&xr)[kX]) {  // the rows srn names
#pragma unroll
    for (int i = 0; i < kX; ++i) xr[i] = ldv4(p.h + (int64_t)srn[i] * D + 4 * ((tid + T * i) % QD));
  };
  auto park_piece = [&](uint4* X, const f32x4_t (&xr)[kX], int i) {  // 4 values of a row -> three planes of 4 bf16:
    uint2* s2 = reinterpret_cast<uint2*>(X);                          // unit (plane, k block, k octet, edge), 8-byte half
    const int idx = tid + T * i, e = idx / QD, c4 = idx - e * QD;
    const int un = ((c4 >> 3) * 4 + ((c4 >> 1) & 3)) * XS + e, half = c4 & 1;
    unsigned w0[2], w1[2], w2[2];
    split_pair_w(xr[i][0], xr[i][1], w0[0], w1[0], w2[0]);
    split_pair_w(xr[i][2], xr[i][3], w0[1], w1[1], w2[1]);
    s2[(0 * KB * 4 * XS + un) * 2 + half] = make_uint2(w0[0], w0[1]);
    s2[(1 * KB * 4 * XS + un) * 2 + half] = make_uint2(w1[0], w1[1]);
    s2[(2 * KB * 4 * XS + un) * 2 + half] = make_uint2(w2[0], w2[1]);
  };
  // (the first source rows are requested together with the run table: one round trip to memory instead of two)
  const int tl = t1 - 1;  // (requests past the share's last tile name it again: no branch around them)
  fetch_sr(t0);
  for (int t = tid; t <= p.nT; t += T) tb_s[t] = p.tilebase[t];
  __syncthreads();
  int ty;
  {  // type of the first tile: largest t with tilebase[t] <= t0 (empty types share a base with their successor)
    int lo = 0, hi = p.nT - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tb_s[mid] <= t0) lo = mid; else hi = mid - 1;
    }
    ty = lo;
  }
  int run_end = tb_s[ty + 1];  // first tile of the next type
  fetch_x(xa);                    // rows of t0
  load_mat(ty);
  fetch_sr(min(t0 + 1, tl));
#pragma unroll
  for (int i = 0; i < kX; ++i) park_piece(Xb, xa, i);
  fetch_x(xa);                    // rows of t0 + 1: parked inside tile t0
  fetch_sr(min(t0 + 2, tl));      // (srn = the rows of t0 + 2: requested at the top of tile t0)
  lds_barrier();
  WIDE_STAMP(p.stamps, 1);
  constexpr int kPa[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, kPb[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};  // (matrix plane, row plane), smallest first
  int cur = 0;
  // tile `tile` out of stage cur; the rows of tile + 1 (in xpark since the tile before) go to the other stage, the rows
  // of tile + 2 are requested into xfetch
  auto do_tile = [&](int tile, f32x4_t (&xpark)[kX], f32x4_t (&xfetch)[kX]) {
    const int nxt = min(tile + 1, tl);
    int ty2 = ty, run_end2 = run_end;
    if (nxt >= run_end) {  // (workgroup-uniform) a new type: step over empty ones
      do {
        ++ty2;
        run_end2 = tb_s[ty2 + 1];
      } while (run_end2 <= nxt);
    }
    fetch_x(xfetch);
    fetch_sr(min(tile + 3, tl));
    __builtin_amdgcn_sched_barrier(0);
    {
      const uint4* X = Xb + cur * UX;
      f32x4_t acc[2][2];  // [feature tile][edge tile]
#pragma unroll
      for (int TL = 0; TL < 2; ++TL)
#pragma unroll
        for (int et = 0; et < 2; ++et) acc[TL][et] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      bf16x8_t xe[2][2][3];  // [buffer][edge tile][plane]
#pragma unroll
      for (int pl = 2; pl >= 0; --pl)  // (in the order the products take them)
#pragma unroll
        for (int et = 0; et < 2; ++et)
          xe[0][et][pl] = __builtin_bit_cast(bf16x8_t, X[((pl * KB + 0) * 4 + q) * XS + 16 * (eg * 2 + et) + a]);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        if (kb + 1 < KB) {
#pragma unroll
          for (int pl = 2; pl >= 0; --pl)
#pragma unroll
            for (int et = 0; et < 2; ++et)
              xe[(kb + 1) & 1][et][pl] = __builtin_bit_cast(bf16x8_t, X[((pl * KB + kb + 1) * 4 + q) * XS + 16 * (eg * 2 + et) + a]);
        }
        // the next tile's rows (requested at the top of this one) are split and parked between the MFMAs of the last
        // two k blocks: half of the thread's pieces each
        if (kb >= KB - 2) {
#pragma unroll
          for (int i = (kb - (KB - 2)) * (kX / 2); i < (kb - (KB - 2) + 1) * (kX / 2); ++i) park_piece(Xb + (cur ^ 1) * UX, xpark, i);
        }
#pragma unroll
        for (int pr = 0; pr < 9; ++pr)
#pragma unroll
          for (int TL = 0; TL < 2; ++TL)
#pragma unroll
            for (int et = 0; et < 2; ++et)
              acc[TL][et] = mfma_bf16(am[kb][TL][kPa[pr]], xe[kb & 1][et][kPb[pr]], acc[TL][et]);
        if (kb >= KB - 2) {
#pragma unroll
          for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int et = 0; et < 2; ++et) {
        float* dst = p.m + ((int64_t)tile * TE + 16 * (eg * 2 + et) + a) * D + 16 * (fg * 2) + 4 * q;
#pragma unroll
        for (int TL = 0; TL < 2; ++TL) stv4(dst + 16 * TL, acc[TL][et]);
      }
    }
    if (ty2 != ty) load_mat(ty2);  // (workgroup-uniform; its latency is exposed once per type run)
    lds_barrier();  // the other stage is complete, this one free: the stores above stay in flight
    cur ^= 1;
    ty = ty2;
    run_end = run_end2;
  };
  for (int tile = t0; tile < t1; tile += 2) {
    do_tile(tile, xa, xb);
    if (tile + 1 < t1) do_tile(tile + 1, xb, xa);
  }
  WIDE_STAMP(p.stamps, 4);
  WIDE_STAMP_REAL(p.stamps, 6);
#ifdef IMPNN_DIAG_WIDE_STAMPS
  if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)(t1 - t0);
#endif
}

// a5 on the compact rows: D/4 lanes per row, the in-edge messages added in edge-slot order with 4 rows in flight.
__global__ __launch_bounds__(256) void wide_reduce_kernel(const float* __restrict__ m, const int2* __restrict__ rowinfo,
                                                          const int32_t* __restrict__ csr, float* __restrict__ agg,
                                                          const int32_t* __restrict__ meta, int n_ions, int D, int skip_upto) {
  const int qd = D >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t / qd;
  const int c4 = (int)(t - row * qd);
  if (row >= meta[kMetaEnd]) return;
  if (n_ions > 1 && row >= meta[kMetaRows] && row < meta[kMetaBase + 1]) return;  // the gap in front of ion 1
  const int2 ri = rowinfo[row];
  if (ri.y <= skip_upto) return;  // the update adds up to two messages itself, and zeros for a row without in-edges (wide_iota_kernel)
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (; i + 4 <= ri.y; i += 4) {
    const int p0 = csr[ri.x + i], p1 = csr[ri.x + i + 1], p2 = csr[ri.x + i + 2], p3 = csr[ri.x + i + 3];
    const f32x4_t v0 = ldv4(m + (int64_t)p0 * D + 4 * c4), v1 = ldv4(m + (int64_t)p1 * D + 4 * c4);
    const f32x4_t v2 = ldv4(m + (int64_t)p2 * D + 4 * c4), v3 = ldv4(m + (int64_t)p3 * D + 4 * c4);
    acc += v0;
    acc += v1;
    acc += v2;
    acc += v3;
  }
  for (; i < ri.y; ++i) acc += ldv4(m + (int64_t)csr[ri.x + i] * D + 4 * c4);
  stv4(agg + row * D + 4 * c4, acc);
}

// a7 on kRT-row tiles of the compact row space, h updated in place.  8 waves per tile, TWO workgroups resident per CU
// (77 KB of LDS and 128 VGPRs each): exact-f32 MFMA and the vector ALU share one issue port, so a tile's barrier
// bubbles, its prologue and its LayerNorm epilogue are only ever hidden by ANOTHER tile's MFMAs.
//   phase 1   [z|r] pre-activations = [h|agg] (R x 2D) x [Wz|Wr] (2D x 2D): 2 NT slices of 16 k; a slice of the rows
//             (4 KB, MFMA operand order [k quad][row][4]) and of the kernels (16 KB at D = 128, the image's own order)
//             goes global -> registers (two slices ahead) -> one of two LDS stages; one barrier per slice, 32 MFMAs
//             per wave between barriers (wave = 32 rows x NL feature tiles of z and of r).
//   phase 2   candidate = [r*h|agg] x Wh, 2 NT slices again: r*h comes from LDS (written once after phase 1), agg and
//             Wh through the stages.
//   epilogue  blend, LayerNorm (row sums across the four feature groups through LDS), residual.
// h of the accumulator positions is read once into registers (for r*h, the blend and the residual).
// float offset from `agg` of a source of aggregated messages (wide_iota_kernel): a row of agg, or ~position of a message
__device__ __forceinline__ int agg_off(int code, int m_off, int D) { return code >= 0 ? code * D : m_off + (~code) * D; }

struct GuParams {
  float* h;
  const float* agg;
  const int32_t* c2a;        // two sources of aggregated messages per row (wide_iota_kernel)
  const int32_t* c2b;
  int m_off;                 // floats from agg to m (both in one workspace; the launch checks the range)
  const float* img[2];  // the step's GatedUpdate image starts at img[g] + gu_off
  size_t gu_off;
  const int32_t* meta;
  float eps;
  int n_ions;
  int tile_rows;  // rows a workgroup updates: kRT, or 16 for launches too small to fill the chip with kRT-row tiles
  int cus, tiles_max;  // wide_update_x3b_kernel: CUs of the device, 128-row tiles of the row space
  unsigned long long* stamps;  // diagnostics builds only (IMPNN_DIAG_WIDE_STAMPS)
};

constexpr int kGuThreads = 512;
constexpr size_t gu_lds_floats(int D) {
  // two stages of (row slice + [Wz|Wr] slice) | r*h | LayerNorm partials
  return 2 * (size_t)(4 * kRT * 4 + 4 * 2 * D * 4) + (size_t)kRT * (D + 4) + 8 * kRT;
}

template <int NT>
__global__ __launch_bounds__(kGuThreads, 4) void wide_update_kernel(GuParams p) {
  constexpr int D = 16 * NT, R = kRT, LDR = D + 4;
  constexpr int RG = R / 32, FG = (kGuThreads / 64) / RG, NL = NT / FG;
  constexpr int A1 = 4 * R * 4;       // floats of a 16-k slice of the rows
  constexpr int B1 = 4 * 2 * D * 4;   // ... of [Wz|Wr]
  constexpr int B2 = 4 * D * 4;       // ... of Wh
  constexpr int ST = A1 + B1;         // stage floats
  constexpr int kQ1 = (B1 / 4 + kGuThreads - 1) / kGuThreads, kQ2 = (B2 / 4 + kGuThreads - 1) / kGuThreads;
  constexpr int kAT = R * 4;          // threads that move a piece of a row slice
  static_assert(NL >= 1 && NT % FG == 0 && kAT <= kGuThreads, "tile shape");
  extern __shared__ __align__(16) float smem[];
  float* stage = smem;                 // 2 x ST
  float* rhs = stage + 2 * ST;         // R x LDR : r * h
  float* part = rhs + R * LDR;         // 2 x FG x R LayerNorm partials
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = lane & 15, q = lane >> 4;
  const int rg = wv % RG, fg = wv / RG;  // row group (32 rows), feature group (NL tiles of z, r and the candidate)
  // A workgroup's LDS tile always spans R rows; with tile_rows < R only its first tile_rows rows are the workgroup's
  // own (the rest is read like any padding and never stored), and the 16-row tiles past them are not multiplied:
  // 768 64-row tiles on 512 slots are two rounds, the second half empty - 1 536 32-row tiles are three short ones.
  const int64_t row0 = (int64_t)blockIdx.x * p.tile_rows;
  const int end = p.meta[kMetaEnd];
  if (row0 >= end) return;
  const int g = (p.n_ions > 1 && row0 >= p.meta[kMetaBase + 1]) ? 1 : 0;
  const int64_t ion_end = p.meta[kMetaBase + g] + p.meta[kMetaRows + g];
  const int64_t row_end = row0 + p.tile_rows < ion_end ? row0 + p.tile_rows : ion_end;  // rows beyond it are not this tile's
  if (row0 >= row_end) return;
  const bool lv[2] = {32 * rg < p.tile_rows, 32 * rg + 16 < p.tile_rows};  // (wave-uniform) this wave's two row tiles
  WIDE_STAMP(p.stamps, 0);
  WIDE_STAMP_REAL(p.stamps, 5);
  const float* img = p.img[g] + p.gu_off;
  const float* P1 = img;
  const float* P2 = img + 4 * D * D;
  const float* bias = img + 6 * D * D;  // bz br bh gamma beta
  // (padding rows of the last tile of an ion lie inside the workspace; whatever they hold stays in their own rows)
  const int a_row = (tid % kAT) >> 2, a_c4 = tid & 3;
  const float* hsrc = p.h + (row0 + a_row) * D + 4 * a_c4;
  // the row's aggregated messages: two sources (wide_iota_kernel), as float offsets from p.agg
  const int goff0 = agg_off(p.c2a[row0 + a_row], p.m_off, D) + 4 * a_c4, goff1 = agg_off(p.c2b[row0 + a_row], p.m_off, D) + 4 * a_c4;
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  struct Pre {
    f32x4_t av, aw, bv[kQ1];  // aw: the second source of a slice of aggregated messages (zeros for a slice of h)
  };
  Pre preA, preB;
  auto fetch1 = [&](int u, Pre& pre) {
#pragma unroll
    for (int i = 0; i < kQ1; ++i)
      if (tid + kGuThreads * i < B1 / 4) pre.bv[i] = ldv4(P1 + (size_t)u * B1 + (tid + kGuThreads * i) * 4);
#ifdef IMPNN_DIAG_WIDE_NOFETCH
    if (tid < kAT) { pre.av = f32x4_t{0.25f, 0.5f, -0.25f, 0.125f}; pre.aw = zero4; }
#else
    if (tid < kAT) {
      if (u < NT) {  // (workgroup-uniform)
        pre.av = ldv4(hsrc + 16 * u);
        pre.aw = zero4;
      } else {
        pre.av = ldv4(p.agg + goff0 + 16 * (u - NT));
        pre.aw = ldv4(p.agg + goff1 + 16 * (u - NT));
      }
    }
#endif
  };
  auto park1 = [&](float* st, const Pre& pre) {
#pragma unroll
    for (int i = 0; i < kQ1; ++i)
      if (tid + kGuThreads * i < B1 / 4) stv4(st + A1 + (tid + kGuThreads * i) * 4, pre.bv[i]);
    if (tid < kAT) stv4(st + (a_c4 * R + a_row) * 4, pre.av + pre.aw);  // (first slot first: the Reduce's order)
  };
  f32x4_t z[2][NL], rr[2][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = 16 * (fg * NL + TL) + a;
    const float b0 = bias[f], b1 = bias[D + f];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      z[rt][TL] = f32x4_t{b0, b0, b0, b0};
      rr[rt][TL] = f32x4_t{b1, b1, b1, b1};
    }
  }
  struct Ops1 {
    f32x4_t av[2], bz[NL], br[NL];
  };
  auto read1 = [&](const float* st, Ops1& o) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) o.av[rt] = ldv4(st + (q * R + 32 * rg + 16 * rt + a) * 4);
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      o.bz[TL] = ldv4(st + A1 + (q * 2 * D + 16 * (fg * NL + TL) + a) * 4);
      o.br[TL] = ldv4(st + A1 + (q * 2 * D + D + 16 * (fg * NL + TL) + a) * 4);
    }
  };
  auto mma1 = [&](const Ops1& o) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      if (lv[rt]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int TL = 0; TL < NL; ++TL) {
            z[rt][TL] = mfma_f32(o.av[rt][r], o.bz[TL][r], z[rt][TL]);
            rr[rt][TL] = mfma_f32(o.av[rt][r], o.br[TL][r], rr[rt][TL]);
          }
      }
  };
  fetch1(0, preA);
  fetch1(1, preB);
  park1(stage, preA);
  __syncthreads();
  WIDE_STAMP(p.stamps, 1);
  // h at this lane's accumulator positions (rows 4q + g of both row tiles, feature a of its NL tiles): requested under
  // the last two slices of phase 1 - held from the start they cost 16 registers the phase does not have
  float hreg[2][NL][4];
  auto load_hreg = [&]() {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int TL = 0; TL < NL; ++TL)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          hreg[rt][TL][gq] = p.h[(row0 + 32 * rg + 16 * rt + 4 * q + gq) * D + 16 * (fg * NL + TL) + a];
  };
  // iteration u: slice u is in stage u & 1, slice u + 1 in registers, slice u + 2 is requested; stage (u + 1) & 1 was
  // last read in iteration u - 1, whose closing barrier every wave has passed
  auto pair1 = [&](int u) {
    Ops1 o;
    if (u + 2 < 2 * NT) fetch1(u + 2, preA);
    read1(stage, o);
    __builtin_amdgcn_sched_barrier(0);
    park1(stage + ST, preB);
    __builtin_amdgcn_sched_barrier(0);
    mma1(o);
    __syncthreads();
    if (u + 3 < 2 * NT) fetch1(u + 3, preB);
    read1(stage + ST, o);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < 2 * NT) park1(stage, preA);
    __builtin_amdgcn_sched_barrier(0);
    mma1(o);
    __syncthreads();
  };
  for (int u = 0; u < 2 * NT - 2; u += 2) pair1(u);
  load_hreg();
  pair1(2 * NT - 2);
  WIDE_STAMP(p.stamps, 2);
  // ---- phase 2
  struct Pre2 {
    f32x4_t av, aw, bv[kQ2];
  };
  Pre2 qA, qB;
  auto fetch2 = [&](int u, Pre2& pre) {
#pragma unroll
    for (int i = 0; i < kQ2; ++i)
      if (tid + kGuThreads * i < B2 / 4) pre.bv[i] = ldv4(P2 + (size_t)u * B2 + (tid + kGuThreads * i) * 4);
#ifdef IMPNN_DIAG_WIDE_NOFETCH
    if (u >= NT && tid < kAT) { pre.av = f32x4_t{0.25f, 0.5f, -0.25f, 0.125f}; pre.aw = zero4; }
#else
    if (u >= NT && tid < kAT) {
      pre.av = ldv4(p.agg + goff0 + 16 * (u - NT));
      pre.aw = ldv4(p.agg + goff1 + 16 * (u - NT));
    }
#endif
  };
  auto park2 = [&](int u, float* st, const Pre2& pre) {
#pragma unroll
    for (int i = 0; i < kQ2; ++i)
      if (tid + kGuThreads * i < B2 / 4) stv4(st + A1 + (tid + kGuThreads * i) * 4, pre.bv[i]);
    if (u >= NT && tid < kAT) stv4(st + (a_c4 * R + a_row) * 4, pre.av + pre.aw);
  };
  fetch2(0, qA);
  fetch2(1, qB);
  // gates; r * h into LDS (phase 2 reads the rows of this wave's row group written by all feature groups)
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        z[rt][TL][gq] = fsig(z[rt][TL][gq]);
        rhs[(32 * rg + 16 * rt + 4 * q + gq) * LDR + 16 * (fg * NL + TL) + a] = gu_rh(rr[rt][TL][gq], hreg[rt][TL][gq]);
      }
  f32x4_t tt[2][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const float b2 = bias[2 * D + 16 * (fg * NL + TL) + a];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) tt[rt][TL] = f32x4_t{b2, b2, b2, b2};
  }
  park2(0, stage, qA);
  __syncthreads();
  struct Ops2 {
    f32x4_t av[2], bv[NL];
  };
  auto read2 = [&](int u, const float* st, Ops2& o) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      o.av[rt] = u < NT ? ldv4(rhs + (32 * rg + 16 * rt + a) * LDR + 16 * u + 4 * q)
                        : ldv4(st + (q * R + 32 * rg + 16 * rt + a) * 4);
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) o.bv[TL] = ldv4(st + A1 + (q * D + 16 * (fg * NL + TL) + a) * 4);
  };
  auto mma2 = [&](const Ops2& o) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      if (lv[rt]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int TL = 0; TL < NL; ++TL) tt[rt][TL] = mfma_f32(o.av[rt][r], o.bv[TL][r], tt[rt][TL]);
      }
  };
  for (int u = 0; u < 2 * NT; u += 2) {
    Ops2 o;
    if (u + 2 < 2 * NT) fetch2(u + 2, qA);
    read2(u, stage, o);
    __builtin_amdgcn_sched_barrier(0);
    park2(u + 1, stage + ST, qB);
    __builtin_amdgcn_sched_barrier(0);
    mma2(o);
    __syncthreads();
    if (u + 3 < 2 * NT) fetch2(u + 3, qB);
    read2(u + 1, stage + ST, o);
    __builtin_amdgcn_sched_barrier(0);
    if (u + 2 < 2 * NT) park2(u + 2, stage, qA);
    __builtin_amdgcn_sched_barrier(0);
    mma2(o);
    __syncthreads();
  }
  WIDE_STAMP(p.stamps, 3);
  // ---- blend, LayerNorm over the D features of a row, residual (models/layers.py:150-156)
  // (row sums over the 16 lanes of a quarter wave, all of the wave's rows step by step: a row's next DPP step is eight
  //  instructions behind its last, no stall between dependent DPP operations)
  auto row16_sum_all = [&](float (&v)[2][4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int iv = __builtin_bit_cast(int, v[rt][gq]);
          const int o = st == 0 ? __builtin_amdgcn_update_dpp(0, iv, 0x121, 0xf, 0xf, true)
                        : st == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0x122, 0xf, 0xf, true)
                        : st == 2 ? __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xf, 0xf, true)
                                  : __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, true);
          v[rt][gq] += __builtin_bit_cast(float, o);
        }
  };
  float sum[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float sacc = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float hv = hreg[rt][TL][gq];
        const float nv = gu_blend(z[rt][TL][gq], hv, tt[rt][TL][gq]);
        tt[rt][TL][gq] = nv;
        sacc += nv;
      }
      sum[rt][gq] = sacc;
    }
  row16_sum_all(sum);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      if (a == 0) part[fg * R + 32 * rg + 16 * rt + 4 * q + gq] = sum[rt][gq];
  __syncthreads();
  float mean[2][4], inv[2][4], var[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = 32 * rg + 16 * rt + 4 * q + gq;
      float ms = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) ms += part[f2 * R + rl];
      mean[rt][gq] = ms * (1.0f / D);
      float vs = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float dv = tt[rt][TL][gq] - mean[rt][gq];
        vs = fmaf(dv, dv, vs);
      }
      var[rt][gq] = vs;
    }
  row16_sum_all(var);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      if (a == 0) part[FG * R + fg * R + 32 * rg + 16 * rt + 4 * q + gq] = var[rt][gq];
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = FG * R + 32 * rg + 16 * rt + 4 * q + gq;
      float vs = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) vs += part[f2 * R + rl];
      inv[rt][gq] = gu_inv_std(vs, 1.0f / D, p.eps);
    }
  {
    float* const out = p.h + (row0 + 32 * rg + 4 * q) * D + 16 * fg * NL + a;
    // a whole tile (tile_rows == R) stores every row: rows past row_end are padding of the row space, which nothing
    // reads as a source, a target or a pooled row; a workgroup that owns only the first rows of its LDS tile checks
    const bool all_rows = p.tile_rows == R;  // (workgroup-uniform)
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const int f = 16 * (fg * NL + TL) + a;
      const float gm = bias[3 * D + f], bt = bias[4 * D + f];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float v = gu_out(tt[rt][TL][gq], mean[rt][gq], inv[rt][gq], gm, bt, hreg[rt][TL][gq]);
          if (all_rows || row0 + 32 * rg + 16 * rt + 4 * q + gq < row_end) out[(16 * rt + gq) * D + 16 * TL] = v;
        }
    }
  }
  WIDE_STAMP(p.stamps, 4);
  WIDE_STAMP_REAL(p.stamps, 6);
}

// ------------------------------------------------------------------------------------------------------------
// a7 for launches too small to fill the chip (16-row tiles: batches of up to ~100 pairs - model.predict at the
// reference's batch 32).  There wide_update_kernel is a chain of 32 weight slices through LDS with a barrier each:
// 29 us per launch whatever the rows, half of the forward's latency.  Here a 4-wave workgroup owns 16 rows, wave w the
// features [w D/4, (w + 1) D/4) of z, r and the candidate, and every operand comes straight from global memory / L2 in
// 16-byte pieces - the kernels in the prepared image's own order ([16-k slice][k quad][column][4 k]: a lane's four k of
// a slice are one load, MFMA step r takes component r of both operands), the rows' h from an LDS copy, the
// aggregated messages from their two sources - three slices ahead of the MFMAs: no staging, four barriers per tile.
// Exact f32 (v_mfma_f32_16x16x4_f32), the products and their order per output as wide_update_kernel's.
// ------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT >= 8 ? 512 : 256) void wide_update_small_kernel(GuParams p) {
  // WV waves: wave w owns the features [w D / WV, (w + 1) D / WV) - one 16-feature tile at D = 128 (8 waves), at D = 64 (4)
  constexpr int WV = NT >= 8 ? 8 : 4, T = 64 * WV;
  constexpr int D = 16 * NT, NL = NT / WV, LDH = D + 4, R = 16, NS = NT;  // NS 16-k slices per D of contraction
  static_assert(NL >= 1, "tile shape");
  __shared__ __align__(16) float hs[R * LDH];   // h of the tile's rows
  __shared__ __align__(16) float rhs[R * LDH];  // r * h
  __shared__ float part[2][4][R];               // LayerNorm partials: [sum | squared deviations][feature group][row]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * R;
  const int end = p.meta[kMetaEnd];
  if (row0 >= end) return;
  const int g = (p.n_ions > 1 && row0 >= p.meta[kMetaBase + 1]) ? 1 : 0;
  const int64_t ion_end = p.meta[kMetaBase + g] + p.meta[kMetaRows + g];
  const int64_t row_end = row0 + R < ion_end ? row0 + R : ion_end;
  if (row0 >= row_end) return;
  const float* img = p.img[g] + p.gu_off;
  const f32x4_t* P1 = reinterpret_cast<const f32x4_t*>(img);              // [Wz|Wr]: unit ((u * 4 + qq) * 2D + column)
  const f32x4_t* P2 = reinterpret_cast<const f32x4_t*>(img + 4 * D * D);  // Wh: unit ((u * 4 + qq) * D + column)
  const float* bias = img + 6 * D * D;                                    // bz br bh gamma beta
  for (int i = tid; i < R * D / 4; i += T) {
    const int r = i / (D / 4), c4 = i - r * (D / 4);
    stv4(hs + r * LDH + 4 * c4, ldv4(p.h + (row0 + r) * D + 4 * c4));
  }
  // the aggregated messages of the lane's row (A operand: row a, k = 4 q .. 4 q + 3 of a slice): two sources
  const int goff0 = agg_off(p.c2a[row0 + a], p.m_off, D) + 4 * q, goff1 = agg_off(p.c2b[row0 + a], p.m_off, D) + 4 * q;
  const int f0 = 16 * (wv * NL) + a;  // the lane's column of the wave's first feature tile
  f32x4_t z[NL], rr[NL], tt[NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const float b0 = bias[f0 + 16 * TL], b1 = bias[D + f0 + 16 * TL], b2 = bias[2 * D + f0 + 16 * TL];
    z[TL] = f32x4_t{b0, b0, b0, b0};
    rr[TL] = f32x4_t{b1, b1, b1, b1};
    tt[TL] = f32x4_t{b2, b2, b2, b2};
  }
  __syncthreads();
  struct Ops {
    f32x4_t av, aw, bz[NL], br[NL];
  };
  constexpr int kAhead = 3;
  // ---- phase 1: [z|r] pre-activations = [h|agg] x [Wz|Wr]
  {
    Ops o[kAhead];
    auto load1 = [&](int u, Ops& x) {
      if (u < NS) {
        x.av = ldv4(hs + a * LDH + 16 * u + 4 * q);
        x.aw = f32x4_t{0.f, 0.f, 0.f, 0.f};
      } else {
        x.av = ldv4(p.agg + goff0 + 16 * (u - NS));
        x.aw = ldv4(p.agg + goff1 + 16 * (u - NS));
      }
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        x.bz[TL] = P1[(u * 4 + q) * 2 * D + f0 + 16 * TL];
        x.br[TL] = P1[(u * 4 + q) * 2 * D + D + f0 + 16 * TL];
      }
    };
#pragma unroll
    for (int u = 0; u < kAhead - 1; ++u) load1(u, o[u]);
#pragma unroll
    for (int u = 0; u < 2 * NS; ++u) {
      if (u + kAhead - 1 < 2 * NS) load1(u + kAhead - 1, o[(u + kAhead - 1) % kAhead]);
      __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of this slice's MFMAs)
      const Ops& x = o[u % kAhead];
      const f32x4_t av = x.av + x.aw;  // (first slot first: the Reduce's order; h + 0 for a slice of h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int TL = 0; TL < NL; ++TL) {
          z[TL] = mfma_f32(av[r], x.bz[TL][r], z[TL]);
          rr[TL] = mfma_f32(av[r], x.br[TL][r], rr[TL]);
        }
    }
  }
  // ---- gates; r * h into LDS (accumulator layout: column a of the tile, rows 4 q + i)
#pragma unroll
  for (int TL = 0; TL < NL; ++TL)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[TL][i] = fsig(z[TL][i]);
      rhs[(4 * q + i) * LDH + f0 + 16 * TL] = gu_rh(rr[TL][i], hs[(4 * q + i) * LDH + f0 + 16 * TL]);
    }
  __syncthreads();
  // ---- phase 2: candidate = [r * h|agg] x Wh
  {
    struct Ops2 {
      f32x4_t av, aw, bh[NL];
    };
    Ops2 o[kAhead];
    auto load2 = [&](int u, Ops2& x) {
      if (u < NS) {
        x.av = ldv4(rhs + a * LDH + 16 * u + 4 * q);
        x.aw = f32x4_t{0.f, 0.f, 0.f, 0.f};
      } else {
        x.av = ldv4(p.agg + goff0 + 16 * (u - NS));
        x.aw = ldv4(p.agg + goff1 + 16 * (u - NS));
      }
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) x.bh[TL] = P2[(u * 4 + q) * D + f0 + 16 * TL];
    };
#pragma unroll
    for (int u = 0; u < kAhead - 1; ++u) load2(u, o[u]);
#pragma unroll
    for (int u = 0; u < 2 * NS; ++u) {
      if (u + kAhead - 1 < 2 * NS) load2(u + kAhead - 1, o[(u + kAhead - 1) % kAhead]);
      __builtin_amdgcn_sched_barrier(0);
      const Ops2& x = o[u % kAhead];
      const f32x4_t av = x.av + x.aw;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int TL = 0; TL < NL; ++TL) tt[TL] = mfma_f32(av[r], x.bh[TL][r], tt[TL]);
    }
  }
  // ---- blend, LayerNorm over the D features of a row, residual.  The partial sums are formed exactly as
  // wide_update_kernel forms them - per feature GROUP of D / 4 features: lane-wise over the group's tiles, then over the
  // 16 lanes, then over the four groups - so that a batch and its chunks agree bit for bit whichever kernel they take.
  // With 8 waves (D = 128) a group is two waves: the odd one hands its blended values to the even one through LDS.
  constexpr bool kPair = WV == 8;
  constexpr int NG = kPair ? 2 : NL;  // tiles of a feature group as the summing wave sees them
  static_assert(!kPair || NL == 1, "pairs of single-tile waves");
  float hv[NL][4];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hv[TL][i] = hs[(4 * q + i) * LDH + f0 + 16 * TL];
      tt[TL][i] = gu_blend(z[TL][i], hv[TL][i], tt[TL][i]);
    }
  float grp[NG][4];  // the group's blended values at this lane's positions
#pragma unroll
  for (int i = 0; i < 4; ++i) grp[0][i] = tt[0][i];  // (an odd wave of a pair does not sum: its copy goes through LDS)
  if (!kPair) {
#pragma unroll
    for (int TL = 1; TL < NL; ++TL)
#pragma unroll
      for (int i = 0; i < 4; ++i) grp[TL < NG ? TL : 0][i] = tt[TL][i];
  } else {
    float* xch = rhs;  // (r * h is dead: every wave is past phase 2's reads only after the barrier below)
    __syncthreads();
    if (wv & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xch[((wv >> 1) * 4 + i) * 64 + lane] = tt[0][i];
    }
    __syncthreads();
    if (!(wv & 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) grp[1][i] = xch[((wv >> 1) * 4 + i) * 64 + lane];
    }
  }
  const bool summing = !kPair || !(wv & 1);  // (wave-uniform)
  const int fgi = kPair ? wv >> 1 : wv;      // feature group
  if (summing) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sacc = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < NG; ++t2) sacc += grp[t2][i];
      const float sm = row16_sum_f(sacc);
      if (a == 0) part[0][fgi][4 * q + i] = sm;
    }
  }
  __syncthreads();
  float mean[4], inv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = 4 * q + i;
    float ms = 0.f;
#pragma unroll
    for (int f2 = 0; f2 < 4; ++f2) ms += part[0][f2][rl];
    mean[i] = ms * (1.0f / D);
    if (summing) {
      float vs = 0.f;
#pragma unroll
      for (int t2 = 0; t2 < NG; ++t2) {
        const float dv = grp[t2][i] - mean[i];
        vs = fmaf(dv, dv, vs);
      }
      const float vr = row16_sum_f(vs);
      if (a == 0) part[1][fgi][rl] = vr;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = 4 * q + i;
    float vs = 0.f;
#pragma unroll
    for (int f2 = 0; f2 < 4; ++f2) vs += part[1][f2][rl];
    inv[i] = gu_inv_std(vs, 1.0f / D, p.eps);
  }
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = f0 + 16 * TL;
    const float gm = bias[3 * D + f], bt = bias[4 * D + f];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = row0 + 4 * q + i;
      if (row < row_end) p.h[row * D + f] = gu_out(tt[TL][i], mean[i], inv[i], gm, bt, hv[TL][i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// a7 in mode IMPNN_ENCODER_F32X3_TYPED ("f32 (bf16x9 emulation)"): the same GatedUpdate with its three GEMMs on the bf16
// matrix pipe.  Every f32 operand is carried EXACTLY as three bf16 terms (x = b0 + b1 + b2: bf16 keeps fp32's exponent,
// 3 x 8 significant bits) and all nine cross products are accumulated in f32 - the f32 products themselves, summed in
// another order (encoder_typed.hip has the D = 32 form and the discussion of non-finite operands).  Nine
// v_mfma_f32_16x16x32_bf16 (16 cycles, 32 k) replace eight v_mfma_f32_16x16x4_f32 (32 cycles, 4 k): 9/16 of the matrix
// time, on a pipe that - unlike the exact-f32 one - co-executes with the vector ALU.
//   * the gate kernels arrive pre-split from the prepared image (wide_gu_image_x3_kernel), in 32-k slices of MFMA
//     B-operand order [plane][k octet][column][8 k]: a slice is copied global -> registers -> LDS verbatim;
//   * the rows ([h | agg], then [r*h | agg]) are split when a slice is parked in LDS (A-operand order
//     [plane][k octet][row][8 k]; 5.5 vector instructions per value);
//   * one workgroup of 8 waves per CU (a 32-k slice of [Wz|Wr] is 48 KB in three planes: two stages fill the LDS),
//     wave = 32 rows x NL feature tiles of z and of r: 72 MFMAs per wave and slice between barriers.
// ------------------------------------------------------------------------------------------------------------
constexpr int kGuX3Threads = 512;
// LDS bytes: two stages of (rows 12 KB + [Wz|Wr] slice 3 x 4 x 2D x 16 B) - phase 2 re-cuts the same memory into two
// stages of (rows + Wh slice) and the f32 copy of r*h - plus the LayerNorm partials
constexpr size_t gu_x3_lds_bytes(int D) { return 2 * (size_t)(12288 + 12 * 2 * D * 16) + 8 * kRT * 4; }

template <int NT>
__global__ __launch_bounds__(kGuX3Threads, 2) void wide_update_x3_kernel(GuParams p) {
  constexpr int D = 16 * NT, R = kRT, LDR = D + 4;
  constexpr int RG = 2, FG = 4, NL = NT / FG;
  constexpr int NS = NT;                     // 32-k slices of a 2D-deep GEMM
  constexpr int UA = 3 * 4 * R;              // 16-byte units of a row slice (768)
  constexpr int UB1 = 3 * 4 * 2 * D;         // ... of a [Wz|Wr] slice
  constexpr int UB2 = 3 * 4 * D;             // ... of a Wh slice
  constexpr int ST1 = UA + UB1, ST2 = UA + UB2;  // stage sizes (units)
  constexpr int kQ1 = (UB1 + kGuX3Threads - 1) / kGuX3Threads, kQ2 = (UB2 + kGuX3Threads - 1) / kGuX3Threads;
  static_assert(NL >= 1 && NT % 2 == 0, "tile shape");
  static_assert((size_t)2 * ST2 * 16 + (size_t)R * LDR * 4 <= (size_t)2 * ST1 * 16, "phase 2 fits phase 1's stages");
  extern __shared__ __align__(16) unsigned char smem_b[];
  uint4* const stage = reinterpret_cast<uint4*>(smem_b);                       // phase 1: 2 x ST1 units
  float* const rhs = reinterpret_cast<float*>(smem_b + (size_t)2 * ST2 * 16);  // phase 2: R x LDR f32, r * h
  float* const part = reinterpret_cast<float*>(smem_b + (size_t)2 * ST1 * 16);  // 2 x FG x R LayerNorm partials
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = lane & 15, q = lane >> 4;
  const int rg = wv % RG, fg = wv / RG;
  const int64_t row0 = (int64_t)blockIdx.x * p.tile_rows;
  const int end = p.meta[kMetaEnd];
  if (row0 >= end) return;
  const int g = (p.n_ions > 1 && row0 >= p.meta[kMetaBase + 1]) ? 1 : 0;
  const int64_t ion_end = p.meta[kMetaBase + g] + p.meta[kMetaRows + g];
  const int64_t row_end = row0 + p.tile_rows < ion_end ? row0 + p.tile_rows : ion_end;
  if (row0 >= row_end) return;
  const float* img = p.img[g] + p.gu_off;
  const uint4* P1 = reinterpret_cast<const uint4*>(img);
  const uint4* P2 = P1 + (size_t)NS * UB1;
  const float* bias = reinterpret_cast<const float*>(P2 + (size_t)NS * UB2);  // bz br bh gamma beta
  // a thread's piece of a row slice: row a_row, k = 4 a_pc .. 4 a_pc + 3 of the slice's 32
  const int a_row = tid >> 3, a_pc = tid & 7;
  const float* hsrc = p.h + (row0 + a_row) * D + 4 * a_pc;
  // the row's aggregated messages: two sources (wide_iota_kernel), as float offsets from p.agg
  const int goff0 = agg_off(p.c2a[row0 + a_row], p.m_off, D) + 4 * a_pc, goff1 = agg_off(p.c2b[row0 + a_row], p.m_off, D) + 4 * a_pc;
  // unit (plane, k octet a_pc >> 1, row a_row), 8-byte half a_pc & 1
  const int a_unit = (a_pc >> 1) * R + a_row, a_half = a_pc & 1;
  auto park_rows = [&](uint4* st, f32x4_t v) {  // 4 values -> three planes of 4 bf16
    unsigned w0[2], w1[2], w2[2];
    split_pair_w(v[0], v[1], w0[0], w1[0], w2[0]);
    split_pair_w(v[2], v[3], w0[1], w1[1], w2[1]);
    uint2* s2 = reinterpret_cast<uint2*>(st);
    s2[(0 * 4 * R + a_unit) * 2 + a_half] = make_uint2(w0[0], w0[1]);
    s2[(1 * 4 * R + a_unit) * 2 + a_half] = make_uint2(w1[0], w1[1]);
    s2[(2 * 4 * R + a_unit) * 2 + a_half] = make_uint2(w2[0], w2[1]);
  };
  struct Pre {
    f32x4_t av;
    uint4 bv[kQ1];
  };
  Pre preA, preB;
  auto fetch1 = [&](int u, Pre& pre) {
#pragma unroll
    for (int i = 0; i < kQ1; ++i)
      if (tid + kGuX3Threads * i < UB1) pre.bv[i] = P1[(size_t)u * UB1 + tid + kGuX3Threads * i];
    // (a slice of aggregated messages: the row's two sources, first slot first - the Reduce's order)
    pre.av = u < NS / 2 ? ldv4(hsrc + 32 * u) : ldv4(p.agg + goff0 + 32 * (u - NS / 2)) + ldv4(p.agg + goff1 + 32 * (u - NS / 2));
  };
  auto park1 = [&](uint4* st, const Pre& pre) {
#pragma unroll
    for (int i = 0; i < kQ1; ++i)
      if (tid + kGuX3Threads * i < UB1) st[UA + tid + kGuX3Threads * i] = pre.bv[i];
    park_rows(st, pre.av);
  };
  f32x4_t z[2][NL], rr[2][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = 16 * (fg * NL + TL) + a;
    const float b0 = bias[f], b1 = bias[D + f];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      z[rt][TL] = f32x4_t{b0, b0, b0, b0};
      rr[rt][TL] = f32x4_t{b1, b1, b1, b1};
    }
  }
  // operands of one slice: rows (A) and kernel columns (B), three planes each
  auto read_a = [&](const uint4* st, int rt, bf16x8_t (&av)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      av[pl] = __builtin_bit_cast(bf16x8_t, st[(pl * 4 + q) * R + 32 * rg + 16 * rt + a]);
  };
  auto read_b = [&](const uint4* st, int ncols, int col, bf16x8_t (&bv)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bv[pl] = __builtin_bit_cast(bf16x8_t, st[UA + (pl * 4 + q) * ncols + col]);
  };
  // acc += A B: all nine cross products, smallest first
  auto mma9 = [&](f32x4_t& acc, const bf16x8_t (&av)[3], const bf16x8_t (&bv)[3]) {
    acc = mfma_bf16(av[2], bv[2], acc);
    acc = mfma_bf16(av[1], bv[2], acc);
    acc = mfma_bf16(av[2], bv[1], acc);
    acc = mfma_bf16(av[0], bv[2], acc);
    acc = mfma_bf16(av[2], bv[0], acc);
    acc = mfma_bf16(av[1], bv[1], acc);
    acc = mfma_bf16(av[0], bv[1], acc);
    acc = mfma_bf16(av[1], bv[0], acc);
    acc = mfma_bf16(av[0], bv[0], acc);
  };
  // One slice of phase 1 on stage `cur` while slice u + 1 is parked in stage `oth`.  The MFMAs of a wave are paced by
  // the matrix pipe (16 cycles each); everything else of the iteration - the LDS reads of the second feature tile's
  // operands, the split of the next row slice and its LDS stores - is interleaved with them (sched_group_barrier:
  // without it the compiler emits reads, stores and MFMAs as three serial blocks and the pipe idles half the time).
  // (Row tiles beyond tile_rows are multiplied too - wasted only in launches too small to fill the chip - so that
  //  the iteration is one basic block.)
  auto slice1 = [&](const uint4* cur, uint4* oth, const Pre* pre, bool do_park) {
    bf16x8_t av[2][3], bz[NL][3], br[NL][3];
    read_a(cur, 0, av[0]);
    read_a(cur, 1, av[1]);
    read_b(cur, 2 * D, 16 * (fg * NL) + a, bz[0]);
    read_b(cur, 2 * D, D + 16 * (fg * NL) + a, br[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int TL = 1; TL < NL; ++TL) {
      read_b(cur, 2 * D, 16 * (fg * NL + TL) + a, bz[TL]);
      read_b(cur, 2 * D, D + 16 * (fg * NL + TL) + a, br[TL]);
    }
    if (do_park) park1(oth, *pre);
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      mma9(z[0][TL], av[0], bz[TL]);
      mma9(z[1][TL], av[1], bz[TL]);
      mma9(rr[0][TL], av[0], br[TL]);
      mma9(rr[1][TL], av[1], br[TL]);
    }
    // 36 NL MFMAs; 6 (NL - 1) LDS reads, ~10 LDS stores and ~45 vector instructions to hide between them
#pragma unroll
    for (int i = 0; i < 6 * (NL - 1); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
  };
  fetch1(0, preA);
  fetch1(1, preB);
  park1(stage, preA);
  __syncthreads();
  float hreg[2][NL][4];
  auto load_hreg = [&]() {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int TL = 0; TL < NL; ++TL)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          hreg[rt][TL][gq] = p.h[(row0 + 32 * rg + 16 * rt + 4 * q + gq) * D + 16 * (fg * NL + TL) + a];
  };
  // iteration u: slice u is in stage u & 1, slice u + 1 in registers, slice u + 2 is requested
  for (int u = 0; u < NS; u += 2) {
    if (u + 2 < NS) fetch1(u + 2, preA);
    else load_hreg();
    slice1(stage, stage + ST1, &preB, true);
    __syncthreads();
    if (u + 3 < NS) fetch1(u + 3, preB);
    slice1(stage + ST1, stage, &preA, u + 2 < NS);
    __syncthreads();
  }
  // ---- gates; r * h (f32) into LDS: phase 2 parks its first NS / 2 row slices from there
  struct Pre2 {
    f32x4_t av;
    uint4 bv[kQ2];
  };
  Pre2 qA, qB;
  auto fetch2 = [&](int u, Pre2& pre) {
#pragma unroll
    for (int i = 0; i < kQ2; ++i)
      if (tid + kGuX3Threads * i < UB2) pre.bv[i] = P2[(size_t)u * UB2 + tid + kGuX3Threads * i];
    if (u >= NS / 2) pre.av = ldv4(p.agg + goff0 + 32 * (u - NS / 2)) + ldv4(p.agg + goff1 + 32 * (u - NS / 2));
  };
  auto park2 = [&](int u, uint4* st, const Pre2& pre) {
#pragma unroll
    for (int i = 0; i < kQ2; ++i)
      if (tid + kGuX3Threads * i < UB2) st[UA + tid + kGuX3Threads * i] = pre.bv[i];
    park_rows(st, u < NS / 2 ? ldv4(rhs + a_row * LDR + 32 * u + 4 * a_pc) : pre.av);
  };
  fetch2(0, qA);
  fetch2(1, qB);
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int TL = 0; TL < NL; ++TL)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        z[rt][TL][gq] = fsig(z[rt][TL][gq]);
        rhs[(32 * rg + 16 * rt + 4 * q + gq) * LDR + 16 * (fg * NL + TL) + a] = gu_rh(rr[rt][TL][gq], hreg[rt][TL][gq]);
      }
  f32x4_t tt[2][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const float b2 = bias[2 * D + 16 * (fg * NL + TL) + a];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) tt[rt][TL] = f32x4_t{b2, b2, b2, b2};
  }
  __syncthreads();  // r * h complete (and every read of phase 1's stages is done)
  uint4* const stage2 = stage;  // 2 x ST2 units
  park2(0, stage2, qA);
  __syncthreads();
  auto slice2 = [&](const uint4* cur, int u_next, uint4* oth, const Pre2* pre, bool do_park) {
    bf16x8_t av[2][3], bv[NL][3];
    read_a(cur, 0, av[0]);
    read_a(cur, 1, av[1]);
    read_b(cur, D, 16 * (fg * NL) + a, bv[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int TL = 1; TL < NL; ++TL) read_b(cur, D, 16 * (fg * NL + TL) + a, bv[TL]);
    if (do_park) park2(u_next, oth, *pre);
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      mma9(tt[0][TL], av[0], bv[TL]);
      mma9(tt[1][TL], av[1], bv[TL]);
    }
#pragma unroll
    for (int i = 0; i < 3 * (NL - 1); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
  };
  for (int u = 0; u < NS; u += 2) {
    if (u + 2 < NS) fetch2(u + 2, qA);
    slice2(stage2, u + 1, stage2 + ST2, &qB, true);
    __syncthreads();
    if (u + 3 < NS) fetch2(u + 3, qB);
    slice2(stage2 + ST2, u + 2, stage2, &qA, u + 2 < NS);
    __syncthreads();
  }
  // ---- blend, LayerNorm over the D features of a row, residual (models/layers.py:150-156): as wide_update_kernel
  float sum[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float sacc = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float hv = hreg[rt][TL][gq];
        const float nv = gu_blend(z[rt][TL][gq], hv, tt[rt][TL][gq]);
        tt[rt][TL][gq] = nv;
        sacc += nv;
      }
      sum[rt][gq] = row16_sum_f(sacc);
      if (a == 0) part[fg * R + 32 * rg + 16 * rt + 4 * q + gq] = sum[rt][gq];
    }
  __syncthreads();
  float mean[2][4], inv[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = 32 * rg + 16 * rt + 4 * q + gq;
      float ms = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) ms += part[f2 * R + rl];
      mean[rt][gq] = ms * (1.0f / D);
      float vs = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float dv = tt[rt][TL][gq] - mean[rt][gq];
        vs = fmaf(dv, dv, vs);
      }
      vs = row16_sum_f(vs);
      if (a == 0) part[FG * R + fg * R + rl] = vs;
    }
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = FG * R + 32 * rg + 16 * rt + 4 * q + gq;
      float vs = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) vs += part[f2 * R + rl];
      inv[rt][gq] = gu_inv_std(vs, 1.0f / D, p.eps);
    }
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = 16 * (fg * NL + TL) + a;
    const float gm = bias[3 * D + f], bt = bias[4 * D + f];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int64_t row = row0 + 32 * rg + 16 * rt + 4 * q + gq;
        if (row < row_end)
          p.h[row * D + f] = gu_out(tt[rt][TL][gq], mean[rt][gq], inv[rt][gq], gm, bt, hreg[rt][TL][gq]);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same update on 128-row tiles (batches that fill the chip): per 32-k slice the 48 KB of pre-split gate kernels are
// shared by twice the rows, and a wave multiplies 64 rows x NL feature tiles of z and of r - 144 MFMAs per slice for 24
// operand fetches (the 64-row form: 72 for 18), so the LDS traffic per MFMA is 0.67 of the 64-row kernel's and the L2 -> LDS
// traffic of the kernels half.  One workgroup of 8 waves per CU, 256 VGPRs per lane.
//   * kernel slices go global -> LDS directly (global_load_lds_dwordx4: 16 B per lane, a wave's 64 lanes fill 1 KB of
//     consecutive LDS; no staging registers, no LDS store instructions); slice u + 1 lands in the other stage while
//     slice u is multiplied;
//   * row slices go global -> registers (one slice ahead) -> split -> the other stage, at the top of a slice;
//   * the nine products of an output tile form a dependent chain: the MFMAs are issued product by product ACROSS the
//     wave's four chains of a row tile (a chain's next link is four instructions away), and the operands of the next
//     row tile are requested in front of them;
//   * the split of slice u + 1's rows and their LDS stores sit between the MFMAs of slice u's first row tile
//     (sched_group_barrier): the bf16 pipe co-executes with the vector ALU;
//   * phase 2 re-cuts the LDS into two (rows + Wh slice) stages and an unpadded f32 copy of r * h - 160 KB at D = 128 -
//     and runs like phase 1 (one barrier per slice); the LayerNorm partials reuse a stage at the end;
//   * the tiles of the last, partial round are cut into 16-row pieces (wide_update_x3b_kernel below, MINI).
// ------------------------------------------------------------------------------------------------------------
constexpr int kRT3 = 128;
// LDS: phase 1 two stages of (rows 24 KB + [Wz|Wr] slice); phase 2 re-cuts the same memory into two row stages, two Wh
// stages and the f32 copy of r * h (unpadded) - 160 KB at D = 128; the LayerNorm partials reuse the stages at the end
constexpr size_t gu_x3b_lds_bytes(int D) {
  const size_t p1 = 2 * (size_t)(3 * 4 * kRT3 * 16 + 12 * 2 * D * 16);
  const size_t p2 = 2 * (size_t)(3 * 4 * kRT3 * 16 + 12 * D * 16) + (size_t)kRT3 * D * 4;
  return p1 > p2 ? p1 : p2;
}


template <int NT, bool MINI>
__device__ __forceinline__ void x3b_tile(const GuParams& p, const int64_t row0, const int g, unsigned char* smem_b) {
  constexpr int D = 16 * NT, R = kRT3, LDR = D, T = kGuX3Threads;
  constexpr int RG = 2, FG = 4, NL = NT / FG, RTW = R / (16 * RG);  // a wave: RTW = 4 row tiles x NL feature tiles
  constexpr int NS = NT;                     // 32-k slices of a 2D-deep GEMM
  constexpr int UA = 3 * 4 * R;              // 16-byte units of a row slice
  constexpr int UB1 = 3 * 4 * 2 * D;         // ... of a [Wz|Wr] slice
  constexpr int UB2 = 3 * 4 * D;             // ... of a Wh slice
  constexpr int ST1 = UA + UB1;
  constexpr int RP = R / 64;                 // row pieces a thread parks per slice
  static_assert(NL >= 1 && NT % 2 == 0 && NS >= 4 && RTW == 4 && RP == 2, "tile shape");
  static_assert(UB1 % 64 == 0 && UB2 % 64 == 0, "a kernel slice is whole 1 KB wave transfers");
  constexpr int ST2 = UA + UB2;
  static_assert((size_t)2 * ST2 * 16 + (size_t)R * LDR * 4 <= 160 * 1024, "phase 2 fits the LDS");
  static_assert((size_t)8 * R * 4 <= (size_t)2 * ST2 * 16, "the LayerNorm partials fit the stages");
  uint4* const stage = reinterpret_cast<uint4*>(smem_b);                                  // phase 1: 2 x ST1 units
  uint4* const stage2 = stage;                                                            // phase 2: 2 x ST2 units (rows | Wh slice)
  float* const rhs = reinterpret_cast<float*>(smem_b + (size_t)2 * ST2 * 16);             // phase 2: R x LDR f32, r * h
  float* const part = reinterpret_cast<float*>(smem_b);                                   // epilogue: 2 x FG x R LayerNorm partials
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = lane & 15, q = lane >> 4;
  const int rg = wv % RG, fg = wv / RG;
  // MINI: a 16-row piece of a tile of the last, partial round (wide_update_x3b_kernel): the same stages and slices, but
  // only the waves of row group 0 multiply, and only their first row tile
  auto active = [&](int rt) { return !MINI || (rg == 0 && rt == 0); };  // (wave-uniform)
  WIDE_STAMP(p.stamps, 0);
  WIDE_STAMP_REAL(p.stamps, 5);
  const float* img = p.img[g] + p.gu_off;
  const uint4* P1 = reinterpret_cast<const uint4*>(img);
  const uint4* P2 = P1 + (size_t)NS * UB1;
  const float* bias = reinterpret_cast<const float*>(P2 + (size_t)NS * UB2);  // bz br bh gamma beta
  // a thread's pieces of a row slice: rows a_row and a_row + 64, k = 4 a_pc .. 4 a_pc + 3 of the slice's 32
  const int a_row = tid >> 3, a_pc = tid & 7;
  const float* hsrc = p.h + (row0 + a_row) * D + 4 * a_pc;
  // the aggregated messages of the thread's two rows: two sources each (wide_iota_kernel), as float offsets from p.agg
  int goff[RP][2];
#pragma unroll
  for (int i = 0; i < RP; ++i) {
    const int ca = p.c2a[row0 + a_row + 64 * i], cb = p.c2b[row0 + a_row + 64 * i];
    goff[i][0] = agg_off(ca, p.m_off, D) + 4 * a_pc;
    goff[i][1] = agg_off(cb, p.m_off, D) + 4 * a_pc;
  }
  f32x4_t pavb[RP];  // the second source's piece (added when the slice is parked)
  const int a_unit = (a_pc >> 1) * R + a_row, a_half = a_pc & 1;  // unit (plane, k octet a_pc >> 1, row), 8-byte half
  auto park_rows = [&](uint4* st, f32x4_t v, int piece) {  // 4 values -> three planes of 4 bf16
    unsigned w0[2], w1[2], w2[2];
    split_pair_w(v[0], v[1], w0[0], w1[0], w2[0]);
    split_pair_w(v[2], v[3], w0[1], w1[1], w2[1]);
    uint2* s2 = reinterpret_cast<uint2*>(st);
    const int un = a_unit + 64 * piece;
    s2[(0 * 4 * R + un) * 2 + a_half] = make_uint2(w0[0], w0[1]);
    s2[(1 * 4 * R + un) * 2 + a_half] = make_uint2(w1[0], w1[1]);
    s2[(2 * 4 * R + un) * 2 + a_half] = make_uint2(w2[0], w2[1]);
  };
  // `units` 16-byte units from global to LDS, verbatim: wave w moves units 64 (8 i + w) .. + 63 with its i-th instruction
  auto dma = [&](const uint4* src, uint4* dst, int units) {
#pragma unroll
    for (int i = 0; i < (units + T - 1) / T; ++i) {
      const int ub = 64 * (8 * i + wv);
      if (ub < units)  // (wave-uniform)
        __builtin_amdgcn_global_load_lds((const void*)(src + ub + lane), (lds_ptr_t)(dst + ub), 16, 0, 0);
    }
  };
  // workgroup barrier behind everything this wave has in flight (kernel slices on their way into LDS included)
  auto wg_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  f32x4_t pav[RP];
  auto fetch_rows1 = [&](int u) {
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      if (u < NS / 2) {
        pav[i] = ldv4(hsrc + 32 * u + (size_t)64 * i * D);
      } else {
        pav[i] = ldv4(p.agg + goff[i][0] + 32 * (u - NS / 2));
        pavb[i] = ldv4(p.agg + goff[i][1] + 32 * (u - NS / 2));
      }
    }
  };
  f32x4_t z[RTW][NL], rr[RTW][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const int f = 16 * (fg * NL + TL) + a;
    const float b0 = bias[f], b1 = bias[D + f];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
      z[rt][TL] = f32x4_t{b0, b0, b0, b0};
      rr[rt][TL] = f32x4_t{b1, b1, b1, b1};
    }
  }
  auto read_a = [&](const uint4* st, int rt, bf16x8_t (&av)[3]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      av[pl] = __builtin_bit_cast(bf16x8_t, st[(pl * 4 + q) * R + 64 * rg + 16 * rt + a]);
  };
  float hreg[RTW][NL][4];
  auto load_hreg = [&]() {
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt)
      if (active(rt)) {
#pragma unroll
        for (int TL = 0; TL < NL; ++TL)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            hreg[rt][TL][gq] = p.h[(row0 + 64 * rg + 16 * rt + 4 * q + gq) * D + 16 * (fg * NL + TL) + a];
      }
  };
  // the nine products, smallest first: (row plane, kernel plane)
  constexpr int kPa[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, kPb[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
  auto slice1 = [&](const uint4* cur, uint4* oth, int u) {
    bf16x8_t bz[NL][3], br[NL][3], av[2][3];
    // operands in the order the products take them (plane 2 of both first): the first MFMA waits for 5 fetches, not 15
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {
      av[0][pl] = __builtin_bit_cast(bf16x8_t, cur[(pl * 4 + q) * R + 64 * rg + a]);
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        bz[TL][pl] = __builtin_bit_cast(bf16x8_t, cur[UA + (pl * 4 + q) * 2 * D + 16 * (fg * NL + TL) + a]);
        br[TL][pl] = __builtin_bit_cast(bf16x8_t, cur[UA + (pl * 4 + q) * 2 * D + D + 16 * (fg * NL + TL) + a]);
      }
    }
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) {
      if (rt + 1 < RTW) read_a(cur, rt + 1, av[(rt + 1) & 1]);
      if (rt > 0) __builtin_amdgcn_sched_barrier(0);
      // Row tile 0 shares its scheduling region with the split of slice u + 1's rows (in the staging registers since the
      // last slice) and their LDS stores: vector instructions issue between the MFMAs of the bf16 pipe for free.
      if (rt == 0 && u + 1 < NS) {
        // (the slices of the aggregated messages: the row's two sources are added here - first slot first)
        park_rows(oth, u + 1 >= NS / 2 ? pav[0] + pavb[0] : pav[0], 0);
        if (!MINI) park_rows(oth, u + 1 >= NS / 2 ? pav[1] + pavb[1] : pav[1], 1);
      }
      if (active(rt)) {
#pragma unroll
        for (int pr = 0; pr < 9; ++pr)
#pragma unroll
          for (int TL = 0; TL < NL; ++TL) {
            z[rt][TL] = mfma_bf16(av[rt & 1][kPa[pr]], bz[TL][kPb[pr]], z[rt][TL]);
            rr[rt][TL] = mfma_bf16(av[rt & 1][kPa[pr]], br[TL][kPb[pr]], rr[rt][TL]);
          }
      }
      if (!MINI && rt == 0 && u + 1 < NS) {
#pragma unroll
        for (int i = 0; i < 6 * NL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
        }
#pragma unroll
        for (int i = 0; i < 3 * NL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rt == 0) {
        if (u + 1 < NS) {  // slice u + 2's rows requested; slice u + 1's kernels on their way into the other stage
          if (u + 2 < NS) fetch_rows1(u + 2);  // (behind the LDS stores: a store behind a transfer in flight waits for it)
          dma(P1 + (size_t)(u + 1) * UB1, oth + UA, UB1);
        } else {
          load_hreg();
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  fetch_rows1(0);
  park_rows(stage, pav[0], 0);
  if (!MINI) park_rows(stage, pav[1], 1);
  __builtin_amdgcn_sched_barrier(0);
  fetch_rows1(1);
  dma(P1, stage + UA, UB1);
  wg_barrier();
  WIDE_STAMP(p.stamps, 1);
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    slice1(stage + (u & 1) * ST1, stage + ((u + 1) & 1) * ST1, u);
    wg_barrier();
  }
  WIDE_STAMP(p.stamps, 2);
  // ---- gates; r * h (f32) into LDS: phase 2 parks its first NS / 2 row slices from there
  auto fetch_rows2 = [&](int u) {    // (u >= NS / 2: the aggregated messages)
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      pav[i] = ldv4(p.agg + goff[i][0] + 32 * (u - NS / 2));
      pavb[i] = ldv4(p.agg + goff[i][1] + 32 * (u - NS / 2));
    }
  };
  auto park2 = [&](uint4* st, int u) {
#pragma unroll
    for (int i = 0; i < (MINI ? 1 : RP); ++i)
      park_rows(st, u < NS / 2 ? ldv4(rhs + (a_row + 64 * i) * LDR + 32 * u + 4 * a_pc) : pav[i] + pavb[i], i);
  };
  // (every wave is past the last barrier of phase 1: the stages are free)
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt)) {
#pragma unroll
      for (int TL = 0; TL < NL; ++TL)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          if (MINI) z[rt][TL][gq] = fsig(z[rt][TL][gq]);  // (whole tiles: between the MFMAs of phase 2, slice2)
          rhs[(64 * rg + 16 * rt + 4 * q + gq) * LDR + 16 * (fg * NL + TL) + a] = gu_rh(rr[rt][TL][gq], hreg[rt][TL][gq]);
        }
    }
  f32x4_t tt[RTW][NL];
#pragma unroll
  for (int TL = 0; TL < NL; ++TL) {
    const float b2 = bias[2 * D + 16 * (fg * NL + TL) + a];
#pragma unroll
    for (int rt = 0; rt < RTW; ++rt) tt[rt][TL] = f32x4_t{b2, b2, b2, b2};
  }
  __builtin_amdgcn_sched_barrier(0);
  dma(P2, stage2 + UA, UB2);  // (behind the LDS stores above: a store behind a transfer in flight would wait for it)
  wg_barrier();  // r * h complete
  park2(stage2, 0);
  wg_barrier();
  auto slice2 = [&](const uint4* cur, uint4* oth, int u) {
    bf16x8_t bv[NL][3], av[2][2][3];
#pragma unroll
    for (int pl = 2; pl >= 0; --pl) {  // (in the order the products take them)
      av[0][0][pl] = __builtin_bit_cast(bf16x8_t, cur[(pl * 4 + q) * R + 64 * rg + a]);
      av[0][1][pl] = __builtin_bit_cast(bf16x8_t, cur[(pl * 4 + q) * R + 64 * rg + 16 + a]);
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) bv[TL][pl] = __builtin_bit_cast(bf16x8_t, cur[UA + (pl * 4 + q) * D + 16 * (fg * NL + TL) + a]);
    }
#pragma unroll
    for (int rp = 0; rp < RTW / 2; ++rp) {  // two row tiles at a time: four chains
      if (rp + 1 < RTW / 2) {
        read_a(cur, 2 * rp + 2, av[(rp + 1) & 1][0]);
        read_a(cur, 2 * rp + 3, av[(rp + 1) & 1][1]);
      }
      if (rp > 0) __builtin_amdgcn_sched_barrier(0);
      if (rp == 0 && u + 1 < NS) park2(oth, u + 1);  // (between the MFMAs, as in phase 1)
      // the update gate's sigmoids are not needed before the blend: they ride between the MFMAs of the second row-tile
      // pair of slices 0 and 1 (two row tiles each) instead of standing in front of phase 2
      if (!MINI && rp == 1 && u < 2) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int TL = 0; TL < NL; ++TL)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) z[2 * u + r2][TL][gq] = fsig(z[2 * u + r2][TL][gq]);
      }
      if (active(2 * rp)) {  // (MINI: row tile 1 rides along with row tile 0 - its rows are never stored)
#pragma unroll
        for (int pr = 0; pr < 9; ++pr)
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int TL = 0; TL < NL; ++TL)
              tt[2 * rp + r2][TL] = mfma_bf16(av[rp & 1][r2][kPa[pr]], bv[TL][kPb[pr]], tt[2 * rp + r2][TL]);
      }
      if (!MINI && rp == 0 && u + 1 < NS) {
#pragma unroll
        for (int i = 0; i < 6 * NL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
#pragma unroll
        for (int i = 0; i < 3 * NL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
      if (!MINI && rp == 1 && u < 2) {
#pragma unroll
        for (int i = 0; i < 9 * NL; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU (two of every four are quarter-rate)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (rp == 0 && u + 1 < NS) {
        if (u + 2 < NS && u + 2 >= NS / 2) fetch_rows2(u + 2);
        dma(P2 + (size_t)(u + 1) * UB2, oth + UA, UB2);  // slice u + 1's Wh slice straight from global
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // (the first slice of aggregated messages, NS / 2, is requested inside slice NS / 2 - 2 and parked inside NS / 2 - 1)
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    slice2(stage2 + (u & 1) * ST2, stage2 + ((u + 1) & 1) * ST2, u);
    wg_barrier();
  }
  WIDE_STAMP(p.stamps, 3);
  // ---- blend, LayerNorm over the D features of a row, residual (models/layers.py:150-156): as wide_update_kernel
  // sum over the 16 lanes of a quarter wave, all of the wave's rows step by step (a row's next step is 16 instructions
  // behind its last: no stall between dependent DPP operations)
  auto row16_sum_all = [&](float (&v)[RTW][4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
        if (active(rt))
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int iv = __builtin_bit_cast(int, v[rt][gq]);
            const int o = st == 0 ? __builtin_amdgcn_update_dpp(0, iv, 0x121, 0xf, 0xf, true)
                          : st == 1 ? __builtin_amdgcn_update_dpp(0, iv, 0x122, 0xf, 0xf, true)
                          : st == 2 ? __builtin_amdgcn_update_dpp(0, iv, 0x124, 0xf, 0xf, true)
                                    : __builtin_amdgcn_update_dpp(0, iv, 0x128, 0xf, 0xf, true);
            v[rt][gq] += __builtin_bit_cast(float, o);
          }
  };
  float sum[RTW][4];
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt))
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      float sacc = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float hv = hreg[rt][TL][gq];
        const float nv = gu_blend(z[rt][TL][gq], hv, tt[rt][TL][gq]);
        tt[rt][TL][gq] = nv;
        sacc += nv;
      }
      sum[rt][gq] = sacc;
    }
  row16_sum_all(sum);
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt))
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        if (a == 0) part[fg * R + 64 * rg + 16 * rt + 4 * q + gq] = sum[rt][gq];
  __syncthreads();
  float mean[RTW][4], inv[RTW][4], var[RTW][4];
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt))
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = 64 * rg + 16 * rt + 4 * q + gq;
      float ms = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) ms += part[f2 * R + rl];
      mean[rt][gq] = ms * (1.0f / D);
      float vs = 0.f;
#pragma unroll
      for (int TL = 0; TL < NL; ++TL) {
        const float dv = tt[rt][TL][gq] - mean[rt][gq];
        vs = fmaf(dv, dv, vs);
      }
      var[rt][gq] = vs;
    }
  row16_sum_all(var);
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt))
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        if (a == 0) part[FG * R + fg * R + 64 * rg + 16 * rt + 4 * q + gq] = var[rt][gq];
  __syncthreads();
#pragma unroll
  for (int rt = 0; rt < RTW; ++rt)
    if (active(rt))
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int rl = FG * R + 64 * rg + 16 * rt + 4 * q + gq;
      float vs = 0.f;
#pragma unroll
      for (int f2 = 0; f2 < FG; ++f2) vs += part[f2 * R + rl];
      inv[rt][gq] = gu_inv_std(vs, 1.0f / D, p.eps);
    }
  WIDE_STAMP(p.stamps, 7);
  // Every row of the tile (MINI: of its 16-row piece) is stored: the rows past the ion's last are padding of the row
  // space (the gap behind an ion, the rows behind the last one) that nothing reads as a source, a target or a pooled row.
  {
    float* const out = p.h + (row0 + 64 * rg + 4 * q) * D + 16 * fg * NL + a;
#pragma unroll
    for (int TL = 0; TL < NL; ++TL) {
      const int f = 16 * (fg * NL + TL) + a;
      const float gm = bias[3 * D + f], bt = bias[4 * D + f];
#pragma unroll
      for (int rt = 0; rt < RTW; ++rt)
        if (active(rt))
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
          out[(16 * rt + gq) * D + 16 * TL] = gu_out(tt[rt][TL][gq], mean[rt][gq], inv[rt][gq], gm, bt, hreg[rt][TL][gq]);
    }
  }
  WIDE_STAMP(p.stamps, 4);
  WIDE_STAMP_REAL(p.stamps, 6);
}

// Grid: tiles_max workgroups, one per 128-row tile of the row space, then 8 x (cus - 1) "mini" workgroups.  The tiles of
// the whole rounds (cus at a time) are updated by their own workgroup; the tiles of the last, partial round - a tile takes
// ~50 us whatever the number of CUs at work - are cut into eight 16-row pieces, one mini workgroup each, so that the round
// costs a third of a tile.  Workgroups are dispatched in grid order: the minis start as the CUs run out of whole tiles.
template <int NT>
__global__ __launch_bounds__(kGuX3Threads, 1) void wide_update_x3b_kernel(GuParams p) {
  extern __shared__ __align__(16) unsigned char smem_b[];
  constexpr int R = kRT3;
  const int end = p.meta[kMetaEnd];
  const int t_live = (end + R - 1) / R;
  const int t_full = p.cus > 0 ? t_live / p.cus * p.cus : t_live;
  const bool split = t_full > 0 && t_full < t_live;  // (a single partial round runs all at once: nothing to gain)
  int tile, sub = -1;
  if ((int)blockIdx.x < p.tiles_max) {
    tile = blockIdx.x;
    if (tile >= t_live || (split && tile >= t_full)) return;
  } else {
    if (!split) return;
    const int m = (int)blockIdx.x - p.tiles_max;
    tile = t_full + (m >> 3);
    sub = m & 7;
    if (tile >= t_live) return;
  }
  const int64_t tile0 = (int64_t)tile * R;
  const int g = (p.n_ions > 1 && tile0 >= p.meta[kMetaBase + 1]) ? 1 : 0;
  const int64_t ion_end = p.meta[kMetaBase + g] + p.meta[kMetaRows + g];
  const int64_t row0 = tile0 + (sub >= 0 ? 16 * sub : 0);
  if (row0 >= ion_end) return;
  if (sub >= 0) x3b_tile<NT, true>(p, row0, g, smem_b);
  else x3b_tile<NT, false>(p, row0, g, smem_b);
}

// a8: one thread per 16-byte piece of a pooled row, 4 rows in flight, ascending n.
__global__ __launch_bounds__(256) void wide_pool_kernel(Inputs in, const int32_t* __restrict__ kept,
                                                        const int32_t* __restrict__ rowbase,
                                                        const float* __restrict__ h, float* __restrict__ pooled0,
                                                        float* __restrict__ pooled1, int D) {
  const int qd = D >> 2;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t mol = t / qd;
  const int c4 = (int)(t - mol * qd);
  if (mol >= (int64_t)in.n_ions * in.B) return;
  const int g = mol >= in.B ? 1 : 0, b = (int)(mol - (int64_t)g * in.B);
  const int r = kept[mol];
  const int32_t* ids = in.atom_ids[g] + (int64_t)b * in.N;
  const float* src = h + (int64_t)rowbase[mol] * D + 4 * c4;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  int n = 0;
  for (; n + 4 <= r; n += 4) {
    const f32x4_t v0 = ldv4(src + (int64_t)n * D), v1 = ldv4(src + (int64_t)(n + 1) * D);
    const f32x4_t v2 = ldv4(src + (int64_t)(n + 2) * D), v3 = ldv4(src + (int64_t)(n + 3) * D);
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
    acc += ids[n] > 0 ? v0 : zero;
    acc += ids[n + 1] > 0 ? v1 : zero;
    acc += ids[n + 2] > 0 ? v2 : zero;
    acc += ids[n + 3] > 0 ? v3 : zero;
  }
  for (; n < r; ++n)
    if (ids[n] > 0) acc += ldv4(src + (int64_t)n * D);
  stv4((g ? pooled1 : pooled0) + (int64_t)b * D + 4 * c4, acc);
}

// The GatedUpdate part of a prepared step: [Wz|Wr] and Wh in slice order [16-k slice][k quad][column][k & 3], then
// the five vectors.  src = the step's canonical weights behind bond_transform (include/impnn.h).
__global__ void wide_gu_image_kernel(const float* __restrict__ src, float* __restrict__ dst, int D) {
  const float* Wz = src;
  const float* bz = Wz + 2 * D * D;
  const float* Wr = bz + D;
  const float* br = Wr + 2 * D * D;
  const float* Wh = br + D;
  const float* bh = Wh + 2 * D * D;
  const float* gamma = bh + D;
  const float* beta = gamma + D;
  const int n1 = 4 * D * D, n2 = 2 * D * D, total = n1 + n2 + 5 * D;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    float v;
    if (t < n1) {
      const int r = t & 3, c = (t >> 2) % (2 * D), qq = ((t >> 2) / (2 * D)) & 3, u = (t >> 2) / (2 * D) >> 2;
      const int k = 16 * u + 4 * qq + r;
      v = c < D ? Wz[k * D + c] : Wr[k * D + c - D];
    } else if (t < n1 + n2) {
      const int s = t - n1;
      const int r = s & 3, c = (s >> 2) % D, qq = ((s >> 2) / D) & 3, u = (s >> 2) / D >> 2;
      v = Wh[(16 * u + 4 * qq + r) * D + c];
    } else {
      const int s = t - n1 - n2, which = s / D, f = s - which * D;
      const float* vec = which == 0 ? bz : which == 1 ? br : which == 2 ? bh : which == 3 ? gamma : beta;
      v = vec[f];
    }
    dst[t] = v;
  }
}

// The same for mode 3: [Wz|Wr] and Wh as three bf16 planes (w = b0 + b1 + b2 exactly) in 32-k slices of MFMA B-operand
// order [slice][plane][k octet][column][8 k] (16-byte units: a lane's operand of one v_mfma_f32_16x16x32_bf16), then the
// five f32 vectors.
__global__ void wide_gu_image_x3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int D) {
  const float* Wz = src;
  const float* bz = Wz + 2 * D * D;
  const float* Wr = bz + D;
  const float* br = Wr + 2 * D * D;
  const float* Wh = br + D;
  const float* bh = Wh + 2 * D * D;
  const float* gamma = bh + D;
  const float* beta = gamma + D;
  const int NS = D / 16;                    // 32-k slices of a 2D-deep GEMM
  const int n1 = 2 * D * 2 * D, n2 = 2 * D * D;  // weights of [Wz|Wr], of Wh
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n1 + n2; t += gridDim.x * blockDim.x) {
    const bool first = t < n1;
    const int e = first ? t : t - n1, ncol = first ? 2 * D : D;
    // e = ((u * 4 + kq) * ncol + col) * 8 + j   (plane-independent index inside a slice's plane)
    const int j = e & 7, col = (e >> 3) % ncol, kq = ((e >> 3) / ncol) & 3, u = ((e >> 3) / ncol) >> 2;
    const int k = 32 * u + 8 * kq + j;
    const float w = first ? (col < D ? Wz[k * D + col] : Wr[k * D + col - D]) : Wh[k * D + col];
    const unsigned u0 = __float_as_uint(w) & 0xffff0000u;
    const float r1 = w - __uint_as_float(u0);
    const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(u1);
    const unsigned u2 = __float_as_uint(r2);  // <= 8 significant bits left: its low half is zero
    const size_t plane = (size_t)4 * ncol * 8;                   // bf16 elements of one plane of a slice
    const size_t base = (first ? 0 : (size_t)NS * 3 * 4 * 2 * D * 8) + (size_t)u * 3 * plane + ((size_t)kq * ncol + col) * 8 + j;
    dst[base] = (unsigned short)(u0 >> 16);
    dst[base + plane] = (unsigned short)(u1 >> 16);
    dst[base + 2 * plane] = (unsigned short)(u2 >> 16);
  }
  float* vec = reinterpret_cast<float*>(dst + (size_t)NS * 3 * 4 * 3 * D * 8);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < 5 * D; t += gridDim.x * blockDim.x) {
    const int which = t / D, f = t - which * D;
    const float* v = which == 0 ? bz : which == 1 ? br : which == 2 ? bh : which == 3 ? gamma : beta;
    vec[t] = v[f];
  }
}

// Mode 3: the step's type matrices A[v] (D x D, row-major [feature][k]) once more as three bf16 planes in the operand
// order of wide_message_x3_kernel: per type 16-byte units [plane][k block][k octet][feature] of 8 consecutive k.
__global__ void wide_mat_planes_kernel(const float* __restrict__ mats, unsigned short* __restrict__ dst, int Vb, int D) {
  const int KB = D / 32;
  const size_t per = (size_t)D * D, total = (size_t)Vb * per;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t v = t / per;
    const int e = (int)(t - v * per), f = e / D, k = e - f * D;
    const float w = mats[t];
    const unsigned u0 = __float_as_uint(w) & 0xffff0000u;
    const float r1 = w - __uint_as_float(u0);
    const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(u1);
    const unsigned u2 = __float_as_uint(r2);  // <= 8 significant bits left: its low half is zero
    const int kb = k >> 5, q = (k >> 3) & 3, j = k & 7;
    const size_t plane = (size_t)KB * 4 * D * 8;  // bf16 elements of one plane of a type
    const size_t base = v * 3 * plane + (((size_t)kb * 4 + q) * D + f) * 8 + j;
    dst[base] = (unsigned short)(u0 >> 16);
    dst[base + plane] = (unsigned short)(u1 >> 16);
    dst[base + 2 * plane] = (unsigned short)(u2 >> 16);
  }
}

}  // namespace wide

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
bool encoder_wide_supported(int N, int E, int D, int K, int S, int Vb) {
  using namespace wide;
  return (D == 64 || D == 128) && K >= 1 && S >= 0 && N >= 1 && N <= kMaxN && E >= 0 && E <= kMaxE && Vb >= 1 &&
         Vb <= kMaxVb;
}

size_t encoder_wide_workspace_bytes(int n_ions, int B, int N, int E, int D, int S, int Vb, bool x3) {
  return wide::ws_layout(n_ions, B, N, E, D, S, Vb, x3).total;
}

size_t encoder_wide_prepared_bytes(int D, int S, int Vb, bool x3) { return wide::prepared_bytes(D, S, Vb, x3); }

int launch_encoder_wide_prepare(const float* weights, const float* bond_table, int D, int K, int S, int Vb, bool x3,
                                void* prepared, hipStream_t s) {
  using namespace wide;
  float* img = static_cast<float*>(prepared);
  const size_t canon = (size_t)impnn_encoder_step_floats(D, K);
  for (int st = 0; st < S; ++st) {
    const float* w = weights + (size_t)st * canon;
    float* dst = img + (size_t)st * step_floats(D, Vb, x3);
    if (int rc = launch_bond_type_matrices(bond_table, w, dst, Vb, K, D, s)) return rc;
    if (x3) {
      wide_gu_image_x3_kernel<<<128, 256, 0, s>>>(w + (size_t)K * D * D,
                                                  reinterpret_cast<unsigned short*>(dst + (size_t)Vb * D * D), D);
      wide_mat_planes_kernel<<<512, 256, 0, s>>>(dst, reinterpret_cast<unsigned short*>(dst + mat_planes_off(D, Vb)), Vb, D);
    }
    else
      wide_gu_image_kernel<<<64, 256, 0, s>>>(w + (size_t)K * D * D, dst + (size_t)Vb * D * D, D);
    if (int rc = check_launch("encoder_wide_prepare")) return rc;
  }
  return IMPNN_OK;
}

namespace {
// Dynamic-LDS opt-in above 64 KB, once per (kernel, device).
template <int SLOT, typename K>
int raise_lds(K kern, size_t bytes) {
  if (bytes <= 64 * 1024) return IMPNN_OK;
  static std::atomic<uint64_t> done{0};  // one instance per SLOT = per kernel instantiation
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const uint64_t bit = 1ull << dev;
  if (done.load(std::memory_order_acquire) & bit) return IMPNN_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(IMPNN_E_LAUNCH, "encoder_wide: cannot raise the LDS limit: %s", hipGetErrorString(e));
  done.fetch_or(bit, std::memory_order_release);
  return IMPNN_OK;
}
}  // namespace

// May the update kernels name messages directly as a row's sources (32-bit float offsets from `agg`)?  Not where the
// message buffer lies beyond that range (~70 000 pairs at D = 128); IMPNN_WIDE_NO_DIRECT=1 (diagnostics, read once per
// process) forces the other path - every row's sum in agg - for the tests.
bool wide_direct_sources(const wide::Ws& w, int D) {
  static const bool forced_off = [] {
    const char* e = getenv("IMPNN_WIDE_NO_DIRECT");
    return e && atoi(e) != 0;
  }();
  return !forced_off && (int64_t)((w.m - w.agg) / 4) + (int64_t)w.vmax * D < ((int64_t)1 << 31);
}

int launch_encoder_wide(const EncoderArgs& a, hipStream_t s) {
  using namespace wide;
  const bool x3 = a.mode == 3;
  const Ws w = ws_layout(a.n_ions, a.B, a.N, a.E, a.D, a.S, a.Vb, x3);
  if (!aligned16(a.workspace)) return fail(IMPNN_E_BADARG, "encoder_fused: workspace must be 16B aligned");
  if (w.vmax >= INT_MAX || w.rmax >= INT_MAX)  // sorted positions and compact rows are 32-bit indices
    return fail(IMPNN_E_UNSUPPORTED, "encoder_fused: batch of %d pairs x (N=%d, E=%d) exceeds 32-bit row / edge indices",
                a.B, a.N, a.E);
  char* base = static_cast<char*>(a.workspace);
  auto I = [&](size_t off) { return reinterpret_cast<int32_t*>(base + off); };
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  Inputs in{};
  for (int g = 0; g < a.n_ions; ++g) {
    in.atom_ids[g] = a.atom_ids[g];
    in.bond_ids[g] = a.bond_ids[g];
    in.conn[g] = a.conn[g];
  }
  in.n_ions = a.n_ions; in.B = a.B; in.N = a.N; in.E = a.E; in.Va = a.Va; in.Vb = a.Vb;
  const int mols = a.n_ions * a.B;
  in.mpw = mols <= 1024 ? 1 : kMolPerWg / 4;
  const int mol_wgs = (mols + 4 * in.mpw - 1) / (4 * in.mpw);
  const int te = tile_edges(a.D);
  if (a.phases & 1) {
    const int nz = (int)((w.kept - w.meta) / 4);  // meta and the type counters
    const bool direct_ok = wide_direct_sources(w, a.D);
    wide_iota_kernel<<<(unsigned)(((w.rmax > nz ? w.rmax : nz) + 255) / 256), 256, 0, s>>>(I(w.aggc2), I(w.aggc2) + w.rmax, F(w.agg),
        (int)w.rmax, a.D, I(w.meta), nz);
    wide_count_kernel<<<mol_wgs, 256, 0, s>>>(in, I(w.kept), I(w.cnt));
    wide_scan_kernel<<<1, 1024, 0, s>>>(I(w.kept), I(w.rowbase), I(w.cnt), I(w.tstart), I(w.cursor), I(w.tilebase),
                                        I(w.srcrow), I(w.meta), a.n_ions, a.B, w.nT, te);
    wide_place_kernel<<<mol_wgs, 256, 0, s>>>(in, I(w.kept), I(w.rowbase), I(w.cursor), I(w.srcrow),
                                              reinterpret_cast<int2*>(base + w.rowinfo), I(w.csr), I(w.aggc2),
                                              I(w.aggc2) + w.rmax, (int)w.rmax, direct_ok ? 1 : 0);
    if (int rc = check_launch("encoder_wide plan")) return rc;
  }
  if (!(a.phases & 2)) return IMPNN_OK;
  if (!aligned16(a.atom_table)) return fail(IMPNN_E_BADARG, "encoder_fused: atom_table must be 16B aligned");
  const float* img[2] = {nullptr, nullptr};
  for (int g = 0; g < a.n_ions && a.S > 0; ++g) {
    if (a.prepared[g]) {
      if (!aligned16(a.prepared[g])) return fail(IMPNN_E_BADARG, "encoder_fused: prepared weights must be 16B aligned");
      img[g] = static_cast<const float*>(a.prepared[g]);
    } else {
      float* dst = reinterpret_cast<float*>(base + w.img + (size_t)g * prepared_bytes(a.D, a.S, a.Vb, x3));
      if (int rc = launch_encoder_wide_prepare(a.weights[g], a.bond_table, a.D, a.K, a.S, a.Vb, x3, dst, s)) return rc;
      img[g] = dst;
    }
  }
  if (a.n_ions == 1) img[1] = img[0];
  profile_record_start(s);
  wide_embed_kernel<<<mols, 256, 0, s>>>(in, I(w.kept), I(w.rowbase), a.atom_table, F(w.h), a.D);
  const int cus = device_compute_units();  // persistent message workgroups: one per CU
  const size_t msg_lds = ((size_t)a.D * (a.D + 4) + 2 * (size_t)te * (a.D + 4)) * 4 + (size_t)(w.nT + 1) * 4;
  const int nt = a.D / 16;
  constexpr int R = kRT;
  const size_t gu_lds = x3 ? gu_x3_lds_bytes(a.D) : gu_lds_floats(a.D) * 4;
  const size_t gu_lds_big = gu_x3b_lds_bytes(a.D);
  // mode 3: the messages on the bf16 pipe too (the choice depends on the shape only, so a batch and its shards run the
  // same kernels)
  const bool x3_msg = x3 && ((nt == 8 && te == 64) || (nt == 4 && te == 128));
  const size_t msg_x3_lds = msg_x3_lds_bytes(a.D, te, w.nT);
  if (x3_msg)
    if (int rc = nt == 8 ? raise_lds<8>(wide_message_x3_kernel<8, 64>, msg_x3_lds)
                         : raise_lds<9>(wide_message_x3_kernel<4, 128>, msg_x3_lds))
      return rc;
  if (a.D == 128) {
    if (int rc = raise_lds<0>(wide_message_kernel<8, 64>, msg_lds)) return rc;
    if (int rc = x3 ? raise_lds<4>(wide_update_x3_kernel<8>, gu_lds) : raise_lds<1>(wide_update_kernel<8>, gu_lds)) return rc;
    if (x3)
      if (int rc = raise_lds<6>(wide_update_x3b_kernel<8>, gu_lds_big)) return rc;
  } else {
    if (int rc = raise_lds<2>(wide_message_kernel<4, 128>, msg_lds)) return rc;
    if (int rc = x3 ? raise_lds<5>(wide_update_x3_kernel<4>, gu_lds) : raise_lds<3>(wide_update_kernel<4>, gu_lds)) return rc;
    if (x3)
      if (int rc = raise_lds<7>(wide_update_x3b_kernel<4>, gu_lds_big)) return rc;
  }
  const int64_t red_threads = w.rmax * (a.D / 4);
  // update tiles: kRT rows, or 16 rows for batches of up to ~100 pairs (the kept rows are only known on the device: the
  // choice goes by the upper bound mols * N)
  // (only while the smaller tiles still fit one round of two workgroups per CU: a tile's cost is mostly its 48 weight
  //  slices and barriers, not its rows - at 256 pairs 16-row tiles took 666 us per forward against 526 us)
  const int64_t max_tiles = ((int64_t)mols * a.N + R - 1) / R;
  int tile_rows = 4 * max_tiles <= 2 * cus ? 16 : R;  // (32-row tiles: 574 us at 200 pairs against ~500 us)
  {  // diagnostics override, read once per process (impnn.h: results depend on the arguments only)
    static const int env_rows = [] {
      const char* e = getenv("IMPNN_WIDE_TILE_ROWS");
      return e ? atoi(e) : 0;
    }();
    if (env_rows) tile_rows = env_rows == 16 ? 16 : (env_rows == 32 ? 32 : R);
  }
  const int gu_grid = (int)(w.rmax / tile_rows);
  // mode 3: 128-row tiles once they fill the chip at one workgroup per CU (two rounds and more)
  bool big_tiles = x3 && tile_rows == R && (int64_t)mols * a.N >= (int64_t)2 * cus * kRT3;
  const int64_t m_off = (int64_t)((w.m - w.agg) / 4);
  {
    static const int env_big = [] {
      const char* e = getenv("IMPNN_WIDE_X3_BIG");
      return e ? atoi(e) : -1;
    }();
    if (env_big >= 0) big_tiles = x3 && env_big != 0;
  }
  // (the update kernels address a row's two sources as 32-bit float offsets from `agg`: batches whose message buffer
  //  lies beyond that range - ~70 000 pairs at D = 128 - keep every row's sum in agg)
  unsigned long long* stamps = nullptr;  // [gu_grid x 8 | cus x 8] words, the last step's launches win
  {
    size_t sb = 0;
    void* sp = debug_stamp_buffer(&sb);
    if (sp && sb >= ((size_t)gu_grid + cus) * 8 * sizeof(unsigned long long)) stamps = static_cast<unsigned long long*>(sp);
  }
  for (int stp = 0; stp < a.S; ++stp) {
    const size_t step_off = (size_t)stp * step_floats(a.D, a.Vb, x3);
    MsgParams mp{};
    mp.h = F(w.h); mp.m = F(w.m);
    mp.img[0] = img[0]; mp.img[1] = img[1];
    mp.mat_off = step_off;
    mp.srcrow = I(w.srcrow); mp.tilebase = I(w.tilebase); mp.meta = I(w.meta);
    mp.nT = w.nT; mp.Vb = a.Vb;
    mp.stamps = stamps ? stamps + (size_t)gu_grid * 8 : nullptr;
    mp.planes_off = step_off + mat_planes_off(a.D, a.Vb);
    if (a.E > 0 && x3_msg) {
      if (nt == 8) wide_message_x3_kernel<8, 64><<<cus, kMsgX3Threads, msg_x3_lds, s>>>(mp);
      else wide_message_x3_kernel<4, 128><<<cus, kMsgX3Threads, msg_x3_lds, s>>>(mp);
    } else if (a.E > 0) {
      if (nt == 8) wide_message_kernel<8, 64><<<cus, 1024, msg_lds, s>>>(mp);
      else wide_message_kernel<4, 128><<<cus, 1024, msg_lds, s>>>(mp);
    }
    wide_reduce_kernel<<<(unsigned)((red_threads + 255) / 256), 256, 0, s>>>(
        F(w.m), reinterpret_cast<const int2*>(base + w.rowinfo), I(w.csr), F(w.agg), I(w.meta), a.n_ions, a.D,
        wide_direct_sources(w, a.D) ? 2 : 0);
    GuParams gp{};
    gp.h = F(w.h); gp.agg = F(w.agg);
    gp.c2a = I(w.aggc2); gp.c2b = I(w.aggc2) + w.rmax; gp.m_off = (int)m_off;
    gp.img[0] = img[0]; gp.img[1] = img[1];
    gp.gu_off = step_off + (size_t)a.Vb * a.D * a.D;
    gp.meta = I(w.meta); gp.eps = a.ln_eps; gp.n_ions = a.n_ions; gp.tile_rows = tile_rows;
    gp.stamps = stamps;
    if (x3 && big_tiles) {  // batches that fill the chip: 128-row tiles
      gp.cus = cus;
      gp.tiles_max = (int)(w.rmax / kRT3);
      const int grid = gp.tiles_max + 8 * (cus - 1);
      if (nt == 8) wide_update_x3b_kernel<8><<<grid, kGuX3Threads, gu_lds_big, s>>>(gp);
      else wide_update_x3b_kernel<4><<<grid, kGuX3Threads, gu_lds_big, s>>>(gp);
    } else if (x3) {
      if (nt == 8) wide_update_x3_kernel<8><<<gu_grid, kGuX3Threads, gu_lds, s>>>(gp);
      else wide_update_x3_kernel<4><<<gu_grid, kGuX3Threads, gu_lds, s>>>(gp);
    } else if (tile_rows == 16) {  // launches too small to fill the chip
      if (nt == 8) wide_update_small_kernel<8><<<gu_grid, 512, 0, s>>>(gp);
      else wide_update_small_kernel<4><<<gu_grid, 256, 0, s>>>(gp);
    } else if (nt == 8) {
      wide_update_kernel<8><<<gu_grid, kGuThreads, gu_lds, s>>>(gp);
    } else {
      wide_update_kernel<4><<<gu_grid, kGuThreads, gu_lds, s>>>(gp);
    }
  }
  const int64_t pool_threads = (int64_t)mols * (a.D / 4);
  wide_pool_kernel<<<(unsigned)((pool_threads + 255) / 256), 256, 0, s>>>(in, I(w.kept), I(w.rowbase), F(w.h),
                                                                         a.pooled[0], a.n_ions > 1 ? a.pooled[1] : nullptr,
                                                                         a.D);
  profile_record_stop(s);
  return check_launch("encoder_wide");
}

}  // namespace impnn
