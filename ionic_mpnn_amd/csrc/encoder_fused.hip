// Fused message-passing encoder for gfx950 (MI355X): encode() of train_viscosity.py:166-187 up to
// and including GlobalSumPool, for both ions in one launch, D = 32, K <= 8.
//
// Shape of the computation:
//   * molecules are cut into CHUNKS of <= 256 packed atom rows (padding atoms are not carried;
//     see "rows" below).  The graph-only preparation of a chunk - in-degrees, placement of rows by
//     descending in-degree, in-edge lists - is done by plan kernels (encoder_plan.hip) that write an
//     8 KB chunk record; this kernel copies the record into LDS;
//   * one 1024-thread workgroup per compute unit, PERSISTENT: the plan deals the rows of the batch
//     to the workgroups in equal contiguous shares, each cut into chunks by next-fit, so every CU
//     carries the same number of 16-atom tiles (no wave of leftover workgroups at the end);
//   * per chunk: 16 waves, one 16-atom tile each, all S steps with the node state h in LDS
//     (double-buffered) and the weights of the current (ion, step) in LDS;
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
//     in-edges in edge-slot order (deterministic), forming
//         G[k][j] = sum_{e -> atom} bond_table[bond_id_e][k] * h[src_e][j]
//     in registers, then agg^T = sum_k W_k * G_k on the matrix cores;
//   * GatedUpdate (models/layers.py:142-156): sigmoid/tanh/LayerNorm on the accumulator registers,
//     LayerNorm row reduction = 8 in-lane adds + 2 cross-lane steps;
//   * GlobalSumPool (models/layers.py:161-164): segmented sum over the chunk's rows from LDS.
//
// rows: molecule b keeps rows [0, r_b), r_b = 1 + max(last n with atom_ids[b,n] > 0, largest atom
// index on a valid edge).  Rows >= r_b can never send (no valid edge names them) and are masked
// by the pool, so dropping them cannot change the output; the kept set is closed under
// "is a source of", which makes the skip exact, not approximate.
#include <atomic>
#include <cstdlib>

#include "encoder_device.h"
#include "encoder_layout.h"

namespace impnn {
namespace enc {

namespace {

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

// image | h | chunk record | bond table (Vb*8 floats, sized at launch so that the plan
// kernels of the next batch still find LDS on the same CU)
constexpr size_t kLdsFixedBytes = sizeof(float) * ((size_t)kImgSlot + kRCap * kHS) + kRecBytes;
static_assert(kLdsFixedBytes + sizeof(float) * kTbCapFloats <= 160 * 1024, "LDS budget");

// KT = compile-time bond_dim (0: run-time K <= 8); SPLIT: mode 1 (fp16 hi/lo products)
template <int KT, bool SPLIT>
__global__ __launch_bounds__(kThreads, kThreads / 256) void encoder_fused_kernel(EncParams p) {
  extern __shared__ __align__(16) float smem[];
  const int K = KT ? KT : p.K;
  float* const wimg = smem;
  float* const hbuf = wimg + kImgSlot;  // node state, updated in place (see the step loop)
  unsigned char* const recl = reinterpret_cast<unsigned char*>(hbuf + kRCap * kHS);
  float* const tbl = reinterpret_cast<float*>(recl + kRecBytes);
  const uint16_t* const r_rowptr = reinterpret_cast<const uint16_t*>(recl + kRecRowptr);
  const unsigned char* const r_tilemax = recl + kRecTilemax;
  const uint16_t* const r_moloff = reinterpret_cast<const uint16_t*>(recl + kRecMoloff);
  const uint16_t* const r_molrows = reinterpret_cast<const uint16_t*>(recl + kRecMolrows);
  const uint16_t* const r_poolrow = reinterpret_cast<const uint16_t*>(recl + kRecPoolrow);
  const int32_t* const r_rowatom = reinterpret_cast<const int32_t*>(recl + kRecRowatom);
  const uint32_t* const r_ent = reinterpret_cast<const uint32_t*>(recl + kRecEnt);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = lane & 15, q = lane >> 4;
  unsigned long long* stamp = p.stamps ? p.stamps + (size_t)blockIdx.x * 32 : nullptr;
  if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memtime();
  {  // the workspace must hold a plan made for this launch geometry (see encoder_typed.hip)
    const PlanHeader hd = *p.header;
    if (hd.magic != kPlanMagic || hd.kind != 0 || hd.nwg != (int)gridDim.x || hd.max_sub != p.max_sub ||
        hd.B != p.B || hd.n_ions != p.n_ions) {
      const float nan = __builtin_nanf("");
      for (int g = 0; g < p.n_ions; ++g)
        for (int64_t t = (int64_t)blockIdx.x * kThreads + tid; t < (int64_t)p.B * kD; t += (int64_t)gridDim.x * kThreads)
          p.pooled[g][t] = nan;
      return;
    }
  }
  const int c_begin = blockIdx.x * p.max_sub;
  const int c_end = c_begin + __builtin_amdgcn_readfirstlane(p.nsub[blockIdx.x]);
  if (c_begin >= c_end) return;

  // bond table copy (mode 1: pre-scaled, so G comes out in the B-operand scale)
  for (int t = tid; t < p.Vb * kKMax; t += kThreads) {
    const int v = t >> 3, k = t & 7;
    tbl[t] = k < K ? p.bond_table[v * K + k] * (SPLIT ? kSX : 1.0f) : 0.f;
  }
  // atom table copy: a chunk prologue then fills h0 from LDS instead of waiting for a global round trip
  float* const atab = tbl + ((p.Vb * kKMax + 127) & ~127);  // Va rows + one zero row (slack / out-of-range ids)
  if (p.atab_lds)  // rows at the h buffer's stride (kHS floats: random rows spread over the LDS banks)
    for (int t = tid; t < (p.Va + 1) * (kD / 4); t += kThreads) {
      const int r = t >> 3, c = t & 7;
      st4(atab + r * kHS + 4 * c, r < p.Va ? ld4(p.atom_table + 4 * t) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  const float* wmsg = wimg;
  const float* wupd = wimg + img_msg_floats(K);
  const float* wvec = SPLIT ? wimg + img16_vec_float_off(K) : wupd + img_upd_floats();
  const _Float16* hmsg = reinterpret_cast<const _Float16*>(wimg);
  const _Float16* hupd = hmsg + img16_msg_halfs(K);
  unsigned long long t_pro = 0, t_steps = 0, t_pool = 0, t_mark = 0;
  if (stamp && tid == 0) t_mark = __builtin_amdgcn_s_memtime();

  // The record of the NEXT chunk travels in two registers per thread while the current chunk runs;
  // the step-0 weight image of the next chunk (same ion: a share never mixes ions) is what the last
  // step's image prefetch brings in.  A chunk prologue is then one dependent load stage (atom rows).
  uint2 rec_next = reinterpret_cast<const uint2*>(p.rec + (size_t)c_begin * kRecBytes)[tid];
  int4 dsc_next = reinterpret_cast<const int4*>(p.desc)[c_begin];
  bool image_ready = false;
  for (int c = c_begin; c < c_end; ++c) {
    // ---- chunk prologue: descriptor, record -> LDS, h0 = atom_table[atom ids], step-0 weights
    const int4 dsc = dsc_next;
    // workgroup-uniform: keep them in SGPRs so every loop bound / branch below stays scalar
    const int m0 = __builtin_amdgcn_readfirstlane(dsc.x), M = __builtin_amdgcn_readfirstlane(dsc.y);
    const int rg = __builtin_amdgcn_readfirstlane(dsc.w);
    const int R = rg & 0xffff, g = rg >> 16;
    const int ntiles = (R + 15) >> 4;
    const int N = p.N;
    const int32_t* ids_g = p.atom_ids[g];
    const float* img_g = p.img[g];
    reinterpret_cast<uint2*>(recl)[tid] = rec_next;  // kRecBytes == 8 * kThreads
    {
      const int cn = (c + 1) < c_end ? (c + 1) : c;  // clamped: unconditional loads
      rec_next = reinterpret_cast<const uint2*>(p.rec + (size_t)cn * kRecBytes)[tid];
      dsc_next = reinterpret_cast<const int4*>(p.desc)[cn];
    }
    f32x4 pf[kPf];
    const bool need_image = p.S > 0 && !image_ready;  // workgroup-uniform
    if (need_image) {
#pragma unroll
      for (int i = 0; i < kPf; ++i) pf[i] = ld4(img_g + 4 * (tid + i * kThreads));
    }
    lds_barrier();
    // h0 (train_viscosity.py:171).  With the atom table in LDS, step 0 reads atom_table[id] directly (the in-edge
    // entries carry the source's atom id) and writes h1 into the buffer: nothing to fill here.
    if (!p.atab_lds || p.S == 0) {  // 4 threads per placed row, 2 x 16 B each; slack rows: zeros
      const int row = tid >> 2, sub = tid & 3;
      const int id = r_rowatom[row];
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if ((unsigned)id < (unsigned)p.Va) {
        if (p.atab_lds) {  // (two branches: one pointer for both would turn the loads into flat_load)
          v0 = ld4(atab + id * kHS + 8 * sub);
          v1 = ld4(atab + id * kHS + 8 * sub + 4);
        } else {
          v0 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub);
          v1 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub + 4);
        }
      }
      st4(hbuf + row * kHS + 8 * sub, v0);
      st4(hbuf + row * kHS + 8 * sub + 4, v1);
    }
    if (need_image) {
#pragma unroll
      for (int i = 0; i < kPf; ++i) st4(wimg + 4 * (tid + i * kThreads), pf[i]);
    }
    image_ready = p.S > 0;  // from now on the last step leaves the step-0 image behind
    lds_barrier();
    if (stamp && tid == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_pro += t - t_mark;
      t_mark = t;
    }

    // ---- message-passing steps: one 16-atom tile per wave (16 waves, 4 per SIMD)
    for (int s = 0; s < p.S; ++s) {
      // One h buffer: every wave gathers (reads other rows) BEFORE the mid-step barrier and writes its own rows
      // AFTER it, and the end-of-step barrier orders those writes before the next step's gathers.  Rows that are
      // skipped (no tile work) simply keep their value.
      const bool g0 = p.atab_lds && s == 0;             // step 0 gathers from the atom table
      // in-edge entries carry ready float4 offsets: [31:20] into the atom table, [19:8] into the h buffer
      const int src_shift = g0 ? 20 : 8;
      const int src_base = (g0 ? (int)(atab - smem) : (int)(hbuf - smem)) + 4 * q;
      const int sn = (s + 1) < p.S ? (s + 1) : 0;  // the last step fetches the step-0 image for the next chunk
      const float* nxt = img_g + (int64_t)sn * kImgSlot;
      bool pf_issued = false;

      // Tile -> wave map.  Waves w, w+4, w+8, w+12 share a SIMD.  With ntiles = 4q + r the first r
      // SIMD groups carry q+1 tiles and the rest q; tiles are ordered heavy -> light (rows are placed
      // by descending in-degree), so the q-tile groups take the heaviest tiles and the (q+1)-tile
      // groups the light ones: the busiest SIMD is not also the one with the longest gathers.
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
      const bool has_tile = my_tile >= 0 && my_tile < ntiles;
      const int tile = has_tile ? my_tile : 0;
      const int row = tile * 16 + a;
      float G[kKMax][8];
      if (has_tile) {
        // A wave's issue priority falls as it advances through its tile (gather 3, message 2, gates 1, rest 0): the SIMD
        // arbiter otherwise serves the oldest wave first, so the four waves of a SIMD finish one after
        // another and the last one runs alone, with every latency exposed (measured: -4% step time).
        __builtin_amdgcn_s_setprio(3);

        // ---- pull gather: G[k][j] = sum_{in-edges} tb[bond][k] * h[src][j]   (edge-slot order)
        const int p0 = r_rowptr[row];
        const int deg = r_rowptr[row + 1] - p0;
        const int maxdeg = __builtin_amdgcn_readfirstlane(r_tilemax[tile]);
        {
          // first in-edge initialises G (rows without in-edges use a zero coefficient vector)
          // (a row without in-edges must still read FINITE h values: 0 * NaN would poison G; entry 0 -> row 0)
          const uint32_t ent = deg > 0 ? r_ent[p0] : 0u;
          const int bid = ent & 0xffu;
          const int soff = (int)(__builtin_amdgcn_ubfe(ent, src_shift, 12) * 4u) + src_base;
          const f32x4 x0 = ld4(smem + soff);
          const f32x4 x1 = ld4(smem + soff + 16);
          f32x4 c0 = ld4(tbl + bid * kKMax);
          f32x4 c1 = ld4(tbl + bid * kKMax + 4);
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
            const uint32_t ent = r_ent[p0 + d];
            const int bid = ent & 0xffu;
            const int soff = (int)(__builtin_amdgcn_ubfe(ent, src_shift, 12) * 4u) + src_base;
            const f32x4 x0 = ld4(smem + soff);
            const f32x4 x1 = ld4(smem + soff + 16);
            const f32x4 c0 = ld4(tbl + bid * kKMax);
            const f32x4 c1 = ld4(tbl + bid * kKMax + 4);
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
        __builtin_amdgcn_s_setprio(2);
        // Mid-step barrier.  (1) h is updated in place: all gathers are done before anybody writes.  (2) The image
        // of this step was stored after the previous step's barrier, without a barrier of its own: the gather
        // above needs no weights, so those LDS stores ran under it.  From here on they are needed.
        // (Waves without a tile meet this barrier in the else branch below: whole waves take either path.)
        if (!g0) lds_barrier();  // (step 0 from the atom table: nobody reads the buffer, the image is in place)
        int own = row;            // the row's own state
        if (g0) {
          const int id = r_rowatom[row];
          own = (unsigned)id < (unsigned)p.Va ? id : p.Va;
        }
        own = own * kHS + src_base;
        const f32x4 h0 = ld4(smem + own);
        const f32x4 h1 = ld4(smem + own + 16);

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
          agg0 *= (kSX / kAcc);
          agg1 *= (kSX / kAcc);
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
        __builtin_amdgcn_s_setprio(1);
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
          const f32x4 hs0 = h0 * kSX, hs1 = h1 * kSX;
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
          z0 = sigmoid4<true>(z0);  // accumulators carry kAcc: folded into the exp2 constant
          z1 = sigmoid4<true>(z1);
          const f32x4 rs0 = sigmoid4<true>(r0) * hs0, rs1 = sigmoid4<true>(r1) * hs1;  // :149, times kSX
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
          z0 = sigmoid4<false>(z0);
          z1 = sigmoid4<false>(z1);
          rh0 = sigmoid4<false>(r0) * h0;  // :149
          rh1 = sigmoid4<false>(r1) * h1;
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
        __builtin_amdgcn_s_setprio(0);
        // ---- blend, LayerNorm, residual  (models/layers.py:153-155)
        // (1-z) h + z t == h + z (t - h); vector arithmetic throughout (packed f32 instructions)
        f32x4 n0 = z0 * (tanh4<SPLIT>(t0) - h0) + h0;
        f32x4 n1 = z1 * (tanh4<SPLIT>(t1) - h1) + h1;
        const f32x4 s4 = n0 + n1;
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum * (1.0f / kD);
        n0 -= mean;
        n1 -= mean;
        const f32x4 q4 = n0 * n0 + n1 * n1;
        float var = (q4[0] + q4[1]) + (q4[2] + q4[3]);
        var += __shfl_xor(var, 16);
        var += __shfl_xor(var, 32);
        const float inv = __builtin_amdgcn_rsqf(var * (1.0f / kD) + p.ln_eps);
        const f32x4 g0 = ld4(wvec + 3 * kD + 4 * q), g1 = ld4(wvec + 3 * kD + 16 + 4 * q);
        const f32x4 b0 = ld4(wvec + 4 * kD + 4 * q), b1 = ld4(wvec + 4 * kD + 16 + 4 * q);
        const f32x4 o0 = n0 * (g0 * inv) + (b0 + h0);
        const f32x4 o1 = n1 * (g1 * inv) + (b1 + h1);
        st4(hbuf + row * kHS + 4 * q, o0);
        st4(hbuf + row * kHS + 16 + 4 * q, o1);

      }
      if (!pf_issued) {  // waves without a tile in this chunk still carry their share of the image
        if (!g0) lds_barrier();
#pragma unroll
        for (int i = 0; i < kPf; ++i) pf[i] = ld4(nxt + 4 * (tid + i * kThreads));
      }
      const bool wstamp = stamp && c == c_begin && s == 1;  // diagnostics: when each wave reaches the step barrier
      if (wstamp && lane == 0) stamp[16 + wave] = __builtin_amdgcn_s_memtime();
      __syncthreads();
      if (wstamp && tid == 0) stamp[13] = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < kPf; ++i) st4(wimg + 4 * (tid + i * kThreads), pf[i]);
      if (stamp && c == c_begin && s < 2 && tid == 0) stamp[14 + s] = __builtin_amdgcn_s_memtime();  // 14: end of step 0, 15: end of step 1
    }
    if (stamp && tid == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_steps += t - t_mark;
      t_mark = t;
    }

    // ---- GlobalSumPool (models/layers.py:161-164): rows whose atom id > 0.  Eight lanes share one
    //      (molecule, 4 features): each sums every 8th row in ascending order, then a fixed 3-step
    //      butterfly on the DPP network - a wavefront segmented reduction with a run-to-run fixed order.
    const float* hfin = hbuf;
    float* out_g = p.pooled[g];
    for (int t0 = 0; t0 < M * 64; t0 += kThreads) {
      const int t = t0 + tid;
      const int part = t & 7, f4 = (t >> 3) & 7, m = t >> 6;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        const int nr = r_molrows[m], mo = r_moloff[m];
        for (int n = part; n < nr; n += 8) {
          const int pr = r_poolrow[mo + n];
          if (pr & 0x8000) acc += ld4(hfin + (pr & 0xff) * kHS + 4 * f4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror: lane ^ 7 within 8
        acc[i] = v;
      }
      if (m < M && part == 0) st4(out_g + (int64_t)(m0 + m) * kD + 4 * f4, acc);
    }
    (void)N; (void)ids_g;
    lds_barrier();  // the record / h buffers are rewritten by the next chunk's prologue (pooled stores stay in flight)
    if (stamp && tid == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_pool += t - t_mark;
      t_mark = t;
    }
  }
  if (stamp && tid == 0) {
    stamp[1] = t_pro;
    stamp[2] = t_steps;
    stamp[3] = t_pool;
    stamp[4] = (unsigned long long)(c_end - c_begin);
    stamp[7] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace

}  // namespace enc

bool encoder_typed_supported(int N, int E, int D, int S, int Vb);
size_t encoder_typed_prepared_bytes(int S, int Vb, bool x3);
int launch_encoder_typed_prepare(const float* weights, const float* bond_table, int K, int S, int Vb, bool x3,
                                 void* prepared, hipStream_t s);
int launch_encoder_typed_run(const EncoderArgs& a, const enc::Ws& w, hipStream_t s);

bool encoder_fused_supported(int mode, int N, int E, int D, int K, int S, int Vb) {
  using namespace enc;
  if ((mode == 2 || mode == 3) && D != kD) return encoder_wide_supported(N, E, D, K, S, Vb);  // atom_dim 64 / 128: encoder_wide.hip
  if (mode == 2 || mode == 3) return K >= 1 && encoder_typed_supported(N, E, D, S, Vb);
  if (mode != 0 && mode != 1) return false;
  if (D != kD || K < 1 || K > kKMax || S < 0) return false;
  if (N < 1 || N > 0xffff || E < 0) return false;
  if (Vb < 1 || Vb > 0xffff || (int64_t)Vb * kKMax > kTbCapFloats) return false;
  const int vrmax = vr_max_of(N, E);
  if (vrmax > kRCap / 2) return false;  // keeps next-fit chunks at least half full; E <= 512 fits the 16-bit slot field
  return true;
}

// Compute units of the CURRENT device (cached per device index; the value never changes).
int device_compute_units() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;  // MI355X
    n = n > 1024 ? 1024 : n;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// 160 KB dynamic LDS opt-in, once per (kernel slot, device).
int ensure_lds_limit(const void* kern, int slot) {
  static std::atomic<uint64_t> done[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const uint64_t bit = 1ull << dev;
  if (done[slot].load(std::memory_order_acquire) & bit) return IMPNN_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return fail(IMPNN_E_LAUNCH, "encoder: cannot raise LDS limit: %s", hipGetErrorString(e));
  done[slot].fetch_or(bit, std::memory_order_release);
  return IMPNN_OK;
}

// Persistent workgroups of one encoder launch.  `requested` > 0: the caller's choice, clamped to [16, CUs];
// 0: the default - IMPNN_ENCODER_WORKGROUPS from the environment if set (read once per process: a diagnostics
// override, see impnn.h), else one per CU.  For very large batches or padded shapes a multiple of that, so that a
// share never holds more molecules (kShareCap) or chunks (kMaxHops) than plan_chunks resolves in LDS (the extra
// workgroups simply run in rounds).  A pure function of its arguments.
int encoder_workgroups(int n_ions, int B, int requested, int N, int E, int mode) {
  int cus = device_compute_units();
  int want = requested;
  if (want <= 0) {
    static const int env_want = [] {
      const char* e = getenv("IMPNN_ENCODER_WORKGROUPS");
      return e ? atoi(e) : 0;
    }();
    want = env_want;
  }
  if (want > 0) cus = want < 16 ? 16 : (want > cus ? cus : want);
  int f = 1;
  while ((int64_t)2 * n_ions * B / ((int64_t)cus * f) + 64 > enc::kShareCap ||
         enc::ws_layout(n_ions, B, N, E, 1, 1, cus * f, mode >= 2, mode == 3).max_sub > enc::kMaxHops)
    ++f;
  return cus * f;
}

size_t encoder_fused_workspace_bytes(int mode, int n_ions, int B, int N, int E, int D, int S, int Vb, int nwg) {
  if (D != enc::kD) return encoder_wide_workspace_bytes(n_ions, B, N, E, D, S, Vb, mode == 3);
  return enc::ws_layout(n_ions, B, N, E, S, Vb, nwg, mode >= 2, mode == 3).total;
}

size_t encoder_prepared_bytes(int mode, int D, int S, int Vb) {
  if (D != enc::kD) return encoder_wide_prepared_bytes(D, S, Vb, mode == 3);
  if (mode >= 2) return encoder_typed_prepared_bytes(S, Vb, mode == 3);
  return (size_t)(S > 0 ? S : 1) * enc::kImgSlot * sizeof(float);
}

int launch_encoder_prepare(const float* weights, const float* bond_table, int D, int K, int S, int Vb, int mode,
                           void* prepared, hipStream_t s) {
  if (S <= 0) return IMPNN_OK;
  if (D != enc::kD) return launch_encoder_wide_prepare(weights, bond_table, D, K, S, Vb, mode == 3, prepared, s);
  if (mode >= 2) return launch_encoder_typed_prepare(weights, bond_table, K, S, Vb, mode == 3, prepared, s);
  enc::ImageParams ip{};
  ip.weights = weights;
  ip.img = static_cast<float*>(prepared);
  ip.K = K;
  ip.mode = mode == 1 ? 1 : 0;
  ip.step_floats = impnn_encoder_step_floats(D, K);
  return enc::launch_weight_image(ip, S, s);
}

int launch_encoder_fused(const EncoderArgs& a, hipStream_t s) {
  if (a.D != enc::kD) return launch_encoder_wide(a, s);
  if (a.phases & 1)
    if (int rc = launch_encoder_phase(a, s, true)) return rc;
  if (a.phases & 2)
    if (int rc = launch_encoder_phase(a, s, false)) return rc;
  return IMPNN_OK;
}

// a.nwg: the resolved workgroup count (encoder_workgroups) - the same value for the plan and the run of a batch.
int launch_encoder_phase(const EncoderArgs& a, hipStream_t s, bool plan_phase) {
  using namespace enc;
  const bool typed = a.mode >= 2;
  const Ws w = ws_layout(a.n_ions, a.B, a.N, a.E, a.S, a.Vb, a.nwg, typed, a.mode == 3);
  if (!aligned16(a.workspace)) return fail(IMPNN_E_BADARG, "encoder_fused: workspace must be 16B aligned");
  if (!plan_phase && !aligned16(a.atom_table)) return fail(IMPNN_E_BADARG, "encoder_fused: atom_table must be 16B aligned");
  char* base = static_cast<char*>(a.workspace);
  const int mode = a.mode == 1 ? 1 : 0;
  PlanParams pp{};
  EncParams ep{};
  for (int g = 0; g < a.n_ions; ++g) {
    if (plan_phase && (reinterpret_cast<uintptr_t>(a.conn[g]) & 7u) != 0)
      return fail(IMPNN_E_BADARG, "encoder_fused: connectivity must be 8B aligned");
    pp.atom_ids[g] = ep.atom_ids[g] = a.atom_ids[g];
    pp.bond_ids[g] = a.bond_ids[g];
    pp.conn[g] = a.conn[g];
    ep.pooled[g] = a.pooled[g];
    if (plan_phase || typed) continue;
    if (a.prepared[g]) {
      if (!aligned16(a.prepared[g])) return fail(IMPNN_E_BADARG, "encoder_fused: prepared weights must be 16B aligned");
      ep.img[g] = static_cast<const float*>(a.prepared[g]);
    } else {  // canonical weights: build the image into the workspace first
      float* img = reinterpret_cast<float*>(base + w.img_off) + (size_t)g * (a.S > 0 ? a.S : 1) * kImgSlot;
      if (int rc = launch_encoder_prepare(a.weights[g], a.bond_table, a.D, a.K, a.S, a.Vb, mode, img, s)) return rc;
      ep.img[g] = img;
    }
  }
  pp.rows = reinterpret_cast<int32_t*>(base + w.rows_off);
  pp.vr = reinterpret_cast<int32_t*>(base + w.vr_off);
  pp.partial = reinterpret_cast<int32_t*>(base + w.partial_off);
  pp.nsub = reinterpret_cast<int32_t*>(base + w.nsub_off);
  pp.desc = reinterpret_cast<int32_t*>(base + w.desc_off);
  pp.rec = reinterpret_cast<unsigned char*>(base + w.rec_off);
  pp.header = reinterpret_cast<PlanHeader*>(base);
  pp.typed = typed ? 1 : 0;
  pp.ecap = tecap_of(a.E);
  pp.vmin = plan_vmin(a.n_ions, a.B, vr_max_of(a.N, a.E, typed), w.nwg);
  pp.n_ions = a.n_ions; pp.B = a.B; pp.N = a.N; pp.E = a.E; pp.Va = a.Va; pp.Vb = a.Vb;
  // chunk workgroups resident at once on 256 CUs: 5 per CU (pull records, <= 96 VGPRs) or 4 (typed, <= 128 VGPRs)
  {
    const int per_cu = typed ? 4 : 5;
    pp.grid_sub = w.max_sub < per_cu ? w.max_sub : per_cu;
  }
  pp.nwg = w.nwg;
  pp.max_sub = w.max_sub;
  pp.nblk = w.nblk;
  pp.stamps = nullptr;
  {
    size_t sb = 0;
    void* sp = debug_stamp_buffer(&sb);
    if (sp && sb >= ((size_t)w.nwg * 32 + 16) * sizeof(unsigned long long))
      pp.stamps = static_cast<unsigned long long*>(sp) + (size_t)w.nwg * 32;
  }
  if (plan_phase) return launch_plan(pp, s);
  if (typed) return launch_encoder_typed_run(a, w, s);

  ep.atom_table = a.atom_table;
  ep.bond_table = a.bond_table;
  ep.nsub = pp.nsub; ep.desc = pp.desc; ep.rec = pp.rec; ep.max_sub = w.max_sub;
  ep.header = pp.header;
  ep.n_ions = a.n_ions; ep.B = a.B; ep.N = a.N; ep.K = a.K; ep.S = a.S;
  ep.Va = a.Va; ep.Vb = a.Vb; ep.ln_eps = a.ln_eps;
  ep.stamps = nullptr;
  {
    size_t sb = 0;
    void* sp = debug_stamp_buffer(&sb);
    if (sp && sb >= (size_t)w.nwg * 32 * sizeof(unsigned long long)) ep.stamps = static_cast<unsigned long long*>(sp);
  }
  const int variant = (a.K == 8 ? 1 : 0) + 2 * mode;
  void (*kern)(EncParams) = variant == 0   ? encoder_fused_kernel<0, false>
                            : variant == 1 ? encoder_fused_kernel<8, false>
                            : variant == 2 ? encoder_fused_kernel<0, true>
                                           : encoder_fused_kernel<8, true>;
  if (int rc = ensure_lds_limit((const void*)kern, variant)) return rc;
  profile_record_start(s);
  size_t lds = kLdsFixedBytes + align_up((size_t)a.Vb * kKMax * sizeof(float), 512);
  const size_t atab_bytes = ((size_t)a.Va + 1) * kHS * sizeof(float);
  ep.atab_lds = a.Va <= kEntMaxAtom && lds + atab_bytes <= 156 * 1024;
  if (ep.atab_lds) lds += atab_bytes;
  kern<<<w.nwg, kThreads, lds, s>>>(ep);
  profile_record_stop(s);
  return check_launch("encoder_fused");
}

}  // namespace impnn
