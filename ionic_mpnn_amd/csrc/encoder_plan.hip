// Plan kernels of the fused encoder (gfx950): everything that depends on the graph batch only
// (not on the weights) and is cheap, latency-bound, index arithmetic:
//
//   plan_stats   one wave per molecule: kept rows r_b, valid edges v_b -> virtual rows; one partial
//                sum per 16 molecules
//   plan_chunks  one 256-thread workgroup per (share, chunk slot) (8+ resident per CU, so its
//                dependent loads overlap).  The rows of the batch are dealt to `nwg` persistent
//                encoder workgroups in equal contiguous shares (per ion, proportional to its rows);
//                the workgroup resolves its share from the partial sums, the share's molecules and
//                its next-fit chain of chunks (<= 256 rows / <= 1024 edges), then builds its chunk: in-degrees,
//                placement of rows by descending in-degree, CSR of in-edges in edge-slot order,
//                pool map -> one 8 KB chunk record in HBM
//   weight_image canonical weights -> the encoder's LDS image (weights only; run when they change)
#include "encoder_layout.h"

namespace impnn {
namespace enc {

namespace {

// -----------------------------------------------------------------------------------------
// plan_stats: one wave per (ion, molecule): kept rows r_b, valid edges v_b, virtual rows
// vr_b = max(1, r_b, ceil(v_b/4)).  Extra blocks convert the canonical packed step weights into
// the LDS image the encoder copies verbatim (message rows padded to 36, gate kernels transposed).
// -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kPB) void plan_stats_kernel(PlanParams p) {
  __shared__ int vsum[kPB], vbad[kPB];
  // Plan kernels are short chains of dependent loads; when they run beside the issue-bound encoder of
  // the previous batch (pipelined callers) they must not queue behind its waves for every instruction.
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = blockIdx.x / p.nblk, blk = blockIdx.x - g * p.nblk;
  const int b = blk * kPB + wv;
  const bool have = b < p.B;
  const int64_t item = (int64_t)g * p.B + (have ? b : 0);
  int my_vr = 0, my_bad = 0;
  if (have) {
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
  {
    // edges per virtual row: 4 (pull records) or ecap / 256 = 2, 2.5 (typed)
    int vr = p.typed ? tvr_of_edges(cnt < 0x100000 ? cnt : 0x100000, p.ecap) : (cnt + 3) >> 2;
    vr = vr > rmax ? vr : rmax;
    my_vr = vr < 1 ? 1 : vr;
    // a molecule that does not fit one chunk (kRCap rows / kRCap virtual rows of edges): the plan is marked as
    // overflowed (kPlanBadBit of the block's partial sum -> plan_chunks -> PlanHeader::overflow), nothing is built
    if (my_vr > kRCap) {
      my_vr = kRCap;
      my_bad = 1;
    }
    if (lane == 0) {
      p.rows[item] = rmax;
      p.vr[item] = my_vr;
    }
  }
  }
  if (lane == 0) {
    vsum[wv] = my_vr;
    vbad[wv] = my_bad;
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    int t = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < kPB; ++i) {
      t += vsum[i];
      bad |= vbad[i];
    }
    p.partial[(int64_t)g * p.nblk + blk] = t | (bad ? kPlanBadBit : 0);  // t <= kPB * kRCap = 4096
    if (blockIdx.x == 0) {  // what this plan was made for: the encoder refuses a workspace planned differently
      PlanHeader h;
      h.magic = kPlanMagic; h.kind = p.typed; h.n_ions = p.n_ions; h.B = p.B; h.N = p.N; h.E = p.E;
      h.nwg = p.nwg; h.max_sub = p.max_sub;
      h.overflow = 0;  // raised by plan_chunks (the next kernel on the stream)
      *p.header = h;
    }
  }
}

// -----------------------------------------------------------------------------------------
// weight_image: canonical packed step weights -> the image the encoder copies verbatim into LDS
// (mode 0: f32, message rows padded to 36, gate kernels transposed; mode 1: fp16 hi/lo blocks in
// MFMA A-operand order).  grid = (slices, steps); depends on the weights only, so callers that
// keep weights fixed run it once (impnn_encoder_prepare_weights).
// -----------------------------------------------------------------------------------------

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
// typed_image: step `s` of the typed encoder's prepared buffer.  The canonical per-bond-type matrices
// A[v] = sum_k bond_table[v,k] W[k] of the step (models/layers.py:108, computed by launch_bond_type_matrices
// into the buffer's scratch area) are re-laid in the B-operand order of v_mfma_f32_4x4x1 (encoder_layout.h),
// and the GatedUpdate kernels of the step in the MFMA A-operand order the update GEMMs read from LDS.
// -----------------------------------------------------------------------------------------
__global__ void typed_image_kernel(TImageParams p, int s) {
  const int t_begin = blockIdx.x * blockDim.x + threadIdx.x, t_stride = gridDim.x * blockDim.x;
  const int S = p.S > 0 ? p.S : 1;
  const size_t uslot = p.x3 ? kXUpdSlot : kTUpdSlot;
  float* upd = p.prepared + (size_t)s * uslot;
  float* tmat = p.prepared + (size_t)S * uslot + (size_t)s * p.Vb * kTMatFloats;
  const float* canon = p.prepared + (size_t)S * uslot + (size_t)S * p.Vb * kTMatFloats;
  const float* w = p.weights + (int64_t)s * p.step_floats;
  const float* Wz = w + (int64_t)p.K * kD * kD;  // (64,32)
  const float* bz = Wz + 2 * kD * kD;
  const float* Wr = bz + kD;
  const float* br = Wr + 2 * kD * kD;
  const float* Wh = br + kD;
  const float* bh = Wh + 2 * kD * kD;
  const float* gamma = bh + kD;
  const float* beta = gamma + kD;
  const int nmat = p.Vb * kTMatFloats;
  for (int t = t_begin; t < nmat; t += t_stride) {
    // destination index: v*1024 + kq*128 + r*4 + c   <-  A[v][r][4*kq + c]
    const int v = t >> 10, rem = t & 1023, kq = rem >> 7, r = (rem >> 2) & 31, c = rem & 3;
    tmat[t] = canon[(size_t)v * kTMatFloats + r * kD + 4 * kq + c];
  }
  if (p.x3) {  // mode 3: the gate kernels as three bf16 planes (w = b0 + b1 + b2 exactly), MFMA A-operand order
    unsigned short* planes = reinterpret_cast<unsigned short*>(upd);
    for (int t = t_begin; t < kXUpdHalfs / 3; t += t_stride) {
      // t = (blk * 64 + lane) * 8 + j,  blk = (gate*2 + T)*2 + half; value W_gate[(32 half + feat_of(q, j)) * 32 + 16 T + i]
      const int j = t & 7, ln = (t >> 3) & 63, blk = t >> 9;
      const int q = ln >> 4, i = ln & 15, f = feat_of(q, j);
      const int half = blk & 1, T = (blk >> 1) & 1, gate = blk >> 2;
      const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
      const float wv = Wg[(int64_t)(half * kD + f) * kD + 16 * T + i];
      const unsigned u0 = __float_as_uint(wv) & 0xffff0000u;
      const float r1 = wv - __uint_as_float(u0);
      const unsigned u1 = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(u1);
      const unsigned u2 = __float_as_uint(r2);  // <= 8 significant bits left: its low half is zero
      const int base = (blk * 3) * 512 + ln * 8 + j;
      planes[base] = (unsigned short)(u0 >> 16);
      planes[base + 512] = (unsigned short)(u1 >> 16);
      planes[base + 1024] = (unsigned short)(u2 >> 16);
    }
    for (int t = t_begin; t < kXUpdSlot - kXVecFloatOff; t += t_stride) {
      float val = 0.f;
      if (t < 5 * kD) {
        const int v = t / kD, i = t - v * kD;
        const float* src = v == 0 ? bz : v == 1 ? br : v == 2 ? bh : v == 3 ? gamma : beta;
        val = src[i];
      }
      upd[kXVecFloatOff + t] = val;
    }
    return;
  }
  for (int t = t_begin; t < kTUpdSlot; t += t_stride) {
    float val = 0.f;
    if (t < kTVecFloatOff) {
      // A-operand order of v_mfma_f32_16x16x4_f32 (encoder_layout.h): t = (blk * 64 + lane) * 4 + r,
      // blk = ((gate * 2 + T) * 2 + half) * 2 + u
      const int r = t & 3, ln = (t >> 2) & 63, blk = t >> 8;
      const int a = ln & 15, q = ln >> 4;
      const int u = blk & 1, half = (blk >> 1) & 1, T = (blk >> 2) & 1, gate = blk >> 3;
      const float* Wg = gate == 0 ? Wz : (gate == 1 ? Wr : Wh);
      val = Wg[(int64_t)(32 * half + 16 * u + 4 * q + r) * kD + 16 * T + a];
    } else if (t < kTVecFloatOff + 5 * kD) {
      const int v = (t - kTVecFloatOff) / kD, i = (t - kTVecFloatOff) - v * kD;
      const float* src = v == 0 ? bz : v == 1 ? br : v == 2 ? bh : v == 3 ? gamma : beta;
      val = src[i];
    }
    upd[t] = val;
  }
}

// -----------------------------------------------------------------------------------------
// Share of persistent encoder workgroup j, resolved by ONE WAVE from the 16-molecule partial sums
// (a few hundred values): the rows of the batch are dealt to the `nwg` workgroups in equal contiguous
// shares, per ion, in proportion to the ion's rows.  Returns false for an empty share.
//   g: ion; k0: 16-molecule block that holds virtual row t_lo; bp0: virtual-row prefix at that block;
//   [t_lo, t_hi): the share's virtual rows.
// Every plan_chunks workgroup of a share recomputes this (a handful of L2 hits) instead of reading it
// from a separate single-workgroup kernel: one launch and one dependent stage fewer.
// -----------------------------------------------------------------------------------------
// Workgroups are dealt to the 8 XCDs round robin (workgroup j runs on XCD j % 8; observed placement, used for speed only).
// The shares are numbered so that the first half of them - ion 0's, when the ions hold about the same number of rows -
// belongs to workgroups on XCDs 0-3 and the second half to XCDs 4-7: an XCD's L2 then holds the type matrices of ONE ion
// (1.7 MB at Vb = 72, S = 3) instead of both (3.4 MB of a 4 MB L2 that also streams the chunk records).
// A bijection on [0, nwg): no result depends on it.
__device__ __forceinline__ int xcd_slot(int j, int nwg) {
#ifdef IMPNN_DIAG_NO_XCD_MAP
  return j;
#else
  const int lo = (nwg >> 3) * 4 + ((nwg & 7) < 4 ? (nwg & 7) : 4);  // workgroups with j % 8 < 4
  const int r = j & 7, qd = j >> 3;
  return r < 4 ? qd * 4 + r : lo + qd * 4 + (r - 4);
#endif
}

__device__ __forceinline__ bool resolve_share(const PlanParams& p, int j, int lane, int& g, int& k0, int& bp0,
                                              int& t_lo, int& t_hi, bool& bad) {
  const int nblk = p.nblk;
  bad = false;
  // All partial sums are fetched up front with clamped, unconditional addresses (kPU x 64 per ion and
  // pass): the loads of a pass are in flight together, and the block search below works on the same
  // registers instead of reading the table a second time.
  constexpr int kPU = 4;
  const int npass = (nblk + 64 * kPU - 1) / (64 * kPU);
  if (npass == 1) {  // B <= 4096 molecules per ion: one pass, everything stays in registers
    int v[2][kPU];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int i = 0; i < kPU; ++i) {
        const int k = lane + 64 * i;
        const int kk = k < nblk ? k : nblk - 1;
        const int x = gi < p.n_ions ? p.partial[(int64_t)gi * nblk + kk] : 0;
        if (__ballot(k < nblk && (x & kPlanBadBit))) bad = true;
        v[gi][i] = k < nblk ? (x & ~kPlanBadBit) : 0;
      }
    if (bad) return false;
    int incl[2][kPU], tot[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      int carry = 0;
#pragma unroll
      for (int i = 0; i < kPU; ++i) {
        incl[gi][i] = carry + wave_incl_scan(v[gi][i]);
        carry = __builtin_amdgcn_readlane(incl[gi][i], 63);
      }
      tot[gi] = carry;
    }
    int nwg0 = p.nwg, nwg1 = 0;
    if (p.n_ions == 2) {
      const long long t0 = tot[0], t1 = tot[1];
      int n0 = (t0 + t1) > 0 ? (int)((p.nwg * t0 + (t0 + t1) / 2) / (t0 + t1)) : p.nwg / 2;
      if (p.nwg >= 2) n0 = n0 < 1 ? 1 : (n0 > p.nwg - 1 ? p.nwg - 1 : n0);
      nwg0 = n0;
      nwg1 = p.nwg - n0;
    }
    const int jp = xcd_slot(j, p.nwg);
    g = jp < nwg0 ? 0 : 1;
    const int jj = jp - (g ? nwg0 : 0);
    const int nwg_g = g ? nwg1 : nwg0;
    const long long tg = tot[g];
    t_lo = (int)(tg * jj / nwg_g);
    t_hi = (jj + 1 == nwg_g) ? (int)tg : (int)(tg * (jj + 1) / nwg_g);
    k0 = 0;
    bp0 = 0;
    if (t_hi <= t_lo) return false;
    bool found = false;
#pragma unroll
    for (int i = 0; i < kPU; ++i) {
      const int vv = g ? v[1][i] : v[0][i];
      const int st = (g ? incl[1][i] : incl[0][i]) - vv;
      const unsigned long long hit = __ballot(st <= t_lo && t_lo < st + vv);
      if (hit && !found) {
        const int src = __builtin_ctzll(hit);
        k0 = 64 * i + src;
        bp0 = __shfl(st, src);
        found = true;
      }
    }
    return found;
  }
  long long tot[2] = {0, 0};
  for (int gi = 0; gi < p.n_ions; ++gi) {
    const int32_t* part = p.partial + (int64_t)gi * nblk;
    int acc = 0, anybad = 0;
    for (int k = lane; k < nblk; k += 64) {
      const int x = part[k];
      anybad |= x & kPlanBadBit;
      acc += x & ~kPlanBadBit;
    }
    if (__ballot(anybad != 0)) bad = true;
    tot[gi] = wave_incl_scan(acc);
    tot[gi] = __shfl((int)tot[gi], 63);
  }
  if (bad) return false;
  int nwg0 = p.nwg, nwg1 = 0;
  if (p.n_ions == 2) {
    const long long t0 = tot[0], t1 = tot[1];
    int n0 = (t0 + t1) > 0 ? (int)((p.nwg * t0 + (t0 + t1) / 2) / (t0 + t1)) : p.nwg / 2;
    if (p.nwg >= 2) n0 = n0 < 1 ? 1 : (n0 > p.nwg - 1 ? p.nwg - 1 : n0);
    nwg0 = n0;
    nwg1 = p.nwg - n0;
  }
  const int jp = xcd_slot(j, p.nwg);
  g = jp < nwg0 ? 0 : 1;
  const int jj = jp - (g ? nwg0 : 0);
  const int nwg_g = g ? nwg1 : nwg0;
  const long long tg = tot[g];
  t_lo = (int)(tg * jj / nwg_g);
  t_hi = (jj + 1 == nwg_g) ? (int)tg : (int)(tg * (jj + 1) / nwg_g);
  k0 = 0;
  bp0 = 0;
  if (t_hi <= t_lo) return false;
  // block that holds virtual row t_lo: prefix[k] <= t_lo < prefix[k] + partial[k]
  const int32_t* part = p.partial + (int64_t)g * nblk;
  int carry = 0;
  bool found = false;
  for (int kbase = 0; kbase < nblk && !found; kbase += 64) {
    const int k = kbase + lane;
    const int v = k < nblk ? (part[k < nblk ? k : nblk - 1] & ~kPlanBadBit) : 0;
    const int incl = wave_incl_scan(v);
    const int st = carry + incl - v;
    const unsigned long long hit = __ballot(k < nblk && st <= t_lo && t_lo < st + v);
    if (hit) {
      const int src = __builtin_ctzll(hit);
      k0 = kbase + src;
      bp0 = __shfl(st, src);
      found = true;
    }
    carry += __shfl(incl, 63);
  }
  return found;
}

// -----------------------------------------------------------------------------------------
// plan_chunks: 256 threads (= kRCap: one per row) per chunk.
// -----------------------------------------------------------------------------------------
constexpr int kDegBins = 18;    // in-degree 0..15, ">= 16", and "row beyond the chunk" (placed last)

// Exclusive prefix sum over the 256 threads of a plan_chunks workgroup (4 waves); `total` gets the sum.
__device__ __forceinline__ int block_excl_scan(int v, int32_t* wsum /* [4] LDS */, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = wave_incl_scan(v);
  if (lane == 63) wsum[wave] = incl;
  lds_barrier();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  lds_barrier();  // wsum may be reused right away
  return off + incl - v;
}

// TYPED: records of the typed encoder (encoder_layout.h "typed"): rows placed by EXACT descending in-degree, message
// slots in jagged-diagonal order, edges grouped by bond type.  Dynamic LDS: the group table (16 B x (128 + Vb)).
template <bool TYPED>
__global__ __launch_bounds__(kRCap, 5) void plan_chunks_kernel(PlanParams p) {  // <= 96 VGPRs: 5 workgroups per CU

  __shared__ int32_t moloff[kRCap + 2], molrows[kRCap], cnt[kRCap], place[kRCap], cursor[kRCap], rowptr[kRCap + 2];
  __shared__ int32_t bins[TYPED ? 2 * kRCap + 2 : 48], tilemax[16], scratch[8];
  __shared__ int32_t jdp[TYPED ? kRCap + 2 : 1], thist[TYPED ? kTVbMax : 1], tgb[TYPED ? kTVbMax : 1];
  extern __shared__ uint4 grp[];  // TYPED only
  __shared__ uint16_t atomof[kRCap];  // placed row -> atom id clamped to [0, Va] (Va = the zero row)
  __shared__ uint32_t ent2[kECap + 1];
  __shared__ int32_t shst[kShareCap + 1];  // virtual-row prefix of the share's molecules (+ end)
  __shared__ int32_t cb[kMaxHops + 1];     // next-fit chain: share-local first molecule of every chunk (+ end)
  __shared__ int chunk_s[3];               // first molecule of the share, chunks of the share, ion
  __builtin_amdgcn_s_setprio(3);
#define CSTAMP(i) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
  CSTAMP(0);
  // grid = nwg x grid_sub: workgroup (j, slot_i) builds chunks slot_i, slot_i + grid_sub, ... of share j.  grid_sub
  // covers the usual chunk count, so that every workgroup is resident at once (slots beyond a share's chunks only
  // resolve and leave); shares with more chunks take another turn of the loop below.
  const int j = blockIdx.x / p.grid_sub, slot_i = blockIdx.x - j * p.grid_sub;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- resolve the share: molecules whose first virtual row lies in [t_lo, t_hi), their prefix, and
  //      the next-fit chain of chunks; wave 0 does it, everybody else waits at the barrier.
  if (wave == 0) {
    int g = 0, k0 = 0, bp0 = 0, t_lo = 0, t_hi = 0;
    bool bad = false;
    const bool have = resolve_share(p, j, lane, g, k0, bp0, t_lo, t_hi, bad);
    if (!have) {
      if (lane == 0) {
        chunk_s[0] = -1; chunk_s[1] = 0; chunk_s[2] = 0;
        if (slot_i == 0) p.nsub[j] = 0;
        if (bad && blockIdx.x == 0) p.header->overflow = 1;  // a molecule larger than a chunk: every share is empty
      }
    } else {
    const int32_t* vrg = p.vr + (int64_t)g * p.B;
    int run = bp0;  // prefix at molecule k0 * 16
    int first = -1, nsh = 0;                          // first share molecule (global index), count
    int end_row = -1;                                 // first virtual row after the share's last molecule
    // clamped addresses, masked values; the loads of the next two 64-molecule groups are always in flight
    auto vr_at = [&](int m) { return vrg[m < p.B ? m : p.B - 1]; };
    int v_n1 = vr_at(k0 * kPB + lane), v_n2 = vr_at(k0 * kPB + 64 + lane);
    for (int mbase = k0 * kPB; mbase < p.B && end_row < 0; mbase += 64) {
      const int m = mbase + lane;
      const int v = v_n1;
      v_n1 = v_n2;
      v_n2 = vr_at(mbase + 128 + lane);
      const int vv = m < p.B ? v : 0;
      const int incl = wave_incl_scan(vv);
      const int st = run + incl - vv;  // first virtual row of molecule m
      const bool in = m < p.B && st >= t_lo && st < t_hi;
      const unsigned long long inb = __ballot(in);
      if (inb) {
        if (first < 0) first = mbase + __builtin_ctzll(inb);
        const int pos = nsh + __builtin_popcountll(inb & ((1ull << lane) - 1));
        if (in && pos < kShareCap) shst[pos] = st;
        nsh += __builtin_popcountll(inb);
      }
      const unsigned long long ge = __ballot(m < p.B && st >= t_hi);
      if (ge) end_row = __shfl(st, __builtin_ctzll(ge));  // the next share's first molecule starts here
      run += __shfl(incl, 63);
    }
    if (end_row < 0) end_row = run;           // ran off the end of the ion
    nsh = nsh < kShareCap ? nsh : kShareCap;  // a share holds ~B/nwg molecules; launch_plan checks the cap
    if (lane == 0) shst[nsh] = end_row;
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): sst is read back by this wave below
    int mb = 0, hop = 0;
    while (mb < nsh && hop < kMaxHops) {  // next-fit: chunk [mb, e), e = largest index with shst[e] - shst[mb] <= 256
      const int lim = shst[mb] + kRCap;
      int e = mb + 1;
      for (int c0 = mb + 1; c0 <= nsh; c0 += 64) {
        const int cidx = c0 + lane;
        const bool ok = cidx <= nsh && shst[cidx <= nsh ? cidx : nsh] <= lim;
        const unsigned long long okb = __ballot(ok);
        if (okb == 0) break;
        e = c0 + 63 - __builtin_clzll(okb);
        if (okb != ~0ull) break;
      }
      if (lane == 0) cb[hop] = mb;
      ++hop;
      mb = e;
    }
    if (lane == 0) {
      cb[hop] = nsh;
      chunk_s[0] = first;
      chunk_s[1] = hop;
      chunk_s[2] = g;
      if (slot_i == 0) p.nsub[j] = hop;
    }
    }
  }
  lds_barrier();
  CSTAMP(1);
  const int first_mol = chunk_s[0], nhop = chunk_s[1], g = chunk_s[2];
  for (int sl = slot_i; sl < nhop; sl += p.grid_sub) {
  const int mb_local = cb[sl], M = cb[sl + 1] - mb_local, m0 = first_mol + mb_local;
  const int R = shst[mb_local + M] - shst[mb_local];
  const int idx = j * p.max_sub + sl;
  if (tid == 0) reinterpret_cast<int4*>(p.desc)[idx] = make_int4(m0, M, 0, R | (g << 16));
  const int N = p.N, E = p.E;
  const int32_t* ids_g = p.atom_ids[g];
  const int32_t* conn_g = p.conn[g];
  const int32_t* bond_g = p.bond_ids[g];
  const int32_t* rows_g = p.rows + (int64_t)g * p.B;
  unsigned char* rec = p.rec + (size_t)idx * (TYPED ? kTRecBytes : kRecBytes);
  uint16_t* r_rowptr = reinterpret_cast<uint16_t*>(rec + kRecRowptr);
  unsigned char* r_tilemax = rec + kRecTilemax;
  uint16_t* r_moloff = reinterpret_cast<uint16_t*>(rec + kRecMoloff);
  uint16_t* r_molrows = reinterpret_cast<uint16_t*>(rec + kRecMolrows);
  uint16_t* r_poolrow = reinterpret_cast<uint16_t*>(rec + kRecPoolrow);
  int32_t* r_rowatom = reinterpret_cast<int32_t*>(rec + kRecRowatom);
  uint32_t* r_ent = reinterpret_cast<uint32_t*>(rec + kRecEnt);

  // P0: molecule tables; the first edge slots of this thread start their flight now (their addresses
  //     need only the descriptor), so the dependent-load chain is descriptor -> {tables, edges} -> ids
  const int n_slots = M * E;
  constexpr int kSC = 4;  // edge slots per thread kept in registers (covers M*E <= 1024)
  int sm[kSC], se[kSC], sbid[kSC];
  int2 sst[kSC];
#pragma unroll
  for (int k = 0; k < kSC; ++k) {
    const int slot = tid + k * kRCap;
    sm[k] = 0; se[k] = 0; sbid[k] = -1; sst[k] = make_int2(0, 0);
    if (slot < n_slots) {
      sm[k] = slot / E;
      se[k] = slot - sm[k] * E;
      const int64_t b = m0 + sm[k];
      sst[k] = *reinterpret_cast<const int2*>(conn_g + (b * E + se[k]) * 2);
      sbid[k] = bond_g[b * E + se[k]];
    }
  }
  for (int m = tid; m <= M; m += kRCap) {
    const int off = shst[mb_local + m] - shst[mb_local];
    moloff[m] = off;
    r_moloff[m] = (uint16_t)off;
    if (m < M) {
      const int rr = rows_g[m0 + m];
      molrows[m] = rr;
      r_molrows[m] = (uint16_t)rr;
    }
  }
  cnt[tid] = 0;
  if constexpr (TYPED) {
    bins[tid] = 0;          // [0, 256): rows per placement bin (bin = 255 - in-degree)
    bins[kRCap + tid] = 0;  // [256, 512): fill cursors
    thist[tid] = 0;
  } else {
    if (tid < 48) bins[tid] = 0;
  }
  if (tid < 16) tilemax[tid] = 0;
  lds_barrier();
  CSTAMP(2);

  // P1: in-degree of every logical row (edge-parallel, coalesced reads of conn / bond ids);
  //     logical row -> (molecule, n), atom id
#pragma unroll
  for (int k = 0; k < kSC; ++k)
    if (edge_valid(sst[k].x, sst[k].y, sbid[k], N, p.Vb)) atomicAdd(&cnt[moloff[sm[k]] + sst[k].y], 1);
  for (int slot = tid + kSC * kRCap; slot < n_slots; slot += kRCap) {
    const int m = slot / E, e = slot - m * E;
    const int64_t b = m0 + m;
    const int2 st = *reinterpret_cast<const int2*>(conn_g + (b * E + e) * 2);
    if (edge_valid(st.x, st.y, bond_g[b * E + e], N, p.Vb)) atomicAdd(&cnt[moloff[m] + st.y], 1);
  }
  int my_id = -1;       // atom id of logical row tid (-1: slack row)
  bool my_real = false;
  if (tid < R) {
    int lo = 0, hi = M - 1;  // largest m with moloff[m] <= row
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (moloff[mid] <= tid) lo = mid; else hi = mid - 1;
    }
    const int n = tid - moloff[lo];
    if (n < molrows[lo]) {
      my_real = true;
      my_id = ids_g[(int64_t)(m0 + lo) * N + n];
    }
  }
  lds_barrier();
  CSTAMP(3);

  // P2: place rows by descending in-degree (counting sort) so that a tile's lanes walk
  //     in-edge lists of similar length.  The placement inside a bin comes from an LDS atomic and
  //     may differ run to run - harmless: no result depends on where a row sits (MFMA columns,
  //     the gather and LayerNorm are per row; the pool walks logical rows in order).
  const int my_deg = cnt[tid];
  int pos;
  if constexpr (TYPED) {
    // exact bins: 255 - in-degree (in-degree <= E <= 255); rows beyond the chunk go last
    const int dcl = my_deg > 255 ? 255 : my_deg;
    if (tid < R && my_deg > 255) p.header->overflow = 1;  // in-degrees travel as 8 bits: the encoder refuses this plan
    if (tid < R) atomicAdd(&bins[255 - dcl], 1);
    lds_barrier();
    CSTAMP(4);
    int tot = 0;
    const int start = block_excl_scan(bins[tid], scratch, tot);  // rows placed before bin `tid` = rows with a larger in-degree
    cursor[tid] = start;  // (reused below for the placed in-degrees: read back first)
    lds_barrier();
    // jagged-diagonal pointers: rows with in-degree > d are exactly the first cursor[255 - d] placed rows
    const int sd = cursor[255 - tid];  // S_d for d = tid
    int nedge = 0;
    const int jd = block_excl_scan(sd, scratch, nedge);
    jdp[tid] = jd;
    if (tid == 0) jdp[kRCap] = nedge;
    const int my_start = tid < R ? cursor[255 - dcl] : 0;
    lds_barrier();
    CSTAMP(5);
    pos = tid < R ? my_start + atomicAdd(&bins[kRCap + 255 - dcl], 1) : 0;
    // rows beyond the chunk: after the R real rows, in thread order (ballot ranks: no atomics needed)
    {
      const unsigned long long beyond = __ballot(tid >= R);
      if (lane == 0) scratch[4 + wave] = __builtin_popcountll(beyond);
      lds_barrier();
      int before = 0;
      for (int w = 0; w < wave; ++w) before += scratch[4 + w];
      if (tid >= R) pos = R + before + __builtin_popcountll(beyond & ((1ull << lane) - 1));
    }
  } else {
    const int my_bin = tid >= R ? 0 : (my_deg >= 16 ? 1 : 17 - my_deg);  // bin 0 = beyond chunk (placed last)
    atomicAdd(&bins[my_bin], 1);
    lds_barrier();
    CSTAMP(4);
    if (wave == 0) {  // exclusive scan in placement order: bins 1..17, then bin 0
      const int bidx = lane < kDegBins ? (lane == kDegBins - 1 ? 0 : lane + 1) : 0;
      const int v = lane < kDegBins ? bins[bidx] : 0;
      int incl = v;
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane < kDegBins) bins[24 + bidx] = incl - v;
    }
    lds_barrier();
    CSTAMP(5);
    pos = bins[24 + my_bin] + atomicAdd(&bins[my_bin], -1) - 1;
  }
  lds_barrier();  // (typed: every thread has read its start before cursor is rewritten)
  place[tid] = pos;
  cursor[pos] = my_deg;  // in-degree per placed row (scanned below)
  if (my_deg > 0) atomicMax(&tilemax[pos >> 4], my_deg > 255 ? 255 : my_deg);
  r_rowatom[pos] = my_real ? my_id : -1;  // out-of-range ids (incl. negative) read as a zero row in the encoder
  atomof[pos] = (uint16_t)((my_real && (unsigned)my_id < (unsigned)p.Va) ? my_id : p.Va);
  r_poolrow[tid] = (uint16_t)(pos | ((my_real && my_id > 0) ? 0x8000 : 0));
  lds_barrier();
  CSTAMP(6);

  // P3: exclusive scan of the placed in-degrees -> rowptr; cursor = fill position
  {
    const int my_cnt = cursor[tid];
    int incl = my_cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) scratch[wave] = incl;
    lds_barrier();
  CSTAMP(7);
    int off = 0;
    for (int w = 0; w < wave; ++w) off += scratch[w];
    const int excl = off + incl - my_cnt;
    rowptr[tid] = excl;
    cursor[tid] = excl;
    if constexpr (TYPED) {
      r_rowptr[tid] = (uint16_t)my_cnt;  // kTRecRowdeg: the encoder needs the in-degree, not the CSR offset
      reinterpret_cast<uint16_t*>(rec + kTRecJdptr)[tid] = (uint16_t)jdp[tid];
      if (tid == kRCap - 1) rowptr[kRCap] = excl + my_cnt;
      if (tid < 2) reinterpret_cast<uint16_t*>(rec + kTRecJdptr)[kRCap + tid] = (uint16_t)jdp[kRCap];
    } else {
      r_rowptr[tid] = (uint16_t)excl;
      if (tid == kRCap - 1) {
        rowptr[kRCap] = excl + my_cnt;
        r_rowptr[kRCap] = (uint16_t)(excl + my_cnt);
      }
    }
    if (tid < 16) r_tilemax[tid] = (unsigned char)tilemax[tid];
  }
  lds_barrier();
  CSTAMP(8);

  // P4: fill.  entry = edge slot (16b) | bond id (8b) | placed source row (8b); the slot in the top
  //     bits lets P5 restore edge-slot order, so the accumulation order is fixed run to run.
#pragma unroll
  for (int k = 0; k < kSC; ++k)
    if (edge_valid(sst[k].x, sst[k].y, sbid[k], N, p.Vb)) {
      const int mo = moloff[sm[k]];
      const int at = atomicAdd(&cursor[place[mo + sst[k].y]], 1);
      ent2[at] = ((uint32_t)se[k] << 16) | ((uint32_t)sbid[k] << 8) | (uint32_t)place[mo + sst[k].x];
      if constexpr (TYPED) atomicAdd(&thist[sbid[k]], 1);
    }
  for (int slot = tid + kSC * kRCap; slot < n_slots; slot += kRCap) {
    const int m = slot / E, e = slot - m * E;
    const int64_t b = m0 + m;
    const int2 st = *reinterpret_cast<const int2*>(conn_g + (b * E + e) * 2);
    const int bid = bond_g[b * E + e];
    if (edge_valid(st.x, st.y, bid, N, p.Vb)) {
      const int mo = moloff[m];
      const int at = atomicAdd(&cursor[place[mo + st.y]], 1);
      ent2[at] = ((uint32_t)e << 16) | ((uint32_t)bid << 8) | (uint32_t)place[mo + st.x];
      if constexpr (TYPED) atomicAdd(&thist[bid], 1);
    }
  }
  lds_barrier();
  CSTAMP(9);

  // P5: every row's in-edge list in edge-slot order: entry-parallel rank sort, straight into the record
  if constexpr (TYPED) {
    // Groups of <= 4 edges of one bond type, in type order: entry x = type | edges << 8 | groups of this type from
    // this one on << 24.  All groups of a type form one "run", processed by one wave of the encoder (which fetches
    // the type's matrix once per chunk-step).
    const int n_t = thist[tid];
    const int ng = (n_t + 3) >> 2;  // <= 128
    int ngrp = 0;
    const int gb = block_excl_scan(ng, scratch, ngrp);
    tgb[tid] = gb;
    thist[tid] = 0;  // becomes the fill cursor of the type
    {
      const uint32_t dump = (uint32_t)tmsg_key(p.ecap) * 0x10001u;  // unused edge lanes write to the dump slot
      for (int i = tid; i < ngrp; i += kRCap) grp[i] = make_uint4(0u, 0u, dump, dump);
    }
    lds_barrier();
    if (ng > 0) {
      for (int jg = 0; jg < ng; ++jg) {
        const int c = n_t - 4 * jg;
        grp[gb + jg].x = (uint32_t)tid | ((uint32_t)(c < 4 ? c : 4) << 8) | ((uint32_t)(ng - jg) << 24);
      }
    }
    // run table: first group of every type that has groups, in type order (+ end).  The encoder's waves take runs
    // from it one at a time (an LDS counter), so the message phase is balanced dynamically.
    {
      int nrun = 0;
      const int ridx = block_excl_scan(ng > 0 ? 1 : 0, scratch, nrun);
      uint16_t* runs = reinterpret_cast<uint16_t*>(rec + trec_runs_off(p.Vb, p.ecap));
      if (ng > 0) runs[ridx] = (uint16_t)gb;
      if (tid == 0) {
        runs[nrun] = (uint16_t)ngrp;
        *reinterpret_cast<uint16_t*>(rec + kTRecNrun) = (uint16_t)nrun;
      }
    }
    if (tid == 0) {
      uint16_t* cw = reinterpret_cast<uint16_t*>(rec + kTRecCounts);
      int md = 0;
      for (int t = 0; t < 16; ++t) md = md > tilemax[t] ? md : tilemax[t];
      cw[0] = (uint16_t)ngrp;
      cw[1] = (uint16_t)rowptr[kRCap];
      cw[2] = (uint16_t)md;
    }
    const int total = rowptr[kRCap];
    for (int i = tid; i < total; i += kRCap) {
      int lo = 0, hi = kRCap - 1;  // largest row with rowptr[row] <= i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
      }
      const int b0 = rowptr[lo], b1 = rowptr[lo + 1];
      const uint32_t v = ent2[i];
      int rank = 0;
      for (int jx = b0; jx < b1; ++jx) rank += ent2[jx] < v;
      const uint32_t srow = v & 0xffu, bid = (v >> 8) & 0xffu;
      // jagged diagonal: d-th in-edge of placed row lo (rank > 255 only in a plan that is marked overflowed)
      const uint32_t mslot = (uint32_t)(jdp[rank < 256 ? rank : 255] + lo);
      const int idx = atomicAdd(&thist[bid], 1);
      unsigned char* ge = reinterpret_cast<unsigned char*>(&grp[tgb[bid] + (idx >> 2)]);
      ge[4 + (idx & 3)] = (unsigned char)srow;
      reinterpret_cast<uint16_t*>(ge + 8)[idx & 3] = (uint16_t)tmsg_key((int)mslot);
    }
    lds_barrier();
    uint4* r_grp = reinterpret_cast<uint4*>(rec + kTRecGrp);
    for (int i = tid; i < ngrp; i += kRCap) r_grp[i] = grp[i];
  } else {
    const int total = rowptr[kRCap];
    for (int i = tid; i < total; i += kRCap) {
      int lo = 0, hi = kRCap - 1;  // largest row with rowptr[row] <= i
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
      }
      const int b0 = rowptr[lo], b1 = rowptr[lo + 1];
      const uint32_t v = ent2[i];
      int rank = 0;
      for (int jx = b0; jx < b1; ++jx) rank += ent2[jx] < v;
      // the edge slot has done its job (order); the encoder gets the source row's atom id in its place, so that
      // step 0 can read h0 = atom_table[id] directly (no h0 fill in the chunk prologue)
      // entry for the encoder: [31:20] atom id of the source * kHS/4, [19:8] placed source row * kHS/4 (both
      // are float4 offsets into the atom table / the h buffer), [7:0] bond id
      const uint32_t srow = v & 0xffu, bid = (v >> 8) & 0xffu;
      const uint32_t aid = atomof[srow] < kEntMaxAtom ? atomof[srow] : kEntMaxAtom;
      r_ent[b0 + rank] = ((aid * (kHS / 4)) << 20) | ((srow * (kHS / 4)) << 8) | bid;
    }
  }
  CSTAMP(15);
  lds_barrier();  // the LDS tables are rebuilt by the next chunk of this workgroup (rare: see above)
  }
#undef CSTAMP
}

}  // namespace

int launch_weight_image(const ImageParams& ip, int S, hipStream_t s) {
  if (S <= 0) return IMPNN_OK;
  weight_image_kernel<<<dim3(16, S), 256, 0, s>>>(ip);
  return check_launch("weight_image");
}

int launch_typed_image(const TImageParams& ip, hipStream_t s) {
  const int S = ip.S > 0 ? ip.S : 1;
  float* canon = ip.prepared + (size_t)S * (ip.x3 ? kXUpdSlot : kTUpdSlot) + (size_t)S * ip.Vb * kTMatFloats;
  for (int st = 0; st < ip.S; ++st) {
    if (int rc = launch_bond_type_matrices(ip.bond_table, ip.weights + (int64_t)st * ip.step_floats, canon, ip.Vb, ip.K,
                                           kD, s))
      return rc;
    typed_image_kernel<<<(ip.Vb * kTMatFloats + kTUpdSlot + 255) / 256 < 256 ? (ip.Vb * kTMatFloats + kTUpdSlot + 255) / 256 : 256,
                         256, 0, s>>>(ip, st);
    if (int rc = check_launch("typed_image")) return rc;
  }
  return IMPNN_OK;
}

int launch_plan(const PlanParams& pp, hipStream_t s) {
  plan_stats_kernel<<<pp.n_ions * pp.nblk, 64 * kPB, 0, s>>>(pp);
  if (int rc = check_launch("plan_stats")) return rc;
  if ((int64_t)2 * pp.n_ions * pp.B / pp.nwg + 64 > kShareCap)
    return fail(IMPNN_E_UNSUPPORTED, "encoder plan: batch of %d molecules per ion is too large", pp.B);
  if (pp.max_sub > kMaxHops)
    return fail(IMPNN_E_UNSUPPORTED, "encoder plan: %d chunk slots per workgroup", pp.max_sub);
  if (pp.typed)
    plan_chunks_kernel<true><<<pp.nwg * pp.grid_sub, kRCap, sizeof(uint4) * tgrp_cap(pp.Vb, pp.ecap), s>>>(pp);
  else
    plan_chunks_kernel<false><<<pp.nwg * pp.grid_sub, kRCap, 0, s>>>(pp);
  return check_launch("plan_chunks");
}

}  // namespace enc
}  // namespace impnn
