// Plan kernels of the fused encoder (gfx950): everything that depends on the graph batch only
// (not on the weights) and is cheap, latency-bound, index arithmetic:
//
//   plan_stats          one wave per molecule: kept rows r_b, valid edges v_b -> virtual rows (>= plan_vmin); one
//                       partial sum per 16 molecules
//   plan_chunks[_typed] one 256-thread workgroup per (share, chunk slot) (4-5 resident per CU, so their dependent loads
//                       overlap).  The rows of the batch are dealt to `nwg` persistent encoder workgroups in equal
//                       contiguous shares (per ion, proportional to its rows); one wave resolves the share from the
//                       partial sums, the share's molecules and its next-fit chain of chunks (resolve_chain), then the
//                       workgroup builds its chunk's record in HBM:
//                         pull form (modes 0 / 1): rows placed by in-degree class, CSR of in-edges in edge-slot order
//                         typed (modes 2 / 3): rows placed by exact in-degree, jagged-diagonal message slots, edges
//                         grouped by bond type in groups of 4 and runs of a few groups (the comment at the kernel)
//   weight_image / typed_image   canonical weights -> the encoder's LDS images (weights only; run when they change)
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
  const int id_first = lane < p.N ? ids[lane] : 0;  // requested with the edge slots below, looked at behind them
  int emax = 0;  // largest atom index on a valid edge (lane-local)
  for (int e0 = 0; e0 < p.E; e0 += 256) {  // four 64-slot groups per turn: their loads are in flight together
    int2 st[4];
    int bid[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + 64 * u + lane;
      st[u] = make_int2(0, 0);
      bid[u] = -1;
      if (e < p.E) {
        st[u] = *reinterpret_cast<const int2*>(cn + 2 * e);
        bid[u] = bd[e];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = edge_valid(st[u].x, st[u].y, bid[u], p.N, p.Vb);
      if (ok) {
        const int m = st[u].x > st[u].y ? st[u].x : st[u].y;
        emax = emax > m ? emax : m;
      }
      cnt += __builtin_popcountll(__ballot(ok));
    }
  }
  {
    const unsigned long long hit = __ballot(id_first > 0);
    if (hit) rmax = 64 - __builtin_clzll(hit);  // 1 + highest n with ids[n] > 0
  }
  for (int n0 = 64; n0 < p.N; n0 += 64) {
    const int n = n0 + lane;
    const unsigned long long hit = __ballot(n < p.N && ids[n] > 0);
    if (hit) rmax = n0 + 64 - __builtin_clzll(hit);
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
    my_vr = vr < p.vmin ? p.vmin : vr;  // (>= 1; encoder_layout.h: plan_vmin)
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

// Share boundaries.  Every workgroup evaluates the same expressions on the same totals, so neighbouring shares agree on
// their common boundary whatever the rounding: the formulas have to be deterministic and monotone in the share index,
// not exact - float arithmetic (a handful of instructions) instead of 64-bit integer divisions (a few hundred, on the
// critical path of a single wave).
__device__ __forceinline__ int share_bound(int tg, int jj, int nwg_g) {  // first virtual row of share jj of nwg_g (tg rows)
  if (jj <= 0) return 0;
  if (jj >= nwg_g) return tg;
  const int t = (int)((float)tg * (float)jj * __frcp_rn((float)nwg_g));
  return t < 0 ? 0 : (t > tg ? tg : t);
}
__device__ __forceinline__ int ion_split(long long t0, long long t1, int nwg) {  // workgroups of ion 0, in proportion to its rows
  int n0 = (t0 + t1) > 0 ? (int)((float)nwg * (float)t0 * __frcp_rn((float)(t0 + t1)) + 0.5f) : nwg / 2;
  if (nwg >= 2) n0 = n0 < 1 ? 1 : (n0 > nwg - 1 ? nwg - 1 : n0);
  return n0;
}

#define CSTAMP(i) do { if (p.stamps && blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
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
    // lane l holds blocks 4l .. 4l+3 of either ion: a prefix inside the lane and ONE wave scan per ion (a single wave
    // runs this code alone: its instruction count is its latency)
    int v[2][kPU];
    int anybad = 0;
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int i = 0; i < kPU; ++i) {
        const int k = kPU * lane + i;
        const int kk = k < nblk ? k : nblk - 1;
        const int x = gi < p.n_ions ? p.partial[(int64_t)gi * nblk + kk] : 0;
        anybad |= k < nblk ? x : 0;
        v[gi][i] = k < nblk ? (x & ~kPlanBadBit) : 0;
      }
    if (__ballot((anybad & kPlanBadBit) != 0)) bad = true;
    if (bad) return false;
    CSTAMP(10);
    int ex[2], tot[2];  // blocks before the lane's first, per ion
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int s4 = v[gi][0] + v[gi][1] + v[gi][2] + v[gi][3];
      const int incl = wave_incl_scan(s4);
      ex[gi] = incl - s4;
      tot[gi] = __builtin_amdgcn_readlane(incl, 63);
    }
    int nwg0 = p.nwg, nwg1 = 0;
    if (p.n_ions == 2) {
      int n0 = ion_split(tot[0], tot[1], p.nwg);
      nwg0 = n0;
      nwg1 = p.nwg - n0;
    }
    const int jp = xcd_slot(j, p.nwg);
    g = jp < nwg0 ? 0 : 1;
    const int jj = jp - (g ? nwg0 : 0);
    const int nwg_g = g ? nwg1 : nwg0;
    const int tg = tot[g];
    t_lo = share_bound(tg, jj, nwg_g);
    t_hi = share_bound(tg, jj + 1, nwg_g);
    k0 = 0;
    bp0 = 0;
    if (t_hi <= t_lo) return false;
    // the block that holds virtual row t_lo: first the lane (its four blocks cover [ex, ex + s4)), then the block in it
    const int e0 = g ? ex[1] : ex[0];
    const int a0 = g ? v[1][0] : v[0][0], a1 = g ? v[1][1] : v[0][1], a2 = g ? v[1][2] : v[0][2], a3 = g ? v[1][3] : v[0][3];
    const unsigned long long hit = __ballot(e0 <= t_lo && t_lo < e0 + a0 + a1 + a2 + a3);
    if (hit == 0ull) return false;
    const int src = __builtin_ctzll(hit);
    const int i_in = t_lo < e0 + a0 ? 0 : (t_lo < e0 + a0 + a1 ? 1 : (t_lo < e0 + a0 + a1 + a2 ? 2 : 3));
    const int st_in = e0 + (i_in > 0 ? a0 : 0) + (i_in > 1 ? a1 : 0) + (i_in > 2 ? a2 : 0);
    k0 = kPU * src + __builtin_amdgcn_readlane(i_in, src);
    bp0 = __builtin_amdgcn_readlane(st_in, src);
    return true;
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
    tot[gi] = __builtin_amdgcn_readlane((int)tot[gi], 63);
  }
  if (bad) return false;
  int nwg0 = p.nwg, nwg1 = 0;
  if (p.n_ions == 2) {
    int n0 = ion_split(tot[0], tot[1], p.nwg);
    nwg0 = n0;
    nwg1 = p.nwg - n0;
  }
  const int jp = xcd_slot(j, p.nwg);
  g = jp < nwg0 ? 0 : 1;
  const int jj = jp - (g ? nwg0 : 0);
  const int nwg_g = g ? nwg1 : nwg0;
  const int tg = (int)tot[g];
  t_lo = share_bound(tg, jj, nwg_g);
  t_hi = share_bound(tg, jj + 1, nwg_g);
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
      bp0 = __builtin_amdgcn_readlane(st, src);
      found = true;
    }
    carry += __builtin_amdgcn_readlane(incl, 63);
  }
  return found;
}

// -----------------------------------------------------------------------------------------
// plan_chunks: 256 threads (= kRCap: one per row) per chunk.
// -----------------------------------------------------------------------------------------
constexpr int kDegBins = 18;    // in-degree 0..15, ">= 16", and "row beyond the chunk" (placed last)

// Share tables of a plan_chunks workgroup (LDS): the virtual-row prefix of the share's molecules and its next-fit chain
struct ShareTab {
  int32_t shst[kShareCap + 1];  // virtual-row prefix of the share's molecules (+ end)
  int32_t cb[kMaxHops + 1];     // next-fit chain: share-local first molecule of every chunk (+ end)
  int32_t vwin[256];            // virtual rows of the molecules around the share's expected start (see resolve_chain)
  int chunk_s[3];               // first molecule of the share, chunks of the share, ion
};

// quotient of two non-negative ints through the float unit (exact for x < 2^24 after the two corrections)
__device__ __forceinline__ int fast_div(int x, int d, float rd) {
  int q = (int)((float)x * rd);
  if (q * d > x) --q;
  if ((q + 1) * d <= x) ++q;
  return q;
}

// Wave 0 of every plan_chunks workgroup: resolve share j (resolve_share), list the virtual-row prefix of its molecules
// and chain them into chunks (next-fit: a chunk takes molecules while their virtual rows stay <= kRCap).
// The share of workgroup j starts near molecule jj * B / nwg_g when the molecules are of similar size; the virtual rows
// of the 256 molecules around that guess are requested together with the partial sums (one round trip to memory
// instead of two dependent ones) and parked in LDS; a share that starts elsewhere reads them from memory as before.
__device__ __forceinline__ void resolve_chain(const PlanParams& p, int j, int slot_i, int lane, ShareTab& T) {
  int gg = 0, w_lo = 0;
  {
    const int half = p.n_ions == 2 ? (p.nwg >> 1 > 0 ? p.nwg >> 1 : 1) : p.nwg;
    const int jp = xcd_slot(j, p.nwg);
    gg = (p.n_ions == 2 && jp >= half) ? 1 : 0;
    const int jj = jp - (gg ? half : 0);
    const int nw = gg ? (p.nwg - half > 0 ? p.nwg - half : 1) : half;
    w_lo = (int)((float)jj * (float)p.B / (float)nw) - 64;  // (a guess: need not be exact)
    w_lo = w_lo < 0 ? 0 : w_lo;
    w_lo &= ~15;
  }
  int vw[4];
  {
    const int32_t* vrg = p.vr + (int64_t)gg * p.B;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = w_lo + 64 * i + lane;
      vw[i] = vrg[m < p.B ? m : p.B - 1];
    }
  }
  int g = 0, k0 = 0, bp0 = 0, t_lo = 0, t_hi = 0;
  bool bad = false;
  const bool have = resolve_share(p, j, lane, g, k0, bp0, t_lo, t_hi, bad);  // (its loads are behind the window's)
  CSTAMP(11);
#pragma unroll
  for (int i = 0; i < 4; ++i) T.vwin[64 * i + lane] = vw[i];
  if (!have) {
    if (lane == 0) {
      T.chunk_s[0] = -1; T.chunk_s[1] = 0; T.chunk_s[2] = 0;
      if (slot_i == 0) p.nsub[j] = 0;
      if (bad && blockIdx.x == 0) p.header->overflow = 1;  // a molecule larger than a chunk: every share is empty
    }
    return;
  }
  const int32_t* vrg = p.vr + (int64_t)g * p.B;
  int run = bp0;  // prefix at molecule k0 * 16
  int first = -1, nsh = 0;                          // first share molecule (global index), count
  int end_row = -1;                                 // first virtual row after the share's last molecule
  __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): vwin is read back by this wave
  // 64 molecules from mbase on: out of the window when they all lie in it, else from memory (clamped addresses)
  auto vr_at = [&](int mbase) {
    const int rel = mbase - w_lo;
    if (g == gg && rel >= 0 && rel + 64 <= 256) return T.vwin[rel + lane];
    const int m = mbase + lane;
    return vrg[m < p.B ? m : p.B - 1];
  };
  int v_n1 = vr_at(k0 * kPB);
  for (int mbase = k0 * kPB; mbase < p.B && end_row < 0; mbase += 64) {
    const int m = mbase + lane;
    const int v = v_n1;
    v_n1 = vr_at(mbase + 64);
    const int vv = m < p.B ? v : 0;
    const int incl = wave_incl_scan(vv);
    const int st = run + incl - vv;  // first virtual row of molecule m
    const bool in = m < p.B && st >= t_lo && st < t_hi;
    const unsigned long long inb = __ballot(in);
    if (inb) {
      if (first < 0) first = mbase + __builtin_ctzll(inb);
      const int pos = nsh + __builtin_popcountll(inb & ((1ull << lane) - 1));
      if (in && pos < kShareCap) T.shst[pos] = st;
      nsh += __builtin_popcountll(inb);
    }
    const unsigned long long ge = __ballot(m < p.B && st >= t_hi);
    if (ge) end_row = __builtin_amdgcn_readlane(st, __builtin_ctzll(ge));  // the next share's first molecule starts here
    run += __builtin_amdgcn_readlane(incl, 63);
  }
  if (end_row < 0) end_row = run;           // ran off the end of the ion
  if (nsh > kShareCap) {  // cannot happen while every molecule counts plan_vmin virtual rows; never truncate silently
    if (lane == 0) p.header->overflow = 1;
    nsh = kShareCap;
  }
  if (lane == 0) T.shst[nsh] = end_row;
  __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): shst is read back by this wave below
  CSTAMP(12);
  int mb = 0, hop = 0;
  if (nsh < 64) {  // the usual share: every lane keeps one prefix value, a hop is a ballot
    const int st_l = T.shst[lane <= nsh ? lane : nsh];
    while (mb < nsh && hop < kMaxHops) {
      const int lim = __builtin_amdgcn_readlane(st_l, mb) + kRCap;
      const unsigned long long okb = __ballot(lane > mb && lane <= nsh && st_l <= lim);
      if (lane == 0) T.cb[hop] = mb;
      ++hop;
      // highest e in (mb, nsh] whose prefix fits: the chunk is [mb, e), the next one starts at e (a molecule alone otherwise)
      mb = okb ? 63 - __builtin_clzll(okb) : mb + 1;
    }
  }
  while (mb < nsh && hop < kMaxHops) {  // next-fit: chunk [mb, e), e = largest index with shst[e] - shst[mb] <= 256
    const int lim = T.shst[mb] + kRCap;
    int e = mb + 1;
    for (int c0 = mb + 1; c0 <= nsh; c0 += 64) {
      const int cidx = c0 + lane;
      const bool ok = cidx <= nsh && T.shst[cidx <= nsh ? cidx : nsh] <= lim;
      const unsigned long long okb = __ballot(ok);
      if (okb == 0) break;
      e = c0 + 63 - __builtin_clzll(okb);
      if (okb != ~0ull) break;
    }
    if (lane == 0) T.cb[hop] = mb;
    ++hop;
    mb = e;
  }
  if (lane == 0) {
    T.cb[hop] = nsh;
    T.chunk_s[0] = first;
    T.chunk_s[1] = hop;
    T.chunk_s[2] = g;
    if (slot_i == 0) p.nsub[j] = hop;
  }
}


// Records of the pull-form encoder (modes 0 / 1): rows placed by descending in-degree class, CSR of in-edges in edge-slot
// order.
__global__ __launch_bounds__(kRCap, 5) void plan_chunks_kernel(PlanParams p) {  // <= 96 VGPRs: 5 workgroups per CU
  __shared__ int32_t moloff[kRCap + 2], molrows[kRCap], cnt[kRCap], place[kRCap], cursor[kRCap], rowptr[kRCap + 2];
  __shared__ int32_t bins[48], tilemax[16], scratch[8];
  __shared__ uint16_t atomof[kRCap];  // placed row -> atom id clamped to [0, Va] (Va = the zero row)
  __shared__ uint32_t ent2[kECap + 1];
  __shared__ ShareTab T;
  __builtin_amdgcn_s_setprio(3);
  CSTAMP(0);
  // grid = nwg x grid_sub: workgroup (j, slot_i) builds chunks slot_i, slot_i + grid_sub, ... of share j.  grid_sub
  // covers the usual chunk count, so that every workgroup is resident at once (slots beyond a share's chunks only
  // resolve and leave); shares with more chunks take another turn of the loop below.
  const int j = blockIdx.x / p.grid_sub, slot_i = blockIdx.x - j * p.grid_sub;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) resolve_chain(p, j, slot_i, lane, T);
  lds_barrier();
  CSTAMP(1);
  const int first_mol = T.chunk_s[0], nhop = T.chunk_s[1], g = T.chunk_s[2];
  for (int sl = slot_i; sl < nhop; sl += p.grid_sub) {
  const int mb_local = T.cb[sl], M = T.cb[sl + 1] - mb_local, m0 = first_mol + mb_local;
  const int R = T.shst[mb_local + M] - T.shst[mb_local];
  const int idx = j * p.max_sub + sl;
  if (tid == 0) reinterpret_cast<int4*>(p.desc)[idx] = make_int4(m0, M, 0, R | (g << 16));
  const int N = p.N, E = p.E;
  const int32_t* ids_g = p.atom_ids[g];
  const int32_t* conn_g = p.conn[g];
  const int32_t* bond_g = p.bond_ids[g];
  const int32_t* rows_g = p.rows + (int64_t)g * p.B;
  unsigned char* rec = p.rec + (size_t)idx * kRecBytes;
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
    const int off = T.shst[mb_local + m] - T.shst[mb_local];
    moloff[m] = off;
    r_moloff[m] = (uint16_t)off;
    if (m < M) {
      const int rr = rows_g[m0 + m];
      molrows[m] = rr;
      r_molrows[m] = (uint16_t)rr;
    }
  }
  cnt[tid] = 0;
  if (tid < 48) bins[tid] = 0;
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
  {
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
  lds_barrier();
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
    r_rowptr[tid] = (uint16_t)excl;
    if (tid == kRCap - 1) {
      rowptr[kRCap] = excl + my_cnt;
      r_rowptr[kRCap] = (uint16_t)(excl + my_cnt);
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
    }
  }
  lds_barrier();
  CSTAMP(9);

  // P5: every row's in-edge list in edge-slot order: entry-parallel rank sort, straight into the record
  {
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
}

// Records of the typed encoder (encoder_layout.h "typed"): rows placed by EXACT descending in-degree, message slots in
// jagged-diagonal order, edges grouped by bond type.  Six barriers per chunk:
//   P0  edge slots -> registers (the loads fly while the row -> molecule search and the table resets run); atom ids
//   P1  per valid edge: a ticket from the target row's in-degree counter - the edge's slot number goes to the row's
//       ticket list (<= kTick entries; rows with more take the slow path of P4) - and the type histogram
//   P2  in-degree histogram of the rows
//   P3  wave 0: rows placed before every in-degree bin (prefix over the histogram) and the jagged-diagonal pointers
//       (prefix over the suffix counts); wave 1: groups per type, first group of every type, the run table
//   P4  rows take their place; per-row tables of the record; the type's groups with their headers
//   P5  per valid edge: its rank among the in-edges of its row in edge-slot order = the tickets with a smaller slot
//       number -> message slot; a place in a group of its type
//   P6  group table -> record
// Dynamic LDS: the group table (16 B x tgrp_cap).
constexpr int kTick = 16;
__global__ __launch_bounds__(kRCap, 4) void plan_chunks_typed_kernel(PlanParams p) {  // <= 128 VGPRs: 4 workgroups per CU
  __shared__ int32_t moloff[kRCap + 2], molrows[kRCap], cnt[kRCap + 64], place[kRCap];  // cnt, thist: + a spare word per lane
  __shared__ int32_t idl[kRCap];  // atom id of the logical row (-1: a slack row)
  __shared__ __attribute__((aligned(16))) int32_t dbins[kRCap], dstart[kRCap], dcur[kRCap];
  __shared__ __attribute__((aligned(16))) int32_t thist[kTVbMax + 64], tgb[kTVbMax], tcur[kTVbMax + 64];
  __shared__ __attribute__((aligned(16))) int32_t jdp[kRCap + 4];
  __shared__ int32_t misc[4];  // groups of the chunk; valid edges listed so far
  __shared__ __attribute__((aligned(16))) uint16_t tick[kRCap * kTick];
  __shared__ uint2 elist[kTECapBig];  // the chunk's valid edges: x = target row | source row << 8 | type << 16 (logical rows), y = edge slot
  extern __shared__ uint4 grp[];
  __shared__ ShareTab T;
  __builtin_amdgcn_s_setprio(3);
  CSTAMP(0);
  const int j = blockIdx.x / p.grid_sub, slot_i = blockIdx.x - j * p.grid_sub;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Wave w of a workgroup runs on SIMD w, and the workgroups that share a CU are 256 apart in the grid (round robin over
  // 8 XCDs x 32 CUs; observed, used for speed only): the resolving wave rotates with the workgroup so that the four
  // resolutions of a CU run on four SIMDs instead of queueing on SIMD 0.
  if (wave == ((blockIdx.x >> 8) & 3)) resolve_chain(p, j, slot_i, lane, T);
  lds_barrier();
  CSTAMP(1);
  const int first_mol = T.chunk_s[0], nhop = T.chunk_s[1], g = T.chunk_s[2];
  const int N = p.N, E = p.E;
  const float rE = 1.0f / (float)E;
  for (int sl = slot_i; sl < nhop; sl += p.grid_sub) {
  const int mb_local = T.cb[sl], M = T.cb[sl + 1] - mb_local, m0 = first_mol + mb_local;
  const int base = T.shst[mb_local];
  const int R = T.shst[mb_local + M] - base;
  const int idx = j * p.max_sub + sl;
  if (tid == 0) reinterpret_cast<int4*>(p.desc)[idx] = make_int4(m0, M, 0, R | (g << 16));
  const int32_t* ids_g = p.atom_ids[g];
  const int32_t* conn_g = p.conn[g];
  const int32_t* bond_g = p.bond_ids[g];
  const int32_t* rows_g = p.rows + (int64_t)g * p.B;
  unsigned char* rec = p.rec + (size_t)idx * kTRecBytes;
  uint16_t* r_rowdeg = reinterpret_cast<uint16_t*>(rec + kTRecRowdeg);
  unsigned char* r_tilemax = rec + kTRecTilemax;
  uint16_t* r_moloff = reinterpret_cast<uint16_t*>(rec + kTRecMoloff);
  uint16_t* r_molrows = reinterpret_cast<uint16_t*>(rec + kTRecMolrows);
  uint16_t* r_poolrow = reinterpret_cast<uint16_t*>(rec + kTRecPoolrow);
  int32_t* r_rowatom = reinterpret_cast<int32_t*>(rec + kTRecRowatom);
  uint16_t* r_jd = reinterpret_cast<uint16_t*>(rec + kTRecJdptr);
  uint16_t* r_cw = reinterpret_cast<uint16_t*>(rec + kTRecCounts);

  // P0: every edge slot of the chunk's molecules is requested now, kSC per thread (M * E <= 1536 slots; more: the loop
  //     of P1), so the chain of dependent loads is descriptor -> {edges, ids, rows}
  const int n_slots = M * E;  // <= 256 molecules x 65535 slots: fits 24 bits, fast_div is exact
  constexpr int kSC = 6;
  int sbid[kSC];
  int2 sst[kSC];
  const int32_t* conn_c = conn_g + (int64_t)m0 * E * 2;  // the chunk's molecules are contiguous: slot -> address
  const int32_t* bond_c = bond_g + (int64_t)m0 * E;
#pragma unroll
  for (int k = 0; k < kSC; ++k) {
    const int slot = tid + k * kRCap;
    sbid[k] = -1; sst[k] = make_int2(0, 0);
    if (slot < n_slots) {
      sst[k] = *reinterpret_cast<const int2*>(conn_c + 2 * (int64_t)slot);
      sbid[k] = bond_c[slot];
    }
  }
  // atom ids of the chunk's molecules: kIC per thread (M * N <= 768; more: the loop of P1)
  constexpr int kIC = 3;
  const int n_ids = M * N;
  const float rN = 1.0f / (float)N;
  const int32_t* ids_c = ids_g + (int64_t)m0 * N;
  int sid[kIC];
#pragma unroll
  for (int k = 0; k < kIC; ++k) {
    const int slot = tid + k * kRCap;
    sid[k] = slot < n_ids ? ids_c[slot] : 0;
  }
  for (int m = tid; m <= M; m += kRCap) {
    const int off = T.shst[mb_local + m] - base;
    moloff[m] = off;
    r_moloff[m] = (uint16_t)off;
    if (m < M) {
      const int rr = rows_g[m0 + m];
      molrows[m] = rr;
      r_molrows[m] = (uint16_t)rr;
    }
  }
  cnt[tid] = 0;
  dbins[tid] = 0;
  dcur[tid] = 0;
  thist[tid] = 0;
  tcur[tid] = 0;
  idl[tid] = -1;
  if (tid == 0) misc[1] = 0;
  {
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);  // empty ticket = slot number 65535 (no edge slot is above it)
    reinterpret_cast<uint4*>(tick)[2 * tid] = ones;
    reinterpret_cast<uint4*>(tick)[2 * tid + 1] = ones;
  }
  lds_barrier();
  CSTAMP(2);

  // P1: atom ids to their logical rows; valid edges -> ticket of the target row, type histogram, the chunk's edge list.
  //     Stage by stage over the register-held slots, so that the LDS operations of all of them are in flight together.
  auto put_id = [&](int slot, int id) {
    const int m = fast_div(slot, N, rN), n = slot - m * N;
    if (slot < n_ids && n < molrows[m]) idl[moloff[m] + n] = id;
  };
  {
    const int dq = fast_div(kRCap, N, rN), dr = kRCap - dq * N;
    int cm = fast_div(tid, N, rN), cn = tid - cm * N;
    int mr[kIC], mo[kIC], nn[kIC];
#pragma unroll
    for (int k = 0; k < kIC; ++k) {  // (branch-free reads: all in flight together)
      const int mm = tid + k * kRCap < n_ids ? cm : 0;
      mr[k] = tid + k * kRCap < n_ids ? molrows[mm] : 0;
      mo[k] = moloff[mm];
      nn[k] = cn;
      cn += dr; cm += dq;
      if (cn >= N) { cn -= N; ++cm; }
    }
#pragma unroll
    for (int k = 0; k < kIC; ++k)
      if (nn[k] < mr[k]) idl[mo[k] + nn[k]] = sid[k];
  }
  for (int slot = tid + kIC * kRCap; slot < n_ids; slot += kRCap) put_id(slot, ids_c[slot]);
  {
    unsigned okm = 0u;
#pragma unroll
    for (int k = 0; k < kSC; ++k) okm |= edge_valid(sst[k].x, sst[k].y, sbid[k], N, p.Vb) ? 1u << k : 0u;
    // places in the edge list: one atomic per wave
    const int mine = __builtin_popcount(okm);
    const int incl = wave_incl_scan(mine);
    int pos = 0;
    if (lane == 63) pos = atomicAdd(&misc[1], incl);
    pos = __builtin_amdgcn_readlane(pos, 63) + incl - mine;
    // slot tid + 256 k = molecule cm, edge slot ce: stepped, not divided
    const int dq = fast_div(kRCap, E, rE), dr = kRCap - dq * E;
    int cm = fast_div(tid, E, rE), ce = tid - cm * E;
#pragma unroll
    for (int k0 = 0; k0 < kSC; k0 += 3) {  // three slots at a time (all six would spill)
      int row[3], srw[3], tk[3], es[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool ok = (okm >> (k0 + k)) & 1u;
        const int m = cm;
        es[k] = ce;
        ce += dr; cm += dq;
        if (ce >= E) { ce -= E; ++cm; }
        const int mo = moloff[m < M ? m : 0];
        row[k] = mo + (ok ? sst[k0 + k].y : 0);
        srw[k] = mo + (ok ? sst[k0 + k].x : 0);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {  // unconditional atomics (slots without a valid edge hit a per-lane spare word): no branches, no waits in between
        const bool ok = (okm >> (k0 + k)) & 1u;
        tk[k] = atomicAdd(&cnt[ok ? row[k] : kRCap + lane], 1);
        atomicAdd(&thist[ok ? sbid[k0 + k] : kTVbMax + lane], 1);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool ok = (okm >> (k0 + k)) & 1u;
        if (ok && tk[k] < kTick) tick[row[k] * kTick + tk[k]] = (uint16_t)es[k];
        if (ok) {
          if (pos < kTECapBig)
            elist[pos] = make_uint2((uint32_t)row[k] | ((uint32_t)srw[k] << 8) | ((uint32_t)sbid[k0 + k] << 16), (uint32_t)es[k]);
          ++pos;
        }
      }
    }
  }
  for (int s0 = kSC * kRCap; s0 < n_slots; s0 += kRCap) {  // (wave-uniform trip count: a ballot inside)
    const int slot = s0 + tid;
    int2 st = make_int2(0, 0);
    int bid = -1;
    if (slot < n_slots) {
      st = *reinterpret_cast<const int2*>(conn_c + 2 * (int64_t)slot);
      bid = bond_c[slot];
    }
    const bool ok = edge_valid(st.x, st.y, bid, N, p.Vb);
    const int m = fast_div(slot, E, rE), e = slot - m * E;
    const int mo = moloff[m < M ? m : 0];
    const int row = mo + (ok ? st.y : 0);
    int tk = kTick;
    if (ok) {
      tk = atomicAdd(&cnt[row], 1);
      atomicAdd(&thist[bid], 1);
    }
    const unsigned long long okb = __ballot(ok);
    int pos0 = 0;
    if (okb != 0ull) {
      if (lane == (int)__builtin_ctzll(okb)) pos0 = atomicAdd(&misc[1], (int)__builtin_popcountll(okb));
      pos0 = __builtin_amdgcn_readlane(pos0, (int)__builtin_ctzll(okb));
    }
    if (tk < kTick) tick[row * kTick + tk] = (uint16_t)e;
    const int pos = pos0 + (int)__builtin_popcountll(okb & ((1ull << lane) - 1));
    if (ok && pos < kTECapBig)
      elist[pos] = make_uint2((uint32_t)row | ((uint32_t)(mo + st.x) << 8) | ((uint32_t)bid << 16), (uint32_t)e);
  }
  lds_barrier();
  CSTAMP(3);

  // P2: bin = 255 - in-degree (in-degrees above 255 cannot travel: the plan is marked, the encoder refuses it)
  const int my_deg = cnt[tid];
  const int dcl = my_deg > 255 ? 255 : my_deg;
  if (tid < R) {
    if (my_deg > 255) p.header->overflow = 1;
    atomicAdd(&dbins[255 - dcl], 1);
  }
  lds_barrier();
  CSTAMP(4);

  // P3
  if (wave == 0) {
    // rows placed before bin b = rows with a larger in-degree; lane l holds bins 4l .. 4l+3
    const int4 h = reinterpret_cast<const int4*>(dbins)[lane];
    const int s4 = h.x + h.y + h.z + h.w;
    const int ex = wave_incl_scan(s4) - s4;
    reinterpret_cast<int4*>(dstart)[lane] = make_int4(ex, ex + h.x, ex + h.x + h.y, ex + h.x + h.y + h.z);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // jagged-diagonal pointers: the rows with an in-edge of index d are the first S_d = dstart[255 - d] placed rows;
    // lane l holds d = 4l .. 4l+3 = dstart[255 - 4l], [254 - 4l], [253 - 4l], [252 - 4l]
    const int4 s = reinterpret_cast<const int4*>(dstart)[63 - lane];  // elements: dstart[252-4l], [253-4l], [254-4l], [255-4l]
    const int t4 = s.x + s.y + s.z + s.w;
    const int incl = wave_incl_scan(t4);
    const int jx = incl - t4;
    reinterpret_cast<int4*>(jdp)[lane] = make_int4(jx, jx + s.w, jx + s.w + s.z, jx + s.w + s.z + s.y);
    if (lane == 63) { jdp[kRCap] = incl; jdp[kRCap + 1] = incl; }
  } else if (wave == 1) {
    // groups of <= 4 edges per type; lane l holds types 4l .. 4l+3
    const int4 n = reinterpret_cast<const int4*>(thist)[lane];
    const int4 ng = make_int4((n.x + 3) >> 2, (n.y + 3) >> 2, (n.z + 3) >> 2, (n.w + 3) >> 2);
    const int s4 = ng.x + ng.y + ng.z + ng.w;
    const int incl = wave_incl_scan(s4);
    const int ex = incl - s4;
    reinterpret_cast<int4*>(tgb)[lane] = make_int4(ex, ex + ng.x, ex + ng.x + ng.y, ex + ng.x + ng.y + ng.z);
    // run table: the groups of every type that has groups, in type order, cut into runs of <= gmax groups (+ end;
    // encoder_layout.h).  The encoder's waves take runs from it one at a time (an LDS counter), so the message phase is
    // balanced dynamically - given enough runs.
    const int ngrp_all = __builtin_amdgcn_readlane(incl, 63);
    int gmax = (ngrp_all + kTRunTarget - 1) / kTRunTarget;
    gmax = gmax < 2 ? 2 : gmax;
    const float rgm = 1.0f / (float)gmax;
    const int4 nr = make_int4(fast_div(ng.x + gmax - 1, gmax, rgm), fast_div(ng.y + gmax - 1, gmax, rgm),
                              fast_div(ng.z + gmax - 1, gmax, rgm), fast_div(ng.w + gmax - 1, gmax, rgm));
    const int r4 = nr.x + nr.y + nr.z + nr.w;
    const int rincl = wave_incl_scan(r4);
    int ri = rincl - r4;
    uint16_t* runs = reinterpret_cast<uint16_t*>(rec + trec_runs_off(p.Vb, p.ecap));
    for (int i = 0; i < nr.x; ++i) runs[ri++] = (uint16_t)(ex + i * gmax);
    for (int i = 0; i < nr.y; ++i) runs[ri++] = (uint16_t)(ex + ng.x + i * gmax);
    for (int i = 0; i < nr.z; ++i) runs[ri++] = (uint16_t)(ex + ng.x + ng.y + i * gmax);
    for (int i = 0; i < nr.w; ++i) runs[ri++] = (uint16_t)(ex + ng.x + ng.y + ng.z + i * gmax);
    if (lane == 63) {
      runs[rincl] = (uint16_t)incl;
      *reinterpret_cast<uint16_t*>(rec + kTRecNrun) = (uint16_t)rincl;
      misc[0] = incl;
    }
  }
  lds_barrier();
  CSTAMP(5);

  // P4
  int pos = tid;  // rows beyond the chunk keep their thread order behind the R real rows
  if (tid < R) pos = dstart[255 - dcl] + atomicAdd(&dcur[255 - dcl], 1);
  place[tid] = pos;
  {
    const int my_id = idl[tid];  // -1: a slack row; out-of-range ids (incl. negative) read as a zero row in the encoder
    r_rowatom[pos] = my_id;
    r_poolrow[tid] = (uint16_t)(pos | (my_id > 0 ? 0x8000 : 0));
    r_rowdeg[pos] = (uint16_t)my_deg;
    if ((pos & 15) == 0) r_tilemax[pos >> 4] = (unsigned char)dcl;  // descending in-degrees: a tile's first row has its maximum
    r_jd[tid] = (uint16_t)jdp[tid];
    if (tid < 2) r_jd[kRCap + tid] = (uint16_t)jdp[kRCap];
    if (pos == 0) {
      r_cw[0] = (uint16_t)misc[0];
      r_cw[1] = (uint16_t)jdp[kRCap];
      r_cw[2] = (uint16_t)dcl;
    }
  }
  {
    // the groups of type `tid`: x = type | edges << 8 | groups of this type from this one on << 24; unused edge lanes
    // write to the dump slot
    const int n_t = thist[tid];
    const int ng = (n_t + 3) >> 2, gb = tgb[tid];
    const uint32_t dump = (uint32_t)tmsg_key(p.ecap) * 0x10001u;
    for (int jg = 0; jg < ng; ++jg) {
      const int c = n_t - 4 * jg;
      grp[gb + jg] = make_uint4((uint32_t)tid | ((uint32_t)(c < 4 ? c : 4) << 8) | ((uint32_t)(ng - jg) << 24), 0u, dump, dump);
    }
  }
  lds_barrier();
  CSTAMP(6);

  // P5: the listed edges, two per thread and turn, stage by stage (the LDS operations of both are in flight together)
  {
    auto below = [](uint4 a, uint32_t ue) {
      return (int)(((a.x & 0xffffu) < ue) + ((a.x >> 16) < ue) + ((a.y & 0xffffu) < ue) + ((a.y >> 16) < ue) +
                   ((a.z & 0xffffu) < ue) + ((a.z >> 16) < ue) + ((a.w & 0xffffu) < ue) + ((a.w >> 16) < ue));
    };
    const int nval = misc[1] < kTECapBig ? misc[1] : kTECapBig;
    for (int i0 = 0; i0 < nval; i0 += 2 * kRCap) {
      bool ok[2];
      uint2 en[2];
      int row[2], srow[2], deg[2], ix[2], gb[2], rank[2];
      uint4 ta[2], tb[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + k * kRCap + tid;
        ok[k] = i < nval;
        en[k] = elist[ok[k] ? i : 0];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int rt = (int)(en[k].x & 0xffu), bid = (int)(en[k].x >> 16);
        deg[k] = cnt[rt];
        ta[k] = reinterpret_cast<const uint4*>(tick)[2 * rt];
        tb[k] = reinterpret_cast<const uint4*>(tick)[2 * rt + 1];
        row[k] = place[rt];
        srow[k] = place[(en[k].x >> 8) & 0xffu];
        gb[k] = tgb[bid];
        ix[k] = atomicAdd(&tcur[ok[k] ? bid : kTVbMax + lane], 1);  // (a spare word per lane where there is no edge: no branch)
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        rank[k] = below(ta[k], en[k].y) + below(tb[k], en[k].y);  // empty tickets (65535) never count
        if (deg[k] > kTick) {
          // a hub: the tickets beyond the list were not kept - count the earlier valid edge slots of the molecule with
          // the same target (the molecule and the target atom come back out of the logical row)
          const int rt = (int)(en[k].x & 0xffu);
          int lo = 0, hi = M - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (moloff[mid] <= rt) lo = mid; else hi = mid - 1;
          }
          const int t = rt - moloff[lo];
          const int32_t* cm = conn_c + 2 * (int64_t)lo * E;
          const int32_t* bm = bond_c + (int64_t)lo * E;
          rank[k] = 0;
          for (int e2 = 0; e2 < (int)en[k].y; ++e2) {
            const int2 s2 = *reinterpret_cast<const int2*>(cm + 2 * e2);
            rank[k] += (s2.y == t && edge_valid(s2.x, s2.y, bm[e2], N, p.Vb)) ? 1 : 0;
          }
        }
        // jagged diagonal: rank-th in-edge of the placed row (rank > 255 only in a plan that is marked overflowed)
        rank[k] = jdp[rank[k] < 256 ? rank[k] : 255];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (ok[k]) {
          unsigned char* ge = reinterpret_cast<unsigned char*>(&grp[gb[k] + (ix[k] >> 2)]);
          ge[4 + (ix[k] & 3)] = (unsigned char)srow[k];
          reinterpret_cast<uint16_t*>(ge + 8)[ix[k] & 3] = (uint16_t)tmsg_key(rank[k] + row[k]);
        }
      }
    }
  }
  lds_barrier();
  CSTAMP(9);

  // P6
  {
    const int ngrp = misc[0];
    uint4* r_grp = reinterpret_cast<uint4*>(rec + kTRecGrp);
    for (int i = tid; i < ngrp; i += kRCap) r_grp[i] = grp[i];
  }
  CSTAMP(15);
  lds_barrier();  // the LDS tables are rebuilt by the next chunk of this workgroup (rare)
  }
}
#undef CSTAMP

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
    plan_chunks_typed_kernel<<<pp.nwg * pp.grid_sub, kRCap, sizeof(uint4) * tgrp_cap(pp.Vb, pp.ecap), s>>>(pp);
  else
    plan_chunks_kernel<<<pp.nwg * pp.grid_sub, kRCap, 0, s>>>(pp);
  return check_launch("plan_chunks");
}

}  // namespace enc
}  // namespace impnn
