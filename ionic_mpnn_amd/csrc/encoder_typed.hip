// Fused message-passing encoder, per-bond-type form (mode 2 "f32t"), gfx950 / MI355X: encode() of
// train_viscosity.py:166-187 and train_melting_point.py:152-171 up to and including GlobalSumPool, both ions in
// one launch, atom_dim 32, ANY bond_dim (K = 8 of the viscosity model, K = D^2 = 1024 of the melting-point model).
//
// Same frame as encoder_fused.hip - persistent 1024-thread workgroups, chunks of <= 256 packed atom rows whose
// graph structure arrives as a record built by the plan kernels, node state h in LDS for all S steps, GatedUpdate
// as transposed f32-MFMA GEMMs with atoms on the MFMA N dimension - but the message is computed the way the
// reference orders it (models/layers.py:108-112): A_e = sum_k bond_state[e,k] W[k], then m_e = A_e h[src_e].
// bond_state is always an Embedding lookup (train_viscosity.py:172), so A_e is one of Vb matrices
// A[v] = sum_k bond_table[v,k] W[k], prepared once per weight version (encoder_plan.hip: typed_image).
//
//   message phase (all 16 waves): the chunk's valid edges are grouped by bond type in groups of <= 4; a group is
//     16 x v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks: 8 feature quads x 2 halves of k), exact f32
//     products, the edge operand broadcast inside the instruction (CBSZ/ABID), the matrix operand straight from L2
//     once per type and chunk-step.  Messages are written to an LDS buffer in jagged-diagonal order.
//   atom phase (one 16-atom tile per wave): agg[a] = sum of the row's in-edge messages in edge-slot order (the
//     reference's sequential scatter_nd, models/layers.py:78-82; slot(row, d) = jdptr[d] + row, so a tile's reads
//     of one d are 16 consecutive slots: conflict-free), then GatedUpdate (models/layers.py:142-156) exactly as in
//     encoder_fused.hip's f32 mode, h updated in place.
//
// MFMA work per row is 12 D^2 (update) + 2 D^2 per in-edge, against (12 + 2 K) D^2 per row in the pull form:
// 2.7 kflop instead of 28.7 kflop per row for the message at bond_dim 8 and 1.7 in-edges per row.
#include "encoder_device.h"
#include "encoder_layout.h"

namespace impnn {
namespace enc {

namespace {

constexpr int kOffUpd = 0;
constexpr int off_h(bool x3) { return kOffUpd + (x3 ? kXUpdLds : kTUpdLds); }
// hs: the LDS row stride of h - kTHS (36: rows spread over the banks), or 32 where mode 3 meets 640-edge chunks (explicit-
// hydrogen shapes, E > 512): the three-plane update image and 642 message slots leave no room for the padding
constexpr int kTHSBig3 = 32;
constexpr int off_msg(bool x3, int hs = kTHS) { return off_h(x3) + kRCap * hs; }
constexpr int off_rec(bool x3, int ecap, int hs = kTHS) { return off_msg(x3, hs) + tmsg_floats(ecap); }
// then: the record (trec_lds_bytes(Vb, ecap)) and, when it fits, the atom table ((Va + 1) rows of kTAS floats)
constexpr size_t lds_fixed_bytes(bool x3, int Vb, int ecap, int hs = kTHS) {
  return sizeof(float) * (size_t)off_rec(x3, ecap, hs) + trec_lds_bytes(Vb, ecap);
}
static_assert(lds_fixed_bytes(true, kTVbMax, kTECap) <= 160 * 1024 && lds_fixed_bytes(false, kTVbMax, kTECapBig) <= 160 * 1024 &&
                  lds_fixed_bytes(true, kTVbMax, kTECapBig, kTHSBig3) <= 160 * 1024,
              "LDS budget");

// ---- mode 3 ("f32x3"): f32 GEMM products on the bf16 matrix pipe without narrowing them.  An f32 value is the exact
// sum of three bf16 terms (bf16 keeps fp32's exponent; 3 x 8 significant bits): b0 = the upper half of the word, b1 =
// the upper half of the exact residual x - b0, b2 = what is left (<= 8 bits).  All nine cross products of two such
// triples are exact in the MFMA's f32 accumulator, so a GEMM differs from the f32-MFMA one only in the order the
// (exact) products are added.  v_mfma_f32_16x16x32_bf16 is 16 cycles for 32 k; nine of them replace eight
// v_mfma_f32_16x16x4_f32 of 32 cycles - on a pipe that, unlike the f32 one, co-executes with the VALU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct B3 {
  bf16x8 p0, p1, p2;
};
// Non-finite operands: a NaN stays a NaN through its first term.  An INFINITY splits into (inf, inf - inf = nan, nan), so
// every product it meets is NaN, where exact f32 arithmetic has w * inf = +-inf and a saturating gate may even turn that
// back into a finite number (sigmoid(+inf) = 1).  This cannot be repaired inside the split - with (inf, 0, 0) the terms
// a1 * inf of weights whose residual a1 is exactly 0 are NaN again - so mode 3 is the more conservative of the two: the
// rows it reports non-finite are a superset of mode 2's, and a finite row equals mode 2's up to summation order
// (tests/test_gpu_encoder.py::test_f32x3_propagates_nan_and_inf_like_f32t; measured: guarding the split costs 4 us per
// 4096-pair launch and still leaves the 0 * inf terms).
// two values -> their three packed bf16 pairs (5.5 VALU per value: and, sub, and, sub per value; three perms per pair)
__device__ __forceinline__ void split_pair(float x, float y, unsigned& w0, unsigned& w1, unsigned& w2) {
  const unsigned xb = __builtin_bit_cast(unsigned, x), yb = __builtin_bit_cast(unsigned, y);
  const float x1 = x - __builtin_bit_cast(float, xb & 0xffff0000u), y1 = y - __builtin_bit_cast(float, yb & 0xffff0000u);
  const unsigned x1b = __builtin_bit_cast(unsigned, x1), y1b = __builtin_bit_cast(unsigned, y1);
  const float x2 = x1 - __builtin_bit_cast(float, x1b & 0xffff0000u), y2 = y1 - __builtin_bit_cast(float, y1b & 0xffff0000u);
  // pack the upper halves: low 16 bits <- x, high 16 bits <- y   (v_perm_b32: bytes [y3 y2 x3 x2])
  w0 = __builtin_amdgcn_perm(yb, xb, 0x07060302u);
  w1 = __builtin_amdgcn_perm(y1b, x1b, 0x07060302u);
  w2 = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, y2), __builtin_bit_cast(unsigned, x2), 0x07060302u);
}
__device__ __forceinline__ B3 split8x3(f32x4 a, f32x4 b) {
  union {
    bf16x8 v;
    unsigned u[4];
  } q0, q1, q2;
  split_pair(a[0], a[1], q0.u[0], q1.u[0], q2.u[0]);
  split_pair(a[2], a[3], q0.u[1], q1.u[1], q2.u[1]);
  split_pair(b[0], b[1], q0.u[2], q1.u[2], q2.u[2]);
  split_pair(b[2], b[3], q0.u[3], q1.u[3], q2.u[3]);
  B3 r;
  r.p0 = q0.v;
  r.p1 = q1.v;
  r.p2 = q2.v;
  return r;
}
__device__ __forceinline__ f32x4 mfmab(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// c0 += W0 b, c1 += W1 b (W: three planes at blk, blk + 512, blk + 1024): all nine cross products, smallest first, the
// two accumulator chains interleaved (a dependent v_mfma_f32_16x16x32_bf16 waits ~4 cycles for its predecessor).
__device__ __forceinline__ void mma9x2(f32x4& c0, f32x4& c1, const __bf16* blk0, const __bf16* blk1, int lane, const B3& b) {
  const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(blk0 + lane * 8);
  const bf16x8 a01 = *reinterpret_cast<const bf16x8*>(blk0 + 512 + lane * 8);
  const bf16x8 a02 = *reinterpret_cast<const bf16x8*>(blk0 + 1024 + lane * 8);
  const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(blk1 + lane * 8);
  const bf16x8 a11 = *reinterpret_cast<const bf16x8*>(blk1 + 512 + lane * 8);
  const bf16x8 a12 = *reinterpret_cast<const bf16x8*>(blk1 + 1024 + lane * 8);
  c0 = mfmab(a02, b.p2, c0);  c1 = mfmab(a12, b.p2, c1);
  c0 = mfmab(a01, b.p2, c0);  c1 = mfmab(a11, b.p2, c1);
  c0 = mfmab(a02, b.p1, c0);  c1 = mfmab(a12, b.p1, c1);
  c0 = mfmab(a00, b.p2, c0);  c1 = mfmab(a10, b.p2, c1);
  c0 = mfmab(a02, b.p0, c0);  c1 = mfmab(a12, b.p0, c1);
  c0 = mfmab(a01, b.p1, c0);  c1 = mfmab(a11, b.p1, c1);
  c0 = mfmab(a00, b.p1, c0);  c1 = mfmab(a10, b.p1, c1);
  c0 = mfmab(a01, b.p0, c0);  c1 = mfmab(a11, b.p0, c1);
  c0 = mfmab(a00, b.p0, c0);  c1 = mfmab(a10, b.p0, c1);
}

// v_mfma_f32_4x4x1_16b_f32 with CBSZ = 3: the 16 blocks form two groups of 8 (lanes 0-31 / 32-63) and every block of a
// group takes its A operand from block ABID of the group.  So ONE register carries the A operands of EIGHT
// instructions: the 4 lanes of block b hold what instruction ABID = b multiplies.
template <int ABID>
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 3, ABID, 0);
}

// One group of the message phase: its record entry and the A operands of its 16 MFMAs - lane l = 32 kh + 4 b + i holds
// h[source row of edge i][16 kh + 2 b] and [.. + 2 b + 1]: one ds_read_b64 per lane, no lane idle, two registers.
typedef float f32x2v __attribute__((ext_vector_type(2)));
struct Grp {
  uint4 ge;
  f32x2v aq;
};
// One type run: the matrix rows of this lane, and the run's first group (requested as soon as the run is known).
struct Run {
  f32x4 bq[4];
  Grp first;
};

// Where the loads of the NEXT step are issued (diagnostics builds may override):
//   kPfWhere   0: the step's update image at the top of its own message phase; 1: during the previous step's atom phase,
//              in front of the GEMMs; 2: behind the GEMMs (registers are free there, the vector-memory path still idle)
//   kRunsEarly how many of a wave's two fixed type runs (0-2) have their matrix rows requested in front of the previous
//              step's GEMMs (16 VGPRs each through the GEMMs); the others at the top of the message phase
#ifndef IMPNN_T_PF
#define IMPNN_T_PF 0
#endif
#ifndef IMPNN_T_EARLY
#define IMPNN_T_EARLY 0
#endif
constexpr int kPfWhere = IMPNN_T_PF, kRunsEarly = IMPNN_T_EARLY;

template <bool STAMPS, bool X3, int HS = kTHS>
__global__ __launch_bounds__(kThreads, kThreads / 256) void encoder_typed_kernel(TEncParams p) {
  extern __shared__ __align__(16) float smem[];
  constexpr int kUpdLds = X3 ? kXUpdLds : kTUpdLds, kUpdSlot = X3 ? kXUpdSlot : kTUpdSlot;
  constexpr int kNPf = X3 ? 3 : 2;  // 16-byte loads per thread that carry one update image
  float* const wupd = smem + kOffUpd;
  float* const wvec = wupd + (X3 ? kXVecFloatOff : kTVecFloatOff);
  float* const hbuf = smem + off_h(X3);
  float* const msg = smem + off_msg(X3, HS);
  const int zero_slot = p.ecap + 1;  // (dump slot: p.ecap)
  unsigned char* const recl = reinterpret_cast<unsigned char*>(smem + off_rec(X3, p.ecap, HS));
  const int rec_lds = p.rec_lds;                                 // trec_lds_bytes(Vb)
  float* const atab = reinterpret_cast<float*>(recl + rec_lds);  // Va rows + one zero row (when it fits)
  const uint16_t* const r_rowdeg = reinterpret_cast<const uint16_t*>(recl + kTRecRowdeg);
  const unsigned char* const r_tilemax = recl + kTRecTilemax;
  const uint16_t* const r_moloff = reinterpret_cast<const uint16_t*>(recl + kTRecMoloff);
  const uint16_t* const r_molrows = reinterpret_cast<const uint16_t*>(recl + kTRecMolrows);
  const uint16_t* const r_poolrow = reinterpret_cast<const uint16_t*>(recl + kTRecPoolrow);
  const int32_t* const r_rowatom = reinterpret_cast<const int32_t*>(recl + kTRecRowatom);
  const uint16_t* const r_jdptr = reinterpret_cast<const uint16_t*>(recl + kTRecJdptr);
  const uint16_t* const r_runs = reinterpret_cast<const uint16_t*>(recl + trec_runs_off(p.Vb, p.ecap));
  int* const run_ctr = reinterpret_cast<int*>(recl + kTRecCounts + 8);  // next dynamically assigned type run
  const uint4* const r_grp = reinterpret_cast<const uint4*>(recl + kTRecGrp);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a = lane & 15, q = lane >> 4;
  // diagnostics build only (impnn_debug_set_stamp_buffer): the stamp bookkeeping costs ~10 VGPRs
  unsigned long long* stamp = STAMPS && p.stamps ? p.stamps + (size_t)blockIdx.x * 32 : nullptr;
  if (stamp && tid == 0) stamp[0] = __builtin_amdgcn_s_memtime();

  // The workspace must hold a typed plan made for this launch geometry (impnn_encoder_plan with the same
  // arguments): anything else would be read at wrong offsets.  A mismatch - or a batch the plan found it cannot
  // cut into chunks (PlanHeader::overflow) - poisons the outputs instead.
  {
    const PlanHeader hd = *p.header;
    if (hd.magic != kPlanMagic || hd.kind != 1 || hd.nwg != (int)gridDim.x || hd.max_sub != p.max_sub ||
        hd.B != p.B || hd.n_ions != p.n_ions || hd.overflow != 0) {
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

  if (p.atab_lds)
    for (int t = tid; t < (p.Va + 1) * (kTAS / 4); t += kThreads)  // kTAS == kD: a verbatim copy + one zero row
      st4(atab + 4 * t, t < p.Va * (kD / 4) ? ld4(p.atom_table + 4 * t) : f32x4{0.f, 0.f, 0.f, 0.f});
  if (tid < kD) msg[zero_slot * kD + tid] = 0.f;  // never written again: what a row reads beyond its in-degree
  unsigned long long t_pro = 0, t_steps = 0, t_pool = 0, t_mark = 0, t_msg = 0;
  if (stamp && tid == 0) t_mark = __builtin_amdgcn_s_memtime();

  // ---- GlobalSumPool (models/layers.py:161-164), as in encoder_fused.hip: eight lanes share one
  //      (molecule, 4 features), ascending rows, then a fixed 3-step butterfly on the DPP network.
  auto pool = [&](int M, int m0, int g) {
    float* out_g = p.pooled[g];
    for (int t0 = 0; t0 < M * 64; t0 += kThreads) {
      const int t = t0 + tid;
      const int part = t & 7, f4 = (t >> 3) & 7, m = t >> 6;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (m < M) {
        const int nr = r_molrows[m], mo = r_moloff[m];
        for (int n = part; n < nr; n += 8) {
          const int pr = r_poolrow[mo + n];
          if (pr & 0x8000) acc += ld4(hbuf + (pr & 0xff) * HS + 4 * f4);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = acc[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xb1, 0xf, 0xf, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4e, 0xf, 0xf, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
        acc[i] = v;
      }
      if (m < M && part == 0) st4(out_g + (int64_t)(m0 + m) * kD + 4 * f4, acc);
    }
  };
  // h0 = atom_table[atom id of the placed row]: 4 threads per row, 2 x 16 B each; slack rows: zeros
  auto fill_h0 = [&]() {
    const int row = tid >> 2, sub = tid & 3;
    const int id = r_rowatom[row];
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if ((unsigned)id < (unsigned)p.Va) {
      if (p.atab_lds) {
        v0 = ld4(atab + id * kTAS + 8 * sub);
        v1 = ld4(atab + id * kTAS + 8 * sub + 4);
      } else {
        v0 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub);
        v1 = ld4(p.atom_table + (int64_t)id * kD + 8 * sub + 4);
      }
    }
    st4(hbuf + row * HS + 8 * sub, v0);
    st4(hbuf + row * HS + 8 * sub + 4, v1);
  };

  if (p.S == 0) {  // no message passing: pooled = GlobalSumPool(Embedding(atom ids)); kept apart from the step machinery
    for (int c = c_begin; c < c_end; ++c) {
      const unsigned char* rc = p.rec + (size_t)c * kTRecBytes;
      if (8 * tid < rec_lds) reinterpret_cast<uint2*>(recl)[tid] = reinterpret_cast<const uint2*>(rc)[tid];
      const int4 dsc = reinterpret_cast<const int4*>(p.desc)[c];
      lds_barrier();
      fill_h0();
      lds_barrier();
      pool(__builtin_amdgcn_readfirstlane(dsc.y), __builtin_amdgcn_readfirstlane(dsc.x),
           __builtin_amdgcn_readfirstlane(dsc.w) >> 16);
      lds_barrier();
    }
    return;
  }

  // The record of the NEXT chunk travels in registers while the current chunk runs.
  const bool rec_big = rec_lds > kTRecPart1;  // workgroup-uniform
  const unsigned char* rec_c = p.rec + (size_t)c_begin * kTRecBytes;
  uint2 rec_n8 = reinterpret_cast<const uint2*>(rec_c)[tid];
  uint32_t rec_n4 = rec_big ? reinterpret_cast<const uint32_t*>(rec_c + kTRecPart1)[tid] : 0u;
  int4 dsc_next = reinterpret_cast<const int4*>(p.desc)[c_begin];

  // message-phase lane roles (see the message phase below)
  const uint32_t* const grp_x = reinterpret_cast<const uint32_t*>(r_grp);
  const int kh = lane >> 5, f = lane & 31;
  const int boff = kh * 512 + f * 4;
  const int ysh = 8 * (lane & 3), zsh = 16 * kh;
  const int acol = 16 * kh + 2 * ((lane >> 2) & 7);  // first of the two h columns this lane feeds (see Grp)

  for (int c = c_begin; c < c_end; ++c) {
    // ---- chunk prologue: descriptor, record -> LDS, h0 = atom_table[atom ids] (only without the LDS table)
    const int4 dsc = dsc_next;
    const int m0 = __builtin_amdgcn_readfirstlane(dsc.x), M = __builtin_amdgcn_readfirstlane(dsc.y);
    const int rg = __builtin_amdgcn_readfirstlane(dsc.w);
    const int R = rg & 0xffff, g = rg >> 16;
    const int ntiles = (R + 15) >> 4;
    const float* upd_g = p.upd[g];
    const float* tmat_g = p.tmat[g];
    if (8 * tid < rec_lds) reinterpret_cast<uint2*>(recl)[tid] = rec_n8;
    if (rec_big && kTRecPart1 + 4 * tid < rec_lds) reinterpret_cast<uint32_t*>(recl + kTRecPart1)[tid] = rec_n4;
    {
      const int cn = (c + 1) < c_end ? (c + 1) : c;  // clamped: unconditional loads
      const unsigned char* rn = p.rec + (size_t)cn * kTRecBytes;
      rec_n8 = reinterpret_cast<const uint2*>(rn)[tid];
      if (rec_big) rec_n4 = reinterpret_cast<const uint32_t*>(rn + kTRecPart1)[tid];
      dsc_next = reinterpret_cast<const int4*>(p.desc)[cn];
    }
    lds_barrier();
    if (!p.atab_lds) fill_h0();
    if (tid == 0) *run_ctr = 2 * kWaves;  // runs 0 .. 2 kWaves - 1 are assigned statically (two per wave)
    lds_barrier();
    if (stamp && tid == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_pro += t - t_mark;
      t_mark = t;
    }
    const int nrun = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const uint16_t*>(recl + kTRecNrun));

    // Two type runs per wave are in flight at any time (P, Q).  The first two of every step are fixed - runs `wave` and
    // `wave + 16` of the chunk's run table, the same in all S steps - so their group range and bond type are resolved
    // once per chunk, and their matrix rows for step s + 1 are requested during step s's atom phase, in front of the
    // GEMMs: the vector-memory path, which bounds the message phase (~280 KB of matrix rows per chunk-step through one
    // CU's L2 port), has nothing else to do there.  The step's update image travels the same way.
    Run P, Q;
    int gP = 0, nP = 0, gQ = 0, nQ = 0;
    bool haveP = false, haveQ = false;
    const bool sP = wave < nrun, sQ = wave + kWaves < nrun;
    const int sgP = __builtin_amdgcn_readfirstlane(r_runs[sP ? wave : 0]);
    const int snP = __builtin_amdgcn_readfirstlane(r_runs[(sP ? wave : 0) + 1]) - sgP;
    const int sgQ = __builtin_amdgcn_readfirstlane(r_runs[sQ ? wave + kWaves : 0]);
    const int snQ = __builtin_amdgcn_readfirstlane(r_runs[(sQ ? wave + kWaves : 0) + 1]) - sgQ;
    // (matrix offset of the run's type; `have` false: four cache lines of type 0 that nobody uses - see `fetch`)
    const int soP = sP ? (__builtin_amdgcn_readfirstlane(grp_x[4 * sgP]) & 0xff) * kTMatFloats : 0;
    const int soQ = sQ ? (__builtin_amdgcn_readfirstlane(grp_x[4 * sgQ]) & 0xff) * kTMatFloats : 0;
    const int lboffP = sP ? boff : 0, lboffQ = sQ ? boff : 0;
    f32x4 pf[kNPf];
    auto fetch_pf = [&](int s_) {  // update image of step s_
#pragma unroll
      for (int i = 0; i < kNPf; ++i) pf[i] = ld4(upd_g + (int64_t)s_ * kUpdSlot + 4 * (tid + i * kThreads));
    };
    auto fetch_P = [&](int s_) {  // matrix rows of the wave's first / second fixed run in step s_
      const float* tm = tmat_g + (size_t)s_ * p.Vb * kTMatFloats;
#pragma unroll
      for (int i = 0; i < 4; ++i) P.bq[i] = ld4(tm + soP + lboffP + i * 128);
    };
    auto fetch_Q = [&](int s_) {
      const float* tm = tmat_g + (size_t)s_ * p.Vb * kTMatFloats;
#pragma unroll
      for (int i = 0; i < 4; ++i) Q.bq[i] = ld4(tm + soQ + lboffQ + i * 128);
    };
    // step 0: everything that later steps request during the previous atom phase is requested here
    if (kPfWhere != 0) fetch_pf(0);
    if (kRunsEarly >= 1) fetch_P(0);
    if (kRunsEarly >= 2) fetch_Q(0);
    for (int s = 0; s < p.S; ++s) {
      const bool g0 = p.atab_lds && s == 0;  // step 0 reads h0 = atom_table[id] straight from the LDS table
      const float* tm_s = tmat_g + (size_t)s * p.Vb * kTMatFloats;
      // ---- message phase: m_e = A[type_e] h[src_e], one group of <= 4 edges of one bond type per 16 MFMAs.
      // v_mfma_f32_4x4x1_16b: 16 independent 4x4 blocks.  Block b < 8 accumulates, for feature quad b, the k < 16
      // half of the dot products of the group's 4 edges; block 8 + b the k >= 16 half:
      //     D_b[i][j] += h[src_i][k] * A[type][4*(b & 7) + j][k],   k = 16*(b >> 3) + step.
      // The edge operand h[src_i][k] comes from ONE block per half-wave (lanes 0-3 / 32-35) and is broadcast to the
      // other seven by CBSZ/ABID: 8 lanes read LDS, not 64.  The matrix operand of lane l is half a row of A[type]
      // (16 floats = 4 x 16 B, straight from L2).  All groups of a type run on one wave, so a type's 4 KB are
      // fetched once per chunk-step.  The two k-halves meet in one v_permlane32_swap per edge pair; the sums go to
      // the LDS message buffer in jagged-diagonal order.
      // The f32 MFMA shares its issue port with the VALU (tools/ubench: no co-execution), so every VALU instruction in
      // this loop costs matrix time: the plan hands over ready-made LDS keys (message slot incl. swizzle; unused edge
      // lanes point at a dump slot, so the stores are unconditional) and the loop body is branch-free.
      __builtin_amdgcn_s_setprio(2);
      {
        const float* const abase = (g0 ? atab : hbuf) + acol;
        const int astride = g0 ? kTAS : HS;
        auto load_group = [&](int e, Grp& G) {
          G.ge = r_grp[e];
          int src = __builtin_amdgcn_ubfe(G.ge.y, ysh, 8);
          if (g0) {
            const int id = r_rowatom[src];
            src = (unsigned)id < (unsigned)p.Va ? id : p.Va;
          }
          G.aq = *reinterpret_cast<const f32x2v*>(abase + src * astride);
        };
        auto compute = [&](const Grp& G, const f32x4 (&bq)[4]) {
          f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#ifndef IMPNN_DIAG_NO_MSG_MFMA  // (diagnostics builds only: wrong results)
          // instruction k (= h column 16 kh + k): A from register k & 1, block k >> 1; B = the lane's matrix element k.
          // (Two accumulator chains; four were measured: +4 % kernel time, the registers cost more than the s_nops.)
          acc0 = mfma1<0>(G.aq[0], bq[0][0], acc0);
          acc1 = mfma1<0>(G.aq[1], bq[0][1], acc1);
          acc0 = mfma1<1>(G.aq[0], bq[0][2], acc0);
          acc1 = mfma1<1>(G.aq[1], bq[0][3], acc1);
          acc0 = mfma1<2>(G.aq[0], bq[1][0], acc0);
          acc1 = mfma1<2>(G.aq[1], bq[1][1], acc1);
          acc0 = mfma1<3>(G.aq[0], bq[1][2], acc0);
          acc1 = mfma1<3>(G.aq[1], bq[1][3], acc1);
          acc0 = mfma1<4>(G.aq[0], bq[2][0], acc0);
          acc1 = mfma1<4>(G.aq[1], bq[2][1], acc1);
          acc0 = mfma1<5>(G.aq[0], bq[2][2], acc0);
          acc1 = mfma1<5>(G.aq[1], bq[2][3], acc1);
          acc0 = mfma1<6>(G.aq[0], bq[3][0], acc0);
          acc1 = mfma1<6>(G.aq[1], bq[3][1], acc1);
          acc0 = mfma1<7>(G.aq[0], bq[3][2], acc0);
          acc1 = mfma1<7>(G.aq[1], bq[3][3], acc1);
#else
          acc0[0] = G.aq[0] + bq[0][0] + bq[3][3];
          acc1[1] = G.aq[1] + bq[1][1] + bq[2][2];
#endif
          acc0 += acc1;
          // element i of lane l: edge i, feature l & 31, k-half l >> 5.
          // v_permlane32_swap x, y: lanes 32-63 of x <-> lanes 0-31 of y.  Afterwards x = {x.lo, y.lo},
          // y = {x.hi, y.hi}, so x + y is edge 0 (2) complete in lanes 0-31 and edge 1 (3) in lanes 32-63.
          // (Inline asm: the compiler's builtin for this gfx950 instruction folded its two operands into one here.
          //  The s_nop covers the VALU-write -> permlane-read wait states the assembler does not insert.)
          float x0 = acc0[0], x1 = acc0[1], x2 = acc0[2], x3 = acc0[3];
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3"
                       : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
          msg[__builtin_amdgcn_ubfe(G.ge.z, zsh, 16) ^ f] = x0 + x1;
          msg[__builtin_amdgcn_ubfe(G.ge.w, zsh, 16) ^ f] = x2 + x3;
        };
        // A run's groups, software-pipelined: the operands of group e + 1 are requested from LDS in front of group e's
        // MFMAs (two buffers, loop unrolled by two so that neither is ever copied); the first group's were requested
        // when the run was taken.
        auto run_type = [&](int g_, int n_, Run& Rn) {
          const int end = g_ + n_;
          Grp T;
          int e = g_;
          while (true) {
            if (e + 1 < end) load_group(e + 1, T);
            compute(Rn.first, Rn.bq);
            if (++e >= end) break;
            if (e + 1 < end) load_group(e + 1, Rn.first);
            compute(T, Rn.bq);
            if (++e >= end) break;
          }
        };
        // Matrix rows of run r -> Rn.bq.  ALWAYS four loads: beyond the last run (`have` false) they read four cache
        // lines of type 0 that nobody uses - the number of vector-memory operations in flight behind any fetch is
        // then the same on every path, so the compiler can wait for a run's rows with a COUNTED s_waitcnt vmcnt(n)
        // and leave the younger fetches in flight (vmcnt retires in order; with conditional loads it must drain).
        auto fetch = [&](bool have, int r, int& g_, int& n_, Run& Rn, const float* tm) {
          const int rc = have ? r : 0;
          g_ = __builtin_amdgcn_readfirstlane(r_runs[rc]);
          n_ = __builtin_amdgcn_readfirstlane(r_runs[rc + 1]) - g_;
          const int type = have ? (__builtin_amdgcn_readfirstlane(grp_x[4 * g_]) & 0xff) : 0;
          const float* bp = tm + (size_t)type * kTMatFloats + (have ? boff : 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) Rn.bq[i] = ld4(bp + i * 128);
        };
        // dynamically assigned runs (beyond the two static ones per wave) come from an LDS counter: the runs differ in
        // length and in how long their matrix rows take to arrive (a static split left the slowest wave 20 % behind).
        auto next_run = [&]() {
          int r = 0;
          if (lane == 0) r = atomicAdd(run_ctr, 1);
          return __builtin_amdgcn_readfirstlane(r);
        };
        auto take = [&](int r, int& g_, int& n_, Run& Rn) {  // r < nrun
          fetch(true, r, g_, n_, Rn, tm_s);
          load_group(g_, Rn.first);
        };
        // `landed`: the rows are consumed here (an empty asm that reads and rewrites the registers), so the compiler
        // places its s_waitcnt vmcnt HERE.
        auto landed = [](f32x4 (&b)[4]) {
          asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
        };
        // the first two runs of a wave are fixed; the rest comes from the counter.  (Loads in program order on every
        // path: image, P, Q - so that the counted waits below hold.)
        if (kPfWhere == 0) fetch_pf(s);
        if (kRunsEarly < 1) fetch_P(s);
        if (kRunsEarly < 2) fetch_Q(s);
        haveP = sP; gP = sgP; nP = snP;
        haveQ = sQ; gQ = sgQ; nQ = snQ;
        if (haveP) load_group(gP, P.first);
        if (haveQ) load_group(gQ, Q.first);
        // Runs are taken in the order P, Q, P, ... (run `wave + 16` exists only if run `wave` does).  Inside the loop
        // every path issues the same loads in the same order, so each `landed` is a counted wait that leaves the other
        // run's rows in flight; the first counter value beyond the last run leaves the loop through a tail that issues
        // no loads at all.
        if (haveP) {
          if (!haveQ) {
            landed(P.bq);
            run_type(gP, nP, P);
          } else {
            while (true) {
              landed(P.bq);
              run_type(gP, nP, P);
              const int rP = next_run();
              if (rP >= nrun) {
                landed(Q.bq);
                run_type(gQ, nQ, Q);
                break;
              }
              take(rP, gP, nP, P);
              landed(Q.bq);
              run_type(gQ, nQ, Q);
              const int rQ = next_run();
              if (rQ >= nrun) {
                landed(P.bq);
                run_type(gP, nP, P);
                break;
              }
              take(rQ, gQ, nQ, Q);
            }
          }
        }
      }
      // this step's update image -> LDS (ordered by the mid-step barrier)
#pragma unroll
      for (int i = 0; i < kNPf; ++i)
        if (4 * (tid + i * kThreads) < kUpdLds) st4(wupd + 4 * (tid + i * kThreads), pf[i]);
      const bool mstamp = stamp && c == c_begin && s == 1;
      if (mstamp && lane == 0) stamp[16 + wave] = __builtin_amdgcn_s_memtime();
      // Mid-step barrier: every message is in LDS and every read of h by the message phase is done (h is updated in
      // place below); the update image of this step is in place too.
      lds_barrier();
      if (tid == 0) *run_ctr = 2 * kWaves;  // for the next step's message phase (ordered by the end-of-step barrier)
      if (mstamp && tid == 0) stamp[12] = __builtin_amdgcn_s_memtime();
      if (stamp && tid == 0 && c == c_begin && s == 1) t_msg = __builtin_amdgcn_s_memtime();

      // ---- atom phase: one 16-atom tile per wave.  Tile -> wave map as in encoder_fused.hip: waves w, w+4, w+8,
      // w+12 share a SIMD; tiles are ordered heavy -> light, the SIMD groups with one tile fewer take the heaviest.
      int my_tile = -1;
      {
        const int grp = wave & 3, slot = wave >> 2, q4 = ntiles >> 2, r4 = ntiles & 3;
        const int heavy = (4 - r4) * q4;
        if (grp >= r4) {
          if (slot < q4) my_tile = slot * (4 - r4) + (grp - r4);
        } else if (slot <= q4) {
          my_tile = heavy + slot * r4 + grp;
        }
      }
      const bool has_tile = my_tile >= 0 && my_tile < ntiles;
      if (has_tile) {
        const int tile = my_tile;
        const int row = tile * 16 + a;
        // ---- Reduce (models/layers.py:57-83): in-edge messages summed in edge-slot order, four in-edge ranks per
        // round trip: the ranks a row does not have read the slot of zeros (x + 0 = x: no branch, every load of a
        // round is in flight together).
        const int deg = r_rowdeg[row];
        const int maxdeg = __builtin_amdgcn_readfirstlane(r_tilemax[tile]);
        f32x4 agg0 = {0.f, 0.f, 0.f, 0.f}, agg1 = agg0;
        for (int d0 = 0; d0 < maxdeg; d0 += 4) {
          const uint2 jd = *reinterpret_cast<const uint2*>(r_jdptr + d0);  // d0 + 3 <= 258 (u16[260] incl. padding)
          int sl[4];
          sl[0] = d0 + 0 < deg ? (int)(jd.x & 0xffffu) + row : zero_slot;
          sl[1] = d0 + 1 < deg ? (int)(jd.x >> 16) + row : zero_slot;
          sl[2] = d0 + 2 < deg ? (int)(jd.y & 0xffffu) + row : zero_slot;
          sl[3] = d0 + 3 < deg ? (int)(jd.y >> 16) + row : zero_slot;
          f32x4 m0v[4], m1v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = tmsg_off(sl[j], q);
            m0v[j] = ld4(msg + o);
            m1v[j] = ld4(msg + (o ^ 16));  // unit 4 + q of the same slot
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            agg0 += m0v[j];
            agg1 += m1v[j];
          }
        }
        if (mstamp && wave == 1 && lane == 0) stamp[8] = __builtin_amdgcn_s_memtime();
        // (Mode 3, measured and dropped: a static priority for two of a SIMD's four waves, so that they run ahead and the
        //  vector work of one pair overlaps the bf16 MFMAs of the other: +2 % kernel time - the waves that fall behind
        //  then set the length of the phase.)
        __builtin_amdgcn_s_setprio(1);
        if (s + 1 < p.S) {  // in flight under the GEMMs (see above)
          if (kPfWhere == 1) fetch_pf(s + 1);
          if (kRunsEarly >= 1) fetch_P(s + 1);
          if (kRunsEarly >= 2) fetch_Q(s + 1);
        }
        int own = row * HS + (int)(hbuf - smem);
        if (g0) {
          const int id = r_rowatom[row];
          own = ((unsigned)id < (unsigned)p.Va ? id : p.Va) * kTAS + (int)(atab - smem);
        }
        own += 4 * q;
        const f32x4 h0 = ld4(smem + own);
        const f32x4 h1 = ld4(smem + own + 16);

        // ---- gates z, r (models/layers.py:144-147) and candidate (:150-151): out^T = W^T [h | agg]^T
        f32x4 z0 = ld4(wvec + 0 * kD + 4 * q), z1 = ld4(wvec + 0 * kD + 16 + 4 * q);
        f32x4 r0 = ld4(wvec + 1 * kD + 4 * q), r1 = ld4(wvec + 1 * kD + 16 + 4 * q);
        f32x4 t0 = ld4(wvec + 2 * kD + 4 * q), t1 = ld4(wvec + 2 * kD + 16 + 4 * q);
        if constexpr (X3) {
          const __bf16* wb = reinterpret_cast<const __bf16*>(wupd);  // block ((gate*2 + T)*2 + half): 3 x 512 bf16
          const B3 sh = split8x3(h0, h1);
          const B3 sa = split8x3(agg0, agg1);
          mma9x2(z0, z1, wb + ((0 * 2 + 0) * 2 + 0) * 1536, wb + ((0 * 2 + 1) * 2 + 0) * 1536, lane, sh);
          mma9x2(r0, r1, wb + ((1 * 2 + 0) * 2 + 0) * 1536, wb + ((1 * 2 + 1) * 2 + 0) * 1536, lane, sh);
          mma9x2(z0, z1, wb + ((0 * 2 + 0) * 2 + 1) * 1536, wb + ((0 * 2 + 1) * 2 + 1) * 1536, lane, sa);
          mma9x2(r0, r1, wb + ((1 * 2 + 0) * 2 + 1) * 1536, wb + ((1 * 2 + 1) * 2 + 1) * 1536, lane, sa);
          z0 = sigmoid4<false>(z0);
          z1 = sigmoid4<false>(z1);
          const f32x4 rh0 = sigmoid4<false>(r0) * h0;  // :149
          const f32x4 rh1 = sigmoid4<false>(r1) * h1;
          const B3 srh = split8x3(rh0, rh1);
          mma9x2(t0, t1, wb + ((2 * 2 + 0) * 2 + 0) * 1536, wb + ((2 * 2 + 1) * 2 + 0) * 1536, lane, srh);
          mma9x2(t0, t1, wb + ((2 * 2 + 0) * 2 + 1) * 1536, wb + ((2 * 2 + 1) * 2 + 1) * 1536, lane, sa);
        } else {
          // A operands: block ((gate*2 + T)*2 + half)*2 + u of the image, 16 B per lane, lane-linear (conflict-free)
          const float* wl = wupd + 4 * lane;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int hu = half * 2 + u;
              const f32x4 Az0 = ld4(wl + ((0 * 2 + 0) * 4 + hu) * 256);
              const f32x4 Az1 = ld4(wl + ((0 * 2 + 1) * 4 + hu) * 256);
              const f32x4 Ar0 = ld4(wl + ((1 * 2 + 0) * 4 + hu) * 256);
              const f32x4 Ar1 = ld4(wl + ((1 * 2 + 1) * 4 + hu) * 256);
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
          const f32x4 rh0 = sigmoid4<false>(r0) * h0;  // :149
          const f32x4 rh1 = sigmoid4<false>(r1) * h1;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int hu = half * 2 + u;
              const f32x4 Ah0 = ld4(wl + ((2 * 2 + 0) * 4 + hu) * 256);
              const f32x4 Ah1 = ld4(wl + ((2 * 2 + 1) * 4 + hu) * 256);
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
        if (kPfWhere == 2 && s + 1 < p.S) {
          asm volatile("" ::: "memory");  // behind the GEMMs: their registers are free only now
          fetch_pf(s + 1);
        }
        if (mstamp && wave == 1 && lane == 0) stamp[9] = __builtin_amdgcn_s_memtime();
        // ---- blend, LayerNorm, residual  (models/layers.py:153-155); (1-z) h + z t == h + z (t - h)
        f32x4 n0 = z0 * (tanh4<false>(t0) - h0) + h0;
        f32x4 n1 = z1 * (tanh4<false>(t1) - h1) + h1;
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
        const f32x4 gm0 = ld4(wvec + 3 * kD + 4 * q), gm1 = ld4(wvec + 3 * kD + 16 + 4 * q);
        const f32x4 bt0 = ld4(wvec + 4 * kD + 4 * q), bt1 = ld4(wvec + 4 * kD + 16 + 4 * q);
        const f32x4 o0 = n0 * (gm0 * inv) + (bt0 + h0);
        const f32x4 o1 = n1 * (gm1 * inv) + (bt1 + h1);
        st4(hbuf + row * HS + 4 * q, o0);
        st4(hbuf + row * HS + 16 + 4 * q, o1);
        if (mstamp && wave == 1 && lane == 0) stamp[10] = __builtin_amdgcn_s_memtime();
      }
      if (!has_tile && s + 1 < p.S) {  // waves without a tile in this chunk
        if (kPfWhere != 0) fetch_pf(s + 1);
        if (kRunsEarly >= 1) fetch_P(s + 1);
        if (kRunsEarly >= 2) fetch_Q(s + 1);
      }
      if (mstamp && wave == 1 && lane == 0) stamp[11] = __builtin_amdgcn_s_memtime();
      // End-of-step barrier: every row of h is updated, every read of the message buffer and of the update image is
      // done.  (LDS traffic only: the run prefetch above stays in flight across it.)
      lds_barrier();
      if (stamp && c == c_begin && s < 2 && tid == 0) stamp[14 + s] = __builtin_amdgcn_s_memtime();
    }
    if (stamp && tid == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      t_steps += t - t_mark;
      t_mark = t;
    }

    pool(M, m0, g);
    lds_barrier();  // the record / h buffers are rewritten by the next chunk's prologue
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
    stamp[5] = t_msg;
    stamp[7] = __builtin_amdgcn_s_memtime();
  }
}

}  // namespace
}  // namespace enc

bool encoder_typed_supported(int N, int E, int D, int S, int Vb) {
  using namespace enc;
  if (D != kD || S < 0) return false;
  // Any padded shape: what bounds a chunk is what a molecule HOLDS - kept rows <= 256, valid edges <= 512, in-degrees
  // <= 255 (640 valid edges for padded E > 512) - and that is checked per batch by the plan kernels
  // (PlanHeader::overflow), not here.
  if (N < 1 || N > 0xffff || E < 0 || E > 0xffff) return false;
  if (Vb < 1 || Vb > kTVbMax) return false;
  return true;
}

size_t encoder_typed_prepared_bytes(int S, int Vb, bool x3) {
  return enc::typed_prepared_floats(S, Vb, x3) * sizeof(float);
}

int launch_encoder_typed_prepare(const float* weights, const float* bond_table, int K, int S, int Vb, bool x3,
                                 void* prepared, hipStream_t s) {
  if (S <= 0) return IMPNN_OK;
  enc::TImageParams ip{};
  ip.weights = weights;
  ip.bond_table = bond_table;
  ip.prepared = static_cast<float*>(prepared);
  ip.K = K;
  ip.S = S;
  ip.Vb = Vb;
  ip.x3 = x3 ? 1 : 0;
  ip.step_floats = impnn_encoder_step_floats(enc::kD, K);
  return enc::launch_typed_image(ip, s);
}

int launch_encoder_typed_run(const EncoderArgs& a, const enc::Ws& w, hipStream_t s) {
  using namespace enc;
  char* base = static_cast<char*>(a.workspace);
  TEncParams ep{};
  const size_t S1 = a.S > 0 ? a.S : 1;
  const bool x3 = a.mode == 3;
  const size_t uslot = x3 ? kXUpdSlot : kTUpdSlot;
  for (int g = 0; g < a.n_ions; ++g) {
    const float* prep;
    if (a.prepared[g]) {
      if (!aligned16(a.prepared[g])) return fail(IMPNN_E_BADARG, "encoder_fused: prepared weights must be 16B aligned");
      prep = static_cast<const float*>(a.prepared[g]);
    } else {  // canonical weights: build the images into the workspace first
      float* img = reinterpret_cast<float*>(base + w.img_off) + (size_t)g * typed_prepared_floats(a.S, a.Vb, x3);
      if (int rc = launch_encoder_typed_prepare(a.weights[g], a.bond_table, a.K, a.S, a.Vb, x3, img, s)) return rc;
      prep = img;
    }
    ep.upd[g] = prep;
    ep.tmat[g] = prep + S1 * uslot;
    ep.pooled[g] = a.pooled[g];
  }
  ep.atom_table = a.atom_table;
  ep.nsub = reinterpret_cast<const int32_t*>(base + w.nsub_off);
  ep.desc = reinterpret_cast<const int32_t*>(base + w.desc_off);
  ep.rec = reinterpret_cast<const unsigned char*>(base + w.rec_off);
  ep.header = reinterpret_cast<const PlanHeader*>(base);
  ep.n_ions = a.n_ions; ep.B = a.B; ep.S = a.S; ep.Va = a.Va; ep.Vb = a.Vb; ep.max_sub = w.max_sub;
  ep.ln_eps = a.ln_eps;
  ep.stamps = nullptr;
  {
    size_t sb = 0;
    void* sp = debug_stamp_buffer(&sb);
    if (sp && sb >= (size_t)w.nwg * 32 * sizeof(unsigned long long)) ep.stamps = static_cast<unsigned long long*>(sp);
  }
  ep.upd_slot = (int)uslot;
  ep.ecap = tecap_of(a.E);
  ep.rec_lds = trec_lds_bytes(a.Vb, ep.ecap);
  const bool big3 = x3 && ep.ecap == kTECapBig;  // mode 3 on 640-edge chunks: unpadded h rows (kTHSBig3)
  void (*kern)(TEncParams) =
      big3 ? (ep.stamps ? encoder_typed_kernel<true, true, kTHSBig3> : encoder_typed_kernel<false, true, kTHSBig3>)
      : x3 ? (ep.stamps ? encoder_typed_kernel<true, true> : encoder_typed_kernel<false, true>)
           : (ep.stamps ? encoder_typed_kernel<true, false> : encoder_typed_kernel<false, false>);
  if (int rc = ensure_lds_limit((const void*)kern, big3 ? (ep.stamps ? 10 : 9) : (ep.stamps ? 5 : 4) + (x3 ? 2 : 0))) return rc;
  size_t lds = lds_fixed_bytes(x3, a.Vb, ep.ecap, big3 ? kTHSBig3 : kTHS);
  const size_t atab_bytes = ((size_t)a.Va + 1) * kTAS * sizeof(float);
  ep.atab_lds = lds + atab_bytes <= 160 * 1024;
  if (ep.atab_lds) lds += atab_bytes;
  profile_record_start(s);
  kern<<<w.nwg, kThreads, lds, s>>>(ep);
  profile_record_stop(s);
  return check_launch("encoder_typed");
}

}  // namespace impnn
