// Shared constants, LDS/HBM layouts and kernel parameter blocks of the fused encoder (gfx950).
// Included by encoder_plan.hip (plan + weight-image kernels) and encoder_fused.hip (the encoder).
#pragma once

#include "common.h"

namespace impnn {
namespace enc {

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

// ---- chunk record: everything the encoder needs to know about one chunk's graph structure, built by
// plan_chunks (one small workgroup per chunk, many per CU) and copied verbatim into LDS by the
// encoder.  Offsets in bytes.
constexpr int kRecRowptr = 0;      // u16[kRCap + 2]: CSR over PLACED rows (rows placed by descending in-degree)
constexpr int kRecTilemax = 528;   // u8[16]        : largest in-degree inside each 16-row tile
constexpr int kRecMoloff = 544;    // u16[kRCap + 2]: first logical row of every molecule (+ end marker)
constexpr int kRecMolrows = 1072;  // u16[kRCap]    : kept rows r_b of every molecule
constexpr int kRecPoolrow = 1584;  // u16[kRCap]    : logical row -> placed row, | 0x8000 if atom id > 0 (pooled)
constexpr int kRecRowatom = 2096;  // i32[kRCap]    : atom id of the PLACED row, -1 for a slack row
constexpr int kEntMaxAtom = 454;   // 12-bit field of atom id * kHS/4 in an in-edge entry (atom table in LDS)
constexpr int kRecEnt = 3136;      // u32[kECap]    : in-edge lists in edge-slot order: slot<<16 | bond id<<8 | placed src row
constexpr int kRecBytes = 8192;
static_assert(kRecEnt + 4 * kECap <= kRecBytes && kRecEnt % 16 == 0, "record layout");

// ---- "typed" encoder (mode 2, encoder_typed.hip): the message is m_e = A[bond type of e] * h[src_e] with the
// per-bond-type matrices A[v] = sum_k bond_table[v,k] W[k] (models/layers.py:108 evaluated once per type), so
// bond_dim drops out of the kernel (K = D^2 of train_melting_point.py:146 included).  Edges of a chunk are grouped
// by type in groups of <= 4 (the 4 rows of a v_mfma_f32_4x4x1 block), messages land in an LDS buffer in
// "jagged diagonal" order - slot(row, d) = jdptr[d] + row for the d-th in-edge (edge-slot order) of placed row
// `row`; rows are placed by descending in-degree, so the rows that have a d-th in-edge are a prefix - and every
// atom row then sums its in-edges in edge-slot order (the reference's sequential scatter_nd, models/layers.py:78-82).
constexpr int kTECap = 512;          // valid edges per chunk (2 per virtual row) for padded shapes E <= 512 ...
constexpr int kTECapBig = 640;       // ... and beyond: 2.5 per virtual row, so that a 160-atom explicit-hydrogen molecule
                                     // (<= 160 bonds x 4 edge slots: train_viscosity.py:288-289) still fits one chunk; the
                                     // message buffer then takes the LDS of the atom table copy (mode 2 only)
__host__ __device__ constexpr int tecap_of(int E) { return E > kTECap ? kTECapBig : kTECap; }
constexpr int kTVbMax = 256;         // bond ids travel as 8 bits
constexpr int kTHS = 36;             // LDS row stride (floats) of h in the typed encoder (as kHS: rows spread over the banks)
constexpr int kTAS = 32;             // row stride of the atom table copy (unpadded: the LDS budget of mode 3 needs the 2 KB)
constexpr int kTRecRowdeg = 0;       // u16[kRCap]    : in-degree of the PLACED row
constexpr int kTRecTilemax = 528;    // u8[16]
constexpr int kTRecMoloff = 544;     // u16[kRCap + 2]
constexpr int kTRecMolrows = 1072;   // u16[kRCap]
constexpr int kTRecPoolrow = 1584;   // u16[kRCap]
constexpr int kTRecRowatom = 2096;   // i32[kRCap]
constexpr int kTRecCounts = 3120;    // u16 groups, u16 edges, u16 max in-degree
constexpr int kTRecJdptr = 3136;     // u16[258]      : first message slot of in-edge index d
constexpr int kTRecNrun = 3664;      // u16           : type runs (bond types present in the chunk)
constexpr int kTRecGrp = 3712;       // uint4[tgrp_cap(Vb)]: x = type | edges << 8 | groups of the type from here on << 24,
                                     //   y = 4 x u8 placed source row, z/w = 4 x u16 message key (tmsg_key of the edge's
                                     //   slot; the dump slot for unused lanes); in type order
// A chunk holds <= ecap edges of <= Vb types: at most ecap/4 full groups plus one partial group per type.  The run
// table (u16: first group of every run, + end) follows the group table, so a record's used bytes - and the LDS the
// encoder spends on it - depend on the bond vocabulary: 7.2 KB at Vb = 72, 12 KB at Vb = 256.
// A run is what one wave of the encoder multiplies with one fetch of the type's matrix: the groups of a type, cut into
// pieces of <= gmax = max(2, ceil(groups of the chunk / kTRunTarget)) groups, so that a chunk of FEW bond types - real
// molecules have a handful - still gives every wave its share (uncut, a chunk of 6 types kept 6 of 16 waves busy:
// 2.46 M pairs/s at the explicit-hydrogen shape against 3.79 M with 71 uniformly drawn types).  Runs <= types + kTRunTarget.
constexpr int kTRunTarget = 32;
__host__ __device__ constexpr int tgrp_cap(int Vb, int ecap) { return ecap / 4 + (Vb < kTVbMax ? Vb : kTVbMax); }
__host__ __device__ constexpr int trec_runs_off(int Vb, int ecap) { return kTRecGrp + 16 * tgrp_cap(Vb, ecap); }
__host__ __device__ constexpr int trec_used_bytes(int Vb, int ecap) {
  return (trec_runs_off(Vb, ecap) + 2 * (Vb + 2 + kTRunTarget) + 15) & ~15;
}
constexpr int kTRecBytes = 12288;    // stride of the records in the workspace
constexpr int kTRecPart1 = 8192;     // the encoder copies a record as 8 B per thread (+ 4 B per thread beyond 8 KB)
// LDS bytes the encoder reserves for the record: its used part
__host__ __device__ constexpr int trec_lds_bytes(int Vb, int ecap) { return trec_used_bytes(Vb, ecap); }
static_assert(kTRecGrp % 16 == 0 && trec_used_bytes(kTVbMax, kTECapBig) <= kTRecBytes, "typed record layout");
// message buffer: ecap slots of 128 B (16-byte units XOR-swizzled by slot) + a dump slot (index ecap) for the unused edge
// lanes of a group + a slot of zeros (index ecap + 1: what a row reads for the in-edges it does not have, so the Reduce
// is branch-free)
__host__ __device__ constexpr int tmsg_floats(int ecap) { return (ecap + 2) * kD; }
// 16-byte unit u (0..7) of message slot s -> float offset.  16 consecutive slots x one unit cover all 16 bank quads
// (the pull of a tile is conflict-free), and the 8 units of a slot stay a permutation of its 32 banks.
__host__ __device__ constexpr int tmsg_off(int s, int u) { return s * kD + ((u ^ ((s >> 1) & 7)) << 2); }
// the same as one XOR per lane: float offset of feature f (0..31) of slot s = tmsg_key(s) ^ f
__host__ __device__ constexpr int tmsg_key(int s) { return s * kD + (((s >> 1) & 7) << 2); }
static_assert(tmsg_key(kTECapBig + 1) < 65536, "message keys travel as 16 bits");
// per-step update image (mode 2): the gate kernels in the A-operand order of v_mfma_f32_16x16x4_f32 - 24 blocks
// (gate, T, half, u) of 64 lanes x 4 floats, lane (a = l & 15, q = l >> 4), element r:
//     W_gate[(32 half + 16 u + 4 q + r) * 32 + 16 T + a]       (keras kernel (64, 32): [input][output])
// so that every A fetch of the update GEMMs is one lane-linear (conflict-free) ds_read_b128 - then the 5 vectors
// (bz, br, bh, gamma, beta).  Stored in a slot of 2 loads per thread.
constexpr int kTUpdBlocks = 24;
constexpr int kTVecFloatOff = kTUpdBlocks * 256;        // 6144
constexpr int kTUpdFloats = kTVecFloatOff + 5 * kD;     // 6304
constexpr int kTUpdSlot = 2 * kThreads * 4;             // 8192 floats
constexpr int kTUpdLds = 6400;                          // floats kept in LDS (>= kTUpdFloats, multiple of 128)
static_assert(kTUpdFloats <= kTUpdLds && kTUpdLds <= kTUpdSlot, "typed update image");
// mode 3 ("f32x3"): the update GEMMs on the bf16 matrix pipe with every f32 operand carried EXACTLY as three bf16 terms
// (x = b0 + b1 + b2: 3 x 8 significant bits, fp32's exponent range) and all nine cross products accumulated in f32.
// Image: 12 blocks (gate, T, half) x 3 planes x 512 bf16 in MFMA A-operand order (feat_of, as the f16x2 image), then
// the 5 f32 vectors.
constexpr int kXUpdHalfs = 3 * 2 * 2 * 3 * 512;               // 18432 bf16 = 36 KB
constexpr int kXVecFloatOff = kXUpdHalfs / 2;                  // 9216
constexpr int kXUpdFloats = kXVecFloatOff + 5 * kD;            // 9376
constexpr int kXUpdSlot = 3 * kThreads * 4;                    // 12288 floats: three 16-byte loads per thread
constexpr int kXUpdLds = 9472;                                 // floats kept in LDS
static_assert(kXUpdFloats <= kXUpdLds && kXUpdLds <= kXUpdSlot, "x3 update image");
// type matrices of one (ion, step): Vb x 1024 floats, each in 4x4x1-MFMA B-operand order:
//   A[v][r][k] at v*1024 + (k >> 2)*128 + r*4 + (k & 3)   (lane l loads the 4 k-quads 4*(l >> 5) + i of row l & 31:
//   a wave's load i is two contiguous 512 B runs)
constexpr int kTMatFloats = kD * kD;
// prepared buffer of one ion: S update slots | S x Vb type matrices | one canonical (Vb,32,32) scratch
inline size_t typed_prepared_floats(int S, int Vb, bool x3 = false) {
  const size_t s = S > 0 ? S : 1;
  return s * (x3 ? kXUpdSlot : kTUpdSlot) + s * (size_t)Vb * kTMatFloats + (size_t)Vb * kTMatFloats;
}

constexpr int kShareCap = kECap;  // molecules of one share that plan_chunks resolves in LDS
// Shares are equal in virtual rows, per ion and in proportion to the ion's rows - so where molecules are tiny (halide
// anions: one atom, no bond) a share would hold as many molecules as rows.  Every molecule therefore counts at least
// plan_vmin virtual rows: a share spans < 2 (all rows) / nwg + vrmax virtual rows (the ion split rounds to whole
// workgroups), hence at most that many / vmin molecules.  Placement only: no result depends on it.
inline int plan_vmin(int n_ions, int B, int vrmax, int nwg) {
  const int64_t span = 2 * (int64_t)n_ions * B * vrmax / (nwg > 0 ? nwg : 1) + vrmax;
  const int64_t v = (span + kShareCap - 2) / (kShareCap - 1);
  return (int)(v < 1 ? 1 : (v > kRCap ? kRCap : v));
}
constexpr int kMaxHops = 128;     // chunks of one share (Ws::max_sub <= kMaxHops: encoder_workgroups sees to it)

// chunk descriptor (int4): {first molecule, molecules, 0, rows | ion << 16}
constexpr int kPB = 16;  // molecules per plan_stats workgroup (= partial-sum granularity)

// ---- workspace layout (bytes, all 256-aligned sections)
struct Ws {
  size_t img_off, rows_off, vr_off, partial_off, share_off, nsub_off, desc_off, rec_off, total;
  int nwg;      // persistent encoder workgroups (= compute units)
  int max_sub;  // chunk slots per workgroup (upper bound of chunks in one share)
  int nblk;     // 16-molecule blocks per ion
  int rec_bytes;  // kRecBytes (pull modes) or kTRecBytes (typed)
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// virtual rows that `edges` valid edges occupy in a typed chunk: edges <= ecap  <=>  rows <= kRCap
__host__ __device__ constexpr int tvr_of_edges(int edges, int ecap) { return (edges * kRCap + ecap - 1) / ecap; }

inline int vr_max_of(int N, int E, bool typed = false) {
  int v = typed ? (E > 4 * kTECapBig ? kRCap + 1 : tvr_of_edges(E, tecap_of(E))) : (E + 3) / 4;
  int m = N > v ? N : v;
  m = m < 1 ? 1 : m;
  // typed plans take any padded shape: a molecule is bounded by what it HOLDS (kept rows, valid edges: plan_stats), not
  // by N and E, and one that exceeds a chunk raises PlanHeader::overflow
  return typed && m > kRCap ? kRCap : m;
}

// The first 256 bytes of a workspace are the plan header (PlanHeader): written by the plan, checked by the encoder.
struct PlanHeader {
  int32_t magic, kind, n_ions, B, N, E, nwg, max_sub;
  int32_t overflow;  // typed plans: 1 when a molecule does not fit a chunk (more than kRCap rows / 2 kRCap valid edges, or an
                     // in-degree above 255): the encoder then writes NaN (callers with such shapes read this word back)
};
constexpr int32_t kPlanMagic = 0x696d706e;  // "impn"
constexpr int32_t kPlanBadBit = 1 << 30;     // in a block's partial sum of virtual rows: the block holds a molecule that
                                             // does not fit a chunk

inline Ws ws_layout(int n_ions, int B, int N, int E, int S, int Vb, int nwg, bool typed, bool x3 = false) {
  Ws w{};
  const int vrmax = vr_max_of(N, E, typed);
  w.rec_bytes = typed ? kTRecBytes : kRecBytes;
  const int win = kRCap - vrmax + 1;  // a chunk closed by next-fit holds at least this many rows
  w.nwg = nwg;
  w.nblk = (B + kPB - 1) / kPB;
  // rows of one share <= 2 * (all rows) / nwg + vrmax (ion split rounds to whole workgroups)
  const int64_t share_rows = (2 * (int64_t)n_ions * B * vrmax) / nwg + vrmax;
  // chunks of a share: a chunk closed by next-fit holds >= win rows; and any two consecutive chunks hold more than
  // kRCap rows together (else next-fit had merged them) - the bound that holds whatever the padded shape is
  const int64_t by_win = win >= 1 ? share_rows / win + 2 : (int64_t)1 << 40;
  const int64_t by_pairs = 2 * share_rows / (kRCap + 1) + 2;
  w.max_sub = (int)(by_win < by_pairs ? by_win : by_pairs);
  size_t off = 256;  // plan header
  w.img_off = off;
  off = align_up(off + (typed ? (size_t)n_ions * typed_prepared_floats(S, Vb, x3)
                              : (size_t)n_ions * (S > 0 ? S : 1) * kImgSlot) * sizeof(float), 256);
  w.rows_off = off;
  off = align_up(off + (size_t)n_ions * B * sizeof(int32_t), 256);
  w.vr_off = off;
  off = align_up(off + (size_t)n_ions * B * sizeof(int32_t), 256);
  w.partial_off = off;
  off = align_up(off + (size_t)n_ions * w.nblk * sizeof(int32_t), 256);
  w.share_off = off;
  off = align_up(off + (size_t)nwg * 4 * sizeof(int32_t), 256);
  w.nsub_off = off;
  off = align_up(off + (size_t)nwg * sizeof(int32_t), 256);
  w.desc_off = off;
  off = align_up(off + (size_t)nwg * w.max_sub * 4 * sizeof(int32_t), 256);
  w.rec_off = off;
  off = align_up(off + (size_t)nwg * w.max_sub * w.rec_bytes, 256);
  w.total = off;
  return w;
}

struct PlanParams {
  const int32_t* atom_ids[2];
  const int32_t* bond_ids[2];
  const int32_t* conn[2];
  int32_t* rows;      // [n_ions][B]     kept rows r_b
  int32_t* vr;        // [n_ions][B]     virtual rows max(1, r_b, ceil(v_b/4))
  int32_t* partial;   // [n_ions][nblk]  sum of vr over 16 molecules
  int32_t* nsub;      // [nwg]           chunks of every encoder workgroup
  int32_t* desc;      // [nwg][max_sub][4]
  unsigned char* rec; // [nwg][max_sub][kRecBytes]
  int n_ions, B, N, E, Va, Vb, nwg, max_sub, nblk;
  int grid_sub;  // plan_chunks workgroups launched per share (<= max_sub)
  int typed;     // 1: typed records (kTRecBytes), ecap / kRCap valid edges per virtual row
  int ecap;      // typed: valid edges per chunk, tecap_of(E)
  int vmin;      // every molecule counts at least this many virtual rows (plan_vmin): a share then holds <= kShareCap molecules
  PlanHeader* header;          // written by plan_stats block 0
  unsigned long long* stamps;  // diagnostics only: 16 words written by plan_chunks workgroup 0
};

struct ImageParams {
  const float* weights;  // S steps, canonical layout
  float* img;            // S slots of kImgSlot floats
  int K, mode;
  int64_t step_floats;
};

struct EncParams {
  const int32_t* atom_ids[2];
  float* pooled[2];
  const float* atom_table;
  const float* bond_table;
  const float* img[2];  // per ion: S weight images (kImgSlot floats each)
  const int32_t* nsub;
  const int32_t* desc;
  const unsigned char* rec;
  const PlanHeader* header;
  int n_ions, B, N, K, S, Va, Vb, max_sub;
  int atab_lds;  // atom table copied to LDS (Va*32 floats after the bond table copy); 0: read from HBM/L2
  float ln_eps;
  unsigned long long* stamps;  // diagnostics only (impnn_debug_set_stamp_buffer): 32 words per workgroup
};

struct TImageParams {
  const float* weights;  // S steps, canonical layout
  const float* bond_table;
  float* prepared;       // typed_prepared_floats(S, Vb, x3)
  int K, S, Vb, x3;
  int64_t step_floats;
};

struct TEncParams {
  float* pooled[2];
  const float* atom_table;
  const float* upd[2];   // per ion: S update slots (kTUpdSlot floats each)
  const float* tmat[2];  // per ion: S x Vb type matrices in operand order
  const int32_t* nsub;
  const int32_t* desc;
  const unsigned char* rec;
  const PlanHeader* header;
  int n_ions, B, S, Va, Vb, max_sub;
  int atab_lds;
  int upd_slot;  // floats between the update images of consecutive steps (kTUpdSlot or kXUpdSlot)
  int rec_lds;   // LDS bytes reserved for the chunk record: trec_lds_bytes(Vb, ecap)
  int ecap;      // valid edges per chunk: tecap_of(E)
  float ln_eps;
  unsigned long long* stamps;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global store of the wave (s_waitcnt vmcnt(0)); in kernels that stream results to HBM between LDS
// phases that wait is pure latency (1-2 us per barrier).  Use only where no thread reads another
// thread's GLOBAL writes after the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Inclusive wave64 prefix sum on the DPP network (no LDS round trips: a __shfl_up chain is six
// dependent ds_bpermute, ~150 cycles each).
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31 -> rows 2,3
  return v;
}

__device__ __forceinline__ bool edge_valid(int s, int t, int bid, int N, int Vb) {
  // models/layers.py:114-115 (src>0 & tgt>0); out-of-range indices behave as padding (impnn.h)
  return (unsigned)s - 1u < (unsigned)(N - 1) && (unsigned)t - 1u < (unsigned)(N - 1) && (unsigned)bid < (unsigned)Vb;
}

// plan side (encoder_plan.hip)
int launch_weight_image(const ImageParams& ip, int S, hipStream_t s);
int launch_plan(const PlanParams& pp, hipStream_t s);
int launch_typed_image(const TImageParams& ip, hipStream_t s);

}  // namespace enc
}  // namespace impnn
