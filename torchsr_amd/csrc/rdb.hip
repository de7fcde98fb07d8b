// ESRGAN ResidualDenseBlock forward as ONE kernel (torchsr/esrgan/residual.py:65-86), bf16 products / fp32 accumulate.
//
//   c_k = LeakyReLU(conv_k(cat(x, c_1..c_{k-1})) + b_k), k = 1..4;   y = (conv_5(cat(x, c_1..c_4)) + b_5) * scale + x
//
// As five launches the block is launch- and fill-bound: at batch 16 x 32x32 pixels every conv is 0.6-1.8 GFLOP, one
// round of <= 256 workgroups that each stream their operands for a few microseconds (11.9-17 us per launch for < 1 us of
// bf16 matrix work, profiles/r02_other_configs.txt), and every 32-channel slice makes an HBM round trip between convs.
// Here one workgroup owns an 8x8 pixel tile of one image -- 16 images x 16 tiles = exactly the chip's 256 CUs -- and runs
// the whole chain on it:
//   * the input patch with a 5-pixel halo (18x18 x 64 channels) is staged ONCE, rounded to bf16 (what the product
//     multiplies anyway), and the four intermediates live in LDS as bf16 on shrinking regions (16x16, 14x14, 12x12,
//     10x10): 84 KB in all.  Halo pixels are recomputed by the neighbouring tiles (1.77x the FLOPs, at a few per cent
//     of the bf16 MFMA peak that is free); pixels outside the image are stored as the zeros the next conv's padding reads.
//   * weights come as bf16 tiles pre-swizzled for the LDS image (srx_rdb_pack, once per optimiser step for all blocks):
//     one unit = one 32-channel source x 9 taps x all output channels (18 / 36 KB).  Waves 4..7 stream them by LDS-DMA into
//     a ring of four 18 KB slots, one GROUP of units (two 18 KB units, or one 36 KB unit of conv 5) ahead; ONE barrier per group,
//     14 groups per block (round 6; before: one barrier per unit, 20 per block).
//   * waves 0..3 (one per SIMD) multiply: a wave owns up to two 32-pixel x 32-channel output tiles of the current conv and
//     reads every weight fragment once for both; fragments are requested two taps ahead, reads and address arithmetic
//     interleaved with the MFMAs (sched_group_barrier).  Operands are (weights as A, pixels as B) so that a lane ends
//     up with four consecutive channels of one pixel per accumulator quad: packed 8-byte LDS stores for the next conv,
//     16-byte global stores for the fp32 copies the backward pass needs (tile centre only).
// Measured (MI355X, batch 16 x 32x32, 69 dependent launches in a hipGraph): 28.6 us per block against 61 us for the five
// launches it replaces; 1944 MFMAs per workgroup = 7.4 us of matrix time per SIMD, the rest is the patch staging (2.9 us),
// the four epilogues (4 us), 20 barriers and LDS reads that are 36 % bank-conflict cycles (region rows of 16..8 pixels
// in images of 18..10 pixels pitch: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r03_pmc_rdb.txt).
// 64-byte LDS rows (32 bf16 channels).  Weight rows: the 16-byte quad j of row r sits at quad j ^ ((r >> 2) & 3) -- sixteen
// consecutive rows cover all 64 banks, and the lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
// + 32: MI355X_MICROARCH.md, LDS) each see sixteen rows that are consecutive mod 16.  Activation images (round 6): quad j of pixel
// (y, x) sits at j ^ (y & 3), and a 32-pixel MFMA tile is 8 x 4 pixels with each lane group on one 4 x 4 block (tile_pixel): the
// 16-byte slot of a read is 4 ((y pitch + x) mod 4) + (j ^ (y & 3)), every image pitch is even, so ANY 4 x 4 block of ANY image --
// whatever the tap, the region's offset in the source and the pitch -- touches sixteen different slots: conflict-free fragment
// reads for all (region, source) pairs of the five convs without padding an image (LDS has 3 KB to spare) or splitting a weight
// unit.  Before: linear 32-pixel runs of a 16..10-pixel-wide region in an 18..10-pixel pitch, 42 % of the LDS cycles conflicts
// (profiles/r05_rdb_ablation.txt).
#include "srx_common.h"
#include <mutex>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int RT = 8;    // output tile edge
constexpr int NSRC = 6;  // 32-channel sources: x[0:32], x[32:64], c1, c2, c3, c4
// source s: edge of its LDS image, origin relative to the tile (region of c_j = tile grown by 5 - j pixels)
__host__ __device__ constexpr int src_w(int s) { return s < 2 ? RT + 10 : RT + 10 - 2 * (s - 1); }
__host__ __device__ constexpr int src_org(int s) { return s < 2 ? -5 : -(5 - (s - 1)); }
__host__ __device__ constexpr int src_off(int s) {  // byte offset of the image in LDS
  int o = 0;
  for (int i = 0; i < s; ++i) o += src_w(i) * src_w(i) * 64;
  return o;
}
constexpr int ACT_BYTES = src_off(NSRC);  // 86016
// conv k (1..5): output region = image of source k + 1 (k < 5) or the tile itself
__host__ __device__ constexpr int reg_w(int k) { return k < 5 ? src_w(k + 1) : RT; }
__host__ __device__ constexpr int reg_org(int k) { return k < 5 ? src_org(k + 1) : 0; }
__host__ __device__ constexpr int conv_n(int k) { return k == 5 ? 64 : 32; }
__host__ __device__ constexpr int conv_cin(int k) { return 64 + 32 * (k - 1); }
// weight units in consumption order: conv 1 (2 sources), conv 2 (3), ... conv 5 (6)
constexpr int NUNITS = 20;
__host__ __device__ constexpr int unit_first(int k) { return (k - 1) * (k + 2) / 2; }  // 0, 2, 5, 9, 14
__host__ __device__ constexpr int unit_conv(int u) { return u < 2 ? 1 : (u < 5 ? 2 : (u < 9 ? 3 : (u < 14 ? 4 : 5))); }
__host__ __device__ constexpr int unit_bytes(int u) { return 9 * conv_n(unit_conv(u)) * 64; }
__host__ __device__ constexpr int unit_off(int u) {
  int o = 0;
  for (int i = 0; i < u; ++i) o += unit_bytes(i);
  return o;
}
constexpr int PACKED_BYTES = unit_off(NUNITS);  // 479232 per block
constexpr int SLOT_BYTES = 9 * 32 * 64;         // 18432: a unit of conv 1..4; conv 5's units (64 outputs) take two slots
constexpr int NSLOT = 4;
constexpr int OFF_W = ACT_BYTES;
constexpr int OFF_BIAS = OFF_W + NSLOT * SLOT_BYTES;
// The weight ring: four 18 KB slots = two halves.  Round 6: the units are multiplied in GROUPS -- two 18 KB units of one conv, or a
// single unit where a conv has an odd count (its last, which carries the epilogue), or one 36 KB unit of conv 5 -- and a group
// fills one half of the ring; the halves alternate.  ONE barrier per group (14 instead of 20): at the barrier that opens group g
// the half of group g - 1 is free and group g + 1 is requested into it, to be multiplied one group later; the two units of a group
// run as ONE 18-tap pipeline.  In-kernel stamps had put ~600-780 cycles of every unit outside its MFMAs: the fragment reads of
// its first two taps (nothing of a unit can be read before its barrier), the drain of its last MFMAs and the rendezvous itself.
// (Round 4's ring had every unit in a slot of its own, requested three units ahead with counted vmcnt waits.)
constexpr int NGROUPS = 14;
__host__ __device__ constexpr int grp_first(int g) {
  constexpr int first[NGROUPS + 1] = {0, 2, 4, 5, 7, 9, 11, 13, 14, 15, 16, 17, 18, 19, NUNITS};
  return first[g];
}
__host__ __device__ constexpr int grp_count(int g) { return grp_first(g + 1) - grp_first(g); }
__host__ __device__ constexpr int unit_group(int u) {
  int g = 0;
  while (grp_first(g + 1) <= u) ++g;
  return g;
}
__host__ __device__ constexpr int unit_slot(int u) { return 2 * (unit_group(u) & 1) + (u - grp_first(unit_group(u))); }
constexpr int LDS_BYTES = OFF_BIAS + 192 * 4;   // 160512
constexpr int NTHREADS = 512;

// Forward (BWD = false): src = buf (x in channels 0..63, row stride ld), the stage outputs c1..c4 go to buf channels 64..191,
// the last stage writes (conv5 + b5) * scale + x to out.
// Backward (BWD = true), the data-gradient chain of the same block -- structurally the same computation run in reverse:
//   g5 = scale * dy (64 channels: the "input patch"),  g_j = LeakyReLU'(c_j) * sum_{k > j} conv_k^T(g_k)[c_j]  (j = 4..1, 32 channels,
//   regions 16x16 .. 10x10),  dx = sum_k conv_k^T(g_k)[x] + skip_scale * skip  (64 channels, the tile).
// Stage K (1..5) has the K + 1 sources g5[0:32], g5[32:64], g4, .., g_{6-K} and produces the slice of c_{5-K} (K < 5) or of
// x (K = 5): the forward's schedule with transposed, tap-flipped weights (srx_rdb_pack, bwd stream).  src = dy (row stride
// ld), buf = the block's saved activations (the masks read c_j > 0), the stage outputs g4..g1 go to gout channels
// 64+32(j-1).. (fp32: the weight gradients read them), the last stage writes dx to out.
struct RdbArgs {
  const float* src;      // forward: the block buffer (x); backward: dy
  float* buf;            // [N][H][W][bld]: forward: c1..c4 are written to channels 64..191; backward: read for the masks
  float* gout;           // backward: [N][H][W][gld], g1..g4 at channels 64..191
  const float* skip;     // backward: [N][H][W][skip_ld], added to dx with skip_scale
  const float* extra;    // optional second addend of the last stage ([N][H][W][extra_ld], channels 0..63): forward
                         //   out = ((conv5 + b5) * scale + x) * post_scale + extra  (the `out * 0.2 + x` that ends an RRDB,
                         //   esrgan/residual.py:128); backward  dx = ... + skip_scale * skip + extra  (the RRDB's own skip gradient)
  const unsigned char* wpk;
  const float* bias[5];  // forward only
  float* out;            // [N][H][W][out_ld], channels 0..63
  int N, H, W, ld, bld, gld, skip_ld, extra_ld, out_ld, tiles_x, tiles_y;
  float scale, slope, skip_scale, post_scale;
  unsigned* dbg;  // developer aid (srx_rdb_fwd_dbg): [workgroup][wave][48] shader-clock stamps, staged in LDS behind LDS_BYTES; null in the product
};
constexpr int DBG_SLOTS = 48;
// stamp `idx` of this wave: 0 kernel start, 1 patch staged, 2 + 2U unit U's barrier passed, 3 + 2U unit U's work done
__device__ __forceinline__ void rdb_stamp(const RdbArgs& a, unsigned char* lds, int wave, int lane, int idx) {
  if (a.dbg && lane == 0) reinterpret_cast<unsigned*>(lds + LDS_BYTES)[wave * DBG_SLOTS + idx] = (unsigned)__builtin_amdgcn_s_memtime();
}

// Roles.  Waves 0..3 (one per SIMD) multiply: a wave owns up to two 32-pixel tiles of the current conv's region and reads
// every weight fragment once for both (LDS traffic is what bounds this kernel: 3 fragment reads per 2 MFMAs instead of 4).
// Waves 4..7 stream the weights: unit u + 1 goes global -> registers -> the other LDS slot while unit u is multiplied.
constexpr int NCOMPUTE = 4, NLOAD_THREADS = NTHREADS - NCOMPUTE * 64;

struct Wave {
  f32x16 acc[2];
  int lane, h, l31, wave;
  int n_img, ty0, tx0;
};

// Round 4: the weight stream by LDS-DMA (buffer_load_dwordx4 ... lds).  The packed units are already the LDS image, so the
// copy is lane-linear: one instruction of one wave moves 1 KiB, nothing passes through registers and no ds_write is issued --
// the nine 16-byte LDS stores per loader thread and unit competed with the compute waves' fragment reads for the LDS and with
// their MFMAs for the SIMD's issue slots.  A unit is a whole number of KiB (18 or 36): KiB j goes to loader wave j % 4, so a
// wave issues dma_count(U, wave) instructions per unit -- the counted vmcnt waits below depend on it -- and no lane ever
// writes past the unit's end (the next slot may hold a live unit).
typedef unsigned rdb_u32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int dma_count(int u, int wave_l) { return (unit_bytes(u) / 1024 - wave_l + 3) / 4; }
template <int U>
__device__ __forceinline__ void dma_unit(const RdbArgs& a, unsigned lds_base, int lt) {
  constexpr int kib = unit_bytes(U) / 1024, rounds = (kib + 3) / 4;
  const unsigned long long p = (unsigned long long)(a.wpk + unit_off(U));
  rdb_u32x4 rs;
  rs[0] = (unsigned)srx_uniform((int)(unsigned)p);
  rs[1] = (unsigned)srx_uniform((int)((unsigned)(p >> 32) & 0xffffu));
  rs[2] = (unsigned)unit_bytes(U);
  rs[3] = 0x00020000u;
  const unsigned wave_l = (unsigned)srx_uniform(lt >> 6);
#pragma unroll
  for (int i = 0; i < rounds; ++i) {
    if ((i + 1) * 4 > kib && i * 4 + (int)wave_l >= kib) continue;  // (wave-uniform; last round of an 18 KiB unit: waves 0, 1 only)
    const unsigned dst = (unsigned)srx_uniform((int)(lds_base + (unsigned)(OFF_W + unit_slot(U) * SLOT_BYTES) + (unsigned)(i * 4) * 1024u + wave_l * 1024u));
    const unsigned voff = (unsigned)(i * NLOAD_THREADS + lt) * 16u;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs) : "memory");
  }
}
// the loader waves' wait: every request has landed except the newest `units_18k` 18 KiB units / `units_36k` 36 KiB units
// (requests retire in issue order; the loader waves issue no other vector memory instruction)
template <int UNITS_18K, int UNITS_36K>
__device__ __forceinline__ void dma_wait(int wave_l) {
  constexpr int HI = UNITS_18K * dma_count(0, 0) + UNITS_36K * dma_count(14, 0), LO = UNITS_18K * dma_count(0, 3) + UNITS_36K * dma_count(14, 3);
  if (wave_l < 2) __builtin_amdgcn_s_waitcnt(0x0f70 | (HI & 15) | ((HI >> 4) << 14));  // vmcnt(HI), expcnt / lgkmcnt untouched
  else __builtin_amdgcn_s_waitcnt(0x0f70 | (LO & 15) | ((LO >> 4) << 14));
}

// workgroup barrier that orders LDS traffic only (see c64.hip): the compute waves' global stores of c1..c4 / g4..g1 fly on
__device__ __forceinline__ void rdb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// conv K: pixel tiles of 32, channel tiles of 32; a compute wave takes TPW consecutive pixel tiles of one channel tile
// pixel tiles of conv K's region: 8 x 4 pixels each; the 10 x 10 region of conv 4 would need six of them on four waves, so it
// takes two 8 x 4 tiles and packs the L-shaped rest into two more (tile_pixel)
template <int K> __host__ __device__ constexpr int m_tiles() { return K == 4 ? 4 : ((reg_w(K) + 7) / 8) * ((reg_w(K) + 3) / 4); }
// lane (l31 = lane & 31) of pixel tile mt -> its pixel (qx, qy) of the region; false: the lane has no pixel (coordinates are then
// clamped into the region: it reads what a neighbour reads and stores nothing).  Row = l31 >> 3; the four-lane runs of a row are
// dealt to the left / right 4 x 4 block so that the hardware's ds_read_b128 lane groups each own one block (mask 0x96: lanes
// 0-3, 12-15, 20-27 left, 4-11, 16-19, 28-31 right).
template <int K>
__device__ __forceinline__ bool tile_pixel(int mt, int l31, int& qx, int& qy) {
  constexpr int WK = reg_w(K);
  const int r = l31 >> 3, right = (0x96 >> (l31 >> 2)) & 1, c = (l31 & 3) + 4 * right;
  if constexpr (K == 4) {
    if (mt < 2) { qx = c; qy = 4 * mt + r; return true; }
    const int idx = 4 * r + (l31 & 3);  // 0..15 inside the lane group
    if (mt == 2) {  // rows 8, 9 x columns 0..7 (left group) and columns 8, 9 x rows 0..7 (right group): two-way conflicts here
      qx = right ? 8 + (idx & 1) : (idx & 7);
      qy = right ? (idx >> 1) : 8 + (idx >> 3);
      return true;
    }
    qx = 8 + (idx & 1); qy = 8 + ((idx >> 1) & 1);  // the corner
    return !right && idx < 4;
  } else {
    constexpr int TX = (WK + 7) / 8;
    const int tx = mt % TX, ty = mt / TX;
    qx = 8 * tx + c; qy = 4 * ty + r;
    const bool ok = qx < WK && qy < WK;
    qx = min(qx, WK - 1); qy = min(qy, WK - 1);
    return ok;
  }
}
template <int K> __host__ __device__ constexpr int n_tiles() { return conv_n(K) / 32; }
template <int K> __host__ __device__ constexpr int tpw() { return (m_tiles<K>() * n_tiles<K>() + NCOMPUTE - 1) / NCOMPUTE; }
// wave -> (first pixel tile, channel tile, number of tiles)
template <int K>
__device__ __forceinline__ void job(int wave, int& mt0, int& nt, int& cnt) {
  constexpr int per_n = (m_tiles<K>() + tpw<K>() - 1) / tpw<K>();  // waves per channel tile
  nt = wave / per_n;
  mt0 = (wave - nt * per_n) * tpw<K>();
  cnt = nt < n_tiles<K>() ? max(0, min(tpw<K>(), m_tiles<K>() - mt0)) : 0;
}

// ABL (developer ablations, SRX_RDB_ABLATE; 0 in the product -- every other value is a separate instantiation, so the product's
// code is untouched): 1 = the MFMAs multiply constant registers, no fragment is read from LDS (what the block costs without
// its LDS operand traffic); 2 = every fragment is read, no MFMA is issued (what the reads cost alone); 3 = ABL 0 without the
// stage epilogues (no conversion, no LDS image of the next source, no global stores): results are garbage in all three.
template <int K, int S>
__device__ __forceinline__ void frag_offsets(const Wave& w, int mt0, int (&ao)[tpw<K>()][3][2]) {
  constexpr int WS = src_w(S), D = reg_org(K) - src_org(S) - 1;  // source coordinate of tap (0,0) = region coordinate + D
  constexpr int BASE = src_off(S);
#pragma unroll
  for (int i = 0; i < tpw<K>(); ++i) {
    int qx, qy;
    tile_pixel<K>(mt0 + i, w.l31, qx, qy);
#pragma unroll
    for (int th = 0; th < 3; ++th) {
      const int sy = qy + D + th;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) ao[i][th][kk] = BASE + (sy * WS + qx + D) * 64 + (((kk * 2 + w.h) ^ (sy & 3)) << 4);
    }
  }
}
// A group: units U0 .. U0 + NU - 1 = sources S0 .. of conv K, 9 NU taps in one software pipeline.
template <int K, int S0, int NU, int U0, int ABL>
__device__ __forceinline__ void mma_group(const unsigned char* lds, Wave& w) {
  constexpr int NK = conv_n(K), TPW = tpw<K>(), NT = 9 * NU;
  int mt0, nt, cnt;
  job<K>(w.wave, mt0, nt, cnt);
  if (cnt == 0) return;
  // byte offset (from the start of LDS) of the fragment (unit, tile i, tap row th, k-half kk); the tap column is an immediate
  int ao[NU][TPW][3][2];
  frag_offsets<K, S0>(w, mt0, ao[0]);
  if constexpr (NU > 1) frag_offsets<K, S0 + 1>(w, mt0, ao[NU - 1]);
  constexpr int slot_first = unit_slot(U0), slot_last = unit_slot(U0 + NU - 1);
  const int nrow = nt * 32 + w.l31;
  const unsigned char* wrow = lds + OFF_W + nrow * 64;
  const int wsw = (nrow >> 2) & 3;
  // fragments of tap t + 2 are requested before tap t is multiplied: the wave is alone on its SIMD and an LDS read
  // under load (four waves reading, the next weights landing) returns after ~250 cycles -- two taps of MFMAs
  // (one tap ahead: 255 cycles per tap measured in-kernel, twice the MFMA time)
  bf16x8 wf[3][2], xf[3][TPW][2];
  if constexpr (ABL == 1) {  // constant operands: the matrix pipe alone
    const bf16x8 one = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < TPW; ++i) w.acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(one, one, w.acc[i], 0, 0, 0);
    }
    (void)wrow; (void)wsw; (void)ao;
    return;
  }
  auto fetch = [&](int tt, int set) {
    const int ui = tt / 9, t = tt - 9 * ui;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      wf[set][kk] = *reinterpret_cast<const bf16x8*>(wrow + (ui == 0 ? slot_first : slot_last) * SLOT_BYTES + t * NK * 64 + (((kk * 2 + w.h) ^ wsw) << 4));
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
        xf[set][i][kk] = *reinterpret_cast<const bf16x8*>(lds + ao[ui][i][t / 3][kk] + (t % 3) * 64);
    }
  };
  fetch(0, 0);
  fetch(1, 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t + 2 < NT) fetch(t + 2, (t + 2) % 3);
    if constexpr (ABL == 2) {  // the reads alone: every fragment is consumed by an empty asm, nothing is multiplied
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        asm volatile("" ::"v"(wf[t % 3][kk]));
#pragma unroll
        for (int i = 0; i < TPW; ++i) asm volatile("" ::"v"(xf[t % 3][i][kk]));
      }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < TPW; ++i)
        w.acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t % 3][kk], xf[t % 3][i][kk], w.acc[i], 0, 0, 0);
    // issue order inside this tap: one MFMA, then a share of the later tap's address arithmetic and fragment reads, so
    // that the wave's VALU / LDS issue slots fall into the MFMAs' shadows; the closing barrier keeps hipcc from sinking
    // the reads towards their use
#pragma unroll
    for (int g = 0; g < 2 * TPW; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// stage K < 5.  Forward: bias + LeakyReLU; backward: the LeakyReLU mask of c_{5-K} (read from the saved buffer, halo pixels
// included).  bf16 to the LDS image of source K + 1 (zeros outside the image), fp32 to the block's buffer / gradient buffer
// (tile centre only: the halo belongs to the neighbouring workgroups)
// backward: the saved activations c_{5-K} at this wave's stage-K pixels (their signs are the LeakyReLU masks), requested
// before the stage's last unit is multiplied so that the epilogue does not wait for them (clamped addresses: every lane
// loads, out-of-image lanes discard)
template <int K>
__device__ __forceinline__ void load_masks(const RdbArgs& a, const Wave& w, f32x4 (&m)[2][4]) {
  constexpr int ORG = reg_org(K), TPW = tpw<K>(), CH = 64 + 32 * (4 - K);
  int mt0, nt, cnt;
  job<K>(w.wave, mt0, nt, cnt);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int qx, qy;
    tile_pixel<K>(mt0 + i, w.l31, qx, qy);
    const int iy = min(max(w.ty0 + qy + ORG, 0), a.H - 1), ix = min(max(w.tx0 + qx + ORG, 0), a.W - 1);
    const float* mp = a.buf + ((size_t)(w.n_img * a.H + iy) * a.W + ix) * a.bld + CH + 4 * w.h;
#pragma unroll
    for (int g = 0; g < 4; ++g) m[i][g] = *reinterpret_cast<const f32x4*>(mp + 8 * g);
  }
}

template <int K, bool BWD>
__device__ __forceinline__ void epilogue_mid(const RdbArgs& a, unsigned char* lds, Wave& w, const f32x4 (&masks)[2][4]) {
  constexpr int WK = reg_w(K), ORG = reg_org(K), TPW = tpw<K>();
  constexpr int CH = BWD ? 64 + 32 * (4 - K) : 64 + 32 * (K - 1);  // channel slot of this stage's tensor in buf / gout
  int mt0, nt, cnt;
  job<K>(w.wave, mt0, nt, cnt);
  f32x4 b[4];
  if constexpr (!BWD) {
    const float* bias = reinterpret_cast<const float*>(lds + OFF_BIAS) + 32 * (K - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(bias + 8 * g + 4 * w.h);
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    int qx, qy;
    if (!tile_pixel<K>(mt0 + i, w.l31, qx, qy) || i >= cnt) continue;
    const int q = qy * WK + qx;
    const int oy = qy + ORG, ox = qx + ORG;
    const int iy = w.ty0 + oy, ix = w.tx0 + ox;
    const bool in_img = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    const bool centre = in_img && (unsigned)oy < (unsigned)RT && (unsigned)ox < (unsigned)RT;
    const size_t pix = (size_t)(w.n_img * a.H + min(max(iy, 0), a.H - 1)) * a.W + min(max(ix, 0), a.W - 1);
    float* gp = (BWD ? a.gout + pix * a.gld : a.buf + pix * a.bld) + CH + 4 * w.h;
    unsigned char* cp = lds + src_off(K + 1) + q * 64 + 8 * w.h;
    const int psw = qy & 3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (BWD) {
          v[e] = masks[i][g][e] > 0.f ? w.acc[i][4 * g + e] : w.acc[i][4 * g + e] * a.slope;
        } else {
          const float z = w.acc[i][4 * g + e] + b[g][e];
          v[e] = z > 0.f ? z : z * a.slope;
        }
      }
      bf16x4 pk = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      if (!in_img) pk = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x4*>(cp + ((g ^ psw) << 4)) = pk;
      if (centre) *reinterpret_cast<f32x4*>(gp + 8 * g) = v;
    }
  }
}

// last stage.  Forward: (acc + bias) * scale + x -> out; backward: acc + skip_scale * skip -> out
template <bool BWD>
__device__ __forceinline__ void epilogue_out(const RdbArgs& a, const unsigned char* lds, Wave& w, const f32x4 (&xs)[4],
                                             const f32x4 (&ex)[4]) {
  int mt0, nt, cnt;
  job<5>(w.wave, mt0, nt, cnt);
  if (cnt == 0) return;
  int qx, qy;
  tile_pixel<5>(mt0, w.l31, qx, qy);  // (8 x 8: every lane has a pixel)
  const int iy = w.ty0 + qy, ix = w.tx0 + qx;
  if (!((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)) return;
  const size_t pix = (size_t)(w.n_img * a.H + iy) * a.W + ix;
  const float* bias = reinterpret_cast<const float*>(lds + OFF_BIAS) + 128 + 32 * nt + 4 * w.h;
  float* op = a.out + pix * a.out_ld + 32 * nt + 4 * w.h;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v;
    if constexpr (BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = w.acc[0][4 * g + e] + a.skip_scale * xs[g][e];
      if (a.extra) v += ex[g];
    } else {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (w.acc[0][4 * g + e] + b[e]) * a.scale + xs[g][e];
      if (a.extra) v = v * a.post_scale + ex[g];
    }
    *reinterpret_cast<f32x4*>(op + 8 * g) = v;
  }
}

// the last stage's fp32 addend (forward: the skip connection's x; backward: the gradient that bypasses the block) for
// this wave's tile, requested a few units before it is needed
template <bool BWD>
__device__ __forceinline__ void load_skip(const RdbArgs& a, const Wave& w, f32x4 (&xs)[4], f32x4 (&ex)[4]) {
  int mt0, nt, cnt;
  job<5>(w.wave, mt0, nt, cnt);
  int qx, qy;
  tile_pixel<5>(min(mt0, m_tiles<5>() - 1), w.l31, qx, qy);
  const int iy = min(w.ty0 + qy, a.H - 1), ix = min(w.tx0 + qx, a.W - 1);  // (clamped: stores are masked, loads are not)
  const size_t pix = (size_t)(w.n_img * a.H + iy) * a.W + ix;
  const float* xp = (BWD ? a.skip + pix * a.skip_ld : a.src + pix * a.ld) + 32 * min(nt, 1) + 4 * w.h;
#pragma unroll
  for (int g = 0; g < 4; ++g) xs[g] = *reinterpret_cast<const f32x4*>(xp + 8 * g);
  if (a.extra) {  // (workgroup-uniform)
    const float* ep = a.extra + pix * a.extra_ld + 32 * min(nt, 1) + 4 * w.h;
#pragma unroll
    for (int g = 0; g < 4; ++g) ex[g] = *reinterpret_cast<const f32x4*>(ep + 8 * g);
  }
}

template <int G, bool BWD, int ABL>
__device__ __forceinline__ void run_groups(const RdbArgs& a, unsigned char* lds, Wave& w, int tid, f32x4 (&xs)[4],
                                           f32x4 (&ex)[4]) {
  constexpr int U0 = grp_first(G), NU = grp_count(G), K = unit_conv(U0), S0 = U0 - unit_first(K), SL = S0 + NU - 1;
  static_assert(unit_conv(U0 + NU - 1) == K, "a group stays inside one conv");
  // group G's weights (and, for S0 == 0, the previous conv's output image) are in LDS; the ring half of group G - 1 is free
  rdb_lds_barrier();
#pragma unroll
  for (int u = U0; u < U0 + NU; ++u) rdb_stamp(a, lds, w.wave, w.lane, 2 + 2 * u);  // (developer stamps: a group's units open together)
  if (w.wave >= NCOMPUTE) {
    const int lt = tid - NCOMPUTE * 64;
    if constexpr (G >= 1 && G + 1 < NGROUPS) {  // (groups 0 and 1 are requested before the first barrier)
      dma_unit<grp_first(G + 1)>(a, (unsigned)(size_t)lds, lt);
      if constexpr (grp_count(G + 1) > 1) dma_unit<grp_first(G + 1) + 1>(a, (unsigned)(size_t)lds, lt);
    }
    // group G + 1 has landed before the next barrier (nothing else is in flight: a group is requested one group ahead)
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), expcnt / lgkmcnt untouched
  } else {
    if (S0 == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) w.acc[i][r] = 0.f;
    }
    if constexpr (U0 == unit_first(5)) load_skip<BWD>(a, w, xs, ex);
    f32x4 masks[2][4];
    if constexpr (BWD && SL == K && K < 5) load_masks<K>(a, w, masks);
    mma_group<K, S0, NU, U0, ABL>(lds, w);
    if constexpr (SL == K && ABL != 3) {  // the group ends stage K
      if constexpr (K < 5) epilogue_mid<K, BWD>(a, lds, w, masks);
      else epilogue_out<BWD>(a, lds, w, xs, ex);
    }
  }
#pragma unroll
  for (int u = U0; u < U0 + NU; ++u) rdb_stamp(a, lds, w.wave, w.lane, 3 + 2 * u);
  if constexpr (G + 1 < NGROUPS) run_groups<G + 1, BWD, ABL>(a, lds, w, tid, xs, ex);
}

template <bool BWD, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void rdb_kernel(const RdbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  Wave w;
  w.lane = tid & 63; w.h = w.lane >> 5; w.l31 = w.lane & 31; w.wave = srx_uniform(tid >> 6);
  // workgroups are dealt to the eight XCDs round robin; tiles are numbered so that an XCD gets a contiguous run of them -- at
  // batch 16 x 32 x 32 two whole images per XCD: the 5-pixel halo a tile re-reads from its neighbours (an 18 x 18 patch for 8 x 8
  // outputs) and the shared weight stream then hit in that XCD's own L2 instead of crossing the fabric from every other one
  int b = blockIdx.x;
  if ((gridDim.x & 7) == 0) b = (b & 7) * (int)(gridDim.x >> 3) + (b >> 3);
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  w.n_img = b / a.tiles_y; w.ty0 = ty * RT; w.tx0 = tx * RT;

  rdb_stamp(a, lds, w.wave, w.lane, 0);
  if (w.wave >= NCOMPUTE) {  // the first two groups of weight units (both halves of the ring)
    const int lt = tid - NCOMPUTE * 64;
    static_assert(grp_first(2) == 4 && grp_count(0) == 2 && grp_count(1) == 2, "groups 0 and 1 are units 0..3");
    dma_unit<0>(a, (unsigned)(size_t)lds, lt);
    dma_unit<1>(a, (unsigned)(size_t)lds, lt);
    dma_unit<2>(a, (unsigned)(size_t)lds, lt);
    dma_unit<3>(a, (unsigned)(size_t)lds, lt);
  }
  // the input patch, rounded to bf16 once: 324 pixels x 8 groups of 8 channels; every load is issued before the first
  // conversion (out-of-image pixels read a clamped address and are zeroed afterwards)
  constexpr int PW = RT + 10, ITEMS = PW * PW * 8, ROUNDS = (ITEMS + NTHREADS - 1) / NTHREADS;  // 6
  {
    f32x4 v[ROUNDS][2];
    bool ok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int i = min(r * NTHREADS + tid, ITEMS - 1);
      const int p = i >> 3, g = i & 7;
      const int py = p / PW, px = p - py * PW;
      const int iy = w.ty0 - 5 + py, ix = w.tx0 - 5 + px;
      ok[r] = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
      const float* xp = a.src + ((size_t)(w.n_img * a.H + cy) * a.W + cx) * a.ld + 8 * g;
      v[r][0] = *reinterpret_cast<const f32x4*>(xp);
      v[r][1] = *reinterpret_cast<const f32x4*>(xp + 4);
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int i = r * NTHREADS + tid;
      if (i >= ITEMS) continue;
      const int p = i >> 3, g = i & 7;
      f32x4 v0 = v[r][0], v1 = v[r][1];
      if (!ok[r]) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
      if constexpr (BWD) { v0 *= a.scale; v1 *= a.scale; }  // g5 = scale * dy, rounded after the product (as autograd hands it on)
      const bf16x8 pk = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3],
                         (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
      *reinterpret_cast<bf16x8*>(lds + (g >> 2) * (PW * PW * 64) + p * 64 + (((g & 3) ^ ((p / PW) & 3)) << 4)) = pk;
    }
  }
  if constexpr (!BWD) {
    if (tid < 192) {
      const int k = tid < 128 ? tid >> 5 : 4;
      reinterpret_cast<float*>(lds + OFF_BIAS)[tid] = a.bias[k][tid < 128 ? (tid & 31) : tid - 128];
    }
  }
  if (w.wave >= NCOMPUTE) dma_wait<2, 0>(srx_uniform((tid - NCOMPUTE * 64) >> 6));  // group 0 has landed (group 1 = units 2, 3 may fly on)
  rdb_stamp(a, lds, w.wave, w.lane, 1);
  f32x4 xs[4], ex[4];
  run_groups<0, BWD, ABL>(a, lds, w, tid, xs, ex);
  if (a.dbg && w.lane < DBG_SLOTS)  // (each wave copies the stamps its own lane 0 wrote)
    a.dbg[((size_t)blockIdx.x * 8 + w.wave) * DBG_SLOTS + w.lane] = reinterpret_cast<const unsigned*>(lds + LDS_BYTES)[w.wave * DBG_SLOTS + w.lane];
}

// OIHW fp32 weights of every block's five convs -> the bf16 unit streams rdb_kernel reads.  One thread per (row, 8 channels)
// of a unit: it reads the nine taps of its eight (input or output) channels -- forward 72 adjacent floats, backward eight runs
// of nine -- and writes the nine 16-byte pieces, one per tap.  (One thread per 16 destination bytes made the nine taps' threads
// pull the same lines through the L2 nine times: 66 us per launch for 66 MB of weights.)
// bwd = 0: stage K = conv K, rows = its output channels, k = the source's 32 input channels, taps as stored.
// bwd = 1: stage K, source S: the conv is k = 5 (S < 2: its output channels 32 S ..) or 6 - S; rows = the INPUT channels of
// that conv that make up the stage's slice (c_{5-K}, or x for K = 5), k = 32 of its output channels, taps flipped.
__global__ void rdb_pack_kernel(const float* const* __restrict__ wtab, unsigned char* __restrict__ dst, int nblk, int bwd) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int per_blk = PACKED_BYTES / (16 * 9);  // (a unit is 9 taps x rows x 64 bytes)
  if (gid >= (int64_t)nblk * per_blk) return;
  const int blk = (int)(gid / per_blk);
  int c = (int)(gid - (int64_t)blk * per_blk) * (16 * 9);  // byte offset of the unit + 144 x (item in the unit)
  int u = 0;
#pragma unroll 1
  while (u + 1 < NUNITS && c >= unit_off(u + 1)) ++u;
  const int K = unit_conv(u), S = u - unit_first(K), NK = conv_n(K);
  const int item = (c - unit_off(u)) / (16 * 9);
  const int n = item >> 2, js = item & 3;
  const int j = js ^ ((n >> 2) & 3);  // logical quad stored in this position
  float v[8][9];                      // [channel][tap as stored in OIHW]
  if (!bwd) {
    const int cin0 = (S < 2 ? 32 * S : 64 + 32 * (S - 2)) + 8 * j;
    const float* wk = wtab[blk * 5 + (K - 1)] + ((size_t)n * conv_cin(K) + cin0) * 9;
    if ((reinterpret_cast<uintptr_t>(wk) & 15) == 0) {  // (288-byte items: aligned whenever the weight tensor is)
      f32x4 q[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) q[i] = reinterpret_cast<const f32x4*>(wk)[i];
#pragma unroll
      for (int i = 0; i < 72; ++i) v[i / 9][i % 9] = q[i >> 2][i & 3];
    } else {
#pragma unroll
      for (int i = 0; i < 72; ++i) v[i / 9][i % 9] = wk[i];
    }
  } else {
    const int k = S < 2 ? 5 : 6 - S;
    const int co0 = (S < 2 ? 32 * S : 0) + 8 * j;
    const int ci = (K < 5 ? 64 + 32 * (4 - K) : 0) + n;
    const float* wk = wtab[blk * 5 + (k - 1)] + ((size_t)co0 * conv_cin(k) + ci) * 9;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 9; ++t) v[e][t] = wk[(size_t)e * conv_cin(k) * 9 + t];
  }
  unsigned char* d = dst + (size_t)blk * PACKED_BYTES + unit_off(u) + n * 64 + js * 16;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ts = bwd ? 8 - t : t;  // taps flipped for the data gradient
    const bf16x8 pk = {(__bf16)v[0][ts], (__bf16)v[1][ts], (__bf16)v[2][ts], (__bf16)v[3][ts],
                       (__bf16)v[4][ts], (__bf16)v[5][ts], (__bf16)v[6][ts], (__bf16)v[7][ts]};
    *reinterpret_cast<bf16x8*>(d + (size_t)t * NK * 64) = pk;
  }
}

}  // namespace

extern "C" size_t srx_rdb_packed_bytes(void) { return (size_t)PACKED_BYTES; }

static int rdb_pack_impl(const float* const* w_table_dev, int nblk, void* dst, int bwd, void* stream) {
  SRX_REQUIRE(w_table_dev && dst && nblk > 0 && nblk <= 4096, "rdb_pack: bad argument");
  const int64_t n = (int64_t)nblk * (PACKED_BYTES / (16 * 9));
  hipLaunchKernelGGL(rdb_pack_kernel, dim3((unsigned)srx_cdiv(n, 256)), dim3(256), 0, srx_stream(stream), w_table_dev,
                     reinterpret_cast<unsigned char*>(dst), nblk, bwd);
  SRX_CHECK_LAUNCH("rdb_pack_kernel");
  return SRX_OK;
}
extern "C" int srx_rdb_pack(const float* const* w_table_dev, int nblk, void* dst, void* stream) {
  return rdb_pack_impl(w_table_dev, nblk, dst, 0, stream);
}
extern "C" int srx_rdb_pack_bwd(const float* const* w_table_dev, int nblk, void* dst, void* stream) {
  return rdb_pack_impl(w_table_dev, nblk, dst, 1, stream);
}

template <bool BWD>
static int rdb_launch(RdbArgs& a, const char* what, void* stream) {
  a.tiles_x = (int)srx_cdiv(a.W, RT); a.tiles_y = (int)srx_cdiv(a.H, RT);
  const int64_t grid = (int64_t)a.N * a.tiles_x * a.tiles_y;
  SRX_REQUIRE(grid < (1LL << 31), "%s: grid too large", what);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rdb_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rdb_kernel<BWD, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rdb_kernel<BWD, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rdb_kernel<BWD, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  if (const int abl = srx_dev().rdb_ablate; abl >= 1 && abl <= 3 && !a.dbg) {  // developer ablations (garbage results): see mma_group
    const dim3 g((unsigned)grid), b(NTHREADS);
    if (abl == 1) hipLaunchKernelGGL((rdb_kernel<BWD, 1>), g, b, LDS_BYTES, srx_stream(stream), a);
    else if (abl == 2) hipLaunchKernelGGL((rdb_kernel<BWD, 2>), g, b, LDS_BYTES, srx_stream(stream), a);
    else hipLaunchKernelGGL((rdb_kernel<BWD, 3>), g, b, LDS_BYTES, srx_stream(stream), a);
    SRX_CHECK_LAUNCH(what);
    return SRX_OK;
  }
  // algorithmic FLOPs: the five convs (or their data gradients) on the tile pixels; the halo recompute is not counted
  const double fl = 2.0 * a.N * a.H * a.W * 9.0 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64);
  char nm[112];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "rdb_kernel<%d> MxNxK=%dx192x(576..1728)", BWD ? 1 : 0, a.N * a.H * a.W);
  SRX_LAUNCH_PROF(nm, fl, rdb_kernel<BWD>, dim3((unsigned)grid), dim3(NTHREADS), LDS_BYTES + (a.dbg ? 8 * DBG_SLOTS * 4 : 0), srx_stream(stream), a);
  SRX_CHECK_LAUNCH(what);
  return SRX_OK;
}

extern "C" int srx_rdb_fwd(int N, int H, int W, float* buf, int ld, const void* wpk, const float* const* bias5, float scale,
                           float slope, float post_scale, const float* extra, int extra_ld, float* out, int out_ld,
                           void* stream) {
  SRX_REQUIRE(buf && wpk && bias5 && out, "rdb_fwd: null pointer");
  // other workgroups read the x channels of `buf` (with their halo) while this one writes `out`: `out` is never the block
  // buffer itself; `extra` may be an OLDER block's buffer but never `out`
  SRX_REQUIRE(out != buf, "rdb_fwd: out must not alias the block buffer (neighbouring tiles read their halo from it)");
  SRX_REQUIRE(N > 0 && H > 0 && W > 0 && ld >= 192 && ld % 4 == 0 && out_ld >= 64 && out_ld % 4 == 0,
              "rdb_fwd: the block buffer needs >= 192 channels per pixel (x, c1..c4), the output >= 64, in whole quads");
  SRX_REQUIRE((int64_t)N * H * W < (1 << 24) && (int64_t)N * H * W * ld < (1LL << 40), "rdb_fwd: more than 2^24 pixels; tile the image");
  RdbArgs a{};
  a.src = buf; a.buf = buf; a.wpk = reinterpret_cast<const unsigned char*>(wpk); a.out = out;
  for (int k = 0; k < 5; ++k) {
    SRX_REQUIRE(bias5[k], "rdb_fwd: null bias %d", k);
    a.bias[k] = bias5[k];
  }
  SRX_REQUIRE(!extra || (extra_ld >= 64 && extra_ld % 4 == 0 && extra != out), "rdb_fwd: the second addend needs >= 64 channels per pixel and a tensor of its own");
  a.N = N; a.H = H; a.W = W; a.ld = ld; a.bld = ld; a.out_ld = out_ld;
  a.scale = scale; a.slope = slope; a.post_scale = post_scale; a.extra = extra; a.extra_ld = extra_ld;
  return rdb_launch<false>(a, "rdb_fwd", stream);
}

// Developer aid (tools/bench_rdb.py stamps): srx_rdb_fwd with in-kernel clock stamps, dbg: [workgroups][8 waves][48] unsigned
extern "C" int srx_rdb_fwd_dbg(int N, int H, int W, float* buf, int ld, const void* wpk, const float* const* bias5, float scale,
                               float slope, float* out, int out_ld, void* stream, unsigned* dbg) {
  SRX_REQUIRE(buf && wpk && bias5 && out && dbg && out != buf && N > 0 && H > 0 && W > 0 && ld >= 192 && out_ld >= 64, "rdb_fwd_dbg: bad argument");
  RdbArgs a{};
  a.src = buf; a.buf = buf; a.wpk = reinterpret_cast<const unsigned char*>(wpk); a.out = out;
  for (int k = 0; k < 5; ++k) a.bias[k] = bias5[k];
  a.N = N; a.H = H; a.W = W; a.ld = ld; a.bld = ld; a.out_ld = out_ld;
  a.scale = scale; a.slope = slope; a.post_scale = 1.f;
  a.dbg = dbg;
  return rdb_launch<false>(a, "rdb_fwd_dbg", stream);
}

extern "C" int srx_rdb_bwd(int N, int H, int W, const float* dy, int dy_ld, float scale, const float* buf, int ld,
                           const void* wpk_bwd, float slope, float* gbuf, int gld, const float* skip, int skip_ld,
                           float skip_scale, const float* extra, int extra_ld, float* dx, int dx_ld, void* stream) {
  SRX_REQUIRE(dy && buf && wpk_bwd && gbuf && skip && dx, "rdb_bwd: null pointer");
  SRX_REQUIRE(N > 0 && H > 0 && W > 0 && dy_ld >= 64 && dy_ld % 4 == 0 && ld >= 192 && ld % 4 == 0 && gld >= 192 && gld % 4 == 0 &&
                  skip_ld >= 64 && skip_ld % 4 == 0 && dx_ld >= 64 && dx_ld % 4 == 0,
              "rdb_bwd: dy / skip / dx need >= 64 channels per pixel, the activation and gradient buffers >= 192, in whole quads");
  SRX_REQUIRE(dx != dy && dx != skip && (const float*)gbuf != buf, "rdb_bwd: dx must not alias dy or skip (neighbouring tiles read their halo)");
  SRX_REQUIRE((int64_t)N * H * W < (1 << 24), "rdb_bwd: more than 2^24 pixels; tile the image");
  RdbArgs a{};
  a.src = dy; a.buf = const_cast<float*>(buf); a.gout = gbuf; a.skip = skip; a.out = dx;
  a.wpk = reinterpret_cast<const unsigned char*>(wpk_bwd);
  a.N = N; a.H = H; a.W = W; a.ld = dy_ld; a.bld = ld; a.gld = gld; a.skip_ld = skip_ld; a.out_ld = dx_ld;
  SRX_REQUIRE(!extra || (extra_ld >= 64 && extra_ld % 4 == 0 && extra != dx), "rdb_bwd: the second addend needs >= 64 channels per pixel and a tensor of its own");
  a.scale = scale; a.slope = slope; a.skip_scale = skip_scale; a.extra = extra; a.extra_ld = extra_ld;
  return rdb_launch<true>(a, "rdb_bwd", stream);
}
