// 3x3 / stride 1 / pad 1 convolutions of wide layers as Winograd F(2x2, 3x3) on the fp32 matrix cores (round 5; recut in round 6).
//
// The direct gather-GEMM (gconv.hip) runs the VGG19 layers of the perceptual loss (srgan/loss.py:30-54: 344 of the SRGAN
// step's 692 GFLOP) at 0.95 of what v_mfma_f32_32x32x2_f32 gives at the clock the part holds under that load -- the only
// way to make them faster is to multiply less.  Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input
// tile with 16 multiplications per (input channel, output channel) pair instead of 36:
//
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// i.e. sixteen independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] U_xi[ci][co]  (xi = the 16 positions of the
// transformed 4x4 tile) with 2.25x fewer multiply-adds than the direct form, all in fp32: the result differs from the
// direct fp32 convolution by rounding only (measured 2-3x its own rounding error against fp64, i.e. ~5e-7 of the tensor's
// scale: tools/experiments/wino_error.py), far inside the 1e-3 the parity tests allow.
//
// What shapes the kernel (round 6, measured: tools/probe/mfma_valu.hip, tools/lab/): ON gfx950 THE f32 MFMA RUNS ON THE VECTOR
// ALUs -- a v_mfma_f32_32x32x2_f32 and a VALU instruction of ANY wave of the SIMD never overlap; every VALU instruction costs
// ~7 clocks of matrix time (v_pk_add_f32 the same 7 for two adds).  So the transforms are not "free under the MFMAs": the kernel
// counts VALU instructions (packed adds, tile coordinates by one float multiply, offsets kept in LDS), spreads them evenly over the
// four SIMDs, and uses co-residency only for what it can hide: memory latency, barrier waits, the prologue's first round trip.
//
// One workgroup = 32 tiles (128 output pixels) x BN output channels x all 16 xi, 256 threads = one wave per SIMD, 80 KB of LDS:
// TWO workgroups share a CU, each with its own barriers (round 5: one 512-thread workgroup owned the CU and its 4.4 us of prologue
// + epilogue per work item ran with the matrix pipe idle).
//   * every wave owns FOUR xi (accumulators 4 xi x BN / 32 channel tiles x 16 registers) and multiplies them for all 32 tiles:
//     U fragments come STRAIGHT from global memory into MFMA operand registers in 8-byte granules through a two-slot ring (each xi
//     belongs to exactly one wave, so nothing is shared through LDS; srx_wino_pack lays U out so that a wave's load is contiguous),
//     V fragments are one ds_read_b128 per eight MFMAs;
//   * all four waves gather the 4x4 input patches of the 32 tiles for a chunk of 16 input channels -- one (tile, channel pair) per
//     thread: 16 x 8-byte loads whose byte offsets (padding pixels: an out-of-range descriptor offset) sit in LDS, not in
//     registers -- apply B^T d B with 32 packed adds and write the sixteen V_xi rows to a two-stage LDS ring (32 KB per stage,
//     XOR-swizzled 64-byte rows); the loads of chunk c + 1 ride between the MFMAs of the first half of chunk c, its transform
//     between those of the second half (sched_group_barrier), so a wave alone on its SIMD keeps issuing;
//   * epilogue, one 32-channel tile at a time: the accumulators (U as the A operand, so a lane holds four consecutive channels
//     of one tile) go to LDS as M[xi][tile][channel], every thread takes one (tile, channel quad), applies A^T M A, adds the bias,
//     applies ReLU or the ReLU mask of the layer below (data gradients: the fold of srx_conv2d_bwd_data_act) and stores four
//     pixels x 16 bytes.
// The data gradient of such a layer IS such a layer (channels swapped, taps flipped): srx_wino_pack(..., transpose = 1).
// Small layers split the input channels over several workgroups (partial outputs + a streaming fix-up); layers that cut into a
// non-integer number of rounds of the chip's CUs run whole tiles for the full rounds and cut only the tiles of the last round along
// the input channels (tile-local partial outputs + wino_tail_fixup_kernel); the planner (wino_plan) picks BN (64 or 32) and the split
// from a cost model.  Inference adds LeakyReLU / PReLU, a skip addend and nn.PixelShuffle(2) in the store (srx_wino_fwd_act).
#include "srx_common.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WT = 32;       // tiles per workgroup
constexpr int WKC = 32;      // input channels per chunk of the packed U (the planner's unit of a channel split)
constexpr int VKC = 16;      // input channels per LDS stage
constexpr int STAGE_BYTES = 16 * WT * VKC * 4;  // 32768: V of one stage
constexpr int RING_BYTES = 2 * STAGE_BYTES;     // the epilogue's M[16][32][32] (one 32-channel tile at a time) reuses both stages
constexpr int POFF_BYTES = 256 * 16 * 4;        // the 16 patch-pixel byte offsets of every thread, [pixel row][thread] uint4s
constexpr int WINO_LDS = RING_BYTES + POFF_BYTES;  // 81920: half a CU's LDS (the statistics epilogue reuses the offsets' area)
constexpr int WINO_THREADS = 256;

struct WinoArgs {
  const float* in; const float* upk; const float* bias; const float* mask; float* out; float* part;
  float* stats;  // training-mode BatchNorm partials of the pre-activation output: [tile block][Cout][2] = (sum, sum of squares), or null
  int N, H, W, Cin, Cout;
  int TH, TW, T, tblocks, ncb, nch, zsplit;
  float inv_TW, inv_TH;  // tile coordinates by one float multiply each while T < 2^24 (srx_divmod); exact integer division above
  int relu;
  // inference forms (FoldedConv, round 5): `lrelu` != 0: LeakyReLU / single-parameter PReLU with slope `slope` instead of ReLU;
  // `add`: a tensor laid out like `out`, added after the activation (the skip of a residual block, srgan/residual.py:86-91)
  int lrelu; float slope; const float* add;
  int shuffle;  // != 0: nn.PixelShuffle(2) in the store (forward only): the value is the channel count per sub-pixel, Cout / 4
  // Tail split (round 5): work items [0, full) are whole (tile block, channel block) tiles; the tiles of the last, partly filled
  // round of the chip -- item full + t is part t % tsplit of tile full + t / tsplit -- take 1 / tsplit of the input channels each and
  // write raw partial outputs to tpart[t][32 tiles][4 pixels][BN]; wino_tail_fixup_kernel finishes those tiles only.
  // (Without a tail: full = every item.)  2.25 rounds of whole tiles cost 3 rounds; 2 rounds + a quarter-length third cost ~2.4.
  int full, tsplit; float* tpart;
  unsigned in_bytes, upk_bytes;
  size_t out_elems;  // N * H * W * Cout (stride between the partial outputs of two splits)
};

__device__ __forceinline__ f32x2 srx_bload2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// two adds for one instruction (see the header: beside f32 MFMAs the instruction count is what costs)
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
  f32x2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_add(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_add(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
  const f32x2 lo = pk_sub(f32x2{a[0], a[1]}, f32x2{b[0], b[1]}), hi = pk_sub(f32x2{a[2], a[3]}, f32x2{b[2], b[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// tile t -> (image, tile row, tile column)
__device__ __forceinline__ void wino_tile_coords(const WinoArgs& a, int t, int& n, int& th, int& tw) {
  if (a.T < (1 << 24)) {  // (uniform)
    int r;
    srx_divmod(t, a.TW, a.inv_TW, r, tw);
    srx_divmod(r, a.TH, a.inv_TH, n, th);
  } else {
    tw = t % a.TW; const int r = t / a.TW; th = r % a.TH; n = r / a.TH;
  }
}

template <int BN>
__global__ __launch_bounds__(WINO_THREADS, 2) void wino_kernel(const WinoArgs a) {
  constexpr int NJ = BN / 32;  // channel tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  // Work item = (split, channel block, tile block), tile blocks fastest.  Workgroups are dealt round-robin over the 8 XCDs; this
  // remap gives every XCD a contiguous run of items (as gconv.hip's weight-gradient kernels do) -- the tile blocks of a run share
  // their U block and their halo rows in ONE L2 (a speed matter only: with the plain order every XCD pulled all of U through the
  // fabric, 142 MB of HBM reads for a 512 -> 512 layer whose operands are 22 MB; PMC, round 5)
  int b = (int)blockIdx.x;
  const bool tailw = b >= a.full;  // (workgroup-uniform) a part of a tail tile
  int tb, cb, z, zs;
  if (!tailw) {
    const int W = a.full, xcd = b & 7, slot = b >> 3, qq = W >> 3, rr = W & 7;
    b = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + slot;
    tb = b % a.tblocks; b /= a.tblocks;
    cb = b % a.ncb;
    z = b / a.ncb; zs = a.zsplit;
  } else {
    const int t = b - a.full, tile = a.full + t / a.tsplit;
    z = t % a.tsplit; zs = a.tsplit;
    tb = tile % a.tblocks; cb = tile / a.tblocks;
  }
  // this split's stages of 16 input channels: [kc0, kc1) -- whole 32-channel chunks of the packed U
  const int kc0 = 2 * srx_uniform(z * a.nch / zs), kc1 = 2 * srx_uniform((z + 1) * a.nch / zs);  // (nch <= 2^15 / 32, zs <= 16)
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in, a.in_bytes);
  const __amdgpu_buffer_rsrc_t ru = srx_rsrc(a.upk, a.upk_bytes);

  // ---- this thread's tile (the one it gathers patches for AND the one it finishes in the epilogue) and channel pair / quad
  const int tl = tid >> 3, pr = tid & 7;
  const int t = tb * WT + tl;
  const bool tvalid = t < a.T;
  int tn, th, tw;
  wino_tile_coords(a, t, tn, th, tw);
  u32x4* spoff = reinterpret_cast<u32x4*>(smem + RING_BYTES);  // [4][256]
  {
    // byte offset of patch pixel (i, j) = base + i * (W Cin 4) + j * (Cin 4): per-lane adds of uniform steps, no per-lane multiplies;
    // a padding pixel (or a tile past the end) gets an out-of-range offset, which reads 0
    const int ih0 = 2 * th - 1, iw0 = 2 * tw - 1;
    const unsigned base = 4u * (unsigned)(((tn * a.H + ih0) * a.W + iw0) * a.Cin + 2 * pr);  // (may wrap for padding pixels: never used then)
    const unsigned cstep = 4u * (unsigned)a.Cin, rstep = cstep * (unsigned)a.W;
    bool cok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cok[j] = (unsigned)(iw0 + j) < (unsigned)a.W;
    unsigned rowb = base;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool rok = tvalid && (unsigned)(ih0 + i) < (unsigned)a.H;
      u32x4 o;
      unsigned v = rowb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = (rok && cok[j]) ? v : 0xffffffffu; v += cstep; }
      spoff[i * WINO_THREADS + tid] = o;  // (read back by this thread only: no barrier)
      rowb += rstep;
    }
  }
  // ---- multiplier state: this wave's four xi; packed U, float4 index (((xi * (Cout / 32) + jg) * nch + kc32) * 4 + s4) * 64 + lane:
  //      a lane part in a VGPR, the (xi, channel tile) part in SGPRs
  const int xi0 = 4 * wave;
  const int cot = a.Cout >> 5;
  const unsigned ulane = 16u * (unsigned)lane;
  unsigned ubase[4][NJ];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int j = 0; j < NJ; ++j) ubase[x][j] = (unsigned)srx_uniform((((xi0 + x) * cot + cb * NJ + j) * a.nch) * 4096);

  f32x16 acc[4][NJ];  // (never zero-filled -- 128 moves that the matrix pipe would wait for: the first MFMAs take C = 0)

  f32x2 pd[16];        // the patch of the stage being gathered
  f32x2 uf[2][4][NJ];  // U fragments, two slots: (phase & 1); a phase = two of the four k-pairs of a sub-step = 8 NJ MFMAs
  f32x4 vf[4];
  auto load_patch = [&](int kc) {
    const unsigned so = (unsigned)srx_uniform(kc * (VKC * 4));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 o = spoff[i * WINO_THREADS + tid];
#pragma unroll
      for (int j = 0; j < 4; ++j) pd[4 * i + j] = srx_bload2(rin, o[j], so);
    }
  };
  // phase ph (0..3) of stage kc: sub-step s = ph >> 1 of the stage (sub-step 2 (kc & 1) + s of the packed 32-channel chunk),
  // k-pairs 2 (ph & 1), 2 (ph & 1) + 1 of it: the second 8 bytes of the packed float4
  auto load_u = [&](int kc, int ph) {
    const unsigned so = (unsigned)srx_uniform(((kc >> 1) * 4 + (kc & 1) * 2 + (ph >> 1)) * 1024 + (ph & 1) * 8);
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int j = 0; j < NJ; ++j) uf[ph & 1][x][j] = srx_bload2(ru, ulane, ubase[x][j] + so);
  };
  // B^T d B of this thread's patch (rows first, in place; then the columns), the sixteen results to row (xi, tl) of stage `st`:
  // 16 floats per row, 16-byte quad q of the row at quad q ^ ((tl >> 2) & 3) (conflict-free ds_read_b128 of the MFMA operands)
  auto stage_patch = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 r0 = pk_sub(pd[4 * i + 0], pd[4 * i + 2]), r1 = pk_add(pd[4 * i + 1], pd[4 * i + 2]), r2 = pk_sub(pd[4 * i + 2], pd[4 * i + 1]),
                  r3 = pk_sub(pd[4 * i + 1], pd[4 * i + 3]);
      pd[4 * i + 0] = r0; pd[4 * i + 1] = r1; pd[4 * i + 2] = r2; pd[4 * i + 3] = r3;
    }
    float* base = reinterpret_cast<float*>(smem + st * STAGE_BYTES) + tl * VKC + (((pr >> 1) ^ ((tl >> 2) & 3)) << 2) + ((pr & 1) << 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 v0 = pk_sub(pd[0 + j], pd[8 + j]), v1 = pk_add(pd[4 + j], pd[8 + j]), v2 = pk_sub(pd[8 + j], pd[4 + j]), v3 = pk_sub(pd[4 + j], pd[12 + j]);
      *reinterpret_cast<f32x2*>(base + (0 + j) * (WT * VKC)) = v0;   // xi = 4 i + j
      *reinterpret_cast<f32x2*>(base + (4 + j) * (WT * VKC)) = v1;
      *reinterpret_cast<f32x2*>(base + (8 + j) * (WT * VKC)) = v2;
      *reinterpret_cast<f32x2*>(base + (12 + j) * (WT * VKC)) = v3;
    }
  };
  // sub-step s of the stage st: quads 2s (lanes 0..31) and 2s + 1 (lanes 32..63) of this wave's four xi
  const int vrow = l31 * VKC, vsw = (l31 >> 2) & 3;
  auto read_v = [&](int st, int s) {
    const float* sv = reinterpret_cast<const float*>(smem + st * STAGE_BYTES) + vrow + (((2 * s + h) ^ vsw) << 2);
#pragma unroll
    for (int x = 0; x < 4; ++x) vf[x] = *reinterpret_cast<const f32x4*>(sv + (xi0 + x) * (WT * VKC));
  };
  auto mma = [&](int ph, auto first) {
    constexpr f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (decltype(first)::value && e == 0)
            acc[x][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[ph & 1][x][j][e], vf[x][2 * (ph & 1) + e], zero, 0, 0, 0);
          else
            acc[x][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[ph & 1][x][j][e], vf[x][2 * (ph & 1) + e], acc[x][j], 0, 0, 0);
        }
  };
  constexpr std::false_type later{};
  constexpr std::true_type very_first{};

  // ---- prologue: first stage gathered, its first U fragments requested
  load_patch(kc0);
  load_u(kc0, 0);
  load_u(kc0, 1);
  stage_patch(0);
  __syncthreads();
  // ---- stage loop.  The first stage is peeled (its first MFMAs start the accumulators) and so is the last (it requests nothing
  // beyond itself, so every vmcnt of the steady state is exact); a split covers whole 32-channel chunks: at least two stages
  int st = 0;
  // first half of a stage: the next stage's patch requests and this stage's later U fragments ride between the MFMAs of phases 0, 1;
  // second half: the patch transform and the next stage's first U fragments between the MFMAs of phases 2, 3
  auto stage = [&](int kc, auto first) {
    load_patch(kc + 1);
    read_v(st, 0);
    mma(0, first);
    load_u(kc, 2);
    mma(1, later);
    load_u(kc, 3);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 16 * NJ; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, NJ == 2 ? 1 : 2, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    read_v(st, 1);  // (before the transform's LDS stores in program order: the compiler cannot tell the two stages apart)
    stage_patch(st ^ 1);
    mma(2, later);
    load_u(kc + 1, 0);
    mma(3, later);
    load_u(kc + 1, 1);
    __syncthreads();
    st ^= 1;
  };
  stage(kc0, very_first);
  for (int kc = kc0 + 1; kc + 1 < kc1; ++kc) stage(kc, later);
  {
    const int kc = kc1 - 1;
    read_v(st, 0);
    mma(0, later);
    load_u(kc, 2);
    mma(1, later);
    load_u(kc, 3);
    read_v(st, 1);
    mma(2, later);
    mma(3, later);
    __syncthreads();
  }

  // ---- epilogue, one 32-channel tile j at a time: M[xi][tile][32 channels] through LDS (16-byte chunk c of row (xi, t) at chunk
  // c ^ (t & 7)), then A^T M A per (tile, channel quad).  Addresses: ONE per-lane byte offset, relative to the first pixel row the
  // tile block touches; everything else (channel tile, the four pixels of the 2x2 tile, the sub-pixel of a PixelShuffle store) is
  // uniform and travels in the buffer descriptor's base or the scalar offset of the access
  float* sM = reinterpret_cast<float*>(smem);
  const int cq = pr;
  int n0, th0, tw0;
  wino_tile_coords(a, srx_uniform(tb * WT), n0, th0, tw0);
  const int row0 = srx_uniform(n0 * a.H + 2 * th0);              // first pixel row of the block (tiles are ordered by (image, row, column))
  const int rowrel = (tn * a.H + 2 * th) - row0;                 // >= 0 for every tile of the block
  const int oW = a.shuffle ? 2 * a.W : a.W, oC = a.shuffle ? a.shuffle : a.Cout, ps = a.shuffle ? 2 : 1;  // output row length, channels, pixel step
  const unsigned voff = 4u * (unsigned)(((ps * rowrel) * oW + 2 * ps * tw) * oC + 4 * cq);
  const unsigned pstep[4] = {0u, 4u * (unsigned)(ps * oC), 4u * (unsigned)(ps * oW * oC), 4u * (unsigned)(ps * oW * oC + ps * oC)};
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    if (j > 0) __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // accumulator rows (channels) 8 g + 4 h + 0..3 of channel tile j, column (tile) l31
        const int c = 2 * g + h;
        const f32x4 v = {acc[x][j][4 * g + 0], acc[x][j][4 * g + 1], acc[x][j][4 * g + 2], acc[x][j][4 * g + 3]};
        *reinterpret_cast<f32x4*>(sM + ((xi0 + x) * WT + l31) * 32 + ((c ^ (l31 & 7)) << 2)) = v;
      }
    __syncthreads();
    f32x4 m[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) m[x] = *reinterpret_cast<const f32x4*>(sM + (x * WT + tl) * 32 + ((cq ^ (tl & 7)) << 2));
    f32x4 s0[4], s1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s0[q] = pk_add4(pk_add4(m[0 + q], m[4 + q]), m[8 + q]);
      s1[q] = pk_sub4(pk_sub4(m[4 + q], m[8 + q]), m[12 + q]);
    }
    f32x4 y[4];
    y[0] = pk_add4(pk_add4(s0[0], s0[1]), s0[2]);
    y[1] = pk_sub4(pk_sub4(s0[1], s0[2]), s0[3]);
    y[2] = pk_add4(pk_add4(s1[0], s1[1]), s1[2]);
    y[3] = pk_sub4(pk_sub4(s1[1], s1[2]), s1[3]);
    const int cu = srx_uniform(cb * BN + 32 * j);  // first channel (GEMM column) of this pass
    if (tailw) {  // a part of a tail tile: raw partial output, tile-local layout; finished by wino_tail_fixup_kernel
      if (tvalid) {
        float* o = a.tpart + (size_t)((int)blockIdx.x - a.full) * (WT * 4 * BN) + (size_t)(tl * 4) * BN + 32 * j + 4 * cq;
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<f32x4*>(o + p * BN) = y[p];
      }
      continue;
    }
    // element offset of (first pixel row of the block, pixel column 0, this pass's first channel) in an output-shaped tensor; with a
    // PixelShuffle store GEMM column c = (sub-pixel ij, channel cc) and a 32-column pass lies inside one sub-pixel (Cout / 4 is a
    // multiple of 32): conv pixel (y, x) lands at (2y + ij / 2, 2x + ij % 2), channel cc
    const int ij = a.shuffle ? cu / a.shuffle : 0;
    const size_t ebase = a.shuffle ? ((size_t)(2 * row0 + (ij >> 1)) * oW + (ij & 1)) * oC + (cu - ij * a.shuffle)
                                   : (size_t)row0 * oW * oC + cu;
    const unsigned span = (unsigned)std::min<size_t>((a.out_elems - ebase) * sizeof(float), 0xffffffffu);
    if (a.part) {  // one of several splits of the input channels: the raw partial output; bias / activation in the fix-up pass
      const __amdgpu_buffer_rsrc_t ro = srx_rsrc(a.part + (size_t)z * a.out_elems + ebase, span);
      if (tvalid) {
#pragma unroll
        for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[p]), ro, (int)voff, (int)pstep[p], 0);
      }
      continue;
    }
    const __amdgpu_buffer_rsrc_t ro = srx_rsrc(a.out + ebase, span);
    f32x4 mk[4], ad[4];
    if (a.mask) {  // (all loads before the first store; an invalid tile's loads are harmless: its offset is in range or reads 0)
      const __amdgpu_buffer_rsrc_t rm = srx_rsrc(a.mask + ebase, span);
#pragma unroll
      for (int p = 0; p < 4; ++p) mk[p] = srx_bload(rm, voff, pstep[p]);
    }
    if (a.add) {
      const __amdgpu_buffer_rsrc_t ra = srx_rsrc(a.add + ebase, span);
#pragma unroll
      for (int p = 0; p < 4; ++p) ad[p] = srx_bload(ra, voff, pstep[p]);
    }
    if (a.bias) {
      f32x4 bv;
      if (a.shuffle) {  // (the bias is in the conv's own channel order: channel cc * 4 + ij)
        const int cc = cu - ij * a.shuffle + 4 * cq;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = a.bias[(cc + e) * 4 + ij];
      } else {
        bv = *reinterpret_cast<const f32x4*>(a.bias + cu + 4 * cq);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) y[p] = pk_add4(y[p], bv);
    }
    if (a.stats) {
      // per-channel sum / sum of squares of this tile block's 128 pixels (what gconv's epilogue writes per row tile): the lanes
      // of a wave that share a channel quad first (fixed butterfly over the 8 tiles of the wave), then the four waves in order
      f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
      if (tvalid) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { s1 += y[p]; s2 += y[p] * y[p]; }
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
      f32x4* red = reinterpret_cast<f32x4*>(smem + RING_BYTES) + j * 64;  // [4 waves][8 quads][2] per channel tile
      if (lane < 8) { red[(wave * 8 + lane) * 2 + 0] = s1; red[(wave * 8 + lane) * 2 + 1] = s2; }
    }
    // the activation forms are uniform branches (not per-element selects: every VALU instruction here is matrix time lost)
    if (a.relu) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[p][e] = fmaxf(y[p][e], 0.f);
    } else if (a.lrelu) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[p][e] = y[p][e] > 0.f ? y[p][e] : y[p][e] * a.slope;
    }
    if (a.mask) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[p][e] = mk[p][e] > 0.f ? y[p][e] : 0.f;
    }
    if (a.add) {
#pragma unroll
      for (int p = 0; p < 4; ++p) y[p] = pk_add4(y[p], ad[p]);
    }
    if (tvalid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y[p]), ro, (int)voff, (int)pstep[p], 0);
    }
  }
  if (a.stats && !tailw && !a.part) {  // (workgroup-uniform) the waves' partial sums, in order
    __syncthreads();
    if (tid < 8 * NJ) {
      const int j = tid >> 3, q = tid & 7;
      const f32x4* red = reinterpret_cast<const f32x4*>(smem + RING_BYTES) + j * 64;
      f32x4 t1 = red[q * 2], t2 = red[q * 2 + 1];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) { t1 += red[(wv * 8 + q) * 2]; t2 += red[(wv * 8 + q) * 2 + 1]; }
      float* o = a.stats + ((size_t)tb * a.Cout + cb * BN + 32 * j + 4 * q) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = t1[e]; o[2 * e + 1] = t2[e]; }
    }
  }
}

// out = act(sum_z part[z] + bias) [masked]: finishes a layer whose input channels were split over several workgroups
__global__ __launch_bounds__(256) void wino_fixup_kernel(const float* __restrict__ part, int zsplit, size_t n4, size_t stride,
                                                         const float* __restrict__ bias, const float* __restrict__ mask,
                                                         float* __restrict__ out, int cq, int relu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v[8];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // (bias and mask are requested with the first trip's partials, not behind the sum: round 6)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, mk = {0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + (i % cq) * 4);
    if (mask) mk = *reinterpret_cast<const f32x4*>(mask + i * 4);
    for (int z0 = 0; z0 < zsplit; z0 += 8) {  // eight partials per trip, loads first, added in order
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (size_t)min(z0 + u, zsplit - 1) * stride + i * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u < zsplit) s += v[u];
    }
    if (bias) s += bv;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = fmaxf(s[e], 0.f);
    }
    if (mask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = mk[e] > 0.f ? s[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = s;
  }
}

// finishes the tail tiles of a launch (WinoArgs::full / tsplit): one workgroup per tile, one thread per (2x2 tile, channel quad) as in
// wino_kernel's epilogue; the parts are added in order, then bias, ReLU / mask, stores
template <int BN>
__global__ __launch_bounds__(512) void wino_tail_fixup_kernel(const WinoArgs a) {
  constexpr int CQ = BN / 4, ITEMS = WT * CQ;
  const int tid = threadIdx.x;
  if (tid >= ITEMS) return;
  const int tile = a.full + (int)blockIdx.x;
  const int tb = tile % a.tblocks, cb = tile / a.tblocks;
  const int et = tid / CQ, cq = tid % CQ;
  const int t = tb * WT + et;
  const bool tvalid = t < a.T;
  if (!tvalid && !a.stats) return;
  const float* src = a.tpart + (size_t)blockIdx.x * a.tsplit * (WT * 4 * BN) + (size_t)(et * 4) * BN + 4 * cq;
  // the mask of the layer below and the bias do not depend on the partial sums: requested FIRST, next to the first trip's parts (a
  // workgroup of this kernel is two or three dependent round trips long and nothing else; round 6)
  const int tw = t % a.TW, r = t / a.TW, th = r % a.TH, n = r / a.TH;
  const int co = cb * BN + 4 * cq;
  const size_t p00 = (((size_t)n * a.H + 2 * th) * a.W + 2 * tw) * a.Cout + co;
  const size_t offs[4] = {p00, p00 + a.Cout, p00 + (size_t)a.W * a.Cout, p00 + (size_t)a.W * a.Cout + a.Cout};
  f32x4 mk[4];
  if (a.mask && tvalid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) mk[p] = *reinterpret_cast<const f32x4*>(a.mask + offs[p]);
  }
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + co);
  f32x4 y[4];
  f32x4 v[4][4];
  for (int z0 = 0; z0 < a.tsplit; z0 += 4) {  // four parts per trip: loads first, added in order
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int p = 0; p < 4; ++p)
        v[u][p] = *reinterpret_cast<const f32x4*>(src + (size_t)min(z0 + u, a.tsplit - 1) * (WT * 4 * BN) + p * BN);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (z0 + u < a.tsplit) {
#pragma unroll
        for (int p = 0; p < 4; ++p) y[p] = (z0 + u == 0) ? v[u][p] : y[p] + v[u][p];
      }
  }
  if (a.bias) {
#pragma unroll
    for (int p = 0; p < 4; ++p) y[p] += bv;
  }
  if (a.stats) {  // the tile block's BatchNorm partials, as wino_kernel's epilogue forms them
    __shared__ f32x4 red[(ITEMS / 64) * CQ * 2];
    const int lane = tid & 63, wave = tid >> 6;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (tvalid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) { s1 += y[p]; s2 += y[p] * y[p]; }
    }
#pragma unroll
    for (int o = CQ; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    constexpr int NW = ITEMS / 64;
    if (lane < CQ) { red[(wave * CQ + lane) * 2 + 0] = s1; red[(wave * CQ + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < CQ) {
      f32x4 t1 = red[tid * 2], t2 = red[tid * 2 + 1];
#pragma unroll
      for (int wv = 1; wv < NW; ++wv) { t1 += red[(wv * CQ + tid) * 2]; t2 += red[(wv * CQ + tid) * 2 + 1]; }
      float* o = a.stats + ((size_t)tb * a.Cout + cb * BN + 4 * tid) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[2 * e] = t1[e]; o[2 * e + 1] = t2[e]; }
    }
    if (!tvalid) return;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float w = y[p][e];
      if (a.relu) w = fmaxf(w, 0.f);
      if (a.mask) w = mk[p][e] > 0.f ? w : 0.f;
      y[p][e] = w;
    }
    *reinterpret_cast<f32x4*>(a.out + offs[p]) = y[p];
  }
}

// U = G g G^T of every (output channel, input channel) pair, laid out as wino_kernel's waves load it.
// transpose = 0: the layer itself (g = w[co][ci]); 1: its data gradient (output channels = the layer's inputs, g =
// w[ci][co] with both taps flipped).  One thread per (row channel, contraction channel) pair.
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ upk, int Cout, int Cin, int transpose) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)Cout * Cin) return;
  srx_wino_pack_one(w, upk, Cout, Cin, transpose, idx);
}

struct WinoPlan { int bn, zsplit; float cost; int tsplit; };
std::atomic<int> g_force[3] = {{0}, {0}, {0}};  // srx_wino_force_plan  // tsplit > 1: the tiles of the last round are cut that many ways (zsplit = 1)

// rounds of the chip x (chunks per workgroup x time per chunk + fixed cost), plus the fix-up pass of a split
WinoPlan wino_plan(int T, int Cin, int Cout, bool no_split = false, bool no_tail = false) {
  const int P = srx_plan_cus();
  const int nch = Cin / WKC, tblocks = (int)srx_cdiv(T, WT);
  WinoPlan best{64, 1, 1e30f};
  for (int bn = 64; bn >= 32; bn -= 32) {
    if (Cout % bn) continue;
    const int ncb = Cout / bn;
    for (int zs = 1; zs <= (no_split ? 1 : nch) && zs <= 16; ++zs) {
      const int64_t wgs = (int64_t)tblocks * ncb * zs;
      const float rounds = (float)srx_cdiv(wgs, P);
      const float chunks = (float)srx_cdiv(nch, zs);
      const float t_chunk = bn == 64 ? 4.4f : 2.5f;  // us: 128 / 64 fp32 MFMAs per SIMD pair and chunk at ~2 GHz, 32-column tiles a little over half
      const float fixed = bn == 64 ? 4.5f : 3.5f;    // prologue (first patch + U round trip) and the epilogue through LDS
      float cost = rounds * (chunks * t_chunk + fixed);
      if (zs > 1) cost += 3.0f + (float)(zs + 1) * T * 4.0f * Cout * 4.0f / 4.0e6f;  // fix-up: (zs + 1) passes over the output at ~4 TB/s
      if (cost < best.cost) best = WinoPlan{bn, zs, cost, 1};
      // whole tiles for the full rounds, the tiles of the last round cut along the input channels so that it is a short one
      const int tail = (int)(wgs % P);
      if (zs == 1 && !no_tail && !srx_dev().wino_no_tail && wgs > P && tail > 0) {
        for (int ts = 2; ts <= nch && ts <= 8; ++ts) {
          if ((int64_t)tail * ts > P) break;
          const float c = (float)(wgs / P) * (nch * t_chunk + fixed) + (float)srx_cdiv(nch, ts) * t_chunk + fixed + 4.0f;  // (+ the fix-up launch)
          if (c < best.cost) best = WinoPlan{bn, 1, c, ts};
        }
      }
    }
  }
  if (const int v = srx_dev().wino_zsplit; v > 0 && v <= nch && !no_split) { best.zsplit = v; best.tsplit = 1; }
  if (const int v = srx_dev().wino_bn; (v == 32 || v == 64) && Cout % v == 0) best.bn = v;
  // srx_wino_force_plan (measurement aid): the forced plan where this launch can run it
  if (const int fb = g_force[0].load(std::memory_order_relaxed); fb != 0) {
    const int fz = g_force[1].load(std::memory_order_relaxed), ft = g_force[2].load(std::memory_order_relaxed);
    const int64_t wgs = (int64_t)tblocks * (Cout / (Cout % fb ? 32 : fb));
    if (Cout % fb == 0 && fz >= 1 && fz <= nch && (fz == 1 || !no_split) &&
        (ft == 1 || (fz == 1 && !no_tail && ft <= nch && wgs > P && wgs % P != 0 && (wgs % P) * ft <= 2 * P)))
      best = WinoPlan{fb, fz, 0.f, ft};
  }
  return best;
}

// shuffle_ok: also layers with nn.PixelShuffle(2) in their store (the forward of the sub-pixel convs, inference)
bool wino_shape_ok(const srx_conv2d_t* d, bool shuffle_ok = false) {
  const bool sh = d->shuffle == 2 && shuffle_ok && d->Cout % 128 == 0 && d->Cout_s * 4 == d->Cout;
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && (!d->shuffle || sh) && d->up != 2 && d->precision == 0 &&
         d->Cin % WKC == 0 && d->Cout % 32 == 0 && d->Cin_s == d->Cin && (sh || d->Cout_s == d->Cout) && d->H % 2 == 0 && d->W % 2 == 0 &&
         // 32-bit byte offsets into the INPUT (the output is addressed with 64 bits); without the shuffle the layer may also run
         // as its own data gradient, whose input has Cout channels
         (int64_t)d->N * d->H * d->W * (sh ? d->Cin : std::max(d->Cin, d->Cout)) < (1LL << 30) && (int64_t)d->N * d->H * d->W < (1LL << 31);
}

}  // namespace

extern "C" int srx_wino_applicable(const srx_conv2d_t* d) {
  // (the small 64 -> 64 layers of the residual tower keep their row-tile kernel: 72 tile blocks would fill a quarter of the chip)
  if (srx_dev().no_wino || !d || !wino_shape_ok(d) || srx_rt36_applicable(d)) return 0;
  // ... and so does any layer too small to fill the chip with 32-tile blocks even without a channel split (the form the
  // BatchNorm-statistics epilogue needs): the direct kernel's K-split plans serve those better
  const double direct_us = 2.0 * d->N * d->H * d->W * (double)d->Cout * 9.0 * d->Cin / 110.0e6 + 5.0;
  return wino_plan(d->N * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, true).cost < direct_us ? 1 : 0;
}

extern "C" int srx_wino_force_plan(int bn, int zsplit, int tsplit) {
  SRX_REQUIRE((bn == 0 && zsplit == 0 && tsplit == 0) || ((bn == 32 || bn == 64) && zsplit >= 1 && zsplit <= 16 && tsplit >= 1 && tsplit <= 8),
              "wino_force_plan: (0, 0, 0) or BN 32 / 64, 1..16 splits, 1..8 tail parts");
  g_force[1].store(zsplit); g_force[2].store(tsplit); g_force[0].store(bn);
  return SRX_OK;
}

extern "C" size_t srx_wino_packed_floats(const srx_conv2d_t* d) {
  return (d && wino_shape_ok(d, true)) ? (size_t)16 * d->Cout * d->Cin : 0;
}

extern "C" int srx_wino_pack(const srx_conv2d_t* d, const float* w, float* upk, int transpose, void* stream) {
  SRX_REQUIRE(d && w && upk, "wino_pack: null pointer");
  if (!wino_shape_ok(d, !transpose)) SRX_FAIL(SRX_E_UNSUPPORTED, "wino_pack: 3x3 / stride 1 / pad 1 fp32 layers with Cin, Cout multiples of 32 and even H, W only");
  const int64_t n = (int64_t)d->Cout * d->Cin;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)srx_cdiv(n, 256)), dim3(256), 0, srx_stream(stream), w, upk, d->Cout, d->Cin,
                     transpose ? 1 : (d->shuffle ? 2 : 0));  // (a PixelShuffle layer's forward: rows in (sub-pixel, channel) order)
  SRX_CHECK_LAUNCH("wino_pack_kernel");
  return SRX_OK;
}

// workspace (floats) of a forward (which = 0) or data-gradient (which = 1) call: the partial outputs of a channel split
extern "C" size_t srx_wino_ws_floats(const srx_conv2d_t* d, int which) {
  if (!d || !wino_shape_ok(d, which != 1)) return 0;
  const int cin = which == 1 ? d->Cout : d->Cin, cout = which == 1 ? d->Cin : d->Cout;
  const int T = d->N * (d->H / 2) * (d->W / 2);
  // (which = 2: the forward with BatchNorm statistics -- whole outputs per tile, so no split of every tile; the last round's may be cut)
  const WinoPlan p = wino_plan(T, cin, cout, d->shuffle != 0 || which == 2, d->shuffle != 0);
  if (p.tsplit > 1) {
    const int64_t wgs = srx_cdiv(T, WT) * (cout / p.bn);
    return (size_t)(wgs % srx_plan_cus()) * p.tsplit * WT * 4 * p.bn;
  }
  return p.zsplit > 1 ? (size_t)p.zsplit * d->N * d->H * d->W * cout : 0;
}

// out[6] = {BN, channel splits, workgroups, tile blocks, chunks per workgroup (max), 0}
extern "C" int srx_wino_plan(const srx_conv2d_t* d, int which, int* out) {
  SRX_REQUIRE(d && out && wino_shape_ok(d), "wino_plan: not a Winograd layer");
  const int cin = which ? d->Cout : d->Cin, cout = which ? d->Cin : d->Cout;
  const int T = d->N * (d->H / 2) * (d->W / 2);
  const WinoPlan p = wino_plan(T, cin, cout);
  out[0] = p.bn; out[1] = p.zsplit; out[3] = (int)srx_cdiv(T, WT); out[2] = out[3] * (cout / p.bn) * p.zsplit;
  out[4] = (int)srx_cdiv(cin / WKC, p.zsplit); out[5] = p.tsplit;
  if (p.tsplit > 1) { const int tail = out[2] % srx_plan_cus(); out[2] += tail * (p.tsplit - 1); }
  return SRX_OK;
}

static int wino_run(const srx_conv2d_t* d, int which, const float* x, const float* upk, const float* bias, const float* mask,
                    int relu, float* y, float* ws, size_t ws_floats, void* stream, float* stats = nullptr, const float* add = nullptr,
                    int lrelu = 0, float slope = 0.f) {
  SRX_REQUIRE(d && x && upk && y, "wino: null pointer");
  if (!wino_shape_ok(d, which == 0)) SRX_FAIL(SRX_E_UNSUPPORTED, "wino: 3x3 / stride 1 / pad 1 fp32 layers with Cin, Cout multiples of 32 and even H, W only");
  SRX_REQUIRE(x != y && (!mask || mask != y), "wino: in place is not possible (neighbouring tiles read their halo)");
  SRX_REQUIRE(!d->shuffle || (!stats && !mask), "wino: PixelShuffle layers: plain forward only");
  WinoArgs a{};
  a.in = x; a.upk = upk; a.bias = bias; a.mask = mask; a.out = y;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = which ? d->Cout : d->Cin;
  a.Cout = which ? d->Cin : d->Cout;
  a.TH = d->H / 2; a.TW = d->W / 2; a.T = d->N * a.TH * a.TW;
  a.tblocks = (int)srx_cdiv(a.T, WT);
  a.nch = a.Cin / WKC;
  a.relu = relu;
  a.inv_TW = 1.0f / (float)a.TW; a.inv_TH = 1.0f / (float)a.TH;
  a.in_bytes = (unsigned)((size_t)d->N * d->H * d->W * a.Cin * sizeof(float));
  a.upk_bytes = (unsigned)((size_t)16 * a.Cin * a.Cout * sizeof(float));
  a.out_elems = (size_t)d->N * d->H * d->W * a.Cout;
  SRX_REQUIRE(!add || (add != y && add != x), "wino: the addend must be a tensor of its own");
  a.add = add; a.lrelu = lrelu; a.slope = slope;
  // (statistics come from whole outputs, and the fix-up pass knows neither an addend nor a LeakyReLU: no channel split then)
  a.shuffle = d->shuffle ? d->Cout / 4 : 0;
  const bool plain = add == nullptr && lrelu == 0 && d->shuffle == 0;  // (the fix-up passes know bias, ReLU, mask and statistics)
  const WinoPlan p = wino_plan(a.T, a.Cin, a.Cout, stats != nullptr || !plain, !plain);
  a.zsplit = p.zsplit;
  a.ncb = a.Cout / p.bn;
  a.stats = stats;
  if (p.zsplit > 1) {
    if (!ws || ws_floats < (size_t)p.zsplit * a.out_elems) SRX_FAIL(SRX_E_WORKSPACE, "wino: workspace %zu < %zu floats", ws_floats, (size_t)p.zsplit * a.out_elems);
    a.part = ws;
  }
  hipStream_t st = srx_stream(stream);
  int64_t items = (int64_t)a.tblocks * a.ncb * a.zsplit;
  a.full = (int)items; a.tsplit = 1;
  int tail = 0;
  if (p.tsplit > 1) tail = (int)(items % srx_plan_cus());
  if (tail > 0) {  // (0: a developer override of BN left no partly filled round)
    const size_t need = (size_t)tail * p.tsplit * WT * 4 * p.bn;
    if (!ws || ws_floats < need) SRX_FAIL(SRX_E_WORKSPACE, "wino: workspace %zu < %zu floats", ws_floats, need);
    a.full = (int)items - tail; a.tsplit = p.tsplit; a.tpart = ws;
    items = a.full + (int64_t)tail * p.tsplit;
  }
  const dim3 grid((unsigned)items);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS);
  });
  char nm[112];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "wino_kernel<%d> MxNxK=%dx%dx%d", p.bn, a.N * a.H * a.W, a.Cout, 9 * a.Cin);
  const double fl = 2.0 * a.N * a.H * a.W * (double)a.Cout * 9.0 * a.Cin;  // algorithmic FLOPs of the convolution (the direct form's)
  if (p.bn == 64) SRX_LAUNCH_PROF(nm, fl, wino_kernel<64>, grid, dim3(WINO_THREADS), WINO_LDS, st, a);
  else SRX_LAUNCH_PROF(nm, fl, wino_kernel<32>, grid, dim3(WINO_THREADS), WINO_LDS, st, a);
  SRX_CHECK_LAUNCH("wino_kernel");
  if (p.zsplit > 1) {
    const size_t n4 = a.out_elems / 4;
    const unsigned blocks = (unsigned)std::min<size_t>(srx_cdiv((int64_t)n4, 256), 4096);
    hipLaunchKernelGGL(wino_fixup_kernel, dim3(blocks), dim3(256), 0, st, ws, p.zsplit, n4, a.out_elems, bias, mask, y, a.Cout / 4, relu);
    SRX_CHECK_LAUNCH("wino_fixup_kernel");
  }
  if (tail > 0) {
    if (p.bn == 64) hipLaunchKernelGGL(wino_tail_fixup_kernel<64>, dim3((unsigned)tail), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(wino_tail_fixup_kernel<32>, dim3((unsigned)tail), dim3(256), 0, st, a);
    SRX_CHECK_LAUNCH("wino_tail_fixup_kernel");
  }
  return SRX_OK;
}

extern "C" int srx_wino_fwd(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* ws,
                            size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && (d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU), "wino_fwd: no activation or ReLU");
  return wino_run(d, 0, x, upk, bias, nullptr, d->act == SRX_ACT_RELU, y, ws, ws_floats, stream);
}

// inference forms (functional.FoldedConv: conv with the eval-mode BatchNorm folded in, srgan/residual.py:86-91): any of no
// activation / ReLU / LeakyReLU(d->slope) (a single-parameter PReLU is that), then `+ residual` (may be NULL)
extern "C" int srx_wino_fwd_act(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, const float* residual,
                                float* y, float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && (d->act == SRX_ACT_NONE || d->act == SRX_ACT_RELU || d->act == SRX_ACT_LRELU), "wino_fwd_act: no activation, ReLU or LeakyReLU");
  return wino_run(d, 0, x, upk, bias, nullptr, d->act == SRX_ACT_RELU, y, ws, ws_floats, stream, nullptr, residual,
                  d->act == SRX_ACT_LRELU, d->slope);
}
// Is the Winograd form of this layer's forward worth it at inference?  As srx_wino_applicable, for layers the training path
// leaves to the 36-pixel row tile as well (3x3 64 -> 64 on a frame: thousands of tile blocks)
extern "C" int srx_wino_infer_applicable(const srx_conv2d_t* d) {
  if (srx_dev().no_wino || !d || !wino_shape_ok(d, true)) return 0;
  const double direct_us = 2.0 * d->N * d->H * d->W * (double)d->Cout * 9.0 * d->Cin / 110.0e6 + 5.0;
  return wino_plan(d->N * (d->H / 2) * (d->W / 2), d->Cin, d->Cout, true).cost < direct_us ? 1 : 0;
}

// forward of a layer followed by a training-mode BatchNorm2d (srgan/discriminator.py:35-61: conv without bias, BatchNorm,
// LeakyReLU): stats [srx_wino_stat_rows(d)][Cout][2] receives per tile block (128 output pixels: 32 tiles of 2x2) the
// per-channel sum and sum of squares of the output, as srx_conv2d_fwd's bn_partials does per row tile
extern "C" int srx_wino_stat_rows(const srx_conv2d_t* d) {
  return (d && wino_shape_ok(d)) ? (int)srx_cdiv((int64_t)d->N * (d->H / 2) * (d->W / 2), WT) : 0;
}
extern "C" int srx_wino_fwd_stats(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* stats,
                                  float* ws, size_t ws_floats, void* stream) {
  SRX_REQUIRE(d && d->act == SRX_ACT_NONE && stats, "wino_fwd_stats: a linear layer and a statistics table");
  return wino_run(d, 0, x, upk, bias, nullptr, 0, y, ws, ws_floats, stream, stats);
}

extern "C" int srx_wino_bwd_data(const srx_conv2d_t* d, const float* dy, const float* upk_t, const float* relu_out, float* dx,
                                 float* ws, size_t ws_floats, void* stream) {
  return wino_run(d, 1, dy, upk_t, nullptr, relu_out, 0, dx, ws, ws_floats, stream);
}
