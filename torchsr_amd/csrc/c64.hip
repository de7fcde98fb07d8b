// 3x3 / stride 1 / pad 1 convolution of a 64-channel bf16 NHWC tensor, bf16 out (round 4): the bf16-NATIVE path.
//
// gconv_kernel<.., PR = 1> multiplies bf16 but lives on fp32: it reads fp32 activations from HBM, rounds them in the VALU
// on their way into LDS and re-reads both operands from LDS for every MFMA -- 264 TFLOP/s on the 64 -> 64 trunk convs of
// the SRGAN generator at 1080p, three times their HBM floor.  Here activations are STORED as bf16
// (128 bytes per pixel, half the HBM traffic), arrive in LDS without a conversion, and the weights never touch LDS:
//
//   * K = 9 taps x 64 channels = 36 MFMA k-steps of 16.  A wave keeps the weight matrix of 32 output channels in
//     registers -- 36 k-steps x 4 VGPRs = 144 -- loaded once per workgroup and reused for every pixel the (persistent)
//     workgroup ever computes.  (All 64 channels per wave, 288 registers, was the first form: past 256 the compiler parks
//     the surplus in AGPRs and copies four back in front of every MFMA.)
//   * The MFMA is v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (rows = output channels) and 32 consecutive
//     pixels of an image row as the B operand (columns = pixels): a tap is an address offset into a rolling window of
//     image rows in LDS, a B fragment is ONE ds_read_b128 (8 channels of one pixel), and a wave multiplies its weights
//     with TWO pixel segments per k-step: one LDS read per MFMA, half the LDS bandwidth, where the generic tile saturates
//     it.  Waves 2p and 2p + 1 own the two channel halves of pixel group p (64 pixels).
//   * The window: rows of (32 CW + 2) pixels x 128 bytes, 16-byte chunk c of pixel p stored at chunk c ^ ((p >> 1) & 7):
//     16 consecutive pixels then sit on 16 different 4-bank groups whatever the tap shift (conflict-free b128 reads).
//     A workgroup (4 waves = RW rows x CW column segments of 32 pixels) walks DOWN a column strip: per step it requests RW
//     new rows by LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS with no register in between; the swizzle sits on the
//     per-lane SOURCE address, the destination is lane-linear; out-of-image pixels fail the buffer descriptor's range check
//     and land as zeros -- probed, tools/probe/lds_dma_oob.hip), one barrier, and computes 128 pixels x 64 channels: 72 MFMAs
//     per wave and step.  The DMA is inline asm and waited for by a counted s_waitcnt that leaves the step's four output
//     stores in flight: with compiler-tracked loads the waits at the loop head drained those stores every step
//     (3.4 us per step where the MFMAs need 1.1).
//   * Epilogue: the accumulator (channel-major per lane) is turned pixel-major through a wave-private LDS area, then
//     bias, activation (v > 0 ? v : v * slope), an optional bf16 addend (the residual block's skip input), rounding to
//     bf16 and ONE 16-byte store per lane and 8 channels: every store instruction writes whole 128-byte pixels.
//     PixelShuffle(2) (srgan/residual.py:27-28) is an output address: the 256 output channels are packed as four groups
//     of 64, group (i, j) = the channels that land on sub-pixel (i, j), each group a pass of its own (blockIdx.y).
//   * Addressing is 64-bit per (image, row): tensors above 4 GiB / 2^24 pixels (the 8K feature map: 4.2 GB) need no tiling.
//
// Serves (eval mode, torchsr/test.py:57-62 -- the `torchsr test` path): the 32 + 1 trunk convs of the SRGAN generator
// with BatchNorm folded into the weights (srgan/residual.py:86-91, srgan/generator.py:76-78) and both sub-pixel layers.
#include "srx_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KSTEPS = 36;       // 9 taps x 4 channel groups of 16

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0)):
// here that queue holds the epilogue's stores of the step before -- a full HBM write round trip exposed once per step, with
// one wave per SIMD and nothing to switch to -- and the next rows' loads, which have a whole step to arrive and are waited
// for where they are consumed.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One LDS-DMA instruction: every active lane moves 16 bytes from buffer offset `voff` (out of range: zeros) to LDS byte
// lds_base + 16 * lane.  M0 (the destination base) is written and restored inside the statement.
__device__ __forceinline__ void dma16(const u32x4& rsrc, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_base), "s"(rsrc) : "memory");
}
constexpr int EPI_PITCH = 144;   // bytes per pixel of the epilogue's fp32 transpose area (32 floats + 16 bytes)

struct C64Args {
  const unsigned char* x;    // bf16 NHWC [N][H][W][64]
  const unsigned char* w;    // packed: [group][kstep 36][half 2][lane 64][8 bf16]
  const float* bias;         // [groups * 64] in packed (group-major) order, or null
  const unsigned char* res;  // bf16, laid out like out; null: none (never with shuffle)
  unsigned char* out;        // bf16 [N][Ho][Wo][out_cs]
  int N, H, W;
  int shuffle;               // 0, or 2: group g writes sub-pixel (g >> 1, g & 1) of a [N][2H][2W][out_cs] tensor
  int out_cs;                // channel stride of the output (and the addend) in bf16 elements, >= 64
  float slope;               // v > 0 ? v : v * slope (none: 1, ReLU: 0)
  int strips, chunks, rows_per_chunk, nwork;
  unsigned* dbg;  // developer aid (srx_conv3x3_c64_bf16_fwd_dbg): per workgroup and wave, cycles spent per phase; null in the product
  int ablate;  // developer aid (SRX_C64_ABLATE, timing only, results are wrong): 1 skip the finishing, 2 skip the MFMAs, 4 skip the window requests
};

// RW x CW: rows x 32-pixel column segments a step covers (RW * CW = 4 segment pairs... 128 pixels).  EIGHT waves:
//   waves 0-3, one per SIMD, do nothing but fragment reads and MFMAs (72 per step) and leave their accumulators in LDS;
//   waves 4-7 (wave w + 4 shares wave w's SIMD) request the next rows by LDS-DMA and finish the tile of the step BEFORE --
//   bias, activation, addend, rounding, stores: ~130 vector instructions per 8 pixels x 8 channels that a single wave per
//   SIMD had to run between its MFMAs (1.5 of 3.2 us per step), and that now issue in the shadow of the other wave's MFMAs.
// One barrier per step orders both hand-overs (window rows: helper -> matrix waves; accumulators: matrix -> helper).
template <int RW, int CW>
__global__ __launch_bounds__(512) void c64_bf16_kernel(const C64Args a) {
  static_assert(RW * CW == 4, "128 pixels per step");
  constexpr int TW = 32 * CW;                    // tile columns
  constexpr int PX = TW + 2;                     // window pixels per row (one halo pixel each side)
  // Rows arrive in groups of RW.  A step reads group k and the first two rows of group k + 1 while group k + 2 is being
  // written (RW = 1: reads groups k .. k + 2, writes k + 3): a ring of 3 RW (4) row slots.
  constexpr int NR = RW == 1 ? 4 : 3 * RW;
  constexpr int AHEAD = RW == 1 ? 3 : 2;         // the group written during step k is group k + AHEAD
  constexpr int ROWB = PX * 128;                 // bytes per window row
  constexpr int UNITS = RW * PX * 8;             // 16-byte units loaded per step
  constexpr int NLD = (UNITS + 255) / 256;       // ... per helper thread
  constexpr int ACCB = 64 * EPI_PITCH;           // one wave's accumulators, pixel-major fp32
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = srx_uniform((int)threadIdx.x >> 6);
  const bool helper = wave >= 4;                 // wave-uniform
  const int w4 = wave & 3;
  const int nh = w4 & 1, grp = w4 >> 1;          // channel half, pixel group of the tile this wave computes / finishes
  // the tile's two 32-pixel segments: (row, column) offsets inside the step's RW x TW pixels
  const int seg_r0 = RW == 1 ? 0 : (RW == 2 ? grp : 2 * grp), seg_r1 = RW == 4 ? seg_r0 + 1 : seg_r0;
  const int seg_c0 = RW == 1 ? 64 * grp : 0, seg_c1 = RW == 4 ? 0 : seg_c0 + 32;
  constexpr unsigned SEG1_OFF = RW == 4 ? 0u : 32u * 128u;  // segment 1's window bytes relative to segment 0's (same swizzle: 32 pixels on)
  const int l31 = lane & 31, h = lane >> 5;
  unsigned char* win = smem;
  unsigned char* accbuf = smem + NR * ROWB + w4 * (2 * ACCB);  // [2][64 pixels][EPI_PITCH]: double-buffered per tile owner
  unsigned char* resbuf = smem + NR * ROWB + 4 * (2 * ACCB) + w4 * 4096;  // the addend's 64 pixels x 64 bytes of a helper's tile
  const int g = blockIdx.y;

  const unsigned win_lds = (unsigned)(size_t)win;               // LDS byte address of the ring
  const size_t in_row_bytes = (size_t)a.W * 128;

  // The two roles run SEPARATE loop nests over the same work items and steps (the register allocator then keeps the matrix
  // waves' 144 weight registers and the helpers' addressing state apart); both execute exactly the same sequence of barriers:
  //   per work item:  barrier A  |  per step k = 0 .. nsteps:  barrier B_k
  // Image row r lives in ring slot (r - (r_beg - 1)) % NR; group j = rows r_beg - 1 + j RW .. + RW - 1.
  // Step k: the matrix waves compute tile k, the helpers request group k + AHEAD and finish tile k - 1; trip k = nsteps
  // finishes the last tile.  B_k: the rows requested during step k - 1 have landed (the helpers waited for them), tile
  // k - 1's accumulators are in LDS, and everyone is done reading what step k overwrites.
  if (!helper) {
    __builtin_amdgcn_s_setprio(3);  // the matrix pipe is what the step waits for: its wave wins every issue arbitration
    // ---- matrix waves: the weights of 32 output channels, 36 fragments of 8 bf16 per lane, resident for the life of the workgroup
    bf16x8 wf[KSTEPS];
    {
      const u32x4* wp = reinterpret_cast<const u32x4*>(a.w + (size_t)g * (KSTEPS * 2 * 1024)) + nh * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, wp[ks * 128]);
    }
    // the bias in accumulator layout (register r of lane l holds channel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the 32): it is the
    // C operand of each tile's first MFMA, so adding it costs no instruction anywhere
    f32x16 biasv;
#pragma unroll
    for (int r = 0; r < 16; ++r) biasv[r] = a.bias ? a.bias[g * 64 + 32 * nh + (r & 3) + 8 * (r >> 2) + 4 * h] : 0.f;
    // per-lane window offsets of segment 0's B fragments: pixel seg_c0 + l31 + tw, chunk (2 cs + h) ^ swizzle(pixel)
    unsigned foff[3][4];
#pragma unroll
    for (int tw = 0; tw < 3; ++tw) {
      const int p0 = seg_c0 + l31 + tw;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) foff[tw][cs] = (unsigned)(p0 * 128 + (((2 * cs + h) ^ ((p0 >> 1) & 7)) * 16));
    }
    unsigned ph[4] = {0u, 0u, 0u, 0u};  // (dbg) cycles: barrier wait, MFMA loop, accumulator dump, steps
    const unsigned t_begin = (unsigned)__builtin_amdgcn_s_memtime();

    // One step.  S0 >= 0: the ring slot of row R - 1 is a compile-time constant (RW = 1: the step loop is unrolled over the
    // ring), so every fragment address is a per-lane register plus an IMMEDIATE -- no address arithmetic between the MFMAs,
    // whose issue slots the helper wave on this SIMD needs.  S0 < 0: the slot is the run-time value s0r.
    auto step = [&](int k, int nsteps, int r_beg, int r_end, int s0r, auto s0_c) {
      constexpr int S0 = decltype(s0_c)::value;
      unsigned t0 = a.dbg ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
      lds_barrier();  // B_k
      if (a.dbg) { const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); ph[0] += t1 - t0; t0 = t1; ph[3] += 1; }
      if (!(k < nsteps && r_beg + k * RW + seg_r0 < r_end) || (a.ablate & 2)) return;
      unsigned char* dst = accbuf + (k & 1) * ACCB;
      f32x16 acc0, acc1;
      // ring-slot byte offsets of the three taps' rows for the two segments
      unsigned sb0[3], sb1[3];
#pragma unroll
      for (int th = 0; th < 3; ++th) {
        if constexpr (S0 >= 0) {
          sb0[th] = (unsigned)(((S0 + th) % NR) * ROWB);  // (RW = 1: both segments sit in the step's one row)
          sb1[th] = sb0[th] + SEG1_OFF;
        } else {
          int x0 = s0r + seg_r0 + th, x1 = s0r + seg_r1 + th;
          x0 = x0 >= NR ? x0 - NR : x0; x1 = x1 >= NR ? x1 - NR : x1;
          sb0[th] = (unsigned)srx_uniform(x0 * ROWB); sb1[th] = (unsigned)srx_uniform(x1 * ROWB) + SEG1_OFF;
        }
      }
      // B fragments run PD k-steps ahead of the MFMAs that consume them (a read issued right in front of its MFMA
      // exposes the whole LDS latency every second instruction)
      constexpr int PD = 4;
      bf16x8 b0[PD], b1[PD];
      auto fetch = [&](int ks, int slot) {
        const int tap = ks >> 2, th = tap / 3, tw = tap - 3 * th, cs = ks & 3;
        b0[slot] = *reinterpret_cast<const bf16x8*>(win + foff[tw][cs] + sb0[th]);
        b1[slot] = *reinterpret_cast<const bf16x8*>(win + foff[tw][cs] + sb1[th]);
      };
#pragma unroll
      for (int ks = 0; ks < PD; ++ks) fetch(ks, ks);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const bf16x8 x0 = b0[ks % PD], x1 = b1[ks % PD];
        if (ks + PD < KSTEPS) fetch(ks + PD, ks % PD);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], x0, ks == 0 ? biasv : acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], x1, ks == 0 ? biasv : acc1, 0, 0, 0);
      }
      // pin the issue order the loop above spells out: 2 PD reads up front, then per k-step the two reads of step
      // ks + PD in front of the two MFMAs of step ks (left alone the scheduler sinks every read to just before its MFMA)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + PD < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      if (a.dbg) {
        asm volatile("" :: "v"(acc0), "v"(acc1));
        const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); ph[1] += t1 - t0; t0 = t1;
      }
      // D[row = channel (r & 3) + 8 (r >> 2) + 4 h][col = pixel l31] -> pixel-major fp32 for the helper wave
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f32x4*>(dst + l31 * EPI_PITCH + (8 * q + 4 * h) * 4) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
        *reinterpret_cast<f32x4*>(dst + (32 + l31) * EPI_PITCH + (8 * q + 4 * h) * 4) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
      }
      if (a.dbg) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ph[2] += (unsigned)__builtin_amdgcn_s_memtime() - t0;
      }
    };

    for (int wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
      const int chunk = wi % a.chunks;
      const int r_beg = chunk * a.rows_per_chunk, r_end = min(a.H, r_beg + a.rows_per_chunk);
      const int nsteps = (r_end - r_beg + RW - 1) / RW;
      lds_barrier();  // A
      if constexpr (RW == 1) {  // ring of 4 rows, one row per step: steps k, k + 1, k + 2, k + 3 have slots 0, 1, 2, 3
        for (int k = 0; k <= nsteps; k += 4) {
          step(k, nsteps, r_beg, r_end, 0, std::integral_constant<int, 0>{});
          if (k + 1 <= nsteps) step(k + 1, nsteps, r_beg, r_end, 0, std::integral_constant<int, 1>{});
          if (k + 2 <= nsteps) step(k + 2, nsteps, r_beg, r_end, 0, std::integral_constant<int, 2>{});
          if (k + 3 <= nsteps) step(k + 3, nsteps, r_beg, r_end, 0, std::integral_constant<int, 3>{});
        }
      } else {
        int s0 = 0;  // ring slot of row R - 1
        for (int k = 0; k <= nsteps; ++k) {
          step(k, nsteps, r_beg, r_end, s0, std::integral_constant<int, -1>{});
          s0 += RW;
          s0 = s0 >= NR ? s0 - NR : s0;
        }
      }
    }
    if (a.dbg && lane == 0) {
      unsigned* d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
      d[0] = ph[0]; d[1] = ph[1]; d[2] = ph[2]; d[3] = ph[3]; d[4] = (unsigned)__builtin_amdgcn_s_memtime() - t_begin;
    }
    return;
  }

  // ---- helper waves: this lane finishes channels 32 nh + 8 (lane & 3) .. + 7 of pixels (lane >> 2) + 16 t of the tile's 64
  const int ech = lane & 3, epx = lane >> 2;
  // activation: none (slope 1), v > 0 ? v : v * slope in general, max(v, v * slope) when 0 <= slope <= 1 (ReLU, LeakyReLU, a
  // PReLU parameter in its usual range): two instructions per pair of elements instead of five
  const int act_mode = a.slope == 1.f ? 0 : ((a.slope >= 0.f && a.slope <= 1.f) ? 1 : 2);  // (uniform)
  // staging units of the four helper waves.  Unit e = 256 u + tid of a group of RW rows is 16-byte position (e % 8) of window
  // pixel (e / 8) % PX of row e / (8 PX): LDS byte 16 e of the group's slots (rows are contiguous), i.e. lane-linear per wave.
  int srow[NLD], scol[NLD];
  unsigned ssrc[NLD];
  bool sok[NLD];
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int e = u * 256 + tid;
    const int rr = e / (PX * 8), rem = e - rr * (PX * 8);
    const int px = rem >> 3, pos = rem & 7;
    sok[u] = e < UNITS;
    srow[u] = rr;
    scol[u] = px - 1;                                           // column relative to the strip's first column
    ssrc[u] = (unsigned)((pos ^ ((px >> 1) & 7)) * 16);         // source chunk (the swizzle is an involution on chunks)
  }
  const int Ho = a.shuffle ? 2 * a.H : a.H, Wo = a.shuffle ? 2 * a.W : a.W;
  const int sh = a.shuffle ? 2 : 1, si = a.shuffle ? (g >> 1) : 0, sj = a.shuffle ? (g & 1) : 0;
  const int gcol = a.shuffle ? 0 : 64 * g;  // without PixelShuffle the groups are channel ranges of one pixel
  const size_t out_px_bytes = (size_t)a.out_cs * 2;
  const unsigned out_row_bytes = (unsigned)((size_t)Wo * out_px_bytes);

  unsigned hp[5] = {0u, 0u, 0u, 0u, 0u};  // (dbg) cycles: barrier wait, requests, addend wait, finishing, final wait
  const unsigned ht_begin = (unsigned)__builtin_amdgcn_s_memtime();
  for (int wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
    int t = wi;
    const int chunk = t % a.chunks; t /= a.chunks;
    const int strip = t % a.strips;
    const int n = t / a.strips;
    const int c0 = strip * TW;
    const int r_beg = chunk * a.rows_per_chunk, r_end = min(a.H, r_beg + a.rows_per_chunk);
    // descriptor over the rows this chunk can touch: [r_beg - 1, r_end + 1) clipped to the image
    const int rb = max(r_beg - 1, 0), re = min(r_end + 1, a.H);
    u32x4 rx;
    {
      const unsigned long long xb = (unsigned long long)(a.x + ((size_t)n * a.H + rb) * in_row_bytes);
      rx[0] = (unsigned)srx_uniform((int)(unsigned)xb);
      rx[1] = (unsigned)srx_uniform((int)((unsigned)(xb >> 32) & 0xffffu));
      rx[2] = (unsigned)srx_uniform((int)(unsigned)((size_t)(re - rb) * in_row_bytes));
      rx[3] = 0x00020000u;
    }
    // Request the NEXT group of RW image rows (groups are requested in order, starting at row r_beg - 1) into ring slots
    // slot0 .. (groups never wrap: NR is a multiple of RW).  goff[u] = this unit's byte offset from the descriptor's first
    // row; it advances by RW rows per request.  Columns outside the image are pointed out of range; rows above the image
    // give a wrapped (huge) offset and rows below it one past the descriptor's range: all arrive as zeros, the conv's padding.
    unsigned goff[NLD];
    bool colok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int col = c0 + scol[u];
      colok[u] = sok[u] && (unsigned)col < (unsigned)a.W;
      goff[u] = (unsigned)(r_beg - 1 + srow[u] - rb) * (unsigned)in_row_bytes + (unsigned)col * 128u + ssrc[u];
    }
    const unsigned gstep = (unsigned)srx_uniform((int)((unsigned)RW * (unsigned)in_row_bytes));
    auto dma_group = [&](int slot0) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) {
        const unsigned dst = (unsigned)srx_uniform((int)(win_lds + (unsigned)(slot0 * ROWB) + (unsigned)((u * 256 + w4 * 64) * 16)));
        if (sok[u]) dma16(rx, colok[u] ? goff[u] : 0xffffffffu, dst);
        goff[u] += gstep;
      }
    };
    // Finishing the tile of step R is split around the window request so that NO vector-memory load of this wave has a
    // register destination: the counter that orders loads (vmcnt) retires in issue order, so a compiler-tracked load of the
    // addend issued behind the window's DMA waited for the DMA's whole latency, and one issued in front of it made the
    // compiler wait for "everything" -- either way the helper, not the matrix pipe, set the step time (3.2 us; PMC: MFMA 42 %
    // busy).  The addend therefore arrives by LDS-DMA too (each lane's 16 bytes land at its own 16 bytes of `resbuf`), and
    // the waits are counted by hand.  Per step k (tile k - 1 is being finished, tile k's addend is requested for the step after):
    //   [window DMA x NLD]  vmcnt(NLD): the addend of tile k - 1, requested early in step k - 1, has landed (it had a whole step)
    //   [resbuf -> registers]  [addend DMA of tile k x4 -> resbuf]  [finish tile k - 1: stores x4]
    //   vmcnt(8): the window rows have landed, addend and stores fly on
    const unsigned res_lds = (unsigned)(size_t)resbuf;
    const bool has_res = a.res != nullptr;
    // Per-item addressing of the tile this wave finishes: the per-lane byte offset of each of its four tasks inside an
    // output row is fixed for the item (a column outside the image: out of range); the row's base pointer is kept per
    // segment and advanced by a constant per step -- no multiplication on the way to a store or an addend request.
    unsigned toff[4];
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
      const int col = c0 + ((t4 >> 1) ? seg_c1 : seg_c0) + epx + 16 * (t4 & 1);
      toff[t4] = col < a.W ? (unsigned)((size_t)(sh * col + sj) * out_px_bytes) + (unsigned)(gcol * 2 + 64 * nh + 16 * ech) : 0xffffffffu;
    }
    const size_t out_row_stride = (size_t)Wo * out_px_bytes;
    // base pointers of the output (addend) rows of the segments of the tile of step R = r_beg: rows sh * (R + seg_r) + si
    const unsigned long long ob0 = (unsigned long long)(a.out + (((size_t)n * Ho + (size_t)(sh * (r_beg + seg_r0) + si)) * out_row_stride));
    const unsigned long long ob1 = (unsigned long long)(a.out + (((size_t)n * Ho + (size_t)(sh * (r_beg + seg_r1) + si)) * out_row_stride));
    const unsigned long long rb0 = (unsigned long long)((a.res ? a.res : a.out) + (((size_t)n * Ho + (size_t)(r_beg + seg_r0)) * out_row_stride));
    const unsigned long long rb1 = (unsigned long long)((a.res ? a.res : a.out) + (((size_t)n * Ho + (size_t)(r_beg + seg_r1)) * out_row_stride));
    const unsigned long long ostep = (unsigned long long)(sh * RW) * out_row_stride, rstep = (unsigned long long)RW * out_row_stride;
    auto row_rsrc = [&](unsigned long long p, bool rowok) {
      u32x4 r;
      r[0] = (unsigned)srx_uniform((int)(unsigned)p);
      r[1] = (unsigned)srx_uniform((int)((unsigned)(p >> 32) & 0xffffu));
      r[2] = rowok ? out_row_bytes : 0u;
      r[3] = 0x00020000u;
      return r;
    };
    // the addend of the tile of step k (rows r_beg + k RW + seg_r) -> resbuf
    auto request_addend = [&](int k) {
      const int R = r_beg + k * RW;
      const u32x4 r0 = row_rsrc(rb0 + (unsigned long long)k * rstep, R + seg_r0 < r_end);
      const u32x4 r1 = row_rsrc(rb1 + (unsigned long long)k * rstep, R + seg_r1 < r_end);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)
        dma16((t4 >> 1) ? r1 : r0, toff[t4], (unsigned)srx_uniform((int)(res_lds + (unsigned)(t4 * 1024))));
    };
    auto finish = [&](int k, const unsigned char* src, const u32x4 (&rreg)[4]) {  // the tile of step k
      const int R = r_beg + k * RW;
      const u32x4 o0 = row_rsrc(ob0 + (unsigned long long)k * ostep, R + seg_r0 < r_end);
      const u32x4 o1 = row_rsrc(ob1 + (unsigned long long)k * ostep, R + seg_r1 < r_end);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const unsigned char* sp = src + (epx + 16 * t4) * EPI_PITCH + ech * 32;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 16);
        const bf16x8 radd = __builtin_bit_cast(bf16x8, rreg[t4]);
        float v[8];  // (the bias is already in: the matrix waves start their accumulators from it)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < 4 ? lo[e] : hi[e - 4];
        if (act_mode == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
        } else if (act_mode == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        }
        if (has_res) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)radd[e];
        }
        const bf16x8 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
        const u32x4 od = __builtin_bit_cast(u32x4, o);
        // (an asm store: the descriptor is built by hand as four scalars, like the DMA's)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(od), "v"(toff[t4]), "s"((t4 >> 1) ? o1 : o0) : "memory");
      }
    };

    lds_barrier();  // A: the previous work item's window is dead (its last accumulators were finished before this barrier)
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) dma_group(j * RW);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int nsteps = (r_end - r_beg + RW - 1) / RW;
    int s0 = 0;                          // ring slot of row R - 1
    for (int k = 0; k <= nsteps; ++k) {
      unsigned t0 = a.dbg ? (unsigned)__builtin_amdgcn_s_memtime() : 0u;
      lds_barrier();  // B_k
      if (a.dbg) { const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); hp[0] += t1 - t0; t0 = t1; }
      const int R = r_beg + k * RW;
      const bool fin = k > 0 && R - RW + seg_r0 < r_end && !(a.ablate & 1);  // (wave-uniform) a tile of step k - 1 to finish
      const bool nxt = has_res && k < nsteps && R + seg_r0 < r_end && !(a.ablate & 1);  // tile k exists: its addend is requested below
      if (k < nsteps && !(a.ablate & 4)) {
        int wslot = s0 + AHEAD * RW;
        wslot = wslot >= NR ? wslot - NR : wslot;
        dma_group(wslot);
        // the addend has landed; the window request flies on.  (A wave issues NLD window instructions, or NLD - 1 when the
        // last one has no lane of this wave inside the group: skipped whole, it must not be counted.)
        if (a.dbg) { const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); hp[1] += t1 - t0; t0 = t1; }
        // (Without an addend there is nothing to wait for here -- and the wait would also drain the stores of the step
        // before, which are older than everything issued in this one.)
        if (fin && has_res) {
          if (UNITS - w4 * 64 > (NLD - 1) * 256) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD - 1) : "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (a.dbg) { const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); hp[2] += t1 - t0; t0 = t1; }
      u32x4 rreg[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4)  // the addend of tile k - 1 leaves resbuf before tile k's is requested into it
        rreg[t4] = (fin && has_res) ? *reinterpret_cast<const u32x4*>(resbuf + t4 * 1024 + lane * 16) : u32x4{0u, 0u, 0u, 0u};
      if (nxt) {
        asm volatile("s_waitcnt lgkmcnt(0)" :: "v"(rreg[0]), "v"(rreg[1]), "v"(rreg[2]), "v"(rreg[3]) : "memory");
        request_addend(k);
      }
      if (fin) finish(k - 1, accbuf + ((k - 1) & 1) * ACCB, rreg);
      if (a.dbg) { const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime(); hp[3] += t1 - t0; t0 = t1; }
      // the window rows must have landed before the next barrier; what was issued behind them may fly on
      if (fin && nxt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (fin || nxt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (a.dbg) hp[4] += (unsigned)__builtin_amdgcn_s_memtime() - t0;
      s0 += RW;
      s0 = s0 >= NR ? s0 - NR : s0;
    }
  }
  if (a.dbg && lane == 0) {
    unsigned* d = a.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
    d[0] = hp[0]; d[1] = hp[1]; d[2] = hp[2]; d[3] = hp[3]; d[4] = hp[4]; d[5] = (unsigned)__builtin_amdgcn_s_memtime() - ht_begin;
  }
}

// OIHW fp32 [Cout][64][3][3] -> [group][kstep][half][lane][8] bf16: lane l of fragment (ks, nh) holds output channel
// 32 nh + (l & 31) of its group, input channels 16 (ks & 3) + 8 (l >> 5) .. + 7 of tap ks >> 2.  With PixelShuffle the group
// g = 2 i + j collects the channels c * 4 + g (c = 0 .. 63), the ones PixelShuffle(2) moves to sub-pixel (i, j).
__global__ void c64_pack_kernel(const float* __restrict__ w, const float* __restrict__ scale, unsigned short* __restrict__ dst,
                                int Cout, int shuffle) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 8-channel fragment lane
  const int total = (Cout / 64) * KSTEPS * 2 * 64;
  if (idx >= total) return;
  const int lane = idx & 63, nh = (idx >> 6) & 1, ks = (idx >> 7) % KSTEPS, g = idx / (KSTEPS * 128);
  const int row = 32 * nh + (lane & 31);
  const int co = shuffle ? row * 4 + g : 64 * g + row;
  const int tap = ks >> 2, ci0 = 16 * (ks & 3) + 8 * (lane >> 5);
  const float s = scale ? scale[co] : 1.f;
  for (int e = 0; e < 8; ++e) {
    const float v = w[((size_t)co * 64 + ci0 + e) * 9 + tap] * s;
    dst[(size_t)idx * 8 + e] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

__global__ void c64_pack_bias_kernel(const float* __restrict__ b, float* __restrict__ dst, int Cout, int shuffle) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Cout) return;
  const int g = idx >> 6, row = idx & 63;
  dst[idx] = b[shuffle ? row * 4 + g : idx];
}

__global__ void f32_to_bf16_kernel(const f32x4* __restrict__ x, uint2* __restrict__ y, int64_t nquads) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = x[i];
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    y[i] = __builtin_bit_cast(uint2, o);
  }
}

__global__ void bf16_to_f32_kernel(const uint2* __restrict__ x, f32x4* __restrict__ y, int64_t nquads) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (int64_t)gridDim.x * blockDim.x) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 v = __builtin_bit_cast(bf16x4, x[i]);
    y[i] = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  }
}

// One workgroup per CU (a wave owns its SIMD's whole register file), persistent over work items = (image, column strip,
// row chunk).  The chunk height is chosen so that the busiest workgroup's rows -- its items x (chunk rows + the 2 halo rows
// a chunk loads without computing) -- are fewest.
int c64_plan(int N, int H, int W, int groups, int RW, int TW, int* rows_per_chunk, int* nchunks, int* nstrips) {
  const int cus = srx_plan_cus();
  const int gx_max = std::max(1, cus / groups);
  const int strips = (int)srx_cdiv(W, TW);
  const int64_t cols = (int64_t)N * strips;
  int best_rpc = (int)srx_roundup(H, RW);
  int64_t best_span = INT64_MAX;
  for (int k = 1; k <= 8; ++k) {
    const int64_t want = std::max<int64_t>(1, (int64_t)gx_max * k / cols);  // chunks per column strip
    const int rpc = (int)srx_roundup(srx_cdiv(H, want), RW);
    const int64_t chunks = srx_cdiv(H, rpc);
    // steps of the busiest workgroup: per item its rows plus the prologue (AHEAD groups loaded and written before the
    // first MFMA, about three steps' worth) -- short chunks are allowed, they just pay that more often
    const int64_t span = srx_cdiv(cols * chunks, gx_max) * (rpc / RW + 3);
    if (span < best_span) { best_span = span; best_rpc = rpc; }
  }
  // A chunk's rows are addressed by 32-bit byte offsets from the chunk's first row (rows x row bytes; the ring runs AHEAD
  // groups of RW rows ahead, plus the row above / below): the whole span must stay below 4 GiB, or offsets wrap silently
  // and rows above the image stop failing the descriptor's range check.  Shorter chunks cost time only.
  const int64_t span_rows = (int64_t)(1LL << 32) / ((int64_t)W * 128) - (2 + 3 * RW + 1);
  if (span_rows < RW) SRX_FAIL(SRX_E_UNSUPPORTED, "conv3x3_c64_bf16_fwd: image rows of %d pixels are too long for 32-bit offsets inside a chunk", W);
  if (best_rpc > span_rows) best_rpc = (int)(span_rows / RW) * RW;
  *rows_per_chunk = best_rpc;
  *nchunks = (int)srx_cdiv(H, best_rpc);
  *nstrips = strips;
  return SRX_OK;
}

template <int RW, int CW>
int launch_c64(C64Args& a, int groups, hipStream_t st) {
  constexpr int TW = 32 * CW, NR = RW == 1 ? 4 : 3 * RW;
  const size_t lds = (size_t)NR * (TW + 2) * 128 + 4 * 2 * 64 * EPI_PITCH + 4 * 4096;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c64_bf16_kernel<RW, CW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int gx_max = std::max(1, srx_plan_cus() / groups);
  if (int rc = c64_plan(a.N, a.H, a.W, groups, RW, TW, &a.rows_per_chunk, &a.chunks, &a.strips)) return rc;
  a.nwork = (int)((int64_t)a.N * a.strips * a.chunks);
  const int gx = std::min(a.nwork, gx_max);
  char nm[112];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "c64_bf16_kernel<%d, %d> MxNxK=%lldx%dx576", RW, CW, (long long)a.N * a.H * a.W, 64 * groups);
  SRX_LAUNCH_PROF(nm, 2.0 * a.N * a.H * a.W * 64.0 * groups * 576.0, (c64_bf16_kernel<RW, CW>), dim3((unsigned)gx, (unsigned)groups),
                  dim3(512), lds, st, a);
  SRX_CHECK_LAUNCH("c64_bf16_kernel");
  return SRX_OK;
}

}  // namespace

extern "C" size_t srx_conv3x3_c64_bf16_packed_bytes(int Cout) {
  return Cout > 0 && Cout % 64 == 0 ? (size_t)(Cout / 64) * KSTEPS * 2 * 1024 + (size_t)Cout * sizeof(float) : 0;
}

extern "C" int srx_conv3x3_c64_bf16_pack(const float* w, const float* bias, const float* out_scale, int Cout, int shuffle,
                                         void* wpk, void* stream) {
  SRX_REQUIRE(w && wpk && Cout > 0 && Cout % 64 == 0 && (shuffle == 0 || (shuffle == 2 && Cout == 256)),
              "conv3x3_c64_bf16_pack: Cout must be a multiple of 64 (256 with PixelShuffle 2)");
  hipStream_t st = srx_stream(stream);
  const int total = (Cout / 64) * KSTEPS * 2 * 64;
  hipLaunchKernelGGL(c64_pack_kernel, dim3((unsigned)srx_cdiv(total, 256)), dim3(256), 0, st, w, out_scale,
                     static_cast<unsigned short*>(wpk), Cout, shuffle);
  SRX_CHECK_LAUNCH("c64_pack_kernel");
  float* bdst = reinterpret_cast<float*>(static_cast<unsigned char*>(wpk) + (size_t)(Cout / 64) * KSTEPS * 2 * 1024);
  if (bias) {
    hipLaunchKernelGGL(c64_pack_bias_kernel, dim3((unsigned)srx_cdiv(Cout, 256)), dim3(256), 0, st, bias, bdst, Cout, shuffle);
    SRX_CHECK_LAUNCH("c64_pack_bias_kernel");
  } else if (hipMemsetAsync(bdst, 0, (size_t)Cout * sizeof(float), st) != hipSuccess) {
    SRX_FAIL(SRX_E_HIP, "conv3x3_c64_bf16_pack: memset failed");
  }
  return SRX_OK;
}

static int c64_fwd_impl(int N, int H, int W, int Cout, int shuffle, const void* x, const void* wpk, float slope,
                        const void* residual, void* y, int y_cs, void* stream, unsigned* dbg) {
  SRX_REQUIRE(x && wpk && y, "conv3x3_c64_bf16_fwd: null pointer");
  SRX_REQUIRE(N > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 64 == 0 && Cout <= 1024, "conv3x3_c64_bf16_fwd: bad size (Cout a multiple of 64)");
  SRX_REQUIRE(shuffle == 0 || (shuffle == 2 && Cout == 256), "conv3x3_c64_bf16_fwd: PixelShuffle(2) needs Cout = 256");
  SRX_REQUIRE(y_cs % 8 == 0 && y_cs >= (shuffle ? 64 : Cout), "conv3x3_c64_bf16_fwd: the output's channel stride must hold its channels in whole 16-byte chunks");
  SRX_REQUIRE(!residual || (!shuffle && residual != y), "conv3x3_c64_bf16_fwd: the addend is a tensor of its own, without PixelShuffle");
  SRX_REQUIRE(x != y, "conv3x3_c64_bf16_fwd: in place is not possible (neighbouring tiles read their halo)");
  SRX_REQUIRE((int64_t)W * 128 * 130 < (1LL << 32) && (int64_t)(shuffle ? 2 : 1) * W * y_cs * 2 < (1LL << 32) && (int64_t)N * H * W < (1LL << 31),
              "conv3x3_c64_bf16_fwd: image rows too long for 32-bit offsets inside a chunk");
  C64Args a{};
  a.x = static_cast<const unsigned char*>(x);
  a.w = static_cast<const unsigned char*>(wpk);
  a.bias = reinterpret_cast<const float*>(a.w + (size_t)(Cout / 64) * KSTEPS * 2 * 1024);
  a.res = static_cast<const unsigned char*>(residual);
  a.out = static_cast<unsigned char*>(y);
  a.N = N; a.H = H; a.W = W; a.shuffle = shuffle; a.out_cs = y_cs; a.slope = slope;
  a.ablate = srx_dev().c64_ablate;
  a.dbg = dbg;
  hipStream_t st = srx_stream(stream);
  const int groups = Cout / 64;
  // narrow images: waves take rows instead of column segments
  if (W <= 32) return launch_c64<4, 1>(a, groups, st);
  if (W <= 64) return launch_c64<2, 2>(a, groups, st);
  return launch_c64<1, 4>(a, groups, st);
}

// out[3] = {rows per chunk, chunks per column strip, column strips} of the launch srx_conv3x3_c64_bf16_fwd would make (host only)
extern "C" int srx_conv3x3_c64_bf16_plan(int N, int H, int W, int Cout, int* out) {
  SRX_REQUIRE(out && N > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 64 == 0, "conv3x3_c64_bf16_plan: bad argument");
  SRX_REQUIRE((int64_t)W * 128 * 130 < (1LL << 32) && (int64_t)N * H * W < (1LL << 31), "conv3x3_c64_bf16_plan: image rows too long for 32-bit offsets inside a chunk");
  const int RW = W <= 32 ? 4 : (W <= 64 ? 2 : 1), CW = W <= 32 ? 1 : (W <= 64 ? 2 : 4);
  return c64_plan(N, H, W, Cout / 64, RW, 32 * CW, out, out + 1, out + 2);
}

extern "C" int srx_conv3x3_c64_bf16_fwd(int N, int H, int W, int Cout, int shuffle, const void* x, const void* wpk, float slope,
                                        const void* residual, void* y, int y_cs, void* stream) {
  return c64_fwd_impl(N, H, W, Cout, shuffle, x, wpk, slope, residual, y, y_cs, stream, nullptr);
}

// Developer aid (tools/bench_c64.py stamps): the same launch with in-kernel cycle stamps; dbg: 256 x 4 (workgroups) x 8 (waves) x 8
// unsigned -- matrix waves {barrier wait, MFMA loop, accumulator dump, steps, total}, helpers {barrier wait, requests, addend
// wait, finishing, final wait, total} in shader cycles.
extern "C" int srx_conv3x3_c64_bf16_fwd_dbg(int N, int H, int W, int Cout, int shuffle, const void* x, const void* wpk, float slope,
                                            const void* residual, void* y, int y_cs, void* stream, unsigned* dbg) {
  SRX_REQUIRE(dbg, "conv3x3_c64_bf16_fwd_dbg: null stamp buffer");
  return c64_fwd_impl(N, H, W, Cout, shuffle, x, wpk, slope, residual, y, y_cs, stream, dbg);
}

extern "C" int srx_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  SRX_REQUIRE(x && y && n > 0 && n % 4 == 0, "f32_to_bf16: a positive multiple of 4 elements");
  const int64_t q = n / 4;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)std::min<int64_t>(srx_cdiv(q, 256), 8192)), dim3(256), 0, srx_stream(stream),
                     reinterpret_cast<const f32x4*>(x), static_cast<uint2*>(y), q);
  SRX_CHECK_LAUNCH("f32_to_bf16_kernel");
  return SRX_OK;
}

extern "C" int srx_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  SRX_REQUIRE(x && y && n > 0 && n % 4 == 0, "bf16_to_f32: a positive multiple of 4 elements");
  const int64_t q = n / 4;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)std::min<int64_t>(srx_cdiv(q, 256), 8192)), dim3(256), 0, srx_stream(stream),
                     static_cast<const uint2*>(x), reinterpret_cast<f32x4*>(y), q);
  SRX_CHECK_LAUNCH("bf16_to_f32_kernel");
  return SRX_OK;
}
