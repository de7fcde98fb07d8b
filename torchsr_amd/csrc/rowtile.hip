// 3x3, 64 -> 64, stride 1, pad 1 convolution on few pixels: the 32 + 2 convs of the SRGAN residual
// tower (srgan/residual.py:64,67, srgan/generator.py:48) and their data gradients, on 16 x 24 x 24
// low-resolution pixels at the reference batch size.
//
// That problem is 9216 x 64 x 576: only 144 of the generic 64x64 tiles, so 44 % of the chip idles and
// the rest runs two latency-bound waves per SIMD.  Here every workgroup owns 36 consecutive pixels
// (9216 = 256 CUs x 36) and all four SIMDs of its CU stay on the matrix pipe:
//
//   wave w:  output columns 32(w&1)..+31, input-channel half (w>>1) of every tap
//     32 pixels  on v_mfma_f32_32x32x2_f32  (64 cycles per channel pair)
//      4 pixels  on v_mfma_f32_4x4x1_16b_f32 ( 8 cycles per channel pair: the 16 blocks are the same 4
//                pixels against 2 x 32 columns, even channels in lanes 0-31, odd in 32-63)
//   Both MFMAs take the SAME weight operand register (lane = column, lane half = channel parity), which is
//   streamed from L2 straight into registers -- every weight element is used by exactly one wave, so
//   staging it in LDS would only add traffic.
//
// The input is staged once as a zero-padded 2-D patch (image rows r-1 .. r+span, columns -1 .. W) in LDS, so
// a tap is a constant address offset and there are no bounds checks in the loop.  Pixel stride 68 floats
// (and a row pitch of (W + 2) x 68 + 56) keeps the b128 fragment reads of a lane group on distinct banks, row wraps included.
// The two channel halves are folded through LDS; bias, activation, BatchNorm partial sums and the store
// follow the generic epilogue (gconv.hip).
#include "srx_common.h"
#include <cstdio>
#include <mutex>

namespace {

constexpr int RT36 = 36;   // pixels per workgroup: 32 + 4 (9216 pixels = 256 workgroups), or
constexpr int RT48 = 48;   // 32 + 4 x 4 (192 workgroups) when fewer than 256 CUs are free (srx_plan_cus: RCCL holds some)
constexpr int RT12 = 12;   // 3 x 4, no 32-pixel block: small batches (2 x 24 x 24 = 96 workgroups instead of 32; a third of the
                           // matrix time per workgroup -- the pre-training step at the reference's CPU batch size, BASELINE configs[0])
constexpr int PSTR = 68;   // floats per patch pixel (64 channels + 4 pad)
constexpr int RSKEW = 56;  // floats of skew per patch row (round 6): consecutive pixels are 17 sixteen-byte slots apart (= 1 mod 16: the
                           // lane groups of a ds_read_b128 -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} -- see sixteen different slots),
                           // but the step from an image row's last pixel to the next row's first is 3 x 17 = 3 mod 16, and the lanes
                           // behind the wrap collided with lanes in front of it (25 % of the LDS cycles, profiles/r05_pmc_srgan.txt);
                           // 14 more slots per patch row make that step 65 = 1 mod 16 as well
constexpr int KTOT = 576;  // 9 taps x 64 channels
constexpr int PF = 8;      // weight fragments in flight per wave
constexpr int PB = 8;      // patch b128 loads per thread and batch

struct RtArgs {
  const float* in; const float* w; const float* bias; const float* res; float* out; float* part;
  int H, W, HW, M;
  float slope;       // branch-free activation: v > 0 ? v : v * slope  (none: 1, ReLU: 0, LeakyReLU: its slope)
  unsigned in_bytes, out_bytes;
  int step_r, step_c;  // 16 / (W+2), 16 % (W+2): patch-fill stride of one thread
  // BNR (data gradients only): this launch's OUTPUT is the gradient arriving at a BatchNorm (+ PReLU) layer -- the one that
  // produced this conv's input in the forward pass -- and the epilogue also does the first pass of that layer's backward:
  // per-channel partial sums of dz = out * act'(z) and dz * xhat over the workgroup's 36 rows (and the PReLU slope's
  // partial), one table row [2C + 4] per workgroup, exactly what bn_bwd_reduce_kernel would compute from a second read
  // of `out` and `y` in a launch of its own.
  srx_rt36_bn_t bn;
  // BNL (forward only): `in` is the conv output below a BatchNorm (+ PReLU) layer; the patch pixels are normalised and
  // activated between their load and their LDS store -- the exact expression of bn_act_fwd_kernel -- and every workgroup
  // writes its own 36 transformed pixels to bnl.z_out, so the layer's normalise launch (5.3 us, 16 per generator forward,
  // a read and a write of 2.4 MB each) disappears while the backward pass still finds the activation tensor it saved.
  srx_rt36_bnl_t bnl;
  // BNB (data gradients only): `in` is the gradient arriving at the OUTPUT of a BatchNorm (+ PReLU) layer; the patch pixels
  // become the layer's input gradient -- the expression of bn_bwd_apply_kernel, from a second patch load of the layer's
  // forward input bnb.y and the finalised sums -- and every workgroup writes its own 36 pixels of it to bnb.dy_out.
  srx_rt36_bnb_t bnb;
};

// NB = batches of PB patch loads per thread (1 up to 2048 b128 slots, 2 up to the 96 KB the kernel may ask for): a
// compile-time count, so that ALL input loads and the first weight fragments are in flight together
// and the compiler can wait on them with exact vmcnt values (a runtime loop drains the queue per trip).
template <int NB, bool BNR = false, bool BNL = false, bool BNB = false, int RT = RT36>
__global__ __launch_bounds__(256) void rt36_conv3x3_c64_kernel(const RtArgs a) {
  constexpr int MAIN = RT >= 32 ? 1 : 0;        // a 32-pixel block on 32x32x2 MFMAs in front of the 4-pixel blocks
  constexpr int XB = (RT - 32 * MAIN) / 4;      // 4-pixel blocks (4x4x1 MFMAs)
  constexpr int FR = 16 * MAIN + 4 * XB;        // accumulator rows a lane holds / folds
  extern __shared__ __attribute__((aligned(16))) float patch[];
  const int tid = threadIdx.x, lane = tid & 63, wave = srx_uniform(tid >> 6);
  const int j = wave & 1, hk = wave >> 1;
  const int i31 = lane & 31, h2 = lane >> 5;
  // tiles are numbered so that each XCD (workgroups are dealt to the eight of them round robin) owns a contiguous run -- two whole
  // images at 16 x 24 x 24: the rows a tile's patch shares with its neighbours then come from that XCD's own L2 (round 6)
  const int bid = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int m0 = bid * RT;
  const int n = m0 / a.HW, p0 = m0 - n * a.HW;  // HW % 36 == 0: a tile never crosses an image
  const int r_first = p0 / a.W;
  const int r_last = (p0 + RT - 1) / a.W;
  const int W2 = a.W + 2;
  const int prows = r_last - r_first + 3;
  const __amdgpu_buffer_rsrc_t rin = srx_rsrc(a.in, a.in_bytes);
  const __amdgpu_buffer_rsrc_t rw = srx_rsrc(a.w, 64 * KTOT * 4);

  // ---- input patch loads (out-of-image and surplus slots read 0 through the descriptor's range check)
  // thread -> channel quad tid&15 of patch pixels (tid>>4) + 16u: the (row, column) of the pixel advances
  // by the host-computed (16 / W2, 16 % W2) per step, no division in the loop
  f32x4 v[NB * PB];
  const unsigned quad16 = 16u * (tid & 15);
  int pr = (tid >> 4) / W2, pc = (tid >> 4) - pr * W2, sidx = tid >> 4;
  const int npix = prows * W2;
  const int ROWP = W2 * PSTR + RSKEW;  // floats per patch row
  int loff[NB * PB];           // LDS offset (floats) of the slot's quad
  unsigned okm = 0, ownm = 0;  // BNL / BNB: slots that hold an image pixel / one of this workgroup's own 36 pixels
  unsigned zoff[NB * PB];      // BNL / BNB: byte offset of the slot's pixel quad (input and side output have the same shape)
  f32x4 yb[(BNB || BNL) ? NB * PB : 1];  // BNB: the BatchNorm layer's forward input at the slot; BNL: the addend (or zeros)
  const bool has_res = BNL && a.bnl.res != nullptr;
  const __amdgpu_buffer_rsrc_t rby = srx_rsrc(BNB ? a.bnb.y : (has_res ? a.bnl.res : a.in), a.in_bytes);
#pragma unroll
  for (int u = 0; u < NB * PB; ++u) {
    const int ih = r_first - 1 + pr, iw = pc - 1;
    const bool ok = sidx < npix && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    const unsigned off = (unsigned)((n * a.H + ih) * a.W + iw) * 256u + quad16;
    v[u] = srx_bload(rin, ok ? off : 0xffffffffu, 0);
    loff[u] = pr * ROWP + pc * PSTR + 4 * (tid & 15);
    if constexpr (BNB) yb[u] = srx_bload(rby, ok ? off : 0xffffffffu, 0);
    if constexpr (BNL) yb[u] = srx_bload(rby, (ok && has_res) ? off : 0xffffffffu, 0);  // (no addend: reads 0, touches nothing)
    if constexpr (BNL || BNB) {
      const unsigned q = (unsigned)(ih * a.W + iw - p0);  // (wraps for pixels in front of the tile)
      okm |= (ok ? 1u : 0u) << u;
      ownm |= ((ok && q < (unsigned)RT) ? 1u : 0u) << u;
      zoff[u] = off;
    }
    sidx += 16; pr += a.step_r; pc += a.step_c;
    if (pc >= W2) { pc -= W2; pr += 1; }
  }
  f32x4 nmu, nis, ngm, nbt, nsd, nsx;
  float nsl = 1.f;
  if constexpr (BNB) {
    nmu = *reinterpret_cast<const f32x4*>(a.bnb.mean + 4 * (tid & 15));
    nis = *reinterpret_cast<const f32x4*>(a.bnb.invstd + 4 * (tid & 15));
    ngm = *reinterpret_cast<const f32x4*>(a.bnb.gamma + 4 * (tid & 15));
    nbt = *reinterpret_cast<const f32x4*>(a.bnb.beta + 4 * (tid & 15));
    nsd = *reinterpret_cast<const f32x4*>(a.bnb.sums + 4 * (tid & 15));
    nsx = *reinterpret_cast<const f32x4*>(a.bnb.sums + 64 + 4 * (tid & 15));
    if (a.bnb.prelu) nsl = a.bnb.prelu[0];
  }
  if constexpr (BNL) {  // this thread's four channels (quad tid & 15) of the layer's constants
    nmu = *reinterpret_cast<const f32x4*>(a.bnl.mean + 4 * (tid & 15));
    nis = *reinterpret_cast<const f32x4*>(a.bnl.invstd + 4 * (tid & 15));
    ngm = *reinterpret_cast<const f32x4*>(a.bnl.gamma + 4 * (tid & 15));
    nbt = *reinterpret_cast<const f32x4*>(a.bnl.beta + 4 * (tid & 15));
    if (a.bnl.prelu) nsl = a.bnl.prelu[0];
  }
  // ---- weight stream: fragment `it` = (tap it>>2, channels 32hk + 8(it&3) + 4h2 .. +3) of column 32j+i31
  const unsigned wvoff = 4u * (unsigned)((32 * j + i31) * KTOT + 32 * hk + 4 * h2);
  f32x4 bq[PF];
#pragma unroll
  for (int it = 0; it < PF; ++it) bq[it] = srx_bload(rw, wvoff, (unsigned)(((it >> 2) * 64 + 8 * (it & 3)) * 4));
  __builtin_amdgcn_sched_barrier(0);  // keep the weight loads ahead of the waits on the patch loads
  if constexpr (BNL) {  // normalise + activate in registers; padding slots stay zero (the conv pads the ACTIVATION with zeros)
    const __amdgpu_buffer_rsrc_t rz = srx_rsrc(a.bnl.z_out, a.in_bytes);
#pragma unroll
    for (int u = 0; u < NB * PB; ++u) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float z = (v[u][e] - nmu[e]) * (nis[e] * ngm[e]) + nbt[e];
        o[e] = z > 0.f ? z : z * nsl;
      }
      if (has_res) o += yb[u];
      v[u] = ((okm >> u) & 1u) ? o : f32x4{0.f, 0.f, 0.f, 0.f};
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[u]), rz,
                                             (int)(((ownm >> u) & 1u) ? zoff[u] : 0xffffffffu), 0, 0);  // (out of range: dropped)
    }
  }
  if constexpr (BNB) {  // BatchNorm (+ PReLU) backward, second pass, in registers; padding slots stay zero
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rz = srx_rsrc(a.bnb.dy_out, a.in_bytes);
    const float invM = a.bnb.inv_m;
    const bool has_act = a.bnb.prelu != nullptr;
#pragma unroll
    for (int u = 0; u < NB * PB; ++u) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (yb[u][e] - nmu[e]) * nis[e];
        const float z = xh * ngm[e] + nbt[e];
        const float dz = v[u][e] * (has_act ? (z > 0.f ? 1.f : nsl) : 1.f);
        o[e] = ngm[e] * nis[e] * (dz - nsd[e] * invM - xh * nsx[e] * invM);
      }
      v[u] = ((okm >> u) & 1u) ? o : f32x4{0.f, 0.f, 0.f, 0.f};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[u]), rz,
                                             (int)(((ownm >> u) & 1u) ? zoff[u] : 0xffffffffu), 0, 0);  // (out of range: dropped)
    }
  }
  // ---- patch -> LDS.  Surplus slots (e >= nslots) land in the slack the host adds behind the patch.
#pragma unroll
  for (int u = 0; u < NB * PB; ++u) *reinterpret_cast<f32x4*>(patch + loff[u]) = v[u];
  __syncthreads();

  // ---- per-lane patch addresses (tap (0,0) = one row up, one column left: the patch origin is (-1,-1))
  auto slot = [&](int q) {  // pixel q of the image -> patch slot of its (-1,-1) neighbour
    const int ih = q / a.W, iw = q - ih * a.W;
    return (ih - r_first) * ROWP + iw * PSTR;
  };
  const int choff = 32 * hk + 4 * h2;
  const float* a32 = patch + (MAIN ? slot(p0 + i31) : 0) + choff;
  const float* a4[XB];
#pragma unroll
  for (int b = 0; b < XB; ++b) a4[b] = patch + slot(p0 + 32 * MAIN + 4 * b + (lane & 3)) + choff;
  const int rowoff = ROWP;

  // two accumulator chains: with one wave per SIMD a single dependent MFMA chain leaves issue gaps
  f32x16 acc, accb;
  f32x4 acc4[XB];
#pragma unroll
  for (int b = 0; b < XB; ++b) acc4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }

  f32x4 fa[2], fb[2][XB];  // A fragments of step it+1 are read while the MFMAs of step it run
  auto frag = [&](int it, int set) {
    const int tap = it >> 2, th = tap / 3, tw = tap - 3 * th;
    const int off = th * rowoff + tw * PSTR + 8 * (it & 3);
    if constexpr (MAIN) fa[set] = *reinterpret_cast<const f32x4*>(a32 + off);
#pragma unroll
    for (int b = 0; b < XB; ++b) fb[set][b] = *reinterpret_cast<const f32x4*>(a4[b] + off);
  };
  frag(0, 0);
#pragma unroll
  for (int it = 0; it < 36; ++it) {
    if (it + 1 < 36) frag(it + 1, (it + 1) & 1);
    const f32x4 b = bq[it % PF];
    if (it + PF < 36) {
      const int nx = it + PF;
      bq[it % PF] = srx_bload(rw, wvoff, (unsigned)(((nx >> 2) * 64 + 8 * (nx & 3)) * 4));
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (MAIN) {
        if (e & 1) accb = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[it & 1][e], b[e], accb, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[it & 1][e], b[e], acc, 0, 0, 0);
      }
#pragma unroll
      for (int xb = 0; xb < XB; ++xb) acc4[xb] = __builtin_amdgcn_mfma_f32_4x4x1f32(fb[it & 1][xb][e], b[e], acc4[xb], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  }
#pragma unroll
  for (int r = 0; r < 16 * MAIN; ++r) acc[r] += accb[r];
  // even / odd channel halves of the 4-pixel block live in lanes l and l+32
#pragma unroll
  for (int b = 0; b < XB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[b][i] += __shfl_xor(acc4[b][i], 32, 64);

  // ---- fold the two input-channel halves (the patch is dead now)
  __syncthreads();
  float* fold = patch;  // [j][FR][64]
  if (hk == 1) {
#pragma unroll
    for (int r = 0; r < 16 * MAIN; ++r) fold[(j * FR + r) * 64 + lane] = acc[r];
#pragma unroll
    for (int i = 0; i < 4 * XB; ++i) fold[(j * FR + 16 * MAIN + i) * 64 + lane] = acc4[i >> 2][i & 3];
  }
  __syncthreads();
  if (hk == 1) return;
#pragma unroll
  for (int r = 0; r < 16 * MAIN; ++r) acc[r] += fold[(j * FR + r) * 64 + lane];
#pragma unroll
  for (int i = 0; i < 4 * XB; ++i) acc4[i >> 2][i & 3] += fold[(j * FR + 16 * MAIN + i) * 64 + lane];

  // ---- epilogue.  32x32 accumulator: col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5);
  //      4x4 accumulator: row 32+i, col = lane&31 (both lane halves hold the folded sum; half 0 stores)
  //      (M is a multiple of 36: every row of the tile exists.)  Straight-line code: buffer stores with
  //      32-bit offsets, activation as a select on a host-prepared slope.
  const int col = 32 * j + i31;
  const float bv = a.bias ? a.bias[col] : 0.f;
  const __amdgpu_buffer_rsrc_t rout = srx_rsrc(a.out, a.out_bytes);
  const __amdgpu_buffer_rsrc_t rres = srx_rsrc(a.res ? a.res : a.out, a.out_bytes);  // eval-mode skip input
  const unsigned obase = ((unsigned)m0 * 64u + (unsigned)col + 256u * h2) * 4u;
  float s1 = 0.f, s2 = 0.f;
  const unsigned obase4 = h2 == 0 ? ((unsigned)m0 * 64u + (unsigned)col) * 4u : 0xffffffffu;  // half 0 stores
  // the addend (eval-mode skip input, or the skip connection's gradient when this is a data gradient) is requested for
  // all 20 rows before the first store: a load issued behind a store cannot be consumed until that store has landed
  float rv[16], rv4[4 * XB];
#pragma unroll
  for (int r = 0; r < 16 * MAIN; ++r)
    rv[r] = a.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)obase, ((r & 3) + 8 * (r >> 2)) * 256, 0)) : 0.f;
#pragma unroll
  for (int i = 0; i < 4 * XB; ++i)
    rv4[i] = a.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rres, (int)obase4, (32 * MAIN + i) * 256, 0)) : 0.f;
  // BNR: the BatchNorm input at this lane's 20 (row, column) positions, and the layer's per-channel constants
  float yv[16], yv4[4 * XB];
  float bmu = 0.f, bis = 0.f, bgm = 0.f, bbt = 0.f, bsl = 1.f, t1 = 0.f, t2 = 0.f, tp = 0.f;
  if constexpr (BNR) {
    const __amdgpu_buffer_rsrc_t rbn = srx_rsrc(a.bn.y, a.out_bytes);
#pragma unroll
    for (int r = 0; r < 16 * MAIN; ++r)
      yv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbn, (int)obase, ((r & 3) + 8 * (r >> 2)) * 256, 0));
#pragma unroll
    for (int i = 0; i < 4 * XB; ++i)
      yv4[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rbn, (int)obase4, (32 * MAIN + i) * 256, 0));
    bmu = a.bn.mean[col]; bis = a.bn.invstd[col]; bgm = a.bn.gamma[col]; bbt = a.bn.beta[col];
    if (a.bn.prelu) bsl = a.bn.prelu[0];
  }
  auto bn_acc = [&](float o, float y) {
    const float xh = (y - bmu) * bis;
    const float z = xh * bgm + bbt;
    const bool pos = z > 0.f;
    const float dz = pos ? o : o * bsl;
    t1 += dz;
    t2 += dz * xh;
    if (a.bn.prelu && !pos) tp += o * z;
  };
#pragma unroll
  for (int r = 0; r < 16 * MAIN; ++r) {
    const float v = acc[r] + bv;
    s1 += v;
    s2 += v * v;
    const float o = (v > 0.f ? v : v * a.slope) + rv[r];
    if constexpr (BNR) bn_acc(o, yv[r]);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rout, obase, ((r & 3) + 8 * (r >> 2)) * 256, 0);
  }
#pragma unroll
  for (int i = 0; i < 4 * XB; ++i) {
    const float v = acc4[i >> 2][i & 3] + bv;
    if (h2 == 0) { s1 += v; s2 += v * v; }
    const float o = (v > 0.f ? v : v * a.slope) + rv4[i];
    if constexpr (BNR) { if (h2 == 0) bn_acc(o, yv4[i]); }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rout, obase4, (32 * MAIN + i) * 256, 0);
  }
  if constexpr (BNR) {  // table row of this workgroup: [0, C) sum dz, [C, 2C) sum dz * xhat, 2C + j the PReLU partial of wave j
    t1 += __shfl_xor(t1, 32, 64);
    t2 += __shfl_xor(t2, 32, 64);
    tp = srx_wave_sum(tp);
    float* row = a.bn.part + (size_t)bid * (2 * 64 + 4);
    if (h2 == 0) { row[col] = t1; row[64 + col] = t2; }
    if (lane == 0) row[128 + j] = tp;
  }
  if (a.part) {  // per-channel (sum, sum of squares) of this workgroup's 36 rows
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (h2 == 0) {
      a.part[((size_t)bid * 64 + col) * 2 + 0] = s1;
      a.part[((size_t)bid * 64 + col) * 2 + 1] = s2;
    }
  }
}

int patch_rows_max(int W, int RT) {  // tiles start at columns (RT t) mod W only
  int rows = 0;
  for (int t = 0; t < W; ++t) {
    const int r = ((RT * t) % W + RT - 1) / W + 3;
    if (r > rows) rows = r;
  }
  return rows;
}

int patch_batches(int W, int RT) { return (int)srx_cdiv((int64_t)patch_rows_max(W, RT) * (W + 2) * 16, 256 * PB); }

size_t lds_bytes(int W, int RT) {  // every thread stores all its NB * PB slots: size for the rounded-up slot count
  // (rows the rounded-up slot count reaches, each W + 2 pixels + the skew)
  const size_t slots = (size_t)patch_batches(W, RT) * (256 * PB / 16);
  const size_t patch = ((slots + (size_t)W + 1) / (size_t)(W + 2)) * ((size_t)(W + 2) * PSTR + RSKEW) * sizeof(float);
  const size_t fold = 2 * (size_t)(RT >= 32 ? RT - 16 : RT) * 64 * sizeof(float);
  return patch > fold ? patch : fold;
}

bool tile_fits(const srx_conv2d_t* d, int RT) {
  const int64_t hw = (int64_t)d->H * d->W;
  return hw % RT == 0 && d->W >= 3 && patch_batches(d->W, RT) <= 2 && lds_bytes(d->W, RT) <= 96 * 1024;
}

// Pixels per workgroup.  36 cuts the reference batch (16 x 24 x 24) into exactly 256 workgroups, one per CU; when the plan may
// not count on every CU (srx_plan_cus() < workgroups: a gradient all-reduce's channel kernels hold some) a second round of
// 36-pixel tiles on a few CUs would double the launch, and 48-pixel tiles (192 workgroups, 4/3 of the work each) are the
// better cut.  0: the layer does not run on this kernel.
int tile_pixels(const srx_conv2d_t* d) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->shuffle || d->up || d->precision) return 0;
  if (d->Cin != 64 || d->Cout != 64 || d->Cin_s != 64 || d->Cout_s != 64) return 0;
  if (srx_dev().no_rt36) return 0;  // developer switch (read at load time): force the generic kernel
  const int64_t m = (int64_t)d->N * d->H * d->W;
  const int cus = srx_plan_cus();
  // small problems only: above ~2 rounds of the chip the generic 128-row tiles re-read far less input
  if (!tile_fits(d, RT36) || m / RT36 > 2 * cus) return 0;
  if (m / RT36 > cus && tile_fits(d, RT48) && m / RT48 <= cus) return RT48;
  if (m / RT36 <= cus / 4 && tile_fits(d, RT12)) return RT12;  // a quarter of the chip or less at 36 pixels: finer tiles
  return RT36;
}

}  // namespace

bool srx_rt36_applicable(const srx_conv2d_t* d) { return tile_pixels(d) != 0; }

int srx_rt36_rows(const srx_conv2d_t* d) { const int rt = tile_pixels(d); return rt ? (int)((int64_t)d->N * d->H * d->W / rt) : 0; }

int srx_rt36_run(const srx_conv2d_t* d, const float* in, const float* wpk, const float* bias, const float* residual,
                 float* out, float* part, int act, float slope, hipStream_t st, const srx_rt36_bn_t* bn,
                 const srx_rt36_bnl_t* bnl, const srx_rt36_bnb_t* bnb) {
  RtArgs a{};
  if (bn) a.bn = *bn;
  if (bnl) a.bnl = *bnl;
  if (bnb) a.bnb = *bnb;
  a.in = in; a.w = wpk; a.bias = bias; a.res = residual; a.out = out; a.part = part;
  a.H = d->H; a.W = d->W; a.HW = d->H * d->W; a.M = d->N * a.HW;
  a.slope = act == SRX_ACT_RELU ? 0.f : (act == SRX_ACT_LRELU ? slope : 1.f);
  a.in_bytes = (unsigned)((size_t)a.M * 64 * sizeof(float));
  a.out_bytes = a.in_bytes;
  a.step_r = 16 / (d->W + 2); a.step_c = 16 % (d->W + 2);
  const int RT = tile_pixels(d);
  if (RT == 0) SRX_FAIL(SRX_E_UNSUPPORTED, "rt36: the layer does not run on the row-tile kernel");
  const size_t lds = lds_bytes(d->W, RT);
  const int nb = patch_batches(d->W, RT);
  const double fl = 2.0 * a.M * 64 * KTOT;
  const dim3 grid((unsigned)(a.M / RT));
  char nm[112];
  // (the BatchNorm folds are part of the name: the bench line reports the forms the step really runs, not only the plain conv)
  if (srx_prof_on())
    snprintf(nm, sizeof(nm), "rt36_conv3x3_c64_kernel<%d%s%s%s> MxNxK=%dx64x%d", nb, bn ? ", BNR" : "", bnl ? ", BNL" : "", bnb ? ", BNB" : "", a.M, KTOT);
  // one instance per (patch batches, BatchNorm modes, tile pixels); each raises its LDS limit once
#define RT_LAUNCH(NB_, R_, L_, B_, T_)                                                                                    \
  do {                                                                                                                    \
    static std::once_flag once_;                                                                                          \
    std::call_once(once_, [] {                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rt36_conv3x3_c64_kernel<NB_, R_, L_, B_, T_>),             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                   \
    });                                                                                                                   \
    SRX_LAUNCH_PROF(nm, fl, (rt36_conv3x3_c64_kernel<NB_, R_, L_, B_, T_>), grid, dim3(256), lds, st, a);                 \
  } while (0)
#define RT_MODES(NB_, T_)                                              \
  do {                                                                 \
    if (bnb && bn) RT_LAUNCH(NB_, true, false, true, T_);              \
    else if (bnb) RT_LAUNCH(NB_, false, false, true, T_);              \
    else if (bnl) RT_LAUNCH(NB_, false, true, false, T_);              \
    else if (bn) RT_LAUNCH(NB_, true, false, false, T_);               \
    else RT_LAUNCH(NB_, false, false, false, T_);                      \
  } while (0)
  if (RT == RT36) { if (nb == 1) RT_MODES(1, RT36); else RT_MODES(2, RT36); }
  else if (RT == RT12) { if (nb == 1) RT_MODES(1, RT12); else RT_MODES(2, RT12); }
  else { if (nb == 1) RT_MODES(1, RT48); else RT_MODES(2, RT48); }
#undef RT_MODES
#undef RT_LAUNCH
  SRX_CHECK_LAUNCH("rt36_conv3x3_c64_kernel");
  return SRX_OK;
}
