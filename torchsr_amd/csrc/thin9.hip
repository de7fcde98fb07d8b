// The generator's output conv -- nn.Conv2d(64, 3, 9, 1, 4), srgan/generator.py:58,80 -- on a bf16 NHWC tensor (round 4).
//
// thin_fwd2_bf16_kernel runs it on v_mfma_f32_4x4x4_16b_bf16 (one MFMA per 4 pixels x 4 channels x 4 k): at 7680 x 4320
// it is 4.4 ms of the 13 ms bf16 frame.  With only 3 output channels the GEMM has no N -- unless the taps supply it:
//
//   D_r[(kh, c)][x] = sum_{kw, ci} in[r][x + kw - 4][ci] * w[c][ci][kh][kw]          (one INPUT row r at a time)
//   out[y][x][c]    = bias[c] + sum_kh D_{y + kh - 4}[(kh, c)][x]
//
// i.e. N = 9 row taps x 3 channels = 27 (-> 32 MFMA rows), K = 9 column taps x 64 channels = 576: exactly the shape of
// c64_bf16_kernel (c64.hip) -- 36 k-steps, 144 weight registers per matrix wave, a column tap is an address offset into the
// window row, one ds_read_b128 per v_mfma_f32_32x32x16_bf16 -- with a SINGLE window row per tile (no vertical taps in the
// MFMA loop) and the vertical sum done by the helper waves: an input row's 27 partial rows are added into a ring of ten
// output rows in LDS; the row tap kh = 8 completes an output row, which then gets its bias and goes out as 16 bytes per pixel.
// Same skeleton as c64.hip: four matrix waves (one 32-pixel segment each, two input rows per step = 72 MFMAs), four helper
// waves (window rows by LDS-DMA one step ahead, the ring), one LDS-only barrier per step, persistent workgroups over
// (image, 128-column strip, row chunk) items, 64-bit row bases.
#include "srx_common.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KSTEPS = 36;          // 9 column taps x 4 channel groups of 16
constexpr int TW = 128, PX = TW + 8;  // output columns per strip; window pixels per row (4 halo pixels each side)
constexpr int RW = 2;               // input rows per step
constexpr int NR = 2 * RW;          // ring: the group being read + the group being requested
constexpr int ROWB = PX * 128;
constexpr int UNITS = RW * PX * 8, NLD = (UNITS + 255) / 256;  // 16-byte units per group / per helper thread
constexpr int EPI_PITCH = 144, ACCB = 64 * EPI_PITCH;            // accumulators pixel-major: [2 rows][32 pixels][32 floats + pad]
constexpr int ORING = 10;           // output rows in flight: y = r - 4 .. r + 4 for one input row, + 1 for the second
constexpr int OFF_ACC = NR * ROWB, OFF_RING = OFF_ACC + 4 * 2 * ACCB, LDS_BYTES = OFF_RING + ORING * TW * 16;  // 163840

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void dma16(const u32x4& rsrc, unsigned voff, unsigned lds_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_base), "s"(rsrc) : "memory");
}

struct T9Args {
  const unsigned char* x;   // bf16 NHWC [N][H][W][64]
  const unsigned char* w;   // packed: [kstep 36][lane 64][8 bf16], then 4 floats of bias
  float* out;               // fp32 [N][H][W][4]
  int N, H, W;
  int strips, chunks, rows_per_chunk, nwork;
};

__global__ __launch_bounds__(512) void t9_bf16_kernel(const T9Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wave = srx_uniform((int)threadIdx.x >> 6);
  const bool helper = wave >= 4;
  const int w4 = wave & 3;            // the 32-pixel segment this wave computes / accumulates
  const int l31 = lane & 31, h = lane >> 5;
  unsigned char* win = smem;
  unsigned char* accbuf = smem + OFF_ACC + w4 * (2 * ACCB);
  unsigned char* oring = smem + OFF_RING;
  const size_t in_row_bytes = (size_t)a.W * 128;

  // Work item = (image, strip, chunk of output rows [r_beg, r_end)); its input rows are r_beg - 4 .. r_end + 3, taken two at a
  // time: step k multiplies input rows rin0 + 2k, + 1 (ring slots 2 (k & 1), + 1) while the helpers request the next two and
  // add the partial rows of step k - 1 into the output ring.  Both roles run the same barriers: A per item, B_k per step.
  if (!helper) {
    __builtin_amdgcn_s_setprio(3);
    bf16x8 wf[KSTEPS];
    {
      const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + lane;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) wf[ks] = __builtin_bit_cast(bf16x8, wp[ks * 64]);
    }
    // window offsets of the B fragments: pixel 32 w4 + l31 + kw (column c0 - 4 + pixel), chunk (2 cs + h) ^ swizzle(pixel);
    // the channel group flips bits 5..6 of the offset: one register per column tap, one v_xor per fragment
    unsigned fbase[9];
#pragma unroll
    for (int kw = 0; kw < 9; ++kw) {
      const int p = 32 * w4 + l31 + kw;
      fbase[kw] = (unsigned)(p * 128 + ((h ^ ((p >> 1) & 7)) * 16));
    }
    for (int wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
      const int chunk = wi % a.chunks;
      const int r_beg = chunk * a.rows_per_chunk, r_end = min(a.H, r_beg + a.rows_per_chunk);
      const int nsteps = (r_end - r_beg + 8 + RW - 1) / RW;
      lds_barrier();  // A
      for (int k = 0; k <= nsteps; ++k) {
        lds_barrier();  // B_k
        if (k == nsteps) break;
        unsigned char* dst = accbuf + (k & 1) * ACCB;
        const unsigned sb0 = (unsigned)srx_uniform(((k & 1) * RW) * ROWB), sb1 = sb0 + ROWB;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        constexpr int PD = 3;
        bf16x8 b0[PD], b1[PD];
        auto fetch = [&](int ks, int slot) {
          const int kw = ks >> 2, cs = ks & 3;
          const unsigned off = fbase[kw] ^ (unsigned)(cs << 5);
          b0[slot] = *reinterpret_cast<const bf16x8*>(win + sb0 + off);
          b1[slot] = *reinterpret_cast<const bf16x8*>(win + sb1 + off);
        };
#pragma unroll
        for (int ks = 0; ks < PD; ++ks) fetch(ks, ks);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bf16x8 x0 = b0[ks % PD], x1 = b1[ks % PD];
          if (ks + PD < KSTEPS) fetch(ks + PD, ks % PD);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], x0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], x1, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (ks + PD < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        // D[row = (kh, c) = (r & 3) + 8 (r >> 2) + 4 h][col = pixel l31] -> pixel-major fp32: [input row of the pair][pixel][32]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f32x4*>(dst + l31 * EPI_PITCH + (8 * q + 4 * h) * 4) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
          *reinterpret_cast<f32x4*>(dst + (32 + l31) * EPI_PITCH + (8 * q + 4 * h) * 4) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
        }
      }
    }
    return;
  }

  // ---- helper waves
  const f32x4 bias = *reinterpret_cast<const f32x4*>(a.w + KSTEPS * 1024);
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
    scol[u] = px - 4;
    ssrc[u] = (unsigned)((pos ^ ((px >> 1) & 7)) * 16);
  }
  const unsigned win_lds = (unsigned)(size_t)win;

  for (int wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
    int t = wi;
    const int chunk = t % a.chunks; t /= a.chunks;
    const int strip = t % a.strips;
    const int n = t / a.strips;
    const int c0 = strip * TW;
    const int r_beg = chunk * a.rows_per_chunk, r_end = min(a.H, r_beg + a.rows_per_chunk);
    const int rin0 = r_beg - 4;
    const int nsteps = (r_end - r_beg + 8 + RW - 1) / RW;
    const int rb = max(rin0, 0), re = min(rin0 + nsteps * RW, a.H);
    u32x4 rx;
    {
      const unsigned long long xb = (unsigned long long)(a.x + ((size_t)n * a.H + rb) * in_row_bytes);
      rx[0] = (unsigned)srx_uniform((int)(unsigned)xb);
      rx[1] = (unsigned)srx_uniform((int)((unsigned)(xb >> 32) & 0xffffu));
      rx[2] = (unsigned)srx_uniform((int)(unsigned)((size_t)(re - rb) * in_row_bytes));
      rx[3] = 0x00020000u;
    }
    unsigned goff[NLD];
    bool colok[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int col = c0 + scol[u];
      colok[u] = sok[u] && (unsigned)col < (unsigned)a.W;
      goff[u] = (unsigned)(rin0 + srow[u] - rb) * (unsigned)in_row_bytes + (unsigned)col * 128u + ssrc[u];
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
    // output rows: base pointer of row r_beg of this strip's image; per lane the byte offset of its pixel
    const unsigned long long ob = (unsigned long long)(a.out + (((size_t)n * a.H + (size_t)r_beg) * a.W) * 4);
    const unsigned out_row_bytes = (unsigned)a.W * 16u;
    const int colx = c0 + 32 * w4 + l31;
    const unsigned xoff = colx < a.W ? (unsigned)colx * 16u : 0xffffffffu;
    unsigned char* myring = oring + (32 * w4 + l31) * 16;

    // the partial rows of the input-row pair of step k: lane (pixel l31, h) adds row taps kh = 0..3 (h = 0) or 4..8 (h = 1) of
    // each of the two input rows into ring row (r - kh + 4) % 10; kh = 8 completes output row r - 4
    auto accumulate = [&](int k, const unsigned char* src) {
#pragma unroll
      for (int ir = 0; ir < 2; ++ir) {
        const int r = rin0 + RW * k + ir;                 // input row
        const unsigned char* sp = src + (32 * ir + l31) * EPI_PITCH + (h ? 48 : 0);
        f32x4 dv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dv[i] = *reinterpret_cast<const f32x4*>(sp + 16 * i);  // 16 floats: row taps 0..3 (+ a spare) or 4..8
        float dvals[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) dvals[i] = dv[i >> 2][i & 3];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int kh = h ? 4 + j : j;                   // (per lane half)
          const int y = r - kh + 4;                       // output row this tap feeds
          int slot = (y - r_beg + 3 * ORING) % ORING;     // (y - r_beg >= -9)
          f32x4* rp = reinterpret_cast<f32x4*>(myring + slot * (TW * 16));
          const bool live = (j < 4 || h == 1);            // h = 0 has four taps
          f32x4 v = *rp;
          v[0] += dvals[3 * j]; v[1] += dvals[3 * j + 1]; v[2] += dvals[3 * j + 2];
          if (j == 4) {                                   // kh = 8 (h = 1 lanes): output row r - 4 is complete
            const int yc = r - 4;                         // (wave-uniform, unlike y)
            const bool store = h == 1 && yc >= r_beg && yc < r_end;
            const unsigned long long rowp = ob + (unsigned long long)(unsigned)(max(yc - r_beg, 0)) * out_row_bytes;
            u32x4 ro;
            ro[0] = (unsigned)srx_uniform((int)(unsigned)rowp);
            ro[1] = (unsigned)srx_uniform((int)((unsigned)(rowp >> 32) & 0xffffu));
            ro[2] = out_row_bytes;
            ro[3] = 0x00020000u;
            const f32x4 o = f32x4{v[0] + bias[0], v[1] + bias[1], v[2] + bias[2], 0.f};
            const u32x4 od = __builtin_bit_cast(u32x4, o);
            const unsigned so = store ? xoff : 0xffffffffu;
            asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(od), "v"(so), "s"(ro) : "memory");
            if (h == 1) *rp = f32x4{0.f, 0.f, 0.f, 0.f};  // the slot's next tenant starts from zero
          } else if (live) {
            *rp = v;
          }
        }
      }
    };

    lds_barrier();  // A
    // the ring starts every item empty: the rows below r_end that the last item fed but never completed are dropped here
    // (a completed row zeroes its own slot as it leaves); ordered before the first accumulation by B_0 / B_1
    for (int i = tid; i < ORING * TW; i += 256) *reinterpret_cast<f32x4*>(oring + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    dma_group(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int k = 0; k <= nsteps; ++k) {
      lds_barrier();  // B_k
      if (k + 1 < nsteps) dma_group(((k + 1) & 1) * RW);   // the pair of step k + 1
      if (k > 0) accumulate(k - 1, accbuf + ((k - 1) & 1) * ACCB);
      // the requested rows have landed before the next barrier; this step's stores (2 per wave) may fly on
      if (k > 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

// OIHW fp32 [Cout <= 3][64][9][9] (+ bias) -> [kstep = kw * 4 + cs][lane][8 bf16]: lane l holds MFMA row (l & 31) = kh * 3 + c
// (rows >= 27 and channels >= Cout: zero), input channels 16 cs + 8 (l >> 5) .. + 7 of column tap kw; then the bias as 4 floats.
__global__ void t9_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias, unsigned char* __restrict__ dst, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < 4) reinterpret_cast<float*>(dst + KSTEPS * 1024)[idx] = (bias && idx < Cout) ? bias[idx] : 0.f;
  if (idx >= KSTEPS * 64) return;
  const int lane = idx & 63, ks = idx >> 6;
  const int row = lane & 31, kw = ks >> 2, ci0 = 16 * (ks & 3) + 8 * (lane >> 5);
  const int kh = row / 3, c = row - 3 * kh;
  unsigned short* d = reinterpret_cast<unsigned short*>(dst) + (size_t)idx * 8;
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (row < 27 && c < Cout) v = w[(((size_t)c * 64 + ci0 + e) * 9 + kh) * 9 + kw];
    d[e] = __builtin_bit_cast(unsigned short, (__bf16)v);
  }
}

}  // namespace

extern "C" size_t srx_conv9x9_c64_thin_bf16_packed_bytes(void) { return (size_t)KSTEPS * 1024 + 16; }

extern "C" int srx_conv9x9_c64_thin_bf16_pack(const float* w, const float* bias, int Cout, void* wpk, void* stream) {
  SRX_REQUIRE(w && wpk && Cout >= 1 && Cout <= 3, "conv9x9_c64_thin_bf16_pack: 1..3 output channels");
  hipLaunchKernelGGL(t9_pack_kernel, dim3((unsigned)srx_cdiv(KSTEPS * 64, 256)), dim3(256), 0, srx_stream(stream), w, bias,
                     static_cast<unsigned char*>(wpk), Cout);
  SRX_CHECK_LAUNCH("t9_pack_kernel");
  return SRX_OK;
}

extern "C" int srx_conv9x9_c64_thin_bf16_fwd(int N, int H, int W, const void* x, const void* wpk, float* y, void* stream) {
  SRX_REQUIRE(x && wpk && y && N > 0 && H > 0 && W > 0, "conv9x9_c64_thin_bf16_fwd: bad argument");
  SRX_REQUIRE((int64_t)W * 128 * 140 < (1LL << 32) && (int64_t)N * H < (1LL << 31), "conv9x9_c64_thin_bf16_fwd: image rows too long for 32-bit offsets inside a chunk");
  T9Args a{};
  a.x = static_cast<const unsigned char*>(x); a.w = static_cast<const unsigned char*>(wpk); a.out = y;
  a.N = N; a.H = H; a.W = W;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&t9_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  // one workgroup per CU, persistent; a chunk re-reads 8 input rows: keep chunks long, and as many items as CUs (x k)
  const int cus = srx_plan_cus();
  a.strips = (int)srx_cdiv(W, TW);
  const int64_t cols = (int64_t)N * a.strips;
  int best_rpc = H;
  int64_t best_span = INT64_MAX;
  for (int k = 1; k <= 8; ++k) {
    const int64_t want = std::max<int64_t>(1, (int64_t)cus * k / cols);
    const int rpc = (int)srx_roundup(srx_cdiv(H, want), RW);
    const int64_t chunks = srx_cdiv(H, rpc);
    const int64_t span = srx_cdiv(cols * chunks, cus) * ((rpc + 8) / RW + 3);
    if (span < best_span) { best_span = span; best_rpc = rpc; }
  }
  {  // 32-bit byte offsets inside a chunk (rows_per_chunk + the 8-row window + the rows in flight): see c64.hip
    const int64_t span_rows = (int64_t)(1LL << 32) / ((int64_t)W * 128) - (8 + 4);
    if (span_rows < RW) SRX_FAIL(SRX_E_UNSUPPORTED, "conv9x9_c64_thin_bf16_fwd: image rows of %d pixels are too long for 32-bit offsets inside a chunk", W);
    if (best_rpc > span_rows) best_rpc = (int)(span_rows / RW) * RW;
  }
  a.rows_per_chunk = best_rpc;
  a.chunks = (int)srx_cdiv(H, best_rpc);
  a.nwork = (int)(cols * a.chunks);
  const int gx = std::min(a.nwork, cus);
  char nm[112];
  if (srx_prof_on()) snprintf(nm, sizeof(nm), "t9_bf16_kernel MxNxK=%lldx3x5184", (long long)N * H * W);
  SRX_LAUNCH_PROF(nm, 2.0 * N * H * W * 3.0 * 5184.0, t9_bf16_kernel, dim3((unsigned)gx), dim3(512), LDS_BYTES, srx_stream(stream), a);
  SRX_CHECK_LAUNCH("t9_bf16_kernel");
  return SRX_OK;
}
