// Whole C2f block (n = 1, hidden width 32) on the stride-4 maps as ONE kernel: cv1 (1x1 Cin->64) -> chunk -> Bottleneck
// (3x3 32->32, 3x3 32->32, optional shortcut) -> cat -> cv2 (1x1 96->64), every convolution with its folded BatchNorm bias
// and SiLU.  Reference: C2f.forward / Bottleneck.forward (nn/modules/block.py:227-249, 337-350) with Conv.forward_fuse
// (nn/modules/conv.py:53-55); yolov8-p2-repvgg.yaml layer 2 (Cin 64, shortcut) and layer 21 (Cin = Upsample(layer 18, 128 ch)
// ++ layer 2 (64 ch): nn.Upsample + Concat (conv.py:323) folded into cv1's gather, no shortcut) at scale s.
//
// Why one kernel: on the 160x160 maps these four convolutions are bandwidth bound one by one (1.3 KB of HBM traffic per pixel
// against 256 B when only the block's input and output move).  Why THIS shape: with 32-wide layers there are only 150 MACs per
// produced activation, so the SiLU epilogues (two quarter-rate transcendentals each) cost more issue cycles than the MFMAs, and a
// design that separates the phases by workgroup barriers leaves both pipes idle most of the time.  Here every WAVE owns a
// 16 x 8 output strip end to end and never meets another wave after the weights are staged:
//   A  cv1 -> y1 (channels 32..63) on the strip's 20 x 12 halo, x fragments loaded straight from global into the MFMA pixel
//      operand (a lane's 16 bytes are its pixel's 8 channels), y1 -> this wave's 15 KB of LDS (64-byte pixel rows, XOR swizzle);
//   B  m.cv1 3x3 on the 18 x 10 ring -> t, written OVER y1 (every y1 read is complete by then; the strip's centre y1 was
//      lifted into registers first);
//   C  m.cv2 3x3 on the 16 x 8 strip (+ y1 when the Bottleneck has a shortcut) -> y2, kept in registers;
//   D  cv1 -> y0 (channels 0..31) on the strip only, kept in registers;
//   E  cv2 on [y0 | y1 | y2]: an MFMA result lane holds channels {4q..4q+3, 16+4q..16+4q+3} of its pixel, which is a valid pixel
//      operand for the next 1x1 as long as the weight operand uses the same k order, so cv2's weight fragments are fetched with
//      that permutation (two 8-byte loads per lane from the FRAG1X1 image) and nothing goes back through LDS.
// Intermediates are rounded to the storage type exactly where the layer-by-layer path rounds.  Halo recompute: 548 MFMAs per
// 128 pixels against 448 without (+22 %).  Two waves per SIMD run independent strips, so one wave's epilogue VALU overlaps the
// other's MFMAs.  LDS: m.cv1 | m.cv2 weights 36 KB + biases + 8 x 15 KB; cv1 / cv2 weights (8-24 KB) are read from L2 per strip.
#include "common_hip.h"

namespace DY_NS {

#ifdef DYOLO_ABLATE
__device__ unsigned long long c2f_phase_cycles[8];  // A, B, C, D+E of block 0 / wave 0, summed over its strips (s_memtime)
#define C2F_STAMP(k)                                                         \
  do {                                                                       \
    const unsigned long long now = __builtin_amdgcn_s_memtime();             \
    if (blockIdx.x == 0 && tid == 0) c2f_phase_cycles[k] += now - stamp;     \
    stamp = now;                                                             \
  } while (0)
#else
#define C2F_STAMP(k)
#endif

struct C2fArgs {
  const void* x;    // direct source, NHWC (N, H, W, 64)
  const void* xlo;  // half-resolution source, NHWC (N, H/2, W/2, 32 * KC_LO), consumed through a 2x nearest upsample (or null)
  void* y;
  const void* w1;   // FRAG1X1 order, cout 64, cin 32 * KC_LO + 64 (upsampled channels first)
  const void* wm1;  // HALO3X3 order, cout 32, cin 32
  const void* wm2;
  const void* w2;   // FRAG1X1 order, cout 64, cin 96
  const float* bias;  // b1[64] | bm1[32] | bm2[32] | b2[64]
  int N, H, W, ldx, ldxlo, ldy, tilesX, tilesY, nStrips, shortcut;
  unsigned x_bytes, xlo_bytes, y_bytes;
  int dbg;  // timing probes, -DDYOLO_ABLATE builds only (DYOLO_C2F_DBG): 1 no global x loads, 2 no SiLU, 4 no stores
};

constexpr int kC2fWM1 = 0, kC2fWM2 = 18432, kC2fBias = 36864, kC2fWave = 37632, kC2fRegion = 240 * 64;
constexpr int kC2fSmem = kC2fWave + 8 * kC2fRegion;  // 160512

template <typename T, int KC_LO>
__global__ __launch_bounds__(512) void c2f_fused_kernel(const C2fArgs p) {
  constexpr int E = Elem<T>::EPC;  // 8
  constexpr int KC = KC_LO + 2;    // 32-channel k chunks of cv1
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const float* sbias = reinterpret_cast<const float*>(smem + kC2fBias);
  unsigned char* reg = smem + kC2fWave + wave * kC2fRegion;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(KC_LO ? p.xlo : p.x), 0, KC_LO ? p.xlo_bytes : p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  // cv1 / cv2 weight fragments are re-read from L2 / L1 per strip on purpose (no LDS left, and 80+ registers if kept), through
  // buffer descriptors so that a lane's address is one 32-bit offset (fragment and piece go into the instruction's immediate).
  // The offsets pass through an empty asm per strip: the loads must not be hoisted out of the strip loop and spilled to scratch.
  const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, (unsigned)(KC * 4 * 1024), 0x00020000);
  const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 12 * 1024, 0x00020000);
  unsigned w1off = lane * 16;                                   // plain fragment image: lane l holds bytes 16 l .. 16 l + 15
  unsigned w2off = ((lq >> 1) * 16 + lr) * 16 + (lq & 1) * 8;   // result-lane k order: the lane's first 8-byte piece
#define C2F_PIN_WEIGHT_POINTERS() asm volatile("" : "+v"(w1off), "+v"(w2off))

  {  // the two 3x3 weight sets + biases -> LDS, once; the only workgroup barrier of the kernel
    const u32x4* s2 = reinterpret_cast<const u32x4*>(p.wm1);
    const u32x4* s3 = reinterpret_cast<const u32x4*>(p.wm2);
    for (int i = tid; i < 1152; i += 512) reinterpret_cast<u32x4*>(smem + kC2fWM1)[i] = s2[i];
    for (int i = tid; i < 1152; i += 512) reinterpret_cast<u32x4*>(smem + kC2fWM2)[i] = s3[i];
    if (tid < 192) reinterpret_cast<float*>(smem + kC2fBias)[tid] = p.bias[tid];
  }
  __syncthreads();

  typedef __attribute__((ext_vector_type(4))) T t4;
  auto pack4 = [&](const f32x4 v) -> u32x2 {
    t4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
    return __builtin_bit_cast(u32x2, o);
  };
  auto silu4 = [&](const f32x4 a) -> f32x4 {
#ifdef DYOLO_ABLATE
    if (p.dbg & 2) return a;
#endif
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f32(a[e]);
    return v;
  };
  auto wave_sync = [&]() {  // LDS written by some lanes of this wave, read by others: order it (no other wave touches the region)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);  // also a phase boundary for the scheduler: nothing of the next phase is hoisted above it
  };
  // byte address of (slot, 16-byte part) in the wave's region; slots are 64-byte pixel rows, four to a 256-byte bank row.  The
  // part XOR (slot >> 1) & 3 makes a ds_read_b128 of 16 consecutive slots conflict-free in the instruction's own lane groups
  // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), whatever the first slot
  auto slot_addr = [&](int slot, int part) -> unsigned char* { return reg + slot * 64 + ((part ^ ((slot >> 1) & 3)) << 4); };
  // Lane coordinates for addresses that are computed where they are used.  They are lane constants, so the compiler would hoist
  // every such address (about 90 of them) out of the strip loop and spill them to scratch - and a scratch reload, like any
  // load, retires behind every older load in flight.  Passing the two values through an empty asm at each phase keeps the
  // arithmetic (a few VALU instructions per address) inside the phase.
  int lrv = lr, lqv = lq;
#define C2F_PIN_LANE() asm volatile("" : "+v"(lrv), "+v"(lqv))
  // cv1's virtual input [upsampled xlo | x]: per pixel the byte offsets of this lane's 16 bytes in both sources (kOob when the
  // pixel is outside the image: the buffer range check then returns zeros), k chunk c adds a constant that folds into the
  // instruction's immediate.  kOob + 6 * 64 + 16 stays below 2^32 and above any accepted view size.
  constexpr unsigned kOob = 0xfffffc00u;
  auto pix_off = [&](bool ok, int n, int gy, int gx, unsigned* lo, unsigned* hi) {
    *hi = ok ? (unsigned)((((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldx + lqv * E) * sizeof(T)) : kOob;
    if (KC_LO) *lo = ok ? (unsigned)((((size_t)(n * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (size_t)p.ldxlo + lqv * E) * sizeof(T)) : kOob;
  };
  auto load_x = [&](int c, unsigned lo, unsigned hi) -> u32x4 {
#ifdef DYOLO_ABLATE
    if (p.dbg & 1) return u32x4{lo, hi, (unsigned)c, 0x3c003c00u};
#endif
    if (KC_LO && c < KC_LO) return __builtin_amdgcn_raw_buffer_load_b128(lrs, lo + (unsigned)(c * 32 * sizeof(T)), 0, 0);
    return __builtin_amdgcn_raw_buffer_load_b128(xrs, hi + (unsigned)((c - KC_LO) * 32 * sizeof(T)), 0, 0);
  };

  // LDS addresses without per-read arithmetic.  A 3x3 tap (r, q) of a row fragment reads slot R * 20 + lr + q (R = row + r) with
  // swizzle key (2 (R & 1) + ((lr + q) >> 1)) & 3 (20 R / 2 = 10 R = 2 R mod 4): per lane one address per (q, R & 1), R * 1280 as
  // the instruction's immediate (16-byte pixel-operand reads, part lq).
  unsigned char* ra[3][2];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int m = 0; m < 2; ++m) ra[q][m] = reg + (lr + q) * 64 + ((lq ^ ((2 * m + ((lr + q) >> 1)) & 3)) << 4);
  // ---- the global-memory side, software-pipelined by hand.  vmcnt retires in order and a wave meets no other wave, so a load
  // issued where it is needed costs its whole latency (with two waves per SIMD there is little to hide it): every batch of loads
  // is issued one stage early, and sched_barrier pins the place.  A batch = the pixel-operand chunks of up to 8 fragments for k
  // chunk c plus cv1's two weight fragments for it (from L2); chunks c and c + 1 are in flight while c computes.
  // Phase A runs in four passes of row-aligned fragments (16 consecutive pixels of one halo row each, slot = 20 hr + hc):
  //   pass 0 / 1: the strip's own rows 0..3 / 4..7 (halo rows 2 + r, columns 2..17) - BOTH halves of cv1, so y0 and the
  //               strip's y1 are born in registers in result-lane order and x is read once;
  //   pass 2:     halo rows 0, 1, 10, 11, columns 2..17 (y1 only);   pass 3: columns 0, 1, 18, 19 of all 12 rows (48 px, y1 only).
  auto pass_pixel = [&](int h, int i, int* hr, int* hc) {
    if (h < 2) {
      *hr = 2 + h * 4 + i, *hc = 2 + lrv;
    } else if (h == 2) {
      *hr = i < 2 ? i : 8 + i, *hc = 2 + lrv;
    } else {
      const int idx = i * 16 + lrv;  // 48 live
      *hr = idx >> 2, *hc = (idx & 3) + ((idx & 2) ? 16 : 0);
    }
  };
  // Fragments i and i - 1 (i odd) of passes 0..2 are the halo rows 2k and 2k + 1: through the 2x upsample they read the SAME
  // half-resolution row, so for the upsampled source's k chunks the odd fragment reuses the even one's registers (half the loads)
  auto lo_twin = [&](int h, int c, int i) -> bool { return KC_LO > 0 && c < KC_LO && h < 3 && (i & 1); };
  auto pass_offsets = [&](int h, int n, int ty0, int tx0, bool valid, unsigned (&olo)[4], unsigned (&ohi)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int hr, hc;
      pass_pixel(h, i, &hr, &hc);
      const int gy = ty0 - 2 + hr, gx = tx0 - 2 + hc;
      pix_off(valid && !(h == 3 && i == 3) && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W, n, gy, gx, &olo[i], &ohi[i]);
    }
  };
  // one batch: k chunk c of the pass's fragments + cv1's weight fragments for it (4 when the pass also makes y0, else the y1 two)
  auto issue_pass = [&](int h, int c, const unsigned (&olo)[4], const unsigned (&ohi)[4], u32x4 (&av)[4], u32x4 (&bv)[4]) {
#pragma unroll
    for (int j = (h < 2 ? 0 : 2); j < 4; ++j) bv[j] = __builtin_amdgcn_raw_buffer_load_b128(w1rs, w1off + (unsigned)((c * 4 + j) * 1024), 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (!(h == 3 && i == 3) && !lo_twin(h, c, i)) av[i] = load_x(c, olo[i], ohi[i]);
  };
  // cv2's weight fragments of k chunk c in result-lane k order: lane (cout lr, quarter lq) takes k = {4 lq .., 16 + 4 lq ..}
  auto issue_w2 = [&](int c, u32x4 (&wq)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(w2rs, w2off + (unsigned)((c * 4 + j) * 1024), 0, 0);
      const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(w2rs, w2off + (unsigned)((c * 4 + j) * 1024 + 512), 0, 0);
      wq[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  };
  auto decode = [&](int strip, int* n, int* ty0, int* tx0) {
    const int ty = strip % p.tilesY;
    const int rest = strip / p.tilesY;
    *tx0 = (rest % p.tilesX) * 16, *ty0 = ty * 8, *n = rest / p.tilesX;
  };

  // strips of one image column are consecutive ids, consecutive ids go to the waves of one workgroup, and the workgroups of
  // one XCD (blockIdx mod 8) take a contiguous range: the 4 shared halo rows between vertical neighbours are L1 / L2 hits.
  const int G = (int)gridDim.x;
  const int lb = (G & 7) ? (int)blockIdx.x : ((int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3));
  int strip = lb * 8 + wave;
  u32x4 a[2][4], b[2][4];  // loop carried: chunks 0 and 1 of the NEXT phase-A pass, in flight
  {
    int n, ty0, tx0;
    decode(strip, &n, &ty0, &tx0);
    C2F_PIN_WEIGHT_POINTERS();
    unsigned olo[4], ohi[4];
    pass_offsets(0, n, ty0, tx0, strip < p.nStrips, olo, ohi);
    issue_pass(0, 0, olo, ohi, a[0], b[0]);
    issue_pass(0, 1, olo, ohi, a[1], b[1]);
  }
  for (; strip < p.nStrips; strip += G * 8) {
    int n, ty0, tx0;
    decode(strip, &n, &ty0, &tx0);
    C2F_PIN_WEIGHT_POINTERS();
    C2F_PIN_LANE();
#ifdef DYOLO_ABLATE
    unsigned long long stamp = __builtin_amdgcn_s_memtime();
#endif

    // ---- A: cv1 on the 20 x 12 halo -> y1 (LDS), and on the strip itself -> y0, y1 in result-lane order (registers: cv2's k
    // chunks 0 and 1, and the shortcut) ----
    u32x2 y0c[8][2], y1c[8][2];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      C2F_PIN_LANE();
      f32x4 acc[4][4];
      unsigned olo[4], ohi[4];
      if (KC > 2) pass_offsets(h, n, ty0, tx0, true, olo, ohi);  // chunks 2.. are issued from here
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = (h < 2 ? 0 : 2); j < 4; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + j * 16 + lq * 4);
#pragma unroll
      for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (!(h == 3 && i == 3)) {
#pragma unroll
            for (int j = (h < 2 ? 0 : 2); j < 4; ++j) acc[i][j] = Elem<T>::mma(b[c & 1][j], a[c & 1][lo_twin(h, c, i) ? i - 1 : i], acc[i][j]);
          }
        if (c + 2 < KC) issue_pass(h, c + 2, olo, ohi, a[c & 1], b[c & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (h < 3) {  // the next pass's first two chunks fly during this pass's epilogue
        unsigned plo[4], phi[4];
        pass_offsets(h + 1, n, ty0, tx0, true, plo, phi);
        issue_pass(h + 1, 0, plo, phi, a[0], b[0]);
        issue_pass(h + 1, 1, plo, phi, a[1], b[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (!(h == 3 && i == 3)) {
          int hr, hc;
          pass_pixel(h, i, &hr, &hc);
          const bool inside = (unsigned)(ty0 - 2 + hr) < (unsigned)p.H && (unsigned)(tx0 - 2 + hc) < (unsigned)p.W;
#pragma unroll
          for (int j = 2; j < 4; ++j) {
            u32x2 v = pack4(silu4(acc[i][j]));
            if (!inside) v = u32x2{0u, 0u};  // the 3x3 that follows pads y1 with zeros, not with SiLU(bias)
            if (h < 2) y1c[h * 4 + i][j - 2] = v;
            *reinterpret_cast<u32x2*>(slot_addr(hr * 20 + hc, (j - 2) * 2 + (lqv >> 1)) + (lqv & 1) * 8) = v;
          }
          if (h < 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) y0c[h * 4 + i][j] = pack4(silu4(acc[i][j]));
          }
        }
    }
    wave_sync();
    C2F_PIN_LANE();
    C2F_STAMP(0);

    // ---- B: m.cv1 3x3 on the 18 x 10 ring -> t, ring row R stored over y1 row R.  Three passes of 4 fragments: ring rows 0..3,
    // rows 4..7 (columns 0..15 each), then rows 8..9 and the columns 16..17 of all ten rows (20 px, two fragments).  A pass
    // overwrites y1 rows / columns that no later pass reads: row passes leave columns 16..19 alone and only touch rows below the
    // next pass's first row; the strip's own y1 (rows 2..9) is in registers since phase A ----
    {
#pragma unroll
      for (int hb = 0; hb < 3; ++hb) {
        C2F_PIN_LANE();
        // fragment i of pass 2: i < 2 -> ring row 8 + i; else lane -> (row (16 (i - 2) + lrv) >> 1, column 16 + (lrv & 1)), 20 px live
        const int xi0 = lrv >> 1, xi1 = lrv < 4 ? 8 + (lrv >> 1) : 9;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + 64 + j * 16 + lq * 4);
        u32x4 ta[2][4], tb[2][2];
        auto read_tap = [&](int tap, u32x4 (&av)[4], u32x4 (&bv)[2]) {
          const int r = tap / 3, q = tap % 3;
#pragma unroll
          for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const u32x4*>(smem + kC2fWM1 + ((tap * 2 + j) * 64 + lane) * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (hb < 2 || i < 2) {
              const int f = hb * 4 + i;  // ring row
              av[i] = *reinterpret_cast<const u32x4*>(ra[q][(f + r) & 1] + (f + r) * 1280);
            } else {
              av[i] = *reinterpret_cast<const u32x4*>(slot_addr(((i == 2 ? xi0 : xi1) + r) * 20 + 16 + (lrv & 1) + q, lqv));
            }
          }
        };
        read_tap(0, ta[0], tb[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) read_tap(tap + 1, ta[(tap + 1) & 1], tb[(tap + 1) & 1]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Elem<T>::mma(tb[tap & 1][j], ta[tap & 1][i], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);  // one tap of reads in flight, not nine
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool row_frag = hb < 2 || i < 2;
          const int trow = row_frag ? hb * 4 + i : (i == 2 ? xi0 : xi1), tcol = row_frag ? lrv : 16 + (lrv & 1);
          const bool inside = (unsigned)(ty0 - 1 + trow) < (unsigned)p.H && (unsigned)(tx0 - 1 + tcol) < (unsigned)p.W;
          if (row_frag || i == 2 || lrv < 4) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              u32x2 v = pack4(silu4(acc[i][j]));  // every y1 read of this pass has returned: its MFMA consumed it
              if (!inside) v = u32x2{0u, 0u};
              *reinterpret_cast<u32x2*>(slot_addr(trow * 20 + tcol, j * 2 + (lqv >> 1)) + (lqv & 1) * 8) = v;
            }
          }
        }
      }
    }
    wave_sync();
    C2F_STAMP(1);

    // ---- C: m.cv2 3x3 on the strip (+ y1 with a shortcut) -> y2 in result-lane order ----
    u32x2 y2c[8][2];
    u32x4 wq[2][4];  // cv2 weight chunks in flight
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {  // four rows at a time: registers
      f32x4 acc[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[r][j] = *reinterpret_cast<const f32x4*>(sbias + 96 + j * 16 + lq * 4);
      {
        u32x4 ta[2][4], tb[2][2];
        auto read_tap = [&](int tap, u32x4 (&av)[4], u32x4 (&bv)[2]) {
          const int r = tap / 3, q = tap % 3;
#pragma unroll
          for (int j = 0; j < 2; ++j) bv[j] = *reinterpret_cast<const u32x4*>(smem + kC2fWM2 + ((tap * 2 + j) * 64 + lane) * 16);
#pragma unroll
          for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const u32x4*>(ra[q][(hc * 4 + i + r) & 1] + (hc * 4 + i + r) * 1280);
        };
        read_tap(0, ta[0], tb[0]);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          if (tap + 1 < 9) read_tap(tap + 1, ta[(tap + 1) & 1], tb[(tap + 1) & 1]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Elem<T>::mma(tb[tap & 1][j], ta[tap & 1][i], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (hc == 1) {  // cv2's first weight chunk flies during this epilogue
        issue_w2(0, wq[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = hc * 4 + i;
          f32x4 v = silu4(acc[i][j]);
          if (p.shortcut) {  // x + cv2(cv1(x)), added after the activation (block.py:348-350)
            const t4 rr = __builtin_bit_cast(t4, y1c[row][j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rr[e]);
          }
          y2c[row][j] = pack4(v);
        }
    }
    wave_sync();
    C2F_PIN_LANE();
    C2F_STAMP(2);

    // ---- E, four rows at a time: cv2 1x1 on [y0 | y1 | y2] -> global.  A result lane holds channels {4q..4q+3, 16+4q..16+4q+3} of
    // its pixel; cv2's weight lane (cout lr, quarter lq) is fetched with the same k order (two 8-byte pieces of the FRAG1X1
    // image), so the three inputs never go back through LDS. ----
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + 128 + j * 16 + lq * 4);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c + 1 < 3) issue_w2(c + 1, wq[(c + 1) & 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = half * 4 + i;
          const u32x2 s0 = c == 0 ? y0c[r][0] : (c == 1 ? y1c[r][0] : y2c[r][0]);
          const u32x2 s1 = c == 0 ? y0c[r][1] : (c == 1 ? y1c[r][1] : y2c[r][1]);
          const u32x4 av = u32x4{s0[0], s0[1], s1[0], s1[1]};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Elem<T>::mma(wq[c & 1][j], av, acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the next batch of loads goes out BEFORE this half's stores (vmcnt retires in order: a load behind a store waits for it)
      if (half == 0) {
        issue_w2(0, wq[0]);
      } else {
        int nn, nty0, ntx0;
        const int next = strip + G * 8;
        decode(next, &nn, &nty0, &ntx0);
        unsigned olo[4], ohi[4];
        pass_offsets(0, nn, nty0, ntx0, next < p.nStrips, olo, ohi);
        issue_pass(0, 0, olo, ohi, a[0], b[0]);
        issue_pass(0, 1, olo, ohi, a[1], b[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // Result lanes hold 8-byte pieces of 16 different pixel rows: stored like that, every lane is its own L1 request (measured:
      // the 32 stores of a strip cost as much as all its MFMAs).  Through the wave's region (idle since phase C; 64 px x 128 B,
      // 16-byte chunk c of pixel px at chunk c ^ (px & 7)) they leave as whole 128-byte pixel rows, eight lanes per row.
      if (half) wave_sync();  // the first half's reads of the region are done
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int px = i * 16 + lrv, ch = j * 2 + (lqv >> 1);
          *reinterpret_cast<u32x2*>(reg + px * 128 + ((ch ^ (px & 7)) << 4) + (lqv & 1) * 8) = pack4(silu4(acc[i][j]));
        }
      wave_sync();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = k * 64 + lane;
        const int px = idx >> 3, ch = idx & 7;
        const int gy = ty0 + half * 4 + (px >> 4), gx = tx0 + (px & 15);
        const u32x4 val = *reinterpret_cast<const u32x4*>(reg + px * 128 + ((ch ^ (px & 7)) << 4));
        const unsigned off = (gy < p.H && gx < p.W) ? (unsigned)((((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldy + ch * E) * sizeof(T)) : 0xfffffff0u;
#ifdef DYOLO_ABLATE
        if ((p.dbg & 4) && val[0] != 0x12345u) continue;
#endif
        __builtin_amdgcn_raw_buffer_store_b128(val, yrs, off, 0, 0);
      }
    }
    wave_sync();  // the next strip's phase A writes the region
    C2F_STAMP(3);
  }
}

template <typename T, int KC_LO>
static void c2f_launch(const C2fArgs& a, int grid, hipStream_t st) {
  static const hipError_t once = hipFuncSetAttribute((const void*)c2f_fused_kernel<T, KC_LO>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)once;
  hipLaunchKernelGGL((c2f_fused_kernel<T, KC_LO>), dim3((unsigned)grid), dim3(512), kC2fSmem, st, a);
}

}  // namespace DY_NS

using namespace DY_NS;

#ifndef DYOLO_L2E_BUILD
#ifdef DYOLO_ABLATE
extern "C" int32_t dy_c2f_debug_phase_cycles(unsigned long long* out8, int32_t reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(c2f_phase_cycles), 64) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(c2f_phase_cycles), z, 64) != hipSuccess) return -1;
  }
  return 0;
}
#endif

extern "C" int32_t dy_c2f_fused_supported(int32_t cin, int32_t cin_lo, int32_t hidden, int32_t cout, int32_t n_bottlenecks, int32_t dtype) {
  return (cin - cin_lo == 64 && (cin_lo == 0 || cin_lo == 128) && hidden == 32 && cout == 64 && n_bottlenecks == 1 && (dtype == DY_BF16 || dtype == DY_F16)) ? 1 : 0;
}

namespace dy_l2e {
int32_t c2f_entry(const dy_c2f_desc* d, dy_stream_t stream);
}
namespace dy {
int32_t c2f_entry(const dy_c2f_desc* d, dy_stream_t stream);
}
extern "C" int32_t dy_c2f_fused(const dy_c2f_desc* d, dy_stream_t stream) {
  return (d != nullptr && d->act_l2e) ? dy_l2e::c2f_entry(d, stream) : dy::c2f_entry(d, stream);
}
#endif

namespace DY_NS {
int32_t c2f_entry(const dy_c2f_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->x && d->y && d->w_cv1 && d->w_m_cv1 && d->w_m_cv2 && d->w_cv2 && d->bias, DY_ERR_INVALID_ARG, "dy_c2f_fused: null pointer");
  DY_REQUIRE(dy_c2f_fused_supported(d->cin, d->cin_lo, d->hidden, d->cout, 1, d->dtype), DY_ERR_UNSUPPORTED,
             "dy_c2f_fused: built for 64 direct input channels (+ 128 upsampled), hidden 32, cout 64, one Bottleneck, 16-bit storage (got cin %d of which %d upsampled / %d / %d dtype %d)",
             d->cin, d->cin_lo, d->hidden, d->cout, d->dtype);
  const int cd = d->cin - d->cin_lo;
  DY_REQUIRE(d->batch > 0 && d->h > 0 && d->w > 0 && d->ld_x >= cd && d->ld_y >= d->cout && d->ld_x % 8 == 0 && d->ld_y % 8 == 0, DY_ERR_INVALID_ARG,
             "dy_c2f_fused: bad dims / pitches");
  DY_REQUIRE(aligned16(d->x) && aligned16(d->y) && aligned16(d->w_cv1) && aligned16(d->w_m_cv1) && aligned16(d->w_m_cv2) && aligned16(d->w_cv2) && aligned16(d->bias),
             DY_ERR_INVALID_ARG, "dy_c2f_fused: views must be 16-byte aligned");
  const long long xb = (long long)d->batch * d->h * d->w * d->ld_x * 2, yb = (long long)d->batch * d->h * d->w * d->ld_y * 2;
  long long lb = 0;
  if (d->cin_lo) {
    DY_REQUIRE(d->x_lo && aligned16(d->x_lo) && d->ld_x_lo >= d->cin_lo && d->ld_x_lo % 8 == 0 && d->h % 2 == 0 && d->w % 2 == 0, DY_ERR_INVALID_ARG,
               "dy_c2f_fused: the upsampled source needs a 16-byte aligned view, a pitch >= its channels and an even output size");
    lb = (long long)d->batch * (d->h / 2) * (d->w / 2) * d->ld_x_lo * 2;
  }
  DY_REQUIRE(xb < (1ll << 32) - 1024 && yb < (1ll << 32) - 1024 && lb < (1ll << 32) - 1024, DY_ERR_UNSUPPORTED, "dy_c2f_fused: views exceed 4 GiB (buffer descriptor range)");
  C2fArgs a{};
  a.x = d->x, a.xlo = d->x_lo, a.y = d->y, a.w1 = d->w_cv1, a.wm1 = d->w_m_cv1, a.wm2 = d->w_m_cv2, a.w2 = d->w_cv2, a.bias = d->bias;
  a.N = d->batch, a.H = d->h, a.W = d->w, a.ldx = d->ld_x, a.ldxlo = d->ld_x_lo, a.ldy = d->ld_y, a.shortcut = d->shortcut;
  a.tilesX = (d->w + 15) / 16, a.tilesY = (d->h + 7) / 8;
  a.nStrips = d->batch * a.tilesY * a.tilesX;
  a.x_bytes = (unsigned)xb, a.xlo_bytes = (unsigned)lb, a.y_bytes = (unsigned)yb;
  a.dbg = dy_ablate("DYOLO_C2F_DBG");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int grid = 256;
  if (grid > (a.nStrips + 7) / 8) grid = (a.nStrips + 7) / 8;
  if (d->dtype == DY_BF16) {
    if (d->cin_lo) c2f_launch<bf16_t, 4>(a, grid, st); else c2f_launch<bf16_t, 0>(a, grid, st);
  } else {
    if (d->cin_lo) c2f_launch<f16_t, 4>(a, grid, st); else c2f_launch<f16_t, 0>(a, grid, st);
  }
  return check_launch("c2f_fused_kernel");
}
}  // namespace DY_NS
