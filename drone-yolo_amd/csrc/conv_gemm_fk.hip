// Flat-K implicit-GEMM convolution with LDS-DMA staging for the channel counts the tap-aligned kernels cannot tile, and the fp8
// kernel of BASELINE config 5 on the block-scaled MFMA.
//
// conv_gemm_glds.hip cuts K = (r, q, c) into steps of 128 bytes of ONE tap, so it needs Cin % 64 == 0 (16-bit) and Cout % 64 == 0.
// The widths of the n / m / x scales of the model YAMLs are multiples of 8 or 16 only (scale x: 80, 160, 320, 640 and 5 c / 8 c
// concatenations); those layers ran on the generic register-staged kernel (conv_igemm.hip, ~310 TFLOP/s).  Here K is walked FLAT:
// step s covers the 16-byte chunks 8 s .. 8 s + 7 of the (r, q, c) axis, whichever taps they fall into (93-98 % of the MFMA K
// slots carry data for Cin = 80 m, against 62-83 % when every tap is padded to whole steps).  The price is a per-lane (tap, channel)
// state in the loader: a lane's chunk column is fixed over the K loop (cc = (lane & 7) ^ swizzle(row)), so its state advances by
// exactly one step per step -- an add and at most two wrap-arounds -- and only two such states exist per lane (the swizzle of a
// piece's rows depends on the piece's parity only).  Zero padding and the K tail are out-of-range buffer offsets (the range check
// of buffer_load ... lds feeds zeros); weights are DY_WLAYOUT_ROWS rows, whose k order IS the flat axis.
//
// Tile: (WM x MFR x 16) pixels x (WN x NFR x 16) couts, waves as WM x WN, wave tile (MFR x 16) x (NFR x 16): NFR = 5 makes 80 / 160
// wide tiles for the x scale, masked on the last cout tile for anything else.  LDS image, swizzle, MFMA operand order (weights as
// the A operand: a lane holds 4 consecutive couts of one pixel) and the transposing epilogue follow conv_gemm_glds.hip.
//
// DY_FP8 (e4m3fn): a K-step is 128 channels = ONE v_mfma_f32_16x16x128_f8f6f4 per fragment pair (unscaled form: E8M0 scale 2^0 on
// both operands; the per-output-channel weight scales and the network-wide activation scale multiply the fp32 accumulator in the
// epilogue, include/dyolo.h).  That opcode retires 4x the K of v_mfma_f32_16x16x32_fp8_fp8 in 2x its cycles: the 5 PFLOP/s rate.
// Lane (lr, lq) of a 16x16x128 operand holds row lr and k = 32 lq .. 32 lq + 31 (tools/mx_probe.hip checks the map on the device
// with exact integer data); A and B only have to AGREE on which k a (lane quarter, byte) pair means, so a lane's 32 bytes are the
// two 16-byte chunks lq and 4 + lq of the 128-byte row -- the same two ds_read_b128 the 16-bit path issues for its two 32-deep
// sub-steps.  The output type is a template parameter: fp8 -> fp8, fp8 -> f16 (a precision-critical tail behind an fp8 trunk) and
// 16-bit -> fp8 (the hand-over into an fp8 trunk) besides the plain same-type form (dy_conv_desc.y_dtype1).
//
// Reference semantics: nn/modules/conv.py:37-55 (Conv), block.py:337-350 (Bottleneck residual), block.py:1480-1490 (RepVGGBlock,
// folded), head.py:43-57 (Detect convs).
#include "common_hip.h"
#include <type_traits>

namespace DY_NS {

typedef __attribute__((ext_vector_type(8))) int i32x8;

struct FkArgs {
  const void* x;
  const void* x2;
  const void* w;
  const float* bias;
  const float* wscale;  // fp8 input: act_scale * weight scale per output channel; else nullptr
  const void* res;
  void* y;
  int H, W, Cin, ldx, ldx2, split;  // channels [0, split) from x, the rest from x2 (1x1 only, split a multiple of the K-step)
  int HB, WB;                        // buffer dims of x (H/2, W/2 with up2x)
  int Ho, Wo, Cout, ldy, ldres;
  int ks, stride, pad;
  int Kpad, M, HoWo, up2x;
  int act;
  int tilesN, nblk;
  int cout_pad;
  unsigned xb, x2b, wb;  // bytes addressable from x / x2 / w (buffer descriptor ranges)
  float res_scale;        // fp8 residual: real value of one quantum (act_scale); 1 otherwise
  float out_scale;        // fp8 output: 1 / act_scale; 1 otherwise
  FastDiv dCin;           // exact division by Cin (tap table)
  int dbg;                // ablate build only (DYOLO_FK_DBG): 1 no MFMAs, 2 no LDS-DMA after the first step, 3 no LDS fragment reads, 4 no epilogue stores
  double* stats;          // optional (dy_conv_desc.bn_stats, STATS kernels): a dy_bn_train_fwd workspace, row block i fills (or, stats_atomic, adds into) slot 1 + i
  int stats_atomic;       // more row blocks than slots: block i ADDS into slot 1 + i % kStatSlots (the host zeroed them)
};

template <typename T> struct FkIsFp8 { static constexpr bool v = std::is_same<T, fp8_t>::value; };

// pack 4 fp32 -> 4 OT at `sp` (8 bytes for 16-bit, 4 bytes for fp8)
template <typename OT>
__device__ __forceinline__ void fk_store4(unsigned char* sp, const float (&v)[4]) {
  if constexpr (FkIsFp8<OT>::v) {
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = __builtin_fminf(__builtin_fmaxf(v[e], -448.f), 448.f);
    int r = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], r, true);
    *reinterpret_cast<int*>(sp) = r;
  } else {
    typedef __attribute__((ext_vector_type(4))) OT t4;
    t4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Elem<OT>::from_f32(v[e]);
    *reinterpret_cast<u32x2*>(sp) = __builtin_bit_cast(u32x2, o);
  }
}

// RES: the call has a residual (Bottleneck shortcut, block.py:348-350).  The first form read it per MFMA result lane — 8 (4) bytes of a
// lane's own pixel row, one L1 request per lane — and ran the residual layers 15-20 % slower than the plain ones.  With RES the
// transposing scratch holds fp32, and the store phase, where a lane owns a 16-byte chunk of a pixel row, loads the residual as the same
// coalesced 16-byte chunk, adds in fp32 and rounds ONCE — the arithmetic of the reference expression x + cv2(cv1(x)).
// TABN: entries of the tap table (3x3 only).  The first version kept a per-lane (channel, kernel row, kernel column) state and advanced
// it every K-step with adds, compares and selects: ~36 vector instructions per step for the two states, ~11 more per staged piece for
// the padding test — the loop issued ~110 vector instructions beside 40 MFMAs and was bound by exactly that (the kernel did not care
// about tile shapes or wave counts, and the ablate build, which adds a dozen instructions, ran at HALF the speed).  Now the
// workgroup writes, once, one word per 16-byte K chunk into LDS — (byte offset of the chunk's tap and channel) << 4 | tap, tap 9 for
// the zero-padded K tail —, every staged row carries a 9-bit mask of the taps that fall inside the image, and a piece's source
// offset is (row base + table offset) | (mask bit - 1): an out-of-range offset wherever the tap is padding.
// STATS (r04; training forward in front of a train-mode BatchNorm): per-channel sum / sum of squares of the STORED outputs of the tile,
// as in conv_gemm_glds.hip.
// W32 (ablate build only, DYOLO_FK_W32=1): the same tile and loop on v_mfma_f32_32x32x16 — a wave tile of 64 x 64 is 2 x 2 fragments of 32
// rows, a 128-byte K-step four 16-deep MFMA steps (lane l: row l % 32, chunk 2 t + l / 32) — for the A/B of the MFMA shape inside a production kernel.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <typename T, typename OT, int MFR, int NFR, int WM, int WN, bool RES = false, int TABN = 768, bool STATS = false, bool W32 = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_gemm_fk_kernel(const FkArgs p) {
  constexpr bool MX = FkIsFp8<T>::v;
  constexpr bool SP = std::is_same<T, f16x2_t>::value;  // DY_F16X2: split float16 pairs, three MFMAs per staged fragment pair (header)
  static_assert(!STATS || (!RES && !MX && sizeof(OT) == 2), "STATS: 16-bit storage, no residual");
  static_assert(!SP || std::is_same<OT, f16x2_t>::value || std::is_same<OT, float>::value, "split input: split or fp32 output");
  constexpr int EPC = Elem<T>::EPC;
  constexpr int NW = WM * WN;
  constexpr int BM = WM * MFR * 16, BN = WN * NFR * 16;
  constexpr int BKE = 8 * EPC;  // K elements per step (128 bytes)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW;                  // A pieces per wave per step
  constexpr int PBT = BN / 8;                      // W pieces per step (all waves)
  constexpr int PB = (PBT + NW - 1) / NW;          // ... per wave (the last ones may not exist: wave-uniform skip)
  static_assert(PA >= 1 && BM % (8 * NW) == 0, "every wave stages whole A pieces");
  constexpr int OES = (int)sizeof(OT);
  static_assert(!RES || std::is_same<T, OT>::value, "the residual has the input's type; built for same-type outputs");
  constexpr int SES = RES ? 4 : OES;               // scratch element: the output type, or fp32 in front of a residual add
  constexpr int EP_PITCH = NFR * 16 * SES + 16;    // one pixel row of the wave tile + a 16-byte skew
  constexpr int CPP = NFR * OES;                   // 16-byte chunks per pixel row of the wave tile
  constexpr int NH = (MFR * 16 * EP_PITCH * NW <= 2 * STAGE) ? 1 : ((MFR * 8 * EP_PITCH * NW <= 2 * STAGE) ? 2 : 4);  // epilogue passes per wave tile
  constexpr int PXP = MFR * 16 / NH;               // pixels per wave and pass
  constexpr int MPP = MFR / NH;                    // pixel fragments per pass
  static_assert(MFR % NH == 0 && PXP * EP_PITCH * NW <= 2 * STAGE, "epilogue scratch must fit the stage memory");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + TABN * 4];
  unsigned* const tab = reinterpret_cast<unsigned*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lq = lane >> 4, lr = lane & 15;
  const unsigned L = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int tileN = (int)(L % (unsigned)p.tilesN);
  const int tileM = (int)(L / (unsigned)p.tilesN);
  constexpr unsigned ES = (unsigned)sizeof(T);
  constexpr unsigned kOob = 0xfffffff0u;

  // ---- buffer descriptors: the base is moved back by `pre` so that the (negative) tap (0, 0) offset of a border pixel stays >= 0 ----
  const unsigned pre1 = (unsigned)((p.pad * p.WB + p.pad) * p.ldx) * ES, pre2 = (unsigned)((p.pad * p.W + p.pad) * p.ldx2) * ES;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre1, 0, p.xb + pre1, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x2)) - pre2, 0, p.x2b + pre2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wb, 0x00020000);

  // ---- per-lane gather bookkeeping (constant over the K loop) ----
  const int prow = lane >> 3;
  const int nsteps = p.Kpad / BKE;
  bool a_ok[PA];
  unsigned a_mask[PA];  // 3x3: bit t = tap t of this row reads inside the image
  unsigned av1[PA], av2[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * 8 + prow;
    const int m = tileM * BM + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int n = mm / p.HoWo;
    const int rem = mm - n * p.HoWo;
    const int ho = rem / p.Wo;
    const int wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_ok[i] = ok;
    // tap (r, q) is inside the image iff row hi0 + r and column wi0 + q are: three row bits x three column bits
    unsigned rm = 0, cm = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      rm |= ((unsigned)(hi0 + t) < (unsigned)p.H) ? (1u << t) : 0u;
      cm |= ((unsigned)(wi0 + t) < (unsigned)p.W) ? (1u << t) : 0u;
    }
    a_mask[i] = ok ? (((rm & 1u) ? cm : 0u) | ((rm & 2u) ? cm << 3 : 0u) | ((rm & 4u) ? cm << 6 : 0u)) : 0u;
    const int hb = p.up2x ? (hi0 >> 1) : hi0, wb_ = p.up2x ? (wi0 >> 1) : wi0;
    av1[i] = (unsigned)(((n * p.HB + hb) * p.WB + wb_) * p.ldx) * ES + pre1;
    av2[i] = (unsigned)(((n * p.H + hi0) * p.W + wi0) * p.ldx2) * ES + pre2;
  }
  unsigned bv[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = (wave * PB + j) * 8 + prow;
    bv[j] = (unsigned)((tileN * BN + row) * p.Kpad + (((lane & 7) ^ ((row >> 1) & 7)) * EPC)) * ES;  // rows beyond cout_pad: past p.wb -> zeros
  }
  // chunk column of this lane in an even / odd piece: (lane & 7) ^ ((row >> 1) & 7), row = 8 piece + prow  ->  bit 2 = piece parity
  const int cc0 = (lane & 7) ^ (prow >> 1), cc1 = cc0 ^ 4;
  unsigned ent[2] = {0u, 0u};  // table words of the NEXT step to issue, for the even / odd pieces
  if (p.ks != 1) {
    const int nf = nsteps * 8;
    for (int f = tid; f < nf; f += NW * 64) {
      const unsigned k = (unsigned)f * EPC;
      const unsigned tap = fastdiv(k, p.dCin);
      const unsigned c = k - tap * (unsigned)p.Cin;
      const unsigned kr = (tap * 11u) >> 5, kq = tap - 3u * kr;
      tab[f] = tap < 9u ? (((kr * (unsigned)p.W + kq) * (unsigned)p.ldx + c) * ES) << 4 | tap : 9u;
    }
    __syncthreads();
    ent[0] = tab[cc0];
    ent[1] = tab[cc1];
  }
  int kstep = 0;

  auto issue = [&](int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
    if (p.ks == 1) {
      // single tap: chunk f = 8 s + cc holds channels f EPC ..; the source (x through a fused 2x upsample, or x2) is wave-uniform per step
      const int kbase = kstep * BKE;
      const bool from_x = kbase < p.split;
      const int c_e = kbase + cc0 * EPC, c_o = kbase + cc1 * EPC;
      const unsigned oe = c_e < p.Cin ? (unsigned)(c_e - (from_x ? 0 : p.split)) * ES : kOob;
      const unsigned oo = c_o < p.Cin ? (unsigned)(c_o - (from_x ? 0 : p.split)) * ES : kOob;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const unsigned o = ((wave * PA + i) & 1) ? oo : oe;
        const bool ok = a_ok[i] && o != kOob;
        const unsigned off = ok ? (from_x ? av1[i] : av2[i]) + o : kOob;
        if (from_x)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)off, 0, 0, 0);
      }
    } else {
      const unsigned toff[2] = {ent[0] >> 4, ent[1] >> 4}, tp[2] = {ent[0] & 15u, ent[1] & 15u};
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int v = (wave * PA + i) & 1;
        const unsigned bit = __builtin_amdgcn_ubfe(a_mask[i], tp[v], 1u);
        const unsigned off = (av1[i] + toff[v]) | (bit - 1u);  // padding / K tail: 0xffffffff, past the descriptor's range -> zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)off, 0, 0, 0);
      }
      const int nx = kstep + 1 < nsteps ? kstep + 1 : kstep;  // (the last prefetch is not used)
      ent[0] = tab[nx * 8 + cc0];
      ent[1] = tab[nx * 8 + cc1];
    }
    const unsigned soffw = (unsigned)(kstep * BKE) * ES;
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (wave * PB + j < PBT)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(sb + (wave * PB + j) * 1024), 16, (int)bv[j], (int)soffw, 0, 0);
    ++kstep;
  };

  // ---- accumulators ----
  f32x4 acc[NFR][MFR];
#pragma unroll
  for (int j = 0; j < NFR; ++j)
#pragma unroll
    for (int i = 0; i < MFR; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // split float16: the x_lo' w_hi products (worth 2^-11 of the others) collect in their own accumulators and join in the epilogue — in the
  // same accumulator they needed w_hi 2^-11 formed per fragment and K-step (16 v_pk_mul_f16 beside 24 MFMAs)
  // (+1.6 % on the s-scale pass; tiles of more than 8 fragments per wave keep the one-accumulator form: 80 more registers halved the
  // 128 x 160 tile's occupancy, -15 % on the x scale)
  constexpr bool SPL = SP && MFR * NFR <= 8;
  f32x4 accl[SPL ? NFR : 1][SPL ? MFR : 1];
  if constexpr (SPL) {
#pragma unroll
    for (int j = 0; j < NFR; ++j)
#pragma unroll
      for (int i = 0; i < MFR; ++i) accl[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x16 acc32[W32 ? NFR / 2 : 1][W32 ? MFR / 2 : 1];
  if constexpr (W32) {
    static_assert(!W32 || (!RES && !STATS && sizeof(T) == 2 && sizeof(OT) == 2 && MFR % 2 == 0 && NFR % 2 == 0), "W32: plain 16-bit tiles of whole 32-row fragments");
#pragma unroll
    for (int j = 0; j < NFR / 2; ++j)
#pragma unroll
      for (int i = 0; i < MFR / 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[j][i][e] = 0.f;
  }
  const int swz = lr >> 1;
  auto compute = [&](int stage) {
    if constexpr (W32) {
      const int r32 = lane & 31, kh = lane >> 5, sw32 = (r32 >> 1) & 7;
      const unsigned char* sa32 = smem + stage * STAGE + (wm * MFR * 16 + r32) * 128;
      const unsigned char* sb32 = smem + stage * STAGE + A_BYTES + (wn * NFR * 16 + r32) * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int slot = ((2 * t + kh) ^ sw32) * 16;
        u32x4 a[MFR / 2], b[NFR / 2];
#pragma unroll
        for (int j = 0; j < NFR / 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(sb32 + j * 32 * 128 + slot);
#pragma unroll
        for (int i = 0; i < MFR / 2; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa32 + i * 32 * 128 + slot);
#pragma unroll
        for (int j = 0; j < NFR / 2; ++j)
#pragma unroll
          for (int i = 0; i < MFR / 2; ++i) {
            if constexpr (std::is_same<T, f16_t>::value)
              acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[j]), __builtin_bit_cast(f16x8, a[i]), acc32[j][i], 0, 0, 0);
            else
              acc32[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b[j]), __builtin_bit_cast(bf16x8, a[i]), acc32[j][i], 0, 0, 0);
          }
      }
      return;
    }
    const unsigned char* sa = smem + stage * STAGE + (wm * MFR * 16 + lr) * 128;
    const unsigned char* sb = smem + stage * STAGE + A_BYTES + (wn * NFR * 16 + lr) * 128;
#ifdef DYOLO_FK_KDBG  // (in-kernel timing probes: `make ABLATE=1 KDBG=1`; they move the accumulators of the MFMA loop through VGPRs — 64 v_accvgpr moves per K-step — so a build that carries them cannot time the loop itself)
    if (p.dbg == 3) {  // no fragment reads: MFMAs on whatever the registers hold
      sa = smem + (lr & 1) * 128;
      sb = smem + (lr & 1) * 128 + 512;
    }
    if (p.dbg == 1) {  // no MFMAs: the reads only (kept alive through the accumulator)
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + ((lq ^ swz) * 16)), u = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + (((4 + lq) ^ swz) * 16));
        acc[j][0][0] += __uint_as_float(t[0] ^ u[1]);
      }
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        const u32x4 t = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + ((lq ^ swz) * 16)), u = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + (((4 + lq) ^ swz) * 16));
        acc[0][i][1] += __uint_as_float(t[2] ^ u[3]);
      }
      return;
    }
#endif
    if constexpr (MX) {
      const int s0 = ((0 + lq) ^ swz) * 16, s1 = ((4 + lq) ^ swz) * 16;
      i32x8 a[MFR], b[NFR];
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + s0), hi = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + s1);
        b[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int i = 0; i < MFR; ++i) {
        const u32x4 lo = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + s0), hi = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + s1);
        a[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
      }
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int i = 0; i < MFR; ++i) {
          // the SCALED opcode with E8M0 127 (2^0) on both operands: bit-identical results to the unscaled form, and 3-8 % more MFMAs per
          // second in a bare issue loop (tools/mx_probe.hip: 4.88-4.99 against 4.51-4.85 PFLOP/s)
          acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a[i], acc[j][i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    } else if constexpr (SP) {
      // a 128-byte row = four (hi, lo) chunk pairs of 8 channels: lane quarter lq takes pair lq.  x w = x_hi w_hi + x_hi w_lo + 2^-11 x_lo' w_hi
      // (x_lo' is stored times 2^11; lo x lo, 2^-22 relative, is dropped): fp32-grade products from three 16-bit MFMAs, and per MFMA
      // two thirds of the LDS reads and DMA bytes of the plain float16 loop
      // WHICH pair a lane quarter takes is free (both operands use the same order), and it decides the bank conflicts: a ds_read_b128 is
      // served in groups of 16 lanes — rows {0..3, 12..15} of quarter q with rows {4..11} of quarter q ^ 1 — whose same-parity rows XOR their
      // chunk with {0, 1, 6, 7} and {2, 3, 4, 5} (the row swizzle lr >> 1): the two sets of positions are disjoint only when the quarters' chunks
      // differ by 6, i.e. their pairs by 3.  Pairs 0, 3, 1, 2 for quarters 0 .. 3 (pair = lq for both: every read a 2-way conflict,
      // SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles, profiles/r05_pmc_split_fk.txt).
      const int pr = (0x9C >> (2 * lq)) & 3;
      const int sh = ((2 * pr) ^ swz) * 16, sl = ((2 * pr + 1) ^ swz) * 16;
      u32x4 ah[MFR], al[MFR];
#pragma unroll
      for (int i = 0; i < MFR; ++i) ah[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + sh), al[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + sl);
#pragma unroll
      for (int j = 0; j < NFR; ++j) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + sh), bl = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + sl);
        if constexpr (SPL) {
#pragma unroll
          for (int i = 0; i < MFR; ++i) {
            acc[j][i] = Elem<f16_t>::mma(bh, ah[i], acc[j][i]);
            acc[j][i] = Elem<f16_t>::mma(bl, ah[i], acc[j][i]);
            accl[j][i] = Elem<f16_t>::mma(bh, al[i], accl[j][i]);
          }
        } else {
          const f16x8 bsv = __builtin_bit_cast(f16x8, bh) * (f16_t)kSplitInv;  // v_pk_mul_f16 x 4: exact (a power of two), rows are scaled to >= 2^13 at the top
          const u32x4 bs = __builtin_bit_cast(u32x4, bsv);
#pragma unroll
          for (int i = 0; i < MFR; ++i) {
            acc[j][i] = Elem<f16_t>::mma(bh, ah[i], acc[j][i]);
            acc[j][i] = Elem<f16_t>::mma(bl, ah[i], acc[j][i]);
            acc[j][i] = Elem<f16_t>::mma(bs, al[i], acc[j][i]);
          }
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int slot = ((s * 4 + lq) ^ swz) * 16;
        u32x4 a[MFR], b[NFR];
#pragma unroll
        for (int j = 0; j < NFR; ++j) b[j] = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + slot);
#pragma unroll
        for (int i = 0; i < MFR; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + slot);
#pragma unroll
        for (int j = 0; j < NFR; ++j)
#pragma unroll
          for (int i = 0; i < MFR; ++i) acc[j][i] = Elem<T>::mma(b[j], a[i], acc[j][i]);
      }
    }
  };

  // ---- main loop: one barrier per K-step, the next step's DMA runs under this step's MFMAs ----
  issue(0);
  for (int s = 0; s < nsteps; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of step s has landed
    __syncthreads();                                   // publishes stage s & 1; everyone is done with the other one
#ifdef DYOLO_FK_KDBG
    if (s + 1 < nsteps && p.dbg != 2) issue((s + 1) & 1);
#else
    if (s + 1 < nsteps) issue((s + 1) & 1);
#endif
    compute(s & 1);
  }
  __syncthreads();  // stage memory becomes the per-wave transpose scratch

  // ---- epilogue: scale / bias / SiLU / residual in fp32, per-wave LDS transpose, 16-byte row stores ----
  unsigned char* escr = smem + wave * (PXP * EP_PITCH);
  OT* __restrict__ yg = reinterpret_cast<OT*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  const int n0 = tileN * BN + wn * NFR * 16;
  constexpr int OEPC = 16 / OES;  // output elements per 16-byte chunk
  float st_sum[STATS ? NFR : 1][4], st_sq[STATS ? NFR : 1][4];
  if constexpr (STATS) {
#pragma unroll
    for (int j = 0; j < NFR; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) st_sum[j][e] = 0.f, st_sq[j][e] = 0.f;
    if (blockIdx.x == 0)  // the totals bn_sum_partials_kernel adds the slots into
      for (int i = tid; i < 2 * p.Cout; i += NW * 64) p.stats[i] = 0.0;
  }
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const int m0 = tileM * BM + wm * MFR * 16 + h * PXP;
    if constexpr (W32) {  // D[32 couts][32 pixels]: lane l holds pixel l % 32 and couts 8 j + 4 (l / 32) + e of the fragment, as acc32[..][..][4 j + e]
      static_assert(!W32 || NH == 1, "W32: one epilogue pass");
      const int r32 = lane & 31, kh = lane >> 5;
#pragma unroll
      for (int jf = 0; jf < NFR / 2; ++jf)
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
          const int cl = jf * 32 + 8 * j4 + 4 * kh, co = n0 + cl;
          const f32x4 bb = co + 3 < p.cout_pad ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int i = 0; i < MFR / 2; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc32[jf][i][4 * j4 + e] + bb[e];
            if (p.act == DY_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
            }
            fk_store4<OT>(escr + (i * 32 + r32) * EP_PITCH + cl * OES, v);
          }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const int co = n0 + j * 16 + lq * 4;
      const bool cok = co + 3 < p.cout_pad;  // (a tile may reach past the padded rows of a narrow layer: its weights read as zeros)
      const f32x4 bb = cok ? *reinterpret_cast<const f32x4*>(p.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 sc4 = f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (MX || SP) sc4 = cok ? *reinterpret_cast<const f32x4*>(p.wscale + co) : sc4;
#pragma unroll
      for (int ii = 0; ii < MPP; ++ii) {
        const int i = h * MPP + ii;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (SPL) v[e] = (acc[j][i][e] + accl[j][i][e] * kSplitInv) * sc4[e] + bb[e];
          else v[e] = (MX || SP) ? acc[j][i][e] * sc4[e] + bb[e] : acc[j][i][e] + bb[e];
        }
        if (p.act == DY_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
        }
        if constexpr (RES || SP) {
          *reinterpret_cast<f32x4*>(escr + (ii * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * 4) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          if constexpr (FkIsFp8<OT>::v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
          }
          fk_store4<OT>(escr + (ii * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * OES, v);
          if constexpr (STATS) {
            if (m0 + ii * 16 + lr < p.M) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float f = Elem<OT>::to_f32(Elem<OT>::from_f32(v[e]));  // what BatchNorm will read back
                st_sum[j][e] += f, st_sq[j][e] += f * f;
              }
            }
          }
        }
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (SP && std::is_same<OT, float>::value) {
      // fp32 output (Detect logits): the scratch row is the output row; the last chunk of a Cout that is no multiple of 4 goes out by element
      float* __restrict__ yf = reinterpret_cast<float*>(p.y);
#pragma unroll
      for (int k = 0; k < (PXP * CPP + 63) / 64; ++k) {
        const int idx = k * 64 + lane;
        const int px = idx / CPP, cc = idx - px * CPP;
        const int m = m0 + px, c0 = n0 + cc * 4;
        if (px < PXP && m < p.M && c0 < p.Cout) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(escr + px * EP_PITCH + cc * 16);
          float* dst = yf + (size_t)m * (size_t)p.ldy + (size_t)c0;
          if (c0 + 4 <= p.Cout) {
            *reinterpret_cast<f32x4*>(dst) = t;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (c0 + e < p.Cout) dst[e] = t[e];
          }
        }
      }
    } else if constexpr (SP) {
      // split output: a lane takes 8 channels of a pixel — 8 fp32 from the scratch (+ the residual's pair, joined in fp32: the reference's
      // x + cv2(cv1(x)) rounded once) -> hi chunk + lo chunk, 32 contiguous bytes
      constexpr int GPP = NFR * 2;
      unsigned char* __restrict__ yb = reinterpret_cast<unsigned char*>(p.y);
      const unsigned char* __restrict__ rb = reinterpret_cast<const unsigned char*>(p.res);
#pragma unroll
      for (int k = 0; k < (PXP * GPP + 63) / 64; ++k) {
        const int idx = k * 64 + lane;
        const int px = idx / GPP, g = idx - px * GPP;
        const int m = m0 + px, c0 = n0 + g * 8;
        if (px < PXP && m < p.M && c0 < p.Cout) {
          const f32x4 t0 = *reinterpret_cast<const f32x4*>(escr + px * EP_PITCH + g * 32), t1 = *reinterpret_cast<const f32x4*>(escr + px * EP_PITCH + g * 32 + 16);
          float f[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
          if constexpr (RES) {
            const unsigned char* rp = rb + ((size_t)m * (size_t)p.ldres + (size_t)c0) * 4;
            float r[8];
            join8(*reinterpret_cast<const u32x4*>(rp), *reinterpret_cast<const u32x4*>(rp + 16), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r[e];
          }
          u32x4 hi, lo;
          split8(f, hi, lo);
          unsigned char* dp = yb + ((size_t)m * (size_t)p.ldy + (size_t)c0) * 4;
          *reinterpret_cast<u32x4*>(dp) = hi;
          *reinterpret_cast<u32x4*>(dp + 16) = lo;
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < (PXP * CPP + 63) / 64; ++k) {
      const int idx = k * 64 + lane;
      const int px = idx / CPP, cc = idx - px * CPP;
      const int m = m0 + px;
      if (px < PXP && m < p.M && n0 + cc * OEPC < p.Cout) {
        u32x4 val;
        if constexpr (RES) {
          float f[OEPC], r[OEPC];
#pragma unroll
          for (int e = 0; e < OEPC; e += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(escr + px * EP_PITCH + (cc * OEPC + e) * 4);
            f[e] = t[0], f[e + 1] = t[1], f[e + 2] = t[2], f[e + 3] = t[3];
          }
          Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(rg + (size_t)m * (size_t)p.ldres + (size_t)(n0 + cc * OEPC)), r);
#pragma unroll
          for (int e = 0; e < OEPC; ++e) f[e] = (f[e] + r[e] * p.res_scale) * p.out_scale;
          val = Chunk<OT>::pack(f);
        } else {
          val = *reinterpret_cast<const u32x4*>(escr + px * EP_PITCH + cc * 16);
        }
#ifdef DYOLO_FK_KDBG
        if (p.dbg == 4 && val[0] != 0x7fc07fc1u) continue;
#endif
        *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + (size_t)(n0 + cc * OEPC)) = val;
      }
    }
    }
    if (h + 1 < NH) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if constexpr (STATS) {
    float* sred = reinterpret_cast<float*>(smem);  // [WM][2][BN]
#pragma unroll
    for (int j = 0; j < NFR; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) st_sum[j][e] += __shfl_xor(st_sum[j][e], m, 64), st_sq[j][e] += __shfl_xor(st_sq[j][e], m, 64);
      }
    __syncthreads();  // every wave is done with its transpose scratch
    if (lr == 0) {
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = wn * NFR * 16 + j * 16 + lq * 4 + e;
          sred[(wm * 2 + 0) * BN + cc] = st_sum[j][e];
          sred[(wm * 2 + 1) * BN + cc] = st_sq[j][e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += NW * 64) {
      const int which = i / BN, cc = i - which * BN;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < WM; ++k) t += sred[(k * 2 + which) * BN + cc];
      const int co = tileN * BN + cc;
      if (co < p.Cout) {
        double* dst = p.stats + (size_t)(1 + (p.stats_atomic ? tileM % kStatSlots : tileM)) * 2 * p.Cout + which * p.Cout + co;
        if (p.stats_atomic) unsafeAtomicAdd(dst, (double)t);  // (global_atomic_add_f64; atomicAdd compiles to a compare-and-swap loop)
        else *dst = (double)t;
      }
    }
  }
}

template <typename T, typename OT, int MFR, int NFR, int WM, int WN, int TABN>
static int launch_fk(const FkArgs& a, hipStream_t st, const char* name) {
  FkArgs p = a;
  constexpr int BM = WM * MFR * 16, BN = WN * NFR * 16;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.Cout + BN - 1) / BN;
  p.nblk = tilesM * p.tilesN;
  if constexpr (std::is_same<T, OT>::value) {
    if (p.res != nullptr) {
      hipLaunchKernelGGL((conv_gemm_fk_kernel<T, OT, MFR, NFR, WM, WN, true, TABN>), dim3((unsigned)p.nblk), dim3(WM * WN * 64), 0, st, p);
      return check_launch(name);
    }
#ifndef DYOLO_L2E_BUILD  // (training convolutions carry no activation: they never come through the scaled-domain build)
    if constexpr (sizeof(T) == 2) {
      if (p.stats != nullptr) {
        p.stats_atomic = tilesM > kStatSlots ? 1 : 0;  // (32 -> 64 1x1 stride 2 @320 at B = 64: 12,800 row blocks)
        if (p.stats_atomic) zero_async(p.stats, (size_t)(1 + kStatSlots) * 2 * p.Cout * sizeof(double), st);
        hipLaunchKernelGGL((conv_gemm_fk_kernel<T, OT, MFR, NFR, WM, WN, false, TABN, true>), dim3((unsigned)p.nblk), dim3(WM * WN * 64), 0, st, p);
        note_stats(p.stats_atomic ? kStatSlots : tilesM);
        return check_launch(name);
      }
    }
#endif
  }
  hipLaunchKernelGGL((conv_gemm_fk_kernel<T, OT, MFR, NFR, WM, WN, false, TABN>), dim3((unsigned)p.nblk), dim3(WM * WN * 64), 0, st, p);
  return check_launch(name);
}

// Tile choice: the cout tile that covers Cout with the fewest padded columns (ties: the wider one); 128 pixels.  Measured and dropped
// (tools/fk_ab.sh, r04): 256-pixel tiles — 256 x 160 as sixteen waves of 32 x 80 (LDS-read bound in fp8: 14 ds_read_b128 per 10 MFMAs)
// 3-12 % slower, 256 x 320 as sixteen waves of 64 x 80 (128-register cap: spills) 2x slower; two 128-pixel workgroups per CU that
// cover for each other's barriers beat one big one; eight waves of 32 x 80 on the 128 x 160 tile, two waves of 64 x 80 on 128 x 80: +-2 %.
// The tap table of a 3x3 layer (one word per 16-byte K chunk) has to fit the tile's LDS: 768 words beside the 128 x 160 / 128 / 64
// tiles (K <= 6144 16-bit / 12288 fp8 elements), 128 beside 128 x 80 (53,760 B: the LDS is handed out in 1,280-byte granules, and a
// 256-word table made the workgroup 43 of them — two per CU instead of three: -9 % on the 80 -> 80 layers); returns 1 when it does not.
template <typename T, typename OT>
static int launch_fk_tiles(const FkArgs& a, hipStream_t st) {
  static const int force = dy_ablate("DYOLO_FK_BN");
  const int cands[4] = {160, 128, 80, 64};
  int best = 160;
  long long bw = 1ll << 60;
  for (int c : cands) {
    const long long w = (long long)((a.Cout + c - 1) / c) * c;
    if (w < bw) bw = w, best = c;
  }
  if (force) best = force;
  const int nf = a.ks == 1 ? 0 : a.Kpad / (8 * Elem<T>::EPC) * 8;  // tap-table words
  if (nf > 768) return 1;
  if (best == 80 && nf > 128) best = 160;
  switch (best) {
    case 160: return launch_fk<T, OT, 4, 5, 2, 2, 768>(a, st, "conv_gemm_fk_kernel<128,160>");
    case 128:
#ifdef DYOLO_ABLATE
      if constexpr (sizeof(T) == 2 && std::is_same<T, OT>::value) {
        const char* v32 = getenv("DYOLO_FK_W32");
        if (v32 && atoi(v32) && a.res == nullptr && a.stats == nullptr) {
          FkArgs q = a;
          q.tilesN = (q.Cout + 127) / 128;
          q.nblk = ((q.M + 127) / 128) * q.tilesN;
          hipLaunchKernelGGL((conv_gemm_fk_kernel<T, OT, 4, 4, 2, 2, false, 768, false, true>), dim3((unsigned)q.nblk), dim3(256), 0, st, q);
          return check_launch("conv_gemm_fk_kernel<128,128,w32>");
        }
      }
#endif
      return launch_fk<T, OT, 4, 4, 2, 2, 768>(a, st, "conv_gemm_fk_kernel<128,128>");
    case 80: return launch_fk<T, OT, 2, 5, 4, 1, 128>(a, st, "conv_gemm_fk_kernel<128,80>");
    default: return launch_fk<T, OT, 2, 4, 4, 1, 768>(a, st, "conv_gemm_fk_kernel<128,64>");
  }
}

// Returns 1 when the call is not one this kernel is built for (the caller then runs the generic kernel), else the launch status.
int conv_gemm_fk_try(const dy_conv_desc* d, hipStream_t st) {
  static const int off = dy_ablate("DYOLO_NO_FK");
  if (off) return 1;
  const int es = dy_dtype_size(d->dtype);
  if (d->dtype == DY_F32 || es == 0 || d->groups > 1 || d->out_f32 || d->w_layout != DY_WLAYOUT_ROWS) return 1;
  const int ydt = d->y_dtype1 ? d->y_dtype1 - 1 : d->dtype;
  const bool in8 = d->dtype == DY_FP8, out8 = ydt == DY_FP8;
  if (!(ydt == d->dtype || (in8 && ydt == DY_F16) || (d->dtype == DY_F16 && out8))) return 1;
  const int epc = 16 / es, bke = 8 * epc, oes = dy_dtype_size(ydt);
  if (!((d->ksize == 1 && d->pad == 0) || (d->ksize == 3 && d->pad == 1)) || (d->stride != 1 && d->stride != 2)) return 1;
  if (d->cin % epc || d->cout % (16 / oes) || d->k_pad % bke) return 1;
  if (d->ksize == 3 && (d->cin < bke / 2 || d->x2 || d->up2x)) return 1;
  if (d->up2x > 1 || (d->up2x && d->stride != 1)) return 1;
  if (d->x2 && (d->cin_split % bke || d->cin_split <= 0 || d->cin_split >= d->cin)) return 1;
  if (!aligned16(d->y) || (d->ld_y * oes) % 16 || (d->ld_x * es) % 16 || !aligned16(d->x)) return 1;
  if (d->residual && ((d->ld_res * es) % 16 || !aligned16(d->residual) || ydt != d->dtype)) return 1;
  const int hb = d->up2x ? d->h / 2 : d->h, wb = d->up2x ? d->w_in / 2 : d->w_in;
  const long long lim = (1ll << 32) - (1ll << 24);
  const long long xb = (long long)d->batch * hb * wb * d->ld_x * es, x2b = d->x2 ? (long long)d->batch * d->h * d->w_in * d->ld_x2 * es : 0;
  const long long wbytes = (long long)d->cout_pad * d->k_pad * es;
  if (xb >= lim || x2b >= lim || wbytes >= lim) return 1;
  if (in8 && (!d->w_scale || !(d->act_scale > 0.f))) return 1;
  if (out8 && !(d->act_scale > 0.f)) return 1;

  FkArgs a{};
  a.x = d->x, a.x2 = d->x2 ? d->x2 : d->x, a.w = d->w, a.bias = d->bias, a.wscale = in8 ? d->w_scale : nullptr, a.res = d->residual, a.y = d->y;
  a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.ldx2 = d->x2 ? d->ld_x2 : d->ld_x, a.split = d->x2 ? d->cin_split : d->cin;
  a.HB = hb, a.WB = wb;
  a.Ho = d->ho, a.Wo = d->wo, a.Cout = d->cout, a.ldy = d->ld_y, a.ldres = d->ld_res;
  a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  a.Kpad = d->k_pad, a.M = d->batch * d->ho * d->wo, a.HoWo = d->ho * d->wo, a.up2x = d->up2x;
  a.act = d->act;
  a.cout_pad = d->cout_pad;
  a.xb = (unsigned)xb, a.x2b = (unsigned)(d->x2 ? x2b : xb), a.wb = (unsigned)wbytes;
  a.dCin = make_fastdiv((unsigned)d->cin);
  if (d->ksize == 3 && ((long long)(2 * d->w_in + 2) * d->ld_x + d->cin) * es >= (1ll << 28)) return 1;  // table words hold a 28-bit byte offset
  a.dbg = dy_ablate("DYOLO_FK_DBG");
  a.stats = (d->y_dtype1 || d->bnb_z) ? nullptr : d->bn_stats;
  a.res_scale = in8 ? d->act_scale : 1.f;
  a.out_scale = out8 ? 1.f / d->act_scale : 1.f;
  if (in8) return out8 ? launch_fk_tiles<fp8_t, fp8_t>(a, st) : launch_fk_tiles<fp8_t, f16_t>(a, st);
  if (d->dtype == DY_F16) return out8 ? launch_fk_tiles<f16_t, fp8_t>(a, st) : launch_fk_tiles<f16_t, f16_t>(a, st);
  return launch_fk_tiles<bf16_t, bf16_t>(a, st);
}

// DY_F16X2 (split float16, include/dyolo.h): every dense convolution of the type runs here — there is no other kernel for it, so a
// call outside the built set is an error, not a fallback.  Tiles as above; the tap table holds up to 1,536 words (3x3 with Cin <= 680).
#ifndef DYOLO_L2E_BUILD
template <typename OT>
static int launch_fk_split_tiles(const FkArgs& a, hipStream_t st) {
#ifdef DYOLO_ABLATE
  if constexpr (std::is_same<OT, f16x2_t>::value) {  // tile study (tools/split_tiles.py): MFR, NFR, WM, WN, table words
    const char* v = getenv("DYOLO_SPLIT_CFG");
    switch (v ? atoi(v) : 0) {
      case 1: return launch_fk<f16x2_t, OT, 2, 4, 4, 1, 256>(a, st, "split<128,64> 4w 32x64 t256");
      case 2: return launch_fk<f16x2_t, OT, 4, 4, 2, 1, 256>(a, st, "split<128,64> 2w 64x64 t256");
      case 3: return launch_fk<f16x2_t, OT, 4, 4, 4, 1, 256>(a, st, "split<256,64> 4w 64x64 t256");
      case 4: return launch_fk<f16x2_t, OT, 2, 4, 8, 1, 256>(a, st, "split<256,64> 8w 32x64 t256");
      case 5: return launch_fk<f16x2_t, OT, 2, 2, 4, 1, 256>(a, st, "split<128,32> 4w 32x32 t256");
      case 6: return launch_fk<f16x2_t, OT, 4, 2, 4, 1, 256>(a, st, "split<256,32> 4w 64x32 t256");
      case 7: return launch_fk<f16x2_t, OT, 2, 2, 8, 1, 256>(a, st, "split<256,32> 8w 32x32 t256");
      case 8: return launch_fk<f16x2_t, OT, 4, 4, 2, 2, 640>(a, st, "split<128,128> 4w 64x64 t640");
      case 9: return launch_fk<f16x2_t, OT, 4, 4, 4, 2, 640>(a, st, "split<256,128> 8w 64x64 t640");
      case 10: return launch_fk<f16x2_t, OT, 2, 4, 4, 2, 640>(a, st, "split<128,128> 8w 32x64 t640");
      case 11: return launch_fk<f16x2_t, OT, 4, 2, 2, 4, 640>(a, st, "split<128,128> 8w 64x32 t640");
      case 12: return launch_fk<f16x2_t, OT, 4, 4, 1, 1, 256>(a, st, "split<64,64> 1w 64x64 t256");
      case 13: return launch_fk<f16x2_t, OT, 4, 2, 2, 1, 256>(a, st, "split<128,32> 2w 64x32 t256");
      default: break;
    }
  }
#endif
  // Tile rule (tools/split_tiles.py on the layer shapes of scale s at B = 256, r05): the kernel is bound by LDS traffic and latency, not
  // by the matrix pipe, and MORE waves beat bigger wave tiles — 128 x 128 as eight waves of 32 x 64 ran 1.35-1.6x the four 64 x 64 waves
  // of the 16-bit kernel's shape, Cout <= 32 (the stride-4 C2f's Bottlenecks) wants a 256-pixel tile of eight 32 x 32 waves (1.9x the
  // half-empty 64-wide tile), and the tap table is sized by need (256 / 640 / 1536 words): 6 KB of table beside a 128 x 64 tile cost
  // the third workgroup of a CU (-15 %).  The x scale's 160 / 80 wide tiles keep the 16-bit kernel's shapes.
  const int words = a.ks == 1 ? 0 : a.Kpad / 32 * 8;
  const int cands[4] = {160, 128, 80, 64};
  int best = 160;
  long long bw = 1ll << 60;
  for (int c : cands) {
    const long long w = (long long)((a.Cout + c - 1) / c) * c;
    if (w < bw) bw = w, best = c;
  }
  if (a.Cout <= 32) best = 32;
#define DY_SPLIT_TAB(MFR, NFR, WM, WN, NAME)                                                   \
  return words <= 256 ? launch_fk<f16x2_t, OT, MFR, NFR, WM, WN, 256>(a, st, NAME)              \
         : words <= 640 ? launch_fk<f16x2_t, OT, MFR, NFR, WM, WN, 640>(a, st, NAME)            \
                        : launch_fk<f16x2_t, OT, MFR, NFR, WM, WN, 1536>(a, st, NAME)
  switch (best) {
    case 160: return launch_fk<f16x2_t, OT, 4, 5, 2, 2, 1536>(a, st, "conv_gemm_fk_kernel<split,128,160>");
    case 80: return launch_fk<f16x2_t, OT, 2, 5, 4, 1, 1536>(a, st, "conv_gemm_fk_kernel<split,128,80>");
    case 128: DY_SPLIT_TAB(2, 4, 4, 2, "conv_gemm_fk_kernel<split,128,128>");
    case 32: DY_SPLIT_TAB(2, 2, 8, 1, "conv_gemm_fk_kernel<split,256,32>");
    default: DY_SPLIT_TAB(2, 4, 4, 1, "conv_gemm_fk_kernel<split,128,64>");
  }
#undef DY_SPLIT_TAB
}

int conv_gemm_fk_split(const dy_conv_desc* d, hipStream_t st) {
  constexpr int es = 4, epc = 4, bke = 32;
  DY_REQUIRE(d->groups <= 1 && d->w_layout == DY_WLAYOUT_ROWS && !d->y_dtype1 && !d->bn_stats, DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2 is built for dense convolutions in DY_WLAYOUT_ROWS (no y_dtype1 / bn_stats)");
  DY_REQUIRE(((d->ksize == 1 && d->pad == 0) || (d->ksize == 3 && d->pad == 1)) && (d->stride == 1 || d->stride == 2), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2 is built for 1x1 (pad 0) and 3x3 (pad 1), stride 1 or 2");
  DY_REQUIRE(d->cin % 8 == 0 && (d->out_f32 || d->cout % 8 == 0) && d->k_pad % bke == 0, DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2 needs cin (and cout, unless out_f32) in whole groups of 8 channels");
  DY_REQUIRE(d->w_scale && aligned16(d->w_scale), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: DY_F16X2 needs w_scale (fp32[cout_pad]: the inverse row scales)");
  DY_REQUIRE(!(d->ksize == 3 && (d->x2 || d->up2x)) && d->up2x <= 1 && !(d->up2x && d->stride != 1), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2: x2 / up2x are built for 1x1 stride 1");
  DY_REQUIRE(!d->x2 || (d->cin_split % bke == 0 && d->cin_split > 0 && d->cin_split < d->cin), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2: cin_split must be a multiple of 32");
  DY_REQUIRE(aligned16(d->y) && d->ld_y % epc == 0 && d->ld_x % epc == 0 && aligned16(d->x) && (!d->x2 || (aligned16(d->x2) && d->ld_x2 % epc == 0)), DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: DY_F16X2 views must be 16-byte aligned");
  DY_REQUIRE(!d->residual || (!d->out_f32 && d->ld_res % epc == 0 && aligned16(d->residual)), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: DY_F16X2 residual view");
  const int hb = d->up2x ? d->h / 2 : d->h, wb = d->up2x ? d->w_in / 2 : d->w_in;
  const long long lim = (1ll << 32) - (1ll << 24);
  const long long xb = (long long)d->batch * hb * wb * d->ld_x * es, x2b = d->x2 ? (long long)d->batch * d->h * d->w_in * d->ld_x2 * es : 0;
  const long long wbytes = (long long)d->cout_pad * d->k_pad * es;
  DY_REQUIRE(xb < lim && x2b < lim && wbytes < lim, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: DY_F16X2 view beyond the 4 GiB a buffer descriptor addresses");
  DY_REQUIRE(d->ksize == 1 || (d->k_pad / bke * 8 <= 1536 && ((long long)(2 * d->w_in + 2) * d->ld_x + d->cin) * es < (1ll << 28)), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: DY_F16X2 3x3 tap table (cin <= 680)");
  FkArgs a{};
  a.x = d->x, a.x2 = d->x2 ? d->x2 : d->x, a.w = d->w, a.bias = d->bias, a.wscale = d->w_scale, a.res = d->residual, a.y = d->y;
  a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.ldx2 = d->x2 ? d->ld_x2 : d->ld_x, a.split = d->x2 ? d->cin_split : d->cin;
  a.HB = hb, a.WB = wb;
  a.Ho = d->ho, a.Wo = d->wo, a.Cout = d->cout, a.ldy = d->ld_y, a.ldres = d->ld_res;
  a.ks = d->ksize, a.stride = d->stride, a.pad = d->pad;
  a.Kpad = d->k_pad, a.M = d->batch * d->ho * d->wo, a.HoWo = d->ho * d->wo, a.up2x = d->up2x;
  a.act = d->act;
  a.cout_pad = d->cout_pad;
  a.xb = (unsigned)xb, a.x2b = (unsigned)(d->x2 ? x2b : xb), a.wb = (unsigned)wbytes;
  a.dCin = make_fastdiv((unsigned)d->cin);
  a.dbg = 0, a.stats = nullptr, a.res_scale = 1.f, a.out_scale = 1.f;
  return d->out_f32 ? launch_fk_split_tiles<float>(a, st) : launch_fk_split_tiles<f16x2_t>(a, st);
}
#else
int conv_gemm_fk_split(const dy_conv_desc*, hipStream_t) {
  set_error("dy_conv2d_nhwc: DY_F16X2 runs in the reference's activation units (no DY_ACT_SILU_L2E)");
  return DY_ERR_UNSUPPORTED;
}
#endif

}  // namespace DY_NS
