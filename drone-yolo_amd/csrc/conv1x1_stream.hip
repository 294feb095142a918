// 1x1 (stride 1) NHWC convolution = GEMM C[M][N] = A[M][K] W[N][K]^T on CDNA4 MFMA, streaming form.
//
// The 1x1 layers of the network are bandwidth bound (K = 64..384, N = 64..256: 12-100 FLOP per byte of
// activations) and their K loop is only 2-12 MFMA k-groups long, so a block-synchronous LDS pipeline
// (conv_igemm.hip) spends its time in barriers and pipeline fill.  Here nothing is synchronised after
// the prologue:
//   - a 512-thread workgroup (8 waves) loads the n-tile's weights (fragment order, <= 96 KB) and bias
//     into LDS once; then every WAVE walks its own list of 16*MF-pixel row tiles:
//   - A fragments are loaded straight from global memory into registers, the next tile's are in flight
//     while the current tile is multiplied — register double buffering, no LDS bytes, no barrier.  The
//     loads run in LOADER order: lane 4 p + q takes chunk q of pixel p, so a quad of lanes reads 64
//     contiguous bytes and the L1 sees 16 requests per instruction.  In MFMA operand order (lane 16 q + p)
//     every lane of a quad is another pixel row = 64 requests per instruction, and the L1's request rate,
//     not HBM, bounded these layers at 4-5 TB/s.  ds_bpermute_b32 turns loader order into operand order
//     when a tile's registers are handed to the multiply (crossbar only, four per 16-byte chunk); used
//     up to K = 192, beyond that the crossbar passes cost more than the requests they save;
//   - the gather understands the folded Concat/Upsample in front of C2f.cv1 (channels [0,split) from x
//     through a 2x nearest upsample, the rest from x2), like conv_igemm.hip;
//   - weights are the MFMA A operand, so a lane holds 4 consecutive output channels of one pixel;
//     bias + SiLU on registers, then a per-wave LDS transpose and 16-byte full-row stores.
// Waves drift apart freely, so one wave's loads / epilogue overlap another's MFMAs.
//
// Reference semantics: Conv k=1 (nn/modules/conv.py:37-55), the plain nn.Conv2d heads of Detect
// (head.py:43-57, fp32 output), C2f.cv1 after Concat(+Upsample) (block.py:237-242, conv.py:323-333).
#include "common_hip.h"
#include <type_traits>

namespace DY_NS {

struct Conv1Args {
  const void* x;
  const void* x2;
  const void* w;
  const float* bias;
  void* y;
  int M, Cin, ldx, ldx2, split;
  int up2x, H, W, HB, WB;  // output (= upsampled) dims and the dims of the x buffer
  int Cout, ldy, act;
  int tilesN, nMT;
  FastDiv div_hw, div_w;  // m / (H*W) and r / W of the up2x pixel decode
  double* stats;  // optional: a dy_bn_train_fwd workspace, pixel group g's per-channel sums / sums of squares of the STORED output go to slot g (dy_conv_desc.bn_stats)
};

// STATS (training forward in front of a train-mode BatchNorm): in the row-store pass of the epilogue a lane always holds the same 8
// output channels (chunk lane % CPR of the tile's rows), so it keeps their sum / sum of squares over all of its wave's tiles in 16
// registers; the kernel ends with a shuffle reduction over the lanes of a chunk, a sum over the 8 waves through LDS and plain stores
// into the workgroup's slot of the BatchNorm workspace (see conv3x3_hreg.hip: same protocol).
template <typename T, int NKG, int MF, int NF, bool OUTF32, bool STATS = false>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(const Conv1Args p) {
  constexpr int NT = 512;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int KCE = 4 * EPC;
  constexpr int BN = NF * 16;
  constexpr int W_CHUNKS = NKG * NF * 64;  // 16-byte chunks of one n-tile's weights
  typedef typename std::conditional<OUTF32, float, T>::type OutT;
  constexpr int EP_PITCH = BN * (int)sizeof(OutT) + 16;
  constexpr int EP_BYTES = MF * 16 * EP_PITCH;

  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const float* const sbias = reinterpret_cast<const float*>(dyn_smem + W_CHUNKS * 16);
  unsigned char* const escr = dyn_smem + W_CHUNKS * 16 + BN * 4 + wave * EP_BYTES;

  // The tilesN workgroups of one group walk the same pixel tiles (one cout tile each).  Workgroups go to the 8 XCDs round
  // robin, so with nt = blockIdx % tilesN the group's members sit behind different L2s and every activation row is fetched
  // from memory tilesN times (measured: 1.7x the algorithmic fetch on the 384 -> 256 layers).  With groups % 8 == 0 the members
  // are blockIdx, blockIdx + 8 ...: same XCD, started together.
  const int groups = (int)gridDim.x / p.tilesN;
  int nt, group;
  if (p.tilesN > 1 && (groups & 7) == 0) {
    const int x = (int)blockIdx.x & 7, r = (int)blockIdx.x >> 3;
    nt = r % p.tilesN, group = (r / p.tilesN) * 8 + x;
  } else {
    nt = (int)blockIdx.x % p.tilesN, group = (int)blockIdx.x / p.tilesN;
  }

  {  // prologue: weights + bias of this n-tile -> LDS (the only barrier of the kernel)
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.w) + (size_t)nt * W_CHUNKS;
    for (int s = tid; s < W_CHUNKS; s += NT) *reinterpret_cast<u32x4*>(dyn_smem + s * 16) = wsrc[s];
    float* sb = const_cast<float*>(sbias);
    for (int s = tid; s < BN; s += NT) sb[s] = p.bias[nt * BN + s];
  }
  __syncthreads();

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ x2g = reinterpret_cast<const T*>(p.x2);
  OutT* __restrict__ yg = reinterpret_cast<OutT*>(p.y);
  const int stride = groups * 8;
  int mt = group * 8 + wave;
  if (!STATS && mt >= p.nMT) return;
  constexpr int CPR_S = BN * (int)sizeof(OutT) / 16;  // 16-byte chunks per output row of the tile (STATS: a lane's chunk is lane % CPR_S)
  float st_sum[STATS ? 8 : 1], st_sq[STATS ? 8 : 1];
  if constexpr (STATS) {
    static_assert(!STATS || (sizeof(OutT) == 2 && 64 % CPR_S == 0), "STATS: 16-bit output, the chunk of a lane must not depend on the round");
#pragma unroll
    for (int e = 0; e < 8; ++e) st_sum[e] = 0.f, st_sq[e] = 0.f;
    if (blockIdx.x == 0)  // the totals the BatchNorm's partial-sum launch adds into
      for (int i = tid; i < 2 * p.Cout; i += NT) p.stats[i] = 0.0;
  }

  // measured per layer (B = 256): K = 128 -10 %, K = 192 -2 %, K = 256 +4 %, K = 384 +11 %: the crossbar passes grow with K, the
  // saved L1 requests do not pay for them from K = 256 up
  constexpr bool LOADER = NKG * KCE * (int)sizeof(T) <= 384;
  const int lp = LOADER ? lane >> 2 : lr, lc = LOADER ? lane & 3 : lq;  // pixel and chunk-in-k-group this lane loads
  const int to_operand = 4 * (lr * 4 + lq);           // ds_bpermute byte index: operand lane (lr, lq) <- loader lane 4 lr + lq
  auto load_tile = [&](int tile, u32x4 (&a)[MF][NKG]) {
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      int m = tile * (MF * 16) + i * 16 + lp;
      m = m < p.M ? m : p.M - 1;  // rows past the end load a valid row and are never stored
      size_t off1, off2 = (size_t)m * (size_t)p.ldx2;
      if (p.up2x) {
        const int n = (int)fastdiv((unsigned)m, p.div_hw), r = m - n * (p.H * p.W);
        const int yy = (int)fastdiv((unsigned)r, p.div_w), xx = r - yy * p.W;
        off1 = (size_t)((n * p.HB + (yy >> 1)) * p.WB + (xx >> 1)) * (size_t)p.ldx;
      } else {
        off1 = (size_t)m * (size_t)p.ldx;
      }
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) {
        const int c = kg * KCE + lc * EPC;
        u32x4 v = zero_chunk();
        if (c < p.split) {
          v = *reinterpret_cast<const u32x4*>(xg + off1 + c);
        } else if (c < p.Cin) {
          v = *reinterpret_cast<const u32x4*>(x2g + off2 + (c - p.split));
        }
        a[i][kg] = v;
      }
    }
  };

  auto process = [&](int tile, const u32x4 (&a)[MF][NKG]) {
    f32x4 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* sw = dyn_smem + lane * 16;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) {
      u32x4 b[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) b[j] = *reinterpret_cast<const u32x4*>(sw + (kg * NF + j) * 1024);
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = Elem<T>::mma(b[j], a[i][kg], acc[i][j]);  // D[cout][pixel]
    }
    mfma_epilogue_fence<T>();
    // epilogue: bias + activation on registers, transpose through the wave's LDS scratch, row stores
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sbias + j * 16 + lq * 4);
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        float v[4] = {acc[i][j][0] + bb[0], acc[i][j][1] + bb[1], acc[i][j][2] + bb[2], acc[i][j][3] + bb[3]};
        if (p.act == DY_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
        }
        unsigned char* sp = escr + (i * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * (int)sizeof(OutT);
        if constexpr (OUTF32 || sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(sp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          typedef __attribute__((ext_vector_type(4))) T t4;
          t4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
          *reinterpret_cast<u32x2*>(sp) = __builtin_bit_cast(u32x2, o);
        }
      }
    }
    constexpr int CPR = BN * (int)sizeof(OutT) / 16;  // 16-byte chunks per pixel row of the tile
    constexpr int VE = 16 / (int)sizeof(OutT);
    constexpr int ROUNDS = (MF * 16 * CPR + 63) / 64;
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
      const int idx = k * 64 + lane;
      const int pixl = idx / CPR, cc = idx - pixl * CPR;
      const int m = tile * (MF * 16) + pixl;
      const int co = nt * BN + cc * VE;
      if ((MF * 16 * CPR) % 64 != 0 && pixl >= MF * 16) continue;
      const u32x4 val = *reinterpret_cast<const u32x4*>(escr + pixl * EP_PITCH + cc * 16);
      if constexpr (STATS) {
        if (m < p.M) {  // (channels past Cout hold zeros: zero weights and bias)
          float f[8];
          Chunk<T>::unpack(val, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) st_sum[e] += f[e], st_sq[e] += f[e] * f[e];
        }
      }
      if (m < p.M && co < p.Cout) {
        OutT* yp = yg + (size_t)m * (size_t)p.ldy + co;
        if (co + VE <= p.Cout) {
          *reinterpret_cast<u32x4*>(yp) = val;
        } else {  // ragged channel tail (e.g. nc = 10 class logits)
          typedef __attribute__((ext_vector_type(VE))) OutT vout_t;
          const vout_t sv = __builtin_bit_cast(vout_t, val);
#pragma unroll
          for (int e = 0; e < VE; ++e)
            if (e < p.Cout - co) yp[e] = sv[e];
        }
      }
    }
  };

  // register double buffering: the next tile's fragments are in flight (a_nxt) while the current tile
  // (a_cur, a register copy) is multiplied.  One load body and one compute body keep the kernel inside
  // the 256-VGPR budget of a 512-thread workgroup.
  u32x4 a_cur[MF][NKG], a_nxt[MF][NKG];
  if (mt < p.nMT) load_tile(mt, a_nxt);
  while (mt < p.nMT) {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) {
#pragma unroll
        for (int w = 0; w < 4; ++w) a_cur[i][kg][w] = LOADER ? (unsigned)__builtin_amdgcn_ds_bpermute(to_operand, (int)a_nxt[i][kg][w]) : a_nxt[i][kg][w];
      }
    const int nx = mt + stride;
    if (nx < p.nMT) load_tile(nx, a_nxt);
    process(mt, a_cur);
    if (nx >= p.nMT) break;
    mt = nx;
  }
  if constexpr (STATS) {
    // lanes of one chunk (lane % CPR_S) -> the wave's sums; the wave leaves them in ITS scratch, then 2 * BN threads add the 8 waves up
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int msk = CPR_S; msk < 64; msk <<= 1) st_sum[e] += __shfl_xor(st_sum[e], msk, 64), st_sq[e] += __shfl_xor(st_sq[e], msk, 64);
    }
    float* wsum = reinterpret_cast<float*>(escr);  // [2][BN]
    if (lane < CPR_S) {
#pragma unroll
      for (int e = 0; e < 8; ++e) wsum[lane * 8 + e] = st_sum[e], wsum[BN + lane * 8 + e] = st_sq[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, ch = tid - which * BN, co = nt * BN + ch;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += reinterpret_cast<const float*>(dyn_smem + W_CHUNKS * 16 + BN * 4 + w * EP_BYTES)[which * BN + ch];
      if (co < p.Cout) p.stats[(size_t)(1 + group) * 2 * p.Cout + which * p.Cout + co] = (double)t;
    }
  }
}

constexpr int kLds1 = 160 * 1024;

template <typename T, int NKG, int MF, int NF, bool OUTF32>
static int launch_1x1(const Conv1Args& a, hipStream_t st) {
  Conv1Args p = a;
  constexpr int BN = NF * 16;
  p.tilesN = (p.Cout + BN - 1) / BN;
  p.nMT = (p.M + MF * 16 - 1) / (MF * 16);
  const int osz = OUTF32 ? 4 : (int)sizeof(T);
  const int smem = NKG * NF * 1024 + BN * 4 + 8 * MF * 16 * (BN * osz + 16);
  DY_REQUIRE(smem <= kLds1, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: FRAG1X1 tile needs %d B of LDS", smem);
  const int per_cu = kLds1 / smem >= 2 ? 2 : 1;
  int groups = (256 * per_cu) / p.tilesN;
  if (groups < 1) groups = 1;
  const int need = (p.nMT + 7) / 8;  // groups that still get at least one tile per wave-slot
  if (groups > need) groups = need;
#ifndef DYOLO_L2E_BUILD
  if constexpr (!OUTF32 && sizeof(T) == 2 && NF >= 4) {
    if (p.stats) {
      auto kern_s = conv1x1_stream_kernel<T, NKG, MF, NF, OUTF32, true>;
      static const hipError_t once_s = hipFuncSetAttribute((const void*)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, kLds1);
      (void)once_s;
      hipLaunchKernelGGL(kern_s, dim3((unsigned)(groups * p.tilesN)), dim3(512), smem, st, p);
      note_stats(groups);  // slots written: one per pixel group
      return check_launch("conv1x1_stream_kernel");
    }
  }
#endif
  auto kern = conv1x1_stream_kernel<T, NKG, MF, NF, OUTF32>;
  static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLds1);
  (void)once;
  hipLaunchKernelGGL(kern, dim3((unsigned)(groups * p.tilesN)), dim3(512), smem, st, p);
  return check_launch("conv1x1_stream_kernel");
}

template <typename T, int NKG>
static int launch_1x1_nkg(const Conv1Args& a, bool out_f32, hipStream_t st) {
  constexpr int MF = NKG <= 2 ? 2 : 1;  // fatter variants would spill (a_cur + a_nxt + acc + b fragments)
  if (out_f32 && sizeof(T) < 4) {
    if constexpr (NKG == 2) {  // Detect heads: 64 -> 64 box bins, 64 -> nc class logits
      if (a.Cout <= 16) return launch_1x1<T, 2, 2, 1, true>(a, st);
      if (a.Cout <= 64) return launch_1x1<T, 2, 2, 4, true>(a, st);
    }
    set_error("dy_conv2d_nhwc: FRAG1X1 fp32-output variant not built for cin %d cout %d", a.Cin, a.Cout);
    return DY_ERR_UNSUPPORTED;
  }
  if (a.Cout <= 16) {
    if constexpr (NKG == 2) return launch_1x1<T, 2, 2, 1, false>(a, st);
    set_error("dy_conv2d_nhwc: FRAG1X1 cout <= 16 only built for cin <= 2 k-groups");
    return DY_ERR_UNSUPPORTED;
  }
  if (a.Cout <= 64) return launch_1x1<T, NKG, MF, 4, false>(a, st);
  return launch_1x1<T, NKG, MF, 8, false>(a, st);
}

template <typename T>
static int launch_1x1_dtype(const Conv1Args& a, bool out_f32, hipStream_t st) {
  const int nkg = (a.Cin + 4 * Elem<T>::EPC - 1) / (4 * Elem<T>::EPC);
  switch (nkg) {
    case 2: return launch_1x1_nkg<T, 2>(a, out_f32, st);
    case 3: return launch_1x1_nkg<T, 3>(a, out_f32, st);
    case 4: return launch_1x1_nkg<T, 4>(a, out_f32, st);
    case 6: return launch_1x1_nkg<T, 6>(a, out_f32, st);
    case 8: return launch_1x1_nkg<T, 8>(a, out_f32, st);
    case 12: return launch_1x1_nkg<T, 12>(a, out_f32, st);
    default:
      set_error("dy_conv2d_nhwc: FRAG1X1 not built for %d k-groups (cin %d); pack with DY_WLAYOUT_ROWS", nkg, a.Cin);
      return DY_ERR_UNSUPPORTED;
  }
}

// Entry used by dy_conv2d_nhwc when d->w_layout == DY_WLAYOUT_FRAG1X1.
int conv1x1_stream_dispatch(const dy_conv_desc* d, hipStream_t st) {
  const int es = dtype_size_no_fp8(d->dtype);
  const int epc = 16 / es;
  DY_REQUIRE(d->ksize == 1 && d->stride == 1 && d->pad == 0 && d->groups <= 1 && !d->residual, DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: FRAG1X1 layout needs a dense 1x1 stride-1 conv without residual");
  DY_REQUIRE(d->cin % epc == 0, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: cin %d not a multiple of %d", d->cin, epc);
  DY_REQUIRE(aligned16(d->x) && (d->ld_x * es) % 16 == 0 && aligned16(d->w) && aligned16(d->bias) && aligned16(d->y), DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: views must be 16-byte aligned");
  DY_REQUIRE((d->ld_y * (d->out_f32 ? 4 : es)) % 16 == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: output pitch must be a multiple of 16 bytes");
  Conv1Args a{};
  a.x = d->x;
  a.x2 = d->x;
  a.w = d->w;
  a.bias = d->bias;
  a.y = d->y;
  a.M = d->batch * d->ho * d->wo;
  a.Cin = d->cin;
  a.ldx = d->ld_x;
  a.ldx2 = d->ld_x;
  a.split = d->cin;
  a.up2x = d->up2x ? 1 : 0;
  a.H = d->h;
  a.W = d->w_in;
  a.HB = d->up2x ? d->h / 2 : d->h;
  a.WB = d->up2x ? d->w_in / 2 : d->w_in;
  a.Cout = d->cout;
  a.ldy = d->ld_y;
  a.act = d->act;
  a.div_hw = make_fastdiv((unsigned)(d->h * d->w_in));
  a.div_w = make_fastdiv((unsigned)d->w_in);
  a.stats = d->bnb_z ? nullptr : d->bn_stats;
  if (d->x2) {
    DY_REQUIRE(d->cin_split > 0 && d->cin_split < d->cin && d->cin_split % epc == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: bad cin_split %d", d->cin_split);
    DY_REQUIRE(aligned16(d->x2) && (d->ld_x2 * es) % 16 == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: x2 view misaligned");
    a.x2 = d->x2;
    a.ldx2 = d->ld_x2;
    a.split = d->cin_split;
  }
  if (d->up2x) DY_REQUIRE(d->up2x == 1 && d->h % 2 == 0 && d->w_in % 2 == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: FRAG1X1 takes up2x 0/1 only and needs even h,w");
  const bool of = d->out_f32 != 0;
  switch (d->dtype) {
    case DY_BF16: return launch_1x1_dtype<bf16_t>(a, of, st);
    case DY_F16: return launch_1x1_dtype<f16_t>(a, of, st);
    default: return launch_1x1_dtype<float>(a, false, st);
  }
}

}  // namespace DY_NS
