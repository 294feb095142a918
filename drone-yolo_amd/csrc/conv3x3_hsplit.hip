// 3x3 stride-1 pad-1 convolution on SPLIT float16 storage (DY_F16X2, include/dyolo.h) for the narrow layers (cin 32 / 64): the
// register-weight decomposition of conv3x3_hreg.hip, for the precision YOLO.predict runs by default (r05).
//
// Why.  The type's dense convolutions ran on the flat-K implicit GEMM (conv_gemm_fk.hip), which gathers a fresh 128-byte row per pixel
// and TAP: every input pixel crosses the L2 -> LDS path nine times, and at 4 bytes per element that path is what bounds the layers —
// 32 -> 32 @160^2, 64 -> 64 @80^2 / @160^2 and 128 -> 128 @40^2 all moved ~10 TB/s of LDS-DMA bytes whatever their MFMA rate
// (profiles/r05_conv_layers_f16x2_b256.txt: 145 - 330 TFLOP/s algorithmic).  Here, as in conv3x3_hreg.hip:
//   * a wave owns ONE 16-cout fragment; its weights — 9 taps x NCH chunks of 32 channels x (hi, lo) MFMA A fragments = 72 / 144 VGPRs — are
//     loaded once per workgroup lifetime;
//   * the tile's halo (10 x 18 pixels of an 8 x 16 tile) is staged ONCE per 32-channel chunk and read at the nine shifts.  A pixel's chunk is
//     128 bytes — four (hi, lo) pairs of 8 channels — and the two halves go to TWO images of conv3x3_hreg's format (64-byte pixel pitch, part
//     index XOR ((column >> 1) & 3)): the hi image holds the four hi halves, the lo image the four lo halves, so every fragment read is that
//     kernel's conflict-free ds_read_b128 (the LDS-DMA picks each lane's source chunk: hi of pair k at byte 32 k, lo at 32 k + 16);
//   * per chunk a wave walks the 10 halo rows once: 6 fragment reads (three column shifts x hi, lo) feed up to 27 MFMAs
//     (w_hi x_hi + w_lo x_hi into the accumulator, w_hi x_lo into a second one that joins with 2^-11 in the epilogue, as conv_gemm_fk.hip);
//   * NF = 2 (cout 32: the Bottlenecks of the stride-4 C2f blocks): waves 0, 1 take the two cout fragments for tile rows 0..3, waves 2, 3
//     for rows 4..7 (6 halo rows each).
// Two stages of (hi, lo) images = 64 KB: two workgroups per CU; the next item's DMA runs under this item's MFMAs, one barrier per item.
// Epilogue: inverse row scale, bias, SiLU, optional residual (a Bottleneck's shortcut), then the lane's 4 channels are half of an
// 8-channel group: v_permlane16_swap between the rows of a pair leaves 16 contiguous bytes of hi and of lo in every lane (conv3x3_hreg's store).
// Reference semantics: Conv (nn/modules/conv.py:37-55, BatchNorm folded), Bottleneck shortcut (block.py:337-350).
#include "common_hip.h"
#include "conv_args.h"

namespace dy {

struct HsArgs {
  const void* x;       // split-float16 NHWC view, pitch ldx (4-byte elements)
  const void* w;       // DY_WLAYOUT_ROWS split rows: [cout_pad][(tap, 8-channel group) x (hi x 8 | lo x 8)] float16
  const float* bias;   // cout_pad floats
  const float* wscale; // cout_pad inverse row scales
  const void* res;
  void* y;
  int N, H, W, Cin, ldx, Cout, ldy, ldres, act, Kpad;
  int tilesX, tilesY, tilesN, nSpatial;
  unsigned x_bytes, y_bytes, r_bytes;
};

constexpr int kHsTH = 8, kHsTW = 16, kHsHH = 10, kHsHW = 24;
constexpr int kHsImage = 16 * 1024;       // one (tile, chunk) halo image of one half: 10 x 24 x 64 = 15,360 B, padded to 16 wave-instructions
constexpr int kHsStage = 2 * kHsImage;    // hi image, lo image
constexpr unsigned kHsOob = 0xffff0000u;  // >= num_records of every descriptor here (+ 16 for the lo half stays inside 32 bits)

// NCH: 32-channel chunks of the input; NF: 16-cout fragments per workgroup (4: cout % 64 == 0; 2: cout % 32 == 0, the waves split the rows)
template <int NCH, int NF, bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_hsplit_kernel(const HsArgs p) {
  constexpr int ROWS = NF == 4 ? kHsTH : kHsTH / 2;  // output rows of this wave
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * kHsStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const int frag = NF == 4 ? wave : (wave & 1);      // the wave's cout fragment inside the workgroup's group
  const int row0 = NF == 4 ? 0 : (wave >> 1) * ROWS; // its first output row
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(RES ? p.res : p.y), 0, RES ? p.r_bytes : 0u, 0x00020000);

  // block -> (cout group, spatial sequence), XCD-contiguous (conv3x3_hreg.hip)
  const int G = (int)gridDim.x;
  const int logical = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const int nt = logical % p.tilesN;
  const int sb = logical / p.tilesN, Gs = G / p.tilesN;
  const int myTiles = sb < p.nSpatial ? (p.nSpatial - sb + Gs - 1) / Gs : 0;
  if (myTiles <= 0) return;
  const int nItems = myTiles * NCH;
  const int co_w = nt * (NF * 16) + frag * 16;  // first cout of this wave

  // ---- this wave's weights: 16 couts x all taps x all chunks, hi and lo halves, in registers ----
  u32x4 whi[NCH][9], wlo[NCH][9];
  {
    const f16_t* row = reinterpret_cast<const f16_t*>(p.w) + (size_t)(co_w + lr) * (size_t)(2 * p.Kpad);
    const int gpt = p.Cin >> 3;  // 8-channel groups per tap
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f16_t* src = row + (size_t)(t * gpt + c * 4 + lq) * 16;
        whi[c][t] = *reinterpret_cast<const u32x4*>(src);
        wlo[c][t] = *reinterpret_cast<const u32x4*>(src + 8);
      }
  }
  // (bias, inverse row scales and the store offsets are formed in the epilogue: 144 weight registers + 64 accumulators leave the main loop of
  // the 64-channel form no room for them)

  // ---- loader: slot s = (k * 4 + wave) * 64 + lane of the 10 x 24 x 4 image; pixel = s >> 2, LDS part = s & 3.  The same lane pattern fills
  // the hi image (source chunk 2 part') and the lo image (source chunk 2 part' + 1: 16 bytes further) ----
  constexpr int NDMA = 4;
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 4u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  unsigned rel[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int s = (k * 4 + wave) * 64 + lane;
    const int pix = s >> 2, part = s & 3;
    const int hy = pix / kHsHW, hx = pix - hy * kHsHW;
    const bool dead = hx >= kHsTW + 2 || hy >= kHsHH;
    rel[k] = dead ? kHsOob : (unsigned)((hy * p.W + hx) * p.ldx) * 4u + (unsigned)((part ^ ((hx >> 1) & 3)) * 32);
  }
  unsigned voff[NDMA];
  unsigned l_base = 0;
  int l_tile = sb, l_chunk = 0, l_item = 0;
  auto setup_tile = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kHsTH, x0 = tx * kHsTW;
    l_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldx) * 4u;
    const bool interior = y0 > 0 && y0 + kHsTH + 1 <= p.H && x0 > 0 && x0 + kHsTW + 1 <= p.W;
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) voff[k] = rel[k];
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int pix = ((k * 4 + wave) * 64 + lane) >> 2;
        const int hy = pix / kHsHW, hx = pix - hy * kHsHW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        voff[k] = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? rel[k] : kHsOob;
      }
    }
  };
  auto issue_dma = [&](int stage) {  // DMA of item (l_tile, l_chunk) into `stage`, then advance the loader; past the last item: nothing
    if (l_item >= nItems) return;
    unsigned char* sa = smem + stage * kHsStage;
    const unsigned soff = l_base + (unsigned)l_chunk * 128u;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)voff[k], (int)soff, 0, 0);
      // (+ 16 in the lane offset, not in the instruction's immediate: the immediate of an LDS-DMA load moves the LDS address as well)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + kHsImage + (k * 4 + wave) * 1024), 16, (int)(voff[k] + 16u), (int)soff, 0, 0);
    }
    ++l_item;
    if (++l_chunk == NCH) {
      l_chunk = 0;
      l_tile += Gs;
      if (l_item < nItems) setup_tile(l_tile);
    }
  };

  // ---- fragment reads: pixel (row iy, column lr + q), part lq -> byte lane_base[q] + iy * 24 * 64, in the hi image and (+ kHsImage) the lo image ----
  int lane_base[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) lane_base[q] = (lr + q) * 64 + ((lq ^ (((lr + q) >> 1) & 3)) * 16);

  f32x4 acc[ROWS], accl[ROWS];
#pragma unroll
  for (int o = 0; o < ROWS; ++o) acc[o] = f32x4{0.f, 0.f, 0.f, 0.f}, accl[o] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stg, int c) {
    const unsigned char* sa = smem + stg * kHsStage + row0 * (kHsHW * 64);
#pragma unroll
    for (int iy = 0; iy < ROWS + 2; ++iy) {
      if constexpr (NCH == 1) {  // registers to spare: all six fragment reads of the row in flight before its MFMAs (32 -> 32 @160^2: 418 against 460-480 us)
        u32x4 ah[3], al[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          ah[q] = *reinterpret_cast<const u32x4*>(sa + lane_base[q] + iy * (kHsHW * 64));
          al[q] = *reinterpret_cast<const u32x4*>(sa + kHsImage + lane_base[q] + iy * (kHsHW * 64));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int o = iy - r;
          if (o >= 0 && o < ROWS) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              acc[o] = Elem<f16_t>::mma(whi[c][r * 3 + q], ah[q], acc[o]);
              acc[o] = Elem<f16_t>::mma(wlo[c][r * 3 + q], ah[q], acc[o]);
              accl[o] = Elem<f16_t>::mma(whi[c][r * 3 + q], al[q], accl[o]);
            }
          }
        }
        continue;
      }
      u32x4 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(sa + lane_base[q] + iy * (kHsHW * 64));
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int o = iy - r;
        if (o >= 0 && o < ROWS) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            acc[o] = Elem<f16_t>::mma(whi[c][r * 3 + q], a[q], acc[o]);  // D[cout][pixel]
            acc[o] = Elem<f16_t>::mma(wlo[c][r * 3 + q], a[q], acc[o]);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(sa + kHsImage + lane_base[q] + iy * (kHsHW * 64));  // the lo halves, through the same registers
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int o = iy - r;
        if (o >= 0 && o < ROWS) {
#pragma unroll
          for (int q = 0; q < 3; ++q) accl[o] = Elem<f16_t>::mma(whi[c][r * 3 + q], a[q], accl[o]);
        }
      }
    }
  };

  // store offsets: lane constant (row o + (lq & 1) of the pair, column lr, the 8-channel group of couts co_w + 8 (lq >> 1)) + a scalar tile offset
  typedef __attribute__((ext_vector_type(4))) f16_t h4;
  // residual of the tile (a Bottleneck's shortcut): the lane's 4 channels are half of an 8-channel group — 8 bytes of its hi chunk, 8 of its lo chunk.
  // EARLY (one chunk per tile, registers to spare): requested before the tile's MFMAs; otherwise at the head of the epilogue.
  constexpr bool EARLY = RES && NCH == 1;
  u32x2 rh[RES ? ROWS : 1], rl[RES ? ROWS : 1];
  auto load_residual = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int xx = tx * kHsTW + lr;
#pragma unroll
    for (int o = 0; o < ROWS; ++o) {
      const int yy = ty * kHsTH + row0 + o;
      const bool ok = yy < p.H && xx < p.W;
      const unsigned off = ok ? (unsigned)(((n * p.H + yy) * p.W + xx) * p.ldres) * 4u + (unsigned)((co_w >> 3) + (lq >> 1)) * 32u + (unsigned)(lq & 1) * 8u : kHsOob;
      rh[o] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrs, off, 0, 0));
      rl[o] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rrs, off + 16u, 0, 0));
    }
  };
  auto epilogue = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kHsTH, x0 = tx * kHsTW;
    const unsigned out_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldy) * 4u;
    const bool whole = y0 + kHsTH <= p.H && x0 + kHsTW <= p.W;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + co_w + lq * 4);
    const f32x4 scl4 = *reinterpret_cast<const f32x4*>(p.wscale + co_w + lq * 4);
    if constexpr (RES && !EARLY) load_residual(tile);
    u32x2 ph[ROWS], pl[ROWS];
#pragma unroll
    for (int o = 0; o < ROWS; ++o) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (acc[o][e] + accl[o][e] * kSplitInv) * scl4[e] + bias4[e];
      if (p.act == DY_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
      }
      if constexpr (RES) {
        const h4 h = __builtin_bit_cast(h4, rh[o]), l = __builtin_bit_cast(h4, rl[o]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)h[e] + (float)l[e] * kSplitInv;
      }
      h4 hh, ll;
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // common_hip.h: split8
        const float x = __builtin_fminf(__builtin_fmaxf(v[e], -65504.f), 65504.f);
        const f16_t t = __builtin_fabsf(x) < 6.103515625e-5f ? (f16_t)0.f : (f16_t)x;
        hh[e] = t;
        ll[e] = (f16_t)((x - (float)t) * kSplitScale);
      }
      ph[o] = __builtin_bit_cast(u32x2, hh);
      pl[o] = __builtin_bit_cast(u32x2, ll);
      acc[o] = f32x4{0.f, 0.f, 0.f, 0.f};
      accl[o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int o = 0; o < ROWS; o += 2) {
      // v_permlane16_swap between the rows of a pair: quarter lq gets channels 8 (lq >> 1) .. + 7 of row o + (lq & 1), hi and lo alike
      const auto hx = __builtin_amdgcn_permlane16_swap(ph[o][0], ph[o + 1][0], false, false);
      const auto hy = __builtin_amdgcn_permlane16_swap(ph[o][1], ph[o + 1][1], false, false);
      const auto lx = __builtin_amdgcn_permlane16_swap(pl[o][0], pl[o + 1][0], false, false);
      const auto ly = __builtin_amdgcn_permlane16_swap(pl[o][1], pl[o + 1][1], false, false);
      unsigned off = (unsigned)(((row0 + o + (lq & 1)) * p.W + lr) * p.ldy) * 4u + (unsigned)((co_w >> 3) + (lq >> 1)) * 32u;
      if (!whole) off = (y0 + row0 + o + (lq & 1) < p.H && x0 + lr < p.W) ? off : kHsOob;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{hx[0], hy[0], hx[1], hy[1]}, yrs, off, (int)out_base, 0);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{lx[0], ly[0], lx[1], ly[1]}, yrs, off + 16u, (int)out_base, 0);
    }
  };

  // ---- item pipeline: two stages, the next item's DMA in flight during this item's MFMAs, one barrier per item ----
  setup_tile(l_tile);
  issue_dma(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int c_tile = sb, stage = 0;
  for (int it = 0; it < nItems; it += NCH) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      issue_dma(stage ^ 1);  // the other stage was last read one item ago: every wave has passed that item's barrier
      if constexpr (EARLY) load_residual(c_tile);
      compute(stage, c);
      if (c == NCH - 1) {
        epilogue(c_tile);
        c_tile += Gs;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next item have landed (and its stores are out)
      __syncthreads();
      stage ^= 1;
    }
  }
}

template <int NCH, int NF>
static int launch_hsplit(const HsArgs& a, hipStream_t st) {
  HsArgs p = a;
  int grid = 256 * 2;  // two 256-thread workgroups per CU (64 KB of LDS each)
  const long long nwork = (long long)p.nSpatial * p.tilesN;
  if (nwork < grid) grid = (int)nwork;
  const int q = 8 * p.tilesN;
  grid = (grid + q - 1) / q * q;  // the XCD remap and the fixed cout group per block need G % (8 * tilesN) == 0
  if (p.res != nullptr) hipLaunchKernelGGL((conv3x3_hsplit_kernel<NCH, NF, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv3x3_hsplit_kernel<NCH, NF, false>), dim3((unsigned)grid), dim3(256), 0, st, p);
  return check_launch("conv3x3_hsplit_kernel");
}

// Returns 1 when the call is not one this kernel is built for (the caller then runs the flat-K kernel), else the launch status.
int conv3x3_hsplit_try(const dy_conv_desc* d, hipStream_t st) {
  static const int off = dy_ablate("DYOLO_NO_HSPLIT");
  if (off) return 1;
  if (d->dtype != DY_F16X2 || d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->groups > 1 || d->up2x || d->x2 || d->out_f32 || d->y_dtype1 || d->w_layout != DY_WLAYOUT_ROWS) return 1;
  if ((d->cin != 32 && d->cin != 64) || d->cout % 32 != 0 || d->cout > 256 || !d->w_scale) return 1;
  if (d->residual && d->cin == 64) return 1;  // (144 weight registers + the residual's: 66 spilled, 64 -> 64 @80^2 +res 590-630 us against the flat-K kernel's 540-560)
  if (d->ho != d->h || d->wo != d->w_in) return 1;
  const long long xb = (long long)d->batch * d->h * d->w_in * d->ld_x * 4, yb = (long long)d->batch * d->ho * d->wo * d->ld_y * 4;
  const long long rb = d->residual ? (long long)d->batch * d->ho * d->wo * d->ld_res * 4 : 0;
  const long long lim = (1ll << 32) - (1ll << 20);
  const long long pre = (long long)(d->w_in + 1) * d->ld_x * 4;
  if (xb + pre >= lim || yb >= lim || rb >= lim) return 1;  // 32-bit byte offsets / buffer descriptors
  // measured at B = 256 against the flat-K kernel: 32 -> 32 @160^2 779 -> 418 us, 64 -> 64 @80^2 470 -> 410, 64 -> 128 @160^2 3213 -> 2877;
  // 64 -> 64 @40^2 128 -> 146 and @20^2 45 -> 65 (few tiles per workgroup: the weight load and the serial item pipeline show): large maps only
  if ((long long)d->ho * d->wo < 3200) return 1;
  if (d->ld_x % 8 || d->ld_y % 8 || (reinterpret_cast<uintptr_t>(d->x) & 15) || (reinterpret_cast<uintptr_t>(d->y) & 15) ||
      (d->residual && (d->ld_res % 8 || (reinterpret_cast<uintptr_t>(d->residual) & 15))) || (reinterpret_cast<uintptr_t>(d->w) & 15) || (reinterpret_cast<uintptr_t>(d->w_scale) & 15) ||
      (reinterpret_cast<uintptr_t>(d->bias) & 15))
    return 1;
  const int nf = d->cout % 64 == 0 ? 4 : 2;
  HsArgs a{};
  a.x = d->x, a.w = d->w, a.bias = d->bias, a.wscale = d->w_scale, a.res = d->residual, a.y = d->y;
  a.N = d->batch, a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.Cout = d->cout, a.ldy = d->ld_y, a.ldres = d->ld_res, a.act = d->act, a.Kpad = d->k_pad;
  a.tilesX = (d->wo + kHsTW - 1) / kHsTW;
  a.tilesY = (d->ho + kHsTH - 1) / kHsTH;
  a.tilesN = d->cout / (nf * 16);
  a.nSpatial = d->batch * a.tilesY * a.tilesX;
  a.x_bytes = (unsigned)xb, a.y_bytes = (unsigned)yb, a.r_bytes = (unsigned)rb;
  if (d->cin == 32) return nf == 4 ? launch_hsplit<1, 4>(a, st) : launch_hsplit<1, 2>(a, st);
  return nf == 4 ? launch_hsplit<2, 4>(a, st) : launch_hsplit<2, 2>(a, st);
}

}  // namespace dy
