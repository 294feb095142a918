// 3x3 stride-1 pad-1 NHWC convolution for the narrow 16-bit layers (cin 32 / 64, cout a multiple of 64) with the
// weights held in REGISTERS: the kernel the 64->64 layers of the model run on (the P2 Detect trunks, the stride-8 C2f
// Bottlenecks: 18 launches, the largest share of a pass).
//
// Why another 3x3 kernel.  conv3x3_halo.hip keeps the layer's weights stationary in LDS (72 KB for 64->64) and every
// wave re-reads them per tap: 0.75 LDS reads per MFMA, one 512-thread workgroup per CU (LDS capacity), two waves per
// SIMD that run in barrier lock-step.  Its counters (profiles/r01_pmc_halo_64x64_3x3_80.txt) show no unit saturated:
// matrix pipe 37 % busy, 37 % of the wave cycles parked in s_waitcnt / s_barrier, 27 % of the LDS cycles lost to bank
// conflicts of the 80-byte pixel pitch.  This kernel changes the decomposition instead of the schedule:
//
//   * a wave owns ONE 16-cout fragment and ALL pixels of the workgroup's 8 x 16 output tile; its weights — 9 taps x
//     NCH chunks of 32 channels x one MFMA A fragment = 36 / 72 VGPRs — are loaded once per workgroup lifetime;
//   * only the halo patch lives in LDS (10 x 18 pixels x 64 B per chunk, a ring of three 16 KB stages), so three 256-thread
//     workgroups fit a CU: three INDEPENDENT waves per SIMD, each at its own point of its item, instead of two in lock-step;
//   * per 32-channel chunk a wave walks the 10 halo rows once: 3 fragment reads (the three column shifts) feed up to 9
//     MFMAs (3 kernel rows x 3 columns) — 30 ds_read_b128 per 72 MFMAs = 0.42 reads per MFMA instead of 0.75;
//   * the halo image has a 64-byte pixel pitch with the 16-byte part index XOR-ed with ((column >> 1) & 3): every
//     fragment read (16 pixels x 4 parts, any of the three column shifts) is bank-conflict free (4 LDS cycles, the
//     minimum for ds_read_b128; the 80-byte pitch took 8) and all read addresses are `lane base[q] + immediate`;
//   * the image is lane-linear per wave-instruction, so it is filled by LDS-DMA (global_load_lds_dwordx4, the swizzle
//     applied to the SOURCE address): no staging registers, no ds_write pass; zero padding comes from a zero page;
//   * MFMA with the weight fragment as the A operand: a lane ends up with 4 consecutive couts of one pixel and stores
//     them (bias-initialised accumulator, SiLU, optional Bottleneck residual) as 8 bytes straight from registers.
//
// Persistent workgroups, XCD-aware tile order (the tiles an XCD's workgroups visit are contiguous in the image, so the
// halo columns / rows shared by neighbouring tiles are L2 hits), one raw s_barrier per (tile, chunk) item.  TWO items' DMA
// stay in flight per workgroup behind counted s_waitcnt vmcnt(N) (the first version had one and waited vmcnt(0), output
// stores included: 46 KB in flight per CU made the kernel latency bound — 2.7 TB/s x 4.4 us = all that was in flight).
// Reference semantics: Conv (nn/modules/conv.py:37-55, BatchNorm folded), Bottleneck shortcut (block.py:337-350).
#include "common_hip.h"
#include "conv_args.h"

namespace DY_NS {


struct HregArgs {
  const void* x;
  const void* w;       // DY_WLAYOUT_HALO3X3 packing with NF = 4: 1 KB blocks [(nt * nChunks + chunk) * 9 + tap][j], lane-major
  const float* bias;   // cout_pad floats
  const void* res;
  void* y;
  int N, H, W, Cin, ldx, Cout, ldy, ldres, act;
  int tilesX, tilesY, tilesN, nSpatial;  // spatial tiles (n, ty, tx) and 64-cout groups
  unsigned x_bytes, y_bytes, r_bytes;
  double* stats;  // optional: a dy_bn_train_fwd workspace; spatial block sb stores its per-channel sum / sum of squares of the STORED outputs in slot sb (dy_conv_desc.bn_stats)
  const float* bnb_mean;  // BNB (dy_conv_desc.bnb_z, which travels as `res`): the BatchNorm in front of the layer whose gradient this launch computes
  const float* bnb_rstd;
  const float* bnb_gamma;
  const float* bnb_beta;
  int bnb_act;
  int dbg;  // -DDYOLO_ABLATE builds only (DYOLO_DBG): 1 no output stores, 2 no MFMAs, 4 no DMA after the prologue, 8 no fragment reads
};

#ifdef DYOLO_ABLATE
#define HR_DBG(bit) (p.dbg & (bit))
#else
#define HR_DBG(bit) 0
#endif

constexpr int kHrTH = 8, kHrTW = 16, kHrHH = 10, kHrHW = 24;  // 10 x 18 halo pixels, rows padded to 24 (swizzle independent of the row)
constexpr int kHrStage = 16 * 1024;  // one (tile, chunk) halo image: 10 x 24 x 64 = 15,360 B, padded to the 16 wave-instructions (4 per wave) that fill it
constexpr int kHrStages = 3;

// STATS: training forward (the convolution in front of a train-mode BatchNorm, conv.py:49-51): the batch statistics of the stored
// output come out of the epilogue -- a lane always holds the same four output channels, so it keeps their sums over all of its tiles
// in 8 registers and the kernel ends with one shuffle reduction over the 16 pixel lanes and 32 plain stores per wave into the
// workgroup's slot of the BatchNorm workspace (dy_bn_train_fwd adds the slots up as it does for its own reduction pass; atomics on
// the 2 x Cout totals from ~770 workgroups serialise: +0.7 ms per step, measured) -- instead of a separate pass that reads the whole
// map again (dy_bn_train_fwd's reduction: 1.4 of 10 ms of BatchNorm per step at B = 64).
// BNB (r05; training backward, dy_conv_desc.bnb_z): the launch computes the gradient dy that reaches the BatchNorm + activation of the layer
// in front, and that BatchNorm's backward needs sum(du) and sum(du xhat) over the batch before it can form dz (du = dy act'(u), u = gamma xhat
// + beta, xhat = (z - mean) rstd) -- a pass of its own over dy and z (dy_bn_train_bwd's reduction: 2.8 of 8.9 ms of BatchNorm per step at
// B = 64).  Here the tile's z arrives the way a residual does (16 bytes per lane and row pair, requested an item ahead), the epilogue forms du
// from the STORED dy, and the sums leave through the STATS slots.  A gradient convolution has no activation, so the epilogue's vector issue
// is free where the forward kernel spends it on SiLU.
// The four BatchNorm constants per channel (rstd, -mean rstd, gamma rstd, beta - gamma rstd mean) wait in 1 KB of LDS and are read in the
// epilogue only: in registers they cost the main loop 16 of its 168 (64 channels) / 256 (128 channels) and spilled 38 / 22.
// BW: workgroups per CU the 64-channel BNB form is compiled for (3: 168 registers, 29 spilled; 2: 198, none).
template <typename T, int NCH, bool RES, bool STATS = false, bool BNB = false, int BW = 3>
__global__ __launch_bounds__(256, NCH <= 2 ? (BNB ? BW : 3) : 2) void conv3x3_hreg_kernel(const HregArgs p) {
  static_assert(!(BNB && (RES || STATS)), "BNB: its own mode");
  constexpr bool SLOTS = STATS || BNB, RLOAD = RES || BNB;
  constexpr int EPC = Elem<T>::EPC;  // 8
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kHrStages * kHrStage];
  __shared__ f32x4 bnc[BNB ? 64 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), 0, p.res ? p.r_bytes : 0u, 0x00020000);

  // block -> (cout group, spatial sequence).  Blocks b and b + 8 share an XCD (guide T1): logical id = xcd * (G/8) + b/8, so an
  // XCD's workgroups walk contiguous tiles.  The host makes G a multiple of 8 * tilesN.
  const int G = (int)gridDim.x;
  const int logical = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const int nt = logical % p.tilesN;
  const int sb = logical / p.tilesN, Gs = G / p.tilesN;
  const int myTiles = sb < p.nSpatial ? (p.nSpatial - sb + Gs - 1) / Gs : 0;
  if constexpr (SLOTS) {
    if (blockIdx.x == 0)  // the totals the BatchNorm's partial-sum launch adds into
      for (int i = tid; i < 2 * p.Cout; i += 256) p.stats[i] = 0.0;
    if (myTiles <= 0 && tid < 128) {  // a slot is summed whether its workgroup had tiles or not
      const int co = nt * 64 + (tid & 63);
      if (co < p.Cout) p.stats[(size_t)(1 + sb) * 2 * p.Cout + (tid >> 6) * p.Cout + co] = 0.0;
    }
  }
  if (myTiles <= 0) return;
  const int nItems = myTiles * NCH;

  // ---- this wave's weights: 16 couts (fragment `wave` of the 64-cout group) x all taps x all chunks, in registers ----
  u32x4 wreg[NCH][9];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.w) + (size_t)nt * NCH * 9 * 4 * 64;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[c][t] = wg[((c * 9 + t) * 4 + wave) * 64 + lane];
  }
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + nt * 64 + wave * 16 + lq * 4);

  // ---- loader: slot s = (k * 4 + wave) * 64 + lane of the 10 x 24 x 4 image; pixel = s >> 2, LDS part = s & 3 ----
  // r03: addressing without per-tile vector arithmetic.  The source is a BUFFER (descriptor shifted back by one image row + one pixel,
  // so that every offset below is non-negative): a lane's byte offset inside a tile's halo, rel[k] = ((hy W + hx) ldx + part') * 2, is a
  // constant of the launch; the tile contributes a SCALAR offset (the DMA instruction's soffset).  Interior tiles use rel[k] as it is;
  // border tiles replace the out-of-image slots by an out-of-range offset — the range check then feeds zeros (the zero padding) — so the
  // 64-bit address sums, the integer multiplies per slot (quarter-rate instructions) and the zero-page selects of the first version are gone.
  constexpr int NDMA = 4;  // every wave issues exactly 4 wave-instructions per item (w, w + 4, w + 8, w + 12): the waits below are counted
  constexpr unsigned kOob = 0xfffffff0u;  // >= num_records of every descriptor here (the host checks the sizes)
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 2u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  unsigned rel[NDMA];  // launch constants
  int hyx[NDMA];       // hy | hx << 8 of the slot's halo pixel
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int s = (k * 4 + wave) * 64 + lane;
    const int pix = s >> 2, part = s & 3;
    const int hy = pix / kHrHW, hx = pix - hy * kHrHW;
    const bool dead = hx >= kHrTW + 2 || hy >= kHrHH;  // the row padding (hx >= 18) and the stage's padding (hy == 10): zeros for ever
    rel[k] = dead ? kOob : (unsigned)((hy * p.W + hx) * p.ldx + (part ^ ((hx >> 1) & 3)) * EPC) * 2u;
    hyx[k] = hy | (hx << 8);
  }
  unsigned voff[NDMA];  // this lane's offsets for the loader's current tile
  unsigned l_base = 0;  // scalar: byte offset of pixel (ty * 8, tx * 16) of image n, in the shifted descriptor's terms the tile's halo origin
  int l_tile = sb, l_chunk = 0, l_item = 0;
  auto setup_tile = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kHrTH, x0 = tx * kHrTW;
    l_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldx) * 2u;
    const bool interior = y0 > 0 && y0 + kHrTH + 1 <= p.H && x0 > 0 && x0 + kHrTW + 1 <= p.W;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) voff[k] = rel[k];
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int gy = y0 - 1 + (hyx[k] & 255), gx = x0 - 1 + (hyx[k] >> 8);
        voff[k] = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? rel[k] : kOob;
      }
    }
  };
  auto issue_dma = [&](int stage) {  // DMA of item (l_tile, l_chunk) into `stage`, then advance the loader; past the last item: zeros
    unsigned char* sa = smem + stage * kHrStage;
    const bool live = l_item < nItems && !(HR_DBG(4) && l_item >= 2);
    if (live) {
      const unsigned soff = l_base + (unsigned)l_chunk * (4u * EPC * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)voff[k], (int)soff, 0, 0);
      ++l_item;
      if (++l_chunk == NCH) {
        l_chunk = 0;
        l_tile += Gs;
        if (l_item < nItems) setup_tile(l_tile);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k)  // keeps the per-item instruction count (the counted waits) — every lane out of range: zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)kOob, 0, 0, 0);
    }
  };

  // ---- fragment reads: pixel (row iy, column lr + q), part lq  ->  byte lane_base[q] + iy * 24 * 64 ----
  int lane_base[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) lane_base[q] = (lr + q) * 64 + ((lq ^ (((lr + q) >> 1) & 3)) * 16);

  f32x4 acc[kHrTH];
#pragma unroll
  for (int o = 0; o < kHrTH; ++o) acc[o] = bias4;

  auto compute = [&](int stg, int c) {
    const unsigned char* sa = smem + stg * kHrStage;
#pragma unroll
    for (int iy = 0; iy < kHrHH; ++iy) {
      u32x4 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(sa + lane_base[q] + (HR_DBG(8) ? 0 : iy * (kHrHW * 64)));
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int o = iy - r;
        if (o >= 0 && o < kHrTH) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            if (HR_DBG(2)) asm volatile("" ::"v"(a[q]));
            else acc[o] = Elem<T>::mma(wreg[c][r * 3 + q], a[q], acc[o]);  // D[cout][pixel]
          }
        }
      }
    }
  };

  // residual of the tile that ends with this item (Bottleneck shortcut): requested at the START of the item, BEFORE the next
  // DMA is issued — the loads are then older than that DMA and their wait (at the epilogue) leaves it in flight
  typedef __attribute__((ext_vector_type(4))) T t4;
  u32x4 rl[kHrTH / 2];  // 16 bytes per lane and row pair, in the store order of the epilogue (quarter lq: row o + (lq & 1), channels 8 (lq >> 1) ..)
  auto load_residual = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int xx = tx * kHrTW + lr;
    const int co16 = nt * 64 + wave * 16 + (lq >> 1) * 8;
#pragma unroll
    for (int o = 0; o < kHrTH; o += 2) {
      const int yy = ty * kHrTH + o + (lq & 1);
      const bool ok = yy < p.H && xx < p.W;
      const unsigned off = ok ? (unsigned)((((size_t)(n * p.H + yy) * p.W + xx) * (size_t)p.ldres + co16) * sizeof(T)) : 0xfffffff0u;
      rl[o / 2] = __builtin_amdgcn_raw_buffer_load_b128(rrs, off, 0, 0);
    }
  };
  // store offsets: lane constant (row o + (lq & 1) of the pair, column lr, 8 channels from co16) + a scalar tile offset (soffset)
  unsigned lane_out[kHrTH / 2];
  {
    const int co16 = nt * 64 + wave * 16 + (lq >> 1) * 8;
#pragma unroll
    for (int o = 0; o < kHrTH; o += 2) lane_out[o / 2] = (unsigned)(((o + (lq & 1)) * p.W + lr) * p.ldy + co16) * (unsigned)sizeof(T);
  }
  float st_sum[4] = {0.f, 0.f, 0.f, 0.f}, st_sq[4] = {0.f, 0.f, 0.f, 0.f};  // STATS: this lane's four channels, all its tiles
  if constexpr (BNB) {  // (published by the barrier that opens the item pipeline)
    if (tid < 64) {
      const int co = nt * 64 + tid;
      const float mu = p.bnb_mean[co], rs = p.bnb_rstd[co], gr = p.bnb_gamma[co] * rs;
      bnc[tid] = f32x4{rs, -mu * rs, gr, p.bnb_beta[co] - gr * mu};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  auto epilogue = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kHrTH, x0 = tx * kHrTW;
    const unsigned out_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldy) * (unsigned)sizeof(T);  // scalar
    const bool whole = y0 + kHrTH <= p.H && x0 + kHrTW <= p.W;                                         // wave-uniform: no ragged edge
    // A result lane holds 8 bytes (4 channels) of pixel lr in tile row o.  Stored like that, each of the 64 lanes is its own
    // L1 request.  v_permlane16_swap between the rows of a pair (o, o + 1) leaves 16 contiguous bytes in every lane - quarter lq
    // gets channels 8 (lq >> 1) .. + 7 of row o + (lq & 1) - so a pair of rows leaves in one 16-byte store instead of two 8-byte ones.
    u32x2 pk[kHrTH];
#pragma unroll
    for (int o = 0; o < kHrTH; ++o) {
      float v[4] = {acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
      if (p.act == DY_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
      }
      if constexpr (RES) {  // the row pair's 16-byte pieces back in result-lane order: the store-side swap run backwards
        const auto sx = __builtin_amdgcn_permlane16_swap(rl[o / 2][0], rl[o / 2][2], false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(rl[o / 2][1], rl[o / 2][3], false, false);
        const t4 rr = __builtin_bit_cast(t4, u32x2{sx[o & 1], sy[o & 1]});
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rr[e]);
      }
      t4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = Elem<T>::from_f32(v[e]);
      if constexpr (STATS) {
        if (whole || (y0 + o < p.H && x0 + lr < p.W)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = Elem<T>::to_f32(ov[e]);  // what BatchNorm will read back
            st_sum[e] += f, st_sq[e] += f * f;
          }
        }
      }
      if constexpr (BNB) {
        const auto sx = __builtin_amdgcn_permlane16_swap(rl[o / 2][0], rl[o / 2][2], false, false);  // z of this lane's pixel and channels, as RES reads its residual
        const auto sy = __builtin_amdgcn_permlane16_swap(rl[o / 2][1], rl[o / 2][3], false, false);
        const t4 zz = __builtin_bit_cast(t4, u32x2{sx[o & 1], sy[o & 1]});
        if (whole || (y0 + o < p.H && x0 + lr < p.W)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 k = bnc[wave * 16 + lq * 4 + e];
            const float zf = Elem<T>::to_f32(zz[e]);
            const float xh = zf * k[0] + k[1];
            float du = Elem<T>::to_f32(ov[e]);  // the gradient as the BatchNorm's apply pass will read it back
            if (p.bnb_act == DY_ACT_SILU) {
              const float u = zf * k[2] + k[3];
              const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u * -1.4426950408889634f));
              du *= sg * (1.0f + u * (1.0f - sg));  // bn_train.hip: silu_grad
            }
            st_sum[e] += du, st_sq[e] += du * xh;
          }
        }
      }
      pk[o] = __builtin_bit_cast(u32x2, ov);
      acc[o] = bias4;
    }
#pragma unroll
    for (int o = 0; o < kHrTH; o += 2) {
      const auto sx = __builtin_amdgcn_permlane16_swap(pk[o][0], pk[o + 1][0], false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(pk[o][1], pk[o + 1][1], false, false);
      unsigned off = lane_out[o / 2];
      if (!whole) off = (y0 + o + (lq & 1) < p.H && x0 + lr < p.W) ? off : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{sx[0], sy[0], sx[1], sy[1]}, yrs, HR_DBG(1) ? kOob : off, (int)out_base, 0);
    }
  };

  // ---- item pipeline: ring of three stages, two items' DMA in flight, counted waits, raw barriers ----
  // Item i lives in stage i % 3.  At the start of item i the DMA of item i + 2 goes into stage (i + 2) % 3, last read in
  // item i - 1 (every wave has passed the barrier that ended it).  At the end of item i the wave waits until ITS pieces of
  // item i + 1 have landed: younger than those are only the 4 DMA instructions of item i + 2 and, when the item ended a
  // tile, the tile's 4 output stores -> s_waitcnt vmcnt(4) / vmcnt(8); the barrier then publishes everyone's pieces.
  setup_tile(l_tile);
  issue_dma(0);
  issue_dma(1 % kHrStages);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int c_tile = sb;
  int stage = 0;
  for (int it = 0; it < nItems; it += NCH) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (RLOAD) {
        if (c == NCH - 1) load_residual(c_tile);
      }
      issue_dma(stage + 2 >= kHrStages ? stage + 2 - kHrStages : stage + 2);
      compute(stage, c);
      if (c == NCH - 1) {
        epilogue(c_tile);
        c_tile += Gs;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      stage = stage + 1 == kHrStages ? 0 : stage + 1;
    }
  }
  if constexpr (SLOTS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = st_sum[e], b = st_sq[e];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);  // over the 16 pixel lanes of a quarter
      const int co = nt * 64 + wave * 16 + lq * 4 + e;
      if (lr == 0 && co < p.Cout) {
        double* slot = p.stats + (size_t)(1 + sb) * 2 * p.Cout;
        slot[co] = (double)a, slot[p.Cout + co] = (double)b;
      }
    }
  }
}

// ---- stride 2 (r03): the same decomposition for the 64-channel DOWNSAMPLING layers (64->128 @160 of the backbone, 64->64 @160 of
// the neck), which ran on the flat-M LDS-DMA GEMM with a per-tap gather (399 / 597 TFLOP/s: every input pixel crossed the L2->LDS path
// 2.25 times and a 128 x 64 tile with K = 576 is nine short steps between a prologue and an epilogue).  Output tile 4 x 16; the halo is
// 9 rows x 33 columns, kept as TWO column-parity planes per row (A: columns 0, 2, .., 32 of the halo, B: columns 1, 3, .., 31), so
// that the three taps of an output column are unit-stride fragment reads again — A[j], B[j], A[j + 1] — with the conflict-free
// swizzle of the stride-1 image (row pitch 40 slots = 2,560 B, plane B at slot 24: both multiples of 256 B).  The LDS-DMA fills the
// planes through its per-lane SOURCE address (buffer offsets, zeros by range check); per chunk a wave walks the 9 halo rows: even
// rows feed the kernel rows r = 0 and r = 2 of two output rows (6 MFMAs per 3 reads), odd rows r = 1 (3 per 3): 36 MFMAs per 27
// reads.  Ring of three 24 KB stages, two workgroups per CU.
constexpr int kH2TH = 4, kH2TW = 16, kH2HH = 9, kH2Pitch = 40, kH2PlaneB = 24;
constexpr int kH2Stage = 24 * 1024;  // 9 x 40 slots x 64 B = 23,040 B, padded to 24 wave-instructions (6 per wave)

template <typename T, int NCH, bool STATS = false>  // STATS: as in the stride-1 kernel (r04)
__global__ __launch_bounds__(256, 2) void conv3x3_hreg_s2_kernel(const HregArgs p) {
  constexpr int EPC = Elem<T>::EPC;  // 8
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kHrStages * kH2Stage];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
  const int Ho = (p.H - 1) / 2 + 1, Wo = (p.W - 1) / 2 + 1;

  const int G = (int)gridDim.x;
  const int logical = ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3);
  const int nt = logical % p.tilesN;
  const int sb = logical / p.tilesN, Gs = G / p.tilesN;
  const int myTiles = sb < p.nSpatial ? (p.nSpatial - sb + Gs - 1) / Gs : 0;
  if constexpr (STATS) {
    if (blockIdx.x == 0)  // the totals the BatchNorm's partial-sum launch adds into
      for (int i = tid; i < 2 * p.Cout; i += 256) p.stats[i] = 0.0;
    if (myTiles <= 0 && tid < 128) {  // a slot is summed whether its workgroup had tiles or not
      const int co = nt * 64 + (tid & 63);
      if (co < p.Cout) p.stats[(size_t)(1 + sb) * 2 * p.Cout + (tid >> 6) * p.Cout + co] = 0.0;
    }
  }
  if (myTiles <= 0) return;
  const int nItems = myTiles * NCH;

  u32x4 wreg[NCH][9];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(p.w) + (size_t)nt * NCH * 9 * 4 * 64;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[c][t] = wg[((c * 9 + t) * 4 + wave) * 64 + lane];
  }
  const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + nt * 64 + wave * 16 + lq * 4);

  // ---- loader: slot = (k * 4 + wave) * 16 + (lane >> 2) of the 9 x 40 image, part = lane & 3 ----
  constexpr int NDMA = 6;
  constexpr unsigned kOob = 0xfffffff0u;
  const unsigned pre = (unsigned)((p.W + 1) * p.ldx) * 2u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre, 0, p.x_bytes + pre, 0x00020000);
  unsigned rel[NDMA];
  int hyx[NDMA];  // hy | hx << 8 (halo coordinates of the slot's pixel)
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int s = (k * 4 + wave) * 64 + lane;
    const int slot = s >> 2, part = s & 3;
    const int hy = slot / kH2Pitch, c = slot - hy * kH2Pitch;
    const bool planeB = c >= kH2PlaneB;
    const int ci = planeB ? c - kH2PlaneB : c;          // column index inside the plane (the swizzle key)
    const int hx = planeB ? 2 * ci + 1 : 2 * ci;         // halo column
    const bool dead = hy >= kH2HH || hx > 2 * kH2TW;     // stage padding, plane padding
    rel[k] = dead ? kOob : (unsigned)((hy * p.W + hx) * p.ldx + (part ^ ((ci >> 1) & 3)) * EPC) * 2u;
    hyx[k] = hy | (hx << 8);
  }
  unsigned voff[NDMA];
  unsigned l_base = 0;
  int l_tile = sb, l_chunk = 0, l_item = 0;
  auto setup_tile = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = 2 * ty * kH2TH, x0 = 2 * tx * kH2TW;  // input coordinates of the tile's first output pixel's centre tap
    l_base = (unsigned)(((n * p.H + y0) * p.W + x0) * p.ldx) * 2u;
    const bool interior = y0 > 0 && y0 - 1 + kH2HH <= p.H && x0 > 0 && x0 + 2 * kH2TW <= p.W;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) voff[k] = rel[k];
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int gy = y0 - 1 + (hyx[k] & 255), gx = x0 - 1 + (hyx[k] >> 8);
        voff[k] = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? rel[k] : kOob;
      }
    }
  };
  auto issue_dma = [&](int stage) {
    unsigned char* sa = smem + stage * kH2Stage;
    const bool live = l_item < nItems;
    if (live) {
      const unsigned soff = l_base + (unsigned)l_chunk * (4u * EPC * (unsigned)sizeof(T));
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)voff[k], (int)soff, 0, 0);
      ++l_item;
      if (++l_chunk == NCH) {
        l_chunk = 0;
        l_tile += Gs;
        if (l_item < nItems) setup_tile(l_tile);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NDMA; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (k * 4 + wave) * 1024), 16, (int)kOob, 0, 0, 0);
    }
  };

  // fragment reads of halo row iy: q = 0 -> plane A column lr, q = 1 -> plane B column lr, q = 2 -> plane A column lr + 1
  int lane_base[3];
  lane_base[0] = lr * 64 + ((lq ^ ((lr >> 1) & 3)) * 16);
  lane_base[1] = (kH2PlaneB + lr) * 64 + ((lq ^ ((lr >> 1) & 3)) * 16);
  lane_base[2] = (lr + 1) * 64 + ((lq ^ (((lr + 1) >> 1) & 3)) * 16);

  f32x4 acc[kH2TH];
#pragma unroll
  for (int o = 0; o < kH2TH; ++o) acc[o] = bias4;

  auto compute = [&](int stg, int c) {
    const unsigned char* sa = smem + stg * kH2Stage;
#pragma unroll
    for (int iy = 0; iy < kH2HH; ++iy) {
      u32x4 a[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(sa + lane_base[q] + iy * (kH2Pitch * 64));
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if ((iy - r) % 2 == 0) {
          const int o = (iy - r) / 2;
          if (iy - r >= 0 && o < kH2TH) {
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[o] = Elem<T>::mma(wreg[c][r * 3 + q], a[q], acc[o]);  // D[cout][pixel]
          }
        }
      }
    }
  };

  typedef __attribute__((ext_vector_type(4))) T t4;
  unsigned lane_out[kH2TH / 2];
  {
    const int co16 = nt * 64 + wave * 16 + (lq >> 1) * 8;
#pragma unroll
    for (int o = 0; o < kH2TH; o += 2) lane_out[o / 2] = (unsigned)(((o + (lq & 1)) * Wo + lr) * p.ldy + co16) * (unsigned)sizeof(T);
  }
  float st_sum[4] = {0.f, 0.f, 0.f, 0.f}, st_sq[4] = {0.f, 0.f, 0.f, 0.f};  // STATS: this lane's four channels, all its tiles
  auto epilogue = [&](int tile) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    const int ty = r % p.tilesY, n = r / p.tilesY;
    const int y0 = ty * kH2TH, x0 = tx * kH2TW;  // output coordinates
    const unsigned out_base = (unsigned)(((n * Ho + y0) * Wo + x0) * p.ldy) * (unsigned)sizeof(T);
    const bool whole = y0 + kH2TH <= Ho && x0 + kH2TW <= Wo;
    u32x2 pk[kH2TH];
#pragma unroll
    for (int o = 0; o < kH2TH; ++o) {
      float v[4] = {acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
      if (p.act == DY_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
      }
      t4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = Elem<T>::from_f32(v[e]);
      pk[o] = __builtin_bit_cast(u32x2, ov);
      if constexpr (STATS) {
        if (whole || (y0 + o < Ho && x0 + lr < Wo)) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = Elem<T>::to_f32(ov[e]);  // what BatchNorm will read back
            st_sum[e] += f, st_sq[e] += f * f;
          }
        }
      }
      acc[o] = bias4;
    }
#pragma unroll
    for (int o = 0; o < kH2TH; o += 2) {
      const auto sx = __builtin_amdgcn_permlane16_swap(pk[o][0], pk[o + 1][0], false, false);
      const auto sy = __builtin_amdgcn_permlane16_swap(pk[o][1], pk[o + 1][1], false, false);
      unsigned off = lane_out[o / 2];
      if (!whole) off = (y0 + o + (lq & 1) < Ho && x0 + lr < Wo) ? off : kOob;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{sx[0], sy[0], sx[1], sy[1]}, yrs, off, (int)out_base, 0);
    }
  };

  // item pipeline as in the stride-1 kernel: 6 DMA instructions per item and wave, 2 output stores per tile
  setup_tile(l_tile);
  issue_dma(0);
  issue_dma(1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int c_tile = sb;
  int stage = 0;
  for (int it = 0; it < nItems; it += NCH) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      issue_dma(stage + 2 >= kHrStages ? stage + 2 - kHrStages : stage + 2);
      compute(stage, c);
      if (c == NCH - 1) {
        epilogue(c_tile);
        c_tile += Gs;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      stage = stage + 1 == kHrStages ? 0 : stage + 1;
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = st_sum[e], b = st_sq[e];
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) a += __shfl_xor(a, m, 64), b += __shfl_xor(b, m, 64);  // over the 16 pixel lanes of a quarter
      const int co = nt * 64 + wave * 16 + lq * 4 + e;
      if (lr == 0 && co < p.Cout) {
        double* slot = p.stats + (size_t)(1 + sb) * 2 * p.Cout;
        slot[co] = (double)a, slot[p.Cout + co] = (double)b;
      }
    }
  }
}

[[maybe_unused]] constexpr int kBnbWgs = 2;  // measured (tools/bn_behind_ab.sh, B = 64 step): 3 per CU with 29 spills +0.2 ms, 2 per CU without -0.1 ms against the two-pass form
template <typename T>
static int launch_hreg(const HregArgs& a, hipStream_t st) {
  HregArgs p = a;
  const int nch = p.Cin / 32;
  // three 256-thread workgroups per CU (48 KB of LDS and <= 168 VGPRs each); r04, 128 input channels (four chunks: 144 weight registers):
  // two per CU with up to 256 registers
  int grid = 256 * (nch <= 2 ? 3 : 2);
  const long long nwork = (long long)p.nSpatial * p.tilesN;
  if (nwork < grid) grid = (int)nwork;
  const int q = 8 * p.tilesN;
  grid = (grid + q - 1) / q * q;  // the XCD remap and the fixed cout group per block need G % (8 * tilesN) == 0
  const bool res = p.res != nullptr;
#ifndef DYOLO_L2E_BUILD
  if (p.bnb_mean) {  // training backward: z travels as the residual view, the sums as the forward statistics do
    static const int bw = dy_ablate("DYOLO_BNB_WGS") ? dy_ablate("DYOLO_BNB_WGS") : kBnbWgs;
    if (nch == 2 && bw == 2) {
      int g2 = 256 * 2;
      if (nwork < g2) g2 = (int)nwork;
      grid = (g2 + q - 1) / q * q;
      hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 2, false, false, true, 2>), dim3((unsigned)grid), dim3(256), 0, st, p);
    } else if (nch == 2) {
      hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 2, false, false, true, 3>), dim3((unsigned)grid), dim3(256), 0, st, p);
    } else {
      hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 4, false, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    }
    note_stats(grid / p.tilesN);
    return check_launch("conv3x3_hreg_kernel<bnb>");
  }
#endif
  if (p.stats) {  // conv3x3_hreg_try admits it without a residual only
    if (nch == 1) hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 1, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else if (nch == 2) hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 2, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 4, false, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    note_stats(grid / p.tilesN);  // slots written: one per spatial block
    return check_launch("conv3x3_hreg_kernel");
  }
  if (nch == 1) {
    if (res) hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 1, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 1, false>), dim3((unsigned)grid), dim3(256), 0, st, p);
  } else if (nch == 2) {
    if (res) hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 2, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 2, false>), dim3((unsigned)grid), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((conv3x3_hreg_kernel<T, 4, false>), dim3((unsigned)grid), dim3(256), 0, st, p);  // (no residual: conv3x3_hreg_try)
  }
  return check_launch("conv3x3_hreg_kernel");
}

template <typename T>
static int launch_hreg_s2(const HregArgs& a, hipStream_t st) {
  HregArgs p = a;
  int grid = 256 * 2;  // two 256-thread workgroups per CU (72 KB of LDS each)
  const long long nwork = (long long)p.nSpatial * p.tilesN;
  if (nwork < grid) grid = (int)nwork;
  const int q = 8 * p.tilesN;
  grid = (grid + q - 1) / q * q;
#ifndef DYOLO_L2E_BUILD
  if (p.stats) {
    hipLaunchKernelGGL((conv3x3_hreg_s2_kernel<T, 2, true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    note_stats(grid / p.tilesN);  // slots written: one per spatial block (at most 512 + 8 * tilesN - 1 workgroups)
    return check_launch("conv3x3_hreg_s2_kernel");
  }
#endif
  hipLaunchKernelGGL((conv3x3_hreg_s2_kernel<T, 2>), dim3((unsigned)grid), dim3(256), 0, st, p);
  return check_launch("conv3x3_hreg_s2_kernel");
}

// Returns 1 when the shape is not one this kernel is built for (the caller then runs conv3x3_halo), else the launch status.
int conv3x3_hreg_try(const dy_conv_desc* d, hipStream_t st) {
  static const int off = dy_ablate("DYOLO_NO_HREG");
  if (off) return 1;
  if (!(d->dtype == DY_BF16 || d->dtype == DY_F16) || d->out_f32 || d->ksize != 3 || d->pad != 1 || d->groups > 1 || d->up2x || d->x2) return 1;
  if (d->stride == 2) {  // the 64-channel downsampling layers (conv3x3_hreg_s2_kernel)
    if (d->cin != 64 || d->cout % 64 != 0 || d->cout > 256 || d->residual) return 1;
    const long long xb2 = (long long)d->batch * d->h * d->w_in * d->ld_x * 2, yb2 = (long long)d->batch * d->ho * d->wo * d->ld_y * 2;
    if (xb2 >= (1ll << 31) || yb2 >= (1ll << 32) - 64 || d->ld_y % 8 || (reinterpret_cast<uintptr_t>(d->y) & 15)) return 1;
    HregArgs a{};
    a.x = d->x, a.w = d->w, a.bias = d->bias, a.res = nullptr, a.y = d->y;
    a.N = d->batch, a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.Cout = d->cout, a.ldy = d->ld_y, a.ldres = 0, a.act = d->act;
    a.tilesX = (d->wo + kH2TW - 1) / kH2TW;
    a.tilesY = (d->ho + kH2TH - 1) / kH2TH;
    a.tilesN = d->cout / 64;
    a.nSpatial = d->batch * a.tilesY * a.tilesX;
    a.x_bytes = (unsigned)xb2, a.y_bytes = (unsigned)yb2, a.r_bytes = 0;
    a.stats = (d->y_dtype1 || d->bnb_z) ? nullptr : d->bn_stats;
    return d->dtype == DY_BF16 ? launch_hreg_s2<bf16_t>(a, st) : launch_hreg_s2<f16_t>(a, st);
  }
  if (d->stride != 1) return 1;
  // measured (B = 256, alternating A/B against conv3x3_halo, tools/bench_conv.py): 64->64 @160 650 -> 598 us, @80 150 -> 141, @40 55 -> 42,
  // @20 29 -> 18, 64->128 @80 285 -> 268; cin 32 (one chunk per tile: an epilogue every item) 205-250 -> 227-252: no gain, stays on
  // the halo kernel; with a Bottleneck residual (halo -> this kernel): 8-byte gathers 210 -> 237 @80, 16-byte pieces per row pair 187 -> 203:
  // stays there too (r04, measured again in the model at B = 256 with the residual form at two workgroups per CU, no spills, and the loads
  // requested a whole item earlier: 64->64 @80 178 -> 230 us, 128->128 @40 177 -> 240 us -- a wave's residual read is 64 16-byte pieces of 32
  // pixel rows, and every in-order vmcnt wait behind it pays for that; the latency was not the cost)
  if ((d->cin != 64 && d->cin != 128) || d->cout % 64 != 0 || d->cout > 256 || d->residual) return 1;
  if (d->ho != d->h || d->wo != d->w_in) return 1;
  const long long xb = (long long)d->batch * d->h * d->w_in * d->ld_x * 2, yb = (long long)d->batch * d->ho * d->wo * d->ld_y * 2;
  const long long rb = d->residual ? (long long)d->batch * d->ho * d->wo * d->ld_res * 2 : 0;
  if (xb >= (1ll << 31) || yb >= (1ll << 32) - 64 || rb >= (1ll << 32) - 64) return 1;  // 32-bit element offsets / buffer descriptors
  if (d->ld_y % 8 || (reinterpret_cast<uintptr_t>(d->y) & 15) || (d->residual && (d->ld_res % 8 || (reinterpret_cast<uintptr_t>(d->residual) & 15)))) return 1;  // 16-byte stores / residual loads
  HregArgs a{};
  a.x = d->x, a.w = d->w, a.bias = d->bias, a.res = d->residual, a.y = d->y;
  a.N = d->batch, a.H = d->h, a.W = d->w_in, a.Cin = d->cin, a.ldx = d->ld_x, a.Cout = d->cout, a.ldy = d->ld_y, a.ldres = d->ld_res, a.act = d->act;
  a.tilesX = (d->wo + kHrTW - 1) / kHrTW;
  a.tilesY = (d->ho + kHrTH - 1) / kHrTH;
  a.tilesN = d->cout / 64;
  a.nSpatial = d->batch * a.tilesY * a.tilesX;
  a.x_bytes = (unsigned)xb, a.y_bytes = (unsigned)yb, a.r_bytes = (unsigned)rb;
  a.stats = d->bn_stats;  // (at most 768 + 8 * tilesN - 1 workgroups: the slot count stays below the workspace's 1024)
  if (d->bnb_z) {
    const long long zb = (long long)d->batch * d->ho * d->wo * d->bnb_ld_z * 2;
#ifdef DYOLO_L2E_BUILD
    const bool built = false;  // (a gradient convolution carries no activation: it never comes through the scaled-domain build)
#else
    const bool built = d->bn_stats && !d->y_dtype1 && d->act == DY_ACT_NONE && d->cin != 32 && zb < (1ll << 32) - 64 && d->bnb_ld_z % 8 == 0 && (reinterpret_cast<uintptr_t>(d->bnb_z) & 15) == 0;
#endif
    if (built) {
      a.res = d->bnb_z, a.ldres = d->bnb_ld_z, a.r_bytes = (unsigned)zb;
      a.bnb_mean = d->bnb_mean, a.bnb_rstd = d->bnb_rstd, a.bnb_gamma = d->bnb_gamma, a.bnb_beta = d->bnb_beta, a.bnb_act = d->bnb_act;
    } else {
      a.stats = nullptr;  // (the caller's BatchNorm backward runs its own reduction: dy_conv_stats_written() stays 0)
    }
  }
  a.dbg = dy_ablate("DYOLO_DBG");
  return d->dtype == DY_BF16 ? launch_hreg<bf16_t>(a, st) : launch_hreg<f16_t>(a, st);
}

}  // namespace DY_NS
