// 3x3 (pad 1, stride 1 or 2) NHWC convolution on CDNA4 MFMA: persistent workgroups, LDS halo tiles,
// weight-stationary when the weights fit.
//
// Why a second kernel: the generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every input
// pixel once per tap and spends ~8 VALU instructions of address arithmetic per MFMA.  Here a
// 512-thread workgroup (8 waves, two per SIMD so one wave's VALU epilogue overlaps the other's
// MFMAs) owns a TH x 16 patch of output pixels x BN output channels and walks the input channels
// in chunks of one MFMA k-group (32 bf16/f16 or 16 f32 channels = 64 bytes per pixel):
//
//   - the halo patch ((TH-1)*S+3) x (15*S+3) pixels of one chunk is fetched ONCE (1.27x the output
//     pixels for stride 1 instead of 9x) into an LDS image with an 80-byte pixel pitch: 16
//     consecutive pixels then land on 16 distinct 16-byte bank slots (5 is odd), and every one of
//     the nine taps reads it as `lane base + compile-time constant` — no address VALU in the loop;
//   - weights are pre-packed by the host in MFMA-fragment order per (n-tile, chunk, tap), so
//     staging them is a linear 16-byte-per-lane copy and a B fragment read is `base + const + lane*16`;
//   - WS = true (weight stationary): the n-tile's whole weight set (nChunks x 9 x BN x 64 B, e.g.
//     72 KB for 64->64) is loaded into LDS once per workgroup and only halo chunks stream
//     (double buffered) — per-CU L2->LDS bandwidth (~25 B/clk) cannot feed both operands per item;
//     WS = false streams (halo, weight) pairs per item for layers whose weights exceed LDS;
//   - workgroups are persistent: a block walks its (tile, chunk) items with the loads of item i+1
//     in flight during the MFMAs of item i, one barrier per item, no drain between tiles;
//   - the MFMA is issued with the WEIGHT fragment as the A operand, so a lane ends up holding 4
//     consecutive output channels of one pixel: the epilogue (bias, SiLU, residual) works on
//     registers and stores 8 (bf16/f16) or 16 (f32) contiguous bytes per lane, no LDS round trip.
//
// Reference semantics: Conv / RepVGGBlock (folded) / Bottleneck residual, as conv_igemm.hip.
#include "common_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace DY_NS {

struct Conv3Args {
  const void* x;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int H, W, Cin, ldx;
  int Ho, Wo, Cout, ldy, ldres;
  int act;
  int tilesX, tilesY, tilesN, nTiles, nChunks;
  unsigned x_bytes, w_bytes, y_bytes, r_bytes;  // extents of the x / y / residual views and of the packed weights (buffer descriptors)
  int dbg;  // ablation switches (DYOLO_DBG env): 1 skip global loads after the first item, 2 skip MFMAs,
            // 4 skip epilogue, 8 skip LDS staging writes, 16 force the streaming (non-WS) variant
  double* stats;  // optional: a dy_bn_train_fwd workspace (dy_conv_desc.bn_stats; see STATS below)
};

constexpr int kHaloPixPitch = 80;  // bytes per halo pixel in LDS (64 data + 16 pad)

// NCH > 0 = PIPE (weight-stationary, exactly NCH chunks per tile, 16-bit storage, SiLU, no residual): the epilogue of a
// tile is deferred into the first item of the NEXT tile and interleaved in source order with that item's MFMAs (one
// accumulator fragment's SiLU + scratch write after each tap), so its VALU / LDS / store work is issued in the shadow of
// the matrix instructions instead of stalling all eight barrier-locked waves between tiles (64->64 @80x80: 151 -> 135 us).
// STATS (r04; training forward in front of a train-mode BatchNorm, Conv3Args.stats; WS without PIPE only): a weight-stationary block
// stays on one n-tile, so a lane keeps sum / sum of squares of its NF x 4 channels over ALL its tiles; the pixel lanes meet in a
// shuffle tree, the eight waves in LDS, and block b writes its BN channels of slot b / tilesN of the dy_bn_train_fwd workspace.
template <typename T, int S, int MF, int NF, bool OUTF32, bool WS, int NCH = 0, bool STATS = false>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(const Conv3Args p) {
  constexpr bool PIPE = NCH > 0;
  static_assert(!STATS || (WS && !PIPE && !OUTF32 && sizeof(T) == 2), "STATS: weight-stationary, 16-bit outputs, plain epilogue");
  constexpr int NT = 512;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int KCE = 4 * EPC;             // channels per chunk (one MFMA k-group)
  constexpr int TH = 8 * MF, TW = 16;      // output patch of the workgroup; wave w owns rows [w*MF, (w+1)*MF)
  constexpr int HH = (TH - 1) * S + 3, HWD = (TW - 1) * S + 3;
  constexpr int NPIX = HH * HWD;
  constexpr int PP = kHaloPixPitch;
  constexpr int A_BYTES = (NPIX * PP + 15) / 16 * 16;
  constexpr int W_CHUNKS = 9 * NF * 64;    // 16-byte chunks of one (n-tile, chunk) weight block
  constexpr int W_BYTES = W_CHUNKS * 16;
  constexpr int STAGE = WS ? A_BYTES : A_BYTES + W_BYTES;
  constexpr int NA = (NPIX * 4 + NT - 1) / NT;
  constexpr int NW = WS ? 1 : (W_CHUNKS + NT - 1) / NT;
  constexpr int BN = NF * 16;
  typedef typename std::conditional<OUTF32, float, T>::type OutT;

  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];
  // WS: [all weight chunks of this n-tile][stage 0][stage 1];  streaming: [stage 0][stage 1]
  unsigned char* const stage0 = dyn_smem + (WS ? p.nChunks * W_BYTES : 0);
  // bias (all n-tiles, fp32) lives behind the stages: epilogue reads must not be global loads, or their
  // s_waitcnt vmcnt(0) would also drain the halo prefetch that is in flight
  const float* const sbias = reinterpret_cast<const float*>(stage0 + 2 * STAGE);
  // per-wave epilogue scratch (MF*16 pixels x BN channels of OutT, pixel pitch padded by 16 B): results are
  // transposed through it so that global stores are whole pixel rows, 16 bytes per lane.  Row-strided
  // 8-byte stores straight from the accumulator layout are store-issue bound (~580 cycles per wave store).
  constexpr int EP_PITCH = BN * (int)sizeof(OutT) + 16;
  constexpr int EP_BYTES = MF * 16 * EP_PITCH;
  // PIPE keeps the scratch in a SEPARATE LDS object: only then can the compiler prove that the epilogue's scratch
  // writes do not alias the stage reads of the MFMAs that follow and interleave the two
  __shared__ __attribute__((aligned(16))) unsigned char pipe_scr[PIPE ? 8 * EP_BYTES : 16];
  unsigned char* const escr = PIPE ? pipe_scr + (threadIdx.x >> 6) * EP_BYTES
                                   : const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(sbias)) +
                                         ((WS ? BN : p.tilesN * BN) * 4) + (threadIdx.x >> 6) * EP_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  // Staging loads are raw buffer loads: an out-of-range offset returns zero in hardware, so the zero padding
  // of the convolution and the ragged channel tail need no branches, and hipcc can keep two items' loads in
  // flight behind counted s_waitcnt vmcnt(N) (predicated global loads made it drain with vmcnt(0)).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);  // PIPE: branch-free masked stores
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res ? p.res : p.y), 0, p.res ? p.r_bytes : 0u, 0x00020000);
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(p.w);
  const int G = (int)gridDim.x;

  const int myTiles = (p.nTiles - (int)blockIdx.x + G - 1) / G;
  const int nItems = myTiles * p.nChunks;
  if (nItems <= 0) return;

  // tile id = ((n*tilesY + ty)*tilesX + tx)*tilesN + nt, this block visits blockIdx.x + k*G.
  // Decode once, then advance by G with carries (no per-tile integer divisions).
  struct TileIt {
    int n, ty, tx, nt;
  };
  auto decode = [&](int tile) {
    TileIt t;
    t.nt = tile % p.tilesN;
    int r = tile / p.tilesN;
    t.tx = r % p.tilesX;
    r /= p.tilesX;
    t.ty = r % p.tilesY;
    t.n = r / p.tilesY;
    return t;
  };
  const TileIt step = decode(G);  // G as a mixed-radix number
  auto advance = [&](TileIt& t) {
    t.nt += step.nt;
    int c = t.nt >= p.tilesN;
    t.nt -= c ? p.tilesN : 0;
    t.tx += step.tx + c;
    c = t.tx >= p.tilesX;
    t.tx -= c ? p.tilesX : 0;
    t.ty += step.ty + c;
    c = t.ty >= p.tilesY;
    t.ty -= c ? p.tilesY : 0;
    t.n += step.n + c;
  };

  // ---- loader state (runs one item ahead of the compute state) -----------------------------------
  TileIt lt = decode((int)blockIdx.x);
  int l_chunk = 0;
  unsigned aoff[NA];  // BYTE offset of this thread's halo slots for the loader's tile (~0u = outside the image)
  // two static register sets: the loads of items it+1 and it+2 are in flight while item it is multiplied
  u32x4 ra0[NA], rw0[NW], ra1[NA], rw1[NW];
  int l_item = 0;

  auto tile_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int s = tid + NT * i;
      const int pix = s >> 2;
      const int hy = pix / HWD, hx = pix - hy * HWD;
      const int gy = lt.ty * (TH * S) - 1 + hy, gx = lt.tx * (TW * S) - 1 + hx;
      const bool ok = (pix < NPIX) && ((unsigned)gy < (unsigned)p.H) && ((unsigned)gx < (unsigned)p.W);
      aoff[i] = ok ? ((unsigned)((lt.n * p.H + gy) * p.W + gx) * (unsigned)p.ldx + (unsigned)((s & 3) * EPC)) * (unsigned)sizeof(T) : ~0u;
    }
  };

  auto issue_loads = [&](u32x4 (&ra)[NA], u32x4 (&rw)[NW]) {  // loads of (lt, l_chunk), then advance the loader
    if (l_chunk == 0) tile_offsets();
    const int cbase = l_chunk * KCE;
    const bool chan_ok = cbase + (tid & 3) * EPC < p.Cin;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const unsigned voff = (aoff[i] != ~0u && chan_ok) ? aoff[i] + (unsigned)(cbase * (int)sizeof(T)) : 0xfffffff0u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, voff, 0, 0);
    }
    if constexpr (!WS) {
      const unsigned wbase = (unsigned)((lt.nt * p.nChunks + l_chunk) * W_CHUNKS) * 16u;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int s = tid + NT * i;
        rw[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (W_CHUNKS % NT == 0 || s < W_CHUNKS) ? wbase + (unsigned)s * 16u : 0xfffffff0u, 0, 0);
      }
    }
    // past the last item the loader stays put and harmlessly re-loads it: every issue is unconditional, so
    // the compiler can use counted s_waitcnt vmcnt(N) instead of draining the younger prefetch
    if (++l_item < nItems) {
      if (++l_chunk == p.nChunks) {
        l_chunk = 0;
        advance(lt);
      }
    } else {
      l_item = nItems - 1;
    }
  };

  auto store_lds = [&](int stage, const u32x4 (&ra)[NA], const u32x4 (&rw)[NW]) {
    unsigned char* sa = stage0 + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int s = tid + NT * i;
      const int pix = s >> 2, ch = s & 3;
      if (NPIX * 4 % NT == 0 || pix < NPIX) *reinterpret_cast<u32x4*>(sa + pix * PP + ch * 16) = ra[i];
    }
    if constexpr (!WS) {
      unsigned char* sw = sa + A_BYTES;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const int s = tid + NT * i;
        if (W_CHUNKS % NT == 0 || s < W_CHUNKS) *reinterpret_cast<u32x4*>(sw + s * 16) = rw[i];
      }
    }
  };

  f32x4 acc[MF][NF];
  // accumulators start from the bias (row constants as the initial C operand): saves the epilogue adds
  auto init_acc = [&](int nt) {
    const float* sb = sbias + (WS ? 0 : nt * BN) + lq * 4;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb + j * 16);
#pragma unroll
      for (int i = 0; i < MF; ++i) acc[i][j] = bb;
    }
  };

  // byte offset of this lane's fragment element for tap (0,0), row i = 0 inside a halo stage
  const int a_lane = ((wave * MF * S) * HWD + lr * S) * PP + lq * 16;
  auto compute_tap = [&](int stage, int chunk, int tap) {
    const unsigned char* sa = stage0 + stage * STAGE + a_lane;
    const unsigned char* sw = (WS ? dyn_smem + chunk * W_BYTES : stage0 + stage * STAGE + A_BYTES) + lane * 16;
    const int r = tap / 3, q = tap % 3;
    u32x4 a[MF], b[NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa + ((i * S + r) * HWD + q) * PP);
#pragma unroll
    for (int j = 0; j < NF; ++j) b[j] = *reinterpret_cast<const u32x4*>(sw + (tap * NF + j) * 1024);
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = Elem<T>::mma(b[j], a[i], acc[i][j]);  // D[cout][pixel]
  };
  auto compute = [&](int stage, int chunk) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) compute_tap(stage, chunk, tap);
  };

  // ---- epilogue from registers: lane holds couts (lq*4 .. +3) of pixel lr for every (i, j) ------------
  OutT* __restrict__ yg = reinterpret_cast<OutT*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  float st_sum[STATS ? NF : 1][4], st_sq[STATS ? NF : 1][4];
  if constexpr (STATS) {
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) st_sum[j][e] = 0.f, st_sq[j][e] = 0.f;
    if (blockIdx.x == 0)  // the totals bn_sum_partials_kernel adds the slots into
      for (int i = tid; i < 2 * p.Cout; i += NT) p.stats[i] = 0.0;
  }
  auto epilogue = [&](const TileIt& t, const f32x4 (&acc)[MF][NF], bool valid) {
    if constexpr (!PIPE) mfma_epilogue_fence<T>();
    const int xx = t.tx * TW + lr;
    const int co0 = t.nt * BN + lq * 4;
    size_t m[MF];
    bool rowok[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int yy = t.ty * TH + wave * MF + i;
      rowok[i] = valid && yy < p.Ho && xx < p.Wo;
      m[i] = (size_t)(t.n * p.Ho + (rowok[i] ? yy : 0)) * p.Wo + (rowok[i] ? xx : 0);
    }
    // all residual loads first (one wait for the lot), then the arithmetic
    float rv[MF][NF][4];
    if constexpr (!OUTF32 && !PIPE) {
      if (rg != nullptr) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int co = co0 + j * 16;
            const bool ok = rowok[i] && co < p.Cout;
            const T* rp = rg + m[i] * (size_t)p.ldres + (ok ? co : 0);
            if constexpr (sizeof(T) == 4) {
              const f32x4 tt = ok ? *reinterpret_cast<const f32x4*>(rp) : f32x4{0.f, 0.f, 0.f, 0.f};
              rv[i][j][0] = tt[0], rv[i][j][1] = tt[1], rv[i][j][2] = tt[2], rv[i][j][3] = tt[3];
            } else {
              typedef __attribute__((ext_vector_type(4))) T t4;
              const u32x2 raw = ok ? *reinterpret_cast<const u32x2*>(rp) : u32x2{0u, 0u};
              const t4 tt = __builtin_bit_cast(t4, raw);
#pragma unroll
              for (int e = 0; e < 4; ++e) rv[i][j][e] = Elem<T>::to_f32(tt[e]);
            }
          }
      }
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};  // bias is already inside
        if (PIPE || p.act == DY_ACT_SILU) {  // PIPE is only launched for SiLU layers without residual: no branches
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
        }
        if constexpr (!OUTF32 && !PIPE) {
          if (rg != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[i][j][e];
          }
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
          if constexpr (STATS) {
            if (rowok[i]) {  // (channels beyond Cout carry zero weights and a zero bias: they add nothing, and are not written out below)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float f = Elem<T>::to_f32(o[e]);  // what BatchNorm will read back
                st_sum[j][e] += f, st_sq[j][e] += f * f;
              }
            }
          }
        }
      }
    }
    // read back as rows: CPR 16-byte chunks per pixel row, consecutive lanes cover consecutive chunks
    constexpr int CPR = BN * (int)sizeof(OutT) / 16;
    constexpr int VE = 16 / (int)sizeof(OutT);
#pragma unroll
    for (int k = 0; k < MF * 16 * CPR / 64; ++k) {
      const int idx = k * 64 + lane;
      const int pixl = idx / CPR, cc = idx - pixl * CPR;
      const int i = pixl >> 4, px = pixl & 15;
      const int yy = t.ty * TH + wave * MF + i, xo = t.tx * TW + px;
      const int co = t.nt * BN + cc * VE;
      const u32x4 val = *reinterpret_cast<const u32x4*>(escr + pixl * EP_PITCH + cc * 16);
      if constexpr (PIPE) {  // out-of-range offset: the store is dropped by the hardware (cout % VE == 0 on this path)
        const bool ok = valid && yy < p.Ho && xo < p.Wo && co < p.Cout;
        const unsigned off = ok ? (unsigned)((((size_t)(t.n * p.Ho + yy) * p.Wo + xo) * (size_t)p.ldy + co) * sizeof(OutT)) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(val, yrs, off, 0, 0);
      } else if (valid && yy < p.Ho && xo < p.Wo && co < p.Cout) {
        OutT* yp = yg + ((size_t)(t.n * p.Ho + yy) * p.Wo + xo) * (size_t)p.ldy + co;
        if (co + VE <= p.Cout) {
          *reinterpret_cast<u32x4*>(yp) = val;
        } else {  // ragged channel tail (cout % 4 == 0 is guaranteed): 8-byte halves
          const u32x2 lo = u32x2{val[0], val[1]};
          *reinterpret_cast<u32x2*>(yp) = lo;
        }
      }
    }
  };

  // PIPE pieces: one (i, j) accumulator fragment -> SiLU -> scratch; and the row-wise read-back + masked buffer stores
  // residual of the PREVIOUS tile (Bottleneck shortcut): 8-byte raw buffer loads, zero when there is no residual (the
  // descriptor then has zero records) or the pixel is outside the image — no branches
  u32x2 rres[MF][NF];
  auto load_residual = [&](const TileIt& t, bool valid) {
    const int xx = t.tx * TW + lr;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int yy = t.ty * TH + wave * MF + i;
      const bool ok = valid && yy < p.Ho && xx < p.Wo;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int co = t.nt * BN + j * 16 + lq * 4;
        const unsigned off = (ok && co < p.Cout) ? (unsigned)((((size_t)(t.n * p.Ho + yy) * p.Wo + xx) * (size_t)p.ldres + co) * sizeof(T)) : 0xfffffff0u;
        rres[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rrs, off, 0, 0);
      }
    }
  };
  auto epi_pair = [&](const f32x4 (&accp)[MF][NF], int i, int j) {
    float v[4] = {accp[i][j][0], accp[i][j][1], accp[i][j][2], accp[i][j][3]};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
    if constexpr (sizeof(T) == 2) {  // PIPE exists for 16-bit storage only
      typedef __attribute__((ext_vector_type(4))) T t4;
      const t4 rr = __builtin_bit_cast(t4, rres[i][j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rr[e]);
      t4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
      *reinterpret_cast<u32x2*>(escr + (i * 16 + lr) * EP_PITCH + (j * 16 + lq * 4) * (int)sizeof(T)) = __builtin_bit_cast(u32x2, o);
    }
  };
  auto epi_store = [&](const TileIt& t, bool valid) {
    constexpr int CPR = BN * (int)sizeof(OutT) / 16;
    constexpr int VE = 16 / (int)sizeof(OutT);
#pragma unroll
    for (int k = 0; k < MF * 16 * CPR / 64; ++k) {
      const int idx = k * 64 + lane;
      const int pixl = idx / CPR, cc = idx - pixl * CPR;
      const int i = pixl >> 4, px = pixl & 15;
      const int yy = t.ty * TH + wave * MF + i, xo = t.tx * TW + px;
      const int co = t.nt * BN + cc * VE;
      const u32x4 val = *reinterpret_cast<const u32x4*>(escr + pixl * EP_PITCH + cc * 16);
      const bool ok = valid && yy < p.Ho && xo < p.Wo && co < p.Cout;
      const unsigned off = ok ? (unsigned)((((size_t)(t.n * p.Ho + yy) * p.Wo + xo) * (size_t)p.ldy + co) * sizeof(OutT)) : 0xfffffff0u;
      __builtin_amdgcn_raw_buffer_store_b128(val, yrs, off, 0, 0);
    }
  };

  // ---- prologue ------------------------------------------------------------------------------------------
  if constexpr (WS) {  // the block's n-tile is constant (host makes gridDim a multiple of tilesN)
    const u32x4* wsrc = wg + (size_t)lt.nt * p.nChunks * W_CHUNKS;
    const int total = p.nChunks * W_CHUNKS;
    for (int s = tid; s < total; s += NT) *reinterpret_cast<u32x4*>(dyn_smem + s * 16) = wsrc[s];
  }
  {
    float* sb = const_cast<float*>(sbias);
    // WS blocks stay on one n-tile and keep only its BN biases; streaming blocks keep all of them.
    // (tilesN * BN <= dy_conv_cout_pad(cout): the bias buffer is zero padded)
    const int nb = WS ? BN : p.tilesN * BN, b0 = WS ? lt.nt * BN : 0;
    for (int s = tid; s < nb; s += NT) sb[s] = p.bias[b0 + s];
  }
  TileIt ct = lt;  // compute-side tile
  issue_loads(ra0, rw0);  // item 0
  store_lds(0, ra0, rw0);
  issue_loads(ra1, rw1);  // item 1
  __syncthreads();
  init_acc(ct.nt);

  // ---- item pipeline: one barrier per (tile, chunk) item, prefetch depth two -------------------------
  if constexpr (PIPE) {
    f32x4 acc_prev[MF][NF];  // the previous tile's results wait here for their deferred epilogue
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc_prev[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    TileIt pt = ct;
    bool pvalid = false;
    // one (tile, chunk) item; FIRST / LAST of its tile are compile-time, items alternate stages and register sets
    auto item = [&](auto first_c, auto last_c, int stage, int chunk, u32x4 (&ra_i)[NA], u32x4 (&rw_i)[NW], const u32x4 (&ra_s)[NA],
                    const u32x4 (&rw_s)[NW]) {
      if constexpr (decltype(first_c)::value) load_residual(pt, pvalid);  // older than the halo prefetch: counted waits suffice
      issue_loads(ra_i, rw_i);
      if constexpr (decltype(first_c)::value) {
        // source-order interleave (the scheduler keeps it): after the MFMAs of tap k, the SiLU + scratch write of one
        // accumulator fragment of the PREVIOUS tile issue while the matrix pipe works; the row stores follow the last tap
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          compute_tap(stage, chunk, tap);
#pragma unroll
          for (int k = tap; k < MF * NF; k += 8)  // 8 slots (taps 0..7) share the MF*NF fragments
            if (tap < 8) epi_pair(acc_prev, k / NF, k % NF);
        }
        epi_store(pt, pvalid);
      } else {
        compute(stage, chunk);
      }
      store_lds(1 - stage, ra_s, rw_s);
      if constexpr (decltype(last_c)::value) {
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
          for (int j = 0; j < NF; ++j) acc_prev[i][j] = acc[i][j];
        pt = ct;
        pvalid = true;
        advance(ct);
        init_acc(ct.nt);
      }
      __syncthreads();
    };
    using Tr = std::true_type;
    using Fa = std::false_type;
    if constexpr (NCH == 1) {
      for (int it = 0; it < nItems; it += 2) {
        item(Tr{}, Tr{}, 0, 0, ra0, rw0, ra1, rw1);
        if (it + 1 < nItems) item(Tr{}, Tr{}, 1, 0, ra1, rw1, ra0, rw0);
      }
    } else if constexpr (NCH == 2) {
      for (int it = 0; it < nItems; it += 2) {
        item(Tr{}, Fa{}, 0, 0, ra0, rw0, ra1, rw1);
        item(Fa{}, Tr{}, 1, 1, ra1, rw1, ra0, rw0);
      }
    } else {
      static_assert(NCH == 4, "PIPE is built for 1, 2 or 4 chunks per tile");
      for (int it = 0; it < nItems; it += 4) {
        item(Tr{}, Fa{}, 0, 0, ra0, rw0, ra1, rw1);
        item(Fa{}, Fa{}, 1, 1, ra1, rw1, ra0, rw0);
        item(Fa{}, Fa{}, 0, 2, ra0, rw0, ra1, rw1);
        item(Fa{}, Tr{}, 1, 3, ra1, rw1, ra0, rw0);
      }
    }
    load_residual(pt, pvalid);
#pragma unroll
    for (int k = 0; k < MF * NF; ++k) epi_pair(acc_prev, k / NF, k % NF);
    epi_store(pt, pvalid);
    return;
  }
  int c_chunk = 0;
  auto tile_end = [&]() {
    if (++c_chunk == p.nChunks) {  // last chunk of a tile
      c_chunk = 0;
      if (!(p.dbg & 4)) epilogue(ct, acc, true);
      advance(ct);
      init_acc(ct.nt);
    }
  };
  for (int it = 0; it < nItems; it += 2) {
    // even item: stage 0 holds it, set 1 holds (in flight) it+1, set 0 is free for it+2
    if (!(p.dbg & 1)) issue_loads(ra0, rw0);
    if (!(p.dbg & 2)) compute(0, c_chunk);
    if (!(p.dbg & 8)) store_lds(1, ra1, rw1);  // waits for it+1 only: it+2 stays in flight
    tile_end();
    __syncthreads();
    if (it + 1 >= nItems) break;
    // odd item: stage 1 holds it+1, set 0 holds it+2, set 1 is free for it+3
    if (!(p.dbg & 1)) issue_loads(ra1, rw1);
    if (!(p.dbg & 2)) compute(1, c_chunk);
    if (!(p.dbg & 8)) store_lds(0, ra0, rw0);
    tile_end();
    __syncthreads();
  }
  if constexpr (STATS) {
    float* sred = reinterpret_cast<float*>(dyn_smem);  // [8 waves][2][BN]
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) st_sum[j][e] += __shfl_xor(st_sum[j][e], m, 64), st_sq[j][e] += __shfl_xor(st_sq[j][e], m, 64);
      }
    __syncthreads();  // the weights at the head of the LDS are not read any more
    if (lr == 0) {
#pragma unroll
      for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = j * 16 + lq * 4 + e;
          sred[(wave * 2 + 0) * BN + cc] = st_sum[j][e];
          sred[(wave * 2 + 1) * BN + cc] = st_sq[j][e];
        }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cc = tid - which * BN;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += sred[(k * 2 + which) * BN + cc];
      const int co = ((int)blockIdx.x % p.tilesN) * BN + cc;
      if (co < p.Cout) p.stats[(size_t)(1 + (int)blockIdx.x / p.tilesN) * 2 * p.Cout + which * p.Cout + co] = (double)t;
    }
  }
}

template <typename T, int S, int MF, int NF>
struct HaloGeom {
  static constexpr int TH = 8 * MF, TW = 16;
  static constexpr int NPIX = ((TH - 1) * S + 3) * ((TW - 1) * S + 3);
  static constexpr int A_BYTES = (NPIX * kHaloPixPitch + 15) / 16 * 16;
  static constexpr int W_BYTES = 9 * NF * 1024;
};

constexpr int kLdsBudget = 160 * 1024;

// LDS bytes of one workgroup; *ws tells whether the weight-stationary layout fits.
template <typename T, int S, int MF, int NF, bool OUTF32>
static int halo_smem(const Conv3Args& a, bool* ws) {
  typedef HaloGeom<T, S, MF, NF> Gm;
  const int tilesN = (a.Cout + NF * 16 - 1) / (NF * 16);
  const int ep_bytes = 8 * MF * 16 * (NF * 16 * (int)(OUTF32 ? 4 : sizeof(T)) + 16);
  const int smem_ws = a.nChunks * Gm::W_BYTES + 2 * Gm::A_BYTES + NF * 16 * 4 + ep_bytes;
  const int smem_st = 2 * (Gm::A_BYTES + Gm::W_BYTES) + tilesN * NF * 16 * 4 + ep_bytes;
  *ws = smem_ws <= kLdsBudget && !(a.dbg & 16);
  return *ws ? smem_ws : smem_st;
}

template <typename T, int S, int MF, int NF, bool OUTF32>
static int launch_halo(const Conv3Args& a, int batch, hipStream_t st) {
  typedef HaloGeom<T, S, MF, NF> Gm;
  Conv3Args p = a;
  p.tilesX = (p.Wo + Gm::TW - 1) / Gm::TW;
  p.tilesY = (p.Ho + Gm::TH - 1) / Gm::TH;
  p.tilesN = (p.Cout + NF * 16 - 1) / (NF * 16);
  p.nTiles = batch * p.tilesY * p.tilesX * p.tilesN;
  bool ws = false;
  const int smem = halo_smem<T, S, MF, NF, OUTF32>(p, &ws);
  DY_REQUIRE(smem <= kLdsBudget, DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: HALO3X3 stride-%d tile needs %d B of LDS (> %d); pack this layer with DY_WLAYOUT_ROWS", S, smem, kLdsBudget);
  const int per_cu = kLdsBudget / smem >= 2 ? 2 : 1;  // 512-thread blocks: at most 2 are useful per CU
  int grid = 256 * per_cu;
  if (grid > p.nTiles) grid = p.nTiles;
  if (ws && grid > p.tilesN) grid -= grid % p.tilesN;  // keep every block on one n-tile
  static const int nopipe = dy_ablate("DYOLO_NO_PIPE");
  if constexpr (sizeof(T) == 2 && !OUTF32) {
    if (ws && (p.nChunks == 1 || p.nChunks == 2 || p.nChunks == 4) && !nopipe && !p.dbg && p.act == DY_ACT_SILU && p.Cout % 8 == 0 &&
        p.y_bytes && (!p.res || p.r_bytes)) {
      constexpr int ep_bytes = 8 * MF * 16 * (NF * 16 * (int)sizeof(T) + 16);  // a static LDS object in this variant
#define DY_PIPE_LAUNCH(N)                                                                                                              \
  do {                                                                                                                                 \
    auto kern = conv3x3_halo_kernel<T, S, MF, NF, OUTF32, true, N>;                                                                    \
    static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget - ep_bytes); \
    (void)once;                                                                                                                        \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem - ep_bytes, st, p);                                                 \
  } while (0)
      if (p.nChunks == 1) DY_PIPE_LAUNCH(1);
      else if (p.nChunks == 2) DY_PIPE_LAUNCH(2);
      else DY_PIPE_LAUNCH(4);
#undef DY_PIPE_LAUNCH
      return check_launch("conv3x3_halo_kernel");
    }
  }
#ifndef DYOLO_L2E_BUILD  // (training convolutions carry no activation: they never come through the scaled-domain build)
  if constexpr (sizeof(T) == 2 && !OUTF32) {
    // every block on one n-tile (grid a multiple of tilesN), every block with at least one tile, one slot per block row
    if (p.stats && ws && !p.res && !p.dbg && grid % p.tilesN == 0 && grid <= p.nTiles && grid / p.tilesN <= kStatSlots) {
      auto kern = conv3x3_halo_kernel<T, S, MF, NF, OUTF32, true, 0, true>;
      static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
      (void)once;
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);
      note_stats(grid / p.tilesN);
      return check_launch("conv3x3_halo_kernel");
    }
  }
#endif
  if (ws) {
    auto kern = conv3x3_halo_kernel<T, S, MF, NF, OUTF32, true>;
    static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    (void)once;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);
  } else {
    auto kern = conv3x3_halo_kernel<T, S, MF, NF, OUTF32, false>;
    static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
    (void)once;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);
  }
  return check_launch("conv3x3_halo_kernel");
}

template <typename T, bool OUTF32>
static int launch_halo_dtype(const Conv3Args& a, int batch, int stride, hipStream_t st) {
  const bool nf4 = a.Cout > 32;
  // TH = 16 (MF 2) when the map is tall enough, there are plenty of tiles and the tile fits LDS; else TH = 8
  const long long tiles16 = (long long)batch * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16) * ((a.Cout + (nf4 ? 63 : 31)) / (nf4 ? 64 : 32));
  bool ws = false;
  bool big = stride == 1 && a.Ho >= 16 && tiles16 >= 256;
  if (big) big = (nf4 ? halo_smem<T, 1, 2, 4, OUTF32>(a, &ws) : halo_smem<T, 1, 2, 2, OUTF32>(a, &ws)) <= kLdsBudget;
  if (stride == 1) {
    if (nf4) return big ? launch_halo<T, 1, 2, 4, OUTF32>(a, batch, st) : launch_halo<T, 1, 1, 4, OUTF32>(a, batch, st);
    // <= 32 couts: 36 MFMAs per wave and item at MF 2 drown in the per-item overhead; 32-row tiles (MF 4) when the map is
    // tall enough, there are plenty of tiles and the bigger halo still fits LDS
    static const int no_mf4 = dy_ablate("DYOLO_NO_MF4");
    const long long tiles32 = (long long)batch * ((a.Ho + 31) / 32) * ((a.Wo + 15) / 16) * ((a.Cout + 31) / 32);
    bool ws4 = false;
    if (!no_mf4 && big && a.Ho >= 32 && tiles32 >= 512 && halo_smem<T, 1, 4, 2, OUTF32>(a, &ws4) <= kLdsBudget && ws4)
      return launch_halo<T, 1, 4, 2, OUTF32>(a, batch, st);
    return big ? launch_halo<T, 1, 2, 2, OUTF32>(a, batch, st) : launch_halo<T, 1, 1, 2, OUTF32>(a, batch, st);
  }
  if (nf4) return launch_halo<T, 2, 1, 4, OUTF32>(a, batch, st);
  return launch_halo<T, 2, 1, 2, OUTF32>(a, batch, st);
}

int conv3x3_hreg_try(const dy_conv_desc* d, hipStream_t st);  // conv3x3_hreg.hip: weights in registers, four workgroups per CU

// Entry used by dy_conv2d_nhwc when d->w_layout == DY_WLAYOUT_HALO3X3.
int conv3x3_halo_dispatch(const dy_conv_desc* d, hipStream_t st) {
  const int es = dtype_size_no_fp8(d->dtype);
  const int epc = 16 / es;
  DY_REQUIRE(d->ksize == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2) && d->groups <= 1 && !d->up2x && !d->x2,
             DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: HALO3X3 layout needs a dense 3x3 pad-1 stride-1/2 single-source conv");
  DY_REQUIRE(d->cin % epc == 0 && d->cout % 4 == 0, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: HALO3X3 needs cin %% %d == 0, cout %% 4 == 0", epc);
  DY_REQUIRE(aligned16(d->x) && (d->ld_x * es) % 16 == 0 && aligned16(d->w) && aligned16(d->bias) && aligned16(d->y), DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: views must be 16-byte aligned");
  DY_REQUIRE((d->ld_y * (d->out_f32 ? 4 : es)) % 16 == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: output pitch must be a multiple of 16 bytes");
  DY_REQUIRE(!d->residual || (aligned16(d->residual) && d->ld_res % 4 == 0), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: residual view misaligned");
  DY_REQUIRE((long long)d->batch * d->h * d->w_in * d->ld_x * es < (1ll << 32) - 64, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: input view exceeds 4 GiB (buffer descriptor range)");
  {
    const int rv = conv3x3_hreg_try(d, st);  // cin 32 / 64, cout % 64 == 0, 16-bit; stride 2 with cin 64 (r03)
    if (rv <= 0) return rv;
  }
  DY_REQUIRE(!(d->stride == 2 && d->cin == 64 && d->cout > 32), DY_ERR_UNSUPPORTED,
             "dy_conv2d_nhwc: a stride-2 64-channel layer in DY_WLAYOUT_HALO3X3 runs on conv3x3_hreg_s2 only (16-bit storage, no residual, views below 2 GiB)");
  Conv3Args a{};
  a.x = d->x;
  a.w = d->w;
  a.bias = d->bias;
  a.res = d->residual;
  a.y = d->y;
  a.H = d->h;
  a.W = d->w_in;
  a.Cin = d->cin;
  a.ldx = d->ld_x;
  a.Ho = d->ho;
  a.Wo = d->wo;
  a.Cout = d->cout;
  a.ldy = d->ld_y;
  a.ldres = d->ld_res;
  a.act = d->act;
  a.stats = (d->out_f32 || d->y_dtype1 || d->bnb_z) ? nullptr : d->bn_stats;  // (bnb_z: backward sums, conv3x3_hreg only)
  a.nChunks = (d->cin + 4 * epc - 1) / (4 * epc);
  a.x_bytes = (unsigned)((long long)d->batch * d->h * d->w_in * d->ld_x * es);
  {
    const long long yb = (long long)d->batch * d->ho * d->wo * d->ld_y * (d->out_f32 ? 4 : es);
    a.y_bytes = yb < (1ll << 32) - 64 ? (unsigned)yb : 0u;  // 0: view too large for a buffer descriptor -> PIPE variant not used
    const long long rb = d->residual ? (long long)d->batch * d->ho * d->wo * d->ld_res * es : 0;
    a.r_bytes = rb < (1ll << 32) - 64 ? (unsigned)rb : 0u;
  }
  {
    const int bn = d->cout > 32 ? 64 : 32;
    a.w_bytes = (unsigned)((long long)((d->cout + bn - 1) / bn) * a.nChunks * 9 * (bn / 16) * 1024);
  }
  {
    static const int dbg = dy_ablate("DYOLO_DBG");
    a.dbg = dbg;
  }
  switch (d->dtype) {
    case DY_BF16:
      return d->out_f32 ? launch_halo_dtype<bf16_t, true>(a, d->batch, d->stride, st) : launch_halo_dtype<bf16_t, false>(a, d->batch, d->stride, st);
    case DY_F16:
      return d->out_f32 ? launch_halo_dtype<f16_t, true>(a, d->batch, d->stride, st) : launch_halo_dtype<f16_t, false>(a, d->batch, d->stride, st);
    default:
      return launch_halo_dtype<float, false>(a, d->batch, d->stride, st);
  }
}

}  // namespace DY_NS
