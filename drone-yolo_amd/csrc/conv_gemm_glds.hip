// Implicit-GEMM convolution with LDS-DMA staging (global_load_lds_dwordx4) for the deep layers: 1x1 convolutions
// with many input channels, stride-2 3x3 downsamples and 3x3 layers on small maps, where the halo kernel's spatial
// tiles waste lanes (20x20 maps fill 39 % of a 16x16 tile) and the generic kernel's 32x64 wave tiles are LDS-read bound.
//
// GEMM view as in conv_igemm.hip: C[M][N] = A[M][K] * W[N][K]^T, M = batch*Ho*Wo pixels flattened (no spatial
// tile, so any map size fills the tile), N = Cout, K = (r, q, c) with c fastest; weights in DY_WLAYOUT_ROWS.
//
// Workgroup = 256 threads = 4 waves as 2 (M) x 2 (N); tile 128 pixels x BN couts (BN = 128 or 64); each wave owns
// 64 x BN/2 -> per 32-deep k-group 4 + BN/32 fragment reads feed 4 * BN/32 MFMAs (0.5 reads per MFMA at BN = 128).
// K-step = 128 bytes of K per row (64 bf16 / 32 fp32 channels of ONE tap: Cin % that == 0 is required).
//
// Staging: every wave-instruction of the LDS-DMA moves a PIECE = 8 rows x 128 B = 1 KiB: lane l fetches the 16-byte
// chunk (l & 7) ^ swz(row) of row 8*piece + (l >> 3), so eight consecutive lanes cover one full 128-byte line of
// global memory (coalesced) and the LDS image is plain row-major [rows][128 B] with the chunks of a row XOR-swizzled
// by swz(row) = (row >> 1) & 7 — the swizzle is applied on the SOURCE address because the LDS side of the DMA is
// lane-linear.  The MFMA operand read (lane (lr, lq): row lr, chunk 4*s + lq) then hits 16 distinct 16-byte units of
// the 256-byte bank row in every 16-lane group of ds_read_b128.
// Rows outside the image (zero padding) or beyond M fetch from a zero page instead.
// Two LDS stages; the DMA of step s+1 is issued right after the barrier that publishes step s and runs under its MFMAs.
// Epilogue: weights are the MFMA A operand, so a lane holds 4 consecutive couts of one pixel: bias (accumulator
// init), SiLU, residual, per-wave LDS transpose, 16-byte row stores.
#include "common_hip.h"
#include "conv_args.h"

namespace DY_NS {

__device__ __attribute__((aligned(256))) const unsigned int g_zero_page[64] = {0};

// BM x BN tile, WM x 2 waves (wave tile 64 x BN/2), STAGES LDS stages.  STAGES == 2: plain barrier per K-step, one step of
// DMA in flight (2 workgroups per CU cover for each other).  STAGES == 3: two steps in flight behind a counted
// s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads() would drain the DMA), one 8-wave workgroup per CU.
// STATS (r04; training forward, the convolution in front of a train-mode BatchNorm): per-channel sum / sum of squares of the STORED
// outputs of this tile go to slot tileM of the dy_bn_train_fwd workspace (ConvArgs.stats), so the BatchNorm needs no reduction pass
// over z (conv3x3_hreg.hip / conv1x1_stream.hip do the same): a lane keeps its four channels of each fragment over the tile's rows,
// the 16 pixel lanes of a quarter meet in a shuffle tree, the BM/64 wave rows in LDS.
template <typename T, int BM, int BN, int STAGES, int WN = 2, bool STATS = false>
__global__ __launch_bounds__(BM * WN) void conv_gemm_glds_kernel(const ConvArgs p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int NW = BM / 64 * WN;      // waves: BM/64 along M x WN along N (WN = 4: 64 x BN/4 wave tiles, twice the waves per SIMD)
  constexpr int BKE = 8 * EPC;          // K elements per step (128 bytes)
  constexpr int NFR = BN / WN / 16;     // cout fragments per wave (wave tile 64 x BN/WN)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW;       // A pieces per wave per step
  constexpr int PB = BN / 8 / NW;       // W pieces per wave per step
  static_assert(PA >= 1 && PB >= 1, "every wave stages at least one piece of each operand");
  constexpr int EG0 = 128 / (16 * (int)sizeof(T));  // cout fragments whose 16 couts fill 128 bytes of a pixel row
  constexpr int EG = NFR < EG0 ? NFR : EG0;          // fragments per epilogue group
  constexpr int CPP = EG * (int)sizeof(T);           // 16-byte chunks per pixel and group
  constexpr int EP_PITCH = 128 + 16;
  constexpr int NH = (64 * EP_PITCH * NW <= STAGES * STAGE) ? 1 : 2;  // epilogue passes per wave tile (64 / NH pixels each)
  constexpr int PXP = 64 / NH;
  static_assert(PXP * EP_PITCH * NW <= STAGES * STAGE, "epilogue scratch must fit the stage memory");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;  // BM/64 x WN wave grid
  const int lq = lane >> 4, lr = lane & 15;
  const unsigned L = xcd_remap(blockIdx.x, (unsigned)p.nblk);
  const int tileN = (int)(L % (unsigned)p.tilesN);
  const int tileM = (int)(L / (unsigned)p.tilesN);

  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ x2g = reinterpret_cast<const T*>(p.x2);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* zp = reinterpret_cast<const T*>(g_zero_page) + (lane & 7) * EPC;

  // ---- per-lane gather bookkeeping: the rows this lane fetches never change over the K loop ----
  const int prow = lane >> 3;  // row inside a piece
  // dilated-class mode (p.dil_cls, see conv_args.h): tileM = class * tilesPerClass + local tile; a row is pixel (n, yo, xo) of the
  // half-resolution source grid and stands for output pixel (2 yo + py, 2 xo + px)
  const int cls = p.dil_cls ? tileM / p.tilesPerClass : 0;
  const int cpy = cls >> 1, cpx = cls & 1;
  const int ltile = p.dil_cls ? tileM - cls * p.tilesPerClass : tileM;
  int a_n[PA], a_hi0[PA], a_wi0[PA], a_sw[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (wave * PA + i) * 8 + prow;
    const int m = ltile * BM + row;
    a_sw[i] = (((lane & 7) ^ ((row >> 1) & 7))) * EPC;
    if (p.dil_cls) {
      const bool ok = m < p.Mq;
      const unsigned mm = ok ? (unsigned)m : 0u;
      const unsigned n = fastdiv(mm, p.dHW);
      const unsigned rem = mm - n * (unsigned)(p.HB * p.WB);
      const unsigned yo = fastdiv(rem, p.dWB);
      a_n[i] = (int)n;
      a_hi0[i] = ok ? (int)yo : -(1 << 28);
      a_wi0[i] = (int)(rem - yo * (unsigned)p.WB);
    } else {
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int n = mm / p.HoWo;
      const int rem = mm - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_n[i] = n;
      a_hi0[i] = ok ? ho * p.stride - p.pad : -(1 << 28);
      a_wi0[i] = wo * p.stride - p.pad;
    }
  }
  const T* b_ptr[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int row = (wave * PB + j) * 8 + prow;
    b_ptr[j] = wg + (size_t)(tileN * BN + row) * (size_t)p.Kpad + (size_t)(((lane & 7) ^ ((row >> 1) & 7)) * EPC);
  }

  // r03: staging addresses without per-step vector arithmetic (fast_addr).  The first version rebuilt every piece's 64-bit source pointer
  // per K-step — (n HB + hb) WB + wb times ldx in 64 bits, selects against the zero page: ~95 vector instructions (8 of them 64-bit
  // multiply-adds) per 32 MFMAs and wave, which made these kernels issue bound.  Now a piece's byte offset for tap (0, 0) is a lane
  // constant (av1 / av2 for the two Concat sources, bv for the weights), the tap and channel step a SCALAR offset of the buffer
  // instruction, zero padding an out-of-range offset (the range check feeds zeros): per step two compares and a select per piece.
  constexpr unsigned kOob = 0xfffffff0u;
  const unsigned ES = (unsigned)sizeof(T);
  const unsigned pre1 = p.dil_cls ? 0u : (unsigned)((p.pad * p.WB + p.pad) * p.ldx) * ES, pre2 = (unsigned)((p.pad * p.W + p.pad) * p.ldx2) * ES;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - pre1, 0, p.xb + pre1, 0x00020000);
  const __amdgpu_buffer_rsrc_t x2rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x2)) - pre2, 0, p.x2b + pre2, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.wb, 0x00020000);
  unsigned av1[PA], av2[PA], bv[PB];
  if (p.fast_addr) {
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int hi0 = a_hi0[i] < 0 && a_hi0[i] < -p.pad ? 0 : a_hi0[i];  // (rows beyond M carry hi0 = -2^28: never valid, any offset will do)
      const int hb = (p.up2x == 1) ? (hi0 >> 1) : hi0, wb_ = (p.up2x == 1) ? (a_wi0[i] >> 1) : a_wi0[i];  // (dil_cls: hi0 / wi0 already are source coordinates)
      av1[i] = (unsigned)(((a_n[i] * p.HB + hb) * p.WB + wb_) * p.ldx + a_sw[i]) * ES + pre1;
      av2[i] = (unsigned)(((a_n[i] * p.H + hi0) * p.W + a_wi0[i]) * p.ldx2 + a_sw[i]) * ES + pre2;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (wave * PB + j) * 8 + prow;
      bv[j] = (unsigned)((tileN * BN + row) * p.Kpad + (((lane & 7) ^ ((row >> 1) & 7)) * EPC)) * ES;
    }
  }

  int kc = 0, kr = p.dil_cls ? (cpy ? 0 : 1) : 0, kq = p.dil_cls ? (cpx ? 0 : 1) : 0;  // (tap, channel) of the NEXT step to issue (wave-uniform)
  auto issue = [&](int step, int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const bool from_x = kc < p.split;
    if (p.dil_cls) {
      // tap (kr, kq) of this class reads source pixel (yo + dr, xo + dq), dr = (py - 1 + kr) / 2 in {0, 1}; the weights of the step sit at
      // K position (kr * 3 + kq) * Cin + kc whatever the step count is
      const int dr = (cpy - 1 + kr) >> 1, dq = (cpx - 1 + kq) >> 1;
      const unsigned soff = (unsigned)((dr * p.WB + dq) * p.ldx + kc) * ES;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = ((unsigned)(a_hi0[i] + dr) < (unsigned)p.HB) && ((unsigned)(a_wi0[i] + dq) < (unsigned)p.WB);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)(ok ? av1[i] : kOob), (int)soff, 0, 0);
      }
      const unsigned soffw = (unsigned)((kr * 3 + kq) * p.Cin + kc) * ES;
#pragma unroll
      for (int j = 0; j < PB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(sb + (wave * PB + j) * 1024), 16, (int)bv[j], (int)soffw, 0, 0);
      kc += BKE;
      if (kc >= p.Cin) {  // next valid tap of the class: kq (and kr) step by 2 from 1 - parity... i.e. {1} for an even row / column, {0, 2} for an odd one
        kc = 0;
        kq += 2;
        if (kq > 2) {
          kq = cpx ? 0 : 1;
          kr += 2;
        }
      }
      return;
    }
    if (p.fast_addr) {
      if (from_x) {
        const unsigned soff = (unsigned)((kr * p.WB + kq) * p.ldx + kc) * ES;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const bool ok = ((unsigned)(a_hi0[i] + kr) < (unsigned)p.H) && ((unsigned)(a_wi0[i] + kq) < (unsigned)p.W);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)(ok ? av1[i] : kOob), (int)soff, 0, 0);
        }
      } else {
        const unsigned soff = (unsigned)((kr * p.W + kq) * p.ldx2 + (kc - p.split)) * ES;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
          const bool ok = ((unsigned)(a_hi0[i] + kr) < (unsigned)p.H) && ((unsigned)(a_wi0[i] + kq) < (unsigned)p.W);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(x2rs, (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, (int)(ok ? av2[i] : kOob), (int)soff, 0, 0);
        }
      }
      const unsigned soffw = (unsigned)(step * BKE) * ES;
#pragma unroll
      for (int j = 0; j < PB; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(sb + (wave * PB + j) * 1024), 16, (int)bv[j], (int)soffw, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int hi = a_hi0[i] + kr, wi = a_wi0[i] + kq;
        const bool ok = ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W) && !(p.up2x == 2 && ((hi | wi) & 1));
        const T* ptr;  // computed for every lane (never dereferenced when !ok): keeps the gather branch-free
        if (from_x) {
          const int hb = p.up2x ? (hi >> 1) : hi, wb = p.up2x ? (wi >> 1) : wi;
          ptr = xg + (long long)((a_n[i] * p.HB + hb) * p.WB + wb) * (long long)p.ldx + (long long)(kc + a_sw[i]);
        } else {
          ptr = x2g + (long long)((a_n[i] * p.H + hi) * p.W + wi) * (long long)p.ldx2 + (long long)(kc - p.split + a_sw[i]);
        }
        const T* src = ok ? ptr : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < PB; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[j] + (size_t)step * BKE),
                                         (__attribute__((address_space(3))) void*)(sb + (wave * PB + j) * 1024), 16, 0, 0);
    }
    kc += BKE;
    if (kc >= p.Cin) {
      kc = 0;
      if (++kq == p.ks) {
        kq = 0;
        ++kr;
      }
    }
  };

  // ---- accumulators, initialised with the bias of this lane's couts ----
  f32x4 acc[NFR][4];
#pragma unroll
  for (int j = 0; j < NFR; ++j) {
    const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + tileN * BN + wn * (BN / WN) + j * 16 + lq * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = bb;
  }

  const int swz = lr >> 1;  // (row >> 1) & 7 of every fragment row this lane reads (fragment bases are multiples of 16)
  auto compute = [&](int stage) {
    const unsigned char* sa = smem + stage * STAGE + (wm * 64 + lr) * 128;
    const unsigned char* sb = smem + stage * STAGE + A_BYTES + (wn * (BN / WN) + lr) * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int slot = ((s * 4 + lq) ^ swz) * 16;
      u32x4 a[4], b[NFR];
#pragma unroll
      for (int j = 0; j < NFR; ++j) b[j] = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + slot);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + slot);
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = Elem<T>::mma(b[j], a[i], acc[j][i]);
    }
  };

  // ---- main loop: one barrier per K-step, the next step's DMA runs under this step's MFMAs ----
  const int nsteps = p.dil_cls ? (1 + cpy) * (1 + cpx) * (p.Cin / BKE) : p.Kpad / BKE;
  if constexpr (STAGES == 2) {
    issue(0, 0);
    for (int s = 0; s < nsteps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA of step s has landed (explicit: see the persistent kernel)
      __syncthreads();  // publishes stage s&1; everyone is done with stage (s+1)&1
      if (s + 1 < nsteps) issue(s + 1, (s + 1) & 1);
      compute(s & 1);
    }
  } else {
    issue(0, 0);
    if (nsteps > 1) issue(1, 1);
    int st = 0;  // stage of step s
    for (int s = 0; s < nsteps; ++s) {
      // step s must have landed; step s+1 (PA + PB younger DMAs of this wave) may stay in flight
      if (s + 1 < nsteps) {
        if constexpr (PA + PB == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (PA + PB == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if constexpr (PA + PB == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // every wave's step-s DMA landed, every wave finished the MFMAs of step s-1
      const int st2 = st == 0 ? 2 : st - 1;  // stage of step s+2 == stage of step s-1, free again
      if (s + 2 < nsteps) issue(s + 2, st2);
      compute(st);
      st = st == 2 ? 0 : st + 1;
    }
  }
  mfma_epilogue_fence<T>();
  __syncthreads();  // stage memory becomes the per-wave transpose scratch

  // ---- epilogue ----
  unsigned char* escr = smem + wave * (PXP * EP_PITCH);
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  const int m0 = ltile * BM + wm * 64;
  const int n0 = tileN * BN + wn * (BN / WN);
  // output pixel index of tile row m (dilated-class mode: row = (n, yo, xo) of the class -> pixel (2 yo + py, 2 xo + px)); -1 = beyond the end
  auto out_px = [&](int m) -> int {
    if (!p.dil_cls) return m < p.M ? m : -1;
    if (m >= p.Mq) return -1;
    const unsigned n = fastdiv((unsigned)m, p.dHW);
    const unsigned rem = (unsigned)m - n * (unsigned)(p.HB * p.WB);
    const unsigned yo = fastdiv(rem, p.dWB);
    const unsigned xo = rem - yo * (unsigned)p.WB;
    return (int)((n * (unsigned)p.H + 2u * yo + (unsigned)cpy) * (unsigned)p.W + 2u * xo + (unsigned)cpx);
  };
  float st_sum[STATS ? NFR : 1][4], st_sq[STATS ? NFR : 1][4];
  if constexpr (STATS) {
    static_assert(!STATS || sizeof(T) == 2, "STATS: 16-bit outputs");
#pragma unroll
    for (int j = 0; j < NFR; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) st_sum[j][e] = 0.f, st_sq[j][e] = 0.f;
    if (blockIdx.x == 0)  // the totals bn_sum_partials_kernel adds the slots into
      for (int i = tid; i < 2 * p.Cout; i += BM * WN) p.stats[i] = 0.0;
  }
#pragma unroll
  for (int g = 0; g < NFR / EG; ++g) {
#pragma unroll
    for (int half = 0; half < NH; ++half) {
#pragma unroll
      for (int jj = 0; jj < EG; ++jj) {
        const int j = g * EG + jj;
#pragma unroll
        for (int ii = 0; ii < 4 / NH; ++ii) {
          const int i = half * (4 / NH) + ii;
          float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
          if (p.act == DY_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
          }
          if (rg != nullptr) {
            const int m = out_px(m0 + i * 16 + lr);
            if (m >= 0 && n0 + j * 16 + lq * 4 < p.Cout) {  // (a tile may be wider than a narrow layer's Cout: weights / bias are zero-padded)
              typedef __attribute__((ext_vector_type(4))) T t4;
              const t4 rv = *reinterpret_cast<const t4*>(rg + (size_t)m * (size_t)p.ldres + (size_t)(n0 + j * 16 + lq * 4));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rv[e]);
            }
          }
          unsigned char* sp = escr + (ii * 16 + lr) * EP_PITCH + (jj * 16 + lq * 4) * (int)sizeof(T);
          if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>(sp) = f32x4{v[0], v[1], v[2], v[3]};
          } else {
            typedef __attribute__((ext_vector_type(4))) T t4;
            t4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
            *reinterpret_cast<u32x2*>(sp) = __builtin_bit_cast(u32x2, o);
            if constexpr (STATS) {
              if (m0 + i * 16 + lr < p.M) {
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
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < (PXP * CPP + 63) / 64; ++k) {  // PXP pixels x CPP chunks of 16 bytes
        const int idx = k * 64 + lane;
        const int px = idx / CPP, cc = idx % CPP;
        const int m = px < PXP ? out_px(m0 + half * PXP + px) : -1;
        if (m >= 0 && n0 + g * EG * 16 + cc * EPC < p.Cout) {
          const u32x4 val = *reinterpret_cast<const u32x4*>(escr + px * EP_PITCH + cc * 16);
          *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + (size_t)(n0 + g * EG * 16 + cc * EPC)) = val;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if constexpr (STATS) {
    constexpr int WM = BM / 64;
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
          const int cc = wn * (BN / WN) + j * 16 + lq * 4 + e;
          sred[(wm * 2 + 0) * BN + cc] = st_sum[j][e];
          sred[(wm * 2 + 1) * BN + cc] = st_sq[j][e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * BN; i += BM * WN) {
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

// ---- persistent variant (default): a workgroup walks tiles blockIdx.x, +gridDim.x, ... and requests the FIRST K-step of
// its next tile before it runs the epilogue of the current one, so the pipeline fill (one L2->LDS round trip, ~1 us: 12 % of
// a K = 512 tile) and the SiLU / transpose / store epilogue overlap.  Two LDS stages as above; the epilogue scratch is cut
// to 32 pixels per wave and pass (18 KiB) so that it fits the stage the prefetch does not use.
template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_gemm_glds_persist_kernel(const ConvArgs p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BM = 128, NW = 4;
  constexpr int BKE = 8 * EPC;
  constexpr int NFR = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;
  constexpr int EG0 = 128 / (16 * (int)sizeof(T));
  constexpr int EG = NFR < EG0 ? NFR : EG0;
  constexpr int CPP = EG * (int)sizeof(T);
  constexpr int EP_PITCH = 128 + 16;
  static_assert(32 * EP_PITCH * NW <= STAGE, "half-tile epilogue scratch must fit one stage");

  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lq = lane >> 4, lr = lane & 15;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ x2g = reinterpret_cast<const T*>(p.x2);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);
  const T* zp = reinterpret_cast<const T*>(g_zero_page) + (lane & 7) * EPC;
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  const int prow = lane >> 3;
  const int nsteps = p.Kpad / BKE;
  const int G = (int)gridDim.x;

  // gather state of the tile whose K-steps are being ISSUED
  int a_n[PA], a_hi0[PA], a_wi0[PA], a_sw[PA];
  const T* b_ptr[PB];
  int kc = 0, kr = 0, kq = 0;
  auto setup_tile = [&](int tile, int* tM, int* tN) {
    const unsigned L = xcd_remap((unsigned)tile, (unsigned)p.nblk);
    *tN = (int)(L % (unsigned)p.tilesN);
    *tM = (int)(L / (unsigned)p.tilesN);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int row = (wave * PA + i) * 8 + prow;
      const int m = *tM * BM + row;
      const bool ok = m < p.M;
      const int mm = ok ? m : 0;
      const int n = mm / p.HoWo;
      const int rem = mm - n * p.HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_n[i] = n;
      a_hi0[i] = ok ? ho * p.stride - p.pad : -(1 << 28);
      a_wi0[i] = wo * p.stride - p.pad;
      a_sw[i] = (((lane & 7) ^ ((row >> 1) & 7))) * EPC;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (wave * PB + j) * 8 + prow;
      b_ptr[j] = wg + (size_t)(*tN * BN + row) * (size_t)p.Kpad + (size_t)(((lane & 7) ^ ((row >> 1) & 7)) * EPC);
    }
    kc = 0, kr = 0, kq = 0;
  };
  auto issue = [&](int step, int stage) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + A_BYTES;
    const bool from_x = kc < p.split;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int hi = a_hi0[i] + kr, wi = a_wi0[i] + kq;
      const bool ok = ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W) && !(p.up2x == 2 && ((hi | wi) & 1));
      const T* ptr;
      if (from_x) {
        const int hb = p.up2x ? (hi >> 1) : hi, wb = p.up2x ? (wi >> 1) : wi;
        ptr = xg + (long long)((a_n[i] * p.HB + hb) * p.WB + wb) * (long long)p.ldx + (long long)(kc + a_sw[i]);
      } else {
        ptr = x2g + (long long)((a_n[i] * p.H + hi) * p.W + wi) * (long long)p.ldx2 + (long long)(kc - p.split + a_sw[i]);
      }
      const T* src = ok ? ptr : zp;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sa + (wave * PA + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[j] + (size_t)step * BKE),
                                       (__attribute__((address_space(3))) void*)(sb + (wave * PB + j) * 1024), 16, 0, 0);
    kc += BKE;
    if (kc >= p.Cin) {
      kc = 0;
      if (++kq == p.ks) {
        kq = 0;
        ++kr;
      }
    }
  };

  f32x4 acc[NFR][4];
  const int swz = lr >> 1;
  auto compute = [&](int stage) {
    const unsigned char* sa = smem + stage * STAGE + (wm * 64 + lr) * 128;
    const unsigned char* sb = smem + stage * STAGE + A_BYTES + (wn * (BN / 2) + lr) * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int slot = ((s * 4 + lq) ^ swz) * 16;
      u32x4 a[4], b[NFR];
#pragma unroll
      for (int j = 0; j < NFR; ++j) b[j] = *reinterpret_cast<const u32x4*>(sb + j * 16 * 128 + slot);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const u32x4*>(sa + i * 16 * 128 + slot);
#pragma unroll
      for (int j = 0; j < NFR; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = Elem<T>::mma(b[j], a[i], acc[j][i]);
    }
  };

  int tile = (int)blockIdx.x;
  if (tile >= p.nblk) return;
  int tM, tN;
  setup_tile(tile, &tM, &tN);
  int st0 = 0;  // stage that receives step 0 of the current tile
  issue(0, st0);
  while (true) {
    const int m0 = tM * BM + wm * 64, n0 = tN * BN + wn * (BN / 2);
#pragma unroll
    for (int j = 0; j < NFR; ++j) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + n0 + j * 16 + lq * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = bb;
    }
    for (int s = 0; s < nsteps; ++s) {
      // hipcc does not reliably order an LDS-DMA before a barrier (seen: no vmcnt wait in this loop): wait explicitly
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // step s landed in every wave; the other stage (and the epilogue scratch in it) is free
      if (s + 1 < nsteps) issue(s + 1, (st0 + s + 1) & 1);
      compute((st0 + s) & 1);
    }
    mfma_epilogue_fence<T>();
    const int last = (st0 + nsteps - 1) & 1;  // stage read by the final compute
    const int next = tile + G;
    const bool more = next < p.nblk;
    __syncthreads();  // every wave is done reading both stages
    if (more) {
      setup_tile(next, &tM, &tN);
      issue(0, last);  // prefetch: lands while the epilogue below runs out of the OTHER stage
    }
    // ---- epilogue of the finished tile: 32 pixels per pass through the per-wave scratch in stage 1 - last ----
    unsigned char* escr = smem + (1 - last) * STAGE + wave * (32 * EP_PITCH);
#pragma unroll
    for (int g = 0; g < NFR / EG; ++g) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int jj = 0; jj < EG; ++jj) {
          const int j = g * EG + jj;
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (p.act == DY_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
            }
            if (rg != nullptr) {
              const int m = m0 + i * 16 + lr;
              if (m < p.M) {
                typedef __attribute__((ext_vector_type(4))) T t4;
                const t4 rv = *reinterpret_cast<const t4*>(rg + (size_t)m * (size_t)p.ldres + (size_t)(n0 + j * 16 + lq * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rv[e]);
              }
            }
            unsigned char* sp = escr + (ii * 16 + lr) * EP_PITCH + (jj * 16 + lq * 4) * (int)sizeof(T);
            if constexpr (sizeof(T) == 4) {
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < (32 * CPP + 63) / 64; ++k) {  // 32 pixels x CPP chunks of 16 bytes
          const int idx = k * 64 + lane;
          const int px = idx / CPP, cc = idx % CPP;
          const int m = m0 + half * 32 + px;
          if (px < 32 && m < p.M) {
            const u32x4 val = *reinterpret_cast<const u32x4*>(escr + px * EP_PITCH + cc * 16);
            *reinterpret_cast<u32x4*>(yg + (size_t)m * (size_t)p.ldy + (size_t)(n0 + g * EG * 16 + cc * EPC)) = val;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    if (!more) break;
    tile = next;
    st0 = last;
  }
}

template <typename T, int BN>
static int launch_glds_persist(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
  const int tilesM = (p.M + 127) / 128;
  p.tilesN = p.Cout / BN;
  p.nblk = tilesM * p.tilesN;
  const int per_cu = (160 * 1024) / (2 * (128 + BN) * 128);  // 2 (BN = 128) or 3 (BN = 64) workgroups per CU
  int grid = 256 * per_cu;
  if (grid > p.nblk) grid = p.nblk;
  hipLaunchKernelGGL((conv_gemm_glds_persist_kernel<T, BN>), dim3((unsigned)grid), dim3(256), 0, st, p);
  return check_launch(BN == 128 ? "conv_gemm_glds_persist_kernel<128>" : "conv_gemm_glds_persist_kernel<64>");
}

template <typename T, int BM, int BN, int STAGES, int WN = 2>
static int launch_glds(const ConvArgs& a, hipStream_t st) {
  ConvArgs p = a;
#ifndef DYOLO_L2E_BUILD  // (training convolutions carry no activation: they never come through the scaled-domain build)
  if constexpr (sizeof(T) == 2 && BM * WN <= 1024 && !(BM == 256 && BN == 128)) {
    if (p.stats && !p.dil_cls && !p.res) {
      const int tilesM = (p.M + BM - 1) / BM;
      p.tilesN = (p.Cout + BN - 1) / BN;
      p.nblk = tilesM * p.tilesN;
      p.stats_atomic = tilesM > kStatSlots ? 1 : 0;  // (64 -> 128 1x1 stride 2 @160 at B = 64: 3,200 row blocks)
      if (p.stats_atomic) zero_async(p.stats, (size_t)(1 + kStatSlots) * 2 * p.Cout * sizeof(double), st);
      auto kern = conv_gemm_glds_kernel<T, BM, BN, STAGES, WN, true>;
      hipLaunchKernelGGL(kern, dim3((unsigned)p.nblk), dim3(BM * WN), 0, st, p);
      note_stats(p.stats_atomic ? kStatSlots : tilesM);
      return check_launch("conv_gemm_glds_kernel<stats>");
    }
  }
#endif
  int tilesM = (p.M + BM - 1) / BM;
  if (p.dil_cls) {  // one tile = one parity class of the output (conv_args.h)
    p.tilesPerClass = (p.Mq + BM - 1) / BM;
    tilesM = 4 * p.tilesPerClass;
  }
  p.tilesN = (p.Cout + BN - 1) / BN;  // (a narrow layer -- Cout < 64, whole chunks -- runs one masked 64-wide tile)
  p.nblk = tilesM * p.tilesN;
  auto kern = conv_gemm_glds_kernel<T, BM, BN, STAGES, WN>;
  hipLaunchKernelGGL(kern, dim3((unsigned)p.nblk), dim3(BM * WN), 0, st, p);  // static LDS only (up to 144 KiB)
  return check_launch(BM == 256 && BN == 256 ? "conv_gemm_glds_kernel<256,256>" : BM == 256 ? "conv_gemm_glds_kernel<256,128>"
                      : BN == 128 ? "conv_gemm_glds_kernel<128,128>" : "conv_gemm_glds_kernel<128,64>");
}

template <typename T>
static int launch_glds_dtype(const ConvArgs& a, hipStream_t st) {
  static const int big = dy_ablate("DYOLO_GLDS_BIG");  // 1: 256x128 three-stage, 2: never persistent, 3: always persistent
  // short K (<= 9 steps, the 64-channel stride-2 layers on 160x160 maps: thousands of tiles) measured faster one tile per
  // workgroup; everything else gains 3-14 % from the persistent walk with the next tile's first K-step prefetched
  const bool persist = big != 2 && !a.dil_cls && !a.stats && a.Cout % 64 == 0 && (big == 3 || a.Kpad > 9 * 8 * (16 / (int)sizeof(T)));  // (the parity-class tiling lives in the plain kernel)
  // 256 x 256 tiles (one workgroup per CU; sixteen waves of 64 x 64 = four per SIMD measured 5-14 % faster than eight of 64 x 128:
  // the kernels are wait-bound, not LDS-bound) halve the gathered-operand bytes per flop: 4-19 % faster on
  // the wide 1x1 layers and the 256-cout stride-2 layers at throughput batch sizes (512->256 @40x40: 241 -> 204 us); slower on
  // 256->512 stride 2 (measured 2x) and pointless when the tile count cannot fill the CUs.  Every output element is still
  // accumulated over K in the same order, so the choice does not change results.  DYOLO_GLDS_BIG=5 turns it off.
  if (big != 5 && big != 2 && big != 3 && a.Cout % 256 == 0 && (a.ks == 1 || (a.stride == 2 && a.Cout == 256)) &&
      (long long)a.M * a.Cout >= 256ll * 256 * 512)
    return big == 7 ? launch_glds<T, 256, 256, 2>(a, st) : launch_glds<T, 256, 256, 2, 4>(a, st);  // 16 waves of 64 x 64 (7: 8 waves of 64 x 128)
  if (a.Cout % 128 == 0) {
    if (big == 1) return launch_glds<T, 256, 128, 3>(a, st);
    // eight waves of 64 x 32 (two workgroups per CU = four waves per SIMD) measured 5-12 % faster than the persistent
    // four-wave 64 x 64 kernel with its cross-tile prefetch (256->512 stride 2 @40x40: 360 -> 315 us): wait-bound kernels
    // gain more from waves to switch to than from fewer LDS reads per MFMA.  DYOLO_GLDS_BIG=9: the four-wave kernels.
    if (big != 9 && big != 2 && big != 3) return launch_glds<T, 128, 128, 2, 4>(a, st);
    return persist ? launch_glds_persist<T, 128>(a, st) : launch_glds<T, 128, 128, 2>(a, st);
  }
  return persist ? launch_glds_persist<T, 64>(a, st) : launch_glds<T, 128, 64, 2>(a, st);
}

int conv_gemm_glds_try(const ConvArgs& a0, int dtype, bool out_f32, hipStream_t st) {
  static const int off = dy_ablate("DYOLO_NO_GLDS");
  ConvArgs a = a0;
  {  // buffer-addressed staging: every view below 4 GiB (32-bit byte offsets) and a gather that is linear in the tap
    const long long es_ = dtype == DY_F32 ? 4 : 2, n = a.HoWo > 0 ? a.M / a.HoWo : 0;
    const long long xb = n * a.HB * a.WB * a.ldx * es_, x2b = n * a.H * a.W * a.ldx2 * es_, wb = (long long)((a.Cout + 63) / 64 * 64) * a.Kpad * es_;
    const long long lim = (1ll << 32) - (1ll << 24);
    static const int slow = dy_ablate("DYOLO_GLDS_SLOW_ADDR");
    const bool small = !slow && xb < lim && x2b < lim && wb < lim;
    static const int nodil = dy_ablate("DYOLO_GLDS_NO_DIL");
    a.dil_cls = (small && !nodil && a.up2x == 2 && a.ks == 3 && a.pad == 1 && a.stride == 1 && a.H == 2 * a.HB && a.W == 2 * a.WB && a.split >= a.Cin) ? 1 : 0;
    a.fast_addr = (small && (a.up2x == 0 || (a.up2x == 1 && a.ks == 1) || a.dil_cls)) ? 1 : 0;
    a.xb = (unsigned)(xb < lim ? xb : 0), a.x2b = (unsigned)(x2b < lim ? x2b : 0), a.wb = (unsigned)(wb < lim ? wb : 0);
    if (a.dil_cls) {
      a.Mq = (int)(n * a.HB * a.WB);
      a.dWB = make_fastdiv((unsigned)a.WB);
      a.dHW = make_fastdiv((unsigned)(a.HB * a.WB));
    }
  }
  if (dtype == DY_FP8) return 1;  // not built for fp8: the generic kernel runs
  const int es = dtype_size_no_fp8(dtype);
  const int bke = 8 * (16 / es);
  if (off || out_f32 || !a.vec_store) return 1;
  // Cout: multiples of 64, or -- for the parity-class tiling of a stride-2 input gradient only (layer 1's 64 -> 32 at 320 x 320: the
  // zero-dilated gather on the generic kernel took 0.7 ms of a training step) -- a narrower layer on one masked 64-wide tile
  const bool narrow = a.dil_cls && a.Cout < 64 && a.Cout % (16 / es) == 0;
  if (a.Cin % bke || a.split % bke || (a.Cout % 64 && !narrow) || a.Kpad != a.ks * a.ks * a.Cin) return 1;
  // no threshold on M: which kernel runs must not depend on the batch size, so that an image's result is bit-identical
  // whatever batch it arrives in (tests/test_model_gpu.py::test_full_size_properties)
  if (a.res && a.ldres % 4) return 1;
  switch (dtype) {
    case DY_BF16: return launch_glds_dtype<bf16_t>(a, st);
    case DY_F16: return launch_glds_dtype<f16_t>(a, st);
    default: return launch_glds_dtype<float>(a, st);
  }
}

}  // namespace DY_NS
