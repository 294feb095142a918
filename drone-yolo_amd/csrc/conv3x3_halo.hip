// 3x3 (pad 1, stride 1 or 2) NHWC convolution on CDNA4 MFMA: persistent workgroups, LDS halo tiles.
//
// Why a second kernel: the generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every input
// pixel once per tap and spends ~8 VALU instructions of address arithmetic per MFMA.  Here a
// workgroup owns a TH x 16 patch of output pixels x BN output channels and walks the input
// channels in chunks of one MFMA k-group (32 bf16/f16 or 16 f32 channels = 64 bytes per pixel):
//
//   LDS stage = [ halo patch: ((TH-1)*S+3) x ((16-1)*S+3) pixels x 64 B ] + [ weights: 9 taps x BN x 64 B ]
//
//   - the halo patch is fetched ONCE per chunk (1.27x the output pixels for stride 1 instead of 9x);
//     all nine taps read it at compile-time-constant pixel offsets;
//   - weights are pre-packed by the host in MFMA-fragment order per (n-tile, chunk, tap), so staging
//     them is a linear 16-byte-per-lane copy and every B fragment read is `stage + constant + lane*16`;
//   - workgroups are persistent (grid = #CUs): a block walks its list of (tile, chunk) items with the
//     loads of item i+1 in flight during the 144 MFMAs (per wave) of item i, one barrier per item,
//     and no pipeline drain between tiles;
//   - the MFMA is issued with the WEIGHT fragment as the A operand, so a lane ends up holding 4
//     consecutive output channels of one pixel: the epilogue (bias, SiLU, residual) works on
//     registers and stores 8 (bf16/f16) or 16 (f32) contiguous bytes per lane, no LDS round trip.
//
// Halo image swizzle: pixel p, 16-byte chunk c lives at p*64 + ((c ^ ((p>>2)&3)) << 4), which makes
// the 16 consecutive pixels of a fragment row hit 16 distinct 16-byte slots of the 256-byte bank row,
// and the staging writes of 8 consecutive lanes (2 pixels x 4 chunks) conflict free.
//
// Reference semantics: Conv / RepVGGBlock (folded) / Bottleneck residual, as conv_igemm.hip.
#include "common.cuh"
#include <type_traits>

namespace dy {

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
};

template <typename T, int S, int MF, int NF, bool OUTF32>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const Conv3Args p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int KCE = 4 * EPC;             // channels per chunk (one MFMA k-group)
  constexpr int TH = 4 * MF, TW = 16;      // output patch of the workgroup; wave w owns rows [w*MF, (w+1)*MF)
  constexpr int HH = (TH - 1) * S + 3, HWD = (TW - 1) * S + 3;
  constexpr int NPIX = HH * HWD;
  constexpr int A_BYTES = NPIX * 64;
  constexpr int W_CHUNKS = 9 * NF * 64;    // 16-byte chunks of one weight stage
  constexpr int W_BYTES = W_CHUNKS * 16;
  constexpr int STAGE = A_BYTES + W_BYTES;
  constexpr int NA = (NPIX * 4 + 255) / 256;
  constexpr int NW = (W_CHUNKS + 255) / 256;
  constexpr int BN = NF * 16;
  typedef typename std::conditional<OUTF32, float, T>::type OutT;

  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane >> 4, lr = lane & 15;
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const u32x4* __restrict__ wg = reinterpret_cast<const u32x4*>(p.w);

  const int nItems = ((p.nTiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * p.nChunks;
  if (nItems <= 0) return;

  // ---- loader state (runs one item ahead of the compute state) -----------------------------------
  unsigned aoff[NA];       // element offset of this thread's halo slots for the loader's tile (~0u = outside)
  int l_tile = -1, l_nt = 0;
  u32x4 ra[NA], rw[NW];

  auto decode_tile = [&](int tile, int& n, int& y0, int& x0, int& nt) {
    nt = tile % p.tilesN;
    int t = tile / p.tilesN;
    const int tx = t % p.tilesX;
    t /= p.tilesX;
    const int ty = t % p.tilesY;
    n = t / p.tilesY;
    y0 = ty * TH;
    x0 = tx * TW;
  };

  auto issue_loads = [&](int item) {
    const int tile = (int)blockIdx.x + (item / p.nChunks) * (int)gridDim.x;
    const int c = item % p.nChunks;
    if (tile != l_tile) {
      l_tile = tile;
      int n, y0, x0;
      decode_tile(tile, n, y0, x0, l_nt);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int s = tid + 256 * i;
        const int pix = s >> 2;
        const int hy = pix / HWD, hx = pix - hy * HWD;
        const int gy = y0 * S - 1 + hy, gx = x0 * S - 1 + hx;
        const bool ok = (pix < NPIX) && ((unsigned)gy < (unsigned)p.H) && ((unsigned)gx < (unsigned)p.W);
        aoff[i] = ok ? (unsigned)((n * p.H + gy) * p.W + gx) * (unsigned)p.ldx + (unsigned)((s & 3) * EPC) : ~0u;
      }
    }
    const int cbase = c * KCE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const bool ok = (aoff[i] != ~0u) && (cbase + (tid & 3) * EPC < p.Cin);
      ra[i] = ok ? *reinterpret_cast<const u32x4*>(xg + (size_t)aoff[i] + cbase) : zero_chunk();
    }
    const u32x4* wsrc = wg + (size_t)(l_nt * p.nChunks + c) * W_CHUNKS;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int s = tid + 256 * i;
      if (W_CHUNKS % 256 == 0 || s < W_CHUNKS) rw[i] = wsrc[s];
    }
  };

  auto store_lds = [&](int stage) {
    unsigned char* sa = dyn_smem + stage * STAGE;
    unsigned char* sw = sa + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int s = tid + 256 * i;
      const int pix = s >> 2, ch = s & 3;
      if (NPIX * 4 % 256 == 0 || pix < NPIX)
        *reinterpret_cast<u32x4*>(sa + pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int s = tid + 256 * i;
      if (W_CHUNKS % 256 == 0 || s < W_CHUNKS) *reinterpret_cast<u32x4*>(sw + s * 16) = rw[i];
    }
  };

  f32x4 acc[MF][NF];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  const int pix_lane = (wave * MF * S) * HWD + lr * S;  // halo pixel of (row wave*MF, col lr), tap (0,0)
  auto compute = [&](int stage) {
    const unsigned char* sa = dyn_smem + stage * STAGE;
    const unsigned char* sw = sa + A_BYTES + lane * 16;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, q = tap % 3;
      u32x4 a[MF], b[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int pix = pix_lane + (i * S + r) * HWD + q;
        a[i] = *reinterpret_cast<const u32x4*>(sa + pix * 64 + ((lq ^ ((pix >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NF; ++j) b[j] = *reinterpret_cast<const u32x4*>(sw + (tap * NF + j) * 1024);
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = Elem<T>::mma(b[j], a[i], acc[i][j]);  // D[cout][pixel]
    }
  };

  // ---- epilogue from registers: lane holds couts (lq*4 .. +3) of pixel lr for every (i, j) ------------
  OutT* __restrict__ yg = reinterpret_cast<OutT*>(p.y);
  const T* __restrict__ rg = reinterpret_cast<const T*>(p.res);
  auto epilogue = [&](int tile) {
    int n, y0, x0, nt;
    decode_tile(tile, n, y0, x0, nt);
    const int xx = x0 + lr;
#pragma unroll
    for (int i = 0; i < MF; ++i) {
      const int yy = y0 + wave * MF + i;
      if (yy >= p.Ho || xx >= p.Wo) continue;
      const size_t m = (size_t)(n * p.Ho + yy) * p.Wo + xx;
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int co = nt * BN + j * 16 + lq * 4;
        if (co >= p.Cout) continue;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(p.bias + co);
        float v[4] = {acc[i][j][0] + bb[0], acc[i][j][1] + bb[1], acc[i][j][2] + bb[2], acc[i][j][3] + bb[3]};
        if (p.act == DY_ACT_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(v[e]);
        }
        if constexpr (!OUTF32) {
          if (rg != nullptr) {
            const T* rp = rg + m * (size_t)p.ldres + co;
            if constexpr (sizeof(T) == 4) {
              const f32x4 t = *reinterpret_cast<const f32x4*>(rp);
              v[0] += t[0], v[1] += t[1], v[2] += t[2], v[3] += t[3];
            } else {
              typedef __attribute__((ext_vector_type(4))) T t4;
              const t4 t = __builtin_bit_cast(t4, *reinterpret_cast<const u32x2*>(rp));
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(t[e]);
            }
          }
        }
        OutT* yp = yg + m * (size_t)p.ldy + co;
        if constexpr (OUTF32 || sizeof(T) == 4) {
          *reinterpret_cast<f32x4*>(yp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          typedef __attribute__((ext_vector_type(4))) T t4;
          t4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
          *reinterpret_cast<u32x2*>(yp) = __builtin_bit_cast(u32x2, o);
        }
      }
    }
  };

  // ---- item pipeline: one barrier per (tile, chunk) item -----------------------------------------------
  issue_loads(0);
  store_lds(0);
  zero_acc();
  __syncthreads();
  for (int it = 0; it < nItems; ++it) {
    const bool more = (it + 1) < nItems;
    if (more) issue_loads(it + 1);
    compute(it & 1);
    if ((it + 1) % p.nChunks == 0) {  // last chunk of a tile
      epilogue((int)blockIdx.x + (it / p.nChunks) * (int)gridDim.x);
      zero_acc();
    }
    if (more) store_lds((it + 1) & 1);
    __syncthreads();
  }
}

template <typename T, int S, int MF, int NF, bool OUTF32>
static int launch_halo(const Conv3Args& a, int batch, hipStream_t st) {
  constexpr int TH = 4 * MF, TW = 16;
  constexpr int HH = (TH - 1) * S + 3, HWD = (TW - 1) * S + 3;
  constexpr int STAGE = HH * HWD * 64 + 9 * NF * 1024;
  Conv3Args p = a;
  p.tilesX = (p.Wo + TW - 1) / TW;
  p.tilesY = (p.Ho + TH - 1) / TH;
  p.tilesN = (p.Cout + NF * 16 - 1) / (NF * 16);
  p.nTiles = batch * p.tilesY * p.tilesX * p.tilesN;
  const int smem = 2 * STAGE;
  const int per_cu = (160 * 1024) / smem;  // LDS-limited residency
  int grid = 256 * (per_cu < 1 ? 1 : per_cu);
  if (grid > p.nTiles) grid = p.nTiles;
  auto kern = conv3x3_halo_kernel<T, S, MF, NF, OUTF32>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, p);
  return check_launch("conv3x3_halo_kernel");
}

template <typename T, bool OUTF32>
static int launch_halo_dtype(const Conv3Args& a, int batch, int stride, hipStream_t st) {
  const bool nf4 = a.Cout > 32;
  // TH = 16 (MF 4) when the map is tall enough and there are plenty of tiles; else TH = 8
  const long long tiles16 = (long long)batch * ((a.Ho + 15) / 16) * ((a.Wo + 15) / 16) * ((a.Cout + (nf4 ? 63 : 31)) / (nf4 ? 64 : 32));
  const bool big = stride == 1 && a.Ho >= 16 && tiles16 >= 256;
  if (stride == 1) {
    if (nf4) return big ? launch_halo<T, 1, 4, 4, OUTF32>(a, batch, st) : launch_halo<T, 1, 2, 4, OUTF32>(a, batch, st);
    return big ? launch_halo<T, 1, 4, 2, OUTF32>(a, batch, st) : launch_halo<T, 1, 2, 2, OUTF32>(a, batch, st);
  }
  if (nf4) return launch_halo<T, 2, 2, 4, OUTF32>(a, batch, st);
  return launch_halo<T, 2, 2, 2, OUTF32>(a, batch, st);
}

// Entry used by dy_conv2d_nhwc when d->w_layout == DY_WLAYOUT_HALO3X3.
int conv3x3_halo_dispatch(const dy_conv_desc* d, hipStream_t st) {
  const int es = dy_dtype_size(d->dtype);
  const int epc = 16 / es;
  DY_REQUIRE(d->ksize == 3 && d->pad == 1 && (d->stride == 1 || d->stride == 2) && d->groups <= 1 && !d->up2x && !d->x2,
             DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: HALO3X3 layout needs a dense 3x3 pad-1 stride-1/2 single-source conv");
  DY_REQUIRE(d->cin % epc == 0 && d->cout % 4 == 0, DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: HALO3X3 needs cin %% %d == 0, cout %% 4 == 0", epc);
  DY_REQUIRE(aligned16(d->x) && (d->ld_x * es) % 16 == 0 && aligned16(d->w) && aligned16(d->bias) && aligned16(d->y), DY_ERR_INVALID_ARG,
             "dy_conv2d_nhwc: views must be 16-byte aligned");
  const int oes = d->out_f32 ? 4 : es;
  DY_REQUIRE((d->ld_y * oes) % (4 * oes) == 0, DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: ld_y must be a multiple of 4 elements");
  DY_REQUIRE(!d->residual || (aligned16(d->residual) && d->ld_res % 4 == 0), DY_ERR_INVALID_ARG, "dy_conv2d_nhwc: residual view misaligned");
  DY_REQUIRE((long long)d->batch * d->h * d->w_in * d->ld_x < (1ll << 32), DY_ERR_UNSUPPORTED, "dy_conv2d_nhwc: input view exceeds 2^32 elements");
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
  a.nChunks = (d->cin + 4 * epc - 1) / (4 * epc);
  switch (d->dtype) {
    case DY_BF16:
      return d->out_f32 ? launch_halo_dtype<bf16_t, true>(a, d->batch, d->stride, st) : launch_halo_dtype<bf16_t, false>(a, d->batch, d->stride, st);
    case DY_F16:
      return d->out_f32 ? launch_halo_dtype<f16_t, true>(a, d->batch, d->stride, st) : launch_halo_dtype<f16_t, false>(a, d->batch, d->stride, st);
    default:
      return launch_halo_dtype<float, false>(a, d->batch, d->stride, st);
  }
}

}  // namespace dy
