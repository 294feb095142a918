// Whole C2f block (n = 1, hidden width 32) on the stride-4 maps as ONE kernel: cv1 (1x1 Cin->64) -> chunk -> Bottleneck
// (3x3 32->32, 3x3 32->32, optional shortcut) -> cat -> cv2 (1x1 96->64), every convolution with its folded BatchNorm bias
// and SiLU.  Reference: C2f.forward / Bottleneck.forward (nn/modules/block.py:227-249, 337-350) with Conv.forward_fuse
// (nn/modules/conv.py:53-55); yolov8-p2-repvgg.yaml layers 2 and 21 at scale s.
//
// Why: on the 160x160 maps these four convolutions are bandwidth bound one by one (64 + 64, 32 + 32, 32(+32) + 32, 96 + 64
// channels in and out per pixel = 1.3 KB of HBM traffic per pixel against 256 B when only the block's input and output
// move), and they were 17 % of the whole pass.  Here a 512-thread workgroup owns a 16 x 16 output tile: it stages the
// 20 x 20 input halo once, keeps every intermediate (y0|y1 on 20 x 20, t on 18 x 18, y2 on 16 x 16) in LDS as bf16/f16 —
// rounded exactly where the layer-by-layer path rounds — recomputes the halo rings (+17 % MFMA work) and writes only the
// block's output.  All four weight sets (56 KB, MFMA fragment order) stay in LDS for the workgroup's lifetime.
//
// LDS map (159.2 KB): W1 8 K | WM1 18 K | WM2 18 K | W2 12 K | biases | Y 400 px x 128 B | X 400 px x 128 B.  X is dead after
// cv1; its space then holds t (324 px x 80 B), y2 (256 px x 80 B) and the store scratch.  128-byte pixel rows of X / Y use the
// chunk ^ ((px >> 1) & 7) swizzle (conflict-free ds_read_b128 fragments); t / y2 use the 80-byte pitch of the halo kernel.
// The next tile's input is fetched into registers during phases 3-5 and written to LDS after the tile's last barrier.
#include "common_hip.h"

namespace dy {

struct C2fArgs {
  const void* x;
  void* y;
  const void* w1;   // FRAG1X1 order, cout 64, cin 64
  const void* wm1;  // HALO3X3 order, cout 32, cin 32
  const void* wm2;
  const void* w2;   // FRAG1X1 order, cout 64, cin 96
  const float* bias;  // b1[64] | bm1[32] | bm2[32] | b2[64]
  int N, H, W, ldx, ldy, tilesX, tilesY, nTiles, shortcut;
  unsigned x_bytes, y_bytes;
};

constexpr int kC2fW1 = 0, kC2fWM1 = 8192, kC2fWM2 = 26624, kC2fW2 = 45056, kC2fBias = 57344, kC2fY = 58368, kC2fX = 109568;
constexpr int kC2fSmem = kC2fX + 400 * 128;  // 160768
constexpr int kC2fT = kC2fX, kC2fY2 = kC2fX + 324 * 80;

template <typename T>
__global__ __launch_bounds__(512) void c2f_fused_kernel(const C2fArgs p) {
  constexpr int E = Elem<T>::EPC;  // 8
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lq = lane >> 4;
  const float* sbias = reinterpret_cast<const float*>(smem + kC2fBias);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);

  // ---- weights + biases -> LDS, once ----
  {
    const u32x4* s1 = reinterpret_cast<const u32x4*>(p.w1);
    const u32x4* s2 = reinterpret_cast<const u32x4*>(p.wm1);
    const u32x4* s3 = reinterpret_cast<const u32x4*>(p.wm2);
    const u32x4* s4 = reinterpret_cast<const u32x4*>(p.w2);
    for (int i = tid; i < 512; i += 512) reinterpret_cast<u32x4*>(smem + kC2fW1)[i] = s1[i];
    for (int i = tid; i < 1152; i += 512) reinterpret_cast<u32x4*>(smem + kC2fWM1)[i] = s2[i];
    for (int i = tid; i < 1152; i += 512) reinterpret_cast<u32x4*>(smem + kC2fWM2)[i] = s3[i];
    for (int i = tid; i < 768; i += 512) reinterpret_cast<u32x4*>(smem + kC2fW2)[i] = s4[i];
    if (tid < 192) reinterpret_cast<float*>(smem + kC2fBias)[tid] = p.bias[tid];
  }

  constexpr int NA = 7;  // 400 px x 8 chunks / 512 threads
  u32x4 ra[NA];
  auto tile_coords = [&](int tile, int* n, int* ty0, int* tx0) {
    const int tx = tile % p.tilesX;
    const int r = tile / p.tilesX;
    *tx0 = tx * 16;
    *ty0 = (r % p.tilesY) * 16;
    *n = r / p.tilesY;
  };
  auto issue_x = [&](int tile) {  // 20 x 20 halo of the tile, zero outside the image (raw buffer loads: OOB offset -> 0)
    int n, ty0, tx0;
    tile_coords(tile, &n, &ty0, &tx0);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int s = tid + 512 * i;
      const int px = s >> 3, ch = s & 7;
      const int hy = px / 20, hx = px - hy * 20;
      const int gy = ty0 - 2 + hy, gx = tx0 - 2 + hx;
      const bool ok = (px < 400) && ((unsigned)gy < (unsigned)p.H) && ((unsigned)gx < (unsigned)p.W) && tile < p.nTiles;
      const unsigned off = ok ? (unsigned)((((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldx + ch * E) * sizeof(T)) : 0xfffffff0u;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
    }
  };
  auto store_x = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int s = tid + 512 * i;
      const int px = s >> 3, ch = s & 7;
      if (px < 400) *reinterpret_cast<u32x4*>(smem + kC2fX + px * 128 + ((ch ^ ((px >> 1) & 7)) * 16)) = ra[i];
    }
  };
  typedef __attribute__((ext_vector_type(4))) T t4;
  auto pack4 = [&](const float (&v)[4]) -> u32x2 {
    t4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Elem<T>::from_f32(v[e]);
    return __builtin_bit_cast(u32x2, o);
  };

  const int G = (int)gridDim.x;
  int tile = (int)blockIdx.x;
  issue_x(tile);
  for (; tile < p.nTiles; tile += G) {
    int n, ty0, tx0;
    tile_coords(tile, &n, &ty0, &tx0);
    store_x();
    __syncthreads();  // B0: X (and, the first time, the weights) visible

    // ---- phase 2: cv1 1x1 64 -> 64 on the 20 x 20 region -> Y (y0 = ch 0..31, y1 = ch 32..63), zero outside the image ----
    {
      f32x4 acc[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + j * 16 + lq * 4);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        u32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const u32x4*>(smem + kC2fW1 + ((c * 4 + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = wave + 8 * i;
          if (f < 25) {
            const int px = 16 * f + lr;
            const u32x4 a = *reinterpret_cast<const u32x4*>(smem + kC2fX + px * 128 + (((c * 4 + lq) ^ ((px >> 1) & 7)) * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = Elem<T>::mma(b[j], a, acc[i][j]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = wave + 8 * i;
        if (f < 25) {
          const int px = 16 * f + lr;
          const int hy = px / 20, hx = px - hy * 20;
          const bool inside = ((unsigned)(ty0 - 2 + hy) < (unsigned)p.H) && ((unsigned)(tx0 - 2 + hx) < (unsigned)p.W);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = inside ? silu_f32(acc[i][j][e]) : 0.f;
            const int chunk = j * 2 + (lq >> 1);
            *reinterpret_cast<u32x2*>(smem + kC2fY + px * 128 + ((chunk ^ ((px >> 1) & 7)) * 16) + (lq & 1) * 8) = pack4(v);
          }
        }
      }
    }
    __syncthreads();  // B1: Y complete, X dead

    issue_x(tile + G);  // next tile's halo -> registers, in flight during phases 3-5

    // ---- phase 3: m.cv1 3x3 32 -> 32 on y1 over the 18 x 18 region -> T, zero outside the image ----
    {
      f32x4 acc[3][2];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + 64 + j * 16 + lq * 4);
      int oy[3], ox[3];
      bool val[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int f = wave + 8 * i;
        int o = 16 * f + lr;
        val[i] = f < 21 && o < 324;
        o = val[i] ? o : 0;
        oy[i] = o / 18;
        ox[i] = o - oy[i] * 18;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int r = tap / 3, q = tap % 3;
        u32x4 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(smem + kC2fWM1 + ((tap * 2 + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if (wave + 8 * i < 21) {
            const int pin = (oy[i] + r) * 20 + ox[i] + q;
            const u32x4 a = *reinterpret_cast<const u32x4*>(smem + kC2fY + pin * 128 + (((4 + lq) ^ ((pin >> 1) & 7)) * 16));
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = Elem<T>::mma(b[j], a, acc[i][j]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (val[i]) {
          const bool inside = ((unsigned)(ty0 - 1 + oy[i]) < (unsigned)p.H) && ((unsigned)(tx0 - 1 + ox[i]) < (unsigned)p.W);
          const int o = oy[i] * 18 + ox[i];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = inside ? silu_f32(acc[i][j][e]) : 0.f;
            *reinterpret_cast<u32x2*>(smem + kC2fT + o * 80 + (j * 16 + lq * 4) * 2) = pack4(v);
          }
        }
      }
    }
    __syncthreads();  // B2: T complete

    // ---- phase 4: m.cv2 3x3 32 -> 32 on T over the 16 x 16 tile (+ y1 when the Bottleneck has a shortcut) -> Y2 ----
    // ---- phase 5: cv2 1x1 96 -> 64 on [y0 | y1 | y2] -> global.  A wave owns the ADJACENT rows 2w, 2w+1 in both phases, so
    // phase 5 only reads y2 rows the wave itself wrote: no workgroup barrier between them, and once its six A fragments
    // are in registers the wave's two y2 rows (2560 B) are dead and serve as its store-transpose scratch (2304 B).
    {
      f32x4 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + 96 + j * 16 + lq * 4);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int r = tap / 3, q = tap % 3;
        u32x4 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const u32x4*>(smem + kC2fWM2 + ((tap * 2 + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = 2 * wave + i;
          const u32x4 a = *reinterpret_cast<const u32x4*>(smem + kC2fT + ((row + r) * 18 + lr + q) * 80 + lq * 16);
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = Elem<T>::mma(b[j], a, acc[i][j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 2 * wave + i;
        const int pin = (row + 2) * 20 + lr + 2;  // this pixel in the 20 x 20 region
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(acc[i][j][e]);
          if (p.shortcut) {  // x + cv2(cv1(x)): y1 channels 32 + (j*16 + lq*4 ..)
            const int chunk = 4 + j * 2 + (lq >> 1);
            const u32x2 raw = *reinterpret_cast<const u32x2*>(smem + kC2fY + pin * 128 + ((chunk ^ ((pin >> 1) & 7)) * 16) + (lq & 1) * 8);
            const t4 rr = __builtin_bit_cast(t4, raw);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += Elem<T>::to_f32(rr[e]);
          }
          *reinterpret_cast<u32x2*>(smem + kC2fY2 + (row * 16 + lr) * 80 + (j * 16 + lq * 4) * 2) = pack4(v);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = *reinterpret_cast<const f32x4*>(sbias + 128 + j * 16 + lq * 4);
      u32x4 a[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 2 * wave + i;
        const int pin = (row + 2) * 20 + lr + 2;
        a[i][0] = *reinterpret_cast<const u32x4*>(smem + kC2fY + pin * 128 + (((0 + lq) ^ ((pin >> 1) & 7)) * 16));
        a[i][1] = *reinterpret_cast<const u32x4*>(smem + kC2fY + pin * 128 + (((4 + lq) ^ ((pin >> 1) & 7)) * 16));
        a[i][2] = *reinterpret_cast<const u32x4*>(smem + kC2fY2 + (row * 16 + lr) * 80 + lq * 16);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        u32x4 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const u32x4*>(smem + kC2fW2 + ((c * 4 + j) * 64 + lane) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = Elem<T>::mma(b[j], a[i][c], acc[i][j]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the y2 rows were read above; they become the scratch
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      unsigned char* escr = smem + kC2fY2 + (2 * wave * 16) * 80;  // this wave's two y2 rows: 2560 B >= 16 px x 144 B
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 2 * wave + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f32(acc[i][j][e]);
          *reinterpret_cast<u32x2*>(escr + lr * 144 + (j * 16 + lq * 4) * 2) = pack4(v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int gy = ty0 + row;
#pragma unroll
        for (int k = 0; k < 2; ++k) {  // 16 px x 8 chunks
          const int idx = k * 64 + lane;
          const int px = idx >> 3, cc = idx & 7;
          const int gx = tx0 + px;
          const u32x4 val = *reinterpret_cast<const u32x4*>(escr + px * 144 + cc * 16);
          const bool ok = gy < p.H && gx < p.W;
          const unsigned off = ok ? (unsigned)((((size_t)(n * p.H + gy) * p.W + gx) * (size_t)p.ldy + cc * E) * sizeof(T)) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(val, yrs, off, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();  // B4: every read of Y / Y2 / scratch is done; X may be overwritten with the next tile
  }
}

}  // namespace dy

using namespace dy;

extern "C" int32_t dy_c2f_fused_supported(int32_t cin, int32_t hidden, int32_t cout, int32_t n_bottlenecks, int32_t dtype) {
  return (cin == 64 && hidden == 32 && cout == 64 && n_bottlenecks == 1 && (dtype == DY_BF16 || dtype == DY_F16)) ? 1 : 0;
}

extern "C" int32_t dy_c2f_fused(const dy_c2f_desc* d, dy_stream_t stream) {
  DY_REQUIRE(d && d->x && d->y && d->w_cv1 && d->w_m_cv1 && d->w_m_cv2 && d->w_cv2 && d->bias, DY_ERR_INVALID_ARG, "dy_c2f_fused: null pointer");
  DY_REQUIRE(dy_c2f_fused_supported(d->cin, d->hidden, d->cout, 1, d->dtype), DY_ERR_UNSUPPORTED,
             "dy_c2f_fused: built for cin 64, hidden 32, cout 64, one Bottleneck, 16-bit storage (got %d/%d/%d dtype %d)", d->cin, d->hidden, d->cout, d->dtype);
  DY_REQUIRE(d->batch > 0 && d->h > 0 && d->w > 0 && d->ld_x >= d->cin && d->ld_y >= d->cout && d->ld_x % 8 == 0 && d->ld_y % 8 == 0, DY_ERR_INVALID_ARG,
             "dy_c2f_fused: bad dims / pitches");
  DY_REQUIRE(aligned16(d->x) && aligned16(d->y) && aligned16(d->w_cv1) && aligned16(d->w_m_cv1) && aligned16(d->w_m_cv2) && aligned16(d->w_cv2) && aligned16(d->bias),
             DY_ERR_INVALID_ARG, "dy_c2f_fused: views must be 16-byte aligned");
  const long long xb = (long long)d->batch * d->h * d->w * d->ld_x * 2, yb = (long long)d->batch * d->h * d->w * d->ld_y * 2;
  DY_REQUIRE(xb < (1ll << 32) - 64 && yb < (1ll << 32) - 64, DY_ERR_UNSUPPORTED, "dy_c2f_fused: views exceed 4 GiB (buffer descriptor range)");
  C2fArgs a{};
  a.x = d->x, a.y = d->y, a.w1 = d->w_cv1, a.wm1 = d->w_m_cv1, a.wm2 = d->w_m_cv2, a.w2 = d->w_cv2, a.bias = d->bias;
  a.N = d->batch, a.H = d->h, a.W = d->w, a.ldx = d->ld_x, a.ldy = d->ld_y, a.shortcut = d->shortcut;
  a.tilesX = (d->w + 15) / 16, a.tilesY = (d->h + 15) / 16;
  a.nTiles = d->batch * a.tilesY * a.tilesX;
  a.x_bytes = (unsigned)xb, a.y_bytes = (unsigned)yb;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  int grid = 256;
  if (grid > a.nTiles) grid = a.nTiles;
  if (d->dtype == DY_BF16) {
    static const hipError_t once = hipFuncSetAttribute((const void*)c2f_fused_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)once;
    hipLaunchKernelGGL((c2f_fused_kernel<bf16_t>), dim3((unsigned)grid), dim3(512), kC2fSmem, st, a);
  } else {
    static const hipError_t once = hipFuncSetAttribute((const void*)c2f_fused_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)once;
    hipLaunchKernelGGL((c2f_fused_kernel<f16_t>), dim3((unsigned)grid), dim3(512), kC2fSmem, st, a);
  }
  return check_launch("c2f_fused_kernel");
}
