// Layout of the dy_nms workspace, shared by nms.hip and the fused filter in detect_decode.hip.
//   [counts: int32 x batch, padded to 256 B][keys: u64 x batch x P][cls: u16 x batch x anchors, padded to 256 B]
// P = next power of two >= anchors (room for the bitonic sort's padding).
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifndef DY_NS
#define DY_NS dy
#endif
namespace DY_NS {

struct NmsWs {
  int* counts;
  unsigned long long* keys;
  unsigned short* cls;
  int P;
};

static inline int nms_next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
static inline size_t nms_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline size_t nms_ws_bytes(int batch, int anchors) {
  const size_t P = (size_t)nms_next_pow2(anchors);
  return nms_align_up((size_t)batch * 4, 256) + (size_t)batch * P * 8 + nms_align_up((size_t)batch * anchors * 2, 256);
}

static inline NmsWs nms_ws_layout(void* ws, int batch, int anchors) {
  NmsWs w;
  unsigned char* b = reinterpret_cast<unsigned char*>(ws);
  w.P = nms_next_pow2(anchors);
  w.counts = reinterpret_cast<int*>(b);
  const size_t off_keys = nms_align_up((size_t)batch * 4, 256);
  w.keys = reinterpret_cast<unsigned long long*>(b + off_keys);
  w.cls = reinterpret_cast<unsigned short*>(b + off_keys + (size_t)batch * w.P * 8);
  return w;
}

}  // namespace DY_NS
