// disco_gather.hip -- pair-embedding gather for the link predictor's training step (gfx950).
//
// The reference builds each batch as
//   concat(feat_transform[edge_index[0][index]], feat_transform[edge_index[1][index]], dim=1).cpu() ... .cuda()
// (/root/reference/LargeScaleCondensing.py:188-192): two gathers, a concat and a device->host->device
// round trip per iteration.  Here one kernel writes [Z[src[p]] | Z[dst[p]]] straight into the batch
// buffer: 2 x width x 4 bytes read and written per pair, HBM bound, 128-bit accesses when the rows allow.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"

namespace disco {

template <typename V>
__global__ void k_pair_gather(const float* __restrict__ Z, int64_t n, int64_t ldz, int wv /* vectors per half row */,
                              const int64_t* __restrict__ sel, const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                              int64_t n_edges, int64_t P, float* __restrict__ out, int64_t ldo, int* __restrict__ err) {
  constexpr int VW = sizeof(V) / 4;
  int64_t t = block_id() * blockDim.x + threadIdx.x;
  int64_t per_pair = 2 * (int64_t)wv;
  if (t >= P * per_pair) return;
  int64_t p = t / per_pair;
  int rem = (int)(t - p * per_pair);
  int half = rem >= wv;
  int k = rem - half * wv;
  int64_t e = sel ? sel[p] : p;
  V v = {};
  bool ok = e >= 0 && e < n_edges;
  if (ok) {
    int64_t node = half ? dst[e] : src[e];
    ok = node >= 0 && node < n;
    if (ok) v = *reinterpret_cast<const V*>(Z + node * ldz + (int64_t)k * VW);
  }
  if (!ok && err) *err = 1;     // plain vector store of a constant: every writer writes the same value
  *reinterpret_cast<V*>(out + p * ldo + (int64_t)rem * VW) = v;
}

// out[i, :] = x[idx[i], :] -- the send buffer of the halo exchange (rows of X that peers asked for, dist.py).
// One lane per 16/8/4-byte vector, consecutive lanes along a row: coalesced on both sides; 2 * width * 4 bytes per
// row, HBM bound.  Ids are the owner's local ids, validated when the partition is built.
template <typename V>
__global__ void k_gather_rows(const float* __restrict__ x, int64_t ldx, int wv, const int64_t* __restrict__ idx, int64_t n_idx,
                              float* __restrict__ out, int64_t ldo) {
  constexpr int VW = sizeof(V) / 4;
  const int64_t t = block_id() * blockDim.x + threadIdx.x;
  if (t >= n_idx * wv) return;
  const int64_t i = t / wv;
  const int k = (int)(t - i * wv);
  *reinterpret_cast<V*>(out + i * ldo + (int64_t)k * VW) = *reinterpret_cast<const V*>(x + idx[i] * ldx + (int64_t)k * VW);
}

// Anchor neighbour features: out[dst(i), 0:width] = max_k T[map(anchor[i, k]), 0:width].
// The reference forms feat_transform[index[c]][anchor, :L*d].max(dim=1).values
// (/root/reference/LargeScaleCondensing.py:299; _induct.py:293; _Sampled.py:259-261): a row subset, an
// [n_c, k, L*d] gather and a reduction.  Here one lane owns one 16/8/4-byte vector of one output row and keeps the
// running maximum in registers: k row reads + one row write, no intermediate, HBM bound.  The maximum follows
// torch.max(dim): the first NaN met wins (it propagates), otherwise `v` replaces `m` when !(v <= m).
template <typename V>
__global__ void k_anchor_max(const float* __restrict__ T, int64_t n, int64_t ldt, int wv, const int64_t* __restrict__ row_map,
                             int64_t n_map, const int64_t* __restrict__ anchor, int64_t n_out, int k,
                             const int64_t* __restrict__ out_rows, int64_t n_dst, float* __restrict__ out, int64_t ldo,
                             int* __restrict__ err) {
  constexpr int VW = sizeof(V) / 4;
  const int64_t t = block_id() * blockDim.x + threadIdx.x;
  if (t >= n_out * wv) return;
  const int64_t i = t / wv;
  const int c = (int)(t - i * wv);
  float m[VW];
  bool ok = true, first = true;
#pragma unroll
  for (int e = 0; e < VW; ++e) m[e] = 0.0f;
  for (int j = 0; j < k; ++j) {
    int64_t a = anchor[i * k + j];
    if (row_map) {
      if (a < 0 || a >= n_map) { ok = false; continue; }
      a = row_map[a];
    }
    if (a < 0 || a >= n) { ok = false; continue; }
    const V v = *reinterpret_cast<const V*>(T + a * ldt + (int64_t)c * VW);
    const float* vf = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      if (first) m[e] = vf[e];
      else if (m[e] == m[e] && !(vf[e] <= m[e])) m[e] = vf[e];
    }
    first = false;
  }
  int64_t d = out_rows ? out_rows[i] : i;
  if (d < 0 || d >= n_dst) { ok = false; d = -1; }
  if (!ok && err) *err = 1;     // plain vector store of a constant: every writer writes the same value
  if (d >= 0) {
    V o;
    float* of = reinterpret_cast<float*>(&o);
#pragma unroll
    for (int e = 0; e < VW; ++e) of[e] = m[e];
    *reinterpret_cast<V*>(out + d * ldo + (int64_t)c * VW) = o;
  }
}

}  // namespace disco

using namespace disco;

extern "C" int disco_anchor_max_f32(const float* T, int64_t n, int64_t ldt, int64_t width, const int64_t* row_map, int64_t n_map,
                                    const int64_t* anchor, int64_t n_out, int64_t k, const int64_t* out_rows, int64_t n_dst,
                                    float* out, int64_t ldo, int* err_flag, void* stream_) {
  if (n < 0 || width < 0 || n_out < 0 || k < 1 || k > 1024 || ldt < width || ldo < width || n_map < 0 || n_dst < 0) return DISCO_EINVAL;
  if (n_out == 0 || width == 0) return DISCO_OK;
  if (!T || !anchor || !out || (n_map > 0 && !row_map)) return DISCO_EINVAL;
  if (!out_rows && n_dst < n_out) return DISCO_EINVAL;
  auto aligned = [&](int a) {
    return width % a == 0 && ldt % a == 0 && ldo % a == 0 && (reinterpret_cast<uintptr_t>(T) % (4 * a)) == 0 &&
           (reinterpret_cast<uintptr_t>(out) % (4 * a)) == 0;
  };
  const int vw = aligned(4) ? 4 : aligned(2) ? 2 : 1;
  const int64_t wv = width / vw;
  const int64_t blocks = (n_out * wv + 255) / 256;
  if (wv > 0x3fffffff) return DISCO_EUNSUPPORTED;
  DISCO_GRID(grid, blocks, 256);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t* rm = n_map > 0 ? row_map : nullptr;
  if (vw == 4)
    hipLaunchKernelGGL(k_anchor_max<float4>, grid, dim3(256), 0, stream, T, n, ldt, (int)wv, rm, n_map, anchor, n_out,
                       (int)k, out_rows, n_dst, out, ldo, err_flag);
  else if (vw == 2)
    hipLaunchKernelGGL(k_anchor_max<float2>, grid, dim3(256), 0, stream, T, n, ldt, (int)wv, rm, n_map, anchor, n_out,
                       (int)k, out_rows, n_dst, out, ldo, err_flag);
  else
    hipLaunchKernelGGL(k_anchor_max<float>, grid, dim3(256), 0, stream, T, n, ldt, (int)wv, rm, n_map, anchor, n_out,
                       (int)k, out_rows, n_dst, out, ldo, err_flag);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_gather_rows_f32(const float* x, int64_t ldx, int64_t width, const int64_t* idx, int64_t n_idx, float* out,
                                     int64_t ldo, void* stream_) {
  if (width < 0 || n_idx < 0 || ldx < width || ldo < width) return DISCO_EINVAL;
  if (n_idx == 0 || width == 0) return DISCO_OK;
  if (!x || !idx || !out) return DISCO_EINVAL;
  auto aligned = [&](int a) {
    return width % a == 0 && ldx % a == 0 && ldo % a == 0 && (reinterpret_cast<uintptr_t>(x) % (4 * a)) == 0 &&
           (reinterpret_cast<uintptr_t>(out) % (4 * a)) == 0;
  };
  const int vw = aligned(4) ? 4 : aligned(2) ? 2 : 1;
  const int64_t wv = width / vw;
  const int64_t blocks = (n_idx * wv + 255) / 256;
  if (wv > 0x3fffffff) return DISCO_EUNSUPPORTED;
  DISCO_GRID(grid, blocks, 256);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (vw == 4) hipLaunchKernelGGL(k_gather_rows<float4>, grid, dim3(256), 0, stream, x, ldx, (int)wv, idx, n_idx, out, ldo);
  else if (vw == 2) hipLaunchKernelGGL(k_gather_rows<float2>, grid, dim3(256), 0, stream, x, ldx, (int)wv, idx, n_idx, out, ldo);
  else hipLaunchKernelGGL(k_gather_rows<float>, grid, dim3(256), 0, stream, x, ldx, (int)wv, idx, n_idx, out, ldo);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_pair_gather_f32(const float* Z, int64_t n, int64_t ldz, int64_t width, const int64_t* sel,
                                     const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_pairs, float* out,
                                     int64_t ldo, int* err_flag, void* stream_) {
  if (n < 0 || width < 0 || n_pairs < 0 || n_edges < 0 || ldz < width || ldo < 2 * width) return DISCO_EINVAL;
  if (n_pairs == 0 || width == 0) return DISCO_OK;
  if (!Z || !src || !dst || !out) return DISCO_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  auto aligned = [&](int a) {
    return width % a == 0 && ldz % a == 0 && ldo % a == 0 && (reinterpret_cast<uintptr_t>(Z) % (4 * a)) == 0 &&
           (reinterpret_cast<uintptr_t>(out) % (4 * a)) == 0;
  };
  int vw = aligned(4) ? 4 : aligned(2) ? 2 : 1;
  int64_t wv = width / vw;
  int64_t total = n_pairs * 2 * wv;
  int64_t blocks = (total + 255) / 256;
  if (wv > 0x3fffffff) return DISCO_EUNSUPPORTED;
  DISCO_GRID(grid, blocks, 256);
  if (vw == 4)
    hipLaunchKernelGGL(k_pair_gather<float4>, grid, dim3(256), 0, stream, Z, n, ldz, (int)wv, sel, src, dst,
                       n_edges, n_pairs, out, ldo, err_flag);
  else if (vw == 2)
    hipLaunchKernelGGL(k_pair_gather<float2>, grid, dim3(256), 0, stream, Z, n, ldz, (int)wv, sel, src, dst,
                       n_edges, n_pairs, out, ldo, err_flag);
  else
    hipLaunchKernelGGL(k_pair_gather<float>, grid, dim3(256), 0, stream, Z, n, ldz, (int)wv, sel, src, dst,
                       n_edges, n_pairs, out, ldo, err_flag);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}
