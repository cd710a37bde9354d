// disco_schedule.hip -- degree-binned row schedule for the CSR SpMM (gfx950).
//
// Built once per graph on the device:
//   1. every row gets a one-byte key: 0 for hub rows (nnz > hub_threshold), then
//      62 - floor(log2 nnz) (heavier rows first), 63 for empty rows;
//   2. a stable radix sort of (key, row) gives `perm`: hubs first, then rows by falling degree
//      class, row ids ascending inside a class (keeps the graph's own locality);
//   3. hub rows are cut into pieces of at most hub_piece entries; rows with more than one piece
//      get consecutive partial-row slots for the fix-up kernel.
// No counterpart in the reference (torch_sparse launches one warp per row in natural order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "disco_internal.h"

namespace disco {

// Defaults (hub_threshold = 0 / hub_piece = 0).  One lane group walks a non-hub row serially at ~10 entries
// per microsecond, so the longest such row bounds the launch: on a small graph (ogbn-arxiv, 0.1 ms per hop) a
// 2048-entry row alone is 0.2 ms.  Threshold = nnz / 4096 rounded down to a power of two, clamped to
// [256, 2048]; piece = 4 x threshold clamped to [1024, 8192]  (profiles/r1/tune_hub_defaults.log).
static inline int64_t default_hub_threshold(int64_t nnz) {
  int64_t t = 256;
  while (t * 2 <= nnz / 4096 && t < 2048) t *= 2;
  return t;
}
static inline int64_t default_hub_piece(int64_t threshold) {
  int64_t p = 4 * threshold;
  return p < 1024 ? 1024 : (p > 8192 ? 8192 : p);
}

struct Counters {
  unsigned long long n_hub;
  unsigned long long max_nnz;
  unsigned long long n_pieces;
  unsigned long long n_partials;
  unsigned long long n_multi;
};

template <typename KEY>
__global__ void k_row_keys(const int64_t* __restrict__ rowptr, int64_t n_rows, int64_t hub_threshold, int natural,
                           const uint32_t* __restrict__ row_group, KEY* __restrict__ keys, uint32_t* __restrict__ rows,
                           Counters* cnt) {
  int64_t r = block_id() * blockDim.x + threadIdx.x;
  unsigned long long nnz = 0;
  bool hub = false;
  if (r < n_rows) {
    nnz = (unsigned long long)(rowptr[r + 1] - rowptr[r]);
    hub = (int64_t)nnz > hub_threshold;
    uint8_t key;
    if (hub) key = 0;
    else if (natural) key = 1;
    else if (nnz == 0) key = 63;
    else { int k = 62 - (63 - __clzll((long long)nnz)); key = (uint8_t)(k < 2 ? 2 : k); }
    if constexpr (sizeof(KEY) == 8) {
      // locality-grouped order: hubs first, then rows group by group (rows of one group share neighbours: their
      // gathers hit the same cache), heavier rows first inside a group
      keys[r] = hub ? (KEY)0 : (((KEY)row_group[r] + 1) << 8) | (KEY)key;
    } else {
      keys[r] = key;
    }
    rows[r] = (uint32_t)r;
  }
  // one atomic per wave
  unsigned long long hub_mask = __ballot(hub);
  unsigned long long m = nnz;
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) {
    if (hub_mask) atomicAdd(&cnt->n_hub, (unsigned long long)__popcll(hub_mask));
    if (m) atomicMax(&cnt->max_nnz, m);
  }
}

// per hub row i (perm[i], i < n_hub): number of pieces, partial slots, multi flag
__global__ void k_hub_count(const int64_t* __restrict__ rowptr, const uint32_t* __restrict__ perm, const Counters* cnt,
                            int64_t hub_piece, int64_t n_max, int64_t* __restrict__ pieces, int64_t* __restrict__ slots,
                            int64_t* __restrict__ multi) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  if (i >= n_max) return;
  int64_t p = 0;
  if (i < (int64_t)cnt->n_hub) {
    int64_t r = perm[i];
    int64_t nnz = rowptr[r + 1] - rowptr[r];
    p = (nnz + hub_piece - 1) / hub_piece;
  }
  pieces[i] = p;
  slots[i] = p > 1 ? p : 0;
  multi[i] = p > 1 ? 1 : 0;
}

__global__ void k_hub_emit(const int64_t* __restrict__ rowptr, const uint32_t* __restrict__ perm, Counters* cnt,
                           int64_t hub_piece, int64_t n_max, const int64_t* __restrict__ piece_first,
                           const int64_t* __restrict__ slot_first, const int64_t* __restrict__ multi_idx,
                           int32_t* __restrict__ piece_row, int64_t* __restrict__ piece_begin,
                           int64_t* __restrict__ piece_end, int32_t* __restrict__ piece_slot,
                           int32_t* __restrict__ multi_row, int64_t* __restrict__ multi_first) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  int64_t n_hub = (int64_t)cnt->n_hub;
  if (i >= n_max || i >= n_hub) return;
  int64_t r = perm[i];
  int64_t b = rowptr[r], e = rowptr[r + 1];
  int64_t np = (e - b + hub_piece - 1) / hub_piece;
  int64_t p0 = piece_first[i];
  // even-sized pieces, each a multiple of 64 entries except the last
  int64_t per = (e - b + np - 1) / np;
  per = (per + 63) / 64 * 64;
  for (int64_t k = 0; k < np; ++k) {
    int64_t pb = b + k * per;
    int64_t pe = pb + per < e ? pb + per : e;
    if (pb > e) pb = e;
    piece_row[p0 + k] = (int32_t)r;
    piece_begin[p0 + k] = pb;
    piece_end[p0 + k] = pe;
    piece_slot[p0 + k] = np > 1 ? (int32_t)(slot_first[i] + k) : -1;
  }
  if (np > 1) {
    multi_row[multi_idx[i]] = (int32_t)r;
    multi_first[multi_idx[i]] = slot_first[i];
  }
  if (i == n_hub - 1) {
    int64_t tot_p = p0 + np;
    int64_t tot_s = slot_first[i] + (np > 1 ? np : 0);
    int64_t tot_m = multi_idx[i] + (np > 1 ? 1 : 0);
    multi_first[tot_m] = tot_s;
    cnt->n_pieces = (unsigned long long)tot_p;
    cnt->n_partials = (unsigned long long)tot_s;
    cnt->n_multi = (unsigned long long)tot_m;
  }
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Layout {
  int64_t n_hub_max, n_piece_max;
  int64_t off_perm, off_piece_row, off_piece_begin, off_piece_end, off_piece_slot, off_multi_row, off_multi_first;
  int64_t off_cnt, off_keys_in, off_keys_out, off_rows_in, off_cntarr, off_scan_a, off_scan_b, off_scan_c, off_tmp;
  size_t sort_tmp, scan_tmp;
  int64_t total;
};

static int make_layout(int64_t n_rows, int64_t nnz, int64_t T, int64_t P, Layout& L, bool grouped = false) {
  L.n_hub_max = nnz / (T + 1);
  if (L.n_hub_max > n_rows) L.n_hub_max = n_rows;
  L.n_piece_max = nnz / P + L.n_hub_max + 1;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t at = o; o = align_up(o + bytes, 256); return at; };
  // persistent part (read by the SpMM kernels)
  L.off_perm = take(4 * n_rows);
  L.off_piece_row = take(4 * L.n_piece_max);
  L.off_piece_begin = take(8 * L.n_piece_max);
  L.off_piece_end = take(8 * L.n_piece_max);
  L.off_piece_slot = take(4 * L.n_piece_max);
  L.off_multi_row = take(4 * (L.n_hub_max + 1));
  L.off_multi_first = take(8 * (L.n_hub_max + 2));
  // build-time scratch
  L.off_cnt = take(sizeof(Counters));
  L.off_keys_in = take(grouped ? 8 * n_rows : n_rows);
  L.off_keys_out = take(grouped ? 8 * n_rows : n_rows);
  L.off_rows_in = take(4 * n_rows);
  L.off_cntarr = take(8 * (L.n_hub_max + 1));
  L.off_scan_a = take(8 * (L.n_hub_max + 1));
  L.off_scan_b = take(8 * (L.n_hub_max + 1));
  L.off_scan_c = take(8 * (L.n_hub_max + 1));
  L.sort_tmp = 0;
  L.scan_tmp = 0;
  if (n_rows > 0) {
    hipError_t e = grouped ? rocprim::radix_sort_pairs(nullptr, L.sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n_rows, 0, 41)
                           : rocprim::radix_sort_pairs(nullptr, L.sort_tmp, (const uint8_t*)nullptr, (uint8_t*)nullptr,
                                                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n_rows, 0, 8);
    if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
  }
  if (L.n_hub_max > 0) {
    hipError_t e = rocprim::exclusive_scan(nullptr, L.scan_tmp, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                           (size_t)L.n_hub_max, rocprim::plus<int64_t>());
    if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
  }
  L.off_tmp = take((int64_t)(L.sort_tmp > L.scan_tmp ? L.sort_tmp : L.scan_tmp) + 256);
  L.total = o;
  return DISCO_OK;
}

}  // namespace disco

using namespace disco;

static size_t schedule_ws_bytes(int64_t n_rows, int64_t nnz, int64_t hub_threshold, int64_t hub_piece, bool grouped) {
  if (n_rows < 0 || nnz < 0) return 0;
  if (hub_threshold < 0) hub_threshold = -hub_threshold;
  int64_t T = hub_threshold > 0 ? hub_threshold : default_hub_threshold(nnz);
  int64_t P = hub_piece > 0 ? hub_piece : default_hub_piece(T);
  Layout L;
  if (make_layout(n_rows, nnz, T, P, L, grouped) != DISCO_OK) return 0;
  return (size_t)L.total;
}

extern "C" size_t disco_schedule_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t hub_threshold, int64_t hub_piece) {
  return schedule_ws_bytes(n_rows, nnz, hub_threshold, hub_piece, false);
}

extern "C" size_t disco_schedule_grouped_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t hub_threshold, int64_t hub_piece) {
  return schedule_ws_bytes(n_rows, nnz, hub_threshold, hub_piece, true);
}

static int schedule_build(const int64_t* rowptr, int64_t n_rows, int64_t nnz, int64_t hub_threshold, int64_t hub_piece,
                          const uint32_t* row_group, void* sched_ws, size_t sched_ws_bytes, disco_schedule_info* info,
                          void* stream_);

extern "C" int disco_schedule_build(const int64_t* rowptr, int64_t n_rows, int64_t nnz, int64_t hub_threshold,
                                    int64_t hub_piece, void* sched_ws, size_t sched_ws_bytes,
                                    disco_schedule_info* info, void* stream_) {
  return schedule_build(rowptr, n_rows, nnz, hub_threshold, hub_piece, nullptr, sched_ws, sched_ws_bytes, info, stream_);
}

extern "C" int disco_schedule_build_grouped(const int64_t* rowptr, int64_t n_rows, int64_t nnz, int64_t hub_threshold,
                                            int64_t hub_piece, const uint32_t* row_group, void* sched_ws,
                                            size_t sched_ws_bytes, disco_schedule_info* info, void* stream_) {
  if (!row_group && n_rows > 0) return DISCO_EINVAL;
  return schedule_build(rowptr, n_rows, nnz, hub_threshold, hub_piece, row_group, sched_ws, sched_ws_bytes, info, stream_);
}

static int schedule_build(const int64_t* rowptr, int64_t n_rows, int64_t nnz, int64_t hub_threshold, int64_t hub_piece,
                          const uint32_t* row_group, void* sched_ws, size_t sched_ws_bytes, disco_schedule_info* info,
                          void* stream_) {
  const bool grouped = row_group != nullptr;
  if (!info || n_rows < 0 || nnz < 0) return DISCO_EINVAL;
  if (n_rows > 0x7fffffffLL) return DISCO_EUNSUPPORTED;     // piece_row / multi_row are int32, perm uint32
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int natural = 0;
  if (hub_threshold < 0) { natural = 1; hub_threshold = -hub_threshold; }  // negative threshold: keep natural row order
  int64_t T = hub_threshold > 0 ? hub_threshold : default_hub_threshold(nnz);
  int64_t P = hub_piece > 0 ? hub_piece : default_hub_piece(T);
  if (P < 64) P = 64;
  Layout L;
  int rc = make_layout(n_rows, nnz, T, P, L, grouped);
  if (rc != DISCO_OK) return rc;
  if ((size_t)L.total > sched_ws_bytes) return DISCO_EWORKSPACE;
  if (n_rows > 0 && (!rowptr || !sched_ws)) return DISCO_EINVAL;
  *info = disco_schedule_info{};
  info->n_rows = n_rows; info->nnz = nnz; info->hub_threshold = T; info->hub_piece = P;
  info->off_perm = L.off_perm; info->off_piece_row = L.off_piece_row; info->off_piece_begin = L.off_piece_begin;
  info->off_piece_end = L.off_piece_end; info->off_piece_slot = L.off_piece_slot;
  info->off_multi_row = L.off_multi_row; info->off_multi_first = L.off_multi_first; info->ws_bytes = L.total;
  info->xcd_run = grouped ? 1024 : 0;    // rows; callers that know their group sizes set it to the typical size (sparse.py)
  if (n_rows == 0) return DISCO_OK;

  char* ws = static_cast<char*>(sched_ws);
  Counters* cnt = reinterpret_cast<Counters*>(ws + L.off_cnt);
  uint32_t* rows_in = reinterpret_cast<uint32_t*>(ws + L.off_rows_in);
  uint32_t* perm = reinterpret_cast<uint32_t*>(ws + L.off_perm);
  void* tmp = ws + L.off_tmp;

  DISCO_HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(Counters), stream));
  {
    DISCO_GRID(key_grid, (n_rows + 255) / 256, 256);
    size_t sort_tmp = L.sort_tmp;
    if (grouped) {
      uint64_t* keys_in = reinterpret_cast<uint64_t*>(ws + L.off_keys_in);
      uint64_t* keys_out = reinterpret_cast<uint64_t*>(ws + L.off_keys_out);
      hipLaunchKernelGGL(k_row_keys<uint64_t>, key_grid, dim3(256), 0, stream, rowptr, n_rows, T, natural,
                         row_group, keys_in, rows_in, cnt);
      DISCO_CHECK_LAUNCH();
      DISCO_HIP_TRY(rocprim::radix_sort_pairs(tmp, sort_tmp, (const uint64_t*)keys_in, keys_out, (const uint32_t*)rows_in,
                                              perm, (size_t)n_rows, 0, 41, stream));
    } else {
      uint8_t* keys_in = reinterpret_cast<uint8_t*>(ws + L.off_keys_in);
      uint8_t* keys_out = reinterpret_cast<uint8_t*>(ws + L.off_keys_out);
      hipLaunchKernelGGL(k_row_keys<uint8_t>, key_grid, dim3(256), 0, stream, rowptr, n_rows, T, natural,
                         (const uint32_t*)nullptr, keys_in, rows_in, cnt);
      DISCO_CHECK_LAUNCH();
      DISCO_HIP_TRY(rocprim::radix_sort_pairs(tmp, sort_tmp, (const uint8_t*)keys_in, keys_out, (const uint32_t*)rows_in,
                                              perm, (size_t)n_rows, 0, 8, stream));
    }
  }
  if (L.n_hub_max > 0) {
    int64_t* pieces = reinterpret_cast<int64_t*>(ws + L.off_cntarr);
    int64_t* a = reinterpret_cast<int64_t*>(ws + L.off_scan_a);
    int64_t* b = reinterpret_cast<int64_t*>(ws + L.off_scan_b);
    int64_t* c = reinterpret_cast<int64_t*>(ws + L.off_scan_c);
    DISCO_GRID(hub_grid, (L.n_hub_max + 255) / 256, 256);
    // slots/multi are produced into b/c then scanned in place after pieces -> a
    hipLaunchKernelGGL(k_hub_count, hub_grid, dim3(256), 0, stream, rowptr, perm, cnt, P, L.n_hub_max, pieces,
                       b, c);
    DISCO_CHECK_LAUNCH();
    size_t scan_tmp = L.scan_tmp;
    DISCO_HIP_TRY(rocprim::exclusive_scan(tmp, scan_tmp, (const int64_t*)pieces, a, (int64_t)0, (size_t)L.n_hub_max,
                                          rocprim::plus<int64_t>(), stream));
    // in-place scans are not guaranteed by rocPRIM: reuse `pieces` as the output of the 2nd scan,
    // then b as the output of the 3rd.
    scan_tmp = L.scan_tmp;
    DISCO_HIP_TRY(rocprim::exclusive_scan(tmp, scan_tmp, (const int64_t*)b, pieces, (int64_t)0, (size_t)L.n_hub_max,
                                          rocprim::plus<int64_t>(), stream));
    scan_tmp = L.scan_tmp;
    DISCO_HIP_TRY(rocprim::exclusive_scan(tmp, scan_tmp, (const int64_t*)c, b, (int64_t)0, (size_t)L.n_hub_max,
                                          rocprim::plus<int64_t>(), stream));
    hipLaunchKernelGGL(k_hub_emit, hub_grid, dim3(256), 0, stream, rowptr, perm, cnt, P, L.n_hub_max,
                       (const int64_t*)a, (const int64_t*)pieces, (const int64_t*)b,
                       reinterpret_cast<int32_t*>(ws + L.off_piece_row), reinterpret_cast<int64_t*>(ws + L.off_piece_begin),
                       reinterpret_cast<int64_t*>(ws + L.off_piece_end), reinterpret_cast<int32_t*>(ws + L.off_piece_slot),
                       reinterpret_cast<int32_t*>(ws + L.off_multi_row), reinterpret_cast<int64_t*>(ws + L.off_multi_first));
    DISCO_CHECK_LAUNCH();
  }
  Counters h;
  DISCO_HIP_TRY(hipMemcpyAsync(&h, cnt, sizeof(Counters), hipMemcpyDeviceToHost, stream));
  DISCO_HIP_TRY(hipStreamSynchronize(stream));
  info->n_hub_rows = (int64_t)h.n_hub;
  info->n_hub_pieces = (int64_t)h.n_pieces;
  info->n_hub_partials = (int64_t)h.n_partials;
  info->n_multi_rows = (int64_t)h.n_multi;
  info->max_row_nnz = (int64_t)h.max_nnz;
  if (info->n_hub_rows > L.n_hub_max || info->n_hub_pieces > L.n_piece_max) return DISCO_EINVAL;  // rowptr inconsistent with nnz
  return DISCO_OK;
}
