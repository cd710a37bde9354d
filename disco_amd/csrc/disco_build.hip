// disco_build.hip -- CSR construction from raw COO edges on the device (gfx950).
//
// What it replaces (all single-threaded host code in the reference's set-up path):
//   * scipy `csr_matrix((ones, (row, col)))`        /root/reference/utils.py:104-106  (duplicates summed)
//   * PyG `to_undirected(edge_index)`                /root/reference/utils.py:99-100   (both directions, de-duplicated)
//   * `SparseTensor(row=, col=, value=).t()`         /root/reference/LargeScaleCondensing.py:486 (row-major order)
//
// Pipeline: pack (row, col) into one 64-bit key with as few significant bits as the sizes allow ->
// stable LSD radix sort (rocPRIM) carrying the value as payload -> [optional] run heads flagged,
// positions by an exclusive scan, one thread per run folds its duplicates in float64 (input order:
// the sort is stable) -> rowptr by one binary search per row.  HBM bound: ~(passes x 24 + 40) bytes
// per edge.  No atomics, results do not depend on launch geometry.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "disco_internal.h"

namespace disco {

static inline int64_t align_up_b(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int bits_for(int64_t n) { int b = 1; while (b < 63 && ((int64_t)1 << b) < n) ++b; return b; }

// keys for the given entries (+ the transposed ones behind them when `sym`); *bad counts entries outside the shape
__global__ void k_coo_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col, const float* __restrict__ val,
                           int64_t n_edges, int64_t n_rows, int64_t n_cols, int cbits, int sym,
                           uint64_t* __restrict__ keys, float* __restrict__ pay, unsigned long long* bad) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  bool oob = false;
  if (i < n_edges) {
    int64_t r = row[i], c = col[i];
    oob = r < 0 || c < 0 || r >= n_rows || c >= n_cols;
    if (oob) { r = 0; c = 0; }
    keys[i] = ((uint64_t)r << cbits) | (uint64_t)c;
    if (sym) keys[n_edges + i] = ((uint64_t)c << cbits) | (uint64_t)r;
    if (pay) {
      float v = val ? val[i] : 1.0f;
      pay[i] = v;
      if (sym) pay[n_edges + i] = v;
    }
  }
  unsigned long long oob_mask = __ballot(oob);     // taken before the branch: inside it only lane 0 is active
  if (oob_mask && (threadIdx.x & 63) == 0) atomicAdd(bad, (unsigned long long)__popcll(oob_mask));
}

__global__ void k_run_heads(const uint64_t* __restrict__ keys, int64_t m, int64_t* __restrict__ head) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  if (i < m) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one thread per run head: write the unique key, the column and the folded value
__global__ void k_fold_runs(const uint64_t* __restrict__ keys, const float* __restrict__ pay, const int64_t* __restrict__ pos,
                            int64_t m, int cbits, int first_only, uint64_t* __restrict__ ukeys, int32_t* __restrict__ col_out,
                            float* __restrict__ val_out, int64_t* __restrict__ nnz_dev) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  if (i >= m) return;
  uint64_t k = keys[i];
  bool head = i == 0 || keys[i - 1] != k;
  if (i == m - 1) *nnz_dev = pos[i] + (head ? 1 : 0);
  if (!head) return;
  int64_t p = pos[i];
  ukeys[p] = k;
  col_out[p] = (int32_t)(k & (((uint64_t)1 << cbits) - 1));
  if (val_out) {
    if (first_only || !pay) {
      val_out[p] = pay ? pay[i] : 1.0f;
    } else {
      double s = (double)pay[i];
      for (int64_t j = i + 1; j < m && keys[j] == k; ++j) s += (double)pay[j];
      val_out[p] = (float)s;
    }
  }
}

__global__ void k_keep_all(const uint64_t* __restrict__ keys, int64_t m, int cbits, int32_t* __restrict__ col_out) {
  int64_t i = block_id() * blockDim.x + threadIdx.x;
  if (i < m) col_out[i] = (int32_t)(keys[i] & (((uint64_t)1 << cbits) - 1));
}

// rowptr[r] = number of (unique) keys below (r << cbits); *count is the device-side entry count
__global__ void k_rowptr_search(const uint64_t* __restrict__ keys, const int64_t* __restrict__ count, int64_t fixed_count,
                                int64_t n_rows, int cbits, int64_t* __restrict__ rowptr) {
  int64_t r = block_id() * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  int64_t n = count ? *count : fixed_count;
  uint64_t target = (uint64_t)r << cbits;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = lo;
}

struct BuildLayout {
  int64_t m;        // entries sorted (n_edges or 2 n_edges)
  int64_t off_keys_a, off_keys_b, off_pay_a, off_pay_b, off_head, off_pos, off_cnt, off_tmp;
  size_t sort_tmp, scan_tmp;
  int64_t total;
};

static int build_layout(int64_t n_edges, int flags, bool with_val, BuildLayout& L) {
  bool sym = flags & DISCO_COO_SYMMETRISE;
  int dup = flags & DISCO_COO_DUP_MASK;
  L.m = sym ? 2 * n_edges : n_edges;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { int64_t at = o; o = align_up_b(o + bytes, 256); return at; };
  L.off_cnt = take(64);
  L.off_keys_a = take(8 * L.m);
  L.off_keys_b = take(8 * L.m);
  L.off_pay_a = take(with_val ? 4 * L.m : 0);
  L.off_pay_b = take(with_val ? 4 * L.m : 0);
  L.off_head = take(dup != DISCO_COO_DUP_KEEP ? 8 * L.m : 0);
  L.off_pos = take(dup != DISCO_COO_DUP_KEEP ? 8 * L.m : 0);
  L.sort_tmp = L.scan_tmp = 0;
  if (L.m > 0) {
    hipError_t e;
    if (with_val)
      e = rocprim::radix_sort_pairs(nullptr, L.sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const float*)nullptr,
                                    (float*)nullptr, (size_t)L.m, 0, 64);
    else
      e = rocprim::radix_sort_keys(nullptr, L.sort_tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)L.m, 0, 64);
    if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
    if (dup != DISCO_COO_DUP_KEEP) {
      e = rocprim::exclusive_scan(nullptr, L.scan_tmp, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)L.m,
                                  rocprim::plus<int64_t>());
      if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
    }
  }
  L.off_tmp = take((int64_t)(L.sort_tmp > L.scan_tmp ? L.sort_tmp : L.scan_tmp) + 256);
  L.total = o;
  return DISCO_OK;
}

}  // namespace disco

using namespace disco;

extern "C" size_t disco_coo_to_csr_workspace_bytes(int64_t n_edges, int flags, int with_values) {
  if (n_edges < 0) return 0;
  BuildLayout L;
  if (build_layout(n_edges, flags, with_values != 0, L) != DISCO_OK) return 0;
  return (size_t)L.total;
}

extern "C" int disco_coo_to_csr(const int64_t* row, const int64_t* col, const float* val, int64_t n_edges, int64_t n_rows,
                                int64_t n_cols, int flags, int64_t* rowptr, int32_t* col_out, float* val_out,
                                int64_t* nnz_out, void* ws_, size_t ws_bytes, void* stream_) {
  if (n_edges < 0 || n_rows < 0 || n_cols < 0 || !nnz_out || !rowptr) return DISCO_EINVAL;
  if (n_cols > 0x7fffffffLL || n_rows > 0xffffffffLL) return DISCO_EUNSUPPORTED;   // int32 columns, uint32 schedule
  bool sym = flags & DISCO_COO_SYMMETRISE;
  int dup = flags & DISCO_COO_DUP_MASK;
  if (dup != DISCO_COO_DUP_KEEP && dup != DISCO_COO_DUP_SUM && dup != DISCO_COO_DUP_FIRST) return DISCO_EINVAL;
  if (sym && n_rows != n_cols) return DISCO_EINVAL;
  if (val && !val_out) return DISCO_EINVAL;
  bool with_val = val_out != nullptr;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  BuildLayout L;
  int rc = build_layout(n_edges, flags, with_val, L);
  if (rc != DISCO_OK) return rc;
  if ((size_t)L.total > ws_bytes) return DISCO_EWORKSPACE;
  int cbits = bits_for(n_cols > 1 ? n_cols : 2), rbits = bits_for(n_rows > 1 ? n_rows : 2);
  if (cbits + rbits > 64) return DISCO_EUNSUPPORTED;
  *nnz_out = 0;
  if (L.m == 0) {
    DISCO_HIP_TRY(hipMemsetAsync(rowptr, 0, 8 * (size_t)(n_rows + 1), stream));
    return DISCO_OK;
  }
  if (!row || !col || !col_out || !ws_) return DISCO_EINVAL;

  char* ws = static_cast<char*>(ws_);
  unsigned long long* bad = reinterpret_cast<unsigned long long*>(ws + L.off_cnt);
  int64_t* nnz_dev = reinterpret_cast<int64_t*>(ws + L.off_cnt + 8);
  uint64_t* keys_a = reinterpret_cast<uint64_t*>(ws + L.off_keys_a);
  uint64_t* keys_b = reinterpret_cast<uint64_t*>(ws + L.off_keys_b);
  float* pay_a = with_val ? reinterpret_cast<float*>(ws + L.off_pay_a) : nullptr;
  float* pay_b = with_val ? reinterpret_cast<float*>(ws + L.off_pay_b) : nullptr;
  void* tmp = ws + L.off_tmp;

  DISCO_HIP_TRY(hipMemsetAsync(ws + L.off_cnt, 0, 64, stream));
  DISCO_GRID(eb, (n_edges + 255) / 256, 256);
  DISCO_GRID(mb, (L.m + 255) / 256, 256);
  DISCO_GRID(rb, (n_rows + 256) / 256, 256);
  hipLaunchKernelGGL(k_coo_keys, eb, dim3(256), 0, stream, row, col, val, n_edges, n_rows, n_cols, cbits, sym ? 1 : 0,
                     keys_a, pay_a, bad);
  DISCO_CHECK_LAUNCH();
  size_t st = L.sort_tmp;
  if (with_val)
    DISCO_HIP_TRY(rocprim::radix_sort_pairs(tmp, st, (const uint64_t*)keys_a, keys_b, (const float*)pay_a, pay_b, (size_t)L.m, 0,
                                            (unsigned)(cbits + rbits), stream));
  else
    DISCO_HIP_TRY(rocprim::radix_sort_keys(tmp, st, (const uint64_t*)keys_a, keys_b, (size_t)L.m, 0, (unsigned)(cbits + rbits),
                                           stream));
  if (dup == DISCO_COO_DUP_KEEP) {
    hipLaunchKernelGGL(k_keep_all, mb, dim3(256), 0, stream, (const uint64_t*)keys_b, L.m, cbits, col_out);
    DISCO_CHECK_LAUNCH();
    if (with_val) DISCO_HIP_TRY(hipMemcpyAsync(val_out, pay_b, 4 * (size_t)L.m, hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(k_rowptr_search, rb, dim3(256), 0, stream, (const uint64_t*)keys_b,
                       (const int64_t*)nullptr, L.m, n_rows, cbits, rowptr);
    DISCO_CHECK_LAUNCH();
  } else {
    int64_t* head = reinterpret_cast<int64_t*>(ws + L.off_head);
    int64_t* pos = reinterpret_cast<int64_t*>(ws + L.off_pos);
    hipLaunchKernelGGL(k_run_heads, mb, dim3(256), 0, stream, (const uint64_t*)keys_b, L.m, head);
    DISCO_CHECK_LAUNCH();
    size_t sc = L.scan_tmp;
    DISCO_HIP_TRY(rocprim::exclusive_scan(tmp, sc, (const int64_t*)head, pos, (int64_t)0, (size_t)L.m, rocprim::plus<int64_t>(),
                                          stream));
    // unique keys go back into keys_a (no longer needed)
    hipLaunchKernelGGL(k_fold_runs, mb, dim3(256), 0, stream, (const uint64_t*)keys_b, (const float*)pay_b,
                       (const int64_t*)pos, L.m, cbits, dup == DISCO_COO_DUP_FIRST ? 1 : 0, keys_a, col_out, val_out, nnz_dev);
    DISCO_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_rowptr_search, rb, dim3(256), 0, stream, (const uint64_t*)keys_a,
                       (const int64_t*)nnz_dev, (int64_t)0, n_rows, cbits, rowptr);
    DISCO_CHECK_LAUNCH();
  }
  struct { unsigned long long bad; int64_t nnz; } h;
  DISCO_HIP_TRY(hipMemcpyAsync(&h, ws + L.off_cnt, 16, hipMemcpyDeviceToHost, stream));
  DISCO_HIP_TRY(hipStreamSynchronize(stream));
  if (h.bad) return DISCO_EINVAL;          // an index outside [0,n_rows) x [0,n_cols)
  *nnz_out = dup == DISCO_COO_DUP_KEEP ? L.m : h.nnz;
  return DISCO_OK;
}

// ---- structure of mx + I without a second sort -----------------------------------------------------------
// normalize_adj adds the identity when node 0 has no self-loop (/root/reference/utils.py:425-426): every row
// needs a diagonal entry.  On a CSR with sorted columns that is a merge, not a sort: one binary search per row
// finds where the diagonal belongs and whether it is missing, an exclusive scan turns "missing" into the shift
// of every row, and an entry-parallel copy moves each entry to its new place (a zero-valued diagonal entry is
// written into the gap; the kernels of disco_norm.hip add the 1 in float64).
namespace disco {

__global__ void k_diag_probe(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n,
                             int64_t* __restrict__ missing, int64_t* __restrict__ dpos) {
  const int64_t r = block_id() * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) { missing[r] = 0; dpos[r] = rowptr[n]; return; }
  int64_t lo = rowptr[r], hi = rowptr[r + 1];
  const int64_t e = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)col[mid] < r) lo = mid + 1; else hi = mid;
  }
  dpos[r] = lo;
  missing[r] = (lo < e && (int64_t)col[lo] == r) ? 0 : 1;
}

constexpr int kInsChunk = 2048;
__global__ __launch_bounds__(256) void k_diag_insert(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const float* __restrict__ val, int64_t n, int64_t nnz,
                                                     const int64_t* __restrict__ missing, const int64_t* __restrict__ shift,
                                                     const int64_t* __restrict__ dpos, int64_t* __restrict__ rowptr_out,
                                                     int32_t* __restrict__ col_out, float* __restrict__ val_out) {
  // rows: new row pointers and the inserted diagonal entries (grid-stride over rows)
  for (int64_t r = block_id() * 256 + threadIdx.x; r <= n; r += (int64_t)gridDim.x * gridDim.y * 256) {
    rowptr_out[r] = rowptr[r] + shift[r];
    if (r < n && missing[r]) {
      const int64_t p = dpos[r] + shift[r];
      col_out[p] = (int32_t)r;
      val_out[p] = 0.0f;
    }
  }
  // entries: this workgroup's chunk, rows found by binary search inside the chunk's row range
  const int64_t c0 = block_id() * kInsChunk;
  const int64_t c1 = c0 + kInsChunk < nnz ? c0 + kInsChunk : nnz;
  if (c0 >= c1) return;
  auto row_of = [&](int64_t lo, int64_t hi, int64_t i) {
    while (lo < hi) {
      const int64_t mid = (lo + hi + 1) >> 1;
      if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  const int64_t r_lo = row_of(0, n - 1, c0);
  const int64_t r_hi = row_of(r_lo, n - 1, c1 - 1);
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
    const int64_t r = row_of(r_lo, r_hi, i);
    const int64_t p = i + shift[r] + ((missing[r] && i >= dpos[r]) ? 1 : 0);
    col_out[p] = col[i];
    val_out[p] = val ? val[i] : 1.0f;
  }
}

}  // namespace disco

extern "C" size_t disco_csr_insert_diagonal_workspace_bytes(int64_t n_rows) {
  if (n_rows < 0) return 0;
  size_t scan_tmp = 0;
  if (rocprim::exclusive_scan(nullptr, scan_tmp, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0, (size_t)(n_rows + 1),
                              rocprim::plus<int64_t>()) != hipSuccess)
    return 0;
  return (size_t)(3 * align_up_b(8 * (n_rows + 1), 256)) + scan_tmp + 512;
}

extern "C" int disco_csr_insert_diagonal(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                         int64_t nnz, int64_t* rowptr_out, int32_t* col_out, float* val_out,
                                         int64_t* nnz_out, void* ws_, size_t ws_bytes, void* stream_) {
  if (n_rows < 0 || nnz < 0 || !nnz_out) return DISCO_EINVAL;
  if (n_rows > 0x7fffffffLL) return DISCO_EUNSUPPORTED;
  *nnz_out = 0;
  if (n_rows == 0) return DISCO_OK;
  if (!rowptr || !rowptr_out || !col_out || !val_out || !ws_ || (nnz > 0 && !col)) return DISCO_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t slot = align_up_b(8 * (n_rows + 1), 256);
  size_t scan_tmp = 0;
  DISCO_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, (const int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                        (size_t)(n_rows + 1), rocprim::plus<int64_t>()));
  if ((size_t)(3 * slot) + scan_tmp + 512 > ws_bytes) return DISCO_EWORKSPACE;
  char* ws = static_cast<char*>(ws_);
  int64_t* missing = reinterpret_cast<int64_t*>(ws);
  int64_t* shift = reinterpret_cast<int64_t*>(ws + slot);
  int64_t* dpos = reinterpret_cast<int64_t*>(ws + 2 * slot);
  void* tmp = ws + 3 * slot;
  DISCO_GRID(g_probe, (n_rows + 256) / 256, 256);
  hipLaunchKernelGGL(k_diag_probe, g_probe, dim3(256), 0, stream, rowptr, col, n_rows, missing, dpos);
  DISCO_CHECK_LAUNCH();
  DISCO_HIP_TRY(rocprim::exclusive_scan(tmp, scan_tmp, (const int64_t*)missing, shift, (int64_t)0, (size_t)(n_rows + 1),
                                        rocprim::plus<int64_t>(), stream));
  int64_t blocks = (nnz + kInsChunk - 1) / kInsChunk;
  const int64_t row_blocks = (n_rows + 256) / 256 < 4096 ? (n_rows + 256) / 256 : 4096;
  if (blocks < row_blocks) blocks = row_blocks;
  DISCO_GRID(g_ins, blocks, 256);
  hipLaunchKernelGGL(k_diag_insert, g_ins, dim3(256), 0, stream, rowptr, col, val, n_rows, nnz,
                     (const int64_t*)missing, (const int64_t*)shift, (const int64_t*)dpos, rowptr_out, col_out, val_out);
  DISCO_CHECK_LAUNCH();
  int64_t added = 0;
  DISCO_HIP_TRY(hipMemcpyAsync(&added, shift + n_rows, sizeof(added), hipMemcpyDeviceToHost, stream));
  DISCO_HIP_TRY(hipStreamSynchronize(stream));
  *nnz_out = nnz + added;
  return DISCO_OK;
}

// ---- exact degree count by sorting (edge-index branch of gcn_norm, /root/reference/utils.py:244) -----------------
// 52 M random node ids are 52 M global atomics at ~26 G/s (k_degree_count: 2.0 ms on the products-shaped graph), while
// a radix sort of 32-bit keys over log2(N) bits followed by run boundaries moves each id four times, coalesced.
namespace disco {

__global__ __launch_bounds__(256) void k_ids_to_u32(const int64_t* __restrict__ idx, int64_t E, int64_t N, uint32_t* __restrict__ key) {
  const int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= E) return;
  const int64_t v = idx[i];
  key[i] = (v >= 0 && v < N) ? (uint32_t)v : (uint32_t)N;      // ids outside [0, N) sort behind every node and are ignored
}

__global__ __launch_bounds__(256) void k_zero_f32(float* __restrict__ p, int64_t n) {
  const int64_t i = block_id() * 256 + threadIdx.x;
  if (i < n) p[i] = 0.0f;
}

// Sorted keys: the thread at the first entry of a node's run records where it starts, the thread at its last entry
// where it ends (into `deg`, as integers); a pass over the nodes turns the difference into float32 like the atomic
// version does (uint32 -> float32).  Nodes without a run keep the zero k_zero_f32 wrote.
__global__ __launch_bounds__(256) void k_run_bounds(const uint32_t* __restrict__ key, int64_t E, int64_t N,
                                                    uint32_t* __restrict__ first, uint32_t* __restrict__ last) {
  const int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= E) return;
  const uint32_t k = key[i];
  if ((int64_t)k >= N) return;
  if (i == 0 || key[i - 1] != k) first[k] = (uint32_t)i;
  if (i == E - 1 || key[i + 1] != k) last[k] = (uint32_t)(i + 1);
}

__global__ __launch_bounds__(256) void k_run_counts(const uint32_t* __restrict__ first, float* __restrict__ deg, int64_t N) {
  const int64_t k = block_id() * 256 + threadIdx.x;
  if (k >= N) return;
  const uint32_t end = __float_as_uint(deg[k]);                // 0: no entry names this node
  deg[k] = end ? (float)(end - first[k]) : 0.0f;
}

struct DegLayout { size_t off_a, off_b, off_first, off_tmp, sort_tmp, total; int bits; };
static int deg_layout(int64_t E, int64_t N, DegLayout& L) {
  if (E < 0 || N < 0 || N >= 0xffffffffLL) return DISCO_EUNSUPPORTED;
  L.bits = 1;
  while ((1ll << L.bits) <= N) ++L.bits;                       // keys go up to N (the "ignored" key)
  L.sort_tmp = 0;
  if (E > 0 && rocprim::radix_sort_keys(nullptr, L.sort_tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)E, 0,
                                        (unsigned)L.bits) != hipSuccess)
    return DISCO_EHIP;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  L.off_a = 0;
  L.off_b = up((size_t)E * 4);
  L.off_first = L.off_b + up((size_t)E * 4);
  L.off_tmp = L.off_first + up((size_t)N * 4);
  L.total = L.off_tmp + up(L.sort_tmp) + 256;
  return DISCO_OK;
}

}  // namespace disco

extern "C" size_t disco_degree_count_sorted_workspace_bytes(int64_t E, int64_t N) {
  DegLayout L;
  return deg_layout(E, N, L) == DISCO_OK ? L.total : 0;
}

extern "C" int disco_degree_count_sorted_f32(const int64_t* idx, int64_t E, int64_t N, float* deg, void* workspace,
                                             size_t workspace_bytes, void* stream_) {
  if (E < 0 || N < 0) return DISCO_EINVAL;
  if (N == 0) return DISCO_OK;
  if (!deg || (E > 0 && !idx)) return DISCO_EINVAL;
  if (E > 0xffffffffLL) return DISCO_EUNSUPPORTED;             // counts are formed in 32 bits, as in disco_degree_count_f32
  DegLayout L;
  int rc = deg_layout(E, N, L);
  if (rc != DISCO_OK) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  DISCO_GRID(nb, (N + 255) / 256, 256);
  DISCO_GRID(eb, (E + 255) / 256, 256);
  hipLaunchKernelGGL(k_zero_f32, nb, dim3(256), 0, stream, deg, N);
  DISCO_CHECK_LAUNCH();
  if (E == 0) return DISCO_OK;
  if (!workspace || workspace_bytes < L.total) return DISCO_EWORKSPACE;
  char* ws = static_cast<char*>(workspace);
  uint32_t* a = reinterpret_cast<uint32_t*>(ws + L.off_a);
  uint32_t* b = reinterpret_cast<uint32_t*>(ws + L.off_b);
  hipLaunchKernelGGL(k_ids_to_u32, eb, dim3(256), 0, stream, idx, E, N, a);
  DISCO_CHECK_LAUNCH();
  size_t st = L.sort_tmp;
  DISCO_HIP_TRY(rocprim::radix_sort_keys(ws + L.off_tmp, st, (const uint32_t*)a, b, (size_t)E, 0, (unsigned)L.bits, stream));
  uint32_t* first = reinterpret_cast<uint32_t*>(ws + L.off_first);
  hipLaunchKernelGGL(k_run_bounds, eb, dim3(256), 0, stream, (const uint32_t*)b, E, N, first, reinterpret_cast<uint32_t*>(deg));
  DISCO_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_run_counts, nb, dim3(256), 0, stream, (const uint32_t*)first, deg, N);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

