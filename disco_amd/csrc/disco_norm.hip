// disco_norm.hip -- GCN normalisation kernels (gfx950) and the library's small utilities.
//
// A3 semantics: /root/reference/utils.py:406-435 (normalize_adj; float64, row degree) + :634 (one
//               rounding to float32).
// A1 semantics: /root/reference/utils.py:242-247 (gcn_norm edge-index branch; float32, degree by
//               column, left-to-right product).  Same code at models/convs_large/gcn_conv.py:65-70.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "disco_internal.h"

namespace disco {

static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }
Tuning& tuning() { static Tuning t; return t; }

// ---- CSR row sums ------------------------------------------------------------------------------
// add_identity != 0: sums of `mx + I` in float64 (scipy promotes float32 + eye() to float64,
//                    utils.py:426-427); 16 lanes per row, fixed-order tree.
// add_identity == 0: scipy keeps the float32 matrix in float32 all the way (utils.py:427-432), and its
//                    row sum is a sequential float32 accumulation in CSR order: one lane per row.
constexpr int64_t kLongRow = 4096;
constexpr long long kLongMarker = 0x7ff8d15c0d15c000LL;   // a quiet NaN payload

__global__ __launch_bounds__(256) void k_csr_rowsum(const int64_t* __restrict__ rowptr, const float* __restrict__ val,
                                                    int64_t n_rows, int add_identity, double* __restrict__ out) {
  const int gl = threadIdx.x & 15;
  int64_t row = (block_id() * 256 + threadIdx.x) >> 4;
  int64_t b = 0, e = 0;
  if (row < n_rows) { b = rowptr[row]; e = rowptr[row + 1]; }
  if (!add_identity) {
    if (row < n_rows && gl == 0) {
      float s = 0.0f;
      if (val) { for (int64_t i = b; i < e; ++i) s += val[i]; }
      else s = (float)(e - b);
      out[row] = (double)s;
    }
    return;
  }
  double s = 0.0;
  if (val && e - b > kLongRow) {
    // a 16-lane group would walk a hub row (ogbn-products: 600 k entries) for milliseconds: mark it for
    // k_csr_rowsum_long (the marker is a NaN with a fixed payload, which no sum of finite inputs produces)
    if (gl == 0) out[row] = __longlong_as_double(kLongMarker);
    return;
  }
  if (val) { for (int64_t i = b + gl; i < e; i += 16) s += (double)val[i]; }
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
  if (!val) s = (double)(e - b);
  s += 1.0;  // mx + I (utils.py:426): every row sum grows by one
  if (row < n_rows && gl == 0) out[row] = s;
}

// rows marked by k_csr_rowsum: a whole 1024-thread workgroup sums one row (strided float64 partials, LDS tree
// in a fixed order); a workgroup scans the 1024 rows it owns for markers.
__global__ __launch_bounds__(1024) void k_csr_rowsum_long(const int64_t* __restrict__ rowptr, const float* __restrict__ val,
                                                          int64_t n_rows, double* __restrict__ out) {
  __shared__ double part[1024];
  __shared__ int marked[1024];
  const int t = threadIdx.x;
  const int64_t r0 = block_id() * 1024;
  const int64_t mine = r0 + t;
  marked[t] = mine < n_rows && __double_as_longlong(out[mine]) == kLongMarker;
  __syncthreads();
  for (int k = 0; k < 1024; ++k) {
    if (!marked[k]) continue;                    // uniform: read from LDS
    const int64_t row = r0 + k;
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    // four independent strided partial sums per thread (four loads in flight), folded in a fixed order
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = b + t;
    for (; i + 3 * 1024 < e; i += 4 * 1024) {
      const float v0 = val[i], v1 = val[i + 1024], v2 = val[i + 2048], v3 = val[i + 3072];
      s0 += (double)v0; s1 += (double)v1; s2 += (double)v2; s3 += (double)v3;
    }
    for (; i < e; i += 1024) s0 += (double)val[i];
    part[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (t < o) part[t] += part[t + o];
      __syncthreads();
    }
    if (t == 0) out[row] = part[0] + 1.0;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_inv_sqrt_norm(double* __restrict__ d, int64_t n, int add_identity) {
  int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= n) return;
  // np.power(rowsum, -1/2); inf -> 0 (utils.py:428-429).  In the float32 case the result is rounded to
  // float32 (numpy's own float32 pow is only accurate to a few ulp and differs between CPUs).
  double r = 1.0 / sqrt(d[i]);
  if (!add_identity) r = (double)(float)r;
  if (isinf(r)) r = 0.0;
  d[i] = r;
}

// Entry-parallel: a workgroup owns a chunk of kScaleChunk consecutive entries, finds the rows its chunk
// touches with one binary search per end and each entry's row with a short search inside that range -- the
// work per workgroup does not depend on row lengths (a row-per-lane-group version spent 30 ms on the one
// 600 k-entry row of ogbn-products).
constexpr int kScaleChunk = 2048;
__device__ __forceinline__ int64_t row_of_entry(const int64_t* __restrict__ rowptr, int64_t lo, int64_t hi, int64_t i) {
  // largest r in [lo, hi] with rowptr[r] <= i
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (rowptr[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_gcn_scale_csr(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ vin, int64_t n_rows, int64_t nnz,
                                                       int add_identity, const double* __restrict__ r,
                                                       float* __restrict__ vout) {
  const int64_t c0 = block_id() * kScaleChunk;
  const int64_t c1 = c0 + kScaleChunk < nnz ? c0 + kScaleChunk : nnz;
  if (c0 >= c1) return;
  const int64_t r_lo = row_of_entry(rowptr, 0, n_rows - 1, c0);
  const int64_t r_hi = row_of_entry(rowptr, r_lo, n_rows - 1, c1 - 1);
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
    const int64_t row = row_of_entry(rowptr, r_lo, r_hi, i);
    const double rr = r[row];
    const int64_t c = col[i];
    if (add_identity) {
      double v = vin ? (double)vin[i] : 1.0;
      if (c == row) v += 1.0;  // the diagonal entry of mx + I, in float64 like scipy
      vout[i] = (float)((rr * v) * r[c]);
    } else {
      float v = vin ? vin[i] : 1.0f;
      vout[i] = __fmul_rn(__fmul_rn((float)rr, v), (float)r[c]);  // float32 products, left to right
    }
  }
}

// ---- edge-index branch ------------------------------------------------------------------------------
// Integer counts: exact and order independent.  A power-law graph sends a large share of its edges to a few
// hub nodes, and same-address global atomics serialise (7.2 ms for 52 M edges, ogbn-products shape), so each
// workgroup first counts into a small LDS table keyed by node (first come, first served: hubs, being
// frequent, claim their slots early); edges that lose the slot go to the global counter directly, the
// table is flushed with one global atomic per occupied slot.  Out-of-range ids are ignored.
constexpr int kDegSlots = 2048;
constexpr int kDegChunk = 16384;   // edges per workgroup
__global__ __launch_bounds__(256) void k_degree_count(const int64_t* __restrict__ idx, int64_t E, int64_t N,
                                                      unsigned int* __restrict__ cnt) {
  __shared__ long long keys[kDegSlots];
  __shared__ unsigned int vals[kDegSlots];
  for (int h = threadIdx.x; h < kDegSlots; h += 256) { keys[h] = -1; vals[h] = 0u; }
  __syncthreads();
  const int64_t c0 = block_id() * kDegChunk;
  const int64_t c1 = c0 + kDegChunk < E ? c0 + kDegChunk : E;
  for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
    const long long k = idx[i];
    if (k < 0 || k >= N) continue;
    const unsigned int h = ((unsigned int)k * 2654435761u) >> 21;          // 11 bits
    const long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[h]), (unsigned long long)-1LL,
                                     (unsigned long long)k);
    if (prev == -1 || prev == k) atomicAdd(&vals[h], 1u);
    else atomicAdd(&cnt[k], 1u);
  }
  __syncthreads();
  for (int h = threadIdx.x; h < kDegSlots; h += 256)
    if (vals[h]) atomicAdd(&cnt[keys[h]], vals[h]);
}

__global__ __launch_bounds__(256) void k_u32_to_f32(float* __restrict__ d, int64_t n) {
  int64_t i = block_id() * 256 + threadIdx.x;
  if (i < n) d[i] = (float)__float_as_uint(d[i]);
}

__global__ __launch_bounds__(256) void k_degree_ordered(const int64_t* __restrict__ segptr, const int64_t* __restrict__ order,
                                                        const float* __restrict__ w, int64_t N, float* __restrict__ deg) {
  int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= N) return;
  float s = 0.0f;
  for (int64_t k = segptr[i]; k < segptr[i + 1]; ++k) s += w[order[k]];  // serial float32, edge order
  deg[i] = s;
}

__global__ __launch_bounds__(256) void k_inv_sqrt_f32(float* __restrict__ d, int64_t n) {
  int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= n) return;
  float r = 1.0f / sqrtf(d[i]);  // two correctly rounded ops == torch CPU pow_(-0.5)
  if (isinf(r)) r = 0.0f;
  d[i] = r;
}

__global__ __launch_bounds__(256) void k_gcn_coo_apply(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                       const float* __restrict__ w, int64_t E, const float* __restrict__ dis,
                                                       float* __restrict__ out) {
  int64_t i = block_id() * 256 + threadIdx.x;
  if (i >= E) return;
  float wi = w ? w[i] : 1.0f;
  out[i] = __fmul_rn(__fmul_rn(dis[row[i]], wi), dis[col[i]]);
}

static inline int64_t blocks_for(int64_t n, int per) { return (n + per - 1) / per; }

}  // namespace disco

using namespace disco;

extern "C" int disco_abi_version(void) { return DISCO_ABI_VERSION; }
extern "C" int disco_last_hip_error(void) { return g_last_hip_error; }

extern "C" const char* disco_status_string(int status) {
  switch (status) {
    case DISCO_OK: return "ok";
    case DISCO_EINVAL: return "invalid argument";
    case DISCO_EUNSUPPORTED: return "unsupported shape";
    case DISCO_EWORKSPACE: return "workspace too small";
    case DISCO_EHIP: return "HIP runtime error";
    case DISCO_ESCRATCH: return "problem exceeds the built-in scratch of the allocation-free entry point (DISCO_NOALLOC_SCRATCH_BYTES)";
    default: return "unknown status";
  }
}

extern "C" int disco_set_tuning(const char* key, int64_t value) {
  if (!key) return DISCO_EINVAL;
  Tuning& t = tuning();
  if (!strcmp(key, "force_g")) t.force_g = (int)value;
  else if (!strcmp(key, "force_nv")) t.force_nv = (int)value;
  else if (!strcmp(key, "max_vec")) t.max_vec = (int)value;
  else if (!strcmp(key, "sweep_mode")) t.sweep_mode = (int)value;
  else if (!strcmp(key, "win_cols")) t.win_cols = (int)value;
  else if (!strcmp(key, "win_rpg")) t.win_rows_per_group = (int)value;
  else if (!strcmp(key, "sweep_blocks_per_cu")) t.sweep_blocks_per_cu = (int)value;
  else if (!strcmp(key, "addr24")) t.addr24 = (int)value;
  else if (!strcmp(key, "xcd_map")) t.xcd_map = (int)value;
  else if (!strcmp(key, "stream_u")) t.stream_u = (int)value;
  else if (!strcmp(key, "stream_lead")) t.stream_lead = (int)value;
  else if (!strcmp(key, "row_cols")) t.row_cols = (int)value;
  else if (!strcmp(key, "narrow_groups")) t.narrow_groups = (int)value;
  else if (!strcmp(key, "psync_pct")) t.psync_pct = (int)value;
  else if (!strcmp(key, "psync_polls")) t.psync_polls = (int)value;
  else return DISCO_EINVAL;
  return DISCO_OK;
}

extern "C" int disco_csr_rowsum_f64(const int64_t* rowptr, const float* val, int64_t n_rows, int add_identity,
                                    double* rowsum, void* stream) {
  if (n_rows < 0) return DISCO_EINVAL;
  if (n_rows == 0) return DISCO_OK;
  if (!rowptr || !rowsum) return DISCO_EINVAL;
  DISCO_GRID(g_rows, blocks_for(n_rows, 16), 256);
  DISCO_GRID(g_long, blocks_for(n_rows, 1024), 1024);
  hipLaunchKernelGGL(k_csr_rowsum, g_rows, dim3(256), 0, static_cast<hipStream_t>(stream), rowptr,
                     val, n_rows, add_identity, rowsum);
  if (add_identity && val)
    hipLaunchKernelGGL(k_csr_rowsum_long, g_long, dim3(1024), 0, static_cast<hipStream_t>(stream), rowptr,
                       val, n_rows, rowsum);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_gcn_norm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val_in, int64_t n_rows,
                                      int add_identity, double* rowsum, float* val_out, void* stream) {
  if (n_rows < 0) return DISCO_EINVAL;
  if (n_rows == 0) return DISCO_OK;
  if (!rowptr || !col || !rowsum || !val_out) return DISCO_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISCO_GRID(g_n, blocks_for(n_rows, 256), 256);
  hipLaunchKernelGGL(k_inv_sqrt_norm, g_n, dim3(256), 0, s, rowsum, n_rows, add_identity);
  int64_t nnz = 0;   // rowptr lives on the device: the entry count is its last element
  DISCO_HIP_TRY(hipMemcpyAsync(&nnz, rowptr + n_rows, sizeof(nnz), hipMemcpyDeviceToHost, s));
  DISCO_HIP_TRY(hipStreamSynchronize(s));
  if (nnz < 0 || (nnz + kScaleChunk - 1) / kScaleChunk > 0x7fffffffLL) return DISCO_EINVAL;
  DISCO_GRID(g_nnz, blocks_for(nnz, kScaleChunk), 256);
  if (nnz > 0)
    hipLaunchKernelGGL(k_gcn_scale_csr, g_nnz, dim3(256), 0, s, rowptr, col,
                       val_in, n_rows, nnz, add_identity, (const double*)rowsum, val_out);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_degree_count_f32(const int64_t* idx, int64_t E, int64_t N, float* deg, void* stream) {
  if (E < 0 || N < 0) return DISCO_EINVAL;
  if (N == 0) return DISCO_OK;
  if (!deg || (E > 0 && !idx)) return DISCO_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DISCO_HIP_TRY(hipMemsetAsync(deg, 0, sizeof(float) * (size_t)N, s));
  DISCO_GRID(g_e, blocks_for(E, kDegChunk), 256);
  DISCO_GRID(g_n, blocks_for(N, 256), 256);
  if (E > 0) hipLaunchKernelGGL(k_degree_count, g_e, dim3(256), 0, s, idx, E, N, reinterpret_cast<unsigned int*>(deg));
  hipLaunchKernelGGL(k_u32_to_f32, g_n, dim3(256), 0, s, deg, N);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_degree_ordered_f32(const int64_t* segptr, const int64_t* order, const float* w, int64_t N, float* deg,
                                        void* stream) {
  if (N < 0) return DISCO_EINVAL;
  if (N == 0) return DISCO_OK;
  if (!segptr || !order || !w || !deg) return DISCO_EINVAL;
  DISCO_GRID(g_n, blocks_for(N, 256), 256);
  hipLaunchKernelGGL(k_degree_ordered, g_n, dim3(256), 0, static_cast<hipStream_t>(stream), segptr, order,
                     w, N, deg);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_inv_sqrt_f32(float* deg, int64_t N, void* stream) {
  if (N < 0) return DISCO_EINVAL;
  if (N == 0) return DISCO_OK;
  if (!deg) return DISCO_EINVAL;
  DISCO_GRID(g_n, blocks_for(N, 256), 256);
  hipLaunchKernelGGL(k_inv_sqrt_f32, g_n, dim3(256), 0, static_cast<hipStream_t>(stream), deg, N);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_gcn_norm_coo_apply_f32(const int64_t* row, const int64_t* col, const float* w_in, int64_t E,
                                            const float* dis, float* w_out, void* stream) {
  if (E < 0) return DISCO_EINVAL;
  if (E == 0) return DISCO_OK;
  if (!row || !col || !dis || !w_out) return DISCO_EINVAL;
  DISCO_GRID(g_e, blocks_for(E, 256), 256);
  hipLaunchKernelGGL(k_gcn_coo_apply, g_e, dim3(256), 0, static_cast<hipStream_t>(stream), row, col, w_in,
                     E, dis, w_out);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}
