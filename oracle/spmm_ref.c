/*
 * CPU oracle, C part.  TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg); the product never links or loads this file.
 *
 * PARITY STATUS: parity unpinned by the reference (it has no tests or golden
 * vectors; see oracle/disco_oracle.py).  The loops restate the published CPU
 * kernel of torch_sparse 0.6.15 `spmm_{sum,mean,max}` [upstream, restated]:
 * parallel_for over rows, per row a sequential walk of the CSR entries,
 * accumulating `val[e] * mat[col[e]][k]` per feature k in the value dtype.
 * Reference call sites: /root/reference/models/convs_large/sg_conv.py:92-93,
 * /root/reference/models/convs_large/gcn_conv.py:170-171,
 * /root/reference/models/convs_large/sage_conv.py:148-150.
 *
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -ffp-contract=off -shared).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORACLE_SUM = 0, ORACLE_MEAN = 1, ORACLE_MAX = 2 };

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* y[r,:] = reduce_e val[e] * x[col[e],:], float32 accumulation in CSR order. */
void oracle_spmm_csr_f32(const int64_t* rowptr, const int64_t* col, const float* val,
                         int64_t n_rows, const float* x, int64_t ldx, float* y, int64_t ldy,
                         int64_t F, int reduce) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rows; ++r) {
    float* out = y + r * ldy;
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    if (reduce == ORACLE_MAX) {
      if (e == b) { for (int64_t k = 0; k < F; ++k) out[k] = 0.0f; continue; }
      for (int64_t k = 0; k < F; ++k) out[k] = -INFINITY;
      for (int64_t i = b; i < e; ++i) {
        const float* xr = x + col[i] * ldx;
        const float w = val ? val[i] : 1.0f;
        for (int64_t k = 0; k < F; ++k) { float m = w * xr[k]; if (m > out[k]) out[k] = m; }
      }
      continue;
    }
    for (int64_t k = 0; k < F; ++k) out[k] = 0.0f;
    for (int64_t i = b; i < e; ++i) {
      const float* xr = x + col[i] * ldx;
      const float w = val ? val[i] : 1.0f;
      for (int64_t k = 0; k < F; ++k) out[k] += w * xr[k];
    }
    if (reduce == ORACLE_MEAN) {
      const float cnt = (float)((e - b) > 0 ? (e - b) : 1);
      for (int64_t k = 0; k < F; ++k) out[k] /= cnt;
    }
  }
}

/* Same, double accumulation of float32 inputs: ground truth for tolerances. */
void oracle_spmm_csr_f64acc(const int64_t* rowptr, const int64_t* col, const float* val,
                            int64_t n_rows, const float* x, int64_t ldx, double* y, int64_t ldy,
                            int64_t F, double* absbound) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < n_rows; ++r) {
    double* out = y + r * ldy;
    double* ab = absbound ? absbound + r * ldy : 0;
    for (int64_t k = 0; k < F; ++k) out[k] = 0.0;
    if (ab) for (int64_t k = 0; k < F; ++k) ab[k] = 0.0;
    for (int64_t i = rowptr[r]; i < rowptr[r + 1]; ++i) {
      const float* xr = x + col[i] * ldx;
      const double w = val ? (double)val[i] : 1.0;
      for (int64_t k = 0; k < F; ++k) {
        const double p = w * (double)xr[k];
        out[k] += p;
        if (ab) ab[k] += fabs(p);
      }
    }
  }
}

/* K-hop driver: x <- A x, K times, with two scratch buffers the caller owns. */
void oracle_khop_csr_f32(const int64_t* rowptr, const int64_t* col, const float* val,
                         int64_t n, const float* x, float* y, float* tmp, int64_t F, int K) {
  const float* src = x;
  for (int k = 0; k < K; ++k) {
    float* dst = ((K - 1 - k) % 2 == 0) ? y : tmp;
    oracle_spmm_csr_f32(rowptr, col, val, n, src, F, dst, F, F, ORACLE_SUM);
    src = dst;
  }
}

/* gcn_norm edge-index branch (/root/reference/utils.py:242-247), float32, serial
 * scatter in edge order like torch_scatter's CPU scatter_add. */
void oracle_gcn_norm_coo_f32(const int64_t* row, const int64_t* col, const float* w_in,
                             int64_t E, int64_t N, int flow_target_to_source, float* deg_ws,
                             float* w_out) {
  const int64_t* idx = flow_target_to_source ? row : col;
  memset(deg_ws, 0, sizeof(float) * (size_t)N);
  for (int64_t e = 0; e < E; ++e) deg_ws[idx[e]] += w_in ? w_in[e] : 1.0f;
  for (int64_t i = 0; i < N; ++i) {
    float d = 1.0f / sqrtf(deg_ws[i]);
    deg_ws[i] = isinf(d) ? 0.0f : d;
  }
#pragma omp parallel for
  for (int64_t e = 0; e < E; ++e)
    w_out[e] = (deg_ws[row[e]] * (w_in ? w_in[e] : 1.0f)) * deg_ws[col[e]];
}
