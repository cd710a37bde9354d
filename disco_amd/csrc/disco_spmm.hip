// disco_spmm.hip -- CSR row-parallel SpMM (sum / mean / max) for gfx950 (MI355X, CDNA4).
//
// Replaces torch_sparse.matmul reached from
//   /root/reference/models/convs_large/sg_conv.py:92-93, gcn_conv.py:170-171,
//   sage_conv.py:148-150, gin_conv.py:88-90
// and MessagePassing(aggr="max").propagate at /root/reference/LargeScaleCondensing.py:152-172.
//
// Design:
//   * a GROUP of G lanes (16/32/64) owns one output row; each lane owns NV vectors of VEC floats
//     of that row, so a row gather is NV coalesced 4*VEC*G-byte wave loads and the sum lives in
//     registers (no atomics, CSR order => deterministic);
//   * (col,val) are fetched G entries at a time by one coalesced load and broadcast inside the
//     group (v_readlane -> SGPR addressing when G == 64, ds_bpermute otherwise);
//   * U neighbour rows are in flight per group (U*NV independent 16-byte loads per lane);
//   * rows come in degree-binned order (heaviest first); rows above the hub threshold are cut
//     into pieces, each piece summed by the 8 waves of a workgroup through LDS, and pieces of
//     one row are combined in piece order by a fix-up kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"
#include "disco_spmm_common.h"

namespace disco {

template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24>
__device__ __forceinline__ void consume_batch(int cv, float wv, int n, const float* __restrict__ x, int64_t ldx,
                                              const int64_t (&off)[NV], typename Vec<VEC>::type (&acc)[NV]) {
  typedef typename Vec<VEC>::type V;
  constexpr int U = Unroll<NV>::value < G ? Unroll<NV>::value : G;   // a batch holds at most G entries
  int j = 0;
  for (; j + U <= n; j += U) {
    V xr[U][NV];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int c = bcast_i<G>(cv, j + u);
      w[u] = HASVAL ? bcast_f<G>(wv, j + u) : 1.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) xr[u][v] = gather_vec<VEC, A24>(x, ldx, c, off[v]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int v = 0; v < NV; ++v) vupdate<VEC, OP, HASVAL>(acc[v], w[u], xr[u][v]);
    }
  }
  if (j < n) {
    // remainder (< U entries): one more full-depth batch instead of a serial tail.  Slots past the
    // end re-read the last valid neighbour row (a cache hit) and are dropped by a select, never by
    // a multiply with 0 (0 * inf would poison the sum).
    V xr[U][NV];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ju = j + u < n ? j + u : n - 1;
      int c = bcast_i<G>(cv, ju);
      w[u] = HASVAL ? bcast_f<G>(wv, ju) : 1.0f;
      // slots past the end are switched off by the execution mask: they cost no cache bandwidth
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        xr[u][v] = 0.0f;
        if (j + u < n) xr[u][v] = gather_vec<VEC, A24>(x, ldx, c, off[v]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (j + u < n) {
#pragma unroll
        for (int v = 0; v < NV; ++v) vupdate<VEC, OP, HASVAL>(acc[v], w[u], xr[u][v]);
      }
    }
  }
}

// Accumulate entries [beg, end) of one CSR row into acc[NV].  All lanes of the group call this
// with the same (beg, end); lane `gl` owns vectors gl + G*v.
template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24 = false>
__device__ __forceinline__ void accumulate_range(int64_t beg, int64_t end, const int32_t* __restrict__ col,
                                                 const float* __restrict__ val, const float* __restrict__ x,
                                                 int64_t ldx, int nvec, int gl,
                                                 typename Vec<VEC>::type (&acc)[NV]) {
  int64_t off[NV];
  lane_offsets<VEC, G, NV>(nvec, gl, off);
  for (int64_t base = beg; base < end; base += G) {
    int64_t rem = end - base;
    int n = rem < G ? (int)rem : G;
    int cv = 0;
    float wv = 0.0f;
    if (gl < n) {
      cv = ld_entries(col + base + gl);
      if constexpr (HASVAL) wv = ld_entries(val + base + gl);
    }
    consume_batch<VEC, G, NV, OP, HASVAL, A24>(cv, wv, n, x, ldx, off, acc);
  }
}

// ---- rows below the hub threshold: one group of G lanes per row ---------------------------------
template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24>
__global__ __launch_bounds__(256) void k_spmm_rows(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ val, const float* __restrict__ x,
                                                   int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                   const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                   int flags, int fcols, int64_t xcd_chunk) {
  typedef typename Vec<VEC>::type V;
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gl = lane % G;
  // XCD-aware block order (speed only): workgroups are dealt round-robin over the 8 XCDs, so workgroup b and
  // b + 8 share an L2.  With xcd_chunk > 0 the row order is cut into runs of xcd_chunk blocks and the runs are dealt
  // to the XCDs in turn: the workgroups of one XCD walk a run of CONSECUTIVE rows (rows of a locality-grouped order
  // share neighbours: their gathers hit that XCD's L2) while all eight stay in the same region of the order.
  // linear workgroup id: the launch is 2-D once rows x G reaches 2^32 work-items (disco_internal.h, grid_for)
  int64_t bid = block_id();
  if (xcd_chunk > 0) {
    const uint32_t lb = (uint32_t)bid, q = lb >> 3, c = (uint32_t)xcd_chunk;
    bid = ((int64_t)(q / c) * 8 + (lb & 7)) * c + q % c;
  }
  int64_t grp = (bid * 4 + wave) * RPW + lane / G;
  if (grp >= count) return;
  int64_t row = perm ? (int64_t)perm[first + grp] : first + grp;
  int64_t beg = rowptr[row], end = rowptr[row + 1];
  if constexpr (G == 64) {
    row = uniform64(row);
    beg = uniform64(beg);
    end = uniform64(end);
  }
  V acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) vinit<VEC, OP>(acc[v]);
  accumulate_range<VEC, G, NV, OP, HASVAL, A24>(beg, end, col, val, x, ldx, nvec, gl, acc);
  const bool empty = end == beg;
  const float cnt = (float)(end - beg);
  const bool mean = flags & FLAG_MEAN, accum = flags & FLAG_ACCUM;
  float* yr = y + row * ldy;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int idx = gl + G * v;
    if (idx < nvec) {
      V a = acc[v];
      V* dst = reinterpret_cast<V*>(yr + (int64_t)idx * VEC);
      if constexpr (OP == OP_MAX) { if (empty) a = 0.0f; }
      else {
        if (mean && !empty) a = vdiv<VEC>(a, cnt);
        if (accum) a = *dst + a;  // y += A x: the earlier (interior) part first, then this part
      }
      store_vec<VEC>(reinterpret_cast<float*>(dst), a, idx, nvec, fcols & (VEC - 1));
    }
  }
}

// ---- windowed variant: the same row walk restricted to ONE window of source rows ---------------------
// Launched once per window w (in order): every row adds the entries whose column falls into window w
// (entries wptr[w][pos] .. wptr[w+1][pos], absolute offsets) to its partial sum in y.  All workgroups of one launch
// gather from the same few MB of x (5.5 MiB measured best), so after first touch every XCD serves most gathers from its own L2
// (26.9 TB/s measured) instead of the fabric (7.5 TB/s); the price is one read-modify-write of the y
// block per window.  Windows are visited in ascending column order, i.e. each row still sums its entries
// in CSR order: the only difference to k_spmm_rows is that the running sum passes through memory (no
// rounding: float32 in, float32 out) -- same bits.
// TAIL: the column block's width is not a multiple of VEC, the last vector is stored masked (costs three registers:
// that variant runs at 5 instead of 6 waves per SIMD rather than spilling).  !TAIL: write-through (`sc1`) stores of
// the partial sums through a buffer descriptor on y (DISCO_WIN_YSTORE in disco_spmm_common.h; y block < 4 GiB).
template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24, bool TAIL>
__global__ __launch_bounds__(256, NV == 1 ? (TAIL ? DISCO_WIN_WAVES - 1 : DISCO_WIN_WAVES) : 1) void k_spmm_rows_win(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ x,
                                                       int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                       const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                       const uint32_t* __restrict__ wbeg, const uint32_t* __restrict__ wend,
                                                       int first_window, int last_window, int flags, int rpg,
                                                       int64_t slice_cp, int n_slices, int phase) {
  typedef typename Vec<VEC>::type V;
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gl = lane % G;
  // A group handles up to MAXR consecutive rows (positions) per launch, software-pipelined: all their
  // table entries are fetched first, then the first (col, val) batch and the y block of row i+1 are in
  // flight while row i gathers -- the per-row chain table -> (col, y) -> gather overlaps across rows.
  constexpr int MAXR = DISCO_WIN_MAXR;
  int64_t chunk = blockIdx.x;
  if (phase >= 0) {
    // XCD-sliced skewed schedule (sweep_mode 5).  Workgroups are dealt to the 8 XCDs in turn, so the workgroups with
    // blockIdx % 8 == j share an L2: in phase t they all gather from ONE window ("slice") s_j -- the one of
    // {t-7 .. t} with s_j % 8 == j -- so the eight L2s hold eight DIFFERENT slices (8 x the distinct bytes of a launch
    // in which every XCD caches the same window), and an XCD keeps its slice for 8 consecutive phases.  The row
    // chunks are dealt into 8 groups (chunk % 8); group g meets slice s in phase s + g: every row visits the slices
    // in ascending order (CSR order: same bits as the other kernels), one slice per phase, and in one phase the eight
    // (group, slice) pairs are disjoint in rows, so y has one writer per row and launch.  Phases run as launches on
    // one stream.  The block -> XCD deal is an observation, not a contract: it decides the speed only.
    const int j = blockIdx.x & 7;
    const int s = phase - ((phase - j) & 7);
    if (s < 0 || s >= n_slices) return;
    chunk = (int64_t)(blockIdx.x >> 3) * 8 + (phase - s);
    wbeg += (int64_t)s * slice_cp;
    wend = wbeg + slice_cp;
    first_window = s == 0;
    last_window = s == n_slices - 1;
  }
  const int64_t grp0 = ((chunk * 4 + wave) * RPW + lane / G) * rpg;
  if (grp0 >= count) return;
  int64_t rem_rows = count - grp0;
  const int nrow = rem_rows < rpg ? (int)rem_rows : rpg;
  int64_t off[NV];
  lane_offsets<VEC, G, NV>(nvec, gl, off);
#if defined(DISCO_DBG_NORMW)
  // timing experiment: the partial sums are neither read back nor stored (only a value that never occurs is): the
  // kernel's speed without the read-modify-write of y
  const bool resume = false;
#else
  const bool resume = !first_window || (flags & FLAG_ACCUM);
#endif
  // last window: the mean divides by the row length, the maximum of an empty row becomes 0
  const bool finalize = last_window && (OP == OP_MAX || (flags & FLAG_MEAN));
  uint32_t mb[MAXR], me[MAXR], mrow[MAXR];
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    mb[i] = 0; me[i] = 0; mrow[i] = 0;
    if (i < nrow) { mb[i] = wbeg[grp0 + i]; me[i] = wend[grp0 + i]; mrow[i] = perm[first + grp0 + i]; }
  }
  int cv_n = 0; float wv_n = 0.0f; V y_n[NV];
  auto fetch = [&](uint32_t b, uint32_t e, uint32_t r, int& cv, float& wv, V (&yv)[NV]) {
    const uint32_t len = e - b;
    cv = 0; wv = 0.0f;
    if ((uint32_t)gl < len) {
      cv = ld_entries(col + (int64_t)b + gl);
      if constexpr (HASVAL) wv = ld_entries(val + (int64_t)b + gl);
    }
    const float* yr = y + (int64_t)r * ldy;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      int idx = gl + G * v;
      vinit<VEC, OP>(yv[v]);
      if (resume && idx < nvec && !DISCO_WIN_YLATE) {
        if constexpr (DISCO_WIN_YLOAD != 0) yv[v] = __builtin_nontemporal_load(reinterpret_cast<const V*>(yr + (int64_t)idx * VEC));
        else yv[v] = *reinterpret_cast<const V*>(yr + (int64_t)idx * VEC);
      }
    }
  };
#if DISCO_WIN_HOIST
  // every row's first (col, val) batch and y block are requested BEFORE the first gather: vector-memory loads return
  // in issue order, so a fabric-latency load issued between two rows' gathers would hold back the L2-hit gathers behind it
  int cv_h[MAXR]; float wv_h[MAXR]; V y_h[MAXR][NV];
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    cv_h[i] = 0; wv_h[i] = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) vinit<VEC, OP>(y_h[i][v]);
    if (i < nrow) fetch(mb[i], me[i], mrow[i], cv_h[i], wv_h[i], y_h[i]);
  }
#else
  fetch(mb[0], me[0], mrow[0], cv_n, wv_n, y_n);
#endif
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    if (i < nrow) {
#if DISCO_WIN_HOIST
      int cv = cv_h[i]; float wv = wv_h[i];
      V acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = y_h[i][v];
#else
      int cv = cv_n; float wv = wv_n;
      V acc[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = y_n[v];
      if (i + 1 < MAXR && i + 1 < nrow) fetch(mb[i + 1], me[i + 1], mrow[i + 1], cv_n, wv_n, y_n);
#endif
      int64_t beg = mb[i], end = me[i], row = mrow[i];
      if constexpr (G == 64) { beg = uniform64(beg); end = uniform64(end); row = uniform64(row); }
      // nothing to add and nothing to finalise: leave the row alone
      if (end != beg || first_window || finalize) {
        const int n0 = end - beg < G ? (int)(end - beg) : G;
        consume_batch<VEC, G, NV, OP, HASVAL, A24>(cv, wv, n0, x, ldx, off, acc);
        if (end - beg > G) accumulate_range<VEC, G, NV, OP, HASVAL, A24>(beg + G, end, col, val, x, ldx, nvec, gl, acc);
        float cnt = 0.0f;
        bool empty_row = false;
        if (finalize) {
          const int64_t rn = rowptr[row + 1] - rowptr[row];
          if constexpr (OP == OP_MAX) empty_row = rn == 0; else cnt = (float)rn;
        }
        float* yr = y + row * ldy;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          int idx = gl + G * v;
          if (idx < nvec) {
            V a = acc[v];
            if constexpr (DISCO_WIN_YLATE != 0) {
              // the running sum is read AFTER this window's gathers (vector loads return in issue order: a memory-latency
              // load at the head of a visit holds back the L2-hit gathers behind it) and this window's sum is added to it
              if (resume) vcombine<VEC, OP>(a, *reinterpret_cast<const V*>(yr + (int64_t)idx * VEC));
            }
            if (cnt > 0.0f) a = vdiv<VEC>(a, cnt);
            if (empty_row) a = 0.0f;
#if defined(DISCO_DBG_NORMW)
            if constexpr (VEC == 4) { if (a[0] != 1.2345678e30f) continue; }
#endif
            if constexpr (!TAIL && VEC == 4 && DISCO_WIN_YSTORE == 2) {
              typedef unsigned int u4 __attribute__((ext_vector_type(4)));
              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, 0xffffffff, 0x27000);
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, a), rs, (int)(uint32_t)((row * ldy + (int64_t)idx * VEC) * 4), 0, 16);
            } else if constexpr (!TAIL && DISCO_WIN_YSTORE == 1) {
              __builtin_nontemporal_store(a, reinterpret_cast<V*>(yr + (int64_t)idx * VEC));
            } else {
              store_vec<VEC>(yr + (int64_t)idx * VEC, a, idx, nvec, TAIL ? ((flags >> 8) & (VEC - 1)) : 0);
            }
          }
        }
      }
    }
  }
}

// ---- hub rows: one 512-thread workgroup per piece, partial sums staged in LDS ----------------------
template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24>
__global__ __launch_bounds__(512) void k_spmm_hub(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  const float* __restrict__ val, const float* __restrict__ x,
                                                  int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                  const int32_t* __restrict__ piece_row,
                                                  const int64_t* __restrict__ piece_begin,
                                                  const int64_t* __restrict__ piece_end,
                                                  const int32_t* __restrict__ piece_slot, float* __restrict__ partial,
                                                  int64_t ldp, int flags, int fcols) {
  typedef typename Vec<VEC>::type V;
  constexpr int NG = 512 / G;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  V* ldsv = reinterpret_cast<V*>(lds);
  const int p = blockIdx.x;
  const int64_t pb = piece_begin[p], pe = piece_end[p];
  const int64_t row = piece_row[p];
  const int slot = piece_slot[p];
  const int g = threadIdx.x / G, gl = threadIdx.x % G;
  const int64_t len = pe - pb;
  int64_t chunk = (len + NG - 1) / NG;
  chunk = (chunk + G - 1) / G * G;  // whole cooperative (col,val) fetches per group
  int64_t b = pb + (int64_t)g * chunk;
  b = b < pe ? b : pe;
  int64_t e = b + chunk;
  e = e < pe ? e : pe;
  if constexpr (G == 64) { b = uniform64(b); e = uniform64(e); }
  V acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) vinit<VEC, OP>(acc[v]);
  accumulate_range<VEC, G, NV, OP, HASVAL, A24>(b, e, col, val, x, ldx, nvec, gl, acc);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int idx = gl + G * v;
    if (idx < nvec) ldsv[(int64_t)g * nvec + idx] = acc[v];
  }
  __syncthreads();
  float* dst = slot < 0 ? y + row * ldy : partial + (int64_t)slot * ldp;
  const float cnt = (float)(rowptr[row + 1] - rowptr[row]);
  for (int i = threadIdx.x; i < nvec; i += 512) {
    V s = ldsv[i];
#pragma unroll 4
    for (int k = 1; k < NG; ++k) vcombine<VEC, OP>(s, ldsv[(int64_t)k * nvec + i]);
    V* out = reinterpret_cast<V*>(dst + (int64_t)i * VEC);
    if constexpr (OP == OP_SUM) {
      if ((flags & FLAG_MEAN) && slot < 0) s = vdiv<VEC>(s, cnt);
      if ((flags & FLAG_ACCUM) && slot < 0) s = *out + s;
    }
    // partial rows live in the caller's workspace (ldp is a multiple of 4): whole vectors; y: masked tail
    store_vec<VEC>(reinterpret_cast<float*>(out), s, i, nvec, slot < 0 ? (fcols & (VEC - 1)) : 0);
  }
}

// ---- hub rows cut into several pieces: add the pieces' partial rows in piece order ------------------
template <int OP>
__global__ __launch_bounds__(256) void k_hub_combine(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ multi_row,
                                                     const int64_t* __restrict__ multi_first, const float* __restrict__ partial,
                                                     int64_t ldp, float* __restrict__ y, int64_t ldy, int64_t F, int flags) {
  const int m = blockIdx.x;
  const int64_t row = multi_row[m];
  const int64_t s0 = multi_first[m], s1 = multi_first[m + 1];
  const float cnt = (float)(rowptr[row + 1] - rowptr[row]);
  for (int64_t c = threadIdx.x; c < F; c += 256) {
    float s = partial[s0 * ldp + c];
    for (int64_t k = s0 + 1; k < s1; ++k) {
      float t = partial[k * ldp + c];
      if constexpr (OP == OP_SUM) s = s + t; else s = fmaxf(s, t);
    }
    if constexpr (OP == OP_SUM) {
      if (flags & FLAG_MEAN) s = s / cnt;
      if (flags & FLAG_ACCUM) s = y[row * ldy + c] + s;
    }
    y[row * ldy + c] = s;
  }
}

// ---- L2-blocked sweep: rows below the hub threshold of DENSE-ish graphs -------------------------------
// Measured on MI355X: a row gather that misses the XCD's L2 is served at ~7.5 TB/s
// chip-wide whether it comes from the Infinity Cache or from HBM, a gather that hits L2 at ~27 TB/s.
// This kernel makes the gathers hit L2: the source rows are cut into WINDOWS of `window_rows` rows
// (window_rows * slice bytes ~ 2 MB < 4 MiB L2), every wave owns R = 16 output rows whose sums stay in
// registers, and ALL resident waves walk the windows in the same order at the same time, so that at any
// moment the whole XCD gathers from one window.  Nothing synchronises the waves (correctness does not
// depend on it): they start together, rows of one generation have the same degree class, and a wave
// that runs ahead is the one that misses L2 and falls back.  wptr[w][pos] (uint32) is the absolute offset of
// the first entry of row `perm[first+pos]` with col >= w * window_rows; entries of a row are visited in
// CSR order, so the result has the same bits as k_spmm_rows.
template <bool HASVAL>
__global__ __launch_bounds__(256, 4) void k_spmm_sweep(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ x,
                                                       int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                       const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                       const uint32_t* __restrict__ wptr, int64_t cp, int P,
                                                       int64_t n_batches, int flags, int fcols) {
  constexpr int R = 16;
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t total_waves = (int64_t)gridDim.x * 4;
  const int li = lane < nvec ? lane : nvec - 1;
  const float* xl = x + (int64_t)li * 4;
  for (int64_t batch = wave_global; batch < n_batches; batch += total_waves) {
    // Slot s of batch b is row position b + s * n_batches of the degree-sorted order: every batch gets
    // one row of each of 16 degree quantiles, so all waves carry the same number of entries and walk
    // the windows at the same pace.  Slot info lives in lanes 0..15 (lane s = slot s).
    const int64_t p = batch + (int64_t)(lane & 15) * n_batches;
    const bool valid = p < count;
    const int64_t row = valid ? (int64_t)perm[first + p] : 0;
    const int64_t rbeg = valid ? rowptr[row] : 0;
    uint32_t cur = wptr[p];
    f4 acc[R];
#pragma unroll
    for (int s = 0; s < R; ++s) acc[s] = 0.0f;
    for (int w = 0; w < P; ++w) {
      const uint32_t nxt = wptr[(int64_t)(w + 1) * cp + p];
      const int seglen = (int)(nxt - cur);
      const int64_t eb = (int64_t)cur;
      cur = nxt;
      if (__ballot(seglen > 0 && lane < 16) == 0) continue;
      // cooperative (col, val) fetch: register r covers slots 4r..4r+3, 16 lanes (= 16 entries) per slot
      int cj[4];
      float vj[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int src = 4 * r + (lane >> 4);
        const int k = lane & 15;
        const int sl = __shfl(seglen, src);
        const int64_t se = ((int64_t)__shfl((int)(eb >> 32), src) << 32) | (uint32_t)__shfl((int)(eb & 0xffffffffLL), src);
        const bool ok = k < sl;
        cj[r] = ok ? col[se + k] : 0;
        vj[r] = (HASVAL && ok) ? val[se + k] : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < R; ++s) {
        const int len = __builtin_amdgcn_readlane(seglen, s);
        const int n16 = len < 16 ? len : 16;
        const int lbase = 16 * (s & 3);
        for (int k = 0; k < n16; k += 4) {
          const int m = n16 - k;
          f4 x0, x1, x2, x3;
          float w0 = 1.0f, w1 = 1.0f, w2 = 1.0f, w3 = 1.0f;
          {
            const int c = __builtin_amdgcn_readlane(cj[s >> 2], lbase + k);
            if (HASVAL) w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vj[s >> 2]), lbase + k));
            x0 = *reinterpret_cast<const f4*>(xl + (int64_t)c * ldx);
          }
          if (m > 1) {
            const int c = __builtin_amdgcn_readlane(cj[s >> 2], lbase + k + 1);
            if (HASVAL) w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vj[s >> 2]), lbase + k + 1));
            x1 = *reinterpret_cast<const f4*>(xl + (int64_t)c * ldx);
          }
          if (m > 2) {
            const int c = __builtin_amdgcn_readlane(cj[s >> 2], lbase + k + 2);
            if (HASVAL) w2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vj[s >> 2]), lbase + k + 2));
            x2 = *reinterpret_cast<const f4*>(xl + (int64_t)c * ldx);
          }
          if (m > 3) {
            const int c = __builtin_amdgcn_readlane(cj[s >> 2], lbase + k + 3);
            if (HASVAL) w3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vj[s >> 2]), lbase + k + 3));
            x3 = *reinterpret_cast<const f4*>(xl + (int64_t)c * ldx);
          }
          vupdate<4, OP_SUM, HASVAL>(acc[s], w0, x0);
          if (m > 1) vupdate<4, OP_SUM, HASVAL>(acc[s], w1, x1);
          if (m > 2) vupdate<4, OP_SUM, HASVAL>(acc[s], w2, x2);
          if (m > 3) vupdate<4, OP_SUM, HASVAL>(acc[s], w3, x3);
        }
        if (len > 16) {
          // rare long segment: the remainder goes through the generic walker (same CSR order)
          const int64_t se = ((int64_t)__builtin_amdgcn_readlane((int)(eb >> 32), s) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane((int)(eb & 0xffffffffLL), s);
          f4 a1[1];
          a1[0] = acc[s];
          accumulate_range<4, 64, 1, OP_SUM, HASVAL>(se + 16, se + len, col, val, x, ldx, nvec, lane, a1);
          acc[s] = a1[0];
        }
      }
    }
    const int64_t end_l = valid ? rowptr[row + 1] : 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const int64_t rs = ((int64_t)__builtin_amdgcn_readlane((int)(row >> 32), s) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(row & 0xffffffffLL), s);
      const int64_t nb = ((int64_t)__builtin_amdgcn_readlane((int)(rbeg >> 32), s) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(rbeg & 0xffffffffLL), s);
      const int64_t ne = ((int64_t)__builtin_amdgcn_readlane((int)(end_l >> 32), s) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(end_l & 0xffffffffLL), s);
      if (lane < nvec && batch + (int64_t)s * n_batches < count) {
        f4 a = acc[s];
        f4* dst = reinterpret_cast<f4*>(y + rs * ldy + (int64_t)lane * 4);
        if ((flags & FLAG_MEAN) && ne > nb) a = vdiv<4>(a, (float)(ne - nb));
        if (flags & FLAG_ACCUM) a = *dst + a;
        store_vec<4>(reinterpret_cast<float*>(dst), a, lane, nvec, fcols & 3);
      }
    }
  }
}

// ---- synchronised persistent windows (sweep_mode 3, experimental) --------------------------------------
// Same windows as k_spmm_rows_win, but ONE launch per column block: every lane group keeps R output rows in
// registers and walks the windows itself, so the y block is written once instead of being read-modified-
// written per window.  What the launch boundary did for k_spmm_rows_win -- keeping all waves inside one
// window -- is done by pacing flags: a workgroup that finishes window w publishes the step number and polls
// until `target` workgroups of its sync group (same blockIdx % 8, i.e. the same XCD under round-robin
// dispatch) have done so, giving up after `poll_cap` polls.  The flags only PACE the
// waves: no data depends on them, every wait is bounded, so neither correctness nor termination can depend
// on co-residency or on the dispatch order.  Rows are dealt round-robin over the groups in the degree-sorted
// order, so all groups carry the same mix of row lengths.  Entries are visited in CSR order: same bits.

// One flag word per workgroup (no read-modify-write, no shared hot word): after window step `s` a workgroup
// stores s + 1 into its own flag; wave 0 then polls the flags of its sync group (contiguous in memory)
// until `target` of them have reached s + 1, at most `poll_cap` times.
__device__ __forceinline__ void window_pace(int* flags, int sg, int members, int me, int step, int target, int poll_cap) {
  __syncthreads();                       // all four waves of this workgroup are done with the window
  if (threadIdx.x < 64) {
    int* grp = flags + (int64_t)sg * members;
    if (threadIdx.x == 0) __hip_atomic_store(grp + me, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int polls = 0; polls < poll_cap; ++polls) {
      int done = 0;
      for (int base = 0; base < members; base += 64) {
        const int j = base + (int)threadIdx.x;
        const int f = j < members ? __hip_atomic_load(grp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        done += __popcll(__ballot(j < members && f > step));
      }
      if (done >= target) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
}

template <int VEC, int G, bool HASVAL, bool A24>
__global__ __launch_bounds__(256) void k_spmm_psync(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const float* __restrict__ val, const float* __restrict__ x,
                                                    int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                    const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                    const uint32_t* __restrict__ wptr, int64_t cp, int P, int flags,
                                                    int n_iter, int* __restrict__ prog, int target, int poll_cap, int fcols) {
  typedef typename Vec<VEC>::type V;
  constexpr int R = 8;
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gl = lane % G;
  const int64_t total_groups = (int64_t)gridDim.x * 4 * RPW;
  const int64_t ggrp = ((int64_t)blockIdx.x * 4 + wave) * RPW + lane / G;
  const int sg = blockIdx.x % PSYNC_GROUPS;
  int64_t off[1];
  lane_offsets<VEC, G, 1>(nvec, gl, off);
  for (int it = 0; it < n_iter; ++it) {
    // lane r < R of a group carries the table entries of the group's row slot r
    const int64_t pos = ((int64_t)it * R + (gl < R ? gl : 0)) * total_groups + ggrp;
    const bool valid = gl < R && pos < count;
    uint32_t cur = valid ? wptr[pos] : 0u;
    const uint32_t rowid = valid ? perm[first + pos] : 0u;
    V acc[R][1];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r][0] = 0.0f;
    for (int w = 0; w < P; ++w) {
      const uint32_t nxt = valid ? wptr[(int64_t)(w + 1) * cp + pos] : cur;
      // the first (col, val) batch of all R segments is requested up front: one memory latency per window
      // instead of one per row (prefetching it one window ahead as well costs a wave of occupancy and loses)
      int cvs[R]; float wvs[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t b = (uint32_t)__shfl((int)cur, r, G);
        const uint32_t e = (uint32_t)__shfl((int)nxt, r, G);
        cvs[r] = 0; wvs[r] = 0.0f;
        if ((uint32_t)gl < e - b) {
          cvs[r] = col[(int64_t)b + gl];
          if constexpr (HASVAL) wvs[r] = val[(int64_t)b + gl];
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t b = (uint32_t)__shfl((int)cur, r, G);
        const uint32_t e = (uint32_t)__shfl((int)nxt, r, G);
        if (e != b) {
          const uint32_t len = e - b;
          const int n0 = len < (uint32_t)G ? (int)len : G;
          consume_batch<VEC, G, 1, OP_SUM, HASVAL, A24>(cvs[r], wvs[r], n0, x, ldx, off, acc[r]);
          if (len > (uint32_t)G)
            accumulate_range<VEC, G, 1, OP_SUM, HASVAL, A24>((int64_t)b + G, (int64_t)e, col, val, x, ldx, nvec, gl, acc[r]);
        }
      }
      cur = nxt;
      if (w + 1 < P) window_pace(prog, sg, (int)(gridDim.x / PSYNC_GROUPS), (int)(blockIdx.x / PSYNC_GROUPS), it * P + w, target, poll_cap);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = __shfl((int)valid, r, G) != 0;
      const int64_t row = (int64_t)(uint32_t)__shfl((int)rowid, r, G);
      if (ok && gl < nvec) {
        V a = acc[r][0];
        V* dst = reinterpret_cast<V*>(y + row * ldy + (int64_t)gl * VEC);
        if (flags & FLAG_MEAN) {
          const float cnt = (float)(rowptr[row + 1] - rowptr[row]);
          if (cnt > 0.0f) a = vdiv<VEC>(a, cnt);
        }
        if (flags & FLAG_ACCUM) a = *dst + a;
        store_vec<VEC>(reinterpret_cast<float*>(dst), a, gl, nvec, fcols & (VEC - 1));
      }
    }
  }
}

// wptr[w][pos] for w in [0, P]: lower bound of w * window_rows among the (sorted) columns of row perm[first+pos]
__global__ __launch_bounds__(256) void k_build_wptr(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                    const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                    int64_t cp, int64_t window_rows, int P, uint32_t* __restrict__ wptr,
                                                    unsigned int* __restrict__ unsorted_flag) {
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int w = blockIdx.y;
  if (pos >= cp) return;
  uint32_t out = 0;
  if (pos < count) {
    const int64_t row = perm[first + pos];
    const int64_t b = rowptr[row], e = rowptr[row + 1];
    if (w >= P) out = (uint32_t)e;
    else {
      const int64_t target = (int64_t)w * window_rows;
      int64_t lo = b, hi = e;
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)col[mid] < target) lo = mid + 1; else hi = mid;
      }
      out = (uint32_t)lo;
    }
    if (w == 0) {  // one thread per row also verifies that the row's columns are sorted
      for (int64_t i = b + 1; i < e; ++i)
        if (col[i] < col[i - 1]) { atomicOr(unsorted_flag, 1u); break; }
    }
  }
  wptr[(int64_t)w * cp + pos] = out;
}

// ---- host-side dispatch -----------------------------------------------------------------------------
struct SpmmArgs {
  const int64_t* rowptr; const int32_t* col; const float* val; int64_t n_rows;
  const float* x; int64_t ldx; float* y; int64_t ldy; int64_t F; int reduce;
  const char* sched; const disco_schedule_info* info; float* partial; int64_t ldp; hipStream_t stream;
  bool addr24 = false;   // gather addresses fit v_mad_u32_u24 (set by spmm_dispatch)
};

static int device_cus();

template <int VEC, int G, int NV, int OP, bool HASVAL, bool A24>
static int launch_cfg(const SpmmArgs& a, int nvec) {
  const int mean = (a.reduce == DISCO_REDUCE_MEAN ? FLAG_MEAN : 0) | (a.reduce == DISCO_REDUCE_SUM_ACC ? FLAG_ACCUM : 0);
  constexpr int RPW = 64 / G;
  const uint32_t* perm = nullptr;
  int64_t first = 0, count = a.n_rows;
  if (a.info) {
    const disco_schedule_info& s = *a.info;
    perm = reinterpret_cast<const uint32_t*>(a.sched + s.off_perm);
    first = s.n_hub_rows;
    count = s.n_rows - s.n_hub_rows;
    if (s.n_hub_pieces > 0) {
      size_t lds_bytes = (size_t)(512 / G) * nvec * VEC * sizeof(float);
      DISCO_GRID_1D(hub_grid, s.n_hub_pieces, 512);
      DISCO_GRID_1D(multi_grid, s.n_multi_rows, 256);
      hipLaunchKernelGGL((k_spmm_hub<VEC, G, NV, OP, HASVAL, A24>), hub_grid, dim3(512), lds_bytes, a.stream,
                         a.rowptr, a.col, a.val, a.x, a.ldx, a.y, a.ldy, nvec,
                         reinterpret_cast<const int32_t*>(a.sched + s.off_piece_row),
                         reinterpret_cast<const int64_t*>(a.sched + s.off_piece_begin),
                         reinterpret_cast<const int64_t*>(a.sched + s.off_piece_end),
                         reinterpret_cast<const int32_t*>(a.sched + s.off_piece_slot), a.partial, a.ldp, mean, (int)a.F);
      if (s.n_multi_rows > 0) {
        hipLaunchKernelGGL((k_hub_combine<OP>), multi_grid, dim3(256), 0, a.stream, a.rowptr,
                           reinterpret_cast<const int32_t*>(a.sched + s.off_multi_row),
                           reinterpret_cast<const int64_t*>(a.sched + s.off_multi_first), a.partial, a.ldp, a.y, a.ldy,
                           a.F, mean);
      }
    }
  }
  bool swept = false;
  if constexpr (OP == OP_SUM && VEC == 4 && NV == 1 && G == 64) {
    if (count > 0 && a.info && a.info->sweep_ws && a.info->sweep_windows >= 2 && tuning().sweep_mode == 4) {
      StreamArgs sa{a.rowptr, a.col, a.val, a.x, a.ldx, a.y, a.ldy, nvec, (int)a.F, perm, first, count, a.info, mean, a.addr24, a.stream};
      int rc = launch_stream(sa);
      if (rc == DISCO_OK) swept = true;
      else if (rc != DISCO_EUNSUPPORTED) return rc;      // EUNSUPPORTED: shape outside the streaming kernel, fall through
    }
  }
  if constexpr (OP == OP_SUM && VEC == 4 && NV == 1 && G >= 16) {
    if (!swept && count > 0 && a.info && a.info->sweep_ws && a.info->sweep_windows > 0 && tuning().sweep_mode == 3) {
      const disco_schedule_info& s = *a.info;
      const uint32_t* wptr = reinterpret_cast<const uint32_t*>(s.sweep_ws);
      const int P = (int)s.sweep_windows;
      constexpr int R = 8;
      int per_cu = 0;
      hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_spmm_psync<VEC, G, HASVAL, A24>), 256, 0);
      if (oe != hipSuccess) { set_last_hip_error((int)oe); return DISCO_EHIP; }
      if (tuning().sweep_blocks_per_cu > 0 && tuning().sweep_blocks_per_cu < per_cu) per_cu = tuning().sweep_blocks_per_cu;
      int64_t blocks = (int64_t)device_cus() * (per_cu > 0 ? per_cu : 1);
      const int64_t want = (count + 4 * RPW * R - 1) / (4 * RPW * R);
      if (blocks > want) blocks = want;
      blocks = blocks / PSYNC_GROUPS * PSYNC_GROUPS;
      if (blocks >= PSYNC_GROUPS) {
        const int64_t per_iter = blocks * 4 * RPW * R;
        const int n_iter = (int)((count + per_iter - 1) / per_iter);
        if (blocks * 4 <= PSYNC_BYTES) {
          int* prog = reinterpret_cast<int*>(const_cast<char*>(static_cast<const char*>(s.sweep_ws)) +
                                             (size_t)(P + 1) * s.sweep_count_padded * sizeof(uint32_t));
          hipError_t me = hipMemsetAsync(prog, 0, (size_t)(blocks * 4), a.stream);
          if (me != hipSuccess) { set_last_hip_error((int)me); return DISCO_EHIP; }
          const int pct = tuning().psync_pct > 0 ? tuning().psync_pct : 100;
          const int target = (int)((blocks / PSYNC_GROUPS) * pct / 100);
          const int cap = tuning().psync_polls > 0 ? tuning().psync_polls : 256;
          hipLaunchKernelGGL((k_spmm_psync<VEC, G, HASVAL, A24>), dim3((unsigned)blocks), dim3(256), 0, a.stream, a.rowptr, a.col,
                             a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, wptr, s.sweep_count_padded, P, mean,
                             n_iter, prog, target, cap, (int)a.F);
          swept = true;
        }
      }
    }
  }
  if constexpr ((VEC == 4 || VEC == 2) && G >= 16) {
    // windowed launches (L2-blocked): mean of an accumulating call is not meaningful, so SUM_ACC + MEAN never meet
    int win_rpg = tuning().win_rows_per_group > 0 ? tuning().win_rows_per_group : DISCO_WIN_MAXR;
    if (win_rpg > DISCO_WIN_MAXR) win_rpg = DISCO_WIN_MAXR;
    // the windowed kernel indexes with blockIdx.x alone: a row count whose 1-D launch would wrap takes the plain kernel
    dim3 win_grid;
    const bool win_fits = grid_1d(((count + 4 * RPW * win_rpg - 1) / (4 * RPW * win_rpg) + 7) / 8 * 8, 256, &win_grid);
    if (!swept && win_fits && count > 0 && a.info && a.info->sweep_ws && a.info->sweep_windows > 0 && tuning().sweep_mode >= 2) {
      const disco_schedule_info& s = *a.info;
      const uint32_t* wptr = reinterpret_cast<const uint32_t*>(s.sweep_ws);
      const int P = (int)s.sweep_windows;
      const int rpg = win_rpg;
      int64_t blocks = (count + 4 * RPW * rpg - 1) / (4 * RPW * rpg);
      // the variant without a masked tail stores through a buffer descriptor on y (32-bit offsets, `sc1`): a y block of
      // 4 GiB or more takes the other variant, whose stores are plain and whose tail count may be 0
      const bool tail = (a.F % VEC) != 0 || (uint64_t)s.n_rows * (uint64_t)a.ldy * 4u >= (1ull << 32);
      if (tuning().sweep_mode == 5 && P >= 2) {
        // XCD-sliced skewed schedule: P + 7 phases, in each of them every XCD works on its own slice (kernel comment)
        const int64_t blocks8 = (blocks + 7) / 8 * 8;
        if (blocks8 > 0x7fffffffLL) return DISCO_EUNSUPPORTED;
        for (int t = 0; t < P + 7; ++t) {
          if (tail)
            hipLaunchKernelGGL((k_spmm_rows_win<VEC, G, NV, OP, HASVAL, A24, true>), dim3((unsigned)blocks8), dim3(256), 0, a.stream, a.rowptr, a.col,
                               a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, wptr, wptr, 0, 0, mean | (((int)a.F & 3) << 8), rpg,
                               s.sweep_count_padded, P, t);
          else
            hipLaunchKernelGGL((k_spmm_rows_win<VEC, G, NV, OP, HASVAL, A24, false>), dim3((unsigned)blocks8), dim3(256), 0, a.stream, a.rowptr, a.col,
                               a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, wptr, wptr, 0, 0, mean, rpg,
                               s.sweep_count_padded, P, t);
        }
      } else
      for (int w = 0; w < P; ++w) {
        if (tail)
          hipLaunchKernelGGL((k_spmm_rows_win<VEC, G, NV, OP, HASVAL, A24, true>), dim3((unsigned)blocks), dim3(256), 0, a.stream, a.rowptr, a.col,
                             a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, wptr + (int64_t)w * s.sweep_count_padded,
                             wptr + (int64_t)(w + 1) * s.sweep_count_padded, w == 0, w == P - 1, mean | (((int)a.F & 3) << 8), rpg, (int64_t)0, 0, -1);
        else
          hipLaunchKernelGGL((k_spmm_rows_win<VEC, G, NV, OP, HASVAL, A24, false>), dim3((unsigned)blocks), dim3(256), 0, a.stream, a.rowptr, a.col,
                             a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, wptr + (int64_t)w * s.sweep_count_padded,
                             wptr + (int64_t)(w + 1) * s.sweep_count_padded, w == 0, w == P - 1, mean, rpg, (int64_t)0, 0, -1);
      }
      swept = true;
    }
  }
  if constexpr (VEC == 4 && NV == 1 && OP == OP_SUM) {
    if (!swept && count > 0 && a.info && a.info->sweep_ws && a.info->sweep_windows > 0 && tuning().sweep_mode == 1) {
      const disco_schedule_info& s = *a.info;
      const int64_t n_batches = (count + 15) / 16;
      int64_t blocks = (n_batches + 3) / 4;
      const int64_t cap = (int64_t)device_cus() * (tuning().sweep_blocks_per_cu > 0 ? tuning().sweep_blocks_per_cu : 4);
      if (blocks > cap) blocks = cap;
      hipLaunchKernelGGL((k_spmm_sweep<HASVAL>), dim3((unsigned)blocks), dim3(256), 0, a.stream, a.rowptr, a.col, a.val, a.x,
                         a.ldx, a.y, a.ldy, nvec, perm, first, count, reinterpret_cast<const uint32_t*>(s.sweep_ws),
                         s.sweep_count_padded, (int)s.sweep_windows, n_batches, mean, (int)a.F);
      swept = true;
    }
  }
  if (count > 0 && !swept) {
    int64_t blocks = (count + 4 * RPW - 1) / (4 * RPW);
    int64_t xcd_chunk = 0;
    const int64_t xcd_run = tuning().xcd_map >= 0 ? tuning().xcd_map : (a.info ? a.info->xcd_run : 0);   // in ROWS
    if (xcd_run > 0 && perm && blocks >= 64) {    // 1: one contiguous eighth per XCD; S > 1: runs of S rows
      xcd_chunk = xcd_run == 1 ? (blocks + 7) / 8 : std::min<int64_t>(std::max<int64_t>(1, xcd_run / (4 * RPW)), (blocks + 63) / 64);
      blocks = (blocks + 8 * xcd_chunk - 1) / (8 * xcd_chunk) * (8 * xcd_chunk);
    }
    if (blocks > 0x7fffffffLL) return DISCO_EUNSUPPORTED;
    DISCO_GRID(row_grid, blocks, 256);        // 2-D beyond 2^30 work-items: a 1-D launch of 2^32 wraps around silently
    hipLaunchKernelGGL((k_spmm_rows<VEC, G, NV, OP, HASVAL, A24>), row_grid, dim3(256), 0, a.stream, a.rowptr, a.col,
                       a.val, a.x, a.ldx, a.y, a.ldy, nvec, perm, first, count, mean, (int)a.F, xcd_chunk);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
  return DISCO_OK;
}

template <int VEC, int G, int NV, bool A24>
static int launch_op2(const SpmmArgs& a, int nvec) {
  const bool hv = a.val != nullptr;
  if (a.reduce == DISCO_REDUCE_MAX)
    return hv ? launch_cfg<VEC, G, NV, OP_MAX, true, A24>(a, nvec) : launch_cfg<VEC, G, NV, OP_MAX, false, A24>(a, nvec);
  return hv ? launch_cfg<VEC, G, NV, OP_SUM, true, A24>(a, nvec) : launch_cfg<VEC, G, NV, OP_SUM, false, A24>(a, nvec);
}

template <int VEC, int G, int NV>
static int launch_op(const SpmmArgs& a, int nvec) {
  // 24-bit address arithmetic (gather_vec) for the 128-bit path when the operand allows it
  if constexpr (VEC == 4) {
    if (a.addr24) return launch_op2<VEC, G, NV, true>(a, nvec);
  }
  return launch_op2<VEC, G, NV, false>(a, nvec);
}

template <int VEC>
static int launch_vec(const SpmmArgs& a, int nvec) {
  const Tuning& t = tuning();
  int G, NV;
  if (t.force_g > 0 && t.force_nv > 0 && (int64_t)t.force_g * t.force_nv >= nvec) { G = t.force_g; NV = t.force_nv; }
  else if (VEC == 4 && nvec <= 4 && t.narrow_groups) { G = 4; NV = 1; }     // rows of <= 64 B / <= 128 B: 16 / 8 rows per wave
  else if (VEC == 4 && nvec <= 8 && t.narrow_groups) { G = 8; NV = 1; }     // (column slabs of the feature partition, dist.py)
  else if (nvec <= 16) { G = 16; NV = 1; }
  else if (nvec <= 32) { G = 32; NV = 1; }
  else { G = 64; NV = (nvec + 63) / 64; }
  if constexpr (VEC == 4) {
    if (G == 4 && NV == 1) return launch_op<VEC, 4, 1>(a, nvec);
    if (G == 8 && NV == 1) return launch_op<VEC, 8, 1>(a, nvec);
  }
  if (G == 16 && NV == 1) return launch_op<VEC, 16, 1>(a, nvec);
  if (G == 16 && NV == 2) return launch_op<VEC, 16, 2>(a, nvec);
  if (G == 32 && NV == 1) return launch_op<VEC, 32, 1>(a, nvec);
  if (G == 32 && NV == 2) return launch_op<VEC, 32, 2>(a, nvec);
  if (G == 64 && NV == 1) return launch_op<VEC, 64, 1>(a, nvec);
  if (G == 64 && NV == 2) return launch_op<VEC, 64, 2>(a, nvec);
  if (G == 64 && NV == 3) return launch_op<VEC, 64, 3>(a, nvec);
  if (G == 64 && NV == 4) return launch_op<VEC, 64, 4>(a, nvec);
  if (G == 64 && NV == 5) return launch_op<VEC, 64, 5>(a, nvec);
  return DISCO_EUNSUPPORTED;
}

static int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

static inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Widest vector the operands allow: the leading dimensions and base addresses must keep every
// row VEC-aligned, and a partial last vector must stay inside the row's leading dimension.
static int pick_vec(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* partial, int64_t ldp, int64_t F) {
  const Tuning& t = tuning();
  for (int vec = (t.max_vec > 0 ? t.max_vec : 4); vec >= 2; vec >>= 1) {
    int64_t fr = (F + vec - 1) / vec * vec;
    if (ldx % vec || ldy % vec || ldx < fr || ldy < fr) continue;
    if (!aligned_to(x, 4 * vec) || !aligned_to(y, 4 * vec)) continue;
    if (partial && (ldp % vec || ldp < fr || !aligned_to(partial, 4 * vec))) continue;
    return vec;
  }
  return 1;
}

int spmm_dispatch(const SpmmArgs& a0) {
  SpmmArgs a = a0;
  // Columns are processed in blocks of at most 5*64 vectors (one pass for F <= 1280 at VEC=4).
  int vec = pick_vec(a.x, a.ldx, a.y, a.ldy, a.partial, a.ldp, a.F);
  int64_t max_cols = (int64_t)5 * 64 * vec;
  a.addr24 = tuning().addr24 && a.info && a.info->n_cols > 0 && a.info->n_cols <= (1 << 24) && a.ldx * 4 < (1 << 24) &&
             a.info->n_cols * a.ldx * 4 <= 0xffffffffLL;
  if (vec >= 2 && a.info && a.info->sweep_ws && a.info->sweep_windows > 0 && tuning().sweep_mode) {
    // L2-blocked modes work on balanced column blocks (multiples of 32 floats): <= 256 floats for the
    // persistent sweep kernel (one float4 per lane), `win_cols` (default 128) for the windowed launches
    const int64_t cap = (tuning().sweep_mode == 1 || tuning().sweep_mode == 4) ? 256 : (tuning().win_cols > 0 ? tuning().win_cols : 128);
    if (a.F > cap) {
      int64_t nblk = (a.F + cap - 1) / cap;
      max_cols = ((a.F + nblk - 1) / nblk + 31) / 32 * 32;
    }
  }
  else if (vec >= 2) {
    // Group-major schedules (xcd_run > 0: an XCD walks the rows of one locality group): column blocks narrow enough that
    // the x rows a group shares stay in that XCD's 4 MiB L2 (8192 rows x 128 floats = 4 MiB); `row_cols` overrides.
    const int64_t cap = tuning().row_cols > 0 ? tuning().row_cols : (a.info && a.info->xcd_run > 0 && tuning().row_cols == 0 ? 128 : 0);
    if (cap > 0 && a.F > cap + cap / 4) {
      int64_t nblk = (a.F + cap - 1) / cap;
      max_cols = ((a.F + nblk - 1) / nblk + 31) / 32 * 32;
    }
  }
  for (int64_t c0 = 0; c0 < a0.F; c0 += max_cols) {
    int64_t f = a0.F - c0 < max_cols ? a0.F - c0 : max_cols;
    a.x = a0.x + c0; a.y = a0.y + c0; a.F = f;
    a.partial = a0.partial;  // partial rows are reused per column block (stream order serialises)
    int nvec = (int)((f + vec - 1) / vec);
    int rc;
    if (vec == 4) rc = launch_vec<4>(a, nvec);
    else if (vec == 2) rc = launch_vec<2>(a, nvec);
    else rc = launch_vec<1>(a, nvec);
    if (rc != DISCO_OK) return rc;
  }
  return DISCO_OK;
}

}  // namespace disco

using namespace disco;

extern "C" size_t disco_sweep_workspace_bytes(const disco_schedule_info* info, int64_t n_cols, int64_t window_rows) {
  if (!info || n_cols <= 0 || window_rows <= 0) return 0;
  const int64_t count = info->n_rows - info->n_hub_rows;
  if (count <= 0) return 0;
  const int64_t cp = (count + 15) / 16 * 16;
  const int64_t P = (n_cols + window_rows - 1) / window_rows;
  return (size_t)((P + 1) * cp) * sizeof(uint32_t) + 256 + (size_t)PSYNC_BYTES;   // table + pacing counters (sweep_mode 3)
}

extern "C" int disco_sweep_build(const int64_t* rowptr, const int32_t* col, const void* sched_ws, disco_schedule_info* info,
                                 int64_t n_cols, int64_t window_rows, void* sweep_ws, size_t sweep_ws_bytes, void* stream_) {
  if (!info || !rowptr || !sched_ws || n_cols <= 0 || window_rows <= 0) return DISCO_EINVAL;
  info->sweep_ws = nullptr; info->sweep_windows = 0;
  const int64_t count = info->n_rows - info->n_hub_rows;
  if (count <= 0) return DISCO_OK;
  if (info->nnz >= 0xffffffffLL) return DISCO_EUNSUPPORTED;  // the table holds absolute uint32 entry offsets
  size_t need = disco_sweep_workspace_bytes(info, n_cols, window_rows);
  if (!sweep_ws || sweep_ws_bytes < need) return DISCO_EWORKSPACE;
  if (!col && info->nnz > 0) return DISCO_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t cp = (count + 15) / 16 * 16;
  const int64_t P = (n_cols + window_rows - 1) / window_rows;
  if (P + 1 > 65535) return DISCO_EUNSUPPORTED;
  char* ws = static_cast<char*>(sweep_ws);
  unsigned int* flag = reinterpret_cast<unsigned int*>(ws);
  uint32_t* wptr = reinterpret_cast<uint32_t*>(ws + 256);
  DISCO_HIP_TRY(hipMemsetAsync(flag, 0, 256, stream));
  const uint32_t* perm = reinterpret_cast<const uint32_t*>(static_cast<const char*>(sched_ws) + info->off_perm);
  hipLaunchKernelGGL(k_build_wptr, dim3((unsigned)((cp + 255) / 256), (unsigned)(P + 1)), dim3(256), 0, stream, rowptr, col, perm,
                     info->n_hub_rows, count, cp, window_rows, (int)P, wptr, flag);
  DISCO_CHECK_LAUNCH();
  unsigned int h = 0;
  DISCO_HIP_TRY(hipMemcpyAsync(&h, flag, sizeof(h), hipMemcpyDeviceToHost, stream));
  DISCO_HIP_TRY(hipStreamSynchronize(stream));
  if (h) return DISCO_EUNSUPPORTED;  // columns are not sorted inside the rows: the plain kernel stays in use
  info->sweep_ws = wptr;
  info->sweep_windows = P;
  info->sweep_window_rows = window_rows;
  info->sweep_count_padded = cp;
  info->sweep_n_cols = n_cols;
  return DISCO_OK;
}

extern "C" size_t disco_spmm_workspace_bytes(const disco_schedule_info* info, int64_t F) {
  if (!info || info->n_hub_partials <= 0 || F <= 0) return 0;
  int64_t ldp = (F + 3) / 4 * 4;
  return (size_t)info->n_hub_partials * (size_t)ldp * sizeof(float);
}

extern "C" int disco_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                  const float* x, int64_t ldx, float* y, int64_t ldy, int64_t F, int reduce,
                                  const void* sched_ws, const disco_schedule_info* info, void* spmm_ws,
                                  size_t spmm_ws_bytes, void* stream) {
  if (n_rows < 0 || F < 0 || reduce < 0 || reduce > 3) return DISCO_EINVAL;
  if (n_rows == 0 || F == 0) return DISCO_OK;
  if (!rowptr || !x || !y || ldx < F || ldy < F) return DISCO_EINVAL;
  if (x == y) return DISCO_EINVAL;
  if (info && (!sched_ws || info->n_rows != n_rows)) return DISCO_EINVAL;
  SpmmArgs a;
  a.rowptr = rowptr; a.col = col; a.val = val; a.n_rows = n_rows; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy;
  a.F = F; a.reduce = reduce; a.sched = static_cast<const char*>(sched_ws); a.info = info;
  a.partial = nullptr; a.ldp = 0; a.stream = static_cast<hipStream_t>(stream);
  if (info && info->n_hub_partials > 0) {
    size_t need = disco_spmm_workspace_bytes(info, F);
    if (!spmm_ws || spmm_ws_bytes < need) return DISCO_EWORKSPACE;
    if (!aligned_to(spmm_ws, 16)) return DISCO_EINVAL;
    a.partial = static_cast<float*>(spmm_ws);
    a.ldp = (F + 3) / 4 * 4;
  }
  if (info && info->nnz > 0 && !col) return DISCO_EINVAL;
  return spmm_dispatch(a);
}

// ---- allocation-free entry point ------------------------------------------------------------------------
// Callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid hipMalloc) have no
// way to bring the hub kernels' partial-row workspace along; this variant keeps it in a file-scope device array.
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process ends.
namespace disco {
static __device__ __align__(256) unsigned char scratch[DISCO_NOALLOC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];   // per device: address of that device's copy of `scratch` (nullptr = not looked up)

// Device address of this device's scratch array.  In host code the name `scratch` is only a stand-in for the
// symbol, never an address: it is resolved ONCE per device with that device current, and everything afterwards
// (kernel arguments) uses the returned pointer.
static int scratch_for_current_device(void** out, hipStream_t stream) {
  int dev = -1;
  DISCO_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return DISCO_EUNSUPPORTED;
  if (stream) {
    // the array is per device: a stream of ANOTHER GPU would hand its kernels this GPU's scratch address
    hipDevice_t sdev = -1;
    DISCO_HIP_TRY(hipStreamGetDevice(stream, &sdev));
    if ((int)sdev != dev) return DISCO_EINVAL;
  }
  void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    DISCO_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return DISCO_EHIP;
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);   // racing first calls resolve the same address
  }
  *out = p;
  return DISCO_OK;
}
}  // namespace disco

extern "C" int disco_spmm_csr_f32_noalloc(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                          const float* x, int64_t ldx, float* y, int64_t ldy, int64_t F, int reduce,
                                          const void* sched_ws, const disco_schedule_info* info, void* stream) {
  size_t need = disco_spmm_workspace_bytes(info, F);
  if (need > (size_t)DISCO_NOALLOC_SCRATCH_BYTES) return DISCO_ESCRATCH;
  void* ws = nullptr;
  if (need) {
    int rc = scratch_for_current_device(&ws, static_cast<hipStream_t>(stream));
    if (rc != DISCO_OK) return rc;
  }
  // Same kernels as disco_spmm_csr_f32.  Nothing in the scratch has to start at zero: every piece writes its whole
  // partial row before k_hub_combine reads it, so the previous call's contents are simply overwritten.
  return disco_spmm_csr_f32(rowptr, col, val, n_rows, x, ldx, y, ldy, F, reduce, sched_ws, info, ws, need, stream);
}

extern "C" int disco_khop_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n, int K,
                                  const float* x, int64_t ldx, float* y, int64_t ldy, float* tmp0, float* tmp1,
                                  int64_t ldt, int64_t F, int n_col_slices, const void* sched_ws,
                                  const disco_schedule_info* info, void* spmm_ws, size_t spmm_ws_bytes, void* stream) {
  if (K < 1 || n < 0 || F < 0) return DISCO_EINVAL;
  if (n == 0 || F == 0) return DISCO_OK;
  if (K >= 2 && (!tmp0 || ldt < F)) return DISCO_EINVAL;
  if (K >= 3 && !tmp1) return DISCO_EINVAL;
  if (n_col_slices < 1) n_col_slices = 1;
  // Slice boundaries are multiples of 32 floats (128 B) so every slice keeps the operands' alignment.
  int64_t per = (F + n_col_slices - 1) / n_col_slices;
  per = (per + 31) / 32 * 32;
  for (int64_t c0 = 0; c0 < F; c0 += per) {
    int64_t f = F - c0 < per ? F - c0 : per;
    const float* src = x + c0;
    int64_t lds = ldx;
    for (int k = 0; k < K; ++k) {
      float* dst; int64_t ldd;
      if (k == K - 1) { dst = y + c0; ldd = ldy; }
      else { dst = ((k & 1) ? tmp1 : tmp0) + c0; ldd = ldt; }
      int rc = disco_spmm_csr_f32(rowptr, col, val, n, src, lds, dst, ldd, f, DISCO_REDUCE_SUM, sched_ws, info,
                                  spmm_ws, spmm_ws_bytes, stream);
      if (rc != DISCO_OK) return rc;
      src = dst; lds = ldd;
    }
  }
  return DISCO_OK;
}
