/*
 * disco_hip.h -- C ABI of libdisco_hip.so, the MI355X (gfx950) implementation of DisCo's
 * GCN-normalised sparse propagation hot path.
 *
 * The reference (BangHonor/DisCo) is pure Python and has no FFI of its own: the arithmetic of
 * this path lives in torch_sparse / torch_scatter / torch_geometric wheels.  Each entry point
 * below therefore cites the *reference call site* whose third-party kernel it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     on it and nothing here synchronises, except disco_schedule_build (documented there);
 *   - no entry point allocates or frees device memory: callers pass outputs and a workspace
 *     whose size they query first;
 *   - return value: 0 = ok, <0 = error (see disco_status_string); no exceptions cross the ABI;
 *   - CSR: rowptr int64[n_rows+1], col int32[nnz], val float32[nnz] (val may be NULL = all 1);
 *   - dense operands are row-major float32 with an explicit leading dimension in elements;
 *   - sizes: every count is int64.  Launches whose size grows with rows, entries or edges are shaped so that no
 *     dimension reaches 2^32 work-items (a 1-D launch of that size would silently run only part of its workgroups
 *     on this stack); what cannot be shaped that way is refused with DISCO_EUNSUPPORTED, never truncated.  Hard
 *     limits: col is int32 (n_cols < 2^31), schedules take at most 2^31-1 rows, a window table needs nnz < 2^32.
 */
#ifndef DISCO_HIP_H
#define DISCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISCO_ABI_VERSION 2

enum disco_status {
  DISCO_OK = 0,
  DISCO_EINVAL = -1,       /* bad argument (null pointer, negative size, misaligned ld) */
  DISCO_EUNSUPPORTED = -2, /* shape outside what the kernels are instantiated for */
  DISCO_EWORKSPACE = -3,   /* workspace too small */
  DISCO_EHIP = -4,         /* a HIP runtime call failed; see disco_last_hip_error() */
  DISCO_ESCRATCH = -5      /* problem needs more than the built-in scratch of the allocation-free entry point */
};

enum disco_reduce {
  DISCO_REDUCE_SUM = 0,
  DISCO_REDUCE_MEAN = 1,
  DISCO_REDUCE_MAX = 2,
  DISCO_REDUCE_SUM_ACC = 3 /* y += A x: used for the halo block of a row-partitioned graph */
};

/* Host-side description of a degree-binned schedule living in a device workspace. */
typedef struct disco_schedule_info {
  int64_t n_rows;
  int64_t nnz;
  int64_t hub_threshold;   /* rows with more entries than this go to the hub kernel */
  int64_t hub_piece;       /* max entries one hub workgroup handles */
  int64_t n_hub_rows;
  int64_t n_hub_pieces;
  int64_t n_hub_partials;  /* pieces that write a partial row (hub rows with >1 piece) */
  int64_t n_multi_rows;    /* hub rows with >1 piece (combined by the fix-up kernel) */
  int64_t max_row_nnz;
  /* byte offsets into the schedule workspace */
  int64_t off_perm, off_piece_row, off_piece_begin, off_piece_end, off_piece_slot;
  int64_t off_multi_row, off_multi_first;
  int64_t ws_bytes;
  /* optional L2-blocked sweep structure, filled by disco_sweep_build (0 / NULL when absent) */
  int64_t sweep_windows, sweep_window_rows, sweep_count_padded, sweep_n_cols;
  const void* sweep_ws;    /* DEVICE pointer kept inside this host struct */
  /* Optional hint set by the CALLER after disco_schedule_build (which zeroes it): number of rows of x,
   * i.e. every col[] entry is < n_cols.  When n_cols <= 2^24, the row pitch of x is < 2^24 bytes and
   * n_cols * pitch < 4 GiB, the gathers use 24-bit multiply / 32-bit offset addressing (one VALU op per
   * gather instead of a 64-bit multiply).  0 = unknown: 64-bit addressing. */
  int64_t n_cols;
  /* XCD-aware workgroup order of the row kernel (speed only, never results).  Workgroups are dealt to the 8 XCDs in
   * turn; with xcd_run = S > 0 the row order is cut into runs of S ROWS and XCD j takes runs j, j + 8, ...: the
   * workgroups that share an L2 walk CONSECUTIVE rows.  Best when a run is about one locality group (Reddit-shaped
   * communities of 4096 rows: 37 ms per step at S = 4096, 48 ms at S = 1024).  disco_schedule_build sets 0 (rows in
   * degree order share nothing), disco_schedule_build_grouped sets 1024; a caller that knows its group sizes sets the
   * typical size.  disco_set_tuning("xcd_map", v >= 0) overrides it for every call (experiments; 1 = one eighth per XCD). */
  int64_t xcd_run;
} disco_schedule_info;

int disco_abi_version(void);
const char* disco_status_string(int status);
int disco_last_hip_error(void);

/* ---- schedule -------------------------------------------------------------------------
 * Degree-binned row schedule for a CSR structure (north_star: "one wavefront (or
 * sub-wavefront) per row chosen by a degree-binned schedule ... LDS staging of hub-node rows").
 * No counterpart in the reference: torch_sparse's spmm launches one warp per row in natural
 * order (/root/reference/models/convs_large/sg_conv.py:92-93 -> torch_sparse.matmul).
 * disco_schedule_build SYNCHRONISES `stream` once (it reads three counters back) -- call it
 * once per graph, outside any capture.  hub_threshold/hub_piece <= 0 select the defaults. */
size_t disco_schedule_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t hub_threshold,
                                      int64_t hub_piece);
int disco_schedule_build(const int64_t* rowptr, int64_t n_rows, int64_t nnz,
                         int64_t hub_threshold, int64_t hub_piece, void* sched_ws,
                         size_t sched_ws_bytes, disco_schedule_info* info_host, void* stream);

/* Locality-grouped variant: `row_group` (uint32 per row, device) is the major sort key of the row order -- rows of one
 * group are processed together, heavier rows first inside a group; hub rows are split as before, and info->xcd_run
 * is set so that the workgroups of one XCD walk consecutive rows: the rows of a group gather through ONE L2.
 * Only the order of processing changes: results are bit-identical.
 * disco_locality_labels computes such groups on the device: `sweeps` rounds of label propagation over the graph
 * (square CSR; every node takes the most frequent label among its neighbours and itself, ties to the label with the
 * smaller hash; a row tallies up to 1024 distinct labels, rows above 4096 entries keep theirs).  max_group = 0: plain propagation, deterministic (on graphs with hubs one label tends to swallow most
 * rows); max_group > 0: no group grows beyond max_group rows (choose it so that a group's rows of x fit one XCD's
 * 4 MiB L2) -- which rows take the last places of a group is then not reproducible from run to run.
 * labels uint32[n_rows]; workspace from disco_locality_workspace_bytes.  Nothing synchronises.
 * No counterpart in the reference (pymetis is pinned in /root/reference/requirements.txt:95 and never used). */
size_t disco_schedule_grouped_workspace_bytes(int64_t n_rows, int64_t nnz, int64_t hub_threshold,
                                              int64_t hub_piece);
int disco_schedule_build_grouped(const int64_t* rowptr, int64_t n_rows, int64_t nnz, int64_t hub_threshold,
                                 int64_t hub_piece, const uint32_t* row_group, void* sched_ws,
                                 size_t sched_ws_bytes, disco_schedule_info* info_host, void* stream);
size_t disco_locality_workspace_bytes(int64_t n_rows);
int disco_locality_labels(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int sweeps, int64_t max_group,
                          uint32_t* labels, void* workspace, size_t workspace_bytes, void* stream);

/* ---- L2 window table (optional, for graphs whose rows re-use source rows heavily) ------------------
 * Cuts the source rows (columns) into windows of `window_rows` rows sized for the XCD's 4 MiB L2 and records,
 * per window and row of the schedule, where the row's entries of that window start (absolute uint32 offsets,
 * hence nnz < 2^32).  Requires column indices sorted inside every row; returns DISCO_EUNSUPPORTED (and leaves
 * the schedule without a table) when they are not.  Once attached to `info_host`, disco_spmm_csr_f32 /
 * disco_khop_csr_f32 run 64/128-bit-aligned operands as one launch per (column block, window) -- all
 * workgroups of a launch gather from the same few MB of x (Reddit: 1.9x).  Rows below the hub
 * threshold get the same bits as without the table (sum, mean: same CSR order; max: order independent).
 * The caller keeps `sweep_ws` alive as long as `info_host` is used (the kernels also use the last 1 MiB of it
 * for pacing flags in the experimental sweep_mode 3).  SYNCHRONISES `stream` once.
 * No counterpart in the reference. */
size_t disco_sweep_workspace_bytes(const disco_schedule_info* info_host, int64_t n_cols,
                                   int64_t window_rows);
int disco_sweep_build(const int64_t* rowptr, const int32_t* col, const void* sched_ws,
                      disco_schedule_info* info_host, int64_t n_cols, int64_t window_rows,
                      void* sweep_ws, size_t sweep_ws_bytes, void* stream);

/* ---- SpMM -------------------------------------------------------------------------------
 * y[r, 0:F] = reduce_{e in row r} val[e] * x[col[e], 0:F]
 * Replaces torch_sparse.matmul(adj_t, x, reduce) reached from
 *   /root/reference/models/convs_large/sg_conv.py:92-93   (SGConv.message_and_aggregate)
 *   /root/reference/models/convs_large/gcn_conv.py:170-171 (GCNConv.message_and_aggregate)
 *   /root/reference/models/convs_large/sage_conv.py:148-150, gin_conv.py:88-90 (reduce=mean)
 * and, with reduce=MAX and val=NULL, MessagePassing(aggr="max").propagate at
 *   /root/reference/LargeScaleCondensing.py:152-153,172 (rows without entries give 0).
 * `sched_ws`/`info` may be NULL: rows are then walked in natural order and no row is split.
 * x rows are addressed by col[], so x needs >= max(col)+1 rows; x and y must not alias.
 * Deterministic: no atomics; per row the entries are accumulated in CSR order (hub rows: per
 * group in CSR order, groups and pieces combined in a fixed order).
 * Only columns [0, F) of y are written (a partial last vector is stored element by element), so y may be a
 * column view of a wider matrix; columns [F, ldx) of x may be READ when ldx allows wider loads. */
size_t disco_spmm_workspace_bytes(const disco_schedule_info* info_host, int64_t F);
int disco_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                       int64_t n_rows, const float* x, int64_t ldx, float* y, int64_t ldy,
                       int64_t F, int reduce, const void* sched_ws,
                       const disco_schedule_info* info_host, void* spmm_ws, size_t spmm_ws_bytes,
                       void* stream);

/* Allocation-free variant of disco_spmm_csr_f32.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc -- and
 * therefore cannot bring the hub kernels' partial-row workspace (`spmm_ws`).  That workspace lives in a file-scope
 * device array inside the library instead: DISCO_NOALLOC_SCRATCH_BYTES (16 MiB) per GPU the library runs on, held
 * until the process ends.  Same arguments as disco_spmm_csr_f32 minus the workspace, same kernels, same bits.
 *   - returns DISCO_ESCRATCH when disco_spmm_workspace_bytes(info_host, F) exceeds DISCO_NOALLOC_SCRATCH_BYTES
 *     (n_hub_partials * roundup(F, 4) * 4 bytes; e.g. 6898 partial rows at F = 602) -- nothing is launched then;
 *   - calls no hipMalloc / hipFree / allocating routine.  The first call on each device resolves the array's
 *     device address (hipGetSymbolAddress): make that call once outside any capture;
 *   - the CURRENT device must be the device `stream` belongs to (the array is per device; DISCO_EINVAL otherwise);
 *   - graph capture: tests/test_zz_no_alloc.py captures a K-hop of these calls in a HIP graph and replays it;
 *   - all calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams) race on the
 *     hub rows' partial sums.  Calls on one stream, or on different GPUs, are safe. */
#define DISCO_NOALLOC_SCRATCH_BYTES (16u << 20)
int disco_spmm_csr_f32_noalloc(const int64_t* rowptr, const int32_t* col, const float* val,
                               int64_t n_rows, const float* x, int64_t ldx, float* y, int64_t ldy,
                               int64_t F, int reduce, const void* sched_ws,
                               const disco_schedule_info* info_host, void* stream);

/* ---- K-hop propagation ------------------------------------------------------------------
 * y = A^K x for a square CSR A: `for k in range(K): x = propagate(adj_t, x=x)`
 * (/root/reference/models/convs_large/sg_conv.py:74-77).  tmp0/tmp1 are caller-owned
 * [n_rows, ldt] ping-pong buffers (tmp0 needed for K>=2, tmp1 for K>=3).
 * n_col_slices >= 1 runs the K hops independently on column slices of x (kept for callers that want
 * a smaller working set; on MI355X one slice is fastest).  Rows below the hub threshold give
 * the same bits for any value; hub rows differ by rounding (their partial sums follow the lane-group width). */
int disco_khop_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                       int K, const float* x, int64_t ldx, float* y, int64_t ldy, float* tmp0,
                       float* tmp1, int64_t ldt, int64_t F, int n_col_slices,
                       const void* sched_ws, const disco_schedule_info* info_host, void* spmm_ws,
                       size_t spmm_ws_bytes, void* stream);

/* ---- normalisation ----------------------------------------------------------------------
 * normalize_adj (/root/reference/utils.py:406-435) on a CSR whose duplicates are already summed.
 * scipy's dtype promotion decides the arithmetic, and the kernels follow it:
 *   add_identity != 0  (node 0 has no self-loop, utils.py:425-426): `mx + sp.eye(n)` is float64, so
 *       row sums, r = rowsum^-1/2 and (r[row]*v)*r[col] are float64 with one rounding to float32
 *       (utils.py:634).  The 1 on the diagonal is added on the fly: the CSR must hold a (possibly
 *       zero-valued) diagonal entry in every row.
 *   add_identity == 0  (node 0 has a self-loop): the float32 matrix stays float32 throughout: row
 *       sums are sequential float32 additions in CSR order, r is rounded to float32, the products
 *       are float32, left to right.  (numpy's float32 pow is only accurate to a few ulp and
 *       CPU-dependent; r here is the correctly rounded value.)
 * disco_csr_rowsum_f64 writes the row sums (as doubles in both modes; val NULL = row counts). */
int disco_csr_rowsum_f64(const int64_t* rowptr, const float* val, int64_t n_rows, int add_identity,
                         double* rowsum, void* stream);
/* r = rowsum^-1/2 with inf -> 0 (utils.py:428-429), then val_out[e] = r[row] * val_in[e] * r[col[e]]
 * (utils.py:430-432) in the arithmetic described above.  `rowsum` is overwritten with r.
 * val_in NULL = all 1; val_out may alias val_in.  Entry-parallel (independent of row lengths); reads
 * rowptr[n_rows] back to size the launch, i.e. synchronises the stream once (a set-up call). */
int disco_gcn_norm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val_in,
                           int64_t n_rows, int add_identity, double* rowsum, float* val_out,
                           void* stream);

/* gcn_norm, edge-index branch (/root/reference/utils.py:242-247; copy at
 * /root/reference/models/convs_large/gcn_conv.py:65-70).
 * Unweighted degree: deg[i] = #edges with idx == i, exact integers, then deg -> float32; ids outside
 * [0, N) are ignored.  Replaces torch_scatter.scatter_add at utils.py:244 for edge_weight=None. */
int disco_degree_count_f32(const int64_t* idx, int64_t E, int64_t N, float* deg, void* stream);
/* The same counts by sorting instead of atomics (32-bit copies of the ids, radix sort over log2(N) bits, run lengths):
 * random ids make one global atomic per edge the bound of the version above (52 M edges: 2.0 ms); this one moves every id
 * four times, coalesced.  Same result, bit for bit (integers).  Workspace from the query; nothing synchronises. */
size_t disco_degree_count_sorted_workspace_bytes(int64_t E, int64_t N);
int disco_degree_count_sorted_f32(const int64_t* idx, int64_t E, int64_t N, float* deg, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* Weighted degree with the CPU reference's summation order: `order` is a stable argsort of
 * idx and `segptr` int64[N+1] its segment pointers; deg[i] = sequential float32 sum of
 * w[order[segptr[i]..segptr[i+1])] -- bit-identical to a serial scatter_add in edge order. */
int disco_degree_ordered_f32(const int64_t* segptr, const int64_t* order, const float* w,
                             int64_t N, float* deg, void* stream);
/* dis = deg^-1/2 (inf -> 0) in place (utils.py:245-246), float32, 1/sqrt correctly rounded. */
int disco_inv_sqrt_f32(float* deg, int64_t N, void* stream);
/* w_out[e] = (dis[row[e]] * w_in[e]) * dis[col[e]], left to right (utils.py:247).
 * w_in NULL = all 1. */
int disco_gcn_norm_coo_apply_f32(const int64_t* row, const int64_t* col, const float* w_in,
                                 int64_t E, const float* dis, float* w_out, void* stream);

/* ---- pair scorer (PGE_Edge over all pairs of condensed nodes) -------------------------------
 * Replaces the loop `for i in range(n): adj_syn[i] = pge_edge.inference(cat([Z[[i]*n], Z], 1))`
 * (/root/reference/LargeScaleCondensing.py:304-305) for the 3-layer model of
 * /root/reference/models/parametrized_adj_lp.py:13-38 in eval mode (BatchNorm running statistics).
 * prepare: folds BatchNorm into the layers and computes the two n x H first-layer factors
 *          (W1 is [H, 2*D] row-major as nn.Linear stores it; H in {32, 64, 128, 256}).
 * block  : out[i - i0, j - j0] = score([z_i || z_j]) for i in [i0, i1), j in [j0, j1);
 *          apply_sigmoid = 0 returns the logit.  Layer 2 runs on the exact-fp32 MFMA. */
size_t disco_pairscore_workspace_bytes(int64_t n, int64_t H);
int disco_pairscore_prepare_f32(const float* Z, int64_t n, int64_t D, int64_t ldz, int64_t H,
                                const float* W1, const float* b1, const float* bn1_weight,
                                const float* bn1_bias, const float* bn1_mean, const float* bn1_var,
                                const float* W2, const float* b2, const float* bn2_weight,
                                const float* bn2_bias, const float* bn2_mean, const float* bn2_var,
                                const float* W3, const float* b3, float eps, void* ws, size_t ws_bytes,
                                void* stream);
int disco_pairscore_block_f32(const void* ws, int64_t n, int64_t H, int64_t i0, int64_t i1, int64_t j0,
                              int64_t j1, float* out, int64_t ldo, int apply_sigmoid, void* stream);

/* Fused edge construction: scores BOTH directions of every pair i < j (the same n^2 pair evaluations as the full
 * matrix), forms a = (S[i][j] + S[j][i]) / 2 and, where !(a < threshold) and a != 0, appends (i, j, a) and (j, i, a)
 * to an UNORDERED list -- the n x n matrix of /root/reference/LargeScaleCondensing.py:291,304-307 is never
 * materialised.  *count_dev (device int64, zeroed by the call) receives the number of entries produced; entries
 * beyond `capacity` are counted but not written, so count > capacity means: call again with room for `count`.
 * The diagonal is NOT emitted (variants differ: the caller adds the n unit entries it wants), and the list is in
 * no particular order: sort it by (row, col) -- keys are unique -- for the torch.nonzero order of the dense path
 * (disco_coo_to_csr with DISCO_COO_DUP_KEEP does that).  Bitwise the same values as the dense path. */
int disco_pairscore_edges_f32(const void* ws, int64_t n, int64_t H, float threshold, int64_t* edge_row,
                              int64_t* edge_col, float* edge_val, int64_t capacity, int64_t* count_dev, void* stream);

/* Post-processing of the n x n score matrix S into a COO edge list in torch.nonzero (row-major) order.
 * A = (S + S^T)/2, then
 *   variant 0 (transductive, LargeScaleCondensing.py:306-308):  A[A<thr]=0; diag=1
 *   variant 1 (inductive, LargeScaleCondensing_induct.py:300-302): diag=1; A[A<thr]=0
 *   variant 2 (sampled, LargeScaleCondensing_Sampled.py:271-284): off-diagonal part only (the caller
 *             appends the n unit self-loops, as add_self_loops does at :285).
 * count writes the number of kept entries per row; the caller turns it into exclusive row starts
 * (and the total E) and calls fill with edge_index int64 [2, E] row-major and edge_weight [E]. */
int disco_sym_threshold_count(const float* S, int64_t n, int64_t lds, float threshold, int variant,
                              int64_t* rowcount, void* stream);
int disco_sym_threshold_fill(const float* S, int64_t n, int64_t lds, float threshold, int variant,
                             const int64_t* rowstart, int64_t E, int64_t* edge_index,
                             float* edge_weight, void* stream);

/* Benchmark / test knobs: "force_g", "force_nv", "max_vec" (lane-group shape), "sweep_mode" (with a window table
 * attached: 0 ignore it, 1 persistent sweep, 2 windowed launches [default], 3 paced persistent windows, 4 streaming
 * persistent windows [experimental, DESIGN.md 3.4], 5 XCD-sliced skewed schedule of the windowed launches [same bits as 2;
 * measured slower, DESIGN.md 3.4b]), "stream_u" (mode 4: gathers in flight per wave), "xcd_map"
 * (k_spmm_rows: -1 [default] = info->xcd_run decides, 0 = off, 1 = one contiguous eighth of the row order per XCD,
 * S > 1 = runs of S rows), "row_cols" (column block of the plain row kernel: 0 = 128 floats under a group-major
 * schedule and one pass otherwise, -1 = never block, > 0 = that width), "narrow_groups" (0 = no 4- / 8-lane row groups),
 * "win_cols" (column block of modes 2/3, default 128), "win_rpg" (rows per lane group and launch, 1 or 2),
 * "sweep_blocks_per_cu", "psync_pct", "psync_polls" (mode 3 pacing), "stream_lead" (mode 4: window steps a wave may run
 * ahead of its XCD group; < 0 = no pacing), "addr24" (0 = always 64-bit gather
 * addresses).  Unknown keys return DISCO_EINVAL. */
int disco_set_tuning(const char* key, int64_t value);

/* ---- CSR construction from raw COO edges ---------------------------------------------
 * Replaces the single-threaded host set-up of the reference: scipy `csr_matrix((ones,(row,col)))` with
 * its duplicate summation (/root/reference/utils.py:104-106), PyG `to_undirected(edge_index)`
 * (/root/reference/utils.py:99-100) and the row-major ordering of `SparseTensor(row=,col=,value=).t()`
 * (/root/reference/LargeScaleCondensing.py:486).
 *   flags = [DISCO_COO_SYMMETRISE] | one of DISCO_COO_DUP_{KEEP,SUM,FIRST}
 *     SYMMETRISE : (col,row) is added for every (row,col) before sorting (square shapes only)
 *     DUP_KEEP   : duplicates stay, in input order (stable sort)       -- torch_sparse.SparseTensor
 *     DUP_SUM    : duplicates folded, values summed in float64 in input order, rounded once -- csr_matrix
 *     DUP_FIRST  : duplicates folded, first value in input order kept  -- to_undirected / coalesce of unit weights
 *   val may be NULL (unit weights); val_out may be NULL when only the structure is wanted (val must then
 *   be NULL).  rowptr[n_rows+1], col_out / val_out need room for n_edges (2 n_edges with SYMMETRISE)
 *   entries; *nnz_out (host) receives the number written.  Columns come out sorted inside every row.
 *   Synchronises the stream once (the entry count is read back).  DISCO_EINVAL if an index lies
 *   outside the shape. */
#define DISCO_COO_SYMMETRISE 1
#define DISCO_COO_DUP_KEEP 0
#define DISCO_COO_DUP_SUM 2
#define DISCO_COO_DUP_FIRST 4
#define DISCO_COO_DUP_MASK 6
size_t disco_coo_to_csr_workspace_bytes(int64_t n_edges, int flags, int with_values);
int disco_coo_to_csr(const int64_t* row, const int64_t* col, const float* val, int64_t n_edges, int64_t n_rows,
                     int64_t n_cols, int flags, int64_t* rowptr, int32_t* col_out, float* val_out, int64_t* nnz_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Structure of `mx + I` (/root/reference/utils.py:425-426) on a CSR with sorted columns, without a second
 * sort: rows that lack a diagonal entry get a ZERO-valued one at its sorted position (existing diagonal
 * entries and all other entries are copied unchanged; disco_csr_rowsum_f64 / disco_gcn_norm_csr_f32 add the 1
 * in float64).  rowptr_out[n_rows+1]; col_out / val_out need room for nnz + n_rows entries; *nnz_out (host)
 * receives the new entry count.  val NULL = unit weights.  SYNCHRONISES the stream once.  Square matrices. */
size_t disco_csr_insert_diagonal_workspace_bytes(int64_t n_rows);
int disco_csr_insert_diagonal(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, int64_t nnz,
                              int64_t* rowptr_out, int32_t* col_out, float* val_out, int64_t* nnz_out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- Pair-embedding gather for the link predictor's training batches --------------------
 * out[p, :] = [ Z[src[e], 0:width] | Z[dst[e], 0:width] ],  e = sel ? sel[p] : p,  p < n_pairs.
 * Replaces `torch.concat((feat_transform[edge_index[0][index]], feat_transform[edge_index[1][index]]),
 * dim=1).cpu() ... .cuda()` (/root/reference/LargeScaleCondensing.py:188-192) without the host round trip.
 * Z is [n, ldz] float32, src/dst are int64[n_edges] (the two rows of an edge_index), sel is an optional
 * int64[n_pairs] selection of edges, out is [n_pairs, ldo >= 2*width].  An index outside its range
 * yields a zero row and sets *err_flag (optional device int, cleared by the caller) to 1. */
int disco_pair_gather_f32(const float* Z, int64_t n, int64_t ldz, int64_t width, const int64_t* sel, const int64_t* src,
                          const int64_t* dst, int64_t n_edges, int64_t n_pairs, float* out, int64_t ldo, int* err_flag,
                          void* stream);

/* Row gather for the halo exchange of the row-partitioned propagation: out[i, 0:width] = x[idx[i], 0:width],
 * i < n_idx; x [*, ldx], out [n_idx, ldo].  The rows peers asked for go into the send buffer of the all-to-all
 * (disco_amd/dist.py); idx must hold valid row ids of x (checked once when the partition is built).  No counterpart in
 * the reference (single-GPU; `--parallel_gpu_ids` is parsed and never used, /root/reference/LargeScaleCondensing.py:31). */
int disco_gather_rows_f32(const float* x, int64_t ldx, int64_t width, const int64_t* idx, int64_t n_idx, float* out,
                          int64_t ldo, void* stream);

/* ---- Anchor neighbour features of edge_construction -----------------------------------
 * out[dst(i), 0:width] = max over j < k of T[map(anchor[i*k + j]), 0:width],  i < n_out
 *   map(a) = row_map ? row_map[a] : a      (row_map int64[n_map]: the class's train rows, `index[c]`; n_map = 0: none)
 *   dst(i) = out_rows ? out_rows[i] : i    (out_rows int64[n_out]: the class's synthetic rows, `index_syn[c]`)
 * Replaces `feat_syn_neighbor[index_syn[c]] = feat_transform[index[c]][anchor, :L*d].max(dim=1).values`
 * (/root/reference/LargeScaleCondensing.py:299; LargeScaleCondensing_induct.py:293;
 * LargeScaleCondensing_Sampled.py:259-261) without the row-subset copy and the [n_c, k, L*d] intermediate.
 * T is [n, ldt] float32, `width` <= ldt the column cut `:L*d`, anchor int64 [n_out, k] row-major (k = 3 in the
 * reference), out [n_dst, ldo].  The maximum follows torch.max(dim): a NaN propagates.  An index outside its range
 * is skipped (an out-of-range destination row is not written) and sets *err_flag (optional device int, cleared by
 * the caller) to 1.  Rows of out that no dst(i) names are left untouched.  Nothing synchronises. */
int disco_anchor_max_f32(const float* T, int64_t n, int64_t ldt, int64_t width, const int64_t* row_map, int64_t n_map,
                         const int64_t* anchor, int64_t n_out, int64_t k, const int64_t* out_rows, int64_t n_dst,
                         float* out, int64_t ldo, int* err_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISCO_HIP_H */
