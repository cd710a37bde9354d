// disco_stream.hip -- streaming persistent L2-window SpMM (sweep_mode 4) for gfx950.
// Part of the path that replaces torch_sparse.matmul (/root/reference/models/convs_large/sg_conv.py:92-93) on graphs
// that re-use source rows heavily (Reddit-shaped); see DESIGN.md 3.4.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"
#include "disco_spmm_common.h"

namespace disco {

// ---- streaming persistent windows (sweep_mode 4) ----------------------------------------------------------
// One wave (G = 64: a column block of up to 256 floats) keeps the running sums of R = 8 output rows in registers
// through ALL windows of the source rows, so y is written once -- and walks each window's entries of its R rows as
// ONE flat stream: lane l holds entry l of the concatenated segments (column, value, row slot), a ring of U row
// gathers stays in flight ACROSS row and window boundaries (entry i + U is requested when entry i is consumed,
// the (col, val) of the next chunk -- same window or next -- are requested a whole chunk ahead, the table entries
// two windows ahead), and the sum of the row being walked lives in one register set that is swapped with the
// row's slot when the stream moves to the next row (a uniform branch, every ~nnz/N/P entries).  Entries of a row
// are consumed in CSR order and the chain continues from the slot's value: same bits as k_spmm_rows.
// Pacing: the workgroups of one sync group (same blockIdx % 8: one XCD under round-robin dispatch) count the
// windows they have finished in per-(group, step) counters; a workgroup starts step s only when everybody has
// finished step s - 2, so the group's gathers stay within two adjacent windows of its L2.  The counter it checks
// was requested one step earlier (no stall when the group is in step), every wait is bounded: the counters only
// pace, no data depends on them, and neither correctness nor termination depends on co-residency.
constexpr int STREAM_R = 8;

struct StreamChunk {         // up to 64 consecutive entries of a window's stream, one per lane
  uint32_t offv;             // byte offset of the entry's source row in x (column * row pitch)
  float valv;                // its weight
  int slotv;                 // the row slot it belongs to
  int n;                     // entries in the chunk (uniform)
};

// (row offset, val, slot) of stream positions [64 k, 64 k + 64) of the window whose segments are [beg, end) (lane r < R
// holds row slot r's).  The segment scalars are recomputed here (a few dozen scalar operations per 64 entries) instead
// of being kept alive across the walk: the kernel is short of scalar registers.
// Every load in this file is issued UNCONDITIONALLY (positions past the end re-read entry 0 / a row that is in flight
// anyway and are dropped by a select): with no vector-memory instruction inside a branch the compiler's s_waitcnt
// insertion keeps exact counts, i.e. consuming ring slot u waits for that load only (vmcnt(U - 1)) instead of
// draining the queue.
template <bool HASVAL>
__device__ __forceinline__ void stream_load_chunk(uint32_t beg, uint32_t end, int k, int lane, const int32_t* __restrict__ col,
                                                  const float* __restrict__ val, uint32_t pitch, StreamChunk& c) {
  const uint32_t p = (uint32_t)k * 64u + (uint32_t)lane;
  uint32_t acc = 0, d = 0;
  int slot = -1;
#pragma unroll
  for (int r = 0; r < STREAM_R; ++r) {
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)beg, r);
    const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)end, r);
    const bool ge = p >= acc;          // row slot r starts at or before this position
    slot += ge ? 1 : 0;
    d = ge ? b - acc : d;
    acc += e - b;
  }
  const uint32_t total = acc, base = (uint32_t)k * 64u;
  c.n = total > base ? (int)(total - base < 64u ? total - base : 64u) : 0;
  c.slotv = slot;
  const uint32_t e = p < total ? p + d : 0u;     // entry 0 always exists (nnz > 0 when this kernel runs)
  c.offv = __umul24((uint32_t)col[e], pitch);
  c.valv = HASVAL ? val[e] : 1.0f;
}

__device__ __forceinline__ uint32_t stream_total(uint32_t beg, uint32_t end) {
  uint32_t acc = 0;
#pragma unroll
  for (int r = 0; r < STREAM_R; ++r)
    acc += (uint32_t)__builtin_amdgcn_readlane((int)end, r) - (uint32_t)__builtin_amdgcn_readlane((int)beg, r);
  return acc;
}

__device__ __forceinline__ int stream_chunks(uint32_t beg, uint32_t end) {
  const int k = (int)((stream_total(beg, end) + 63u) >> 6);
  return k > 0 ? k : 1;            // an empty window still takes one (empty) chunk: the walk visits every window
}

// Gather addresses are 32-bit byte offsets from x (host-checked: columns < 2^24, row pitch < 2^24 bytes, table < 4 GiB).
template <bool HASVAL, int U>
__global__ __launch_bounds__(256, U <= 8 ? 5 : 4) void k_spmm_stream(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                     const float* __restrict__ val, const float* __restrict__ x,
                                                     int64_t ldx, float* __restrict__ y, int64_t ldy, int nvec,
                                                     const uint32_t* __restrict__ perm, int64_t first, int64_t count,
                                                     const uint32_t* __restrict__ wptr, int64_t cp, int P, int flags,
                                                     int n_iter, int* __restrict__ prog, int members, int poll_cap,
                                                     int fcols, int lead) {
  constexpr int R = STREAM_R;
  __shared__ int arrivals[8];          // waves of this workgroup that have finished step s, slot s & 7
  __shared__ int known;                // a step that every workgroup of the sync group is known to have finished
  // The R running sums of a wave live in LDS (wave-private: R x nvec x 16 bytes, no barrier ever needed); only the
  // sum of the row slot being walked is in registers.  The stream enters another slot every nnz/N/P entries: one
  // ds_write_b128 + one ds_read_b128 with a computed address, instead of a register array indexed through branches.
  extern __shared__ __attribute__((aligned(16))) f4 parked[];
  if (threadIdx.x < 8) arrivals[threadIdx.x] = 0;
  if (threadIdx.x == 0) known = -1;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  // slot stride = nvec vectors; lanes past the block alias lane nvec - 1 (they hold the very same values)
  f4* __restrict__ mine = parked + ((size_t)wave * R * nvec + (lane < nvec ? lane : nvec - 1));
  const int64_t total_waves = (int64_t)gridDim.x * 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
  int* __restrict__ myprog = prog + (int64_t)(blockIdx.x & 7) * ((int64_t)n_iter * P);
  const int li = lane < nvec ? lane : nvec - 1;      // lanes past the block re-read its last vector (never stored)
  const uint32_t pitch = (uint32_t)ldx * 4u;
  const uint32_t loff = (uint32_t)li * 16u;
  // uniform base in SGPRs + one 32-bit per-lane offset: `global_load_dwordx4 v, voff, s[base]`
  auto gather = [&](uint32_t row_off) -> f4 { return *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(x) + (row_off + loff)); };
  auto table = [&](int w, int64_t pos) -> uint32_t { return wptr[(int64_t)(w < P ? w : P) * cp + pos]; };

  for (int it = 0; it < n_iter; ++it) {
    // lane r < R carries the table entries of row slot r; lanes without a row read position 0 and see empty segments
    const int64_t pos_raw = ((int64_t)it * R + (lane < R ? lane : 0)) * total_waves + gw;
    const bool valid = lane < R && pos_raw < count;
    const int64_t pos = valid ? pos_raw : 0;
    const uint32_t rowid = perm[first + pos];
    // segment boundaries of windows w .. w + 4 (window j of this row is [t_j, t_j+1)); windows past the last are empty
    uint32_t ta = table(0, pos), tb = table(1, pos), tc = table(2, pos), td = table(3, pos), te = table(4, pos);
    if (!valid) { tb = ta; tc = ta; td = ta; te = ta; }
    // three chunks in flight: `cur` is walked, `nxt` has arrived (requested a whole chunk ago), `nx2` is being fetched
    int w = 0, k = 0;
    StreamChunk cur, nxt, nx2;
    stream_load_chunk<HASVAL>(ta, tb, 0, lane, col, val, pitch, cur);
    {
      const bool same = 1 < stream_chunks(ta, tb);
      stream_load_chunk<HASVAL>(same ? ta : tb, same ? tb : tc, same ? 1 : 0, lane, col, val, pitch, nxt);
    }
    // Both chunks must have ARRIVED before the ring is primed: the walk below then starts in the state every later
    // chunk starts in (only ring loads and the chunk after next outstanding), and the compiler's wait counts inside
    // the loop are the steady-state ones (vmcnt(U - 1)) instead of the prologue's.
    asm volatile("" : "+v"(cur.offv), "+v"(cur.valv), "+v"(nxt.offv), "+v"(nxt.valv));
#pragma unroll
    for (int r = 0; r < R; ++r) mine[r * nvec] = 0.0f;
    f4 run = 0.0f;        // sum of the row slot being walked
    int cs = 0;           // that slot
    f4 ring[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ring[u] = gather((uint32_t)__builtin_amdgcn_readlane((int)cur.offv, u));   // lanes >= n hold the row of col[0]
      __builtin_amdgcn_sched_barrier(0);     // issue in ring order: the wait counts in the walk assume it
    }
    while (w < P) {
      // where the walk goes next, and after that
      const int K0 = stream_chunks(ta, tb), K1 = stream_chunks(tb, tc);
      const int w1 = k + 1 < K0 ? w : w + 1, k1 = k + 1 < K0 ? k + 1 : 0;
      const int Kw1 = w1 == w ? K0 : K1;
      const int w2 = k1 + 1 < Kw1 ? w1 : w1 + 1, k2 = k1 + 1 < Kw1 ? k1 + 1 : 0;
      {
        const int j = w2 - w;                    // 0, 1 or 2
        const uint32_t nb = j == 0 ? ta : (j == 1 ? tb : tc);
        const uint32_t ne = j == 0 ? tb : (j == 1 ? tc : td);
        stream_load_chunk<HASVAL>(nb, ne, k2, lane, col, val, pitch, nx2);
      }
      const int n = cur.n;
      const int nfull = n / U * U;                         // positions walked without a predicate
      const int nr = n > 0 ? (n + U - 1) / U * U : U;      // an empty chunk still takes one pass: it primes the ring
      // lookahead source of every position: stream position lane + U of (this chunk, padded to nr) ++ (next chunk)
      uint32_t lav;
      {
        const int pl = lane + U;
        const uint32_t from_cur = (uint32_t)__builtin_amdgcn_ds_bpermute((pl & 63) << 2, (int)cur.offv);
        int ql = pl - nr;
        ql = ql < 0 ? 0 : ql;
        const uint32_t from_nxt = (uint32_t)__builtin_amdgcn_ds_bpermute((ql & 63) << 2, (int)nxt.offv);
        lav = pl < n ? from_cur : from_nxt;
      }
      // positions where the stream enters another row slot (position 0 always checks)
      const unsigned long long chg = __ballot(lane == 0 || cur.slotv != __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) << 2, cur.slotv));
      int base = 0;
      for (; base < nfull; base += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u;
          if ((chg >> i) & 1ull) {
            const int sl = __builtin_amdgcn_readlane(cur.slotv, i);
            if (sl != cs) {
              // the stream moves on to the next row slot: park the running sum, continue that slot's chain
              mine[cs * nvec] = run;
              run = mine[sl * nvec];
              cs = sl;
            }
          }
          const float wv = HASVAL ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur.valv), i)) : 1.0f;
          vupdate<4, OP_SUM, HASVAL>(run, wv, ring[u]);
          ring[u] = gather((uint32_t)__builtin_amdgcn_readlane((int)lav, i));     // keep the ring full
        }
      }
      if (base < nr) {
        // last, partial pass (also the one pass of an empty chunk): positions past the end are dropped by a select
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u;
          if (i < n && ((chg >> i) & 1ull)) {
            const int sl = __builtin_amdgcn_readlane(cur.slotv, i);
            if (sl != cs) {
              mine[cs * nvec] = run;
              run = mine[sl * nvec];
              cs = sl;
            }
          }
          const float wv = HASVAL ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cur.valv), i)) : 1.0f;
          f4 t = run;
          vupdate<4, OP_SUM, HASVAL>(t, wv, ring[u]);
          run = i < n ? t : run;
          ring[u] = gather((uint32_t)__builtin_amdgcn_readlane((int)lav, i));
        }
      }
      cur = nxt;
      nxt = nx2;
      if (w1 != w) {
        // window w is done.  Count it: the last wave of the workgroup to arrive adds one to the sync group's counter.
        const int step = it * P + w;
        if (lane == 0) {
          const int old = __hip_atomic_fetch_add(&arrivals[step & 7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (old == 3) {
            __hip_atomic_store(&arrivals[step & 7], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(myprog + step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        // Pacing: start step + 1 only when the whole sync group has finished step - lead (the counters of a group
        // complete in step order).  What the workgroup already knows is cached in LDS (`known` = a step every member has
        // finished), so the group counter -- one hot word per (group, step) -- is read by one wave per workgroup at
        // most, and by none when a later step is already known complete.  Bounded: the counters pace, no data
        // depends on them.
        if (members > 0 && step >= lead) {
          const int need = step - lead;
          int have = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          for (int polls = 0; have < need && polls < poll_cap; ++polls) {
            // the newest step that can be complete is the one this wave has just finished; probe towards it
            const int probe = polls == 0 ? (need + lead - 1 > need ? need + (lead - 1) / 2 : need) : need;
            const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(myprog + probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (v >= members) {
              if (lane == 0) __hip_atomic_fetch_max(&known, probe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              have = probe;
            } else if (polls > 0) {
              __builtin_amdgcn_s_sleep(8);
              have = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            }
          }
        }
        // move the segment tables on; the entries five windows ahead are requested now
        ta = tb; tb = tc; tc = td; td = te;
        te = table(w + 5, pos);
        if (!valid) te = ta;
      }
      w = w1;
      k = k1;
    }
    mine[cs * nvec] = run;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool ok = __builtin_amdgcn_readlane((int)valid, r) != 0;
      const int64_t row = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)rowid, r);
      if (ok && lane < nvec) {
        f4 a = mine[r * nvec];
        float* dst = y + row * ldy + (int64_t)lane * 4;
        if (flags & FLAG_MEAN) {
          const float cnt = (float)(rowptr[row + 1] - rowptr[row]);
          if (cnt > 0.0f) a = vdiv<4>(a, cnt);
        }
        if (flags & FLAG_ACCUM) a = *reinterpret_cast<const f4*>(dst) + a;
        store_vec<4>(dst, a, lane, nvec, fcols & 3);
      }
    }
  }
}

static int stream_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <bool HASVAL, int U>
static int launch_stream_t(const StreamArgs& a) {
  const disco_schedule_info& s = *a.info;
  const uint32_t* wptr = reinterpret_cast<const uint32_t*>(s.sweep_ws);
  const int P = (int)s.sweep_windows;
  int per_cu = 0;
  const size_t lds_bytes = (size_t)4 * STREAM_R * a.nvec * 16;      // 4 waves x R rows x nvec float4
  hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k_spmm_stream<HASVAL, U>), 256, lds_bytes);
  if (oe != hipSuccess) { set_last_hip_error((int)oe); return DISCO_EHIP; }
  if (tuning().sweep_blocks_per_cu > 0 && tuning().sweep_blocks_per_cu < per_cu) per_cu = tuning().sweep_blocks_per_cu;
  int64_t blocks = (int64_t)stream_cus() * (per_cu > 0 ? per_cu : 1);
  const int64_t want = (a.count + 4 * STREAM_R - 1) / (4 * STREAM_R);
  if (blocks > want) blocks = want;
  blocks = blocks / PSYNC_GROUPS * PSYNC_GROUPS;
  if (blocks < PSYNC_GROUPS) return DISCO_EUNSUPPORTED;
  const int64_t per_iter = blocks * 4 * STREAM_R;
  const int n_iter = (int)((a.count + per_iter - 1) / per_iter);
  const size_t prog_bytes = (size_t)n_iter * P * PSYNC_GROUPS * sizeof(int);
  if (prog_bytes > (size_t)PSYNC_BYTES) return DISCO_EUNSUPPORTED;
  int* prog = reinterpret_cast<int*>(const_cast<char*>(static_cast<const char*>(s.sweep_ws)) +
                                     (size_t)(P + 1) * s.sweep_count_padded * sizeof(uint32_t));
  DISCO_HIP_TRY(hipMemsetAsync(prog, 0, prog_bytes, a.stream));
  const int cap = tuning().psync_polls > 0 ? tuning().psync_polls : 64;
  const int members = tuning().stream_lead < 0 ? 0 : (int)(blocks / PSYNC_GROUPS);     // stream_lead < 0: no pacing
  hipLaunchKernelGGL((k_spmm_stream<HASVAL, U>), dim3((unsigned)blocks), dim3(256), lds_bytes, a.stream, a.rowptr, a.col, a.val,
                     a.x, a.ldx, a.y, a.ldy, a.nvec, a.perm, a.first, a.count, wptr, s.sweep_count_padded, P, a.flags, n_iter,
                     prog, members, cap, a.fcols, tuning().stream_lead > 0 ? tuning().stream_lead : 2);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

int launch_stream(const StreamArgs& a) {
  // 32-bit gather offsets only (columns < 2^24, pitch < 2^24 bytes, table < 4 GiB): anything else stays on the
  // windowed launches
  if (a.nvec < 1 || a.nvec > 64 || !a.perm || !a.addr24) return DISCO_EUNSUPPORTED;
  if (tuning().stream_u == 16 && a.val) return launch_stream_t<true, 16>(a);     // experiment
  if (tuning().stream_u == 12 && a.val) return launch_stream_t<true, 12>(a);     // experiment
  return a.val ? launch_stream_t<true, 8>(a) : launch_stream_t<false, 8>(a);
}

}  // namespace disco
