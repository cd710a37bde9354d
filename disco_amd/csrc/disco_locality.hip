// disco_locality.hip -- device-side locality pass for the row schedule (gfx950).
//
// On a graph WITH community structure (co-purchase, citation, social graphs: the reference's datasets) rows of one
// community gather mostly the same source rows.  If they are processed together -- and, with the XCD-aware block
// order, by the same XCD -- those gathers hit that XCD's 4 MiB L2 instead of crossing the fabric.  The ids a dataset
// ships with need not reflect that structure, so this pass recovers it: label propagation (every node takes the most
// frequent label among its neighbours and itself, ties to the label with the smaller hash -- never to the smaller id,
// which on a graph whose ids follow its communities lets the low-numbered communities flood their neighbours;
// semi-synchronous by id parity and double buffered, so the result does not depend on scheduling), the final labels
// are the `row_group` key of disco_schedule_build_grouped.
// Only the ORDER in which rows are processed changes: X, the column ids and every row's summation order stay as they
// are, so results are bit-identical with and without it.  No counterpart in the reference (`pymetis` is pinned in
// /root/reference/requirements.txt:95 but never used).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"

namespace disco {

constexpr int kLpaSlots = 1024;         // distinct neighbour labels a row can tally; labels that find no slot are not counted
constexpr int64_t kLpaMaxDeg = 4096;    // longer rows (hubs: connected to everything) keep their label
constexpr uint32_t kLpaEmpty = 0xffffffffu;
constexpr uint32_t kLpaHashMul = 2654435761u;   // odd: label -> label * mul (mod 2^32) is a bijection ...
constexpr uint32_t kLpaHashInv = 244002641u;    // ... and this is its inverse (mul * inv == 1 mod 2^32)

// CAPPED: `size[label]` (device, uint32 per label, maintained with atomics) bounds every group at `cap` members -- a
// plain label propagation lets one label swallow most of a graph that has hubs.  A row that wants to join a full
// group takes its next choice (up to three tries).  Which rows get the last places of a group depends on the order
// in which waves arrive, so with a cap the labels are not reproducible from run to run; only the ORDER of processing
// depends on them, never a result.
template <bool CAPPED>
__global__ __launch_bounds__(256) void k_lpa_step(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                  int64_t n, const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                  int phase, uint32_t* __restrict__ size, uint32_t cap) {
  __shared__ uint32_t keys[4][kLpaSlots];
  __shared__ uint32_t cnts[4][kLpaSlots];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t v = block_id() * 4 + wave;
  if (v >= n) return;
  const uint32_t own = in[v];
  const int64_t b = rowptr[v], e = rowptr[v + 1];
  if (((v + phase) & 1) || e == b || e - b > kLpaMaxDeg) {      // not this half-step's turn / nothing to learn from
    if (lane == 0) out[v] = own;
    return;
  }
  uint32_t* k = keys[wave];
  uint32_t* c = cnts[wave];
  for (int s = lane; s < kLpaSlots; s += 64) { k[s] = kLpaEmpty; c[s] = 0u; }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the table is wave-private: order its phases within the wave
  bool overflow = false;
  auto tally = [&](uint32_t label) {
    uint32_t slot = (label * kLpaHashMul) >> 22;                // 10 bits
    for (int probes = 0; probes < 64; ++probes) {
      const uint32_t old = atomicCAS(&k[slot], kLpaEmpty, label);
      if (old == kLpaEmpty || old == label) { atomicAdd(&c[slot], 1u); return; }
      slot = (slot + 1) & (kLpaSlots - 1);
    }
    overflow = true;                                             // crowded table: this neighbour's vote is dropped
  };
  if (lane == 0) tally(own);
  for (int64_t i = b + lane; i < e; i += 64) {
    const int64_t u = col[i];
    if (u != v && u < n) tally(in[u]);
  }
  (void)overflow;   // a row with more distinct neighbour labels than slots decides on the votes that fitted
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  uint32_t chosen = own;
  for (int attempt = 0; attempt < (CAPPED ? 3 : 1); ++attempt) {
    // most frequent label, ties to the smaller hash (a bijection of the label): max of (count, ~hash)
    unsigned long long best = 0;
    for (int s = lane; s < kLpaSlots; s += 64) {
      if (k[s] != kLpaEmpty && c[s] != 0u) {
        const unsigned long long cand = ((unsigned long long)c[s] << 32) | (unsigned long long)(~(k[s] * kLpaHashMul));
        best = cand > best ? cand : best;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = ((unsigned long long)(unsigned int)__shfl_xor((int)(best >> 32), o) << 32) |
                                       (unsigned int)__shfl_xor((int)(best & 0xffffffffu), o);
      best = other > best ? other : best;
    }
    if (best == 0) break;                                        // every choice was full: stay
    const uint32_t want = (~(uint32_t)(best & 0xffffffffu)) * kLpaHashInv;
    if (!CAPPED || want == own) { chosen = want; break; }
    int ok = 0;
    if (lane == 0) {
      const uint32_t had = atomicAdd(&size[want], 1u);
      if (had < cap) { atomicSub(&size[own], 1u); ok = 1; }
      else atomicSub(&size[want], 1u);
    }
    ok = __shfl(ok, 0);
    if (ok) { chosen = want; break; }
    for (int s = lane; s < kLpaSlots; s += 64)                   // strike the full group and choose again
      if (k[s] == want) c[s] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (lane == 0) out[v] = chosen;
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* __restrict__ p, int64_t n, uint32_t value) {
  const int64_t i = block_id() * 256 + threadIdx.x;
  if (i < n) p[i] = value;
}

__global__ __launch_bounds__(256) void k_iota_u32(uint32_t* __restrict__ p, int64_t n) {
  const int64_t i = block_id() * 256 + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

}  // namespace disco

using namespace disco;

extern "C" size_t disco_locality_workspace_bytes(int64_t n_rows) { return n_rows > 0 ? (size_t)n_rows * 8 + 512 : 0; }

extern "C" int disco_locality_labels(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int sweeps,
                                     int64_t max_group, uint32_t* labels, void* ws, size_t ws_bytes, void* stream_) {
  if (n_rows < 0 || sweeps < 0 || max_group < 0) return DISCO_EINVAL;
  if (n_rows == 0) return DISCO_OK;
  if (n_rows > 0xfffffffeLL) return DISCO_EUNSUPPORTED;
  if (!ws || ws_bytes < disco_locality_workspace_bytes(n_rows)) return DISCO_EWORKSPACE;
  if (!rowptr || !labels || (!col && sweeps)) return DISCO_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  uint32_t* a = labels;
  uint32_t* b = static_cast<uint32_t*>(ws);
  uint32_t* size = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + (((size_t)n_rows * 4 + 255) & ~(size_t)255));
  DISCO_GRID(fill_grid, (n_rows + 255) / 256, 256);
  hipLaunchKernelGGL(k_iota_u32, fill_grid, dim3(256), 0, s, a, n_rows);
  DISCO_CHECK_LAUNCH();
  const bool capped = max_group > 0;
  const uint32_t cap = (uint32_t)(max_group > 0xffffffffLL ? 0xffffffffLL : max_group);
  if (capped) {
    hipLaunchKernelGGL(k_fill_u32, fill_grid, dim3(256), 0, s, size, n_rows, 1u);   // every row alone in its group
    DISCO_CHECK_LAUNCH();
  }
  const int64_t blocks = (n_rows + 3) / 4;
  DISCO_GRID(step_grid, blocks, 256);      // one wave per row: 64 work-items per row, 2^32 of them at 67 M rows
  for (int h = 0; h < 2 * sweeps; ++h) {        // two half-steps per sweep: an even number, the result ends up in `labels`
    if (capped)
      hipLaunchKernelGGL(k_lpa_step<true>, step_grid, dim3(256), 0, s, rowptr, col, n_rows, (const uint32_t*)a, b, h & 1, size, cap);
    else
      hipLaunchKernelGGL(k_lpa_step<false>, step_grid, dim3(256), 0, s, rowptr, col, n_rows, (const uint32_t*)a, b, h & 1, size, cap);
    DISCO_CHECK_LAUNCH();
    uint32_t* t = a; a = b; b = t;
  }
  return DISCO_OK;
}
