// Internal declarations shared by the translation units of libdisco_hip.so.
#ifndef DISCO_INTERNAL_H
#define DISCO_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/disco_hip.h"

namespace disco {

struct Tuning {
  int force_g = 0;    // 0 = automatic
  int force_nv = 0;   // 0 = automatic
  int max_vec = 0;    // 0 = 4
  int sweep_mode = 2; // with a window table attached: 0 = ignore it, 1 = persistent sweep, 2 = windowed launches, 3 = paced persistent windows, 4 = streaming persistent windows, 5 = XCD-sliced skewed schedule of the windowed launches
  int win_cols = 0;   // column block width of the windowed launches (0 = 128)
  int win_rows_per_group = 0;  // consecutive rows one lane group handles per windowed launch (0 = 2, at most 2)
  int sweep_blocks_per_cu = 0;  // 0 = 4
  int psync_pct = 0;  // sweep_mode 3: share of a sync group's waves a wave waits for (0 = 100)
  int psync_polls = 0;  // sweep_mode 3: polls before a wave gives up waiting (0 = 256)
  int addr24 = 1;     // 0 = always 64-bit gather addressing
  int row_cols = 0;   // column block of the plain row kernel: 0 = auto (128 floats under a group-major schedule, else one pass up to 1280), -1 = never block, > 0 = that many floats
  int narrow_groups = 1;   // 0 = never use the 4- / 8-lane row groups (rows of <= 8 float4)
  int stream_u = 0;   // sweep_mode 4: gathers in flight per wave (0 = 8; 12 / 16: experiments)
  int stream_lead = 0;  // sweep_mode 4: window steps a wave may run ahead of its XCD group (0 = 2; < 0: no pacing)
  int xcd_map = -1;   // k_spmm_rows workgroup order: -1 = what the schedule says (info->xcd_run); 0 = off; 1 = each XCD walks one contiguous eighth of the row order; S > 1 = runs of S rows dealt to the XCDs in turn
};

Tuning& tuning();
void set_last_hip_error(int e);

// streaming persistent window kernel (disco_stream.hip), one column block of one hop
struct StreamArgs {
  const int64_t* rowptr; const int32_t* col; const float* val; const float* x; int64_t ldx; float* y; int64_t ldy;
  int nvec; int fcols; const uint32_t* perm; int64_t first; int64_t count; const disco_schedule_info* info; int flags;
  bool addr24; hipStream_t stream;
};
int launch_stream(const StreamArgs& a);   // DISCO_EUNSUPPORTED: shape outside the kernel, caller falls back

// ---- launch shapes ---------------------------------------------------------------------------------------------
// A dispatch carries its size per dimension as a 32-bit count of WORK-ITEMS: a 1-D launch of 2^32 or more work-items
// (17 M workgroups of 256) does not fail, it silently runs `count mod 2^32` of them (measured on MI355X / ROCm 7.2,
// tools/r3/gridlimit: 28 M workgroups asked for, 11.2 M ran, hipSuccess everywhere).  A 2-D grid whose dimensions
// each stay below the limit runs in full.  grid_for() therefore cuts a workgroup count into (x, y) with at most 2^30
// work-items along x (x a multiple of 8 when y > 1, so that `id % 8` keeps naming the XCD of a round-robin dispatch);
// kernels launched this way read their linear workgroup id with block_id() and return for ids >= their count
// (x * y may exceed it).  Kernels that keep using blockIdx.x alone are launched through grid_1d(), which refuses
// (DISCO_EUNSUPPORTED) what would wrap.
inline bool grid_for(int64_t blocks, int threads, dim3* g) {
  if (blocks < 0 || threads <= 0) return false;
  if (blocks == 0) blocks = 1;
  const int64_t max_x = ((int64_t)1 << 30) / threads;
  if (blocks <= max_x) { *g = dim3((unsigned)blocks, 1, 1); return true; }
  const int64_t by = (blocks + max_x - 1) / max_x;
  int64_t bx = (blocks + by - 1) / by;
  bx = (bx + 7) / 8 * 8;
  if (by > 65535) return false;
  *g = dim3((unsigned)bx, (unsigned)by, 1);
  return true;
}
inline bool grid_1d(int64_t blocks, int threads, dim3* g) {
  if (blocks < 0 || threads <= 0 || (uint64_t)(blocks > 0 ? blocks : 1) * (uint64_t)threads >= ((uint64_t)1 << 32)) return false;
  *g = dim3((unsigned)(blocks > 0 ? blocks : 1), 1, 1);
  return true;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int64_t block_id() { return (int64_t)blockIdx.y * gridDim.x + blockIdx.x; }
#endif
#define DISCO_GRID(g, blocks, threads) \
  dim3 g;                              \
  if (!::disco::grid_for((blocks), (threads), &g)) return DISCO_EUNSUPPORTED
#define DISCO_GRID_1D(g, blocks, threads) \
  dim3 g;                                 \
  if (!::disco::grid_1d((blocks), (threads), &g)) return DISCO_EUNSUPPORTED

#define DISCO_HIP_TRY(expr)                                    \
  do {                                                         \
    hipError_t e__ = (expr);                                   \
    if (e__ != hipSuccess) {                                   \
      ::disco::set_last_hip_error((int)e__);                   \
      return DISCO_EHIP;                                       \
    }                                                          \
  } while (0)

#define DISCO_CHECK_LAUNCH()                                   \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) {                                   \
      ::disco::set_last_hip_error((int)e__);                   \
      return DISCO_EHIP;                                       \
    }                                                          \
  } while (0)

}  // namespace disco

// Not part of the public header: tuning knobs for benchmarks ("force_g", "force_nv", "max_vec").
extern "C" int disco_set_tuning(const char* key, int64_t value);

#endif
