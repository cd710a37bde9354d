// Device helpers shared by the SpMM translation units (disco_spmm.hip, disco_stream.hip).
#ifndef DISCO_SPMM_COMMON_H
#define DISCO_SPMM_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"

namespace disco {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef float type; };
template <> struct Vec<2> { typedef f2 type; };
template <> struct Vec<4> { typedef f4 type; };

#ifndef DISCO_U1
#define DISCO_U1 8
#endif
#ifndef DISCO_U2
#define DISCO_U2 4
#endif
#ifndef DISCO_U3
#define DISCO_U3 4
#endif
#ifndef DISCO_U45
#define DISCO_U45 2
#endif

// Cache policy of the windowed kernel's streams (compile-time; measured in profiles/r2/win_cache_policy.log, Reddit):
//   DISCO_WIN_YSTORE 2 (default): partial sums are stored with `buffer_store_dwordx4 ... sc1` -- write-through, the
//     line leaves the XCD's L2 instead of displacing rows of x that the next gathers would hit: -6 % time.
//     1: `nt` stores (no gain).  0: plain.  (`sc0 sc1` stores, what a volatile store compiles to, gave nothing.)
//   DISCO_WIN_YLOAD 1: partial-sum loads carry `nt`; DISCO_CSR_NT 1: (col, val) loads carry `nt` (both: no gain).
#ifndef DISCO_WIN_YLOAD
#define DISCO_WIN_YLOAD 0
#endif
#ifndef DISCO_WIN_YSTORE
#define DISCO_WIN_YSTORE 2
#endif
// waves per SIMD the windowed kernel is compiled for (register budget 512 / waves; with DISCO_U1 gathers of 4 registers
// in flight per lane): 6 waves x 8 gathers measured best (profiles/r3/exp_unroll_occupancy.log)
#ifndef DISCO_WIN_WAVES
#define DISCO_WIN_WAVES 6
#endif
// 1: the windowed kernel requests the (col, val) batch and the y block of ALL its rows before the first gather
#ifndef DISCO_WIN_HOIST
#define DISCO_WIN_HOIST 0
#endif
// consecutive rows one lane group of the windowed kernel handles per launch (software-pipelined one row ahead)
#ifndef DISCO_WIN_MAXR
#define DISCO_WIN_MAXR 2
#endif
#ifndef DISCO_CSR_NT
#define DISCO_CSR_NT 0
#endif
// 1: the windowed kernel starts every visit from zero and adds the running sum from y at the END of the visit
// (y_new = y_old + sum of this window's entries: another association than the plain kernel's single CSR-order chain)
#ifndef DISCO_WIN_YLATE
#define DISCO_WIN_YLATE 0
#endif

template <class T> __device__ __forceinline__ T ld_entries(const T* p) {
  if constexpr (DISCO_CSR_NT != 0) return __builtin_nontemporal_load(p); else return *p;
}

template <int NV> struct Unroll { static constexpr int value = NV == 1 ? DISCO_U1 : NV == 2 ? DISCO_U2 : NV == 3 ? DISCO_U3 : DISCO_U45; };

constexpr int OP_SUM = 0;
constexpr int OP_MAX = 1;
constexpr int FLAG_MEAN = 1;
constexpr int FLAG_ACCUM = 2;

template <int G> __device__ __forceinline__ int bcast_i(int v, int j) {
  if constexpr (G == 64) {
    return __builtin_amdgcn_readlane(v, j);
  } else {
    return __shfl(v, j, G);
  }
}
template <int G> __device__ __forceinline__ float bcast_f(float v, int j) {
  return __int_as_float(bcast_i<G>(__float_as_int(v), j));
}

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

template <int VEC, int OP> __device__ __forceinline__ void vinit(typename Vec<VEC>::type& a) {
  if constexpr (OP == OP_SUM) a = 0.0f; else a = -INFINITY;
}

template <int VEC, int OP, bool HASVAL>
__device__ __forceinline__ void vupdate(typename Vec<VEC>::type& a, float w, const typename Vec<VEC>::type& x) {
  if constexpr (VEC == 1) {
    if constexpr (OP == OP_SUM) a = HASVAL ? fmaf(w, x, a) : a + x;
    else a = fmaxf(a, HASVAL ? w * x : x);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if constexpr (OP == OP_SUM) a[i] = HASVAL ? fmaf(w, x[i], a[i]) : a[i] + x[i];
      else a[i] = fmaxf(a[i], HASVAL ? w * x[i] : x[i]);
    }
  }
}

template <int VEC, int OP> __device__ __forceinline__ void vcombine(typename Vec<VEC>::type& a, const typename Vec<VEC>::type& b) {
  if constexpr (VEC == 1) {
    if constexpr (OP == OP_SUM) a = a + b; else a = fmaxf(a, b);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if constexpr (OP == OP_SUM) a[i] = a[i] + b[i]; else a[i] = fmaxf(a[i], b[i]);
    }
  }
}

// Lane offsets of the NV vectors a lane owns.  Lanes past the row's last vector re-read the last valid
// vector (no branch in the load stream); their accumulators are never stored.
template <int VEC, int G, int NV>
__device__ __forceinline__ void lane_offsets(int nvec, int gl, int64_t (&off)[NV]) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    int idx = gl + G * v;
    idx = idx < nvec ? idx : nvec - 1;
    off[v] = (int64_t)idx * VEC;
  }
}

// Consume one cooperative batch: lane j of the group holds entry j (column cv, value wv), n <= G valid.
// Row gather address.  A24 (host-checked: columns < 2^24, row pitch < 2^24 bytes, table < 4 GiB): the byte
// offset col * pitch + lane offset is ONE full-rate v_mad_u32_u24 and the load takes the uniform base from
// SGPRs; otherwise a full 64-bit multiply-add per gather (three quarter-rate multiplies on gfx950, which
// made the L2-blocked kernel VALU-bound: profiles/r1/pmc_sq_instmix.csv).
template <int VEC, bool A24>
__device__ __forceinline__ typename Vec<VEC>::type gather_vec(const float* __restrict__ x, int64_t ldx, int c, int64_t off) {
  typedef typename Vec<VEC>::type V;
  // Timing experiments only (tools/r3/exp_bounds.sh builds side libraries with these; results are then wrong by design):
  //   DISCO_DBG_COLMASK=m   every gather goes to row (col & m): the kernel's speed if every gather hit the L2
  //   DISCO_DBG_NOGATHER    no gather at all (the value is made from the column id): everything but the gathers
#if defined(DISCO_DBG_COLMASK)
  c &= DISCO_DBG_COLMASK;
#endif
#if defined(DISCO_DBG_NOGATHER)
  { V r; r = (float)c * 1e-9f + (float)off; return r; }
#endif
#if defined(DISCO_GATHER_AUX)
  // experiment (tools/r3/exp_gather_policy.sh): 128-bit gathers through a buffer descriptor with cache-policy bits
  //   aux 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0 sc1 -- does a gather that does not allocate in the L1 go faster?
  if constexpr (A24 && VEC == 4) {
    const uint32_t bo = __umul24((uint32_t)c, (uint32_t)ldx * 4u) + (uint32_t)off * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0xffffffff, 0x27000);
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)bo, 0, DISCO_GATHER_AUX));
  }
#endif
#if defined(DISCO_GATHER_NT)
  // experiment: every gather carries `nt` (does the hint change what the L2 retains?  profiles/r2/gather_nt.log)
  if constexpr (A24) {
    const uint32_t bo = __umul24((uint32_t)c, (uint32_t)ldx * 4u) + (uint32_t)off * 4u;
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(x) + bo));
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(x + (int64_t)c * ldx + off));
  }
#else
  if constexpr (A24) {
    const uint32_t bo = __umul24((uint32_t)c, (uint32_t)ldx * 4u) + (uint32_t)off * 4u;
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(x) + bo);
  } else {
    return *reinterpret_cast<const V*>(x + (int64_t)c * ldx + off);
  }
#endif
}

template <int VEC> __device__ __forceinline__ typename Vec<VEC>::type vdiv(typename Vec<VEC>::type a, float d) {
  if constexpr (VEC == 1) return a / d;
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = a[i] / d;
    return a;
  }
}

// Store vector `idx` of an output row of `nvec` vectors; `tail` (uniform) = number of valid floats in the LAST vector
// (0 = the row width is a multiple of VEC).  A partial last vector is written element by element, so nothing
// beyond column F - 1 of y is ever touched (y may be a column view of a wider live matrix).
template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ dst, const typename Vec<VEC>::type& a, int idx, int nvec, int tail) {
  if constexpr (VEC == 1) {
    *dst = a;
  } else {
    if (tail != 0 && idx == nvec - 1) {
      dst[0] = a[0];
      if constexpr (VEC == 4) {
        if (tail > 1) dst[1] = a[1];
        if (tail > 2) dst[2] = a[2];
      }
    } else {
      *reinterpret_cast<typename Vec<VEC>::type*>(dst) = a;
    }
  }
}

constexpr int PSYNC_GROUPS = 8;
constexpr int64_t PSYNC_BYTES = 1 << 20;       // pacing area at the end of the sweep workspace

}  // namespace disco

#endif
