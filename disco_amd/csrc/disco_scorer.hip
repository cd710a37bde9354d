// disco_scorer.hip -- PGE_Edge pair scorer over all condensed-node pairs (gfx950, fp32 MFMA).
//
// Replaces the pair loop of edge_construction():
//   for i in range(n): adj_syn[i] = pge_edge.inference(cat([Z[[i]*n], Z[arange(n)]], 1))
//   (/root/reference/LargeScaleCondensing.py:304-305; model: /root/reference/models/parametrized_adj_lp.py:29-38)
// and the symmetrise / threshold / nonzero post-processing (LargeScaleCondensing.py:306-310,
// LargeScaleCondensing_induct.py:300-304, LargeScaleCondensing_Sampled.py:271-285).
//
// Factorisation (eval-mode BatchNorm folded, s = gamma / sqrt(var + eps), t = beta - mean * s):
//   layer 1  h1[i,j,:] = relu(U[i,:] + V[j,:]),  U = s1 * (Z W1a^T),  V = s1 * (Z W1b^T + b1) + t1
//            (the concatenation [z_i || z_j] splits W1 = [W1a | W1b]: two n x H GEMMs instead of n^2)
//   layer 2  h2 = relu(W2' h1 + b2'),  W2' = diag(s2) W2,  b2' = s2 * b2 + t2      <- the only n^2 GEMM
//   layer 3  logit = w3 . h2 + b3;  S = sigmoid(logit)
// Layer 2 is a true [n^2, H] x [H, H] GEMM and runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32:
// bit-for-bit an fmaf chain, no reduced-precision path) because edges are kept where the symmetrised
// sigmoid exceeds 0.99..0.9999 and small logit errors would flip them.  h1 is generated on the fly into
// LDS, W2' lives in registers (each of the 8 waves of a workgroup owns 32 output features), the layer-3
// dot product and the sigmoid are fused into the epilogue: nothing but S is written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "disco_internal.h"

namespace disco {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4s __attribute__((ext_vector_type(4)));

struct ScorerWs {  // layout of the scorer workspace (all float32)
  int64_t off_u, off_v, off_w2, off_b2, off_w3, off_b3, total;
};

static ScorerWs scorer_layout(int64_t n, int64_t H) {
  ScorerWs L;
  int64_t o = 0;
  auto take = [&](int64_t floats) { int64_t at = o; o = (o + floats * 4 + 255) / 256 * 256; return at; };
  L.off_u = take(n * H);
  L.off_v = take(n * H);
  L.off_w2 = take(H * H);
  L.off_b2 = take(H);
  L.off_w3 = take(H);
  L.off_b3 = take(64);
  L.total = o;
  return L;
}

// ---- prepare: U, V (float32 dot products in k order), folded W2', b2', w3, b3 ----------------------
// A workgroup computes a 64-node x 64-unit tile of U and V; Z and the two halves of W1 are staged through LDS
// in chunks of 16 k-values, each thread owns 4 nodes x 4 units.  Every output is one fmaf chain over
// k = 0 .. D-1 in ascending order (what a plain dot-product loop gives), whatever the tiling.
constexpr int kUvTile = 64, kUvK = 16;
__global__ __launch_bounds__(256) void k_scorer_uv(const float* __restrict__ Z, int64_t n, int64_t D, int64_t ldz,
                                                   const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ g1, const float* __restrict__ be1,
                                                   const float* __restrict__ m1, const float* __restrict__ v1, float eps,
                                                   int H, float* __restrict__ U, float* __restrict__ V) {
  __shared__ float zt[kUvTile][kUvK + 1], wa[kUvTile][kUvK + 1], wb[kUvTile][kUvK + 1];
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * kUvTile;
  const int h0 = blockIdx.y * kUvTile;
  float su[4][4], sv[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { su[a][b] = 0.0f; sv[a][b] = 0.0f; }
  for (int64_t k0 = 0; k0 < D; k0 += kUvK) {
    // stage: 64 x 16 elements of each operand, 4 per thread (k fastest: coalesced 64-byte rows)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 256 * q, r = e >> 4, k = e & 15;
      const int64_t kk = k0 + k;
      const int64_t i = i0 + r;
      const int h = h0 + r;
      zt[r][k] = (i < n && kk < D) ? Z[i * ldz + kk] : 0.0f;
      wa[r][k] = (h < H && kk < D) ? W1[(int64_t)h * 2 * D + kk] : 0.0f;
      wb[r][k] = (h < H && kk < D) ? W1[(int64_t)h * 2 * D + D + kk] : 0.0f;
    }
    __syncthreads();
    const int kmax = D - k0 < kUvK ? (int)(D - k0) : kUvK;     // padding must not extend the fmaf chain
    for (int k = 0; k < kmax; ++k) {
      float zr[4], ar[4], br[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) zr[a] = zt[ty * 4 + a][k];
#pragma unroll
      for (int b = 0; b < 4; ++b) { ar[b] = wa[tx * 4 + b][k]; br[b] = wb[tx * 4 + b][k]; }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { su[a][b] = fmaf(zr[a], ar[b], su[a][b]); sv[a][b] = fmaf(zr[a], br[b], sv[a][b]); }
    }
    __syncthreads();
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int h = h0 + tx * 4 + b;
    if (h >= H) continue;
    const float s = g1[h] / sqrtf(v1[h] + eps);
    const float tt = be1[h] - m1[h] * s;
    const float bb = b1[h];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int64_t i = i0 + ty * 4 + a;
      if (i < n) {
        U[i * H + h] = s * su[a][b];
        V[i * H + h] = s * (sv[a][b] + bb) + tt;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_scorer_fold(const float* __restrict__ W2, const float* __restrict__ b2,
                                                     const float* __restrict__ g2, const float* __restrict__ be2,
                                                     const float* __restrict__ m2, const float* __restrict__ v2,
                                                     const float* __restrict__ W3, const float* __restrict__ b3, float eps, int H,
                                                     float* __restrict__ W2f, float* __restrict__ b2f, float* __restrict__ w3,
                                                     float* __restrict__ b3o) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < H * H) {
    const int o = idx / H;
    const float s = g2[o] / sqrtf(v2[o] + eps);
    W2f[idx] = s * W2[idx];
  }
  if (idx < H) {
    const float s = g2[idx] / sqrtf(v2[idx] + eps);
    b2f[idx] = s * b2[idx] + (be2[idx] - m2[idx] * s);
    w3[idx] = W3[idx];
  }
  if (idx == 0) b3o[0] = b3[0];
}

// ---- the n^2 part --------------------------------------------------------------------------------------
// Workgroup = 8 waves (512 threads); tile = one i and 64 consecutive j (64 pairs); wave w owns output
// features [32w, 32w+32) of layer 2 (waves beyond H/32 idle for small H).
// MFMA 32x32x2 f32: lane (r = l & 31, hh = l >> 5) feeds A[pair r][k] and B[k][feature r] with
// k = hh * H/2 + m at step m (the k order of the fma chain is a free choice as long as A and B agree).
// sum over the 32 lanes of a half (= 32 features) of w3 * relu(acc[g] + bias) for all 16 registers g at once:
// butterfly transpose-reduce (8+4+2+1 exchanges), then one exchange with lane ^ 16.  Lane r returns the
// total for register g = r & 15.
template <int W>
__device__ __forceinline__ void fold_step(float (&v)[16], int r) {
  // lanes with bit W set keep the upper W values, the others the lower W; the partner supplies the matching half
  const bool upper = (r & W) != 0;
#pragma unroll
  for (int q = 0; q < W; ++q) {
    const float lo = v[q], hi = v[q + W];
    const float keep = upper ? hi : lo;
    const float send = upper ? lo : hi;
    v[q] = keep + __shfl_xor(send, W);
  }
}

__device__ __forceinline__ float reduce_features(const f32x16& acc, float bias, float w3v, int r) {
  float v[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) v[g] = w3v * fmaxf(acc[g] + bias, 0.0f);
  fold_step<8>(v, r);
  fold_step<4>(v, r);
  fold_step<2>(v, r);
  fold_step<1>(v, r);
  return v[0] + __shfl_xor(v[0], 16);
}

template <int H>
__global__ __launch_bounds__(512) void k_scorer_pairs(const float* __restrict__ U, const float* __restrict__ V,
                                                      const float* __restrict__ W2f, const float* __restrict__ b2f,
                                                      const float* __restrict__ w3, const float* __restrict__ b3,
                                                      int64_t i0, int64_t i1, int64_t j0, int64_t j1, float* __restrict__ out,
                                                      int64_t ldo, int apply_sigmoid) {
  constexpr int HH = H / 2;        // k values per lane half
  constexpr int LDA = H + 4;       // padded LDS row (floats): conflict-free ds_read_b128
  constexpr int NW = H / 32;       // waves with work
  constexpr int PER = H / 8;       // h1 elements per thread and tile (= MFMA loop trip count HH / 4)
  constexpr int PSTEP = 512 / H;   // pair rows between two elements of one thread
  constexpr int DEPTH = PER < 8 ? PER : 8;   // V loads in flight per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // lds: h1 tile [2][64][LDA] (double buffered), then per-wave partial logits [2][8][64]
  float* red_base = lds + 2 * 64 * LDA;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool active = wave < NW;
  // B fragments: W2'[feature 32*wave + r][hh*HH + m], m = 0..HH-1, kept in registers for the whole kernel
  float breg[HH];
  float bias = 0.0f, w3v = 0.0f;
  if (active) {
    const float* wrow = W2f + (int64_t)(32 * wave + r) * H + hh * HH;
#pragma unroll
    for (int m = 0; m < HH; m += 4) {
      f4s t = *reinterpret_cast<const f4s*>(wrow + m);
      breg[m] = t[0]; breg[m + 1] = t[1]; breg[m + 2] = t[2]; breg[m + 3] = t[3];
    }
    bias = b2f[32 * wave + r];
    w3v = w3[32 * wave + r];
  }
  const float b3v = b3[0];
  const int64_t nj = j1 - j0;
  const int64_t jblocks = (nj + 63) / 64;
  const int64_t tiles = (i1 - i0) * jblocks;
  // generation role of this thread: column k of pair rows pofs, pofs + PSTEP, ...
  const int gk = tid % H, pofs = tid / H;
  const int gen_off = pofs * LDA + gk;                 // + q * PSTEP * LDA
  const int mma_off = r * LDA + hh * HH;               // + 32 * LDA for the second pair block

  int64_t t = blockIdx.x;
  if (t >= tiles) return;
  {  // prologue: first tile straight into buffer 0
    const int64_t i = i0 + t / jblocks, jb = j0 + (t % jblocks) * 64;
    const int pmax = (int)((j1 - 1 - jb) < 63 ? (j1 - 1 - jb) : 63);
    const float u = U[i * H + gk];
    const float* vb = V + jb * H + gk;
#pragma unroll 8
    for (int q = 0; q < PER; ++q) {
      int p = q * PSTEP + pofs;
      const int pc = p < pmax ? p : pmax;
      lds[gen_off + q * PSTEP * LDA] = fmaxf(u + vb[pc * H], 0.0f);
    }
  }
  __syncthreads();
  int buf = 0;
  for (; t < tiles; t += gridDim.x) {
    const int64_t i = i0 + t / jblocks, jb = j0 + (t % jblocks) * 64;
    // the tile generated while this one is multiplied (the last round regenerates its own tile: harmless)
    const int64_t tn = t + gridDim.x < tiles ? t + gridDim.x : t;
    const int64_t in = i0 + tn / jblocks, jbn = j0 + (tn % jblocks) * 64;
    const int pmax = (int)((j1 - 1 - jbn) < 63 ? (j1 - 1 - jbn) : 63);
    const float un = U[in * H + gk];
    const float* vb = V + jbn * H + gk;
    const float* a_cur = lds + buf * (64 * LDA);
    float* a_nxt = lds + (buf ^ 1) * (64 * LDA) + gen_off;
    float ring[DEPTH];
    f32x16 acc0 = {0}, acc1 = {0};
    const float* a0 = a_cur + mma_off;
    const float* a1 = a0 + 32 * LDA;
    f4s x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
    if (active) {
      x0 = *reinterpret_cast<const f4s*>(a0);
      x1 = *reinterpret_cast<const f4s*>(a1);
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      {  // issue the V load of element `it` of the next tile
        const int p = it * PSTEP + pofs;
        const int pc = p < pmax ? p : pmax;
        ring[it % DEPTH] = vb[pc * H];
      }
      if (active) {
        // fragments of the next step are fetched while this step's 8 MFMAs issue; the scheduling barrier keeps
        // the compiler from hoisting more LDS reads / V loads than that (register budget: 128 for W2')
        const int m = 4 * it;
        f4s n0 = x0, n1 = x1;
        if (m + 4 < HH) {
          n0 = *reinterpret_cast<const f4s*>(a0 + m + 4);
          n1 = *reinterpret_cast<const f4s*>(a1 + m + 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[q], breg[m + q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[q], breg[m + q], acc1, 0, 0, 0);
        }
        x0 = n0; x1 = n1;
      }
      if (it >= DEPTH - 1) {  // retire the oldest load into the other buffer
        const int e = it - (DEPTH - 1);
        a_nxt[e * PSTEP * LDA] = fmaxf(un + ring[e % DEPTH], 0.0f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = PER - (DEPTH - 1); e < PER; ++e) a_nxt[e * PSTEP * LDA] = fmaxf(un + ring[e % DEPTH], 0.0f);
    float* red = red_base + buf * (8 * 64);
    if (active) {
      // epilogue: h2 = relu(acc + b2'), partial logit = sum over this wave's 32 features of w3 * h2.
      // acc register g of lane (r, hh) is pair row (g & 3) + 8 (g >> 2) + 4 hh, feature r.
      const float s0 = reduce_features(acc0, bias, w3v, r);
      const float s1 = reduce_features(acc1, bias, w3v, r);
      const int g = r & 15;
      const int prow = (g & 3) + 8 * (g >> 2) + 4 * hh;
      if (r < 16) red[wave * 64 + prow] = s0;
      else red[wave * 64 + 32 + prow] = s1;
    }
    // one barrier per tile: the next h1 tile and this tile's partial logits are complete, and nobody still
    // reads the buffer the next round overwrites (red is double buffered for the same reason)
    __syncthreads();
    if (tid < 64) {
      float s = b3v;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * 64 + tid];
      const int64_t j = jb + tid;
      if (j < j1) {
        if (apply_sigmoid) s = 1.0f / (1.0f + expf(-s));
        out[(i - i0) * ldo + (j - j0)] = s;
      }
    }
    buf ^= 1;
  }
}

// ---- fused: score both directions of every pair, symmetrise, threshold, emit edges --------------------------
// A = (S + S^T)/2 needs S[i][j] and S[j][i] together, so a work item is the PAIR of tiles (i, J) and (J, i): 64
// forward scores S[i][j] (u = U[i], v = V[j]) followed by the 64 backward scores S[j][i] (u = V[i], v = U[j]: the
// same h1 = relu(u + v) generator with the two tables swapped), j in one 64-column block J at or right of i's block.
// Only the upper triangle is walked (both directions of it = the same n^2 pair evaluations as the full matrix);
// thread p < 64 keeps the forward score, averages it with the backward one and, where it survives the threshold,
// appends (i, j, a) and (j, i, a) to an UNORDERED edge list (one atomic slot reservation per tile).  The host sorts
// the list by (row, col) -- the keys are unique, so the final COO is deterministic and in torch.nonzero order --
// and never sees the n x n matrix.  Each score is the same fmaf chain as in k_scorer_pairs: identical bits.
template <int H>
__global__ __launch_bounds__(512) void k_scorer_edges(const float* __restrict__ U, const float* __restrict__ V,
                                                      const float* __restrict__ W2f, const float* __restrict__ b2f,
                                                      const float* __restrict__ w3, const float* __restrict__ b3,
                                                      int64_t n, float thr, int64_t* __restrict__ erow,
                                                      int64_t* __restrict__ ecol, float* __restrict__ eval, int64_t cap,
                                                      unsigned long long* __restrict__ counter) {
  constexpr int HH = H / 2;
  constexpr int LDA = H + 4;
  constexpr int NW = H / 32;
  constexpr int PER = H / 8;
  constexpr int PSTEP = 512 / H;
  constexpr int DEPTH = PER < 8 ? PER : 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red_base = lds + 2 * 64 * LDA;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const bool active = wave < NW;
  float breg[HH];
  float bias = 0.0f, w3v = 0.0f;
  if (active) {
    const float* wrow = W2f + (int64_t)(32 * wave + r) * H + hh * HH;
#pragma unroll
    for (int m = 0; m < HH; m += 4) {
      f4s t = *reinterpret_cast<const f4s*>(wrow + m);
      breg[m] = t[0]; breg[m + 1] = t[1]; breg[m + 2] = t[2]; breg[m + 3] = t[3];
    }
    bias = b2f[32 * wave + r];
    w3v = w3[32 * wave + r];
  }
  const float b3v = b3[0];
  const int64_t B = (n + 63) / 64;                      // 64-node blocks
  const int64_t n_pairs = B * (B + 1) / 2 * 64;         // (block pair bi <= bj) x (row of block bi)
  const int gk = tid % H, pofs = tid / H;
  const int gen_off = pofs * LDA + gk;
  const int mma_off = r * LDA + hh * HH;

  // half-tile `dir` of pair `pi`: which single row supplies u, which 64-row block supplies v
  auto decode = [&](int64_t pi, int dir, int64_t& i, int64_t& jb, const float*& urow, const float*& vblk, int& pmax) {
    const int64_t q = pi >> 6;                          // block pair, row-major over the upper triangle
    // bi = largest b with b * B - b (b - 1) / 2 <= q
    int64_t bi = (int64_t)(((double)(2 * B + 1) - sqrt((double)(2 * B + 1) * (double)(2 * B + 1) - 8.0 * (double)q)) * 0.5);
    if (bi < 0) bi = 0;
    if (bi >= B) bi = B - 1;
    while (bi > 0 && bi * B - bi * (bi - 1) / 2 > q) --bi;
    while (bi + 1 < B && (bi + 1) * B - (bi + 1) * bi / 2 <= q) ++bi;
    const int64_t bj = bi + (q - (bi * B - bi * (bi - 1) / 2));
    i = bi * 64 + (pi & 63);
    jb = bj * 64;
    const int64_t ic = i < n ? i : n - 1;                // rows past the end (last block): computed, never emitted
    urow = (dir == 0 ? U : V) + ic * H;
    vblk = (dir == 0 ? V : U) + jb * H;
    pmax = (int)((n - 1 - jb) < 63 ? (n - 1 - jb) : 63);
  };

  int64_t pi = blockIdx.x;
  if (pi >= n_pairs) return;
  int dir = 0;
  {
    int64_t i, jb; const float *urow, *vblk; int pmax;
    decode(pi, 0, i, jb, urow, vblk, pmax);
    const float u = urow[gk];
    const float* vb = vblk + gk;
#pragma unroll 8
    for (int q = 0; q < PER; ++q) {
      int p = q * PSTEP + pofs;
      const int pc = p < pmax ? p : pmax;
      lds[gen_off + q * PSTEP * LDA] = fmaxf(u + vb[pc * H], 0.0f);
    }
  }
  __syncthreads();
  int buf = 0;
  float fwd = 0.0f;                                      // forward score of pair (i, jb + tid), threads < 64
  while (pi < n_pairs) {
    int64_t i, jb; const float *urow_c, *vblk_c; int pmax_c;
    decode(pi, dir, i, jb, urow_c, vblk_c, pmax_c);
    // the half-tile generated while this one is multiplied (the very last one regenerates itself: harmless)
    int64_t pin = dir == 0 ? pi : pi + gridDim.x;
    const int dirn = dir ^ 1;
    if (pin >= n_pairs) pin = pi;
    int64_t in_, jbn; const float *urow, *vblk; int pmax;
    decode(pin, pin == pi && dir == 1 ? 1 : dirn, in_, jbn, urow, vblk, pmax);
    const float un = urow[gk];
    const float* vb = vblk + gk;
    const float* a_cur = lds + buf * (64 * LDA);
    float* a_nxt = lds + (buf ^ 1) * (64 * LDA) + gen_off;
    float ring[DEPTH];
    f32x16 acc0 = {0}, acc1 = {0};
    const float* a0 = a_cur + mma_off;
    const float* a1 = a0 + 32 * LDA;
    f4s x0 = {0, 0, 0, 0}, x1 = {0, 0, 0, 0};
    if (active) {
      x0 = *reinterpret_cast<const f4s*>(a0);
      x1 = *reinterpret_cast<const f4s*>(a1);
    }
#pragma unroll
    for (int it = 0; it < PER; ++it) {
      {
        const int p = it * PSTEP + pofs;
        const int pc = p < pmax ? p : pmax;
        ring[it % DEPTH] = vb[pc * H];
      }
      if (active) {
        const int m = 4 * it;
        f4s n0 = x0, n1 = x1;
        if (m + 4 < HH) {
          n0 = *reinterpret_cast<const f4s*>(a0 + m + 4);
          n1 = *reinterpret_cast<const f4s*>(a1 + m + 4);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0[q], breg[m + q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1[q], breg[m + q], acc1, 0, 0, 0);
        }
        x0 = n0; x1 = n1;
      }
      if (it >= DEPTH - 1) {
        const int e = it - (DEPTH - 1);
        a_nxt[e * PSTEP * LDA] = fmaxf(un + ring[e % DEPTH], 0.0f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = PER - (DEPTH - 1); e < PER; ++e) a_nxt[e * PSTEP * LDA] = fmaxf(un + ring[e % DEPTH], 0.0f);
    float* red = red_base + buf * (8 * 64);
    if (active) {
      const float s0 = reduce_features(acc0, bias, w3v, r);
      const float s1 = reduce_features(acc1, bias, w3v, r);
      const int g = r & 15;
      const int prow = (g & 3) + 8 * (g >> 2) + 4 * hh;
      if (r < 16) red[wave * 64 + prow] = s0;
      else red[wave * 64 + 32 + prow] = s1;
    }
    __syncthreads();
    if (tid < 64) {
      float s = b3v;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * 64 + tid];
      s = 1.0f / (1.0f + expf(-s));
      if (dir == 0) {
        fwd = s;
      } else {
        const int64_t j = jb + tid;
        const float a = (fwd + s) / 2.0f;                 // (S[i][j] + S[j][i]) / 2, the operand order of the dense path
        const bool keep = i < n && j < n && j > i && !(a < thr) && a != 0.0f;   // `A[A < thr] = 0` then nonzero()
        const unsigned long long m = __ballot(keep);
        if (m) {
          unsigned long long base = 0;
          if (tid == 0) base = atomicAdd(counter, 2ull * (unsigned long long)__popcll(m));
          base = ((unsigned long long)__shfl((int)(base >> 32), 0) << 32) | (unsigned int)__shfl((int)(base & 0xffffffffu), 0);
          if (keep) {
            const int64_t pos = (int64_t)base + 2 * __popcll(m & ((1ull << tid) - 1ull));
            // beyond the capacity nothing is written: the counter still runs, the caller sees the overflow
            if (pos < cap) { erow[pos] = i; ecol[pos] = j; eval[pos] = a; }
            if (pos + 1 < cap) { erow[pos + 1] = j; ecol[pos + 1] = i; eval[pos + 1] = a; }
          }
        }
      }
    }
    buf ^= 1;
    if (dir == 1) pi += gridDim.x;
    dir ^= 1;
  }
}

// ---- post-processing: symmetrise, threshold, diagonal, compaction ------------------------------------
// variant 0 transductive: A = (S + S^T)/2; A[A < thr] = 0; diag = 1        (LargeScaleCondensing.py:306-308)
// variant 1 inductive   : A = (S + S^T)/2; diag = 1; A[A < thr] = 0        (LargeScaleCondensing_induct.py:300-302)
// variant 2 sampled     : off-diagonal as above (both directions averaged, thresholded, symmetric);
//                         the n unit self-loops are APPENDED after the sorted edge list (_Sampled.py:271-285)
__device__ __forceinline__ float sym_value(const float* __restrict__ S, int64_t lds, int64_t i, int64_t j, float thr,
                                           int variant) {
  if (i == j) {
    if (variant == 2) return 0.0f;
    if (variant == 1) return 1.0f < thr ? 0.0f : 1.0f;
    return 1.0f;
  }
  const float a = (S[i * lds + j] + S[j * lds + i]) / 2.0f;
  return a < thr ? 0.0f : a;
}

__global__ __launch_bounds__(256) void k_sym_count(const float* __restrict__ S, int64_t n, int64_t lds, float thr, int variant,
                                                   int64_t* __restrict__ rowcount) {
  const int64_t i = blockIdx.x;
  int cnt = 0;
  for (int64_t j = threadIdx.x; j < n; j += 256) cnt += sym_value(S, lds, i, j, thr, variant) != 0.0f;
  __shared__ int part[256];
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowcount[i] = part[0];
}

// one 64-lane wave per row keeps the columns in ascending order (torch.nonzero order)
__global__ __launch_bounds__(64) void k_sym_fill(const float* __restrict__ S, int64_t n, int64_t lds, float thr, int variant,
                                                 const int64_t* __restrict__ rowstart, int64_t E, int64_t* __restrict__ ei,
                                                 float* __restrict__ ew) {
  const int64_t i = blockIdx.x;
  int64_t pos = rowstart[i];
  const int lane = threadIdx.x;
  for (int64_t jb = 0; jb < n; jb += 64) {
    const int64_t j = jb + lane;
    float a = 0.0f;
    if (j < n) a = sym_value(S, lds, i, j, thr, variant);
    const unsigned long long m = __ballot(a != 0.0f);
    if (a != 0.0f) {
      const int64_t p = pos + __popcll(m & ((1ull << lane) - 1ull));
      ei[p] = i;
      ei[E + p] = j;
      ew[p] = a;
    }
    pos += __popcll(m);
  }
}

}  // namespace disco

using namespace disco;

static int scorer_cus() {      // compute units of the current device (256 on MI355X), asked once
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) cus = v;
    if (cus <= 0) cus = 256;
  }
  return cus;
}


extern "C" size_t disco_pairscore_workspace_bytes(int64_t n, int64_t H) {
  if (n < 0 || H <= 0) return 0;
  return (size_t)scorer_layout(n, H).total;
}

extern "C" int disco_pairscore_prepare_f32(const float* Z, int64_t n, int64_t D, int64_t ldz, int64_t H, const float* W1,
                                           const float* b1, const float* bn1_w, const float* bn1_b, const float* bn1_mean,
                                           const float* bn1_var, const float* W2, const float* b2, const float* bn2_w,
                                           const float* bn2_b, const float* bn2_mean, const float* bn2_var, const float* W3,
                                           const float* b3, float eps, void* ws, size_t ws_bytes, void* stream_) {
  if (n < 0 || D <= 0 || ldz < D) return DISCO_EINVAL;
  if (H != 32 && H != 64 && H != 128 && H != 256) return DISCO_EUNSUPPORTED;
  if (!Z || !W1 || !b1 || !bn1_w || !bn1_b || !bn1_mean || !bn1_var || !W2 || !b2 || !bn2_w || !bn2_b || !bn2_mean ||
      !bn2_var || !W3 || !b3 || !ws)
    return DISCO_EINVAL;
  ScorerWs L = scorer_layout(n, H);
  if ((size_t)L.total > ws_bytes) return DISCO_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  char* w = static_cast<char*>(ws);
  if (n > 0) {
    hipLaunchKernelGGL(k_scorer_uv, dim3((unsigned)((n + kUvTile - 1) / kUvTile), (unsigned)((H + kUvTile - 1) / kUvTile)), dim3(256), 0, s, Z, n, D, ldz, W1, b1, bn1_w, bn1_b,
                       bn1_mean, bn1_var, eps, (int)H, reinterpret_cast<float*>(w + L.off_u), reinterpret_cast<float*>(w + L.off_v));
  }
  hipLaunchKernelGGL(k_scorer_fold, dim3((unsigned)((H * H + 255) / 256)), dim3(256), 0, s, W2, b2, bn2_w, bn2_b, bn2_mean, bn2_var,
                     W3, b3, eps, (int)H, reinterpret_cast<float*>(w + L.off_w2), reinterpret_cast<float*>(w + L.off_b2),
                     reinterpret_cast<float*>(w + L.off_w3), reinterpret_cast<float*>(w + L.off_b3));
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

template <int H>
static int launch_pairs(const char* w, const ScorerWs& L, int64_t i0, int64_t i1, int64_t j0, int64_t j1, float* out, int64_t ldo,
                        int apply_sigmoid, hipStream_t s) {
  const int64_t tiles = (i1 - i0) * ((j1 - j0 + 63) / 64);
  const int64_t resident = (int64_t)scorer_cus() * 4;       // persistent workgroups: four per CU share the tile loop
  int64_t blocks = tiles < resident ? tiles : resident;
  if (blocks < 1) return DISCO_OK;
  const size_t lds_bytes = (size_t)(2 * 64 * (H + 4) + 2 * 8 * 64) * sizeof(float);
  // the dynamic LDS limit is a per-DEVICE attribute of the kernel: raised on every call (microseconds against a kernel
  // that runs for hundreds of milliseconds) rather than remembered in a process-wide flag that a second GPU would skip
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scorer_pairs<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
  }
  hipLaunchKernelGGL((k_scorer_pairs<H>), dim3((unsigned)blocks), dim3(512), lds_bytes, s,
                     reinterpret_cast<const float*>(w + L.off_u), reinterpret_cast<const float*>(w + L.off_v),
                     reinterpret_cast<const float*>(w + L.off_w2), reinterpret_cast<const float*>(w + L.off_b2),
                     reinterpret_cast<const float*>(w + L.off_w3), reinterpret_cast<const float*>(w + L.off_b3), i0, i1, j0, j1, out,
                     ldo, apply_sigmoid);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_pairscore_block_f32(const void* ws, int64_t n, int64_t H, int64_t i0, int64_t i1, int64_t j0, int64_t j1,
                                         float* out, int64_t ldo, int apply_sigmoid, void* stream_) {
  if (!ws || n < 0 || i0 < 0 || j0 < 0 || i1 > n || j1 > n || i1 < i0 || j1 < j0) return DISCO_EINVAL;
  if (i1 == i0 || j1 == j0) return DISCO_OK;
  if (!out || ldo < j1 - j0) return DISCO_EINVAL;
  ScorerWs L = scorer_layout(n, H);
  const char* w = static_cast<const char*>(ws);
  hipStream_t s = static_cast<hipStream_t>(stream_);
  switch (H) {
    case 32: return launch_pairs<32>(w, L, i0, i1, j0, j1, out, ldo, apply_sigmoid, s);
    case 64: return launch_pairs<64>(w, L, i0, i1, j0, j1, out, ldo, apply_sigmoid, s);
    case 128: return launch_pairs<128>(w, L, i0, i1, j0, j1, out, ldo, apply_sigmoid, s);
    case 256: return launch_pairs<256>(w, L, i0, i1, j0, j1, out, ldo, apply_sigmoid, s);
    default: return DISCO_EUNSUPPORTED;
  }
}

extern "C" int disco_sym_threshold_count(const float* S, int64_t n, int64_t lds, float threshold, int variant, int64_t* rowcount,
                                         void* stream_) {
  if (n < 0 || variant < 0 || variant > 2) return DISCO_EINVAL;
  if (n == 0) return DISCO_OK;
  if (!S || !rowcount || lds < n) return DISCO_EINVAL;
  DISCO_GRID_1D(g_cnt, n, 256);
  hipLaunchKernelGGL(k_sym_count, g_cnt, dim3(256), 0, static_cast<hipStream_t>(stream_), S, n, lds, threshold, variant,
                     rowcount);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_sym_threshold_fill(const float* S, int64_t n, int64_t lds, float threshold, int variant,
                                        const int64_t* rowstart, int64_t E, int64_t* edge_index, float* edge_weight, void* stream_) {
  if (n < 0 || E < 0 || variant < 0 || variant > 2) return DISCO_EINVAL;
  if (n == 0 || E == 0) return DISCO_OK;
  if (!S || !rowstart || !edge_index || !edge_weight || lds < n) return DISCO_EINVAL;
  DISCO_GRID_1D(g_fill, n, 64);
  hipLaunchKernelGGL(k_sym_fill, g_fill, dim3(64), 0, static_cast<hipStream_t>(stream_), S, n, lds, threshold, variant,
                     rowstart, E, edge_index, edge_weight);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

template <int H>
static int launch_edges(const char* w, const ScorerWs& L, int64_t n, float thr, int64_t* erow, int64_t* ecol, float* eval,
                        int64_t cap, unsigned long long* counter, hipStream_t s) {
  const int64_t B = (n + 63) / 64;
  const int64_t n_pairs = B * (B + 1) / 2 * 64;
  const int64_t resident = (int64_t)scorer_cus() * 4;
  int64_t blocks = n_pairs < resident ? n_pairs : resident;
  if (blocks < 1) return DISCO_OK;
  const size_t lds_bytes = (size_t)(2 * 64 * (H + 4) + 2 * 8 * 64) * sizeof(float);
  // the dynamic LDS limit is a per-DEVICE attribute of the kernel: raised on every call (microseconds against a kernel
  // that runs for hundreds of milliseconds) rather than remembered in a process-wide flag that a second GPU would skip
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_scorer_edges<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds_bytes);
    if (e != hipSuccess) { set_last_hip_error((int)e); return DISCO_EHIP; }
  }
  hipLaunchKernelGGL((k_scorer_edges<H>), dim3((unsigned)blocks), dim3(512), lds_bytes, s,
                     reinterpret_cast<const float*>(w + L.off_u), reinterpret_cast<const float*>(w + L.off_v),
                     reinterpret_cast<const float*>(w + L.off_w2), reinterpret_cast<const float*>(w + L.off_b2),
                     reinterpret_cast<const float*>(w + L.off_w3), reinterpret_cast<const float*>(w + L.off_b3), n, thr, erow, ecol,
                     eval, cap, counter);
  DISCO_CHECK_LAUNCH();
  return DISCO_OK;
}

extern "C" int disco_pairscore_edges_f32(const void* ws, int64_t n, int64_t H, float threshold, int64_t* edge_row,
                                         int64_t* edge_col, float* edge_val, int64_t capacity, int64_t* count_dev,
                                         void* stream_) {
  if (!ws || n < 0 || capacity < 0 || !count_dev) return DISCO_EINVAL;
  if (capacity > 0 && (!edge_row || !edge_col || !edge_val)) return DISCO_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream_);
  DISCO_HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int64_t), s));
  if (n == 0) return DISCO_OK;
  ScorerWs L = scorer_layout(n, H);
  const char* w = static_cast<const char*>(ws);
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(count_dev);
  switch (H) {
    case 32: return launch_edges<32>(w, L, n, threshold, edge_row, edge_col, edge_val, capacity, counter, s);
    case 64: return launch_edges<64>(w, L, n, threshold, edge_row, edge_col, edge_val, capacity, counter, s);
    case 128: return launch_edges<128>(w, L, n, threshold, edge_row, edge_col, edge_val, capacity, counter, s);
    case 256: return launch_edges<256>(w, L, n, threshold, edge_row, edge_col, edge_val, capacity, counter, s);
    default: return DISCO_EUNSUPPORTED;
  }
}
