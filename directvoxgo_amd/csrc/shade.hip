// Fused colour head ("rgbnet") for gfx950 -- row N3 of SURVEY.md section 8f.
//
// Replaces, for the fine-stage model (/root/reference/lib/dvgo.py:516-541):
//   viewdirs_emb[ray_id] gather, cat([k0_view, viewdirs_emb]), Linear(D_in,W)+ReLU, Linear(W,W)+ReLU,
//   Linear(W,3), (+ k0_diffuse), sigmoid
// i.e. three skinny fp32 GEMMs (M ~ 2 M rows, N <= 128) plus five element-wise passes over [M,W].
//
// This is the one dense contraction of the path, so it is the one place MFMA applies.  It stays in
// fp32 (the reference runs the MLP in fp32; bf16 would break the stated tolerances): gfx950's
// v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain at the vector-FMA rate and needs ONE VGPR per
// operand per lane.
//
// Layout trick (no LDS transposes between layers): every layer is computed TRANSPOSED,
//   H^T[f][row] = sum_k W[f][k] * X^T[k][row],
// with the weights as the MFMA A operand and the activations as the B operand.  The 32x32 result tile
// then has `row` on the lane (col = lane & 31) and the output features on the 16 registers
// (f = (r&3) + 8*(r>>2) + 4*(lane>>5)).  A B operand needs B[k = 2s + (lane>>5)][col = lane & 31]:
// register r of the previous layer's tile already IS such an operand for the feature pair
// {f(r,0), f(r,1)}, as long as the A operand of that k-step is W[f_out][f(r, lane>>5)] -- the k order
// inside a contraction is free.  So the weights are stored in LDS pre-permuted to that order (once
// per workgroup) and the activations never leave their registers.
// Layer 3 (3 outputs) is 192 VALU FMAs per lane plus one cross-half add.
#include "common.h"
#include "x3.h"

#include <type_traits>


static int g_shade_experiment = 0;   // timing experiments only (tools/): bit 0 = skip G1/G2 stores in shade_bwd
// kernel variants (dvgo_shade_variant): bit 0 = forward, bit 1 = data gradients, bit 2 = weight gradients on the bf16
// matrix cores with a 3-way operand split (shade_x3.hip)
static int g_shade_variant = 67;        // include/dvgo_hip.h: dvgo_shade_variant
extern "C" int dvgo_shade_wgrad_x3(const float* G1, const float* gz, const uint64_t* masks, const float* W3, const float* H1,
                                   const float* H2, const float* feat, int C, const float* emb, int E, const int64_t* ray_id,
                                   int64_t M, const int64_t* m_dev, int width, int diffuse, int n_parts, float* part, int form_b,
                                   void* stream);
extern "C" int dvgo_shade_bwd_x3(const float* g_rgb, const float* rgb, const uint64_t* masks, int64_t M, const int64_t* m_dev,
                                 const float* W1, const float* W2, const float* W3, int width, int d_in, int C, int diffuse,
                                 float* g_feat, float* G1, float* gz, void* scratch, int prebuilt, void* stream);

extern "C" int dvgo_shade_fwd_x3(const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                                 const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                                 const float* b3, int width, int d_in, int diffuse, float* rgb, float* H1, float* H2,
                                 uint64_t* masks, void* scratch, void* scratch_bwd, int experiment, void* stream);

// Saved activations / gradients are plain row-major [M, features]: in the accumulator layout a lane owns
// 4 consecutive features per register quad, i.e. one 16-byte piece of its row.
// one workgroup per CU (the permuted weights take 86 KB of LDS); 12 wavefronts = 3 per SIMD (<= 170 registers each):
// a wave alternates MFMA-dense phases with VALU / store phases, and the third wave fills what two leave idle
// (forward 0.956 -> 0.92 ms, data gradients 0.775 -> 0.75 ms against 8 wavefronts)
#define SHADE_THREADS 768
#define SHADE_WAVES (SHADE_THREADS / 64)
#define SHADE_BWD_THREADS 1024      // data-gradient kernel: 16 wavefronts (128 registers each fit): 0.75 -> 0.72 ms
#define SHADE_BWD_WAVES (SHADE_BWD_THREADS / 64)

template <int WIDTH, int S1>
struct ShadeLds {
  static constexpr int T = WIDTH / 32;
  float w1a[T][S1][64];        // [out tile][k-step][lane]   A operand of layer 1
  float w2a[T][T][16][64];     // [out tile][in tile][reg][lane]  A operand of layer 2
  float w3p[3][2][T * 16];     // [c][lane half][in tile*16 + reg]
  float b1p[2][T * 16];        // accumulator-layout biases
  float b2p[2][T * 16];
  float b3[4];
};

__device__ __forceinline__ int acc_feature(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

// Row-major store of one 32(sample) x 32(feature) accumulator tile through a wave-private LDS patch.
// In the accumulator layout a lane owns 16-byte pieces of ITS row, so a direct store touches 32 rows x 32 B per
// wave instruction (measured: 0.20 ms of the training forward, 0.11 ms of the data-gradient kernel).  Staged
// through LDS (row stride 36 floats: conflict-free 16-B writes), 8 lanes cover the 128 contiguous bytes of one
// row and a wave instruction writes 8 full 128-B segments instead.  Same wave writes and reads: no barrier.
#define SHADE_STAGE_STRIDE 36
__device__ __forceinline__ void shade_store_tile(float* __restrict__ stage /* [32][36] of this wave */,
                                                 const f32x16& acc, float* __restrict__ dst /* row 0 of the tile, at
                                                 its feature offset */, int row_stride, int lane, int rows_valid) {
  const int smp = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(stage + smp * SHADE_STAGE_STRIDE + 8 * q + 4 * h) =
        make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  const int c = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (lane >> 3) + 8 * i;
    const float4 v = *reinterpret_cast<const float4*>(stage + r * SHADE_STAGE_STRIDE + 4 * c);
    if (r < rows_valid) *reinterpret_cast<float4*>(dst + (int64_t)r * row_stride + 4 * c) = v;
  }
}

template <int WIDTH, int S1>
__device__ __forceinline__ void shade_load_weights(ShadeLds<WIDTH, S1>& L, const float* __restrict__ W1,
                                                   const float* __restrict__ b1, const float* __restrict__ W2,
                                                   const float* __restrict__ b2, const float* __restrict__ W3,
                                                   const float* __restrict__ b3, int D_in) {
  constexpr int T = WIDTH / 32;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < T * S1 * 64; i += nt) {
    const int l = i & 63, s = (i >> 6) % S1, t = (i >> 6) / S1;
    const int k = 2 * s + (l >> 5);
    (&L.w1a[0][0][0])[i] = (k < D_in) ? W1[(32 * t + (l & 31)) * D_in + k] : 0.0f;
  }
  for (int i = tid; i < T * T * 16 * 64; i += nt) {
    const int l = i & 63, r = (i >> 6) & 15, t = (i >> 10) % T, t2 = (i >> 10) / T;
    (&L.w2a[0][0][0][0])[i] = W2[(32 * t2 + (l & 31)) * WIDTH + acc_feature(t, r, l >> 5)];
  }
  for (int i = tid; i < 3 * 2 * T * 16; i += nt) {
    const int tr = i % (T * 16), h = (i / (T * 16)) & 1, c = i / (2 * T * 16);
    (&L.w3p[0][0][0])[i] = W3[c * WIDTH + acc_feature(tr >> 4, tr & 15, h)];
  }
  for (int i = tid; i < 2 * T * 16; i += nt) {
    const int tr = i % (T * 16), h = i / (T * 16);
    const int f = acc_feature(tr >> 4, tr & 15, h);
    (&L.b1p[0][0])[i] = b1[f];
    (&L.b2p[0][0])[i] = b2[f];
  }
  if (tid < 3) L.b3[tid] = b3[tid];
}

// forward of one 32-row tile.  Layer 1 keeps all WIDTH features (T accumulator tiles) because every
// layer-2 output needs them; layer 2 is walked one 32-feature output tile at a time and each tile is
// consumed immediately (optional H2 store + its share of the three layer-3 dot products), so only
// T + 1 accumulator tiles are live and two waves fit on a SIMD.
// Returns the three logits (before the optional diffuse term and the sigmoid) in z[3], valid on every
// lane; acc1 holds the post-ReLU layer-1 activations.
template <int WIDTH, int S1>
__device__ __forceinline__ void shade_tile_forward(const ShadeLds<WIDTH, S1>& L, const float (&x)[S1], int lane,
                                                   f32x16 (&acc1)[WIDTH / 32], float (&z)[3],
                                                   float* __restrict__ H2tile /* H2 + tile_row0*WIDTH or nullptr */,
                                                   float* __restrict__ stage, int rows_valid,
                                                   unsigned long long& mask2 /* bit 16*t + r = (H2 feature f(t,r,h) > 0) */) {
  constexpr int T = WIDTH / 32;
  const int h = lane >> 5;
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = L.b1p[h][t * 16 + r];
#pragma unroll
    for (int s = 0; s < S1; ++s) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(L.w1a[t][s][lane], x[s], acc1[t], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[t][r] = fmaxf(acc1[t][r], 0.0f);
    __builtin_amdgcn_sched_barrier(0);       // one tile's A-operand reads at a time (three waves per SIMD: 170 registers)
  }
  float p[3] = {0.0f, 0.0f, 0.0f};
  mask2 = 0ull;
#pragma unroll 1      // keeps the scheduler from hoisting every tile's LDS reads (and spilling)
  for (int t2 = 0; t2 < T; ++t2) {
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = L.b2p[h][t2 * 16 + r];
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(L.w2a[t2][t][r][lane], acc1[t][r], acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = fmaxf(acc2[r], 0.0f);
    if (H2tile != nullptr) {
      unsigned int bits = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) bits |= (acc2[r] > 0.0f ? 1u : 0u) << r;
      mask2 |= (unsigned long long)bits << (16 * t2);
      shade_store_tile(stage, acc2, H2tile + 32 * t2, WIDTH, lane, rows_valid);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) p[c] = fmaf(L.w3p[c][h][t2 * 16 + r], acc2[r], p[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) z[c] = p[c] + __shfl_xor(p[c], 32) + L.b3[c];
}

// X^T operand of layer 1 for this lane: x[s] = input feature k = 2s + (lane>>5) of row (lane & 31):
//   k < n_view           : feat[row, c_view0 + k]          (k0_view,  lib/dvgo.py:518-523)
//   k < n_view + E       : emb[ray_id[row], k - n_view]    (viewdirs_emb[ray_id], lib/dvgo.py:524-526)
template <int S1>
__device__ __forceinline__ void shade_load_x(const float* __restrict__ feat, int C, int c_view0, int n_view,
                                             const float* __restrict__ emb, int E, int64_t row, int64_t ray,
                                             int lane, float (&x)[S1]) {
  const int h = lane >> 5;
  const float* fr = feat + row * C + c_view0;
  const float* er = emb + ray * E - n_view;
#pragma unroll
  for (int s = 0; s < S1; ++s) {
    const int k = 2 * s + h;
    float v = 0.0f;
    if (k < n_view) v = fr[k];
    else if (k < n_view + E) v = er[k];
    x[s] = v;
  }
}

template <int WIDTH, int S1, bool DIFFUSE>
__global__ void __launch_bounds__(SHADE_THREADS)
shade_fwd_kernel(const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                 const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev, const float* __restrict__ W1,
                 const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                 const float* __restrict__ W3, const float* __restrict__ b3, int D_in, float* __restrict__ rgb,
                 float* __restrict__ H1, float* __restrict__ H2, unsigned long long* __restrict__ masks, int experiment) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  __shared__ ShadeLds<WIDTH, S1> L;
  __shared__ __attribute__((aligned(16))) float s_stage[SHADE_WAVES][32 * SHADE_STAGE_STRIDE];
  shade_load_weights<WIDTH, S1>(L, W1, b1, W2, b2, W3, b3, D_in);
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  float* stage = s_stage[threadIdx.x >> 6];
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = gw; tile < n_tiles; tile += nw) {
    const int64_t row = tile * 32 + (lane & 31);
    const bool valid = row < M;
    const int64_t rowc = valid ? row : (M - 1);
    float x[S1];
    if (experiment & 16) { for (int s = 0; s < S1; ++s) x[s] = 0.01f * (float)(lane + s); }
    else shade_load_x<S1>(feat, C, c_view0, n_view, emb, E, rowc, ray_id[rowc], lane, x);
    f32x16 acc1[T];
    float z[3];
    const int rows_valid = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
    unsigned long long mask2;
    shade_tile_forward<WIDTH, S1>(L, x, lane, acc1, z, (H1 != nullptr && !(experiment & 32)) ? (H2 + tile * 32 * WIDTH) : nullptr,
                                  stage, rows_valid, mask2);
    if (valid) {
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float zz = DIFFUSE ? z[c] + feat[row * C + c] : z[c];
          rgb[row * 3 + c] = 1.0f / (1.0f + expf(-zz));
        }
      }
      if (H1 != nullptr) {     // training: keep the post-ReLU activations (weight gradients) and their
        unsigned long long mask1 = 0ull;   // sign bits (ReLU masks of the data-gradient kernel: 32 B / row)
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mask1 |= (unsigned long long)(acc1[t][r] > 0.0f ? 1u : 0u) << (16 * t + r);
        }
        masks[(row * 2 + 0) * 2 + h] = mask1;
        masks[(row * 2 + 1) * 2 + h] = mask2;
      }
    }
    if (H1 != nullptr && !(experiment & 32)) {     // whole wave: the staged store is cooperative
#pragma unroll
      for (int t = 0; t < T; ++t) shade_store_tile(stage, acc1[t], H1 + tile * 32 * WIDTH + 32 * t, WIDTH, lane, rows_valid);
    }
  }
}

// ----------------------------------------------------------------------------------
// Backward, data-gradient part.  Per 32-row tile (activations H1/H2 saved by the training forward):
//   gz  = g_rgb * rgb * (1 - rgb)                                  (sigmoid')
//   G2  = (H2 > 0) * (W3^T gz)                   VALU, accumulator layout
//   G1  = (H1 > 0) * (W2^T G2)                   MFMA, A = W2^T pre-permuted so that G2's registers are
//                                                the B operands (same trick as the forward)
//   gx  = W1[:, :32]^T G1                        MFMA, only the first 32 input features are produced
//                                                (the feature-grid part; the view embedding needs no grad)
// G1 and gz are written out for the weight gradients (dvgo_shade_wgrad); G2 is NOT: it is three FMAs and a
// mask bit away from gz, so the weight-gradient kernel rebuilds it (same expression, same bits) instead of
// paying 512 B/sample of store here and 512 B/sample of load there.  g_feat gets gz (diffuse) and gx.
// ----------------------------------------------------------------------------------
template <int WIDTH>
struct ShadeBwdLds {
  static constexpr int T = WIDTH / 32;
  float w2ta[T][T][16][64];    // [in tile][out tile][reg][lane] = W2[f_out(t2,r,lane>>5)][32*t_in + (lane&31)]
  float w1ta[T][16][64];       // [tile][reg][lane]              = W1[f(t,r,lane>>5)][lane&31]   (k < D_in else 0)
  float w3p[3][2][T * 16];
};

template <int WIDTH, bool DIFFUSE>
__global__ void __launch_bounds__(SHADE_BWD_THREADS)
shade_bwd_kernel(const float* __restrict__ g_rgb, const float* __restrict__ rgb,
                 const unsigned long long* __restrict__ masks, int64_t M_cap, const int64_t* __restrict__ m_dev,
                 const float* __restrict__ W1, const float* __restrict__ W2,
                 const float* __restrict__ W3, int D_in, int C, int c_view0, int n_view,
                 float* __restrict__ g_feat, float* __restrict__ G1, float* __restrict__ gz_out, int experiment) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  __shared__ ShadeBwdLds<WIDTH> L;
  __shared__ __attribute__((aligned(16))) float s_stage[SHADE_BWD_WAVES][32 * SHADE_STAGE_STRIDE];
  float* stage = s_stage[threadIdx.x >> 6];
  {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i = tid; i < T * T * 16 * 64; i += nt) {
      const int l = i & 63, r = (i >> 6) & 15, t2 = (i >> 10) % T, tin = (i >> 10) / T;
      (&L.w2ta[0][0][0][0])[i] = W2[acc_feature(t2, r, l >> 5) * WIDTH + 32 * tin + (l & 31)];
    }
    for (int i = tid; i < T * 16 * 64; i += nt) {
      const int l = i & 63, r = (i >> 6) & 15, t = i >> 10;
      const int k = l & 31;
      (&L.w1ta[0][0][0])[i] = (k < D_in) ? W1[acc_feature(t, r, l >> 5) * D_in + k] : 0.0f;
    }
    for (int i = tid; i < 3 * 2 * T * 16; i += nt) {
      const int tr = i % (T * 16), h = (i / (T * 16)) & 1, c = i / (2 * T * 16);
      (&L.w3p[0][0][0])[i] = W3[c * WIDTH + acc_feature(tr >> 4, tr & 15, h)];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = gw; tile < n_tiles; tile += nw) {
    const int64_t row = tile * 32 + (lane & 31);
    const bool valid = row < M;
    const int64_t rowc = valid ? row : (M - 1);
    float gz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float o = rgb[rowc * 3 + c];
      gz[c] = valid ? g_rgb[rowc * 3 + c] * o * (1.0f - o) : 0.0f;
    }
    if (valid && h == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gz_out[row * 3 + c] = gz[c];
        if (DIFFUSE) g_feat[row * C + c] = gz[c];
      }
    }
    // ReLU sign bits of this lane's 64 features per layer (bit 16*t + r <-> feature f(t,r,h))
    const unsigned long long m1 = masks[(rowc * 2 + 0) * 2 + h];
    const unsigned long long m2 = masks[(rowc * 2 + 1) * 2 + h];
    // G2 in accumulator layout (all T tiles stay live: they are the B operands below)
    f32x16 g2[T];
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaf(L.w3p[2][h][t2 * 16 + r], gz[2],
                             fmaf(L.w3p[1][h][t2 * 16 + r], gz[1], L.w3p[0][h][t2 * 16 + r] * gz[0]));
        g2[t2][r] = ((m2 >> (16 * t2 + r)) & 1ull) ? v : 0.0f;   // gz == 0 on rows past M, so G2 == 0 there
      }
    }
    f32x16 gx;
#pragma unroll
    for (int r = 0; r < 16; ++r) gx[r] = 0.0f;
#pragma unroll 1
    for (int tin = 0; tin < T; ++tin) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int t2 = 0; t2 < T; ++t2) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(L.w2ta[tin][t2][r][lane], g2[t2][r], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);     // keeps the A-operand reads of later groups out of the register file
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int mb = (unsigned int)(m1 >> (16 * tin + 4 * q)) & 15u;
        acc[4 * q + 0] = (mb & 1u) ? acc[4 * q + 0] : 0.0f;
        acc[4 * q + 1] = (mb & 2u) ? acc[4 * q + 1] : 0.0f;
        acc[4 * q + 2] = (mb & 4u) ? acc[4 * q + 2] : 0.0f;
        acc[4 * q + 3] = (mb & 8u) ? acc[4 * q + 3] : 0.0f;
      }
      if (!(experiment & 1))
        shade_store_tile(stage, acc, G1 + tile * 32 * WIDTH + 32 * tin, WIDTH, lane,
                         (int)(M - tile * 32 < 32 ? M - tile * 32 : 32));
#pragma unroll
      for (int r = 0; r < 16; ++r)
        gx = __builtin_amdgcn_mfma_f32_32x32x2f32(L.w1ta[tin][r][lane], acc[r], gx, 0, 0, 0);
    }
    if (valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < n_view) g_feat[row * C + c_view0 + k] = gx[r];
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// Backward, weight-gradient part:  dW2 = G2^T H1,  dW1 = G1^T X,  dW3 = gz^T H2,  db = column sums.
// G2[row][f] = bit(mask2[row], f) ? W3[:, f] . gz[row] : 0 is rebuilt per operand from gz and the layer-2 sign
// bits (3 FMAs, the expression of shade_bwd_kernel), so only G1, H1, H2 are streamed.
// All operands are row-major [M, features], and a contraction over ROWS wants exactly that:
//   A[i = out feature][k = row]  : lane (i, h) reads G[row = 2s + h][f0 + i]
//   B[k = row][j = in feature]   : lane (j, h) reads H[row = 2s + h][f0' + j]
// i.e. consecutive lanes read consecutive floats of one row: no transpose anywhere.  Each 32-row tile
// is copied once into LDS by LDS-DMA, one tile ahead (double buffer), and the MFMA operands are
// conflict-free ds_read_b32.
// The 4 waves of a workgroup walk the SAME 32-row tiles and split the output features between them
// (wave w owns out-feature tile w), which keeps every wave's persistent accumulators at 80 registers
// (dW2: 4 tiles, dW1: 1 tile; dW3 and the <= 8 trailing input columns of dW1 are a few VALU FMAs per
// row instead of mostly-empty 32-wide MFMA tiles).
// Each workgroup writes its partial sums to part[blockIdx]; the caller sums over workgroups.
// ----------------------------------------------------------------------------------
template <int WIDTH>
struct ShadeWgradLds {
  // single-buffered 32-row tiles (74 KB): two workgroups share a CU and cover each other's load phase
  float g1[1][32][WIDTH], h1[1][32][WIDTH], h2[1][32][WIDTH];
  float x[1][32][40];
  float gz[1][32][4];
  unsigned int m2[1][32][2][2];     // layer-2 sign bits [row][lane half of the forward][32-bit half]
};

typedef const __attribute__((address_space(1))) void* dvgo_gptr_t;
typedef __attribute__((address_space(3))) void* dvgo_lptr_t;

template <int WIDTH>
__global__ void __launch_bounds__(2 * WIDTH) __attribute__((amdgpu_num_vgpr(104)))   // one wavefront per 32-wide out-feature tile; 208 registers
shade_wgrad_kernel(const float* __restrict__ G1, const float* __restrict__ gz, const unsigned int* __restrict__ masks,
                   const float* __restrict__ W3, const float* __restrict__ H1, const float* __restrict__ H2, const float* __restrict__ feat, int C,
                   int c_view0, int n_view, const float* __restrict__ emb, int E, const int64_t* __restrict__ ray_id,
                   int64_t M_cap, const int64_t* __restrict__ m_dev,
                   float* __restrict__ part /* [gridDim][WIDTH*WIDTH + WIDTH*64 + 32*WIDTH + 3*WIDTH] */) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  static_assert(T == 4 || T == 2, "widths 128 and 64");
  constexpr int TPR = 2 * T;                 // staging threads per X row (32 rows over the 64*T threads)
  constexpr int NXI = 40 / TPR;              // X columns per staging thread (40 columns: d_in <= 40)
  constexpr int LPR = WIDTH / 4;             // lanes per operand row in a DMA instruction (16 B per lane)
  constexpr int RPI = 64 / LPR;              // rows per DMA wave instruction (1 KB)
  __shared__ __attribute__((aligned(16))) ShadeWgradLds<WIDTH> L;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31, w = tid >> 6;
  f32x16 aW2[T], aW1;
  float vW3[3] = {0.0f, 0.0f, 0.0f};     // dW3[c][32w + j]: 3 outputs only -> VALU, not a 32-wide MFMA tile
  float vW1[8];                            // dW1[32w + j][32 + kk], kk < d_in - 32 (<= 8 columns) -> VALU
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) aW1[r] = 0.0f;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) vW1[kk] = 0.0f;
  float sb1 = 0.0f, sb2 = 0.0f, gz_acc = 0.0f;   // bias gradients: column sums of the A operands / of gz
  // this lane's out feature f = 32w + j of layer 2: its W3 column, and where its sign bit lives in the forward's
  // accumulator-order masks (f = 32t + (r&3) + 8(r>>2) + 4h'  ->  word h', bit 16t + r)
  const float w30 = W3[32 * w + j], w31 = W3[WIDTH + 32 * w + j], w32 = W3[2 * WIDTH + 32 * w + j];
  const int m_half = (j >> 2) & 1, m_word = w >> 1, m_bit = 16 * (w & 1) + (j & 3) + 4 * (j >> 3);
  const int d_in = n_view + E;
  const int64_t n_tiles = (M + 31) / 32;

  // The four 32 x WIDTH operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging
  // registers, 1 KB = two 512-B rows per wave instruction, LDS image == the row-major global image), one
  // tile at a time; the second workgroup resident on the CU computes meanwhile.  Rows past M are read from a
  // clamped row and masked at operand read.
  // The small assembled X tile and gz travel through registers.
  float px[NXI], pgz;
  unsigned int pm2;
  bool pvalid, pgvalid;
  int64_t ray_nx;
  const int xrow = tid / TPR, xcol = tid - xrow * TPR;
  {
    const int64_t row = (int64_t)blockIdx.x * 32 + xrow;
    ray_nx = ray_id[row < M ? row : M - 1];
  }
#define SHADE_WGRAD_DMA(TILE, BUF)                                                                              \
  {                                                                                                             \
    const int64_t r0_ = (TILE) * 32;                                                                            \
    {                                                                                                           \
      const int64_t row = r0_ + xrow;                                                                           \
      const int64_t rc = row < M ? row : M - 1;                                                                 \
      const float* fr = feat + rc * C + c_view0;                                                                \
      const float* er = emb + ray_nx * E - n_view;      /* ray_nx was fetched one tile earlier */               \
      _Pragma("unroll") for (int i = 0; i < NXI; ++i) { /* columns 0..39; d_in <= 40 */                         \
        const int k = xcol + TPR * i;                                                                           \
        const int kc = k < d_in ? k : d_in - 1;                                                                 \
        px[i] = *((kc < n_view) ? fr + kc : er + kc);                                                           \
      }                                                                                                         \
      pvalid = row < M;                                                                                         \
      const int64_t row2 = r0_ + (int64_t)gridDim.x * 32 + xrow;                                                \
      ray_nx = ray_id[row2 < M ? row2 : M - 1];         /* for the next tile of this workgroup */               \
    }                                                                                                           \
    {                                                                                                           \
      const int64_t row = r0_ + ((tid & 127) >> 2);                                                             \
      const int64_t rc = row < M ? row : M - 1;                                                                 \
      pgz = gz[rc * 3 + ((tid & 3) < 3 ? (tid & 3) : 0)];                                                       \
      pgvalid = (tid & 3) < 3 && row < M;                                                                       \
      /* masks [M][2 layers][2 halves] u64 = [M][8] u32: layer 2 = words 4..7 (values past M are never used: gz == 0) */ \
      pm2 = masks[rc * 8 + 4 + (tid & 3)];                                                                      \
    }                                                                                                           \
    /* the bulk DMA goes last: the (in-order) wait for ray_nx above then costs nothing */                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {     /* 32 / RPI instructions per tile, 4 per wave */        \
      const int rl = (4 * w + i) * RPI;                 /* this wave instruction covers rows rl .. rl + RPI - 1 */ \
      const int64_t row = r0_ + rl + lane / LPR;                                                                \
      const int64_t off = (row < M ? row : M - 1) * WIDTH + 4 * (lane % LPR);                                   \
      __builtin_amdgcn_global_load_lds((dvgo_gptr_t)(G1 + off), (dvgo_lptr_t)&L.g1[BUF][rl][0], 16, 0, 0);     \
      __builtin_amdgcn_global_load_lds((dvgo_gptr_t)(H1 + off), (dvgo_lptr_t)&L.h1[BUF][rl][0], 16, 0, 0);     \
      __builtin_amdgcn_global_load_lds((dvgo_gptr_t)(H2 + off), (dvgo_lptr_t)&L.h2[BUF][rl][0], 16, 0, 0);     \
    }                                                                                                           \
  }

  constexpr int buf = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    SHADE_WGRAD_DMA(tile, 0);
    // the small loads above are unconditional and all in flight together (one round trip); masked here
    asm volatile("" : "+v"(px[0]), "+v"(px[1]), "+v"(px[2]), "+v"(px[3]), "+v"(px[4]), "+v"(pgz), "+v"(pm2));
    if (NXI > 5) asm volatile("" : "+v"(px[NXI - 5]), "+v"(px[NXI - 4]), "+v"(px[NXI - 3]), "+v"(px[NXI - 2]), "+v"(px[NXI - 1]));
#pragma unroll
    for (int i = 0; i < NXI; ++i) L.x[buf][xrow][xcol + TPR * i] = (pvalid && xcol + TPR * i < d_in) ? px[i] : 0.0f;
    if (tid < 128) {
      gz_acc += pgvalid ? pgz : 0.0f;                     // db3[c] = sum of gz[:, c]: thread (row, c) of the staging
      L.gz[buf][tid >> 2][tid & 3] = pgvalid ? pgz : 0.0f;
      (&L.m2[buf][tid >> 2][0][0])[tid & 3] = pm2;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the DMA has landed (the compiler does not
    __syncthreads();                      // track LDS-DMA completion for us); X / gz / masks committed
    const int rows_valid = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);      // wave-uniform
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
      const int row = 2 * s + h;
      const float4 gzr = *reinterpret_cast<const float4*>(&L.gz[buf][row][0]);      // broadcast; 0 on rows past M
      const float g2v = fmaf(w32, gzr.z, fmaf(w31, gzr.y, w30 * gzr.x));
      const float a2 = ((L.m2[buf][row][m_half][m_word] >> m_bit) & 1u) ? g2v : 0.0f;
      const float a1 = row < rows_valid ? L.g1[buf][row][32 * w + j] : 0.0f;   // rows past M hold a clamped copy
      sb2 += a2; sb1 += a1;
#pragma unroll
      for (int t = 0; t < T; ++t)
        aW2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, L.h1[buf][row][32 * t + j], aW2[t], 0, 0, 0);
      aW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, L.x[buf][row][j], aW1, 0, 0, 0);
      const float4 xa = *reinterpret_cast<const float4*>(&L.x[buf][row][32]);       // broadcast; zero past d_in
      const float4 xb = *reinterpret_cast<const float4*>(&L.x[buf][row][36]);
      vW1[0] = fmaf(a1, xa.x, vW1[0]); vW1[1] = fmaf(a1, xa.y, vW1[1]); vW1[2] = fmaf(a1, xa.z, vW1[2]);
      vW1[3] = fmaf(a1, xa.w, vW1[3]); vW1[4] = fmaf(a1, xb.x, vW1[4]); vW1[5] = fmaf(a1, xb.y, vW1[5]);
      vW1[6] = fmaf(a1, xb.z, vW1[6]); vW1[7] = fmaf(a1, xb.w, vW1[7]);
      const float hv = L.h2[buf][row][32 * w + j];
      vW3[0] = fmaf(gzr.x, hv, vW3[0]); vW3[1] = fmaf(gzr.y, hv, vW3[1]); vW3[2] = fmaf(gzr.z, hv, vW3[2]);
    }
    __syncthreads();      // every wave is done with the tile before the next DMA overwrites it
  }
#undef SHADE_WGRAD_DMA
  // D[row = (r&3) + 8*(r>>2) + 4*h][col = j]
  float* p = part + (int64_t)blockIdx.x * (WIDTH * WIDTH + WIDTH * 64 + 32 * WIDTH + 3 * WIDTH);
  float* pW2 = p;                          // [WIDTH out][WIDTH in]
  float* pW1 = pW2 + WIDTH * WIDTH;        // [WIDTH out][64]
  float* pW3 = pW1 + WIDTH * 64;           // [32 (c padded)][WIDTH], rows 0..2 written
  float* pb = pW3 + 32 * WIDTH;            // [3][WIDTH]: db1, db2, db3 (first 3 entries)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
    for (int t = 0; t < T; ++t) pW2[(32 * w + i) * WIDTH + 32 * t + j] = aW2[t][r];
    pW1[(32 * w + i) * 64 + j] = aW1[r];
  }
  // VALU parts: lanes j and j+32 hold the even / odd rows
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const float v = vW1[kk] + __shfl_xor(vW1[kk], 32);
    if (h == 0) pW1[(32 * w + j) * 64 + 32 + kk] = v;
  }
  if (h == 0) {
#pragma unroll
    for (int kk = 8; kk < 32; ++kk) pW1[(32 * w + j) * 64 + 32 + kk] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = vW3[c] + __shfl_xor(vW3[c], 32);
    if (h == 0) pW3[c * WIDTH + 32 * w + j] = v;
  }
  // db3[c]: the staging threads 4*row + c (waves 0 and 1, 16 rows each) hold per-row sums; fold the row bits
  float gz_sum = gz_acc;
  gz_sum += __shfl_xor(gz_sum, 4); gz_sum += __shfl_xor(gz_sum, 8);
  gz_sum += __shfl_xor(gz_sum, 16); gz_sum += __shfl_xor(gz_sum, 32);
  sb1 += __shfl_xor(sb1, 32); sb2 += __shfl_xor(sb2, 32);
  if (h == 0) {
    pb[32 * w + j] = sb1;
    pb[WIDTH + 32 * w + j] = sb2;
    if (j < 8) pb[2 * WIDTH + 8 * w + j] = (w < 2 && j < 3) ? gz_sum : 0.0f;    // db3 = entries [0,3) + [8,11)
    if (T == 2 && j >= 16) pb[2 * WIDTH + 16 * w + j] = 0.0f;                   // (rest of the db3 record: keep it defined)
  }
}

// ----------------------------------------------------------------------------------
// The same contraction, software-pipelined: two workgroups per CU, each walking 16-row tiles through a double buffer.
// Everything a tile needs -- the three operand tiles, the view-dependent feature columns, the ray's view embedding
// (gathered 4 bytes per lane), gz and the layer-2 sign words -- travels global -> LDS by LDS-DMA through buffer
// descriptors (lane-constant offsets, the tile as the scalar offset, rows past M arrive as zeros), issued one tile ahead
// of its use; the ray ids the embedding gather needs at issue time travel the same way two tiles further ahead, so no
// load with a register destination is in flight inside the loop.  One bare barrier per tile.
// What bounds it (profiles/r2, s_memtime stamps): one wave per SIMD issues in order and does not issue its own VALU /
// LDS work under its own MFMAs, so a k-step costs its five MFMAs (74 cycles each at the sustained clock,
// tools/micro/mfma_shape) PLUS its ~30 other instructions; the second workgroup's wave on the SIMD fills those slots
// and the tile-boundary bubble (barrier, DMA issue, first operand read).
// ----------------------------------------------------------------------------------
// LDS reads the compiler must not see: hipcc orders every ds_read it knows of behind ALL LDS-DMA in flight (vmcnt(0)),
// i.e. behind the groups that are meant to stay in flight.  Address = byte offset into LDS, OFF = immediate offset.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
template <int OFF> __device__ __forceinline__ float lds_f32(unsigned a) {
  float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ unsigned lds_u32(unsigned a) {
  unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
typedef float dvgo_f32x4 __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ dvgo_f32x4 lds_f32x4(unsigned a) {
  dvgo_f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int WIDTH>
struct ShadeWgradRing {
  static constexpr int NB = 2, NR = 4, TR = 16;
  float g1[NB][TR][WIDTH], h1[NB][TR][WIDTH], h2[NB][TR][WIDTH];
  float xf[NB][TR][16];                // the view-dependent feature columns (n_view <= 16), zero-padded
  float xe[NB][TR][32];                // the ray's view embedding (E <= 32), zero-padded
  float gz[NB][TR][4];                 // [row][0..2], [3] = 0
  unsigned int m2[NB][TR][4];          // layer-2 sign words [lane half][32-bit half]
  unsigned int rid[NR][64];            // [row] ray of the tile's rows (first TR words; one DMA instruction writes 64)
};

// the operands of one k-step (rows 2s, 2s + 1 of a tile) as this lane reads them
template <int T>
struct WgradOperands {
  dvgo_f32x4 gq, xa, xb;
  unsigned mw;
  float g1v, h1v[T], xv, h2v;
};

#define DVGO_OOB 0x80000000u           // a byte offset past every buffer below: the DMA then writes zeros

template <int WIDTH>
__global__ void __launch_bounds__(2 * WIDTH, 2)      // two workgroups per CU: at most 256 registers per lane
shade_wgrad_ring_kernel(const float* __restrict__ G1, const float* __restrict__ gz, const unsigned int* __restrict__ masks,
                        const float* __restrict__ W3, const float* __restrict__ H1, const float* __restrict__ H2,
                        const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                        const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev,
                        float* __restrict__ part /* [gridDim][WIDTH*WIDTH + WIDTH*64 + 32*WIDTH + 3*WIDTH] */,
                        float* __restrict__ total, int total_size /* zeroed here for the reduce kernel's atomics */) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_size; i += gridDim.x * blockDim.x) total[i] = 0.0f;   // buffer descriptors are device-only types: the host pass needs the launch stub only
  // (everything the DMA descriptors, scalar offsets and LDS targets are built from must be PROVABLY wave-uniform, or the
  // compiler wraps each DMA in a readfirstlane loop: the row count comes from memory, the wave index from threadIdx)
  int64_t M;
  {
    const int64_t m = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;
    M = ((int64_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }
  using Ring = ShadeWgradRing<WIDTH>;
  constexpr int T = WIDTH / 32, NW = T, NB = Ring::NB, NR = Ring::NR, TR = Ring::TR;      // NW waves
  static_assert(T == 4 || T == 2, "widths 128 and 64");
  constexpr int LPR = WIDTH / 4;             // lanes per operand row in a 16-byte DMA instruction
  constexpr int RPI = 64 / LPR;              // rows per such instruction (1 KB)
  constexpr int IPW = TR / RPI / NW;         // bulk instructions per wave, operand and tile
  constexpr int FPW = 4 / NW, EPW = 8 / NW;  // feature / embedding gather instructions per wave and tile
  static_assert(IPW == 2 && NB == 2, "double buffer: the wait at the top of a tile is vmcnt(0)");
  __shared__ __attribute__((aligned(16))) Ring L;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 aW2[T], aW1;
  float vW3[3] = {0.0f, 0.0f, 0.0f}, vW1[8], sgz[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) aW1[r] = 0.0f;
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) vW1[kk] = 0.0f;
  float sb1 = 0.0f, sb2 = 0.0f;
  const float w30 = W3[32 * w + j], w31 = W3[WIDTH + 32 * w + j], w32 = W3[2 * WIDTH + 32 * w + j];
  const int m_idx = 2 * ((j >> 2) & 1) + (w >> 1), m_bit = 16 * (w & 1) + (j & 3) + 4 * (j >> 3);
  const int d_in = n_view + E;
  const int n_tiles = (int)((M + TR - 1) / TR);        // 32-bit scalar arithmetic from here on (M * WIDTH * 4 < 2^31)
  auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };

  // Buffer descriptors sized by the M valid rows: a row past M (the tail of the last tile, the run-ahead tiles past the
  // end) is out of range and arrives as zeros -- no clamping, no masking at operand read, and the operation count per
  // group never varies.  Per-lane byte offsets are constants of the lane; the tile enters as the scalar offset.
  const unsigned rows_b = (unsigned)(M * WIDTH * 4);
  const auto bG1 = __builtin_amdgcn_make_buffer_rsrc((void*)G1, 0, rows_b, 0x00020000);
  const auto bH1 = __builtin_amdgcn_make_buffer_rsrc((void*)H1, 0, rows_b, 0x00020000);
  const auto bH2 = __builtin_amdgcn_make_buffer_rsrc((void*)H2, 0, rows_b, 0x00020000);
  const auto bF = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (unsigned)(M * C * 4), 0x00020000);
  const auto bE = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, DVGO_OOB, 0x00020000);     // ray count not known here
  const auto bGz = __builtin_amdgcn_make_buffer_rsrc((void*)gz, 0, (unsigned)(M * 12), 0x00020000);
  const auto bM = __builtin_amdgcn_make_buffer_rsrc((void*)masks, 0, (unsigned)(M * 32), 0x00020000);
  const auto bR = __builtin_amdgcn_make_buffer_rsrc((void*)ray_id, 0, (unsigned)(M * 8), 0x00020000);
  unsigned vo_bulk[IPW], vo_f[FPW], vo_ecol[EPW];
  int e_row[EPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) vo_bulk[i] = (unsigned)(((IPW * w + i) * RPI + lane / LPR) * WIDTH * 4 + 16 * (lane % LPR));
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int e = 64 * (w + NW * i) + lane, row = e >> 4, col = e & 15;
    vo_f[i] = col < n_view ? (unsigned)((row * C + c_view0 + col) * 4) : DVGO_OOB;
  }
#pragma unroll
  for (int i = 0; i < EPW; ++i) {
    const int e = 64 * (w + NW * i) + lane, col = e & 31;
    e_row[i] = e >> 5;
    vo_ecol[i] = col < E ? (unsigned)(col * 4) : DVGO_OOB;
  }
  const unsigned vo_gz = (lane & 3) < 3 ? (unsigned)(((lane >> 2) * 3 + (lane & 3)) * 4) : DVGO_OOB;
  const unsigned vo_m2 = (unsigned)(((lane >> 2) * 8 + 4 + (lane & 3)) * 4);
  const unsigned vo_rid = (unsigned)((lane & (TR - 1)) * 8);                         // rays < 2^31: the low words

  // group(t) = the DMA of tile t (its embedding gather reads the tile's ray ids from LDS) + the ray ids of tile t + 2
  auto issue = [&](int t) {
    const int sl = t & (NB - 1);
    const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(tile_of(t) * TR);
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int rl = (IPW * w + i) * RPI;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bG1, (dvgo_lptr_t)&L.g1[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bH1, (dvgo_lptr_t)&L.h1[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bH2, (dvgo_lptr_t)&L.h2[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < FPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bF, (dvgo_lptr_t)(&L.xf[sl][0][0] + 64 * (w + NW * i)), 4, vo_f[i], r0 * (unsigned)(C * 4), 0, 0);
    unsigned rids[EPW];
#pragma unroll
    for (int i = 0; i < EPW; ++i) rids[i] = lds_u32<0>(lds_addr(&L.rid[t & (NR - 1)][e_row[i]]));
    if constexpr (EPW == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rids[0]), "+v"(rids[1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rids[0]), "+v"(rids[1]), "+v"(rids[EPW - 2]), "+v"(rids[EPW - 1]));
#pragma unroll
    for (int i = 0; i < EPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bE, (dvgo_lptr_t)(&L.xe[sl][0][0] + 64 * (w + NW * i)), 4,
                                               vo_ecol[i] + rids[i] * (unsigned)(E * 4), 0, 0, 0);
    // gz, the sign words and the ray ids of tile t + 2: one instruction each, spread over the waves (the last repeated)
    const unsigned rn = (unsigned)__builtin_amdgcn_readfirstlane(tile_of(t + 2) * TR);
    const int rsl = (t + 2) & (NR - 1);
    auto small = [&](int which) {
      if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(bGz, (dvgo_lptr_t)&L.gz[sl][0][0], 4, vo_gz, r0 * 12u, 0, 0);
      else if (which == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(bM, (dvgo_lptr_t)&L.m2[sl][0][0], 4, vo_m2, r0 * 32u, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(bR, (dvgo_lptr_t)&L.rid[rsl][0], 4, vo_rid, rn * 8u, 0, 0);
    };
    if constexpr (NW == 4) {
      small(w < 2 ? w : 2);
    } else {
      small(w);
      small(2);
    }
  };

  // this lane's operand addresses in slot 0, row h
  const unsigned a_gz = lds_addr(&L.gz[0][h][0]), a_mw = lds_addr(&L.m2[0][h][m_idx]);
  const unsigned a_g1 = lds_addr(&L.g1[0][h][32 * w + j]), a_h1 = lds_addr(&L.h1[0][h][j]);
  const unsigned a_h2 = lds_addr(&L.h2[0][h][32 * w + j]);
  // X column j: a feature column or an embedding column; the trailing columns 32 .. 39 are embedding columns
  // 32 - n_view .. (the host routes shapes with d_in > 32 here only when that is a 16-byte boundary)
  const bool x_in_f = j < n_view;
  const unsigned a_x = x_in_f ? lds_addr(&L.xf[0][h][j]) : lds_addr(&L.xe[0][h][j - n_view]);
  const unsigned x_slot = x_in_f ? (unsigned)(TR * 16 * 4) : (unsigned)(TR * 32 * 4), x_step = x_in_f ? 128u : 256u;
  const bool trail = d_in > 32;
  const unsigned a_xa = lds_addr(&L.xe[0][h][trail ? 32 - n_view : 0]);
  constexpr unsigned SLOT_ROWS = TR * WIDTH * 4;                                   // bytes per slot
  using Ops = WgradOperands<T>;
  auto wait_ops = [&](Ops& o) {
    if constexpr (T == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.gq), "+v"(o.xa), "+v"(o.xb), "+v"(o.mw), "+v"(o.g1v), "+v"(o.h1v[0]),
                   "+v"(o.h1v[1]), "+v"(o.h1v[2]), "+v"(o.h1v[3]), "+v"(o.xv), "+v"(o.h2v));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.gq), "+v"(o.xa), "+v"(o.xb), "+v"(o.mw), "+v"(o.g1v), "+v"(o.h1v[0]),
                   "+v"(o.h1v[1]), "+v"(o.xv), "+v"(o.h2v));
  };

  // the ray ids of the first two tiles by hand, then group 0
  if (tid < 64) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t row = (int64_t)tile_of(t) * TR + (tid & (TR - 1));
      L.rid[t][tid] = row < M ? (unsigned int)ray_id[row] : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NB - 1; ++t) issue(t);
  for (int k = 0; tile_of(k) < n_tiles; ++k) {
    const int sl = k & (NB - 1);
    // behind the barrier every wave's share of tile k has landed and every wave is done with tile k - 1, whose slot
    // group k + 1 overwrites (its ray ids came with group k - 1)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(k + NB - 1);
    const unsigned b_gz = a_gz + sl * (TR * 16), b_mw = a_mw + sl * (TR * 16), b_g1 = a_g1 + sl * SLOT_ROWS;
    const unsigned b_h1 = a_h1 + sl * SLOT_ROWS, b_h2 = a_h2 + sl * SLOT_ROWS;
    const unsigned b_xa = a_xa + sl * (TR * 32 * 4);
    unsigned b_x = a_x + sl * x_slot;                   // advances by two rows per k-step (row length depends on the lane)
    Ops ops[2];
    float a2s[2], a1s[2];
    auto read_ops = [&](auto Sc, Ops& o) {
      constexpr int s = decltype(Sc)::value;
      o.gq = lds_f32x4<32 * s>(b_gz);                  // broadcast
      o.mw = lds_u32<32 * s>(b_mw);
      o.g1v = lds_f32<8 * WIDTH * s>(b_g1);
      static_for<0, T>([&](auto Tc) { constexpr int t = decltype(Tc)::value; o.h1v[t] = lds_f32<8 * WIDTH * s + 128 * t>(b_h1); });
      o.xv = lds_f32<0>(b_x);
      b_x += x_step;
      o.xa = lds_f32x4<256 * s>(b_xa);                 // broadcast; zero past d_in
      o.xb = lds_f32x4<256 * s + 16>(b_xa);
      o.h2v = lds_f32<8 * WIDTH * s>(b_h2);
    };
    // the A operands of a step: G2 rebuilt from gz and the layer-2 sign bit (zeros on rows past M), G1 as read
    auto prep = [&](const Ops& o, float& a2, float& a1) {
      const float g2v = fmaf(w32, o.gq.z, fmaf(w31, o.gq.y, w30 * o.gq.x));
      a2 = ((o.mw >> m_bit) & 1u) ? g2v : 0.0f;
      a1 = o.g1v;
    };
    read_ops(std::integral_constant<int, 0>{}, ops[0]);
    wait_ops(ops[0]);
    prep(ops[0], a2s[0], a1s[0]);
    // One wave per SIMD issues in order, and an MFMA enters the pipe 64 cycles after the previous one: whatever follows a
    // run of MFMAs in program order waits for all of them.  So a step is laid out around its five MFMAs, the gaps between
    // them holding (pinned by sched_barrier) the NEXT step's operand reads -- issued behind the first MFMA, waited for
    // behind the third, turned into A operands behind the fourth -- and this step's bias / dW3 / trailing-column FMAs.
    // Between the last MFMA of a step and the first of the next there is nothing.
    static_for<0, TR / 2>([&](auto Sc) {
      constexpr int s = decltype(Sc)::value;
      constexpr bool more = s + 1 < TR / 2;
      Ops& cur = ops[s & 1];
      Ops& nxt = ops[(s + 1) & 1];
      const float a2 = a2s[s & 1], a1 = a1s[s & 1];
      aW2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, cur.h1v[0], aW2[0], 0, 0, 0);
      if constexpr (more) read_ops(std::integral_constant<int, s + 1>{}, nxt);
      sb2 += a2; sb1 += a1;
      sgz[0] += cur.gq.x; sgz[1] += cur.gq.y; sgz[2] += cur.gq.z;
      __builtin_amdgcn_sched_barrier(0);
      aW2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, cur.h1v[1], aW2[1], 0, 0, 0);
      vW1[0] = fmaf(a1, cur.xa.x, vW1[0]); vW1[1] = fmaf(a1, cur.xa.y, vW1[1]); vW1[2] = fmaf(a1, cur.xa.z, vW1[2]);
      vW1[3] = fmaf(a1, cur.xa.w, vW1[3]);
      vW3[0] = fmaf(cur.gq.x, cur.h2v, vW3[0]); vW3[1] = fmaf(cur.gq.y, cur.h2v, vW3[1]); vW3[2] = fmaf(cur.gq.z, cur.h2v, vW3[2]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T == 4) aW2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, cur.h1v[2], aW2[2], 0, 0, 0);
      vW1[4] = fmaf(a1, cur.xb.x, vW1[4]); vW1[5] = fmaf(a1, cur.xb.y, vW1[5]); vW1[6] = fmaf(a1, cur.xb.z, vW1[6]);
      vW1[7] = fmaf(a1, cur.xb.w, vW1[7]);
      if constexpr (more) wait_ops(nxt);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (T == 4) aW2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, cur.h1v[3], aW2[3], 0, 0, 0);
      if constexpr (more) prep(nxt, a2s[(s + 1) & 1], a1s[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      aW1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, cur.xv, aW1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead groups: nothing may land after the workgroup is gone
  // D[row = (r&3) + 8*(r>>2) + 4*h][col = j]
  float* p = part + (int64_t)blockIdx.x * (WIDTH * WIDTH + WIDTH * 64 + 32 * WIDTH + 3 * WIDTH);
  float* pW2 = p;                          // [WIDTH out][WIDTH in]
  float* pW1 = pW2 + WIDTH * WIDTH;        // [WIDTH out][64]
  float* pW3 = pW1 + WIDTH * 64;           // [32 (c padded)][WIDTH], rows 0..2 written
  float* pb = pW3 + 32 * WIDTH;            // [3][WIDTH]: db1, db2, db3 (first 3 entries)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
    for (int t = 0; t < T; ++t) pW2[(32 * w + i) * WIDTH + 32 * t + j] = aW2[t][r];
    pW1[(32 * w + i) * 64 + j] = aW1[r];
  }
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {         // VALU parts: lanes j and j + 32 hold the even / odd rows
    const float v = vW1[kk] + __shfl_xor(vW1[kk], 32);
    if (h == 0) pW1[(32 * w + j) * 64 + 32 + kk] = (trail && 32 + kk < d_in) ? v : 0.0f;
  }
  if (h == 0) {
#pragma unroll
    for (int kk = 8; kk < 32; ++kk) pW1[(32 * w + j) * 64 + 32 + kk] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = vW3[c] + __shfl_xor(vW3[c], 32);
    if (h == 0) pW3[c * WIDTH + 32 * w + j] = v;
    sgz[c] += __shfl_xor(sgz[c], 32);
  }
  sb1 += __shfl_xor(sb1, 32); sb2 += __shfl_xor(sb2, 32);
  if (h == 0) {
    pb[32 * w + j] = sb1;
    pb[WIDTH + 32 * w + j] = sb2;
    if (j < 8) pb[2 * WIDTH + 8 * w + j] = (w == 0 && j < 3) ? sgz[j] : 0.0f;   // db3 = entries [0,3) (+ [8,11): zero here)
    if (T == 2 && j >= 16) pb[2 * WIDTH + 16 * w + j] = 0.0f;
  }
#endif
}
// ----------------------------------------------------------------------------------
// The same pipeline with the contraction on the bf16 matrix cores (round 3; dvgo_shade_variant bit 6).  Data movement is
// the ring kernel's, unchanged: 16-row tiles of every operand by LDS-DMA one tile ahead, two workgroups per CU.  The body:
//   * wave w reads ITS columns of the tile in MFMA fragment order -- lane (column j, half h) holds rows 8h .. 8h+7 -- and
//     splits each value exactly into three bf16 pieces (x3.h): the A fragments of its own out-feature tile (G2 rebuilt
//     from gz and the sign bits, G1) stay in registers; the B fragment of ONE in-feature tile of H1 (waves 0 / 1: also one
//     of the two X tiles) goes to LDS, where all four waves read it: nothing is split twice (the barrier-free form splits
//     every H1 tile in every wave: 2.2x the VALU work, and VALU work is what bounds these kernels);
//   * second barrier; 36 MFMAs per wave and tile (4 x 6 for dW2, 2 x 6 for dW1: the trailing X columns are a second
//     MFMA tile here instead of 64 FMAs), fragments read from LDS two tiles at a time, MFMAs of two accumulators alternating.
// ----------------------------------------------------------------------------------
typedef float dvgo_f32x2 __attribute__((ext_vector_type(2)));
// two dwords 256 * (O1 - O0) bytes apart with one LDS instruction (rows e and e + 1 of a column: a request costs its wave
// ~10 cycles of issue, whatever its width)
template <int O0, int O1> __device__ __forceinline__ dvgo_f32x2 lds_f32_pair_st64(unsigned a) {
  dvgo_f32x2 v; asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(O0), "n"(O1)); return v;
}
template <int O0, int O1> __device__ __forceinline__ dvgo_f32x2 lds_f32_pair(unsigned a) {       // offsets in dwords
  dvgo_f32x2 v; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(O0), "n"(O1)); return v;
}
template <int OFF> __device__ __forceinline__ u32x4 lds_u32x4(unsigned a) {
  u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v;
}
template <int OFF> __device__ __forceinline__ void lds_store_u32x4(unsigned a, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory");
}

template <int WIDTH>
struct ShadeWgradRingX3 : ShadeWgradRing<WIDTH> {
  u32x4 fb[WIDTH / 32 + 2][3][64];     // shared B fragments: [in-feature tile of H1, then the two X tiles][piece][lane]
};

template <int WIDTH>
__global__ void __launch_bounds__(2 * WIDTH, 2)
shade_wgrad_ring_x3_kernel(const float* __restrict__ G1, const float* __restrict__ gz, const unsigned int* __restrict__ masks,
                           const float* __restrict__ W3, const float* __restrict__ H1, const float* __restrict__ H2,
                           const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                           const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev,
                           float* __restrict__ part, float* __restrict__ total, int total_size) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_size; i += gridDim.x * blockDim.x) total[i] = 0.0f;
  int64_t M;
  {
    const int64_t m = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;
    M = ((int64_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  }
  using Ring = ShadeWgradRingX3<WIDTH>;
  constexpr int T = WIDTH / 32, NW = T, NB = Ring::NB, NR = Ring::NR, TR = Ring::TR;
  static_assert(T == 4 || T == 2, "widths 128 and 64");
  constexpr int LPR = WIDTH / 4, RPI = 64 / LPR, IPW = TR / RPI / NW;
  // The small requests (feature columns, embedding gather, gz, sign words, ray ids) are issued by the waves that do NOT
  // split an X tile: at width 128 waves 2 and 3 (a request costs its wave ~90 cycles; waves 0 / 1 have ~500 cycles of X
  // split the others would otherwise spend waiting at the second barrier).  DW = waves that issue them, dw = index among them.
  constexpr int DW = NW == 4 ? 2 : NW, FPW = 4 / DW, EPW = 8 / DW;
  static_assert(IPW == 2 && NB == 2 && TR == 16, "double buffer of 16-row tiles: the wait at the top of a tile is vmcnt(0)");
  __shared__ __attribute__((aligned(16))) Ring L;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dw = NW == 4 ? w - 2 : w;
  f32x16 aW2[T], aW1[2];
  float vW3[3] = {0.0f, 0.0f, 0.0f}, sgz[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW1[q][r] = 0.0f;
  float sb1 = 0.0f, sb2 = 0.0f;
  const float w30 = W3[32 * w + j], w31 = W3[WIDTH + 32 * w + j], w32 = W3[2 * WIDTH + 32 * w + j];
  const int m_idx = 2 * ((j >> 2) & 1) + (w >> 1);
  const unsigned m_bit = 16 * (w & 1) + (j & 3) + 4 * (j >> 3);
  const int d_in = n_view + E;
  const int n_tiles = (int)((M + TR - 1) / TR);
  auto tile_of = [&](int k) { return (int)blockIdx.x + k * (int)gridDim.x; };
  const unsigned rows_b = (unsigned)(M * WIDTH * 4);
  const auto bG1 = __builtin_amdgcn_make_buffer_rsrc((void*)G1, 0, rows_b, 0x00020000);
  const auto bH1 = __builtin_amdgcn_make_buffer_rsrc((void*)H1, 0, rows_b, 0x00020000);
  const auto bH2 = __builtin_amdgcn_make_buffer_rsrc((void*)H2, 0, rows_b, 0x00020000);
  const auto bF = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, (unsigned)(M * C * 4), 0x00020000);
  const auto bE = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, DVGO_OOB, 0x00020000);
  const auto bGz = __builtin_amdgcn_make_buffer_rsrc((void*)gz, 0, (unsigned)(M * 12), 0x00020000);
  const auto bM = __builtin_amdgcn_make_buffer_rsrc((void*)masks, 0, (unsigned)(M * 32), 0x00020000);
  const auto bR = __builtin_amdgcn_make_buffer_rsrc((void*)ray_id, 0, (unsigned)(M * 8), 0x00020000);
  unsigned vo_bulk[IPW], vo_f[FPW], vo_ecol[EPW];
  int e_row[EPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) vo_bulk[i] = (unsigned)(((IPW * w + i) * RPI + lane / LPR) * WIDTH * 4 + 16 * (lane % LPR));
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int e = 64 * ((dw < 0 ? 0 : dw) + DW * i) + lane, row = e >> 4, col = e & 15;
    vo_f[i] = col < n_view ? (unsigned)((row * C + c_view0 + col) * 4) : DVGO_OOB;
  }
#pragma unroll
  for (int i = 0; i < EPW; ++i) {
    const int e = 64 * ((dw < 0 ? 0 : dw) + DW * i) + lane, col = e & 31;
    e_row[i] = e >> 5;
    vo_ecol[i] = col < E ? (unsigned)(col * 4) : DVGO_OOB;
  }
  const unsigned vo_gz = (lane & 3) < 3 ? (unsigned)(((lane >> 2) * 3 + (lane & 3)) * 4) : DVGO_OOB;
  const unsigned vo_m2 = (unsigned)(((lane >> 2) * 8 + 4 + (lane & 3)) * 4);
  const unsigned vo_rid = (unsigned)((lane & (TR - 1)) * 8);
  auto issue = [&](int t) {           // (the ring kernel's group: tile t, and the ray ids of tile t + 2)
    const int sl = t & (NB - 1);
    const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane(tile_of(t) * TR);
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int rl = (IPW * w + i) * RPI;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bG1, (dvgo_lptr_t)&L.g1[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bH1, (dvgo_lptr_t)&L.h1[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bH2, (dvgo_lptr_t)&L.h2[sl][rl][0], 16, vo_bulk[i], r0 * (WIDTH * 4), 0, 0);
    }
    if (dw < 0) return;                 // (waves 0 / 1 at width 128: the bulk rows above were their whole share)
#pragma unroll
    for (int i = 0; i < FPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bF, (dvgo_lptr_t)(&L.xf[sl][0][0] + 64 * (dw + DW * i)), 4, vo_f[i], r0 * (unsigned)(C * 4), 0, 0);
    unsigned rids[EPW];
#pragma unroll
    for (int i = 0; i < EPW; ++i) rids[i] = lds_u32<0>(lds_addr(&L.rid[t & (NR - 1)][e_row[i]]));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rids[0]), "+v"(rids[1]), "+v"(rids[2]), "+v"(rids[3]));
#pragma unroll
    for (int i = 0; i < EPW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bE, (dvgo_lptr_t)(&L.xe[sl][0][0] + 64 * (dw + DW * i)), 4,
                                               vo_ecol[i] + rids[i] * (unsigned)(E * 4), 0, 0, 0);
    // gz, the sign words and the ray ids of tile t + 2: one instruction each
    const unsigned rn = (unsigned)__builtin_amdgcn_readfirstlane(tile_of(t + 2) * TR);
    const int rsl = (t + 2) & (NR - 1);
    auto small = [&](int which) {
      if (which == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(bGz, (dvgo_lptr_t)&L.gz[sl][0][0], 4, vo_gz, r0 * 12u, 0, 0);
      else if (which == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(bM, (dvgo_lptr_t)&L.m2[sl][0][0], 4, vo_m2, r0 * 32u, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(bR, (dvgo_lptr_t)&L.rid[rsl][0], 4, vo_rid, rn * 8u, 0, 0);
    };
    if (dw == 0) { small(0); small(1); } else small(2);
  };
  // this lane's operand addresses in slot 0: its column, rows 8h .. 8h + 7 (the row enters as the immediate offset)
  const unsigned a_gz = lds_addr(&L.gz[0][8 * h][0]), a_mw = lds_addr(&L.m2[0][8 * h][m_idx]);
  const unsigned a_g1 = lds_addr(&L.g1[0][8 * h][32 * w + j]), a_h1 = lds_addr(&L.h1[0][8 * h][32 * w + j]);
  const unsigned a_h2 = lds_addr(&L.h2[0][8 * h][32 * w + j]);
  // the X tile wave q (= 0, 1) splits: column 32 q + j of X = a feature column (k < n_view), an embedding column, or nothing
  const int xcol = 32 * (w & 1) + j;
  const bool x_in_f = xcol < n_view, x_ok = xcol < d_in;
  const unsigned a_x = x_in_f ? lds_addr(&L.xf[0][8 * h][xcol]) : lds_addr(&L.xe[0][8 * h][x_ok ? xcol - n_view : 0]);
  const unsigned a_fb = lds_addr(&L.fb[0][0][lane]);
  constexpr unsigned SLOT_ROWS = TR * WIDTH * 4, FB_TILE = 3 * 64 * 16, FB_PIECE = 64 * 16;
  if (tid < 64) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t row = (int64_t)tile_of(t) * TR + (tid & (TR - 1));
      L.rid[t][tid] = row < M ? (unsigned int)ray_id[row] : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NB - 1; ++t) issue(t);
  for (int k = 0; tile_of(k) < n_tiles; ++k) {
    const int sl = k & (NB - 1);
    // behind the barrier every wave's share of tile k has landed and every wave is done with tile k - 1: its slot (group
    // k + 1 overwrites it) and the shared fragments
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    issue(k + NB - 1);
    const unsigned b_gz = a_gz + sl * (TR * 16), b_mw = a_mw + sl * (TR * 16), b_g1 = a_g1 + sl * SLOT_ROWS;
    const unsigned b_h1 = a_h1 + sl * SLOT_ROWS, b_h2 = a_h2 + sl * SLOT_ROWS;
    const unsigned b_x = a_x + sl * (x_in_f ? (unsigned)(TR * 16 * 4) : (unsigned)(TR * 32 * 4));
    dvgo_f32x4 gq[8];
    unsigned mw[8];
    float g1v[8], h1v[8], h2v[8], xv[8];
    constexpr int RS = WIDTH * 4 / 256;            // row stride in the 256-byte units of ds_read2st64
    static_for<0, 4>([&](auto Ec) {
      constexpr int e = 2 * decltype(Ec)::value;
      const dvgo_f32x2 v = lds_f32_pair_st64<e * RS, (e + 1) * RS>(b_h1);
      h1v[e] = v.x; h1v[e + 1] = v.y;
    });
    if (w < 2) {
      if (x_in_f) static_for<0, 8>([&](auto Ec) { constexpr int e = decltype(Ec)::value; xv[e] = lds_f32<e * 16 * 4>(b_x); });
      else static_for<0, 8>([&](auto Ec) { constexpr int e = decltype(Ec)::value; xv[e] = lds_f32<e * 32 * 4>(b_x); });
    }
    static_for<0, 8>([&](auto Ec) {
      constexpr int e = decltype(Ec)::value;
      gq[e] = lds_f32x4<16 * e>(b_gz);                 // broadcast
    });
    static_for<0, 4>([&](auto Ec) {
      constexpr int e = 2 * decltype(Ec)::value;
      const dvgo_f32x2 m = lds_f32_pair<4 * e, 4 * (e + 1)>(b_mw);
      mw[e] = __float_as_uint(m.x); mw[e + 1] = __float_as_uint(m.y);
      const dvgo_f32x2 a = lds_f32_pair_st64<e * RS, (e + 1) * RS>(b_g1);
      g1v[e] = a.x; g1v[e + 1] = a.y;
      const dvgo_f32x2 b = lds_f32_pair_st64<e * RS, (e + 1) * RS>(b_h2);
      h2v[e] = b.x; h2v[e + 1] = b.y;
    });
    // the shared B fragments first: the other waves wait for them
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h1v[0]), "+v"(h1v[1]), "+v"(h1v[2]), "+v"(h1v[3]), "+v"(h1v[4]), "+v"(h1v[5]),
                 "+v"(h1v[6]), "+v"(h1v[7]), "+v"(gq[0]), "+v"(gq[1]), "+v"(gq[2]), "+v"(gq[3]), "+v"(gq[4]), "+v"(gq[5]), "+v"(gq[6]),
                 "+v"(gq[7]));
    asm volatile("" : "+v"(mw[0]), "+v"(mw[1]), "+v"(mw[2]), "+v"(mw[3]), "+v"(mw[4]), "+v"(mw[5]), "+v"(mw[6]), "+v"(mw[7]),
                 "+v"(g1v[0]), "+v"(g1v[1]), "+v"(g1v[2]), "+v"(g1v[3]), "+v"(g1v[4]), "+v"(g1v[5]), "+v"(g1v[6]), "+v"(g1v[7]));
    asm volatile("" : "+v"(h2v[0]), "+v"(h2v[1]), "+v"(h2v[2]), "+v"(h2v[3]), "+v"(h2v[4]), "+v"(h2v[5]), "+v"(h2v[6]), "+v"(h2v[7]));
    {
      u32x4 p0, p1, p2;
      x3_split8(h1v, p0, p1, p2);
      lds_store_u32x4<0>(a_fb + w * FB_TILE, p0);
      lds_store_u32x4<FB_PIECE>(a_fb + w * FB_TILE, p1);
      lds_store_u32x4<2 * FB_PIECE>(a_fb + w * FB_TILE, p2);
      if (w < 2) {
        asm volatile("" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3]), "+v"(xv[4]), "+v"(xv[5]), "+v"(xv[6]), "+v"(xv[7]));
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = x_ok ? xv[e] : 0.0f;
        x3_split8(xv, p0, p1, p2);
        lds_store_u32x4<0>(a_fb + (T + w) * FB_TILE, p0);
        lds_store_u32x4<FB_PIECE>(a_fb + (T + w) * FB_TILE, p1);
        lds_store_u32x4<2 * FB_PIECE>(a_fb + (T + w) * FB_TILE, p2);
      }
    }
    // this wave's A fragments: G2 rebuilt from gz and the layer-2 sign bit (zeros on rows past M), G1 as read; fp32 tails
    u32x4 a2f[3], a1f[3];
    {
      float a2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g2v = fmaf(w32, gq[e].z, fmaf(w31, gq[e].y, w30 * gq[e].x));
        a2[e] = __uint_as_float(__float_as_uint(g2v) & (unsigned)__builtin_amdgcn_sbfe((int)mw[e], m_bit, 1u));
        sb2 += a2[e]; sb1 += g1v[e];
        vW3[0] = fmaf(gq[e].x, h2v[e], vW3[0]); vW3[1] = fmaf(gq[e].y, h2v[e], vW3[1]); vW3[2] = fmaf(gq[e].z, h2v[e], vW3[2]);
      }
      if (w == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sgz[0] += gq[e].x; sgz[1] += gq[e].y; sgz[2] += gq[e].z; }
      }
      x3_split8(a2, a2f[0], a2f[1], a2f[2]);
      x3_split8(g1v, a1f[0], a1f[1], a1f[2]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // the shared fragments are complete
    // 36 MFMAs: fragments two tiles at a time, the MFMAs of the two accumulators alternating (no dependent pair back to back)
    u32x4 fr[2][2][3];
    auto read_pair = [&](auto Pc, u32x4 (&f)[2][3]) {
      constexpr int p = decltype(Pc)::value;                      // tiles 2p, 2p + 1 of fb
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        f[0][q] = lds_u32x4<(2 * p) * FB_TILE>(a_fb + q * FB_PIECE);
        f[1][q] = lds_u32x4<(2 * p + 1) * FB_TILE>(a_fb + q * FB_PIECE);
      }
    };
    auto wait_pair = [&](u32x4 (&f)[2][3]) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]));
    };
    auto mfma_pair = [&](f32x16& accA, f32x16& accB, const u32x4 (&a)[3], const u32x4 (&f)[2][3]) {
#define X3D_MF(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0)
      X3D_MF(accA, a[2], f[0][0]); X3D_MF(accB, a[2], f[1][0]);       // smallest terms first (x3_mfma6's order per accumulator)
      X3D_MF(accA, a[1], f[0][1]); X3D_MF(accB, a[1], f[1][1]);
      X3D_MF(accA, a[0], f[0][2]); X3D_MF(accB, a[0], f[1][2]);
      X3D_MF(accA, a[1], f[0][0]); X3D_MF(accB, a[1], f[1][0]);
      X3D_MF(accA, a[0], f[0][1]); X3D_MF(accB, a[0], f[1][1]);
      X3D_MF(accA, a[0], f[0][0]); X3D_MF(accB, a[0], f[1][0]);
#undef X3D_MF
    };
    constexpr int NP = T / 2 + 1;                                  // pairs of tiles: T / 2 of H1, then the X pair
    read_pair(std::integral_constant<int, 0>{}, fr[0]);
    static_for<0, NP>([&](auto Pc) {
      constexpr int p = decltype(Pc)::value;
      wait_pair(fr[p & 1]);
      if constexpr (p + 1 < NP) read_pair(std::integral_constant<int, p + 1>{}, fr[(p + 1) & 1]);
      if constexpr (p < T / 2) mfma_pair(aW2[2 * p], aW2[2 * p + 1], a2f, fr[p & 1]);
      else mfma_pair(aW1[0], aW1[1], a1f, fr[p & 1]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead groups: nothing may land after the workgroup is gone
  float* p = part + (int64_t)blockIdx.x * (WIDTH * WIDTH + WIDTH * 64 + 32 * WIDTH + 3 * WIDTH);
  float* pW2 = p;
  float* pW1 = pW2 + WIDTH * WIDTH;
  float* pW3 = pW1 + WIDTH * 64;
  float* pb = pW3 + 32 * WIDTH;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
    for (int t = 0; t < T; ++t) pW2[(32 * w + i) * WIDTH + 32 * t + j] = aW2[t][r];
    pW1[(32 * w + i) * 64 + j] = aW1[0][r];
    pW1[(32 * w + i) * 64 + 32 + j] = aW1[1][r];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = vW3[c] + __shfl_xor(vW3[c], 32);
    if (h == 0) pW3[c * WIDTH + 32 * w + j] = v;
    sgz[c] += __shfl_xor(sgz[c], 32);
  }
  sb1 += __shfl_xor(sb1, 32); sb2 += __shfl_xor(sb2, 32);
  if (h == 0) {
    pb[32 * w + j] = sb1;
    pb[WIDTH + 32 * w + j] = sb2;
    if (j < 8) pb[2 * WIDTH + 8 * w + j] = (w == 0 && j < 3) ? sgz[j] : 0.0f;
    if (T == 2 && j >= 16) pb[2 * WIDTH + 16 * w + j] = 0.0f;
  }
#endif
}
#undef DVGO_OOB

// sum of the per-workgroup partials: [n_parts][n] -> [n].  blockIdx.y takes a slice of the parts so that a few
// thousand wavefronts stream the 60 MB (one column block alone would leave the chip idle and latency-bound:
// 124 us -> ~20 us); slices meet in `out` (zeroed by the caller) with one float atomic per element.
#define SHADE_REDUCE_SLICES 16
// The sums leave in the COMPACT record the caller hands to the optimizer as views, no repacking launches:
//   { dW2 [W][W], dW1 [W][d_in], dW3 [3][W], db1 [W], db2 [W], db3 [3] }   (the padded columns / rows of a part are dropped,
//   the two halves of db3 meet here)
__global__ void __launch_bounds__(256)
shade_wgrad_reduce_kernel(const float* __restrict__ part, int n_parts, int n, int W, int d_in, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int o;                                    // index in the compact record, -1: padding
  const int nW2 = W * W, nW1 = W * 64, nW3 = 32 * W;
  if (i < nW2) o = i;
  else if (i < nW2 + nW1) {
    const int r = (i - nW2) >> 6, c = (i - nW2) & 63;
    o = c < d_in ? nW2 + r * d_in + c : -1;
  } else if (i < nW2 + nW1 + nW3) {
    const int e = i - nW2 - nW1;
    o = e < 3 * W ? nW2 + W * d_in + e : -1;
  } else {
    const int e = i - nW2 - nW1 - nW3, base = nW2 + W * d_in + 3 * W;
    if (e < 2 * W) o = base + e;
    else {
      const int c = e - 2 * W;
      o = c < 3 ? base + 2 * W + c : (c >= 8 && c < 11) ? base + 2 * W + c - 8 : -1;
    }
  }
  if (o < 0) return;
  const int per = (n_parts + SHADE_REDUCE_SLICES - 1) / SHADE_REDUCE_SLICES;
  const int p0 = blockIdx.y * per, p1 = min(n_parts, p0 + per);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int p = p0;
  for (; p + 4 <= p1; p += 4) {
    a0 += part[(int64_t)p * n + i];
    a1 += part[(int64_t)(p + 1) * n + i];
    a2 += part[(int64_t)(p + 2) * n + i];
    a3 += part[(int64_t)(p + 3) * n + i];
  }
  for (; p < p1; ++p) a0 += part[(int64_t)p * n + i];
  if (p1 > p0) atomicAdd(out + o, (a0 + a1) + (a2 + a3));
}

extern "C" {

int dvgo_shade_fwd(const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                   const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                   const float* b3, int width, int d_in, int diffuse, float* rgb, float* H1, float* H2,
                   uint64_t* masks, void* scratch, void* scratch_bwd, void* stream) {
  if (M < 0 || C <= 0 || E < 0) return DVGO_EINVAL;
  if (M == 0) return 0;
  if (!feat || !emb || !ray_id || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !rgb) return DVGO_EINVAL;
  if ((H1 == nullptr) != (H2 == nullptr) || (H1 == nullptr) != (masks == nullptr)) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if (n_view < 0 || d_in != n_view + E) return DVGO_EINVAL;
  if ((width != 128 && width != 64) || d_in > 40) return DVGO_ERANGE;   // outside the instantiated set: caller falls back
  if ((g_shade_variant & 1) && scratch != nullptr)
    return dvgo_shade_fwd_x3(feat, C, emb, E, ray_id, M, m_dev, W1, b1, W2, b2, W3, b3, width, d_in, diffuse, rgb, H1, H2, masks,
                             scratch, (g_shade_variant & 2) ? scratch_bwd : nullptr, g_shade_experiment, stream);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t cap = width == 128 ? 256 : 512;          // width 128: one workgroup per CU (LDS); width 64: two
  int blocks = (int)((n_tiles + SHADE_WAVES - 1) / SHADE_WAVES < cap ? (n_tiles + SHADE_WAVES - 1) / SHADE_WAVES : cap);
#define DVGO_SHADE(W, S1, DIFF)                                                                           \
  shade_fwd_kernel<W, S1, DIFF><<<blocks, SHADE_THREADS, 0, s>>>(feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, W1, b1, W2, \
                                                          b2, W3, b3, d_in, rgb, H1, H2, (unsigned long long*)masks, g_shade_experiment)
  // S1 = k-steps of layer 1 (2 inputs each, zero-padded): 128-wide head of configs/default.py: d_in 36 / 39;
  // 64-wide head of configs/llff (lib/dmpigo.py): d_in = 9 + 3
  if (width == 128) {
    if (d_in <= 36) { if (diffuse) DVGO_SHADE(128, 18, true); else DVGO_SHADE(128, 18, false); }
    else            { if (diffuse) DVGO_SHADE(128, 20, true); else DVGO_SHADE(128, 20, false); }
  } else {
    if (d_in <= 12) { if (diffuse) DVGO_SHADE(64, 6, true); else DVGO_SHADE(64, 6, false); }
    else            { if (diffuse) DVGO_SHADE(64, 20, true); else DVGO_SHADE(64, 20, false); }
  }
#undef DVGO_SHADE
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_shade_experiment(int flags) { g_shade_experiment = flags; return 0; }

int dvgo_shade_variant(int flags) { const int old = g_shade_variant; if (flags >= 0) g_shade_variant = flags; return old; }

int dvgo_shade_bwd(const float* g_rgb, const float* rgb, const uint64_t* masks, int64_t M, const int64_t* m_dev,
                   const float* W1, const float* W2, const float* W3, int width, int d_in, int C, int diffuse,
                   float* g_feat, float* G1, float* gz, void* scratch, int prebuilt, void* stream) {
  if (M < 0 || C <= 0) return DVGO_EINVAL;
  if (M == 0) return 0;
  if (!g_rgb || !rgb || !masks || !W1 || !W2 || !W3 || !g_feat || !G1 || !gz) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if ((width != 128 && width != 64) || n_view < 0 || n_view > 32 || d_in < n_view) return DVGO_ERANGE;
  if ((g_shade_variant & 2) && scratch != nullptr)
    return dvgo_shade_bwd_x3(g_rgb, rgb, masks, M, m_dev, W1, W2, W3, width, d_in, C, diffuse, g_feat, G1, gz, scratch, prebuilt, stream);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t cap = width == 128 ? 256 : 512;
  int blocks = (int)((n_tiles + SHADE_BWD_WAVES - 1) / SHADE_BWD_WAVES < cap ? (n_tiles + SHADE_BWD_WAVES - 1) / SHADE_BWD_WAVES : cap);
#define DVGO_SHADE_BWD(W, DIFF)                                                                                                       \
  shade_bwd_kernel<W, DIFF><<<blocks, SHADE_BWD_THREADS, 0, s>>>(g_rgb, rgb, (const unsigned long long*)masks, M, m_dev, W1, W2, W3, d_in, C, c_view0, \
                                                            n_view, g_feat, G1, gz, g_shade_experiment)
  if (width == 128) { if (diffuse) DVGO_SHADE_BWD(128, true); else DVGO_SHADE_BWD(128, false); }
  else              { if (diffuse) DVGO_SHADE_BWD(64, true); else DVGO_SHADE_BWD(64, false); }
#undef DVGO_SHADE_BWD
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_shade_wgrad(const float* G1, const float* gz, const uint64_t* masks, const float* W3, const float* H1,
                     const float* H2, const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M,
                     const int64_t* m_dev, int width, int diffuse, int n_parts, float* part, float* total, void* stream) {
  if (M < 0 || n_parts <= 0 || C <= 0 || E < 0) return DVGO_EINVAL;
  if (!G1 || !gz || !masks || !W3 || !H1 || !H2 || !feat || !emb || !ray_id || !part || !total) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if ((width != 128 && width != 64) || n_view < 0 || n_view + E > 40) return DVGO_ERANGE;
  bool ring = false;                        // (the pipelined kernel zeroes `total` itself: one launch less)
  if (g_shade_variant & (4 | 32)) {
    // bit 5: two barrier-free kernels; bit 4: 4-wave workgroups, two per CU (68 KB of LDS each)
    const int form_b = (g_shade_variant & 32) ? 2 : (g_shade_variant & 16) ? 1 : 0;
    if (!form_b && n_parts > 256) n_parts = 256;           // else: one 8-wave workgroup per CU (141 KB of LDS)
    const int rc = dvgo_shade_wgrad_x3(G1, gz, masks, W3, H1, H2, feat, C, emb, E, ray_id, M, m_dev, width, diffuse, n_parts, part,
                                       form_b, stream);
    if (rc != 0) return rc;
  } else if (!(g_shade_variant & 8) && n_view <= 16 && E <= 32 && (n_view + E <= 32 || (n_view % 4 == 0 && n_view >= 8)) &&
             M * width * 4 < ((int64_t)1 << 31) && M * C * 4 < ((int64_t)1 << 31)) {
    if (n_parts > 512) n_parts = 512;       // two workgroups per CU (55 KB of LDS each; 74 KB with the shared fragments)
    ring = true;
    const int tsz = width * width + width * (n_view + E) + 5 * width + 3;
    if ((g_shade_variant & 64) && width == 128)
      shade_wgrad_ring_x3_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(
          G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part, total, tsz);
    else if (g_shade_variant & 64)
      shade_wgrad_ring_x3_kernel<64><<<n_parts, 128, 0, (hipStream_t)stream>>>(
          G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part, total, tsz);
    else if (width == 128)
      shade_wgrad_ring_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(
          G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part, total, tsz);
    else
      shade_wgrad_ring_kernel<64><<<n_parts, 128, 0, (hipStream_t)stream>>>(
          G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part, total, tsz);
  } else if (width == 128)
    shade_wgrad_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(
        G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  else
    shade_wgrad_kernel<64><<<n_parts, 128, 0, (hipStream_t)stream>>>(
        G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  DVGO_LAUNCH_CHECK();
  const int psize = width * width + width * 64 + 32 * width + 3 * width;
  const int d_in = n_view + E;
  const int tsize = width * width + width * d_in + 3 * width + 2 * width + 3;
  if (!ring && hipMemsetAsync(total, 0, (size_t)tsize * sizeof(float), (hipStream_t)stream) != hipSuccess) return DVGO_EINVAL;
  shade_wgrad_reduce_kernel<<<dim3((psize + 255) / 256, SHADE_REDUCE_SLICES), 256, 0, (hipStream_t)stream>>>(part, n_parts, psize,
                                                                                                       width, d_in, total);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
