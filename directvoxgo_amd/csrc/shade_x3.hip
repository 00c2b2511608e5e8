// Colour head on the bf16 matrix cores with fp32 results: every fp32 operand is split EXACTLY into three bf16 pieces,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)        (8 + 8 + 8 mantissa bits)
// and a product is taken as the six partial products of order < 2^-24,
//     x w  ~  x0 w0 + x0 w1 + x1 w0 + x0 w2 + x1 w1 + x2 w0        (dropped: x1 w2 + x2 w1 + x2 w2 <= 2^-23 |x w|)
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The rounding of the result is
// that of an fp32 dot product (the dropped terms are below the fp32 rounding of the sum), but six bf16 MFMAs cover a
// 16-deep k-step in 6 x 32 cycles where v_mfma_f32_32x32x2_f32 needs 8 x 64: 2.7x fewer matrix-pipe cycles
// (tools/micro/bf16x3_loop.hip: the 128 x 128 layer sustains 260-270 TFLOP/s fp32-equivalent against 122 for the f32
// MFMA loop of shade.hip).  The reference runs this MLP in fp32 (lib/dvgo.py:123-131,516-541); the tests that hold the
// f32-MFMA kernels against the torch modules (rtol 1e-5 values, 2e-4 gradients) hold these kernels unchanged.
//
// Structure = shade.hip's: every layer is computed transposed (weights = A operand, activations = B operand), so a
// 32 x 32 result tile has the sample on the lane and 16 output features in the registers; registers 8s .. 8s+7 of a
// tile, converted, ARE the B fragment of k-step s of the next layer (cdna_hip_programming.md "An accumulator tile as
// the next MFMA's operand"), with the weights stored in LDS pre-split and pre-permuted to that k order.  Activations
// never leave registers; the splits cost ~11 VALU instructions per pair of values, issued between the MFMAs.
#include "common.h"
#include "x3.h"

#ifndef X3_THREADS
#define X3_THREADS 512                 // 8 wavefronts: two per SIMD (<= 256 registers), one workgroup per CU (LDS)
#endif
#define X3_WAVES (X3_THREADS / 64)
#ifndef X3_BWD_THREADS
#define X3_BWD_THREADS 512             // (768 = three waves per SIMD: fits the LDS, but 168 registers mean 70 B of scratch: 0.53 against 0.50 ms)
#endif
#define X3_BWD_WAVES (X3_BWD_THREADS / 64)
#define X3_STAGE_ROWS 16
#define X3_STAGE_STRIDE 36

__device__ __forceinline__ int x3_acc_feature(int t, int r, int h) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void x3_split1(float v, unsigned short& s0, unsigned short& s1, unsigned short& s2) {
  const unsigned int h = x3_pack(v, 0.f) & 0xffffu;
  const float r = v - __uint_as_float(h << 16);
  const unsigned int m = x3_pack(r, 0.f) & 0xffffu;
  const float q = r - __uint_as_float(m << 16);
  s0 = (unsigned short)h; s1 = (unsigned short)m; s2 = (unsigned short)(x3_pack(q, 0.f) & 0xffffu);
}

// The weights as the kernels want them in LDS -- split into bf16 pieces and permuted to the MFMA fragment order -- are
// built ONCE per call by a small kernel (x3_prep_*) into a global image that every workgroup then copies with 16-byte
// loads; splitting in every workgroup cost 50-100 us per launch, which is most of a sparse step.
template <int WIDTH, int KS1>
struct X3Img {
  static constexpr int T = WIDTH / 32;
  u32x4 w1s[T][KS1][3][64];          // [out tile][k-step][piece][lane]: A fragments of layer 1 (k = 16 s + 8 h + j)
  u32x4 w2s[T][2 * T][3][64];        // [out tile][k-step = 2 t_in + s2][piece][lane]: A fragments of layer 2, k order of
                                     //   the accumulator chain: k = 32 t_in + 16 s2 + 8 (j >> 2) + 4 h + (j & 3)
  float w3p[3][2][T * 16];           // [c][lane half][in tile * 16 + reg]
  float b1p[2][T * 16];              // accumulator-layout biases
  float b2p[2][T * 16];
  float b3[4];
};

template <int WIDTH, int KS1>
struct X3Lds : X3Img<WIDTH, KS1> {
  float stage[X3_WAVES][X3_STAGE_ROWS * X3_STAGE_STRIDE];
};

template <typename Img>
__device__ __forceinline__ void x3_copy_image(Img* __restrict__ dst /* LDS */, const void* __restrict__ src) {
  static_assert(sizeof(Img) % 16 == 0, "16-byte pieces");
  const u32x4* g = reinterpret_cast<const u32x4*>(src);
  u32x4* l = reinterpret_cast<u32x4*>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(Img) / 16); i += blockDim.x) l[i] = g[i];
}

// Row-major store of one 32 (sample) x 32 (feature) accumulator tile through a wave-private LDS patch, 16 rows at a time
// (see shade.hip shade_store_tile: a direct store would touch 32 rows x 32 B per wave instruction).
__device__ __forceinline__ void x3_store_tile(float* __restrict__ stage, const f32x16& acc, float* __restrict__ dst, int row_stride,
                                              int lane, int rows_valid) {
  const int smp = lane & 31, h = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((smp >> 4) == half) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(stage + (smp & 15) * X3_STAGE_STRIDE + 8 * q + 4 * h) =
            make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    const int c = lane & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl = (lane >> 3) + 8 * i;
      const float4 v = *reinterpret_cast<const float4*>(stage + rl * X3_STAGE_STRIDE + 4 * c);
      const int r = 16 * half + rl;
      if (r < rows_valid) *reinterpret_cast<float4*>(dst + (int64_t)r * row_stride + 4 * c) = v;
    }
  }
}

template <int WIDTH, int KS1>
__device__ __forceinline__ void
x3_prep_fwd_body(X3Img<WIDTH, KS1>* __restrict__ Lp, const float* __restrict__ W1, const float* __restrict__ b1,
                 const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ W3,
                 const float* __restrict__ b3, int D_in, const int tid, const int nt) {
  constexpr int T = WIDTH / 32;
  X3Img<WIDTH, KS1>& L = *Lp;
  for (int i = tid; i < T * KS1 * 64 * 8; i += nt) {
    const int j = i & 7, l = (i >> 3) & 63, s = (i >> 9) % KS1, t = (i >> 9) / KS1;
    const int k = 16 * s + 8 * (l >> 5) + j;
    const float v = (k < D_in) ? W1[(32 * t + (l & 31)) * D_in + k] : 0.0f;
    unsigned short p0, p1, p2;
    x3_split1(v, p0, p1, p2);
    reinterpret_cast<unsigned short*>(&L.w1s[t][s][0][l])[j] = p0;
    reinterpret_cast<unsigned short*>(&L.w1s[t][s][1][l])[j] = p1;
    reinterpret_cast<unsigned short*>(&L.w1s[t][s][2][l])[j] = p2;
  }
  for (int i = tid; i < T * 2 * T * 64 * 8; i += nt) {
    const int j = i & 7, l = (i >> 3) & 63, ks = (i >> 9) % (2 * T), t2 = (i >> 9) / (2 * T);
    const int k = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);          // = 32 t_in + 16 s2 + ...
    const float v = W2[(32 * t2 + (l & 31)) * WIDTH + k];
    unsigned short p0, p1, p2;
    x3_split1(v, p0, p1, p2);
    reinterpret_cast<unsigned short*>(&L.w2s[t2][ks][0][l])[j] = p0;
    reinterpret_cast<unsigned short*>(&L.w2s[t2][ks][1][l])[j] = p1;
    reinterpret_cast<unsigned short*>(&L.w2s[t2][ks][2][l])[j] = p2;
  }
  for (int i = tid; i < 3 * 2 * T * 16; i += nt) {
    const int tr = i % (T * 16), h = (i / (T * 16)) & 1, c = i / (2 * T * 16);
    (&L.w3p[0][0][0])[i] = W3[c * WIDTH + x3_acc_feature(tr >> 4, tr & 15, h)];
  }
  for (int i = tid; i < 2 * T * 16; i += nt) {
    const int tr = i % (T * 16), h = i / (T * 16);
    const int f = x3_acc_feature(tr >> 4, tr & 15, h);
    (&L.b1p[0][0])[i] = b1[f];
    (&L.b2p[0][0])[i] = b2[f];
  }
  if (tid < 4) L.b3[tid] = tid < 3 ? b3[tid] : 0.0f;
}

template <int WIDTH, int KS1, bool DIFFUSE>
__global__ void __launch_bounds__(X3_THREADS)
shade_fwd_x3_kernel(const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                    const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev,
                    const void* __restrict__ image, float* __restrict__ rgb,
                    float* __restrict__ H1, float* __restrict__ H2, unsigned long long* __restrict__ masks, int experiment) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  __shared__ __attribute__((aligned(16))) X3Lds<WIDTH, KS1> L;
  x3_copy_image<X3Img<WIDTH, KS1>>(&L, image);
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  float* stage = L.stage[threadIdx.x >> 6];
  const bool train = H1 != nullptr;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  // A wave walks its tiles alone, so what it loads at the top of a tile it waits for, once per tile, behind the 128
  // activation stores of the tile before (stamps: a quarter of the tile's time).  The 24 raw inputs of tile i + 1 are
  // requested at the top of tile i, and -- the embedding row hangs off the ray id -- the ray id of tile i + 2 with them.
  // (Round 2 measured this at 32 spilled registers and +0.14 ms; without the packed-fp32 instructions the same source
  // allocates as before: 0.63-0.67 -> 0.60 ms.)
  const int d_in = n_view + E;
  const int64_t last = M > 0 ? M - 1 : 0;
  auto row_of = [&](int64_t tile) { const int64_t r = tile * 32 + (lane & 31); return r < M ? r : last; };
  float xin[KS1][8];
  auto load_inputs = [&](int64_t tile, int64_t rid) {
    const float* fr = feat + row_of(tile) * C + c_view0;
    const float* er = emb + rid * E - n_view;
#pragma unroll
    for (int s = 0; s < KS1; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {            // branch-free: every lane loads from a valid address, selected at use
        const int k = 16 * s + 8 * h + j;
        const int kc = k < d_in ? k : d_in - 1;
        xin[s][j] = *((kc < n_view) ? fr + kc : er + kc);
      }
  };
  int64_t rid_n = 0;
  if (gw < n_tiles) { load_inputs(gw, ray_id[row_of(gw)]); rid_n = ray_id[row_of(gw + nw)]; }
  for (int64_t tile = gw; tile < n_tiles; tile += nw) {
    const int64_t row = tile * 32 + (lane & 31);
    const bool valid = row < M;
    const int rows_valid = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
    // ---- X^T fragments of layer 1: element j of k-step s = input feature k = 16 s + 8 h + j of this lane's row:
    //   k < n_view      : feat[row, c_view0 + k]          (k0_view,  lib/dvgo.py:518-523)
    //   k < n_view + E  : emb[ray_id[row], k - n_view]    (viewdirs_emb[ray_id], lib/dvgo.py:524-526)
    u32x4 x3[KS1][3];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (16 * s + 8 * h + j) < d_in ? xin[s][j] : 0.0f;
      x3_split8(v, x3[s][0], x3[s][1], x3[s][2]);
    }
    load_inputs(tile + nw, rid_n);
    rid_n = ray_id[row_of(tile + 2 * nw)];
    // ---- layer 1: all WIDTH features (every layer-2 output needs them)
    u32x4 h3[2 * T][3];                       // its post-ReLU activations as the B fragments of layer 2
    unsigned long long mask1 = 0ull;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = L.b1p[h][t * 16 + r];
#pragma unroll
      for (int s = 0; s < KS1; ++s)
        x3_mfma6(acc, L.w1s[t][s][0][lane], L.w1s[t][s][1][lane], L.w1s[t][s][2][lane], x3[s][0], x3[s][1], x3[s][2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
      if (train) {                             // post-ReLU activations (weight gradients) + their sign bits
        unsigned int bits = 0u;                // (after the ReLU: positive <=> bit pattern non-zero; -0 cannot occur)
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= min(__float_as_uint(acc[r]), 1u) << r;
        mask1 |= (unsigned long long)bits << (16 * t);
        if (!(experiment & 512)) x3_store_tile(stage, acc, H1 + tile * 32 * WIDTH + 32 * t, WIDTH, lane, rows_valid);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[8 * s2 + j];
        x3_split8(v, h3[2 * t + s2][0], h3[2 * t + s2][1], h3[2 * t + s2][2]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- layer 2, one 32-feature output tile at a time, consumed immediately (H2 store, its share of layer 3)
    float p[3] = {0.0f, 0.0f, 0.0f};
    unsigned long long mask2 = 0ull;
#pragma unroll 1
    for (int t2 = 0; t2 < T; ++t2) {
      f32x16 acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = L.b2p[h][t2 * 16 + r];
      // the A fragments of k-step ks + 1 are requested before the MFMAs of k-step ks (explicit double buffer: left to
      // itself the scheduler either hoists every read of the tile -- spills -- or none -- an LDS round trip per k-step)
      u32x4 a[2][3];
#pragma unroll
      for (int q = 0; q < 3; ++q) a[0][q] = L.w2s[t2][0][q][lane];
#pragma unroll
      for (int ks = 0; ks < 2 * T; ++ks) {
        if (ks + 1 < 2 * T) {
#pragma unroll
          for (int q = 0; q < 3; ++q) a[(ks + 1) & 1][q] = L.w2s[t2][ks + 1][q][lane];
        }
        x3_mfma6(acc2, a[ks & 1][0], a[ks & 1][1], a[ks & 1][2], h3[ks][0], h3[ks][1], h3[ks][2]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = fmaxf(acc2[r], 0.0f);
      if (train) {
        unsigned int bits = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= min(__float_as_uint(acc2[r]), 1u) << r;
        mask2 |= (unsigned long long)bits << (16 * t2);
        if (!(experiment & 512)) x3_store_tile(stage, acc2, H2 + tile * 32 * WIDTH + 32 * t2, WIDTH, lane, rows_valid);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) p[c] = fmaf(L.w3p[c][h][t2 * 16 + r], acc2[r], p[c]);
      }
    }
    float z[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) z[c] = p[c] + __shfl_xor(p[c], 32) + L.b3[c];
    if (valid) {
      if (h == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float zz = DIFFUSE ? z[c] + feat[row * C + c] : z[c];
          rgb[row * 3 + c] = 1.0f / (1.0f + expf(-zz));
        }
      }
      if (train) {
        masks[(row * 2 + 0) * 2 + h] = mask1;
        masks[(row * 2 + 1) * 2 + h] = mask2;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// Backward, data-gradient part (shade.hip shade_bwd_kernel on the split operands).  Per 32-row tile:
//   gz  = g_rgb * rgb * (1 - rgb)
//   G2  = (H2 > 0) * (W3^T gz)                 VALU, accumulator layout; its registers 8s..8s+7, split, are the B fragments of
//   G1  = (H1 > 0) * (W2^T G2)                 bf16 MFMA x 6, A = W2^T pre-split / pre-permuted to the accumulator's k order
//   gx  = W1[:, :32]^T G1                      the same chaining once more (only the feature-grid inputs need a gradient)
// G1 and gz are written out for the weight gradients; the sign bits of H1 / H2 come from the forward's masks.
// ----------------------------------------------------------------------------------
template <int WIDTH>
struct X3BwdImg {
  static constexpr int T = WIDTH / 32;
  u32x4 w2t[T][2 * T][3][64];      // [in tile][k-step over f_out][piece][lane]: A[i = f_in][k] = W2[f_out(k)][32 tin + i]
  u32x4 w1t[2 * T][3][64];         // [k-step over f][piece][lane]:              A[i = k_in][k] = W1[f(k)][i]   (i < D_in else 0)
  float w3p[3][2][T * 16];
};

template <int WIDTH>
struct X3BwdLds : X3BwdImg<WIDTH> {
  float stage[X3_BWD_WAVES][X3_STAGE_ROWS * X3_STAGE_STRIDE];
};

template <int WIDTH>
__device__ __forceinline__ void
x3_prep_bwd_body(X3BwdImg<WIDTH>* __restrict__ Lp, const float* __restrict__ W1, const float* __restrict__ W2,
                 const float* __restrict__ W3, int D_in, const int tid, const int nt) {
  constexpr int T = WIDTH / 32;
  X3BwdImg<WIDTH>& L = *Lp;
  for (int i = tid; i < T * 2 * T * 64 * 8; i += nt) {
    const int j = i & 7, l = (i >> 3) & 63, ks = (i >> 9) % (2 * T), tin = (i >> 9) / (2 * T);
    const int f_out = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
    unsigned short p0, p1, p2;
    x3_split1(W2[f_out * WIDTH + 32 * tin + (l & 31)], p0, p1, p2);
    reinterpret_cast<unsigned short*>(&L.w2t[tin][ks][0][l])[j] = p0;
    reinterpret_cast<unsigned short*>(&L.w2t[tin][ks][1][l])[j] = p1;
    reinterpret_cast<unsigned short*>(&L.w2t[tin][ks][2][l])[j] = p2;
  }
  for (int i = tid; i < 2 * T * 64 * 8; i += nt) {
    const int j = i & 7, l = (i >> 3) & 63, ks = i >> 9;
    const int f = 16 * ks + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3), k = l & 31;
    unsigned short p0, p1, p2;
    x3_split1((k < D_in) ? W1[f * D_in + k] : 0.0f, p0, p1, p2);
    reinterpret_cast<unsigned short*>(&L.w1t[ks][0][l])[j] = p0;
    reinterpret_cast<unsigned short*>(&L.w1t[ks][1][l])[j] = p1;
    reinterpret_cast<unsigned short*>(&L.w1t[ks][2][l])[j] = p2;
  }
  for (int i = tid; i < 3 * 2 * T * 16; i += nt) {
    const int tr = i % (T * 16), h = (i / (T * 16)) & 1, c = i / (2 * T * 16);
    (&L.w3p[0][0][0])[i] = W3[c * WIDTH + x3_acc_feature(tr >> 4, tr & 15, h)];
  }
}

// One launch builds the weight image of the forward and, for a training step, the image the data-gradient kernel will want
// (the first 32 workgroups the one, the next 32 the other): the backward then starts without a prep launch of its own.
template <int WIDTH, int KS1>
__global__ void __launch_bounds__(256)
x3_prep_fwd_kernel(X3Img<WIDTH, KS1>* __restrict__ Lp, X3BwdImg<WIDTH>* __restrict__ Lb, const float* __restrict__ W1,
                   const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                   const float* __restrict__ W3, const float* __restrict__ b3, int D_in) {
  if (blockIdx.x < 32) x3_prep_fwd_body<WIDTH, KS1>(Lp, W1, b1, W2, b2, W3, b3, D_in, blockIdx.x * blockDim.x + threadIdx.x, 32 * blockDim.x);
  else x3_prep_bwd_body<WIDTH>(Lb, W1, W2, W3, D_in, (blockIdx.x - 32) * blockDim.x + threadIdx.x, 32 * blockDim.x);
}

template <int WIDTH>
__global__ void __launch_bounds__(256)
x3_prep_bwd_kernel(X3BwdImg<WIDTH>* __restrict__ Lp, const float* __restrict__ W1, const float* __restrict__ W2,
                   const float* __restrict__ W3, int D_in) {
  x3_prep_bwd_body<WIDTH>(Lp, W1, W2, W3, D_in, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <int WIDTH, bool DIFFUSE>
__global__ void __launch_bounds__(X3_BWD_THREADS)
shade_bwd_x3_kernel(const float* __restrict__ g_rgb, const float* __restrict__ rgb,
                    const unsigned long long* __restrict__ masks, int64_t M_cap, const int64_t* __restrict__ m_dev,
                    const void* __restrict__ image, int C, int c_view0,
                    int n_view, float* __restrict__ g_feat, float* __restrict__ G1, float* __restrict__ gz_out) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  __shared__ __attribute__((aligned(16))) X3BwdLds<WIDTH> L;
  x3_copy_image<X3BwdImg<WIDTH>>(&L, image);
  __syncthreads();
  const int lane = threadIdx.x & 63, h = lane >> 5;
  float* stage = L.stage[threadIdx.x >> 6];
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  // A wave walks its tiles alone, so the loads at the top of a tile are exposed once per tile: the per-row inputs of tile
  // i + 1 are requested at the top of tile i.  (Round 2 withdrew exactly this: results then differed from run to run -- the
  // packed-fp32 instructions of that build, profiles/r3/packed_f32.md; the library is built without them now and
  // test_shade_is_bitwise_repeatable holds this kernel to repeating bit for bit.)
  const int64_t last = M > 0 ? M - 1 : 0;
  float o_n[3], g_n[3];
  unsigned long long m1_n = 0ull, m2_n = 0ull;
  auto load_rows = [&](int64_t tile) {
    const int64_t r = tile * 32 + (lane & 31);
    const int64_t rc = r < M ? r : last;         // (past the last tile: a clamped row, never used)
#pragma unroll
    for (int c = 0; c < 3; ++c) { o_n[c] = rgb[rc * 3 + c]; g_n[c] = g_rgb[rc * 3 + c]; }
    m1_n = masks[(rc * 2 + 0) * 2 + h];
    m2_n = masks[(rc * 2 + 1) * 2 + h];
  };
  if (gw < n_tiles) load_rows(gw);
  for (int64_t tile = gw; tile < n_tiles; tile += nw) {
    const int64_t row = tile * 32 + (lane & 31);
    const bool valid = row < M;
    const int rows_valid = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);
    float gz[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) gz[c] = valid ? g_n[c] * o_n[c] * (1.0f - o_n[c]) : 0.0f;
    // ReLU sign bits of this lane's 64 features per layer (bit 16*t + r <-> feature f(t,r,h))
    const unsigned long long m1 = m1_n, m2 = m2_n;
    load_rows(tile + nw);
    if (valid && h == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gz_out[row * 3 + c] = gz[c];
        if (DIFFUSE) g_feat[row * C + c] = gz[c];
      }
    }
    // G2 in accumulator layout, split into the B fragments of the W2^T product as it is formed
    u32x4 g3[2 * T][3];
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = 8 * s2 + j;
          const float w = fmaf(L.w3p[2][h][t2 * 16 + r], gz[2],
                               fmaf(L.w3p[1][h][t2 * 16 + r], gz[1], L.w3p[0][h][t2 * 16 + r] * gz[0]));
          v[j] = ((m2 >> (16 * t2 + r)) & 1ull) ? w : 0.0f;       // gz == 0 on rows past M, so G2 == 0 there
        }
        x3_split8(v, g3[2 * t2 + s2][0], g3[2 * t2 + s2][1], g3[2 * t2 + s2][2]);
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
      for (int ks = 0; ks < 2 * T; ++ks) {
        x3_mfma6(acc, L.w2t[tin][ks][0][lane], L.w2t[tin][ks][1][lane], L.w2t[tin][ks][2][lane], g3[ks][0], g3[ks][1], g3[ks][2]);
        if (ks & 1) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned int mb = (unsigned int)(m1 >> (16 * tin + 4 * q)) & 15u;
        acc[4 * q + 0] = (mb & 1u) ? acc[4 * q + 0] : 0.0f;
        acc[4 * q + 1] = (mb & 2u) ? acc[4 * q + 1] : 0.0f;
        acc[4 * q + 2] = (mb & 4u) ? acc[4 * q + 2] : 0.0f;
        acc[4 * q + 3] = (mb & 8u) ? acc[4 * q + 3] : 0.0f;
      }
      x3_store_tile(stage, acc, G1 + tile * 32 * WIDTH + 32 * tin, WIDTH, lane, rows_valid);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = acc[8 * s2 + j];
        u32x4 b0, b1, b2;
        x3_split8(v, b0, b1, b2);
        const int ks = 2 * tin + s2;
        x3_mfma6(gx, L.w1t[ks][0][lane], L.w1t[ks][1][lane], L.w1t[ks][2][lane], b0, b1, b2);
      }
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
// Backward, weight-gradient part on the split operands:  dW2 = G2^T H1,  dW1 = G1^T X,  dW3 = gz^T H2,  db = column sums
// (shade.hip shade_wgrad_kernel; same inputs, same `part` record per workgroup, so the reduce kernel and the host code
// are shared).  The contraction runs over ROWS: a 32-row tile is two 16-deep k-steps, A[i = out feature][k = row] and
// B[k = row][j = in feature] are COLUMNS of the row-major operand tiles -- 8 rows of one column per lane -- so every
// streamed element is split on the fly (that is the price of this orientation: the kernel is bound by the split
// arithmetic and by the 1.5 KB / sample it streams, not by the matrix pipe).
//   * operand tiles (G1, H1, H2: 32 x WIDTH fp32 each) go global -> LDS by LDS-DMA, double buffered;
//   * wave w = (out-feature tile ot = w >> 1, k-step s = w & 1) builds ITS A fragments in registers (G2 rebuilt from gz
//     and the layer-2 sign bits as in shade.hip, G1 from the tile) and ONE of the 8 shared B fragments of H1 (+ the two of
//     X by waves 0 / 1), which travel through LDS to the waves that need them: nothing is split twice;
//   * 4 in-feature tiles + the X tile = 30 MFMAs per wave and tile; dW3, the <= 8 trailing columns of dW1 and the bias
//     sums are fp32 VALU on the unsplit values;
//   * the two k-step halves of an output tile are added through LDS once, at the end of the kernel.
// ----------------------------------------------------------------------------------
template <int WIDTH>
struct X3WgradLds {
  static constexpr int T = WIDTH / 32;
  float g1[2][32][WIDTH], h1[2][32][WIDTH], h2[2][32][WIDTH];
  float x[2][32][40];
  float gz[2][32][4];
  unsigned int m2[2][32][2][2];            // layer-2 sign bits [row][lane half of the forward][32-bit half]
  u32x4 fb[T + 1][2][3][64];               // shared B fragments: [in tile (T = the X tile)][k-step][piece][lane]
};

typedef const __attribute__((address_space(1))) void* x3_gptr_t;
typedef __attribute__((address_space(3))) void* x3_lptr_t;

template <int WIDTH>
__global__ void __launch_bounds__(WIDTH * 4)          // 2 * T waves: 512 threads for width 128, 256 for 64
shade_wgrad_x3_kernel(const float* __restrict__ G1, const float* __restrict__ gz, const unsigned int* __restrict__ masks,
                      const float* __restrict__ W3, const float* __restrict__ H1, const float* __restrict__ H2,
                      const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                      const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev,
                      float* __restrict__ part /* [gridDim][WIDTH*WIDTH + WIDTH*64 + 32*WIDTH + 3*WIDTH] */) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  constexpr int NW = 2 * T;                  // waves
  constexpr int NT = NW * 64;                // threads
  constexpr int TPR = NT / 32;               // staging threads per X row
  constexpr int NXI = 40 / TPR > 0 ? (40 + TPR - 1) / TPR : 1;    // X columns per staging thread
  constexpr int LPR = WIDTH / 4;             // lanes per operand row in a DMA instruction (16 B per lane)
  constexpr int RPI = 64 / LPR;              // rows per DMA wave instruction (1 KB)
  constexpr int IPW = 32 / RPI / NW;         // DMA instructions per wave and operand
  __shared__ __attribute__((aligned(16))) X3WgradLds<WIDTH> L;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, i = lane & 31, w = tid >> 6;
  const int ot = w >> 1, ks = w & 1;
  f32x16 aW2[T], aW1;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) aW1[r] = 0.0f;
  float vW3[3] = {0.0f, 0.0f, 0.0f}, vW1[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) vW1[kk] = 0.0f;
  float sb1 = 0.0f, sb2 = 0.0f, gz_acc = 0.0f;
  // this lane's out feature f = 32 ot + i of layer 2: its W3 column, and where its sign bit lives in the forward's
  // accumulator-order masks (f = 32t + (r&3) + 8(r>>2) + 4h'  ->  word h', bit 16t + r)
  const float w30 = W3[32 * ot + i], w31 = W3[WIDTH + 32 * ot + i], w32 = W3[2 * WIDTH + 32 * ot + i];
  const int m_half = (i >> 2) & 1, m_word = ot >> 1, m_bit = 16 * (ot & 1) + (i & 3) + 4 * (i >> 3);
  const int d_in = n_view + E;
  const int64_t n_tiles = (M + 31) / 32;

  float px[NXI], pgz;
  unsigned int pm2;
  bool pvalid, pgvalid;
  const int xrow = tid / TPR, xcol = tid - xrow * TPR;
  auto issue = [&](int64_t tile, int buf) {
    const int64_t r0 = tile * 32;
    {
      const int64_t row = r0 + xrow;
      const int64_t rc = row < M ? row : M - 1;
      const float* fr = feat + rc * C + c_view0;
      const float* er = emb + ray_id[rc] * E - n_view;
#pragma unroll
      for (int q = 0; q < NXI; ++q) {
        const int k = xcol + TPR * q;
        const int kc = k < d_in ? k : d_in - 1;
        px[q] = *((kc < n_view) ? fr + kc : er + kc);
      }
      pvalid = row < M;
    }
    {
      const int64_t row = r0 + ((tid & 127) >> 2);
      const int64_t rc = row < M ? row : M - 1;
      pgz = gz[rc * 3 + ((tid & 3) < 3 ? (tid & 3) : 0)];
      pgvalid = (tid & 3) < 3 && row < M;
      pm2 = masks[rc * 8 + 4 + (tid & 3)];
    }
#pragma unroll
    for (int q = 0; q < IPW; ++q) {
      const int rl = (IPW * w + q) * RPI;
      const int64_t row = r0 + rl + lane / LPR;
      const int64_t off = (row < M ? row : M - 1) * WIDTH + 4 * (lane % LPR);
      __builtin_amdgcn_global_load_lds((x3_gptr_t)(G1 + off), (x3_lptr_t)&L.g1[buf][rl][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((x3_gptr_t)(H1 + off), (x3_lptr_t)&L.h1[buf][rl][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((x3_gptr_t)(H2 + off), (x3_lptr_t)&L.h2[buf][rl][0], 16, 0, 0);
    }
  };

  int buf = 0;
  if ((int64_t)blockIdx.x < n_tiles) issue(blockIdx.x, 0);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
    // ---- the tile's small operands to LDS; its DMA has landed
#pragma unroll
    for (int q = 0; q < NXI; ++q) {
      const int k = xcol + TPR * q;
      if (k < 40) L.x[buf][xrow][k] = (pvalid && k < d_in) ? px[q] : 0.0f;
    }
    if (tid < 128) {
      gz_acc += pgvalid ? pgz : 0.0f;                     // db3[c] = sum of gz[:, c]
      L.gz[buf][tid >> 2][tid & 3] = pgvalid ? pgz : 0.0f;
      (&L.m2[buf][tid >> 2][0][0])[tid & 3] = pm2;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the DMA has landed
    __syncthreads();
    if (tile + gridDim.x < n_tiles) issue(tile + gridDim.x, buf ^ 1);       // next tile: in flight under this one
    const int rows_valid = (int)(M - tile * 32 < 32 ? M - tile * 32 : 32);  // wave-uniform
    // ---- shared B fragments: wave w builds H1 (in tile w & (T-1), k-step w / T); waves 0 / 1 also the X tile's
    {
      const int bt = w & (T - 1), bs = w / T;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = L.h1[buf][16 * bs + 8 * h + e][32 * bt + i];     // rows past M: multiplied by a zero A
      u32x4 p0, p1, p2;
      x3_split8(v, p0, p1, p2);
      L.fb[bt][bs][0][lane] = p0; L.fb[bt][bs][1][lane] = p1; L.fb[bt][bs][2][lane] = p2;
      if (w < 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = L.x[buf][16 * w + 8 * h + e][i];
        x3_split8(v, p0, p1, p2);
        L.fb[T][w][0][lane] = p0; L.fb[T][w][1][lane] = p1; L.fb[T][w][2][lane] = p2;
      }
    }
    // ---- this wave's A fragments (k-step ks): G2 rebuilt from gz and the sign bits, G1 from the tile; fp32 tails
    u32x4 a2f[3], a1f[3];
    {
      float a2[8], a1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = 16 * ks + 8 * h + e;
        const float4 gzr = *reinterpret_cast<const float4*>(&L.gz[buf][row][0]);     // 0 on rows past M
        const float g2v = fmaf(w32, gzr.z, fmaf(w31, gzr.y, w30 * gzr.x));
        a2[e] = ((L.m2[buf][row][m_half][m_word] >> m_bit) & 1u) ? g2v : 0.0f;
        a1[e] = row < rows_valid ? L.g1[buf][row][32 * ot + i] : 0.0f;            // rows past M hold a clamped copy
        sb2 += a2[e]; sb1 += a1[e];
        const float hv = L.h2[buf][row][32 * ot + i];
        vW3[0] = fmaf(gzr.x, hv, vW3[0]); vW3[1] = fmaf(gzr.y, hv, vW3[1]); vW3[2] = fmaf(gzr.z, hv, vW3[2]);
        const float4 xa = *reinterpret_cast<const float4*>(&L.x[buf][row][32]);     // zero past d_in
        const float4 xb = *reinterpret_cast<const float4*>(&L.x[buf][row][36]);
        vW1[0] = fmaf(a1[e], xa.x, vW1[0]); vW1[1] = fmaf(a1[e], xa.y, vW1[1]); vW1[2] = fmaf(a1[e], xa.z, vW1[2]);
        vW1[3] = fmaf(a1[e], xa.w, vW1[3]); vW1[4] = fmaf(a1[e], xb.x, vW1[4]); vW1[5] = fmaf(a1[e], xb.y, vW1[5]);
        vW1[6] = fmaf(a1[e], xb.z, vW1[6]); vW1[7] = fmaf(a1[e], xb.w, vW1[7]);
        if (e & 1) __builtin_amdgcn_sched_barrier(0);     // two rows' worth of LDS reads in flight, not all eight
      }
      x3_split8(a2, a2f[0], a2f[1], a2f[2]);
      x3_split8(a1, a1f[0], a1f[1], a1f[2]);
    }
    __syncthreads();                        // the B fragments are complete
#pragma unroll
    for (int t = 0; t < T; ++t) {
      x3_mfma6(aW2[t], a2f[0], a2f[1], a2f[2], L.fb[t][ks][0][lane], L.fb[t][ks][1][lane], L.fb[t][ks][2][lane]);
      __builtin_amdgcn_sched_barrier(0);      // one fragment triple in registers at a time
    }
    x3_mfma6(aW1, a1f[0], a1f[1], a1f[2], L.fb[T][ks][0][lane], L.fb[T][ks][1][lane], L.fb[T][ks][2][lane]);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  // ---- the two k-step halves of each output tile meet in LDS (reusing the operand buffers), then the record
  float* red = &L.g1[0][0][0];              // (T + 1) * 16 * 64 floats per out tile: 80 KB at width 128 (g1, h1, h2 are contiguous)
  if (ks == 1) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((ot * (T + 1) + t) * 16 + r) * 64 + lane] = aW2[t][r];
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((ot * (T + 1) + T) * 16 + r) * 64 + lane] = aW1[r];
  }
  static_assert((T * (T + 1) * 16 * 64 + T * 14 * 64) * 4 <= 3 * 2 * 32 * WIDTH * 4, "the exchange fits the operand buffers");
  float* red2 = red + T * (T + 1) * 16 * 64;   // the VALU sums of the ks == 1 waves: 14 floats per lane, behind `red`
  if (ks == 1) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) red2[(ot * 14 + kk) * 64 + lane] = vW1[kk];
#pragma unroll
    for (int c = 0; c < 3; ++c) red2[(ot * 14 + 8 + c) * 64 + lane] = vW3[c];
    red2[(ot * 14 + 11) * 64 + lane] = sb1;
    red2[(ot * 14 + 12) * 64 + lane] = sb2;
  }
  __syncthreads();
  float* p = part + (int64_t)blockIdx.x * (WIDTH * WIDTH + WIDTH * 64 + 32 * WIDTH + 3 * WIDTH);
  float* pW2 = p;                          // [WIDTH out][WIDTH in]
  float* pW1 = pW2 + WIDTH * WIDTH;        // [WIDTH out][64]
  float* pW3 = pW1 + WIDTH * 64;           // [32 (c padded)][WIDTH], rows 0..2 written
  float* pb = pW3 + 32 * WIDTH;            // [3][WIDTH]: db1, db2, db3 (entries [0,3) + [8,11))
  if (ks == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oi = (r & 3) + 8 * (r >> 2) + 4 * h;       // D[row = out feature oi][col = in feature i]
#pragma unroll
      for (int t = 0; t < T; ++t)
        pW2[(32 * ot + oi) * WIDTH + 32 * t + i] = aW2[t][r] + red[((ot * (T + 1) + t) * 16 + r) * 64 + lane];
      pW1[(32 * ot + oi) * 64 + i] = aW1[r] + red[((ot * (T + 1) + T) * 16 + r) * 64 + lane];
    }
    // VALU parts: this lane's out feature is 32 ot + i; lane halves and the partner wave hold other rows
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float v = vW1[kk] + red2[(ot * 14 + kk) * 64 + lane];
      v += __shfl_xor(v, 32);
      if (h == 0) pW1[(32 * ot + i) * 64 + 32 + kk] = v;
    }
    if (h == 0) {
#pragma unroll
      for (int kk = 8; kk < 32; ++kk) pW1[(32 * ot + i) * 64 + 32 + kk] = 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = vW3[c] + red2[(ot * 14 + 8 + c) * 64 + lane];
      v += __shfl_xor(v, 32);
      if (h == 0) pW3[c * WIDTH + 32 * ot + i] = v;
    }
    float s1 = sb1 + red2[(ot * 14 + 11) * 64 + lane], s2 = sb2 + red2[(ot * 14 + 12) * 64 + lane];
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if (h == 0) { pb[32 * ot + i] = s1; pb[WIDTH + 32 * ot + i] = s2; }
  }
  // db3[c]: the staging threads 4*row + c (tid < 128: waves 0 and 1) hold per-row sums; fold the row bits of the lane
  float gz_sum = gz_acc;
  gz_sum += __shfl_xor(gz_sum, 4); gz_sum += __shfl_xor(gz_sum, 8);
  gz_sum += __shfl_xor(gz_sum, 16); gz_sum += __shfl_xor(gz_sum, 32);
  // record layout shared with shade.hip: db3 = entries [0,3) + [8,11) of the third bias row, everything else there zero
  if (w < 2 && lane < 3) pb[2 * WIDTH + 8 * w + lane] = gz_sum;
  for (int q = tid; q < WIDTH; q += NT)
    if (q >= 16 || (q & 7) >= 3) pb[2 * WIDTH + q] = 0.0f;
}

// ----------------------------------------------------------------------------------
// Weight gradients on the split operands, second form (round 3; dvgo_shade_variant bit 4 picks it over the one above).
// Same arithmetic, same `part` record; what changes is who waits for whom.  The first form is one 8-wave workgroup per CU
// (141 KB of LDS: double-buffered operand tiles) whose waves meet at two barriers per 32-row tile with ~2,800 cycles of work
// in between -- less than a DMA round trip -- and it ends at the time of the f32-MFMA kernel although it issues a third of
// its matrix cycles.  Here a workgroup is T = 4 waves, wave w owns out-feature tile w for BOTH k-steps (no exchange of
// partial tiles at the end), the operand tiles are single-buffered (G1, H1; H2 is only needed column-wise for dW3 and is
// read straight from global memory), which leaves 68 KB of LDS per workgroup: TWO workgroups per CU, each covering the
// other's DMA round trip and barriers.
// ----------------------------------------------------------------------------------
template <int WIDTH>
struct X3WgradLdsB {
  static constexpr int T = WIDTH / 32;
  float g1[32][WIDTH], h1[32][WIDTH];
  float x[32][40];
  float gz[32][4];
  unsigned int m2[32][2][2];               // layer-2 sign bits [row][lane half of the forward][32-bit half]
  u32x4 fb[T + 1][2][3][64];               // shared B fragments: [in tile (T = the X tile)][k-step][piece][lane]
};

template <int WIDTH>
__global__ void __launch_bounds__(WIDTH * 2, 2)       // T waves; two workgroups per CU
shade_wgrad_x3b_kernel(const float* __restrict__ G1, const float* __restrict__ gz, const unsigned int* __restrict__ masks,
                       const float* __restrict__ W3, const float* __restrict__ H1, const float* __restrict__ H2,
                       const float* __restrict__ feat, int C, int c_view0, int n_view, const float* __restrict__ emb, int E,
                       const int64_t* __restrict__ ray_id, int64_t M_cap, const int64_t* __restrict__ m_dev,
                       float* __restrict__ part /* [gridDim][WIDTH*WIDTH + WIDTH*64 + 32*WIDTH + 3*WIDTH] */) {
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  constexpr int T = WIDTH / 32;
  constexpr int NW = T;                      // waves
  constexpr int NT = NW * 64;                // threads
  constexpr int TPR = NT / 32;               // staging threads per X row
  constexpr int NXI = (40 + TPR - 1) / TPR;  // X columns per staging thread
  constexpr int LPR = WIDTH / 4;             // lanes per operand row in a DMA instruction (16 B per lane)
  constexpr int RPI = 64 / LPR;              // rows per DMA wave instruction (1 KB)
  constexpr int IPW = 32 / RPI / NW;         // DMA instructions per wave and operand
  __shared__ __attribute__((aligned(16))) X3WgradLdsB<WIDTH> L;
  static_assert(sizeof(X3WgradLdsB<WIDTH>) <= 80 * 1024, "two workgroups per CU");
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, i = lane & 31, w = tid >> 6;
  const int ot = w;
  f32x16 aW2[T], aW1;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) aW1[r] = 0.0f;
  float vW3[3] = {0.0f, 0.0f, 0.0f}, vW1[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) vW1[kk] = 0.0f;
  float sb1 = 0.0f, sb2 = 0.0f, gz_acc = 0.0f;
  const float w30 = W3[32 * ot + i], w31 = W3[WIDTH + 32 * ot + i], w32 = W3[2 * WIDTH + 32 * ot + i];
  const int m_half = (i >> 2) & 1, m_word = ot >> 1, m_bit = 16 * (ot & 1) + (i & 3) + 4 * (i >> 3);
  const int d_in = n_view + E;
  const int64_t n_tiles = (M + 31) / 32;
  const int xrow = tid / TPR, xcol = tid - xrow * TPR;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t r0 = tile * 32;
    // ---- this tile's operands: G1 / H1 by LDS-DMA, the small ones through registers
#pragma unroll
    for (int q = 0; q < IPW; ++q) {
      const int rl = (IPW * w + q) * RPI;
      const int64_t row = r0 + rl + lane / LPR;
      const int64_t off = (row < M ? row : M - 1) * WIDTH + 4 * (lane % LPR);
      __builtin_amdgcn_global_load_lds((x3_gptr_t)(G1 + off), (x3_lptr_t)&L.g1[rl][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((x3_gptr_t)(H1 + off), (x3_lptr_t)&L.h1[rl][0], 16, 0, 0);
    }
    {
      const int64_t row = r0 + xrow;
      const int64_t rc = row < M ? row : M - 1;
      const float* fr = feat + rc * C + c_view0;
      const float* er = emb + ray_id[rc] * E - n_view;
#pragma unroll
      for (int q = 0; q < NXI; ++q) {
        const int k = xcol + TPR * q;
        const int kc = k < d_in ? k : d_in - 1;
        const float v = *((kc < n_view) ? fr + kc : er + kc);
        if (k < 40) L.x[xrow][k] = (row < M && k < d_in) ? v : 0.0f;
      }
    }
    if (tid < 128) {
      const int64_t row = r0 + (tid >> 2);
      const int64_t rc = row < M ? row : M - 1;
      const bool ok = (tid & 3) < 3 && row < M;
      const float g = gz[rc * 3 + ((tid & 3) < 3 ? (tid & 3) : 0)];
      gz_acc += ok ? g : 0.0f;                            // db3[c] = sum of gz[:, c]
      L.gz[tid >> 2][tid & 3] = ok ? g : 0.0f;
      (&L.m2[tid >> 2][0][0])[tid & 3] = masks[rc * 8 + 4 + (tid & 3)];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's share of the DMA has landed
    __syncthreads();
    const int rows_valid = (int)(M - r0 < 32 ? M - r0 : 32);  // wave-uniform
    // ---- shared B fragments: wave w builds H1's in tile w for both k-steps; waves 0 / 1 also the X tile's
    {
      u32x4 p0, p1, p2;
      float v[8];
#pragma unroll
      for (int bs = 0; bs < 2; ++bs) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = L.h1[16 * bs + 8 * h + e][32 * w + i];      // rows past M: multiplied by a zero A
        x3_split8(v, p0, p1, p2);
        L.fb[w][bs][0][lane] = p0; L.fb[w][bs][1][lane] = p1; L.fb[w][bs][2][lane] = p2;
      }
      if (w < 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = L.x[16 * w + 8 * h + e][i];
        x3_split8(v, p0, p1, p2);
        L.fb[T][w][0][lane] = p0; L.fb[T][w][1][lane] = p1; L.fb[T][w][2][lane] = p2;
      }
    }
    __syncthreads();                        // the B fragments are complete
#pragma unroll 1
    for (int ks = 0; ks < 2; ++ks) {
      // ---- this wave's A fragments of k-step ks: G2 rebuilt from gz and the sign bits, G1 from the tile; fp32 tails
      u32x4 a2f[3], a1f[3];
      {
        float hv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {          // H2 column of this lane's out feature: straight from global memory
          const int64_t row = r0 + 16 * ks + 8 * h + e;
          hv[e] = H2[(row < M ? row : M - 1) * WIDTH + 32 * ot + i];
        }
        float a2[8], a1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int row = 16 * ks + 8 * h + e;
          const float4 gzr = *reinterpret_cast<const float4*>(&L.gz[row][0]);     // 0 on rows past M
          const float g2v = fmaf(w32, gzr.z, fmaf(w31, gzr.y, w30 * gzr.x));
          a2[e] = ((L.m2[row][m_half][m_word] >> m_bit) & 1u) ? g2v : 0.0f;
          a1[e] = row < rows_valid ? L.g1[row][32 * ot + i] : 0.0f;            // rows past M hold a clamped copy
          sb2 += a2[e]; sb1 += a1[e];
          vW3[0] = fmaf(gzr.x, hv[e], vW3[0]); vW3[1] = fmaf(gzr.y, hv[e], vW3[1]); vW3[2] = fmaf(gzr.z, hv[e], vW3[2]);
          const float4 xa = *reinterpret_cast<const float4*>(&L.x[row][32]);     // zero past d_in
          const float4 xb = *reinterpret_cast<const float4*>(&L.x[row][36]);
          vW1[0] = fmaf(a1[e], xa.x, vW1[0]); vW1[1] = fmaf(a1[e], xa.y, vW1[1]); vW1[2] = fmaf(a1[e], xa.z, vW1[2]);
          vW1[3] = fmaf(a1[e], xa.w, vW1[3]); vW1[4] = fmaf(a1[e], xb.x, vW1[4]); vW1[5] = fmaf(a1[e], xb.y, vW1[5]);
          vW1[6] = fmaf(a1[e], xb.z, vW1[6]); vW1[7] = fmaf(a1[e], xb.w, vW1[7]);
          if (e & 1) __builtin_amdgcn_sched_barrier(0);     // two rows' worth of LDS reads in flight, not all eight
        }
        x3_split8(a2, a2f[0], a2f[1], a2f[2]);
        x3_split8(a1, a1f[0], a1f[1], a1f[2]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        x3_mfma6(aW2[t], a2f[0], a2f[1], a2f[2], L.fb[t][ks][0][lane], L.fb[t][ks][1][lane], L.fb[t][ks][2][lane]);
        __builtin_amdgcn_sched_barrier(0);      // one fragment triple in registers at a time
      }
      x3_mfma6(aW1, a1f[0], a1f[1], a1f[2], L.fb[T][ks][0][lane], L.fb[T][ks][1][lane], L.fb[T][ks][2][lane]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                        // every read of this tile's LDS is done: the next tile may land
  }
  // ---- the record (layout of the first form: the reduce kernel and the host code are shared)
  float* p = part + (int64_t)blockIdx.x * (WIDTH * WIDTH + WIDTH * 64 + 32 * WIDTH + 3 * WIDTH);
  float* pW2 = p;                          // [WIDTH out][WIDTH in]
  float* pW1 = pW2 + WIDTH * WIDTH;        // [WIDTH out][64]
  float* pW3 = pW1 + WIDTH * 64;           // [32 (c padded)][WIDTH], rows 0..2 written
  float* pb = pW3 + 32 * WIDTH;            // [3][WIDTH]: db1, db2, db3 (entries [0,3) + [8,11))
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int oi = (r & 3) + 8 * (r >> 2) + 4 * h;       // D[row = out feature oi][col = in feature i]
#pragma unroll
    for (int t = 0; t < T; ++t) pW2[(32 * ot + oi) * WIDTH + 32 * t + i] = aW2[t][r];
    pW1[(32 * ot + oi) * 64 + i] = aW1[r];
  }
  // VALU parts: this lane's out feature is 32 ot + i; the two lane halves hold different rows
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    float v = vW1[kk];
    v += __shfl_xor(v, 32);
    if (h == 0) pW1[(32 * ot + i) * 64 + 32 + kk] = v;
  }
  if (h == 0) {
#pragma unroll
    for (int kk = 8; kk < 32; ++kk) pW1[(32 * ot + i) * 64 + 32 + kk] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = vW3[c];
    v += __shfl_xor(v, 32);
    if (h == 0) pW3[c * WIDTH + 32 * ot + i] = v;
  }
  float s1 = sb1, s2 = sb2;
  s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
  if (h == 0) { pb[32 * ot + i] = s1; pb[WIDTH + 32 * ot + i] = s2; }
  // db3[c]: the staging threads 4*row + c (tid < 128: waves 0 and 1) hold per-row sums; fold the row bits of the lane
  float gz_sum = gz_acc;
  gz_sum += __shfl_xor(gz_sum, 4); gz_sum += __shfl_xor(gz_sum, 8);
  gz_sum += __shfl_xor(gz_sum, 16); gz_sum += __shfl_xor(gz_sum, 32);
  if (w < 2 && lane < 3) pb[2 * WIDTH + 8 * w + lane] = gz_sum;
  for (int q = tid; q < WIDTH; q += NT)
    if (q >= 16 || (q & 7) >= 3) pb[2 * WIDTH + q] = 0.0f;
}

// ----------------------------------------------------------------------------------
// Weight gradients on the split operands, third form (round 3; dvgo_shade_variant bit 5): NO LDS, NO barriers.
// Forms one and two hand the split B fragments from wave to wave through LDS, which costs two barriers per 32-row tile and
// a DMA round trip the waves cannot cover (both end where the f32-MFMA kernel ends, with a third of its matrix cycles).
// Here every wave is on its own: wave w owns the out-feature tile w for the whole launch and takes EVERY operand straight
// from global memory in MFMA fragment order -- the contraction runs over rows, so lane (feature i, half h) loads 8 rows
// of its feature column with 8 dword loads (32 lanes x 4 B = one 128-B line per row: coalesced) -- and splits it itself.
// The B operand (H1 / X) is split by all T waves of a workgroup (the redundancy costs VALU slots the matrix pipe leaves
// free: per 16-row k-step ~300 VALU beside 24 MFMAs) and loaded by all of them (the second to fourth request hit L1 / L2).
// Raw operands of k-step n + 1 are requested into the registers k-step n has just consumed, so ~56 loads per wave are in
// flight the whole time and nothing ever waits for anything else.  The work is cut in two kernels by operand set:
//   c2: dW2 = G2^T H1 (G2 rebuilt from gz and the layer-2 sign bits), dW3 = gz^T H2, db2, db3     reads H1, H2, gz, masks
//   c1: dW1 = G1^T X, db1                                                                         reads G1, feat, emb, ray_id
// Both write the `part` record of the other forms (each its own entries).  Rows past M: every buffer descriptor is sized
// by the M valid rows, so they arrive as zeros (gz = 0 -> G2 = 0; G1 = 0).
// ----------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(3))) unsigned int u32x3;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define X3C_PART(W) ((W) * (W) + (W) * 64 + 32 * (W) + 3 * (W))

template <int WIDTH, int EXP = 0>                      // EXP: switch-off experiments of round 3 (1: no MFMAs, 2: no re-requests, 4: no H1 splits; profiles/r3/wgrad_c2_switch_off.txt)
__global__ void __launch_bounds__(WIDTH * 2, 2)       // T waves; two workgroups per CU
shade_wgrad_c2_kernel(const float* __restrict__ gz, const unsigned int* __restrict__ masks, const float* __restrict__ W3,
                      const float* __restrict__ H1, const float* __restrict__ H2, int64_t M_cap,
                      const int64_t* __restrict__ m_dev, float* __restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)                   // (buffer descriptors are device-only types)
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;
  constexpr int T = WIDTH / 32;
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int ot = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned int Mu = (unsigned int)M;
  const auto bH1 = __builtin_amdgcn_make_buffer_rsrc((void*)H1, 0, Mu * (WIDTH * 4u), 0x00020000);
  const auto bH2 = __builtin_amdgcn_make_buffer_rsrc((void*)H2, 0, Mu * (WIDTH * 4u), 0x00020000);
  const auto bGz = __builtin_amdgcn_make_buffer_rsrc((void*)gz, 0, Mu * 12u, 0x00020000);
  const auto bM = __builtin_amdgcn_make_buffer_rsrc((void*)masks, 0, Mu * 32u, 0x00020000);
  f32x16 aW2[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) aW2[t][r] = 0.0f;
  float vW3[3] = {0.0f, 0.0f, 0.0f}, g3[3] = {0.0f, 0.0f, 0.0f}, sb2 = 0.0f;
  // this lane's out feature f = 32 ot + i of layer 2: its W3 column, and where its sign bit lives in the forward's
  // accumulator-order masks (f = 32t + (r&3) + 8(r>>2) + 4h'  ->  64-bit word h', bit 16t + r)
  const float w30 = W3[32 * ot + i], w31 = W3[WIDTH + 32 * ot + i], w32 = W3[2 * WIDTH + 32 * ot + i];
  const unsigned int m_bit = 16 * (ot & 1) + (i & 3) + 4 * (i >> 3);
  // lane-constant byte offsets; the k-step (16 rows) and the wave's feature tile ride in the scalar offset, row e of the
  // lane's 8 and the in-feature tile in the instruction's immediate
  int vo_h = (8 * h * WIDTH + i) * 4;
  int vo_hw = (8 * h * WIDTH + T * i) * 4;
  int vo_gz = 8 * h * 12;
  int vo_m = (8 * h * 8 + 4 + 2 * ((i >> 2) & 1) + (ot >> 1)) * 4;
  const int n_ks = (int)((M + 15) >> 4);
  float rh1[T][8], rh2[8];
  u32x3 rgz[8];
  unsigned int rm[8];
  auto ld_small = [&](int k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      rgz[e] = __builtin_amdgcn_raw_buffer_load_b96(bGz, vo_gz + e * 12, k * (16 * 12), 0);
      rm[e] = __builtin_amdgcn_raw_buffer_load_b32(bM, vo_m + e * 32, k * (16 * 32), 0);
      rh2[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bH2, vo_h + e * (WIDTH * 4), k * (16 * WIDTH * 4) + ot * 128, 0));
    }
  };
  // H1: ONE request per row brings this lane's column of all T in-feature tiles -- column i of tile t is in-feature
  // T i + t, T consecutive floats -- so a row's 512 bytes leave as one 16-byte-per-lane request instead of four 4-byte ones
  // (the texture addresser, not the ALUs, bounded the first version: TA busy 92 %, 8.8 cycles per dword request).
  auto ld_h1 = [&](int e, int k) {
    if constexpr (T == 4) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(bH1, vo_hw + e * (WIDTH * 4), k * (16 * WIDTH * 4), 0);
#pragma unroll
      for (int t = 0; t < T; ++t) rh1[t][e] = __uint_as_float(v[t]);
    } else {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(bH1, vo_hw + e * (WIDTH * 4), k * (16 * WIDTH * 4), 0);
#pragma unroll
      for (int t = 0; t < T; ++t) rh1[t][e] = __uint_as_float(v[t]);
    }
  };
  int ks = blockIdx.x;
  if (ks < n_ks) {
    // first requests, in the order of the loop's (its waits count requests; guard, requests and fences in ONE block: requests
    // hoisted above the guard are sunk below it again, past their fences, and re-ordered there)
    __builtin_amdgcn_sched_barrier(0);
    ld_small(ks);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 8; ++e) ld_h1(e, ks);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; ks < n_ks; ks += gridDim.x) {
    const int kn = ks + (int)gridDim.x < n_ks ? ks + (int)gridDim.x : n_ks - 1;     // (the last k-step is fetched once more, unused)
    // The lane offsets are made opaque once per trip: derived inside the loop, `offset + constant` folds into the load's
    // immediate; hoisted out of it, every one of the 56 loads keeps an address register of its own.  The fences keep
    // each group of requests where it is written -- right behind the last use of the registers it refills; left alone
    // the scheduler sinks all of them to the end of the trip, where the next trip waits for them at once.
    asm volatile("" : "+v"(vo_h), "+v"(vo_hw), "+v"(vo_gz), "+v"(vo_m));
    float a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g0 = __uint_as_float(rgz[e][0]), g1 = __uint_as_float(rgz[e][1]), g2 = __uint_as_float(rgz[e][2]);
      const float g2v = fmaf(w32, g2, fmaf(w31, g1, w30 * g0));
      a2[e] = __uint_as_float(__float_as_uint(g2v) & (unsigned int)__builtin_amdgcn_sbfe((int)rm[e], m_bit, 1u));
      sb2 += a2[e];
      const float hv = rh2[e];
      vW3[0] = fmaf(g0, hv, vW3[0]); vW3[1] = fmaf(g1, hv, vW3[1]); vW3[2] = fmaf(g2, hv, vW3[2]);
    }
    if (ot == 0) {                                   // db3 = column sums of gz (wave 0; the lanes of a half agree)
      asm volatile("" : "+v"(g3[0]), "+v"(g3[1]), "+v"(g3[2]));     // (a real branch, not 24 selects in every wave)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        g3[0] += __uint_as_float(rgz[e][0]); g3[1] += __uint_as_float(rgz[e][1]); g3[2] += __uint_as_float(rgz[e][2]);
      }
    }
    // every value derived from the raw registers exists before they are requested again (plain arithmetic is not ordered
    // by a scheduling fence: it is placed when the block is linearised, and landed behind the MFMAs)
    asm volatile("" : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]),
                      "+v"(vW3[0]), "+v"(vW3[1]), "+v"(vW3[2]), "+v"(sb2), "+v"(g3[0]), "+v"(g3[1]), "+v"(g3[2]));
    if (!(EXP & 2)) ld_small(kn);
    __builtin_amdgcn_sched_barrier(0);
    u32x4 a2f[3];
    x3_split8(a2, a2f[0], a2f[1], a2f[2]);
    // Tile t's six MFMAs are issued one by one with the split of tile t + 1 in between (group barriers: 1 MFMA, then 8
    // of the split's ~46 VALU instructions): a wave's two phases otherwise alternate -- ~1,300 cycles of VALU only, then 24
    // MFMAs back to back -- and two waves on a SIMD fall into step instead of covering each other (measured: a k-step pair
    // took the SUM of the two waves' VALU and MFMA time).  The raw H1 registers are requested again behind the last split.
    u32x4 bf[2][3];
    {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rh1[0][e];
      x3_split8_free(v, bf[0][0], bf[0][1], bf[0][2]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = rh1[t + 1][e];
        if (!(EXP & 4)) x3_split8_free(v, bf[(t + 1) & 1][0], bf[(t + 1) & 1][1], bf[(t + 1) & 1][2]);
        else {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) bf[(t + 1) & 1][q][c] = __float_as_uint(v[(c + q) & 7]) ^ __float_as_uint(v[c + 4]);
        }
      }
      if (!(EXP & 1)) x3_mfma6(aW2[t], a2f[0], a2f[1], a2f[2], bf[t & 1][0], bf[t & 1][1], bf[t & 1][2]);
      else { aW2[t][0] += __uint_as_float(bf[t & 1][0][0] ^ bf[t & 1][1][1] ^ bf[t & 1][2][2] ^ a2f[0][0] ^ a2f[1][0] ^ a2f[2][0]); }
      if (t + 1 < T) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // eight VALU
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 == T) {                // every raw H1 register has been consumed
        asm volatile("" : "+v"(bf[(t + 1) & 1][0]), "+v"(bf[(t + 1) & 1][1]), "+v"(bf[(t + 1) & 1][2]));
        if (!(EXP & 2)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ld_h1(e, kn);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float* p = part + (int64_t)blockIdx.x * X3C_PART(WIDTH);
  float* pW2 = p;                               // [WIDTH out][WIDTH in]
  float* pW3 = pW2 + WIDTH * WIDTH + WIDTH * 64;   // [32 (c padded)][WIDTH], rows 0..2 written
  float* pb = pW3 + 32 * WIDTH;                 // [3][WIDTH]: db1, db2, db3 (entries [0,3) + [8,11))
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int oi = (r & 3) + 8 * (r >> 2) + 4 * h;       // D[row = out feature oi][col = in feature i]
#pragma unroll
    for (int t = 0; t < T; ++t) pW2[(32 * ot + oi) * WIDTH + T * i + t] = aW2[t][r];      // (column i of tile t = in-feature T i + t)
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = vW3[c];
    v += __shfl_xor(v, 32);
    if (h == 0) pW3[c * WIDTH + 32 * ot + i] = v;
  }
  float s2 = sb2;
  s2 += __shfl_xor(s2, 32);
  if (h == 0) pb[WIDTH + 32 * ot + i] = s2;
  // third bias row: db3 in entries 0..2 (the record's second share, entries 8..10, stays zero), zeros elsewhere
  if (ot == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = g3[c] + __shfl_xor(g3[c], 32);
      if (lane == 0) pb[2 * WIDTH + c] = v;
    }
  }
  for (int q = threadIdx.x; q < WIDTH; q += WIDTH * 2)
    if (q >= 3) pb[2 * WIDTH + q] = 0.0f;
#endif
}

template <int WIDTH>
__global__ void __launch_bounds__(256, 2)             // 4 waves = 2 row groups x 2 halves of the out features; two workgroups per CU
shade_wgrad_c1_kernel(const float* __restrict__ G1, const float* __restrict__ feat, int C, int c_view0, int n_view,
                      const float* __restrict__ emb, int E, const int64_t* __restrict__ ray_id, int64_t M_cap,
                      const int64_t* __restrict__ m_dev, float* __restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)
  // Here the B operand (X: 39 of 64 columns, gathered from two sources) is what every wave would split again, and its
  // 4-byte requests are what keeps the texture addresser busy: so a wave takes HALF the out features (TA = T / 2 tiles, one
  // 8- or 4-byte request per row: column i of tile t is out-feature T i + t) and the two row groups of a workgroup walk
  // different k-steps; their sums meet in LDS at the end.
  const int64_t M = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;
  constexpr int T = WIDTH / 32, TA = T / 2;
  __shared__ float red[2][TA * 2 * 16 * 64 + 64 * TA];      // [half][acc register][lane], + the bias sums
  const int lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = w & 1, grp = w >> 1;
  const unsigned int Mu = (unsigned int)M;
  const auto bG1 = __builtin_amdgcn_make_buffer_rsrc((void*)G1, 0, Mu * (WIDTH * 4u), 0x00020000);
  const auto bF = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, Mu * (unsigned int)(C * 4), 0x00020000);
  const auto bE = __builtin_amdgcn_make_buffer_rsrc((void*)emb, 0, 0x80000000u, 0x00020000);      // ray count not known here
  const auto bR = __builtin_amdgcn_make_buffer_rsrc((void*)ray_id, 0, Mu * 8u, 0x00020000);
  f32x16 aW1[TA][2];
  float sb1[TA];
#pragma unroll
  for (int t = 0; t < TA; ++t) {
    sb1[t] = 0.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) aW1[t][q][r] = 0.0f;
  }
  const int d_in = n_view + E;
  // X[row][k]: k < n_view: feat[row, c_view0 + k] (lib/dvgo.py:518-523); k < d_in: emb[ray_id[row], k - n_view]
  // (lib/dvgo.py:524-526).  Lane i serves column i of the first X tile and column 32 + i of the second; every lane loads
  // from a valid address (columns clamped into their source) and columns >= d_in, whatever they hold, only reach columns
  // of the product the reduction drops.
  const bool from_feat = i < n_view;
  const int kf = i < n_view ? i : (n_view > 0 ? n_view - 1 : 0);    // feature-grid column of the first tile
  const int ke0 = (i < n_view ? n_view : (i < d_in ? i : d_in - 1)) - n_view;       // embedding column, first tile
  const int ke1 = (32 + i < d_in ? 32 + i : d_in - 1) - n_view;                     // embedding column, second tile
  int vo_g = (8 * h * WIDTH + T * i + TA * half) * 4;
  int vo_f = (8 * h * C + c_view0 + kf) * 4;
  int vo_r = 8 * h * 8;
  const int n_ks = (int)((M + 15) >> 4);
  const int stride = 2 * (int)gridDim.x;
  float rg[TA][8], rf[8], re0[8], re1[8];
  int rid[8];
  auto ld_rid = [&](int k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) rid[e] = __builtin_amdgcn_raw_buffer_load_b32(bR, vo_r + e * 8, k * (16 * 8), 0);   // low word: ray ids < 2^31
  };
  auto ld_g = [&](int k) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (TA == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(bG1, vo_g + e * (WIDTH * 4), k * (16 * WIDTH * 4), 0);
        rg[0][e] = __uint_as_float(v[0]); rg[1][e] = __uint_as_float(v[1]);
      } else {
        rg[0][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bG1, vo_g + e * (WIDTH * 4), k * (16 * WIDTH * 4), 0));
      }
    }
  };
  auto ld_x = [&](int k) {          // needs rid of k-step k
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      rf[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bF, vo_f + e * (C * 4), k * (16 * C * 4), 0));
      const int vo_e = rid[e] * (E * 4);
      re0[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bE, vo_e + ke0 * 4, 0, 0));
      re1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bE, vo_e + ke1 * 4, 0, 0));
    }
  };
  auto clampk = [&](int k) { return k < n_ks ? k : (n_ks > 0 ? n_ks - 1 : 0); };
  int ks = 2 * (int)blockIdx.x + grp;
  if (ks < n_ks) {                  // (guard, requests and fences in one block: see c2)
    __builtin_amdgcn_sched_barrier(0);
    ld_rid(ks);
    __builtin_amdgcn_sched_barrier(0);
    ld_g(ks);
    __builtin_amdgcn_sched_barrier(0);
    ld_x(ks);
    __builtin_amdgcn_sched_barrier(0);
    ld_rid(clampk(ks + stride));
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; ks < n_ks; ks += stride) {
    const int kn = clampk(ks + stride), kn2 = clampk(ks + 2 * stride);
    asm volatile("" : "+v"(vo_g), "+v"(vo_f), "+v"(vo_r));      // (see c2)
    u32x4 af[TA][3];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      float a1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { a1[e] = rg[t][e]; sb1[t] += a1[e]; }
      x3_split8_free(a1, af[t][0], af[t][1], af[t][2]);
    }
    if constexpr (TA == 2)
      asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(sb1[0]), "+v"(sb1[1]));
    else
      asm volatile("" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(sb1[0]));
    ld_g(kn);
    __builtin_amdgcn_sched_barrier(0);
    // both X tiles are split before their registers are requested again (a raw value still live behind the request
    // costs a second register and a copy that waits for the request at the end of the trip)
    u32x4 b0, b1, b2, c0, c1, c2;
    {
      float x0[8], x1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { x0[e] = from_feat ? rf[e] : re0[e]; x1[e] = re1[e]; }
      x3_split8(x0, b0, b1, b2);
      x3_split8(x1, c0, c1, c2);
    }
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c0), "+v"(c1), "+v"(c2));
    ld_x(kn);                       // rid holds k-step kn's ray ids (requested one k-step ago)
    __builtin_amdgcn_sched_barrier(0);
    ld_rid(kn2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      x3_mfma6(aW1[t][0], af[t][0], af[t][1], af[t][2], b0, b1, b2);
      x3_mfma6(aW1[t][1], af[t][0], af[t][1], af[t][2], c0, c1, c2);
    }
  }
  // the second row group hands its sums to the first through LDS
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < TA; ++t) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[half][((t * 2 + q) * 16 + r) * 64 + lane] = aW1[t][q][r];
      red[half][TA * 2 * 16 * 64 + 64 * t + lane] = sb1[t];
    }
  }
  __syncthreads();
  if (grp == 0) {
    float* p = part + (int64_t)blockIdx.x * X3C_PART(WIDTH);
    float* pW1 = p + WIDTH * WIDTH;               // [WIDTH out][64]
    float* pb = pW1 + WIDTH * 64 + 32 * WIDTH;    // [3][WIDTH]: db1 first
#pragma unroll
    for (int t = 0; t < TA; ++t) {
      const int tile = TA * half + t;             // row oi of tile `tile` is out-feature T oi + tile
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oi = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int q = 0; q < 2; ++q)
          pW1[(T * oi + tile) * 64 + 32 * q + i] = aW1[t][q][r] + red[half][((t * 2 + q) * 16 + r) * 64 + lane];
      }
      float s1 = sb1[t] + red[half][TA * 2 * 16 * 64 + 64 * t + lane];
      s1 += __shfl_xor(s1, 32);
      if (h == 0) pb[T * i + tile] = s1;
    }
  }
#endif
}

extern "C" {

// bytes of scratch the bf16 variants need per call (the weight image of the larger of the two kernels)
int64_t dvgo_shade_scratch_bytes(int width) {
  if (width == 128) return (int64_t)sizeof(X3Img<128, 3>) > (int64_t)sizeof(X3BwdImg<128>) ? sizeof(X3Img<128, 3>) : sizeof(X3BwdImg<128>);
  if (width == 64) return (int64_t)sizeof(X3Img<64, 3>) > (int64_t)sizeof(X3BwdImg<64>) ? sizeof(X3Img<64, 3>) : sizeof(X3BwdImg<64>);
  return 0;
}

int dvgo_shade_fwd_x3(const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M, const int64_t* m_dev,
                      const float* W1, const float* b1, const float* W2, const float* b2, const float* W3,
                      const float* b3, int width, int d_in, int diffuse, float* rgb, float* H1, float* H2,
                      uint64_t* masks, void* scratch, void* scratch_bwd, int experiment, void* stream) {
  if (M < 0 || C <= 0 || E < 0) return DVGO_EINVAL;
  if (M == 0) return 0;
  if (!feat || !emb || !ray_id || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !rgb || !scratch) return DVGO_EINVAL;
  if ((H1 == nullptr) != (H2 == nullptr) || (H1 == nullptr) != (masks == nullptr)) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if (n_view < 0 || d_in != n_view + E) return DVGO_EINVAL;
  if ((width != 128 && width != 64) || d_in > 48) return DVGO_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t cap = 256;                                   // one workgroup per CU (the split weights fill the LDS)
  const int blocks = (int)((n_tiles + X3_WAVES - 1) / X3_WAVES < cap ? (n_tiles + X3_WAVES - 1) / X3_WAVES : cap);
#define DVGO_SHADE_X3(W, KS, DIFF)                                                                                       \
  do {                                                                                                                   \
    x3_prep_fwd_kernel<W, KS><<<scratch_bwd ? 64 : 32, 256, 0, s>>>((X3Img<W, KS>*)scratch, (X3BwdImg<W>*)scratch_bwd, W1, b1, W2, b2, W3, b3, d_in); \
    shade_fwd_x3_kernel<W, KS, DIFF><<<blocks, X3_THREADS, 0, s>>>(feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, scratch, rgb, \
                                                                    H1, H2, (unsigned long long*)masks, experiment);     \
  } while (0)
  if (width == 128) {
    if (diffuse) DVGO_SHADE_X3(128, 3, true); else DVGO_SHADE_X3(128, 3, false);
  } else {
    if (d_in <= 16) { if (diffuse) DVGO_SHADE_X3(64, 1, true); else DVGO_SHADE_X3(64, 1, false); }
    else            { if (diffuse) DVGO_SHADE_X3(64, 3, true); else DVGO_SHADE_X3(64, 3, false); }
  }
#undef DVGO_SHADE_X3
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_shade_bwd_x3(const float* g_rgb, const float* rgb, const uint64_t* masks, int64_t M, const int64_t* m_dev,
                      const float* W1, const float* W2, const float* W3, int width, int d_in, int C, int diffuse,
                      float* g_feat, float* G1, float* gz, void* scratch, int prebuilt, void* stream) {
  if (M < 0 || C <= 0) return DVGO_EINVAL;
  if (M == 0) return 0;
  if (!g_rgb || !rgb || !masks || !W1 || !W2 || !W3 || !g_feat || !G1 || !gz || !scratch) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if ((width != 128 && width != 64) || n_view < 0 || n_view > 32 || d_in < n_view) return DVGO_ERANGE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (M + 31) / 32;
  const int64_t cap = 256;
  const int blocks = (int)((n_tiles + X3_BWD_WAVES - 1) / X3_BWD_WAVES < cap ? (n_tiles + X3_BWD_WAVES - 1) / X3_BWD_WAVES : cap);
#define DVGO_SHADE_BWD_X3(W, DIFF)                                                                                        \
  do {                                                                                                                    \
    if (!prebuilt) x3_prep_bwd_kernel<W><<<32, 256, 0, s>>>((X3BwdImg<W>*)scratch, W1, W2, W3, d_in);                   \
    shade_bwd_x3_kernel<W, DIFF><<<blocks, X3_BWD_THREADS, 0, s>>>(g_rgb, rgb, (const unsigned long long*)masks, M, m_dev, scratch, C, \
                                                                c_view0, n_view, g_feat, G1, gz);                         \
  } while (0)
  if (width == 128) { if (diffuse) DVGO_SHADE_BWD_X3(128, true); else DVGO_SHADE_BWD_X3(128, false); }
  else              { if (diffuse) DVGO_SHADE_BWD_X3(64, true); else DVGO_SHADE_BWD_X3(64, false); }
#undef DVGO_SHADE_BWD_X3
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_shade_wgrad_x3(const float* G1, const float* gz, const uint64_t* masks, const float* W3, const float* H1,
                        const float* H2, const float* feat, int C, const float* emb, int E, const int64_t* ray_id, int64_t M,
                        const int64_t* m_dev, int width, int diffuse, int n_parts, float* part, int form_b, void* stream) {
  if (M < 0 || n_parts <= 0 || C <= 0 || E < 0) return DVGO_EINVAL;
  if (!G1 || !gz || !masks || !W3 || !H1 || !H2 || !feat || !emb || !ray_id || !part) return DVGO_EINVAL;
  const int c_view0 = diffuse ? 3 : 0;
  const int n_view = C - c_view0;
  if ((width != 128 && width != 64) || n_view < 0 || n_view + E > 40) return DVGO_ERANGE;
  if (form_b == 2) {
    // no LDS, no barriers: 32-bit byte offsets everywhere
    if (M * width * 4 >= ((int64_t)1 << 32) || M * C * 4 >= ((int64_t)1 << 32) || M * 32 >= ((int64_t)1 << 32)) return DVGO_ERANGE;
    if (width == 128) {
      shade_wgrad_c2_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(gz, (const unsigned int*)masks, W3, H1, H2, M, m_dev, part);
      shade_wgrad_c1_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(G1, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part);
    } else {
      shade_wgrad_c2_kernel<64><<<n_parts, 128, 0, (hipStream_t)stream>>>(gz, (const unsigned int*)masks, W3, H1, H2, M, m_dev, part);
      shade_wgrad_c1_kernel<64><<<n_parts, 256, 0, (hipStream_t)stream>>>(G1, feat, C, c_view0, n_view, emb, E, ray_id, M, m_dev, part);
    }
  } else if (form_b && width == 128)
    shade_wgrad_x3b_kernel<128><<<n_parts, 256, 0, (hipStream_t)stream>>>(G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C,
                                                                          c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  else if (form_b)
    shade_wgrad_x3b_kernel<64><<<n_parts, 128, 0, (hipStream_t)stream>>>(G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C,
                                                                         c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  else if (width == 128)
    shade_wgrad_x3_kernel<128><<<n_parts, 512, 0, (hipStream_t)stream>>>(G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C,
                                                                         c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  else
    shade_wgrad_x3_kernel<64><<<n_parts, 256, 0, (hipStream_t)stream>>>(G1, gz, (const unsigned int*)masks, W3, H1, H2, feat, C,
                                                                        c_view0, n_view, emb, E, ray_id, M, m_dev, part);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
