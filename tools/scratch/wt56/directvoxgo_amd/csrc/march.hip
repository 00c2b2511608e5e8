// Fused volumetric ray march for gfx950 -- the MI355X-native pipeline behind
// DirectVoxGO.forward (/root/reference/lib/dvgo.py:450-577) and its backward.
//
// The reference runs ~20 kernels with 5 host syncs and materialises every intermediate
// (ray_pts, int64 ids, masks, four boolean compactions).  Here the path is four kernels:
//
//   march_density   one wavefront per ray, lanes = 64 consecutive steps:
//                   position -> bbox test -> occupancy byte -> density trilinear -> alpha ->
//                   alpha filter -> wave product-scan of (1-alpha) with ballot early stop ->
//                   weight filter -> ballot/popcount compaction into per-ray scratch records
//   (scan of the per-ray survivor counts, one workgroup)
//   march_gather    flat over the surviving samples in the reference's (ray, step) order:
//                   feature-grid trilinear (16-byte channel vectors) + final ids/weights
//   march_composite one wavefront per ray: weighted colour/depth sum + background
//
// and the backward mirrors it (composite_bwd, feat_bwd, density_bwd).  Filter order is the
// reference's: mask -> alpha > thres (before transmittance) -> T < 1e-3 stop -> weight > thres.
#include <stdlib.h>

#include "common.h"

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) {
  return (lane == 0) ? 0ull : (~0ull >> (64 - lane));
}


// inclusive product scan across the wave
__device__ __forceinline__ float wave_prod_scan(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d);
    if (lane >= d) v *= o;
  }
  return v;
}

// exclusive suffix sum across the wave: sum of v over lanes > lane
__device__ __forceinline__ float wave_suffix_excl(float v, int lane, float& total) {
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(inc, d);
    if (lane + d < 64) inc += o;
  }
  total = dvgo_readlane_f(inc, 0);
  float ex = __shfl_down(inc, 1);
  if (lane == 63) ex = 0.0f;
  return ex;
}

// Scratch records of ray r start at cum[r]-n_steps[r] (exact, ray-major M0 layout) when the
// inclusive cumsum is given, else at r*rec_stride (fixed stride = an upper bound of n_steps,
// which saves the scan and the host read of M0).
__device__ __forceinline__ int64_t rec_base(const int64_t* __restrict__ cum, const int64_t* __restrict__ n_steps,
                                            int64_t rec_stride, int64_t ray) {
  return cum ? (cum[ray] - n_steps[ray]) : ray * rec_stride;
}

// run-time selectable kernel variants (A/B measurements in one process; defaults = fastest measured)
static int g_tuning[DVGO_TUNE_COUNT] = {1, 1, 0, 0, 0, 0, 0, 0};

struct MarchParams {
  float mnx, mny, mnz, mxx, mxy, mxz;
  float stepdist;          // > 0: metric step (K6); < 0: NDC spacing, dist = step / (-stepdist) (K7)
  float scx, scy, scz, shx, shy, shz;   // xyz2ijk scale / shift
  int mX, mY, mZ;
  int X, Y, Z;
  float act_shift, interval, thres;
};

// ----------------------------------------------------------------------------------
// Brick lists (brick.hip): every sample that entered compositing is listed by each 8x8x8 brick that holds one of
// its 8 corner voxels (1 brick for 2/3 of the samples, up to 8 on brick faces / edges / corners).  The same routine
// counts (forward) and fills (backward), so both see the same (sample, brick) incidences.
// The 64 lanes of a chunk are consecutive steps of one ray and touch only a few dozen distinct bricks, so the wave
// first DISCOVERS the distinct bricks with ballots only -- pick the first lane with an unlisted incidence, broadcast
// its brick, ballot the lanes that touch it; lane #i remembers brick #i and its incidence count, every touching lane
// remembers (i, its rank) -- then ONE atomic wave-instruction counts / reserves slots for all bricks of the chunk
// (lane i for brick i), and the records go out.  One atomic round trip per chunk: per-incidence atomics on the 8000
// counters cost 75 us per pass at 2 M samples, one returning atomic per distinct brick 87 us (a serial chain of
// round trips).
//   FILL = false : cur = per-brick counters
//   FILL = true  : cur = per-brick fill cursors (start at the brick's offset); writes recs[slot] = rec
// Must be called by all 64 lanes.
// ----------------------------------------------------------------------------------
template <bool FILL>
__device__ __forceinline__ void brick_emit(bool act, int i0, int j0, int k0, int X, int Y, int Z, int lane,
                                           int32_t* __restrict__ cur, int4* __restrict__ recs, const int4 rec) {
  const int BY = (Y + DVGO_BRICK - 1) >> DVGO_BRICK_LOG, BZ = (Z + DVGO_BRICK - 1) >> DVGO_BRICK_LOG;
  int bx0 = 0, bx1 = 0, by0 = 0, by1 = 0, bz0 = 0, bz1 = 0, nx = 0, ny = 0, nz = 0;
  if (act) {
    nx = dvgo_brick_axis(i0, X, bx0, bx1);
    ny = dvgo_brick_axis(j0, Y, by0, by1);
    nz = dvgo_brick_axis(k0, Z, bz0, bz1);
  }
  // incidence e = (a * ny + b) * nz + c, a < nx, b < ny, c < nz; bit e of `pending` = not listed yet
  unsigned pending = (act && nx > 0 && ny > 0 && nz > 0) ? ((1u << (nx * ny * nz)) - 1u) : 0u;
  const unsigned long long lt = lanemask_lt(lane);
  while (__ballot(pending != 0u)) {
    // ---- discover up to 64 distinct bricks
    int it = 0, my_id = 0, my_cnt = 0;
    unsigned round = 0u;                               // incidences listed in this round
    unsigned long long iters = 0ull, ranks = 0ull;     // 8 bits per incidence: brick number in the round / rank in brick
    for (;;) {
      const unsigned long long bal = __ballot(pending != 0u);
      if (!bal || it == 64) break;
      const int leader = __ffsll((long long)bal) - 1;
      const int e = __ffs((int)pending) - 1;                  // (garbage on lanes without pending work; never the leader)
      const int c = (nz == 2) ? (e & 1) : 0, ab = (nz == 2) ? (e >> 1) : e;          // nx, ny, nz are 1 or 2
      const int b = (ny == 2) ? (ab & 1) : 0, a = (ny == 2) ? (ab >> 1) : ab;
      const int mine = ((a ? bx1 : bx0) << 20) | ((b ? by1 : by0) << 10) | (c ? bz1 : bz0);
      const int Xp = __builtin_amdgcn_readlane(mine, leader);
      const int Xx = Xp >> 20, Xy = (Xp >> 10) & 1023, Xz = Xp & 1023;
      const int ma = (bx0 == Xx) ? 0 : ((nx == 2 && bx1 == Xx) ? 1 : -1);
      const int mb = (by0 == Xy) ? 0 : ((ny == 2 && by1 == Xy) ? 1 : -1);
      const int mc = (bz0 == Xz) ? 0 : ((nz == 2 && bz1 == Xz) ? 1 : -1);
      const bool touch = (pending != 0u) && ma >= 0 && mb >= 0 && mc >= 0;
      const unsigned long long m = __ballot(touch);
      if (lane == it) { my_id = (Xx * BY + Xy) * BZ + Xz; my_cnt = __popcll(m); }
      if (touch) {
        const int em = (((ma << (ny - 1)) + mb) << (nz - 1)) + mc;
        pending &= ~(1u << em);
        round |= 1u << em;
        iters |= (unsigned long long)it << (8 * em);
        ranks |= (unsigned long long)__popcll(m & lt) << (8 * em);
      }
      ++it;
    }
    // ---- one atomic wave-instruction for all bricks of the round
    int base = 0;
    if (lane < it) {
      if (FILL) base = atomicAdd(&cur[my_id], my_cnt);
      else atomicAdd(&cur[my_id], my_cnt);
    }
    if (FILL) {
#pragma unroll
      for (int em = 0; em < 8; ++em) {
        const bool have = (round >> em) & 1u;
        if (!__ballot(have)) continue;
        const int src = (int)((iters >> (8 * em)) & 255ull);
        const int bs = __shfl(base, src);
        if (have) recs[bs + (int)((ranks >> (8 * em)) & 255ull)] = rec;
      }
    }
  }
}

__global__ void __launch_bounds__(DVGO_BLOCK)
march_density_kernel(const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                     const int64_t* __restrict__ n_steps, const int64_t* __restrict__ cum,
                     int64_t rec_stride, int64_t n_rays, const uint8_t* __restrict__ mask,
                     const float* __restrict__ density, MarchParams P,
                     dvgo_rec2_t* __restrict__ rec2,
                     int32_t* __restrict__ n2, int32_t* __restrict__ n3,
                     float* __restrict__ alphainv_last, int32_t* __restrict__ brick_cnt) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int ns = __builtin_amdgcn_readfirstlane((int)n_steps[ray]);
  const int64_t cs0 = rec_base(cum, n_steps, rec_stride, ray);
  const float sx = rays_start[3 * ray], sy = rays_start[3 * ray + 1], sz = rays_start[3 * ray + 2];
  const float dx = rays_dir[3 * ray], dy = rays_dir[3 * ray + 1], dz = rays_dir[3 * ray + 2];
  const int64_t YZ = (int64_t)P.Y * P.Z;
  const bool filt = P.thres > 0.0f;

  float Tc = 1.0f;
  int c2 = 0, c3 = 0;
  for (int base = 0; base < ns; base += 64) {
    const int step = base + lane;
    const bool act = step < ns;
    const float dist = march_dist(P.stepdist, step);
    const float px = fmaf(dx, dist, sx), py = fmaf(dy, dist, sy), pz = fmaf(dz, dist, sz);
    bool keep = act && !((P.mnx > px) | (P.mny > py) | (P.mnz > pz) | (P.mxx < px) | (P.mxy < py) | (P.mxz < pz));
    if (mask != nullptr && keep) {
      const int i = (int)roundf(fmaf(px, P.scx, P.shx));
      const int j = (int)roundf(fmaf(py, P.scy, P.shy));
      const int k = (int)roundf(fmaf(pz, P.scz, P.shz));
      keep = (0 <= i) & (i < P.mX) & (0 <= j) & (j < P.mY) & (0 <= k) & (k < P.mZ);
      if (keep) keep = mask[((int64_t)i * P.mY + j) * P.mZ + k] != 0;
    }
    float e = 0.f, a = 0.f;
    TriSetup t;
    t.i0 = t.j0 = t.k0 = 0;
    if (keep) {
      t = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
      float d = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (dvgo_tri_inb(t, n, P.X, P.Y, P.Z)) {
          const int64_t off = (int64_t)(t.i0 + ((n >> 2) & 1)) * YZ + (int64_t)(t.j0 + ((n >> 1) & 1)) * P.Z +
                              (t.k0 + (n & 1));
          d = fmaf(density[off], dvgo_tri_weight(t, n), d);
        }
      }
      e = expf(d + P.act_shift);
      a = 1.0f - powf(1.0f + e, -P.interval);
      if (filt) keep = a > P.thres;
    }
    // transmittance, in the reference's order and precision (K12, render_utils_kernel.cu:448-454):
    //   T_cum = (float)((double)T_cum * (1. - alpha + 1e-10)), stop after the first sample with (double)T_cum < 1e-3.
    // The double factor is lane-parallel; the float carry is walked over the KEPT lanes only (a dropped sample
    // never enters compositing): each step is one exec-masked multiply on lane j and a v_readlane of the result,
    // so T, the weights and every threshold decision are bit-identical to the serial code.
    const double f = 1.0 - (double)a + 1e-10;
    float T_before = 1.0f, T_after = 1.0f;
    unsigned long long todo = __ballot(keep);
    int stop_lane = -1;
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      if (lane == j) { T_before = Tc; T_after = (float)((double)Tc * f); }
      Tc = dvgo_readlane_f(T_after, j);
      // (double)T < 1e-3  <=>  T < 1e-3f for floats (1e-3f is the float just above the double 1e-3); T >= 0, so the
      // ordered-unsigned compare of the bit patterns is the same test and stays on the scalar unit
      if (__float_as_uint(Tc) < 0x3A83126Fu) { stop_lane = j; break; }
    }
    const bool stop = stop_lane >= 0;
    const bool valid2 = keep && (!stop || lane <= stop_lane);
    const float w = T_before * a;
    const bool keep3 = valid2 && (!filt || (w > P.thres));
    const unsigned long long m2 = __ballot(valid2), m3 = __ballot(keep3);
    const unsigned long long lt = lanemask_lt(lane);
    if (valid2) {
      dvgo_rec2_t r;
      r.step = step | (keep3 ? (int32_t)0x80000000 : 0);
      r.exp_d = e; r.alpha = a; r.T = T_before;
      rec2[cs0 + c2 + __popcll(m2 & lt)] = r;
    }
    c2 += __popcll(m2);
    c3 += __popcll(m3);
    if (brick_cnt != nullptr) brick_emit<false>(valid2, t.i0, t.j0, t.k0, P.X, P.Y, P.Z, lane, brick_cnt, nullptr, make_int4(0, 0, 0, 0));
    if (stop) break;
  }
  if (lane == 0) {
    n2[ray] = c2;
    n3[ray] = c3;
    alphainv_last[ray] = Tc;
  }
}

// ----------------------------------------------------------------------------------
// hit test: does a ray have at least one in-box sample in known-occupied space?  Fused form of
// DirectVoxGO.hit_coarse_geo (lib/dvgo.py:412-423: sample_pts_on_rays + boolean compaction + maskcache_lookup
// + index_put), one wavefront per ray, nothing materialised.  Used to pre-filter the training rays
// (lib/ray_utils.py:145-183 walks every pixel of every training image through it).
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(DVGO_BLOCK)
march_hit_kernel(const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                 const int64_t* __restrict__ n_steps, int64_t n_rays, const uint8_t* __restrict__ mask, MarchParams P,
                 uint8_t* __restrict__ hit) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int ns = __builtin_amdgcn_readfirstlane((int)n_steps[ray]);
  const float sx = rays_start[3 * ray], sy = rays_start[3 * ray + 1], sz = rays_start[3 * ray + 2];
  const float dx = rays_dir[3 * ray], dy = rays_dir[3 * ray + 1], dz = rays_dir[3 * ray + 2];
  bool any = false;
  for (int base = 0; base < ns && !any; base += 64) {
    const int step = base + lane;
    const float dist = march_dist(P.stepdist, step);
    const float px = fmaf(dx, dist, sx), py = fmaf(dy, dist, sy), pz = fmaf(dz, dist, sz);
    bool keep = (step < ns) && !((P.mnx > px) | (P.mny > py) | (P.mnz > pz) | (P.mxx < px) | (P.mxy < py) | (P.mxz < pz));
    if (keep) {
      const int i = (int)roundf(fmaf(px, P.scx, P.shx));
      const int j = (int)roundf(fmaf(py, P.scy, P.shy));
      const int k = (int)roundf(fmaf(pz, P.scz, P.shz));
      keep = (0 <= i) & (i < P.mX) & (0 <= j) & (j < P.mY) & (0 <= k) & (k < P.mZ);
      if (keep) keep = mask[((int64_t)i * P.mY + j) * P.mZ + k] != 0;
    }
    any = __ballot(keep) != 0ull;
  }
  if (lane == 0) hit[ray] = any ? 1 : 0;
}

// ----------------------------------------------------------------------------------
// march_gather: one wavefront per ray over the ray's rec2 records (lanes = 64 consecutive records).  The records
// flagged "kept" by march_density are compacted with a ballot to their final position off3[ray] + rank -- the
// reference's (ray, step) order after its 4th boolean compaction (lib/dvgo.py:488-509) -- and each kept lane
// interpolates its feature row (8 corners x 16-byte channel vectors when the grid is channels-last) and writes the
// ids / weight / alpha of its sample.  weight = T * alpha is the product march_density formed for its filter, from
// the same two floats, so no second record array travels between the two kernels.
// ----------------------------------------------------------------------------------
template <int CVEC, int CS = 0>   // CVEC > 0: channels-last, C == 4*CVEC, 16-B aligned; CS > 0: channels-last, C == CS, dword
__global__ void __launch_bounds__(DVGO_BLOCK)   // loads (rows of 3 / 9 floats: coarse stage, LLFF); both 0: generic strides
march_gather_kernel(const dvgo_rec2_t* __restrict__ rec2, const int32_t* __restrict__ n2, const int64_t* __restrict__ n_steps,
                    const int64_t* __restrict__ cum, int64_t rec_stride, const int64_t* __restrict__ off3,
                    int64_t n_rays, const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                    MarchParams P, const float* __restrict__ k0, int C, int64_t sC, int64_t sX,
                    int64_t sY, int64_t sZ, int64_t* __restrict__ ray_id, int64_t* __restrict__ step_id,
                    float* __restrict__ weights, float* __restrict__ alpha, float* __restrict__ feat) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (r >= n_rays) return;
  const int c2 = __builtin_amdgcn_readfirstlane(n2[r]);
  if (c2 == 0) return;
  const int64_t cs0 = rec_base(cum, n_steps, rec_stride, r);
  int64_t out = off3[r];
  const float sx = rays_start[3 * r], sy = rays_start[3 * r + 1], sz = rays_start[3 * r + 2];
  const float dx = rays_dir[3 * r], dy = rays_dir[3 * r + 1], dz = rays_dir[3 * r + 2];
  const unsigned long long lt = lanemask_lt(lane);
  for (int lo = 0; lo < c2; lo += 64) {
    const int j = lo + lane;
    dvgo_rec2_t rec;
    rec.step = 0; rec.exp_d = 0.f; rec.alpha = 0.f; rec.T = 0.f;
    if (j < c2) rec = rec2[cs0 + j];
    const bool kept = (j < c2) && (rec.step < 0);
    const unsigned long long m = __ballot(kept);
    const int64_t i = out + __popcll(m & lt);
    out += __popcll(m);
    if (!kept) continue;
    const int step = rec.step & 0x7fffffff;
    ray_id[i] = r;
    step_id[i] = step;
    weights[i] = rec.T * rec.alpha;
    alpha[i] = rec.alpha;
    const float dist = march_dist(P.stepdist, step);
    const float px = fmaf(dx, dist, sx), py = fmaf(dy, dist, sy), pz = fmaf(dz, dist, sz);
    const TriSetup t = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
    float w[8];
    int64_t off[8];
    bool ok[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      w[n] = dvgo_tri_weight(t, n);
      ok[n] = dvgo_tri_inb(t, n, P.X, P.Y, P.Z);
      off[n] = (int64_t)(t.i0 + ((n >> 2) & 1)) * sX + (int64_t)(t.j0 + ((n >> 1) & 1)) * sY +
               (int64_t)(t.k0 + (n & 1)) * sZ;
    }
    if (CVEC > 0) {
      float4 acc[CVEC > 0 ? CVEC : 1];
#pragma unroll
      for (int c = 0; c < CVEC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (ok[n]) {
          const float4* p = reinterpret_cast<const float4*>(k0 + off[n]);
#pragma unroll
          for (int c = 0; c < CVEC; ++c) {
            const float4 v = p[c];
            acc[c].x = fmaf(v.x, w[n], acc[c].x);
            acc[c].y = fmaf(v.y, w[n], acc[c].y);
            acc[c].z = fmaf(v.z, w[n], acc[c].z);
            acc[c].w = fmaf(v.w, w[n], acc[c].w);
          }
        }
      }
      float4* o = reinterpret_cast<float4*>(feat + i * (int64_t)(4 * CVEC));
#pragma unroll
      for (int c = 0; c < CVEC; ++c) o[c] = acc[c];
    } else if (CS > 0) {
      float acc[CS > 0 ? CS : 1];
#pragma unroll
      for (int c = 0; c < CS; ++c) acc[c] = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        if (ok[n]) {
          const float* p = k0 + off[n];
#pragma unroll
          for (int c = 0; c < CS; ++c) acc[c] = fmaf(p[c], w[n], acc[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < CS; ++c) feat[i * CS + c] = acc[c];
    } else {
      for (int c = 0; c < C; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n)
          if (ok[n]) acc = fmaf(k0[c * sC + off[n]], w[n], acc);
        feat[i * C + c] = acc;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// march_composite: one wavefront per ray over [off3[r], off3[r+1]).
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__global__ void __launch_bounds__(DVGO_BLOCK)
march_composite_kernel(const float* __restrict__ weights, const float* __restrict__ rgb,
                       const int64_t* __restrict__ step_id, const int64_t* __restrict__ off3,
                       int64_t n_rays, const float* __restrict__ alphainv_last, float bg,
                       float* __restrict__ rgb_marched, float* __restrict__ depth) {
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int64_t b = off3[ray], e = off3[ray + 1];
  float r = 0.f, g = 0.f, bl = 0.f, dsum = 0.f;
  for (int64_t i = b + lane; i < e; i += 64) {
    const float w = weights[i];
    r = fmaf(w, rgb[3 * i + 0], r);
    g = fmaf(w, rgb[3 * i + 1], g);
    bl = fmaf(w, rgb[3 * i + 2], bl);
    if (depth) dsum = fmaf(w, (float)step_id[i], dsum);
  }
  r = wave_sum(r); g = wave_sum(g); bl = wave_sum(bl);
  if (depth) dsum = wave_sum(dsum);
  if (lane == 0) {
    const float last = alphainv_last[ray] * bg;
    rgb_marched[3 * ray + 0] = r + last;
    rgb_marched[3 * ray + 1] = g + last;
    rgb_marched[3 * ray + 2] = bl + last;
    if (depth) depth[ray] = dsum;
  }
}

__global__ void __launch_bounds__(DVGO_BLOCK)
march_composite_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ weights,
                           const float* __restrict__ rgb, const int64_t* __restrict__ ray_id, int64_t M_cap,
                           const int64_t* __restrict__ m_dev,
                           float* __restrict__ grad_weights, float* __restrict__ grad_rgb, int64_t n_rays, float bg,
                           float* __restrict__ grad_last) {
  const int64_t M3 = m_dev ? (*m_dev < M_cap ? *m_dev : M_cap) : M_cap;      // sample count kept on the device (train.py)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // d rgb_marched / d alphainv_last = bg (lib/dvgo.py:559): the N-sized reduction rides on the first n_rays threads
  if (grad_last != nullptr && i < n_rays) grad_last[i] = (gout[3 * i] + gout[3 * i + 1] + gout[3 * i + 2]) * bg;
  if (i >= M3) return;
  const int64_t r = ray_id[i];
  const float g0 = gout[3 * r], g1 = gout[3 * r + 1], g2 = gout[3 * r + 2];
  const float w = weights[i];
  if (grad_weights)
    grad_weights[i] = fmaf(g2, rgb[3 * i + 2], fmaf(g1, rgb[3 * i + 1], g0 * rgb[3 * i]));
  if (grad_rgb) {
    grad_rgb[3 * i + 0] = g0 * w;
    grad_rgb[3 * i + 1] = g1 * w;
    grad_rgb[3 * i + 2] = g2 * w;
  }
}

// ----------------------------------------------------------------------------------
// march_feat_bwd: thread = (sample, channel); 8 float atomics each.  Channels-last keeps the
// C channels of a corner on adjacent lanes (contiguous 4*C-byte runs per atomic instruction).
// ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(DVGO_BLOCK)
march_feat_bwd_kernel(const float* __restrict__ grad_feat, const int64_t* __restrict__ ray_id,
                      const int64_t* __restrict__ step_id, int64_t M3,
                      const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                      MarchParams P, int C, int64_t sC, int64_t sX, int64_t sY, int64_t sZ,
                      float* __restrict__ grad_k0) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= M3 * C) return;
  const int64_t i = tid / C;
  const int c = (int)(tid - i * C);
  float px, py, pz;
  march_pos(rays_start, rays_dir, ray_id[i], P.stepdist, (int)step_id[i], px, py, pz);
  const TriSetup t = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
  const float g = grad_feat[tid];
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    if (!dvgo_tri_inb(t, n, P.X, P.Y, P.Z)) continue;
    const int64_t off = (int64_t)(t.i0 + ((n >> 2) & 1)) * sX + (int64_t)(t.j0 + ((n >> 1) & 1)) * sY +
                        (int64_t)(t.k0 + (n & 1)) * sZ;
    atomicAdd(grad_k0 + c * sC + off, dvgo_tri_weight(t, n) * g);
  }
}

// ----------------------------------------------------------------------------------
// march_feat_bwd, de-duplicating form (channels-last grids).
//
// Float atomics leave the chip as 64-B memory-side requests at a fixed chip-wide rate
// (MI355X_MICROARCH "Global float atomics"; measured here: 29.5 M requests in 1.50 ms = 19.7 G/s for
// the one-atomic-per-(sample,corner,channel) form), so the scatter is bound by how many corner rows
// are emitted, not by HBM.  Consecutive samples of a ray are half a voxel apart and share most of
// their corners, so every wavefront first groups the 256 corner references of SPP = 32 consecutive
// samples by voxel and emits each distinct corner once, as one contiguous 4*C-byte row.
//
// Grouping is a counting sort in the wave's private LDS, built so that the accumulation itself needs
// no LDS float atomics (those serialise on equal addresses, which is exactly the case being merged;
// a first version that accumulated with ds_add_f32 was LDS-bound at 1.23 ms):
//   A  lanes = (sample, half of its 8 corners); insert the corner's voxel index into an open-addressing
//      table (atomicCAS), take a ticket in the slot's reference counter;
//   B  list the occupied slots and exclusive-scan their counters (wave prefix sum);
//   C  every reference writes {weight, sample} at offset[slot] + ticket;
//   D  owner computes: C adjacent lanes own one corner row, walk its references, read the sample's
//      gradient row from LDS, accumulate in a register, emit ONE global atomic per (corner, channel).
// A corner that cannot be placed within the probe bound falls back to direct global atomics.
// ----------------------------------------------------------------------------------
// EXTRA: one more per-sample scalar (grad_extra[i], the density gradient of the kept samples) travels as channel C
// of the same row, so that with 64-byte rows (RS = 16 floats) the density scatter costs no atomic request of its
// own: float atomics are bound by 64-B requests (~19.7 G/s), and a 48-B row at a 48-B stride straddles two
// requests half of the time (0.465 -> 0.34 ms from the alignment alone on the roofline case).
template <int C, bool EXTRA>
__global__ void __launch_bounds__(DVGO_BLOCK)
march_feat_bwd_dedup_kernel(const float* __restrict__ grad_feat, const float* __restrict__ grad_extra,
                            const int64_t* __restrict__ ray_id,
                            const int64_t* __restrict__ step_id, int64_t M3,
                            const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                            MarchParams P, float* __restrict__ grad_k0, int RS) {
  constexpr int H = 256;        // table slots per wave (>= corner references per pass)
  constexpr int SPP = 32;       // samples per pass
  constexpr int CE = C + (EXTRA ? 1 : 0);       // channels scattered per corner row
  constexpr int GS = EXTRA ? (C + 4) : C;       // LDS row stride (keeps the float4 stores aligned)
  constexpr int RPI = 64 / CE;  // corner rows per atomic wave-instruction
  struct WaveLds {
    int keys[H];
    int cnts[H];
    int offs[H];
    int list[H];
    float ref_w[H];
    int ref_s[H];
    float g[SPP][GS];
  };
  __shared__ __attribute__((aligned(16))) WaveLds s_lds[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  WaveLds& L = s_lds[wave];
  for (int s = lane; s < H; s += 64) { L.keys[s] = -1; L.cnts[s] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  const unsigned long long lt = lanemask_lt(lane);
  const int64_t n_pass = (M3 + SPP - 1) / SPP;
  const int64_t gwave = (int64_t)blockIdx.x * 4 + wave, nwaves = (int64_t)gridDim.x * 4;
  const int YZ = P.Y * P.Z;
  const int rsub = lane / CE, ch = lane - rsub * CE;
  for (int64_t pass = gwave; pass < n_pass; pass += nwaves) {
    const int sl = lane >> 1, half = lane & 1;
    const int64_t i = pass * SPP + sl;
    int slot[4], tick[4];
    float wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { slot[q] = -1; tick[q] = 0; wq[q] = 0.0f; }
    // ---- A: table insert + tickets
    if (i < M3) {
      float px, py, pz;
      march_pos(rays_start, rays_dir, ray_id[i], P.stepdist, (int)step_id[i], px, py, pz);
      const TriSetup t = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
      float gs[C];                       // this sample's gradient row (16-B loads when the row allows it)
      if constexpr (C % 4 == 0) {
        const float4* gp = reinterpret_cast<const float4*>(grad_feat + i * C);
#pragma unroll
        for (int c = 0; c < C / 4; ++c) {
          const float4 v = gp[c];
          gs[4 * c] = v.x; gs[4 * c + 1] = v.y; gs[4 * c + 2] = v.z; gs[4 * c + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) gs[c] = grad_feat[i * C + c];
      }
      const float ge = EXTRA ? grad_extra[i] : 0.0f;
      if (half == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) L.g[sl][c] = gs[c];
        if (EXTRA) L.g[sl][C] = ge;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = half * 4 + q;
        if (!dvgo_tri_inb(t, n, P.X, P.Y, P.Z)) continue;
        const float w = dvgo_tri_weight(t, n);
        if (w == 0.0f) continue;
        const int key = (t.i0 + ((n >> 2) & 1)) * YZ + (t.j0 + ((n >> 1) & 1)) * P.Z + (t.k0 + (n & 1));
        int sidx = (int)((unsigned)key * 2654435761u >> 24);     // H = 256
        bool placed = false;
        for (int probes = 0; probes < 16; ++probes) {
          const int prev = atomicCAS(&L.keys[sidx], -1, key);
          if (prev == -1 || prev == key) { placed = true; break; }
          sidx = (sidx + 1) & (H - 1);
        }
        if (placed) {
          slot[q] = sidx;
          tick[q] = atomicAdd(&L.cnts[sidx], 1);
          wq[q] = w;
        } else {
          float* dst = grad_k0 + (int64_t)key * RS;
#pragma unroll
          for (int c = 0; c < C; ++c) atomicAdd(dst + c, w * gs[c]);
          if (EXTRA) atomicAdd(dst + C, w * ge);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- B: occupied-slot list + exclusive scan of the reference counters
    int cnt = 0, run = 0;
#pragma unroll
    for (int it = 0; it < H / 64; ++it) {
      const int s = it * 64 + lane;
      const int c = L.cnts[s];
      const unsigned long long bal = __ballot(c > 0);
      int inc = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
      }
      if (c > 0) {
        L.list[cnt + __popcll(bal & lt)] = s;
        L.offs[s] = run + inc - c;
      }
      run += __builtin_amdgcn_readlane(inc, 63);
      cnt += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- C: references to their sorted position
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (slot[q] >= 0) {
        const int idx = L.offs[slot[q]] + tick[q];
        L.ref_w[idx] = wq[q];
        L.ref_s[idx] = sl;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- D: owner computes, one global atomic per (corner, channel)
    for (int r0 = 0; r0 < cnt; r0 += RPI) {
      const int row = r0 + rsub;
      if (rsub < RPI && row < cnt) {
        const int s = L.list[row];
        const int key = L.keys[s];
        const int beg = L.offs[s], n = L.cnts[s];
        float acc = 0.0f;
        for (int t = 0; t < n; ++t) acc = fmaf(L.ref_w[beg + t], L.g[L.ref_s[beg + t]][ch], acc);
        atomicAdd(grad_k0 + (int64_t)key * RS + ch, acc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int r = lane; r < cnt; r += 64) { const int s = L.list[r]; L.keys[s] = -1; L.cnts[s] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
}

// ----------------------------------------------------------------------------------
// march_density_bwd: one wavefront per ray, rec2 chunks walked from the far end.
//   K13 (render_utils_kernel.cu:521-530): g_alpha = g_w*T - acc/((1-alpha)+1e-10), acc += g_w*w
//   K10 (:402-405) raw2alpha backward, then the 8-corner scatter into grad_density.
// ----------------------------------------------------------------------------------
// DEDUP: the 8 x 64 corner contributions of a chunk are first merged in a per-wave LDS hash table
// (single channel: ds_add_f32 on the slot is cheap here) and each distinct voxel is emitted once.
template <bool DEDUP>
__global__ void __launch_bounds__(DVGO_BLOCK)
march_density_bwd_kernel(const dvgo_rec2_t* __restrict__ rec2, const int32_t* __restrict__ n2,
                         const int64_t* __restrict__ n_steps, const int64_t* __restrict__ cum,
                         int64_t rec_stride, const int64_t* __restrict__ off3, int64_t n_rays,
                         const float* __restrict__ rays_start, const float* __restrict__ rays_dir,
                         MarchParams P, const float* __restrict__ alphainv_last,
                         const float* __restrict__ grad_weights, const float* __restrict__ grad_last,
                         float* __restrict__ grad_density, int64_t gstride /* elements between voxels */,
                         float* __restrict__ grad_kept /* [M3] or null: kept samples hand their gradient to the
                                                          feature scatter instead of scattering it here */,
                         int32_t* __restrict__ brick_cursor /* null, or: no scatter here at all -- every sample is
                                                               appended to the brick lists (brick.hip) */,
                         int4* __restrict__ brick_recs) {
  constexpr int H = 512;
  __shared__ int s_keys[DEDUP ? 4 : 1][DEDUP ? H : 1];
  __shared__ float s_vals[DEDUP ? 4 : 1][DEDUP ? H : 1];
  const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (ray >= n_rays) return;
  const int c2 = __builtin_amdgcn_readfirstlane(n2[ray]);
  if (c2 == 0) return;
  int* keys = s_keys[DEDUP ? (threadIdx.x >> 6) : 0];
  float* vals = s_vals[DEDUP ? (threadIdx.x >> 6) : 0];
  if (DEDUP) {
    for (int s = lane; s < H; s += 64) { keys[s] = -1; vals[s] = 0.0f; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  const int64_t cs0 = rec_base(cum, n_steps, rec_stride, ray);
  const int64_t o3 = off3[ray];
  int c3_rem = (int)(off3[ray + 1] - o3);
  const float sx = rays_start[3 * ray], sy = rays_start[3 * ray + 1], sz = rays_start[3 * ray + 2];
  const float dx = rays_dir[3 * ray], dy = rays_dir[3 * ray + 1], dz = rays_dir[3 * ray + 2];
  const int64_t YZ = (int64_t)P.Y * P.Z;
  const unsigned long long lt = lanemask_lt(lane);
  float acc = (grad_last ? grad_last[ray] : 0.0f) * alphainv_last[ray];
  for (int hi = c2; hi > 0; hi -= 64) {
    const int lo = max(0, hi - 64);
    const int n = hi - lo;
    const bool act = lane < n;
    dvgo_rec2_t rec;
    rec.step = 0; rec.exp_d = 0.f; rec.alpha = 0.f; rec.T = 0.f;
    if (act) rec = rec2[cs0 + lo + lane];
    const bool flag = act && (rec.step < 0);
    const int step = rec.step & 0x7fffffff;
    const unsigned long long m = __ballot(flag);
    const int cnt = __popcll(m);
    const int rank = c3_rem - cnt + __popcll(m & lt);
    c3_rem -= cnt;
    const float gw = flag ? grad_weights[o3 + rank] : 0.0f;
    const float w = rec.T * rec.alpha;
    float total;
    const float suffix = wave_suffix_excl(gw * w, lane, total);   // inactive lanes contribute 0
    const float my_acc = acc + suffix;
    acc += total;
    float g_d = 0.0f;
    TriSetup t;
    t.i0 = t.j0 = t.k0 = 0; t.gx = t.gy = t.gz = 0.f;
    if (act) {
      const float gt = gw * rec.T;
      const float one_minus = 1.0f - rec.alpha;
      const float g_alpha = (float)((double)gt - (double)my_acc / ((double)one_minus + 1e-10));
      double v = fmin((double)rec.exp_d, 1e10) * (double)powf(1.0f + rec.exp_d, -P.interval - 1.0f);
      v = v * (double)P.interval;
      v = v * (double)g_alpha;
      g_d = (float)v;
      const float dist = march_dist(P.stepdist, step);
      const float px = fmaf(dx, dist, sx), py = fmaf(dy, dist, sy), pz = fmaf(dz, dist, sz);
      t = dvgo_tri_setup(px, py, pz, P.mnx, P.mny, P.mnz, P.mxx, P.mxy, P.mxz, P.X, P.Y, P.Z);
    }
    if (brick_cursor != nullptr) {
      // record = {kept index in the M3 order or -1, ray, step, density gradient}: 16 bytes, one store
      brick_emit<true>(act, t.i0, t.j0, t.k0, P.X, P.Y, P.Z, lane, brick_cursor, brick_recs,
                       make_int4(flag ? (int)(o3 + rank) : -1, (int)ray, step, __float_as_int(g_d)));
    } else if (act) {
      if (grad_kept != nullptr && flag) grad_kept[o3 + rank] = g_d;
      else if (g_d != 0.0f) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          if (!dvgo_tri_inb(t, c, P.X, P.Y, P.Z)) continue;
          const int64_t off = (int64_t)(t.i0 + ((c >> 2) & 1)) * YZ + (int64_t)(t.j0 + ((c >> 1) & 1)) * P.Z +
                              (t.k0 + (c & 1));
          const float v = dvgo_tri_weight(t, c) * g_d;
          if (DEDUP) {
            const int key = (int)off;
            int sidx = (int)((unsigned)key * 2654435761u >> 23);     // H = 512
            bool placed = false;
            for (int probes = 0; probes < 16; ++probes) {
              const int prev = atomicCAS(&keys[sidx], -1, key);
              if (prev == -1 || prev == key) { placed = true; break; }
              sidx = (sidx + 1) & (H - 1);
            }
            if (placed) atomicAdd(&vals[sidx], v);
            else atomicAdd(grad_density + off * gstride, v);
          } else {
            atomicAdd(grad_density + off * gstride, v);
          }
        }
      }
    }
    if (DEDUP) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
      for (int it = 0; it < H / 64; ++it) {
        const int s = it * 64 + lane;
        const int k = keys[s];
        if (k != -1) {
          atomicAdd(grad_density + k * gstride, vals[s]);
          keys[s] = -1;
          vals[s] = 0.0f;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
}

static MarchParams make_params(const float* mn, const float* mx, float stepdist, const float* sc,
                               const float* sh, int mX, int mY, int mZ, int X, int Y, int Z,
                               float act_shift, float interval, float thres) {
  MarchParams P;
  P.mnx = mn[0]; P.mny = mn[1]; P.mnz = mn[2];
  P.mxx = mx[0]; P.mxy = mx[1]; P.mxz = mx[2];
  P.stepdist = stepdist;
  P.scx = sc ? sc[0] : 0.f; P.scy = sc ? sc[1] : 0.f; P.scz = sc ? sc[2] : 0.f;
  P.shx = sh ? sh[0] : 0.f; P.shy = sh ? sh[1] : 0.f; P.shz = sh ? sh[2] : 0.f;
  P.mX = mX; P.mY = mY; P.mZ = mZ;
  P.X = X; P.Y = Y; P.Z = Z;
  P.act_shift = act_shift; P.interval = interval; P.thres = thres;
  return P;
}

extern "C" {

int dvgo_set_tuning(int key, int value) {
  if (key < 0 || key >= DVGO_TUNE_COUNT) return DVGO_EINVAL;
  g_tuning[key] = value;
  return 0;
}

// NOTE: xyz_min / xyz_max / xyz2ijk_scale / xyz2ijk_shift are HOST pointers (3 floats each)
// in the fused entry points: they are model constants and travel as kernel arguments.

int dvgo_march_density(const float* rays_start, const float* rays_dir, const int64_t* n_steps,
                       const int64_t* n_steps_cumsum, int64_t rec_stride, int64_t n_rays, const float* xyz_min,
                       const float* xyz_max, float stepdist, const uint8_t* mask, int mX, int mY, int mZ,
                       const float* xyz2ijk_scale, const float* xyz2ijk_shift, const float* density,
                       int X, int Y, int Z, float act_shift, float interval, float fast_color_thres,
                       dvgo_rec2_t* rec2, int32_t* n2, int32_t* n3,
                       float* alphainv_last, int32_t* brick_cnt, void* stream) {
  if (n_rays < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rays_start || !rays_dir || !n_steps || !xyz_min || !xyz_max || !density ||
      !rec2 || !n2 || !n3 || !alphainv_last)
    return DVGO_EINVAL;
  if (!n_steps_cumsum && rec_stride <= 0) return DVGO_EINVAL;
  if (mask && (!xyz2ijk_scale || !xyz2ijk_shift || mX <= 0 || mY <= 0 || mZ <= 0)) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  const MarchParams P = make_params(xyz_min, xyz_max, stepdist, xyz2ijk_scale, xyz2ijk_shift, mX, mY, mZ,
                                    X, Y, Z, act_shift, interval, fast_color_thres);
  march_density_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_start, rays_dir, n_steps, n_steps_cumsum, rec_stride, n_rays, mask, density, P, rec2, n2, n3,
      alphainv_last, brick_cnt);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_hit(const float* rays_start, const float* rays_dir, const int64_t* n_steps, int64_t n_rays,
                   const float* xyz_min, const float* xyz_max, float stepdist, const uint8_t* mask, int mX, int mY,
                   int mZ, const float* xyz2ijk_scale, const float* xyz2ijk_shift, uint8_t* hit, void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rays_start || !rays_dir || !n_steps || !xyz_min || !xyz_max || !mask || !xyz2ijk_scale || !xyz2ijk_shift || !hit ||
      mX <= 0 || mY <= 0 || mZ <= 0)
    return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  const MarchParams P = make_params(xyz_min, xyz_max, stepdist, xyz2ijk_scale, xyz2ijk_shift, mX, mY, mZ, 1, 1, 1, 0.f, 0.f, 0.f);
  march_hit_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      rays_start, rays_dir, n_steps, n_rays, mask, P, hit);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_gather(const dvgo_rec2_t* rec2, const int32_t* n2, const int64_t* n_steps, const int64_t* n_steps_cumsum,
                      int64_t rec_stride, const int64_t* off3, int64_t n_rays, int64_t M3, const float* rays_start,
                      const float* rays_dir, float stepdist, const float* xyz_min, const float* xyz_max,
                      const float* k0, int C, int X, int Y, int Z, int64_t sC, int64_t sX, int64_t sY,
                      int64_t sZ, int64_t* ray_id, int64_t* step_id, float* weights, float* alpha,
                      float* feat, void* stream) {
  if (n_rays < 0 || M3 < 0 || C < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (M3 == 0 || n_rays == 0) return 0;
  if (!rec2 || !n2 || !n_steps || !off3 || !rays_start || !rays_dir || !xyz_min || !xyz_max ||
      !ray_id || !step_id || !weights || !alpha || (C > 0 && (!k0 || !feat)))
    return DVGO_EINVAL;
  if (!n_steps_cumsum && rec_stride <= 0) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  const MarchParams P = make_params(xyz_min, xyz_max, stepdist, nullptr, nullptr, 0, 0, 0, X, Y, Z, 0.f, 0.f, 0.f);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = dvgo_blocks(n_rays * 64, DVGO_BLOCK);
  const bool vec = (sC == 1) && (C % 4 == 0) && (sX % 4 == 0) && (sY % 4 == 0) && (sZ % 4 == 0) &&
                   ((((uintptr_t)k0) & 15) == 0) && ((((uintptr_t)feat) & 15) == 0);
#define DVGO_GATHER(CV, CSS)                                                                                      \
  march_gather_kernel<CV, CSS><<<blocks, DVGO_BLOCK, 0, s>>>(rec2, n2, n_steps, n_steps_cumsum, rec_stride, off3, \
      n_rays, rays_start, rays_dir, P, k0, C, sC, sX, sY, sZ, ray_id, step_id, weights, alpha, feat)
  if (vec && C == 12) DVGO_GATHER(3, 0);
  else if (vec && C == 4) DVGO_GATHER(1, 0);
  else if (vec && C == 8) DVGO_GATHER(2, 0);
  else if (vec && C == 16) DVGO_GATHER(4, 0);
  else if (sC == 1 && C == 9) DVGO_GATHER(0, 9);
  else if (sC == 1 && C == 3) DVGO_GATHER(0, 3);
  else DVGO_GATHER(0, 0);
#undef DVGO_GATHER
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_composite(const float* weights, const float* rgb, const int64_t* step_id,
                         const int64_t* off3, int64_t n_rays, const float* alphainv_last, float bg,
                         float* rgb_marched, float* depth, void* stream) {
  if (n_rays < 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!off3 || !alphainv_last || !rgb_marched) return DVGO_EINVAL;
  // step_id / weights / rgb may be NULL when no sample survived (M3 == 0): they are only read inside [off3[r], off3[r+1])
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  march_composite_kernel<<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      weights, rgb, step_id, off3, n_rays, alphainv_last, bg, rgb_marched, depth);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_composite_bwd(const float* grad_rgb_marched, const float* weights, const float* rgb,
                             const int64_t* ray_id, int64_t M3, const int64_t* m_dev, int64_t n_rays, float bg,
                             float* grad_weights, float* grad_rgb, float* grad_last, void* stream) {
  if (M3 < 0 || n_rays < 0) return DVGO_EINVAL;
  const int64_t n_last = grad_last ? n_rays : 0;
  if (M3 == 0 && n_last == 0) return 0;
  if (!grad_rgb_marched || (M3 > 0 && (!weights || !rgb || !ray_id))) return DVGO_EINVAL;
  const int64_t threads = M3 > n_last ? M3 : n_last;
  if (!dvgo_fits(threads)) return DVGO_ERANGE;
  march_composite_bwd_kernel<<<dvgo_blocks(threads, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
      grad_rgb_marched, weights, rgb, ray_id, M3, m_dev, grad_weights, grad_rgb, n_last, bg, grad_last);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_feat_bwd(const float* grad_feat, const float* grad_extra, const int64_t* ray_id, const int64_t* step_id,
                        int64_t M3, const float* rays_start, const float* rays_dir, float stepdist,
                        const float* xyz_min, const float* xyz_max, int C, int X, int Y, int Z,
                        int64_t sC, int64_t sX, int64_t sY, int64_t sZ, float* grad_k0, void* stream) {
  if (M3 < 0 || C < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (M3 == 0 || C == 0) return 0;
  if (!grad_feat || !ray_id || !step_id || !rays_start || !rays_dir || !xyz_min || !xyz_max || !grad_k0)
    return DVGO_EINVAL;
  if (!dvgo_fits(M3 * C)) return DVGO_ERANGE;
  const MarchParams P = make_params(xyz_min, xyz_max, stepdist, nullptr, nullptr, 0, 0, 0, X, Y, Z, 0.f, 0.f, 0.f);
  hipStream_t s = (hipStream_t)stream;
  const int variant = g_tuning[DVGO_TUNE_FEAT_BWD];
  // voxel-major rows of RS = sZ floats (RS == C: the channels-last gradient itself; RS == 16: 64-byte rows of a
  // combined gradient buffer)
  const bool rows = (sC == 1) && (sZ >= C) && (sY == (int64_t)Z * sZ) && (sX == (int64_t)Y * Z * sZ) &&
                    ((int64_t)X * Y * Z < ((int64_t)1 << 31)) && ((C % 4 != 0) || ((((uintptr_t)grad_feat) & 15) == 0));
  const int RS = (int)sZ;
  const int64_t n_pass = (M3 + 31) / 32;
  const int blocks = (int)((n_pass + 3) / 4 < 4096 ? (n_pass + 3) / 4 : 4096);
#define DVGO_FEAT_BWD(CC, EX)                                                                             \
  march_feat_bwd_dedup_kernel<CC, EX><<<blocks, DVGO_BLOCK, 0, s>>>(grad_feat, grad_extra, ray_id, step_id, M3, rays_start, \
                                                                     rays_dir, P, grad_k0, RS)
  if (grad_extra != nullptr) {      // the extra channel is only built for the 12-feature, row-layout case
    if (!(rows && C == 12 && sZ >= C + 1)) return DVGO_ERANGE;
    DVGO_FEAT_BWD(12, true);
  } else if (variant == 1 && rows && (C == 12 || C == 4 || C == 8 || C == 16 || C == 9 || C == 3)) {
    if (C == 12) DVGO_FEAT_BWD(12, false);
    else if (C == 4) DVGO_FEAT_BWD(4, false);
    else if (C == 8) DVGO_FEAT_BWD(8, false);
    else if (C == 9) DVGO_FEAT_BWD(9, false);         // LLFF (lib/dmpigo.py, rgbnet_dim 9)
    else if (C == 3) DVGO_FEAT_BWD(3, false);         // coarse stage (k0 = RGB)
    else DVGO_FEAT_BWD(16, false);
  } else {
    march_feat_bwd_kernel<<<dvgo_blocks(M3 * C, DVGO_BLOCK), DVGO_BLOCK, 0, s>>>(
        grad_feat, ray_id, step_id, M3, rays_start, rays_dir, P, C, sC, sX, sY, sZ, grad_k0);
  }
#undef DVGO_FEAT_BWD
  DVGO_LAUNCH_CHECK();
  return 0;
}

// combined gradient rows [n_vox][RS] -> channels-last feature gradient [n_vox][C] and density gradient [n_vox]
__global__ void __launch_bounds__(DVGO_BLOCK)
grid_grad_split_kernel(const float* __restrict__ G, int64_t n_vox, float* __restrict__ grad_k0,
                       float* __restrict__ grad_density) {
  // 16-float rows, 12 + 1 channels: 4 lanes per row, float4 each
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t >> 2;
  const int q = (int)(t & 3);
  if (row >= n_vox) return;
  const float4 v = reinterpret_cast<const float4*>(G)[row * 4 + q];
  if (q < 3) reinterpret_cast<float4*>(grad_k0)[row * 3 + q] = v;
  else grad_density[row] = v.x;
}

int dvgo_grid_grad_split(const float* G, int64_t n_vox, int row_stride, int C, float* grad_k0, float* grad_density,
                         void* stream) {
  if (n_vox < 0) return DVGO_EINVAL;
  if (n_vox == 0) return 0;
  if (!G || !grad_k0 || !grad_density) return DVGO_EINVAL;
  if (row_stride != 16 || C != 12) return DVGO_ERANGE;
  if (!dvgo_fits(n_vox * 4)) return DVGO_ERANGE;
  grid_grad_split_kernel<<<dvgo_blocks(n_vox * 4, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(G, n_vox, grad_k0,
                                                                                                   grad_density);
  DVGO_LAUNCH_CHECK();
  return 0;
}

int dvgo_march_density_bwd(const dvgo_rec2_t* rec2, const int32_t* n2, const int64_t* n_steps,
                           const int64_t* n_steps_cumsum, int64_t rec_stride, const int64_t* off3, int64_t n_rays,
                           const float* rays_start, const float* rays_dir, float stepdist,
                           const float* xyz_min, const float* xyz_max, const float* alphainv_last,
                           float interval, const float* grad_weights, const float* grad_last, int X,
                           int Y, int Z, float* grad_density, int64_t grad_stride, float* grad_kept,
                           int32_t* brick_cursor, void* brick_recs, void* stream) {
  if (n_rays < 0 || X <= 0 || Y <= 0 || Z <= 0) return DVGO_EINVAL;
  if (n_rays == 0) return 0;
  if (!rec2 || !n2 || !n_steps || !off3 || !rays_start || !rays_dir || !xyz_min ||
      !xyz_max || !alphainv_last || (!grad_density && !brick_cursor))
    return DVGO_EINVAL;       // grad_weights may be NULL when M3 == 0 (it is only read for flagged samples)
  if (brick_cursor && !brick_recs) return DVGO_EINVAL;
  if ((!n_steps_cumsum && rec_stride <= 0) || (!brick_cursor && grad_stride <= 0)) return DVGO_EINVAL;
  if (!dvgo_fits(n_rays * 64)) return DVGO_ERANGE;
  const MarchParams P = make_params(xyz_min, xyz_max, stepdist, nullptr, nullptr, 0, 0, 0, X, Y, Z, 0.f,
                                    interval, 0.f);
  if (!brick_cursor && g_tuning[DVGO_TUNE_DENSITY_BWD] == 1 && (int64_t)X * Y * Z < ((int64_t)1 << 31))
    march_density_bwd_kernel<true><<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
        rec2, n2, n_steps, n_steps_cumsum, rec_stride, off3, n_rays, rays_start, rays_dir, P, alphainv_last,
        grad_weights, grad_last, grad_density, grad_stride, grad_kept, nullptr, nullptr);
  else
    march_density_bwd_kernel<false><<<dvgo_blocks(n_rays * 64, DVGO_BLOCK), DVGO_BLOCK, 0, (hipStream_t)stream>>>(
        rec2, n2, n_steps, n_steps_cumsum, rec_stride, off3, n_rays, rays_start, rays_dir, P, alphainv_last,
        grad_weights, grad_last, grad_density, grad_stride, grad_kept, brick_cursor, (int4*)brick_recs);
  DVGO_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
