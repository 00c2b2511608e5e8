// Single-workgroup scans shared by csrc/brick.hip (the scan kernels) and csrc/march.hip (the tail of march_density: the
// last workgroup to finish turns the per-ray survivor counts into output offsets and the brick counts into the brick tables,
// which saves the scan launch between march_density and march_gather).
#pragma once
#include "common.h"

// Exclusive scan of n values by ONE workgroup of 64 * NW threads, 8 * 64 * NW values per pass (one pass of a 1024-thread
// workgroup covers a training batch's rays and the bricks of a 160^3 grid): every thread loads its 8 values of the pass up
// front (coalesced: value r * T + tid), the 8 rows are scanned per wave with shuffles, the 8 x NW wave totals by wave 0 through
// LDS, and each value is handed its exclusive prefix (and itself).  One memory round trip and two barriers per pass.  Values
// are 64-bit so that several running sums can ride in one scan.  Returns the total.
// eight int32 values p[i0], p[i0 + stride], ... through write-through-coherent (sc1) loads, all eight in flight together
// (a C++ atomic load per value is waited for one by one: eight round trips); indices are clamped to [0, n), values past n
// read as 0.  The reader of another CU's sc1 stores (cdna_hip_programming.md Guideline 16, R1).
__device__ __forceinline__ void dvgo_load8_sc1(const int32_t* __restrict__ p, int i0, int stride, int n, int (&v)[8]) {
  const int32_t* a[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { const int i = i0 + r * stride; a[r] = p + (i < n ? i : n - 1); }
  asm volatile(
      "global_load_dword %0, %8, off sc1\n\t"
      "global_load_dword %1, %9, off sc1\n\t"
      "global_load_dword %2, %10, off sc1\n\t"
      "global_load_dword %3, %11, off sc1\n\t"
      "global_load_dword %4, %12, off sc1\n\t"
      "global_load_dword %5, %13, off sc1\n\t"
      "global_load_dword %6, %14, off sc1\n\t"
      "global_load_dword %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
      : "memory");
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = (i0 + r * stride < n) ? v[r] : 0;
}

// `load8(base, tid, T, v)`: fills v[r] (r < 8) with value base + r * T + tid, 0 past n -- one call per pass, so that a
// loader can keep all eight requests in flight (dvgo_load8_sc1) where a per-value load would be waited for eight times
template <int NW, typename Load8, typename Store>
__device__ __forceinline__ unsigned long long block_scan8_u64(int n, Load8 load8, Store store) {
  constexpr int R = 8, T = 64 * NW;
  static_assert(R * NW <= 128, "the wave totals are scanned two per lane by one wave");
  __shared__ unsigned long long s_part[R * NW];
  __shared__ unsigned long long s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long carry = 0ull;
  for (int base = 0; base < n; base += R * T) {
    unsigned long long v[R], inc[R];
    load8(base, tid, T, v);
#pragma unroll
    for (int r = 0; r < R; ++r) inc[r] = v[r];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const unsigned long long o = __shfl_up(inc[r], d);
        if (lane >= d) inc[r] += o;
      }
    }
    if (lane == 63) {
#pragma unroll
      for (int r = 0; r < R; ++r) s_part[r * NW + wave] = inc[r];
    }
    __syncthreads();
    if (wave == 0) {                               // the R * NW wave totals, in value order: two per lane
      const bool in = 2 * lane < R * NW;
      const unsigned long long p0 = in ? s_part[2 * lane] : 0ull, p1 = in ? s_part[2 * lane + 1] : 0ull;
      unsigned long long t = p0 + p1;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(t, d);
        if (lane >= d) t += o;
      }
      if (in) {
        s_part[2 * lane] = t - p0 - p1;
        s_part[2 * lane + 1] = t - p1;
      }
      if (lane == 63) s_total = t;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = base + r * T + tid;
      if (i < n) store(i, carry + s_part[r * NW + wave] + inc[r] - v[r], v[r]);
    }
    carry += s_total;
    __syncthreads();
  }
  return carry;
}

// the same with a per-value loader
template <int NW, typename Load, typename Store>
__device__ __forceinline__ unsigned long long block_scan_u64(int n, Load load, Store store) {
  return block_scan8_u64<NW>(n, [&](int base, int tid, int T, unsigned long long (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int i = base + r * T + tid;
      v[r] = (i < n) ? load(i) : 0ull;
    }
  }, store);
}

// exclusive scan of n int32 counts (read through `load`): out[i] = sum of cnt[0..i), out[n] = total
template <int NW, typename OutT, typename Load>
__device__ __forceinline__ void block_scan_i32(int n, OutT* __restrict__ out, Load load) {
  const unsigned long long total = block_scan_u64<NW>(
      n, [&](int i) { return (unsigned long long)load(i); }, [&](int i, unsigned long long ex, unsigned long long) { out[i] = (OutT)ex; });
  if (threadIdx.x == 0) out[n] = (OutT)total;
}

// out = exclusive scan of n int32 counts written by OTHER workgroups of the running kernel with sc1 stores
template <int NW, typename OutT>
__device__ __forceinline__ void block_scan_i32_sc1(const int32_t* __restrict__ cnt, int n, OutT* __restrict__ out) {
  const unsigned long long total = block_scan8_u64<NW>(
      n,
      [&](int base, int tid, int T, unsigned long long (&v)[8]) {
        int q[8];
        dvgo_load8_sc1(cnt, base + tid, T, n, q);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (unsigned long long)q[r];
      },
      [&](int i, unsigned long long ex, unsigned long long) { out[i] = (OutT)ex; });
  if (threadIdx.x == 0) out[n] = (OutT)total;
}

// Heavy bricks (a thin surface crossed by every ray: tens of thousands of entries where the median brick has hundreds)
// are cut into SLICES of `slice_len` entries.  Slice 0 is the brick's own work item; the further slices are EXTRA work
// items appended after the bricks, and the slices of a brick meet in scratch tiles (see brick_accumulate_kernel).
// This scan turns the per-brick counts (read through `load`) into
//   off      [nb + 1]  first entry of each brick's list (and `cursor`, the fill cursors)
//   extra    [nb + 1]  first extra work item of each brick (ceil(cnt / slice_len) - 1 of them, none for most)
//   active   [nb + 1]  the non-empty bricks, in brick order; active[nb] = their number (a sparse scene touches a tenth
//                      of the bricks: the workgroups beyond that number leave after one load)
//   extra_brick [<= n_extra_max]  the brick of every extra work item
// and clears the counters, which then serve as the arrival counters of the slices.
#define DVGO_BRICK_SLICE_DEFAULT 1024   // entries per work item

template <int NW, bool SC1>        // SC1: the counts were accumulated by other workgroups of the running kernel (atomics)
__device__ __forceinline__ void brick_tables(int32_t* __restrict__ cnt, int nb, int32_t* __restrict__ off,
                                             int32_t* __restrict__ cursor, int32_t* __restrict__ extra, int32_t* __restrict__ active,
                                             int32_t* __restrict__ extra_brick, int n_extra_max, int slice_len) {
  // the three running sums in one 64-bit scan: entries (28 bits) | extra items (18) | non-empty bricks (18)
  auto slices = [&](int c) { return c > slice_len ? (c + slice_len - 1) / slice_len : 1; };
  const unsigned long long total = block_scan8_u64<NW>(
      nb,
      [&](int base, int tid, int T, unsigned long long (&v)[8]) {
        int q[8];
        if (SC1) {
          dvgo_load8_sc1(cnt, base + tid, T, nb, q);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) { const int i = base + r * T + tid; q[r] = (i < nb) ? cnt[i] : 0; }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int c = q[r];
          v[r] = !extra ? (unsigned long long)c
                        : ((unsigned long long)c | ((unsigned long long)(slices(c) - 1) << 28) | ((unsigned long long)(c > 0) << 46));
        }
      },
      [&](int i, unsigned long long ex, unsigned long long v) {
        const int e0 = extra ? (int)(ex & 0xfffffffull) : (int)ex;
        off[i] = e0; cursor[i] = e0;
        if (extra) {
          const int e1 = (int)((ex >> 28) & 0x3ffffull), n_extra = (int)((v >> 28) & 0x3ffffull);
          extra[i] = e1; cnt[i] = 0;
          if (v >> 46) active[(int)(ex >> 46)] = i;
          for (int k = 0; k < n_extra; ++k)
            if (e1 + k < n_extra_max) extra_brick[e1 + k] = i;
        }
      });
  if (threadIdx.x == 0) {
    off[nb] = extra ? (int)(total & 0xfffffffull) : (int)total;
    if (extra) { extra[nb] = (int)((total >> 28) & 0x3ffffull); active[nb] = (int)(total >> 46); }
  }
}
