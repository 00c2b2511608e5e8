// Micro-benchmark: what does an LDS atomic cost on gfx950?  One workgroup of 256 threads per CU issues `iters` rounds of
// 8 atomics per thread on a 512-entry LDS table, for {ds_add_rtn_u32, ds_add_u32, ds_add_f32} x {every lane its own
// address (stride 1), random addresses, 2 / 4 / 16 lanes per address}.  Reported: cycles per wave instruction per CU
// (4 waves in flight), i.e. how long the LDS atomic unit is busy with one 64-lane instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic.hip -o tools/micro/lds_atomic && tools/micro/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(const int* __restrict__ addr, int iters, int* __restrict__ out, unsigned long long* cyc) {
  __shared__ int tab[512];
  tab[threadIdx.x] = 0; tab[threadIdx.x + 256] = 0;
  int a[8];
  for (int q = 0; q < 8; ++q) a[q] = addr[threadIdx.x * 8 + q];
  __syncthreads();
  int acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (OP == 0) acc += atomicAdd(&tab[a[q]], 1);
      else if (OP == 1) __hip_atomic_fetch_add(&tab[a[q]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)&tab[a[q]], 1.0f, 0, 0, false);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc + tab[threadIdx.x];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 2000;
  int *d_addr, *d_out; unsigned long long* d_cyc;
  hipMalloc(&d_addr, 2048 * 4); hipMalloc(&d_out, blocks * 256 * 4); hipMalloc(&d_cyc, blocks * 8);
  const char* opn[3] = {"ds_add_rtn_u32", "ds_add_u32    ", "ds_add_f32    "};
  const char* patn[5] = {"own address", "random", "2 lanes/addr", "4 lanes/addr", "16 lanes/addr"};
  for (int pat = 0; pat < 5; ++pat) {
    std::vector<int> h(2048);
    srand(1);
    for (int t = 0; t < 256; ++t)
      for (int q = 0; q < 8; ++q) {
        int v;
        if (pat == 0) v = (t + 37 * q) & 511;
        else if (pat == 1) v = rand() & 511;
        else { const int share = pat == 2 ? 2 : pat == 3 ? 4 : 16; v = ((t / share) * 13 + 37 * q) & 511; }
        h[t * 8 + q] = v;
      }
    hipMemcpy(d_addr, h.data(), 2048 * 4, hipMemcpyHostToDevice);
    for (int op = 0; op < 3; ++op) {
      for (int rep = 0; rep < 2; ++rep) {
        if (op == 0) k<0><<<blocks, 256>>>(d_addr, iters, d_out, d_cyc);
        else if (op == 1) k<1><<<blocks, 256>>>(d_addr, iters, d_out, d_cyc);
        else k<2><<<blocks, 256>>>(d_addr, iters, d_out, d_cyc);
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> c(blocks);
      hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : c) s += (double)v;
      s /= blocks;
      // per CU: 4 waves x 8 x iters wave instructions
      printf("%s  %-14s %7.1f cycles per 64-lane instruction per CU\n", opn[op], patn[pat], s / (4.0 * 8 * iters));
    }
  }
  return 0;
}
