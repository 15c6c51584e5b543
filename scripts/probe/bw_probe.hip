// HBM bandwidth probe for MI355X: what a streaming read (and read + 0.64x write) of 1 GiB can reach, per access style.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probe/bw_probe scripts/probe/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int NT, int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const v4u* __restrict__ in, unsigned* out, size_t n16, int per_block) {
  // each block reads per_block*256*UNROLL consecutive 16-byte chunks
  size_t base = (size_t)blockIdx.x * per_block * 256 * UNROLL;
  unsigned acc = 0;
  for (int it = 0; it < per_block; ++it) {
    v4u v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      size_t i = base + (size_t)it * 256 * UNROLL + u * 256 + threadIdx.x;
      if (i >= n16) i = n16 - 1;
      v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int NT>
__global__ __launch_bounds__(256) void rw_kernel(const v4u* __restrict__ in, v4u* __restrict__ out, size_t n16, int per_block) {
  // read 16 B per lane x5, write 16 B per lane x3 (0.6x) 
  size_t base = (size_t)blockIdx.x * per_block * 256 * 5;
  size_t obase = (size_t)blockIdx.x * per_block * 256 * 3;
  for (int it = 0; it < per_block; ++it) {
    v4u v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      size_t i = base + (size_t)it * 256 * 5 + u * 256 + threadIdx.x;
      if (i >= n16) i = n16 - 1;
      v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      size_t o = obase + (size_t)it * 256 * 3 + u * 256 + threadIdx.x;
      v4u w = v[u] ^ v[u + 2];
      if (NT) __builtin_nontemporal_store(w, out + o); else out[o] = w;
    }
  }
}

int main() {
  const size_t N = 1ull << 30, n16 = N / 16;
  v4u *in, *out; unsigned* flag;
  hipMalloc(&in, N); hipMalloc(&out, N); hipMalloc(&flag, 64);
  hipMemset(in, 1, N); hipMemset(out, 0, N);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch, double bytes) {
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
  };
  for (int per_block : {1, 4, 16}) {
    {
      int grid = (int)((n16 + (size_t)per_block * 256 * 4 - 1) / ((size_t)per_block * 256 * 4));
      char nm[96];
      snprintf(nm, 96, "read plain unroll4 per_block=%d grid=%d", per_block, grid);
      timeit(nm, [&] { hipLaunchKernelGGL((read_kernel<0, 4>), dim3(grid), dim3(256), 0, 0, in, flag, n16, per_block); }, (double)N);
      snprintf(nm, 96, "read nt    unroll4 per_block=%d grid=%d", per_block, grid);
      timeit(nm, [&] { hipLaunchKernelGGL((read_kernel<1, 4>), dim3(grid), dim3(256), 0, 0, in, flag, n16, per_block); }, (double)N);
    }
    {
      int grid = (int)((n16 + (size_t)per_block * 256 * 8 - 1) / ((size_t)per_block * 256 * 8));
      char nm[96];
      snprintf(nm, 96, "read plain unroll8 per_block=%d grid=%d", per_block, grid);
      timeit(nm, [&] { hipLaunchKernelGGL((read_kernel<0, 8>), dim3(grid), dim3(256), 0, 0, in, flag, n16, per_block); }, (double)N);
    }
    {
      int grid = (int)(n16 / ((size_t)per_block * 256 * 5));
      char nm[96];
      snprintf(nm, 96, "read 1.0 + write 0.6 plain per_block=%d", per_block);
      timeit(nm, [&] { hipLaunchKernelGGL((rw_kernel<0>), dim3(grid), dim3(256), 0, 0, in, out, n16, per_block); }, (double)N * 1.6);
      snprintf(nm, 96, "read 1.0 + write 0.6 nt    per_block=%d", per_block);
      timeit(nm, [&] { hipLaunchKernelGGL((rw_kernel<1>), dim3(grid), dim3(256), 0, 0, in, out, n16, per_block); }, (double)N * 1.6);
    }
  }
  return 0;
}
