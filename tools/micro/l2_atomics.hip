// Microbenchmark: scattered global float atomics into a block-private, L2-sized window (gfx950).
// Question it answers (round 3): can the rows above 4096 products accumulate by RANK straight into their final range of C
// (global_atomic_add_f32, no LDS table, no probing, no parking) at a rate that beats the LDS hash kernel (~100 G products/s)?
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o l2_atomics.x l2_atomics.hip ; run: ./l2_atomics.x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int THREADS = 1024, ITERS = 256, UNR = 4;
enum Mode { ATOMIC_F32 = 0, STORE_F32, LOAD_F32, ATOMIC_F32_SORTEDISH, ATOMIC_F32_RTN, NMODES };
__global__ __launch_bounds__(THREADS) void k(int mode, int win, float* __restrict__ buf, const unsigned* __restrict__ rnd, int* sink) {
  float* const w = buf + (size_t)blockIdx.x * (size_t)win;
  unsigned r = rnd[blockIdx.x * THREADS + threadIdx.x];
  float acc = 0.f;
  for (int it = 0; it < ITERS; it += UNR) {
    int a[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      r = r * 1664525u + 1013904223u;
      a[u] = (int)(((unsigned long long)(r >> 4) * (unsigned)win) >> 28);
      if (mode == ATOMIC_F32_SORTEDISH) a[u] = (int)(((unsigned long long)(threadIdx.x * 16 + (r >> 28)) * (unsigned)win) >> 14) % win;  // lanes of a wave within a few lines
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      switch (mode) {
        case ATOMIC_F32: case ATOMIC_F32_SORTEDISH: atomicAdd(&w[a[u]], 1.0f); break;
        case ATOMIC_F32_RTN: acc += atomicAdd(&w[a[u]], 1.0f); break;
        case STORE_F32: __builtin_nontemporal_store(1.0f, &w[a[u]]); break;
        case LOAD_F32: acc += w[a[u]]; break;
      }
    }
  }
  if (acc == 12345.678f) *sink = 1;
}
int main() {
  int dev = 0; hipSetDevice(dev);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
  const int blocks = p.multiProcessorCount;
  const int wins[] = {4096, 16384, 65536, 262144};
  std::vector<unsigned> hr((size_t)blocks * THREADS);
  unsigned x = 777u; for (auto& v : hr) { x = x * 1664525u + 1013904223u; v = x; }
  unsigned* dr; hipMalloc(&dr, hr.size() * 4); hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice);
  int* sink; hipMalloc(&sink, 4);
  float* buf; hipMalloc(&buf, (size_t)blocks * 262144 * 4); hipMemset(buf, 0, (size_t)blocks * 262144 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[NMODES] = {"atomic_add_f32 (no return)", "store f32 (nt)", "load f32", "atomic_add_f32 lanes on neighbouring lines", "atomic_add_f32 with return"};
  printf("CUs=%d, %d threads/block, %d ops per thread\n", blocks, THREADS, ITERS);
  for (int wi = 0; wi < 4; ++wi)
    for (int mode = 0; mode < NMODES; ++mode) {
      hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), 0, 0, mode, wins[wi], buf, dr, sink);   // warm-up
      hipEventRecord(e0, 0);
      for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), 0, 0, mode, wins[wi], buf, dr, sink);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double ops = (double)blocks * THREADS * ITERS;
      printf("window %7d floats (%4d KB/block)  %-44s %8.3f ms  %7.1f G ops/s\n", wins[wi], wins[wi] * 4 / 1024, names[mode], ms, ops / ms / 1e6);
    }
  return 0;
}
