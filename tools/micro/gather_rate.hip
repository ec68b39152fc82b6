// Microbenchmark: what the memory system gives for the gather pattern of the SpGEMM kernels (gfx950).
// A wave-load = 64 lanes x 4 (or 8) bytes; the lanes form runs of RUN consecutive elements (one B row), every run starts
// at a random place of a 128 MB array (B of the headline matrix is 125 MB: beyond L2, inside the Infinity Cache).
// K independent wave-loads are in flight per wave before their results are consumed; W waves per block, blocks fill the
// chip.  Reported: wave-loads per microsecond per CU and the implied bytes of distinct 128-byte lines per second.
// The numeric kernels of round 2 issue 2 wave-loads per 64 products and reach ~15 wave-loads/us/CU in total.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o gather_rate.x gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int ITERS = 64;
template <int K, int BYTES>
__global__ void k(const int* __restrict__ arr, unsigned n, int run, unsigned seed, int* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned r = seed ^ (wave * 2654435761u);
  int acc = 0;
  for (int it = 0; it < ITERS; ++it) {
    int v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      // run g of this wave-load starts at a random element; lane's element = start + (lane % run)
      const unsigned g = (unsigned)lane / (unsigned)run;
      unsigned h = (r + g * 0x9E3779B9u + q * 0x85EBCA6Bu) * 2246822519u;
      h ^= h >> 15; h *= 3266489917u; h ^= h >> 13;
      const unsigned start = (unsigned)(((unsigned long long)h * (n - 4096u)) >> 32);
      const unsigned idx = start + (unsigned)lane % (unsigned)run;
      if (BYTES == 4) v[q] = arr[idx];
      else { const int2 t = reinterpret_cast<const int2*>(arr)[idx >> 1]; v[q] = t.x + t.y; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < K; ++q) acc += v[q];
    r = r * 1664525u + 1013904223u + (unsigned)(acc & 1);   // next addresses depend on the data: no run-ahead across iterations
  }
  if (acc == 0x7fffffff) *sink = 1;
}
template <int K, int BYTES>
void run(const int* arr, unsigned n, int runlen, int wavesPerBlock, int blocksPerCU, int cus, int* sink) {
  const dim3 grid(cus * blocksPerCU), block(64 * wavesPerBlock);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<K, BYTES>), grid, block, 0, 0, arr, n, runlen, 1u, sink);
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<K, BYTES>), grid, block, 0, 0, arr, n, runlen, 7u + rep, sink);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double wl = (double)grid.x * wavesPerBlock * ITERS * K;
  const double lines = wl * ((64 + runlen - 1) / runlen) * ((runlen * BYTES + 127) / 128 + (runlen * BYTES % 128 ? 0.5 : 0));
  printf("run %3d  %dB/lane  K=%d  waves/CU=%3d : %7.3f ms  %6.1f wave-loads/us/CU  ~%5.2f TB/s of 128B lines\n", runlen, BYTES, K,
         wavesPerBlock * blocksPerCU, ms, wl / ms / 1e3 / cus, lines * 128 / ms / 1e9);
}
int main() {
  hipSetDevice(0);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const unsigned n = 32u << 20;                       // 32 M ints = 128 MB
  int* arr; hipMalloc(&arr, (size_t)n * 4); hipMemset(arr, 1, (size_t)n * 4);
  int* sink; hipMalloc(&sink, 4);
  printf("CUs=%d, array 128 MB, %d dependent iterations per wave\n", cus, ITERS);
  for (int runlen : {1, 4, 8, 16, 64})
    for (int waves : {16, 24, 32}) {
      run<1, 4>(arr, n, runlen, 1, waves, cus, sink);
      run<2, 4>(arr, n, runlen, 1, waves, cus, sink);
      run<4, 4>(arr, n, runlen, 1, waves, cus, sink);
      run<8, 4>(arr, n, runlen, 1, waves, cus, sink);
    }
  for (int runlen : {4, 8, 16}) { run<2, 8>(arr, n, runlen, 1, 24, cus, sink); run<4, 8>(arr, n, runlen, 1, 24, cus, sink); }
  return 0;
}
