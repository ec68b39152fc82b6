// Microbenchmark: LDS atomic throughput on one CU's worth of waves (gfx950).  Build: hipcc --offload-arch=gfx950 -O3
// -munsafe-fp-atomics -o lds_atomics.x lds_atomics.hip ; run: ./lds_atomics.x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int SLOTS = 16384, ITERS = 2000, THREADS = 1024;
enum Mode { CAS_RANDOM, CAS_LINEAR, CAS_8LANES_EXEC, CAS_8LANES_DUMMY, ADDF_RANDOM, READ_RANDOM, CAS64_RANDOM, ADDF_LINEAR, CAS_SAMEBANK,
            ADDF_6LANES_EXEC, ADDF_1LANE_EXEC, ADDF_6LANES_ZERO, ADDU_RANDOM, ADDF_RTN_RANDOM, ADDF_CASLOOP, ADDF_32LANES_EXEC, OR_RANDOM, CAS_X4, READ_X4, CAS64_X4, OR_X4, ADDU_X4, NMODES };
__global__ __launch_bounds__(THREADS) void k(int mode, const unsigned* __restrict__ rnd, unsigned long long* out, int* sink) {
  __shared__ __attribute__((aligned(16))) int tab[SLOTS * 2];
  __shared__ int dummy[THREADS];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < SLOTS * 2; i += THREADS) tab[i] = -1;
  __syncthreads();
  unsigned r = rnd[blockIdx.x * THREADS + tid];
  int acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
    r = r * 1664525u + 1013904223u;
    const int a = (int)(r >> 18);               // 0..16383
    switch (mode) {
      case CAS_RANDOM: acc += atomicCAS(&tab[a], -1, it); break;
      case CAS_LINEAR: acc += atomicCAS(&tab[(tid + it * 64) & (SLOTS - 1)], -1, it); break;
      case CAS_8LANES_EXEC: if (lane < 8) acc += atomicCAS(&tab[a], -1, it); break;
      case CAS_8LANES_DUMMY: acc += atomicCAS(lane < 8 ? &tab[a] : &dummy[tid], -1, it); break;
      case ADDF_RANDOM: atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f); break;
      case ADDF_LINEAR: atomicAdd(reinterpret_cast<float*>(&tab[(tid + it * 64) & (SLOTS - 1)]), 1.0f); break;
      case READ_RANDOM: acc += tab[a]; break;
      case CAS64_RANDOM: acc += (int)atomicCAS(reinterpret_cast<unsigned long long*>(&tab[2 * a]), ~0ull, (unsigned long long)it); break;
      case CAS_SAMEBANK: acc += atomicCAS(&tab[(a & ~63) | 5], -1, it); break;   // all lanes bank 5
      case ADDF_6LANES_EXEC: if (lane < 6) atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f); break;
      case ADDF_32LANES_EXEC: if (lane < 32) atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f); break;
      case ADDF_1LANE_EXEC: if (lane == 0) atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f); break;
      case ADDF_6LANES_ZERO: atomicAdd(reinterpret_cast<float*>(lane < 6 ? &tab[a] : &dummy[tid]), lane < 6 ? 1.0f : 0.f); break;
      case ADDU_RANDOM: atomicAdd(reinterpret_cast<unsigned*>(&tab[a]), 1u); break;
      case ADDF_RTN_RANDOM: acc += (int)atomicAdd(reinterpret_cast<float*>(&tab[a]), 1.0f); break;
      case OR_RANDOM: atomicOr(reinterpret_cast<unsigned*>(&tab[a]), 1u << (r & 31)); break;
      case CAS_X4: { int t = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += atomicCAS(&tab[(a + q * 4099) & (SLOTS - 1)], -1, it);
        acc += t; } break;
      case READ_X4: { int t = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += tab[(a + q * 4099) & (SLOTS - 1)];
        acc += t; } break;
      case CAS64_X4: { int t = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) t += (int)atomicCAS(reinterpret_cast<unsigned long long*>(&tab[2 * ((a + q * 4099) & (SLOTS - 1))]), ~0ull, (unsigned long long)it);
        acc += t; } break;
      case OR_X4: {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicOr(reinterpret_cast<unsigned*>(&tab[(a + q * 4099) & (SLOTS - 1)]), 1u << (r & 31)); } break;
      case ADDU_X4: {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(reinterpret_cast<unsigned*>(&tab[(a + q * 4099) & (SLOTS - 1)]), 1u); } break;
      case ADDF_CASLOOP: { const int old = tab[a]; acc += atomicCAS(&tab[a], old, __float_as_int(__int_as_float(old) + 1.0f)); } break;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 123456789) *sink = acc;
}
int main() {
  unsigned* drnd; unsigned long long* dout; int* dsink;
  const int blocks = 256;
  std::vector<unsigned> h(blocks * THREADS);
  for (auto& x : h) x = (unsigned)rand() * 2654435761u + (unsigned)rand();
  hipMalloc(&drnd, h.size() * 4); hipMalloc(&dout, blocks * 8); hipMalloc(&dsink, 4);
  hipMemcpy(drnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const char* names[] = {"CAS32 random", "CAS32 linear (conflict-free)", "CAS32 random, 8 lanes by EXEC", "CAS32 8 lanes real + 56 private dummies",
                         "ADD f32 random", "READ b32 random", "CAS64 random", "ADD f32 linear", "CAS32 all lanes same bank", "ADD f32 random, 6 lanes by EXEC", "ADD f32 random, 1 lane by EXEC", "ADD f32 6 lanes real + 58 dummies(+0)", "ADD u32 random", "ADD f32 rtn random", "f32 add by read+CAS32 (one try)", "ADD f32 random, 32 lanes by EXEC", "OR b32 random", "4 x CAS32 random per iteration", "4 x READ b32 random per iteration", "4 x CAS64 random per iteration", "4 x OR b32 per iteration", "4 x ADD u32 per iteration"};
  for (int m = 0; m < NMODES; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(THREADS), 0, 0, m, drnd, dout, dsink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) {
        // one 1024-thread block per CU (LDS 132 KB): 16 waves x ITERS instructions per CU
        const double ns_per_instr_per_cu = ms * 1e6 / (16.0 * ITERS);
        printf("%-45s %8.3f ms  -> %6.2f ns per wave-instruction per CU (~%5.1f cycles @2.4GHz)\n", names[m], ms, ns_per_instr_per_cu, ns_per_instr_per_cu * 2.4);
      }
    }
  }
  return 0;
}
