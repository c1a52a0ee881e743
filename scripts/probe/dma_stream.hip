// dma_stream.hip -- microprobe: what rate does a persistent workgroup per CU reach streaming HBM into LDS with LDS-DMA
// (global_load_lds, 1 KiB per wave instruction), as a function of the pieces each wave keeps in flight, the waves per
// workgroup, the number of separate streams and a workgroup barrier every so many pieces?  Nothing reads the LDS.
// build: hipcc --offload-arch=gfx950 -O3 -o dma_stream dma_stream.hip ; run: ./dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// each workgroup streams `pieces` KiB-pieces per stream; piece p of stream s = base_s + (wg * pieces + p) * 1024
template <int WAVES, int DEPTH, int STREAMS, int BAR>
__global__ __launch_bounds__(64 * WAVES, 1) void stream_kernel(const char* a, const char* b, int pieces, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  constexpr int RING = DEPTH + 1;
  char* ring = smem + wave * RING * 1024;
  const int n = pieces / WAVES * STREAMS;   // pieces of this wave over all streams
  const size_t wg0 = (size_t)blockIdx.x * pieces * 1024;
  auto issue = [&](int i) {
    const int s = i % STREAMS, p = wave + WAVES * (i / STREAMS);
    const char* src = (s == 0 ? a : b) + wg0 + (size_t)p * 1024 + lane * 16;
    __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(ring + (i % RING) * 1024), 16, 0, 0);
  };
  for (int i = 0; i < DEPTH && i < n; ++i) issue(i);
  for (int i = 0; i < n; ++i) {
    if (i + DEPTH < n) { issue(i + DEPTH); wait_vm<DEPTH>(); }
    else wait_vm<0>();
    if (BAR > 0 && (i % (BAR > 0 ? BAR : 1)) == 0) __builtin_amdgcn_s_barrier();
  }
  if (sink && lane == 0 && wave == 0) sink[blockIdx.x] = *(volatile int*)smem;
}

// the weight-gradient kernels' pattern: per chunk every wave issues 8 pieces of stream b then 8 of stream a (piece
// u = wave + 4t = rows 8u..8u+7 of a 256-row x 128-B image, SWZ: 16-B slots of a row XOR-permuted on the source
// side), waits until all but its youngest KEEP pieces have landed, and meets the others at a barrier
template <int KEEP, bool SWZ>
__global__ __launch_bounds__(256, 1) void chunk_kernel(const char* a, const char* b, int chunks, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const size_t wg0 = (size_t)blockIdx.x * chunks * 32768;
  uint32_t off[8];
  for (int t = 0; t < 8; ++t) {
    const int r = 8 * (wave + 4 * t) + (lane >> 3), sl = SWZ ? (lane & 7) ^ ((r >> 1) & 7) : (lane & 7);
    off[t] = (uint32_t)(r * 128 + 16 * sl);
  }
  for (int c = 0; c < chunks; ++c) {
    char* dst = smem + (c % 5) * 32768;
    char* dst2 = smem + ((c + 2) % 5) * 32768;
#pragma unroll
    for (int t = 0; t < 8; ++t)
      __builtin_amdgcn_global_load_lds(GPTR(b + wg0 + (size_t)c * 32768 + off[t]), LPTR(dst + (wave + 4 * t) * 1024), 16, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t)
      __builtin_amdgcn_global_load_lds(GPTR(a + wg0 + (size_t)c * 32768 + off[t]), LPTR(dst2 + (wave + 4 * t) * 1024), 16, 0, 0);
    wait_vm<KEEP>();
    __builtin_amdgcn_s_barrier();
  }
  wait_vm<0>();
  if (sink && lane == 0 && wave == 0) sink[blockIdx.x] = *(volatile int*)smem;
}
template <int KEEP, bool SWZ>
static void run_chunk(const char* a, const char* b, size_t bytes_per_stream, int* sink) {
  const int grid = 256, chunks = (int)(bytes_per_stream / 32768 / grid);
  const size_t lds = 5 * 32768;
  hipFuncSetAttribute((const void*)chunk_kernel<KEEP, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chunk_kernel<KEEP, SWZ>), dim3(grid), dim3(256), lds, 0, a, b, chunks, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  const double bytes = (double)chunks * 65536 * grid;
  printf("chunk pattern: keep %2d swizzle %d: %.3f ms  %.2f TB/s  (%.2f us per 64-KiB chunk)\n", KEEP, (int)SWZ, best,
         bytes / best / 1e9, 1e3 * best / chunks);
}

template <int WAVES, int DEPTH, int STREAMS, int BAR>
static void run(const char* a, const char* b, size_t bytes_per_stream, int grid, int* sink) {
  const int pieces = (int)(bytes_per_stream / 1024 / grid) / WAVES * WAVES;
  const size_t lds = (size_t)WAVES * (DEPTH + 1) * 1024;
  hipFuncSetAttribute((const void*)stream_kernel<WAVES, DEPTH, STREAMS, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<WAVES, DEPTH, STREAMS, BAR>), dim3(grid), dim3(64 * WAVES), lds, 0, a, b, pieces, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it > 0 && ms < best) best = ms;
  }
  const double bytes = (double)pieces * 1024 * grid * STREAMS;
  printf("waves %d depth %2d (%3zu KiB in flight/WG) streams %d barrier-every %2d grid %d: %.3f ms  %.2f TB/s\n", WAVES, DEPTH,
         (size_t)WAVES * DEPTH, STREAMS, BAR, grid, best, bytes / best / 1e9);
  if (hipGetLastError() != hipSuccess) { printf("error\n"); exit(1); }
}

int main() {
  const size_t bytes = (size_t)4 << 30;   // per stream
  char *a, *b; int* sink;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  hipMalloc(&sink, 4096 * sizeof(int));
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
  run_chunk<0, false>(a, b, bytes, sink);
  run_chunk<0, true>(a, b, bytes, sink);
  run_chunk<8, false>(a, b, bytes, sink);
  run_chunk<8, true>(a, b, bytes, sink);
  run_chunk<12, true>(a, b, bytes, sink);
  run<4, 2, 2, 0>(a, b, bytes, 256, sink);
  run<4, 4, 2, 0>(a, b, bytes, 256, sink);
  run<4, 8, 2, 0>(a, b, bytes, 256, sink);
  run<4, 16, 2, 0>(a, b, bytes, 256, sink);
  run<4, 16, 1, 0>(a, b, bytes, 256, sink);
  run<8, 8, 2, 0>(a, b, bytes, 256, sink);
  run<4, 16, 2, 16>(a, b, bytes, 256, sink);
  run<4, 24, 2, 16>(a, b, bytes, 256, sink);
  run<4, 16, 2, 0>(a, b, bytes, 512, sink);
  return 0;
}
