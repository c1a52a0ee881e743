// wgrad.hip -- weight-gradient GEMM of the MLP on fp32 MFMA (gfx950), plus the small reductions around it.
//
//   dW[i][j] = sum_m  A[i][m] * B[j][m]          A = dY^T (gradient rows), B = X^T (input feature rows)
// Both operands are stored tile-major [Mpad/32][rows][32] by the forward/backward kernels (common.h), i.e.
// K(=sample)-contiguous for BOTH MFMA operands ("NT" GEMM with a huge reduction dimension and a <=256x256
// output) and one K-chunk of an operand is a single contiguous rows*128-byte block of HBM.  At the fp32 MFMA
// peak this kernel has to stream 64 flop/byte = 2.45 TB/s of operands, so the access pattern matters.
// One workgroup (4 waves, 1 per SIMD, 512-register file) owns a whole <=256x256 output tile in
// accumulators (each wave 128x128 = 4x4 MFMA tiles = 256 registers) and walks a slice of the sample axis
// in chunks of 32 samples staged by LDS-DMA into an XOR-swizzled image; slices of the sample axis are
// combined with fp32 atomics into the (pre-zeroed) canonical gradient buffer: per 256x256 layer that is
// (#slices) x 256 KiB of atomic traffic, two to three orders of magnitude below the operand stream.
// The host cuts every (layer, network) into work entries of at most `wg_chunks` chunks so all workgroups carry
// about the same work (the coarse network has 1/3 of the fine one's samples) and launches the longest first.
// The k order inside an 8-sample group is permuted (lane half h takes samples 4h..4h+3) identically for A
// and B, so one 16-byte LDS read feeds four MFMAs.
#include "mlp_common.cuh"
#include "launchers.h"

namespace nh {


constexpr int kWgRows = 256;                 // max rows per operand tile
constexpr int kWgChunkFloats = kWgRows * 32; // one operand, one chunk: 256 rows x 32 samples

// LDS image of one operand chunk: [row][8 x 16-byte slots], slot s of row r holds samples
// 4*(s ^ swz(r)) .. +3 with swz(r) = (r >> 1) & 7  (conflict-free ds_read_b128 across 16 rows)
__device__ __forceinline__ void stage_operand(const float* g, int rows, float* lds, int wave, int lane) {
  // wave-instruction u covers rows 8u..8u+7 (64 lanes x 16 B = 8 rows x 128 B)
  const int nrow8 = (rows + 7) >> 3;
  for (int u = wave; u < nrow8; u += 4) {
    int r = 8 * u + (lane >> 3);
    const int s = lane & 7;
    const int rs = r < rows ? r : rows - 1;  // clamp: rows past the end duplicate the last row (never stored)
    const int src_slot = s ^ ((r >> 1) & 7);
    __builtin_amdgcn_global_load_lds(NERFHIP_GPTR(g + rs * 32 + 4 * src_slot),
                                     NERFHIP_LPTR(lds + u * 256), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256, 1) void wgrad_f32_kernel(const WgradItem* __restrict__ items) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 buffers x (A chunk | B chunk)
  const WgradItem it = items[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int c0 = it.c0, c1 = it.c1;   // this work entry's slice of the sample axis, in 32-sample chunks
  if (c0 >= c1) return;
  const int wa = wave >> 1, wb = wave & 1;          // this wave: A rows [128 wa, +128) x B rows [128 wb, +128)
  const int a_tiles = min(4, max(0, (it.a_rows - 128 * wa + 31) / 32));
  const int b_tiles = min(4, max(0, (it.b_rows - 128 * wb + 31) / 32));

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient = row sums of A: the waves with wb == 0 add up the A fragments they hold anyway (lane half h
  // has 4 of every 8 samples of row 32i + li; the halves are combined at the end) -- no extra LDS traffic
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = it.db != nullptr && wb == 0;

  // buffer b: A chunk at smem + 2b*chunk, B chunk at smem + (2b+1)*chunk
  const int64_t a_step = (int64_t)it.a_stride * 32, b_step = (int64_t)it.b_stride * 32;   // floats per chunk
  stage_operand(it.A + c0 * a_step, it.a_rows, smem, wave, lane);
  stage_operand(it.B + c0 * b_step, it.b_rows, smem + kWgChunkFloats, wave, lane);
  for (int c = c0; c < c1; ++c) {
    const int cur = (c - c0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < c1) {
      float* nb = smem + (cur ^ 1) * 2 * kWgChunkFloats;
      stage_operand(it.A + (c + 1) * a_step, it.a_rows, nb, wave, lane);
      stage_operand(it.B + (c + 1) * b_step, it.b_rows, nb + kWgChunkFloats, wave, lane);
    }
    const float* As = smem + cur * 2 * kWgChunkFloats;
    const float* Bs = As + kWgChunkFloats;
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {      // 8-sample group: lane half h takes samples 8 g8 + 4h .. +3
      const int slot = 2 * g8 + h;
      f32x4 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = 128 * wa + 32 * i + li;
        af[i] = *reinterpret_cast<const f32x4*>(As + ra * 32 + 4 * (slot ^ ((ra >> 1) & 7)));
        const int rb = 128 * wb + 32 * i + li;
        bf[i] = *reinterpret_cast<const f32x4*>(Bs + rb * 32 + 4 * (slot ^ ((rb >> 1) & 7)));
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[i] += (af[i][0] + af[i][1]) + (af[i][2] + af[i][3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < a_tiles) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < b_tiles) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][2], bf[j][2], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][3], bf[j][3], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  // combine the sample-axis slices: fp32 atomics, 128 contiguous bytes per lane half per instruction
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < a_tiles) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < b_tiles) {
          const int col = 128 * wb + 32 * j + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = 128 * wa + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < it.a_rows && col < it.b_rows)
              atomicAdd(it.dW + (int64_t)row * it.ldo + it.col0 + col, acc[i][j][r]);
          }
        }
      }
    }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = bs[i] + __shfl_xor(bs[i], 32);
      const int row = 128 * wa + 32 * i + li;
      if (h == 0 && row < it.a_rows) atomicAdd(it.db + row, v);
    }
  }
}

// d(dirvec)^T(j, ray) = sum over the ray's samples of dh(j, ray*S + s)   (one wave per (row, ray)); both
// buffers tile-major (common.h)
__global__ __launch_bounds__(256) void ray_sum_kernel(const float* __restrict__ dh, int dh_rows, int rows, int64_t n,
                                                      int S, float* __restrict__ out, int out_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (int64_t)rows * n) return;
  const int j = (int)(wid / n);
  const int64_t ray = wid - (int64_t)j * n;
  float s = 0.f;
  for (int i = lane; i < S; i += 64) s += dh[tiled_index(j, ray * S + i, dh_rows)];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
  if (lane == 0) out[tiled_index(j, ray, out_rows)] = s;
}

hipError_t launch_wgrad_f32(const WgradItem* d_items, int nitems, hipStream_t s) {
  if (nitems <= 0) return hipSuccess;
  const size_t lds = (size_t)4 * kWgChunkFloats * sizeof(float);  // 128 KiB
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wgrad_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(nitems), dim3(256), lds, s, d_items);
  return hipGetLastError();
}

hipError_t launch_ray_sum(const float* dh, int dh_rows, int rows, int64_t n, int S, float* out, int out_rows,
                          hipStream_t s) {
  const int64_t waves = (int64_t)rows * n;
  hipLaunchKernelGGL(ray_sum_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, dh, dh_rows, rows, n, S, out,
                     out_rows);
  return hipGetLastError();
}

}  // namespace nh
