// Fusion kernels: float64 Euclidean cost matrix, exact DTW on
// a cost matrix (anti-diagonal wavefront) and the path-weighted row gather.
#include "avs_internal.h"
#include <math.h>

// ---------------------------------------------------------------------------
// cdist: out[i,j] = sqrt(sum_d (double(v[i,d]) - double(a[j,d]))^2)
// 32x32 outputs per 16x16-thread block, 2x2 per thread, operands staged in LDS.
// Written as the direct difference form in float64 (as SciPy does), so exact
// zeros stay exact; the output write (8 bytes per pair) is the HBM cost.
// ---------------------------------------------------------------------------
#define CD_T 32
#define CD_K 32
__global__ __launch_bounds__(256) void cdist_kernel(const float* __restrict__ v, int tv, const float* __restrict__ a,
                                                    int ta, int d, double* __restrict__ out) {
  __shared__ float sv[CD_T][CD_K + 1];
  __shared__ float sa[CD_T][CD_K + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * CD_T, j0 = blockIdx.x * CD_T;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int k0 = 0; k0 < d; k0 += CD_K) {
    for (int e = threadIdx.x; e < CD_T * CD_K; e += 256) {
      const int r = e / CD_K, k = e - r * CD_K;
      const int gi = i0 + r, gj = j0 + r, gk = k0 + k;
      sv[r][k] = (gi < tv && gk < d) ? v[(long long)gi * d + gk] : 0.f;
      sa[r][k] = (gj < ta && gk < d) ? a[(long long)gj * d + gk] : 0.f;
    }
    __syncthreads();
    const int kn = (d - k0) < CD_K ? (d - k0) : CD_K;
    for (int k = 0; k < kn; ++k) {
      const double v0 = (double)sv[ty][k], v1 = (double)sv[ty + 16][k];
      const double a0 = (double)sa[tx][k], a1 = (double)sa[tx + 16][k];
      double t;
      t = v0 - a0; acc[0][0] += t * t;
      t = v0 - a1; acc[0][1] += t * t;
      t = v1 - a0; acc[1][0] += t * t;
      t = v1 - a1; acc[1][1] += t * t;
    }
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int gi = i0 + ty + 16 * p, gj = j0 + tx + 16 * q;
      if (gi < tv && gj < ta) out[(long long)gi * ta + gj] = sqrt(acc[p][q]);
    }
}

extern "C" int avs_cdist_f64(const float* d_v, int tv, const float* d_a, int ta, int d, double* d_out,
                             avs_stream_t stream) {
  AVS_REQUIRE(tv >= 0 && ta >= 0 && d > 0, AVS_E_SHAPE, "avs_cdist_f64: tv=%d ta=%d d=%d", tv, ta, d);
  if (tv == 0 || ta == 0) return AVS_OK;
  AVS_REQUIRE(d_v && d_a && d_out, AVS_E_ARG, "avs_cdist_f64: null pointer");
  dim3 grid((unsigned)avs_cdiv(ta, CD_T), (unsigned)avs_cdiv(tv, CD_T));
  AVS_REQUIRE(grid.y <= 65535, AVS_E_SHAPE, "avs_cdist_f64: tv too large");
  hipLaunchKernelGGL(cdist_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_v, tv, d_a, ta, d, d_out);
  AVS_CHECK_LAUNCH("avs_cdist_f64");
  return AVS_OK;
}

// ---------------------------------------------------------------------------
// Exact DTW.  One 1024-thread workgroup sweeps the anti-diagonals; the three
// live diagonals stay in LDS (indexed by row i), only the 1-byte predecessor
// code of every cell goes to HBM.  A second one-thread kernel walks the codes
// back from (n-1, m-1).  Tie order: up (i-1,j), left (i,j-1), diagonal.
// ---------------------------------------------------------------------------
#define DTW_MAX_N 6400  // 3 * 6400 * 8 B = 150 KiB of the 160 KiB LDS
__global__ __launch_bounds__(1024) void dtw_sweep_kernel(const double* __restrict__ cost, int n, int m,
                                                         unsigned char* __restrict__ dir, double* __restrict__ total) {
  extern __shared__ double diag[];  // [3][n]
  double* d0 = diag;
  double* d1 = diag + n;
  double* d2 = diag + 2 * (long long)n;
  const double INF = INFINITY;
  for (int d = 0; d <= n + m - 2; ++d) {
    double* cur = d % 3 == 0 ? d0 : (d % 3 == 1 ? d1 : d2);
    const double* p1 = (d + 2) % 3 == 0 ? d0 : ((d + 2) % 3 == 1 ? d1 : d2);  // diagonal d-1
    const double* p2 = (d + 1) % 3 == 0 ? d0 : ((d + 1) % 3 == 1 ? d1 : d2);  // diagonal d-2
    const int ilo = d - (m - 1) > 0 ? d - (m - 1) : 0;
    const int ihi = d < n - 1 ? d : n - 1;
    for (int i = ilo + threadIdx.x; i <= ihi; i += blockDim.x) {
      const int j = d - i;
      const double up = i > 0 ? p1[i - 1] : INF;              // (i-1, j)   on diagonal d-1
      const double left = j > 0 ? p1[i] : INF;                // (i, j-1)   on diagonal d-1
      const double dg = (i > 0 && j > 0) ? p2[i - 1] : INF;   // (i-1, j-1) on diagonal d-2
      double best = up;
      unsigned char code = 0;
      if (left < best) { best = left; code = 1; }
      if (dg < best) { best = dg; code = 2; }
      if (i == 0 && j == 0) { best = 0.0; code = 3; }
      const double val = cost[(long long)i * m + j] + best;
      cur[i] = val;
      dir[(long long)i * m + j] = code;
      if (i == n - 1 && j == m - 1) *total = val;
    }
    __syncthreads();
  }
}

__global__ void dtw_backtrack_kernel(const unsigned char* __restrict__ dir, int n, int m, int64_t* __restrict__ path,
                                     int64_t* __restrict__ path_len) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  // first pass: length; second pass: fill from the end so the path runs start -> end
  int i = n - 1, j = m - 1;
  long long len = 1;
  while (i > 0 || j > 0) {
    const unsigned char c = dir[(long long)i * m + j];
    if (c == 0) --i; else if (c == 1) --j; else { --i; --j; }
    ++len;
  }
  *path_len = len;
  i = n - 1;
  j = m - 1;
  long long pos = len - 1;
  path[2 * pos] = i;
  path[2 * pos + 1] = j;
  while (i > 0 || j > 0) {
    const unsigned char c = dir[(long long)i * m + j];
    if (c == 0) --i; else if (c == 1) --j; else { --i; --j; }
    --pos;
    path[2 * pos] = i;
    path[2 * pos + 1] = j;
  }
}

extern "C" int64_t avs_dtw_workspace_bytes(int n, int m) {
  if (n <= 0 || m <= 0) return 0;
  return (((int64_t)n * m) + 255) & ~(int64_t)255;
}

extern "C" int avs_dtw_path_f64(const double* d_cost_matrix, int n, int m, void* d_workspace, int64_t workspace_bytes,
                                int64_t* d_path, int64_t* d_path_len, double* d_cost, avs_stream_t stream) {
  AVS_REQUIRE(n > 0 && m > 0, AVS_E_SHAPE, "avs_dtw_path_f64: n=%d m=%d", n, m);
  AVS_REQUIRE(n <= DTW_MAX_N, AVS_E_SHAPE, "avs_dtw_path_f64: n=%d exceeds the LDS-resident limit %d", n, DTW_MAX_N);
  AVS_REQUIRE(d_cost_matrix && d_workspace && d_path && d_path_len && d_cost, AVS_E_ARG,
              "avs_dtw_path_f64: null pointer");
  AVS_REQUIRE(workspace_bytes >= avs_dtw_workspace_bytes(n, m), AVS_E_WORKSPACE,
              "avs_dtw_path_f64: workspace %lld < %lld bytes", (long long)workspace_bytes,
              (long long)avs_dtw_workspace_bytes(n, m));
  const size_t shmem = (size_t)3 * n * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)dtw_sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       3 * DTW_MAX_N * (int)sizeof(double));
    if (e != hipSuccess) {
      avs_set_error("avs_dtw_path_f64: cannot raise dynamic LDS limit: %s", hipGetErrorString(e));
      return AVS_E_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(dtw_sweep_kernel, dim3(1), dim3(1024), shmem, (hipStream_t)stream, d_cost_matrix, n, m,
                     (unsigned char*)d_workspace, d_cost);
  AVS_CHECK_LAUNCH("avs_dtw_path_f64(sweep)");
  hipLaunchKernelGGL(dtw_backtrack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned char*)d_workspace, n, m, d_path, d_path_len);
  AVS_CHECK_LAUNCH("avs_dtw_path_f64(backtrack)");
  return AVS_OK;
}

// out[u,:] = x[idx[u],:] * float(w[u])
__global__ __launch_bounds__(256) void gather_scale_kernel(const float* __restrict__ x, long long ldx, int d,
                                                           const int64_t* __restrict__ idx,
                                                           const double* __restrict__ w, int count,
                                                           float* __restrict__ out) {
  const long long total = (long long)count * d;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long u = i / d;
    const int k = (int)(i - u * d);
    out[i] = x[idx[u] * ldx + k] * (float)w[u];
  }
}

extern "C" int avs_gather_scale_f32(const float* d_x, int64_t ldx, int d, const int64_t* d_idx, const double* d_w,
                                    int count, float* d_out, avs_stream_t stream) {
  AVS_REQUIRE(d > 0 && ldx >= d && count >= 0, AVS_E_SHAPE, "avs_gather_scale_f32: d=%d ldx=%lld count=%d", d,
              (long long)ldx, count);
  if (count == 0) return AVS_OK;
  AVS_REQUIRE(d_x && d_idx && d_w && d_out, AVS_E_ARG, "avs_gather_scale_f32: null pointer");
  long long gx = avs_cdiv((long long)count * d, 256);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(gather_scale_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, d_x, (long long)ldx, d,
                     d_idx, d_w, count, d_out);
  AVS_CHECK_LAUNCH("avs_gather_scale_f32");
  return AVS_OK;
}
