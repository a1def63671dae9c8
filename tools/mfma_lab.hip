// MFMA A/B for the 6x6 block-row product ("one MFMA A/B, recorded").
//
// y = A x, A = n block rows of W = 15 blocks (6x6 fp64, row-major, 3D-stencil-like band), one right-hand side.
//   variant VALU : lane = scalar row of a block row (10 block rows per wave), 6 FMAs per block and lane
//                  (the arithmetic of bcsr_rowlane_kernel / bsell_spmv_kernel)
//   variant MFMA : v_mfma_f64_16x16x4_f64.  A wave multiplies TWO block rows at a time: M = 16 rows = 6 + 6 scalar rows
//                  (+ 4 idle), K = 4 columns of the current blocks (a 6-wide block takes two K steps, the second half
//                  padded), N = 16 columns of which TWO are used: column 0 carries the x chunk of block row 0, column 1
//                  that of block row 1 (each block row multiplies its OWN x entries -- there is no shared operand, which
//                  is the point: SpMV with one right-hand side is not a contraction).  Useful MACs per instruction:
//                  2 * 6 * 4 (3 on the padded step) of 1024.
// Both variants are checked against each other.  build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_lab tools/mfma_lab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int BS = 6, W = 15, BB = BS * BS;
typedef double double4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void valu_kernel(int64_t n, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rb = lane / BS, r = lane % BS;
  const int64_t i = wave * 10 + rb;
  if (rb >= 10 || i >= n) return;
  double acc = 0.0;
#pragma unroll 5
  for (int b = 0; b < W; ++b) {
    const double* a = val + ((i * W + b) * BB + r * BS);
    const double* xv = x + (int64_t)col[i * W + b] * BS;
#pragma unroll
    for (int c = 0; c < BS; ++c) acc += a[c] * xv[c];
  }
  y[i * BS + r] = acc;
}

__global__ __launch_bounds__(256) void mfma_kernel(int64_t n, const int32_t* __restrict__ col, const double* __restrict__ val,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t i0 = wave * 2, i1 = i0 + 1;
  if (i0 >= n) return;
  const int m = lane & 15, k = lane >> 4;                 // A operand: A[m][k];  B operand: B[k][nn], nn = lane & 15
  const int which = m < 6 ? 0 : (m < 12 ? 1 : 2);         // block row of this A lane (2: idle rows)
  const int64_t irow = which == 0 ? i0 : i1;
  const int rr = which == 0 ? m : m - 6;
  const bool a_on = which < 2 && irow < n;
  const int nn = lane & 15;
  const int64_t brow = nn == 0 ? i0 : i1;
  const bool b_on = nn < 2 && brow < n;
  double4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 3
  for (int b = 0; b < W; ++b) {
    const double* ablk = val + ((irow * W + b) * BB + rr * BS);
    const double* xv = x + (int64_t)(b_on ? col[brow * W + b] : 0) * BS;
    // K step 0: block columns 0..3, K step 1: block columns 4, 5 (+ 2 padded)
    const double a0 = a_on ? ablk[k] : 0.0;
    const double a1 = (a_on && k < 2) ? ablk[4 + k] : 0.0;
    const double b0 = b_on ? xv[k] : 0.0;
    const double b1 = (b_on && k < 2) ? xv[4 + k] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
  }
  // D[row][col]: col = lane & 15, row = (lane >> 4) + 4 * reg.  Column 0 = block row i0 (rows 0..5), column 1 = i1 (rows 6..11)
  if (nn == 0) {
    const int q = lane >> 4;
    y[i0 * BS + q] = acc[0];                               // rows 0..3
    if (q < 2) y[i0 * BS + 4 + q] = acc[1];                // rows 4, 5
  } else if (nn == 1 && i1 < n) {
    const int q = lane >> 4;
    if (q >= 2) y[i1 * BS + (q - 2)] = acc[1];             // rows 6, 7   -> 0, 1
    y[i1 * BS + 2 + q] = acc[2];                           // rows 8..11  -> 2..5
  }
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
  const int offs[W] = {-1025, -1024, -1023, -33, -32, -31, -1, 0, 1, 31, 32, 33, 1023, 1024, 1025};
  std::vector<int32_t> col((size_t)n * W);
  std::vector<double> val((size_t)n * W * BB), x((size_t)n * BS);
  uint64_t s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  for (int64_t i = 0; i < n; ++i)
    for (int b = 0; b < W; ++b) { int64_t c = i + offs[b]; c = c < 0 ? 0 : (c >= n ? n - 1 : c); col[i * W + b] = (int32_t)c; }
  for (auto& v : val) v = rnd();
  for (auto& v : x) v = rnd();
  int32_t* dcol; double *dval, *dx, *dy0, *dy1;
  CK(hipMalloc(&dcol, col.size() * 4)); CK(hipMalloc(&dval, val.size() * 8)); CK(hipMalloc(&dx, x.size() * 8));
  CK(hipMalloc(&dy0, x.size() * 8)); CK(hipMalloc(&dy1, x.size() * 8));
  CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dval, val.data(), val.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice));
  const double bytes = (double)n * W * (BB * 8 + 4) + 2.0 * n * BS * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) -> int {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("%-44s %8.1f us  %7.1f GB/s (matrix + vectors = %.2f GB)\n", name, ms * 1e3, bytes / ms / 1e6, bytes / 1e9);
    return 0;
  };
  const int g_valu = (int)(((n + 9) / 10 + 3) / 4), g_mfma = (int)(((n + 1) / 2 + 3) / 4);
  time("VALU  (lane = scalar row, 6 FMA per block)", [&] { valu_kernel<<<g_valu, 256>>>(n, dcol, dval, dx, dy0); });
  time("MFMA  (v_mfma_f64_16x16x4, 2 block rows/wave)", [&] { mfma_kernel<<<g_mfma, 256>>>(n, dcol, dval, dx, dy1); });
  std::vector<double> y0(x.size()), y1(x.size());
  CK(hipMemcpy(y0.data(), dy0, y0.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(y1.data(), dy1, y1.size() * 8, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t q = 0; q < y0.size(); ++q) { md = std::fmax(md, std::fabs(y0[q] - y1[q])); mx = std::fmax(mx, std::fabs(y0[q])); }
  printf("max |y_valu - y_mfma| = %.3e (max |y| = %.3e)  => %s\n", md, mx, md <= 1e-12 * mx ? "results agree" : "MISMATCH");
  return md <= 1e-12 * mx ? 0 : 2;
}
