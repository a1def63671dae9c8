// spmv_lab -- stand-alone A/B harness for the level-0 SpMV kernel variants (not part of the product).
// Builds the 15-point Kuhn stencil pattern of an n^3 grid with random values in the SELL-64-pair layout of
// ngsamg_amd/csrc/device/kernels.hpp and times kernel variants interleaved in one process (median of
// rounds), as cdna_hip_programming.md rule 24 asks.  Usage: spmv_lab [nv=215] [reps=20] [rounds=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <random>
#include <vector>
#include <string>
#include <functional>
#include <array>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ---- V0: product kernel (RES epilogue) ----------------------------------------------------------------
template <int BLOCK, bool REMAP, bool NT, int UNROLL>
__global__ __launch_bounds__(BLOCK) void k_sell(int64_t n_rows, int n_slices, const int64_t* __restrict__ slice_ptr,
                                                const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                const double* __restrict__ x, const double* __restrict__ b, double* y) {
  const int lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int s = lb * (BLOCK / 64) + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)((slice_ptr[s + 1] - base) >> 6);
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
  const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base);
  const int np = w >> 1;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll UNROLL
  for (int p = 0; p < np; ++p) {
    double2 v; int2 c;
    if (NT) {
      v.x = __builtin_nontemporal_load(&v2[p * 64 + lane].x); v.y = __builtin_nontemporal_load(&v2[p * 64 + lane].y);
      c.x = __builtin_nontemporal_load(&c2[p * 64 + lane].x); c.y = __builtin_nontemporal_load(&c2[p * 64 + lane].y);
    } else { v = v2[p * 64 + lane]; c = c2[p * 64 + lane]; }
    acc0 += v.x * x[c.x];
    acc1 += v.y * x[c.y];
  }
  if (w & 1) {
    const int64_t o = base + (int64_t)(w - 1) * 64 + lane;
    acc0 += vals[o] * x[cols[o]];
  }
  const int64_t row = (int64_t)s * 64 + lane;
  if (row < n_rows) y[row] = b[row] - (acc0 + acc1);
}

// ---- persistent grid-stride version of the base kernel (no remap) ---------------------------------------
template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void k_sell_persist(int64_t n_rows, int n_slices, const int64_t* __restrict__ slice_ptr,
                                                        const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                        const double* __restrict__ x, const double* __restrict__ b, double* y) {
  const int lane = threadIdx.x & 63;
  const int wpb = BLOCK / 64;
  for (int s = blockIdx.x * wpb + (threadIdx.x >> 6); s < n_slices; s += gridDim.x * wpb) {
    const int64_t base = slice_ptr[s];
    const int w = (int)((slice_ptr[s + 1] - base) >> 6);
    const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
    const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base);
    const int np = w >> 1;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
    for (int p = 0; p < np; ++p) {
      double2 v; int2 c;
      if (NT) {
        v.x = __builtin_nontemporal_load(&v2[p * 64 + lane].x); v.y = __builtin_nontemporal_load(&v2[p * 64 + lane].y);
        c.x = __builtin_nontemporal_load(&c2[p * 64 + lane].x); c.y = __builtin_nontemporal_load(&c2[p * 64 + lane].y);
      } else { v = v2[p * 64 + lane]; c = c2[p * 64 + lane]; }
      acc0 += v.x * x[c.x];
      acc1 += v.y * x[c.y];
    }
    const int64_t row = (int64_t)s * 64 + lane;
    if (row < n_rows) y[row] = b[row] - (acc0 + acc1);
  }
}

// ---- stream-only ceiling: same loads, no gather -------------------------------------------------------
template <int BLOCK, bool REMAP>
__global__ __launch_bounds__(BLOCK) void k_stream(int64_t n_rows, int n_slices, const int64_t* __restrict__ slice_ptr,
                                                  const int32_t* __restrict__ cols, const double* __restrict__ vals,
                                                  const double* __restrict__ x, const double* __restrict__ b, double* y) {
  const int lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int s = lb * (BLOCK / 64) + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)((slice_ptr[s + 1] - base) >> 6);
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
  const int2* __restrict__ c2 = reinterpret_cast<const int2*>(cols + base);
  const int np = w >> 1;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
  for (int p = 0; p < np; ++p) {
    const double2 v = v2[p * 64 + lane];
    const int2 c = c2[p * 64 + lane];
    acc0 += v.x * (double)c.x;
    acc1 += v.y * (double)c.y;
  }
  const int64_t row = (int64_t)s * 64 + lane;
  if (row < n_rows) y[row] = b[row] - (acc0 + acc1);
}

// ---- 16-bit column deltas: per (slice, pair-step) base column (int32) + ushort2 per lane ---------------
// layout: cbase[slice_cb[s] + p] = min column over the 128 entries of pair-step p; cd16 has the same element
// offsets as vals (ushort per entry).
template <int BLOCK, int UNROLL, bool REMAP, bool NT>
__global__ __launch_bounds__(BLOCK) void k_sell16(int64_t n_rows, int n_slices, const int64_t* __restrict__ slice_ptr,
                                                  const int32_t* __restrict__ cbase, const uint16_t* __restrict__ cd16,
                                                  const double* __restrict__ vals, const double* __restrict__ x,
                                                  const double* __restrict__ b, double* y) {
  const int lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int s = lb * (BLOCK / 64) + (threadIdx.x >> 6);
  if (s >= n_slices) return;
  const int64_t base = slice_ptr[s];
  const int w = (int)((slice_ptr[s + 1] - base) >> 6);
  const double2* __restrict__ v2 = reinterpret_cast<const double2*>(vals + base);
  const ushort2* __restrict__ c2 = reinterpret_cast<const ushort2*>(cd16 + base);
  const int32_t* __restrict__ cb = cbase + (base >> 7);     // one base per 128 entries (w even here)
  const int np = w >> 1;
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll UNROLL
  for (int p = 0; p < np; ++p) {
    double2 v; ushort2 c;
    if (NT) {
      v.x = __builtin_nontemporal_load(&v2[p * 64 + lane].x); v.y = __builtin_nontemporal_load(&v2[p * 64 + lane].y);
      const unsigned u = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(&c2[p * 64 + lane]));
      c.x = (unsigned short)(u & 0xffff); c.y = (unsigned short)(u >> 16);
    } else { v = v2[p * 64 + lane]; c = c2[p * 64 + lane]; }
    const int cbp = cb[p];
    acc0 += v.x * x[cbp + c.x];
    acc1 += v.y * x[cbp + c.y];
  }
  const int64_t row = (int64_t)s * 64 + lane;
  if (row < n_rows) y[row] = b[row] - (acc0 + acc1);
}

// plain copy ceiling (double2 read + write)
__global__ __launch_bounds__(256) void k_copy(int64_t n2, const double2* __restrict__ a, double2* __restrict__ b) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n2; i += stride) b[i] = a[i];
}
// read-only ceiling
__global__ __launch_bounds__(256) void k_read(int64_t n2, const double2* __restrict__ a, double* out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  double s = 0;
  for (; i < n2; i += stride) { double2 v = a[i]; s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}

// read-only calibration kernels for the FETCH_SIZE counter: 8 B / lane and 4 B / lane (nt) streams
__global__ __launch_bounds__(256) void k_read8(int64_t n, const double* __restrict__ a, double* out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  double s = 0;
  for (; i < n; i += stride) s += a[i];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read4nt(int64_t n, const uint32_t* __restrict__ a, double* out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  uint32_t s = 0;
  for (; i < n; i += stride) s += __builtin_nontemporal_load(a + i);
  if (s == 123456u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read16nt(int64_t n2, const double* __restrict__ a, double* out) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  double s = 0;
  for (; i < n2; i += stride) s += __builtin_nontemporal_load(a + 2 * i) + __builtin_nontemporal_load(a + 2 * i + 1);
  if (s == 123.456) out[0] = s;
}

int main(int argc, char** argv) {
  const int nv = argc > 1 ? atoi(argv[1]) : 215;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  const int64_t n = (int64_t)nv * nv * nv;
  // 15-point pattern: monotone offsets and their negatives
  const int offs[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  std::vector<int64_t> deltas;
  std::vector<std::array<int, 3>> all;
  for (int sgn = -1; sgn <= 1; sgn += 2) for (auto& o : offs) all.push_back({sgn * o[0], sgn * o[1], sgn * o[2]});
  all.push_back({0, 0, 0});
  std::sort(all.begin(), all.end(), [&](auto& a, auto& b) { return (int64_t)a[0] * nv * nv + a[1] * nv + a[2] < (int64_t)b[0] * nv * nv + b[1] * nv + b[2]; });
  std::vector<int64_t> rowptr(n + 1, 0);
  for (int64_t v = 0; v < n; ++v) {
    int i = v / ((int64_t)nv * nv), j = (v / nv) % nv, k = v % nv, c = 0;
    for (auto& o : all) { int a = i + o[0], b = j + o[1], d = k + o[2]; c += (a >= 0 && a < nv && b >= 0 && b < nv && d >= 0 && d < nv); }
    rowptr[v + 1] = rowptr[v] + c;
  }
  const int64_t nnz = rowptr[n];
  const int64_t ns = (n + 63) / 64;
  std::vector<int64_t> sp(ns + 1, 0);
  for (int64_t s = 0; s < ns; ++s) {
    int w = 0;
    for (int64_t r = s * 64; r < std::min(n, (s + 1) * 64); ++r) w = std::max<int>(w, (int)(rowptr[r + 1] - rowptr[r]));
    w = (w + 1) & ~1;    // lab: even widths only (keeps the 16-bit variant simple); pads <= 1 column
    sp[s + 1] = sp[s] + (int64_t)w * 64;
  }
  const int64_t stored = sp[ns];
  printf("nv=%d n=%lld nnz=%lld stored=%lld (pad %.3f)\n", nv, (long long)n, (long long)nnz, (long long)stored, (double)stored / nnz);
  std::vector<int32_t> sc(stored, 0);
  std::vector<double> sv(stored, 0.0);
  std::mt19937_64 rng(1);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int64_t s = 0; s < ns; ++s) {
    const int64_t base = sp[s];
    const int w = (int)((sp[s + 1] - base) / 64);
    for (int l = 0; l < 64; ++l) {
      const int64_t v = s * 64 + l;
      int cnt = 0;
      int32_t first = 0;
      if (v < n) {
        int i = v / ((int64_t)nv * nv), j = (v / nv) % nv, k = v % nv;
        for (auto& o : all) {
          int a = i + o[0], b = j + o[1], d = k + o[2];
          if (a >= 0 && a < nv && b >= 0 && b < nv && d >= 0 && d < nv) {
            const int64_t o2 = base + (int64_t)(cnt >> 1) * 128 + l * 2 + (cnt & 1);
            sc[o2] = (int32_t)(((int64_t)a * nv + b) * nv + d);
            sv[o2] = U(rng);
            if (cnt == 0) first = sc[o2];
            cnt++;
          }
        }
      }
      for (int q = cnt; q < w; ++q) { const int64_t o2 = base + (int64_t)(q >> 1) * 128 + l * 2 + (q & 1); sc[o2] = v < n ? first : 0; sv[o2] = 0.0; }
    }
  }
  // 16-bit deltas
  std::vector<int32_t> cbase(stored / 128);
  std::vector<uint16_t> cd16(stored);
  int64_t overflow = 0;
  for (int64_t g = 0; g < stored / 128; ++g) {
    int32_t mn = sc[g * 128];
    for (int q = 1; q < 128; ++q) mn = std::min(mn, sc[g * 128 + q]);
    cbase[g] = mn;
    for (int q = 0; q < 128; ++q) { int64_t d = sc[g * 128 + q] - mn; if (d > 65535) { overflow++; d = 0; } cd16[g * 128 + q] = (uint16_t)d; }
  }
  printf("16-bit delta overflow entries: %lld\n", (long long)overflow);

  int64_t* d_sp; int32_t* d_sc; double* d_sv; int32_t* d_cb; uint16_t* d_c16;
  CK(hipMalloc(&d_sp, (ns + 1) * 8)); CK(hipMalloc(&d_sc, stored * 4)); CK(hipMalloc(&d_sv, stored * 8));
  CK(hipMalloc(&d_cb, cbase.size() * 4)); CK(hipMalloc(&d_c16, stored * 2));
  CK(hipMemcpy(d_sp, sp.data(), (ns + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sc, sc.data(), stored * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sv, sv.data(), stored * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cb, cbase.data(), cbase.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_c16, cd16.data(), stored * 2, hipMemcpyHostToDevice));
  const int NSET = 4;
  double *d_x[NSET], *d_b[NSET], *d_y[NSET];
  std::vector<double> hx(n);
  for (auto& t : hx) t = U(rng);
  for (int q = 0; q < NSET; ++q) {
    CK(hipMalloc(&d_x[q], n * 8)); CK(hipMalloc(&d_b[q], n * 8)); CK(hipMalloc(&d_y[q], n * 8));
    CK(hipMemcpy(d_x[q], hx.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_b[q], hx.data(), n * 8, hipMemcpyHostToDevice));
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  struct Var { std::string name; std::function<void(int)> launch; double bytes; std::vector<double> t; };
  const double alg_bytes = (double)nnz * 12 + 4.0 * (n + 1) + 3.0 * 8 * n;
  std::vector<Var> vars;
  auto grid = [&](int block) { return (int)((ns + block / 64 - 1) / (block / 64)); };
#define ADD(NAME, KERNEL, BLOCK) vars.push_back({NAME, [&](int q) { hipLaunchKernelGGL((KERNEL), dim3(grid(BLOCK)), dim3(BLOCK), 0, st, n, (int)ns, d_sp, d_sc, d_sv, d_x[q], d_b[q], d_y[q]); }, alg_bytes, {}})
  ADD("base b256 remap u4", (k_sell<256, true, false, 4>), 256);
  ADD("no-remap", (k_sell<256, false, false, 4>), 256);
  ADD("no-remap nt", (k_sell<256, false, true, 4>), 256);
  ADD("no-remap b512", (k_sell<512, false, false, 4>), 512);
  ADD("no-remap b64", (k_sell<64, false, false, 4>), 64);
  ADD("stream-only remap", (k_stream<256, true>), 256);
  ADD("stream-only no-remap", (k_stream<256, false>), 256);
#undef ADD
#define ADD16(NAME, U, RM, NTL) vars.push_back({NAME, [&](int q) { hipLaunchKernelGGL((k_sell16<256, U, RM, NTL>), dim3(grid(256)), dim3(256), 0, st, n, (int)ns, d_sp, d_cb, d_c16, d_sv, d_x[q], d_b[q], d_y[q]); }, alg_bytes, {}})
  ADD16("16-bit remap", 4, true, false);
  ADD16("16-bit no-remap", 4, false, false);
  ADD16("16-bit no-remap nt", 4, false, true);
#undef ADD16
  vars.push_back({"persist 2048 blocks", [&](int q) { hipLaunchKernelGGL((k_sell_persist<256, false>), dim3(2048), dim3(256), 0, st, n, (int)ns, d_sp, d_sc, d_sv, d_x[q], d_b[q], d_y[q]); }, alg_bytes, {}});
  vars.push_back({"persist 2048 blocks nt", [&](int q) { hipLaunchKernelGGL((k_sell_persist<256, true>), dim3(2048), dim3(256), 0, st, n, (int)ns, d_sp, d_sc, d_sv, d_x[q], d_b[q], d_y[q]); }, alg_bytes, {}});
  vars.push_back({"persist 1024 blocks", [&](int q) { hipLaunchKernelGGL((k_sell_persist<256, false>), dim3(1024), dim3(256), 0, st, n, (int)ns, d_sp, d_sc, d_sv, d_x[q], d_b[q], d_y[q]); }, alg_bytes, {}});
  vars.push_back({"read-only 1.2GB double2", [&](int q) { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, st, stored / 2, (const double2*)d_sv, d_y[q]); }, (double)stored * 8, {}});
  vars.push_back({"calib read 8B/lane 1.27GB", [&](int q) { hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, st, stored, (const double*)d_sv, d_y[q]); }, (double)stored * 8, {}});
  vars.push_back({"calib read 4B/lane nt 0.63GB", [&](int q) { hipLaunchKernelGGL(k_read4nt, dim3(4096), dim3(256), 0, st, stored, (const uint32_t*)d_sc, d_y[q]); }, (double)stored * 4, {}});
  vars.push_back({"calib read 16B/lane nt 1.27GB", [&](int q) { hipLaunchKernelGGL(k_read16nt, dim3(4096), dim3(256), 0, st, stored / 2, (const double*)d_sv, d_y[q]); }, (double)stored * 8, {}});
  vars.push_back({"read-only 8192 blocks", [&](int q) { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, st, stored / 2, (const double2*)d_sv, d_y[q]); }, (double)stored * 8, {}});

  for (int mode = 0; mode < 2; ++mode) {       // 0: same vector set every launch (warm x/b), 1: rotate 4 sets (cold)
    for (auto& v : vars) v.t.clear();
    for (int r = 0; r < rounds; ++r)
      for (auto& v : vars) {
        v.launch(0);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) v.launch(mode ? i % NSET : 0);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        v.t.push_back(ms / reps);
      }
    printf("---- vectors %s ----\n", mode ? "rotating (cold)" : "same set (warm in MALL)");
    for (auto& v : vars) {
      std::sort(v.t.begin(), v.t.end());
      const double med = v.t[v.t.size() / 2], mn = v.t[0];
      printf("%-28s median %8.1f us  min %8.1f us   %7.1f GB/s (alg. bytes, median)\n", v.name.c_str(), med * 1e3, mn * 1e3, v.bytes / med / 1e6);
    }
  }
  // correctness of the 16-bit variant vs base
  std::vector<double> y0(n), y1(n);
  vars[0].launch(0); CK(hipStreamSynchronize(st)); CK(hipMemcpy(y0.data(), d_y[0], n * 8, hipMemcpyDeviceToHost));
  vars[8].launch(0); CK(hipStreamSynchronize(st)); CK(hipMemcpy(y1.data(), d_y[0], n * 8, hipMemcpyDeviceToHost));
  double md = 0; for (int64_t i = 0; i < n; ++i) md = std::max(md, std::fabs(y0[i] - y1[i]));
  printf("max |base - 16bit| = %g\n", md);
  return 0;
}
