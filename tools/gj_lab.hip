// Stand-alone check and timing of the device-side dense SPD inverse (ngsamg_amd/csrc/device/dense_spd.hpp).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ngsamg_amd/csrc/device -o tools/bin/gj_lab tools/gj_lab.hip
//   run:   tools/bin/gj_lab [n ...]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include <chrono>
namespace amgx {
struct Err : std::runtime_error { using std::runtime_error::runtime_error; };
constexpr int BLOCK = 256;
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw ::amgx::Err(std::string(#call) + " failed: " + hipGetErrorString(e_)); } while (0)
template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  void alloc(size_t c) { n = c; HIPCHK(hipMalloc((void**)&p, c * sizeof(T))); }
};
}
#include "dense_spd.hpp"

int main(int argc, char** argv) {
  std::vector<int> sizes;
  for (int i = 1; i < argc; ++i) sizes.push_back(atoi(argv[i]));
  if (sizes.empty()) sizes = {64, 192, 1920, 4096};
  try {
    hipStream_t st;
    HIPCHK(hipStreamCreate(&st));
    for (int n : sizes) {
      const int64_t N = n;
      std::vector<double> A((size_t)N * N), B;
      uint64_t s = 777;
      auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
      // SPD, banded-ish, diagonally dominant enough: A = L + L^T + c I
      for (int64_t i = 0; i < N; ++i) for (int64_t j = 0; j < N; ++j) A[i * N + j] = 0.0;
      for (int64_t i = 0; i < N; ++i) {
        double rs = 0;
        for (int64_t j = std::max<int64_t>(0, i - 40); j < i; ++j) { const double v = rnd(); A[i * N + j] = v; A[j * N + i] = v; }
        (void)rs;
      }
      for (int64_t i = 0; i < N; ++i) { double rs = 0; for (int64_t j = 0; j < N; ++j) if (j != i) rs += std::fabs(A[i * N + j]); A[i * N + i] = rs * 1.05 + 0.1; }
      double* D;
      HIPCHK(hipMalloc(&D, (size_t)N * N * 8));
      HIPCHK(hipMemcpy(D, A.data(), (size_t)N * N * 8, hipMemcpyHostToDevice));
      auto t0 = std::chrono::steady_clock::now();
      const double piv = amgx::dense_spd_inverse(D, N, N, st);
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      B.resize((size_t)N * N);
      HIPCHK(hipMemcpy(B.data(), D, (size_t)N * N * 8, hipMemcpyDeviceToHost));
      // check a sample of entries of A * B against I
      double worst = 0;
      for (int q = 0; q < 2000; ++q) {
        const int64_t i = (int64_t)((rnd() + 0.5) * N) % N, j = (q & 1) ? i : (int64_t)((rnd() + 0.5) * N) % N;
        double acc = 0;
        for (int64_t k = 0; k < N; ++k) acc += A[i * N + k] * B[k * N + j];
        worst = std::fmax(worst, std::fabs(acc - (i == j ? 1.0 : 0.0)));
      }
      printf("n = %5d: %8.2f ms, %.2f TFLOP/s (2 n^3), min pivot ratio %.3e, max |A inv(A) - I| (sample) = %.3e  %s\n", n, ms, 2.0 * N * N * N / ms / 1e9, piv, worst,
             worst < 1e-9 ? "ok" : "MISMATCH");
      (void)hipFree(D);
    }
  } catch (const std::exception& e) { printf("ERROR: %s\n", e.what()); return 1; }
  return 0;
}
