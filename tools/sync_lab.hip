// Cross-stream dependency price on MI355X: ping-pong of tiny kernels between two streams with (a) event record/wait,
// (b) hipStreamWriteValue64 / hipStreamWaitValue64, against the same kernels on one stream.
// Also: cost of an event record on a stream that simply continues (no waiter blocks it).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/sync_lab tools/sync_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(double* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void mid(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void check(const double* p, double expect_min, int* bad) { if (p[threadIdx.x] < expect_min - 0.5) atomicAdd(bad, 1); }
int main() {
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  double* d; CK(hipMalloc(&d, 64 << 20)); CK(hipMemset(d, 0, 64 << 20));
  uint64_t* flag; CK(hipMalloc(&flag, 256)); CK(hipMemset(flag, 0, 256));
  const int N = 2000;
  auto run = [&](const char* name, auto body) -> int {
    for (int w = 0; w < 2; ++w) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int i = 0; i < N; ++i) body(i);
      CK(hipDeviceSynchronize());
      double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count() / N;
      if (w) printf("%-70s %8.2f us / iteration\n", name, us);
    }
    return 0;
  };
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  run("2 tiny kernels, one stream", [&](int) { tiny<<<1, 64, 0, s1>>>(d); tiny<<<1, 64, 0, s1>>>(d + 8); });
  run("2 tiny kernels, one stream, event record after each (nobody waits)", [&](int) { tiny<<<1, 64, 0, s1>>>(d); hipEventRecord(e1, s1); tiny<<<1, 64, 0, s1>>>(d + 8); hipEventRecord(e2, s1); });
  run("ping-pong s1 -> s2 -> s1 with events", [&](int) {
    tiny<<<1, 64, 0, s1>>>(d); hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0);
    tiny<<<1, 64, 0, s2>>>(d + 8); hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); });
  uint64_t v = 0;
  run("ping-pong s1 -> s2 -> s1 with stream write/wait value", [&](int) {
    tiny<<<1, 64, 0, s1>>>(d); ++v; hipStreamWriteValue64(s1, flag, v, 0); hipStreamWaitValue64(s2, flag, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
    tiny<<<1, 64, 0, s2>>>(d + 8); ++v; hipStreamWriteValue64(s2, flag + 8, v, 0); hipStreamWaitValue64(s1, flag + 8, v, hipStreamWaitValueGte, 0xffffffffffffffffull); });
  // the shape of one halo exchange: compute stream runs a mid-size kernel while the side stream does two tiny ones
  const int n = 4 << 20;
  run("mid kernel alone (one stream) + tiny", [&](int) { mid<<<n / 256, 256, 0, s1>>>(d, n); tiny<<<1, 64, 0, s1>>>(d); });
  run("exchange shape with events: rec, [s2: wait, tiny, tiny, rec], mid, wait, tiny", [&](int) {
    hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0); tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); hipEventRecord(e2, s2);
    mid<<<n / 256, 256, 0, s1>>>(d, n); hipStreamWaitEvent(s1, e2, 0); tiny<<<1, 64, 0, s1>>>(d); });
  run("exchange shape with write/wait value", [&](int) {
    ++v; hipStreamWriteValue64(s1, flag, v, 0); hipStreamWaitValue64(s2, flag, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
    tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); ++v; hipStreamWriteValue64(s2, flag + 8, v, 0);
    mid<<<n / 256, 256, 0, s1>>>(d, n); hipStreamWaitValue64(s1, flag + 8, v, hipStreamWaitValueGte, 0xffffffffffffffffull); tiny<<<1, 64, 0, s1>>>(d); });
  run("exchange shape, side work launched with NO dependency at all (lower bound)", [&](int) {
    tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); mid<<<n / 256, 256, 0, s1>>>(d, n); tiny<<<1, 64, 0, s1>>>(d); });
  // GPU-bound versions (the host runs far ahead): big kernel (~100+ us) on s1, dependent side chain on s2
  double* big; CK(hipMalloc(&big, (size_t)1 << 30)); CK(hipMemset(big, 0, (size_t)1 << 30));
  const int nb = 96 << 20;
  int* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
  run("GPU-bound: big, tiny, tiny, tiny on one stream", [&](int) { mid<<<nb / 256, 256, 0, s1>>>(big, nb); tiny<<<1, 64, 0, s1>>>(d); tiny<<<1, 64, 0, s1>>>(d + 8); tiny<<<1, 64, 0, s1>>>(d + 16); });
  run("GPU-bound: exchange shape with events", [&](int) {
    hipEventRecord(e1, s1); hipStreamWaitEvent(s2, e1, 0); tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); hipEventRecord(e2, s2);
    mid<<<nb / 256, 256, 0, s1>>>(big, nb); hipStreamWaitEvent(s1, e2, 0); tiny<<<1, 64, 0, s1>>>(d); });
  run("GPU-bound: exchange shape with write/wait value", [&](int) {
    ++v; hipStreamWriteValue64(s1, flag, v, 0); hipStreamWaitValue64(s2, flag, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
    tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); ++v; hipStreamWriteValue64(s2, flag + 8, v, 0);
    mid<<<nb / 256, 256, 0, s1>>>(big, nb); hipStreamWaitValue64(s1, flag + 8, v, hipStreamWaitValueGte, 0xffffffffffffffffull); tiny<<<1, 64, 0, s1>>>(d); });
  run("GPU-bound: exchange shape, no dependency", [&](int) {
    tiny<<<1, 64, 0, s2>>>(d + 8); tiny<<<1, 64, 0, s2>>>(d + 16); mid<<<nb / 256, 256, 0, s1>>>(big, nb); tiny<<<1, 64, 0, s1>>>(d); });
  // correctness of write/wait value on plain hipMalloc memory: the consumer on s2 must see what the big kernel on s1 wrote
  CK(hipMemset(big, 0, (size_t)1 << 30)); CK(hipDeviceSynchronize());
  for (int i = 0; i < 50; ++i) {
    mid<<<nb / 256, 256, 0, s1>>>(big, nb);                     // every entry += ~1
    ++v; hipStreamWriteValue64(s1, flag, v, 0); hipStreamWaitValue64(s2, flag, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
    check<<<1, 64, 0, s2>>>(big + nb - 64, (double)(i + 1), bad);
    ++v; hipStreamWriteValue64(s2, flag + 8, v, 0); hipStreamWaitValue64(s1, flag + 8, v, hipStreamWaitValueGte, 0xffffffffffffffffull);
  }
  CK(hipDeviceSynchronize());
  int hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  printf("write/wait value ordering check: %d violations in 50 rounds\n", hb);
  return 0;
}
