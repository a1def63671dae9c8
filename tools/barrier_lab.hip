// Micro-benchmark: cost of a device-wide barrier inside a kernel (atomic counter + agent-scope fences) on MI355X,
// against the ~5 us it costs to end one tiny kernel and start the next inside a hipGraph.  Decides whether multicolour
// sweeps of small levels should become one persistent kernel.   hipcc --offload-arch=gfx950 -O3 tools/barrier_lab.hip
// Every spin is bounded: a lost barrier ends the kernel with an error flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void barrier_kernel(int n_barriers, unsigned* counter, int* err, double* data, int work) {
  const unsigned nb = gridDim.x;
  double acc = 0.0;
  for (int k = 0; k < n_barriers; ++k) {
    // a little "work": every workgroup updates its own cache line and reads its neighbour's
    if (work) {
      const int me = blockIdx.x, nbr = (blockIdx.x + 1) % nb;
      if (threadIdx.x == 0) data[me * 16] = k + 1.0;
      acc += data[nbr * 16];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                     // release (agent scope: L2 write-back on multi-XCD parts)
      atomicAdd(counter, 1u);
      const unsigned target = (unsigned)(k + 1) * nb;
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000) { *err = 1; break; }
      }
      __threadfence();                                     // acquire
    }
    __syncthreads();
    if (*err) return;
  }
  if (acc < 0) data[0] = acc;
}

__global__ void tiny_kernel(double* data) { if (threadIdx.x == 0) data[blockIdx.x * 16] += 1.0; }

int main() {
  unsigned* counter; int* err; double* data;
  CHK(hipMalloc(&counter, 4)); CHK(hipMalloc(&err, 4)); CHK(hipMalloc(&data, 8 * 16 * 4096));
  CHK(hipMemset(data, 0, 8 * 16 * 4096));
  hipStream_t st; CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int N = 500;
  for (int work = 0; work < 2; ++work)
    for (int grid : {32, 64, 128, 256, 512, 1024}) {
      // co-residency: 256 CUs x (at least 4 workgroups of 256 threads) -> 1024 blocks are resident for sure
      CHK(hipMemsetAsync(counter, 0, 4, st)); CHK(hipMemsetAsync(err, 0, 4, st));
      hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, st, 10, counter, err, data, work);   // warm-up
      CHK(hipMemsetAsync(counter, 0, 4, st));
      CHK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), 0, st, N, counter, err, data, work);
      CHK(hipEventRecord(e1, st));
      CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      int herr = 0; CHK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      printf("grid %5d x 256, work %d: %7.2f us per barrier%s\n", grid, work, 1e3 * ms / N, herr ? "  (BARRIER TIMED OUT)" : "");
    }
  // reference: N dependent tiny kernels in a graph
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int k = 0; k < N; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(256), 0, st, data);
  CHK(hipStreamEndCapture(st, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
  CHK(hipEventRecord(e0, st)); CHK(hipGraphLaunch(ge, st)); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("graph of %d dependent tiny kernels: %7.2f us per kernel\n", N, 1e3 * ms / N);
  return 0;
}
