// Positive and negative controls of the race check: a kernel that hands a value from one thread
// to the next through LDS without a barrier must be reported, the same kernel with the barrier
// (or with a cross-lane operation inside one wave) must not.  Test infrastructure only.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void k_neighbour(int* out, int mode) {
  __shared__ int cell[128];
  const int t = threadIdx.x;
  cell[t] = t * 3;
  if (mode == 1) __syncthreads();
  if (mode == 2) __builtin_amdgcn_wave_barrier();
  // mode 2 reads inside the own wave only
  const int src = mode == 2 ? ((t & ~63) | ((t + 1) & 63)) : (t + 1) % 128;
  out[t] = cell[src];
}

// device memory: every workgroup adds to the same word, plainly (mode 0) or atomically (mode 1).
// Each does so many times: workgroups that follow each other on one worker thread of the
// simulation count as ordered, so the control needs several workers busy at the same time.
__global__ void k_counter(int* word, int mode) {
  if (threadIdx.x != 0) return;
  for (int k = 0; k < 100000; ++k) {
    if (mode == 0) *(volatile int*)word = *(volatile int*)word + 1;
    else atomicAdd(word, 1);
  }
}

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "";
  int* d;
  hipMalloc((void**)&d, 128 * sizeof(int));
  hipMemset(d, 0, 128 * sizeof(int));
  if (!strcmp(which, "lds-race")) hipLaunchKernelGGL(k_neighbour, 1, 128, 0, 0, d, 0);
  else if (!strcmp(which, "lds-barrier")) hipLaunchKernelGGL(k_neighbour, 1, 128, 0, 0, d, 1);
  else if (!strcmp(which, "lds-wave")) hipLaunchKernelGGL(k_neighbour, 1, 128, 0, 0, d, 2);
  else if (!strcmp(which, "global-race")) hipLaunchKernelGGL(k_counter, 64, 64, 0, 0, d, 0);
  else if (!strcmp(which, "global-atomic")) hipLaunchKernelGGL(k_counter, 64, 64, 0, 0, d, 1);
  else { fprintf(stderr, "usage: race_selftest lds-race|lds-barrier|lds-wave|global-race|global-atomic\n"); return 2; }
  int h[128];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%d\n", h[0]);
  hipFree(d);
  return 0;
}
