// Diagnostic microbenchmark (not part of the product): write bandwidth of the GRU stash pattern.  512 workgroups each
// write one 16 KB block per "step" for 240 steps: (a) tile-major — every workgroup owns a contiguous 3.9 MB stream
// (the current stash layout), (b) step-major — the 512 blocks of a step are adjacent (8 MB per step for the chip).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(float4* buf, int n_steps, int n_tiles, int step_major, int spin) {
  const int tile = blockIdx.x, tid = threadIdx.x;
  float4 v = make_float4(tid, tile, 1.f, 2.f);
  for (int s = 0; s < n_steps; ++s) {
    const size_t unit = step_major ? (size_t)s * n_tiles + tile : (size_t)tile * n_steps + s;
    float4* p = buf + unit * 1024 + tid;                 // 16 KB = 1024 float4 per (tile, step)
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j * 256] = v;
    for (int i = 0; i < spin; ++i) v.x = v.x * 1.000001f + 0.5f;   // some "compute" between the bursts
  }
  if (v.x == 12345.f) buf[0] = v;
}
int main() {
  const int n_steps = 240, n_tiles = 1024;
  float4* buf; (void)hipMalloc(&buf, (size_t)n_steps * n_tiles * 16384);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int spin = 0; spin <= 400; spin += 200)
    for (int sm = 0; sm < 2; ++sm) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); k<<<n_tiles, 256>>>(buf, n_steps, n_tiles, sm, spin); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("%-10s spin %3d: %.3f ms  %.2f TB/s\n", sm ? "step-major" : "tile-major", spin, best, (double)n_steps * n_tiles * 16384 / (best * 1e-3) / 1e12);
    }
  return 0;
}
