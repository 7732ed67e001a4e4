// Where do the waves of co-resident workgroups land?  Launches workgroups shaped like the split3 flux kernel
// (4 waves, ~78 kB of LDS: two per CU) and records HW_ID (wave slot, SIMD, CU, SE) and XCC_ID of every wave.
// build: hipcc -O2 --offload-arch=gfx950 tools/probes/hwid_probe.hip -o gpurun_out/hwid_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>

__global__ __launch_bounds__(256) void probe(unsigned *out, long long *when, int spin)
{
  __shared__ double big[9700]; // 77.6 kB
  const int w = threadIdx.x >> 6;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  big[threadIdx.x] = hw;
  __syncthreads();
  long long t0 = clock64();
  // stay resident for a while so that the second workgroup of the CU arrives while this one runs
  double acc = big[(threadIdx.x * 7) % 256];
  for (int i = 0; i < spin; i++) acc = acc * 1.0000001 + 1e-9;
  if ((threadIdx.x & 63) == 0)
  {
    out[(blockIdx.x * 4 + w) * 2 + 0] = hw;
    out[(blockIdx.x * 4 + w) * 2 + 1] = xcc;
    when[blockIdx.x * 4 + w] = t0;
  }
  if (acc == 12345.678) out[0] = 0;
}

int main()
{
  const int nwg = 2048;
  unsigned *d;
  long long *dw;
  hipMalloc(&d, sizeof(unsigned) * nwg * 8);
  hipMalloc(&dw, sizeof(long long) * nwg * 4);
  hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, 0, d, dw, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nwg * 8);
  hipMemcpy(h.data(), d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  std::map<unsigned, std::vector<int>> by_cu;
  for (int b = 0; b < nwg; b++)
  {
    const unsigned hw = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("distinct CUs seen: %zu\n", by_cu.size());
  int shown = 0;
  for (auto &kv : by_cu)
  {
    if (shown++ >= 6) break;
    printf("CU key %05x:", kv.first);
    for (int k = 0; k < (int)kv.second.size() && k < 4; k++)
    {
      const int b = kv.second[k];
      printf("  wg %d [", b);
      for (int w = 0; w < 4; w++)
      {
        const unsigned hw = h[(b * 4 + w) * 2];
        printf(" simd%u/slot%u", (hw >> 4) & 3, hw & 0xf);
      }
      printf(" ]");
    }
    printf("\n");
  }
  // statistics: for the first two workgroups of every CU, do wave w of both sit on the same SIMD?
  long same[4] = {0, 0, 0, 0}, pairs = 0, distinct4 = 0, total = 0;
  for (auto &kv : by_cu)
  {
    for (int b : kv.second)
    {
      unsigned m = 0;
      for (int w = 0; w < 4; w++) m |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
      distinct4 += (m == 0xf);
      total++;
    }
    if (kv.second.size() < 2) continue;
    pairs++;
    for (int w = 0; w < 4; w++)
      same[w] += ((h[(kv.second[0] * 4 + w) * 2] >> 4) & 3) == ((h[(kv.second[1] * 4 + w) * 2] >> 4) & 3);
  }
  printf("workgroups with 4 waves on 4 distinct SIMDs: %ld of %ld\n", distinct4, total);
  printf("first two workgroups of a CU: wave w on the same SIMD in %ld %ld %ld %ld of %ld CUs\n", same[0], same[1], same[2], same[3], pairs);
  return 0;
}
