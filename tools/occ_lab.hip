// occ_lab.hip -- DIAGNOSTIC: how many 256-thread blocks does a CU really host at once, and what clock do they see?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* rec, int iters) {
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 1e-3f + i;
    unsigned long long t0, t1, c0, c1; unsigned hw;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_getreg_b32 %2, hwreg(HW_REG_HW_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(c0), "=s"(hw) :: "memory");
    for (int it = 0; it < iters; ++it)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(0.999f), "v"(0.1f));
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(c1) :: "memory");
    float s = 0; for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* q = rec + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        q[0] = t0; q[1] = t1; q[2] = c1 - c0; q[3] = ((unsigned long long)xcc << 32) | hw;
    }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, 0);
    printf("device %s CUs %d clock %d kHz maxThreadsPerMP %d regsPerBlock %d sharedPerMP %zu; occupancy API: %d blocks of 256 per CU\n",
           p.name, p.multiProcessorCount, p.clockRate, p.maxThreadsPerMultiProcessor, p.regsPerBlock, p.sharedMemPerMultiprocessor, nb);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = p.multiProcessorCount * w, iters = 20000;
        float* out; unsigned long long* rec;
        hipMalloc(&out, blocks * 256 * 4); hipMalloc(&rec, blocks * 4 * 4 * 8);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, rec, iters); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, rec, iters); hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 16);
        hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long tmin = ~0ull, tmax = 0; double csum = 0, rsum = 0;
        std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;   // per (xcc, se, cu, simd): +1/-1 events
        for (int i = 0; i < blocks * 4; ++i) {
            unsigned long long t0 = h[i * 4], t1 = h[i * 4 + 1], c = h[i * 4 + 2], id = h[i * 4 + 3];
            tmin = std::min(tmin, t0); tmax = std::max(tmax, t1); csum += c; rsum += (t1 - t0);
            unsigned hw = (unsigned)id; unsigned xcc = (unsigned)(id >> 32) & 0xf;
            // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
            unsigned long long key = ((unsigned long long)xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 15) << 4) | ((hw >> 4) & 3);
            ev[key].push_back({t0, +1}); ev[key].push_back({t1, -1});
        }
        int maxc = 0; double avgc = 0;
        for (auto& kv : ev) { auto& v = kv.second; std::sort(v.begin(), v.end()); int c = 0, m = 0; for (auto& e : v) { c += e.second; m = std::max(m, c); } maxc = std::max(maxc, m); avgc += m; }
        printf("w=%d blocks=%d: distinct SIMDs seen %zu, max concurrent waves on one SIMD %d (avg of per-SIMD max %.2f); span %.1f us (100 MHz ticks), "
               "mean wave life %.1f us, shader clock seen by waves %.2f GHz\n", w, blocks, ev.size(), maxc, avgc / ev.size(),
               (tmax - tmin) / 100.0, rsum / (blocks * 4) / 100.0, csum / rsum / 10.0);
        hipFree(out); hipFree(rec);
    }
    return 0;
}
