// fetch_cal.hip -- calibrates rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ against a known byte count for the access patterns of
// this repo's kernels (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
// Every kernel reads a 1 GiB buffer exactly once (four times the Infinity Cache) and sums it:
//   wide   a wave reads 64 x 16 contiguous bytes per instruction (conv1d, the B / C repack)
//   line   a wave reads 8 rows x one 128-byte line per instruction, rows 40 KB apart (lanes=channels tiles, whole lines)
//   half   a wave reads 16 rows x 64 bytes per instruction; the other half of each line one "tile" later (64-byte tile rows)
//   piece  a wave reads 4 rows x 32 bytes (16 lanes x 2 bytes) per instruction, the next 32 bytes of the rows the next time
//          (lanes=states)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/fetch_cal tools/fetch_cal.hip ;  run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./tools/fetch_cal     (and a pass with TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr size_t kBytes = (size_t)1 << 30;
constexpr int kRow = 40960;                       // bytes between rows (a 20480-token bf16 row)

__global__ void __launch_bounds__(256) cal_wide(const u32x4* __restrict__ p, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *out = acc;
}
// rows of kRow bytes; a wave owns 64 rows and walks along them.  PIECE bytes of each row per visit.
template <int PIECE>
__global__ void __launch_bounds__(256) cal_rows(const unsigned char* __restrict__ p, size_t rows, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t row0 = wave * 64;
    if (row0 >= rows) return;
    unsigned acc = 0;
    constexpr int LPR = PIECE >= 16 ? PIECE / 16 : 1;       // lanes per row piece (16-byte vectors)
    for (int off = 0; off < kRow; off += PIECE) {
        if (PIECE >= 16) {
            constexpr int RPI = 64 / LPR;                    // rows per instruction
            for (int i = 0; i < 64 / RPI; ++i) {
                const size_t r = row0 + i * RPI + lane / LPR;
                const u32x4 v = *reinterpret_cast<const u32x4*>(p + r * kRow + off + (lane % LPR) * 16);
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        } else {                                             // PIECE == 2: 16 lanes x 2 bytes of 4 rows per instruction
            for (int i = 0; i < 16; ++i) {
                const size_t r = row0 + i * 4 + lane / 16;
                acc += *reinterpret_cast<const unsigned short*>(p + r * kRow + (off / 2) * 32 + (lane % 16) * 2);
            }
        }
    }
    if (acc == 0x12345678u) *out = acc;
}
// the lanes=states pattern proper: 32 bytes of a row per visit (16 lanes x 2 bytes), 4 rows per instruction
__global__ void __launch_bounds__(256) cal_piece(const unsigned char* __restrict__ p, size_t rows, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t row0 = wave * 16;                           // 16 rows per wave: 4 per instruction, 4 "channels" in turn
    if (row0 >= rows) return;
    unsigned acc = 0;
    for (int off = 0; off < kRow; off += 32)
        for (int c = 0; c < 4; ++c) {
            const size_t r = row0 + (lane / 16) * 4 + c;
            acc += *reinterpret_cast<const unsigned short*>(p + r * kRow + off + (lane % 16) * 2);
        }
    if (acc == 0x12345678u) *out = acc;
}

int main() {
    unsigned char* buf; unsigned* out;
    hipMalloc(&buf, kBytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, kBytes);
    const size_t rows = kBytes / kRow;                       // 26214 rows
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(cal_wide, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const u32x4*>(buf), kBytes / 16, out);
        hipLaunchKernelGGL(cal_rows<128>, dim3((unsigned)((rows / 64 + 3) / 4)), dim3(256), 0, 0, buf, rows - rows % 64, out);
        hipLaunchKernelGGL(cal_rows<64>, dim3((unsigned)((rows / 64 + 3) / 4)), dim3(256), 0, 0, buf, rows - rows % 64, out);
        hipLaunchKernelGGL(cal_rows<32>, dim3((unsigned)((rows / 64 + 3) / 4)), dim3(256), 0, 0, buf, rows - rows % 64, out);
        hipLaunchKernelGGL(cal_piece, dim3((unsigned)((rows / 16 + 3) / 4)), dim3(256), 0, 0, buf, rows - rows % 16, out);
    }
    hipDeviceSynchronize();
    printf("bytes read once per kernel: wide %zu, rows<128/64/32> %zu, piece %zu\n", kBytes, (rows - rows % 64) * (size_t)kRow,
           (rows - rows % 16) * (size_t)kRow);
    return 0;
}
