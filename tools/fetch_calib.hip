// FETCH_SIZE / WRITE_SIZE calibration on KNOWN byte counts in the access shapes the GEMM kernels use (MI355X_MICROARCH.md, HBM section: the x 2 of
// gfx950's FETCH_SIZE is pinned for wide coalesced streaming reads only -- "calibrate on a known byte count in your own access pattern").
// One kernel per pattern so the counters can be read per kernel name (rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE, kernel-trace only);
// every kernel moves exactly BYTES bytes once (503 MB: past the 256 MB Infinity Cache).
//   calib_read_stream    16 B per lane, lanes consecutive (the pinned case: expect FETCH_SIZE = BYTES / 2)
//   calib_read_rowseg    the persistent GEMM's epilogue reading its residual / saved-gelu' rows (csrc/gemm8.hip load_res): per wave 8 rows x 128 B,
//                        lane -> (row = lane >> 3, 16-byte group = lane & 7), non-temporal, four row groups in flight, 256 x 256 tiles
//   calib_read_lds_dma   16 B per lane global_load ... lds (the operand stream of the GEMMs and the scan)
//   calib_write_stream / calib_write_rowseg   the matching stores (the epilogue's 16-byte streaming stores of 128-byte row segments)
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/fetch_calib ; run: tools/fetch_calib   (prints BYTES)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
#define M_ROWS 327680
#define N_COLS 768

__global__ void __launch_bounds__(256) calib_read_stream(const u32x4_t *in, unsigned *sink, size_t n16)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u32x4_t v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(256) calib_write_stream(u32x4_t *out, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = (u32x4_t){(unsigned)i, 1u, 2u, 3u};
}
// 256 x 256 tile per workgroup of 512 threads: wave (wr 0..1, wc 0..3) owns 128 rows x 64 columns (bf16), row order like gemm8's epilogue
__device__ __forceinline__ size_t rowseg_off(int tile_m, int tile_n, int wave, int lane, int mi, int ps)
{
    const int m_w = tile_m * 256 + (wave >> 2) * 128, n_w = tile_n * 256 + (wave & 3) * 64;
    const int row = m_w + (mi * 4 + ps) * 8 + (lane >> 3);
    return ((size_t)row * N_COLS + n_w + (lane & 7) * 8) * 2;
}
__global__ void __launch_bounds__(512) calib_read_rowseg(const char *in, unsigned *sink)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (int t = blockIdx.x; t < (M_ROWS / 256) * (N_COLS / 256); t += gridDim.x) {
        const int tm = t / (N_COLS / 256), tn = t % (N_COLS / 256);
        for (int mi = 0; mi < 4; ++mi) {
            u32x4_t v[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) v[ps] = __builtin_nontemporal_load((const u32x4_t *)(in + rowseg_off(tm, tn, wave, lane, mi, ps)));
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) acc ^= v[ps].x ^ v[ps].y ^ v[ps].z ^ v[ps].w;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void __launch_bounds__(512) calib_write_rowseg(char *out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int t = blockIdx.x; t < (M_ROWS / 256) * (N_COLS / 256); t += gridDim.x) {
        const int tm = t / (N_COLS / 256), tn = t % (N_COLS / 256);
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
                __builtin_nontemporal_store((u32x4_t){(unsigned)t, 1u, 2u, 3u}, (u32x4_t *)(out + rowseg_off(tm, tn, wave, lane, mi, ps)));
    }
}
__global__ void __launch_bounds__(256) calib_read_lds_dma(const char *in, unsigned *sink, size_t n16)
{
    __shared__ __attribute__((aligned(16))) char buf[4][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned acc = 0;
    // every wave copies 1 KiB pieces (64 lanes x 16 B) into its own LDS slot
    for (size_t piece = (size_t)blockIdx.x * 4 + wave; piece < n16 / 64; piece += (size_t)gridDim.x * 4) {
        __builtin_amdgcn_global_load_lds((gptr_t *)(in + piece * 1024 + lane * 16), (lptr_t *)buf[wave], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= ((const unsigned *)buf[wave])[lane];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)M_ROWS * N_COLS * 2, n16 = bytes / 16;
    char *a, *b; unsigned *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes); (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read_stream, dim3(2048), dim3(256), 0, 0, (const u32x4_t *)a, sink, n16); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_read_rowseg, dim3(1024), dim3(512), 0, 0, (const char *)b, sink); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_read_lds_dma, dim3(2048), dim3(256), 0, 0, (const char *)a, sink, n16); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_write_stream, dim3(2048), dim3(256), 0, 0, (u32x4_t *)b, n16); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(calib_write_rowseg, dim3(1024), dim3(512), 0, 0, a); (void)hipDeviceSynchronize();
    }
    printf("BYTES %zu\n", bytes);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
