// emdr2_amd/csrc/mips_scan.hip -- the index scan: fused fp16 MFMA skinny-GEMM + threshold filter.
//
// Replaces the reference's  C = Q * E^T  (dense [Q,N] fp16, 21.5 GB at Q=512)  + torch.topk
// (megatron/data/emdr2_index.py:281-295) with one pass over the HBM-resident stripe-tiled index
// that never materialises C: every workgroup owns a [BM rows x BN queries] fp32 accumulator tile
// in registers (8 waves x 128 VGPRs), streams E and Q chunks through a 3-stage LDS ring filled by
// LDS-DMA (global_load_lds, 16 B/lane), and filters the finished tile against per-query
// thresholds; survivors go to per-query candidate buffers.  See DESIGN.md section 5.
#include "mips_device.h"
#include "mips_kernels.h"

#ifndef NST
#define NST 3                  // LDS ring depth (chunks)
#endif
#define QCAP 2048              // LDS survivor queue entries (16 B each)
#define FLUSH_AT (QCAP / 2)

template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<5>() { asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void glds16(const char *src, char *lds_dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)lds_dst, 16, 0, 0);
}
// same with the non-temporal cache policy (aux bit 1 = nt): for the index rows, which are read exactly once per search
__device__ __forceinline__ void glds16_nt(const char *src, char *lds_dst)
{
    __builtin_amdgcn_global_load_lds((gptr_t *)src, (lptr_t *)lds_dst, 16, 0, 2);
}

// MODE 0: threshold filter -> survivor queue -> candidate buffers (the production path)
// MODE 1: dense: every (row, query) score is written to cand[q][row - row0] (first segment)
// MODE 2: dense: fp32 score matrix out[q][row] (diagnostics)
// ABL (timing experiments only, results are garbage): 1 = no MFMA, 2 = no LDS reads + no MFMA, 3 = no LDS-DMA fill
template <int WM, int WN, int MODE, int ABL = 0>
__global__ void __launch_bounds__(512) mips_scan_kernel(ScanParams p)
{
    constexpr int BM = WM * 64, BN = WN * 128;
    constexpr int E_STAGE = BM * 64, Q_STAGE = BN * 64, STAGE = E_STAGE + Q_STAGE;
    constexpr int E_PW = BM / 128, Q_PW = BN / 128, PPW = E_PW + Q_PW; // LDS-DMA pieces per wave per chunk
    constexpr int STRIPES_PER_TILE = BM / STRIPE_ROWS;
    static_assert(WM * WN == 8 && WM >= 2, "8 waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const qbuf = smem + NST * STAGE;
    unsigned *const qcnt = (unsigned *)(qbuf + QCAP * 16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int swz = (l31 >> 2) & 3;

    int first_tile = p.tile_begin + (int)blockIdx.x;
    if (first_tile >= p.tile_end) return;

    if (MODE == 0 && tid == 0) *qcnt = 0;

    // per-lane operand offsets inside a stage (bytes); mi / ni add multiples of 2048
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sp = ((ks * 2 + hi) ^ swz) << 4;
        a_off[ks] = (wm * 64 + l31) * 64 + sp;
        b_off[ks] = E_STAGE + (wn * 128 + l31) * 64 + sp;
    }

    float tau[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int q = wn * 128 + ni * 32 + l31;
        tau[ni] = (MODE == 0) ? ((q < p.n_q && ABL == 0) ? p.tau[q] : __builtin_inff()) : 0.f;
    }

    const int nch = p.nch;
    const int tstep = (int)gridDim.x;

    // prefetch cursor (two chunks ahead of compute)
    int pf_tile = first_tile, pf_c = 0, pf_stage = 0;
    auto issue = [&]() {
        const int t = (ABL == 4) ? 0 : (pf_tile < p.tile_end ? pf_tile : first_tile); // past the end: harmless re-read
        char *sb = smem + pf_stage * STAGE;
#pragma unroll
        for (int j = 0; j < E_PW; ++j) {
            const int pe = wave + 8 * j;
            const size_t stripe = (size_t)t * STRIPES_PER_TILE + (pe >> 3);
            if (ABL != 3) {
                const char *src = p.e_tiled + (stripe * nch + pf_c) * STRIPE_CHUNK_BYTES + (pe & 7) * 1024 + lane * 16;
                if (p.tune & 16) glds16_nt(src, sb + pe * 1024); else glds16(src, sb + pe * 1024);
            }
        }
#pragma unroll
        for (int j = 0; j < Q_PW; ++j) {
            const int pq = wave + 8 * j;
            if (ABL != 3 && ABL != 5) glds16(p.q_tiled + (size_t)pf_c * Q_STAGE + pq * 1024 + lane * 16, sb + E_STAGE + pq * 1024);
        }
        if (++pf_c == nch) { pf_c = 0; pf_tile += tstep; }
        pf_stage = (pf_stage == NST - 1) ? 0 : pf_stage + 1;
    };

    for (int i = 0; i < NST - 1; ++i) issue();
    int cs = 0; // compute stage

    for (int tile = first_tile; tile < p.tile_end; tile += tstep) {
        floatx16 acc[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

        for (int c = 0; c < nch; ++c) {
            wait_vmcnt<PPW *(NST - 2)>(); // my pieces of the chunk about to be consumed have landed
            __builtin_amdgcn_s_barrier(); // everyone's have; everyone is done with the stage refilled next
            if (MODE == 0 && c == 0) {
                // survivor queue high-water check (uniform: all pushes of the previous tile are
                // ordered before this barrier, none of this tile's happen before the next one)
                const unsigned n = *(volatile __attribute__((address_space(3))) unsigned *)qcnt;
                if (n >= FLUSH_AT) {
                    const unsigned m = n < QCAP ? n : QCAP;
                    for (unsigned i = tid; i < m; i += 512) {
                        const uint4 e = ((const uint4 *)qbuf)[i];
                        const unsigned slot = atomicAdd(&p.count[e.z], 1u);
                        if (slot < p.capq) p.cand[(size_t)e.z * p.capq + slot] = make_uint2(e.x, e.y);
                        else atomicOr(&p.flags[e.z], 2u);
                    }
                    __syncthreads();
                    if (tid == 0) *qcnt = 0;
                    __syncthreads();
                }
            }
            // where in the chunk this wave issues its LDS-DMA pieces (p.tune bit 3): the two waves of a SIMD at different points,
            // so a wave stalled in a back-pressured VMEM issue has a partner that is issuing MFMAs
            const int ipos = (p.tune & 8) ? (wave >> 2) : 0;
            if (ipos == 0) issue();
            const char *sb = smem + cs * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 1 && ipos == 1) issue();
                half8 a[2], b[4];
                if (ABL == 2) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[mi] = half8{};
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) b[ni] = half8{};
                } else {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) a[mi] = *(const half8 *)(sb + a_off[ks] + mi * 2048);
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) b[ni] = *(const half8 *)(sb + b_off[ks] + ni * 2048);
                }
                if (ABL == 1 || ABL == 2) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) asm volatile("" ::"v"(a[mi]));
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(b[ni]));
                } else {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 4; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
            }
            cs = (cs == NST - 1) ? 0 : cs + 1;
        }

        // ---- tile epilogue -----------------------------------------------------------------
        // C layout of v_mfma_f32_32x32x16: col (query) = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const int row0 = tile * BM + wm * 64 + 4 * hi;
        if (MODE == 0) {
            bool stored = false;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const unsigned q = wn * 128 + ni * 32 + l31;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    // one max + one ballot per 32 x 32 accumulator block (a survivor costs 16 register checks, not 32)
                    float m = acc[mi][ni][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[mi][ni][r]);
                    if (__builtin_amdgcn_ballot_w64(m >= tau[ni]) == 0) continue; // the common case
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mi][ni][r];
                        const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                        const bool pass = (v >= tau[ni]) && (row < p.n_rows);
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
                        if (mask == 0) continue;
                        unsigned base = 0;
                        if (lane == 0) base = atomicAdd(qcnt, (unsigned)__popcll(mask));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (pass) {
                            const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                            if (slot < QCAP) {
                                ((uint4 *)qbuf)[slot] = make_uint4(__float_as_uint(v), (unsigned)row, q, 0u);
                            } else { // queue full: straight to the candidate buffer
                                const unsigned g = atomicAdd(&p.count[q], 1u);
                                if (g < p.capq) p.cand[(size_t)q * p.capq + g] = make_uint2(__float_as_uint(v), (unsigned)row);
                                else atomicOr(&p.flags[q], 2u);
                                stored = true;
                            }
                        }
                    }
                }
            }
            if (__builtin_amdgcn_ballot_w64(stored)) wait_vmcnt0(); // keep the counted LDS-DMA waits exact
        } else {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int q = wn * 128 + ni * 32 + l31;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                        if (q < p.n_q && row < p.n_rows) {
                            if (MODE == 1) p.cand[(size_t)q * p.capq + (unsigned)(row - p.dense_row0)] = make_uint2(__float_as_uint(acc[mi][ni][r]), (unsigned)row);
                            else p.dense_out[(size_t)q * p.n_rows + row] = acc[mi][ni][r];
                        }
                    }
            }
            wait_vmcnt0();
        }
    }

    wait_vmcnt0(); // drain the two speculative chunks before the LDS is released
    if (MODE == 0) {
        __syncthreads();
        const unsigned n = *(volatile __attribute__((address_space(3))) unsigned *)qcnt;
        const unsigned m = n < QCAP ? n : QCAP;
        for (unsigned i = tid; i < m; i += 512) {
            const uint4 e = ((const uint4 *)qbuf)[i];
            const unsigned slot = atomicAdd(&p.count[e.z], 1u);
            if (slot < p.capq) p.cand[(size_t)e.z * p.capq + slot] = make_uint2(e.x, e.y);
            else atomicOr(&p.flags[e.z], 2u);
        }
    }
}


// ================================================================================================
// v2: "ping-pong" schedule.  Waves 0-3 (one per SIMD) and waves 4-7 (their SIMD partners) run the
// same chunk sequence half a period apart: while one group issues its 12 fragment ds_reads and the
// LDS-DMA for a later chunk, the other group owns the matrix pipe with 16 back-to-back MFMAs.
// Barrier events e: chunk c is visible to all at e=2c, read by group A in (2c,2c+1), by group B in
// (2c+1,2c+2), and its ring slot is refilled (chunk c+3) right after e=2c+2.  Production (MODE 0) only.
// ================================================================================================
struct Frags {
    half8 a[2][2]; // [ks][mi]
    half8 b[2][4]; // [ks][ni]
};

__device__ __forceinline__ void pp_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void read_frags(const char *sb, const int (&a_off)[2], const int (&b_off)[2], Frags &f)
{
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) f.a[ks][mi] = *(const half8 *)(sb + a_off[ks] + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) f.b[ks][ni] = *(const half8 *)(sb + b_off[ks] + ni * 2048);
    }
}

__device__ __forceinline__ void mma_chunk(const Frags &f, floatx16 (&acc)[2][4], int prio = 1)
{
    if (prio & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[ks][mi], f.b[ks][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

__device__ __forceinline__ void flush_queue(const ScanParams &p, char *qbuf, unsigned *qcnt, unsigned n, int tid)
{
    const unsigned m = n < QCAP ? n : QCAP;
    for (unsigned i = tid; i < m; i += 512) {
        const uint4 e = ((const uint4 *)qbuf)[i];
        const unsigned slot = atomicAdd(&p.count[e.z], 1u);
        if (slot < p.capq) p.cand[(size_t)e.z * p.capq + slot] = make_uint2(e.x, e.y);
        else atomicOr(&p.flags[e.z], 2u);
    }
}

// threshold filter of a finished accumulator tile (C layout: query = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5))
__device__ __forceinline__ void filter_tile(const ScanParams &p, const floatx16 (&acc)[2][4], const float (&tau)[4], int row0,
                                            unsigned qbase, int lane, char *qbuf, unsigned *qcnt)
{
    bool stored = false;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        float m = acc[0][ni][0];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mi][ni][r]);
        if (__builtin_amdgcn_ballot_w64(m >= tau[ni]) == 0) continue; // the common case
        const unsigned q = qbase + ni * 32;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[mi][ni][r];
                const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                const bool pass = (v >= tau[ni]) && (row < p.n_rows);
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
                if (mask == 0) continue;
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(qcnt, (unsigned)__popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (pass) {
                    const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    if (slot < QCAP) {
                        ((uint4 *)qbuf)[slot] = make_uint4(__float_as_uint(v), (unsigned)row, q, 0u);
                    } else {
                        const unsigned g = atomicAdd(&p.count[q], 1u);
                        if (g < p.capq) p.cand[(size_t)q * p.capq + g] = make_uint2(__float_as_uint(v), (unsigned)row);
                        else atomicOr(&p.flags[q], 2u);
                        stored = true;
                    }
                }
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(stored)) wait_vmcnt0();
}

// survivors straight to the candidate buffers: one returning atomic per lane and query block
__device__ __forceinline__ void filter_tile_direct(const ScanParams &p, const floatx16 (&acc)[2][4], const float (&tau)[4], int row0,
                                                   unsigned qbase)
{
    bool stored = false;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        float m = acc[0][ni][0];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[mi][ni][r]);
        if (__builtin_amdgcn_ballot_w64(m >= tau[ni]) == 0) continue; // the common case
        if (m >= tau[ni]) {
            const unsigned q = qbase + ni * 32;
            unsigned cnt = 0;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                    cnt += (acc[mi][ni][r] >= tau[ni]) && (row < p.n_rows);
                }
            if (cnt) {
                unsigned g = atomicAdd(&p.count[q], cnt);
                if (g + cnt > p.capq) atomicOr(&p.flags[q], 2u);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = row0 + mi * 32 + (r & 3) + 8 * (r >> 2);
                        if ((acc[mi][ni][r] >= tau[ni]) && (row < p.n_rows)) {
                            if (g < p.capq) p.cand[(size_t)q * p.capq + g] = make_uint2(__float_as_uint(acc[mi][ni][r]), (unsigned)row);
                            ++g;
                        }
                    }
                stored = true;
            }
        }
    }
    if (__builtin_amdgcn_ballot_w64(stored)) wait_vmcnt0(); // keep the counted LDS-DMA waits exact
}

#ifdef EMDR2_EXPERIMENTS   // schedule variants kept for tools/ only (`make exp`): not product code, not in this directory
#include "../../tools/exp/mips_scan_variants.inc"
#endif
template <int WM, int WN, int MODE, int ABL = 0>
static int launch_scan_t(const ScanParams &p, int grid, hipStream_t stream)
{
    constexpr int STAGE = (WM * 64 + WN * 128) * 64;
    constexpr int LDS = NST * STAGE + QCAP * 16 + 16;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)mips_scan_kernel<WM, WN, MODE, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -3;
        attr_done = true;
    }
    hipLaunchKernelGGL((mips_scan_kernel<WM, WN, MODE, ABL>), dim3(grid), dim3(512), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

#ifdef EMDR2_EXPERIMENTS
#include "../../tools/exp/mips_scan_variants_launch.inc"
#endif

int mips_launch_scan(int variant, int mode, const ScanParams &p, int grid, hipStream_t stream)
{
    switch (variant * 3 + mode) {
    case 0: return launch_scan_t<2, 4, 0>(p, grid, stream);
    case 1: return launch_scan_t<2, 4, 1>(p, grid, stream);
    case 2: return launch_scan_t<2, 4, 2>(p, grid, stream);
    case 3: return launch_scan_t<4, 2, 0>(p, grid, stream);
    case 4: return launch_scan_t<4, 2, 1>(p, grid, stream);
    case 5: return launch_scan_t<4, 2, 2>(p, grid, stream);
    case 6: return launch_scan_t<8, 1, 0>(p, grid, stream);
    case 7: return launch_scan_t<8, 1, 1>(p, grid, stream);
    case 8: return launch_scan_t<8, 1, 2>(p, grid, stream);
    }
    return -1;
}
