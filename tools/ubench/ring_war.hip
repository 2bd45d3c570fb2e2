// Minimal reproducer of the LDS-DMA ring race that made K14's two-workgroups-per-CU forms differ run to run (round 5:
// profiles/r05_fine_level.txt; root cause round 6: docs/rounds/r06.md section 1).  Only the ring is kept:
//
//   ring of 3 slots x 16 KiB; slab s of a global image (every dword = s << 16 | index: L2-hot, like the weight images) is requested
//   by asm global_load_lds_dwordx4 two phases ahead, 16 / NW pieces per wave, exactly as K13 / K14 do.  Phase p:
//       early reads of slab p (8 x ds_read_b128 per round, waited for, checked: the RAW side -- control)
//       VALU filler of a per-wave pseudo-random length (desynchronises the waves, as the real kernels' phases do)
//       LATE reads: the slab's last two fragments, issued and NOT waited for -- what hipcc makes of "reads, MFMAs, barrier"
//                   when it sinks the last MFMAs (and their s_waitcnt lgkmcnt) below an `asm volatile` barrier
//       s_waitcnt vmcnt(0)                       own pieces of slab p + 1 landed
//       [FIX: s_waitcnt lgkmcnt(0)]              the rule: no LDS read outstanding at a barrier behind which its slot is re-requested
//       s_barrier
//       request slab p + 3 into slot p % 3       the slot the late reads are still queued for
//       s_waitcnt lgkmcnt(0); check the late reads: they must be slab p.  A value of slab p + 3 = the DMA write overtook the read.
//
// Parameters: workgroups per CU (through the dynamic LDS size), the fix on / off, the byte stride of the early reads (16 = conflict
// free, larger = bank conflicts = a longer LDS queue), the number of early-read rounds, the filler range.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/ubench/ring_war.hip -o /tmp/ring_war && /tmp/ring_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int SLAB = 16384, RING = 3, NIMG = 64;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
__device__ __forceinline__ u32x4 lds_read(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

struct Result { unsigned late_bad, early_bad, late_total, first[8][4]; };

template <int NW, bool FIX>
__global__ __launch_bounds__(64 * NW) void k_ring(const unsigned char* __restrict__ img, int phases, int early_rounds, int stride,
                                                  int fill_max, int compact, Result* res) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NPIECE = 16 / NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned ring_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lptr_t)smem);
    const unsigned char* wsrc = img + (size_t)lane * 16;
    auto request = [&](int s) {
        const unsigned dst = ring_base + (unsigned)((s % RING) * SLAB);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i)      // compact: all 16 pieces of a slab come from the slab's first KiB (a 64 KiB working set: L1 hits, the shortest DMA latency)
            glds16(wsrc + (size_t)(s % NIMG) * SLAB + (compact ? 0 : (wave + NW * i) * 1024), dst + (wave + NW * i) * 1024);
    };
    request(0);
    request(1);
    request(2);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    unsigned late_bad = 0, early_bad = 0;
    float junk = (float)lane;
    unsigned rng = (blockIdx.x * 9781u + wave * 6271u) | 1u;
    for (int p = 0; p < phases; ++p) {
        const unsigned slot = ring_base + (unsigned)((p % RING) * SLAB);
        // ---- early reads of slab p: waited for and checked (control: the RAW direction of the protocol)
        for (int r = 0; r < early_rounds; ++r) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = lds_read(slot + (unsigned)(((lane * stride + (j + 8 * r) * 1024) & (SLAB - 1)) & ~15));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) early_bad += (v[j].x >> 16) != (unsigned)(p % NIMG);
        }
        // ---- filler of a pseudo-random length: the waves of a workgroup (and the two workgroups of a CU) drift apart
        rng = rng * 1664525u + 1013904223u;
        const int fill = fill_max > 0 ? (int)((rng >> 16) % (unsigned)fill_max) : 0;
        for (int i = 0; i < fill; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(junk));
        // ---- the slab's last two fragments: issued, not waited for
        u32x4 la = lds_read(slot + 14 * 1024 + lane * 16), lb = lds_read(slot + 15 * 1024 + lane * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FIX) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(la), "+v"(lb) :: "memory");
        asm volatile("s_barrier" ::: "memory");
        request(p + 3);                                        // into slot p % 3
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(la), "+v"(lb) :: "memory");
        const bool bad = (la.x >> 16) != (unsigned)(p % NIMG) || (lb.x >> 16) != (unsigned)(p % NIMG) ||
                         (la.w >> 16) != (unsigned)(p % NIMG) || (lb.w >> 16) != (unsigned)(p % NIMG);
        if (bad) {
            ++late_bad;
            const unsigned k = atomicAdd(&res->late_bad, 1u);
            if (k < 8) { res->first[k][0] = blockIdx.x; res->first[k][1] = wave * 64 + lane; res->first[k][2] = p; res->first[k][3] = la.x; }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (early_bad) atomicAdd(&res->early_bad, early_bad);
    if (lane == 0) atomicAdd(&res->late_total, (unsigned)phases);
    if (junk == 12345.678f) res->first[7][3] = late_bad;
}

__global__ void k_neighbour(const float4* __restrict__ a, float4* __restrict__ b, long n) {      // a busy second stream: HBM copy
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

struct Env { const unsigned char* img; Result* dres; float4 *na, *nb; hipStream_t side; };

template <int NW, bool FIX>
static void run(const Env& e, int nwg, size_t lds, int phases, int rounds, int stride, int fill, int compact, int neighbour) {
    hipFuncSetAttribute((const void*)k_ring<NW, FIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_ring<NW, FIX>, 64 * NW, lds);
    unsigned long long bad = 0, ebad = 0;
    float ms_sum = 0;
    Result h{};
    const int launches = 5;
    for (int it = 0; it < launches; ++it) {
        hipMemset(e.dres, 0, sizeof(Result));
        hipDeviceSynchronize();
        if (neighbour) hipLaunchKernelGGL(k_neighbour, dim3(512), dim3(256), 0, e.side, e.na, e.nb, (long)(64 << 20) / 16 * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_ring<NW, FIX>), dim3(nwg), dim3(64 * NW), lds, 0, e.img, phases, rounds, stride, fill, compact, e.dres);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
        hipMemcpy(&h, e.dres, sizeof(Result), hipMemcpyDeviceToHost);
        bad += h.late_bad; ebad += h.early_bad;
    }
    printf(" | %dw %dwg/CU fix%d: %7llu late, %llu early, %.2f ms", NW, occ, (int)FIX, bad, ebad, ms_sum / launches);
}

int main(int argc, char** argv) {
    const int nwg = argc > 1 ? atoi(argv[1]) : 8192, phases = 48;
    std::vector<unsigned> himg((size_t)NIMG * SLAB / 4);
    for (int s = 0; s < NIMG; ++s)
        for (int i = 0; i < SLAB / 4; ++i) himg[(size_t)s * (SLAB / 4) + i] = ((unsigned)s << 16) | (unsigned)(i & 0xffff);
    Env e{};
    unsigned char* img;
    hipMalloc(&img, himg.size() * 4); hipMalloc(&e.dres, sizeof(Result));
    hipMalloc(&e.na, 256 << 20); hipMalloc(&e.nb, 256 << 20);
    hipMemset(e.na, 1, 256 << 20);
    hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking);
    hipMemcpy(img, himg.data(), himg.size() * 4, hipMemcpyHostToDevice);
    e.img = img;
    const size_t two = RING * SLAB + 16 * 1024;                // 64 KiB: two workgroups per CU (K14's round-3 form, K13)
    const size_t one = RING * SLAB + 48 * 1024;                // 96 KiB: one
    printf("# per row: 5 launches x %d workgroups x 48 phases; 'late' = lanes whose un-waited fragment read returned the NEXT slab's bytes (the DMA\n"
           "# write overtook the queued read), 'early' = waited reads that were wrong (the RAW direction: control, must be 0)\n", nwg);
    for (int compact : {0, 1})
        for (int neighbour : {0, 1})
            for (int stride : {16, 256})
                for (int rounds : {2, 8})
                    for (int fill : {0, 256, 1024}) {
                        printf("src %s, neighbour %d, early reads %2d x stride %3d, filler < %4d", compact ? "L1-hot" : "L2-hot", neighbour, 8 * rounds, stride, fill);
                        run<4, false>(e, nwg, two, phases, rounds, stride, fill, compact, neighbour);
                        run<4, true>(e, nwg, two, phases, rounds, stride, fill, compact, neighbour);
                        run<4, false>(e, nwg, one, phases, rounds, stride, fill, compact, neighbour);
                        run<8, false>(e, nwg / 2, RING * SLAB + 80 * 1024, phases, rounds, stride, fill, compact, neighbour);
                        run<8, true>(e, nwg / 2, RING * SLAB + 80 * 1024, phases, rounds, stride, fill, compact, neighbour);
                        printf("\n");
                        fflush(stdout);
                    }
    return 0;
}
