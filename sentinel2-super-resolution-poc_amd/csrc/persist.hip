// Diagnostic prototype (nothing of the product calls it; bench.py / tools only): what would a trunk schedule sustain whose workgroups STAY across
// the layers of the RDBs -- no launch boundaries, planes handed to the neighbours through flags -- so that a launch group small enough for the
// Infinity Cache (profiles/r05_mall_probe.txt) runs without the start and drain of 345 launches?  (profiles/r05_mfma_ceiling.txt: the conv1-4 fill
// from a cache-resident source sustains 1.27 PFLOP/s against 1.02 from HBM; profiles/r05_two_streams.txt: small launch groups do not get there.)
//
// The loop has the RDB's shape and traffic and none of its arithmetic meaning (like ceiling.hip): one workgroup per CU, 8 waves, P patches each;
// per patch twelve 32-KiB slab planes (x: 4, x1..x4: 2 each; 16 channels x 1024 pixels of fp16) + a second x (the trunk ping-pong) = 512 KiB, the
// real figure.  Layer L = 1..4 of an RDB runs 2 + 2L stages per patch (x, x1 .. x_{L-1}: oldest planes first), layer 5 twelve stages with twice
// the MFMAs (64 output channels); a stage = 288 (576) MFMAs, 0.75 (0.44) LDS reads per MFMA, 48 KiB of LDS-DMA into a 3-deep ring: 32 KiB of the
// patch's own slab, 4 KiB of a neighbour workgroup's slab (the halo), 12 KiB of weights from a small cached buffer.  Behind a patch's last stage
// its output planes are stored (64 KiB; layer 5: 128 KiB into the other x).  Layer-major order: all patches of layer L, then layer L + 1.
//
// Dependencies: a per-patch counter of finished layers in UNCACHED device memory.  It is published two stages after the stores were issued (the
// counted vmcnt of that stage's top covers them, and the barrier behind it makes that true for all eight waves), and read with scalar loads
// (lgkmcnt: the LDS-DMA ring's vmcnt is not drained by a poll) before the LDS-DMA of the first stage that needs the plane: own patch, left and right
// neighbour.  Only the last two stages of a layer read the plane its predecessor just wrote, so the wait sits behind 2..10 stages of older planes.
// Every wait is bounded (kMaxSpin polls, then a global abort word: the kernel always drains; `timeouts` reports it).  All workgroups must be
// resident at once: the host launches at most one per CU on an otherwise idle device.
//
// COH = 0: plain loads and stores (cross-XCD visibility of the planes is NOT guaranteed inside a kernel: timing only);
// COH = 1: slab and halo pieces loaded with sc1 (device scope: past the XCD's own L2), outputs stored with sc0 sc1 (written through).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "s2sr_internal.h"

namespace {

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Two geometries of the same work (same MFMAs, LDS reads per MFMA and bytes per MFMA):
//   G0  the trunk kernel's: 48-KiB stages, 3-deep ring (two stages of look-ahead), 6 LDS-DMA pieces and 36 (72) MFMAs per wave and stage,
//       32-KiB slabs: x 4, x1..x4 2 each;
//   G1  deeper look-ahead on the same 144 KiB of LDS: 32-KiB stages, 4-deep ring (three stages ahead), 4 pieces and 24 (48) MFMAs per wave and
//       stage, 20-KiB slabs: x 6, x1..x4 3 each (layer L: 3 + 3L stages, layer 5: 18) -- does the latency a 2.4-GHz clock exposes go away?
template <int DEEP>
struct Geo {
    static constexpr int STAGE = DEEP ? 32 * 1024 : 48 * 1024, RING = DEEP ? 4 : 3, LOOK = RING - 1, PW = DEEP ? 4 : 6;
    static constexpr int SLAB = DEEP ? 20 * 1024 : 32 * 1024, OWN = DEEP ? 20 : 32;           // KiB of the own slab per stage
    static constexpr int XS = DEEP ? 6 : 4, GS = DEEP ? 3 : 2;                                  // slabs of x, of a growth plane
    static constexpr int SLABS = 2 * XS + 4 * GS;
    static constexpr int STEPS = DEEP ? 8 : 12;                                                  // light stage; heavy: twice
};
constexpr uint32_t kMaxSpin = 200000;

__device__ __forceinline__ void mfma(f32x16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int COH>
__device__ __forceinline__ void glds16(const char* base, uint32_t voff, uint32_t lds_addr, bool coherent) {
    const uint64_t v = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    base = (const char*)(((uint64_t)hi << 32) | lo);
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    if (COH && coherent) asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ f16x8 lds16(const char* smem, uint32_t off) { return *(const f16x8*)(smem + off); }
__device__ __forceinline__ uint32_t poll(const uint32_t* p) {        // scalar, past the scalar cache; the memory itself is uncached
    uint32_t v;
    asm volatile("s_dcache_inv\n\ts_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// the flattened sequence of stages of one workgroup: RDB r, layer L (1..5), patch p, stage st
template <class G>
struct Cursor {
    int r, L, p, st;
    __device__ __forceinline__ int ns() const { return G::XS + G::GS * (L - 1); }              // layer 5 reads x and all four growth planes
    __device__ __forceinline__ bool last_of_patch() const { return st + 1 == ns(); }
    __device__ __forceinline__ void next(int P) {
        if (++st < ns()) return;
        st = 0;
        if (++p < P) return;
        p = 0;
        if (++L <= 5) return;
        L = 1;
        ++r;
    }
    // which slab the stage reads, and the layer count its plane needs (0: the input, always there)
    __device__ __forceinline__ int slab() const { return st < G::XS ? ((r & 1) ? G::XS : 0) + st : 2 * G::XS + (st - G::XS); }
    __device__ __forceinline__ uint32_t needs() const {
        return st < G::XS ? 5u * (uint32_t)r : 5u * (uint32_t)r + (uint32_t)((st - G::XS) / G::GS + 1);
    }
};

template <int COH, int DEEP, int CHECK = 0>
__global__ void __launch_bounds__(512) rdb_persistent_kernel(const char* __restrict__ wts, uint32_t wchunks, char* __restrict__ ws, uint32_t* flags,
                                                             float* __restrict__ sink, int P, int rdbs, uint32_t* timeouts) {
    using G = Geo<DEEP>;
    constexpr int STAGE = G::STAGE, RING = G::RING, LOOK = G::LOOK, PW = G::PW, SLAB = G::SLAB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x, nwg = gridDim.x;
    const size_t nq = (size_t)nwg * P;
    {
        const uint4* s4 = (const uint4*)(wts + (size_t)(w % wchunks) * (48 * 1024));
        for (int i = threadIdx.x; i < RING * STAGE / 16; i += 512) ((uint4*)smem)[i] = s4[i % (STAGE / 16)];
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const uint32_t lane_off = (uint32_t)lane * 16;
    f16x8 a0 = lds16(smem, lane_off), a1 = lds16(smem, 1024 + lane_off), a2 = lds16(smem, 2048 + lane_off), b = lds16(smem, 3072 + lane_off);
    const int wl = (w + nwg - 1) % nwg, wr = (w + 1) % nwg;
    uint32_t* abort_word = flags + nq;                         // behind the counters
    uint64_t seen = 0;                                         // per patch (P <= 4) 16 bits: the lowest of the three counters when last polled
    auto seen_get = [&](int p) { return (uint32_t)(seen >> (16 * p)) & 0xffffu; };                    // (shifts, not an array: an array lands in scratch,
    auto seen_set = [&](int p, uint32_t v) {                                                          //  and a scratch load drains the ring's vmcnt)
        seen = (seen & ~((uint64_t)0xffffu << (16 * p))) | ((uint64_t)(v & 0xffffu) << (16 * p));
    };
    uint32_t my_timeouts = 0;

    auto slab_ptr = [&](int slab, int wg, int p) { return ws + ((size_t)slab * nq + (size_t)wg * P + p) * SLAB; };
    // wait until the own and both neighbour patches p have finished `need` layers (bounded)
    auto acquire = [&](int p, uint32_t need) {
        if (need == 0 || seen_get(p) >= need) return;
        uint32_t spins = 0;
        for (;;) {
            const uint32_t c0 = poll(flags + (size_t)w * P + p), c1 = poll(flags + (size_t)wl * P + p), c2 = poll(flags + (size_t)wr * P + p);
            const uint32_t m = c0 < c1 ? (c0 < c2 ? c0 : c2) : (c1 < c2 ? c1 : c2);
            seen_set(p, m);
            if (m >= need) return;
            if (poll(abort_word) != 0) return;
            if (++spins > kMaxSpin) {
                ++my_timeouts;
                if (lane == 0) atomicAdd(timeouts, 1u), atomicExch(abort_word, 1u);
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };
    // LDS-DMA of one stage: this wave's PW pieces into ring slot `slot` -- three quarters of the stage streamed (own slab + halo), a quarter weights
    auto dma = [&](const Cursor<G>& c, int slot, int piece_i) {
        const uint32_t lds = (uint32_t)slot * STAGE + (uint32_t)(wave * PW + piece_i) * 1024;
        int kind, off;                                          // 0 own slab (KiB offset), 1 halo, 2 weights
        if (!DEEP) {                                            // 48 KiB: own 32 (4 per wave), halo 4 (even waves' last piece), weights 12
            kind = piece_i < 4 ? 0 : (piece_i == 5 && (wave & 1) == 0) ? 1 : 2;
            off = wave * 4 + piece_i;
        } else {                                                // 32 KiB: own 20 (2 per wave + waves 0-3 a third), halo 4 (waves 4-7), weights 8
            kind = piece_i < 2 ? 0 : piece_i == 2 ? (wave < 4 ? 0 : 1) : 2;
            off = piece_i < 2 ? wave * 2 + piece_i : 16 + wave;
        }
        if (kind == 0) glds16<COH>(slab_ptr(c.slab(), w, c.p) + (size_t)off * 1024, lane_off, lds, true);
        else if (kind == 1) glds16<COH>(slab_ptr(c.slab(), (wave & 2) ? wr : wl, c.p) + (size_t)(wave * 2) * 1024, lane_off, lds, true);
        else {
            const uint32_t chunk = (uint32_t)(c.L * 18 + c.st) % wchunks;
            glds16<COH>(wts + (size_t)chunk * (48 * 1024) + (size_t)(wave * PW + piece_i) * 1024, lane_off, lds, false);
        }
    };

    Cursor<G> cur{0, 1, 0, 0}, ahead{0, 1, 0, 0};               // `ahead`: the stage whose DMA is issued next (LOOK ahead of `cur`)
    long total = 0;
    for (int L = 1; L <= 5; ++L) total += (long)(G::XS + G::GS * (L - 1)) * P;
    total *= rdbs;
    for (int k = 0; k < LOOK && k < total; ++k) {               // prologue: the first LOOK stages
        acquire(ahead.p, ahead.needs());
#pragma unroll
        for (int i = 0; i < PW; ++i) dma(ahead, k % RING, i);
        ahead.next(P);
    }
    int e_age = 0, e_cnt = 0;                                   // epilogue stores in flight: counted in the next LOOK tops
    int pub_p = -1;
    uint32_t pub_val = 0;
    for (long s = 0; s < total; ++s) {
        const uint32_t slot = (uint32_t)(s % RING) * STAGE;
        // my pieces of stage s have landed (younger: the pieces of the LOOK - 1 stages behind it, and the stores of a patch that ended within the
        // last LOOK stages, which were issued behind those pieces)
        if (e_age > 0 && e_cnt == 16) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW * (LOOK - 1) + 16) : "memory");
        else if (e_age > 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW * (LOOK - 1) + 8) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW * (LOOK - 1)) : "memory");
        if (e_age > 0) --e_age;
        else if (pub_p >= 0) {
            // LOOK stages behind the stores: every wave's have completed (its wait above, then the barrier) -- publish the patch's counter
            if (wave == 0 && lane == 0) __hip_atomic_store(flags + (size_t)w * P + pub_p, pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pub_p = -1;
        }
        if (CHECK && !DEEP && cur.needs() != 0 && (wave & 1) == 0) {
            // CHECK: the plane stores carry (layer count << 12 | writer); the halo piece of this stage, written by the neighbour workgroup (another
            // XCD: workgroups go round the XCDs) and landed in LDS just now, must carry the count this stage waited for -- and so must the own slab
            const uint32_t got_h = *(const uint32_t*)(smem + slot + (uint32_t)(wave * PW + 5) * 1024 + lane_off);
            const uint32_t got_o = *(const uint32_t*)(smem + slot + (uint32_t)(wave * PW) * 1024 + lane_off);
            const uint32_t wn = (uint32_t)((wave & 2) ? wr : wl);
            if (got_h != ((cur.needs() << 12) | wn)) atomicAdd(timeouts + 1, 1u);
            if (got_o != ((cur.needs() << 12) | (uint32_t)w)) atomicAdd(timeouts + 2, 1u);
        }
        const bool have_ahead = s + LOOK < total;
        if (have_ahead) acquire(ahead.p, ahead.needs());
        const bool heavy = cur.L == 5;
        auto body = [&](auto heavy_c) __attribute__((always_inline)) {
            constexpr bool HEAVY = decltype(heavy_c)::value;
            constexpr int NSTEPS = HEAVY ? 2 * G::STEPS : G::STEPS;
            int rd = 0;
#pragma unroll
            for (int st = 0; st < NSTEPS; ++st) {
                f16x8 na0 = a0, na1 = a1, nb = b;
                auto piece = [&](int k) { return slot + (((uint32_t)wave * PW + (uint32_t)k) % (STAGE / 1024)) * 1024 + lane_off; };
                nb = lds16(smem, piece(rd++));                  // 0.75 LDS reads per MFMA (heavy: 0.44: an A fragment serves both output-channel tiles)
                if (HEAVY) {
                    if (st % 3 == 2) na0 = lds16(smem, piece(rd++));
                } else {
                    na0 = lds16(smem, piece(rd++));
                    if (st % 4 == 3) na1 = lds16(smem, piece(rd++));
                }
                if (st < PW) { if (have_ahead) dma(ahead, (int)((s + LOOK) % RING), st); }
                mfma(acc[(3 * st + 0) & 3], a0, b);
                mfma(acc[(3 * st + 1) & 3], a1, b);
                mfma(acc[(3 * st + 2) & 3], a2, b);
                a0 = na0; a1 = na1; b = nb;
            }
        };
        if (heavy) body(std::true_type{});
        else body(std::false_type{});
        if (have_ahead) ahead.next(P);
        if (cur.last_of_patch()) {
            // the layer's output planes of this patch: x_L (64 KiB) or, layer 5, the other x (128 KiB); 1 KiB per store and wave
            const int nst = heavy ? 16 : 8;
            const int first = heavy ? (((cur.r + 1) & 1) ? G::XS : 0) : 2 * G::XS + G::GS * (cur.L - 1);
            f32x4 v4 = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
            if (CHECK) {
                const float tag = __builtin_bit_cast(float, ((5u * (uint32_t)cur.r + (uint32_t)cur.L) << 12) | (uint32_t)w);
                v4 = (f32x4){tag, tag, tag, tag};
            }
            for (int k = 0; k < nst; ++k) {
                const int j = wave * nst + k;
                int sl = first + j / (SLAB / 1024);
                if (sl >= G::SLABS) sl = G::SLABS - 1;             // (G1: 64 KiB into 3 slabs of 20: the last 4 KiB stay inside the patch's planes)
                char* q = slab_ptr(sl, w, cur.p) + (size_t)(j % (SLAB / 1024)) * 1024 + lane_off;
                if (COH) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v4) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(v4) : "memory");
            }
            e_age = LOOK;
            e_cnt = nst;
            pub_p = cur.p;
            pub_val = 5u * (uint32_t)cur.r + (uint32_t)cur.L;
        }
        cur.next(P);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += acc[i][j];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = sum + (float)my_timeouts;
}

}  // namespace

namespace s2sr {

// variant: bit 0 = device-scope plane loads + written-through plane stores, bit 1 = the deeper geometry (G1); 4 / 5 = variant 0 / 1 with the
// hand-over CHECKED: plane stores carry (layer count, writer), every landed halo piece and own piece is compared (timeouts[1], [2] = mismatches).
// wts: >= 8 chunks of 48 KiB (cached operand / weight data); ws: rdb_persistent_ws_bytes; flags: grid x P + 1 words of UNCACHED device memory,
// zeroed by the caller before every launch; sink: grid x 512 floats; timeouts: three words (zeroed by the caller).
hipError_t launch_rdb_persistent(int variant, const char* d_wts, size_t wts_bytes, char* d_ws, uint32_t* d_flags, float* d_sink, int grid, int P,
                                 int rdbs, uint32_t* d_timeouts, hipStream_t st) {
    if (variant < 0 || variant > 5 || grid <= 0 || P < 2 || P > 4 || rdbs < 1 || wts_bytes < (size_t)48 * 1024 * 8) return hipErrorInvalidValue;
    typedef void (*K)(const char*, uint32_t, char*, uint32_t*, float*, int, int, uint32_t*);
    static const K kern[6] = {rdb_persistent_kernel<0, 0>, rdb_persistent_kernel<1, 0>, rdb_persistent_kernel<0, 1>, rdb_persistent_kernel<1, 1>,
                              rdb_persistent_kernel<0, 0, 1>, rdb_persistent_kernel<1, 0, 1>};
    static const size_t lds[6] = {(size_t)Geo<0>::RING * Geo<0>::STAGE, (size_t)Geo<0>::RING * Geo<0>::STAGE, (size_t)Geo<1>::RING * Geo<1>::STAGE,
                                  (size_t)Geo<1>::RING * Geo<1>::STAGE, (size_t)Geo<0>::RING * Geo<0>::STAGE, (size_t)Geo<0>::RING * Geo<0>::STAGE};
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [&] {
        for (int v = 0; v < 6 && attr_err == hipSuccess; ++v)
            attr_err = hipFuncSetAttribute((const void*)kern[v], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds[v]);
    });
    if (attr_err != hipSuccess) return attr_err;
    uint32_t wchunks = (uint32_t)(wts_bytes / (48 * 1024));
    if (wchunks > 160) wchunks = 160;
    hipLaunchKernelGGL(kern[variant], dim3(grid), dim3(512), lds[variant], st, d_wts, wchunks, d_ws, d_flags, d_sink, P, rdbs, d_timeouts);
    return hipGetLastError();
}

size_t rdb_persistent_ws_bytes(int variant, int grid, int P) {
    return (variant < 4 && (variant & 2)) ? (size_t)Geo<1>::SLABS * grid * P * Geo<1>::SLAB : (size_t)Geo<0>::SLABS * grid * P * Geo<0>::SLAB;
}
double rdb_persistent_flop_per_launch(int grid, int P, int rdbs) {
    // per patch and RDB, either geometry: layers 1-4 8064 MFMAs (28 stages of 288 / 42 of 192), layer 5 6912 (12 of 576 / 18 of 384)
    return (double)grid * P * rdbs * (8064.0 + 6912.0) * 32768.0;
}

}  // namespace s2sr
