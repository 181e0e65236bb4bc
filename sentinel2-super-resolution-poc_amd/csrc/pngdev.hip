// PNG encoding of a tile level ON THE DEVICE (the XYZ pyramid of a job is 12.8k RGBA tiles = 3.3 GB of pixels; fetching them to
// deflate them on 16 host CPUs was the whole cost of the stage: 0.2 s of copies + 0.3 s of encoders).  The level stays where the
// pyramid kernels left it; what crosses PCIe is the compressed stream (~70 KB per tile instead of 256 KB).
//
//   stats kernel   one workgroup per tile, one WAVE per row (lane l owns bytes 16 l .. 16 l + 15 of the row's stream: filter byte 1
//                  + 1024 Sub-filtered bytes): distance-1 runs inside the row become matches, the rest literals (the host
//                  encoder's token alphabet, png_internal.h) -- into a per-tile token histogram (LDS, one table per wave), the
//                  row's Adler-32 partial sums and the tile's "any alpha" flag.
//   host           per tile: Huffman code + block header from the histogram (build_block_code, the host encoder's own), the exact
//                  compressed size (so the output buffer is laid out exactly), Adler-32 from the row sums.  Tiles that stored
//                  blocks would serve better (noise) go to the host encoder.
//   emit kernel    the rows again, twice: bits per row -> exclusive scan -> every row assembled in LDS and written at its offset.
//   host           chunk framing, CRC-32, file write, on threads.
// Integer / byte work.  The first form of the two kernels (one THREAD walks one row byte by byte: walk_row, *_kernel without
// "wave") is kept behind S2SR_PNG_ROW_THREADS as the check: both forms must write the same bytes (tests/test_gpu_tiles.py).
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "png_internal.h"
#include "s2sr_internal.h"

namespace s2sr {

namespace {

constexpr int kRow = 1024;          // bytes of pixels per tile row (256 RGBA pixels)
constexpr int kRows = 256;

// Walks one row's stream and hands every token to `tok(t)` and every stream byte to `byte(b)`.
template <class TokFn, class ByteFn>
__device__ __forceinline__ void walk_row(const uint8_t* __restrict__ row, TokFn&& tok, ByteFn&& byte) {
    uint32_t prev = 1;              // the filter-type byte: Sub
    int run = 0;                    // bytes equal to `prev` seen behind it and not yet emitted
    tok(1u);
    byte(1u);
    uint32_t left = 0;              // the pixel to the left (Sub, bpp = 4)
    for (int q = 0; q < kRow / 16; ++q) {
        const uint4 v = ((const uint4*)row)[q];
        const uint32_t px[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t a = px[k];
            const uint32_t d = ((a | 0x80808080u) - (left & 0x7F7F7F7Fu)) ^ ((a ^ ~left) & 0x80808080u);   // four byte-wise a - left
            left = a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t b = (d >> (8 * j)) & 0xFFu;
                byte(b);
                if (b == prev) {
                    if (++run == 258) { tok(256u + 255u); run = 0; }
                } else {
                    if (run >= 3) tok(256u + (uint32_t)(run - 3));
                    else for (int r = 0; r < run; ++r) tok(prev);
                    run = 0;
                    tok(b);
                    prev = b;
                }
            }
        }
    }
    if (run >= 3) tok(256u + (uint32_t)(run - 3));
    else for (int r = 0; r < run; ++r) tok(prev);
}

__global__ void __launch_bounds__(256) png_tile_stats_kernel(const uint8_t* __restrict__ tiles, int ntiles, uint32_t* __restrict__ hist,
                                                             uint32_t* __restrict__ adler, uint32_t* __restrict__ flags) {
    __shared__ uint32_t h[4][512];
    __shared__ uint32_t any_alpha;
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    for (int i = threadIdx.x; i < 4 * 512; i += 256) (&h[0][0])[i] = 0;
    if (threadIdx.x == 0) any_alpha = 0;
    __syncthreads();
    const uint8_t* row = tiles + ((size_t)t * kRows + threadIdx.x) * kRow;
    uint32_t* hw = h[threadIdx.x >> 6];
    uint32_t a = 0, b = 0, last = 0xFFFFFFFFu, cnt = 0, alpha = 0;
    {   // alpha of the row (byte 3 of every pixel), before the filter
        for (int q = 0; q < kRow / 16; ++q) {
            const uint4 v = ((const uint4*)row)[q];
            alpha |= (v.x | v.y | v.z | v.w) >> 24;
        }
    }
    walk_row(row,
             [&](uint32_t tk) {
                 if (tk == last) { ++cnt; return; }
                 if (cnt) atomicAdd(&hw[last], cnt);
                 last = tk;
                 cnt = 1;
             },
             [&](uint32_t by) { a += by; b += a; });
    if (cnt) atomicAdd(&hw[last], cnt);
    if (alpha) atomicOr(&any_alpha, 1u);
    adler[((size_t)t * kRows + threadIdx.x) * 2] = a;
    adler[((size_t)t * kRows + threadIdx.x) * 2 + 1] = b;
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) hist[(size_t)t * 512 + i] = h[0][i] + h[1][i] + h[2][i] + h[3][i];
    if (threadIdx.x == 0) flags[t] = any_alpha;
}

// ---- one wave per row --------------------------------------------------------------------------
// The same token stream as walk_row, computed without a serial walk: lane l owns stream bytes 16 l .. 16 l + 15 of the row (one
// coalesced 1-KB load per wave).  e[p] = "byte p equals byte p - 1" as a 16-bit mask per lane; a maximal stretch of e = 1 is a
// run of R bytes behind a literal, cut into chunks of 258 from its start: a chunk of c >= 3 bytes is ONE match token at its
// first byte, a last chunk of 1 or 2 bytes stays literals.  What a lane needs from its neighbours -- how many e = 1 bytes lie
// directly in front of its first byte (P) and directly behind its last one (S) -- comes from one ballot of the "all 16 equal"
// lanes and two shuffles.  No branch depends on the data.
struct RowTokens {
    uint32_t tok[16];       // token at each of the lane's 16 positions (valid where bit i of `valid` is set)
    uint32_t valid;
    uint32_t f[4];          // the 16 filtered bytes
};

__device__ __forceinline__ uint32_t sub_bytes(uint32_t a, uint32_t b) {   // four byte-wise a - b
    return ((a | 0x80808080u) - (b & 0x7F7F7F7Fu)) ^ ((a ^ ~b) & 0x80808080u);
}
__device__ __forceinline__ uint32_t zero_byte_bits(uint32_t x) {         // bit k set where byte k of x is zero
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);     // 0x80 in every zero byte
    return ((((z >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}

__device__ __forceinline__ void row_tokens(const uint4 v, int lane, RowTokens& rt, uint32_t& alpha_bits) {   // v: the lane's 16 bytes of the row
    alpha_bits = (v.x | v.y | v.z | v.w) >> 24;
    uint32_t left = __shfl_up(v.w, 1);
    if (lane == 0) left = 0;
    rt.f[0] = sub_bytes(v.x, left);
    rt.f[1] = sub_bytes(v.y, v.x);
    rt.f[2] = sub_bytes(v.z, v.y);
    rt.f[3] = sub_bytes(v.w, v.z);
    uint32_t prevb = __shfl_up(rt.f[3] >> 24, 1);
    if (lane == 0) prevb = 1;                                    // the filter-type byte in front of the row
    const uint32_t s0 = (rt.f[0] << 8) | prevb, s1 = (rt.f[1] << 8) | (rt.f[0] >> 24), s2 = (rt.f[2] << 8) | (rt.f[1] >> 24),
                   s3 = (rt.f[3] << 8) | (rt.f[2] >> 24);
    const uint32_t E = zero_byte_bits(rt.f[0] ^ s0) | (zero_byte_bits(rt.f[1] ^ s1) << 4) | (zero_byte_bits(rt.f[2] ^ s2) << 8) |
                       (zero_byte_bits(rt.f[3] ^ s3) << 12);
    const bool full = E == 0xFFFFu;
    const int lead = full ? 16 : __builtin_ctz(~E);                         // e = 1 bytes from the lane's first byte on
    const int trail = full ? 16 : __builtin_clz((~E & 0xFFFFu) << 16);      // ... up to its last byte
    const unsigned long long F = __ballot(full);
    const unsigned long long below = ~F & ((1ull << lane) - 1ull);
    const int j = below ? 63 - __builtin_clzll(below) : 0;
    const int trail_j = __shfl(trail, j);
    const int P = below ? 16 * (lane - 1 - j) + trail_j : 16 * lane;
    const unsigned long long above = lane == 63 ? 0ull : (~F & ~((2ull << lane) - 1ull));
    const int k = above ? __builtin_ctzll(above) : 63;
    const int lead_k = __shfl(lead, k);
    const int S = above ? 16 * (k - lane - 1) + lead_k : 16 * (63 - lane);
    rt.valid = 0;
    int q = 0, R = 0;                                            // position inside the current run, its length
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t fb = (rt.f[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const bool e = (E >> i) & 1u;
        const bool starts = e && (i == 0 || !((E >> (i - 1)) & 1u));
        if (i == 0) {                                            // a run that comes in from the left
            q = P;
            R = P + lead + (lead == 16 ? S : 0);
        } else if (starts) {
            const uint32_t rest = ~(E >> i) & (0xFFFFu >> i);    // first e = 0 behind i, inside the lane
            const int len = rest ? __builtin_ctz(rest) : 16 - i;
            q = 0;
            R = len + (len == 16 - i ? S : 0);
        } else {
            ++q;
        }
        const int qm = q < 258 ? q : q < 516 ? q - 258 : q < 774 ? q - 516 : q - 774;     // q mod 258 (q < 1032)
        const int c = min(258, R - (q - qm));                    // length of the chunk this byte is in
        const bool is_match = e && c >= 3 && qm == 0;
        const bool is_lit = !e || c < 3;
        rt.tok[i] = is_match ? 256u + (uint32_t)(c - 3) : fb;
        rt.valid |= (uint32_t)(is_match || is_lit) << i;
    }
}

__global__ void __launch_bounds__(256) png_tile_stats_wave_kernel(const uint8_t* __restrict__ tiles, int ntiles, uint32_t* __restrict__ hist,
                                                                  uint32_t* __restrict__ adler, uint32_t* __restrict__ flags) {
    __shared__ uint32_t h[4][512];
    __shared__ uint32_t any_alpha;
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    for (int i = threadIdx.x; i < 4 * 512; i += 256) (&h[0][0])[i] = 0;
    if (threadIdx.x == 0) any_alpha = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* hw = h[wave];
    uint32_t alpha = 0;
    const uint4* rows = (const uint4*)(tiles + (size_t)t * kRows * kRow) + lane;      // row r of this lane: rows[r * 64]
    uint4 vn = rows[(size_t)wave * 64];
    for (int r = wave; r < kRows; r += 4) {
        const uint4 v = vn;
        if (r + 4 < kRows) vn = rows[(size_t)(r + 4) * 64];     // the next row is on its way while this one is worked on
        RowTokens rt;
        uint32_t ab;
        row_tokens(v, lane, rt, ab);
        alpha |= ab;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((rt.valid >> i) & 1u) atomicAdd(&hw[rt.tok[i]], 1u);
        // Adler partial sums of the row's 1025 stream bytes (filter byte first): A = sum x_i, B = sum (1025 - i) x_i
        uint32_t A = 0, B = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t fb = (rt.f[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            A += fb;
            B += fb * (uint32_t)(kRow - (16 * lane + i));         // stream index 1 + 16 lane + i
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { A += __shfl_xor(A, o); B += __shfl_xor(B, o); }
        if (lane == 0) {
            adler[((size_t)t * kRows + r) * 2] = A + 1u;
            adler[((size_t)t * kRows + r) * 2 + 1] = B + (uint32_t)(kRow + 1);
        }
    }
    if (lane == 0) atomicAdd(&hw[1], (uint32_t)(kRows / 4));       // 64 rows per wave, one filter-byte literal each
    if (alpha) atomicOr(&any_alpha, 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) hist[(size_t)t * 512 + i] = h[0][i] + h[1][i] + h[2][i] + h[3][i];
    if (threadIdx.x == 0) flags[t] = any_alpha;
}

struct TileMeta {
    uint64_t out_word;      // first 32-bit word of the tile's region in the output buffer
    uint32_t header_bits;   // block header bits in front of the rows' bits
    uint32_t skip;          // nothing to emit for this tile
};

__global__ void __launch_bounds__(256) png_tile_emit_kernel(const uint8_t* __restrict__ tiles, int ntiles, const TileMeta* __restrict__ meta,
                                                            const uint32_t* __restrict__ tbs, const uint32_t* __restrict__ hdrs,
                                                            uint32_t* __restrict__ out) {
    __shared__ uint32_t tb[512];
    __shared__ uint32_t row_bits[kRows];
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    const TileMeta m = meta[t];
    if (m.skip) return;
    for (int i = threadIdx.x; i < 512; i += 256) tb[i] = tbs[(size_t)t * 512 + i];
    __syncthreads();
    const uint8_t* row = tiles + ((size_t)t * kRows + threadIdx.x) * kRow;
    uint32_t bits = 0;
    walk_row(row, [&](uint32_t tk) { bits += tb[tk] >> 24; }, [](uint32_t) {});
    row_bits[threadIdx.x] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {                          // exclusive scan of 256 values: not worth more than this
        uint32_t acc = m.header_bits;
        for (int r = 0; r < kRows; ++r) { const uint32_t v = row_bits[r]; row_bits[r] = acc; acc += v; }
    }
    __syncthreads();
    uint32_t* o = out + m.out_word;
    // block header: whole words, the last one partial (zero above header_bits)
    for (uint32_t w = threadIdx.x; w * 32 < m.header_bits; w += 256) atomicOr(&o[w], hdrs[(size_t)t * 160 + w]);
    // this row's bits
    uint32_t at = row_bits[threadIdx.x];
    uint32_t w = at >> 5;
    uint64_t acc = 0;
    int n = (int)(at & 31u);
    walk_row(row,
             [&](uint32_t tk) {
                 const uint32_t e = tb[tk];
                 acc |= (uint64_t)(e & 0xFFFFFFu) << n;
                 n += (int)(e >> 24);
                 if (n >= 32) { atomicOr(&o[w++], (uint32_t)acc); acc >>= 32; n -= 32; }
             },
             [](uint32_t) {});
    if (n > 0) atomicOr(&o[w], (uint32_t)acc);
}

__global__ void __launch_bounds__(256) png_tile_emit_wave_kernel(const uint8_t* __restrict__ tiles, int ntiles, const TileMeta* __restrict__ meta,
                                                                 const uint32_t* __restrict__ tbs, const uint32_t* __restrict__ hdrs,
                                                                 uint32_t* __restrict__ out) {
    __shared__ uint32_t tb[512];
    __shared__ uint32_t row_bits[kRows];
    __shared__ uint32_t rowbuf[4][512];
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    const TileMeta m = meta[t];
    if (m.skip) return;
    for (int i = threadIdx.x; i < 512; i += 256) tb[i] = tbs[(size_t)t * 512 + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lit1_bits = tb[1] >> 24;                      // the filter byte's literal in front of every row
    const uint4* rows = (const uint4*)(tiles + (size_t)t * kRows * kRow) + lane;      // row r of this lane: rows[r * 64]
    uint4 vn = rows[(size_t)wave * 64];
    for (int r = wave; r < kRows; r += 4) {
        const uint4 v = vn;
        if (r + 4 < kRows) vn = rows[(size_t)(r + 4) * 64];
        RowTokens rt;
        uint32_t ab;
        row_tokens(v, lane, rt, ab);
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) bits += ((rt.valid >> i) & 1u) ? (tb[rt.tok[i]] >> 24) : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bits += __shfl_xor(bits, o);
        if (lane == 0) row_bits[r] = bits + lit1_bits;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                          // exclusive scan of 256 values
        uint32_t acc = m.header_bits;
        for (int r = 0; r < kRows; ++r) { const uint32_t v = row_bits[r]; row_bits[r] = acc; acc += v; }
    }
    __syncthreads();
    uint32_t* o = out + m.out_word;
    for (uint32_t w = threadIdx.x; w * 32 < m.header_bits; w += 256) atomicOr(&o[w], hdrs[(size_t)t * 160 + w]);
    vn = rows[(size_t)wave * 64];
    for (int r = wave; r < kRows; r += 4) {
        const uint4 v = vn;
        if (r + 4 < kRows) vn = rows[(size_t)(r + 4) * 64];
        RowTokens rt;
        uint32_t ab;
        row_tokens(v, lane, rt, ab);
        // the lane's tokens as one bit string (at most 16 x 21 bits, + the filter byte's literal in lane 0), assembled in the
        // wave's LDS row buffer: a row is at most 1025 x 15 bits (all literals at the longest code) + 31 bits of offset = 482 words
        uint32_t bits = lane == 0 ? lit1_bits : 0u;
#pragma unroll
        for (int i = 0; i < 16; ++i) bits += ((rt.valid >> i) & 1u) ? (tb[rt.tok[i]] >> 24) : 0u;
        uint32_t incl = bits;                                    // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const uint32_t row_at = row_bits[r];                     // bit offset of the row in the tile's stream
        const uint32_t total = __shfl(incl, 63);
        const uint32_t nwords = ((row_at & 31u) + total + 31u) >> 5;
        uint32_t* rb = rowbuf[wave];
        for (uint32_t w = lane; w < nwords; w += 64) rb[w] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t at = (row_at & 31u) + (incl - bits);
        uint32_t w = at >> 5;
        uint64_t acc = 0;
        int n = (int)(at & 31u);
        if (lane == 0) {
            acc |= (uint64_t)(tb[1] & 0xFFFFFFu) << n;
            n += (int)lit1_bits;
            if (n >= 32) { atomicOr(&rb[w++], (uint32_t)acc); acc >>= 32; n -= 32; }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if ((rt.valid >> i) & 1u) {
                const uint32_t e = tb[rt.tok[i]];
                acc |= (uint64_t)(e & 0xFFFFFFu) << n;
                n += (int)(e >> 24);
                if (n >= 32) { atomicOr(&rb[w++], (uint32_t)acc); acc >>= 32; n -= 32; }
            }
        }
        if (n > 0 && acc) atomicOr(&rb[w], (uint32_t)acc);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // out: the first and the last word are shared with the neighbouring rows, the ones between belong to this row alone
        uint32_t* og = o + (row_at >> 5);
        for (uint32_t k = lane; k < nwords; k += 64) {
            const uint32_t val = rb[k];
            if (k == 0 || k + 1 == nwords) { if (val) atomicOr(&og[k], val); }
            else og[k] = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// CPUs this process may really use: its affinity mask, capped by the cgroup quota (cpu.max) -- what s2sr/hostpool.py counts; a
// container may see 256 logical CPUs and be allowed 16 of them, and a pool of 32 on such a box only adds context switches
int usable_cpus() {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 4;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        long long per = 0;
        if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
            const long long quota = atoll(q);
            if (quota > 0) {
                const int c = (int)((quota + per - 1) / per);
                if (c >= 1 && c < n) n = c;
            }
        }
        fclose(f);
    }
    return n;
}

int host_threads() {
    int n = usable_cpus();
    if (const char* e = getenv("S2SR_HOST_THREADS")) { const int v = atoi(e); if (v > 0 && v < n) n = v; }
    else if (n > 32) n = 32;
    return n;
}

// Host threads that stay: a pyramid makes ~40 parallel_for calls, and a thread per call and worker was 1200 thread starts per
// pyramid -- and, on a 256-thread host, as many malloc arenas as glibc allows for short-lived threads to park freed memory in
// (the 15-minute soak's host RSS crept 0.5 MB/s).  One parallel_for runs at a time (callers from several handles take turns); the
// calling thread works too.
class HostPool {
    std::vector<std::thread> threads_;
    std::mutex mu_, serial_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(int)>* body_ = nullptr;
    int n_ = 0, active_ = 0;
    std::atomic<int> next_{0};
    uint64_t generation_ = 0;
    bool stop_ = false;

    std::atomic<bool> threw_{false};
    const pid_t pid_ = getpid();          // the process the threads live in: a forked child has this object but none of them

    void drain() {
        for (;;) {
            const int i0 = next_.fetch_add(16);
            if (i0 >= n_) return;
            for (int i = i0; i < std::min(n_, i0 + 16); ++i) {
                try {
                    (*body_)(i);
                } catch (...) {           // (bad_alloc from a buffer resize: reported by run(), never std::terminate in a worker)
                    threw_.store(true);
                }
            }
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || generation_ != seen; });
            if (stop_) return;
            seen = generation_;
            lk.unlock();
            drain();
            lk.lock();
            if (--active_ == 0) cv_done_.notify_one();
        }
    }

public:
    explicit HostPool(int nthreads) {
        for (int k = 1; k < nthreads; ++k) threads_.emplace_back([this] { worker(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // false: some body(i) threw (every other index still ran)
    bool run(int n, const std::function<void(int)>& body) {
        if (n <= 0) return true;
        if (n <= 16 || threads_.empty() || getpid() != pid_) {      // small jobs, and a forked child (its pool threads do not exist), run inline
            bool ok = true;
            for (int i = 0; i < n; ++i) {
                try {
                    body(i);
                } catch (...) {
                    ok = false;
                }
            }
            return ok;
        }
        std::lock_guard<std::mutex> one_at_a_time(serial_);
        threw_.store(false);
        {
            std::lock_guard<std::mutex> lk(mu_);
            body_ = &body;
            n_ = n;
            next_.store(0);
            active_ = (int)threads_.size();
            ++generation_;
        }
        cv_work_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return active_ == 0; });
        body_ = nullptr;
        return !threw_.load();
    }
};

HostPool& host_pool() {
    static HostPool pool(host_threads());          // made at first use; in a child forked afterwards run() works inline (no threads there)
    return pool;
}

template <class F> bool parallel_for(int n, F&& body) {      // body(i) for i in [0, n), dynamic chunks of 16; false: a body threw
    const std::function<void(int)> f(std::forward<F>(body));
    return host_pool().run(n, f);
}

}  // namespace

hipError_t launch_png_tile_stats(const uint8_t* d_tiles, int ntiles, uint32_t* d_hist, uint32_t* d_adler, uint32_t* d_flags, bool row_threads,
                                 hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    if (row_threads) hipLaunchKernelGGL(png_tile_stats_kernel, dim3(ntiles), dim3(256), 0, st, d_tiles, ntiles, d_hist, d_adler, d_flags);
    else hipLaunchKernelGGL(png_tile_stats_wave_kernel, dim3(ntiles), dim3(256), 0, st, d_tiles, ntiles, d_hist, d_adler, d_flags);
    return hipGetLastError();
}

hipError_t launch_png_tile_emit(const uint8_t* d_tiles, int ntiles, const void* d_meta, const uint32_t* d_tb, const uint32_t* d_hdr,
                                uint32_t* d_out, bool row_threads, hipStream_t st) {
    if (ntiles <= 0) return hipSuccess;
    if (row_threads)
        hipLaunchKernelGGL(png_tile_emit_kernel, dim3(ntiles), dim3(256), 0, st, d_tiles, ntiles, (const TileMeta*)d_meta, d_tb, d_hdr, d_out);
    else
        hipLaunchKernelGGL(png_tile_emit_wave_kernel, dim3(ntiles), dim3(256), 0, st, d_tiles, ntiles, (const TileMeta*)d_meta, d_tb, d_hdr,
                           d_out);
    return hipGetLastError();
}

// ---- host side of the plan ---------------------------------------------------------------------
// From the stats of `n` tiles: per tile the Huffman table / header / meta the emit kernel reads, the Adler-32 of the stream and the
// size of its deflate bytes.  mode[t]: 0 = nothing to write (no path, or fully transparent), 1 = device stream, 2 = host encoder
// (stored blocks would be smaller than this tile's Huffman block).  Returns the number of 32-bit words the output buffer needs.
size_t png_plan_bytes(int n) { return (size_t)n * (512 * 4 + 160 * 4 + sizeof(TileMeta)); }

size_t png_plan_tiles(int n, const uint32_t* hist, const uint32_t* adler_rows, const uint32_t* flags, const char* const* paths,
                      bool skip_transparent, bool force_host, PngTilePlan* plan) {
    plan->mode.assign(n, 0);
    plan->adler.assign(n, 0);
    plan->deflate_bytes.assign(n, 0);
    plan->eob.assign(n, 0);
    plan->eob_at.assign(n, 0);
    // the upload block: token tables and headers of tiles that emit nothing are never read (TileMeta::skip), so the block is not
    // cleared as a whole (a page-locked arena that is reused holds the previous level's bytes); emitting tiles clear their header slot
    plan->upload_bytes = png_plan_bytes(n);
    uint8_t* block = (uint8_t*)plan->arena;
    if (!block || plan->arena_bytes < plan->upload_bytes) {
        plan->own.resize(plan->upload_bytes);
        block = plan->own.data();
    }
    plan->tb = (uint32_t*)block;
    plan->hdr = plan->tb + (size_t)n * 512;
    plan->meta = (uint8_t*)(plan->hdr + (size_t)n * 160);
    std::vector<uint32_t> words(n, 0);
    plan->failed = !parallel_for(n, [&](int t) {
        TileMeta* m = (TileMeta*)plan->meta + t;
        m->out_word = 0;
        m->header_bits = 0;
        m->skip = 1;
        if (!paths[t] || (skip_transparent && !flags[t])) return;
        png::BlockCode bc;
        png::build_block_code(hist + (size_t)t * 512, true, &bc);
        const uint64_t bits = bc.header_bits + bc.body_bits;
        const size_t nraw = (size_t)kRows * (kRow + 1);
        if (bits >= 8 * (uint64_t)nraw + 40 * ((nraw + 65534) / 65535) || bc.header_bits > 160 * 32 || force_host) { plan->mode[t] = 2; return; }
        plan->mode[t] = 1;
        m->skip = 0;
        m->header_bits = bc.header_bits;
        memcpy(&plan->tb[(size_t)t * 512], bc.tb, sizeof bc.tb);
        memset(&plan->hdr[(size_t)t * 160], 0, 160 * 4);        // the kernel takes whole words: nothing stale behind the header's last bit
        memcpy(&plan->hdr[(size_t)t * 160], bc.header, (bc.header_bits + 7) / 8);
        // the end-of-block code is the last thing in the stream: the host ORs it in after the copy back (its offset is known here)
        plan->eob[t] = bc.eob;
        plan->eob_at[t] = bits - (bc.eob >> 24);
        uint32_t a = 1, b = 0;      // Adler-32 over the 256 rows from their partial sums (a row of len bytes adds len * a + B to b)
        for (int r = 0; r < kRows; ++r) {
            const uint32_t A = adler_rows[((size_t)t * kRows + r) * 2], B = adler_rows[((size_t)t * kRows + r) * 2 + 1];
            b = (uint32_t)((b + (uint64_t)(kRow + 1) * a + B) % 65521u);
            a = (uint32_t)((a + (uint64_t)A) % 65521u);
        }
        plan->adler[t] = (b << 16) | a;
        plan->deflate_bytes[t] = (uint32_t)((bits + 7) / 8);
        words[t] = (uint32_t)((bits + 31) / 32 + 1);
    });
    size_t total = 0;
    for (int t = 0; t < n; ++t) {
        TileMeta* m = (TileMeta*)plan->meta + t;
        m->out_word = total;
        total += words[t];
    }
    plan->out_word.resize(n);
    for (int t = 0; t < n; ++t) plan->out_word[t] = ((const TileMeta*)plan->meta + t)->out_word;
    return total;
}

// One tile's file from its deflate bytes (end-of-block code not yet in): PNG signature, IHDR 256 x 256 RGBA, one IDAT, IEND.
bool png_write_tile_file(const char* path, const uint32_t* words, uint32_t deflate_bytes, uint32_t eob, uint64_t eob_at, uint32_t adler,
                         std::vector<uint8_t>& buf) {
    struct Head {
        uint8_t b[33];
        Head() {
            static const uint8_t h[29] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n', 0, 0, 0, 13, 'I', 'H', 'D', 'R', 0, 0, 1, 0, 0, 0, 1, 0, 8, 6, 0, 0, 0};
            memcpy(b, h, 29);
            const uint32_t c = png::crc32_update(0, b + 12, 17);
            b[29] = (uint8_t)(c >> 24); b[30] = (uint8_t)(c >> 16); b[31] = (uint8_t)(c >> 8); b[32] = (uint8_t)c;
        }
    };
    static const Head H;
    // The file goes out as three pieces (writev): the constant head + IDAT header, the deflate bytes straight from the staging
    // buffer the DMA filled, and a small tail (the stream's last bytes with the end-of-block code merged in, Adler-32, the chunk's
    // CRC, IEND).  r04 assembled every file in a per-thread buffer first: a 30-KB copy per tile in front of a table-walk CRC.
    (void)buf;
    const size_t idat = 2 + (size_t)deflate_bytes + 4;
    uint8_t head[43];
    memcpy(head, H.b, 33);
    head[33] = (uint8_t)(idat >> 24); head[34] = (uint8_t)(idat >> 16); head[35] = (uint8_t)(idat >> 8); head[36] = (uint8_t)idat;
    memcpy(head + 37, "IDAT", 4);
    head[41] = 0x78; head[42] = 0x01;
    const size_t ntail = deflate_bytes < 16 ? deflate_bytes : 16;       // the end-of-block code lands in the last bytes (<= 4 of them)
    const size_t nbody = deflate_bytes - ntail;
    const uint8_t* src = (const uint8_t*)words;
    uint8_t tail[16 + 20];
    memcpy(tail, src + nbody, ntail);
    {   // the end-of-block code, LSB first at bit eob_at
        uint64_t v = (uint64_t)(eob & 0xFFFFFFu) << (eob_at & 7);
        for (size_t k = eob_at >> 3; v; ++k, v >>= 8) {
            if (k < nbody || k >= deflate_bytes) return false;          // (cannot happen: the code ends inside the stream's last bytes)
            tail[k - nbody] |= (uint8_t)v;
        }
    }
    uint8_t* t = tail + ntail;
    t[0] = (uint8_t)(adler >> 24); t[1] = (uint8_t)(adler >> 16); t[2] = (uint8_t)(adler >> 8); t[3] = (uint8_t)adler;
    uint32_t crc = png::crc32_update(0, head + 37, 6);
    crc = png::crc32_update(crc, src, nbody);
    crc = png::crc32_update(crc, tail, ntail + 4);
    t[4] = (uint8_t)(crc >> 24); t[5] = (uint8_t)(crc >> 16); t[6] = (uint8_t)(crc >> 8); t[7] = (uint8_t)crc;
    static const uint8_t iend[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xae, 0x42, 0x60, 0x82};
    memcpy(t + 8, iend, 12);
    const png::Piece pieces[3] = {{head, sizeof head}, {src, nbody}, {tail, ntail + 20}};
    return png::write_file_pieces(path, pieces, 3);
}

bool png_parallel_for(int n, const std::function<void(int)>& body) { return parallel_for(n, body); }

}  // namespace s2sr
