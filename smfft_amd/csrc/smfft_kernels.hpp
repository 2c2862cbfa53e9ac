// smfft_kernels.hpp -- the benchmark kernels around the engine (reference names kept).
//
//   SMFFT_DIT_external<P>             CT/FFT-GPU-32bit.cu:534-551   global -> FFT -> global  (HBM bound)
//   SMFFT_DIT_multiple<P>             CT/FFT-GPU-32bit.cu:553-572   NREUSES FFTs in LDS per load/store
//   FFT_GPU_external/multiple<P>      ST/FFT-GPU-32bit-Stockham.cu:243-278
//   FFT_GPU_R2C_C2R_external/multiple<P,D>  RC/FFT-GPU-32bit-Stockham.cu:349-384
//
// The EXTERNAL kernels use 256-thread workgroups that own a tile of 4096 float2 (= 4096/N FFTs) and
// P::fft_sm_required = 4352 float2 of LDS (34 KiB -> 4 workgroups = 16 waves per CU), on the float2 engine
// (smfft_engine.hpp).  The IN-LDS (`multiple`) kernels use compact workgroups (one wave per 1024 elements, one FFT per
// workgroup above) on the planar engine (smfft_planar.hpp; N = 32 and N = 64 without reorder: the lane engines of smfft_engine.hpp).  For
// N >= 256 the external kernels do not stage through LDS on the way in or out: pass 1 loads
// straight from global memory into registers (each wave instruction reads 512 contiguous bytes
// for N >= 1024) and the last pass stores straight from registers, so the only LDS traffic is the
// exchanges; N <= 128 (an FFT is at most 8 threads) moves wave-sized chunks through the LDS regions
// with 512-byte instructions.  Between a tile's loads and its stores sits a run-time-switched rate
// limiter (vmem_throttle below).  Grids are grid-strided over tiles so a capped ("persistent") grid
// keeps twiddles in registers.
#pragma once
#include <type_traits>

#include "smfft/smfft_device_functions.hpp"
#include "smfft/smfft_planar.hpp"

namespace smfft {

// external kernels stage through LDS with wave-coalesced global access up to this length
constexpr int kStagedMaxN = 128;

// ------------------------------------------------------------------------------------------------
// C2C, external: out[f] = FFT(in[f]) for f < nFFTs.
// ------------------------------------------------------------------------------------------------
// Wave-coalesced staging for N <= 128 (external kernels only).  With 16 elements per thread an FFT
// of N <= 128 has 8 or fewer threads, so direct register I/O would touch N/16 * 8 = 16..64
// contiguous bytes per FFT per instruction.  Instead each wave moves its own 1024-element chunk
// (its 1024/N FFTs) with 512-byte-contiguous instructions through the FFTs' LDS regions.
// `full` (wave-uniform) = every FFT of the chunk is inside the batch: the loads are then issued
// back to back with no per-element predicate.  (A per-element "load or zero" select makes hipcc
// branch around every load and wait vmcnt(0) after each one: 16 serialised HBM round trips.)
template <int N>
__device__ __forceinline__ void wave_chunk_to_lds(const float2* __restrict__ gwave, const float2* __restrict__ gsafe, float2* swave, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    const int lane = threadIdx.x & 63;
    float2 v[16];
    if (first_fft + 1024 / N <= limit_fft) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = gload(gwave + lane + 64 * c);
    } else {
        // ragged tail: out-of-range lanes read a safe address and the value is replaced by zero
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int e = lane + 64 * c;
            const bool ok = first_fft + e / N < limit_fft;
            float2 t = gload(ok ? gwave + e : gsafe);
            v[c] = ok ? t : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = lane + 64 * c;
        swave[(e / N) * G::SF + (e % N)] = v[c];
    }
}
template <int N>
__device__ __forceinline__ void lds_to_wave_chunk(float2* __restrict__ gwave, const float2* swave, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    const int lane = threadIdx.x & 63;
    float2 v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int e = lane + 64 * c;
        v[c] = swave[(e / N) * G::SF + (e % N)];
    }
    if (first_fft + 1024 / N <= limit_fft) {
#pragma unroll
        for (int c = 0; c < 16; ++c) gstore(gwave + lane + 64 * c, v[c]);
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int e = lane + 64 * c;
            if (first_fft + e / N < limit_fft) gstore(gwave + e, v[c]);
        }
    }
}

// Pacing of the HBM-bound external kernels: a RATE LIMITER AT THE SOURCE, chosen per launch.
// After a wave's global loads have arrived, K serialised FLAT loads of 8 B/lane from the wave's own LDS rows (results
// discarded; no LDS writes, the data registers untouched; each waits for the previous one) occupy the CU's vector-memory
// address path between the wave's loads and its stores without going to memory.  What the counters say it does
// (profiles/r02_placement_pmc.json, study_copy vs study_copy_paced, same buffers): L2 -> DRAM credit stalls fall 8-10 x
// (TCC_EA0_WRREQ_DRAM_CREDIT_STALL 1.8e7 -> 1.7e6, RDREQ 6.1e6 -> 0.8e6 per launch), TCC_TAG_STALL 6 x, the write queue
// depth (TCC_EA0_WRREQ_LEVEL) 25 %, the L1 -> L2 read / write latencies 15 % / 27 %: the bursts each wave sends are
// spread out, the queues in front of the DRAM stop overflowing.  How much of it PAYS depends on the write target, so K
// is a kernel argument the host API decides per launch from the output pointer (smfft_api.hip, pacing_for; sweeps on
// the same buffers: tools/pacing_sweep.py, profiles/r02_pacing_sweep_*.txt): into ordinary memory (a caller's plain
// hipMalloc buffer: writes 5.6 TB/s at best) K = 12 for N <= 1024 and 8 above is worth 2-8 %; into the mixed memory
// smfft_malloc_pair builds its outputs from (6.9 TB/s) K = 4 is worth 0.3-1.6 % and more costs.  SMFFT_PACING=K forces it.
// (Round 1 had two more forms that did the same thing through `volatile` generic-pointer reads of LDS -- i.e. through
//  how hipcc happens to lower them; they measured equal to this explicit one, profiles/r02_ab_pacing_plain.txt, and are gone.)
__device__ __forceinline__ void vmem_throttle(const float2* rows, float2 (&r)[16], int k) {
#pragma unroll
    for (int c = 0; c < 16; ++c) asm volatile("" : "+v"(r[c].x), "+v"(r[c].y));   // all global loads have arrived
    for (int c = 0; c < k; ++c) {                                                  // k is a kernel argument: a scalar loop
        const float2* q = rows + (threadIdx.x & 63) + 64 * (c & 15);
        v2f d;
        asm volatile("flat_load_dwordx2 %0, %1\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=v"(d) : "v"(q) : "memory");
    }
}

template <int N, int DIR, int REORDER>
__device__ __forceinline__ void c2c_external_body(const float2* __restrict__ d_input, float2* __restrict__ d_output, int nFFTs, int pace, float2* s) {
    using G = Geometry<N>;
    constexpr bool kStaged = (N <= kStagedMaxN);
    constexpr int kFftsPerWave = (N <= 1024) ? 1024 / N : 1;
    Engine<N, DIR, REORDER> eng;
    eng.init(threadIdx.x);
    float2* sf = s + eng.fft * G::SF;
    const int wave = threadIdx.x >> 6;
    float2* swave = s + wave * kFftsPerWave * G::SF;
    const int ntiles = (nFFTs + G::kFftsPerBlock - 1) / G::kFftsPerBlock;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f = (long)tile * G::kFftsPerBlock + eng.fft;
        const bool active = f < nFFTs;
        float2 r[16];
        if constexpr (kStaged) {
            const long fw = (long)tile * G::kFftsPerBlock + wave * kFftsPerWave;
            wave_chunk_to_lds<N>(d_input + fw * N, d_input, swave, fw, nFFTs);
            fft_sync<false>();
            eng.load_lds(r, sf);
            fft_sync<false>();
            if (pace) vmem_throttle(swave, r, pace);   // kernel argument: wave-uniform branch
            eng.transform(r, sf);
            fft_sync<false>();
            eng.store_lds(r, sf);
            fft_sync<false>();
            lds_to_wave_chunk<N>(d_output + fw * N, swave, fw, nFFTs);
            fft_sync<false>();
        } else {
            eng.load_global(r, d_input + (active ? f : 0) * N);
            if (G::kMultiWave) __syncthreads();   // the previous tile's exchange reads are complete
            if (pace) vmem_throttle(s + (threadIdx.x >> 6) * 1088, r, pace);
            eng.transform(r, sf);
            eng.store_global(r, d_output + f * N, active);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// In-LDS (`multiple`) path.  COMPACT workgroups: the smallest workgroup that holds whole FFTs -- one wave owning
// 1024 elements (1024/N FFTs) for N <= 1024, N/16 threads owning one FFT above -- so that (a) an N <= 1024 kernel
// has no workgroup barrier anywhere (a wave's DS operations execute in order) and (b) the README batches, which
// are 5242 waves' worth of FFT slots for every N, spread over the chip's wave slots one wave at a time instead
// of in 1311 four-wave blocks on 1024 block slots (the tail quantisation round 1 measured at 25 %).
// tile <-> LDS copies: element e of the tile belongs to FFT e/N, position n = e%N of its region; PADDED: position
// n + (n >> kPadShift), the image the no-reorder variants keep their data in (Engine::bitrev_write / bitrev_read).
// ------------------------------------------------------------------------------------------------
// LDS index of tile element e = threadIdx.x + kCompactThreads * c, split into a per-thread base and a compile-time
// offset (so that the sixteen accesses share one address register): element e is position n = e % N of FFT e / N.
// Either N >= kCompactThreads (the thread index is the low part of n and never carries into the offset) or N divides
// kCompactThreads (the thread index contributes its own FFT number and position); the pad term splits the same way
// because kCompactThreads * c is a multiple of 2^kPadShift.
template <int N, bool PADDED>
__device__ __forceinline__ int compact_lds_base(int tid) {
    using G = Geometry<N>;
    const int t = tid % N, j = tid / N;
    return j * G::SF + t + (PADDED ? (t >> G::kPadShift) : 0);
}
template <int N, bool PADDED>
constexpr int compact_lds_offset(int c) {
    using G = Geometry<N>;
    const int e = G::kCompactThreads * c;
    return (e / N) * G::SF + (e % N) + (PADDED ? ((e % N) >> G::kPadShift) : 0);
}
template <int N, bool PADDED>
__device__ __forceinline__ void tile_to_lds(const float2* g, float2* s, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    constexpr int C = G::kCompactTile / G::kCompactThreads;
    static_assert(N >= G::kCompactThreads || G::kCompactThreads % N == 0, "see compact_lds_base");
    // (eight elements at a time: these kernels -- the lane engines' -- sit at the 128-register cap of four waves per SIMD with 30 registers
    //  of twiddles alive across the copy, and sixteen values in flight pushed addresses of the piece loop into scratch; a copy is once per
    //  hundred applications)
    constexpr int kBatch = C > 8 ? 8 : C;
    const int tid = (int)threadIdx.x;
    float2* base = s + compact_lds_base<N, PADDED>(tid);
    const bool full = first_fft + G::kCompactFfts <= limit_fft;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += kBatch) {
        float2 v[kBatch];
        if (full) {                                      // whole tile inside the batch: loads back to back, no predicate
#pragma unroll
            for (int c = 0; c < kBatch; ++c) v[c] = g[tid + G::kCompactThreads * (c0 + c)];
        } else {
#pragma unroll
            for (int c = 0; c < kBatch; ++c) {
                const int e = tid + G::kCompactThreads * (c0 + c);
                const bool ok = first_fft + e / N < limit_fft;
                const float2 t = g[ok ? e : 0];          // g[0] belongs to FFT first_fft, which exists
                v[c] = ok ? t : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int c = 0; c < kBatch; ++c) base[compact_lds_offset<N, PADDED>(c0 + c)] = v[c];
        if (kBatch < C) asm volatile("" ::: "memory");   // the batches stay apart
    }
}
template <int N, bool PADDED>
__device__ __forceinline__ void lds_to_tile(float2* g, const float2* s, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    constexpr int C = G::kCompactTile / G::kCompactThreads;
    constexpr int kBatch = C > 8 ? 8 : C;                // (as in tile_to_lds)
    int tid = (int)threadIdx.x;
    const float2* base = s + compact_lds_base<N, PADDED>(tid);
    const bool full = first_fft + G::kCompactFfts <= limit_fft;
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += kBatch) {
        float2 v[kBatch];
#pragma unroll
        for (int c = 0; c < kBatch; ++c) v[c] = base[compact_lds_offset<N, PADDED>(c0 + c)];
        if (full) {
#pragma unroll
            for (int c = 0; c < kBatch; ++c) g[tid + G::kCompactThreads * (c0 + c)] = v[c];
        } else {
#pragma unroll
            for (int c = 0; c < kBatch; ++c) {
                const int e = tid + G::kCompactThreads * (c0 + c);
                if (first_fft + e / N < limit_fft) g[e] = v[c];
            }
        }
        if (kBatch < C) asm volatile("" ::: "memory");
    }
}

// The same two copies for a tile that changes hands inside the launch (SharedTile: write-through stores, sc1 loads, 16 bytes per
// lane: elements 2 * tid, 2 * tid + 1 of every 2 * kCompactThreads)
template <int N, bool PADDED>
__device__ __forceinline__ int compact_lds_index(int e) {
    using G = Geometry<N>;
    const int n = e % N;
    return (e / N) * G::SF + n + (PADDED ? (n >> G::kPadShift) : 0);
}
template <int N, bool PADDED>
__device__ __forceinline__ void shared_tile_to_lds(const float2* g, float2* s, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    constexpr int C = G::kCompactTile / G::kCompactThreads / 2;
    const long ffts = limit_fft - first_fft < G::kCompactFfts ? limit_fft - first_fft : G::kCompactFfts;
    const SharedTile tile(g, ffts * N * 8);
    int tid = (int)threadIdx.x;
    float2 v[2 * C];
#pragma unroll
    for (int c = 0; c < C; ++c) tile.load2(2 * tid + 2 * G::kCompactThreads * c, v[2 * c], v[2 * c + 1]);
#pragma unroll
    for (int c = 0; c < 2 * C; ++c) s[compact_lds_index<N, PADDED>(2 * tid + (c & 1) + 2 * G::kCompactThreads * (c >> 1))] = v[c];
}
template <int N, bool PADDED>
__device__ __forceinline__ void lds_to_shared_tile(float2* g, const float2* s, long first_fft, long limit_fft) {
    using G = Geometry<N>;
    constexpr int C = G::kCompactTile / G::kCompactThreads / 2;
    const long ffts = limit_fft - first_fft < G::kCompactFfts ? limit_fft - first_fft : G::kCompactFfts;
    const SharedTile tile(g, ffts * N * 8);
    int tid = (int)threadIdx.x;
    // (opaque copy of the thread index HERE ONLY: this copy's addresses are then computed where they are used instead of being carried
    //  -- and, at N = 32, spilled -- across the applications.  The same in the other three copies frees forty registers and costs the
    //  N = 32 natural-order kernel 8 % -- same instruction counts, another register assignment; profiles/r06_no_scratch.txt)
    asm volatile("" : "+v"(tid));
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int e = 2 * tid + 2 * G::kCompactThreads * c;
        tile.store2(e, s[compact_lds_index<N, PADDED>(e)], s[compact_lds_index<N, PADDED>(e + 1)]);
    }
}

// ------------------------------------------------------------------------------------------------
// The work of a `multiple` launch and how it is spread over the chip.
// A CHAIN = one tile's data loaded once, transformed `nreuses` times in LDS, stored once (CT:553-572).  The README batches are
// 5242 wave-tiles at every length -- 42.9 MB of LDS images on a chip that has 41.9 MB -- so with one chain per workgroup slot the
// launch runs one full round of resident workgroups and then a second round of a few hundred lone ones at a quarter of the
// rate: a 14-28 % tail (profiles/r03_*).  BALANCED schedule (round 4): the launch is a persistent grid of about as many
// workgroups as are co-resident, and the total of ntiles * nreuses APPLICATIONS is cut into that many equal intervals
// (McNaughton's wrap-around rule): a workgroup's interval covers the last part of one chain, whole chains, and the first part of
// another.  A chain that straddles two workgroups is cut ONCE: the workgroup with the lower index (its OWNER) runs its first
// applications -- as the FIRST thing it does -- and parks the data in the chain's own output slot; the next workgroup (its
// RESUMER) continues from there as the LAST thing it does (both are ordinary tile stores / loads: same bits).  Because an interval
// is longer than a chain, the first part is finished long before the second is due, so nobody waits in practice.
//
// The hand-over is a word per chain (ChainState below), agent-scope atomics (the two workgroups sit on different XCDs, whose L2s
// are not coherent with each other).  It does NOT rely on the two workgroups being co-resident or on any dispatch order (round 5):
// a resumer that has waited `wait_ticks` for a chain nobody has committed to parking TAKES the chain -- it runs ALL of the chain's
// applications itself from d_input, which gives the same bits -- and an owner that finds its chain taken when it starts leaves it
// alone.  So every wait is for a workgroup that is running, whatever else occupies the device (another stream's kernels, another
// process, a balanced launch of another host thread); nothing is computed twice.
// per_wg = 0: the old schedule (one chain at a time, grid-strided; no hand-overs).
struct MultipleSchedule {
    int per_wg;              // applications per workgroup (> nreuses), or 0: grid-stride over whole chains
    unsigned base;           // 4 * the launch's epoch: flags[c] - base is the ChainState of chain c in THIS launch (smaller: an earlier launch's)
    unsigned* flags;         // one per chain, device memory owned by the host API (one buffer per launch in flight)
    int rotate;              // > 0: the wave's scheduling priority rotates every 2^rotate shader clocks (see WavePriority)
    unsigned wait_ticks;     // 100 MHz ticks a resumer waits for a chain whose owner has not started before it takes it over
    int delay_chain;         // fault injection (tests; -1: none): the owner of this chain sleeps delay_ticks ...
    unsigned delay_ticks;    //   ... before it commits to the chain (delay_after_commit = 0) or between its tile store and the parked word (1)
    int delay_after_commit;
    unsigned* residency;     // calibration launches only: [0] workgroups alive now, [1] the most that were alive at once
    unsigned long long* trace;   // experiments (SMFFT_SCHEDULE_TRACE): per workgroup {start, end} of s_memrealtime + {HW_ID, XCC_ID}; nullptr otherwise
};
// (WavePriority -- the rotation of the waves' scheduling priorities -- lives in smfft/smfft_engine.hpp: user kernels may use it too)
__device__ __forceinline__ void residency_enter(unsigned* r) {
    if (r && threadIdx.x == 0) atomicMax(r + 1, atomicAdd(r, 1u) + 1u);
}
__device__ __forceinline__ void residency_leave(unsigned* r) {
    if (r && threadIdx.x == 0) atomicSub(r, 1u);
}
// per workgroup: [0] start, [1] end (shader clock of its XCD), [2] HW_ID, [3] XCC_ID, [4] start, [5] end (100 MHz, device-wide),
// then for each of its first four pieces: [6 + 3k] tile in LDS, [7 + 3k] applications done, [8 + 3k] tile stored (100 MHz)
constexpr int kTraceWords = 18;
__device__ __forceinline__ void trace_mark(unsigned long long* trace, int slot) {
    if (trace && threadIdx.x == 0) {
        unsigned long long* w = trace + (size_t)kTraceWords * blockIdx.x;
        if (slot == 0) {
            w[0] = __builtin_readcyclecounter();
            w[2] = __builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((32 - 1) << 11));
            w[3] = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((32 - 1) << 11));
            w[4] = wall_clock64();
        } else if (slot == 1) {
            w[1] = __builtin_readcyclecounter();
            w[5] = wall_clock64();
        } else if (slot < kTraceWords) {
            w[slot] = wall_clock64();
        }
    }
}
__device__ __forceinline__ void trace_piece(unsigned long long* trace, int k, int what) {
    if (trace && k < 4) trace_mark(trace, 6 + 3 * k + what);
}

// Where the hand-over of a cut chain stands: flags[chain] - base.
//   (anything below 1: nothing yet)  ->  kChainOwned (the owner is at work on the chain's head: it WILL park it)  ->  kChainParked
//   (anything below 1: nothing yet)  ->  kChainTaken (the resumer has waited long enough for an owner that has not even started:
//                                                     the chain is its own from application 0)
// Both transitions out of "nothing yet" are atomics on the same word (the owner's a max, the resumer's a compare-and-swap of the
// stale value it polled), so exactly one of them happens.
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility; the XCDs' L2s are not coherent with each other): the parked tile is
// stored WRITE-THROUGH and read with `sc1` loads (SharedTile, smfft_engine.hpp), every storing wave drains its stores
// (s_waitcnt vmcnt(0)) in front of the workgroup barrier behind which ONE lane sets the word with an agent-scope atomic; the
// resumer polls that word relaxed with one lane, tells its workgroup through LDS, and every load of the tile is an `sc1` load.
// No cache-wide write-back or invalidate anywhere: with `fence(release / acquire, "agent")` around plain tile accesses -- rounds
// 4's form -- every workgroup's `buffer_wbl2` scanned its XCD's L2, 10-90 us of a README launch (profiles/r05_handover_forms.txt).
enum ChainState : unsigned { kChainOwned = 1u, kChainParked = 2u, kChainTaken = 3u };
constexpr unsigned long long kChainWaitLimit = 120ull * 100000000ull;      // 100 MHz ticks

// one word from thread 0 to every thread of the workgroup (single-wave workgroups: the barriers compile to nothing)
__device__ __forceinline__ unsigned workgroup_broadcast(unsigned value) {
    __shared__ unsigned word;
    __syncthreads();                                    // the previous broadcast has been read
    if (threadIdx.x == 0) word = value;
    __syncthreads();
    return word;
}
__device__ __forceinline__ void sleep_ticks(unsigned ticks) {
    const unsigned long long t0 = wall_clock64();       // 100 MHz
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__device__ __forceinline__ global_u32* chain_word(const MultipleSchedule& sch, int tile) { return (global_u32*)(sch.flags + tile); }
// OWNER, the first thing it does with a chain's head: commit to it -- ONE atomic max (the states are ordered and anything an
// earlier launch left in the word is smaller than this launch's kChainOwned).  0: the resumer has taken the chain (this workgroup
// started that late): the head is not computed and nothing is stored.
__device__ __forceinline__ unsigned chain_own(const MultipleSchedule& sch, int tile) {
    unsigned go = 0;
    if (threadIdx.x == 0) {
        if (tile == sch.delay_chain && !sch.delay_after_commit) sleep_ticks(sch.delay_ticks);
        go = __hip_atomic_fetch_max(chain_word(sch, tile), sch.base + kChainOwned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sch.base + kChainTaken;
    }
    return go;
}
__device__ __forceinline__ void chain_park(const MultipleSchedule& sch, int tile) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave: its write-through stores have arrived ...
    __syncthreads();                                    // ... before thread 0 says so
    if (threadIdx.x == 0) {
        if (tile == sch.delay_chain && sch.delay_after_commit) sleep_ticks(sch.delay_ticks);
        __hip_atomic_store(chain_word(sch, tile), sch.base + kChainParked, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// RESUMER: the application its piece really starts from -- app0 when the parked data are there (to be loaded from d_output, SHARED),
// 0 when it has taken the chain (to be loaded from d_input).  Nobody waits here in practice (the head of a chain is the first thing
// its owner does, the tail the last thing the resumer does); when the owner has not even started after wait_ticks -- it is not
// resident yet, or the device is shared -- the resumer stops waiting.  An owner that HAS started is running: that wait is bounded
// by the head's applications.
__device__ __forceinline__ int chain_resume_or_take(const MultipleSchedule& sch, int tile, int app0) {
    unsigned take = 0;
    if (threadIdx.x == 0) {
        global_u32* word = chain_word(sch, tile);
        const unsigned long long t0 = wall_clock64();   // 100 MHz
        for (;;) {
            unsigned v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == sch.base + kChainParked) break;
            // An owner that has committed is running, so this wait ends -- unless the owner's workgroup faulted or the words are not
            // this launch's (host-side misuse): after kChainWaitLimit (two minutes: a README chain takes milliseconds) the launch
            // ends with an error instead of hanging the device silently.
            if (wall_clock64() - t0 > kChainWaitLimit) __builtin_trap();
            if (v != sch.base + kChainOwned && wall_clock64() - t0 > sch.wait_ticks) {
                if (__hip_atomic_compare_exchange_strong(word, &v, sch.base + kChainTaken, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    take = 1;
                    break;
                }
                continue;                               // the owner started under our hands: look again
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    take = workgroup_broadcast(take);                   // (the other waves load behind this barrier)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // no instruction: keeps the compiler from moving the tile loads above the poll
    return take ? 0 : app0;
}
// The parts of chains this workgroup owns, in the order the schedule needs: the chain it shares with the NEXT workgroup (its
// head) first, the one it shares with the PREVIOUS (its tail) last.  One loop for both schedules: chain = first + k * step,
// k < count, clipped to the workgroup's interval [lo, hi) of applications.
struct PieceLoop {
    long lo, hi;
    int first, step, count, nreuses;
    __device__ __forceinline__ PieceLoop(const MultipleSchedule& sch, int ntiles, int nreuses_) : nreuses(nreuses_) {
        const long total = (long)ntiles * nreuses;
        lo = 0;
        hi = total;
        first = blockIdx.x;
        step = gridDim.x;
        count = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
        if (sch.per_wg != 0) {
            lo = (long)blockIdx.x * sch.per_wg;
            hi = lo + sch.per_wg < total ? lo + sch.per_wg : total;
            // (64-bit division runs on the vector unit: the quotients are wave-uniform, readfirstlane returns them to scalar registers)
            first = __builtin_amdgcn_readfirstlane((int)((hi - 1) / nreuses));
            step = -1;
            count = lo < hi ? first - __builtin_amdgcn_readfirstlane((int)(lo / nreuses)) + 1 : 0;
        }
    }
    __device__ __forceinline__ int tile(int k) const { return first + k * step; }
    __device__ __forceinline__ int app0(int k) const {
        const long c0 = (long)tile(k) * nreuses;
        return __builtin_amdgcn_readfirstlane((int)((lo > c0 ? lo : c0) - c0));
    }
    __device__ __forceinline__ int app1(int k) const {
        const long c0 = (long)tile(k) * nreuses;
        return __builtin_amdgcn_readfirstlane((int)((hi < c0 + nreuses ? hi : c0 + nreuses) - c0));
    }
};
// One piece of the loop above with its hand-overs resolved: `skip` (an owner whose chain was taken before it started), else
// applications [app0, app1) of `tile`, loaded from d_output when app0 > 0 (parked there by the owner: `resumed`) and from d_input
// otherwise.
struct Piece {
    int tile, app0, app1;
    bool park;               // the piece ends before the chain does -- its data are handed to the next workgroup
    bool skip;
    __device__ __forceinline__ Piece(const PieceLoop& pieces, const MultipleSchedule& sch, int k) : tile(pieces.tile(k)), app0(pieces.app0(k)), app1(pieces.app1(k)), skip(false) {
        park = app1 < pieces.nreuses;
        if (park) skip = workgroup_broadcast(chain_own(sch, tile)) == 0;
        else if (app0 > 0) app0 = chain_resume_or_take(sch, tile, app0);
    }
    __device__ __forceinline__ bool resumed() const { return app0 > 0; }
};

// the engines whose transforms live on lanes and registers alone (smfft_engine.hpp): N = 32 (a pair of lanes), N = 64 without reorder (a quad)
template <int N, int DIR, int REORDER> struct LaneEngine;
template <int DIR, int REORDER> struct LaneEngine<32, DIR, REORDER> { using type = PairEngine32<DIR, REORDER>; };
template <int DIR> struct LaneEngine<64, DIR, 0> { using type = QuadEngine64<DIR>; };
constexpr bool on_lanes(int n, int reorder) { return n == 32 || (n == 64 && !reorder); }

// C2C, multiple, N = 32 and N = 64 without reorder (CT:553-572): the first nSlots FFTs are loaded once, transformed nreuses (= NREUSES = 100 in the benchmark; a
// kernel argument so the tests can run 1, 2 and 4 applications) times, stored once.  An FFT is a pair (N = 64: a quad) of lanes and lives in their
// registers from the first application of a piece to its last (PairEngine32, smfft_engine.hpp: the stage across the pair is one
// DPP-fed v_fmac_f32 per dword, the layout alternates instead of being restored); the image in LDS -- natural order, padded for
// the no-reorder variants as the tile copies have it -- is touched where a piece starts and ends.  (Rounds 2-4 ran this length on
// the general float2 engine -- a radix-2 stage, a lane <-> register transpose of 32 selects, a radix-16 stage and, without reorder,
// a bit-reversal through LDS per application: profiles/r05_pair32.txt has both; the planar engine measured 10 % slower than that
// one, profiles/r03_ab_planar_small.txt.)
// (d_input / d_output are not __restrict__ here: a resumed piece READS the tile another workgroup of this launch parked in d_output)
// FUSED = false (smfft_launch path 2; round 6): every application loads its registers from the image in LDS and stores its results
// there -- the shape of the reference's own loop, which calls do_SMFFT_CT_DIT(s_input) NREUSES times (CT:553-572), and what ONE call
// of the device function costs on these lengths.  Same bits as the fused loop (load / store flip sign BITS where the fused loop carries
// a negated lane; flipping twice is exact).
template <int N, int DIR, int REORDER, bool FUSED = true>
__device__ __forceinline__ void c2c_multiple_body(const float2* d_input, float2* d_output, int nSlots, int nreuses, MultipleSchedule sch, float2* s) {
    static_assert(N == 32 || (N == 64 && !REORDER), "the planar engine takes everything else");
    using G = Geometry<N>;
    constexpr bool kPaddedImage = !REORDER;
    typename LaneEngine<N, DIR, REORDER>::type eng;
    eng.init(threadIdx.x);
    float2* sf = s + eng.fft * G::SF;
    const int ntiles = (nSlots + G::kCompactFfts - 1) / G::kCompactFfts;
    const PieceLoop pieces(sch, ntiles, nreuses);
    const WavePriority priority(sch.rotate);
    priority.between_applications();
    trace_mark(sch.trace, 0);
    residency_enter(sch.residency);
    for (int k = 0; k < pieces.count; ++k) {
        const Piece piece(pieces, sch, k);
        if (piece.skip) continue;
        const long first = (long)piece.tile * G::kCompactFfts;
        fft_sync<G::kMultiWave>();
        if (piece.resumed()) shared_tile_to_lds<N, kPaddedImage>(d_output + first * N, s, first, nSlots);
        else tile_to_lds<N, kPaddedImage>(d_input + first * N, s, first, nSlots);
        fft_sync<G::kMultiWave>();
        trace_piece(sch.trace, k, 0);
        // The applications run on the lanes' registers alone: the image in LDS is read where the piece starts and written where it
        // ends.  An application's form follows from its number in the chain, so the loop is unrolled by two -- which puts several
        // inlined copies of the same source into the kernel; their arithmetic is written with its rounding fixed (cmul_fixed,
        // SmallDft<..., FIXED>), or a chain cut on an odd application would not end with the bits of an uncut one.
        float2 r[16];
        if constexpr (!FUSED) {
            // (an application's form follows from its number in the chain: unrolled by two like the fused loop, so that load, apply and
            //  store see their parity at compile time)
            auto one = [&](auto odd) {
                priority.at_application();
                eng.load(r, sf, odd ? 1 : 0);
                fft_sync<G::kMultiWave>();          // (the wave's loads of the image precede its stores)
                eng.apply(r, odd);
                eng.store(r, sf, odd ? 0 : 1);
                fft_sync<G::kMultiWave>();
            };
            int f = piece.app0;
            if (f & 1) {
                one(std::true_type());
                ++f;
            }
            for (; f + 1 < piece.app1; f += 2) {
                one(std::false_type());
                one(std::true_type());
            }
            if (f < piece.app1) one(std::false_type());
        } else {
        eng.load(r, sf, piece.app0);
        int f = piece.app0;
        if (f & 1) {
            priority.at_application();
            eng.apply(r, true);
            ++f;
        }
        for (; f + 1 < piece.app1; f += 2) {
            priority.at_application();
            eng.apply(r, false);
            priority.at_application();
            eng.apply(r, true);
        }
        if (f < piece.app1) {
            priority.at_application();
            eng.apply(r, false);
        }
        fft_sync<G::kMultiWave>();              // (the wave's loads of the image precede its stores)
        eng.store(r, sf, piece.app1);
        fft_sync<G::kMultiWave>();
        }
        priority.between_applications();
        trace_piece(sch.trace, k, 1);
        if (!piece.park) {
            lds_to_tile<N, kPaddedImage>(d_output + first * N, s, first, nSlots);
        } else {
            lds_to_shared_tile<N, kPaddedImage>(d_output + first * N, s, first, nSlots);
            chain_park(sch, piece.tile);
        }
        trace_piece(sch.trace, k, 2);
    }
    trace_mark(sch.trace, 1);
    residency_leave(sch.residency);
}

// The same kernel on the planar engine (smfft_planar.hpp; N >= 64): every LDS image as two planes of dwords, stored with
// ds_write_addtid_b32 and read back in contiguous runs.  The image between applications is the natural-order image in
// planar form (row c, dword of thread u = x[u + T*c]); the no-reorder variants read their bit-reversed rows from it,
// the reorder variants store every result and forward their own registers into the next application (FUSED; false: every
// application re-loads its input from the LDS image -- SMFFT_DIT_multiple_unfused).
constexpr int kPlanarMinN = 64;
template <int N, int DIR, int REORDER, bool FUSED = true>
__device__ __forceinline__ void c2c_multiple_body_planar(const float2* d_input, float2* d_output, int nSlots, int nreuses, MultipleSchedule sch, float* planes) {
    using G = Geometry<N>;
    PlanarEngine<N, DIR, REORDER> eng;
    eng.init(threadIdx.x, planes);
    const int ntiles = (nSlots + G::kCompactFfts - 1) / G::kCompactFfts;
    const PieceLoop pieces(sch, ntiles, nreuses);
    const WavePriority priority(sch.rotate);
    priority.between_applications();
    trace_mark(sch.trace, 0);
    residency_enter(sch.residency);
    for (int k = 0; k < pieces.count; ++k) {
        const Piece piece(pieces, sch, k);
        if (piece.skip) continue;
        const long first = (long)piece.tile * G::kCompactFfts;
        const int napps = piece.app1 - piece.app0;
        planar_sync<G::kMultiWave>();
        if (piece.resumed()) shared_tile_to_planes<N, DIR, REORDER>(d_output + first * N, planes, first, nSlots);
        else tile_to_planes<N, DIR, REORDER>(d_input + first * N, planes, first, nSlots);
        planar_sync<G::kMultiWave>();
        trace_piece(sch.trace, k, 0);
        float2 r[16];
        if constexpr (REORDER && FUSED) {
            eng.image_load_own(r, planes);
            for (int f = 0; f < napps; ++f) {
                priority.at_application();
                eng.natural_to_slots(r);
                eng.transform_from_pass1_slots(r, planes);
                planar_sync<G::kMultiWave>();       // the last pass's reads are done before the result overwrites them
                eng.image_store(r);
            }
        } else if constexpr (REORDER) {
            // unfused: what ONE call of the device function costs when its input is data in LDS and its output is data in
            // LDS -- the application reads its sixteen inputs back from the stored image instead of keeping them in registers
            for (int f = 0; f < napps; ++f) {
                priority.at_application();
                planar_sync<G::kMultiWave>();       // the image is complete (stored by the threads that computed it)
                eng.image_load_own(r, planes);
                eng.natural_to_slots(r);
                eng.transform_from_pass1_slots(r, planes);
                planar_sync<G::kMultiWave>();
                eng.image_store(r);
            }
        } else {
#ifdef SMFFT_TIMING_ONLY_NOREORDER_FORWARD
            // TIMING ONLY (wrong results; tools/noreorder_bound.sh): the no-reorder loop WITHOUT its bit-reversed re-read -- what the kernel
            // would cost if an application's results could be forwarded in registers as the natural-order kernel's are (DESIGN.md 5.2 shows
            // why they cannot: the sixteen results of a thread are a coset of index bits, its sixteen inputs a contiguous block)
            eng.image_load_bitrev(r, planes);
#endif
            for (int f = 0; f < napps; ++f) {
                priority.at_application();
#ifndef SMFFT_TIMING_ONLY_NOREORDER_FORWARD
                eng.image_load_bitrev(r, planes);
#endif
                eng.transform_from_pass1_slots(r, planes);
                planar_sync<G::kMultiWave>();
                eng.image_store(r);
                planar_sync<G::kMultiWave>();       // the reference omits this (latent race, CT:563-565)
            }
        }
        planar_sync<G::kMultiWave>();
        priority.between_applications();
        trace_piece(sch.trace, k, 1);
        if (!piece.park) {
            planes_to_tile<N, DIR, REORDER>(d_output + first * N, planes, first, nSlots);
        } else {
            planes_to_shared_tile<N, DIR, REORDER>(d_output + first * N, planes, first, nSlots);
            chain_park(sch, piece.tile);
        }
        trace_piece(sch.trace, k, 2);
    }
    trace_mark(sch.trace, 1);
    residency_leave(sch.residency);
}

// ------------------------------------------------------------------------------------------------
// R2C / C2R external kernels (real length 2L through a complex FFT of length L, RC:269-365): tiled form.
// ------------------------------------------------------------------------------------------------
// The same split / merge on the NATURAL REGISTER layout (r[q] = element u + T*q of the thread's FFT), for FFTs that
// live in one wave (L <= 1024).  Element i pairs with L - i, which thread (T - u) mod T holds in register 15 - q
// (thread 0 pairs with itself, register 16 - q): one ds_bpermute per dword fetches it -- no LDS memory, no
// barrier -- and the ONE formula  out[i] = H1 + W^i * H2,  A = x[i], B = x[L - i],  covers both halves (for L - i it
// evaluates to the conjugate expression the LDS version writes there, RC:302-308).  W^i = W^u * W_32^q: one table
// value per thread and 15 constants.  Replaces store_lds + hermitian_pass + load_lds (4 LDS passes, 3 syncs) in the
// external kernels where that measured faster.
template <int L, int DIR>
struct HermitianRegisters {
    static constexpr int T = L / 16;
    // used by the external kernels of real N = 1024 and 2048 (measured: 2048 R2C +2.6 %, C2R +3.0 %; 1024 0 / +1.8 %;
    // 512 -1.8 / -0.7 %: LDS form there; in-LDS path 1-5 % slower with it: LDS form there too)
    // kFromLds: the partner values come out of the FFT's LDS region (the registers are written there in natural order
    // first) instead of from ds_bpermute -- for the C2R of L = 2048, where the partner thread sits in another wave: one
    // LDS write + read of the data instead of the LDS-resident merge's two (C2R +3 %; the R2C of that length measured
    // 1.3 % SLOWER this way and L = 256 unchanged, so both keep the LDS form)
    static constexpr bool kFromLds = (L == 2048 && DIR == 1);
    static constexpr bool kEnabled = (L == 512 || L == 1024 || kFromLds);
    float2 wu;          // (-+i / 2) * W_{2L}^u (W conjugated, +i, for DIR = 1): see combine
    int partner_addr;   // byte address of the partner lane for ds_bpermute
    bool first;         // u == 0
    __device__ __forceinline__ void init(int tid) {
        const int u = tid % T, lane = tid & 63;
        first = (u == 0);
        partner_addr = 4 * ((lane - u) + ((T - u) % T));
        const float2 w = twiddle<DIR>(u * (4096 / (2 * L)));
        wu = DIR ? make_float2(-0.5f * w.y, 0.5f * w.x) : make_float2(0.5f * w.y, -0.5f * w.x);
    }
    // out[i] = H1 + W^i * H2 with A = x[i], B = x[L - i]  (i = u + T*q) and, as upstream (RC:289-328),
    //   H1 = ((A.x + B.x)/2, (A.y - B.y)/2),  H2 = (ohx * (A.y + B.y), ohy * (A.x - B.x)),  (ohx, ohy) = (1/2, -1/2) forward, (-1/2, 1/2) inverse.
    // With S = A + conj(B) and D = A - conj(B): H1 = S/2 and H2 = (-+i/2) * D, so out = S/2 + V * D, V = (-+i/2) * W^u * W_32^q:
    // 4 additions + 6 multiply-adds per element once V is there (`wu` holds (-+i/2) * W^u; the fifteen products V are the
    // same for every tile and stay in registers across the grid-stride loop).
    // All sixteen partner values fetched at once: thirty-two ds_bpermute back to back, their latencies overlapped, at the
    // price of 135-146 VGPRs (3 waves per SIMD).  An in-place form that fetched two partners at a time, software-pipelined,
    // at 93-113 VGPRs = 4 waves per SIMD measured 0.6-2 % SLOWER on the same buffers (profiles/r02_ab_rc.txt) and is gone.
    // sf: the FFT's LDS region (kFromLds only; free on entry, the caller orders its later re-use).
    __device__ __forceinline__ void apply(float2 (&r)[16], float2* sf = nullptr) const {
        constexpr float c32[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f, 0.55557023301960229f,
                                   0.38268343236508984f, 0.19509032201612833f, 0.f, -0.19509032201612819f, -0.38268343236508973f, -0.55557023301960196f,
                                   -0.70710678118654746f, -0.83146961230254535f, -0.92387953251128674f, -0.98078528040323043f};
        constexpr float s32[16] = {0.f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654746f, 0.83146961230254524f,
                                   0.92387953251128674f, 0.98078528040323043f, 1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254546f,
                                   0.70710678118654757f, 0.55557023301960218f, 0.38268343236508989f, 0.19509032201612861f};
        float2 B[16];
        if constexpr (kFromLds) {
            const int u = threadIdx.x % T;
#pragma unroll
            for (int q = 0; q < 16; ++q) sf[u + T * q] = r[q];
            fft_sync<(T > 64)>();
            // x[L - (u + T*q)] = x[(T - u) + T*(15 - q)] -- for thread 0, too (x[T*(16 - q)], q >= 1); its q = 0 reads one
            // element past the data (inside the region's padding) and is replaced by the packed DC / Nyquist value below
            const float2* partner = sf + (T - u);
#pragma unroll
            for (int q = 0; q < 16; ++q) B[q] = partner[T * (15 - q)];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float2 src = r[15 - q];
                const float bx = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(src.x)));
                const float by = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(src.y)));
                const float2 own = r[(16 - q) & 15];
                B[q] = first ? own : make_float2(bx, by);
            }
        }
        // the fifteen products V are the same for every tile: the compiler keeps them in 30 registers across the grid-stride loop
        const float2 w = wu;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float2 A = r[q];
            const float2 S = make_float2(A.x + B[q].x, A.y - B[q].y);       // A + conj(B)
            const float2 D = make_float2(A.x - B[q].x, A.y + B[q].y);       // A - conj(B)
            const float2 V = (q == 0) ? w : cmul(w, make_float2(c32[q], DIR ? s32[q] : -s32[q]));   // (-+i/2) * W^i
            float2 out = make_float2(fmaf(V.x, D.x, fmaf(-V.y, D.y, 0.5f * S.x)), fmaf(V.x, D.y, fmaf(V.y, D.x, 0.5f * S.y)));
            if (q == 0) {   // element 0 of thread 0 packs DC and Nyquist (RC:280-286, 332-339)
                const float2 packed = DIR ? make_float2(0.5f * (A.x + A.y), 0.5f * (A.x - A.y)) : make_float2(A.x + A.y, A.x - A.y);
                out = first ? packed : out;
            }
            r[q] = out;
        }
    }
};

template <int L, int DIR>
__device__ __forceinline__ void r2c_c2r_external_body(const float2* __restrict__ d_input, float2* __restrict__ d_output, int nFFTs, int pace, float2* s) {
    using G = Geometry<L>;
    Engine<L, DIR, 1> eng;
    eng.init(threadIdx.x);
    HermitianRegisters<L, DIR> herm;
    herm.init(threadIdx.x);
    float2* sf = s + eng.fft * G::SF;
    const int ntiles = (nFFTs + G::kFftsPerBlock - 1) / G::kFftsPerBlock;
    float2 r[16];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long f = (long)tile * G::kFftsPerBlock + eng.fft;
        const bool active = f < nFFTs;
        if constexpr (HermitianRegisters<L, DIR>::kEnabled) {
            eng.load_global(r, d_input + (active ? f : 0) * L);
            if (G::kMultiWave) __syncthreads();   // the previous tile's LDS reads are complete
            if (pace) vmem_throttle(s + (threadIdx.x >> 6) * 1088, r, pace);
            if (DIR == 1) {
                herm.apply(r, sf);
                if (HermitianRegisters<L, DIR>::kFromLds) fft_sync<G::kMultiWave>();   // partner reads done before the exchanges write
            }
            eng.transform(r, sf);
            if (DIR == 0) {
                if (HermitianRegisters<L, DIR>::kFromLds) fft_sync<G::kMultiWave>();   // last-pass reads done before the registers are written over them
                herm.apply(r, sf);
            }
            eng.store_global(r, d_output + f * L, active);
        } else if (DIR == 0) {
            eng.load_global(r, d_input + (active ? f : 0) * L);
            if (G::kMultiWave) __syncthreads();
            eng.transform(r, sf);
            fft_sync<G::kMultiWave>();
            eng.store_lds(r, sf);
            fft_sync<G::kMultiWave>();
            hermitian_pass<L, 0>(sf, eng.u);
            fft_sync<G::kMultiWave>();
            eng.load_lds(r, sf);               // natural order: r[c] = sf[u + T*c]
            eng.store_global(r, d_output + f * L, active);
        } else {
            eng.load_global(r, d_input + (active ? f : 0) * L);
            if (G::kMultiWave) __syncthreads();
            eng.store_lds(r, sf);
            fft_sync<G::kMultiWave>();
            hermitian_pass<L, 1>(sf, eng.u);
            fft_sync<G::kMultiWave>();
            eng.load_lds(r, sf);
            fft_sync<G::kMultiWave>();
            eng.transform(r, sf);
            eng.store_global(r, d_output + f * L, active);
        }
    }
}

// R2C / C2R in-LDS path on the planar engine: the complex transform of length L as in c2c_multiple_body_planar (reorder
// roles, registers forwarded from one application to the next) with the Hermitian split / merge done on the registers,
// pair-wise (PlanarEngine::hermitian_apply_pairs).  Per application: the C2C's LDS traffic + 32 dword stores + 32 dword reads.
template <int L, int DIR>
__device__ __forceinline__ void r2c_c2r_multiple_body_planar(const float2* d_input, float2* d_output, int nSlots, int nreuses, MultipleSchedule sch, float* planes) {
    using G = Geometry<L>;
    PlanarEngine<L, DIR, 1> eng;
    eng.init(threadIdx.x, planes);
    eng.init_hermitian();
    const int ntiles = (nSlots + G::kCompactFfts - 1) / G::kCompactFfts;
    const PieceLoop pieces(sch, ntiles, nreuses);
    const WavePriority priority(sch.rotate);
    priority.between_applications();
    residency_enter(sch.residency);
    for (int k = 0; k < pieces.count; ++k) {
        const Piece piece(pieces, sch, k);
        if (piece.skip) continue;
        const long first = (long)piece.tile * G::kCompactFfts;
        const int napps = piece.app1 - piece.app0;
        planar_sync<G::kMultiWave>();
        if (piece.resumed()) shared_tile_to_planes<L, DIR, 1>(d_output + first * L, planes, first, nSlots);
        else tile_to_planes<L, DIR, 1>(d_input + first * L, planes, first, nSlots);
        planar_sync<G::kMultiWave>();
        float2 r[16];
        eng.image_load_own(r, planes);
        // pair-wise split / merge (PlanarEngine::hermitian_apply_pairs): only the rows 8..15 go through the image
        // (R2C: the split of application f runs at the head of iteration f + 1 and once more after the loop -- the loop is
        //  then shaped like the C2R's, split / merge in front of the transform, which the compiler keeps at four waves per SIMD)
        for (int f = 0; f < napps; ++f) {
            priority.at_application();
            if (DIR == 1 || f > 0) eng.hermitian_apply_pairs(r, planes);   // rows 8..15: the tile (C2R), or stored below
            eng.natural_to_slots(r);
            eng.transform_from_pass1_slots(r, planes);
            planar_sync<G::kMultiWave>();
            if (DIR == 0 || f + 1 < napps) {
                eng.image_store_upper(r);
                planar_sync<G::kMultiWave>();
            }
        }
        if (DIR == 0) eng.hermitian_apply_pairs(r, planes);
        planar_sync<G::kMultiWave>();
        eng.image_store(r);
        planar_sync<G::kMultiWave>();
        priority.between_applications();
        if (!piece.park) {
            planes_to_tile<L, DIR, 1>(d_output + first * L, planes, first, nSlots);
        } else {
            planes_to_shared_tile<L, DIR, 1>(d_output + first * L, planes, first, nSlots);
            chain_park(sch, piece.tile);
        }
    }
    residency_leave(sch.residency);
}

}  // namespace smfft

// ------------------------------------------------------------------------------------------------
// Calibration kernel: the external kernels' exact global access shape (one wave per 8 KiB chunk,
// 16 x 8 B/lane non-temporal loads at 512 B stride, 16 stores, same grid-stride over 4096-element
// tiles) with no FFT in between.  bench.py times it next to the FFT so the HBM-bound kernels are
// reported against a same-run, same-shape copy ceiling as well as against the 8 TB/s datasheet peak.
// ------------------------------------------------------------------------------------------------
template <int kUnused = 0>
__global__ void __launch_bounds__(256) SMFFT_stream_copy(const float2* __restrict__ d_input, float2* __restrict__ d_output, long ntiles, int pace) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float2 s_rows[4352];
    // (a contiguous run of tiles per workgroup instead of this grid stride measured the same, round 1)
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float2* g = d_input + tile * 4096 + wave * 1024 + lane;
        float2* o = d_output + tile * 4096 + wave * 1024 + lane;
        float2 r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = smfft::gload(g + 64 * c);
        if (pace) smfft::vmem_throttle(s_rows + wave * 1088, r, pace);   // the same rate limiter as the FFT kernels
#pragma unroll
        for (int c = 0; c < 16; ++c) smfft::gstore(o + 64 * c, r[c]);
    }
}

// Read-only pass in the same access shape (the pure read rate of a buffer: the device's own ceiling that
// smfft_malloc_pair judges its copy probes against).  Nothing is stored: the sum can never equal the sentinel.
template <int kUnused = 0>
__global__ void __launch_bounds__(256) SMFFT_stream_read(const float2* __restrict__ d_input, float2* __restrict__ sink, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ax = 0.f, ay = 0.f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const float2* g = d_input + tile * 4096 + wave * 1024 + lane;
        float2 r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = smfft::gload(g + 64 * c);
#pragma unroll
        for (int c = 0; c < 16; ++c) { ax += r[c].x; ay += r[c].y; }
    }
    if (ax == 1.2345e38f && ay == -1.2345e38f) sink[threadIdx.x] = make_float2(ax, ay);
}

// Write-only pass in the same access shape: the pure write rate of a piece of memory, which is what tells a MIXED
// physical chunk (pages from several of the three HBM classes: writes ~20 % faster) from an ordinary one (smfft_api.hip).
template <int kUnused = 0>
__global__ void __launch_bounds__(256) SMFFT_stream_write(float2* __restrict__ d_output, long ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float2 v = make_float2((float)lane, (float)wave);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float2* o = d_output + tile * 4096 + wave * 1024 + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) smfft::gstore(o + 64 * c, v);
    }
}

// ------------------------------------------------------------------------------------------------
// The library's kernels (reference names, one more argument than upstream: the batch size, because grids are
// capped and grid-strided).  The reference-shaped two-argument forms are in smfft/smfft_device_functions.hpp.
// ------------------------------------------------------------------------------------------------
template <class const_params>
__global__ void __launch_bounds__(256) SMFFT_DIT_external(const float2* d_input, float2* d_output, int nFFTs, int pace) {
    __shared__ float2 s_input[const_params::tile_sm_required];
    smfft::c2c_external_body<const_params::fft_size, const_params::fft_direction, const_params::fft_reorder>(d_input, d_output, nFFTs, pace, s_input);
}
// The same kernel compiled for exactly 3 waves per SIMD.  The launcher uses it for the N = 4096 reorder
// transforms: their 134 VGPRs give 3 waves per SIMD either way, but with the target stated the scheduler
// stops trading instruction-level parallelism for registers it cannot turn into a 4th wave: 5.89 -> 6.07 TB/s
// (tools/ab_probe.py).  Every other length measured equal or worse with a stated target.
template <class const_params>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
SMFFT_DIT_external_occ3(const float2* d_input, float2* d_output, int nFFTs, int pace) {
    __shared__ float2 s_input[const_params::tile_sm_required];
    smfft::c2c_external_body<const_params::fft_size, const_params::fft_direction, const_params::fft_reorder>(d_input, d_output, nFFTs, pace, s_input);
}

// Launch bounds of the compact (in-LDS) kernels: every one is compiled for FOUR waves per SIMD -- what the schedule is built on
// (the rotation of the wave priorities has four ranks; five waves per SIMD at 96 registers measured 3-27 % slower:
// profiles/r04_ab_balance_variants.txt, r04_ab_five_waves.txt).  N = 2048 / 4096: at least four (what their LDS allows, 8-9 / 4
// workgroups per CU; left alone the planar engine takes 133-170 registers = 3 or 2 waves per SIMD, profiles/r03_ab_planar_b.txt).
// The single-wave kernels: exactly four -- left alone most of them take 100-128 registers anyway, but the allocator is free to
// land on 92 (N = 64 did, after an unrelated change of the kernel's prologue: a fifth wave, 19 workgroups per CU, -35 %).
// And so that the residency does not hang on the allocator's mood, a single-wave kernel's LDS is padded to what lets exactly
// sixteen workgroups share a CU's 160 KiB (their images are 8.3-8.7 KiB: nineteen would fit).
#define SMFFT_COMPACT_BOUNDS(N) __launch_bounds__(smfft::Geometry<N>::kCompactThreads, 4)
namespace smfft {
constexpr int kSingleWaveLdsBytes = 9728;      // 160 KiB / 9728 = 16.8
template <int N>
constexpr int compact_lds_floats(int needed) { return (N <= 1024 && needed * 4 < kSingleWaveLdsBytes) ? kSingleWaveLdsBytes / 4 : needed; }
}  // namespace smfft

template <class const_params>
__global__ void SMFFT_COMPACT_BOUNDS(const_params::fft_size) SMFFT_DIT_multiple(const float2* d_input, float2* d_output, int nSlots, int nreuses, smfft::MultipleSchedule sch) {
    constexpr int N = const_params::fft_size;
    if constexpr (!smfft::on_lanes(N, const_params::fft_reorder)) {
        __shared__ __attribute__((aligned(16))) float s_planes[smfft::compact_lds_floats<N>(smfft::PlanarGeometry<N, const_params::fft_reorder>::kLdsFloats)];
        smfft::c2c_multiple_body_planar<N, const_params::fft_direction, const_params::fft_reorder>(d_input, d_output, nSlots, nreuses, sch, s_planes);
    } else {
        __shared__ float2 s_input[smfft::compact_lds_floats<N>(2 * smfft::Geometry<N>::kCompactLds) / 2];
        smfft::c2c_multiple_body<N, const_params::fft_direction, const_params::fft_reorder>(d_input, d_output, nSlots, nreuses, sch, s_input);
    }
}
// The same kernel WITHOUT cross-application fusion (natural-order variants, planar lengths): every application reads its
// input from the LDS image and leaves its output there, which is what ONE call of the device function costs a user kernel
// whose data live in LDS (CT:553-572 calls do_SMFFT_CT_DIT on s_input NREUSES times; nothing survives a call in registers).
// smfft_launch(..., path = 2); bench.py reports it next to the fused figure and the reference-contract path.
// (the planar no-reorder variants read their input from the image in every application as they are: path 2 launches SMFFT_DIT_multiple
//  for them.  The lane engines -- N = 32, N = 64 without reorder -- keep a chain in registers from its first application to its last:
//  their path 2 is c2c_multiple_body<..., false>, one image load and one image store per application.)
template <class const_params>
__global__ void SMFFT_COMPACT_BOUNDS(const_params::fft_size) SMFFT_DIT_multiple_unfused(const float2* d_input, float2* d_output, int nSlots, int nreuses, smfft::MultipleSchedule sch) {
    constexpr int N = const_params::fft_size;
    if constexpr (smfft::on_lanes(N, const_params::fft_reorder)) {
        __shared__ float2 s_input[smfft::compact_lds_floats<N>(2 * smfft::Geometry<N>::kCompactLds) / 2];
        smfft::c2c_multiple_body<N, const_params::fft_direction, const_params::fft_reorder, false>(d_input, d_output, nSlots, nreuses, sch, s_input);
    } else {
        static_assert(N >= smfft::kPlanarMinN && const_params::fft_reorder, "the planar no-reorder variants are unfused as they are");
        __shared__ __attribute__((aligned(16))) float s_planes[smfft::compact_lds_floats<N>(smfft::PlanarGeometry<N, 1>::kLdsFloats)];
        smfft::c2c_multiple_body_planar<N, const_params::fft_direction, 1, false>(d_input, d_output, nSlots, nreuses, sch, s_planes);
    }
}

// Stockham C2C program: un-normalised INVERSE (+i) transform, natural order (ST:76, :429).
template <class const_params>
__global__ void __launch_bounds__(256) FFT_GPU_external(const float2* d_input, float2* d_output, int nFFTs, int pace) {
    __shared__ float2 s_input[4352];
    smfft::c2c_external_body<const_params::fft_length, 1, 1>(d_input, d_output, nFFTs, pace, s_input);
}
template <class const_params>
__global__ void SMFFT_COMPACT_BOUNDS(const_params::fft_length) FFT_GPU_multiple(const float2* d_input, float2* d_output, int nSlots, int nreuses, smfft::MultipleSchedule sch) {
    constexpr int N = const_params::fft_length;
    if constexpr (N >= smfft::kPlanarMinN) {
        __shared__ __attribute__((aligned(16))) float s_planes[smfft::compact_lds_floats<N>(smfft::PlanarGeometry<N, 1>::kLdsFloats)];
        smfft::c2c_multiple_body_planar<N, 1, 1>(d_input, d_output, nSlots, nreuses, sch, s_planes);
    } else {
        __shared__ float2 s_input[smfft::compact_lds_floats<N>(2 * smfft::Geometry<N>::kCompactLds) / 2];
        smfft::c2c_multiple_body<N, 1, 1>(d_input, d_output, nSlots, nreuses, sch, s_input);
    }
}

// R2C/C2R program.
template <class const_params, class const_direction>
__global__ void __launch_bounds__(256) FFT_GPU_R2C_C2R_external(const float2* d_input, float2* d_output, int nFFTs, int pace) {
    __shared__ float2 s_input[4352];
    smfft::r2c_c2r_external_body<const_params::fft_length, const_direction::fft_direction>(d_input, d_output, nFFTs, pace, s_input);
}
template <class const_params, class const_direction>
__global__ void SMFFT_COMPACT_BOUNDS(const_params::fft_length) FFT_GPU_R2C_C2R_multiple(const float2* d_input, float2* d_output, int nSlots, int nreuses, smfft::MultipleSchedule sch) {
    constexpr int L = const_params::fft_length;
    __shared__ __attribute__((aligned(16))) float s_planes[smfft::compact_lds_floats<L>(smfft::PlanarGeometry<L, 1>::kLdsFloats)];
    smfft::r2c_c2r_multiple_body_planar<L, const_direction::fft_direction>(d_input, d_output, nSlots, nreuses, sch, s_planes);
}
