// kernels_stream_pipe.cuh -- software-pipelined variant of the streaming kernel (opt-in: LPHQ_STREAM_PIPELINE=1).
//
// STATUS: verified on the CPU emulator (tests/test_emu_cpu.py) and by register / LDS budget of the gfx950 build only.
// It has NOT been timed on hardware yet (GPU access was closed when it was written, profiles/r01_incident_gpu_access.txt),
// so the launcher keeps k_query_tiles_stream as the default until a measurement says otherwise.
//
// Same phases A-F and the same results as k_query_tiles_stream (kernels_device.cuh); what changes is the schedule.
// profiles/r01_stream_phase_counters.txt: a third of that kernel's time is phase E, a chain of three dependent
// gathers (pilot codeword -> H(pilot) -> fused block) executed by ~60 % of the lanes with nothing else to do, and
// phase A waits on the tile's own sequence loads.  Here a workgroup is persistent and walks tiles; the gathers of
// tile i are issued into registers and consumed one phase later, with the integer phases of tile i+1 in between:
//
//      issue sequence loads (i+1)
//      E1(i)  leader -> Murmur -> bucket          issue pilot-codeword load
//      A(i+1) pack bases                          (waits for the sequence loads only)
//      E2(i)  codeword                            issue H(pilot) load
//      B(i+1) one minimizer hash per position
//      E3(i)  position, free-slot fix-up          issue the 128-B block loads
//      C(i+1) argmin per window, D(i+1) leader queue
//      E4(i)  decode block -> (base, mode)
//      F(i)   codes of every window of tile i, coalesced stores
//
// The minimizer function must be non-empty and every divisor below 2^32 (launch_query_records checks; otherwise the
// default kernel runs).  Per-tile LDS state is double-buffered except the hash array, whose two uses never overlap in this order (hashes of
// tile i+1 are dead after C(i+1); the bases of tile i are written in E4(i)).  ~18 KB LDS per workgroup.
#pragma once
#include "kernels_device.cuh"

// In a persistent loop the compiler hoists every per-lane constant (4 * lane, hash-slot and LDS addresses, ...) out of
// the loop and keeps them live across all phases: 150+ VGPRs when first compiled.  Each phase therefore starts from a
// lane id the optimiser cannot see through, so that these values are recomputed where they are used (a few VALU
// instructions) and die with the phase.  (The same trick on the image pointer is harmful: header fields then come in
// through flat loads with vmcnt(0) waits, which drain the very gathers this kernel keeps in flight.)
#ifndef LPHQ_OPAQUE_LANE  // (tests/emu/hip/hip_runtime.h supplies the host version)
#define LPHQ_OPAQUE_LANE(x) asm volatile("" : "+v"(x))
#endif

namespace lphq {

__device__ __forceinline__ uint32_t phase_lane() {
    uint32_t t = threadIdx.x;
    LPHQ_OPAQUE_LANE(t);
    return t;
}

template <int BLOCK>
struct StreamTileLds {  // what phases A-D of a tile leave behind for E-F
    static constexpr int kPos = 4 * BLOCK, NW = tile_words(kPos), NBADW = (NW + 8) / 2 + 2, NDW = (NW + 1) * 4;
    static constexpr int NPASS = (NDW + BLOCK - 1) / BLOCK;  // sequence dwords per lane
    uint32_t words[NW + 8];                                // 2-bit packed bases
    uint32_t bad[NBADW];                                   // 1 bit per base: not ACGT / outside the batch
    uint16_t q[kPos + 8] __attribute__((aligned(16)));     // per window: minimizer position, 0xFFFF invalid, 0xFFFE no window
    uint16_t queue[kPos];                                  // leaders (their minimizer positions)
    uint64_t roff[3];                                      // seq_off[rlo], seq_off[rlo+1], out_off[rlo]
    uint64_t rtab[2 * kRecCap + 1];                        // multi-record tiles: the tile's slice of seq_off / out_off
    uint32_t rec[2], count;
    __device__ __forceinline__ void reset(uint32_t tid) {  // state phase A accumulates into
        for (int c = (int)tid; c < NBADW; c += BLOCK) bad[c] = (c < NDW / 8) ? 0u : 0xFFFFFFFFu;
        if (tid == 0) count = 0;
    }
};

struct PipeSeq {  // sequence dwords of the next tile, in flight
    uint32_t v[2];
};
struct PipeRec {  // record location of the next tile, in flight (lanes 0 and 64)
    uint32_t r;
    uint64_t o0, o1, o2;
};

template <int BLOCK>
__device__ __forceinline__ void pipe_issue_tile(uint64_t t0, uint32_t TW, const uint8_t* __restrict__ seq_al, uint64_t shift, uint64_t total,
                                                const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ out_off, uint32_t n_records,
                                                PipeSeq& sq, PipeRec& rc) {
    const uint32_t tid = phase_lane();
    using TL = StreamTileLds<BLOCK>;
    static_assert(TL::NPASS == 2, "two sequence dwords per lane");
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int d = (int)tid + ps * BLOCK;
        const uint64_t a = t0 + (uint64_t)d * 4u;
        sq.v[ps] = 0;
        if (d < TL::NDW && a < shift + total && a + 4u > shift) sq.v[ps] = *(const uint32_t*)(seq_al + a);
    }
    rc.r = 0;
    rc.o0 = rc.o1 = rc.o2 = 0;
    if (tid < 128) {  // wave 0: record of the first base the tile can touch; wave 1: of the last
        const uint64_t gfirst = t0 > shift ? t0 - shift : 0;
        uint64_t glast = t0 + TW + 64 > shift ? t0 + TW + 64 - shift : 0;
        if (glast >= total) glast = total ? total - 1 : 0;
        rc.r = find_record_wave(seq_off, n_records, tid < 64 ? gfirst : glast);
        if (tid == 0) {
            rc.o0 = seq_off[rc.r];
            rc.o1 = seq_off[rc.r + 1];
            rc.o2 = out_off[rc.r];
        }
    }
}

// ---- A: 4 bases per lane -> one byte of the 2-bit stream + 4 bits of the bad-base mask (t.bad / t.count were reset before) ----
template <int BLOCK>
__device__ __forceinline__ void pipe_pack(StreamTileLds<BLOCK>& t, uint64_t t0, uint64_t shift, uint64_t total, const PipeSeq& sq, const PipeRec& rc) {
    const uint32_t tid = phase_lane();
    using TL = StreamTileLds<BLOCK>;
    if (tid == 0) {
        t.rec[0] = rc.r;
        t.roff[0] = rc.o0;
        t.roff[1] = rc.o1;
        t.roff[2] = rc.o2;
    }
    if (tid == 64) t.rec[1] = rc.r;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int d = (int)tid + ps * BLOCK;
        if (d >= TL::NDW) continue;
        const uint64_t a = t0 + (uint64_t)d * 4u;  // shifted coordinate of the dword
        uint32_t p8 = 0, b4 = 0xFu;
        if (a < shift + total && a + 4u > shift) {
            p8 = pack4(sq.v[ps], b4);
            if (a < shift || a + 4u > shift + total)  // bytes outside [shift, shift+total) are not bases
                for (int i = 0; i < 4; ++i)
                    if (a + i < shift || a + i >= shift + total) b4 |= 1u << i;
        }
        ((uint8_t*)t.words)[(d & ~3) + (3 - (d & 3))] = (uint8_t)p8;  // first base in the top bits of each 32-bit word
        if (b4) atomicOr(&t.bad[d >> 3], b4 << (4 * (d & 7)));
    }
}

// ---- B: hashes of the m-mers starting at local positions 4*lane .. 4*lane+3 ----
template <int BLOCK>
__device__ __forceinline__ void pipe_hash(StreamTileLds<BLOCK>& t, uint64_t* __restrict__ sm_hash, uint32_t m, uint64_t mm_seedx,
                                          const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ out_off) {
    const uint32_t tid = phase_lane();
    if (t.rec[0] != t.rec[1]) load_record_table(t.rtab, seq_off, out_off, t.rec[0], t.rec[1], tid);  // for C and F (t.rec: visible since A's barrier)
    const uint32_t a = 4u * tid;
    const uint32_t w0 = a >> 4, o = (a & 15u) * 2u;
    const uint32_t x0 = t.words[w0], x1 = t.words[w0 + 1], x2 = t.words[w0 + 2], x3 = t.words[w0 + 3];
    const uint64_t A64 = ((uint64_t)x0 << 32) | x1, C64 = ((uint64_t)x2 << 32) | x3;
    const uint64_t s0 = o ? ((A64 << o) | ((uint64_t)x2 >> (32 - o))) : A64;  // bits [0, 64) of the stream at base a
    const uint64_t s1 = o ? (C64 << o) : C64;                                  // bits [64, 128): only the top 6 are needed
    const uint32_t sh = 64u - 2u * m;
    const uint64_t mask = m >= 32 ? ~0ull : ((1ull << (2 * m)) - 1ull);
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const uint64_t v = r == 0 ? (s0 >> sh) : (((s0 << (2 * r)) | (s1 >> (64 - 2 * r))) >> sh);
        sm_hash[r * hash_row<BLOCK>() + tid] = murmur8(v & mask, mm_seedx);
    }
    hash_pad<BLOCK>(sm_hash, tid);
}

// ---- C: leftmost minimum of 4 consecutive windows per lane + their validity -> t.q ----
template <int BLOCK>
__device__ __forceinline__ void pipe_argmin(StreamTileLds<BLOCK>& t, const uint64_t* __restrict__ sm_hash, uint64_t t0, uint32_t TW, uint32_t k, uint32_t W,
                                            uint64_t shift, uint64_t total, const uint64_t* __restrict__ seq_off) {
    const uint32_t tid = phase_lane();
    const uint32_t a = 4u * tid;
    uint32_t q[4];
    window_argmin4<BLOCK>(sm_hash, tid, W, q);
    const uint32_t rlo = t.rec[0], rhi = t.rec[1];
    const uint64_t kmask = (1ull << k) - 1ull;  // k <= 63
    const uint32_t bw = a >> 5, bo = a & 31u;
    const uint32_t b0 = t.bad[bw], b1 = t.bad[bw + 1], b2 = t.bad[bw + 2], b3 = t.bad[bw + 3];
    const uint64_t blo = bo ? (((((uint64_t)b1 << 32) | b0) >> bo) | ((uint64_t)b2 << (64u - bo))) : ((((uint64_t)b1) << 32) | b0);
    const uint32_t bhi = bo ? (uint32_t)(((((uint64_t)b3 << 32) | b2) >> bo)) : b2;
    const uint64_t end1 = t.roff[1];
    uint16_t qv[4];
#pragma unroll
    for (int jw = 0; jw < 4; ++jw) {
        const uint32_t i = a + jw;
        const uint64_t ga = t0 + i;
        uint16_t v = 0xFFFEu;  // not a window of any record
        if (i < TW && ga >= shift && ga - shift + k <= total) {
            const uint64_t g = ga - shift;
            bool inside;
            if (rlo == rhi) inside = g + k <= end1;
            else {
                uint64_t rs, re;
                tile_record(t.rtab, seq_off, rlo, rhi, g, rs, re);
                inside = g + k <= re;
            }
            if (inside) {
                const uint64_t x = jw ? ((blo >> jw) | ((uint64_t)bhi << (64 - jw))) : blo;
                v = (x & kmask) ? (uint16_t)0xFFFFu : (uint16_t)q[jw];
            }
        }
        qv[jw] = v;
    }
    *(uint2*)&t.q[a] = make_uint2((uint32_t)qv[0] | ((uint32_t)qv[1] << 16), (uint32_t)qv[2] | ((uint32_t)qv[3] << 16));
}

// ---- D: leaders -> queue (one LDS atomic per wave and round) ----
template <int BLOCK>
__device__ __forceinline__ void pipe_leaders(StreamTileLds<BLOCK>& t) {
    const uint32_t tid = phase_lane();
    const uint32_t lane = tid & 63u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = tid + r * BLOCK;
        const uint16_t qv = t.q[i];
        const bool leader = qv < 0xFFFEu && (i == 0 || t.q[i - 1] != qv);
        const uint64_t bal = __ballot(leader);
        uint32_t base = 0;
        if (lane == 0 && bal) base = atomicAdd(&t.count, (uint32_t)__popcll(bal));
        base = __shfl(base, 0, 64);
        if (leader) t.queue[base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = qv;
    }
}

// ---- F: every window of the tile, lane-contiguous stores ----
template <bool K128, int BLOCK>
__device__ __forceinline__ void pipe_store(const StreamTileLds<BLOCK>& t, const uint64_t* __restrict__ sm_hash, const uint8_t* __restrict__ sm_mode,
                                           const uint8_t* __restrict__ img, const lphq_image_header& H, uint64_t t0, uint64_t shift,
                                           const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ out_off, uint64_t* __restrict__ out) {
    const uint32_t tid = phase_lane();
    const uint32_t rlo = t.rec[0], rhi = t.rec[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t i = tid + r * BLOCK;
        const uint32_t q = t.q[i];
        if (q == 0xFFFEu) continue;
        const uint64_t g = t0 + i - shift;
        uint64_t oidx;
        if (rlo == rhi) oidx = t.roff[2] + (g - t.roff[0]);
        else {
            uint64_t rs, re;
            const uint32_t j = tile_record(t.rtab, seq_off, rlo, rhi, g, rs, re);
            oidx = tile_record_out(t.rtab, out_off, rlo, rhi, j) + (g - rs);
        }
        uint64_t hval = ~0ull;  // LPHQ_INVALID
        if (q != 0xFFFFu) {
            LookupCtx c;
            c.base = sm_hash[hslot<BLOCK>(q)];
            c.mode = sm_mode[q];
            uint64_t lo = 0, hi = 0;
            if (c.mode == 2u) extract_kmer<K128>(t.words, i, H.k, lo, hi);
            hval = finish_hval<K128, false>(img, H, c, lo, hi, q - i);
        }
        out[oidx] = hval;
    }
}

template <int KIND, int P>
struct PipeBlock;  // the gathered index block of a leader, in registers
template <int P>
struct PipeBlock<0, P> {
    UniRegs<P> r;
};
template <int P>
struct PipeBlock<1, P> {
    PairRegs r;
};

template <int KIND, bool K128, int P, int BLOCK, bool EARLY_BLOCK>
__global__ __launch_bounds__(BLOCK, 5) void k_query_tiles_stream_pipe(const uint8_t* __restrict__ img, const uint8_t* __restrict__ seq_al, uint64_t shift,
                                                                        uint64_t total, const uint64_t* __restrict__ seq_off,
                                                                        const uint64_t* __restrict__ out_off, uint32_t n_records, uint64_t* __restrict__ out,
                                                                        uint32_t n_tiles) {
    using TL = StreamTileLds<BLOCK>;
    constexpr int kPos = TL::kPos;
    __shared__ uint64_t sm_hash[hash_slots<BLOCK>()];  // B-C of tile i+1: minimizer hashes; E4-F of tile i: the leaders' bases
    __shared__ uint8_t sm_mode[kPos];
    __shared__ TL sm_tile[2];
    const lphq_image_header& H = *(const lphq_image_header*)img;
    const uint32_t TW = stream_tile_windows(H.W, kPos);
    uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // ---- prologue: phases A-D of the first tile ----
    {
        for (int b = 0; b < 2; ++b) {
            sm_tile[b].reset(threadIdx.x);
            for (int c = TL::NW + 1 + threadIdx.x; c < TL::NW + 8; c += BLOCK) sm_tile[b].words[c] = 0;  // never written again
        }
        PipeSeq sq;
        PipeRec rc;
        const uint64_t t0 = (uint64_t)tile * TW;
        pipe_issue_tile<BLOCK>(t0, TW, seq_al, shift, total, seq_off, out_off, n_records, sq, rc);
        __syncthreads();
        pipe_pack<BLOCK>(sm_tile[0], t0, shift, total, sq, rc);
        __syncthreads();
        pipe_hash<BLOCK>(sm_tile[0], sm_hash, H.m, murmur_seedx(H.mm_seed), seq_off, out_off);
        __syncthreads();
        pipe_argmin<BLOCK>(sm_tile[0], sm_hash, t0, TW, H.k, H.W, shift, total, seq_off);
        __syncthreads();
        pipe_leaders<BLOCK>(sm_tile[0]);
        __syncthreads();
    }

    uint32_t cur = 0;
#pragma clang loop unroll(disable)
    for (;; tile += gridDim.x, cur ^= 1u) {
        TL& tc = sm_tile[cur];
        TL& tn = sm_tile[cur ^ 1u];
        const uint32_t next = tile + gridDim.x;
        const bool has_next = next < n_tiles;  // uniform
        const uint32_t nlead = tc.count;
        const uint64_t t0 = (uint64_t)tile * TW, t0n = (uint64_t)next * TW;

        PipeSeq sq;
        PipeRec rc;
        if (has_next) pipe_issue_tile<BLOCK>(t0n, TW, seq_al, shift, total, seq_off, out_off, n_records, sq, rc);

        // ---- E1: the first BLOCK leaders, one per lane (more only for W < ~6: finished after E4 without overlap) ----
        // The stages are branch-free: a lane without a leader looks position 0 up again (same cache lines as its owner, every
        // address in range) and only its final LDS write is masked.  With the loads under `if (on)` the wait-count
        // insertion sees paths on which a stage's load was issued but its consumer skipped, and drains everything
        // (s_waitcnt vmcnt(0)) at the next loop header -- which is exactly the overlap this kernel exists for.
        const bool on = phase_lane() < nlead;
        const uint32_t q = on ? (uint32_t)tc.queue[phase_lane()] : 0u;
        PhfRank rk;
        uint64_t x, two;
        {
            uint64_t mm, hi;
            extract_kmer<false>(tc.words, q, H.m, mm, hi);
            x = murmur8(mm, murmur_seedx(H.mm.seed));
            two = phf_load_code<false>(img, H.mm, x, rk);
        }
        if (has_next) pipe_pack<BLOCK>(tn, t0n, shift, total, sq, rc);
        __syncthreads();

        // ---- E2 ----
        const uint64_t hp = phf_load_hp(img, H.mm, two, rk);
        if (has_next) pipe_hash<BLOCK>(tn, sm_hash, H.m, murmur_seedx(H.mm_seed), seq_off, out_off);
        __syncthreads();

        // ---- E3 ----  (EARLY_BLOCK: the block gather flies over C and D, at the price of ~28 VGPRs held across C;
        //                otherwise it is issued after C and only D separates it from its use)
        const uint32_t h = (uint32_t)phf_finish<false>(img, H.mm, x, hp);
        PipeBlock<KIND, P> blk;
        auto issue_block = [&] {
            if constexpr (KIND == 0) uni_load<P>(img, H.off_uni, h, blk.r);
            else pair_load(img, H.off_upair, h, blk.r);
        };
        if (EARLY_BLOCK) issue_block();
        if (has_next) {
            pipe_argmin<BLOCK>(tn, sm_hash, t0n, TW, H.k, H.W, shift, total, seq_off);
            __syncthreads();
        }
        if (!EARLY_BLOCK) issue_block();
        if (has_next) pipe_leaders<BLOCK>(tn);

        // ---- E4 ----
        {
            LookupCtx c;
            if constexpr (KIND == 0) {
                uint32_t type, size, pos;
                uint64_t base;
                uni_decode<P>(blk.r, H.W, h, type, base, size, pos);
                c = ctx_partitioned(H, type, base, size, pos);
            } else {
                uint64_t prefix;
                uint32_t size, pos;
                pair_decode<P>(blk.r, h, prefix, size, pos);
                c = ctx_unpartitioned(H, prefix, size, pos);
            }
            if (on) {
                sm_hash[hslot<BLOCK>(q)] = c.base;
                sm_mode[q] = (uint8_t)c.mode;
            }
        }
        for (uint32_t xq = phase_lane() + BLOCK; xq < nlead; xq += BLOCK) {
            const uint32_t q2 = tc.queue[xq];
            uint64_t mm, hi;
            extract_kmer<false>(tc.words, q2, H.m, mm, hi);
            const LookupCtx c = lookup_ctx<KIND, K128, P, false>(img, H, mm);
            sm_hash[hslot<BLOCK>(q2)] = c.base;
            sm_mode[q2] = (uint8_t)c.mode;
        }
        __syncthreads();

        // ---- F ----
        pipe_store<K128, BLOCK>(tc, sm_hash, sm_mode, img, H, t0, shift, seq_off, out_off, out);
        if (!has_next) break;
        tc.reset(phase_lane());  // this buffer is the next iteration's "next tile"
        __syncthreads();
    }
}

}  // namespace lphq
