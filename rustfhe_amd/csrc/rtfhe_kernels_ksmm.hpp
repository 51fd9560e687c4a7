// rtfhe_kernels_ksmm.hpp -- the identity key switch of a whole BATCH as one exact integer contraction on the i8 matrix pipe.
//
// identity_key_switch (hom_nand/src/tlwe.rs:43-73) subtracts, per coefficient i and level l, the row KS[i][l][digit - 1] of the
// key from the result.  Inside the bootstrap kernel every gate gathers its own ~6,000 rows of 2.5 KB (9.8 MB per gate through the
// L2, 10 GB per 1,024 gates: 0.52 ms of a 7.1 ms launch, with the FP64 units idle).  For a batch the same sum is a matrix product
//
//     S[g][col] = sum_k  H[g][k] * KS[k][col] ,   k = (i, l, d),   H[g][k] = [ digit_l(a'_i of gate g) == d ]   (one-hot)
//
// in which every key row is reused by all the gates of a tile instead of being fetched per gate.  It is evaluated EXACTLY:
// each 32-bit key word is split into four signed byte limbs (w = sum_j s_j 256^j mod 2^32, s_j in [-128, 127]), the one-hot
// operand is 0 / 1 in i8, v_mfma_i32_16x16x64_i8 accumulates in i32 (|sum| <= N t 128 = 2^20 for N = 1024: no overflow), and the
// limb sums are recombined with shifts mod 2^32 -- the same torus words as the reference's row-by-row wrapping subtraction
// (u32 addition is associative and commutative).  The matrix pipe is otherwise unused on this path; nothing about the blind
// rotation changes.  This is a gather turned into a contraction to get row reuse, not a floating-point reshaping: bits are equal.
//
// K order (free, as long as both operands use it): K-step ks = 2 kk + h (kk < N/4, h < 2); lane group q = lane / 16 owns coefficient
// i = q N/4 + kk; the 16 operand bytes of a lane are t = 4 j + d: level l = 4 h + j, digit value d (d = 0: a zero key row).
// Key matrix in HBM: [colgroup = col / 16][ks][limb][lane][16 B] -- one 1 KiB dwordx4 load per MFMA operand, contiguous 4 KiB per
// K-step.  A wave owns MT = 4 tiles of 16 gates x one column group (16 columns x 4 limbs) x one K-slice: 16 MFMAs per K-step, 64 accumulator
// registers; K-slices add their parts to the (zeroed) output with wrapping u32 atomics (order-free).
// The OTHER operand -- the batch's lvl1 samples -- lies in the order this kernel reads it (ext_slot, rtfhe_kernels.hpp: tiles of 16 gates, the
// four coefficients a lane needs for two chunks are 16 contiguous bytes, a wave's 64 lanes one contiguous KiB): the extract launch writes it so.
// Round 5 measured what rows of N + 1 words cost here: every wave-load touched 64 cache lines for 1 KiB, each line re-fetched from the L2 for its
// other pieces and by each of the 40 column groups -- 13 GB of L2 requests and 2.3 GB from the fabric side per 8,192 gates, the launch waited for
// those (matrix pipe 41 % busy); tiled: 0.79 -> 0.40 ms per 8,192 gates, 0.115 -> 0.082 ms per 1,024 (profiles/r05/key_switch_mm_ab.log).
// The one-hot operand of a K-step is a function of ONE byte of a gate's rounded coefficient (four 2-bit digits -> four words with one byte
// set): it is read from a 256-entry table in LDS (one ds_read_b128) instead of being built with 16 shifts and masks per gate tile and K-step.
// (Building it with v_cvt_pk_u8_f32 -- 8 instructions per operand, no table -- measured 2-4 % slower than the table, whose lookups conflict on
// banks in 46 % of the LDS cycles without the LDS being what the launch waits for; same log.)
#pragma once

#include "rtfhe_kernels.hpp"

namespace rtfhe {

struct KsMmArgs {
    const uint32_t* tlwe1;   // extracted lvl1 samples a'[0..N), b' of `count` gates in tiles of 16 (ext_slot, rtfhe_kernels.hpp)
    const uint4* kmat;       // [colgroups][N/2 K-steps][4 limbs][64 lanes] x 16 B
    uint32_t* out;           // row g (plain batch) or row idx_out[g] (netlist wave) of [..][n+1]; ZERO on entry (the extract launch
                             // zeroes it): every K-slice adds its part with a wrapping atomic (order-free in u32)
    int32_t count, n, N, colgroups;
    int32_t splitk;          // K-slices per (gate group, column group): N/4 must be divisible by 4 * splitk
    int32_t mgroups;         // workgroups along the gates: ceil(count / (64 * KSMM_WAVES))
    // netlist wave (all null for a plain batch): the same validity rule as gate_io -- a gate the bootstrap launch skipped is skipped here
    const int32_t* ops; const int32_t* idx0; const int32_t* idx1; const int32_t* idx_out;
    int32_t num_wires;
};

typedef int v4i_t __attribute__((ext_vector_type(4)));

// A workgroup is KSMM_WAVES waves: each owns MT = 4 gate tiles (64 gates) and ALL of them walk the same K-slice of the same column group, so a
// K-step's key operand is fetched ONCE per workgroup -- cooperatively, a chunk of 4 K-steps (16 KiB) at a time into one of two LDS buffers while
// the other is being multiplied -- and read from LDS by every wave.  (Round 4's one-wave workgroups each streamed the panel themselves: 0.43 GB
// from the fabric side for an 84 MB operand per 1,024-gate launch, and the launch waited for exactly that; round 5.)
// grid: x = ((slice * ceil(colgroups / 8) + cg / 8) * mgroups + mgb) * 8 + cg % 8, mgb = group of 64 * KSMM_WAVES gates: workgroups go to the
// XCDs round-robin, so the mgroups workgroups that walk one (column group, slice) panel are neighbours in dispatch order ON ONE XCD -- they start
// together, keep step (same work each) and the panel comes from HBM once, the others' reads hit that XCD's L2.
constexpr int KSMM_WAVES = 8;
constexpr int KSMM_CHUNK = 4;         // K-steps per LDS chunk (= 2 coefficients per lane group)
template <int KS_T, int KS_BB>
__global__ __launch_bounds__(64 * KSMM_WAVES, 1) void k_key_switch_mm(const KsMmArgs a) {
    static_assert(KS_T == 8 && KS_BB == 2, "operand packing: 4 levels x 4 digit values per 16-byte operand chunk");
    constexpr int MT = 4, W = KSMM_WAVES, KC = KSMM_CHUNK, NT = 64 * W;
    constexpr int CHUNK_V4 = KC * 4 * 64;                 // uint4 per chunk
    constexpr int PER_THREAD = CHUNK_V4 / NT;             // uint4 a thread stages per chunk
    static_assert(CHUNK_V4 % NT == 0, "whole uint4 per thread");
    constexpr uint32_t ROUND = 1u << (32 - KS_T * KS_BB - 1);
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg8 = (a.colgroups + 7) >> 3, rest = blockIdx.x >> 3, panel = rest / a.mgroups;
    const int cg = (panel % cg8) * 8 + (blockIdx.x & 7), slice = panel / cg8;
    if (cg >= a.colgroups) return;                                              // (the column groups padded to whole XCD rounds)
    const int mg = (rest % a.mgroups) * W + wave;                               // this wave's group of 64 gates
    const int quarter = a.N / 4, ksteps = a.N / 2;
    const int kk_begin = quarter / a.splitk * slice, kk_end = kk_begin + quarter / a.splitk;     // coefficients (per lane group) of this slice
    // one-hot operands by digit byte: entry b = { 1 << 8 ((b >> 6) & 3), 1 << 8 ((b >> 4) & 3), 1 << 8 ((b >> 2) & 3), 1 << 8 (b & 3) }
    __shared__ v4i_t onehot[256];
    __shared__ v4i_t kbuf[2][CHUNK_V4];                   // [buffer][K-step of the chunk][limb][lane]
    for (int b = tid; b < 256; b += NT)
        onehot[b] = (v4i_t){(int)(1u << (((b >> 6) & 3) * 8)), (int)(1u << (((b >> 4) & 3) * 8)), (int)(1u << (((b >> 2) & 3) * 8)), (int)(1u << ((b & 3) * 8))};
    v4i_t acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[mt][j] = (v4i_t){0, 0, 0, 0};
    // The samples lie in tiles of 16 gates (ext_slot, rtfhe_kernels.hpp): the four coefficients kk .. kk + 3 of a tile's 64 (q, gate) rows are
    // one contiguous KiB, lane-linear -- read through a buffer resource with the tile and the coefficient group in the scalar offset.  A tile past
    // the batch reads the last one (its rows are never written); the rows past the batch inside the last tile are allocated, whatever they hold.
    const int ntiles = (a.count + 15) >> 4, tile_bytes = (int)(ext_tile_words(a.N) * 4);
    const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.tlwe1), 0, ntiles * tile_bytes, 0x00020000);
    int tile_soff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) tile_soff[mt] = __builtin_amdgcn_readfirstlane(min(mg * MT + mt, ntiles - 1) * tile_bytes);
    const int lane16 = lane * 16;
    // The chunk of K-steps [ks0, ks0 + KC) of this column group's panel is CHUNK_V4 consecutive uint4: thread t stages uint4 t, t + NT, ...
    // through a buffer resource (scalar chunk offset + one lane VGPR + immediates).
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.kmat + (size_t)cg * ksteps * 4 * 64), 0, 0x7fffffff, 0x00020000);
    const int tid16 = tid * 16;
    v4u stage[PER_THREAD];
    auto fetch_chunk = [&](int ks0) {
        const int soff = __builtin_amdgcn_readfirstlane(ks0 * 4096);
#pragma unroll
        for (int p = 0; p < PER_THREAD; p++) stage[p] = __builtin_amdgcn_raw_buffer_load_b128(krsrc, tid16 + p * NT * 16, soff, 0);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PER_THREAD; p++) kbuf[buf][tid + p * NT] = (v4i_t){(int)stage[p].x, (int)stage[p].y, (int)stage[p].z, (int)stage[p].w};
    };
    const int nchunks = (kk_end - kk_begin) * 2 / KC;
    fetch_chunk(2 * kk_begin);
    store_chunk(0);
    __syncthreads();
    uint4 aw[MT];
#pragma unroll 1
    for (int c = 0; c < nchunks; c++) {
        const int kk2 = kk_begin + c * (KC / 2);          // first of the chunk's two coefficients (per lane group)
        if (c + 1 < nchunks) fetch_chunk(2 * kk2 + KC);   // the next chunk: in flight under this chunk's multiplies
        if ((c & 1) == 0) {                               // gate words of four coefficients at a time (two chunks)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const v4u w = __builtin_amdgcn_raw_buffer_load_b128(srsrc, lane16, tile_soff[mt] + __builtin_amdgcn_readfirstlane(kk2 * 256), 0);
                aw[mt] = make_uint4(w.x, w.y, w.z, w.w);
            }
        }
        const v4i_t* kb = kbuf[c & 1] + lane;
#pragma unroll
        for (int t = 0; t < KC; t++) {                    // K-step 2 kk + h: coefficient e = t / 2 of the chunk, byte h = t % 2
            const int h = t & 1;
            v4i_t b[4];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = kb[(t * 4 + j) * 64];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const uint32_t w0 = (c & 1) ? ((t >> 1) ? aw[mt].w : aw[mt].z) : ((t >> 1) ? aw[mt].y : aw[mt].x);
                const uint32_t byte8 = ((w0 + ROUND) >> (24 - 8 * h)) & 0xffu;         // levels 4h .. 4h+3, most significant first
                const v4i_t av = onehot[byte8];
#pragma unroll
                for (int j = 0; j < 4; j++) acc[mt][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[j], acc[mt][j], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);    // (that buffer's last readers finished chunk c - 1 before the barrier below of iteration c - 1)
        __syncthreads();
    }
    // D layout of a 16x16 i32 tile: lane holds column lane % 16, rows 4 (lane / 16) + r in register r
    const int col = cg * 16 + r16;
    if (col > a.n) return;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int g = (mg * MT + mt) * 16 + 4 * q + r;
            if (g >= a.count) continue;
            size_t row = (size_t)g;
            if (a.idx_out) {
                const int i0 = a.idx0[g], i1 = a.idx1[g], o = a.idx_out[g];
                const unsigned nw = (unsigned)a.num_wires;
                if (!((unsigned)i0 < nw && (unsigned)i1 < nw && (unsigned)o < nw && (unsigned)a.ops[g] <= (unsigned)OP_ANDNY)) continue;
                row = (size_t)o;
            }
            const uint32_t s = (uint32_t)acc[mt][0][r] + ((uint32_t)acc[mt][1][r] << 8) + ((uint32_t)acc[mt][2][r] << 16) + ((uint32_t)acc[mt][3][r] << 24);
            const uint32_t bprime = (col == a.n && slice == 0) ? a.tlwe1[(size_t)(g >> 4) * ext_tile_words(a.N) + 16 * a.N + (g & 15)] : 0u;
            atomicAdd(a.out + row * ((size_t)a.n + 1) + col, bprime - s);
        }
    }
}

// key matrix from the reference-order rows: raw[(i * t + l) * (base - 1) + d - 1][ksw]  ->  signed byte limbs in operand order
struct KsMatArgs {
    const uint32_t* raw;     // [N * t * (base-1) (+1)][ksw]
    uint4* kmat;
    int32_t N, n, ksw, colgroups;
};
template <int KS_T, int KS_BB>
__global__ __launch_bounds__(256) void k_ksmat_build(const KsMatArgs a) {
    const int ksteps = a.N / 2, quarter = a.N / 4;
    const size_t total = (size_t)a.colgroups * ksteps * 4 * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63), limb = (int)((idx >> 6) & 3);
        const size_t rest = idx >> 8;
        const int ks = (int)(rest % ksteps), cg = (int)(rest / ksteps);
        const int col = cg * 16 + (lane & 15), q = lane >> 4, kk = ks >> 1, h = ks & 1, i = q * quarter + kk;
        uint32_t out[4] = {0, 0, 0, 0};
        if (col <= a.n) {
            for (int t = 0; t < 16; t++) {
                const int l = 4 * h + (t >> 2), d = t & 3;
                if (d == 0) continue;
                uint32_t w = a.raw[((size_t)(i * KS_T + l) * ((1 << KS_BB) - 1) + (d - 1)) * a.ksw + col];
                int32_t s = 0;
                for (int j = 0; j <= limb; j++) {          // balanced base-256 digits, least significant first; the last carry drops mod 2^32
                    s = (int32_t)(int8_t)(w & 0xffu);
                    w = (w - (uint32_t)s) >> 8;
                }
                out[t >> 2] |= ((uint32_t)s & 0xffu) << (8 * (t & 3));
            }
        }
        a.kmat[idx] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

}  // namespace rtfhe
