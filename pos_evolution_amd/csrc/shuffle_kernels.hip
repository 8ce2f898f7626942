// shuffle_kernels.hip -- whole-epoch committee computation on gfx950.
//
// Replaces compute_committee (pe:495-504) over compute_shuffled_index (pe:513-534): the "swap-or-not" shuffle,
// SHUFFLE_ROUND_COUNT rounds, two SHA-256 per index and round in the literal form.  Here:
//   k_shuffle_tables   one SHA-256 per (round, 256-position block) -> the `source` hashes, plus the per-round pivot
//                      (bytes_to_uint64(hash(seed + round)[0:8]) % index_count)
//   k_shuffle_indices  one lane per list position walks its index through all rounds against those tables
//                      (flip, position = max(index, flip), bit (position % 256) of the block hash) and gathers
//                      members[i] = indices[shuffled(i)].
// Both the hash inputs and the bit addressing follow the reference text byte for byte:
//   pivot  = hash(seed + uint_to_bytes(uint8(round)))[0:8]  little-endian            (pe:522)
//   source = hash(seed + uint_to_bytes(uint8(round)) + uint_to_bytes(uint32(position // 256)))   (pe:525-529)
//   byte   = source[(position % 256) // 8] ; bit = (byte >> (position % 8)) % 2        (pe:530-531)
// Integer/hash work, no MFMA.  SURVEY.md 8(f) rank 1.
#include "kernels.h"

namespace posevo {

__device__ __constant__ uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98,
    0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786,
    0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8,
    0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
    0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a,
    0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7,
    0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// SHA-256 of a message that fits one block (<= 55 bytes), given as 16 big-endian words already padded.
__device__ __forceinline__ void sha256_one_block(const uint32_t (&w_in)[16], uint32_t (&out)[8])
{
    uint32_t w[64];
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = w_in[i];
#pragma unroll
    for (int i = 16; i < 64; ++i) {
        const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = 0x6a09e667, b = 0xbb67ae85, c = 0x3c6ef372, d = 0xa54ff53a, e = 0x510e527f, f = 0x9b05688c,
             g = 0x1f83d9ab, h = 0x5be0cd19;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        const uint32_t ch = (e & f) ^ (~e & g);
        const uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
        const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        const uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    out[0] = a + 0x6a09e667; out[1] = b + 0xbb67ae85; out[2] = c + 0x3c6ef372; out[3] = d + 0xa54ff53a;
    out[4] = e + 0x510e527f; out[5] = f + 0x9b05688c; out[6] = g + 0x1f83d9ab; out[7] = h + 0x5be0cd19;
}

// seed: 8 big-endian words.  grid: (n_blocks256 + 1) hashes per round; the extra one (b == n_blocks256) is the pivot.
__global__ void __launch_bounds__(256)
k_shuffle_tables(const uint32_t* __restrict__ seed_be, uint32_t n, uint32_t n_blocks256, uint32_t rounds,
                 uint32_t* __restrict__ source /* [rounds][n_blocks256][8] big-endian words */,
                 uint32_t* __restrict__ pivots /* [rounds] */)
{
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_round = (uint64_t)n_blocks256 + 1;
    if (gid >= per_round * rounds) return;
    const uint32_t r = (uint32_t)(gid / per_round);
    const uint32_t b = (uint32_t)(gid % per_round);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = seed_be[i];
#pragma unroll
    for (int i = 8; i < 16; ++i) w[i] = 0;
    uint32_t dig[8];
    if (b == n_blocks256) {
        // hash(seed + uint8(round)): 33 bytes -> byte 32 = round, 0x80 pad at byte 33, bit length 264
        w[8] = (r << 24) | (0x80u << 16);
        w[15] = 33 * 8;
        sha256_one_block(w, dig);
        // bytes_to_uint64(digest[0:8]) little-endian: digest bytes are the big-endian bytes of dig[0], dig[1]
        const uint64_t lo = __builtin_bswap32(dig[0]), hi = __builtin_bswap32(dig[1]);
        pivots[r] = (uint32_t)(((hi << 32) | lo) % n);
    } else {
        // hash(seed + uint8(round) + uint32_le(b)): 37 bytes; bytes 33..36 = b little-endian, 0x80 at byte 37
        w[8] = (r << 24) | ((b & 0xffu) << 16) | (((b >> 8) & 0xffu) << 8) | ((b >> 16) & 0xffu);
        w[9] = ((b >> 24) << 24) | (0x80u << 16);
        w[15] = 37 * 8;
        sha256_one_block(w, dig);
        uint32_t* dst = source + ((uint64_t)r * n_blocks256 + b) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = dig[i];
    }
}

__global__ void __launch_bounds__(256)
k_shuffle_indices(const uint32_t* __restrict__ source, const uint32_t* __restrict__ pivots, uint32_t n,
                  uint32_t n_blocks256, uint32_t rounds, const uint32_t* __restrict__ indices,
                  uint32_t* __restrict__ members)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t index = i;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t pivot = pivots[r];
        uint32_t flip = pivot + n - index;  // (pivot + index_count - index) % index_count, operands < n
        if (flip >= n) flip -= n;
        const uint32_t position = max(index, flip);
        // source[(position % 256) // 8]: byte k of the digest = byte (3 - k%4) of big-endian word k/4
        const uint32_t byte_idx = (position & 255u) >> 3;
        const uint32_t word = source[((uint64_t)r * n_blocks256 + (position >> 8)) * 8 + (byte_idx >> 2)];
        const uint32_t byte = (word >> (8 * (3 - (byte_idx & 3)))) & 0xffu;
        const uint32_t bit = (byte >> (position & 7u)) & 1u;
        index = bit ? flip : index;
    }
    members[i] = indices ? indices[index] : index;
}

// The same walk with each round's table in LDS.  The global-memory form above is bound by L2 line requests: every lane
// of a wave reads 4 bytes of a different 128-byte line (after a few rounds the 64 indices of a wave are spread over the
// whole list), 94 M requests per 1 M validators x 90 rounds = 0.3 ms.  But a round touches only HALF of its table:
//   index <= pivot: flip = pivot - index          -> position = max(index, flip) in [ceil(pivot / 2), pivot]
//   index >  pivot: flip = pivot + n - index      -> position                     in [ceil((pivot + n) / 2), n - 1]
// two contiguous ranges of n / 2 positions together = n / 16 bytes of hashes (64 KB per 1 M validators).  One workgroup
// of 1024 lanes x PER indices copies those two ranges into LDS with coalesced 16-byte loads (4x fewer line requests
// than the gather, all of them full lines that the workgroups of an XCD share in its L2), looks its indices up there,
// and goes on to the next round: 2 barriers per round.  Same arithmetic, same result.
constexpr int SHUF_WG = 1024;
constexpr size_t SHUF_LDS_CAP = 96 * 1024;
constexpr int SHUF_NQ = (int)(SHUF_LDS_CAP / 16 / SHUF_WG);  // 16-byte pieces of a round's table per lane: 6

// the two block ranges a round reads, in 256-position blocks
struct ShufRanges { uint32_t a0, len_a, b0, len_b; };
__device__ __forceinline__ ShufRanges shuffle_ranges(uint32_t pivot, uint32_t n)
{
    ShufRanges g;
    g.a0 = ((pivot + 1) >> 1) >> 8;                                  // [ceil(pivot / 2), pivot]
    g.len_a = (pivot >> 8) - g.a0 + 1;
    const uint32_t lo_b = (uint32_t)(((uint64_t)pivot + n + 1) >> 1);  // [ceil((pivot + n) / 2), n - 1]
    g.b0 = min(lo_b, n - 1) >> 8;
    g.len_b = ((n - 1) >> 8) - g.b0 + 1;
    return g;
}

// 16-byte pieces of a round's two ranges in registers: SHUF_NQ = 6 named values per lane (an indexed array, even with
// constant indices after unrolling, was kept in scratch memory by the compiler: 208 B per lane and a 3x slower kernel)
struct ShufPieces { uint4 v0, v1, v2, v3, v4, v5; };
static_assert(SHUF_NQ == 6, "ShufPieces holds six pieces");

__device__ __forceinline__ uint4 shuffle_piece(const uint4* __restrict__ src, const ShufRanges& g, uint32_t qa,
                                               uint32_t total, int j)
{
    const uint32_t q = min(threadIdx.x + j * SHUF_WG, total - 1);  // clamped: an unconditional load of a valid piece
    return src[q < qa ? 2 * (uint64_t)g.a0 + q : 2 * (uint64_t)g.b0 + (q - qa)];
}
__device__ __forceinline__ ShufPieces shuffle_prefetch(const uint32_t* __restrict__ source,
                                                       const uint32_t* __restrict__ pivots, uint32_t r, uint32_t n,
                                                       uint32_t n_blocks256)
{
    const ShufRanges g = shuffle_ranges(pivots[r], n);
    const uint4* src = reinterpret_cast<const uint4*>(source + (uint64_t)r * n_blocks256 * 8);
    const uint32_t qa = 2 * g.len_a, total = qa + 2 * g.len_b;  // 2 x uint4 per 32-byte block
    ShufPieces p;
    p.v0 = shuffle_piece(src, g, qa, total, 0);
    p.v1 = shuffle_piece(src, g, qa, total, 1);
    p.v2 = shuffle_piece(src, g, qa, total, 2);
    p.v3 = shuffle_piece(src, g, qa, total, 3);
    p.v4 = shuffle_piece(src, g, qa, total, 4);
    p.v5 = shuffle_piece(src, g, qa, total, 5);
    return p;
}

template <int PER>
__device__ __forceinline__ void shuffle_round(const ShufPieces p, uint32_t (&index)[PER], uint32_t* tab,
                                              uint32_t pivot, uint32_t n)
{
    const ShufRanges g = shuffle_ranges(pivot, n);
    const uint32_t total = 2 * (g.len_a + g.len_b);
    uint4* dst = reinterpret_cast<uint4*>(tab);
    const uint32_t t = threadIdx.x;
    if (t < total) dst[t] = p.v0;
    if (t + SHUF_WG < total) dst[t + SHUF_WG] = p.v1;
    if (t + 2 * SHUF_WG < total) dst[t + 2 * SHUF_WG] = p.v2;
    if (t + 3 * SHUF_WG < total) dst[t + 3 * SHUF_WG] = p.v3;
    if (t + 4 * SHUF_WG < total) dst[t + 4 * SHUF_WG] = p.v4;
    if (t + 5 * SHUF_WG < total) dst[t + 5 * SHUF_WG] = p.v5;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t idx = index[k];
        uint32_t flip = pivot + n - idx;
        if (flip >= n) flip -= n;
        const uint32_t position = min(max(idx, flip), n - 1);  // idx >= n (padding lanes) stays harmless
        const uint32_t blk = position >> 8;
        const uint32_t lb = position <= pivot ? blk - g.a0 : g.len_a + (blk - g.b0);
        const uint32_t byte_idx = (position & 255u) >> 3;
        const uint32_t word = tab[lb * 8 + (byte_idx >> 2)];
        const uint32_t byte = (word >> (8 * (3 - (byte_idx & 3)))) & 0xffu;
        index[k] = (idx < n && ((byte >> (position & 7u)) & 1u)) ? flip : idx;
    }
}

template <int PER>
__global__ void __launch_bounds__(SHUF_WG)
k_shuffle_indices_lds(const uint32_t* __restrict__ source, const uint32_t* __restrict__ pivots, uint32_t n,
                      uint32_t n_blocks256, uint32_t rounds, const uint32_t* __restrict__ indices,
                      uint32_t* __restrict__ members)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];  // [blocks of range A | blocks of range B] x 8 words
    const uint32_t base = blockIdx.x * (SHUF_WG * PER) + threadIdx.x;
    uint32_t index[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) index[k] = base + k * SHUF_WG;  // lane-contiguous per k: coalesced final store
    // The tables of the next TWO rounds travel from L2 into registers while this round's lookups run from LDS: a round
    // costs an LDS write, two barriers and PER lookups instead of an exposed L2 round trip.
    ShufPieces p0 = shuffle_prefetch(source, pivots, 0, n, n_blocks256);
    ShufPieces p1 = shuffle_prefetch(source, pivots, rounds > 1 ? 1 : 0, n, n_blocks256);
    for (uint32_t r = 0; r < rounds; r += 2) {
        shuffle_round<PER>(p0, index, tab, pivots[r], n);
        p0 = shuffle_prefetch(source, pivots, min(r + 2, rounds - 1), n, n_blocks256);
        __syncthreads();  // the table is overwritten by the next round
        if (r + 1 < rounds) {
            shuffle_round<PER>(p1, index, tab, pivots[r + 1], n);
            p1 = shuffle_prefetch(source, pivots, min(r + 3, rounds - 1), n, n_blocks256);
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const uint32_t i = base + k * SHUF_WG;
        if (i < n) members[i] = indices ? indices[index[k]] : index[k];
    }
}

int launch_shuffle(hipStream_t s, const uint32_t* d_seed_be, uint32_t n, uint32_t rounds, uint32_t* d_source,
                   uint32_t* d_pivots, const uint32_t* d_indices, uint32_t* d_members)
{
    if (n == 0) return 0;
    const uint32_t nb = (n + 255) / 256;
    const uint64_t hashes = ((uint64_t)nb + 1) * rounds;
    if (hashes)
        hipLaunchKernelGGL(k_shuffle_tables, dim3((unsigned)((hashes + 255) / 256)), dim3(256), 0, s, d_seed_be, n, nb,
                           rounds, d_source, d_pivots);
    // the LDS form needs n / 16 bytes (+ the ragged ends) of LDS: up to ~1.5 M indices inside 96 KB; larger (and tiny) lists
    // take the gather
    constexpr bool lds_ok = true;
    const size_t lds_bytes = 32ull * ((size_t)nb / 2 + 4);
    constexpr size_t LDS_CAP = SHUF_LDS_CAP;
    if (lds_ok && lds_bytes <= LDS_CAP && n >= 4096 && rounds > 0) {
        constexpr int PER = 4;
        if (first_use_on_this_device<4242>())
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_shuffle_indices_lds<PER>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CAP);
        hipLaunchKernelGGL(k_shuffle_indices_lds<PER>, dim3((n + SHUF_WG * PER - 1) / (SHUF_WG * PER)), dim3(SHUF_WG), lds_bytes,
                           s, d_source, d_pivots, n, nb, rounds, d_indices, d_members);
        return 0;
    }
    hipLaunchKernelGGL(k_shuffle_indices, dim3((n + 255) / 256), dim3(256), 0, s, d_source, d_pivots, n, nb, rounds,
                       d_indices, d_members);
    return 0;
}

}  // namespace posevo
