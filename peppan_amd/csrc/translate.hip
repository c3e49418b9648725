// K1: nucleotide -> protein on the GPU.
//   query side : three forward frames per gene, keep the frame with the fewest stop/ambiguous-separated
//                segments (ties -> lowest frame)                      [replaces uberBlast.py:525-529]
//   reference  : 6 (or 3) frames per sequence, each frame string cut after the first stop at or beyond
//                1000 residues from the chunk start                    [replaces uberBlast.py:535-544]
//   both       : residues written as codes (letter - 'A') into the padded packed layout that the seed
//                and Smith-Waterman kernels read (16-byte aligned sequence starts, >= 16 pad bytes of
//                code 31 between sequences, 64 at both ends).          [translation: configure.py:160-194]
// Byte-granular, HBM-bound: 3 nt bytes read per residue byte written.
#include "common.h"
#include "lookback.h"
#include <algorithm>

namespace {

// codon index a<<4 | b<<2 | c with A0 C1 G2 T3 -> residue code; stops are X (23), as in the reference's table
__constant__ uint8_t c_codon[2][64];

__device__ __forceinline__ int base2(uint8_t ch)
{
    // A 0x41, C 0x43, G 0x47, T 0x54 (case folded): bits 1 and 2 spell 0..3 in that order; then check that it really was that letter
    const uint32_t u = ch & 0xDFu;
    const int code = (int)(((u >> 1) ^ (u >> 2)) & 3u);
    if (u == ((0x54474341u >> (code * 8)) & 0xFFu)) return code;
    return ch == '-' ? -2 : -1;
}

// four characters at once (one per byte of w): their codes packed two bits each, first character (byte 0) in bits 7:6 - complemented when
// `rev` - and a mask of the characters that are not A, C, G, T in either case, first character = bit 3.  Same rule as base2.
__device__ __forceinline__ void nt4(uint32_t w, bool rev, uint32_t &pk, uint32_t &bad)
{
    const uint32_t u = w & 0xDFDFDFDFu;
    uint32_t code = ((u >> 1) ^ (u >> 2)) & 0x03030303u;
    const uint32_t diff = u ^ __builtin_amdgcn_perm(0u, 0x54474341u, code);            // byte i: the letter that code i stands for
    const uint32_t nz = ((((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) >> 7) & 0x01010101u;
    if (rev) code ^= 0x03030303u;
    pk = (code * 0x40100401u) >> 24;
    bad = (nz * 0x08040201u) >> 24;
}

// residue `aa` of frame `frame` (1..6) of the sequence nt[0..L): returns code 0..25, X for ambiguous/partial codons;
// *is_gap set when the codon contains '-' (the reference emits '-', which is not an 'X' for frame choice)
__device__ __forceinline__ int translate_at(const uint8_t *__restrict__ nt, int L, int frame, int aa, int tab, bool *is_gap)
{
    // 32-bit indices inside one sequence (upload_nt keeps a sequence below 2^31 - 256 nucleotides): with 64-bit positions half of
    // the kernel's instructions were address arithmetic
    int b[3];
    const int p0 = (frame <= 3 ? frame - 1 : frame - 4) + 3 * aa;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = p0 + k;
        int v = -1;
        if (p < L) {
            if (frame <= 3) v = base2(nt[p]);
            else { v = base2(nt[L - 1 - p]); if (v >= 0) v = 3 - v; }
        }
        b[k] = v;
    }
    const bool gap = (b[0] == -2) | (b[1] == -2) | (b[2] == -2);
    if (is_gap) *is_gap = gap;
    if (gap || (b[0] | b[1] | b[2]) < 0) return 23;
    return c_codon[tab][(b[0] << 4) | (b[1] << 2) | b[2]];
}

__device__ __forceinline__ int64_t frame_len(int64_t L, int frame)
{
    const int64_t rem = L - (frame <= 3 ? frame - 1 : frame - 4);
    return rem > 0 ? (rem + 2) / 3 : 0;
}

// What the host needs of a packed set before it can queue a search: computed on the device (k1_offsets) and downloaded behind the
// descriptors (two descriptor slots), so that the host does not have to walk 60 k descriptors while the GPU waits for its next kernel -
// the per-sequence tables (h_off, h_len, the meta records) are built from the downloaded descriptors only when somebody asks for them.
struct K1Summary {
    unsigned long long residues, total;
    uint32_t n, max_len;
    uint32_t pad[2];
};
static_assert(sizeof(K1Summary) == 2 * 16, "the summary takes two descriptor slots");

// (PackDesc - one per packed sequence: where its residues come from - is common.h's: K7 behind a search reads the descriptors too)

// where a packed sequence's nucleotides lie (one record per descriptor slot, written by the kernel that writes the descriptor): k1_pack
// fetches it together with the descriptor instead of chasing nt_off[desc.seq] behind it - a wave of k1_pack lives for a chain of
// dependent loads, and this takes a link out of it
struct K1Src { uint64_t off; uint32_t L, pad; };
static_assert(sizeof(K1Src) == sizeof(PackDesc), "source records lie behind the descriptors, in slots of the same size");
__host__ __device__ __forceinline__ K1Src *k1_src_of(const PackDesc *desc, uint32_t cap) { return reinterpret_cast<K1Src *>(const_cast<PackDesc *>(desc) + cap + 3); }

__device__ __forceinline__ uint32_t padded_len(uint32_t len) { return (len + 15u) / 16u * 16u + PEP_SEQ_GAP; }

// one wavefront per query gene; the gene's descriptor, source record, padded length and (in pinned memory, for the host) length are written here too
__global__ __launch_bounds__(256) void k1_query_frames(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ off, uint32_t n, int tab,
                                                       PackDesc *__restrict__ desc, uint32_t *__restrict__ padded, uint32_t *__restrict__ len_out, uint32_t *__restrict__ n_out,
                                                       K1Summary *__restrict__ sum, uint32_t *__restrict__ pin_len, K1Src *__restrict__ src_of)
{
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) { *n_out = n; sum->residues = 0ull; sum->max_len = 0u; }       // (k1_offsets accumulates into it)
    if (g >= n) return;
    const uint64_t o = off[g];
    const uint8_t *s = nt + o;
    const int L = (int)(off[g + 1] - o);           // (upload_nt keeps a sequence below 2^31 - 256 nucleotides)
    // Same geometry as k1_pack: the wavefront copies the 1536 + 2 nucleotide bytes behind 512 codon positions into LDS with aligned dword
    // loads, then every lane looks at eight positions.  The three frames are walked together: codon a of frames 1, 2, 3 is bytes 3a .. 3a+4.
    // The first version read its bytes from global memory five at a time and looked the codons up in the constant array with per-lane
    // indices, i.e. a chain of two global round trips per 128 positions; the second judged the lane's 26 characters one by one (600 vector
    // instructions per chunk).  Now: four characters per register (nt4), and "is this codon a stop" is a bit of a 64-bit mask.
    __shared__ uint32_t stage_all[4][392];
    uint32_t *stage = stage_all[threadIdx.x >> 6];
    const unsigned long long stop_codon = __ballot(c_codon[tab][lane] == 23);
    const int na1 = (int)frame_len(L, 1), na2 = (int)frame_len(L, 2), na3 = (int)frame_len(L, 3);
    uint32_t x1 = 0, x2 = 0, x3 = 0;
    for (int c0 = 0; c0 + 1 < na1; c0 += 512) {                // s[:-1] of every frame; frame 1 is the longest
        const int lo = 3 * c0;
        const uintptr_t A = reinterpret_cast<uintptr_t>(s + lo);
        const int shift = (int)(A & 3u);
        const uint32_t *al = reinterpret_cast<const uint32_t *>(A - shift);
        const int n_dw = (min(1538, L - lo) + shift + 3) >> 2;            // <= 386; the buffer is padded by 64 bytes behind the last sequence
        for (int x = lane; x < n_dw; x += 64) stage[x] = al[x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // the lane's characters 0 .. 27 (26 are used): codes two bits each (character j at bits 55-2j, 54-2j), one bit each for "not A, C, G, T"
        // and for '-' (character j at bit 27-j)
        uint32_t raw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) raw[k] = stage[6 * lane + k];
        unsigned long long codes = 0;
        uint32_t bad = 0, gap = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint32_t w = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)shift);
            uint32_t pk, bd;
            nt4(w, false, pk, bd);
            const uint32_t g = w ^ 0x2D2D2D2Du;
            const uint32_t is_gap = (~((((g & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | g)) >> 7) & 0x01010101u;
            codes |= (unsigned long long)pk << (48 - 8 * k);
            bad |= bd << (24 - 4 * k);
            gap |= ((is_gap * 0x08040201u) >> 24) << (24 - 4 * k);
        }
        const int nv = min(max(L - (lo + 24 * lane), 0), 28);
        bad |= (1u << (28 - nv)) - 1u;                                    // characters behind the sequence's end
        gap &= ~((1u << (28 - nv)) - 1u);
        const int lim[3] = {na1, na2, na3};
        uint32_t *cnt[3] = {&x1, &x2, &x3};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int a = c0 + 8 * lane + h;
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                const int j = 3 * h + f;
                const uint32_t idx = (uint32_t)(codes >> (50 - 2 * j)) & 63u;
                const uint32_t b3 = (bad >> (25 - j)) & 7u, g3 = (gap >> (25 - j)) & 7u;
                const bool stop = g3 == 0u && (b3 != 0u || ((stop_codon >> idx) & 1ull));
                *cnt[f] += (a + 1 < lim[f] && stop) ? 1u : 0u;
            }
        }
        __builtin_amdgcn_wave_barrier();                                 // the next chunk overwrites the staging area
    }
    for (int d = 32; d > 0; d >>= 1) { x1 += __shfl_xor(x1, d, 64); x2 += __shfl_xor(x2, d, 64); x3 += __shfl_xor(x3, d, 64); }
    uint32_t best_cnt = x1, best_f = 1, best_len = (uint32_t)na1;
    if (x2 < best_cnt) { best_cnt = x2; best_f = 2; best_len = (uint32_t)na2; }
    if (x3 < best_cnt) { best_cnt = x3; best_f = 3; best_len = (uint32_t)na3; }
    if (lane == 0) {
        desc[g] = PackDesc{g, best_f, 0u, best_len};
        src_of[g] = K1Src{o, (uint32_t)L, 0u};
        padded[g] = padded_len(best_len);
        len_out[g] = best_len;
        pin_len[g] = best_len;                   // the host's copy (pinned memory): the search derives its score thresholds from the lengths
    }
}

// The chunks of a reference frame (uberBlast.py:539-544: regex `.{1000,}?X|.{1,1000}$` over the frame string + 'X'): a chunk that starts at c0 ends with the first stop at
// or behind c0 + 1000, or with the string.  chunk_end: that position, found by the whole wavefront 64 characters at a time.
__device__ __forceinline__ int64_t chunk_end(const uint8_t *s, int64_t L, int f, int tab, int64_t na, int64_t c0, int lane)
{
    if (na + 1 - c0 < 1001) return na;
    int64_t x0 = c0 + 1000;
    for (;;) {
        const int64_t a = x0 + lane;
        bool isx = false;
        if (a < na) { bool gap; isx = (translate_at(s, (int)L, f, (int)a, tab, &gap) == 23) && !gap; }
        else if (a == na) isx = true;
        const unsigned long long m = __ballot(isx);
        if (m) return x0 + (int64_t)__ffsll((long long)m) - 1;
        x0 += 64;
    }
}

// A chunk's start follows from the chunk in front of it: one wavefront walking a 7.7 Mb contig's frame this way takes 2 500 dependent round trips to memory - 4 ms per
// search of a mapping batch at 50 000 exemplars, a third of the batch's GPU time, whatever the number of contigs.  (Chains that start at different places do merge - two
// chunks whose starts + 1000 fall between the same two stops end with the same stop - and a first version walked segments speculatively and joined them to the true chain:
// in frames with a stop every ~21 characters two chains keep their distance for tens of chunks, the join did 40 % of the serial work again: 1.65 ms.)  So the chain stays
// serial, and what a step costs is taken out of it.  Frames of more than K1_LONG characters:
//   k1_stop_mask    every character of the frame + 'X' as one bit "is a stop" (64 per word, a wavefront per 4 096 characters, all frames side by side);
//   k1_ref_chunks_mask   one wavefront per frame keeps a WINDOW of the mask in registers - 256 words, 16 384 characters, four words per lane - and the next window on
//                   its way; "the first stop at or behind c0 + 1000" is a mask, a ballot and a count of trailing zeros: no memory in a step, sixteen steps per window.
constexpr int64_t K1_LONG = 98304;               // characters of a frame (+ 'X') from which its chunks are found through the stop mask
struct K1Seg { uint32_t w, seg; };               // (frame, tile of 4 096 characters / of 256 chunk slots)
constexpr uint32_t K1_FILL_FROM = 128;           // frames with this many chunks or more get their descriptors from k1_ref_desc_fill.  Only the long frames have tiles: a frame with 128 chunks holds more than 128 x 1001 characters
static_assert((int64_t)K1_FILL_FROM * 1001 > K1_LONG, "a frame that leaves its descriptors to k1_ref_desc_fill is a long one");
struct K1Long { uint32_t w, pad; uint64_t mask_off; };          // a long frame: its words of the stop mask start at mask_off

// "character aa of the frame is a stop" as the chunking sees it (translate_at(...) == 23 and no gap in the codon: a stop codon, or an ambiguous / partial one) without the
// dependent look-up in the codon table: stop_codons = bit idx set for the codons the table turns into 'X' (a ballot over the table's 64 entries)
__device__ __forceinline__ bool stop_at(const uint8_t *__restrict__ nt, int L, int frame, int aa, unsigned long long stop_codons)
{
    int b[3];
    const int p0 = (frame <= 3 ? frame - 1 : frame - 4) + 3 * aa;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = p0 + k;
        int v = -1;
        if (p < L) {
            if (frame <= 3) v = base2(nt[p]);
            else { v = base2(nt[L - 1 - p]); if (v >= 0) v = 3 - v; }
        }
        b[k] = v;
    }
    if ((b[0] == -2) | (b[1] == -2) | (b[2] == -2)) return false;
    if ((b[0] | b[1] | b[2]) < 0) return true;
    return (stop_codons >> ((b[0] << 4) | (b[1] << 2) | b[2])) & 1ull;
}

__global__ __launch_bounds__(256) void k1_stop_mask(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ off, int n_frames, int tab, const K1Seg *__restrict__ tiles,
                                                    uint32_t n_tiles, const K1Long *__restrict__ longs, const uint32_t *__restrict__ long_of_w, unsigned long long *__restrict__ mask)
{
    const int lane = threadIdx.x & 63;
    const uint32_t x = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= n_tiles) return;
    const K1Seg tl = tiles[x];
    const uint32_t g = tl.w / (uint32_t)n_frames;
    const int f = (int)(tl.w % (uint32_t)n_frames) + 1;
    const uint8_t *s = nt + off[g];
    const int64_t L = (int64_t)(off[g + 1] - off[g]);
    const int64_t na = frame_len(L, f);
    unsigned long long *out = mask + longs[long_of_w[tl.w]].mask_off;
    const int64_t n_words = (na + 1 + 63) >> 6;
    const unsigned long long stop_codons = __ballot(c_codon[tab][lane] == 23);
    // four words per trip: their twelve byte reads per lane are in flight together (one word per trip was a round trip to memory per 64 characters)
    for (int64_t wd = (int64_t)tl.seg * 64; wd < (int64_t)(tl.seg + 1) * 64 && wd < n_words; wd += 4) {
        bool isx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t a = (wd + u) * 64 + lane;
            isx[u] = a < na ? stop_at(s, (int)L, f, (int)a, stop_codons) : a == na;      // (a == na: the 'X' the reference appends)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long m = __ballot(isx[u]);
            if (lane == 0 && wd + u < n_words) out[wd + u] = m;
        }
    }
}

// one wavefront per LONG frame; chunk_off / chunk_len / chunk_cnt as k1_ref_chunks writes them
__global__ __launch_bounds__(256) void k1_ref_chunks_mask(const uint64_t *__restrict__ off, int n_frames, const K1Long *__restrict__ longs, uint32_t n_long,
                                                          const unsigned long long *__restrict__ mask, const uint64_t *__restrict__ chunk_base,
                                                          uint32_t *__restrict__ chunk_cnt, uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ chunk_len)
{
    const int lane = threadIdx.x & 63;
    const uint32_t x = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= n_long) return;
    const K1Long lg = longs[x];
    const uint32_t g = lg.w / (uint32_t)n_frames;
    const int f = (int)(lg.w % (uint32_t)n_frames) + 1;
    const int64_t L = (int64_t)(off[g + 1] - off[g]);
    const int64_t na = frame_len(L, f);
    const int64_t n_words = (na + 1 + 63) >> 6;
    const unsigned long long *mk = mask + lg.mask_off;
    uint32_t *starts = chunk_off + chunk_base[lg.w];               // the chunk starts in order (their lengths are taken from them at the end)
    // the window: ROWS rows of 64 words, word (win + 64 r + lane) in cur[r]; positions are 32-bit (a sequence stays below 2^31 nucleotides).  A step looks at ONE row as a
    // rule - a single wavefront issues its instructions one behind the other: with all four rows judged in every step a step took 0.54 us, the walk 1.35 ms
    constexpr int ROWS = 4, WIN = 64 * ROWS;
    const int n_w = (int)n_words, n_a = (int)na;
    // (the reads of a window are unconditional - the index clamped, words behind the mask's end zeroed when the window is TAKEN: a read under a lane condition is a branch, and
    // the compiler waits for every read in flight where branches meet - the window that should have been on its way for sixteen steps was waited for on the spot)
    auto load = [&](int base, unsigned long long *dst) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) dst[r] = mk[min(base + 64 * r + lane, n_w - 1)];
    };
    auto take = [&](int base, const unsigned long long *src, unsigned long long *dst) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) dst[r] = base + 64 * r + lane < n_w ? src[r] : 0ull;
    };
    unsigned long long cur[ROWS], nxt[ROWS];
    int win = 0;
    load(0, nxt);
    take(0, nxt, cur);
    load(WIN, nxt);
    uint32_t cnt = 0;
    int c0 = 0;
    while (c0 < n_a + 1) {
        if (lane == 0) starts[cnt] = (uint32_t)c0;
        ++cnt;
        int end = n_a;
        if (n_a + 1 - c0 >= 1001) {
            int p = c0 + 1000;                                      // the first stop at or behind p (bit na is set: there is one)
            for (;;) {
                int wi = p >> 6;
                if (wi >= win + 2 * WIN) { win = wi; load(win, nxt); take(win, nxt, cur); load(win + WIN, nxt); }     // (a chunk longer than a window: start over where it ends)
                else if (wi >= win + WIN) {
                    win += WIN;
                    take(win, nxt, cur);
                    load(win + WIN, nxt);                           // on its way while the next sixteen chunks are cut from cur
                }
                const int r0 = (wi - win) >> 6, j = (wi - win) & 63; // row and lane of the word that holds p
                unsigned long long v = r0 == 0 ? cur[0] : r0 == 1 ? cur[1] : r0 == 2 ? cur[2] : cur[3];
                v = lane > j ? v : (lane == j ? v & (~0ull << (p & 63)) : 0ull);
                const unsigned long long any = __ballot(v != 0ull);
                if (any) {
                    const int l = __ffsll((long long)any) - 1;
                    const unsigned long long word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
                    end = ((win + 64 * r0 + l) << 6) + (int)__builtin_ctzll(word);
                    break;
                }
                p = (win + 64 * (r0 + 1)) << 6;                     // no stop in the rest of this row: on with the next row (or window)
            }
        }
        c0 = end + 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // lengths: a chunk reaches up to the start of the next one; the last one ends with the string (the appended 'X' does not count) and is dropped when nothing is left
    uint32_t *lens = chunk_len + chunk_base[lg.w];
    uint32_t kept = cnt;
    for (uint32_t i = (uint32_t)lane; i < cnt; i += 64) {
        const uint32_t s0 = __builtin_nontemporal_load(&starts[i]);
        lens[i] = i + 1 < cnt ? __builtin_nontemporal_load(&starts[i + 1]) - s0 : (uint32_t)(na - (int64_t)s0);
    }
    if (cnt && (int64_t)__builtin_nontemporal_load(&starts[cnt - 1]) >= na) kept = cnt - 1;
    if (lane == 0) chunk_cnt[lg.w] = kept;
}

// one wavefront per (reference sequence, frame): chunk boundaries.  chunk_base[w] = first slot of this frame's chunk list.
__global__ __launch_bounds__(256) void k1_ref_chunks(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ off, uint32_t n, int n_frames, int tab,
                                                     const uint64_t *__restrict__ chunk_base, uint32_t *__restrict__ chunk_cnt,
                                                     uint32_t *__restrict__ chunk_off, uint32_t *__restrict__ chunk_len, int64_t long_from)
{
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= (uint64_t)n * n_frames) return;
    const uint32_t g = (uint32_t)(w / n_frames);
    const int f = (int)(w % n_frames) + 1;
    const uint8_t *s = nt + off[g];
    const int64_t L = (int64_t)(off[g + 1] - off[g]);
    const int64_t na = frame_len(L, f);         // the frame string; the reference appends one more 'X' at index na
    if (na + 1 > long_from) return;             // a long frame: k1_stop_mask / k1_ref_chunks_mask
    const uint64_t base = chunk_base[w];
    uint32_t cnt = 0;
    int64_t c0 = 0;
    while (c0 < na + 1) {
        const int64_t end = chunk_end(s, L, f, tab, na, c0, lane);      // index of the chunk's last character in s + 'X'
        int64_t len = end - c0 + 1;
        if (end == na) --len;                    // drop the appended 'X'
        if (len > 0) {
            if (lane == 0) { chunk_off[base + cnt] = (uint32_t)c0; chunk_len[base + cnt] = (uint32_t)len; }
            ++cnt;
        }
        c0 = end + 1;
    }
    if (lane == 0) chunk_cnt[w] = cnt;
}

// One wavefront per packed sequence: lanes stride over its residues (adjacent lanes read adjacent codons - 192 contiguous
// nucleotide bytes per step - and store 64 contiguous residue bytes), then over the padding up to the next sequence, and fill
// the block -> sequence map of the blocks it owns.  The number of packed sequences lives on the device (*n_ptr): the grid is
// sized from a host upper bound and the surplus waves leave at once, so the host never waits for the chunk count.
// (The first version ran one thread per 16-byte block of the layout with a 16-step binary search for the owner: 0.22 ms for
// the 20 M reference residues, bound by the latency of that search.)
__global__ __launch_bounds__(256) void k1_pack(const uint8_t *__restrict__ nt, const K1Src *__restrict__ src_of, int tab,
                                               const PackDesc *__restrict__ desc, const uint32_t *__restrict__ pk_off, const uint32_t *__restrict__ n_ptr, uint32_t cap,
                                               uint8_t *__restrict__ res, uint2 *__restrict__ blk2seq, const K1Summary *__restrict__ d_sum, K1Summary *__restrict__ pin_sum)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    // the set's summary (k1_offsets finished it) goes to the host from here: a 32-byte store into pinned memory instead of a copy command,
    // which would cost the GPU 10 us of idle time in front of it
    if (blockIdx.x == 0 && threadIdx.x == 0) *pin_sum = *d_sum;
    // the count, the descriptor and the offsets are fetched together (slot s exists in all three arrays whether or not it is in use):
    // a wave lives for a chain of dependent loads, and this takes one link out of it
    const uint32_t n_packed = *n_ptr;
    const uint32_t sc = min(s, cap);                                   // (the last block's surplus waves stay inside the arrays)
    const PackDesc d = desc[sc];
    const K1Src so = src_of[sc];
    const uint32_t start = pk_off[sc], next = pk_off[sc + 1];        // pk_off[n] = size of the whole layout (includes the trailing pad)
    // the codon table in LDS, one copy per wavefront (no block-level barrier: waves leave early): a look-up in the constant array with a
    // per-lane index is a vector load from global memory, one more link in the chain
    __shared__ uint32_t codon_all[4][64];
    uint32_t *codon = codon_all[threadIdx.x >> 6];
    codon[lane] = c_codon[tab][lane];
    if (n_packed == 0) {                      // nothing but the two end pads
        if (s == 0) for (uint32_t x = lane; x < pk_off[0]; x += 64) res[x] = (uint8_t)PEP_PAD_CODE;
        return;
    }
    if (s >= n_packed) return;
    const uint8_t *src = nt + so.off;
    const int L = (int)so.L;
    if (s == 0) for (uint32_t x = lane; x < start; x += 64) res[x] = (uint8_t)PEP_PAD_CODE;          // leading pad
    // Chunks of 512 residues: the wavefront copies the 1536 nucleotide bytes behind them into LDS with aligned dword loads (coalesced:
    // 256 contiguous bytes per instruction), then every lane translates eight residues out of LDS and stores 8 bytes.  Reading the
    // bytes straight from global memory (three byte loads per residue, 24 bytes apart from lane to lane) made every instruction touch
    // 24 cache lines: both sides of K1 ran at the same 230 G residues/s whatever their size.
    // A lane's 24 characters are handled four to a register (nt4): character by character the kernel spent 850 vector instructions per
    // chunk - at four cycles per wave64 instruction that, not memory, was its time.
    __shared__ uint32_t stage_all[4][392];
    uint32_t *stage = stage_all[threadIdx.x >> 6];
    const int frame = (int)d.frame;
    const bool fwd = frame <= 3;
    const int fo = fwd ? frame - 1 : frame - 4;
    for (uint32_t c0 = 0; c0 < next - start; c0 += 512) {
        const int P0 = fo + 3 * (int)(d.aa_off + c0);                    // first nucleotide position (in reading direction) of the chunk
        // window of source bytes [lo, lo + 1536) that holds positions P0 .. P0 + 1535 (forward: as they are; reverse: mirrored - lo may
        // lie in front of the sequence then: those bytes are not fetched and not used)
        const int lo = fwd ? P0 : L - 1 - (P0 + 1535);
        int shift = 0;
        if (c0 < d.len && P0 < L) {
            const uintptr_t A = reinterpret_cast<uintptr_t>(src) + (intptr_t)lo;
            shift = (int)(A & 3u);
            const uint32_t *al = reinterpret_cast<const uint32_t *>(A - shift);
            const int first_dw = lo < 0 ? (-lo + shift) >> 2 : 0;         // (the dword that holds the sequence's first byte starts inside the buffer)
            const int n_dw = (min(1536, L - lo) + shift + 3) >> 2;        // <= 385; the buffer is padded by 64 bytes behind the last sequence
            for (int x = first_dw + lane; x < n_dw; x += 64) stage[x] = al[x];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t x0 = c0 + 8 * lane;
        if (x0 < next - start) {
            // the lane's 24 characters in source order: seven dwords from LDS, shifted down by the window's misalignment
            const int di = fwd ? 6 * lane : 378 - 6 * lane;
            uint32_t raw[7], ch[6];
#pragma unroll
            for (int k = 0; k < 7; ++k) raw[k] = stage[di + k];
#pragma unroll
            for (int k = 0; k < 6; ++k) ch[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)shift);
            if (!fwd) {                                                   // reading direction = descending source order
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t lo_w = __builtin_bswap32(ch[5 - k]), hi_w = __builtin_bswap32(ch[k]);
                    ch[k] = lo_w; ch[5 - k] = hi_w;
                }
            }
            const int n_valid = min(max(L - (P0 + 24 * lane), 0), 24);          // characters of the lane that lie inside the sequence
            const int n_res = (int)min(max((int64_t)d.len - (int64_t)x0, (int64_t)0), (int64_t)8);
            uint32_t word[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                 // twelve characters = four codons
                uint32_t pk[3], bad[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) nt4(ch[3 * h + k], !fwd, pk[k], bad[k]);
                const uint32_t codes = (pk[0] << 16) | (pk[1] << 8) | pk[2];
                uint32_t bad12 = (bad[0] << 8) | (bad[1] << 4) | bad[2];
                const int nv = min(max(n_valid - 12 * h, 0), 12);
                bad12 |= (1u << (12 - nv)) - 1u;                          // characters behind the sequence's end
                uint32_t w = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t c = codon[(codes >> (18 - 6 * k)) & 63u];
                    if ((bad12 >> (9 - 3 * k)) & 7u) c = 23u;
                    if (4 * h + k >= n_res) c = PEP_PAD_CODE;
                    w |= c << (8 * k);
                }
                word[h] = w;
            }
            *reinterpret_cast<uint2 *>(res + start + x0) = make_uint2(word[0], word[1]);
        }
        __builtin_amdgcn_wave_barrier();                                 // the next chunk overwrites the staging area
    }
    // (sequence, start) per 32-byte block that holds residues of this sequence: [start, next - 16) - the gap in front of `next` is at least 16 bytes of
    // padding, so a block never holds residues of two sequences; the block that `next` starts in the middle of belongs to the next sequence
    for (uint32_t b = (start >> 5) + lane, last = ((next & 31u) == 16u) ? ((next - 16u) >> 5) - 1u : (next - 17u) >> 5; b <= last && last != ~0u; b += 64) blk2seq[b] = make_uint2(s, start);
}

// ---- nucleotide sets as residue sets (the blastn-equivalent tool, uberBlast.py:294, 482-509): base codes A0 C1 G2 T3, anything else 4;
// a reverse-strand target is the reverse complement.  One wavefront per packed sequence, eight consecutive bytes per lane (one unaligned
// 8-byte load where the whole window lies inside the sequence, byte loads at its two ends), padding and block -> sequence map as k1_pack.
// (NuclDesc { seq, rev }: common.h)
// one wavefront per (sequence, slice of NUCL_SLICE packed bytes): blockIdx.y is the slice.  (One wavefront per SEQUENCE until round 4: fine for genes,
// but a genome's contig of 2 Mb was packed by a single wavefront - 4 - 8 ms per mapping batch, the largest kernel of the mapping path's trace.)
constexpr uint32_t NUCL_SLICE = 32768;
__global__ __launch_bounds__(256) void nucl_pack(const uint8_t *__restrict__ nt, const uint64_t *__restrict__ nt_off, const NuclDesc *__restrict__ desc,
                                                 const uint32_t *__restrict__ pk_off, uint32_t n_packed, uint8_t *__restrict__ res, uint2 *__restrict__ blk2seq)
{
    const int lane = threadIdx.x & 63;
    const uint32_t s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n_packed == 0) {
        if (s == 0) for (uint32_t x = lane; x < pk_off[0]; x += 64) res[x] = (uint8_t)PEP_PAD_CODE;
        return;
    }
    if (s >= n_packed) return;
    const NuclDesc d = desc[s];
    const uint32_t start = pk_off[s], next = pk_off[s + 1];
    const uint8_t *src = nt + nt_off[d.seq];
    const int64_t L = (int64_t)(nt_off[d.seq + 1] - nt_off[d.seq]);
    const uint32_t x_lo = blockIdx.y * NUCL_SLICE, x_hi = min(next - start, x_lo + NUCL_SLICE);       // (slices are multiples of 512: whole trips)
    if (x_lo >= next - start) return;
    if (s == 0 && blockIdx.y == 0) for (uint32_t x = lane; x < start; x += 64) res[x] = (uint8_t)PEP_PAD_CODE;          // leading pad
    for (uint32_t x0 = x_lo + 8 * lane; x0 < x_hi; x0 += 512) {
        uint32_t word[2] = {0, 0};
        uint8_t raw[8];
        const bool whole = (int64_t)x0 + 8 <= L;
        if (whole) __builtin_memcpy(raw, d.rev ? src + (L - 8 - x0) : src + x0, 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t c = PEP_PAD_CODE;
            if ((int64_t)x0 + k < L) {
                const uint8_t ch = whole ? raw[d.rev ? 7 - k : k] : (d.rev ? src[L - 1 - x0 - k] : src[x0 + k]);
                const int v = base2(ch);
                c = v < 0 ? 4u : (uint32_t)(d.rev ? 3 - v : v);
            }
            word[k >> 2] |= c << (8 * (k & 3));
        }
        *reinterpret_cast<uint2 *>(res + start + x0) = make_uint2(word[0], word[1]);                 // starts and paddings are multiples of 16: whole words
    }
    // (sequence, start) per 32-byte block that holds residues of this sequence: [start, next - 16) - the gap in front of `next` is at least 16 bytes of
    // padding, so a block never holds residues of two sequences; the block that `next` starts in the middle of belongs to the next sequence
    for (uint32_t b = ((start + x_lo) >> 5) + lane, last = min(((next & 31u) == 16u) ? ((next - 16u) >> 5) - 1u : (next - 17u) >> 5, (start + x_hi - 1u) >> 5); b <= last && last != ~0u; b += 64)
        blk2seq[b] = make_uint2(s, start);       // (this slice's part of the map)
}

void fill_codon_table(uint8_t tab[2][64])
{
    static const char *aa = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVVXYXYSSSSXCWCLFLF";
    for (int i = 0; i < 64; ++i) tab[0][i] = tab[1][i] = (uint8_t)(aa[i] - 'A');
    tab[1][56] = (uint8_t)('W' - 'A');          // translation table 4: TGA -> W
}

int upload_codon_table(pep_ctx *ctx)
{
    if (ctx->codon_ready) return PEP_OK;            // (a constant of the device's code object: once per context is enough)
    uint8_t tab[2][64];
    fill_codon_table(tab);
    PEP_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_codon), tab, sizeof(tab), 0, hipMemcpyHostToDevice));
    ctx->codon_ready = true;
    return PEP_OK;
}

// query side: one packed sequence per gene, the chosen frame from its start
__global__ void k1_query_desc(uint32_t n, const uint32_t *__restrict__ frame, const uint32_t *__restrict__ len, PackDesc *__restrict__ desc,
                              uint32_t *__restrict__ padded, uint32_t *__restrict__ len_out, uint32_t *__restrict__ n_out, K1Summary *__restrict__ sum,
                              uint32_t *__restrict__ pin_len)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { *n_out = n; sum->residues = 0ull; sum->max_len = 0u; }       // (k1_offsets accumulates into it)
    if (i >= n) return;
    desc[i] = PackDesc{i, frame[i], 0u, len[i]};
    padded[i] = padded_len(len[i]);
    len_out[i] = len[i];
    pin_len[i] = len[i];                     // the host's copy (pinned memory): the search derives its score thresholds from the lengths
}

// reference side: the chunks of (sequence, frame) w become packed sequences first[w] .. first[w] + cnt[w] - 1
// look-back state of a K1 kernel that scans while it works (lookback.h)
struct K1Scan { uint64_t *state; uint32_t ticket_base; uint64_t epoch; };

// the chunks of (sequence, frame) w become packed sequences first[w] .. first[w] + cnt[w] - 1, first = exclusive scan of the chunk counts -
// taken inside this kernel (tile totals by look-back) instead of by a scan launch in front of it; *n_targets = their number
__global__ __launch_bounds__(256) void k1_ref_desc(uint64_t nw, int n_frames, const uint64_t *__restrict__ chunk_base, const uint32_t *__restrict__ chunk_cnt,
                                                   const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ chunk_len,
                                                   PackDesc *__restrict__ desc, uint32_t *__restrict__ padded, uint32_t *__restrict__ len_out, K1Summary *__restrict__ sum,
                                                   uint32_t *__restrict__ n_targets, K1Scan sc, const uint64_t *__restrict__ nt_off, K1Src *__restrict__ src_of,
                                                   uint32_t *__restrict__ first_t, uint32_t *__restrict__ first_w, uint32_t fill_from)
{
    __shared__ uint32_t s_tile, lds[4];
    __shared__ uint64_t s_pre;
    const uint32_t tile = lb_take_tile(sc.state, sc.ticket_base, &s_tile);
    const uint64_t w = (uint64_t)tile * 256 + threadIdx.x;
    if (w == 0) { sum->residues = 0ull; sum->max_len = 0u; }
    const uint32_t cnt = w < nw ? chunk_cnt[w] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256<uint32_t>(cnt, &tot, lds);
    if (threadIdx.x < 64) {
        const uint64_t p = lb_tile_prefix<32>(sc.state + 1, tile, tot, sc.epoch, (int)threadIdx.x);
        if (threadIdx.x == 0) s_pre = p;
    }
    __syncthreads();
    if (tile == gridDim.x - 1 && threadIdx.x == 255) *n_targets = (uint32_t)(s_pre + tot);
    if (w >= nw) return;
    const uint32_t g = (uint32_t)(w / n_frames), f = (uint32_t)(w % n_frames) + 1;
    const uint64_t base = chunk_base[w];
    const uint32_t at = (uint32_t)s_pre + ex;
    if (f == 1u) first_t[g] = cnt ? at : PEP_SELF_NONE;          // the packed sequence frame 1 of reference sequence g starts with (seeds.hip: self_prepare)
    const uint64_t o = nt_off[g];
    const K1Src so{o, (uint32_t)(nt_off[g + 1] - o), 0u};
    first_w[w] = at;
    if (cnt >= fill_from) return;            // a contig's frame - thousands of chunks: its records are written by k1_ref_desc_fill, a thread per chunk (one thread walking them here was 1.1 ms of a mapping batch's K1)
    for (uint32_t c = 0; c < cnt; ++c) {
        desc[at + c] = PackDesc{g, f, chunk_off[base + c], chunk_len[base + c]};
        src_of[at + c] = so;
        padded[at + c] = padded_len(chunk_len[base + c]);
        len_out[at + c] = chunk_len[base + c];
    }
}

// the descriptor, source record and lengths of every chunk of the LONG frames (k1_ref_desc left them out): tile x of 256 chunk slots of frame tiles[x].w
__global__ __launch_bounds__(256) void k1_ref_desc_fill(const K1Seg *__restrict__ tiles, int n_frames, const uint64_t *__restrict__ chunk_base, const uint32_t *__restrict__ chunk_cnt,
                                                        const uint32_t *__restrict__ chunk_off, const uint32_t *__restrict__ chunk_len, const uint32_t *__restrict__ first_w,
                                                        uint32_t fill_from, PackDesc *__restrict__ desc, uint32_t *__restrict__ padded, uint32_t *__restrict__ len_out,
                                                        const uint64_t *__restrict__ nt_off, K1Src *__restrict__ src_of)
{
    const K1Seg tl = tiles[blockIdx.x];
    const uint32_t w = tl.w, c = tl.seg * 256u + threadIdx.x, cnt = chunk_cnt[w];
    if (cnt < fill_from || c >= cnt) return;
    const uint32_t g = w / (uint32_t)n_frames, f = w % (uint32_t)n_frames + 1u;
    const uint64_t base = chunk_base[w], o = nt_off[g];
    const uint32_t at = first_w[w] + c, len = chunk_len[base + c];
    desc[at] = PackDesc{g, f, chunk_off[base + c], len};
    src_of[at] = K1Src{o, (uint32_t)(nt_off[g + 1] - o), 0u};
    padded[at] = padded_len(len);
    len_out[at] = len;
}

// pk_off[i] = start of packed sequence i = END_PAD + exclusive scan of the padded lengths (taken inside this kernel by look-back; slots behind
// the last sequence count as empty); entries n .. cap hold the layout's total size (sentinel of the owner search).
// Also the set's summary: number of sequences, layout size, residues, longest sequence.
__global__ __launch_bounds__(256) void k1_offsets(const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ padded, uint32_t cap, uint32_t *__restrict__ pk_off,
                                                  const uint32_t *__restrict__ len, K1Summary *__restrict__ sum, K1Scan sc)
{
    __shared__ uint32_t s_tile, lds[4];
    __shared__ uint64_t s_pre;
    __shared__ unsigned long long blk_s[4];
    __shared__ uint32_t blk_m[4];
    const uint32_t tile = lb_take_tile(sc.state, sc.ticket_base, &s_tile);
    const uint32_t i = tile * 256 + threadIdx.x;
    const uint32_t n = *n_ptr;
    const uint32_t pad = i < n ? padded[i] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256<uint32_t>(pad, &tot, lds);
    if (threadIdx.x < 64) {
        const uint64_t p = lb_tile_prefix<32>(sc.state + 1, tile, tot, sc.epoch, (int)threadIdx.x);
        if (threadIdx.x == 0) s_pre = p;
    }
    __syncthreads();
    const uint32_t at = (uint32_t)s_pre + ex;                // padded bytes in front of slot i (behind the last sequence: all of them)
    if (i <= cap) pk_off[i] = i < n ? at + PEP_END_PAD : at + 2 * PEP_END_PAD;
    // residues and longest sequence: one pair of atomics per block (a thousand wavefronts adding to the same two words took 18 us)
    uint32_t L = i < n ? len[i] : 0u, m = L;
    unsigned long long s = L;
    for (int d = 32; d > 0; d >>= 1) { s += __shfl_xor(s, d, 64); m = max(m, (uint32_t)__shfl_xor((int)m, d, 64)); }
    if ((threadIdx.x & 63) == 0) { blk_s[threadIdx.x >> 6] = s; blk_m[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t bm = max(max(blk_m[0], blk_m[1]), max(blk_m[2], blk_m[3]));
        if (bm) { atomicAdd(&sum->residues, blk_s[0] + blk_s[1] + blk_s[2] + blk_s[3]); atomicMax(&sum->max_len, bm); }
    }
    if (i == cap) { sum->n = n; sum->total = (unsigned long long)at + 2ull * PEP_END_PAD; }
}

int reserve_packed(pep_ctx *ctx, SeqSet &out, uint32_t cap, uint64_t upper)
{
    if (upper > PEP_MAX_RESIDUES + (uint64_t)cap * 32) return pep_fail(ctx, PEP_ERR_LIMIT, "packed protein set exceeds 2^29 bytes");
    PEP_TRY(dev_reserve(ctx, out.res, upper + 64));
    PEP_TRY(dev_reserve(ctx, out.off, ((size_t)cap + 2) * 4));
    PEP_TRY(dev_reserve(ctx, out.len, ((size_t)cap + 2) * 4));
    PEP_TRY(dev_reserve(ctx, out.blk2seq, (upper / 32 + 2) * sizeof(uint2)));
    return PEP_OK;
}

// Device-side layout + packing (after reserve_packed).  d_desc / d_padded hold up to `cap` entries (padded = 0 beyond the *d_n real ones); `upper` bounds the
// layout's size.  Nothing is read back here: the caller downloads the descriptors once, after everything is queued.
int layout_and_pack(pep_ctx *ctx, const NtSet &nt, int tab, const PackDesc *d_desc, const uint32_t *d_padded, uint32_t cap, const uint32_t *d_n,
                    uint64_t upper, SeqSet &out, DevBuf &d_scan, DevBuf &tmp, K1Summary *pin_sum)
{
    (void)d_scan; (void)tmp;
    K1Summary *d_sum = reinterpret_cast<K1Summary *>(const_cast<PackDesc *>(d_desc) + cap);        // behind the descriptors
    K1Scan sc;
    const uint64_t tiles = ceil_div((uint64_t)cap + 1, 256);
    PEP_TRY(pep_lookback_begin(ctx, ctx->scan_state[0], tiles, (1u << 30) - 1, &sc.state, &sc.ticket_base, &sc.epoch));
    hipLaunchKernelGGL(k1_offsets, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, d_n, d_padded, cap, out.off.as<uint32_t>(), out.len.as<const uint32_t>(), d_sum, sc);
    hipLaunchKernelGGL(k1_pack, dim3((unsigned)ceil_div((uint64_t)cap + 1, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), (const K1Src *)k1_src_of(d_desc, cap), tab,
                       d_desc, out.off.as<const uint32_t>(), d_n, cap, out.res.as<uint8_t>(), out.blk2seq.as<uint2>(), (const K1Summary *)d_sum, pin_sum);
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

// the eager part of a finished K1 side: the set's summary (computed by k1_offsets, downloaded behind the descriptors)
int take_summary(pep_ctx *ctx, const void *pinned, SeqSet &out)
{
    const K1Summary *sum = reinterpret_cast<const K1Summary *>(pinned);
    if (sum->total > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "packed protein set exceeds 2^29 bytes");
    if (sum->max_len > PEP_MAX_SEQ_LEN) return pep_fail(ctx, PEP_ERR_LIMIT, "protein longer than PEP_MAX_SEQ_LEN");
    out.n = sum->n; out.total = sum->total; out.residues = sum->residues; out.max_len = sum->max_len;
    return PEP_OK;
}

// host mirrors of a packed set from its downloaded descriptors (same arithmetic as k1_offsets)
void finish_layout(const PackDesc *desc, uint32_t n, SeqSet &out)
{
    out.h_off.resize((size_t)n + 1);
    out.h_len.resize(n);
    uint64_t pos = PEP_END_PAD;
    for (uint32_t i = 0; i < n; ++i) {
        out.h_off[i] = (uint32_t)pos;
        out.h_len[i] = desc[i].len;
        pos += ((uint64_t)desc[i].len + 15) / 16 * 16 + PEP_SEQ_GAP;
    }
    out.h_off[n] = (uint32_t)(pos + PEP_END_PAD);
}

int k1_ref_finish(pep_ctx *ctx)
{
    PEP_HIP(ctx, pep_event_wait(ctx->k1_event));
    PEP_TRY(take_summary(ctx, ctx->pin_k1.p, ctx->t));
    ctx->t_tables_lazy = true;               // t_meta, h_off, h_len: pep_k1_host_tables, when somebody needs them
    return PEP_OK;
}

}  // namespace

// phase 1 = queue the device work and the download of the descriptors, 2 = wait for it and build the host-side tables, 0 = both.
// pep_translate queues the reference side, then the query side, and builds the reference tables while the query kernels run.
int pep_k1_query(pep_ctx *ctx, int gtable, int phase)
{
    const NtSet &nt = ctx->q_nt;
    const int tab = gtable == 4 ? 1 : 0;
    const uint32_t n = nt.n;
    if (phase != 2) {
        PEP_TRY(upload_codon_table(ctx));
        if (n > PEP_MAX_QUERIES) return pep_fail(ctx, PEP_ERR_LIMIT, "too many queries");
        DevBuf *W = ctx->ws;
        PEP_TRY(dev_reserve(ctx, W[0], ((size_t)n + 1) * 4));
        PEP_TRY(dev_reserve(ctx, W[1], ((size_t)n + 1) * 4));
        DevBuf &D = ctx->d_k1_desc_q;              // the descriptors stay on the device (a buffer of their own): fetched when somebody asks for the host tables
        PEP_TRY(dev_reserve(ctx, D, (2 * (size_t)n + 4) * sizeof(PackDesc)));      // + the summary and the source records behind the descriptors
        PEP_TRY(pin_reserve(ctx, ctx->pin_k1q, sizeof(K1Summary) + ((size_t)n + 1) * 4));
        K1Summary *pin_sum = reinterpret_cast<K1Summary *>(ctx->pin_k1q.p);
        uint32_t *pin_len = reinterpret_cast<uint32_t *>(ctx->pin_k1q.p + sizeof(K1Summary));
        PEP_TRY(dev_reserve(ctx, W[3], ((size_t)n + 1) * 4));
        PEP_TRY(dev_reserve(ctx, W[5], 16));
        const uint64_t upper = 2 * PEP_END_PAD + (nt.total + 2 * (uint64_t)n) / 3 + (uint64_t)n * (16 + PEP_SEQ_GAP);
        PEP_TRY(reserve_packed(ctx, ctx->q, n, upper));
        if (n) hipLaunchKernelGGL(k1_query_frames, dim3((unsigned)ceil_div(n, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), n, tab,
                                  D.as<PackDesc>(), W[3].as<uint32_t>(), ctx->q.len.as<uint32_t>(), W[5].as<uint32_t>(), reinterpret_cast<K1Summary *>(D.as<PackDesc>() + n), pin_len,
                                  k1_src_of(D.as<const PackDesc>(), n));
        else hipLaunchKernelGGL(k1_query_desc, dim3(1), dim3(64), 0, ctx->stream, 0u, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                                D.as<PackDesc>(), W[3].as<uint32_t>(), ctx->q.len.as<uint32_t>(), W[5].as<uint32_t>(), reinterpret_cast<K1Summary *>(D.as<PackDesc>() + n), pin_len);
        PEP_TRY(layout_and_pack(ctx, nt, tab, D.as<const PackDesc>(), W[3].as<const uint32_t>(), n, W[5].as<const uint32_t>(), upper, ctx->q, W[4], W[6], pin_sum));
        // an event of its own: whoever waits for the query side must not wait for what was queued behind it (pep_search queues the reference side next)
        ctx->k1q_event_set = false;
        if (ctx->k1q_event || hipEventCreateWithFlags(&ctx->k1q_event, pep_wait_event_flags()) == hipSuccess)
            ctx->k1q_event_set = hipEventRecord(ctx->k1q_event, ctx->stream) == hipSuccess;
    }
    if (phase == 1) return PEP_OK;
    if (ctx->k1q_event_set) PEP_HIP(ctx, pep_event_wait(ctx->k1q_event));
    else PEP_HIP(ctx, pep_stream_wait(ctx));
    ctx->k1q_event_set = false;
    PEP_TRY(take_summary(ctx, ctx->pin_k1q.p, ctx->q));
    // the lengths are here already (k1_query_desc wrote them into the pinned buffer); q_meta and h_off wait for pep_k1_host_tables_q
    const uint32_t *pin_len = reinterpret_cast<const uint32_t *>(ctx->pin_k1q.p + sizeof(K1Summary));
    ctx->q.h_len.assign(pin_len, pin_len + n);
    ctx->q_tables_lazy = true;
    return PEP_OK;
}

// the per-sequence host tables of both sides (meta records, offsets, lengths) from the descriptors their last K1 downloaded
int pep_k1_host_tables(pep_ctx *ctx)
{
    if (ctx->t_tables_lazy) {
        const uint32_t n = ctx->t.n;
        std::vector<PackDesc> fetched((size_t)n + 1);
        if (n) { PEP_TRY(pep_d2h_queue(ctx, fetched.data(), ctx->d_k1_desc_t.p, (size_t)n * sizeof(PackDesc))); PEP_HIP(ctx, pep_stream_wait(ctx)); pep_d2h_finish(ctx); }
        const PackDesc *desc = fetched.data();
        ctx->t_meta.resize(n);
        for (uint32_t i = 0; i < n; ++i) ctx->t_meta[i] = pep_target_meta{desc[i].seq, desc[i].frame, desc[i].aa_off, desc[i].len};
        finish_layout(desc, n, ctx->t);
        ctx->t_tables_lazy = false;
    }
    return pep_k1_host_tables_q(ctx);
}

int pep_k1_host_tables_q(pep_ctx *ctx)
{
    if (ctx->q_tables_lazy) {
        const NtSet &nt = ctx->q_nt;
        const uint32_t n = ctx->q.n;
        std::vector<PackDesc> fetched((size_t)n + 1);
        if (n) { PEP_TRY(pep_d2h_queue(ctx, fetched.data(), ctx->d_k1_desc_q.p, (size_t)n * sizeof(PackDesc))); PEP_HIP(ctx, pep_stream_wait(ctx)); pep_d2h_finish(ctx); }
        const PackDesc *desc = fetched.data();
        ctx->q_meta.resize(n);
        for (uint32_t i = 0; i < n; ++i) ctx->q_meta[i] = pep_query_meta{i, desc[i].frame, desc[i].len, (uint32_t)(nt.h_off[i + 1] - nt.h_off[i])};
        finish_layout(desc, n, ctx->q);
        ctx->q_tables_lazy = false;
    }
    return PEP_OK;
}

int pep_k1_ref(pep_ctx *ctx, int frames, int gtable, int phase)
{
    const NtSet &nt = ctx->r_nt;
    const int tab = gtable == 4 ? 1 : 0;
    if (phase == 2) return k1_ref_finish(ctx);
    PEP_TRY(upload_codon_table(ctx));
    const uint32_t n = nt.n;
    const int nf = frames == 3 ? 3 : 6;
    const uint64_t nw = (uint64_t)n * nf;
    if (ctx->k1_base_frames != nf) {
        // upper bound of chunks per frame: one per started 1001 residues of (frame string + 'X'); and of the packed layout's size.
        // Depends on the input lengths only: computed and uploaded once per reference set.
        std::vector<uint64_t> &base = ctx->k1_base;
        base.assign(nw + 1, 0);
        uint64_t upper = 2 * PEP_END_PAD;
        for (uint32_t g = 0; g < n; ++g) {
            const uint64_t L = nt.h_off[g + 1] - nt.h_off[g];
            for (int f = 1; f <= nf; ++f) {
                const uint64_t shift = (uint64_t)(f <= 3 ? f - 1 : f - 4);
                const uint64_t na = L > shift ? (L - shift + 2) / 3 : 0;
                const uint64_t slots_w = (na + 1) / 1001 + 1;
                base[(uint64_t)g * nf + f] = base[(uint64_t)g * nf + f - 1] + slots_w;
                upper += na + slots_w * (15 + PEP_SEQ_GAP);
            }
        }
        ctx->k1_upper = upper;
        PEP_TRY(dev_reserve(ctx, ctx->d_k1_base, (nw + 1) * 8));
        PEP_HIP(ctx, hipMemcpy(ctx->d_k1_base.p, base.data(), (nw + 1) * 8, hipMemcpyHostToDevice));
        // the long frames' tables (k1_stop_mask / k1_ref_chunks_mask / k1_ref_desc_fill): a function of the lengths as well (PEPPAN_K1_PLAIN_CHUNKS=1: every frame by
        // the one-wavefront walk over the nucleotides, for comparison)
        static const bool plain_chunks = [] { const char *e = getenv("PEPPAN_K1_PLAIN_CHUNKS"); return e && atoi(e) != 0; }();
        std::vector<K1Seg> segs, tiles;                 // tiles of 4 096 characters of the stop mask / of 256 chunk slots
        std::vector<K1Long> longs;
        std::vector<uint32_t> long_of_w;
        uint64_t mask_words = 0;
        for (uint32_t g = 0; g < n; ++g) {
            const uint64_t L = nt.h_off[g + 1] - nt.h_off[g];
            for (int f = 1; f <= nf; ++f) {
                const uint64_t shift = (uint64_t)(f <= 3 ? f - 1 : f - 4);
                const uint64_t na = L > shift ? (L - shift + 2) / 3 : 0;
                if (na + 1 <= (uint64_t)K1_LONG || plain_chunks) continue;
                const uint32_t w = g * (uint32_t)nf + (uint32_t)(f - 1);
                if (long_of_w.empty()) long_of_w.assign(nw, 0u);
                long_of_w[w] = (uint32_t)longs.size();
                longs.push_back(K1Long{w, 0u, mask_words});
                const uint64_t n_words = (na + 1 + 63) / 64;
                mask_words += n_words;
                for (uint32_t k = 0, nt_ = (uint32_t)((n_words + 63) / 64); k < nt_; ++k) segs.push_back(K1Seg{w, k});
                for (uint32_t k = 0, n_tiles = (uint32_t)(((na + 1) / 1001 + 1 + 255) / 256); k < n_tiles; ++k) tiles.push_back(K1Seg{w, k});
            }
        }
        ctx->k1_n_seg = (uint32_t)segs.size();
        ctx->k1_n_long = (uint32_t)longs.size();
        if (!segs.empty()) {
            PEP_TRY(dev_reserve(ctx, ctx->d_k1_seg, segs.size() * sizeof(K1Seg)));
            PEP_TRY(dev_reserve(ctx, ctx->d_k1_long, longs.size() * sizeof(K1Long) + (size_t)nw * 4));       // (the frame -> long frame table behind the records)
            PEP_TRY(dev_reserve(ctx, ctx->d_k1_spec, (mask_words + 1) * 8));
            PEP_HIP(ctx, hipMemcpy(ctx->d_k1_seg.p, segs.data(), segs.size() * sizeof(K1Seg), hipMemcpyHostToDevice));
            PEP_HIP(ctx, hipMemcpy(ctx->d_k1_long.p, longs.data(), longs.size() * sizeof(K1Long), hipMemcpyHostToDevice));
            PEP_HIP(ctx, hipMemcpy(ctx->d_k1_long.as<char>() + longs.size() * sizeof(K1Long), long_of_w.data(), (size_t)nw * 4, hipMemcpyHostToDevice));
            PEP_TRY(dev_reserve(ctx, ctx->d_k1_tiles, tiles.size() * sizeof(K1Seg)));
            PEP_HIP(ctx, hipMemcpy(ctx->d_k1_tiles.p, tiles.data(), tiles.size() * sizeof(K1Seg), hipMemcpyHostToDevice));
        }
        ctx->k1_n_tiles = (uint32_t)tiles.size();
        ctx->k1_base_frames = nf;
    }
    const uint64_t slots = ctx->k1_base[nw], upper = ctx->k1_upper;
    if (slots > PEP_MAX_TARGETS) return pep_fail(ctx, PEP_ERR_LIMIT, "too many targets");
    DevBuf *W = ctx->ws;
    PEP_TRY(dev_reserve(ctx, W[1], (nw + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[2], (slots + 1) * 4));
    PEP_TRY(dev_reserve(ctx, W[3], (slots + 1) * 4));
    DevBuf &D = ctx->d_k1_desc_t;                  // the descriptors stay on the device (a buffer of their own: ws[] is the seed stage's next)
    PEP_TRY(dev_reserve(ctx, D, (2 * slots + 4) * sizeof(PackDesc)));             // + the summary and the source records behind the descriptors
    PEP_TRY(pin_reserve(ctx, ctx->pin_k1, sizeof(K1Summary)));
    K1Summary *pin_sum = reinterpret_cast<K1Summary *>(ctx->pin_k1.p);
    PEP_TRY(dev_reserve(ctx, W[5], (nw + 2) * 4));
    PEP_TRY(dev_reserve(ctx, W[6], (slots + 1) * 4));
    PEP_TRY(reserve_packed(ctx, ctx->t, (uint32_t)slots, upper));
    PEP_TRY(dev_reserve(ctx, ctx->d_self_t, ((size_t)n + 1) * 4));
    ctx->self_first_n = n;
    const uint64_t *d_base = ctx->d_k1_base.as<const uint64_t>();
    if (nw) {
        hipLaunchKernelGGL(k1_ref_chunks, dim3((unsigned)ceil_div(nw, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), n, nf, tab,
                           d_base, W[1].as<uint32_t>(), W[2].as<uint32_t>(), W[3].as<uint32_t>(), ctx->k1_n_long ? K1_LONG : INT64_MAX);
        if (ctx->k1_n_long) {
            // the frames of contigs: their stops as a bit mask (all frames side by side), then the chunk chain of every frame out of register windows of that mask
            const K1Long *d_longs = ctx->d_k1_long.as<const K1Long>();
            const uint32_t *d_long_of_w = reinterpret_cast<const uint32_t *>(ctx->d_k1_long.as<const char>() + (size_t)ctx->k1_n_long * sizeof(K1Long));
            hipLaunchKernelGGL(k1_stop_mask, dim3((unsigned)ceil_div(ctx->k1_n_seg, 4)), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(), nf, tab,
                               ctx->d_k1_seg.as<const K1Seg>(), ctx->k1_n_seg, d_longs, d_long_of_w, ctx->d_k1_spec.as<unsigned long long>());
            hipLaunchKernelGGL(k1_ref_chunks_mask, dim3((unsigned)ceil_div(ctx->k1_n_long, 4)), dim3(256), 0, ctx->stream, nt.off.as<const uint64_t>(), nf, d_longs, ctx->k1_n_long,
                               ctx->d_k1_spec.as<const unsigned long long>(), d_base, W[1].as<uint32_t>(), W[2].as<uint32_t>(), W[3].as<uint32_t>());
        }
        {
            K1Scan sc;
            const uint64_t tiles = ceil_div(nw, 256);
            PEP_TRY(pep_lookback_begin(ctx, ctx->scan_state[0], tiles, (1u << 30) - 1, &sc.state, &sc.ticket_base, &sc.epoch));
            hipLaunchKernelGGL(k1_ref_desc, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, nw, nf, d_base, W[1].as<const uint32_t>(),
                               W[2].as<const uint32_t>(), W[3].as<const uint32_t>(), D.as<PackDesc>(), W[6].as<uint32_t>(),
                               ctx->t.len.as<uint32_t>(), reinterpret_cast<K1Summary *>(D.as<PackDesc>() + slots), W[5].as<uint32_t>() + nw, sc,
                               nt.off.as<const uint64_t>(), k1_src_of(D.as<const PackDesc>(), (uint32_t)slots), ctx->d_self_t.as<uint32_t>(), W[5].as<uint32_t>(),
                               ctx->k1_n_tiles ? K1_FILL_FROM : 0xFFFFFFFFu);         // W[5][nw] = number of targets, W[5][w] = the first packed sequence of frame w
            if (ctx->k1_n_tiles)
                hipLaunchKernelGGL(k1_ref_desc_fill, dim3(ctx->k1_n_tiles), dim3(256), 0, ctx->stream, ctx->d_k1_tiles.as<const K1Seg>(), nf, d_base, W[1].as<const uint32_t>(),
                                   W[2].as<const uint32_t>(), W[3].as<const uint32_t>(), W[5].as<const uint32_t>(), K1_FILL_FROM, D.as<PackDesc>(), W[6].as<uint32_t>(),
                                   ctx->t.len.as<uint32_t>(), nt.off.as<const uint64_t>(), k1_src_of(D.as<const PackDesc>(), (uint32_t)slots));
        }
        PEP_TRY(layout_and_pack(ctx, nt, tab, D.as<const PackDesc>(), W[6].as<const uint32_t>(), (uint32_t)slots, W[5].as<const uint32_t>() + nw, upper, ctx->t,
                                W[7], W[8], pin_sum));
    } else {
        PEP_HIP(ctx, hipMemsetAsync(W[5].p, 0, 8, ctx->stream));
        PEP_HIP(ctx, hipMemsetAsync(W[6].p, 0, 4, ctx->stream));
        PEP_HIP(ctx, hipMemsetAsync(D.p, 0, 3 * sizeof(PackDesc), ctx->stream));               // (the summary's accumulators)
        PEP_TRY(layout_and_pack(ctx, nt, tab, D.as<const PackDesc>(), W[6].as<const uint32_t>(), 0, W[5].as<const uint32_t>(), upper, ctx->t, W[7], W[8], pin_sum));
    }
    // the summary has been written into pinned memory by k1_pack; an event marks the point of the stream where it is there
    ctx->k1_desc_cap = (uint32_t)slots;
    if (!ctx->k1_event && hipEventCreateWithFlags(&ctx->k1_event, pep_wait_event_flags()) != hipSuccess) return pep_fail(ctx, PEP_ERR_HIP, "hipEventCreate failed");
    PEP_HIP(ctx, hipEventRecord(ctx->k1_event, ctx->stream));
    if (phase == 1) return PEP_OK;
    return k1_ref_finish(ctx);
}

// ---- self-search (round 6): PEPPAN's hot call searches a gene set against itself (-r CL -q CL, PEPPAN.py:229-230).  Frame 1 of reference gene g then IS protein
// query g - the same residues at the same offsets - and 57 % of the raw seed hits of the 10 000-gene search are a gene against itself on diagonal 0
// (DESIGN.md section 4.11).  K1 leaves d_self_t[g] = the packed sequence that frame 1 of reference sequence g starts with (k1_ref_desc); whether that target
// really repeats query g is decided on the device from the packed residues (seeds.hip: self_prepare), nothing is assumed about the caller.  The nucleotide tool's sets
// (pep_use_nt_as_residues) have the forward strand of reference sequence g in that place (pep_nucl_sets).  Here: is the map applicable to the current sets at all, which table
// it is, and the distance table cleared.
int pep_self_map(pep_ctx *ctx, int *on)
{
    *on = 0;
    const SeqSet &Q = ctx->q, &T = ctx->t;
    if (!ctx->q_from_nt || !ctx->t_from_nt || Q.n == 0 || T.n == 0) return PEP_OK;
    if (ctx->resid_from_nucl) {
        // the nucleotide tool's sets (pep_use_nt_as_residues): the forward strand of reference sequence g; it must have the query's length exactly - the matcher judges the
        // bases next to a look-up word by the target's side alone where the hit is a gene against itself
        if (!ctx->nucl_valid || !ctx->nucl_t.d_first.p || ctx->nucl_t.n_first == 0) return PEP_OK;
        ctx->self_first = ctx->nucl_t.d_first.as<const uint32_t>(); ctx->self_first_cnt = ctx->nucl_t.n_first; ctx->self_exact_len = 1;
    } else {
        if (!ctx->d_self_t.p || ctx->self_first_n == 0) return PEP_OK;
        ctx->self_first = ctx->d_self_t.as<const uint32_t>(); ctx->self_first_cnt = ctx->self_first_n; ctx->self_exact_len = 0;
    }
    // a self-search has as many queries as reference sequences; anything else - genes against the contigs of genomes, above all - is not worth the distance table's fill,
    // self_prepare's launch and the matcher's extra word per tile (8 % of the nucleotide matcher's launch over a mapping batch)
    if (ctx->self_first_cnt != Q.n) return PEP_OK;
    const size_t blocks = (size_t)(T.total / 32 + 16);                   // (the matcher reads the word of every position of its last 256-position tile)
    PEP_TRY(dev_reserve(ctx, ctx->d_self_delta, blocks * 4));
    PEP_HIP(ctx, hipMemsetAsync(ctx->d_self_delta.p, 0x80, blocks * 4, ctx->stream));            // PEP_SELF_NO_DELTA
    *on = 1;
    return PEP_OK;
}

// The packed residue set of one side from its nucleotide set: `order` lists (sequence, strand) per packed sequence.  Everything but the residues
// themselves - offsets, lengths, descriptors, the host mirrors - depends on the uploaded nucleotide sets (and the target groups) only and is kept
// (NuclSide) between calls: PEPPAN's hot call runs the nucleotide tool and the translated tool in turn on the same sets, and K1 overwrites the packed
// sets in between.  A repeat costs four small device-to-device copies and the two pack kernels, queued without a wait.
static int nucl_build(pep_ctx *ctx, const NtSet &nt, const std::vector<NuclDesc> &order, uint32_t max_n, SeqSet &out, pep_ctx::NuclSide &keep)
{
    const uint32_t n = (uint32_t)order.size();
    if (n > max_n) return pep_fail(ctx, PEP_ERR_LIMIT, "too many sequences");
    keep.h_off.assign((size_t)n + 1, 0);
    keep.h_len.assign(n, 0);
    uint64_t pos = PEP_END_PAD, residues = 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t len = nt.h_off[order[i].seq + 1] - nt.h_off[order[i].seq];
        if (len > PEP_MAX_SEQ_LEN) return pep_fail(ctx, PEP_ERR_LIMIT, "sequence longer than PEP_MAX_SEQ_LEN");
        keep.h_off[i] = (uint32_t)pos; keep.h_len[i] = (uint32_t)len;
        residues += len; max_len = std::max(max_len, (uint32_t)len);
        pos += (len + 15) / 16 * 16 + PEP_SEQ_GAP;
        if (pos > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "packed sequence set exceeds 2^29 bytes");
    }
    pos += PEP_END_PAD;
    keep.h_off[n] = (uint32_t)pos;
    keep.n = n; keep.total = pos; keep.residues = residues; keep.max_len = max_len;
    PEP_TRY(dev_reserve(ctx, keep.d_off, ((size_t)n + 2) * 4));
    PEP_TRY(dev_reserve(ctx, keep.d_len, ((size_t)n + 2) * 4));
    PEP_TRY(dev_reserve(ctx, keep.d_desc, ((size_t)n + 1) * sizeof(NuclDesc)));
    PEP_HIP(ctx, hipMemcpyAsync(keep.d_off.p, keep.h_off.data(), ((size_t)n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n) PEP_HIP(ctx, hipMemcpyAsync(keep.d_len.p, keep.h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (n) PEP_HIP(ctx, hipMemcpyAsync(keep.d_desc.p, order.data(), (size_t)n * sizeof(NuclDesc), hipMemcpyHostToDevice, ctx->stream));
    PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));           // (the descriptor vector is the caller's; once per pair of uploads)
    (void)out;
    return PEP_OK;
}

// a kept side -> the context's packed set: offsets and lengths copied on the device, residues and the block map packed from the nucleotides
static int nucl_apply(pep_ctx *ctx, const NtSet &nt, const pep_ctx::NuclSide &keep, SeqSet &out)
{
    const uint32_t n = keep.n;
    out.n = n; out.total = keep.total; out.residues = keep.residues; out.max_len = keep.max_len;
    out.h_off = keep.h_off; out.h_len = keep.h_len;
    PEP_TRY(dev_reserve(ctx, out.res, keep.total + 64));
    PEP_TRY(dev_reserve(ctx, out.off, ((size_t)n + 2) * 4));
    PEP_TRY(dev_reserve(ctx, out.len, ((size_t)n + 2) * 4));
    PEP_TRY(dev_reserve(ctx, out.blk2seq, (keep.total / 32 + 2) * sizeof(uint2)));
    PEP_HIP(ctx, hipMemcpyAsync(out.off.p, keep.d_off.p, ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (n) PEP_HIP(ctx, hipMemcpyAsync(out.len.p, keep.d_len.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(nucl_pack, dim3((unsigned)ceil_div((uint64_t)n + 1, 4), (unsigned)ceil_div((uint64_t)keep.max_len + 15 + PEP_SEQ_GAP, NUCL_SLICE) + 1), dim3(256), 0, ctx->stream, nt.nt.as<const uint8_t>(), nt.off.as<const uint64_t>(),
                       keep.d_desc.as<const NuclDesc>(), out.off.as<const uint32_t>(), n, out.res.as<uint8_t>(), out.blk2seq.as<uint2>());
    PEP_HIP(ctx, hipGetLastError());
    return PEP_OK;
}

int pep_nucl_sets(pep_ctx *ctx, int strands)
{
    if (!ctx->nucl_valid || ctx->nucl_strands != strands) {
        ctx->nucl_valid = false;
        std::vector<NuclDesc> qo(ctx->q_nt.n), to;
        for (uint32_t i = 0; i < ctx->q_nt.n; ++i) qo[i] = NuclDesc{i, 0u};
        PEP_TRY(nucl_build(ctx, ctx->q_nt, qo, PEP_MAX_QUERIES, ctx->q, ctx->nucl_q));
        // targets: per reference set (pep_set_target_groups; one set otherwise) all forward strands, then all reverse complements
        const uint32_t nr = ctx->r_nt.n;
        const bool grouped = ctx->group_of_seq.size() == nr && nr > 0;
        to.reserve((size_t)nr * strands);
        for (uint32_t a = 0; a < nr;) {
            uint32_t b = a + 1;
            while (b < nr && (!grouped || ctx->group_of_seq[b] == ctx->group_of_seq[a])) ++b;
            for (int rev = 0; rev < strands; ++rev)
                for (uint32_t i = a; i < b; ++i) to.push_back(NuclDesc{i, (uint32_t)rev});
            a = b;
        }
        PEP_TRY(nucl_build(ctx, ctx->r_nt, to, PEP_MAX_TARGETS, ctx->t, ctx->nucl_t));
        {
            // per reference sequence its forward strand among the targets: where a self-search finds the target that repeats query g (seeds.hip: self_prepare)
            std::vector<uint32_t> first(nr, PEP_SELF_NONE);
            for (size_t i = 0; i < to.size(); ++i) if (!to[i].rev) first[to[i].seq] = (uint32_t)i;
            PEP_TRY(dev_reserve(ctx, ctx->nucl_t.d_first, ((size_t)nr + 1) * 4));
            if (nr) PEP_HIP(ctx, hipMemcpy(ctx->nucl_t.d_first.p, first.data(), (size_t)nr * 4, hipMemcpyHostToDevice));
            ctx->nucl_t.n_first = nr;
        }
        ctx->nucl_q.q_meta.resize(ctx->q_nt.n);
        for (uint32_t i = 0; i < ctx->q_nt.n; ++i) ctx->nucl_q.q_meta[i] = pep_query_meta{i, 1u, ctx->nucl_q.h_len[i], ctx->nucl_q.h_len[i]};
        ctx->nucl_t.t_meta.resize(to.size());
        for (size_t i = 0; i < to.size(); ++i) ctx->nucl_t.t_meta[i] = pep_target_meta{to[i].seq, to[i].rev ? 4u : 1u, 0u, ctx->nucl_t.h_len[i]};
        ctx->nucl_strands = strands;
        ctx->nucl_valid = true;
    }
    PEP_TRY(nucl_apply(ctx, ctx->q_nt, ctx->nucl_q, ctx->q));
    PEP_TRY(nucl_apply(ctx, ctx->r_nt, ctx->nucl_t, ctx->t));
    ctx->q_tables_lazy = false;
    ctx->t_tables_lazy = false;
    ctx->q_meta = ctx->nucl_q.q_meta;
    ctx->t_meta = ctx->nucl_t.t_meta;
    return PEP_OK;
}
