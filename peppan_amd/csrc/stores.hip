// Members of the genome-mapping stores straight from numeric columns (host C++, no GPU work, no context).
//
// get_map_bsn (PEPPAN.py:950-966) keeps, per group of merged hits, the group's 16-column hit rows (store .mat) and its packed allele
// (store .seq), 1000 groups per member; a member is one .npy file holding an OBJECT array whose elements are arrays - which numpy
// writes as a pickle.  Making 12 000 x 16 Python objects per genome only to have the pickler walk them again was the floor of the
// mapping path (18 ms of 84 per genome).  Here the pickle stream itself is emitted from the columns of the hit table: the same
// value and Python type per cell the object rows held (int, float, str), one inner ndarray(object)[k, 16] per group, protocol-3
// opcodes only, the two globals / the b'b' argument / the dtype objects shared through the memo exactly as numpy's own pickles
// share them.  np.load(..., allow_pickle=True) - how PEPPAN reads its stores (PEPPAN.py:40-42, 1931) - returns equal arrays.
//
// The callers pass the module that holds `_reconstruct` in the running numpy ("numpy._core.multiarray" in numpy 2,
// "numpy.core.multiarray" before), so a store is written the way the numpy at hand would write it.
#include "../../include/peppan_hip.h"
#include <cstdint>
#include <cstring>

namespace {

struct Stream {
    uint8_t *p;
    int64_t cap, n;
    void bytes(const void *src, int64_t len)
    {
        if (n + len <= cap) memcpy(p + n, src, (size_t)len);
        n += len;                                        // (keeps counting past the end: the caller learns the size it needs)
    }
    void byte(uint8_t b) { bytes(&b, 1); }
    void text(const char *s) { bytes(s, (int64_t)strlen(s)); }
    void u32(uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; bytes(b, 4); }
    void integer(int64_t v)
    {
        if (v >= 0 && v < 256) { byte('K'); byte((uint8_t)v); }
        else if (v >= 0 && v < 65536) { byte('M'); byte((uint8_t)v); byte((uint8_t)(v >> 8)); }
        else if (v >= INT32_MIN && v <= INT32_MAX) { byte('J'); u32((uint32_t)(int32_t)v); }
        else {                                           // LONG1: little-endian two's complement, as few bytes as hold the sign
            int len = 8;
            while (len > 1) {
                const uint8_t top = (uint8_t)((uint64_t)v >> (8 * (len - 1))), below = (uint8_t)((uint64_t)v >> (8 * (len - 2)));
                if ((top == 0x00 && !(below & 0x80)) || (top == 0xFF && (below & 0x80))) --len; else break;
            }
            byte(0x8a); byte((uint8_t)len);
            for (int k = 0; k < len; ++k) byte((uint8_t)((uint64_t)v >> (8 * k)));
        }
    }
    void real(double v)
    {
        uint64_t b;
        memcpy(&b, &v, 8);
        byte('G');
        for (int k = 7; k >= 0; --k) byte((uint8_t)(b >> (8 * k)));
    }
    void unicode(const char *s, uint32_t len) { byte('X'); u32(len); bytes(s, len); }
    void get(uint8_t slot) { byte('h'); byte(slot); }
    void put(uint8_t slot) { byte('q'); byte(slot); }
};

// memo slots
enum { M_RECON = 0, M_NDARRAY = 1, M_B = 2, M_DTYPE_FN = 3, M_BAR = 4, M_OBJ_DTYPE = 5, M_U1_DTYPE = 6 };

// numpy.dtype(code, False, True) + its state (3, '|', None, None, None, -1, -1, flags); slot = where the finished dtype is remembered
void emit_dtype(Stream &s, const char *code, int flags, uint8_t slot, bool first_dtype)
{
    if (first_dtype) { s.text("cnumpy\ndtype\n"); s.put(M_DTYPE_FN); } else s.get(M_DTYPE_FN);
    s.unicode(code, 2);
    s.byte(0x89); s.byte(0x88); s.byte(0x87); s.byte('R'); s.put(slot);
    s.byte('('); s.byte('K'); s.byte(3);
    if (first_dtype) { s.unicode("|", 1); s.put(M_BAR); } else s.get(M_BAR);
    s.byte('N'); s.byte('N'); s.byte('N'); s.byte('J'); s.u32(0xFFFFFFFFu); s.byte('J'); s.u32(0xFFFFFFFFu); s.byte('K'); s.byte((uint8_t)flags);
    s.byte('t'); s.byte('b');
}

// _reconstruct(ndarray, (0,), b'b'): an empty array for BUILD to fill
void emit_blank(Stream &s, const char *recon_module, bool first)
{
    if (first) {
        s.byte('c'); s.text(recon_module); s.text("\n_reconstruct\n"); s.put(M_RECON);
        s.text("cnumpy\nndarray\n"); s.put(M_NDARRAY);
        s.byte('K'); s.byte(0); s.byte(0x85);
        s.byte('C'); s.byte(1); s.byte('b'); s.put(M_B);
    } else {
        s.get(M_RECON); s.get(M_NDARRAY); s.byte('K'); s.byte(0); s.byte(0x85); s.get(M_B);
    }
    s.byte(0x87); s.byte('R');
}

// the outer array: ndarray(object)[n]; afterwards the stream stands inside its element list
void open_outer(Stream &s, const char *recon_module, int64_t n)
{
    s.byte(0x80); s.byte(3);
    emit_blank(s, recon_module, true);
    s.byte('('); s.byte('K'); s.byte(1);
    s.integer(n); s.byte(0x85);
    emit_dtype(s, "O8", 63, M_OBJ_DTYPE, true);
    s.byte(0x89); s.byte(']'); s.byte('(');
}

void close_outer(Stream &s)
{
    s.byte('e'); s.byte('t'); s.byte('b'); s.byte('.');
}

uint32_t cigar_text(char *dst, const uint32_t *runs, int64_t n)
{
    uint32_t len = 0;
    for (int64_t x = 0; x < n; ++x) {
        uint32_t v = runs[x] >> 2;
        char tmp[12];
        int d = 0;
        do { tmp[d++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (d) dst[len++] = tmp[--d];
        dst[len++] = "MID?"[runs[x] & 3u];
    }
    return len;
}

}  // namespace

extern "C" int64_t pep_store_mat_member(const pep_mat_cols *c, const int64_t *row_off, int64_t n_groups, const char *recon_module, uint8_t *out, int64_t cap)
{
    if (!c || !row_off || !recon_module || n_groups < 0 || (cap > 0 && !out)) return PEP_ERR_ARG;
    Stream s{out, cap, 0};
    open_outer(s, recon_module, n_groups);
    char small[256];
    char *text = small;
    int64_t text_cap = sizeof(small);
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t lo = row_off[g], hi = row_off[g + 1];
        if (hi < lo) { if (text != small) delete[] text; return PEP_ERR_ARG; }
        emit_blank(s, recon_module, false);
        s.byte('('); s.byte('K'); s.byte(1);
        s.integer(hi - lo); s.integer(16); s.byte(0x86);
        s.get(M_OBJ_DTYPE);
        s.byte(0x89); s.byte(']'); s.byte('(');
        for (int64_t k = lo; k < hi; ++k) {
            s.integer(c->q[k]); s.integer(c->r[k]); s.real(c->iden[k]);
            s.integer(c->aln[k]); s.integer(c->mis[k]); s.integer(c->gap[k]);
            s.integer(c->qs[k]); s.integer(c->qe[k]); s.integer(c->ss[k]); s.integer(c->se[k]);
            s.real(c->evalue[k]);
            if (c->score_is_int) s.integer((int64_t)c->score[k]); else s.real(c->score[k]);
            s.integer(c->ql[k]); s.integer(c->sl[k]);
            const int64_t need = c->c_runs[k] * 12 + 1;
            if (need > text_cap) {
                if (text != small) delete[] text;
                text_cap = need * 2;
                text = new char[(size_t)text_cap];
            }
            const uint32_t len = cigar_text(text, c->arena + c->c_off[k], c->c_runs[k]);
            s.unicode(text, len);
            s.integer(c->rid[k]);
        }
        s.byte('e'); s.byte('t'); s.byte('b');
    }
    if (text != small) delete[] text;
    close_outer(s);
    return s.n;
}

extern "C" int64_t pep_store_seq_member(const uint8_t *packed, const int64_t *pack_off, int64_t n_groups, const char *recon_module, uint8_t *out, int64_t cap)
{
    if (!pack_off || !recon_module || n_groups < 0 || (cap > 0 && !out)) return PEP_ERR_ARG;
    Stream s{out, cap, 0};
    open_outer(s, recon_module, n_groups);
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t lo = pack_off[g], len = pack_off[g + 1] - lo;
        if (len < 0 || (len > 0 && !packed)) return PEP_ERR_ARG;
        emit_blank(s, recon_module, false);
        s.byte('('); s.byte('K'); s.byte(1);
        s.integer(len); s.byte(0x85);
        if (g == 0) emit_dtype(s, "u1", 0, M_U1_DTYPE, false); else s.get(M_U1_DTYPE);
        s.byte(0x89);
        if (len < 256) { s.byte('C'); s.byte((uint8_t)len); } else { s.byte('B'); s.u32((uint32_t)len); }
        s.bytes(packed + lo, len);
        s.byte('t'); s.byte('b');
    }
    close_outer(s);
    return s.n;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The gene table store (.tab): one member per gene, int64 rows.  MapBsn.update (PEPPAN.py:91-113) writes them one by one - at 10 000 genes
// that is 10 000 trips through numpy's header writer, zlib and zipfile, a second of interpreter time per update.  When the archive is EMPTY
// (the usual case: the table is kept in memory and written once) every member's complete zip entry - local file header, .npy header, rows,
// raw-deflated at level 1 from 4 KiB on - is made here, the members dealt to a few threads, and the caller writes the lot with one write()
// and lists the entries in the archive's directory.
#include <zlib.h>
#include <string>
#include <thread>
#include <vector>

namespace {

void put16(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
void put32(std::vector<uint8_t> &v, uint32_t x) { put16(v, x & 0xFFFFu); put16(v, x >> 16); }

struct TabJob {
    const int64_t *rows, *order, *off, *key;
    int64_t n_cols, lo, hi;
    uint32_t dos_time, dos_date;
    uint32_t *crc;
    int64_t *csize, *usize, *at;            // at[]: relative to the job's buffer until the pieces are put together
    uint8_t *method;                        // (optional) 0 stored / 8 deflated per member
    std::vector<uint8_t> buf;
    bool ok = true;
};

}   // namespace
extern "C" int64_t pep_deflate_fast(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap);
extern "C" uint32_t pep_crc32(const uint8_t *src, int64_t n, uint32_t crc);
namespace {

void tab_members(TabJob *j)
{
    std::vector<uint8_t> npy, packed;
    for (int64_t m = j->lo; m < j->hi; ++m) {
        const int64_t k = j->off[m + 1] - j->off[m];
        // .npy, format 1.0: magic, version, header length, the dictionary padded with blanks to a multiple of 64 and closed by a newline
        char dict[128];
        const int dl = snprintf(dict, sizeof dict, "{'descr': '<i8', 'fortran_order': False, 'shape': (%lld, %lld), }", (long long)k, (long long)j->n_cols);
        const int hlen = ((10 + dl + 1 + 63) / 64) * 64 - 10;
        const int64_t body = k * j->n_cols * 8;
        npy.resize((size_t)(10 + hlen + body));
        memcpy(npy.data(), "\x93NUMPY\x01\x00", 8);
        npy[8] = (uint8_t)hlen; npy[9] = (uint8_t)(hlen >> 8);
        memcpy(npy.data() + 10, dict, (size_t)dl);
        memset(npy.data() + 10 + dl, ' ', (size_t)(hlen - dl - 1));
        npy[10 + hlen - 1] = '\n';
        if (j->order) {                              // the table is not sorted: row i of the member is row order[off[m] + i] of it
            uint8_t *dst = npy.data() + 10 + hlen;
            for (int64_t i = 0; i < k; ++i) memcpy(dst + i * j->n_cols * 8, j->rows + j->order[j->off[m] + i] * j->n_cols, (size_t)j->n_cols * 8);
        } else memcpy(npy.data() + 10 + hlen, j->rows + j->off[m] * j->n_cols, (size_t)body);
        const uint32_t crc = pep_crc32(npy.data(), (int64_t)npy.size(), 0u);
        const uint8_t *payload = npy.data();
        int64_t plen = (int64_t)npy.size();
        int method = 0;
        if (plen >= 4096) {                          // small integers, eight bytes each: the single-probe matcher of the .mat members (three times zlib's level-1 rate at its sizes)
            packed.resize(npy.size() + npy.size() / 8 + 1024);
            const int64_t got = pep_deflate_fast(npy.data(), (int64_t)npy.size(), packed.data(), (int64_t)packed.size());
            if (got < 0 || got > (int64_t)packed.size()) { j->ok = false; break; }
            payload = packed.data(); plen = got; method = 8;
        }
        const std::string name = std::to_string((long long)j->key[m]);
        j->crc[m] = crc; j->csize[m] = plen; j->usize[m] = (int64_t)npy.size(); j->at[m] = (int64_t)j->buf.size();
        if (j->method) j->method[m] = (uint8_t)method;
        std::vector<uint8_t> &b = j->buf;
        put32(b, 0x04034b50u); put16(b, 20); put16(b, 0); put16(b, (uint32_t)method); put16(b, j->dos_time); put16(b, j->dos_date);
        put32(b, crc); put32(b, (uint32_t)plen); put32(b, (uint32_t)npy.size()); put16(b, (uint32_t)name.size()); put16(b, 0);
        b.insert(b.end(), name.begin(), name.end());
        b.insert(b.end(), payload, payload + plen);
    }
}

}   // namespace

static int64_t tab_entries(const int64_t *rows, int64_t n_cols, const int64_t *order, const int64_t *off, const int64_t *key, int64_t n_members, uint32_t dos_time, uint32_t dos_date,
                           int32_t threads, uint8_t *out, int64_t cap, uint32_t *crc, int64_t *csize, int64_t *usize, int64_t *at, uint8_t *method)
{
    if (n_members < 0 || n_cols < 1 || !off || (n_members && (!rows || !key || !crc || !csize || !usize || !at)) || (cap > 0 && !out)) return PEP_ERR_ARG;
    for (int64_t m = 0; m < n_members; ++m)
        if (off[m + 1] < off[m] || (off[m + 1] - off[m]) * n_cols * 8 > (int64_t)0x7FFFFF00) return PEP_ERR_ARG;       // (a member stays below 2 GiB: no zip64 entry)
    const int64_t T = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(threads, 64), (n_members + 255) / 256));
    std::vector<TabJob> jobs((size_t)T);
    std::vector<std::thread> pool;
    for (int64_t t = 0; t < T; ++t) {
        TabJob &j = jobs[(size_t)t];
        j.rows = rows; j.order = order; j.off = off; j.key = key; j.n_cols = n_cols; j.lo = n_members * t / T; j.hi = n_members * (t + 1) / T;
        j.dos_time = dos_time; j.dos_date = dos_date; j.crc = crc; j.csize = csize; j.usize = usize; j.at = at; j.method = method;
        if (t + 1 < T) pool.emplace_back(tab_members, &j);
    }
    tab_members(&jobs[(size_t)T - 1]);
    for (auto &th : pool) th.join();
    int64_t total = 0;
    for (auto &j : jobs) { if (!j.ok) return PEP_ERR_ARG; total += (int64_t)j.buf.size(); }
    if (total > cap) return total;                       // (the caller calls again with a buffer of this size)
    int64_t base = 0;
    for (auto &j : jobs) {
        memcpy(out + base, j.buf.data(), j.buf.size());
        for (int64_t m = j.lo; m < j.hi; ++m) at[m] += base;
        base += (int64_t)j.buf.size();
    }
    return total;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Entropy coding for the members of the .seq store: a raw DEFLATE stream (RFC 1951) made of dynamic-Huffman blocks that hold literals only.
// Packed alleles carry no repeats worth a match search (section 5 of DESIGN.md); zlib's own Z_HUFFMAN_ONLY mode gets the same sizes at
// 120 MB/s per thread - 18 ms of CPU per mapped genome, 40 % of what a genome costs the host on a box that grants 16 CPUs.  This coder does
// nothing but count, build a length-limited Huffman code per 128 KiB block and pack bits.  Any inflate reads the result.
#include <algorithm>

namespace {

struct BitSink {
    uint8_t *p;
    int64_t cap, n;
    uint64_t acc;
    int fill;
    void put(uint32_t bits, int len)
    {
        acc |= (uint64_t)bits << fill;
        fill += len;
        while (fill >= 32) {
            if (n + 4 <= cap) { p[n] = (uint8_t)acc; p[n + 1] = (uint8_t)(acc >> 8); p[n + 2] = (uint8_t)(acc >> 16); p[n + 3] = (uint8_t)(acc >> 24); }
            n += 4; acc >>= 32; fill -= 32;
        }
    }
    void finish()
    {
        while (fill > 0) { if (n < cap) p[n] = (uint8_t)acc; ++n; acc >>= 8; fill -= 8; }
        fill = 0;
    }
    // room for `bits` more bits written eight bytes at a time (the stores below run past the last bit by up to seven bytes)
    bool roomy(int64_t bits) const { return n + ((fill + bits + 7) >> 3) + 16 <= cap; }
};

// The sink's hot loop when the room is known to suffice: whole bytes leave the 64-bit accumulator with ONE unaligned 8-byte store per group of codes
// (<= 56 bits a group, <= 7 bits left behind), no bounds checked.  Opened on a BitSink, closed back into it.
struct FastBits {
    BitSink &s;
    uint8_t *o;
    uint64_t acc;
    int fill;
    explicit FastBits(BitSink &sink) : s(sink), o(sink.p + sink.n), acc(sink.acc), fill(sink.fill)
    {
        while (fill >= 8) { *o++ = (uint8_t)acc; acc >>= 8; fill -= 8; }
    }
    __attribute__((always_inline)) void put(uint64_t bits, int len)          // len <= 56
    {
        acc |= bits << fill;
        fill += len;
        memcpy(o, &acc, 8);
        o += fill >> 3;
        acc >>= fill & ~7;
        fill &= 7;
    }
    void close() { s.n = o - s.p; s.acc = acc; s.fill = fill; }
};

// code lengths (<= limit <= 15) of a Huffman code for the symbols with freq > 0 (at least two of them)
void huffman_lengths(const uint32_t *freq_in, int n_sym, uint8_t *len, int limit = 15)
{
    uint32_t freq[288];
    for (int i = 0; i < n_sym; ++i) freq[i] = freq_in[i];
    for (;;) {
        struct Node { uint64_t w; int left, right; };
        Node nodes[2 * 288];
        int order[288], n_leaf = 0;
        for (int i = 0; i < n_sym; ++i) if (freq[i]) order[n_leaf++] = i;
        std::sort(order, order + n_leaf, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        // two-queue construction: leaves in weight order, internal nodes are made in weight order
        int n_nodes = 0, qa = 0, qb = 0, first_internal = n_leaf;
        for (int i = 0; i < n_leaf; ++i) nodes[n_nodes++] = Node{freq[order[i]], -1, -1};
        auto take = [&]() {
            const bool leaf = qa < n_leaf && (first_internal + qb >= n_nodes || nodes[qa].w <= nodes[first_internal + qb].w);
            return leaf ? qa++ : first_internal + qb++;
        };
        while (n_leaf - qa + (n_nodes - first_internal - qb) > 1) {
            const int x = take(), y = take();
            nodes[n_nodes++] = Node{nodes[x].w + nodes[y].w, x, y};
        }
        int depth[2 * 288];
        depth[n_nodes - 1] = 0;
        int deepest = 0;
        for (int k = n_nodes - 1; k >= n_leaf; --k) {
            depth[nodes[k].left] = depth[nodes[k].right] = depth[k] + 1;
            deepest = std::max(deepest, depth[k] + 1);
        }
        if (deepest <= limit) {
            for (int i = 0; i < n_sym; ++i) len[i] = 0;
            for (int i = 0; i < n_leaf; ++i) len[order[i]] = (uint8_t)depth[i];
            return;
        }
        for (int i = 0; i < n_sym; ++i) if (freq[i]) freq[i] = (freq[i] + 1) / 2;          // flatter weights, shallower tree
    }
}

void canonical_codes(const uint8_t *len, int n_sym, uint16_t *code)
{
    int count[16] = {0}, next[16];
    for (int i = 0; i < n_sym; ++i) ++count[len[i]];
    count[0] = 0;
    int c = 0;
    for (int b = 1; b <= 15; ++b) { c = (c + count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n_sym; ++i) {
        if (!len[i]) { code[i] = 0; continue; }
        uint32_t v = (uint32_t)next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) { r = (r << 1) | (v & 1u); v >>= 1; }      // packed LSB first: the code goes in bit-reversed
        code[i] = (uint16_t)r;
    }
}

}   // namespace

// CRC-32 of the zip format (IEEE 802.3, reflected) by carry-less multiplication: four 128-bit lanes folded per 64 bytes, then 128 -> 64 -> 32 bits by Barrett
// reduction - the published folding scheme for this polynomial (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ").  zlib's table
// walk does 1 - 2 GB/s; a store member is checksummed once where it is made, 2.9 MB per mapped genome.  Without the instruction: zlib's crc32.
#include <immintrin.h>

namespace {

__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_load(const uint8_t *p) { return _mm_loadu_si128(reinterpret_cast<const __m128i *>(p)); }
__attribute__((target("pclmul,sse4.1"))) inline __m128i crc_fold(__m128i x, __m128i k, __m128i next)
{
    return _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x11), _mm_clmulepi64_si128(x, k, 0x00)), next);
}

__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(const uint8_t *buf, int64_t len, uint32_t crc)        // len >= 64, a multiple of 16; crc: the running (inverted) register
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll), k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124ll), poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x1 = _mm_xor_si128(crc_load(buf), _mm_cvtsi32_si128((int)crc)), x2 = crc_load(buf + 16), x3 = crc_load(buf + 32), x4 = crc_load(buf + 48);
    buf += 64; len -= 64;
    while (len >= 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00), a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), a1), crc_load(buf));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), a2), crc_load(buf + 16));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), a3), crc_load(buf + 32));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), a4), crc_load(buf + 48));
        buf += 64; len -= 64;
    }
    x1 = crc_fold(x1, k3k4, x2); x1 = crc_fold(x1, k3k4, x3); x1 = crc_fold(x1, k3k4, x4);
    while (len >= 16) { x1 = crc_fold(x1, k3k4, crc_load(buf)); buf += 16; len -= 16; }
    const __m128i low32 = _mm_setr_epi32(~0, 0, ~0, 0);
    x2 = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, low32), k5, 0x00), x2);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x1, low32), poly, 0x10);
    x2 = _mm_clmulepi64_si128(_mm_and_si128(x2, low32), poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, x2), 1);
}

const bool g_have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");

}   // namespace

// crc32(src[0 .. n)) continued from `crc` (0 to start): the value zlib.crc32 / the zip directory holds
extern "C" uint32_t pep_crc32(const uint8_t *src, int64_t n, uint32_t crc)
{
    if (n <= 0 || !src) return crc;
    if (g_have_clmul && n >= 64) {
        const int64_t body = n & ~(int64_t)15;
        crc = ~crc32_clmul(src, body, ~crc);
        src += body; n -= body;
    }
    while (n > 0) {                                   // (zlib takes a 32-bit length)
        const int64_t m = std::min<int64_t>(n, (int64_t)1 << 30);
        crc = (uint32_t)crc32((uLong)crc, src, (uInt)m);
        src += m; n -= m;
    }
    return crc;
}

extern "C" int64_t pep_deflate_literals(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap)
{
    if (n < 0 || (n > 0 && !src) || (cap > 0 && !out)) return PEP_ERR_ARG;
    BitSink s{out, cap, 0, 0, 0};
    if (n == 0) { s.put(1, 1); s.put(1, 2); s.put(0, 7); s.finish(); return s.n; }         // one final block of the fixed code holding the end mark only
    const int64_t BLOCK = 128 << 10;
    static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int64_t at = 0; at < n; at += BLOCK) {
        const int64_t m = std::min(BLOCK, n - at);
        const uint8_t *p = src + at;
        uint32_t freq[257] = {0}, f1[256] = {0}, f2[256] = {0}, f3[256] = {0};            // four counters side by side: a run of one byte value does not wait for its own last store
        int64_t i = 0;
        for (; i + 4 <= m; i += 4) { ++freq[p[i]]; ++f1[p[i + 1]]; ++f2[p[i + 2]]; ++f3[p[i + 3]]; }
        for (; i < m; ++i) ++freq[p[i]];
        for (int k = 0; k < 256; ++k) freq[k] += f1[k] + f2[k] + f3[k];
        freq[256] = 1;                                                                   // the end mark
        uint8_t len[257];
        uint16_t code[257];
        huffman_lengths(freq, 257, len, 14);                                             // (14 bits at most: four codes per store of the loop below; the 15th bit is worth nothing on a 256-letter alphabet)
        canonical_codes(len, 257, code);
        s.put(at + m >= n ? 1u : 0u, 1);                                                 // BFINAL
        s.put(2, 2);                                                                     // BTYPE = dynamic Huffman
        s.put(0, 5); s.put(1, 5); s.put(15, 4);                                          // 257 literal/length codes, 2 distance codes, all 19 code-length codes listed
        for (int k = 0; k < 19; ++k) s.put(cl_order[k] >= 16 ? 0u : 4u, 3);              // lengths 0..15 cost four bits each (the complete 4-bit code), no repeat codes
        auto put_len = [&](int v) { uint32_t r = 0; for (int b = 0; b < 4; ++b) r = (r << 1) | ((v >> b) & 1); s.put(r, 4); };      // code of length symbol v = v itself, bit-reversed
        for (int k = 0; k < 257; ++k) put_len(len[k]);
        put_len(1); put_len(1);                                                          // two distance codes of one bit: a complete code that the data never uses
        int64_t bits = len[256];
        for (int k = 0; k < 256; ++k) bits += (int64_t)freq[k] * len[k];
        i = 0;
        if (s.roomy(bits)) {
            FastBits f(s);
            for (; i + 4 <= m; i += 4) {                                                  // four symbols (<= 56 bits) per store
                const int la = len[p[i]], lb = la + len[p[i + 1]], lc = lb + len[p[i + 2]];
                f.put((uint64_t)code[p[i]] | ((uint64_t)code[p[i + 1]] << la) | ((uint64_t)code[p[i + 2]] << lb) | ((uint64_t)code[p[i + 3]] << lc), lc + len[p[i + 3]]);
            }
            for (; i < m; ++i) f.put(code[p[i]], len[p[i]]);
            f.close();
        } else {
            for (; i < m; ++i) s.put(code[p[i]], len[p[i]]);
        }
        s.put(code[256], len[256]);
    }
    s.finish();
    return s.n;
}

// A raw DEFLATE stream with matches: one probe of a 4-byte hash per position (greedy, no chains, no lazy evaluation - the matcher of the "fastest"
// levels of the modern deflate libraries), then dynamic-Huffman blocks over literals, lengths and distances.  For the members of the .mat store - the
// .npy pickle stream of a thousand groups of hit rows: repeated opcodes, shared prefixes of numbers - zlib's level 1 gets 0.56 of the size at 39 MB/s on
// a core of this container; this gets about the same size at several times the rate (the share of a mapped genome's CPU time that went into deflating
// its hit rows was the largest single item left).  Any inflate reads the result.
namespace {

struct LenDistTables {
    uint16_t len_sym[259];          // match length 3..258 -> literal/length symbol 257..285
    uint8_t len_extra[29], dist_extra[30];
    uint16_t len_base[29], dist_base[30];
    LenDistTables()
    {
        static const uint16_t lb[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t db[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t de[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (int i = 0; i < 29; ++i) { len_base[i] = lb[i]; len_extra[i] = le[i]; }
        for (int i = 0; i < 30; ++i) { dist_base[i] = db[i]; dist_extra[i] = de[i]; }
        for (int l = 3; l <= 258; ++l) {
            int k = 28;
            while (lb[k] > l) --k;
            if (l == 258) k = 28;
            len_sym[l] = (uint16_t)(257 + k);
        }
    }
    int dist_sym(uint32_t d) const           // distance 1..32768 -> symbol 0..29
    {
        if (d <= 4) return (int)d - 1;
        const int b = 31 - __builtin_clz(d - 1);          // d - 1 in [2^b, 2^(b+1))
        return 2 * b + (int)(((d - 1) >> (b - 1)) & 1u);
    }
};
const LenDistTables g_ld;

}   // namespace

extern "C" int64_t pep_deflate_fast(const uint8_t *src, int64_t n, uint8_t *out, int64_t cap)
{
    if (n < 0 || (n > 0 && !src) || (cap > 0 && !out)) return PEP_ERR_ARG;
    BitSink s{out, cap, 0, 0, 0};
    if (n == 0) { s.put(1, 1); s.put(1, 2); s.put(0, 7); s.finish(); return s.n; }
    const int HASH_BITS = 15;
    std::vector<int32_t> head((size_t)1 << HASH_BITS, -1);
    const int64_t CHUNK = 64 << 10;
    std::vector<uint32_t> tok((size_t)CHUNK + 8);                  // literal: the byte; match: 1 << 31 | length << 16 | distance - 1
    auto load32 = [&](int64_t p) { uint32_t v; memcpy(&v, src + p, 4); return v; };
    auto put_len4 = [&](int v) { uint32_t r = 0; for (int b = 0; b < 4; ++b) r = (r << 1) | ((v >> b) & 1); s.put(r, 4); };
    int64_t pos = 0;
    while (pos < n) {
        const int64_t end = std::min(n, pos + CHUNK);
        size_t nt = 0;
        uint32_t f_lit[286] = {0}, f_dist[30] = {0};
        uint32_t misses = 0;                                                 // positions in a row without a match: data that does not repeat is probed at every second, fourth ... position
        while (pos < end) {
            if (pos + 4 <= n && (misses < 32 || (pos & ((1u << std::min(6u, misses >> 5)) - 1u)) == 0)) {
                const uint32_t v = load32(pos);
                const uint32_t h = (v * 2654435761u) >> (32 - HASH_BITS);
                const int32_t cand = head[h];
                head[h] = (int32_t)(pos & 0x7FFFFFFF);
                if (cand >= 0 && pos - cand <= 32768 && pos < 0x7FFFFFFF && load32(cand) == v) {
                    int64_t len = 4;
                    const int64_t lim = std::min<int64_t>(258, n - pos);
                    while (len + 8 <= lim) {
                        uint64_t a, b;
                        memcpy(&a, src + pos + len, 8); memcpy(&b, src + cand + len, 8);
                        if (a != b) { len += __builtin_ctzll(a ^ b) >> 3; goto matched; }
                        len += 8;
                    }
                    while (len < lim && src[pos + len] == src[cand + len]) ++len;
                matched:
                    if (len > lim) len = lim;
                    const uint32_t d = (uint32_t)(pos - cand);
                    tok[nt++] = 0x80000000u | ((uint32_t)len << 16) | (d - 1u);
                    ++f_lit[g_ld.len_sym[len]];
                    ++f_dist[g_ld.dist_sym(d)];
                    // (one more table entry inside the match keeps long repeats findable; every position would cost more than it finds)
                    if (len >= 8 && pos + len + 4 <= n) { const uint32_t w = load32(pos + len - 4); head[(w * 2654435761u) >> (32 - HASH_BITS)] = (int32_t)(pos + len - 4); }
                    pos += len;
                    misses = 0;
                    continue;
                }
            }
            ++misses;
            tok[nt++] = src[pos];
            ++f_lit[src[pos]];
            ++pos;
        }
        f_lit[256] = 1;
        int n_dist_used = 0;
        for (int i = 0; i < 30; ++i) n_dist_used += f_dist[i] ? 1 : 0;
        if (n_dist_used < 2) { f_dist[0] += 1; f_dist[1] += 1; }            // (a complete distance code needs two symbols)
        int n_lit_used = 0;
        for (int i = 0; i < 286; ++i) n_lit_used += f_lit[i] ? 1 : 0;
        if (n_lit_used < 2) f_lit[f_lit[0] ? 1 : 0] += 1;
        uint8_t l_len[286], d_len[30];
        uint16_t l_code[286], d_code[30];
        huffman_lengths(f_lit, 286, l_len);
        huffman_lengths(f_dist, 30, d_len);
        canonical_codes(l_len, 286, l_code);
        canonical_codes(d_len, 30, d_code);
        static const uint8_t cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        s.put(pos >= n ? 1u : 0u, 1);                                        // BFINAL
        s.put(2, 2);                                                         // BTYPE = dynamic Huffman
        s.put(286 - 257, 5); s.put(30 - 1, 5); s.put(15, 4);                 // all 286 literal/length codes, all 30 distance codes, all 19 code-length codes listed
        for (int k = 0; k < 19; ++k) s.put(cl_order[k] >= 16 ? 0u : 4u, 3);  // lengths 0..15 as the complete 4-bit code, no repeat codes (158 bytes per 64 KiB block)
        for (int i = 0; i < 286; ++i) put_len4(l_len[i]);
        for (int i = 0; i < 30; ++i) put_len4(d_len[i]);
        int64_t bits = l_len[256];
        for (int i = 0; i < 286; ++i) bits += (int64_t)f_lit[i] * (l_len[i] + (i >= 257 ? g_ld.len_extra[i - 257] : 0));
        for (int i = 0; i < 30; ++i) bits += (int64_t)f_dist[i] * (d_len[i] + g_ld.dist_extra[i]);
        tok[nt] = 0x80000000u;                                               // (a sentinel behind the last token: the pairing below looks one ahead)
        if (s.roomy(bits)) {
            FastBits f(s);
            for (size_t k = 0; k < nt;) {
                const uint32_t t = tok[k];
                if (!(t & 0x80000000u)) {
                    const uint32_t u = tok[k + 1];
                    if (!(u & 0x80000000u)) {                                // two literals (<= 30 bits) per store
                        f.put((uint64_t)l_code[t] | ((uint64_t)l_code[u] << l_len[t]), l_len[t] + l_len[u]);
                        k += 2;
                    } else { f.put(l_code[t], l_len[t]); ++k; }
                    continue;
                }
                const uint32_t len = (t >> 16) & 0x1FFu, d = (t & 0xFFFFu) + 1u;
                const int ls = g_ld.len_sym[len], li = ls - 257, ds = g_ld.dist_sym(d);
                // length symbol + its extra bits (<= 20 bits), then distance symbol + its extra bits (<= 28 bits): one store
                const int ll = l_len[ls] + g_ld.len_extra[li];
                f.put(((uint64_t)l_code[ls] | ((uint64_t)(len - g_ld.len_base[li]) << l_len[ls])) |
                      (((uint64_t)d_code[ds] | ((uint64_t)(d - g_ld.dist_base[ds]) << d_len[ds])) << ll), ll + d_len[ds] + g_ld.dist_extra[ds]);
                ++k;
            }
            f.close();
        } else {
            for (size_t k = 0; k < nt; ++k) {
                const uint32_t t = tok[k];
                if (!(t & 0x80000000u)) { s.put(l_code[t], l_len[t]); continue; }
                const uint32_t len = (t >> 16) & 0x1FFu, d = (t & 0xFFFFu) + 1u;
                const int ls = g_ld.len_sym[len], li = ls - 257, ds = g_ld.dist_sym(d);
                s.put((uint32_t)l_code[ls] | ((len - g_ld.len_base[li]) << l_len[ls]), l_len[ls] + g_ld.len_extra[li]);
                s.put((uint32_t)d_code[ds] | ((d - g_ld.dist_base[ds]) << d_len[ds]), d_len[ds] + g_ld.dist_extra[ds]);
            }
        }
        s.put(l_code[256], l_len[256]);
    }
    s.finish();
    return s.n;
}

// A store member ready for its archive in ONE call: the raw DEFLATE stream of `src` by the coder asked for (0: pep_deflate_literals, 1: pep_deflate_fast) and the
// CRC-32 of `src` that the member's zip header carries.  Returns the stream's length (> cap: nothing usable written, as the coders do).
extern "C" int64_t pep_pack_member(const uint8_t *src, int64_t n, int32_t coder, uint8_t *out, int64_t cap, uint32_t *crc)
{
    if (!crc || coder < 0 || coder > 1) return PEP_ERR_ARG;
    const int64_t got = coder == 0 ? pep_deflate_literals(src, n, out, cap) : pep_deflate_fast(src, n, out, cap);
    if (got < 0) return got;
    *crc = pep_crc32(src, n, 0u);
    return got;
}

extern "C" int64_t pep_store_tab_members(const int64_t *rows, int64_t n_cols, const int64_t *order, const int64_t *off, const int64_t *key, int64_t n_members, uint32_t dos_time, uint32_t dos_date,
                                         int32_t threads, uint8_t *out, int64_t cap, uint32_t *crc, int64_t *csize, int64_t *usize, int64_t *at)
{
    return tab_entries(rows, n_cols, order, off, key, n_members, dos_time, dos_date, threads, out, cap, crc, csize, usize, at, nullptr);
}

// The whole <prefix>.tab.npz in one piece: the entries as above, then the archive's central directory and end record - a store that is written
// once and closed needs no zipfile object at all (10 000 ZipInfo objects made and walked again by ZipFile.close(): 0.2 s; 50 000: 1 s).
// Plain zip only: fewer than 65 535 members and less than 4 GiB, else PEP_ERR_LIMIT (the caller takes the member-wise way).
extern "C" int64_t pep_store_tab_archive(const int64_t *rows, int64_t n_cols, const int64_t *order, const int64_t *off, const int64_t *key, int64_t n_members, uint32_t dos_time, uint32_t dos_date,
                                         int32_t threads, uint8_t *out, int64_t cap)
{
    if (n_members < 0) return PEP_ERR_ARG;
    if (n_members >= 65535) return PEP_ERR_LIMIT;
    std::vector<uint32_t> crc((size_t)n_members + 1);
    std::vector<int64_t> csize((size_t)n_members + 1), usize((size_t)n_members + 1), at((size_t)n_members + 1);
    std::vector<uint8_t> method((size_t)n_members + 1);
    std::vector<uint8_t> dir;
    // the directory's size is known before the entries are made: 46 bytes + the name per member, 22 for the end record
    int64_t dir_bytes = 22;
    for (int64_t m = 0; m < n_members; ++m) dir_bytes += 46 + (int64_t)std::to_string((long long)key[m]).size();
    const int64_t room = cap > dir_bytes ? cap - dir_bytes : 0;
    const int64_t body = tab_entries(rows, n_cols, order, off, key, n_members, dos_time, dos_date, threads, out, room, crc.data(), csize.data(), usize.data(), at.data(), method.data());
    if (body < 0) return body;
    if (body + dir_bytes >= (int64_t)0xFFFFFFFFll) return PEP_ERR_LIMIT;
    if (body > room || cap < body + dir_bytes || !out) return body + dir_bytes;      // (nothing usable written: call again with this much - also for an archive without members and no room for its end record)
    dir.reserve((size_t)dir_bytes);
    for (int64_t m = 0; m < n_members; ++m) {
        const std::string name = std::to_string((long long)key[m]);
        put32(dir, 0x02014b50u); put16(dir, 20 | (3u << 8)); put16(dir, 20); put16(dir, 0); put16(dir, method[(size_t)m]); put16(dir, dos_time); put16(dir, dos_date);
        put32(dir, crc[(size_t)m]); put32(dir, (uint32_t)csize[(size_t)m]); put32(dir, (uint32_t)usize[(size_t)m]); put16(dir, (uint32_t)name.size()); put16(dir, 0); put16(dir, 0);
        put16(dir, 0); put16(dir, 0); put32(dir, 0600u << 16); put32(dir, (uint32_t)at[(size_t)m]);
        dir.insert(dir.end(), name.begin(), name.end());
    }
    const uint32_t cd_size = (uint32_t)dir.size();
    put32(dir, 0x06054b50u); put16(dir, 0); put16(dir, 0); put16(dir, (uint32_t)n_members); put16(dir, (uint32_t)n_members); put32(dir, cd_size); put32(dir, (uint32_t)body); put16(dir, 0);
    memcpy(out + body, dir.data(), dir.size());
    return body + (int64_t)dir.size();
}
