// C ABI of libpeppan_hip.so (declared in include/peppan_hip.h): context, inputs, orchestration of K1..K8.
#include "common.h"
#include <chrono>
#include <time.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

int pep_fail(pep_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) { ctx->n_down = 0; ctx->pin_down_used = 0; }          // (downloads that were on their way belong to the call that failed)
    if (ctx) {
        ctx->err = msg;
        ctx->n_pending = 0;              // queued read-backs point at the failing caller's locals: drop them
        ctx->pin_small_used = 0;
        // A failure between "tickets taken on the host" and "kernel queued" (a reservation that fails between two pep_lookback_begin calls,
        // a launch that is never made) leaves the host's ticket bases ahead of the device counters, and counter blocks half used: whatever
        // runs next on this context starts from cleared state areas and a filled counter block instead of trusting any of it.
        for (auto &st : ctx->fused_state) st.dirty = true;
        for (auto &st : ctx->scan_state) st.dirty = true;
        ctx->sort_dirty = true;
        ctx->zero_clean = false;
        for (bool &f : ctx->zero_ok) f = false;
        ctx->set_clean_slots = 0;
    }
    return code;
}

int dev_reserve(pep_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return PEP_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = std::max<size_t>(bytes + bytes / 4, 256);
    want = (want + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return pep_fail(ctx, PEP_ERR_HIP, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    b.cap = want;
    return PEP_OK;
}

void dev_release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int pin_reserve(pep_ctx *ctx, PinBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return PEP_OK;
    if (b.p) (void)hipHostFree(b.p);
    b.p = nullptr; b.cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    PEP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&b.p), want, hipHostMallocDefault));
    b.cap = want;
    return PEP_OK;
}

int pep_h2d(pep_ctx *ctx, void *d_dst, const void *h_src, size_t n)
{
    if (n == 0) return PEP_OK;
    if (n < ((size_t)64 << 10) || n > ((size_t)256 << 20)) {          // small: the runtime's own staging does as well; huge: not worth a staging area of that size
        PEP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, ctx->stream));
        return PEP_OK;
    }
    const size_t need = (n + 255) & ~(size_t)255;
    if (ctx->pin_up_used + need > ctx->pin_up.cap) {
        // the area is full or too small: whatever was queued out of it must have left before it is used again (or replaced by a larger one)
        if (ctx->up_event_set) { PEP_HIP(ctx, pep_event_wait(ctx->up_event)); ctx->up_event_set = false; }
        ctx->pin_up_used = 0;
        if (need > ctx->pin_up.cap) PEP_TRY(pin_reserve(ctx, ctx->pin_up, std::max(need, (size_t)32 << 20)));
    }
    memcpy(ctx->pin_up.p + ctx->pin_up_used, h_src, n);
    PEP_HIP(ctx, hipMemcpyAsync(d_dst, ctx->pin_up.p + ctx->pin_up_used, n, hipMemcpyHostToDevice, ctx->stream));
    ctx->pin_up_used += need;
    if (!ctx->up_event && hipEventCreateWithFlags(&ctx->up_event, pep_wait_event_flags()) != hipSuccess) return pep_fail(ctx, PEP_ERR_HIP, "hipEventCreate failed");
    PEP_HIP(ctx, hipEventRecord(ctx->up_event, ctx->stream));
    ctx->up_event_set = true;
    return PEP_OK;
}

int pep_d2h_queue(pep_ctx *ctx, void *h_dst, const void *d_src, size_t n)
{
    if (n == 0) return PEP_OK;
    const size_t need = (n + 255) & ~(size_t)255;
    if (n < ((size_t)64 << 10) || n > ((size_t)256 << 20) || ctx->n_down >= 16) {
        PEP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, ctx->stream));
        return PEP_OK;
    }
    if (ctx->pin_down_used + need > ctx->pin_down.cap) {
        if (ctx->n_down) {                                       // (downloads are waiting in the area: it cannot be replaced under them)
            PEP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, ctx->stream));
            return PEP_OK;
        }
        PEP_HIP(ctx, pep_stream_wait(ctx));
        PEP_TRY(pin_reserve(ctx, ctx->pin_down, std::max(need, (size_t)16 << 20)));
        ctx->pin_down_used = 0;
    }
    PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_down.p + ctx->pin_down_used, d_src, n, hipMemcpyDeviceToHost, ctx->stream));
    ctx->down[ctx->n_down++] = pep_ctx::PendingDown{h_dst, ctx->pin_down_used, n};
    ctx->pin_down_used += need;
    return PEP_OK;
}

void pep_d2h_finish(pep_ctx *ctx)
{
    for (int k = 0; k < ctx->n_down; ++k) memcpy(ctx->down[k].dst, ctx->pin_down.p + ctx->down[k].off, ctx->down[k].n);
    ctx->n_down = 0;
    ctx->pin_down_used = 0;
}

void pep_timer_begin(pep_ctx *ctx, int id)
{
    if (ctx->timing_level < (id == TM_SW ? 1 : 2)) return;        // (an event between two kernels costs about 6 us of idle GPU: pep_set_timing)
    if (!ctx->tm_a[id] && (hipEventCreate(&ctx->tm_a[id]) != hipSuccess || hipEventCreate(&ctx->tm_b[id]) != hipSuccess)) { ctx->tm_a[id] = nullptr; return; }
    ctx->tm_state[id] = hipEventRecord(ctx->tm_a[id], ctx->stream) == hipSuccess ? 1 : 0;
}

void pep_timer_end(pep_ctx *ctx, int id)
{
    if (ctx->tm_state[id] == 1) ctx->tm_state[id] = hipEventRecord(ctx->tm_b[id], ctx->stream) == hipSuccess ? 2 : 0;
}

void pep_timers_resolve(pep_ctx *ctx)
{
    double match[4] = {0., 0., 0., 0.};
    double *dst[TM_COUNT] = {&ctx->stats.ms_seed, &ctx->stats.ms_total, &ctx->stats.ms_sw, &ctx->stats.ms_sw_trace, &ctx->stats.ms_trace,
                             &match[0], &match[1], &match[2], &match[3]};
    for (int id = 0; id < TM_COUNT; ++id) {
        float ms = 0.f;
        if (ctx->tm_state[id] == 2 && hipEventSynchronize(ctx->tm_b[id]) == hipSuccess && hipEventElapsedTime(&ms, ctx->tm_a[id], ctx->tm_b[id]) == hipSuccess) *dst[id] = ms;
        ctx->tm_state[id] = 0;
    }
    ctx->stats.ms_seed_match = match[0] + match[1] + match[2] + match[3];
}

int pep_zero_block(pep_ctx *ctx, int which, size_t offset, size_t bytes, void **out)
{
    PEP_TRY(dev_reserve(ctx, ctx->d_zero, PEP_ZERO_TOTAL));
    char *p = ctx->d_zero.as<char>() + offset;
    if (!ctx->zero_ok[which]) PEP_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
    ctx->zero_ok[which] = false;
    *out = p;
    return PEP_OK;
}

int pep_read_back(pep_ctx *ctx, void *dst, const void *d_src, size_t n)
{
    const size_t n8 = (n + 7) & ~size_t(7);
    if (ctx->n_pending >= 32 || ctx->pin_small_used + n8 > ctx->pin_small.cap) PEP_TRY(pep_sync_reads(ctx));
    if (n8 > ctx->pin_small.cap) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_read_back: value larger than the pinned page");
    PEP_HIP(ctx, hipMemcpyAsync(ctx->pin_small.p + ctx->pin_small_used, d_src, n, hipMemcpyDeviceToHost, ctx->stream));
    ctx->pending[ctx->n_pending++] = pep_ctx::PendingRead{dst, ctx->pin_small_used, n};
    ctx->pin_small_used += n8;
    return PEP_OK;
}

int pep_read_back_with_upload(pep_ctx *ctx, void *dst, const void *d_src, size_t n, void *d_up_dst, const void *pinned_up_src, uint64_t n_up_words)
{
    const size_t n8 = (n + 7) & ~size_t(7);
    if (n % 4) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_read_back_with_upload: size not a multiple of 4");
    if (ctx->n_pending >= 32 || ctx->pin_small_used + n8 > ctx->pin_small.cap) PEP_TRY(pep_sync_reads(ctx));
    if (n8 > ctx->pin_small.cap) return pep_fail(ctx, PEP_ERR_INTERNAL, "pep_read_back: value larger than the pinned page");
    PEP_TRY(pep_exchange_pinned(ctx, d_up_dst, pinned_up_src, n_up_words, ctx->pin_small.p + ctx->pin_small_used, d_src, n / 4));
    ctx->pending[ctx->n_pending++] = pep_ctx::PendingRead{dst, ctx->pin_small_used, n};
    ctx->pin_small_used += n8;
    return PEP_OK;
}

// PEPPAN_HIP_SPIN_US: how long a wait for the GPU polls before it gives the CPU up (default 1.5 ms: a search's waits are shorter than a sleep's wake-up).
// 0 = never spin: the wait asks the event and naps (20 us growing to 200 us) in between - what the mapping path's worker processes use, which share a
// GPU and a machine's CPU allowance (peppan_amd/mapworkers.py).  Naps, not blocking events: a wait that depends on an interrupt was seen to turn a
// 2 s mapping into 218 s on one box of the pool (profiles/r04_map_pool_rate.txt), and asking costs a few per cent of a CPU.
static long pep_spin_us()
{
    static const long spin_us = [] { const char *e = getenv("PEPPAN_HIP_SPIN_US"); return e ? atol(e) : 1500L; }();
    return spin_us;
}

unsigned pep_wait_event_flags() { return hipEventDisableTiming; }

hipError_t pep_event_wait(hipEvent_t ev)
{
    const long spin_us = pep_spin_us();
    if (spin_us <= 0) {
        // (most waits of the small per-genome calls end within a few tens of microseconds: asked for without a nap first)
        const auto t0 = std::chrono::steady_clock::now();
        do {
            for (int spin = 0; spin < 32; ++spin) {
                const hipError_t q = hipEventQuery(ev);
                if (q != hipErrorNotReady) return q;
                __builtin_ia32_pause();
            }
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(60));
        long nap_ns = 20000;
        for (;;) {
            const hipError_t q = hipEventQuery(ev);
            if (q != hipErrorNotReady) return q;
            struct timespec ts = {0, nap_ns};
            nanosleep(&ts, nullptr);
            if (nap_ns < 200000) nap_ns += nap_ns / 2;
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int spin = 0; spin < 64; ++spin) {
            const hipError_t q = hipEventQuery(ev);
            if (q != hipErrorNotReady) return q;
            __builtin_ia32_pause();
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;      // long waits sleep
    }
    return hipEventSynchronize(ev);
}

hipError_t pep_stream_wait(pep_ctx *ctx)
{
    if (!ctx->wait_event && hipEventCreateWithFlags(&ctx->wait_event, pep_wait_event_flags()) != hipSuccess) return hipStreamSynchronize(ctx->stream);
    const hipError_t r = hipEventRecord(ctx->wait_event, ctx->stream);
    if (r != hipSuccess) return hipStreamSynchronize(ctx->stream);
    return pep_event_wait(ctx->wait_event);
}

int pep_sync_reads(pep_ctx *ctx)
{
    const hipError_t e = pep_stream_wait(ctx);
    if (e == hipSuccess)
        for (int i = 0; i < ctx->n_pending; ++i) memcpy(ctx->pending[i].dst, ctx->pin_small.p + ctx->pending[i].off, ctx->pending[i].n);
    ctx->n_pending = 0;
    ctx->pin_small_used = 0;
    if (e != hipSuccess) return pep_fail(ctx, PEP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return PEP_OK;
}

// the device copy of a result's table is about to be overwritten (or its context to go away)
void pep_drop_dev_result(pep_ctx *ctx)
{
    pep_result *r = ctx->dev_result;
    ctx->dev_result = nullptr;
    if (r && ctx->staged_result != r) r->ctx = nullptr;      // nothing ties it to the context any more
}

// the hit table of the newest search lives in the context's pinned staging area until it is copied out; before that area is
// reused (next search, K1) a result that is still alive takes its own copy
void pep_materialise_staged(pep_ctx *ctx)
{
    pep_result *r = ctx->staged_result;
    if (!r) return;
    r->hits.assign(r->st_hits, r->st_hits + r->n_hits);
    r->cigar.assign(r->st_cigar, r->st_cigar + r->n_cigar);
    r->st_hits = nullptr; r->st_cigar = nullptr;
    if (ctx->dev_result != r) r->ctx = nullptr;      // a result that owns its table and has no device copy needs the context no more (it may outlive it)
    ctx->staged_result = nullptr;
}

// block -> (sequence, start of the sequence) map of a packed set (host build from the offsets, then upload)
int pep_upload_blk2seq(pep_ctx *ctx, SeqSet &s)
{
    const uint64_t nblk = s.total / 32 + 2;
    std::vector<uint2> m(nblk, make_uint2(0u, 0u));
    for (uint32_t i = 0; i < s.n; ++i) {
        if (s.h_len[i] == 0) continue;
        const uint64_t b0 = s.h_off[i] / 32, b1 = ((uint64_t)s.h_off[i] + s.h_len[i] - 1) / 32;          // the 32-byte blocks that hold residues of sequence i
        for (uint64_t b = b0; b <= b1 && b < nblk; ++b) m[b] = make_uint2(i, s.h_off[i]);
    }
    PEP_TRY(dev_reserve(ctx, s.blk2seq, nblk * sizeof(uint2)));
    PEP_HIP(ctx, hipMemcpy(s.blk2seq.p, m.data(), nblk * sizeof(uint2), hipMemcpyHostToDevice));
    return PEP_OK;
}

namespace {

// BLOSUM62 in the NCBI text layout (public domain); parsed once into the 32x32 code-indexed table
const char *kBlosum62Text =
    "   A  R  N  D  C  Q  E  G  H  I  L  K  M  F  P  S  T  W  Y  V  B  Z  X  *\n"
    "A  4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0 -2 -1  0 -4\n"
    "R -1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3 -1  0 -1 -4\n"
    "N -2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3  3  0 -1 -4\n"
    "D -2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3  4  1 -1 -4\n"
    "C  0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1 -3 -3 -2 -4\n"
    "Q -1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2  0  3 -1 -4\n"
    "E -1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4\n"
    "G  0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3 -1 -2 -1 -4\n"
    "H -2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3  0  0 -1 -4\n"
    "I -1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3 -3 -3 -1 -4\n"
    "L -1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1 -4 -3 -1 -4\n"
    "K -1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2  0  1 -1 -4\n"
    "M -1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1 -3 -1 -1 -4\n"
    "F -2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1 -3 -3 -1 -4\n"
    "P -1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2 -2 -1 -2 -4\n"
    "S  1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2  0  0  0 -4\n"
    "T  0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0 -1 -1  0 -4\n"
    "W -3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3 -4 -3 -2 -4\n"
    "Y -2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1 -3 -2 -1 -4\n"
    "V  0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4 -3 -2 -1 -4\n"
    "B -2 -1  3  4 -3  0  1 -1  0 -3 -4  0 -3 -3 -2  0 -1 -4 -3 -3  4  1 -1 -4\n"
    "Z -1  0  0  1 -3  3  4 -2  0 -3 -3  1 -1 -3 -1  0 -1 -3 -2 -2  1  4 -1 -4\n"
    "X  0 -1 -1 -1 -2 -1 -1 -1 -1 -1 -1 -1 -1 -1 -2  0  0 -2 -1 -1 -1 -1 -1 -4\n"
    "* -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4 -4  1\n";

void parse_blosum(int8_t sub[1024])
{
    char cols[32];
    int ncol = 0;
    const char *p = kBlosum62Text;
    while (*p != '\n') { if (*p != ' ') cols[ncol++] = *p; ++p; }
    ++p;
    int m[26][26];
    bool have[26] = {false};
    for (int r = 0; r < ncol; ++r) {
        while (*p == ' ') ++p;
        const char row = *p++;
        for (int c = 0; c < ncol; ++c) {
            char *end;
            const long v = strtol(p, &end, 10);
            p = end;
            if (row >= 'A' && row <= 'Z' && cols[c] >= 'A' && cols[c] <= 'Z') { m[row - 'A'][cols[c] - 'A'] = (int)v; have[row - 'A'] = true; }
        }
        while (*p && *p != '\n') ++p;
        if (*p) ++p;
    }
    const int X = 'X' - 'A';
    for (int a = 0; a < 32; ++a)
        for (int b = 0; b < 32; ++b) {
            int v = -64;                                   // padding codes kill every path
            if (a < 26 && b < 26) v = m[have[a] ? a : X][have[b] ? b : X];    // J, O, U score as X
            sub[a * 32 + b] = (int8_t)v;
        }
}

// image of the substitution table as the SW kernel keeps it in LDS: dword (q*8 + t/4)*PEP_TAB_REP + copy
int upload_sub_image(pep_ctx *ctx)
{
    std::vector<uint32_t> img(256 * PEP_TAB_REP);
    const int8_t *s = ctx->params.sub;
    for (int w = 0; w < 256; ++w) {
        const int q = w >> 3, t0 = (w & 7) * 4;
        uint32_t v = 0;
        for (int k = 0; k < 4; ++k) v |= (uint32_t)(uint8_t)s[q * 32 + t0 + k] << (8 * k);
        for (int b = 0; b < PEP_TAB_REP; ++b) img[w * PEP_TAB_REP + b] = v;
    }
    PEP_TRY(dev_reserve(ctx, ctx->sub_lds, img.size() * 4));
    PEP_HIP(ctx, hipMemcpy(ctx->sub_lds.p, img.data(), img.size() * 4, hipMemcpyHostToDevice));
    return PEP_OK;
}

int upload_nt(pep_ctx *ctx, NtSet &s, const uint8_t *nt, const uint64_t *off, uint32_t n)
{
    if (n && (!nt || !off)) return pep_fail(ctx, PEP_ERR_ARG, "null sequence buffer");
    s.n = n;
    s.h_off.assign(n + 1, 0);
    for (uint32_t i = 0; i <= n; ++i) {
        s.h_off[i] = n ? off[i] : 0;
        if (i && s.h_off[i] < s.h_off[i - 1]) return pep_fail(ctx, PEP_ERR_ARG, "offsets must be non-decreasing");
        if (i && s.h_off[i] - s.h_off[i - 1] > 0x7FFFFF00ull) return pep_fail(ctx, PEP_ERR_LIMIT, "nucleotide sequence longer than 2^31 - 256");   // K1 indexes a sequence with 32 bits
    }
    s.total = s.h_off[n];
    PEP_TRY(dev_reserve(ctx, s.nt, s.total + 64));
    PEP_TRY(dev_reserve(ctx, s.off, (size_t)(n + 1) * 8));
    if (s.total) PEP_TRY(pep_h2d(ctx, s.nt.p, nt, s.total));                  // (stream-ordered: K1, K7 and the nucleotide tool's packing are queued behind it)
    PEP_TRY(pep_h2d(ctx, s.off.p, s.h_off.data(), (size_t)(n + 1) * 8));
    return PEP_OK;
}

int upload_aa(pep_ctx *ctx, SeqSet &s, const uint8_t *codes, const uint64_t *off, uint32_t n, uint32_t max_n)
{
    if (n && (!codes || !off)) return pep_fail(ctx, PEP_ERR_ARG, "null sequence buffer");
    if (n > max_n) return pep_fail(ctx, PEP_ERR_LIMIT, "too many sequences");
    if (&s == &ctx->t) ctx->t_tables_lazy = false;          // the host tables are built right here
    if (&s == &ctx->q) ctx->q_tables_lazy = false;
    s.n = n;
    s.h_off.assign(n + 1, 0);
    s.h_len.assign(n, 0);
    uint64_t pos = PEP_END_PAD, residues = 0;
    uint32_t max_len = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (off[i + 1] < off[i]) return pep_fail(ctx, PEP_ERR_ARG, "offsets must be non-decreasing");
        const uint64_t len = off[i + 1] - off[i];
        if (len > PEP_MAX_SEQ_LEN) return pep_fail(ctx, PEP_ERR_LIMIT, "sequence longer than PEP_MAX_SEQ_LEN");
        s.h_off[i] = (uint32_t)pos;
        s.h_len[i] = (uint32_t)len;
        residues += len;
        max_len = std::max<uint32_t>(max_len, (uint32_t)len);
        pos += (len + 15) / 16 * 16 + PEP_SEQ_GAP;
        if (pos > PEP_MAX_RESIDUES) return pep_fail(ctx, PEP_ERR_LIMIT, "packed protein set exceeds 2^29 bytes");
    }
    pos += PEP_END_PAD;
    s.h_off[n] = (uint32_t)pos;
    s.total = pos; s.residues = residues; s.max_len = max_len;
    std::vector<uint8_t> img(pos, (uint8_t)PEP_PAD_CODE);
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t x = 0; x < s.h_len[i]; ++x) {
            const uint8_t c = codes[off[i] + x];
            img[s.h_off[i] + x] = c < 26 ? c : (uint8_t)23;     // anything that is not a letter is X
        }
    PEP_TRY(dev_reserve(ctx, s.res, pos + 64));
    PEP_TRY(dev_reserve(ctx, s.off, (size_t)(n + 1) * 4));
    PEP_TRY(dev_reserve(ctx, s.len, (size_t)(n + 1) * 4));
    PEP_HIP(ctx, hipMemcpy(s.res.p, img.data(), pos, hipMemcpyHostToDevice));
    PEP_HIP(ctx, hipMemcpy(s.off.p, s.h_off.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    if (n) PEP_HIP(ctx, hipMemcpy(s.len.p, s.h_len.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    return pep_upload_blk2seq(ctx, s);
}

int download_aa(pep_ctx *ctx, const SeqSet &s, uint8_t *codes, uint64_t cap, uint64_t *off)
{
    if (s.residues > cap) return pep_fail(ctx, PEP_ERR_ARG, "output buffer too small");
    PEP_TRY(pep_k1_host_tables(ctx));
    std::vector<uint8_t> img(s.total);
    if (s.total) PEP_HIP(ctx, hipMemcpy(img.data(), s.res.p, s.total, hipMemcpyDeviceToHost));
    uint64_t pos = 0;
    for (uint32_t i = 0; i < s.n; ++i) {
        off[i] = pos;
        memcpy(codes + pos, img.data() + s.h_off[i], s.h_len[i]);
        pos += s.h_len[i];
    }
    off[s.n] = pos;
    return PEP_OK;
}

}  // namespace

// entry points for other translation units (K9's gapped verification drives the alignment engine on its own sequence sets)
int pep_upload_codes(pep_ctx *ctx, SeqSet &s, const uint8_t *codes, const uint64_t *off, uint32_t n, uint32_t max_n) { return upload_aa(ctx, s, codes, off, n, max_n); }
int pep_upload_sub(pep_ctx *ctx) { return upload_sub_image(ctx); }

extern "C" {

int pep_version(void) { return PEP_ABI_VERSION; }

int pep_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pep_default_params(pep_search_params *p)
{
    memset(p, 0, sizeof(*p));
    p->gap_open = 11; p->gap_ext = 1;
    parse_blosum(p->sub);
    memset(p->reduce, 0xFF, sizeof(p->reduce));
    // reduced alphabet of Buchfink, Xie & Huson 2015: [KREDQN] [C] [G] [H] [ILV] [M] [F] [Y] [W] [P] [STA]
    const char *groups = "KREDQN/C/G/H/ILV/M/F/Y/W/P/STA";
    int g = 0;
    for (const char *c = groups; *c; ++c) { if (*c == '/') ++g; else p->reduce[*c - 'A'] = (uint8_t)g; }
    p->base = g + 1;
    const char *shapes[2] = {"111101110111", "111011010010111"};
    p->n_shapes = 2;
    for (int s = 0; s < 2; ++s) {
        int w = 0;
        for (int k = 0; shapes[s][k]; ++k) if (shapes[s][k] == '1') p->offs[s][w++] = k;
        p->weight[s] = w;
    }
    p->min_id_pct = 0.; p->min_qcov_pct = 0.; p->top_k = 10; p->n_splits = 5;
    p->dbsize = 5e6; p->max_evalue = 1.;
    p->use_lds = 1;
    p->ungapped_min = 55; p->xdrop = 12; p->ext_right = 40; p->ext_left = 24; p->stage1_min = 24; p->reserved2 = 0;
    p->ka_lambda = 0.267; p->ka_k = 0.041;
}

// Sensitivity of the translated search = the set of spaced seed shapes (all weight 10 over the 11-letter alphabet).
//   0  DIAMOND's two default-sensitivity shapes - what `diamond blastp` runs with on the reference's command line (uberBlast.py:550)
//   1  four shapes: the two above plus 110010011111011 and 10111110011011 - recall between 0.45 and 0.7 identity 0.93 -> 0.985 against
//      exhaustive Smith-Waterman at about twice the seed-stage cost
int pep_set_sensitivity(pep_search_params *p, int level)
{
    if (!p || level < 0 || level > 1) return PEP_ERR_ARG;
    const char *shapes[4] = {"111101110111", "111011010010111", "110010011111011", "10111110011011"};
    p->n_shapes = level == 0 ? 2 : 4;
    for (int s = 0; s < 4; ++s) {
        int w = 0;
        for (int k = 0; k < 32; ++k) p->offs[s][k] = 0;
        if (s < p->n_shapes) for (int k = 0; shapes[s][k]; ++k) if (shapes[s][k] == '1') p->offs[s][w++] = k;
        p->weight[s] = w;
    }
    return PEP_OK;
}

int32_t pep_min_score_ka(uint32_t qlen, double dbsize, double max_evalue, double ka_lambda, double ka_k)
{
    // E = K m n exp(-lambda S)  ->  smallest integer S with E <= max_evalue
    const double s = log(ka_k * (double)qlen * dbsize / max_evalue) / ka_lambda;
    const int32_t r = (int32_t)ceil(s);
    return r < 1 ? 1 : r;
}

int32_t pep_min_score(uint32_t qlen, double dbsize, double max_evalue)
{
    return pep_min_score_ka(qlen, dbsize, max_evalue, 0.267, 0.041);      // gapped BLOSUM62 11/1
}

int pep_ctx_create(int device, pep_ctx **out)
{
    if (!out) return PEP_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return PEP_ERR_HIP;
    pep_ctx *ctx = new (std::nothrow) pep_ctx();
    if (!ctx) return PEP_ERR_INTERNAL;
    ctx->device = device;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    pep_default_params(&ctx->params);
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return PEP_ERR_HIP; }
    if (hipStreamCreate(&ctx->stream) != hipSuccess) { delete ctx; return PEP_ERR_HIP; }
    if (pin_reserve(ctx, ctx->pin_small, 16384) != PEP_OK) { *out = ctx; return PEP_ERR_HIP; }
    int rc = pep_selftest_dpp(ctx);
    if (rc != PEP_OK) { *out = ctx; return rc; }      // caller can read the message, then destroy
    *out = ctx;
    return PEP_OK;
}

void pep_ctx_destroy(pep_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto &b : ctx->ws) dev_release(b);
    for (int id = 0; id < TM_COUNT; ++id) { if (ctx->tm_a[id]) (void)hipEventDestroy(ctx->tm_a[id]); if (ctx->tm_b[id]) (void)hipEventDestroy(ctx->tm_b[id]); }
    pep_drop_dev_result(ctx);
    if (ctx->staged_result) {
        // the result outlives the context (freeing it afterwards is allowed): it takes its own copy and forgets the context
        pep_materialise_staged(ctx);
    }
    if (ctx->pin_small.p) (void)hipHostFree(ctx->pin_small.p);
    if (ctx->pin_stage.p) (void)hipHostFree(ctx->pin_stage.p);
    if (ctx->pin_k1.p) (void)hipHostFree(ctx->pin_k1.p);
    if (ctx->pin_k1q.p) (void)hipHostFree(ctx->pin_k1q.p);
    if (ctx->pin_labels.p) (void)hipHostFree(ctx->pin_labels.p);
    if (ctx->pin_nt_match.p) (void)hipHostFree(ctx->pin_nt_match.p);
    if (ctx->k1_event) (void)hipEventDestroy(ctx->k1_event);
    if (ctx->k1q_event) (void)hipEventDestroy(ctx->k1q_event);
    if (ctx->k1_t0) (void)hipEventDestroy(ctx->k1_t0);
    if (ctx->k1_t1) (void)hipEventDestroy(ctx->k1_t1);
    if (ctx->wait_event) (void)hipEventDestroy(ctx->wait_event);
    if (ctx->up_event) (void)hipEventDestroy(ctx->up_event);
    if (ctx->pin_down.p) (void)hipHostFree(ctx->pin_down.p);
    if (ctx->pin_up.p) (void)hipHostFree(ctx->pin_up.p);
    if (ctx->pin_ms.p) (void)hipHostFree(ctx->pin_ms.p);
    DevBuf *bufs[] = {&ctx->sub_lds, &ctx->d_params, &ctx->scan_state[0].buf, &ctx->scan_state[1].buf, &ctx->fused_state[0].buf, &ctx->fused_state[1].buf, &ctx->fused_state[2].buf, &ctx->fused_state[3].buf, &ctx->d_min_score, &ctx->d_trace_mode, &ctx->d_trace_defer, &ctx->d_k1_base, &ctx->d_k1_seg, &ctx->d_k1_long, &ctx->d_k1_spec, &ctx->d_k1_tiles, &ctx->d_t_class, &ctx->d_t_subject, &ctx->d_nt_match, &ctx->q_nt.nt, &ctx->q_nt.off, &ctx->r_nt.nt, &ctx->r_nt.off,
                      &ctx->q.res, &ctx->q.off, &ctx->q.len, &ctx->t.res, &ctx->t.off, &ctx->t.len, &ctx->q.blk2seq, &ctx->t.blk2seq,
                      &ctx->sort_state, &ctx->sort_hist, &ctx->d_set, &ctx->d_zero, &ctx->d_k1_desc_q, &ctx->d_k1_desc_t, &ctx->d_self_delta, &ctx->d_self_t, &ctx->d_mail_copy,
                      &ctx->nucl_q.d_off, &ctx->nucl_q.d_len, &ctx->nucl_q.d_desc, &ctx->nucl_t.d_off, &ctx->nucl_t.d_len, &ctx->nucl_t.d_desc, &ctx->nucl_t.d_first};
    for (DevBuf *b : bufs) dev_release(*b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pep_last_error(const pep_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pep_set_query_nt(pep_ctx *ctx, const uint8_t *nt, const uint64_t *off, uint32_t n, int gtable)
{
    if (!ctx) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    if (n > PEP_MAX_QUERIES) return pep_fail(ctx, PEP_ERR_LIMIT, "too many queries");
    PEP_TRY(upload_nt(ctx, ctx->q_nt, nt, off, n));
    ctx->nucl_valid = false;
    ctx->q_from_nt = true; ctx->q_gtable = gtable; ctx->q_ready = false; ctx->resid_from_nucl = false;
    return PEP_OK;
}

int pep_set_ref_nt(pep_ctx *ctx, const uint8_t *nt, const uint64_t *off, uint32_t n, int frames, int gtable)
{
    if (!ctx) return PEP_ERR_ARG;
    if (frames != 3 && frames != 6) return pep_fail(ctx, PEP_ERR_ARG, "frames must be 3 or 6");
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    PEP_TRY(upload_nt(ctx, ctx->r_nt, nt, off, n));
    ctx->nucl_valid = false;
    ctx->t_from_nt = true; ctx->t_gtable = gtable; ctx->t_frames = frames; ctx->t_ready = false; ctx->k1_base_frames = 0; ctx->resid_from_nucl = false;
    ctx->group_of_seq.clear(); ctx->t_class_ready = false;
    return PEP_OK;
}

int pep_set_query_aa(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n)
{
    if (!ctx) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    PEP_TRY(upload_aa(ctx, ctx->q, codes, off, n, PEP_MAX_QUERIES));
    ctx->q_meta.resize(n);
    for (uint32_t i = 0; i < n; ++i) ctx->q_meta[i] = pep_query_meta{i, 0u, ctx->q.h_len[i], 0u};
    ctx->q_from_nt = false; ctx->q_ready = true; ctx->resid_from_nucl = false;
    return PEP_OK;
}

int pep_set_ref_aa(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n)
{
    if (!ctx) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    PEP_TRY(upload_aa(ctx, ctx->t, codes, off, n, PEP_MAX_TARGETS));
    ctx->t_meta.resize(n);
    for (uint32_t i = 0; i < n; ++i) ctx->t_meta[i] = pep_target_meta{i, 0u, 0u, ctx->t.h_len[i]};
    ctx->t_from_nt = false; ctx->t_ready = true; ctx->resid_from_nucl = false;
    ctx->group_of_seq.clear(); ctx->t_class_ready = false;
    return PEP_OK;
}

int pep_translate(pep_ctx *ctx, int force)
{
    if (!ctx) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    // (the two timing events live as long as the context)
    if (!ctx->k1_t0 && (hipEventCreate(&ctx->k1_t0) != hipSuccess || hipEventCreate(&ctx->k1_t1) != hipSuccess)) { ctx->k1_t0 = ctx->k1_t1 = nullptr; }
    // both sides are queued first (reference, then queries); the reference's summary is taken while the query kernels run
    ctx->resid_from_nucl = false;           // (pep_use_nt_as_residues left q_ready / t_ready false: K1 runs again)
    const bool do_q = ctx->q_from_nt && (force || !ctx->q_ready), do_t = ctx->t_from_nt && (force || !ctx->t_ready);
    if (!do_q && !do_t) return PEP_OK;       // nothing to translate (pep_search calls this every time): no events, no waiting
    const bool timed = ctx->timing_level >= 2 && ctx->k1_t0;
    if (timed) (void)hipEventRecord(ctx->k1_t0, ctx->stream);
    if (do_t) PEP_TRY(pep_k1_ref(ctx, ctx->t_frames, ctx->t_gtable, 1));
    if (do_q) PEP_TRY(pep_k1_query(ctx, ctx->q_gtable, 1));
    if (do_t) { PEP_TRY(pep_k1_ref(ctx, ctx->t_frames, ctx->t_gtable, 2)); ctx->t_ready = true; }
    if (do_q) { PEP_TRY(pep_k1_query(ctx, ctx->q_gtable, 2)); ctx->q_ready = true; }
    float ms = 0.f;
    if (timed && ctx->k1_t1 && hipEventRecord(ctx->k1_t1, ctx->stream) == hipSuccess && pep_event_wait(ctx->k1_t1) == hipSuccess) (void)hipEventElapsedTime(&ms, ctx->k1_t0, ctx->k1_t1);
    ctx->stats.ms_k1 = ms;
    return PEP_OK;
}

int pep_invalidate_translation(pep_ctx *ctx)
{
    if (!ctx) return PEP_ERR_ARG;
    if (ctx->q_from_nt) ctx->q_ready = false;
    if (ctx->t_from_nt) ctx->t_ready = false;
    return PEP_OK;
}

int pep_use_nt_as_residues(pep_ctx *ctx, int strands)
{
    if (!ctx) return PEP_ERR_ARG;
    if (strands != 1 && strands != 2) return pep_fail(ctx, PEP_ERR_ARG, "strands must be 1 or 2");
    if (!ctx->q_from_nt || !ctx->t_from_nt) return pep_fail(ctx, PEP_ERR_STATE, "pep_use_nt_as_residues needs pep_set_query_nt and pep_set_ref_nt first");
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    pep_materialise_staged(ctx);
    ctx->q_ready = ctx->t_ready = false;      // the packed protein sets are gone: a later pep_translate runs K1 again
    ctx->resid_from_nucl = false;
    ctx->t_class_ready = false;
    PEP_TRY(pep_nucl_sets(ctx, strands));
    ctx->resid_from_nucl = true;
    return PEP_OK;
}

int pep_query_count(pep_ctx *ctx, uint32_t *n, uint64_t *residues)
{
    if (!ctx || !(ctx->q_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "queries not set / not translated");
    if (n) *n = ctx->q.n;
    if (residues) *residues = ctx->q.residues;
    return PEP_OK;
}

int pep_target_count(pep_ctx *ctx, uint32_t *n, uint64_t *residues)
{
    if (!ctx || !(ctx->t_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "targets not set / not translated");
    if (n) *n = ctx->t.n;
    if (residues) *residues = ctx->t.residues;
    return PEP_OK;
}

int pep_get_query_meta(pep_ctx *ctx, pep_query_meta *out, uint32_t cap)
{
    if (!ctx || !(ctx->q_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "queries not set / not translated");
    PEP_TRY(pep_k1_host_tables_q(ctx));
    if (cap < ctx->q_meta.size()) return pep_fail(ctx, PEP_ERR_ARG, "output buffer too small");
    if (!ctx->q_meta.empty()) memcpy(out, ctx->q_meta.data(), ctx->q_meta.size() * sizeof(pep_query_meta));
    return PEP_OK;
}

int pep_get_target_meta(pep_ctx *ctx, pep_target_meta *out, uint32_t cap)
{
    if (!ctx || !(ctx->t_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "targets not set / not translated");
    PEP_TRY(pep_k1_host_tables(ctx));
    if (cap < ctx->t_meta.size()) return pep_fail(ctx, PEP_ERR_ARG, "output buffer too small");
    if (!ctx->t_meta.empty()) memcpy(out, ctx->t_meta.data(), ctx->t_meta.size() * sizeof(pep_target_meta));
    return PEP_OK;
}

int pep_get_query_aa(pep_ctx *ctx, uint8_t *codes, uint64_t cap, uint64_t *off)
{
    if (!ctx || !(ctx->q_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "queries not set / not translated");
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return download_aa(ctx, ctx->q, codes, cap, off);
}

int pep_get_target_aa(pep_ctx *ctx, uint8_t *codes, uint64_t cap, uint64_t *off)
{
    if (!ctx || !(ctx->t_ready || ctx->resid_from_nucl)) return pep_fail(ctx, PEP_ERR_STATE, "targets not set / not translated");
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return download_aa(ctx, ctx->t, codes, cap, off);
}

int pep_set_target_groups(pep_ctx *ctx, const uint32_t *group, uint32_t n)
{
    if (!ctx || (n && !group)) return PEP_ERR_ARG;
    for (uint32_t i = 1; i < n; ++i)
        if (group[i] < group[i - 1]) return pep_fail(ctx, PEP_ERR_ARG, "pep_set_target_groups: groups must be contiguous and non-decreasing");
    if (ctx->group_of_seq.size() == n && (n == 0 || memcmp(ctx->group_of_seq.data(), group, (size_t)n * sizeof(uint32_t)) == 0)) return PEP_OK;      // (nothing changes: what was derived from the groups stays)
    ctx->group_of_seq.assign(group, group + n);
    ctx->t_class_ready = false;
    ctx->nucl_valid = false;                    // (the nucleotide tool's targets are laid out group by group)
    return PEP_OK;
}

// competition class per target from the group of its reference sequence (host build, small)
static int build_t_class(pep_ctx *ctx)
{
    ctx->t_class_ready = false;
    if (ctx->group_of_seq.empty()) return PEP_OK;
    const uint32_t n_seq = ctx->t_from_nt ? ctx->r_nt.n : ctx->t.n;
    if (ctx->group_of_seq.size() != n_seq) return pep_fail(ctx, PEP_ERR_ARG, "pep_set_target_groups: one group per reference sequence expected");
    PEP_TRY(pep_k1_host_tables(ctx));
    const uint32_t nt = ctx->t.n, ns = (uint32_t)ctx->params.n_splits;
    std::vector<uint32_t> cls(nt + 1, 0u);
    uint32_t cur = 0xFFFFFFFFu, local = 0;
    for (uint32_t t = 0; t < nt; ++t) {
        const uint32_t g = ctx->group_of_seq[ctx->t_meta[t].seq];
        if (g != cur) { cur = g; local = 0; }
        if ((uint64_t)g * ns + ns > 0xFFFFFFFFull) return pep_fail(ctx, PEP_ERR_LIMIT, "too many target groups");
        cls[t] = g * ns + local % ns;
        ++local;
    }
    PEP_TRY(dev_reserve(ctx, ctx->d_t_class, (size_t)(nt + 1) * 4));
    PEP_TRY(pep_h2d(ctx, ctx->d_t_class.p, cls.data(), (size_t)(nt + 1) * 4));
    ctx->t_class_ready = true;
    return PEP_OK;
}

int pep_search(pep_ctx *ctx, const pep_search_params *params, pep_result **out)
{
    if (!ctx || !out) return PEP_ERR_ARG;
    *out = nullptr;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    pep_materialise_staged(ctx);                 // an earlier result that was not copied out yet keeps its own copy
    if (params) {
        if (params->n_shapes < 1 || params->n_shapes > 4 || params->base < 2 || params->top_k < 1 || params->n_splits < 1)
            return pep_fail(ctx, PEP_ERR_ARG, "invalid search parameters");
        // (the packed 16-bit passes keep H - open - extend inside a signed half word)
        if (params->gap_open < 0 || params->gap_open > 255 || params->gap_ext < 1 || params->gap_ext > 255)
            return pep_fail(ctx, PEP_ERR_ARG, "gap costs: 0 <= gap_open <= 255 and 1 <= gap_ext <= 255");
        for (int s = 0; s < params->n_shapes; ++s) {
            if (params->weight[s] < 1 || params->weight[s] > 32) return pep_fail(ctx, PEP_ERR_ARG, "invalid seed weight");
            if (params->offs[s][params->weight[s] - 1] > 31) return pep_fail(ctx, PEP_ERR_ARG, "seed span above 32");
            if (pow((double)params->base, params->weight[s]) > 34359738368.0) return pep_fail(ctx, PEP_ERR_ARG, "seed key does not fit 35 bits");
            if (params->base > 15 || pow((double)params->base, (params->weight[s] + 1) / 2) > 4294967295.0)
                return pep_fail(ctx, PEP_ERR_ARG, "reduced alphabet above 15 letters / half key above 32 bits");
            for (int c = 0; c < 32; ++c)
                if (params->reduce[c] != 0xFF && params->reduce[c] >= params->base) return pep_fail(ctx, PEP_ERR_ARG, "reduced letter outside the alphabet");
        }
        if (params->t_index_base < 0) return pep_fail(ctx, PEP_ERR_ARG, "t_index_base must not be negative");
        if (params->hsp_mode < 0 || params->hsp_mode > 2) return pep_fail(ctx, PEP_ERR_ARG, "hsp_mode must be 0, 1 or 2");
        if (params->stage1_min < 0 || (params->ungapped_min > 0 && params->stage1_min > params->ungapped_min))
            return pep_fail(ctx, PEP_ERR_ARG, "stage1_min must lie between 0 and ungapped_min");
        if (!(params->ka_lambda > 0.) || !(params->ka_k > 0.)) return pep_fail(ctx, PEP_ERR_ARG, "invalid Karlin-Altschul parameters");
        if (!(params->dbsize > 0.) || !(params->max_evalue > 0.)) return pep_fail(ctx, PEP_ERR_ARG, "dbsize and max_evalue must be positive");
        if (params->xdrop < 0 || params->xdrop > 48 || params->ext_right < 1 || params->ext_right > 48 || params->ext_left < 0 || params->ext_left > 48)
            return pep_fail(ctx, PEP_ERR_ARG, "invalid ungapped-extension parameters");
        // code 31 is the padding between packed sequences: its scores end every extension and keep the DP out of the padding
        for (int c = 0; c < 32; ++c)
            if (params->sub[PEP_PAD_CODE * 32 + c] > -64 || params->sub[c * 32 + PEP_PAD_CODE] > -64)
                return pep_fail(ctx, PEP_ERR_ARG, "substitution table: row and column 31 (the padding code) must be <= -64");
        if (memcmp(ctx->params.sub, params->sub, sizeof(params->sub)) != 0) ctx->sub_ready = false;      // the LDS image follows the table only
        ctx->params = *params;
    }
    ctx->k1_t_deferred = false;
    if (!ctx->resid_from_nucl) {
        const bool do_q = ctx->q_from_nt && !ctx->q_ready, do_t = ctx->t_from_nt && !ctx->t_ready;
        if ((do_q || do_t) && ctx->timing_level < 2 && ctx->group_of_seq.empty()) {
            // K1 inside the search (sets given or invalidated since the last one): both sides are queued, queries first, and only the query
            // side is waited for here - the reference side's summary is picked up after the first index build has been queued behind it
            // (pep_find_candidates' need_targets), so the GPU runs from K1 into the seed stage without waiting for the host
            if (do_q) PEP_TRY(pep_k1_query(ctx, ctx->q_gtable, 1));
            if (do_t) PEP_TRY(pep_k1_ref(ctx, ctx->t_frames, ctx->t_gtable, 1));
            if (do_q) { PEP_TRY(pep_k1_query(ctx, ctx->q_gtable, 2)); ctx->q_ready = true; }
            ctx->k1_t_deferred = do_t;
            ctx->stats.ms_k1 = 0.;
        } else PEP_TRY(pep_translate(ctx, 0));
        if (!ctx->q_ready || !(ctx->t_ready || ctx->k1_t_deferred)) { ctx->k1_t_deferred = false; return pep_fail(ctx, PEP_ERR_STATE, "pep_search before both sequence sets were given"); }
    }
    auto finish_targets = [](pep_ctx *c) -> int {
        if (!c->k1_t_deferred) return PEP_OK;
        c->k1_t_deferred = false;
        PEP_TRY(pep_k1_ref(c, c->t_frames, c->t_gtable, 2));
        c->t_ready = true;
        c->stats.target_residues = c->t.residues;
        return PEP_OK;
    };
    if (!ctx->sub_ready) { const int rc0 = upload_sub_image(ctx); if (rc0 != PEP_OK) { ctx->k1_t_deferred = false; return rc0; } ctx->sub_ready = true; }
    { const int rc0 = build_t_class(ctx); if (rc0 != PEP_OK) { ctx->k1_t_deferred = false; return rc0; } }

    pep_result *res = new (std::nothrow) pep_result();
    if (!res) { ctx->k1_t_deferred = false; return pep_fail(ctx, PEP_ERR_INTERNAL, "out of host memory"); }
    res->ctx = ctx;
    const double ms_k1 = ctx->stats.ms_k1;
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    ctx->stats.ms_k1 = ms_k1;
    ctx->stats.query_residues = ctx->q.residues;
    ctx->stats.target_residues = ctx->t.residues;

    for (int id = 0; id < TM_COUNT; ++id) ctx->tm_state[id] = 0;
    pep_timer_begin(ctx, TM_SEED); pep_timer_begin(ctx, TM_TOTAL);
    uint64_t *d_cands = nullptr, n_cands = 0;
    // host work that only the alignment stage needs - the score thresholds (from the query lengths K1 left in pinned memory) and their upload - is done while the
    // seed stage's kernels run (between its last launch and its synchronisation)
    auto prepare_thresholds = [](pep_ctx *c) -> int {
        PEP_TRY(dev_reserve(c, c->d_min_score, (size_t)(c->q.n + 1) * 4));
        PEP_TRY(pin_reserve(c, c->pin_ms, (size_t)(c->q.n + 1) * 4));
        int32_t *ms = reinterpret_cast<int32_t *>(c->pin_ms.p);
        // the threshold depends on the length only and lengths repeat: one logarithm per distinct length (direct-mapped memo)
        uint32_t memo_len[1024];
        int32_t memo_val[1024];
        for (int i = 0; i < 1024; ++i) memo_len[i] = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < c->q.n; ++i) {
            const uint32_t L = c->q.h_len[i], slot = L & 1023u;
            if (memo_len[slot] != L) {
                memo_len[slot] = L;
                memo_val[slot] = pep_min_score_ka(L, c->params.dbsize, c->params.max_evalue, c->params.ka_lambda, c->params.ka_k);
            }
            ms[i] = memo_val[slot];
        }
        c->upload.d_dst = c->d_min_score.p; c->upload.pinned_src = c->pin_ms.p; c->upload.n_words = c->q.n;      // (rides on the seed stage's read-back kernel)
        return PEP_OK;
    };
    int rc = pep_find_candidates(ctx, &d_cands, &n_cands, prepare_thresholds, finish_targets);
    if (rc == PEP_OK && ctx->k1_t_deferred) rc = finish_targets(ctx);
    ctx->k1_t_deferred = false;                   // (after a failure the reference side simply is not ready: the next search translates again)
    pep_timer_end(ctx, TM_SEED);
    if (rc == PEP_OK) rc = pep_extend(ctx, d_cands, n_cands, nullptr, res, true);
    const bool grouped = rc == PEP_OK && ctx->grp_nodes != 0;
    if (grouped)                                  // single linkage over the table that was just emitted, same stream, same wait
        rc = ctx->ext.pending ? pep_k10_queue(ctx, ctx->ext.n_bound, reinterpret_cast<const pep_hit *>(ctx->ext.d_hits), ctx->ext.d_n_hits)
                              : pep_k10_queue(ctx, res->n_hits, res->d_hits);
    // K7's count of identical nucleotide columns for every hit of this search (pep_set_nt_match), from the table on the device: same stream, same wait
    const bool counted = rc == PEP_OK && ctx->want_nt_match;
    if (counted)
        rc = ctx->ext.pending ? pep_k7_hits_queue(ctx, ctx->ext.n_bound, reinterpret_cast<const pep_hit *>(ctx->ext.d_hits), reinterpret_cast<const uint32_t *>(ctx->ext.d_cig), ctx->ext.d_n_hits)
                              : pep_k7_hits_queue(ctx, res->n_hits, res->d_hits, res->d_cigar);
    pep_timer_end(ctx, TM_TOTAL);
    const hipError_t se = pep_stream_wait(ctx);
    if (rc == PEP_OK && se != hipSuccess) rc = pep_fail(ctx, PEP_ERR_HIP, std::string("stream sync: ") + hipGetErrorString(se));
    if (rc == PEP_OK) rc = pep_extend_finish(ctx);      // (a result that left through pack_out: its sizes and statistics are in the staging area now)
    ctx->ext.pending = false;
    if (rc != PEP_OK) { delete res; return rc; }
    pep_timers_resolve(ctx);
    if (grouped) {
        const uint32_t *lab = reinterpret_cast<const uint32_t *>(ctx->pin_labels.p);
        res->labels.assign(lab, lab + ctx->grp_nodes);
    }
    if (counted && res->n_hits) {
        const uint32_t *cnt = reinterpret_cast<const uint32_t *>(ctx->pin_nt_match.p);
        res->nt_match.assign(cnt, cnt + res->n_hits);
    }
    res->stats = ctx->stats;
    ctx->dev_result = res->d_hits ? res : nullptr;
    if (res->st_hits || res->st_cigar) ctx->staged_result = res;
    else if (!res->d_hits) res->ctx = nullptr;             // owns its (possibly empty) table from the start: nothing ties it to the context
    *out = res;
    return PEP_OK;
}

int pep_set_nt_match(pep_ctx *ctx, int on)
{
    if (!ctx || (on != 0 && on != 1)) return PEP_ERR_ARG;
    ctx->want_nt_match = on != 0;
    return PEP_OK;
}

int pep_result_nt_match(const pep_result *r, const uint32_t **nt_match)
{
    if (!r || !nt_match) return PEP_ERR_ARG;
    *nt_match = (r->n_hits && r->nt_match.size() == r->n_hits) ? r->nt_match.data() : nullptr;
    return PEP_OK;
}

int pep_set_timing(pep_ctx *ctx, int level)
{
    if (!ctx || level < 0 || level > 2) return PEP_ERR_ARG;
    ctx->timing_level = level;
    return PEP_OK;
}

int pep_set_grouping(pep_ctx *ctx, uint32_t n_nodes, uint32_t q_base, const uint32_t *node_of_target, uint64_t n_targets)
{
    if (!ctx || (n_nodes && n_targets && !node_of_target)) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k10_set_grouping(ctx, n_nodes, q_base, node_of_target, n_targets);
}

int pep_result_labels(const pep_result *r, uint32_t *label, uint32_t n_nodes)
{
    if (!r || (n_nodes && !label)) return PEP_ERR_ARG;
    if (r->labels.size() != n_nodes) return PEP_ERR_STATE;          // the search ran without pep_set_grouping (or with another node count)
    if (n_nodes) memcpy(label, r->labels.data(), (size_t)n_nodes * 4);
    return PEP_OK;
}

int pep_set_result_mode(pep_ctx *ctx, int on_device)
{
    if (!ctx) return PEP_ERR_ARG;
    ctx->device_results = on_device != 0;
    return PEP_OK;
}

// the host copy of a result that was left on the device (pep_set_result_mode): fetched once, on demand
static int pep_fetch_result(const pep_result *cr)
{
    pep_result *r = const_cast<pep_result *>(cr);
    if (r->st_hits || r->n_hits == 0 || !r->hits.empty()) return PEP_OK;
    pep_ctx *ctx = r->ctx;
    if (!ctx || ctx->dev_result != r) return PEP_ERR_STATE;         // its device copy has been overwritten
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    r->hits.resize(r->n_hits);
    r->cigar.resize(r->n_cigar);
    PEP_HIP(ctx, hipMemcpyAsync(r->hits.data(), r->d_hits, r->n_hits * sizeof(pep_hit), hipMemcpyDeviceToHost, ctx->stream));
    if (r->n_cigar) PEP_HIP(ctx, hipMemcpyAsync(r->cigar.data(), r->d_cigar, r->n_cigar * 4, hipMemcpyDeviceToHost, ctx->stream));
    PEP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PEP_OK;
}

int pep_result_size(const pep_result *r, uint64_t *n_hits, uint64_t *n_cigar)
{
    if (!r) return PEP_ERR_ARG;
    if (n_hits) *n_hits = r->n_hits;
    if (n_cigar) *n_cigar = r->n_cigar;
    return PEP_OK;
}

int pep_result_copy(const pep_result *r, pep_hit *hits, uint32_t *cigar)
{
    if (!r) return PEP_ERR_ARG;
    PEP_TRY(pep_fetch_result(r));
    const pep_hit *h = r->st_hits ? r->st_hits : r->hits.data();
    const uint32_t *c = r->st_cigar ? r->st_cigar : r->cigar.data();
    if (hits && r->n_hits) memcpy(hits, h, r->n_hits * sizeof(pep_hit));
    if (cigar && r->n_cigar) memcpy(cigar, c, r->n_cigar * sizeof(uint32_t));
    return PEP_OK;
}

int pep_result_data(const pep_result *r, const pep_hit **hits, const uint32_t **cigar)
{
    if (!r) return PEP_ERR_ARG;
    PEP_TRY(pep_fetch_result(r));
    if (hits) *hits = r->st_hits ? r->st_hits : r->hits.data();
    if (cigar) *cigar = r->st_cigar ? r->st_cigar : r->cigar.data();
    return PEP_OK;
}

int pep_result_stats(const pep_result *r, pep_stats *stats)
{
    if (!r || !stats) return PEP_ERR_ARG;
    *stats = r->stats;
    return PEP_OK;
}

void pep_result_free(pep_result *r)
{
    if (r && r->ctx && r->ctx->staged_result == r) r->ctx->staged_result = nullptr;
    if (r && r->ctx && r->ctx->dev_result == r) r->ctx->dev_result = nullptr;
    delete r;
}

int pep_result_device(const pep_result *r, const pep_hit **d_hits, const uint32_t **d_cigar)
{
    if (!r) return PEP_ERR_ARG;
    const bool live = r->ctx && r->ctx->dev_result == r;
    if (d_hits) *d_hits = live ? r->d_hits : nullptr;
    if (d_cigar) *d_cigar = live ? r->d_cigar : nullptr;
    return live ? PEP_OK : PEP_ERR_STATE;
}

int pep_components_of_result(pep_ctx *ctx, const pep_result *r, uint32_t n_nodes, uint32_t q_base, const uint32_t *node_of_target, uint64_t n_targets, uint32_t *label)
{
    if (!ctx || !r || (n_nodes && !label) || (n_targets && !node_of_target)) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    if (r->ctx == ctx && ctx->dev_result == r && n_targets >= ctx->t.n)
        return pep_k10_components_dev(ctx, n_nodes, r->n_hits, r->d_hits, q_base, node_of_target, n_targets, label);
    PEP_TRY(pep_fetch_result(r));
    const pep_hit *h = r->st_hits ? r->st_hits : r->hits.data();
    return pep_components_of_hits(ctx, n_nodes, r->n_hits, h, q_base, node_of_target, n_targets, label);
}

int pep_merge_hits(uint64_t n, const pep_hit *hits, const uint32_t *cigar, uint64_t n_cigar, int32_t top_k, int32_t n_splits,
                   pep_hit *out_hits, uint32_t *out_cigar, uint64_t *n_out, uint64_t *n_cigar_out)
{
    if (!n_out || !n_cigar_out || top_k < 1 || n_splits < 1 || (n && (!hits || !out_hits)) || (n_cigar && (!cigar || !out_cigar))) return PEP_ERR_ARG;
    *n_out = 0; *n_cigar_out = 0;
    if (n == 0) return PEP_OK;
    if (n > 0xFFFFFFFFull) return PEP_ERR_LIMIT;
    // The input is the concatenation of per-rank tables, each already in (q, t, bin) order, and the ranks of one grid row hold
    // increasing target ranges: a STABLE counting sort by q alone restores the global order in one linear pass.  Any group that
    // turns out not to be in (t, bin) order (a caller with another layout) is sorted on its own, so the result never depends on
    // that property - only the speed does.
    uint32_t q_hi = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (hits[i].cigar_off + hits[i].cigar_runs > n_cigar) return PEP_ERR_ARG;
        q_hi = std::max(q_hi, hits[i].q);
    }
    auto by_t = [&](uint32_t a, uint32_t b) {
        const pep_hit &x = hits[a], &y = hits[b];
        if (x.t != y.t) return x.t < y.t;
        return x.bin < y.bin;
    };
    std::vector<uint32_t> order((size_t)n), first;
    if ((uint64_t)q_hi <= 64 * n + (1u << 20)) {
        first.assign((size_t)q_hi + 2, 0);
        for (uint64_t i = 0; i < n; ++i) ++first[hits[i].q + 1];
        for (size_t q = 1; q < first.size(); ++q) first[q] += first[q - 1];
        std::vector<uint32_t> at(first.begin(), first.end() - 1);
        for (uint64_t i = 0; i < n; ++i) order[at[hits[i].q]++] = (uint32_t)i;
    } else {
        for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hits[a].q < hits[b].q; });
    }
    std::vector<uint32_t> grp, count((size_t)n_splits);
    std::vector<uint8_t> drop;                                   // allocated only when some (query, split) exceeds top_k
    uint64_t no = 0, nc = 0;
    for (uint64_t a = 0; a < n;) {
        uint64_t b = a + 1;
        bool sorted = true;
        const uint32_t q = hits[order[a]].q;
        while (b < n && hits[order[b]].q == q) { sorted = sorted && !by_t(order[b], order[b - 1]); ++b; }
        if (!sorted) std::sort(order.begin() + a, order.begin() + b, by_t);
        bool over = false;
        if (b - a > (uint64_t)top_k) {
            std::fill(count.begin(), count.end(), 0u);
            for (uint64_t k = a; k < b; ++k) over = (++count[hits[order[k]].t % (uint32_t)n_splits] > (uint32_t)top_k) || over;
        }
        if (over) {
            // the reference's `-k` per database split (uberBlast.py:546-552): best score first, then target order
            if (drop.empty()) drop.assign((size_t)n, 0);
            for (int32_t sp = 0; sp < n_splits; ++sp) {
                if (count[sp] <= (uint32_t)top_k) continue;
                grp.clear();
                for (uint64_t k = a; k < b; ++k)
                    if ((int32_t)(hits[order[k]].t % (uint32_t)n_splits) == sp) grp.push_back(order[k]);
                std::sort(grp.begin(), grp.end(), [&](uint32_t u, uint32_t v) {
                    const pep_hit &x = hits[u], &y = hits[v];
                    if (x.score != y.score) return x.score > y.score;
                    return by_t(u, v);
                });
                for (size_t k = (size_t)top_k; k < grp.size(); ++k) drop[grp[k]] = 1;
            }
        }
        for (uint64_t k = a; k < b; ++k) {
            const uint32_t i = order[k];
            if (over && drop[i]) continue;
            pep_hit h = hits[i];
            if (h.cigar_runs) memcpy(out_cigar + nc, cigar + h.cigar_off, (size_t)h.cigar_runs * sizeof(uint32_t));
            h.cigar_off = nc;
            nc += h.cigar_runs;
            out_hits[no++] = h;
        }
        a = b;
    }
    *n_out = no; *n_cigar_out = nc;
    return PEP_OK;
}

int pep_rescore_nt(pep_ctx *ctx, uint64_t n, const pep_nt_hit *hits, const uint32_t *cigar, uint64_t n_cigar, int64_t *out)
{
    if (!ctx || (n && (!hits || !cigar || !out))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k7_rescore(ctx, n, hits, cigar, n_cigar, out);
}

int pep_components(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_edges, const uint32_t *a, const uint32_t *b, uint32_t *label)
{
    if (!ctx || (n_nodes && !label) || (n_edges && (!a || !b))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k10_components(ctx, n_nodes, n_edges, a, b, label);
}

int pep_components_of_hits(pep_ctx *ctx, uint32_t n_nodes, uint64_t n_hits, const pep_hit *hits, uint32_t q_base,
                           const uint32_t *node_of_target, uint64_t n_targets, uint32_t *label)
{
    if (!ctx || (n_nodes && !label) || (n_hits && (!hits || !node_of_target))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> a(n_hits + 1), b(n_hits + 1);
    for (uint64_t h = 0; h < n_hits; ++h) {
        if (hits[h].t >= n_targets) return pep_fail(ctx, PEP_ERR_ARG, "pep_components_of_hits: target index out of range");
        a[h] = hits[h].q + q_base;
        b[h] = node_of_target[hits[h].t];
    }
    return pep_k10_components(ctx, n_nodes, n_hits, a.data(), b.data(), label);
}

int pep_linclust(pep_ctx *ctx, const uint8_t *codes, const uint64_t *off, uint32_t n, int base, int k, int m, double min_id, double min_cov,
                 uint32_t *rep, uint64_t *stats)
{
    if (!ctx || (n && (!codes || !off || !rep))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return pep_fail(ctx, PEP_ERR_ARG, "offsets must be non-decreasing");
    return pep_k9_linclust(ctx, codes, off, n, base, k, m, min_id, min_cov, rep, stats);
}

int pep_overlaps(pep_ctx *ctx, uint64_t n, const int32_t *contig, const int64_t *start, const int64_t *end, const int64_t *row_id,
                 double ovl_l, double ovl_p, int64_t *out, uint64_t cap, uint64_t *n_pairs)
{
    if (!ctx || !n_pairs || (n && (!contig || !start || !end || !row_id)) || (cap && !out)) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k11_overlaps(ctx, n, contig, start, end, row_id, ovl_l, ovl_p, out, cap, n_pairs);
}

int pep_alleles(pep_ctx *ctx, const uint8_t *nt, const uint64_t *nt_off, uint32_t n_contigs, uint64_t n_rows, const pep_locus *rows,
                const uint32_t *cigar, uint64_t n_cigar, uint32_t n_groups, const uint64_t *grp_off, const uint32_t *grp_qlen, int gtable,
                int64_t *in_frame, int64_t *orf, uint8_t *packed, uint64_t packed_cap)
{
    if (!ctx || !nt_off || (n_rows && (!rows || !cigar || !in_frame || !orf)) || (n_groups && (!grp_off || !grp_qlen || !packed))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < n_contigs; ++i)
        if (nt_off[i + 1] < nt_off[i]) return pep_fail(ctx, PEP_ERR_ARG, "offsets must be non-decreasing");
    if (n_contigs && nt_off[n_contigs] && !nt) return PEP_ERR_ARG;
    return pep_k12_alleles(ctx, nt, nt_off, n_contigs, n_rows, rows, cigar, n_cigar, n_groups, grp_off, grp_qlen, gtable, in_frame, orf, packed, packed_cap);
}

int pep_sha1(pep_ctx *ctx, const uint8_t *bytes, const uint64_t *off, uint32_t n, uint8_t *digest)
{
    if (!ctx || (n && (!off || !digest))) return PEP_ERR_ARG;
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    for (uint32_t i = 0; i < n; ++i)
        if (off[i + 1] < off[i]) return pep_fail(ctx, PEP_ERR_ARG, "offsets must be non-decreasing");
    if (n && off[n] && !bytes) return PEP_ERR_ARG;
    return pep_k13_sha1(ctx, bytes, off, n, digest);
}

int pep_dedup(pep_ctx *ctx, uint32_t n, const uint32_t *len, const uint8_t *digest, uint32_t *rep)
{
    if (!ctx || (n && (!len || !digest || !rep))) return PEP_ERR_ARG;
    if (n >= 0x7FFFFFFFu) return pep_fail(ctx, PEP_ERR_LIMIT, "pep_dedup: at most 2^31 - 2 genes");
    PEP_HIP(ctx, hipSetDevice(ctx->device));
    return pep_k13_dedup(ctx, n, len, digest, rep);
}

}  // extern "C"
