// Host rANS coder + CDF quantiser behind a C ABI (see include/stem_rans.h).
// Written from the published rANS algorithm (F. Giesen, "rANS in practice": 64-bit state,
// 32-bit word renormalisation, lower bound L = 2^31) and CompressAI's stream conventions.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <vector>

#include "../../include/stem_rans.h"

#define API extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint64_t kLower = 1ull << 31;
constexpr int kPrec = 16;          // probability resolution
constexpr int kEscBits = 4;        // bypass nibble
constexpr int kEscMax = (1 << kEscBits) - 1;

thread_local char g_err[256] = "";
int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// One coded event: a table symbol (freq > 0) or a raw nibble (freq == 0, value in start).
struct Event {
    uint16_t start, freq;
};

struct Tables {
    const int32_t *cdfs, *sizes, *offsets;
    int ncdf, stride;
    bool ok() const { return cdfs && sizes && offsets && ncdf > 0 && stride >= 2; }
};

struct Encoder {
    std::vector<Event> events;
};

int push_events(std::vector<Event> &ev, const int32_t *symbols, const int32_t *indexes, size_t n, const Tables &t)
{
    if (!t.ok() || (n && (!symbols || !indexes))) return fail(-1, "rans encode: null table or input");
    ev.reserve(ev.size() + n + 16);
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= t.ncdf) return fail(-1, "rans encode: index %d out of range at %zu", ci, i);
        const int32_t len = t.sizes[ci];
        if (len < 2 || len > t.stride) return fail(-1, "rans encode: cdf %d has invalid length %d", ci, len);
        const int32_t *cdf = t.cdfs + (size_t)ci * t.stride;
        const int32_t sentinel = len - 2;                   // last table symbol doubles as the escape flag
        int64_t v = (int64_t)symbols[i] - t.offsets[ci];
        uint32_t raw = 0;
        bool esc = false;
        if (v < 0) {
            raw = (uint32_t)(-2 * v - 1);
            v = sentinel;
            esc = true;
        } else if (v >= sentinel) {
            raw = (uint32_t)(2 * (v - sentinel));
            v = sentinel;
            esc = true;
        }
        const int32_t lo = cdf[v], hi = cdf[v + 1];
        if (hi <= lo || lo < 0 || hi > (1 << kPrec)) return fail(-1, "rans encode: cdf %d not increasing at %lld", ci, (long long)v);
        ev.push_back({(uint16_t)lo, (uint16_t)(hi - lo)});
        if (esc) {
            int nn = 0;
            while ((raw >> (nn * kEscBits)) != 0) ++nn;
            int c = nn;                                     // nibble count, unary-ish in base 15
            while (c >= kEscMax) {
                ev.push_back({(uint16_t)kEscMax, 0});
                c -= kEscMax;
            }
            ev.push_back({(uint16_t)c, 0});
            for (int j = 0; j < nn; ++j) ev.push_back({(uint16_t)((raw >> (j * kEscBits)) & kEscMax), 0});
        }
    }
    return 0;
}

// x / freq without a division instruction on the coder's serial path: floor(x * m / 2^(63 + s)) with s = ceil(log2 freq),
// m = ceil(2^(63 + s) / freq) < 2^64.  Exact for every 0 <= x < 2^63 (Granlund & Montgomery, "Division by invariant integers using
// multiplication", theorem 4.2 with N = 63: 2^(N+s) <= m * freq <= 2^(N+s) + 2^s) -- and the state in front of a table symbol is
// below ((L >> 16) << 32) * freq <= 2^63.  One table for all frequencies 1 .. 2^16, built (and checked against `/`) on first use.
struct Rcp {
    uint64_t m;
    uint32_t sh;        // q = mulhi64(x, m) >> sh   (sh = s - 1; freq == 1: m = 0 marks "q = x")
};
const Rcp *rcp_table()
{
    static const std::vector<Rcp> table = [] {
        std::vector<Rcp> t((size_t)(1u << kPrec) + 1);
        t[0] = {0, 0};
        t[1] = {0, 0};
        for (uint32_t d = 2; d <= (1u << kPrec); ++d) {
            uint32_t s = 0;
            while ((1u << s) < d) ++s;
            const unsigned __int128 num = (unsigned __int128)1 << (63 + s);
            const uint64_t m = (uint64_t)((num + d - 1) / d);
            t[d] = {m, s - 1};
            // spot checks on both sides of multiples of d, at the top of the range and at the renormalisation bound
            const uint64_t top = (((uint64_t)kLower >> kPrec) << 32) * d - 1;
            const uint64_t probes[] = {0, 1, d - 1, d, d + 1, (uint64_t)d * d - 1, (uint64_t)d * d, top, top - d, top / 2 + 1, 0x7FFFFFFFFFFFFFFFull,
                                       0x7FFFFFFFFFFFFFFFull / d * d - 1, 0x7FFFFFFFFFFFFFFFull / d * d, 0x123456789ABCDEFull};
            for (uint64_t x : probes) {
                const uint64_t q = (uint64_t)(((unsigned __int128)x * m) >> 64) >> (s - 1);
                if (q != x / d) {
                    fprintf(stderr, "rans_host: reciprocal table wrong for d=%u x=%llu\n", d, (unsigned long long)x);
                    abort();
                }
            }
        }
        return t;
    }();
    return table.data();
}

// Encode events last-to-first; words are emitted back-to-front into a scratch vector.
long flush_events(std::vector<Event> &ev, uint8_t *out, size_t cap)
{
    const Rcp *rcp = rcp_table();
    std::vector<uint32_t> words(ev.size() + 2);
    size_t pos = words.size();
    uint64_t x = kLower;
    for (size_t k = ev.size(); k-- > 0;) {
        const Event e = ev[k];
        if (e.freq) {
            const uint64_t limit = ((kLower >> kPrec) << 32) * e.freq;
            if (x >= limit) {
                words[--pos] = (uint32_t)x;
                x >>= 32;
            }
            const Rcp r = rcp[e.freq];
            const uint64_t q = r.m ? (uint64_t)(((unsigned __int128)x * r.m) >> 64) >> r.sh : x;
            x = (q << kPrec) + (x - q * e.freq) + e.start;
        } else {
            const uint64_t limit = ((kLower >> 16) << 32) * (uint64_t)(1u << (16 - kEscBits));
            if (x >= limit) {
                words[--pos] = (uint32_t)x;
                x >>= 32;
            }
            x = (x << kEscBits) | e.start;
        }
    }
    words[--pos] = (uint32_t)(x >> 32);
    words[--pos] = (uint32_t)x;
    const size_t nbytes = (words.size() - pos) * 4;
    ev.clear();
    if (nbytes > cap) return fail(-2, "rans flush: need %zu bytes, capacity %zu", nbytes, cap);
    memcpy(out, words.data() + pos, nbytes);
    return (long)nbytes;
}

// Symbol search of the decoder: lut[row][cum >> kLutShift] = the symbol whose interval holds the bucket's first value; from there a
// short walk forward finds the symbol of `cum` (a binary search over up to 3133 entries mispredicts a dozen branches per symbol, and
// the decoder's symbols sit on the dependent path of the raster-order loop: csrc/ar_persistent.hip).  Built per decoder object once it
// has seen kLutAfter symbols with the same tables; a row that is not a strictly increasing 0 .. 2^16 sequence keeps the binary search.
constexpr int kLutShift = 6, kLutBuckets = 1 << (kPrec - kLutShift);
constexpr size_t kLutAfter = 2048;

struct Decoder {
    std::vector<uint32_t> words;
    size_t pos = 0;
    uint64_t x = 0;
    bool ready = false;
    std::vector<uint16_t> lut;
    std::vector<uint8_t> lut_row_ok;
    const int32_t *lut_cdfs = nullptr, *lut_sizes = nullptr;
    int lut_ncdf = 0, lut_stride = 0;
    size_t seen = 0;

    void build_lut(const Tables &t)
    {
        lut.assign((size_t)t.ncdf * kLutBuckets, 0);
        lut_row_ok.assign((size_t)t.ncdf, 0);
        for (int ci = 0; ci < t.ncdf; ++ci) {
            const int32_t len = t.sizes[ci];
            if (len < 2 || len > t.stride || len > 65535) continue;
            const int32_t *cdf = t.cdfs + (size_t)ci * t.stride;
            bool ok = cdf[0] == 0 && cdf[len - 1] == (1 << kPrec);
            for (int s = 1; ok && s < len; ++s) ok = cdf[s] > cdf[s - 1];
            if (!ok) continue;
            int s = 0;
            for (int b = 0; b < kLutBuckets; ++b) {
                const uint32_t first = (uint32_t)b << kLutShift;
                while ((uint32_t)cdf[s + 1] <= first) ++s;          // cdf[len - 1] = 2^16 > first: s stays below len - 1
                lut[(size_t)ci * kLutBuckets + b] = (uint16_t)s;
            }
            lut_row_ok[(size_t)ci] = 1;
        }
        lut_cdfs = t.cdfs; lut_sizes = t.sizes; lut_ncdf = t.ncdf; lut_stride = t.stride;
    }

    inline bool refill()
    {
        if (x < kLower) {
            if (pos >= words.size()) return false;
            x = (x << 32) | words[pos++];
        }
        return true;
    }
    inline bool nibble(int &v)
    {
        v = (int)(x & kEscMax);
        x >>= kEscBits;
        return refill();
    }
};

int set_stream(Decoder &d, const uint8_t *stream, size_t nbytes)
{
    if (!stream || nbytes < 8 || (nbytes & 3)) return fail(-1, "rans decode: stream of %zu bytes is not a rANS stream", nbytes);
    d.words.resize(nbytes / 4);
    memcpy(d.words.data(), stream, nbytes);
    d.x = (uint64_t)d.words[0] | ((uint64_t)d.words[1] << 32);
    d.pos = 2;
    d.ready = true;
    d.lut_cdfs = nullptr;           // a new stream may come with new tables behind the same pointers
    d.seen = 0;
    return 0;
}

int decode(Decoder &d, const int32_t *indexes, size_t n, const Tables &t, int32_t *out)
{
    if (!d.ready) return fail(-1, "rans decode: set_stream was not called");
    if (!t.ok() || (n && (!indexes || !out))) return fail(-1, "rans decode: null table or input");
    const bool same = d.lut_cdfs == t.cdfs && d.lut_sizes == t.sizes && d.lut_ncdf == t.ncdf && d.lut_stride == t.stride;
    d.seen += n;
    if (!same && d.seen > kLutAfter) d.build_lut(t);
    const bool fast = d.lut_cdfs == t.cdfs && d.lut_sizes == t.sizes && d.lut_ncdf == t.ncdf && d.lut_stride == t.stride;
    for (size_t i = 0; i < n; ++i) {
        const int32_t ci = indexes[i];
        if (ci < 0 || ci >= t.ncdf) return fail(-1, "rans decode: index %d out of range at %zu", ci, i);
        const int32_t len = t.sizes[ci];
        if (len < 2 || len > t.stride) return fail(-1, "rans decode: cdf %d has invalid length %d", ci, len);
        const int32_t *cdf = t.cdfs + (size_t)ci * t.stride;
        const uint32_t cum = (uint32_t)(d.x & ((1u << kPrec) - 1));
        // largest s with cdf[s] <= cum (cdf is strictly increasing, cdf[0] = 0, cdf[len-1] = 2^16)
        int lo = 0;
        bool found = false;
        if (fast && d.lut_row_ok[(size_t)ci]) {
            // The lookup table is keyed on the tables' ADDRESSES: should their contents have been rewritten in place since it was
            // built, the start may be wrong and the walk must neither leave the row nor return a wrong symbol -- it is bounded by the
            // row, its answer is checked (cdf[lo] <= cum < cdf[lo + 1]), and anything else goes to the binary search below.
            lo = d.lut[(size_t)ci * kLutBuckets + (cum >> kLutShift)];
            if (lo > len - 2) lo = len - 2;
            while (lo < len - 2 && (uint32_t)cdf[lo + 1] <= cum) ++lo;
            found = (uint32_t)cdf[lo] <= cum && cum < (uint32_t)cdf[lo + 1];
            if (!found) {
                d.lut_cdfs = nullptr;                                 // stale: rebuilt from the current contents at the next call
                lo = 0;
            }
        }
        if (!found) {
            int hi = len - 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((uint32_t)cdf[mid] <= cum)
                    lo = mid;
                else
                    hi = mid;
            }
        }
        const uint32_t start = (uint32_t)cdf[lo], freq = (uint32_t)(cdf[lo + 1] - cdf[lo]);
        d.x = (uint64_t)freq * (d.x >> kPrec) + cum - start;
        if (!d.refill()) return fail(-3, "rans decode: stream exhausted at symbol %zu", i);
        int32_t v = lo;
        if (v == len - 2) {
            int c;
            if (!d.nibble(c)) return fail(-3, "rans decode: stream exhausted in escape at %zu", i);
            int nn = c;
            // the count is stored in unary-ish nibbles (rans_interface.cpp:137-163); a 32-bit raw value never needs more
            // than 32 / kEscBits of them, so anything larger is a corrupt (or hostile) stream: shifting by >= 32 below
            // would be undefined behaviour and the loop count would be attacker controlled
            constexpr int kMaxEscNibbles = 32 / kEscBits;
            while (c == kEscMax) {
                if (nn > kMaxEscNibbles) break;
                if (!d.nibble(c)) return fail(-3, "rans decode: stream exhausted in escape at %zu", i);
                nn += c;
            }
            if (nn > kMaxEscNibbles) return fail(-4, "rans decode: corrupt escape at symbol %zu (%d bypass nibbles > %d)", i, nn, kMaxEscNibbles);
            uint32_t raw = 0;
            for (int j = 0; j < nn; ++j) {
                if (!d.nibble(c)) return fail(-3, "rans decode: stream exhausted in escape at %zu", i);
                raw |= (uint32_t)c << (j * kEscBits);
            }
            v = (int32_t)(raw >> 1);
            v = (raw & 1) ? -v - 1 : v + (len - 2);
        }
        out[i] = v + t.offsets[ci];
    }
    return 0;
}

}   // namespace

API const char *stem_rans_last_error(void) { return g_err; }

API long stem_rans_encode(const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs, int ncdf,
                          int cdf_stride, const int32_t *sizes, const int32_t *offsets, uint8_t *out, size_t cap)
{
    std::vector<Event> ev;
    const Tables t{cdfs, sizes, offsets, ncdf, cdf_stride};
    if (int rc = push_events(ev, symbols, indexes, n, t)) return rc;
    if (!out) return fail(-1, "rans encode: null output");
    return flush_events(ev, out, cap);
}

API int stem_rans_decode(const uint8_t *stream, size_t nbytes, const int32_t *indexes, size_t n, const int32_t *cdfs,
                         int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, int32_t *out)
{
    Decoder d;
    if (int rc = set_stream(d, stream, nbytes)) return rc;
    return decode(d, indexes, n, Tables{cdfs, sizes, offsets, ncdf, cdf_stride}, out);
}

API void *stem_rans_encoder_create(void) { return new (std::nothrow) Encoder(); }
API void stem_rans_encoder_destroy(void *enc) { delete static_cast<Encoder *>(enc); }
API int stem_rans_encoder_push(void *enc, const int32_t *symbols, const int32_t *indexes, size_t n, const int32_t *cdfs,
                               int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets)
{
    if (!enc) return fail(-1, "rans encoder: null handle");
    return push_events(static_cast<Encoder *>(enc)->events, symbols, indexes, n, Tables{cdfs, sizes, offsets, ncdf, cdf_stride});
}
API size_t stem_rans_encoder_pending_bytes(void *enc)
{
    return enc ? (static_cast<Encoder *>(enc)->events.size() + 2) * 4 : 0;
}
API long stem_rans_encoder_flush(void *enc, uint8_t *out, size_t cap)
{
    if (!enc || !out) return fail(-1, "rans encoder: null handle or output");
    return flush_events(static_cast<Encoder *>(enc)->events, out, cap);
}

API void *stem_rans_decoder_create(void) { return new (std::nothrow) Decoder(); }
API void stem_rans_decoder_destroy(void *dec) { delete static_cast<Decoder *>(dec); }
API int stem_rans_decoder_set_stream(void *dec, const uint8_t *stream, size_t nbytes)
{
    if (!dec) return fail(-1, "rans decoder: null handle");
    return set_stream(*static_cast<Decoder *>(dec), stream, nbytes);
}
API int stem_rans_decoder_decode(void *dec, const int32_t *indexes, size_t n, const int32_t *cdfs, int ncdf, int cdf_stride,
                                 const int32_t *sizes, const int32_t *offsets, int32_t *out)
{
    if (!dec) return fail(-1, "rans decoder: null handle");
    return decode(*static_cast<Decoder *>(dec), indexes, n, Tables{cdfs, sizes, offsets, ncdf, cdf_stride}, out);
}

// Quantise a pmf to a strictly increasing 2^precision CDF: round, renormalise, and where two
// neighbours collide take one count from the least-frequent symbol that can spare it.
API int stem_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf)
{
    if (!pmf || !cdf || n < 1 || precision < 1 || precision > 16) return fail(-1, "pmf_to_quantized_cdf: bad arguments");
    const uint32_t one = 1u << precision;
    cdf[0] = 0;
    uint32_t total = 0;
    for (int i = 0; i < n; ++i) {
        cdf[i + 1] = (uint32_t)std::round(pmf[i] * (float)one);
        total += cdf[i + 1];
    }
    if (total == 0) return fail(-1, "pmf_to_quantized_cdf: pmf sums to zero");
    uint32_t run = 0;
    for (int i = 1; i <= n; ++i) {
        run += (uint32_t)(((uint64_t)one * cdf[i]) / total);
        cdf[i] = run;
    }
    cdf[n] = one;
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        uint32_t best = ~0u;
        int donor = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best) {
                best = f;
                donor = j;
            }
        }
        if (donor < 0) return fail(-1, "pmf_to_quantized_cdf: more symbols than probability mass");
        if (donor < i)
            for (int j = donor + 1; j <= i; ++j) --cdf[j];
        else
            for (int j = i + 1; j <= donor; ++j) ++cdf[j];
    }
    return 0;
}
