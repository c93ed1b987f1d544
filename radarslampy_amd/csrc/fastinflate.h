// A DEFLATE (RFC 1951) decoder for whole, contiguous streams - the inflate of csrc/pngdec.hip (f2: record ingest from PNG files).
//
// zlib 1.2.11's inflate() takes 4.5 ms of a host core for one Oxford record (3779 x 400, Sub-filtered, ~486 KB of literal-heavy
// Huffman data) and the single-sequence engine wants 2 500 of them a second from the ~16 cores a rank may count on.  This decoder does
// the same job in about a third of the time by not being general: the whole compressed stream and the whole output are in memory (no
// streaming state, no window copy), both buffers carry slack (reads of 8 bytes past the input's end and copies up to 258 + 8 bytes past
// the output's end stay in bounds, and are checked AFTER the fact), the bit buffer is 64 bits wide and refilled with one unaligned load,
// literal / length codes resolve through a 2 048-entry table (longer codes through sub-tables) that holds base value, extra-bit count and
// code length in one word.  Anything it does not like - a malformed block, an over-subscribed or incomplete code, a distance before the
// start of the output, output that does not end exactly where it should - is reported as failure and the caller inflates the stream again
// with zlib, which owns the verdict.  Nothing here is derived from zlib's or libdeflate's sources; the format is the RFC's.
#pragma once
#include <cstdint>
#include <cstring>

namespace fastinflate {

constexpr int LIT_ROOT = 11, DIST_ROOT = 8;
constexpr int LIT_TABLE = (1 << LIT_ROOT) + 1024, DIST_TABLE = (1 << DIST_ROOT) + 512;     // roots + room for every sub-table (checked)
// table word: [4:0] code length (sub-table pointer: bits of the sub-table index; a literal PAIR: both codes), [8:5] extra bits, [11:9] kind,
// [13:12] literals in the word (1, or 2: a root entry whose index bits hold two whole literal codes decodes both at once - Sub-filtered
// radar speckle is ~3 bits a literal and nearly all literals), [31:16] value (literal pair: first | second << 8)
enum : uint32_t { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
static inline uint32_t mk(uint32_t kind, uint32_t value, uint32_t extra, uint32_t len) { return len | (extra << 5) | (kind << 9) | (value << 16) | ((kind == K_LIT ? 1u : 0u) << 12); }

struct Tables { uint32_t lit[LIT_TABLE], dist[DIST_TABLE]; };

static inline uint32_t bitrev(uint32_t c, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (c & 1); c >>= 1; }
    return r;
}

// canonical Huffman code of lens[0..n) -> two-level table; entry(sym, len) makes the word of a symbol.  false: not a usable code
template <class Entry>
static bool build(const uint8_t *lens, int n, int root, uint32_t *table, int cap, Entry entry)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    const int rootsize = 1 << root;
    if (count[0] == n) {                                            // no code at all (a block of literals only has no distance code)
        for (int i = 0; i < rootsize; i++) table[i] = mk(K_BAD, 0, 0, 1);
        return true;
    }
    int left = 1, used = 0;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return false; used += count[l]; }
    if (left > 0 && !(used == 1 && count[1] == 1)) return false;    // incomplete: only the one-code case of the RFC
    uint32_t next[16];
    { uint32_t c = 0; for (int l = 1; l <= 15; l++) { c = (c + (uint32_t)count[l - 1] * (l > 1)) << 1; next[l] = c; } }
    for (int i = 0; i < rootsize; i++) table[i] = mk(K_BAD, 0, 0, 1);
    // longest code behind every root prefix -> sub-table sizes
    uint8_t subbits[1 << LIT_ROOT];
    memset(subbits, 0, (size_t)rootsize);
    {
        uint32_t nx[16];
        memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t rev = bitrev(nx[l]++, l);
            if (l > root) { const uint32_t p = rev & (uint32_t)(rootsize - 1); if (l - root > subbits[p]) subbits[p] = (uint8_t)(l - root); }
        }
    }
    int top = rootsize;
    uint16_t suboff[1 << LIT_ROOT];
    for (int p = 0; p < rootsize; p++)
        if (subbits[p]) {
            if (top + (1 << subbits[p]) > cap) return false;
            suboff[p] = (uint16_t)top;
            table[p] = mk(K_SUB, (uint32_t)top, 0, subbits[p]);
            for (int i = 0; i < (1 << subbits[p]); i++) table[top + i] = mk(K_BAD, 0, 0, (uint32_t)root + 1);
            top += 1 << subbits[p];
        }
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = bitrev(next[l]++, l);
        const uint32_t e = entry(s, l);
        if (l <= root)
            for (uint32_t i = rev; i < (uint32_t)rootsize; i += 1u << l) table[i] = e;
        else {
            const uint32_t p = rev & (uint32_t)(rootsize - 1);
            for (uint32_t i = rev >> root; i < (1u << subbits[p]); i += 1u << (l - root)) table[suboff[p] + i] = e;
        }
    }
    return true;
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline bool build_litlen(const uint8_t *lens, int n, uint32_t *t)
{
    if (!build(lens, n, LIT_ROOT, t, LIT_TABLE, [](int s, int l) -> uint32_t {
        if (s < 256) return mk(K_LIT, (uint32_t)s, 0, (uint32_t)l);
        if (s == 256) return mk(K_EOB, 0, 0, (uint32_t)l);
        if (s > 285) return mk(K_BAD, 0, 0, (uint32_t)l);
        return mk(K_BASE, LEN_BASE[s - 257], LEN_EXTRA[s - 257], (uint32_t)l);
    })) return false;
    // literal pairs: index i = [code of literal 1][next bits]; if the next bits hold a whole second literal code, the entry decodes both
    uint32_t one[1 << LIT_ROOT];
    memcpy(one, t, sizeof one);
    for (uint32_t i = 0; i < (1u << LIT_ROOT); i++) {
        const uint32_t e1 = one[i];
        if (((e1 >> 9) & 7) != K_LIT) continue;
        const uint32_t l1 = e1 & 31, rem = LIT_ROOT - l1;
        const uint32_t e2 = one[i >> l1];                            // (the unknown upper index bits read as zero: right whenever the code fits in rem)
        if (((e2 >> 9) & 7) != K_LIT || (e2 & 31) > rem) continue;
        t[i] = (l1 + (e2 & 31)) | (K_LIT << 9) | (2u << 12) | (((e1 >> 16) | ((e2 >> 16) << 8)) << 16);
    }
    return true;
}
static inline bool build_dist(const uint8_t *lens, int n, uint32_t *t)
{
    return build(lens, n, DIST_ROOT, t, DIST_TABLE, [](int s, int l) -> uint32_t {
        if (s > 29) return mk(K_BAD, 0, 0, (uint32_t)l);
        return mk(K_BASE, DIST_BASE[s], DIST_EXTRA[s], (uint32_t)l);
    });
}

static inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }      // (little-endian host: x86-64)

// in[0, in_len): a raw DEFLATE stream, readable up to in_len + 8.  out[0, out_len): where exactly out_len bytes must land, writable up
// to out_len + 258 + 16.  *consumed = bytes of input used (the next whole byte after the final block).  false: see the header.
static bool inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len, size_t *consumed, Tables &T)
{
    const uint8_t *ip = in, *const iend = in + in_len;
    uint8_t *op = out, *const oend = out + out_len;
    uint64_t bb = 0;
    int bc = 0;                                                     // valid bits in bb
#define FI_REFILL() do { bb |= load64(ip) << bc; ip += (63 - bc) >> 3; bc |= 56; } while (0)
#define FI_DROP(n_) do { bb >>= (n_); bc -= (n_); } while (0)
    for (;;) {
        if (ip > iend) return false;
        FI_REFILL();
        const int final = (int)(bb & 1), type = (int)((bb >> 1) & 3);
        FI_DROP(3);
        if (type == 0) {                                            // stored: to the byte boundary, LEN, ~LEN, bytes
            FI_DROP(bc & 7);
            // the bytes still in the bit buffer belong to the block: step the pointer back over them
            ip -= bc >> 3; bb = 0; bc = 0;
            if (ip + 4 > iend) return false;
            const uint32_t len = ip[0] | (ip[1] << 8), nlen = ip[2] | (ip[3] << 8);
            ip += 4;
            if ((len ^ nlen) != 0xffffu || ip + len > iend || op + len > oend) return false;
            memcpy(op, ip, len);
            op += len; ip += len;
        } else if (type == 3)
            return false;
        else {
            if (type == 1) {                                        // the fixed code
                uint8_t l[288 + 32];
                for (int i = 0; i < 144; i++) l[i] = 8;
                for (int i = 144; i < 256; i++) l[i] = 9;
                for (int i = 256; i < 280; i++) l[i] = 7;
                for (int i = 280; i < 288; i++) l[i] = 8;
                for (int i = 0; i < 32; i++) l[288 + i] = 5;
                if (!build_litlen(l, 288, T.lit) || !build_dist(l + 288, 32, T.dist)) return false;
            } else {                                                // dynamic: code-length code, then the two codes' lengths
                const int hlit = (int)(bb & 31) + 257, hdist = (int)((bb >> 5) & 31) + 1, hclen = (int)((bb >> 10) & 15) + 4;
                FI_DROP(14);
                if (hlit > 286 || hdist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) {
                    if (bc < 3) { if (ip > iend) return false; FI_REFILL(); }
                    cl[order[i]] = (uint8_t)(bb & 7);
                    FI_DROP(3);
                }
                uint32_t clt[1 << 7];                               // codes of at most 7 bits: one level
                if (!build(cl, 19, 7, clt, 1 << 7, [](int s, int l) -> uint32_t { return mk(K_LIT, (uint32_t)s, 0, (uint32_t)l); })) return false;
                uint8_t l[286 + 30 + 138];
                int i = 0;
                const int n = hlit + hdist;
                while (i < n) {
                    if (ip > iend) return false;
                    if (bc < 15) FI_REFILL();
                    const uint32_t e = clt[bb & 127];
                    if (((e >> 9) & 7) != K_LIT) return false;
                    FI_DROP((int)(e & 31));
                    const int s = (int)(e >> 16);
                    if (s < 16) l[i++] = (uint8_t)s;
                    else {
                        int rep, v = 0;
                        if (s == 16) { if (!i) return false; v = l[i - 1]; rep = 3 + (int)(bb & 3); FI_DROP(2); }
                        else if (s == 17) { rep = 3 + (int)(bb & 7); FI_DROP(3); }
                        else { rep = 11 + (int)(bb & 127); FI_DROP(7); }
                        if (i + rep > n) return false;
                        while (rep--) l[i++] = (uint8_t)v;
                    }
                }
                if (!l[256]) return false;                          // no end-of-block code
                if (!build_litlen(l, hlit, T.lit) || !build_dist(l + hlit, hdist, T.dist)) return false;
            }
            // ---- the block's symbols
            const uint32_t *lit = T.lit, *dst = T.dist;
            for (;;) {
                if (ip > iend || op > oend) return false;           // (slack absorbs what was read / written since the last check)
                FI_REFILL();
                uint32_t e = lit[bb & ((1u << LIT_ROOT) - 1)];
                if (((e >> 9) & 7) == K_LIT) {                      // literals come in runs: up to three words (six literals) per refill (3 x 15 bits < 56)
#define FI_PUT() do { const uint16_t v_ = (uint16_t)(e >> 16); memcpy(op, &v_, 2); op += (e >> 12) & 3; FI_DROP((int)(e & 31)); } while (0)
                    FI_PUT();
                    e = lit[bb & ((1u << LIT_ROOT) - 1)];
                    if (((e >> 9) & 7) == K_LIT) {
                        FI_PUT();
                        e = lit[bb & ((1u << LIT_ROOT) - 1)];
                        if (((e >> 9) & 7) == K_LIT) {
                            FI_PUT();
                            continue;
                        }
                    }
                    if (bc < 48) { if (ip > iend) return false; FI_REFILL(); }
                }
                uint32_t kind = (e >> 9) & 7;
                if (kind == K_SUB) {
                    e = lit[(e >> 16) + ((bb >> LIT_ROOT) & ((1u << (e & 31)) - 1))];
                    kind = (e >> 9) & 7;
                    if (kind == K_LIT) { FI_DROP((int)(e & 31)); *op++ = (uint8_t)(e >> 16); continue; }    // (sub-table entries are single literals)
                }
                if (kind == K_EOB) { FI_DROP((int)(e & 31)); break; }
                if (kind != K_BASE) return false;
                FI_DROP((int)(e & 31));
                const int xl = (int)((e >> 5) & 15);
                const uint32_t len = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1));
                FI_DROP(xl);
                // at least 56 - 15 - 5 = 36 bits left... not enough for 15 + 13: top up
                if (bc < 32) { if (ip > iend) return false; FI_REFILL(); }
                uint32_t d = dst[bb & ((1u << DIST_ROOT) - 1)];
                if (((d >> 9) & 7) == K_SUB) d = dst[(d >> 16) + ((bb >> DIST_ROOT) & ((1u << (d & 31)) - 1))];
                if (((d >> 9) & 7) != K_BASE) return false;
                FI_DROP((int)(d & 31));
                const int xd = (int)((d >> 5) & 15);
                const uint32_t dist = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1));
                FI_DROP(xd);
                if (dist > (size_t)(op - out) || op + len > oend) return false;
                const uint8_t *src = op - dist;
                uint8_t *const mend = op + len;
                if (dist >= 8) {
                    do { memcpy(op, src, 8); op += 8; src += 8; } while (op < mend);
                } else if (dist == 1) {
                    memset(op, *src, len);
                } else {
                    do { *op++ = *src++; } while (op < mend);
                }
                op = mend;
            }
        }
        if (final) break;
    }
#undef FI_REFILL
#undef FI_DROP
#undef FI_PUT
    if (op != oend) return false;
    // whole bytes of lookahead still in the bit buffer were never used
    ip -= bc >> 3;
    if (ip > iend) return false;
    *consumed = (size_t)(ip - in);
    return true;
}

}  // namespace fastinflate
