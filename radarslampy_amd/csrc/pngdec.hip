// f2: native PNG decode of Oxford radar records, host code only (no device work in this file).
//
// Replaces cv2.imread(path, cv2.IMREAD_GRAYSCALE) of the reference's loader (parseData.py:160-226, the read at :178;
// RawROAMSystem.py:162-165) for the one format the data set uses: 8-bit greyscale, non-interlaced.  The IDAT stream is inflated by
// csrc/fastinflate.h (whole stream in, whole filtered image out; Adler-32 checked) and the PNG filter (None / Sub / Up / Average / Paeth,
// one byte per pixel; the data set's files are Sub on every line: a byte prefix sum, SSE2) is undone while the lines move to the
// caller's destination rows - a slot of the pinned upload ring (roam_host_alloc): no Python, no GIL, no copy between processes.  Whatever
// the fast inflate refuses is inflated again by zlib, a group of scanlines at a time, which owns the verdict on a damaged file.
// Any other colour type, bit depth or interlacing is refused with ROAM_E_ARG.
// A pool of host threads (roam_png_pool_*) decodes files ahead of the consumer; tickets complete in any order and are awaited one by one.
#include "roam_internal.h"
#include "fastinflate.h"
#include <zlib.h>
#include <emmintrin.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

static inline int paeth_pred(int a, int b, int c)
{
    // branch-free: which of the three wins is a coin toss on radar speckle, and a mispredicted branch per byte cost more than the arithmetic
    const int u = b - c, v = a - c, w = u + v;
    const int su = u >> 31, sv = v >> 31, sw = w >> 31;
    const int pa = (u ^ su) - su, pb = (v ^ sv) - sv, pc = (w ^ sw) - sw;
    const int mb = -(int)(pb <= pc), ma = -(int)((pa <= pb) & (pa <= pc));
    int pr = c ^ ((b ^ c) & mb);
    pr ^= (a ^ pr) & ma;
    return pr;
}
// undo one scanline's filter; cur: filtered bytes in, raw bytes out (n of them), prev: the raw line above (nullptr for the first line)
void unfilter_row(int ft, uint8_t *__restrict__ cur, const uint8_t *__restrict__ prev, int n)
{
    switch (ft) {
    case 0: break;
    case 1:                                                         // Sub: + left
        for (int i = 1; i < n; i++) cur[i] = (uint8_t)(cur[i] + cur[i - 1]);
        break;
    case 2:                                                         // Up: + above
        if (prev) for (int i = 0; i < n; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
        break;
    case 3:                                                         // Average: + floor((left + above) / 2)
        if (prev) {
            cur[0] = (uint8_t)(cur[0] + (prev[0] >> 1));
            for (int i = 1; i < n; i++) cur[i] = (uint8_t)(cur[i] + ((cur[i - 1] + prev[i]) >> 1));
        } else
            for (int i = 1; i < n; i++) cur[i] = (uint8_t)(cur[i] + (cur[i - 1] >> 1));
        break;
    case 4:                                                         // Paeth: + the one of left / above / upper-left nearest to left + above - upper-left
        if (prev) {
            cur[0] = (uint8_t)(cur[0] + prev[0]);
            int a = cur[0], c = prev[0];
            for (int i = 1; i < n; i++) {
                const int b = prev[i];
                a = (uint8_t)(cur[i] + paeth_pred(a, b, c));
                cur[i] = (uint8_t)a;
                c = b;
            }
        } else
            for (int i = 1; i < n; i++) cur[i] = (uint8_t)(cur[i] + cur[i - 1]);    // above and upper-left are zero: the predictor is left
        break;
    }
}

// Paeth on NR consecutive scanlines at once.  A row is one dependent chain (every byte needs the byte to its left: ~8 cycles a byte, 9 of the
// 19 ms of a 400 x 3779 record on one core); row k only needs row k - 1 up to the column it is at, so NR rows advance together, each one
// column behind the row above - NR independent chains in flight.  rows[k]: filtered bytes in, raw bytes out; rows[0]'s line above is prev
// (never nullptr here).
template <int NR>
void unfilter_paeth_rows(uint8_t *const *rows, const uint8_t *__restrict__ prev, int n)
{
    // state in registers only: left[k] = the byte row k wrote last (its left neighbour now, and the "above" of row k + 1 in the next
    // step), ul[k] = the "above" row k used last (its upper-left now).  Nothing a step reads from memory was written by an earlier step.
    int left[NR], ul[NR];
    for (int k = 0; k < NR; k++) { left[k] = 0; ul[k] = 0; }
    uint8_t *r[NR];
    for (int k = 0; k < NR; k++) r[k] = rows[k];
    // step t: row k is at column t - k
    for (int t = 0; t < n + NR - 1; t++) {
        int up[NR], nw[NR];
        up[0] = t < n ? prev[t] : 0;
#pragma unroll
        for (int k = 1; k < NR; k++) up[k] = left[k - 1];
        if (t >= NR - 1 && t < n) {                                 // every row is inside: the hot part, no guards
#pragma unroll
            for (int k = 0; k < NR; k++) nw[k] = (uint8_t)(r[k][t - k] + paeth_pred(left[k], up[k], ul[k]));
#pragma unroll
            for (int k = 0; k < NR; k++) { r[k][t - k] = (uint8_t)nw[k]; left[k] = nw[k]; ul[k] = up[k]; }
        } else {
            for (int k = 0; k < NR; k++) {
                const int x = t - k;
                nw[k] = (x >= 0 && x < n) ? (uint8_t)(r[k][x] + paeth_pred(left[k], up[k], ul[k])) : left[k];
            }
            for (int k = 0; k < NR; k++) {
                const int x = t - k;
                if (x >= 0 && x < n) { r[k][x] = (uint8_t)nw[k]; left[k] = nw[k]; ul[k] = up[k]; }
            }
        }
    }
}

// Sub on one line, source and destination apart: dst[i] = src[i] + dst[i - 1] - a byte prefix sum, 16 bytes a step (log-step shifts and adds
// inside the register, the last byte carried to the next step)
void unfilter_sub_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, int n)
{
    int i = 0;
    __m128i carry = _mm_setzero_si128();
    for (; i + 16 <= n; i += 16) {
        __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 1));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 2));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 4));
        x = _mm_add_epi8(x, _mm_slli_si128(x, 8));
        x = _mm_add_epi8(x, carry);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(dst + i), x);
        carry = _mm_set1_epi8((char)(_mm_extract_epi16(x, 7) >> 8));
    }
    int a = i ? dst[i - 1] : 0;
    for (; i < n; i++) { a = (uint8_t)(src[i] + a); dst[i] = (uint8_t)a; }
}

struct Scratch { std::vector<uint8_t> file, line, z, raw; fastinflate::Tables tab; };

// undo the filters of lines [y0, y0 + nl) whose filtered bytes (filter type first) lie LP apart from `lines`, writing the image rows of `out`.
// false: a filter type that does not exist
bool unfilter_lines(const uint8_t *lines, size_t LP, uint32_t y0, uint32_t nl, uint32_t W, uint8_t *out, int64_t out_stride)
{
    for (uint32_t i = 0; i < nl; i++)
        if (lines[LP * i] > 4) return false;
    uint32_t i = 0;
    while (i < nl) {
        const int ft = lines[LP * i];
        uint8_t *dst = out + (int64_t)(y0 + i) * out_stride;
        if (ft == 1) { unfilter_sub_copy(lines + LP * i + 1, dst, (int)W); i++; continue; }
        uint32_t run = 1;
        if (ft == 4 && y0 + i > 0)
            while (i + run < nl && run < 4 && lines[LP * (i + run)] == 4) run++;
        for (uint32_t k = 0; k < run; k++) memcpy(dst + (int64_t)k * out_stride, lines + LP * (i + k) + 1, W);
        if (run >= 2) {
            uint8_t *rw[4];
            for (uint32_t k = 0; k < run; k++) rw[k] = dst + (int64_t)k * out_stride;
            if (run == 4) unfilter_paeth_rows<4>(rw, dst - out_stride, (int)W);
            else if (run == 3) unfilter_paeth_rows<3>(rw, dst - out_stride, (int)W);
            else unfilter_paeth_rows<2>(rw, dst - out_stride, (int)W);
        } else
            unfilter_row(ft, dst, (y0 + i) ? dst - out_stride : nullptr, (int)W);
        i += run;
    }
    return true;
}

// the fast path: every IDAT payload gathered into one buffer, inflated in one go, Adler-32 compared, filters undone.  false: use zlib
bool decode_fast(const uint8_t *png, int64_t nbytes, int64_t pos, uint32_t W, uint32_t H, uint8_t *out, int64_t out_stride, Scratch &sc)
{
    size_t zn = 0;
    for (int64_t p = pos; p + 12 <= nbytes;) {
        const uint32_t len = be32(png + p);
        if (p + 12 + (int64_t)len > nbytes) return false;
        if (memcmp(png + p + 4, "IDAT", 4) == 0) zn += len;
        else if (memcmp(png + p + 4, "IEND", 4) == 0) break;
        p += 12 + (int64_t)len;
    }
    if (zn < 2 + 4) return false;
    sc.z.resize(zn + 64);
    size_t at = 0;
    for (int64_t p = pos; p + 12 <= nbytes;) {
        const uint32_t len = be32(png + p);
        if (memcmp(png + p + 4, "IDAT", 4) == 0) { memcpy(sc.z.data() + at, png + p + 8, len); at += len; }
        else if (memcmp(png + p + 4, "IEND", 4) == 0) break;
        p += 12 + (int64_t)len;
    }
    memset(sc.z.data() + zn, 0, 64);
    const uint8_t *z = sc.z.data();
    if ((z[0] & 15) != 8 || (z[0] >> 4) > 7 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return false;     // deflate, window <= 32 KB, no dictionary
    const size_t LP = (size_t)W + 1, rawn = LP * H;
    sc.raw.resize(rawn + 258 + 64);
    size_t used = 0;
    if (!fastinflate::inflate(z + 2, zn - 2, sc.raw.data(), rawn, &used, sc.tab)) return false;
    if (2 + used + 4 > zn) return false;
    if (be32(z + 2 + used) != (uint32_t)adler32(adler32(0L, Z_NULL, 0), sc.raw.data(), (uInt)rawn)) return false;
    return unfilter_lines(sc.raw.data(), LP, 0, H, W, out, out_stride);
}

int32_t decode_gray8(const uint8_t *png, int64_t nbytes, uint8_t *out, int64_t out_bytes, int64_t out_stride, int32_t *rows, int32_t *cols,
                     Scratch &sc)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!png || nbytes < 8 + 25 || memcmp(png, sig, 8) != 0) return ROAM_E_ARG;
    int64_t pos = 8;
    if (be32(png + pos) != 13 || memcmp(png + pos + 4, "IHDR", 4) != 0) return ROAM_E_ARG;
    const uint32_t W = be32(png + pos + 8), H = be32(png + pos + 12);
    const int depth = png[pos + 16], ctype = png[pos + 17], comp = png[pos + 18], filt = png[pos + 19], lace = png[pos + 20];
    if (depth != 8 || ctype != 0 || comp != 0 || filt != 0 || lace != 0) return ROAM_E_ARG;      // the Oxford format, nothing else
    if (W == 0 || H == 0 || W > (1u << 20) || H > (1u << 20)) return ROAM_E_ARG;
    if (rows) *rows = (int32_t)H;
    if (cols) *cols = (int32_t)W;
    if (out_stride == 0) out_stride = W;
    if (!out || out_stride < (int64_t)W || (int64_t)(H - 1) * out_stride + W > out_bytes) return ROAM_E_CAPACITY;
    pos += 12 + 13;
    if (decode_fast(png, nbytes, pos, W, H, out, out_stride, sc)) return ROAM_OK;
    // ---- zlib, a group of scanlines at a time (what the fast path refused: it may be a damaged file, it may be a stream shape it does not take)
    constexpr uint32_t G = 8;                                       // scanlines inflated per call
    const size_t LP = (size_t)W + 1;
    sc.line.resize(LP * G);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return ROAM_E_HIP;
    uint32_t y = 0;
    uint32_t want = H < G ? H : G;                                  // lines asked of the current inflate
    zs.next_out = sc.line.data();
    zs.avail_out = (uInt)(LP * want);
    bool ended = false, bad = false;
    auto flush_lines = [&](uint32_t nl) {
        if (!unfilter_lines(sc.line.data(), LP, y, nl, W, out, out_stride)) { bad = true; return; }
        y += nl;
    };
    while (pos + 12 <= nbytes && !bad) {
        const uint32_t len = be32(png + pos);
        const uint8_t *type = png + pos + 4, *data = png + pos + 8;
        if (pos + 12 + (int64_t)len > nbytes) { bad = true; break; }
        if (memcmp(type, "IDAT", 4) == 0 && !ended) {
            zs.next_in = const_cast<Bytef *>(data);
            zs.avail_in = len;
            while (zs.avail_in > 0 && y < H) {
                const int r = inflate(&zs, Z_NO_FLUSH);
                if (r != Z_OK && r != Z_STREAM_END) { bad = true; break; }
                if (zs.avail_out == 0) {                             // a whole group of filtered scanlines
                    flush_lines(want);
                    if (bad) break;
                    want = H - y < G ? H - y : G;
                    zs.next_out = sc.line.data();
                    zs.avail_out = (uInt)(LP * want);
                }
                if (r == Z_STREAM_END) { ended = true; break; }
            }
        } else if (memcmp(type, "IEND", 4) == 0)
            break;
        pos += 12 + (int64_t)len;
    }
    inflateEnd(&zs);
    return (bad || y != H) ? ROAM_E_ARG : ROAM_OK;
}

int32_t decode_file(const char *path, uint8_t *out, int64_t out_bytes, int64_t out_stride, int32_t *rows, int32_t *cols, Scratch &sc)
{
    FILE *f = fopen(path, "rb");
    if (!f) return ROAM_E_ARG;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n <= 0) { fclose(f); return ROAM_E_ARG; }
    sc.file.resize((size_t)n);
    const size_t got = fread(sc.file.data(), 1, (size_t)n, f);
    fclose(f);
    if (got != (size_t)n) return ROAM_E_ARG;
    return decode_gray8(sc.file.data(), n, out, out_bytes, out_stride, rows, cols, sc);
}

struct Job { std::string path; uint8_t *dst; int64_t dst_bytes, dst_stride, ticket; };
struct Done { int32_t status, rows, cols; };

}  // namespace

struct roam_png_pool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::deque<Job> jobs;
    std::unordered_map<int64_t, Done> done;
    bool stop = false;

    void worker()
    {
        Scratch sc;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;                            // stop, and nothing left to do
                j = std::move(jobs.front());
                jobs.pop_front();
            }
            Done d = {ROAM_OK, 0, 0};
            d.status = decode_file(j.path.c_str(), j.dst, j.dst_bytes, j.dst_stride, &d.rows, &d.cols, sc);
            {
                std::lock_guard<std::mutex> lk(mu);
                done[j.ticket] = d;
            }
            cv_done.notify_all();
        }
    }
};

extern "C" {

int32_t roam_png_decode_gray8(const uint8_t *png, int64_t png_bytes, uint8_t *out, int64_t out_bytes, int64_t out_stride,
                              int32_t *rows, int32_t *cols)
{
    static thread_local Scratch sc;                                 // (buffers of the calling thread, kept from call to call)
    return decode_gray8(png, png_bytes, out, out_bytes, out_stride, rows, cols, sc);
}

int32_t roam_png_decode_file(const char *path, uint8_t *out, int64_t out_bytes, int64_t out_stride, int32_t *rows, int32_t *cols)
{
    if (!path) return ROAM_E_ARG;
    static thread_local Scratch sc;
    return decode_file(path, out, out_bytes, out_stride, rows, cols, sc);
}

int32_t roam_png_pool_create(int32_t workers, roam_png_pool **out)
{
    if (!out || workers < 1 || workers > 1024) return ROAM_E_ARG;
    roam_png_pool *p = new roam_png_pool();
    try {
        for (int i = 0; i < workers; i++) p->threads.emplace_back([p] { p->worker(); });
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(p->mu);
            p->stop = true;
        }
        p->cv_job.notify_all();
        for (auto &t : p->threads) t.join();
        delete p;
        return ROAM_E_HIP;
    }
    *out = p;
    return ROAM_OK;
}

int32_t roam_png_pool_submit(roam_png_pool *p, const char *path, uint8_t *dst, int64_t dst_bytes, int64_t dst_stride, int64_t ticket)
{
    if (!p || !path || !dst) return ROAM_E_ARG;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->stop) return ROAM_E_STATE;
        p->jobs.push_back(Job{path, dst, dst_bytes, dst_stride, ticket});
    }
    p->cv_job.notify_one();
    return ROAM_OK;
}

int32_t roam_png_pool_wait(roam_png_pool *p, int64_t ticket, int32_t *rows, int32_t *cols)
{
    if (!p) return ROAM_E_ARG;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->done.count(ticket) != 0; });
    const Done d = p->done[ticket];
    p->done.erase(ticket);
    if (rows) *rows = d.rows;
    if (cols) *cols = d.cols;
    return d.status;
}

int32_t roam_png_pool_destroy(roam_png_pool *p)
{
    if (!p) return ROAM_E_ARG;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
        p->jobs.clear();                                             // jobs nobody will wait for
    }
    p->cv_job.notify_all();
    for (auto &t : p->threads) t.join();
    delete p;
    return ROAM_OK;
}

}  // extern "C"
