/*
 * corpus_main.cpp -- `srla_corpus`: the encode half of the reference's command line tool for MANY files
 * (tools/srla_codec/srla_codec.c:75-158 runs once per file; its WAV reader is libs/wav/src/wav.c:136-282, :479-556),
 * native and multi-threaded, on top of the C ABI of libsrla_mi355x.so:
 *
 *   reader threads   read() a .wav into pinned memory and parse its header as WAV_CreateFromFile does -- nothing else: the data
 *                    chunk goes to the device as it is (SRLAMI355X_EncodeBatchPcm de-interleaves, widens and OR-reduces it
 *                    there), so this read is the only time the host touches a sample.  --host-deinterleave: mmap the file
 *                    and de-interleave to planar int32 in pinned memory on the host instead (SRLAMI355X_EncodeBatchEx)
 *   main thread      gathers the loaded files of one format into batches (windows of different files share the device
 *                    jobs); one encoder handle per format, kept for the whole corpus
 *   writer threads   write <out>/<relative name>.srl, optionally hashing the streams for the manifest
 *
 * Same options, defaults and output-buffer rule (2 x the input file size) as `srla -e`.  One process per GPU: with
 * RANK / WORLD_SIZE / LOCAL_RANK in the environment (torchrun, or --rank / --world) every rank encodes the files the
 * deterministic longest-first assignment gives it (the same one as srla_amd/corpus.py) -- no communication at all.
 *
 *   srla_corpus -e [-m 4] [-B 4096] [-V 1] [-L 4] [-P 0] [--manifest FILE [--sha256]] [--batch-samples N] [--readers N] [--writers N] [--host-deinterleave] IN_DIR OUT_DIR
 */
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/srla_mi355x.h"

namespace fs = std::filesystem;

namespace {

/* file names and error texts inside the manifest's JSON strings */
std::string json_escape(const std::string &in)
{
    std::string out;
    for (unsigned char c : in) {
        if (c == '"' || c == '\\') { out += '\\'; out += (char)c; }
        else if (c < 0x20) { char b[8]; snprintf(b, sizeof(b), "\\u%04x", c); out += b; }
        else out += (char)c;
    }
    return out;
}

/* ---- SHA-256 (FIPS 180-4), for the manifest ------------------------------------------------------------- */
struct Sha256 {
    uint32_t h[8]; uint64_t len = 0; uint8_t buf[64]; size_t fill = 0;
    Sha256()
    {
        static const uint32_t init[8] = { 0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u };
        memcpy(h, init, sizeof(h));
    }
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const uint8_t *p)
    {
        static const uint32_t k[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
            0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
            0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
            0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u };
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + k[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const uint8_t *p, size_t n)
    {
        len += n;
        if (fill) { const size_t t = std::min(n, 64 - fill); memcpy(buf + fill, p, t); fill += t; p += t; n -= t; if (fill == 64) { block(buf); fill = 0; } }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) { memcpy(buf, p, n); fill = n; }
    }
    std::string hex()
    {
        const uint64_t bits = len * 8;
        const uint8_t pad = 0x80; update(&pad, 1);
        const uint8_t zero = 0; while (fill != 56) update(&zero, 1);
        uint8_t lb[8]; for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
        update(lb, 8);
        char out[65];
        for (int i = 0; i < 8; i++) snprintf(out + 8 * i, 9, "%08x", h[i]);
        return std::string(out, 64);
    }
};

/* ---- WAV ------------------------------------------------------------------------------------------------- */
/* Pinned buffers for the planes (SRLAMI355X_AllocHost: the device reads them by DMA, no staging copy in the library), kept
 * on a free list: page-locking memory is slow, files of a corpus are of similar size. */
class PinnedPool {
public:
    void *take(size_t bytes, size_t *cap)
    {
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].second >= bytes && free_[i].second <= 2 * bytes + (1u << 20)) {
                    const auto e = free_[i]; free_.erase(free_.begin() + (long)i); *cap = e.second; return e.first;
                }
        }
        const size_t want = bytes + bytes / 8 + 4096;
        void *p = SRLAMI355X_AllocHost(want);
        *cap = p ? want : 0;
        return p;
    }
    void give(void *p, size_t cap) { if (p) { std::lock_guard<std::mutex> l(m_); free_.emplace_back(p, cap); } }
private:
    std::mutex m_;
    std::vector<std::pair<void *, size_t>> free_;
};
PinnedPool g_pinned;

struct Pcm {
    std::string path, rel;
    uint32_t nch = 0, bps = 0, rate = 0, n = 0;
    uint64_t file_size = 0;
    /* default: the file as read, in pinned memory; `frames` points at its data chunk (handed to SRLAMI355X_EncodeBatchPcm: the
     * device de-interleaves).  --host-deinterleave: planar [nch][n] samples, pinned, and their OR (SRLAMI355X_EncodeBatchEx) */
    uint8_t *raw = nullptr;
    size_t raw_cap = 0;
    const uint8_t *frames = nullptr;
    int32_t *samples = nullptr;
    size_t samples_cap = 0;
    uint32_t sample_or = 0;                /* OR of every sample (the offset left shift comes from it, srla_utility.c:177) */
    std::vector<const int32_t *> planes;
    std::string error;
    ~Pcm() { g_pinned.give(samples, samples_cap); g_pinned.give(raw, raw_cap); }
};

uint32_t rd16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* What WAV_CreateFromFile accepts (libs/wav/src/wav.c:136-282): RIFF/WAVE whose first chunk is `fmt ` of 16 bytes
 * (format tag 1) or 40 bytes (tag 0xFFFE with a 22-byte extension); chunks between `fmt ` and `data` are skipped by
 * their raw size; 8-bit samples are offset binary, 16 / 24 / 32-bit little endian two's complement (:479-556); the
 * samples come out planar and sign-extended, not left-justified (:848-852). */
bool g_host_deinterleave = false;

bool load_wav(Pcm &f)
{
    const int fd = open(f.path.c_str(), O_RDONLY);
    if (fd < 0) { f.error = "cannot open"; return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 44) { close(fd); f.error = "not a RIFF/WAVE file"; return false; }
    f.file_size = (uint64_t)sb.st_size;
    const size_t size = (size_t)sb.st_size;
    const uint8_t *b = nullptr;
    if (g_host_deinterleave) {
        b = static_cast<const uint8_t *>(mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0));
        close(fd);
        if (b == MAP_FAILED) { f.error = "mmap failed"; return false; }
    } else {
        /* the whole file into pinned memory: this read() is the only time the host touches the samples */
        f.raw = static_cast<uint8_t *>(g_pinned.take(size + 64, &f.raw_cap));
        if (f.raw == nullptr) { close(fd); f.error = "out of (pinned) memory"; return false; }
        size_t got = 0;
        while (got < size) {
            const ssize_t r = read(fd, f.raw + got, size - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        close(fd);
        if (got != size) { f.error = "read failed"; return false; }
        b = f.raw;
    }
    auto done = [&](const char *err) { if (g_host_deinterleave) munmap(const_cast<uint8_t *>(b), size); if (err) f.error = err; return err == nullptr; };
    if (memcmp(b, "RIFF", 4) != 0 || memcmp(b + 8, "WAVE", 4) != 0) return done("not a RIFF/WAVE file");
    size_t pos = 12;
    if (memcmp(b + pos, "fmt ", 4) != 0) return done("'fmt ' chunk expected right after 'WAVE'");
    const uint32_t fmt_size = rd32(b + pos + 4);
    if (fmt_size != 16 && fmt_size != 40) return done("unsupported fmt chunk size");
    if (pos + 8 + fmt_size > size) return done("truncated fmt chunk");
    const uint32_t tag = rd16(b + pos + 8);
    f.nch = rd16(b + pos + 10); f.rate = rd32(b + pos + 12); f.bps = rd16(b + pos + 22);
    if ((fmt_size == 16 && tag != 1) || (fmt_size == 40 && tag != 0xFFFEu)) return done("unsupported format tag");
    if (fmt_size == 40 && rd16(b + pos + 24) != 22) return done("bad WAVEFORMATEXTENSIBLE extension size");
    pos += 8 + fmt_size;
    uint32_t data_size = 0;
    for (;;) {
        if (pos + 8 > size) return done("no 'data' chunk");
        const uint32_t csz = rd32(b + pos + 4);
        const bool is_data = memcmp(b + pos, "data", 4) == 0;
        pos += 8;
        if (is_data) { data_size = csz; break; }
        pos += csz;
    }
    if ((f.bps != 8 && f.bps != 16 && f.bps != 24 && f.bps != 32) || f.nch == 0) return done("unsupported sample format");
    const uint32_t bytes_ps = f.bps / 8, frame = bytes_ps * f.nch;
    if (data_size % frame) return done("data size is not a whole number of sample frames");
    if (pos + data_size > size) return done("truncated data chunk");
    f.n = data_size / frame;
    if (f.n == 0) return done("no samples");
    if (!g_host_deinterleave) { f.frames = b + pos; return done(nullptr); }
    f.samples = static_cast<int32_t *>(g_pinned.take((size_t)f.nch * f.n * 4 + 64, &f.samples_cap));
    if (f.samples == nullptr) return done("out of (pinned) memory");
    const uint8_t *d = b + pos;
    const uint32_t nch = f.nch, n = f.n;
    uint32_t m = 0;
    if (f.bps == 16 && nch == 2) {
        int32_t *l = f.samples, *r = f.samples + n;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t v = rd32(d + 4 * (size_t)i);
            const int32_t a = (int16_t)(v & 0xFFFFu), c = (int16_t)(v >> 16);
            l[i] = a; r[i] = c; m |= (uint32_t)a | (uint32_t)c;
        }
    } else {
        for (uint32_t ch = 0; ch < nch; ch++) {
            int32_t *o = f.samples + (size_t)ch * n;
            const uint8_t *s = d + (size_t)ch * bytes_ps;
            switch (f.bps) {
            case 8:  for (uint32_t i = 0; i < n; i++) { o[i] = (int32_t)s[(size_t)i * frame] - 128; m |= (uint32_t)o[i]; } break;
            case 16: for (uint32_t i = 0; i < n; i++) { o[i] = (int16_t)rd16(s + (size_t)i * frame); m |= (uint32_t)o[i]; } break;
            case 24: for (uint32_t i = 0; i < n; i++) { const uint8_t *p = s + (size_t)i * frame; o[i] = ((int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24))) >> 8; m |= (uint32_t)o[i]; } break;
            default: for (uint32_t i = 0; i < n; i++) { o[i] = (int32_t)rd32(s + (size_t)i * frame); m |= (uint32_t)o[i]; } break;
            }
        }
    }
    f.sample_or = m;
    for (uint32_t ch = 0; ch < nch; ch++) f.planes.push_back(f.samples + (size_t)ch * n);
    return done(nullptr);
}

/* ---- a small blocking queue ------------------------------------------------------------------------------ */
template <typename T>
class Queue {
public:
    void push(T v) { { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    bool pop(T &out)
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front()); q_.pop_front();
        return true;
    }
    /* without waiting: 1 = got one, 0 = nothing there right now, -1 = closed and drained */
    int try_pop(T &out)
    {
        std::lock_guard<std::mutex> l(m_);
        if (q_.empty()) return closed_ ? -1 : 0;
        out = std::move(q_.front()); q_.pop_front();
        return 1;
    }
    void close() { { std::lock_guard<std::mutex> l(m_); closed_ = true; } cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_; bool closed_ = false;
};

struct Encoded {
    std::unique_ptr<Pcm> pcm;
    /* Pinned and recycled like the input buffers: the device stores the blocks straight into it.  (A fresh pageable buffer per
     * file, as srla_codec.c:125-129 allocates it, has to be faulted in and page-locked by the library call by call: 230 MB per
     * batch of four 300 s files, half of the batch's time.)  A stream cannot outgrow its samples by more than the block headers. */
    uint8_t *data = nullptr;
    size_t cap = 0, pool_cap = 0;
    uint32_t size = 0;
    ~Encoded() { g_pinned.give(data, pool_cap); }
    SRLAApiResult rc = SRLA_APIRESULT_OK;
};

struct Options {
    int mode = 4, lookahead = 4, divisions = 1, ltp = 0;
    uint32_t max_block = 4096;
    std::string in_dir, out_dir, manifest;
    uint64_t batch_samples = 48ull << 20;   /* sample frames per EncodeBatch call (about a dozen device jobs) */
    unsigned readers = 6, writers = 2;
    bool sha = false, verbose = false;
    int rank = 0, world = 1, local_rank = 0;
};

int usage()
{
    fprintf(stderr, "usage: srla_corpus -e [-m 4] [-B 4096] [-V 1] [-L 4] [-P 0] [--manifest FILE] [--sha256] [--batch-samples N] [--readers N] [--writers N] [--host-deinterleave] [--verbose] IN_DIR OUT_DIR\n");
    return 1;
}

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    if (const char *e = getenv("RANK")) o.rank = atoi(e);
    if (const char *e = getenv("WORLD_SIZE")) o.world = std::max(1, atoi(e));
    if (const char *e = getenv("LOCAL_RANK")) o.local_rank = atoi(e); else o.local_rank = o.rank;
    bool encode = false;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : "0"; };
        if (a == "-e" || a == "--encode") encode = true;
        else if (a == "-m" || a == "--mode") o.mode = atoi(val());
        else if (a == "-B" || a == "--max-block-size") o.max_block = (uint32_t)atoi(val());
        else if (a == "-V" || a == "--variable-block-divisions") o.divisions = atoi(val());
        else if (a == "-L" || a == "--lookahead-sample-factor") o.lookahead = atoi(val());
        else if (a == "-P" || a == "--long-term-prediction") o.ltp = atoi(val());
        else if (a == "--manifest") o.manifest = val();
        else if (a == "--batch-samples") o.batch_samples = strtoull(val(), nullptr, 10);
        else if (a == "--readers") o.readers = (unsigned)std::max(1, atoi(val()));
        else if (a == "--writers") o.writers = (unsigned)std::max(1, atoi(val()));
        else if (a == "--sha256") o.sha = true;
        else if (a == "--verbose") o.verbose = true;
        else if (a == "--host-deinterleave") g_host_deinterleave = true;
        else if (a == "--rank") o.rank = atoi(val());
        else if (a == "--world") o.world = std::max(1, atoi(val()));
        else if (a == "--device") o.local_rank = atoi(val());
        else if (!a.empty() && a[0] == '-') return usage();
        else pos.push_back(a);
    }
    if (!encode || pos.size() != 2) return usage();
    if (o.mode < 0 || o.mode >= SRLA_NUM_PARAMETER_PRESETS) { fprintf(stderr, "srla_corpus: encode preset number is out of range.\n"); return 1; }
    o.in_dir = pos[0]; o.out_dir = pos[1];
    if (SRLAMI355X_SetDevice(o.local_rank) != 0) {
        fprintf(stderr, "srla_corpus: no MI355X for local rank %d (there is no CPU fallback)\n", o.local_rank);
        return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();

    /* the corpus, sorted; owner by longest-processing-time greedy on the file sizes (ties by index), like srla_amd/corpus.py */
    std::vector<std::string> paths;
    for (auto it = fs::recursive_directory_iterator(o.in_dir); it != fs::recursive_directory_iterator(); ++it) {
        if (!it->is_regular_file()) continue;
        std::string ext = it->path().extension().string();
        std::transform(ext.begin(), ext.end(), ext.begin(), ::tolower);
        if (ext == ".wav") paths.push_back(it->path().string());
    }
    std::sort(paths.begin(), paths.end());
    std::vector<uint64_t> sizes(paths.size());
    for (size_t i = 0; i < paths.size(); i++) sizes[i] = std::max<uint64_t>(1, (uint64_t)fs::file_size(paths[i]));
    std::vector<int> owner(paths.size(), 0);
    {
        std::vector<size_t> order(paths.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return sizes[a] > sizes[b]; });
        std::vector<uint64_t> load((size_t)o.world, 0);
        for (size_t i : order) {
            size_t r = 0;
            for (size_t k = 1; k < load.size(); k++) if (load[k] < load[r]) r = k;
            owner[i] = (int)r; load[r] += sizes[i];
        }
    }
    std::vector<std::string> mine;
    for (size_t i = 0; i < paths.size(); i++) if (owner[i] == o.rank) mine.push_back(paths[i]);

    /* readers */
    Queue<std::unique_ptr<Pcm>> loaded;
    std::atomic<size_t> next{ 0 };
    std::atomic<int64_t> in_flight_bytes{ 0 };
    const int64_t flight_cap = (int64_t)std::max<uint64_t>(o.batch_samples * 8 * 3, 1ull << 30);   /* planar int32 stereo: 8 B per frame; ~3 batches ahead */
    std::mutex flight_m; std::condition_variable flight_cv;
    std::vector<std::thread> readers;
    std::atomic<unsigned> readers_left{ o.readers };
    for (unsigned t = 0; t < o.readers; t++)
        readers.emplace_back([&] {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= mine.size()) break;
                {
                    std::unique_lock<std::mutex> l(flight_m);
                    flight_cv.wait(l, [&] { return in_flight_bytes.load() < flight_cap; });
                }
                std::unique_ptr<Pcm> f(new Pcm());
                f->path = mine[i];
                f->rel = fs::relative(fs::path(mine[i]), fs::path(o.in_dir)).string();
                (void)load_wav(*f);
                in_flight_bytes.fetch_add((int64_t)f->nch * f->n * 4);
                loaded.push(std::move(f));
            }
            if (readers_left.fetch_sub(1) == 1) loaded.close();
        });

    /* writer */
    Queue<std::unique_ptr<Encoded>> finished;
    struct Entry { std::string rel; uint64_t in_bytes, samples; uint32_t out_bytes; std::string sha, error; };
    std::vector<Entry> manifest;
    std::mutex manifest_m;
    std::vector<std::thread> writers;
    std::atomic<int> write_failures{ 0 };        /* feeds the exit status like every other failure */
    for (unsigned t = 0; t < o.writers; t++)
        writers.emplace_back([&] {
            std::unique_ptr<Encoded> e;
            while (finished.pop(e)) {
                Entry en{ e->pcm->rel, e->pcm->file_size, e->pcm->n, 0, "", e->pcm->error };
                if (en.error.empty() && e->rc != SRLA_APIRESULT_OK) en.error = "encode failed: " + std::to_string((int)e->rc);
                if (en.error.empty()) {
                    fs::path out = fs::path(o.out_dir) / fs::path(e->pcm->rel);
                    out.replace_extension(".srl");
                    std::error_code ec;
                    fs::create_directories(out.parent_path(), ec);
                    FILE *fp = fopen(out.string().c_str(), "wb");
                    if (!fp || fwrite(e->data, 1, e->size, fp) != e->size) en.error = "cannot write " + out.string();
                    if (fp && fclose(fp) != 0 && en.error.empty()) en.error = "cannot write " + out.string();
                    if (!en.error.empty()) { write_failures.fetch_add(1); if (fp) fs::remove(out, ec); }   /* no partial .srl left behind */
                    else en.out_bytes = e->size;
                    if (o.sha) { Sha256 s; s.update(e->data, e->size); en.sha = s.hex(); }
                }
                const int64_t bytes = (int64_t)e->pcm->nch * e->pcm->n * 4;
                e.reset();
                in_flight_bytes.fetch_sub(bytes);
                { std::lock_guard<std::mutex> l(flight_m); }
                flight_cv.notify_all();
                std::lock_guard<std::mutex> l(manifest_m);
                manifest.push_back(en);
            }
        });

    /* main: batches per format */
    struct Key { uint32_t nch, bps, rate; bool operator<(const Key &k) const { return std::tie(nch, bps, rate) < std::tie(k.nch, k.bps, k.rate); } };
    std::map<Key, SRLAEncoder *> encoders;
    std::map<Key, std::vector<std::unique_ptr<Pcm>>> pending;
    std::map<Key, uint64_t> pending_samples;
    int failures = 0;
    auto flush = [&](const Key &k) {
        std::vector<std::unique_ptr<Pcm>> files = std::move(pending[k]);
        pending[k].clear(); pending_samples[k] = 0;
        if (files.empty()) return;
        SRLAEncoder *&enc = encoders[k];
        if (enc == nullptr) {
            SRLAEncoderConfig cfg;                                            /* srla_codec.c:91-95 */
            cfg.max_num_channels = 8;
            cfg.min_num_samples_per_block = o.max_block >> o.divisions;
            cfg.max_num_samples_per_block = o.max_block;
            cfg.max_num_lookahead_samples = (uint32_t)o.lookahead * o.max_block;
            cfg.max_num_parameters = 255;
            enc = SRLAEncoder_Create(&cfg, nullptr, 0);
            SRLAEncodeParameter p;                                            /* srla_codec.c:104-116 */
            memset(&p, 0, sizeof(p));
            p.num_channels = (uint16_t)k.nch; p.bits_per_sample = (uint16_t)k.bps; p.sampling_rate = k.rate;
            p.min_num_samples_per_block = cfg.min_num_samples_per_block; p.max_num_samples_per_block = cfg.max_num_samples_per_block;
            p.num_lookahead_samples = cfg.max_num_lookahead_samples; p.ltp_order = (uint32_t)o.ltp;
            p.num_svr_filter_learning_iteration = 0; p.preset = (uint8_t)o.mode;
            const SRLAApiResult rc = enc ? SRLAEncoder_SetEncodeParameter(enc, &p) : SRLA_APIRESULT_NG;
            if (rc != SRLA_APIRESULT_OK) {
                fprintf(stderr, "srla_corpus: failed to set encode parameter for %u ch / %u bit / %u Hz: %d\n", k.nch, k.bps, k.rate, (int)rc);
                if (enc) { SRLAEncoder_Destroy(enc); enc = nullptr; }
            }
        }
        const uint32_t ns = (uint32_t)files.size();
        std::vector<std::unique_ptr<Encoded>> outs(ns);
        std::vector<const int32_t *const *> inputs(ns);
        std::vector<uint32_t> nsmp(ns), caps(ns), sizes_out(ns, 0), ors(ns);
        std::vector<uint8_t *> datas(ns);
        std::vector<SRLAApiResult> res(ns, SRLA_APIRESULT_NG);
        for (uint32_t i = 0; i < ns; i++) {
            outs[i].reset(new Encoded());
            /* the true bound: stream header + an 11-byte header per block (a RAW block carries its samples as they are,
             * srla_encoder.c:823-852) + the PCM itself; the reference's tool allocates twice the file (srla_codec.c:125-129) */
            const uint64_t min_block = std::max<uint32_t>(1u, o.max_block >> o.divisions);
            outs[i]->cap = (size_t)((uint64_t)files[i]->file_size + ((uint64_t)files[i]->n / min_block + 2u) * 16u + 65536u);
            outs[i]->data = static_cast<uint8_t *>(g_pinned.take(outs[i]->cap, &outs[i]->pool_cap));
            if (outs[i]->data == nullptr) outs[i]->cap = 0;                    /* out of pinned memory: the call refuses the batch */
            inputs[i] = files[i]->planes.data(); nsmp[i] = files[i]->n; ors[i] = files[i]->sample_or;
            datas[i] = outs[i]->data; caps[i] = (uint32_t)std::min<uint64_t>(outs[i]->cap, 0xFFFFFFFFull);
        }
        SRLAApiResult rc = SRLA_APIRESULT_NG;
        const auto tb = std::chrono::steady_clock::now();
        if (enc && !g_host_deinterleave) {
            std::vector<const void *> frames(ns);
            for (uint32_t i = 0; i < ns; i++) frames[i] = files[i]->frames;
            rc = SRLAMI355X_EncodeBatchPcm(enc, ns, frames.data(), nsmp.data(), k.bps / 8u, datas.data(), caps.data(), sizes_out.data(), res.data());
        } else if (enc) rc = SRLAMI355X_EncodeBatchEx(enc, ns, inputs.data(), nsmp.data(), ors.data(), datas.data(), caps.data(), sizes_out.data(), res.data());
        if (o.verbose) {
            uint64_t tot = 0; for (uint32_t i = 0; i < ns; i++) tot += nsmp[i];
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
            fprintf(stderr, "[srla_corpus] batch of %u files, %llu samples: %.2f ms (%.0f Msamples/s), at %.3f s\n", ns, (unsigned long long)tot, ms, (double)tot / ms / 1e3,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (rc != SRLA_APIRESULT_OK && rc != SRLA_APIRESULT_INSUFFICIENT_BUFFER) for (auto &r : res) r = rc;
        for (uint32_t i = 0; i < ns; i++) {
            outs[i]->size = sizes_out[i]; outs[i]->rc = res[i];
            if (res[i] != SRLA_APIRESULT_OK) failures++;
            outs[i]->pcm = std::move(files[i]);
            finished.push(std::move(outs[i]));
        }
    };
    {
        std::unique_ptr<Pcm> f;
        for (;;) {
            /* a batch goes out when it is full -- or when the readers have nothing more ready: the GPU should not wait for
             * a batch to fill up while files are still being read */
            const int got = loaded.try_pop(f);
            if (got < 0) break;
            if (got == 0) {
                const Key *fullest = nullptr; uint64_t most = 0;
                for (auto &kv : pending_samples) if (kv.second > most) { most = kv.second; fullest = &kv.first; }
                if (fullest) { const Key k = *fullest; flush(k); continue; }
                if (!loaded.pop(f)) break;
            }
            if (!f->error.empty()) {
                fprintf(stderr, "srla_corpus: %s: %s\n", f->path.c_str(), f->error.c_str());
                failures++;
                std::unique_ptr<Encoded> e(new Encoded());
                e->pcm = std::move(f); e->rc = SRLA_APIRESULT_NG;
                finished.push(std::move(e));
                continue;
            }
            const Key k{ f->nch, f->bps, f->rate };
            pending_samples[k] += f->n;
            pending[k].push_back(std::move(f));
            if (pending_samples[k] >= o.batch_samples) flush(k);
        }
        std::vector<Key> keys;
        for (auto &kv : pending) keys.push_back(kv.first);
        for (const Key &k : keys) flush(k);
    }
    auto stamp = [&](const char *what) {
        if (o.verbose) fprintf(stderr, "[srla_corpus] %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    };
    stamp("all batches encoded");
    for (auto &t : readers) t.join();
    finished.close();
    for (auto &t : writers) t.join();
    stamp("all files written");

    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::sort(manifest.begin(), manifest.end(), [](const Entry &a, const Entry &b) { return a.rel < b.rel; });
    uint64_t tin = 0, tout = 0, tsmp = 0;
    for (const Entry &e : manifest) { tin += e.in_bytes; tout += e.out_bytes; tsmp += e.samples; }
    printf("finished: %zu files, %llu -> %llu (%6.2f %%) in %.3f s, %.1f Msamples/s (rank %d of %d)\n", manifest.size(), (unsigned long long)tin,
           (unsigned long long)tout, tin ? 100.0 * (double)tout / (double)tin : 0.0, dt, (double)tsmp / dt / 1e6, o.rank, o.world);
    if (!o.manifest.empty()) {
        FILE *fp = fopen(o.manifest.c_str(), "w");
        if (fp) {
            fprintf(fp, "{\"rank\": %d, \"world\": %d, \"seconds\": %.6f, \"samples\": %llu, \"files\": [\n", o.rank, o.world, dt, (unsigned long long)tsmp);
            for (size_t i = 0; i < manifest.size(); i++) {
                const Entry &e = manifest[i];
                fprintf(fp, "  {\"name\": \"%s\", \"samples\": %llu, \"in_bytes\": %llu, \"bytes\": %u, \"sha256\": \"%s\", \"error\": \"%s\"}%s\n", json_escape(e.rel).c_str(),
                        (unsigned long long)e.samples, (unsigned long long)e.in_bytes, e.out_bytes, e.sha.c_str(), json_escape(e.error).c_str(), i + 1 < manifest.size() ? "," : "");
            }
            fprintf(fp, "]}\n");
            fclose(fp);
        }
    }
    /* Everything is on disk.  Unpinning a few GB of buffers and tearing the device context down takes a quarter of a second
     * that buys nothing at the end of a process: the operating system releases it all. */
    fflush(stdout); fflush(stderr);
    _exit((failures || write_failures.load()) ? 1 : 0);
}
