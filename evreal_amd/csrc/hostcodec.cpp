// Host-only entry points that expose pieces of the split arithmetic's host side to the CPU test-suite (include/evreal_hip.h):
// the weight packer the model and LPIPS builders run (conv.h pack_split_weights) and the multiply-high division constants of
// the launch plans (conv.h fastdiv_magic).  No GPU code.
#include "conv.h"

using namespace evr;

extern "C" int evr_split_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_split_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_split_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}

extern "C" int evr_fastdiv_magic(unsigned d, unsigned* mul, unsigned* shift) {
    EVR_REQUIRE(d >= 1 && mul && shift, "evr_fastdiv_magic: divisor must be >= 1");
    fastdiv_magic(d, mul, shift);
    return EVR_OK;
}

// The H2 format of the fp32-grade mode (conv.h): activations (fixed exponent H2_ACT_EXP), weights (per-tensor exponent), decode.
extern "C" int evr_h2_pack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_h2_pack: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    pack_h2_act(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_h2_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_h2_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_h2_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_h2_unpack(const float* src, float* dst, int64_t n, int exponent) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_h2_unpack: n = %lld must be a multiple of 16", (long long)n);
    unpack_h2(src, dst, (size_t)n, exponent);
    return EVR_OK;
}
extern "C" int evr_h2_act_exponent(void) { return H2_ACT_EXP; }

// The P6 format of the f16 + MX-fp6 mode (conv.h): activations, weights (swapped code order, per-tensor exponent), decode.
extern "C" int evr_p6_pack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_p6_pack: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    pack_p6_act(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_p6_pack_weights(const float* src, float* dst, int64_t n, int* exponent) {
    EVR_REQUIRE(src && dst && exponent && n >= 0 && n % 16 == 0, "evr_p6_pack_weights: n = %lld must be a multiple of 16", (long long)n);
    std::vector<float> w(src, src + n);
    *exponent = pack_p6_weights(w);
    memcpy(dst, w.data(), (size_t)n * sizeof(float));
    return EVR_OK;
}
extern "C" int evr_p6_unpack(const float* src, float* dst, int64_t n) {
    EVR_REQUIRE(src && dst && n >= 0 && n % 16 == 0, "evr_p6_unpack: n = %lld must be a multiple of 16", (long long)n);
    unpack_p6(src, dst, (size_t)n);
    return EVR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Native PNG writers (the reference writes one PNG per frame with cv2.imwrite, utils/eval_utils.py:80-84, and
// `save_images` is on in config/eval/std.json:9).  Pixels arrive as the uint8 the reference would hand to cv2 --
// round(clip(img) * 255), computed on the GPU -- and every file decodes to exactly those bytes; only the container is ours:
// 8-bit gray or RGB, filter type 1 (Sub) on every row and ONE zlib stream of level 1 with Z_RLE -- cv2.imwrite's own defaults -- or,
// level 0, unfiltered rows in stored blocks; CRCs from zlib.  A pool of C++ threads does the encoding and the file I/O with no GIL in sight: the Python pool
// of PIL writers it replaces took 26 % off the drop-in's frame rate at 8 sequences.
#include <zlib.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>

namespace {

struct PngJob { std::string path; std::string dir; std::vector<unsigned char> px; int H, W, ch; };

void put_be32(std::vector<unsigned char>& v, unsigned x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void put_chunk(std::vector<unsigned char>& out, const char type[4], const unsigned char* data, size_t n) {
    put_be32(out, (unsigned)n);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put_be32(out, (unsigned)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

// -> the whole file in memory; empty on a zlib failure
std::vector<unsigned char> png_encode(const PngJob& j, int level) {
    const size_t row = (size_t)j.W * j.ch;
    std::vector<unsigned char> raw((row + 1) * j.H);
    const int bpp = j.ch;
    for (int y = 0; y < j.H; ++y) {
        unsigned char* dst = &raw[(row + 1) * y];
        const unsigned char* src = &j.px[row * y];
        if (level > 0) {                                                         // filter type 1 (Sub): what cv2.imwrite's defaults use too
            dst[0] = 1;
            for (int i = 0; i < bpp; ++i) dst[1 + i] = src[i];
            for (size_t i = bpp; i < row; ++i) dst[1 + i] = (unsigned char)(src[i] - src[i - bpp]);
        } else {                                                                 // stored blocks: filter type 0 (None)
            dst[0] = 0;
            memcpy(dst + 1, src, row);
        }
    }
    z_stream zs; memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, level > 0 ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK) return {};
    std::vector<unsigned char> z(deflateBound(&zs, (uLong)raw.size()));
    zs.next_in = raw.data(); zs.avail_in = (uInt)raw.size(); zs.next_out = z.data(); zs.avail_out = (uInt)z.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t zn = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return {};
    std::vector<unsigned char> out;
    out.reserve(zn + 64);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (unsigned)j.W); put_be32(ihdr, (unsigned)j.H);
    ihdr.push_back(8); ihdr.push_back(j.ch == 3 ? 2 : 0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), zn);
    put_chunk(out, "IEND", nullptr, 0);
    return out;
}

}  // namespace

struct evr_png_pool {
    int level = 1;
    // At most `per_dir` writers work in one directory at a time: 16 threads creating files in ONE directory (one sequence at a time, the
    // reference's own loop) contend for its lock and slowed the whole call from 1.89 k to 1.45-1.77 k frames/s, while 8 sequences
    // (8 directories) need more than 4 writers to keep up with 4.4 k frames/s (round 6, EVREAL_PNG_PER_DIR)
    int per_dir = 4;
    std::map<std::string, int> active;
    std::vector<std::thread> workers;
    std::deque<PngJob> queue;
    std::mutex mu;
    std::condition_variable cv_job, cv_idle;
    int64_t submitted = 0, done = 0, failed = 0, written = 0;
    std::string first_error;
    bool stop = false;

    void run() {
        for (;;) {
            PngJob j;
            {
                std::unique_lock<std::mutex> lk(mu);
                std::deque<PngJob>::iterator it;
                cv_job.wait(lk, [&] {
                    for (it = queue.begin(); it != queue.end(); ++it) if (active[it->dir] < per_dir) return true;
                    return stop && queue.empty();
                });
                if (queue.empty()) return;      // (stop and nothing left)
                j = std::move(*it);
                queue.erase(it);
                ++active[j.dir];
            }
            std::string err;
            const std::vector<unsigned char> file = png_encode(j, level);
            if (file.empty()) err = "zlib failed on " + j.path;
            else {
                FILE* f = fopen(j.path.c_str(), "wb");
                if (!f) err = "cannot open " + j.path;
                else {
                    if (fwrite(file.data(), 1, file.size(), f) != file.size()) err = "short write to " + j.path;
                    if (fclose(f) != 0 && err.empty()) err = "cannot close " + j.path;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                ++done;
                if (--active[j.dir] == 0) active.erase(j.dir);
                if (err.empty()) ++written; else { if (!failed++) first_error = err; }
            }
            cv_idle.notify_all();
            cv_job.notify_all();      // (a directory slot is free again)
        }
    }
};

extern "C" int evr_png_pool_create(int n_threads, int level, evr_png_pool** out) {
    EVR_REQUIRE(out && n_threads >= 1 && n_threads <= 256 && level >= 0 && level <= 9, "evr_png_pool_create: 1..256 threads, zlib level 0..9");
    evr_png_pool* p = new evr_png_pool();
    p->level = level;
    if (const char* e = getenv("EVREAL_PNG_PER_DIR")) { const int v = atoi(e); if (v >= 1) p->per_dir = v; }
    for (int i = 0; i < n_threads; ++i) p->workers.emplace_back([p] { p->run(); });
    *out = p;
    return EVR_OK;
}

// frames [n, H, W] (channels == 1) or [n, H, W, 3] uint8 in HOST memory -> <folder>/frame_%010d.png (eval_utils.py:81) for the given
// indices.  The pixels are copied before the call returns; encoding and writing happen on the pool's threads.
// frame_stride: bytes from one frame to the next in `frames_host` (0: dense) -- a slot of a step-major [steps, slots, H, W] buffer.
extern "C" int evr_png_pool_submit(evr_png_pool* p, const char* folder, const int64_t* indices, int n, const unsigned char* frames_host,
                                   int H, int W, int channels, int64_t frame_stride) {
    EVR_REQUIRE(p && folder && indices && frames_host && n >= 0 && H >= 1 && W >= 1 && (channels == 1 || channels == 3),
                "evr_png_pool_submit: bad argument (channels must be 1 or 3)");
    const size_t per = (size_t)H * W * channels;
    const size_t step = frame_stride > 0 ? (size_t)frame_stride : per;
    EVR_REQUIRE(step >= per, "evr_png_pool_submit: frame_stride smaller than a frame");
    std::vector<PngJob> jobs(n);
    for (int i = 0; i < n; ++i) {
        char name[40];
        snprintf(name, sizeof(name), "/frame_%010lld.png", (long long)indices[i]);
        jobs[i].path = std::string(folder) + name;
        jobs[i].dir = folder;
        jobs[i].px.assign(frames_host + step * i, frames_host + step * i + per);
        jobs[i].H = H; jobs[i].W = W; jobs[i].ch = channels;
    }
    {
        std::lock_guard<std::mutex> lk(p->mu);
        for (auto& j : jobs) p->queue.push_back(std::move(j));
        p->submitted += n;
    }
    p->cv_job.notify_all();
    return EVR_OK;
}

// Blocks until every frame submitted so far is on disk.  *n_written: files written since the pool was created; returns EVR_ERR_INVALID
// (message = the first failure) when any frame since the last wait could not be written.
extern "C" int evr_png_pool_wait(evr_png_pool* p, int64_t* n_written) {
    EVR_REQUIRE(p, "evr_png_pool_wait: null pool");
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_idle.wait(lk, [&] { return p->done == p->submitted; });
    if (n_written) *n_written = p->written;
    if (p->failed) {
        const std::string msg = p->first_error;
        const long long nf = (long long)p->failed;
        p->failed = 0; p->first_error.clear();
        lk.unlock();
        set_error("evr_png_pool_wait: %lld frame(s) not written; first: %s", nf, msg.c_str());
        return EVR_ERR_INVALID;
    }
    return EVR_OK;
}

extern "C" int evr_png_pool_destroy(evr_png_pool* p) {
    if (!p) return EVR_OK;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_job.notify_all();
    for (auto& t : p->workers) t.join();       // (drains the queue first: run() leaves only when it is empty)
    delete p;
    return EVR_OK;
}
