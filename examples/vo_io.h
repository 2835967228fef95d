// vo_io.h -- image input for the C++ host programs (vo_run, vo_seq_run, vo_multi_gpu): what the reference's
// loadImageLeft / loadImageRight do (utils.cpp:172-190: imread(IMREAD_COLOR) of <dir>/image_{0,1}/%06d.png followed by
// cvtColor(BGR2GRAY)) without OpenCV: a PNG reader on zlib alone (8-bit gray or RGB[A], non-interlaced -- what KITTI
// ships; an RGB file goes through the same integer BT.601 weights cvtColor uses, which is the identity for KITTI's
// R = G = B), binary PGM as the no-decoder alternative, and a small thread pool so that the frames of step k + 1 are
// decoded while step k runs on the GPU (the ingest half of SURVEY.md section 8 row f2).
#pragma once

#include <zlib.h>

#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

namespace voio {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px; // 8-bit gray, row-major, stride = w
};

inline bool read_file(const std::string &path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f)
        return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    return ok;
}

inline bool decode_pgm(const std::vector<uint8_t> &buf, Image &im)
{
    // "P5" <w> <h> <maxval> <single whitespace> <raster>; '#' comments allowed between the header tokens
    size_t p = 0;
    auto token = [&](std::string &t) {
        for (;;) {
            while (p < buf.size() && (buf[p] == ' ' || buf[p] == '\n' || buf[p] == '\r' || buf[p] == '\t'))
                p++;
            if (p < buf.size() && buf[p] == '#')
                while (p < buf.size() && buf[p] != '\n')
                    p++;
            else
                break;
        }
        t.clear();
        while (p < buf.size() && buf[p] > ' ')
            t.push_back((char)buf[p++]);
        return !t.empty();
    };
    std::string t;
    if (!token(t) || t != "P5")
        return false;
    int v[3];
    for (int &x : v) {
        if (!token(t))
            return false;
        x = atoi(t.c_str());
    }
    if (v[0] < 1 || v[1] < 1 || v[2] != 255)
        return false;
    p++; // the single whitespace after maxval
    const size_t n = (size_t)v[0] * v[1];
    if (p + n > buf.size())
        return false;
    im.w = v[0];
    im.h = v[1];
    im.px.assign(buf.begin() + p, buf.begin() + p + n);
    return true;
}

inline uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// PNG (ISO/IEC 15948): signature, IHDR, concatenated IDAT payloads = one zlib stream of filtered scanlines.
inline bool decode_png(const std::vector<uint8_t> &buf, Image &im)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0)
        return false;
    size_t p = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> z;
    while (p + 12 <= buf.size()) {
        const uint32_t len = be32(&buf[p]);
        const uint8_t *type = &buf[p + 4], *data = &buf[p + 8];
        if (p + 12 + (size_t)len > buf.size())
            return false;
        if (!memcmp(type, "IHDR", 4) && len == 13) {
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!memcmp(type, "IDAT", 4)) {
            z.insert(z.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        p += 12 + (size_t)len;
    }
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (w < 1 || h < 1 || w > 16384 || h > 16384 || depth != 8 || ch == 0 || interlace != 0 || z.empty())
        return false; // palette / 16-bit / interlaced files: convert them first (tools/kitti_to_pgm.py)
    const size_t row = (size_t)w * ch;
    std::vector<uint8_t> raw((row + 1) * (size_t)h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, z.data(), (uLong)z.size()) != Z_OK || out_len != raw.size())
        return false;
    // undo the per-scanline filters in place (bpp = bytes per complete pixel)
    const int bpp = ch;
    std::vector<uint8_t> zero(row, 0);
    for (int y = 0; y < h; y++) {
        uint8_t *cur = &raw[(row + 1) * (size_t)y + 1];
        const uint8_t *up = y ? &raw[(row + 1) * (size_t)(y - 1) + 1] : zero.data();
        switch (raw[(row + 1) * (size_t)y]) {
        case 0:
            break;
        case 1:
            for (size_t i = bpp; i < row; i++)
                cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < row; i++)
                cur[i] = (uint8_t)(cur[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < row; i++)
                cur[i] = (uint8_t)(cur[i] + (((i >= (size_t)bpp ? cur[i - bpp] : 0) + up[i]) >> 1));
            break;
        case 4:
            for (size_t i = 0; i < row; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
                const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                cur[i] = (uint8_t)(cur[i] + (pa <= pb && pa <= pc ? a : pb <= pc ? b : c));
            }
            break;
        default:
            return false;
        }
    }
    im.w = w;
    im.h = h;
    im.px.resize((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *s = &raw[(row + 1) * (size_t)y + 1];
        uint8_t *d = &im.px[(size_t)w * y];
        if (ch <= 2) { // gray (+ alpha)
            for (int x = 0; x < w; x++)
                d[x] = s[(size_t)x * ch];
        } else { // RGB(A) -> Y with cvtColor's 14-bit fixed-point weights (R 4899, G 9617, B 1868)
            for (int x = 0; x < w; x++) {
                const uint8_t *q = s + (size_t)x * ch;
                d[x] = (uint8_t)((q[0] * 4899 + q[1] * 9617 + q[2] * 1868 + (1 << 13)) >> 14);
            }
        }
    }
    return true;
}

// <dir>/image_<cam>/%06d.png, else .pgm
inline bool read_frame(const std::string &dir, int cam, int id, Image &im)
{
    char name[64];
    std::vector<uint8_t> buf;
    snprintf(name, sizeof(name), "/image_%d/%06d.png", cam, id);
    if (read_file(dir + name, buf))
        return decode_png(buf, im);
    snprintf(name, sizeof(name), "/image_%d/%06d.pgm", cam, id);
    return read_file(dir + name, buf) && decode_pgm(buf, im);
}

class ThreadPool {
  public:
    explicit ThreadPool(int n)
    {
        for (int i = 0; i < (n < 1 ? 1 : n); i++)
            workers_.emplace_back([this] {
                for (;;) {
                    std::function<void()> job;
                    {
                        std::unique_lock<std::mutex> lk(mu_);
                        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                        if (stop_ && jobs_.empty())
                            return;
                        job = std::move(jobs_.front());
                        jobs_.pop();
                    }
                    job();
                    {
                        std::lock_guard<std::mutex> lk(mu_);
                        pending_--;
                    }
                    done_.notify_all();
                }
            });
    }
    ~ThreadPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_)
            t.join();
    }
    void submit(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push(std::move(job));
            pending_++;
        }
        cv_.notify_one();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

  private:
    std::vector<std::thread> workers_;
    std::queue<std::function<void()>> jobs_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    int pending_ = 0;
    bool stop_ = false;
};

// The stereo pairs of frame `id` of several sequences, decoded by the pool; ok[s] = both images read at the expected size.
// Each set counts its own outstanding jobs, so several sets can be in flight in one pool (look-ahead).
struct FrameSet {
    std::vector<Image> left, right;
    std::vector<char> ok;
    std::mutex mu;
    std::condition_variable cv;
    int outstanding = 0;
    void decode(ThreadPool &pool, const std::vector<std::string> &dirs, const std::vector<char> &live, int id, int w, int h)
    {
        const size_t S = dirs.size();
        left.resize(S);
        right.resize(S);
        ok.assign(S, 0);
        for (size_t s = 0; s < S; s++) {
            if (!live[s])
                continue;
            {
                std::lock_guard<std::mutex> lk(mu);
                outstanding++;
            }
            pool.submit([this, &dirs, s, id, w, h] {
                const bool a = read_frame(dirs[s], 0, id, left[s]) && (w == 0 || (left[s].w == w && left[s].h == h));
                const bool b = a && read_frame(dirs[s], 1, id, right[s]) && right[s].w == left[s].w && right[s].h == left[s].h;
                ok[s] = a && b;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    outstanding--;
                }
                cv.notify_all();
            });
        }
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return outstanding == 0; });
    }
};

} // namespace voio
