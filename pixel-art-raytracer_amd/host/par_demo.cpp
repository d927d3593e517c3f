// par_demo.cpp — headless counterpart of the reference's `main` loop (src/alternative.cpp:502-833) on the C ABI.
//
// Builds the reference's default graybox world (alt:517-599), plays a key script the way the reference's event
// loop applies keys (alt:641-681: arrows/PgUp/PgDn move entity 0 by 5, a k j u h o move the light by 5), renders
// every frame on the GPU through libpar_raytracer.so, optionally draws the debug line (alt:763-772) and writes the
// frames as binary PPM (P6) instead of presenting them through SDL (alt:774-788). Prints the per-frame time like
// alt:815-817.
//
//   par_demo [--keys RRRRUUUUhhhhjjPP] [--frames N] [--out DIR] [--gif FILE] [--debug-line] [--as-sdl] [--size W H L]
//
// --gif writes the frames as one animated GIF89a (100 ms per frame like the reference's gif.gif); a frame's colours
// are palette entries times a brightness, at most a few hundred distinct values, so each frame gets an exact local
// colour table (frames with more than 256 colours fall back to a 3-3-2 bit table). --as-sdl shows the frame as the
// reference's window does: its SDL_PIXELFORMAT_RGB888 texture reads the struct's `red` byte as blue on little-endian
// machines (alt:613, 772: the debug line comes out blue).
//
// Letters: R L U D P N = right, left, up, down, page-up, page-down; a k j u h o as in the reference. Frame 0 gets no
// key; frame k applies key k-1 (cycling when --frames exceeds the script).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <utility>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "par_raytracer.h"

static void apply_key(char k, par_aabb& player, par_light& light) {
    switch (k) {
        case 'L': player.px -= 5; break;  // SDLK_LEFT  alt:643-645
        case 'R': player.px += 5; break;  // SDLK_RIGHT alt:646-648
        case 'U': player.pz += 5; break;  // SDLK_UP    alt:649-651
        case 'D': player.pz -= 5; break;  // SDLK_DOWN  alt:652-654
        case 'N': player.py -= 5; break;  // PAGEDOWN   alt:655-657
        case 'P': player.py += 5; break;  // PAGEUP     alt:658-660
        case 'a': light.z -= 5; break;    // alt:661-663
        case 'k': light.z += 5; break;    // alt:664-666
        case 'j': light.y -= 5; break;    // alt:667-669
        case 'u': light.y += 5; break;    // alt:670-672
        case 'h': light.x -= 5; break;    // alt:673-675
        case 'o': light.x += 5; break;    // alt:676-678
        default: break;
    }
}

static bool write_ppm(const std::string& path, const par_color* fb, int w, int h) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<unsigned char> row((size_t)w * 3);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const par_color c = fb[(size_t)y * w + x];
            row[(size_t)x * 3 + 0] = c.red;
            row[(size_t)x * 3 + 1] = c.green;
            row[(size_t)x * 3 + 2] = c.blue;
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}

// ---- animated GIF89a writer (frame sink replacing the SDL present, alt:774-788) ---------------------------------
class GifWriter {
  public:
    bool open(const std::string& path, int w, int h) {
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) return false;
        w_ = w; h_ = h;
        std::fwrite("GIF89a", 1, 6, f_);
        put16(w); put16(h);
        std::fputc(0x70, f_);  // no global colour table, 8 bits of colour resolution
        std::fputc(0, f_);
        std::fputc(0, f_);
        const unsigned char loop[] = {0x21, 0xFF, 0x0B, 'N', 'E', 'T', 'S', 'C', 'A', 'P', 'E', '2', '.', '0', 3, 1, 0, 0, 0};
        std::fwrite(loop, 1, sizeof(loop), f_);
        return true;
    }
    void frame(const std::vector<unsigned char>& rgb, int delay_cs) {
        // exact local colour table when the frame has at most 256 colours
        std::map<uint32_t, int> index;
        std::vector<unsigned char> table;
        std::vector<unsigned char> pix((size_t)w_ * h_);
        bool exact = true;
        for (size_t i = 0; i < pix.size() && exact; i++) {
            const uint32_t c = (uint32_t)rgb[i * 3] | ((uint32_t)rgb[i * 3 + 1] << 8) | ((uint32_t)rgb[i * 3 + 2] << 16);
            auto it = index.find(c);
            if (it == index.end()) {
                if (index.size() == 256) { exact = false; break; }
                it = index.emplace(c, (int)index.size()).first;
                table.push_back(rgb[i * 3]); table.push_back(rgb[i * 3 + 1]); table.push_back(rgb[i * 3 + 2]);
            }
            pix[i] = (unsigned char)it->second;
        }
        if (!exact) {  // 3-3-2 bit quantisation
            table.clear();
            for (int c = 0; c < 256; c++) {
                table.push_back((unsigned char)(((c >> 5) & 7) * 255 / 7));
                table.push_back((unsigned char)(((c >> 2) & 7) * 255 / 7));
                table.push_back((unsigned char)((c & 3) * 255 / 3));
            }
            for (size_t i = 0; i < pix.size(); i++) {
                pix[i] = (unsigned char)(((rgb[i * 3] >> 5) << 5) | ((rgb[i * 3 + 1] >> 5) << 2) | (rgb[i * 3 + 2] >> 6));
            }
        }
        table.resize(256 * 3, 0);
        const unsigned char gce[] = {0x21, 0xF9, 4, 0, (unsigned char)(delay_cs & 0xFF), (unsigned char)(delay_cs >> 8), 0, 0};
        std::fwrite(gce, 1, sizeof(gce), f_);
        std::fputc(0x2C, f_);
        put16(0); put16(0); put16(w_); put16(h_);
        std::fputc(0x87, f_);  // local colour table, 256 entries
        std::fwrite(table.data(), 1, table.size(), f_);
        lzw(pix);
    }
    void close() {
        if (f_) { std::fputc(0x3B, f_); std::fclose(f_); f_ = nullptr; }
    }

  private:
    void put16(int v) { std::fputc(v & 0xFF, f_); std::fputc((v >> 8) & 0xFF, f_); }
    void emit(unsigned code, int bits) {
        acc_ |= (uint64_t)code << nacc_;
        nacc_ += bits;
        while (nacc_ >= 8) {
            block_.push_back((unsigned char)(acc_ & 0xFF));
            acc_ >>= 8; nacc_ -= 8;
            if (block_.size() == 255) flush_block();
        }
    }
    void flush_block() {
        if (block_.empty()) return;
        std::fputc((int)block_.size(), f_);
        std::fwrite(block_.data(), 1, block_.size(), f_);
        block_.clear();
    }
    void lzw(const std::vector<unsigned char>& pix) {
        const int min_bits = 8, clear = 1 << min_bits, eoi = clear + 1;
        std::fputc(min_bits, f_);
        std::unordered_map<uint32_t, int> dict;
        int next = eoi + 1, bits = min_bits + 1;
        acc_ = 0; nacc_ = 0;
        emit(clear, bits);
        int prefix = pix[0];
        for (size_t i = 1; i < pix.size(); i++) {
            const uint32_t key = ((uint32_t)prefix << 8) | pix[i];
            auto it = dict.find(key);
            if (it != dict.end()) { prefix = it->second; continue; }
            emit(prefix, bits);
            if (next < 4096) {
                dict.emplace(key, next);
                if (next == (1 << bits)) bits++;  // the code just added needs one more bit from now on
                next++;
            } else {
                emit(clear, bits);
                dict.clear();
                next = eoi + 1;
                bits = min_bits + 1;
            }
            prefix = pix[i];
        }
        emit(prefix, bits);
        emit(eoi, bits);
        if (nacc_ > 0) { block_.push_back((unsigned char)(acc_ & 0xFF)); acc_ = 0; nacc_ = 0; }
        flush_block();
        std::fputc(0, f_);  // block terminator
    }
    FILE* f_ = nullptr;
    int w_ = 0, h_ = 0;
    uint64_t acc_ = 0;
    int nacc_ = 0;
    std::vector<unsigned char> block_;
};

int main(int argc, char** argv) {
    std::string keys = "RRRRUUUUhhhhjjPP", out_dir, gif_path;
    int frames = -1, W = 480, H = 320, L = 320;
    bool debug_line = false, as_sdl = false;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_dir = argv[++i];
        else if (!std::strcmp(argv[i], "--gif") && i + 1 < argc) gif_path = argv[++i];
        else if (!std::strcmp(argv[i], "--debug-line")) debug_line = true;
        else if (!std::strcmp(argv[i], "--as-sdl")) as_sdl = true;
        else if (!std::strcmp(argv[i], "--size") && i + 3 < argc) { W = std::atoi(argv[++i]); H = std::atoi(argv[++i]); L = std::atoi(argv[++i]); }
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (frames < 0) frames = (int)keys.size() + 1;

    par_params params;
    par_default_params(&params);
    params.width = W; params.height = H; params.length = L;

    // scene, alt:517-599 + light alt:624-626
    const int n = par_scene_graybox(W, L, nullptr, 0);
    std::vector<par_aabb> aabbs((size_t)n);
    par_scene_graybox(W, L, aabbs.data(), n);
    par_sprite tile;
    par_sprite_tile_floor(&tile);
    par_light light{(int16_t)W, (int16_t)(H / 2), (int16_t)(L / 4), 10};

    par_context* ctx = nullptr;
    int rc = par_create(&params, 0, &ctx);
    if (rc != PAR_OK) { std::fprintf(stderr, "par_create: %s\n", par_status_string(rc)); return 1; }
    if ((rc = par_set_sprites(ctx, &tile, 1)) != PAR_OK || (rc = par_set_entities(ctx, aabbs.data(), nullptr, n)) != PAR_OK ||
        (rc = par_set_light(ctx, &light)) != PAR_OK) {
        std::fprintf(stderr, "scene upload: %s (%s)\n", par_status_string(rc), par_last_error(ctx));
        return 1;
    }

    std::vector<par_color> fb((size_t)W * H);
    std::vector<par_pixel> gbuf((size_t)W * H);
    std::vector<unsigned char> rgb((size_t)W * H * 3);
    GifWriter gif;
    if (!gif_path.empty() && !gif.open(gif_path, W, H)) { std::fprintf(stderr, "cannot write %s\n", gif_path.c_str()); return 1; }
    const int mouse_x = 0, mouse_y = 0;  // the reference's mouse position before any motion event (alt:133-134)
    for (int f = 0; f < frames; f++) {
        if (f > 0 && !keys.empty()) {
            apply_key(keys[(size_t)(f - 1) % keys.size()], aabbs[0], light);
            par_update_aabbs(ctx, &aabbs[0], 0, 1);
            par_set_light(ctx, &light);
        }
        const auto t0 = std::chrono::steady_clock::now();
        par_outputs o{fb.data(), gbuf.data(), nullptr, nullptr, nullptr};
        if ((rc = par_render(ctx, &o, 0)) != PAR_OK) {
            std::fprintf(stderr, "par_render: %s (%s)\n", par_status_string(rc), par_last_error(ctx));
            return 1;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (debug_line) par_debug_line(&params, &gbuf[(size_t)mouse_y * W + mouse_x], mouse_x, &light, fb.data());
        std::printf("frame %d: %.3fms  player <%d, %d, %d>  light <%d, %d, %d>\n", f, ms, aabbs[0].px, aabbs[0].py,
                    aabbs[0].pz, light.x, light.y, light.z);  // alt:815-817 prints the frame time
        if (as_sdl) {
            for (auto& c : fb) std::swap(c.red, c.blue);
        }
        if (!out_dir.empty()) {
            char name[64];
            std::snprintf(name, sizeof(name), "/frame_%03d.ppm", f);
            if (!write_ppm(out_dir + name, fb.data(), W, H)) { std::fprintf(stderr, "cannot write %s%s\n", out_dir.c_str(), name); return 1; }
        }
        if (!gif_path.empty()) {
            for (size_t i = 0; i < fb.size(); i++) { rgb[i * 3] = fb[i].red; rgb[i * 3 + 1] = fb[i].green; rgb[i * 3 + 2] = fb[i].blue; }
            gif.frame(rgb, 10);  // 100 ms per frame, as the reference's gif.gif
        }
    }
    gif.close();
    par_destroy(ctx);
    return 0;
}
