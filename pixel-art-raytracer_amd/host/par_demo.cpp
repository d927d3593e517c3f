// par_demo.cpp — headless counterpart of the reference's `main` loop (src/alternative.cpp:502-833) on the C ABI.
//
// Builds the reference's default graybox world (alt:517-599), plays a key script the way the reference's event
// loop applies keys (alt:641-681: arrows/PgUp/PgDn move entity 0 by 5, a k j u h o move the light by 5), renders
// every frame on the GPU through libpar_raytracer.so, optionally draws the debug line (alt:763-772) and writes the
// frames as binary PPM (P6) instead of presenting them through SDL (alt:774-788). Prints the per-frame time like
// alt:815-817.
//
//   par_demo [--keys RRRRUUUUhhhhjjPP] [--frames N] [--out DIR] [--debug-line] [--size W H L]
//
// Letters: R L U D P N = right, left, up, down, page-up, page-down; a k j u h o as in the reference. Frame 0 gets no
// key; frame k applies key k-1 (cycling when --frames exceeds the script).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

int main(int argc, char** argv) {
    std::string keys = "RRRRUUUUhhhhjjPP", out_dir;
    int frames = -1, W = 480, H = 320, L = 320;
    bool debug_line = false;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--keys") && i + 1 < argc) keys = argv[++i];
        else if (!std::strcmp(argv[i], "--frames") && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--out") && i + 1 < argc) out_dir = argv[++i];
        else if (!std::strcmp(argv[i], "--debug-line")) debug_line = true;
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
        if (!out_dir.empty()) {
            char name[64];
            std::snprintf(name, sizeof(name), "/frame_%03d.ppm", f);
            if (!write_ppm(out_dir + name, fb.data(), W, H)) { std::fprintf(stderr, "cannot write %s%s\n", out_dir.c_str(), name); return 1; }
        }
    }
    par_destroy(ctx);
    return 0;
}
