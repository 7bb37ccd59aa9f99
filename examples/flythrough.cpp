// flythrough.cpp -- the reference's main.rs (src/main.rs:9-28) driven headless through the C++
// host mirror: GraphicsCore + GameState, a per-frame closure that updates the player and returns
// the push constants, and scripted input in place of the window's key / mouse events.
//
//   flythrough [frames] [width] [height] [out.bgra]
//
// Prints the camera transform of the last frame (16 floats, column major) and an FNV-1a hash
// of its B8G8R8A8 frame; optionally writes the raw frame.
#include "game_logic.hpp"

#include <cstdlib>
#include <fstream>

using namespace tube;

static const float *c_sens()
{
    static const GraphicsSettings s;
    return s.mouse_sensitivity;
}

int main(int argc, char **argv)
{
    const uint32_t frames = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 120;
    const uint32_t width  = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 640;
    const uint32_t height = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 480;
    try {
        GraphicsCore core(width, height);                 // main.rs:10
        GameState    game_state(core);                    // main.rs:12
        game_state.speed_modifier = 40.0f;                // a few mouse-wheel notches (mod.rs:447-454)
        int regenerated = 0;
        core.run(game_state,
                 [&](GraphicsCore &c, GameState &gs) {    // main.rs:15-27
                     const Vector3i before = gs.current_player_chunk();
                     gs.update_player(gs.speed_modifier, c);
                     const Vector3i after = gs.current_player_chunk();
                     if ((after - before).amax() != 0) {
                         ++regenerated;
                         std::printf("chunk %d %d %d\n", after.x, after.y, after.z);
                     }
                     return PushConstants{gs.get_player_transform(), 0};
                 },
                 frames,
                 [&](uint32_t f, GameState &gs) {         // scripted input
                     if (f == 0) { gs.key_states.update(VirtualKeyCode::W, true); gs.key_states.update(VirtualKeyCode::Space, true); }
                     if (f == frames / 2) gs.key_states.update(VirtualKeyCode::Space, false);
                     gs.player_look_dir(1.5, f < frames / 2 ? 0.4 : -0.2, c_sens());
                 });
        const Matrix4 m = game_state.get_player_transform();
        std::printf("transform");
        for (float v : m.m) std::printf(" %.9g", v);
        std::printf("\n");
        uint64_t h = 1469598103934665603ull;
        for (uint8_t b : core.frame()) { h ^= b; h *= 1099511628211ull; }
        std::printf("frames %u size %ux%u chunk_moves %d fnv1a %016llx\n", frames, width, height, regenerated, (unsigned long long)h);
        if (argc > 4) {
            std::ofstream f(argv[4], std::ios::binary);
            f.write((const char *)core.frame().data(), (std::streamsize)core.frame().size());
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "flythrough: %s\n", e.what());
        return 1;
    }
    return 0;
}
