// game_logic.hpp -- the CALLER's side of the reference, in C++: what src/main.rs and src/game_logic do around
// GraphicsCore, for the example (flythrough.cpp) and the test that drives the path the way the game does.
// Not part of the library's interface (include/graphics_core.hpp, include/vrt.h).
//
//   Player         src/game_logic/player/mod.rs:5-64
//   KeyStates      src/game_logic/key_states.rs:4-35
//   GameState      src/game_logic/game_state.rs:10-89
#pragma once

#include "../include/graphics_core.hpp"

#include <set>

namespace tube {

enum class VirtualKeyCode { W, A, S, D, Space, LShift, T, R, Escape };

struct KeyStates {  // key_states.rs:4-35
    std::set<VirtualKeyCode> currently_pressed, new_pressed;
    void update(VirtualKeyCode key, bool pressed)
    {
        if (pressed) { if (currently_pressed.insert(key).second) new_pressed.insert(key); }
        else currently_pressed.erase(key);
    }
    void reset() { new_pressed.clear(); }
};

class Player {  // player/mod.rs:5-64
public:
    static constexpr float BASE_SPEED = 0.02f;
    float look_direction[2] = {0.0f, 0.0f};  // (yaw, pitch)

    static Player spawn() { return Player(); }
    Matrix4  get_transform() const { return matrix_; }
    Vector3f get_position() const { return {matrix_.at(0, 3), matrix_.at(1, 3), matrix_.at(2, 3)}; }
    void     set_position(Vector3f p) { matrix_.at(0, 3) = p.x; matrix_.at(1, 3) = p.y; matrix_.at(2, 3) = p.z; }
    void apply_rotation()  // :25-30  matrix = T * Rx(pitch) * Ry(yaw)
    {
        reset_rotation();
        const Matrix4 transform = Matrix4::from_scaled_axis({look_direction[1], 0, 0}) * Matrix4::from_scaled_axis({0, look_direction[0], 0});
        matrix_ = matrix_ * transform;
    }
    void update_pos(const std::set<VirtualKeyCode> &keys, float speed_modifier)  // :37-63
    {
        const float yaw = look_direction[0];
        const Vector3f forward{-std::sin(yaw), 0.0f, std::cos(yaw)}, right{std::cos(yaw), 0.0f, std::sin(yaw)}, up{0, 1, 0};
        Vector3f mov;
        auto add = [&](Vector3f v, float s) { mov.x += s * v.x; mov.y += s * v.y; mov.z += s * v.z; };
        if (keys.count(VirtualKeyCode::W)) add(forward, 1);
        if (keys.count(VirtualKeyCode::A)) add(right, -1);
        if (keys.count(VirtualKeyCode::S)) add(forward, -1);
        if (keys.count(VirtualKeyCode::D)) add(right, 1);
        if (keys.count(VirtualKeyCode::Space)) add(up, 1);
        if (keys.count(VirtualKeyCode::LShift)) add(up, -1);
        const float cap = BASE_SPEED * speed_modifier;
        const float norm = std::sqrt(mov.x * mov.x + mov.y * mov.y + mov.z * mov.z);
        if (norm > cap) { const float s = cap / norm; mov.x *= s; mov.y *= s; mov.z *= s; }  // cap_magnitude
        matrix_.at(0, 3) += mov.x; matrix_.at(1, 3) += mov.y; matrix_.at(2, 3) += mov.z;      // append_translation_mut
    }

private:
    void reset_rotation()  // :32-36
    {
        const Vector3f t = get_position();
        matrix_ = Matrix4::identity();
        set_position(t);
    }
    Matrix4 matrix_ = Matrix4::identity();
};

class GameState {  // game_state.rs:10-89
public:
    Player    player = Player::spawn();
    KeyStates key_states;
    float     speed_modifier = 1.0f;

    explicit GameState(GraphicsCore &render_core) { upload_chunks_around_player(render_core, nullptr); }  // :27-39

    Vector3i current_player_chunk() const  // :20-22
    {
        const Vector3f p = player.get_position();
        const float cs = (float)GraphicsCore::CHUNK_SIZE;
        return {(int32_t)std::floor(p.x / cs), (int32_t)std::floor(p.y / cs), (int32_t)std::floor(p.z / cs)};
    }
    void player_look_dir(double dx, double dy, const float sens[2])  // :23-26
    {
        player.look_direction[0] = player.look_direction[0] - (float)dx * sens[0];
        player.look_direction[1] = std::min(1.7f, std::max(-1.7f, player.look_direction[1] - (float)dy * sens[1]));
    }
    void update_player(float speed, GraphicsCore &render_core)  // :41-48
    {
        const Vector3i old_chunk = current_player_chunk();
        player.apply_rotation();
        player.update_pos(key_states.currently_pressed, speed);
        upload_chunks_around_player(render_core, &old_chunk);
    }
    // :50-83; returns the number of chunks regenerated
    int upload_chunks_around_player(GraphicsCore &core, const Vector3i *old_chunk)
    {
        const Vector3i new_chunk = current_player_chunk();
        if (old_chunk && (new_chunk - *old_chunk).amax() == 0) return 0;
        const int gen_dist = (int)core.get_settings().render_distance + 1;
        int count = 0;
        for (int x = -gen_dist; x <= gen_dist; ++x)
            for (int y = -gen_dist; y <= gen_dist; ++y)
                for (int z = -gen_dist; z <= gen_dist; ++z) {
                    const Vector3i chunk_position = new_chunk + Vector3i{x, y, z};
                    if (old_chunk && (chunk_position - *old_chunk).amax() <= gen_dist) continue;
                    core.upload_chunk_gpu(chunk_position, *this);
                    ++count;
                }
        core.previous_frame_end.then_signal_fence_and_flush().wait();  // :77
        core.reset_distance_field();                                    // :82
        return count;
    }
    Matrix4 get_player_transform() const { return player.get_transform(); }  // :86-88
};

}  // namespace tube
