// graphics_core.hpp -- C++ host-side mirror of the reference's interface for the trace path,
// on top of the C ABI (vrt.h).  Header only; needs nothing but the standard library.
//
// The reference is a Rust crate and no Rust toolchain exists where this was built, so the
// host side is written in C++ with the reference's names, argument meaning and error
// behaviour, for callers that drive the path the way src/main.rs and src/game_logic do:
//
//   GraphicsCore   src/graphics/mod.rs:134-148,150-183,245-291,292-428,430-475,531-634
//   PushConstants  src/graphics/push_constants.rs:4-9
//   GraphicsSettings src/graphics/graphics_settings.rs:1-17
//
// The callers' own code (Player, KeyStates, GameState of src/game_logic) is not part of this interface; a C++
// rendering of it, used by the example and its test, lives in examples/game_logic.hpp.
//
// What is NOT mirrored: window, swapchain, event loop and input devices (presentation, not
// part of this path).  `GraphicsCore::run` therefore takes a frame count and an optional
// per-frame input callback instead of never returning, and `redraw` leaves the
// B8G8R8A8 frame in host memory (`frame()`) instead of presenting it.  Errors are C++
// exceptions where the reference panics (`unwrap`).
#pragma once

#include "vrt.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstddef>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <vector>

namespace tube {  // the reference crate's name (Cargo.toml:2)

struct Vector3f { float x = 0, y = 0, z = 0; };
struct Vector3i {
    int32_t x = 0, y = 0, z = 0;
    int32_t amax() const { return std::max(std::abs(x), std::max(std::abs(y), std::abs(z))); }
};
inline Vector3i operator+(Vector3i a, Vector3i b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3i operator-(Vector3i a, Vector3i b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

// nalgebra Matrix4<f32>: column major, m[4*col + row] (= GLSL mat4).
struct Matrix4 {
    float m[16];
    static Matrix4 identity()
    {
        Matrix4 r{};
        for (int i = 0; i < 16; ++i) r.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
        return r;
    }
    float &at(int row, int col) { return m[4 * col + row]; }
    float at(int row, int col) const { return m[4 * col + row]; }
    // Matrix4::from_scaled_axis: rotation by |v| about v/|v| (Rodrigues), identity for v = 0
    static Matrix4 from_scaled_axis(Vector3f v)
    {
        Matrix4 r = identity();
        const float angle = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
        if (angle == 0.0f) return r;
        const float ux = v.x / angle, uy = v.y / angle, uz = v.z / angle;
        const float s = std::sin(angle), c = std::cos(angle), omc = 1.0f - c;
        r.at(0, 0) = ux * ux + (1.0f - ux * ux) * c; r.at(0, 1) = ux * uy * omc - uz * s; r.at(0, 2) = ux * uz * omc + uy * s;
        r.at(1, 0) = ux * uy * omc + uz * s; r.at(1, 1) = uy * uy + (1.0f - uy * uy) * c; r.at(1, 2) = uy * uz * omc - ux * s;
        r.at(2, 0) = ux * uz * omc - uy * s; r.at(2, 1) = uy * uz * omc + ux * s; r.at(2, 2) = uz * uz + (1.0f - uz * uz) * c;
        return r;
    }
};
inline Matrix4 operator*(const Matrix4 &a, const Matrix4 &b)
{
    Matrix4 r{};
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc += a.at(row, k) * b.at(k, col);
            r.at(row, col) = acc;
        }
    return r;
}

// #[repr(C)] struct PushConstants { transform: Matrix4<f32>, image_index: u32 }  (68 bytes)
struct PushConstants {
    Matrix4  transform;
    uint32_t image_index;
};
static_assert(sizeof(PushConstants) == 68, "PushConstants must match the shader's push-constant block");

struct GraphicsSettings {  // graphics_settings.rs:9-16
    uint8_t render_distance = 1;
    double  fov = 90.0;  // never read by the reference either
    float   mouse_sensitivity[2] = {0.002f, 0.002f};
};

// Stands in for `Option<Box<dyn GpuFuture>>` (src/graphics/mod.rs:140): game_state.rs:77 calls
// previous_frame_end.take().unwrap().then_signal_fence_and_flush().unwrap().wait(None).unwrap().
struct FrameToken {
    vrt_ctx *ctx = nullptr;
    FrameToken then_signal_fence_and_flush() const { return *this; }
    void wait() const
    {
        if (vrt_sync(ctx) != VRT_OK) throw std::runtime_error(vrt_last_error());
    }
};

class GraphicsCore {
public:
    static constexpr uint32_t CHUNK_SIZE = 32;                                   // mod.rs:147
    static constexpr uint32_t CHUNK_SIZE_3 = CHUNK_SIZE * CHUNK_SIZE * CHUNK_SIZE; // mod.rs:148

    // ≙ GraphicsCore::new (mod.rs:292-425) without the window: `width` x `height` is the size the
    // window's inner_size() would report; the world images have extent 32*(2*render_distance+3).
    explicit GraphicsCore(uint32_t width = 640, uint32_t height = 480, int device = 0,
                          GraphicsSettings settings = GraphicsSettings())
        : settings_(settings), width_(width), height_(height)
    {
        vrt_config cfg{};
        cfg.struct_size  = sizeof cfg;
        cfg.device       = device;
        cfg.grid_n       = CHUNK_SIZE * (2u * settings.render_distance + 3u);  // mod.rs:359
        cfg.voxel_format = VRT_VOXEL_REF_U16U16;
        cfg.max_steps    = 0;
        cfg.trace_mode   = VRT_TRACE_AUTO;
        check(vrt_create(&cfg, &ctx_));
        check(vrt_kernel_time_sampling(ctx_, 0));  // a game loop does not read kernel times: no events around the launches
        previous_frame_end = FrameToken{ctx_};
        resize(width_, height_);
    }
    ~GraphicsCore()
    {
        vrt_free_host(ctx_, frame_.ptr);
        vrt_destroy(ctx_);
    }
    GraphicsCore(const GraphicsCore &) = delete;
    GraphicsCore &operator=(const GraphicsCore &) = delete;

    GraphicsSettings get_settings() const { return settings_; }  // mod.rs:426-428

    // mod.rs:245-291
    // (the reference passes its GameState along and does not use it; any type will do)
    template <class GameStateT>
    void upload_chunk_gpu(Vector3i chunk_pos, GameStateT &) { check(vrt_generate_chunk(ctx_, chunk_pos.x, chunk_pos.y, chunk_pos.z)); }
    // mod.rs:150-183
    void reset_distance_field() { check(vrt_reset_distance_field(ctx_)); }

    // mod.rs:531-634: returns true when the frame was skipped (zero-sized window)
    bool redraw(const PushConstants &push_constants)
    {
        if (width_ == 0 || height_ == 0) return true;  // mod.rs:534-536
        vrt_outputs out{};
        out.struct_size = sizeof out;
        out.memory      = VRT_MEM_HOST;
        out.bgra8       = frame_.ptr;
        check(vrt_trace(ctx_, push_constants.transform.m, width_, height_, 0, height_, &out));
        return false;
    }

    // mod.rs:430-475 without the OS event loop: `frames` redraws; `input` is called before each
    // frame in place of the window/device events the reference feeds into game_state.
    // GameStateT is the caller's game state (the reference's GameState, src/game_logic/game_state.rs; a C++
    // rendering of it lives in examples/game_logic.hpp); the loop touches its `key_states.reset()` only.
    template <class GameStateT, class GameUpdate, class Input = std::nullptr_t>
    void run(GameStateT &game_state, GameUpdate &&game_update, uint32_t frames, Input &&input = nullptr)
    {
        for (uint32_t f = 0; f < frames; ++f) {
            if constexpr (!std::is_same_v<std::decay_t<Input>, std::nullptr_t>) input(f, game_state);  // ≙ keyboard / mouse events (mod.rs:437-454)
            const PushConstants pc = game_update(*this, game_state);  // ≙ RedrawEventsCleared (mod.rs:467-471)
            if (redraw(pc)) continue;
            game_state.key_states.reset();
        }
    }

    void resize(uint32_t width, uint32_t height)  // ≙ WindowEvent::Resized -> recreate_swapchain
    {
        width_ = width; height_ = height;
        check(vrt_free_host(ctx_, frame_.ptr));
        frame_ = Frame{};
        const size_t bytes = (size_t)width_ * height_ * 4;
        if (bytes == 0) return;
        void *p = nullptr;
        check(vrt_alloc_host(ctx_, bytes, &p));  // page-locked: the frame comes back at the bus rate
        frame_.ptr = static_cast<uint8_t *>(p);
        frame_.bytes = bytes;
        std::memset(frame_.ptr, 0, bytes);
    }
    // B,G,R,A bytes, rows top to bottom, in page-locked host memory owned by the core
    struct Frame {
        uint8_t *ptr = nullptr;
        size_t   bytes = 0;
        const uint8_t *data() const { return ptr; }
        size_t         size() const { return bytes; }
        const uint8_t *begin() const { return ptr; }
        const uint8_t *end() const { return ptr + bytes; }
        uint8_t        operator[](size_t i) const { return ptr[i]; }
    };
    const Frame &frame() const { return frame_; }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    vrt_ctx *context() { return ctx_; }

    FrameToken previous_frame_end;  // pub(crate) in the reference; game_state.rs:77-79 touches it

private:
    static void check(int rc)
    {
        if (rc != VRT_OK) throw std::runtime_error(std::string("vrt: ") + vrt_last_error());
    }
    vrt_ctx             *ctx_ = nullptr;
    GraphicsSettings     settings_;
    uint32_t             width_, height_;
    Frame                frame_;
};

}  // namespace tube
