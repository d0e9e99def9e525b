// Generation driver behind the libsdod C API: the MI355X counterpart of the reference's libsdod::Context
// (csrc/libsdod/src/context.h:27-99, context.cpp:49-403).  Same life cycle -- load tokenizer and models,
// prepare buffers (uncond embedding), prepare the schedule (solver tables + cached time embeddings),
// generate(prompt, guidance) -> uint8 HWC image -- with the four QNN graphs replaced by engine graphs and the
// host-side CFG / solver arithmetic moved onto the GPU.
#pragma once
#include <chrono>
#include <memory>
#include <optional>
#include <random>
#include <string>
#include <vector>

#include "../engine.h"
#include "dpm_solver.h"
#include "tokenizer.h"

namespace sdod {

enum class LogLevel : unsigned { NOTHING = 0, ERROR = 1, INFO = 2, DEBUG = 3, ABUSIVE = 4 }; // libsdod.h:20-26

class Logger {
public:
    void set_level(LogLevel l) { level_ = l; }
    LogLevel level() const { return level_; }
    void log(LogLevel l, const std::string& msg) const;
    void error(const std::string& m) const { log(LogLevel::ERROR, m); }
    void info(const std::string& m) const { log(LogLevel::INFO, m); }
    void debug(const std::string& m) const { log(LogLevel::DEBUG, m); }

private:
    LogLevel level_ = LogLevel::ERROR;
};

class Context {
public:
    Context(const std::string& models_dir, unsigned latent_channels, unsigned latent_spatial, unsigned upscale_factor,
            LogLevel log_level, int device);
    ~Context();

    void init(unsigned steps);                 // context.cpp:49-80 (load everything, then the schedule)
    void prepare_schedule(unsigned steps);     // context.cpp:245-282 (any steps >= 1; the reference accepts only 20)
    void set_seed(unsigned seed);              // context.cpp:285-289
    void set_initial_latent(const float* x, size_t n); // parity runs inject x_T (RNG streams are not portable)
    size_t image_bytes() const { return (size_t)3 * latent_spatial_ * upscale_ * latent_spatial_ * upscale_; }
    void generate(const std::string& prompt, float guidance, unsigned char* out); // context.cpp:292-403

    Logger& logger() { return logger_; }
    std::string error_slots[6];                // per-context ErrorTable (errors.h:23)
    bool error_set[6] = {false, false, false, false, false, false};

private:
    std::string models_dir_;
    unsigned latent_channels_, latent_spatial_, upscale_;
    int device_;
    Logger logger_;
    std::mt19937 rng_;
    std::normal_distribution<float> normal_{0.0f, 1.0f};

    sdod_model_config cfg_{};
    std::optional<Tokenizer> tokenizer_;
    std::optional<DpmSolver> solver_;
    std::unique_ptr<Graph> unet_, text_, vae_, temb_;
    unsigned steps_ = 0;
    std::vector<float> model_ts_;

    hipStream_t stream_ = nullptr;
    std::vector<hipEvent_t> phase_events_; // generate()'s phase boundaries: start, conditioning, every iteration, decode
    f16* temb_cache_ = nullptr;   // [steps][4*model_ch]
    f16* ctx_uncond_ = nullptr;   // [77][ctx_dim]
    float* e_dev_ = nullptr;      // CFG result, fp32 NCHW
    float* y_prev_ = nullptr;     // solver history
    float* x_dev_ = nullptr;      // latent, fp32 NCHW
    uint8_t* img_u8_ = nullptr;
    std::vector<float> x_host_;
    bool injected_ = false;

    void encode_prompt(const std::string& prompt, f16* dst);
    std::string weight_path(const char* stem) const { return models_dir_ + "/" + stem + ".sdodw"; }
};

} // namespace sdod
