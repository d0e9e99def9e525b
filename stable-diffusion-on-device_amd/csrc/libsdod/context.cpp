#include "context.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include "sdod_hip.h"

namespace sdod {

static void rc_check(int rc) {
    if (rc != 0) throw Error(rc, get_last_error());
}

void Logger::log(LogLevel l, const std::string& msg) const {
    if (l == LogLevel::NOTHING || (unsigned)l > (unsigned)level_) return;
    static const char* tag[] = {"", "E", "I", "D", "A"};
    std::FILE* sink = l == LogLevel::ERROR ? stderr : stdout;
    std::fprintf(sink, "[libsdod %s] %s\n", tag[(unsigned)l], msg.c_str());
    std::fflush(sink);
}

namespace {
struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
} // namespace

Context::Context(const std::string& models_dir, unsigned latent_channels, unsigned latent_spatial, unsigned upscale_factor,
                 LogLevel log_level, int device)
    : models_dir_(models_dir), latent_channels_(latent_channels), latent_spatial_(latent_spatial), upscale_(upscale_factor),
      device_(device), rng_(std::random_device{}()) {
    logger_.set_level(log_level);
    if (models_dir_.empty()) models_dir_ = ".";
    else if (models_dir_.size() > 1 && models_dir_.back() == '/') models_dir_.pop_back();
    SDOD_REQUIRE(latent_channels_ >= 1 && latent_channels_ <= 7, "latent_channels must be in [1, 7]");
    SDOD_REQUIRE(latent_spatial_ >= 8 && latent_spatial_ % 8 == 0, "latent_spatial must be a positive multiple of 8");
    SDOD_REQUIRE(upscale_ == 8, "upscale_factor must be 8 (three 2x decoder levels)");
    sdod_model_config_sd14(&cfg_);
    cfg_.latent_channels = (int)latent_channels_;
    cfg_.latent_h = cfg_.latent_w = (int)latent_spatial_;
}

Context::~Context() {
    (void)hipSetDevice(device_);
    for (void* p : {(void*)temb_cache_, (void*)ctx_uncond_, (void*)e_dev_, (void*)y_prev_, (void*)x_dev_, (void*)img_u8_})
        if (p) (void)hipFree(p);
    unet_.reset(); text_.reset(); vae_.reset(); temb_.reset();
    for (hipEvent_t e : phase_events_)
        if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void Context::init(unsigned steps) {
    Timer t;
    // host-only pieces first: a models_dir without its files is reported as such even where no device is visible
    tokenizer_.emplace(models_dir_ + "/ctokenizer.txt"); // context.cpp:180-188
    logger_.info("Tokenizer created!");
    solver_.emplace(1000, 0.00085f, 0.0120f);            // context.cpp:191-198
    logger_.info("ODE solver prepared!");
    SDOD_HIP_CHECK(hipSetDevice(device_));
    SDOD_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));

    // context.cpp:94-177: the four graphs.  cond + uncond run as ONE batch-2 UNet evaluation here.
    struct Item { std::unique_ptr<Graph>* slot; int kind; int batch; const char* stem; };
    const Item items[] = {{&unet_, SDOD_GRAPH_UNET, 2, "unet"}, {&text_, SDOD_GRAPH_TEXT_ENCODER, 1, "text_encoder"},
                          {&vae_, SDOD_GRAPH_VAE_DECODER, 1, "vae_decoder"}};
    for (const Item& it : items) {
        logger_.info(std::string("Attempting to load a model: ") + it.stem + ".sdodw");
        it.slot->reset(new Graph(it.kind, cfg_, it.batch));
        (*it.slot)->load_file(weight_path(it.stem), "");
        (*it.slot)->finalize();
        logger_.info(std::string("Model ") + it.stem + " loaded");
    }
    logger_.info("All models loaded!");

    const size_t lat = (size_t)latent_channels_ * latent_spatial_ * latent_spatial_;
    SDOD_HIP_CHECK(hipMalloc((void**)&x_dev_, 2 * lat * sizeof(float)));
    SDOD_HIP_CHECK(hipMalloc((void**)&e_dev_, lat * sizeof(float)));
    SDOD_HIP_CHECK(hipMalloc((void**)&y_prev_, lat * sizeof(float)));
    SDOD_HIP_CHECK(hipMalloc((void**)&img_u8_, image_bytes()));
    SDOD_HIP_CHECK(hipMalloc((void**)&ctx_uncond_, (size_t)cfg_.context_len * cfg_.context_dim * sizeof(f16)));
    x_host_.resize(lat);

    encode_prompt("", ctx_uncond_); // context.cpp:233-239: unconditional embedding computed once
    logger_.info("Input/output buffers created and prepared!");
    prepare_schedule(steps);
    logger_.info("Initialization took " + std::to_string((long)t.ms()) + "ms");
}

void Context::encode_prompt(const std::string& prompt, f16* dst) {
    const auto ids16 = tokenizer_->tokenize(prompt, (unsigned)cfg_.context_len);
    std::vector<int32_t> ids(ids16.begin(), ids16.end());
    const IoSlot in = text_->io(false, 0), out = text_->io(true, 0);
    SDOD_HIP_CHECK(hipMemcpyAsync(in.ptr, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
    text_->execute(stream_, true);
    SDOD_HIP_CHECK(hipMemcpyAsync(dst, out.ptr, out.bytes, hipMemcpyDeviceToDevice, stream_));
    SDOD_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Context::prepare_schedule(unsigned steps) {
    SDOD_REQUIRE(steps >= 1 && steps <= 1000, "steps must be in [1, 1000], got: " + std::to_string(steps));
    SDOD_HIP_CHECK(hipSetDevice(device_));
    model_ts_ = solver_->prepare(steps); // dpm_solver.cpp:100-131
    // context.cpp:257-278: sinusoidal features + time-embedding MLP for the first `steps` model times, cached
    temb_.reset(new Graph(SDOD_GRAPH_TEMB, cfg_, (int)steps));
    temb_->load_file(weight_path("temb"), "");
    temb_->finalize();
    const IoSlot in = temb_->io(false, 0), out = temb_->io(true, 0);
    SDOD_HIP_CHECK(hipMemcpyAsync(in.ptr, model_ts_.data(), steps * sizeof(float), hipMemcpyHostToDevice, stream_));
    temb_->execute(stream_, false);
    if (temb_cache_) SDOD_HIP_CHECK(hipFree(temb_cache_));
    temb_cache_ = nullptr;
    SDOD_HIP_CHECK(hipMalloc((void**)&temb_cache_, out.bytes));
    SDOD_HIP_CHECK(hipMemcpyAsync(temb_cache_, out.ptr, out.bytes, hipMemcpyDeviceToDevice, stream_));
    SDOD_HIP_CHECK(hipStreamSynchronize(stream_));
    steps_ = steps;
    logger_.info("Time schedule prepared for " + std::to_string(steps) + " steps!");
}

void Context::set_seed(unsigned seed) {
    logger_.info("Using seed: " + std::to_string(seed));
    normal_.reset();
    rng_.seed(seed);
}

void Context::set_initial_latent(const float* x, size_t n) {
    SDOD_REQUIRE(x != nullptr && n == x_host_.size(), "initial latent must have latent_channels*latent_spatial^2 floats");
    std::copy(x, x + n, x_host_.begin());
    injected_ = true;
}

namespace {
std::string ms_str(double ms) {
    char b[32];
    std::snprintf(b, sizeof b, "%.2f", ms);
    return b;
}
} // namespace

void Context::generate(const std::string& prompt, float guidance, unsigned char* out) {
    SDOD_REQUIRE(unet_ && text_ && vae_ && temb_cache_ && steps_ > 0, "context is not initialised");
    SDOD_HIP_CHECK(hipSetDevice(device_));
    Timer total;
    logger_.info("Starting image generation for prompt: \"" + prompt + "\" and guidance " + std::to_string(guidance));
    logger_.debug("Current steps: " + std::to_string(steps_));

    const int C = (int)latent_channels_, HW = (int)(latent_spatial_ * latent_spatial_);
    const size_t lat = (size_t)C * HW;
    const size_t ctx_bytes = (size_t)cfg_.context_len * cfg_.context_dim * sizeof(f16);
    const IoSlot ux = unet_->io(false, 0), ut = unet_->io(false, 1), uc = unet_->io(false, 2), ue = unet_->io(true, 0);
    const size_t temb_row = ut.bytes / 2; // one row of projected time conditioning (two batch rows in the UNet slot)

    // The reference brackets its four timers with host clocks because its host loop waits for the device after every graph
    // (context.cpp:324-331, :343-381, :383-398, :402).  Here nothing waits before the image copy, so the phases are bracketed
    // by EVENTS on the stream and read once after the final synchronisation: every one of the reference's INFO lines, no stall.
    const bool timed = (unsigned)logger_.level() >= (unsigned)LogLevel::INFO;
    if (timed && phase_events_.size() < (size_t)steps_ + 3) {
        const size_t have = phase_events_.size();
        phase_events_.resize((size_t)steps_ + 3, nullptr);
        for (size_t i = have; i < phase_events_.size(); ++i) SDOD_HIP_CHECK(hipEventCreate(&phase_events_[i]));
    }
    auto mark = [&](size_t i) {
        if (timed) SDOD_HIP_CHECK(hipEventRecord(phase_events_[i], stream_));
    };

    mark(0);
    // row 0 = conditional, row 1 = unconditional
    encode_prompt(prompt, static_cast<f16*>(uc.ptr));
    SDOD_HIP_CHECK(hipMemcpyAsync(static_cast<char*>(uc.ptr) + ctx_bytes, ctx_uncond_, ctx_bytes, hipMemcpyDeviceToDevice, stream_));
    mark(1);

    if (!injected_)
        for (auto& f : x_host_) f = normal_(rng_); // context.cpp:333-334
    injected_ = false;
    SDOD_HIP_CHECK(hipMemcpyAsync(x_dev_, x_host_.data(), lat * sizeof(float), hipMemcpyHostToDevice, stream_));

    // both batch rows see the same latent and time embedding (context.cpp:348-352, :364-366): one staging launch for the first
    // step; from then on the step's own launch stages the next one
    rc_check(sdod_stage_unet_inputs(x_dev_, static_cast<float*>(ux.ptr), lat, 2, temb_cache_, ut.ptr, temb_row / sizeof(f16), 2, stream_));
    for (unsigned step = 0; step < steps_; ++step) {
        unet_->execute(stream_, true, /*skip_static=*/step > 0); // the text context only changes between images
        // ONE launch: e = g*e_cond + (1-g)*e_uncond (context.cpp:359-373; g == 1 keeps e_cond only), the DPM-Solver++(2M) update
        // (dpm_solver.cpp:139-180) and the staging of the next step's inputs -- sdod_cfg_combine + sdod_dpm_update +
        // sdod_stage_unet_inputs, bit for bit
        const DpmSolver::StepCoef k = solver_->coef(step);
        sdod_dpm_step_args a{};
        a.eps_nhwc = ue.ptr; a.e_out = e_dev_; a.x = x_dev_; a.y_prev = y_prev_;
        a.n = 1; a.c = C; a.hw = HW; a.uncond_first = 0; a.mode = 0; a.order = k.order;
        a.guidance = guidance; a.sigma_s = k.sigma_s; a.alpha_s = k.alpha_s; a.sigma_ratio = k.sigma_ratio; a.c_prev = k.c_prev; a.c_cur = k.c_cur;
        if (step + 1 < steps_) {
            a.x_stage = static_cast<float*>(ux.ptr); a.stage_reps = 2;
            a.temb_row = reinterpret_cast<const char*>(temb_cache_) + (size_t)(step + 1) * temb_row;
            a.temb_dst = ut.ptr; a.temb_width = (int)(temb_row / sizeof(f16)); a.temb_reps = 2;
        }
        rc_check(sdod_dpm_step(&a, stream_));
        mark(2 + step);
    }

    const IoSlot vz = vae_->io(false, 0), vi = vae_->io(true, 0);
    SDOD_HIP_CHECK(hipMemcpyAsync(vz.ptr, x_dev_, lat * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    vae_->execute(stream_, true);
    // context.cpp:392-395: uint8(clamp(255*f, 0, 255)) with f = (decoded + 1)/2
    rc_check(sdod_image_to_u8(vi.ptr, img_u8_, image_bytes(), 0.5f, 0.5f, 0, stream_));
    SDOD_HIP_CHECK(hipMemcpyAsync(out, img_u8_, image_bytes(), hipMemcpyDeviceToHost, stream_));
    mark(2 + steps_);
    SDOD_HIP_CHECK(hipStreamSynchronize(stream_));
    unet_->check_health(); // a launch that could not report its own failure (GroupNorm grid barrier): no image is better than a wrong one

    if (timed) {
        auto between = [&](size_t a, size_t b) {
            float ms = 0.f;
            SDOD_HIP_CHECK(hipEventElapsedTime(&ms, phase_events_[a], phase_events_[b]));
            return (double)ms;
        };
        logger_.info("Conditioning took " + ms_str(between(0, 1)) + "ms");                 // context.cpp:331
        for (unsigned step = 0; step < steps_; ++step)                                      // context.cpp:381
            logger_.info("Single iteration took " + ms_str(between(1 + step, 2 + step)) + "ms");
        logger_.info("Decoding took " + ms_str(between(1 + steps_, 2 + steps_)) + "ms");   // context.cpp:398 (decode + uint8 + copy only)
    }
    logger_.info("Image successfully generated!");
    logger_.info("Image generation took " + ms_str(total.ms()) + "ms");                    // context.cpp:402 (host clock, whole call)
}

} // namespace sdod
