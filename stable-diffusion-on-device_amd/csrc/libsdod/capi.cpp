// C API of the generation driver (include/libsdod.h) and of its host pieces (include/sdod_host.h).
// Conventions follow the reference's libsdod.cpp: opaque handle {magic, version, ref_count, Context*}
// validated on every call (:22-27, :48-63), exceptions translated to status code + "func: reason [file:line]"
// message stored per context and per code (:29-45, :102-108), context-less table for setup / invalid handles.
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "context.h"
#include "libsdod.h"
#include "sdod_host.h"

namespace {

constexpr unsigned kMagic = 0x00534443; // "CDS\0", libsdod.cpp:16
constexpr unsigned kVersion = 1;        // libsdod.cpp:17

struct Handle {
    unsigned magic = kMagic;
    unsigned version = kVersion;
    unsigned ref_count = 0;
    sdod::Context* ctx = nullptr;
};

const char* const kDescriptions[6] = { // errors.cpp:8-15
    "No error", "Invalid context", "Invalid argument", "Failed to allocate memory or initialise an object",
    "Runtime error occurred", "Internal error occurred"};

std::string g_global_slots[6];
bool g_global_set[6] = {false, false, false, false, false, false};

int record(sdod::Context* c, int code, const std::string& msg) {
    if (code < 0 || code > 5) code = LIBSDOD_INTERNAL_ERROR;
    if (c) {
        c->error_slots[code] = msg;
        c->error_set[code] = true;
    } else {
        g_global_slots[code] = msg;
        g_global_set[code] = true;
    }
    return code;
}

std::string where(const char* func, const std::string& reason, const char* file, int line) {
    const char* base = std::strrchr(file, '/');
    return std::string(func) + ": " + reason + " [" + (base ? base + 1 : file) + ":" + std::to_string(line) + "]";
}

#define FAIL(ctx, code, reason) record(ctx, code, where(__func__, reason, __FILE__, __LINE__))

// validates the opaque handle exactly as TRY_RETRIEVE_CONTEXT does (libsdod.cpp:48-63)
int retrieve(void* context, Handle** out, const char* func) {
    *out = nullptr;
    auto bad = [&](const std::string& why) { return record(nullptr, LIBSDOD_INVALID_CONTEXT, where(func, why, __FILE__, __LINE__)); };
    if (context == nullptr) return bad("context is nullptr");
    Handle* h = static_cast<Handle*>(context);
    if (h->magic != kMagic) return bad("context magic header mismatch! got: " + std::to_string(h->magic));
    if (h->version != kVersion) return bad("context version mismatch! got: " + std::to_string(h->version));
    if (h->ref_count == 0) return bad("context has been released!");
    if (h->ctx == nullptr) return bad("corrupted context, internal pointer is nullptr");
    *out = h;
    return LIBSDOD_NO_ERROR;
}

template <typename F>
int guarded(sdod::Context* c, const char* func, F&& body) {
    try {
        body();
        return LIBSDOD_NO_ERROR;
    } catch (const sdod::Error& e) {
        return record(c, e.code, e.what());
    } catch (const std::bad_alloc& e) {
        return record(c, LIBSDOD_FAILED_ALLOCATION, where(func, e.what(), __FILE__, __LINE__));
    } catch (const std::invalid_argument& e) {
        return record(c, LIBSDOD_INVALID_ARGUMENT, where(func, e.what(), __FILE__, __LINE__));
    } catch (const std::exception& e) {
        return record(c, LIBSDOD_INTERNAL_ERROR, where(func, e.what(), __FILE__, __LINE__));
    } catch (...) {
        return record(c, LIBSDOD_INTERNAL_ERROR, where(func, "Unspecified error", __FILE__, __LINE__));
    }
}

} // namespace

extern "C" {

int libsdod_setup(void** context, const char* models_dir, unsigned int latent_channels, unsigned int latent_spatial,
                  unsigned int upscale_factor, unsigned int steps, unsigned int log_level, int use_htp) {
    if (context == nullptr) return FAIL(nullptr, LIBSDOD_INVALID_ARGUMENT, "Context argument should not be nullptr!");
    if (*context != nullptr) return FAIL(nullptr, LIBSDOD_INVALID_ARGUMENT, "Context should point to a nullptr-initialized variable!");
    if (log_level > LIBSDOD_LOG_ABUSIVE) return FAIL(nullptr, LIBSDOD_INVALID_ARGUMENT, "Invalid log_level");
    if (models_dir == nullptr) return FAIL(nullptr, LIBSDOD_INVALID_ARGUMENT, "models_dir is nullptr");
    Handle* h = new (std::nothrow) Handle;
    if (h == nullptr) return FAIL(nullptr, LIBSDOD_FAILED_ALLOCATION, "Could not create a new context handle");
    const int device = use_htp <= 1 ? 0 : use_htp - 1;
    const int rc = guarded(nullptr, __func__, [&]() {
        h->ctx = new sdod::Context(models_dir, latent_channels, latent_spatial, upscale_factor, static_cast<sdod::LogLevel>(log_level), device);
    });
    if (rc != LIBSDOD_NO_ERROR) {
        delete h; // nothing usable was created: *context stays NULL (fixes the leak noted as reference quirk Q10)
        return rc;
    }
    h->ref_count = 1;
    *context = h; // from here on the caller owns a handle, even if initialisation below fails
    return guarded(h->ctx, __func__, [&]() { h->ctx->init(steps); });
}

int libsdod_set_steps(void* context, unsigned int steps) {
    Handle* h;
    if (int rc = retrieve(context, &h, __func__)) return rc;
    return guarded(h->ctx, __func__, [&]() { h->ctx->prepare_schedule(steps); });
}

int libsdod_set_log_level(void* context, unsigned int log_level) {
    Handle* h;
    if (int rc = retrieve(context, &h, __func__)) return rc;
    if (log_level > LIBSDOD_LOG_ABUSIVE) return FAIL(h->ctx, LIBSDOD_INVALID_ARGUMENT, "Invalid log_level");
    h->ctx->logger().set_level(static_cast<sdod::LogLevel>(log_level));
    return LIBSDOD_NO_ERROR;
}

int libsdod_ref_context(void* context) {
    Handle* h;
    if (int rc = retrieve(context, &h, __func__)) return rc;
    ++h->ref_count;
    return LIBSDOD_NO_ERROR;
}

int libsdod_release(void* context) {
    Handle* h;
    if (int rc = retrieve(context, &h, __func__)) return rc;
    if (--h->ref_count == 0) {
        delete h->ctx;
        h->ctx = nullptr; // the handle stays allocated: a later call sees ref_count == 0 and reports INVALID_CONTEXT
    }
    return LIBSDOD_NO_ERROR;
}

int libsdod_generate_image(void* context, const char* prompt, float guidance_scale, unsigned char** image_out,
                           unsigned int* image_buffer_size) {
    Handle* h;
    if (int rc = retrieve(context, &h, __func__)) return rc;
    if (image_out == nullptr) return FAIL(h->ctx, LIBSDOD_INVALID_ARGUMENT, "image_out is nullptr");
    if (image_buffer_size == nullptr) return FAIL(h->ctx, LIBSDOD_INVALID_ARGUMENT, "image_buffer_size is nullptr");
    if (prompt == nullptr) return FAIL(h->ctx, LIBSDOD_INVALID_ARGUMENT, "prompt is nullptr");
    const size_t need = h->ctx->image_bytes();
    unsigned char* buf = *image_out;
    bool mine = false;
    if (buf == nullptr) {
        buf = static_cast<unsigned char*>(std::malloc(need));
        if (buf == nullptr) return FAIL(h->ctx, LIBSDOD_FAILED_ALLOCATION, "Could not allocate the output image");
        mine = true;
    } else if (*image_buffer_size < need) {
        return FAIL(h->ctx, LIBSDOD_INVALID_ARGUMENT, "Provided buffer is too small, missing " + std::to_string(need - *image_buffer_size) + " bytes");
    }
    const int rc = guarded(h->ctx, __func__, [&]() { h->ctx->generate(prompt, guidance_scale, buf); });
    if (rc != LIBSDOD_NO_ERROR) {
        if (mine) std::free(buf);
        return rc;
    }
    *image_out = buf; // ownership passes to the caller (libsdod.cpp:174-175: out.own(false))
    *image_buffer_size = static_cast<unsigned int>(need);
    return LIBSDOD_NO_ERROR;
}

const char* libsdod_get_error_description(int errorcode) {
    if (errorcode < 0 || errorcode > 5) return nullptr;
    return kDescriptions[errorcode];
}

const char* libsdod_get_last_error_extra_info(int errorcode, void* context) {
    if (errorcode < 0 || errorcode > 5) return nullptr;
    if (context != nullptr && errorcode != LIBSDOD_INVALID_CONTEXT) {
        Handle* h = static_cast<Handle*>(context);
        if (h->magic == kMagic && h->version == kVersion && h->ref_count > 0 && h->ctx != nullptr)
            return h->ctx->error_set[errorcode] ? h->ctx->error_slots[errorcode].c_str() : nullptr;
    }
    return g_global_set[errorcode] ? g_global_slots[errorcode].c_str() : nullptr;
}

// ------------------------------------------------------------------------------------------ sdod_host.h
#define HOST_TRY try {
#define HOST_CATCH                                                              \
    }                                                                           \
    catch (const sdod::Error& e) { sdod::set_last_error(e.what()); return e.code; } \
    catch (const std::invalid_argument& e) { sdod::set_last_error(e.what()); return LIBSDOD_INVALID_ARGUMENT; } \
    catch (const std::out_of_range& e) { sdod::set_last_error(e.what()); return LIBSDOD_INVALID_ARGUMENT; } \
    catch (const std::bad_alloc& e) { sdod::set_last_error(e.what()); return LIBSDOD_FAILED_ALLOCATION; } \
    catch (const std::exception& e) { sdod::set_last_error(e.what()); return LIBSDOD_INTERNAL_ERROR; } \
    catch (...) { sdod::set_last_error("unspecified error"); return LIBSDOD_INTERNAL_ERROR; }

#define HOST_REQUIRE(cond, msg) do { if (!(cond)) throw std::invalid_argument(msg); } while (0)

int sdod_tokenizer_create(void** tok, const char* path) {
    HOST_TRY
    HOST_REQUIRE(tok && path, "null argument");
    *tok = nullptr;
    *tok = new sdod::Tokenizer(path);
    return 0;
    HOST_CATCH
}

int sdod_tokenizer_destroy(void* tok) {
    delete static_cast<sdod::Tokenizer*>(tok);
    return 0;
}

int sdod_tokenizer_encode(void* tok, const char* utf8, uint16_t* ids_out, unsigned context_len) {
    HOST_TRY
    HOST_REQUIRE(tok && utf8 && ids_out, "null argument");
    const auto ids = static_cast<sdod::Tokenizer*>(tok)->tokenize(utf8, context_len);
    std::memcpy(ids_out, ids.data(), ids.size() * sizeof(uint16_t));
    return 0;
    HOST_CATCH
}

int sdod_tokenizer_special(void* tok, unsigned* start_token, unsigned* end_token, unsigned* vocab_size) {
    HOST_TRY
    HOST_REQUIRE(tok != nullptr, "null argument");
    auto* t = static_cast<sdod::Tokenizer*>(tok);
    if (start_token) *start_token = t->start_token();
    if (end_token) *end_token = t->end_token();
    if (vocab_size) *vocab_size = (unsigned)t->vocab_size();
    return 0;
    HOST_CATCH
}

int sdod_dpm_create(void** solver, unsigned timesteps, float lin_start, float lin_end) {
    HOST_TRY
    HOST_REQUIRE(solver != nullptr, "null argument");
    *solver = nullptr;
    *solver = new sdod::DpmSolver(timesteps, lin_start, lin_end);
    return 0;
    HOST_CATCH
}

int sdod_dpm_destroy(void* solver) {
    delete static_cast<sdod::DpmSolver*>(solver);
    return 0;
}

int sdod_dpm_prepare(void* solver, unsigned steps) {
    HOST_TRY
    HOST_REQUIRE(solver != nullptr, "null argument");
    static_cast<sdod::DpmSolver*>(solver)->prepare(steps);
    return 0;
    HOST_CATCH
}

int sdod_dpm_table(void* solver, int which, float* out, unsigned cap, unsigned* n) {
    HOST_TRY
    HOST_REQUIRE(solver != nullptr, "null argument");
    const auto& t = static_cast<sdod::DpmSolver*>(solver)->table(which);
    if (n) *n = (unsigned)t.size();
    if (out) std::memcpy(out, t.data(), std::min<size_t>(cap, t.size()) * sizeof(float));
    return 0;
    HOST_CATCH
}

int sdod_dpm_coef(void* solver, unsigned step, int* order, float* sigma_s, float* alpha_s, float* sigma_ratio, float* c_prev,
                  float* c_cur) {
    HOST_TRY
    HOST_REQUIRE(solver != nullptr, "null argument");
    const auto k = static_cast<sdod::DpmSolver*>(solver)->coef(step);
    if (order) *order = k.order;
    if (sigma_s) *sigma_s = k.sigma_s;
    if (alpha_s) *alpha_s = k.alpha_s;
    if (sigma_ratio) *sigma_ratio = k.sigma_ratio;
    if (c_prev) *c_prev = k.c_prev;
    if (c_cur) *c_cur = k.c_cur;
    return 0;
    HOST_CATCH
}

int sdod_dpm_update_host(void* solver, unsigned step, float* x, const float* eps, float* y_prev, unsigned n) {
    HOST_TRY
    HOST_REQUIRE(solver && x && eps && y_prev, "null argument");
    static_cast<sdod::DpmSolver*>(solver)->update_host(step, x, eps, y_prev, n);
    return 0;
    HOST_CATCH
}

int sdod_context_set_seed(void* libsdod_context, unsigned seed) {
    Handle* h;
    if (int rc = retrieve(libsdod_context, &h, __func__)) return rc;
    h->ctx->set_seed(seed);
    return LIBSDOD_NO_ERROR;
}

int sdod_context_set_initial_latent(void* libsdod_context, const float* x, size_t n) {
    Handle* h;
    if (int rc = retrieve(libsdod_context, &h, __func__)) return rc;
    return guarded(h->ctx, __func__, [&]() { h->ctx->set_initial_latent(x, n); });
}

} // extern "C"
