// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Thin extern "C" shim (our code) around the REFERENCE's own dpm_solver.cpp and
// tokenizer.cpp, which are compiled from where they lie under /root/reference by
// oracle/Makefile into oracle/_ref/libref.so.  No reference source is copied into
// this repository; this file only includes the reference headers at build time.
// Used to (1) validate oracle/sdod_oracle.c and oracle/tokenizer_oracle.py and
// (2) generate tests/golden/* (see oracle/gen_golden.py).
#include "dpm_solver.h"   // /root/reference/csrc/libsdod/src/dpm_solver.h
#include "tokenizer.h"    // /root/reference/csrc/libsdod/src/tokenizer.h

#include <cstring>
#include <string>
#include <vector>

extern "C" {

__attribute__((visibility("default"))) void* ref_dpm_create(unsigned timesteps, float lin_start, float lin_end) {
    return new libsdod::DPMSolver(timesteps, lin_start, lin_end);
}
__attribute__((visibility("default"))) void ref_dpm_destroy(void* h) { delete static_cast<libsdod::DPMSolver*>(h); }

// returns n = steps+1; model_ts out must hold steps+1 floats
__attribute__((visibility("default"))) unsigned ref_dpm_prepare(void* h, unsigned steps, float* model_ts) {
    auto* s = static_cast<libsdod::DPMSolver*>(h);
    std::vector<float> mts;
    s->prepare(steps, mts);
    std::memcpy(model_ts, mts.data(), mts.size() * sizeof(float));
    return static_cast<unsigned>(mts.size());
}

// which: 0 ts,1 log_alphas,2 lambdas,3 sigmas,4 alphas,5 phis,6 i2rs,8 all_t,9 all_log_alpha
__attribute__((visibility("default"))) unsigned ref_dpm_table(void* h, int which, float* out) {
    auto* s = static_cast<libsdod::DPMSolver*>(h);
    const std::vector<float>* v = nullptr;
    switch (which) {
    case 0: v = &s->get_ts(); break;
    case 1: v = &s->get_log_alphas(); break;
    case 2: v = &s->get_lambdas(); break;
    case 3: v = &s->get_sigmas(); break;
    case 4: v = &s->get_alphas(); break;
    case 5: v = &s->get_phis(); break;
    case 6: v = &s->get_i2rs(); break;
    case 8: v = &s->get_all_t(); break;
    case 9: v = &s->get_all_log_alpha(); break;
    default: return 0;
    }
    if (out) std::memcpy(out, v->data(), v->size() * sizeof(float));
    return static_cast<unsigned>(v->size());
}

// x, y are n floats; y enters as eps and is overwritten exactly as DPMSolver::update does
__attribute__((visibility("default"))) void ref_dpm_update(void* h, unsigned step, float* x, float* y, unsigned n) {
    auto* s = static_cast<libsdod::DPMSolver*>(h);
    std::vector<float> xv(x, x + n), yv(y, y + n);
    s->update(step, xv, yv);
    std::memcpy(x, xv.data(), n * sizeof(float));
    std::memcpy(y, yv.data(), n * sizeof(float));
}

__attribute__((visibility("default"))) void* ref_tok_create(const char* path) {
    try { return new libsdod::Tokenizer(path); } catch (...) { return nullptr; }
}
__attribute__((visibility("default"))) void ref_tok_destroy(void* h) { delete static_cast<libsdod::Tokenizer*>(h); }

// returns number of tokens written (== context_len) or -1 on exception
__attribute__((visibility("default"))) int ref_tok_tokenize(void* h, const char* text, unsigned short* out, unsigned context_len) {
    auto* t = static_cast<libsdod::Tokenizer*>(h);
    try {
        auto v = t->tokenize(std::string(text), context_len);
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
        return static_cast<int>(v.size());
    } catch (...) { return -1; }
}

}
