/*
 * sdod_host.h -- C ABI of the host-side pieces of the generation driver that the reference keeps internal to
 * libsdod but whose results its tests pin (csrc/libsdod/test/test_dpm.cpp, test_tokenizer.cpp): the CLIP BPE
 * tokenizer and the DPM-Solver++ tables.  The Python sampler loop (sdod.amd.pipeline) binds these, so there is ONE
 * implementation of each, in C++, shared with libsdod_generate_image.
 */
#ifndef SDOD_HOST_H
#define SDOD_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifndef SDOD_API
#define SDOD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* replaces libsdod::Tokenizer(bpe_file) / ::tokenize (tokenizer.h:23-31, tokenizer.cpp:228-276) */
SDOD_API int sdod_tokenizer_create(void** tok, const char* ctokenizer_path);
SDOD_API int sdod_tokenizer_destroy(void* tok);
/* writes context_len uint16 ids (SOT, tokens, EOT padding); utf8 is NUL-terminated */
SDOD_API int sdod_tokenizer_encode(void* tok, const char* utf8, uint16_t* ids_out, unsigned context_len);
SDOD_API int sdod_tokenizer_special(void* tok, unsigned* start_token, unsigned* end_token, unsigned* vocab_size);

/* replaces libsdod::DPMSolver (dpm_solver.h:11-50): ctor :84-97, prepare :100-131 */
SDOD_API int sdod_dpm_create(void** solver, unsigned timesteps, float lin_start, float lin_end);
SDOD_API int sdod_dpm_destroy(void* solver);
SDOD_API int sdod_dpm_prepare(void* solver, unsigned steps);
/* which: 0 ts, 1 log_alphas, 2 lambdas, 3 sigmas, 4 alphas, 5 phis, 6 i2rs, 7 model_ts, 8 all_t, 9 all_log_alpha.
 * returns the table length through *n; copies min(*n, cap) floats when out != NULL */
SDOD_API int sdod_dpm_table(void* solver, int which, float* out, unsigned cap, unsigned* n);
/* coefficients of update(step) as consumed by sdod_dpm_update (include/sdod_hip.h); dpm_solver.cpp:136-181 */
SDOD_API int sdod_dpm_coef(void* solver, unsigned step, int* order, float* sigma_s, float* alpha_s, float* sigma_ratio,
                           float* c_prev, float* c_cur);
/* host version of the same update on n floats (x, y_prev in place) */
SDOD_API int sdod_dpm_update_host(void* solver, unsigned step, float* x, const float* eps, float* y_prev, unsigned n);

/* extension: reseed the latent generator of a libsdod context (the reference has Context::set_seed, context.cpp:285-289,
 * but no C entry point reaches it) */
SDOD_API int sdod_context_set_seed(void* libsdod_context, unsigned seed);
/* extension: inject x_T (fp32 NCHW, latent_channels*latent_spatial^2 values) for the NEXT libsdod_generate_image call
 * instead of drawing it from the generator (context.cpp:333-334); parity runs need this (SURVEY 7.2 "RNG") */
SDOD_API int sdod_context_set_initial_latent(void* libsdod_context, const float* x, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* SDOD_HOST_H */
