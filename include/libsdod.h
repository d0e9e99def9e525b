/*
 * libsdod.h -- generation-driver C API of the MI355X build: a drop-in for the reference's
 * csrc/libsdod/api/libsdod.h (vaenyr/stable-diffusion-on-device).  Same eight entry points, same enum
 * values, same handle / ownership / error conventions; every prototype below cites the reference line
 * it replaces.  A program written against the reference header links against lib/libsdod.so unchanged
 * (see INTEGRATION.md).
 *
 * Behavioural notes for this implementation:
 *   - models_dir must contain `ctokenizer.txt` (format of gen_tokenizer_file.py:27-42) and the weight
 *     containers `unet.sdodw`, `text_encoder.sdodw`, `vae_decoder.sdodw`, `temb.sdodw` (the reference
 *     loads unet.serialized.bin / text_encoder.serialized.bin / vae_decoder.serialized.bin / temb.bin,
 *     context.cpp:105-115 -- QNN blobs that cannot exist for this hardware);
 *   - `steps` may be any value in [1, 1000] (the reference rejects steps != 20, context.cpp:250);
 *   - `use_htp` selects the HIP device ordinal: values <= 1 mean device 0 (so the reference's 0/1 both
 *     work), a value n >= 2 means device n-1;
 *   - no call is thread-safe per context (same as the reference, libsdod.cpp:25).
 */
#ifndef LIBSDOD_H
#define LIBSDOD_H

#ifndef LIBSDOD_API
#define LIBSDOD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* reference libsdod.h:11-18 */
enum libsdod_status_code {
    LIBSDOD_NO_ERROR,
    LIBSDOD_INVALID_CONTEXT,
    LIBSDOD_INVALID_ARGUMENT,
    LIBSDOD_FAILED_ALLOCATION,
    LIBSDOD_RUNTIME_ERROR,
    LIBSDOD_INTERNAL_ERROR,
};

/* reference libsdod.h:21-27 */
enum libsdod_log_level {
   LIBSDOD_LOG_NOTHING,
   LIBSDOD_LOG_ERROR,
   LIBSDOD_LOG_INFO,
   LIBSDOD_LOG_DEBUG,
   LIBSDOD_LOG_ABUSIVE
};

/* reference libsdod.h:47 (impl libsdod.cpp:66-111).  *context must be NULL on entry; it may be set even when the
 * call fails, and must then still be released (and may be used to query error details, not to generate). */
LIBSDOD_API int libsdod_setup(void** context, const char* models_dir, unsigned int latent_channels, unsigned int latent_spatial, unsigned int upscale_factor, unsigned int steps, unsigned int log_level, int use_htp);

/* reference libsdod.h:57 (impl libsdod.cpp:113-126): re-prepare the schedule for a new number of steps */
LIBSDOD_API int libsdod_set_steps(void* context, unsigned int steps);

/* reference libsdod.h:67 (impl libsdod.cpp:128-144) */
LIBSDOD_API int libsdod_set_log_level(void* context, unsigned int log_level);

/* reference libsdod.h:75 (impl libsdod.cpp:146-150): one more release() needed per call */
LIBSDOD_API int libsdod_ref_context(void* context);

/* reference libsdod.h:81 (impl libsdod.cpp:152-161): the handle itself stays allocated so later misuse is detected */
LIBSDOD_API int libsdod_release(void* context);

/* reference libsdod.h:117 (impl libsdod.cpp:163-185).  *image_out == NULL: the library mallocs 3*(latent_spatial*
 * upscale_factor)^2 bytes and the caller frees them with free(); otherwise the caller's buffer of *image_buffer_size
 * bytes is reused (INVALID_ARGUMENT if too small).  On return *image_buffer_size = bytes written; layout [H][W][3] RGB. */
LIBSDOD_API int libsdod_generate_image(void* context, const char* prompt, float guidance_scale, unsigned char** image_out, unsigned int* image_buffer_size);

/* reference libsdod.h:124 (impl libsdod.cpp:187-192): NULL for an invalid code */
LIBSDOD_API const char* libsdod_get_error_description(int errorcode);

/* reference libsdod.h:138 (impl libsdod.cpp:194-209): per-context, per-code last message; context-less table when
 * context is NULL / invalid or errorcode is LIBSDOD_INVALID_CONTEXT */
LIBSDOD_API const char* libsdod_get_last_error_extra_info(int errorcode, void* context);

#ifdef __cplusplus
}
#endif
#endif /* LIBSDOD_H */
