/* capi_consumer.c -- a plain-C program written against include/libsdod.h only, the way an application written against
 * the reference's csrc/libsdod/api/libsdod.h is (its contract: csrc/libsdod/test/simple_app.cpp:7-36 -- setup, one
 * generate into a library-allocated buffer, write output.bin, release).  Built by tests/test_capi_consumer.py with
 *     gcc -std=c99 -Wall -Werror -I include tests/capi_consumer.c -L<lib> -lsdod
 * and run as a fresh process.  Beyond the sample app it also checks the conventions a C caller relies on:
 * free() of the library's buffer (libsdod.h:84-116), reuse of a caller buffer, a too-small buffer, ref/release counting,
 * use after release.
 *
 * usage: capi_consumer <models_dir> <latent_spatial> <steps> <output.bin> [prompt]
 * exit: 0 ok; 1 setup failed (message on stdout); 2 generate failed; 3+ a convention check failed */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libsdod.h"

static int report(const char* what, int status, void* ctx) {
    const char* desc = libsdod_get_error_description(status);
    const char* extra = libsdod_get_last_error_extra_info(status, ctx);
    printf("%s error: %s; %s\n", what, desc ? desc : "(unknown status)", extra ? extra : "(no details)");
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <models_dir> <latent_spatial> <steps> <output.bin> [prompt]\n", argv[0]);
        return 64;
    }
    const unsigned spatial = (unsigned)strtoul(argv[2], NULL, 10), steps = (unsigned)strtoul(argv[3], NULL, 10);
    const char* prompt = argc > 5 ? argv[5] : "A photograph of an astronaut riding a horse";
    void* ctx = NULL;
    int status = libsdod_setup(&ctx, argv[1], 4, spatial, 8, steps, LIBSDOD_LOG_ERROR, 1);
    if (status) {
        report("Initialization", status, ctx);
        if (ctx) libsdod_release(ctx); /* *context may be set on failure and must then still be released */
        return 1;
    }

    unsigned char* img = NULL;
    unsigned int img_len = 0;
    status = libsdod_generate_image(ctx, prompt, 7.5f, &img, &img_len);
    if (status) {
        report("Generation", status, ctx);
        libsdod_release(ctx);
        return 2;
    }
    const unsigned int want = 3u * (spatial * 8u) * (spatial * 8u);
    if (img == NULL || img_len != want) return 3;
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(img, 1, img_len, f) != img_len) return 4;
    fclose(f);

    /* second image into a caller-owned buffer: reused, size reported back */
    unsigned char* mine = (unsigned char*)malloc(want + 16);
    unsigned char* slot = mine;
    unsigned int cap = want + 16;
    memset(mine, 0xA5, cap);
    status = libsdod_generate_image(ctx, prompt, 7.5f, &slot, &cap);
    if (status || slot != mine || cap != want) return 5;
    if (mine[want] != 0xA5) return 6; /* nothing written past the image */
    /* a buffer that is too small is an argument error and is left alone */
    unsigned int tiny = 16;
    slot = mine;
    status = libsdod_generate_image(ctx, prompt, 7.5f, &slot, &tiny);
    if (status != LIBSDOD_INVALID_ARGUMENT || libsdod_get_last_error_extra_info(status, ctx) == NULL) return 7;
    free(mine);
    free(img); /* the library's buffer is malloc()ed: the caller frees it with free() */

    /* reference counting: one extra reference needs one extra release; afterwards the handle is detectably dead */
    if (libsdod_ref_context(ctx) != LIBSDOD_NO_ERROR) return 8;
    if (libsdod_release(ctx) != LIBSDOD_NO_ERROR || libsdod_release(ctx) != LIBSDOD_NO_ERROR) return 9;
    if (libsdod_release(ctx) != LIBSDOD_INVALID_CONTEXT) return 10;
    img = NULL;
    if (libsdod_generate_image(ctx, prompt, 7.5f, &img, &img_len) != LIBSDOD_INVALID_CONTEXT) return 11;
    printf("capi_consumer ok: %u bytes\n", want);
    return 0;
}
