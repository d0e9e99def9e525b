/*
 * oracle/sdod_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C CPU restatement of the host-side arithmetic of the reference's
 * generation driver (vaenyr/stable-diffusion-on-device, csrc/libsdod).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Each function cites the reference file:line it follows.
 *
 * Pinning: dpm_* is checked bit-for-bit against the reference's own
 * dpm_solver.cpp compiled in place (oracle/_ref/libref_dpm.so, see
 * oracle/Makefile) and against tests/golden/dpm_*.json generated from it.
 * temb / cfg / dequant / to_uint8 restate context.cpp / qnn_context.cpp
 * fragments that cannot be compiled here (QNN SDK headers absent): the
 * reference holds no test for them -> "parity unpinned" by reference tests,
 * pinned by hand-derivable known answers in tests/test_oracle_host.py.
 *
 * float/double mixing below is deliberate: it mirrors the reference
 * (value_type = float, dpm_solver.h:13; double step in linspace,
 * dpm_solver.cpp:15; double cumulative product, dpm_solver.cpp:91).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

/* dpm_solver.cpp:12-26  linspace<float>(buffer, start, end, num_steps, offset) */
static void linspace_f(float *buf, float start, float end, unsigned num_steps, unsigned offset) {
    double step = (double)(end - start) / (double)(num_steps - 1);
    unsigned insert = 0;
    for (unsigned i = 0; i < num_steps; ++i) {
        if (!offset)
            buf[insert++] = start;
        else
            --offset;
        start = (float)((double)start + step); /* float += double */
    }
}

/* dpm_solver.cpp:29-32 */
static float interp2(float x, float x1, float y1, float x2, float y2) {
    float a = (y2 - y1) / (x2 - x1);
    return a * (x - x1) + y1;
}

typedef struct {
    unsigned total_timesteps;
    float *all_t;         /* [T]   dpm_solver.cpp:85 */
    float *all_log_alpha; /* [T]   dpm_solver.cpp:88-96 */
    unsigned n;           /* steps+1 */
    float *ts, *log_alphas, *lambdas, *sigmas, *alphas, *phis, *i2rs, *model_ts;
    float *prev_y;
    unsigned prev_n;
} oracle_dpm;

/* dpm_solver.cpp:84-97  DPMSolver::DPMSolver */
ORACLE_API void *oracle_dpm_create(unsigned timesteps, float lin_start, float lin_end) {
    oracle_dpm *s = (oracle_dpm *)calloc(1, sizeof(oracle_dpm));
    s->total_timesteps = timesteps;
    s->all_t = (float *)malloc(sizeof(float) * timesteps);
    s->all_log_alpha = (float *)malloc(sizeof(float) * timesteps);
    linspace_f(s->all_t, 0.0f, 1.0f, timesteps + 1, 1);
    linspace_f(s->all_log_alpha, sqrtf(lin_start), sqrtf(lin_end), timesteps, 0);
    double cum = 1.0;
    for (unsigned i = 0; i < timesteps; ++i) {
        float b = s->all_log_alpha[i];
        b = 1 - b * b;
        cum *= b;
        s->all_log_alpha[i] = (float)(0.5 * log(cum));
    }
    return s;
}

ORACLE_API void oracle_dpm_destroy(void *h) {
    oracle_dpm *s = (oracle_dpm *)h;
    if (!s) return;
    free(s->all_t); free(s->all_log_alpha);
    free(s->ts); free(s->log_alphas); free(s->lambdas); free(s->sigmas);
    free(s->alphas); free(s->phis); free(s->i2rs); free(s->model_ts); free(s->prev_y);
    free(s);
}

/* dpm_solver.cpp:35-54  interpolate(x, xs, ys, hint); xs ascending, x descending over calls */
static float interpolate_f(float x, const float *xs, const float *ys, unsigned n, unsigned *hint) {
    if (x < xs[0] || x > xs[n - 1])
        return interp2(x, xs[n - 1], ys[n - 1], xs[0], ys[0]);
    while (xs[*hint - 1] > x)
        --*hint;
    /* reference asserts hint < n here (dpm_solver.cpp:47); x == xs[n-1] with the initial
       hint == n would read xs[n] -- guarded: treat as the last segment end point */
    if (*hint >= n)
        return ys[n - 1];
    return interp2(x, xs[*hint - 1], ys[*hint - 1], xs[*hint], ys[*hint]);
}

/* dpm_solver.cpp:100-131  DPMSolver::prepare */
ORACLE_API void oracle_dpm_prepare(void *h, unsigned steps) {
    oracle_dpm *s = (oracle_dpm *)h;
    unsigned n = steps + 1;
    float **arrs[] = {&s->ts, &s->log_alphas, &s->lambdas, &s->sigmas, &s->alphas, &s->phis, &s->i2rs, &s->model_ts};
    for (unsigned k = 0; k < 8; ++k) {
        free(*arrs[k]);
        *arrs[k] = (float *)malloc(sizeof(float) * n);
    }
    s->n = n;
    float first_t = 1.0f;
    float last_t = (float)(1.0 / s->total_timesteps);
    linspace_f(s->ts, first_t, last_t, n, 0);
    unsigned hint = s->total_timesteps;
    for (unsigned i = 0; i < n; ++i) {
        s->model_ts[i] = (float)(((double)s->ts[i] - 1.0 / s->total_timesteps) * 1000);
        s->log_alphas[i] = interpolate_f(s->ts[i], s->all_t, s->all_log_alpha, s->total_timesteps, &hint);
        float la = s->log_alphas[i];
        s->lambdas[i] = (float)((double)la - (0.5 * (double)logf(1 - expf(2 * la))));
        s->sigmas[i] = sqrtf(1 - expf(2 * la));
        s->alphas[i] = expf(la);
        if (i)
            s->phis[i] = expm1f(-(s->lambdas[i] - s->lambdas[i - 1]));
        else
            s->phis[i] = INFINITY;
        if (i >= 2)
            s->i2rs[i] = (float)(1.0 / (double)(2 * ((s->lambdas[i - 1] - s->lambdas[i - 2]) / (s->lambdas[i] - s->lambdas[i - 1]))));
        else
            s->i2rs[i] = INFINITY;
    }
}

/* which: 0 ts,1 log_alphas,2 lambdas,3 sigmas,4 alphas,5 phis,6 i2rs,7 model_ts,8 all_t,9 all_log_alpha */
ORACLE_API unsigned oracle_dpm_table(void *h, int which, float *out) {
    oracle_dpm *s = (oracle_dpm *)h;
    const float *src = NULL;
    unsigned n = s->n;
    switch (which) {
    case 0: src = s->ts; break;
    case 1: src = s->log_alphas; break;
    case 2: src = s->lambdas; break;
    case 3: src = s->sigmas; break;
    case 4: src = s->alphas; break;
    case 5: src = s->phis; break;
    case 6: src = s->i2rs; break;
    case 7: src = s->model_ts; break;
    case 8: src = s->all_t; n = s->total_timesteps; break;
    case 9: src = s->all_log_alpha; n = s->total_timesteps; break;
    default: return 0;
    }
    if (out && src) memcpy(out, src, sizeof(float) * n);
    return n;
}

/* dpm_solver.cpp:136-181  DPMSolver::update(step, x, y): y enters as eps, is overwritten */
ORACLE_API void oracle_dpm_update(void *h, unsigned step, float *x, float *y, unsigned n) {
    oracle_dpm *s = (oracle_dpm *)h;
    /* :137 -- degenerates to order 1 at step 0, order 2 afterwards */
    unsigned order = (step == 0 ? 1u : (step < 10 ? (2u < s->n - step ? 2u : s->n - step) : 2u));
    /* :139 normalize(y, x, y, -sigma, alpha): y = (x + (-sigma)*y)/alpha */
    {
        float a = -s->sigmas[step], b = s->alphas[step];
        for (unsigned i = 0; i < n; ++i) y[i] = (x[i] + a * y[i]) / b;
    }
    float sc = s->sigmas[step + 1] / s->sigmas[step];
    for (unsigned i = 0; i < n; ++i) x[i] *= sc; /* :153 / :168 */
    if (order == 1) {
        float a = -s->alphas[step + 1] * s->phis[step + 1]; /* :154 */
        for (unsigned i = 0; i < n; ++i) x[i] += a * y[i];
    } else {
        float a1 = s->alphas[step + 1] * s->phis[step + 1] * s->i2rs[step + 1];        /* :169 */
        float a2 = -s->alphas[step + 1] * s->phis[step + 1] * (1 + s->i2rs[step + 1]); /* :170 */
        for (unsigned i = 0; i < n; ++i) x[i] += a1 * s->prev_y[i];
        for (unsigned i = 0; i < n; ++i) x[i] += a2 * y[i];
    }
    /* :177-180  first call copies, later calls swap */
    if (!s->prev_y) {
        s->prev_y = (float *)malloc(sizeof(float) * n);
        s->prev_n = n;
        memcpy(s->prev_y, y, sizeof(float) * n);
    } else {
        for (unsigned i = 0; i < n; ++i) { float t = y[i]; y[i] = s->prev_y[i]; s->prev_y[i] = t; }
    }
}

/* context.cpp:257-274  sinusoidal timestep features: mode[j]=cos(t*e^{-ln(1e4) j/half}), mode[half+j]=sin(.) */
ORACLE_API void oracle_timestep_features(float t, unsigned mode_dim, float *mode) {
    const float max_period = 10000.0f;
    float log_period = -logf(max_period);
    unsigned half = mode_dim / 2;
    for (unsigned j = 0; j < half; ++j) {
        float arg = t * expf(log_period * j / half);
        mode[j] = cosf(arg);
        mode[half + j] = sinf(arg);
    }
}

/* context.cpp:359-373 + qnn_context.cpp:1065-1081 (simple_cast<Accum,Scale>):
   e = g*e_cond ; e += (1-g)*e_uncond   (fp32 graph outputs) */
ORACLE_API void oracle_cfg_combine(float *e, const float *e_cond, const float *e_uncond, float g, unsigned n) {
    if (g == 1.0f) { /* context.cpp:359-360 */
        memcpy(e, e_cond, sizeof(float) * n);
        return;
    }
    for (unsigned i = 0; i < n; ++i) e[i] = e_cond[i] * g;
    float s = 1 - g;
    for (unsigned i = 0; i < n; ++i) e[i] += e_uncond[i] * s;
}

/* qnn_context.cpp:1018-1033  tf2any<Accum,Scale,float,uint8_t>: real = (q + offset) * scale in double */
ORACLE_API void oracle_dequant_u8(float *out, const uint8_t *in, int32_t offset, float scale, unsigned n,
                                  int accum, int use_scale, float accum_scale) {
    double offset_d = (double)offset;
    for (unsigned i = 0; i < n; ++i) {
        double quant = (double)in[i];
        float v;
        if (use_scale)
            v = accum_scale * (float)((quant + offset_d) * scale);
        else
            v = (float)((quant + offset_d) * scale);
        if (accum) out[i] += v; else out[i] = v;
    }
}

/* context.cpp:392-395  uint8(clamp(255*f, 0, 255)) -- truncating cast */
ORACLE_API void oracle_to_uint8(uint8_t *out, const float *img, unsigned n) {
    for (unsigned i = 0; i < n; ++i) {
        float f = 255 * img[i];
        if (f < 0.0f) f = 0.0f;
        if (f > 255.0f) f = 255.0f;
        out[i] = (uint8_t)f;
    }
}
