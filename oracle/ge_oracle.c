/* ge_oracle.c -- CPU oracle (plain C, fp32) for the holE.py hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library.  It restates the
 * arithmetic of /root/reference/holE.py; every function cites the lines it
 * follows.  PARITY UNPINNED by reference tests (the reference has none and
 * cannot run here); see oracle/hole_oracle.py for the pins that
 * do exist.  This C port is checked against the NumPy fp64 restatement in
 * tests/test_oracle.py and serves as the timed "port" CPU baseline.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC ge_oracle.c -o libge_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- a2: get_embedding (holE.py:161-168): clip scale = min(rsqrt(ss), 1/c)*c */
static inline float clip_scale(const float *x, int d, float max_norm, float *ss_out) {
    float ss = 0.f;
    for (int j = 0; j < d; ++j) ss += x[j] * x[j];
    *ss_out = ss;
    float inv = 1.0f / sqrtf(ss); /* rsqrt(0) = inf -> min picks 1/c */
    float lim = 1.0f / max_norm;
    return (inv < lim ? inv : lim) * max_norm;
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline int bad_id(int64_t N, int32_t a, int32_t b, int32_t c) {
    return a < 0 || b < 0 || c < 0 || a >= N || b >= N || c >= N;
}

/* ---- a3/a4: evaluate_triples (holE.py:179-198). triples = [B,3] (h,t,r). */
int oracle_complex_score(const float *table, int64_t N, int32_t d, const int32_t *triples,
                         int64_t B, float max_norm, int apply_sigmoid, float *out, int threads) {
    if (d <= 0 || (d & 1)) return -22;
    const int k = d / 2;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
#endif
    for (int64_t i = 0; i < B; ++i) {
        int32_t hi = triples[3 * i], ti = triples[3 * i + 1], ri = triples[3 * i + 2];
        if (bad_id(N, hi, ti, ri)) { out[i] = NAN; continue; }
        const float *h = table + (int64_t)hi * d, *t = table + (int64_t)ti * d, *r = table + (int64_t)ri * d;
        float ssh, sst, ssr;
        float sh = clip_scale(h, d, max_norm, &ssh), st = clip_scale(t, d, max_norm, &sst),
              sr = clip_scale(r, d, max_norm, &ssr);
        float s = 0.f;
        for (int j = 0; j < k; ++j) {
            float a = h[j], b = h[j + k], e = t[j], f = t[j + k], c = r[j], dd = r[j + k];
            /* Re(h r conj(t)) = a(ce+df) + b(cf-de)   (holE.py:191-192) */
            s += a * (c * e + dd * f) + b * (c * f - dd * e);
        }
        s *= sh * st * sr;
        out[i] = apply_sigmoid ? sigmoidf_(s) : s;
    }
    (void)threads;
    return 0;
}

/* ---- HolE (README.md:42): s = sum_k r_k sum_i h_i t_{(i+k)%d}, O(d^2) definition */
static float hole_raw(const float *h, const float *t, const float *r, int d) {
    float s = 0.f;
    for (int kk = 0; kk < d; ++kk) {
        float c = 0.f;
        for (int i = 0; i < d; ++i) { int j = i + kk; if (j >= d) j -= d; c += h[i] * t[j]; }
        s += r[kk] * c;
    }
    return s;
}

int oracle_hole_score(const float *table, int64_t N, int32_t d, const int32_t *triples,
                      int64_t B, float max_norm, int apply_sigmoid, float *out, int threads) {
    if (d <= 0) return -22;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
#endif
    for (int64_t i = 0; i < B; ++i) {
        int32_t hi = triples[3 * i], ti = triples[3 * i + 1], ri = triples[3 * i + 2];
        if (bad_id(N, hi, ti, ri)) { out[i] = NAN; continue; }
        const float *h = table + (int64_t)hi * d, *t = table + (int64_t)ti * d, *r = table + (int64_t)ri * d;
        float ss;
        float sc = clip_scale(h, d, max_norm, &ss) * clip_scale(t, d, max_norm, &ss) * clip_scale(r, d, max_norm, &ss);
        float s = hole_raw(h, t, r, d) * sc;
        out[i] = apply_sigmoid ? sigmoidf_(s) : s;
    }
    (void)threads;
    return 0;
}

/* ---- a6: gradient through the clip (MinimumGrad routes to rsqrt iff rsqrt <= 1/c) */
static void clip_backward(const float *x, float ss, float max_norm, const float *gy, float *gx, int d) {
    float inv = 1.0f / sqrtf(ss);
    if (inv <= 1.0f / max_norm) {
        float dot = 0.f;
        for (int j = 0; j < d; ++j) dot += gy[j] * x[j];
        float inv3 = inv * inv * inv;
        for (int j = 0; j < d; ++j) gx[j] = max_norm * (gy[j] * inv - x[j] * dot * inv3);
    } else {
        memcpy(gx, gy, sizeof(float) * d);
    }
}

/* one side of one pair: forward value and (unscaled-by-coef) raw-row gradients */
typedef struct { float sig; } side_fwd;

static float side_forward_backward(const float *table, int32_t d, const int32_t *tr, float max_norm,
                                   int hole, float *yh, float *yt, float *yr, float *ssq) {
    const float *h = table + (int64_t)tr[0] * d, *t = table + (int64_t)tr[1] * d, *r = table + (int64_t)tr[2] * d;
    float sh = clip_scale(h, d, max_norm, &ssq[0]), st = clip_scale(t, d, max_norm, &ssq[1]),
          sr = clip_scale(r, d, max_norm, &ssq[2]);
    for (int j = 0; j < d; ++j) { yh[j] = h[j] * sh; yt[j] = t[j] * st; yr[j] = r[j] * sr; }
    float s = 0.f;
    if (!hole) {
        int k = d / 2;
        for (int j = 0; j < k; ++j)
            s += yh[j] * (yr[j] * yt[j] + yr[j + k] * yt[j + k]) + yh[j + k] * (yr[j] * yt[j + k] - yr[j + k] * yt[j]);
    } else {
        s = hole_raw(yh, yt, yr, d);
    }
    return s;
}

static void side_grads(const float *yh, const float *yt, const float *yr, int d, int hole, float coef,
                       float *gh, float *gt, float *gr) {
    if (!hole) {
        int k = d / 2;
        for (int j = 0; j < k; ++j) {
            float a = yh[j], b = yh[j + k], e = yt[j], f = yt[j + k], c = yr[j], dd = yr[j + k];
            gh[j] = coef * (c * e + dd * f); gh[j + k] = coef * (c * f - dd * e);
            gt[j] = coef * (a * c - b * dd); gt[j + k] = coef * (a * dd + b * c);
            gr[j] = coef * (a * e + b * f);  gr[j + k] = coef * (a * f - b * e);
        }
    } else {
        for (int m = 0; m < d; ++m) {
            float ar = 0.f, ah = 0.f, at = 0.f;
            for (int i = 0; i < d; ++i) {
                int j = i + m; if (j >= d) j -= d;   /* (i+m)%d */
                int q = m - i; if (q < 0) q += d;    /* (m-i)%d */
                ar += yh[i] * yt[j];                 /* d s/d r_m = (h star t)_m */
                ah += yr[i] * yt[j];                 /* d s/d h_m = sum_k r_k t_{m+k} */
                at += yr[i] * yh[q];                 /* d s/d t_m = sum_k r_k h_{m-k} */
            }
            gr[m] = coef * ar; gh[m] = coef * ah; gt[m] = coef * at;
        }
    }
}

/* ---- a5+a6+a7: one training step (holE.py:222-234, 296): hinge, grad of the SUM,
 * ScatterSub with duplicates accumulating.  All gradients are computed against the
 * table as it was BEFORE the step (TF computes every IndexedSlices value before the
 * ScatterSub), then applied in the graph's concat order r+,r-,t+,t-,h+,h-. */
int oracle_hinge_step(float *table, int64_t N, int32_t d, const int32_t *pos, const int32_t *neg,
                      int64_t B, float margin, float lr, float max_norm, int hole, float *loss,
                      int threads) {
    if (d <= 0 || (!hole && (d & 1))) return -22;
    /* the gradient buffer is kept between calls: a fresh 20 MB malloc per step is first-touched by
     * every thread at once, and the page-fault storm made 8 threads slower than 1 */
    static float *G = NULL;
    static size_t G_cap = 0;
    const size_t need = sizeof(float) * 6 * (size_t)B * d;
    if (need > G_cap) {
        free(G);
        G = (float *)malloc(need);
        G_cap = G ? need : 0;
        if (!G) return -12;
        memset(G, 0, need);
    }
    const int T = threads > 0 ? threads : 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(T)
#endif
    {
        float buf[9 * 1024];
        float *heap = d > 1024 ? (float *)malloc(sizeof(float) * 9 * (size_t)d) : NULL;
        float *b0 = heap ? heap : buf;
        float *yh = b0, *yt = b0 + d, *yr = b0 + 2 * d, *gh = b0 + 3 * d, *gt = b0 + 4 * d, *gr = b0 + 5 * d;
        float *yh2 = b0 + 6 * d, *yt2 = b0 + 7 * d, *yr2 = b0 + 8 * d;
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t i = 0; i < B; ++i) {
            const int32_t *p = pos + 3 * i, *n = neg + 3 * i;
            float *g = G + (size_t)i * 6 * d; /* rows: r+, r-, t+, t-, h+, h- */
            if (bad_id(N, p[0], p[1], p[2]) || bad_id(N, n[0], n[1], n[2])) {
                loss[i] = NAN; memset(g, 0, sizeof(float) * 6 * d); continue;
            }
            float ssp[3], ssn[3];
            float sp = sigmoidf_(side_forward_backward(table, d, p, max_norm, hole, yh, yt, yr, ssp));
            float sn = sigmoidf_(side_forward_backward(table, d, n, max_norm, hole, yh2, yt2, yr2, ssn));
            float pre = sp - sn + margin;
            float m = pre >= 0.f ? 1.f : 0.f; /* MaximumGrad: x >= y */
            loss[i] = pre > 0.f ? pre : 0.f;
            side_grads(yh, yt, yr, d, hole, m * sp * (1.f - sp), gh, gt, gr);
            clip_backward(table + (int64_t)p[2] * d, ssp[2], max_norm, gr, g + 0 * d, d);
            clip_backward(table + (int64_t)p[1] * d, ssp[1], max_norm, gt, g + 2 * d, d);
            clip_backward(table + (int64_t)p[0] * d, ssp[0], max_norm, gh, g + 4 * d, d);
            side_grads(yh2, yt2, yr2, d, hole, -m * sn * (1.f - sn), gh, gt, gr);
            clip_backward(table + (int64_t)n[2] * d, ssn[2], max_norm, gr, g + 1 * d, d);
            clip_backward(table + (int64_t)n[1] * d, ssn[1], max_norm, gt, g + 3 * d, d);
            clip_backward(table + (int64_t)n[0] * d, ssn[0], max_norm, gh, g + 5 * d, d);
        }
        free(heap);
        /* ScatterSub in concat order (slot-major).  Every thread walks the whole index list but applies
         * only the rows it owns (row % T == tid): per row the order of the subtractions is exactly the
         * serial one, so the result does not depend on the thread count.  (The implicit barrier of the
         * omp for above guarantees all gradients were computed against the pre-step table.) */
        static const int col_of_slot[6] = {2, 2, 1, 1, 0, 0};
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nth = omp_get_num_threads();
#else
        const int tid = 0, nth = 1;
#endif
        for (int slot = 0; slot < 6; ++slot) {
            const int32_t *src = (slot & 1) ? neg : pos;
            for (int64_t i = 0; i < B; ++i) {
                if (loss[i] != loss[i]) continue;
                const int32_t r = src[3 * i + col_of_slot[slot]];
                if (r % nth != tid) continue;
                float *row = table + (int64_t)r * d;
                const float *g = G + ((size_t)i * 6 + slot) * d;
                for (int j = 0; j < d; ++j) row[j] -= lr * g[j];
            }
        }
    }
    return 0;
}

/* ---- a9/a10: type-safe corruption, Philox4x32-10 stream defined in hole_oracle.py */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
#define TAG_COIN 0x636F696Eu
#define TAG_SLOT 0x736C6F74u
#define TAG_PICK 0x7069636Bu
#define TAG_SIDE 0x73696465u

int oracle_corrupt_batch(const int32_t *pos, int64_t B, const int32_t *id_to_type, int64_t N,
                         const int64_t *type_offsets, int32_t n_types, const int32_t *type_ids,
                         uint64_t seed, uint64_t step, int32_t padded_size, int32_t mode,
                         int32_t *neg) {
    uint32_t slo = (uint32_t)step, shi = (uint32_t)(step >> 32), klo = (uint32_t)seed, khi = (uint32_t)(seed >> 32);
    int batch_heads = 0;
    if (mode == 0) { uint32_t c[4] = {slo, shi, 0, 0}; philox4x32_10(c, klo ^ TAG_COIN, khi); batch_heads = (c[0] >> 31) == 0; }
    for (int64_t i = 0; i < B; ++i) {
        int heads;
        if (mode == 0) heads = batch_heads;
        else if (mode == 1) { uint32_t c[4] = {slo, shi, (uint32_t)i, (uint32_t)((uint64_t)i >> 32)}; philox4x32_10(c, klo ^ TAG_SIDE, khi); heads = (c[0] >> 31) == 0; }
        else heads = (mode == 2);
        int col = heads ? 0 : 1;
        int32_t x = pos[3 * i + col];
        int32_t out = -1; /* unknown id -> default row of -1s (holE.py:39) */
        if (x >= 0 && x < N) {
            int32_t ty = id_to_type[x];
            if (ty >= 0 && ty < n_types) {
                int64_t off = type_offsets[ty];
                uint64_t len = (uint64_t)(type_offsets[ty + 1] - off);
                if (len > 0) {
                    uint32_t c[4] = {slo, shi, (uint32_t)i, (uint32_t)((uint64_t)i >> 32)};
                    philox4x32_10(c, klo ^ TAG_SLOT, khi);
                    uint32_t w = c[0];
                    if (padded_size > 0) {
                        uint32_t slot = w % (uint32_t)padded_size;
                        uint32_t q[4] = {slo, shi, (uint32_t)ty, slot};
                        philox4x32_10(q, klo ^ TAG_PICK, khi);
                        w = q[0];
                    }
                    out = type_ids[off + (int64_t)(((uint64_t)w * len) >> 32)];
                }
            }
        }
        neg[3 * i] = pos[3 * i]; neg[3 * i + 1] = pos[3 * i + 1]; neg[3 * i + 2] = pos[3 * i + 2];
        neg[3 * i + col] = out;
    }
    return 0;
}

/* ---- 1-vs-K scoring (holE.py:564-569 inference shape: fixed (h,r), many tails) */
int oracle_complex_score_1vK(const float *table, int64_t N, int32_t d, const int32_t *hr, int64_t B,
                             const int32_t *cand, int64_t K, float max_norm, int apply_sigmoid,
                             int cand_is_head, float *out, int threads) {
    if (d <= 0 || (d & 1)) return -22;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
#endif
    for (int64_t i = 0; i < B; ++i) {
        for (int64_t c = 0; c < K; ++c) {
            int32_t tr[3];
            if (cand_is_head) { tr[0] = cand[c]; tr[1] = hr[2 * i]; } else { tr[0] = hr[2 * i]; tr[1] = cand[c]; }
            tr[2] = hr[2 * i + 1];
            oracle_complex_score(table, N, d, tr, 1, max_norm, apply_sigmoid, out + i * K + c, 1);
        }
    }
    (void)threads;
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
