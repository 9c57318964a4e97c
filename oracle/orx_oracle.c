/* CPU oracle, C restatement (OpenMP) of the OpenRec tf2 pairwise train step.
 *
 * TEST INFRASTRUCTURE ONLY -- used by tests/ (cross-checked against
 * oracle/numpy_oracle.py) and by bench.py's `cpu_baseline` leg ("kind":"port").
 * Never linked into or called from the product library.
 *
 * PARITY UNPINNED: see the header of oracle/numpy_oracle.py.  TensorFlow, which
 * holds the reference's arithmetic, is absent; this file restates
 *   openrec/tf2/recommenders/bpr.py:21-37, ucml.py:21-42,
 *   openrec/tf2/modules/pairwise_log_loss.py:15-34,
 *   tf2_examples/bpr_citeulike.py:33-39 (tape over (loss, l2_loss) + apply)
 * with TF-2.0 Keras SGD / Adagrad / Adam sparse-apply semantics (snapshot gradients,
 * SGD accumulates every occurrence, Adagrad and Adam sum duplicates first, Adam decays and
 * moves every row of a table at every step).
 *
 * Build:  gcc -O3 -march=native -fopenmp -shared -fPIC orx_oracle.c -o _build/liborx_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORC_SGD = 0, ORC_ADAGRAD = 1 };
enum { ORC_BPR = 0, ORC_UCML = 1 };

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float log_sigmoidf(float x) {      /* -softplus(-x), stable */
    return -(fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x))));
}
static inline float sigmoidf_(float x) {
    float e = expf(-fabsf(x));
    return x >= 0 ? 1.f / (1.f + e) : e / (1.f + e);
}

/* weight of l2_loss in the differentiated objective: 1 = tape over the tuple (loss, l2_loss) as in
 * tf2_examples/bpr_citeulike.py:35-37; 0 = tape over `loss` alone (the library's ORX_NO_L2).  The l2 VALUE is
 * reported either way. */
static float g_l2w = 1.f;
void orc_set_l2w(float w) { g_l2w = w; }

/* scratch layout: gu,gp,gn [B*D] each, gb [B] -> (3*B*D + B) floats */

/* forward + per-occurrence gradients on the PRE-step tables (objective loss + l2_loss) */
static void forward_grads(int model, const float* U, const float* V, const float* b, int D,
                          const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t B, float margin,
                          float* gu, float* gp, float* gn, float* gb, double* out) {
    double loss = 0.0, l2 = 0.0;
    const float invB = 1.0f / (float)B;
    const float l2w = g_l2w;

#pragma omp parallel for reduction(+ : loss, l2) schedule(static)
    for (int64_t k = 0; k < B; ++k) {
        const float* u = U + (size_t)uid[k] * D;
        const float* p = V + (size_t)pid[k] * D;
        const float* n = V + (size_t)nid[k] * D;
        float s = 0.f, sq = 0.f, g;
        if (model == ORC_BPR) {
            float dp = 0.f, dn = 0.f;
            for (int d = 0; d < D; ++d) { dp += u[d] * p[d]; dn += u[d] * n[d]; sq += u[d] * u[d] + p[d] * p[d] + n[d] * n[d]; }
            s = (dp + b[pid[k]]) - (dn + b[nid[k]]);
            float m = fmaxf(s, -30.f);
            loss += -(double)log_sigmoidf(m) * invB;
            g = (s >= -30.f) ? -sigmoidf_(-s) * invB : 0.f;
            for (int d = 0; d < D; ++d) {
                gu[k * D + d] = g * (p[d] - n[d]) + l2w * u[d];
                gp[k * D + d] = g * u[d] + l2w * p[d];
                gn[k * D + d] = -g * u[d] + l2w * n[d];
            }
            gb[k] = g;                               /* d/db[p] = g, d/db[n] = -g */
        } else {
            float dp = 0.f, dn = 0.f;
            for (int d = 0; d < D; ++d) {
                float a = u[d] - p[d], c = u[d] - n[d];
                dp += a * a; dn += c * c; sq += u[d] * u[d] + p[d] * p[d] + n[d] * n[d];
            }
            float diff = (-dp + b[pid[k]]) - (-dn + b[nid[k]]);
            float h = margin - diff;
            loss += (double)fmaxf(h, 0.f);
            float a = (h >= 0.f) ? 1.f : 0.f;
            for (int d = 0; d < D; ++d) {
                gu[k * D + d] = -2.f * a * (p[d] - n[d]) + l2w * u[d];
                gp[k * D + d] = -2.f * a * (u[d] - p[d]) + l2w * p[d];
                gn[k * D + d] = 2.f * a * (u[d] - n[d]) + l2w * n[d];
            }
            gb[k] = -a;                              /* d/db[p] = -a, d/db[n] = +a */
        }
        l2 += 0.5 * (double)sq;
    }
    out[0] = loss; out[1] = l2;
}

/* One train step.  Tables U[NU,D], V[NI,D], b[NI] are updated in place.
 * accU/accV/accb are the Adagrad accumulators (ignored for SGD).
 * Returns loss and l2_loss through out[0], out[1]. */
int orc_pairwise_step(int model, int opt, float* U, float* V, float* b,
                      float* accU, float* accV, float* accb,
                      int64_t NU, int64_t NI, int D,
                      const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t B,
                      float lr, float eps, float margin, float* scratch, double* out) {
    float* gu = scratch;
    float* gp = gu + (size_t)B * D;
    float* gn = gp + (size_t)B * D;
    float* gb = gn + (size_t)B * D;
    forward_grads(model, U, V, b, D, uid, pid, nid, B, margin, gu, gp, gn, gb, out);

    /* ---- optimizer sparse apply ---- */
    if (opt == ORC_SGD) {
        /* scatter_add(idx, -lr*grad): every occurrence accumulated */
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < B; ++k) {
            float* u = U + (size_t)uid[k] * D;
            float* p = V + (size_t)pid[k] * D;
            float* n = V + (size_t)nid[k] * D;
            for (int d = 0; d < D; ++d) {
                float du = -lr * gu[k * D + d], dp = -lr * gp[k * D + d], dn = -lr * gn[k * D + d];
#pragma omp atomic
                u[d] += du;
#pragma omp atomic
                p[d] += dp;
#pragma omp atomic
                n[d] += dn;
            }
            float db = -lr * gb[k];
#pragma omp atomic
            b[pid[k]] += db;
#pragma omp atomic
            b[nid[k]] -= db;
        }
        return 0;
    }
    /* Adagrad: dedup-sum first.  Chain references per row (single thread for
     * determinism: the oracle is a checker, not a speed record). */
    {
        int64_t* headU = (int64_t*)malloc(sizeof(int64_t) * (size_t)NU);
        int64_t* headV = (int64_t*)malloc(sizeof(int64_t) * (size_t)NI);
        int64_t* nextU = (int64_t*)malloc(sizeof(int64_t) * (size_t)B);
        int64_t* nextV = (int64_t*)malloc(sizeof(int64_t) * (size_t)B * 2);
        float* G = (float*)malloc(sizeof(float) * (size_t)D);
        if (!headU || !headV || !nextU || !nextV || !G) return -1;
        for (int64_t k = 0; k < B; ++k) { headU[uid[k]] = -1; headV[pid[k]] = -1; headV[nid[k]] = -1; }
        for (int64_t k = B - 1; k >= 0; --k) { nextU[k] = headU[uid[k]]; headU[uid[k]] = k; }
        for (int64_t k = 2 * B - 1; k >= 0; --k) {      /* refs 0..B-1 = p lookups, B..2B-1 = n lookups */
            int32_t r = k < B ? pid[k] : nid[k - B];
            nextV[k] = headV[r]; headV[r] = k;
        }
        for (int64_t k = 0; k < B; ++k) {
            int32_t r = uid[k];
            if (headU[r] != k) continue;                 /* first occurrence owns the row */
            memset(G, 0, sizeof(float) * D);
            for (int64_t j = k; j >= 0; j = nextU[j]) for (int d = 0; d < D; ++d) G[d] += gu[j * D + d];
            float* w = U + (size_t)r * D; float* a = accU + (size_t)r * D;
            for (int d = 0; d < D; ++d) { a[d] += G[d] * G[d]; w[d] -= lr * G[d] / (sqrtf(a[d]) + eps); }
        }
        for (int64_t k = 0; k < 2 * B; ++k) {
            int32_t r = k < B ? pid[k] : nid[k - B];
            if (headV[r] != k) continue;
            memset(G, 0, sizeof(float) * D);
            float Gb = 0.f;
            for (int64_t j = k; j >= 0; j = nextV[j]) {
                const float* src = j < B ? gp + j * D : gn + (j - B) * D;
                for (int d = 0; d < D; ++d) G[d] += src[d];
                Gb += j < B ? gb[j] : -gb[j - B];
            }
            float* w = V + (size_t)r * D; float* a = accV + (size_t)r * D;
            for (int d = 0; d < D; ++d) { a[d] += G[d] * G[d]; w[d] -= lr * G[d] / (sqrtf(a[d]) + eps); }
            accb[r] += Gb * Gb; b[r] -= lr * Gb / (sqrtf(accb[r]) + eps);
        }
        free(headU); free(headV); free(nextU); free(nextV); free(G);
    }
    return 0;
}

/* TF-2.0 Keras Adam, sparse apply ("dense decay", numpy_oracle.AdamTFSparse): for every variable
 *   m <- b1*m (whole table);  m[idx] += (1-b1)*G   (G: duplicates summed first)
 *   v <- b2*v (whole table);  v[idx] += (1-b2)*G*G
 *   var <- var - lr_t * m / (sqrt(v) + eps)        (whole table)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller (step counter t). */
static void adam_decay(float* m, float* v, size_t n, float b1, float b2) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) { m[i] *= b1; v[i] *= b2; }
}
static void adam_update(float* w, const float* m, const float* v, size_t n, float lr_t, float eps) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) w[i] -= lr_t * m[i] / (sqrtf(v[i]) + eps);
}

int orc_pairwise_step_adam(int model, float* U, float* V, float* b,
                           float* mU, float* vU, float* mV, float* vV, float* mb, float* vb,
                           int64_t NU, int64_t NI, int D,
                           const int32_t* uid, const int32_t* pid, const int32_t* nid, int64_t B,
                           float lr_t, float b1, float b2, float eps, float margin, float* scratch, double* out) {
    float* gu = scratch;
    float* gp = gu + (size_t)B * D;
    float* gn = gp + (size_t)B * D;
    float* gb = gn + (size_t)B * D;
    forward_grads(model, U, V, b, D, uid, pid, nid, B, margin, gu, gp, gn, gb, out);
    adam_decay(mU, vU, (size_t)NU * D, b1, b2);
    adam_decay(mV, vV, (size_t)NI * D, b1, b2);
    adam_decay(mb, vb, (size_t)NI, b1, b2);
    {   /* summed gradient of every distinct row (reference chains as in the Adagrad branch) */
        int64_t* headU = (int64_t*)malloc(sizeof(int64_t) * (size_t)NU);
        int64_t* headV = (int64_t*)malloc(sizeof(int64_t) * (size_t)NI);
        int64_t* nextU = (int64_t*)malloc(sizeof(int64_t) * (size_t)B);
        int64_t* nextV = (int64_t*)malloc(sizeof(int64_t) * (size_t)B * 2);
        float* G = (float*)malloc(sizeof(float) * (size_t)D);
        if (!headU || !headV || !nextU || !nextV || !G) return -1;
        for (int64_t k = 0; k < B; ++k) { headU[uid[k]] = -1; headV[pid[k]] = -1; headV[nid[k]] = -1; }
        for (int64_t k = B - 1; k >= 0; --k) { nextU[k] = headU[uid[k]]; headU[uid[k]] = k; }
        for (int64_t k = 2 * B - 1; k >= 0; --k) {
            int32_t r = k < B ? pid[k] : nid[k - B];
            nextV[k] = headV[r]; headV[r] = k;
        }
        for (int64_t k = 0; k < B; ++k) {
            int32_t r = uid[k];
            if (headU[r] != k) continue;
            memset(G, 0, sizeof(float) * D);
            for (int64_t j = k; j >= 0; j = nextU[j]) for (int d = 0; d < D; ++d) G[d] += gu[j * D + d];
            float* m = mU + (size_t)r * D; float* v = vU + (size_t)r * D;
            for (int d = 0; d < D; ++d) { m[d] += (1.f - b1) * G[d]; v[d] += (1.f - b2) * G[d] * G[d]; }
        }
        for (int64_t k = 0; k < 2 * B; ++k) {
            int32_t r = k < B ? pid[k] : nid[k - B];
            if (headV[r] != k) continue;
            memset(G, 0, sizeof(float) * D);
            float Gb = 0.f;
            for (int64_t j = k; j >= 0; j = nextV[j]) {
                const float* src = j < B ? gp + j * D : gn + (j - B) * D;
                for (int d = 0; d < D; ++d) G[d] += src[d];
                Gb += j < B ? gb[j] : -gb[j - B];
            }
            float* m = mV + (size_t)r * D; float* v = vV + (size_t)r * D;
            for (int d = 0; d < D; ++d) { m[d] += (1.f - b1) * G[d]; v[d] += (1.f - b2) * G[d] * G[d]; }
            mb[r] += (1.f - b1) * Gb; vb[r] += (1.f - b2) * Gb * Gb;
        }
        free(headU); free(headV); free(nextU); free(nextV); free(G);
    }
    adam_update(U, mU, vU, (size_t)NU * D, lr_t, eps);
    adam_update(V, mV, vV, (size_t)NI * D, lr_t, eps);
    adam_update(b, mb, vb, (size_t)NI, lr_t, eps);
    return 0;
}

/* LatentFactor.censor (latent_factor.py:17-23) on first-occurrence-unique ids. */
int orc_censor(float* W, int64_t N, int D, const int32_t* ids, int64_t n, float min_norm) {
    unsigned char* seen = (unsigned char*)calloc((size_t)N, 1);
    if (!seen) return -1;
    for (int64_t k = 0; k < n; ++k) {
        int32_t r = ids[k];
        if (seen[r]) continue;
        seen[r] = 1;
        float* w = W + (size_t)r * D;
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += w[d] * w[d];
        float den = fmaxf(sqrtf(s), min_norm);
        for (int d = 0; d < D; ++d) w[d] = w[d] / den;
    }
    free(seen);
    return 0;
}
