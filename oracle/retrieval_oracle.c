/*
 * retrieval_oracle.c — plain-C restatement of the cosine retrieval on the reference's query path.
 *
 * TEST INFRASTRUCTURE ONLY (linked by tests/ and timed by bench.py's cpu_baseline leg through
 * oracle/retrieval_oracle_c.py); never part of the product. Same arithmetic as oracle/retrieval_oracle.py,
 * which documents what is restated and why parity is UNPINNED at the chromadb boundary:
 *   collection.add    backend/app/main.py:735-740   -> mo_normalize_rows
 *   collection.query  backend/app/main.py:761-765   -> mo_query   (cosine space, utils.py:127-130)
 *   blend             backend/app/main.py:852-860   -> mo_blend
 *
 * Canonical order: 64 partial sums p[l] += v[l + 64 i] (i ascending) in double, then
 * p[l] += p[l + s] for s = 32,16,8,4,2,1. Products of floats are exact in double, so an fma contraction
 * cannot change a result; -ffp-contract=off is set in the Makefile anyway.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* round-to-nearest-even float -> half (finite inputs in half range; overflow -> inf) */
static uint16_t float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    int32_t exp = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t man = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000u;
        int shift = 14 - exp;
        uint32_t half_man = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half_man & 1))) ++half_man;
        return (uint16_t)(sign | half_man);
    }
    uint32_t half_man = man >> 13, rem = man & 0x1fffu;
    uint16_t h = (uint16_t)(sign | ((uint32_t)exp << 10) | half_man);
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return h;
}

static double canon_reduce(double p[64]) {
    for (int s = 32; s >= 1; s >>= 1)
        for (int l = 0; l < s; ++l) p[l] = p[l] + p[l + s];
    return p[0];
}

static double canon_sumsq(const float* x, int D) {
    double p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0;
    for (int d = 0; d < D; ++d) {
        const double v = (double)x[d];
        p[d & 63] = p[d & 63] + v * v;
    }
    return canon_reduce(p);
}

static double canon_dot(const float* q, const void* row, int dtype, int D) {
    double p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0;
    if (dtype == 1) {
        const uint16_t* c = (const uint16_t*)row;
        for (int d = 0; d < D; ++d) p[d & 63] = p[d & 63] + (double)q[d] * (double)half_to_float(c[d]);
    } else {
        const float* c = (const float*)row;
        for (int d = 0; d < D; ++d) p[d & 63] = p[d & 63] + (double)q[d] * (double)c[d];
    }
    return canon_reduce(p);
}

/* dtype: 0 = f32 storage, 1 = f16 storage. dst: n*D elements of the storage type. */
void mo_normalize_rows(const float* src, int64_t n, int D, int dtype, void* dst) {
    for (int64_t r = 0; r < n; ++r) {
        const float* x = src + r * D;
        const double nrm = sqrt(canon_sumsq(x, D));
        for (int d = 0; d < D; ++d) {
            const float y = (float)((double)x[d] / nrm);
            if (dtype == 1)
                ((uint16_t*)dst)[r * D + d] = float_to_half(y);
            else
                ((float*)dst)[r * D + d] = y;
        }
    }
}

typedef struct { float dist; int64_t label; } hit_t;

static int hit_before(const hit_t* a, const hit_t* b) {
    return a->dist < b->dist || (a->dist == b->dist && a->label < b->label);
}

/* heap with the WORST kept hit at the root */
static void sift_down(hit_t* h, int n, int i) {
    for (;;) {
        int l = 2 * i + 1, r = l + 1, w = i;
        if (l < n && hit_before(&h[w], &h[l])) w = l;
        if (r < n && hit_before(&h[w], &h[r])) w = r;
        if (w == i) return;
        hit_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static int cmp_hit(const void* a, const void* b) {
    const hit_t *x = (const hit_t*)a, *y = (const hit_t*)b;
    if (hit_before(x, y)) return -1;
    if (hit_before(y, x)) return 1;
    return 0;
}

/* queries: raw float32 [Q,D]; stored: mo_normalize_rows output [N,D]; labels int64 [N].
 * out_labels [Q,k], out_dist [Q,k], out_count [Q]; unused slots -1 / +inf. */
static void mo_query_impl(const float* queries, int Q, const void* stored, int dtype, int64_t N, int D, const int64_t* labels,
                          int k, int64_t* out_labels, float* out_dist, int32_t* out_count, const float* inv);

void mo_query(const float* queries, int Q, const void* stored, int dtype, int64_t N, int D, const int64_t* labels,
              int k, int64_t* out_labels, float* out_dist, int32_t* out_count) {
    mo_query_impl(queries, Q, stored, dtype, N, D, labels, k, out_labels, out_dist, out_count, NULL);
}

/* MMISS_F8 rows: stored = the float32 VALUES of the codes (dtype 0), inv[r] = float(1 / canonical norm of row r's values);
 * distance = (float)(1.0 - canon_dot(q^, values[r]) * (double)inv[r])  (retrieval_oracle.py F8Rows) */
void mo_query_f8(const float* queries, int Q, const float* values, const float* inv, int64_t N, int D, const int64_t* labels,
                 int k, int64_t* out_labels, float* out_dist, int32_t* out_count) {
    mo_query_impl(queries, Q, values, 0, N, D, labels, k, out_labels, out_dist, out_count, inv);
}

static void mo_query_impl(const float* queries, int Q, const void* stored, int dtype, int64_t N, int D, const int64_t* labels,
                          int k, int64_t* out_labels, float* out_dist, int32_t* out_count, const float* inv) {
    float* qn = (float*)malloc((size_t)D * sizeof(float));
    hit_t* heap = (hit_t*)malloc((size_t)k * sizeof(hit_t));
    const size_t elt = dtype == 1 ? 2 : 4;
    for (int qi = 0; qi < Q; ++qi) {
        mo_normalize_rows(queries + (size_t)qi * D, 1, D, 0, qn);
        int n = 0;
        for (int64_t r = 0; r < N; ++r) {
            hit_t h;
            double dot = canon_dot(qn, (const char*)stored + (size_t)r * D * elt, dtype, D);
            if (inv) dot = dot * (double)inv[r];
            h.dist = (float)(1.0 - dot);
            h.label = labels[r];
            if (h.dist != h.dist) continue; /* NaN rows are never returned */
            if (n < k) {
                heap[n++] = h;
                if (n == k)
                    for (int i = k / 2 - 1; i >= 0; --i) sift_down(heap, k, i);
            } else if (hit_before(&h, &heap[0])) {
                heap[0] = h;
                sift_down(heap, k, 0);
            }
        }
        qsort(heap, (size_t)n, sizeof(hit_t), cmp_hit);
        for (int i = 0; i < k; ++i) {
            out_labels[(size_t)qi * k + i] = i < n ? heap[i].label : -1;
            out_dist[(size_t)qi * k + i] = i < n ? heap[i].dist : INFINITY;
        }
        out_count[qi] = n;
    }
    free(qn);
    free(heap);
}

void mo_blend(const float* img, const float* txt, double w, int Q, int D, float* out) {
    float* a = (float*)malloc((size_t)D * 4);
    float* b = (float*)malloc((size_t)D * 4);
    float* c = (float*)malloc((size_t)D * 4);
    const float wi = (float)w, wt = (float)(1.0 - w);
    for (int q = 0; q < Q; ++q) {
        mo_normalize_rows(img + (size_t)q * D, 1, D, 0, a);
        mo_normalize_rows(txt + (size_t)q * D, 1, D, 0, b);
        for (int d = 0; d < D; ++d) {
            volatile float pi = wi * a[d], pt = wt * b[d]; /* two rounded products, one rounded add */
            c[d] = pi + pt;
        }
        mo_normalize_rows(c, 1, D, 0, out + (size_t)q * D);
    }
    free(a); free(b); free(c);
}
