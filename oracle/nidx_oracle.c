/*
 * nidx_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See nidx_oracle.h.
 *
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -mavx2 -mfma; every FMA is an explicit fmaf)
 * Citations are file:line relative to /root/reference/nidx/.
 */
#include "nidx_oracle.h"

#include <math.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * a1. dense_f32::{dot,cosine}_similarity (nidx_vector/src/vector_types/dense_f32.rs:29-39)
 *     -> simsimd 6.5.16 `f32::dot` / `f32::cosine` [third party, restated].
 * ------------------------------------------------------------------------------------------ */

static void sums_serial(const float *x, const float *y, size_t n, double *ab, double *xx, double *yy) {
    /* SimSIMD portable macro (SIMSIMD_MAKE_DOT / SIMSIMD_MAKE_COS, f32 accumulators) */
    float sab = 0.0f, sxx = 0.0f, syy = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float xi = x[i], yi = y[i];
        sab += xi * yi;
        sxx += xi * xi;
        syy += yi * yi;
    }
    *ab = sab; *xx = sxx; *yy = syy;
}

static void sums_serial_fma(const float *x, const float *y, size_t n, double *ab, double *xx, double *yy) {
    /* one k-ordered fmaf chain == gfx950 v_mfma_f32_32x32x2_f32 / 16x16x4_f32 numerics */
    float sab = 0.0f, sxx = 0.0f, syy = 0.0f;
    for (size_t i = 0; i < n; i++) {
        float xi = x[i], yi = y[i];
        sab = fmaf(xi, yi, sab);
        sxx = fmaf(xi, xi, sxx);
        syy = fmaf(yi, yi, syy);
    }
    *ab = sab; *xx = sxx; *yy = syy;
}

static double reduce8_f64(const float a[8]) {
    /* _simsimd_reduce_f32x8_haswell: widen halves to f64, add, horizontal add */
    double s0 = (double)a[0] + (double)a[4];
    double s1 = (double)a[1] + (double)a[5];
    double s2 = (double)a[2] + (double)a[6];
    double s3 = (double)a[3] + (double)a[7];
    double t0 = s0 + s2;
    double t1 = s1 + s3;
    return t0 + t1;
}

static void sums_haswell(const float *x, const float *y, size_t n, double *ab, double *xx, double *yy) {
    float vab[8] = {0}, vxx[8] = {0}, vyy[8] = {0};
    size_t i = 0;
#if defined(__AVX2__) && defined(__FMA__)
    /* the same 8 independent fmaf lanes, issued as the AVX2 instructions SimSIMD's haswell kernel
     * uses (bit-identical to the scalar lanes below; this is what makes the CPU baseline fair) */
    __m256 aab = _mm256_setzero_ps(), axx = _mm256_setzero_ps(), ayy = _mm256_setzero_ps();
    for (; i + 8 <= n; i += 8) {
        __m256 vx = _mm256_loadu_ps(x + i), vy = _mm256_loadu_ps(y + i);
        aab = _mm256_fmadd_ps(vx, vy, aab);
        axx = _mm256_fmadd_ps(vx, vx, axx);
        ayy = _mm256_fmadd_ps(vy, vy, ayy);
    }
    _mm256_storeu_ps(vab, aab);
    _mm256_storeu_ps(vxx, axx);
    _mm256_storeu_ps(vyy, ayy);
#else
    for (; i + 8 <= n; i += 8) {
        for (int l = 0; l < 8; l++) {
            float xi = x[i + l], yi = y[i + l];
            vab[l] = fmaf(xi, yi, vab[l]);
            vxx[l] = fmaf(xi, xi, vxx[l]);
            vyy[l] = fmaf(yi, yi, vyy[l]);
        }
    }
#endif
    double sab = reduce8_f64(vab), sxx = reduce8_f64(vxx), syy = reduce8_f64(vyy);
    for (; i < n; i++) {
        float xi = x[i], yi = y[i];
        sab += (double)(xi * yi);
        sxx += (double)(xi * xi);
        syy += (double)(yi * yi);
    }
    *ab = sab; *xx = sxx; *yy = syy;
}

static float butterfly64(float a[64]) {
    /* __shfl_xor butterfly, offsets 32,16,8,4,2,1: every lane ends with the same value */
    float b[64];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) b[l] = a[l] + a[l ^ off];
        memcpy(a, b, sizeof(b));
    }
    return a[0];
}

static void sums_wave64(const float *x, const float *y, size_t n, double *ab, double *xx, double *yy) {
    /* canonical order of the HIP kernels: lane l owns elements j*256 + 4l .. +3 (one 16 B load per
     * lane per 1 KiB row chunk), fmaf chain in (j, component) order, then the xor butterfly. */
    float vab[64], vxx[64], vyy[64];
    for (int l = 0; l < 64; l++) {
        float sab = 0.0f, sxx = 0.0f, syy = 0.0f;
        for (size_t base = (size_t)l * 4; base < n; base += 256) {
            for (size_t c = 0; c < 4 && base + c < n; c++) {
                float xi = x[base + c], yi = y[base + c];
                sab = fmaf(xi, yi, sab);
                sxx = fmaf(xi, xi, sxx);
                syy = fmaf(yi, yi, syy);
            }
        }
        vab[l] = sab; vxx[l] = sxx; vyy[l] = syy;
    }
    *ab = butterfly64(vab);
    *xx = butterfly64(vxx);
    *yy = butterfly64(vyy);
}

static void sums_d(const float *x, const float *y, size_t n, int order, double *ab, double *xx, double *yy) {
    switch (order) {
        case ORC_ORDER_SERIAL: sums_serial(x, y, n, ab, xx, yy); break;
        case ORC_ORDER_SERIAL_FMA: sums_serial_fma(x, y, n, ab, xx, yy); break;
        case ORC_ORDER_HASWELL: sums_haswell(x, y, n, ab, xx, yy); break;
        default: sums_wave64(x, y, n, ab, xx, yy); break;
    }
}

void orc_sums(const float *x, const float *y, size_t n, int order, float *ab, float *xx, float *yy) {
    double a, b, c;
    sums_d(x, y, n, order, &a, &b, &c);
    *ab = (float)a; *xx = (float)b; *yy = (float)c;
}

static float cosine_from_sums_d(double ab, double xx, double yy) {
    /* SimSIMD cosine distance: a2==b2==0 -> 0; ab==0 -> 1; else clamp(1 - ab/(|a||b|), >= 0).
     * SimSIMD normalises with rsqrt + one Newton step (hardware approximation, not bit
     * reproducible); restated with exact f64 sqrt/div.  nidx then does `1.0 - (dist as f32)`
     * in f32 (dense_f32.rs:32). */
    double dist;
    if (xx == 0.0 && yy == 0.0) {
        dist = 0.0;
    } else if (ab == 0.0) {
        dist = 1.0;
    } else {
        double d = 1.0 - ab / (sqrt(xx) * sqrt(yy));
        dist = d > 0.0 ? d : 0.0;
    }
    return 1.0f - (float)dist;
}

float orc_cosine_from_sums(float ab, float xx, float yy) {
    return cosine_from_sums_d((double)ab, (double)xx, (double)yy);
}

float orc_dot(const float *x, const float *y, size_t n, int order) {
    double ab, xx, yy;
    sums_d(x, y, n, order, &ab, &xx, &yy);
    return (float)ab; /* dense_f32.rs:38 `as f32` */
}

float orc_cosine(const float *x, const float *y, size_t n, int order) {
    double ab, xx, yy;
    sums_d(x, y, n, order, &ab, &xx, &yy);
    return cosine_from_sums_d(ab, xx, yy);
}

float orc_similarity(const float *x, const float *y, size_t n, int similarity, int order) {
    /* VectorConfig::similarity_function (config.rs:163-168) */
    return similarity == ORC_SIM_COSINE ? orc_cosine(x, y, n, order) : orc_dot(x, y, n, order);
}

/* a11. utils::normalize_vector (nidx_vector/src/utils.rs:20-23): f32 sequential fold of x.powi(2) */
void orc_normalize(const float *in, float *out, size_t n) {
    float acc = 0.0f;
    for (size_t i = 0; i < n; i++) acc = acc + in[i] * in[i];
    float magnitude = sqrtf(acc);
    for (size_t i = 0; i < n; i++) out[i] = in[i] / magnitude;
}

/* Rust f32::total_cmp */
static inline int32_t total_key(float f) {
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}
int orc_total_cmp(float a, float b) {
    int32_t ka = total_key(a), kb = total_key(b);
    return (ka > kb) - (ka < kb);
}


/* ------------------------------------------------------------------------------------------
 * RaBitQ (nidx_vector/src/vector_types/rabitq.rs).  Integer popcounts and plain f32 arithmetic in
 * the reference's operation order (this file is compiled with -ffp-contract=off), so every value
 * below except dot_quant_original (a SimSIMD dot: summation order = `order`) is exactly the
 * reference's.
 * ------------------------------------------------------------------------------------------ */
#define RABITQ_EPSILON 1.9f /* rabitq.rs:30 */

size_t orc_rabitq_encoded_len(size_t dim) { return dim / 8 + 8; }

void orc_rabitq_encode(const float *v, size_t dim, int order, uint8_t *out) {
    float root_dim = sqrtf((float)dim);
    size_t n_words = dim / 64;
    uint64_t *quantized = (uint64_t *)calloc(n_words ? n_words : 1, 8);
    float *v_repr = (float *)malloc((dim ? dim : 1) * sizeof(float));
    uint32_t sum_bits = 0;
    for (size_t i = 0; i < dim; i++) {
        if (v[i] > 0.0f) {
            quantized[i / 64] += (uint64_t)1 << (i % 64);
            sum_bits++;
            v_repr[i] = 1.0f / root_dim;
        } else {
            v_repr[i] = -1.0f / root_dim;
        }
    }
    float dot_quant_original = orc_dot(v, v_repr, dim, order);
    memcpy(out, &dot_quant_original, 4);
    memcpy(out + 4, &sum_bits, 4);
    memcpy(out + 8, quantized, n_words * 8);
    free(quantized);
    free(v_repr);
}

/* Rust `f32 as u64`: saturating, NaN -> 0 */
static inline uint64_t f32_as_u64(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)x;
}

void orc_rabitq_query_init(orc_rabitq_query *q, const float *v, size_t dim) {
    float low = v[0], hi = v[0];
    for (size_t i = 0; i < dim; i++) {
        if (v[i] < low) low = v[i];
        if (v[i] > hi) hi = v[i];
    }
    hi += 0.00001f;
    float delta = (hi - low) / 16.0f;
    size_t n_words = dim / 64;
    q->planes = (uint64_t *)calloc(4 * (n_words ? n_words : 1), 8);
    uint64_t sum_quantized = 0;
    for (size_t i = 0; i < dim; i++) {
        uint64_t wq = f32_as_u64((v[i] - low) / delta);
        sum_quantized += wq;
        q->planes[0 * n_words + i / 64] += (wq % 2) << (i % 64);
        q->planes[1 * n_words + i / 64] += ((wq / 2) % 2) << (i % 64);
        q->planes[2 * n_words + i / 64] += ((wq / 4) % 2) << (i % 64);
        q->planes[3 * n_words + i / 64] += ((wq / 8) % 2) << (i % 64);
    }
    q->low = low;
    q->delta = delta;
    q->root_dim = sqrtf((float)dim);
    q->sum_quantized = (uint32_t)sum_quantized;
    q->n_words = (uint32_t)n_words;
}

void orc_rabitq_query_free(orc_rabitq_query *q) { free(q->planes); q->planes = NULL; }

void orc_rabitq_similarity(const orc_rabitq_query *q, const uint8_t *encoded, float *estimate, float *error) {
    float dot_quant_original;
    uint32_t sum_bits;
    memcpy(&dot_quant_original, encoded, 4);
    memcpy(&sum_bits, encoded + 4, 4);
    uint32_t d[4] = {0, 0, 0, 0};
    for (int p = 0; p < 4; p++)
        for (uint32_t w = 0; w < q->n_words; w++) {
            uint64_t s;
            memcpy(&s, encoded + 8 + (size_t)w * 8, 8);
            d[p] += (uint32_t)__builtin_popcountll(q->planes[(size_t)p * q->n_words + w] & s);
        }
    float dot = (float)(d[0] + d[1] * 2 + d[2] * 4 + d[3] * 8);
    float dot_quant_query_vector = 2.0f * q->delta / q->root_dim * dot
                                   + 2.0f * q->low * (float)sum_bits / q->root_dim
                                   - q->delta * (float)q->sum_quantized / q->root_dim
                                   - q->low * q->root_dim;
    *estimate = dot_quant_query_vector / dot_quant_original;
    float d2 = dot_quant_original * dot_quant_original;
    *error = sqrtf((1.0f - d2) / d2) * RABITQ_EPSILON / q->root_dim;
}

/* ------------------------------------------------------------------------------------------
 * small containers
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint32_t addr; float score; } cnx_t;

/* Ranking order used wherever the reference leaves ties unspecified (sort_unstable, BinaryHeap
 * on a score-only Ord): higher score first, then LOWER address first.  `better(a,b)` is a strict
 * total order. */
static inline int better(cnx_t a, cnx_t b) {
    int c = orc_total_cmp(a.score, b.score);
    if (c != 0) return c > 0;
    return a.addr < b.addr;
}

typedef struct { cnx_t *d; size_t len, cap; int max_heap; } heap_t;

static void heap_init(heap_t *h, int max_heap) { h->d = NULL; h->len = h->cap = 0; h->max_heap = max_heap; }
static void heap_free(heap_t *h) { free(h->d); }
/* top-of-heap predicate: in a max-heap the best element is on top, in a min-heap the worst */
static inline int heap_above(const heap_t *h, cnx_t a, cnx_t b) { return h->max_heap ? better(a, b) : better(b, a); }
static void heap_push(heap_t *h, cnx_t v) {
    if (h->len == h->cap) { h->cap = h->cap ? h->cap * 2 : 64; h->d = (cnx_t *)realloc(h->d, h->cap * sizeof(cnx_t)); }
    size_t i = h->len++;
    h->d[i] = v;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!heap_above(h, h->d[i], h->d[p])) break;
        cnx_t t = h->d[i]; h->d[i] = h->d[p]; h->d[p] = t;
        i = p;
    }
}
static cnx_t heap_pop(heap_t *h) {
    cnx_t top = h->d[0];
    h->d[0] = h->d[--h->len];
    size_t i = 0;
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < h->len && heap_above(h, h->d[l], h->d[m])) m = l;
        if (r < h->len && heap_above(h, h->d[r], h->d[m])) m = r;
        if (m == i) break;
        cnx_t t = h->d[i]; h->d[i] = h->d[m]; h->d[m] = t;
        i = m;
    }
    return top;
}

/* open-addressing u32 set (stands for FxHashSet<VectorAddr> / BitSet) */
typedef struct { uint32_t *slots; size_t cap, len; } u32set_t;
static void set_init(u32set_t *s, size_t cap_pow2) {
    s->cap = cap_pow2; s->len = 0;
    s->slots = (uint32_t *)malloc(cap_pow2 * sizeof(uint32_t));
    memset(s->slots, 0xff, cap_pow2 * sizeof(uint32_t));
}
static void set_free(u32set_t *s) { free(s->slots); }
static int set_insert(u32set_t *s, uint32_t v); /* returns 1 if newly inserted */
static void set_grow(u32set_t *s) {
    u32set_t n;
    set_init(&n, s->cap * 2);
    for (size_t i = 0; i < s->cap; i++) if (s->slots[i] != 0xffffffffu) set_insert(&n, s->slots[i]);
    free(s->slots);
    *s = n;
}
static int set_insert(u32set_t *s, uint32_t v) {
    if ((s->len + 1) * 2 > s->cap) set_grow(s);
    size_t i = ((size_t)v * 2654435761u) & (s->cap - 1);
    while (s->slots[i] != 0xffffffffu) {
        if (s->slots[i] == v) return 0;
        i = (i + 1) & (s->cap - 1);
    }
    s->slots[i] = v;
    s->len++;
    return 1;
}
static inline int bit_get(const uint64_t *bits, uint32_t i) { return (int)((bits[i >> 6] >> (i & 63)) & 1u); }

/* ------------------------------------------------------------------------------------------
 * HNSW graph in RAM (nidx_vector/src/hnsw/ram_hnsw.rs:33-143)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t n_slots;        /* capacity of slot_of */
    int32_t *slot_of;        /* node -> slot or -1 (node not in this layer) */
    uint32_t n_present;
    uint32_t cap_present;
    uint32_t *node_of;       /* slot -> node */
    uint32_t **adj;          /* slot -> edges */
    float **w;               /* slot -> edge weights */
    uint32_t *deg;
    uint32_t *acap;
} layer_t;

struct orc_hnsw {
    uint32_t n_layers;
    layer_t *layers;
    uint32_t ep_node, ep_layer;
    uint32_t n_nodes;        /* 1 + highest node id added */
};

orc_hnsw *orc_hnsw_new(void) {
    orc_hnsw *g = (orc_hnsw *)calloc(1, sizeof(orc_hnsw));
    return g;
}

static void layer_free(layer_t *l) {
    for (uint32_t s = 0; s < l->n_present; s++) { free(l->adj[s]); free(l->w[s]); }
    free(l->slot_of); free(l->node_of); free(l->adj); free(l->w); free(l->deg); free(l->acap);
}

void orc_hnsw_free(orc_hnsw *g) {
    if (!g) return;
    for (uint32_t l = 0; l < g->n_layers; l++) layer_free(&g->layers[l]);
    free(g->layers);
    free(g);
}

static void layer_add_node(layer_t *l, uint32_t node) {
    if (node >= l->n_slots) {
        uint32_t ncap = l->n_slots ? l->n_slots : 1024;
        while (ncap <= node) ncap *= 2;
        l->slot_of = (int32_t *)realloc(l->slot_of, ncap * sizeof(int32_t));
        for (uint32_t i = l->n_slots; i < ncap; i++) l->slot_of[i] = -1;
        l->n_slots = ncap;
    }
    if (l->slot_of[node] >= 0) return;
    if (l->n_present == l->cap_present) {
        uint32_t nc = l->cap_present ? l->cap_present * 2 : 256;
        l->node_of = (uint32_t *)realloc(l->node_of, nc * sizeof(uint32_t));
        l->adj = (uint32_t **)realloc(l->adj, nc * sizeof(uint32_t *));
        l->w = (float **)realloc(l->w, nc * sizeof(float *));
        l->deg = (uint32_t *)realloc(l->deg, nc * sizeof(uint32_t));
        l->acap = (uint32_t *)realloc(l->acap, nc * sizeof(uint32_t));
        l->cap_present = nc;
    }
    uint32_t s = l->n_present++;
    l->slot_of[node] = (int32_t)s;
    l->node_of[s] = node;
    l->adj[s] = NULL; l->w[s] = NULL; l->deg[s] = 0; l->acap[s] = 0;
}

static inline int layer_contains(const layer_t *l, uint32_t node) {
    return node < l->n_slots && l->slot_of[node] >= 0;
}

static void layer_set_edges(layer_t *l, uint32_t node, const cnx_t *e, uint32_t n) {
    uint32_t s = (uint32_t)l->slot_of[node];
    if (n > l->acap[s]) {
        l->acap[s] = n + 4;
        l->adj[s] = (uint32_t *)realloc(l->adj[s], l->acap[s] * sizeof(uint32_t));
        l->w[s] = (float *)realloc(l->w[s], l->acap[s] * sizeof(float));
    }
    for (uint32_t i = 0; i < n; i++) { l->adj[s][i] = e[i].addr; l->w[s][i] = e[i].score; }
    l->deg[s] = n;
}

/* RAMHnsw::add_node (ram_hnsw.rs:86-95) */
void orc_hnsw_add_node(orc_hnsw *g, uint32_t node, uint32_t top_layer) {
    if (g->n_layers <= top_layer) {
        g->layers = (layer_t *)realloc(g->layers, (top_layer + 1) * sizeof(layer_t));
        for (uint32_t l = g->n_layers; l <= top_layer; l++) memset(&g->layers[l], 0, sizeof(layer_t));
        g->n_layers = top_layer + 1;
    }
    for (uint32_t l = 0; l <= top_layer; l++) layer_add_node(&g->layers[l], node);
    if (node + 1 > g->n_nodes) g->n_nodes = node + 1;
}

/* RAMHnsw::update_entry_point (ram_hnsw.rs:98-107).  The reference takes "the first key of the top
 * layer's FxHashMap" (hash-iteration order, unspecified); the oracle takes the LOWEST node id of
 * the top layer. */
void orc_hnsw_update_entry_point(orc_hnsw *g) {
    if (g->n_layers > g->ep_layer + 1) {
        const layer_t *top = &g->layers[g->n_layers - 1];
        uint32_t best = 0xffffffffu;
        for (uint32_t s = 0; s < top->n_present; s++) if (top->node_of[s] < best) best = top->node_of[s];
        g->ep_node = best;
        g->ep_layer = g->n_layers - 1;
    }
}

void orc_hnsw_set_edges(orc_hnsw *g, uint32_t layer, uint32_t node, const uint32_t *edges, const float *w, uint32_t n) {
    cnx_t *tmp = (cnx_t *)malloc((n ? n : 1) * sizeof(cnx_t));
    for (uint32_t i = 0; i < n; i++) { tmp[i].addr = edges[i]; tmp[i].score = w ? w[i] : 0.0f; }
    layer_set_edges(&g->layers[layer], node, tmp, n);
    free(tmp);
}

uint32_t orc_hnsw_num_layers(const orc_hnsw *g) { return g->n_layers; }
uint32_t orc_hnsw_num_nodes(const orc_hnsw *g) { return g->n_nodes; }
void orc_hnsw_entry_point(const orc_hnsw *g, uint32_t *node, uint32_t *layer) { *node = g->ep_node; *layer = g->ep_layer; }
void orc_hnsw_set_entry_point(orc_hnsw *g, uint32_t node, uint32_t layer) { g->ep_node = node; g->ep_layer = layer; }
int orc_hnsw_contains(const orc_hnsw *g, uint32_t layer, uint32_t node) {
    return layer < g->n_layers && layer_contains(&g->layers[layer], node);
}
uint32_t orc_hnsw_edges(const orc_hnsw *g, uint32_t layer, uint32_t node, uint32_t *out, float *w_out, uint32_t cap) {
    if (!orc_hnsw_contains(g, layer, node)) return 0;
    const layer_t *l = &g->layers[layer];
    uint32_t s = (uint32_t)l->slot_of[node];
    for (uint32_t i = 0; i < l->deg[s] && i < cap; i++) {
        if (out) out[i] = l->adj[s][i];
        if (w_out) w_out[i] = l->w[s][i];
    }
    return l->deg[s];
}

/* RAMHnsw::fix_broken_graph (ram_hnsw.rs:118-142) */
void orc_hnsw_fix_broken_graph(orc_hnsw *g) {
    for (uint32_t li = 1; li < g->n_layers; li++) {
        layer_t *l = &g->layers[li];
        for (uint32_t s = 0; s < l->n_present; s++) {
            uint32_t k = 0;
            for (uint32_t i = 0; i < l->deg[s]; i++) {
                if (layer_contains(l, l->adj[s][i])) { l->adj[s][k] = l->adj[s][i]; l->w[s][k] = l->w[s][i]; k++; }
            }
            l->deg[s] = k;
        }
    }
    while (g->n_layers > 0 && g->layers[g->n_layers - 1].n_present == 0) {
        layer_free(&g->layers[g->n_layers - 1]);
        g->n_layers--;
    }
    if (g->n_layers > 0 && g->ep_layer >= g->n_layers) {
        g->ep_layer = g->n_layers - 1;
        const layer_t *top = &g->layers[g->n_layers - 1];
        if (!layer_contains(top, g->ep_node)) {
            uint32_t best = 0xffffffffu;
            for (uint32_t s = 0; s < top->n_present; s++) if (top->node_of[s] < best) best = top->node_of[s];
            g->ep_node = best;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Retriever (segment.rs:288-357) and HnswSearcher (hnsw/search.rs:174-384)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const orc_segment *seg;
    const float *query;      /* SearchVector::Query, or stored vector for SearchVector::Stored */
    float min_score;
    orc_stats *stats;
    const orc_rabitq_query *rq; /* SearchVector::RabitQ when set: similarity() = the estimate (segment.rs:331-335) */
} retr_t;

static inline const float *seg_vec(const orc_segment *seg, uint32_t a) { return seg->vectors + (size_t)a * seg->dim; }
static inline uint32_t seg_paragraph(const orc_segment *seg, uint32_t a) { return seg->vec_paragraph ? seg->vec_paragraph[a] : a; }

static inline float retr_sim(const retr_t *r, uint32_t a) {
    if (r->stats) r->stats->distance_evals++;
    if (r->rq) {
        float est, err;
        orc_rabitq_similarity(r->rq, r->seg->quantized + (size_t)a * orc_rabitq_encoded_len(r->seg->dim), &est, &err);
        return est;
    }
    return orc_similarity(seg_vec(r->seg, a), r->query, r->seg->dim, r->seg->similarity, r->seg->order);
}

/* out-edge access: RAM graph */
static inline uint32_t graph_edges(const orc_hnsw *g, uint32_t layer, uint32_t node, const uint32_t **edges) {
    const layer_t *l = &g->layers[layer];
    if (!layer_contains(l, node)) { *edges = NULL; return 0; }
    uint32_t s = (uint32_t)l->slot_of[node];
    *edges = l->adj[s];
    return l->deg[s];
}

/* a3. HnswSearcher::layer_search (hnsw/search.rs:242-304).  Heaps pop in the `better` total order
 * where the reference's score-only Ord leaves ties to BinaryHeap internals. Output: (score desc). */
static size_t layer_search(const retr_t *r, const orc_hnsw *g, uint32_t layer, size_t k,
                           const uint32_t *eps, size_t n_ep, cnx_t **out) {
    u32set_t visited;
    heap_t cand, res;
    set_init(&visited, 1024);
    heap_init(&cand, 1);
    heap_init(&res, 0);
    for (size_t i = 0; i < n_ep; i++) {
        set_insert(&visited, eps[i]);
        cnx_t c = {eps[i], retr_sim(r, eps[i])};
        heap_push(&cand, c);
        heap_push(&res, c);
    }
    while (cand.len > 0) {
        cnx_t c = heap_pop(&cand);
        float ws = res.d[0].score;
        if (c.score < ws) break; /* candidate worse than the worst result */
        const uint32_t *edges;
        uint32_t deg = graph_edges(g, layer, c.addr, &edges);
        if (r->stats) { r->stats->expansions++; r->stats->edges_read += deg; }
        for (uint32_t i = 0; i < deg; i++) {
            uint32_t y = edges[i];
            if (!set_insert(&visited, y)) continue;
            float s = retr_sim(r, y);
            if (s > ws || res.len < k) {
                cnx_t n = {y, s};
                heap_push(&cand, n);
                heap_push(&res, n);
                if (res.len > k) heap_pop(&res);
                ws = res.d[0].score;
            }
        }
    }
    /* into_sorted_vec of Reverse<_>: descending score */
    size_t n = res.len;
    cnx_t *o = (cnx_t *)malloc((n ? n : 1) * sizeof(cnx_t));
    for (size_t i = n; i-- > 0;) o[i] = heap_pop(&res); /* min-heap pops worst first */
    *out = o;
    heap_free(&cand); heap_free(&res); set_free(&visited);
    return n;
}

/* NodeFilter::passes + RepCounter (hnsw/search.rs:129-172,386-412) */
typedef struct {
    const orc_segment *seg;
    const uint64_t *filter;
    int dedupe;              /* !with_duplicates */
    int multi;
    const cnx_t *results;    /* accepted so far (their vectors form the RepCounter) */
    size_t n_results;
    u32set_t paragraphs;
} nodefilter_t;

static int filter_passes(nodefilter_t *f, uint32_t v) {
    uint32_t p = seg_paragraph(f->seg, v);
    if (f->filter && !bit_get(f->filter, p)) return 0;
    if (f->dedupe) {
        const float *vec = seg_vec(f->seg, v);
        for (size_t i = 0; i < f->n_results; i++)
            if (memcmp(seg_vec(f->seg, f->results[i].addr), vec, (size_t)f->seg->dim * 4) == 0) return 0;
    }
    if (f->multi && !set_insert(&f->paragraphs, p)) return 0;
    return 1;
}

static int cmp_cnx_asc(const void *pa, const void *pb) {
    /* ascending in `better` order reversed: the LAST element is the best */
    cnx_t a = *(const cnx_t *)pa, b = *(const cnx_t *)pb;
    if (better(a, b)) return 1;
    if (better(b, a)) return -1;
    return 0;
}

/* a5. closest_up_nodes (hnsw/search.rs:188-240) */
static size_t closest_up_nodes(const retr_t *r, const orc_hnsw *g, cnx_t *entry, size_t n_entry, size_t k,
                               nodefilter_t *filter, cnx_t *results) {
    u32set_t visited;
    set_init(&visited, 1024);
    size_t cap = n_entry + 64, len = n_entry;
    cnx_t *cand = (cnx_t *)malloc(cap * sizeof(cnx_t));
    memcpy(cand, entry, n_entry * sizeof(cnx_t));
    for (size_t i = 0; i < n_entry; i++) set_insert(&visited, entry[i].addr);
    qsort(cand, len, sizeof(cnx_t), cmp_cnx_asc);
    size_t n_res = 0;
    while (len > 0) {
        cnx_t c = cand[--len];
        if (c.score < r->min_score) break;
        filter->results = results;
        filter->n_results = n_res;
        if (!isnan(c.score) && filter_passes(filter, c.addr)) results[n_res++] = c;
        if (n_res == k) break;
        const uint32_t *edges;
        uint32_t deg = graph_edges(g, 0, c.addr, &edges);
        if (r->stats) { r->stats->expansions++; r->stats->edges_read += deg; }
        for (uint32_t i = 0; i < deg; i++) {
            uint32_t y = edges[i];
            if (!set_insert(&visited, y)) continue;
            float s = retr_sim(r, y);
            if (s >= r->min_score) {
                if (len == cap) { cap *= 2; cand = (cnx_t *)realloc(cand, cap * sizeof(cnx_t)); }
                cand[len].addr = y; cand[len].score = s; len++;
            }
        }
        qsort(cand, len, sizeof(cnx_t), cmp_cnx_asc);
    }
    free(cand);
    set_free(&visited);
    return n_res;
}

#define EF_SEARCH 30          /* hnsw/params.rs:46 */
#define EF_CONSTRUCTION 100   /* hnsw/params.rs:43 */
#define HNSW_M 30             /* hnsw/params.rs:40 */
#define HNSW_M_MAX 30         /* hnsw/params.rs:37 */
#define HNSW_M_MAX_0 60       /* hnsw/params.rs:34 */

static void stable_sort_desc(cnx_t *a, size_t n) {
    /* filtered_result.sort_by(|a, b| b.1.total_cmp(&a.1)) — stable insertion sort */
    for (size_t i = 1; i < n; i++) {
        cnx_t v = a[i];
        size_t j = i;
        while (j > 0 && orc_total_cmp(a[j - 1].score, v.score) < 0) { a[j] = a[j - 1]; j--; }
        a[j] = v;
    }
}

/* rerank_top (rabitq.rs:221-244).  `best` is a min-heap on the real score; where BinaryHeap leaves ties to its
 * internals the `better` order decides (the evicted one of two equal scores is the higher address). */
static size_t rerank_top(const retr_t *exact, const cnx_t *cand, const float *upper_bound, size_t n_cand, size_t k,
                         cnx_t *out, uint64_t *n_evaluated) {
    heap_t best;
    heap_init(&best, 0);
    float best_k = 0.0f;
    for (size_t i = 0; i < n_cand; i++) {
        if (best.len < k || best_k < upper_bound[i]) {
            float real = retr_sim(exact, cand[i].addr);
            if (n_evaluated) (*n_evaluated)++;
            if (real >= exact->min_score && (best.len < k || best_k < real)) {
                cnx_t c = {cand[i].addr, real};
                heap_push(&best, c);
                if (best.len > k) heap_pop(&best);
                best_k = best.d[0].score;
            }
        }
    }
    size_t n = best.len;
    for (size_t i = n; i-- > 0;) out[i] = heap_pop(&best);
    heap_free(&best);
    return n;
}

size_t orc_rabitq_rerank_top(const orc_segment *seg, const float *query, float min_score, const uint32_t *cand,
                             const float *cand_upper_bound, size_t n_cand, size_t k, uint32_t *out_vec, float *out_score,
                             uint64_t *n_evaluated) {
    retr_t exact = {seg, query, min_score, NULL, NULL};
    cnx_t *c = (cnx_t *)malloc((n_cand ? n_cand : 1) * sizeof(cnx_t));
    cnx_t *o = (cnx_t *)malloc((k ? k : 1) * sizeof(cnx_t));
    for (size_t i = 0; i < n_cand; i++) { c[i].addr = cand[i]; c[i].score = 0.0f; }
    if (n_evaluated) *n_evaluated = 0;
    size_t n = k ? rerank_top(&exact, c, cand_upper_bound, n_cand, k, o, n_evaluated) : 0;
    for (size_t i = 0; i < n; i++) { out_vec[i] = o[i].addr; out_score[i] = o[i].score; }
    free(c); free(o);
    return n;
}

static float rabitq_upper_bound(const retr_t *r, uint32_t a) {
    float est, err;
    orc_rabitq_similarity(r->rq, r->seg->quantized + (size_t)a * orc_rabitq_encoded_len(r->seg->dim), &est, &err);
    return est + err; /* EstimatedScore::new_with_error (hnsw/search.rs:69-75) */
}

/* a4. HnswSearcher::search (hnsw/search.rs:306-383), both branches */
static size_t hnsw_search(const retr_t *r, const orc_hnsw *g, size_t k, const uint64_t *filter_bits,
                          int with_duplicates, int multi, cnx_t *results) {
    if (k == 0 || g->n_layers == 0) return 0;
    uint32_t layer = g->ep_layer;
    uint32_t *eps = (uint32_t *)malloc(sizeof(uint32_t));
    size_t n_ep = 1;
    eps[0] = g->ep_node;
    const size_t ef_upper = r->seg->ef_upper ? r->seg->ef_upper : 1;
    while (layer != 0) {
        cnx_t *lr;
        size_t n = layer_search(r, g, layer, ef_upper, eps, n_ep, &lr);
        eps = (uint32_t *)realloc(eps, (n ? n : 1) * sizeof(uint32_t));
        for (size_t i = 0; i < n; i++) eps[i] = lr[i].addr;
        n_ep = n;
        free(lr);
        layer--;
    }
    const size_t ef_search = r->seg->ef_search ? r->seg->ef_search : EF_SEARCH;
    size_t last_k = k > ef_search ? k : ef_search;
    if (r->rq) { /* RaBitQ: over-fetch, rerank later (:333-340) */
        last_k = k * ORC_RABITQ_RERANKING_FACTOR;
        if (last_k > ORC_RABITQ_RERANKING_LIMIT) last_k = ORC_RABITQ_RERANKING_LIMIT;
    }
    cnx_t *neigh;
    size_t n = layer_search(r, g, 0, last_k, eps, n_ep, &neigh);
    free(eps);
    retr_t exact = *r;
    exact.rq = NULL;
    if (r->rq) { /* rerank with the original vectors (:354-363) */
        float *ub = (float *)malloc((n ? n : 1) * sizeof(float));
        for (size_t i = 0; i < n; i++) ub[i] = rabitq_upper_bound(r, neigh[i].addr);
        cnx_t *rr = (cnx_t *)malloc((k ? k : 1) * sizeof(cnx_t));
        size_t nr = rerank_top(&exact, neigh, ub, n, k, rr, NULL);
        free(ub);
        free(neigh);
        neigh = rr;
        n = nr;
    }
    r = &exact; /* closest_up_nodes runs on the original query (:369-375) */
    nodefilter_t nf;
    nf.seg = r->seg; nf.filter = filter_bits; nf.dedupe = !with_duplicates; nf.multi = multi;
    nf.results = NULL; nf.n_results = 0;
    set_init(&nf.paragraphs, 64);
    size_t n_res = closest_up_nodes(r, g, neigh, n, k, &nf, results);
    set_free(&nf.paragraphs);
    free(neigh);
    stable_sort_desc(results, n_res);
    return n_res;
}

int orc_layer_search(const orc_segment *seg, const float *query, int query_is_stored, uint32_t stored_addr,
                     int layer, size_t k, const uint32_t *entry_points, size_t n_ep,
                     uint32_t *out_vec, float *out_score, orc_stats *stats) {
    retr_t r = {seg, query_is_stored ? seg_vec(seg, stored_addr) : query, -1.0f, stats, NULL};
    cnx_t *o;
    size_t n = layer_search(&r, seg->graph, (uint32_t)layer, k, entry_points, n_ep, &o);
    for (size_t i = 0; i < n; i++) { out_vec[i] = o[i].addr; out_score[i] = o[i].score; }
    free(o);
    return (int)n;
}

int orc_hnsw_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                    size_t k, float min_score, int with_duplicates, int multi_vector,
                    uint32_t *out_vec, float *out_score, orc_stats *stats) {
    retr_t r = {seg, query, min_score, stats, NULL};
    orc_rabitq_query rq;
    if (seg->quantized) { orc_rabitq_query_init(&rq, query, seg->dim); r.rq = &rq; }
    const uint64_t *bits = filter ? filter : seg->alive;
    cnx_t *res = (cnx_t *)malloc((k ? k : 1) * sizeof(cnx_t));
    size_t n = hnsw_search(&r, seg->graph, k, bits, with_duplicates, multi_vector, res);
    if (seg->quantized) orc_rabitq_query_free(&rq);
    for (size_t i = 0; i < n; i++) { out_vec[i] = res[i].addr; out_score[i] = res[i].score; }
    free(res);
    return (int)n;
}

/* a6. use_hnsw (segment.rs:626-660) */
int orc_use_hnsw(size_t total_nodes, size_t matching_nodes, size_t top_k, int has_rabitq) {
    size_t full_cost, search_mult, rerank_mult;
    const size_t RERANKING_FACTOR = 100; /* rabitq.rs:30-36 */
    if (has_rabitq) { full_cost = 16; search_mult = RERANKING_FACTOR * 3 / 4; rerank_mult = RERANKING_FACTOR / 2; }
    else { full_cost = 1; search_mult = 1; rerank_mult = 0; }
    float l = logf((float)total_nodes) - 2.0f;
    float hnsw_rq = (l * l) * logf((float)top_k) * (float)search_mult;
    size_t hnsw_full = (top_k * rerank_mult) + (top_k * HNSW_M * total_nodes / matching_nodes);
    size_t bf_rq = matching_nodes;
    size_t bf_full = top_k * rerank_mult;
    size_t hnsw_rq_u; /* Rust `as usize`: saturating, NaN -> 0 */
    if (!(hnsw_rq > 0.0f)) hnsw_rq_u = 0;
    else if (hnsw_rq >= 18446744073709551615.0f) hnsw_rq_u = SIZE_MAX;
    else hnsw_rq_u = (size_t)hnsw_rq;
    size_t hnsw_cost = hnsw_rq_u + hnsw_full * full_cost;
    size_t bf_cost = bf_rq + bf_full * full_cost;
    return hnsw_cost < bf_cost;
}

/* a7. OpenSegment::brute_force_search (segment.rs:569-623), both branches.
 * sort_unstable_by leaves ties unspecified -> `better` order (score desc, addr asc). */
static int cmp_cnx_better_first(const void *pa, const void *pb) { return -cmp_cnx_asc(pa, pb); }

int orc_brute_force_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                           size_t k, float min_score, uint32_t *out_vec, float *out_score) {
    const uint64_t *bits = filter ? filter : seg->alive;
    retr_t r = {seg, query, min_score, NULL, NULL};
    orc_rabitq_query rq;
    if (seg->quantized) { orc_rabitq_query_init(&rq, query, seg->dim); r.rq = &rq; }
    size_t cap = 1024, len = 0;
    cnx_t *scored = (cnx_t *)malloc(cap * sizeof(cnx_t));
    for (uint32_t p = 0; p < seg->n_paragraphs; p++) {
        if (bits && !bit_get(bits, p)) continue;
        uint32_t first = seg->para_first_vec ? seg->para_first_vec[p] : p;
        uint32_t num = seg->para_num_vec ? seg->para_num_vec[p] : 1;
        if (num == 0) continue;
        cnx_t best = {first, retr_sim(&r, first)};
        for (uint32_t v = first + 1; v < first + num; v++) {
            float s = retr_sim(&r, v);
            /* Iterator::max_by returns the LAST maximum */
            if (orc_total_cmp(s, best.score) >= 0) { best.addr = v; best.score = s; }
        }
        /* `upper_bound >= min_score` (:596): the upper bound is the score itself without RaBitQ */
        float bound = r.rq ? rabitq_upper_bound(&r, best.addr) : best.score;
        if (bound >= min_score) {
            if (len == cap) { cap *= 2; scored = (cnx_t *)realloc(scored, cap * sizeof(cnx_t)); }
            scored[len++] = best;
        }
    }
    if (r.rq) { /* rerank the candidates, in bitset order, with the raw vectors (:604-611) */
        float *ub = (float *)malloc((len ? len : 1) * sizeof(float));
        for (size_t i = 0; i < len; i++) ub[i] = rabitq_upper_bound(&r, scored[i].addr);
        retr_t exact = r;
        exact.rq = NULL;
        cnx_t *rr = (cnx_t *)malloc((k ? k : 1) * sizeof(cnx_t));
        size_t nr = k ? rerank_top(&exact, scored, ub, len, k, rr, NULL) : 0;
        for (size_t i = 0; i < nr; i++) { out_vec[i] = rr[i].addr; out_score[i] = rr[i].score; }
        free(ub); free(rr); free(scored);
        orc_rabitq_query_free(&rq);
        return (int)nr;
    }
    qsort(scored, len, sizeof(cnx_t), cmp_cnx_better_first);
    size_t n = len < k ? len : k;
    for (size_t i = 0; i < n; i++) { out_vec[i] = scored[i].addr; out_score[i] = scored[i].score; }
    free(scored);
    return (int)n;
}

/* maxsim_similarity (multivector.rs:33-46) */
float orc_maxsim(const float *query_vectors, size_t n_query, const float *doc_vectors, size_t n_doc, size_t dim, int similarity, int order) {
    float summaxsim = 0.0f;
    for (size_t q = 0; q < n_query; q++) {
        float maxsim = 0.0f;
        for (size_t v = 0; v < n_doc; v++) {
            float sim = orc_similarity(doc_vectors + v * dim, query_vectors + q * dim, dim, similarity, order);
            if (sim > maxsim) maxsim = sim;
        }
        summaxsim = summaxsim + maxsim;
    }
    return summaxsim;
}

/* OpenSegment::_search (segment.rs:496-567) */
int orc_segment_search(const orc_segment *seg, const float *query, const uint64_t *filter,
                       size_t k, float min_score, int with_duplicates,
                       uint32_t *out_vec, float *out_score, int *method_out) {
    const uint64_t *bits = filter ? filter : seg->alive;
    size_t matching = 0;
    for (uint32_t p = 0; p < seg->n_paragraphs; p++) matching += bits ? (size_t)bit_get(bits, p) : 1;
    if (method_out) *method_out = 0;
    if (matching == 0) return 0;
    if (seg->graph && orc_use_hnsw(seg->n_paragraphs, matching, k, seg->quantized != NULL)) {
        if (method_out) *method_out = 1;
        int n = orc_hnsw_search(seg, query, bits, k, min_score, with_duplicates, seg->para_num_vec != NULL, out_vec, out_score, NULL);
        return n > (int)k ? (int)k : n;
    }
    if (method_out) *method_out = 2;
    return orc_brute_force_search(seg, query, bits, k, min_score, out_vec, out_score);
}

/* ------------------------------------------------------------------------------------------
 * a12. HnswBuilder (hnsw/build.rs:28-167).  Sequential insertion (the reference inserts with
 * rayon and is non-deterministic, segment.rs:908); graph identity is therefore unpinned.
 * ------------------------------------------------------------------------------------------ */

/* rand 0.10 SmallRng on 64-bit = Xoshiro256++ seeded through SplitMix64 [third party, restated] */
typedef struct { uint64_t s[4]; } xoshiro_t;
static uint64_t splitmix64(uint64_t *state) {
    uint64_t z = (*state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void xoshiro_seed(xoshiro_t *r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xoshiro_next(xoshiro_t *r) {
    uint64_t *s = r->s;
    uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}
static double uniform01(xoshiro_t *r) {
    /* Uniform<f64>::new(0.0, 1.0): 52 random mantissa bits in [1,2) minus 1 */
    uint64_t bits = (xoshiro_next(r) >> 12) | 0x3ff0000000000000ULL;
    double d;
    memcpy(&d, &bits, 8);
    return (d - 1.0) * 1.0 + 0.0;
}

/* get_random_layer (build.rs:97-101): round(-ln(u) * 1/ln(M)) — `round`, not floor */
static uint32_t random_layer(xoshiro_t *r) {
    double sample = uniform01(r);
    double picked = -log(sample) * (1.0 / log((double)HNSW_M));
    double rounded = round(picked);
    if (!(rounded > 0.0)) return 0;
    if (rounded > 64.0) return 64;
    return (uint32_t)rounded;
}

void orc_hnsw_levels(uint64_t seed, uint32_t n, uint32_t *levels_out) {
    xoshiro_t r;
    xoshiro_seed(&r, seed);
    for (uint32_t i = 0; i < n; i++) levels_out[i] = random_layer(&r);
}

/* select_neighbours_heuristic (build.rs:57-95) */
static size_t select_neighbours(const orc_segment *seg, size_t k, const cnx_t *cands, size_t n_cands, cnx_t *results) {
    size_t n_res = 0;
    heap_t discarded;
    heap_init(&discarded, 1);
    for (size_t i = 0; i < n_cands; i++) {
        if (n_res == k) break;
        cnx_t x = cands[i];
        int check = 1;
        for (size_t j = 0; j < n_res; j++) {
            float inter = orc_similarity(seg_vec(seg, x.addr), seg_vec(seg, results[j].addr), seg->dim, seg->similarity, seg->order);
            if (!(x.score > inter)) { check = 0; break; }
        }
        if (check) results[n_res++] = x;
        else heap_push(&discarded, x);
    }
    if (n_res < k) {
        while (n_res < k && discarded.len > 0) results[n_res++] = heap_pop(&discarded);
        /* results.sort_unstable_by(|y, x| x.1.total_cmp(&y.1)) -> `better` order */
        qsort(results, n_res, sizeof(cnx_t), cmp_cnx_better_first);
    }
    heap_free(&discarded);
    return n_res;
}

static inline uint32_t m_max_for_layer(uint32_t layer) { return layer == 0 ? HNSW_M_MAX_0 : HNSW_M_MAX; }
static inline size_t prune_m(size_t m) { return m * 95 / 100; }

/* layer_insert (build.rs:104-119) */
static void layer_insert(const orc_segment *seg, layer_t *layer, uint32_t x, const cnx_t *found, size_t n_found, uint32_t mmax) {
    cnx_t neigh[HNSW_M + 1];
    size_t n = select_neighbours(seg, HNSW_M, found, n_found, neigh);
    layer_set_edges(layer, x, neigh, (uint32_t)n);
    cnx_t tmp[HNSW_M_MAX_0 + 2], pruned[HNSW_M_MAX_0 + 2];
    for (size_t i = 0; i < n; i++) {
        uint32_t y = neigh[i].addr;
        uint32_t s = (uint32_t)layer->slot_of[y];
        uint32_t d = layer->deg[s];
        for (uint32_t j = 0; j < d; j++) { tmp[j].addr = layer->adj[s][j]; tmp[j].score = layer->w[s][j]; }
        tmp[d].addr = x; tmp[d].score = neigh[i].score;
        d++;
        if (d > mmax) {
            size_t np = select_neighbours(seg, prune_m(mmax), tmp, d, pruned);
            layer_set_edges(layer, y, pruned, (uint32_t)np);
        } else {
            layer_set_edges(layer, y, tmp, d);
        }
    }
}

/* insert (build.rs:121-166) */
static void hnsw_insert(const orc_segment *seg, orc_hnsw *g, uint32_t node) {
    retr_t r = {seg, seg_vec(seg, node), -1.0f, NULL, NULL};
    uint32_t *eps = (uint32_t *)malloc(sizeof(uint32_t));
    size_t n_ep = 1;
    eps[0] = g->ep_node;
    cnx_t **found = (cnx_t **)calloc(g->n_layers, sizeof(cnx_t *));
    size_t *n_found = (size_t *)calloc(g->n_layers, sizeof(size_t));
    int node_in_layer = 0;
    for (uint32_t l = g->n_layers; l-- > 0;) {
        if (!node_in_layer && (l == 0 || layer_contains(&g->layers[l], node))) node_in_layer = 1;
        size_t k = node_in_layer ? EF_CONSTRUCTION : 1;
        cnx_t *res;
        size_t n = layer_search(&r, g, l, k, eps, n_ep, &res);
        eps = (uint32_t *)realloc(eps, (n ? n : 1) * sizeof(uint32_t));
        for (size_t i = 0; i < n; i++) eps[i] = res[i].addr;
        n_ep = n;
        if (node_in_layer) { found[l] = res; n_found[l] = n; }
        else free(res);
    }
    for (uint32_t l = 0; l < g->n_layers; l++) {
        if (found[l]) {
            layer_insert(seg, &g->layers[l], node, found[l], n_found[l], m_max_for_layer(l));
            free(found[l]);
        }
    }
    free(found); free(n_found); free(eps);
}

orc_hnsw *orc_hnsw_build(const orc_segment *seg, uint64_t level_seed) {
    orc_hnsw *g = orc_hnsw_new();
    xoshiro_t rng;
    xoshiro_seed(&rng, level_seed);
    /* initialize_graph (build.rs:50-55) */
    for (uint32_t i = 0; i < seg->n_vectors; i++) orc_hnsw_add_node(g, i, random_layer(&rng));
    if (seg->n_vectors == 0) return g;
    orc_hnsw_update_entry_point(g); /* single layer: stays RAMHnsw::new's default (node 0, layer 0) */
    for (uint32_t i = 0; i < seg->n_vectors; i++) hnsw_insert(seg, g, i);
    return g;
}

/* ------------------------------------------------------------------------------------------
 * a13. DiskHnswV2 byte format (hnsw/disk/v2.rs:16-49,109-245)
 * ------------------------------------------------------------------------------------------ */
static inline void put_u32(uint8_t *buf, size_t cap, size_t pos, uint32_t v) {
    if (buf && pos + 4 <= cap) { buf[pos] = (uint8_t)v; buf[pos + 1] = (uint8_t)(v >> 8); buf[pos + 2] = (uint8_t)(v >> 16); buf[pos + 3] = (uint8_t)(v >> 24); }
}
static inline uint32_t get_u32(const uint8_t *buf, size_t pos) {
    return (uint32_t)buf[pos] | ((uint32_t)buf[pos + 1] << 8) | ((uint32_t)buf[pos + 2] << 16) | ((uint32_t)buf[pos + 3] << 24);
}

size_t orc_hnsw_serialize_v2(const orc_hnsw *g, uint32_t num_nodes, uint8_t *graph, size_t graph_cap,
                             float *edges, size_t edges_cap, size_t *n_edges_out) {
    size_t pos = 0, ne = 0;
    if (num_nodes == 0) { if (n_edges_out) *n_edges_out = 0; return 0; }
    size_t *node_end = (size_t *)malloc(num_nodes * sizeof(size_t));
    size_t *layer_start = (size_t *)malloc((g->n_layers ? g->n_layers : 1) * sizeof(size_t));
    for (uint32_t node = 0; node < num_nodes; node++) {
        size_t node_begin = pos;
        for (uint32_t l = 0; l < g->n_layers; l++) {
            const uint32_t *e;
            uint32_t deg = graph_edges(g, l, node, &e);
            layer_start[l] = pos - node_begin;
            put_u32(graph, graph_cap, pos, deg); pos += 4;
            if (deg) {
                const layer_t *ly = &g->layers[l];
                uint32_t s = (uint32_t)ly->slot_of[node];
                for (uint32_t i = 0; i < deg; i++) {
                    put_u32(graph, graph_cap, pos, e[i]); pos += 4;
                    if (edges && ne < edges_cap) edges[ne] = ly->w[s][i];
                    ne++;
                }
            }
        }
        size_t node_len = (pos - node_begin) + (size_t)g->n_layers * 4;
        for (uint32_t l = g->n_layers; l-- > 0;) {
            put_u32(graph, graph_cap, pos, (uint32_t)(node_len - layer_start[l])); pos += 4;
        }
        node_end[node] = pos;
    }
    for (uint32_t node = num_nodes; node-- > 0;) { put_u32(graph, graph_cap, pos, (uint32_t)node_end[node]); pos += 4; }
    put_u32(graph, graph_cap, pos, g->ep_layer); pos += 4;
    put_u32(graph, graph_cap, pos, g->ep_node); pos += 4;
    free(node_end); free(layer_start);
    if (n_edges_out) *n_edges_out = ne;
    return pos;
}

void orc_disk_v2_entry_point(const uint8_t *graph, size_t len, uint32_t *node, uint32_t *layer) {
    *node = get_u32(graph, len - 4);
    *layer = get_u32(graph, len - 8);
}

uint32_t orc_disk_v2_edges(const uint8_t *graph, size_t len, uint32_t layer, uint32_t node, uint32_t *out, uint32_t cap) {
    size_t indexing_end = len - 8;
    size_t pos = indexing_end - ((size_t)node + 1) * 4;
    size_t node_end = get_u32(graph, pos);
    size_t lpos = node_end - ((size_t)layer + 1) * 4;
    size_t off = get_u32(graph, lpos);
    size_t start = node_end - off;
    uint32_t n = get_u32(graph, start);
    for (uint32_t i = 0; i < n && i < cap; i++) out[i] = get_u32(graph, start + 4 + (size_t)i * 4);
    return n;
}

orc_hnsw *orc_hnsw_deserialize_v2(const uint8_t *graph, size_t len, const float *edges, size_t n_edges) {
    orc_hnsw *g = orc_hnsw_new();
    if (len == 0) return g;
    size_t end = len, ne = 0;
    orc_disk_v2_entry_point(graph, len, &g->ep_node, &g->ep_layer);
    uint32_t node = 0;
    cnx_t tmp[4096];
    for (;;) {
        size_t indexing_pos = end - ((size_t)node + 3) * 4;
        size_t node_end = get_u32(graph, indexing_pos);
        uint32_t l = 0;
        for (;;) {
            if (g->n_layers == l) {
                g->layers = (layer_t *)realloc(g->layers, (l + 1) * sizeof(layer_t));
                memset(&g->layers[l], 0, sizeof(layer_t));
                g->n_layers = l + 1;
            }
            size_t layer_pos = node_end - ((size_t)l + 1) * 4;
            size_t off = get_u32(graph, layer_pos);
            size_t start = node_end - off;
            uint32_t n = get_u32(graph, start);
            size_t cnx_start = start + 4, cnx_end = cnx_start + (size_t)n * 4;
            if (l == 0 || n > 0) {
                layer_add_node(&g->layers[l], node);
                for (uint32_t i = 0; i < n && i < 4096; i++) {
                    tmp[i].addr = get_u32(graph, cnx_start + (size_t)i * 4);
                    tmp[i].score = (edges && ne < n_edges) ? edges[ne] : 0.0f;
                    ne++;
                }
                layer_set_edges(&g->layers[l], node, tmp, n);
            }
            if (cnx_end == layer_pos) break;
            l++;
        }
        if (node + 1 > g->n_nodes) g->n_nodes = node + 1;
        if (node_end == indexing_pos) break;
        node++;
    }
    return g;
}

/* ------------------------------------------------------------------------------------------
 * a9. Searcher::_search + Fssc (searcher.rs:149-199,241-290)
 * ------------------------------------------------------------------------------------------ */
int orc_searcher_search(const orc_segment *segs, const uint64_t *const *para_keys, size_t n_segs,
                        const float *query_in, const uint64_t *const *filters, size_t k, float min_score,
                        int with_duplicates, int normalize_query, orc_scored_paragraph *out) {
    if (n_segs == 0) return 0;
    uint32_t dim = segs[0].dim;
    float *query = (float *)malloc((size_t)dim * sizeof(float));
    if (normalize_query) orc_normalize(query_in, query, dim);
    else memcpy(query, query_in, (size_t)dim * sizeof(float));

    orc_scored_paragraph *buff = (orc_scored_paragraph *)malloc((k + 1) * sizeof(orc_scored_paragraph));
    size_t n_buff = 0;
    /* Fssc.seen: vector bytes already offered */
    size_t seen_cap = 64, n_seen = 0;
    const float **seen = (const float **)malloc(seen_cap * sizeof(float *));
    uint32_t *tv = (uint32_t *)malloc((k ? k : 1) * sizeof(uint32_t));
    float *ts = (float *)malloc((k ? k : 1) * sizeof(float));

    for (size_t s = 0; s < n_segs; s++) {
        const orc_segment *seg = &segs[s];
        int n = orc_segment_search(seg, query, filters ? filters[s] : NULL, k, min_score, with_duplicates, tv, ts, NULL);
        for (int i = 0; i < n; i++) {
            const float *vec = seg_vec(seg, tv[i]);
            if (!with_duplicates) {
                int dup = 0;
                for (size_t j = 0; j < n_seen; j++) if (memcmp(seen[j], vec, (size_t)dim * 4) == 0) { dup = 1; break; }
                if (dup) continue;
                if (n_seen == seen_cap) { seen_cap *= 2; seen = (const float **)realloc(seen, seen_cap * sizeof(float *)); }
                seen[n_seen++] = vec;
            }
            uint32_t p = seg_paragraph(seg, tv[i]);
            orc_scored_paragraph cand = {para_keys ? para_keys[s][p] : (((uint64_t)s << 32) | p), ts[i], (uint32_t)s, tv[i]};
            if (n_buff == k) {
                /* among buffered entries with a lower score than the candidate, evict the lowest */
                int victim = -1;
                for (size_t j = 0; j < n_buff; j++) {
                    if (cand.score > buff[j].score && (victim < 0 || buff[j].score < buff[victim].score)) victim = (int)j;
                }
                if (victim < 0) continue;
                buff[victim] = buff[--n_buff];
            }
            /* HashSet::insert keyed by paragraph id: no-op when the id is already buffered */
            int present = 0;
            for (size_t j = 0; j < n_buff; j++) if (buff[j].paragraph_key == cand.paragraph_key) { present = 1; break; }
            if (!present && k > 0) buff[n_buff++] = cand;
        }
    }
    /* sort desc by score; HashSet iteration order is unspecified -> ties by (segment, vector) */
    for (size_t i = 1; i < n_buff; i++) {
        orc_scored_paragraph v = buff[i];
        size_t j = i;
        while (j > 0) {
            const orc_scored_paragraph *u = &buff[j - 1];
            int lower = (u->score < v.score) ||
                        (u->score == v.score && (u->segment > v.segment || (u->segment == v.segment && u->vector > v.vector)));
            if (!lower) break;
            buff[j] = buff[j - 1];
            j--;
        }
        buff[j] = v;
    }
    memcpy(out, buff, n_buff * sizeof(orc_scored_paragraph));
    free(buff); free(seen); free(tv); free(ts); free(query);
    return (int)n_buff;
}

/* ------------------------------------------------------------------------------------------
 * a15/a16. BM25 — tantivy 0.26.1 [third party, restated]: K1=1.2, B=0.75, 1-byte fieldnorms.
 * ------------------------------------------------------------------------------------------ */
uint32_t orc_fieldnorm_from_id(uint8_t id) {
    /* FIELD_NORMS_TABLE: 0..=40 exact, then groups of 8 with doubling step (2,4,8,...) */
    if (id <= 40) return id;
    uint32_t i = (uint32_t)id - 41;
    uint32_t grp = i / 8, pos = i % 8;
    uint64_t base = 24u + ((uint64_t)1 << (grp + 4));
    uint64_t step = (uint64_t)1 << (grp + 1);
    return (uint32_t)(base + (uint64_t)(pos + 1) * step);
}

uint8_t orc_fieldnorm_to_id(uint32_t fieldnorm) {
    /* largest id whose table value is <= fieldnorm */
    int lo = 0, hi = 255;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (orc_fieldnorm_from_id((uint8_t)mid) <= fieldnorm) lo = mid; else hi = mid - 1;
    }
    return (uint8_t)lo;
}

#define BM25_K1 1.2f
#define BM25_B 0.75f

float orc_bm25_idf(uint64_t doc_freq, uint64_t doc_count) {
    float x = ((float)(doc_count - doc_freq) + 0.5f) / ((float)doc_freq + 0.5f);
    return logf(1.0f + x);
}

void orc_bm25_tf_cache(float average_fieldnorm, float cache[256]) {
    for (int id = 0; id < 256; id++) {
        float fieldnorm = (float)orc_fieldnorm_from_id((uint8_t)id);
        cache[id] = BM25_K1 * (1.0f - BM25_B + BM25_B * fieldnorm / average_fieldnorm);
    }
}

typedef struct { float score; uint64_t docaddr; } bm_hit_t;

/* TopDocs order: score desc, then DocAddress asc */
static inline int bm_better(bm_hit_t a, bm_hit_t b) {
    int c = orc_total_cmp(a.score, b.score);
    if (c != 0) return c > 0;
    return a.docaddr < b.docaddr;
}

/* is_after (nidx_paragraph/src/reader.rs:379-390) */
static int is_after(const orc_search_after *after, float score, uint64_t docaddr) {
    if (!after || !after->has_after) return 1;
    int c = orc_total_cmp(score, after->score);
    if (c < 0) return 1;
    if (c > 0) return 0;
    if (after->tie_break == 0) return 1;
    if (after->tie_break == 1) return docaddr > after->docaddr;
    return 0;
}

typedef struct { float score; uint64_t docaddr; int64_t value; } bm_ohit_t;
/* order_by_fast_field: the fast value decides (desc or asc), then the lower DocAddress */
static inline int bm_obetter(bm_ohit_t a, bm_ohit_t b, int desc) {
    if (a.value != b.value) return desc ? a.value > b.value : a.value < b.value;
    return a.docaddr < b.docaddr;
}

/* The statistics Bm25Weight is built from (tantivy Bm25Weight::for_terms over a Bm25StatisticsProvider): for a Searcher they are
 * SEARCHER-wide — total_num_docs = the sum of the segments' max_doc, total_num_tokens = the sum over the segments' inverted
 * indexes, doc_freq(term) = the sum of the segments' doc_freq — and every segment is then scored with the same weight and the
 * same fieldnorm cache (nidx_tantivy/src/index_reader.rs:39-74 opens all segments under one searcher; nidx_text/src/reader.rs:433-435
 * and nidx_paragraph/src/reader.rs:290-292,330-332 call searcher.search once).  st == NULL: the segment is the whole index. */
typedef struct {
    const orc_bm25_index *segs;
    size_t n_segs;
    uint64_t total_docs;
    float avg_fieldnorm;
} bm25_stats;

static uint64_t st_doc_freq(const bm25_stats *st, const orc_bm25_index *idx, uint32_t term) {
    if (!st) return idx->term_offsets[term + 1] - idx->term_offsets[term];
    uint64_t df = 0;
    for (size_t s = 0; s < st->n_segs; s++)
        if (term < st->segs[s].n_terms) df += st->segs[s].term_offsets[term + 1] - st->segs[s].term_offsets[term];
    return df;
}
static uint64_t st_total_docs(const bm25_stats *st, const orc_bm25_index *idx) { return st ? st->total_docs : idx->n_docs; }

static int bm25_segment_search(const orc_bm25_index *idx, const bm25_stats *st, const orc_bm25_clause *clauses, size_t n_clauses,
                               size_t k, const orc_search_after *after, uint32_t segment_ord,
                               const int64_t *order_values, int order_desc, uint64_t *match_bits_out,
                               uint64_t *out_docaddr, float *out_score, int64_t *out_order_value, uint64_t *total_out) {
    uint32_t n = idx->n_docs;
    float *acc = (float *)calloc(n ? n : 1, sizeof(float));
    uint8_t *should_hit = (uint8_t *)calloc(n ? n : 1, 1);
    uint16_t *must_cnt = (uint16_t *)calloc(n ? n : 1, sizeof(uint16_t));
    uint8_t *excluded = (uint8_t *)calloc(n ? n : 1, 1);
    uint8_t *group_hit = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *last_clause = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t)); /* term sets: a doc counts once per clause */
    uint8_t groups_present = 0;   /* required Should groups g = occur - ORC_OCCUR_SHOULD_GROUP (g < 8) that have clauses */
    float cache[256];
    float avg = st ? st->avg_fieldnorm : (idx->n_docs ? (float)idx->total_num_tokens / (float)idx->n_docs : 0.0f);
    orc_bm25_tf_cache(avg, cache);
    size_t n_must = 0, n_should = 0;
    if (match_bits_out) memset(match_bits_out, 0, (size_t)((n + 63) / 64) * 8);
    /* term-at-a-time in clause order: per-doc f32 sums accumulate in clause order */
    for (size_t c = 0; c < n_clauses; c++) {
        const orc_bm25_clause *cl = &clauses[c];
        const int is_set = cl->n_set_terms > 0;
        const size_t n_lists = is_set ? cl->n_set_terms : 1;
        if (cl->occur == ORC_OCCUR_MUST) n_must++;
        if (cl->occur == ORC_OCCUR_SHOULD) n_should++;
        if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) groups_present |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
        if (is_set && cl->set_phrase) {
            /* PhraseQuery: walk the first term's postings, look the document up in the others', count the start positions */
            float idf_sum = 0.0f;
            for (size_t t = 0; t < n_lists; t++) {
                idf_sum += orc_bm25_idf(st_doc_freq(st, idx, cl->set_terms[t]), st_total_docs(st, idx));
            }
            float weight = idf_sum * (1.0f + BM25_K1) * cl->boost;
            uint64_t b0 = idx->term_offsets[cl->set_terms[0]], e0 = idx->term_offsets[cl->set_terms[0] + 1];
            for (uint64_t i0 = b0; i0 < e0 && idx->pos_offsets; i0++) {
                uint32_t d = idx->doc_ids[i0];
                uint32_t count = 0;
                for (uint64_t pi = idx->pos_offsets[i0]; pi < idx->pos_offsets[i0 + 1]; pi++) {
                    uint32_t p = idx->positions[pi];
                    int all = 1;
                    for (size_t t = 1; t < n_lists && all; t++) {
                        uint64_t b = idx->term_offsets[cl->set_terms[t]], e = idx->term_offsets[cl->set_terms[t] + 1];
                        int found = 0;
                        for (uint64_t i = b; i < e && !found; i++) {
                            if (idx->doc_ids[i] != d) continue;
                            for (uint64_t pj = idx->pos_offsets[i]; pj < idx->pos_offsets[i + 1]; pj++)
                                if (idx->positions[pj] == p + (uint32_t)t) { found = 1; break; }
                        }
                        all = found;
                    }
                    count += (uint32_t)all;
                }
                if (count == 0) continue;
                if (cl->occur == ORC_OCCUR_MUST_NOT) { excluded[d] = 1; continue; }
                float tf = (float)count;
                acc[d] = acc[d] + weight * (tf / (tf + cache[idx->fieldnorm_ids[d]]));
                if (cl->occur == ORC_OCCUR_MUST) must_cnt[d]++;
                else if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) group_hit[d] |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
                else should_hit[d] = 1;
            }
            continue;
        }
        if (is_set && cl->set_complement) {
            /* every document outside the union, once */
            for (size_t t = 0; t < n_lists; t++) {
                uint64_t b = idx->term_offsets[cl->set_terms[t]], e = idx->term_offsets[cl->set_terms[t] + 1];
                for (uint64_t i = b; i < e; i++) last_clause[idx->doc_ids[i]] = (uint32_t)c + 1;
            }
            for (uint32_t d = 0; d < n; d++) {
                if (last_clause[d] == (uint32_t)c + 1) continue;
                if (cl->occur == ORC_OCCUR_MUST_NOT) { excluded[d] = 1; continue; }
                acc[d] = acc[d] + cl->boost;
                if (cl->occur == ORC_OCCUR_MUST) must_cnt[d]++;
                else if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) group_hit[d] |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
                else should_hit[d] = 1;
            }
            continue;
        }
        for (size_t t = 0; t < n_lists; t++) {
            uint32_t term = is_set ? cl->set_terms[t] : cl->term;
            uint64_t b = idx->term_offsets[term], e = idx->term_offsets[term + 1];
            float weight = 0.0f;
            if (!is_set && cl->mode != ORC_CONST_SCORE) weight = orc_bm25_idf(st_doc_freq(st, idx, term), st_total_docs(st, idx)) * (1.0f + BM25_K1) * cl->boost;
            for (uint64_t i = b; i < e; i++) {
                uint32_t d = idx->doc_ids[i];
                if (is_set) {
                    if (last_clause[d] == (uint32_t)c + 1) continue; /* BitSet insert: already in this clause's doc set */
                    last_clause[d] = (uint32_t)c + 1;
                }
                if (cl->occur == ORC_OCCUR_MUST_NOT) { excluded[d] = 1; continue; }
                float s;
                if (is_set || cl->mode == ORC_CONST_SCORE) s = cl->boost;
                else {
                    float tf = cl->mode == ORC_TF_BASIC ? 1.0f : (float)idx->tfs[i];
                    s = weight * (tf / (tf + cache[idx->fieldnorm_ids[d]]));
                }
                acc[d] = acc[d] + s;
                if (cl->occur == ORC_OCCUR_MUST) must_cnt[d]++;
                else if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) group_hit[d] |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
                else should_hit[d] = 1;
            }
        }
    }
    bm_ohit_t *top = (bm_ohit_t *)malloc((k + 1) * sizeof(bm_ohit_t));
    size_t n_top = 0;
    uint64_t total = 0;
    for (uint32_t d = 0; d < n; d++) {
        if (excluded[d]) continue;
        if (must_cnt[d] != n_must) continue;
        if ((group_hit[d] & groups_present) != groups_present) continue;   /* a clause of EVERY required group */
        if (n_must == 0 && groups_present == 0 && !should_hit[d]) continue;
        if (idx->alive && !bit_get(idx->alive, d)) continue;
        total++;
        if (match_bits_out) match_bits_out[d >> 6] |= (uint64_t)1 << (d & 63);
        uint64_t docaddr = ((uint64_t)segment_ord << 32) | d;
        float s = acc[d];
        if (!order_values && !is_after(after, s, docaddr)) s = -INFINITY; /* tweak_score (reader.rs:350-376) */
        bm_ohit_t h = {s, docaddr, order_values ? order_values[d] : 0};
        if (k == 0) continue;
        int better_than_last = 1;
        if (n_top == k) {
            if (order_values) better_than_last = bm_obetter(h, top[n_top - 1], order_desc);
            else { bm_hit_t a = {h.score, h.docaddr}, b = {top[n_top - 1].score, top[n_top - 1].docaddr}; better_than_last = bm_better(a, b); }
        }
        if (!better_than_last) continue;
        size_t j = n_top < k ? n_top++ : k - 1;
        while (j > 0) {
            int bt;
            if (order_values) bt = bm_obetter(h, top[j - 1], order_desc);
            else { bm_hit_t a = {h.score, h.docaddr}, b = {top[j - 1].score, top[j - 1].docaddr}; bt = bm_better(a, b); }
            if (!bt) break;
            top[j] = top[j - 1];
            j--;
        }
        top[j] = h;
    }
    (void)n_should;
    for (size_t i = 0; i < n_top; i++) {
        out_docaddr[i] = top[i].docaddr;
        out_score[i] = top[i].score;
        if (out_order_value) out_order_value[i] = top[i].value;
    }
    if (total_out) *total_out = total;
    free(top); free(acc); free(should_hit); free(must_cnt); free(excluded); free(group_hit); free(last_clause);
    return (int)n_top;
}

int orc_bm25_search_ex(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                       size_t k, const orc_search_after *after, uint32_t segment_ord,
                       const int64_t *order_values, int order_desc, uint64_t *match_bits_out,
                       uint64_t *out_docaddr, float *out_score, int64_t *out_order_value, uint64_t *total_out) {
    return bm25_segment_search(idx, NULL, clauses, n_clauses, k, after, segment_ord, order_values, order_desc, match_bits_out,
                               out_docaddr, out_score, out_order_value, total_out);
}

void orc_bm25_searcher_stats(const orc_bm25_index *segs, size_t n_segs, uint64_t *total_docs, uint64_t *total_tokens, float *avg_fieldnorm) {
    uint64_t docs = 0, tokens = 0;
    for (size_t s = 0; s < n_segs; s++) docs += segs[s].n_docs, tokens += segs[s].total_num_tokens;
    if (total_docs) *total_docs = docs;
    if (total_tokens) *total_tokens = tokens;
    if (avg_fieldnorm) *avg_fieldnorm = docs ? (float)tokens / (float)docs : 0.0f;
}

uint64_t orc_bm25_searcher_doc_freq(const orc_bm25_index *segs, size_t n_segs, uint32_t term) {
    bm25_stats st = {segs, n_segs, 0, 0.0f};
    return st_doc_freq(&st, NULL, term);
}

/* searcher.search(query, (TopDocs, Count[, facets])) over all segments of an index: one weight from the searcher's statistics,
 * collect_segment per segment (segment_ord = its position), merge_fruits — TopDocs keeps the k best by (score desc, DocAddress
 * asc) resp. (fast value, DocAddress asc), Count adds up. */
int orc_bm25_searcher_search_ex(const orc_bm25_index *segs, size_t n_segs, const orc_bm25_clause *clauses, size_t n_clauses,
                                size_t k, const orc_search_after *after, const int64_t *const *order_values, int order_desc,
                                uint64_t *const *match_bits_out, uint64_t *out_docaddr, float *out_score,
                                int64_t *out_order_value, uint64_t *total_out) {
    bm25_stats st = {segs, n_segs, 0, 0.0f};
    orc_bm25_searcher_stats(segs, n_segs, &st.total_docs, NULL, &st.avg_fieldnorm);
    bm_ohit_t *all = (bm_ohit_t *)malloc((n_segs * k + 1) * sizeof(bm_ohit_t));
    uint64_t *d = (uint64_t *)malloc((k + 1) * sizeof(uint64_t));
    float *sc = (float *)malloc((k + 1) * sizeof(float));
    int64_t *ov = (int64_t *)calloc(k + 1, sizeof(int64_t));
    size_t n_all = 0;
    uint64_t total = 0;
    const int by_value = order_values != NULL;
    for (size_t s = 0; s < n_segs; s++) {
        uint64_t t = 0;
        int m = bm25_segment_search(&segs[s], &st, clauses, n_clauses, k, after, (uint32_t)s, by_value ? order_values[s] : NULL, order_desc,
                                    match_bits_out ? match_bits_out[s] : NULL, d, sc, ov, &t);
        total += t;
        for (int i = 0; i < m; i++) { bm_ohit_t h = {sc[i], d[i], ov[i]}; all[n_all++] = h; }
    }
    /* merge_fruits: insertion sort by the collector's order (stable input order does not matter: DocAddresses are distinct) */
    for (size_t i = 1; i < n_all; i++) {
        bm_ohit_t h = all[i];
        size_t j = i;
        while (j > 0) {
            int bt;
            if (by_value) bt = bm_obetter(h, all[j - 1], order_desc);
            else { bm_hit_t a = {h.score, h.docaddr}, b = {all[j - 1].score, all[j - 1].docaddr}; bt = bm_better(a, b); }
            if (!bt) break;
            all[j] = all[j - 1];
            j--;
        }
        all[j] = h;
    }
    size_t n_out = n_all < k ? n_all : k;
    for (size_t i = 0; i < n_out; i++) {
        out_docaddr[i] = all[i].docaddr;
        out_score[i] = all[i].score;
        if (out_order_value) out_order_value[i] = all[i].value;
    }
    if (total_out) *total_out = total;
    free(all); free(d); free(sc); free(ov);
    return (int)n_out;
}

int orc_bm25_search(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                    size_t k, const orc_search_after *after, uint32_t segment_ord,
                    uint64_t *out_docaddr, float *out_score, uint64_t *total_out) {
    return orc_bm25_search_ex(idx, clauses, n_clauses, k, after, segment_ord, NULL, 0, NULL, out_docaddr, out_score, NULL, total_out);
}

/* ---- FuzzyTermQuery's automaton, restated as a dynamic program over unicode scalar values ---- */
static size_t utf8_decode(const uint8_t *s, size_t len, uint32_t *out, size_t cap) {
    size_t n = 0, i = 0;
    while (i < len && n < cap) {
        uint32_t c = s[i];
        size_t extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : 0;
        if (extra == 1) c &= 0x1f; else if (extra == 2) c &= 0x0f; else if (extra == 3) c &= 0x07;
        i++;
        for (size_t e = 0; e < extra && i < len; e++, i++) c = (c << 6) | (s[i] & 0x3f);
        out[n++] = c;
    }
    return n;
}

int orc_fuzzy_match(const uint8_t *query, size_t qlen, const uint8_t *term, size_t tlen, int distance, int prefix) {
    uint32_t q[256], t[256];
    size_t nq = utf8_decode(query, qlen, q, 256), nt = utf8_decode(term, tlen, t, 256);
    /* d[i][j] = restricted Damerau-Levenshtein distance between q[0..i) and t[0..j) */
    static __thread int d[257][257];
    for (size_t i = 0; i <= nq; i++) d[i][0] = (int)i;
    for (size_t j = 0; j <= nt; j++) d[0][j] = (int)j;
    for (size_t i = 1; i <= nq; i++)
        for (size_t j = 1; j <= nt; j++) {
            int cost = q[i - 1] == t[j - 1] ? 0 : 1;
            int v = d[i - 1][j] + 1;
            if (d[i][j - 1] + 1 < v) v = d[i][j - 1] + 1;
            if (d[i - 1][j - 1] + cost < v) v = d[i - 1][j - 1] + cost;
            if (i > 1 && j > 1 && q[i - 1] == t[j - 2] && q[i - 2] == t[j - 1] && d[i - 2][j - 2] + 1 < v) v = d[i - 2][j - 2] + 1;
            d[i][j] = v;
        }
    if (!prefix) return d[nq][nt] <= distance;
    for (size_t j = 0; j <= nt; j++)
        if (d[nq][j] <= distance) return 1;
    return 0;
}

size_t orc_fuzzy_terms(const uint8_t *bytes, const uint64_t *offsets, size_t n_terms, const uint8_t *query, size_t qlen,
                       int distance, int prefix, uint32_t *out, size_t cap) {
    size_t n = 0;
    for (size_t i = 0; i < n_terms; i++)
        if (orc_fuzzy_match(query, qlen, bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i]), distance, prefix)) {
            if (n < cap) out[n] = (uint32_t)i;
            n++;
        }
    return n;
}

/* ---- prefilter ------------------------------------------------------------------------------------------------- */
/* index of document d in the posting list [b, e), or e when absent */
static uint64_t posting_of(const orc_bm25_index *idx, uint64_t b, uint64_t e, uint32_t d) {
    uint64_t lo = b, hi = e;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (idx->doc_ids[mid] < d) lo = mid + 1;
        else hi = mid;
    }
    return (lo < e && idx->doc_ids[lo] == d) ? lo : e;
}

/* does document d hold terms[0..n) at consecutive positions? */
static int doc_has_phrase(const orc_bm25_index *idx, const uint32_t *terms, size_t n, uint32_t d) {
    if (!idx->pos_offsets || n == 0) return 0;
    uint64_t at[16];
    if (n > 16) return 0;
    for (size_t t = 0; t < n; t++) {
        uint64_t b = idx->term_offsets[terms[t]], e = idx->term_offsets[terms[t] + 1];
        at[t] = posting_of(idx, b, e, d);
        if (at[t] == e) return 0;
    }
    for (uint64_t pi = idx->pos_offsets[at[0]]; pi < idx->pos_offsets[at[0] + 1]; pi++) {
        uint32_t p = idx->positions[pi];
        int all = 1;
        for (size_t t = 1; t < n && all; t++) {
            int found = 0;
            for (uint64_t pj = idx->pos_offsets[at[t]]; pj < idx->pos_offsets[at[t] + 1] && !found; pj++)
                found = idx->positions[pj] == p + (uint32_t)t;
            all = found;
        }
        if (all) return 1;
    }
    return 0;
}

size_t orc_bm25_prefilter(const orc_bm25_index *idx, const orc_filter_op *ops, size_t n_ops, const uint32_t *lists,
                          const orc_date_range *ranges, const int64_t *created, const int64_t *modified,
                          const uint32_t *phrase_terms, const uint64_t *phrase_offsets, uint32_t *out_docs, size_t cap,
                          uint64_t *live_out) {
    size_t n_out = 0;
    uint64_t live = 0;
    int *stack = (int *)malloc((n_ops + 1) * sizeof(int));
    for (uint32_t d = 0; d < idx->n_docs; d++) {
        if (idx->alive && !((idx->alive[d >> 6] >> (d & 63)) & 1)) continue; /* deleted documents are not searched */
        live++;
        int sp = 0;
        if (n_ops == 0) stack[sp++] = 1;
        for (size_t i = 0; i < n_ops; i++) {
            const orc_filter_op *op = &ops[i];
            switch (op->op) {
                case ORC_FILTER_TERMS: {
                    int hit = 0;
                    for (uint32_t l = op->a; l < op->b && !hit; l++) {
                        uint64_t b = idx->term_offsets[lists[l]], e = idx->term_offsets[lists[l] + 1];
                        hit = posting_of(idx, b, e, d) != e;
                    }
                    stack[sp++] = hit;
                    break;
                }
                case ORC_FILTER_RANGE: {
                    const orc_date_range *r = &ranges[op->a];
                    const int64_t *vals = r->field == 0 ? created : modified;
                    int hit = 1;
                    if (r->has_since || r->has_until) {
                        int64_t v = vals[d];
                        if (r->has_since && v < r->since) hit = 0;
                        if (r->has_until && v > r->until) hit = 0;
                    }
                    stack[sp++] = hit;
                    break;
                }
                case ORC_FILTER_PHRASE:
                    stack[sp++] = doc_has_phrase(idx, phrase_terms + phrase_offsets[op->a],
                                                 (size_t)(phrase_offsets[op->a + 1] - phrase_offsets[op->a]), d);
                    break;
                case ORC_FILTER_ALL: stack[sp++] = 1; break;
                case ORC_FILTER_NONE: stack[sp++] = 0; break;
                case ORC_FILTER_AND: sp--; stack[sp - 1] = stack[sp - 1] && stack[sp]; break;
                case ORC_FILTER_OR: sp--; stack[sp - 1] = stack[sp - 1] || stack[sp]; break;
                case ORC_FILTER_NOT: stack[sp - 1] = !stack[sp - 1]; break;
                default: break;
            }
        }
        if (sp == 1 && stack[0]) {
            if (n_out < cap) out_docs[n_out] = d;
            n_out++;
        }
    }
    free(stack);
    if (live_out) *live_out = live;
    return n_out;
}

/* Document-at-a-time form of orc_bm25_search (what tantivy's union/intersection scorers do): the clause
 * cursors advance together over ascending doc ids and every doc's clause scores are summed in clause
 * order — the same f32 arithmetic as the term-at-a-time loop above, without the dense accumulator.  Used
 * as the CPU baseline of bench.py (a dense 4*n_docs-byte accumulator per query would be a strawman). */
static int bm25_segment_search_daat(const orc_bm25_index *idx, const bm25_stats *st, const orc_bm25_clause *clauses, size_t n_clauses,
                                    size_t k, const orc_search_after *after, uint32_t segment_ord,
                                    uint64_t *out_docaddr, float *out_score, uint64_t *total_out) {
    float cache[256];
    float avg = st ? st->avg_fieldnorm : (idx->n_docs ? (float)idx->total_num_tokens / (float)idx->n_docs : 0.0f);
    orc_bm25_tf_cache(avg, cache);
    uint64_t *cur = (uint64_t *)malloc((n_clauses ? n_clauses : 1) * sizeof(uint64_t));
    uint64_t *end = (uint64_t *)malloc((n_clauses ? n_clauses : 1) * sizeof(uint64_t));
    float *weight = (float *)malloc((n_clauses ? n_clauses : 1) * sizeof(float));
    size_t n_must = 0;
    uint8_t groups_present = 0;
    for (size_t c = 0; c < n_clauses; c++) {
        const orc_bm25_clause *cl = &clauses[c];
        cur[c] = idx->term_offsets[cl->term];
        end[c] = idx->term_offsets[cl->term + 1];
        weight[c] = cl->mode == ORC_CONST_SCORE ? cl->boost : orc_bm25_idf(st_doc_freq(st, idx, cl->term), st_total_docs(st, idx)) * (1.0f + BM25_K1) * cl->boost;
        if (cl->occur == ORC_OCCUR_MUST) n_must++;
        if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) groups_present |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
    }
    bm_hit_t *top = (bm_hit_t *)malloc((k + 1) * sizeof(bm_hit_t));
    size_t n_top = 0;
    uint64_t total = 0;
    for (;;) {
        uint32_t d = 0xffffffffu;
        for (size_t c = 0; c < n_clauses; c++)
            if (cur[c] < end[c] && idx->doc_ids[cur[c]] < d) d = idx->doc_ids[cur[c]];
        if (d == 0xffffffffu) break;
        float acc = 0.0f;
        size_t must = 0;
        int should = 0, excluded = 0;
        uint8_t group = 0;
        for (size_t c = 0; c < n_clauses; c++) {
            if (cur[c] >= end[c] || idx->doc_ids[cur[c]] != d) continue;
            const orc_bm25_clause *cl = &clauses[c];
            uint64_t i = cur[c]++;
            if (cl->occur == ORC_OCCUR_MUST_NOT) { excluded = 1; continue; }
            float s;
            if (cl->mode == ORC_CONST_SCORE) s = cl->boost;
            else {
                float tf = cl->mode == ORC_TF_BASIC ? 1.0f : (float)idx->tfs[i];
                s = weight[c] * (tf / (tf + cache[idx->fieldnorm_ids[d]]));
            }
            acc = acc + s;
            if (cl->occur == ORC_OCCUR_MUST) must++;
            else if (cl->occur >= ORC_OCCUR_SHOULD_GROUP) group |= (uint8_t)(1u << (cl->occur - ORC_OCCUR_SHOULD_GROUP));
            else should = 1;
        }
        if (excluded || must != n_must) continue;
        if ((group & groups_present) != groups_present) continue;
        if (n_must == 0 && groups_present == 0 && !should) continue;
        if (idx->alive && !bit_get(idx->alive, d)) continue;
        total++;
        uint64_t docaddr = ((uint64_t)segment_ord << 32) | d;
        if (!is_after(after, acc, docaddr)) acc = -INFINITY;
        bm_hit_t h = {acc, docaddr};
        if (k == 0) continue;
        if (n_top == k && !bm_better(h, top[n_top - 1])) continue;
        size_t j = n_top < k ? n_top++ : k - 1;
        while (j > 0 && bm_better(h, top[j - 1])) { top[j] = top[j - 1]; j--; }
        top[j] = h;
    }
    for (size_t i = 0; i < n_top; i++) { out_docaddr[i] = top[i].docaddr; out_score[i] = top[i].score; }
    if (total_out) *total_out = total;
    free(top); free(cur); free(end); free(weight);
    return (int)n_top;
}

int orc_bm25_search_daat(const orc_bm25_index *idx, const orc_bm25_clause *clauses, size_t n_clauses,
                         size_t k, const orc_search_after *after, uint32_t segment_ord,
                         uint64_t *out_docaddr, float *out_score, uint64_t *total_out) {
    return bm25_segment_search_daat(idx, NULL, clauses, n_clauses, k, after, segment_ord, out_docaddr, out_score, total_out);
}

/* orc_bm25_searcher_search_ex for plain term clauses ordered by score, document at a time per segment (benchmark-scale checks) */
int orc_bm25_searcher_search_daat(const orc_bm25_index *segs, size_t n_segs, const orc_bm25_clause *clauses, size_t n_clauses,
                                  size_t k, const orc_search_after *after, uint64_t *out_docaddr, float *out_score, uint64_t *total_out) {
    bm25_stats st = {segs, n_segs, 0, 0.0f};
    orc_bm25_searcher_stats(segs, n_segs, &st.total_docs, NULL, &st.avg_fieldnorm);
    bm_hit_t *all = (bm_hit_t *)malloc((n_segs * k + 1) * sizeof(bm_hit_t));
    uint64_t *d = (uint64_t *)malloc((k + 1) * sizeof(uint64_t));
    float *sc = (float *)malloc((k + 1) * sizeof(float));
    size_t n_all = 0;
    uint64_t total = 0;
    for (size_t s = 0; s < n_segs; s++) {
        uint64_t t = 0;
        int m = bm25_segment_search_daat(&segs[s], &st, clauses, n_clauses, k, after, (uint32_t)s, d, sc, &t);
        total += t;
        for (int i = 0; i < m; i++) { bm_hit_t h = {sc[i], d[i]}; all[n_all++] = h; }
    }
    for (size_t i = 1; i < n_all; i++) {
        bm_hit_t h = all[i];
        size_t j = i;
        while (j > 0 && bm_better(h, all[j - 1])) { all[j] = all[j - 1]; j--; }
        all[j] = h;
    }
    size_t n_out = n_all < k ? n_all : k;
    for (size_t i = 0; i < n_out; i++) { out_docaddr[i] = all[i].docaddr; out_score[i] = all[i].score; }
    if (total_out) *total_out = total;
    free(all); free(d); free(sc);
    return (int)n_out;
}

/* ------------------------------------------------------------------------------------------
 * a18. shard merge (nidx/src/searcher/shard_merge.rs:332-348, 197-250, 274-330) over
 * itertools::kmerge_by [third party, restated: binary heap of list heads, sift_down]
 * ------------------------------------------------------------------------------------------ */
typedef struct { size_t list, pos; } head_t;
typedef int (*less_fn)(const void *ctx, head_t a, head_t b);

static void km_sift_down(head_t *heap, size_t len, size_t index, less_fn less, const void *ctx) {
    size_t pos = index, child = 2 * pos + 1;
    while (child + 1 < len) {
        child += (size_t)(less(ctx, heap[child + 1], heap[child]) != 0);
        if (!less(ctx, heap[child], heap[pos])) return;
        head_t t = heap[pos]; heap[pos] = heap[child]; heap[child] = t;
        pos = child;
        child = 2 * pos + 1;
    }
    if (child + 1 == len && less(ctx, heap[child], heap[pos])) {
        head_t t = heap[pos]; heap[pos] = heap[child]; heap[child] = t;
    }
}

static size_t kmerge(const size_t *lens, size_t n_lists, size_t limit, less_fn less, const void *ctx, head_t *order_out) {
    head_t *heap = (head_t *)malloc((n_lists ? n_lists : 1) * sizeof(head_t));
    size_t len = 0;
    for (size_t l = 0; l < n_lists; l++) if (lens[l] > 0) { heap[len].list = l; heap[len].pos = 0; len++; }
    for (size_t i = len / 2; i-- > 0;) km_sift_down(heap, len, i, less, ctx);
    size_t n_out = 0;
    while (len > 0 && n_out < limit) {
        order_out[n_out++] = heap[0];
        if (heap[0].pos + 1 < lens[heap[0].list]) heap[0].pos++;
        else { heap[0] = heap[len - 1]; len--; }
        km_sift_down(heap, len, 0, less, ctx);
    }
    free(heap);
    return n_out;
}

static int vec_less(const void *ctx, head_t a, head_t b) {
    const orc_vec_hit *const *lists = (const orc_vec_hit *const *)ctx;
    return lists[a.list][a.pos].score >= lists[b.list][b.pos].score; /* kmerge_by(|a, b| a.score >= b.score) */
}

size_t orc_merge_vector(const orc_vec_hit *const *lists, const size_t *lens, size_t n_lists, size_t limit, orc_vec_hit *out) {
    head_t *order = (head_t *)malloc((limit ? limit : 1) * sizeof(head_t));
    size_t n = kmerge(lens, n_lists, limit, vec_less, lists, order);
    for (size_t i = 0; i < n; i++) out[i] = lists[order[i].list][order[i].pos];
    free(order);
    return n;
}

static int bytes_cmp(const uint8_t *a, size_t la, const uint8_t *b, size_t lb) {
    size_t m = la < lb ? la : lb;
    int c = m ? memcmp(a, b, m) : 0;
    if (c != 0) return c;
    return (la > lb) - (la < lb);
}

static int bm25_less(const void *ctx, head_t a, head_t b) {
    /* a before b iff bm25.total_cmp, then shard_id cmp, then docaddr reversed is Greater */
    const orc_bm25_hit *const *lists = (const orc_bm25_hit *const *)ctx;
    const orc_bm25_hit *x = &lists[a.list][a.pos], *y = &lists[b.list][b.pos];
    int c = orc_total_cmp(x->bm25, y->bm25);
    if (c == 0) c = bytes_cmp(x->shard_id, x->shard_id_len, y->shard_id, y->shard_id_len);
    if (c == 0) c = -((x->docaddr > y->docaddr) - (x->docaddr < y->docaddr));
    return c > 0;
}

size_t orc_merge_bm25(const orc_bm25_hit *const *lists, const size_t *lens, size_t n_lists, size_t limit, orc_bm25_hit *out) {
    head_t *order = (head_t *)malloc((limit ? limit : 1) * sizeof(head_t));
    size_t n = kmerge(lens, n_lists, limit, bm25_less, lists, order);
    for (size_t i = 0; i < n; i++) out[i] = lists[order[i].list][order[i].pos];
    free(order);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Batch runners (bench.py's cpu_baseline leg and the benchmark-scale parity checks): the per-query
 * functions above, one query per work item, on `threads` POSIX threads pulling items from a shared
 * counter — the reference serves one request per blocking thread (src/searcher/shard_search.rs:139-153).
 * Nothing here changes a result: every item calls exactly the single-query function.
 * ------------------------------------------------------------------------------------------ */
#include <pthread.h>

typedef struct batch_job batch_job;
typedef void (*batch_fn)(const batch_job *job, size_t i);
struct batch_job {
    batch_fn fn;
    size_t n;
    size_t next;          /* atomic work counter */
    /* vector searches */
    const orc_segment *segs;
    const uint64_t *const *para_keys;
    size_t n_segs;
    const float *queries;
    size_t k;
    float min_score;
    int with_duplicates;
    uint32_t *out_vec;
    float *out_score;
    uint32_t *out_count;
    orc_scored_paragraph *out_sp;
    orc_stats *stats;     /* [n] or NULL */
    /* bm25 */
    const orc_bm25_index *bm;
    const orc_bm25_clause *clauses;
    const uint64_t *clause_offsets;
    uint64_t *out_docaddr;
    uint64_t *out_total;
};

static void *batch_worker(void *p) {
    batch_job *job = (batch_job *)p;
    for (;;) {
        size_t i = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (i >= job->n) break;
        job->fn(job, i);
    }
    return NULL;
}

static void batch_run(batch_job *job, unsigned threads) {
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    if ((size_t)threads > job->n) threads = (unsigned)(job->n ? job->n : 1);
    job->next = 0;
    pthread_t *t = (pthread_t *)malloc(threads * sizeof(pthread_t));
    unsigned started = 0;
    for (unsigned i = 1; i < threads; i++)
        if (pthread_create(&t[started], NULL, batch_worker, job) == 0) started++;
    batch_worker(job);
    for (unsigned i = 0; i < started; i++) pthread_join(t[i], NULL);
    free(t);
}

static void job_hnsw(const batch_job *j, size_t i) {
    const orc_segment *seg = j->segs;
    j->out_count[i] = (uint32_t)orc_hnsw_search(seg, j->queries + i * seg->dim, NULL, j->k, j->min_score, j->with_duplicates, 0,
                                                j->out_vec + i * j->k, j->out_score + i * j->k, j->stats ? &j->stats[i] : NULL);
}
static void job_brute(const batch_job *j, size_t i) {
    const orc_segment *seg = j->segs;
    j->out_count[i] = (uint32_t)orc_brute_force_search(seg, j->queries + i * seg->dim, NULL, j->k, j->min_score,
                                                       j->out_vec + i * j->k, j->out_score + i * j->k);
}
static void job_searcher(const batch_job *j, size_t i) {
    j->out_count[i] = (uint32_t)orc_searcher_search(j->segs, j->para_keys, j->n_segs, j->queries + i * j->segs[0].dim, NULL, j->k,
                                                    j->min_score, j->with_duplicates, 0, j->out_sp + i * j->k);
}
static void job_bm25(const batch_job *j, size_t i) {
    const uint64_t b = j->clause_offsets[i], e = j->clause_offsets[i + 1];
    j->out_count[i] = (uint32_t)orc_bm25_search_daat(j->bm, j->clauses + b, (size_t)(e - b), j->k, NULL, 0, j->out_docaddr + i * j->k,
                                                     j->out_score + i * j->k, &j->out_total[i]);
}

void orc_hnsw_search_batch(const orc_segment *seg, const float *queries, size_t n_queries, size_t k, float min_score,
                           int with_duplicates, unsigned threads, uint32_t *out_vec, float *out_score, uint32_t *out_count,
                           orc_stats *stats) {
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.fn = job_hnsw; j.n = n_queries; j.segs = seg; j.queries = queries; j.k = k; j.min_score = min_score;
    j.with_duplicates = with_duplicates; j.out_vec = out_vec; j.out_score = out_score; j.out_count = out_count; j.stats = stats;
    if (stats) memset(stats, 0, n_queries * sizeof(orc_stats));
    batch_run(&j, threads);
}

void orc_brute_force_batch(const orc_segment *seg, const float *queries, size_t n_queries, size_t k, float min_score,
                           unsigned threads, uint32_t *out_vec, float *out_score, uint32_t *out_count) {
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.fn = job_brute; j.n = n_queries; j.segs = seg; j.queries = queries; j.k = k; j.min_score = min_score;
    j.out_vec = out_vec; j.out_score = out_score; j.out_count = out_count;
    batch_run(&j, threads);
}

void orc_searcher_search_batch(const orc_segment *segs, const uint64_t *const *para_keys, size_t n_segs, const float *queries,
                               size_t n_queries, size_t k, float min_score, int with_duplicates, unsigned threads,
                               orc_scored_paragraph *out, uint32_t *out_count) {
    batch_job j;
    memset(&j, 0, sizeof(j));
    if (n_segs == 0) { for (size_t i = 0; i < n_queries; i++) out_count[i] = 0; return; }
    j.fn = job_searcher; j.n = n_queries; j.segs = segs; j.para_keys = para_keys; j.n_segs = n_segs; j.queries = queries; j.k = k;
    j.min_score = min_score; j.with_duplicates = with_duplicates; j.out_sp = out; j.out_count = out_count;
    batch_run(&j, threads);
}

void orc_bm25_search_daat_batch(const orc_bm25_index *idx, const orc_bm25_clause *clauses, const uint64_t *clause_offsets,
                                size_t n_queries, size_t k, unsigned threads, uint64_t *out_docaddr, float *out_score,
                                uint32_t *out_count, uint64_t *out_total) {
    batch_job j;
    memset(&j, 0, sizeof(j));
    j.fn = job_bm25; j.n = n_queries; j.bm = idx; j.clauses = clauses; j.clause_offsets = clause_offsets; j.k = k;
    j.out_docaddr = out_docaddr; j.out_score = out_score; j.out_count = out_count; j.out_total = out_total;
    batch_run(&j, threads);
}

/* ------------------------------------------------------------------------------------------
 * a8 / f1. ParagraphInvertedIndexes::filter (nidx_vector/src/inverted_index/paragraph.rs:124-184) restated
 * DOCUMENT AT A TIME over the paragraph store — no posting lists, no FST: for every paragraph the formula is
 * evaluated on the paragraph's own key and labels, which is what the inverted indexes are an index OF
 * (ParagraphInvertedIndexes::build, :68-103: field_index keyed by FieldKey::from_field_id(paragraph id),
 * label_index keyed by labels_key(label) = label[1..] + "/", searched by prefix).
 * ------------------------------------------------------------------------------------------ */
static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

/* FieldKey::from_field_id (utils.rs:80-111): "<uuid>[/<type>/<name>[/...]]" -> 16 uuid bytes [+ type + "/" + name].
 * Returns the key length, or -1 when the id does not parse (invalid uuid; a type without a name). */
int orc_field_key(const uint8_t *id, size_t len, uint8_t *out, size_t cap) {
    size_t p = 0, n_hex = 0;
    uint8_t uuid[16];
    memset(uuid, 0, sizeof(uuid));
    while (p < len && id[p] != '/') {   /* uuid::Uuid::parse_str: 32 hex digits, simple or hyphenated */
        if (id[p] == '-') { p++; continue; }
        int h = hexval(id[p]);
        if (h < 0 || n_hex >= 32) return -1;
        uuid[n_hex / 2] = (uint8_t)(n_hex % 2 ? (uuid[n_hex / 2] | h) : (h << 4));
        n_hex++;
        p++;
    }
    if (n_hex != 32) return -1;
    if (cap < 16) return -1;
    memcpy(out, uuid, 16);
    if (p >= len) return 16;
    size_t t0 = ++p;
    while (p < len && id[p] != '/') p++;
    size_t t1 = p;
    if (p >= len) return -1;            /* has a field type but no name */
    size_t n0 = ++p;
    while (p < len && id[p] != '/') p++;
    size_t n1 = p;
    size_t need = 16 + (t1 - t0) + 1 + (n1 - n0);
    if (cap < need) return -1;
    memcpy(out + 16, id + t0, t1 - t0);
    out[16 + (t1 - t0)] = '/';
    memcpy(out + 16 + (t1 - t0) + 1, id + n0, n1 - n0);
    return (int)need;
}

/* strings: concatenated bytes + offsets.  para_keys: the paragraph ids ("<uuid>/<type>/<name>/<start>-<end>").
 * para_label_offsets[p] .. [p+1]: the paragraph's labels, indices into `labels`.
 * program: postfix ORC_FORMULA_* ops; LABEL a = atom string a of `atoms` (a label, "/l/x");
 * KEYSET a, b = atoms a..b (field ids: "<uuid_simple><field_id>" or "<uuid_simple>") — AtomClause::KeyPrefixSet;
 * AND / OR pop two; NOT complements the top (Clause::Compound with BooleanOperator::Not = complement of the AND
 * of its operands: emit the ANDs first).
 * resource_prefix != 0: a 16-byte key (resource-granular entry) matches every field of the resource — the intent
 * documented at searcher.rs:300-313; 0: `field_index.get` as written (:147-151), an exact match that a bare resource
 * key never satisfies.
 * out: bitset over paragraph addresses.  Returns the number of set bits, or -1 on a malformed program. */
enum { ORC_FORMULA_LABEL = 0, ORC_FORMULA_AND = 1, ORC_FORMULA_OR = 2, ORC_FORMULA_NOT = 3, ORC_FORMULA_ALL = 4, ORC_FORMULA_NONE = 5,
       ORC_FORMULA_KEYSET = 6 };
long orc_formula_filter(const uint8_t *key_bytes, const uint64_t *key_offsets, size_t n_paragraphs,
                        const uint8_t *label_bytes, const uint64_t *label_offsets, const uint64_t *para_label_offsets, const uint32_t *para_labels,
                        const uint8_t *atom_bytes, const uint64_t *atom_offsets, const orc_filter_op *ops, size_t n_ops, int resource_prefix,
                        uint64_t *out) {
    size_t words = (n_paragraphs + 63) / 64;
    memset(out, 0, words * 8);
    long count = 0;
    uint8_t pk[512], ak[512];
    int stack[64];
    for (size_t p = 0; p < n_paragraphs; p++) {
        int pklen = orc_field_key(key_bytes + key_offsets[p], (size_t)(key_offsets[p + 1] - key_offsets[p]), pk, sizeof(pk));
        int sp = 0;
        for (size_t i = 0; i < n_ops; i++) {
            const orc_filter_op *op = &ops[i];
            switch (op->op) {
                case ORC_FORMULA_LABEL: {
                    /* label_index.get_prefix(labels_key(atom)): some label l of the paragraph with
                     * (l[1..] + "/") starting with (atom[1..] + "/") */
                    const uint8_t *a = atom_bytes + atom_offsets[op->a];
                    size_t alen = (size_t)(atom_offsets[op->a + 1] - atom_offsets[op->a]);
                    int hit = 0;
                    for (uint64_t j = para_label_offsets[p]; j < para_label_offsets[p + 1] && !hit; j++) {
                        const uint8_t *l = label_bytes + label_offsets[para_labels[j]];
                        size_t llen = (size_t)(label_offsets[para_labels[j] + 1] - label_offsets[para_labels[j]]);
                        if (alen == 0 || llen == 0) continue;
                        /* compare l[1..] + "/" against the prefix a[1..] + "/" */
                        size_t need = alen - 1;
                        if (llen - 1 < need) continue;
                        if (memcmp(l + 1, a + 1, need) != 0) continue;
                        if (llen - 1 == need || l[1 + need] == '/') hit = 1;
                    }
                    if (sp >= 64) return -1;
                    stack[sp++] = hit;
                    break;
                }
                case ORC_FORMULA_KEYSET: {
                    int hit = 0;
                    for (uint32_t a = op->a; a < op->b && !hit; a++) {
                        int aklen = orc_field_key(atom_bytes + atom_offsets[a], (size_t)(atom_offsets[a + 1] - atom_offsets[a]), ak, sizeof(ak));
                        if (aklen < 0 || pklen < 0) continue;   /* filter_map: ids that do not parse are skipped */
                        if (aklen == pklen && memcmp(ak, pk, (size_t)aklen) == 0) hit = 1;
                        else if (resource_prefix && aklen == 16 && pklen >= 16 && memcmp(ak, pk, 16) == 0) hit = 1;
                    }
                    if (sp >= 64) return -1;
                    stack[sp++] = hit;
                    break;
                }
                case ORC_FORMULA_ALL: if (sp >= 64) return -1; stack[sp++] = 1; break;
                case ORC_FORMULA_NONE: if (sp >= 64) return -1; stack[sp++] = 0; break;
                case ORC_FORMULA_AND: if (sp < 2) return -1; stack[sp - 2] = stack[sp - 2] && stack[sp - 1]; sp--; break;
                case ORC_FORMULA_OR: if (sp < 2) return -1; stack[sp - 2] = stack[sp - 2] || stack[sp - 1]; sp--; break;
                case ORC_FORMULA_NOT: if (sp < 1) return -1; stack[sp - 1] = !stack[sp - 1]; break;
                default: return -1;
            }
        }
        if (sp != 1) return -1;
        if (stack[0]) { out[p >> 6] |= 1ull << (p & 63); count++; }
    }
    return count;
}
