/* Host-side ingestion fast path (CPython extension, no numpy headers needed).
 *
 * grakel_amd/batch.py flattens grakel's input objects into a packed CSR batch with Python +
 * numpy; for the common input form -- every element is `[edge_dict, node_labels, ...]` with
 * `edge_dict` a dict of neighbour lists `{u: [v, ...]}` or a dict of dicts `{u: {v: w}}` -- this
 * module does the same walk in C (what the reference does with one `Graph` object per element,
 * grakel/graph.py:147-232,1613-1705 and weisfeiler_lehman.py:142-194).  It is an ACCELERATOR OF
 * HOST LOGIC ONLY: whenever an element is of any other form, or something looks unusual (a
 * neighbour that is not an exact int under identity numbering, an unhashable symbol, ...), the
 * function returns None and batch.py's Python path -- the behavioural reference, including every
 * error the reference raises -- handles the whole input.  tests/test_host.py checks that both
 * paths produce identical batches.
 *
 *   wl_ingest(X: list, min_len: int, want_mask=False, max_len=0, n_threads=0) -> None | (sizes, row_ptr, col_idx, values[, mask])
 *       n_threads: 0 = as many as the host has (at most 32; 64 for the tuple-set form), 1 = the calling thread only
 *       sizes   bytearray of int32[n_graphs]     nodes per graph (= labelled vertices)
 *       row_ptr bytearray of int32[V + 1]
 *       col_idx bytearray of int32[E]            GLOBAL node ids, ascending and unique per row
 *       values  list[V] | bytearray of int64[V]  the label objects in node order (packed when every one is an exact int64)
 *   Node index = position of the vertex in the label dictionary (weisfeiler_lehman.py:234).
 *   A neighbour without a label raises KeyError like the reference (weisfeiler_lehman.py:238).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t* p;
    size_t n, cap;
} vec32;

static int vec_push(vec32* v, int32_t x) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 4096;
        int32_t* q = (int32_t*)realloc(v->p, nc * sizeof(int32_t));
        if (!q) return -1;
        v->p = q, v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

typedef struct { int64_t* p; size_t n, cap; } vec64;
static int v64_push(vec64* v, int64_t x) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        int64_t* q = (int64_t*)realloc(v->p, nc * sizeof(int64_t));
        if (!q) return -1;
        v->p = q, v->cap = nc;
    }
    v->p[v->n++] = x;
    return 0;
}

/* a growing array that IS the bytearray handed back to Python: no copy of the 20 MB of column indices at the end */
typedef struct { PyObject* ba; size_t n, cap; } bvec;            /* n, cap in bytes */
static int bvec_reserve(bvec* v, size_t need) {
    if (need <= v->cap) return 0;
    size_t nc = v->cap ? v->cap * 2 : 65536;
    while (nc < need) nc *= 2;
    if (!v->ba) {
        v->ba = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)nc);
        if (!v->ba) return -1;
    } else if (PyByteArray_Resize(v->ba, (Py_ssize_t)nc)) return -1;
    v->cap = nc;
    return 0;
}
#define BVEC_P(v, T) ((T*)PyByteArray_AS_STRING((v).ba))
static PyObject* bvec_finish(bvec* v) {          /* trims to the used size; the caller owns the reference */
    if (!v->ba && bvec_reserve(v, 1)) return NULL;
    if (PyByteArray_Resize(v->ba, (Py_ssize_t)v->n)) return NULL;
    PyObject* r = v->ba;
    v->ba = NULL;
    return r;
}

static int cmp32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* sort row[0..m) ascending and drop duplicates; returns the new length */
static size_t sort_unique(int32_t* row, size_t m) {
    if (m < 2) return m;
    if (m <= 24) {
        for (size_t i = 1; i < m; ++i) {
            int32_t key = row[i];
            size_t j = i;
            while (j > 0 && row[j - 1] > key) { row[j] = row[j - 1]; --j; }
            row[j] = key;
        }
    } else {
        qsort(row, m, sizeof(int32_t), cmp32);
    }
    size_t w = 1;
    for (size_t i = 1; i < m; ++i)
        if (row[i] != row[w - 1]) row[w++] = row[i];
    return w;
}

/* raise KeyError(key) -- PyErr_SetObject alone would unpack a tuple key into the exception's args */
static void set_key_error(PyObject* key) {
    PyObject* t = PyTuple_Pack(1, key);
    if (!t) return;
    PyErr_SetObject(PyExc_KeyError, t);
    Py_DECREF(t);
}

enum { ST_OK = 0, ST_DECLINE = 1, ST_ERROR = 2 };

/* GK_SMALL_INT(o, out): 1 and the value when o is an exact int of at most one 30-bit digit and not negative -- every
 * vertex number and almost every label; no call, three loads from the object's first cache line.  CPython 3.8 .. 3.11
 * keep the sign in ob_size and the digits in ob_digit; later versions never take this path (the callers fall through to
 * the API calls). */
#if PY_VERSION_HEX < 0x030C0000
#define GK_SMALL_INT(o, out) \
    (Py_TYPE(o) == &PyLong_Type && (size_t)Py_SIZE(o) <= 1u && ((out) = Py_SIZE(o) ? (long long)((PyLongObject*)(o))->ob_digit[0] : 0, 1))
#else
#define GK_SMALL_INT(o, out) 0
#endif

/* integer value of an exact int, or of anything with __index__ (numpy integer scalars, bool): such
 * objects hash and compare like the int, so dictionary semantics are unchanged.  ST_DECLINE otherwise. */
static int int_value(PyObject* o, long long* out) {
    int overflow = 0;
#if PY_VERSION_HEX < 0x030C0000
    /* one-digit exact ints (every vertex number below 2^30) without a call: CPython 3.8 .. 3.11 keep the sign in
     * ob_size and 30-bit digits in ob_digit; later versions take the API path below */
    if (PyLong_CheckExact(o)) {
        const Py_ssize_t sz = Py_SIZE(o);
        if (sz == 1) { *out = (long long)((PyLongObject*)o)->ob_digit[0]; return ST_OK; }
        if (sz == 0) { *out = 0; return ST_OK; }
    }
#endif
    if (PyLong_CheckExact(o)) {
        *out = PyLong_AsLongLongAndOverflow(o, &overflow);
        return overflow ? ST_DECLINE : ST_OK;
    }
    if (!PyIndex_Check(o) || PyFloat_Check(o)) return ST_DECLINE;
    PyObject* i = PyNumber_Index(o);
    if (!i) { PyErr_Clear(); return ST_DECLINE; }
    *out = PyLong_AsLongLongAndOverflow(i, &overflow);
    Py_DECREF(i);
    return overflow ? ST_DECLINE : ST_OK;
}

/* index of neighbour `nb` among the n labelled vertices of the current graph */
static int neighbour_index(PyObject* nb, int identity, Py_ssize_t n, PyObject* pos, Py_ssize_t* out) {
    if (identity) {
        long long j;
        if (int_value(nb, &j) != ST_OK) return ST_DECLINE;
        if (j < 0 || j >= (long long)n) {
            set_key_error(nb);          /* unlabelled neighbour */
            return ST_ERROR;
        }
        *out = (Py_ssize_t)j;
        return ST_OK;
    }
    PyObject* idx = PyDict_GetItemWithError(pos, nb);     /* borrowed */
    if (!idx) {
        if (PyErr_Occurred()) { PyErr_Clear(); return ST_DECLINE; }   /* unhashable symbol: let Python decide */
        set_key_error(nb);
        return ST_ERROR;
    }
    *out = PyLong_AsSsize_t(idx);
    return ST_OK;
}

/* ------------------------------------------------------------------------------------------
 * The walk on several host threads, for the ONE most common form only: every element `[g, labels, ...]` with `labels`
 * keyed 0, 1, ..., n-1 in this order, every label value a small exact int, `g` keyed 0 .. n-1 in the same order with
 * exact lists of in-range small exact ints as values.  The calling thread HOLDS THE GIL for the whole call and only
 * waits, so no Python code runs anywhere and no object can change; the workers only READ objects (type pointer, size,
 * first digit, list item pointers, PyDict_Next -- none of which touches a reference count, allocates or sets an error).
 * Anything else at all makes a worker give up, and wl_ingest then walks the whole input on the calling thread as
 * before: the threaded walk never decides behaviour, it only produces the same arrays sooner (tests/test_host.py
 * compares the two).  Ten thousand graphs of 100 vertices: 70 ms on one thread.
 * ------------------------------------------------------------------------------------------ */
#if PY_VERSION_HEX < 0x030C0000
#include <pthread.h>
#include "cpu_budget.h"
#include <stdio.h>
#include <unistd.h>
#define GK_PAR_MIN_ELEMENTS 256
#define GK_PAR_MAX_THREADS 64          /* an explicit n_threads may go this far */
#define GK_PAR_DEFAULT_THREADS 32      /* n_threads = 0: one per host core, at most this many -- measured on the 256-thread box
                                        * (tools/dev/ingest_scaling.py, 10 000 graphs of 100 vertices): dict of lists 40 / 7.5 / 4.1 / 3.1 / 3.8 ms
                                        * on 1 / 8 / 16 / 32 / 64 threads, adjacency matrices 113 / 18 / 11 / 7.8 / 8.8 ms; the tuple-set walk
                                        * is memory-latency bound (a tuple and two int objects per edge) and keeps scaling: 440 / 93 / 47 /
                                        * 25 / 15 ms -- it takes up to 64 */
/* runnable threads this process may have at once: cpu_budget.h, shared with gram.hip's widening threads */
static int cpu_budget(void) { return gk_cpu_budget(); }

typedef struct {
    PyObject* X;
    Py_ssize_t e0, e1, min_len, max_len;
    int ok;
    int32_t* sizes;                 /* [e1 - e0] */
    int32_t* deg; size_t n_deg, cap_deg;       /* out-degree per vertex */
    int32_t* col; size_t n_col, cap_col;       /* neighbour indices LOCAL to their graph */
    int64_t* lab;                   /* label values, same length as deg */
    /* second phase */
    int64_t v_base, e_base;
    int32_t *out_rowp, *out_col;
    int64_t* out_lab;
    const void* aux;                /* MATRIX form: the acquired buffer views */
    int sp_mode;                    /* the batch must also be ShortestPath's (sp_batch_from_input): vertices = ALL vertices of the graph in
                                     * sorted symbol order = the label keys in their order, unit weights */
} par_job;

static int par_grow(par_job* j, size_t need_deg, size_t need_col) {
    if (need_deg > j->cap_deg) {
        size_t nc = j->cap_deg ? j->cap_deg * 2 : 16384;
        while (nc < need_deg) nc *= 2;
        int32_t* d = (int32_t*)realloc(j->deg, nc * 4);
        if (!d) return -1;
        j->deg = d;
        int64_t* l = (int64_t*)realloc(j->lab, nc * 8);
        if (!l) return -1;
        j->lab = l, j->cap_deg = nc;
    }
    if (need_col > j->cap_col) {
        size_t nc = j->cap_col ? j->cap_col * 2 : 65536;
        while (nc < need_col) nc *= 2;
        int32_t* c = (int32_t*)realloc(j->col, nc * 4);
        if (!c) return -1;
        j->col = c, j->cap_col = nc;
    }
    return 0;
}

static void* par_walk(void* arg) {
    par_job* j = (par_job*)arg;
    j->ok = 0;
    for (Py_ssize_t e = j->e0; e < j->e1; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(j->X, e);
        if (!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) return NULL;
        const Py_ssize_t xl = PySequence_Fast_GET_SIZE(x);
        if (xl < j->min_len || (j->max_len > 0 && xl > j->max_len)) return NULL;
        PyObject* g = PySequence_Fast_GET_ITEM(x, 0);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        if (!PyDict_CheckExact(g) || !PyDict_CheckExact(labels)) return NULL;
        const Py_ssize_t n = PyDict_GET_SIZE(labels);
        if (n == 0 || n > 0x3fffffff || PyDict_GET_SIZE(g) != n) return NULL;
        if (par_grow(j, j->n_deg + (size_t)n, j->n_col)) return NULL;
        Py_ssize_t it = 0, itg = 0, i = 0;
        PyObject *k, *lv, *gk, *d;
        while (PyDict_Next(labels, &it, &k, &lv)) {
            long long kv, iv;
            if (!GK_SMALL_INT(k, kv) || kv != (long long)i || !GK_SMALL_INT(lv, iv)) return NULL;
            if (!PyDict_Next(g, &itg, &gk, &d) || !GK_SMALL_INT(gk, kv) || kv != (long long)i || !PyList_CheckExact(d)) return NULL;
            const Py_ssize_t len = PyList_GET_SIZE(d);
            if (par_grow(j, j->n_deg, j->n_col + (size_t)len)) return NULL;
            int32_t* const row = j->col + j->n_col;
            PyObject* const* const items = ((PyListObject*)d)->ob_item;
            int ascending = 1;
            int32_t prev = -1;
            size_t m = 0;
            for (Py_ssize_t q = 0; q < len; ++q) {
                long long c;
                if (!GK_SMALL_INT(items[q], c) || c >= (long long)n) return NULL;
                ascending &= (int32_t)c > prev;
                prev = (int32_t)c;
                row[m++] = (int32_t)c;
            }
            if (!ascending) m = sort_unique(row, m);
            j->n_col += m;
            j->deg[j->n_deg + (size_t)i] = (int32_t)m;
            j->lab[j->n_deg + (size_t)i] = (int64_t)iv;
            ++i;
        }
        if (i != n) return NULL;
        j->n_deg += (size_t)n;
        j->sizes[e - j->e0] = (int32_t)n;
    }
    j->ok = 1;
    return NULL;
}

static void* par_place(void* arg) {
    par_job* j = (par_job*)arg;
    int64_t run = j->e_base;
    int32_t* rp = j->out_rowp + j->v_base;          /* rp[v + 1] = end of row v */
    for (size_t v = 0; v < j->n_deg; ++v) {
        run += j->deg[v];
        rp[v + 1] = (int32_t)run;
    }
    memcpy(j->out_lab + j->v_base, j->lab, j->n_deg * 8);
    int32_t* oc = j->out_col + j->e_base;
    const int32_t* c = j->col;
    int64_t vg = j->v_base;
    size_t v = 0;
    for (Py_ssize_t e = 0; e < j->e1 - j->e0; ++e) {
        const int32_t n = j->sizes[e];
        size_t cnt = 0;
        for (int32_t i = 0; i < n; ++i) cnt += (size_t)j->deg[v + (size_t)i];
        for (size_t q = 0; q < cnt; ++q) oc[q] = c[q] + (int32_t)vg;
        oc += cnt, c += cnt, v += (size_t)n, vg += n;
    }
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * Round 5: two more input forms on the same threaded machinery (par_job / par_place), because they are what real data
 * arrives as:
 *   PAIRS   `[edges, labels, ...]` with `edges` a set / frozenset / list / tuple of `(u, v)` tuples or a dict keyed by such
 *           tuples (numeric values), `labels` a dict keyed by the vertex ids -- the form `grakel.datasets.fetch_dataset` /
 *           `read_data` produce (datasets/base.py:273-279: global 1-based ids), which batch.py's Python path walked at
 *           ~0.3 ms per graph (2.8 s per 10 000 graphs of 100 vertices);
 *   MATRIX  `[A, labels, ...]` with `A` a C-contiguous 2-D buffer (numpy.ndarray) and `labels` keyed 0 .. n-1 in order
 *           (np.nonzero(A > 0) per graph in Python: 0.07 ms per graph).
 * Same rules as above: vertex ids and labels small exact ints, the workers only read; anything else -- an unlabelled
 * neighbour (the reference's KeyError), a 3-tuple, a float label -- makes the walk give up and the established paths
 * decide.  Semantics (batch.py: _edge_lists / _wl_graph_arrays): nodes = the labelled vertices in label order, an edge
 * whose SOURCE has no label is ignored, duplicates collapse, rows ascending.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t* key; int32_t* val; uint32_t* stamp; size_t cap; uint32_t gen;       /* open addressing, cleared by generation */
    int32_t *src, *dst; size_t n_e, cap_e;                                       /* the graph's (source, target) indices */
    int32_t *cnt, *tmp; size_t cap_n, cap_t;
} pair_scratch;

static int ps_table(pair_scratch* s, size_t n) {
    size_t need = 64;
    while (need < 2 * n + 2) need <<= 1;
    if (need > s->cap) {
        free(s->key); free(s->val); free(s->stamp);
        s->key = (int64_t*)malloc(need * 8), s->val = (int32_t*)malloc(need * 4), s->stamp = (uint32_t*)calloc(need, 4);
        s->cap = (s->key && s->val && s->stamp) ? need : 0;
        s->gen = 0;
        if (!s->cap) return -1;
    }
    if (++s->gen == 0) { memset(s->stamp, 0, s->cap * 4); s->gen = 1; }
    return 0;
}
static inline size_t ps_hash(int64_t k, size_t mask) { return (size_t)(((uint64_t)k * 0x9E3779B97F4A7C15ULL) >> 20) & mask; }
static inline int ps_insert(pair_scratch* s, size_t mask, int64_t k, int32_t v) {      /* 1: the key was there already */
    size_t h = ps_hash(k, mask);
    while (s->stamp[h] == s->gen) {
        if (s->key[h] == k) return 1;
        h = (h + 1) & mask;
    }
    s->stamp[h] = s->gen, s->key[h] = k, s->val[h] = v;
    return 0;
}
static inline int32_t ps_find(const pair_scratch* s, size_t mask, int64_t k) {
    size_t h = ps_hash(k, mask);
    while (s->stamp[h] == s->gen) {
        if (s->key[h] == k) return s->val[h];
        h = (h + 1) & mask;
    }
    return -1;
}
static int ps_edge(pair_scratch* s, int32_t a, int32_t b) {
    if (s->n_e == s->cap_e) {
        size_t nc = s->cap_e ? s->cap_e * 2 : 4096;
        int32_t* p = (int32_t*)realloc(s->src, nc * 4);
        if (!p) return -1;
        s->src = p;
        p = (int32_t*)realloc(s->dst, nc * 4);
        if (!p) return -1;
        s->dst = p, s->cap_e = nc;
    }
    s->src[s->n_e] = a, s->dst[s->n_e] = b, ++s->n_e;
    return 0;
}
/* the graph's collected (source, target) pairs -> rows of the job: counting sort by source, rows sorted and de-duplicated */
static int ps_rows(par_job* j, pair_scratch* s, Py_ssize_t n) {
    if ((size_t)n + 1 > s->cap_n) {
        free(s->cnt);
        s->cnt = (int32_t*)malloc(((size_t)n + 1) * 4);
        s->cap_n = s->cnt ? (size_t)n + 1 : 0;
        if (!s->cnt) return -1;
    }
    if (s->n_e > s->cap_t) {
        free(s->tmp);
        s->tmp = (int32_t*)malloc(s->n_e * 4);
        s->cap_t = s->tmp ? s->n_e : 0;
        if (!s->tmp) return -1;
    }
    memset(s->cnt, 0, ((size_t)n + 1) * 4);
    for (size_t q = 0; q < s->n_e; ++q) ++s->cnt[s->src[q] + 1];
    for (Py_ssize_t i = 0; i < n; ++i) s->cnt[i + 1] += s->cnt[i];
    {   /* placement needs a moving cursor per row: reuse deg[] of the job as the cursor (it is overwritten below) */
        int32_t* cur = j->deg + j->n_deg;
        for (Py_ssize_t i = 0; i < n; ++i) cur[i] = s->cnt[i];
        for (size_t q = 0; q < s->n_e; ++q) s->tmp[cur[s->src[q]]++] = s->dst[q];
    }
    if (par_grow(j, j->n_deg + (size_t)n, j->n_col + s->n_e)) return -1;
    for (Py_ssize_t i = 0; i < n; ++i) {
        int32_t* row = s->tmp + s->cnt[i];
        const size_t m = sort_unique(row, (size_t)(s->cnt[i + 1] - s->cnt[i]));
        memcpy(j->col + j->n_col, row, m * 4);
        j->n_col += m;
        j->deg[j->n_deg + (size_t)i] = (int32_t)m;
    }
    return 0;
}
static void ps_free(pair_scratch* s) {
    free(s->key); free(s->val); free(s->stamp); free(s->src); free(s->dst); free(s->cnt); free(s->tmp);
}

static void* par_walk_pairs(void* arg) {
    par_job* j = (par_job*)arg;
    j->ok = 0;
    pair_scratch s;
    memset(&s, 0, sizeof s);
    for (Py_ssize_t e = j->e0; e < j->e1; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(j->X, e);
        if (!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) goto out;
        const Py_ssize_t xl = PySequence_Fast_GET_SIZE(x);
        if (xl < j->min_len || (j->max_len > 0 && xl > j->max_len)) goto out;
        PyObject* g = PySequence_Fast_GET_ITEM(x, 0);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        if (!PyDict_CheckExact(labels)) goto out;
        const int g_set = PySet_CheckExact(g) || PyFrozenSet_CheckExact(g), g_dict = PyDict_CheckExact(g);
        const int g_seq = PyList_CheckExact(g) || PyTuple_CheckExact(g);
        if (!g_set && !g_dict && !g_seq) goto out;
        const Py_ssize_t n = PyDict_GET_SIZE(labels);
        const Py_ssize_t ne = g_set ? PySet_GET_SIZE(g) : (g_dict ? PyDict_GET_SIZE(g) : PySequence_Fast_GET_SIZE(g));
        if (n == 0 || n > 0x3fffffff || ne == 0) goto out;
        if (par_grow(j, j->n_deg + (size_t)n, j->n_col) || ps_table(&s, (size_t)n)) goto out;
        const size_t mask = s.cap - 1;
        {
            Py_ssize_t it = 0, i = 0;
            PyObject *k, *lv;
            long long prev_key = -1;
            while (PyDict_Next(labels, &it, &k, &lv)) {
                long long kv, iv;
                if (!GK_SMALL_INT(k, kv) || !GK_SMALL_INT(lv, iv)) goto out;
                if (j->sp_mode && kv <= prev_key) goto out;            /* ShortestPath numbers the vertices by sorted symbol */
                prev_key = kv;
                if (ps_insert(&s, mask, (int64_t)kv, (int32_t)i)) goto out;
                j->lab[j->n_deg + (size_t)i] = (int64_t)iv;
                ++i;
            }
            if (i != n) goto out;
        }
        s.n_e = 0;
        {
            Py_ssize_t it = 0, q = 0;
            PyObject *t = NULL, *w = NULL;
            Py_hash_t hsh;
            for (;;) {
                if (g_set) { if (!_PySet_NextEntry(g, &it, &t, &hsh)) break; }
                else if (g_dict) {
                    if (!PyDict_Next(g, &it, &t, &w)) break;
                    if (!PyFloat_CheckExact(w) && !PyLong_CheckExact(w)) goto out;          /* {(u, v): weight}: numbers only */
                    if (j->sp_mode && !(PyFloat_CheckExact(w) ? PyFloat_AS_DOUBLE(w) == 1.0 : (Py_SIZE(w) == 1 && ((PyLongObject*)w)->ob_digit[0] == 1)))
                        goto out;                                                           /* weighted: sp_ingest / the Python path */
                } else { if (q >= ne) break; t = PySequence_Fast_GET_ITEM(g, q++); }
                long long a, b;
                if (!PyTuple_CheckExact(t) || PyTuple_GET_SIZE(t) != 2) goto out;
                if (!GK_SMALL_INT(PyTuple_GET_ITEM(t, 0), a) || !GK_SMALL_INT(PyTuple_GET_ITEM(t, 1), b)) goto out;
                const int32_t ia = ps_find(&s, mask, (int64_t)a);
                if (ia < 0) {
                    if (j->sp_mode) goto out;                          /* a vertex without a label: ShortestPath's KeyError */
                    continue;                                          /* the source has no label: never visited (batch.py) */
                }
                const int32_t ib = ps_find(&s, mask, (int64_t)b);
                if (ib < 0) goto out;                                  /* unlabelled neighbour: the reference's KeyError */
                if (ps_edge(&s, ia, ib)) goto out;
            }
        }
        if (ps_rows(j, &s, n)) goto out;
        if (j->sp_mode) {          /* every labelled vertex must BE a vertex of the graph (an endpoint of some edge) */
            int32_t* touched = s.cnt;                                  /* n + 1 words, free again after ps_rows */
            memset(touched, 0, (size_t)n * 4);
            for (size_t q = 0; q < s.n_e; ++q) touched[s.src[q]] = 1, touched[s.dst[q]] = 1;
            for (Py_ssize_t i = 0; i < n; ++i)
                if (!touched[i]) goto out;
        }
        j->n_deg += (size_t)n;
        j->sizes[e - j->e0] = (int32_t)n;
    }
    j->ok = 1;
out:
    ps_free(&s);
    return NULL;
}

/* MATRIX form: the buffers were acquired by the calling thread (PyObject_GetBuffer touches reference counts) */
typedef struct { const char* p; Py_ssize_t n, itemsize; char kind; } mat_view;     /* kind: 'i' signed, 'u' unsigned, 'f' float, 'b' bool */
static inline double mat_value(const mat_view* m, const char* q) {
    switch (m->kind) {
    case 'f': return m->itemsize == 8 ? *(const double*)q : (double)*(const float*)q;
    case 'i': return m->itemsize == 8 ? (double)*(const int64_t*)q : m->itemsize == 4 ? (double)*(const int32_t*)q
                   : m->itemsize == 2 ? (double)*(const int16_t*)q : (double)*(const int8_t*)q;
    default:  return m->itemsize == 8 ? (double)*(const uint64_t*)q : m->itemsize == 4 ? (double)*(const uint32_t*)q
                   : m->itemsize == 2 ? (double)*(const uint16_t*)q : (double)*(const uint8_t*)q;
    }
}
static inline int mat_positive(const mat_view* m, const char* q) {
    switch (m->kind) {
    case 'f': return m->itemsize == 8 ? *(const double*)q > 0.0 : *(const float*)q > 0.0f;
    case 'i': return m->itemsize == 8 ? *(const int64_t*)q > 0 : m->itemsize == 4 ? *(const int32_t*)q > 0
                   : m->itemsize == 2 ? *(const int16_t*)q > 0 : *(const int8_t*)q > 0;
    default:  return m->itemsize == 8 ? *(const uint64_t*)q != 0 : m->itemsize == 4 ? *(const uint32_t*)q != 0
                   : m->itemsize == 2 ? *(const uint16_t*)q != 0 : *(const uint8_t*)q != 0;
    }
}
static void* par_walk_matrix(void* arg) {
    par_job* j = (par_job*)arg;
    j->ok = 0;
    const mat_view* views = (const mat_view*)j->aux;
    for (Py_ssize_t e = j->e0; e < j->e1; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(j->X, e);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        const mat_view* m = &views[e];
        const Py_ssize_t n = m->n;
        if (!PyDict_CheckExact(labels) || PyDict_GET_SIZE(labels) != n || n == 0) return NULL;
        if (par_grow(j, j->n_deg + (size_t)n, j->n_col)) return NULL;
        Py_ssize_t it = 0, i = 0;
        PyObject *k, *lv;
        while (PyDict_Next(labels, &it, &k, &lv)) {
            long long kv, iv;
            if (!GK_SMALL_INT(k, kv) || kv != (long long)i || !GK_SMALL_INT(lv, iv)) return NULL;
            j->lab[j->n_deg + (size_t)i] = (int64_t)iv;
            ++i;
        }
        for (Py_ssize_t r = 0; r < n; ++r) {
            if (par_grow(j, j->n_deg, j->n_col + (size_t)n)) return NULL;
            int32_t* row = j->col + j->n_col;
            const char* q = m->p + (size_t)r * (size_t)n * (size_t)m->itemsize;
            size_t cnt = 0;
            if (j->sp_mode) {                      /* ShortestPath: any non-zero entry is an edge WITH that weight: units only here */
                for (Py_ssize_t c = 0; c < n; ++c, q += m->itemsize) {
                    const double v = mat_value(m, q);
                    if (v == 1.0) row[cnt++] = (int32_t)c;
                    else if (v != 0.0) return NULL;
                }
            } else
                for (Py_ssize_t c = 0; c < n; ++c, q += m->itemsize)
                    if (mat_positive(m, q)) row[cnt++] = (int32_t)c;
            j->n_col += cnt;
            j->deg[j->n_deg + (size_t)r] = (int32_t)cnt;
        }
        j->n_deg += (size_t)n;
        j->sizes[e - j->e0] = (int32_t)n;
    }
    j->ok = 1;
    return NULL;
}

/* SPARSE form (round 6): `[A, labels, ...]` with `A` a scipy.sparse matrix (graph.py:1564-1580 takes those as adjacency
 * matrices).  The calling thread takes CSR matrices (other formats keep the Python path), checks that each is square and in
 * canonical form (no duplicate entries: the reference densifies, which would ADD them) and acquires the three buffers; the
 * threads then walk indptr / indices / data.  An explicit zero is no edge; ShortestPath (sp_mode) wants unit weights here. */
typedef struct { const char *indptr, *indices; Py_ssize_t ip_size, ix_size; mat_view data; Py_ssize_t n; } csr_view;
static inline int64_t idx_value(const char* p, Py_ssize_t itemsize, Py_ssize_t k) {
    return itemsize == 8 ? ((const int64_t*)p)[k] : (int64_t)((const int32_t*)p)[k];
}
static void* par_walk_csr(void* arg) {
    par_job* j = (par_job*)arg;
    j->ok = 0;
    const csr_view* views = (const csr_view*)j->aux;
    for (Py_ssize_t e = j->e0; e < j->e1; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(j->X, e);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        const csr_view* m = &views[e];
        const Py_ssize_t n = m->n;
        if (!PyDict_CheckExact(labels) || PyDict_GET_SIZE(labels) != n || n == 0) return NULL;
        if (par_grow(j, j->n_deg + (size_t)n, j->n_col)) return NULL;
        Py_ssize_t it = 0, i = 0;
        PyObject *k, *lv;
        while (PyDict_Next(labels, &it, &k, &lv)) {
            long long kv, iv;
            if (!GK_SMALL_INT(k, kv) || kv != (long long)i || !GK_SMALL_INT(lv, iv)) return NULL;
            j->lab[j->n_deg + (size_t)i] = (int64_t)iv;
            ++i;
        }
        for (Py_ssize_t r = 0; r < n; ++r) {
            const int64_t p0 = idx_value(m->indptr, m->ip_size, r), p1 = idx_value(m->indptr, m->ip_size, r + 1);
            if (p0 < 0 || p1 < p0 || p1 > (int64_t)m->data.n) return NULL;
            if (par_grow(j, j->n_deg, j->n_col + (size_t)(p1 - p0))) return NULL;
            int32_t* row = j->col + j->n_col;
            size_t cnt = 0;
            for (int64_t q = p0; q < p1; ++q) {
                const int64_t c = idx_value(m->indices, m->ix_size, (Py_ssize_t)q);
                if (c < 0 || c >= (int64_t)n) return NULL;
                const char* dq = m->data.p + (size_t)q * (size_t)m->data.itemsize;
                if (j->sp_mode) {
                    const double v = mat_value(&m->data, dq);
                    if (v == 1.0) row[cnt++] = (int32_t)c;
                    else if (v != 0.0) return NULL;
                } else if (mat_positive(&m->data, dq)) row[cnt++] = (int32_t)c;
            }
            j->n_col += cnt;
            j->deg[j->n_deg + (size_t)r] = (int32_t)cnt;
        }
        j->n_deg += (size_t)n;
        j->sizes[e - j->e0] = (int32_t)n;
    }
    j->ok = 1;
    return NULL;
}
/* the item type of a 1-D buffer as mat_view understands it: 0 = not a plain numeric type */
static char buf_kind(const Py_buffer* b) {
    const char* f = b->format ? b->format : "B";
    if (*f == '@' || *f == '=' || *f == '<') ++f;
    if (!f[0] || f[1]) return 0;
    if (strchr("bhilq", f[0])) return 'i';
    if (strchr("BHILQ?", f[0])) return 'u';
    if (strchr("fd", f[0])) return 'f';
    return 0;
}

/* NULL without an error set: not taken (the caller walks the input itself) */
/* form: 0 = dict of neighbour lists under identity numbering (par_walk), 1 = PAIRS, 2 = MATRIX (aux = the buffer views),
 * 3 = SPARSE (aux = the csr views) */
static PyObject* wl_ingest_threads(PyObject* X, Py_ssize_t min_len, Py_ssize_t max_len, int n_threads, int form, const void* aux, int sp_mode) {
    const Py_ssize_t n_el = PySequence_Fast_GET_SIZE(X);
    if (n_threads <= 0) {
        n_threads = cpu_budget();
        if (n_threads > (form == 1 ? GK_PAR_MAX_THREADS : GK_PAR_DEFAULT_THREADS)) n_threads = form == 1 ? GK_PAR_MAX_THREADS : GK_PAR_DEFAULT_THREADS;
    }
    if (n_threads > GK_PAR_MAX_THREADS) n_threads = GK_PAR_MAX_THREADS;
    if ((Py_ssize_t)n_threads > n_el / 64) n_threads = (int)(n_el / 64);
    if (n_threads < 2) {
        if (form == 0 && !sp_mode) return NULL;      /* the one-thread walk of wl_ingest knows this form (and more) */
        n_threads = 1;                               /* the new forms / ShortestPath: this walk on the calling thread */
    }
    void* (*walk)(void*) = form == 1 ? par_walk_pairs : (form == 2 ? par_walk_matrix : (form == 3 ? par_walk_csr : par_walk));
    par_job jobs[GK_PAR_MAX_THREADS];
    pthread_t tid[GK_PAR_MAX_THREADS];
    int started[GK_PAR_MAX_THREADS] = {0};
    memset(jobs, 0, sizeof jobs);
    int ok = 1;
    for (int t = 0; t < n_threads; ++t) {
        par_job* j = &jobs[t];
        j->X = X, j->min_len = min_len, j->max_len = max_len, j->aux = aux, j->sp_mode = sp_mode;
        j->e0 = n_el * t / n_threads, j->e1 = n_el * (t + 1) / n_threads;
        j->sizes = (int32_t*)malloc((size_t)(j->e1 - j->e0 + 1) * 4);
        if (!j->sizes) { ok = 0; break; }
    }
    if (ok) {
        for (int t = 1; t < n_threads; ++t) started[t] = pthread_create(&tid[t], NULL, walk, &jobs[t]) == 0;
        walk(&jobs[0]);
        for (int t = 1; t < n_threads; ++t) {
            if (started[t]) pthread_join(tid[t], NULL);
            else walk(&jobs[t]);
        }
        for (int t = 0; t < n_threads; ++t) ok = ok && jobs[t].ok;
    }
    PyObject *a = NULL, *b = NULL, *c = NULL, *l = NULL, *result = NULL;
    if (ok) {
        int64_t V = 0, E = 0;
        for (int t = 0; t < n_threads; ++t) {
            jobs[t].v_base = V, jobs[t].e_base = E;
            V += (int64_t)jobs[t].n_deg, E += (int64_t)jobs[t].n_col;
        }
        if (V < 2147483647LL && E < 2147483647LL) {
            a = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)(n_el * 4));
            b = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)((V + 1) * 4));
            c = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)(E * 4));
            l = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)(V * 8));
            if (a && b && c && l) {
                ((int32_t*)PyByteArray_AS_STRING(b))[0] = 0;
                for (int t = 0; t < n_threads; ++t) {
                    par_job* j = &jobs[t];
                    memcpy((int32_t*)PyByteArray_AS_STRING(a) + j->e0, j->sizes, (size_t)(j->e1 - j->e0) * 4);
                    j->out_rowp = (int32_t*)PyByteArray_AS_STRING(b), j->out_col = (int32_t*)PyByteArray_AS_STRING(c);
                    j->out_lab = (int64_t*)PyByteArray_AS_STRING(l);
                }
                for (int t = 1; t < n_threads; ++t) started[t] = pthread_create(&tid[t], NULL, par_place, &jobs[t]) == 0;
                par_place(&jobs[0]);
                for (int t = 1; t < n_threads; ++t) {
                    if (started[t]) pthread_join(tid[t], NULL);
                    else par_place(&jobs[t]);
                }
                result = PyTuple_Pack(4, a, b, c, l);
            }
            if (!result && PyErr_Occurred()) PyErr_Clear();      /* out of memory here: the one-thread walk reports it (or succeeds) */
        }
    }
    Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(l);
    for (int t = 0; t < n_threads; ++t) { free(jobs[t].sizes); free(jobs[t].deg); free(jobs[t].col); free(jobs[t].lab); }
    return result;
}
#endif

static PyObject* wl_ingest(PyObject* self, PyObject* args) {
    PyObject* X;
    Py_ssize_t min_len = 2, max_len = 0;
    int want_mask = 0, n_threads = 0, sp_mode = 0;
    if (!PyArg_ParseTuple(args, "O|npnii", &X, &min_len, &want_mask, &max_len, &n_threads, &sp_mode)) return NULL;
    if (!PyList_CheckExact(X) && !PyTuple_CheckExact(X)) Py_RETURN_NONE;
    const Py_ssize_t n_el = PySequence_Fast_GET_SIZE(X);
    if (n_el == 0) Py_RETURN_NONE;
    if (min_len < 2) min_len = 2;
#if PY_VERSION_HEX < 0x030C0000
    if (!want_mask) {
        /* which form does the FIRST element have?  (a mixed input fails the chosen walk and takes the established paths) */
        PyObject* x0 = PySequence_Fast_GET_ITEM(X, 0);
        PyObject* g0 = ((PyList_CheckExact(x0) || PyTuple_CheckExact(x0)) && PySequence_Fast_GET_SIZE(x0) >= 2) ? PySequence_Fast_GET_ITEM(x0, 0) : NULL;
        int form = -1;
        if (g0 && PyDict_CheckExact(g0)) {
            Py_ssize_t it = 0;
            PyObject *k0, *v0;
            if (PyDict_Next(g0, &it, &k0, &v0)) form = PyTuple_CheckExact(k0) ? 1 : (PyList_CheckExact(v0) ? 0 : -1);
        } else if (g0 && (PySet_CheckExact(g0) || PyFrozenSet_CheckExact(g0) || PyList_CheckExact(g0) || PyTuple_CheckExact(g0))) {
            /* a list of lists is an adjacency matrix (graph.py:1564-1580), a list of tuples an edge list */
            form = 1;
            if ((PyList_CheckExact(g0) || PyTuple_CheckExact(g0)) && PySequence_Fast_GET_SIZE(g0) > 0 &&
                !PyTuple_CheckExact(PySequence_Fast_GET_ITEM(g0, 0))) form = -1;
        } else if (g0 && PyObject_CheckBuffer(g0)) form = 2;
        else if (g0 && PyObject_HasAttrString(g0, "indptr") + PyObject_HasAttrString(g0, "tocsr") + PyObject_HasAttrString(g0, "nnz") >= 2 &&
                 PyObject_HasAttrString(g0, "tocsr")) form = 3;               /* a scipy.sparse matrix of any format */
        if (sp_mode && form < 0) Py_RETURN_NONE;
        if (form == 0 && (sp_mode || (n_threads != 1 && n_el >= GK_PAR_MIN_ELEMENTS))) {
            PyObject* r = wl_ingest_threads(X, min_len, max_len, n_threads, 0, NULL, sp_mode);
            if (r) return r;
            if (sp_mode) Py_RETURN_NONE;     /* not ShortestPath's vertex set / weights: sp_ingest or the Python path */
        } else if (form == 1) {
            PyObject* r = wl_ingest_threads(X, min_len, max_len, n_threads, 1, NULL, sp_mode);
            if (r) return r;
            Py_RETURN_NONE;                  /* not the plain form after all: batch.py's Python path */
        } else if (form == 2) {
            /* acquire every element's buffer on this thread: C-contiguous, square, a plain numeric item type */
            mat_view* views = (mat_view*)calloc((size_t)n_el, sizeof(mat_view));
            Py_buffer* bufs = (Py_buffer*)calloc((size_t)n_el, sizeof(Py_buffer));
            Py_ssize_t got = 0;
            int good = views && bufs;
            for (; good && got < n_el; ++got) {
                PyObject* x = PySequence_Fast_GET_ITEM(X, got);
                if ((!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) || PySequence_Fast_GET_SIZE(x) < min_len ||
                    (max_len > 0 && PySequence_Fast_GET_SIZE(x) > max_len)) { good = 0; break; }
                PyObject* A = PySequence_Fast_GET_ITEM(x, 0);
                if (!PyObject_CheckBuffer(A) || PyObject_GetBuffer(A, &bufs[got], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT)) {
                    if (PyErr_Occurred()) PyErr_Clear();
                    good = 0;
                    break;
                }
                const Py_buffer* b = &bufs[got];
                const char* f = b->format ? b->format : "B";
                if (*f == '@' || *f == '=' || *f == '<') ++f;
                char kind = 0;
                if (f[0] && !f[1]) {
                    if (strchr("bhilq", f[0])) kind = 'i';
                    else if (strchr("BHILQ?", f[0])) kind = 'u';
                    else if (strchr("fd", f[0])) kind = 'f';
                }
                if (b->ndim != 2 || b->shape[0] != b->shape[1] || !kind || (b->itemsize != 1 && b->itemsize != 2 && b->itemsize != 4 && b->itemsize != 8) ||
                    (kind == 'f' && b->itemsize < 4)) { ++got; good = 0; break; }
                views[got].p = (const char*)b->buf, views[got].n = b->shape[0], views[got].itemsize = b->itemsize, views[got].kind = kind;
            }
            PyObject* r = good ? wl_ingest_threads(X, min_len, max_len, n_threads, 2, views, sp_mode) : NULL;
            for (Py_ssize_t q = 0; q < got; ++q)
                if (bufs && bufs[q].obj) PyBuffer_Release(&bufs[q]);
            free(views); free(bufs);
            if (r) return r;
            Py_RETURN_NONE;
        } else if (form == 3) {
            /* every element's three arrays as buffers: all of it on this thread (attribute look-ups) */
            csr_view* views = (csr_view*)calloc((size_t)n_el, sizeof(csr_view));
            Py_buffer* bufs = (Py_buffer*)calloc((size_t)n_el * 3, sizeof(Py_buffer));
            PyObject** keep = (PyObject**)calloc((size_t)n_el * 4, sizeof(PyObject*));     /* csr object + its three arrays */
            int good = views && bufs && keep;
            for (Py_ssize_t e = 0; good && e < n_el; ++e) {
                PyObject* x = PySequence_Fast_GET_ITEM(X, e);
                if ((!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) || PySequence_Fast_GET_SIZE(x) < min_len ||
                    (max_len > 0 && PySequence_Fast_GET_SIZE(x) > max_len)) { good = 0; break; }
                PyObject* A = PySequence_Fast_GET_ITEM(x, 0);
                PyObject* fmt = PyObject_GetAttrString(A, "format");
                int is_csr = fmt && PyUnicode_Check(fmt) && PyUnicode_CompareWithASCIIString(fmt, "csr") == 0;
                Py_XDECREF(fmt);
                /* other formats: a conversion per graph costs more than the densifying Python path saves (measured: csc / coo / lil
                 * 117-180 ms per 3 000 graphs of 40 vertices through tocsr() against 71-160 ms) -- they keep that path */
                if (!is_csr) { good = 0; break; }
                PyObject* C = A;
                Py_INCREF(C);
                keep[4 * e] = C;
                PyObject* canon = PyObject_GetAttrString(C, "has_canonical_format");
                const int canonical = canon && PyObject_IsTrue(canon) == 1;
                Py_XDECREF(canon);
                PyObject* shape = PyObject_GetAttrString(C, "shape");
                Py_ssize_t n0 = -1, n1 = -2;
                if (shape && PyTuple_CheckExact(shape) && PyTuple_GET_SIZE(shape) == 2) {
                    n0 = PyLong_AsSsize_t(PyTuple_GET_ITEM(shape, 0)), n1 = PyLong_AsSsize_t(PyTuple_GET_ITEM(shape, 1));
                }
                Py_XDECREF(shape);
                if (!canonical || n0 != n1 || n0 <= 0) { good = 0; break; }
                static const char* names[3] = {"indptr", "indices", "data"};
                for (int a = 0; good && a < 3; ++a) {
                    PyObject* arr = PyObject_GetAttrString(C, names[a]);
                    keep[4 * e + 1 + a] = arr;
                    if (!arr || !PyObject_CheckBuffer(arr) || PyObject_GetBuffer(arr, &bufs[3 * e + a], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT)) good = 0;
                }
                if (!good) break;
                const Py_buffer *bp = &bufs[3 * e], *bi = &bufs[3 * e + 1], *bd = &bufs[3 * e + 2];
                const char kp = buf_kind(bp), ki = buf_kind(bi), kd = buf_kind(bd);
                if (kp != 'i' || ki != 'i' || !kd || (bp->itemsize != 4 && bp->itemsize != 8) || (bi->itemsize != 4 && bi->itemsize != 8) ||
                    bp->ndim != 1 || bi->ndim != 1 || bd->ndim != 1 || bp->shape[0] != n0 + 1 || bi->shape[0] != bd->shape[0] ||
                    (bd->itemsize != 1 && bd->itemsize != 2 && bd->itemsize != 4 && bd->itemsize != 8) || (kd == 'f' && bd->itemsize < 4)) { good = 0; break; }
                views[e].indptr = (const char*)bp->buf, views[e].ip_size = bp->itemsize;
                views[e].indices = (const char*)bi->buf, views[e].ix_size = bi->itemsize;
                views[e].data.p = (const char*)bd->buf, views[e].data.n = bd->shape[0], views[e].data.itemsize = bd->itemsize, views[e].data.kind = kd;
                views[e].n = n0;
            }
            if (PyErr_Occurred()) PyErr_Clear();
            PyObject* r = good ? wl_ingest_threads(X, min_len, max_len, n_threads, 3, views, sp_mode) : NULL;
            if (bufs)
                for (Py_ssize_t q = 0; q < 3 * n_el; ++q)
                    if (bufs[q].obj) PyBuffer_Release(&bufs[q]);
            if (keep)
                for (Py_ssize_t q = 0; q < 4 * n_el; ++q) Py_XDECREF(keep[q]);
            free(views); free(bufs); free(keep);
            if (r) return r;
            Py_RETURN_NONE;
        }
    }
#endif

    vec32 sizes = {0};
    bvec rowp = {0}, col = {0};        /* int32 */
    bvec ivals = {0};                  /* int64: the label values while every one of them is an exact int64 */
    int all_int = 1;
    /* want_mask (WL-OA, weisfeiler_lehman_optimal_assignment.py:176): per labelled vertex, does it own an
     * entry in the reference's edge dictionary?  dict of lists: a key with a non-empty list, or a vertex
     * that only occurs as a neighbour; dict of dicts: any key or neighbour (batch.py: _edge_lists).
     * flag bits per vertex: 1 = key of g, 2 = key with out-edges, 4 = somebody's neighbour */
    unsigned char* flag = NULL;
    size_t flag_cap = 0;
    PyObject* values = PyList_New(0);
    PyObject* pos = NULL;
    int status = ST_OK;
    int64_t V = 0;
    if (!values || bvec_reserve(&rowp, 4) || bvec_reserve(&col, 4)) { status = ST_ERROR; goto done; }
    BVEC_P(rowp, int32_t)[0] = 0, rowp.n = 4;

    for (Py_ssize_t e = 0; e < n_el && status == ST_OK; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(X, e);
        if (!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) { status = ST_DECLINE; break; }
        if (PySequence_Fast_GET_SIZE(x) < min_len) { status = ST_DECLINE; break; }
        if (max_len > 0 && PySequence_Fast_GET_SIZE(x) > max_len) { status = ST_DECLINE; break; }
        PyObject* g = PySequence_Fast_GET_ITEM(x, 0);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        if (!PyDict_CheckExact(g) || !PyDict_CheckExact(labels)) { status = ST_DECLINE; break; }
        const Py_ssize_t n = PyDict_GET_SIZE(labels);
        if (n == 0 || PyDict_GET_SIZE(g) == 0) { status = ST_DECLINE; break; }
        if (V + n >= 2147483647LL) { status = ST_DECLINE; break; }

        /* form of the edge dictionary: all values lists, or all values dicts (graph.py:1640-1690).  The row pass below
         * sees the type of every value it uses; when it has walked the WHOLE edge dictionary in step with the labels (the
         * usual case) that is the form check, otherwise a separate walk over g follows it */
        int all_list = 1, all_dict = 1, seen_list = 0, seen_dict = 0;
        Py_ssize_t g_nonempty = 0;                 /* keys of g with out-edges */
        if (want_mask) {
            if ((size_t)(V + n) > flag_cap) {
                size_t nc = flag_cap ? flag_cap * 2 : 65536;
                while (nc < (size_t)(V + n)) nc *= 2;
                unsigned char* q = (unsigned char*)realloc(flag, nc);
                if (!q) { status = ST_ERROR; PyErr_NoMemory(); break; }
                flag = q, flag_cap = nc;
            }
            memset(flag + V, 0, (size_t)n);
        }
        Py_ssize_t lab_in_g = 0, lab_nonempty = 0;  /* the same counts over the LABELLED keys */

        /* identity numbering: the label keys are exactly 0, 1, ..., n-1 in this order */
        int identity = 1;
        {
            Py_ssize_t it = 0, i = 0;
            PyObject *k, *lv;
            while (PyDict_Next(labels, &it, &k, &lv)) {
                long long kv;
                if (identity && !(GK_SMALL_INT(k, kv) ? kv == (long long)i : (PyLong_CheckExact(k) && PyLong_AsSsize_t(k) == i))) {
                    identity = 0;
                    if (PyErr_Occurred()) PyErr_Clear();
                }
                if (all_int) {                 /* exact ints that fit int64: the caller gets them as an array */
                    int ovf = 0;
                    long long iv = 0;
                    const int small = GK_SMALL_INT(lv, iv);
                    if (!small) iv = PyLong_CheckExact(lv) ? PyLong_AsLongLongAndOverflow(lv, &ovf) : 0;
                    if (!small && (!PyLong_CheckExact(lv) || ovf)) {
                        /* first label that is not an int64: from here on the values travel as a list -- catch up on
                         * the graphs walked so far (their dictionaries have not changed: we hold the GIL) */
                        all_int = 0;
                        for (Py_ssize_t e2 = 0; e2 < e && status == ST_OK; ++e2) {
                            PyObject* l2 = PySequence_Fast_GET_ITEM(PySequence_Fast_GET_ITEM(X, e2), 1);
                            Py_ssize_t it2 = 0;
                            PyObject *k2, *v2;
                            while (PyDict_Next(l2, &it2, &k2, &v2))
                                if (PyList_Append(values, v2)) { status = ST_ERROR; break; }
                        }
                        Py_ssize_t it2 = 0, seen = 0;
                        PyObject *k2, *v2;
                        while (status == ST_OK && seen < i && PyDict_Next(labels, &it2, &k2, &v2)) {
                            if (PyList_Append(values, v2)) status = ST_ERROR;
                            ++seen;
                        }
                        if (status != ST_OK) break;
                    } else {
                        if (bvec_reserve(&ivals, ivals.n + 8)) { status = ST_ERROR; break; }
                        *(int64_t*)(PyByteArray_AS_STRING(ivals.ba) + ivals.n) = (int64_t)iv, ivals.n += 8;
                    }
                }
                if (!all_int && PyList_Append(values, lv)) { status = ST_ERROR; break; }
                ++i;
            }
            if (status != ST_OK) break;
        }
        if (!identity) {
            Py_XDECREF(pos);
            pos = PyDict_New();
            if (!pos) { status = ST_ERROR; break; }
            Py_ssize_t it = 0, i = 0;
            PyObject *k, *lv;
            while (PyDict_Next(labels, &it, &k, &lv)) {
                PyObject* idx = PyLong_FromSsize_t(i++);
                if (!idx || PyDict_SetItem(pos, k, idx)) { Py_XDECREF(idx); status = ST_ERROR; break; }
                Py_DECREF(idx);
            }
            if (status != ST_OK) break;
            if (PyDict_GET_SIZE(pos) != n) { status = ST_DECLINE; break; }    /* keys that compare equal: Python path */
        }

        /* rows in label order.  The edge dictionary usually lists its keys in the order of the label dictionary (both
         * were filled vertex by vertex): walk it in step, no hash look-up per vertex; under identity numbering the label
         * dictionary need not be walked again at all (vertex i's key is the integer i).  Rows are written straight into
         * the column array (their length is known), sorted there if they turn out not to be strictly ascending. */
        Py_ssize_t it = 0, itg = 0;
        int lockstep = 1;
        if (bvec_reserve(&rowp, rowp.n + 4 * (size_t)n)) { status = ST_ERROR; break; }
        for (Py_ssize_t vi = 0; vi < n && status == ST_OK; ++vi) {
            PyObject *k = NULL, *lv, *k_owned = NULL;
            if (!identity && !PyDict_Next(labels, &it, &k, &lv)) { status = ST_DECLINE; break; }
            PyObject* d = NULL;
            if (lockstep) {
                PyObject *gk, *gd;
                Py_ssize_t save = itg;
                long long gv;
                if (PyDict_Next(g, &itg, &gk, &gd) &&
                    (identity ? (PyLong_CheckExact(gk) && int_value(gk, &gv) == ST_OK && gv == (long long)vi)
                              : (gk == k || (PyLong_CheckExact(gk) && PyLong_CheckExact(k) && PyObject_RichCompareBool(gk, k, Py_EQ) == 1))))
                    d = gd;
                else { lockstep = 0; itg = save; }
            }
            if (!d) {
                if (identity) {
                    k = k_owned = PyLong_FromSsize_t(vi);
                    if (!k) { status = ST_ERROR; break; }
                }
                d = PyDict_GetItemWithError(g, k);          /* borrowed; absent: no out-edges */
                Py_XDECREF(k_owned);
                if (!d && PyErr_Occurred()) { PyErr_Clear(); status = ST_DECLINE; break; }
            }
            const size_t vrow = rowp.n / 4 - 1;              /* this vertex */
            size_t m = 0;
            if (d) {
                const int is_list = PyList_CheckExact(d);
                if (is_list) seen_list = 1;
                else if (PyDict_CheckExact(d)) seen_dict = 1;
                else { status = ST_DECLINE; break; }
                const Py_ssize_t len = is_list ? PyList_GET_SIZE(d) : PyDict_GET_SIZE(d);
                g_nonempty += len > 0;         /* complete only if every entry of g is visited: checked below */
                ++lab_in_g;
                lab_nonempty += len > 0;
                if (want_mask) flag[vrow] |= (unsigned char)(1 | (len > 0 ? 2 : 0));
                if (bvec_reserve(&col, col.n + 4 * (size_t)len)) { status = ST_ERROR; break; }
                int32_t* const row = (int32_t*)(PyByteArray_AS_STRING(col.ba) + col.n);
                int ascending = 1;           /* strictly ascending neighbour lists (the usual case) need no sort */
                int32_t prev = -1;
                if (is_list) {
                    Py_ssize_t q = 0;
                    if (identity) {              /* the common form without a call per neighbour; anything else resumes below */
                        PyObject* const* const items = ((PyListObject*)d)->ob_item;
                        for (; q < len; ++q) {
                            long long j;
                            if (!GK_SMALL_INT(items[q], j) || j >= (long long)n) break;
                            const int32_t c = (int32_t)(V + j);
                            ascending &= c > prev;
                            prev = c;
                            row[m++] = c;
                        }
                    }
                    for (; q < len; ++q) {
                        Py_ssize_t j;
                        status = neighbour_index(PyList_GET_ITEM(d, q), identity, n, pos, &j);
                        if (status != ST_OK) break;
                        const int32_t c = (int32_t)(V + j);
                        ascending &= c > prev;
                        prev = c;
                        row[m++] = c;
                    }
                } else {
                    Py_ssize_t it2 = 0;
                    PyObject *nb, *w;
                    while (PyDict_Next(d, &it2, &nb, &w)) {
                        if (!PyFloat_CheckExact(w) && !PyLong_CheckExact(w)) { status = ST_DECLINE; break; }   /* weights: numbers only */
                        Py_ssize_t j;
                        status = neighbour_index(nb, identity, n, pos, &j);
                        if (status != ST_OK) break;
                        const int32_t c = (int32_t)(V + j);
                        ascending &= c > prev;
                        prev = c;
                        row[m++] = c;
                    }
                }
                if (status != ST_OK) break;
                if (!ascending) m = sort_unique(row, m);
                if (want_mask)
                    for (size_t q = 0; q < m; ++q) flag[row[q]] |= 4;
            }
            col.n += 4 * m;
            if (col.n / 4 >= 2147483647ULL) { status = ST_DECLINE; break; }
            *(int32_t*)(PyByteArray_AS_STRING(rowp.ba) + rowp.n) = (int32_t)(col.n / 4), rowp.n += 4;
        }
        if (status != ST_OK) {
            /* nothing about this element's form has been established yet: whatever is wrong with it, the Python path
             * (the behavioural reference) reports it */
            if (PyErr_Occurred()) PyErr_Clear();
            status = ST_DECLINE;
            break;
        }
        if (lockstep && PyDict_GET_SIZE(g) == n) {           /* every entry of g was visited, once */
            all_list = !seen_dict, all_dict = !seen_list;
        } else {
            g_nonempty = 0;
            Py_ssize_t it2 = 0;
            PyObject *k2, *d2;
            while (PyDict_Next(g, &it2, &k2, &d2)) {
                if (!PyList_CheckExact(d2)) all_list = 0; else if (PyList_GET_SIZE(d2) > 0) ++g_nonempty;
                if (!PyDict_CheckExact(d2)) all_dict = 0; else if (PyDict_GET_SIZE(d2) > 0) ++g_nonempty;
                if (!all_list && !all_dict) break;
            }
        }
        if (!all_list && !all_dict) { status = ST_DECLINE; break; }
        if (want_mask) {
            /* an entry vertex without a label is the reference's KeyError: which one it names depends on
             * set order, so let the Python path raise it */
            if (all_list ? lab_nonempty != g_nonempty : lab_in_g != PyDict_GET_SIZE(g)) { status = ST_DECLINE; break; }
            for (Py_ssize_t i = 0; i < n; ++i) {
                const unsigned char f = flag[V + i];
                flag[V + i] = all_list ? ((f & 2) || (!(f & 1) && (f & 4))) : ((f & 1) || (f & 4));
            }
        }
        if (vec_push(&sizes, (int32_t)n)) { status = ST_ERROR; PyErr_NoMemory(); break; }
        V += n;
    }

done:;
    PyObject* result = NULL;
    if (status == ST_OK) {
        PyObject* a = PyByteArray_FromStringAndSize((const char*)sizes.p, (Py_ssize_t)(sizes.n * 4));
        PyObject* b = bvec_finish(&rowp);
        PyObject* c = bvec_finish(&col);
        /* labels: a bytearray of int64 when all are exact ints (no million-element list -> array conversion), else the list */
        PyObject* vals = values;
        PyObject* packed = NULL;
        if (all_int && ivals.n == 8 * (size_t)V) {
            /* the value LIST was not filled while all_int held: without the packed form there are no labels to hand
             * back, so a failed allocation is the MemoryError it is (never an empty label list for V vertices) */
            packed = bvec_finish(&ivals);
            vals = packed;
        }
        if (!vals) {
            if (!PyErr_Occurred()) PyErr_NoMemory();
        } else if (a && b && c && want_mask) {
            PyObject* mk = PyByteArray_FromStringAndSize((const char*)flag, (Py_ssize_t)V);
            if (mk) result = PyTuple_Pack(5, a, b, c, vals, mk);
            Py_XDECREF(mk);
        } else if (a && b && c) result = PyTuple_Pack(4, a, b, c, vals);
        Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(packed);
    } else if (status == ST_DECLINE) {
        if (PyErr_Occurred()) PyErr_Clear();
        result = Py_None;
        Py_INCREF(result);
    }
    Py_XDECREF(values);
    Py_XDECREF(pos);
    free(sizes.p); free(flag);
    Py_XDECREF(rowp.ba); Py_XDECREF(col.ba); Py_XDECREF(ivals.ba);
    return result;
}

/* ------------------------------------------------------------------------------------------
 * ShortestPath ingestion (batch.py: sp_batch_from_input / _sp_graph_arrays).  Nodes are ALL vertices
 * of the graph: 0..n-1 for an adjacency matrix (graph.py:912-981), the sorted vertex symbols for an
 * edge dictionary (graph.py:894-907) -- handled here only when every symbol is an exact int.
 *   sp_ingest(X, with_labels, min_len, max_len) -> None | (sizes, row_ptr, col_idx, weight, values)
 * Weights must be positive integers below 2^20 (what the device path supports); anything else is
 * declined so that the Python path raises what it raises.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int64_t key; int32_t w; } edge_t;      /* key = src * n + dst (local indices) */

static int cmp_edge(const void* a, const void* b) {
    int64_t x = ((const edge_t*)a)->key, y = ((const edge_t*)b)->key;
    return (x > y) - (x < y);
}
static int cmp64(const void* a, const void* b) {
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}
static Py_ssize_t find64(const int64_t* a, Py_ssize_t n, int64_t x) {
    Py_ssize_t lo = 0, hi = n;
    while (lo < hi) { Py_ssize_t m = (lo + hi) >> 1; if (a[m] < x) lo = m + 1; else hi = m; }
    return lo;      /* present by construction */
}

/* exact int -> int64, or decline */
static int as_i64(PyObject* o, int64_t* out) {
    long long v;
    if (int_value(o, &v) != ST_OK) return ST_DECLINE;
    *out = (int64_t)v;
    return ST_OK;
}

/* a weight the device path takes: positive integer < 2^20 (int, or float with an integer value) */
static int as_weight(PyObject* o, int32_t* out) {
    double d;
    if (PyLong_CheckExact(o)) {
        int overflow = 0;
        long long v = PyLong_AsLongLongAndOverflow(o, &overflow);
        if (overflow || v <= 0 || v >= (1 << 20)) return ST_DECLINE;
        *out = (int32_t)v;
        return ST_OK;
    }
    if (!PyFloat_CheckExact(o)) return ST_DECLINE;
    d = PyFloat_AS_DOUBLE(o);
    if (!(d > 0.0) || d >= 1048576.0 || d != (double)(int32_t)d) return ST_DECLINE;
    *out = (int32_t)d;
    return ST_OK;
}

typedef struct { edge_t* p; size_t n, cap; } vecE;
static int vE_push(vecE* v, int64_t key, int32_t w) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        edge_t* q = (edge_t*)realloc(v->p, nc * sizeof(edge_t));
        if (!q) return -1;
        v->p = q, v->cap = nc;
    }
    v->p[v->n].key = key, v->p[v->n].w = w;
    ++v->n;
    return 0;
}

static PyObject* sp_ingest(PyObject* self, PyObject* args) {
    PyObject* X;
    int with_labels = 1;
    Py_ssize_t min_len = 1, max_len = 3;
    if (!PyArg_ParseTuple(args, "Op|nn", &X, &with_labels, &min_len, &max_len)) return NULL;
    if (!PyList_CheckExact(X) && !PyTuple_CheckExact(X)) Py_RETURN_NONE;
    const Py_ssize_t n_el = PySequence_Fast_GET_SIZE(X);
    if (n_el == 0) Py_RETURN_NONE;

    vec32 sizes = {0}, rowp = {0}, col = {0}, wts = {0};
    vec64 verts = {0};
    vecE edges = {0};
    PyObject* values = PyList_New(0);
    int status = ST_OK;
    int64_t V = 0;
    if (!values || vec_push(&rowp, 0)) { status = ST_ERROR; PyErr_NoMemory(); goto done; }

    for (Py_ssize_t e = 0; e < n_el && status == ST_OK; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(X, e);
        if (!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) { status = ST_DECLINE; break; }
        const Py_ssize_t xl = PySequence_Fast_GET_SIZE(x);
        if (xl < min_len || xl < 1 || (max_len > 0 && xl > max_len)) { status = ST_DECLINE; break; }
        PyObject* g = PySequence_Fast_GET_ITEM(x, 0);
        PyObject* labels = xl > 1 ? PySequence_Fast_GET_ITEM(x, 1) : NULL;
        if (with_labels && !(labels && PyDict_CheckExact(labels) && PyDict_GET_SIZE(labels) > 0)) { status = ST_DECLINE; break; }
        Py_ssize_t n = 0;
        edges.n = 0, verts.n = 0;
        int is_matrix = 0;

        if (PyDict_CheckExact(g)) {
            if (PyDict_GET_SIZE(g) == 0) { status = ST_DECLINE; break; }
            int all_list = 1, all_dict = 1;
            Py_ssize_t it = 0;
            PyObject *k, *d;
            while (PyDict_Next(g, &it, &k, &d)) {
                if (!PyList_CheckExact(d)) all_list = 0;
                if (!PyDict_CheckExact(d)) all_dict = 0;
                if (!all_list && !all_dict) break;
            }
            if (!all_list && !all_dict) { status = ST_DECLINE; break; }
            /* pass 1: vertex symbols */
            it = 0;
            while (PyDict_Next(g, &it, &k, &d) && status == ST_OK) {
                int64_t a;
                if ((status = as_i64(k, &a)) != ST_OK) break;
                if (v64_push(&verts, a)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                if (all_list) {
                    for (Py_ssize_t q = 0; q < PyList_GET_SIZE(d); ++q) {
                        int64_t bnb;
                        if ((status = as_i64(PyList_GET_ITEM(d, q), &bnb)) != ST_OK) break;
                        if (v64_push(&verts, bnb)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                    }
                } else {
                    Py_ssize_t it2 = 0;
                    PyObject *nb, *w;
                    while (PyDict_Next(d, &it2, &nb, &w)) {
                        int64_t bnb;
                        if ((status = as_i64(nb, &bnb)) != ST_OK) break;
                        if (v64_push(&verts, bnb)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                    }
                }
            }
            if (status != ST_OK) break;
            qsort(verts.p, verts.n, sizeof(int64_t), cmp64);
            {
                size_t w = 0;
                for (size_t i = 0; i < verts.n; ++i)
                    if (w == 0 || verts.p[i] != verts.p[w - 1]) verts.p[w++] = verts.p[i];
                verts.n = w;
            }
            n = (Py_ssize_t)verts.n;
            /* pass 2: edges with local indices */
            it = 0;
            while (PyDict_Next(g, &it, &k, &d) && status == ST_OK) {
                int64_t a = 0;
                as_i64(k, &a);
                const int64_t ia = find64(verts.p, n, a);
                if (all_list) {
                    for (Py_ssize_t q = 0; q < PyList_GET_SIZE(d); ++q) {
                        int64_t bnb = 0;
                        as_i64(PyList_GET_ITEM(d, q), &bnb);
                        if (vE_push(&edges, ia * n + find64(verts.p, n, bnb), 1)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                    }
                } else {
                    Py_ssize_t it2 = 0;
                    PyObject *nb, *w;
                    while (PyDict_Next(d, &it2, &nb, &w)) {
                        int64_t bnb = 0;
                        int32_t wi;
                        as_i64(nb, &bnb);
                        if ((status = as_weight(w, &wi)) != ST_OK) break;
                        if (vE_push(&edges, ia * n + find64(verts.p, n, bnb), wi)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                    }
                }
            }
            if (status != ST_OK) break;
            qsort(edges.p, edges.n, sizeof(edge_t), cmp_edge);
            {   /* duplicates only arise from repeated list entries (all weight 1): keep one */
                size_t w = 0;
                for (size_t i = 0; i < edges.n; ++i)
                    if (w == 0 || edges.p[i].key != edges.p[w - 1].key) edges.p[w++] = edges.p[i];
                edges.n = w;
            }
        } else {
            /* adjacency matrix through the buffer protocol: 2-D, C-contiguous, a plain numeric type */
            Py_buffer view;
            if (!PyObject_CheckBuffer(g) || PyDict_Check(g) || PyBytes_Check(g) || PyByteArray_Check(g)) { status = ST_DECLINE; break; }
            if (PyObject_GetBuffer(g, &view, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS | PyBUF_ND) != 0) { PyErr_Clear(); status = ST_DECLINE; break; }
            const char* f = view.format ? view.format : "B";
            if (*f == '@' || *f == '=' || *f == '<') ++f;
            const char code = f[0];
            const int known = f[1] == 0 && ((code == 'l' && view.itemsize == 8) || (code == 'q' && view.itemsize == 8) ||
                                            (code == 'i' && view.itemsize == 4) || (code == 'd' && view.itemsize == 8) ||
                                            (code == 'B' && view.itemsize == 1) || (code == '?' && view.itemsize == 1));
            if (view.ndim != 2 || view.shape[0] != view.shape[1] || !known || view.shape[0] == 0) {
                PyBuffer_Release(&view);
                status = ST_DECLINE;
                break;
            }
            n = view.shape[0];
            is_matrix = 1;
            const char* base = (const char*)view.buf;
            for (Py_ssize_t i = 0; i < n && status == ST_OK; ++i)
                for (Py_ssize_t jx = 0; jx < n; ++jx) {
                    const char* p = base + (i * n + jx) * view.itemsize;
                    double d;
                    if (code == 'l' || code == 'q') d = (double)*(const int64_t*)p;
                    else if (code == 'i') d = (double)*(const int32_t*)p;
                    else if (code == 'd') d = *(const double*)p;
                    else d = (double)*(const unsigned char*)p;
                    if (d == 0.0) continue;
                    if (!(d > 0.0) || d >= 1048576.0 || d != (double)(int32_t)d) { status = ST_DECLINE; break; }
                    if (vE_push(&edges, (int64_t)i * n + jx, (int32_t)d)) { status = ST_ERROR; PyErr_NoMemory(); break; }
                }
            PyBuffer_Release(&view);
            if (status != ST_OK) break;
        }
        if (V + n >= 2147483647LL || col.n + edges.n >= 2147483647ULL) { status = ST_DECLINE; break; }

        /* labels in vertex order: labels[v] for v in verts (KeyError like the reference) */
        if (with_labels) {
            for (Py_ssize_t i = 0; i < n; ++i) {
                PyObject* key = PyLong_FromLongLong(is_matrix ? (long long)i : (long long)verts.p[i]);
                if (!key) { status = ST_ERROR; break; }
                PyObject* lv = PyDict_GetItemWithError(labels, key);
                if (!lv) {
                    if (!PyErr_Occurred()) set_key_error(key);
                    Py_DECREF(key);
                    status = ST_ERROR;
                    break;
                }
                Py_DECREF(key);
                if (PyList_Append(values, lv)) { status = ST_ERROR; break; }
            }
            if (status != ST_OK) break;
        }
        /* CSR rows of this graph */
        {
            size_t q = 0;
            for (Py_ssize_t i = 0; i < n && status == ST_OK; ++i) {
                while (q < edges.n && edges.p[q].key / n == i) {
                    if (vec_push(&col, (int32_t)(V + edges.p[q].key % n)) || vec_push(&wts, edges.p[q].w)) {
                        status = ST_ERROR; PyErr_NoMemory(); break;
                    }
                    ++q;
                }
                if (status == ST_OK && vec_push(&rowp, (int32_t)col.n)) { status = ST_ERROR; PyErr_NoMemory(); }
            }
        }
        if (status != ST_OK) break;
        if (vec_push(&sizes, (int32_t)n)) { status = ST_ERROR; PyErr_NoMemory(); break; }
        V += n;
    }

done:;
    PyObject* result = NULL;
    if (status == ST_OK) {
        PyObject* a = PyByteArray_FromStringAndSize((const char*)sizes.p, (Py_ssize_t)(sizes.n * 4));
        PyObject* b = PyByteArray_FromStringAndSize((const char*)rowp.p, (Py_ssize_t)(rowp.n * 4));
        PyObject* c = PyByteArray_FromStringAndSize((const char*)col.p, (Py_ssize_t)(col.n * 4));
        PyObject* d = PyByteArray_FromStringAndSize((const char*)wts.p, (Py_ssize_t)(wts.n * 4));
        if (a && b && c && d) result = PyTuple_Pack(5, a, b, c, d, values);
        Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c); Py_XDECREF(d);
    } else if (status == ST_DECLINE) {
        if (PyErr_Occurred()) PyErr_Clear();
        result = Py_None;
        Py_INCREF(result);
    }
    Py_XDECREF(values);
    free(sizes.p); free(rowp.p); free(col.p); free(wts.p); free(verts.p); free(edges.p);
    return result;
}

static PyMethodDef methods[] = {
    {"sp_ingest", sp_ingest, METH_VARARGS,
     "sp_ingest(X, with_labels, min_len=1, max_len=3) -> None | (sizes, row_ptr, col_idx, weight, values)"},
    {"wl_ingest", wl_ingest, METH_VARARGS,
     "wl_ingest(X, min_len=2, want_mask=False, max_len=0) -> None | (sizes, row_ptr, col_idx, values[, mask])"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gk_ingest", "C fast path of grakel_amd.batch", -1, methods};

PyMODINIT_FUNC PyInit__gk_ingest(void) { return PyModule_Create(&moddef); }
