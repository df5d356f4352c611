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
 *   wl_ingest(X: list, min_len: int) -> None | (sizes, row_ptr, col_idx, values)
 *       sizes   bytearray of int32[n_graphs]     nodes per graph (= labelled vertices)
 *       row_ptr bytearray of int32[V + 1]
 *       col_idx bytearray of int32[E]            GLOBAL node ids, ascending and unique per row
 *       values  list[V]                          the label objects in node order
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

enum { ST_OK = 0, ST_DECLINE = 1, ST_ERROR = 2 };

/* index of neighbour `nb` among the n labelled vertices of the current graph */
static int neighbour_index(PyObject* nb, int identity, Py_ssize_t n, PyObject* pos, Py_ssize_t* out) {
    if (identity) {
        if (!PyLong_CheckExact(nb)) return ST_DECLINE;
        int overflow = 0;
        long long j = PyLong_AsLongLongAndOverflow(nb, &overflow);
        if (overflow || j < 0 || j >= (long long)n) {
            if (!overflow && j == -1 && PyErr_Occurred()) return ST_ERROR;
            PyErr_SetObject(PyExc_KeyError, nb);          /* unlabelled neighbour */
            return ST_ERROR;
        }
        *out = (Py_ssize_t)j;
        return ST_OK;
    }
    PyObject* idx = PyDict_GetItemWithError(pos, nb);     /* borrowed */
    if (!idx) {
        if (PyErr_Occurred()) { PyErr_Clear(); return ST_DECLINE; }   /* unhashable symbol: let Python decide */
        PyErr_SetObject(PyExc_KeyError, nb);
        return ST_ERROR;
    }
    *out = PyLong_AsSsize_t(idx);
    return ST_OK;
}

static PyObject* wl_ingest(PyObject* self, PyObject* args) {
    PyObject* X;
    Py_ssize_t min_len = 2;
    if (!PyArg_ParseTuple(args, "O|n", &X, &min_len)) return NULL;
    if (!PyList_CheckExact(X) && !PyTuple_CheckExact(X)) Py_RETURN_NONE;
    const Py_ssize_t n_el = PySequence_Fast_GET_SIZE(X);
    if (n_el == 0) Py_RETURN_NONE;
    if (min_len < 2) min_len = 2;

    vec32 sizes = {0}, rowp = {0}, col = {0}, tmp = {0};
    PyObject* values = PyList_New(0);
    PyObject* pos = NULL;
    int status = ST_OK;
    int64_t V = 0;
    if (!values || vec_push(&rowp, 0)) { status = ST_ERROR; PyErr_NoMemory(); goto done; }

    for (Py_ssize_t e = 0; e < n_el && status == ST_OK; ++e) {
        PyObject* x = PySequence_Fast_GET_ITEM(X, e);
        if (!PyList_CheckExact(x) && !PyTuple_CheckExact(x)) { status = ST_DECLINE; break; }
        if (PySequence_Fast_GET_SIZE(x) < min_len) { status = ST_DECLINE; break; }
        PyObject* g = PySequence_Fast_GET_ITEM(x, 0);
        PyObject* labels = PySequence_Fast_GET_ITEM(x, 1);
        if (!PyDict_CheckExact(g) || !PyDict_CheckExact(labels)) { status = ST_DECLINE; break; }
        const Py_ssize_t n = PyDict_GET_SIZE(labels);
        if (n == 0 || PyDict_GET_SIZE(g) == 0) { status = ST_DECLINE; break; }
        if (V + n >= 2147483647LL) { status = ST_DECLINE; break; }

        /* form of the edge dictionary: all values lists, or all values dicts (graph.py:1640-1690) */
        int all_list = 1, all_dict = 1;
        {
            Py_ssize_t it = 0;
            PyObject *k, *d;
            while (PyDict_Next(g, &it, &k, &d)) {
                if (!PyList_CheckExact(d)) all_list = 0;
                if (!PyDict_CheckExact(d)) all_dict = 0;
                if (!all_list && !all_dict) break;
            }
        }
        if (!all_list && !all_dict) { status = ST_DECLINE; break; }

        /* identity numbering: the label keys are exactly 0, 1, ..., n-1 in this order */
        int identity = 1;
        {
            Py_ssize_t it = 0, i = 0;
            PyObject *k, *lv;
            while (PyDict_Next(labels, &it, &k, &lv)) {
                if (identity && !(PyLong_CheckExact(k) && PyLong_AsSsize_t(k) == i)) {
                    identity = 0;
                    if (PyErr_Occurred()) PyErr_Clear();
                }
                if (PyList_Append(values, lv)) { status = ST_ERROR; break; }
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

        /* rows in label order */
        Py_ssize_t it = 0;
        PyObject *k, *lv;
        while (PyDict_Next(labels, &it, &k, &lv) && status == ST_OK) {
            PyObject* d = PyDict_GetItemWithError(g, k);          /* borrowed; absent: no out-edges */
            if (!d && PyErr_Occurred()) { PyErr_Clear(); status = ST_DECLINE; break; }
            tmp.n = 0;
            if (d && all_list) {
                const Py_ssize_t m = PyList_GET_SIZE(d);
                for (Py_ssize_t q = 0; q < m; ++q) {
                    Py_ssize_t j;
                    status = neighbour_index(PyList_GET_ITEM(d, q), identity, n, pos, &j);
                    if (status != ST_OK) break;
                    if (vec_push(&tmp, (int32_t)(V + j))) { status = ST_ERROR; PyErr_NoMemory(); break; }
                }
            } else if (d) {
                Py_ssize_t it2 = 0;
                PyObject *nb, *w;
                while (PyDict_Next(d, &it2, &nb, &w)) {
                    if (!PyFloat_CheckExact(w) && !PyLong_CheckExact(w)) { status = ST_DECLINE; break; }   /* weights: numbers only */
                    Py_ssize_t j;
                    status = neighbour_index(nb, identity, n, pos, &j);
                    if (status != ST_OK) break;
                    if (vec_push(&tmp, (int32_t)(V + j))) { status = ST_ERROR; PyErr_NoMemory(); break; }
                }
            }
            if (status != ST_OK) break;
            const size_t m = sort_unique(tmp.p, tmp.n);
            for (size_t q = 0; q < m; ++q)
                if (vec_push(&col, tmp.p[q])) { status = ST_ERROR; PyErr_NoMemory(); break; }
            if (col.n >= 2147483647ULL) { status = ST_DECLINE; break; }
            if (status == ST_OK && vec_push(&rowp, (int32_t)col.n)) { status = ST_ERROR; PyErr_NoMemory(); }
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
        if (a && b && c) result = PyTuple_Pack(4, a, b, c, values);
        Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c);
    } else if (status == ST_DECLINE) {
        if (PyErr_Occurred()) PyErr_Clear();
        result = Py_None;
        Py_INCREF(result);
    }
    Py_XDECREF(values);
    Py_XDECREF(pos);
    free(sizes.p); free(rowp.p); free(col.p); free(tmp.p);
    return result;
}

static PyMethodDef methods[] = {
    {"wl_ingest", wl_ingest, METH_VARARGS,
     "wl_ingest(X, min_len=2) -> None | (sizes, row_ptr, col_idx, values): see grakel_amd/csrc/ingest.c"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_gk_ingest", "C fast path of grakel_amd.batch", -1, methods};

PyMODINIT_FUNC PyInit__gk_ingest(void) { return PyModule_Create(&moddef); }
