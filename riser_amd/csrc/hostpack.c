/* _hostpack: the per-read host work of a ReadUntil batch as three C loops (CPython extension, no GPU code).
 *
 * The batched control loop (riser_amd/control.py) receives a ReadUntil batch as a Python list of read objects whose
 * `raw_data` attribute holds the int16 ADC samples as bytes (riser/client.py:46-47: np.frombuffer(read.raw_data, dtype)).
 * Walking 18 000 of them in Python - one get_raw_signal call, one slice, one entry of np.concatenate and one f-string per
 * read - is most of a PromethION-scale batch's host time.  A client that declares its raw_data to be the int16 signal
 * (`raw_data_dtype = np.int16`) lets the loop use these instead:
 *
 *   lengths(reads, out_i64)                      samples per read (len(raw_data) / 2)
 *   gather(reads, start_i64, out_i16) -> total   out = concat(raw[start[i]:]) in batch order (the staging buffer of the
 *                                                upload: whole reads, or only the samples the device does not hold yet)
 *   format_rows(...) -> str                      the CSV rows of riser/control.py:145-153, floats printed as repr() does
 *   unpack(entries, channels_i64) -> list        entries = [(channel, read), ...]: the channels, and the reads as a list
 *   attrs(reads, name) -> list                   [getattr(r, name) for r in reads] (the read ids of a batch)
 *   lookup(dict, keys, out_i64)                  out[i] = dict.get(keys[i], 0) for integer values (the poly(A) cache)
 *
 * Clients without the flag go through the eight-method duck type unchanged (get_raw_signal per read).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int get_wbuf(PyObject* o, Py_buffer* v, const char* what) {
    if (PyObject_GetBuffer(o, v, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) {
        PyErr_Format(PyExc_TypeError, "%s: a writable contiguous buffer is required", what);
        return -1;
    }
    return 0;
}

static int get_rbuf(PyObject* o, Py_buffer* v, const char* what) {
    if (PyObject_GetBuffer(o, v, PyBUF_C_CONTIGUOUS) != 0) {
        PyErr_Format(PyExc_TypeError, "%s: a contiguous buffer is required", what);
        return -1;
    }
    return 0;
}

/* raw_data of reads[i] as a read-only byte view; the caller releases it */
static int raw_view(PyObject* read, Py_buffer* v) {
    PyObject* raw = PyObject_GetAttrString(read, "raw_data");
    if (!raw) return -1;
    const int rc = PyObject_GetBuffer(raw, v, PyBUF_SIMPLE);
    Py_DECREF(raw);                       /* the view keeps its exporter alive */
    return rc;
}

static PyObject* hp_lengths(PyObject* self, PyObject* args) {
    PyObject *reads, *out;
    if (!PyArg_ParseTuple(args, "OO", &reads, &out)) return NULL;
    if (!PyList_Check(reads)) {
        PyErr_SetString(PyExc_TypeError, "lengths: reads must be a list");
        return NULL;
    }
    Py_buffer ov;
    if (get_wbuf(out, &ov, "lengths") != 0) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(reads);
    if (ov.len < (Py_ssize_t)(n * sizeof(int64_t))) {
        PyBuffer_Release(&ov);
        PyErr_SetString(PyExc_ValueError, "lengths: output buffer too small");
        return NULL;
    }
    int64_t* o = (int64_t*)ov.buf;
    for (Py_ssize_t i = 0; i < n; ++i) {
        Py_buffer v;
        if (raw_view(PyList_GET_ITEM(reads, i), &v) != 0) {
            PyBuffer_Release(&ov);
            return NULL;
        }
        o[i] = (int64_t)(v.len / 2);
        PyBuffer_Release(&v);
    }
    PyBuffer_Release(&ov);
    Py_RETURN_NONE;
}

/* one memcpy job list, split over a few threads when the batch is large (the copies are scattered 2-40 KB pieces: a
 * single core runs them at a fraction of the memory bandwidth) */
typedef struct {
    const char* src;
    char* dst;
    size_t nb;
} piece_t;
typedef struct {
    const piece_t* pc;
    Py_ssize_t lo, hi;
} job_t;

static void* copy_job(void* arg) {
    const job_t* j = (const job_t*)arg;
    for (Py_ssize_t i = j->lo; i < j->hi; ++i) memcpy(j->pc[i].dst, j->pc[i].src, j->pc[i].nb);
    return NULL;
}

#define HP_MAX_THREADS 8
/* copy threads per gather: RS_HOST_THREADS (1..8) caps it - a launcher that runs several ranks on one host gives each
 * rank a slice of the cores (riser_amd/supervise.py) and this pool must fit the slice */
static int hp_thread_cap = HP_MAX_THREADS;

static PyObject* hp_gather(PyObject* self, PyObject* args) {
    PyObject *reads, *start, *out;
    if (!PyArg_ParseTuple(args, "OOO", &reads, &start, &out)) return NULL;
    if (!PyList_Check(reads)) {
        PyErr_SetString(PyExc_TypeError, "gather: reads must be a list");
        return NULL;
    }
    Py_buffer sv, ov;
    if (get_rbuf(start, &sv, "gather(start)") != 0) return NULL;
    if (get_wbuf(out, &ov, "gather(out)") != 0) {
        PyBuffer_Release(&sv);
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(reads);
    const int64_t* st = (const int64_t*)sv.buf;
    char* dst = (char*)ov.buf;
    Py_ssize_t at = 0, held = 0;
    int bad = 0;
    Py_buffer* views = (Py_buffer*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(Py_buffer));
    piece_t* pc = (piece_t*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(piece_t));
    if (!views || !pc) {
        PyErr_NoMemory();
        bad = 1;
    }
    if (!bad && sv.len < (Py_ssize_t)(n * sizeof(int64_t))) {
        PyErr_SetString(PyExc_ValueError, "gather: start has fewer entries than reads");
        bad = 1;
    }
    for (Py_ssize_t i = 0; i < n && !bad; ++i) {
        if (raw_view(PyList_GET_ITEM(reads, i), &views[i]) != 0) {
            bad = 1;
            break;
        }
        ++held;
        const Py_ssize_t nsamp = views[i].len / 2;
        const int64_t s = st[i];
        if (s < 0 || s > nsamp) {
            PyErr_Format(PyExc_ValueError, "gather: start %lld outside read %zd of %zd samples", (long long)s, i, nsamp);
            bad = 1;
            break;
        }
        const Py_ssize_t nb = (nsamp - (Py_ssize_t)s) * 2;
        if (at + nb > ov.len) {
            PyErr_SetString(PyExc_ValueError, "gather: staging buffer too small");
            bad = 1;
            break;
        }
        pc[i].src = (const char*)views[i].buf + 2 * s;
        pc[i].dst = dst + at;
        pc[i].nb = (size_t)nb;
        at += nb;
    }
    if (!bad) {
        /* the views pin the exporters: the copies need no interpreter state */
        Py_BEGIN_ALLOW_THREADS
        int nt = at > (8 << 20) ? HP_MAX_THREADS : at > (3 << 20) ? 4 : 1;
        if (nt > hp_thread_cap) nt = hp_thread_cap;
        if (nt > n) nt = (int)(n ? n : 1);
        job_t jobs[HP_MAX_THREADS];
        pthread_t th[HP_MAX_THREADS];
        int started[HP_MAX_THREADS] = {0};
        /* equal byte shares, cut at piece boundaries */
        Py_ssize_t lo = 0;
        size_t acc = 0;
        for (int t = 0; t < nt; ++t) {
            const size_t want = (size_t)at * (size_t)(t + 1) / (size_t)nt;
            Py_ssize_t hi = lo;
            while (hi < n && (t == nt - 1 || acc + pc[hi].nb <= want || hi == lo)) acc += pc[hi++].nb;
            jobs[t].pc = pc;
            jobs[t].lo = lo;
            jobs[t].hi = hi;
            lo = hi;
        }
        for (int t = 1; t < nt; ++t) started[t] = pthread_create(&th[t], NULL, copy_job, &jobs[t]) == 0;
        copy_job(&jobs[0]);
        for (int t = 1; t < nt; ++t) {
            if (started[t])
                pthread_join(th[t], NULL);
            else
                copy_job(&jobs[t]);
        }
        Py_END_ALLOW_THREADS
    }
    for (Py_ssize_t i = 0; i < held; ++i) PyBuffer_Release(&views[i]);
    PyMem_Free(views);
    PyMem_Free(pc);
    PyBuffer_Release(&sv);
    PyBuffer_Release(&ov);
    if (bad) return NULL;
    return PyLong_FromSsize_t(at / 2);
}

/* growable byte sink */
typedef struct {
    char* p;
    size_t n, cap;
} sink_t;

static int sink_put(sink_t* s, const char* src, size_t k) {
    if (s->n + k + 1 > s->cap) {
        size_t cap = s->cap ? s->cap * 2 : 1 << 16;
        while (cap < s->n + k + 1) cap *= 2;
        char* q = (char*)PyMem_Realloc(s->p, cap);
        if (!q) {
            PyErr_NoMemory();
            return -1;
        }
        s->p = q;
        s->cap = cap;
    }
    memcpy(s->p + s->n, src, k);
    s->n += k;
    return 0;
}

static int sink_i64(sink_t* s, long long v) {
    char tmp[32];
    const int k = snprintf(tmp, sizeof(tmp), "%lld", v);
    return sink_put(s, tmp, (size_t)k);
}

/* format_rows(head, reads, sel_i64, channels_i64, nsamp_i32, mid, p_on_f64 [n, m], m, tail, dec_u8, names) -> str
 * row k: head + str(reads[sel[k]].id) + "," + channels[k] + "," + nsamp[k] + mid + ";".join(repr(p_on[k][j])) + tail +
 * names[dec[k]] + "\n"   (riser/control.py:145-153: p.item() of an fp32 tensor printed with str() = repr of the double) */
static PyObject* hp_format_rows(PyObject* self, PyObject* args) {
    const char *head, *mid, *tail;
    Py_ssize_t head_n, mid_n, tail_n;
    PyObject *reads, *sel, *chan, *nsamp, *pon, *dec, *names;
    int m;
    if (!PyArg_ParseTuple(args, "s#OOOOs#Ois#OO", &head, &head_n, &reads, &sel, &chan, &nsamp, &mid, &mid_n, &pon, &m, &tail,
                          &tail_n, &dec, &names))
        return NULL;
    if (!PyList_Check(reads) || !PyTuple_Check(names) || m < 1) {
        PyErr_SetString(PyExc_TypeError, "format_rows: reads must be a list, names a tuple, m >= 1");
        return NULL;
    }
    Py_buffer bs, bc, bn, bp, bd;
    if (get_rbuf(sel, &bs, "format_rows(sel)") != 0) return NULL;
    if (get_rbuf(chan, &bc, "format_rows(channels)") != 0) {
        PyBuffer_Release(&bs);
        return NULL;
    }
    if (get_rbuf(nsamp, &bn, "format_rows(nsamp)") != 0) {
        PyBuffer_Release(&bs);
        PyBuffer_Release(&bc);
        return NULL;
    }
    if (get_rbuf(pon, &bp, "format_rows(p_on)") != 0) {
        PyBuffer_Release(&bs);
        PyBuffer_Release(&bc);
        PyBuffer_Release(&bn);
        return NULL;
    }
    if (get_rbuf(dec, &bd, "format_rows(dec)") != 0) {
        PyBuffer_Release(&bs);
        PyBuffer_Release(&bc);
        PyBuffer_Release(&bn);
        PyBuffer_Release(&bp);
        return NULL;
    }
    const Py_ssize_t n = bs.len / (Py_ssize_t)sizeof(int64_t);
    const Py_ssize_t n_reads = PyList_GET_SIZE(reads), n_names = PyTuple_GET_SIZE(names);
    sink_t s = {NULL, 0, 0};
    int bad = 0;
    if (bc.len < (Py_ssize_t)(n * sizeof(int64_t)) || bn.len < (Py_ssize_t)(n * sizeof(int32_t)) ||
        bp.len < (Py_ssize_t)(n * m * sizeof(double)) || bd.len < n) {
        PyErr_SetString(PyExc_ValueError, "format_rows: array shorter than the selection");
        bad = 1;
    }
    const int64_t* sel_ = (const int64_t*)bs.buf;
    const int64_t* ch_ = (const int64_t*)bc.buf;
    const int32_t* ns_ = (const int32_t*)bn.buf;
    const double* p_ = (const double*)bp.buf;
    const uint8_t* d_ = (const uint8_t*)bd.buf;
    for (Py_ssize_t k = 0; k < n && !bad; ++k) {
        if (sel_[k] < 0 || sel_[k] >= n_reads || d_[k] >= n_names) {
            PyErr_SetString(PyExc_ValueError, "format_rows: index out of range");
            bad = 1;
            break;
        }
        PyObject* id = PyObject_GetAttrString(PyList_GET_ITEM(reads, sel_[k]), "id");
        if (!id) {
            bad = 1;
            break;
        }
        PyObject* ids = PyUnicode_Check(id) ? (Py_INCREF(id), id) : PyObject_Str(id);
        Py_DECREF(id);
        if (!ids) {
            bad = 1;
            break;
        }
        Py_ssize_t idn;
        const char* idc = PyUnicode_AsUTF8AndSize(ids, &idn);
        if (!idc || sink_put(&s, head, (size_t)head_n) || sink_put(&s, idc, (size_t)idn) || sink_put(&s, ",", 1) ||
            sink_i64(&s, ch_[k]) || sink_put(&s, ",", 1) || sink_i64(&s, ns_[k]) || sink_put(&s, mid, (size_t)mid_n))
            bad = 1;
        Py_DECREF(ids);
        for (int j = 0; j < m && !bad; ++j) {
            char* txt = PyOS_double_to_string(p_[k * m + j], 'r', 0, Py_DTSF_ADD_DOT_0, NULL);
            if (!txt) {
                bad = 1;
                break;
            }
            if ((j && sink_put(&s, ";", 1)) || sink_put(&s, txt, strlen(txt))) bad = 1;
            PyMem_Free(txt);
        }
        if (!bad) {
            Py_ssize_t dn;
            const char* dc = PyUnicode_AsUTF8AndSize(PyTuple_GET_ITEM(names, d_[k]), &dn);
            if (!dc || sink_put(&s, tail, (size_t)tail_n) || sink_put(&s, dc, (size_t)dn) || sink_put(&s, "\n", 1)) bad = 1;
        }
    }
    PyBuffer_Release(&bs);
    PyBuffer_Release(&bc);
    PyBuffer_Release(&bn);
    PyBuffer_Release(&bp);
    PyBuffer_Release(&bd);
    PyObject* res = NULL;
    if (!bad) res = PyUnicode_DecodeUTF8(s.p ? s.p : "", (Py_ssize_t)s.n, "strict");
    PyMem_Free(s.p);
    return res;
}

static PyObject* hp_unpack(PyObject* self, PyObject* args) {
    PyObject *entries, *out;
    if (!PyArg_ParseTuple(args, "OO", &entries, &out)) return NULL;
    if (!PyList_Check(entries)) {
        PyErr_SetString(PyExc_TypeError, "unpack: entries must be a list");
        return NULL;
    }
    Py_buffer ov;
    if (get_wbuf(out, &ov, "unpack(channels)") != 0) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(entries);
    PyObject* reads = NULL;
    if (ov.len < (Py_ssize_t)(n * sizeof(int64_t))) {
        PyErr_SetString(PyExc_ValueError, "unpack: output buffer too small");
        goto done;
    }
    reads = PyList_New(n);
    if (!reads) goto done;
    {
        int64_t* ch = (int64_t*)ov.buf;
        for (Py_ssize_t i = 0; i < n; ++i) {
            PyObject* e = PyList_GET_ITEM(entries, i);
            PyObject *c, *r;
            if (PyTuple_Check(e) && PyTuple_GET_SIZE(e) == 2) {
                c = PyTuple_GET_ITEM(e, 0);
                r = PyTuple_GET_ITEM(e, 1);
            } else if (PyList_Check(e) && PyList_GET_SIZE(e) == 2) {
                c = PyList_GET_ITEM(e, 0);
                r = PyList_GET_ITEM(e, 1);
            } else {
                PyErr_SetString(PyExc_TypeError, "unpack: every entry must be a (channel, read) pair");
                Py_CLEAR(reads);
                goto done;
            }
            const long long v = PyLong_AsLongLong(c);
            if (v == -1 && PyErr_Occurred()) {
                Py_CLEAR(reads);
                goto done;
            }
            ch[i] = (int64_t)v;
            Py_INCREF(r);
            PyList_SET_ITEM(reads, i, r);
        }
    }
done:
    PyBuffer_Release(&ov);
    return reads;
}

static PyObject* hp_attrs(PyObject* self, PyObject* args) {
    PyObject *reads, *name;
    if (!PyArg_ParseTuple(args, "OU", &reads, &name)) return NULL;
    if (!PyList_Check(reads)) {
        PyErr_SetString(PyExc_TypeError, "attrs: reads must be a list");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(reads);
    PyObject* out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* v = PyObject_GetAttr(PyList_GET_ITEM(reads, i), name);
        if (!v) {
            Py_DECREF(out);
            return NULL;
        }
        PyList_SET_ITEM(out, i, v);
    }
    return out;
}

static PyObject* hp_lookup(PyObject* self, PyObject* args) {
    PyObject *dict, *keys, *out;
    if (!PyArg_ParseTuple(args, "O!OO", &PyDict_Type, &dict, &keys, &out)) return NULL;
    /* a contiguous numpy object array exports its PyObject* items (format "O"): read them in place; anything else
     * is walked as a sequence */
    PyObject* seq = NULL;
    Py_buffer kv;
    int have_kv = 0;
    if (PyObject_GetBuffer(keys, &kv, PyBUF_FORMAT | PyBUF_C_CONTIGUOUS) == 0) {
        if (kv.format && strcmp(kv.format, "O") == 0 && kv.itemsize == (Py_ssize_t)sizeof(PyObject*))
            have_kv = 1;
        else
            PyBuffer_Release(&kv);
    } else {
        PyErr_Clear();
    }
    if (!have_kv) {
        seq = PySequence_Fast(keys, "lookup: keys must be a sequence");
        if (!seq) return NULL;
    }
    Py_buffer ov;
    if (get_wbuf(out, &ov, "lookup(out)") != 0) {
        if (have_kv) PyBuffer_Release(&kv);
        Py_XDECREF(seq);
        return NULL;
    }
    const Py_ssize_t n = have_kv ? kv.len / (Py_ssize_t)sizeof(PyObject*) : PySequence_Fast_GET_SIZE(seq);
    if (ov.len < (Py_ssize_t)(n * sizeof(int64_t))) {
        PyBuffer_Release(&ov);
        if (have_kv) PyBuffer_Release(&kv);
        Py_XDECREF(seq);
        PyErr_SetString(PyExc_ValueError, "lookup: output buffer too small");
        return NULL;
    }
    int64_t* o = (int64_t*)ov.buf;
    PyObject** items = have_kv ? (PyObject**)kv.buf : PySequence_Fast_ITEMS(seq);
    int bad = 0;
    for (Py_ssize_t i = 0; i < n && !bad; ++i) {
        PyObject* v = PyDict_GetItemWithError(dict, items[i]);        /* borrowed */
        if (!v) {
            if (PyErr_Occurred()) bad = 1;
            o[i] = 0;
            continue;
        }
        const long long x = PyLong_AsLongLong(v);
        if (x == -1 && PyErr_Occurred()) bad = 1;
        o[i] = (int64_t)x;
    }
    PyBuffer_Release(&ov);
    if (have_kv) PyBuffer_Release(&kv);
    Py_XDECREF(seq);
    if (bad) return NULL;
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"lengths", hp_lengths, METH_VARARGS, "lengths(reads, out_int64): samples of every read's raw_data"},
    {"gather", hp_gather, METH_VARARGS, "gather(reads, start_int64, out_int16) -> samples written"},
    {"format_rows", hp_format_rows, METH_VARARGS, "CSV rows of one batch as one string"},
    {"unpack", hp_unpack, METH_VARARGS, "unpack(entries, channels_int64) -> reads: splits [(channel, read), ...]"},
    {"attrs", hp_attrs, METH_VARARGS, "attrs(reads, name) -> [getattr(r, name) for r in reads]"},
    {"lookup", hp_lookup, METH_VARARGS, "lookup(dict, keys, out_int64): out[i] = dict.get(keys[i], 0)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hostpack", "per-read host loops of a ReadUntil batch", -1, methods};

PyMODINIT_FUNC PyInit__hostpack(void) {
    const char* cap = getenv("RS_HOST_THREADS");
    if (cap && *cap) {
        const long v = strtol(cap, NULL, 10);
        hp_thread_cap = v < 1 ? 1 : v > HP_MAX_THREADS ? HP_MAX_THREADS : (int)v;
    }
    PyObject* m = PyModule_Create(&moddef);
    if (m) PyModule_AddIntConstant(m, "thread_cap", hp_thread_cap);
    return m;
}
