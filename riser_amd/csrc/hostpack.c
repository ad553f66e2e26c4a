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

/* A persistent pool for the copy jobs: a PromethION-scale batch calls gather() once per slice (five times per batch, ~11 MB
 * each), and creating + joining seven threads per call cost more than a millisecond per batch.  HP_MAX_THREADS - 1 workers are
 * started at the first large gather and sleep on a condition variable between jobs; the calling thread takes share 0 itself.
 * One gather at a time (pool_busy): a second caller - the staging thread of the slice pipeline next to the loop's thread -
 * falls back to copying alone.  The workers never touch the interpreter. */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    int started;                 /* worker threads alive */
    unsigned long gen;           /* job generation */
    int n_jobs, pending;         /* shares of the current job (share 0 is the caller's), shares not finished yet */
    const job_t* jobs;
    int busy;
} pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, 0, NULL, 0};

static void* pool_worker(void* arg) {
    const int me = (int)(intptr_t)arg;          /* 1 .. HP_MAX_THREADS - 1: takes share `me` of a job that has one */
    unsigned long seen = 0;
    pthread_mutex_lock(&pool.mu);
    for (;;) {
        while (pool.gen == seen) pthread_cond_wait(&pool.go, &pool.mu);
        seen = pool.gen;
        if (me < pool.n_jobs) {
            const job_t* j = &pool.jobs[me];
            pthread_mutex_unlock(&pool.mu);
            copy_job((void*)j);
            pthread_mutex_lock(&pool.mu);
            if (--pool.pending == 0) pthread_cond_signal(&pool.done);
        }
    }
    return NULL;
}

/* run jobs[0 .. nt) : share 0 on this thread, the others on the pool (or on this thread, when the pool is busy or absent) */
static void run_copy_jobs(const job_t* jobs, int nt) {
    int use_pool = 0;
    if (nt > 1) {
        pthread_mutex_lock(&pool.mu);
        if (!pool.busy) {
            while (pool.started < HP_MAX_THREADS - 1) {
                pthread_t th;
                pthread_attr_t at;
                pthread_attr_init(&at);
                pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
                const int ok = pthread_create(&th, &at, pool_worker, (void*)(intptr_t)(pool.started + 1)) == 0;
                pthread_attr_destroy(&at);
                if (!ok) break;
                ++pool.started;
            }
            if (pool.started >= nt - 1) {
                pool.busy = 1;
                pool.jobs = jobs;
                pool.n_jobs = nt;
                pool.pending = nt - 1;
                ++pool.gen;
                pthread_cond_broadcast(&pool.go);
                use_pool = 1;
            }
        }
        pthread_mutex_unlock(&pool.mu);
    }
    copy_job((void*)&jobs[0]);
    if (use_pool) {
        pthread_mutex_lock(&pool.mu);
        while (pool.pending) pthread_cond_wait(&pool.done, &pool.mu);
        pool.n_jobs = 0;
        pool.busy = 0;
        pthread_mutex_unlock(&pool.mu);
    } else {
        for (int t = 1; t < nt; ++t) copy_job((void*)&jobs[t]);
    }
}

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
        run_copy_jobs(jobs, nt);
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

/* store_slice(reads, ids_obj, row_ids_obj, rows_i64, lens_i64, fits_u8, row_have_i64, row_tail_i16, T, stage_i16,
 *             start_i64_out, cand_u8_out, stats_i64_out[3]) -> samples staged
 * One slice of a batch through the device-resident signal store's host side (riser_amd/control.py:_SignalStore.update), as ONE
 * call.  Per read i with row r = rows[i]: it is RE-SEEN when it fits a row, the row holds the same id (the same object, or ==) and
 * samples; a re-seen read that is not shorter than what the row holds is a CANDIDATE for the delta path, and it takes it when its
 * samples [have - T, have) equal the T samples the row kept (row_tail) - checked in the read's own buffer, before anything is
 * copied.  start[i] = have - T for a delta read, else 0; raw[start:] of every read goes to `stage` back to back (the copy-thread
 * pool); then the rows learn their reads: row_have = the read's length (0 for a read too long for a row), row_tail = its last T
 * samples (reads of at least T), row_ids = its id.  stats = (re-seen, delta, candidates whose overlap did not match).
 * The id comparison and the row_ids update hold the interpreter lock; the overlap check, the copies and the tails do not. */
static PyObject* hp_store_slice(PyObject* self, PyObject* args) {
    PyObject *reads, *ids, *row_ids, *rows, *lens, *fits, *row_have, *row_tail, *stage, *start, *cand, *stats;
    int T;
    if (!PyArg_ParseTuple(args, "OOOOOOOOiOOOO", &reads, &ids, &row_ids, &rows, &lens, &fits, &row_have, &row_tail, &T, &stage, &start,
                          &cand, &stats))
        return NULL;
    if (!PyList_Check(reads) || T < 1 || T > 4096) {
        PyErr_SetString(PyExc_TypeError, "store_slice: reads must be a list, 1 <= T <= 4096");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(reads);
    Py_buffer b[10];
    PyObject* objs[10] = {ids, row_ids, rows, lens, fits, row_have, row_tail, stage, start, cand};
    const int writable[10] = {0, 1, 0, 0, 0, 1, 1, 1, 1, 1};
    const int as_objects[10] = {1, 1, 0, 0, 0, 0, 0, 0, 0, 0};
    int got = 0;
    Py_buffer bst;
    int have_bst = 0;
    PyObject* result = NULL;
    Py_buffer* views = NULL;
    piece_t* pc = NULL;
    unsigned char* same = NULL;
    Py_ssize_t held = 0;
    for (; got < 10; ++got) {
        const int flags = (writable[got] ? PyBUF_WRITABLE : 0) | (as_objects[got] ? PyBUF_FORMAT : 0) | PyBUF_C_CONTIGUOUS;
        if (PyObject_GetBuffer(objs[got], &b[got], flags) != 0) goto done;
        if (as_objects[got] && !(b[got].format && strcmp(b[got].format, "O") == 0 && b[got].itemsize == (Py_ssize_t)sizeof(PyObject*))) {
            ++got;
            PyErr_SetString(PyExc_TypeError, "store_slice: ids / row_ids must be contiguous object arrays");
            goto done;
        }
    }
    if (get_wbuf(stats, &bst, "store_slice(stats)") != 0) goto done;
    have_bst = 1;
    {
        const Py_ssize_t n_rows = b[5].len / (Py_ssize_t)sizeof(int64_t);
        if (b[0].len < (Py_ssize_t)(n * sizeof(PyObject*)) || b[1].len < (Py_ssize_t)(n_rows * sizeof(PyObject*)) ||
            b[2].len < (Py_ssize_t)(n * 8) || b[3].len < (Py_ssize_t)(n * 8) || b[4].len < n ||
            b[6].len < (Py_ssize_t)(n_rows * T * 2) || b[8].len < (Py_ssize_t)(n * 8) || b[9].len < n || bst.len < 24) {
            PyErr_SetString(PyExc_ValueError, "store_slice: an array is shorter than the slice / the row table");
            goto done;
        }
        PyObject** id_ = (PyObject**)b[0].buf;
        PyObject** rid_ = (PyObject**)b[1].buf;
        const int64_t* row_ = (const int64_t*)b[2].buf;
        const int64_t* len_ = (const int64_t*)b[3].buf;
        const uint8_t* fit_ = (const uint8_t*)b[4].buf;
        int64_t* have_ = (int64_t*)b[5].buf;
        int16_t* tail_ = (int16_t*)b[6].buf;
        char* stage_ = (char*)b[7].buf;
        int64_t* start_ = (int64_t*)b[8].buf;
        uint8_t* cand_ = (uint8_t*)b[9].buf;
        int64_t* stats_ = (int64_t*)bst.buf;
        views = (Py_buffer*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(Py_buffer));
        pc = (piece_t*)PyMem_Malloc((size_t)(n ? n : 1) * sizeof(piece_t));
        same = (unsigned char*)PyMem_Malloc((size_t)(n ? n : 1));
        if (!views || !pc || !same) {
            PyErr_NoMemory();
            goto done;
        }
        /* ---- with the interpreter lock: the reads' buffers, "does the row hold this id" ---- */
        for (Py_ssize_t i = 0; i < n; ++i) {
            if (row_[i] < 0 || row_[i] >= n_rows) {
                PyErr_SetString(PyExc_ValueError, "store_slice: row out of range");
                goto done;
            }
            if (raw_view(PyList_GET_ITEM(reads, i), &views[i]) != 0) goto done;
            ++held;
            if (views[i].len / 2 != (Py_ssize_t)len_[i]) {
                PyErr_SetString(PyExc_ValueError, "store_slice: a read's length changed under the batch");
                goto done;
            }
            same[i] = 0;
            if (fit_[i]) {
                PyObject* r = rid_[row_[i]];
                if (r == id_[i])
                    same[i] = 1;
                else if (r != NULL && r != Py_None) {
                    const int eq = PyObject_RichCompareBool(r, id_[i], Py_EQ);
                    if (eq < 0) goto done;
                    same[i] = (unsigned char)eq;
                }
            }
        }
        /* ---- without it: overlap check, piece list, copies, the rows' new lengths and tails ---- */
        Py_ssize_t at = 0;
        int64_t n_reseen = 0, n_delta = 0, n_bad = 0;
        int overflow = 0;
        Py_BEGIN_ALLOW_THREADS
        for (Py_ssize_t i = 0; i < n; ++i) {
            const int64_t L = len_[i];
            const char* raw = (const char*)views[i].buf;
            int64_t st = 0;
            uint8_t c = 0;
            if (fit_[i]) {
                const int64_t have = have_[row_[i]];
                if (same[i] && have > 0) {
                    ++n_reseen;
                    if (have <= L && have >= T) {
                        if (memcmp(raw + 2 * (have - T), tail_ + row_[i] * (int64_t)T, (size_t)T * 2) == 0) {
                            c = 1;
                            st = have - T;
                            ++n_delta;
                        } else {
                            ++n_bad;
                        }
                    }
                }
            }
            start_[i] = st;
            cand_[i] = c;
            const Py_ssize_t nb = (Py_ssize_t)(L - st) * 2;
            if (at + nb > b[7].len) {
                overflow = 1;
                break;
            }
            pc[i].src = raw + 2 * st;
            pc[i].dst = stage_ + at;
            pc[i].nb = (size_t)nb;
            at += nb;
        }
        if (!overflow) {
            int nt = at > (8 << 20) ? HP_MAX_THREADS : at > (3 << 20) ? 4 : 1;
            if (nt > hp_thread_cap) nt = hp_thread_cap;
            if (nt > n) nt = (int)(n ? n : 1);
            job_t jobs[HP_MAX_THREADS];
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
            if (n) run_copy_jobs(jobs, nt);
            for (Py_ssize_t i = 0; i < n; ++i) {
                if (fit_[i]) {
                    have_[row_[i]] = len_[i];
                    if (len_[i] >= T)
                        memcpy(tail_ + row_[i] * (int64_t)T, (const char*)views[i].buf + 2 * (len_[i] - T), (size_t)T * 2);
                } else {
                    have_[row_[i]] = 0;
                }
            }
        }
        Py_END_ALLOW_THREADS
        if (overflow) {
            PyErr_SetString(PyExc_ValueError, "store_slice: staging buffer too small");
            goto done;
        }
        /* ---- with it again: the rows' ids ---- */
        for (Py_ssize_t i = 0; i < n; ++i)
            if (fit_[i] && rid_[row_[i]] != id_[i]) {
                PyObject* old = rid_[row_[i]];
                Py_INCREF(id_[i]);
                rid_[row_[i]] = id_[i];
                Py_XDECREF(old);
            }
        stats_[0] = n_reseen;
        stats_[1] = n_delta;
        stats_[2] = n_bad;
        result = PyLong_FromSsize_t(at / 2);
    }
done:
    for (Py_ssize_t i = 0; i < held; ++i) PyBuffer_Release(&views[i]);
    PyMem_Free(views);
    PyMem_Free(pc);
    PyMem_Free(same);
    if (have_bst) PyBuffer_Release(&bst);
    for (int k = 0; k < got; ++k) PyBuffer_Release(&b[k]);
    return result;
}

/* growable byte sink on the C heap: usable without the interpreter lock */
typedef struct {
    char* p;
    size_t n, cap;
    int oom;
} csink_t;

static void cs_put(csink_t* s, const char* src, size_t k) {
    if (s->oom) return;
    if (s->n + k + 1 > s->cap) {
        size_t cap = s->cap ? s->cap * 2 : 1 << 16;
        while (cap < s->n + k + 1) cap *= 2;
        char* q = (char*)realloc(s->p, cap);
        if (!q) {
            s->oom = 1;
            return;
        }
        s->p = q;
        s->cap = cap;
    }
    memcpy(s->p + s->n, src, k);
    s->n += k;
}

static void cs_i64(csink_t* s, long long v) {
    char tmp[32];
    const int k = snprintf(tmp, sizeof(tmp), "%lld", v);
    cs_put(s, tmp, (size_t)k);
}

/* repr() of a Python float without the interpreter: the shortest decimal string that reads back as the same double, in
 * repr's layout (exponent form below 1e-4 and from 1e16 on, ".0" behind an integer).  A correctly rounded 15-digit
 * decimal is the shortest form whenever one of <= 15 digits exists (trailing zeros stripped by %g); otherwise the
 * correctly rounded 16-digit decimal if it reads back, else 17 digits - which is what David Gay's mode-0 conversion
 * behind repr() returns (tests/test_host_cpu.py compares the two on 400 k values).  Returns the length. */
static int fmt_double_repr(double v, char* out /* >= 40 bytes */) {
    if (v != v) return snprintf(out, 40, "nan");
    if (v - v != 0.0) return snprintf(out, 40, v > 0 ? "inf" : "-inf");
    int k = 0;
    const double mag = v < 0 ? -v : v;
    /* a subnormal carries fewer than 15 significant digits: its shortest form can be shorter than its 15-digit rounding */
    for (int prec = (mag != 0.0 && mag < 2.2250738585072014e-308) ? 1 : 15; prec <= 17; ++prec) {
        k = snprintf(out, 40, "%.*g", prec, v);
        if (prec == 17 || strtod(out, NULL) == v) break;
    }
    /* %g switches to the exponent form at 10^precision, repr at 10^16: values in [1e15, 1e17) can differ */
    const double a = v < 0 ? -v : v;
    if (a >= 1e15 && a < 1e17) {
        if (a < 1e16 && memchr(out, 'e', (size_t)k)) {            /* repr: positional */
            k = snprintf(out, 40, "%.1f", v);
        } else if (a >= 1e16 && !memchr(out, 'e', (size_t)k)) {   /* repr: exponent form, shortest mantissa */
            for (int prec = 0; prec <= 16; ++prec) {
                k = snprintf(out, 40, "%.*e", prec, v);
                if (prec == 16 || strtod(out, NULL) == v) break;
            }
        }
    }
    if (!memchr(out, '.', (size_t)k) && !memchr(out, 'e', (size_t)k)) {
        out[k++] = '.';
        out[k++] = '0';
        out[k] = 0;
    }
    return k;
}

static PyObject* hp_repr_double(PyObject* self, PyObject* arg) {
    const double v = PyFloat_AsDouble(arg);
    if (v == -1.0 && PyErr_Occurred()) return NULL;
    char tmp[40];
    const int k = fmt_double_repr(v, tmp);
    return PyUnicode_FromStringAndSize(tmp, k);
}

/* format_rows(head, reads, sel_i64, channels_i64, nsamp_i32, mid, p_on_f64 [n, m], m, tail, dec_u8, names) -> str
 * row k: head + str(reads[sel[k]].id) + "," + channels[k] + "," + nsamp[k] + mid + ";".join(repr(p_on[k][j])) + tail +
 * names[dec[k]] + "\n"   (riser/control.py:145-153: p.item() of an fp32 tensor printed with str() = repr of the double).
 * Two phases: the ids are collected with the interpreter lock held (references kept until the end), the text is built
 * WITHOUT it - a writer thread formats the rows of batch k while the control loop assesses batch k + 1. */
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
    /* ---- phase 1, interpreter lock held: the id text of every row, the decision names ---- */
    PyObject* keep = bad ? NULL : PyList_New(n);                  /* owns the id strings while phase 2 reads their bytes */
    const char** idp = (const char**)malloc((size_t)(n ? n : 1) * sizeof(char*));
    Py_ssize_t* idn = (Py_ssize_t*)malloc((size_t)(n ? n : 1) * sizeof(Py_ssize_t));
    const char* namep[256];
    Py_ssize_t namen[256];
    if (!bad && (!keep || !idp || !idn)) {
        PyErr_NoMemory();
        bad = 1;
    }
    if (!bad && n_names > 256) {
        PyErr_SetString(PyExc_ValueError, "format_rows: more than 256 decision names");
        bad = 1;
    }
    for (Py_ssize_t j = 0; j < n_names && !bad; ++j) {
        namep[j] = PyUnicode_AsUTF8AndSize(PyTuple_GET_ITEM(names, j), &namen[j]);
        if (!namep[j]) bad = 1;
    }
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
        PyList_SET_ITEM(keep, k, ids);                            /* steals the reference */
        idp[k] = PyUnicode_AsUTF8AndSize(ids, &idn[k]);
        if (!idp[k]) bad = 1;
    }
    /* ---- phase 2, no interpreter lock: the text ---- */
    csink_t s = {NULL, 0, 0, 0};
    if (!bad) {
        Py_BEGIN_ALLOW_THREADS
        char tmp[40];
        for (Py_ssize_t k = 0; k < n; ++k) {
            cs_put(&s, head, (size_t)head_n);
            cs_put(&s, idp[k], (size_t)idn[k]);
            cs_put(&s, ",", 1);
            cs_i64(&s, ch_[k]);
            cs_put(&s, ",", 1);
            cs_i64(&s, ns_[k]);
            cs_put(&s, mid, (size_t)mid_n);
            for (int j = 0; j < m; ++j) {
                if (j) cs_put(&s, ";", 1);
                cs_put(&s, tmp, (size_t)fmt_double_repr(p_[k * m + j], tmp));
            }
            cs_put(&s, tail, (size_t)tail_n);
            cs_put(&s, namep[d_[k]], (size_t)namen[d_[k]]);
            cs_put(&s, "\n", 1);
        }
        Py_END_ALLOW_THREADS
        if (s.oom) {
            PyErr_NoMemory();
            bad = 1;
        }
    }
    PyBuffer_Release(&bs);
    PyBuffer_Release(&bc);
    PyBuffer_Release(&bn);
    PyBuffer_Release(&bp);
    PyBuffer_Release(&bd);
    PyObject* res = NULL;
    if (!bad) res = PyUnicode_DecodeUTF8(s.p ? s.p : "", (Py_ssize_t)s.n, "strict");
    free(s.p);
    free((void*)idp);
    free(idn);
    Py_XDECREF(keep);
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

/* decided(reads, sel_i64, channels_i64, dec_u8, codes) -> tuple of len(codes) lists: list c holds (channels[k], key(reads[sel[k]]))
 * for every k with dec[k] == codes[c], in order; key(read) = read.number if the read HAS that attribute (whatever its value),
 * else read.id - riser/control.py:137-143.  The per-read Python of the reject / finish lists of a batch. */
static PyObject* hp_decided(PyObject* self, PyObject* args) {
    PyObject *reads, *sel, *chan, *dec, *codes;
    if (!PyArg_ParseTuple(args, "OOOOO", &reads, &sel, &chan, &dec, &codes)) return NULL;
    if (!PyList_Check(reads) || !PyTuple_Check(codes)) {
        PyErr_SetString(PyExc_TypeError, "decided: reads must be a list, codes a tuple");
        return NULL;
    }
    const Py_ssize_t n_codes = PyTuple_GET_SIZE(codes);
    long code[8];
    if (n_codes < 1 || n_codes > 8) {
        PyErr_SetString(PyExc_ValueError, "decided: 1 to 8 codes");
        return NULL;
    }
    for (Py_ssize_t c = 0; c < n_codes; ++c) {
        code[c] = PyLong_AsLong(PyTuple_GET_ITEM(codes, c));
        if (code[c] == -1 && PyErr_Occurred()) return NULL;
    }
    Py_buffer bs, bc, bd;
    if (get_rbuf(sel, &bs, "decided(sel)") != 0) return NULL;
    if (get_rbuf(chan, &bc, "decided(channels)") != 0) {
        PyBuffer_Release(&bs);
        return NULL;
    }
    if (get_rbuf(dec, &bd, "decided(dec)") != 0) {
        PyBuffer_Release(&bs);
        PyBuffer_Release(&bc);
        return NULL;
    }
    const Py_ssize_t n = bs.len / (Py_ssize_t)sizeof(int64_t), n_reads = PyList_GET_SIZE(reads);
    PyObject* out = NULL;
    PyObject* s_number = NULL;
    PyObject* s_id = NULL;
    int bad = 0;
    if (bc.len < (Py_ssize_t)(n * sizeof(int64_t)) || bd.len < n) {
        PyErr_SetString(PyExc_ValueError, "decided: array shorter than the selection");
        bad = 1;
    }
    if (!bad) {
        out = PyTuple_New(n_codes);
        s_number = PyUnicode_InternFromString("number");
        s_id = PyUnicode_InternFromString("id");
        if (!out || !s_number || !s_id) bad = 1;
        for (Py_ssize_t c = 0; c < n_codes && !bad; ++c) {
            PyObject* l = PyList_New(0);
            if (!l) bad = 1; else PyTuple_SET_ITEM(out, c, l);
        }
    }
    const int64_t* sel_ = (const int64_t*)bs.buf;
    const int64_t* ch_ = (const int64_t*)bc.buf;
    const uint8_t* d_ = (const uint8_t*)bd.buf;
    for (Py_ssize_t k = 0; k < n && !bad; ++k) {
        Py_ssize_t c = 0;
        while (c < n_codes && code[c] != (long)d_[k]) ++c;
        if (c == n_codes) continue;
        if (sel_[k] < 0 || sel_[k] >= n_reads) {
            PyErr_SetString(PyExc_ValueError, "decided: index out of range");
            bad = 1;
            break;
        }
        PyObject* read = PyList_GET_ITEM(reads, sel_[k]);
        PyObject* key = PyObject_GetAttr(read, s_number);
        if (!key) {
            if (!PyErr_ExceptionMatches(PyExc_AttributeError)) {
                bad = 1;
                break;
            }
            PyErr_Clear();
            key = PyObject_GetAttr(read, s_id);
            if (!key) {
                bad = 1;
                break;
            }
        }
        PyObject* chn = PyLong_FromLongLong((long long)ch_[k]);
        PyObject* tup = chn ? PyTuple_Pack(2, chn, key) : NULL;
        Py_XDECREF(chn);
        Py_DECREF(key);
        if (!tup || PyList_Append(PyTuple_GET_ITEM(out, c), tup) != 0) bad = 1;
        Py_XDECREF(tup);
    }
    PyBuffer_Release(&bs);
    PyBuffer_Release(&bc);
    PyBuffer_Release(&bd);
    Py_XDECREF(s_number);
    Py_XDECREF(s_id);
    if (bad) {
        Py_XDECREF(out);
        return NULL;
    }
    return out;
}

static PyMethodDef methods[] = {
    {"lengths", hp_lengths, METH_VARARGS, "lengths(reads, out_int64): samples of every read's raw_data"},
    {"gather", hp_gather, METH_VARARGS, "gather(reads, start_int64, out_int16) -> samples written"},
    {"format_rows", hp_format_rows, METH_VARARGS, "CSV rows of one batch as one string (text built without the GIL)"},
    {"repr_double", hp_repr_double, METH_O, "repr(float) as format_rows writes it (test hook)"},
    {"store_slice", hp_store_slice, METH_VARARGS, "one slice through the signal store's host side: delta candidates, staging, row state"},
    {"decided", hp_decided, METH_VARARGS, "decided(reads, sel, channels, dec, codes) -> per code a list of (channel, read key)"},
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
