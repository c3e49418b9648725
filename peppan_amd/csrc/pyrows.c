/* _pyrows.so - the ONE place where the numeric hit table turns into the reference's table of Python objects
 * (ndarray(dtype=object)[n, 15 | 16], layout: SURVEY.md section 8; uberBlast.py:57-58, 280-288, 354, 480).
 *
 * Callers inside this package work on the numeric columns (peppan_amd/hittable.py); an unmodified PEPPAN caller of uberBlast() gets the
 * 16-column object rows the reference returns - 70 000 rows x 16 cells for the 10k-gene all-vs-all.  Built cell by cell in Python
 * (16 tolist() columns, CIGAR strings, one column assignment each) that cost 100 ms for a 3 ms search.  Here it is one C pass over
 * the columns through the CPython API: every cell gets the same Python type and value as before, and immutable values that repeat
 * (small integers, gene lengths, 3-decimal identities, integer-valued scores, the names) are created once and shared by reference.
 *
 * Loaded with ctypes.PyDLL (the GIL is held during the call); no module initialisation, no numpy C API: the caller passes the address
 * of the object array's buffer, whose n * width cells hold owned references (to None) that are replaced here. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define INT_CACHE 65536
static PyObject *int_cache[INT_CACHE];          /* persistent: 0 .. 65535 */

static PyObject *cached_int(int64_t v)
{
    if (v >= 0 && v < INT_CACHE) {
        if (!int_cache[v]) int_cache[v] = PyLong_FromLongLong(v);
        Py_XINCREF(int_cache[v]);
        return int_cache[v];
    }
    return PyLong_FromLongLong(v);
}

/* per-call cache of float objects by bit pattern (open addressing; a full table simply stops caching) */
#define F_SLOTS 8192
typedef struct { uint64_t bits[F_SLOTS]; PyObject *obj[F_SLOTS]; int used; } fcache;

static PyObject *cached_float(fcache *c, double v)
{
    uint64_t b;
    memcpy(&b, &v, 8);
    uint32_t h = (uint32_t)((b * 0x9E3779B97F4A7C15ull) >> 51);       /* 13 bits */
    for (int probe = 0; probe < 16; ++probe, h = (h + 1) & (F_SLOTS - 1)) {
        if (c->obj[h] == NULL) {
            if (c->used > F_SLOTS / 2) break;
            PyObject *o = PyFloat_FromDouble(v);
            if (!o) return NULL;
            c->bits[h] = b; c->obj[h] = o; c->used++;
            Py_INCREF(o);
            return o;
        }
        if (c->bits[h] == b) { Py_INCREF(c->obj[h]); return c->obj[h]; }
    }
    return PyFloat_FromDouble(v);
}

static void fcache_release(fcache *c)
{
    for (int i = 0; i < F_SLOTS; ++i) Py_XDECREF(c->obj[i]);
}

static int put(PyObject **cell, PyObject *o)
{
    if (!o) return -1;
    PyObject *old = *cell;
    *cell = o;
    Py_XDECREF(old);
    return 0;
}

/* cigar_mode: 0 = leave column 14 alone, 1 = text "150M3D150M" (uberBlast.py:480), 2 = [[n, 'M'], ...] (uberBlast.py:33, 316-319).
 * rid == NULL: 15 columns.  Returns 0, or -1 with a Python exception set. */
int pep_rows_fill(PyObject **cells, int64_t n, int64_t width, PyObject *q_names, PyObject *r_names, const int64_t *qi, const int64_t *ri,
                  const double *iden, const int64_t *aln, const int64_t *mis, const int64_t *gap, const int64_t *qs, const int64_t *qe,
                  const int64_t *ss, const int64_t *se, const double *evalue, const double *score, int score_is_int, const int64_t *ql,
                  const int64_t *sl, const uint32_t *arena, const int64_t *c_off, const int64_t *c_runs, int cigar_mode, const int64_t *rid)
{
    if (!PyList_Check(q_names) || !PyList_Check(r_names)) { PyErr_SetString(PyExc_TypeError, "name tables must be lists"); return -1; }
    const Py_ssize_t nq = PyList_GET_SIZE(q_names), nr = PyList_GET_SIZE(r_names);
    fcache *fc = (fcache *)calloc(1, sizeof(fcache));
    if (!fc) { PyErr_NoMemory(); return -1; }
    PyObject *ops[3] = {PyUnicode_InternFromString("M"), PyUnicode_InternFromString("I"), PyUnicode_InternFromString("D")};
    char *buf = NULL;
    size_t buf_cap = 0;
    int rc = 0;
    for (int64_t k = 0; k < n && rc == 0; ++k) {
        PyObject **row = cells + k * width;
        if (qi[k] < 0 || qi[k] >= nq || ri[k] < 0 || ri[k] >= nr) { PyErr_SetString(PyExc_IndexError, "name index out of range"); rc = -1; break; }
        PyObject *qn = PyList_GET_ITEM(q_names, qi[k]), *rn = PyList_GET_ITEM(r_names, ri[k]);
        Py_INCREF(qn); Py_INCREF(rn);
        rc |= put(row + 0, qn);
        rc |= put(row + 1, rn);
        rc |= put(row + 2, cached_float(fc, iden[k]));
        rc |= put(row + 3, cached_int(aln[k]));
        rc |= put(row + 4, cached_int(mis[k]));
        rc |= put(row + 5, cached_int(gap[k]));
        rc |= put(row + 6, cached_int(qs[k]));
        rc |= put(row + 7, cached_int(qe[k]));
        rc |= put(row + 8, cached_int(ss[k]));
        rc |= put(row + 9, cached_int(se[k]));
        rc |= put(row + 10, cached_float(fc, evalue[k]));
        rc |= put(row + 11, score_is_int ? cached_int((int64_t)score[k]) : cached_float(fc, score[k]));
        rc |= put(row + 12, cached_int(ql[k]));
        rc |= put(row + 13, cached_int(sl[k]));
        if (cigar_mode == 1) {
            const size_t need = (size_t)c_runs[k] * 12 + 1;
            if (need > buf_cap) { buf_cap = need * 2; char *nb = (char *)realloc(buf, buf_cap); if (!nb) { PyErr_NoMemory(); rc = -1; break; } buf = nb; }
            size_t len = 0;
            for (int64_t x = 0; x < c_runs[k]; ++x) {
                const uint32_t run = arena[c_off[k] + x];
                uint32_t v = run >> 2;
                char tmp[12];
                int d = 0;
                do { tmp[d++] = (char)('0' + v % 10); v /= 10; } while (v);
                while (d) buf[len++] = tmp[--d];
                buf[len++] = "MID?"[run & 3u];
            }
            PyObject *txt = PyUnicode_New((Py_ssize_t)len, 127);                    /* ASCII: no decoding pass */
            if (txt && len) memcpy(PyUnicode_1BYTE_DATA(txt), buf, len);
            rc |= put(row + 14, txt);
        } else if (cigar_mode == 2) {
            PyObject *lst = PyList_New((Py_ssize_t)c_runs[k]);
            if (!lst) { rc = -1; break; }
            for (int64_t x = 0; x < c_runs[k]; ++x) {
                const uint32_t run = arena[c_off[k] + x];
                PyObject *pair = PyList_New(2), *num = cached_int((int64_t)(run >> 2));
                if (!pair || !num || (run & 3u) > 2) { Py_XDECREF(pair); Py_XDECREF(num); Py_DECREF(lst); lst = NULL; if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "bad CIGAR run"); break; }
                Py_INCREF(ops[run & 3u]);
                PyList_SET_ITEM(pair, 0, num);
                PyList_SET_ITEM(pair, 1, ops[run & 3u]);
                PyList_SET_ITEM(lst, (Py_ssize_t)x, pair);
            }
            rc |= put(row + 14, lst);
        }
        if (rid) rc |= put(row + 15, cached_int(rid[k]));
    }
    free(buf);
    for (int i = 0; i < 3; ++i) Py_XDECREF(ops[i]);
    fcache_release(fc);
    free(fc);
    if (rc != 0 && !PyErr_Occurred()) PyErr_SetString(PyExc_MemoryError, "pep_rows_fill: object creation failed");
    return rc == 0 ? 0 : -1;
}

/* ---- the front end of writeGenes (PEPPAN.py:1023-1031) in one pass over the two dictionaries.
 * For every (name, value) of `priority`, in dictionary order, whose name is a key of `genes` with a non-empty sequence (genes[name][6]):
 *   names_out.append(name);  p0[i], p1[i] = value[0], value[1];  code[20 i ..] = value[2] as 20 big-endian bytes;
 *   seq_len[i] = len(genes[name][6]);  digest[20 i ..] = genes[name][5] as 20 big-endian bytes.
 * Five million instances cost 12 s as Python comprehensions (two dictionary look-ups, a len(), three int conversions and two
 * to_bytes() per instance); here it is one C loop.  Returns the number of instances, -2 when a value does not have the expected shape
 * (priority values that are not [int, int, non-negative int below 2^160]: the caller then sorts the plain way), -1 with an exception set. */
Py_ssize_t pep_genes_scan(PyObject *priority, PyObject *genes, PyObject *names_out, int64_t *p0, int64_t *p1, uint8_t *code, int64_t *seq_len,
                          uint8_t *digest, Py_ssize_t cap)
{
    if (!PyDict_Check(priority) || !PyDict_Check(genes) || !PyList_Check(names_out)) { PyErr_SetString(PyExc_TypeError, "pep_genes_scan: dict, dict, list expected"); return -1; }
    Py_ssize_t pos = 0, n = 0;
    PyObject *name, *val;
    while (PyDict_Next(priority, &pos, &name, &val)) {
        PyObject *g = PyDict_GetItemWithError(genes, name);            /* borrowed */
        if (!g) { if (PyErr_Occurred()) return -1; continue; }
        PyObject *seq = PySequence_GetItem(g, 6);
        if (!seq) return -1;
        const Py_ssize_t L = PyObject_Length(seq);
        Py_DECREF(seq);
        if (L < 0) return -1;
        if (L == 0) continue;
        if (n >= cap) { PyErr_SetString(PyExc_IndexError, "pep_genes_scan: output arrays too small"); return -1; }
        PyObject *h = PySequence_GetItem(g, 5);
        if (!h) return -1;
        int bad = !PyLong_Check(h) || _PyLong_AsByteArray((PyLongObject *)h, digest + 20 * n, 20, 0, 0) < 0;
        Py_DECREF(h);
        if (bad) { PyErr_Clear(); return -2; }
        if (!(PyList_Check(val) || PyTuple_Check(val)) || PySequence_Fast_GET_SIZE(val) != 3) return -2;
        PyObject **it = PySequence_Fast_ITEMS(val);
        if (!PyLong_CheckExact(it[0]) || !PyLong_CheckExact(it[1]) || !PyLong_Check(it[2])) return -2;
        int o0 = 0, o1 = 0;
        p0[n] = PyLong_AsLongLongAndOverflow(it[0], &o0);
        p1[n] = PyLong_AsLongLongAndOverflow(it[1], &o1);
        if (o0 || o1) return -2;
        if (_PyLong_AsByteArray((PyLongObject *)it[2], code + 20 * n, 20, 0, 0) < 0) { PyErr_Clear(); return -2; }
        seq_len[n] = (int64_t)L;
        if (PyList_Append(names_out, name) < 0) return -1;
        ++n;
    }
    return n;
}

/* ---- the input of K13 (pep_sha1) and its output as PEPPAN wants it ------------------------------------------------------------------------
 * pep_strs_measure: total number of bytes of a list of str (ASCII only) / bytes objects, lengths written to len_out[0 .. n); -2 when an
 * element is something else (the caller then packs the plain way), -1 with an exception set.
 * pep_strs_pack: their bytes, back to back, into out (one copy; ''.join(seqs).encode() makes two and holds both). */
int64_t pep_strs_measure(PyObject *seqs, int64_t *len_out)
{
    if (!PyList_Check(seqs)) { PyErr_SetString(PyExc_TypeError, "pep_strs_measure: list expected"); return -1; }
    const Py_ssize_t n = PyList_GET_SIZE(seqs);
    int64_t total = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *s = PyList_GET_ITEM(seqs, i);
        Py_ssize_t L;
        if (PyUnicode_Check(s)) {
            if (PyUnicode_READY(s) < 0) return -1;
            if (!PyUnicode_IS_ASCII(s)) return -2;
            L = PyUnicode_GET_LENGTH(s);
        } else if (PyBytes_Check(s)) L = PyBytes_GET_SIZE(s);
        else return -2;
        len_out[i] = (int64_t)L;
        total += (int64_t)L;
    }
    return total;
}

int pep_strs_pack(PyObject *seqs, uint8_t *out, int64_t cap)
{
    if (!PyList_Check(seqs)) { PyErr_SetString(PyExc_TypeError, "pep_strs_pack: list expected"); return -1; }
    const Py_ssize_t n = PyList_GET_SIZE(seqs);
    int64_t at = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *s = PyList_GET_ITEM(seqs, i);
        const void *src;
        Py_ssize_t L;
        if (PyUnicode_Check(s) && PyUnicode_IS_ASCII(s)) { src = PyUnicode_DATA(s); L = PyUnicode_GET_LENGTH(s); }
        else if (PyBytes_Check(s)) { src = PyBytes_AS_STRING(s); L = PyBytes_GET_SIZE(s); }
        else return -2;
        if (at + (int64_t)L > cap) { PyErr_SetString(PyExc_IndexError, "pep_strs_pack: output buffer too small"); return -1; }
        memcpy(out + at, src, (size_t)L);
        at += (int64_t)L;
    }
    return 0;
}

/* [int.from_bytes(digest[w * i : w * (i + 1)], 'big') for i in range(n)] in one C loop (five million 160-bit integers: 2 s as a comprehension) */
PyObject *pep_digest_ints(const uint8_t *digest, Py_ssize_t n, Py_ssize_t width)
{
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *v = _PyLong_FromByteArray(digest + width * i, (size_t)width, 0, 0);
        if (!v) { Py_DECREF(out); return NULL; }
        PyList_SET_ITEM(out, i, v);
    }
    return out;
}

/* {name: sequence} of the records pep_fasta_records found (peppan_amd/_native.py: fasta_records_dict): names = ASCII tokens inside `data`, sequences =
 * stretches of `codes` (ASCII).  One str per name and per sequence made straight from the buffers - the Python form decoded every name, turned all codes
 * into one 10 MB str and sliced it 10 000 times (2.4 of the 9.8 ms a fresh exemplar file cost the hot call).  Of two records with one name the
 * later counts (configure.py:118-128: the reference's dictionary assignment). */
PyObject *pep_records_dict(const uint8_t *data, const uint64_t *name_off, const uint32_t *name_len, const uint8_t *codes, const uint64_t *off, Py_ssize_t n)
{
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *name = PyUnicode_DecodeASCII((const char *)data + name_off[i], (Py_ssize_t)name_len[i], NULL);
        if (!name) { Py_DECREF(out); return NULL; }
        const Py_ssize_t len = (Py_ssize_t)(off[i + 1] - off[i]);
        PyObject *seq = PyUnicode_New(len, 127);
        if (!seq) { Py_DECREF(name); Py_DECREF(out); return NULL; }
        if (len) memcpy(PyUnicode_DATA(seq), codes + off[i], (size_t)len);
        const int rc = PyDict_SetItem(out, name, seq);
        Py_DECREF(name);
        Py_DECREF(seq);
        if (rc < 0) { Py_DECREF(out); return NULL; }
    }
    return out;
}
