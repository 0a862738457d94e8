/*
 * snf_pyfast.c - CPython extension `sniffles_amd._snf_fast`: the record table of a fetched result -> SVCall objects.
 *
 * The drop-in boundary hands Python objects to the reference's pipeline (Task.call_candidates -> list[SVCall],
 * reference src/sniffles/parallel.py:104-201, sv.py:87-223).  Once the device pass takes milliseconds, building ~10^5
 * objects with ~35 attributes each in Python is the wall clock of a task; this module builds the same objects with the C
 * API: the instance `__dict__` is filled directly from the snf_call_t records (include/sniffles_amd.h).  It is a host-side
 * formatter: no arithmetic of the hot path happens here, and sniffles_amd/sv.py keeps the pure-Python twin
 * (`materialize_candidates_py`, `apply_final_py`) that the tests compare it with.
 */
#define PY_SSIZE_T_CLEAN
#include <pthread.h>
#include <Python.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/sniffles_amd.h"

static const char* SVTYPES[7] = {"INS", "DEL", "DUP", "INV", "BND", "SINGLE_LEFT", "SINGLE_RIGHT"};
static PyObject *S_svtype[7], *S_alt_sym[7], *S_N;
static PyObject *K_contig, *K_pos, *K_id, *K_ref, *K_alt, *K_qual, *K_filter, *K_info, *K_svtype, *K_svlen, *K_end, *K_genotypes,
    *K_precise, *K_support, *K_rnames, *K_qc, *K_nm, *K_postprocess, *K_svlens, *K_fwd, *K_rev, *K_fds, *K_cov_up, *K_cov_dn,
    *K_cov_st, *K_cov_ce, *K_cov_en, *K_sample, *K_bnd_info, *K_sup_inline, *K_sup_splits, *K_raw, *K_raw_idx;
static PyObject *I_CHR2, *I_SUPPORT_LONG, *I_SUPPORT_SA, *I_STDEV_POS, *I_STDEV_LEN, *I_COVERAGE_VAR, *I_PHASE, *I_VAF;
static PyObject *F_n, *F_m1, *F_m2, *F_last;
static PyObject *S_dot, *S_comma, *O_zero, *T_none2, *I_COVERAGE;
static PyObject *B_mate_contig, *B_mate_ref_start, *B_is_first, *B_is_reverse, *P_batch, *P_index, *S_NULL, *S_PASS, *S_FAIL;
static PyObject *K_class, *K_lz, *K_lzi;

static int set_steal(PyObject* d, PyObject* k, PyObject* v) {   /* d[k] = v, steals v */
  if (!v) return -1;
  int rc = PyDict_SetItem(d, k, v);
  Py_DECREF(v);
  return rc;
}
/* raw writers: the caller has reserved the room (a local buffer, or ob_room below), `w` runs through it */
static inline char* raw_ll(char* w, long long v) {
  char t[24]; int n = 24;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { t[--n] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) t[--n] = '-';
  memcpy(w, t + n, (size_t)(24 - n));
  return w + (24 - n);
}
static inline char* raw_hex(char* w, unsigned long long u) {     /* "%llX" */
  char t[16]; int n = 16;
  do { t[--n] = "0123456789ABCDEF"[u & 15]; u >>= 4; } while (u);
  memcpy(w, t + n, (size_t)(16 - n));
  return w + (16 - n);
}
static PyObject* new_instance(PyObject* cls, PyObject* dict) {  /* object.__new__(cls) with __dict__ = dict (steals dict) */
  PyObject* empty = PyTuple_New(0);
  if (!empty) { Py_DECREF(dict); return NULL; }
  PyObject* obj = ((PyTypeObject*)cls)->tp_new((PyTypeObject*)cls, empty, NULL);
  Py_DECREF(empty);
  if (!obj) { Py_DECREF(dict); return NULL; }
  PyObject** dp = _PyObject_GetDictPtr(obj);
  if (!dp) { Py_DECREF(obj); Py_DECREF(dict); PyErr_SetString(PyExc_TypeError, "record class has no __dict__"); return NULL; }
  Py_XDECREF(*dp);
  *dp = dict;
  return obj;
}
static PyObject* name_of(PyObject* list, long i, const char* fallback_fmt) {   /* new reference */
  if (list != Py_None) {
    PyObject* s = PyList_GetItem(list, i);
    if (!s) return NULL;
    Py_INCREF(s);
    return s;
  }
  if (fallback_fmt[0] == 'q' && i >= 0) {       /* f"q{i}" without the format machinery (a dozen names per call) */
    char buf[24]; int n = 0; unsigned long x = (unsigned long)i;
    char tmp[22]; int m = 0;
    do { tmp[m++] = (char)('0' + x % 10); x /= 10; } while (x);
    buf[n++] = 'q';
    while (m) buf[n++] = tmp[--m];
    PyObject* s = PyUnicode_New(n, 127);
    if (s) memcpy(PyUnicode_1BYTE_DATA(s), buf, (size_t)n);
    return s;
  }
  return PyUnicode_FromFormat(fallback_fmt, i);
}
static PyObject* ps_str(int code, PyObject* ps_names) {       /* records._ps_str: -1 None, -2 "NULL", else the name */
  if (code == -1) { Py_RETURN_NONE; }
  if (code == -2) { Py_INCREF(S_NULL); return S_NULL; }
  if (ps_names != Py_None) {
    PyObject* s = PyList_GetItem(ps_names, code);
    if (!s) return NULL;
    Py_INCREF(s);
    return s;
  }
  return PyUnicode_FromFormat("%d", code);
}

/* materialize(svcall_cls, bnd_cls, fds_cls, post_cls | None, batch | None, calls: buffer, lo, hi, rnames: buffer(uint32),
 *             qnames: list | None, contig: str, task_id: int, contig_names: list | None, filters: list[str]) -> list */
static PyObject* py_materialize(PyObject* self, PyObject* args) {
  PyObject *cls, *bnd_cls, *fds_cls, *post_cls, *batch, *qnames, *contig, *contig_names, *filters, *targets = NULL;
  Py_buffer calls, rn, idx; idx.buf = NULL; idx.obj = NULL; idx.len = 0;
  long long lo, hi, task_id;
  /* the two optional arguments: `idx` (int64 record indices relative to lo; the calls built are those records, in that order) and
   * `targets` (existing objects - the lazy stand-ins of sv.py - that BECOME those calls: their instance dict is replaced and their
   * class set to `cls`; the list returned holds them) */
  if (!PyArg_ParseTuple(args, "OOOOOy*LLy*OOLOO|y*O", &cls, &bnd_cls, &fds_cls, &post_cls, &batch, &calls, &lo, &hi, &rn, &qnames, &contig,
                        &task_id, &contig_names, &filters, &idx, &targets))
    return NULL;
  PyObject *out = NULL, *tmpl = NULL;
  if (lo < 0 || hi < lo || (size_t)hi * sizeof(snf_call_t) > (size_t)calls.len) { PyErr_SetString(PyExc_ValueError, "call range outside the record table"); goto done; }
  const snf_call_t* C = (const snf_call_t*)calls.buf;
  const uint32_t* RN = (const uint32_t*)rn.buf;
  const long long rn_n = rn.len / 4;
  const int64_t* IDX = idx.buf ? (const int64_t*)idx.buf : NULL;
  const long long n_out = IDX ? (long long)(idx.len / 8) : hi - lo;
  if (targets == Py_None) targets = NULL;
  if (targets && (!PyList_Check(targets) || PyList_GET_SIZE(targets) != n_out)) { PyErr_SetString(PyExc_ValueError, "targets do not match the records"); goto done; }
  for (long long k = 0; IDX && k < n_out; k++) if (IDX[k] < 0 || IDX[k] >= hi - lo) { PyErr_SetString(PyExc_ValueError, "record index outside the range"); goto done; }
  out = PyList_New(n_out);
  if (!out) goto done;
  /* the instance dict of a call: 33 attributes in the dataclass's order.  A template holds the keys and the eight values that are the
   * same for every call of the task; a call's dict is a copy of it (one allocation, no inserts) with the other 25 values replaced */
  tmpl = _PyDict_NewPresized(34);
  {
    PyObject* keys33[33] = {K_contig, K_pos, K_id, K_ref, K_alt, K_qual, K_filter, K_info, K_svtype, K_svlen, K_end, K_genotypes, K_precise, K_support,
                            K_rnames, K_qc, K_nm, K_postprocess, K_svlens, K_fwd, K_rev, K_fds, K_cov_up, K_cov_dn, K_cov_st, K_cov_ce, K_cov_en, K_sample,
                            K_bnd_info, K_sup_inline, K_sup_splits, K_raw, K_raw_idx};
    for (int k = 0; tmpl && k < 33; k++) if (PyDict_SetItem(tmpl, keys33[k], keys33[k] == K_contig ? contig : keys33[k] == K_ref ? S_N : Py_None)) Py_CLEAR(tmpl);
    if (!tmpl) goto fail;
  }
  for (long long kk = 0; kk < n_out; kk++) {
    const long long i = lo + (IDX ? IDX[kk] : kk);
    const snf_call_t* c = &C[i];
    if (c->svtype < 0 || c->svtype > 6 || c->filter < 0 || c->filter >= PyList_GET_SIZE(filters)) { PyErr_SetString(PyExc_ValueError, "record field out of range"); goto fail; }
    PyObject* d = PyDict_Copy(tmpl);
    if (!d) goto fail;
    PyObject* info = _PyDict_NewPresized(6);   /* up to six keys by the end of finalize (COVERAGE_VAR, PHASE, VAF): no rehash on the way */
    PyObject* alt = S_alt_sym[c->svtype]; Py_INCREF(alt);
    PyObject* bi = Py_None; Py_INCREF(bi);
    int bad = !info;
    if (!bad && c->svtype == SNF_BND) {
      PyObject* mc = name_of(contig_names, c->mate_contig, "ctg%ld");
      bad = !mc;
      if (!bad) {
        PyObject* bd = PyDict_New();
        bad = !bd || PyDict_SetItem(bd, B_mate_contig, mc) || set_steal(bd, B_mate_ref_start, PyLong_FromLong(c->mate_ref_start)) ||
              PyDict_SetItem(bd, B_is_first, c->bnd_is_first ? Py_True : Py_False) || PyDict_SetItem(bd, B_is_reverse, c->bnd_is_reverse ? Py_True : Py_False);
        if (!bad) { Py_DECREF(bi); bi = new_instance(bnd_cls, bd); bad = !bi; if (bad) { bi = Py_None; Py_INCREF(bi); } }
        else Py_XDECREF(bd);
        if (!bad) {
          /* sv.py:630-634: ("N" if is_first else "") + br + f"{mate_contig}:{mate_ref_start}" + br + ("N" if not is_first else "") */
          const char* br = c->bnd_is_reverse ? "]" : "[";
          Py_DECREF(alt);
          alt = PyUnicode_FromFormat("%s%s%U:%d%s%s", c->bnd_is_first ? "N" : "", br, mc, (int)c->mate_ref_start, br, c->bnd_is_first ? "" : "N");
          bad = !alt || PyDict_SetItem(info, I_CHR2, mc);
          if (!alt) { alt = Py_None; Py_INCREF(alt); }
        }
        Py_DECREF(mc);
      }
    } else if (!bad && c->svtype == SNF_INS) bad = set_steal(info, I_SUPPORT_LONG, PyLong_FromLong(c->support_long));
    else if (!bad && c->svtype == SNF_DEL) bad = set_steal(info, I_SUPPORT_SA, PyLong_FromLong(c->support_sa));
    /* util.stdev returns the int 0 for fewer than two values (util.py:25-27): a single lead (fwd + rev < 2) */
    const int single = c->fwd + c->rev < 2;
    if (!bad) bad = set_steal(info, I_STDEV_POS, single ? PyLong_FromLong(0) : PyFloat_FromDouble(c->stdev_pos));
    if (!bad && !isnan(c->stdev_len)) bad = set_steal(info, I_STDEV_LEN, single ? PyLong_FromLong(0) : PyFloat_FromDouble(c->stdev_len));
    /* supporting read names */
    PyObject* names = NULL;
    if (!bad) {
      if (c->rn_off < 0 || c->rn_len < 0 || c->rn_off + c->rn_len > rn_n) { PyErr_SetString(PyExc_ValueError, "read-name range outside the table"); bad = 1; }
      else {
        names = PyList_New(c->rn_len);
        bad = !names;
        for (int k = 0; !bad && k < c->rn_len; k++) {
          PyObject* s = name_of(qnames, (long)RN[c->rn_off + k], "q%ld");
          if (!s) bad = 1; else PyList_SET_ITEM(names, k, s);
        }
      }
    }
    char idbuf[64]; Py_ssize_t idlen;      /* f"{svtype}.{sv_id:X}S{task_id:X}" (sv.py call_id) by hand: snprintf costs more than the 33 dict fills */
    {
      const size_t tl = strlen(SVTYPES[c->svtype]);
      char* w = idbuf; memcpy(w, SVTYPES[c->svtype], tl); w += tl; *w++ = '.';
      w = raw_hex(w, (unsigned)c->sv_id); *w++ = 'S'; w = raw_hex(w, (unsigned long long)task_id);
      idlen = (Py_ssize_t)(w - idbuf);
    }
    /* ForwardDifferenceWelford(): n = m1 = m2 = 0, last = None (sniffles_amd/sv.py; the test suite compares with the class) */
    PyObject* fds = NULL;
    if (!bad) {
      PyObject* fd = PyDict_New();
      PyObject* zero = PyLong_FromLong(0);
      bad = !fd || !zero || PyDict_SetItem(fd, F_n, zero) || PyDict_SetItem(fd, F_m1, zero) || PyDict_SetItem(fd, F_m2, zero) || PyDict_SetItem(fd, F_last, Py_None);
      Py_XDECREF(zero);
      if (!bad) { fds = new_instance(fds_cls, fd); bad = !fds; } else Py_XDECREF(fd);
    }
    PyObject* post = Py_None; Py_INCREF(post);
    if (!bad && post_cls != Py_None) {
      PyObject* pd = PyDict_New();
      bad = !pd || PyDict_SetItem(pd, P_batch, batch) || set_steal(pd, P_index, PyLong_FromLongLong(i - lo));
      if (!bad) { Py_DECREF(post); post = new_instance(post_cls, pd); bad = !post; if (bad) { post = Py_None; Py_INCREF(post); } }
      else Py_XDECREF(pd);
    }
    if (!bad)
      bad = set_steal(d, K_pos, PyLong_FromLong(c->pos)) || set_steal(d, K_id, PyUnicode_FromStringAndSize(idbuf, idlen)) ||
            PyDict_SetItem(d, K_alt, alt) || set_steal(d, K_qual, PyLong_FromLong(c->qual)) ||
            PyDict_SetItem(d, K_filter, PyList_GET_ITEM(filters, c->filter)) || PyDict_SetItem(d, K_info, info) ||
            PyDict_SetItem(d, K_svtype, S_svtype[c->svtype]) || set_steal(d, K_svlen, PyLong_FromLong(c->svlen)) ||
            set_steal(d, K_end, PyLong_FromLong(c->end)) || set_steal(d, K_genotypes, PyDict_New()) ||
            PyDict_SetItem(d, K_precise, c->precise ? Py_True : Py_False) || set_steal(d, K_support, PyLong_FromLong(c->support)) ||
            PyDict_SetItem(d, K_rnames, names) || PyDict_SetItem(d, K_qc, c->qc ? Py_True : Py_False) ||
            set_steal(d, K_nm, PyFloat_FromDouble(c->nm)) || (post != Py_None && PyDict_SetItem(d, K_postprocess, post)) ||
            set_steal(d, K_fwd, PyLong_FromLong(c->fwd)) || set_steal(d, K_rev, PyLong_FromLong(c->rev)) || PyDict_SetItem(d, K_fds, fds) ||
            set_steal(d, K_cov_up, PyLong_FromLong(c->cov[0])) || set_steal(d, K_cov_dn, PyLong_FromLong(c->cov[4])) ||
            set_steal(d, K_cov_st, PyLong_FromLong(c->cov[1])) || set_steal(d, K_cov_ce, PyLong_FromLong(c->cov[2])) ||
            set_steal(d, K_cov_en, PyLong_FromLong(c->cov[3])) || (bi != Py_None && PyDict_SetItem(d, K_bnd_info, bi));
    Py_XDECREF(info); Py_XDECREF(alt); Py_XDECREF(bi); Py_XDECREF(names); Py_XDECREF(fds); Py_XDECREF(post);
    if (bad) { Py_DECREF(d); goto fail; }
    PyObject* obj;
    if (targets) {       /* the stand-in becomes the call: its dict is replaced, its class set through object's own __class__ setter */
      obj = PyList_GET_ITEM(targets, kk);
      PyObject** dp = _PyObject_GetDictPtr(obj);
      if (!dp) { Py_DECREF(d); PyErr_SetString(PyExc_TypeError, "target without an instance dict"); goto fail; }
      PyObject* old = *dp; *dp = d; Py_XDECREF(old);
      if ((PyObject*)Py_TYPE(obj) != cls && PyObject_GenericSetAttr(obj, K_class, cls) != 0) goto fail;
      Py_INCREF(obj);
    } else {
      obj = new_instance(cls, d);
      if (!obj) goto fail;
    }
    PyList_SET_ITEM(out, kk, obj);
  }
  goto done;
fail:
  Py_CLEAR(out);
done:
  Py_XDECREF(tmpl);
  PyBuffer_Release(&calls); PyBuffer_Release(&rn);
  if (idx.obj) PyBuffer_Release(&idx);
  return out;
}

/* make_stubs(lazy_cls, src, records: buffer, lo, hi) -> list: the stand-ins of sv.py's lazy calls, one per record of [lo, hi): an
 * instance of `lazy_cls` whose dict holds `qc` (what `[s for s in svcalls if s.qc]`, parallel.py:267, reads), the source that can
 * turn it into the real call, and its place there.  stub_refresh_qc(calls: list, records: buffer, lo): `qc` of the final records for
 * the elements that still are stand-ins. */
static PyObject** g_idx_cache = NULL; static Py_ssize_t g_idx_cache_n = 0;      /* the ints 0 .. n-1, made once (a stand-in's place) */
static PyObject* idx_object(long long i) {      /* borrowed */
  if (i >= g_idx_cache_n) {
    Py_ssize_t n = g_idx_cache_n ? g_idx_cache_n : 4096;
    while (n <= i) n *= 2;
    PyObject** q = (PyObject**)realloc(g_idx_cache, (size_t)n * sizeof(PyObject*));
    if (!q) { PyErr_NoMemory(); return NULL; }
    g_idx_cache = q;
    for (Py_ssize_t k = g_idx_cache_n; k < n; k++) { g_idx_cache[k] = PyLong_FromSsize_t(k); if (!g_idx_cache[k]) { g_idx_cache_n = k; return NULL; } }
    g_idx_cache_n = n;
  }
  return g_idx_cache[i];
}
static PyObject* py_make_stubs(PyObject* self, PyObject* args) {
  PyObject *cls, *src; Py_buffer calls; long long lo, hi; int all_qc = 0;      /* all_qc: every stand-in starts with qc = True */
  if (!PyArg_ParseTuple(args, "OOy*LL|p", &cls, &src, &calls, &lo, &hi, &all_qc)) return NULL;
  PyObject* out = NULL;
  if (lo < 0 || hi < lo || (size_t)hi * sizeof(snf_call_t) > (size_t)calls.len) { PyErr_SetString(PyExc_ValueError, "call range outside the record table"); goto done; }
  const snf_call_t* C = (const snf_call_t*)calls.buf;
  out = PyList_New(hi - lo);
  for (long long i = lo; out && i < hi; i++) {
    PyObject* d = _PyDict_NewPresized(3);
    PyObject* ix = d ? idx_object(i - lo) : NULL;
    if (!ix || PyDict_SetItem(d, K_qc, (all_qc || C[i].qc) ? Py_True : Py_False) || PyDict_SetItem(d, K_lz, src) || PyDict_SetItem(d, K_lzi, ix)) { Py_XDECREF(d); Py_CLEAR(out); break; }
    PyObject* obj = new_instance(cls, d);
    if (!obj) { Py_CLEAR(out); break; }
    PyList_SET_ITEM(out, i - lo, obj);
  }
done:
  PyBuffer_Release(&calls);
  return out;
}
/* stub_select(stubs: list, src, mode) -> (targets: list, idx: bytes int64): the elements of `stubs` that still are stand-ins of `src`
 * (mode 1: those whose `qc` is set; mode 2: those that something besides the list `stubs` refers to) and their places.  stub_others(calls: list, src) -> list: the elements that are NOT
 * stand-ins of `src` (real calls - or stand-ins of another source, which the caller refuses), in list order. */
static PyObject* py_stub_select(PyObject* self, PyObject* args) {
  PyObject *lst, *src; int mode = 0;      /* 0: every stand-in of src; 1: those with qc set; 2: those somebody besides `stubs` holds */
  if (!PyArg_ParseTuple(args, "O!Oi", &PyList_Type, &lst, &src, &mode)) return NULL;
  const int only_qc = mode == 1;
  const Py_ssize_t n = PyList_GET_SIZE(lst);
  PyObject* targets = PyList_New(0);
  int64_t* idx = (int64_t*)malloc((size_t)(n ? n : 1) * 8);
  PyObject* out = NULL;
  Py_ssize_t m = 0;
  if (!targets || !idx) { PyErr_NoMemory(); goto done; }
  for (Py_ssize_t k = 0; k < n; k++) {
    PyObject* o = PyList_GET_ITEM(lst, k);
    PyObject** dp = _PyObject_GetDictPtr(o);
    if (!dp || !*dp) continue;
    PyObject* s = PyDict_GetItemWithError(*dp, K_lz);
    if (!s) { if (PyErr_Occurred()) goto done; continue; }
    if (s != src) continue;
    if (only_qc) { PyObject* q = PyDict_GetItemWithError(*dp, K_qc); if (!q) { if (PyErr_Occurred()) goto done; continue; } if (q != Py_True) continue; }
    if (mode == 2 && Py_REFCNT(o) <= 1) continue;
    if (PyList_Append(targets, o) != 0) goto done;
    idx[m++] = (int64_t)k;
  }
  {
    PyObject* ib = PyBytes_FromStringAndSize((const char*)idx, m * 8);
    if (ib) out = Py_BuildValue("(ON)", targets, ib);
  }
done:
  free(idx); Py_XDECREF(targets);
  return out;
}
static PyObject* py_stub_others(PyObject* self, PyObject* args) {
  PyObject *lst, *src;
  if (!PyArg_ParseTuple(args, "O!O", &PyList_Type, &lst, &src)) return NULL;
  PyObject* out = PyList_New(0);
  for (Py_ssize_t k = 0; out && k < PyList_GET_SIZE(lst); k++) {
    PyObject* o = PyList_GET_ITEM(lst, k);
    PyObject** dp = _PyObject_GetDictPtr(o);
    PyObject* s = (dp && *dp) ? PyDict_GetItemWithError(*dp, K_lz) : NULL;
    if (!s && PyErr_Occurred()) { Py_CLEAR(out); break; }
    if (s == src && src != Py_None) continue;
    if (PyList_Append(out, o) != 0) Py_CLEAR(out);
  }
  return out;
}
static PyObject* py_stub_refresh_qc(PyObject* self, PyObject* args) {
  PyObject* lst; Py_buffer calls; long long lo;
  if (!PyArg_ParseTuple(args, "O!y*L", &PyList_Type, &lst, &calls, &lo)) return NULL;
  PyObject* ret = NULL;
  const snf_call_t* C = (const snf_call_t*)calls.buf;
  const long long n_rec = (long long)(calls.len / (Py_ssize_t)sizeof(snf_call_t));
  for (Py_ssize_t k = 0; k < PyList_GET_SIZE(lst); k++) {
    PyObject** dp = _PyObject_GetDictPtr(PyList_GET_ITEM(lst, k));
    if (!dp || !*dp) continue;
    PyObject* ix = PyDict_GetItemWithError(*dp, K_lzi);
    if (!ix) { if (PyErr_Occurred()) goto done; continue; }      /* a real call: apply_final's business */
    const long long i = lo + PyLong_AsLongLong(ix);
    if (i < lo || i >= n_rec) { PyErr_SetString(PyExc_ValueError, "stand-in outside the record table"); goto done; }
    if (PyDict_SetItem(*dp, K_qc, C[i].qc ? Py_True : Py_False)) goto done;
  }
  ret = Py_None; Py_INCREF(ret);
done:
  PyBuffer_Release(&calls);
  return ret;
}

/* apply_final(calls: list, records: buffer, lo, alt_pool: buffer, ps_names: list | None, filters: list[str], early_exit: frozenset,
 *             finalize: bool = False)      finalize: also what SVCall.finalize() does (postprocess = None; sv.py:293-294) */
static PyObject* small_int_str(int v) {      /* str(v), new reference; 0..9 without the format machinery */
  if (v >= 0 && v <= 9) { PyObject* s = PyUnicode_New(1, 127); if (s) PyUnicode_1BYTE_DATA(s)[0] = (Py_UCS1)('0' + v); return s; }
  return PyUnicode_FromFormat("%d", v);
}
static PyObject* py_apply_final(PyObject* self, PyObject* args) {
  PyObject *lst, *ps_names, *filters, *early;
  Py_buffer rec, pool, idx; idx.buf = NULL; idx.obj = NULL; idx.len = 0;
  long long lo; int fin = 0;
  /* optional `idx` (int64, relative to lo): call k takes record lo + idx[k] instead of lo + k */
  if (!PyArg_ParseTuple(args, "Oy*Ly*OOO|py*", &lst, &rec, &lo, &pool, &ps_names, &filters, &early, &fin, &idx)) return NULL;
  PyObject* ret = NULL;
  const Py_ssize_t n = PyList_Size(lst);
  const int64_t* IDX = idx.buf ? (const int64_t*)idx.buf : NULL;
  const long long n_rec = (long long)(rec.len / (Py_ssize_t)sizeof(snf_call_t));
  if (n < 0 || lo < 0 || (!IDX && lo + n > n_rec) || (IDX && idx.len / 8 != n)) { PyErr_SetString(PyExc_ValueError, "calls do not match the record table"); goto done; }
  for (Py_ssize_t k = 0; IDX && k < n; k++) if (IDX[k] < 0 || lo + IDX[k] >= n_rec) { PyErr_SetString(PyExc_ValueError, "record index outside the table"); goto done; }
  const snf_call_t* C = (const snf_call_t*)rec.buf + lo;
  for (Py_ssize_t i = 0; i < n; i++) {
    const snf_call_t* c = &C[IDX ? IDX[i] : i];
    PyObject* obj = PyList_GET_ITEM(lst, i);
    PyObject** dp = _PyObject_GetDictPtr(obj);
    if (!dp || !*dp || c->filter < 0 || c->filter >= PyList_GET_SIZE(filters)) { PyErr_SetString(PyExc_TypeError, "not a materialised call"); goto done; }
    PyObject* d = *dp;
    PyObject* flt = PyList_GET_ITEM(filters, c->filter);
    PyObject* info = PyDict_GetItemWithError(d, K_info);
    PyObject* gts = PyDict_GetItemWithError(d, K_genotypes);
    if (!info || !gts) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "call without info / genotypes"); goto done; }
    if (PyDict_SetItem(d, K_qc, c->qc ? Py_True : Py_False) || PyDict_SetItem(d, K_filter, flt)) goto done;
    if (fin && PyDict_SetItem(d, K_postprocess, Py_None)) goto done;
    int ee = PySet_Contains(early, flt);
    if (ee < 0) goto done;
    if (!ee && PyDict_SetItem(info, I_COVERAGE_VAR, Py_None)) goto done;     /* see sv.fill_final */
    if (c->ph_set) {
      /* f"{hp},{ps},{hp_support},{ps_support},{hp_filter},{ps_filter}" (postprocessing.py:651) - every call carries one: written by hand
       * when the phase-set name is ASCII (PyUnicode_FromFormat takes ~0.5 us a call) */
      PyObject* ps = ps_str(c->ph_ps, ps_names);
      if (!ps) goto done;
      PyObject* s = NULL;
      const char* pu = NULL; Py_ssize_t pl = 0;
      if (ps == Py_None) { pu = "None"; pl = 4; }
      else if (PyUnicode_Check(ps) && PyUnicode_IS_COMPACT_ASCII(ps)) pu = PyUnicode_AsUTF8AndSize(ps, &pl);
      if (pu && pl <= 160) {
        char buf[256]; char* w = raw_ll(buf, (int)c->ph_hp);
        *w++ = ','; memcpy(w, pu, (size_t)pl); w += pl; *w++ = ',';
        w = raw_ll(w, (int)c->ph_hp_support); *w++ = ','; w = raw_ll(w, (int)c->ph_ps_support);
        memcpy(w, c->ph_hp_pass ? ",PASS" : ",FAIL", 5); w += 5; memcpy(w, c->ph_ps_pass ? ",PASS" : ",FAIL", 5); w += 5;
        s = PyUnicode_New(w - buf, 127);
        if (s) memcpy(PyUnicode_1BYTE_DATA(s), buf, (size_t)(w - buf));
      } else {
        PyErr_Clear();
        s = PyUnicode_FromFormat("%d,%S,%d,%d,%s,%s", (int)c->ph_hp, ps, (int)c->ph_hp_support, (int)c->ph_ps_support,
                                 c->ph_hp_pass ? "PASS" : "FAIL", c->ph_ps_pass ? "PASS" : "FAIL");
      }
      Py_DECREF(ps);
      if (set_steal(info, I_PHASE, s)) goto done;
    }
    if (c->gt_set) {
      PyObject* hp = c->gt_hp < 0 ? (Py_INCREF(Py_None), Py_None) : small_int_str((int)c->gt_hp);
      PyObject* ps = ps_str(c->gt_ps, ps_names);
      /* (a, b, gq, dr, dv, (hp, ps)) built by hand: Py_BuildValue parses its format string for every call */
      PyObject* t = (hp && ps) ? PyTuple_New(6) : NULL; PyObject* ph = t ? PyTuple_New(2) : NULL;
      int bad = !t || !ph;
      if (!bad) {
        PyTuple_SET_ITEM(ph, 0, hp); PyTuple_SET_ITEM(ph, 1, ps); hp = ps = NULL;
        const int v5[5] = {(int)c->gt_a, (int)c->gt_b, (int)c->gt_gq, (int)c->gt_dr, (int)c->gt_dv};
        for (int q = 0; q < 5 && !bad; q++) { PyObject* x = PyLong_FromLong(v5[q]); if (!x) bad = 1; else PyTuple_SET_ITEM(t, q, x); }
        PyTuple_SET_ITEM(t, 5, ph); ph = NULL;
      }
      Py_XDECREF(hp); Py_XDECREF(ps); Py_XDECREF(ph);
      bad = bad || PyDict_SetItem(gts, O_zero, t);
      Py_XDECREF(t);
      if (bad || set_steal(info, I_VAF, PyFloat_FromDouble(c->vaf))) goto done;
    }
    if (c->alt_len >= 0) {
      if (c->alt_off < 0 || c->alt_off + c->alt_len > pool.len) { PyErr_SetString(PyExc_ValueError, "ALT range outside the pool"); goto done; }
      if (set_steal(d, K_alt, PyUnicode_DecodeLatin1((const char*)pool.buf + c->alt_off, c->alt_len, NULL))) goto done;
    }
  }
  ret = Py_None; Py_INCREF(ret);
done:
  PyBuffer_Release(&rec); PyBuffer_Release(&pool);
  if (idx.obj) PyBuffer_Release(&idx);
  return ret;
}


/* ======================================================================================================================
 * Columnar candidate store of the multi-sample combine (sniffles_amd/parallel.py::CombineTask).
 * collect():       SVCall objects of the SNF blocks -> snf_group_cand_t records + ALT pool + BND mate columns (one pass)
 * flush_windows(): the bin walk of CombineTask.execute (parallel.py:516-534) over the sorted table
 * group_calls():   snf_group_out_t records + membership -> the combined SVCall objects of SVGroup.call (sv.py:419-481)
 * None of this is arithmetic of the hot path: attribute reads, string joins, dict fills.
 * ====================================================================================================================== */

static PyObject* aget(PyObject* obj, PyObject* key) {   /* new reference; instance dict first */
  PyObject** dp = _PyObject_GetDictPtr(obj);
  if (dp && *dp) {
    PyObject* v = PyDict_GetItemWithError(*dp, key);
    if (v) { Py_INCREF(v); return v; }
    if (PyErr_Occurred()) return NULL;
  }
  return PyObject_GetAttr(obj, key);
}
static int as_i32(PyObject* v, int32_t* out, int none_ok) {   /* steals v */
  if (!v) return -1;
  int rc = 0;
  if (v == Py_None) { if (none_ok) *out = SNF_NONE_I32; else { PyErr_SetString(PyExc_TypeError, "candidate field is None"); rc = -1; } }
  else if (PyFloat_Check(v)) *out = (int32_t)PyFloat_AS_DOUBLE(v);          /* int(x) truncates */
  else { long x = PyLong_AsLong(v); if (x == -1 && PyErr_Occurred()) rc = -1; else *out = (int32_t)x; }
  Py_DECREF(v);
  return rc;
}
static int gt_code(PyObject* v, int8_t* out) {   /* borrowed: "." -> -1 */
  if (PyUnicode_Check(v)) { *out = -1; return 0; }
  long x = PyLong_AsLong(v);
  if (x == -1 && PyErr_Occurred()) return -1;
  *out = (int8_t)x;
  return 0;
}
static int is_str(PyObject* v, PyObject* interned, const char* text) {   /* borrowed */
  return v == interned || (PyUnicode_Check(v) && PyUnicode_CompareWithASCIIString(v, text) == 0);
}

/* collect(blocks: list (per block index: list per reader of `read_blocks(contig, index)` = list of block dicts | None),
 *         sids: buffer int32 (per reader), types: tuple[str], support_threshold: int, mate_ids: dict)
 *   -> (objs: list, records: bytearray, cand_block: bytearray int32, cand_type: bytearray int32, mate: bytearray int32 x 2,
 *       alt_off: bytes int64 (n + 1), alt_pool: bytes)
 * Candidates leave in the order CombineTask.execute visits them (parallel.py:497-512): block, SV type, reader, block part,
 * list order; those below the support threshold are dropped; `sample_internal_id` is set on the objects as the reference does. */
static PyObject* py_collect(PyObject* self, PyObject* args) {
  PyObject *blocks, *types, *mate_ids;
  Py_buffer sidb;
  long long thr;
  if (!PyArg_ParseTuple(args, "O!y*O!LO!", &PyList_Type, &blocks, &sidb, &PyTuple_Type, &types, &thr, &PyDict_Type, &mate_ids)) return NULL;
  PyObject *ret = NULL, *objs = NULL, *rec = NULL, *cblk = NULL, *ctyp = NULL, *mate = NULL, **sid_objs = NULL;
  int64_t* aoff = NULL; uint8_t* pool = NULL; size_t pool_cap = 0, pool_n = 0;
  Py_ssize_t cap = 0, n = 0;
  const Py_ssize_t nblk = PyList_GET_SIZE(blocks), ntyp = PyTuple_GET_SIZE(types), nsid = sidb.len / 4;
  const int32_t* SID = (const int32_t*)sidb.buf;
  objs = PyList_New(0);
  rec = PyByteArray_FromStringAndSize(NULL, 0); cblk = PyByteArray_FromStringAndSize(NULL, 0); ctyp = PyByteArray_FromStringAndSize(NULL, 0);
  mate = PyByteArray_FromStringAndSize(NULL, 0);
  sid_objs = (PyObject**)calloc((size_t)nsid + 1, sizeof(PyObject*));
  if (!objs || !rec || !cblk || !ctyp || !mate || !sid_objs) { if (!PyErr_Occurred()) PyErr_NoMemory(); goto done; }
  for (Py_ssize_t i = 0; i < nsid; i++) { sid_objs[i] = PyLong_FromLong(SID[i]); if (!sid_objs[i]) goto done; }
  for (Py_ssize_t eb = 0; eb < nblk; eb++) {
    PyObject* per = PyList_GET_ITEM(blocks, eb);
    if (!PyList_Check(per) || PyList_GET_SIZE(per) != nsid) { PyErr_SetString(PyExc_TypeError, "collect: one entry per reader expected"); goto done; }
    for (Py_ssize_t ty = 0; ty < ntyp; ty++) {
      PyObject* tkey = PyTuple_GET_ITEM(types, ty);
      for (Py_ssize_t si = 0; si < nsid; si++) {
        PyObject* parts = PyList_GET_ITEM(per, si);
        if (parts == Py_None) continue;
        if (!PyList_Check(parts)) { PyErr_SetString(PyExc_TypeError, "read_blocks must return a list or None"); goto done; }
        for (Py_ssize_t pi = 0; pi < PyList_GET_SIZE(parts); pi++) {
          PyObject* l = PyObject_GetItem(PyList_GET_ITEM(parts, pi), tkey);      /* block[svtype] */
          if (!l) goto done;
          if (!PyList_Check(l)) { Py_DECREF(l); PyErr_SetString(PyExc_TypeError, "a block's candidates must be a list"); goto done; }
          const Py_ssize_t nl = PyList_GET_SIZE(l);
          if (n + nl > cap) {
            cap = (n + nl) * 2 + 1024;
            int64_t* na = (int64_t*)realloc(aoff, ((size_t)cap + 1) * sizeof(int64_t));
            if (!na || PyByteArray_Resize(rec, cap * (Py_ssize_t)sizeof(snf_group_cand_t)) || PyByteArray_Resize(cblk, cap * 4) ||
                PyByteArray_Resize(ctyp, cap * 4) || PyByteArray_Resize(mate, cap * 8)) { if (na) aoff = na; if (!PyErr_Occurred()) PyErr_NoMemory(); Py_DECREF(l); goto done; }
            aoff = na;
          }
          snf_group_cand_t* R = (snf_group_cand_t*)PyByteArray_AS_STRING(rec);
          int32_t* CB = (int32_t*)PyByteArray_AS_STRING(cblk); int32_t* CT = (int32_t*)PyByteArray_AS_STRING(ctyp);
          int32_t* MT = (int32_t*)PyByteArray_AS_STRING(mate);
          for (Py_ssize_t q = 0; q < nl; q++) {
            PyObject* c = PyList_GET_ITEM(l, q);
            int32_t support;
            if (as_i32(aget(c, K_support), &support, 0)) { Py_DECREF(l); goto done; }
            if (support < thr) continue;
            snf_group_cand_t* r = &R[n];
            memset(r, 0, sizeof *r);
            r->support = support; r->sample = SID[si];
            int bad = PyObject_SetAttr(c, K_sample, sid_objs[si]) ||                 /* parallel.py:509 */
                      as_i32(aget(c, K_pos), &r->pos, 0) || as_i32(aget(c, K_svlen), &r->svlen, 0) || as_i32(aget(c, K_end), &r->end, 0) ||
                      as_i32(aget(c, K_qual), &r->qual, 1) || as_i32(aget(c, K_fwd), &r->fwd, 0) || as_i32(aget(c, K_rev), &r->rev, 0) ||
                      as_i32(aget(c, K_cov_up), &r->cov[0], 1) || as_i32(aget(c, K_cov_st), &r->cov[1], 1) || as_i32(aget(c, K_cov_ce), &r->cov[2], 1) ||
                      as_i32(aget(c, K_cov_en), &r->cov[3], 1) || as_i32(aget(c, K_cov_dn), &r->cov[4], 1);
            PyObject* v = NULL;
            if (!bad) { v = aget(c, K_qc); bad = !v; if (v) { int t = PyObject_IsTrue(v); bad = t < 0; r->qc = t > 0; Py_DECREF(v); } }
            if (!bad) { v = aget(c, K_precise); bad = !v; if (v) { int t = PyObject_IsTrue(v); bad = t < 0; r->precise = t > 0; Py_DECREF(v); } }
            if (!bad) { v = aget(c, K_filter); bad = !v; if (v) { r->pass = is_str(v, S_PASS, "PASS"); Py_DECREF(v); } }
            int is_bnd = 0;
            if (!bad) { v = aget(c, K_svtype); bad = !v; if (v) { r->is_ins = is_str(v, S_svtype[0], "INS"); is_bnd = is_str(v, S_svtype[4], "BND"); Py_DECREF(v); } }
            if (!bad) {                                                              /* genotypes[0], or the default of sv.py:392 */
              v = aget(c, K_genotypes); bad = !v;
              if (v) {
                PyObject* t = PyDict_Check(v) ? PyDict_GetItemWithError(v, O_zero) : NULL;
                if (t && PyTuple_Check(t) && PyTuple_GET_SIZE(t) >= 5) {
                  bad = gt_code(PyTuple_GET_ITEM(t, 0), &r->gt_a) || gt_code(PyTuple_GET_ITEM(t, 1), &r->gt_b);
                  Py_INCREF(PyTuple_GET_ITEM(t, 2)); Py_INCREF(PyTuple_GET_ITEM(t, 3)); Py_INCREF(PyTuple_GET_ITEM(t, 4));
                  bad = as_i32(PyTuple_GET_ITEM(t, 2), &r->gq, 0) | as_i32(PyTuple_GET_ITEM(t, 3), &r->dr, 0) | as_i32(PyTuple_GET_ITEM(t, 4), &r->dv, 0) | bad;
                } else if (PyErr_Occurred()) bad = 1;
                else { r->gt_a = r->gt_b = -1; r->gq = 0; r->dr = 0; r->dv = support; }
                Py_DECREF(v);
              }
            }
            MT[2 * n] = 0; MT[2 * n + 1] = 0;
            if (!bad && is_bnd) {
              PyObject* bi = aget(c, K_bnd_info); bad = !bi;
              if (bi) {
                PyObject* mc = aget(bi, B_mate_contig); bad = !mc;
                if (mc) {
                  PyObject* idv = PyDict_GetItemWithError(mate_ids, mc);       /* equality-only id of the mate contig */
                  if (!idv && !PyErr_Occurred()) { PyObject* nv = PyLong_FromSsize_t(PyDict_GET_SIZE(mate_ids)); if (nv && !PyDict_SetItem(mate_ids, mc, nv)) idv = PyDict_GetItemWithError(mate_ids, mc); Py_XDECREF(nv); }
                  if (!idv) bad = 1; else MT[2 * n] = (int32_t)PyLong_AsLong(idv);
                  Py_DECREF(mc);
                }
                if (!bad) bad = as_i32(aget(bi, B_mate_ref_start), &MT[2 * n + 1], 0);
                Py_DECREF(bi);
              }
            }
            if (!bad) {                                                              /* ALT: latin-1 bytes, len(c.alt) code points */
              v = aget(c, K_alt); bad = !v;
              if (v) {
                const char* data = NULL; Py_ssize_t len = 0; PyObject* enc = NULL;
                if (PyUnicode_Check(v)) {
                  if (PyUnicode_READY(v)) bad = 1;
                  else if (PyUnicode_KIND(v) == PyUnicode_1BYTE_KIND) { data = (const char*)PyUnicode_1BYTE_DATA(v); len = PyUnicode_GET_LENGTH(v); }
                  else { enc = PyUnicode_AsLatin1String(v); bad = !enc; if (enc) { data = PyBytes_AS_STRING(enc); len = PyBytes_GET_SIZE(enc); } }
                } else if (PyBytes_Check(v)) { data = PyBytes_AS_STRING(v); len = PyBytes_GET_SIZE(v); }
                else { PyErr_SetString(PyExc_TypeError, "SVCall.alt must be str or bytes"); bad = 1; }
                if (!bad) {
                  if (pool_n + (size_t)len + 64 > pool_cap) {
                    pool_cap = (pool_n + (size_t)len + 64) * 2;
                    uint8_t* np_ = (uint8_t*)realloc(pool, pool_cap);
                    if (!np_) { PyErr_NoMemory(); bad = 1; } else pool = np_;
                  }
                  if (!bad) { memcpy(pool + pool_n, data, (size_t)len); pool_n += (size_t)len; r->alt_len = (int32_t)len; }
                }
                Py_XDECREF(enc); Py_DECREF(v);
              }
            }
            if (bad || PyList_Append(objs, c)) { Py_DECREF(l); goto done; }
            CB[n] = (int32_t)eb; CT[n] = (int32_t)ty;
            if (n == 0) aoff[0] = 0;
            aoff[n + 1] = (int64_t)pool_n;
            n++;
          }
          Py_DECREF(l);
        }
      }
    }
  }
  if (PyByteArray_Resize(rec, n * (Py_ssize_t)sizeof(snf_group_cand_t)) || PyByteArray_Resize(cblk, n * 4) || PyByteArray_Resize(ctyp, n * 4) ||
      PyByteArray_Resize(mate, n * 8)) goto done;
  {
    int64_t zero = 0;
    PyObject* ao = PyBytes_FromStringAndSize(n ? (const char*)aoff : (const char*)&zero, ((Py_ssize_t)n + 1) * 8);
    PyObject* ap = PyBytes_FromStringAndSize(pool ? (const char*)pool : "", (Py_ssize_t)pool_n);
    if (ao && ap) ret = PyTuple_Pack(7, objs, rec, cblk, ctyp, mate, ao, ap);
    Py_XDECREF(ao); Py_XDECREF(ap);
  }
done:
  Py_XDECREF(objs); Py_XDECREF(rec); Py_XDECREF(cblk); Py_XDECREF(ctyp); Py_XDECREF(mate);
  if (sid_objs) { for (Py_ssize_t i = 0; i < nsid; i++) Py_XDECREF(sid_objs[i]); free(sid_objs); }
  free(aoff); free(pool);
  PyBuffer_Release(&sidb);
  return ret;
}

/* gather_pool(off: buffer int64 (n + 1), pool: buffer, order: buffer int64 (m)) -> (new_off: bytes int64 (m + 1), new_pool: bytes):
 * the strings order[0], order[1], ... back to back */
static PyObject* py_gather_pool(PyObject* self, PyObject* args) {
  Py_buffer ofb, plb, orb;
  if (!PyArg_ParseTuple(args, "y*y*y*", &ofb, &plb, &orb)) return NULL;
  PyObject* ret = NULL;
  const int64_t* OFF = (const int64_t*)ofb.buf; const Py_ssize_t n = ofb.len / 8 - 1;
  const int64_t* ORD = (const int64_t*)orb.buf; const Py_ssize_t m = orb.len / 8;
  int64_t* no = (int64_t*)malloc(((size_t)m + 1) * 8);
  size_t total = 0;
  if (!no) { PyErr_NoMemory(); goto done; }
  no[0] = 0;
  for (Py_ssize_t i = 0; i < m; i++) {
    if (ORD[i] < 0 || ORD[i] >= n || OFF[ORD[i] + 1] < OFF[ORD[i]] || OFF[ORD[i] + 1] > plb.len) { PyErr_SetString(PyExc_ValueError, "gather_pool: index or offset out of range"); goto done; }
    total += (size_t)(OFF[ORD[i] + 1] - OFF[ORD[i]]);
    no[i + 1] = (int64_t)total;
  }
  {
    PyObject* np_ = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (np_) {
      char* w = PyBytes_AS_STRING(np_);
      for (Py_ssize_t i = 0; i < m; i++) { const size_t l = (size_t)(OFF[ORD[i] + 1] - OFF[ORD[i]]); memcpy(w, (const char*)plb.buf + OFF[ORD[i]], l); w += l; }
      PyObject* oo = PyBytes_FromStringAndSize((const char*)no, ((Py_ssize_t)m + 1) * 8);
      if (oo) ret = PyTuple_Pack(2, oo, np_);
      Py_XDECREF(oo); Py_DECREF(np_);
    }
  }
done:
  free(no);
  PyBuffer_Release(&ofb); PyBuffer_Release(&plb); PyBuffer_Release(&orb);
  return ret;
}

/* gather_pool_parts(pools: list of buffers, part: buffer int32 (pool of every string), start: buffer int64 (its first byte there),
 *                   length: buffer int64, order: buffer int64) -> (offsets: bytes int64 (len(order) + 1), pool: bytes)
 * gather_pool over strings that still lie in the pools of their tables (one per reader and contig): the merge's ALT pool is written
 * once, in sorted order, instead of being concatenated first and permuted then. */
/* argsort_i64(keys: buffer int64) -> bytes int64: the stable ascending order of the keys (numpy.argsort(kind="stable")) by a radix sort,
 * eleven bits a pass, passes in which every key has the same digit skipped - the merge sorts its candidate table by ONE 63-bit key
 * (task, SV type, block, bin), whose upper digits are mostly equal */
static PyObject* py_argsort_i64(PyObject* self, PyObject* args) {
  Py_buffer kb;
  if (!PyArg_ParseTuple(args, "y*", &kb)) return NULL;
  const Py_ssize_t n = kb.len / 8;
  PyObject* ret = NULL;
  uint64_t *k0 = (uint64_t*)malloc((size_t)n * 8 + 8), *k1 = (uint64_t*)malloc((size_t)n * 8 + 8);
  int64_t *i0 = (int64_t*)malloc((size_t)n * 8 + 8), *i1 = (int64_t*)malloc((size_t)n * 8 + 8);
  if (!k0 || !k1 || !i0 || !i1) { PyErr_NoMemory(); goto done; }
  Py_BEGIN_ALLOW_THREADS
  for (Py_ssize_t i = 0; i < n; i++) { k0[i] = ((const uint64_t*)kb.buf)[i] ^ 0x8000000000000000ull; i0[i] = i; }      /* signed order as unsigned order */
  for (int pass = 0; pass < 6; pass++) {
    const int shift = 11 * pass;
    size_t hist[2048]; memset(hist, 0, sizeof hist);
    for (Py_ssize_t i = 0; i < n; i++) hist[(k0[i] >> shift) & 2047]++;
    int single = 0;
    for (int d = 0; d < 2048; d++) if (hist[d] == (size_t)n) single = 1;
    if (single || n == 0) continue;
    size_t at = 0;
    for (int d = 0; d < 2048; d++) { const size_t c = hist[d]; hist[d] = at; at += c; }
    for (Py_ssize_t i = 0; i < n; i++) { const size_t to = hist[(k0[i] >> shift) & 2047]++; k1[to] = k0[i]; i1[to] = i0[i]; }
    { uint64_t* tk = k0; k0 = k1; k1 = tk; int64_t* ti = i0; i0 = i1; i1 = ti; }
  }
  Py_END_ALLOW_THREADS
  ret = PyBytes_FromStringAndSize((const char*)i0, n * 8);
done:
  free(k0); free(k1); free(i0); free(i1);
  PyBuffer_Release(&kb);
  return ret;
}

typedef struct { const Py_buffer* pb; const int32_t* PT; const int64_t *ST, *LN, *ORD, *no; char* w; Py_ssize_t i0, i1; } GatherJob;
static void* gather_thread(void* arg) {
  const GatherJob* j = (const GatherJob*)arg;
  for (Py_ssize_t i = j->i0; i < j->i1; i++) { const int64_t c = j->ORD[i]; memcpy(j->w + j->no[i], (const char*)j->pb[j->PT[c]].buf + j->ST[c], (size_t)j->LN[c]); }
  return NULL;
}
static PyObject* py_gather_pool_parts(PyObject* self, PyObject* args) {
  PyObject* pools; Py_buffer ptb, stb, lnb, orb;
  if (!PyArg_ParseTuple(args, "O!y*y*y*y*", &PyList_Type, &pools, &ptb, &stb, &lnb, &orb)) return NULL;
  PyObject* ret = NULL;
  const Py_ssize_t np_ = PyList_GET_SIZE(pools), n = ptb.len / 4, m = orb.len / 8;
  const int32_t* PT = (const int32_t*)ptb.buf; const int64_t* ST = (const int64_t*)stb.buf; const int64_t* LN = (const int64_t*)lnb.buf;
  const int64_t* ORD = (const int64_t*)orb.buf;
  Py_buffer* pb = (Py_buffer*)calloc((size_t)np_ + 1, sizeof(Py_buffer)); Py_ssize_t got = 0;
  int64_t* no = (int64_t*)malloc(((size_t)m + 1) * 8);
  size_t total = 0;
  if (!pb || !no) { PyErr_NoMemory(); goto done; }
  if (stb.len / 8 < n || lnb.len / 8 < n) { PyErr_SetString(PyExc_ValueError, "gather_pool_parts: columns of different lengths"); goto done; }
  for (; got < np_; got++) if (PyObject_GetBuffer(PyList_GET_ITEM(pools, got), &pb[got], PyBUF_SIMPLE) != 0) goto done;
  no[0] = 0;
  for (Py_ssize_t i = 0; i < m; i++) {
    const int64_t c = ORD[i];
    if (c < 0 || c >= n || PT[c] < 0 || PT[c] >= np_ || ST[c] < 0 || LN[c] < 0 || ST[c] + LN[c] > pb[PT[c]].len) {
      PyErr_SetString(PyExc_ValueError, "gather_pool_parts: index or range out of bounds"); goto done; }
    total += (size_t)LN[c];
    no[i + 1] = (int64_t)total;
  }
  {
    PyObject* out = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (out) {
      /* the copies read cold memory all over the tables' pools: four threads share them when there are megabytes to move */
      GatherJob jobs[8]; pthread_t tids[8]; int started[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int nt = total >= ((size_t)32 << 20) && m >= 8192 ? 8 : total >= ((size_t)4 << 20) && m >= 4096 ? 4 : 1;
      for (int t = 0; t < nt; t++) {
        jobs[t].pb = pb; jobs[t].PT = PT; jobs[t].ST = ST; jobs[t].LN = LN; jobs[t].ORD = ORD; jobs[t].no = no; jobs[t].w = PyBytes_AS_STRING(out);
      }
      /* equal BYTES per thread (the sorted table keeps the long strings - insertions - together): cut where the output offset passes
       * t / nt of the total */
      jobs[0].i0 = 0;
      for (int t = 1; t < nt; t++) {
        const int64_t want = (int64_t)(total / (size_t)nt) * t;
        Py_ssize_t lo_ = jobs[t - 1].i0, hi_ = m;
        while (lo_ < hi_) { const Py_ssize_t mid = lo_ + (hi_ - lo_) / 2; if (no[mid] < want) lo_ = mid + 1; else hi_ = mid; }
        jobs[t].i0 = lo_; jobs[t - 1].i1 = lo_;
      }
      jobs[nt - 1].i1 = m;
      Py_BEGIN_ALLOW_THREADS
      for (int t = 1; t < nt; t++) started[t] = pthread_create(&tids[t], NULL, gather_thread, &jobs[t]) == 0;
      gather_thread(&jobs[0]);
      for (int t = 1; t < nt; t++) { if (started[t]) pthread_join(tids[t], NULL); else gather_thread(&jobs[t]); }
      Py_END_ALLOW_THREADS
      PyObject* oo = PyBytes_FromStringAndSize((const char*)no, ((Py_ssize_t)m + 1) * 8);
      if (oo) ret = PyTuple_Pack(2, oo, out);
      Py_XDECREF(oo); Py_DECREF(out);
    }
  }
done:
  if (pb) for (Py_ssize_t k = 0; k < got; k++) PyBuffer_Release(&pb[k]);
  free(pb); free(no);
  PyBuffer_Release(&ptb); PyBuffer_Release(&stb); PyBuffer_Release(&lnb); PyBuffer_Release(&orb);
  return ret;
}

/* flush_windows(key: buffer int64, bin: buffer int32, bin_min_size, max_candidates, exhaustive)
 *   -> (win_end: bytes int64 - one past the last candidate of every window, win_bin: bytes int32, win_size: bytes int32)
 * key / bin: the candidate table sorted by (key, bin); one key = one (task, SV type, block).  parallel.py:516-534: the bins of a
 * block are visited in ascending order, `size` grows by bin_min_size per bin, a window is flushed when it holds at least
 * max_candidates candidates (unless --combine-exhaustive) or at the block's last bin. */
static PyObject* py_flush_windows(PyObject* self, PyObject* args) {
  Py_buffer kb, bb;
  long long bin_min_size, max_cands; int exhaustive;
  if (!PyArg_ParseTuple(args, "y*y*LLp", &kb, &bb, &bin_min_size, &max_cands, &exhaustive)) return NULL;
  PyObject* ret = NULL;
  const Py_ssize_t n = kb.len / 8;
  const int64_t* K = (const int64_t*)kb.buf; const int32_t* B = (const int32_t*)bb.buf;
  int64_t* wend = (int64_t*)malloc(((size_t)n + 1) * 8); int32_t* wbin = (int32_t*)malloc(((size_t)n + 1) * 4); int32_t* wsize = (int32_t*)malloc(((size_t)n + 1) * 4);
  if (bb.len / 4 != n || !wend || !wbin || !wsize) { PyErr_SetString(PyExc_ValueError, "flush_windows: bad arguments"); goto done; }
  Py_ssize_t nw = 0, i = 0;
  while (i < n) {
    Py_ssize_t j = i;                                  /* [i, j): one (task, type, block) */
    while (j < n && K[j] == K[i]) j++;
    long long size = 0, cnt = 0;
    Py_ssize_t a = i;
    while (a < j) {
      Py_ssize_t b = a;                                /* [a, b): one bin */
      while (b < j && B[b] == B[a]) b++;
      cnt += b - a; size += bin_min_size;
      if ((!exhaustive && cnt >= max_cands) || b == j) { wend[nw] = b; wbin[nw] = B[a]; wsize[nw] = (int32_t)size; nw++; size = 0; cnt = 0; }
      a = b;
    }
    i = j;
  }
  {
    PyObject* o1 = PyBytes_FromStringAndSize((const char*)wend, nw * 8);
    PyObject* o2 = PyBytes_FromStringAndSize((const char*)wbin, nw * 4);
    PyObject* o3 = PyBytes_FromStringAndSize((const char*)wsize, nw * 4);
    if (o1 && o2 && o3) ret = PyTuple_Pack(3, o1, o2, o3);
    Py_XDECREF(o1); Py_XDECREF(o2); Py_XDECREF(o3);
  }
done:
  free(wend); free(wbin); free(wsize);
  PyBuffer_Release(&kb); PyBuffer_Release(&bb);
  return ret;
}

static PyObject* int_or_none(int32_t v) { if (v == SNF_NONE_I32) { Py_RETURN_NONE; } return PyLong_FromLong(v); }

/* group_calls(svcall_cls, fds_cls, objs: list, out: buffer snf_group_out_t, emit: buffer int64 (groups in emission order),
 *             group_off: buffer int64, member: buffer int32, chosen: buffer uint8, sv_id: buffer int64, task_id: buffer int64,
 *             sample_ids: buffer int32 (config.snf_input_info order), sample_pos: buffer int32 (internal id -> position, -1),
 *             block_cov: list (per block: list per sample position of the block's _COVERAGE dict or None),
 *             ev_off: buffer int64 (len(emit) + 1), ev_block: buffer int32, ev_bin: buffer int32,
 *             null_min_coverage: int, id_prefix: str, single_sample: bool, cand_sample: buffer int32) -> list[SVCall] */
typedef struct { char* p; size_t n, cap; } OutBuf;
static int ob_room(OutBuf* b, size_t extra) {
  if (b->n + extra <= b->cap) return 0;
  size_t cap = b->cap ? b->cap : 1 << 16;
  while (cap < b->n + extra) cap *= 2;
  char* q = (char*)realloc(b->p, cap);
  if (!q) { PyErr_NoMemory(); return -1; }
  b->p = q; b->cap = cap;
  return 0;
}
static inline int ob_put(OutBuf* b, const char* s, size_t n) {
  if (__builtin_expect(b->n + n > b->cap, 0) && ob_room(b, n)) return -1;
  memcpy(b->p + b->n, s, n); b->n += n; return 0;
}
static int ob_str(OutBuf* b, const char* s) { return ob_put(b, s, strlen(s)); }
/* a string LITERAL: its length is known to the compiler, the copy becomes a few moves (a merged record is ~300 such pieces) */
#define OB_LIT(b, lit) ob_put((b), "" lit, sizeof(lit) - 1)
static inline int ob_ull(OutBuf* b, unsigned long long u, int neg) {     /* decimal digits by hand: snprintf costs ~100 ns each */
  if (__builtin_expect(b->n + 24 > b->cap, 0) && ob_room(b, 24)) return -1;
  char t[24]; int n = 24;
  do { t[--n] = (char)('0' + u % 10); u /= 10; } while (u);
  if (neg) t[--n] = '-';
  memcpy(b->p + b->n, t + n, (size_t)(24 - n)); b->n += (size_t)(24 - n);
  return 0;
}
static inline int ob_ll(OutBuf* b, long long v) { return ob_ull(b, v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v, v < 0); }
static int ob_hex(OutBuf* b, unsigned long long u) {     /* "%llX" */
  char t[16]; int n = 16;
  do { t[--n] = "0123456789ABCDEF"[u & 15]; u >>= 4; } while (u);
  return ob_put(b, t + n, (size_t)(16 - n));
}
/* f"{v:.3f}" = "%.3f": the exactly rounded decimal (ties of the binary value to even, as printf and Python print it).  printf's own
 * conversion costs ~0.3 us a value; below 10^12 the thousandths are one multiplication, with the product's rounding error taken from
 * an fma (v * 1000 = s + e exactly): the fraction d of s is a multiple of ulp(s) and |e| <= ulp(s) / 2, so d alone decides unless it is
 * exactly one half, where the sign of e does - and e == 0 is a true tie. */
static int ob_f3(OutBuf* b, double v) {
  if (isnan(v)) return OB_LIT(b, "nan");
  if (isinf(v)) return v < 0 ? OB_LIT(b, "-inf") : OB_LIT(b, "inf");
  const double a = fabs(v);
  if (a < 1e12) {
    const double s = a * 1000.0, e = fma(a, 1000.0, -s), f = floor(s), d = s - f;
    unsigned long long r = (unsigned long long)f;
    if (d > 0.5 || (d == 0.5 && (e > 0.0 || (e == 0.0 && (r & 1))))) r++;
    if (ob_ull(b, r / 1000, signbit(v) != 0)) return -1;
    const unsigned m = (unsigned)(r % 1000);
    const char t[4] = {'.', (char)('0' + m / 100), (char)('0' + m / 10 % 10), (char)('0' + m % 10)};
    return ob_put(b, t, 4);
  }
  char t[352]; int n = snprintf(t, sizeof t, "%.3f", v); return ob_put(b, t, (size_t)n);
}
static int ob_py(OutBuf* b, PyObject* s) {   /* a str */
  Py_ssize_t n; const char* u = PyUnicode_AsUTF8AndSize(s, &n);
  return u ? ob_put(b, u, (size_t)n) : -1;
}
static int ob_name(OutBuf* b, PyObject* list, long i, const char* prefix) {   /* list[i] or f"{prefix}{i}" */
  if (list != Py_None) { PyObject* s = PyList_GetItem(list, i); return s ? ob_py(b, s) : -1; }
  if (ob_str(b, prefix)) return -1;
  return ob_ll(b, i);
}
static int ob_ps(OutBuf* b, int code, PyObject* ps_names, const char* none) {   /* sv._ps as text */
  if (code == -1) return ob_str(b, none);
  if (code == -2) return OB_LIT(b, "NULL");
  return ob_name(b, ps_names, code, "");
}

static int ob_obj(OutBuf* b, PyObject* o) {   /* str(o) */
  if (PyUnicode_Check(o)) return ob_py(b, o);
  if (PyLong_Check(o)) { int ovf = 0; const long long v = PyLong_AsLongLongAndOverflow(o, &ovf); if (!ovf && !(v == -1 && PyErr_Occurred())) return ob_ll(b, v); PyErr_Clear(); }
  PyObject* t = PyObject_Str(o);
  if (!t) return -1;
  const int rc = ob_py(b, t);
  Py_DECREF(t);
  return rc;
}
/* one sample column of a merged record (vcf.py:54-83 format_genotype on a 7-tuple): a, b, qual, dr, dv, phase, id */
static int ob_genotype(OutBuf* b, PyObject* a, PyObject* bb, PyObject* qual, PyObject* dr, PyObject* dv, PyObject* phase, PyObject* id, int phased) {
  PyObject *hp = Py_None, *ps = NULL;      /* unpack_phase (vcf.py:40-51) */
  if (phase != Py_None) {
    if (PyTuple_Check(phase) && PyTuple_GET_SIZE(phase) == 2) { hp = PyTuple_GET_ITEM(phase, 0); ps = PyTuple_GET_ITEM(phase, 1); }
    else hp = phase;
  }
  if (ps && (ps == Py_None || (PyUnicode_Check(ps) && PyUnicode_CompareWithASCIIString(ps, "NULL") == 0))) ps = NULL;
  int swap = 0; const char* sep = "/";
  if (phased && hp != Py_None && PyLong_Check(a) && PyLong_Check(bb)) {
    const long x = PyLong_AsLong(a), y = PyLong_AsLong(bb);
    if ((x == 0 && y == 1) || (x == 1 && y == 1)) { sep = "|"; swap = PyUnicode_Check(hp) && PyUnicode_CompareWithASCIIString(hp, "1") == 0; }
  }
  if (ob_obj(b, swap ? bb : a) || ob_str(b, sep) || ob_obj(b, swap ? a : bb) || OB_LIT(b, ":") || ob_obj(b, qual) || OB_LIT(b, ":") || ob_obj(b, dr) ||
      OB_LIT(b, ":") || ob_obj(b, dv)) return -1;
  if (phased) { if (OB_LIT(b, ":")) return -1; if (ps ? ob_obj(b, ps) : OB_LIT(b, ".")) return -1; }
  return OB_LIT(b, ":") || (id ? ob_obj(b, id) : 0);      /* (id NULL: the caller appends the chained ids itself) */
}

/* ---- merged records from arrays alone, on several threads ------------------------------------------------------------------------
 * group_calls' text mode when every column is there (candidate columns, head columns, coverage vectors) and RNAMES are not printed:
 * nothing of the record is a Python object, so the emitted groups are cut into consecutive ranges and formatted by one thread each,
 * outside the interpreter lock, into buffers of their own (the caller joins them).  fmt_range is the loop body of py_group_calls for
 * that case, statement for statement (vcf.py:216-300 over sv.py:386-481); tests/test_pipeline.py holds its text against the objects'. */
typedef struct {
  const snf_group_out_t* O; const int64_t* E; Py_ssize_t ne; const int64_t* GO; Py_ssize_t ng; const int32_t* M; Py_ssize_t nm; const uint8_t* CH;
  const int64_t *SV, *TK; Py_ssize_t ns; const int32_t* SPOS; Py_ssize_t nspos; const int32_t* CS; Py_ssize_t nobj;
  const int64_t* EVO; const int32_t *EVB, *EVN; Py_ssize_t nblk; long long null_min;
  const snf_group_cand_t* REC; const char* idpool; Py_ssize_t idpool_len; const int64_t* ID_ST; const int32_t* ID_LN; const int8_t* PH_HP;
  const int64_t* PS_ST; const int32_t* PS_LN;
  const int32_t *EM_TASK, *EM_TYP; const int64_t* AOFF; const char* apool; Py_ssize_t apool_len;
  const char** ctg; size_t* ctg_len; Py_ssize_t n_ctg; const char** typ; size_t* typ_len; Py_ssize_t n_typ;
  const char* prefix; size_t prefix_len; const char* fmt; size_t fmt_len;
  const int32_t** dn_ptr; const Py_ssize_t* dn_len; const int32_t* EBT; const int64_t* EBS; long long cvx_cb, cvx_bs;
  int t_phase, t_symbolic, t_mosaic, t_nm; long long t_minsvlen;
} FmtCtx;
typedef struct { const FmtCtx* c; Py_ssize_t e0, e1; OutBuf tb; int64_t* off; const char* err; } FmtJob;      /* off: e1 - e0 + 1 offsets inside tb */

static int ob_room_quiet(OutBuf* b, size_t extra) {      /* ob_room without the Python error (no interpreter lock here) */
  if (b->n + extra <= b->cap) return 0;
  size_t cap = b->cap ? b->cap : 1 << 16;
  while (cap < b->n + extra) cap *= 2;
  char* q = (char*)realloc(b->p, cap);
  if (!q) return -1;
  b->p = q; b->cap = cap;
  return 0;
}
static inline int qb_put(OutBuf* b, const char* s, size_t n) {
  if (__builtin_expect(b->n + n > b->cap, 0) && ob_room_quiet(b, n)) return -1;
  memcpy(b->p + b->n, s, n); b->n += n; return 0;
}
#define QB_LIT(b, lit) qb_put((b), "" lit, sizeof(lit) - 1)
static inline int qb_ll(OutBuf* b, long long v) { if (ob_room_quiet(b, 24)) return -1; b->n = (size_t)(raw_ll(b->p + b->n, v) - b->p); return 0; }
static inline int qb_hex(OutBuf* b, unsigned long long u) { if (ob_room_quiet(b, 16)) return -1; b->n = (size_t)(raw_hex(b->p + b->n, u) - b->p); return 0; }
static int qb_f3(OutBuf* b, double v) {      /* ob_f3 (the same conversion) into a quiet buffer */
  OutBuf t = {NULL, 0, 0}; char tmp[400]; t.p = tmp; t.cap = sizeof tmp;      /* (352 characters at most: never grows) */
  if (ob_f3(&t, v)) return -1;
  return qb_put(b, tmp, t.n);
}

static void fmt_range(FmtJob* J) {
  const FmtCtx* c = J->c; OutBuf* tb = &J->tb; const Py_ssize_t ns = c->ns;
  OutBuf *scol = (OutBuf*)calloc((size_t)ns + 1, sizeof(OutBuf)), *idc = (OutBuf*)calloc((size_t)ns + 1, sizeof(OutBuf));
  uint8_t* present = (uint8_t*)malloc((size_t)ns + 1);
  int* head = NULL; Py_ssize_t cap = 0; Py_ssize_t* evx_row = NULL; long long* evx_idx = NULL; int64_t evx_cap = 0;
#define FAIL(msg) do { J->err = (msg); goto out; } while (0)
  if (!scol || !idc || !present) FAIL("out of memory");
  J->off[0] = 0;
  for (Py_ssize_t e = J->e0; e < J->e1; e++) {
    const int64_t g = c->E[e];
    if (g < 0 || g >= c->ng) FAIL("group index out of range");
    const snf_group_out_t* o = &c->O[g];
    const int64_t lo = c->GO[g], hi = c->GO[g + 1], n = hi - lo;
    if (lo < 0 || hi > c->nm || n <= 0 || o->alt_member < lo || o->alt_member >= hi) FAIL("group range out of bounds");
    if (n > cap) { free(head); cap = n + 16; head = (int*)malloc((size_t)cap * sizeof(int)); if (!head) FAIL("out of memory"); }
    const int32_t* M = c->M;
    for (int64_t k = 0; k < n; k++) if (M[lo + k] < 0 || M[lo + k] >= c->nobj) FAIL("member out of range");
    long long t_ac = 0;
    for (Py_ssize_t si = 0; si < ns; si++) { scol[si].n = 0; idc[si].n = 0; }
    memset(present, 0, (size_t)ns);
    /* ---- the chained ids of every sample in the group, in add order (sv.py:386-404) */
    for (int64_t k = 0; k < n; k++) {
      const long sidv = c->CS[M[lo + k]];
      head[k] = (int)k;
      for (int64_t j = 0; j < k; j++) { if (head[j] != j) continue; if (c->CS[M[lo + j]] == sidv) { head[k] = (int)j; break; } }
      const Py_ssize_t sx = (sidv >= 0 && sidv < c->nspos) ? c->SPOS[sidv] : -1;
      const int64_t is_ = c->ID_ST[M[lo + k]]; const int32_t il = c->ID_LN[M[lo + k]];
      if (is_ < 0 || il < 0 || is_ + il > c->idpool_len) FAIL("id outside the pool");
      if (sx >= 0 && sx < ns) {
        if ((idc[sx].n && QB_LIT(&idc[sx], ",")) || qb_put(&idc[sx], c->prefix, c->prefix_len) || qb_put(&idc[sx], c->idpool + is_, (size_t)il)) FAIL("out of memory");
        present[sx] = 1;
      }
    }
    /* ---- the genotype column of every sample: from the record of the candidate that speaks for it */
    for (int64_t k = 0; k < n; k++) {
      if (head[k] != k) continue;
      int64_t pick = -1;
      for (int64_t j = k; j < n; j++) if (head[j] == k && c->CH[lo + j]) pick = j;
      if (pick < 0) FAIL("no chosen genotype for a sample");
      const snf_group_cand_t* r = &c->REC[M[lo + pick]];
      const long sidv = c->CS[M[lo + k]];
      const Py_ssize_t si = (sidv >= 0 && sidv < c->nspos) ? c->SPOS[sidv] : -1;
      if (si < 0 || si >= ns) continue;
      OutBuf* sc = &scol[si];
      int ga = r->gt_a, gb_ = r->gt_b; char sep = '/';
      const int8_t hp = c->PH_HP[M[lo + pick]];
      if (c->t_phase && hp >= 0 && ((ga == 0 && gb_ == 1) || (ga == 1 && gb_ == 1))) { sep = '|'; if (hp == 1) { const int x = ga; ga = gb_; gb_ = x; } }   /* vcf.py:66-72 */
      const int64_t ps0 = c->t_phase ? c->PS_ST[M[lo + pick]] : 0; const int32_t psl = c->t_phase ? c->PS_LN[M[lo + pick]] : -1;
      if (c->t_phase && psl >= 0 && (ps0 < 0 || ps0 + psl > c->idpool_len)) FAIL("phase set outside the pool");
      if (ob_room_quiet(sc, 160 + (size_t)(psl > 0 ? psl : 0) + idc[si].n)) FAIL("out of memory");
      char* w = sc->p + sc->n;
      if (ga < 0) *w++ = '.'; else w = raw_ll(w, ga);
      *w++ = sep;
      if (gb_ < 0) *w++ = '.'; else w = raw_ll(w, gb_);
      *w++ = ':'; w = raw_ll(w, r->gq); *w++ = ':'; w = raw_ll(w, r->dr); *w++ = ':'; w = raw_ll(w, r->dv); *w++ = ':';
      if (c->t_phase) { if (psl < 0) *w++ = '.'; else { memcpy(w, c->idpool + ps0, (size_t)psl); w += psl; } *w++ = ':'; }
      memcpy(w, idc[si].p, idc[si].n); w += idc[si].n;
      sc->n = (size_t)(w - sc->p);
      if (r->gt_a >= 0 && r->dv > 0) { t_ac += (long long)r->gt_a + r->gt_b; present[si] = 2; }
    }
    /* ---- samples without a candidate in the group (sv.py:405-414): the deepest coverage bin the group saw while it was active */
    const int64_t nev = c->EVO[e + 1] - c->EVO[e];
    if (nev > evx_cap) {
      free(evx_row); free(evx_idx);
      evx_cap = nev + 16; evx_row = (Py_ssize_t*)malloc((size_t)evx_cap * sizeof(Py_ssize_t)); evx_idx = (long long*)malloc((size_t)evx_cap * sizeof(long long));
      if (!evx_row || !evx_idx) FAIL("out of memory");
    }
    for (int64_t q = c->EVO[e]; q < c->EVO[e + 1]; q++) {
      if (c->EVB[q] < 0 || c->EVB[q] >= c->nblk) FAIL("block index out of range");
      const long long key = c->EVN[q], bstart = c->EBS[c->EVB[q]];
      evx_row[q - c->EVO[e]] = (Py_ssize_t)c->EBT[c->EVB[q]] * ns;
      evx_idx[q - c->EVO[e]] = (key >= bstart && key < bstart + c->cvx_bs && key % c->cvx_cb == 0) ? key / c->cvx_cb : -1;
    }
    for (Py_ssize_t si = 0; si < ns; si++) {
      if (present[si]) continue;
      long cov = 0; int firstev = 1;
      for (int64_t q = 0; q < nev; q++) {
        long cv = 0;
        const int32_t* dv_ = c->dn_ptr[evx_row[q] + si];
        if (dv_ && evx_idx[q] >= 0 && evx_idx[q] < c->dn_len[evx_row[q] + si]) { const int32_t x = dv_[evx_idx[q]]; if (x >= 0) cv = x; }
        cov = firstev ? cv : (cv > cov ? cv : cov);
        firstev = 0;
      }
      if (ob_room_quiet(&scol[si], 64)) FAIL("out of memory");
      char* w = scol[si].p + scol[si].n;
      memcpy(w, cov >= c->null_min ? "0/0:0:" : "./.:0:", 6); w = raw_ll(w + 6, cov);
      if (c->t_phase) { memcpy(w, ":0:.:NULL", 9); w += 9; } else { memcpy(w, ":0:NULL", 7); w += 7; }
      scol[si].n = (size_t)(w - scol[si].p);
    }
    /* ---- the record (vcf.py:216-300 write_call over the call of sv.py:440-481) */
    {
      const int32_t tk = c->EM_TASK[e], ty = c->EM_TYP[e];
      const int64_t a0 = c->AOFF[M[o->alt_member]], a1 = c->AOFF[M[o->alt_member] + 1];
      if (tk < 0 || tk >= c->n_ctg || ty < 0 || ty >= c->n_typ || a0 < 0 || a1 < a0 || a1 > c->apool_len) FAIL("head column out of range");
      const char* tname = c->typ[ty]; const size_t tlen = c->typ_len[ty];
      const char* altp = c->apool + a0; const Py_ssize_t alen = (Py_ssize_t)(a1 - a0);
      int skip = 0, any = 0;
      for (Py_ssize_t si = 0; si < ns; si++) any |= present[si] == 2;
      if (ns > 1 && !any) skip = 1;                        /* int(supp_vec) == 0 (vcf.py:246-247) */
      if (!skip) {
        const int bnd = strcmp(tname, "BND") == 0, ins = strcmp(tname, "INS") == 0, del = strcmp(tname, "DEL") == 0;
        long long svlen = o->svlen;
        if (ins && !c->t_symbolic && svlen != alen && !(alen == 5 && memcmp(altp, "<INS>", 5) == 0)) svlen = alen;   /* vcf.py:253-254 */
        if (ins && svlen < c->t_minsvlen) skip = 1;
        if (!skip) {
          const long long pos = o->pos > 0 ? o->pos : 1;
          const long long end = (o->precise && del) ? pos + (svlen < 0 ? -svlen : svlen) : o->end;
          int bad = qb_put(tb, c->ctg[tk], c->ctg_len[tk]) || QB_LIT(tb, "\t") || qb_ll(tb, pos) || QB_LIT(tb, "\t") || qb_put(tb, c->prefix, c->prefix_len) ||
                    qb_put(tb, tname, tlen < 40 ? tlen : 40) || QB_LIT(tb, ".") || qb_hex(tb, (unsigned long long)c->SV[e]) || QB_LIT(tb, "M") ||
                    qb_hex(tb, (unsigned long long)c->TK[e]) || QB_LIT(tb, "\tN\t");
          if (!bad) bad = (c->t_symbolic && !bnd) ? (QB_LIT(tb, "<") || qb_put(tb, tname, tlen) || QB_LIT(tb, ">")) : qb_put(tb, altp, (size_t)alen);
          if (!bad) {
            if (o->qual == SNF_NONE_I32) bad = QB_LIT(tb, "\t.");
            else { const long long q = o->qual < 0 ? 0 : o->qual > 60 ? 60 : o->qual; bad = QB_LIT(tb, "\t") || qb_ll(tb, q); }
          }
          if (!bad) bad = ((ns > 1 && t_ac == 0) ? QB_LIT(tb, "\tGT\t") : QB_LIT(tb, "\tPASS\t")) || (o->precise ? QB_LIT(tb, "PRECISE") : QB_LIT(tb, "IMPRECISE")) ||
                          (c->t_mosaic && QB_LIT(tb, ";MOSAIC")) || QB_LIT(tb, ";SVTYPE=") || qb_put(tb, tname, tlen);
          if (!bad && !bnd) bad = QB_LIT(tb, ";SVLEN=") || qb_ll(tb, svlen) || QB_LIT(tb, ";END=") || qb_ll(tb, end);
          if (!bad) bad = QB_LIT(tb, ";SUPPORT=") || qb_ll(tb, o->support) || QB_LIT(tb, ";COVERAGE=");
          for (int q = 0; !bad && q < 5; q++) {
            bad = q && QB_LIT(tb, ",");
            if (!bad) bad = o->cov[q] == SNF_NONE_I32 ? QB_LIT(tb, "None") : qb_ll(tb, o->cov[q]);
          }
          if (!bad) bad = QB_LIT(tb, ";STRAND=") || (o->fwd > 0 && QB_LIT(tb, "+")) || (o->rev > 0 && QB_LIT(tb, "-")) || (c->t_nm && QB_LIT(tb, ";NM=-1"));
          if (!bad && ns > 1) bad = QB_LIT(tb, ";AC=") || qb_ll(tb, t_ac);      /* call.info, sorted: AC, STDEV_LEN, STDEV_POS, SUPP_VEC */
          if (!bad) {
            if (o->n < 2) bad = QB_LIT(tb, ";STDEV_LEN=0;STDEV_POS=0");
            else bad = QB_LIT(tb, ";STDEV_LEN=") || qb_f3(tb, o->stdev_len) || QB_LIT(tb, ";STDEV_POS=") || qb_f3(tb, o->stdev_pos);
          }
          if (!bad && ns > 1) { bad = QB_LIT(tb, ";SUPP_VEC="); for (Py_ssize_t si = 0; !bad && si < ns; si++) bad = qb_put(tb, present[si] == 2 ? "1" : "0", 1); }
          if (!bad) bad = QB_LIT(tb, "\t") || qb_put(tb, c->fmt, c->fmt_len);
          for (Py_ssize_t si = 0; !bad && si < ns; si++) bad = QB_LIT(tb, "\t") || qb_put(tb, scol[si].p, scol[si].n);
          if (!bad) bad = QB_LIT(tb, "\n");
          if (bad) FAIL("out of memory");
        }
      }
    }
    J->off[e - J->e0 + 1] = (int64_t)tb->n;
  }
out:
#undef FAIL
  if (scol) { for (Py_ssize_t si = 0; si < ns; si++) free(scol[si].p); free(scol); }
  if (idc) { for (Py_ssize_t si = 0; si < ns; si++) free(idc[si].p); free(idc); }
  free(present); free(head); free(evx_row); free(evx_idx);
}
static void* fmt_thread(void* arg) { fmt_range((FmtJob*)arg); return NULL; }

static PyObject* py_group_calls(PyObject* self, PyObject* args) {
  PyObject *cls, *fds_cls, *objs, *block_cov, *prefix, *topt = Py_None, *covx = Py_None;
  Py_buffer ob, eb, gb, mb, cb, svb, tkb, sidb, sposb, evo, evb, evn, csb;
  long long null_min; int single;
  if (!PyArg_ParseTuple(args, "OOO!y*y*y*y*y*y*y*y*y*O!y*y*y*LUpy*|OO", &cls, &fds_cls, &PyList_Type, &objs, &ob, &eb, &gb, &mb, &cb, &svb, &tkb, &sidb,
                        &sposb, &PyList_Type, &block_cov, &evo, &evb, &evn, &null_min, &prefix, &single, &csb, &topt, &covx))
    return NULL;
  /* covx (dict, optional): the `_COVERAGE` dicts of the blocks as dense vectors - dense[task][sample] = int32 depth per bin of `cb`
   * bp (-1: no entry), eb_task / eb_start = task and start of every block of `block_cov`.  A lookup `block["_COVERAGE"][bin]`
   * (parallel.py:543-551) is then dense[task][sample][bin / cb] when the bin lies inside the block, else "no entry": the coverage
   * of the samples without a candidate costs an index instead of a dict probe with a freshly built key (most of this function's time). */
  const int32_t** dn_ptr = NULL; Py_ssize_t* dn_len = NULL; Py_buffer* dn_buf = NULL; Py_ssize_t dn_n = 0, dn_tasks = 0;
  const int32_t* EBT = NULL; const int64_t* EBS = NULL; long long cvx_cb = 0, cvx_bs = 0; Py_buffer ebtb, ebsb; int have_covx = 0;
  /* text mode (topt: dict): the merged calls as VCF lines - what VCF.write_call (vcf.py:216-350) prints for the objects the other mode
   * builds - without building them: returns (text: bytes, line_off: bytes int64 (ne + 1), pos: bytes int64 (ne)); a call that is not
   * written (no supporting sample, INS below minsvlen) has an empty line.  Plain multi-sample merges only (the caller checks). */
  const int text = topt != Py_None;
  int t_phase = 0, t_symbolic = 0, t_mosaic = 0, t_rnames = 0, t_nm = 0; long long t_minsvlen = 0; PyObject *t_fmt = NULL;
  OutBuf tb = {NULL, 0, 0}, *scol = NULL, *idc = NULL; int64_t *t_off = NULL, *t_pos = NULL;
  Py_buffer fr_rec, fr_pool, fr_st, fr_ln, fr_hp, fr_pss, fr_psl; int fast_cols = 0, have_ph = 0;
  /* head columns (optional, with the candidate columns): contig and SV type of every emitted group and the ALT pool of the candidate
   * table - the record's CHROM / ID / ALT then come from arrays too and no candidate object is read at all (they are ~2 KB each and
   * cold: three attribute reads per record were a fifth of this function) */
  Py_buffer hd_task, hd_typ, hd_aoff, hd_apool; int head_cols = 0; PyObject *hd_contigs = NULL, *hd_types = NULL;
  /* (an error while the options are read leaves through here: the thirteen buffers PyArg_ParseTuple acquired - and whatever column
   * buffers were taken so far - are released; a numpy array stays export-locked otherwise) */
  Py_buffer* fr_all[11] = {&fr_hp, &fr_pss, &fr_psl, &fr_rec, &fr_pool, &fr_st, &fr_ln, &hd_task, &hd_typ, &hd_aoff, &hd_apool}; int fr_got = 0;
#define EARLY_FAIL() do { for (int k_ = 0; k_ < fr_got; k_++) PyBuffer_Release(fr_all[k_]); \
    PyBuffer_Release(&ob); PyBuffer_Release(&eb); PyBuffer_Release(&gb); PyBuffer_Release(&mb); PyBuffer_Release(&cb); PyBuffer_Release(&svb); \
    PyBuffer_Release(&tkb); PyBuffer_Release(&sidb); PyBuffer_Release(&sposb); PyBuffer_Release(&evo); PyBuffer_Release(&evb); PyBuffer_Release(&evn); \
    PyBuffer_Release(&csb); return NULL; } while (0)
  if (text) {
    if (!PyDict_Check(topt)) { PyErr_SetString(PyExc_TypeError, "group_calls: text options must be a dict"); EARLY_FAIL(); }
#define TOPT(name) PyDict_GetItemString(topt, name)
    if (!TOPT("phase") || !TOPT("symbolic") || !TOPT("mosaic") || !TOPT("output_rnames") || !TOPT("nm") || !TOPT("minsvlen") || !TOPT("genotype_format")) {
      PyErr_SetString(PyExc_KeyError, "group_calls: incomplete text options"); EARLY_FAIL(); }
    t_phase = PyObject_IsTrue(TOPT("phase")); t_symbolic = PyObject_IsTrue(TOPT("symbolic")); t_mosaic = PyObject_IsTrue(TOPT("mosaic"));
    t_rnames = PyObject_IsTrue(TOPT("output_rnames")); t_nm = PyObject_IsTrue(TOPT("nm")); t_minsvlen = PyLong_AsLongLong(TOPT("minsvlen"));
    t_fmt = TOPT("genotype_format");
    if (PyErr_Occurred()) EARLY_FAIL();
    /* candidate columns (optional): with them a record is
     * formatted from arrays alone: the genotype of a sample from its chosen candidate's record, the chained ids and the phase set
     * from a string pool (ph_hp: haplotype of genotypes[0]'s phase, -1 None; ph_ps_len -1: no phase set) */
    if (TOPT("rec") && TOPT("id_pool") && TOPT("id_start") && TOPT("id_len") && TOPT("ph_hp") && TOPT("ph_ps_start") && TOPT("ph_ps_len")) {
      static const char* names[7] = {"ph_hp", "ph_ps_start", "ph_ps_len", "rec", "id_pool", "id_start", "id_len"};
      for (int k_ = 0; k_ < 7; k_++) {
        if (PyObject_GetBuffer(TOPT(names[k_]), fr_all[k_], PyBUF_SIMPLE) != 0) EARLY_FAIL();
        fr_got = k_ + 1;
      }
      have_ph = 1; fast_cols = 1;
      if (TOPT("contigs") && TOPT("types") && TOPT("em_task") && TOPT("em_typ") && TOPT("alt_off") && TOPT("alt_pool") &&
          PyList_Check(TOPT("contigs")) && PyTuple_Check(TOPT("types"))) {
        static const char* hnames[4] = {"em_task", "em_typ", "alt_off", "alt_pool"};
        for (int k_ = 0; k_ < 4; k_++) {
          if (PyObject_GetBuffer(TOPT(hnames[k_]), fr_all[7 + k_], PyBUF_SIMPLE) != 0) EARLY_FAIL();
          fr_got = 8 + k_;
        }
        hd_contigs = TOPT("contigs"); hd_types = TOPT("types"); head_cols = 1;
      }
    }
#undef TOPT
  }
#undef EARLY_FAIL
  const int32_t* CS = (const int32_t*)csb.buf;      /* sample_internal_id per candidate (table order) */
  PyObject* ret = NULL;
  const snf_group_out_t* O = (const snf_group_out_t*)ob.buf;
  const int64_t* E = (const int64_t*)eb.buf; const Py_ssize_t ne = eb.len / 8;
  const int64_t* GO = (const int64_t*)gb.buf; const Py_ssize_t ng = gb.len / 8 - 1;
  const int32_t* M = (const int32_t*)mb.buf; const Py_ssize_t nm = mb.len / 4;
  const uint8_t* CH = (const uint8_t*)cb.buf;
  const int64_t* SV = (const int64_t*)svb.buf; const int64_t* TK = (const int64_t*)tkb.buf;
  const int32_t* SIDS = (const int32_t*)sidb.buf; const Py_ssize_t ns = sidb.len / 4;
  const int32_t* SPOS = (const int32_t*)sposb.buf; const Py_ssize_t nspos = sposb.len / 4;
  const int64_t* EVO = (const int64_t*)evo.buf; const int32_t* EVB = (const int32_t*)evb.buf; const int32_t* EVN = (const int32_t*)evn.buf;
  const Py_ssize_t nobj = PyList_GET_SIZE(objs), nblk = PyList_GET_SIZE(block_cov);
  PyObject** sid_objs = NULL; uint8_t* present = NULL; int* head = NULL; PyObject** chain = NULL; Py_ssize_t cap = 0;
  Py_ssize_t* evx_row = NULL; long long* evx_idx = NULL; int64_t evx_cap = 0;
  PyObject* out = NULL;
  if ((Py_ssize_t)(ob.len / sizeof(snf_group_out_t)) < ng || cb.len < nm || svb.len / 8 < ne || tkb.len / 8 < ne || evo.len / 8 < ne + 1 ||
      evb.len != evn.len || csb.len / 4 < nobj) { PyErr_SetString(PyExc_ValueError, "group_calls: table sizes do not match"); goto done; }
  if (fast_cols && ((Py_ssize_t)(fr_rec.len / sizeof(snf_group_cand_t)) < nobj || fr_st.len / 8 < nobj || fr_ln.len / 4 < nobj || fr_hp.len < nobj ||
                    fr_pss.len / 8 < nobj || fr_psl.len / 4 < nobj)) {
    PyErr_SetString(PyExc_ValueError, "group_calls: candidate columns shorter than the candidate list"); goto done; }
  if (head_cols && (hd_task.len / 4 < ne || hd_typ.len / 4 < ne || hd_aoff.len / 8 < nobj + 1)) {
    PyErr_SetString(PyExc_ValueError, "group_calls: head columns shorter than the tables"); goto done; }
  sid_objs = (PyObject**)calloc((size_t)ns + 1, sizeof(PyObject*)); present = (uint8_t*)malloc((size_t)ns + 1);
  if (!sid_objs || !present) { PyErr_NoMemory(); goto done; }
  for (Py_ssize_t i = 0; i < ns; i++) { sid_objs[i] = PyLong_FromLong(SIDS[i]); if (!sid_objs[i]) goto done; }
  if (covx != Py_None) {
    PyObject *dense = PyDict_Check(covx) ? PyDict_GetItemString(covx, "dense") : NULL, *o_ebt = dense ? PyDict_GetItemString(covx, "eb_task") : NULL,
             *o_ebs = dense ? PyDict_GetItemString(covx, "eb_start") : NULL, *o_cb = dense ? PyDict_GetItemString(covx, "cb") : NULL,
             *o_bs = dense ? PyDict_GetItemString(covx, "block_size") : NULL;
    if (!dense || !PyList_Check(dense) || !o_ebt || !o_ebs || !o_cb || !o_bs) { PyErr_SetString(PyExc_TypeError, "group_calls: malformed coverage vectors"); goto done; }
    cvx_cb = PyLong_AsLongLong(o_cb); cvx_bs = PyLong_AsLongLong(o_bs);
    if (PyErr_Occurred() || cvx_cb <= 0 || cvx_bs <= 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "group_calls: bin / block size"); goto done; }
    if (PyObject_GetBuffer(o_ebt, &ebtb, PyBUF_SIMPLE) != 0) goto done;
    if (PyObject_GetBuffer(o_ebs, &ebsb, PyBUF_SIMPLE) != 0) { PyBuffer_Release(&ebtb); goto done; }
    have_covx = 1;
    EBT = (const int32_t*)ebtb.buf; EBS = (const int64_t*)ebsb.buf;
    if (ebtb.len / 4 < nblk || ebsb.len / 8 < nblk) { PyErr_SetString(PyExc_ValueError, "group_calls: block tables too short"); goto done; }
    dn_tasks = PyList_GET_SIZE(dense);
    dn_ptr = (const int32_t**)calloc((size_t)(dn_tasks * ns) + 1, sizeof(int32_t*)); dn_len = (Py_ssize_t*)calloc((size_t)(dn_tasks * ns) + 1, sizeof(Py_ssize_t));
    dn_buf = (Py_buffer*)calloc((size_t)(dn_tasks * ns) + 1, sizeof(Py_buffer));
    if (!dn_ptr || !dn_len || !dn_buf) { PyErr_NoMemory(); goto done; }
    for (Py_ssize_t t = 0; t < dn_tasks; t++) {
      PyObject* row = PyList_GET_ITEM(dense, t);
      if (!PyList_Check(row) || PyList_GET_SIZE(row) != ns) { PyErr_SetString(PyExc_TypeError, "group_calls: one coverage vector per sample expected"); goto done; }
      for (Py_ssize_t si = 0; si < ns; si++) {
        PyObject* a = PyList_GET_ITEM(row, si);
        if (a == Py_None) continue;
        if (PyObject_GetBuffer(a, &dn_buf[dn_n], PyBUF_SIMPLE) != 0) goto done;
        dn_ptr[t * ns + si] = (const int32_t*)dn_buf[dn_n].buf; dn_len[t * ns + si] = dn_buf[dn_n].len / 4;
        dn_n++;
      }
    }
    for (Py_ssize_t q = 0; q < nblk; q++) if (EBT[q] < 0 || EBT[q] >= dn_tasks) { PyErr_SetString(PyExc_ValueError, "group_calls: block task out of range"); goto done; }
  }
  if (text && fast_cols && head_cols && have_covx && !t_rnames) {
    /* every column is there and no RNAMES: the records from arrays alone, by `threads` threads outside the interpreter lock (fmt_range) */
    FmtCtx c; memset(&c, 0, sizeof c);
    FmtJob* jobs = NULL; pthread_t* tids = NULL; int64_t* all_off = NULL; int64_t* all_pos = NULL; int failed = 0;
    PyObject* thr_o = PyDict_GetItemString(topt, "threads");
    long nthr = thr_o ? PyLong_AsLong(thr_o) : 1;
    if (PyErr_Occurred()) goto done;
    if (nthr < 1) nthr = 1;
    if (nthr > 64) nthr = 64;
    if (nthr > ne / 64 + 1) nthr = (long)(ne / 64 + 1);
    c.n_ctg = PyList_GET_SIZE(hd_contigs); c.n_typ = PyTuple_GET_SIZE(hd_types);
    c.ctg = (const char**)calloc((size_t)c.n_ctg + 1, sizeof(char*)); c.ctg_len = (size_t*)calloc((size_t)c.n_ctg + 1, sizeof(size_t));
    c.typ = (const char**)calloc((size_t)c.n_typ + 1, sizeof(char*)); c.typ_len = (size_t*)calloc((size_t)c.n_typ + 1, sizeof(size_t));
    jobs = (FmtJob*)calloc((size_t)nthr, sizeof(FmtJob)); tids = (pthread_t*)calloc((size_t)nthr, sizeof(pthread_t));
    all_off = (int64_t*)calloc((size_t)ne + 1, 8); all_pos = (int64_t*)calloc((size_t)ne + 1, 8);
    if (!c.ctg || !c.ctg_len || !c.typ || !c.typ_len || !jobs || !tids || !all_off || !all_pos) { PyErr_NoMemory(); failed = 1; }
    for (Py_ssize_t k = 0; !failed && k < c.n_ctg + c.n_typ; k++) {      /* the strings of the head as UTF-8, taken while the lock is held */
      PyObject* u = k < c.n_ctg ? PyList_GET_ITEM(hd_contigs, k) : PyTuple_GET_ITEM(hd_types, k - c.n_ctg);
      Py_ssize_t ul = 0; const char* us = PyUnicode_Check(u) ? PyUnicode_AsUTF8AndSize(u, &ul) : NULL;
      if (!us) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "group_calls: contig / type names must be str"); failed = 1; break; }
      if (k < c.n_ctg) { c.ctg[k] = us; c.ctg_len[k] = (size_t)ul; } else { c.typ[k - c.n_ctg] = us; c.typ_len[k - c.n_ctg] = (size_t)ul; }
    }
    if (!failed) {
      Py_ssize_t pl = 0, fl = 0;
      c.prefix = PyUnicode_AsUTF8AndSize(prefix, &pl); c.fmt = PyUnicode_Check(t_fmt) ? PyUnicode_AsUTF8AndSize(t_fmt, &fl) : NULL;
      if (!c.prefix || !c.fmt) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "group_calls: genotype_format must be str"); failed = 1; }
      c.prefix_len = (size_t)pl; c.fmt_len = (size_t)fl;
    }
    if (!failed) {
      c.O = O; c.E = E; c.ne = ne; c.GO = GO; c.ng = ng; c.M = M; c.nm = nm; c.CH = CH; c.SV = SV; c.TK = TK; c.ns = ns; c.SPOS = SPOS; c.nspos = nspos;
      c.CS = CS; c.nobj = nobj; c.EVO = EVO; c.EVB = EVB; c.EVN = EVN; c.nblk = nblk; c.null_min = null_min;
      c.REC = (const snf_group_cand_t*)fr_rec.buf; c.idpool = (const char*)fr_pool.buf; c.idpool_len = fr_pool.len; c.ID_ST = (const int64_t*)fr_st.buf;
      c.ID_LN = (const int32_t*)fr_ln.buf; c.PH_HP = (const int8_t*)fr_hp.buf; c.PS_ST = (const int64_t*)fr_pss.buf; c.PS_LN = (const int32_t*)fr_psl.buf;
      c.EM_TASK = (const int32_t*)hd_task.buf; c.EM_TYP = (const int32_t*)hd_typ.buf; c.AOFF = (const int64_t*)hd_aoff.buf; c.apool = (const char*)hd_apool.buf;
      c.apool_len = hd_apool.len; c.dn_ptr = dn_ptr; c.dn_len = dn_len; c.EBT = EBT; c.EBS = EBS; c.cvx_cb = cvx_cb; c.cvx_bs = cvx_bs;
      c.t_phase = t_phase; c.t_symbolic = t_symbolic; c.t_mosaic = t_mosaic; c.t_nm = t_nm; c.t_minsvlen = t_minsvlen;
      for (long t = 0; t < nthr; t++) {
        jobs[t].c = &c; jobs[t].e0 = ne * t / nthr; jobs[t].e1 = ne * (t + 1) / nthr;
        jobs[t].off = (int64_t*)calloc((size_t)(jobs[t].e1 - jobs[t].e0) + 1, 8);
        if (!jobs[t].off) { PyErr_NoMemory(); failed = 1; }
      }
    }
    if (!failed) {
      int started[64]; memset(started, 0, sizeof started);
      Py_BEGIN_ALLOW_THREADS
      for (long t = 1; t < nthr; t++) started[t] = pthread_create(&tids[t], NULL, fmt_thread, &jobs[t]) == 0;
      fmt_range(&jobs[0]);
      for (long t = 1; t < nthr; t++) { if (started[t]) pthread_join(tids[t], NULL); else fmt_range(&jobs[t]); }      /* (no thread to be had: here) */
      Py_END_ALLOW_THREADS
      size_t total = 0;
      for (long t = 0; t < nthr && !failed; t++) {
        if (jobs[t].err) { PyErr_Format(strcmp(jobs[t].err, "out of memory") == 0 ? PyExc_MemoryError : PyExc_ValueError, "group_calls: %s", jobs[t].err); failed = 1; }
        total += jobs[t].tb.n;
      }
      if (!failed) {
        PyObject* a = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
        if (a) {
          char* w = PyBytes_AS_STRING(a); size_t base = 0;
          for (long t = 0; t < nthr; t++) {
            if (jobs[t].tb.n) memcpy(w + base, jobs[t].tb.p, jobs[t].tb.n);
            for (Py_ssize_t e = jobs[t].e0; e <= jobs[t].e1; e++) all_off[e] = (int64_t)base + jobs[t].off[e - jobs[t].e0];
            base += jobs[t].tb.n;
          }
          for (Py_ssize_t e = 0; e < ne; e++) all_pos[e] = O[E[e]].pos;      /* (E checked by the ranges) */
        }
        PyObject* b2 = a ? PyBytes_FromStringAndSize((const char*)all_off, ((Py_ssize_t)ne + 1) * 8) : NULL;
        PyObject* c2 = b2 ? PyBytes_FromStringAndSize((const char*)all_pos, (Py_ssize_t)ne * 8) : NULL;
        if (a && b2 && c2) ret = PyTuple_Pack(3, a, b2, c2);
        Py_XDECREF(a); Py_XDECREF(b2); Py_XDECREF(c2);
      }
    }
    if (jobs) for (long t = 0; t < nthr; t++) { free(jobs[t].tb.p); free(jobs[t].off); }
    free(jobs); free(tids); free(all_off); free(all_pos); free((void*)c.ctg); free(c.ctg_len); free((void*)c.typ); free(c.typ_len);
    goto done;
  }
  out = PyList_New(text ? 0 : ne);
  if (!out) goto done;
  if (text) {
    scol = (OutBuf*)calloc((size_t)ns + 1, sizeof(OutBuf)); idc = (OutBuf*)calloc((size_t)ns + 1, sizeof(OutBuf)); t_off = (int64_t*)calloc((size_t)ne + 1, 8); t_pos = (int64_t*)calloc((size_t)ne + 1, 8);
    if (!scol || !idc || !t_off || !t_pos) { PyErr_NoMemory(); goto done; }
  }
  for (Py_ssize_t e = 0; e < ne; e++) {
    const int64_t g = E[e];
    if (g < 0 || g >= ng) { PyErr_SetString(PyExc_ValueError, "group index out of range"); goto fail; }
    const snf_group_out_t* o = &O[g];
    const int64_t lo = GO[g], hi = GO[g + 1], n = hi - lo;
    if (lo < 0 || hi > nm || n <= 0 || o->alt_member < lo || o->alt_member >= hi) { PyErr_SetString(PyExc_ValueError, "group range out of bounds"); goto fail; }
    if (n > cap) {
      free(head); free(chain);
      cap = n + 16; head = (int*)malloc((size_t)cap * sizeof(int)); chain = (PyObject**)calloc((size_t)cap, sizeof(PyObject*));
      if (!head || !chain) { PyErr_NoMemory(); goto fail; }
    }
    for (int64_t k = 0; k < n; k++) if (M[lo + k] < 0 || M[lo + k] >= nobj) { PyErr_SetString(PyExc_ValueError, "member out of range"); goto fail; }
    PyObject* first = PyList_GET_ITEM(objs, M[lo]);
    PyObject *d = text ? NULL : _PyDict_NewPresized(34), *gts = text ? NULL : _PyDict_NewPresized(ns), *names = PyList_New(0), *info = NULL, *fds = NULL;
    int bad = (!text && (!d || !gts)) || !names;
    long long t_ac = 0;
    if (text) for (Py_ssize_t si = 0; si < ns; si++) { scol[si].n = 0; idc[si].n = 0; }
    memset(present, 0, (size_t)ns);
    /* ---- genotypes of the samples in the group (sv.py:386-404): first appearance order; ids chained in add order */
    for (int64_t k = 0; !bad && k < n; k++) {
      PyObject* c = PyList_GET_ITEM(objs, M[lo + k]);
      const long sidv = CS[M[lo + k]];
      head[k] = (int)k; chain[k] = NULL;
      for (int64_t j = 0; j < k; j++) {
        if (head[j] != j) continue;
        if (CS[M[lo + j]] == sidv) { head[k] = (int)j; break; }
      }
      if (fast_cols) {      /* arrays only: id from the pool, nothing of the candidate object is touched */
        const Py_ssize_t sx = (sidv >= 0 && sidv < nspos) ? SPOS[sidv] : -1;
        const int64_t is_ = ((const int64_t*)fr_st.buf)[M[lo + k]]; const int32_t il = ((const int32_t*)fr_ln.buf)[M[lo + k]];
        if (is_ < 0 || il < 0 || is_ + il > fr_pool.len) { PyErr_SetString(PyExc_ValueError, "group_calls: id outside the pool"); bad = 1; break; }
        if (sx >= 0 && sx < ns) bad = (idc[sx].n && OB_LIT(&idc[sx], ",")) || ob_py(&idc[sx], prefix) || ob_put(&idc[sx], (const char*)fr_pool.buf + is_, (size_t)il);
        if (bad) break;
        if (sx >= 0) present[sx] = 1;
        if (t_rnames) {
          PyObject* rn = aget(c, K_rnames);
          if (!rn) { bad = 1; break; }
          if (rn != Py_None) { PyObject* r2 = PySequence_InPlaceConcat(names, rn); if (!r2) bad = 1; else Py_DECREF(r2); }
          Py_DECREF(rn);
        }
        continue;
      }
      PyObject* cid = aget(c, K_id);
      if (text) {      /* the chained ids of a sample go straight into its id buffer (no string objects) */
        const Py_ssize_t sx = (sidv >= 0 && sidv < nspos) ? SPOS[sidv] : -1;
        if (!cid) { bad = 1; break; }
        if (sx >= 0 && sx < ns) bad = (idc[sx].n && OB_LIT(&idc[sx], ",")) || ob_py(&idc[sx], prefix) || ob_obj(&idc[sx], cid);
        Py_DECREF(cid);
        if (bad) break;
        goto chained;
      }
      PyObject* pid = cid ? PyUnicode_Concat(prefix, cid) : NULL;
      Py_XDECREF(cid);
      if (!pid) { bad = 1; break; }
      if (head[k] == k) chain[k] = pid;
      else {
        PyObject* t = PyUnicode_Concat(chain[head[k]], S_comma);
        PyObject* t2 = t ? PyUnicode_Concat(t, pid) : NULL;
        Py_XDECREF(t); Py_DECREF(pid);
        if (!t2) { bad = 1; break; }
        Py_DECREF(chain[head[k]]); chain[head[k]] = t2;
      }
    chained:
      if (sidv >= 0 && sidv < nspos && SPOS[sidv] >= 0) present[SPOS[sidv]] = 1;
      if (!text || t_rnames) {
        PyObject* rn = aget(c, K_rnames);                                    /* sv.py:389-390 */
        if (!rn) { bad = 1; break; }
        if (rn != Py_None) { PyObject* r2 = PySequence_InPlaceConcat(names, rn); if (!r2) bad = 1; else Py_DECREF(r2); }
        Py_DECREF(rn);
      }
      PyObject* cg = aget(c, K_genotypes);                                 /* sv.py:391-392: the default is stored on the candidate */
      if (!cg) { bad = 1; break; }
      int has = PyDict_Check(cg) ? PyDict_Contains(cg, O_zero) : -1;
      if (has < 0) bad = 1;
      else if (!has) {
        PyObject* sup = aget(c, K_support);
        PyObject* t = sup ? Py_BuildValue("(OOiiOO)", S_dot, S_dot, 0, 0, sup, T_none2) : NULL;
        Py_XDECREF(sup);
        bad = !t || PyDict_SetItem(cg, O_zero, t);
        Py_XDECREF(t);
      }
      Py_DECREF(cg);
    }
    for (int64_t k = 0; !bad && k < n; k++) {
      if (head[k] != k) continue;
      int64_t pick = -1;
      for (int64_t j = k; j < n; j++) if (head[j] == k && CH[lo + j]) pick = j;
      if (pick < 0) { PyErr_SetString(PyExc_ValueError, "no chosen genotype for a sample"); bad = 1; break; }
      if (fast_cols) {
        const snf_group_cand_t* r = &((const snf_group_cand_t*)fr_rec.buf)[M[lo + pick]];
        const long sidv = CS[M[lo + k]];
        const Py_ssize_t si = (sidv >= 0 && sidv < nspos) ? SPOS[sidv] : -1;
        if (si >= 0 && si < ns) {
          OutBuf* sc = &scol[si];
          int ga = r->gt_a, gb_ = r->gt_b; const char* sep = "/";
          const int8_t hp = ((const int8_t*)fr_hp.buf)[M[lo + pick]];
          if (t_phase && hp >= 0 && ((ga == 0 && gb_ == 1) || (ga == 1 && gb_ == 1))) { sep = "|"; if (hp == 1) { const int x = ga; ga = gb_; gb_ = x; } }   /* vcf.py:66-72 */
          const int64_t ps0 = t_phase ? ((const int64_t*)fr_pss.buf)[M[lo + pick]] : 0; const int32_t psl = t_phase ? ((const int32_t*)fr_psl.buf)[M[lo + pick]] : -1;
          if (t_phase && psl >= 0 && (ps0 < 0 || ps0 + psl > fr_pool.len)) { PyErr_SetString(PyExc_ValueError, "group_calls: phase set outside the pool"); bad = 1; }
          else if (!(bad = ob_room(sc, 160 + (size_t)(psl > 0 ? psl : 0) + idc[si].n))) {      /* a / b : GQ : DR : DV [: PS] : ids, one reservation */
            char* w = sc->p + sc->n;
            if (ga < 0) *w++ = '.'; else w = raw_ll(w, ga);
            *w++ = sep[0];
            if (gb_ < 0) *w++ = '.'; else w = raw_ll(w, gb_);
            *w++ = ':'; w = raw_ll(w, r->gq); *w++ = ':'; w = raw_ll(w, r->dr); *w++ = ':'; w = raw_ll(w, r->dv); *w++ = ':';
            if (t_phase) {
              if (psl < 0) *w++ = '.'; else { memcpy(w, (const char*)fr_pool.buf + ps0, (size_t)psl); w += psl; }
              *w++ = ':';
            }
            memcpy(w, idc[si].p, idc[si].n); w += idc[si].n;
            sc->n = (size_t)(w - sc->p);
          }
          if (!bad && r->gt_a >= 0 && r->dv > 0) { t_ac += (long long)r->gt_a + r->gt_b; present[si] = 2; }
        }
        continue;
      }
      PyObject* c = PyList_GET_ITEM(objs, M[lo + pick]);
      PyObject* cg = aget(c, K_genotypes);
      PyObject* t = cg ? PyDict_GetItemWithError(cg, O_zero) : NULL;
      PyObject* sv = PyLong_FromLong(CS[M[lo + k]]);
      if (!t || !sv || !PyTuple_Check(t) || PyTuple_GET_SIZE(t) < 6) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "genotypes[0] must be a 6-tuple"); bad = 1; }
      else if (text) {
        const long sidv = CS[M[lo + k]];
        const Py_ssize_t si = (sidv >= 0 && sidv < nspos) ? SPOS[sidv] : -1;
        if (si >= 0 && si < ns) {      /* (a sample that has no VCF column is not printed) */
          PyObject *ga = PyTuple_GET_ITEM(t, 0), *gb_ = PyTuple_GET_ITEM(t, 1), *gdv = PyTuple_GET_ITEM(t, 4);
          bad = ob_genotype(&scol[si], ga, gb_, PyTuple_GET_ITEM(t, 2), PyTuple_GET_ITEM(t, 3), gdv, PyTuple_GET_ITEM(t, 5), NULL, t_phase) != 0 ||
                ob_put(&scol[si], idc[si].p, idc[si].n);
          /* vcf.py:238-242: allele count and support vector */
          if (!bad && !(PyUnicode_Check(ga) && PyUnicode_CompareWithASCIIString(ga, ".") == 0)) {
            const int pos_dv = PyObject_RichCompareBool(gdv, O_zero, Py_GT);
            if (pos_dv < 0) bad = 1;
            else if (pos_dv) { t_ac += PyLong_AsLongLong(ga) + PyLong_AsLongLong(gb_); if (PyErr_Occurred()) bad = 1; present[si] = 2; }
          }
        }
      }
      else {
        PyObject* g7 = PyTuple_Pack(7, PyTuple_GET_ITEM(t, 0), PyTuple_GET_ITEM(t, 1), PyTuple_GET_ITEM(t, 2), PyTuple_GET_ITEM(t, 3),
                                    PyTuple_GET_ITEM(t, 4), PyTuple_GET_ITEM(t, 5), chain[k]);
        bad = !g7 || PyDict_SetItem(gts, sv, g7);
        Py_XDECREF(g7);
      }
      Py_XDECREF(cg); Py_XDECREF(sv);
    }
    for (int64_t k = 0; k < n; k++) if (head[k] == k) Py_CLEAR(chain[k]);
    /* ---- samples without a candidate in the group (sv.py:405-414): the deepest coverage bin the group saw while it was active */
    if (have_covx && !bad) {      /* the events of the group once: row of the block's task, index of the bin in a coverage vector (-1: no entry) */
      const int64_t nev = EVO[e + 1] - EVO[e];
      if (nev > evx_cap) {
        free(evx_row); free(evx_idx);
        evx_cap = nev + 16; evx_row = (Py_ssize_t*)malloc((size_t)evx_cap * sizeof(Py_ssize_t)); evx_idx = (long long*)malloc((size_t)evx_cap * sizeof(long long));
        if (!evx_row || !evx_idx) { PyErr_NoMemory(); Py_XDECREF(d); Py_XDECREF(gts); Py_XDECREF(names); goto fail; }
      }
      for (int64_t q = EVO[e]; q < EVO[e + 1]; q++) {
        if (EVB[q] < 0 || EVB[q] >= nblk) { PyErr_SetString(PyExc_ValueError, "block index out of range"); bad = 1; break; }
        const long long key = EVN[q], bstart = EBS[EVB[q]];
        evx_row[q - EVO[e]] = (Py_ssize_t)EBT[EVB[q]] * ns;
        evx_idx[q - EVO[e]] = (key >= bstart && key < bstart + cvx_bs && key % cvx_cb == 0) ? key / cvx_cb : -1;
      }
    }
    for (Py_ssize_t si = 0; !bad && si < ns; si++) {
      if (present[si]) continue;
      long cov = 0; int firstev = 1;
      if (have_covx) {
        for (int64_t q = 0; q < EVO[e + 1] - EVO[e]; q++) {
          long cv = 0;
          const int32_t* dv_ = dn_ptr[evx_row[q] + si];
          if (dv_ && evx_idx[q] >= 0 && evx_idx[q] < dn_len[evx_row[q] + si]) {
            const int32_t x = dv_[evx_idx[q]];
            if (x >= 0) cv = x;
          }
          cov = firstev ? cv : (cv > cov ? cv : cov);
          firstev = 0;
        }
      }
      else for (int64_t q = EVO[e]; q < EVO[e + 1]; q++) {
        long cv = 0;
        if (EVB[q] < 0 || EVB[q] >= nblk) { PyErr_SetString(PyExc_ValueError, "block index out of range"); bad = 1; break; }
        PyObject* per = PyList_GET_ITEM(block_cov, EVB[q]);
        PyObject* dct = PyList_Check(per) && si < PyList_GET_SIZE(per) ? PyList_GET_ITEM(per, si) : Py_None;
        if (dct != Py_None) {
          PyObject* key = PyLong_FromLong(EVN[q]);
          PyObject* val = key ? PyObject_GetItem(dct, key) : NULL;
          if (!val && PyErr_ExceptionMatches(PyExc_KeyError)) PyErr_Clear();
          else if (!val) { Py_XDECREF(key); bad = 1; break; }
          if (val) { cv = PyLong_AsLong(val); Py_DECREF(val); if (cv == -1 && PyErr_Occurred()) { Py_XDECREF(key); bad = 1; break; } }
          Py_XDECREF(key);
        }
        cov = firstev ? cv : (cv > cov ? cv : cov);
        firstev = 0;
      }
      if (bad) break;
      if (text) {                                               /* (0, 0, 0, cov, 0, (None, None), "NULL") or (".", ".", ...) as a column */
        if (!(bad = ob_room(&scol[si], 64))) {
          char* w = scol[si].p + scol[si].n;
          memcpy(w, cov >= null_min ? "0/0:0:" : "./.:0:", 6); w = raw_ll(w + 6, cov);
          if (t_phase) { memcpy(w, ":0:.:NULL", 9); w += 9; } else { memcpy(w, ":0:NULL", 7); w += 7; }
          scol[si].n = (size_t)(w - scol[si].p);
        }
        continue;
      }
      PyObject* covo = PyLong_FromLong(cov);
      PyObject* ab = cov >= null_min ? O_zero : S_dot;          /* (0, 0, 0, cov, 0, (None, None), "NULL") or (".", ".", ...) */
      PyObject* g7 = covo ? PyTuple_Pack(7, ab, ab, O_zero, covo, O_zero, T_none2, S_NULL) : NULL;
      Py_XDECREF(covo);
      bad = !g7 || PyDict_SetItem(gts, sid_objs[si], g7);
      Py_XDECREF(g7);
    }
    if (text && !bad) {
      /* ---- the record (vcf.py:216-300 write_call over the call of sv.py:440-481) */
      PyObject *contig = NULL, *svtype = NULL, *alt = NULL;
      const char* altp = NULL; Py_ssize_t alen = 0;             /* head columns: the ALT as bytes of the pool (ASCII: the caller checks) */
      if (head_cols) {
        const int32_t tk = ((const int32_t*)hd_task.buf)[e], ty = ((const int32_t*)hd_typ.buf)[e];
        const int64_t a0 = ((const int64_t*)hd_aoff.buf)[M[o->alt_member]], a1 = ((const int64_t*)hd_aoff.buf)[M[o->alt_member] + 1];
        if (tk < 0 || tk >= PyList_GET_SIZE(hd_contigs) || ty < 0 || ty >= PyTuple_GET_SIZE(hd_types) || a0 < 0 || a1 < a0 || a1 > hd_apool.len) {
          PyErr_SetString(PyExc_ValueError, "group_calls: head column out of range"); bad = 1; }
        else {
          contig = PyList_GET_ITEM(hd_contigs, tk); svtype = PyTuple_GET_ITEM(hd_types, ty); Py_INCREF(contig); Py_INCREF(svtype);
          altp = (const char*)hd_apool.buf + a0; alen = (Py_ssize_t)(a1 - a0);
        }
      } else {
        contig = aget(first, K_contig); svtype = aget(first, K_svtype);
        alt = aget(PyList_GET_ITEM(objs, M[o->alt_member]), K_alt);
      }
      const char* tname = svtype && PyUnicode_Check(svtype) ? PyUnicode_AsUTF8(svtype) : NULL;
      t_off[e] = (int64_t)tb.n; t_pos[e] = o->pos;
      int skip = 0, any = 0;
      for (Py_ssize_t si = 0; si < ns; si++) any |= present[si] == 2;
      if (bad || !contig || !PyUnicode_Check(contig) || !tname || (!altp && (!alt || !PyUnicode_Check(alt)))) bad = 1;
      else if (ns > 1 && !any) skip = 1;                        /* int(supp_vec) == 0 (vcf.py:246-247) */
      if (!bad && !skip) {
        const int bnd = strcmp(tname, "BND") == 0, ins = strcmp(tname, "INS") == 0, del = strcmp(tname, "DEL") == 0;
        long long svlen = o->svlen;
        if (!altp) alen = PyUnicode_GET_LENGTH(alt);
        if (ins && !t_symbolic && svlen != alen &&
            (altp ? !(alen == 5 && memcmp(altp, "<INS>", 5) == 0) : PyUnicode_CompareWithASCIIString(alt, "<INS>") != 0)) svlen = alen;   /* vcf.py:253-254 */
        if (ins && svlen < t_minsvlen) skip = 1;
        if (!skip) {
          const long long pos = o->pos > 0 ? o->pos : 1;
          const long long end = (o->precise && del) ? pos + (svlen < 0 ? -svlen : svlen) : o->end;
          const size_t tlen = strlen(tname);      /* the id: f"{svtype}.{sv_id:X}M{task_id:X}" behind the prefix */
          bad = ob_py(&tb, contig) || OB_LIT(&tb, "\t") || ob_ll(&tb, pos) || OB_LIT(&tb, "\t") || ob_py(&tb, prefix) || ob_put(&tb, tname, tlen < 40 ? tlen : 40) ||
                OB_LIT(&tb, ".") || ob_hex(&tb, (unsigned long long)SV[e]) || OB_LIT(&tb, "M") || ob_hex(&tb, (unsigned long long)TK[e]) || OB_LIT(&tb, "\tN\t");
          if (!bad) {
            if (t_symbolic && !bnd) bad = OB_LIT(&tb, "<") || ob_put(&tb, tname, tlen) || OB_LIT(&tb, ">");
            else bad = altp ? ob_put(&tb, altp, (size_t)alen) : ob_py(&tb, alt);
          }
          if (!bad) {
            if (o->qual == SNF_NONE_I32) bad = OB_LIT(&tb, "\t.");
            else { const long long q = o->qual < 0 ? 0 : o->qual > 60 ? 60 : o->qual; bad = OB_LIT(&tb, "\t") || ob_ll(&tb, q); }
          }
          if (!bad) bad = ((ns > 1 && t_ac == 0) ? OB_LIT(&tb, "\tGT\t") : OB_LIT(&tb, "\tPASS\t")) || (o->precise ? OB_LIT(&tb, "PRECISE") : OB_LIT(&tb, "IMPRECISE")) ||
                          (t_mosaic && OB_LIT(&tb, ";MOSAIC")) || OB_LIT(&tb, ";SVTYPE=") || ob_put(&tb, tname, tlen);
          if (!bad && !bnd) bad = OB_LIT(&tb, ";SVLEN=") || ob_ll(&tb, svlen) || OB_LIT(&tb, ";END=") || ob_ll(&tb, end);
          if (!bad) bad = OB_LIT(&tb, ";SUPPORT=") || ob_ll(&tb, o->support);
          if (!bad && t_rnames) {
            bad = OB_LIT(&tb, ";RNAMES=");
            for (Py_ssize_t q = 0; !bad && q < PyList_GET_SIZE(names); q++) bad = (q && OB_LIT(&tb, ",")) || ob_obj(&tb, PyList_GET_ITEM(names, q));
          }
          if (!bad) {
            bad = OB_LIT(&tb, ";COVERAGE=");
            for (int q = 0; !bad && q < 5; q++) {
              bad = q && OB_LIT(&tb, ",");
              if (!bad) bad = o->cov[q] == SNF_NONE_I32 ? OB_LIT(&tb, "None") : ob_ll(&tb, o->cov[q]);
            }
          }
          if (!bad) bad = OB_LIT(&tb, ";STRAND=") || (o->fwd > 0 && OB_LIT(&tb, "+")) || (o->rev > 0 && OB_LIT(&tb, "-")) || (t_nm && OB_LIT(&tb, ";NM=-1"));
          if (!bad && ns > 1) bad = OB_LIT(&tb, ";AC=") || ob_ll(&tb, t_ac);      /* call.info, sorted: AC, STDEV_LEN, STDEV_POS, SUPP_VEC */
          if (!bad) {
            if (o->n < 2) bad = OB_LIT(&tb, ";STDEV_LEN=0;STDEV_POS=0");
            else bad = OB_LIT(&tb, ";STDEV_LEN=") || ob_f3(&tb, o->stdev_len) || OB_LIT(&tb, ";STDEV_POS=") || ob_f3(&tb, o->stdev_pos);
          }
          if (!bad && ns > 1) { bad = OB_LIT(&tb, ";SUPP_VEC="); for (Py_ssize_t si = 0; !bad && si < ns; si++) bad = ob_put(&tb, present[si] == 2 ? "1" : "0", 1); }
          if (!bad) bad = OB_LIT(&tb, "\t") || ob_py(&tb, t_fmt);
          for (Py_ssize_t si = 0; !bad && si < ns; si++) bad = OB_LIT(&tb, "\t") || ob_put(&tb, scol[si].p, scol[si].n);
          if (!bad) bad = OB_LIT(&tb, "\n");
        }
      }
      Py_XDECREF(contig); Py_XDECREF(svtype); Py_XDECREF(alt); Py_XDECREF(names);
      if (bad) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "group_calls: cannot format a merged record"); goto fail; }
      t_off[e + 1] = (int64_t)tb.n;
      continue;
    }
    if (text) { Py_XDECREF(names); if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "group_calls: cannot format a merged record"); goto fail; }
    /* ---- the call (sv.py:440-481) */
    PyObject *contig = NULL, *svtype = NULL, *alt = NULL, *flt = NULL;
    if (!bad) {
      contig = aget(first, K_contig); svtype = aget(first, K_svtype);
      alt = aget(PyList_GET_ITEM(objs, M[o->alt_member]), K_alt);
      if (single) { flt = aget(first, K_filter); info = aget(first, K_info); } else { flt = S_PASS; Py_INCREF(flt); info = PyDict_New(); }
      bad = !contig || !svtype || !alt || !flt || !info;
    }
    if (!bad && !single) {
      bad = set_steal(info, I_STDEV_POS, o->n < 2 ? PyLong_FromLong(0) : PyFloat_FromDouble(o->stdev_pos)) ||
            set_steal(info, I_STDEV_LEN, o->n < 2 ? PyLong_FromLong(0) : PyFloat_FromDouble(o->stdev_len));
    }
    if (!bad) {
      PyObject* fd = PyDict_New();
      PyObject* zero = PyLong_FromLong(0);
      bad = !fd || !zero || PyDict_SetItem(fd, F_n, zero) || PyDict_SetItem(fd, F_m1, zero) || PyDict_SetItem(fd, F_m2, zero) || PyDict_SetItem(fd, F_last, Py_None);
      Py_XDECREF(zero);
      if (!bad) { fds = new_instance(fds_cls, fd); bad = !fds; } else Py_XDECREF(fd);
    }
    if (!bad) {
      char idbuf[96];
      const char* tname = PyUnicode_AsUTF8(svtype);
      if (!tname) bad = 1;
      else {
        snprintf(idbuf, sizeof idbuf, "%.40s.%llXM%llX", tname, (unsigned long long)SV[e], (unsigned long long)TK[e]);
        bad = PyDict_SetItem(d, K_contig, contig) || set_steal(d, K_pos, PyLong_FromLong(o->pos)) || set_steal(d, K_id, PyUnicode_FromString(idbuf)) ||
              PyDict_SetItem(d, K_ref, S_N) || PyDict_SetItem(d, K_alt, alt) || set_steal(d, K_qual, int_or_none(o->qual)) ||
              PyDict_SetItem(d, K_filter, flt) || PyDict_SetItem(d, K_info, info) || PyDict_SetItem(d, K_svtype, svtype) ||
              set_steal(d, K_svlen, PyLong_FromLong(o->svlen)) || set_steal(d, K_end, PyLong_FromLong(o->end)) || PyDict_SetItem(d, K_genotypes, gts) ||
              PyDict_SetItem(d, K_precise, o->precise ? Py_True : Py_False) || set_steal(d, K_support, PyLong_FromLong(o->support)) ||
              PyDict_SetItem(d, K_rnames, names) || PyDict_SetItem(d, K_qc, Py_True) || set_steal(d, K_nm, PyLong_FromLong(-1)) ||
              PyDict_SetItem(d, K_postprocess, Py_None) || PyDict_SetItem(d, K_svlens, Py_None) || set_steal(d, K_fwd, PyLong_FromLong(o->fwd)) ||
              set_steal(d, K_rev, PyLong_FromLong(o->rev)) || PyDict_SetItem(d, K_fds, fds) || set_steal(d, K_cov_up, int_or_none(o->cov[0])) ||
              set_steal(d, K_cov_dn, int_or_none(o->cov[4])) || set_steal(d, K_cov_st, int_or_none(o->cov[1])) ||
              set_steal(d, K_cov_ce, int_or_none(o->cov[2])) || set_steal(d, K_cov_en, int_or_none(o->cov[3])) || PyDict_SetItem(d, K_sample, Py_None) ||
              PyDict_SetItem(d, K_bnd_info, Py_None) || PyDict_SetItem(d, K_sup_inline, Py_None) || PyDict_SetItem(d, K_sup_splits, Py_None) ||
              PyDict_SetItem(d, K_raw, Py_None) || PyDict_SetItem(d, K_raw_idx, Py_None);
      }
    }
    Py_XDECREF(contig); Py_XDECREF(svtype); Py_XDECREF(alt); Py_XDECREF(flt); Py_XDECREF(info); Py_XDECREF(fds); Py_XDECREF(gts); Py_XDECREF(names);
    if (bad) { Py_XDECREF(d); goto fail; }
    PyObject* obj = new_instance(cls, d);
    if (!obj) goto fail;
    PyList_SET_ITEM(out, e, obj);
  }
  if (text) {
    PyObject* a = PyBytes_FromStringAndSize(tb.p ? tb.p : "", (Py_ssize_t)tb.n);
    PyObject* b2 = PyBytes_FromStringAndSize((const char*)t_off, ((Py_ssize_t)ne + 1) * 8);
    PyObject* c2 = PyBytes_FromStringAndSize((const char*)t_pos, (Py_ssize_t)ne * 8);
    if (a && b2 && c2) ret = PyTuple_Pack(3, a, b2, c2);
    Py_XDECREF(a); Py_XDECREF(b2); Py_XDECREF(c2);
    goto done;
  }
  ret = out; out = NULL;
  goto done;
fail:
  if (chain) for (Py_ssize_t k = 0; k < cap; k++) Py_CLEAR(chain[k]);
done:
  Py_XDECREF(out);
  if (sid_objs) { for (Py_ssize_t i = 0; i < ns; i++) Py_XDECREF(sid_objs[i]); free(sid_objs); }
  free(present); free(head); free(chain); free(evx_row); free(evx_idx);
  if (scol) { for (Py_ssize_t si = 0; si < ns; si++) free(scol[si].p); free(scol); }
  if (idc) { for (Py_ssize_t si = 0; si < ns; si++) free(idc[si].p); free(idc); }
  if (fast_cols) { PyBuffer_Release(&fr_rec); PyBuffer_Release(&fr_pool); PyBuffer_Release(&fr_st); PyBuffer_Release(&fr_ln); }
  if (have_ph) { PyBuffer_Release(&fr_hp); PyBuffer_Release(&fr_pss); PyBuffer_Release(&fr_psl); }
  if (head_cols) { PyBuffer_Release(&hd_task); PyBuffer_Release(&hd_typ); PyBuffer_Release(&hd_aoff); PyBuffer_Release(&hd_apool); }
  if (dn_buf) { for (Py_ssize_t k = 0; k < dn_n; k++) PyBuffer_Release(&dn_buf[k]); free(dn_buf); }
  free(dn_ptr); free(dn_len);
  if (have_covx) { PyBuffer_Release(&ebtb); PyBuffer_Release(&ebsb); }
  free(tb.p); free(t_off); free(t_pos);
  PyBuffer_Release(&ob); PyBuffer_Release(&eb); PyBuffer_Release(&gb); PyBuffer_Release(&mb); PyBuffer_Release(&cb); PyBuffer_Release(&svb);
  PyBuffer_Release(&tkb); PyBuffer_Release(&sidb); PyBuffer_Release(&sposb); PyBuffer_Release(&evo); PyBuffer_Release(&evb); PyBuffer_Release(&evn);
  PyBuffer_Release(&csb);
  return ret;
}


/* ======================================================================================================================
 * vcf_records(): single-sample VCF lines straight from the record table (sniffles_amd/vcf.py::VCF.write_records): what
 * materialize + apply_final + VCF.write_call (vcf.py:216-350 of the reference) produce for the same records, without the
 * SVCall objects in between.  Serves the BAM -> VCF flow when no reference FASTA is attached and no SNF is written.
 * ====================================================================================================================== */
/* vcf_records(records: buffer, order: buffer int64 (record indices in output order), rnames: buffer uint32, alt_pool: buffer,
 *             qnames: list | None, ps_names: list | None, contig: str, task_id: int, contig_names: list | None,
 *             filters: list[str], opts: dict(id_prefix str, mosaic, mosaic_af_max, output_rnames, nm, phase, symbolic, minsvlen,
 *             genotype_format str, genotype_none str)) -> (bytes, records written) */
static PyObject* py_vcf_records(PyObject* self, PyObject* args) {
  PyObject *qnames, *ps_names, *contig, *contig_names, *filters, *opts;
  Py_buffer rec, ord, rn, pool;
  long long task_id;
  if (!PyArg_ParseTuple(args, "y*y*y*y*OOULOO!O!", &rec, &ord, &rn, &pool, &qnames, &ps_names, &contig, &task_id, &contig_names,
                        &PyList_Type, &filters, &PyDict_Type, &opts))
    return NULL;
  PyObject* ret = NULL;
  OutBuf b = {NULL, 0, 0};
  long long written = 0;
#define OPT(name) PyDict_GetItemString(opts, name)
  PyObject *o_prefix = OPT("id_prefix"), *o_fmt = OPT("genotype_format"), *o_none = OPT("genotype_none");
  if (!o_prefix || !o_fmt || !o_none || !OPT("mosaic") || !OPT("mosaic_af_max") || !OPT("output_rnames") || !OPT("nm") || !OPT("phase") ||
      !OPT("symbolic") || !OPT("minsvlen")) { PyErr_SetString(PyExc_KeyError, "vcf_records: incomplete options"); goto done; }
  const int mosaic = PyObject_IsTrue(OPT("mosaic")), out_rn = PyObject_IsTrue(OPT("output_rnames")), with_nm = PyObject_IsTrue(OPT("nm"));
  const int phase = PyObject_IsTrue(OPT("phase")), symbolic = PyObject_IsTrue(OPT("symbolic"));
  const double af_max = PyFloat_AsDouble(OPT("mosaic_af_max"));
  const long long minsvlen = PyLong_AsLongLong(OPT("minsvlen"));
  if (PyErr_Occurred()) goto done;
#undef OPT
  const snf_call_t* C = (const snf_call_t*)rec.buf;
  const long long n_rec = rec.len / (long long)sizeof(snf_call_t), n_ord = ord.len / 8, rn_n = rn.len / 4;
  const int64_t* O = (const int64_t*)ord.buf;
  const uint32_t* RN = (const uint32_t*)rn.buf;
  for (long long k = 0; k < n_ord; k++) {
    if (O[k] < 0 || O[k] >= n_rec) { PyErr_SetString(PyExc_ValueError, "record index outside the table"); goto done; }
    const snf_call_t* c = &C[O[k]];
    if (c->svtype < 0 || c->svtype > 6 || c->filter < 0 || c->filter >= PyList_GET_SIZE(filters)) { PyErr_SetString(PyExc_ValueError, "record field out of range"); goto done; }
    if (c->svtype >= 5) continue;                                            /* single breaks are not written (vcf.py:218) */
    const int bnd = c->svtype == SNF_BND, ins = c->svtype == SNF_INS;
    const long long pos = c->pos > 0 ? c->pos : 1;
    long long svlen = c->svlen;
    const int resolved = ins && c->alt_len >= 0 && !symbolic;                /* the consensus / best read, not "<INS>" */
    if (resolved) {
      if (c->alt_off < 0 || c->alt_off + c->alt_len > pool.len) { PyErr_SetString(PyExc_ValueError, "ALT range outside the pool"); goto done; }
      if (svlen != c->alt_len) svlen = c->alt_len;                          /* SVLEN follows the sequence (vcf.py:253-254) */
    }
    if (ins && svlen < minsvlen) continue;
    const long long end = (c->precise && c->svtype == SNF_DEL) ? pos + (svlen < 0 ? -svlen : svlen) : c->end;
    const size_t line_start = b.n;
    /* CHROM POS ID REF ALT */
    if (ob_py(&b, contig) || OB_LIT(&b, "\t") || ob_ll(&b, pos) || OB_LIT(&b, "\t") || ob_py(&b, o_prefix)) goto done;
    { char idbuf[64]; snprintf(idbuf, sizeof idbuf, "%s.%XS%llX", SVTYPES[c->svtype], (unsigned)c->sv_id, (unsigned long long)task_id);
      if (ob_str(&b, idbuf) || OB_LIT(&b, "\tN\t")) goto done; }
    if (bnd) {     /* sv.py:630-634; also with --symbolic (vcf.py:322-324 leaves BND ALTs alone) */
      const char* br = c->bnd_is_reverse ? "]" : "[";
      if (ob_str(&b, c->bnd_is_first ? "N" : "") || ob_str(&b, br) || ob_name(&b, contig_names, c->mate_contig, "ctg") || OB_LIT(&b, ":") ||
          ob_ll(&b, c->mate_ref_start) || ob_str(&b, br) || ob_str(&b, c->bnd_is_first ? "" : "N")) goto done;
    } else if (resolved) {
      if (ob_put(&b, (const char*)pool.buf + c->alt_off, (size_t)c->alt_len)) goto done;
    } else {
      if (OB_LIT(&b, "<") || ob_str(&b, SVTYPES[c->svtype]) || OB_LIT(&b, ">")) goto done;
    }
    /* QUAL FILTER */
    { const long long q = c->qual < 0 ? 0 : c->qual > 60 ? 60 : c->qual;
      if (OB_LIT(&b, "\t") || ob_ll(&b, q) || OB_LIT(&b, "\t") || ob_py(&b, PyList_GET_ITEM(filters, c->filter)) || OB_LIT(&b, "\t")) goto done; }
    /* INFO */
    if (ob_str(&b, c->precise ? "PRECISE" : "IMPRECISE")) goto done;
    if (mosaic && (c->gt_set ? c->vaf : 0.0) <= af_max) { if (OB_LIT(&b, ";MOSAIC")) goto done; }
    if (OB_LIT(&b, ";SVTYPE=") || ob_str(&b, SVTYPES[c->svtype])) goto done;
    if (!bnd) { if (OB_LIT(&b, ";SVLEN=") || ob_ll(&b, svlen) || OB_LIT(&b, ";END=") || ob_ll(&b, end)) goto done; }
    if (OB_LIT(&b, ";SUPPORT=") || ob_ll(&b, c->support)) goto done;
    if (out_rn) {
      if (c->rn_off < 0 || c->rn_len < 0 || c->rn_off + c->rn_len > rn_n) { PyErr_SetString(PyExc_ValueError, "read-name range outside the table"); goto done; }
      if (OB_LIT(&b, ";RNAMES=")) goto done;
      for (int q = 0; q < c->rn_len; q++) { if ((q && OB_LIT(&b, ",")) || ob_name(&b, qnames, (long)RN[c->rn_off + q], "q")) goto done; }
    }
    if (OB_LIT(&b, ";COVERAGE=")) goto done;
    { const int idx[5] = {0, 1, 2, 3, 4};
      for (int q = 0; q < 5; q++) { if ((q && OB_LIT(&b, ",")) || ob_ll(&b, c->cov[idx[q]])) goto done; } }
    if (OB_LIT(&b, ";STRAND=") || ob_str(&b, c->fwd > 0 ? "+" : "") || ob_str(&b, c->rev > 0 ? "-" : "")) goto done;
    if (with_nm) { if (OB_LIT(&b, ";NM=") || ob_f3(&b, c->nm)) goto done; }
    /* call.info, sorted by key: CHR2 < COVERAGE_VAR (None: not written) < PHASE < STDEV_LEN < STDEV_POS < SUPPORT_LONG < SUPPORT_SA < VAF */
    if (bnd) { if (OB_LIT(&b, ";CHR2=") || ob_name(&b, contig_names, c->mate_contig, "ctg")) goto done; }
    if (c->ph_set) {
      if (OB_LIT(&b, ";PHASE=") || ob_ll(&b, c->ph_hp) || OB_LIT(&b, ",") || ob_ps(&b, c->ph_ps, ps_names, "None") || OB_LIT(&b, ",") ||
          ob_ll(&b, c->ph_hp_support) || OB_LIT(&b, ",") || ob_ll(&b, c->ph_ps_support) || ob_str(&b, c->ph_hp_pass ? ",PASS" : ",FAIL") ||
          ob_str(&b, c->ph_ps_pass ? ",PASS" : ",FAIL")) goto done;
    }
    { const int single = c->fwd + c->rev < 2;      /* util.stdev returns the int 0 for fewer than two values */
      if (!isnan(c->stdev_len)) { if (OB_LIT(&b, ";STDEV_LEN=") || (single ? OB_LIT(&b, "0") : ob_f3(&b, c->stdev_len))) goto done; }
      if (OB_LIT(&b, ";STDEV_POS=") || (single ? OB_LIT(&b, "0") : ob_f3(&b, c->stdev_pos))) goto done; }
    if (ins) { if (OB_LIT(&b, ";SUPPORT_LONG=") || ob_ll(&b, c->support_long)) goto done; }
    if (c->svtype == SNF_DEL) { if (OB_LIT(&b, ";SUPPORT_SA=") || ob_ll(&b, c->support_sa)) goto done; }
    if (c->gt_set) { if (OB_LIT(&b, ";VAF=") || ob_f3(&b, c->vaf)) goto done; }
    /* FORMAT + the sample column (vcf.py:54-83) */
    if (OB_LIT(&b, "\t") || ob_py(&b, o_fmt) || OB_LIT(&b, "\t")) goto done;
    if (!c->gt_set) { if (ob_py(&b, o_none)) goto done; }
    else {
      int a = c->gt_a, bb = c->gt_b;
      const int hp_set = c->gt_hp >= 0;
      const char* sep = "/";
      if (phase && hp_set && ((a == 0 && bb == 1) || (a == 1 && bb == 1))) { sep = "|"; if (c->gt_hp == 1) { const int t = a; a = bb; bb = t; } }
      if (ob_ll(&b, a) || ob_str(&b, sep) || ob_ll(&b, bb) || OB_LIT(&b, ":") || ob_ll(&b, c->gt_gq) || OB_LIT(&b, ":") || ob_ll(&b, c->gt_dr) ||
          OB_LIT(&b, ":") || ob_ll(&b, c->gt_dv)) goto done;
      if (phase) { if (OB_LIT(&b, ":") || (c->gt_ps == -1 || c->gt_ps == -2 ? OB_LIT(&b, ".") : ob_ps(&b, c->gt_ps, ps_names, "."))) goto done; }
    }
    if (OB_LIT(&b, "\n")) goto done;
    (void)line_start;
    written++;
  }
  { PyObject* bytes = PyBytes_FromStringAndSize(b.p ? b.p : "", (Py_ssize_t)b.n);
    if (bytes) { ret = Py_BuildValue("(NL)", bytes, written); } }
done:
  free(b.p);
  PyBuffer_Release(&rec); PyBuffer_Release(&ord); PyBuffer_Release(&rn); PyBuffer_Release(&pool);
  return ret;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * lead_columns(leads: list, cols: dict[str, writable buffer], svt: dict[str, int], src: dict[str, int], svlen_none: int,
 *              seq_none: int, ps_none: int, contig: str)
 *   -> (qnames: list[str] unique in first-seen order, ps: list[str] unique first-seen, contigs: list[str] unique first-seen, pool: bytes)
 * The input side of the drop-in (reference `Lead`, leadprov.py:34-56; `LeadProvider.record_lead`, :400-418): ONE walk over the Lead
 * objects fills the typed columns of a TaskInput (sniffles_amd/soa.py LEAD_FIELDS).  The string attributes the path only compares
 * (read_qname, phase_set, bnd_info.mate_contig) are interned to first-seen indices here (columns qname_id / ps_rank / mate_contig);
 * the caller turns those into ranks in Python string order with one sort of the unique names.  The INS sequences are appended to
 * one pool (seq_off / seq_len).  Pure marshalling: the pure-Python twin is leadprov.LeadProvider._columns_py.
 */
typedef struct { Py_buffer b; int ok; } Col;
static int col_get(PyObject* cols, const char* name, Py_ssize_t n, Py_ssize_t item, Col* c) {
  c->ok = 0;
  PyObject* o = PyDict_GetItemString(cols, name);
  if (!o) { PyErr_Format(PyExc_KeyError, "column %s missing", name); return -1; }
  if (PyObject_GetBuffer(o, &c->b, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return -1;
  c->ok = 1;
  if (c->b.len != n * item) { PyErr_Format(PyExc_ValueError, "column %s has the wrong size", name); return -1; }
  return 0;
}
static long intern_first_seen(PyObject* dict, PyObject* list, PyObject* key) {   /* index of key in list, appended if new; -1 on error */
  PyObject* v = PyDict_GetItemWithError(dict, key);
  if (v) return PyLong_AsLong(v);
  if (PyErr_Occurred()) return -1;
  const long k = (long)PyList_GET_SIZE(list);
  PyObject* kv = PyLong_FromLong(k);
  if (!kv || PyDict_SetItem(dict, key, kv) != 0 || PyList_Append(list, key) != 0) { Py_XDECREF(kv); return -1; }
  Py_DECREF(kv);
  return k;
}
static PyObject* py_lead_columns(PyObject* self, PyObject* args) {
  PyObject *leads, *cols, *svt, *src, *contig;
  long long svlen_none, seq_none, ps_none;
  if (!PyArg_ParseTuple(args, "O!O!O!O!LLLU", &PyList_Type, &leads, &PyDict_Type, &cols, &PyDict_Type, &svt, &PyDict_Type, &src,
                        &svlen_none, &seq_none, &ps_none, &contig)) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(leads);
  enum { C_RS, C_RE, C_QS, C_QE, C_SVLEN, C_RLEN, C_QN, C_RID, C_PS, C_MC, C_MP, C_SLEN, C_SOFF, C_NM, C_SVT, C_STR, C_MAPQ, C_SRC,
         C_HAP, C_SA, C_BF, C_BR, NCOL };
  static const char* names[NCOL] = {"ref_start", "ref_end", "qry_start", "qry_end", "svlen", "read_len", "qname_id", "read_id", "ps_rank",
                                    "mate_contig", "mate_ref_start", "seq_len", "seq_off", "nm", "svtype", "strand", "mapq", "source",
                                    "hap", "is_sa", "bnd_is_first", "bnd_is_reverse"};
  static const int items[NCOL] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8, 1, 1, 1, 1, 1, 1, 1, 1};
  Col c[NCOL]; memset(c, 0, sizeof(c));
  PyObject *qd = NULL, *ql = NULL, *pd = NULL, *pl = NULL, *cd = NULL, *cl = NULL, *pool = NULL, *out = NULL;
  PyObject *a_read_id = NULL, *a_qname = NULL, *a_rs = NULL, *a_re = NULL, *a_qs = NULL, *a_qe = NULL, *a_strand = NULL, *a_mapq = NULL, *a_nm = NULL,
           *a_source = NULL, *a_svtype = NULL, *a_svlen = NULL, *a_seq = NULL, *a_bnd = NULL, *a_hap = NULL, *a_ps = NULL, *a_sa = NULL, *a_rlen = NULL,
           *a_mc = NULL, *a_mp = NULL, *a_bf = NULL, *a_br = NULL;
  int fail = 1;
  for (int k = 0; k < NCOL; k++) if (col_get(cols, names[k], n, items[k], &c[k]) != 0) goto done;
#define ATTR(var, nm) if (!(var = PyUnicode_InternFromString(nm))) goto done;
  ATTR(a_read_id, "read_id") ATTR(a_qname, "read_qname") ATTR(a_rs, "ref_start") ATTR(a_re, "ref_end") ATTR(a_qs, "qry_start") ATTR(a_qe, "qry_end")
  ATTR(a_strand, "strand") ATTR(a_mapq, "mapq") ATTR(a_nm, "nm") ATTR(a_source, "source") ATTR(a_svtype, "svtype") ATTR(a_svlen, "svlen")
  ATTR(a_seq, "seq") ATTR(a_bnd, "bnd_info") ATTR(a_hap, "hap") ATTR(a_ps, "phase_set") ATTR(a_sa, "is_sa") ATTR(a_rlen, "read_len")
  ATTR(a_mc, "mate_contig") ATTR(a_mp, "mate_ref_start") ATTR(a_bf, "is_first") ATTR(a_br, "is_reverse")
  qd = PyDict_New(); ql = PyList_New(0); pd = PyDict_New(); pl = PyList_New(0); cd = PyDict_New(); cl = PyList_New(0);
  if (!qd || !ql || !pd || !pl || !cd || !cl) goto done;
  if (intern_first_seen(cd, cl, contig) < 0) goto done;            /* the task's own contig is always in the table (index 0) */
  {
    /* pass 1: sequence bytes */
    int64_t total = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
      PyObject* q = PyObject_GetAttr(PyList_GET_ITEM(leads, i), a_seq);
      if (!q) goto done;
      if (q != Py_None) {
        if (!PyUnicode_Check(q)) { Py_DECREF(q); PyErr_SetString(PyExc_TypeError, "Lead.seq must be str or None"); goto done; }
        if (PyUnicode_MAX_CHAR_VALUE(q) > 255) { Py_DECREF(q); PyErr_SetString(PyExc_ValueError, "Lead.seq must be latin-1"); goto done; }
        total += (int64_t)PyUnicode_GET_LENGTH(q);
      }
      Py_DECREF(q);
    }
    pool = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (!pool) goto done;
    char* pb = PyBytes_AS_STRING(pool);
    int64_t off = 0;
    int32_t *rs = (int32_t*)c[C_RS].b.buf, *re = (int32_t*)c[C_RE].b.buf, *qs = (int32_t*)c[C_QS].b.buf, *qe = (int32_t*)c[C_QE].b.buf,
            *svl = (int32_t*)c[C_SVLEN].b.buf, *rl = (int32_t*)c[C_RLEN].b.buf, *psr = (int32_t*)c[C_PS].b.buf, *mc = (int32_t*)c[C_MC].b.buf,
            *mp = (int32_t*)c[C_MP].b.buf, *sl = (int32_t*)c[C_SLEN].b.buf;
    uint32_t *qn = (uint32_t*)c[C_QN].b.buf, *rid = (uint32_t*)c[C_RID].b.buf;
    int64_t* so = (int64_t*)c[C_SOFF].b.buf;
    double* nm = (double*)c[C_NM].b.buf;
    uint8_t *svtc = (uint8_t*)c[C_SVT].b.buf, *str = (uint8_t*)c[C_STR].b.buf, *mq = (uint8_t*)c[C_MAPQ].b.buf, *srcc = (uint8_t*)c[C_SRC].b.buf,
            *hp = (uint8_t*)c[C_HAP].b.buf, *sa = (uint8_t*)c[C_SA].b.buf, *bf = (uint8_t*)c[C_BF].b.buf, *br = (uint8_t*)c[C_BR].b.buf;
#define GET(var, attr) PyObject* var = PyObject_GetAttr(ld, attr); if (!var) goto done;
#define AS_I32(dst, obj) { const long long _x = PyLong_AsLongLong(obj); Py_DECREF(obj); if (_x == -1 && PyErr_Occurred()) goto done; dst = (int32_t)_x; }
    for (Py_ssize_t i = 0; i < n; i++) {
      PyObject* ld = PyList_GET_ITEM(leads, i);
      { GET(o, a_rs) AS_I32(rs[i], o) } { GET(o, a_re) AS_I32(re[i], o) } { GET(o, a_qs) AS_I32(qs[i], o) } { GET(o, a_qe) AS_I32(qe[i], o) }
      { GET(o, a_svlen) if (o == Py_None) { Py_DECREF(o); svl[i] = (int32_t)svlen_none; } else AS_I32(svl[i], o) }
      { GET(o, a_rlen) if (o == Py_None) { Py_DECREF(o); rl[i] = 0; } else AS_I32(rl[i], o) }
      { GET(o, a_read_id) const unsigned long long x = PyLong_AsUnsignedLongLongMask(o); Py_DECREF(o); if (PyErr_Occurred()) goto done; rid[i] = (uint32_t)x; }
      { GET(o, a_mapq) AS_I32(psr[i], o) mq[i] = (uint8_t)psr[i]; }
      { GET(o, a_nm) if (o == Py_None) { nm[i] = Py_NAN; Py_DECREF(o); } else { const double x = PyFloat_AsDouble(o); Py_DECREF(o); if (x == -1.0 && PyErr_Occurred()) goto done; nm[i] = x; } }
      { GET(o, a_strand) const int minus = PyUnicode_Check(o) && PyUnicode_GET_LENGTH(o) == 1 && PyUnicode_READ_CHAR(o, 0) == '-'; Py_DECREF(o); str[i] = minus ? 1 : 0; }
      { GET(o, a_sa) const int t = PyObject_IsTrue(o); Py_DECREF(o); if (t < 0) goto done; sa[i] = (uint8_t)t; }
      { GET(o, a_hap) PyObject* h = PyNumber_Long(o); Py_DECREF(o); if (!h) goto done; const long x = PyLong_AsLong(h); Py_DECREF(h); if (x == -1 && PyErr_Occurred()) goto done; hp[i] = (uint8_t)x; }
      { GET(o, a_svtype) PyObject* k = PyDict_GetItemWithError(svt, o); Py_DECREF(o); if (!k) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_KeyError, "unknown svtype"); goto done; } svtc[i] = (uint8_t)PyLong_AsLong(k); }
      { GET(o, a_source) PyObject* k = PyDict_GetItemWithError(src, o); Py_DECREF(o); if (!k) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_KeyError, "unknown lead source"); goto done; } srcc[i] = (uint8_t)PyLong_AsLong(k); }
      { GET(o, a_qname) const long k = intern_first_seen(qd, ql, o); Py_DECREF(o); if (k < 0) goto done; qn[i] = (uint32_t)k; }
      { GET(o, a_ps) if (o == Py_None) { psr[i] = (int32_t)ps_none; Py_DECREF(o); } else { const long k = intern_first_seen(pd, pl, o); Py_DECREF(o); if (k < 0) goto done; psr[i] = (int32_t)k; } }
      { GET(q, a_seq)
        if (q == Py_None) { sl[i] = (int32_t)seq_none; so[i] = 0; }
        else {
          const Py_ssize_t L = PyUnicode_GET_LENGTH(q);
          sl[i] = (int32_t)L; so[i] = off;
          if (PyUnicode_KIND(q) == PyUnicode_1BYTE_KIND) memcpy(pb + off, PyUnicode_1BYTE_DATA(q), (size_t)L);
          else for (Py_ssize_t k = 0; k < L; k++) pb[off + k] = (char)PyUnicode_READ_CHAR(q, k);
          off += L;
        }
        Py_DECREF(q); }
      mc[i] = -1; mp[i] = 0; bf[i] = 0; br[i] = 0;      /* -1: no bnd_info */
      { GET(bi, a_bnd)
        if (bi != Py_None) {
          PyObject* x = PyObject_GetAttr(bi, a_mc); if (!x) { Py_DECREF(bi); goto done; }
          const long k = intern_first_seen(cd, cl, x); Py_DECREF(x); if (k < 0) { Py_DECREF(bi); goto done; }
          mc[i] = (int32_t)k;
          x = PyObject_GetAttr(bi, a_mp); if (!x) { Py_DECREF(bi); goto done; }
          { const long long y = PyLong_AsLongLong(x); Py_DECREF(x); if (y == -1 && PyErr_Occurred()) { Py_DECREF(bi); goto done; } mp[i] = (int32_t)y; }
          x = PyObject_GetAttr(bi, a_bf); if (!x) { Py_DECREF(bi); goto done; } bf[i] = (uint8_t)(PyObject_IsTrue(x) == 1); Py_DECREF(x);
          x = PyObject_GetAttr(bi, a_br); if (!x) { Py_DECREF(bi); goto done; } br[i] = (uint8_t)(PyObject_IsTrue(x) == 1); Py_DECREF(x);
        }
        Py_DECREF(bi); }
    }
#undef GET
#undef AS_I32
  }
  out = Py_BuildValue("(OOOO)", ql, pl, cl, pool);
  fail = 0;
done:
  for (int k = 0; k < NCOL; k++) if (c[k].ok) PyBuffer_Release(&c[k].b);
  Py_XDECREF(qd); Py_XDECREF(ql); Py_XDECREF(pd); Py_XDECREF(pl); Py_XDECREF(cd); Py_XDECREF(cl); Py_XDECREF(pool);
  Py_XDECREF(a_read_id); Py_XDECREF(a_qname); Py_XDECREF(a_rs); Py_XDECREF(a_re); Py_XDECREF(a_qs); Py_XDECREF(a_qe); Py_XDECREF(a_strand);
  Py_XDECREF(a_mapq); Py_XDECREF(a_nm); Py_XDECREF(a_source); Py_XDECREF(a_svtype); Py_XDECREF(a_svlen); Py_XDECREF(a_seq); Py_XDECREF(a_bnd);
  Py_XDECREF(a_hap); Py_XDECREF(a_ps); Py_XDECREF(a_sa); Py_XDECREF(a_rlen); Py_XDECREF(a_mc); Py_XDECREF(a_mp); Py_XDECREF(a_bf); Py_XDECREF(a_br);
  if (fail) { Py_XDECREF(out); return NULL; }
  return out;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * LeadSink: `LeadProvider.record_lead` (reference leadprov.py:400-418) as the place where a `Lead` becomes columns.  The reference
 * pays its binning when a lead is recorded (dict / list inserts, the hap counters); here `record(ld)` reads the object's attributes
 * ONCE, appends the 22 typed values to growable buffers, interns read_qname / phase_set / bnd_info.mate_contig to first-seen indices
 * and appends `seq` to the sequence pool - what `lead_columns` does for a whole list, one lead at a time - so that
 * `LeadProvider.to_task_input` is a copy of finished columns (`take`) plus the string-rank remap instead of a walk over ~10^5 objects
 * inside `Task.call_candidates`.  The lead is a snapshot at record time (the reference's `record_lead` is called with the finished lead).
 *   LeadSink(svt: dict, src: dict, svlen_none, seq_none, ps_none, contig)
 *   .record(ld[, pos_leadtab])      .take(cols: dict[str, writable buffer of len(self) rows]) -> (qnames, ps, contigs, pool: bytes)
 */
enum { LC_RS, LC_RE, LC_QS, LC_QE, LC_SVLEN, LC_RLEN, LC_QN, LC_RID, LC_PS, LC_MC, LC_MP, LC_SLEN, LC_SOFF, LC_NM, LC_SVT, LC_STR, LC_MAPQ, LC_SRC,
       LC_HAP, LC_SA, LC_BF, LC_BR, LC_NCOL };
static const char* LC_NAMES[LC_NCOL] = {"ref_start", "ref_end", "qry_start", "qry_end", "svlen", "read_len", "qname_id", "read_id", "ps_rank",
                                        "mate_contig", "mate_ref_start", "seq_len", "seq_off", "nm", "svtype", "strand", "mapq", "source",
                                        "hap", "is_sa", "bnd_is_first", "bnd_is_reverse"};
static const int LC_ITEMS[LC_NCOL] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 8, 8, 1, 1, 1, 1, 1, 1, 1, 1};
static PyObject *A_read_id, *A_qname, *A_rs, *A_re, *A_qs, *A_qe, *A_strand, *A_mapq, *A_nm, *A_source, *A_svtype, *A_svlen, *A_seq, *A_bnd, *A_hap,
                *A_ps, *A_sa, *A_rlen;
typedef struct {
  PyObject_HEAD
  char* col[LC_NCOL];
  Py_ssize_t n, cap;
  char* pool; int64_t pool_n, pool_cap;
  PyObject *qd, *ql, *pd, *pl, *cd, *cl, *svt, *src;
  long long svlen_none, seq_none, ps_none;
} LeadSink;

static int sink_grow(LeadSink* S) {
  const Py_ssize_t cap = S->cap ? S->cap * 2 : 1024;
  for (int k = 0; k < LC_NCOL; k++) {
    char* q = (char*)realloc(S->col[k], (size_t)cap * (size_t)LC_ITEMS[k]);
    if (!q) { PyErr_NoMemory(); return -1; }
    S->col[k] = q;
  }
  S->cap = cap;
  return 0;
}
static int sink_pool_room(LeadSink* S, int64_t extra) {
  if (S->pool_n + extra <= S->pool_cap) return 0;
  int64_t cap = S->pool_cap ? S->pool_cap * 2 : (1 << 16);
  while (cap < S->pool_n + extra) cap *= 2;
  char* q = (char*)realloc(S->pool, (size_t)cap);
  if (!q) { PyErr_NoMemory(); return -1; }
  S->pool = q; S->pool_cap = cap;
  return 0;
}
/* one Lead -> row i of the sink's columns (the body of lead_columns' walk) */
static int sink_row(LeadSink* S, PyObject* ld, Py_ssize_t i) {
  int32_t *rs = (int32_t*)S->col[LC_RS], *re = (int32_t*)S->col[LC_RE], *qs = (int32_t*)S->col[LC_QS], *qe = (int32_t*)S->col[LC_QE],
          *svl = (int32_t*)S->col[LC_SVLEN], *rl = (int32_t*)S->col[LC_RLEN], *psr = (int32_t*)S->col[LC_PS], *mc = (int32_t*)S->col[LC_MC],
          *mp = (int32_t*)S->col[LC_MP], *sl = (int32_t*)S->col[LC_SLEN];
  uint32_t *qn = (uint32_t*)S->col[LC_QN], *rid = (uint32_t*)S->col[LC_RID];
  int64_t* so = (int64_t*)S->col[LC_SOFF];
  double* nm = (double*)S->col[LC_NM];
  uint8_t *svtc = (uint8_t*)S->col[LC_SVT], *str = (uint8_t*)S->col[LC_STR], *mq = (uint8_t*)S->col[LC_MAPQ], *srcc = (uint8_t*)S->col[LC_SRC],
          *hp = (uint8_t*)S->col[LC_HAP], *sa = (uint8_t*)S->col[LC_SA], *bf = (uint8_t*)S->col[LC_BF], *br = (uint8_t*)S->col[LC_BR];
#define GET(var, attr) PyObject* var = PyObject_GetAttr(ld, attr); if (!var) return -1;
#define AS_I32(dst, obj) { const long long _x = PyLong_AsLongLong(obj); Py_DECREF(obj); if (_x == -1 && PyErr_Occurred()) return -1; dst = (int32_t)_x; }
  { GET(o, A_rs) AS_I32(rs[i], o) } { GET(o, A_re) AS_I32(re[i], o) } { GET(o, A_qs) AS_I32(qs[i], o) } { GET(o, A_qe) AS_I32(qe[i], o) }
  { GET(o, A_svlen) if (o == Py_None) { Py_DECREF(o); svl[i] = (int32_t)S->svlen_none; } else AS_I32(svl[i], o) }
  { GET(o, A_rlen) if (o == Py_None) { Py_DECREF(o); rl[i] = 0; } else AS_I32(rl[i], o) }
  { GET(o, A_read_id) const unsigned long long x = PyLong_AsUnsignedLongLongMask(o); Py_DECREF(o); if (PyErr_Occurred()) return -1; rid[i] = (uint32_t)x; }
  { GET(o, A_mapq) int32_t m; AS_I32(m, o) mq[i] = (uint8_t)m; }
  { GET(o, A_nm) if (o == Py_None) { nm[i] = Py_NAN; Py_DECREF(o); } else { const double x = PyFloat_AsDouble(o); Py_DECREF(o); if (x == -1.0 && PyErr_Occurred()) return -1; nm[i] = x; } }
  { GET(o, A_strand) const int minus = PyUnicode_Check(o) && PyUnicode_GET_LENGTH(o) == 1 && PyUnicode_READ_CHAR(o, 0) == '-'; Py_DECREF(o); str[i] = minus ? 1 : 0; }
  { GET(o, A_sa) const int t = PyObject_IsTrue(o); Py_DECREF(o); if (t < 0) return -1; sa[i] = (uint8_t)t; }
  { GET(o, A_hap) PyObject* h = PyNumber_Long(o); Py_DECREF(o); if (!h) return -1; const long x = PyLong_AsLong(h); Py_DECREF(h); if (x == -1 && PyErr_Occurred()) return -1; hp[i] = (uint8_t)x; }
  { GET(o, A_svtype) PyObject* k = PyDict_GetItemWithError(S->svt, o); Py_DECREF(o); if (!k) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_KeyError, "unknown svtype"); return -1; } svtc[i] = (uint8_t)PyLong_AsLong(k); }
  { GET(o, A_source) PyObject* k = PyDict_GetItemWithError(S->src, o); Py_DECREF(o); if (!k) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_KeyError, "unknown lead source"); return -1; } srcc[i] = (uint8_t)PyLong_AsLong(k); }
  { GET(o, A_qname) const long k = intern_first_seen(S->qd, S->ql, o); Py_DECREF(o); if (k < 0) return -1; qn[i] = (uint32_t)k; }
  { GET(o, A_ps) if (o == Py_None) { psr[i] = (int32_t)S->ps_none; Py_DECREF(o); } else { const long k = intern_first_seen(S->pd, S->pl, o); Py_DECREF(o); if (k < 0) return -1; psr[i] = (int32_t)k; } }
  { GET(q, A_seq)
    if (q == Py_None) { sl[i] = (int32_t)S->seq_none; so[i] = 0; }
    else {
      if (!PyUnicode_Check(q)) { Py_DECREF(q); PyErr_SetString(PyExc_TypeError, "Lead.seq must be str or None"); return -1; }
      if (PyUnicode_MAX_CHAR_VALUE(q) > 255) { Py_DECREF(q); PyErr_SetString(PyExc_ValueError, "Lead.seq must be latin-1"); return -1; }
      const Py_ssize_t L = PyUnicode_GET_LENGTH(q);
      if (sink_pool_room(S, (int64_t)L) != 0) { Py_DECREF(q); return -1; }
      sl[i] = (int32_t)L; so[i] = S->pool_n;
      if (PyUnicode_KIND(q) == PyUnicode_1BYTE_KIND) memcpy(S->pool + S->pool_n, PyUnicode_1BYTE_DATA(q), (size_t)L);
      else for (Py_ssize_t k = 0; k < L; k++) S->pool[S->pool_n + k] = (char)PyUnicode_READ_CHAR(q, k);
      S->pool_n += L;
    }
    Py_DECREF(q); }
  mc[i] = -1; mp[i] = 0; bf[i] = 0; br[i] = 0;      /* -1: no bnd_info */
  { GET(bi, A_bnd)
    if (bi != Py_None) {
      PyObject* x = PyObject_GetAttr(bi, B_mate_contig); if (!x) { Py_DECREF(bi); return -1; }
      const long k = intern_first_seen(S->cd, S->cl, x); Py_DECREF(x); if (k < 0) { Py_DECREF(bi); return -1; }
      mc[i] = (int32_t)k;
      x = PyObject_GetAttr(bi, B_mate_ref_start); if (!x) { Py_DECREF(bi); return -1; }
      { const long long y = PyLong_AsLongLong(x); Py_DECREF(x); if (y == -1 && PyErr_Occurred()) { Py_DECREF(bi); return -1; } mp[i] = (int32_t)y; }
      x = PyObject_GetAttr(bi, B_is_first); if (!x) { Py_DECREF(bi); return -1; } bf[i] = (uint8_t)(PyObject_IsTrue(x) == 1); Py_DECREF(x);
      x = PyObject_GetAttr(bi, B_is_reverse); if (!x) { Py_DECREF(bi); return -1; } br[i] = (uint8_t)(PyObject_IsTrue(x) == 1); Py_DECREF(x);
    }
    Py_DECREF(bi); }
#undef GET
#undef AS_I32
  return 0;
}
static void sink_dealloc(LeadSink* S) {
  for (int k = 0; k < LC_NCOL; k++) free(S->col[k]);
  free(S->pool);
  Py_XDECREF(S->qd); Py_XDECREF(S->ql); Py_XDECREF(S->pd); Py_XDECREF(S->pl); Py_XDECREF(S->cd); Py_XDECREF(S->cl); Py_XDECREF(S->svt); Py_XDECREF(S->src);
  Py_TYPE(S)->tp_free((PyObject*)S);
}
static int sink_init(LeadSink* S, PyObject* args, PyObject* kw) {
  PyObject *svt, *src, *contig;
  if (!PyArg_ParseTuple(args, "O!O!LLLU", &PyDict_Type, &svt, &PyDict_Type, &src, &S->svlen_none, &S->seq_none, &S->ps_none, &contig)) return -1;
  Py_INCREF(svt); Py_INCREF(src); S->svt = svt; S->src = src;
  S->qd = PyDict_New(); S->ql = PyList_New(0); S->pd = PyDict_New(); S->pl = PyList_New(0); S->cd = PyDict_New(); S->cl = PyList_New(0);
  if (!S->qd || !S->ql || !S->pd || !S->pl || !S->cd || !S->cl) return -1;
  if (intern_first_seen(S->cd, S->cl, contig) < 0) return -1;      /* the task's own contig is always in the table (index 0) */
  return 0;
}
static PyObject* sink_record(LeadSink* S, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs < 1 || nargs > 2) { PyErr_SetString(PyExc_TypeError, "record(lead[, pos_leadtab])"); return NULL; }
  if (!S->qd) { PyErr_SetString(PyExc_RuntimeError, "LeadSink was not initialised"); return NULL; }
  if (S->n == S->cap && sink_grow(S) != 0) return NULL;
  const int64_t pool_before = S->pool_n;
  if (sink_row(S, args[0], S->n) != 0) { S->pool_n = pool_before; return NULL; }     /* (a lead that fails leaves no row) */
  S->n++;
  Py_RETURN_NONE;
}
static PyObject* sink_take(LeadSink* S, PyObject* cols) {
  if (!PyDict_Check(cols)) { PyErr_SetString(PyExc_TypeError, "take(cols: dict)"); return NULL; }
  Col c[LC_NCOL]; memset(c, 0, sizeof(c));
  PyObject* out = NULL;
  for (int k = 0; k < LC_NCOL; k++) if (col_get(cols, LC_NAMES[k], S->n, LC_ITEMS[k], &c[k]) != 0) goto done;
  for (int k = 0; k < LC_NCOL; k++) if (S->n) memcpy(c[k].b.buf, S->col[k], (size_t)S->n * (size_t)LC_ITEMS[k]);
  {
    PyObject* pool = PyBytes_FromStringAndSize(S->pool ? S->pool : "", (Py_ssize_t)S->pool_n);
    if (!pool) goto done;
    out = Py_BuildValue("(OOON)", S->ql, S->pl, S->cl, pool);
  }
done:
  for (int k = 0; k < LC_NCOL; k++) if (c[k].ok) PyBuffer_Release(&c[k].b);
  return out;
}
static Py_ssize_t sink_len(LeadSink* S) { return S->n; }
static PyMethodDef sink_methods[] = {
    {"record", (PyCFunction)(void (*)(void))sink_record, METH_FASTCALL, "append one Lead (attributes read now)"},
    {"take", (PyCFunction)sink_take, METH_O, "copy the columns into the given buffers -> (qnames, ps, contigs, pool)"},
    {NULL, NULL, 0, NULL}};
static PySequenceMethods sink_seq = {(lenfunc)sink_len};
static PyTypeObject LeadSinkType = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "_snf_fast.LeadSink", .tp_basicsize = sizeof(LeadSink), .tp_flags = Py_TPFLAGS_DEFAULT,
    .tp_new = PyType_GenericNew, .tp_init = (initproc)sink_init, .tp_dealloc = (destructor)sink_dealloc, .tp_methods = sink_methods,
    .tp_as_sequence = &sink_seq, .tp_doc = "record-time columns of LeadProvider.record_lead"};

/* rank_strings(names: list[str]) -> (sorted: list[str], rank: bytes int64[n]): the names in Python string order (code points; equal
 * to the byte order of UTF-8) and rank[i] = place of names[i] in it; names must be distinct (an interning table).  The remap of
 * first-seen indices to ranks whose integer order is the reference's string order (leadprov: read_qname, phase_set, mate contig).
 * An LSD radix sort over the first eight bytes (big-endian, zero-padded), then the runs of equal prefixes by comparison. */
typedef struct { uint64_t pre; const char* p; Py_ssize_t len; int64_t idx; } RankKey;
static int rank_cmp(const void* a, const void* b) {
  const RankKey *x = (const RankKey*)a, *y = (const RankKey*)b;
  if (x->pre != y->pre) return x->pre < y->pre ? -1 : 1;
  const Py_ssize_t m = x->len < y->len ? x->len : y->len;
  const int c = m > 8 ? memcmp(x->p + 8, y->p + 8, (size_t)(m - 8)) : 0;
  if (c) return c;
  return x->len < y->len ? -1 : (x->len > y->len ? 1 : 0);
}
static PyObject* py_rank_strings(PyObject* self, PyObject* names) {
  if (!PyList_Check(names)) { PyErr_SetString(PyExc_TypeError, "rank_strings(list[str])"); return NULL; }
  const Py_ssize_t n = PyList_GET_SIZE(names);
  RankKey *k = (RankKey*)malloc((size_t)(n ? n : 1) * sizeof(RankKey)), *k2 = (RankKey*)malloc((size_t)(n ? n : 1) * sizeof(RankKey));
  PyObject *sorted = PyList_New(n), *rank = PyBytes_FromStringAndSize(NULL, n * 8), *out = NULL;
  if (!k || !k2 || !sorted || !rank) { PyErr_NoMemory(); goto done; }
  uint64_t all_or = 0, all_and = ~0ull;
  for (Py_ssize_t i = 0; i < n; i++) {
    PyObject* o = PyList_GET_ITEM(names, i);
    if (!PyUnicode_Check(o)) { PyErr_SetString(PyExc_TypeError, "rank_strings: names must be str"); goto done; }
    k[i].p = PyUnicode_AsUTF8AndSize(o, &k[i].len);
    if (!k[i].p) goto done;
    uint64_t pre = 0;
    for (int b = 0; b < 8; b++) pre = (pre << 8) | (uint64_t)(b < k[i].len ? (unsigned char)k[i].p[b] : 0);
    k[i].pre = pre; k[i].idx = i;
    all_or |= pre; all_and &= pre;
  }
  if (n > 64) {
    for (int byte = 0; byte < 8; byte++) {      /* least significant byte first; a byte all keys share is skipped */
      const int sh = 8 * byte;
      if ((((all_or ^ all_and) >> sh) & 0xffu) == 0) continue;
      size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
      for (Py_ssize_t i = 0; i < n; i++) cnt[((k[i].pre >> sh) & 0xffu) + 1]++;
      for (int c = 0; c < 256; c++) cnt[c + 1] += cnt[c];
      for (Py_ssize_t i = 0; i < n; i++) k2[cnt[(k[i].pre >> sh) & 0xffu]++] = k[i];
      RankKey* t = k; k = k2; k2 = t;
    }
    for (Py_ssize_t a = 0; a < n;) {            /* runs of equal prefixes: longer names that share eight bytes */
      Py_ssize_t b = a + 1;
      while (b < n && k[b].pre == k[a].pre) b++;
      if (b - a > 1) qsort(k + a, (size_t)(b - a), sizeof(RankKey), rank_cmp);
      a = b;
    }
  } else qsort(k, (size_t)n, sizeof(RankKey), rank_cmp);
  {
    int64_t* rd = (int64_t*)PyBytes_AS_STRING(rank);
    for (Py_ssize_t j = 0; j < n; j++) {
      rd[k[j].idx] = j;
      PyObject* o = PyList_GET_ITEM(names, k[j].idx);
      Py_INCREF(o); PyList_SET_ITEM(sorted, j, o);
    }
  }
  out = Py_BuildValue("(OO)", sorted, rank);
done:
  free(k); free(k2); Py_XDECREF(sorted); Py_XDECREF(rank);
  return out;
}

/* format_f3(values: buffer float64) -> bytes: the values as this module prints them (ob_f3), '\n' between: lets the test suite pin the
 * hand-written "%.3f" against Python's own f"{v:.3f}" */
static PyObject* py_format_f3(PyObject* self, PyObject* args) {
  Py_buffer vb;
  if (!PyArg_ParseTuple(args, "y*", &vb)) return NULL;
  OutBuf b = {NULL, 0, 0}; PyObject* ret = NULL; int bad = 0;
  for (Py_ssize_t i = 0; !bad && i < vb.len / 8; i++) bad = ob_f3(&b, ((const double*)vb.buf)[i]) || OB_LIT(&b, "\n");
  if (!bad) ret = PyBytes_FromStringAndSize(b.p ? b.p : "", (Py_ssize_t)b.n);
  free(b.p); PyBuffer_Release(&vb);
  return ret;
}

static PyMethodDef methods[] = {
    {"format_f3", py_format_f3, METH_VARARGS, "float64 values as f\"{v:.3f}\" lines (test hook of the formatter)"},
    {"rank_strings", py_rank_strings, METH_O, "order and rank of distinct names in Python string order"},
    {"lead_columns", py_lead_columns, METH_VARARGS, "Lead objects -> typed TaskInput columns in one walk (input side of the drop-in)"},
    {"materialize", py_materialize, METH_VARARGS, "records [lo, hi) -> list of SVCall objects (candidate-stage fields)"},
    {"make_stubs", py_make_stubs, METH_VARARGS, "lazy stand-ins of the calls of a record range"},
    {"stub_select", py_stub_select, METH_VARARGS, "the stand-ins of a source in a list, and their places"},
    {"stub_others", py_stub_others, METH_VARARGS, "the elements of a list that are not stand-ins of a source"},
    {"stub_refresh_qc", py_stub_refresh_qc, METH_VARARGS, "qc of the final records onto the stand-ins of a list"},
    {"apply_final", py_apply_final, METH_VARARGS, "finalize-stage fields of the records onto materialised calls"},
    {"collect", py_collect, METH_VARARGS, "SVCall objects of SNF blocks -> candidate records, ALT pool, BND mates"},
    {"gather_pool", py_gather_pool, METH_VARARGS, "strings of a pool in a given order, back to back"},
    {"argsort_i64", py_argsort_i64, METH_VARARGS, "stable ascending order of int64 keys (radix sort)"},
    {"gather_pool_parts", py_gather_pool_parts, METH_VARARGS, "gather_pool over strings that lie in several pools"},
    {"flush_windows", py_flush_windows, METH_VARARGS, "flush windows of CombineTask.execute over the sorted candidate table"},
    {"group_calls", py_group_calls, METH_VARARGS, "group records + membership -> combined SVCall objects"},
    {"vcf_records", py_vcf_records, METH_VARARGS, "single-sample VCF lines straight from the record table"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_snf_fast", "C materialiser of sniffles_amd.sv", -1, methods};

#define INTERN(var, s) do { var = PyUnicode_InternFromString(s); if (!var) return NULL; } while (0)
PyMODINIT_FUNC PyInit__snf_fast(void) {
  if (sizeof(snf_call_t) != 240) { PyErr_SetString(PyExc_ImportError, "snf_call_t layout changed"); return NULL; }
  for (int t = 0; t < 7; t++) {
    INTERN(S_svtype[t], SVTYPES[t]);
    char b[32]; snprintf(b, sizeof b, "<%s>", SVTYPES[t]);
    INTERN(S_alt_sym[t], b);
  }
  INTERN(S_N, "N"); INTERN(S_NULL, "NULL"); INTERN(S_PASS, "PASS"); INTERN(S_FAIL, "FAIL");
  INTERN(K_contig, "contig"); INTERN(K_pos, "pos"); INTERN(K_id, "id"); INTERN(K_ref, "ref"); INTERN(K_alt, "alt"); INTERN(K_qual, "qual");
  INTERN(K_filter, "filter"); INTERN(K_info, "info"); INTERN(K_svtype, "svtype"); INTERN(K_svlen, "svlen"); INTERN(K_end, "end");
  INTERN(K_genotypes, "genotypes"); INTERN(K_precise, "precise"); INTERN(K_support, "support"); INTERN(K_rnames, "rnames"); INTERN(K_qc, "qc");
  INTERN(K_nm, "nm"); INTERN(K_postprocess, "postprocess"); INTERN(K_svlens, "svlens"); INTERN(K_fwd, "fwd"); INTERN(K_rev, "rev");
  INTERN(K_fds, "forward_difference_sampler"); INTERN(K_cov_up, "coverage_upstream"); INTERN(K_cov_dn, "coverage_downstream");
  INTERN(K_cov_st, "coverage_start"); INTERN(K_cov_ce, "coverage_center"); INTERN(K_cov_en, "coverage_end");
  INTERN(K_sample, "sample_internal_id"); INTERN(K_bnd_info, "bnd_info"); INTERN(K_sup_inline, "support_inline");
  INTERN(K_sup_splits, "support_splits"); INTERN(K_raw, "raw_vcf_line"); INTERN(K_raw_idx, "raw_vcf_line_index");
  INTERN(I_CHR2, "CHR2"); INTERN(I_SUPPORT_LONG, "SUPPORT_LONG"); INTERN(I_SUPPORT_SA, "SUPPORT_SA"); INTERN(I_STDEV_POS, "STDEV_POS");
  INTERN(I_STDEV_LEN, "STDEV_LEN"); INTERN(I_COVERAGE_VAR, "COVERAGE_VAR"); INTERN(I_PHASE, "PHASE"); INTERN(I_VAF, "VAF");
  INTERN(B_mate_contig, "mate_contig"); INTERN(B_mate_ref_start, "mate_ref_start"); INTERN(B_is_first, "is_first"); INTERN(B_is_reverse, "is_reverse");
  INTERN(P_batch, "batch"); INTERN(P_index, "index"); INTERN(K_class, "__class__"); INTERN(K_lz, "_lz"); INTERN(K_lzi, "_lzi");
  INTERN(S_dot, "."); INTERN(S_comma, ","); INTERN(I_COVERAGE, "_COVERAGE");
  O_zero = PyLong_FromLong(0); T_none2 = PyTuple_Pack(2, Py_None, Py_None);
  if (!O_zero || !T_none2 || sizeof(snf_group_cand_t) != 76 || sizeof(snf_group_out_t) != 96) { PyErr_SetString(PyExc_ImportError, "group record layout changed"); return NULL; }
  INTERN(F_n, "n"); INTERN(F_m1, "m1"); INTERN(F_m2, "m2"); INTERN(F_last, "last");
  INTERN(A_read_id, "read_id"); INTERN(A_qname, "read_qname"); INTERN(A_rs, "ref_start"); INTERN(A_re, "ref_end"); INTERN(A_qs, "qry_start");
  INTERN(A_qe, "qry_end"); INTERN(A_strand, "strand"); INTERN(A_mapq, "mapq"); INTERN(A_nm, "nm"); INTERN(A_source, "source");
  INTERN(A_svtype, "svtype"); INTERN(A_svlen, "svlen"); INTERN(A_seq, "seq"); INTERN(A_bnd, "bnd_info"); INTERN(A_hap, "hap");
  INTERN(A_ps, "phase_set"); INTERN(A_sa, "is_sa"); INTERN(A_rlen, "read_len");
  if (PyType_Ready(&LeadSinkType) < 0) return NULL;
  PyObject* mod = PyModule_Create(&moddef);
  if (!mod) return NULL;
  Py_INCREF(&LeadSinkType);
  if (PyModule_AddObject(mod, "LeadSink", (PyObject*)&LeadSinkType) < 0) { Py_DECREF(&LeadSinkType); Py_DECREF(mod); return NULL; }
  return mod;
}
