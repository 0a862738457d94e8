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
static PyObject *B_mate_contig, *B_mate_ref_start, *B_is_first, *B_is_reverse, *P_batch, *P_index, *S_NULL, *S_PASS, *S_FAIL;

static int set_steal(PyObject* d, PyObject* k, PyObject* v) {   /* d[k] = v, steals v */
  if (!v) return -1;
  int rc = PyDict_SetItem(d, k, v);
  Py_DECREF(v);
  return rc;
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
  PyObject *cls, *bnd_cls, *fds_cls, *post_cls, *batch, *qnames, *contig, *contig_names, *filters;
  Py_buffer calls, rn;
  long long lo, hi, task_id;
  if (!PyArg_ParseTuple(args, "OOOOOy*LLy*OOLOO", &cls, &bnd_cls, &fds_cls, &post_cls, &batch, &calls, &lo, &hi, &rn, &qnames, &contig,
                        &task_id, &contig_names, &filters))
    return NULL;
  PyObject* out = NULL;
  if (lo < 0 || hi < lo || (size_t)hi * sizeof(snf_call_t) > (size_t)calls.len) { PyErr_SetString(PyExc_ValueError, "call range outside the record table"); goto done; }
  const snf_call_t* C = (const snf_call_t*)calls.buf;
  const uint32_t* RN = (const uint32_t*)rn.buf;
  const long long rn_n = rn.len / 4;
  out = PyList_New(hi - lo);
  if (!out) goto done;
  for (long long i = lo; i < hi; i++) {
    const snf_call_t* c = &C[i];
    if (c->svtype < 0 || c->svtype > 6 || c->filter < 0 || c->filter >= PyList_GET_SIZE(filters)) { PyErr_SetString(PyExc_ValueError, "record field out of range"); goto fail; }
    PyObject* d = PyDict_New();
    if (!d) goto fail;
    PyObject* info = PyDict_New();
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
    char idbuf[64];
    snprintf(idbuf, sizeof idbuf, "%s.%XS%llX", SVTYPES[c->svtype], (unsigned)c->sv_id, (unsigned long long)task_id);
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
      bad = PyDict_SetItem(d, K_contig, contig) || set_steal(d, K_pos, PyLong_FromLong(c->pos)) || set_steal(d, K_id, PyUnicode_FromString(idbuf)) ||
            PyDict_SetItem(d, K_ref, S_N) || PyDict_SetItem(d, K_alt, alt) || set_steal(d, K_qual, PyLong_FromLong(c->qual)) ||
            PyDict_SetItem(d, K_filter, PyList_GET_ITEM(filters, c->filter)) || PyDict_SetItem(d, K_info, info) ||
            PyDict_SetItem(d, K_svtype, S_svtype[c->svtype]) || set_steal(d, K_svlen, PyLong_FromLong(c->svlen)) ||
            set_steal(d, K_end, PyLong_FromLong(c->end)) || set_steal(d, K_genotypes, PyDict_New()) ||
            PyDict_SetItem(d, K_precise, c->precise ? Py_True : Py_False) || set_steal(d, K_support, PyLong_FromLong(c->support)) ||
            PyDict_SetItem(d, K_rnames, names) || PyDict_SetItem(d, K_qc, c->qc ? Py_True : Py_False) ||
            set_steal(d, K_nm, PyFloat_FromDouble(c->nm)) || PyDict_SetItem(d, K_postprocess, post) || PyDict_SetItem(d, K_svlens, Py_None) ||
            set_steal(d, K_fwd, PyLong_FromLong(c->fwd)) || set_steal(d, K_rev, PyLong_FromLong(c->rev)) || PyDict_SetItem(d, K_fds, fds) ||
            set_steal(d, K_cov_up, PyLong_FromLong(c->cov[0])) || set_steal(d, K_cov_dn, PyLong_FromLong(c->cov[4])) ||
            set_steal(d, K_cov_st, PyLong_FromLong(c->cov[1])) || set_steal(d, K_cov_ce, PyLong_FromLong(c->cov[2])) ||
            set_steal(d, K_cov_en, PyLong_FromLong(c->cov[3])) || PyDict_SetItem(d, K_sample, Py_None) || PyDict_SetItem(d, K_bnd_info, bi) ||
            PyDict_SetItem(d, K_sup_inline, Py_None) || PyDict_SetItem(d, K_sup_splits, Py_None) || PyDict_SetItem(d, K_raw, Py_None) ||
            PyDict_SetItem(d, K_raw_idx, Py_None);
    Py_XDECREF(info); Py_XDECREF(alt); Py_XDECREF(bi); Py_XDECREF(names); Py_XDECREF(fds); Py_XDECREF(post);
    if (bad) { Py_DECREF(d); goto fail; }
    PyObject* obj = new_instance(cls, d);
    if (!obj) goto fail;
    PyList_SET_ITEM(out, i - lo, obj);
  }
  goto done;
fail:
  Py_CLEAR(out);
done:
  PyBuffer_Release(&calls); PyBuffer_Release(&rn);
  return out;
}

/* apply_final(calls: list, records: buffer, lo, alt_pool: buffer, ps_names: list | None, filters: list[str], early_exit: frozenset) */
static PyObject* py_apply_final(PyObject* self, PyObject* args) {
  PyObject *lst, *ps_names, *filters, *early;
  Py_buffer rec, pool;
  long long lo;
  if (!PyArg_ParseTuple(args, "Oy*Ly*OOO", &lst, &rec, &lo, &pool, &ps_names, &filters, &early)) return NULL;
  PyObject* ret = NULL;
  const Py_ssize_t n = PyList_Size(lst);
  if (n < 0 || lo < 0 || (size_t)(lo + n) * sizeof(snf_call_t) > (size_t)rec.len) { PyErr_SetString(PyExc_ValueError, "calls do not match the record table"); goto done; }
  const snf_call_t* C = (const snf_call_t*)rec.buf + lo;
  for (Py_ssize_t i = 0; i < n; i++) {
    const snf_call_t* c = &C[i];
    PyObject* obj = PyList_GET_ITEM(lst, i);
    PyObject** dp = _PyObject_GetDictPtr(obj);
    if (!dp || !*dp || c->filter < 0 || c->filter >= PyList_GET_SIZE(filters)) { PyErr_SetString(PyExc_TypeError, "not a materialised call"); goto done; }
    PyObject* d = *dp;
    PyObject* flt = PyList_GET_ITEM(filters, c->filter);
    PyObject* info = PyDict_GetItemWithError(d, K_info);
    PyObject* gts = PyDict_GetItemWithError(d, K_genotypes);
    if (!info || !gts) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "call without info / genotypes"); goto done; }
    if (PyDict_SetItem(d, K_qc, c->qc ? Py_True : Py_False) || PyDict_SetItem(d, K_filter, flt)) goto done;
    int ee = PySet_Contains(early, flt);
    if (ee < 0) goto done;
    if (!ee && PyDict_SetItem(info, I_COVERAGE_VAR, Py_None)) goto done;     /* see sv.fill_final */
    if (c->ph_set) {
      PyObject* ps = ps_str(c->ph_ps, ps_names);
      if (!ps) goto done;
      PyObject* s = PyUnicode_FromFormat("%d,%S,%d,%d,%s,%s", (int)c->ph_hp, ps, (int)c->ph_hp_support, (int)c->ph_ps_support,
                                         c->ph_hp_pass ? "PASS" : "FAIL", c->ph_ps_pass ? "PASS" : "FAIL");
      Py_DECREF(ps);
      if (set_steal(info, I_PHASE, s)) goto done;
    }
    if (c->gt_set) {
      PyObject* hp = c->gt_hp < 0 ? (Py_INCREF(Py_None), Py_None) : PyUnicode_FromFormat("%d", (int)c->gt_hp);
      PyObject* ps = ps_str(c->gt_ps, ps_names);
      PyObject* t = (hp && ps) ? Py_BuildValue("(iiiii(OO))", (int)c->gt_a, (int)c->gt_b, (int)c->gt_gq, (int)c->gt_dr, (int)c->gt_dv, hp, ps) : NULL;
      Py_XDECREF(hp); Py_XDECREF(ps);
      PyObject* zero = PyLong_FromLong(0);
      int bad = !t || !zero || PyDict_SetItem(gts, zero, t);
      Py_XDECREF(t); Py_XDECREF(zero);
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
  return ret;
}

static PyMethodDef methods[] = {
    {"materialize", py_materialize, METH_VARARGS, "records [lo, hi) -> list of SVCall objects (candidate-stage fields)"},
    {"apply_final", py_apply_final, METH_VARARGS, "finalize-stage fields of the records onto materialised calls"},
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
  INTERN(P_batch, "batch"); INTERN(P_index, "index");
  INTERN(F_n, "n"); INTERN(F_m1, "m1"); INTERN(F_m2, "m2"); INTERN(F_last, "last");
  return PyModule_Create(&moddef);
}
