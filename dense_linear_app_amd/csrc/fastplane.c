/* _fastplane: the per-task loops of the in-process control plane (armonik.py) and of the worker's wave-level
 * execution (worker.py: ExecuteBatch) in C, on the SAME Python objects -- the dicts of results and tasks, the
 * _Result / _Task / DeviceBlob instances, the ProcessStatus values.  Nothing here computes: it is bookkeeping that
 * the reference does in C++ (client_distrib.cpp:459-503, worker_distrib.cpp:99-268) and that cost 5-6 us per task in
 * the interpreter -- more than the GPU needs for a 512 x 512 tile update.  Every function has a pure-Python twin in
 * armonik.py / worker.py (used when this module is missing, and by the tests to compare the two).
 * Host only; no HIP. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static PyObject *s_result_id, *s_name, *s_session_id, *s_data, *s_status, *s_created, *s_completed, *s_pending,
    *s_task_id, *s_payload_id, *s_expected_output_keys, *s_data_dependencies, *s_options, *s_output, *s_attempts,
    *s_partition_id, *s_priority, *s_ptr, *s_nbytes, *s_parent, *s_offset, *s_epoch, *s_view, *s_task;

static int is_str(PyObject *o, PyObject *interned) {
  if (o == interned) return 1;
  return PyUnicode_Check(o) && PyUnicode_Compare(o, interned) == 0;
}

/* obj.attr = value (borrowed value) */
static int set(PyObject *obj, PyObject *attr, PyObject *value) { return PyObject_SetAttr(obj, attr, value); }

static PyObject *new_instance(PyObject *cls) {
  PyTypeObject *tp = (PyTypeObject *)cls;
  return tp->tp_alloc(tp, 0);
}

/* ---- create_results(results, cls, prefix, start, names, session) -> {name: id} --------------------------------
 * ResultsClient.create_results_metadata (C2:373, 471): a fresh write-once result per name. */
static PyObject *create_results(PyObject *self, PyObject *args) {
  PyObject *results, *cls, *prefix, *names, *session;
  long long start;
  if (!PyArg_ParseTuple(args, "O!OULOU", &PyDict_Type, &results, &cls, &prefix, &start, &names, &session)) return NULL;
  PyObject *seq = PySequence_Fast(names, "names must be a sequence");
  if (!seq) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject *out = PyDict_New();
  if (!out) goto fail;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *nm = PySequence_Fast_GET_ITEM(seq, i);
    PyObject *rid = PyUnicode_FromFormat("%U-%08x", prefix, (unsigned int)(start + i));
    if (!rid) goto fail;
    PyObject *r = new_instance(cls);
    if (!r || set(r, s_result_id, rid) || set(r, s_name, nm) || set(r, s_session_id, session) || set(r, s_data, Py_None) ||
        set(r, s_status, s_created) || PyDict_SetItem(results, rid, r) || PyDict_SetItem(out, nm, rid)) {
      Py_XDECREF(r);
      Py_DECREF(rid);
      goto fail;
    }
    Py_DECREF(r);
    Py_DECREF(rid);
  }
  Py_DECREF(seq);
  return out;
fail:
  Py_XDECREF(out);
  Py_DECREF(seq);
  return NULL;
}

/* ---- complete_many(results, ids, datas, host_prefix) -> bool: _complete_result for every (id, bytes) pair ------
 * host_prefix (str or None): every result's name must start with it, else nothing is touched and False comes back
 * (the caller's general path decides what goes to HBM: ResultsClient.upload_results_data). */
static PyObject *complete_many(PyObject *self, PyObject *args) {
  PyObject *results, *ids, *datas, *prefix;
  if (!PyArg_ParseTuple(args, "O!O!O!O", &PyDict_Type, &results, &PyList_Type, &ids, &PyList_Type, &datas, &prefix)) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(ids);
  if (PyList_GET_SIZE(datas) != n) {
    PyErr_SetString(PyExc_ValueError, "complete_many: ids and datas differ in length");
    return NULL;
  }
  for (int pass = 0; pass < 2; ++pass)
    for (Py_ssize_t i = 0; i < n; ++i) {
      PyObject *rid = PyList_GET_ITEM(ids, i), *r = PyDict_GetItemWithError(results, rid);
      if (!r) {
        if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "unknown result id %S", rid);
        return NULL;
      }
      if (pass == 0) {
        if (!PyBytes_Check(PyList_GET_ITEM(datas, i))) Py_RETURN_FALSE;
        if (prefix == Py_None) continue;
        PyObject *nm = PyObject_GetAttr(r, s_name);
        if (!nm) return NULL;
        const int ok = PyUnicode_Check(nm) && PyUnicode_Check(prefix) && PyUnicode_Tailmatch(nm, prefix, 0, PY_SSIZE_T_MAX, -1) == 1 &&
                       PyBytes_Check(PyList_GET_ITEM(datas, i));
        Py_DECREF(nm);
        if (!ok) Py_RETURN_FALSE;
        continue;
      }
      PyObject *st = PyObject_GetAttr(r, s_status);
      if (!st) return NULL;
      const int done = is_str(st, s_completed);
      Py_DECREF(st);
      if (done) {
        PyErr_Format(PyExc_RuntimeError, "result %S is write-once and already has data", rid);
        return NULL;
      }
      if (set(r, s_data, PyList_GET_ITEM(datas, i)) || set(r, s_status, s_completed)) return NULL;
    }
  Py_RETURN_TRUE;
}

/* ---- payload_join(tasks, results) -> (buf, offsets) or None --------------------------------------------------
 * the payloads of the tasks back to back and their n + 1 offsets (int64, as bytes): what chol_parse_payloads takes.
 * None when a payload is not host bytes (the caller's general path fetches them one by one). */
static PyObject *payload_join(PyObject *self, PyObject *args) {
  PyObject *tasks, *results;
  if (!PyArg_ParseTuple(args, "O!O!", &PyList_Type, &tasks, &PyDict_Type, &results)) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(tasks);
  PyObject *offs = PyBytes_FromStringAndSize(NULL, (n + 1) * 8);
  if (!offs) return NULL;
  int64_t *off = (int64_t *)PyBytes_AS_STRING(offs);
  Py_ssize_t total = 0;
  for (int pass = 0; pass < 2; ++pass) {
    PyObject *buf = pass ? PyBytes_FromStringAndSize(NULL, total) : NULL;
    if (pass && !buf) {
      Py_DECREF(offs);
      return NULL;
    }
    Py_ssize_t pos = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
      PyObject *pid = PyObject_GetAttr(PyList_GET_ITEM(tasks, i), s_payload_id);
      PyObject *r = pid ? PyDict_GetItemWithError(results, pid) : NULL;
      Py_XDECREF(pid);
      PyObject *data = r ? PyObject_GetAttr(r, s_data) : NULL;
      if (!data || !PyBytes_Check(data)) {
        Py_XDECREF(data);
        Py_XDECREF(buf);
        Py_DECREF(offs);
        if (PyErr_Occurred()) return NULL;
        Py_RETURN_NONE;
      }
      const Py_ssize_t len = PyBytes_GET_SIZE(data);
      if (pass) memcpy(PyBytes_AS_STRING(buf) + pos, PyBytes_AS_STRING(data), (size_t)len);
      off[i] = pos;
      pos += len;
      Py_DECREF(data);
    }
    off[n] = pos;
    total = pos;
    if (pass) return Py_BuildValue("(NN)", buf, offs);
  }
  return NULL; /* not reached */
}

/* ---- submit(tasks, pending, results, tcs, task_cls, prefix, start, session, options) -> [task ids] ------------
 * TasksClient.submit_tasks (C2:498): every id a task names must exist; one _Task per TaskCreation. */
static PyObject *submit(PyObject *self, PyObject *args) {
  PyObject *tasks, *pending, *results, *tcs, *cls, *prefix, *session, *options;
  long long start;
  if (!PyArg_ParseTuple(args, "O!O!O!OOULUO", &PyDict_Type, &tasks, &PyList_Type, &pending, &PyDict_Type, &results, &tcs, &cls, &prefix,
                        &start, &session, &options))
    return NULL;
  PyObject *seq = PySequence_Fast(tcs, "task_creations must be a sequence");
  if (!seq) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject *ids = PyList_New(n), *zero = PyLong_FromLong(0);
  if (!ids || !zero) goto fail;
  /* validate everything first: a bad id must not leave half a submission behind */
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *tc = PySequence_Fast_GET_ITEM(seq, i);
    PyObject *pid = PyObject_GetAttr(tc, s_payload_id), *keys = PyObject_GetAttr(tc, s_expected_output_keys),
             *deps = PyObject_GetAttr(tc, s_data_dependencies);
    int bad = !pid || !keys || !deps || !PyList_Check(keys) || !PyList_Check(deps);
    if (!bad && PyDict_Contains(results, pid) != 1) {
      if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "submit_tasks: unknown result id %S", pid);
      bad = 1;
    }
    for (int which = 0; !bad && which < 2; ++which) {
      PyObject *lst = which ? deps : keys;
      for (Py_ssize_t q = 0; q < PyList_GET_SIZE(lst); ++q)
        if (PyDict_Contains(results, PyList_GET_ITEM(lst, q)) != 1) {
          if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "submit_tasks: unknown result id %S", PyList_GET_ITEM(lst, q));
          bad = 1;
          break;
        }
    }
    if (bad && !PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "submit_tasks: malformed TaskCreation");
    Py_XDECREF(pid);
    Py_XDECREF(keys);
    Py_XDECREF(deps);
    if (bad) goto fail;
  }
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *tc = PySequence_Fast_GET_ITEM(seq, i);
    PyObject *pid = PyObject_GetAttr(tc, s_payload_id), *keys = PyObject_GetAttr(tc, s_expected_output_keys),
             *deps = PyObject_GetAttr(tc, s_data_dependencies);
    PyObject *k2 = keys ? PyList_GetSlice(keys, 0, PyList_GET_SIZE(keys)) : NULL, *d2 = deps ? PyList_GetSlice(deps, 0, PyList_GET_SIZE(deps)) : NULL;
    PyObject *tid = PyUnicode_FromFormat("%U-%08x", prefix, (unsigned int)(start + i));
    PyObject *t = new_instance(cls);
    int bad = !pid || !k2 || !d2 || !tid || !t || set(t, s_task_id, tid) || set(t, s_session_id, session) || set(t, s_payload_id, pid) ||
              set(t, s_expected_output_keys, k2) || set(t, s_data_dependencies, d2) || set(t, s_options, options) ||
              set(t, s_status, s_pending) || set(t, s_output, Py_None) || set(t, s_attempts, zero) || PyDict_SetItem(tasks, tid, t) ||
              PyList_Append(pending, tid);
    if (!bad) {
      Py_INCREF(tid);
      PyList_SET_ITEM(ids, i, tid);
    }
    Py_XDECREF(pid);
    Py_XDECREF(keys);
    Py_XDECREF(deps);
    Py_XDECREF(k2);
    Py_XDECREF(d2);
    Py_XDECREF(tid);
    Py_XDECREF(t);
    if (bad) goto fail;
  }
  Py_DECREF(zero);
  Py_DECREF(seq);
  return ids;
fail:
  Py_XDECREF(zero);
  Py_XDECREF(ids);
  Py_DECREF(seq);
  return NULL;
}

/* ---- task_creations(cls, payload_ids, output_ids, items) -> [TaskCreation] -------------------------------------
 * The client's per-task TaskCreation{payload_id, expected_output_keys = [output], data_dependencies = sorted unique
 * ids} (C2:480-492) for a whole submission; items[i] = (payload text, [dependency ids]). */
static PyObject *task_creations(PyObject *self, PyObject *args) {
  PyObject *cls, *pids, *outs, *items;
  if (!PyArg_ParseTuple(args, "OO!O!O!", &cls, &PyList_Type, &pids, &PyList_Type, &outs, &PyList_Type, &items)) return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(items);
  if (PyList_GET_SIZE(pids) != n || PyList_GET_SIZE(outs) != n) {
    PyErr_SetString(PyExc_ValueError, "task_creations: lists of different lengths");
    return NULL;
  }
  PyObject *res = PyList_New(n);
  if (!res) return NULL;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *it = PyList_GET_ITEM(items, i);
    PyObject *src = (PyTuple_Check(it) && PyTuple_GET_SIZE(it) >= 2) ? PyTuple_GET_ITEM(it, 1) : NULL;
    if (!src || !PyList_Check(src)) {
      PyErr_SetString(PyExc_TypeError, "task_creations: items must be (payload, [dependency ids])");
      goto fail;
    }
    PyObject *deps = PyList_GetSlice(src, 0, PyList_GET_SIZE(src));
    if (!deps || PyList_Sort(deps)) {
      Py_XDECREF(deps);
      goto fail;
    }
    for (Py_ssize_t q = PyList_GET_SIZE(deps) - 1; q > 0; --q) {  /* unique (sorted: equal ids are neighbours) */
      const int eq = PyObject_RichCompareBool(PyList_GET_ITEM(deps, q), PyList_GET_ITEM(deps, q - 1), Py_EQ);
      if (eq < 0 || (eq && PyList_SetSlice(deps, q, q + 1, NULL))) {
        Py_DECREF(deps);
        goto fail;
      }
    }
    PyObject *keys = PyList_New(1), *tc = new_instance(cls);
    int bad = !keys || !tc;
    if (!bad) {
      Py_INCREF(PyList_GET_ITEM(outs, i));
      PyList_SET_ITEM(keys, 0, PyList_GET_ITEM(outs, i));
      bad = set(tc, s_payload_id, PyList_GET_ITEM(pids, i)) || set(tc, s_expected_output_keys, keys) || set(tc, s_data_dependencies, deps);
    }
    Py_XDECREF(keys);
    Py_DECREF(deps);
    if (bad) {
      Py_XDECREF(tc);
      goto fail;
    }
    PyList_SET_ITEM(res, i, tc);
  }
  return res;
fail:
  Py_DECREF(res);
  return NULL;
}

/* 1 ready, 0 not, -1 error: the payload and every data dependency of the task are completed (ControlPlane._ready) */
static int task_ready(PyObject *t, PyObject *results) {
  PyObject *pid = PyObject_GetAttr(t, s_payload_id), *deps = PyObject_GetAttr(t, s_data_dependencies);
  int ok = pid && deps && PyList_Check(deps) ? 1 : -1;
  for (Py_ssize_t q = -1; ok == 1 && q < PyList_GET_SIZE(deps); ++q) {
    PyObject *rid = q < 0 ? pid : PyList_GET_ITEM(deps, q), *r = PyDict_GetItemWithError(results, rid);
    if (!r) {
      if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "unknown result id %S", rid);
      ok = -1;
      break;
    }
    PyObject *st = PyObject_GetAttr(r, s_status);
    if (!st) {
      ok = -1;
      break;
    }
    if (!is_str(st, s_completed)) ok = 0;
    Py_DECREF(st);
  }
  Py_XDECREF(pid);
  Py_XDECREF(deps);
  return ok;
}

/* ---- split_ready(pending, tasks, results, batch_partitions) -> ({partition: [tasks]}, rest) -------------------
 * one pass of ControlPlane._pump over the pending list: the ready tasks of partitions with a batch-capable worker */
static PyObject *split_ready(PyObject *self, PyObject *args) {
  PyObject *pending, *tasks, *results, *parts;
  if (!PyArg_ParseTuple(args, "O!O!O!O", &PyList_Type, &pending, &PyDict_Type, &tasks, &PyDict_Type, &results, &parts)) return NULL;
  PyObject *by = PyDict_New(), *rest = PyList_New(0);
  if (!by || !rest) goto fail;
  for (Py_ssize_t i = 0; i < PyList_GET_SIZE(pending); ++i) {
    PyObject *tid = PyList_GET_ITEM(pending, i), *t = PyDict_GetItemWithError(tasks, tid);
    if (!t) {
      if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "unknown task id %S", tid);
      goto fail;
    }
    int go = task_ready(t, results);
    if (go < 0) goto fail;
    if (go) {
      PyObject *opt = PyObject_GetAttr(t, s_options), *part = opt ? PyObject_GetAttr(opt, s_partition_id) : NULL;
      Py_XDECREF(opt);
      if (!part) goto fail;
      const int has = PySequence_Contains(parts, part);
      if (has < 0) {
        Py_DECREF(part);
        goto fail;
      }
      if (has) {
        PyObject *lst = PyDict_GetItemWithError(by, part);
        if (!lst) {
          if (PyErr_Occurred() || !(lst = PyList_New(0)) || PyDict_SetItem(by, part, lst)) {
            Py_XDECREF(lst);
            Py_DECREF(part);
            goto fail;
          }
          Py_DECREF(lst);
        }
        if (PyList_Append(lst, t)) {
          Py_DECREF(part);
          goto fail;
        }
      } else {
        go = 0;
      }
      Py_DECREF(part);
    }
    if (!go && PyList_Append(rest, tid)) goto fail;
  }
  return Py_BuildValue("(NN)", by, rest);
fail:
  Py_XDECREF(by);
  Py_XDECREF(rest);
  return NULL;
}

/* zlib's crc32 and adler32 (worker._tag_of: the 64-bit name of a write-once result's content) */
static uint32_t crc_tab[256];
static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    crc_tab[i] = c;
  }
}
static uint64_t tag_of(const char *p, Py_ssize_t n) {
  uint32_t c = 0xFFFFFFFFu, a = 1, b = 0;
  for (Py_ssize_t i = 0; i < n; ++i) {
    c = crc_tab[(c ^ (unsigned char)p[i]) & 0xFF] ^ (c >> 8);
    a = (a + (unsigned char)p[i]) % 65521u;
    b = (b + a) % 65521u;
  }
  const uint64_t v = ((uint64_t)(c ^ 0xFFFFFFFFu) << 32) | (uint64_t)((b << 16) | a);
  return v ? v : 1;
}

/* ---- resolve_batch(tasks, results, buf, ops, Bs, id_off, id_len, blob_cls, async_potrf) -> (groups, fallback) --
 * worker.ExecuteBatch's first loop.  ops / Bs (int32), id_off (int64), id_len (int32): chol_parse_payloads' outputs
 * (buffer protocol).  A task goes into a group when its op is one of the four, B is a positive multiple of 128 and
 * every tile id its payload names is a declared data dependency whose blob is HBM-resident (blob_cls) with B*B*8
 * bytes; everything else is returned in `fallback` (indices) for the one-task path, whose messages are the reference's.
 * groups: {(launch class, B, urgent): ([task index], [ptr0], [ptr1], [ptr2], [tag])}, launch class 1 TRSM, 2 SYRK and
 * GEMM together, 4 POTRF; urgent = TaskOptions.priority > 1. */
/* one resolved task of a batch: its group, operand addresses, content tag */
typedef struct {
  int group;
  Py_ssize_t idx;
  unsigned long long ptr[3], tag;
} resolved_t;
static int by_l_then_index(const void *pa, const void *pb) {  /* TRSM: tasks that share an L side by side (ascending L, stable) */
  const resolved_t *a = (const resolved_t *)pa, *b = (const resolved_t *)pb;
  if (a->ptr[1] != b->ptr[1]) return a->ptr[1] < b->ptr[1] ? -1 : 1;
  return a->idx < b->idx ? -1 : (a->idx > b->idx);
}
static int syrk_last_then_index(const void *pa, const void *pb) {  /* updates: the SYRK tasks (no second operand) last, stable */
  const resolved_t *a = (const resolved_t *)pa, *b = (const resolved_t *)pb;
  const int sa = a->ptr[2] == 0, sb = b->ptr[2] == 0;
  if (sa != sb) return sa - sb;
  return a->idx < b->idx ? -1 : (a->idx > b->idx);
}

static PyObject *resolve_batch(PyObject *self, PyObject *args) {
  PyObject *tasks, *results, *blob_cls;
  Py_buffer buf, ops, Bs, ioff, ilen;
  int async_potrf;
  if (!PyArg_ParseTuple(args, "O!O!y*y*y*y*y*Op", &PyList_Type, &tasks, &PyDict_Type, &results, &buf, &ops, &Bs, &ioff, &ilen, &blob_cls,
                        &async_potrf))
    return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(tasks);
  PyObject *groups = PyDict_New(), *fallback = PyList_New(0), *ret = NULL;
  const int32_t *op = (const int32_t *)ops.buf, *Bv = (const int32_t *)Bs.buf, *il = (const int32_t *)ilen.buf;
  const int64_t *io = (const int64_t *)ioff.buf;
  enum { MAXG = 32 };
  int gcode[MAXG], gB[MAXG], gurgent[MAXG], gcount[MAXG], ng = 0;
  resolved_t *rs = (resolved_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(resolved_t));
  Py_ssize_t nr = 0;
  if (!groups || !fallback || !rs) {
    if (!rs) PyErr_NoMemory();
    goto done;
  }
  if (ops.len < (Py_ssize_t)(n * 4) || Bs.len < (Py_ssize_t)(n * 4) || ioff.len < (Py_ssize_t)(n * 24) || ilen.len < (Py_ssize_t)(n * 12)) {
    PyErr_SetString(PyExc_ValueError, "resolve_batch: parse arrays shorter than the task list");
    goto done;
  }
  for (Py_ssize_t i = 0; i < n; ++i) {
    const int code = op[i], B = Bv[i];
    int ok = code >= 1 && code <= 4 && B > 0 && B % 128 == 0 && (code != 4 || async_potrf);
    unsigned long long ptr[3] = {0, 0, 0}, tag = 0;
    PyObject *t = PyList_GET_ITEM(tasks, i);
    PyObject *deps = ok ? PyObject_GetAttr(t, s_data_dependencies) : NULL;
    if (ok && !deps) goto done;
    const int nid = code == 4 ? 1 : code == 3 ? 3 : 2;
    const long long want = (long long)B * B * 8;
    for (int r = 0; ok && r < nid; ++r) {
      const int64_t o = io[3 * i + r];
      const int len = il[3 * i + r];
      if (len < 0 || o < 0 || o + len > buf.len) {
        ok = 0;
        break;
      }
      PyObject *rid = PyUnicode_DecodeUTF8((const char *)buf.buf + o, len, NULL);
      if (!rid) {
        PyErr_Clear();
        ok = 0;
        break;
      }
      int has = PySequence_Contains(deps, rid);
      PyObject *res = has == 1 ? PyDict_GetItemWithError(results, rid) : NULL;
      if (has < 0 || (!res && PyErr_Occurred())) PyErr_Clear();
      PyObject *data = res ? PyObject_GetAttr(res, s_data) : NULL;
      if (res && !data) PyErr_Clear();
      if (data && (PyObject *)Py_TYPE(data) == blob_cls) {
        PyObject *nb = PyObject_GetAttr(data, s_nbytes), *pp = PyObject_GetAttr(data, s_ptr);
        if (nb && pp && PyLong_AsLongLong(nb) == want) ptr[r] = PyLong_AsUnsignedLongLong(pp);
        else ok = 0;
        if (PyErr_Occurred()) PyErr_Clear(), ok = 0;
        Py_XDECREF(nb);
        Py_XDECREF(pp);
      } else {
        ok = 0;
      }
      Py_XDECREF(data);
      if (ok && code == 1 && r == 1) tag = tag_of((const char *)buf.buf + o, len);
      Py_DECREF(rid);
    }
    Py_XDECREF(deps);
    int g = -1;
    if (ok) {
      PyObject *opt = PyObject_GetAttr(t, s_options), *pr = opt ? PyObject_GetAttr(opt, s_priority) : NULL;
      Py_XDECREF(opt);
      if (!pr) goto done;
      const long prio = PyLong_AsLong(pr);
      Py_DECREF(pr);
      if (prio == -1 && PyErr_Occurred()) goto done;
      const int gc = code == 3 ? 2 : code, urgent = prio > 1;
      for (g = 0; g < ng; ++g)
        if (gcode[g] == gc && gB[g] == B && gurgent[g] == urgent) break;
      if (g == ng) {
        if (ng == MAXG) ok = 0, g = -1;  /* (more launch classes in one call than anybody submits: the one-task path takes the rest) */
        else gcode[ng] = gc, gB[ng] = B, gurgent[ng] = urgent, gcount[ng] = 0, ++ng;
      }
    }
    if (!ok) {
      PyObject *ix = PyLong_FromSsize_t(i);
      if (!ix || PyList_Append(fallback, ix)) {
        Py_XDECREF(ix);
        goto done;
      }
      Py_DECREF(ix);
      continue;
    }
    resolved_t *r = rs + nr++;
    r->group = g, r->idx = i, r->ptr[0] = ptr[0], r->ptr[1] = ptr[1], r->ptr[2] = ptr[2], r->tag = tag;
    ++gcount[g];
  }
  /* per launch class: the task indices in launch order and the 5 x m table of 64-bit words (c_in, a, b, c_out -- filled
   * in by the caller once the outputs are allocated --, tag) that chol_tile_batch / chol_potrf_batch take, as a bytearray */
  for (int g = 0; g < ng; ++g) {
    const Py_ssize_t m = gcount[g];
    resolved_t *sel = (resolved_t *)malloc((size_t)m * sizeof(resolved_t));
    if (!sel) {
      PyErr_NoMemory();
      goto done;
    }
    Py_ssize_t q = 0, nsyrk = 0;
    for (Py_ssize_t i = 0; i < nr; ++i)
      if (rs[i].group == g) sel[q++] = rs[i];
    if (gcode[g] == 1) qsort(sel, (size_t)m, sizeof(resolved_t), by_l_then_index);
    else if (gcode[g] == 2) qsort(sel, (size_t)m, sizeof(resolved_t), syrk_last_then_index);
    PyObject *idxs = PyList_New(m), *raw = PyByteArray_FromStringAndSize(NULL, (Py_ssize_t)(5 * m * 8));
    int bad = !idxs || !raw;
    if (!bad) {
      unsigned long long *w = (unsigned long long *)PyByteArray_AS_STRING(raw);
      for (q = 0; q < m && !bad; ++q) {
        w[q] = sel[q].ptr[0], w[m + q] = sel[q].ptr[1], w[2 * m + q] = sel[q].ptr[2], w[3 * m + q] = 0, w[4 * m + q] = sel[q].tag;
        if (gcode[g] == 2 && sel[q].ptr[2] == 0) ++nsyrk;
        PyObject *ix = PyLong_FromSsize_t(sel[q].idx);
        if (!ix) bad = 1;
        else PyList_SET_ITEM(idxs, q, ix);
      }
    }
    free(sel);
    PyObject *key = bad ? NULL : Py_BuildValue("(iiO)", gcode[g], gB[g], gurgent[g] ? Py_True : Py_False);
    PyObject *val = key ? Py_BuildValue("(OOn)", idxs, raw, nsyrk) : NULL;
    if (!val || PyDict_SetItem(groups, key, val)) bad = 1;
    Py_XDECREF(key);
    Py_XDECREF(val);
    Py_XDECREF(idxs);
    Py_XDECREF(raw);
    if (bad) goto done;
  }
  ret = Py_BuildValue("(OO)", groups, fallback);
done:
  free(rs);
  Py_XDECREF(groups);
  Py_XDECREF(fallback);
  PyBuffer_Release(&buf);
  PyBuffer_Release(&ops);
  PyBuffer_Release(&Bs);
  PyBuffer_Release(&ioff);
  PyBuffer_Release(&ilen);
  return ret;
}

/* ---- complete_batch(tasks, idxs, results, blob_cls, parent, base, tb, epoch, out, ok_status) -> [output ids] ----
 * the send_result of every task of a grouped launch (W2:261): task idxs[q]'s first expected result becomes the
 * device blob (parent, offset q tb); out[idxs[q]] = ok_status.  A result that is unknown or already completed raises
 * (write-once), as TaskHandler.send_result(...).get() does. */
static PyObject *complete_batch(PyObject *self, PyObject *args) {
  PyObject *tasks, *idxs, *results, *blob_cls, *parent, *out, *ok_status;
  unsigned long long base, tb;
  long long epoch;
  if (!PyArg_ParseTuple(args, "O!O!O!OOKKLO!O", &PyList_Type, &tasks, &PyList_Type, &idxs, &PyDict_Type, &results, &blob_cls, &parent, &base,
                        &tb, &epoch, &PyList_Type, &out, &ok_status))
    return NULL;
  const Py_ssize_t m = PyList_GET_SIZE(idxs);
  PyObject *oids = PyList_New(m), *ep = PyLong_FromLongLong(epoch), *nb = PyLong_FromUnsignedLongLong(tb);
  if (!oids || !ep || !nb) goto fail;
  for (Py_ssize_t q = 0; q < m; ++q) {
    const Py_ssize_t i = PyLong_AsSsize_t(PyList_GET_ITEM(idxs, q));
    if (i < 0 || i >= PyList_GET_SIZE(tasks) || i >= PyList_GET_SIZE(out)) {
      if (!PyErr_Occurred()) PyErr_SetString(PyExc_IndexError, "complete_batch: task index out of range");
      goto fail;
    }
    PyObject *t = PyList_GET_ITEM(tasks, i), *keys = PyObject_GetAttr(t, s_expected_output_keys);
    if (!keys || !PyList_Check(keys) || PyList_GET_SIZE(keys) < 1) {
      if (keys && !PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "task without an expected result");
      Py_XDECREF(keys);
      goto fail;
    }
    PyObject *oid = PyList_GET_ITEM(keys, 0);
    Py_INCREF(oid);
    Py_DECREF(keys);
    PyList_SET_ITEM(oids, q, oid); /* (steals) */
    PyObject *r = PyDict_GetItemWithError(results, oid);
    if (!r) {
      if (!PyErr_Occurred()) PyErr_Format(PyExc_KeyError, "unknown result id %S", oid);
      goto fail;
    }
    PyObject *st = PyObject_GetAttr(r, s_status);
    if (!st) goto fail;
    const int was = is_str(st, s_completed);
    Py_DECREF(st);
    if (was) {
      PyErr_Format(PyExc_RuntimeError, "result %S is write-once and already has data", oid);
      goto fail;
    }
    PyObject *blob = new_instance(blob_cls), *off = PyLong_FromUnsignedLongLong((unsigned long long)q * tb),
             *pp = PyLong_FromUnsignedLongLong(base + (unsigned long long)q * tb);
    const int bad = !blob || !off || !pp || set(blob, s_parent, parent) || set(blob, s_offset, off) || set(blob, s_nbytes, nb) ||
                    set(blob, s_ptr, pp) || set(blob, s_epoch, ep) || set(blob, s_view, Py_None) || set(r, s_data, blob) ||
                    set(r, s_status, s_completed);
    Py_XDECREF(blob);
    Py_XDECREF(off);
    Py_XDECREF(pp);
    if (bad) goto fail;
    Py_INCREF(ok_status);
    if (PyList_SetItem(out, i, ok_status)) goto fail;
  }
  Py_DECREF(ep);
  Py_DECREF(nb);
  return oids;
fail:
  /* (PyList_New leaves NULL items: safe to release) */
  Py_XDECREF(oids);
  Py_XDECREF(ep);
  Py_XDECREF(nb);
  return NULL;
}

/* ---- book_batch(tasks, statuses, results, executed, ok_status) -> [indices for the slow path] ------------------
 * ControlPlane._run_batch's bookkeeping for the common case: the task answered Ok and its one result is there. */
static PyObject *book_batch(PyObject *self, PyObject *args) {
  PyObject *tasks, *statuses, *results, *executed, *ok_status;
  if (!PyArg_ParseTuple(args, "O!O!O!O!O", &PyList_Type, &tasks, &PyList_Type, &statuses, &PyDict_Type, &results, &PyList_Type, &executed,
                        &ok_status))
    return NULL;
  const Py_ssize_t n = PyList_GET_SIZE(tasks);
  if (PyList_GET_SIZE(statuses) != n) {
    PyErr_SetString(PyExc_RuntimeError, "ExecuteBatch returned a status list of the wrong length");
    return NULL;
  }
  PyObject *slow = PyList_New(0);
  if (!slow) return NULL;
  for (Py_ssize_t i = 0; i < n; ++i) {
    PyObject *t = PyList_GET_ITEM(tasks, i), *status = PyList_GET_ITEM(statuses, i);
    int fast = status == ok_status;
    PyObject *keys = fast ? PyObject_GetAttr(t, s_expected_output_keys) : NULL;
    if (fast && !keys) goto fail;
    if (fast) {
      fast = PyList_Check(keys) && PyList_GET_SIZE(keys) == 1;
      if (fast) {
        PyObject *r = PyDict_GetItemWithError(results, PyList_GET_ITEM(keys, 0));
        PyObject *st = r ? PyObject_GetAttr(r, s_status) : NULL;
        if (PyErr_Occurred()) {
          Py_XDECREF(st);
          Py_DECREF(keys);
          goto fail;
        }
        fast = st && is_str(st, s_completed);
        Py_XDECREF(st);
      }
      Py_DECREF(keys);
    }
    if (!fast) {
      PyObject *ix = PyLong_FromSsize_t(i);
      if (!ix || PyList_Append(slow, ix)) {
        Py_XDECREF(ix);
        goto fail;
      }
      Py_DECREF(ix);
      continue;
    }
    PyObject *att = PyObject_GetAttr(t, s_attempts);
    if (!att) goto fail;
    PyObject *att1 = PyLong_FromLong(PyLong_AsLong(att) + 1);
    Py_DECREF(att);
    PyObject *tid = PyObject_GetAttr(t, s_task_id);
    const int bad = !att1 || !tid || set(t, s_attempts, att1) || set(t, s_output, status) || set(t, s_status, s_completed) ||
                    PyList_Append(executed, tid);
    Py_XDECREF(att1);
    Py_XDECREF(tid);
    if (bad) goto fail;
  }
  return slow;
fail:
  Py_DECREF(slow);
  return NULL;
}

/* ---- tasks_of(handlers) -> [handler._task] */
static PyObject *tasks_of(PyObject *self, PyObject *handlers) {
  PyObject *seq = PySequence_Fast(handlers, "handlers must be a sequence");
  if (!seq) return NULL;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject *out = PyList_New(n);
  for (Py_ssize_t i = 0; out && i < n; ++i) {
    PyObject *t = PyObject_GetAttr(PySequence_Fast_GET_ITEM(seq, i), s_task);
    if (!t) {
      Py_CLEAR(out);
      break;
    }
    PyList_SET_ITEM(out, i, t);
  }
  Py_DECREF(seq);
  return out;
}

static PyMethodDef methods[] = {
    {"create_results", create_results, METH_VARARGS, "create_results(results, cls, prefix, start, names, session) -> {name: id}"},
    {"complete_many", complete_many, METH_VARARGS, "complete_many(results, ids, datas, host_prefix) -> bool"},
    {"payload_join", payload_join, METH_VARARGS, "payload_join(tasks, results) -> (buf, offsets) or None"},
    {"submit", submit, METH_VARARGS, "submit(tasks, pending, results, tcs, task_cls, prefix, start, session, options) -> [task ids]"},
    {"split_ready", split_ready, METH_VARARGS, "split_ready(pending, tasks, results, batch_partitions) -> ({partition: [tasks]}, rest)"},
    {"resolve_batch", resolve_batch, METH_VARARGS, "resolve_batch(tasks, results, buf, ops, Bs, id_off, id_len, blob_cls, async_potrf) -> ({(code, B, urgent): (task indices in launch order, bytearray of the 5 x m operand table, number of SYRK tasks)}, fallback)"},
    {"complete_batch", complete_batch, METH_VARARGS, "complete_batch(tasks, idxs, results, blob_cls, parent, base, tb, epoch, out, ok_status) -> [output ids]"},
    {"book_batch", book_batch, METH_VARARGS, "book_batch(tasks, statuses, results, executed, ok_status) -> [slow indices]"},
    {"task_creations", task_creations, METH_VARARGS, "task_creations(cls, payload_ids, output_ids, items) -> [TaskCreation]"},
    {"tasks_of", tasks_of, METH_O, "tasks_of(handlers) -> [handler._task]"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastplane", "per-task loops of the in-process control plane and worker, in C", -1, methods};

PyMODINIT_FUNC PyInit__fastplane(void) {
  crc_init();
#define S(var, text) \
  if (!(var = PyUnicode_InternFromString(text))) return NULL
  S(s_result_id, "result_id");
  S(s_name, "name");
  S(s_session_id, "session_id");
  S(s_data, "data");
  S(s_status, "status");
  S(s_created, "created");
  S(s_completed, "completed");
  S(s_pending, "pending");
  S(s_task_id, "task_id");
  S(s_payload_id, "payload_id");
  S(s_expected_output_keys, "expected_output_keys");
  S(s_data_dependencies, "data_dependencies");
  S(s_options, "options");
  S(s_output, "output");
  S(s_attempts, "attempts");
  S(s_partition_id, "partition_id");
  S(s_priority, "priority");
  S(s_ptr, "ptr");
  S(s_nbytes, "nbytes");
  S(s_parent, "_parent");
  S(s_offset, "_offset");
  S(s_epoch, "epoch");
  S(s_view, "_view");
  S(s_task, "_task");
#undef S
  return PyModule_Create(&moddef);
}
