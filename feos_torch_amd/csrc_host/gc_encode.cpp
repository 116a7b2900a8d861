// Native front-end for the gc-PC-SAFT molecule encoding: walks the reference's
// constructor arguments `segments` / `bonds` (feos_torch/gc_pcsaft.py:13-63: per row and component a
// list of segment identifiers and a list of [i, j] bonds between list positions) and writes the
// 80-byte device rows of include/pcsaft_hip.h.  Host code only (CPython C API, no GPU, no torch).
//
//   _gc_encode.encode_rows(segment_identifier, segment_lists, bond_lists, out)
//       out: writable C-contiguous buffer of n*80 bytes (numpy uint8 [n, 80])
//
// Semantics are those of the pure-Python encoder in feos_torch_amd/gc_pcsaft.py (kept as the
// specification and checked against this module in tests/test_abi.py): per molecule the distinct
// segment types sorted by index with multiplicities, and the distinct bond types (larger segment
// index first, feos_torch/gc_pcsaft.py:35) sorted, with multiplicities; at most 8 of each.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

constexpr int MAXE = 8;

struct Enc {
    uint8_t b[40];  // seg_id[8], seg_cnt[8], bond_a[8], bond_b[8], bond_cnt[8]
};

struct PairHash {
    size_t operator()(const std::pair<PyObject*, PyObject*>& p) const {
        return std::hash<const void*>()(p.first) * 1000003u ^ std::hash<const void*>()(p.second);
    }
};

struct Encoder {
    std::unordered_map<std::string, int> by_name;
    std::unordered_map<PyObject*, int> by_obj;  // identifier objects already resolved (strings are usually shared)
    std::unordered_map<std::pair<PyObject*, PyObject*>, Enc, PairHash> by_lists;

    int segment_index(PyObject* s) {
        auto it = by_obj.find(s);
        if (it != by_obj.end()) return it->second;
        if (!PyUnicode_Check(s)) {
            PyErr_SetString(PyExc_TypeError, "segment identifiers must be str");
            return -1;
        }
        Py_ssize_t len = 0;
        const char* u = PyUnicode_AsUTF8AndSize(s, &len);
        if (!u) return -1;
        auto jt = by_name.find(std::string(u, (size_t)len));
        if (jt == by_name.end()) {
            PyErr_Format(PyExc_KeyError, "%U", s);
            return -1;
        }
        by_obj.emplace(s, jt->second);
        return jt->second;
    }

    // encodes one molecule; returns false with a Python error set
    bool molecule(PyObject* segs, PyObject* bonds, Enc& out) {
        PyObject* fs = PySequence_Fast(segs, "segments of a molecule must be a sequence");
        if (!fs) return false;
        const Py_ssize_t ns = PySequence_Fast_GET_SIZE(fs);
        std::vector<int> ids((size_t)ns);
        std::map<int, int> scount;
        for (Py_ssize_t k = 0; k < ns; k++) {
            int id = segment_index(PySequence_Fast_GET_ITEM(fs, k));
            if (id < 0) { Py_DECREF(fs); return false; }
            ids[(size_t)k] = id;
            scount[id]++;
        }
        Py_DECREF(fs);
        PyObject* fb = PySequence_Fast(bonds, "bonds of a molecule must be a sequence");
        if (!fb) return false;
        const Py_ssize_t nb = PySequence_Fast_GET_SIZE(fb);
        std::map<std::pair<int, int>, int> bcount;
        for (Py_ssize_t k = 0; k < nb; k++) {
            PyObject* pr = PySequence_Fast(PySequence_Fast_GET_ITEM(fb, k), "a bond must be a pair of indices");
            if (!pr) { Py_DECREF(fb); return false; }
            if (PySequence_Fast_GET_SIZE(pr) != 2) {
                Py_DECREF(pr); Py_DECREF(fb);
                PyErr_SetString(PyExc_ValueError, "a bond must be a pair of indices");
                return false;
            }
            long i = PyLong_AsLong(PySequence_Fast_GET_ITEM(pr, 0));
            long j = PyLong_AsLong(PySequence_Fast_GET_ITEM(pr, 1));
            Py_DECREF(pr);
            if ((i == -1 || j == -1) && PyErr_Occurred()) { Py_DECREF(fb); return false; }
            if (i < 0) i += (long)ns;  // Python list indexing semantics of the specification
            if (j < 0) j += (long)ns;
            if (i < 0 || j < 0 || i >= (long)ns || j >= (long)ns) {
                Py_DECREF(fb);
                PyErr_SetString(PyExc_IndexError, "list index out of range");
                return false;
            }
            int a = ids[(size_t)i], b = ids[(size_t)j];
            if (a < b) std::swap(a, b);  // larger index first (feos_torch/gc_pcsaft.py:35)
            bcount[{a, b}]++;
        }
        Py_DECREF(fb);
        if ((int)scount.size() > MAXE || (int)bcount.size() > MAXE) {
            PyErr_Format(PyExc_ValueError, "a molecule may use at most %d distinct segment types and %d distinct bond types",
                         MAXE, MAXE);
            return false;
        }
        std::memset(out.b, 0, sizeof(out.b));
        int k = 0;
        for (auto& kv : scount) {
            if (kv.second > 255) { PyErr_SetString(PyExc_ValueError, "segment / bond multiplicity above 255"); return false; }
            out.b[k] = (uint8_t)kv.first;
            out.b[8 + k] = (uint8_t)kv.second;
            k++;
        }
        k = 0;
        for (auto& kv : bcount) {
            if (kv.second > 255) { PyErr_SetString(PyExc_ValueError, "segment / bond multiplicity above 255"); return false; }
            out.b[16 + k] = (uint8_t)kv.first.first;
            out.b[24 + k] = (uint8_t)kv.first.second;
            out.b[32 + k] = (uint8_t)kv.second;
            k++;
        }
        return true;
    }
};

PyObject* encode_rows(PyObject*, PyObject* args) {
    PyObject *ident, *seg_lists, *bond_lists, *out_obj;
    if (!PyArg_ParseTuple(args, "OOOO", &ident, &seg_lists, &bond_lists, &out_obj)) return nullptr;
    Encoder enc;
    {
        PyObject* fi = PySequence_Fast(ident, "segment_identifier must be a sequence of str");
        if (!fi) return nullptr;
        const Py_ssize_t S = PySequence_Fast_GET_SIZE(fi);
        if (S > 255) {
            Py_DECREF(fi);
            PyErr_SetString(PyExc_ValueError, "at most 255 segment types (uint8 row encoding)");
            return nullptr;
        }
        for (Py_ssize_t k = 0; k < S; k++) {
            PyObject* s = PySequence_Fast_GET_ITEM(fi, k);
            Py_ssize_t len = 0;
            const char* u = PyUnicode_Check(s) ? PyUnicode_AsUTF8AndSize(s, &len) : nullptr;
            if (!u) {
                Py_DECREF(fi);
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "segment identifiers must be str");
                return nullptr;
            }
            enc.by_name[std::string(u, (size_t)len)] = (int)k;  // later duplicates win, like the dict of the specification
        }
        Py_DECREF(fi);
    }
    Py_buffer view;
    if (PyObject_GetBuffer(out_obj, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    PyObject* fs = PySequence_Fast(seg_lists, "segment_lists must be a sequence");
    PyObject* fb = fs ? PySequence_Fast(bond_lists, "bond_lists must be a sequence") : nullptr;
    bool ok = fs && fb;
    if (ok) {
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(fs);
        if (PySequence_Fast_GET_SIZE(fb) != n) {
            PyErr_SetString(PyExc_ValueError, "segment_lists and bond_lists differ in length");
            ok = false;
        } else if (view.len != n * 80) {
            PyErr_SetString(PyExc_ValueError, "output buffer must hold n*80 bytes");
            ok = false;
        }
        uint8_t* rows = static_cast<uint8_t*>(view.buf);
        for (Py_ssize_t r = 0; ok && r < n; r++) {
            PyObject* rs = PySequence_Fast(PySequence_Fast_GET_ITEM(fs, r), "a row of segment_lists must hold two molecules");
            PyObject* rb = rs ? PySequence_Fast(PySequence_Fast_GET_ITEM(fb, r), "a row of bond_lists must hold two molecules") : nullptr;
            if (!rs || !rb || PySequence_Fast_GET_SIZE(rs) != 2 || PySequence_Fast_GET_SIZE(rb) != 2) {
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "each row must hold two molecules");
                ok = false;
            }
            for (int c = 0; ok && c < 2; c++) {
                PyObject* segs = PySequence_Fast_GET_ITEM(rs, c);
                PyObject* bonds = PySequence_Fast_GET_ITEM(rb, c);
                const Enc* e;
                auto it = enc.by_lists.find({segs, bonds});
                if (it != enc.by_lists.end()) {
                    e = &it->second;
                } else {
                    Enc fresh;
                    if (!enc.molecule(segs, bonds, fresh)) { ok = false; break; }
                    e = &enc.by_lists.emplace(std::make_pair(segs, bonds), fresh).first->second;
                }
                uint8_t* row = rows + r * 80;
                std::memcpy(row + 8 * c, e->b, 8);
                std::memcpy(row + 16 + 8 * c, e->b + 8, 8);
                std::memcpy(row + 32 + 8 * c, e->b + 16, 8);
                std::memcpy(row + 48 + 8 * c, e->b + 24, 8);
                std::memcpy(row + 64 + 8 * c, e->b + 32, 8);
            }
            Py_XDECREF(rs);
            Py_XDECREF(rb);
        }
    }
    Py_XDECREF(fs);
    Py_XDECREF(fb);
    PyBuffer_Release(&view);
    if (!ok) return nullptr;
    Py_RETURN_NONE;
}

PyMethodDef methods[] = {
    {"encode_rows", encode_rows, METH_VARARGS,
     "encode_rows(segment_identifier, segment_lists, bond_lists, out) -> None; fills out [n, 80] uint8"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_gc_encode", "native gc-PC-SAFT row encoder", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__gc_encode(void) { return PyModule_Create(&module); }
