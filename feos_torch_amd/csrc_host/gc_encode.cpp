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

// shared front half of both entry points: identifier table
bool load_identifiers(Encoder& enc, PyObject* ident) {
    PyObject* fi = PySequence_Fast(ident, "segment_identifier must be a sequence of str");
    if (!fi) return false;
    const Py_ssize_t S = PySequence_Fast_GET_SIZE(fi);
    if (S > 255) {
        Py_DECREF(fi);
        PyErr_SetString(PyExc_ValueError, "at most 255 segment types (uint8 row encoding)");
        return false;
    }
    for (Py_ssize_t k = 0; k < S; k++) {
        PyObject* s = PySequence_Fast_GET_ITEM(fi, k);
        Py_ssize_t len = 0;
        const char* u = PyUnicode_Check(s) ? PyUnicode_AsUTF8AndSize(s, &len) : nullptr;
        if (!u) {
            Py_DECREF(fi);
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "segment identifiers must be str");
            return false;
        }
        enc.by_name[std::string(u, (size_t)len)] = (int)k;  // later duplicates win, like the dict of the specification
    }
    Py_DECREF(fi);
    return true;
}

// Walks the rows and hands `sink(row, component, encoding, id)` every molecule: id = running number of the distinct
// (segments, bonds) object pair, encoding = its 40 bytes.  Batches draw their molecules from a small library and re-use the
// same list objects (the reference's tests build their rows that way): a molecule is encoded ONCE per distinct pair, and a row
// costs two probes of a direct-mapped pointer cache in front of the hash map.
template <class Sink>
bool walk_rows(Encoder& enc, PyObject* fs, PyObject* fb, Py_ssize_t n, std::vector<const Enc*>& unique, Sink&& sink) {
    constexpr size_t CACHE = 4096;
    struct Slot { PyObject *segs, *bonds; const Enc* e; int id; };
    std::vector<Slot> cache(CACHE, Slot{nullptr, nullptr, nullptr, -1});
    std::unordered_map<const Enc*, int> ids;
    PyObject** seg_rows = PySequence_Fast_ITEMS(fs);
    PyObject** bond_rows = PySequence_Fast_ITEMS(fb);
    bool ok = true;
    for (Py_ssize_t r = 0; ok && r < n; r++) {
        PyObject *row_s = seg_rows[r], *row_b = bond_rows[r];
        PyObject *rs = nullptr, *rb = nullptr;  // owned only on the generic path
        PyObject *mol_s[2], *mol_b[2];
        if (PyList_CheckExact(row_s) && PyList_CheckExact(row_b) && PyList_GET_SIZE(row_s) == 2 && PyList_GET_SIZE(row_b) == 2) {
            mol_s[0] = PyList_GET_ITEM(row_s, 0); mol_s[1] = PyList_GET_ITEM(row_s, 1);
            mol_b[0] = PyList_GET_ITEM(row_b, 0); mol_b[1] = PyList_GET_ITEM(row_b, 1);
        } else {
            rs = PySequence_Fast(row_s, "a row of segment_lists must hold two molecules");
            rb = rs ? PySequence_Fast(row_b, "a row of bond_lists must hold two molecules") : nullptr;
            if (!rs || !rb || PySequence_Fast_GET_SIZE(rs) != 2 || PySequence_Fast_GET_SIZE(rb) != 2) {
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "each row must hold two molecules");
                ok = false;
            } else {
                for (int c = 0; c < 2; c++) { mol_s[c] = PySequence_Fast_GET_ITEM(rs, c); mol_b[c] = PySequence_Fast_GET_ITEM(rb, c); }
            }
        }
        for (int c = 0; ok && c < 2; c++) {
            PyObject *segs = mol_s[c], *bonds = mol_b[c];
            Slot& slot = cache[((reinterpret_cast<uintptr_t>(segs) >> 4) * 0x9E3779B97F4A7C15ull ^ (reinterpret_cast<uintptr_t>(bonds) >> 4)) % CACHE];
            if (!(slot.segs == segs && slot.bonds == bonds)) {
                const Enc* e;
                auto it = enc.by_lists.find({segs, bonds});
                if (it != enc.by_lists.end()) {
                    e = &it->second;
                } else {
                    Enc fresh;
                    if (!enc.molecule(segs, bonds, fresh)) { ok = false; break; }
                    e = &enc.by_lists.emplace(std::make_pair(segs, bonds), fresh).first->second;  // (node-based map: stable addresses)
                }
                auto jt = ids.find(e);
                int id;
                if (jt == ids.end()) {
                    id = (int)unique.size();
                    unique.push_back(e);
                    ids.emplace(e, id);
                } else {
                    id = jt->second;
                }
                slot = Slot{segs, bonds, e, id};
            }
            sink(r, c, slot.e, slot.id);
        }
        Py_XDECREF(rs);
        Py_XDECREF(rb);
    }
    return ok;
}

PyObject* encode_rows(PyObject*, PyObject* args) {
    PyObject *ident, *seg_lists, *bond_lists, *out_obj;
    if (!PyArg_ParseTuple(args, "OOOO", &ident, &seg_lists, &bond_lists, &out_obj)) return nullptr;
    Encoder enc;
    if (!load_identifiers(enc, ident)) return nullptr;
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
        if (ok) {
            uint8_t* rows = static_cast<uint8_t*>(view.buf);
            std::vector<const Enc*> unique;
            ok = walk_rows(enc, fs, fb, n, unique, [rows](Py_ssize_t r, int c, const Enc* e, int) {
                uint8_t* row = rows + r * 80;
                std::memcpy(row + 8 * c, e->b, 8);
                std::memcpy(row + 16 + 8 * c, e->b + 8, 8);
                std::memcpy(row + 32 + 8 * c, e->b + 16, 8);
                std::memcpy(row + 48 + 8 * c, e->b + 24, 8);
                std::memcpy(row + 64 + 8 * c, e->b + 32, 8);
            });
        }
    }
    Py_XDECREF(fs);
    Py_XDECREF(fb);
    PyBuffer_Release(&view);
    if (!ok) return nullptr;
    Py_RETURN_NONE;
}

// encode_indices(segment_identifier, segment_lists, bond_lists, index_out) -> bytes
//   index_out: writable C-contiguous int32 buffer [n, 2]: for every row and component the number of its molecule in the
//   returned table; the table holds 40 bytes per DISTINCT molecule (seg_id[8], seg_cnt[8], bond_a[8], bond_b[8],
//   bond_cnt[8]).  The 80-byte rows are assembled from the two on the device (feos_torch_amd.gc_pcsaft.encode_rows_device):
//   the host writes 8 bytes per row instead of 80.
PyObject* encode_indices(PyObject*, PyObject* args) {
    PyObject *ident, *seg_lists, *bond_lists, *out_obj;
    if (!PyArg_ParseTuple(args, "OOOO", &ident, &seg_lists, &bond_lists, &out_obj)) return nullptr;
    Encoder enc;
    if (!load_identifiers(enc, ident)) return nullptr;
    Py_buffer view;
    if (PyObject_GetBuffer(out_obj, &view, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) return nullptr;
    PyObject* fs = PySequence_Fast(seg_lists, "segment_lists must be a sequence");
    PyObject* fb = fs ? PySequence_Fast(bond_lists, "bond_lists must be a sequence") : nullptr;
    bool ok = fs && fb;
    std::vector<const Enc*> unique;
    if (ok) {
        const Py_ssize_t n = PySequence_Fast_GET_SIZE(fs);
        if (PySequence_Fast_GET_SIZE(fb) != n) {
            PyErr_SetString(PyExc_ValueError, "segment_lists and bond_lists differ in length");
            ok = false;
        } else if (view.len != n * 8) {
            PyErr_SetString(PyExc_ValueError, "index buffer must hold n*2 int32");
            ok = false;
        }
        if (ok) {
            int32_t* idx = static_cast<int32_t*>(view.buf);
            ok = walk_rows(enc, fs, fb, n, unique, [idx](Py_ssize_t r, int c, const Enc*, int id) { idx[2 * r + c] = id; });
        }
    }
    Py_XDECREF(fs);
    Py_XDECREF(fb);
    PyBuffer_Release(&view);
    if (!ok) return nullptr;
    PyObject* table = PyBytes_FromStringAndSize(nullptr, (Py_ssize_t)unique.size() * 40);
    if (!table) return nullptr;
    char* dst = PyBytes_AS_STRING(table);
    for (size_t k = 0; k < unique.size(); k++) std::memcpy(dst + 40 * k, unique[k]->b, 40);
    return table;
}

PyMethodDef methods[] = {
    {"encode_rows", encode_rows, METH_VARARGS,
     "encode_rows(segment_identifier, segment_lists, bond_lists, out) -> None; fills out [n, 80] uint8"},
    {"encode_indices", encode_indices, METH_VARARGS,
     "encode_indices(segment_identifier, segment_lists, bond_lists, index_out[n,2] int32) -> bytes (40 per distinct molecule)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_gc_encode", "native gc-PC-SAFT row encoder", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__gc_encode(void) { return PyModule_Create(&module); }
