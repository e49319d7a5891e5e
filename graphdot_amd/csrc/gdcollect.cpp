// _gdcollect -- CPython helper of the native graph packer (hip/hostlib.py):
// gathers one attribute table of a whole list of graphs into flat columns.
//
// The packer (gdh_pack_graphs, gdhost.cpp) takes the node / edge tables of all
// graphs of a call concatenated column by column.  Concatenating them in
// Python -- a dictionary lookup per graph and column, a type check per graph,
// numpy.concatenate of a thousand small arrays per column -- was 3 ms of the
// 4.5 ms that packing 1000 molecules took; here it is one pass over the
// objects through the C APIs of CPython and numpy (the columns are ndarrays:
// data pointer, size and type descriptor are struct members -- acquiring a
// buffer with a format string per column and graph, 14 000 of them for a
// thousand molecules, was most of the pass).
//
//   collect(graphs, attr, keys) -> (columns, lengths, formats) | None
//     graphs: list of Graph; attr: "nodes" / "edges"; keys: tuple of column
//     names (the first graph's).  columns: tuple of bytes, column k of all
//     graphs back to back; lengths: bytes of int64[len(graphs)], rows per
//     graph; formats: tuple of (struct format, itemsize) per column.
//   None: some graph's table has other columns, another element type, a
//   column that is not a one-dimensional contiguous ndarray of numbers -- the
//   caller takes the Python path.
//
// Role in the reference: the per-graph table handling at the top of
// OctileGraph.__init__ (graphdot/kernel/marginalized/_octilegraph.py:37-105).
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

// a column as the packer can take it: a one-dimensional C-contiguous ndarray
// without object references; its type descriptor (borrowed) or nullptr
PyArray_Descr *plain_column(PyObject *col) {
    if (!PyArray_Check(col)) return nullptr;
    PyArrayObject *a = reinterpret_cast<PyArrayObject *>(col);
    if (PyArray_NDIM(a) != 1 || !PyArray_IS_C_CONTIGUOUS(a)) return nullptr;
    PyArray_Descr *d = PyArray_DESCR(a);
    if (PyDataType_REFCHK(d) || PyDataType_ELSIZE(d) <= 0) return nullptr;
    return d;
}

bool same_type(PyArray_Descr *a, PyArray_Descr *b) {
    return a == b || PyArray_EquivTypes(a, b);
}

PyObject *collect(PyObject *, PyObject *args) {
    PyObject *graphs, *attr, *keys;
    if (!PyArg_ParseTuple(args, "O!UO!", &PyList_Type, &graphs, &attr, &PyTuple_Type, &keys)) return nullptr;
    const Py_ssize_t G = PyList_GET_SIZE(graphs), K = PyTuple_GET_SIZE(keys);
    std::vector<std::vector<char>> cols((size_t)K);
    std::vector<PyArray_Descr *> descr((size_t)K, nullptr);   // the first graph's (it stays alive: borrowed)
    std::vector<std::string> fmt((size_t)K);
    std::vector<Py_ssize_t> isz((size_t)K, 0);
    std::vector<int64_t> lengths((size_t)G, 0);
    PyObject *data_name = PyUnicode_InternFromString("_data");
    if (!data_name) return nullptr;
    bool mismatch = false;
    for (Py_ssize_t g = 0; g < G && !mismatch; ++g) {
        PyObject *frame = PyObject_GetAttr(PyList_GET_ITEM(graphs, g), attr);
        if (!frame) {
            Py_DECREF(data_name);
            return nullptr;
        }
        PyObject *data = PyObject_GetAttr(frame, data_name);
        Py_DECREF(frame);
        if (!data) {
            Py_DECREF(data_name);
            return nullptr;
        }
        if (!PyDict_Check(data) || PyDict_GET_SIZE(data) != K) {
            Py_DECREF(data);
            mismatch = true;
            break;
        }
        int64_t rows = -1;
        for (Py_ssize_t k = 0; k < K; ++k) {
            PyObject *col = PyDict_GetItemWithError(data, PyTuple_GET_ITEM(keys, k));   // borrowed
            if (!col) {
                if (PyErr_Occurred()) {
                    Py_DECREF(data);
                    Py_DECREF(data_name);
                    return nullptr;
                }
                mismatch = true;
                break;
            }
            PyArray_Descr *d = plain_column(col);      // not one: the Python path decides
            bool ok = d != nullptr;
            if (ok && g == 0) {
                descr[(size_t)k] = d;
                fmt[(size_t)k] = std::string(1, d->type);
                isz[(size_t)k] = (Py_ssize_t)PyDataType_ELSIZE(d);
            } else if (ok) {
                ok = same_type(descr[(size_t)k], d);
            }
            PyArrayObject *a = reinterpret_cast<PyArrayObject *>(col);
            const int64_t n = ok ? (int64_t)PyArray_DIM(a, 0) : -1;
            if (ok && rows < 0) rows = n;
            if (!ok || n != rows) {
                mismatch = true;
                break;
            }
            const size_t len = (size_t)n * (size_t)isz[(size_t)k];
            std::vector<char> &out = cols[(size_t)k];
            // (room for every graph at the size of the first ones: one
            // allocation per column instead of a dozen doublings)
            if (g == 1) out.reserve((out.size() + len) * (size_t)G / 2 + 64);
            out.insert(out.end(), (const char *)PyArray_DATA(a), (const char *)PyArray_DATA(a) + len);
        }
        Py_DECREF(data);
        lengths[(size_t)g] = rows < 0 ? 0 : rows;
    }
    Py_DECREF(data_name);
    if (mismatch) Py_RETURN_NONE;
    PyObject *tcols = PyTuple_New(K), *tfmt = PyTuple_New(K);
    if (!tcols || !tfmt) {
        Py_XDECREF(tcols);
        Py_XDECREF(tfmt);
        return nullptr;
    }
    for (Py_ssize_t k = 0; k < K; ++k) {
        PyObject *b = PyBytes_FromStringAndSize(cols[(size_t)k].data(), (Py_ssize_t)cols[(size_t)k].size());
        PyObject *f = Py_BuildValue("(sn)", fmt[(size_t)k].c_str(), isz[(size_t)k]);
        if (!b || !f) {
            Py_XDECREF(b);
            Py_XDECREF(f);
            Py_DECREF(tcols);
            Py_DECREF(tfmt);
            return nullptr;
        }
        PyTuple_SET_ITEM(tcols, k, b);
        PyTuple_SET_ITEM(tfmt, k, f);
    }
    PyObject *len = PyBytes_FromStringAndSize((const char *)lengths.data(), (Py_ssize_t)(lengths.size() * sizeof(int64_t)));
    if (!len) {
        Py_DECREF(tcols);
        Py_DECREF(tfmt);
        return nullptr;
    }
    return Py_BuildValue("(NNN)", tcols, len, tfmt);
}

// same_tables(graphs, attr) -> True | None.  True: the table `attr` of every
// graph has the first graph's columns, in its order, each a one-dimensional
// contiguous array of numbers of the same type -- i.e. DataFrame.rowtype() is
// the same for all of them (what Graph.has_unified_types establishes with two
// rowtype() calls per graph, 3-4 ms per 1000 graphs on the first call).  None:
// anything else -- the caller decides in Python.
PyObject *same_tables(PyObject *, PyObject *args) {
    PyObject *graphs, *attr;
    if (!PyArg_ParseTuple(args, "O!U", &PyList_Type, &graphs, &attr)) return nullptr;
    const Py_ssize_t G = PyList_GET_SIZE(graphs);
    PyObject *data_name = PyUnicode_InternFromString("_data");
    if (!data_name) return nullptr;
    std::vector<PyObject *> keys;         // the first graph's column names (owned)
    std::vector<PyArray_Descr *> descr;   // ... and element types (borrowed: graph 0 stays alive)
    bool same = true, error = false;
    for (Py_ssize_t g = 0; g < G && same && !error; ++g) {
        PyObject *frame = PyObject_GetAttr(PyList_GET_ITEM(graphs, g), attr);
        PyObject *data = frame ? PyObject_GetAttr(frame, data_name) : nullptr;
        Py_XDECREF(frame);
        if (!data) {
            error = true;
            break;
        }
        if (!PyDict_Check(data) || (g > 0 && PyDict_GET_SIZE(data) != (Py_ssize_t)keys.size())) {
            Py_DECREF(data);
            same = false;
            break;
        }
        Py_ssize_t pos = 0, k = 0;
        PyObject *key, *col;
        while (same && PyDict_Next(data, &pos, &key, &col)) {
            if (g > 0 && key != keys[(size_t)k]) {
                const int eq = PyObject_RichCompareBool(key, keys[(size_t)k], Py_EQ);
                if (eq < 0) error = true;
                if (eq <= 0) {
                    same = false;
                    break;
                }
            }
            PyArray_Descr *d = plain_column(col);
            bool ok = d != nullptr;
            if (ok && g == 0) {
                Py_INCREF(key);
                keys.push_back(key);
                descr.push_back(d);
            } else if (ok) {
                ok = same_type(descr[(size_t)k], d);
            }
            if (!ok) same = false;
            ++k;
        }
        Py_DECREF(data);
    }
    for (PyObject *k : keys) Py_DECREF(k);
    Py_DECREF(data_name);
    if (error) return nullptr;
    if (same) Py_RETURN_TRUE;
    Py_RETURN_NONE;
}

PyMethodDef methods[] = {
    {"collect", collect, METH_VARARGS, "collect(graphs, attr, keys) -> (columns, lengths, formats) | None"},
    {"same_tables", same_tables, METH_VARARGS, "same_tables(graphs, attr) -> True | None"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_gdcollect", "table gathering for the native graph packer", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__gdcollect(void) {
    import_array();      // (returns nullptr with an exception set if numpy cannot be bound)
    return PyModule_Create(&module);
}
