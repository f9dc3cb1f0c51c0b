// Host-side helper of detr_amd.HungarianMatcher: turns the flat index buffer of one matcher step
//     [src (S*M) | tgt (S*M) | status]            (int64, host memory)
// into the reference's return value (/root/reference/models/matcher.py:95), per output set a list
// of per-image tuples (src, tgt) of int64 tensors that view the buffer.
//
// A step of 6 sets x 64 images needs 768 tensor objects; creating (and later destroying) them one
// dispatcher call at a time costs more than the GPU work of the step.  Two things make it cheap:
//   * the views are built directly on the buffer's storage (no operator dispatch per view);
//   * the tensor objects of earlier calls are recycled when nobody else holds them any more (the
//     caller has dropped that step's indices): they are re-pointed at the new buffer instead of
//     being freed and allocated again.  Two generations are kept and used in turn, so that a loop
//     of the form `indices = matcher(...)`, which still holds the previous result while the next
//     one is made, finds the objects of the call before that free.  An object that is still
//     referenced anywhere -- Python reference, C++ reference, weak reference -- is never touched; a
//     fresh one takes its place in the pool.
// No GPU code here; the kernels are in the .hip files.
#include <Python.h>

#include <vector>

#include <ATen/core/Tensor.h>
#include <c10/core/TensorImpl.h>
#include <torch/csrc/autograd/python_variable.h>

namespace {

// strong references to the tensor objects handed out by the last two calls, used alternately
std::vector<PyObject *> g_pools[2];
unsigned g_turn = 0;

// The tensor object at pool slot `slot`, viewing `n` elements of `storage` from element `offset`.
// Returns a NEW reference (the pool keeps its own).
PyObject *view_at(std::vector<PyObject *> &g_pool, size_t slot, const c10::Storage &storage, c10::DispatchKeySet keys, caffe2::TypeMeta dtype,
                  int64_t offset, int64_t n)
{
    if (slot < g_pool.size()) {
        PyObject *obj = g_pool[slot];
        PyObject **weak = Py_TYPE(obj)->tp_weaklistoffset ? PyObject_GET_WEAKREFS_LISTPTR(obj) : nullptr;
        if (Py_REFCNT(obj) == 1 && (!weak || !*weak)) {
            const at::Tensor &t = THPVariable_Unpack(obj);
            if (t.use_count() == 1 && t.weak_use_count() == 1) {
                c10::TensorImpl *impl = t.unsafeGetTensorImpl();
                impl->set_storage_keep_dtype(storage);
                impl->set_storage_offset(offset);
                impl->set_sizes_contiguous({n});
                Py_INCREF(obj);
                return obj;
            }
        }
    }
    at::Tensor t = at::detail::make_tensor<c10::TensorImpl>(c10::TensorImpl::VIEW, c10::Storage(storage), keys, dtype);
    c10::TensorImpl *impl = t.unsafeGetTensorImpl();
    impl->set_storage_offset(offset);
    impl->set_sizes_contiguous({n});
    PyObject *obj = THPVariable_Wrap(std::move(t));
    if (!obj) return nullptr;
    if (slot < g_pool.size()) {
        Py_DECREF(g_pool[slot]);
        g_pool[slot] = obj;
    } else {
        g_pool.push_back(obj);
    }
    Py_INCREF(obj);
    return obj;
}

bool read_ints(PyObject *seq, std::vector<int64_t> &out, const char *what)
{
    PyObject *fast = PySequence_Fast(seq, what);
    if (!fast) return false;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    out.resize((size_t)n);
    for (Py_ssize_t i = 0; i < n; ++i) {
        out[(size_t)i] = PyLong_AsLongLong(PySequence_Fast_GET_ITEM(fast, i));
        if (out[(size_t)i] == -1 && PyErr_Occurred()) { Py_DECREF(fast); return false; }
    }
    Py_DECREF(fast);
    return true;
}

// index_pairs(flat, M, sizes, queries, offsets) -> list (sets) of lists (images) of (src, tgt)
//   flat     1-D contiguous int64 CPU tensor of at least 2 * S * M elements
//   sizes    targets per image (B ints); offsets: their exclusive prefix sums (B ints are read)
//   queries  queries per output set (S ints)
PyObject *index_pairs(PyObject *, PyObject *args)
{
    PyObject *flat_obj, *sizes_obj, *queries_obj, *offsets_obj;
    long long M;
    if (!PyArg_ParseTuple(args, "OLOOO", &flat_obj, &M, &sizes_obj, &queries_obj, &offsets_obj)) return nullptr;
    if (!THPVariable_Check(flat_obj)) {
        PyErr_SetString(PyExc_TypeError, "index_pairs: flat must be a tensor");
        return nullptr;
    }
    const at::Tensor &flat = THPVariable_Unpack(flat_obj);
    std::vector<int64_t> sizes, queries, offsets;
    if (!read_ints(sizes_obj, sizes, "sizes must be a sequence") ||
        !read_ints(queries_obj, queries, "queries must be a sequence") ||
        !read_ints(offsets_obj, offsets, "offsets must be a sequence"))
        return nullptr;
    const int64_t S = (int64_t)queries.size(), B = (int64_t)sizes.size();
    if (!flat.defined() || flat.dim() != 1 || flat.scalar_type() != at::kLong || !flat.is_contiguous() ||
        !flat.device().is_cpu() || M < 0 || flat.numel() < 2 * S * M || (int64_t)offsets.size() < B) {
        PyErr_SetString(PyExc_ValueError, "index_pairs: flat must be a contiguous 1-D int64 CPU tensor of >= 2*S*M elements");
        return nullptr;
    }
    for (int64_t b = 0; b < B; ++b)
        if (sizes[(size_t)b] < 0 || offsets[(size_t)b] < 0 || offsets[(size_t)b] + sizes[(size_t)b] > M) {
            PyErr_SetString(PyExc_ValueError, "index_pairs: sizes / offsets do not fit M");
            return nullptr;
        }
    const c10::Storage &storage = flat.storage();
    const c10::DispatchKeySet keys = flat.key_set();
    const caffe2::TypeMeta dtype = flat.dtype();
    const int64_t base = flat.storage_offset();

    PyObject *result = PyList_New(S);
    if (!result) return nullptr;
    std::vector<PyObject *> &g_pool = g_pools[g_turn++ & 1u];
    size_t slot = 0;
    for (int64_t s = 0; s < S; ++s) {
        PyObject *per_image = PyList_New(B);
        if (!per_image) { Py_DECREF(result); return nullptr; }
        PyList_SET_ITEM(result, s, per_image);
        for (int64_t b = 0; b < B; ++b) {
            const int64_t n = sizes[(size_t)b] < queries[(size_t)s] ? sizes[(size_t)b] : queries[(size_t)s];
            const int64_t lo = base + s * M + offsets[(size_t)b];
            PyObject *src = view_at(g_pool, slot++, storage, keys, dtype, lo, n);
            PyObject *tgt = src ? view_at(g_pool, slot++, storage, keys, dtype, lo + S * M, n) : nullptr;
            PyObject *pair = tgt ? PyTuple_New(2) : nullptr;
            if (!pair) {
                Py_XDECREF(src);
                Py_XDECREF(tgt);
                Py_DECREF(result);
                return nullptr;
            }
            PyTuple_SET_ITEM(pair, 0, src);
            PyTuple_SET_ITEM(pair, 1, tgt);
            PyList_SET_ITEM(per_image, b, pair);
        }
    }
    // objects beyond this call's needs go back to the interpreter
    while (g_pool.size() > slot) {
        Py_DECREF(g_pool.back());
        g_pool.pop_back();
    }
    return result;
}

PyObject *pool_size(PyObject *, PyObject *) { return PyLong_FromSize_t(g_pools[0].size() + g_pools[1].size()); }

PyObject *clear_pool(PyObject *, PyObject *)
{
    for (auto &pool : g_pools) {
        for (PyObject *o : pool) Py_DECREF(o);
        pool.clear();
    }
    Py_RETURN_NONE;
}

PyMethodDef methods[] = {
    {"index_pairs", index_pairs, METH_VARARGS, "flat index buffer -> per set, per image (src, tgt) views"},
    {"pool_size", pool_size, METH_NOARGS, "tensor objects kept for recycling"},
    {"clear_pool", clear_pool, METH_NOARGS, "drop the recycled tensor objects"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_index_views", "bulk creation of index views for detr_amd", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__index_views(void) { return PyModule_Create(&module); }
