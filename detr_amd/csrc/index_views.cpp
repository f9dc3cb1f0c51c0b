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
//   * the lists and tuples are recycled with them: when the caller has dropped the whole result of the call
//     before the last (nobody holds the list of sets, a list of images or a tuple any more) and the new
//     result has the same numbers of sets and images, the same containers are handed out again around the
//     re-pointed tensors.  No container is allocated then, so the interpreter's cyclic collector -- which
//     counts container allocations and, with tens of thousands per step, runs a full collection every few
//     steps -- is not triggered by the matcher at all.
//
// The module also has the two per-tensor scans of a matcher call (scan_targets, scan_outputs): with 64
// images and 6 output sets a call touches 140 tensors, and checking dtype / device / layout and taking
// the data pointer of each costs more in the interpreter (about 0.3 us per tensor) than the GPU spends
// per image.  They only read tensor metadata; anything unusual (other dtypes, non-contiguous tensors,
// mismatched lengths) is reported as None and the Python code takes its general path, which raises the
// reference's exceptions.
// launch_targets is the usual step's host side in one call: output scan, pinned result tensor, the library's
// entry point (through its address).
// No GPU code here; the kernels are in the .hip files.
//
// Contract.  The pools are process-global and shared by every matcher object; every function here runs from
// start to end under the interpreter lock (none releases it), so calls from several Python threads are
// serialised and each sees the pools in a consistent state.  An object is re-pointed only when the pool (and,
// for a recycled result, its tuple) holds the ONLY references to it -- the reference counts are read under
// that same lock -- so a result somebody still holds, in any thread, through any matcher, is never changed;
// what interleaving callers lose is only the recycling (their objects are still held when their turn comes
// round again, fresh ones take their place).  The functions reach into tensor internals that are not a
// stable interface: the module records the torch version it was compiled against (`built_for_torch`) and the
// Python code refuses to use it under any other.
#include <Python.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#include <ATen/ATen.h>
#include <ATen/core/Tensor.h>
#include <c10/core/TensorImpl.h>
#include <torch/csrc/autograd/python_variable.h>

namespace {

// strong references to the tensor objects handed out by the last two calls, used alternately
std::vector<PyObject *> g_pools[2];
// ... and to the list of sets each of the two calls returned (nullptr: none kept)
PyObject *g_results[2] = {nullptr, nullptr};
unsigned g_turn = 0;

// Re-points tensor object `obj` at `n` elements of `storage` from `offset` if nobody but its `holders` known
// owners (the pool, possibly a recycled tuple) references it.
bool repoint(PyObject *obj, Py_ssize_t holders, const c10::Storage &storage, c10::DispatchKeySet keys, int64_t offset, int64_t n)
{
    PyObject **weak = Py_TYPE(obj)->tp_weaklistoffset ? PyObject_GET_WEAKREFS_LISTPTR(obj) : nullptr;
    if (Py_REFCNT(obj) != holders || (weak && *weak)) return false;
    const at::Tensor &t = THPVariable_Unpack(obj);
    // (an object is only re-pointed at storage of the kind it was made for: its dispatch keys stay)
    if (t.use_count() != 1 || t.weak_use_count() != 1 || t.key_set() != keys) return false;
    c10::TensorImpl *impl = t.unsafeGetTensorImpl();
    impl->set_storage_keep_dtype(storage);
    impl->set_storage_offset(offset);
    impl->set_sizes_contiguous({n});
    return true;
}

// The tensor object at pool slot `slot`, viewing `n` elements of `storage` from element `offset`.
// Returns a NEW reference (the pool keeps its own).
PyObject *view_at(std::vector<PyObject *> &g_pool, size_t slot, const c10::Storage &storage, c10::DispatchKeySet keys, caffe2::TypeMeta dtype,
                  int64_t offset, int64_t n)
{
    if (slot < g_pool.size() && repoint(g_pool[slot], 1, storage, keys, offset, n)) {
        Py_INCREF(g_pool[slot]);
        return g_pool[slot];
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
//   flat     1-D contiguous int64 tensor (host or device) of at least 2 * S * M elements
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
        flat.requires_grad() || M < 0 || flat.numel() < 2 * S * M || (int64_t)offsets.size() < B) {
        PyErr_SetString(PyExc_ValueError, "index_pairs: flat must be a contiguous 1-D int64 tensor of >= 2*S*M elements");
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

    const unsigned turn = g_turn++ & 1u;
    std::vector<PyObject *> &g_pool = g_pools[turn];
    size_t slot = 0;

    // The containers of the call before the last, if the caller has dropped all of them and the shape is the same:
    // list of S lists of B tuples of the pool's tensors, nobody but this module (and each other) holding any of them.
    PyObject *kept = g_results[turn];
    bool reuse = kept && Py_REFCNT(kept) == 1 && PyList_GET_SIZE(kept) == S && g_pool.size() == (size_t)(2 * S * B);
    for (int64_t s = 0; reuse && s < S; ++s) {
        PyObject *per_image = PyList_GET_ITEM(kept, s);
        reuse = PyList_CheckExact(per_image) && Py_REFCNT(per_image) == 1 && PyList_GET_SIZE(per_image) == B;
        for (int64_t b = 0; reuse && b < B; ++b) {
            PyObject *pair = PyList_GET_ITEM(per_image, b);
            const size_t at = (size_t)(2 * (s * B + b));
            reuse = PyTuple_CheckExact(pair) && Py_REFCNT(pair) == 1 && PyTuple_GET_SIZE(pair) == 2 &&
                    PyTuple_GET_ITEM(pair, 0) == g_pool[at] && PyTuple_GET_ITEM(pair, 1) == g_pool[at + 1];
        }
    }
    if (reuse) {
        for (int64_t s = 0; s < S; ++s) {
            PyObject *per_image = PyList_GET_ITEM(kept, s);
            for (int64_t b = 0; b < B; ++b) {
                const int64_t n = sizes[(size_t)b] < queries[(size_t)s] ? sizes[(size_t)b] : queries[(size_t)s];
                const int64_t lo = base + s * M + offsets[(size_t)b];
                PyObject *pair = PyList_GET_ITEM(per_image, b);
                for (int side = 0; side < 2; ++side, ++slot) {
                    const int64_t off = lo + (side ? S * M : 0);
                    // held by the pool and by its tuple; anything more: somebody kept that tensor, it is left alone
                    if (repoint(g_pool[slot], 2, storage, keys, off, n)) continue;
                    PyObject *kept_tensor = PyTuple_GET_ITEM(pair, side);
                    PyObject *fresh = view_at(g_pool, slot, storage, keys, dtype, off, n);      // a new object, also in the pool
                    if (!fresh) { Py_CLEAR(g_results[turn]); return nullptr; }
                    PyTuple_SET_ITEM(pair, side, fresh);       // (this module holds the only reference to the tuple)
                    Py_DECREF(kept_tensor);
                }
            }
        }
        Py_INCREF(kept);
        return kept;
    }
    Py_CLEAR(g_results[turn]);      // its tuples' references to the pool's tensors go with it

    PyObject *result = PyList_New(S);
    if (!result) return nullptr;
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
    Py_INCREF(result);
    g_results[turn] = result;
    return result;
}

PyObject *pool_size(PyObject *, PyObject *) { return PyLong_FromSize_t(g_pools[0].size() + g_pools[1].size()); }

PyObject *clear_pool(PyObject *, PyObject *)
{
    for (auto &kept : g_results) Py_CLEAR(kept);
    for (auto &pool : g_pools) {
        for (PyObject *o : pool) Py_DECREF(o);
        pool.clear();
    }
    Py_RETURN_NONE;
}

// Borrowed reference to d[key] if d is a dict and the value is a tensor, else nullptr.
const at::Tensor *dict_tensor(PyObject *d, PyObject *key)
{
    if (!PyDict_Check(d)) return nullptr;
    PyObject *v = PyDict_GetItemWithError(d, key);
    if (!v) { PyErr_Clear(); return nullptr; }
    if (!THPVariable_Check(v)) return nullptr;
    return &THPVariable_Unpack(v);
}

PyObject *g_key_labels = nullptr, *g_key_boxes = nullptr, *g_key_logits = nullptr, *g_key_pboxes = nullptr;

// scan_targets(targets, device_index) -> None, or
//   (sizes: list[int], total: int, max_t: int, pointers: bytes, tensors: list, versions: bytes,
//    offsets: list[int] (B + 1 exclusive prefix sums), meta: bytes (int32 offsets (B + 1) | processing order (B)))
// pointers: per image (labels data pointer, boxes data pointer) as two uint64; tensors: labels_0, boxes_0,
// labels_1, ... (kept by the caller: the pointers stay valid and a later call can be compared with them);
// versions: their version counters as uint32.  None unless every image has int64 1-D contiguous labels and
// float32 [T, 4] contiguous boxes of equal length on CUDA device `device_index`.
PyObject *scan_targets(PyObject *, PyObject *args)
{
    PyObject *targets;
    long device_index;
    if (!PyArg_ParseTuple(args, "Ol", &targets, &device_index)) return nullptr;
    PyObject *fast = PySequence_Fast(targets, "targets must be a sequence");
    if (!fast) return nullptr;
    const Py_ssize_t B = PySequence_Fast_GET_SIZE(fast);
    std::vector<uint64_t> ptrs((size_t)(2 * B));
    std::vector<uint32_t> versions((size_t)(2 * B));
    PyObject *sizes = PyList_New(B), *tensors = PyList_New(2 * B);
    int64_t total = 0, max_t = 0;
    bool ok = sizes && tensors;
    for (Py_ssize_t b = 0; ok && b < B; ++b) {
        PyObject *d = PySequence_Fast_GET_ITEM(fast, b);
        const at::Tensor *l = dict_tensor(d, g_key_labels), *x = dict_tensor(d, g_key_boxes);
        if (!l || !x) { ok = false; break; }
        const int64_t n = x->dim() >= 1 ? x->size(0) : -1;
        ok = n >= 0 && l->scalar_type() == at::kLong && x->scalar_type() == at::kFloat && l->dim() == 1 &&
             l->size(0) == n && x->numel() == 4 * n && (x->dim() == 2 || n == 0) && l->is_contiguous() &&
             x->is_contiguous() &&
             (device_index < 0 ? (l->device().is_cpu() && x->device().is_cpu())      // -1: host tensors (tests)
                               : (l->device().is_cuda() && x->device().is_cuda() &&
                                  l->device().index() == device_index && x->device().index() == device_index)) &&
             (n == 0 || (reinterpret_cast<uintptr_t>(x->data_ptr()) & 15) == 0);
        if (!ok) break;
        ptrs[(size_t)(2 * b)] = reinterpret_cast<uint64_t>(l->data_ptr());
        ptrs[(size_t)(2 * b + 1)] = reinterpret_cast<uint64_t>(x->data_ptr());
        versions[(size_t)(2 * b)] = l->_version();
        versions[(size_t)(2 * b + 1)] = x->_version();
        PyObject *lo = PyDict_GetItem(d, g_key_labels), *xo = PyDict_GetItem(d, g_key_boxes);
        Py_INCREF(lo);
        Py_INCREF(xo);
        PyList_SET_ITEM(tensors, 2 * b, lo);
        PyList_SET_ITEM(tensors, 2 * b + 1, xo);
        PyList_SET_ITEM(sizes, b, PyLong_FromLongLong(n));
        total += n;
        max_t = n > max_t ? n : max_t;
    }
    Py_DECREF(fast);
    if (!ok) {
        Py_XDECREF(sizes);
        Py_XDECREF(tensors);
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    // [offsets (B + 1) | order (B)]: exclusive prefix sums, and the images by falling size (stable): the
    // assignment kernel's tail is set by the largest problems, they go first
    std::vector<int32_t> meta((size_t)(2 * B + 1));
    PyObject *offsets = PyList_New(B + 1);
    if (!offsets || total > INT32_MAX) {
        Py_XDECREF(offsets);
        Py_DECREF(sizes);
        Py_DECREF(tensors);
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    int64_t run = 0;
    for (Py_ssize_t b = 0; b <= B; ++b) {
        meta[(size_t)b] = (int32_t)run;
        PyList_SET_ITEM(offsets, b, PyLong_FromLongLong(run));
        if (b < B) run += PyLong_AsLongLong(PyList_GET_ITEM(sizes, b));
    }
    std::vector<int32_t> order((size_t)B);
    for (Py_ssize_t b = 0; b < B; ++b) order[(size_t)b] = (int32_t)b;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t c) {
        return meta[(size_t)a + 1] - meta[(size_t)a] > meta[(size_t)c + 1] - meta[(size_t)c];
    });
    for (Py_ssize_t b = 0; b < B; ++b) meta[(size_t)(B + 1 + b)] = order[(size_t)b];
    PyObject *pb = PyBytes_FromStringAndSize(reinterpret_cast<const char *>(ptrs.data()), (Py_ssize_t)(ptrs.size() * 8));
    PyObject *vb = PyBytes_FromStringAndSize(reinterpret_cast<const char *>(versions.data()), (Py_ssize_t)(versions.size() * 4));
    PyObject *mb = PyBytes_FromStringAndSize(reinterpret_cast<const char *>(meta.data()), (Py_ssize_t)(meta.size() * 4));
    return Py_BuildValue("(NLLNNNNN)", sizes, (long long)total, (long long)max_t, pb, tensors, vb, offsets, mb);
}

// same_targets(targets, tensors, versions) -> bool: `targets` are the very tensor objects of an earlier
// scan_targets, with unchanged version counters.
PyObject *same_targets(PyObject *, PyObject *args)
{
    PyObject *targets, *tensors, *versions;
    if (!PyArg_ParseTuple(args, "OOS", &targets, &tensors, &versions)) return nullptr;
    if (!PyList_Check(tensors)) Py_RETURN_FALSE;
    PyObject *fast = PySequence_Fast(targets, "targets must be a sequence");
    if (!fast) return nullptr;
    const Py_ssize_t B = PySequence_Fast_GET_SIZE(fast);
    bool same = PyList_GET_SIZE(tensors) == 2 * B && PyBytes_GET_SIZE(versions) == 8 * B;
    const uint32_t *ver = reinterpret_cast<const uint32_t *>(PyBytes_AS_STRING(versions));
    for (Py_ssize_t b = 0; same && b < B; ++b) {
        PyObject *d = PySequence_Fast_GET_ITEM(fast, b);
        if (!PyDict_Check(d)) { same = false; break; }
        PyObject *lo = PyDict_GetItem(d, g_key_labels), *xo = PyDict_GetItem(d, g_key_boxes);
        same = lo && xo && lo == PyList_GET_ITEM(tensors, 2 * b) && xo == PyList_GET_ITEM(tensors, 2 * b + 1) &&
               THPVariable_Check(lo) && THPVariable_Check(xo) &&
               THPVariable_Unpack(lo)._version() == ver[2 * b] && THPVariable_Unpack(xo)._version() == ver[2 * b + 1];
    }
    Py_DECREF(fast);
    if (same) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

// The C ABI's descriptor of one output set (include/detr_match.h, DmSet).
struct SetRecord {
    const void *logits, *boxes;
    int64_t cost_off, out_off;
    int32_t Q, q_stride;
    int32_t logits_stride, boxes_stride;      // floats from one image to the next; 0 = contiguous
};
static_assert(sizeof(SetRecord) == 48, "must match DmSet");

// [B, Q, K] with dense rows inside every image; the images themselves may lie further apart (what
// tensor.split(sizes, dim=1) of a contiguous [B, Q + Q', K] tensor yields).  *image_stride: 0 when contiguous.
bool rows_dense(const at::Tensor &t, int32_t *image_stride)
{
    const int64_t B = t.size(0), Q = t.size(1), K = t.size(2);
    if (t.is_contiguous()) { *image_stride = 0; return true; }
    if (t.stride(2) != 1 || t.stride(1) != K) return false;
    const int64_t st = t.stride(0);
    if (B > 1 && (st < Q * K || st > INT32_MAX)) return false;
    *image_stride = B > 1 ? (int32_t)st : 0;
    return true;
}

// scan_outputs(outputs_list, batch, num_targets) -> None, or
//   (device_index, C, queries: list[int], descriptors: bytes (S DmSet records), cost_floats: int)
// None unless every set has float32 pred_logits [batch, Q, C] and pred_boxes [batch, Q, 4] (16-byte aligned, also
// from image to image) with dense rows per image -- contiguous, or the images a fixed distance apart -- on one CUDA
// device with one C.  Cost regions are laid out one after the other, row stride = Q
// rounded up to 32; index regions at s * num_targets.
struct OutputScan {
    std::vector<SetRecord> recs;
    PyObject *queries = nullptr;      // new reference, list[int]
    int64_t cost_off = 0, C = 0;
    int device_index = -1;
};

// false (with or without a Python error set) unless every set is of the usual form
bool scan_output_sets(PyObject *outputs, long long batch, long long M, bool allow_host, OutputScan &r)
{
    PyObject *fast = PySequence_Fast(outputs, "outputs must be a sequence");
    if (!fast) return false;
    const Py_ssize_t S = PySequence_Fast_GET_SIZE(fast);
    r.recs.resize((size_t)S);
    r.queries = PyList_New(S);
    bool ok = r.queries != nullptr && S > 0;
    for (Py_ssize_t s = 0; ok && s < S; ++s) {
        PyObject *d = PySequence_Fast_GET_ITEM(fast, s);
        const at::Tensor *l = dict_tensor(d, g_key_logits), *x = dict_tensor(d, g_key_pboxes);
        if (!l || !x) { ok = false; break; }
        ok = l->dim() == 3 && x->dim() == 3 && l->scalar_type() == at::kFloat && x->scalar_type() == at::kFloat &&
             l->size(0) == batch && x->size(0) == batch && l->size(1) == x->size(1) && x->size(2) == 4 &&
             l->size(1) >= 1 && l->size(1) <= INT32_MAX - 32 && l->size(2) >= 1 &&
             (l->device().is_cuda() || allow_host) && x->device() == l->device() &&
             (reinterpret_cast<uintptr_t>(x->data_ptr()) & 15) == 0;
        int32_t l_stride = 0, x_stride = 0;
        ok = ok && rows_dense(*l, &l_stride) && rows_dense(*x, &x_stride) && (x_stride & 3) == 0;
        if (!ok) break;
        if (s == 0) { r.C = l->size(2); r.device_index = l->device().index(); }
        ok = l->size(2) == r.C && l->device().index() == r.device_index;
        if (!ok) break;
        const int64_t Q = l->size(1), stride = (Q + 31) / 32 * 32;
        r.recs[(size_t)s] = SetRecord{l->data_ptr(), x->data_ptr(), r.cost_off, (int64_t)s * M, (int32_t)Q, (int32_t)stride,
                                      l_stride, x_stride};
        r.cost_off += M * stride;
        PyList_SET_ITEM(r.queries, s, PyLong_FromLongLong(Q));
    }
    Py_DECREF(fast);
    if (!ok) {
        Py_XDECREF(r.queries);
        r.queries = nullptr;
    }
    return ok;
}

PyObject *scan_outputs(PyObject *, PyObject *args)
{
    PyObject *outputs;
    long long batch, M;
    int allow_host = 0;      // tests only: accept host tensors
    if (!PyArg_ParseTuple(args, "OLL|p", &outputs, &batch, &M, &allow_host)) return nullptr;
    OutputScan r;
    if (!scan_output_sets(outputs, batch, M, allow_host != 0, r)) {
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    PyObject *rb = PyBytes_FromStringAndSize(reinterpret_cast<const char *>(r.recs.data()), (Py_ssize_t)(r.recs.size() * sizeof(SetRecord)));
    return Py_BuildValue("(iLNNL)", r.device_index, (long long)r.C, r.queries, rb, (long long)r.cost_off);
}

// dm_match_targets_f32 of include/detr_match.h, called through its address (this module does not link the
// kernel library)
typedef int (*MatchTargetsFn)(const void *sets, int S, int B, int C, const uint64_t *pointers, const int32_t *meta,
                              int max_T, const void *params, int64_t *labels_buf, float *boxes_buf, int32_t *meta_buf,
                              float *workspace, int64_t *src_out, int64_t *tgt_out, int64_t *flat_out, int32_t *status,
                              void *stream, int *packed);

// launch_targets(outputs, batch, M, max_t, pointers, meta, entry, params, workspace, workspace_floats, stream,
//                device_index)
// One matcher step for a caller that wants the reference's CPU tuples: checks the output sets, allocates the
// pinned host tensor [status | src (S*M) | tgt (S*M)] the kernels store their indices into, clears its status
// word and calls dm_match_targets_f32 (entry: its address; params: address of the DmParams; pointers, meta:
// from scan_targets; workspace: device floats for the rows of a cost block that do not fit in LDS) without
// buffers for packed targets.  Returns
//   None                        unusual output sets, or not on device `device_index`: nothing done
//   int                         floats of workspace the call needs, more than it was handed: nothing done
//   (host, queries, rc, packed) rc as returned by the entry point; 1 (DM_NEED_BUFFERS): the step has to pack its targets and
//                               was handed no room for them (labels_buf, boxes_buf, meta_buf: optional trailing
//                               arguments, device addresses), nothing was launched; packed: the targets were packed
PyObject *launch_targets(PyObject *, PyObject *args)
{
    PyObject *outputs, *pointers, *meta;
    long long batch, M, max_t, ws_floats;
    unsigned long long entry, params, workspace, stream, labels_buf = 0, boxes_buf = 0, meta_buf = 0;
    int device_index;
    if (!PyArg_ParseTuple(args, "OLLLSSKKKLKi|KKK", &outputs, &batch, &M, &max_t, &pointers, &meta, &entry, &params,
                          &workspace, &ws_floats, &stream, &device_index, &labels_buf, &boxes_buf, &meta_buf))
        return nullptr;
    if (batch < 1 || M < 0 || batch > INT32_MAX / 2 || max_t > INT32_MAX || !entry || !params ||
        PyBytes_GET_SIZE(pointers) != 16 * batch || PyBytes_GET_SIZE(meta) != 4 * (2 * batch + 1)) {
        PyErr_SetString(PyExc_ValueError, "launch_targets: inconsistent arguments");
        return nullptr;
    }
    OutputScan r;
    if (!scan_output_sets(outputs, batch, M, false, r)) {
        if (PyErr_Occurred()) return nullptr;
        Py_RETURN_NONE;
    }
    const int64_t S = (int64_t)r.recs.size(), SM = S * M;
    if (r.device_index != device_index || S > INT32_MAX || r.C > INT32_MAX) {
        Py_DECREF(r.queries);
        Py_RETURN_NONE;
    }
    const int64_t need = r.cost_off > 1 ? r.cost_off : 1;
    if (need > ws_floats || !workspace) {
        Py_DECREF(r.queries);
        return PyLong_FromLongLong(need);
    }
    at::Tensor host;
    try {
        host = at::empty({2 * SM + 1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    } catch (const std::exception &e) {
        Py_DECREF(r.queries);
        PyErr_SetString(PyExc_RuntimeError, e.what());
        return nullptr;
    }
    int64_t *base = host.data_ptr<int64_t>();
    base[0] = 0;        // the status word: the kernels only ever OR bits into it
    int packed = 0;
    const int rc = reinterpret_cast<MatchTargetsFn>(entry)(
        r.recs.data(), (int)S, (int)batch, (int)r.C, reinterpret_cast<const uint64_t *>(PyBytes_AS_STRING(pointers)),
        reinterpret_cast<const int32_t *>(PyBytes_AS_STRING(meta)), (int)max_t, reinterpret_cast<const void *>(params),
        reinterpret_cast<int64_t *>(labels_buf), reinterpret_cast<float *>(boxes_buf), reinterpret_cast<int32_t *>(meta_buf),
        reinterpret_cast<float *>(workspace), base + 1, base + 1 + SM, nullptr,
        reinterpret_cast<int32_t *>(base), reinterpret_cast<void *>(stream), &packed);
    PyObject *ht = THPVariable_Wrap(host);
    if (!ht) {
        Py_DECREF(r.queries);
        return nullptr;
    }
    return Py_BuildValue("(NNii)", ht, r.queries, rc, packed);
}

PyMethodDef methods[] = {
    {"scan_targets", scan_targets, METH_VARARGS, "metadata and data pointers of a list of target dicts"},
    {"same_targets", same_targets, METH_VARARGS, "are these the unchanged tensors of an earlier scan_targets"},
    {"scan_outputs", scan_outputs, METH_VARARGS, "checked DmSet descriptors of a list of output dicts"},
    {"launch_targets", launch_targets, METH_VARARGS, "pinned index tensor + one dm_match_targets_f32 call"},
    {"index_pairs", index_pairs, METH_VARARGS, "flat index buffer -> per set, per image (src, tgt) views"},
    {"pool_size", pool_size, METH_NOARGS, "tensor objects kept for recycling"},
    {"clear_pool", clear_pool, METH_NOARGS, "drop the recycled tensor objects"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_index_views", "bulk creation of index views for detr_amd", -1, methods};

}  // namespace

#ifndef DM_BUILT_FOR_TORCH
#define DM_BUILT_FOR_TORCH "unknown"
#endif
// (also found by build_ext.py in the file's bytes: a library built for another torch is rebuilt)
extern "C" const char dm_index_views_stamp[] = "detr_amd_index_views built for torch " DM_BUILT_FOR_TORCH ";";

PyMODINIT_FUNC PyInit__index_views(void)
{
    // compiled against one CPython's object layout (reference counts, weak-reference lists, tuple items)
    int major = 0, minor = 0;
    if (sscanf(Py_GetVersion(), "%d.%d", &major, &minor) != 2 || major != PY_MAJOR_VERSION || minor != PY_MINOR_VERSION) {
        PyErr_Format(PyExc_ImportError, "_index_views was built for Python %d.%d", PY_MAJOR_VERSION, PY_MINOR_VERSION);
        return nullptr;
    }
    g_key_labels = PyUnicode_InternFromString("labels");
    g_key_boxes = PyUnicode_InternFromString("boxes");
    g_key_logits = PyUnicode_InternFromString("pred_logits");
    g_key_pboxes = PyUnicode_InternFromString("pred_boxes");
    PyObject *m = PyModule_Create(&module);
    if (m && PyModule_AddStringConstant(m, "built_for_torch", DM_BUILT_FOR_TORCH) < 0) {
        Py_DECREF(m);
        return nullptr;
    }
    return m;
}
