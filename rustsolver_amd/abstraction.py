"""Host mirror of the card-abstraction plumbing (card_abstraction.rs) on top of the C ABI: bucket files, index_to_cluster,
deterministic dense-id maps.  The canonical hand index (rust_poker hand_indexer_s) is an input."""
import ctypes as C

import numpy as np

from . import _lib as L


def read_cluster_file(path):
    """flat LE u32 per canonical hand index (card_abstraction.rs:227-229)"""
    out, n = C.POINTER(C.c_uint32)(), C.c_size_t()
    L.check(L.load().rs_cluster_file_read(path.encode(), C.byref(out), C.byref(n)))
    try:
        return np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
    finally:
        L.load().rs_free_u32(out)


def write_cluster_file(path, clusters):
    """gen_abstraction/main.rs:372-380; refuses to overwrite (create_new)"""
    a = np.ascontiguousarray(clusters, dtype=np.uint32)
    L.check(L.load().rs_cluster_file_write(path.encode(), a.ctypes.data_as(C.POINTER(C.c_uint32)), len(a)))


def index_to_cluster(indices, cluster_arr=None):
    """card_abstraction.rs:20-29"""
    idx = np.ascontiguousarray(indices, dtype=np.uint64)
    out = np.empty(len(idx), dtype=np.uint64)
    if cluster_arr is None:
        rc = L.load().rs_index_to_cluster(None, 0, idx.ctypes.data_as(C.POINTER(C.c_uint64)), len(idx), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    else:
        arr = np.ascontiguousarray(cluster_arr, dtype=np.uint32)
        rc = L.load().rs_index_to_cluster(arr.ctypes.data_as(C.POINTER(C.c_uint32)), len(arr), idx.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          len(idx), out.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc == L.ERR_OOB:
        raise IndexError(L.load().rs_last_error().decode())
    L.check(rc)
    return out


class DenseMap:
    """cluster_map[player] + size[player] of generate_maps (card_abstraction.rs:75-184), first-appearance order"""

    def __init__(self, buckets):
        b = np.ascontiguousarray(buckets, dtype=np.uint64)
        h = C.c_void_p()
        L.check(L.load().rs_dense_map_create(b.ctypes.data_as(C.POINTER(C.c_uint64)), len(b), C.byref(h)))
        self._h = h

    def __len__(self):          # get_size
        return L.load().rs_dense_map_size(self._h)

    def lookup(self, buckets):  # get_cluster's map step; KeyError where Rust would unwrap() a None
        b = np.ascontiguousarray(buckets, dtype=np.uint64)
        out = np.empty(len(b), dtype=np.uint32)
        rc = L.load().rs_dense_map_lookup(self._h, b.ctypes.data_as(C.POINTER(C.c_uint64)), len(b), out.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc == L.ERR_OOB:
            raise KeyError(L.load().rs_last_error().decode())
        L.check(rc)
        return out

    def keys(self):
        out = np.empty(len(self), dtype=np.uint64)
        L.check(L.load().rs_dense_map_keys(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def __del__(self):
        try:
            L.load().rs_dense_map_destroy(self._h)
        except Exception:
            pass
