"""Minimal HDF5 access through ctypes on the system's libhdf5 (no h5py in this image).

Only what the XDMF mesh / result files of the drivers need: contiguous n-d datasets of int32 / int64 / float32 /
float64 in (nested) groups -- the layout DOLFINx's `XDMFFile` writes (`mesh.h5:/Mesh/mesh/topology`, ...).  The
library is looked up in `KNPEMI_HDF5_LIB`, the loader path and `/opt/conda/lib`; a missing library is an error at
the first use, not at import.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os

import numpy as np

_lib = None
hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5T_INTEGER, H5T_FLOAT = 0, 1


def lib():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("KNPEMI_HDF5_LIB"), ctypes.util.find_library("hdf5"), "libhdf5.so",
             "/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103"]
    err = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            break
        except OSError as e:
            err = e
    else:
        raise RuntimeError(f"no HDF5 library found (set KNPEMI_HDF5_LIB): {err}")
    L.H5open()
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dclose": (C.c_int, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (C.c_int, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tclose": (C.c_int, [hid_t]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (C.c_int, [hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    # silence the library's own error stack printing; failures are raised from the return codes
    try:
        L.H5Eset_auto2.argtypes = [hid_t, C.c_void_p, C.c_void_p]
        L.H5Eset_auto2(0, None, None)
    except AttributeError:
        pass
    _lib = L
    return L


def _native(dtype):
    name = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g",
            np.dtype(np.int64): "H5T_NATIVE_INT64_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g"}[np.dtype(dtype)]
    return hid_t.in_dll(lib(), name).value


class File:
    """`File(path, "r")` / `File(path, "w")` with `read(name)` and `write(name, array)`."""

    def __init__(self, path, mode="r"):
        L = lib()
        self.path = path
        if mode == "r":
            self.id = L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, 0)
        elif mode == "w":
            self.id = L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, 0, 0)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self.id < 0:
            raise OSError(f"cannot open HDF5 file {path!r} (mode {mode})")

    def close(self):
        if self.id >= 0:
            lib().H5Fclose(self.id)
            self.id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def exists(self, name):
        L = lib()
        cur = ""
        for part in name.strip("/").split("/"):
            cur += "/" + part
            if L.H5Lexists(self.id, cur.encode(), 0) <= 0:
                return False
        return True

    def read(self, name):
        L = lib()
        d = L.H5Dopen2(self.id, name.encode(), 0)
        if d < 0:
            raise KeyError(f"{self.path}: no dataset {name!r}")
        try:
            sp, tp = L.H5Dget_space(d), L.H5Dget_type(d)
            nd = L.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sp, dims, None)
            cls, size = L.H5Tget_class(tp), L.H5Tget_size(tp)
            L.H5Sclose(sp)
            L.H5Tclose(tp)
            if cls == H5T_FLOAT:
                dtype = np.float64 if size > 4 else np.float32
            elif cls == H5T_INTEGER:
                dtype = np.int64 if size > 4 else np.int32
            else:
                raise TypeError(f"{self.path}:{name}: unsupported HDF5 type class {cls}")
            out = np.empty(tuple(int(x) for x in dims[:nd]), dtype)
            if out.size and L.H5Dread(d, _native(dtype), 0, 0, 0, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"{self.path}:{name}: H5Dread failed")
            return out
        finally:
            L.H5Dclose(d)

    def write(self, name, array):
        L = lib()
        a = np.ascontiguousarray(array)
        if a.dtype not in (np.float64, np.float32, np.int64, np.int32):
            a = a.astype(np.float64 if a.dtype.kind == "f" else np.int64)
        parts = name.strip("/").split("/")
        cur = ""
        for part in parts[:-1]:
            cur += "/" + part
            if L.H5Lexists(self.id, cur.encode(), 0) <= 0:
                g = L.H5Gcreate2(self.id, cur.encode(), 0, 0, 0)
                if g < 0:
                    raise OSError(f"{self.path}: cannot create group {cur}")
                L.H5Gclose(g)
        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
        sp = L.H5Screate_simple(a.ndim, dims, None)
        d = L.H5Dcreate2(self.id, ("/" + "/".join(parts)).encode(), _native(a.dtype), sp, 0, 0, 0)
        if d < 0:
            L.H5Sclose(sp)
            raise OSError(f"{self.path}: cannot create dataset {name}")
        rc = L.H5Dwrite(d, _native(a.dtype), 0, 0, 0, a.ctypes.data_as(C.c_void_p)) if a.size else 0
        L.H5Dclose(d)
        L.H5Sclose(sp)
        if rc < 0:
            raise OSError(f"{self.path}:{name}: H5Dwrite failed")
