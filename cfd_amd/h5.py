"""Minimal HDF5 reader / writer over the system's libhdf5 (ctypes), for the `model.weights.h5` member of Keras 3
`.keras` archives (U/train.py:198-203, 214-216; U/valid.py:39).

h5py is not installed for this interpreter, but the image carries the HDF5 C library itself; this module binds the two
dozen calls the checkpoint interchange needs: create / open a file, nested groups, dense little-endian datasets of
float32 / float64 / int32 / int64, and a recursive listing.  Nothing else of HDF5 is used (no attributes, chunking,
compression or references -- Keras' H5IOStore writes none of them for model weights).

    available()                      -> bool (a usable libhdf5 was found)
    write(path, {"a/b/0": ndarray}, empty_groups=("vars",))
    read(path)                       -> ({"a/b/0": ndarray, ...}, ["every", "group", "path"])

Library search order: $CFD_AMD_HDF5 (a file path), ctypes.util.find_library("hdf5"), then the usual conda / distro
locations.  `available()` is False when none loads; callers then keep to the npz member and say so.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Iterable, List, Tuple

import numpy as np

hid_t, herr_t, hsize_t = C.c_int64, C.c_int, C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1

_lib = None
_tried = False
_types: Dict[str, int] = {}


class H5Error(RuntimeError):
    pass


def _candidates() -> List[str]:
    out = []
    env = os.environ.get("CFD_AMD_HDF5")
    if env:
        out.append(env)
    found = ctypes.util.find_library("hdf5")
    if found:
        out.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        out += sorted(p for p in glob.glob(pat) if "_hl" not in p and "_cpp" not in p and "fortran" not in p)
    return out


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for path in _candidates():
        try:
            lib = C.CDLL(path)
            if lib.H5open() < 0:
                continue
        except (OSError, AttributeError):
            continue
        try:
            _bind(lib)
        except AttributeError:          # a build without one of the calls below (HDF5 < 1.8): not usable here
            continue
        _lib = lib
        break
    return _lib


def _bind(lib) -> None:
    sig = {
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Screate": (hid_t, [C.c_int]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    it = getattr(lib, "H5Literate", None) or getattr(lib, "H5Literate1")
    it.restype = herr_t
    it.argtypes = [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), C.c_void_p, C.c_void_p]
    lib._iterate = it
    for key, sym in (("f32", "H5T_NATIVE_FLOAT_g"), ("f64", "H5T_NATIVE_DOUBLE_g"), ("i32", "H5T_NATIVE_INT32_g"),
                     ("i64", "H5T_NATIVE_INT64_g"), ("f32le", "H5T_IEEE_F32LE_g"), ("f64le", "H5T_IEEE_F64LE_g"),
                     ("i32le", "H5T_STD_I32LE_g"), ("i64le", "H5T_STD_I64LE_g")):
        _types[key] = hid_t.in_dll(lib, sym).value
    lib.H5Eset_auto2(0, None, None)      # failures come back as negative ids / codes; the library prints nothing


def available() -> bool:
    return _load() is not None


def _need():
    lib = _load()
    if lib is None:
        raise H5Error("no usable HDF5 C library found (set CFD_AMD_HDF5 to a libhdf5.so); looked at: "
                      + ", ".join(_candidates() or ["<nothing>"]))
    return lib


_NP2H5 = {np.dtype("float32"): ("f32", "f32le"), np.dtype("float64"): ("f64", "f64le"),
          np.dtype("int32"): ("i32", "i32le"), np.dtype("int64"): ("i64", "i64le")}


def write(path: str, datasets: Dict[str, np.ndarray], empty_groups: Iterable[str] = ()) -> None:
    """Create `path` (truncating) with one dense dataset per entry; intermediate groups are made as needed."""
    lib = _need()
    f = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise H5Error(f"cannot create {path}")
    made = set()

    def ensure(group: str) -> None:
        parts = [p for p in group.split("/") if p]
        for i in range(1, len(parts) + 1):
            g = "/".join(parts[:i])
            if g in made:
                continue
            gid = lib.H5Gcreate2(f, g.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if gid < 0:
                raise H5Error(f"cannot create group {g} in {path}")
            lib.H5Gclose(gid)
            made.add(g)

    try:
        for g in empty_groups:
            ensure(g)
        for name, arr in datasets.items():
            a = np.asarray(arr)
            if not a.flags.c_contiguous:
                a = a.copy()          # (np.ascontiguousarray would turn a 0-d value into shape (1,))
            if a.dtype not in _NP2H5:
                raise H5Error(f"{name}: dtype {a.dtype} is not one of float32/float64/int32/int64")
            mem, filet = _NP2H5[a.dtype]
            if "/" in name:
                ensure(name.rsplit("/", 1)[0])
            if a.ndim == 0:
                sp = lib.H5Screate(0)                                   # H5S_SCALAR
            else:
                dims = (hsize_t * a.ndim)(*a.shape)
                sp = lib.H5Screate_simple(a.ndim, dims, None)
            d = lib.H5Dcreate2(f, name.encode(), _types[filet], sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if sp < 0 or d < 0:
                raise H5Error(f"cannot create dataset {name} in {path}")
            rc = 0
            if a.size:
                rc = lib.H5Dwrite(d, _types[mem], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p))
            lib.H5Dclose(d)
            lib.H5Sclose(sp)
            if rc < 0:
                raise H5Error(f"cannot write dataset {name} in {path}")
    finally:
        lib.H5Fclose(f)


_ITER_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)


def _members(lib, gid: int) -> List[str]:
    names: List[str] = []

    def cb(_g, name, _info, _data):
        names.append(name.decode())
        return 0

    idx = hsize_t(0)
    if lib._iterate(gid, 0, 0, C.byref(idx), _ITER_CB(cb), None) < 0:      # H5_INDEX_NAME, H5_ITER_INC
        raise H5Error("H5Literate failed")
    return names


def _read_dataset(lib, d: int, what: str) -> np.ndarray:
    sp, tp = lib.H5Dget_space(d), lib.H5Dget_type(d)
    try:
        nd = lib.H5Sget_simple_extent_ndims(sp)
        if nd < 0:
            raise H5Error(f"{what}: not a simple dataspace")
        dims = (hsize_t * max(nd, 1))()
        if nd:
            lib.H5Sget_simple_extent_dims(sp, dims, None)
        shape = tuple(int(dims[i]) for i in range(nd))
        cls, size = lib.H5Tget_class(tp), lib.H5Tget_size(tp)
        if cls == H5T_FLOAT:
            key, dt = ("f64", np.float64) if size == 8 else ("f32", np.float32)    # (f16 converts up to f32)
        elif cls == H5T_INTEGER:
            key, dt = ("i64", np.int64) if size == 8 else ("i32", np.int32)
        else:
            raise H5Error(f"{what}: HDF5 type class {cls} is not numeric")
        out = np.empty(shape, dtype=dt)
        if out.size and lib.H5Dread(d, _types[key], H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
            raise H5Error(f"{what}: read failed")
        return out
    finally:
        lib.H5Tclose(tp)
        lib.H5Sclose(sp)


def read(path: str) -> Tuple[Dict[str, np.ndarray], List[str]]:
    """Every dataset of the file by its full path (no leading slash), and every group path."""
    lib = _need()
    f = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise H5Error(f"{path} is not a readable HDF5 file")
    data: Dict[str, np.ndarray] = {}
    groups: List[str] = []

    def walk(gid: int, prefix: str) -> None:
        for name in _members(lib, gid):
            full = prefix + name
            d = lib.H5Dopen2(gid, name.encode(), H5P_DEFAULT)
            if d >= 0:
                try:
                    data[full] = _read_dataset(lib, d, full)
                finally:
                    lib.H5Dclose(d)
                continue
            g = lib.H5Gopen2(gid, name.encode(), H5P_DEFAULT)
            if g < 0:
                continue                     # a named datatype or a dangling link: nothing Keras writes
            try:
                groups.append(full)
                walk(g, full + "/")
            finally:
                lib.H5Gclose(g)

    try:
        root = lib.H5Gopen2(f, b"/", H5P_DEFAULT)
        try:
            walk(root, "")
        finally:
            lib.H5Gclose(root)
    finally:
        lib.H5Fclose(f)
    return data, groups
