"""Minimal HDF5 reader / writer over the system's libhdf5 (ctypes) -- the container of the reference's on-disk formats.

The reference reads and writes its episode files with `h5py` (`SMNet/loader.py:199-303`, `SMNet/build_data.py:276-286`,
`SMNet/build_memory_data.py:151-153`, `custom_rcnn.py:526-530`); `h5py` is not installed in this image, but the HDF5 C library
is (`/opt/conda/lib/libhdf5.so.103`, 1.10.6), so this module binds the handful of calls those files need:

    with H5File(path) as f:              # h5py.File(path, 'r')
        "proj_indices" in f              # name in h5file
        f.keys()                         # list(h5file)
        f.shape("detection_data")        # h5file[name].shape
        f.read("proj_indices")           # np.array(h5file[name])   numeric datasets of any rank
        f.read_strings("detection_data") # [h5file[name][i] ...]    variable- or fixed-length strings -> list of bytes
    with H5File(path, "w") as f:         # h5py.File(path, 'w')
        f.write("memory_features", arr)  # f.create_dataset(name, data=arr, dtype=arr.dtype)   contiguous layout
        f.write_strings("detection_data", ["..."])   # dtype=h5py.special_dtype(vlen=str)

Chunked / gzip-compressed datasets are read transparently by the library.  No fallback: a missing library raises `H5Error`
with the search list (set `EOD_HDF5_LIB` to point at one).
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0x0000, 0x0002
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5T_SGN_NONE = 0
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1

_CANDIDATES = ["/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/libhdf5.so*",
               "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"]


class H5Error(RuntimeError):
    pass


_lib = None


def available() -> bool:
    try:
        _load()
        return True
    except H5Error:
        return False


def _load():
    global _lib
    if _lib is not None:
        return _lib
    tried: List[str] = []
    paths: List[str] = []
    if os.environ.get("EOD_HDF5_LIB"):
        paths.append(os.environ["EOD_HDF5_LIB"])
    found = ctypes.util.find_library("hdf5")
    if found:
        paths.append(found)
    for pat in _CANDIDATES:
        paths.extend(sorted(p for p in glob.glob(pat) if "_hl" not in p and "fortran" not in p and "_cpp" not in p))
    lib = None
    for p in paths:
        tried.append(p)
        try:
            lib = C.CDLL(p)
            break
        except OSError:
            continue
    if lib is None:
        raise H5Error(f"libhdf5 not found (tried {tried or _CANDIDATES}); set EOD_HDF5_LIB=/path/to/libhdf5.so")
    sig = {
        "H5open": (herr_t, []),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Gget_num_objs": (herr_t, [hid_t, C.POINTER(hsize_t)]),
        "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
        "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
        "H5Tclose": (herr_t, [hid_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.H5open() < 0:
        raise H5Error("H5open failed")
    lib.H5Eset_auto2(0, None, None)      # errors are reported through return codes, not printed stacks
    _lib = lib
    return lib


def _native(name: str) -> int:
    return hid_t.in_dll(_load(), f"H5T_{name}_g").value


_NP2H5 = {np.dtype(np.int8): "NATIVE_INT8", np.dtype(np.uint8): "NATIVE_UINT8", np.dtype(np.int16): "NATIVE_INT16",
          np.dtype(np.uint16): "NATIVE_UINT16", np.dtype(np.int32): "NATIVE_INT32", np.dtype(np.uint32): "NATIVE_UINT32",
          np.dtype(np.int64): "NATIVE_INT64", np.dtype(np.uint64): "NATIVE_UINT64", np.dtype(np.float32): "NATIVE_FLOAT",
          np.dtype(np.float64): "NATIVE_DOUBLE", np.dtype(np.bool_): "NATIVE_UINT8"}


class H5File:
    def __init__(self, path: str, mode: str = "r"):
        lib = _load()
        self._lib = lib
        self.path = path
        if mode == "r":
            self._id = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            self._id = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r' or 'w'")
        if self._id < 0:
            raise H5Error(f"cannot open {path!r} (mode {mode})")

    def close(self):
        if self._id >= 0:
            self._lib.H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inspection -------------------------------------------------------------------------------------
    def __contains__(self, name: str) -> bool:
        return self._lib.H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def keys(self) -> List[str]:
        n = hsize_t(0)
        if self._lib.H5Gget_num_objs(self._id, C.byref(n)) < 0:
            raise H5Error("H5Gget_num_objs failed")
        out = []
        for i in range(n.value):
            ln = self._lib.H5Gget_objname_by_idx(self._id, i, None, 0)
            buf = C.create_string_buffer(ln + 1)
            self._lib.H5Gget_objname_by_idx(self._id, i, buf, ln + 1)
            out.append(buf.value.decode())
        return out

    def _open(self, name: str) -> int:
        d = self._lib.H5Dopen2(self._id, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError(f"{name!r} not in {self.path}")
        return d

    def _dims(self, dset: int) -> Tuple[int, ...]:
        sp = self._lib.H5Dget_space(dset)
        nd = self._lib.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            self._lib.H5Sget_simple_extent_dims(sp, dims, None)
        self._lib.H5Sclose(sp)
        return tuple(int(dims[i]) for i in range(nd))

    def shape(self, name: str) -> Tuple[int, ...]:
        d = self._open(name)
        try:
            return self._dims(d)
        finally:
            self._lib.H5Dclose(d)

    # ---- reading ------------------------------------------------------------------------------------------
    def read(self, name: str) -> np.ndarray:
        """Numeric dataset -> ndarray of the stored width / signedness (the file's byte order is converted by the library)."""
        lib = self._lib
        d = self._open(name)
        t = lib.H5Dget_type(d)
        try:
            cls, size = lib.H5Tget_class(t), lib.H5Tget_size(t)
            if cls == H5T_INTEGER or cls == H5T_ENUM:          # h5py stores np.bool_ as an 8-bit enum
                unsigned = cls == H5T_INTEGER and lib.H5Tget_sign(t) == H5T_SGN_NONE
                dt = np.dtype(f"{'u' if unsigned else 'i'}{size}")
            elif cls == H5T_FLOAT:
                dt = np.dtype(f"f{size}")
            else:
                raise H5Error(f"{name!r}: type class {cls} is not numeric (use read_strings for strings)")
            mem = _native(_NP2H5[dt])
            out = np.empty(self._dims(d), dtype=dt)
            if out.size and lib.H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise H5Error(f"H5Dread({name!r}) failed")
            return out
        finally:
            lib.H5Tclose(t)
            lib.H5Dclose(d)

    def read_strings(self, name: str) -> List[bytes]:
        """1-D string dataset (variable length as written by `h5py.special_dtype(vlen=str)`, or fixed length) -> list of bytes,
        like `h5file[name][i]` in h5py >= 3."""
        lib = self._lib
        d = self._open(name)
        t = lib.H5Dget_type(d)
        try:
            if lib.H5Tget_class(t) != H5T_STRING:
                raise H5Error(f"{name!r} is not a string dataset")
            dims = self._dims(d)
            n = int(np.prod(dims)) if dims else 1
            if lib.H5Tis_variable_str(t) > 0:
                mem = lib.H5Tcopy(_native("C_S1"))
                lib.H5Tset_size(mem, H5T_VARIABLE)
                lib.H5Tset_cset(mem, H5T_CSET_UTF8)
                ptrs = (C.c_char_p * n)()
                sp = lib.H5Dget_space(d)
                try:
                    if n and lib.H5Dread(d, mem, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs) < 0:
                        raise H5Error(f"H5Dread({name!r}) failed")
                    out = [bytes(p) if p is not None else b"" for p in ptrs]
                    lib.H5Dvlen_reclaim(mem, sp, H5P_DEFAULT, ptrs)
                finally:
                    lib.H5Sclose(sp)
                    lib.H5Tclose(mem)
                return out
            size = lib.H5Tget_size(t)
            buf = C.create_string_buffer(n * size)
            if n and lib.H5Dread(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0:
                raise H5Error(f"H5Dread({name!r}) failed")
            raw = buf.raw
            return [raw[i * size:(i + 1) * size].rstrip(b"\x00") for i in range(n)]
        finally:
            lib.H5Tclose(t)
            lib.H5Dclose(d)

    # ---- writing ------------------------------------------------------------------------------------------
    def write(self, name: str, data, dtype=None) -> None:
        """`create_dataset(name, data=data, dtype=dtype)`: contiguous layout, native byte order."""
        lib = self._lib
        arr = np.ascontiguousarray(np.asarray(data) if dtype is None else np.asarray(data).astype(dtype))
        if arr.dtype not in _NP2H5:
            raise H5Error(f"{name!r}: dtype {arr.dtype} not supported")
        stored = arr.view(np.uint8) if arr.dtype == np.bool_ else arr
        t = _native(_NP2H5[arr.dtype])
        dims = (hsize_t * max(arr.ndim, 1))(*arr.shape)
        sp = lib.H5Screate_simple(arr.ndim, dims, None)
        d = lib.H5Dcreate2(self._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        try:
            if d < 0:
                raise H5Error(f"H5Dcreate2({name!r}) failed")
            if arr.size and lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, stored.ctypes.data_as(C.c_void_p)) < 0:
                raise H5Error(f"H5Dwrite({name!r}) failed")
        finally:
            if d >= 0:
                lib.H5Dclose(d)
            lib.H5Sclose(sp)

    def write_strings(self, name: str, strings: Sequence[str]) -> None:
        """`create_dataset(name, data=list_of_str, dtype=h5py.special_dtype(vlen=str))`."""
        lib = self._lib
        enc = [s.encode() if isinstance(s, str) else bytes(s) for s in strings]
        t = lib.H5Tcopy(_native("C_S1"))
        lib.H5Tset_size(t, H5T_VARIABLE)
        lib.H5Tset_cset(t, H5T_CSET_UTF8)
        dims = (hsize_t * 1)(len(enc))
        sp = lib.H5Screate_simple(1, dims, None)
        d = lib.H5Dcreate2(self._id, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        try:
            if d < 0:
                raise H5Error(f"H5Dcreate2({name!r}) failed")
            ptrs = (C.c_char_p * len(enc))(*enc)
            if enc and lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs) < 0:
                raise H5Error(f"H5Dwrite({name!r}) failed")
        finally:
            if d >= 0:
                lib.H5Dclose(d)
            lib.H5Sclose(sp)
            lib.H5Tclose(t)
