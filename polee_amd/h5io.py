"""The reference's on-disk contracts, read and written through the HDF5 C library (ctypes).

  likelihood-matrix HDF5   src/rnaseq_sample.jl:34-47 (reader), :505-519 (writer)
  prepared-sample HDF5     src/likelihood-approximation.jl:61-87 (writer), src/estimate.jl:380-432 (reader)
  version check            src/likelihood-approximation.jl:94-101 (PREPARED_SAMPLE_FORMAT_VERSION = 2, constants.jl:12)

h5py / HDF5.jl are not in the image; libhdf5 (1.10) is (under /opt/conda/lib).  Set POLEE_HDF5_LIB to point
elsewhere.  This module is host-side I/O only: nothing numeric happens here."""
import base64
import ctypes as C
import ctypes.util
import datetime
import os

import numpy as np

PREPARED_SAMPLE_FORMAT_VERSION = 2  # src/constants.jl:12

_lib = None
hid_t = C.c_int64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT = 0
H5S_SCALAR, H5S_ALL = 0, 0
H5T_CSET_UTF8 = 1
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_VARIABLE = C.c_size_t(-1).value


class HDF5Error(IOError):
    pass


def _candidates():
    if os.environ.get("POLEE_HDF5_LIB"):
        yield os.environ["POLEE_HDF5_LIB"]
    for p in ("/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so",
              "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"):
        yield p
    f = ctypes.util.find_library("hdf5")
    if f:
        yield f


def lib():
    global _lib
    if _lib is None:
        err = None
        for p in _candidates():
            try:
                L = C.CDLL(p)
                break
            except OSError as e:
                err = e
        else:
            raise HDF5Error("libhdf5 not found (set POLEE_HDF5_LIB): %s" % err)
        L.H5open()
        for name in ("H5Fopen", "H5Fcreate", "H5Dopen2", "H5Dcreate2", "H5Dget_space", "H5Dget_type", "H5Screate",
                     "H5Screate_simple", "H5Gopen2", "H5Gcreate2", "H5Aopen", "H5Acreate2", "H5Aget_type", "H5Aget_space",
                     "H5Pcreate", "H5Tcopy"):
            getattr(L, name).restype = hid_t
        L.H5Sget_simple_extent_npoints.restype = C.c_int64
        L.H5Tget_size.restype = C.c_size_t
        L.H5Eset_auto2(hid_t(0), None, None)  # errors surface as negative return codes, not stderr noise
        _lib = L
    return _lib


def _g(name):
    return hid_t.in_dll(lib(), name).value


def _native(dtype):
    dtype = np.dtype(dtype)
    return _g({"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int32": "H5T_NATIVE_INT32_g",
               "int64": "H5T_NATIVE_INT64_g", "uint32": "H5T_NATIVE_UINT32_g", "uint64": "H5T_NATIVE_UINT64_g"}[dtype.name])


def _chk(v, what):
    if v < 0:
        raise HDF5Error("HDF5 call failed: %s" % what)
    return v


class File:
    def __init__(self, filename, mode="r"):
        L = lib()
        fn = os.fsencode(filename)
        if mode == "r":
            self.id = _chk(L.H5Fopen(fn, H5F_ACC_RDONLY, hid_t(H5P_DEFAULT)), "open " + filename)
        else:
            self.id = _chk(L.H5Fcreate(fn, H5F_ACC_TRUNC, hid_t(H5P_DEFAULT), hid_t(H5P_DEFAULT)), "create " + filename)
        self.filename = filename

    def close(self):
        if self.id is not None:
            lib().H5Fclose(hid_t(self.id))
            self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- datasets
    def read(self, name, dtype):
        """Reads a whole dataset converted to `dtype` (like HDF5.readarray with an explicit memory type)."""
        L = lib()
        d = _chk(L.H5Dopen2(hid_t(self.id), name.encode(), hid_t(H5P_DEFAULT)), "dataset " + name)
        try:
            sp = L.H5Dget_space(hid_t(d))
            npts = L.H5Sget_simple_extent_npoints(hid_t(sp))
            L.H5Sclose(hid_t(sp))
            out = np.empty(int(npts), dtype)
            _chk(L.H5Dread(hid_t(d), hid_t(_native(dtype)), hid_t(H5S_ALL), hid_t(H5S_ALL), hid_t(H5P_DEFAULT),
                           out.ctypes.data_as(C.c_void_p)), "read " + name)
        finally:
            L.H5Dclose(hid_t(d))
        return out

    def exists(self, name):
        L = lib()
        return L.H5Lexists(hid_t(self.id), name.encode(), hid_t(H5P_DEFAULT)) > 0

    def write(self, name, value, compress=0):
        L = lib()
        a = np.ascontiguousarray(value)
        scalar = a.ndim == 0
        if scalar:
            sp = L.H5Screate(H5S_SCALAR)
        else:
            dims = (C.c_uint64 * a.ndim)(*a.shape)
            sp = L.H5Screate_simple(a.ndim, dims, None)
        pl = hid_t(H5P_DEFAULT)
        plist = None
        if compress and not scalar and a.size:
            plist = L.H5Pcreate(hid_t(_g("H5P_CLS_DATASET_CREATE_ID_g")))
            chunk = (C.c_uint64 * a.ndim)(*[min(s, 1 << 20) for s in a.shape])
            L.H5Pset_chunk(hid_t(plist), a.ndim, chunk)
            L.H5Pset_deflate(hid_t(plist), C.c_uint(int(compress)))
            pl = hid_t(plist)
        t = _native(a.dtype)
        d = _chk(L.H5Dcreate2(hid_t(self.id), name.encode(), hid_t(t), hid_t(sp), hid_t(H5P_DEFAULT), pl,
                              hid_t(H5P_DEFAULT)), "create dataset " + name)
        _chk(L.H5Dwrite(hid_t(d), hid_t(t), hid_t(H5S_ALL), hid_t(H5S_ALL), hid_t(H5P_DEFAULT),
                        a.ctypes.data_as(C.c_void_p)), "write " + name)
        L.H5Dclose(hid_t(d))
        L.H5Sclose(hid_t(sp))
        if plist is not None:
            L.H5Pclose(hid_t(plist))

    # ---- groups / attributes
    def create_group(self, name):
        L = lib()
        g = _chk(L.H5Gcreate2(hid_t(self.id), name.encode(), hid_t(H5P_DEFAULT), hid_t(H5P_DEFAULT), hid_t(H5P_DEFAULT)),
                 "create group " + name)
        L.H5Gclose(hid_t(g))

    def write_attr(self, group, name, value):
        L = lib()
        g = _chk(L.H5Gopen2(hid_t(self.id), group.encode(), hid_t(H5P_DEFAULT)), "group " + group)
        sp = L.H5Screate(H5S_SCALAR)
        if isinstance(value, (int, np.integer)):
            t, own = _native(np.int64), False
            buf = C.c_int64(int(value))
            p = C.byref(buf)
        else:
            data = str(value).encode("utf-8") + b"\0"
            t, own = L.H5Tcopy(hid_t(_g("H5T_C_S1_g"))), True
            L.H5Tset_size(hid_t(t), C.c_size_t(len(data)))
            L.H5Tset_cset(hid_t(t), H5T_CSET_UTF8)
            buf = C.create_string_buffer(data, len(data))
            p = buf
        a = _chk(L.H5Acreate2(hid_t(g), name.encode(), hid_t(t), hid_t(sp), hid_t(H5P_DEFAULT), hid_t(H5P_DEFAULT)),
                 "create attribute " + name)
        _chk(L.H5Awrite(hid_t(a), hid_t(t), p), "write attribute " + name)
        L.H5Aclose(hid_t(a))
        if own:
            L.H5Tclose(hid_t(t))
        L.H5Sclose(hid_t(sp))
        L.H5Gclose(hid_t(g))

    def read_attr(self, group, name):
        L = lib()
        g = _chk(L.H5Gopen2(hid_t(self.id), group.encode(), hid_t(H5P_DEFAULT)), "group " + group)
        try:
            if L.H5Aexists(hid_t(g), name.encode()) <= 0:
                raise KeyError(name)
            a = _chk(L.H5Aopen(hid_t(g), name.encode(), hid_t(H5P_DEFAULT)), "attribute " + name)
            t = L.H5Aget_type(hid_t(a))
            try:
                cls = L.H5Tget_class(hid_t(t))
                if cls == H5T_STRING:
                    if L.H5Tis_variable_str(hid_t(t)) > 0:
                        p = C.c_char_p()
                        _chk(L.H5Aread(hid_t(a), hid_t(t), C.byref(p)), "read attribute " + name)
                        return (p.value or b"").decode("utf-8", "replace")
                    size = L.H5Tget_size(hid_t(t))
                    buf = C.create_string_buffer(size + 1)
                    _chk(L.H5Aread(hid_t(a), hid_t(t), buf), "read attribute " + name)
                    return buf.raw[:size].split(b"\0")[0].decode("utf-8", "replace")
                if cls == H5T_INTEGER:
                    v = C.c_int64()
                    _chk(L.H5Aread(hid_t(a), hid_t(_native(np.int64)), C.byref(v)), "read attribute " + name)
                    return int(v.value)
                v = C.c_double()
                _chk(L.H5Aread(hid_t(a), hid_t(_native(np.float64)), C.byref(v)), "read attribute " + name)
                return float(v.value)
            finally:
                L.H5Tclose(hid_t(t))
                L.H5Aclose(hid_t(a))
        finally:
            L.H5Gclose(hid_t(g))


# ---- likelihood matrix (src/rnaseq_sample.jl:34-47, 505-519) ---------------------------------------------
def read_likelihood_matrix(filename):
    with File(filename) as f:
        return dict(m=int(f.read("m", np.int64)[0]), n=int(f.read("n", np.int64)[0]),
                    colptr=f.read("colptr", np.uint32), rowval=f.read("rowval", np.uint32),
                    nzval=f.read("nzval", np.float32), effective_lengths=f.read("effective_lengths", np.float32))


def write_likelihood_matrix(filename, m, n, colptr, rowval, nzval, effective_lengths, metadata=None):
    with File(filename, "w") as f:
        f.write("m", np.int64(m))
        f.write("n", np.int64(n))
        f.write("colptr", np.ascontiguousarray(colptr, np.uint32), compress=1)
        f.write("rowval", np.ascontiguousarray(rowval, np.uint32), compress=1)
        f.write("nzval", np.ascontiguousarray(nzval, np.float32), compress=1)
        f.write("effective_lengths", np.ascontiguousarray(effective_lengths, np.float32), compress=1)
        f.create_group("metadata")
        for k, v in (metadata or {}).items():
            f.write_attr("metadata", k, v)


# ---- prepared sample (src/likelihood-approximation.jl:61-101) ---------------------------------------------
def write_approximation(output_filename, m, n, efflens, params, approx_type="Polee.LogitSkewNormalPTTApprox",
                        gfffilename="", gffhash=b"", fafilename="", fahash=b"", args=""):
    """write_approximation (likelihood-approximation.jl:61-87); params: mu, omega, alpha[, node_parent_idxs, node_js]."""
    def b64(h):
        return h if isinstance(h, str) else base64.b64encode(bytes(h)).decode()
    with File(output_filename, "w") as f:
        f.write("n", np.int64(n))
        f.write("m", np.int64(m))
        f.write("effective_lengths", np.ascontiguousarray(efflens, np.float32))
        for key in ("mu", "omega", "alpha"):
            f.write(key, np.ascontiguousarray(params[key], np.float32))
        for key in ("node_parent_idxs", "node_js"):
            if key in params and params[key] is not None:
                f.write(key, np.ascontiguousarray(params[key], np.int32))
        f.create_group("metadata")
        f.write_attr("metadata", "version", PREPARED_SAMPLE_FORMAT_VERSION)
        f.write_attr("metadata", "approximation", approx_type)
        f.write_attr("metadata", "gfffilename", gfffilename)
        f.write_attr("metadata", "gffhash", b64(gffhash))
        f.write_attr("metadata", "fafilename", fafilename)
        f.write_attr("metadata", "fahash", b64(fahash))
        f.write_attr("metadata", "date", datetime.datetime.now().isoformat())
        f.write_attr("metadata", "args", args)


def check_prepared_sample_version(f, filename=""):
    """check_prepared_sample_version (likelihood-approximation.jl:94-101)."""
    try:
        v = f.read_attr("metadata", "version")
    except KeyError:
        v = None
    if v != PREPARED_SAMPLE_FORMAT_VERSION:
        older = v is None or v < PREPARED_SAMPLE_FORMAT_VERSION
        raise RuntimeError("Prepared sample %s was generated using a%s version of the software."
                           % (filename, "n older" if older else " newer"))


def read_prepared_sample(filename, check_version=True):
    with File(filename) as f:
        if check_version:
            check_prepared_sample_version(f, filename)
        out = dict(n=int(f.read("n", np.int64)[0]), m=int(f.read("m", np.int64)[0]),
                   effective_lengths=f.read("effective_lengths", np.float32), mu=f.read("mu", np.float32),
                   omega=f.read("omega", np.float32), alpha=f.read("alpha", np.float32))
        for key in ("node_parent_idxs", "node_js"):
            out[key] = f.read(key, np.int32) if f.exists(key) else None
        meta = {}
        for key in ("version", "approximation", "gfffilename", "gffhash", "fafilename", "fahash", "date", "args"):
            try:
                meta[key] = f.read_attr("metadata", key)
            except KeyError:
                pass
        out["metadata"] = meta
    return out


def read_transformation(filename):
    """PTT file of `polee fit-tree` / --ptt-tree (main.jl:650-659): node_parent_idxs, node_js."""
    with File(filename) as f:
        return f.read("node_parent_idxs", np.int32), f.read("node_js", np.int32)
