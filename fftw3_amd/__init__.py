"""fftw3_amd -- Python host mirror of the FFTW3 C API over libfftw3_amd.so.

The product is the C-ABI shared library built from ``fftw3_amd/csrc`` (HIP
kernels for gfx950 + C planner).  This module is a thin ctypes binding whose
function names and argument order mirror the reference API
(reference ``fftw/fftw3.h:144-463``), so that tests read like FFTW client code:

    p = fftw3_amd.plan_many_dft(1, [n], howmany, x, None, 1, n, y, None, 1, n,
                                fftw3_amd.FORWARD, fftw3_amd.ESTIMATE)
    p.execute()

Arrays may be numpy arrays (host memory: staged through PCIe by the library),
torch tensors (device memory: transformed in place in HBM) or raw integer
addresses.  There is no Python or CPU fallback: if the shared library is
missing the import fails, and executing without a HIP device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libfftw3_amd.so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        "fftw3_amd: %s not found. Build it with `make` (or __graft_entry__.build()); "
        "there is no fallback implementation." % _LIB_PATH)

lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)

FORWARD = -1
BACKWARD = +1
MEASURE = 0
DESTROY_INPUT = 1 << 0
UNALIGNED = 1 << 1
CONSERVE_MEMORY = 1 << 2
EXHAUSTIVE = 1 << 3
PRESERVE_INPUT = 1 << 4
PATIENT = 1 << 5
ESTIMATE = 1 << 6
WISDOM_ONLY = 1 << 21

MAX_DIMS = 8
MAX_RADICES = 16

STEP_PASS, STEP_COPY, STEP_R2C_POST, STEP_C2R_PRE, STEP_RADER_MUL, STEP_HERM_EXPAND = 1, 2, 3, 4, 5, 6
STEP_R2C_POST4, STEP_C2R_PRE4 = 7, 8
STEP_R2R = 9
# fftw_r2r_kind (include/fftw3.h)
R2HC, HC2R, DHT, REDFT00, REDFT01, REDFT10, REDFT11, RODFT00, RODFT01, RODFT10, RODFT11 = range(11)
# FFTW_AMD_R2R_* step modes (include/fftw3_amd.h)
(R2R_PRE_HC2R, R2R_PRE_E10, R2R_PRE_O10, R2R_PRE_E01, R2R_PRE_O01, R2R_PRE_E00, R2R_PRE_O00,
 R2R_PRE_E11, R2R_PRE_O11, R2R_PRE_E11ODD, R2R_PRE_O11ODD, R2R_POST_R2HC, R2R_POST_DHT,
 R2R_POST_E10, R2R_POST_O10, R2R_POST_E01, R2R_POST_O01, R2R_POST_E00, R2R_POST_O00,
 R2R_POST_E11, R2R_POST_O11, R2R_POST_E11ODD, R2R_POST_O11ODD) = range(1, 24)
F_SWAP_IN, F_SWAP_OUT, F_REAL_IN, F_REAL_OUT = 1, 2, 4, 8
F_MUL_TABLE, F_MUL_CONJ, F_PERM_SRC, F_PERM_DST, F_CONJ_OUT, F_TW_IN = 16, 32, 64, 128, 256, 512
F_R2C_ROWS = 1024
F_C2R_ROWS = 2048
F_LO_DFT = 1 << 14
F_REAL_DEC = 1 << 15


class StepDesc(C.Structure):
    """Mirror of fftw_amd_step_desc (include/fftw3_amd.h)."""
    _fields_ = [
        ("kind", C.c_int), ("src_buf", C.c_int), ("dst_buf", C.c_int),
        ("src_base", C.c_longlong), ("dst_base", C.c_longlong),
        ("src_im", C.c_longlong), ("dst_im", C.c_longlong),
        ("flags", C.c_int), ("L", C.c_int),
        ("nradices", C.c_int), ("radices", C.c_int * MAX_RADICES),
        ("is_l", C.c_longlong), ("os_l", C.c_longlong),
        ("ndims", C.c_int),
        ("dim_n", C.c_longlong * MAX_DIMS), ("dim_is", C.c_longlong * MAX_DIMS),
        ("dim_os", C.c_longlong * MAX_DIMS), ("dim_tw", C.c_longlong * MAX_DIMS),
        ("tw_n", C.c_longlong),
        ("tw_shift", C.c_int), ("tw_lo", C.c_int), ("tw_hi", C.c_int),
        ("tile", C.c_int), ("batch_dim", C.c_int),
        ("aux_n", C.c_longlong), ("aux_valid", C.c_longlong),
        ("table", C.c_int), ("table2", C.c_int),
        ("aux_buf", C.c_int), ("aux_base", C.c_longlong),
        ("variant", C.c_int),
        ("tile_lo_n", C.c_int), ("tile_lo_is", C.c_longlong), ("tile_lo_os", C.c_longlong),
        ("kpos", C.c_int),
    ]


class iodim64(C.Structure):
    _fields_ = [("n", C.c_ssize_t), ("is_", C.c_ssize_t), ("os", C.c_ssize_t)]


_vp = C.c_void_p
_ip = C.POINTER(C.c_int)


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("fftw_execute", None, _vp)
_sig("fftw_execute_dft", None, _vp, _vp, _vp)
_sig("fftw_execute_split_dft", None, _vp, _vp, _vp, _vp, _vp)
_sig("fftw_execute_dft_r2c", None, _vp, _vp, _vp)
_sig("fftw_execute_dft_c2r", None, _vp, _vp, _vp)
_sig("fftw_plan_dft", _vp, C.c_int, _ip, _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_dft_1d", _vp, C.c_int, _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_dft_2d", _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_dft_3d", _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_many_dft", _vp, C.c_int, _ip, C.c_int, _vp, _ip, C.c_int, C.c_int,
     _vp, _ip, C.c_int, C.c_int, C.c_int, C.c_uint)
_sig("fftw_plan_many_dft_r2c", _vp, C.c_int, _ip, C.c_int, _vp, _ip, C.c_int, C.c_int,
     _vp, _ip, C.c_int, C.c_int, C.c_uint)
_sig("fftw_plan_many_dft_c2r", _vp, C.c_int, _ip, C.c_int, _vp, _ip, C.c_int, C.c_int,
     _vp, _ip, C.c_int, C.c_int, C.c_uint)
_sig("fftw_plan_dft_r2c", _vp, C.c_int, _ip, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_c2r", _vp, C.c_int, _ip, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_r2c_1d", _vp, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_c2r_1d", _vp, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_r2c_2d", _vp, C.c_int, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_c2r_2d", _vp, C.c_int, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_r2c_3d", _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_dft_c2r_3d", _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_uint)
_sig("fftw_plan_guru64_dft", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_guru64_split_dft", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, _vp, _vp, C.c_uint)
_sig("fftw_plan_guru64_dft_r2c", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, C.c_uint)
_sig("fftw_plan_guru64_dft_c2r", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, C.c_uint)
_sig("fftw_plan_guru64_split_dft_r2c", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, _vp, C.c_uint)
_sig("fftw_plan_guru64_split_dft_c2r", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, _vp, C.c_uint)
_sig("fftw_execute_split_dft_r2c", None, _vp, _vp, _vp, _vp)
_sig("fftw_execute_split_dft_c2r", None, _vp, _vp, _vp, _vp)
_vpp = C.POINTER(C.c_void_p)
_sig("fftw_amd_shard_range", None, C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
_sig("fftw_amd_plan_many_dft_sharded", _vp, C.c_int, _ip, C.c_int, C.c_int, _ip, _vpp, _ip, C.c_int, C.c_int,
     _vpp, _ip, C.c_int, C.c_int, C.c_int, C.c_uint)
_sig("fftw_amd_plan_many_dft_r2c_sharded", _vp, C.c_int, _ip, C.c_int, C.c_int, _ip, _vpp, _ip, C.c_int, C.c_int,
     _vpp, _ip, C.c_int, C.c_int, C.c_uint)
_sig("fftw_amd_plan_many_dft_c2r_sharded", _vp, C.c_int, _ip, C.c_int, C.c_int, _ip, _vpp, _ip, C.c_int, C.c_int,
     _vpp, _ip, C.c_int, C.c_int, C.c_uint)
_sig("fftw_amd_execute_sharded", None, _vp)
_sig("fftw_amd_sharded_sync", None, _vp)
_sig("fftw_amd_sharded_all_gather", C.c_int, _vp, _vpp, C.c_int)
_sig("fftw_amd_slab_local_size", C.c_longlong, C.c_int, C.POINTER(C.c_longlong), C.c_int, C.c_int,
     C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
_sig("fftw_amd_slab_plan_dft", _vp, C.c_int, C.POINTER(C.c_longlong), C.c_int, C.POINTER(C.c_int), _vpp, _vpp, C.c_int, C.c_uint)
_sig("fftw_amd_slab_execute", None, _vp)
_sig("fftw_amd_slab_sync", None, _vp)
_sig("fftw_amd_slab_num_devices", C.c_int, _vp)
_sig("fftw_amd_slab_local_plan", _vp, _vp, C.c_int, C.c_int)
_sig("fftw_amd_destroy_slab_plan", None, _vp)
_sig("fftw_amd_sharded_gather_ops", C.c_int, _vp, _vpp, C.POINTER(C.c_longlong), C.c_int)
_sig("fftw_amd_rccl_probe", C.c_int)
_sig("fftw_amd_plan_workspace_device", C.c_int, _vp)
_sig("fftw_amd_sharded_num_shards", C.c_int, _vp)
_sig("fftw_amd_sharded_device", C.c_int, _vp, C.c_int)
_sig("fftw_amd_sharded_range", None, _vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong))
_sig("fftw_amd_sharded_replica", _vp, _vp, C.c_int)
_sig("fftw_amd_destroy_sharded_plan", None, _vp)
_sig("fftw_execute_r2r", None, _vp, _vp, _vp)
_sig("fftw_plan_r2r", _vp, C.c_int, _ip, _vp, _vp, _ip, C.c_uint)
_sig("fftw_plan_r2r_1d", _vp, C.c_int, _vp, _vp, C.c_int, C.c_uint)
_sig("fftw_plan_r2r_2d", _vp, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_uint)
_sig("fftw_plan_r2r_3d", _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_uint)
_sig("fftw_plan_many_r2r", _vp, C.c_int, _ip, C.c_int, _vp, _ip, C.c_int, C.c_int,
     _vp, _ip, C.c_int, C.c_int, _ip, C.c_uint)
_sig("fftw_plan_guru64_r2r", _vp, C.c_int, C.POINTER(iodim64), C.c_int, C.POINTER(iodim64),
     _vp, _vp, _ip, C.c_uint)
_sig("fftw_destroy_plan", None, _vp)
_sig("fftw_cleanup", None)
_sig("fftw_forget_wisdom", None)
_sig("fftw_export_wisdom_to_string", _vp)
_sig("fftw_import_wisdom_from_string", C.c_int, C.c_char_p)
_sig("fftw_export_wisdom_to_filename", C.c_int, C.c_char_p)
_sig("fftw_import_wisdom_from_filename", C.c_int, C.c_char_p)
_sig("fftw_malloc", _vp, C.c_size_t)
_sig("fftw_free", None, _vp)
_sig("fftw_sprint_plan", _vp, _vp)
_sig("fftw_flops", None, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
_sig("fftw_amd_device_count", C.c_int)
_sig("fftw_amd_malloc_device", _vp, C.c_size_t)
_sig("fftw_amd_free_device", None, _vp)
_sig("fftw_amd_plan_set_stream", None, _vp, _vp)
_sig("fftw_amd_plan_sync", None, _vp)
_sig("fftw_amd_plan_workspace_bytes", C.c_size_t, _vp)
_sig("fftw_amd_set_chunk_bytes", None, C.c_size_t)
_sig("fftw_amd_plan_paired", C.c_int, _vp)
_sig("fftw_amd_plan_lanes", C.c_int, _vp)
_sig("fftw_amd_set_device", C.c_int, C.c_int)
_sig("fftw_amd_get_device", C.c_int)
_sig("fftw_amd_plan_num_steps", C.c_int, _vp)
_sig("fftw_amd_plan_get_step", C.c_int, _vp, C.c_int, C.POINTER(StepDesc))
_sig("fftw_amd_plan_chunk", C.c_longlong, _vp)
_sig("fftw_amd_plan_batch", C.c_longlong, _vp)
_sig("fftw_amd_plan_table", C.c_longlong, _vp, C.c_int, C.POINTER(C.c_double), C.c_longlong)
_sig("fftw_amd_execute_profiled", C.c_int, _vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int)
_sig("fftw_amd_cexp", None, C.c_longlong, C.c_longlong, C.POINTER(C.c_double))
_sig("fftw_amd_find_generator", C.c_longlong, C.c_longlong)
_sig("fftw_amd_power_mod", C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong)
_sig("fftw_amd_factor_passes", C.c_int, C.c_longlong, C.c_int, C.POINTER(C.c_longlong))
_libc_free = C.CDLL(None).free
_libc_free.argtypes = [_vp]


def device_count():
    return lib.fftw_amd_device_count()


def ptr(x):
    """Address of an array-like: numpy array, torch tensor, int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if hasattr(x, "ctypes"):
        return x.ctypes.data
    raise TypeError("cannot take the address of %r" % type(x))


def _ints(v):
    if v is None:
        return None
    v = list(v)
    return (C.c_int * len(v))(*v)


def _current_stream_of(*arrays):
    for a in arrays:
        if hasattr(a, "data_ptr") and getattr(a, "is_cuda", False):
            import torch
            return torch.cuda.current_stream(a.device).cuda_stream
    return None


class Plan(object):
    """Owns one fftw_plan.  Keeps the arrays alive that it was planned on."""

    def __init__(self, handle, keep=()):
        if not handle:
            raise ValueError("FFTW planner returned NULL (invalid or unsupported problem)")
        self.handle = handle
        self._keep = keep
        s = _current_stream_of(*keep)
        if s:
            lib.fftw_amd_plan_set_stream(self.handle, s)

    def _need_device(self):
        if device_count() <= 0:
            raise RuntimeError("fftw3_amd: no HIP device; the executor has no CPU fallback")

    def set_stream(self, hip_stream):
        lib.fftw_amd_plan_set_stream(self.handle, hip_stream)

    def execute(self):
        self._need_device()
        lib.fftw_execute(self.handle)

    def execute_dft(self, i, o):
        self._need_device()
        lib.fftw_execute_dft(self.handle, ptr(i), ptr(o))

    def execute_split_dft(self, ri, ii, ro, io):
        self._need_device()
        lib.fftw_execute_split_dft(self.handle, ptr(ri), ptr(ii), ptr(ro), ptr(io))

    def execute_split_dft_r2c(self, i, ro, io):
        self._need_device()
        lib.fftw_execute_split_dft_r2c(self.handle, ptr(i), ptr(ro), ptr(io))

    def execute_split_dft_c2r(self, ri, ii, o):
        self._need_device()
        lib.fftw_execute_split_dft_c2r(self.handle, ptr(ri), ptr(ii), ptr(o))

    def execute_dft_r2c(self, i, o):
        self._need_device()
        lib.fftw_execute_dft_r2c(self.handle, ptr(i), ptr(o))

    def execute_dft_c2r(self, i, o):
        self._need_device()
        lib.fftw_execute_dft_c2r(self.handle, ptr(i), ptr(o))

    def execute_r2r(self, i, o):
        self._need_device()
        lib.fftw_execute_r2r(self.handle, ptr(i), ptr(o))

    def sync(self):
        lib.fftw_amd_plan_sync(self.handle)

    def execute_profiled(self):
        """one execution with HIP events around every launch:
        [(step, total_ms, launches), ...]"""
        self._need_device()
        st = self.steps()
        ms = (C.c_double * max(1, len(st)))()
        cnt = (C.c_longlong * max(1, len(st)))()
        k = lib.fftw_amd_execute_profiled(self.handle, ms, cnt, len(st))
        return [(st[i], ms[i], cnt[i]) for i in range(max(0, k))]

    def sprint(self):
        p = lib.fftw_sprint_plan(self.handle)
        s = C.string_at(p).decode()
        _libc_free(p)
        return s

    def flops(self):
        a, m, f = C.c_double(), C.c_double(), C.c_double()
        lib.fftw_flops(self.handle, C.byref(a), C.byref(m), C.byref(f))
        return a.value, m.value, f.value

    @property
    def batch(self):
        return lib.fftw_amd_plan_batch(self.handle)

    @property
    def chunk(self):
        return lib.fftw_amd_plan_chunk(self.handle)

    @property
    def paired(self):
        return bool(lib.fftw_amd_plan_paired(self.handle))

    @property
    def lanes(self):
        """chunk lanes of the plan: chunk c runs on stream c % lanes (1: serial chunks); known after the first execution"""
        return lib.fftw_amd_plan_lanes(self.handle)

    @property
    def workspace_bytes(self):
        return lib.fftw_amd_plan_workspace_bytes(self.handle)

    def steps(self):
        out = []
        for i in range(lib.fftw_amd_plan_num_steps(self.handle)):
            d = StepDesc()
            lib.fftw_amd_plan_get_step(self.handle, i, C.byref(d))
            out.append(d)
        return out

    def table(self, tid):
        """(kind-agnostic) host copy of table `tid` as a float64 numpy array, or
        ('dft_of', src_id) when the device has yet to compute it."""
        import numpy as np
        probe = (C.c_double * 1)()
        n = lib.fftw_amd_plan_table(self.handle, tid, probe, 1)
        if n == 0:
            return ("dft_of", int(probe[0]))
        buf = np.empty(n, dtype=np.float64)
        lib.fftw_amd_plan_table(self.handle, tid, buf.ctypes.data_as(C.POINTER(C.c_double)), n)
        return buf

    def destroy(self):
        if self.handle:
            lib.fftw_destroy_plan(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ---- planners, same names and argument order as the C API -----------------

def plan_dft(rank, n, i, o, sign, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft(rank, _ints(n), ptr(i), ptr(o), sign, flags), (i, o))


def plan_dft_1d(n, i, o, sign, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_1d(n, ptr(i), ptr(o), sign, flags), (i, o))


def plan_dft_2d(n0, n1, i, o, sign, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_2d(n0, n1, ptr(i), ptr(o), sign, flags), (i, o))


def plan_dft_3d(n0, n1, n2, i, o, sign, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_3d(n0, n1, n2, ptr(i), ptr(o), sign, flags), (i, o))


def plan_many_dft(rank, n, howmany, i, inembed, istride, idist, o, onembed, ostride, odist,
                  sign, flags=ESTIMATE):
    return Plan(lib.fftw_plan_many_dft(rank, _ints(n), howmany, ptr(i), _ints(inembed), istride,
                                       idist, ptr(o), _ints(onembed), ostride, odist, sign, flags),
                (i, o))


def plan_many_dft_r2c(rank, n, howmany, i, inembed, istride, idist, o, onembed, ostride, odist,
                      flags=ESTIMATE):
    return Plan(lib.fftw_plan_many_dft_r2c(rank, _ints(n), howmany, ptr(i), _ints(inembed),
                                           istride, idist, ptr(o), _ints(onembed), ostride, odist,
                                           flags), (i, o))


def plan_many_dft_c2r(rank, n, howmany, i, inembed, istride, idist, o, onembed, ostride, odist,
                      flags=ESTIMATE):
    return Plan(lib.fftw_plan_many_dft_c2r(rank, _ints(n), howmany, ptr(i), _ints(inembed),
                                           istride, idist, ptr(o), _ints(onembed), ostride, odist,
                                           flags), (i, o))


def plan_dft_r2c(rank, n, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_r2c(rank, _ints(n), ptr(i), ptr(o), flags), (i, o))


def plan_dft_c2r(rank, n, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_c2r(rank, _ints(n), ptr(i), ptr(o), flags), (i, o))


def plan_dft_r2c_1d(n, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_r2c_1d(n, ptr(i), ptr(o), flags), (i, o))


def plan_dft_c2r_1d(n, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_c2r_1d(n, ptr(i), ptr(o), flags), (i, o))


def plan_dft_r2c_2d(n0, n1, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_r2c_2d(n0, n1, ptr(i), ptr(o), flags), (i, o))


def plan_dft_c2r_2d(n0, n1, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_c2r_2d(n0, n1, ptr(i), ptr(o), flags), (i, o))


def plan_dft_r2c_3d(n0, n1, n2, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_r2c_3d(n0, n1, n2, ptr(i), ptr(o), flags), (i, o))


def plan_dft_c2r_3d(n0, n1, n2, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_dft_c2r_3d(n0, n1, n2, ptr(i), ptr(o), flags), (i, o))


def plan_r2r(rank, n, i, o, kind, flags=ESTIMATE):
    return Plan(lib.fftw_plan_r2r(rank, _ints(n), ptr(i), ptr(o), _ints(kind), flags), (i, o))


def plan_r2r_1d(n, i, o, kind, flags=ESTIMATE):
    return Plan(lib.fftw_plan_r2r_1d(n, ptr(i), ptr(o), kind, flags), (i, o))


def plan_r2r_2d(n0, n1, i, o, kind0, kind1, flags=ESTIMATE):
    return Plan(lib.fftw_plan_r2r_2d(n0, n1, ptr(i), ptr(o), kind0, kind1, flags), (i, o))


def plan_r2r_3d(n0, n1, n2, i, o, kind0, kind1, kind2, flags=ESTIMATE):
    return Plan(lib.fftw_plan_r2r_3d(n0, n1, n2, ptr(i), ptr(o), kind0, kind1, kind2, flags), (i, o))


def plan_many_r2r(rank, n, howmany, i, inembed, istride, idist, o, onembed, ostride, odist,
                  kind, flags=ESTIMATE):
    return Plan(lib.fftw_plan_many_r2r(rank, _ints(n), howmany, ptr(i), _ints(inembed), istride,
                                       idist, ptr(o), _ints(onembed), ostride, odist, _ints(kind),
                                       flags), (i, o))


def plan_guru64_r2r(dims, howmany_dims, i, o, kind, flags=ESTIMATE):
    """dims / howmany_dims: sequences of (n, is, os) in doubles; kind: one fftw_r2r_kind per dim."""
    return Plan(lib.fftw_plan_guru64_r2r(len(dims), _iodims(dims), len(howmany_dims),
                                         _iodims(howmany_dims), ptr(i), ptr(o), _ints(kind), flags),
                (i, o))


def _iodims(dims):
    arr = (iodim64 * max(1, len(dims)))()
    for k, (n, is_, os_) in enumerate(dims):
        arr[k].n, arr[k].is_, arr[k].os = n, is_, os_
    return arr


def plan_guru64_dft(dims, howmany_dims, i, o, sign, flags=ESTIMATE):
    """dims / howmany_dims: sequences of (n, is, os) in complex elements."""
    return Plan(lib.fftw_plan_guru64_dft(len(dims), _iodims(dims), len(howmany_dims),
                                         _iodims(howmany_dims), ptr(i), ptr(o), sign, flags), (i, o))


def plan_guru64_split_dft(dims, howmany_dims, ri, ii, ro, io, flags=ESTIMATE):
    return Plan(lib.fftw_plan_guru64_split_dft(len(dims), _iodims(dims), len(howmany_dims),
                                               _iodims(howmany_dims), ptr(ri), ptr(ii), ptr(ro),
                                               ptr(io), flags), (ri, ii, ro, io))


def plan_guru64_split_dft_r2c(dims, howmany_dims, i, ro, io, flags=ESTIMATE):
    """dims strides: `is` in reals, `os` in reals of the split output planes"""
    return Plan(lib.fftw_plan_guru64_split_dft_r2c(len(dims), _iodims(dims), len(howmany_dims),
                                                   _iodims(howmany_dims), ptr(i), ptr(ro), ptr(io),
                                                   flags), (i, ro, io))


def plan_guru64_split_dft_c2r(dims, howmany_dims, ri, ii, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_guru64_split_dft_c2r(len(dims), _iodims(dims), len(howmany_dims),
                                                   _iodims(howmany_dims), ptr(ri), ptr(ii), ptr(o),
                                                   flags), (ri, ii, o))


def plan_guru64_dft_r2c(dims, howmany_dims, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_guru64_dft_r2c(len(dims), _iodims(dims), len(howmany_dims),
                                             _iodims(howmany_dims), ptr(i), ptr(o), flags), (i, o))


def plan_guru64_dft_c2r(dims, howmany_dims, i, o, flags=ESTIMATE):
    return Plan(lib.fftw_plan_guru64_dft_c2r(len(dims), _iodims(dims), len(howmany_dims),
                                             _iodims(howmany_dims), ptr(i), ptr(o), flags), (i, o))


def shard_range(howmany, nshards, g):
    """block rule of the C layer (fftw_amd_shard_range): [g*ceil(B/P), min(B, (g+1)*ceil(B/P)))"""
    lo, hi = C.c_longlong(0), C.c_longlong(0)
    lib.fftw_amd_shard_range(howmany, nshards, g, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def _ptrs(arrays):
    return (C.c_void_p * len(arrays))(*[ptr(a) for a in arrays])


class ShardedPlan(object):
    """fftw_amd_plan_many_dft*_sharded: one batched transform cut over several GPUs inside the C library
    (one plan replica, stream and host thread per device)."""

    def __init__(self, handle, keep):
        if not handle:
            raise ValueError("sharded planner returned NULL (invalid or unsupported problem)")
        self.handle = handle
        self._keep = keep

    def execute(self):
        if device_count() <= 0:
            raise RuntimeError("no HIP device: the sharded plan cannot execute (no CPU fallback)")
        lib.fftw_amd_execute_sharded(self.handle)

    def sync(self):
        lib.fftw_amd_sharded_sync(self.handle)

    def all_gather(self, full, mode=0):
        """full[d]: whole-batch buffer on shard d's device.  Returns 1 (RCCL), 0 (peer-to-peer)."""
        self._keep_full = full
        rc = lib.fftw_amd_sharded_all_gather(self.handle, _ptrs(full), mode)
        if rc < 0:
            raise RuntimeError("all-gather of the output shards failed")
        return rc

    def gather_ops(self, full):
        """the RCCL calls all_gather(full) would issue: list of (kind, rank, root, send, recv, bytes);
        kind 0 = ncclAllGather, 1 = ncclBroadcast.  Pointers may be plain integers (no call is made)."""
        n = self.num_shards
        cap = n * n + n
        ops = (C.c_longlong * (6 * cap))()
        arr = (C.c_void_p * n)(*[a if isinstance(a, int) else ptr(a) for a in full])
        k = lib.fftw_amd_sharded_gather_ops(self.handle, arr, ops, cap)
        if k < 0:
            raise ValueError("bad arguments")
        return [tuple(ops[6 * i + j] for j in range(6)) for i in range(k)]

    def replica_device(self, g):
        """device that holds replica g's tables and scratch (-1: none / empty shard)"""
        h = lib.fftw_amd_sharded_replica(self.handle, g)
        return lib.fftw_amd_plan_workspace_device(h) if h else -1

    @property
    def num_shards(self):
        return lib.fftw_amd_sharded_num_shards(self.handle)

    def device(self, g):
        return lib.fftw_amd_sharded_device(self.handle, g)

    def range(self, g):
        lo, hi = C.c_longlong(0), C.c_longlong(0)
        lib.fftw_amd_sharded_range(self.handle, g, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def replica_sprint(self, g):
        h = lib.fftw_amd_sharded_replica(self.handle, g)
        if not h:
            return None
        s = lib.fftw_sprint_plan(h)
        try:
            return C.cast(s, C.c_char_p).value.decode()
        finally:
            lib.fftw_free(s)

    def destroy(self):
        if self.handle:
            lib.fftw_amd_destroy_sharded_plan(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def plan_many_dft_sharded(rank, n, howmany, devs, ins, inembed, istride, idist, outs, onembed, ostride, odist,
                          sign, flags=ESTIMATE):
    return ShardedPlan(lib.fftw_amd_plan_many_dft_sharded(rank, _ints(n), howmany, len(ins), _ints(devs), _ptrs(ins),
                                                          _ints(inembed), istride, idist, _ptrs(outs), _ints(onembed),
                                                          ostride, odist, sign, flags), (ins, outs))


def plan_many_dft_r2c_sharded(rank, n, howmany, devs, ins, inembed, istride, idist, outs, onembed, ostride, odist,
                              flags=ESTIMATE):
    return ShardedPlan(lib.fftw_amd_plan_many_dft_r2c_sharded(rank, _ints(n), howmany, len(ins), _ints(devs), _ptrs(ins),
                                                              _ints(inembed), istride, idist, _ptrs(outs),
                                                              _ints(onembed), ostride, odist, flags), (ins, outs))


def plan_many_dft_c2r_sharded(rank, n, howmany, devs, ins, inembed, istride, idist, outs, onembed, ostride, odist,
                              flags=ESTIMATE):
    return ShardedPlan(lib.fftw_amd_plan_many_dft_c2r_sharded(rank, _ints(n), howmany, len(ins), _ints(devs), _ptrs(ins),
                                                              _ints(inembed), istride, idist, _ptrs(outs),
                                                              _ints(onembed), ostride, odist, flags), (ins, outs))


def cexp(m, n):
    out = (C.c_double * 2)()
    lib.fftw_amd_cexp(m, n, out)
    return out[0], out[1]


def factor_passes(n, max_passes=4):
    lens = (C.c_longlong * 8)()
    k = lib.fftw_amd_factor_passes(n, max_passes, lens)
    return [lens[j] for j in range(k)]


def set_chunk_bytes(nbytes):
    lib.fftw_amd_set_chunk_bytes(nbytes)


def export_wisdom_to_string():
    p = lib.fftw_export_wisdom_to_string()
    s = C.string_at(p).decode()
    _libc_free(p)
    return s


def import_wisdom_from_string(s):
    return lib.fftw_import_wisdom_from_string(s.encode())


def forget_wisdom():
    lib.fftw_forget_wisdom()


# ---- one transform in slabs over the GPUs of this process (fftw3_amd/csrc/slab.c; the multi-process form is slab.py)

def slab_local_size(n, ndev, g):
    """fftw_amd_slab_local_size: (elements, local_n0, local_0_start) of device g"""
    nn = (C.c_longlong * len(n))(*n)
    ln0, lo = C.c_longlong(0), C.c_longlong(0)
    tot = lib.fftw_amd_slab_local_size(len(n), nn, ndev, g, C.byref(ln0), C.byref(lo))
    return tot, ln0.value, lo.value


class SlabPlanC(object):
    """fftw_amd_slab_plan_dft: a 2-D / 3-D complex transform whose first dimension is cut over several devices"""

    def __init__(self, n, devs, ins, outs, sign, flags=ESTIMATE):
        nn = (C.c_longlong * len(n))(*n)
        dv = (C.c_int * len(devs))(*devs)
        self.handle = lib.fftw_amd_slab_plan_dft(len(n), nn, len(devs), dv, _ptrs(ins), _ptrs(outs), sign, flags)
        if not self.handle:
            raise ValueError("slab planner returned NULL (invalid or unsupported problem)")
        self._keep = (ins, outs)

    def execute(self):
        if device_count() <= 0:
            raise RuntimeError("no HIP device: the slab plan cannot execute (no CPU fallback)")
        lib.fftw_amd_slab_execute(self.handle)

    def sync(self):
        lib.fftw_amd_slab_sync(self.handle)

    def local_plan_sprint(self, g, which):
        h = lib.fftw_amd_slab_local_plan(self.handle, g, which)
        if not h:
            return None
        s = lib.fftw_sprint_plan(h)
        try:
            return C.cast(s, C.c_char_p).value.decode()
        finally:
            lib.fftw_free(s)

    def destroy(self):
        if self.handle:
            lib.fftw_amd_destroy_slab_plan(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
