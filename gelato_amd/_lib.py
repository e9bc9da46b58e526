"""ctypes binding of libgelato_amd.so (the C-ABI of include/gelato_amd.h).

There is no CPU fallback anywhere in this package: if the HIP library is missing
or no MI355X is visible, calls fail loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# GELATO_AMD_LIB selects another build of the same C-ABI (profiling / ablation variants)
SO_PATH = os.environ.get("GELATO_AMD_LIB") or os.path.join(_HERE, "libgelato_amd.so")
_LIB = None

GEL_OK, GEL_NONFINITE = 0, 1
NUM_BLOCKS = 13

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class GelProblemDesc(C.Structure):
    _fields_ = [
        ("num_sections", C.c_int32),
        ("num_nodes", _ip),
        ("thrust", _dp), ("massflow", _dp), ("reference_area", _dp), ("nozzle_area", _dp),
        ("engine_on", _ip), ("attitude_hold", _ip),
        ("unit_mass", C.c_double), ("unit_position", C.c_double), ("unit_velocity", C.c_double),
        ("unit_u", C.c_double), ("unit_t", C.c_double),
        ("dx", C.c_double), ("barC20", C.c_double),
        ("wind_rows", C.c_int32), ("wind_table", _dp),
        ("ca_rows", C.c_int32), ("ca_table", _dp),
        ("D", _dp), ("tau", _dp),
        ("device", C.c_int32), ("flags", C.c_int32),
    ]


class GelLinearRow(C.Structure):
    _fields_ = [("idx0", C.c_int32), ("idx1", C.c_int32), ("coef0", C.c_double), ("coef1", C.c_double), ("c0", C.c_double)]


class GelNodefnRow(C.Structure):
    _fields_ = [("fn", C.c_int32), ("node", C.c_int32), ("tcol", C.c_int32), ("mode", C.c_int32), ("p", C.c_double * 8)]


class GelCallbackIO(C.Structure):
    _fields_ = [("res", _dp), ("vals_full", _dp), ("fill_constants", C.c_int32), ("rows_con", _dp), ("rows_jfn", _dp),
                ("aero_con", _dp * 3), ("aero_jac", _dp * 3)]


class GelDims(C.Structure):
    _fields_ = [
        ("S", C.c_int32), ("N", C.c_int32), ("M", C.c_int32), ("num_vars", C.c_int32),
        ("num_rows", C.c_int32 * 4),
        ("block_nnz", C.c_int64 * NUM_BLOCKS),
        ("block_shape", (C.c_int64 * 2) * NUM_BLOCKS),
        ("total_nnz", C.c_int64), ("num_var_entries", C.c_int64), ("algorithmic_bytes", C.c_int64),
        ("stored_bytes", C.c_int64),
    ]


# every symbol include/gelato_amd.h declares, with its signature
SIGNATURES = {
    "gel_lgr_nodes": (C.c_int, [C.c_int32, _dp]),
    "gel_lgr_diffmat": (C.c_int, [C.c_int32, _dp]),
    "gel_problem_create": (C.c_int, [C.POINTER(GelProblemDesc), C.POINTER(C.c_void_p)]),
    "gel_problem_destroy": (C.c_int, [C.c_void_p]),
    "gel_problem_dims": (C.c_int, [C.c_void_p, C.POINTER(GelDims)]),
    "gel_problem_D": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "gel_problem_tau": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "gel_pattern": (C.c_int, [C.c_void_p, C.c_int32, _ip, _ip]),
    "gel_pattern_all": (C.c_int, [C.c_void_p, _ip, _ip]),
    "gel_const_values": (C.c_int, [C.c_void_p, _dp]),
    "gel_var_index": (C.c_int, [C.c_void_p, _lp]),
    "gel_full_source": (C.c_int, [C.c_void_p, _ip]),
    "gel_eval_residual": (C.c_int, [C.c_void_p, _dp, _dp]),
    "gel_eval_jacobian": (C.c_int, [C.c_void_p, _dp, _dp, C.c_int32]),
    "gel_eval": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_int32]),
    "gel_eval_batch": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "gel_eval_batch_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_expand_full_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_eval_shard_units_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_void_p]),
    "gel_unit_owner": (C.c_int, [C.c_void_p, _ip, _ip]),
    "gel_pinned_buffers": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4),
    "gel_fill_full_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gel_update_full_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_eval_full_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_shard_plan": (C.c_int, [C.c_void_p, C.c_int32, _ip, _lp, _lp, _lp]),
    "gel_eval_shard_packed_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "gel_shard_unpack_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]),
    "gel_num_chunks": (C.c_int, [C.c_void_p, _ip]),
    "gel_chunk_phase": (C.c_int, [C.c_void_p, _ip]),
    "gel_launch_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _ip]),
    "gel_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gel_jac_fd": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    "gel_jac_fd_block_dims": (C.c_int, [C.c_void_p, C.c_int32, _lp, _lp, _lp, _lp]),
    "gel_jac_fd_block_cols": (C.c_int, [C.c_void_p, C.c_int32, _ip]),
    "gel_jac_fd_blocks": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp]),
    "gel_jac_fd_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gel_aero_configure": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _ip, _ip, _dp]),
    "gel_aero_dims": (C.c_int, [C.c_void_p, C.c_int32, _ip, _lp]),
    "gel_aero_pattern": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _ip, _ip]),
    "gel_eval_aero": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "gel_eval_aero_all": (C.c_int, [C.c_void_p, C.c_int32, _dp, C.POINTER(_dp), C.POINTER(_dp)]),
    "gel_eval_aero_all_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.c_void_p]),
    "gel_aero_record_layout": (C.c_int, [C.c_void_p, _lp, _lp, _lp]),
    "gel_aero_record_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _lp]),
    "gel_eval_batch_aero_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_rows_configure": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(GelLinearRow), C.c_int32, C.POINTER(GelNodefnRow)]),
    "gel_rows_dims": (C.c_int, [C.c_void_p, _ip, _ip]),
    "gel_rows_eval": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp]),
    "gel_rows_eval_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gel_eval_callback": (C.c_int, [C.c_void_p, _dp, C.POINTER(GelCallbackIO)]),
    "gel_initial_guess": (C.c_int, [C.c_void_p, C.c_int32, _dp, _dp, _dp, _dp]),
    "gel_output_table": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double, C.c_double, _dp]),
    "gel_dynamics_velocity": (C.c_int, [C.c_int32, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int32, _dp, C.c_int32,
                                         _dp, C.c_double, _dp]),
    "gel_dynamics_velocity_NoAir": (C.c_int, [C.c_int32, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp]),
    "gel_dynamics_quaternion": (C.c_int, [C.c_int32, _dp, _dp, C.c_double, _dp]),
    "gel_point_eval": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, C.c_int32, _dp]),
    "gel_last_error": (C.c_char_p, []),
    "gel_version": (C.c_char_p, []),
}


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force and os.path.exists(SO_PATH):
        os.remove(SO_PATH)
    subprocess.check_call(["make", "-s", "-j3", "-C", src_dir])   # three translation units (kernels, the AERO instantiation, host side)
    if not os.path.exists(SO_PATH):
        raise RuntimeError("building %s failed" % SO_PATH)
    _write_build_info()
    return SO_PATH


BUILD_INFO_PATH = os.path.join(_HERE, "build_info.json")


def so_sha256(path=None):
    """sha256 of the engine library as it lies on disk (the build is bit-reproducible: same sources + same hipcc -> same bytes)"""
    import hashlib
    h = hashlib.sha256()
    with open(path or SO_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def device_code_sha256(path=None):
    """sha256 over the gfx950 code objects embedded in the library (the AMDGPU ELF images of its fat binary): what the device executes.
    A change of the host code alone (a line number in an error string) changes so_sha256 but not this -- hardware counters recorded
    with one library describe every library with the same device code."""
    import hashlib, struct
    b = open(path or SO_PATH, "rb").read()
    h = hashlib.sha256()
    off, n = 1, 0
    while True:
        i = b.find(b"\x7fELF", off)
        if i < 0:
            break
        if i + 0x40 <= len(b) and struct.unpack_from("<H", b, i + 18)[0] == 224:      # EM_AMDGPU
            shoff = struct.unpack_from("<Q", b, i + 0x28)[0]
            shentsize, shnum = struct.unpack_from("<HH", b, i + 0x3A)
            size = shoff + shentsize * shnum
            h.update(b[i:i + size])
            n += 1
            off = i + max(size, 4)
        else:
            off = i + 4
    return h.hexdigest() if n else None


def _write_build_info():
    """Provenance of the in-tree library, written where it is built (this container has the git history, the GPU box does not;
    the file travels with the snapshot): recorded counter files name the build they describe, bench.py ignores the others."""
    import json
    info = {"so_sha256": so_sha256(os.path.join(_HERE, "libgelato_amd.so")), "git_head": None, "git_dirty": None}
    try:
        root = os.path.dirname(_HERE)
        info["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
        info["git_dirty"] = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "gelato_amd/csrc", "include"],
                                                         text=True, stderr=subprocess.DEVNULL).strip())
    except Exception:  # noqa: BLE001  (no git on the GPU box)
        pass
    try:
        old = json.load(open(BUILD_INFO_PATH))
        if old.get("so_sha256") == info["so_sha256"] and info["git_head"] is None:
            return      # same library, rebuilt where there is no git: keep what the build container wrote
    except Exception:  # noqa: BLE001
        pass
    with open(BUILD_INFO_PATH, "w") as f:
        json.dump(info, f)


def build_info():
    """{"so_sha256": of the library that is LOADED (GELATO_AMD_LIB honoured), "git_head", "git_dirty": of the build that produced
    the in-tree library, if that is the one loaded}"""
    import json
    out = {"so_sha256": so_sha256(), "git_head": None, "git_dirty": None, "device_code_sha256": device_code_sha256()}
    try:
        rec = json.load(open(BUILD_INFO_PATH))
        if rec.get("so_sha256") == out["so_sha256"]:
            out.update(git_head=rec.get("git_head"), git_dirty=rec.get("git_dirty"))
    except Exception:  # noqa: BLE001
        pass
    return out


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as
    /opt/rocm's).  Whichever copy is mapped first serves every later user of that SONAME, and torch
    cannot initialise on top of a foreign copy ("No HIP GPUs are available").  Device pointers and
    stream handles cross between torch and this library (plumbing: memory, streams, RCCL), so both must
    sit on the SAME runtime: if torch is installed, it is imported (and its runtime initialised) before
    libgelato_amd.so is loaded, so the library binds to torch's copy."""
    # Importing torch first (what bench.py does) is the load order that is known to be stable; a late
    # torch.cuda initialisation on top of an already-initialised foreign mapping was seen to deadlock.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "gelato_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C gelato_amd/csrc`).  There is no CPU fallback." % SO_PATH)
        _preload_torch_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError here = the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class GelatoAmdError(RuntimeError):
    pass


def check(rc):
    """negative -> raise; 0 / GEL_NONFINITE are returned to the caller."""
    if rc < 0:
        msg = lib().gel_last_error()
        raise GelatoAmdError("gelato_amd C-ABI error %d: %s" % (rc, msg.decode() if msg else "?"))
    return rc
