"""Engine: Python handle on a device-resident LGR defect/Jacobian problem.

Thin wrapper over the C-ABI (include/gelato_amd.h).  Holds no numerics of its own.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GelCallbackIO, GelDims, GelLinearRow, GelNodefnRow, GelProblemDesc, check, lib

GROUPS = ["mass", "pos", "vel", "quat"]
# funcs / funcsSens keys of the reference callbacks (Trajectory_Optimization.py:199-210,250-261)
CON_NAMES = {"mass": "eqcon_dyn_mass", "pos": "eqcon_dyn_pos", "vel": "eqcon_dyn_vel", "quat": "eqcon_dyn_quat"}
# Jacobian blocks in C-ABI order: (group, var); var order per group = the reference's dict order
BLOCKS = [("mass", "mass"), ("mass", "t"),
          ("pos", "position"), ("pos", "velocity"), ("pos", "t"),
          ("vel", "mass"), ("vel", "position"), ("vel", "velocity"), ("vel", "quaternion"), ("vel", "t"),
          ("quat", "quaternion"), ("quat", "u"), ("quat", "t")]
XKEYS = ["mass", "position", "velocity", "quaternion", "u", "t"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _d(a):
    return a.ctypes.data_as(_dp)


def pack_x(xdict, out=None):
    """xdict (reference layout, Trajectory_Optimization.py:318-352) -> packed decision vector (into `out` if given)."""
    return np.concatenate([np.asarray(xdict[k], dtype=np.float64).ravel() for k in XKEYS], out=out)


class _PinnedView(np.ndarray):
    """numpy view of the handle's pinned host memory that keeps its Engine alive (attribute _engine on the base view)"""


class Engine:
    """prob: dict with num_nodes, thrust, massflow, reference_area, nozzle_area, engine_on,
    attitude_hold, units (mass, position, velocity, u, t), dx, wind_table [K,3], ca_table [K,2];
    optional D / tau (lists per phase, e.g. pdict["ps_params"].D(i)); barC20 (0 -> C++ constant)."""

    def __init__(self, prob, D=None, tau=None, barC20=0.0, device=0, flags=0):
        L = lib()
        self._keep = []
        nn = np.ascontiguousarray(prob["num_nodes"], dtype=np.int32)
        S = len(nn)

        def darr(key):
            a = _f64(prob[key])
            assert a.size == S, key
            self._keep.append(a)
            return _d(a)

        def iarr(key):
            a = np.ascontiguousarray(prob[key], dtype=np.int32)
            assert a.size == S, key
            self._keep.append(a)
            return a.ctypes.data_as(_ip)

        wind, ca = _f64(prob["wind_table"]), _f64(prob["ca_table"])
        units = _f64(prob["units"])
        self.prob = {k: np.array(prob[k]) for k in ("num_nodes", "thrust", "massflow", "reference_area", "nozzle_area",
                                                    "engine_on", "attitude_hold")}
        d = GelProblemDesc()
        d.num_sections = S
        d.num_nodes = nn.ctypes.data_as(_ip)
        d.thrust, d.massflow = darr("thrust"), darr("massflow")
        d.reference_area, d.nozzle_area = darr("reference_area"), darr("nozzle_area")
        d.engine_on, d.attitude_hold = iarr("engine_on"), iarr("attitude_hold")
        d.unit_mass, d.unit_position, d.unit_velocity, d.unit_u, d.unit_t = [float(u) for u in units]
        d.dx = float(prob["dx"])
        d.barC20 = float(barC20)
        d.wind_rows, d.wind_table = wind.shape[0], _d(wind)
        d.ca_rows, d.ca_table = ca.shape[0], _d(ca)
        if D is not None:
            Dall = _f64(np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in D]))
            tall = _f64(np.concatenate([np.asarray(x, dtype=np.float64).ravel() for x in tau]))
            d.D, d.tau = _d(Dall), _d(tall)
        else:
            d.D, d.tau = None, None
        d.device = device
        self.flags = int(flags)
        d.flags = int(flags)  # 0, or GEL_FLAG_DX_MFMA (1) / GEL_FLAG_DX_VALU (2) to force the D.X path, GEL_FLAG_NO_PACK (4), GEL_FLAG_FD_RECOMPUTE (8)
        h = C.c_void_p()
        check(L.gel_problem_create(C.byref(d), C.byref(h)))
        self._h = h
        dims = GelDims()
        check(L.gel_problem_dims(h, C.byref(dims)))
        self.S, self.N, self.M, self.nvars = dims.S, dims.N, dims.M, dims.num_vars
        self.num_nodes = nn.copy()
        self.nrows = list(dims.num_rows)
        self.nres = 11 * self.N
        self.block_nnz = [int(v) for v in dims.block_nnz]
        self.block_shape = [(int(s[0]), int(s[1])) for s in dims.block_shape]
        self.total_nnz = int(dims.total_nnz)
        self.V = int(dims.num_var_entries)
        self.algorithmic_bytes = int(dims.algorithmic_bytes)   # SURVEY 8(d) A_min per eval
        self.stored_bytes = int(dims.stored_bytes)             # residual + compact values actually written
        self.block_off = np.concatenate([[0], np.cumsum(self.block_nnz)]).astype(np.int64)
        self.row_off = {"mass": 0, "pos": self.N, "vel": 4 * self.N, "quat": 7 * self.N}
        self._pattern = None
        self._vals = None      # persistent full COO values (constants pre-filled)
        self._var_idx = None
        self._src = None
        self._nlin = self._nfn = 0
        self._cfg_gen = 0       # bumped by every (re)configuration of the row table or of an aero kind
        self.shard_plan_key = None   # (ranks, width, unit ranges) of the plan the handle holds (shard_plan)

    # ------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib().gel_problem_destroy(self._h)
            self._h = None
            # the pinned buffers died with the handle: drop the views this object handed out of them (arrays a caller still holds
            # point at freed memory, like any view of a closed resource)
            self.__dict__.pop("_pinned", None)
            self.__dict__.pop("_cb_out", None)
            self._vals = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def D(self, i):
        n = int(self.num_nodes[i])
        out = np.zeros((n, n + 1))
        check(lib().gel_problem_D(self._h, i, _d(out)))
        return out

    def tau(self, i):
        out = np.zeros(int(self.num_nodes[i]))
        check(lib().gel_problem_tau(self._h, i, _d(out)))
        return out

    def pattern(self):
        """[(rows int32, cols int32)] for the 13 blocks (fixed; depends only on the static problem)."""
        if self._pattern is None:
            rows = np.zeros(self.total_nnz, dtype=np.int32)
            cols = np.zeros(self.total_nnz, dtype=np.int32)
            check(lib().gel_pattern_all(self._h, rows.ctypes.data_as(_ip), cols.ctypes.data_as(_ip)))   # one pass
            self._pattern = [(rows[self.block_off[b]:self.block_off[b + 1]].copy(),
                              cols[self.block_off[b]:self.block_off[b + 1]].copy()) for b in range(13)]
        return self._pattern

    def const_values(self):
        v = np.zeros(self.total_nnz)
        check(lib().gel_const_values(self._h, _d(v)))
        return v

    def var_index(self):
        if self._var_idx is None:
            idx = np.zeros(self.V, dtype=np.int64)
            check(lib().gel_var_index(self._h, idx.ctypes.data_as(_lp)))
            self._var_idx = idx
        return self._var_idx

    def full_source(self):
        """gather map full <- compact: -1 constant, s >= 0: compact[s], s <= -2: -compact[-2 - s]"""
        if self._src is None:
            src = np.zeros(self.total_nnz, dtype=np.int32)
            check(lib().gel_full_source(self._h, src.ctypes.data_as(_ip)))
            self._src = src
        return self._src

    def var_mask(self):
        """boolean [total_nnz]: entries that depend on x (everything the gather map takes from the compact vector)"""
        return self.full_source() != -1

    # ------------------------------------------------------------------
    def eval_residual(self, x):
        """-> (res [11N], status)"""
        x = _f64(x)
        assert x.size == self.nvars
        res = np.empty(self.nres)
        self._direct_call()
        rc = check(lib().gel_eval_residual(self._h, _d(x), _d(res)))
        return res, rc

    def _direct_call(self):
        """A one-vector call outside eval_callback rewrites the handle's pinned value array (the COO-direct path writes through
        it even when the caller names its own output) and residual vector -- the memory eval_callback hands out as its frame's
        `vals` / `res`.  A frame cached by the reference-named functions (con_dynamics._State.frame) is only valid for the
        generation it was evaluated in: bump it, so that the next mirror call evaluates again."""
        self._cfg_gen = getattr(self, "_cfg_gen", 0) + 1

    def eval_jacobian(self, x, out=None):
        """-> (vals_full [total_nnz], status).  `out` (from a previous call) is updated in place:
        only the x-dependent entries are rewritten."""
        x = _f64(x)
        assert x.size == self.nvars
        fill = 0
        if out is None:
            out = np.empty(self.total_nnz)
            fill = 1
        self._direct_call()
        rc = check(lib().gel_eval_jacobian(self._h, _d(x), _d(out), fill))
        return out, rc

    def pinned_buffers(self):
        """(res [11N], vals_full [total_nnz]): numpy views of the handle's own pinned host buffers (gel_pinned_buffers).  Passing
        them as `res_out` / `out` of eval / eval_jacobian (eval_callback uses them by itself) makes the one-vector calls
        zero-copy: the kernel writes the residual rows and every all-x-dependent block of the value vector straight into them.
        They belong to the engine and are rewritten by its next one-vector call."""
        pb = self.__dict__.get("_pinned")
        if pb is None:
            r, v, x0, x1 = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
            check(lib().gel_pinned_buffers(self._h, C.byref(r), C.byref(v), C.byref(x0), C.byref(x1)))
            dp = C.POINTER(C.c_double)
            # the views hold a reference to this engine (every slice of them reaches it through .base): the pinned memory is
            # freed by close(), and a garbage-collected engine closes itself -- not while one of these arrays is alive
            def view(ptr, n):
                a = np.ctypeslib.as_array(C.cast(ptr, dp), shape=(n,)).view(_PinnedView)
                a._engine = self
                return a
            res = view(r, self.nres)
            vals = view(v, max(self.total_nnz, 1))[:self.total_nnz]
            xb = [view(q, self.nvars) for q in (x0, x1)]
            pb = self._pinned = (res, vals, _d(res), _d(vals), xb, [_d(b) for b in xb])
        return pb[0], pb[1]

    def pinned_x(self):
        """-> ([x0, x1], [their ctypes pointers]): two pinned decision-vector buffers of the handle; a one-vector call whose x is
        one of them (eval_callback(x, .., xptr=pointer)) reads it in place instead of copying it into the staging buffer"""
        self.pinned_buffers()
        return self._pinned[4], self._pinned[5]

    def eval(self, x, out=None, res_out=None):
        """-> (res, vals_full, status).  out / res_out: arrays to write into (the engine's pinned_buffers() make it zero-copy)."""
        x = _f64(x)
        pb = self.__dict__.get("_pinned")
        if res_out is None:
            res, rp = np.empty(self.nres), None
        else:
            res = res_out
            rp = pb[2] if (pb is not None and res_out is pb[0]) else _d(res_out)
        fill = 0
        if out is None:
            out = np.empty(self.total_nnz)
            fill = 1
        vp = pb[3] if (pb is not None and out is pb[1]) else _d(out)
        self._direct_call()
        rc = check(lib().gel_eval(self._h, _d(x), rp if rp is not None else _d(res), vp, fill))
        return res, out, rc

    def eval_batch(self, X, want_res=True, want_jac=True, out=None):
        """X [B, nvars] -> (res [B, 11N] | None, jvar [B, V] | None, status).  `out` = (res, jvar) arrays of a
        previous call to write into (saves the page faults of fresh 100-MB arrays on large batches)."""
        X = _f64(X).reshape(-1, self.nvars)
        B = X.shape[0]
        if out is not None:
            res, jv = out
            for a, shp, want in ((res, (B, self.nres), want_res), (jv, (B, self.V), want_jac)):
                if want and (a is None or a.shape != shp or a.dtype != np.float64 or not a.flags.c_contiguous):
                    raise ValueError("out arrays must be C-contiguous float64 of shape %r" % (shp,))
        else:
            res = np.empty((B, self.nres)) if want_res else None
            jv = np.empty((B, self.V)) if want_jac else None
        rc = check(lib().gel_eval_batch(self._h, B, _d(X), _d(res) if want_res else None,
                                        _d(jv) if want_jac else None))
        return res, jv, rc

    def expand(self, jvar):
        """compact [.., V] -> full [.., total_nnz] on the host (numpy), using the constant template."""
        jvar = np.asarray(jvar)
        src = self.full_source()
        full = np.broadcast_to(self.const_values(), jvar.shape[:-1] + (self.total_nnz,)).copy()
        pos, neg = np.nonzero(src >= 0)[0], np.nonzero(src <= -2)[0]
        full[..., pos] = jvar[..., src[pos]]
        full[..., neg] = -jvar[..., -2 - src[neg]]
        return full

    # device-pointer API (pointers are integers, e.g. torch.Tensor.data_ptr())
    def eval_batch_device(self, B, d_x, d_res, d_jvar, stream=0):
        check(lib().gel_eval_batch_device(self._h, B, d_x, d_res or None, d_jvar or None, stream or None))

    def expand_full_device(self, B, d_jvar, d_jfull, stream=0):
        check(lib().gel_expand_full_device(self._h, B, d_jvar, d_jfull, stream or None))

    def fill_full_device(self, B, d_jfull, stream=0):
        """the constant template into d_jfull [B][total_nnz], once (then update_full_device after every evaluation)"""
        check(lib().gel_fill_full_device(self._h, B, d_jfull, stream or None))

    def update_full_device(self, B, d_jvar, d_jfull, stream=0):
        """only the x-dependent entries of d_jfull [B][total_nnz] from d_jvar [B][V] (the buffer holds the constants already)"""
        check(lib().gel_update_full_device(self._h, B, d_jvar, d_jfull, stream or None))

    def eval_full_device(self, B, d_x, d_res, d_jvar, d_jfull, stream=0):
        """one evaluation with every COO value valid in d_jfull [B][total_nnz] afterwards (fill_full_device once before): the fused
        launch + the update in place; a launch that fits the Infinity Cache keeps its compact values there for the update"""
        check(lib().gel_eval_full_device(self._h, B, d_x, d_res or None, d_jvar, d_jfull, stream or None))

    def launch_info(self, B, want_res=True, want_jac=True):
        """-> [jacobian, mfma, split, wavefronts, pack] of the kernel form a launch of B vectors takes"""
        info = (C.c_int32 * 5)()
        check(lib().gel_launch_info(self._h, int(B), int(bool(want_res)), int(bool(want_jac)), info))
        return [int(v) for v in info]

    def sync(self, stream=0):
        return check(lib().gel_sync(self._h, stream or None))

    # work items (one 64-node chunk of one phase each) for phase-sharded multi-GPU evaluation
    def num_chunks(self):
        n = C.c_int32()
        check(lib().gel_num_chunks(self._h, C.byref(n)))
        return n.value

    def chunk_phase(self):
        out = np.zeros(self.num_chunks(), dtype=np.int32)
        check(lib().gel_chunk_phase(self._h, out.ctypes.data_as(_ip)))
        return out

    def unit_owner(self):
        """(res_owner [11N], jvar_owner [V]): the unit (4 * work item + part) that writes each output entry"""
        ro = np.zeros(self.nres, dtype=np.int32)
        jo = np.zeros(self.V, dtype=np.int32)
        check(lib().gel_unit_owner(self._h, ro.ctypes.data_as(_ip), jo.ctypes.data_as(_ip)))
        return ro, jo

    def eval_shard_units_device(self, B, d_x, d_res, d_jvar, unit_begin, unit_count, stream=0):
        """unit = 4 * work_item + part (part 0: all but the three position sweeps; 1..3: one position sweep)."""
        check(lib().gel_eval_shard_units_device(self._h, B, d_x, d_res or None, d_jvar, int(unit_begin),
                                                int(unit_count), stream or None))

    def shard_plan(self, unit_begin):
        """Packed unit-shard exchange layout (gel_shard_plan) for ranks holding the contiguous unit ranges
        [unit_begin[r], unit_begin[r + 1]) -> (width, res_pos [11N], jvar_pos [V]); pos = rank * width + offset."""
        ub = np.ascontiguousarray(unit_begin, dtype=np.int32)
        width = C.c_int64(0)
        rp = np.zeros(self.nres, dtype=np.int64)
        jp = np.zeros(self.V, dtype=np.int64)
        _lp = C.POINTER(C.c_int64)
        check(lib().gel_shard_plan(self._h, len(ub) - 1, ub.ctypes.data_as(_ip), C.byref(width), rp.ctypes.data_as(_lp),
                                   jp.ctypes.data_as(_lp)))
        self.shard_plan_key = (len(ub) - 1, int(width.value), ub.tobytes())   # the plan the handle holds NOW (the next call replaces it)
        return int(width.value), rp, jp

    def eval_shard_packed_device(self, B, d_x, d_out, rank, stream=0, plan=None):
        """rank `rank`'s units of all B vectors straight into its slice of the exchange buffer d_out [nranks][B][width].
        plan = (nranks, width) the buffer was sized for (default: the handle's current plan); the call fails if the handle
        holds another plan by now."""
        if plan is None and self.shard_plan_key is None:
            raise _lib.GelatoAmdError("no shard plan on this handle: call shard_plan(unit_begin) first (or pass plan=(ranks, width))")
        nr, w = plan if plan is not None else self.shard_plan_key[:2]
        check(lib().gel_eval_shard_packed_device(self._h, B, d_x, d_out, int(rank), int(nr), int(w), stream or None))

    def shard_unpack_device(self, B, d_out, d_res, d_jvar, stream=0, plan=None):
        """exchange buffer -> the ordinary res [B][11N] / jvar [B][V] layouts (one gather launch; either may be 0)"""
        if plan is None and self.shard_plan_key is None:
            raise _lib.GelatoAmdError("no shard plan on this handle: call shard_plan(unit_begin) first (or pass plan=(ranks, width))")
        nr, w = plan if plan is not None else self.shard_plan_key[:2]
        check(lib().gel_shard_unpack_device(self._h, B, d_out, d_res or None, d_jvar or None, int(nr), int(w), stream or None))

    def jac_fd(self, group, x):
        gi = GROUPS.index(group)
        x = _f64(x)
        J = np.empty((self.nrows[gi], self.nvars))
        rc = check(lib().gel_jac_fd(self._h, gi, _d(x), _d(J)))
        return J, rc

    def jac_fd_block_dims(self, group):
        """-> (rows [S], cols [S], row0 [S], offset [S + 1]) of the group's per-phase blocks, (cols of phase i: global column of each
        local column) -- the phase's rows see only these columns (every other column of lib/jac_fd.py's dense result is zero)."""
        key = ("jfd_dims", group)
        if key not in self.__dict__:
            S = self.S
            r, c, r0, off = (np.zeros(S + (1 if k == 3 else 0), dtype=np.int64) for k in range(4))
            check(lib().gel_jac_fd_block_dims(self._h, GROUPS.index(group), *(a.ctypes.data_as(C.POINTER(C.c_int64)) for a in (r, c, r0, off))))
            cols = []
            for i in range(S):
                ci = np.zeros(int(c[i]), dtype=np.int32)
                check(lib().gel_jac_fd_block_cols(self._h, i, ci.ctypes.data_as(_ip)))
                cols.append(ci)
            self.__dict__[key] = (r, c, r0, off, cols)
        return self.__dict__[key]

    def jac_fd_blocks(self, group, x):
        """the group's forward-difference Jacobian as per-phase blocks: [(row0, global cols, block [rows, cols])], rc"""
        r, c, r0, off, cols = self.jac_fd_block_dims(group)
        x = _f64(x)
        buf = np.empty(int(off[-1]))
        rc = check(lib().gel_jac_fd_blocks(self._h, GROUPS.index(group), _d(x), _d(buf)))
        return [(int(r0[i]), cols[i], buf[int(off[i]):int(off[i + 1])].reshape(int(r[i]), int(c[i]))) for i in range(self.S)], rc

    def jac_fd_device(self, group, d_x, d_J, blocks=False, stream=0):
        """device pointers (ints): x [nvars] -> dense J [nrows[group]][nvars] or the blocks (jac_fd_block_dims), all in HBM"""
        check(lib().gel_jac_fd_device(self._h, GROUPS.index(group), d_x, d_J, 1 if blocks else 0, stream or None))

    # ---- aero path constraints (lib/con_aero.py): kind in AERO_KINDS ----
    AERO_KINDS = ["alpha", "q", "qalpha"]
    AERO_VARS = ["position", "velocity", "quaternion", "t"]

    def aero_configure(self, kind, spec):
        """spec: rows of (phase, range_all, limit); limit = units[3] of con_aero.py."""
        spec = np.asarray(spec, dtype=np.float64).reshape(-1, 3)
        ph = np.ascontiguousarray(spec[:, 0], dtype=np.int32)
        ra = np.ascontiguousarray(spec[:, 1], dtype=np.int32)
        lim = _f64(spec[:, 2])
        check(lib().gel_aero_configure(self._h, self.AERO_KINDS.index(kind), len(ph), ph.ctypes.data_as(_ip),
                                       ra.ctypes.data_as(_ip), _d(lim)))
        self._aero_dims = {}
        self._aero_out = {}
        self._cb_out = {}
        self._cfg_gen = getattr(self, "_cfg_gen", 0) + 1

    def aero_dims(self, kind):
        d = self.__dict__.setdefault("_aero_dims", {})
        if kind not in d:
            n = C.c_int32()
            nnz = (C.c_int64 * 4)()
            check(lib().gel_aero_dims(self._h, self.AERO_KINDS.index(kind), C.byref(n), nnz))
            d[kind] = (n.value, [int(v) for v in nnz])
        return d[kind]

    def aero_pattern(self, kind):
        nrow, nnz = self.aero_dims(kind)
        out = []
        for v in range(4):
            r = np.zeros(nnz[v], dtype=np.int32)
            c = np.zeros(nnz[v], dtype=np.int32)
            check(lib().gel_aero_pattern(self._h, self.AERO_KINDS.index(kind), v, r.ctypes.data_as(_ip),
                                         c.ctypes.data_as(_ip)))
            out.append((r, c))
        return out

    def eval_aero(self, kind, X, want_jac=True):
        """X [B, nvars] (or [nvars]) -> (con [B, nrows], jac_vals [B, sum nnz] | None, status)"""
        X = _f64(X).reshape(-1, self.nvars)
        B = X.shape[0]
        nrow, nnz = self.aero_dims(kind)
        con = np.empty((B, nrow))
        jv = np.empty((B, sum(nnz))) if want_jac else None
        rc = check(lib().gel_eval_aero(self._h, self.AERO_KINDS.index(kind), B, _d(X), _d(con),
                                       _d(jv) if want_jac else None))
        return con, jv, rc

    def eval_aero_all(self, X, want_jac=True, kinds=None, reuse=False):
        """all configured kinds in one launch: X [B, nvars] (or [nvars]) -> ({kind: con [B, nrows]}, {kind: jac_vals} | None,
        status); kinds without rows are left out.  reuse=True hands out the engine's own output arrays (overwritten by
        the next call of the same shape) instead of fresh ones."""
        X = _f64(X).reshape(-1, self.nvars)
        B = X.shape[0]
        key = (B, bool(want_jac), None if kinds is None else tuple(kinds))
        slot = self.__dict__.setdefault("_aero_out", {}).get(key) if reuse else None
        if slot is None:
            con, jac = {}, {}
            cp, jp = (_dp * 3)(), (_dp * 3)()
            for i, kind in enumerate(self.AERO_KINDS):
                nrow, nnz = self.aero_dims(kind)
                if nrow and (kinds is None or kind in kinds):
                    con[kind] = np.empty((B, nrow))
                    cp[i] = _d(con[kind])
                    if want_jac:
                        jac[kind] = np.empty((B, sum(nnz)))
                        jp[i] = _d(jac[kind])
            slot = (con, jac, cp, jp)
            if reuse:
                self._aero_out[key] = slot
        con, jac, cp, jp = slot
        rc = check(lib().gel_eval_aero_all(self._h, B, _d(X), cp, jp if want_jac else None))
        return con, (jac if want_jac else None), rc

    def eval_aero_all_device(self, B, d_x, d_con, d_jac=None, stream=0):
        """device pointers (ints): d_con / d_jac = 3 entries per kind (0 = not wanted)"""
        cp = (C.c_void_p * 3)(*[p or None for p in d_con])
        jp = (C.c_void_p * 3)(*[p or None for p in d_jac]) if d_jac is not None else None
        check(lib().gel_eval_aero_all_device(self._h, B, d_x, cp, jp, stream or None))

    def aero_record_layout(self):
        """-> (width, {kind: con index [nrows]}, {kind: jac index [sum nnz]}): the per-vector record of eval_batch_aero_device and
        the gather that restores eval_aero_all's arrays from it -- con[kind] = aero_gather(record, con_idx[kind]), jac likewise (the four
        blocks position | velocity | quaternion | t concatenated, as eval_aero_all returns them; index -1: an exact zero, not stored)"""
        w = C.c_int64()
        oc, oj = (C.c_int64 * 6)(), (C.c_int64 * 6)()
        check(lib().gel_aero_record_layout(self._h, C.byref(w), oc, oj))
        con, jac = {}, {}
        for i, kind in enumerate(self.AERO_KINDS):
            nrow, nnz = self.aero_dims(kind)
            ci = np.zeros(nrow, dtype=np.int64)
            if nrow:
                check(lib().gel_aero_record_map(self._h, i, -1, ci.ctypes.data_as(_lp)))
            parts = []
            for v in range(4):
                ji = np.zeros(nnz[v], dtype=np.int64)
                if nnz[v]:
                    check(lib().gel_aero_record_map(self._h, i, v, ji.ctypes.data_as(_lp)))
                parts.append(ji)
            con[kind], jac[kind] = ci, np.concatenate(parts)
        return int(w.value), con, jac

    @staticmethod
    def aero_gather(records, idx):
        """records [..., width], idx from aero_record_layout -> the reference's array; index -1 = an exact zero that is not stored"""
        records = np.asarray(records)
        out = records[..., np.maximum(idx, 0)]
        if (idx < 0).any():
            out[..., idx < 0] = 0.0
        return out

    def eval_batch_aero_device(self, B, d_x, d_res, d_jvar, d_aero, stream=0):
        """defect groups + aero path constraints of a resident batch (device pointers as ints): res [B, nres], jvar [B, V] as
        eval_batch_device, aero [B, width] one record per vector (aero_record_layout)"""
        check(lib().gel_eval_batch_aero_device(self._h, B, d_x, d_res, d_jvar, d_aero, stream or None))

    def eval_callback(self, x, want_jac, xptr=None):
        """ONE device round trip for one decision vector: the four defect groups, the row table (if configured) and the aero
        kinds (if configured), values only or values + derivatives.  -> dict of the engine's own output arrays (overwritten
        by the next call): res, vals (full COO values) | None, rows_con, rows_jfn | None, aero_con {kind}, aero_jac {kind} |
        None, rc."""
        if xptr is None:      # xptr: the caller's cached pointer to x (a persistent, contiguous float64 buffer of nvars doubles)
            x = _f64(x)
            assert x.size == self.nvars
            xptr = _d(x)
        key = bool(want_jac)
        slot = self.__dict__.setdefault("_cb_out", {}).get(key)
        if slot is None:
            io = GelCallbackIO()
            # the handle's own pinned buffers: the kernel writes the residual rows and the all-x-dependent blocks of the value
            # vector straight into them (gel_pinned_buffers: no host copy of either)
            pres, pvals = self.pinned_buffers()
            out = {"res": pres, "vals": None, "rows_con": None, "rows_jfn": None, "aero_con": {}, "aero_jac": {}}
            io.res = self._pinned[2]
            if want_jac:
                if self._vals is None:
                    self._vals = pvals          # constants in place; x-dependent entries rewritten per call
                out["vals"] = self._vals
                io.vals_full, io.fill_constants = (self._pinned[3] if self._vals is pvals else _d(self._vals)), 0
            if self._nlin + self._nfn:
                out["rows_con"] = np.empty(self._nlin + self._nfn)
                io.rows_con = _d(out["rows_con"])
                if want_jac:
                    out["rows_jfn"] = np.empty((max(self._nfn, 1), 7))
                    io.rows_jfn = _d(out["rows_jfn"])
            for i, kind in enumerate(self.AERO_KINDS):
                nrow, nnz = self.aero_dims(kind)
                if nrow:
                    out["aero_con"][kind] = np.empty(nrow)
                    io.aero_con[i] = _d(out["aero_con"][kind])
                    if want_jac:
                        out["aero_jac"][kind] = np.empty(sum(nnz))
                        io.aero_jac[i] = _d(out["aero_jac"][kind])
            slot = (io, out)
            self._cb_out[key] = slot
        io, out = slot
        out["rc"] = check(lib().gel_eval_callback(self._h, xptr, C.byref(io)))
        if out["rows_jfn"] is not None:
            out["rows_jfn"] = out["rows_jfn"][:self._nfn]
        return out

    OUTPUT_COLUMNS = ["thrust", "lat", "lon", "lat_IIP", "lon_IIP", "downrange", "altitude", "altitude_apogee", "altitude_perigee",
                      "inclination", "argument_perigee", "lon_ascending_node", "true_anomaly", "vel_ground_NED_X",
                      "vel_ground_NED_Y", "vel_ground_NED_Z", "accel_BODY_X", "aero_BODY_X", "heading_NED2BODY", "pitch_NED2BODY",
                      "roll_NED2BODY", "flightpath_vel_inertial_geocentric", "azimuth_vel_inertial_geocentric",
                      "thrust_direction_ECI_X", "thrust_direction_ECI_Y", "thrust_direction_ECI_Z", "vel_ground", "vel_air",
                      "AOA_total", "AOA_pitch", "AOA_yaw", "dynamic_pressure", "Q_alpha", "M"]   # gel_output_column

    def output_table(self, x, tx_res, launch_lat, launch_lon):
        """x [nvars], tx_res [M] (seconds) -> [M, 34] derived quantities of the reference's post-processing table
        (output_result.py:121-262), columns OUTPUT_COLUMNS; one device thread per state node"""
        x, tx = _f64(x), _f64(tx_res)
        if x.shape != (self.nvars,) or tx.shape != (self.M,):
            raise ValueError("x must have nvars entries and tx_res one time per state node")
        out = np.empty((self.M, len(self.OUTPUT_COLUMNS)))
        check(lib().gel_output_table(self._h, _d(x), _d(tx), float(launch_lat), float(launch_lon), _d(out)))
        return out

    def initial_guess(self, t_ref, table, knot_times):
        """initialize.py:322-409 behind the C-ABI: reference trajectory (t_ref [n], table [n, 13] = mass | pos 3 | vel 3 |
        quat 4 | body rates y, z) interpolated at the mesh's node times -> packed decision vector"""
        t_ref, table, knot_times = _f64(t_ref), _f64(table), _f64(knot_times)
        assert table.shape == (t_ref.size, 13) and knot_times.size == self.S + 1
        x = np.empty(self.nvars)
        rc = check(lib().gel_initial_guess(self._h, t_ref.size, _d(t_ref), _d(table), _d(knot_times), _d(x)))
        if rc == _lib.GEL_NONFINITE:
            # np.interp semantics of the reference (initialize.py:346-409): a node time on a zero-width interval of the table
            # (a repeated time at either end) extrapolates to NaN / Inf there; the guess is returned as the reference would
            # return it, with a warning
            import warnings
            warnings.warn("initial_guess: non-finite node values (the reference table repeats a time where a node time "
                          "falls): %d of %d" % (int(np.count_nonzero(~np.isfinite(x))), x.size), RuntimeWarning, stacklevel=2)
        return x

    # ---- knot / terminal / user rows (lib/con_init_terminal_knot.py, example/user_constraints.py) ----
    NODE_FUNCTIONS = {"orbit_energy": 0, "angular_momentum": 1, "inclination_rad": 2, "semi_major_axis": 3,
                      "eccentricity": 4, "periapsis_radius": 5, "apoapsis_radius": 6, "radius": 7, "speed": 8,
                      "latitude_deg": 9, "longitude_deg": 10, "altitude": 11, "lat_IIP_deg": 12, "lon_IIP_deg": 13,
                      "sin_elevation": 14, "downrange": 15}
    # row modes (include/gelato_amd.h): value f / p0 - p1 | (f - p1) / p0; difference of the value | scaled raw difference
    MODE_SHIFTED, MODE_RAW_DIFFERENCE, MODE_NEGATED = 1, 4, 8

    def var_offset(self, key):
        """first index of xdict[key] inside the packed decision vector"""
        M, N = self.M, self.N
        return {"mass": 0, "position": M, "velocity": 4 * M, "quaternion": 7 * M, "u": 11 * M, "t": 11 * M + 2 * N}[key]

    def rows_configure(self, linear, nodefn):
        """linear: rows (idx0, coef0, idx1 | -1, coef1, c0) -> (coef0 x[idx0] + coef1 x[idx1]) + c0;
        nodefn: rows (fn, node, p0, p1) -> f(r, v at state node) / p0 - p1 with its forward difference, or the long form
        (fn, node, tcol, mode, [p0 .. p7]) of include/gelato_amd.h (functions of the knot time, other value / difference forms)."""
        lin = (GelLinearRow * max(1, len(linear)))()
        for k, (i0, c0_, i1, c1_, cc) in enumerate(linear):
            lin[k] = GelLinearRow(int(i0), int(i1), float(c0_), float(c1_), float(cc))
        fn = (GelNodefnRow * max(1, len(nodefn)))()
        for k, row in enumerate(nodefn):
            if len(row) == 4:
                f, node, p0, p1 = row
                tcol, mode, pp = -1, 0, [p0, p1]
            else:
                f, node, tcol, mode, pp = row
            pp = [float(v) for v in pp] + [0.0] * (8 - len(pp))
            fn[k] = GelNodefnRow(int(self.NODE_FUNCTIONS.get(f, f)), int(node), int(tcol), int(mode), (C.c_double * 8)(*pp))
        check(lib().gel_rows_configure(self._h, len(linear), lin, len(nodefn), fn))
        self._nlin, self._nfn = len(linear), len(nodefn)
        self._cb_out = {}
        self._cfg_gen = getattr(self, "_cfg_gen", 0) + 1

    def rows_eval(self, X, want_jac=True):
        """X [B, nvars] (or [nvars]) -> (con [B, nlin + nfn], jfn [B, nfn, 7] | None, status); the seven difference columns
        are position xyz, velocity xyz of the row's node, then its knot time"""
        X = _f64(X).reshape(-1, self.nvars)
        B = X.shape[0]
        con = np.empty((B, self._nlin + self._nfn))
        jfn = np.empty((B, self._nfn, 7)) if want_jac else None
        rc = check(lib().gel_rows_eval(self._h, B, _d(X), _d(con), _d(jfn) if want_jac else None))
        return con, jfn, rc

    def rows_eval_device(self, B, d_x, d_con, d_jfn, stream=0):
        check(lib().gel_rows_eval_device(self._h, B, d_x, d_con, d_jfn or None, stream or None))

    # ------------------------------------------------------------------
    def split_x(self, x):
        M, N, S = self.M, self.N, self.S
        o = np.cumsum([0, M, 3 * M, 3 * M, 4 * M, 2 * N, S + 1])
        return {k: x[o[i]:o[i + 1]] for i, k in enumerate(XKEYS)}

    def split_res(self, res):
        N = self.N
        return {"mass": res[..., :N], "pos": res[..., N:4 * N], "vel": res[..., 4 * N:7 * N], "quat": res[..., 7 * N:]}

    def jac_dicts(self, vals_full):
        """full values -> {group: {var: {"coo": [rows, cols, vals], "shape": ...}}} (reference layout,
        lib/con_dynamics.py:75-76,108-113)."""
        pat = self.pattern()
        out = {g: {} for g in GROUPS}
        for b, (g, var) in enumerate(BLOCKS):
            r, c = pat[b]
            v = vals_full[self.block_off[b]:self.block_off[b + 1]]
            out[g][var] = {"coo": [r, c, v], "shape": self.block_shape[b]}
        return out
