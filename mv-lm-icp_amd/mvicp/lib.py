"""ctypes binding of include/mvicp.h.  No compute happens here; every call goes through the C ABI."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVICP_LIB") or os.path.join(os.path.dirname(_HERE), "libmvicp_hip.so")   # MVICP_LIB: tuning builds only

PARAM_EIGEN_QUATERNION, PARAM_ANGLE_AXIS, PARAM_SOPHUS_SE3 = 0, 1, 2
NN_AUTO, NN_BRUTE, NN_GRID, NN_TILE = 0, 1, 2, 3
EDGE_BLOCK = 91

SYMBOLS = [
    "mvicp_last_error", "mvicp_version", "mvicp_create", "mvicp_destroy", "mvicp_set_num_frames", "mvicp_set_frame",
    "mvicp_recompute_normals", "mvicp_set_graph", "mvicp_set_shard", "mvicp_edge_owner", "mvicp_comm_unique_id", "mvicp_comm_init", "mvicp_comm_nranks", "mvicp_comm_set_callback", "mvicp_correspond",
    "mvicp_get_correspondences", "mvicp_map_correspondences", "mvicp_map_correspondences_async", "mvicp_wait_correspondences", "mvicp_correspondence_epochs", "mvicp_set_correspondences", "mvicp_nn_query", "mvicp_linearize", "mvicp_optimize",
    "mvicp_lm_solve", "mvicp_set_option", "mvicp_nn_census", "mvicp_nn_census_ex", "mvicp_reset_history", "mvicp_profile_enable", "mvicp_profile_reset", "mvicp_profile_get", "mvicp_profile_get_ex", "mvicp_stream", "mvicp_sync",
    "mvicp_closedform_point_to_point", "mvicp_closedform_point_to_plane",
]


class MvicpError(RuntimeError):
    pass


class Summary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int), ("successful_steps", C.c_int),
                ("termination", C.c_int), ("evaluations", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))

_lib = None


def load_library(path=None):
    """Load libmvicp_hip.so (fails loudly if it has not been built: there is no Python/CPU fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MvicpError(f"{p} not found: build it with `make -C mv-lm-icp_amd` (or __graft_entry__.build())")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    vp, ip, dp, fp, u8p = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_ubyte)
    lib.mvicp_last_error.restype = C.c_char_p
    lib.mvicp_version.restype = C.c_char_p
    lib.mvicp_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mvicp_destroy.argtypes = [vp]
    lib.mvicp_set_num_frames.argtypes = [vp, C.c_int]
    lib.mvicp_set_frame.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.mvicp_recompute_normals.argtypes = [vp, C.c_int, C.c_int, dp, ip]
    lib.mvicp_set_graph.argtypes = [vp, C.c_int, ip, ip]
    lib.mvicp_set_shard.argtypes = [vp, C.c_int, C.c_int]
    lib.mvicp_edge_owner.argtypes = [C.c_int, ip, C.c_int, ip]
    lib.mvicp_comm_unique_id.argtypes = [C.c_char_p, vp]
    lib.mvicp_comm_init.argtypes = [vp, C.c_char_p, vp, C.c_int, C.c_int]
    lib.mvicp_comm_set_callback.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.mvicp_comm_nranks.argtypes = [vp]
    lib.mvicp_correspond.argtypes = [vp, dp, u8p, C.c_float, C.c_int, ip, fp]
    lib.mvicp_get_correspondences.argtypes = [vp, C.c_int, C.c_int, ip, ip, dp]
    lib.mvicp_map_correspondences.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(C.c_longlong))]
    lib.mvicp_map_correspondences_async.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(C.c_longlong))]
    lib.mvicp_wait_correspondences.argtypes = [vp, C.c_int]
    lib.mvicp_correspondence_epochs.argtypes = [vp, C.POINTER(C.POINTER(C.c_ulonglong))]
    lib.mvicp_set_correspondences.argtypes = [vp, C.c_int, C.c_int, ip, ip, C.c_float]
    lib.mvicp_nn_query.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, ip, dp]
    lib.mvicp_linearize.argtypes = [vp, dp, C.c_int, C.c_int, dp]
    lib.mvicp_optimize.argtypes = [vp, dp, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Summary)]
    lib.mvicp_lm_solve.argtypes = [C.c_int, C.c_int, ip, ip, dp, u8p, C.c_int, C.c_int, EVAL_FN, vp, C.POINTER(Summary)]
    lib.mvicp_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    lib.mvicp_nn_census.argtypes = [vp, dp]
    lib.mvicp_nn_census_ex.argtypes = [vp, dp, C.c_int]
    lib.mvicp_reset_history.argtypes = [vp]
    lib.mvicp_profile_enable.argtypes = [vp, C.c_int]
    lib.mvicp_profile_reset.argtypes = [vp]
    lib.mvicp_profile_get.argtypes = [vp, C.c_char_p, dp, C.POINTER(C.c_longlong), dp]
    lib.mvicp_profile_get_ex.argtypes = [vp, C.c_char_p, dp, C.c_int]
    lib.mvicp_stream.argtypes = [vp]
    lib.mvicp_stream.restype = vp
    lib.mvicp_sync.argtypes = [vp]
    lib.mvicp_closedform_point_to_point.argtypes = [dp, dp, C.c_int, dp]
    lib.mvicp_closedform_point_to_plane.argtypes = [dp, dp, dp, C.c_int, dp]
    if path is None:
        _lib = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _check(lib, st):
    if st < 0:
        raise MvicpError(f"mvicp status {st}: {lib.mvicp_last_error().decode()}")
    return st


def poses_to_c(poses):
    """(K,4,4) row-major numpy matrices -> K x 16 column-major doubles (Eigen Isometry3d::data())."""
    P = np.asarray(poses, dtype=np.float64)
    return np.ascontiguousarray(np.transpose(P, (0, 2, 1)).reshape(len(P), 16))


def poses_from_c(buf):
    return np.ascontiguousarray(np.transpose(np.asarray(buf, dtype=np.float64).reshape(-1, 4, 4), (0, 2, 1)))


def edge_owner(n_src, world):
    lib = load_library()
    n_src = np.ascontiguousarray(n_src, dtype=np.int32)
    owner = np.zeros(len(n_src), dtype=np.int32)
    _check(lib, lib.mvicp_edge_owner(len(n_src), _ip(n_src), world, _ip(owner)))
    return owner


def closedform_point_to_point(src, dst):
    """ICP_Closedform::pointToPoint (icp-closedform.cpp:9-26): least-squares rigid transform src -> dst.  Host only."""
    lib = load_library()
    a = np.ascontiguousarray(src, dtype=np.float64); b = np.ascontiguousarray(dst, dtype=np.float64)
    out = np.zeros(16)
    _check(lib, lib.mvicp_closedform_point_to_point(_dp(a), _dp(b), len(a), _dp(out)))
    return poses_from_c(out)[0]


def closedform_point_to_plane(src, dst, nor):
    """ICP_Closedform::pointToPlane (icp-closedform.cpp:30-54): one linearised point-to-plane step from identity.  Host only."""
    lib = load_library()
    a = np.ascontiguousarray(src, dtype=np.float64); b = np.ascontiguousarray(dst, dtype=np.float64); c = np.ascontiguousarray(nor, dtype=np.float64)
    out = np.zeros(16)
    _check(lib, lib.mvicp_closedform_point_to_plane(_dp(a), _dp(b), _dp(c), len(a), _dp(out)))
    return poses_from_c(out)[0]


def lm_solve_host(n_frames, src, dst, poses, fixed, param, eval_callback, max_iterations=50):
    """mvicp_lm_solve with a Python evaluator: eval_callback(poses(K,4,4)) -> blocks (E,91).  Host only."""
    lib = load_library()
    src = np.ascontiguousarray(src, dtype=np.int32)
    dst = np.ascontiguousarray(dst, dtype=np.int32)
    E = len(src)
    P = poses_to_c(poses)
    fx = np.ascontiguousarray(fixed, dtype=np.uint8).copy()
    err = []

    def _cb(_user, p_ptr, b_ptr):
        try:
            pp = np.ctypeslib.as_array(p_ptr, shape=(n_frames, 16)).copy()
            blocks = np.asarray(eval_callback(poses_from_c(pp)), dtype=np.float64).reshape(E, EDGE_BLOCK)
            np.ctypeslib.as_array(b_ptr, shape=(E, EDGE_BLOCK))[:] = blocks
            return 0
        except Exception as ex:  # pragma: no cover - surfaced below
            err.append(ex)
            return -1

    sm = Summary()
    cb = EVAL_FN(_cb)
    st = lib.mvicp_lm_solve(n_frames, E, _ip(src), _ip(dst), _dp(P), fx.ctypes.data_as(C.POINTER(C.c_ubyte)), param, max_iterations, cb, None, C.byref(sm))
    if err:
        raise err[0]
    _check(lib, st)
    return poses_from_c(P), sm.as_dict()


class Engine:
    """One GPU context.  Mirrors the reference loop: set_frames -> set_graph -> (correspond -> optimize)*."""

    def __init__(self, device=0, rank=0, world=1):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.mvicp_create(device, C.byref(h)))
        self.h = h
        self.n_frames = 0
        self.E = 0
        self.rank, self.world = rank, world
        if world > 1:
            _check(self.lib, self.lib.mvicp_set_shard(self.h, rank, world))

    def close(self):
        if self.h:
            self.lib.mvicp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- uploads
    def set_frames(self, pts_list, nor_list=None):
        self.n_frames = len(pts_list)
        _check(self.lib, self.lib.mvicp_set_num_frames(self.h, self.n_frames))
        self.npts = []
        for i, p in enumerate(pts_list):
            p = np.ascontiguousarray(p, dtype=np.float64)
            n = None if nor_list is None or nor_list[i] is None else np.ascontiguousarray(nor_list[i], dtype=np.float64)
            _check(self.lib, self.lib.mvicp_set_frame(self.h, i, _dp(p), _dp(n) if n is not None else None, len(p)))
            self.npts.append(len(p))

    def set_frame(self, frame, pts, nor=None):
        """Re-upload ONE cloud (before set_graph): mvicp_set_frame."""
        p = np.ascontiguousarray(pts, dtype=np.float64)
        n = None if nor is None else np.ascontiguousarray(nor, dtype=np.float64)
        _check(self.lib, self.lib.mvicp_set_frame(self.h, frame, _dp(p), _dp(n) if n is not None else None, len(p)))
        self.npts[frame] = len(p)

    def recompute_normals(self, frame, k=10, want_knn=False):
        n = self.npts[frame]
        nrm = np.zeros((n, 3), dtype=np.float64)
        knn = np.zeros((n, k), dtype=np.int32) if want_knn else None
        _check(self.lib, self.lib.mvicp_recompute_normals(self.h, frame, k, _dp(nrm), _ip(knn) if want_knn else None))
        return (nrm, knn) if want_knn else nrm

    def set_graph(self, src, dst):
        self.src = np.ascontiguousarray(src, dtype=np.int32)
        self.dst = np.ascontiguousarray(dst, dtype=np.int32)
        self.E = len(self.src)
        _check(self.lib, self.lib.mvicp_set_graph(self.h, self.E, _ip(self.src), _ip(self.dst)))

    def reset_history(self):
        """New registration on the same clouds / graph: the next correspond() behaves like the first one after set_graph()."""
        _check(self.lib, self.lib.mvicp_reset_history(self.h))

    def comm_init(self, unique_id, librccl_path=None):
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        path = librccl_path.encode() if librccl_path else None
        _check(self.lib, self.lib.mvicp_comm_init(self.h, path, C.cast(buf, C.c_void_p), self.rank, self.world))

    def comm_nranks(self):
        """ranks of the RCCL communicator as RCCL reports them (0: none)."""
        return int(self.lib.mvicp_comm_nranks(self.h))

    def comm_set_callback(self, allreduce):
        """allreduce(numpy float64 array) must sum it in place over all ranks (host-staged exchange, no RCCL)."""
        def _cb(_user, ptr, n):
            try:
                allreduce(np.ctypeslib.as_array(ptr, shape=(n,)))
                return 0
            except Exception:
                return -1
        self._ar_cb = ALLREDUCE_FN(_cb)   # keep alive
        _check(self.lib, self.lib.mvicp_comm_set_callback(self.h, self._ar_cb, None))

    @staticmethod
    def comm_unique_id(librccl_path=None):
        lib = load_library()
        buf = (C.c_char * 128)()
        path = librccl_path.encode() if librccl_path else None
        _check(lib, lib.mvicp_comm_unique_id(path, C.cast(buf, C.c_void_p)))
        return bytes(buf)

    # ---- S1
    def _round_buffers(self, K):
        """Per-engine staging buffers + their ctypes pointers for the two per-round calls (building numpy arrays and ctypes pointers
        anew costs ~10 us per call — a few per cent of a converged cfg4 round).  Results are handed back as copies."""
        b = getattr(self, "_rb", None)
        if b is None or b["K"] != K or b["E"] != self.E:
            P = np.zeros((K, 16)); fx = np.zeros(K, dtype=np.uint8)
            counts = np.zeros(self.E, dtype=np.int32); weights = np.zeros(self.E, dtype=np.float32)
            sm = Summary()
            b = {"K": K, "E": self.E, "P": P, "P44": P.reshape(K, 4, 4), "fx": fx, "counts": counts, "weights": weights, "sm": sm,
                 "pP": _dp(P), "pfx": fx.ctypes.data_as(C.POINTER(C.c_ubyte)), "pc": _ip(counts), "pw": weights.ctypes.data_as(C.POINTER(C.c_float)),
                 "psm": C.byref(sm)}
            self._rb = b
        return b

    def correspond(self, poses, fixed, thresh, nn_method=NN_AUTO):
        b = self._round_buffers(len(poses))
        np.copyto(b["P44"], np.transpose(np.asarray(poses, dtype=np.float64), (0, 2, 1)))   # 4x4 row-major -> column-major (Eigen)
        b["fx"][:] = fixed
        _check(self.lib, self.lib.mvicp_correspond(self.h, b["pP"], b["pfx"], np.float32(thresh), nn_method, b["pc"], b["pw"]))
        self.counts = b["counts"].copy()
        return self.counts, b["weights"].copy()

    # ---- the per-round calls without the (K,4,4) <-> column-major conversions: the caller keeps the poses in the engine's own K x 16
    # buffer (Eigen's layout) between the calls, like a C++ driver does.  bench.py's timed loop uses these.
    def round_state(self, K, fixed):
        """-> dict with "P" (K x 16 column-major poses, in/out), "counts", "weights", "sm" (the Summary struct of the last optimize_raw)."""
        b = self._round_buffers(K)
        b["fx"][:] = fixed
        return b

    def correspond_raw(self, thresh, nn_method=NN_AUTO):
        b = self._rb
        _check(self.lib, self.lib.mvicp_correspond(self.h, b["pP"], b["pfx"], thresh, nn_method, b["pc"], b["pw"]))

    def optimize_raw(self, param, point_to_plane, robust, max_iterations=50):
        b = self._rb
        _check(self.lib, self.lib.mvicp_optimize(self.h, b["pP"], b["pfx"], param, int(point_to_plane), int(robust), max_iterations, b["psm"]))
        return b["sm"]

    def get_correspondences(self, edge):
        cap = self.npts[self.src[edge]]
        first = np.zeros(cap, dtype=np.int32)
        second = np.zeros(cap, dtype=np.int32)
        dist = np.zeros(cap, dtype=np.float64)
        n = _check(self.lib, self.lib.mvicp_get_correspondences(self.h, edge, cap, _ip(first), _ip(second), _dp(dist)))
        return first[:n].copy(), second[:n].copy(), dist[:n].copy()

    CORR_DTYPE = np.dtype([("first", np.int32), ("second", np.int32), ("dist", np.float64)])   # struct Correspondance (include/frame.h:18-22)

    def map_correspondences(self, copy=True):
        """ALL lists of the last correspond() in the reference's layout: (triples, offsets) — a structured array of {first, second, dist}
        and E + 1 positions; edge e = triples[offsets[e]:offsets[e + 1]], ascending `first`.  copy=False returns a VIEW of the library's
        pinned buffer, valid until the next correspond() / set_correspondences() on this engine."""
        tp, op = C.c_void_p(), C.POINTER(C.c_longlong)()
        _check(self.lib, self.lib.mvicp_map_correspondences(self.h, C.byref(tp), C.byref(op)))
        off = np.ctypeslib.as_array(op, shape=(self.E + 1,)).copy()
        total = int(off[-1])
        if total == 0:
            return np.zeros(0, dtype=self.CORR_DTYPE), off
        buf = (C.c_char * (16 * total)).from_address(tp.value)
        t = np.frombuffer(buf, dtype=self.CORR_DTYPE, count=total)
        return (t.copy() if copy else t), off

    def map_correspondences_async(self):
        """mvicp_map_correspondences_async: VIEWS (triples, offsets) of the library's pinned buffer; edge e's bytes are defined after wait_correspondences(e)."""
        tp, op = C.c_void_p(), C.POINTER(C.c_longlong)()
        _check(self.lib, self.lib.mvicp_map_correspondences_async(self.h, C.byref(tp), C.byref(op)))
        off = np.ctypeslib.as_array(op, shape=(self.E + 1,)).copy()
        total = int(off[-1])
        if total == 0:
            return np.zeros(0, dtype=self.CORR_DTYPE), off
        buf = (C.c_char * (16 * total)).from_address(tp.value)
        return np.frombuffer(buf, dtype=self.CORR_DTYPE, count=total), off

    def wait_correspondences(self, edge):
        _check(self.lib, self.lib.mvicp_wait_correspondences(self.h, int(edge)))

    def correspondence_epochs(self):
        """Per-edge change counters of the lists (mvicp_correspondence_epochs): an edge whose list is provably last search's keeps its epoch."""
        ep = C.POINTER(C.c_ulonglong)()
        _check(self.lib, self.lib.mvicp_correspondence_epochs(self.h, C.byref(ep)))
        return np.ctypeslib.as_array(ep, shape=(self.E,)).copy()

    def set_correspondences(self, edge, first, second, weight=0.0):
        first = np.ascontiguousarray(first, dtype=np.int32)
        second = np.ascontiguousarray(second, dtype=np.int32)
        _check(self.lib, self.lib.mvicp_set_correspondences(self.h, edge, len(first), _ip(first), _ip(second), np.float32(weight)))

    def nn_query(self, frame, queries, nn_method=NN_AUTO):
        q = np.ascontiguousarray(queries, dtype=np.float64)
        idx = np.zeros(len(q), dtype=np.int32)
        d2 = np.zeros(len(q), dtype=np.float64)
        _check(self.lib, self.lib.mvicp_nn_query(self.h, frame, _dp(q), len(q), nn_method, _ip(idx), _dp(d2)))
        return idx, d2

    # ---- normal equations / S2
    def linearize(self, poses, point_to_plane, robust):
        P = poses_to_c(poses)
        out = np.zeros((self.E, EDGE_BLOCK), dtype=np.float64)
        _check(self.lib, self.lib.mvicp_linearize(self.h, _dp(P), int(point_to_plane), int(robust), _dp(out)))
        return out

    def optimize(self, poses, fixed, param=PARAM_SOPHUS_SE3, point_to_plane=True, robust=True, max_iterations=50):
        b = self._round_buffers(len(poses))
        np.copyto(b["P44"], np.transpose(np.asarray(poses, dtype=np.float64), (0, 2, 1)))
        b["fx"][:] = fixed          # (the solver forces fixed[0] = 1 in this private copy, like icp-ceres.cpp:244,341,417)
        _check(self.lib, self.lib.mvicp_optimize(self.h, b["pP"], b["pfx"], param, int(point_to_plane), int(robust), max_iterations, b["psm"]))
        return np.ascontiguousarray(np.transpose(b["P44"], (0, 2, 1))), b["sm"].as_dict()

    def set_option(self, name, value):
        _check(self.lib, self.lib.mvicp_set_option(self.h, name.encode(), float(value)))

    def nn_census(self):
        out = np.zeros(10)
        _check(self.lib, self.lib.mvicp_nn_census_ex(self.h, _dp(out), 10))
        return {"queries": out[0], "candidates": out[1], "nodes": out[2], "far": out[3], "hits": out[4], "fetched": out[5],
                "rescreens": out[6], "confirm_rounds": out[7], "blocks": out[8], "confirmations": out[9]}

    # ---- profiling
    def profile(self, on=True):
        """on: False/0 off, True/1 every scope, 2 only the roofline scopes "nn" and "linearize"."""
        _check(self.lib, self.lib.mvicp_profile_enable(self.h, int(on)))

    def profile_reset(self):
        _check(self.lib, self.lib.mvicp_profile_reset(self.h))

    def profile_get(self, kernel):
        ms, n, b = C.c_double(), C.c_longlong(), C.c_double()
        _check(self.lib, self.lib.mvicp_profile_get(self.h, kernel.encode(), C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value

    def profile_get_ex(self, kernel):
        """{"ms", "launches", "model_bytes", "survey_bytes", "queries"} of a scope ("nn" = every NN kernel together)."""
        out = np.zeros(5)
        _check(self.lib, self.lib.mvicp_profile_get_ex(self.h, kernel.encode(), _dp(out), 5))
        return {"ms": float(out[0]), "launches": int(out[1]), "model_bytes": float(out[2]), "survey_bytes": float(out[3]), "queries": float(out[4])}

    def sync(self):
        _check(self.lib, self.lib.mvicp_sync(self.h))


def unpack_block(b):
    """91 -> (H 12x12 symmetric, g 12, cost)."""
    H = np.zeros((12, 12))
    iu = np.triu_indices(12)
    H[iu] = b[:78]
    H = H + np.triu(H, 1).T
    return H, np.array(b[78:90]), float(b[90])
