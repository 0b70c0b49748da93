"""ctypes access to the CPU oracle (oracle/) for tests, smoke() and bench.py's cpu_baseline leg ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "_build", "liborc.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_nanoflann.so")

PARAM_QUAT, PARAM_ANGLEAXIS, PARAM_SOPHUS = 0, 1, 2


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


class Problem(C.Structure):
    _fields_ = [("K", C.c_int), ("pts", C.c_void_p), ("nor", C.c_void_p), ("foff", C.c_void_p), ("fixed", C.c_void_p),
                ("E", C.c_int), ("esrc", C.c_void_p), ("edst", C.c_void_p), ("eoff", C.c_void_p), ("first", C.c_void_p),
                ("second", C.c_void_p), ("eweight", C.c_void_p), ("param", C.c_int), ("plane", C.c_int), ("robust", C.c_int)]


class Summary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int), ("successful_steps", C.c_int),
                ("termination", C.c_int), ("jacobian_evals", C.c_int), ("cost_evals", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def to_c(poses):
    P = np.asarray(poses, dtype=np.float64)
    return np.ascontiguousarray(np.transpose(P, (0, 2, 1)).reshape(len(P), 16))


def from_c(buf):
    return np.ascontiguousarray(np.transpose(np.asarray(buf).reshape(-1, 4, 4), (0, 2, 1)))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.orc_evaluate.restype = C.c_double
        lib.orc_evaluate_x.restype = C.c_double

    # ---- (a)
    def query_transform(self, pose_src, pose_dst, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        q = np.empty_like(p)
        ps, pd = to_c([pose_src])[0], to_c([pose_dst])[0]
        self.lib.orc_query_transform(_p(ps), _p(pd), _p(p), C.c_int(len(p)), _p(q))
        return q

    def nn_brute(self, dst, queries):
        dst = np.ascontiguousarray(dst, dtype=np.float64)
        q = np.ascontiguousarray(queries, dtype=np.float64)
        idx = np.empty(len(q), dtype=np.int32)
        d2 = np.empty(len(q), dtype=np.float64)
        self.lib.orc_nn_brute(_p(dst), C.c_int(len(dst)), _p(q), C.c_int(len(q)), _p(idx), _p(d2))
        return idx, d2

    def filter_median(self, idx, d2, thresh):
        n = len(idx)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        d2 = np.ascontiguousarray(d2, dtype=np.float64)
        first = np.empty(n, dtype=np.int32); second = np.empty(n, dtype=np.int32); dist = np.empty(n, dtype=np.float64)
        w = C.c_float(0)
        c = self.lib.orc_filter_median(_p(idx), _p(d2), C.c_int(n), C.c_float(thresh), _p(first), _p(second), _p(dist), C.byref(w))
        return first[:c].copy(), second[:c].copy(), dist[:c].copy(), np.float32(w.value)

    def correspond_edge(self, src, pose_src, dst, pose_dst, thresh):
        src = np.ascontiguousarray(src, dtype=np.float64); dst = np.ascontiguousarray(dst, dtype=np.float64)
        n = len(src)
        first = np.empty(n, dtype=np.int32); second = np.empty(n, dtype=np.int32); dist = np.empty(n, dtype=np.float64)
        nn_idx = np.empty(n, dtype=np.int32); nn_d2 = np.empty(n, dtype=np.float64)
        w = C.c_float(0)
        ps, pd = to_c([pose_src])[0], to_c([pose_dst])[0]
        c = self.lib.orc_correspond_edge(_p(src), C.c_int(n), _p(ps), _p(dst), C.c_int(len(dst)), _p(pd), C.c_float(thresh), _p(first), _p(second),
                                         _p(dist), C.byref(w), _p(nn_idx), _p(nn_d2))
        return first[:c].copy(), second[:c].copy(), dist[:c].copy(), np.float32(w.value), nn_idx, nn_d2

    # ---- (b)
    def make_problem(self, pts, nor, fixed, src, dst, corr, weights, param, plane, robust):
        """corr: list over edges of (first, second).  Keeps the numpy buffers alive on the returned object."""
        K = len(pts)
        keep = {}
        keep["pts"] = np.ascontiguousarray(np.concatenate(pts), dtype=np.float64)
        keep["nor"] = np.ascontiguousarray(np.concatenate([n if n is not None else np.zeros_like(p) for p, n in zip(pts, nor)]), dtype=np.float64)
        keep["foff"] = np.concatenate([[0], np.cumsum([len(p) for p in pts])]).astype(np.int32)
        keep["fixed"] = np.ascontiguousarray(fixed, dtype=np.uint8)
        keep["esrc"] = np.ascontiguousarray(src, dtype=np.int32)
        keep["edst"] = np.ascontiguousarray(dst, dtype=np.int32)
        keep["eoff"] = np.concatenate([[0], np.cumsum([len(c[0]) for c in corr])]).astype(np.int32)
        keep["first"] = np.ascontiguousarray(np.concatenate([c[0] for c in corr]) if corr else np.zeros(0), dtype=np.int32)
        keep["second"] = np.ascontiguousarray(np.concatenate([c[1] for c in corr]) if corr else np.zeros(0), dtype=np.int32)
        keep["w"] = np.ascontiguousarray(weights, dtype=np.float32)
        pb = Problem(K, _p(keep["pts"]), _p(keep["nor"]), _p(keep["foff"]), _p(keep["fixed"]), len(src), _p(keep["esrc"]), _p(keep["edst"]),
                     _p(keep["eoff"]), _p(keep["first"]), _p(keep["second"]), _p(keep["w"]), param, int(plane), int(robust))
        pb._keep = keep
        return pb

    def nfree(self, pb):
        return int((pb._keep["fixed"] == 0).sum())

    def evaluate(self, pb, poses, jac=True):
        n = 6 * self.nfree(pb)
        P = to_c(poses)
        if not jac:
            return self.lib.orc_evaluate(C.byref(pb), _p(P), None, None), None, None
        H = np.zeros((n, n)); g = np.zeros(n)
        cost = self.lib.orc_evaluate(C.byref(pb), _p(P), _p(H), _p(g))
        return cost, H, g

    def evaluate_x(self, pb, x, jac=True):
        n = 6 * self.nfree(pb)
        x = np.ascontiguousarray(x, dtype=np.float64)
        if not jac:
            return self.lib.orc_evaluate_x(C.byref(pb), _p(x), None, None), None, None
        H = np.zeros((n, n)); g = np.zeros(n)
        cost = self.lib.orc_evaluate_x(C.byref(pb), _p(x), _p(H), _p(g))
        return cost, H, g

    def optimize(self, pb, poses, max_iterations=50):
        P = to_c(poses)
        sm = Summary()
        self.lib.orc_optimize(C.byref(pb), _p(P), C.c_int(max_iterations), C.byref(sm))
        return from_c(P), sm.as_dict()

    def ambient(self, param):
        return self.lib.orc_ambient(param)

    LM_OPTION_ORDER = ("initial_radius", "min_relative_decrease", "function_tolerance", "parameter_tolerance", "jacobi_scaling", "radius_rule",
                       "legacy_minimizer", "min_diag", "gradient_tolerance")
    LM_DEFAULTS = {"initial_radius": 1e4, "min_relative_decrease": 1e-3, "function_tolerance": 1e-6, "parameter_tolerance": 1e-8, "jacobi_scaling": 1,
                   "radius_rule": 0, "legacy_minimizer": 0, "min_diag": 1e-6, "gradient_tolerance": 1e-10}

    def set_lm_options(self, **kw):
        """Perturb the oracle's trust-region schedule (sensitivity tests only); no arguments = Ceres defaults."""
        unknown = set(kw) - set(self.LM_OPTION_ORDER)
        assert not unknown, unknown
        if not kw:
            self.lib.orc_set_lm_options(None, C.c_int(0))
            return
        v = np.array([float(kw.get(k, self.LM_DEFAULTS[k])) for k in self.LM_OPTION_ORDER], dtype=np.float64)
        self.lib.orc_set_lm_options(_p(v), C.c_int(len(v)))

    def pose_to_param(self, param, pose):
        x = np.zeros(self.ambient(param))
        P = to_c([pose])[0]
        self.lib.orc_pose_to_param(param, _p(P), _p(x))
        return x

    def param_to_pose(self, param, x):
        P = np.zeros(16)
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.lib.orc_param_to_pose(param, _p(x), _p(P))
        return from_c(P)[0]

    def local_plus(self, param, x, d):
        out = np.zeros(self.ambient(param))
        x = np.ascontiguousarray(x, dtype=np.float64); d = np.ascontiguousarray(d, dtype=np.float64)
        self.lib.orc_local_plus(param, _p(x), _p(d), _p(out))
        return out

    def add_noise(self, pose, sigma, sigmat, reset=False, stdlib="libstdc++"):
        """common.h:36-67 with the reference's default-seeded std::mt19937 stream.  stdlib picks std::normal_distribution's
        implementation-defined variate order: "libstdc++" (g++) or "libc++" (clang / OS X: the stream behind README.md:141-146)."""
        out = np.zeros(16)
        self.lib.orc_add_noise_stream(_p(to_c([pose])[0]), C.c_double(sigma), C.c_double(sigmat), C.c_int(1 if reset else 0),
                                      C.c_int({"libstdc++": 0, "libc++": 1}[stdlib]), _p(out))
        return from_c(out)[0]

    def pose_diff(self, P1, P2):
        a, b = C.c_double(), C.c_double()
        self.lib.orc_pose_diff(_p(to_c([P1])[0]), _p(to_c([P2])[0]), C.byref(a), C.byref(b))
        return a.value, b.value

    def edge_blocks(self, pts, nor, src, dst, corr, weights, poses, plane, robust):
        """Per-edge canonical 91-blocks [78 upper H | 12 g | cost]: a 2-frame SOPHUS problem per edge with both poses
        free — the SophusSE3 local coordinates (T <- T exp(delta), delta = (upsilon, omega)) ARE the canonical ones."""
        out = np.zeros((len(src), 91))
        iu = np.triu_indices(12)
        for e, (s, d) in enumerate(zip(src, dst)):
            pb = self.make_problem([pts[s], pts[d]], [nor[s] if nor else None, nor[d] if nor else None], [0, 0], [0], [1], [corr[e]], [weights[e]],
                                   PARAM_SOPHUS, plane, robust)
            cost, H, g = self.evaluate(pb, [poses[s], poses[d]])
            out[e, :78] = H[iu]; out[e, 78:90] = g; out[e, 90] = cost
        return out


class RefNN:
    """The real nanoflann (reference NN), src/internal/frame.cpp:187-206 semantics."""

    def __init__(self, lib):
        self.lib = lib
        lib.ref_nn_build.restype = C.c_void_p

    def query(self, dst, queries):
        dst = np.ascontiguousarray(dst, dtype=np.float64); q = np.ascontiguousarray(queries, dtype=np.float64)
        h = C.c_void_p(self.lib.ref_nn_build(_p(dst), C.c_int(len(dst))))
        idx = np.empty(len(q), dtype=np.int32); d2 = np.empty(len(q), dtype=np.float64)
        self.lib.ref_nn_query(h, _p(q), C.c_int(len(q)), _p(idx), _p(d2))
        self.lib.ref_nn_free(h)
        return idx, d2

    def knn_self(self, pts, k):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        h = C.c_void_p(self.lib.ref_nn_build(_p(pts), C.c_int(len(pts))))
        idx = np.empty((len(pts), k), dtype=np.int32); d2 = np.empty((len(pts), k), dtype=np.float64)
        ri = np.empty(k, dtype=np.int32); rd = np.empty(k, dtype=np.float64)
        for i in range(len(pts)):
            self.lib.ref_nn_knn_self(h, C.c_int(i), C.c_int(k), _p(ri), _p(rd))
            idx[i] = ri; d2[i] = rd
        self.lib.ref_nn_free(h)
        return idx, d2


_orc = None
_ref = None


def load():
    global _orc
    if _orc is None:
        if not os.path.exists(ORC_SO):
            build()
        _orc = Oracle(C.CDLL(ORC_SO))
    return _orc


def load_ref():
    global _ref
    if _ref is None:
        if not os.path.exists(REF_SO):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(REF_SO):
            return None
        _ref = RefNN(C.CDLL(REF_SO))
    return _ref
