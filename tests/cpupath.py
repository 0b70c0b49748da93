"""The reference-equivalent CPU path of one ICP round — TEST INFRASTRUCTURE (tests/, smoke(), bench.py's cpu_baseline leg only).

One round = the loop body of src/main_multiview.cpp:150-169 without visualisation:
  * Frame::computeClosestPointsToNeighbours for every non-fixed source (src/internal/frame.cpp:91-185): the oracle's query transform
    (frame.cpp:117-118,131,136), exact 1-NN through the REAL vendored nanoflann (oracle/_ref, one KD-tree per destination cloud,
    built once like the reference's lazy index, frame.cpp:188-193) — or the oracle's brute force when oracle/_ref was never
    built — then the cutoff / upper-median rule (frame.cpp:156-176);
  * ICP_Ceres::ceresOptimizer* (src/internal/icp-ceres.cpp:220-475): the oracle's Jet restatement + Ceres-style LM.

`fast=True` loads the -O3 AVX2 + OpenMP builds (oracle/Makefile FASTFLAGS: same arithmetic with -ffp-contract=off, but the
per-edge sums are accumulated per thread, so results differ from the -O2 build in the last bits)."""
import ctypes as C
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import orclib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_features():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def usable_cores(cap=64):
    """Cores this process may really use: affinity mask and cgroup CPU quota (a container often shows every host core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def fast_build_usable():
    return {"avx2", "fma", "bmi2"} <= cpu_features() and os.path.exists(os.path.join(ROOT, "oracle", "_build", "liborc_fast.so"))


def load_libs(fast):
    """-> (Oracle, RefNN or None) of the requested build."""
    suf = "_fast" if fast else ""
    orc_so = os.path.join(ROOT, "oracle", "_build", f"liborc{suf}.so")
    ref_so = os.path.join(ROOT, "oracle", "_ref", f"libref_nanoflann{suf}.so")
    if not fast:
        return orclib.load(), orclib.load_ref()
    orc = orclib.Oracle(C.CDLL(orc_so))
    ref = orclib.RefNN(C.CDLL(ref_so)) if os.path.exists(ref_so) else None
    return orc, ref


def set_omp_threads(n):
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(int(n)))
        return True
    except OSError:
        return False


class CpuPath:
    """Walks a registration on the CPU.  pts / nor: per-view arrays; src / dst: the pose graph; param / plane as mvicp_optimize."""

    def __init__(self, pts, nor, src, dst, fixed, param, plane, robust=True, cutoff=0.05, fast=False, threads=1, orc=None, ref=None):
        if orc is None:
            orc, ref = load_libs(fast)
        self.orc, self.ref = orc, ref
        self.pts = [np.ascontiguousarray(p, dtype=np.float64) for p in pts]
        self.nor = [None if n is None else np.ascontiguousarray(n, dtype=np.float64) for n in nor]
        self.src = np.asarray(src, dtype=np.int32); self.dst = np.asarray(dst, dtype=np.int32)
        self.fixed = np.asarray(fixed, dtype=np.uint8).copy()
        self.param, self.plane, self.robust, self.cutoff = int(param), int(plane), int(bool(robust)), float(cutoff)
        # fast build: OpenMP inside the oracle / nanoflann calls (threads = OMP threads, edges one after another);
        # -O2 build: no OpenMP, so `threads` > 1 only spreads the NN of different edges over a thread pool (the LM stays serial)
        self.fast = bool(fast)
        self.threads = max(1, int(threads))
        self.pool = 1 if self.fast else self.threads
        if self.fast:
            set_omp_threads(self.threads)
        self.trees = {}
        self.tree_build_s = 0.0
        self.last = {}

    def _tree(self, d):
        if d not in self.trees:
            t0 = time.perf_counter()
            p = self.pts[d]
            self.trees[d] = C.c_void_p(self.ref.lib.ref_nn_build(p.ctypes.data_as(C.c_void_p), C.c_int(len(p))))
            self.tree_build_s += time.perf_counter() - t0
        return self.trees[d]

    def close(self):
        if self.ref is not None:
            for h in self.trees.values():
                self.ref.lib.ref_nn_free(h)
        self.trees = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _edge(self, e, poses):
        s, d = int(self.src[e]), int(self.dst[e])
        if self.fixed[s]:                                  # frame.cpp:93: a fixed frame never searches
            return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), np.float32(0)
        if self.ref is None:
            f, sec, dist, w, _, _ = self.orc.correspond_edge(self.pts[s], poses[s], self.pts[d], poses[d], self.cutoff)
            return f, sec, dist, w
        q = self.orc.query_transform(poses[s], poses[d], self.pts[s])
        idx = np.empty(len(q), dtype=np.int32); d2 = np.empty(len(q), dtype=np.float64)
        self.ref.lib.ref_nn_query(self.trees[d], q.ctypes.data_as(C.c_void_p), C.c_int(len(q)), idx.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p))
        return self.orc.filter_median(idx, d2, self.cutoff)

    def correspond(self, poses):
        """-> list over edges of (first, second, dist, weight)."""
        if self.ref is not None:
            for d in sorted(set(self.dst.tolist())):
                self._tree(d)
        E = len(self.src)
        if self.pool > 1:
            with ThreadPoolExecutor(max_workers=self.pool) as ex:     # the ctypes calls release the GIL
                return list(ex.map(lambda e: self._edge(e, poses), range(E)))
        return [self._edge(e, poses) for e in range(E)]

    def optimize(self, poses, corr, max_iterations=50):
        fx = self.fixed.copy(); fx[0] = 1                                  # icp-ceres.cpp:244,341,417
        prob = self.orc.make_problem(self.pts, self.nor, fx, self.src, self.dst, [c[:2] for c in corr], [c[3] for c in corr], self.param, self.plane, self.robust)
        return self.orc.optimize(prob, poses, max_iterations)

    def round(self, poses):
        """One ICP round from `poses` (K,4,4) -> (new poses, LM summary).  Timings / lists of the round are left in self.last."""
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        tb = self.tree_build_s
        t0 = time.perf_counter()
        corr = self.correspond(poses)
        t1 = time.perf_counter()
        new, sm = self.optimize(poses, corr)
        t2 = time.perf_counter()
        built = self.tree_build_s - tb
        self.last = {"corr": corr, "nn_s": t1 - t0 - built, "lm_s": t2 - t1, "tree_build_s": built, "counts": np.array([len(c[0]) for c in corr]),
                     "weights": np.array([c[3] for c in corr], dtype=np.float32)}
        return new, sm
