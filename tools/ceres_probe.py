#!/usr/bin/env python3
"""Is the reference's LM back-end (Ceres Solver + Eigen3 + Sophus; SURVEY.md §8c "On the GPU box") installed on THIS machine?

The LM half of the oracle is a restatement of Ceres' published trust-region algorithm because none of these libraries is in
the build image.  smoke() runs this probe on the GPU box and records the answer (gpurun_out/ceres_probe.json + one stdout
line), so every report can say which it was: "pose parity vs fp64 CPU restatement; Ceres unavailable" or "real Ceres present".
Pure filesystem / ldconfig look-ups; nothing is compiled or imported."""
import glob
import json
import os
import subprocess
import sys

INCLUDE_ROOTS = ["/usr/include", "/usr/local/include", "/opt/include", "/opt/conda/include", "/opt/rocm/include", os.path.expanduser("~/.local/include")]


def find_header(rel):
    hits = []
    for root in INCLUDE_ROOTS:
        hits += glob.glob(os.path.join(root, rel)) + glob.glob(os.path.join(root, "*", rel))
    return sorted(set(hits))


def find_lib(stem):
    hits = []
    try:
        out = subprocess.run(["ldconfig", "-p"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20).stdout.decode()
        hits += [ln.split("=>")[-1].strip() for ln in out.splitlines() if stem in ln]
    except Exception:
        pass
    for d in ("/usr/lib", "/usr/lib/x86_64-linux-gnu", "/usr/local/lib", "/opt/conda/lib"):
        hits += glob.glob(os.path.join(d, f"lib{stem}*.so*")) + glob.glob(os.path.join(d, f"lib{stem}*.a"))
    return sorted(set(hits))


def probe():
    res = {
        "ceres_header": find_header("ceres/ceres.h"),
        "ceres_lib": find_lib("ceres"),
        "eigen_header": find_header("Eigen/Core") + find_header("eigen3/Eigen/Core"),
        "sophus_header": find_header("sophus/se3.hpp"),
        "gflags_header": find_header("gflags/gflags.h"),
        "glog_header": find_header("glog/logging.h"),
    }
    res["ceres_usable"] = bool(res["ceres_header"] and res["ceres_lib"] and res["eigen_header"])
    res["verdict"] = ("real Ceres present: a Ceres-linked harness could pin the LM half" if res["ceres_usable"]
                      else "Ceres unavailable here: pose parity is vs the fp64 CPU restatement (oracle/), which reproduces the reference's published Ceres result "
                           "(README.md:141-146) to six digits — see the README-vector line of smoke()")
    return res


def main():
    res = probe()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "ceres_probe.json"), "w") as f:
            json.dump(res, f, indent=1)
    except OSError:
        pass
    print("ceres probe:", res["verdict"])
    return res


if __name__ == "__main__":
    main()
    sys.exit(0)
