"""CPU-side checks of the headless drivers (no GPU touched: the device session is only created by the first search)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "mv-lm-icp_amd", "bin")
GOLD = os.path.join(ROOT, "tests", "golden", "bunny18.npz")


def write_bunny(d, g, views):
    """tests/golden/bunny18.npz -> the reference's on-disk layout (see tests/test_gpu_drivers.py::write_bunny)."""
    off = g["row_off"]
    for k in range(views):
        xyz = g["xyz_e8"][off[k]:off[k + 1]].astype(np.float64) / 1e8
        np.savetxt(os.path.join(d, f"cloudXYZ_{2 * k}.xyz"), np.hstack([xyz, np.zeros_like(xyz)]), fmt="%.8f %.8f %.8f %g %g %g")
        np.savetxt(os.path.join(d, f"poses_{2 * k}.txt"), g["gt"][k], fmt="%.17g")
        np.savetxt(os.path.join(d, f"cloudXYZ_{2 * k + 1}.xyz"), np.zeros((3, 6)), fmt="%g")
        np.savetxt(os.path.join(d, f"poses_{2 * k + 1}.txt"), np.eye(4), fmt="%g")


def test_multiview_driver_prints_the_readme_adjacency_matrix(tmp_path):
    """The reference's README (README.md:152-177) prints the pose graph of its default run — Bunny, limit 40 step 2 -> 18 views, knn 2 on
    the NOISY initial poses (Frame::computePoseNeighboursKnn, frame.cpp:67-89; main_multiview.cpp:104-117) — as an 18 x 18 adjacency
    matrix: the ring i <-> i +- 1.  bin/multiview on the same data (tests/golden/bunny18.npz written back in the reference's layout) prints
    the same matrix, with either standard library's noise stream (rows a4 / f3 of SURVEY.md §8 on the reference's published output)."""
    g = np.load(GOLD)
    d = tmp_path / "data"; d.mkdir()
    write_bunny(str(d), g, 18)
    ring = np.zeros((18, 18), dtype=int)
    for i in range(18):
        ring[i, (i - 1) % 18] = ring[i, (i + 1) % 18] = 1
    for stream in ("libstdc++", "libc++"):
        out = subprocess.check_output([os.path.join(BIN, "multiview"), "--dir", str(d), "--norecomputeNormals", "--rounds", "0", "--noise_stream", stream],
                                      timeout=300).decode().splitlines()
        at = out.index("graph adjacency matrix == block structure")
        got = np.array([[int(v) for v in out[at + 1 + i].split()] for i in range(18)])
        assert np.array_equal(got, ring), (stream, got)
        # ... and the graph the committed trajectory was recorded on is that ring too (edges listed src-ascending, nearest first)
    adj = np.zeros((18, 18), dtype=int)
    adj[g["src"], g["dst"]] = 1
    assert np.array_equal(adj, ring)
