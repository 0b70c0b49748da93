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


def test_binary_xyz_header_is_not_trusted(tmp_path):
    """the binary .xyz variant (host/common_io.h: "MVXYZB1\\n" + int64 row count + raw rows) used to size its buffer from the header before
    looking at the file — a corrupt count asked for a multi-GB allocation or threw out of a function that reports failure by return value.  Now the count is
    bounded by the bytes the file holds: a corrupt file is reported and skipped (an empty cloud), a good one loads, and the g++ noise stream draws other
    initial poses than the default one (the reference's constructor-argument evaluation order, common.h:43,52)."""
    import struct
    d = tmp_path / "data"; d.mkdir()
    rng = np.random.default_rng(3)
    rows = np.hstack([rng.normal(size=(50, 3)), np.tile([0.0, 0.0, -1.0], (50, 1))])
    for k in range(3):
        with open(d / f"cloud_{k}.xyz", "wb") as f:
            f.write(b"MVXYZB1\n"); f.write(struct.pack("<q", 50)); rows.astype("<f8").tofile(f)
        T = np.eye(4); T[0, 3] = 0.1 * k
        np.savetxt(d / f"pose_{k}.txt", T, fmt="%.17g")
    good = subprocess.run([os.path.join(BIN, "multiview"), "--dir", str(d), "--step", "1", "--norecomputeNormals", "--rounds", "0"], capture_output=True, text=True, timeout=120)
    assert good.returncode == 0 and "graph adjacency matrix" in good.stdout and "claims" not in good.stderr
    for bad_count in (1 << 40, (1 << 62) + 5, 51):
        with open(d / "cloud_1.xyz", "wb") as f:
            f.write(b"MVXYZB1\n"); f.write(struct.pack("<q", bad_count)); rows.astype("<f8").tofile(f)
        r = subprocess.run([os.path.join(BIN, "multiview"), "--dir", str(d), "--step", "1", "--norecomputeNormals", "--rounds", "0"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and f"binary header claims {bad_count} rows, the file holds 50" in r.stderr, (bad_count, r.stderr[-300:])
    # noise streams: three different initial pose sets from the same default-seeded generator
    os.remove(d / "cloud_1.xyz")
    with open(d / "cloud_1.xyz", "wb") as f:
        f.write(b"MVXYZB1\n"); f.write(struct.pack("<q", 50)); rows.astype("<f8").tofile(f)
    outs = {}
    for stream in ("libstdc++", "libc++", "g++"):
        o = tmp_path / f"out_{stream.replace('+', 'p')}"; o.mkdir()
        subprocess.check_call([os.path.join(BIN, "multiview"), "--dir", str(d), "--step", "1", "--norecomputeNormals", "--rounds", "0", "--quiet", "--noise_stream", stream,
                               "--out", str(o)], timeout=120)
        outs[stream] = np.array([np.loadtxt(o / f"pose_{k}.txt") for k in range(3)])
    assert np.array_equal(outs["g++"][0], outs["libstdc++"][0])                      # frame 0 = ground truth, never noisy (main_multiview.cpp:83)
    assert not np.allclose(outs["g++"][1], outs["libstdc++"][1]) and not np.allclose(outs["libc++"][1], outs["libstdc++"][1])
    # g++ = the libstdc++ variates with each triple reversed: the translation noise of frame 1 is the same three numbers in reverse order
    gt1 = np.eye(4); gt1[0, 3] = 0.1
    dn_std = outs["libstdc++"][1][:3, 3] - gt1[:3, 3]; dn_gpp = outs["g++"][1][:3, 3] - gt1[:3, 3]
    assert np.allclose(dn_gpp, dn_std[::-1], rtol=0, atol=1e-15)
