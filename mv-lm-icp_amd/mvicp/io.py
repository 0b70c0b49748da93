"""The reference's on-disk formats (include/common.h:172-187, 224-239; src/main_multiview.cpp:53-100)."""
import glob
import os

import numpy as np


def load_xyz(path, phantom_row=True):
    """`x y z nx ny nz` per row (common.h:224-239).  The reference's `while(file){ Vector3d pt,no; file >> ...; push_back }` loop
    appends one extra element after the last row (the stream only fails on the NEXT read, and a failed extraction at end of file
    stores nothing): an exact DUPLICATE of the last row.  phantom_row=True (default) reproduces it — with it the pairwise known-answer
    test matches README.md:141-146 to six digits (profiles/r05_lm_pin_sweep.txt); False loads exactly the rows of the file."""
    a = np.loadtxt(path, dtype=np.float64).reshape(-1, 6)
    if phantom_row and len(a):
        a = np.vstack([a, a[-1:]])
    return np.ascontiguousarray(a[:, :3]), np.ascontiguousarray(a[:, 3:6])


def load_matrix4(path):
    """4x4 row-major text (common.h:172-187)."""
    v = np.loadtxt(path, dtype=np.float64).reshape(-1)
    m = np.zeros(16)
    m[15] = 1.0
    m[: min(16, len(v))] = v[:16]
    return m.reshape(4, 4)


def save_matrix4(path, M):
    np.savetxt(path, np.asarray(M).reshape(4, 4), fmt="%.17g")


def _sorted_files(folder, prefix):
    """getAllTextFilesFromFolder (common.h:119-170): prefix match, .txt/.xyz suffix, sort by length then lexicographic."""
    out = [p for p in glob.glob(os.path.join(folder, prefix + "*")) if p.endswith(".txt") or p.endswith(".xyz")]
    return sorted(out, key=lambda s: (len(s), s))


def load_frames(folder, limit=40, step=2, phantom_row=True):
    """loadFrames (main_multiview.cpp:53-100) without the noise step: returns (pts, nor, poses, groundtruth or None)."""
    clouds = _sorted_files(folder, "cloud")
    poses = _sorted_files(folder, "pose")
    gts = _sorted_files(folder, "groundtruth")
    pts, nor, P, G = [], [], [], []
    i = 0
    while i < len(clouds) and i < limit * step:
        p, n = load_xyz(clouds[i], phantom_row)
        pts.append(p); nor.append(n)
        P.append(load_matrix4(poses[i]))
        if len(gts) == len(clouds):
            G.append(load_matrix4(gts[i]))
        i += step
    return pts, nor, np.array(P), (np.array(G) if G else None)
