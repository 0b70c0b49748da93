"""Deterministic synthetic multi-view clouds for the BASELINE.json configs (build-owned).

World surface: bumpy sphere r(theta, phi) = 0.10 (1 + 0.15 sin 3theta sin 4phi) m (Bunny-scale), analytic unit
normals.  K cameras on a turntable ring: ground-truth pose T_k = Ry(yaw_k) * Translation(0, 0, -0.4), so a view's
points sit at local z ~ +0.4 with camera-facing normals n_z < 0 like samples/Bunny_RealData.  View k holds N points
drawn uniformly (in direction) inside a 100 degree cone around the camera direction, plus isotropic Gaussian noise
sigma = 1e-4 m, stored in the view's LOCAL frame (fp64).  Initial poses: view 0 = ground truth (fixed); the others
addNoise(T_k, 0.02, 0.01) with the semantics of include/common.h:38-67 (right-multiplied so(3) noise, additive
translation noise).  numpy Generator(PCG64) streams, seed 1000+k per view, seed 5489 for the pose noise.
"""
import numpy as np

R0 = 0.10
BUMP = 0.15
RING = 0.4
CONE_DEG = 100.0
POINT_SIGMA = 1e-4


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def so3_exp(w):
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + W
    return np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * (W @ W)


def surface(dirs):
    """Unit directions (n,3) -> (points, outward unit normals) on the bumpy sphere (polar axis = +y)."""
    u = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    theta = np.arccos(np.clip(u[:, 1], -1, 1))
    phi = np.arctan2(u[:, 2], u[:, 0])
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    r = R0 * (1 + BUMP * np.sin(3 * theta) * np.sin(4 * phi))
    r_t = R0 * BUMP * 3 * np.cos(3 * theta) * np.sin(4 * phi)
    r_p = R0 * BUMP * np.sin(3 * theta) * 4 * np.cos(4 * phi)
    uu = np.stack([st * cp, ct, st * sp], 1)
    u_t = np.stack([ct * cp, -st, ct * sp], 1)
    u_p = np.stack([-st * sp, np.zeros_like(st), st * cp], 1)
    S_t = r_t[:, None] * uu + r[:, None] * u_t
    S_p = r_p[:, None] * uu + r[:, None] * u_p
    n = np.cross(S_t, S_p)
    nn = np.linalg.norm(n, axis=1, keepdims=True)
    bad = nn[:, 0] < 1e-14
    n = np.where(bad[:, None], uu, n / np.where(nn == 0, 1, nn))
    n *= np.sign(np.sum(n * uu, 1, keepdims=True) + 1e-300)
    return r[:, None] * uu, n


def yaw_step(K):
    return np.deg2rad(min(360.0 / K, 20.0))


def gt_pose(k, K):
    T = np.eye(4)
    R = rot_y(k * yaw_step(K))
    T[:3, :3] = R
    T[:3, 3] = R @ np.array([0, 0, -RING])
    return T


def make_view(k, K, N, seed_base=1000, cone_deg=CONE_DEG):
    rng = np.random.Generator(np.random.PCG64(seed_base + k))
    T = gt_pose(k, K)
    R = T[:3, :3]
    c = R @ np.array([0.0, 0.0, -1.0])  # direction from the object centre towards the camera
    # orthonormal frame around c
    a = np.cross(c, [0.0, 1.0, 0.0]); a /= np.linalg.norm(a)
    b = np.cross(c, a)
    cos_a = rng.uniform(np.cos(np.deg2rad(cone_deg)), 1.0, N)
    beta = rng.uniform(0, 2 * np.pi, N)
    sin_a = np.sqrt(1 - cos_a ** 2)
    dirs = cos_a[:, None] * c + sin_a[:, None] * (np.cos(beta)[:, None] * a + np.sin(beta)[:, None] * b)
    pw, nw = surface(dirs)
    pw = pw + rng.normal(0, POINT_SIGMA, pw.shape)
    pl = (pw - T[:3, 3]) @ R      # R^T (p - t)
    nl = nw @ R
    return np.ascontiguousarray(pl), np.ascontiguousarray(nl)


def add_noise(T, sigma, sigmat, rng):
    """common.h:38-67: noisyPose = pose * Exp(sigma * N(0,1)^3); translation += sigmat * N(0,1)^3 (draw order w then t)."""
    w = rng.normal(0, 1, 3) * sigma
    out = T.copy()
    out[:3, :3] = T[:3, :3] @ so3_exp(w)
    out[:3, 3] = T[:3, 3] + rng.normal(0, 1, 3) * sigmat
    return out


def make_poses(K, sigma=0.02, sigmat=0.01, knn=2, pose_seed=5489):
    """Ground-truth / initial poses, pose graph and fixed mask of a K-view problem — everything of make_problem except the
    clouds (which are independent streams, seed_base + k per view), so a test can build a few views of a large config."""
    gt = [gt_pose(k, K) for k in range(K)]
    rng = np.random.Generator(np.random.PCG64(pose_seed))
    init = [gt[0].copy()] + [add_noise(gt[k], sigma, sigmat, rng) for k in range(1, K)]
    src, dst = pose_graph_knn(np.array(init), knn)
    fixed = np.zeros(K, dtype=np.uint8)
    fixed[0] = 1
    return {"gt": np.array(gt), "init": np.array(init), "src": src, "dst": dst, "fixed": fixed}


def make_problem(K, N, sigma=0.02, sigmat=0.01, knn=2, seed_base=1000, pose_seed=5489, cone_deg=CONE_DEG):
    """cone_deg: half-angle of the cone of directions a view sees (100 = the BASELINE configs: ring neighbours overlap almost fully;
    20 = a partial-overlap variant: a third of a view's points have no counterpart in a ring neighbour)."""
    pts, nor = [], []
    for k in range(K):
        p, n = make_view(k, K, N, seed_base, cone_deg)
        pts.append(p); nor.append(n)
    pb = make_poses(K, sigma, sigmat, knn, pose_seed)
    pb["pts"] = pts; pb["nor"] = nor
    return pb


def pose_graph_knn(poses, knn, skip_fixed0=True):
    """Frame::computePoseNeighboursKnn (src/internal/frame.cpp:67-89) for every frame: the k nearest views by the
    float32 norm of the translation difference; edges listed src-ascending, neighbours nearest first.  Frame 0 is
    never a source (frame.cpp:93, icp-ceres.cpp:255,351,426) so its edges are omitted unless skip_fixed0=False."""
    K = len(poses)
    src, dst = [], []
    for i in range(K):
        if skip_fixed0 and i == 0:
            continue
        cand = [j for j in range(K) if j != i]
        d = np.array([np.float32(np.linalg.norm(poses[i][:3, 3] - poses[j][:3, 3])) for j in cand], dtype=np.float32)
        order = np.argsort(d, kind="stable")[: min(knn, len(cand))]
        for o in order:
            src.append(i); dst.append(cand[o])
    return np.array(src, dtype=np.int32), np.array(dst, dtype=np.int32)


def pose_diff(P1, P2):
    """common.h:259-282 poseDiff semantics — (||t1 - t2||, relative rotation angle in radians) — with the angle
    taken as atan2(|sin|, cos) from the skew part so it stays accurate below sqrt(eps) (the reference's acos form
    bottoms out near 1.7e-6 degrees: README.md:142-146)."""
    dt = float(np.linalg.norm(P1[:3, 3] - P2[:3, 3]))
    R = P1[:3, :3].T @ P2[:3, :3]
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    ang = float(np.arctan2(np.linalg.norm(v), (np.trace(R) - 1) / 2))
    return dt, ang
