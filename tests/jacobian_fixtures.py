"""Maps the reference-derived fixtures of tests/golden/jacobian_golden.npz (values of the reference's symbolic residuals and
Jacobians, applications/badslam/scripts/jacobians_derivation.py, written by tests/golden/make_jacobian_golden.py) onto the
inputs of the point-wise Jacobian probes -- oracle `bso_jacobian_probe`, HIP `bslam_debug_jacobians`, same layouts
(oracle/bslam_oracle.c) -- and names the expected outputs.  Everything here is float64 numpy; the probes run in fp32."""
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jacobian_golden.npz")
IN_WIDTH = {0: 10, 1: 1, 2: 21, 3: 11, 4: 14, 5: 8}
OUT_WIDTH = {0: 7, 1: 1, 2: 7, 3: 9, 4: 1, 5: 4}


class Kind:
    def __init__(self, data, name):
        self.name = name
        syms = [str(s) for s in data[name + "/symbols"]]
        inputs = data[name + "/inputs"]
        self.v = {s: inputs[:, i] for i, s in enumerate(syms)}
        self.residual = data[name + "/residual"]
        self.jacobian = data[name + "/jacobian"]
        self.n = inputs.shape[0]

    def vec(self, prefix):
        return np.stack([self.v["%s_%d" % (prefix, i)] for i in range(3)], axis=1)

    def rigid(self, prefix):
        M = np.stack([np.stack([self.v["%s_%d_%d" % (prefix, i, j)] for j in range(4)], axis=1) for i in range(3)], axis=1)
        return M[:, :, :3], M[:, :, 3]


def load():
    data = np.load(PATH, allow_pickle=False)
    return {name: Kind(data, name) for name in ("depth_pose", "depth_position", "depth_intrinsics", "depth_deformation",
                                                "desc_pose", "desc_position", "desc_color_intrinsics")}


def frac(v):
    return v - np.floor(v)


def cases():
    """List of (fixture kind, probe kind, probe inputs [n, IN_WIDTH], expected {output column(s): values}, scale note)."""
    K = load()
    out = []

    k = K["depth_pose"]                       # dot(n, gtf exp(T) l - s): n_local = R^T n, ls = R^T (s - t), lu = l
    R, t = k.rigid("gtf")
    n_local = np.einsum("nji,nj->ni", R, k.vec("n"))
    ls = np.einsum("nji,nj->ni", R, k.vec("s") - t)
    inp = np.concatenate([np.ones((k.n, 1)), n_local, k.vec("l"), ls], axis=1)
    out.append(("depth_pose", 0, inp, {"residual": (slice(0, 1), k.residual[:, None]), "jacobian": (slice(1, 7), k.jacobian)}))

    k = K["depth_position"]                   # d/dt dot(n, g - (s + t n)) = -|n|^2; the kernels assume a unit normal
    nn = (k.vec("n") ** 2).sum(axis=1)
    inp = np.ones((k.n, 1))
    out.append(("depth_position", 1, inp, {"jacobian": (slice(0, 1), k.jacobian / nn[:, None])}))

    k = K["depth_intrinsics"]                 # wrt (fx_inv, fy_inv, cx_inv, cy_inv); frame_T_global = gtf^-1: rows of R^T
    R, t = k.rigid("gtf")
    n = k.vec("n")
    ln = np.einsum("nji,nj->ni", R, n)
    nx = k.v["fx_inv"] * k.v["x"] + k.v["cx_inv"]
    ny = k.v["fy_inv"] * k.v["y"] + k.v["cy_inv"]
    col = lambda a: a[:, None]
    inp = np.concatenate([np.ones((k.n, 1)), col(k.v["depth"]), col(k.v["x"]), col(k.v["y"]), col(nx), col(ny), n, R[:, :, 0], R[:, :, 1], ln,
                          np.zeros((k.n, 2)), col(1.0 / k.v["depth"])], axis=1)
    out.append(("depth_intrinsics", 2, inp, {"jacobian": (slice(1, 5), k.jacobian)}))

    k = K["depth_deformation"]                # wrt (cfactor, a): probe columns dj[5] (cfactor), dj[4] (a)
    R, t = k.rigid("gtf")
    n = k.vec("n")
    ln = np.einsum("nji,nj->ni", R, n)
    nx = k.v["fx_inv"] * k.v["x"] + k.v["cx_inv"]
    ny = k.v["fy_inv"] * k.v["y"] + k.v["cy_inv"]
    rid, cf, a = k.v["raw_inv_depth"], k.v["cfactor"], k.v["a"]
    depth = 1.0 / (rid + cf * np.exp(-a * rid))
    inp = np.concatenate([np.ones((k.n, 1)), col(depth), col(k.v["x"]), col(k.v["y"]), col(nx), col(ny), n, R[:, :, 0], R[:, :, 1], ln,
                          col(cf), col(a), col(rid)], axis=1)
    out.append(("depth_deformation", 2, inp, {"jacobian": ([6, 5], k.jacobian), "corrected_inv_depth": (slice(0, 1), col(rid + cf * np.exp(-a * rid)))}))

    def quad(k):
        return np.stack([k.v["top_left"], k.v["top_right"], k.v["bottom_left"], k.v["bottom_right"]], axis=1)

    k = K["desc_pose"]                        # Interp(Project(SE3Inverse(exp(T)) ls)) - d
    ls = k.vec("ls")
    u = k.v["fx"] * ls[:, 0] / ls[:, 2] + k.v["cx"]
    v = k.v["fy"] * ls[:, 1] / ls[:, 2] + k.v["cy"]
    inp = np.concatenate([quad(k), col(frac(u)), col(frac(v)), col(k.v["fx"]), col(k.v["fy"]), ls], axis=1)
    out.append(("desc_pose", 3, inp, {"residual": (slice(0, 1), col(k.residual + k.v["surfel_gradmag"])), "jacobian": (slice(3, 9), k.jacobian)}))

    k = K["desc_position"]                    # Interp(Project(ftg (s + t n))) - d
    R, t = k.rigid("ftg")
    ls = np.einsum("nij,nj->ni", R, k.vec("s")) + t
    rn = np.einsum("nij,nj->ni", R, k.vec("n"))
    u = k.v["fx"] * ls[:, 0] / ls[:, 2] + k.v["cx"]
    v = k.v["fy"] * ls[:, 1] / ls[:, 2] + k.v["cy"]
    inp = np.concatenate([quad(k), col(frac(u)), col(frac(v)), col(k.v["fx"]), col(k.v["fy"]), rn, ls], axis=1)
    # a single number formed as a SUM of two products (x and y gradient parts) that may cancel: the error is judged against
    # the larger of the two parts, not against their sum
    q = quad(k)
    gxv = (q[:, 3] - q[:, 2]) * frac(v) + (q[:, 1] - q[:, 0]) * (1 - frac(v))
    gyv = (q[:, 3] - q[:, 1]) * frac(u) + (q[:, 2] - q[:, 0]) * (1 - frac(u))
    part_x = np.abs(gxv * k.v["fx"] * (rn[:, 0] * ls[:, 2] - rn[:, 2] * ls[:, 0]) / ls[:, 2] ** 2)
    part_y = np.abs(gyv * k.v["fy"] * (rn[:, 1] * ls[:, 2] - rn[:, 2] * ls[:, 1]) / ls[:, 2] ** 2)
    out.append(("desc_position", 4, inp, {"jacobian": (slice(0, 1), k.jacobian, col(np.maximum(part_x, part_y)))}))

    k = K["desc_color_intrinsics"]            # wrt the colour camera's (fx, fy, cx, cy); nx, ny := ls.x / ls.z, ls.y / ls.z
    ls = k.vec("ls")
    u = k.v["fx"] * ls[:, 0] / ls[:, 2] + k.v["cx"]
    v = k.v["fy"] * ls[:, 1] / ls[:, 2] + k.v["cy"]
    inp = np.concatenate([quad(k), col(frac(u)), col(frac(v)), col(ls[:, 0] / ls[:, 2]), col(ls[:, 1] / ls[:, 2])], axis=1)
    out.append(("desc_color_intrinsics", 5, inp, {"jacobian": (slice(0, 4), k.jacobian)}))
    return out


def check(probe, rel_tol):
    """probe(kind, float32 inputs [n, w]) -> float32 outputs [n, OUT_WIDTH[kind]].  Every expected block must match within
    rel_tol of that sample's largest |expected| entry (Jacobian rows mix magnitudes; a zero entry has no relative error)."""
    report = {}
    for name, kind, inp, expected in cases():
        assert inp.shape[1] == IN_WIDTH[kind], (name, inp.shape)
        got = np.asarray(probe(kind, np.ascontiguousarray(inp, np.float32)), np.float64)
        assert got.shape == (inp.shape[0], OUT_WIDTH[kind]), (name, got.shape)
        for what, spec in expected.items():
            cols, exp = spec[0], spec[1]
            g = got[:, cols]
            scale = np.maximum(np.abs(exp).max(axis=1, keepdims=True), 1e-30)
            if len(spec) > 2:
                scale = np.maximum(scale, spec[2])
            err = (np.abs(g - exp) / scale).max()
            report[name + "/" + what] = float(err)
            assert err <= rel_tol, (name, what, err)
    return report
