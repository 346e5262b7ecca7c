#!/usr/bin/env python3
"""Generates tests/golden/jacobian_golden.npz: residual + Jacobian values of the reference's OWN symbolic
derivation, evaluated in float64 at random points.

Runs in the build container only (needs /root/reference and sympy); nothing of the reference travels: the output
is numbers -- inputs and expected outputs.  It imports

    applications/badslam/scripts/jacobian_functions.py    (ComputeValueAndJacobian: value + symbolic Jacobian of one stage)
    applications/badslam/scripts/jacobians_derivation.py  (SE3exp, SE3Inverse, Project, Unproject, CorrectDepth,
                                                           InterpolateBilinear / frac, DotProduct3, ... :42-164)

and builds the residuals exactly as jacobians_derivation.py:169-310 does (same stage lists, same parameters, same
evaluation points), chaining the per-stage Jacobians with the chain rule like jacobian_functions.ComputeJacobian
(:131-175; that function only prints, so the loop is repeated here around the reference's ComputeValueAndJacobian).
The one renamed sympy module the scripts import (sympy.printing.cxxcode -> sympy.printing.cxx) is aliased first.

Kinds (symbols as in the script):
  depth_pose        dot(n, gtf * exp(T) * l - s)                              wrt T (6) at 0      script :218-231
  depth_position    dot(n, g - (s + t n))                                     wrt t at 0          script :236-243
  depth_intrinsics  dot(n, gtf * Unproject(x, y, depth, K) - s)               wrt K = (fx_inv, fy_inv, cx_inv, cy_inv)   :247-255
  depth_deformation dot(n, gtf * Unproject(x, y, CorrectDepth(cfactor, a, raw_inv_depth), K) - s)   wrt (cfactor, a)   :259-268
  desc_pose         Interp(Project(SE3Inverse(exp(T)) * ls)) - d              wrt T (6) at 0      script :285-296
  desc_position     Interp(Project(ftg * (s + t n))) - d                      wrt t at 0          script :301-310
  desc_color_intrinsics  Interp(Project(ls, fx, fy, cx, cy)) - d              wrt (fx, fy, cx, cy): the parameterisation the
                    kernel uses (BS/kernel_opt_intrinsics.cu:141-149; the script's variant :314-320 differentiates wrt the
                    INVERSE parameters, which no kernel uses), built from the script's Project / InterpolateBilinear.
"""
import os
import sys
import time

import numpy as np

REF_SCRIPTS = "/root/reference/applications/badslam/scripts"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jacobian_golden.npz")
N = 1000
SEED = 0xBAD51A4


def import_reference():
    import sympy
    import sympy.printing.cxx
    sys.modules["sympy.printing.cxxcode"] = sympy.printing.cxx      # renamed in sympy >= 1.7
    sys.dont_write_bytecode = True                                   # never write into /root/reference
    sys.path.insert(0, REF_SCRIPTS)
    import jacobian_functions as jf
    import jacobians_derivation as jd
    return sympy, jf, jd


def chain(sympy, jf, functions, parameters, parameter_values):
    """Residual and Jacobian of functions[0](functions[1](... functions[-1](parameters))) at parameter_values: the loop of
    jacobian_functions.ComputeJacobian (:131-175) around the reference's ComputeValueAndJacobian."""
    previous = None
    cur_params, cur_values = parameters, parameter_values
    for i in range(len(functions) - 1, -1, -1):
        values, jac = jf.ComputeValueAndJacobian(functions[i], cur_params, cur_values)
        cur_values = values
        if isinstance(values, sympy.Matrix):
            cur_params = sympy.Matrix(values.rows, values.cols, lambda r, c: sympy.var("T_%d%d" % (r, c)))
        else:
            cur_params = sympy.symbols("T")
        previous = jac if previous is None else sympy.simplify(jac * previous)
    residual = parameter_values
    for i in range(len(functions) - 1, -1, -1):
        residual = functions[i](residual)
    return residual, previous


def main():
    sympy, jf, jd = import_reference()
    S, M = sympy.Symbol, sympy.Matrix
    n = M(3, 1, lambda i, j: S("n_%d" % i, real=True))
    gtf = M(3, 4, lambda i, j: S("gtf_%d_%d" % (i, j), real=True))
    ftg = M(3, 4, lambda i, j: S("ftg_%d_%d" % (i, j), real=True))
    l = M(3, 1, lambda i, j: S("l_%d" % i, real=True))
    g = M(3, 1, lambda i, j: S("g_%d" % i, real=True))
    s = M(3, 1, lambda i, j: S("s_%d" % i, real=True))
    ls = M(3, 1, lambda i, j: S("ls_%d" % i, real=True))
    fx, fy, cx, cy = [S(v, real=True) for v in ("fx", "fy", "cx", "cy")]
    fx_inv, fy_inv, cx_inv, cy_inv = [S(v, real=True) for v in ("fx_inv", "fy_inv", "cx_inv", "cy_inv")]
    x, y, t, depth, rid, cfactor, a = [S(v, real=True) for v in ("x", "y", "t", "depth", "raw_inv_depth", "cfactor", "a")]
    tl, tr, bl, br, d = [S(v, real=True) for v in ("top_left", "top_right", "bottom_left", "bottom_right", "surfel_gradmag")]
    # (the script declares its pose delta with var("T_i"), i.e. complex symbols; on sympy >= 1.5 omega.norm() of complex symbols
    # is sqrt(|T_3|^2 + ...), whose derivative at 0 evaluates to nan -- the delta is real, and declared so here)
    T = M(6, 1, lambda i, j: S("T_%d" % i, real=True))
    zero6 = sympy.zeros(6, 1)
    interp = lambda p: jd.InterpolateBilinear(p[0], p[1], tl, tr, bl, br)

    kinds = {}
    kinds["depth_pose"] = chain(sympy, jf, [lambda p: jd.DotProduct3(n, p), lambda p: p - s,
                                            lambda p: jd.MatrixVectorMultiplyHomogeneous(gtf, p),
                                            lambda m: jd.MatrixVectorMultiplyHomogeneous(m, l), jd.SE3exp], T, zero6)
    kinds["depth_position"] = chain(sympy, jf, [lambda p: jd.DotProduct3(n, p), lambda sf: g - sf, lambda pd: s + pd, lambda tt: tt * n], t, 0)
    K = M([[fx_inv], [fy_inv], [cx_inv], [cy_inv]])
    kinds["depth_intrinsics"] = chain(sympy, jf, [lambda p: jd.DotProduct3(n, p), lambda p: p - s,
                                                  lambda lp: jd.MatrixVectorMultiplyHomogeneous(gtf, lp),
                                                  lambda k: jd.Unproject(x, y, depth, k[0], k[1], k[2], k[3])], K, K)
    P2 = M([[cfactor], [a]])
    kinds["depth_deformation"] = chain(sympy, jf, [lambda p: jd.DotProduct3(n, p), lambda p: p - s,
                                                   lambda lp: jd.MatrixVectorMultiplyHomogeneous(gtf, lp),
                                                   lambda dd: jd.Unproject(x, y, dd, fx_inv, fy_inv, cx_inv, cy_inv),
                                                   lambda p: jd.CorrectDepth(p[0], p[1], rid)], P2, P2)
    kinds["desc_pose"] = chain(sympy, jf, [lambda v: v - d, interp, lambda p: jd.Project(p, fx, fy, cx, cy),
                                           lambda left: jd.MatrixVectorMultiplyHomogeneous(left, ls), jd.SE3Inverse, jd.SE3exp], T, zero6)
    kinds["desc_position"] = chain(sympy, jf, [lambda v: v - d, interp, lambda p: jd.Project(p, fx, fy, cx, cy),
                                               lambda gs: jd.MatrixVectorMultiplyHomogeneous(ftg, gs), lambda pd: s + pd, lambda tt: tt * n], t, 0)
    C = M([[fx], [fy], [cx], [cy]])
    kinds["desc_color_intrinsics"] = chain(sympy, jf, [lambda v: v - d, interp, lambda k: jd.Project(ls, k[0], k[1], k[2], k[3])], C, C)

    rng = np.random.default_rng(SEED)

    def rot(count):
        q = rng.normal(size=(count, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        w, xx, yy, zz = q.T
        return np.stack([1 - 2 * (yy * yy + zz * zz), 2 * (xx * yy - zz * w), 2 * (xx * zz + yy * w),
                         2 * (xx * yy + zz * w), 1 - 2 * (xx * xx + zz * zz), 2 * (yy * zz - xx * w),
                         2 * (xx * zz - yy * w), 2 * (yy * zz + xx * w), 1 - 2 * (xx * xx + yy * yy)], axis=1).reshape(count, 3, 3)

    def unit(count):
        v = rng.normal(size=(count, 3))
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    # random points in the ranges the path sees: depths 0.5 .. 5 m, 640x480 images, TUM-like cameras
    val = {}
    R, tr_ = rot(N), rng.uniform(-1.5, 1.5, (N, 3))
    rigid = np.concatenate([R, tr_[:, :, None]], axis=2)           # 3x4
    for i in range(3):
        for j in range(4):
            val["gtf_%d_%d" % (i, j)] = rigid[:, i, j]
            val["ftg_%d_%d" % (i, j)] = rigid[:, i, j]
    loc = np.stack([rng.uniform(-1.5, 1.5, N), rng.uniform(-1.2, 1.2, N), rng.uniform(0.5, 5.0, N)], axis=1)
    for i in range(3):
        val["l_%d" % i] = loc[:, i]
        val["ls_%d" % i] = loc[:, i] + rng.uniform(-0.02, 0.02, N)
        val["s_%d" % i] = rng.uniform(-2.0, 2.0, N)
        val["g_%d" % i] = rng.uniform(-2.0, 2.0, N)
    val["fx"], val["fy"] = rng.uniform(200, 600, N), rng.uniform(200, 600, N)
    val["cx"], val["cy"] = rng.uniform(300, 340, N), rng.uniform(220, 260, N)
    val["fx_inv"], val["fy_inv"] = 1.0 / val["fx"], 1.0 / val["fy"]
    val["cx_inv"], val["cy_inv"] = -(val["cx"] - 0.5) / val["fx"], -(val["cy"] - 0.5) / val["fy"]
    val["x"], val["y"] = rng.integers(0, 640, N).astype(np.float64), rng.integers(0, 480, N).astype(np.float64)
    # surfel normal: within ~37 degrees of the direction back along the pixel's viewing ray in the frame (what the association
    # test lets through: normal towards the camera, within 40 degrees of the pixel normal), rotated to global by gtf
    ray = np.stack([val["fx_inv"] * val["x"] + val["cx_inv"], val["fy_inv"] * val["y"] + val["cy_inv"], np.ones(N)], axis=1)
    ray /= np.linalg.norm(ray, axis=1, keepdims=True)
    ln = -ray + 0.6 * unit(N)
    ln /= np.linalg.norm(ln, axis=1, keepdims=True)
    nn = np.einsum("nij,nj->ni", R, ln)
    for i in range(3):
        val["n_%d" % i] = nn[:, i]
    val["depth"] = rng.uniform(0.5, 5.0, N)
    val["raw_inv_depth"] = 1.0 / rng.uniform(0.5, 5.0, N)
    val["cfactor"], val["a"] = rng.uniform(-0.01, 0.01, N), rng.uniform(-0.05, 0.05, N)
    for name in ("top_left", "top_right", "bottom_left", "bottom_right"):
        val[name] = rng.integers(0, 256, N) / 255.0
    val["surfel_gradmag"] = rng.uniform(0, 1, N)
    # desc_position differentiates through ftg * s: keep that point in front of the camera
    s_front = np.stack([rng.uniform(-1.5, 1.5, N), rng.uniform(-1.2, 1.2, N), rng.uniform(0.5, 5.0, N)], axis=1)
    s_global = np.einsum("nji,nj->ni", R, s_front - tr_)            # s = R^T (p_local - t)  =>  ftg * s = p_local
    val_pos = dict(val)
    for i in range(3):
        val_pos["s_%d" % i] = s_global[:, i]

    out = {"seed": np.array([SEED], np.uint64), "count": np.array([N])}
    frac_np = {"frac": lambda v: v - np.floor(v)}
    for kind, (res, jac) in kinds.items():
        t0 = time.time()
        exprs = [res if not isinstance(res, sympy.MatrixBase) else res[0]] + list(sympy.Matrix(jac))
        syms = sorted({str(sm) for e in exprs for sm in sympy.sympify(e).free_symbols})
        fn = sympy.lambdify([sympy.Symbol(nm, real=True) for nm in syms], exprs, modules=[frac_np, "numpy"])
        src = val_pos if kind == "desc_position" else val
        args = [src[nm] for nm in syms]
        vals = [np.broadcast_to(np.asarray(v, np.float64), (N,)) for v in fn(*args)]
        out[kind + "/symbols"] = np.array(syms)
        out[kind + "/inputs"] = np.stack(args, axis=1)
        out[kind + "/residual"] = vals[0].copy()
        out[kind + "/jacobian"] = np.stack(vals[1:], axis=1)
        print(f"{kind}: {len(syms)} inputs, jacobian {out[kind + '/jacobian'].shape}, {time.time() - t0:.1f} s", flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
