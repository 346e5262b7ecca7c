"""Scene generators: the reference's known-answer test scenes re-stated on the host
(BS/test/*.cc, BS = /root/reference/applications/badslam/src/badslam) plus the synthetic
multi-keyframe stacks of SURVEY.md 8(d).  Random draws come from a seeded numpy
generator instead of glibc rand()/srand(0): the pass criteria are on the recovered
pose / depth, which does not depend on the particular draw.
"""
import numpy as np

from badslam_amd import abi
from tests import bso

W, H = 640, 480


def reference_test_camera(width=W, height=H):
    # BS/test/test_pose_optimization_geometric_residual.cc:56: {0.5h, 0.5h, 0.5w - 0.5, 0.5h - 0.5}
    return bso.make_camera(0.5 * height, 0.5 * height, 0.5 * width - 0.5, 0.5 * height - 0.5, width, height)


def unproject_dirs(cam, width, height):
    """PinholeCamera4f::UnprojectFromPixelCenterConv for every pixel -> (x/z, y/z) float32 grids."""
    xs = np.arange(width, dtype=np.float32)
    ys = np.arange(height, dtype=np.float32)
    dx = (xs - np.float32(cam.cx - 0.5)) / np.float32(cam.fx)
    dy = (ys - np.float32(cam.cy - 0.5)) / np.float32(cam.fy)
    return np.meshgrid(dx, dy)


def pose_geometric_scene(seed=0, width=W, height=H, cell=1, max_surfels=1000 * 1000, return_raw=False):
    """Optimization.PoseOptimizationWithGeometricResidual
    (BS/test/test_pose_optimization_geometric_residual.cc:50-131): one keyframe at identity
    looking at 3 vertical plane strips, depth residuals only."""
    rng = np.random.default_rng(seed)
    cam = reference_test_camera(width, height)
    raw_to_float_depth = np.float32(1.0 / 1000)
    scene = bso.HostScene(cam, cam, float(raw_to_float_depth), 40.0, cell, max_surfels,
                          use_depth_residuals=True, use_descriptor_residuals=False)
    depth = np.full((height, width), 65535, np.uint16)
    dxg, dyg = unproject_dirs(cam, width, height)
    for plane_index in range(3):
        n = rng.uniform(-1, 1, 3).astype(np.float32)
        n[2] = -1.0
        n /= np.linalg.norm(n)
        max_x, min_x = width - 10 - 1, 10
        left = int(min_x + (max_x - min_x) * ((2 * plane_index) / (2.0 * 3 - 1)))
        right = int(min_x + (max_x - min_x) * ((2 * plane_index + 1) / (2.0 * 3 - 1)))
        # Hyperplane(n, 2.5): n.x + 2.5 = 0; ray = t * (dx, dy, 1) from the origin
        denom = (n[0] * dxg + n[1] * dyg + n[2])[10:height - 10, left:right]
        z = (-2.5 / denom).astype(np.float32)
        depth[10:height - 10, left:right] = (z / raw_to_float_depth + np.float32(0.5)).astype(np.uint16)
    rgb = np.zeros((height, width, 3), np.uint8)
    kf = scene.add_keyframe_from_images(depth, rgb, bso.se3_identity())
    scene.create_surfels_for_keyframe(kf)
    if return_raw:
        return scene, kf, depth, rgb
    return scene, kf


def smooth_random_texture(rng, width, height):
    """BS/test/test_pose_optimization_photometric_residual.cc:100-119 (u8 wrap-around kept)."""
    img = np.zeros((height, width), np.uint8)
    r = rng.integers(0, 16, size=(height, width)).astype(np.uint32)
    for y in range(height):
        top = img[y - 1].astype(np.uint32) if y > 0 else np.zeros(width, np.uint32)
        row = np.zeros(width, np.uint32)
        for x in range(1, width):
            row[x] = (((row[x - 1] + top[x]) // 2) + r[y, x]) & 0xFF
        img[y] = row.astype(np.uint8)
    return np.repeat(img[:, :, None], 3, axis=2)


def pose_photometric_scene(seed=0, width=W, height=H, max_surfels=1000 * 1000, tex_mode=abi.TEX_FIXED_POINT_1_8):
    """Optimization.PoseOptimizationColorOnlyCues
    (BS/test/test_pose_optimization_photometric_residual.cc:50-139)."""
    rng = np.random.default_rng(seed)
    cam = reference_test_camera(width, height)
    raw_to_float_depth = np.float32(1.0 / 1000)
    scene = bso.HostScene(cam, cam, float(raw_to_float_depth), 40.0, 1, max_surfels,
                          use_depth_residuals=False, use_descriptor_residuals=True, tex_mode=tex_mode)
    depth = np.full((height, width), np.uint16(2 / raw_to_float_depth), np.uint16)
    depth[0, :] = depth[-1, :] = 65535
    depth[:, 0] = depth[:, -1] = 65535
    rgb = smooth_random_texture(rng, width, height)
    gt = bso.se3_exp([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    kf = scene.add_keyframe_from_images(depth, rgb, gt)
    scene.create_surfels_for_keyframe(kf)
    return scene, kf, gt


def geometry_geometric_scene(seed=0, width=W, height=H, max_surfels=1000 * 1000):
    """{Alternating,PCG}GeometryOptimizationWithGeometricResidual
    (BS/test/test_geometry_optimization_geometric_residual.cc:50-140).  Returns the scene with
    the keyframe's depth already perturbed, and the perturbed raw depth image."""
    rng = np.random.default_rng(seed)
    cam = reference_test_camera(width, height)
    raw_to_float_depth = np.float32(1.0 / 5000)
    scene = bso.HostScene(cam, cam, float(raw_to_float_depth), 40.0, 1, max_surfels,
                          use_depth_residuals=True, use_descriptor_residuals=True)
    r = rng.integers(0, 100, size=(height, width)).astype(np.float32)
    depth = ((np.float32(1) + np.float32(0.01) * r) / raw_to_float_depth + np.float32(0.5)).astype(np.uint16)
    depth[0, :] = depth[-1, :] = 65535
    depth[:, 0] = depth[:, -1] = 65535
    rgb = np.zeros((height, width, 3), np.uint8)
    gt = bso.se3_exp([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    kf = scene.add_keyframe_from_images(depth, rgb, gt)
    # force normals to the viewing ray (:113-123)
    dxg, dyg = unproject_dirs(cam, width, height)
    nrm = np.sqrt(dxg * dxg + dyg * dyg + 1).astype(np.float32)
    nx, ny = (-dxg / nrm).astype(np.float32), (-dyg / nrm).astype(np.float32)

    def s8(v):  # SmallFloatToEightBitSigned BS/util.cuh:102
        return (v * np.float32(127) + np.where(v > 0, np.float32(0.5), np.float32(-0.5))).astype(np.int8)

    kf.normals[:, :] = (s8(nx).view(np.uint8).astype(np.uint16) | (s8(ny).view(np.uint8).astype(np.uint16) << 8))
    scene.create_surfels_for_keyframe(kf)
    # perturb the measured depth (:131-137).  The reference adds to the ORIGINAL image
    # (border pixels wrap around from 65535 and stay "invalid-bit" or become small; they are
    # never associated because surfels were only created from interior pixels).
    r2 = rng.integers(0, 50, size=(height, width)).astype(np.float32)
    new_depth = (depth.astype(np.float32) + (np.float32(0.0001) * r2) / raw_to_float_depth).astype(np.uint32).astype(np.uint16)
    new_depth[0, :] = new_depth[-1, :] = 65535
    new_depth[:, 0] = new_depth[:, -1] = 65535
    kf.depth[:, :] = new_depth
    return scene, kf, new_depth


def geometry_photometric_scene(width=W, height=H, max_surfels=1000 * 1000, seed=0, tex_mode=abi.TEX_FIXED_POINT_1_8):
    """{Alternating,PCG}GeometryOptimizationWithPhotometricResidual
    (BS/test/test_geometry_optimization_photometric_residual.cc:128-268): two keyframes 100 px
    apart looking at a fronto-parallel plane at 2 m; surfels are created from keyframe 1, whose
    depth carries up to 1 cm of noise.  Returns (scene, kf0, kf1, expected_z)."""
    rng = np.random.default_rng(seed)
    cam = reference_test_camera(width, height)
    raw_to_float_depth = np.float32(1.0 / 5000)
    scene = bso.HostScene(cam, cam, float(raw_to_float_depth), 40.0, 1, max_surfels,
                          use_depth_residuals=False, use_descriptor_residuals=True, tex_mode=tex_mode)
    off = 100 * width // W          # kFrameOffsetPx, scaled with reduced-size images
    kdepth = np.float32(2.0)
    d0 = np.full((height, width), np.uint16(kdepth / raw_to_float_depth + np.float32(0.5)), np.uint16)
    d0[:, :off] = 65535
    d0[0, :] = d0[-1, :] = 65535
    d0[:, -1] = 65535
    xs = np.arange(width, dtype=np.float32)[None, :]
    ys = np.arange(height, dtype=np.float32)[:, None]
    lum = (np.float32(255 / 2.0) * (np.float32(1) + np.sin(np.float32(0.15) * xs + np.float32(0.5) * np.sin(np.float32(0.25) * ys)))).astype(np.float32)
    c0 = lum.astype(np.int32).astype(np.uint8)     # + rand() % 1 == 0 (:188)
    rgb0 = np.repeat(c0[:, :, None], 3, axis=2)
    T0 = bso.se3_exp([0.1, 0.2, 0.3, 0.4, 0.5, 0.6])
    kf0 = scene.add_keyframe_from_images(d0, rgb0, T0)
    kf0.normals[:, :] = 0                            # ImageSpaceNormalToU16(0, 0) (:206-213)
    # keyframe 1: the same content shifted by `off` px, depth + noise (:217-227); u16 wrap-around of the
    # column that reads keyframe 0's invalid right border is kept.
    r = rng.integers(0, 100, size=(height, width)).astype(np.float32)
    d1 = d0.copy()
    c1 = c0.copy()
    shifted = d0[:, off:].astype(np.float32) + (np.float32(0.0001) * r[:, :width - off]) / raw_to_float_depth
    d1[:, :width - off] = shifted.astype(np.uint32).astype(np.uint16)
    c1[:, :width - off] = c0[:, off:]
    d1[:, 0] = 65535
    d1[0, :] = d1[-1, :] = 65535
    d1[:, width - off:] = 65535
    c1[:, 0] = c0[:, 0]
    c1[0, :], c1[-1, :] = c0[0, :], c0[-1, :]
    rgb1 = np.repeat(c1[:, :, None], 3, axis=2)
    rel = bso.se3_identity()
    rel.t[0] = float(kdepth * np.float32(off) / np.float32(cam.fx))
    T1 = bso.se3_mul(T0, rel)
    kf1 = scene.add_keyframe_from_images(d1, rgb1, T1)
    kf1.normals[:, :] = 0
    scene.create_surfels_for_keyframe(kf1)
    return scene, kf0, kf1, float(kdepth)


def offsets_13(translation_offset, rotation_offset):
    """The 13 start offsets of BS/test/test_pose_optimization_geometric_residual.cc:136-152."""
    out = [bso.se3_identity()]
    for sign in (1.0, -1.0):
        for i in range(6):
            x = np.zeros(6, np.float32)
            x[i] = sign * (translation_offset if i < 3 else rotation_offset)
            out.append(bso.se3_exp(x))
    return out


# ----------------------------------------------------------------------------- synthetic stacks

def random_planes(rng, count):
    """20 random planes n = (u1, u2, -1)/|.|, offset 2.5 m
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:259-267)."""
    planes = []
    for _ in range(count):
        n = rng.uniform(-1, 1, 3).astype(np.float32)
        n[2] = -1.0
        n /= np.linalg.norm(n)
        planes.append(n)
    return planes


def render_planes(cam, width, height, T_global_frame_R, T_global_frame_t, planes, offset=2.5):
    """Nearest positive intersection of each pixel ray with the plane set {n.x + offset = 0}."""
    dxg, dyg = unproject_dirs(cam, width, height)
    d = np.stack([dxg, dyg, np.ones_like(dxg)], axis=-1).astype(np.float64)   # frame-space directions
    dg = d @ T_global_frame_R.T.astype(np.float64)
    o = T_global_frame_t.astype(np.float64)
    best = np.full((height, width), np.inf)
    best_plane = np.zeros((height, width), np.int32)
    for pi, n in enumerate(planes):
        n64 = n.astype(np.float64)
        denom = dg @ n64
        num = -(offset + o @ n64)
        with np.errstate(divide="ignore", invalid="ignore"):
            t = num / denom
        t = np.where((denom < 0) & (t > 0.3), t, np.inf)   # front-facing, in front of the camera
        upd = t < best
        best = np.where(upd, t, best)
        best_plane = np.where(upd, pi, best_plane)
    return best, best_plane, dg, o


def rotation_from_log(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)


def texture_at(points, planes_idx, seed_phase):
    """Smooth deterministic world-space texture (sum of sinusoids), 0..255, photo-consistent across views."""
    x, y, z = points[..., 0], points[..., 1], points[..., 2]
    v = (np.sin(7.0 * x + seed_phase) + np.sin(9.0 * y + 1.3 * seed_phase) + np.sin(11.0 * z + 0.7)
         + np.sin(23.0 * (x + y)) * 0.5 + np.cos(17.0 * (y - z)) * 0.5)
    v = (v + 4.0) / 8.0
    return np.clip(v * 255.0 + 0.5 * planes_idx, 0, 255).astype(np.uint8)


def synthetic_scene(num_keyframes, seed=0xBAD51A4, width=W, height=H, cell=4, max_surfels=None,
                    use_depth_residuals=True, use_descriptor_residuals=False,
                    translation_range=0.25, rotation_range=0.12, depth_max_m=6.0,
                    camera=None, tex_mode=abi.TEX_FIXED_POINT_1_8, color_scale=1):
    """Synthetic 640x480 keyframe stack of SURVEY.md 8(d): random planes, photo-consistent
    texture, keyframe poses T0 * exp(xi); surfels created keyframe by keyframe at `cell`.
    color_scale = s > 1: the colour camera has s times the resolution of the depth camera (same pose and field of view)."""
    rng = np.random.default_rng(seed)
    cam = camera or bso.make_camera(525.0, 525.0, 319.5 + 0.5, 239.5 + 0.5, width, height)
    color_cam = cam if color_scale == 1 else bso.make_camera(cam.fx * color_scale, cam.fy * color_scale, cam.cx * color_scale, cam.cy * color_scale,
                                                             width * color_scale, height * color_scale)
    raw_to_float_depth = np.float32(1.0 / 5000)
    if max_surfels is None:
        max_surfels = ((width - 1) // cell + 1) * ((height - 1) // cell + 1) * num_keyframes
    scene = bso.HostScene(color_cam, cam, float(raw_to_float_depth), 40.0, cell, max_surfels,
                          use_depth_residuals=use_depth_residuals, use_descriptor_residuals=use_descriptor_residuals,
                          tex_mode=tex_mode)
    planes = random_planes(rng, 20)
    for k in range(num_keyframes):
        xi_t = rng.uniform(-translation_range, translation_range, 3)
        xi_r = rng.uniform(-rotation_range, rotation_range, 3)
        T = bso.se3_exp(np.concatenate([xi_t, xi_r]).astype(np.float32))
        M = np.array(list(bso.se3_matrix3x4(T).m), np.float64).reshape(3, 4)
        R, t = M[:, :3], M[:, 3]
        tt, pidx, dg, o = render_planes(cam, width, height, R, t, planes)
        z = tt  # directions have unit z in the frame, so the ray parameter is the frame-space depth
        valid = np.isfinite(z) & (z < depth_max_m)
        depth = np.where(valid, z / float(raw_to_float_depth) + 0.5, 65535).astype(np.uint32)
        depth = np.where(depth >= 32768, 65535, depth).astype(np.uint16)
        if color_scale == 1:
            pts = o[None, None, :] + dg * np.where(valid, tt, 0.0)[..., None]
            lum = texture_at(pts, pidx, 0.37)
        else:   # the same surface seen through the finer colour pixel grid
            ctt, cpidx, cdg, co = render_planes(color_cam, width * color_scale, height * color_scale, R, t, planes)
            cvalid = np.isfinite(ctt)
            cpts = co[None, None, :] + cdg * np.where(cvalid, ctt, 0.0)[..., None]
            lum = texture_at(cpts, cpidx, 0.37)
        rgb = np.repeat(lum[:, :, None], 3, axis=2)
        kf = scene.add_keyframe_from_images(depth, rgb, T)
        scene.create_surfels_for_keyframe(kf)
    return scene


def intrinsics_test_camera(width=W, height=H):
    """True camera of the two intrinsics tests: {0.5h, 0.45h, 0.5w - 0.5, 0.5h - 0.5}
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:378, ..._photometric_residual.cc:112)."""
    return bso.make_camera(0.5 * height, 0.45 * height, 0.5 * width - 0.5, 0.5 * height - 0.5, width, height)


def distorted_camera(cam, scale=1.0):
    """Start estimate of the intrinsics tests: the true camera + (+0.5, -0.6, +1.23, -2.17) px
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:416, ..._photometric_residual.cc:150); `scale` shrinks the
    offsets with the image for the reduced-size CPU runs."""
    return bso.make_camera(cam.fx + 0.5 * scale, cam.fy - 0.6 * scale, cam.cx + 1.23 * scale, cam.cy - 2.17 * scale, cam.width, cam.height)


def intrinsics_scene(num_keyframes, seed=0, width=W, height=H, cell=2, max_surfels=1000 * 1000, use_descriptor_residuals=False,
                    photometric=False, distortion=None, create_surfels=True, camera=None, filter_new_surfels=False,
                    min_observation_count=2, covisibility=None):
    """Scene of {Alternating,PCG}IntrinsicsOptimizationWithGeometricResidual
    (BS/test/test_intrinsics_optimization_geometric_residual.cc:371-560): 20 random planes rendered
    from `num_keyframes` poses global_T_0 * exp(xi) (:286-296), undistorted depth, cell size 2; surfels
    are created from every keyframe with the true camera.  (The reference filters new surfels by
    observation count, :516; creation is unfiltered here.)
    photometric=True gives the scene of {Alternating,PCG}IntrinsicsOptimizationWithPhotometricResidual
    (BS/test/test_intrinsics_optimization_photometric_residual.cc:25-160): border pixels keep their depth,
    the colour is a world-space sinusoid texture (:50-53), descriptor residuals only.
    distortion=(a, cfactor) distorts the rendered depth like TestDepthDeformationOptimizationWithGeometricResidual
    (:111-125: the inverse of the depth deformation model through the Lambert W function); create_surfels=False
    leaves surfel creation to the BA under test (do_surfel_updates).
    camera: the true camera -- intrinsics_test_camera() (fy = 0.45 h) for the two intrinsics tests, the default
    reference_test_camera() (fy = 0.5 h) for the depth-deformation test (:187).  filter_new_surfels=True creates the surfels
    like the tests do (CreateSurfelsForKeyframe(stream, true, keyframe), :516 / :217, after ALL keyframes were added, so every
    keyframe's co-visibility list is complete) with the constructor's min_observation_count of 2; covisibility = {keyframe id:
    [ids]} (e.g. the host class's frustum-intersection lists, BS/direct_ba.cc:231-249), default: every other keyframe."""
    rng = np.random.default_rng(seed)
    cam = camera or reference_test_camera(width, height)
    raw_to_float_depth = np.float32(1.0 / 1000)
    scene = bso.HostScene(cam, cam, float(raw_to_float_depth), 40.0, cell, max_surfels,
                          use_depth_residuals=not photometric, use_descriptor_residuals=use_descriptor_residuals or photometric)
    planes = random_planes(rng, 20)
    T0 = bso.se3_exp([0.01, 0.02, 0.03, 0.004, 0.005, 0.006])
    dxg, dyg = unproject_dirs(cam, width, height)
    d = np.stack([dxg, dyg, np.ones_like(dxg)], axis=-1).astype(np.float64)
    for k in range(num_keyframes):
        xi = np.concatenate([3.0 * (rng.integers(0, 200, 3) / 200.0 - 0.5), 3.5 * ((rng.integers(0, 200, 3) - 100) / 500.0)]).astype(np.float32)
        T = bso.se3_mul(T0, bso.se3_exp(xi))
        M = np.array(list(bso.se3_matrix3x4(T).m), np.float64).reshape(3, 4)
        dg = d @ M[:, :3].T
        o = M[:, 3]
        best = np.zeros((height, width))
        for n in planes:   # RenderPlanes (:134-160): nearest positive ray parameter, no facing test
            n64 = n.astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                z = -(2.5 + o @ n64) / (dg @ n64)
            upd = (z > 0) & ((best == 0) | (z < best))
            best = np.where(upd, z, best)
        depth = np.full((height, width), 65535, np.uint16)
        if distortion is not None:
            from scipy.special import lambertw
            ta, tc = float(distortion[0]), float(distortion[1])
            with np.errstate(divide="ignore", invalid="ignore"):
                wv = lambertw(-ta * tc * np.exp(-ta / best)).real
                best = np.where(best == 0, 0.0, np.float32(1.0) / ((ta + best * wv) / (ta * best)).astype(np.float32))
        inner = np.minimum(65535, best[1:-1, 1:-1] / float(raw_to_float_depth) + 0.5)
        depth[1:-1, 1:-1] = np.where(best[1:-1, 1:-1] == 0, 65535, inner).astype(np.uint16)
        rgb = np.zeros((height, width, 3), np.uint8)
        if photometric:
            depth = np.where(best == 0, 65535, np.minimum(65535, best / float(raw_to_float_depth) + 0.5)).astype(np.uint16)
            g = (o[None, None, :] + dg * best[..., None]).astype(np.float32)
            f = np.float32(200)

            def chan(p, q):
                v = np.float32(255 / 2.0) * (np.float32(1) + np.sin(np.float32(0.15) * f * p + np.float32(0.5) * np.sin(np.float32(0.25) * f * q)))
                return v.astype(np.int32).astype(np.uint8)
            col = np.stack([chan(g[..., 0], g[..., 1]), chan(g[..., 1], g[..., 2]), chan(g[..., 2], g[..., 0])], axis=-1)
            rgb = np.where((best > 0)[..., None], col, 0).astype(np.uint8)
        kf = scene.add_keyframe_from_images(depth, rgb, T)
        if create_surfels and not filter_new_surfels:
            scene.create_surfels_for_keyframe(kf)
    if create_surfels and filter_new_surfels:
        for kf in scene.keyframes:
            others = [o for o in scene.keyframes if o is not kf] if covisibility is None else [scene.keyframes[j] for j in covisibility[kf.id]]
            scene.create_surfels_for_keyframe_ex(kf, True, min_observation_count, others)
    return scene
