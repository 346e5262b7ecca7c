"""CPU: the file formats on either side of the BA path (badslam_amd/host/io.*): PNG decoding,
the TUM RGB-D associated + calibrated reader, pose export and the calibration export / import.  The fixtures are
written by this test (stdlib zlib PNG encoder); the expectations follow the reference's reader / writer code."""
import struct
import zlib

import numpy as np
import pytest

from badslam_amd import direct_ba as dba


def write_png(path, img, filter_type=0):
    """Minimal PNG writer: (h, w) uint16 gray or (h, w, 3|4) uint8.  filter_type 0-4 is applied to every scanline."""
    img = np.ascontiguousarray(img)
    if img.dtype == np.uint16:
        h, w = img.shape
        depth, ctype, bpp = 16, 0, 2
        rows = img.astype(">u2").tobytes()
    else:
        h, w, ch = img.shape
        depth, ctype, bpp = 8, {3: 2, 4: 6}[ch], ch
        rows = img.tobytes()
    stride = w * bpp
    raw = np.frombuffer(rows, np.uint8).reshape(h, stride).astype(np.int32)
    out = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = raw[y]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if filter_type == 0:
            f = cur
        elif filter_type == 1:
            f = cur - left
        elif filter_type == 2:
            f = cur - prev
        elif filter_type == 3:
            f = cur - (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        out.append(filter_type)
        out += (f & 0xFF).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    comp = zlib.compress(bytes(out), 6)
    half = len(comp) // 2   # two IDAT chunks: the decoder must concatenate them
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + chunk(b"IDAT", comp[:half]) +
                 chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("filter_type", [0, 1, 2, 3, 4])
def test_png_round_trip(tmp_path, filter_type):
    rng = np.random.default_rng(filter_type)
    depth = rng.integers(0, 65536, (37, 53), dtype=np.uint16)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    write_png(tmp_path / "d.png", depth, filter_type)
    write_png(tmp_path / "c.png", rgb, filter_type)
    write_png(tmp_path / "a.png", rgba, filter_type)
    assert np.array_equal(dba.read_png(tmp_path / "d.png"), depth)
    assert np.array_equal(dba.read_png(tmp_path / "c.png"), rgb)
    assert np.array_equal(dba.read_png(tmp_path / "a.png"), rgba[:, :, :3])       # alpha is dropped
    with pytest.raises(dba.DirectBAError):
        dba.read_png(tmp_path / "missing.png")


def test_tum_dataset_reader(tmp_path):
    (tmp_path / "rgb").mkdir()
    (tmp_path / "depth").mkdir()
    rng = np.random.default_rng(0)
    for name in ("1.0", "1.5", "2.0"):
        write_png(tmp_path / "rgb" / f"{name}.png", rng.integers(0, 256, (24, 32, 3), dtype=np.uint8))
        write_png(tmp_path / "depth" / f"{name}.png", rng.integers(0, 65536, (24, 32), dtype=np.uint16))
    (tmp_path / "calibration.txt").write_text("525.0 526.0 319.5 239.5\n")
    (tmp_path / "associated.txt").write_text("# comment\n1.000000 rgb/1.0.png 1.010000 depth/1.0.png\n\n1.500000 rgb/1.5.png 1.510000 depth/1.5.png\n"
                                             "2.000000 rgb/2.0.png 2.010000 depth/2.0.png\n")
    # 90 degrees about z between t = 1 and t = 2, translation 0 -> (2, 4, 6)
    s = np.sqrt(0.5)
    (tmp_path / "groundtruth.txt").write_text(f"# ts tx ty tz qx qy qz qw\n1.0 0 0 0 0 0 0 1\n2.0 2 4 6 0 0 {s:.17g} {s:.17g}\n")
    ds = dba.read_tum_dataset(tmp_path, "groundtruth.txt")
    assert (ds["width"], ds["height"]) == (32, 24)
    assert np.allclose(ds["camera"], [525.0, 526.0, 320.0, 240.0])                 # cx, cy + 0.5 (LV/rgbd_video_io_tum_dataset.h:232-233)
    assert [f["rgb_timestamp"] for f in ds["frames"]] == ["1.000000", "1.500000", "2.000000"]
    assert ds["frames"][1]["depth_path"].endswith("depth/1.5.png")
    mid = ds["frames"][1]["rgb_global_T_frame"]                                     # qx qy qz qw tx ty tz at t = 1.5: 45 degrees, half way
    assert np.allclose(mid[:4], [0, 0, np.sin(np.pi / 8), np.cos(np.pi / 8)], atol=1e-6)
    assert np.allclose(mid[4:], [1, 2, 3], atol=1e-6)
    assert np.allclose(ds["frames"][0]["rgb_global_T_frame"], [0, 0, 0, 1, 0, 0, 0])
    late = ds["frames"][2]["depth_global_T_frame"]                                  # beyond the last pose: clamped
    assert np.allclose(late, [0, 0, s, s, 2, 4, 6], atol=1e-6)
    assert np.array_equal(dba.read_png(ds["frames"][2]["depth_path"]).shape, (24, 32))
    no_traj = dba.read_tum_dataset(tmp_path)
    assert np.allclose(no_traj["frames"][1]["rgb_global_T_frame"], [0, 0, 0, 1, 0, 0, 0])
    with pytest.raises(dba.DirectBAError):
        dba.read_tum_dataset(tmp_path / "rgb")


def test_save_poses_format(tmp_path):
    s = np.sqrt(0.5)
    poses = np.array([[0, 0, s, s, 1, 2, 3], [0, 0, 0, 1, 0, 0, 0], [s, 0, 0, s, -1, 0.5, 0.25]], np.float32)
    dba.save_poses(["10.5", "11.25", "12.0"], poses, 0, tmp_path / "poses.txt")
    lines = (tmp_path / "poses.txt").read_text().splitlines()
    assert lines[0].startswith("# Format: Each line gives one global_T_frame pose")
    rows = [ln.split() for ln in lines[1:]]
    assert [r[0] for r in rows] == ["10.5", "11.25", "12.0"] and all(len(r) == 8 for r in rows)
    vals = np.array([[float(v) for v in r[1:]] for r in rows])
    assert np.allclose(vals[0], [0, 0, 0, 0, 0, 0, 1], atol=1e-7)                   # re-based: the start frame is the identity
    # frame 1 = start^-1 * identity: rotation -90 degrees about z, translation R^T * (-t)
    assert np.allclose(vals[1], [-2, 1, -3, 0, 0, -s, s], atol=1e-6)
    assert any(len(v.split(".")[-1]) >= 8 for v in rows[2][1:])                    # float values printed with 17 significant digits


def test_calibration_files_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    cf = rng.uniform(-0.01, 0.01, (6, 8)).astype(np.float32)
    d4, c4 = np.array([525.1, 526.2, 320.3, 240.4], np.float32), np.array([530.5, 531.6, 321.7, 241.8], np.float32)
    dba.save_calibration_arrays(tmp_path / "calib", d4, c4, 0.0123, cf)
    text = (tmp_path / "calib.depth_intrinsics.txt").read_text().split()
    assert np.allclose([float(t) for t in text], [525.1, 526.2, 319.8, 239.9], atol=1e-3)   # cx, cy written in the pixel-centre convention
    head = (tmp_path / "calib.deformation.txt").read_text().splitlines()
    assert head[0] == "8 6" and abs(float(head[1]) - 0.0123) < 1e-8 and len(head) == 2 + 48
    d, c, a, cf2 = dba.load_calibration_arrays(tmp_path / "calib", cf.shape)
    assert np.allclose(d, d4, atol=2e-4) and np.allclose(c, c4, atol=2e-4) and abs(a - 0.0123) < 1e-8   # 6 significant digits in the text files
    assert np.allclose(cf2, cf, rtol=1e-7, atol=1e-10)
    with pytest.raises(dba.DirectBAError):
        dba.load_calibration_arrays(tmp_path / "calib", (6, 9))                     # size mismatch (BS/io.cc:676-679)


def reference_state_file_bytes(rng, n_frames=5, n_kf=3, n_surfels=7, cf_w=4, cf_h=3, cf_stride_pad=8):
    """A version-1 state file written field by field in the order of SaveState (BS/io.cc:38-178) and
    BadSlamConfig::Save (BS/bad_slam_config.cc:47-96), independently of the C++ writer under test."""
    out = bytearray(b"BADSLAM" + bytes([1]))
    i32 = lambda v: out.extend(struct.pack("<i", v))
    u32 = lambda v: out.extend(struct.pack("<I", v))
    f32 = lambda v: out.extend(struct.pack("<f", v))
    b8 = lambda v: out.extend(bytes([1 if v else 0]))
    se3 = lambda: out.extend(rng.standard_normal(7).astype("<f4").tobytes())
    s = lambda t: (u32(len(t)), out.extend(t))
    i32(2)                                   # base_kf_id
    u32(2); se3(); se3()                      # motion model
    u32(1); i32(4); se3()                     # queued keyframes
    i32(4)                                   # last_frame_index
    f32(0.0002); i32(0); i32(100); f32(0.0); i32(30); i32(0); i32(0); f32(3.0); f32(40.0); i32(0); f32(3.0); f32(2.0); f32(0.005)
    i32(25000000); i32(4); f32(0.8); i32(1); i32(2); i32(2); i32(5); b8(1); i32(10); i32(10); b8(0); b8(1); b8(1); b8(0); i32(10); b8(1); b8(1); b8(0)
    b8(1); i32(250); b8(1); b8(1); s(b"/path/to/vocabulary.yml.gz"); s(b"pattern.yml"); f32(0.5); i32(640); i32(480)
    u32(n_frames)
    for _ in range(n_frames):
        se3()
    i32(1); i32(640); i32(480); i32(4); out.extend(np.array([525.0, 526.0, 320.0, 240.0], "<f4").tobytes()); i32(0)
    i32(1); i32(640); i32(480); i32(4); out.extend(np.array([524.0, 523.0, 321.0, 239.0], "<f4").tobytes())
    stride = cf_w * 4 + cf_stride_pad          # Image<float> rows may be padded: the reader must honour the stride
    i32(cf_w); i32(cf_h); i32(stride)
    cf = rng.uniform(-0.01, 0.01, (cf_h, cf_w)).astype("<f4")
    for y in range(cf_h):
        out.extend(cf[y].tobytes() + bytes(cf_stride_pad))
    f32(0.0125); f32(0.0002); f32(40.0); i32(4)
    i32(n_kf)
    for k in range(n_kf):
        if k == 1:
            i32(-1)                            # a deleted keyframe: id only
        else:
            i32(k); i32(k + 1); i32(k % 3); i32(7); i32(6)
    i32(n_surfels); i32(n_surfels)
    out.extend(rng.standard_normal((8, n_surfels)).astype("<f4").tobytes())
    i32(9); i32(8); b8(1); b8(0); i32(1); i32(2); i32(2); f32(0.8)
    return bytes(out), cf


def test_state_file_v1_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    data, cf = reference_state_file_bytes(rng, cf_stride_pad=0)
    (tmp_path / "in.state").write_bytes(data)
    ints, floats = dba.state_file_round_trip(tmp_path / "in.state", tmp_path / "out.state")
    assert list(ints) == [2, 4, 5, 3, 7, 7, 9, 4]
    assert np.allclose(floats, [0.0125, 0.0002, 40.0, 524.0, 523.0, 321.0, 239.0, 525.0, 526.0, 320.0, 240.0])
    assert (tmp_path / "out.state").read_bytes() == data                      # byte-identical: every field read and written in order
    # a padded cfactor stride is accepted on load and written back unpadded
    padded, _ = reference_state_file_bytes(np.random.default_rng(5), cf_stride_pad=8)
    (tmp_path / "padded.state").write_bytes(padded)
    dba.state_file_round_trip(tmp_path / "padded.state", tmp_path / "out2.state")
    assert (tmp_path / "out2.state").read_bytes() == data
    # corrupt headers / truncated files are rejected
    (tmp_path / "bad.state").write_bytes(b"BADSLAN" + data[7:])
    with pytest.raises(dba.DirectBAError):
        dba.state_file_round_trip(tmp_path / "bad.state", tmp_path / "x")
    (tmp_path / "v2.state").write_bytes(data[:7] + bytes([2]) + data[8:])
    with pytest.raises(dba.DirectBAError):
        dba.state_file_round_trip(tmp_path / "v2.state", tmp_path / "x")
    (tmp_path / "short.state").write_bytes(data[:-5])
    with pytest.raises(dba.DirectBAError):
        dba.state_file_round_trip(tmp_path / "short.state", tmp_path / "x")
