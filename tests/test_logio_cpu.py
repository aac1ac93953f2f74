"""Log I/O: .klg round trips in every encoding the reference reader accepts, the data.txt / PNG reader,
the .freiburg trajectory and the PLY export."""
import os
import struct

import numpy as np

from instancefusion_amd import logio


def _frames(n, w, h, seed=0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        depth = (1000 + 5 * xx + 3 * yy + 40 * i).astype(np.uint16)
        depth[rng.rand(h, w) < 0.05] = 0
        rgb = np.stack([(xx * 2 + i * 9) % 256, (yy * 3) % 256, ((xx + yy) // 2) % 256], -1).astype(np.uint8)
        out.append((1000000 * i + 17, rgb, depth))
    return out


def test_klg_round_trip_all_encodings(tmp_path):
    w, h = 64, 48
    fr = _frames(5, w, h)
    for dmode, imode in (("raw", "raw"), ("zlib", "raw"), ("zlib", "jpeg"), ("zlib", "none")):
        p = str(tmp_path / f"log_{dmode}_{imode}.klg")
        wr = logio.RawLogWriter(p, depth=dmode, image=imode, jpeg_quality=95)
        for ts, rgb, depth in fr:
            wr.add(ts, rgb, depth)
        wr.close()
        with open(p, "rb") as f:                       # header + first record, byte for byte (RawLogReader.cpp:27,59-62)
            assert struct.unpack("<i", f.read(4))[0] == 5
            ts0, dsz, isz = struct.unpack("<qii", f.read(16))
            assert ts0 == 17 and (dsz == w * h * 2) == (dmode == "raw") and (isz == w * h * 3) == (imode == "raw")
        rd = logio.RawLogReader(p, w, h)
        assert rd.getNumFrames() == 5
        got = 0
        while rd.hasMore():                            # the reference never delivers the last frame (currentFrame + 1 < numFrames)
            rd.getNext()
            ts, rgb, depth = fr[got]
            assert rd.timestamp == ts and np.array_equal(rd.depth, depth)
            if imode == "raw":
                assert np.array_equal(rd.rgb, rgb)
            elif imode == "jpeg":
                assert np.abs(rd.rgb.astype(int) - rgb.astype(int)).mean() < 12
            else:
                assert not rd.rgb.any()
            got += 1
        assert got == 4
        rd.getBack()                                   # re-reads the frame just delivered
        assert rd.timestamp == fr[3][0]
        rd.rewind(); rd.fastForward(2); rd.getNext()
        assert rd.timestamp == fr[2][0]
        flipped = logio.RawLogReader(p, w, h, flipColors=True)
        flipped.getNext()
        if imode == "raw":
            assert np.array_equal(flipped.rgb, fr[0][1][:, :, ::-1])
        rd.close(); flipped.close()


def test_png_list_reader(tmp_path):
    from PIL import Image

    w, h = 40, 30
    fr = _frames(3, w, h, seed=3)
    lines = []
    for i, (ts, rgb, depth) in enumerate(fr):
        Image.fromarray(rgb).save(tmp_path / f"{i:05d}-color.png")
        Image.fromarray(depth).save(tmp_path / f"{i:05d}-depth.png")
        lines.append(f"{ts} {i:05d}-depth.png {i:05d}-color.png {i:05d} {i:05d}")
    (tmp_path / "data.txt").write_text("\n".join(lines) + "\n")
    rd = logio.PNGLogReader(str(tmp_path / "data.txt"), w, h)
    assert rd.getNumFrames() == 3
    k = 0
    while rd.hasMore():
        rd.getNext()
        assert rd.timestamp == fr[k][0] and np.array_equal(rd.rgb, fr[k][1]) and np.array_equal(rd.depth, fr[k][2])
        assert not rd.has_depth_filled
        k += 1
    assert k == 3


def test_freiburg_and_ply(tmp_path):
    from scipy.spatial.transform import Rotation

    rng = np.random.RandomState(1)
    poses, ts = [], []
    for i in range(6):
        P = np.eye(4, dtype=np.float32)
        P[:3, :3] = Rotation.from_rotvec(rng.uniform(-2.5, 2.5, 3)).as_matrix().astype(np.float32)
        P[:3, 3] = rng.uniform(-2, 2, 3)
        poses.append(P); ts.append(1000000 * i + 250000)
    p = str(tmp_path / "traj.freiburg")
    logio.save_freiburg(p, ts, poses)
    rows = np.loadtxt(p)
    assert rows.shape == (6, 8) and np.allclose(rows[:, 0], np.array(ts) / 1e6)
    for r, P in zip(rows, poses):
        assert np.allclose(r[1:4], P[:3, 3], atol=1e-5)
        q = Rotation.from_matrix(P[:3, :3].astype(np.float64)).as_quat()      # x y z w
        assert min(np.abs(r[4:] - q).max(), np.abs(r[4:] + q).max()) < 1e-4
    n = 50
    m = dict(pc=rng.rand(n, 4).astype(np.float32) * 20, nr=rng.rand(n, 4).astype(np.float32),
             col=np.stack([rng.randint(0, 1 << 24, n), rng.randint(0, 1 << 24, n)], 1).astype(np.float32))
    for inst in (False, True):
        pp = str(tmp_path / ("m_Instance.ply" if inst else "m.ply"))
        kept = logio.save_ply(pp, m, confidence=10.0, instance=inst)
        raw = open(pp, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        assert b"element vertex %d" % kept in head and b"binary_little_endian" in head and kept == int((m["pc"][:, 3] > 10).sum())
        assert len(body) == kept * 31
        first = np.nonzero(m["pc"][:, 3] > 10)[0][0]
        x, y, z = struct.unpack("<fff", body[:12])
        assert (x, y, z) == tuple(m["pc"][first, :3])
        c = int(m["col"][first, 1 if inst else 0])
        assert tuple(body[12:15]) == ((c >> 16) & 255, (c >> 8) & 255, c & 255)
        nx, = struct.unpack("<f", body[15:19])
        assert nx == -m["nr"][first, 0]


def test_klg_through_the_oracle(orc, tmp_path, small_stream):
    """Plumbing configuration (BASELINE config 1 in miniature): a .klg written from the synthetic stream, read back and
    pushed through the CPU path gives the same poses as feeding the arrays directly."""
    from conftest import SMALL

    st = small_stream
    p = str(tmp_path / "synthetic.klg")
    wr = logio.RawLogWriter(p, depth="zlib", image="raw")
    for i in range(5):
        wr.add(33333 * i, st["rgb"][i], st["depth"][i])
    wr.close()
    rd = logio.RawLogReader(p, SMALL["w"], SMALL["h"])
    a, b = orc.Oracle(**SMALL, max_surfels=300000), orc.Oracle(**SMALL, max_surfels=300000)
    i = 0
    while rd.hasMore():
        rd.getNext()
        pa = a.process_frame(rd.rgb, rd.depth)
        pb = b.process_frame(st["rgb"][i], st["depth"][i])
        assert np.array_equal(pa, pb)
        i += 1
    assert i == 4
    a.close(); b.close(); rd.close()
