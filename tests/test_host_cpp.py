"""C++ host layer (instancefusion_amd/host/ifx_host.hpp, ifx_replay): the readers against the Python readers on CPU; the replay
program against the Python main loop on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SMALL

HOST = os.path.join(ROOT, "instancefusion_amd", "host")
REPLAY = os.path.join(ROOT, "instancefusion_amd", "ifx_replay")
LIBDIR = os.path.join(ROOT, "instancefusion_amd")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostcpp") / "host_readers_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", HOST,
                    os.path.join(ROOT, "tests", "cpp", "host_readers_check.cpp"), "-L", LIBDIR, "-lifx", "-lz", f"-Wl,-rpath,{LIBDIR}", "-o", exe], check=True)
    return exe


def _frames(blob, w, h):
    rec = 8 + w * h * 2 + w * h * 3
    assert len(blob) % rec == 0
    out = []
    for k in range(len(blob) // rec):
        b = blob[k * rec:(k + 1) * rec]
        out.append((int(np.frombuffer(b[:8], np.int64)[0]), np.frombuffer(b[8:8 + w * h * 2], np.uint16).reshape(h, w),
                    np.frombuffer(b[8 + w * h * 2:], np.uint8).reshape(h, w, 3)))
    return out


@pytest.mark.parametrize("depth_mode", ["raw", "zlib"])
def test_raw_log_reader_equals_python(checker, tmp_path, depth_mode):
    from instancefusion_amd import logio

    w, h, n = 64, 48, 5
    rng = np.random.default_rng(5)
    klg = str(tmp_path / "a.klg")
    wr = logio.RawLogWriter(klg, depth=depth_mode, image="raw")
    for k in range(n):
        wr.add(1000 * k + 7, rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 8000, (h, w), dtype=np.uint16))
    wr.close()
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([checker, "klg", klg, str(w), str(h), out], capture_output=True, text=True, check=True)
    assert int(r.stdout) == n
    got = _frames(open(out, "rb").read(), w, h)
    rd = logio.RawLogReader(klg, w, h)
    k = 0
    while rd.hasMore():
        rd.getNext()
        assert got[k][0] == rd.timestamp and np.array_equal(got[k][1], rd.depth) and np.array_equal(got[k][2], rd.rgb)
        k += 1
    assert k == len(got) == n - 1        # the last frame is never delivered (RawLogReader.cpp:134-137)


def _test_image(w, h, seed):
    """smooth structure + texture + hard edges + saturated colours: exercises AC runs, EOB, ZRL, chroma upsampling edges and clamping"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 9.0 + seed) * np.cos(y / 13.0), 127 + 120 * np.cos(x / 5.0) * np.sin(y / 7.0 + 1), 255.0 * ((x // 16 + y // 16) % 2)], -1)
    img += rng.normal(0, 12, img.shape)
    img[: h // 5, : w // 4] = (255, 0, 0)
    img[-h // 6:, -w // 3:] = (0, 0, 255)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,quality,subsampling", [(64, 48, 90, 2), (640, 480, 95, 2), (61, 43, 75, 2), (33, 17, 100, 2), (50, 30, 85, 1), (47, 29, 60, 1),
                                                     (40, 40, 92, 0), (8, 8, 90, 2), (1, 1, 90, 2), (17, 9, 30, 2)])
def test_jpeg_decoder_equals_libjpeg(checker, tmp_path, w, h, quality, subsampling):
    """ifx_jpeg.hpp against libjpeg (through PIL): every pixel identical, for 4:2:0 / 4:2:2 / 4:4:4, odd sizes, low and high quality."""
    from PIL import Image

    img = _test_image(w, h, w + h + quality)
    f = str(tmp_path / "a.jpg")
    Image.fromarray(img).save(f, format="JPEG", quality=quality, subsampling=subsampling)
    ref = np.asarray(Image.open(f).convert("RGB"))
    out = str(tmp_path / "a.bin")
    subprocess.run([checker, "jpg", f, out], check=True)
    b = open(out, "rb").read()
    assert tuple(np.frombuffer(b[:8], np.int32)) == (w, h)
    got = np.frombuffer(b[8:], np.uint8).reshape(h, w, 3)
    assert np.array_equal(got, ref), (np.abs(got.astype(int) - ref.astype(int)).max(), (got != ref).mean())


def test_jpeg_decoder_grey_restart_and_refusals(checker, tmp_path):
    from PIL import Image

    img = _test_image(70, 50, 3)
    f, out = str(tmp_path / "g.jpg"), str(tmp_path / "g.bin")
    Image.fromarray(img[..., 0]).save(f, format="JPEG", quality=88)                      # single component
    subprocess.run([checker, "jpg", f, out], check=True)
    got = np.frombuffer(open(out, "rb").read()[8:], np.uint8).reshape(50, 70, 3)
    assert np.array_equal(got, np.asarray(Image.open(f).convert("RGB")))
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):             # restart intervals (DRI / RSTn)
        Image.fromarray(img).save(f, format="JPEG", quality=88, **kw)
        assert b"\xff\xdd" in open(f, "rb").read()
        subprocess.run([checker, "jpg", f, out], check=True)
        got = np.frombuffer(open(out, "rb").read()[8:], np.uint8).reshape(50, 70, 3)
        assert np.array_equal(got, np.asarray(Image.open(f).convert("RGB"))), kw
    Image.fromarray(img).save(f, format="JPEG", quality=88, progressive=True)            # progressive: refused, loudly
    r = subprocess.run([checker, "jpg", f, out], capture_output=True, text=True)
    assert r.returncode == 1 and "progressive" in r.stderr
    open(f, "wb").write(b"not a jpeg at all")
    assert subprocess.run([checker, "jpg", f, out], capture_output=True).returncode == 1


def test_raw_log_reader_decodes_jpeg_frames(checker, tmp_path):
    """The usual .klg (zlib depth + JPEG colour, what ElasticFusion's logger writes): same frames as the Python reader, which decodes with PIL."""
    from instancefusion_amd import logio

    w, h = 64, 48
    klg = str(tmp_path / "j.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg", jpeg_quality=90)
    for k in range(4):
        wr.add(1000 * k, _test_image(w, h, k), np.full((h, w), 1000 + k, np.uint16))
    wr.close()
    out = str(tmp_path / "o.bin")
    subprocess.run([checker, "klg", klg, str(w), str(h), out], check=True)
    got = _frames(open(out, "rb").read(), w, h)
    rd = logio.RawLogReader(klg, w, h)
    k = 0
    while rd.hasMore():
        rd.getNext()
        assert got[k][0] == rd.timestamp and np.array_equal(got[k][1], rd.depth) and np.array_equal(got[k][2], rd.rgb), k
        k += 1
    assert k == 3


def test_png_log_reader_equals_python(checker, tmp_path):
    from PIL import Image

    from instancefusion_amd import logio

    w, h, n = 40, 30, 4
    rng = np.random.default_rng(9)
    lines = []
    for k in range(n):
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        dep = rng.integers(0, 60000, (h, w), dtype=np.uint16)
        if k == 1:
            rgb[:] = np.arange(w, dtype=np.uint8)[None, :, None]          # smooth rows: the encoder picks Sub / Up / Paeth filters
            dep[:] = (np.arange(h, dtype=np.uint16) * 50)[:, None]
        Image.fromarray(rgb).save(tmp_path / f"{k}_color.png")
        Image.fromarray(dep).save(tmp_path / f"{k}_depth.png")
        lines.append(f"{100 + k} {k}_depth.png {k}_color.png {k} {k}")
    (tmp_path / "data.txt").write_text("\n".join(lines) + "\n")
    out = str(tmp_path / "dump.bin")
    subprocess.run([checker, "png", str(tmp_path / "data.txt"), str(w), str(h), out], check=True)
    got = _frames(open(out, "rb").read(), w, h)
    rd = logio.PNGLogReader(str(tmp_path / "data.txt"), w, h)
    k = 0
    while rd.hasMore():
        rd.getNext()
        assert got[k][0] == rd.timestamp and np.array_equal(got[k][1], rd.depth) and np.array_equal(got[k][2], rd.rgb)
        k += 1
    assert k == len(got) and k >= n - 1


@pytest.mark.parametrize("compressed", [False, True])
def test_mask_replay_reads_npz(checker, tmp_path, compressed):
    w, h = 64, 48
    rng = np.random.default_rng(3)
    masks = (rng.random((3, h, w)) > 0.5).astype(np.uint8) * 255
    cls = np.array([57, 1, 63], np.int64 if compressed else np.int32)
    (np.savez_compressed if compressed else np.savez)(tmp_path / "000012.npz", masks=masks, class_ids=cls)
    out = str(tmp_path / "m.bin")
    subprocess.run([checker, "npz", str(tmp_path), "12", str(w), str(h), out], check=True)
    b = open(out, "rb").read()
    n = int(np.frombuffer(b[:4], np.int32)[0])
    assert n == 3
    assert np.array_equal(np.frombuffer(b[4:16], np.int32), cls.astype(np.int32))
    assert np.array_equal(np.frombuffer(b[16:], np.uint8).reshape(3, h, w), masks)
    subprocess.run([checker, "npz", str(tmp_path), "13", str(w), str(h), out], check=True)     # no file for that frame
    assert int(np.frombuffer(open(out, "rb").read()[:4], np.int32)[0]) == -1
    r = subprocess.run([checker, "npz", str(tmp_path), "12", "32", "48", out], capture_output=True, text=True)    # wrong frame size
    assert r.returncode == 1 and "size" in r.stderr


def test_quaternion_equals_python(checker, tmp_path):
    from instancefusion_amd import logio

    out = str(tmp_path / "q.bin")
    subprocess.run([checker, "quat", out], check=True)
    q = np.frombuffer(open(out, "rb").read(), np.float32).reshape(3, 4)
    Rs = [np.eye(3), np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]]), np.array([[-1, 0, 0], [0, -0.6, 0.8], [0, 0.8, 0.6]])]
    for k, R in enumerate(Rs):
        assert np.allclose(q[k], np.array(logio._quaternion(R.astype(np.float32)), np.float32), atol=1e-7)


def test_class_surface_compiles_and_refuses_without_gpu(checker):
    import torch

    r = subprocess.run([checker, "api"], capture_output=True, text=True, check=True)
    if torch.cuda.is_available():
        assert r.stdout.startswith("created")
    else:
        assert r.stdout.startswith("refused") and "no CPU fallback" in r.stdout


def test_replay_program_built_and_fails_loudly_without_gpu(tmp_path):
    import torch

    from instancefusion_amd import logio

    assert os.path.exists(REPLAY), "make -C instancefusion_amd/csrc builds instancefusion_amd/ifx_replay"
    assert subprocess.run([REPLAY, "--help"], capture_output=True).returncode == 0
    if torch.cuda.is_available():
        pytest.skip("GPU present: the replay itself is covered by the gpu test")
    klg = str(tmp_path / "t.klg")
    wr = logio.RawLogWriter(klg)
    for k in range(3):
        wr.add(k, np.zeros((48, 64, 3), np.uint8), np.full((48, 64), 1000, np.uint16))
    wr.close()
    r = subprocess.run([REPLAY, klg, "--width", "64", "--height", "48"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_cpp_replay_equals_python_main_loop(small_stream, tmp_path):
    """ifx_replay (C++ classes of ifx_host.hpp) and tools/run_log.py (Python mirror) over the same .klg and masks: identical
    trajectory file, identical PLY models."""
    import importlib.util

    from instancefusion_amd import logio, synth

    st = small_stream
    n = 16                                                    # long enough for surfels to become stable (confidence > 10) and take labels
    src = [i if i < 10 else 18 - i for i in range(n + 1)]     # the 10-frame stream forth and back
    klg = str(tmp_path / "s.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg", jpeg_quality=95)       # what ElasticFusion's logger writes; both readers decode identically
    for i in range(n + 1):
        wr.add(33333 * i, st["rgb"][src[i]], st["depth"][src[i]])
    wr.close()
    mdir = tmp_path / "masks"
    mdir.mkdir()
    for i in range(n):
        mk, cl = synth.canned_masks(st["obj"][src[i]], st["scene"])
        (np.savez_compressed if i % 2 else np.savez)(mdir / f"{i:06d}.npz", masks=mk, class_ids=cl)
    common = ["--width", str(SMALL["w"]), "--height", str(SMALL["h"]), "--fx", str(SMALL["fx"]), "--fy", str(SMALL["fy"]), "--cx", str(SMALL["cx"]),
              "--cy", str(SMALL["cy"]), "--max-surfels", "400000", "--masks", str(mdir), "--flann-every", "2", "--confidence", "2"]
    out_c = str(tmp_path / "C")
    r = subprocess.run([REPLAY, klg] + common + ["--out", out_c, "--labels", out_c + ".labels"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"{n} frames" in r.stdout and " 0 segmentation calls" not in r.stdout
    spec = importlib.util.spec_from_file_location("run_log", os.path.join(ROOT, "tools", "run_log.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out_p = str(tmp_path / "P")
    assert mod.main([klg] + common + ["--out", out_p, "--labels", out_p + ".labels"]) == 0
    assert open(out_c + ".freiburg").read() == open(out_p + ".freiburg").read()
    for suffix in (".ply", "_Instance.ply"):
        assert open(out_c + suffix, "rb").read() == open(out_p + suffix, "rb").read(), suffix
    lab = np.fromfile(out_c + ".labels", np.int32)
    assert lab.size > 0 and np.array_equal(lab, np.fromfile(out_p + ".labels", np.int32))
    head = open(out_c + ".ply", "rb").read(200)
    assert int(head.split(b"element vertex ")[1].split(b"\n")[0]) > 1000          # stable surfels were exported


@pytest.mark.gpu
def test_cpp_replay_png_log_equals_klg(small_stream, tmp_path):
    """The same frames as a data.txt image list (IF/utilities/PNGLogReader.cpp) and as a .klg: identical trajectories from ifx_replay.
    (The PNG reader delivers every frame, the .klg reader never its last one: the .klg gets one more frame.)"""
    from PIL import Image

    from instancefusion_amd import logio

    st = small_stream
    n = 6
    klg = str(tmp_path / "s.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="raw")
    lines = []
    for i in range(n + 1):
        k = min(i, n - 1)
        wr.add(33333 * i, st["rgb"][k], st["depth"][k])
        if i < n:
            Image.fromarray(st["rgb"][k]).save(tmp_path / f"{i}_color.png")
            Image.fromarray(st["depth"][k]).save(tmp_path / f"{i}_depth.png")
            lines.append(f"{33333 * i} {i}_depth.png {i}_color.png {i} {i}")
    wr.close()
    (tmp_path / "data.txt").write_text("\n".join(lines) + "\n")
    common = ["--width", str(SMALL["w"]), "--height", str(SMALL["h"]), "--fx", str(SMALL["fx"]), "--fy", str(SMALL["fy"]), "--cx", str(SMALL["cx"]),
              "--cy", str(SMALL["cy"]), "--max-surfels", "400000"]
    outs = []
    for src, tag in ((klg, "K"), (str(tmp_path / "data.txt"), "P")):
        out = str(tmp_path / tag)
        r = subprocess.run([REPLAY, src] + common + ["--out", out], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and f"{n} frames" in r.stdout, (r.stdout, r.stderr)
        outs.append(open(out + ".freiburg").read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == n
