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
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", HOST,
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


@pytest.mark.parametrize("depth_mode", ["raw", "zlib", "zlib-ahead"])
def test_raw_log_reader_equals_python(checker, tmp_path, depth_mode):
    from instancefusion_amd import logio

    ahead = depth_mode.endswith("-ahead")          # records decoded ahead by worker threads: same frames, same order
    depth_mode = depth_mode.split("-")[0]
    w, h, n = 64, 48, 23 if ahead else 5
    rng = np.random.default_rng(5)
    klg = str(tmp_path / "a.klg")
    wr = logio.RawLogWriter(klg, depth=depth_mode, image="jpeg" if ahead else "raw")
    for k in range(n):
        wr.add(1000 * k + 7, rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 8000, (h, w), dtype=np.uint16))
    wr.close()
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([checker, "klg", klg, str(w), str(h), out] + (["6", "3"] if ahead else []), capture_output=True, text=True, check=True)
    assert int(r.stdout) == n
    got = _frames(open(out, "rb").read(), w, h)
    rd = logio.RawLogReader(klg, w, h)
    k = 0
    while rd.hasMore():
        rd.getNext()
        assert got[k][0] == rd.timestamp and np.array_equal(got[k][1], rd.depth) and np.array_equal(got[k][2], rd.rgb)
        k += 1
    assert k == len(got) == n - 1        # the last frame is never delivered (RawLogReader.cpp:134-137)


def test_raw_log_reader_get_back_under_read_ahead(checker, tmp_path):
    """getBack() while records are being decoded ahead: the read-ahead is dropped, the file position is put back behind the last delivered frame, and the
    sequence of frames is the one the synchronous reader delivers for the same calls."""
    from instancefusion_amd import logio

    w, h, n = 64, 48, 17
    rng = np.random.default_rng(9)
    klg = str(tmp_path / "b.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg")
    for k in range(n):
        wr.add(100 * k, rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 8000, (h, w), dtype=np.uint16))
    wr.close()
    outs = []
    for ahead in (["0", "0", "5"], ["6", "3", "5"]):
        out = str(tmp_path / f"dump{ahead[0]}.bin")
        subprocess.run([checker, "klg", klg, str(w), str(h), out] + ahead, check=True, capture_output=True, timeout=60)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 0
    ts = [f[0] for f in _frames(outs[0], w, h)]
    assert ts[4] == ts[5]                      # getBack re-delivers the frame it was called on


def test_raw_log_reader_peek_next_hands_out_what_get_next_publishes(checker, tmp_path):
    """LogReader::peekNext (the reader's read-ahead as the source of ifx_hint_next_frame): the buffers it names are the ones the next getNext() publishes -- same
    addresses (the announced-frame entry recognises its frame by them), same content -- for every frame; without read-ahead there is nothing to peek at."""
    from instancefusion_amd import logio

    w, h, n = 64, 48, 13
    rng = np.random.default_rng(4)
    klg = str(tmp_path / "p.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg")
    for k in range(n):
        wr.add(100 * k, rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 8000, (h, w), dtype=np.uint16))
    wr.close()
    plain = str(tmp_path / "plain.bin")
    subprocess.run([checker, "klg", klg, str(w), str(h), plain, "0", "0"], check=True, capture_output=True, timeout=60)
    for ahead, threads in (("6", "3"), ("2", "1"), ("0", "0")):
        out = str(tmp_path / f"peek{ahead}.bin")
        r = subprocess.run([checker, "klgpeek", klg, str(w), str(h), out, ahead, threads], check=True, capture_output=True, text=True, timeout=60)
        peeked, same, frames = (int(r.stdout.split()[k]) for k in (1, 3, 5))
        assert frames == n - 1                                      # (RawLogReader never delivers the last frame: RawLogReader.cpp:134-137)
        assert peeked == same == (frames if threads != "0" else 0), r.stdout
        assert open(out, "rb").read() == open(plain, "rb").read()   # what was peeked at is the stream the synchronous reader delivers


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


def test_jpeg_decoder_survives_malformed_input(tmp_path):
    """Memory safety of the hand-written decoder on hostile input (the .klg reader feeds it whatever the file holds): crafted headers --
    code lengths that are no prefix code, table selectors and sampling factors out of range, segments shorter than their fields, absurd
    sizes -- truncations and 3000 seeded random corruptions, under AddressSanitizer + UBSan on the CPU: every variant either decodes or
    is refused with an exception."""
    from PIL import Image

    exe = str(tmp_path / "jpeg_fuzz")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", HOST,
                    os.path.join(ROOT, "tests", "cpp", "jpeg_fuzz.cpp"), "-o", exe], check=True)
    for sub, q in ((2, 90), (1, 70), (0, 95)):
        f = str(tmp_path / f"v{sub}.jpg")
        Image.fromarray(_test_image(72, 56, 11 + sub)).save(f, format="JPEG", quality=q, subsampling=sub, restart_marker_blocks=4 if sub == 1 else 0)
        r = subprocess.run([exe, f, "1000"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout[-400:], r.stderr[-2000:])
        assert "refused" in r.stdout


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


def test_sharding_unique_id_refuses_a_stale_file(checker, tmp_path):
    """Sharding::uniqueId on a rank other than 0: a file carrying another run's nonce is never taken for this run's ncclUniqueId (ncclCommInitRank would hang on it); a file
    of this run is.  Without a nonce no clock is compared (a rank that started 2 s after rank 0 wrote the file used to refuse it): a file that appears or
    changes while the rank waits is taken at once, one that was already there only when it is still there, unchanged, after the grace period (3 s) in which rank 0 of a new
    run would have removed it."""
    import struct
    import time

    f = str(tmp_path / "id.bin")
    open(f, "wb").write(bytes([7] * 128) + struct.pack("<Q", 0))
    os.utime(f, (time.time() - 600, time.time() - 600))                       # already there, no nonce: not taken inside the grace period, whatever its age
    r = subprocess.run([checker, "shardid", f, "0", "1"], capture_output=True, text=True, check=True)
    assert r.stdout.startswith("refused"), r.stdout
    t0 = time.time()
    p = subprocess.Popen([checker, "shardid", f, "0", "6"], stdout=subprocess.PIPE, text=True)   # ... while rank 0 of this run replaces it: taken at once
    time.sleep(0.6)
    open(f + ".tmp", "wb").write(bytes([6] * 128) + struct.pack("<Q", 0)); os.replace(f + ".tmp", f)
    out, _ = p.communicate(timeout=30)
    assert out.strip() == "id 6" and time.time() - t0 < 2.5, (out, time.time() - t0)
    open(f, "wb").write(bytes([7] * 128) + struct.pack("<Q", 41))             # fresh, but another run's nonce
    r = subprocess.run([checker, "shardid", f, "42", "1"], capture_output=True, text=True, check=True)
    assert r.stdout.startswith("refused"), r.stdout
    open(f, "wb").write(bytes([9] * 128) + struct.pack("<Q", 42))
    os.utime(f, (time.time() - 600, time.time() - 600))                       # this run's nonce: the clock does not matter
    r = subprocess.run([checker, "shardid", f, "42", "1"], capture_output=True, text=True, check=True)
    assert r.stdout.strip() == "id 9", r.stdout
    open(f, "wb").write(bytes([5] * 128) + struct.pack("<Q", 0))              # no nonce in use, the file is there before the rank looks (a rank that starts late): taken after the grace period
    r = subprocess.run([checker, "shardid", f, "0", "6"], capture_output=True, text=True, check=True)
    assert r.stdout.strip() == "id 5", r.stdout
    os.utime(f, (time.time() - 3600, time.time() - 3600))                     # ... but never an hour-old leftover: rank 0 of this run may just be late -- keep waiting, then refuse
    r = subprocess.run([checker, "shardid", f, "0", "5"], capture_output=True, text=True, check=True)
    assert r.stdout.startswith("refused"), r.stdout
    open(f, "wb").write(bytes([5] * 100))                                     # truncated
    r = subprocess.run([checker, "shardid", f, "0", "1"], capture_output=True, text=True, check=True)
    assert r.stdout.startswith("refused"), r.stdout


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
    assert int(r.stdout.split(" frames announced ahead")[0].split()[-1]) >= n - 2, r.stdout   # the reader's read-ahead fed ifx_hint_next_frame
    # the fern data base ran inside every frame (findFrame in the callback, addFrame after it): keyframes were admitted, nothing is old enough to match,
    # and the frames are what the Python loop (no data base) computes
    assert int(r.stdout.split(" fern keyframes")[0].split()[-1]) >= 1 and " 0 fern matches" in r.stdout
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
    # the run above announced every next frame from the reader's read-ahead (ifx_hint_next_frame); frames handed over one at a time give the same files
    out_n = str(tmp_path / "N")
    r = subprocess.run([REPLAY, klg] + common + ["--out", out_n, "--no-lookahead"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("0 frames announced ahead") or "\n0 frames announced ahead" in r.stdout, r.stdout
    assert open(out_c + ".freiburg").read() == open(out_n + ".freiburg").read()
    assert open(out_c + ".ply", "rb").read() == open(out_n + ".ply", "rb").read()


@pytest.mark.gpu
def test_cpp_replay_sharded_world_of_one_equals_unsharded(small_stream, tmp_path):
    """The C++ host on the spatially sharded map: `ifx_replay --shard-ranks -1` (ElasticFusion constructed with a Sharding spec: ifx_comm_unique_id +
    ifx_owner_init_comm, then ifx_owner_process_frame / ifx_owner_process_segmentation -- every exchange a RCCL collective enqueued by libifx.so)
    against the unsharded replay without loop closing: identical trajectory file, PLY models and labels."""
    from instancefusion_amd import logio, synth

    st = small_stream
    n = 16                                                    # long enough for surfels to become stable (confidence > 2) and take labels
    src = [i if i < 10 else 18 - i for i in range(n + 1)]
    klg = str(tmp_path / "s.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="raw")
    for i in range(n + 1):
        wr.add(33333 * i, st["rgb"][src[i]], st["depth"][src[i]])
    wr.close()
    mdir = tmp_path / "masks"
    mdir.mkdir()
    for i in range(n):
        mk, cl = synth.canned_masks(st["obj"][src[i]], st["scene"])
        np.savez(mdir / f"{i:06d}.npz", masks=mk, class_ids=cl)
    common = ["--width", str(SMALL["w"]), "--height", str(SMALL["h"]), "--fx", str(SMALL["fx"]), "--fy", str(SMALL["fy"]), "--cx", str(SMALL["cx"]),
              "--cy", str(SMALL["cy"]), "--max-surfels", "400000", "--masks", str(mdir), "--flann-every", "2", "--confidence", "2", "--no-close-loops"]
    outs = {}
    for tag, extra in (("one", []), ("shard", ["--shard-ranks", "-1"])):
        out = str(tmp_path / tag)
        r = subprocess.run([REPLAY, klg] + common + extra + ["--out", out, "--labels", out + ".labels"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert f"{n} frames" in r.stdout and " 0 segmentation calls" not in r.stdout
        outs[tag] = out
    assert open(outs["one"] + ".freiburg").read() == open(outs["shard"] + ".freiburg").read()
    for suffix in (".ply", "_Instance.ply"):
        assert open(outs["one"] + suffix, "rb").read() == open(outs["shard"] + suffix, "rb").read(), suffix
    lab = np.fromfile(outs["one"] + ".labels", np.int32)
    assert lab.size > 0 and np.array_equal(lab, np.fromfile(outs["shard"] + ".labels", np.int32))


@pytest.mark.gpu
def test_cpp_replay_baseline_size_reference_defaults(tmp_path):
    """ElasticFusionInterface::ProcessFrame at the BASELINE size with the reference's own configuration -- 640x480, Init() defaults, i.e.
    closeLoops = true with the fern data base inside every frame (IF/map_interface/ElasticFusionInterface.cpp:43-45) -- through the C++
    classes of ifx_host.hpp (ifx_replay) against the Python main loop over the same .klg (zlib depth, JPEG colour) and masks: identical
    trajectory file, models and labels."""
    import importlib.util

    from instancefusion_amd import logio, synth

    W, H, n = 640, 480, 20
    K = dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    st = synth.make_stream(n + 1, W, H, noise=True, loop_len=90, **K)
    klg = str(tmp_path / "b.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="jpeg", jpeg_quality=95)
    for i in range(n + 1):
        wr.add(33333 * i, st["rgb"][i], st["depth"][i])
    wr.close()
    mdir = tmp_path / "masks"
    mdir.mkdir()
    for i in range(n):
        mk, cl = synth.canned_masks(st["obj"][i], st["scene"])
        np.savez_compressed(mdir / f"{i:06d}.npz", masks=mk, class_ids=cl)
    common = ["--max-surfels", "2000000", "--masks", str(mdir), "--confidence", "2"]       # intrinsics, resolution, closeLoops: the defaults
    out_c = str(tmp_path / "C")
    r = subprocess.run([REPLAY, klg] + common + ["--out", out_c, "--labels", out_c + ".labels"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert f"{n} frames" in r.stdout and " 0 segmentation calls" not in r.stdout and "fern keyframes" in r.stdout
    spec = importlib.util.spec_from_file_location("run_log", os.path.join(ROOT, "tools", "run_log.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out_p = str(tmp_path / "P")
    assert mod.main([klg] + common + ["--out", out_p, "--labels", out_p + ".labels"]) == 0
    assert open(out_c + ".freiburg").read() == open(out_p + ".freiburg").read()
    assert len(open(out_c + ".freiburg").read().splitlines()) == n
    for suffix in (".ply", "_Instance.ply"):
        assert open(out_c + suffix, "rb").read() == open(out_p + suffix, "rb").read(), suffix
    lab = np.fromfile(out_c + ".labels", np.int32)
    assert lab.size > 100000 and np.array_equal(lab, np.fromfile(out_p + ".labels", np.int32)) and (lab >= 0).sum() > 100


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


# ---------------------------------------------------------------- the fern data base (ifx_ferns.hpp) against oracle/orc_ferns.py, no GPU
def _fern_scene(rng, w, h, fx, fy, cx, cy, holes):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = rng.uniform(0.6, 2.4)
    z = (base + 0.3 * np.sin(xx / rng.uniform(2, 5) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(2, 5))).astype(np.float32)
    z[rng.random((h, w)) < holes] = 0
    verts = np.stack([(xx - cx) * z / fx, (yy - cy) * z / fy, z, np.ones_like(z)], axis=-1).astype(np.float32)
    verts[z == 0] = 0
    norms = np.zeros((h, w, 4), np.float32); norms[..., 2] = -1; norms[..., 3] = 1
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    img = ((img.astype(np.int32) + np.roll(img, 1, axis=1) + np.roll(img, 1, axis=0)) // 3).astype(np.uint8)
    return img, verts, norms


def test_fern_database_equals_restatement(checker, tmp_path):
    """Ferns::addFrame / findFrame (EF/Ferns.cpp): fern codes, co-occurrence search, keyframe admission, blockHDAware, photometric check, the gates of
    :644 and the constraints, C++ class against the Python restatement on the same read-back images, the same fern table and the same tracker answers."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc_ferns

    FW, FH, NF, MAXD, SEED, GAP = 160, 120, 200, 3000, 77, 2
    fx, fy, cx, cy, photo = 130.0, 130.0, 80.0, 60.0, 115.0
    w, h = FW // 8, FH // 8
    rng = np.random.default_rng(3)
    scenes = [_fern_scene(rng, w, h, fx / 8, fy / 8, cx / 8, cy / 8, holes) for holes in (0.05, 0.1, 0.3, 0.05)]
    ops = []
    t = 1
    for rep in range(3):
        for si, (img, verts, norms) in enumerate(scenes):
            im = np.clip(img.astype(np.int32) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8) if rep else img
            pose = np.eye(4, dtype=np.float32); pose[:3, 3] = rng.uniform(-0.5, 0.5, 3)
            if rep:   # a lookup first (as inside a frame), then the admission test
                dp = rng.uniform(-0.01, 0.01, 3).astype(np.float32)
                diag = np.zeros(8, np.float32)
                diag[0] = [1e-4, 1e-4, 5e-4, 1e-4][si] if rep == 1 else 1e-4       # lastICPError: one failing the gate
                diag[1] = [3000, 2000, 3000, 3000][si] if rep == 1 else 3000        # lastICPCount: one failing the gate
                ops.append(dict(kind=1, time=t, thr=0.0, pose=pose, img=im, verts=verts, norms=norms, dp=dp, diag=diag))
            ops.append(dict(kind=0, time=t, thr=0.3095 if rep < 2 else 0.0, pose=pose, img=im, verts=verts, norms=norms))
            t += 1
    blank = np.zeros_like(scenes[0][1])
    ops.append(dict(kind=0, time=t, thr=0.3, pose=np.eye(4, dtype=np.float32), img=scenes[0][0], verts=blank, norms=blank))     # no valid sample: never admitted
    ops.append(dict(kind=1, time=t, thr=0.0, pose=np.eye(4, dtype=np.float32), img=scenes[0][0], verts=blank, norms=blank, dp=np.zeros(3, np.float32),
                    diag=np.zeros(8, np.float32)))
    inp, outp = str(tmp_path / "ferns.in"), str(tmp_path / "ferns.out")
    with open(inp, "wb") as f:
        f.write(np.array([FW, FH, NF, MAXD, SEED, GAP, len(ops)], np.int32).tobytes())
        f.write(np.array([fx, fy, cx, cy, photo], np.float32).tobytes())
        for o in ops:
            f.write(np.array([o["kind"], o["time"]], np.int32).tobytes()); f.write(np.float32(o["thr"]).tobytes()); f.write(o["pose"].tobytes())
            f.write(o["img"].tobytes()); f.write(np.zeros_like(o["img"]).tobytes()); f.write(o["verts"].tobytes()); f.write(o["norms"].tobytes())
            if o["kind"] == 1:
                f.write(o["dp"].tobytes()); f.write(o["diag"].tobytes())
    subprocess.run([checker, "ferns", inp, outp], check=True)
    blob = open(outp, "rb").read()
    table = np.frombuffer(blob[:NF * 24], np.int32).reshape(NF, 6)
    assert table[:, 0].min() >= 0 and table[:, 0].max() < w and table[:, 1].max() < h and table[:, 5].min() >= 400 and table[:, 5].max() <= MAXD
    assert len(np.unique(table[:, 0])) > w // 2                     # drawn, not constant
    ref = orc_ferns.Ferns(table, w, h, MAXD, photo, fx / 8, fy / 8, cx / 8, cy / 8, min_gap=GAP)
    off = NF * 24
    n_match = n_cand = n_admit = n_reject = 0
    for o in ops:
        if o["kind"] == 0:
            ok, nf = np.frombuffer(blob[off:off + 8], np.int32); off += 8
            exp = ref.add_frame(o["img"], o["verts"], o["norms"], o["pose"], o["time"], o["thr"])
            assert bool(ok) == exp and nf == len(ref.frames)
            n_admit += exp; n_reject += not exp
        else:
            cand, closest, nc = np.frombuffer(blob[off:off + 12], np.int32); off += 12
            dissim, photo_err = np.frombuffer(blob[off:off + 8], np.float32); off += 8
            est = np.frombuffer(blob[off:off + 64], np.float32).reshape(4, 4); off += 64
            cons = np.frombuffer(blob[off:off + nc * 32], np.float32).reshape(nc, 2, 4); off += nc * 32

            def tracker(mv, mn, cv, cn, p, o=o):
                p[:3, 3] += o["dp"]
                return p, o["diag"][0], o["diag"][1]

            r = ref.find_frame(o["pose"], o["img"], o["verts"], o["norms"], o["time"], False, tracker)
            assert (cand, closest, nc) == (r["candidate"], r["closest"], len(r["constraints"]))
            assert dissim == r["dissim"] and np.array_equal(est, r["est"])
            assert photo_err == r["photo"] or (np.isnan(photo_err) and np.isnan(r["photo"]))
            for a, (src, dst) in zip(cons, r["constraints"]):
                assert np.array_equal(a[0], src) and np.array_equal(a[1], dst)
            n_cand += cand != -1; n_match += closest != -1
    assert off == len(blob)
    assert n_admit >= 4 and n_reject >= 3 and n_cand >= 6 and 2 <= n_match < n_cand      # every branch was taken


@pytest.mark.gpu
def test_fern_database_inside_frames(checker, tmp_path):
    """The C++ data base driven by real frames (640 x 480, fern resolution 80 x 60): the camera goes forth and back, a keyframe of the way out is found on
    the way back (findFrame inside the frame: read-back of the tracked-pose prediction, ifx_track_maps on a handle of fern resolution, photometric check,
    gates), the handler's graph is applied as a fern deformation and the recovery pose is adopted.  The time gap of EF/Ferns.cpp:238 is shortened."""
    from instancefusion_amd import logio, synth

    W, H, K = 640, 480, dict(fx=528.0, fy=528.0, cx=320.0, cy=240.0)
    # no depth noise: the first keyframe is taken before anything is stable, i.e. from the fill-in of the raw frame, whose normals are forward differences
    st = synth.make_stream(7, W, H, K["fx"], K["fy"], K["cx"], K["cy"], noise=False)
    src = [0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0, 1, 2]
    klg = str(tmp_path / "v.klg")
    wr = logio.RawLogWriter(klg, depth="zlib", image="raw")
    for i, s in enumerate(src + [0]):
        wr.add(33333 * i, st["rgb"][s], st["depth"][s])
    wr.close()
    def run(handler, time_delta):
        r = subprocess.run([checker, "fernrun", klg, str(W), str(H), str(K["fx"]), str(K["fy"]), str(K["cx"]), str(K["cy"]), "5", "2", handler, str(time_delta)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        rows = []
        for l in r.stdout.splitlines():
            if l.startswith("tick"):
                t = l.split()
                i = t.index("pos")
                d = dict(zip(t[0:i:2], t[1:i:2]))
                d["pos"] = np.array([float(v) for v in t[i + 1:i + 4]]); d["surfels"] = int(t[i + 5])
                rows.append(d)
        return r, rows

    r, rows = run("rigid", 200)
    assert len(rows) == len(src)
    assert int(rows[0]["keyframes"]) == 1 and int(rows[5]["candidate"]) == -1          # the first frame is always a keyframe; nothing is old enough yet
    hit = [q for q in rows if int(q["closest"]) != -1]
    assert hit, r.stdout
    q = hit[0]
    assert float(q["icpErr"]) < 3e-4 and float(q["icpCount"]) > 2400 and float(q["photo"]) < 115      # the gates of :644
    assert float(q["shift"]) < 0.03                                                                    # estPose * v against currPose * v: the same place
    assert int(rows[-1]["matches"]) >= 1 and int(rows[-1]["deforms"]) >= 1
    # the trajectory stays the trajectory (ground truth: forth and back to the start)
    assert np.abs(rows[12]["pos"] - rows[0]["pos"]).max() < 0.03 and rows[-1]["surfels"] > 100000
    # the reference's own optimiser (ifx_deformation.hpp) in the same place: the keyframe is found again, but the map already agrees with it (mean constraint
    # error < 6 cm), so the global deformation is refused (DeformationGraph.cpp:469) and the frames are those of a run without loop closures
    r2, rows2 = run("builtin", 200)
    assert int(rows2[-1]["matches"]) >= 1 and int(rows2[-1]["deforms"]) == 0 and int(rows2[-1]["local"]) == 0
    # ... and with a short time window (surfels count as old after 3 frames) local loop closures fire and are optimised and applied inside the frames
    r3, rows3 = run("builtin", 3)
    assert int(rows3[-1]["lccand"]) >= 1 and int(rows3[-1]["local"]) >= 1, r3.stdout
    assert float([q for q in rows3 if int(q["local"]) >= 1][0]["consErr"]) < 0.01
    assert np.abs(rows3[12]["pos"] - rows3[0]["pos"]).max() < 0.03 and rows3[-1]["surfels"] > 100000


# ---------------------------------------------------------------- the deformation-graph optimiser (ifx_deformation.hpp) against oracle/orc_deformation.py, no GPU
def _deform_case(seed, n_nodes, fern_match, with_relative, last_deform_time, shift, noise=0.002):
    rng = np.random.default_rng(seed)
    # nodes: a walk through space, one node every few "frames" (times sorted, as the map order guarantees)
    steps = rng.normal(0, 0.12, (n_nodes, 3)).cumsum(axis=0)
    times = np.sort(rng.integers(1, 600, n_nodes))
    xyzt = np.concatenate([steps, times[:, None]], axis=1).astype(np.float32)
    cons = []
    late = np.nonzero(times > times[-1] - 80)[0]
    for _ in range(20):                                        # the recent part of the map is pulled onto where the old model says it should be
        j = rng.choice(late)
        src = xyzt[j, :3] + rng.normal(0, 0.05, 3).astype(np.float32)
        tgt = (src + np.asarray(shift, np.float32) + rng.normal(0, 1.0, 3) * noise).astype(np.float32)
        cons.append(dict(src=src, target=tgt, src_time=int(times[j]), target_time=int(times[rng.integers(0, 5)]), relative=False, pin=fern_match))
    if with_relative:                                          # constraints of an earlier closure: two parts of the map stay together
        for _ in range(6):
            a, b = rng.integers(0, n_nodes // 3), rng.integers(n_nodes // 2, n_nodes)
            p = xyzt[a, :3] + rng.normal(0, 0.03, 3).astype(np.float32)
            cons.append(dict(src=p.astype(np.float32), target=(p + rng.normal(0, 0.004, 3)).astype(np.float32), src_time=int(times[a]), target_time=int(times[b]),
                             relative=True, pin=False))
    poses = []
    for j in rng.choice(n_nodes, 6, replace=False):
        P = np.eye(4, dtype=np.float32)
        a = rng.normal(0, 0.3, 3)
        Rz = np.array([[np.cos(a[0]), -np.sin(a[0]), 0], [np.sin(a[0]), np.cos(a[0]), 0], [0, 0, 1]])
        Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
        P[:3, :3] = (Rz @ Ry).astype(np.float32); P[:3, 3] = xyzt[j, :3] + rng.normal(0, 0.05, 3).astype(np.float32)
        poses.append((int(times[j]), P))
    return xyzt, cons, poses


@pytest.mark.parametrize("case", [dict(seed=1, n=40, fern=False, rel=False, ldt=0, shift=(0.05, -0.03, 0.02)),
                                  dict(seed=2, n=120, fern=False, rel=True, ldt=0, shift=(0.08, 0.02, -0.04)),
                                  dict(seed=3, n=60, fern=False, rel=False, ldt=300, shift=(0.04, 0.04, 0.0)),      # nodes before the last deformation stay fixed
                                  dict(seed=4, n=80, fern=True, rel=False, ldt=300, shift=(0.12, -0.07, 0.04), noise=0.0),   # a fern match re-opens the whole graph
                                  dict(seed=4, n=80, fern=True, rel=True, ldt=300, shift=(0.12, -0.07, 0.04), noise=0.0),    # optimised, then refused by the gates of :166
                                  dict(seed=5, n=30, fern=True, rel=False, ldt=0, shift=(0.01, 0.0, 0.0))])         # mean constraint error < 0.06: refused before optimising
def test_deformation_optimiser_equals_dense_restatement(checker, tmp_path, case):
    """Deformation::constrain (EF/Deformation.cpp:89-220): vertex weights, Gauss-Newton on the embedded deformation graph with the reference's energy,
    stopping rules and acceptance gates, the graph applied to poses and constraint points -- C++ (skyline Cholesky, f32 state as in the reference) against
    a dense f64 numpy restatement."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc_deformation

    xyzt, cons, poses = _deform_case(case["seed"], case["n"], case["fern"], case["rel"], case["ldt"], case["shift"], case.get("noise", 0.002))
    # (constraint order of Deformation::addConstraint: a pinned constraint is followed by its pin target -> target)
    flat = []
    for c in cons:
        flat.append(c)
        if c["pin"] and not c["relative"]:
            flat.append(dict(src=c["target"], target=c["target"], src_time=c["target_time"], target_time=c["target_time"], relative=False, pin=True))
            flat[-2] = dict(c, pin=False)
    inp, outp = str(tmp_path / "d.in"), str(tmp_path / "d.out")
    with open(inp, "wb") as f:
        f.write(np.array([len(xyzt), len(cons), len(poses), int(case["fern"]), 0, case["ldt"], 700], np.int32).tobytes())
        f.write(xyzt.tobytes())
        for c in cons:
            f.write(np.asarray(c["src"], np.float32).tobytes()); f.write(np.asarray(c["target"], np.float32).tobytes())
            f.write(np.array([c["src_time"], c["target_time"], int(c["relative"]), int(c["pin"])], np.int32).tobytes())
        for t, P in poses:
            f.write(np.int32(t).tobytes()); f.write(P.tobytes())
    subprocess.run([checker, "deform", inp, outp], check=True)
    b = open(outp, "rb").read()
    ok = int(np.frombuffer(b[:4], np.int32)[0]); err, mc = np.frombuffer(b[4:12], np.float32); ng = int(np.frombuffer(b[12:16], np.int32)[0])
    off = 16
    raw = np.frombuffer(b[off:off + ng * 4], np.float32).reshape(-1, 16); off += ng * 4
    P_out = np.frombuffer(b[off:off + len(poses) * 64], np.float32).reshape(-1, 4, 4); off += len(poses) * 64
    nr = int(np.frombuffer(b[off:off + 4], np.int32)[0]); off += 4
    rel = np.frombuffer(b[off:off + nr * 24], np.float32).reshape(nr, 2, 3); off += nr * 24
    ldt_out = int(np.frombuffer(b[off:off + 4], np.int32)[0])
    ref = orc_deformation.constrain(xyzt, flat, [P.copy() for _, P in poses], [t for t, _ in poses], case["fern"], False, case["ldt"])
    assert bool(ok) == ref["ok"]
    assert abs(mc - ref["mean_cons"]) < 2e-5
    assert abs(err - ref["error"]) < 1e-4 * max(1.0, ref["error"])
    assert ref["ok"] == (not case["fern"] or (not case["rel"] and case["seed"] == 4))              # the case exercises the branch it was built for
    if not ref["ok"]:
        assert ng == 0 and nr == 0 and np.array_equal(P_out, np.array([P for _, P in poses]))      # refused: nothing moves
        assert (ref["error"] > 0) == (case["seed"] == 4)
        return
    assert raw.shape == ref["raw_graph"].shape and np.abs(raw - ref["raw_graph"]).max() < 2e-4
    assert np.abs(P_out - np.array(ref["poses"])).max() < 2e-4
    for P in P_out:
        assert np.abs(P[:3, :3] @ P[:3, :3].T - np.eye(3)).max() < 1e-5                        # re-orthonormalised
    assert np.abs(raw[:, 12:15]).max() > 0.02                                                   # the graph did move
    if case["ldt"] and not case["fern"]:
        old = xyzt[:, 3] <= case["ldt"]
        assert old.any() and np.array_equal(raw[old, 3:12], np.tile(np.eye(3, dtype=np.float32).reshape(9), (old.sum(), 1))) and not raw[old, 12:15].any()
    if not case["fern"]:
        assert nr == len(ref["new_rel"]) == 20 and ldt_out == 700
        for a, (s_, t_, _, _) in zip(rel, ref["new_rel"]):
            assert np.abs(a[0] - s_).max() < 2e-4 and np.abs(a[1] - t_).max() < 1e-6
        moved_err = np.linalg.norm(rel[:, 0] - rel[:, 1], axis=1).mean()
        assert moved_err < 0.01                                                                  # the constraint points arrived (they started ~6-9 cm away)
    else:
        assert nr == 0 and ldt_out == case["ldt"]
