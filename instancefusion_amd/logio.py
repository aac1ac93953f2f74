"""Log input and result output of the reference: host-side plumbing around the C-ABI.

* `RawLogReader`  <- IF/utilities/RawLogReader.cpp:19-143 (`.klg`: int32 frame count; per frame int64 timestamp,
  int32 depthSize, int32 imageSize, depth raw or zlib, RGB raw or JPEG; `flipColors` swaps channels 0 and 2)
* `PNGLogReader`  <- IF/utilities/PNGLogReader.cpp:28-212 (`data.txt`: `timestamp depth_path rgb_path depth_id rgb_id`
  per line, paths relative to the file; 16-bit depth PNGs, 8-bit colour images delivered as RGB)
* `RawLogWriter`  -- the inverse of RawLogReader (the reference records with a separate Logger tool; used for tests,
  for the plumbing configuration and to turn synthetic streams into `.klg` files)
* `save_freiburg` <- EF/ElasticFusion.cpp:99-128 (`timestamp tx ty tz qx qy qz qw`, timestamp in seconds = us / 1e6)
* `save_ply`      <- EF/ElasticFusion.cpp:796-894 (binary little-endian, stable surfels only, normals negated) and the
  `_Instance.ply` variant of :896-990 with the instance colour instead of the photometric one

The method names of the readers follow `LogReader` (IF/utilities/LogReader.h): getNext, getBack, hasMore, rewind,
fastForward, getNumFrames; the current frame is in `.rgb`, `.depth`, `.timestamp`.
"""
from __future__ import annotations

import io
import os
import struct
import zlib

import numpy as np


class RawLogReader:
    def __init__(self, file: str, width: int, height: int, flipColors: bool = False):
        assert os.path.exists(file), file
        self.file, self.w, self.h, self.flipColors = file, width, height, flipColors
        self.numPixels = width * height
        self.fp = open(file, "rb")
        (self.numFrames,) = struct.unpack("<i", self.fp.read(4))
        self.currentFrame = 0
        self.filePointers: list[int] = []
        self.timestamp = 0
        self.depth = np.zeros((height, width), np.uint16)
        self.rgb = np.zeros((height, width, 3), np.uint8)

    def close(self):
        self.fp.close()

    def getNumFrames(self) -> int:
        return self.numFrames

    def hasMore(self) -> bool:
        return self.currentFrame + 1 < self.numFrames     # RawLogReader.cpp:134-137 (the last frame is never delivered)

    def getNext(self):
        self.filePointers.append(self.fp.tell())
        self._getCore()

    def getBack(self):
        assert self.filePointers
        self.fp.seek(self.filePointers.pop())
        self._getCore()

    def rewind(self):
        self.fp.seek(4)
        self.filePointers.clear()
        self.currentFrame = 0

    def fastForward(self, frame: int):
        while self.currentFrame < frame and self.hasMore():
            self.filePointers.append(self.fp.tell())
            ts, depth_size, image_size = struct.unpack("<qii", self.fp.read(16))
            self.fp.seek(depth_size + image_size, os.SEEK_CUR)
            self.currentFrame += 1

    def _getCore(self):
        self.timestamp, depth_size, image_size = struct.unpack("<qii", self.fp.read(16))
        dbuf = self.fp.read(depth_size)
        ibuf = self.fp.read(image_size) if image_size > 0 else b""
        if depth_size != self.numPixels * 2:
            dbuf = zlib.decompress(dbuf)
        self.depth = np.frombuffer(dbuf, np.uint16, self.numPixels).reshape(self.h, self.w).copy()
        if image_size == self.numPixels * 3:
            rgb = np.frombuffer(ibuf, np.uint8).reshape(self.h, self.w, 3).copy()
        elif image_size > 0:
            from PIL import Image

            rgb = np.asarray(Image.open(io.BytesIO(ibuf)).convert("RGB")).copy()
        else:
            rgb = np.zeros((self.h, self.w, 3), np.uint8)
        if self.flipColors:
            rgb = rgb[:, :, ::-1].copy()
        self.rgb = rgb
        self.currentFrame += 1


class RawLogWriter:
    """Writes the `.klg` layout RawLogReader expects.  depth: 'raw' | 'zlib'; image: 'raw' | 'jpeg' | 'none'."""

    def __init__(self, file: str, depth: str = "zlib", image: str = "raw", jpeg_quality: int = 90):
        self.fp = open(file, "wb")
        self.fp.write(struct.pack("<i", 0))
        self.n, self.depth_mode, self.image_mode, self.q = 0, depth, image, jpeg_quality

    def add(self, timestamp: int, rgb: np.ndarray, depth: np.ndarray):
        d = np.ascontiguousarray(depth, np.uint16).tobytes()
        if self.depth_mode == "zlib":
            d = zlib.compress(d)
        if self.image_mode == "raw":
            im = np.ascontiguousarray(rgb, np.uint8).tobytes()
        elif self.image_mode == "jpeg":
            from PIL import Image

            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(rgb, np.uint8)).save(buf, format="JPEG", quality=self.q)
            im = buf.getvalue()
        else:
            im = b""
        self.fp.write(struct.pack("<qii", int(timestamp), len(d), len(im)))
        self.fp.write(d)
        self.fp.write(im)
        self.n += 1

    def close(self):
        self.fp.seek(0)
        self.fp.write(struct.pack("<i", self.n))
        self.fp.close()


class PNGLogReader:
    def __init__(self, file: str, width: int, height: int):
        self.w, self.h = width, height
        base = os.path.dirname(os.path.abspath(file))
        self.frames = []
        with open(file) as f:
            for line in f:
                t = line.split()
                if len(t) < 5:
                    continue
                self.frames.append(dict(timestamp=int(float(t[0])), depth_path=os.path.join(base, t[1]), rgb_path=os.path.join(base, t[2]), depth_id=t[3], rgb_id=t[4]))
        self.lastGot = -1
        self.timestamp = 0
        self.depth = np.zeros((height, width), np.uint16)
        self.rgb = np.zeros((height, width, 3), np.uint8)
        self.has_depth_filled = False
        self.depthfilled = None

    def getNumFrames(self) -> int:
        return len(self.frames)

    def hasMore(self) -> bool:
        return self.lastGot + 1 < len(self.frames)

    def rewind(self):
        self.lastGot = -1

    def fastForward(self, frame: int):
        self.lastGot = min(frame, len(self.frames)) - 1

    def getBack(self):
        self.lastGot = max(self.lastGot - 2, -1)
        self.getNext()

    def getNext(self):
        from PIL import Image

        if not self.hasMore():
            return
        self.lastGot += 1
        info = self.frames[self.lastGot]
        self.timestamp = info["timestamp"]
        self.rgb = np.asarray(Image.open(info["rgb_path"]).convert("RGB")).copy()          # imread BGR + flipColors == RGB
        self.depth = np.asarray(Image.open(info["depth_path"])).astype(np.uint16).copy()
        filled = info["depth_path"][:-9] + "depthfilled.png"                                # PNGLogReader.cpp:160-163
        self.has_depth_filled = os.path.exists(filled)
        self.depthfilled = np.asarray(Image.open(filled)).astype(np.uint16).copy() if self.has_depth_filled else None


def _quaternion(R: np.ndarray):
    """Eigen::Quaternionf(Matrix3f) (Shepperd's method as in Eigen/src/Geometry/Quaternion.h), returns x, y, z, w."""
    R = np.asarray(R, np.float32)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        s = np.sqrt(np.float32(t + 1.0), dtype=np.float32)
        w = np.float32(0.5) * s
        s = np.float32(0.5) / s
        return (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(np.float32(R[i, i] - R[j, j] - R[k, k] + 1.0), dtype=np.float32)
    q = [0.0, 0.0, 0.0]
    q[i] = np.float32(0.5) * s
    s = np.float32(0.5) / s
    w = (R[k, j] - R[j, k]) * s
    q[j] = (R[j, i] + R[i, j]) * s
    q[k] = (R[k, i] + R[i, k]) * s
    return q[0], q[1], q[2], w


def save_freiburg(path: str, timestamps_us, poses, iclnuim: bool = False):
    """EF/ElasticFusion.cpp:104-128: one line per pose, `%.6f tx ty tz qx qy qz qw` (stream default float formatting)."""
    with open(path, "w") as f:
        for ts, P in zip(timestamps_us, poses):
            P = np.asarray(P, np.float32)
            t = float(ts) if iclnuim else float(ts) / 1000000.0
            x, y, z, w = _quaternion(P[:3, :3])
            f.write(f"{t:.6f} " + " ".join(f"{float(v):g}" for v in (P[0, 3], P[1, 3], P[2, 3])) + " " + " ".join(f"{float(v):g}" for v in (x, y, z, w)) + "\n")


def _decode_color(c: np.ndarray):
    ci = c.astype(np.int64)
    return ((ci >> 16) & 0xFF).astype(np.uint8), ((ci >> 8) & 0xFF).astype(np.uint8), (ci & 0xFF).astype(np.uint8)


def save_ply(path: str, map_dict: dict, confidence: float = 10.0, instance: bool = False):
    """EF/ElasticFusion.cpp:796-894 (and :896-990 with instance=True): stable surfels (confidence > threshold) as
    x y z | r g b | nx ny nz | radius, normals negated, binary little-endian.  `map_dict` is ElasticFusion.download()."""
    pc, nr, col = map_dict["pc"], map_dict["nr"], map_dict["col"]
    keep = pc[:, 3] > confidence
    n = int(keep.sum())
    r, g, b = _decode_color(col[keep, 1 if instance else 0])
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("radius", "<f4")])
    rec["x"], rec["y"], rec["z"] = pc[keep, 0], pc[keep, 1], pc[keep, 2]
    rec["r"], rec["g"], rec["b"] = r, g, b
    rec["nx"], rec["ny"], rec["nz"] = -nr[keep, 0], -nr[keep, 1], -nr[keep, 2]
    rec["radius"] = nr[keep, 3]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny\nproperty float nz\n"
              "property float radius\nend_header\n") % n
    with open(path, "wb") as f:
        f.write(header.encode())
        f.write(rec.tobytes())
    return n
