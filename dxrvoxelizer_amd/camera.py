"""Camera helpers for the display pass: the matrices the reference app hands to
Voxelizer::UpdateFrame (DXRVoxelizer.cpp:225-234, :249-254), DirectXMath conventions
(row-major storage, row vectors: v' = v @ M, left-handed)."""
import struct
import zlib

import numpy as np

FOV_Y = np.pi / 4        # g_FOVAngleY, DXRVoxelizer.cpp:21
Z_NEAR, Z_FAR = 1.0, 1000.0
DEFAULT_EYE = (8.0, 12.0, -14.0)      # DXRVoxelizer.cpp:230
DEFAULT_FOCUS = (0.0, 4.0, 0.0)       # :229


def look_at_lh(eye, focus, up=(0.0, 1.0, 0.0)):
    eye, focus, up = (np.asarray(v, np.float64) for v in (eye, focus, up))
    z = focus - eye
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2] = x, y, z
    m[3, :3] = [-x @ eye, -y @ eye, -z @ eye]
    return m.astype(np.float32)


def perspective_fov_lh(fov_y, aspect, zn, zf):
    h = 1.0 / np.tan(fov_y / 2)
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1] = h / aspect, h
    m[2, 2], m[2, 3] = zf / (zf - zn), 1.0
    m[3, 2] = -zn * zf / (zf - zn)
    return m.astype(np.float32)


def default_view_proj(width=1280, height=720, eye=DEFAULT_EYE, focus=DEFAULT_FOCUS):
    """(eye, view @ proj) of the app's start-up camera (Main.cpp:17: 1280 x 720)."""
    vp = look_at_lh(eye, focus) @ perspective_fov_lh(FOV_Y, width / height, Z_NEAR, Z_FAR)
    return np.asarray(eye, np.float32), vp.astype(np.float32)


def write_png(path, rgba):
    """Minimal PNG writer (the app saves screenshots with stb_image_write, DXRVoxelizer.cpp:531-551)."""
    img = np.ascontiguousarray(rgba, np.uint8)
    h, w, c = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
