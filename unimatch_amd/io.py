"""Caller-side data formats of the reference's evaluation scripts ("next" row 4):
input padding and the on-disk result formats, so that predictions of the drop-in module can be written / read in
exactly the bytes the reference tools produce.  Pure host code (numpy); nothing here touches the GPU path.

  InputPadder          utils/utils.py:6-24        replicate padding to a multiple of `padding_factor`
  write_flo/read_flo   utils/frame_utils.py:10-99  Middlebury .flo  (magic 202021.25 'PIEH', int32 w, h, interleaved fp32 u v)
  write_pfm/read_pfm   utils/file_io.py:60-127     PFM (rows bottom-to-top, negative scale = little endian)
  write/read_kitti_*   utils/frame_utils.py:102-121  KITTI 16-bit PNG: flow (u, v, valid) = 64 * uv + 2^15; disparity * 256
"""
import re
import struct
import zlib

import numpy as np
import torch.nn.functional as F

FLO_MAGIC = np.array([202021.25], np.float32)


class InputPadder:
    """Replicate-pad ``[..., H, W]`` images so that H and W become multiples of ``padding_factor`` and crop predictions back.

    Same interface and the same pad amounts as the reference's helper (utils/utils.py:6-24): the missing rows / columns are
    ``(-size) mod factor``; ``mode='sintel'`` splits the vertical padding between top and bottom (top gets the smaller half),
    every other mode (the reference passes 'kitti') puts all of it at the bottom; horizontal padding is always split."""

    def __init__(self, dims, mode='sintel', padding_factor=8):
        height, width = int(dims[-2]), int(dims[-1])
        extra_h, extra_w = (-height) % padding_factor, (-width) % padding_factor
        left = extra_w // 2
        top = extra_h // 2 if mode == 'sintel' else 0
        # F.pad order: (left, right, top, bottom)
        self._pad = [left, extra_w - left, top, extra_h - top]
        self.ht, self.wd = height, width

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode='replicate') for x in inputs]

    def unpad(self, x):
        left, right, top, bottom = self._pad
        return x[..., top:x.shape[-2] - bottom, left:x.shape[-1] - right]


# ------------------------------------------------------------------ Middlebury .flo
def write_flo(filename, flow):
    """flow: [H, W, 2] (u, v)."""
    flow = np.asarray(flow)
    assert flow.ndim == 3 and flow.shape[2] == 2
    h, w = flow.shape[:2]
    with open(filename, 'wb') as f:
        f.write(FLO_MAGIC.tobytes())
        f.write(struct.pack('<ii', w, h))
        f.write(np.ascontiguousarray(flow, dtype='<f4').tobytes())


def read_flo(filename):
    with open(filename, 'rb') as f:
        magic = np.frombuffer(f.read(4), np.float32)[0]
        if magic != 202021.25:
            raise ValueError('Magic number incorrect. Invalid .flo file')
        w, h = struct.unpack('<ii', f.read(8))
        data = np.frombuffer(f.read(8 * w * h), '<f4')
    return data.reshape(h, w, 2).copy()


# ------------------------------------------------------------------ PFM
def write_pfm(filename, image, scale=1):
    image = np.asarray(image)
    if image.dtype.name != 'float32':
        raise ValueError('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise ValueError('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    flipped = np.flipud(image)
    with open(filename, 'wb') as f:
        f.write(b'PF\n' if color else b'Pf\n')
        f.write(b'%d %d\n' % (flipped.shape[1], flipped.shape[0]))
        f.write(b'%f\n' % -scale)                      # negative: little endian payload
        f.write(np.ascontiguousarray(flipped, dtype='<f4').tobytes())


def read_pfm(filename):
    with open(filename, 'rb') as f:
        header = f.readline().rstrip()
        if header == b'PF':
            color = True
        elif header == b'Pf':
            color = False
        else:
            raise ValueError('Not a PFM file.')
        m = re.match(rb'^(\d+)\s(\d+)\s$', f.readline())
        if not m:
            raise ValueError('Malformed PFM header.')
        width, height = map(int, m.groups())
        scale = float(f.readline().rstrip())
        endian = '<' if scale < 0 else '>'
        data = np.frombuffer(f.read(), endian + 'f4')
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


# ------------------------------------------------------------------ minimal 16-bit PNG codec (grey or RGB)
def _png_chunk(tag, data):
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)


def write_png16(filename, arr):
    """arr: uint16 [H, W] (grey) or [H, W, 3] (RGB)."""
    arr = np.asarray(arr)
    assert arr.dtype == np.uint16 and (arr.ndim == 2 or (arr.ndim == 3 and arr.shape[2] == 3))
    h, w = arr.shape[:2]
    color_type = 0 if arr.ndim == 2 else 2
    rows = arr.astype('>u2').reshape(h, -1).view(np.uint8)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], 1).tobytes()     # filter type 0 on every scanline
    with open(filename, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n')
        f.write(_png_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 16, color_type, 0, 0, 0)))
        f.write(_png_chunk(b'IDAT', zlib.compress(raw, 6)))
        f.write(_png_chunk(b'IEND', b''))


def read_png16(filename):
    with open(filename, 'rb') as f:
        blob = f.read()
    if blob[:8] != b'\x89PNG\r\n\x1a\n':
        raise ValueError('not a PNG file')
    pos, idat, ihdr = 8, b'', None
    while pos < len(blob):
        n, tag = struct.unpack('>I4s', blob[pos:pos + 8])
        body = blob[pos + 8:pos + 8 + n]
        if tag == b'IHDR':
            ihdr = struct.unpack('>IIBBBBB', body)
        elif tag == b'IDAT':
            idat += body
        pos += 12 + n
    w, h, depth, color_type, _, _, interlace = ihdr
    if depth != 16 or color_type not in (0, 2) or interlace:
        raise ValueError('only non-interlaced 16-bit grey / RGB PNGs are supported')
    ch = 1 if color_type == 0 else 3
    bpp, stride = 2 * ch, 2 * ch * w
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                                          # sub / average / paeth depend on the pixel to the left
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur.astype(np.uint8)
        prev = cur
    arr = out.view('>u2').astype(np.uint16)
    return arr.reshape(h, w) if ch == 1 else arr.reshape(h, w, 3)


# ------------------------------------------------------------------ KITTI encodings
def write_kitti_flow(filename, uv):
    """uv: [H, W, 2].  Stored RGB = (64 u + 2^15, 64 v + 2^15, valid = 1)."""
    uv = 64.0 * np.asarray(uv, np.float64) + 2 ** 15
    valid = np.ones(uv.shape[:2] + (1,))
    write_png16(filename, np.concatenate([uv, valid], -1).astype(np.uint16))


def read_kitti_flow(filename):
    rgb = read_png16(filename).astype(np.float32)
    return (rgb[:, :, :2] - 2 ** 15) / 64.0, rgb[:, :, 2]


def write_kitti_disp(filename, disp):
    """disparity * 256 TRUNCATED to uint16, as the reference writer does (evaluate_stereo.py:91: ``(disp * 256.).astype(np.uint16)``)."""
    write_png16(filename, (np.asarray(disp) * 256.).astype(np.uint16))


def read_kitti_disp(filename):
    disp = read_png16(filename).astype(np.float32) / 256.0
    return disp, disp > 0.0
