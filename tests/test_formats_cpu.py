"""Caller-side formats (unimatch_amd/io.py) against bytes written by the reference's own writers (tests/golden/
formats.npz) and by round trips; KITTI PNGs by their published encoding (cv2, which the reference uses for them, is
not available here)."""
import os

import numpy as np
import torch

from unimatch_amd import io as uio


def test_flo_bytes_and_roundtrip(golden, tmp_path):
    g = golden('formats')
    flow = g['flow'].numpy()
    path = os.path.join(tmp_path, 'a.flo')
    uio.write_flo(path, flow)
    raw = open(path, 'rb').read()
    assert raw[:4] == b'PIEH'                                   # 202021.25 as little-endian fp32
    assert np.array_equal(np.frombuffer(raw, np.uint8), g['flo_bytes'].numpy())
    assert np.array_equal(uio.read_flo(path), flow)


def test_pfm_bytes_and_roundtrip(golden, tmp_path):
    g = golden('formats')
    disp = g['disp'].numpy()
    path = os.path.join(tmp_path, 'a.pfm')
    uio.write_pfm(path, disp)
    assert np.array_equal(np.frombuffer(open(path, 'rb').read(), np.uint8), g['pfm_bytes'].numpy())
    back, scale = uio.read_pfm(path)
    assert np.array_equal(back, disp) and scale == 1.0
    rgb = np.random.RandomState(0).rand(4, 5, 3).astype(np.float32)
    uio.write_pfm(path, rgb)
    assert np.array_equal(uio.read_pfm(path)[0], rgb)


def test_input_padder_matches_reference(golden):
    g = golden('formats')
    for i, mode in enumerate(('sintel', 'kitti', 'sintel')):
        hh, ww, pf, ph, pw, *pad = (int(v) for v in g[f'pad{i}.meta'])
        gen = torch.Generator().manual_seed(510 + i)
        x = torch.randn(1, 3, hh, ww, generator=gen)
        p = uio.InputPadder(x.shape, mode=mode, padding_factor=pf)
        y = p.pad(x)[0]
        assert list(p._pad) == pad and tuple(y.shape[-2:]) == (ph, pw)
        assert ph % pf == 0 and pw % pf == 0
        assert abs(y.double().sum().item() - g[f'pad{i}.sum'].item()) < 1e-6 * max(1.0, abs(g[f'pad{i}.sum'].item()))
        assert torch.equal(p.unpad(y), x)


def test_kitti_png_encodings(tmp_path):
    rs = np.random.RandomState(1)
    flow = (rs.rand(9, 11, 2).astype(np.float32) - 0.5) * 300
    path = os.path.join(tmp_path, 'f.png')
    uio.write_kitti_flow(path, flow)
    rgb = uio.read_png16(path)
    assert rgb.dtype == np.uint16 and rgb.shape == (9, 11, 3) and (rgb[:, :, 2] == 1).all()
    assert np.array_equal(rgb[:, :, :2], (64.0 * flow.astype(np.float64) + 2 ** 15).astype(np.uint16))
    back, valid = uio.read_kitti_flow(path)
    assert np.abs(back - flow).max() <= 1 / 64 and (valid == 1).all()
    disp = rs.rand(6, 7).astype(np.float32) * 200
    uio.write_kitti_disp(path, disp)
    # the reference writer truncates (evaluate_stereo.py:91): exactly those uint16 values must be on disk
    assert np.array_equal(uio.read_png16(path), (disp * 256.).astype(np.uint16))
    back, valid = uio.read_kitti_disp(path)
    assert (back <= disp + 1e-6).all() and np.abs(back - disp).max() < 1 / 256 + 1e-6 and valid.all()


def test_png_reader_handles_all_filter_types(tmp_path):
    """Files written by other encoders use sub/up/average/paeth filters: re-encode a file with each and read back."""
    import struct
    import zlib
    rs = np.random.RandomState(2)
    img = rs.randint(0, 65536, size=(5, 6, 3)).astype(np.uint16)
    h, w = img.shape[:2]
    rows = img.astype('>u2').reshape(h, -1).view(np.uint8).astype(np.int32)
    bpp = 6
    for ft in (1, 2, 3, 4):
        lines = []
        prev = np.zeros(rows.shape[1], np.int32)
        for y in range(h):
            cur = rows[y]
            enc = np.zeros_like(cur)
            for i in range(cur.size):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 2:
                    pred = b
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                enc[i] = (cur[i] - pred) & 255
            lines.append(bytes([ft]) + enc.astype(np.uint8).tobytes())
            prev = cur
        path = os.path.join(tmp_path, f'f{ft}.png')
        with open(path, 'wb') as f:
            f.write(b'\x89PNG\r\n\x1a\n')
            f.write(uio._png_chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 16, 2, 0, 0, 0)))
            f.write(uio._png_chunk(b'IDAT', zlib.compress(b''.join(lines))))
            f.write(uio._png_chunk(b'IEND', b''))
        assert np.array_equal(uio.read_png16(path), img), ft
