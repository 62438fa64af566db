"""SURVEY 8(f) #3/#4 extras against vectors made by running the reference (oracle/gen_golden_extras.py): KITTI flow
PNGs, removeSmallSegments, and the host half of the packedksets / pakovanjeZaC writers (the GPU half: test_gpu_parity)."""
import hashlib
import struct
import zlib

import numpy as np
import pytest

from conftest import pkg


def digest(a, dtype):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=dtype).tobytes()).hexdigest()


def test_kitti_png_reader_matches_reference(golden, tmp_path):
    g, fio = golden("extras"), pkg("flowio")
    path = str(tmp_path / "gt.png")
    fio.write_kitti_flow_png(path, g["png_uvv"])
    assert np.array_equal(np.frombuffer(open(path, "rb").read(), np.uint8), g["png_bytes"])
    got = fio.read_kitti_flow_png(path)
    assert got.dtype == np.float32 and np.array_equal(got, g["png_field_by_reference"])   # what visualization.py:37-53 decoded
    assert np.array_equal(got, g["png_uvv"])                                             # codes are exact multiples of 1/64


@pytest.mark.parametrize("depth", (8, 16))
def test_png_decoder_handles_every_filter_type(tmp_path, depth):
    """PNG files written by libpng use adaptive filters: encode each line with filter (y mod 5) and decode it back."""
    fio = pkg("flowio")
    rng = np.random.default_rng(3)
    h, w, bpc = 11, 7, depth // 8
    img = rng.integers(0, 1 << depth, size=(h, w, 3)).astype(np.uint16)
    raw = img.astype(">u2").tobytes() if depth == 16 else img.astype(np.uint8).tobytes()
    bpp, stride = 3 * bpc, w * 3 * bpc
    lines, prev = [], bytes(stride)
    for y in range(h):
        cur, ft, enc = raw[y * stride:(y + 1) * stride], y % 5, bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b, c = prev[i], (prev[i - bpp] if i >= bpp else 0)
            paeth = min((abs(b - c), 0, a), (abs(a - c), 1, b), (abs(a + b - 2 * c), 2, c))[2]
            enc[i] = (cur[i] - (0, a, b, (a + b) >> 1, paeth)[ft]) & 255
        lines.append(bytes([ft]) + bytes(enc)); prev = cur
    ch = lambda t, body: struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    comp = zlib.compress(b"".join(lines))
    path = str(tmp_path / "f.png")
    with open(path, "wb") as f:                      # two IDAT chunks: the decoder must concatenate them
        f.write(b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, 2, 0, 0, 0))
                + ch(b"IDAT", comp[:5]) + ch(b"IDAT", comp[5:]) + ch(b"IEND", b""))
    assert np.array_equal(fio.read_png16(path), img)
    with open(path, "wb") as f:
        f.write(b"not a png")
    with pytest.raises(ValueError):
        fio.read_png16(path)


@pytest.mark.parametrize("case", (0, 1, 2))
def test_remove_small_segments_matches_reference(golden, case):
    g, compat = golden("extras"), pkg("compat")
    f = np.ascontiguousarray(g["seg%d_in" % case])
    tresh, mins = (int(v) for v in g["seg%d_par" % case])
    compat.remove_small_segments(f, tresh, mins)
    assert np.array_equal(f, g["seg%d_out" % case])
    with pytest.raises(ValueError):
        compat.remove_small_segments(f.astype(np.float64), tresh, mins)


def _fixture_a_proposals(O):
    """State after nasumicni of golden fixture a, forward pass (what gen_golden_extras ran the reference on)."""
    synth = pkg("synth")
    H, W, ch, cw, seed = 40, 48, 5, 6, 11
    p = O.make_params(H, W, ch, cw, seed=seed)
    img1, img2, _ = synth.make_pair(H, W, seed=seed, amp_x=0.12 * W, amp_y=0.12 * H)
    d1, d2 = O.daisy(img1), O.daisy(img2)
    pr, lc, npr, bl = O.knn_proposals(p, d1, d2)
    O.neighbour_proposals(p, d1, d2, pr, lc, npr, bl)
    return p, pr, npr


def test_pakovani_za_c_and_border_replay_match_reference(oracle, golden):
    """Host half of compat.py: from clean per-pixel matrices (here: numpy) to the reference's packedksets (border scratch
    replay, checked against the reference-pinned oracle) and to the four pakovanjeZaC files (checked against the reference)."""
    g, compat = golden("extras"), pkg("compat")
    p, pr, npr = _fixture_a_proposals(oracle)
    assert np.array_equal(npr, g["za_c_nprop"])
    H, W, L = p.pich, p.picw, p.maxnprop
    clean = np.zeros((H, W, 2, L * L // 8 + 1), np.uint8)
    for y in range(H):
        for x in range(W):
            for slot, (ny, nx) in enumerate(((y + 1, x), (y, x + 1))):
                if ny >= H or nx >= W:
                    continue
                a, b = pr[y, x, :npr[y, x]], pr[ny, nx, :npr[ny, nx]]
                m = np.zeros((L, L), bool)
                m[:len(a), :len(b)] = p.tpsi > np.abs(a[:, None, :] - b[None, :, :]).sum(-1)
                clean[y, x, slot] = np.packbits(m.reshape(-1))
    compat.replay_border_scratch(clean, npr, L)
    assert np.array_equal(clean, oracle.pack_compat(p, pr, npr))
    for k, arr in enumerate(compat.pakovani_za_c(clean)):
        assert tuple(arr.shape) == tuple(g["za_c%d_shape" % k])
        assert digest(arr, np.uint8) == str(g["za_c%d_sha" % k]), "pakovani za c %d" % k


def test_ucitajflow_dispatch(golden, tmp_path):
    """evaluate.ucitajFlow = FlowImage.ucitajFlow (visualization.py:97-124) for the three file kinds."""
    g, ev, fio = golden("extras"), pkg("evaluate"), pkg("flowio")
    png = str(tmp_path / "gt.png")
    open(png, "wb").write(g["png_bytes"].tobytes())
    assert np.array_equal(ev.ucitajFlow(png), g["png_field_by_reference"])
    dydx = np.arange(2 * 3 * 2, dtype=np.float64).reshape(2, 3, 2)
    np.save(str(tmp_path / "f.npy"), dydx)
    a = ev.ucitajFlow(str(tmp_path / "f.npy"))
    assert np.array_equal(a[..., 0], dydx[..., 1]) and np.array_equal(a[..., 1], dydx[..., 0]) and a[..., 2].all()
    fio.write_flo(str(tmp_path / "f.flo"), dydx)
    assert np.array_equal(ev.ucitajFlow(str(tmp_path / "f.flo")), a)
    with pytest.raises(ValueError):
        ev.ucitajFlow(str(tmp_path / "f.txt"))
