"""8f-f2: the library's PNG decoder (csrc/pngdec.hip: zlib inflate + un-filter, host code) against Pillow, byte for byte - on the
reference's own 11 `data/tiny` payloads (tests/golden/tiny_track.npz), on every PNG filter type (files written here with a chosen
filter per scanline), on the formats it must refuse, and through the thread pool (order, slot lifetime, errors).  No GPU."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

from radarslampy_amd import _ffi, parseData

HERE = os.path.dirname(os.path.abspath(__file__))


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def write_png(img, filters, idat_split=0, depth=8, ctype=0, interlace=0, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """8-bit greyscale PNG of `img` with filter type filters[y % len] on scanline y (None 0, Sub 1, Up 2, Average 3, Paeth 4)"""
    h, w = img.shape
    a = img.astype(np.int32)
    raw = bytearray()
    for y in range(h):
        ft = filters[y % len(filters)]
        cur = a[y]
        left = np.concatenate(([0], cur[:-1]))
        up = a[y - 1] if y else np.zeros(w, np.int32)
        ul = np.concatenate(([0], up[:-1]))
        if ft == 0:
            pred = np.zeros(w, np.int32)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = up
        elif ft == 3:
            pred = (left + up) >> 1
        else:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
    z = co.compress(bytes(raw)) + co.flush()
    parts = [z] if idat_split <= 0 else [z[i:i + idat_split] for i in range(0, len(z), idat_split)]
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    out += _chunk(b"tEXt", b"Comment\x00made by the test")          # an ancillary chunk before the data
    for p in parts:
        out += _chunk(b"IDAT", p)
    return out + _chunk(b"IEND", b"")


def decode(png, shape=None, stride=0):
    import ctypes as C
    lib = _ffi.load_library()
    rows, cols = C.c_int32(0), C.c_int32(0)
    buf = np.frombuffer(png, np.uint8)
    rc = lib.roam_png_decode_gray8(buf.ctypes.data_as(C.c_void_p), len(png), None, 0, 0, C.byref(rows), C.byref(cols))
    if rc != _ffi.ROAM_E_CAPACITY:
        return rc, None
    h, w = rows.value, cols.value
    st = stride or w
    out = np.full((h, st), 0xA5, np.uint8)
    rc = lib.roam_png_decode_gray8(buf.ctypes.data_as(C.c_void_p), len(png), out.ctypes.data_as(C.c_void_p), out.size, st, None, None)
    return rc, out


def pillow(png):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(png)))


@pytest.fixture(scope="module")
def tiny_payloads():
    return np.load(os.path.join(HERE, "golden", "tiny_track.npz"))["payload"]


def test_the_reference_tiny_payloads_from_png_bytes(tiny_payloads):
    """the 11 real scans of the reference's data/tiny, written by Pillow (its encoder picks the filters), decoded by the library"""
    from PIL import Image
    for k, p in enumerate(tiny_payloads):
        for level in (1, 6):
            bio = io.BytesIO()
            Image.fromarray(p).save(bio, format="PNG", compress_level=level)
            png = bio.getvalue()
            rc, got = decode(png)
            assert rc == _ffi.ROAM_OK
            assert np.array_equal(got, p) and np.array_equal(got, pillow(png)), f"scan {k} level {level}"


@pytest.mark.parametrize("filters", [(0,), (1,), (2,), (3,), (4,), (4, 4, 4, 1, 4, 4, 2, 3, 0, 4, 4, 4, 4, 4), (3, 4, 2, 1, 0)])
def test_every_filter_type_and_idat_split(tiny_payloads, filters):
    img = np.ascontiguousarray(tiny_payloads[3][:97, :531])          # odd sizes; rows of Paeth runs of every length 1..5
    for split in (0, 4096, 7):
        png = write_png(img, filters, idat_split=split)
        assert np.array_equal(pillow(png), img)                      # the writer is right
        rc, got = decode(png)
        assert rc == _ffi.ROAM_OK and np.array_equal(got, img), (filters, split)
    rc, got = decode(write_png(img, filters), stride=600)            # rows written `stride` bytes apart, the rest untouched
    assert rc == _ffi.ROAM_OK and np.array_equal(got[:, :531], img) and (got[:, 531:] == 0xA5).all()


def test_random_images_all_filters():
    rng = np.random.default_rng(5)
    for trial in range(6):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 300))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        filt = tuple(int(v) for v in rng.integers(0, 5, 9))
        png = write_png(img, filt, idat_split=int(rng.integers(0, 3)) * 33)
        rc, got = decode(png)
        assert rc == _ffi.ROAM_OK and np.array_equal(got, img), (h, w, filt)


def test_every_deflate_block_shape(tiny_payloads):
    """the library's own inflate (csrc/fastinflate.h) on stored blocks, the fixed code, dynamic codes of every strategy, long matches at
    distance 1, short distances 2..7, distances up to the window - always against Pillow (zlib)"""
    rng = np.random.default_rng(11)
    real = np.ascontiguousarray(tiny_payloads[5][:120, :1500])
    imgs = {"real": real, "zeros": np.zeros((200, 3000), np.uint8), "noise": rng.integers(0, 256, (64, 1000), dtype=np.uint8),
            "period3": np.tile(np.array([7, 200, 31], np.uint8), (90, 333)), "period5": np.tile(np.arange(5, dtype=np.uint8) * 50, (70, 400)),
            "far": np.vstack([rng.integers(0, 256, (8, 3900), dtype=np.uint8)] * 6),          # matches 31 KB back
            "ramp": (np.arange(300 * 700, dtype=np.int64).reshape(300, 700) % 251).astype(np.uint8)}
    for name, img in imgs.items():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_RLE), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FILTERED)):
            for filt in ((1,), (0,), (4, 1, 2)):
                png = write_png(img, filt, idat_split=8192, level=level, strategy=strategy)
                rc, got = decode(png)
                assert rc == _ffi.ROAM_OK and np.array_equal(got, img), (name, level, strategy, filt)
    from radarslampy_amd.synth import png_bytes_gray8                 # the writer bench.py uses: the data set's format
    png = png_bytes_gray8(real)
    assert png[8 + 25 + 8:8 + 25 + 10] == b"\x78\x01"
    rc, got = decode(png)
    assert rc == _ffi.ROAM_OK and np.array_equal(got, real) and np.array_equal(pillow(png), real)


def test_refused_formats_and_corrupt_files(tiny_payloads):
    img = np.ascontiguousarray(tiny_payloads[0][:16, :64])
    ok = write_png(img, (4,))
    assert decode(ok)[0] == _ffi.ROAM_OK
    assert decode(write_png(img, (4,), ctype=2))[0] == _ffi.ROAM_E_ARG          # RGB
    assert decode(write_png(img, (4,), depth=16))[0] == _ffi.ROAM_E_ARG         # 16-bit
    assert decode(write_png(img, (4,), interlace=1))[0] == _ffi.ROAM_E_ARG      # Adam7
    assert decode(b"not a png at all" * 4)[0] == _ffi.ROAM_E_ARG
    assert decode(ok[:len(ok) // 2])[0] == _ffi.ROAM_E_ARG                      # truncated
    big = np.ascontiguousarray(tiny_payloads[0][:64, :900])
    okb = write_png(big, (1,), level=1, strategy=zlib.Z_RLE)
    rng = np.random.default_rng(3)
    for _ in range(40):                                                         # a flipped byte anywhere in the zlib stream: refused (Adler-32 at the latest),
        bad = bytearray(okb)                                                    # never a wrong image, never a crash
        k = int(rng.integers(8 + 25 + 8 + 27, len(bad) - 16))
        bad[k] ^= int(rng.integers(1, 256))
        rc, got = decode(bytes(bad))
        assert rc == _ffi.ROAM_E_ARG or np.array_equal(got, big)
    assert decode(write_png(img, (7,)))[0] == _ffi.ROAM_E_ARG                   # no such filter
    import ctypes as C
    lib = _ffi.load_library()
    small = np.zeros(100, np.uint8)
    buf = np.frombuffer(ok, np.uint8)
    assert lib.roam_png_decode_gray8(buf.ctypes.data_as(C.c_void_p), len(ok), small.ctypes.data_as(C.c_void_p), small.size, 0, None, None) == _ffi.ROAM_E_CAPACITY


def test_pool_order_slot_lifetime_and_errors(tmp_path, tiny_payloads):
    from PIL import Image
    recs = [np.ascontiguousarray(np.pad(p, ((0, 0), (11, 3779 - 11 - p.shape[1])))) for p in tiny_payloads]     # Oxford-sized records
    paths = []
    for i, r in enumerate(recs):
        paths.append(str(tmp_path / f"{i:04d}.png"))
        Image.fromarray(r).save(paths[-1], compress_level=1)
    assert np.array_equal(parseData.readRadarRecord(paths[2]), recs[2])
    with pytest.raises(FileNotFoundError):
        parseData.readRadarRecord(str(tmp_path / "nope.png"))
    with parseData.NativeRecordReader(workers=3, hold=4) as rd:
        assert rd.depth == 4 + 6 and not rd.pinned
        seq = [i % len(paths) for i in range(40)]
        views = []
        for i, r in enumerate(rd.records([paths[j] for j in seq])):
            views.append(r)
            for back in range(min(5, len(views))):                   # this frame and the four before it are still intact
                assert np.array_equal(views[-1 - back], recs[seq[i - back]]), (i, back)
        assert len(views) == 40
        # an iteration abandoned half way leaves nothing in flight; the next one starts clean
        for i, r in enumerate(rd.records(paths)):
            if i == 2:
                break
        assert [np.array_equal(r, recs[i]) for i, r in enumerate(rd.records(paths))] == [True] * len(paths)
        # a missing file raises at ITS turn, the frames before it arrive
        got = []
        with pytest.raises(FileNotFoundError):
            for r in rd.records(paths[:3] + [str(tmp_path / "missing.png")] + paths[3:]):
                got.append(r.copy())
        assert len(got) == 3
        # not the data set's format: refused, with the file's name
        rgb = str(tmp_path / "rgb.png")
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(rgb)
        with pytest.raises(RuntimeError, match="rgb.png"):
            list(rd.records([rgb]))
        # a smaller image comes back as the view of its rows
        small = str(tmp_path / "small.png")
        Image.fromarray(recs[0][:50, :700]).save(small)
        (v,) = list(rd.records([small]))
        assert v.shape == (50, 700) and np.array_equal(v, recs[0][:50, :700])
    assert np.array_equal(parseData.readRadarRecord(rgb), np.zeros((8, 8), np.uint8))       # the single-file reader converts, as cv2 would
