"""-m gpu parity tests of the device JPEG front-end (C ABI df3d_jpeg_decode_luma) -- integer work, so BIT-EXACT:
against libjpeg-turbo itself (through Pillow, the library the reference's loaders sit on) and against the C
oracle, on the reference's own test JPEGs and on encoder-made files that cover the format's corners."""
import glob
import io

import numpy as np
import pytest
from PIL import Image

from oracle import jpeg as oj

pytestmark = pytest.mark.gpu


def pil_luma(blob):
    im = Image.open(io.BytesIO(blob))
    if im.mode != "L":
        im.draft("L", im.size)  # libjpeg's own grayscale output of a YCbCr file: the luma plane
    assert im.mode == "L"
    return np.asarray(im)


def _smooth(rng, h, w, c=None):
    shape = (h // 4 + 2, w // 4 + 2) + ((c,) if c else ())
    a = np.kron(rng.random(shape) * 255, np.ones((4, 4) + ((1,) if c else ())))[:h, :w]
    return (a + rng.normal(0, 6, a.shape)).clip(0, 255).astype(np.uint8)


def _encode(img, **kw):
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", **kw)
    return buf.getvalue()


SEQ = pytest.mark.parametrize("sequential", [False, True], ids=["parallel", "sequential"])


@SEQ
def test_reference_test_images_bit_exact(native_lib, cuda, golden_dir, sequential):
    from deepfly3d_amd import jpeg

    blobs = [open(p, "rb").read() for p in sorted(glob.glob(f"{golden_dir}/images/*.jpg"))]
    assert len(blobs) == 14
    out = jpeg.decode_luma(blobs, 960, 480, sequential=sequential).cpu().numpy()
    for i, b in enumerate(blobs):
        assert np.array_equal(out[i], pil_luma(b))
        assert np.array_equal(out[i], oj.decode_luma(b))
        assert np.array_equal(out[i], np.asarray(Image.open(io.BytesIO(b)).convert("L")))  # chroma-neutral: also the RGB->L image


@SEQ
@pytest.mark.parametrize("hw", [(8, 8), (16, 16), (17, 33), (100, 75), (1, 1), (7, 250), (480, 960)])
def test_encoder_matrix_bit_exact(native_lib, cuda, hw, sequential):
    """Qualities 30..100 (8- and 16-entry code lengths, long codes), standard and optimised Huffman tables,
    grayscale / 4:4:4 / 4:2:2 / 4:2:0, restart intervals, sizes that are not multiples of the MCU."""
    from deepfly3d_amd import jpeg

    h, w = hw
    rng = np.random.default_rng(h * 1000 + w)
    blobs = []
    for q in (30, 75, 95, 100):
        for kw in ({}, {"optimize": True}, {"subsampling": 0}, {"subsampling": 1}, {"subsampling": 2}, {"restart_marker_blocks": 5},
                   {"restart_marker_rows": 1}):
            for colour in (False, True):
                if not colour and "subsampling" in kw:
                    continue
                blobs.append(_encode(_smooth(rng, h, w, 3 if colour else None), quality=q, **kw))
    out = jpeg.decode_luma(blobs, w, h, sequential=sequential).cpu().numpy()
    for i, b in enumerate(blobs):
        assert np.array_equal(out[i], oj.decode_luma(b)), f"file {i} differs from the oracle"
        assert np.array_equal(out[i], pil_luma(b)), f"file {i} differs from libjpeg"


@SEQ
def test_noise_images_long_codes(native_lib, cuda, sequential):
    """White noise at quality 100: almost every coefficient non-zero, 16-bit Huffman codes, big magnitudes."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(3)
    blobs = [_encode(rng.integers(0, 256, size=(64, 96), dtype=np.uint8), quality=100, optimize=o) for o in (False, True)]
    blobs += [_encode(rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8), quality=100, subsampling=2)]
    out = jpeg.decode_luma(blobs, 96, 64, sequential=sequential).cpu().numpy()
    for i, b in enumerate(blobs):
        assert np.array_equal(out[i], pil_luma(b))


def test_large_app_segment_and_fill_bytes(native_lib, cuda):
    """A 20 KB APP1 segment in front of the tables (longer than the parser's LDS window) and a COM segment."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(4)
    b = _encode(_smooth(rng, 48, 64), quality=80)
    app = b"\xff\xe1" + (20000 + 2).to_bytes(2, "big") + bytes(rng.integers(0, 256, size=20000, dtype=np.uint8))
    com = b"\xff\xfe" + (5 + 2).to_bytes(2, "big") + b"hello"
    fat = b[:2] + app + com + b[2:]
    out = jpeg.decode_luma([fat, b], 64, 48).cpu().numpy()
    assert np.array_equal(out[0], pil_luma(b)) and np.array_equal(out[1], pil_luma(b))


@SEQ
def test_status_codes_and_mixed_batches(native_lib, cuda, sequential):
    from deepfly3d_amd import jpeg as jpeg_mod

    class jpeg:  # route every call of this test through the selected Huffman path
        JpegDecodeError = jpeg_mod.JpegDecodeError

        @staticmethod
        def decode_luma(*a, **kw):
            return jpeg_mod.decode_luma(*a, sequential=sequential, **kw)

    rng = np.random.default_rng(5)
    img = _smooth(rng, 40, 56)
    good = _encode(img, quality=85)
    progressive = _encode(img, quality=85, progressive=True)
    other_size = _encode(_smooth(rng, 32, 56), quality=85)
    not_jpeg = b"\x89PNG\r\n\x1a\n" + bytes(64)
    cut_header = good[:60]
    batch = [good, progressive, other_size, not_jpeg, cut_header, good]
    out, st = jpeg.decode_luma(batch, 56, 40, check=False, return_status=True)
    assert list(st) == [0, 3, 5, 2, 1, 0]
    for i, b in enumerate(batch):
        assert oj.status(b, 56, 40) == st[i]  # the oracle classifies the same way
    out = out.cpu().numpy()
    assert np.array_equal(out[0], pil_luma(good)) and np.array_equal(out[5], pil_luma(good))
    with pytest.raises(jpeg.JpegDecodeError, match="file 1 of the batch: unsupported"):
        jpeg.decode_luma(batch, 56, 40)
    # a file cut inside the entropy-coded data decodes the intact MCU rows and never reads out of bounds
    cut = good[: len(good) * 2 // 3]
    got, st = jpeg.decode_luma([cut], 56, 40, check=False, return_status=True)
    assert st[0] == 0 and np.array_equal(got.cpu().numpy()[0][:8], pil_luma(good)[:8])
    assert jpeg.decode_luma([], 56, 40).shape == (0, 40, 56)


@SEQ
def test_oversubscribed_huffman_table_is_rejected(native_lib, cuda, sequential):
    """Untrusted DHT whose code counts over-subscribe a length (12 codes of 1 bit; 255 of 1 bit): the table builder
    must flag the file as corrupt without indexing past its 2^LB-entry look-up table, and the neighbours of the
    batch decode normally (the oracle classifies the same way)."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(6)
    good = _encode(_smooth(rng, 40, 56), quality=85)
    i = good.index(b"\xff\xc4")
    nv = sum(good[i + 5 : i + 21])
    crafted = bytearray(good)
    crafted[i + 5 : i + 21] = bytes([nv] + [0] * 15)
    # a DHT segment declaring 255 one-bit codes (segment grown accordingly)
    seg_len = int.from_bytes(good[i + 2 : i + 4], "big")
    big = bytearray(good[: i + 2]) + (seg_len - nv + 255).to_bytes(2, "big") + good[i + 4 : i + 5] + bytes([255] + [0] * 15) + bytes(range(255)) + good[i + 4 + seg_len - 2 :]
    batch = [good, bytes(crafted), bytes(big), good]
    out, st = jpeg.decode_luma(batch, 56, 40, check=False, return_status=True, sequential=sequential)
    assert list(st) == [0, 4, 4, 0]
    assert [oj.status(b, 56, 40) for b in batch] == [0, 4, 4, 0]
    out = out.cpu().numpy()
    assert np.array_equal(out[0], pil_luma(good)) and np.array_equal(out[3], pil_luma(good))


def test_many_files_one_call(native_lib, cuda, golden_dir):
    """224 files (one 32-frame step of the pipeline) in one call; every copy decodes identically."""
    from deepfly3d_amd import jpeg

    blobs = [open(p, "rb").read() for p in sorted(glob.glob(f"{golden_dir}/images/*.jpg"))] * 16
    out = jpeg.decode_luma(blobs, 960, 480)
    ref = out[:14]
    for r in range(1, 16):
        assert bool((out[14 * r : 14 * (r + 1)] == ref).all())


def test_parallel_decoder_paths(native_lib, cuda, golden_dir):
    """The reference's JPEGs take the parallel (self-synchronising) Huffman kernel; restart-interval files fall back to the
    sequential kernel; all bit-identical."""
    from deepfly3d_amd import jpeg

    blobs = [open(p, "rb").read() for p in sorted(glob.glob(f"{golden_dir}/images/*.jpg"))]
    ref = np.stack([pil_luma(b) for b in blobs])
    out, path = jpeg.decode_luma(blobs, 960, 480, return_path=True)
    assert np.array_equal(out.cpu().numpy(), ref) and (path > 0).all() and path.max() <= 12  # value = synchronisation passes
    out, path = jpeg.decode_luma(blobs, 960, 480, return_path=True, sequential=True)
    assert np.array_equal(out.cpu().numpy(), ref) and (path == 0).all()
    rng = np.random.default_rng(8)
    img = _smooth(rng, 96, 160)
    col = _smooth(rng, 96, 160, 3)
    mix = [_encode(img, quality=90, restart_marker_blocks=7), _encode(img, quality=90), _encode(col, quality=90)]
    out, path = jpeg.decode_luma(mix, 160, 96, return_path=True)
    # restart markers -> sequential; single-component stream -> parallel; a busy 4:2:0 colour stream may or may not
    # synchronise its block-in-MCU index within the pass budget (then the sequential kernel decodes it): same pixels
    assert path[0] == 0 and path[1] > 0
    for i, b in enumerate(mix):
        assert np.array_equal(out.cpu().numpy()[i], pil_luma(b))


def test_large_files_with_long_chunks(native_lib, cuda):
    """Noise at quality 97, 1600x1200: files of several hundred KB (chunks of ~10 000 bits per lane), and a larger block grid
    than the camera frames."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(9)
    blobs = [_encode(rng.integers(0, 256, size=(1200, 1600), dtype=np.uint8), quality=97),
             _encode(_smooth(rng, 1200, 1600, 3), quality=97, subsampling=2)]
    assert max(len(b) for b in blobs) > 200 * 1024
    out, path = jpeg.decode_luma(blobs, 1600, 1200, return_path=True)
    for i, b in enumerate(blobs):
        assert np.array_equal(out[i].cpu().numpy(), pil_luma(b)), f"file {i} (path {path[i]})"


def _segments(blob):
    """[(marker, start, end)] of the marker segments in front of the entropy-coded data; end of the SOS header."""
    out, pos = [], 2
    while True:
        assert blob[pos] == 0xFF
        marker = blob[pos + 1]
        length = int.from_bytes(blob[pos + 2 : pos + 4], "big")
        out.append((marker, pos, pos + 2 + length))
        pos += 2 + length
        if marker == 0xDA:
            return out


def _relabel_tables(blob, dc_ids, ac_ids, extra):
    """Same entropy-coded data, other table numbering: `extra` = [(class, new id, copy of id)] tables are added as copies and
    scan component i is pointed at DC table dc_ids[i] / AC table ac_ids[i]."""
    segs = _segments(blob)
    tables = {}
    for m, a, b in segs:
        if m == 0xC4:
            p = a + 4
            while p < b:
                tc_th = blob[p]
                n = sum(blob[p + 1 : p + 17])
                tables[tc_th] = blob[p + 1 : p + 17 + n]
                p += 17 + n
    add = b""
    for tc, new, src in extra:
        body = bytes([(tc << 4) | new]) + tables[(tc << 4) | src]
        add += b"\xff\xc4" + (len(body) + 2).to_bytes(2, "big") + body
    m, a, b = segs[-1]
    sos = bytearray(blob[a:b])
    ns = sos[4]
    for i in range(ns):
        sos[5 + 2 * i + 1] = (dc_ids[i] << 4) | ac_ids[i]
    return blob[:a] + add + bytes(sos) + blob[b:]


def test_table_numbering_variants(native_lib, cuda):
    """The parallel decoder keeps the (at most four) tables a scan refers to in compact rows: scans that use other table ids,
    or more than four tables (copies under new ids: every component its own pair -> sequential kernel), decode the same."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(21)
    col = _smooth(rng, 96, 160, 3)
    for sub in (0, 2):
        b = _encode(col, quality=88, subsampling=sub)
        ref = pil_luma(b)
        variants = {
            "as encoded": (b, True),
            # chroma tables moved to ids 3 / 2 (still four tables)
            "other ids": (_relabel_tables(b, [0, 3, 3], [0, 2, 2], [(0, 3, 1), (1, 2, 1)]), True),
            # Cr gets its own copies: six tables in the scan
            "six tables": (_relabel_tables(b, [0, 1, 2], [0, 1, 2], [(0, 2, 1), (1, 2, 1)]), False),
            # Cb and Cr share a DC table but use different AC tables (five tables)
            "shared DC, own AC": (_relabel_tables(b, [0, 1, 1], [0, 1, 2], [(1, 2, 1)]), False),
        }
        for name, (v, parallel) in variants.items():
            assert np.array_equal(pil_luma(v), ref), name  # libjpeg agrees that the relabelled file is the same picture
            out, path = jpeg.decode_luma([v], 160, 96, return_path=True)
            assert np.array_equal(out.cpu().numpy()[0], ref), (sub, name)
            if not parallel:
                assert path[0] == 0, (sub, name, path)   # too many tables for the parallel kernel's rows
            assert np.array_equal(jpeg.decode_luma([v], 160, 96, sequential=True).cpu().numpy()[0], ref), (sub, name)


def test_folder_reader_grows_its_staging_buffer(native_lib, cuda, tmp_path):
    """The reader sizes its pinned staging buffer from the first file of a batch; a batch whose other files are much larger
    makes the native reader report the size (nothing read), the buffer is replaced and the batch read again."""
    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(22)
    flat = _encode(np.full((96, 160), 128, np.uint8), quality=50)                       # a few hundred bytes
    busy = [_encode(rng.integers(0, 256, size=(96, 160), dtype=np.uint8), quality=95) for _ in range(6)]
    assert min(len(b) for b in busy) > 20 * len(flat)
    blobs = [flat] + busy + [flat]
    paths = []
    for i, b in enumerate(blobs):
        paths.append(str(tmp_path / f"camera_0_img_{i}.jpg"))
        open(paths[-1], "wb").write(b)
    rd = jpeg.JpegFolderReader(160, 96, cuda, pinned=True)
    try:
        got = [luma.cpu().numpy().copy() for luma in rd.stream([paths[:5], paths[5:]])]
    finally:
        rd.finish()
    out = np.concatenate(got)
    for i, b in enumerate(blobs):
        assert np.array_equal(out[i], pil_luma(b)), i
    rd = jpeg.JpegFolderReader(160, 96, cuda)
    with pytest.raises(FileNotFoundError):
        try:
            list(rd.stream([paths[:2] + [str(tmp_path / "camera_0_img_99.jpg")]]))
        finally:
            rd.finish()


def test_folder_reader_fails_within_two_batches_of_a_bad_file(native_lib, cuda, tmp_path):
    """A corrupt and a progressive frame in the middle of a six-batch folder: JpegDecodeError names the file while the
    stream is still running -- at most two batches later, not in finish() after the whole folder has been inferred."""
    import torch

    from deepfly3d_amd import jpeg

    rng = np.random.default_rng(5)
    good = [_encode(rng.integers(0, 256, size=(96, 160), dtype=np.uint8), quality=80) for _ in range(24)]
    cut = good[9][:60]                                                   # cut inside the header: "truncated file" (a cut inside the
                                                                         # entropy-coded data decodes the intact rows, like libjpeg)
    prog = io.BytesIO()
    Image.fromarray(rng.integers(0, 256, size=(96, 160), dtype=np.uint8)).save(prog, "JPEG", quality=80, progressive=True)
    for bad_index, blob, word in ((9, cut, "truncated"), (5, prog.getvalue(), "unsupported")):
        folder = tmp_path / f"bad{bad_index}"
        folder.mkdir()
        paths = []
        for i, b in enumerate(good):
            paths.append(str(folder / f"camera_0_img_{i}.jpg"))
            open(paths[-1], "wb").write(blob if i == bad_index else b)
        batches = [paths[i : i + 4] for i in range(0, 24, 4)]             # the bad file sits in batch bad_index // 4
        rd = jpeg.JpegFolderReader(160, 96, cuda)
        seen = 0
        with pytest.raises(jpeg.JpegDecodeError, match=word) as err:
            try:
                for luma in rd.stream(batches):
                    seen += 1
                    torch.cuda.synchronize()
            finally:
                rd.finish(check=False)
        assert paths[bad_index] in str(err.value)
        assert seen <= bad_index // 4 + 3 and seen < len(batches), (seen, bad_index)
    # pending statuses are still checked by finish() when the consumer stops early
    rd = jpeg.JpegFolderReader(160, 96, cuda)
    paths_bad_last = [str(tmp_path / "bad9" / f"camera_0_img_{i}.jpg") for i in (8, 9)]
    list(rd.stream([paths_bad_last]))
    with pytest.raises(jpeg.JpegDecodeError):
        rd.finish()
