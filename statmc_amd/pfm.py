"""PFM codec for statistics dumps (the `<stem>-<spp>-<buffer>.pfm` files of the reference,
src/statistics/buffer.cpp:40-53 / statpath.cpp:449-454): 32-bit float, 1 or 3 channels (RGB
order in the file), rows bottom-to-top, negative scale = little endian.  `n` is stored as float."""
import numpy as np


def write_pfm(path, img):
    img = np.asarray(img)
    if img.dtype != np.float32:
        img = img.astype(np.float32)          # Buffer::outMat: n is converted to CV_32F (buffer.h:51-54)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[..., 0]
    if img.ndim == 2:
        magic = b"Pf"
    elif img.ndim == 3 and img.shape[2] == 3:
        magic = b"PF"
    else:
        raise ValueError("PFM holds 1 or 3 channels")
    h, w = img.shape[:2]
    with open(path, "wb") as f:
        f.write(magic + b"\n%d %d\n-1.000000\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1]).astype("<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        if magic not in (b"PF", b"Pf"):
            raise ValueError("not a PFM file: %s" % path)
        dims = f.readline().split()
        while len(dims) < 2:
            dims += f.readline().split()
        w, h = int(dims[0]), int(dims[1])
        scale = float(f.readline().strip())
        c = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(w * h * c * 4), dtype="<f4" if scale < 0 else ">f4")
    img = data.reshape(h, w, c)[::-1].astype(np.float32)
    return np.ascontiguousarray(img if c == 3 else img[..., 0])
