"""Image I/O shim for the dataset mirrors.  The reference reads JPEG/PNG with cv2 (data/STdatas.py:50,62-65);
cv2 is optional here (absent in the build image): fall back to PIL when present, else raise a clear error.
Out of the kernel scope: only the tensor contracts matter to the hot path."""
import numpy as np


def imread(path, gray=False):
    try:
        import cv2
        im = cv2.imread(path, 0 if gray else 1)
        if im is None:
            raise FileNotFoundError(path)
        return im
    except ImportError:
        pass
    try:
        from PIL import Image
    except ImportError as e:
        raise ImportError("reading dataset images needs cv2 (as the reference) or PIL") from e
    im = Image.open(path)
    if gray:
        return np.asarray(im.convert("L"))
    return np.asarray(im.convert("RGB"))[:, :, ::-1].copy()      # BGR like cv2


def imwrite(path, arr):
    try:
        import cv2
        cv2.imwrite(path, arr)
        return
    except ImportError:
        from PIL import Image
        Image.fromarray(arr).save(path)


def resize(arr, size):
    try:
        import cv2
        return cv2.resize(arr, size)
    except ImportError:
        from PIL import Image
        return np.asarray(Image.fromarray(arr).resize(size, Image.BILINEAR))
