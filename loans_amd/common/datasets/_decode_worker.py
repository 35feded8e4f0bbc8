"""Decode worker of ``decode_farm.DecodeFarm`` -- run as a script (``python _decode_worker.py``), never imported: it needs
Pillow and NumPy only, so a worker starts in a tenth of a second and never loads torch or the HIP library.

Protocol (binary, over stdin / stdout): request = one UTF-8 path per line; reply = 12-byte header ``<status, H, W>`` (three
little-endian int32) followed, for status 0, by H * W * 3 bytes: the frame as uint8 HWC RGB, exactly what
``ImageDataset._read_u8`` returns (reference common/datasets/image_dataset.py:31-44,76-79: PIL load, grey -> 3 channels, first
3 channels).  status 1: not an 8-bit image (the parent decodes it itself); status 2: the file could not be read (the parent
repeats the read to raise the real error).  EOF on stdin ends the worker, so it cannot outlive its parent."""
import struct
import sys

import numpy
from PIL import Image


def decode(path):
    with Image.open(path) as f:
        image = numpy.asarray(f)
    if image.dtype != numpy.uint8:
        return None
    if image.ndim == 2:
        return numpy.repeat(image[:, :, None], 3, axis=2)
    if image.shape[2] == 1:
        return numpy.repeat(image, 3, axis=2)
    return numpy.ascontiguousarray(image[:, :, :3]) if image.shape[2] >= 3 else None


def main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        line = inp.readline()
        if not line:
            return
        try:
            image = decode(line.rstrip(b'\n').decode('utf-8'))
        except Exception:
            out.write(struct.pack('<iii', 2, 0, 0))
            out.flush()
            continue
        if image is None:
            out.write(struct.pack('<iii', 1, 0, 0))
        else:
            out.write(struct.pack('<iii', 0, image.shape[0], image.shape[1]))
            out.write(memoryview(numpy.ascontiguousarray(image)).cast('B'))
        out.flush()


if __name__ == '__main__':
    main()
