#!/usr/bin/env python
"""Golden vectors for the LANCZOS resize of the input contract (reference common/datasets/image_dataset.py:16-28):
seeded uint8 RGB frames and what Pillow's ``Image.resize((w, h), Image.LANCZOS)`` -- the call the reference makes --
returns for them.  Generated with the Pillow of this image (printed below); tests/test_resample_cpu.py and
tests/test_gpu_resample.py compare the restated coefficient tables and the HIP kernels with these bytes."""
import os

import numpy as np
import PIL
from PIL import Image

CASES = [(37, 53, 20, 24), (24, 20, 64, 48), (120, 160, 75, 75), (50, 60, 50, 33), (33, 47, 33, 47), (9, 7, 3, 2),
         (96, 128, 224, 224)]


def main():
    rng = np.random.RandomState(2024)
    out = {'pillow_version': np.array(PIL.__version__)}
    for n, (H, W, oh, ow) in enumerate(CASES):
        a = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        if n % 2:                               # smooth content as well as noise
            a = np.asarray(Image.fromarray(a).resize((W, H), Image.BILINEAR).resize((W // 2 + 1, H // 2 + 1)).resize((W, H)))
        out['src_%d' % n] = a
        out['dst_%d' % n] = np.asarray(Image.fromarray(a).resize((ow, oh), Image.LANCZOS))
        # the generator's final resize (datasets/sheep/paste_and_crop_sheep.py:218: Image.LINEAR = today's Image.BILINEAR)
        out['bil_%d' % n] = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'resample_lanczos.npz')
    np.savez_compressed(path, **out)
    print('Pillow', PIL.__version__, '->', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
