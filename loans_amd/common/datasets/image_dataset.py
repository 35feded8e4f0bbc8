"""``ImageDataset`` / ``LabeledImageDataset`` (reference common/datasets/image_dataset.py:47-182): the input
contract of the hot path -- float32 CHW RGB in [0,1] (exactly ``uint8 / 255``), PIL load, LANCZOS resize,
labelled variant returning ``(image, label, zeros(1))`` with bounding boxes rescaled to the resized image.

Same constructor keywords and ``get_example`` behaviour.  ``imgaug`` is not installable here: with ``use_imgaug=True`` (the
reference default) and ``transform_probability > 0`` the three operations of the reference's imgaug pipeline (:57-70) run in
the restated form of ``augment.py`` (mirror, hue / saturation jitter, crop-and-pad; host NumPy in ``get_example``, HIP kernels
in ``device_batch``, same bytes); ``use_imgaug=False`` is the reference's own naive crop / flip branch (:86-90).  Decode and
the naive crop / flip stay on host threads; ``ImageDataset.device_batch`` moves the LANCZOS resize,
``/ 255`` and the layout change to the GPU (resample.py, bit-identical to Pillow)."""
import csv
import os
import random

import numpy
from PIL import Image


def _read_image_as_array(path, dtype):
    with Image.open(path) as f:
        image = numpy.asarray(f, dtype=dtype)
    if image.ndim == 2:
        return image[numpy.newaxis]                      # (1, H, W)
    return image.transpose(2, 0, 1)                     # CHW


def _as_pil(image, image_mode):
    """A CHW (or HW) array holding 0..255 as a PIL image of ``image_mode``; values are cut to uint8 the way ``astype`` cuts them."""
    pixels = numpy.asarray(image).astype(numpy.uint8)
    if pixels.ndim == 3:
        pixels = numpy.moveaxis(pixels, 0, -1)
    return Image.fromarray(pixels).convert(image_mode)


def resize_image(image, image_size, image_mode='RGB'):
    """Pillow's LANCZOS resize to ``image_size`` = (height, width) -- the resampler resample.hip restates bit for bit
    (reference common/datasets/image_dataset.py:16-28).  float32 CHW, or HW for a single-band mode."""
    height, width = image_size
    resized = numpy.asarray(_as_pil(image, image_mode).resize((width, height), Image.LANCZOS), dtype=numpy.float32)
    return resized if resized.ndim == 2 else numpy.moveaxis(resized, -1, 0)


def random_crop(image, size, rng=random):
    H, W = image.shape[-2:]
    y = rng.randint(0, H - size[0]) if H > size[0] else 0
    x = rng.randint(0, W - size[1]) if W > size[1] else 0
    return image[:, y:y + size[0], x:x + size[1]]


def random_flip(image, x_random=False, rng=random):
    if x_random and rng.choice([True, False]):
        image = image[:, :, ::-1]
    return image


def resize_bbox(bbox, in_size, out_size):
    """chainercv.transforms.resize_bbox: boxes (y_min, x_min, y_max, x_max)."""
    bbox = bbox.copy()
    y_scale = float(out_size[0]) / in_size[0]
    x_scale = float(out_size[1]) / in_size[1]
    bbox[:, 0] = y_scale * bbox[:, 0]
    bbox[:, 2] = y_scale * bbox[:, 2]
    bbox[:, 1] = x_scale * bbox[:, 1]
    bbox[:, 3] = x_scale * bbox[:, 3]
    return bbox


class _ShapeOnly:
    """what the augmentation's random draws look at when a frame is not on the host"""

    def __init__(self, shape):
        self.shape = tuple(shape)


class ImageDataset:

    def __init__(self, paths, root='.', dtype=numpy.float32, **kwargs):
        # the reference's keywords and defaults (common/datasets/image_dataset.py:50-56)
        for name, default in (('image_size', None), ('image_mode', 'RGB'), ('transform_probability', 0), ('use_imgaug', True),
                              ('min_crop_ratio', 0.6), ('max_crop_ratio', 0.9), ('crop_always', False)):
            setattr(self, name, kwargs.pop(name, default))
        # BOTH augmentation branches draw from a stream that belongs to this dataset (reseed() restarts it): get_example and
        # device_batch consume one set of draws per example, in call order.  (The reference's naive branch draws from the global
        # `random` module; with two iterators preparing batches on threads of their own -- train and reference -- two datasets
        # on ONE global stream interleave their draws in thread-timing order and a seeded run is not reproducible)
        self._aug_rng = random.Random(kwargs.pop('augment_seed', None))
        # decode once (frame_cache.py; not in the reference): `frame_cache_gb` > 0 keeps every decoded frame -- in HBM
        # (`frame_cache_where='device'`) or in host memory ('host') -- up to that many GB; only `decode_batch` / `finish_batch`
        # (the training feed) use it, `get_example` reads the file as the reference does
        gb, where = kwargs.pop('frame_cache_gb', 0), kwargs.pop('frame_cache_where', 'device')
        self._cache = None
        if gb and gb > 0:
            self.enable_frame_cache(gb, where)
        if isinstance(paths, str):
            with open(paths) as paths_file:
                paths = [path.strip() for path in paths_file]
        self._paths, self._root, self._dtype = paths, root, dtype

    def __len__(self):
        return len(self._paths)

    def __getitem__(self, i):
        return self.get_example(i)

    def reseed(self, seed):
        self._aug_rng = random.Random(seed)

    def enable_frame_cache(self, gigabytes, where='device'):
        """keep decoded frames for the next epochs (frame_cache.py).  The naive crop / flip branch hands strided views of a frame
        to the resize: it always caches on the host."""
        from .frame_cache import FrameCache
        naive = (not self.use_imgaug) and self.transform_probability > 0
        self._cache = FrameCache(int(gigabytes * (1 << 30)), 'host' if (naive or where == 'host') else 'device')
        return self._cache

    def _imgaug_rows(self, image_chw):
        """the imgaug branch's draws for one example (None: branch off)"""
        if not (self.use_imgaug and self.transform_probability > 0):
            return None
        from .augment import sample_params
        return sample_params(self._aug_rng, image_chw.shape[-2], image_chw.shape[-1], self.transform_probability)

    def _read(self, i):
        """decode only (no random draws: safe to run on a pool thread): CHW in the dataset's dtype, values 0..255, 3 channels"""
        image = _read_image_as_array(os.path.join(self._root, self._paths[i]), self._dtype)
        if image.shape[0] == 1:
            image = numpy.tile(image, (3, 1, 1))
        return image[:3]

    def _augment(self, image, imgaug_rows=False):
        """the augmentation of one decoded example (reference :76-90); every random draw of an example happens here, so the
        order of the calls alone fixes the streams.  ``imgaug_rows``: return the imgaug branch's parameter rows beside the
        un-augmented frame instead of applying them (finish_batch applies them on the GPU)."""
        rows = self._imgaug_rows(image)
        if rows is not None:
            if imgaug_rows:
                return image, rows
            from .augment import apply_host
            u8 = numpy.ascontiguousarray(image.transpose(1, 2, 0).astype(numpy.uint8))        # reference :80-83
            return apply_host(u8, rows).astype(self._dtype).transpose(2, 0, 1)
        if imgaug_rows:
            rows = [[0] * 8] * 3
        rng = self._aug_rng
        if not self.use_imgaug and rng.random() < self.transform_probability:
            if self.crop_always or rng.random() <= 0.5:
                crop_ratio = rng.uniform(self.min_crop_ratio, self.max_crop_ratio)
                image = random_crop(image, tuple([int(size * crop_ratio) for size in image.shape[-2:]]), rng)
            image = random_flip(image, x_random=True, rng=rng)
        return (image, rows) if imgaug_rows else image

    def _decoded(self, i, imgaug_rows=False):
        """Decode + augmentation (reference :76-90): CHW in the dataset's dtype, values 0..255."""
        return self._augment(self._read(i), imgaug_rows)

    def _finish(self, image):
        """resize + ``/ 255`` of one augmented example (reference :92-98); no random draws"""
        if self.image_size is not None:
            image = resize_image(image, self.image_size, image_mode=self.image_mode)
        if image.ndim == 2:                                 # single-band mode: a channel axis of one
            image = image[numpy.newaxis]
        return numpy.ascontiguousarray(image / 255, dtype=numpy.float32)

    def get_example(self, i):
        return self._finish(self._decoded(i))

    def get_examples(self, indices, map_fn=map):
        """``[get_example(i) for i in indices]`` with decode and resize spread over ``map_fn`` (a thread pool's ``map``) and the
        random draws in between taken one example after the other in index order: the same bytes whatever the pool size"""
        images = list(map_fn(self._read, list(indices)))
        return list(map_fn(self._finish, [self._augment(image) for image in images]))

    def get_raw_example(self, i):
        """The frame as it stands before ``resize_image``: uint8 HWC RGB (what ``Image.fromarray(...astype('uint8'))``
        sees at reference :20).  ``device_batch`` finishes the example on the GPU."""
        return numpy.ascontiguousarray(self._decoded(i).transpose(1, 2, 0).astype(numpy.uint8))

    def _read_u8(self, i):
        """decode only, straight to what the GPU stages take: uint8 HWC RGB (no float round trip, no transposes -- the
        values ``_read`` + ``astype('uint8')`` give for 8-bit files); None for anything else (16-bit, float files), which goes
        through ``_read``"""
        with Image.open(os.path.join(self._root, self._paths[i])) as f:
            image = numpy.asarray(f)
        if image.dtype != numpy.uint8:
            return None
        if image.ndim == 2:
            return numpy.repeat(image[:, :, None], 3, axis=2)
        if image.shape[2] == 1:
            return numpy.repeat(image, 3, axis=2)
        return image[:, :, :3] if image.shape[2] >= 3 else None

    def _augment_u8(self, image):
        """``_augment(..., imgaug_rows=True)`` on a uint8 HWC frame: the same draws in the same order"""
        rows = self._imgaug_rows(image.transpose(2, 0, 1))
        if rows is not None:
            return image, rows
        rng = self._aug_rng
        if not self.use_imgaug and rng.random() < self.transform_probability:
            if self.crop_always or rng.random() <= 0.5:
                crop_ratio = rng.uniform(self.min_crop_ratio, self.max_crop_ratio)
                image = random_crop(image.transpose(2, 0, 1), tuple([int(size * crop_ratio) for size in image.shape[:2]]), rng).transpose(1, 2, 0)
            image = random_flip(image.transpose(2, 0, 1), x_random=True, rng=rng).transpose(1, 2, 0)
        return image, [[0] * 8] * 3

    def decode_batch(self, indices, map_fn=map, farm=None):
        """Host half of ``device_batch``: the frames of a batch as uint8 HWC RGB arrays (as ``Image.fromarray(...astype('uint8'))``
        sees them at reference :20) with the imgaug branch's parameter rows beside them.  ``map_fn`` decodes (a thread pool's
        ``map``: PIL releases the GIL while it inflates); the random draws are taken afterwards, one example after the other in
        index order, so a pooled batch consumes the streams exactly like ``[get_example(i) for i in indices]``."""
        if self.image_mode != 'RGB' or self.image_size is None:
            raise ValueError('device_batch covers the training configuration: RGB frames resized to image_size')
        indices = list(indices)
        cache = self._cache
        cached = {i: cache.lookup(i) for i in indices} if cache is not None else {}
        todo = [i for i in indices if cached.get(i) is None]
        if farm is not None:        # decode processes (decode_farm.py): Pillow's decoders hold the GIL, threads do not scale
            fresh = farm.decode([os.path.join(self._root, self._paths[i]) for i in todo], map_fn) if todo else []
        else:
            fresh = list(map_fn(self._read_u8, todo))
        fresh = dict(zip(todo, fresh))
        frames, rows = [], []
        for i in indices:
            image = cached.get(i)
            if image is None:
                image = fresh[i]
                if image is None or isinstance(image, int):          # not an 8-bit file, or unreadable (the read raises here)
                    image, r = self._augment(self._read(i), imgaug_rows=True)
                    frames.append(image.transpose(1, 2, 0).astype(numpy.uint8))
                    rows.append(r)
                    continue
                if cache is not None:
                    cache.keep_host(i, image)
            if isinstance(image, numpy.ndarray):
                image, r = self._augment_u8(image)
            else:                                       # resident in HBM: the draws need the frame's size only
                r = self._imgaug_rows(_ShapeOnly((3,) + image.shape[:2])) or [[0] * 8] * 3
            frames.append(image)                      # possibly a strided view (crop / flip): frames_to_device copies it once
            rows.append(r)
        rows = rows if self.use_imgaug and self.transform_probability > 0 else None
        return (frames, rows, indices) if (cache is not None and cache.where == 'device') else (frames, rows)

    def finish_batch(self, decoded, device, map_fn=map):
        """Device half: upload the uint8 frames, run the imgaug stages, the LANCZOS resize, ``/ 255`` and the CHW layout on the
        GPU (resample.py / augment.py, the bytes of the host path).  ``map_fn`` spreads the copies into the pinned staging
        buffer over a pool."""
        from .resample import frames_to_device
        if len(decoded) == 3:                           # frames resident in HBM among them (frame_cache.py)
            frames, rows, indices = decoded
            return self._cache.assemble(frames, rows, indices, self.image_size, device, map_fn)
        frames, rows = decoded
        return frames_to_device(frames, self.image_size, device, augment_rows=rows, map_fn=map_fn)

    def device_batch(self, indices, device, map_fn=map):
        """``concat_examples([self.get_example(i) for i in indices])`` with the LANCZOS resize, ``/ 255`` and the CHW
        layout done on the GPU (bit-identical, see resample.py): uint8 frames cross PCIe, not float32 ones."""
        return self.finish_batch(self.decode_batch(indices, map_fn), device, map_fn)


class LabeledImageDataset:

    def __init__(self, pairs, root='.', dtype=numpy.float32, label_dtype=numpy.int32, image_size=None,
                 image_mode='RGB', transform_probability=0, return_dummy_scores=True):
        if isinstance(pairs, str):
            with open(pairs) as pairs_file:
                reader = csv.reader(pairs_file, delimiter='\t')
                pairs = [(pair[0], list(map(label_dtype, pair[1:]))) for pair in reader]
        self.transform_probability = transform_probability
        self._pairs, self._root, self._dtype, self._label_dtype = pairs, root, dtype, label_dtype
        self.image_size, self.image_mode, self.return_dummy_scores = image_size, image_mode, return_dummy_scores

    def __len__(self):
        return len(self._pairs)

    def __getitem__(self, i):
        return self.get_example(i)

    def shrink_dataset(self, new_size):
        self._pairs = self._pairs[:new_size]

    def check_for_bad_label(self, label, image_size):
        """Boxes are (y_min, x_min, y_max, x_max) in pixels of the frame as it was read; one that leaves the frame by more than
        a tenth of the frame's side means labels and images do not belong together (reference :137-145: AssertionError)."""
        height, width = image_size
        slack_y, slack_x = height * 0.1, width * 0.1
        fits = (label[:, 0] >= -slack_y).all() and (label[:, 1] >= -slack_x).all() and \
            (label[:, 2] <= height + slack_y).all() and (label[:, 3] <= width + slack_x).all()
        if not fits:
            raise AssertionError(f'bounding boxes {label} do not lie in a frame of {height} x {width} px (+- 10 %): '
                                 'was the dataset written for other image sizes?')

    def _base_example(self, i):
        path, label = self._pairs[i]
        image = _read_image_as_array(os.path.join(self._root, path), self._dtype)
        return image, numpy.array(label, dtype=self._label_dtype)

    def get_example(self, i):
        try:
            image, label = self._base_example(i)
        except Exception as e:          # reference :148-152: fall back to sample 0
            print(e)
            image, label = self._base_example(0)

        if label.ndim and len(label) % 4 == 0:
            label = label.reshape(len(label) // 4, -1)              # a flat run of corner values: one row per box

        if image.shape[0] == 1:                                     # greyscale file: three equal channels
            image = numpy.repeat(image, 3, axis=0)
        image = image[:3]

        if self.image_size is not None:
            read_size = image.shape[-2:]
            if label.ndim > 1:                                      # boxes follow the frame to its new size
                self.check_for_bad_label(label, read_size)
                label = resize_bbox(label.astype(numpy.float32), read_size, self.image_size)
            label = label.astype(self._label_dtype)
            image = resize_image(image, self.image_size, image_mode=self.image_mode)

        if image.ndim == 2:
            image = image[numpy.newaxis]

        image = numpy.ascontiguousarray(image / 255, dtype=numpy.float32)
        if self.return_dummy_scores:
            return image, label, numpy.zeros((1,))
        return image, label
