#!/usr/bin/env python
"""files -> device batch: images/s of the training input path (MultithreadIterator(device=...): pooled PNG decode on the host,
uint8 upload, augmentation + LANCZOS resize + / 255 on the GPU) against the images/s the step consumes.
usage: feed_bench.py [--frames 480 640] [--out 224 224] [-b 256] [--threads 16] [--batches 12] [--no-imgaug] [--jpeg]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, nargs=2, default=(480, 640))
ap.add_argument('--out', type=int, nargs=2, default=(224, 224))
ap.add_argument('-b', type=int, default=256)
ap.add_argument('--threads', type=int, default=min(16, os.cpu_count() or 1))
ap.add_argument('--batches', type=int, default=12)
ap.add_argument('--files', type=int, default=512)
ap.add_argument('--processes', type=int, default=0, help='decode processes (0: threads)')
ap.add_argument('--no-imgaug', action='store_true')
ap.add_argument('--cache', default='', choices=['', 'device', 'host'], help='decode once (frame_cache.py): also time the epochs after the first')
ap.add_argument('--jpeg', action='store_true')
ap.add_argument('--cores', type=int, default=0,
                help="pin this process and its decode workers to the first N cores: ONE rank's share of the host (an 8-GPU node with "
                     "64 cores gives each rank 8; the 1-GPU box this runs on has 16)")
args = ap.parse_args()
if args.cores:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:args.cores]))

from loans_amd.common.datasets.image_dataset import ImageDataset      # noqa: E402
from loans_amd.datasets import synthetic                                # noqa: E402
from loans_amd.runtime import training                                  # noqa: E402

root = tempfile.mkdtemp(prefix='loans_feed_')
rng = np.random.Generator(np.random.PCG64(0))
names = []
for i in range(args.files):
    img, _ = synthetic.make_composite(rng, args.frames[0], args.frames[1])
    name = 'f%04d.%s' % (i, 'jpg' if args.jpeg else 'png')
    Image.fromarray(np.asarray(img)[..., :3].astype(np.uint8)).save(os.path.join(root, name), **({'quality': 90} if args.jpeg else {'compress_level': 1}))
    names.append(name)
ds = ImageDataset(names, root, image_size=tuple(args.out), use_imgaug=not args.no_imgaug, transform_probability=0.5, augment_seed=1)

# host alone: decode (+ draws), no GPU stage
from concurrent.futures import ThreadPoolExecutor      # noqa: E402
from loans_amd.common.datasets.decode_farm import DecodeFarm      # noqa: E402
farm = DecodeFarm(args.processes) if args.processes else None
with ThreadPoolExecutor(max(args.threads, args.processes)) as pool:
    ds.decode_batch(range(min(args.b, args.files)), pool.map, farm)
    t0 = time.perf_counter()
    for k in range(3):
        ds.decode_batch(range(k * args.b % args.files, k * args.b % args.files + min(args.b, args.files)), pool.map, farm)
    decode = 3 * min(args.b, args.files) / (time.perf_counter() - t0)
if farm:
    farm.close()

# GPU stage alone on decoded frames
dec = ds.decode_batch(range(min(args.b, args.files)))
ds.finish_batch(dec, 'cuda:0')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ds.finish_batch(dec, 'cuda:0')
torch.cuda.synchronize()
gpu_stage = 5 * len(dec[0]) / (time.perf_counter() - t0)

feed = training.MultithreadIterator(ds, args.b, shuffle=True, n_threads=args.threads, n_prefetch=2, device=0, n_processes=args.processes)
next(feed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.batches):
    batch = next(feed)
torch.cuda.synchronize()
whole = args.batches * args.b / (time.perf_counter() - t0)
feed.finalize()
cached = None
if args.cache:
    # decode once: epoch 1 fills the cache (every file is visited once), the epochs after it are timed
    ds.enable_frame_cache(64, args.cache)
    feed = training.MultithreadIterator(ds, args.b, shuffle=True, n_threads=args.threads, n_prefetch=2, device=0, n_processes=args.processes)
    per_epoch = max(1, args.files // args.b)
    t0 = time.perf_counter()
    for _ in range(per_epoch + 1):
        next(feed)
    torch.cuda.synchronize()
    first = (per_epoch + 1) * args.b / (time.perf_counter() - t0)
    for _ in range(2):
        next(feed)                      # batches that were prepared before the cache was complete
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2 * args.batches):
        batch = next(feed)
    torch.cuda.synchronize()
    cached = 2 * args.batches * args.b / (time.perf_counter() - t0)
    c = ds._cache
    feed.finalize()
if args.cores:
    print('host share: %d cores (affinity of this process and its decode workers)' % len(os.sched_getaffinity(0)))
print('frames %dx%d %s -> %dx%d, batch %d, %d decode %s, augmentation %s' % (
    args.frames[0], args.frames[1], 'JPEG' if args.jpeg else 'PNG', args.out[0], args.out[1], args.b, args.processes or args.threads,
    'processes' if args.processes else 'threads',
    'naive' if args.no_imgaug else 'imgaug branch'))
print('host decode alone      : %8.0f images/s' % decode)
print('GPU stages alone       : %8.0f images/s (upload + augment + LANCZOS + /255)' % gpu_stage)
print('files -> device batches: %8.0f images/s (MultithreadIterator, prefetch 2)' % whole)
if cached is not None:
    print('decode once (%s cache): epoch 1 %8.0f images/s, epochs >= 2 %8.0f images/s (%d frames resident, %.2f GB, %d hits / %d misses)'
          % (args.cache, first, cached, len(c), c.bytes / 1e9, c.hits, c.misses))
