#!/usr/bin/env python
"""The reference's training driver (train_sheep_localizer.py:50-255) for the MI355X build: the same command line --
positional ``train_file val_file reference_file``, ``--no-validation``, ``--num-epoch``, ``--no-imgaug``, ``--ln``,
``--no-snapshot-every-epoch`` ... (:52-74) --, the same wiring (datasets -> MultithreadIterator -> SheepLocalizer /
ResnetAssessor -> two Adam(amsgrad=True) -> SheepAssessor -> loop until ``num_epoch`` -> snapshots), the validation loop of
:106-113,192-197 (a validation iterator run through ``SheepMAPEvaluator`` at every log interval / new epoch) and the
timestamped ``<log-dir>/<iso time>_<log name>/`` directory with a JSON ``log`` whose first entry carries the configuration
(:158-180).

    python train_sheep_localizer.py train.txt val.csv reference/images.csv --use-resnet-18 -b 64 -g 0

Input path (MI355X side): frames are decoded by a pool of host threads one or two batches ahead and finished on the GPU --
uint8 upload, the augmentation branch (``ImageDataset(use_imgaug=..., transform_probability=0.5)``, :84-91), LANCZOS resize,
``/ 255``, NCHW -- on a stream of their own (``loans_amd.runtime.training.MultithreadIterator(device=...)``), so a step
starts with its batch resident in HBM.

Extensions (not in the reference, all optional): the literal ``synthetic`` for any of the three files = the seeded synthetic
paste-and-crop generator (like the reference's own ``mnist`` literal for the reference file, :94-96) -- omitted positionals
default to it --, ``--iterations N`` (stop early), ``--gpus N`` (data parallel, forks its own ranks), ``--dtype bf16``,
``--use-graph``, seeds.  The Trainer extensions that are not on the path (BBOXPlotter and its ``--port`` / ``--test-image``,
``--anchor-image``, Logger's source copies, the interactive prompt, dump_graph) are accepted on the command line and absent.

    python train_sheep_localizer.py --use-resnet-18 -b 64 --iterations 20
    python train_sheep_localizer.py --gpus 8 --use-resnet-18 -b 128        (forks its own 8 ranks)
"""
import argparse
import datetime
import json
import os
import sys
import time

if __name__ == "__main__":
    # `--gpus N` (N > 1) outside a launcher: fork the N ranks as a child process group before anything touches the GPU,
    # the way the reference's data-parallel trainer forks its workers (schaaaafrichter/train.py:159-191); does not return
    import importlib.util
    _spec = importlib.util.spec_from_file_location(
        '_loans_launch', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'loans_amd', 'launch.py'))
    _launch = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_launch)
    _launch.launch_if_parent(os.path.abspath(__file__))

import numpy as np          # noqa: E402
import torch                # noqa: E402

import loans_amd            # noqa: E402
from loans_amd import parallel                  # noqa: E402
from loans_amd.datasets import synthetic        # noqa: E402
from loans_amd.runtime import training          # noqa: E402

SYNTHETIC = 'synthetic'


class SyntheticFrames:
    """`ImageDataset` stand-in: float32 CHW RGB frames in [0,1] (common/datasets/image_dataset.py:47-98)."""

    def __init__(self, n, image_size, seed=0):
        self.frames = synthetic.make_frames(seed, n, image_size[0], image_size[1])

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i]

    get_example = __getitem__


class SyntheticAssessorSamples:
    """`LabeledImageDataset` stand-in: (crop, IoU label, zeros(1)) triples (image_dataset.py:180-181)."""

    def __init__(self, n, target_size, seed=1):
        self.x, self.y = synthetic.make_assessor_batch(seed, n, target_size[0], target_size[1])

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i], np.zeros(1, np.float32)

    get_example = __getitem__


class SyntheticValidationFrames:
    """Validation `LabeledImageDataset` stand-in: (frame, ground-truth box (1, 4) = (top, left, bottom, right) in pixels,
    zeros(1)) -- the box is where the generator pasted the stamp (datasets/sheep/paste_and_crop_sheep.py:109-136)."""

    def __init__(self, n, image_size, seed=2):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.frames, self.boxes = [], []
        for _ in range(n):
            img, (x0, y0, x1, y1) = synthetic.make_composite(rng, image_size[0], image_size[1])
            self.frames.append(synthetic.to_chw_float(img))
            self.boxes.append(np.array([[y0, x0, y1, x1]], np.float32))

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i], self.boxes[i], np.zeros(1, np.float32)

    get_example = __getitem__


def load_train_paths(train_file, with_label=False):
    """The reference's ``.json`` ground-truth lists (its train_sheep_localizer.py:24-32): one record per frame,
    ``{"image": path, "bounding_boxes": [box, ...]}``.  Plain path list, or (path, first box) pairs for the labelled datasets."""
    with open(train_file) as f:
        records = json.load(f)
    if not with_label:
        return [record['image'] for record in records]
    return [(record['image'], record['bounding_boxes'][0]) for record in records]


def load_pretrained_model(model_file, model):
    loans_amd.load_npz(model_file, model, strict=False)       # NpzDeserializer(strict=False), reference :45-47


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Train a sheep localizer (MI355X-native LoANs hot path)")
    # ---- the reference's command line (:52-74), same names, defaults and meaning ----
    parser.add_argument("train_file", nargs='?', default=SYNTHETIC, help="path to train csv ('synthetic': seeded synthetic frames)")
    parser.add_argument("val_file", nargs='?', default=SYNTHETIC,
                        help="path to validation file (if you do not want to do validation just enter gibberish here)")
    parser.add_argument("reference_file", nargs='?', default=SYNTHETIC, help="path to reference images with different zoom levels")
    parser.add_argument("--no-validation", dest='validation', action='store_false', default=True, help="don't do validation")
    parser.add_argument("--image-size", type=int, nargs=2, default=(224, 224), help="input size for localizer")
    parser.add_argument("--target-size", type=int, nargs=2, default=(75, 75), help="crop size for each image")
    parser.add_argument("-b", "--batch-size", type=int, default=16, help="batch size for training (per GPU)")
    parser.add_argument("-g", "--gpu", type=int, default=-1,
                        help="gpu id to use (the reference's -1 = cpu does not exist here: -1 takes the current device)")
    parser.add_argument("--lr", "--learning-rate", dest="learning_rate", type=float, default=0.001, help="learning rate")
    parser.add_argument("-l", "--log-dir", default='sheep_logs', help="path to log dir")
    parser.add_argument("--ln", "--log-name", dest="ln", default="test", help="name of log")
    parser.add_argument("--num-epoch", type=int, default=100, help="number of epochs to train")
    parser.add_argument("--snapshot-interval", type=int, default=1000, help="number of iterations after which a snapshot will be taken")
    parser.add_argument("--no-snapshot-every-epoch", dest="snapshot_every_epoch", action='store_false', default=True,
                        help="Do not take a snapshot on every epoch")
    parser.add_argument("--log-interval", type=int, default=100, help="log interval")
    parser.add_argument("--port", type=int, default=1337, help="(BBOXPlotter: accepted, not used)")
    parser.add_argument("--test-image", help="(BBOXPlotter: accepted, not used)")
    parser.add_argument("--anchor-image", help="(accepted, not used)")
    parser.add_argument("--rl", dest="resume_localizer", help="localizer snapshot (.npz) to start from (reference --rl)")
    parser.add_argument("--rd", dest="resume_discriminator", help="assessor snapshot (.npz) to start from; also FREEZES the assessor (reference --rd)")
    parser.add_argument("--use-resnet-18", action='store_true', default=False, help="ResNet-18-variant localizer (sheep/resnet.py) instead of the ResNet-50 one")
    parser.add_argument("--localizer-target", type=float, default=1.0, help="the IoU score the localizer is trained towards, in [0, 1]")
    parser.add_argument("--no-imgaug", action='store_false', dest='use_imgaug', default=True,
                        help="the naive crop / flip augmentation instead of the imgaug pipeline")
    # ---- extensions ----
    parser.add_argument("--validation", dest='validation', action='store_true', help="(default; kept for older command lines)")
    parser.add_argument("--iterations", type=int, default=None, help="stop after this many iterations (before --num-epoch epochs)")
    parser.add_argument("--gpus", type=int, default=1,
                        help="data-parallel ranks on this node, one process per GPU, forked by this script itself")
    parser.add_argument("--dataset-size", type=int, default=256, help="synthetic training / reference examples")
    parser.add_argument("--validation-size", type=int, default=32, help="synthetic validation frames")
    parser.add_argument("--seed", type=int, default=None,
                        help="seed NumPy's global RNG before the models are built (initialisers draw from it, like Chainer's) "
                             "and the augmentation streams of the datasets")
    parser.add_argument("--data-seed", type=int, default=10, help="seed of the synthetic datasets")
    parser.add_argument("--no-shuffle", action='store_true', help="iterate the datasets in order (reproducible trajectories)")
    parser.add_argument("--loader-threads", type=int, default=0, help="decode threads per iterator (0: host cores / ranks, at most 16)")
    parser.add_argument("--loader-processes", type=int, default=-1,
                        help="decode PROCESSES of the training-frame iterator (Pillow's decoders hold the GIL: threads do not "
                             "scale); -1: as many as --loader-threads resolves to, 0: decode on the threads")
    parser.add_argument("--frame-cache-gb", type=float, default=16.0,
                        help="decode once: keep up to this many GB of decoded training frames (uint8 HWC) for the following epochs, "
                             "0 = decode every frame every epoch like the reference")
    parser.add_argument("--frame-cache-where", default='device', choices=['device', 'host'],
                        help="where the decoded frames live: in HBM (a batch is then a device-side gather: no decode, no staging "
                             "copy, no upload) or in host memory (only the decode is skipped)")
    parser.add_argument("--host-input", action='store_true',
                        help="finish the frames on the host (Pillow resize in get_example) instead of on the GPU")
    parser.add_argument("--flat-log-dir", action='store_true', help="write into --log-dir itself (no <time>_<name> sub-directory)")
    parser.add_argument("--record-history", action='store_true',
                        help="keep losses and theta of EVERY iteration (a host sync per iteration: parity tests only)")
    parser.add_argument("--use-graph", action='store_true', default=False,
                        help="capture the step into a hipGraph after two eager iterations.  HOST OFFLOAD ONLY: a replay costs "
                             "1.6 - 4.8 us per node on this stack (profiles/r5_b16_graph_nodes.txt), so the captured step is 8 - 15 %% "
                             "SLOWER than the eager one at batch 16 (6.4 vs 5.7 ms); what it saves is the host: 0.13 ms instead of "
                             "4.4 ms of enqueue per step")
    parser.add_argument("--dtype", default='f32', choices=['f32', 'bf16'],
                        help="f32: the parity path; bf16: bf16 activations / gradients in HBM and bf16 MFMA, fp32 master "
                             "weights and gradient accumulation (not in the reference, which is fp32 only)")
    return parser.parse_args(argv)


def build_models(args):
    """localizer + assessor exactly as the loop uses them (reference :118-127).  Host-side only (no GPU needed): the
    fixture generator builds the very same initial weights from the same seed.  ``Linear(None, 1)`` of the assessor is
    resolved here from the target size, which is what its first Chainer call would do."""
    if args.seed is not None:
        np.random.seed(args.seed)
    localizer_class = loans_amd.SheepLocalizer if args.use_resnet_18 else loans_amd.Resnet50SheepLocalizer
    localizer = localizer_class(tuple(args.target_size))
    if args.resume_localizer is not None:
        load_pretrained_model(args.resume_localizer, localizer)
    discriminator = loans_amd.ResnetAssessor(output_dim=1)
    h, w = args.target_size
    for _ in range(2):
        h, w = loans_amd.ops.conv_outsize(h, 4, 2, 1), loans_amd.ops.conv_outsize(w, 4, 2, 1)
    discriminator.l4.ensure_initialized(128 * h * w, nhwc_input=(h, w, 128))
    if args.resume_discriminator is not None:
        load_pretrained_model(args.resume_discriminator, discriminator)
    return localizer, discriminator


def build_datasets(args, rank=0):
    """(train, reference, validation or None) -- reference :77-111, keyword for keyword."""
    from loans_amd.common.datasets.image_dataset import ImageDataset, LabeledImageDataset
    if args.train_file == SYNTHETIC:
        train = SyntheticFrames(args.dataset_size, args.image_size, seed=args.data_seed + rank)
    else:
        train_image_paths = load_train_paths(args.train_file) if args.train_file.endswith('.json') else args.train_file
        train = ImageDataset(
            train_image_paths,
            os.path.dirname(args.train_file),
            image_size=tuple(args.image_size),
            dtype=np.float32,
            use_imgaug=args.use_imgaug,
            transform_probability=0.5,
            augment_seed=None if args.seed is None else args.seed + 7919 * rank,
            frame_cache_gb=0 if args.host_input else args.frame_cache_gb,
            frame_cache_where=args.frame_cache_where,
        )
    if args.reference_file == SYNTHETIC:
        reference = SyntheticAssessorSamples(args.dataset_size, args.target_size, seed=args.data_seed + 990 + rank)
    else:
        reference = LabeledImageDataset(
            args.reference_file,
            os.path.dirname(args.reference_file),
            image_size=tuple(args.target_size),
            dtype=np.float32,
            label_dtype=np.float32,
        )
    validation = None
    if args.validation:
        if args.val_file == SYNTHETIC:
            validation = SyntheticValidationFrames(args.validation_size, args.image_size, seed=args.data_seed + 5000)
        else:
            validation_data = load_train_paths(args.val_file, with_label=True) if args.val_file.endswith('.json') else args.val_file
            # reference :111 -- default label dtype (int32), default dtype
            validation = LabeledImageDataset(validation_data, os.path.dirname(args.val_file), image_size=tuple(args.image_size))
    return train, reference, validation


def loader_threads(args, world):
    if args.loader_threads > 0:
        return args.loader_threads
    return max(1, min(16, len(os.sched_getaffinity(0)) // max(world, 1)))      # 16 decode processes feed 6.9 k frames/s


def run(args, log=print):
    """The training loop.  Returns (history, localizer, discriminator): the entries of the log intervals (every iteration with
    ``--record-history``: losses, theta of the batch -- what tests/test_gpu_trainer.py compares with the oracle's trajectory)."""
    comm = parallel.init_from_env()
    if comm.size > 1:
        args.gpu = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    elif args.gpu < 0:
        args.gpu = torch.cuda.current_device()
    torch.cuda.set_device(args.gpu)

    train_dataset, reference_dataset, validation_dataset = build_datasets(args, comm.rank)
    shuffle = not args.no_shuffle
    threads = loader_threads(args, comm.size)
    # while a hipGraph is being captured no other thread may allocate device memory: the GPU stages of the feed then run on
    # the training thread (decode still runs ahead on the pool)
    feed = dict(n_threads=threads, device=None if args.host_input else args.gpu,
                device_stage='consumer' if args.use_graph else 'producer')
    procs = threads if args.loader_processes < 0 else args.loader_processes
    data_iter = training.MultithreadIterator(train_dataset, args.batch_size, shuffle=shuffle, n_processes=procs, **feed)
    reference_iter = training.MultithreadIterator(reference_dataset, args.batch_size, shuffle=shuffle, **feed)

    localizer, discriminator = build_models(args)
    if args.dtype == 'bf16':
        localizer.set_precision('bf16', 'bf16')
        discriminator.set_precision('bf16', 'bf16')
    localizer.to_gpu(args.gpu)
    discriminator.to_gpu(args.gpu)
    comm.bcast_data(localizer)
    comm.bcast_data(discriminator)

    localizer_optimizer = loans_amd.Adam(alpha=args.learning_rate, amsgrad=True)
    localizer_optimizer.setup(localizer)
    discriminator_optimizer = loans_amd.Adam(alpha=args.learning_rate, amsgrad=True)
    discriminator_optimizer.setup(discriminator)
    parallel.create_multi_node_optimizer(localizer_optimizer, comm)
    parallel.create_multi_node_optimizer(discriminator_optimizer, comm)

    updater = loans_amd.SheepAssessor(
        models=[localizer, discriminator],
        iterator={'main': data_iter, 'real': reference_iter},
        optimizer={'opt_gen': localizer_optimizer, 'opt_dis': discriminator_optimizer},
        device=args.gpu, comm=comm, create_pca=False,
        resume_discriminator=args.resume_discriminator, localizer_target=args.localizer_target,
        use_graph=args.use_graph)

    evaluator = None
    if validation_dataset is not None:
        validation_iter = training.MultithreadIterator(validation_dataset, args.batch_size, repeat=False, shuffle=False,
                                                       n_threads=threads)
        sheep_evaluator = loans_amd.SheepMAPEvaluator(localizer, args.gpu)
        evaluator = training.Evaluator(validation_iter, localizer, device=args.gpu, eval_func=sheep_evaluator)

    # reference :158-162: <log-dir>/<iso time>_<log name>
    if not args.flat_log_dir:
        stamp = datetime.datetime.now().isoformat() if comm.size == 1 else os.environ.get('TORCHELASTIC_RUN_ID', 'dp') + \
            '_' + datetime.datetime.now().strftime('%Y-%m-%dT%H')
        args.log_dir = os.path.join(args.log_dir, "{}_{}".format(stamp, args.ln))
    os.makedirs(args.log_dir, exist_ok=True)
    # reference :166-180: the first log entry carries the configuration (evaluate.py and the inference tools read it back)
    data_to_log = {'log_dir': args.log_dir, 'image_size': list(args.image_size),
                   'updater': [updater.__class__.__name__, 'updater.py'],
                   'discriminator': [discriminator.__class__.__name__, 'discriminator.py'], 'discriminator_output_dim': 1,
                   'localizer': [localizer.__class__.__name__, 'localizer.py']}
    for argument in filter(lambda x: not x.startswith('_'), dir(args)):
        data_to_log[argument] = getattr(args, argument)
    log_entries = []

    def snapshot(it):
        for model in (localizer, discriminator):
            loans_amd.save_npz(os.path.join(args.log_dir, '%s_%d.npz' % (model.__class__.__name__, it)), model)

    def theta_of_batch():
        return localizer.last_transform_params.data.reshape(-1, 6).cpu().numpy().copy()

    history = []
    t0 = time.time()
    it = 0
    while updater.epoch < args.num_epoch and (args.iterations is None or it < args.iterations):
        updater.update()
        it = updater.iteration
        last = (args.iterations is not None and it == args.iterations) or updater.epoch >= args.num_epoch
        # log train information every time a new epoch starts or log_interval iterations have been done (reference :188-190)
        log_now = updater.is_new_epoch or it % args.log_interval == 0 or last
        entry = None
        if log_now or args.record_history:
            # the only host synchronisation of an iteration; between log intervals the launch queue runs ahead freely
            obs = loans_amd.reporter.observation
            entry = {'iteration': it, 'epoch': updater.epoch, 'loss_localizer': float(obs['loss_localizer']),
                     'loss_dis': float(obs['loss_dis'])}
            if args.record_history:
                entry['theta'] = theta_of_batch()
        if log_now:
            if evaluator is not None:
                # the evaluator's test-mode forward rebinds the localizer's last_transform_params (and, under a captured
                # graph, the training step's own tensor must stay the one the replays write): restored afterwards
                keep = localizer.last_transform_params
                entry['validation'] = evaluator()
                localizer.last_transform_params = keep
            if comm.rank == 0:
                val = ''
                if 'validation' in entry:
                    val = '  mean_iou %.4f  map %.4f' % (entry['validation']['mean_iou'], entry['validation']['map'])
                log('iteration %4d  epoch %d  loss_localizer %.5f  loss_dis %.5f%s  (%.1f s)' % (
                    it, updater.epoch, entry['loss_localizer'], entry['loss_dis'], val, time.time() - t0))
                stats = {k: v for k, v in entry.items() if k not in ('theta', 'validation')}
                stats.update(entry.get('validation', {}))
                stats['elapsed_time'] = time.time() - t0
                if it == args.log_interval or not log_entries:
                    stats.update(data_to_log)
                log_entries.append(stats)
                with open(os.path.join(args.log_dir, 'log'), 'w') as f:
                    json.dump(log_entries, f, indent=4, default=str)
        if entry is not None:
            history.append(entry)
        # reference :182-186: a snapshot on every new epoch, or every snapshot_interval iterations
        if comm.rank == 0 and (updater.is_new_epoch if args.snapshot_every_epoch else it % args.snapshot_interval == 0):
            snapshot(it)
    for iterator in (data_iter, reference_iter):
        iterator.finalize()
    if comm.rank == 0:
        if not os.path.exists(os.path.join(args.log_dir, '%s_%d.npz' % (localizer.__class__.__name__, updater.iteration))):
            snapshot(updater.iteration)
        frame0 = train_dataset.get_example(0)
        bboxes, rois, scores, _ = localizer.predict([frame0[0] if isinstance(frame0, tuple) else frame0])
        log('predict() on frame 0: bbox (top,left,bottom,right) = %s' % np.round(bboxes[0], 2).tolist())
    return history, localizer, discriminator


def main(argv=None):
    run(parse_args(argv))
    parallel.shutdown()


if __name__ == "__main__":
    main()
