#!/usr/bin/env python
"""The reference's training driver (train_sheep_localizer.py:50-255) for the MI355X build: same wiring
(datasets -> iterators -> SheepLocalizer / ResnetAssessor -> two Adam(amsgrad=True) -> SheepAssessor ->
loop -> snapshots), same flags where they touch the hot path, and the validation loop of :106-113,192-197
(``--validation``: a validation iterator run through ``SheepMAPEvaluator`` at every log interval).

Datasets: the seeded synthetic paste-and-crop generator by default; ``--train-file`` / ``--reference-file`` /
``--val-file`` read the reference's file formats (``ImageDataset`` path list, ``LabeledImageDataset`` tab-separated
csv: common/datasets/image_dataset.py).  The Trainer extensions that are not on the path (BBOXPlotter, Logger,
interactive prompt, dump_graph) are intentionally absent.

    python train_sheep_localizer.py --use-resnet-18 -b 64 --iterations 20 --validation
    python train_sheep_localizer.py --gpus 8 --use-resnet-18 -b 128        (forks its own 8 ranks)
"""
import argparse
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


class SyntheticFrames:
    """`ImageDataset` stand-in: float32 CHW RGB frames in [0,1] (common/datasets/image_dataset.py:47-98)."""

    def __init__(self, n, image_size, seed=0):
        self.frames = synthetic.make_frames(seed, n, image_size[0], image_size[1])

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i]


class SyntheticAssessorSamples:
    """`LabeledImageDataset` stand-in: (crop, IoU label, zeros(1)) triples (image_dataset.py:180-181)."""

    def __init__(self, n, target_size, seed=1):
        self.x, self.y = synthetic.make_assessor_batch(seed, n, target_size[0], target_size[1])

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i], np.zeros(1, np.float32)


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


def load_pretrained_model(model_file, model):
    loans_amd.load_npz(model_file, model, strict=False)       # NpzDeserializer(strict=False), reference :45-47


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Train a sheep localizer (MI355X-native LoANs hot path)")
    parser.add_argument("--train-file", help="path list of training frames (ImageDataset); default: synthetic frames")
    parser.add_argument("--reference-file", help="tab-separated <crop>\\t<iou> csv (LabeledImageDataset, the output of "
                                                 "datasets/sheep/paste_and_crop_sheep.py); default: synthetic crops")
    parser.add_argument("--val-file", help="tab-separated <frame>\\t<top>\\t<left>\\t<bottom>\\t<right> csv for --validation")
    parser.add_argument("--image-size", type=int, nargs=2, default=(224, 224), help="input size for localizer")
    parser.add_argument("--target-size", type=int, nargs=2, default=(75, 75), help="crop size for each image")
    parser.add_argument("-b", "--batch-size", type=int, default=16, help="batch size for training (per GPU)")
    parser.add_argument("-g", "--gpu", type=int, default=0, help="gpu id to use")
    parser.add_argument("--gpus", type=int, default=1,
                        help="data-parallel ranks on this node, one process per GPU, forked by this script itself")
    parser.add_argument("--lr", "--learning-rate", dest="learning_rate", type=float, default=0.001)
    parser.add_argument("-l", "--log-dir", default='sheep_logs')
    parser.add_argument("--iterations", type=int, default=20)
    parser.add_argument("--dataset-size", type=int, default=256)
    parser.add_argument("--snapshot-interval", type=int, default=0, help="0 = only at the end")
    parser.add_argument("--log-interval", type=int, default=5)
    parser.add_argument("--rl", dest="resume_localizer")
    parser.add_argument("--rd", dest="resume_discriminator")
    parser.add_argument("--use-resnet-18", action='store_true', default=False)
    parser.add_argument("--localizer-target", type=float, default=1.0)
    parser.add_argument("--no-validation", dest='validation', action='store_false', default=False)
    parser.add_argument("--validation", dest='validation', action='store_true',
                        help="run SheepMAPEvaluator over a validation set at every log interval (reference :106-113,192-197)")
    parser.add_argument("--validation-size", type=int, default=32, help="synthetic validation frames (without --val-file)")
    parser.add_argument("--seed", type=int, default=None,
                        help="seed NumPy's global RNG before the models are built (initialisers draw from it, like Chainer's)")
    parser.add_argument("--data-seed", type=int, default=10, help="seed of the synthetic datasets")
    parser.add_argument("--no-shuffle", action='store_true', help="iterate the datasets in order (reproducible trajectories)")
    parser.add_argument("--use-graph", action='store_true', default=False,
                        help="capture the step into a hipGraph after two eager iterations (launch-bound small batches)")
    parser.add_argument("--dtype", default='f32', choices=['f32', 'bf16'],
                        help="f32: the parity path; bf16: bf16 activations / gradients in HBM and bf16 MFMA, fp32 master "
                             "weights and gradient accumulation (not in the reference, which is fp32 only)")
    return parser.parse_args(argv)


def build_models(args):
    """localizer + assessor exactly as the loop uses them (reference :115-127).  Host-side only (no GPU needed): the
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
    """(train, reference, validation or None) -- reference :77-111."""
    from loans_amd.common.datasets.image_dataset import ImageDataset, LabeledImageDataset
    if args.train_file:
        train = ImageDataset(args.train_file, os.path.dirname(args.train_file), image_size=tuple(args.image_size))
    else:
        train = SyntheticFrames(args.dataset_size, args.image_size, seed=args.data_seed + rank)
    if args.reference_file:
        reference = LabeledImageDataset(args.reference_file, os.path.dirname(args.reference_file),
                                        image_size=tuple(args.target_size), dtype=np.float32, label_dtype=np.float32)
    else:
        reference = SyntheticAssessorSamples(args.dataset_size, args.target_size, seed=args.data_seed + 990 + rank)
    validation = None
    if args.validation:
        if args.val_file:
            validation = LabeledImageDataset(args.val_file, os.path.dirname(args.val_file), image_size=tuple(args.image_size),
                                             dtype=np.float32, label_dtype=np.float32)
        else:
            validation = SyntheticValidationFrames(args.validation_size, args.image_size, seed=args.data_seed + 5000)
    return train, reference, validation


def run(args, log=print):
    """The training loop.  Returns the per-iteration history (losses, theta of the batch; validation metrics at the log
    intervals) -- what tests/test_gpu_trainer.py compares with the oracle's trajectory."""
    if args.dtype == 'bf16':
        loans_amd.set_compute_dtype('bf16')
        loans_amd.set_storage_dtype('bf16')

    comm = parallel.init_from_env()
    if comm.size > 1:
        args.gpu = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(args.gpu)

    train_dataset, reference_dataset, validation_dataset = build_datasets(args, comm.rank)
    shuffle = not args.no_shuffle
    data_iter = training.MultithreadIterator(train_dataset, args.batch_size, shuffle=shuffle)
    reference_iter = training.MultithreadIterator(reference_dataset, args.batch_size, shuffle=shuffle)

    localizer, discriminator = build_models(args)
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
        validation_iter = training.MultithreadIterator(validation_dataset, args.batch_size, repeat=False, shuffle=False)
        sheep_evaluator = loans_amd.SheepMAPEvaluator(localizer, args.gpu)
        evaluator = training.Evaluator(validation_iter, localizer, device=args.gpu, eval_func=sheep_evaluator)

    os.makedirs(args.log_dir, exist_ok=True)
    history = []
    t0 = time.time()
    for _ in range(args.iterations):
        updater.update()
        it = updater.iteration
        obs = loans_amd.reporter.observation
        entry = {'iteration': it, 'loss_localizer': float(obs['loss_localizer']), 'loss_dis': float(obs['loss_dis']),
                 'theta': localizer.last_transform_params.data.reshape(-1, 6).cpu().numpy().copy()}
        # log train information every time a new epoch starts or log_interval iterations have been done (reference :188-190)
        if updater.is_new_epoch or it % args.log_interval == 0 or it == args.iterations:
            if evaluator is not None:
                entry['validation'] = evaluator()
            if comm.rank == 0:
                val = ''
                if 'validation' in entry:
                    val = '  mean_iou %.4f  map %.4f' % (entry['validation']['mean_iou'], entry['validation']['map'])
                log('iteration %4d  epoch %d  loss_localizer %.5f  loss_dis %.5f%s  (%.1f s)' % (
                    it, updater.epoch, entry['loss_localizer'], entry['loss_dis'], val, time.time() - t0))
        history.append(entry)
        if comm.rank == 0 and args.snapshot_interval and it % args.snapshot_interval == 0:
            for model in (localizer, discriminator):
                loans_amd.save_npz(os.path.join(args.log_dir, '%s_%d.npz' % (model.__class__.__name__, it)), model)
    if comm.rank == 0:
        for model in (localizer, discriminator):
            loans_amd.save_npz(os.path.join(args.log_dir, '%s_%d.npz' % (model.__class__.__name__, updater.iteration)), model)
        frame0 = train_dataset[0]
        bboxes, rois, scores, _ = localizer.predict([frame0[0] if isinstance(frame0, tuple) else frame0])
        log('predict() on frame 0: bbox (top,left,bottom,right) = %s' % np.round(bboxes[0], 2).tolist())
    return history, localizer, discriminator


def main(argv=None):
    run(parse_args(argv))
    parallel.shutdown()


if __name__ == "__main__":
    main()
