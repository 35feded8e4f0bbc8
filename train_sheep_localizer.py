#!/usr/bin/env python
"""The reference's training driver (train_sheep_localizer.py:50-255) for the MI355X build: same wiring
(datasets -> iterators -> SheepLocalizer / ResnetAssessor -> two Adam(amsgrad=True) -> SheepAssessor ->
loop), same flags where they touch the hot path.  Datasets are the seeded synthetic paste-and-crop
generator (the PIL / imgaug datasets are SURVEY §8f.2); the Trainer extensions that are not on the path
(BBOXPlotter, Logger, interactive prompt, dump_graph) are intentionally absent.

    python train_sheep_localizer.py --use-resnet-18 -b 64 --iterations 20
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


def load_pretrained_model(model_file, model):
    loans_amd.load_npz(model_file, model, strict=False)       # NpzDeserializer(strict=False), reference :45-47


def main():
    parser = argparse.ArgumentParser(description="Train a sheep localizer (MI355X-native LoANs hot path)")
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
    parser.add_argument("--use-graph", action='store_true', default=False,
                        help="capture the step into a hipGraph after two eager iterations (launch-bound small batches)")
    parser.add_argument("--dtype", default='f32', choices=['f32', 'bf16'],
                        help="f32: the parity path; bf16: bf16 activations / gradients in HBM and bf16 MFMA, fp32 master "
                             "weights and gradient accumulation (not in the reference, which is fp32 only)")
    args = parser.parse_args()
    if args.dtype == 'bf16':
        loans_amd.set_compute_dtype('bf16')
        loans_amd.set_storage_dtype('bf16')

    comm = parallel.init_from_env()
    if comm.size > 1:
        args.gpu = int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count()
    torch.cuda.set_device(args.gpu)

    train_dataset = SyntheticFrames(args.dataset_size, args.image_size, seed=10 + comm.rank)
    reference_dataset = SyntheticAssessorSamples(args.dataset_size, args.target_size, seed=1000 + comm.rank)
    data_iter = training.MultithreadIterator(train_dataset, args.batch_size)
    reference_iter = training.MultithreadIterator(reference_dataset, args.batch_size)

    localizer_class = loans_amd.SheepLocalizer if args.use_resnet_18 else loans_amd.Resnet50SheepLocalizer
    localizer = localizer_class(args.target_size)
    if args.resume_localizer is not None:
        load_pretrained_model(args.resume_localizer, localizer)
    discriminator = loans_amd.ResnetAssessor(output_dim=1)
    with loans_amd.using_config('enable_backprop', False):      # resolves Linear(None, 1) like the first Chainer call
        discriminator(torch.zeros((1, 3) + tuple(args.target_size), device='cuda'))
    if args.resume_discriminator is not None:
        load_pretrained_model(args.resume_discriminator, discriminator)
    localizer.to_gpu(args.gpu)
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

    os.makedirs(args.log_dir, exist_ok=True)
    t0 = time.time()
    for _ in range(args.iterations):
        updater.update()
        it = updater.iteration
        if comm.rank == 0 and (it % args.log_interval == 0 or it == args.iterations):
            obs = loans_amd.reporter.observation
            print('iteration %4d  epoch %d  loss_localizer %.5f  loss_dis %.5f  (%.1f s)' % (
                it, updater.epoch, float(obs['loss_localizer']), float(obs['loss_dis']), time.time() - t0), flush=True)
        if comm.rank == 0 and args.snapshot_interval and it % args.snapshot_interval == 0:
            for model in (localizer, discriminator):
                loans_amd.save_npz(os.path.join(args.log_dir, '%s_%d.npz' % (model.__class__.__name__, it)), model)
    if comm.rank == 0:
        for model in (localizer, discriminator):
            loans_amd.save_npz(os.path.join(args.log_dir, '%s_%d.npz' % (model.__class__.__name__, updater.iteration)), model)
        bboxes, rois, scores, _ = localizer.predict([train_dataset[0]])
        print('predict() on frame 0: bbox (top,left,bottom,right) =', np.round(bboxes[0], 2).tolist())


if __name__ == "__main__":
    main()
