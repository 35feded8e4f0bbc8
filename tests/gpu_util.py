"""Helpers shared by the -m gpu parity tests: build the HIP-backed models and the CPU
oracle from the same weights and the same seeded synthetic inputs."""
import numpy as np
import torch

import loans_amd
from loans_amd.datasets import synthetic
from oracle import model as M


def randomize_bn_and_predictor(localizer, rng, w_std=2e-2):
    """Non-degenerate state for parity runs (SURVEY §7 'hard parts'): param_predictor.W != 0,
    BN affine parameters away from (1, 0), conv1 bias non-zero."""
    for key, p in localizer.namedparams():
        if key.endswith('/gamma'):
            p.set_logical((1 + 0.1 * rng.standard_normal(p.logical_shape)).astype(np.float32))
        elif key.endswith('/beta'):
            p.set_logical((0.1 * rng.standard_normal(p.logical_shape)).astype(np.float32))
        elif key == '/feature_extractor/conv1/b':
            p.set_logical((0.1 * rng.standard_normal(p.logical_shape)).astype(np.float32))
        elif key == '/param_predictor/W':
            p.set_logical((w_std * rng.standard_normal(p.logical_shape)).astype(np.float32))


def build_pair(seed, crop, randomize=True):
    np.random.seed(seed)
    loc = loans_amd.SheepLocalizer(crop)
    dis = loans_amd.ResnetAssessor()
    if randomize:
        randomize_bn_and_predictor(loc, np.random.RandomState(seed + 100))
    return loc, dis


def oracle_params(link, dtype=np.float64):
    return M.cast_params(link.state_dict_chainer(), dtype)


def inputs(seed, B, H, W, crop):
    frames = synthetic.make_frames(seed, B, H, W)
    real, labels = synthetic.make_assessor_batch(seed + 1, B, crop[0], crop[1], src=max(64, min(H, 224)))
    return frames, real, labels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def rel_err(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-30))
