"""BASELINE configs[1] at its OWN size -- 256 x 3 x 224 x 224 frames, crop 75 x 75, ResNet-18 localizer + assessor, fp32 --
as a seeded case shared by the fixture generator (tests/golden/make_fullsize_golden.py, CPU oracle) and the GPU test
(tests/test_gpu_fullsize.py::test_configs1_against_oracle_fixture): models, weights and batches are rebuilt from seeds on
both sides, only the oracle's outputs are stored.  Semantics at stake at this size: train-mode BatchNormalization over
802 816 positions per channel (sheep/resnet.py:129-134) and the joint step of sheep/sheep_updater.py:26-68."""
import numpy as np

B, HW, CROP = 256, 224, (75, 75)
SEED = 2562
FIXTURE = 'configs1_b256_224.npz'
# BASELINE configs[3] is the same graph on ONE rank's shard -- 128 frames per GPU (global 1024 at N = 8, local BatchNormalization
# statistics): the same seeded models on the first 128 frames / crops (make_fullsize_golden.py --batch 128)
SHARD_B = 128
SHARD_FIXTURE = 'configs3_b128_224.npz'
# BN layers whose batch statistics (read back through the running averages of one step) the fixture pins
BN_KEYS = ('feature_extractor/bn1', 'feature_extractor/res3/0/bn3', 'feature_extractor/res5/1/bn2')


def build_models():
    """the HIP-backed links with seeded, non-degenerate weights (host-side only: no GPU needed)"""
    import loans_amd
    from loans_amd import ops
    from tests.gpu_util import randomize_bn_and_predictor
    np.random.seed(SEED)
    loc = loans_amd.SheepLocalizer(CROP)
    dis = loans_amd.ResnetAssessor()
    h, w = CROP
    for _ in range(2):
        h, w = ops.conv_outsize(h, 4, 2, 1), ops.conv_outsize(w, 4, 2, 1)
    dis.l4.ensure_initialized(128 * h * w, nhwc_input=(h, w, 128))
    randomize_bn_and_predictor(loc, np.random.RandomState(SEED + 100))
    return loc, dis


def build_inputs(batch=B):
    from loans_amd.datasets import synthetic
    frames = synthetic.make_frames(SEED + 1, batch, HW, HW)
    real, labels = synthetic.make_assessor_batch(SEED + 2, batch, CROP[0], CROP[1], src=HW)
    return frames, real, labels
