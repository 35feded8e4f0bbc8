"""-m gpu: `python bench.py --gpus 2` through its own launcher with the REAL step (two ranks of the HIP joint step on
GPU 0, gradients exchanged through gloo: RCCL wants one device per rank, everything else is what the driver's 8-GPU run
executes), and `--gpus 1` as a single process.  The parent process never touches the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ['--batch', '4', '--image-size', '64', '--target-size', '16', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']


def _run(script, args, **env):
    e = dict(os.environ, **env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LOANS_SPLITK'):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, cwd=ROOT, env=e, timeout=900,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _one_line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    return lines[0]


def test_bench_two_ranks_through_the_launcher():
    out = _one_line(_run('bench.py', ['--gpus', '2'] + TINY, LOANS_DIST_BACKEND='gloo'))
    assert out['n_gpus'] == 2 and out['config']['world_size'] == 2 and out['config']['dist_backend'] == 'gloo'
    assert out['config']['global_batch'] == 8 and out['scaling'] == 'weak'
    assert out['value'] > 0 and abs(out['value'] - 8 * 1e3 / out['ms_per_step']) < 1e-2 * out['value']
    assert out['roofline']['achieved'] > 0


def test_bench_single_process():
    out = _one_line(_run('bench.py', ['--gpus', '1'] + TINY))
    assert out['n_gpus'] == 1 and out['config']['dist_backend'] is None and out['value'] > 0


def test_bench_rccl_at_world_size_one():
    """LOANS_DIST_SELFTEST=1: every collective of the data-parallel step is issued even at world size 1 -- the RCCL (nccl backend)
    path on a one-GPU box: process group, parameter broadcast, the staged all-reduce of both gradient arenas"""
    e = dict(os.environ, LOANS_DIST_SELFTEST='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29731')
    e.pop('LOANS_SPLITK', None)
    e.pop('LOANS_DIST_BACKEND', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + TINY, cwd=ROOT, env=e, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out = _one_line(r)
    assert out['config']['dist_backend'] == 'nccl' and out['config']['world_size'] == 1 and out['value'] > 0


def test_bench_rccl_at_world_size_one_captured_step():
    """the same with --graph: the data-parallel step captured as hipGraph SEGMENTS with the RCCL exchanges issued between them
    (sheep_updater.py: localizer chain | all-reduce + Adam | assessor chain | all-reduce + Adam) -- until round 6 only the eager
    step had met RCCL"""
    e = dict(os.environ, LOANS_DIST_SELFTEST='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29733')
    e.pop('LOANS_SPLITK', None)
    e.pop('LOANS_DIST_BACKEND', None)
    args = ['--gpus', '1', '--graph'] + [a if a != '2' else '4' for a in TINY]          # 4 timed steps: capture happens in warm-up
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=e, timeout=900,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out = _one_line(r)
    assert out['config']['dist_backend'] == 'nccl' and out['config']['hip_graph'] is True and out['value'] > 0
    eager = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + TINY, cwd=ROOT, env=e, timeout=900,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert _one_line(eager)['config']['hip_graph'] is False


def test_bench_secondary_legs_schema(tmp_path):
    """The driver's one bench line carries the other single-GPU configurations as `secondary` legs (configs[2]: bf16 joint
    step; configs[4] per GPU: ResNet-50 localizer, bf16; configs[3] per GPU: fp32, 128 frames) -- five numbers each on the
    stdout line, which stays a few KB (round 5's 26.5 KB line was not parsed by the driver), the full legs with their rooflines
    and tables in --detail-file; here on tiny shapes (2 x 3 x 320 x 320: res6 and res7 active) behind a tiny primary leg; the
    primary keys stay those of the fp32 leg."""
    detail = os.path.join(str(tmp_path), 'detail.json')
    r = _run('bench.py', ['--gpus', '1', '--secondary-shape', '2,320', '--secondary-steps', '2',
                          '--secondary-warmup', '1', '--detail-file', detail] + TINY)
    line = _one_line(r)
    assert len(r.stdout.strip().splitlines()[-1]) < 4096 and line['detail'] == detail
    assert line['dtype'] == 'f32' and line['config']['baseline_config'] == 'custom' and 0 < line['roofline']['frac'] < 1
    assert 'binding' not in line['roofline'] and set(line['roofline']['whole_step']) == {'frac', 'binding_frac', 'machine_frac'}
    out = json.load(open(detail))
    assert out['value'] == line['value'] and out['ms_per_step'] == line['ms_per_step']
    for label, leg in out['secondary'].items():
        c = line['secondary'][label]
        assert c['value'] == leg['value'] and c['ms_per_step'] == leg['ms_per_step'] and c['dtype'] == leg['dtype']
        assert c['roofline_frac'] == (leg['roofline'] or {}).get('frac')
    assert out['dtype'] == 'f32' and out['config']['activation_storage'] == 'f32' and out['config']['frame'] == '3x64x64'
    sec = out['secondary']
    b16 = {k: sec.pop(k) for k in list(sec) if k.startswith('reference default')}
    b128 = sec.pop('configs[3] per GPU')
    assert b128['dtype'] == 'f32' and b128['config']['activation_storage'] == 'f32' and b128['roofline']['peak'] == 157.3
    assert sorted(sec) == ['configs[2]', 'configs[4] per GPU']
    # the reference's default batch, eager and as a hipGraph (here shrunk like the others): fp32, the same workload twice
    eager, graph = b16['reference default (-b 16, 224 x 224), eager'], b16['reference default (-b 16, 224 x 224), hipGraph']
    assert eager['dtype'] == graph['dtype'] == 'f32' and not eager['config']['hip_graph'] and graph['config']['hip_graph']
    assert eager['value'] > 0 and graph['value'] > 0
    assert abs(graph['graph_over_eager'] - graph['value'] / eager['value']) < 1e-2 and graph['roofline'] is None
    for name, leg in sec.items():
        assert leg['dtype'] == 'bf16' and leg['config']['activation_storage'] == 'bf16' and leg['config']['frame'] == '3x320x320'
        assert leg['config']['per_gpu_batch'] == 2 and leg['steps'] == 2 and leg['unit'] == 'images/s'
        assert leg['value'] > 0 and abs(leg['value'] - 2 * 1e3 / leg['ms_per_step']) < 1e-2 * leg['value']
        r = leg['roofline']
        assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and 0 < r['frac'] < 1 and 'traffic' in r
        assert 0 < r['binding']['frac'] and 0 < r['binding']['frac_write_priced'] and r['binding']['layers']
        assert 0 < r['whole_step']['frac'] < 1 and set(r['whole_step']['by_kind']) == {'dgrad', 'fprop', 'wgrad'}
        # the whole step priced per kernel class (round 5): algorithmic bytes / FLOP counted live, bound per launch
        b = r['whole_step']['binding']
        assert {'conv', 'wgrad', 'bn_fwd', 'bn_bwd', 'stem', 'optimizer'} <= set(b['classes'])
        assert 0 < b['bound_ms_machine'] <= b['bound_ms_sum'] and 0 < b['binding_frac'] < 1
        assert all(c['bound_ms'] >= 0 and c['mbytes'] >= 0 for c in b['classes'].values())
    assert 'ResNet-50' in sec['configs[4] per GPU']['config']['workload'] and 'ResNet-18' in sec['configs[2]']['config']['workload']
    # 31 convolutions of the ResNet-18 variant at 320 px (res6 + res7), 53 + 10 of the ResNet-50 localizer
    assert '31 convs' in sec['configs[2]']['roofline']['kernel']


def test_trainer_two_ranks_through_the_launcher(tmp_path):
    r = _run('train_sheep_localizer.py', ['--gpus', '2', '--use-resnet-18', '-b', '2', '--image-size', '64', '64',
                                          '--target-size', '16', '16', '--iterations', '3', '--dataset-size', '8',
                                          '--log-interval', '1', '--no-validation', '--flat-log-dir', '-l', str(tmp_path)],
             LOANS_DIST_BACKEND='gloo')
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count('iteration') == 3, r.stdout                    # rank 0 alone reports
    assert os.path.exists(os.path.join(str(tmp_path), 'SheepLocalizer_3.npz'))
