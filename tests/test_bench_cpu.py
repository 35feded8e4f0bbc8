"""bench.py's stdout contract on the CPU: the ONE line the driver parses is compact (round 5's line carried three per-layer tables,
grew to 26.5 KB and `BENCH_r05.parsed` came out null), the full object goes to --detail-file, and the per-GPU batch default follows
BASELINE.json (configs[1] = 256 on one GPU, configs[3] = 128 per GPU on several)."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def bench():
    spec = importlib.util.spec_from_file_location('_bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)            # __name__ != '__main__': nothing is launched
    return mod


def _full_size_payload():
    """the full result object of a default N = 1 run: round 5's committed 26.5 KB line, plus the configs[3] leg added since"""
    out = json.load(open(os.path.join(ROOT, 'profiles', 'r5_default_bench.json')))
    leg = json.loads(json.dumps(out['secondary']['configs[2]']))
    leg['config']['baseline_config'] = 'configs[3] per GPU (128 of the global 1024)'
    out['secondary']['configs[3] per GPU'] = leg
    return out


def test_stdout_line_is_compact_and_keeps_the_contract(bench):
    out = _full_size_payload()
    assert len(json.dumps(out)) > 20000                                     # the payload really is the one that broke the driver
    text = json.dumps(bench.compact_line(out, os.path.join(ROOT, 'bench_detail.json')))
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in line, k
        if k not in ('config', 'roofline'):
            assert line[k] == out[k]
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and r['traffic'] == out['roofline']['traffic']
    assert r['whole_step']['binding_frac'] == out['roofline']['whole_step']['binding']['binding_frac']
    assert 'binding' not in r and 'layers' not in text
    assert line['cpu_baseline']['value'] > 0 and line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['cores'] >= 1
    assert line['config']['baseline_config'] == 'configs[1]' and line['config']['workload'].startswith('LoANs joint step')
    assert 'model' not in line['config']
    assert set(line['secondary']) == set(out['secondary'])
    for label, leg in line['secondary'].items():
        full = out['secondary'][label]
        assert leg['value'] == full['value'] and leg['ms_per_step'] == full['ms_per_step'] and leg['dtype'] == full['dtype']
        assert leg['baseline_config'] == full['config']['baseline_config']
        assert leg['roofline_frac'] == (full['roofline'] or {}).get('frac')
    assert line['secondary']['reference default (-b 16, 224 x 224), hipGraph']['graph_over_eager'] > 0
    assert line['detail'] == 'bench_detail.json'


def test_compact_line_of_a_leg_without_roofline_or_secondary(bench):
    out = {"metric": "m", "value": 1.0, "unit": "images/s", "n_gpus": 2, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "w", "allocator_in_timed_region": {"x": 1}}, "roofline": None}
    line = bench.compact_line(out)
    assert line['roofline'] is None and 'secondary' not in line and 'cpu_baseline' not in line and 'detail' not in line
    assert line['config'] == {"workload": "w"}


def test_per_gpu_batch_default_follows_baseline_configs(bench, monkeypatch):
    def parsed(*argv):
        monkeypatch.setattr(sys, 'argv', ['bench.py'] + list(argv))
        return bench.parse()
    one, eight, forced = parsed(), parsed('--gpus', '8'), parsed('--gpus', '8', '--batch', '256')
    w1, w8, wf = bench.workload_of(one), bench.workload_of(eight), bench.workload_of(forced)
    assert (w1.batch, w8.batch, wf.batch) == (256, 128, 256)
    assert bench.config_label(w1, 1) == 'configs[1]' and bench.config_label(w8, 8) == 'configs[3]'
    assert bench.config_label(wf, 8) == 'configs[1] per GPU, data parallel'
    assert bench.config_label(bench.workload_of(one, batch=128), 1).startswith('configs[3] per GPU')
    labels = [label for label, over in bench.secondary_legs(one)]
    assert labels[:3] == ['configs[2]', 'configs[4] per GPU', 'configs[3] per GPU'] and len(labels) == 5
    # the CPU leg is bounded: a handful of B = 8 oracle steps (about 5 s each on the GPU box's host)
    assert one.cpu_batch == 8 and 2 <= one.cpu_iters <= 5
