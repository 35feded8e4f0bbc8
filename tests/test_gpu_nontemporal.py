"""The cache policy of an output store changes nothing but where the bytes travel: every kernel that picks one (round 3:
LOANS_CONV_NT_MB for the convolution epilogues and the stem, LOANS_BN_NT for the BN apply passes; the product decides by tensor
size, which test-sized tensors never reach) is run in a child process whose environment FORCES the non-temporal stores and
loads, and its outputs are compared bit for bit with this process's default-policy ones."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_nontemporal_stores_change_no_byte(tmp_path):
    from tests import _nt_outputs
    assert 'LOANS_CONV_NT_MB' not in os.environ and 'LOANS_BN_NT' not in os.environ
    ours = _nt_outputs.outputs()
    path = str(tmp_path / 'nt.npz')
    env = dict(os.environ, LOANS_CONV_NT_MB='0', LOANS_BN_NT='1', PYTHONPATH=ROOT)
    subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_nt_outputs.py'), path], check=True, env=env, cwd=ROOT, timeout=600)
    theirs = np.load(path)
    assert sorted(theirs.files) == sorted(ours) and len(ours) > 20
    for key, val in ours.items():
        np.testing.assert_array_equal(theirs[key], val, err_msg=key)
