"""The VALU kernels (NJODE_ODE=valu: segment and lockstep plans without the matrix
cores) stay covered: the library reads the switch once per process, so the parity suite is
re-run in a child process with the switch set (together with the lockstep plan's dropout
tests, whose VALU sweep keys the masks differently from the matrix-core one)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1500)
def test_parity_suite_with_valu_kernels():
    env = dict(os.environ, NJODE_ODE='valu')
    cmd = [sys.executable, '-m', 'pytest', os.path.join(REPO, 'tests', 'test_hip_parity.py'),
           os.path.join(REPO, 'tests', 'test_hip_lockstep_dropout.py'), '-m',
           'gpu', '-q', '-x', '--timeout', '600', '-p', 'no:cacheprovider']
    p = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True)
    assert p.returncode == 0, p.stdout[-4000:]
