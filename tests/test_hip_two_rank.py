"""Two data-parallel ranks on the test box's single GPU: fresh child
processes (torch.distributed.run, started before this process hands them any GPU state)
run tests/dp_two_rank_worker.py; see its docstring for what is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_on_one_gpu_match_each_other_and_the_single_process_step(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(HERE, 'dp_two_rank_worker.py'), str(tmp_path)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    with open(tmp_path / 'result.json') as f:
        r = json.load(f)
    assert r['world'] == 2
    assert r['identical_across_ranks'], r
    # same global batches, same dropout masks (keyed by global path id), fp32 summation order
    # differs between 1 and 2 shards: 1e-5 relative
    assert r['rel_vs_single'] < 1e-5, r
    for a, b in zip(r['losses_dp'], r['losses_single']):
        assert a == pytest.approx(b, rel=1e-5)
