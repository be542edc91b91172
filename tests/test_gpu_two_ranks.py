"""Data-parallel training step on two ranks without a second GPU: two processes (gloo) share cuda:0 and run
tests/two_rank_worker.py, which asserts bit-identical parameters on both ranks equal to Adam on the mean of the
two local gradients."""
import os
import signal
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_train_step_equals_adam_on_mean_gradient():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29611', os.path.join(ROOT, 'tests', 'two_rank_worker.py')]
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                            start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=100)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)            # the launcher and both ranks
        out, _ = proc.communicate()
        raise AssertionError('two-rank step did not finish within 100 s:\n' + out[-3000:])
    assert proc.returncode == 0, out[-3000:]
    assert out.count(': ok, parameters moved') == 2, out[-3000:]
