"""-m gpu: a bounded wait inside a launch that really times out on the device (codes 0x71 / 0x81 -> IFA_ERR_STATE ->
waits off -> the same call through the launches that do not wait).  Runs in a child process: the downgrade is process-wide."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_split_k_wait_times_out_fails_the_call_and_switches_the_waiting_launches_off():
    p = subprocess.run([sys.executable, os.path.join(HERE, "wait_timeout_worker.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "wait timeout ok" in p.stdout, (p.stdout[-1500:], p.stderr[-1500:])
