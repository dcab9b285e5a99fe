"""bench.py's watchdog: a rank that makes no progress within a phase's limit ends the run with ONE
JSON line carrying an "error" key on rank 0's stdout and a non-zero exit code -- a hang in communicator setup or in a
collective must not cost the driver its whole scaling record."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, time, argparse
sys.path.insert(0, %r)
import bench
a = argparse.Namespace(steps=4, warmup=1)
dog = bench.Watchdog(int(sys.argv[1]), 2, a)
dog.arm("pretend collective", 1.0)
time.sleep(30)
print("not reached")
"""


def _run(rank):
    p = subprocess.run([sys.executable, "-c", SCRIPT % ROOT, str(rank)], capture_output=True, text=True, timeout=60)
    return p.returncode, p.stdout, p.stderr


def test_watchdog_prints_one_json_error_line_on_rank0_and_exits():
    rc, out, err = _run(0)
    assert rc == 3
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and "pretend collective" in d["error"] and d["n_gpus"] == 2
    assert "watchdog" in err


def test_watchdog_on_other_ranks_exits_without_a_json_line():
    rc, out, err = _run(1)
    assert rc == 3 and out.strip() == "" and "watchdog" in err


def test_disarmed_watchdog_lets_the_run_finish():
    script = SCRIPT.replace('dog.arm("pretend collective", 1.0)\ntime.sleep(30)', 'dog.arm("x", 1.0); dog.disarm(); time.sleep(2)')
    p = subprocess.run([sys.executable, "-c", script % ROOT, "0"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0 and "not reached" in p.stdout


def test_reference_cpu_baseline_picks_the_depth_the_host_can_hold():
    """bench.py's cpu_baseline is the reference's CPU path on ALL 32 layers when the host can hold the 26.4 GB F32 checkpoint twice
    (file + the reference's heap), else the largest slice that fits -- and says which"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.pick_reference_layers(avail_gb=3000, disk_gb=500) == 32
    assert b.pick_reference_layers(avail_gb=62, disk_gb=118) == 32
    assert b.pick_reference_layers(avail_gb=30, disk_gb=118) == 16
    assert b.pick_reference_layers(avail_gb=18, disk_gb=118) == 8
    assert b.pick_reference_layers(avail_gb=8, disk_gb=118) == 0
    assert b.pick_reference_layers(avail_gb=3000, disk_gb=20) == 16      # the file itself must fit the temporary directory
