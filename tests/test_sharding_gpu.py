"""The N > 1 path at the BOUNDARY on real hardware: two ranks (one process each, gloo for the barrier and the timing reduce) share one shared
file list and call cs_batch_compress on their shards -- on the ONE GPU of the test box (--same-device), which is the code path of N ranks
on N GPUs without needing them.  The driver's SCALE run does the N-GPU measurement; tests/test_sharding_gloo.py covers the host logic on CPU."""
import json
import os
import subprocess
import sys

import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_share_one_list_at_the_boundary():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--boundary", "--same-device", "--boundary-total", "192", "--unique", "16"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["files"] == 192 and line["ok"] == 192 and line["scaling"] == "strong" and line["value"] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--boundary", "--boundary-total", "192", "--unique", "16"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-800:]
    assert json.loads(one.stdout.decode().strip().splitlines()[-1])["out_bytes"] == line["out_bytes"]   # the shards' outputs add up to the unsharded run's


def test_the_default_invocations_scaling_legs_run_on_one_device(monkeypatch):
    """the records `bench.py --gpus N` adds beside the resident number (round 5): the strong-scaling boundary leg (here with one rank) and rank 0's
    `caesiumclt --gpus N` over the shared list -- N = 2 device slots on the one GPU of this box (CSH_CLI_SAME_DEVICE), so that the code a multi-GPU
    sweep runs has run before the sweep does"""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    from _util import package, product_api
    from gen_synth import synth_jpeg
    api, pkg = product_api(), package()
    args = argparse.Namespace(unique=8, boundary_total=96)
    monkeypatch.setattr(bench, "make_inputs", lambda first, count: [synth_jpeg(first + i, 320, 240, texture=10) for i in range(count)])
    rec = bench.boundary_strong_leg(api, pkg, args, 0, 1, 0, lambda: None, lambda mx, sums: (mx, sums))
    assert rec["files"] == 96 and rec["ok"] == 96 and rec["value"] > 0 and rec["scaling"] == "strong"
    monkeypatch.setenv("CSH_CLI_SAME_DEVICE", "1")
    blobs = [synth_jpeg(i, 320, 240, texture=10) for i in range(8)]
    cli = bench.cli_gpus_leg(blobs, 64, 2)
    assert cli.get("files_written") == 64 and cli["n_gpus"] == 2 and cli["seconds"] > 0, cli

