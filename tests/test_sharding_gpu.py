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
