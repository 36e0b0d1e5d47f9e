"""`python bench.py --gpus 2` started bare -- the way the round-end driver starts the one-GPU run --
must come up as two ranks on its own and print ONE result line for both (rehearsed here with both
ranks on one GPU and gloo; on a multi-GPU node the same command runs RCCL, one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

from wavegen import ROOT

pytestmark = pytest.mark.gpu


def test_bench_gpus_2_starts_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo",
                        "--steps", "3", "--warmup", "1", "--packets", "256", "--elems", "16384"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["achieved"] > 0       # per-GPU kernel figures
    assert out["extra"]["c5_packets"] == 2 * 131072                                    # the exchange step ran
