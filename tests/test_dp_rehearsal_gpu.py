"""GPU: the data-parallel schedule of bench.py at one rank over RCCL (DL4J_AMD_FORCE_COLLECTIVES=1).

Two forms: the whole step in one captured graph with the collectives inside it, and the multi-rank form - the capture cut at bucket
boundaries with eager collectives between the graph segments (DL4J_AMD_DP_CUTS_AT_ONE_RANK=1), which is what N > 1 runs. Each in its own
process (a process group and captured graphs should not leak into the other tests): the run must end with exit code 0, a finite score,
all buckets launched every step and the parameter checksum in agreement."""
import json
import math
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("cuts", ["0", "1"])
def test_one_rank_rccl_rehearsal_of_the_data_parallel_step(cuts):
    env = dict(os.environ, DL4J_AMD_FORCE_COLLECTIVES="1", DL4J_AMD_DP_CUTS_AT_ONE_RANK=cuts, MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29560 + int(cuts)), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DL4J_AMD_DP_BUCKET_MB="8")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "4", "--warmup", "3", "--batch", "8", "--image", "224",
                        "--classes", "100"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert math.isfinite(out["config"]["final_score"])
    c = out["comm"]
    assert c["backend"] == "rccl" and c["world_size_seen"] == 1 and c["params_in_sync"] is True
    assert c["buckets"] >= 2 and c["collectives_per_step"] == c["buckets"], c
