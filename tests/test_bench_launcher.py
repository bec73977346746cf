"""CPU tests of bench.py's own multi-rank entry (no GPU): `python bench.py --gpus 2` with WORLD_SIZE unset must start
its ranks itself (torch.distributed.run as a child of a parent that never touches the GPU), run the exchange step the
way the timed loop does -- shard -> pipeline records -> background gather on alternating buffer sets -> wait before the
set is rewritten -- and report the number of ranks that really joined; a rank-count mismatch is an error, not a warning.
The stand-in pipeline (bench.StubPipeline, gloo) publishes records that encode (rank, step), rank 0 checks them."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_gpus2_launches_its_own_ranks_gloo():
    r = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--frames", "3", "--backend", "gloo", "--stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["stub"] is True and out["n_gpus"] == 2 and out["gather_check"] == "ok" and out["exchanges"] == 7
    assert out["exchange"] == "packed"
    packed_bytes = out["bytes_per_rank_per_step"]
    # the capacity-sized exchange of round 2 moves more: 3 frames x 6 slots x (28 + 32 + 16) bytes + the fixed records
    r2 = _run(["--gpus", "2", "--steps", "5", "--warmup", "2", "--frames", "3", "--backend", "gloo", "--stub", "--exchange", "padded"])
    assert r2.returncode == 0, r2.stderr[-2000:]
    out2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert out2["gather_check"] == "ok" and out2["exchange"] == "padded" and 0 < packed_bytes < out2["bytes_per_rank_per_step"]


def test_bench_rank_count_mismatch_is_an_error():
    # a launcher that provides fewer ranks than --gpus asks for: one rank, WORLD_SIZE=1
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "2", "--backend", "gloo", "--stub"],
             env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "rank(s) joined" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_parent_does_not_initialise_the_gpu():
    # the launcher path returns before any torch.cuda call: statically, launch_ranks() and the code in front of it
    # contain none
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[src.index("def main("):src.index("rank, world, local = tbd.init_from_env")]
    assert "torch.cuda" not in head and "launch_ranks" in head
    body = src[src.index("def launch_ranks("):src.index("# ------------------------------------------------------------------ CPU stand-in")]
    assert "torch.cuda" not in body and "os.exec" not in body and "subprocess.call" in body
