"""GPU: bench.py's contract line, and its N=2 path (two ranks sharing the one GPU of this box, collectives on gloo) against the
N=1 run of the same workload: region sharding must not change what is counted or called."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, **(env or {})), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_contract_line_and_sharded_run():
    one = run_bench(["--reads", "3e5", "--steps", "2", "--warmup", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in one, k
    assert one["n_gpus"] == 1 and one["steps"] == 2 and one["unit"] == "sites/s" and one["vs_baseline"] is None
    r = one["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    cb = one["cpu_baseline"]
    assert cb["kind"] == "port" and 1 <= cb["cores"] <= 16 and cb["value"] > 0 and cb["gpu_matches_oracle_on_sample"] is True
    assert cb["native_1core"]["value"] > 0 and cb["pyloop"]["value_1core"] > 0 and cb["pyloop"]["value_allcores"] > 0 and cb["pyloop"]["cores"] == cb["cores"]
    assert "traffic_stale" in r and (r["traffic"] is None or r["traffic_stale"] is False)
    assert abs(one["value"] - one["config"]["sites_counted"] / (one["ms_per_step"] / 1e3)) < 1e-3 * one["value"]

    two = run_bench(["--gpus", "2", "--reads", "3e5", "--steps", "2", "--warmup", "1"], env={"LSG_BENCH_DEVICE": "0", "LSG_BENCH_BACKEND": "gloo"},
                    launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", "29533"])
    assert two["n_gpus"] == 2 and "cpu_baseline" not in two
    for k in ("sites_counted", "rows_emitted", "merged_sites", "step1_candidates"):
        assert two["config"][k] == one["config"][k], k
    assert two["config"]["reads_loaded_all_ranks"] >= one["config"]["reads_loaded_all_ranks"]          # boundary-crossing reads are loaded twice
    # the single fixed-capacity all-gather: a capacity that is too small at first is grown in warm-up and moves the same rows
    assert two["config"]["pass_rows_gathered"] is not None and two["config"]["pass_rows_gathered"] >= 0
    # ... and started the documented way: `python bench.py --gpus 2` with no WORLD_SIZE launches its own ranks (a child torch.distributed.run)
    tight = run_bench(["--gpus", "2", "--reads", "3e5", "--steps", "1", "--warmup", "1"],
                      env={"LSG_BENCH_DEVICE": "0", "LSG_BENCH_BACKEND": "gloo", "LSG_BENCH_GATHER_CAP": "1"})
    assert tight["n_gpus"] == 2 and "gloo" in tight["config"]["exchange"]
    assert tight["config"]["pass_rows_gathered"] == two["config"]["pass_rows_gathered"]


def test_rccl_code_path_with_one_rank():
    """RCCL cannot host two ranks on one GPU, but a ONE-rank nccl group can: the bench's collective code (device tensors through
    all_gather_into_tensor, barrier, all_reduce on backend nccl = RCCL) runs for real, and moves the rows the gloo rehearsal moves"""
    forced = run_bench(["--reads", "3e5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env={"LSG_BENCH_FORCE_DIST": "1"})
    plain = run_bench(["--reads", "3e5", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert forced["n_gpus"] == 1 and "nccl" in forced["config"]["exchange"] and forced["config"]["pass_rows_gathered"] is not None
    for k in ("sites_counted", "rows_emitted", "merged_sites", "step1_candidates"):
        assert forced["config"][k] == plain["config"][k], k


def test_product_allgather_over_rccl_with_one_rank(tmp_path):
    """regions.Comm on backend nccl with a single rank: the product's byte all-gather (candidate rows of the sharded SNV run)"""
    import subprocess, textwrap
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", LSG_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
        from longsom_amd import regions
        comm = regions.Comm.from_env()
        assert comm.device is not None and comm.device.type == "cuda"
        got = comm.allgather_bytes(regions.pack_rows({("chr1", 7): "row\\n"}))
        comm.barrier()
        assert regions.unpack_rows(got) == "row\\n"
        comm.close()
        print("rccl one-rank ok")
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stderr[-2000:]


def test_end_to_end_leg_is_measured_in_the_run():
    """config.end_to_end is a measurement of this run (a BAM written in the run, files in -> files out) or null: never a quoted file"""
    one = run_bench(["--reads", "3e5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--e2e-reads", "2e4"])
    e = one["config"]["end_to_end"]
    assert e["measured"] == "in this run" and e["wall_s"] > 0 and e["step3_rows"] >= 0 and e["out_MB"]["step1"] > 0
    assert {"gpu_count_call", "step2", "step3"} <= set(e["seconds"])
    none = run_bench(["--reads", "3e5", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert none["config"]["end_to_end"] is None
